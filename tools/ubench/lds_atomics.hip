// lds_atomics.hip -- what an LDS float atomic costs on gfx950 (VERDICT r3 #6 ii: the "~800 cycles per ds_add_f32" of
// profiles/r03_bwd_mapping_study.md was measured once and its case was skipped in the committed tool).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomics lds_atomics.hip && ./lds_atomics
// Every case: one wave64 per workgroup, W workgroups per SIMD resident (grid = 1024 W), ITER x 16 wave-instructions of the
// kind under test, s_waitcnt lgkmcnt(0) every 16.  Cycles are per wave-instruction per SIMD at the nominal 2.4 GHz.
//   same   : all 16 instructions of a lane hit the SAME word (the r03 measurement: a read-modify-write chain on one address)
//   rot    : 16 different words per lane (offset i * 272 bytes), all 64 lanes distinct words, conflict-free banks
//   rtn    : returning form (ds_add_rtn_f32), rotating words
//   conf64 : all 64 lanes add to ONE word (the worst case of an in-LDS reduction)
//   conf8  : 8 lanes per word
//   hip    : atomicAdd(&lds[i], x) as the compiler emits it with -munsafe-fp-atomics (ISA: ds_add_f32, see the dump)
//   hip_cas: the same source built WITHOUT -munsafe-fp-atomics semantics (explicit CAS loop) for comparison
//   write  : ds_write_b32 rotating words (the store the transposed reduction of ggs_render_common.h uses)
//   u32    : ds_add_u32 rotating words
//   block  : the transposed reduction itself: 4 ds_write2_b32 + 2 ds_read2_b64 + 7 v_add + 3 DPP adds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define ITER 4096
#define LDS_WORDS (16 * 68 + 64)

template <int K>
__global__ __launch_bounds__(64) void k_rate(float* out, float seed) {
    __shared__ float lds[LDS_WORDS];
    for (int i = threadIdx.x; i < LDS_WORDS; i += 64) lds[i] = 0.f;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    const unsigned base = (unsigned)(uintptr_t)lds;
    const unsigned addr = base + threadIdx.x * 4;               // lane-distinct word of row 0; rows are 68 words = 272 B apart
    const unsigned addr1 = base;                                 // one word for everybody
    const unsigned addr8 = base + (threadIdx.x >> 3) * 4;        // 8 lanes per word
    float r0 = 0.f;
    __syncthreads();
    for (int it = 0; it < ITER; ++it) {
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
        if (K == 0) {
#define OP(i) asm volatile("ds_add_f32 %0, %1" :: "v"(addr), "v"(a[i & 7]) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 1) {
#define OP(i) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(addr), "v"(a[i & 7]), "n"(i * 272) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 2) {
#define OP(i) asm volatile("ds_add_rtn_f32 %0, %1, %2 offset:%3" : "=v"(r0) : "v"(addr), "v"(a[i & 7]), "n"(i * 272) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 3) {
#define OP(i) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(addr1), "v"(a[i & 7]), "n"(i * 272) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 4) {
#define OP(i) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(addr8), "v"(a[i & 7]), "n"(i * 272) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&lds[i * 68 + threadIdx.x], a[i & 7]);
        } else if (K == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {       // what a CAS loop costs (explicit, for comparison)
                unsigned* p = reinterpret_cast<unsigned*>(&lds[i * 68 + threadIdx.x]);
                unsigned old = *p, assumed;
                do { assumed = old; old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + a[i & 7])); } while (old != assumed);
            }
        } else if (K == 7) {
#define OP(i) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(a[i & 7]), "n"(i * 272) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 8) {
#define OP(i) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(addr), "v"(a[i & 7]), "n"(i * 272) : "memory");
            REP16(OP)
#undef OP
        } else if (K == 9) {
            // two transposed reductions of 8 values each (= 16 "value-instructions"): plane [8][80] words
            const unsigned rd = base + ((threadIdx.x >> 3) * 80 + (threadIdx.x & 7) * 2) * 4;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                float4 p, q;
                asm volatile("ds_write2_b32 %2, %3, %4 offset1:80\n\t"
                             "ds_write2_b32 %2, %5, %6 offset0:160 offset1:240\n\t"
                             "ds_write2_b32 %2, %7, %8 offset0:64 offset1:144\n\t"        // (rows 4..7 through a second base in the kernel; same cost)
                             "ds_write2_b32 %2, %9, %10 offset0:96 offset1:176\n\t"
                             "ds_read2_b64 %0, %11 offset1:8\n\t"
                             "ds_read2_b64 %1, %11 offset0:16 offset1:24\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(p), "=v"(q)
                             : "v"(addr), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(rd)
                             : "memory");
                float s = ((p.x + p.y) + (p.z + p.w)) + ((q.x + q.y) + (q.z + q.w));
                s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xf, 0xf, true));
                s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xf, 0xf, true));
                s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xf, 0xf, true));
                a[rep] += s * 1e-9f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = r0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    __syncthreads();
    if (s == 12345.678f) out[0] = lds[threadIdx.x];
    if (blockIdx.x == 0 && K != 9) out[64 + threadIdx.x] = lds[threadIdx.x];      // functional check of the adds
}

int main() {
    float* d_f; CHECK(hipMalloc(&d_f, 4096));
    const char* names[10] = {"ds_add_f32 same word x16 (r03 case)", "ds_add_f32 rotating words", "ds_add_rtn_f32 rotating words",
                             "ds_add_f32 64 lanes -> one word", "ds_add_f32 8 lanes per word", "atomicAdd(float) as compiled",
                             "explicit CAS loop", "ds_write_b32 rotating words", "ds_add_u32 rotating words",
                             "transposed reduction block (8 values; per value)"};
    for (int K = 0; K < 10; ++K)
        for (int w : {1, 2, 4, 6}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                const dim3 g(1024 * w), b(64);
                switch (K) {
                    case 0: hipLaunchKernelGGL(k_rate<0>, g, b, 0, 0, d_f, 1.0f); break;
                    case 1: hipLaunchKernelGGL(k_rate<1>, g, b, 0, 0, d_f, 1.0f); break;
                    case 2: hipLaunchKernelGGL(k_rate<2>, g, b, 0, 0, d_f, 1.0f); break;
                    case 3: hipLaunchKernelGGL(k_rate<3>, g, b, 0, 0, d_f, 1.0f); break;
                    case 4: hipLaunchKernelGGL(k_rate<4>, g, b, 0, 0, d_f, 1.0f); break;
                    case 5: hipLaunchKernelGGL(k_rate<5>, g, b, 0, 0, d_f, 1.0f); break;
                    case 6: hipLaunchKernelGGL(k_rate<6>, g, b, 0, 0, d_f, 1.0f); break;
                    case 7: hipLaunchKernelGGL(k_rate<7>, g, b, 0, 0, d_f, 1.0f); break;
                    case 8: hipLaunchKernelGGL(k_rate<8>, g, b, 0, 0, d_f, 1.0f); break;
                    default: hipLaunchKernelGGL(k_rate<9>, g, b, 0, 0, d_f, 1.0f); break;
                }
            };
            launch();
            hipEventRecord(e0); launch(); hipEventRecord(e1); CHECK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)ITER * 16 * w;     // wave-instructions per SIMD (1024 SIMDs, 1024 w waves)
            float h[128]; CHECK(hipMemcpy(h, d_f, sizeof(h), hipMemcpyDeviceToHost));
            printf("%-52s W=%d  %8.3f ms  %7.2f cyc @2.4GHz per wave-instr per SIMD   lds[0]=%g lds[1]=%g\n", names[K], w, ms,
                   ms * 1e6 / n * 2.4, h[64], h[65]);
        }
    return 0;
}
