// systolic_parts.hip -- what the parts of a splat-major (row-systolic) backward cost on gfx950:
//   (1) v_mov_b32_dpp row_ror:1 : direction check + issue rate
//   (2) ds_add_f32 with per-lane distinct addresses (9 per step), ds_read_b128 with per-lane addresses (stride 48 B)
//   (3) scattered global float atomics: 64 lanes -> 64 different 48-byte records of a [n_rec] array, 9 fields each
//       (the flush of a splat-major backward), against 9 lanes -> one record (the tile-major backward)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o systolic_parts systolic_parts.hip && ./systolic_parts
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_dir(int* out) {
    int v = threadIdx.x;
    int r = __builtin_amdgcn_update_dpp(-1, v, 0x121 /* row_ror:1 */, 0xf, 0xf, false);
    out[threadIdx.x] = r;
    int r2 = __builtin_amdgcn_update_dpp(-1, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    out[64 + threadIdx.x] = r2;
    int r3 = __builtin_amdgcn_update_dpp(-1, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    out[128 + threadIdx.x] = r3;
}

#define ITER 8192
template <int K>
__global__ __launch_bounds__(64) void k_rate(float* out, float seed) {
    __shared__ float lds[64 * 48 / 4 * 4 + 4096];
    for (int i = threadIdx.x; i < 64 * 12 * 4 + 4096; i += 64) lds[i] = seed;
    float a[8];
    float4 l4[4] = {};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    const unsigned addr_rec = (unsigned)(uintptr_t)lds + threadIdx.x * 48;            // per-lane 48-byte stride
    const unsigned addr_acc = (unsigned)(uintptr_t)lds + 64 * 48 + ((threadIdx.x >> 4) * 16 * 17 + (threadIdx.x & 15) * 17) * 4;
    __syncthreads();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (K == 0) asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (K == 1) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(addr_acc), "v"(a[i]), "n"(0) : "memory");
                if (K == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[i & 3]) : "v"(addr_rec) : "memory");
                if (K == 4) asm volatile("v_add_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(0.f));
                if (K == 5) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 3) & 7]));
                if (K == 6) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(a[i]) : "v"(a[(i + 3) & 7]));
                if (K == 7) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 3) & 7]));
                if (K == 3) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tds_add_f32 %2, %0" : "+v"(a[i]) : "v"(seed), "v"(addr_acc) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + l4[i & 3].x;
    if (s == 12345.678f) out[0] = lds[threadIdx.x];
}

// mode 0: every lane adds 9 fields to ITS OWN random record (9 instructions x 64 lanes)
// mode 1: 9 lanes add the 9 fields of ONE random record per instruction (1 instruction, 9 active lanes), 64 instructions
// mode 2: 63 lanes = 7 random records x 9 fields per instruction;  mode 3: as mode 1 with plain stores;  mode 4: as mode 0 with
// plain stores (3 x 16-byte: 48 contiguous bytes per lane);  stride = floats per record (12: 48-byte records, 16: one 64-byte line each)
__global__ __launch_bounds__(64) void k_atomic(float* acc, int n_rec, int mode, int iters, int stride) {
    uint32_t rng = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t wrng = blockIdx.x * 2246822519u + 777u;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
            rng = rng * 1664525u + 1013904223u;
            float* r = acc + (size_t)((rng >> 8) % (uint32_t)n_rec) * stride;
#pragma unroll
            for (int f = 0; f < 9; ++f) atomicAdd(r + f, 1.0f);
        } else if (mode == 4) {
            rng = rng * 1664525u + 1013904223u;
            float4* r = reinterpret_cast<float4*>(acc + (size_t)((rng >> 8) % (uint32_t)n_rec) * stride);
            r[0] = make_float4(1.f, 2.f, 3.f, 4.f); r[1] = make_float4(1.f, 2.f, 3.f, 4.f); r[2] = make_float4(1.f, 2.f, 3.f, 4.f);
        } else if (mode == 2) {
#pragma unroll 1
            for (int k = 0; k < 9; ++k) {
                wrng = wrng * 1664525u + 1013904223u;
                const uint32_t my = (wrng ^ ((threadIdx.x / 9) * 2654435761u)) >> 8;
                float* r = acc + (size_t)(my % (uint32_t)n_rec) * stride;
                if (threadIdx.x < 63) atomicAdd(r + threadIdx.x % 9, 1.0f);
            }
        } else {
#pragma unroll 1
            for (int k = 0; k < 64; ++k) {
                wrng = wrng * 1664525u + 1013904223u;
                float* r = acc + (size_t)((wrng >> 8) % (uint32_t)n_rec) * stride;
                if (mode == 3) { if (threadIdx.x < 9) r[threadIdx.x] = 1.0f; }
                else if (threadIdx.x < 9) atomicAdd(r + threadIdx.x, 1.0f);
            }
        }
    }
}

int main() {
    int* d_i; CHECK(hipMalloc(&d_i, 192 * 4));
    hipLaunchKernelGGL(k_dir, dim3(1), dim3(64), 0, 0, d_i);
    int h[192]; CHECK(hipMemcpy(h, d_i, sizeof(h), hipMemcpyDeviceToHost));
    printf("row_ror:1  lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane17<-%d\n", h[0], h[1], h[15], h[16], h[17]);
    printf("row_shr:1  lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane17<-%d\n", h[64], h[65], h[79], h[80], h[81]);
    printf("wave_shr:1 lane0<-%d lane1<-%d lane15<-%d lane16<-%d lane17<-%d\n", h[128], h[129], h[143], h[144], h[145]);
    float* d_f; CHECK(hipMalloc(&d_f, 4096));
    const char* names[8] = {"v_mov_b32_dpp row_ror:1", "ds_add_f32 (per-lane addresses, stride 17 words)", "ds_read_b128 (per-lane, stride 48 B)", "4 v_fma + 1 ds_add_f32 (per group)", "v_add_f32_dpp x, 0 row_ror:1", "v_mov_b32_dpp other dst row_ror:1", "v_mov_b32_dpp row_shr:1 bound_ctrl", "v_mov_b32_dpp quad_perm"};
    for (int K = 0; K < 8; ++K) {
        // K == 1 / 3 (ds_add_f32): also measured case by case in tools/ubench/lds_atomics.hip (profiles/r04_lds_atomic_rates.md)
        for (int w : {1, 2, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (K == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 2) hipLaunchKernelGGL(k_rate<2>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 3) hipLaunchKernelGGL(k_rate<3>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 4) hipLaunchKernelGGL(k_rate<4>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 5) hipLaunchKernelGGL(k_rate<5>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 6) hipLaunchKernelGGL(k_rate<6>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
                if (K == 7) hipLaunchKernelGGL(k_rate<7>, dim3(1024 * w), dim3(64), 0, 0, d_f, 1.0f);
            };
            launch();
            hipEventRecord(e0); launch(); hipEventRecord(e1); CHECK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)ITER * 16 * w;
            printf("%-52s W=%d  %8.3f ms  %6.2f cyc @2.4GHz per wave-instr(group) per SIMD\n", names[K], w, ms, ms * 1e6 / n * 2.4);
        }
    }
    const int n_rec = 100000;
    float* d_acc; CHECK(hipMalloc(&d_acc, (size_t)n_rec * 48 * 8));
    CHECK(hipMemset(d_acc, 0, (size_t)n_rec * 48 * 8));
    for (int stride : {12, 16})
    for (int nrec : {100000})
        for (int mode = 0; mode < 5; ++mode)
            for (int waves : {8192}) {
                const int iters = 64;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(k_atomic, dim3(waves), dim3(64), 0, 0, d_acc, nrec, mode, iters, stride);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_atomic, dim3(waves), dim3(64), 0, 0, d_acc, nrec, mode, iters, stride);
                hipEventRecord(e1); CHECK(hipDeviceSynchronize());
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double n_rec_done = (double)waves * iters * (mode == 2 ? 63 : 64);
                const char* mn[5] = {"atomics: 64 lanes x own record x 9 instr", "atomics: 9 lanes x one record per instr", "atomics: 63 lanes = 7 records x 9 fields per instr",
                                     "plain stores: 9 lanes x one record per instr", "plain stores: 64 lanes x own record, 3 x 16 B"};
                printf("mode %d (%s) stride %d B records %d waves %d: %8.3f ms, %6.1f records/ns\n", mode, mn[mode], stride * 4, nrec, waves, ms, n_rec_done / (ms * 1e6));
            }
    return 0;
}
