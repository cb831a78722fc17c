// valu_rate.hip -- issue-rate microbenchmark of the VALU / LDS instructions the render kernels are made of (gfx950).
// For every instruction kind: each wave runs ITER iterations of 16 independent instances (8 register chains x 2),
// timed with s_memtime inside the wave; grid = 1024 SIMDs x W waves per SIMD.  Output: shader cycles per wave-instruction
// per SIMD = (cycles of the slowest wave) * 1 / (ITER * 16 * W)  -- i.e. the throughput one SIMD sustains.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define ITER 2048

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { FMA, MUL, ADD, PK_FMA, PK_MUL, PK_ADD, EXP, RCP, CMP, CNDMASK, MINF, DPP_ADD, PERMSWAP32, MOV, READLANE,
            LDS_B128, FMA_DEP, MIX_FMA_EXP, MIX_FMA_CMP, SALU_AND, MIX_FMA_SALU, MIX_FMA_LDS, CVT_I2F, ADD_U32, KINDS };
static const char* kname[KINDS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
    "v_exp_f32", "v_rcp_f32", "v_cmp_le_f32 (sgpr dst)", "v_cndmask_b32 (sgpr mask)", "v_min_f32", "v_add_f32_dpp row_ror",
    "v_permlane32_swap", "v_mov_b32", "v_readlane_b32", "ds_read_b128 (broadcast)", "v_fma_f32 dependent chain",
    "1 v_fma + 1 v_exp pairs (per pair)", "1 v_fma + 1 v_cmp pairs (per pair)", "s_and_b64", "1 v_fma + 1 s_and pairs (per pair)",
    "4 v_fma + 1 ds_read_b128 (per group of 5)", "v_cvt_f32_i32", "v_add_u32"};

template <int K>
__global__ __launch_bounds__(64) void bench(uint64_t* out, float seed) {
    __shared__ float4 lds[64];
    lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    typedef float v2 __attribute__((ext_vector_type(2)));
    float a[8], b = seed + 1.f, c = seed * 0.5f;
    v2 p[8], pb = {b, c}, pc = {c, b};
    uint64_t m[8];
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2{a[i], a[i] + 1.f}; m[i] = 0x5555555555555555ull + i; u[i] = threadIdx.x + i; }
    float4 l4[4] = {};
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
#define DO(i)                                                                                                            \
    if (K == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                                   \
    if (K == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                               \
    if (K == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                               \
    if (K == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));                           \
    if (K == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                        \
    if (K == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                                        \
    if (K == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                                            \
    if (K == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                                            \
    if (K == CMP) asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(m[i]) : "v"(a[i]), "v"(b));                                 \
    if (K == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(m[i]));                    \
    if (K == MINF) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                              \
    if (K == DPP_ADD) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));           \
    if (K == PERMSWAP32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));                   \
    if (K == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));                                                   \
    if (K == READLANE) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(u[i]) : "v"(a[i]));                                   \
    if (K == LDS_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[i & 3]) : "v"(0u) : "memory");                          \
    if (K == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));                               \
    if (K == MIX_FMA_EXP) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_exp_f32 %1, %1" : "+v"(a[i]), "+v"(p[i].x) : "v"(b), "v"(c)); \
    if (K == MIX_FMA_CMP) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_cmp_le_f32 %1, %0, %2" : "+v"(a[i]), "=s"(m[i]) : "v"(b), "v"(c)); \
    if (K == SALU_AND) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[i]) : "s"(m[(i + 1) & 7]));                             \
    if (K == MIX_FMA_SALU) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_and_b64 %1, %1, %4" : "+v"(a[i]), "+s"(m[i]) : "v"(b), "v"(c), "s"(m[(i + 1) & 7])); \
    if (K == MIX_FMA_LDS) asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tds_read_b128 %2, %5" \
                                       : "+v"(a[i]), "+v"(p[i].x), "=v"(l4[i & 3]) : "v"(b), "v"(c), "v"(0u) : "memory");      \
    if (K == CVT_I2F) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));                                                    \
    if (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            R8(DO)
#undef DO
        }
        if (K == LDS_B128 || K == MIX_FMA_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)(m[i] & 1) + (float)u[i] + l4[i & 3].x;
    if (s == 12345.678f) out[0] = 0;          // keep everything live
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int K>
void run(uint64_t* d_out, std::vector<uint64_t>& h, int waves_per_simd) {
    const int grid = 1024 * waves_per_simd;
    hipLaunchKernelGGL(bench<K>, dim3(grid), dim3(64), 0, 0, d_out, 1.0f);   // warm-up
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<K>, dim3(grid), dim3(64), 0, 0, d_out, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_out, grid * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::vector<uint64_t> v(h.begin(), h.begin() + grid);
    std::sort(v.begin(), v.end());
    const double per_wave = (double)v[grid / 2] / (ITER * 16.0);            // s_memtime ticks per instruction, median wave
    // s_memtime runs at a constant 100 MHz on gfx9: convert through the kernel's wall time instead
    const double n_inst_simd = (double)ITER * 16.0 * waves_per_simd;
    printf("%-44s W=%d  kernel %8.3f ms  ns per wave-instr per SIMD %7.3f  (= %5.2f cyc @2.4GHz)  memtime ticks/instr/wave %7.3f\n",
           kname[K], waves_per_simd, ms, ms * 1e6 / n_inst_simd, ms * 1e6 / n_inst_simd * 2.4, per_wave);
}

template <int K>
void sweep(uint64_t* d_out, std::vector<uint64_t>& h) {
    for (int w : {1, 2, 4, 8}) run<K>(d_out, h, w);
}

int main() {
    uint64_t* d_out;
    hipMalloc(&d_out, 8192 * sizeof(uint64_t));
    std::vector<uint64_t> h(8192);
    sweep<FMA>(d_out, h); sweep<MUL>(d_out, h); sweep<ADD>(d_out, h); sweep<PK_FMA>(d_out, h); sweep<PK_MUL>(d_out, h);
    sweep<PK_ADD>(d_out, h); sweep<EXP>(d_out, h); sweep<RCP>(d_out, h); sweep<CMP>(d_out, h); sweep<CNDMASK>(d_out, h);
    sweep<MINF>(d_out, h); sweep<DPP_ADD>(d_out, h); sweep<PERMSWAP32>(d_out, h); sweep<MOV>(d_out, h); sweep<READLANE>(d_out, h);
    sweep<LDS_B128>(d_out, h); sweep<FMA_DEP>(d_out, h); sweep<MIX_FMA_EXP>(d_out, h); sweep<MIX_FMA_CMP>(d_out, h);
    // The SALU cases (SALU_AND, MIX_FMA_SALU) hung the GPU box's run for its whole time limit when first collected and
    // are left out; the three cases behind them were never reached and are untested.
    // sweep<SALU_AND>(d_out, h); sweep<MIX_FMA_SALU>(d_out, h); sweep<MIX_FMA_LDS>(d_out, h); sweep<CVT_I2F>(d_out, h); sweep<ADD_U32>(d_out, h);
    return 0;
}
