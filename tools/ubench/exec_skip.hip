// exec_skip.hip -- does a wave64 VALU instruction get cheaper when half (or three quarters) of EXEC is off?  (gfx950)
// Each wave runs ITER x 16 independent instructions under a fixed EXEC mask; time per instruction per SIMD, 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o exec_skip exec_skip.hip && ./exec_skip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 8192
template <int K>
__global__ __launch_bounds__(64) void bench(float* out, float seed, unsigned long long mask) {
    float a[8], b = seed + 1.f, c = seed * 0.5f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(saved) : "s"(mask));
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (K == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (K == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (K == 3) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
            }
        }
    }
    asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 64);
    const char* nm[4] = {"v_fma_f32", "v_exp_f32", "v_min_f32", "v_cmp_le_f32"};
    const unsigned long long masks[5] = {~0ull, 0xffffffffull, 0xffffull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull};
    const char* mn[5] = {"all 64", "low 32", "low 16", "rows 0,2", "8 of each 16"};
    for (int K = 0; K < 4; ++K)
        for (int m = 0; m < 5; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (K == 0) hipLaunchKernelGGL(bench<0>, dim3(4096), dim3(64), 0, 0, d, 1.0f, masks[m]);
                if (K == 1) hipLaunchKernelGGL(bench<1>, dim3(4096), dim3(64), 0, 0, d, 1.0f, masks[m]);
                if (K == 2) hipLaunchKernelGGL(bench<2>, dim3(4096), dim3(64), 0, 0, d, 1.0f, masks[m]);
                if (K == 3) hipLaunchKernelGGL(bench<3>, dim3(4096), dim3(64), 0, 0, d, 1.0f, masks[m]);
            };
            launch(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-14s exec %-14s %7.3f ms  %5.2f cyc @2.4GHz per wave-instr per SIMD\n", nm[K], mn[m], ms, ms * 1e6 / (ITER * 16.0 * 4) * 2.4);
        }
    return 0;
}
