// mfma_f32.hip -- v_mfma_f32_16x16x4_f32 on gfx950: (1) operand / result lane layout, (2) does it issue beside plain VALU work?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32 mfma_f32.hip && ./mfma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));

// D = A (16x4) * B (4x16).  Hypothesis: A[i][k] from lane 16 k + i, B[k][j] from lane 16 k + j, D[4 (l / 16) + r][l % 16] in reg r.
__global__ void k_layout(float* out) {
    const int l = threadIdx.x;
    const int i = l & 15, k = l >> 4;
    const float a = (float)(100 * i + k);          // A[i][k] = 100 i + k
    const float b = (float)(k == 0 ? 1 : 0) * (float)(l & 15) + (k == 1 ? 1000.f : 0.f);   // B[0][j] = j, B[1][j] = 1000, B[2..3][j] = 0
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

#define ITER 4096
template <int MODE>   // 0: MFMA only, 1: v_fma only, 2: both interleaved (4 v_fma per MFMA), 3: both (8 per MFMA), 4: v_exp + MFMA (1 per MFMA), 5: ds_read_b32 + MFMA
__global__ __launch_bounds__(64) void k_rate(float* out, float seed) {
    __shared__ float lds[256];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 64] = seed;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float f[8];
    v4f acc[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = seed + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const unsigned addr = (unsigned)(uintptr_t)lds + threadIdx.x * 4;
    float ld0 = 0.f;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0 || MODE >= 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
                for (int i = 0; i < (MODE == 3 ? 8 : 4); ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[(u * 4 + i) & 7]) : "v"(a), "v"(b));
            }
            if (MODE == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(f[u]));
            if (MODE == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(ld0) : "v"(addr) : "memory");
        }
        if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = ld0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = sum_k A[i][k] B[k][j] = (100 i) j + (100 i + 1) 1000
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l / 16) + r, j = l % 16;
        const float e = (100.f * i) * j + (100.f * i + 1.f) * 1000.f;
        if (h[l * 4 + r] != e) { if (bad < 6) printf("layout mismatch lane %d reg %d: got %g expected %g\n", l, r, h[l * 4 + r], e); ++bad; }
    }
    printf("layout hypothesis (A[i][k] lane 16k+i, B[k][j] lane 16k+j, D[4(l/16)+r][l%%16]): %s\n", bad ? "WRONG" : "confirmed");
    const char* nm[6] = {"v_mfma_f32_16x16x4_f32 alone", "v_fma_f32 alone (4 per slot)", "MFMA + 4 v_fma per MFMA", "MFMA + 8 v_fma per MFMA", "MFMA + 1 v_exp per MFMA", "MFMA + 1 ds_read_b32 per MFMA"};
    for (int m = 0; m < 6; ++m)
        for (int w : {1, 2, 4}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (m == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1024 * w), dim3(64), 0, 0, d, 1.0f);
                if (m == 1) hipLaunchKernelGGL(k_rate<1>, dim3(1024 * w), dim3(64), 0, 0, d, 1.0f);
                if (m == 2) hipLaunchKernelGGL(k_rate<2>, dim3(1024 * w), dim3(64), 0, 0, d, 1.0f);
                if (m == 3) hipLaunchKernelGGL(k_rate<3>, dim3(1024 * w), dim3(64), 0, 0, d, 1.0f);
                if (m == 4) hipLaunchKernelGGL(k_rate<4>, dim3(1024 * w), dim3(64), 0, 0, d, 1.0f);
                if (m == 5) hipLaunchKernelGGL(k_rate<5>, dim3(1024 * w), dim3(64), 0, 0, d, 1.0f);
            };
            launch(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s W=%d  %7.3f ms  %6.1f cyc @2.4GHz per slot (1 MFMA [+ fillers]) per SIMD\n", nm[m], w, ms, ms * 1e6 / (ITER * 4.0 * w) * 2.4);
        }
    return 0;
}
