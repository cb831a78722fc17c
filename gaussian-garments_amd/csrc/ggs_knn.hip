// ggs_knn.hip -- mean squared distance to the 3 nearest neighbours of every point.
//
// Replaces `simple_knn._C.distCUDA2`, the reference's second external CUDA dependency
// (scene/gaussian_model.py:20,135 and scene/mesh_gaussian_model.py:23,233; built by setup.sh:35-37 from
// graphdeco-inria/gaussian-splatting/submodules/simple-knn).  It is called once, at model initialisation,
// to size the initial Gaussians: scales = log(sqrt(clamp_min(distCUDA2(xyz), 1e-7))).
//
// Upstream sorts the points along a Morton curve and searches neighbouring boxes; the result is the exact
// 3-NN mean (self excluded).  On MI355X the exact answer is cheaper to get by brute force: one lane per
// query, all P candidates streamed through LDS in 1024-point tiles and read back as wave-uniform
// (broadcast) ds_read_b128 -- P = 100k is 1e10 pair tests = ~1.3 ms, with no sort, no tree and no
// tuning parameter.  VALU-bound; HBM traffic is P * 12 B per workgroup, L2-resident.
#include "ggs_kernels.h"

namespace {

#define KNN_TILE 1024

__device__ __forceinline__ void insert3(float d, float& b0, float& b1, float& b2) {
    // keep the three smallest of {b0 <= b1 <= b2, d}
    const float t0 = fminf(b0, d), u0 = fmaxf(b0, d);
    const float t1 = fminf(b1, u0), u1 = fmaxf(b1, u0);
    b0 = t0; b1 = t1; b2 = fminf(b2, u1);
}

__global__ __launch_bounds__(256) void k_dist2_3nn(int P, const float* __restrict__ pts, float* __restrict__ out) {
    __shared__ float4 s_p[KNN_TILE];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < P ? i : P - 1;
    const float qx = pts[3 * (size_t)ii], qy = pts[3 * (size_t)ii + 1], qz = pts[3 * (size_t)ii + 2];
    float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;
    for (int base = 0; base < P; base += KNN_TILE) {
        const int n = min(KNN_TILE, P - base);
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += 256) {
            const size_t j = (size_t)(base + t);
            s_p[t] = make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.f);
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < n; ++t) {
            const float4 c = s_p[t];
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            float d = dx * dx + dy * dy + dz * dz;
            d = (base + t == i) ? 3.402823466e+38f : d;       // a point is not its own neighbour
            insert3(d, b0, b1, b2);
        }
    }
    if (i < P) out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

extern "C" int ggs_dist2_3nn(int P, const float* points, float* out, void* stream) {
    ggs_clear_error_();
    if (P < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_dist2_3nn: bad size");
    if (P == 0) return GGS_OK;
    if (!points || !out) return ggs_fail_(GGS_ERR_ARG, "ggs_dist2_3nn: NULL pointer argument");
    hipLaunchKernelGGL(k_dist2_3nn, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, points, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "dist2_3nn launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}
