// ggs_adam.hip -- the optimiser update of the registration / appearance loops as kernels that can live inside a
// captured hipGraph.  Reference: torch.optim.Adam(l, lr=0.0, eps=1e-15) built at scene/mesh_gaussian_model.py:375
// (gaussian_model.py:165, avatar_net.py:50) and stepped at s2_registration.py:316-318, s3_appearance.py:143-145.
//
// Why native: torch's Adam reads its step count and learning rates on the host, so an optimisation step cannot be
// replayed as a graph without re-capturing whenever the xyz schedule moves -- and it cannot be made conditional.
// Here the step count / bias corrections and the learning rate live in device memory, and every kernel honours a
// device-side guard word (the binning-overflow flag of the forward of the same step): a replayed step whose
// rasterization overflowed its static binning capacity leaves parameters and moments untouched, the host grows
// the capacity and replays.  HBM-bound, 28 B per element: 16-byte accesses, grid-stride.
#include "ggs_kernels.h"

namespace {

struct AdamState { long long step; float bias1; float bias2_sqrt; };   // 16 bytes, device

__global__ void k_adam_tick(AdamState* s, double beta1, double beta2, const unsigned long long* guard) {
    if (guard && *guard) return;
    const long long t = s->step + 1;
    s->step = t;
    s->bias1 = (float)(1.0 - pow(beta1, (double)t));                   // torch computes these in Python floats
    s->bias2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)t));
}

struct AdamArgs {
    size_t n;
    float* p; const float* g; float* m; float* v;
    const float* lr; const AdamState* s; const unsigned long long* guard;
    float beta1, beta2, omb1, omb2, eps;    // omb = 1 - beta rounded from double, as torch passes it
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a, float step_size,
                                      float bias2_sqrt) {
    const float eps = a.eps;
    m = fmaf(a.omb1, g - m, m);                                         // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(a.omb2, g * g, v * a.beta2);                               // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p = fmaf(-step_size, m / denom, p);                                 // param.addcdiv_(exp_avg, denom, -step_size)
}

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    if (a.guard && *a.guard) return;
    const float step_size = *a.lr / a.s->bias1, bs = a.s->bias2_sqrt;
    const size_t n4 = a.n / 4, stride = (size_t)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam1(p.x, g.x, m.x, v.x, a, step_size, bs);
        adam1(p.y, g.y, m.y, v.y, a, step_size, bs);
        adam1(p.z, g.z, m.z, v.z, a, step_size, bs);
        adam1(p.w, g.w, m.w, v.w, a, step_size, bs);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {                   // tail
        const size_t i = n4 * 4 + threadIdx.x;
        adam1(a.p[i], a.g[i], a.m[i], a.v[i], a, step_size, bs);
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

size_t ggs_adam_state_bytes(void) { return sizeof(AdamState); }

int ggs_adam_tick(void* state, double beta1, double beta2, const void* guard, void* stream) {
    ggs_clear_error_();
    if (!state) return ggs_fail_(GGS_ERR_ARG, "ggs_adam_tick: NULL state");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, static_cast<AdamState*>(state),
                       beta1, beta2, static_cast<const unsigned long long*>(guard));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "adam_tick launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

int ggs_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lr,
                  double beta1, double beta2, double eps, const void* state, const void* guard, void* stream) {
    ggs_clear_error_();
    if (n == 0) return GGS_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !lr || !state)
        return ggs_fail_(GGS_ERR_ARG, "ggs_adam_step: NULL pointer argument");
    if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
        return ggs_fail_(GGS_ERR_ARG, "ggs_adam_step: tensors must be 16-byte aligned");
    AdamArgs a;
    a.n = n; a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.lr = lr;
    a.s = static_cast<const AdamState*>(state); a.guard = static_cast<const unsigned long long*>(guard);
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    size_t blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "adam_step launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

}  // extern "C"
