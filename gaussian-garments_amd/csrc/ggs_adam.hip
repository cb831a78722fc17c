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

struct TickMulti { int n; AdamState* s[16]; };

__global__ void k_adam_tick(AdamState* s, double beta1, double beta2, const unsigned long long* guard) {
    if (guard && *guard) return;
    ggs_adam_tick_one(s, beta1, beta2);
}
// one thread per tensor: torch keeps a step count PER PARAMETER (a parameter without gradient skips the step and its
// bias corrections lag behind), so every tensor has its own state
__global__ void k_adam_tick_multi(TickMulti m, double beta1, double beta2, const unsigned long long* guard) {
    if (guard && *guard) return;
    if ((int)threadIdx.x < m.n) ggs_adam_tick_one(m.s[threadIdx.x], beta1, beta2);
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

// All parameter tensors of an optimiser in ONE launch: at 100k Gaussians every tensor is a few hundred KB, so seven
// launches are seven launch latencies (~5 us each inside a graph) for ~6 us of memory traffic.
#define GGS_ADAM_MAX_TENSORS 16
struct AdamMulti {
    int n;
    unsigned first_block[GGS_ADAM_MAX_TENSORS + 1];     // workgroup range of tensor t: [first_block[t], first_block[t + 1])
    size_t numel[GGS_ADAM_MAX_TENSORS];
    float* p[GGS_ADAM_MAX_TENSORS]; const float* g[GGS_ADAM_MAX_TENSORS];
    float* m[GGS_ADAM_MAX_TENSORS]; float* v[GGS_ADAM_MAX_TENSORS];
    const float* lr[GGS_ADAM_MAX_TENSORS];
    AdamState* s[GGS_ADAM_MAX_TENSORS]; const unsigned long long* guard;
    float beta1, beta2, omb1, omb2, eps;
    double beta1d, beta2d;
};

// TICK: the launch also advances the step count / bias corrections of every tensor (what k_adam_tick_multi does in a launch
// of its own -- ~5 us of a graph-replayed iteration for one wave of work).  Every workgroup derives the corrections of step + 1
// from the stored state (read-only while any workgroup of the tensor may still read it); the LAST workgroup of a tensor to
// finish -- a ticket counter in the state -- stores the advanced state and resets the ticket.
template <bool TICK>
__device__ __forceinline__ void adam_multi_body(const AdamMulti& mt) {
    if (mt.guard && *mt.guard) return;
    int t = 0;
    while (t + 1 < mt.n && blockIdx.x >= mt.first_block[t + 1]) ++t;
    AdamArgs a;
    a.n = mt.numel[t]; a.p = mt.p[t]; a.g = mt.g[t]; a.m = mt.m[t]; a.v = mt.v[t];
    a.beta1 = mt.beta1; a.beta2 = mt.beta2; a.omb1 = mt.omb1; a.omb2 = mt.omb2; a.eps = mt.eps;
    AdamState* st = mt.s[t];
    float bias1 = st->bias1, bs = st->bias2_sqrt;
    long long tt = 0;
    double p1 = 0.0, p2 = 0.0;
    if (TICK) {                                                          // adam_tick_one, without the stores
        tt = st->step + 1;
        p1 = tt == 1 ? mt.beta1d : st->pow1 * mt.beta1d;
        p2 = tt == 1 ? mt.beta2d : st->pow2 * mt.beta2d;
        bias1 = (float)(1.0 - p1);
        bs = (float)sqrt(1.0 - p2);
    }
    const float step_size = *mt.lr[t] / bias1;
    const unsigned b0 = mt.first_block[t], nb = mt.first_block[t + 1] - b0;
    const size_t n4 = a.n / 4, stride = (size_t)nb * 256;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (size_t i = (size_t)(blockIdx.x - b0) * 256 + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam1(p.x, g.x, m.x, v.x, a, step_size, bs);
        adam1(p.y, g.y, m.y, v.y, a, step_size, bs);
        adam1(p.z, g.z, m.z, v.z, a, step_size, bs);
        adam1(p.w, g.w, m.w, v.w, a, step_size, bs);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == b0 && threadIdx.x < (a.n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        adam1(a.p[i], a.g[i], a.m[i], a.v[i], a, step_size, bs);
    }
    if (TICK) __syncthreads();               // every wave of this workgroup has read the state (its values fed the loop above)
    if (TICK && threadIdx.x == 0) {
        // No fence: a workgroup's reads of the state completed before its stores were formed; the new state is read by later
        // launches only.  (A __threadfence() here is an L2 write-back per workgroup on this part: 9 -> 64 us for the launch.)
        if (atomicAdd(&st->ticket, 1u) == nb - 1) {                      // every other workgroup of the tensor has read it
            st->step = tt; st->pow1 = p1; st->pow2 = p2; st->bias1 = bias1; st->bias2_sqrt = bs;
            atomicExch(&st->ticket, 0u);
        }
    }
}
__global__ __launch_bounds__(256) void k_adam_multi(AdamMulti mt) { adam_multi_body<false>(mt); }
__global__ __launch_bounds__(256) void k_adam_tick_multi_step(AdamMulti mt) { adam_multi_body<true>(mt); }

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

int ggs_adam_tick_multi(int n_states, void* const* states, double beta1, double beta2, const void* guard, void* stream) {
    ggs_clear_error_();
    if (n_states <= 0) return GGS_OK;
    if (n_states > 16) return ggs_fail_(GGS_ERR_SIZE, "ggs_adam_tick_multi: at most 16 states per call");
    if (!states) return ggs_fail_(GGS_ERR_ARG, "ggs_adam_tick_multi: NULL states");
    TickMulti m;
    m.n = n_states;
    for (int i = 0; i < n_states; ++i) {
        if (!states[i]) return ggs_fail_(GGS_ERR_ARG, "ggs_adam_tick_multi: NULL state");
        m.s[i] = static_cast<AdamState*>(states[i]);
    }
    hipLaunchKernelGGL(k_adam_tick_multi, dim3(1), dim3(64), 0, (hipStream_t)stream, m, beta1, beta2,
                       static_cast<const unsigned long long*>(guard));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "adam_tick_multi launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

static int adam_step_multi(bool tick, const char* who, int n_tensors, const size_t* numel, float* const* params,
                           const float* const* grads, float* const* exp_avgs, float* const* exp_avg_sqs,
                           const float* const* lrs, void* const* states, double beta1, double beta2, double eps,
                           const void* guard, void* stream) {
    ggs_clear_error_();
    if (n_tensors <= 0) return GGS_OK;
    if (n_tensors > GGS_ADAM_MAX_TENSORS) return ggs_fail_(GGS_ERR_SIZE, "%s: at most %d tensors per call", who, GGS_ADAM_MAX_TENSORS);
    if (!numel || !params || !grads || !exp_avgs || !exp_avg_sqs || !lrs || !states)
        return ggs_fail_(GGS_ERR_ARG, "%s: NULL pointer argument", who);
    AdamMulti mt;
    mt.n = 0; mt.first_block[0] = 0;
    for (int t = 0; t < n_tensors; ++t) {
        if (numel[t] == 0) continue;
        if (!params[t] || !grads[t] || !exp_avgs[t] || !exp_avg_sqs[t] || !lrs[t] || !states[t])
            return ggs_fail_(GGS_ERR_ARG, "%s: NULL tensor pointer", who);
        if (!aligned16(params[t]) || !aligned16(grads[t]) || !aligned16(exp_avgs[t]) || !aligned16(exp_avg_sqs[t]))
            return ggs_fail_(GGS_ERR_ARG, "%s: tensors must be 16-byte aligned", who);
        const int k = mt.n++;
        mt.numel[k] = numel[t]; mt.p[k] = params[t]; mt.g[k] = grads[t]; mt.m[k] = exp_avgs[t]; mt.v[k] = exp_avg_sqs[t];
        mt.lr[k] = lrs[t]; mt.s[k] = static_cast<AdamState*>(states[t]);
        size_t blocks = (numel[t] / 4 + 255) / 256;
        blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
        mt.first_block[k + 1] = mt.first_block[k] + (unsigned)blocks;
    }
    if (mt.n == 0) return GGS_OK;
    mt.guard = static_cast<const unsigned long long*>(guard);
    mt.beta1 = (float)beta1; mt.beta2 = (float)beta2; mt.omb1 = (float)(1.0 - beta1); mt.omb2 = (float)(1.0 - beta2);
    mt.eps = (float)eps; mt.beta1d = beta1; mt.beta2d = beta2;
    if (tick) hipLaunchKernelGGL(k_adam_tick_multi_step, dim3(mt.first_block[mt.n]), dim3(256), 0, (hipStream_t)stream, mt);
    else hipLaunchKernelGGL(k_adam_multi, dim3(mt.first_block[mt.n]), dim3(256), 0, (hipStream_t)stream, mt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
    return GGS_OK;
}

int ggs_adam_step_multi(int n_tensors, const size_t* numel, float* const* params, const float* const* grads,
                        float* const* exp_avgs, float* const* exp_avg_sqs, const float* const* lrs,
                        const void* const* states, double beta1, double beta2, double eps, const void* guard,
                        void* stream) {
    return adam_step_multi(false, "ggs_adam_step_multi", n_tensors, numel, params, grads, exp_avgs, exp_avg_sqs, lrs,
                           const_cast<void* const*>(states), beta1, beta2, eps, guard, stream);
}
// ggs_adam_tick_multi + ggs_adam_step_multi in ONE launch (tensors given once each: a state is advanced once per call).
int ggs_adam_tick_step_multi(int n_tensors, const size_t* numel, float* const* params, const float* const* grads,
                             float* const* exp_avgs, float* const* exp_avg_sqs, const float* const* lrs,
                             void* const* states, double beta1, double beta2, double eps, const void* guard,
                             void* stream) {
    return adam_step_multi(true, "ggs_adam_tick_step_multi", n_tensors, numel, params, grads, exp_avgs, exp_avg_sqs, lrs,
                           states, beta1, beta2, eps, guard, stream);
}

}  // extern "C"
