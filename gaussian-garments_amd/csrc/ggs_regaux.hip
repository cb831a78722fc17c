// ggs_regaux.hip -- everything of one s2 registration iteration that is neither rasterizer, photometric loss nor
// optimiser, in two small kernels ("fused Adam + densification stats"):
//   * the two hinge regularisers of the first-frame template and their gradients
//       loss_xyz   = lambda_xyz   * mean_vis relu(|_xyz_i| - threshold_xyz)                (s2_registration.py:262-263)
//       loss_scale = lambda_scale * mean_vis |relu(exp(_scaling_i) - threshold_scale)|_2   (s2_registration.py:264-265)
//     (mean over the Gaussians with radii > 0), gradients ADDED to the local-parameter gradient buffers;
//   * the chain rule through the opacity activation: dL/d_opacity = dL/dopacity * o (1 - o)  (scene/gaussian_model.py:107-108)
//   * densification statistics: max_radii2D[vis] = max(., radii), xyz_gradient_accum[vis] += |dL/dmeans2D.xy|,
//     denom[vis] += 1                                            (s2_registration.py:301-303, scene/gaussian_model.py:410-412)
// In PyTorch this is ~45 elementwise / reduction / indexing kernels per iteration (and, with boolean indexing, two
// host syncs); a captured iteration spends more time in those launches than in the rasterizer's forward.
// HBM-bound, ~100 B per Gaussian.  `guard` as in ggs_adam.hip: statistics are left untouched when it is set.
#include "ggs_kernels.h"

namespace {

struct RegArgs {
    int P;
    const float* xyz; const float* log_scaling; const int* radii; const float* dL_dmeans2D;
    const float* opacity; const float* dL_dopacity; float* dL_dopacity_logit;
    float thr_xyz, lam_xyz, thr_scale, lam_scale;
    float* dL_dxyz; float* dL_dlog_scaling;
    float* max_radii2D; float* xyz_gradient_accum; float* denom;
    float* out_losses;           // {loss_xyz, loss_scale, n_visible}
    float* sums;                 // scratch: {n_visible, sum hinge_xyz, sum hinge_scale}
    const unsigned long long* guard;
    const float* tail_sums; const unsigned long long* tail_header; float* tail_out;     // GgsStepTail (tail_out NULL: none)
    int n_adam; AdamState* adam[16]; double beta1, beta2;                                // GgsStepTail: optimiser states to advance
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Grid-stride over at most REG_REDUCE_BLOCKS workgroups: the three sums end in one atomic each per workgroup on the SAME cache
// line, and same-address atomics retire one every ~10 ns -- with one workgroup per 256 Gaussians (391 x 3 at 100k) that tail was
// most of this kernel's 7 us.
#define REG_REDUCE_BLOCKS 128
__global__ __launch_bounds__(256) void k_reg_reduce(RegArgs a) {
    __shared__ float red[3][4];
    float n = 0.f, hx = 0.f, hs = 0.f;
    const int stride = gridDim.x * 256;
    // branch-free body on clamped indices, four Gaussians per thread in flight: the loads of a thread do not wait for each other
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < a.P; i0 += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = i0 + u * stride, i = j < a.P ? j : a.P - 1;
            const bool vis = j < a.P && a.radii[i] > 0;
            const float x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
            const float s0 = fmaxf(expf(a.log_scaling[3 * i]) - a.thr_scale, 0.f);
            const float s1 = fmaxf(expf(a.log_scaling[3 * i + 1]) - a.thr_scale, 0.f);
            const float s2 = fmaxf(expf(a.log_scaling[3 * i + 2]) - a.thr_scale, 0.f);
            n += vis ? 1.f : 0.f;                                            // selects, not products: an invisible Gaussian may hold inf
            hx += vis ? fmaxf(sqrtf(x * x + y * y + z * z) - a.thr_xyz, 0.f) : 0.f;
            hs += vis ? sqrtf(s0 * s0 + s1 * s1 + s2 * s2) : 0.f;
        }
    }
    n = wave_sum(n); hx = wave_sum(hx); hs = wave_sum(hs);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = n; red[1][w] = hx; red[2][w] = hs; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (t != 0.f) atomicAdd(&a.sums[threadIdx.x], t);
    }
}

__global__ __launch_bounds__(256) void k_reg_apply(RegArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float n_vis = a.sums ? a.sums[0] : 0.f;
    // optimiser states (GgsStepTail): thread t < n_adam of the first workgroup advances state t, like k_adam_tick_multi
    if (blockIdx.x == 0 && (int)threadIdx.x < a.n_adam && !(a.guard && *a.guard))
        ggs_adam_tick_one(a.adam[threadIdx.x], a.beta1, a.beta2);
    if (i == 0) {
        // 0/0 = NaN like torch's mean over an empty selection
        const float lx = a.sums ? a.sums[1] / n_vis * a.lam_xyz : 0.f, ls = a.sums ? a.sums[2] / n_vis * a.lam_scale : 0.f;
        if (a.out_losses && a.sums) { a.out_losses[0] = lx; a.out_losses[1] = ls; a.out_losses[2] = n_vis; }
        if (a.tail_out) {                                          // the step's 48-byte result block (device or host-mapped)
            float* o = a.tail_out;
            o[0] = a.tail_sums ? a.tail_sums[0] : 0.f; o[1] = a.tail_sums ? a.tail_sums[1] : 0.f;
            o[2] = lx; o[3] = ls; o[4] = n_vis; o[5] = o[6] = o[7] = 0.f;
            unsigned long long* h = reinterpret_cast<unsigned long long*>(o + 8);
            h[0] = a.tail_header ? a.tail_header[0] : 0ull; h[1] = a.tail_header ? a.tail_header[1] : 0ull;
        }
    }
    if (i >= a.P) return;
    if (a.dL_dopacity_logit) {                                     // sigmoid backward
        const float o = a.opacity[i];
        a.dL_dopacity_logit[i] = a.dL_dopacity[i] * ((1.f - o) * o);
    }
    const bool vis = a.radii[i] > 0;
    if (a.dL_dxyz && vis) {
        const float x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
        const float nrm = sqrtf(x * x + y * y + z * z);
        if (nrm > a.thr_xyz) {                                     // relu'(0) = 0; nrm > thr >= 0 here
            const float c = a.lam_xyz / n_vis / nrm;
            a.dL_dxyz[3 * i] += c * x; a.dL_dxyz[3 * i + 1] += c * y; a.dL_dxyz[3 * i + 2] += c * z;
        }
        const float e0 = expf(a.log_scaling[3 * i]), e1 = expf(a.log_scaling[3 * i + 1]), e2 = expf(a.log_scaling[3 * i + 2]);
        const float s0 = fmaxf(e0 - a.thr_scale, 0.f), s1 = fmaxf(e1 - a.thr_scale, 0.f), s2 = fmaxf(e2 - a.thr_scale, 0.f);
        const float ns = sqrtf(s0 * s0 + s1 * s1 + s2 * s2);
        if (ns > 0.f) {                                            // torch: the 2-norm has zero subgradient at 0
            const float c = a.lam_scale / n_vis / ns;
            a.dL_dlog_scaling[3 * i] += c * s0 * e0;
            a.dL_dlog_scaling[3 * i + 1] += c * s1 * e1;
            a.dL_dlog_scaling[3 * i + 2] += c * s2 * e2;
        }
    }
    if (a.max_radii2D && vis && !(a.guard && *a.guard)) {
        a.max_radii2D[i] = fmaxf(a.max_radii2D[i], (float)a.radii[i]);
        const float gx = a.dL_dmeans2D[3 * i], gy = a.dL_dmeans2D[3 * i + 1];
        a.xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy);
        a.denom[i] += 1.f;
    }
}

}  // namespace

extern "C" int ggs_registration_aux_tail(int P, const float* xyz, const float* log_scaling, const int* radii,
                                         const float* dL_dmeans2D, const float* opacity, const float* dL_dopacity,
                                         float* dL_dopacity_logit, float threshold_xyz, float lambda_xyz,
                                         float threshold_scale, float lambda_scale, float* dL_dxyz, float* dL_dlog_scaling,
                                         float* max_radii2D, float* xyz_gradient_accum, float* denom, float* out_losses,
                                         void* scratch, const void* guard, const GgsStepTail* tail, void* stream) {
    ggs_clear_error_();
    // the step's last consumer of pre-clear marks: whatever is left when this call returns -- on any path -- ends here
    struct DropMarks { ~DropMarks() { ggs_drop_clear_marks_(); } } drop_marks_on_return;
    if (P <= 0) return GGS_OK;
    if (tail && (reinterpret_cast<uintptr_t>(tail->out_block) & 7))
        return ggs_fail_(GGS_ERR_ARG, "ggs_registration_aux_tail: out_block is not 8-byte aligned");
    if (tail && (tail->n_adam_states < 0 || tail->n_adam_states > 16))
        return ggs_fail_(GGS_ERR_SIZE, "ggs_registration_aux_tail: at most 16 optimiser states");
    for (int t = 0; tail && t < tail->n_adam_states; ++t)
        if (!tail->adam_states[t]) return ggs_fail_(GGS_ERR_ARG, "ggs_registration_aux_tail: NULL optimiser state");
    if (!radii) return ggs_fail_(GGS_ERR_ARG, "ggs_registration_aux: radii is NULL");
    const bool hinge = dL_dxyz || dL_dlog_scaling;
    if (hinge && (!xyz || !log_scaling || !dL_dxyz || !dL_dlog_scaling || !scratch))
        return ggs_fail_(GGS_ERR_ARG, "ggs_registration_aux: the hinge terms need xyz, log_scaling, both gradient buffers and scratch");
    if (max_radii2D && (!xyz_gradient_accum || !denom || !dL_dmeans2D))
        return ggs_fail_(GGS_ERR_ARG, "ggs_registration_aux: the statistics need xyz_gradient_accum, denom and dL_dmeans2D");
    if (dL_dopacity_logit && (!opacity || !dL_dopacity))
        return ggs_fail_(GGS_ERR_ARG, "ggs_registration_aux: the opacity chain rule needs opacity and dL_dopacity");
    hipStream_t s = (hipStream_t)stream;
    RegArgs a;
    a.P = P; a.xyz = xyz; a.log_scaling = log_scaling; a.radii = radii; a.dL_dmeans2D = dL_dmeans2D;
    a.opacity = opacity; a.dL_dopacity = dL_dopacity; a.dL_dopacity_logit = dL_dopacity_logit;
    a.thr_xyz = threshold_xyz; a.lam_xyz = lambda_xyz; a.thr_scale = threshold_scale; a.lam_scale = lambda_scale;
    a.dL_dxyz = dL_dxyz; a.dL_dlog_scaling = dL_dlog_scaling;
    a.max_radii2D = max_radii2D; a.xyz_gradient_accum = xyz_gradient_accum; a.denom = denom;
    a.out_losses = out_losses; a.sums = hinge ? static_cast<float*>(scratch) : nullptr;
    a.guard = static_cast<const unsigned long long*>(guard);
    a.tail_sums = tail ? tail->loss_sums : nullptr;
    a.tail_header = tail ? static_cast<const unsigned long long*>(tail->header) : nullptr;
    a.tail_out = tail ? static_cast<float*>(tail->out_block) : nullptr;
    a.n_adam = tail ? tail->n_adam_states : 0; a.beta1 = tail ? tail->beta1 : 0.0; a.beta2 = tail ? tail->beta2 : 0.0;
    for (int t = 0; t < 16; ++t) a.adam[t] = t < a.n_adam ? static_cast<AdamState*>(tail->adam_states[t]) : nullptr;
    const dim3 grid((unsigned)((P + 255) / 256));
    if (hinge) {
        if (ggs_zero_async(scratch, 16, s) != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "ggs_registration_aux: clearing the sums failed");
        const dim3 grid_r(grid.x < REG_REDUCE_BLOCKS ? grid.x : REG_REDUCE_BLOCKS);
        hipLaunchKernelGGL(k_reg_reduce, grid_r, dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(k_reg_apply, grid, dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "registration_aux launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

extern "C" int ggs_registration_aux(int P, const float* xyz, const float* log_scaling, const int* radii,
                                    const float* dL_dmeans2D, const float* opacity, const float* dL_dopacity,
                                    float* dL_dopacity_logit, float threshold_xyz, float lambda_xyz,
                                    float threshold_scale, float lambda_scale, float* dL_dxyz, float* dL_dlog_scaling,
                                    float* max_radii2D, float* xyz_gradient_accum, float* denom, float* out_losses,
                                    void* scratch, const void* guard, void* stream) {
    return ggs_registration_aux_tail(P, xyz, log_scaling, radii, dL_dmeans2D, opacity, dL_dopacity, dL_dopacity_logit,
                                     threshold_xyz, lambda_xyz, threshold_scale, lambda_scale, dL_dxyz, dL_dlog_scaling,
                                     max_radii2D, xyz_gradient_accum, denom, out_losses, scratch, guard, nullptr, stream);
}
