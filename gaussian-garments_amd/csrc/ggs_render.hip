// ggs_render.hip -- per-tile front-to-back compositing (forward) and the per-pixel
// reverse walk that accumulates per-splat gradients (backward).
//
// One workgroup = one 16x16 tile = 4 wave64; wave w owns the 8x8 quadrant
// (w&1, w>>1) so that a splat's footprint test is as wave-coherent as possible
// (the exec-mask skip of a whole quadrant is the common case).  The tile's
// depth-sorted splat list is staged through LDS in rounds of GGS_BATCH records
// (48 B each, read back with broadcast ds_read_b128).
//
// Roofline: HBM is the nominal bound (algorithmic bytes: N*48 B record gathers +
// 28 B/pixel of outputs forward; N*48 + 20 B/pixel + N*40 B of gradient atomics
// backward), but with LDS staging the kernels are VALU/exp bound: ~25 VALU ops per
// (pixel, splat) forward, ~50 + 60 (wave reduction) backward.
#include "ggs_kernels.h"

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// Sum over the 64 lanes of the wave; the total is valid in lane 63 only.
// row_shr 1/2/4/8 build row totals in lane 15 of each 16-lane row, row_bcast:15 / :31
// fold the four rows (GCN/CDNA DPP; no LDS traffic, 6 VALU ops).
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_fetch<0x111, 0xf>(v);
    v += dpp_fetch<0x112, 0xf>(v);
    v += dpp_fetch<0x114, 0xf>(v);
    v += dpp_fetch<0x118, 0xf>(v);
    v += dpp_fetch<0x142, 0xa>(v);
    v += dpp_fetch<0x143, 0xc>(v);
    return v;
}

}  // namespace

// K4b: grid (T, V), block 256.
__global__ __launch_bounds__(256) void ggs_k_render_fwd(RenderArgs a) {
    if (a.header->overflow) return;
    const int t = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tx = t % a.gx, ty = t / a.gx;
    const int px = tx * GGS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GGS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const int L = (int)a.tile_count[(size_t)v * a.T + t];
    const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    const uint32_t* ids = a.ids + base;
    const float4* rec = reinterpret_cast<const float4*>(a.rec + (size_t)v * a.P);

    __shared__ float4 s_rec[GGS_BATCH * 3];

    const float pxf = (float)px, pyf = (float)py;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t contributor = 0, last = 0;
    bool done = !inside;

    for (int start = 0; start < L; start += GGS_BATCH) {
        if (__syncthreads_and(done)) break;
        const int n = min(GGS_BATCH, L - start);
        if (tid < n) {
            const float4* r = rec + (size_t)ids[start + tid] * 3;
            s_rec[tid * 3 + 0] = r[0];
            s_rec[tid * 3 + 1] = r[1];
            s_rec[tid * 3 + 2] = r[2];
        }
        __syncthreads();
        for (int j = 0; j < n && !done; ++j) {
            contributor++;
            const float4 g0 = s_rec[j * 3 + 0];        // px py cx cy
            const float4 g1 = s_rec[j * 3 + 1];        // cz opacity r g
            const float dx = g0.x - pxf, dy = g0.y - pyf;
            const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
            if (power > 0.f) continue;
            const float alpha = ggs_min(GGS_ALPHA_MAX, g1.y * __expf(power));
            if (alpha < GGS_ALPHA_MIN) continue;
            const float test_T = T * (1.f - alpha);
            if (test_T < GGS_T_MIN) { done = true; continue; }
            const float4 g2 = s_rec[j * 3 + 2];        // b depth radius clamped
            const float w = alpha * T;
            C0 = fmaf(g1.z, w, C0);
            C1 = fmaf(g1.w, w, C1);
            C2 = fmaf(g2.x, w, C2);
            D = fmaf(g2.y, w, D);
            A += w;
            T = test_T;
            last = contributor;
        }
    }
    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)py * a.W + px;
        const float* bg = a.bg + 3 * v;
        float* oc = a.out_color + (size_t)v * 3 * HW;
        a.final_T[(size_t)v * HW + pix] = T;
        a.n_contrib[(size_t)v * HW + pix] = last;
        oc[pix] = fmaf(T, bg[0], C0);
        oc[HW + pix] = fmaf(T, bg[1], C1);
        oc[2 * HW + pix] = fmaf(T, bg[2], C2);
        a.out_depth[(size_t)v * HW + pix] = D;
        a.out_alpha[(size_t)v * HW + pix] = A;
    }
}

namespace {

#define GGS_NGRAD 10

// K5 body.  DA: gradients of the depth / alpha outputs are present.
template <bool DA>
__device__ __forceinline__ void render_bwd_body(const RenderBwdArgs& a) {
    const int t = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const int L = (int)a.tile_count[(size_t)v * a.T + t];
    if (L == 0) return;
    const int lane = tid & 63, wave = tid >> 6;
    const int tx = t % a.gx, ty = t / a.gx;
    const int px = tx * GGS_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GGS_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = (size_t)py * a.W + px;
    const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    const uint32_t* ids = a.ids + base;
    const float4* rec = reinterpret_cast<const float4*>(a.rec + (size_t)v * a.P);
    GradRec* acc = a.acc + (size_t)v * a.P;

    __shared__ float4 s_rec[GGS_BATCH * 3];
    __shared__ uint32_t s_id[GGS_BATCH];
    __shared__ float s_acc[GGS_BATCH * GGS_NGRAD];
    __shared__ int s_max;

    const int nc = inside ? (int)a.n_contrib[(size_t)v * HW + pix] : 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    if (nc > 0) atomicMax(&s_max, nc);
    __syncthreads();
    const int maxc = s_max;
    if (maxc == 0) return;

    const float Tf = inside ? a.final_T[(size_t)v * HW + pix] : 1.f;
    float T = Tf;
    float dC0 = 0.f, dC1 = 0.f, dC2 = 0.f, dD = 0.f, dA = 0.f;
    if (inside) {
        const float* dc = a.dL_dcolor + (size_t)v * 3 * HW;
        dC0 = dc[pix]; dC1 = dc[HW + pix]; dC2 = dc[2 * HW + pix];
        if (DA) {
            if (a.dL_ddepth) dD = a.dL_ddepth[(size_t)v * HW + pix];
            if (a.dL_dalpha) dA = a.dL_dalpha[(size_t)v * HW + pix];
        }
    }
    const float* bg = a.bg + 3 * v;
    const float bgdot = bg[0] * dC0 + bg[1] * dC1 + bg[2] * dC2;
    float rec0 = 0.f, rec1 = 0.f, rec2 = 0.f, recD = 0.f, recA = 0.f;
    float last_a = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_d = 0.f;
    const float pxf = (float)px, pyf = (float)py;

    for (int hi = maxc; hi > 0; hi -= GGS_BATCH) {
        const int lo = max(0, hi - GGS_BATCH);
        const int n = hi - lo;
        if (tid < n) {
            const uint32_t id = ids[lo + tid];
            s_id[tid] = id;
            const float4* r = rec + (size_t)id * 3;
            s_rec[tid * 3 + 0] = r[0];
            s_rec[tid * 3 + 1] = r[1];
            s_rec[tid * 3 + 2] = r[2];
        }
        for (int i = tid; i < n * GGS_NGRAD; i += GGS_BLOCK) s_acc[i] = 0.f;
        __syncthreads();

        for (int jj = n - 1; jj >= 0; --jj) {
            const float4 g0 = s_rec[jj * 3 + 0];       // px py cx cy
            const float4 g1 = s_rec[jj * 3 + 1];       // cz opacity r g
            const float dx = g0.x - pxf, dy = g0.y - pyf;
            const float power = -0.5f * (g0.z * dx * dx + g1.x * dy * dy) - g0.w * dx * dy;
            const float G = __expf(power);
            const float alpha = ggs_min(GGS_ALPHA_MAX, g1.y * G);
            const bool valid = (lo + jj < nc) && (power <= 0.f) && (alpha >= GGS_ALPHA_MIN);
            if (!__any(valid)) continue;               // wave-uniform: nobody in this quadrant blended it
            float v_r = 0.f, v_g = 0.f, v_b = 0.f, v_op = 0.f, v_mx = 0.f, v_my = 0.f;
            float v_cx = 0.f, v_cy = 0.f, v_cz = 0.f, v_dep = 0.f;
            if (valid) {
                const float4 g2 = s_rec[jj * 3 + 2];   // b depth
                T = T / (1.f - alpha);
                const float w = alpha * T;
                rec0 = last_a * lc0 + (1.f - last_a) * rec0; lc0 = g1.z;
                rec1 = last_a * lc1 + (1.f - last_a) * rec1; lc1 = g1.w;
                rec2 = last_a * lc2 + (1.f - last_a) * rec2; lc2 = g2.x;
                float dL_da = (g1.z - rec0) * dC0 + (g1.w - rec1) * dC1 + (g2.x - rec2) * dC2;
                v_r = w * dC0; v_g = w * dC1; v_b = w * dC2;
                if (DA) {
                    recD = last_a * last_d + (1.f - last_a) * recD; last_d = g2.y;
                    recA = last_a + (1.f - last_a) * recA;
                    dL_da += (g2.y - recD) * dD + (1.f - recA) * dA;
                    v_dep = w * dD;
                }
                dL_da *= T;
                last_a = alpha;
                dL_da += (-Tf / (1.f - alpha)) * bgdot;
                const float dL_dG = g1.y * dL_da;      // straight through the 0.99 clamp
                const float gdx = G * dx, gdy = G * dy;
                v_mx = dL_dG * (-gdx * g0.z - gdy * g0.w);
                v_my = dL_dG * (-gdy * g1.x - gdx * g0.w);
                v_cx = -0.5f * gdx * dx * dL_dG;
                v_cy = -gdx * dy * dL_dG;
                v_cz = -0.5f * gdy * dy * dL_dG;
                v_op = G * dL_da;
            }
            // wave-level sums (uniform control flow), one LDS atomic per wave per value
            v_mx = wave_sum_lane63(v_mx); v_my = wave_sum_lane63(v_my);
            v_cx = wave_sum_lane63(v_cx); v_cy = wave_sum_lane63(v_cy); v_cz = wave_sum_lane63(v_cz);
            v_op = wave_sum_lane63(v_op);
            v_r = wave_sum_lane63(v_r); v_g = wave_sum_lane63(v_g); v_b = wave_sum_lane63(v_b);
            if (DA) v_dep = wave_sum_lane63(v_dep);
            if (lane == 63) {
                float* s = s_acc + jj * GGS_NGRAD;
                atomicAdd(s + 0, v_mx); atomicAdd(s + 1, v_my);
                atomicAdd(s + 2, v_cx); atomicAdd(s + 3, v_cy); atomicAdd(s + 4, v_cz);
                atomicAdd(s + 5, v_op);
                atomicAdd(s + 6, v_r); atomicAdd(s + 7, v_g); atomicAdd(s + 8, v_b);
                if (DA) atomicAdd(s + 9, v_dep);
            }
        }
        __syncthreads();
        if (tid < n) {
            float* dst = reinterpret_cast<float*>(acc + s_id[tid]);
            const float* s = s_acc + tid * GGS_NGRAD;
#pragma unroll
            for (int c = 0; c < (DA ? 10 : 9); ++c) {
                const float val = s[c];
                if (val != 0.f) atomicAdd(dst + c, val);
            }
        }
        __syncthreads();
    }
}

}  // namespace

// K5: grid (T, V), block 256.  Two entry points so the common case (no loss on depth /
// alpha: s2_registration.py:258-267, s3_appearance.py:131-140) carries no dead work.
__global__ __launch_bounds__(256) void ggs_k_render_bwd(RenderBwdArgs a) { render_bwd_body<false>(a); }
__global__ __launch_bounds__(256) void ggs_k_render_bwd_da(RenderBwdArgs a) { render_bwd_body<true>(a); }
