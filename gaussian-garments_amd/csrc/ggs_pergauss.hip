// ggs_pergauss.hip -- one-lane-per-Gaussian kernels: forward preprocess (+ tile
// histogram), key scatter, and the backward of the per-Gaussian stage.
//
// Roofline: HBM.  Algorithmic bytes per (view, Gaussian): forward reads 44+12K
// (xyz 12, scale 12, rot 16, opacity 4, SH 12K), writes 48 (SplatRec) + 4 (radii);
// backward reads 44+12K + 48 (SplatRec) + 48 (GradRec), writes 56+12K.
// Parameters are shared by the V views of a launch and stay L2/MALL resident.
//
// The whole library is compiled with -ffp-contract=off; this file relies on it so
// that radii / tile rectangles / sort keys are bit-identical to oracle/splat_oracle.c.
#include "ggs_kernels.h"

namespace {

template <int DEG>
__device__ __forceinline__ void sh_to_rgb(const float* __restrict__ sh, const float* dir, float* rgb,
                                          unsigned& clamped) {
    float bas[(DEG + 1) * (DEG + 1)];
    ggs_sh_basis<DEG>(dir, bas);
    clamped = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < (DEG + 1) * (DEG + 1); ++k) r = r + bas[k] * sh[k * 3 + ch];
        r = r + 0.5f;
        if (r < 0.f) { clamped |= 1u << ch; r = 0.f; }
        rgb[ch] = r;
    }
}

// ---- block-aggregated tile histogram ---------------------------------------------------------
// The 256 splats of a workgroup are neighbours on the garment mesh, so together they touch a
// small window of tiles.  Instead of one global atomic per (splat, tile) instance (hundreds of
// increments on the same few counters), the instances are first counted in a dense LDS array
// over the workgroup's tile window (ds_add), and ONE global atomic per touched tile and
// workgroup carries the sum.  Falls back to direct global atomics if the window is too large.
#define BIN_WINDOW_CAP 2048

struct TileWindow { int x0, y0, w, h; bool dense; };

__device__ __forceinline__ TileWindow block_tile_window(int* s_box, bool has, int x0, int y0, int x1, int y1) {
    if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -0x7fffffff; s_box[3] = -0x7fffffff; }
    __syncthreads();
    // Wave reduction first, one LDS atomic per wave and bound.  (Plain per-lane LDS atomicMin / atomicMax are turned by
    // the compiler into a serial v_readlane loop over the 64 lanes: 4 x 64 x 7 scalar instructions per wave -- it was
    // most of this kernel's SALU stream.)
    int bx0 = has ? x0 : 0x7fffffff, by0 = has ? y0 : 0x7fffffff, bx1 = has ? x1 : -0x7fffffff, by1 = has ? y1 : -0x7fffffff;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, d)); by0 = min(by0, __shfl_xor(by0, d));
        bx1 = max(bx1, __shfl_xor(bx1, d)); by1 = max(by1, __shfl_xor(by1, d));
    }
    if ((threadIdx.x & 63) == 0 && bx0 != 0x7fffffff) {
        atomicMin(&s_box[0], bx0); atomicMin(&s_box[1], by0);
        atomicMax(&s_box[2], bx1); atomicMax(&s_box[3], by1);
    }
    __syncthreads();
    TileWindow w;
    w.x0 = s_box[0]; w.y0 = s_box[1];
    w.w = s_box[2] - s_box[0]; w.h = s_box[3] - s_box[1];
    w.dense = w.w > 0 && w.h > 0 && (long long)w.w * w.h <= BIN_WINDOW_CAP;
    if (w.w <= 0 || w.h <= 0) { w.w = 0; w.h = 0; }
    return w;
}

// K1 body.  DEG0: the colour comes from precomputed colours or from SH degree 0 only -- the s2 setting and the headline workload.
// Without the degree 1-3 colour paths the kernel needs 36 instead of 92 VGPRs: eight waves per SIMD instead of five, and this
// pass is latency-sensitive (profiles/r05_preprocess_div_audit.md: -0.3 us per view at eight waves; forcing the general kernel
// there spills its degree-3 path).
template <bool DEG0>
__device__ __forceinline__ void preprocess_body(const PreArgs& a) {
    __shared__ int s_box[4];
    __shared__ uint32_t s_cnt[BIN_WINDOW_CAP];
    const int g0 = blockIdx.x * 256 + threadIdx.x;
    const bool live = g0 < a.P;
    const int g = live ? g0 : a.P - 1;           // idle lanes shadow the last splat, write nothing
    const int v = blockIdx.y;
    const float* __restrict__ view = a.view + 16 * v;
    const float* __restrict__ proj = a.proj + 16 * v;
    const float tanfovx = a.tanfov[2 * v], tanfovy = a.tanfov[2 * v + 1];
    const size_t vg = (size_t)v * a.P + g;
    SplatRec* rec = a.rec + vg;
    const int gx = a.gx, gy = a.gy;
    const int gx16 = (a.W + 15) / 16;           // the reference's 16-pixel tile columns (== gx unless tiles are wider)

    int radius = 0;
    SplatRec out;
    out.px = out.py = out.cx = out.cy = out.cz = out.opacity = out.r = out.g = out.b = out.depth = 0.f;
    out.bbx = 0x00000001u; out.bby = 0x00000001u;   // empty AABB (min 1 > max 0)
    unsigned clamped = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;

    const float m[3] = {a.means3D[3 * (size_t)g], a.means3D[3 * (size_t)g + 1], a.means3D[3 * (size_t)g + 2]};
    const float pvz = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
    if (pvz > GGS_NEAR_Z) {
        float hx = proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12];
        float hy = proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13];
        float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
        float pw = 1.0f / (hw + GGS_W_EPS);
        float ndx = hx * pw, ndy = hy * pw;
        float c6[6];
        if (a.cov3d) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = a.cov3d[6 * (size_t)g + k];
        } else {
            const float sc[3] = {a.scales[3 * (size_t)g], a.scales[3 * (size_t)g + 1], a.scales[3 * (size_t)g + 2]};
            const float4 q4 = reinterpret_cast<const float4*>(a.rots)[g];
            const float q[4] = {q4.x, q4.y, q4.z, q4.w};
            ggs_cov3d(sc, a.scale_modifier, q, c6);
        }
        Ewa e;
        ggs_ewa(view, m, tanfovx, tanfovy, a.W, a.H, e);
        float s0[3], s1[3];
        ggs_sym6_mul(c6, e.M, s0);
        ggs_sym6_mul(c6, e.M + 3, s1);
        float ca = (e.M[0] * s0[0] + e.M[1] * s0[1] + e.M[2] * s0[2]) + GGS_LOWPASS;
        float cb = e.M[0] * s1[0] + e.M[1] * s1[1] + e.M[2] * s1[2];
        float cc = (e.M[3] * s1[0] + e.M[4] * s1[1] + e.M[5] * s1[2]) + GGS_LOWPASS;
        float det = ca * cc - cb * cb;
        if (det != 0.0f) {
            float det_inv = 1.f / det;
            float mid = 0.5f * (ca + cc);
            float root = sqrtf(ggs_max(GGS_LAMBDA_FLOOR, mid * mid - det));
            float l1 = mid + root, l2 = mid - root;
            float rad = ceilf(3.f * sqrtf(ggs_max(l1, l2)));
            float px = ((ndx + 1.0f) * (float)a.W - 1.0f) * 0.5f;
            float py = ((ndy + 1.0f) * (float)a.H - 1.0f) * 0.5f;
            ggs_tile_rect(px, py, rad, gx16, gy, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) != 0) {
                radius = (int)rad;
                out.px = px; out.py = py;
                out.cx = GGS_KA * (cc * det_inv); out.cy = GGS_KB * (-cb * det_inv); out.cz = GGS_KA * (ca * det_inv);
                out.opacity = a.opacities[g];
                out.depth = pvz;
                ggs_alpha_bbox(px, py, ca, cc, out.opacity, out.bbx, out.bby);
                if (a.colors) {
                    out.r = a.colors[3 * (size_t)g]; out.g = a.colors[3 * (size_t)g + 1]; out.b = a.colors[3 * (size_t)g + 2];
                } else {
                    const float* campos = a.campos + 3 * v;
                    float d[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
                    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    d[0] = d[0] / len; d[1] = d[1] / len; d[2] = d[2] / len;
                    const float* sh = a.shs + (size_t)g * a.K * 3;
                    float rgb[3];
                    if (DEG0) sh_to_rgb<0>(sh, d, rgb, clamped);
                    else switch (a.deg) {
                        case 0: sh_to_rgb<0>(sh, d, rgb, clamped); break;
                        case 1: sh_to_rgb<1>(sh, d, rgb, clamped); break;
                        case 2: sh_to_rgb<2>(sh, d, rgb, clamped); break;
                        default: sh_to_rgb<3>(sh, d, rgb, clamped); break;
                    }
                    out.r = rgb[0]; out.g = rgb[1]; out.b = rgb[2];
                }
            }
        }
    }
    if (live) {
        a.radii[vg] = radius;
        SplatAux ax; ax.radius = radius; ax.clamped = clamped; ax.tile_bits = 0;
        a.aux[vg] = ax;
        float4* dst = reinterpret_cast<float4*>(rec);
        const float4* src = reinterpret_cast<const float4*>(&out);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
    bool has = live && radius > 0;
    if (has) {
        ggs_cull_rect(out.bbx, out.bby, x0, y0, x1, y1);
        has = x0 < x1 && y0 < y1;
    }
    const int c0 = x0, c1 = x1;                 // reference columns the splat may be blended in
    ggs_tile_columns(c0, c1, x0, x1);           // from here on x0, x1 are TILE columns
    (void)gx;
    const Footprint fp = ggs_footprint(out.px, out.py, out.cx, out.cy, out.cz, out.opacity);
    uint32_t* cnt = a.tile_count + (size_t)v * a.T;
    const TileWindow w = block_tile_window(s_box, has, x0, y0, x1, y1);
    // The exact ellipse-vs-sub-block tests run ONCE per (splat, tile): list membership (any sub-block reachable) and the quadrant
    // mask are one result, kept GGS_NQ bits per tile of the culled rect (row-major, <= GGS_TILE_BITS_MAX tiles) in SplatAux so the
    // scatter kernel's passes only read bits (round 6: it used to repeat four box tests per member, 160 VALU instructions).
    unsigned long long bits = 0;
    unsigned idx = 0;
    if (w.dense) {
        const int area = w.w * w.h;
        for (int i = threadIdx.x; i < area; i += 256) s_cnt[i] = 0;
        __syncthreads();
        if (has)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x, ++idx) {
                    const unsigned qm = ggs_quad_bits(fp, out.bbx, out.bby, x, y, c0, c1);
                    if (qm) {
                        atomicAdd(&s_cnt[(y - w.y0) * w.w + (x - w.x0)], 1u);
                        if (idx < GGS_TILE_BITS_MAX) bits |= (unsigned long long)qm << (GGS_NQ * idx);
                    }
                }
        __syncthreads();
        for (int i = threadIdx.x; i < area; i += 256) {
            const uint32_t c = s_cnt[i];
            if (c) atomicAdd(&cnt[(w.y0 + i / w.w) * gx + w.x0 + i % w.w], c);
        }
    } else if (has) {
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x, ++idx) {
                const unsigned qm = ggs_quad_bits(fp, out.bbx, out.bby, x, y, c0, c1);
                if (qm) {
                    atomicAdd(&cnt[y * gx + x], 1u);
                    if (idx < GGS_TILE_BITS_MAX) bits |= (unsigned long long)qm << (GGS_NQ * idx);
                }
            }
    }
    if (has) a.aux[vg].tile_bits = bits;
}

}  // namespace

// K1: grid (ceil(P/256), V).  Writes SplatRec + radii, and counts tiles per splat into tile_count[v][t] (cleared by the caller).
__global__ __launch_bounds__(256) void ggs_k_preprocess(PreArgs a) { preprocess_body<false>(a); }
__global__ __launch_bounds__(256) void ggs_k_preprocess_deg0(PreArgs a) { preprocess_body<true>(a); }

// K3: grid (ceil(P/256), V).  For every (splat, tile) instance take a slot in the tile's
// segment and write the 64-bit key  depth bits << 32 | id << GGS_NQ | sub-block mask  -- the mask sits BELOW the id so
// that plain 64-bit comparisons order by (depth, id) without masking anything out (an id occurs once per tile).
// Slot order is arbitrary; the per-tile sort makes the final order deterministic.
__global__ __launch_bounds__(256) void ggs_k_scatter(ScatterArgs a) {
    if (a.header->overflow) return;
    __shared__ int s_box[4];
    __shared__ uint32_t s_cnt[BIN_WINDOW_CAP];
    __shared__ uint32_t s_base[BIN_WINDOW_CAP];
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    unsigned long long key = 0;
    unsigned bbx = 1u, bby = 1u;
    Footprint fp = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 0.f, 0.f};
    bool has = false;
    unsigned long long bits = 0;
    if (g < a.P) {
        const SplatAux ax = a.aux[(size_t)v * a.P + g];
        const int radius = ax.radius;
        bits = ax.tile_bits;
        if (radius > 0) {
            const SplatRec* rec = a.rec + (size_t)v * a.P + g;
            const float4 r0 = reinterpret_cast<const float4*>(rec)[0];
            const float4 r2 = reinterpret_cast<const float4*>(rec)[2];
            ggs_tile_rect(r0.x, r0.y, (float)radius, a.gx16, a.gy, x0, y0, x1, y1);
            bbx = __float_as_uint(r2.z); bby = __float_as_uint(r2.w);
            ggs_cull_rect(bbx, bby, x0, y0, x1, y1);
            // the footprint (a logarithm, two divisions) only where the stored bits do not cover the rectangle: large splats
            if ((x1 - x0) * (y1 - y0) > GGS_TILE_BITS_MAX) {       // (16-pixel columns here: at least as many as tile columns)
                const float4 r1 = reinterpret_cast<const float4*>(rec)[1];
                fp = ggs_footprint(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y);
            }
            key = ((unsigned long long)__float_as_uint(r2.y) << 32) | ((unsigned)g << GGS_NQ);   // low bits: sub-block mask
            has = x0 < x1 && y0 < y1;
        }
    }
    const int c0 = x0, c1 = x1;                 // reference columns; x0, x1 become tile columns
    ggs_tile_columns(c0, c1, x0, x1);
    uint32_t* cur = a.tile_cursor + (size_t)v * a.T;
    const uint32_t* off = a.tile_offset + (size_t)v * a.T;
    unsigned long long* keys = a.keys + a.view_base[v];
    const TileWindow w = block_tile_window(s_box, has, x0, y0, x1, y1);
    // sub-block mask of tile (x, y) (0 = not a member) = the bits the preprocess pass stored (rects of <= GGS_TILE_BITS_MAX tiles), else
    // re-tested with the same function on the same record fields
    const bool small = (x1 - x0) * (y1 - y0) <= GGS_TILE_BITS_MAX;
    auto member = [&](unsigned idx, int x, int y) -> unsigned {
        return small ? (unsigned)(bits >> (GGS_NQ * idx)) & ((1u << GGS_NQ) - 1u) : ggs_quad_bits(fp, bbx, bby, x, y, c0, c1);
    };
    unsigned idx = 0;
    if (w.dense) {
        // count in LDS -> one returning global atomic per touched tile claims the workgroup's run of
        // slots -> second LDS pass hands out the slots inside the run
        const int area = w.w * w.h;
        for (int i = threadIdx.x; i < area; i += 256) s_cnt[i] = 0;
        __syncthreads();
        if (has)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x, ++idx)
                    if (member(idx, x, y)) atomicAdd(&s_cnt[(y - w.y0) * w.w + (x - w.x0)], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < area; i += 256) {
            const uint32_t c = s_cnt[i];
            if (c) {
                const int t = (w.y0 + i / w.w) * a.gx + w.x0 + i % w.w;
                s_base[i] = off[t] + atomicAdd(&cur[t], c);
                s_cnt[i] = 0;
            }
        }
        __syncthreads();
        idx = 0;
        if (has)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x, ++idx) {
                    const unsigned qm = member(idx, x, y);
                    if (!qm) continue;
                    const int i = (y - w.y0) * w.w + (x - w.x0);
                    keys[(size_t)s_base[i] + atomicAdd(&s_cnt[i], 1u)] = key | qm;
                }
    } else if (has) {
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x, ++idx) {
                const unsigned qm = member(idx, x, y);
                if (!qm) continue;
                const int t = y * a.gx + x;
                const uint32_t slot = atomicAdd(&cur[t], 1u);
                keys[(size_t)off[t] + slot] = key | qm;
            }
    }
}

namespace {

template <int DEG>
__device__ __forceinline__ void sh_backward(const float* __restrict__ sh, float* dshr,
                                            const float* d, float len, const float* g_rgb, float* dmean) {
    constexpr int NK = (DEG + 1) * (DEG + 1);
    float bas[NK];
    ggs_sh_basis<DEG>(d, bas);
    // sum_c sh[k][c] * g[c] per coefficient -> direction gradient through d(basis)/d(dir)
    float sg[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float gch = g_rgb[ch];
            const float val = bas[k] * gch;
            dshr[k * 3 + ch] += val;        // summed over this lane's views in registers
            acc += sh[k * 3 + ch] * gch;
        }
        sg[k] = acc;
    }
    float dd[3] = {0.f, 0.f, 0.f};
    if constexpr (DEG > 0) {
        const float x = d[0], y = d[1], z = d[2];
        dd[1] += -GGS_SH_C1 * sg[1]; dd[2] += GGS_SH_C1 * sg[2]; dd[0] += -GGS_SH_C1 * sg[3];
        if constexpr (DEG > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float c0 = 1.0925484305920792f, c2 = 0.31539156525252005f, c4 = 0.5462742152960396f;
            dd[0] += c0 * y * sg[4];               dd[1] += c0 * x * sg[4];
            dd[1] += -c0 * z * sg[5];              dd[2] += -c0 * y * sg[5];
            dd[0] += c2 * -2.f * x * sg[6];        dd[1] += c2 * -2.f * y * sg[6];   dd[2] += c2 * 4.f * z * sg[6];
            dd[0] += -c0 * z * sg[7];              dd[2] += -c0 * x * sg[7];
            dd[0] += c4 * 2.f * x * sg[8];         dd[1] += c4 * -2.f * y * sg[8];
            if constexpr (DEG > 2) {
                const float k0 = -0.5900435899266435f, k1 = 2.890611442640554f, k2 = -0.4570457994644658f;
                const float k3 = 0.3731763325901154f, k5 = 1.445305721320277f;
                dd[0] += k0 * 6.f * x * y * sg[9];            dd[1] += k0 * (3.f * xx - 3.f * yy) * sg[9];
                dd[0] += k1 * y * z * sg[10];                 dd[1] += k1 * x * z * sg[10];    dd[2] += k1 * x * y * sg[10];
                dd[0] += k2 * -2.f * x * y * sg[11];          dd[1] += k2 * (4.f * zz - xx - 3.f * yy) * sg[11];
                dd[2] += k2 * 8.f * y * z * sg[11];
                dd[0] += k3 * -6.f * x * z * sg[12];          dd[1] += k3 * -6.f * y * z * sg[12];
                dd[2] += k3 * (6.f * zz - 3.f * xx - 3.f * yy) * sg[12];
                dd[0] += k2 * (4.f * zz - 3.f * xx - yy) * sg[13]; dd[1] += k2 * -2.f * x * y * sg[13];
                dd[2] += k2 * 8.f * x * z * sg[13];
                dd[0] += k5 * 2.f * x * z * sg[14];           dd[1] += k5 * -2.f * y * z * sg[14];
                dd[2] += k5 * (xx - yy) * sg[14];
                dd[0] += k0 * (3.f * xx - 3.f * yy) * sg[15]; dd[1] += k0 * -6.f * x * y * sg[15];
            }
        }
    }
    const float dot = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) dmean[j] += (dd[j] - d[j] * dot) / len;
}

}  // namespace

// K6: grid (ceil(P/256), S).  One lane per Gaussian loops over the views v = y, y + S, ... and keeps
// every sum over views in registers (incl. the (DEG+1)^2 x 3 SH gradient).  S = 1: plain stores,
// deterministic view sum.  S > 1 (enough waves to fill the chip when P is small against 256 CUs):
// each split adds its partial sums atomically into outputs the host zeroed.
namespace {
// Address of output component c of Gaussian g (nullptr: not produced in this input mode).
__device__ __forceinline__ float* ggs_grad_slot(const PreBwdArgs& a, int c, int g) {
    if (c < 3) return a.dL_dmeans3D + 3 * (size_t)g + c;
    if (c == 3) return a.dL_dopac + g;
    if (c < 11) {
        if (a.cov3d) return (c < 10 && a.dL_dcov3D) ? a.dL_dcov3D + 6 * (size_t)g + (c - 4) : nullptr;
        if (c < 7) return a.dL_dscales ? a.dL_dscales + 3 * (size_t)g + (c - 4) : nullptr;
        return a.dL_drots ? a.dL_drots + 4 * (size_t)g + (c - 7) : nullptr;
    }
    if (c < 14) return (a.colors && a.dL_dcolors) ? a.dL_dcolors + 3 * (size_t)g + (c - 11) : nullptr;
    return (!a.colors && a.dL_dsh) ? a.dL_dsh + (size_t)g * a.K * 3 + (c - 14) : nullptr;
}

template <int DEG>
__device__ __forceinline__ void preprocess_bwd_body(const PreBwdArgs& a) {
    constexpr int NK3 = 3 * (DEG + 1) * (DEG + 1);
    // SH coefficients of the workgroup's 256 Gaussians, staged ONCE: the view loop below re-read them from global memory for every
    // view -- a lane's 12 NK bytes as 16-byte loads whose 64 lanes touch 64 different cache lines each (19 MB per view through
    // the L2 at K = 16; K = 16: 7.0 -> 5.3 us per view; the global-read form: tools/dbg/variants/r06_prebwd_sh_global_reads.patch).
    // Row stride NK3 | 1 floats: lane l reads bank (stride l + j) mod 32, conflict-free.
    constexpr int SH_LD = NK3 | 1;
    __shared__ float s_sh[DEG > 0 ? 256 * SH_LD : 1];
    const bool stage_sh = DEG > 0 && !a.colors && a.dL_dsh;
    if (stage_sh) {
        const size_t g0 = (size_t)blockIdx.x * 256;
        const int n_here = min(256, a.P - (int)g0);
        for (int i = threadIdx.x; i < n_here * NK3; i += 256) {
            const int sidx = i / NK3, j = i - sidx * NK3;
            s_sh[sidx * SH_LD + j] = a.shs[(g0 + sidx) * a.K * 3 + j];
        }
        __syncthreads();
    }
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= a.P) return;
    const float m[3] = {a.means3D[3 * (size_t)g], a.means3D[3 * (size_t)g + 1], a.means3D[3 * (size_t)g + 2]};
    float c6[6], sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f};
    if (a.cov3d) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = a.cov3d[6 * (size_t)g + k];
    } else {
        sc[0] = a.scales[3 * (size_t)g]; sc[1] = a.scales[3 * (size_t)g + 1]; sc[2] = a.scales[3 * (size_t)g + 2];
        const float4 q4 = reinterpret_cast<const float4*>(a.rots)[g];
        q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
        ggs_cov3d(sc, a.scale_modifier, q, c6);
    }
    float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dop = 0.f, dcol[3] = {0.f, 0.f, 0.f};
    constexpr int NK = (DEG + 1) * (DEG + 1);
    float dshr[NK * 3];
#pragma unroll
    for (int k = 0; k < NK * 3; ++k) dshr[k] = 0.f;
    float* dsh = a.dL_dsh ? a.dL_dsh + (size_t)g * a.K * 3 : nullptr;
    const bool split = gridDim.y > 1;

    for (int v = blockIdx.y; v < a.V; v += gridDim.y) {
        const size_t vg = (size_t)v * a.P + g;
        const SplatAux ax = a.aux[vg];
        const int radius = ax.radius;
        float* o2 = a.dL_dmeans2D ? a.dL_dmeans2D + 3 * vg : nullptr;
        // The gradient record and the opacity are requested TOGETHER with the aux word, not behind the branch on it: a lane's
        // view loop is a chain of dependent global-load latencies (3 waves per SIMD at SH degree 3), and this halves it.  A culled
        // splat's record is read for nothing (48 B; the record is zero-filled, never unmapped).
        const float4* gp = reinterpret_cast<const float4*>(a.acc + vg);
        float4 g0 = gp[0], g1 = gp[1], g2 = gp[2];   // mx my cx cy | cz op r g | b depth - -
        float opac_early = reinterpret_cast<const float4*>(a.rec + vg)[1].y;
        asm volatile("" : "+v"(g0.x), "+v"(g1.x), "+v"(g2.x), "+v"(opac_early));      // keep the loads in front of the branch
        if (radius <= 0) {
            if (o2) { o2[0] = 0.f; o2[1] = 0.f; o2[2] = 0.f; }
            continue;
        }
        const float* __restrict__ view = a.view + 16 * v;
        const float* __restrict__ proj = a.proj + 16 * v;
        const float tanfovx = a.tanfov[2 * v], tanfovy = a.tanfov[2 * v + 1];

        // conic -> cov2D -> (Sigma, t)
        Ewa e;
        ggs_ewa(view, m, tanfovx, tanfovy, a.W, a.H, e);
        float s0[3], s1[3];
        ggs_sym6_mul(c6, e.M, s0);
        ggs_sym6_mul(c6, e.M + 3, s1);
        const float ca = (e.M[0] * s0[0] + e.M[1] * s0[1] + e.M[2] * s0[2]) + GGS_LOWPASS;
        const float cb = e.M[0] * s1[0] + e.M[1] * s1[1] + e.M[2] * s1[2];
        const float cc = (e.M[3] * s1[0] + e.M[4] * s1[1] + e.M[5] * s1[2]) + GGS_LOWPASS;
        const float det = ca * cc - cb * cb;
        const float d2i = 1.f / (det * det + GGS_DET_EPS);
        // raw pixel moments -> gradients w.r.t. the conic and the pixel mean (constants applied once here)
        const float opac = opac_early;                     // SplatRec.opacity
        const float det_inv = 1.f / det;
        const float kx = cc * det_inv, ky = -cb * det_inv, kz = ca * det_inv;     // the conic
        const float q0 = -0.5f * opac * g0.z, q1 = -opac * g0.w, q2 = -0.5f * opac * g1.x;
        const float g_mx = -opac * (kx * g0.x + ky * g0.y);
        const float g_my = -opac * (kz * g0.y + ky * g0.x);
        const float da = d2i * (-cc * cc * q0 + cb * cc * q1 - cb * cb * q2);
        const float dc = d2i * (-cb * cb * q0 + ca * cb * q1 - ca * ca * q2);
        const float db = d2i * (2.f * cb * cc * q0 - (det + 2.f * cb * cb) * q1 + 2.f * ca * cb * q2);
        const float* M0 = e.M;
        const float* M1 = e.M + 3;
        dcov[0] += M0[0] * M0[0] * da + M0[0] * M1[0] * db + M1[0] * M1[0] * dc;
        dcov[3] += M0[1] * M0[1] * da + M0[1] * M1[1] * db + M1[1] * M1[1] * dc;
        dcov[5] += M0[2] * M0[2] * da + M0[2] * M1[2] * db + M1[2] * M1[2] * dc;
        dcov[1] += 2.f * M0[0] * M0[1] * da + (M0[0] * M1[1] + M0[1] * M1[0]) * db + 2.f * M1[0] * M1[1] * dc;
        dcov[2] += 2.f * M0[0] * M0[2] * da + (M0[0] * M1[2] + M0[2] * M1[0]) * db + 2.f * M1[0] * M1[2] * dc;
        dcov[4] += 2.f * M0[1] * M0[2] * da + (M0[1] * M1[2] + M0[2] * M1[1]) * db + 2.f * M1[1] * M1[2] * dc;
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float dM0 = 2.f * da * s0[j] + db * s1[j];
            const float dM1 = db * s0[j] + 2.f * dc * s1[j];
            dJ00 += dM0 * view[4 * j + 0]; dJ02 += dM0 * view[4 * j + 2];
            dJ11 += dM1 * view[4 * j + 1]; dJ12 += dM1 * view[4 * j + 2];
        }
        const float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = e.xmul * -e.fx * tz2 * dJ02;
        const float dty = e.ymul * -e.fy * tz2 * dJ12;
        float dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + 2.f * e.fx * e.tx * tz3 * dJ02 + 2.f * e.fy * e.ty * tz3 * dJ12;
        dtz += g2.y;  // depth = t.z
        dmean[0] += view[0] * dtx + view[1] * dty + view[2] * dtz;
        dmean[1] += view[4] * dtx + view[5] * dty + view[6] * dtz;
        dmean[2] += view[8] * dtx + view[9] * dty + view[10] * dtz;

        // pixel mean -> NDC -> mean3D
        const float gnx = g_mx * 0.5f * (float)a.W, gny = g_my * 0.5f * (float)a.H;
        if (o2) { o2[0] = gnx; o2[1] = gny; o2[2] = 0.f; }
        const float hx = proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12];
        const float hy = proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13];
        const float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
        const float pw = 1.0f / (hw + GGS_W_EPS);
        const float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            dmean[j] += (proj[4 * j] * pw - proj[4 * j + 3] * mul1) * gnx + (proj[4 * j + 1] * pw - proj[4 * j + 3] * mul2) * gny;

        dop += g1.y;
        const float grgb[3] = {g1.z, g1.w, g2.x};
        if (a.colors) {
            dcol[0] += grgb[0]; dcol[1] += grgb[1]; dcol[2] += grgb[2];
        } else if (dsh) {
            const unsigned clamped = ax.clamped;
            const float gsh[3] = {(clamped & 1u) ? 0.f : grgb[0], (clamped & 2u) ? 0.f : grgb[1], (clamped & 4u) ? 0.f : grgb[2]};
            const float* campos = a.campos + 3 * v;
            float d[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] = d[0] / len; d[1] = d[1] / len; d[2] = d[2] / len;
            const float* sh = DEG > 0 ? s_sh + threadIdx.x * SH_LD : a.shs + (size_t)g * a.K * 3;   // degree 0: sh is never read
            sh_backward<DEG>(sh, dshr, d, len, gsh, dmean);
        }
    }

    // Output component c of Gaussian g: 0-2 means3D | 3 opacity | 4-6 scales or 4-9 cov3D | 7-10 rotations |
    // 11-13 colours | 14 + 3k + ch SH.  One view split: straight to the gradient tensors.  Several splits:
    // to this split's slab of the partial buffer part[split][c][g] (coalesced), summed by
    // ggs_k_reduce_partials -- no atomics, deterministic.
    const bool acc = a.accumulate != 0;
    const int NC = 14 + 3 * a.K;
    auto put = [&](int c, float v) {
        if (split) { a.part[((size_t)blockIdx.y * NC + c) * a.P + g] = v; return; }
        float* p = ggs_grad_slot(a, c, g);
        if (acc) *p += v; else *p = v;
    };
    if (dsh && !a.colors) {
#pragma unroll
        for (int k = 0; k < NK * 3; ++k) put(14 + k, dshr[k]);
        if (!acc && !split)                     // coefficients above the active degree get zero
            for (int k = NK * 3; k < a.K * 3; ++k) dsh[k] = 0.f;
    }
    put(0, dmean[0]); put(1, dmean[1]); put(2, dmean[2]);
    put(3, dop);
    if (a.colors && a.dL_dcolors) { put(11, dcol[0]); put(12, dcol[1]); put(13, dcol[2]); }
    if (a.cov3d) {
        if (a.dL_dcov3D)
#pragma unroll
            for (int k = 0; k < 6; ++k) put(4 + k, dcov[k]);
    } else if (a.dL_dscales && a.dL_drots) {
        // Sigma = M^T M, M[k][j] = s_k R[j][k]  ->  dM = 2 M Gs, ds_k = sum_j dM[k][j] R[j][k], dR[j][k] = s_k dM[k][j]
        float R[9], Mm[9], sv[3];
        ggs_quat_R(q, R);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sv[k] = a.scale_modifier * sc[k];
#pragma unroll
            for (int j = 0; j < 3; ++j) Mm[k * 3 + j] = sv[k] * R[j * 3 + k];
        }
        const float Gs[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3],
                             0.5f * dcov[4], 0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
        float dR[9], ds[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float accs = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float dM = 2.f * (Mm[k * 3] * Gs[j] + Mm[k * 3 + 1] * Gs[3 + j] + Mm[k * 3 + 2] * Gs[6 + j]);
                accs += dM * R[j * 3 + k];
                dR[j * 3 + k] = sv[k] * dM;
            }
            // Upstream's computeCov3D backward differentiates M = S R with S = diag(modifier * scale) and returns
            // dL/dS_kk as dL/dscale_k: the chain-rule factor `modifier` is NOT applied there.  The bar is "results
            // identical to the reference's", so neither is it here (exact derivative = modifier * accs; s2 / s3 always
            // pass modifier = 1, gaussian_renderer/__init__.py:46).
            ds[k] = accs;
        }
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        const float dq0 = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        const float dq1 = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
        const float dq2 = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
        const float dq3 = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        put(4, ds[0]); put(5, ds[1]); put(6, ds[2]);
        put(7, dq0); put(8, dq1); put(9, dq2); put(10, dq3);
    }
}
}  // namespace

// one kernel per SH degree: the register footprint (and occupancy) follows the degree actually used
__global__ __launch_bounds__(256) void ggs_k_preprocess_bwd_sh0(PreBwdArgs a) { preprocess_bwd_body<0>(a); }
__global__ __launch_bounds__(256) void ggs_k_preprocess_bwd_sh1(PreBwdArgs a) { preprocess_bwd_body<1>(a); }
__global__ __launch_bounds__(256) void ggs_k_preprocess_bwd_sh2(PreBwdArgs a) { preprocess_bwd_body<2>(a); }
__global__ __launch_bounds__(256) void ggs_k_preprocess_bwd_sh3(PreBwdArgs a) { preprocess_bwd_body<3>(a); }

// K6b: grid ceil(P / 64), block 256.  Sums the per-split partials part[s][c][g] into the gradient tensors.
// The partials are component-major (coalesced writes of the per-Gaussian kernel, coalesced reads here: 64 consecutive
// Gaussians of one component = 256 B), the gradient tensors are Gaussian-major ([P,3], [P,4], [P,K,3]): a block sums 64
// Gaussians x all components into an LDS tile and writes it back with the component index fastest, so the SH block of a
// Gaussian (3K consecutive floats) leaves as contiguous runs.  (One thread per (component, Gaussian) with the Gaussian
// fastest wrote 4 bytes every 12K bytes: 0.6 TB/s at K = 16, profiles/r02a_c5_kernel_stats.md.)
#define GGS_RP_G 64
__global__ __launch_bounds__(256) void ggs_k_reduce_partials(PreBwdArgs a, int splits) {
    extern __shared__ float s_tile[];                       // [GGS_RP_G][NC + 1]
    const int NC = 14 + 3 * a.K, LD = NC + 1;
    const int g0 = blockIdx.x * GGS_RP_G;
    const int gl = threadIdx.x & (GGS_RP_G - 1), cl = threadIdx.x / GGS_RP_G;      // 4 component lanes of 64 Gaussians
    const int nk3 = a.colors ? 0 : 3 * (a.deg + 1) * (a.deg + 1);
    const bool in = g0 + gl < a.P;
    for (int c = cl; c < NC; c += 256 / GGS_RP_G) {
        float sum = 0.f;
        if (in && c < 14 + nk3)                             // SH coefficients above the active degree were never written
            for (int s = 0; s < splits; ++s) sum += a.part[((size_t)s * NC + c) * a.P + g0 + gl];
        s_tile[gl * LD + c] = sum;
    }
    __syncthreads();
    const int n = min(GGS_RP_G, a.P - g0) * NC;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int g = i / NC, c = i - g * NC;
        float* dst = ggs_grad_slot(a, c, g0 + g);
        if (!dst) continue;
        const float sum = s_tile[g * LD + c];
        if (a.accumulate) *dst += sum; else *dst = sum;
    }
}
