// ggs_loss.hip -- fused photometric loss of the inner steps: masked L1 + 11x11 Gaussian-window SSIM,
// value AND gradient w.r.t. the rendered image in two tile passes.
//
// Replaces ("next" #1) the PyTorch chain of utils/loss_utils.py:17-68 as composed at
// s2_registration.py:259-260 / s3_appearance.py:132-133:
//     loss_img  = mean(|(img - gt) * mask|) * (1 - lambda)
//     loss_ssim = 1 - mean(ssim_map(img * mask, gt * mask)) * lambda
// which costs 5 grouped conv2d forward + their backward (~10x the image bytes, milliseconds at 1080p --
// several times the HIP rasterizer's forward+backward).  Here:
//   pass A (ggs_k_loss_stats): x = img*mask and y = gt*mask (zero padded like conv2d(padding=5)); separable 11-tap
//           window (same fp32 taps as create_window) gives mu1, mu2, E[xx + yy], E[xy] -- FOUR filtered maps, not five: the
//           SSIM map needs sigma1^2 + sigma2^2 = E[xx] + E[yy] - mu1^2 - mu2^2 only as a sum, and d/dE[xx] = d/dE[yy] -- the
//           SSIM map value and its three partial derivatives (d/dmu1 total, d/dE[xx], d/dE[xy]) are formed per pixel;
//           workgroup-reduced sums of |x - y| and of the map go to sums[v] (one atomic pair per workgroup).
//   pass B (ggs_k_loss_grad): the three derivative maps are filtered with the same (symmetric) window:
//           dSSIM/dx = G*dmu1 + 2 x (G*dExx) + y (G*dExy); combined with the L1 sign term and the mask.
// Both passes stream rows through one wave per 64-column strip (see below): the horizontal filter reads a wave-private
// LDS row, the vertical filter is a ring of partial sums in registers.
// Roofline: HBM for the batched call (A: reads 24 B/px(+mask) writes 36 B/px; B: reads 60 B/px writes 12 B/px ~ 270 MB
// per 1080p view; B moves ~4.8 TB/s at 32 views); pass A is bound by the latency of its row steps (see GGS_LOSS_WAVES).
#include "ggs_kernels.h"

namespace {

#define LH 5                  // window half width
#define SSIM_C1 0.0001f       // 0.01^2
#define SSIM_C2 0.0009f       // 0.03^2

// fp32 taps of gaussian(11, 1.5) / sum, bit patterns as utils/loss_utils.py:26-28 produces them
__device__ __constant__ const float G11[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                               2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                               3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

struct LossArgs {
    int V, H, W;
    const float *img, *gt, *mask;     // [V][3][H][W], [V][3][H][W], [V][1][H][W] or null
    const float* const* gt_tab;       // table form: device array of V device pointers to [3][H][W] images (gt is null then)
    const float* const* mask_tab;     //             device array of V device pointers to [H][W] masks, or null
    const float* w;                   // [V][2] device: weights of mean|x-y| and of mean ssim_map in the loss
    float inv_n;                      // 1 / (3 H W)
    float* sums;                      // [V][2] = {sum |x - y|, sum ssim_map}
    float* dmap;                      // [V][3 channels][3 maps][H][W] scratch
    float* dL_dimg;                   // [V][3][H][W]
    const uint32_t* tile_count;       // region-of-interest form: [V][tgy * tgx] list lengths of the forward's 16x16 tiles, or null
    int tgx, tgy;
    const uint32_t* mask_tiles;       // sparse-mask form: [V][tgy * tgx] non-zero mask pixels per 16x16 tile (ggs_mask_tiles), or null
    const uint32_t* const* mask_tiles_tab;   //          ... or a device array of V device pointers to [tgy * tgx] tables
};

// Region of interest.  dL/dimage reaches a parameter only through pixels of tiles that HAVE a list (the render backward returns
// on an empty tile before it reads anything): with the forward's tile_count at hand, pass B runs only the boxes (64-column
// strip x LS_HB_ROI-row band) that overlap a non-empty tile and leaves dL/dimage alone elsewhere.  What is left of pass B is a
// few hundred waves, each bound by its chain of row steps -- so the region-of-interest pass uses SHORT bands (12 rows: 22 row
// steps per wave instead of 44; the 1.8 x filtered rows that made short bands lose on a whole image no longer matter).
// A box of pass B reads the derivative maps up to LH pixels outside itself: within the strips next to its own, and within
// LH + LS_HB_ROI - 1 rows of the non-empty tile that made it active -- pass A stores the maps of a box iff a non-empty tile
// lies within that distance of it, and skips the derivative arithmetic otherwise; its sums cover every pixel as before.
// any_tile: does a non-empty tile of view v intersect the pixel rectangle [x0, x1] x [y0, y1]?  (wave-uniform result)
// (tc: the view's table -- list lengths of the forward, or the mask's tile occupancy of the sparse-mask form)
__device__ __forceinline__ bool any_tile(const uint32_t* __restrict__ tc, const LossArgs& a, int x0, int y0, int x1, int y1, int lane) {
    const int tx0 = max(x0, 0) >> 4, tx1 = min(min(x1, a.W - 1) >> 4, a.tgx - 1);
    const int ty0 = max(y0, 0) >> 4, ty1 = min(min(y1, a.H - 1) >> 4, a.tgy - 1);
    const int nx = tx1 - tx0 + 1, n = nx * (ty1 - ty0 + 1);
    bool hit = false;
    for (int i = lane; i < n; i += 64) hit |= tc[(size_t)(ty0 + i / nx) * a.tgx + tx0 + i % nx] != 0;
    return __builtin_amdgcn_ballot_w64(hit) != 0;
}
__device__ __forceinline__ const uint32_t* view_tiles(const LossArgs& a, int v) { return a.tile_count + (size_t)v * a.tgx * a.tgy; }

template <int NW>
__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += s_red[w];
    __syncthreads();
    return v;
}

// ---- row streaming ------------------------------------------------------------------------------------------------
// One wave64 = one strip of 64 output columns x LS_HB output rows of one (view, channel); the wave walks its
// LS_HB + 10 input rows top to bottom.  Per input row: the 74 masked x / y values go through a wave-private LDS row
// (double buffered; one wave = no barrier), every lane filters its column horizontally (11 taps, 5 maps), and the
// vertical filter is a REGISTER ring: the row's five horizontal sums are added, with tap k, to the partial sums of the 11
// output rows they belong to; the output row that received its last tap is finished (SSIM value + derivative maps).
// Taps are accumulated in the order 0..10 in both directions.  Against round 1's 32x32 LDS tiles (five intermediate maps
// = 27 KB per tile, 3 workgroups per CU, three barriers per channel) there are no workgroup barriers, no LDS round trip of
// the intermediate maps and ~2.4 KB of LDS per wave: occupancy is set by the registers alone (pass A: 3 waves per SIMD, see
// GGS_LOSS_WAVES below; 32 views 1.84 -> 1.54 ms and 1.61 -> 1.40 ms, one view 76 -> 56 us and 56 -> 42 us in round 2).  The ring index is static because the row loop is unrolled by 11.
// Memory pipeline of a step: park the row loaded during the previous step -> write the previous step's outputs -> start
// the loads of the next row -> filter.  Loads are branch-free (clamped addresses, padding applied at park time) and the
// stores are issued BEFORE the loads that the next step waits for: vmcnt retires in order, so a store issued after
// them would put its whole write latency on the critical path of every step.
// Shape of pass A: GGS_LOSS_WAVES waves per SIMD is what the register allocation aims at, and the loads run TWO input rows ahead
// of the one being filtered (the alternatives -- one row ahead, 2 / 4 waves, the second group of LDS reads fenced behind the
// first -- were A/B builds: tools/dbg/variants/r06_loss_ahead_and_split_reads_switches.patch).  The pass is bound
// by the LATENCY of a row step (load -> LDS -> 44 reads -> filter -> SSIM chain), not by its instruction count: going from
// five maps / 215 VALU per row to four / 167 changed nothing at 4 waves per SIMD, where 15-28 registers spill into the
// chain (scratch round trips); 3 waves without spills and loads two rows ahead measure, us per 1080p view (tools/dbg/time_loss.py,
// one view | 16 views):  round 3 kernel 69 | 59;  4 waves, 1 row ahead 65 | 57;  3 waves, 1 ahead 61 | 57;  4 waves, 2 ahead 76 | 71;
// 2 waves, 2 ahead 69 | 56;  **3 waves, 2 ahead 54 | 50.5** (region-of-interest form 55 | 36).
#define GGS_LOSS_WAVES 3
// Pass A comes in two workgroup shapes.  NCH = 3: a workgroup holds the three colour channels of its four bands (12 waves = one
// CU's worth at 3 waves per SIMD): 240 workgroups per view instead of 720 end in the two same-address atomics of the sums -- those
// retire one every ~10 ns, and when all workgroups of a view finish together that tail is 7.5 of the kernel's 48 us (one view:
// 54 -> 50 us, 48 -> 43 inside the captured step).  NCH = 1 (four waves, one channel) for launches of several views, whose
// workgroups finish spread out and want the finer scheduling grain (16 views: 50.6 against 52.4 us per view).
#define LS_CH3_MAX_VIEWS 2
#define LS_COLS 64
#define LS_HB 34                          // LS_HB + 2 LH = 44 input rows = 4 x 11.  (12-row bands for single-view launches --
                                          // 2.8x the waves, half the dependent row steps each, 1.8x the filtered rows --
                                          // measured 0.166 against 0.104 ms per 1080p view in round 2: the waves no longer
                                          // fit in one round of the chip, two rounds of 22 steps are no shorter than one of 44.)
#define LS_HB_ROI 12                      // band height of pass B in the region-of-interest form (see any_tile)
#define LS_IN (LS_COLS + 2 * LH)          // 74
#define LS_WAVES 4                        // waves per workgroup: consecutive bands of one strip

template <int NM>
struct RowRing { float v[11][NM]; };

// tap k of input row (ring phase R) lands in slot (R - k) mod 11; k = 0 starts the sum of a new output row
template <int NM, int R>
__device__ __forceinline__ void ring_add(RowRing<NM>& ring, const float (&h)[NM]) {
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const int slot = (R - k + 11) % 11;
#pragma unroll
        for (int m = 0; m < NM; ++m)
            ring.v[slot][m] = k == 0 ? G11[0] * h[m] : fmaf(G11[k], h[m], ring.v[slot][m]);
    }
}

// BYTE offsets of the two input columns of a lane in input row y within one image plane, clamped into the image (ok = not
// padding).  32-bit: a plane base is wave-uniform, so every access is "scalar base + 32-bit lane offset" -- as size_t
// indices the per-lane address arithmetic was ~17 64-bit VALU instructions per row step (loss_args bounds 12 H W < 2^32).
struct RowAddr { uint32_t p1, p2; bool ok1, ok2; };
__device__ __forceinline__ RowAddr row_addr(int y, int H, int W, int ox, int lane) {
    const int c1 = ox - LH + lane, c2 = ox + LS_COLS - LH + lane;
    const bool has2 = lane < 2 * LH;
    const bool oky = y >= 0 && y < H;
    RowAddr r;
    r.ok1 = oky && c1 >= 0 && c1 < W;
    r.ok2 = oky && has2 && c2 < W;
    const uint32_t row = (uint32_t)min(max(y, 0), H - 1) * (uint32_t)W;
    r.p1 = (row + (uint32_t)min(max(c1, 0), W - 1)) * 4u;
    r.p2 = (row + (uint32_t)min(max(has2 ? c2 : c1, 0), W - 1)) * 4u;
    return r;
}
// (The explicit global address space matters for the image pointers that are LOADED from the pointer tables: the compiler
// cannot infer it for them and then forms 64-bit per-lane addresses instead of "global_load v, v_off, s[base]".)
typedef const __attribute__((address_space(1))) char* gbytes;
typedef __attribute__((address_space(1))) char* gbytes_w;
__device__ __forceinline__ float ld_off(const float* base, uint32_t byte_off) {
    return *(const __attribute__((address_space(1))) float*)((gbytes)base + byte_off);
}
__device__ __forceinline__ void st_off(float* base, uint32_t byte_off, float v) {
    *(__attribute__((address_space(1))) float*)((gbytes_w)base + byte_off) = v;
}

struct StatsRow { float x1, y1, m1, x2, y2, m2; bool ok1, ok2; };       // raw values of the next input row
template <bool MASK>
__device__ __forceinline__ StatsRow stats_load_row(const float* __restrict__ img, const float* __restrict__ gt,
                                                  const float* __restrict__ mask, int y, int H, int W, int ox, int lane) {
    const RowAddr ad = row_addr(y, H, W, ox, lane);
    StatsRow r;
    r.ok1 = ad.ok1; r.ok2 = ad.ok2;
    r.x1 = ld_off(img, ad.p1); r.y1 = ld_off(gt, ad.p1); r.m1 = MASK ? ld_off(mask, ad.p1) : 1.f;
    r.x2 = ld_off(img, ad.p2); r.y2 = ld_off(gt, ad.p2); r.m2 = MASK ? ld_off(mask, ad.p2) : 1.f;
    return r;
}

struct StatsCtx {
    const float *img, *gt, *mask;
    float* dm;                // null: the maps of this box are not needed (region-of-interest form)
    float *dm1, *dm2;         // dm + HW, dm + 2 HW
    int H, W, oy, ox, lane;
    size_t HW;
    float (*sx)[LS_IN];       // [2][LS_IN] wave-private: x, y, x x + y y, x y of the current input row
    float (*sy)[LS_IN];
    float (*ss)[LS_IN];
    float (*sp)[LS_IN];
    float l1, ssum;
    float s0;                 // sparse-mask form: the SSIM value of an all-zero window (ssum then collects S - s0)
    StatsRow nxt;
    StatsRow nxt2;            // the row after `nxt`, in flight
    float o0, o1, o2;         // outputs of the previous step, written at the start of this one
    bool pend;
};

// SSIM value of a window from its four filtered moments (+ the factors the derivative maps reuse)
struct SsimTerms { float S, A1, A2, iB1, iB2, inv; };
__device__ __forceinline__ SsimTerms ssim_terms(float m1, float m2, float ess, float e12) {
    SsimTerms t;
    const float mm = m1 * m1 + m2 * m2, cv = e12 - m1 * m2;          // ess - mm = sigma1^2 + sigma2^2
    t.A1 = 2.f * m1 * m2 + SSIM_C1; t.A2 = 2.f * cv + SSIM_C2;
    const float B1 = mm + SSIM_C1, B2 = (ess - mm) + SSIM_C2;
    t.iB1 = __builtin_amdgcn_rcpf(B1); t.iB2 = __builtin_amdgcn_rcpf(B2);
    t.inv = t.iB1 * t.iB2;
    t.S = t.A1 * t.A2 * t.inv;
    return t;
}

template <int R, bool MASK, int HB, bool SPARSE>
__device__ __forceinline__ void stats_step(StatsCtx& c, RowRing<4>& ring, int i) {
    const int buf = i & 1;
    const int lane = c.lane;
    const int x = c.ox + lane;
    {   // the products are formed ONCE per input pixel here, not once per tap in the 11 lanes that filter it
        const float xv = c.nxt.ok1 ? c.nxt.x1 * c.nxt.m1 : 0.f, yv = c.nxt.ok1 ? c.nxt.y1 * c.nxt.m1 : 0.f;
        c.sx[buf][lane] = xv; c.sy[buf][lane] = yv;
        c.ss[buf][lane] = fmaf(xv, xv, yv * yv); c.sp[buf][lane] = xv * yv;
    }
    if (lane < 2 * LH) {
        const float xv = c.nxt.ok2 ? c.nxt.x2 * c.nxt.m2 : 0.f, yv = c.nxt.ok2 ? c.nxt.y2 * c.nxt.m2 : 0.f;
        c.sx[buf][LS_COLS + lane] = xv; c.sy[buf][LS_COLS + lane] = yv;
        c.ss[buf][LS_COLS + lane] = fmaf(xv, xv, yv * yv); c.sp[buf][LS_COLS + lane] = xv * yv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c.pend && c.dm) {                                                // output row i - 1 - 2 LH
        const uint32_t p = ((uint32_t)(c.oy + i - 1 - 2 * LH) * (uint32_t)c.W + (uint32_t)x) * 4u;
        st_off(c.dm, p, c.o0); st_off(c.dm1, p, c.o1); st_off(c.dm2, p, c.o2);
    }
    c.nxt = c.nxt2;
    c.nxt2 = stats_load_row<MASK>(c.img, c.gt, c.mask, c.oy - LH + i + 2, c.H, c.W, c.ox, lane);
    // two maps at a time (22 values in flight, not 44: the ring of vertical partial sums already holds 44 registers)
    float h[4] = {0.f, 0.f, 0.f, 0.f};
    {
        float xs[11], ys[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) { xs[k] = c.sx[buf][lane + k]; ys[k] = c.sy[buf][lane + k]; }
#pragma unroll
        for (int k = 0; k < 11; ++k) { h[0] = fmaf(G11[k], xs[k], h[0]); h[1] = fmaf(G11[k], ys[k], h[1]); }
        // L1 over the strip's own pixels (the row is an own row when LH <= i < LH + LS_HB)
        if (i >= LH && i < LH + HB && c.oy + i - LH < c.H && x < c.W) c.l1 += fabsf(xs[LH] - ys[LH]);
    }
    {
        float sq[11], pr[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) { sq[k] = c.ss[buf][lane + k]; pr[k] = c.sp[buf][lane + k]; }
#pragma unroll
        for (int k = 0; k < 11; ++k) { h[2] = fmaf(G11[k], sq[k], h[2]); h[3] = fmaf(G11[k], pr[k], h[3]); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ring_add<4, R>(ring, h);
    const int o = i - 2 * LH;                       // the output row that just received tap 10
    c.pend = o >= 0 && c.oy + o < c.H && x < c.W;
    {
        constexpr int slot = (R + 1) % 11;
        const float m1 = ring.v[slot][0], m2 = ring.v[slot][1];
        const SsimTerms t = ssim_terms(m1, m2, ring.v[slot][2], ring.v[slot][3]);
        const float S = t.S;
        if (c.pend) c.ssum += SPARSE ? S - c.s0 : S;
        if (c.dm) {                                  // wave-uniform: null where nobody will read the maps of this box
            c.o0 = 2.f * m2 * (t.A2 - t.A1) * t.inv - 2.f * m1 * S * t.iB1 + 2.f * m1 * S * t.iB2;
            c.o1 = -S * t.iB2;
            c.o2 = 2.f * t.A1 * t.inv;
        }
    }
}

template <int R, typename Step, typename Ctx, typename Ring>
__device__ __forceinline__ void unroll11(Ctx& c, Ring& ring, int i0) {
    if constexpr (R < 11) {
        Step::template run<R>(c, ring, i0 + R);
        unroll11<R + 1, Step>(c, ring, i0);
    }
}
template <bool MASK, int HB, bool SPARSE>
struct StatsStep {
    template <int R> static __device__ __forceinline__ void run(StatsCtx& c, RowRing<4>& r, int i) { stats_step<R, MASK, HB, SPARSE>(c, r, i); }
};

// Sparse-mask form (SPARSE; single-view launches use bands of LS_HB_SPARSE rows).  A real mask is the silhouette of the garment: ~90 %
// of a 1080p frame is masked out, and there the masked x and y are zero, every window statistic is zero and the SSIM value
// is ONE constant s0 = C1 C2 / (C1 C2) in the kernel's own arithmetic.  With the mask's tile occupancy at hand (ggs_mask_tiles:
// computed once per mask, the masks of a capture are static) a wave whose input window holds no mask pixel, and whose maps no
// box of pass B will read, does nothing at all.  The sum is kept as  sum over the computed pixels of (S - s0)  +  3 H W s0
// (the constant is added once per view, by workgroup (0, 0)), so a skipped pixel needs no contribution of its own -- and the
// chip is no longer filled by one round of 34-row bands: the few boxes that remain run as SHORT bands (22 dependent row
// steps instead of 44), the shape that lost on a dense image for lack of wave slots (see LS_HB).
// Measured (profiles/r05_sparse_mask_loss.md; 1080p, silhouette of 16 %): one view 45.3 -> 31.2 us (23-row bands 33.6, 34-row
// bands 39.7), 16 views 32.7 -> 13.2 us per view; on a dense mask, where nothing is skipped, the short bands cost 1.45x.
template <bool MASK, int NCH, int HB, bool SPARSE>
__device__ __forceinline__ void loss_stats_stream_body(const LossArgs& a, float (*s_x)[4][2][LS_IN], float* s_red) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int v = NCH == 3 ? blockIdx.z : blockIdx.z / 3, ch = NCH == 3 ? wave / LS_WAVES : blockIdx.z % 3;
    const int band = blockIdx.y * LS_WAVES + wave % LS_WAVES;
    const size_t HW = (size_t)a.H * a.W;
    StatsCtx c;
    c.img = a.img + ((size_t)v * 3 + ch) * HW;
    c.gt = a.gt_tab ? a.gt_tab[v] + (size_t)ch * HW : a.gt + ((size_t)v * 3 + ch) * HW;
    c.mask = !MASK ? nullptr : a.mask_tab ? a.mask_tab[v] : a.mask + (size_t)v * HW;
    c.dm = a.dmap + ((size_t)v * 3 + ch) * 3 * HW; c.dm1 = c.dm + HW; c.dm2 = c.dm + 2 * HW;
    c.H = a.H; c.W = a.W; c.oy = band * HB; c.ox = blockIdx.x * LS_COLS; c.lane = lane; c.HW = HW;
    c.sx = s_x[wave][0]; c.sy = s_x[wave][1]; c.ss = s_x[wave][2]; c.sp = s_x[wave][3];
    c.l1 = 0.f; c.ssum = 0.f; c.pend = false; c.o0 = c.o1 = c.o2 = 0.f; c.s0 = 0.f;
    bool run = c.oy < a.H;
    if (a.tile_count && run &&
        !any_tile(view_tiles(a, v), a, c.ox - LS_COLS, c.oy - (LH + LS_HB_ROI - 1), c.ox + 2 * LS_COLS - 1,
                  c.oy + HB - 1 + LH + LS_HB_ROI - 1, lane))
        c.dm = nullptr;
    if (SPARSE) {
        float z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));          // a zero the compiler cannot fold: s0 through the kernel's own instructions
        c.s0 = ssim_terms(z, z, z, z).S;
        const uint32_t* mt = a.mask_tiles_tab ? a.mask_tiles_tab[v] : a.mask_tiles + (size_t)v * a.tgx * a.tgy;
        // nothing to do iff nobody reads the maps of this box AND no mask pixel lies in its input window
        if (run && !c.dm && !any_tile(mt, a, c.ox - LH, c.oy - LH, c.ox + LS_COLS - 1 + LH, c.oy + HB - 1 + LH, lane)) run = false;
    }
    if (run) {
        RowRing<4> ring;
#pragma unroll
        for (int j = 0; j < 11; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) ring.v[j][m] = 0.f;
        c.nxt = stats_load_row<MASK>(c.img, c.gt, c.mask, c.oy - LH, c.H, c.W, c.ox, lane);
        c.nxt2 = stats_load_row<MASK>(c.img, c.gt, c.mask, c.oy - LH + 1, c.H, c.W, c.ox, lane);
        static_assert((HB + 2 * LH) % 11 == 0, "the row loop is unrolled by 11");
        for (int i0 = 0; i0 < HB + 2 * LH; i0 += 11) unroll11<0, StatsStep<MASK, HB, SPARSE>>(c, ring, i0);
        if (c.pend && c.dm) {                                            // the last output row
            const uint32_t p = ((uint32_t)(c.oy + HB - 1) * (uint32_t)c.W + (uint32_t)(c.ox + lane)) * 4u;
            st_off(c.dm, p, c.o0); st_off(c.dm1, p, c.o1); st_off(c.dm2, p, c.o2);
        }
    }
    float l1 = block_sum<NCH * LS_WAVES>(c.l1, s_red), ssum = block_sum<NCH * LS_WAVES>(c.ssum, s_red);
    if (threadIdx.x == 0) {
        if (SPARSE) {       // the constant part, once per view; workgroups that skipped everything end without an atomic
            if (blockIdx.x == 0 && blockIdx.y == 0 && (NCH == 3 || ch == 0)) ssum += 3.f * (float)a.H * (float)a.W * c.s0;
            if (l1 != 0.f) atomicAdd(&a.sums[2 * v], l1);
            if (ssum != 0.f) atomicAdd(&a.sums[2 * v + 1], ssum);
        } else {
            atomicAdd(&a.sums[2 * v], l1);
            atomicAdd(&a.sums[2 * v + 1], ssum);
        }
    }
}

}  // namespace

// Pass A: grid (ceil(W/64), ceil(ceil(H/HB)/4), V * 3 / NCH), block 256 NCH = 4 NCH independent waves.
// (Masked and unmasked forms are separate kernels: as two branches of one kernel the register allocation of the shared
// prologue pushed the masked body over its register budget.)
#define LS_HB_SPARSE 12                   // 12, 23 or 34 (LS_HB_SPARSE + 10 must be a multiple of 11)
#define GGS_LOSS_STATS_KERNEL(NAME, MASK, NCH, HB, SPARSE)                                                                  \
    __global__ __launch_bounds__(256 * NCH) __attribute__((amdgpu_waves_per_eu(GGS_LOSS_WAVES, GGS_LOSS_WAVES))) void NAME(LossArgs a) { \
        __shared__ float s_x[LS_WAVES * NCH][4][2][LS_IN];                                                                  \
        __shared__ float s_red[LS_WAVES * NCH];                                                                             \
        loss_stats_stream_body<MASK, NCH, HB, SPARSE>(a, s_x, s_red);                                                       \
    }
GGS_LOSS_STATS_KERNEL(ggs_k_loss_stats, false, 1, LS_HB, false)
GGS_LOSS_STATS_KERNEL(ggs_k_loss_stats_masked, true, 1, LS_HB, false)
GGS_LOSS_STATS_KERNEL(ggs_k_loss_stats_ch3, false, 3, LS_HB, false)
GGS_LOSS_STATS_KERNEL(ggs_k_loss_stats_masked_ch3, true, 3, LS_HB, false)
GGS_LOSS_STATS_KERNEL(ggs_k_loss_stats_sparse, true, 1, LS_HB, true)
GGS_LOSS_STATS_KERNEL(ggs_k_loss_stats_sparse_ch3, true, 3, LS_HB_SPARSE, true)

// Tile occupancy of a mask: tiles[v][ty * ceil(W/16) + tx] = number of non-zero pixels of mask v in that 16x16 tile.
__global__ __launch_bounds__(256) void ggs_k_mask_tiles(int H, int W, int tgx, int tgy, const float* __restrict__ mask,
                                                        uint32_t* __restrict__ tiles) {
    const int t = blockIdx.x, v = blockIdx.y;
    const int x = (t % tgx) * 16 + (threadIdx.x & 15), y = (t / tgx) * 16 + (threadIdx.x >> 4);
    const bool nz = x < W && y < H && mask[(size_t)v * H * W + (size_t)y * W + x] != 0.f;
    const int n = __syncthreads_count(nz);
    if (threadIdx.x == 0) tiles[(size_t)v * tgx * tgy + t] = (uint32_t)n;
}

namespace {

// next input row of the three derivative maps + the image values at the NEXT output pixel (consumed one step later)
struct GradRow { float a1, b1, c1, a2, b2, c2, xo, yo, mo; bool ok1, ok2; };
template <bool MASK>
__device__ __forceinline__ GradRow grad_load_row(const float* __restrict__ dm, const float* __restrict__ img,
                                                const float* __restrict__ gt, const float* __restrict__ mask, size_t HW,
                                                int y, int yo, int H, int W, int ox, int lane) {
    const RowAddr ad = row_addr(y, H, W, ox, lane);
    GradRow r;
    r.ok1 = ad.ok1; r.ok2 = ad.ok2;
    const float *dm1 = dm + HW, *dm2 = dm + 2 * HW;
    r.a1 = ld_off(dm, ad.p1); r.b1 = ld_off(dm1, ad.p1); r.c1 = ld_off(dm2, ad.p1);
    r.a2 = ld_off(dm, ad.p2); r.b2 = ld_off(dm1, ad.p2); r.c2 = ld_off(dm2, ad.p2);
    const uint32_t po = ((uint32_t)min(max(yo, 0), H - 1) * (uint32_t)W + (uint32_t)min(ox + lane, W - 1)) * 4u;
    r.xo = ld_off(img, po); r.yo = ld_off(gt, po); r.mo = MASK ? ld_off(mask, po) : 1.f;
    return r;
}
struct GradCtx {
    const float *img, *gt, *mask, *dm;
    float* out;
    int H, W, oy, ox, lane;
    size_t HW;
    float w_l1, w_ssim;
    float (*sd)[3][LS_IN];    // [2][3][LS_IN] wave-private
    GradRow nxt;
    float o0;
    bool pend;
};
template <int R, bool MASK>
__device__ __forceinline__ void grad_step(GradCtx& c, RowRing<3>& ring, int i) {
    const int buf = i & 1, lane = c.lane;
    const int x = c.ox + lane;
    c.sd[buf][0][lane] = c.nxt.ok1 ? c.nxt.a1 : 0.f; c.sd[buf][1][lane] = c.nxt.ok1 ? c.nxt.b1 : 0.f;
    c.sd[buf][2][lane] = c.nxt.ok1 ? c.nxt.c1 : 0.f;
    if (lane < 2 * LH) {
        c.sd[buf][0][LS_COLS + lane] = c.nxt.ok2 ? c.nxt.a2 : 0.f; c.sd[buf][1][LS_COLS + lane] = c.nxt.ok2 ? c.nxt.b2 : 0.f;
        c.sd[buf][2][LS_COLS + lane] = c.nxt.ok2 ? c.nxt.c2 : 0.f;
    }
    const float xo = c.nxt.xo, yo = c.nxt.yo, mo = c.nxt.mo;          // image at this step's output pixel
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (c.pend) st_off(c.out, ((uint32_t)(c.oy + i - 1 - 2 * LH) * (uint32_t)c.W + (uint32_t)x) * 4u, c.o0);
    c.nxt = grad_load_row<MASK>(c.dm, c.img, c.gt, c.mask, c.HW, c.oy - LH + i + 1, c.oy + i + 1 - 2 * LH, c.H, c.W, c.ox, lane);
    float d[3][11];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int k = 0; k < 11; ++k) d[m][k] = c.sd[buf][m][lane + k];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float h[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float g = G11[k];
        h[0] = fmaf(g, d[0][k], h[0]); h[1] = fmaf(g, d[1][k], h[1]); h[2] = fmaf(g, d[2][k], h[2]);
    }
    ring_add<3, R>(ring, h);
    const int o = i - 2 * LH;
    c.pend = o >= 0 && c.oy + o < c.H && x < c.W;
    {
        constexpr int slot = (R + 1) % 11;
        const float f0 = ring.v[slot][0], f1 = ring.v[slot][1], f2 = ring.v[slot][2];
        const float xv = xo * mo, yv = yo * mo;
        const float dssim = f0 + 2.f * xv * f1 + yv * f2;
        const float df = xv - yv;
        const float dl1 = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        c.o0 = mo * (c.w_ssim * dssim + c.w_l1 * dl1);
    }
}
template <bool MASK>
struct GradStep {
    template <int R> static __device__ __forceinline__ void run(GradCtx& c, RowRing<3>& r, int i) { grad_step<R, MASK>(c, r, i); }
};

template <bool MASK, int HB>
__device__ __forceinline__ void loss_grad_stream_body(const LossArgs& a, float (*s_d)[2][3][LS_IN]) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int v = blockIdx.z / 3, ch = blockIdx.z % 3;
    const int band = blockIdx.y * LS_WAVES + wave;
    const size_t HW = (size_t)a.H * a.W;
    GradCtx c;
    c.img = a.img + ((size_t)v * 3 + ch) * HW;
    c.gt = a.gt_tab ? a.gt_tab[v] + (size_t)ch * HW : a.gt + ((size_t)v * 3 + ch) * HW;
    c.mask = !MASK ? nullptr : a.mask_tab ? a.mask_tab[v] : a.mask + (size_t)v * HW;
    c.dm = a.dmap + ((size_t)v * 3 + ch) * 3 * HW;
    c.out = a.dL_dimg + ((size_t)v * 3 + ch) * HW;
    c.H = a.H; c.W = a.W; c.oy = band * HB; c.ox = blockIdx.x * LS_COLS; c.lane = lane; c.HW = HW;
    c.w_l1 = a.w[2 * v] * a.inv_n; c.w_ssim = a.w[2 * v + 1] * a.inv_n;
    c.sd = s_d[wave];
    c.pend = false; c.o0 = 0.f;
    if (c.oy >= a.H) return;
    if (a.tile_count && !any_tile(view_tiles(a, v), a, c.ox, c.oy, c.ox + LS_COLS - 1, c.oy + HB - 1, lane)) return;
    RowRing<3> ring;
#pragma unroll
    for (int j = 0; j < 11; ++j)
#pragma unroll
        for (int m = 0; m < 3; ++m) ring.v[j][m] = 0.f;
    c.nxt = grad_load_row<MASK>(c.dm, c.img, c.gt, c.mask, HW, c.oy - LH, c.oy - 2 * LH, c.H, c.W, c.ox, lane);
    static_assert((HB + 2 * LH) % 11 == 0, "the row loop is unrolled by 11");
    for (int i0 = 0; i0 < HB + 2 * LH; i0 += 11) unroll11<0, GradStep<MASK>>(c, ring, i0);
    if (c.pend) c.out[(size_t)(c.oy + HB - 1) * c.W + c.ox + lane] = c.o0;
}

}  // namespace

// Pass B: same decomposition as pass A.
__global__ __launch_bounds__(256) void ggs_k_loss_grad(LossArgs a) {
    __shared__ float s_d[LS_WAVES][2][3][LS_IN];
    if (a.mask || a.mask_tab) loss_grad_stream_body<true, LS_HB>(a, s_d);
    else loss_grad_stream_body<false, LS_HB>(a, s_d);
}
// Pass B of the region-of-interest form: bands of LS_HB_ROI rows, only the boxes that overlap a non-empty tile do anything.
__global__ __launch_bounds__(256) void ggs_k_loss_grad_roi(LossArgs a) {
    __shared__ float s_d[LS_WAVES][2][3][LS_IN];
    if (a.mask || a.mask_tab) loss_grad_stream_body<true, LS_HB_ROI>(a, s_d);
    else loss_grad_stream_body<false, LS_HB_ROI>(a, s_d);
}

extern "C" {

size_t ggs_photometric_scratch_bytes(int n_views, int H, int W) {
    if (n_views <= 0 || H <= 0 || W <= 0) return 0;
    return ggs_align((size_t)n_views * 9 * (size_t)H * W * sizeof(float));
}

static int loss_args(LossArgs& a, int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                     const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count, void* scratch,
                     const char* who) {
    ggs_clear_error_();
    if (n_views <= 0 || H <= 0 || W <= 0) return ggs_fail_(GGS_ERR_ARG, "%s: bad sizes", who);
    if (!img || !(gt || gt_tab) || !scratch) return ggs_fail_(GGS_ERR_ARG, "%s: NULL pointer argument", who);
    if ((size_t)n_views * 3 > 65535) return ggs_fail_(GGS_ERR_SIZE, "%s: n_views too large", who);
    if ((size_t)H * W * 12 >= ((size_t)1 << 32)) return ggs_fail_(GGS_ERR_SIZE, "%s: image too large (12 H W must fit 32 bits)", who);
    a.V = n_views; a.H = H; a.W = W; a.img = img; a.gt = gt; a.mask = mask; a.gt_tab = gt_tab; a.mask_tab = mask_tab;
    a.inv_n = 1.f / (3.f * (float)H * (float)W);
    a.w = nullptr; a.sums = nullptr; a.dL_dimg = nullptr; a.dmap = (float*)scratch;
    a.tile_count = tile_count; a.tgx = (W + 15) / 16; a.tgy = (H + 15) / 16;
    a.mask_tiles = nullptr; a.mask_tiles_tab = nullptr;
    return GGS_OK;
}

static int photometric_forward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                               const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                               const uint32_t* mask_tiles, const uint32_t* const* mask_tiles_tab,
                               float* sums, void* scratch, void* stream_) {
    LossArgs a;
    int rc = loss_args(a, n_views, H, W, img, gt, mask, gt_tab, mask_tab, tile_count, scratch, "ggs_photometric_forward");
    if (rc != GGS_OK) return rc;
    if (!sums) return ggs_fail_(GGS_ERR_ARG, "ggs_photometric_forward: sums is NULL");
    hipStream_t s = (hipStream_t)stream_;
    a.sums = sums;
    if (ggs_zero_async(sums, (size_t)n_views * 2 * sizeof(float), s) != hipSuccess)
        return ggs_fail_(GGS_ERR_HIP, "ggs_photometric_forward: clearing the sums failed");
    const bool ch3 = n_views <= LS_CH3_MAX_VIEWS;
    const bool masked = a.mask || a.mask_tab;
    const bool sparse = mask_tiles || mask_tiles_tab;
    if (sparse && !(masked && tile_count))
        return ggs_fail_(GGS_ERR_ARG, "ggs_photometric_forward_sparse: the mask's tile table needs the mask and the forward's tile_count");
    a.mask_tiles = mask_tiles; a.mask_tiles_tab = mask_tiles_tab;
    const int hb = sparse && ch3 ? LS_HB_SPARSE : LS_HB;
    const int bands = (H + hb - 1) / hb;
    const dim3 grid((unsigned)((W + LS_COLS - 1) / LS_COLS), (unsigned)((bands + LS_WAVES - 1) / LS_WAVES),
                    (unsigned)(ch3 ? n_views : n_views * 3));
    if (grid.y > 65535u) return ggs_fail_(GGS_ERR_SIZE, "ggs_photometric_forward: image too tall");
    if (sparse && ch3) hipLaunchKernelGGL(ggs_k_loss_stats_sparse_ch3, grid, dim3(768), 0, s, a);
    else if (sparse) hipLaunchKernelGGL(ggs_k_loss_stats_sparse, grid, dim3(256), 0, s, a);
    else if (ch3 && masked) hipLaunchKernelGGL(ggs_k_loss_stats_masked_ch3, grid, dim3(768), 0, s, a);
    else if (ch3) hipLaunchKernelGGL(ggs_k_loss_stats_ch3, grid, dim3(768), 0, s, a);
    else if (masked) hipLaunchKernelGGL(ggs_k_loss_stats_masked, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ggs_k_loss_stats, grid, dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "loss_stats launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

static int photometric_backward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                const void* scratch, const float* weights, float* dL_dimg, void* stream_) {
    LossArgs a;
    int rc = loss_args(a, n_views, H, W, img, gt, mask, gt_tab, mask_tab, tile_count, (void*)scratch, "ggs_photometric_backward");
    if (rc != GGS_OK) return rc;
    if (!weights || !dL_dimg) return ggs_fail_(GGS_ERR_ARG, "ggs_photometric_backward: NULL pointer argument");
    a.w = weights; a.dL_dimg = dL_dimg;
    const int hb = tile_count ? LS_HB_ROI : LS_HB;
    const int bands = (H + hb - 1) / hb;
    const dim3 grid((unsigned)((W + LS_COLS - 1) / LS_COLS), (unsigned)((bands + LS_WAVES - 1) / LS_WAVES), (unsigned)(n_views * 3));
    if (tile_count) hipLaunchKernelGGL(ggs_k_loss_grad_roi, grid, dim3(256), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(ggs_k_loss_grad, grid, dim3(256), 0, (hipStream_t)stream_, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "loss_grad launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

int ggs_photometric_forward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                            float* sums, void* scratch, void* stream) {
    return photometric_forward(n_views, H, W, img, gt, mask, nullptr, nullptr, nullptr, nullptr, nullptr, sums, scratch, stream);
}
int ggs_photometric_backward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                             const void* scratch, const float* weights, float* dL_dimg, void* stream) {
    return photometric_backward(n_views, H, W, img, gt, mask, nullptr, nullptr, nullptr, scratch, weights, dL_dimg, stream);
}
// Table form: the ground-truth images (and masks) are named by DEVICE-resident pointer tables that the kernels read at run
// time, so a captured hipGraph can be replayed on another camera's images by rewriting 8 bytes per image instead of copying
// 33 MB into static buffers.
int ggs_photometric_forward_tab(int n_views, int H, int W, const float* img, const float* const* gt_tab,
                                const float* const* mask_tab, float* sums, void* scratch, void* stream) {
    return photometric_forward(n_views, H, W, img, nullptr, nullptr, gt_tab, mask_tab, nullptr, nullptr, nullptr, sums, scratch, stream);
}
int ggs_photometric_backward_tab(int n_views, int H, int W, const float* img, const float* const* gt_tab,
                                 const float* const* mask_tab, const void* scratch, const float* weights, float* dL_dimg,
                                 void* stream) {
    return photometric_backward(n_views, H, W, img, nullptr, nullptr, gt_tab, mask_tab, nullptr, scratch, weights, dL_dimg, stream);
}
// Region-of-interest form (see any_tile above): `tile_count` = the list lengths of the forward that rendered `img`
// (section 1 of ggs_bin_layout: uint32 [n_views][ceil(H/16) * ceil(W/16)]).  The sums are those of the plain form; dL/dimg is
// written only in the boxes that overlap a non-empty tile and left untouched elsewhere.  Images by plain pointers (gt / mask)
// or by tables (gt_tab / mask_tab), as in the two forms above.
int ggs_photometric_forward_roi(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                float* sums, void* scratch, void* stream) {
    return photometric_forward(n_views, H, W, img, gt, mask, gt_tab, mask_tab, tile_count, nullptr, nullptr, sums, scratch, stream);
}
int ggs_photometric_backward_roi(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                 const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                 const void* scratch, const float* weights, float* dL_dimg, void* stream) {
    return photometric_backward(n_views, H, W, img, gt, mask, gt_tab, mask_tab, tile_count, scratch, weights, dL_dimg, stream);
}
// Sparse-mask form of pass A (see loss_stats_stream_body): as ggs_photometric_forward_roi, plus the tile occupancy of the masks
// (ggs_mask_tiles; by plain pointer [n_views][tiles] or by a device table of per-view pointers) -- boxes without a mask pixel
// in their window and without a reader of their maps are skipped.  Same sums (to fp32 rounding of the summation order), same
// maps where pass B reads them: ggs_photometric_backward_roi follows unchanged.  Meant for silhouette masks; on a dense mask
// nothing is skipped and the single-view launch pays for its short bands (~1.6x the plain form).
int ggs_photometric_forward_sparse(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                   const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                   const uint32_t* mask_tiles, const uint32_t* const* mask_tiles_tab,
                                   float* sums, void* scratch, void* stream) {
    if (!mask_tiles && !mask_tiles_tab) {
        ggs_clear_error_();
        return ggs_fail_(GGS_ERR_ARG, "ggs_photometric_forward_sparse: NULL mask tile table");
    }
    return photometric_forward(n_views, H, W, img, gt, mask, gt_tab, mask_tab, tile_count, mask_tiles, mask_tiles_tab, sums,
                               scratch, stream);
}
// tiles[v][ty * ceil(W/16) + tx] = number of non-zero pixels of mask v ([n_views][H][W] floats) in the 16x16 tile (tx, ty)
int ggs_mask_tiles(int n_views, int H, int W, const float* mask, uint32_t* tiles, void* stream) {
    ggs_clear_error_();
    if (n_views <= 0 || H <= 0 || W <= 0) return ggs_fail_(GGS_ERR_ARG, "ggs_mask_tiles: bad sizes");
    if (!mask || !tiles) return ggs_fail_(GGS_ERR_ARG, "ggs_mask_tiles: NULL pointer argument");
    if (n_views > 65535) return ggs_fail_(GGS_ERR_SIZE, "ggs_mask_tiles: n_views too large");
    const int tgx = (W + 15) / 16, tgy = (H + 15) / 16;
    hipLaunchKernelGGL(ggs_k_mask_tiles, dim3((unsigned)(tgx * tgy), (unsigned)n_views), dim3(256), 0, (hipStream_t)stream, H, W,
                       tgx, tgy, mask, tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "ggs_mask_tiles launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

}  // extern "C"
