// ggs_common.h -- shared device-side definitions of libggsplat (gfx950 only).
//
// HBM layouts (all per call, caller-allocated, see include/ggsplat.h):
//   geom : SplatRec rec[V][P]            48 B per (view, Gaussian), AoS so that the
//                                        render kernels gather one record = 3 x 16 B
//                                        from one or two cache lines;
//          SplatAux aux[V][P]           16 B: radius + SH clamp bits + reachable-tile bitmask (binning / backward only).
//   bin  : BinHeader | tile_count[V][T] | tile_cursor[V][T] | tile_offset[V][T] |
//          view_base[V] | order[V*T] (work items v*T+t, heaviest tile lists first) |
//          keys[cap] (u64: depth bits << 32 | Gaussian id << 4 | quadrant mask) | ids[cap] (u32:
//          quadrant mask << 28 | id, sorted by (depth bits, id))
//          (BinHeader bytes 64.. hold the 8 list-length bucket counters / cursors of `order`)
//   img  : final_T[V][H*W] f32 | n_contrib[V][H*W] u32   (defined on the pixels of NON-EMPTY tiles only: nothing reads the rest)
//   bwd scratch : GradRec acc[V][P]      48 B per (view, Gaussian), atomically accumulated
//                                        per-Gaussian screen-space gradients.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ggsplat.h"

#define GGS_TILE 16              // tile HEIGHT, and the reference's tile size (the rect rule of A.1 step 6 is in 16-pixel units)
// Tile WIDTH: 16 (the product) or 32 (A/B build `make HIPFLAGS+=-DGGS_TILE_W=32`: 8 pixels per lane, 0.80 x the (tile, splat)
// entries -- VERDICT r3 #1a; measured in profiles/r04_bwd_variants.md).  A tile is GGS_QX x 2 sub-blocks of 8x8 pixels
// ("quadrants"): sub-block q covers x in [8 (q % GGS_QX), +8), y in [8 (q / GGS_QX), +8); one wave64 owns a tile, lane l owns pixel
// (l & 7, l >> 3) of every sub-block.
#ifndef GGS_TILE_W
#define GGS_TILE_W 16
#endif
#define GGS_TS (GGS_TILE_W / 16)     // reference 16-pixel columns per tile
#define GGS_QX (GGS_TILE_W / 8)
#define GGS_NQ (2 * GGS_QX)
#define GGS_BLOCK 256            // threads per render block = one 16x16 tile = 4 wave64
#define GGS_BATCH 256            // splats staged in LDS per round
#define GGS_SORT_CAP 4096        // per-tile list length sorted in LDS (above: global fallback)
// id word of the per-tile lists: low GGS_ID_BITS bits Gaussian id, the GGS_NQ bits above it the sub-block mask (bit q set: the
// splat can reach / was blended in sub-block q of the tile).  Limits P to 2^28 (2^24 with 32-pixel tiles).
#define GGS_ID_BITS (32 - GGS_NQ)
#define GGS_ID_MASK ((1u << GGS_ID_BITS) - 1u)
#define GGS_NBUCKET 16            // list-length classes used to order the per-tile work items
#define GGS_BUCKET_COUNT_OFF 64   // byte offsets inside the header region
#define GGS_BUCKET_CURSOR_OFF 128
// XCD-aware work order (ggs_k_order_tiles): the same histogram and cursors per (list-length class, XCD region of the tile)
#define GGS_NREGION 8
#define GGS_REGION_COUNT_OFF 256      // [GGS_NBUCKET][GGS_NREGION] u32
#define GGS_REGION_CURSOR_OFF 768     // [GGS_NBUCKET][GGS_NREGION] u32
#define GGS_HEADER_BYTES 2048

// Constants of the algorithm, one line each.
#define GGS_NEAR_Z 0.2f
#define GGS_W_EPS 0.0000001f
#define GGS_FOV_CLAMP 1.3f
#define GGS_LOWPASS 0.3f
#define GGS_LAMBDA_FLOOR 0.1f
#define GGS_ALPHA_MAX 0.99f
#define GGS_ALPHA_MIN (1.0f / 255.0f)
#define GGS_T_MIN 0.0001f
#define GGS_DET_EPS 0.0000001f

// The conic is stored PRE-SCALED for the compositing loops: with d = mean - pixel,
//   log2(G) = qa dx^2 + qb dx dy + qc dy^2,   qa = -log2(e)/2 conic.x, qb = -log2(e) conic.y, qc = -log2(e)/2 conic.z
// so the exponent costs 3 mul + 2 fma + one v_exp_f32 per pixel instead of 9 VALU ops + mul + v_exp.
#define GGS_KA (-0.72134752044448170f)   // -log2(e) / 2
#define GGS_KB (-1.44269504088896340f)   // -log2(e)
#define GGS_LOG2E 1.44269504088896340f

struct SplatRec {                // float index
    float px, py, cx, cy;        // 0..3   pixel mean, qa, qb  (pre-scaled conic, see above)
    float cz, opacity, r, g;     // 4..7   qc, opacity, colour r, g
    float b, depth;              // 8..9   colour b, view-space depth
    unsigned bbx, bby;           // 10..11 pixel AABB of the region where alpha can reach 1/255:
                                 //        int16 min | int16 max << 16 (conservative; empty if min > max)
};
static_assert(sizeof(SplatRec) == 48, "SplatRec must be 48 bytes");

struct SplatAux {
    int radius;                  // 3-sigma radius in px (the `radii` output); 0 = culled
    unsigned clamped;            // bit c set: SH colour channel c was clamped at 0
    unsigned long long tile_bits;  // GGS_NQ bits per tile of the culled rect, row-major (tile index ry * rect_w + rx): the sub-blocks of
                                   // tile (x0 + rx, y0 + ry) the splat can be blended in (ggs_quad_bits; 0 = not on that tile's list);
                                   // valid when the rect has <= GGS_TILE_BITS_MAX tiles, else recomputed by the scatter pass
};
static_assert(sizeof(SplatAux) == 16, "SplatAux must be 16 bytes");
#define GGS_TILE_BITS_MAX (64 / GGS_NQ)      // tiles of a culled rect whose GGS_NQ-bit sub-block masks fit SplatAux.tile_bits

struct GradRec {                 // per-(view, Gaussian) accumulators over the pixels that blended the splat,
                                 // with t = G dL/dalpha and d = mean - pixel:
    float mx, my;                // sum t dx, sum t dy          (-> d/d pixel mean = -opacity (conic . sums))
    float cx, cy, cz;            // sum t dx^2, sum t dx dy, sum t dy^2   (-> d/d conic = -opacity (1/2, 1, 1/2) .)
    float opacity;               // sum t                       (= d/d opacity)
    float r, g, b;
    float depth;
    float pad0, pad1;
};
static_assert(sizeof(GradRec) == 48, "GradRec must be 48 bytes");

// Segmented backward of the latency mapping (round 6).  A single view does not fill the chip, and its backward used to be as long
// as the walks of its longest lists (~1000 entries).  The REVERSE walk can be cut exactly where the forward left a checkpoint: at
// list position b the backward needs, per pixel, the transmittance in front of entry b (the forward's own T at that point) and
//   B_b = T_final (bg . dL/dC) + sum_{k >= b} w_k (c_k . dL/dC)  =  T_final (bg . dL/dC) + (C_final - C_b) . dL/dC
// -- the colour the forward had accumulated at b against the colour it ended with.  The per-quadrant forward stores {T, C0, C1, C2} of
// its 64 pixels in front of every round (list positions that are multiples of GGS_SEG = 64), and its final accumulators once per
// tile with more than GGS_SEG entries; the backward then runs one wave per (tile, segment of GGS_SEG positions), all independent:
// ~5 000 walks of <= 64 entries instead of ~1 100 walks of up to ~1 000.  (Gradients of the depth / alpha outputs -- no loss of the
// reference has them -- would need D and A in the records too: that backward keeps the unsegmented per-quadrant walk.)
//   slot of the checkpoint at list position b of a tile whose list starts at global entry `base`:  (base + b) / GGS_SEG
//   slot of the tile's final record:  n_slots + base / GGS_SEG        (both collision-free: lists do not overlap)
//   record layout [slot][GGS_NQ][GGS_CKPT_PLANES][64] floats, behind final_T | n_contrib in the img workspace (latency mapping only).
#define GGS_SEG 64
#define GGS_CKPT_PLANES 4
#define GGS_MAX_SEG 24           // segments per list; the last scheduled one takes everything behind it
static inline size_t ggs_ckpt_slots(size_t cap) { return cap / GGS_SEG + 2; }
static inline size_t ggs_ckpt_bytes(size_t cap) { return 2 * ggs_ckpt_slots(cap) * (size_t)GGS_NQ * GGS_CKPT_PLANES * 64 * 4; }

struct BinLayout {               // byte offsets inside the binning buffer
    size_t header, tile_count, tile_cursor, tile_offset, view_base, order, keys, ids, total;
    size_t zero_bytes;           // header + tile_count + tile_cursor are cleared each forward
};

static inline size_t ggs_align(size_t x) { return (x + 255) & ~(size_t)255; }

static inline BinLayout ggs_bin_layout(int V, int T, size_t cap) {
    BinLayout L;
    size_t o = 0;
    L.header = o;      o += GGS_HEADER_BYTES;
    L.tile_count = o;  o += ggs_align((size_t)V * T * 4);
    L.tile_cursor = o; o += ggs_align((size_t)V * T * 4);
    L.zero_bytes = o;
    L.tile_offset = o; o += ggs_align((size_t)V * T * 4);
    L.view_base = o;   o += ggs_align((size_t)V * 8);
    L.order = o;       o += ggs_align((size_t)V * T * 4);
    L.keys = o;        o += ggs_align(cap * 8);
    L.ids = o;         o += ggs_align(cap * 4);
    L.total = o;
    return L;
}

#ifdef __HIPCC__
__device__ __forceinline__ float ggs_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float ggs_max(float a, float b) { return a > b ? a : b; }

// Work-item class of a tile by list length: 0 = longest lists ... 6 = 1..63 splats, 7 = empty.
// The per-tile kernels walk `order[]` (class 0 first): the longest serial chains start first and the
// tail of every launch is made of short / empty tiles (LPT scheduling), instead of whatever the last
// view happens to contain.
// Half-octave classes from 64 splats up (class 0: >= 3072, 1: 2048.., 2: 1536.., 3: 1024.., 4: 768.., 5: 512.., ...,
// 11: 64..95), 12: 32..63, 13: < 32, 15: empty.  With whole octaves a 1000-splat tile could start after a 520-splat one.
__device__ __forceinline__ int ggs_len_bucket(uint32_t L) {
    if (L == 0) return GGS_NBUCKET - 1;
    const int lg = 31 - __clz((int)L);
    if (lg >= 12) return 0;
    if (lg <= 4) return 13;
    if (lg == 5) return 12;
    const int half = (int)((L >> (lg - 1)) & 1u);          // second most significant bit
    return 2 * (11 - lg) + (1 - half);
}

// XCD region of a tile.  Workgroup b of a launch runs on XCD b % 8 (observed), every XCD has its own L2, and a splat's record
// is gathered by every tile it touches: blocks of 4 x 4 tiles (64 x 64 px -- a splat's tiles almost always lie inside one) are
// dealt round-robin to the 8 regions, and ggs_k_order_tiles places a tile of region g at a work-item position whose XCD is g
// wherever the class sizes allow, so that the tiles sharing a record share an L2.
__device__ __forceinline__ int ggs_tile_region(int t, int gx) {
    const int tx = t % gx, ty = t / gx;
    return ((tx >> 2) + (ty >> 2) * ((gx + 3) >> 2)) & (GGS_NREGION - 1);
}

// Position in order[] of the r-th non-empty work item (longest first); mirrors ggs_k_order_tiles.
struct NonEmptyItems {
    uint32_t n, stride;        // n non-empty items, at order[r * stride]
    uint32_t n_long;           // the first n_long of them have >= 1024 splats (classes 0..3)
};
__device__ __forceinline__ NonEmptyItems ggs_nonempty_items(const uint32_t* bucket_count, uint32_t n_items) {
    const uint32_t E = bucket_count[GGS_NBUCKET - 1];
    NonEmptyItems it;
    it.n = n_items - E;
    it.stride = (it.n ? (E / it.n) & ~1u : 0) + 1;
    it.n_long = bucket_count[0] + bucket_count[1] + bucket_count[2] + bucket_count[3];
    return it;
}

// Which (list, segment) a block of the segmented backward walks.  Ranks below it.n are segment 0 of the non-empty work items,
// as ever.  Segment k >= 1 (list positions >= k GGS_SEG) exists for the lists longer than k GGS_SEG; order[] holds the lists by
// length CLASS, longest first, so "the lists of at least lenlo[c] entries" are exactly the first cum[c] ranks, and segment k is
// given to the first m_k = cum[c(k)] ranks, c(k) the class with the largest lower bound <= k GGS_SEG (a superset of the lists that
// have the segment: the few blocks too many return at once).  The later segments ride on the blocks of EMPTY tiles, which have
// nothing to do: spare rank e = rank - it.n takes the (e - offset_k)-th list of segment k, offset_k = m_1 + ... + m_(k-1).  When the
// spare ranks do not reach all segments, the last one that fits (k = n_extra) is open-ended for every list: correct for any count,
// and every block computes the same numbers from the bin header.
struct SegItem { uint32_t rank; int seg, n_extra; bool valid; };
__device__ __forceinline__ SegItem ggs_seg_item(const uint32_t* bucket_count, const NonEmptyItems& it, uint32_t n_items, uint32_t rank) {
    SegItem r;
    r.rank = rank; r.seg = 0; r.n_extra = 0; r.valid = rank < it.n;
    // lower bounds of the list-length classes 0..11 (ggs_len_bucket) and their cumulative counts
    const int lenlo[12] = {3072, 2048, 1536, 1024, 768, 512, 384, 256, 192, 128, 96, 64};
    uint32_t cum[12], run = 0;
#pragma unroll
    for (int c = 0; c < 12; ++c) { run += bucket_count[c]; cum[c] = run; }
    const uint32_t n_spare = n_items - it.n;
    uint32_t offset = 0;
    const uint32_t e = rank - it.n;            // (meaningful for spare ranks only)
    int c = 11;
    for (int k = 1; k < GGS_MAX_SEG; ++k) {
        while (c > 0 && lenlo[c - 1] <= k * GGS_SEG) --c;      // class with the largest lower bound <= k GGS_SEG
        const uint32_t m = cum[c];
        if (m == 0 || offset + m > n_spare) break;
        r.n_extra = k;
        if (rank >= it.n && e >= offset && e < offset + m) { r.rank = e - offset; r.seg = k; r.valid = true; }
        offset += m;
    }
    if (r.seg > r.n_extra) r.valid = false;
    return r;
}

// Tile rectangle of a splat (A.1 step 6) in the REFERENCE's 16x16 tiles (gx = ceil(W / 16) columns); must be bit-identical
// wherever it is recomputed.
__device__ __forceinline__ void ggs_tile_rect(float px, float py, float r, int gx, int gy, int& x0, int& y0,
                                              int& x1, int& y1) {
    int a0 = (int)((px - r) / (float)GGS_TILE), b0 = (int)((py - r) / (float)GGS_TILE);
    int a1 = (int)((px + r + (float)(GGS_TILE - 1)) / (float)GGS_TILE);
    int b1 = (int)((py + r + (float)(GGS_TILE - 1)) / (float)GGS_TILE);
    x0 = a0 < 0 ? 0 : (a0 > gx ? gx : a0);
    y0 = b0 < 0 ? 0 : (b0 > gy ? gy : b0);
    x1 = a1 < 0 ? 0 : (a1 > gx ? gx : a1);
    y1 = b1 < 0 ? 0 : (b1 > gy ? gy : b1);
}

// Conservative pixel AABB of {alpha >= 1/255} for a splat: d^T C d <= 2 tau with
// tau = ln(255 opacity), so |dx| <= sqrt(2 tau Sigma_xx).  Margins make it a strict superset of
// what the per-pixel fp32 test accepts, so using it to skip work never changes a result.
__device__ __forceinline__ void ggs_alpha_bbox(float px, float py, float var_x, float var_y, float opacity,
                                               unsigned& bbx, unsigned& bby) {
    const float tau = (opacity > 0.f ? logf(255.f * opacity) : -1.f) * 1.01f + 0.02f;
    int xmin = 1, xmax = 0, ymin = 1, ymax = 0;
    if (tau > 0.f) {
        const float ex = sqrtf(2.f * tau * var_x) * 1.001f + 0.01f;
        const float ey = sqrtf(2.f * tau * var_y) * 1.001f + 0.01f;
        const float lo = -32768.f, hi = 32767.f;
        xmin = (int)ggs_max(lo, ggs_min(hi, floorf(px - ex)));
        xmax = (int)ggs_max(lo, ggs_min(hi, ceilf(px + ex)));
        ymin = (int)ggs_max(lo, ggs_min(hi, floorf(py - ey)));
        ymax = (int)ggs_max(lo, ggs_min(hi, ceilf(py + ey)));
    }
    bbx = ((unsigned)xmin & 0xffffu) | ((unsigned)xmax << 16);
    bby = ((unsigned)ymin & 0xffffu) | ((unsigned)ymax << 16);
}
__device__ __forceinline__ int ggs_bb_min(unsigned w) { return (int)(w << 16) >> 16; }
__device__ __forceinline__ int ggs_bb_max(unsigned w) { return (int)w >> 16; }

// Tiles of the reference rectangle [x0,x1) x [y0,y1) that the alpha AABB can reach.
__device__ __forceinline__ void ggs_cull_rect(unsigned bbx, unsigned bby, int& x0, int& y0, int& x1, int& y1) {
    const int tx0 = ggs_bb_min(bbx) >> 4, tx1 = (ggs_bb_max(bbx) >> 4) + 1;
    const int ty0 = ggs_bb_min(bby) >> 4, ty1 = (ggs_bb_max(bby) >> 4) + 1;
    x0 = x0 > tx0 ? x0 : tx0; x1 = x1 < tx1 ? x1 : tx1;
    y0 = y0 > ty0 ? y0 : ty0; y1 = y1 < ty1 ? y1 : ty1;
}

// ---- exact (conservative) footprint test -----------------------------------------------------------
// alpha >= 1/255 inside a pixel box [x0,x1] x [y0,y1] needs  min over the box of  q(d) = A dx^2 + 2 B dx dy + C dy^2
// <= 2 tau  (d = pixel - mean, (A,B,C) = conic, tau = ln(255 opacity) with safety margins).  q is convex: the
// minimum over the box is 0 if the mean is inside, else it sits on one of the four edges, where it is a
// clamped 1-D parabola.  The box is continuous, so this is a superset of what any pixel centre can pass.
struct Footprint { float mx, my, A, B, C, lim, BoC, BoA; };   // lim = 2 tau (<= 0: never blended); B/C, B/A

// From the STORED record fields (qa, qb, qc pre-scaled): the quadratic form is scaled by log2(e)/2, and so is
// the limit.  Histogram and scatter both build it from the record, so they agree bit for bit.
__device__ __forceinline__ Footprint ggs_footprint(float px, float py, float qa, float qb, float qc, float opacity) {
    Footprint f;
    f.mx = px; f.my = py; f.A = -qa; f.B = -0.5f * qb; f.C = -qc;
    const float tau = (opacity > 0.f ? logf(255.f * opacity) : -1.f) * 1.01f + 0.02f;
    f.lim = tau * GGS_LOG2E;
    f.BoC = f.B / f.C; f.BoA = f.B / f.A;
    return f;
}

// Only the edges FACING the mean can hold the minimum: from any point of the box the segment to the mean decreases q and leaves
// the box through one of them.  Branch-free: ONE vertical and ONE horizontal edge are always evaluated -- the facing one where there
// is one, else an arbitrary one, whose minimum is the value at a box point and therefore >= the box minimum (harmless in the min).

// Which 8x8 sub-blocks of tile (tx, ty) the splat can be blended in: bit q set = sub-block q (x in [8 (q % GGS_QX), +8), y in
// [8 (q / GGS_QX), +8) of the tile) is reachable.  0 = the tile is NOT on the splat's list (output-invariant: no pixel centre of the
// tile can reach alpha >= 1/255) -- list membership and quadrant mask are ONE decision since round 6 (the histogram pass computes
// it once per (splat, tile) and leaves it in SplatAux.tile_bits; the scatter pass reads it back).
// The test per sub-block is ggs_box_reachable's: the minimum of q over the box sits on an edge facing the mean, a clamped 1-D
// parabola.  On the regular 8-pixel grid of a tile the per-COLUMN terms of the vertical-edge parabolas (dx, A dx^2, 2 B dx, the
// unclamped minimiser -B/C dx) and the per-ROW terms of the horizontal ones are shared by the sub-blocks of that column / row:
// they are formed once per tile, and a sub-block costs two clamps (v_med3) and four fused multiply-adds.
// c0, c1: the splat's (culled) reference rectangle in 16-pixel columns [c0, c1): with tiles wider than 16 pixels the reference
// still cuts the splat at ITS tile columns, so sub-blocks outside them stay clear.
__device__ __forceinline__ float ggs_med3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
__device__ __forceinline__ unsigned ggs_quad_bits(const Footprint& f, unsigned bbx, unsigned bby, int tx, int ty, int c0, int c1) {
    const int xmin = ggs_bb_min(bbx), xmax = ggs_bb_max(bbx), ymin = ggs_bb_min(bby), ymax = ggs_bb_max(bby);
    const int ox = tx * GGS_TILE_W, oy = ty * GGS_TILE;
    // per column of sub-blocks: the x interval relative to the mean, the vertical edge facing the mean and its parabola terms
    float xl[GGS_QX], xh[GGS_QX], vA[GGS_QX], vB[GGS_QX], vT[GGS_QX];
    bool inx[GGS_QX], colok[GGS_QX];
#pragma unroll
    for (int cx = 0; cx < GGS_QX; ++cx) {
        const int bx = ox + 8 * cx;
        xl[cx] = (float)bx - f.mx; xh[cx] = (float)(bx + 7) - f.mx;
        inx[cx] = xl[cx] <= 0.f && xh[cx] >= 0.f;
        const float dx = xh[cx] < 0.f ? xh[cx] : xl[cx];                 // mean right of the box: its right edge, else the left one
        vA[cx] = f.A * dx * dx; vB[cx] = 2.f * f.B * dx; vT[cx] = -f.BoC * dx;
        colok[cx] = (GGS_TS == 1 || ((bx >> 4) >= c0 && (bx >> 4) < c1)) && xmin <= bx + 7 && xmax >= bx;
    }
    unsigned m = 0;
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
        const int by = oy + 8 * cy;
        const float yl = (float)by - f.my, yh = (float)(by + 7) - f.my;
        const bool iny = yl <= 0.f && yh >= 0.f;
        const float dy = yh < 0.f ? yh : yl;                             // mean below the box: its bottom edge, else the top one
        const float hC = f.C * dy * dy, hB = 2.f * f.B * dy, hT = -f.BoA * dy;
        const bool rowok = ymin <= by + 7 && ymax >= by;
#pragma unroll
        for (int cx = 0; cx < GGS_QX; ++cx) {
            // vertical edge: x fixed, y clamped to the box; horizontal edge: y fixed, x clamped
            const float ey = ggs_med3(vT[cx], yl, yh);
            const float qv = fmaf(ey, fmaf(f.C, ey, vB[cx]), vA[cx]);
            const float ex = ggs_med3(hT, xl[cx], xh[cx]);
            const float qh = fmaf(ex, fmaf(f.A, ex, hB), hC);
            const bool reach = (inx[cx] && iny) ? f.lim > 0.f : ggs_min(qv, qh) <= f.lim;
            if (rowok && colok[cx] && reach) m |= 1u << (cy * GGS_QX + cx);
        }
    }
    return m;
}

// [x0, x1) in the reference's 16-pixel columns -> tile columns [X0, X1)
__device__ __forceinline__ void ggs_tile_columns(int x0, int x1, int& X0, int& X1) {
    X0 = x0 / GGS_TS; X1 = (x1 + GGS_TS - 1) / GGS_TS;
}

// Rotation matrix (row-major) of a (w,x,y,z) quaternion, no normalisation (A.0).
__device__ __forceinline__ void ggs_quat_R(const float* q, float* R) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R S^2 R^T as (xx,xy,xz,yy,yz,zz) via M[k][i] = s_k R[i][k].
__device__ __forceinline__ void ggs_cov3d(const float* scale, float mod, const float* q, float* c6) {
    float R[9], M[9];
    ggs_quat_R(q, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = mod * scale[k];
#pragma unroll
        for (int i = 0; i < 3; ++i) M[k * 3 + i] = s * R[i * 3 + k];
    }
    c6[0] = M[0] * M[0] + M[3] * M[3] + M[6] * M[6];
    c6[1] = M[0] * M[1] + M[3] * M[4] + M[6] * M[7];
    c6[2] = M[0] * M[2] + M[3] * M[5] + M[6] * M[8];
    c6[3] = M[1] * M[1] + M[4] * M[4] + M[7] * M[7];
    c6[4] = M[1] * M[2] + M[4] * M[5] + M[7] * M[8];
    c6[5] = M[2] * M[2] + M[5] * M[5] + M[8] * M[8];
}

// EWA projection pieces shared by forward and backward (A.1 step 4).
struct Ewa { float tx, ty, tz, fx, fy, xmul, ymul, M[6]; };

__device__ __forceinline__ void ggs_ewa(const float* view, const float* m, float tanfovx, float tanfovy, int W,
                                        int H, Ewa& e) {
    float t0 = view[0] * m[0] + view[4] * m[1] + view[8] * m[2] + view[12];
    float t1 = view[1] * m[0] + view[5] * m[1] + view[9] * m[2] + view[13];
    float t2 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
    float limx = GGS_FOV_CLAMP * tanfovx, limy = GGS_FOV_CLAMP * tanfovy;
    float txtz = t0 / t2, tytz = t1 / t2;
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    e.tx = ggs_min(limx, ggs_max(-limx, txtz)) * t2;
    e.ty = ggs_min(limy, ggs_max(-limy, tytz)) * t2;
    e.tz = t2;
    e.fx = (float)W / (2.f * tanfovx);
    e.fy = (float)H / (2.f * tanfovy);
    float j00 = e.fx / e.tz, j02 = -(e.fx * e.tx) / (e.tz * e.tz);
    float j11 = e.fy / e.tz, j12 = -(e.fy * e.ty) / (e.tz * e.tz);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        e.M[j] = j00 * view[4 * j + 0] + j02 * view[4 * j + 2];
        e.M[3 + j] = j11 * view[4 * j + 1] + j12 * view[4 * j + 2];
    }
}

__device__ __forceinline__ void ggs_sym6_mul(const float* c6, const float* v, float* o) {
    o[0] = c6[0] * v[0] + c6[1] * v[1] + c6[2] * v[2];
    o[1] = c6[1] * v[0] + c6[3] * v[1] + c6[4] * v[2];
    o[2] = c6[2] * v[0] + c6[4] * v[1] + c6[5] * v[2];
}

// SH basis (degree <= 3), same polynomials / constants as utils/sh_utils.py:25-111.
#define GGS_SH_C0 0.28209479177387814f
#define GGS_SH_C1 0.4886025119029199f
template <int DEG>
__device__ __forceinline__ void ggs_sh_basis(const float* d, float* b) {
    b[0] = GGS_SH_C0;
    if constexpr (DEG > 0) {
        float x = d[0], y = d[1], z = d[2];
        b[1] = -GGS_SH_C1 * y; b[2] = GGS_SH_C1 * z; b[3] = -GGS_SH_C1 * x;
        if constexpr (DEG > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = 1.0925484305920792f * xy; b[5] = -1.0925484305920792f * yz;
            b[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
            b[7] = -1.0925484305920792f * xz; b[8] = 0.5462742152960396f * (xx - yy);
            if constexpr (DEG > 2) {
                b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
                b[10] = 2.890611442640554f * xy * z;
                b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
                b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
                b[14] = 1.445305721320277f * z * (xx - yy);
                b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
            }
        }
    }
}
#endif  // __HIPCC__
