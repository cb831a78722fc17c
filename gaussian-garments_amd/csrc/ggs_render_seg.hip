// ggs_render_seg.hip -- segment-parallel compositing: the forward / backward walk of ONE tile list spread over several
// waves.
//
// Why: the reference's production loop renders one view per optimisation step (s2_registration.py:241-251, :306).  A single
// 1080p view has ~1.1k non-empty tiles with ~300 (up to ~1000) list entries each for 1024 SIMDs: a launch is bounded by the
// SERIAL walk of its longest list (~50 us of a ~95 us kernel, profiles/r02b_v1_sq_counters.md), not by throughput.  The same
// holds for the heaviest tiles of large scenes.  The walk is sequential per pixel only through the transmittance
//     T_{j+1} = T_j (1 - alpha_j),
// so a list is cut into segments of seg_len entries (a multiple of 64) and
//   1. ggs_k_seg_trans  : every segment but the last of a tile computes, per pixel, its own transmittance factor
//                         prod (1 - alpha_j) over the entries that pass the alpha rules (1/255 skip, 0.99 clamp) -- no
//                         termination rule, alpha evaluation only (~half the forward's arithmetic);
//   2. ggs_k_seg_fwd    : every segment starts from T_start = product of the factors of the segments in front of it and
//                         composites its entries with the full rule set.  The true T never drops below 1e-4 (a splat that
//                         would take it there is not applied and ends the pixel), and the unterminated product equals the
//                         true T up to the terminating splat, so "T_start < 1e-4" <=> "the pixel ended in an earlier
//                         segment": such pixels are parked.  Tiles with one segment write the image directly; the others
//                         write per-segment partial sums (colour, depth, alpha, T at the segment end, last contributor);
//   3. ggs_k_seg_combine: sums the partials of a tile in list order (deterministic), adds the background, writes the
//                         image, final_T and n_contrib (absolute list position, as the unsegmented kernels define it);
//   4. ggs_k_seg_bwd    : the reverse walk of a segment starts from T = T at the segment end (kept from the forward) and
//                         B = T_final (bg . dL/dC) + sum over the segments BEHIND it of (partial colour . dL/dC [+ depth,
//                         alpha terms]) -- the same two-scalar state as ggs_render.hip, just entered in the middle.
// Rules and index semantics are those of ggs_render.hip; what changes is the association of the products / sums across a
// segment boundary, i.e. results agree with the unsegmented kernels to fp32 rounding (~1e-7), not bit for bit.
//
// Work distribution: ggs_k_seg_items turns the LPT-ordered work items into "virtual items" (item, segment); the per-segment
// kernels are launched over an upper bound of their number and the surplus waves exit on the device-side count.  NQ = 1:
// one wave per (virtual item, quadrant) (latency mapping); NQ = 4: one wave per virtual item, 4 pixels per lane.
#include "ggs_render_common.h"

// S0: grid ceil(n_items / 256).  One lane per work item, in LPT order.
__global__ __launch_bounds__(256) void ggs_k_seg_items(SegItemsArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_items) return;
    const uint32_t item = a.order[r];
    const uint32_t L = a.header->overflow ? 0u : a.tile_count[item];
    const uint32_t sl = (uint32_t)a.seg.seg_len;
    const uint32_t S = L > sl ? (L + sl - 1) / sl : 1u;
    const uint32_t pos = atomicAdd(&a.seg.counts[0], S);
    uint32_t slot = 0;
    if (S > 1) {
        slot = atomicAdd(&a.seg.counts[1], S);
        const uint32_t m = atomicAdd(&a.seg.counts[2], 1u);
        if (m < a.seg.max_multi) a.seg.mitem[m] = item;
    }
    a.seg.seg_slot[item] = slot;
    for (uint32_t s = 0; s < S; ++s)
        if (pos + s < a.seg.max_vitems) a.seg.vitem[pos + s] = make_uint2(item, s);
}

// Waves per workgroup: the per-segment kernels do not synchronise their waves, but launching them four to a workgroup
// quarters the number of workgroups the dispatcher has to place (it places about one per cycle: a grid of single-wave
// workgroups that exit at once was measured at ~1 wave per cycle, and these launches hold 20-60k waves of a few us each).
#define SEG_WPB 4
#define SEG_WAVE ((int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))   // wave-uniform: keeps everything derived from it scalar
#define SEG_VB ((int)(blockIdx.x * SEG_WPB) + SEG_WAVE)                 // the wave's own "block" index
#define SEG_LANE ((int)(threadIdx.x & 63))

namespace {

struct SegWork {
    bool ok;
    int v, t, q0, seg, S, L, first0, hi, px0, py0;
    uint32_t slot0;
    size_t base;
};

template <int NQ>
__device__ __forceinline__ SegWork seg_work(const SegInfo& sg, const GgsBinHeader* header, const uint32_t* tile_count,
                                            const uint32_t* tile_offset, const unsigned long long* view_base, int T, int gx) {
    SegWork w;
    const uint32_t vb = (uint32_t)SEG_VB;
    const uint32_t vi = NQ == 4 ? vb : vb >> 2;
    w.q0 = NQ == 4 ? 0 : (int)(vb & 3);
    w.ok = vi < min(sg.counts[0], sg.max_vitems);
    if (!w.ok) return w;
    const uint2 it = sg.vitem[vi];
    w.seg = (int)it.y;
    w.v = (int)(it.x / (uint32_t)T); w.t = (int)(it.x % (uint32_t)T);
    w.L = header->overflow ? 0 : (int)tile_count[it.x];
    w.S = w.L > sg.seg_len ? (w.L + sg.seg_len - 1) / sg.seg_len : 1;
    w.first0 = w.seg * sg.seg_len;
    w.hi = min(w.L, w.first0 + sg.seg_len);
    w.slot0 = sg.seg_slot[it.x];
    w.base = (size_t)view_base[w.v] + tile_offset[it.x];
    const int lane = SEG_LANE;
    w.px0 = (w.t % gx) * GGS_TILE + (lane & 7);
    w.py0 = (w.t / gx) * GGS_TILE + (lane >> 3);
    return w;
}

// Forward of one segment.  TRANS: transmittance factor only (pass 1).
template <int NQ, bool TRANS>
__device__ __forceinline__ void seg_fwd_body(const RenderArgs& a) {
    const SegInfo& sg = a.seg;
    const SegWork w = seg_work<NQ>(sg, a.header, a.tile_count, a.tile_offset, a.view_base, a.T, a.gx);
    if (!w.ok) return;
    if (TRANS && (w.S <= 1 || w.seg == w.S - 1)) return;       // nobody multiplies by the factor of a last segment
    const int lane = SEG_LANE, q0 = w.q0;
    uint32_t* ids = a.ids + w.base;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.rec + (size_t)w.v * a.P);
    const float inf_v = __builtin_inff();

    float pxf[NQ], pyf[NQ], T[NQ], C0[NQ], C1[NQ], C2[NQ], D[NQ], A[NQ];
    uint32_t last[NQ];
    bool inside[NQ], alive[NQ];
    int rem[NQ], remaining = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = w.px0 + ((q0 + q) & 1) * 8, py = w.py0 + ((q0 + q) >> 1) * 8;
        inside[q] = px < a.W && py < a.H;
        float t = 1.f;
        if (!TRANS)                                             // product of the factors in front, in list order
            for (int s = 0; s < w.seg; ++s) t *= sg.seg_T[((size_t)w.slot0 + s) * 256 + (q0 + q) * 64 + lane];
        T[q] = t;
        alive[q] = inside[q] && (w.seg == 0 || t >= GGS_T_MIN);
        pxf[q] = alive[q] ? (float)px : inf_v; pyf[q] = (float)py;
        C0[q] = C1[q] = C2[q] = D[q] = A[q] = 0.f;
        last[q] = 0;
        rem[q] = (int)__popcll(__builtin_amdgcn_ballot_w64(alive[q]));
        remaining += rem[q];
    }

    __shared__ float4 s_all[SEG_WPB][64 * 3];
    float4* s_rec = s_all[SEG_WAVE];
    RoundLds lds{s_rec};
    if (w.hi > w.first0 && remaining != 0) {
        Rec3 nxt = gather_round(rec, ids, w.first0, w.hi, lane);
        for (int first = w.first0; first < w.hi; first += 64) {
            if (remaining == 0) break;
            const Rec3 cur = nxt;
            if (first + 64 < w.hi) nxt = gather_round(rec, ids, first + 64, w.hi, lane);
            const int n = min(64, w.hi - first);
            uint64_t plane[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) plane[q] = 0;
            lds.put(cur, lane);
            uint32_t my_bits = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) my_bits |= 1u << (GGS_ID_BITS + q0 + q);
            uint64_t todo = __builtin_amdgcn_ballot_w64((cur.w & my_bits) != 0);      // entries that reach this wave's pixels
            if (n < 64) todo &= (1ull << n) - 1ull;
            while (todo != 0 && remaining != 0) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur.w, j);
                const float4 ra = s_rec[j * 3 + 0], rb = s_rec[j * 3 + 1], rc = s_rec[j * 3 + 2];
                const float gx = ra.x, gy = ra.y, cxx = ra.z, cxy = ra.w, cyy = rb.x, op = rb.y;
                const float cr = rb.z, cg = rb.w, cb = rc.x, dep = rc.y;
                uint32_t posv;
                asm volatile("v_mov_b32 %0, %1" : "=v"(posv) : "s"(first + j + 1));
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!(word & (1u << (GGS_ID_BITS + q0 + q))) || rem[q] == 0) continue;
                    const float dx = gx - pxf[q], dy = gy - pyf[q];
                    const float power = fmaf(cxx * dx, dx, fmaf(cyy * dy, dy, (cxy * dx) * dy));
                    const float alpha = __builtin_fminf(GGS_ALPHA_MAX, op * __builtin_amdgcn_exp2f(power));
                    const uint64_t m_ok = __builtin_amdgcn_ballot_w64(power <= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= GGS_ALPHA_MIN);
                    if (m_ok == 0) continue;
                    const float wa = alpha * T[q];
                    if (TRANS) {
                        T[q] -= sel_or_zero(m_ok, wa);
                        continue;
                    }
                    const float test_T = T[q] - wa;
                    const uint64_t m_stop = m_ok & __builtin_amdgcn_ballot_w64(test_T < GGS_T_MIN);
                    const uint64_t m_app = m_ok & ~m_stop;
                    const int n_stop = (int)__popcll(m_stop);
                    rem[q] -= n_stop; remaining -= n_stop;
                    pxf[q] = sel(m_stop, inf_v, pxf[q]);
                    const float ww = sel_or_zero(m_app, wa);
                    C0[q] = fmaf(cr, ww, C0[q]);
                    C1[q] = fmaf(cg, ww, C1[q]);
                    C2[q] = fmaf(cb, ww, C2[q]);
                    D[q] = fmaf(dep, ww, D[q]);
                    A[q] += ww;
                    T[q] -= ww;
                    last[q] = __float_as_uint(sel(m_app, __uint_as_float(posv), __uint_as_float(last[q])));
                    plane[q] |= 1ull << j;
                }
            }
            if (!TRANS) {
                // narrowed quadrant masks, as in ggs_render.hip.  NQ == 1: the four quadrant waves share the word and each
                // clears only its own bit (entries this wave never reached keep it: the backward re-tests every pixel)
                if (NQ == 4) {
                    uint32_t neww = cur.w & GGS_ID_MASK;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) neww |= ((uint32_t)(plane[q] >> lane) & 1u) << (GGS_ID_BITS + q);
                    if (lane < n) ids[first + lane] = neww;
                } else {
                    const bool mine = (cur.w >> (GGS_ID_BITS + q0)) & 1u;
                    const bool kept = (plane[0] >> lane) & 1ull;
                    // entries behind the point where the wave stopped (remaining == 0) were not visited: clear them too
                    if (lane < n && mine && !kept) atomicAnd(&ids[first + lane], ~(1u << (GGS_ID_BITS + q0)));
                }
            }
        }
        if (!TRANS && NQ == 1) {
            // rounds never started because every pixel of the quadrant had finished: nothing there blends in this quadrant
            // (the unsegmented kernel leaves those bits set and the backward skips them through n_contrib; same here)
        }
    }

    if (TRANS) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) sg.seg_T[((size_t)w.slot0 + w.seg) * 256 + (q0 + q) * 64 + lane] = T[q];
        return;
    }
    const size_t HW = (size_t)a.H * a.W;
    if (w.S == 1) {
        const float* bg = a.bg + 3 * w.v;
        const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
        float* oc = a.out_color + (size_t)w.v * 3 * HW;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (!inside[q]) continue;
            const size_t pix = (size_t)(w.py0 + ((q0 + q) >> 1) * 8) * a.W + (w.px0 + ((q0 + q) & 1) * 8);
            a.final_T[(size_t)w.v * HW + pix] = T[q];
            a.n_contrib[(size_t)w.v * HW + pix] = last[q];
            oc[pix] = fmaf(T[q], bg0, C0[q]);
            oc[HW + pix] = fmaf(T[q], bg1, C1[q]);
            oc[2 * HW + pix] = fmaf(T[q], bg2, C2[q]);
            a.out_depth[(size_t)w.v * HW + pix] = D[q];
            a.out_alpha[(size_t)w.v * HW + pix] = A[q];
        }
        return;
    }
    float* part = sg.part + ((size_t)w.slot0 + w.seg) * (GGS_SEG_FIELDS * 256);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = (q0 + q) * 64 + lane;
        part[0 * 256 + p] = C0[q]; part[1 * 256 + p] = C1[q]; part[2 * 256 + p] = C2[q];
        part[3 * 256 + p] = D[q]; part[4 * 256 + p] = A[q];
        part[5 * 256 + p] = alive[q] ? T[q] : -1.f;            // -1: ended in front of this segment (or outside the image)
        part[6 * 256 + p] = __uint_as_float(last[q]);
    }
}

// S3: one wave per (multi-segment item[, quadrant]).
template <int NQ>
__device__ __forceinline__ void seg_combine_body(const RenderArgs& a) {
    const SegInfo& sg = a.seg;
    const uint32_t vb = (uint32_t)SEG_VB;
    const uint32_t mi = NQ == 4 ? vb : vb >> 2;
    const int q0 = NQ == 4 ? 0 : (int)(vb & 3);
    if (mi >= min(sg.counts[2], sg.max_multi)) return;
    const uint32_t item = sg.mitem[mi];
    const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T), lane = SEG_LANE;
    const int L = (int)a.tile_count[item];
    const int S = (L + sg.seg_len - 1) / sg.seg_len;
    const float* part0 = sg.part + (size_t)sg.seg_slot[item] * (GGS_SEG_FIELDS * 256);
    const size_t HW = (size_t)a.H * a.W;
    const float* bg = a.bg + 3 * v;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    float* oc = a.out_color + (size_t)v * 3 * HW;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = (t % a.gx) * GGS_TILE + (lane & 7) + ((q0 + q) & 1) * 8;
        const int py = (t / a.gx) * GGS_TILE + (lane >> 3) + ((q0 + q) >> 1) * 8;
        if (px >= a.W || py >= a.H) continue;
        const int p = (q0 + q) * 64 + lane;
        float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f, Tf = 1.f;
        uint32_t last = 0;
        for (int s = 0; s < S; ++s) {
            const float* ps = part0 + (size_t)s * (GGS_SEG_FIELDS * 256);
            const float te = ps[5 * 256 + p];
            if (te < 0.f) continue;                               // the pixel had ended before this segment
            Tf = te;
            C0 += ps[0 * 256 + p]; C1 += ps[1 * 256 + p]; C2 += ps[2 * 256 + p];
            D += ps[3 * 256 + p]; A += ps[4 * 256 + p];
            const uint32_t l = __float_as_uint(ps[6 * 256 + p]);
            if (l) last = l;
        }
        const size_t pix = (size_t)py * a.W + px;
        a.final_T[(size_t)v * HW + pix] = Tf;
        a.n_contrib[(size_t)v * HW + pix] = last;
        oc[pix] = fmaf(Tf, bg0, C0);
        oc[HW + pix] = fmaf(Tf, bg1, C1);
        oc[2 * HW + pix] = fmaf(Tf, bg2, C2);
        a.out_depth[(size_t)v * HW + pix] = D;
        a.out_alpha[(size_t)v * HW + pix] = A;
    }
}

// S4: backward of one segment (see render_bwd_body in ggs_render.hip for the recurrence).
template <bool DA, int NQ>
__device__ __forceinline__ void seg_bwd_body(const RenderBwdArgs& a) {
    const SegInfo& sg = a.seg;
    const SegWork w = seg_work<NQ>(sg, a.header, a.tile_count, a.tile_offset, a.view_base, a.T, a.gx);
    if (!w.ok || w.L == 0) return;
    const int lane = SEG_LANE, q0 = w.q0;
    const size_t HW = (size_t)a.H * a.W;
    const uint32_t* __restrict__ ids = a.ids + w.base;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.rec + (size_t)w.v * a.P);
    GradRec* acc = a.acc + (size_t)w.v * a.P;
    const float* bg = a.bg + 3 * w.v;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    float pxf[NQ], pyf[NQ], T[NQ], B[NQ], dC0[NQ], dC1[NQ], dC2[NQ], dD[NQ], dA[NQ];
    int nc[NQ];
    int maxc = 0;
    uint32_t my_bits = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        my_bits |= 1u << (GGS_ID_BITS + q0 + q);
        const int px = w.px0 + ((q0 + q) & 1) * 8, py = w.py0 + ((q0 + q) >> 1) * 8;
        const bool inside = px < a.W && py < a.H;
        const size_t pix = (size_t)py * a.W + px;
        const int p = (q0 + q) * 64 + lane;
        pxf[q] = (float)px; pyf[q] = (float)py;
        int n_abs = inside ? (int)a.n_contrib[(size_t)w.v * HW + pix] : 0;
        const float Tf = inside ? a.final_T[(size_t)w.v * HW + pix] : 1.f;
        dC0[q] = dC1[q] = dC2[q] = 0.f; dD[q] = dA[q] = 0.f;
        if (inside) {
            const float* dc = a.dL_dcolor + (size_t)w.v * 3 * HW;
            dC0[q] = dc[pix]; dC1[q] = dc[HW + pix]; dC2[q] = dc[2 * HW + pix];
            if (DA) {
                if (a.dL_ddepth) dD[q] = a.dL_ddepth[(size_t)w.v * HW + pix];
                if (a.dL_dalpha) dA[q] = a.dL_dalpha[(size_t)w.v * HW + pix];
            }
        }
        float Tq = Tf, Bq = Tf * (bg0 * dC0[q] + bg1 * dC1[q] + bg2 * dC2[q]);
        if (w.S > 1) {
            const float* ps = sg.part + ((size_t)w.slot0 + w.seg) * (GGS_SEG_FIELDS * 256);
            const float te = ps[5 * 256 + p];
            if (te < 0.f) n_abs = 0;                              // ended in front of this segment: nothing blended here
            else Tq = te;                                         // T behind the last entry of this segment
            for (int s = w.seg + 1; s < w.S; ++s) {               // what lies behind this segment, by whole segments
                const float* pb = sg.part + ((size_t)w.slot0 + s) * (GGS_SEG_FIELDS * 256);
                float sd = fmaf(pb[2 * 256 + p], dC2[q], fmaf(pb[1 * 256 + p], dC1[q], pb[0 * 256 + p] * dC0[q]));
                if (DA) sd += fmaf(pb[3 * 256 + p], dD[q], pb[4 * 256 + p] * dA[q]);
                Bq += sd;
            }
        }
        T[q] = Tq; B[q] = Bq;
        nc[q] = min(n_abs, w.hi);                                 // entries [first0, nc) of this segment were candidates
        maxc = max(maxc, nc[q]);
    }
    const int row = lane >> 4, quad = (lane >> 2) & 3;
    int fld = -1;
    if ((lane & 3) == 0 && quad != 3) {
        const int pr = ((row & 1) << 1) | (row >> 1);
        fld = quad == 0 ? pr : quad == 2 ? 4 + pr : row == 3 ? (DA ? 9 : 8) : (DA && row == 1) ? 8 : -1;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) maxc = max(maxc, __shfl_xor(maxc, d));
    maxc = __builtin_amdgcn_readfirstlane(maxc);
    if (maxc <= w.first0) return;

    __shared__ float4 s_all[SEG_WPB][64 * 3];
    float4* s_rec = s_all[SEG_WAVE];
    RoundLds lds{s_rec};
    const int r_lo = w.first0 >> 6;                               // seg_len is a multiple of 64
    int r = (maxc - 1) >> 6;
    Rec3 nxt = gather_round(rec, ids, r * 64, w.L, lane);
    for (; r >= r_lo; --r) {
        const Rec3 cur = nxt;
        if (r > r_lo) nxt = gather_round(rec, ids, (r - 1) * 64, w.L, lane);
        const int first = r * 64;
        const int n = min(64, maxc - first);
        lds.put(cur, lane);
        uint64_t todo = __builtin_amdgcn_ballot_w64((cur.w & my_bits) != 0);
        if (n < 64) todo &= (1ull << n) - 1ull;
        while (todo) {
            const int j = 63 - __builtin_clzll(todo);
            todo &= ~(1ull << j);
            const int pos = first + j;
            const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur.w, j);
            const float4 ra = s_rec[j * 3 + 0], rb = s_rec[j * 3 + 1], rc = s_rec[j * 3 + 2];
            const float gx = ra.x, gy = ra.y, cxx = ra.z, cxy = ra.w, cyy = rb.x, op = rb.y;
            const float cr = rb.z, cg = rb.w, cb = rc.x, dep = rc.y;
            float v_mx = 0.f, v_my = 0.f, v_cx = 0.f, v_cy = 0.f, v_cz = 0.f, v_op = 0.f;
            float v_r = 0.f, v_g = 0.f, v_b = 0.f, v_dep = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (NQ > 1 && !(word & (1u << (GGS_ID_BITS + q0 + q)))) continue;
                const float dx = gx - pxf[q], dy = gy - pyf[q];
                const float power = fmaf(cxx * dx, dx, fmaf(cyy * dy, dy, (cxy * dx) * dy));
                const float Gr = __builtin_amdgcn_exp2f(power);
                const float ar = __builtin_fminf(GGS_ALPHA_MAX, op * Gr);
                const bool valid = (pos < nc[q]) & (power <= 0.f) & (ar >= GGS_ALPHA_MIN);
                const float alpha = valid ? ar : 0.f;
                const float G = valid ? Gr : 0.f;
                const float ra1 = __builtin_amdgcn_rcpf(1.f - alpha);
                T[q] *= ra1;
                const float ww = alpha * T[q];
                float sdot = fmaf(cb, dC2[q], fmaf(cg, dC1[q], cr * dC0[q]));
                if (DA) sdot += fmaf(dep, dD[q], dA[q]);
                const float dL_da = fmaf(T[q], sdot, -B[q] * ra1);
                B[q] = fmaf(ww, sdot, B[q]);
                v_r = fmaf(ww, dC0[q], v_r); v_g = fmaf(ww, dC1[q], v_g); v_b = fmaf(ww, dC2[q], v_b);
                if (DA) v_dep = fmaf(ww, dD[q], v_dep);
                const float tt = G * dL_da;
                v_op += tt;
                const float hx = tt * dx, hy = tt * dy;
                v_mx += hx; v_my += hy;
                v_cx = fmaf(hx, dx, v_cx);
                v_cy = fmaf(hx, dy, v_cy);
                v_cz = fmaf(hy, dy, v_cz);
            }
            const float Q1 = swap16_add(swap32_add(v_mx, v_my), swap32_add(v_cx, v_cy));
            const float Q2 = swap16_add(swap32_add(v_cz, v_op), swap32_add(v_r, v_g));
            const float Sm = DA ? fold_rows<2>(Q1, Q2, swap32_add(v_b, v_dep)) : fold_rows<4>(Q1, Q2, v_b);
            float* dst = reinterpret_cast<float*>(acc + (word & GGS_ID_MASK));
            if (fld >= 0) atomicAdd(dst + fld, Sm);
        }
    }
}

}  // namespace

__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_trans(RenderArgs a) { seg_fwd_body<4, true>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_trans_quad(RenderArgs a) { seg_fwd_body<1, true>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_fwd(RenderArgs a) { seg_fwd_body<4, false>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_fwd_quad(RenderArgs a) { seg_fwd_body<1, false>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_combine(RenderArgs a) { seg_combine_body<4>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_combine_quad(RenderArgs a) { seg_combine_body<1>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_bwd(RenderBwdArgs a) { seg_bwd_body<false, 4>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_bwd_da(RenderBwdArgs a) { seg_bwd_body<true, 4>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_bwd_quad(RenderBwdArgs a) { seg_bwd_body<false, 1>(a); }
__global__ __launch_bounds__(64 * SEG_WPB) void ggs_k_seg_bwd_da_quad(RenderBwdArgs a) { seg_bwd_body<true, 1>(a); }
