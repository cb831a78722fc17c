// ggs_render.hip -- per-tile front-to-back compositing (forward) and the per-pixel
// reverse walk that accumulates per-splat gradients (backward).
//
// CDNA4 mapping: ONE wave64 owns ONE 16x16 tile.  Lane l holds 4 pixels, one in each 8x8
// quadrant q of the tile (x = 8(q&1) + (l&7), y = 8(q>>1) + (l>>3)), so a quadrant a splat
// does not reach is skipped with a single exec-mask branch, exactly like a 4-wave block
// would skip a wave -- but the per-splat overhead (record fetch, loop, reduction) is paid
// once per tile instead of four times, and there is no workgroup barrier at all:
//   * the tile's depth-sorted id list is walked in rounds of 64: lane i gathers the 48-byte
//     record of splat (base + i) with three 16-byte loads (the next round is prefetched while
//     the current one is composited) and parks it in a wave-private 3 KB LDS slice;
//   * splat j of the round is read back by all lanes with wave-uniform (broadcast) ds_read_b128:
//     the LDS pipe issues beside the VALU (11 v_readlane per splat cost 17 % of the kernel time);
//   * backward: the 4 pixels of a lane are summed in registers, the nine (ten) per-splat sums cross the 64 lanes through a
//     wave-private LDS transpose (ggs_render_common.h lds_transpose_reduce: 8 ds_write_addtid + 4 ds_read_b64 + 10 VALU; the
//     ninth value rides a DPP chain) so that each total sits in its own lane of ONE register, and a single vector float-atomic
//     instruction adds them to the splat's GradRec (record address formed on the scalar unit).
//
// Roofline: HBM nominally (algorithmic bytes per view: forward 44 N + 28 HW + 16 T, backward 20 HW + 44 N + 36 P_vis); in practice
// both kernels are bound by VALU issue: backward 2 685 VALU instructions per wave at 65 % lane activity, ~0.89 of the SIMDs' issue
// cycles (profiles/r04_sh0_valu.md, r04_sh0_sq_counters.md), HBM traffic 0.64 x / 0.97 x the algorithmic count for backward /
// forward.  Priced per instruction class with measured issue rates (profiles/r02_valu_issue_rates.md, r04_isa_audit.md): backward
// ~45 cycles of per-entry overhead + ~115 per active quadrant (1.6-1.9 per entry) + ~60 for the reduction; forward ~44 per visited
// quadrant (2.09 per entry) + ~40 per blended one.  Round 5 (profiles/r05_pipeline_overlap.md): run side by side on two streams the
// two kernels DO interleave and each slows down by what the other takes -- there is no idle issue slot for a second kernel to use.
// Launches too small to fill the chip (a single view: ~1 100 non-empty tiles for 1 024 SIMDs) take the LATENCY mapping: the forward
// one wave per (tile, quadrant) with two entries per pass (render_fwd_quadwave), which also leaves {T, C} checkpoints in front of every
// round; the backward this same tile-wave body per SEGMENT of 64 list positions, started from those checkpoints (round 6,
// ggs_common.h GGS_SEG; only a backward with depth / alpha gradients still walks whole lists, per quadrant, two entries per pass).
// Mappings examined and not adopted (splat-major / systolic, larger tiles, MFMA moments, mod-8 lane folding, deferred / all-LDS
// reductions): profiles/r03_bwd_mapping_study.md, r04_bwd_variants.md.  The variants that were BUILT for those measurements (the
// permlane-swap butterfly of rounds 1-3, the all-values-through-LDS reduction, what-if builds that drop a stage and compute
// wrong results on purpose) live outside this translation unit as patches: tools/dbg/variants/*.patch (tools/dbg/build_variant.sh).

#include "ggs_render_common.h"

namespace {

// Empty tile (~90 % of the tiles of a view) fully inside the image: nothing to composite, the wave only stores the
// background.  Lane l owns 4 consecutive pixels of row l / 4, so each plane leaves as ONE 16-byte store per lane
// (16 rows x 64 contiguous bytes) instead of four 4-byte stores in the quadrant layout.  Where most of a launch is background
// (config 5: 4K frames) the forward is bound by exactly these stores -- and 9 of 10 bytes the forward writes at config 2 come
// from here -- so only the five OUTPUT planes are written: the per-pixel workspace (final_T, n_contrib) of an empty tile is
// never read (the backward and ggs_count_blends return on tile_count == 0 before touching it) and stays undefined
// (forward 27-29 -> 25 us per view at config 2, 109 -> 100 at config 5).  Returns false when the tile must take the
// general path (image edge, row pitch not a multiple of 16 bytes).
__device__ __forceinline__ bool store_empty_tile(const RenderArgs& a, int v, int ox, int oy, int lane) {
    if ((a.W & 3) != 0 || ox + GGS_TILE_W > a.W || oy + GGS_TILE > a.H) return false;
    const size_t HW = (size_t)a.H * a.W;
    const float* bg = a.bg + 3 * v;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
#pragma unroll
    for (int h = 0; h < GGS_TS; ++h) {           // 16 rows x 64 contiguous bytes per 16-pixel column block of the tile
        const size_t pix = (size_t)(oy + (lane >> 2)) * a.W + ox + 16 * h + (lane & 3) * 4;
        float* oc = a.out_color + (size_t)v * 3 * HW + pix;
        *reinterpret_cast<float4*>(oc) = make_float4(b0, b0, b0, b0);
        *reinterpret_cast<float4*>(oc + HW) = make_float4(b1, b1, b1, b1);
        *reinterpret_cast<float4*>(oc + 2 * HW) = make_float4(b2, b2, b2, b2);
        *reinterpret_cast<float4*>(a.out_depth + (size_t)v * HW + pix) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(a.out_alpha + (size_t)v * HW + pix) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return true;
}

// K4b body.  NQ = 4: one wave per tile, lane = 4 pixels (one per quadrant) -- the throughput mapping.
// (The latency mapping for launches too small to fill the chip -- one wave per (tile, quadrant), grid 4x larger:
// a single view has ~1.1k non-empty tiles of ~300 splats for 1024 SIMDs and is bounded by the serial walk of its
// longest tile -- is render_fwd_quadwave below; the backward template holds both walks: NQ = 4 per tile or per (tile, segment),
// NQ = 1 per (tile, quadrant) for the depth / alpha case.)
// Same arithmetic per pixel, bit-identical results.
// CENSUS (ggs_k_count_forward_visits, a diagnostic: bench.py's evaluated-against-blended pair counts): the same walk with the
// same tests on the lists as the binning left them, nothing stored -- it counts the quadrant passes it makes instead.
template <bool CENSUS>
__device__ __forceinline__ void render_fwd_body(const RenderArgs& a, unsigned long long* census = nullptr) {
    constexpr int NQ = GGS_NQ, q0 = 0;
    unsigned n_visit = 0, n_blend = 0, n_entry = 0;
    // A forward that overflowed its binning capacity (only possible with a static capacity inside a captured graph) has no
    // lists: every tile is composited as EMPTY, so the outputs are deterministic (background, zero depth / alpha)
    // instead of uninitialised memory.  The overflow word tells the caller to re-run.
    const bool overflow = a.header->overflow != 0;
    const uint32_t item = a.order[blockIdx.x];   // work items, longest lists first
    const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T), lane = threadIdx.x;
    const int tx = t % a.gx, ty = t / a.gx;
    const int ox = tx * GGS_TILE_W, oy = ty * GGS_TILE;
    const int px0 = ox + (lane & 7), py0 = oy + (lane >> 3);
    const int L = overflow ? 0 : (int)a.tile_count[(size_t)v * a.T + t];
    const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    uint32_t* ids = a.ids + base;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.rec + (size_t)v * a.P);

    if (CENSUS && L == 0) return;
    if (!CENSUS && L == 0 && store_empty_tile(a, v, ox, oy, lane)) return;

    float pxf[NQ], pyf[NQ], T[NQ], C0[NQ], C1[NQ], C2[NQ], D[NQ], A[NQ];
    uint32_t last[NQ];
    bool inside[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int px = px0 + ((q0 + q) % GGS_QX) * 8, py = py0 + ((q0 + q) / GGS_QX) * 8;
        inside[q] = px < a.W && py < a.H;
        // a finished pixel parks its x coordinate at +inf: the falloff exponent becomes -inf or NaN and
        // every later splat fails the (power <= 0, alpha >= 1/255) test without a separate flag
        pxf[q] = inside[q] ? (float)px : __builtin_inff(); pyf[q] = (float)py;
        T[q] = 1.f; C0[q] = C1[q] = C2[q] = D[q] = A[q] = 0.f;
        last[q] = 0;
    }

    __shared__ float4 s_rec[64 * 3];
    RoundLds lds{s_rec};
    // pixels of this wave still compositing: a wave-uniform counter kept on the scalar unit (s_bcnt1 of the stop
    // mask), so the walk ends at the very splat that finishes the tile without any per-splat VALU test
    const float inf_v = __builtin_inff();
    int rem[NQ], remaining = 0;     // per quadrant too: a finished quadrant is skipped by the same scalar branch as a masked one
#pragma unroll
    for (int q = 0; q < NQ; ++q) { rem[q] = (int)__popcll(__builtin_amdgcn_ballot_w64(inside[q])); remaining += rem[q]; }
    // `live`: sub-blocks with unfinished pixels, positioned like the id word's mask -- one and + one bit test per sub-block instead
    // of the (mask bit, rem[q]) pair (the CU's one scalar unit is 72 % busy in this kernel); the record reads stay in front of it.
    uint32_t live = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) live |= rem[q] ? 1u << (GGS_ID_BITS + q0 + q) : 0u;
    if (L > 0) {
        Rec3 nxt = gather_round(rec, ids, 0, L, lane);
        for (int first = 0; first < L; first += 64) {
            if (remaining == 0) break;
            const Rec3 cur = nxt;
            if (first + 64 < L) nxt = gather_round(rec, ids, first + 64, L, lane);
            const int n = min(64, L - first);
            // The id words of this round are rewritten with the quadrant mask narrowed to the quadrants that actually
            // blended the splat (0 for splats not reached): the backward skips everything else without testing.  The
            // narrowed masks are collected as four 64-bit bit planes (bit j of plane q: entry j blended in quadrant q) on
            // the scalar unit -- updating lane j of a VGPR per entry costs a compare, a move, a select and an or.
            uint64_t plane[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) plane[q] = 0;
            lds.put(cur, lane);
            for (int j = 0; j < n; ++j) {
                const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur.w, j);   // lane j gathered entry j
                const float4 ra = s_rec[j * 3 + 0], rb = s_rec[j * 3 + 1], rc = s_rec[j * 3 + 2];
                const float gx = ra.x, gy = ra.y, cxx = ra.z, cxy = ra.w, cyy = rb.x, op = rb.y;
                const float cr = rb.z, cg = rb.w, cb = rc.x, dep = rc.y;
                uint32_t posv;   // list position + 1, materialised in a VGPR once per splat (not once per quadrant)
                asm volatile("v_mov_b32 %0, %1" : "=v"(posv) : "s"(first + j + 1));
                const uint32_t act = word & live;
                if (CENSUS) ++n_entry;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (!(act & (1u << (GGS_ID_BITS + q0 + q)))) continue;                   // wave-uniform: scalar branch
                    if (CENSUS) ++n_visit;
                    // predicated, branch-free per-pixel update: lane masks instead of nested exec juggling
                    const float dx = gx - pxf[q], dy = gy - pyf[q];
                    const float power = ggs_falloff_log2(cxx, cxy, cyy, dx, dy);
                    const float alpha = __builtin_fminf(GGS_ALPHA_MAX, op * __builtin_amdgcn_exp2f(power));
                    const uint64_t m_ok = __builtin_amdgcn_ballot_w64(power <= 0.f) & __builtin_amdgcn_ballot_w64(alpha >= GGS_ALPHA_MIN);
                    if (m_ok == 0) continue;
                    if (CENSUS) ++n_blend;
                    const float wa = alpha * T[q];
                    const float test_T = T[q] - wa;           // = T (1 - alpha)
                    const uint64_t m_stop = m_ok & __builtin_amdgcn_ballot_w64(test_T < GGS_T_MIN);
                    const uint64_t m_app = m_ok & ~m_stop;
                    if (m_stop != 0) {                        // a pixel finishes at most once: the rare path, kept a BRANCH
                        asm volatile("" ::: "memory");
                        const int n_stop = (int)__popcll(m_stop);
                        rem[q] -= n_stop; remaining -= n_stop;
                        if (rem[q] == 0) live &= ~(1u << (GGS_ID_BITS + q0 + q));
                        pxf[q] = sel(m_stop, inf_v, pxf[q]);  // park the finished pixels (only when there are any)
                    }
                    const float w = sel_or_zero(m_app, wa);
                    C0[q] = fmaf(cr, w, C0[q]);
                    C1[q] = fmaf(cg, w, C1[q]);
                    C2[q] = fmaf(cb, w, C2[q]);
                    D[q] = fmaf(dep, w, D[q]);
                    A[q] += w;
                    T[q] -= w;
                    last[q] = __float_as_uint(sel(m_app, __uint_as_float(posv), __uint_as_float(last[q])));
                    // some pixel of the quadrant passed the alpha test: keep the quadrant in the backward's mask (a
                    // superset of "some pixel blended it" -- the backward re-tests every pixel -- at no VALU cost)
                    plane[q] |= 1ull << j;
                }
                if (remaining == 0) break;
            }
            uint32_t neww = cur.w & GGS_ID_MASK;
#pragma unroll
            for (int q = 0; q < NQ; ++q) neww |= ((uint32_t)(plane[q] >> lane) & 1u) << (GGS_ID_BITS + q);
            if (!CENSUS && lane < n) ids[first + lane] = neww;
        }
    }
    if (CENSUS) {
        if (lane == 0) { atomicAdd(census, (unsigned long long)n_visit); atomicAdd(census + 1, (unsigned long long)n_blend); atomicAdd(census + 2, (unsigned long long)n_entry); }
        return;
    }

    const size_t HW = (size_t)a.H * a.W;
    const float* bg = a.bg + 3 * v;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    float* oc = a.out_color + (size_t)v * 3 * HW;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!inside[q]) continue;
        const size_t pix = (size_t)(py0 + ((q0 + q) / GGS_QX) * 8) * a.W + (px0 + ((q0 + q) % GGS_QX) * 8);
        a.final_T[(size_t)v * HW + pix] = T[q];
        a.n_contrib[(size_t)v * HW + pix] = last[q];
        oc[pix] = fmaf(T[q], bg0, C0[q]);
        oc[HW + pix] = fmaf(T[q], bg1, C1[q]);
        oc[2 * HW + pix] = fmaf(T[q], bg2, C2[q]);
        a.out_depth[(size_t)v * HW + pix] = D[q];
        a.out_alpha[(size_t)v * HW + pix] = A[q];
    }
}


// GgsParams.debug (the reference's pipe.debug) also arms a self-check of the latency mapping: the walks below park the records of
// a round COMPACTED into the low slots of their LDS slice and deliberately read one pair past the last one (read, tested, never
// blended: slots past count + 1 hold whatever an earlier round or an earlier kernel left there).  With the slice filled with NaNs
// in front of every round, anything stale that reaches an output shows up as a NaN -- tests/test_gpu_fullsize.py compares such a
// run bit for bit with the tile-wave mapping.  What keeps stale slots out: the forward blends an entry only where okA /
// okB is set, and a NaN exponent fails `p <= 0`; the filler behind an odd count has opacity 0 (0 x exp2(finite) = 0, where
// fminf(0.99, NaN) would be 0.99); the backward multiplies by alpha = G = 0 selected through `valid`, not by a product with 0.
__device__ __forceinline__ void poison_slots(float4* s_rec, int n, int lane) {
    const float q = __builtin_nanf("");
    for (int i = lane; i < n; i += 64) s_rec[i] = make_float4(q, q, q, q);
    __builtin_amdgcn_wave_barrier();
}

// The walk of the latency mapping takes the entries that reach its quadrant two at a time.  The records of a round are parked
// in the LDS COMPACTED to those entries (slot = rank among them), each with its 1-based list position in the spare field c.z: the
// walk is then a counter over the slots -- on a wave that has its SIMD to itself every instruction costs issue time, the scalar
// ones too, and picking the next two set bits out of a lane mask took ~20 of them per pair.
struct PairRec { float4 a0, a1, b0, b1; float a2x, a2y, b2x, b2y, posA, posB; };
struct PairAlpha { float alA, alB; uint64_t okA, okB; };
__device__ __forceinline__ PairRec load_pair(const float4* s_rec, int slot) {      // slots `slot`, `slot + 1`
    PairRec r;
    const float4* p = s_rec + slot * 3;
    r.a0 = p[0]; r.a1 = p[1];
    r.b0 = p[3]; r.b1 = p[4];
    const float4 a2 = p[2], b2 = p[5];
    r.a2x = a2.x; r.a2y = a2.y; r.posA = a2.z; r.b2x = b2.x; r.b2y = b2.y; r.posB = b2.z;
    return r;
}
// (hi:lo) <- (hi:lo) << 1 | (mask != 0): three scalar instructions, the compare's SCC shifted in through the carry
__device__ __forceinline__ void push_taken(uint32_t& lo, uint32_t& hi, uint64_t mask) {
    asm volatile("s_cmp_lg_u64 %2, 0\n\t"
                 "s_addc_u32 %0, %0, %0\n\t"
                 "s_addc_u32 %1, %1, %1" : "+s"(lo), "+s"(hi) : "s"(mask) : "scc");
}
// both alpha tests (independent of T and of each other)
__device__ __forceinline__ PairAlpha test_pair(const PairRec& r, float pxf, float pyf) {
    PairAlpha p;
    const float dxA = r.a0.x - pxf, dyA = r.a0.y - pyf, dxB = r.b0.x - pxf, dyB = r.b0.y - pyf;
    const float pA = ggs_falloff_log2(r.a0.z, r.a0.w, r.a1.x, dxA, dyA);
    const float pB = ggs_falloff_log2(r.b0.z, r.b0.w, r.b1.x, dxB, dyB);
    p.alA = __builtin_fminf(GGS_ALPHA_MAX, r.a1.y * __builtin_amdgcn_exp2f(pA));
    p.alB = __builtin_fminf(GGS_ALPHA_MAX, r.b1.y * __builtin_amdgcn_exp2f(pB));
    p.okA = __builtin_amdgcn_ballot_w64(pA <= 0.f) & __builtin_amdgcn_ballot_w64(p.alA >= GGS_ALPHA_MIN);
    p.okB = __builtin_amdgcn_ballot_w64(pB <= 0.f) & __builtin_amdgcn_ballot_w64(p.alB >= GGS_ALPHA_MIN);
    return p;
}

// Checkpoint record of one (slot, quadrant): GGS_CKPT_PLANES planes of 64 floats (ggs_common.h GGS_SEG).
__device__ __forceinline__ float* ckpt_record(float* ckpt, size_t slot, int q0, int lane) {
    return ckpt + ((slot * GGS_NQ + (size_t)q0) * GGS_CKPT_PLANES) * 64 + lane;
}
__device__ __forceinline__ void ckpt_store(float* rec, float T, float C0, float C1, float C2) {
    rec[0] = T; rec[64] = C0; rec[128] = C1; rec[192] = C2;
}

// K4b body, latency mapping (one wave per (tile, quadrant), launches too small to fill the chip).  The kernel is as long as its
// dozen longest walks (profiles/r05_wave_timeline.md), and those run on SIMDs they have to themselves.  What a lone wave pays
// for is the NUMBER of instructions it issues -- ~6-9 cycles per VALU instruction whether or not it depends on the one before
// (profiles/r02_valu_issue_rates.md), and the scalar ones are not free either -- not their latencies: hiding the LDS read and
// the alpha test of the next pair behind the blend chain of the current one (software pipeline, round 5) bought 1.6 %, taking
// instructions out of the walk bought the rest (90.4 -> 72 us for one 1080p view of config 2, profiles/r05_quad_forward.md):
//   * the round's records are parked COMPACTED to the entries that reach the quadrant, with their list position in a spare
//     field: a slot counter walks them (picking set bits out of a lane mask was ~20 scalar instructions per pair);
//   * finished pixels are kept out by a scalar mask, exits are plain compare + branch pairs;
//   * entries nobody in the quadrant took lose their bit through ONE vector atomic per round.
// The arithmetic per pixel is the same as in render_fwd_body: bit-identical outputs.
__device__ __forceinline__ void render_fwd_quadwave(const RenderArgs& a) {
    const bool overflow = a.header->overflow != 0;      // see render_fwd_body: overflowed forward = all tiles empty
    const uint32_t item = a.order[blockIdx.x / GGS_NQ];
    const int q0 = (int)(blockIdx.x % GGS_NQ);
    const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T), lane = threadIdx.x;
    const int tx = t % a.gx, ty = t / a.gx;
    const int px = tx * GGS_TILE_W + (lane & 7) + (q0 % GGS_QX) * 8, py = ty * GGS_TILE + (lane >> 3) + (q0 / GGS_QX) * 8;
    const int L = overflow ? 0 : (int)a.tile_count[(size_t)v * a.T + t];
    const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    uint32_t* ids = a.ids + base;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.rec + (size_t)v * a.P);
    const float inf_v = __builtin_inff();
    if (L == 0) {                 // empty tile: the wave of quadrant 0 stores the whole tile, the other three have nothing to do
        if ((a.W & 3) == 0 && tx * GGS_TILE_W + GGS_TILE_W <= a.W && ty * GGS_TILE + GGS_TILE <= a.H) {
            if (q0 == 0) store_empty_tile(a, v, tx * GGS_TILE_W, ty * GGS_TILE, lane);
            return;
        }
    }
    const bool inside = px < a.W && py < a.H;
    const float pxf = inside ? (float)px : inf_v;
    const float pyf = (float)py;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    uint64_t dead = ~__builtin_amdgcn_ballot_w64(inside);          // pixels outside the image, later also the finished ones
    const uint32_t mine = 1u << (GGS_ID_BITS + q0);

    __shared__ float4 s_rec[66 * 3];          // 64 slots + the two a walk reads ahead of its last pair (never used)
    if (L > 0) {
        // id words two rounds ahead, records one round ahead (ggs_render_common.h): a lone wave would otherwise sit out the
        // id load of every round before it can ask for the records (one view: 101 -> 94 us; the tile-wave mapping has other
        // waves to run meanwhile and only pays for the extra register: left as it is)
        Rec3 nxt = gather_round(rec, ids, 0, L, lane);
        uint32_t w_ahead = gather_ids(ids, 64, L, lane);
        for (int first = 0; first < L && dead != ~0ull; first += 64) {
            // segmented backward: the state in front of list position `first`, i.e. of every round but the first (a wave that finished all
            // its pixels earlier never gets here: then no pixel's last contributor lies behind, and nobody reads the record)
            static_assert(GGS_SEG == 64, "one checkpoint in front of every round");
            if (a.ckpt && first)
                ckpt_store(ckpt_record(a.ckpt, (base + (size_t)first) / GGS_SEG, q0, lane), T, C0, C1, C2);
            const Rec3 cur = nxt;
            if (first + 64 < L) { nxt = gather_recs(rec, w_ahead); w_ahead = gather_ids(ids, first + 128, L, lane); }
            const int n = min(64, L - first);
            // entries of the round that reach this quadrant (about half of a tile's list), compacted into the LDS slice
            uint64_t todo = __builtin_amdgcn_ballot_w64((cur.w & mine) != 0);
            if (n < 64) todo &= (1ull << n) - 1ull;
            if (todo == 0) continue;
            const int count = (int)__popcll(todo);
            {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
                __builtin_amdgcn_wave_barrier();
                if (a.poison) poison_slots(s_rec, 66 * 3, lane);       // debug mode: see poison_slots
                if ((todo >> lane) & 1ull) {
                    const float4 c = make_float4(cur.c.x, cur.c.y, __uint_as_float((uint32_t)(first + lane + 1)), 0.f);
                    s_rec[rank * 3 + 0] = cur.a; s_rec[rank * 3 + 1] = cur.b; s_rec[rank * 3 + 2] = c;
                    // an odd count: a filler behind the last entry -- that entry again with opacity 0, so that it fails the alpha test
                    // in every pixel and is blended with weight zero (zero times whatever the slot held before -- a NaN of an
                    // earlier kernel -- would not be zero)
                    if (rank == count - 1 && count < 64) {
                        s_rec[count * 3 + 0] = cur.a; s_rec[count * 3 + 1] = make_float4(cur.b.x, 0.f, cur.b.z, cur.b.w); s_rec[count * 3 + 2] = c;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // SOFTWARE PIPELINE over the pairs of the round: while pair i is blended (a chain through T: multiply, subtract,
            // compare, lane masks, select, subtract -- per entry) the records of pair i + 1 are on their way from the LDS and its
            // two alpha tests (which depend on neither T nor on pair i) run beside the chain.  A pixel that has finished is kept
            // out by the scalar mask `dead`, not through its coordinate, so the tests of pair i + 1 need nothing that pair i
            // produces.  One step: start the loads of the next pair, blend the current one, test the next one.  The two register
            // sets swap roles from step to step (the loop body is two steps), so nothing is copied at the back edge.
            // (Slots past `count` + 1 hold stale records: read, tested, never blended.)
            int slot = 0;
            uint32_t tk_lo = 0, tk_hi = 0;      // shift register: one bit per visited entry, "some pixel took it", newest in bit 0
            // (two plain exits per step instead of one combined condition: on the scalar unit a compare + branch each, where the
            // combined form materialised both conditions as lane masks first)
#define GGS_FWD_STEP(rc, pc, rn, pn)                                                                                           \
            {                                                                                                                  \
                rn = load_pair(s_rec, slot + 2);                                                                               \
                asm volatile("" ::: "memory");   /* the six reads stay whole and up here (else they are split and sunk to their uses) */ \
                /* ---- blend A, then B (in list order) */                                                                    \
                const uint64_t okA = pc.okA & ~dead;                                                                           \
                {                                                                                                              \
                    const float wa = pc.alA * T;                                                                               \
                    const float test_T = T - wa;                                                                               \
                    const uint64_t stop = okA & __builtin_amdgcn_ballot_w64(test_T < GGS_T_MIN);                               \
                    const uint64_t app = okA & ~stop;                                                                          \
                    dead |= stop;                                                                                              \
                    const float w = sel_or_zero(app, wa);                                                                      \
                    C0 = fmaf(rc.a1.z, w, C0); C1 = fmaf(rc.a1.w, w, C1); C2 = fmaf(rc.a2x, w, C2); D = fmaf(rc.a2y, w, D);    \
                    A += w; T -= w;                                                                                            \
                    last = __float_as_uint(sel(app, rc.posA, __uint_as_float(last)));                                          \
                }                                                                                                              \
                const uint64_t okB = pc.okB & ~dead;     /* a pixel that stopped at A no longer takes B */                     \
                {                                                                                                              \
                    const float wa = pc.alB * T;                                                                               \
                    const float test_T = T - wa;                                                                               \
                    const uint64_t stop = okB & __builtin_amdgcn_ballot_w64(test_T < GGS_T_MIN);                               \
                    const uint64_t app = okB & ~stop;                                                                          \
                    dead |= stop;                                                                                              \
                    const float w = sel_or_zero(app, wa);                                                                      \
                    C0 = fmaf(rc.b1.z, w, C0); C1 = fmaf(rc.b1.w, w, C1); C2 = fmaf(rc.b2x, w, C2); D = fmaf(rc.b2y, w, D);    \
                    A += w; T -= w;                                                                                            \
                    last = __float_as_uint(sel(app, rc.posB, __uint_as_float(last)));                                          \
                }                                                                                                              \
                /* ---- the alpha tests of the next pair (independent of everything above) */                                  \
                pn = test_pair(rn, pxf, pyf);                                                                                  \
                /* which of the two entries anybody in this quadrant took (see below the walk) */                              \
                push_taken(tk_lo, tk_hi, okA); push_taken(tk_lo, tk_hi, okB);                                                  \
                slot += 2;                                                                                                     \
                if (slot >= count) break;                                                                                      \
                if (dead == ~0ull) goto round_done;                                                                            \
            }
            PairRec r0 = load_pair(s_rec, 0), r1;
            PairAlpha p0 = test_pair(r0, pxf, pyf), p1;
            for (;;) {
                GGS_FWD_STEP(r0, p0, r1, p1)
                GGS_FWD_STEP(r1, p1, r0, p0)
            }
#undef GGS_FWD_STEP
        round_done:
            // An entry that reached the quadrant and passed the alpha test in none of its live pixels (about one in three: the
            // binning tests the splat's ellipse against the quadrant's box, the pixels sample it) loses this quadrant's bit in its id
            // word, so that the backward skips it here.  ONE vector atomic per round, issued by the lanes that hold such an entry
            // (four waves share the word, each clears its own bit) -- as a scalar branch + a lane-0 atomic per entry this was a
            // fifth of the walk's instructions.  Entries behind the point where the last pixel finished keep their bits: the
            // backward stops at the last contributor.
            {
                // the shift register holds one bit per entry of every pair walked (the filler behind an odd count included), the
                // newest in bit 0: the entry of slot s sits at bit slot - 1 - s
                const int visited = min(slot, count);
                const uint64_t tk = ((uint64_t)tk_hi << 32) | tk_lo;
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
                if (((todo >> lane) & 1ull) && rank < visited && !((tk >> (slot - 1 - rank)) & 1ull)) atomicAnd(&ids[first + lane], ~mine);
            }
            if (dead == ~0ull) break;
        }
    }
    // ... and what the accumulators ended with, once per tile whose list has more than one segment
    if (a.ckpt && L > GGS_SEG) ckpt_store(ckpt_record(a.ckpt, (size_t)a.ckpt_slots + base / GGS_SEG, q0, lane), T, C0, C1, C2);
    if (!inside) return;
    const size_t HW = (size_t)a.H * a.W;
    const float* bg = a.bg + 3 * v;
    float* oc = a.out_color + (size_t)v * 3 * HW;
    const size_t pix = (size_t)py * a.W + px;
    a.final_T[(size_t)v * HW + pix] = T;
    a.n_contrib[(size_t)v * HW + pix] = last;
    oc[pix] = fmaf(T, bg[0], C0);
    oc[HW + pix] = fmaf(T, bg[1], C1);
    oc[2 * HW + pix] = fmaf(T, bg[2], C2);
    a.out_depth[(size_t)v * HW + pix] = D;
    a.out_alpha[(size_t)v * HW + pix] = A;
}


}  // namespace

// K4b: grid V*T work items (x4 for the per-quadrant variant), block 64.
__global__ __launch_bounds__(64) void ggs_k_render_fwd(RenderArgs a) { render_fwd_body<false>(a); }
// out[0] quadrant passes (64 alpha tests each), out[1] passes in which some pixel passed the test, out[2] list entries walked.
// Must run between the binning and the compositing of a forward: the compositing narrows the masks this walk reads.
__global__ __launch_bounds__(64) void ggs_k_count_forward_visits(RenderArgs a, unsigned long long* out) { render_fwd_body<true>(a, out); }
__global__ __launch_bounds__(64) void ggs_k_render_fwd_quad(RenderArgs a) { render_fwd_quadwave(a); }

namespace {

// K5 body.  DA: gradients of the depth / alpha outputs are present.
template <bool DA, int NQ>
__device__ __forceinline__ void render_bwd_body(const RenderBwdArgs& a) {
    if (a.header->overflow) return;          // lists and per-pixel state were never written (captured-step replay)
    // Only the non-empty work items, longest lists first (they sit at order[r * stride], ggs_k_order_tiles): the ~90 % empty
    // tiles of a view have nothing to differentiate, and their waves -- which used to be interleaved with the real ones for
    // the forward's sake -- exit on one scalar compare behind all the work instead of in front of it.
    const NonEmptyItems it = ggs_nonempty_items(a.bucket_count, (uint32_t)a.n_items);
    uint32_t rank = NQ == GGS_NQ ? blockIdx.x : blockIdx.x / GGS_NQ;
    // Latency mapping, SEGMENTED (ggs_common.h GGS_SEG, ggs_seg_item): with the forward's checkpoints at hand the walk of a list is
    // cut into segments of GGS_SEG positions, one wave each; the later segments ride on the blocks of the empty tiles.
    int seg = 0, n_extra = 0;
    if (a.ckpt) {
        const SegItem si = ggs_seg_item(a.bucket_count, it, (uint32_t)a.n_items, rank);
        if (!si.valid) return;
        rank = si.rank; seg = si.seg; n_extra = si.n_extra;
    } else if (rank >= it.n) return;
    const uint32_t item = a.order[(size_t)rank * it.stride];
    const int q0 = NQ == GGS_NQ ? 0 : (int)(blockIdx.x % GGS_NQ);            // NQ = 1: one wave per sub-block
    const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T), lane = threadIdx.x;
    const int L = (int)a.tile_count[(size_t)v * a.T + t];
    const int lo = seg * GGS_SEG;                                            // this wave walks the list positions [lo, hi)
    const int hi = seg == n_extra ? 0x7fffffff : lo + GGS_SEG;
    if (L <= lo) return;
    const int tx = t % a.gx, ty = t / a.gx;
    const int ox = tx * GGS_TILE_W, oy = ty * GGS_TILE;
    const int px0 = ox + (lane & 7), py0 = oy + (lane >> 3);
    const size_t HW = (size_t)a.H * a.W;
    const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    const uint32_t* __restrict__ ids = a.ids + base;
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.rec + (size_t)v * a.P);
    GradRec* acc = a.acc + (size_t)v * a.P;
    const float* bg = a.bg + 3 * v;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    // Per-pixel state of the reverse walk is just (T, B):
    //   T = transmittance in front of the splat being visited,
    //   B = Tf * (bg . dL/dC)  +  sum over the splats BEHIND it of  w_k * s_k,
    //       s_k = c_k . dL/dC (+ depth_k dL/dD + dL/dA),  w_k = alpha_k T_k,
    // because dL/dalpha_j = T_j s_j - B / (1 - alpha_j).  This is the upstream recurrence
    // (accum_rec / last_alpha / last_color + the background term) folded into one scalar:
    // accum_rec_j = (sum_{k>j} c_k w_k) / (T_j (1 - alpha_j)).
    float pxf[NQ], pyf[NQ], T[NQ], B[NQ], dC0[NQ], dC1[NQ], dC2[NQ], dD[NQ], dA[NQ];
    int nc[NQ];
    int maxc = 0;
    uint32_t my_bits = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        my_bits |= 1u << (GGS_ID_BITS + q0 + q);
        const int px = px0 + ((q0 + q) % GGS_QX) * 8, py = py0 + ((q0 + q) / GGS_QX) * 8;
        const bool inside = px < a.W && py < a.H;
        const size_t pix = (size_t)py * a.W + px;
        pxf[q] = (float)px; pyf[q] = (float)py;
        nc[q] = inside ? (int)a.n_contrib[(size_t)v * HW + pix] : 0;
        T[q] = inside ? a.final_T[(size_t)v * HW + pix] : 1.f;
        dC0[q] = dC1[q] = dC2[q] = 0.f; dD[q] = dA[q] = 0.f;
        if (inside) {
            const float* dc = a.dL_dcolor + (size_t)v * 3 * HW;
            dC0[q] = dc[pix]; dC1[q] = dc[HW + pix]; dC2[q] = dc[2 * HW + pix];
            if (DA) {
                if (a.dL_ddepth) dD[q] = a.dL_ddepth[(size_t)v * HW + pix];
                if (a.dL_dalpha) dA[q] = a.dL_dalpha[(size_t)v * HW + pix];
            }
        }
        B[q] = T[q] * (bg0 * dC0[q] + bg1 * dC1[q] + bg2 * dC2[q]);
        if (a.ckpt) {
            // A pixel whose last contributor lies BEHIND this segment starts from the forward's checkpoint at `hi`: the
            // transmittance in front of entry hi, and B = T_final (bg . dL/dC) + (what the forward accumulated from hi on) . dL/dC.
            // One whose last contributor lies inside starts as the unsegmented walk does; one that ends in front has nothing here.
            if (nc[q] > hi) {
                const float* ck = ckpt_record(const_cast<float*>(a.ckpt), (base + (size_t)hi) / GGS_SEG, q0 + q, lane);
                const float* fin = ckpt_record(const_cast<float*>(a.ckpt), (size_t)a.ckpt_slots + base / GGS_SEG, q0 + q, lane);
                static_assert(GGS_CKPT_PLANES == 4, "T, C0, C1, C2: the depth / alpha backward is never segmented (ggs_backward)");
                T[q] = ck[0];
                B[q] += (fin[64] - ck[64]) * dC0[q] + (fin[128] - ck[128]) * dC1[q] + (fin[192] - ck[192]) * dC2[q];
                nc[q] = hi;
            }
            if (nc[q] <= lo) nc[q] = 0;
        }
        maxc = max(maxc, nc[q]);
    }
    // GradRec field this lane adds to after the reduction (lds_transpose_reduce leaves total v in lane 8 v, the ninth in lane 63)
    const int fld = lds_reduce_field<DA>(lane);
    // wave max of the per-pixel contributor counts
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) maxc = max(maxc, __shfl_xor(maxc, d));
    // tell the compiler the maximum is wave-uniform: otherwise the walk below becomes a per-lane loop (VGPR counter,
    // exec-mask bookkeeping: 4 VALU per list entry, skipped ones included)
    maxc = __builtin_amdgcn_readfirstlane(maxc);
    if (maxc == 0) return;

    __shared__ float4 s_rec[64 * 3];
    RoundLds lds{s_rec};
    __shared__ float s_red[8 * GGS_RED_STRIDE];
    __shared__ float s_red2[NQ == 1 ? 8 * GGS_RED_STRIDE : 1];          // second plane: the pair reductions of the latency mapping
    // The nine per-entry sums start from zero: nine v_mov per list entry on the pipe that bounds this kernel.  Three broadcast
    // LDS reads of a zeroed slot deliver the same zeros on the LDS pipe, right behind the record reads the entry waits for anyway.
    __shared__ float4 s_zero[2];
    if (lane < 8) reinterpret_cast<float*>(s_zero)[lane] = 0.f;
    __builtin_amdgcn_wave_barrier();
    const int fld_bytes = fld * 4;
    // rounds of 64 list positions, walked from the back: round r covers [64 r, 64 r + 64); a segment ends at round lo / 64
    int r = (maxc - 1) >> 6;
    const int r_lo = lo >> 6;
    Rec3 nxt = gather_round(rec, ids, r * 64, L, lane);
    // Latency mapping: id words two rounds ahead, records one round ahead, as in render_fwd_quadwave -- a lone wave otherwise sits
    // out the id load of every round before it can even ask for the records (the tile waves have other waves to run meanwhile).
    uint32_t w_ahead = 0;
    if (NQ == 1 && r > r_lo) w_ahead = gather_ids(ids, (r - 1) * 64, L, lane);
    for (; r >= r_lo; --r) {
        const Rec3 cur = nxt;
        if (NQ == 1) {
            if (r > r_lo) { nxt = gather_recs(rec, w_ahead); if (r > r_lo + 1) w_ahead = gather_ids(ids, (r - 2) * 64, L, lane); }
        } else if (r > r_lo) nxt = gather_round(rec, ids, (r - 1) * 64, L, lane);
        const int first = r * 64;
        const int n = min(64, maxc - first);
        // Entries of the round the forward blended somewhere in this wave's pixels, as a lane mask (lane i holds entry i):
        // an entry that is not ours (21 % of a tile's list, 63 % for a per-quadrant wave) costs nothing -- not even the LDS
        // latency of reading its id word.
        uint64_t todo = __builtin_amdgcn_ballot_w64((cur.w & my_bits) != 0);
        if (n < 64) todo &= (1ull << n) - 1ull;
        if constexpr (NQ == 1) {
            // Latency mapping: two entries per iteration.  Their falloff / alpha evaluations, colour dot products and
            // the two gradient reductions are independent chains the scheduler interleaves; only the (T, B)
            // recurrence is sequential (A = the entry further back, then B).
            // The records are parked in the LDS COMPACTED to our entries, BACK TO FRONT (slot 0 = the last one), each with its
            // list position and Gaussian id in the spare fields c.z / c.w, so that the walk is a slot counter: picking the two
            // highest set bits out of the lane mask and reading their id words across lanes was ~15 scalar + 2 VALU instructions
            // per pair on a wave that pays issue time for every one of them (as in render_fwd_quadwave).
            const int count = (int)__popcll(todo);
            if (count == 0) continue;
            {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
                __builtin_amdgcn_wave_barrier();
                if (a.poison) poison_slots(s_rec, 64 * 3, lane);       // debug mode: see poison_slots
                if ((todo >> lane) & 1ull) {
                    const int sl = count - 1 - rank;
                    const float4 c = make_float4(cur.c.x, cur.c.y, __uint_as_float((uint32_t)(first + lane)),
                                                 __uint_as_float(cur.w & GGS_ID_MASK));
                    s_rec[sl * 3 + 0] = cur.a; s_rec[sl * 3 + 1] = cur.b; s_rec[sl * 3 + 2] = c;
                    // an odd count: a filler behind the last slot (finite values; its B half is never reduced)
                    if (rank == 0 && count < 64) { s_rec[count * 3 + 0] = cur.a; s_rec[count * 3 + 1] = cur.b; s_rec[count * 3 + 2] = c; }
                }
                __builtin_amdgcn_wave_barrier();
            }
            auto reduce_add = [&](uint32_t gid, float v_mx, float v_my, float v_cx, float v_cy, float v_cz, float v_op,
                                  float v_r, float v_g, float v_b, float v_dep) {
                const float S = lds_transpose_reduce<DA>(s_red, lane, v_mx, v_my, v_cx, v_cy, v_cz, v_op, v_r, v_g, v_b, v_dep);
                float* dst = reinterpret_cast<float*>(acc + gid);
                if (fld >= 0) atomicAdd(dst + fld, S);
            };
            for (int slot = 0; slot < count; slot += 2) {
                const bool hasB = slot + 1 < count;
                const float4* p = s_rec + slot * 3;
                const float4 a0 = p[0], a1 = p[1], a2 = p[2];
                const float4 b0 = p[3], b1 = p[4], b2 = p[5];
                const int posA = (int)__float_as_uint(a2.z), posB = (int)__float_as_uint(b2.z);
                // independent of (T, B)
                const float dxA = a0.x - pxf[0], dyA = a0.y - pyf[0], dxB = b0.x - pxf[0], dyB = b0.y - pyf[0];
                const float pA = ggs_falloff_log2(a0.z, a0.w, a1.x, dxA, dyA);
                const float pB = ggs_falloff_log2(b0.z, b0.w, b1.x, dxB, dyB);
                const float GrA = __builtin_amdgcn_exp2f(pA), GrB = __builtin_amdgcn_exp2f(pB);
                const float arA = __builtin_fminf(GGS_ALPHA_MAX, a1.y * GrA), arB = __builtin_fminf(GGS_ALPHA_MAX, b1.y * GrB);
                const bool validA = (posA < nc[0]) & (pA <= 0.f) & (arA >= GGS_ALPHA_MIN);
                const bool validB = hasB & (posB < nc[0]) & (pB <= 0.f) & (arB >= GGS_ALPHA_MIN);
                float sdotA = fmaf(a2.x, dC2[0], fmaf(a1.w, dC1[0], a1.z * dC0[0]));
                float sdotB = fmaf(b2.x, dC2[0], fmaf(b1.w, dC1[0], b1.z * dC0[0]));
                if (DA) { sdotA += fmaf(a2.y, dD[0], dA[0]); sdotB += fmaf(b2.y, dD[0], dA[0]); }
                const float alA = validA ? arA : 0.f, GA = validA ? GrA : 0.f;
                const float alB = validB ? arB : 0.f, GB = validB ? GrB : 0.f;
                const float raA = __builtin_amdgcn_rcpf(1.f - alA), raB = __builtin_amdgcn_rcpf(1.f - alB);
                // the recurrence: A, then B
                T[0] *= raA;
                const float wA = alA * T[0];
                const float dLA = fmaf(T[0], sdotA, -B[0] * raA);
                B[0] = fmaf(wA, sdotA, B[0]);
                T[0] *= raB;
                const float wB = alB * T[0];
                const float dLB = fmaf(T[0], sdotB, -B[0] * raB);
                B[0] = fmaf(wB, sdotB, B[0]);
                // per-splat sums over this wave's pixels
                const float tA = GA * dLA, tB = GB * dLB;
                const float hxA = tA * dxA, hyA = tA * dyA, hxB = tB * dxB, hyB = tB * dyB;
                const uint32_t gidA = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(a2.w));
                if (hasB) {         // both reductions with their LDS round trips in flight together, one wait
                    const uint32_t gidB = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(b2.w));
                    const float xa[10] = {hxA, hyA, hxA * dxA, hxA * dyA, hyA * dyA, tA, wA * dC0[0], wA * dC1[0], wA * dC2[0],
                                          DA ? wA * dD[0] : 0.f};
                    const float xb[10] = {hxB, hyB, hxB * dxB, hxB * dyB, hyB * dyB, tB, wB * dC0[0], wB * dC1[0], wB * dC2[0],
                                          DA ? wB * dD[0] : 0.f};
                    float SA, SB;
                    lds_transpose_reduce2<DA>(s_red, s_red2, lane, xa, xb, SA, SB);
                    if (fld >= 0) {
                        atomicAdd(reinterpret_cast<float*>(acc + gidA) + fld, SA);
                        atomicAdd(reinterpret_cast<float*>(acc + gidB) + fld, SB);
                    }
                } else
                    reduce_add(gidA, hxA, hyA, hxA * dxA, hxA * dyA, hyA * dyA, tA, wA * dC0[0], wA * dC1[0], wA * dC2[0],
                               DA ? wA * dD[0] : 0.f);
            }
            continue;
        }
        lds.put(cur, lane);
        // Per-tile waves keep the plain descending loop: they are throughput bound and the mask bookkeeping costs
        // what the skipped entries save.
        for (int j = n - 1; j >= 0; --j) {
            const int pos = first + j;                  // list position; pixel q blended it iff pos < nc[q]
            const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur.w, j);   // lane j gathered entry j
            if (!(word & my_bits)) continue;            // the forward blended this splat nowhere in this tile
            const float4 ra = s_rec[j * 3 + 0], rb = s_rec[j * 3 + 1], rc = s_rec[j * 3 + 2];
            const float gx = ra.x, gy = ra.y, cxx = ra.z, cxy = ra.w, cyy = rb.x, op = rb.y;
            const float cr = rb.z, cg = rb.w, cb = rc.x, dep = rc.y;
            typedef float v4f __attribute__((ext_vector_type(4)));
            typedef volatile __attribute__((address_space(3))) v4f* lds_v4f;
            typedef volatile __attribute__((address_space(3))) float* lds_f;
            const v4f z0 = ((lds_v4f)s_zero)[0], z1 = ((lds_v4f)s_zero)[1];
            const float z2 = ((lds_f)s_zero)[0];
            float v_mx = z0.x, v_my = z0.y, v_cx = z0.z, v_cy = z0.w, v_cz = z1.x, v_op = z1.y;
            float v_r = z1.z, v_g = z1.w, v_b = z2, v_dep = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (!(word & (1u << (GGS_ID_BITS + q0 + q)))) continue;   // quadrant did not blend it (forward's mask)
                const float dx = gx - pxf[q], dy = gy - pyf[q];
                const float power = ggs_falloff_log2(cxx, cxy, cyy, dx, dy);
                const float Gr = __builtin_amdgcn_exp2f(power);
                const float ar = __builtin_fminf(GGS_ALPHA_MAX, op * Gr);
                const bool valid = (pos < nc[q]) & (power <= 0.f) & (ar >= GGS_ALPHA_MIN);
                // Lanes that did not blend this splat sit out (EXEC mask): their (T, B) and sums keep their values, and a
                // quadrant nobody blended skips the block on the scalar branch.  (The select form -- alpha = G = 0 in those
                // lanes -- costs two v_cndmask per quadrant: -2.3 % kernel time, tools/dbg/job.sh ab.)
                // Geometry terms are accumulated as RAW moments of h = G dL/dG over the pixels
                //   (sum h, sum h dx, sum h dy, sum h dx^2, sum h dx dy, sum h dy^2);
                // the per-splat constants (conic, opacity, -1/2) are applied once per Gaussian in the
                // per-Gaussian backward instead of once per pixel here (straight through the 0.99 clamp).
                if (valid) {
                    const float ra = __builtin_amdgcn_rcpf(1.f - ar);
                    T[q] *= ra;
                    const float w = ar * T[q];
                    float sdot = fmaf(cb, dC2[q], fmaf(cg, dC1[q], cr * dC0[q]));
                    if (DA) sdot += fmaf(dep, dD[q], dA[q]);
                    const float dL_da = fmaf(T[q], sdot, -B[q] * ra);
                    B[q] = fmaf(w, sdot, B[q]);
                    v_r = fmaf(w, dC0[q], v_r); v_g = fmaf(w, dC1[q], v_g); v_b = fmaf(w, dC2[q], v_b);
                    if (DA) v_dep = fmaf(w, dD[q], v_dep);
                    const float t = Gr * dL_da;
                    v_op += t;
                    const float hx = t * dx, hy = t * dy;
                    v_mx += hx; v_my += hy;
                    v_cx = fmaf(hx, dx, v_cx);
                    v_cy = fmaf(hx, dy, v_cy);
                    v_cz = fmaf(hy, dy, v_cz);
                }
            }
            const float S = lds_transpose_reduce<DA>(s_red, lane, v_mx, v_my, v_cx, v_cy, v_cz, v_op, v_r, v_g, v_b, v_dep);
            const uint32_t gid = word & GGS_ID_MASK;
            // record address on the scalar unit (the id word is wave-uniform), field offset in a loop-invariant VGPR: the
            // compiler's form is a 64-bit v_mad_u64_u32 per entry
            const uint64_t rec_addr = reinterpret_cast<uint64_t>(acc) + (uint64_t)gid * sizeof(GradRec);
            // one instruction, 9 (10) lanes, all addresses distinct
            if (fld >= 0) asm volatile("global_atomic_add_f32 %0, %1, %2" :: "v"(fld_bytes), "v"(S), "s"(rec_addr) : "memory");
        }
    }
}

}  // namespace

// K5: grid V*T work items (x4 for the per-quadrant variant), block 64.  Separate entry points so the common case
// (no loss on depth / alpha: s2_registration.py:258-267, s3_appearance.py:131-140) carries no dead work.
__global__ __launch_bounds__(64) void ggs_k_render_bwd(RenderBwdArgs a) { render_bwd_body<false, GGS_NQ>(a); }
__global__ __launch_bounds__(64) void ggs_k_render_bwd_da(RenderBwdArgs a) { render_bwd_body<true, GGS_NQ>(a); }
// Latency mapping of the backward (launches too small to fill the chip).  Without depth / alpha gradients: ggs_k_render_bwd with the
// forward's checkpoints (a.ckpt) -- one tile wave per SEGMENT of 64 list positions.  With them (no loss of the reference): the
// unsegmented per-quadrant walk, two entries per iteration.
__global__ __launch_bounds__(64) void ggs_k_render_bwd_da_quad(RenderBwdArgs a) { render_bwd_body<true, 1>(a); }

// Introspection (bench.py's compute-side roofline): the number of (splat, pixel) pairs the forward BLENDED, i.e. the pairs
// the backward differentiates -- list position below the pixel's last contributor, falloff exponent <= 0, alpha >= 1/255 --
// counted with the backward's own tests over the forward's lists.  One wave per work item; not on any timed path.
// n_out >= 4 (ggs_count_pairs): out[1] quadrant passes of the backward (64 pixels evaluated each), out[2] entries it reduces (some
// quadrant bit set), out[3] list entries it walks.
__global__ __launch_bounds__(64) void ggs_k_count_blends(RenderBwdArgs a, unsigned long long* out, int n_out) {
    if (a.header->overflow) return;
    const uint32_t item = blockIdx.x;
    const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T), lane = threadIdx.x;
    const int L = (int)a.tile_count[(size_t)v * a.T + t];
    if (L == 0) return;
    const int tx = t % a.gx, ty = t / a.gx;
    const int px0 = tx * GGS_TILE_W + (lane & 7), py0 = ty * GGS_TILE + (lane >> 3);
    const size_t HW = (size_t)a.H * a.W;
    const uint32_t* __restrict__ ids = a.ids + (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(a.rec + (size_t)v * a.P);
    float pxf[GGS_NQ], pyf[GGS_NQ];
    int nc[GGS_NQ], maxc = 0;
#pragma unroll
    for (int q = 0; q < GGS_NQ; ++q) {
        const int px = px0 + (q % GGS_QX) * 8, py = py0 + (q / GGS_QX) * 8;
        pxf[q] = (float)px; pyf[q] = (float)py;
        nc[q] = px < a.W && py < a.H ? (int)a.n_contrib[(size_t)v * HW + (size_t)py * a.W + px] : 0;
        maxc = max(maxc, nc[q]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) maxc = max(maxc, __shfl_xor(maxc, d));
    maxc = __builtin_amdgcn_readfirstlane(maxc);
    __shared__ float4 s_rec[64 * 3];
    RoundLds lds{s_rec};
    unsigned long long n = 0;
    unsigned n_pass = 0, n_red = 0;
    for (int first = 0; first < maxc; first += 64) {
        const Rec3 cur = gather_round(rec, ids, first, L, lane);
        lds.put(cur, lane);
        const int cnt = min(64, maxc - first);
        for (int j = 0; j < cnt; ++j) {
            const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)cur.w, j);
            const float4 ra = s_rec[j * 3 + 0], rb = s_rec[j * 3 + 1];
            if (word & ~GGS_ID_MASK) ++n_red;
#pragma unroll
            for (int q = 0; q < GGS_NQ; ++q) {
                if (!(word & (1u << (GGS_ID_BITS + q)))) continue;
                ++n_pass;
                const float dx = ra.x - pxf[q], dy = ra.y - pyf[q];
                const float power = ggs_falloff_log2(ra.z, ra.w, rb.x, dx, dy);
                const float ar = __builtin_fminf(GGS_ALPHA_MAX, rb.y * __builtin_amdgcn_exp2f(power));
                n += __popcll(__builtin_amdgcn_ballot_w64((first + j < nc[q]) & (power <= 0.f) & (ar >= GGS_ALPHA_MIN)));
            }
        }
    }
    if (lane == 0 && n) atomicAdd(out, n);
    if (lane == 0 && n_out >= 4) {
        atomicAdd(out + 1, (unsigned long long)n_pass); atomicAdd(out + 2, (unsigned long long)n_red);
        atomicAdd(out + 3, (unsigned long long)maxc);
    }
}

