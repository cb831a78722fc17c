// ggs_render_common.h -- wave-level helpers of the compositing kernels (ggs_render.hip: one wave per tile /
// per (tile, quadrant)).
#pragma once
#include "ggs_kernels.h"

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// gfx950's v_permlane32_swap exchanges the wave halves between TWO registers: one add then folds two values at once
//   swap32_add(x, y)   -> lanes 0-31: 32 partials of x        | lanes 32-63: 32 partials of y
// (the round-1-3 butterfly built on it and on v_permlane16_swap: tools/dbg/variants/r04_bwd_reductions_and_whatifs.patch)
__device__ __forceinline__ float swap32_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// log2 of the Gaussian falloff at offset (dx, dy) from the splat's centre, conic pre-scaled by the preprocess pass
// (cxx, cxy, cyy) = (-log2e / 2, -log2e, -log2e / 2) x conic.  ONE definition: the forward, the backward and ggs_count_blends
// must agree to the bit on which pixels pass the alpha test.  dx (cxx dx + cxy dy) + (cyy dy) dy as three products and two fused
// multiply-adds (the six-operation form of rounds 1-3: tools/dbg/variants/r06_falloff_six_ops.patch; backward +0.6 us per view).
__device__ __forceinline__ float ggs_falloff_log2(float cxx, float cxy, float cyy, float dx, float dy) {
    return fmaf(dx, fmaf(cxx, dx, cxy * dy), (cyy * dy) * dy);
}

// ---- wave reduction of the 9 (10) per-splat gradient sums through an LDS transpose ------------------------
// A permlane-swap / masked-DPP butterfly spends ~105 VALU-issue cycles per list entry, most of them in v_permlane32/16_swap
// (8.3 cycles each).  Here eight of the values cross the lanes through a wave-private LDS plane instead: every lane stores its 8 partials
// (plane [value][80] floats: bank = lane mod 32, conflict-free), then lane 8 v + s reads back the 8 partials
// {2 s, 2 s + 1} + 16 k of value v (4 x ds_read_b64: bank = (16 v + 2 s) mod 64 within each half-wave, conflict-free), adds
// them (7 v_add_f32) and three DPP adds inside its 8-lane group finish the sum: 10 VALU instructions for 8 values, the LDS pipe
// issues beside the VALU.  The ninth value (and the tenth with depth / alpha gradients) stays on a DPP chain that ends in
// lanes the eight groups do not need: every total sits in its own lane of ONE register, as before (one vector atomic).
//   result lanes: 8 v -> value v (v = 0..7);  !DA: lane 63 -> value 8;  DA: lane 31 -> value 8, lane 63 -> value 9
// Measured (profiles/r04_bwd_variants.md): ggs_k_render_bwd 42.2 -> 38.7 us per view; with the DS instructions the compiler
// picks from plain C++ (ds_write2_b32 pairs, ds_read2_b64) 39.7; a what-if build with NO reduction at all: 31.0.
#define GGS_RED_STRIDE 80
template <bool DA>
__device__ __forceinline__ float lds_transpose_reduce(float* s_red, int lane, float v0, float v1, float v2, float v3,
                                                      float v4, float v5, float v6, float v7, float v8, float v9) {
    // The LDS side is written out: ds_write_addtid_b32 (address = M0 + offset + 4 lane, no address VGPR: 2 cycles of the
    // VGPR -> LDS path per instruction against 4 for ds_write_b32 / 6 for the ds_write2_b32 pairs the compiler forms) and four
    // plain ds_read_b64 (2 LDS cycles each; the compiler merges them into two ds_read2_b64 of 8).  The waits are explicit:
    // the compiler's own lgkmcnt bookkeeping stays conservative with extra DS operations in flight (they return in order).
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f a, b, c, d;
    const unsigned base = (unsigned)(uintptr_t)s_red;
    const unsigned rd = base + ((lane >> 3) * GGS_RED_STRIDE + (lane & 7) * 2) * 4;
    asm volatile("s_mov_b32 m0, %8\n\t"
                 "s_nop 0\n\t"
                 "ds_write_addtid_b32 %0\n\t"
                 "ds_write_addtid_b32 %1 offset:320\n\t"
                 "ds_write_addtid_b32 %2 offset:640\n\t"
                 "ds_write_addtid_b32 %3 offset:960\n\t"
                 "ds_write_addtid_b32 %4 offset:1280\n\t"
                 "ds_write_addtid_b32 %5 offset:1600\n\t"
                 "ds_write_addtid_b32 %6 offset:1920\n\t"
                 "ds_write_addtid_b32 %7 offset:2240"
                 :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "s"(base) : "memory", "m0");
    asm volatile("ds_read_b64 %0, %4\n\t"
                 "ds_read_b64 %1, %4 offset:64\n\t"
                 "ds_read_b64 %2, %4 offset:128\n\t"
                 "ds_read_b64 %3, %4 offset:192"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(rd) : "memory");
    // the ninth / tenth value: in-row DPP sums while the LDS reads are in flight
    float t = DA ? swap32_add(v8, v9) : v8;           // DA: lanes 0-31 partials of v8, lanes 32-63 of v9
    t += dpp_fetch<0xB1, 0xf>(t);                     // quad_perm:[1,0,3,2]
    t += dpp_fetch<0x4E, 0xf>(t);                     // quad_perm:[2,3,0,1]
    t += dpp_fetch<0x141, 0xf>(t);                    // row_half_mirror
    t += dpp_fetch<0x140, 0xf>(t);                    // row_mirror: every lane holds its row total
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    float s = ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
    s += dpp_fetch<0xB1, 0xf>(s);
    s += dpp_fetch<0x4E, 0xf>(s);
    s += dpp_fetch<0x141, 0xf>(s);                    // every lane of group v holds the total of value v
    if (DA) {        // rows 1 / 3 <- rows 0 + 1 / 2 + 3, then lanes 28-31 / 60-63 of s take them
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_mov_b32_dpp %1, %0 quad_perm:[0,1,2,3] row_mask:0xa bank_mask:0x8" : "+v"(t), "+v"(s));
    } else {         // row 1 / 3 <- + row 0 / 2, then lanes 60-63 of s <- lane 31 + own = the wave total
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %1, %0, %0 row_bcast:31 row_mask:0x8 bank_mask:0x8" : "+v"(t), "+v"(s));
    }
    return s;
}
// Two reductions at once (the two entries of a pair in the latency mapping): both sets of partials go out to two LDS planes, both
// read-backs are in flight together and the wave waits ONCE -- a wave that has its SIMD to itself sits through the LDS round trip
// of every reduction it does one after the other.  Same arithmetic per value as lds_transpose_reduce.
template <bool DA>
__device__ __forceinline__ void lds_transpose_reduce2(float* s_redA, float* s_redB, int lane, const float (&x)[10], const float (&y)[10],
                                                      float& SA, float& SB) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f a, b, c, d, e, f, g, h;
    const unsigned baseA = (unsigned)(uintptr_t)s_redA, baseB = (unsigned)(uintptr_t)s_redB;
    const unsigned off = ((lane >> 3) * GGS_RED_STRIDE + (lane & 7) * 2) * 4;
    const unsigned rdA = baseA + off, rdB = baseB + off;
#define GGS_RED_WRITES                                                                                             \
    "s_mov_b32 m0, %8\n\t"                                                                                         \
    "s_nop 0\n\t"                                                                                                  \
    "ds_write_addtid_b32 %0\n\t"                                                                                   \
    "ds_write_addtid_b32 %1 offset:320\n\t"                                                                        \
    "ds_write_addtid_b32 %2 offset:640\n\t"                                                                        \
    "ds_write_addtid_b32 %3 offset:960\n\t"                                                                        \
    "ds_write_addtid_b32 %4 offset:1280\n\t"                                                                       \
    "ds_write_addtid_b32 %5 offset:1600\n\t"                                                                       \
    "ds_write_addtid_b32 %6 offset:1920\n\t"                                                                       \
    "ds_write_addtid_b32 %7 offset:2240"
    asm volatile(GGS_RED_WRITES :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "s"(baseA)
                 : "memory", "m0");
    asm volatile(GGS_RED_WRITES :: "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "s"(baseB)
                 : "memory", "m0");
#undef GGS_RED_WRITES
    asm volatile("ds_read_b64 %0, %8\n\t"
                 "ds_read_b64 %1, %8 offset:64\n\t"
                 "ds_read_b64 %2, %8 offset:128\n\t"
                 "ds_read_b64 %3, %8 offset:192\n\t"
                 "ds_read_b64 %4, %9\n\t"
                 "ds_read_b64 %5, %9 offset:64\n\t"
                 "ds_read_b64 %6, %9 offset:128\n\t"
                 "ds_read_b64 %7, %9 offset:192"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h) : "v"(rdA), "v"(rdB) : "memory");
    // the ninth / tenth values of both: in-row DPP sums while the LDS reads are in flight
    float t = DA ? swap32_add(x[8], x[9]) : x[8];
    float u = DA ? swap32_add(y[8], y[9]) : y[8];
    t += dpp_fetch<0xB1, 0xf>(t); u += dpp_fetch<0xB1, 0xf>(u);
    t += dpp_fetch<0x4E, 0xf>(t); u += dpp_fetch<0x4E, 0xf>(u);
    t += dpp_fetch<0x141, 0xf>(t); u += dpp_fetch<0x141, 0xf>(u);
    t += dpp_fetch<0x140, 0xf>(t); u += dpp_fetch<0x140, 0xf>(u);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    float s = ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y));
    float r = ((e.x + e.y) + (f.x + f.y)) + ((g.x + g.y) + (h.x + h.y));
    s += dpp_fetch<0xB1, 0xf>(s); r += dpp_fetch<0xB1, 0xf>(r);
    s += dpp_fetch<0x4E, 0xf>(s); r += dpp_fetch<0x4E, 0xf>(r);
    s += dpp_fetch<0x141, 0xf>(s); r += dpp_fetch<0x141, 0xf>(r);
    if (DA) {
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_mov_b32_dpp %1, %0 quad_perm:[0,1,2,3] row_mask:0xa bank_mask:0x8\n\t"
                     "v_mov_b32_dpp %3, %2 quad_perm:[0,1,2,3] row_mask:0xa bank_mask:0x8" : "+v"(t), "+v"(s), "+v"(u), "+v"(r));
    } else {
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %1, %0, %0 row_bcast:31 row_mask:0x8 bank_mask:0x8\n\t"
                     "v_add_f32_dpp %3, %2, %2 row_bcast:31 row_mask:0x8 bank_mask:0x8" : "+v"(t), "+v"(s), "+v"(u), "+v"(r));
    }
    SA = s; SB = r;
}
// GradRec field the lane adds to after lds_transpose_reduce (-1: none)
template <bool DA>
__device__ __forceinline__ int lds_reduce_field(int lane) {
    if ((lane & 7) == 0) return lane >> 3;
    if (lane == 63) return DA ? 9 : 8;
    if (DA && lane == 31) return 8;
    return -1;
}

// Lane selects driven by explicit 64-bit lane masks.  The forward keeps its predicates (alpha test, stop, blend) as
// SGPR masks so that their population counts run on the scalar unit; going through `bool` the compiler rebuilds a
// mask from a 0/1 VGPR (v_cndmask + v_cmp) every time a ballot of a combined predicate is needed.
__device__ __forceinline__ float sel_or_zero(uint64_t m, float a) {            // m ? a : 0
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ float sel(uint64_t m, float a, float b) {           // m ? a : b
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

struct Rec3 { float4 a, b, c; uint32_t w; };   // record of splat (first + lane) and its id word

// id word of list position (first + lane), clamped into the list
__device__ __forceinline__ uint32_t gather_ids(const uint32_t* __restrict__ ids, int first, int L, int lane) {
    int i = first + lane;
    i = i < L ? i : L - 1;
    i = i < 0 ? 0 : i;
    return ids[i];
}
__device__ __forceinline__ Rec3 gather_recs(const float4* __restrict__ rec, uint32_t w) {
    Rec3 o;
    o.w = w;
    const float4* r = rec + (size_t)(w & GGS_ID_MASK) * 3;
    o.a = r[0]; o.b = r[1]; o.c = r[2];
    return o;
}
__device__ __forceinline__ Rec3 gather_round(const float4* __restrict__ rec, const uint32_t* __restrict__ ids,
                                             int first, int L, int lane) {
    return gather_recs(rec, gather_ids(ids, first, L, lane));
}
// The gather of a round is two dependent loads (id word -> record).  The latency-mapped forward keeps the id words TWO
// rounds ahead and the records one round ahead.

// The 64 records of a round are parked in a wave-private LDS slice and every splat is read back with
// wave-uniform (broadcast) ds_read_b128: the LDS pipe issues beside the VALU, where 11 v_readlane per splat
// would take VALU issue slots.  One wave per workgroup: no barrier, only the compiler fence.
struct RoundLds {
    float4* rec;          // [64][3]
    __device__ __forceinline__ void put(const Rec3& r, int lane) {
        __builtin_amdgcn_wave_barrier();
        rec[lane * 3 + 0] = r.a; rec[lane * 3 + 1] = r.b; rec[lane * 3 + 2] = r.c;
        __builtin_amdgcn_wave_barrier();
    }
};

}  // namespace
