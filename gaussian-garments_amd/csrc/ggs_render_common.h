// ggs_render_common.h -- wave-level helpers of the compositing kernels (ggs_render.hip: one wave per tile /
// per (tile, quadrant)).
#pragma once
#include "ggs_kernels.h"

namespace {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_fetch(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// ---- wave reduction of the 10 per-splat gradient values ----------------------------------------
// gfx950 has v_permlane32_swap / v_permlane16_swap: exchanging halves (rows) between TWO registers
// and adding folds two values at once, so the 64-lane sums of 10 values cost 28 VALU ops instead of
// 10 x 8 with one DPP chain per value (21 ops without depth/alpha gradients):
//   swap32_add(x, y)   -> lanes 0-31: 32 partials of x        | lanes 32-63: 32 partials of y
//   swap16_add(z1, z2) -> rows 0..3 (16 lanes each): partials of (z1.lo, z2.lo, z1.hi, z2.hi)
//   fold_rows          -> the three row-partial registers folded into one, each total in one lane
__device__ __forceinline__ float swap32_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float z1, float z2) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(z1), __float_as_uint(z2), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Rows -> quads.  q1 and q2 hold one value per 16-lane row, r5 one value per row (or per pair of rows); DPP adds
// with a bank mask write only part of a row, so two registers fold into one per step instead of each being
// reduced on its own:
//   row_ror:8  : lanes 0-7 of a row <- q1 pair sums, lanes 8-15 <- q2 pair sums;  r5 += ror8(r5)
//   row_ror:4/12: lanes 0-3 / 8-11 <- q1 / q2 sums of 4,  lanes 4-7 / 12-15 <- r5 sums of 4
//   quad_perm  : two more adds leave every quad with its total.
// Result, per row: quad 0 = q1's row total, quad 2 = q2's row total; r5's totals: see the end of the function.
// (s_nop: 2 wait states between a VALU write and a DPP read of the same VGPR; the assembler does not add them.)
template <int R5_ROWS>
__device__ __forceinline__ float fold_rows(float q1, float q2, float r5) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa"
        : "+v"(q1), "+v"(r5) : "v"(q2));
    q1 += dpp_fetch<0x4E, 0xf>(q1);   // quad_perm:[2,3,0,1]
    q1 += dpp_fetch<0xB1, 0xf>(q1);   // quad_perm:[1,0,3,2]
    // r5's row totals (quads 1 and 3) are folded across rows so that every value ends in exactly ONE lane: several
    // lanes of one atomic instruction hitting the same address serialise in the L2 (measured: +50 % kernel time).
    if (R5_ROWS == 2) {      // r5 = (b, b, depth, depth) by rows -> quad 1 of row 1 = b, of row 3 = depth
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0x2" : "+v"(q1));
    } else {                 // r5 = b in all four rows -> quad 1 of row 3 = b
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xa\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0x8 bank_mask:0x2" : "+v"(q1));
    }
    return q1;
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
