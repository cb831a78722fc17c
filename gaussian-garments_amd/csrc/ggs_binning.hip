// ggs_binning.hip -- per-view tile-histogram scan and per-tile key sort.
//
// Binning on MI355X is organised around the 160 KB LDS instead of a global radix
// sort: splats are first dropped, unsorted, into their tile's segment (atomic slot
// allocation, ggs_k_scatter), then every tile sorts its own segment in LDS.  The
// whole pipeline moves each (splat, tile) key through HBM once in and once out
// (N * (8 + 8 + 4) B) instead of the ~6 read+write passes of an LSD radix sort
// over 64-bit keys.  Roofline: HBM for the traffic, but the sort itself is
// LDS/barrier bound.
#include "ggs_kernels.h"

// K2: grid V, block 1024.  Exclusive scan of tile_count[v][:] -> tile_offset[v][:],
// view total -> claims a segment [view_base, view_base + total) of the key buffer.
__global__ __launch_bounds__(1024) void ggs_k_scan_tiles(ScanArgs a) {
    const int v = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t* cnt = a.tile_count + (size_t)v * a.T;
    uint32_t* off = a.tile_offset + (size_t)v * a.T;
    const int per = (a.T + 1023) / 1024;
    const int t0 = tid * per;
    __shared__ uint32_t s_bucket[GGS_NBUCKET], s_region[GGS_NBUCKET * GGS_NREGION];
    if (tid < GGS_NBUCKET) s_bucket[tid] = 0;
    if (tid < GGS_NBUCKET * GGS_NREGION) s_region[tid] = 0;
    __syncthreads();
    // ~90 % of the tiles of an image are empty: counted in a register and added once per wave -- as per-tile LDS
    // atomics on the one "empty" counter they serialised (8160 same-address atomics = most of this kernel's 15 us)
    uint32_t local = 0, n_empty = 0;
    for (int i = 0; i < per; ++i)
        if (t0 + i < a.T) {
            const uint32_t c = cnt[t0 + i];
            local += c;
            if (c == 0) ++n_empty;
            else {
                const int b = ggs_len_bucket(c);
                atomicAdd(&s_bucket[b], 1u);
                atomicAdd(&s_region[b * GGS_NREGION + ggs_tile_region(t0 + i, a.gx)], 1u);
            }
        }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_empty += __shfl_xor(n_empty, d);
    if ((tid & 63) == 0 && n_empty) atomicAdd(&s_bucket[GGS_NBUCKET - 1], n_empty);
    // inclusive scan inside each wave64, then across the 16 waves
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t x = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __shared__ uint32_t wsum[16];
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) wbase += wsum[w];
        total += wsum[w];
    }
    uint32_t run = wbase + x - local;
    for (int i = 0; i < per; ++i)
        if (t0 + i < a.T) {
            off[t0 + i] = run;
            run += cnt[t0 + i];
        }
    if (tid < GGS_NBUCKET && s_bucket[tid]) atomicAdd(&a.bucket_count[tid], s_bucket[tid]);
    if (tid < GGS_NBUCKET * GGS_NREGION && s_region[tid]) atomicAdd(&a.region_count[tid], s_region[tid]);
    if (tid == 0) {
        const unsigned long long base = atomicAdd(&a.header->num_rendered, (unsigned long long)total);
        a.view_base[v] = base;
        if (base + total > a.capacity) atomicExch(&a.header->overflow, 1ull);
    }
}

namespace {
// Rank inside its class (0 .. n_c - 1) of the kk-th tile of region g: see ggs_k_order_tiles.  `start` = first non-empty rank of
// the class, k + 1 = distance of consecutive non-empty ranks in order[] (odd), m[h] = tiles of the class in region h.
__device__ __forceinline__ uint32_t ggs_region_rank(uint32_t start, uint32_t n_c, uint32_t k, const uint32_t* m, int g, uint32_t kk) {
    const uint32_t step = (k + 1) & 7u;                       // odd
    uint32_t inv = 1;                                         // step * inv == 1 (mod 8)
    for (uint32_t t = 1; t < 8; t += 2) if (((step * t) & 7u) == 1u) inv = t;
    const uint32_t x0 = (start * step) & 7u;                  // XCD of the class's first rank
    // ranks of the class on XCD h: first_h + 8 j, j < S_h
    const uint32_t first = (((uint32_t)g + 8u - x0) * inv) & 7u;
    const uint32_t S = first < n_c ? (n_c - 1 - first) / 8 + 1 : 0;
    if (kk < S) return first + 8 * kk;
    uint32_t o = kk - S;                                      // overflow index inside the region ...
    for (int h = 0; h < g; ++h) {                             // ... made global over the regions in front
        const uint32_t fh = (((uint32_t)h + 8u - x0) * inv) & 7u;
        const uint32_t Sh = fh < n_c ? (n_c - 1 - fh) / 8 + 1 : 0;
        if (m[h] > Sh) o += m[h] - Sh;
    }
    for (int h = 0; h < GGS_NREGION; ++h) {                   // the o-th rank left free by an under-full region
        const uint32_t fh = (((uint32_t)h + 8u - x0) * inv) & 7u;
        const uint32_t Sh = fh < n_c ? (n_c - 1 - fh) / 8 + 1 : 0;
        const uint32_t free_h = Sh > m[h] ? Sh - m[h] : 0;
        if (o < free_h) return fh + 8 * (m[h] + o);
        o -= free_h;
    }
    return 0;       // not reached: the overflow of the full regions equals the room of the others
}
}  // namespace

// K2b: grid ceil(V*T/256), block 256.  Counting sort of the (view, tile) work items by list-length
// class (longest first) into order[]; inside a class the order is arbitrary.  Empty tiles (class 7:
// the forward only writes the background there, HBM-bound) are interleaved evenly between the
// non-empty ones (ALU-bound) so that the two kinds of work overlap instead of running back to back:
// with NE non-empty and E empty items and k = E / NE rounded down to even, item slots repeat
// [1 non-empty, k empty] and the E - k NE left-over empties go last.
__global__ __launch_bounds__(256) void ggs_k_order_tiles(OrderArgs a) {
    // XCD-aware placement inside a class (ggs_tile_region): the class owns the non-empty ranks [start, start + n_c); rank r sits
    // at order[r (k + 1)], i.e. on XCD (r (k + 1)) % 8 -- k + 1 is odd, so the XCDs of consecutive ranks cycle through all 8 with
    // a fixed step.  Region g of the class gets the ranks whose XCD is g, in arrival order; what does not fit (the class holds
    // more tiles of g than an eighth of its ranks) takes the ranks the under-full regions leave free, through a fixed
    // (prefix-sum) assignment: a bijection, no second pass.
    __shared__ uint32_t s_n[GGS_NBUCKET * GGS_NREGION], s_base[GGS_NBUCKET * GGS_NREGION], s_ne[1], s_nb[1];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    if (tid < GGS_NBUCKET * GGS_NREGION) s_n[tid] = 0;
    if (tid == 0) { s_ne[0] = 0; }
    __syncthreads();
    int b = 0, g = 0;
    uint32_t rank = 0;
    if (i < a.n_items) {
        b = ggs_len_bucket(a.tile_count[i]);
        g = b == GGS_NBUCKET - 1 ? 0 : ggs_tile_region(i % a.T, a.gx);
        rank = atomicAdd(&s_n[b * GGS_NREGION + g], 1u);
    }
    __syncthreads();
    if (tid < GGS_NBUCKET * GGS_NREGION) s_base[tid] = s_n[tid] ? atomicAdd(&a.region_cursor[tid], s_n[tid]) : 0;
    __syncthreads();
    (void)s_nb;
    if (i < a.n_items) {
        const uint32_t E = a.bucket_count[GGS_NBUCKET - 1];
        const uint32_t NE = (uint32_t)a.n_items - E;
        // k even => period k + 1 odd: workgroup b lands on XCD b % 8 (observed), so an even period would
        // park all the non-empty tiles on a subset of the 8 XCDs
        const uint32_t k = NE ? (E / NE) & ~1u : 0;
        const uint32_t kk = s_base[b * GGS_NREGION + g] + rank;      // index inside (class, region); class 15: among the empties
        uint32_t pos;
        if (b != GGS_NBUCKET - 1) {
            uint32_t start = 0;
            for (int c = 0; c < b; ++c) start += a.bucket_count[c];
            pos = (start + ggs_region_rank(start, a.bucket_count[b], k, a.region_count + b * GGS_NREGION, g, kk)) * (k + 1);
        } else if (kk < k * NE) pos = (kk / k) * (k + 1) + 1 + (kk % k);
        else pos = NE * (k + 1) + (kk - k * NE);
        a.order[pos] = (uint32_t)i;
    }
}

// K2 + K2b in one launch for a SINGLE view (grid 1, block 1024): the one workgroup that scans the view's histogram already
// holds the complete list-length class histogram in LDS, so it places the work items right away -- a single-view iteration
// is a chain of latency-bound launches and this removes one kernel boundary and the second pass over
// tile_count.  Same order[] layout as ggs_k_order_tiles (non-empty items longest class first at r * (k + 1), empties
// interleaved); inside a class the order is arbitrary (the XCD-aware placement of ggs_k_order_tiles measured no
// gain for a single view: 2.19k against 2.21k iterations / s).
__global__ __launch_bounds__(1024) void ggs_k_scan_order_one(ScanArgs a, uint32_t* order) {
    const int tid = threadIdx.x;
    const uint32_t* cnt = a.tile_count;
    uint32_t* off = a.tile_offset;
    const int per = (a.T + 1023) / 1024;
    const int t0 = tid * per;
    __shared__ uint32_t s_bucket[GGS_NBUCKET], s_start[GGS_NBUCKET], s_cur[GGS_NBUCKET];
    __shared__ uint32_t wsum[16], wemp[16];
    if (tid < GGS_NBUCKET) { s_bucket[tid] = 0; s_cur[tid] = 0; }
    __syncthreads();
    // The thread's tile counts are loaded ONCE, all loads in flight together (this single workgroup is a latency chain: as two
    // conditional loops that re-read tile_count it spent most of its 14 us waiting for one dependent global load after the other).
    constexpr int PER_REG = 16;                      // up to 16384 tiles (4K: 32640 tiles take the generic path)
    const bool in_regs = per <= PER_REG;
    uint32_t creg[PER_REG];
#pragma unroll
    for (int i = 0; i < PER_REG; ++i) creg[i] = (in_regs && i < per && t0 + i < a.T) ? cnt[t0 + i] : 0u;
    uint32_t local = 0, n_empty = 0;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < PER_REG; ++i)
            if (i < per && t0 + i < a.T) {
                const uint32_t c = creg[i];
                local += c;
                if (c == 0) ++n_empty;
                else atomicAdd(&s_bucket[ggs_len_bucket(c)], 1u);
            }
    } else {
    for (int i = 0; i < per; ++i)
        if (t0 + i < a.T) {
            const uint32_t c = cnt[t0 + i];
            local += c;
            if (c == 0) ++n_empty;
            else atomicAdd(&s_bucket[ggs_len_bucket(c)], 1u);
        }
    }
    // inclusive scans (list entries, empty tiles) inside each wave64, then across the 16 waves
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t x = local, e = n_empty;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d), z = __shfl_up(e, d);
        if (lane >= d) { x += y; e += z; }
    }
    if (lane == 63) { wsum[wave] = x; wemp[wave] = e; }
    __syncthreads();
    uint32_t wbase = 0, total = 0, ebase = 0, E = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) { wbase += wsum[w]; ebase += wemp[w]; }
        total += wsum[w]; E += wemp[w];
    }
    if (tid < GGS_NBUCKET) {
        uint32_t start = 0;
        for (int k = 0; k < tid; ++k) start += s_bucket[k];
        s_start[tid] = start;
        const uint32_t n = tid == GGS_NBUCKET - 1 ? E : s_bucket[tid];
        if (n) atomicAdd(&a.bucket_count[tid], n);
    }
    __syncthreads();
    const uint32_t NE = (uint32_t)a.T - E;
    const uint32_t k = NE ? (E / NE) & ~1u : 0;            // even: see ggs_k_order_tiles
    uint32_t run = wbase + x - local, er = ebase + e - n_empty;
    for (int i = 0; i < per; ++i)
        if (t0 + i < a.T) {
            uint32_t c = 0;
            if (in_regs) {
#pragma unroll
                for (int k = 0; k < PER_REG; ++k) c = k == i ? creg[k] : c;     // static register indices (no scratch)
            } else c = cnt[t0 + i];
            off[t0 + i] = run;
            run += c;
            uint32_t pos;
            if (c) {
                const int b = ggs_len_bucket(c);
                pos = (s_start[b] + atomicAdd(&s_cur[b], 1u)) * (k + 1);
            } else {
                const uint32_t r = er++;
                pos = r < k * NE ? (r / k) * (k + 1) + 1 + (r % k) : NE * (k + 1) + (r - k * NE);
            }
            order[pos] = (uint32_t)(t0 + i);
        }
    if (tid == 0) {
        const unsigned long long base = atomicAdd(&a.header->num_rendered, (unsigned long long)total);
        a.view_base[0] = base;
        if (base + total > a.capacity) atomicExch(&a.header->overflow, 1ull);
    }
}

namespace {

// sort order = (depth bits, Gaussian id): the key is depth << 32 | id << GGS_NQ | sub-block mask, so the plain 64-bit
// comparison already is that order.  The id word handed to the render kernels is mask << GGS_ID_BITS | id = the low
// key word rotated right by GGS_NQ.
__device__ __forceinline__ uint32_t key_to_id_word(unsigned long long k) {
    return __builtin_amdgcn_alignbit((uint32_t)k, (uint32_t)k, GGS_NQ);
}

template <typename KeyPtr>
__device__ __forceinline__ void cmp_exchange(KeyPtr key, int i, int p, int L) {
    if (p < L) {
        const unsigned long long a = key[i], b = key[p];
        if (a > b) { key[i] = b; key[p] = a; }
    }
}

// Ascending-only bitonic network over `n2` = 2^m >= L slots; slots >= L are virtual +inf keys, so a
// compare-exchange whose upper partner is >= L is a no-op.  Thread t handles pair-slot t (+256 m).
// With 256 threads a wave64 owns the pairs of one aligned 128-element chunk, so every sub-step whose
// partner distance is < 128 only touches data the SAME wave wrote: those need no workgroup barrier
// (DS operations of one wave execute in order), only the compiler fence of wave_barrier().
template <typename KeyPtr, bool WAVE_LOCAL_OK>
__device__ __forceinline__ void bitonic_sort(KeyPtr key, int L, int n2, int tid) {
    const int half = n2 >> 1;
    for (int lk = 1; (1 << lk) <= n2; ++lk) {
        const int k = 1 << lk, lhk = lk - 1;
        // flip step: partner = mirror inside the block of size k (distance up to k - 1)
        for (int t = tid; t < half; t += 256) {
            const int blk = t >> lhk, o = t & ((1 << lhk) - 1);
            cmp_exchange(key, (blk << lk) + o, (blk << lk) + k - 1 - o, L);
        }
        if (WAVE_LOCAL_OK && k <= 128) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        for (int lj = lk - 2; lj >= 0; --lj) {
            const int j = 1 << lj;
            for (int t = tid; t < half; t += 256) {
                const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1));
                cmp_exchange(key, i, i + j, L);
            }
            // the NEXT sub-step (distance j/2) reads what this one wrote: wave-local iff j <= 64
            if (WAVE_LOCAL_OK && j <= 64 && !(lj == 0 && (k << 1) > 128)) __builtin_amdgcn_wave_barrier();
            else __syncthreads();
        }
    }
}

}  // namespace

// K4a-small: persistent grid, block 256 = 4 independent wave64s, ONE TILE PER WAVE for lists of up to
// GGS_SORT_WAVE_CAP keys, sorted IN REGISTERS: lane l holds the E = n2 / 64 consecutive keys [l E, l E + E) of the
// (+inf padded) list.  Of the 45 sub-steps of the 512-key bitonic network 24 have both partners in the same lane
// (compare + 4 selects, no memory at all); in the other 21 the partner key comes from lane l ^ x through the LDS
// crossbar (ds_swizzle / ds_bpermute: no LDS memory, no bank conflicts) and each lane keeps the minimum or the
// maximum.  ~1000 VALU + 340 crossbar ops per 512 keys against ~3200 VALU + 720 64-bit LDS reads / writes for the
// network run out of an LDS array; no barriers of any kind.
#define GGS_SORT_WAVE_CAP 1024
namespace {

template <int X>
__device__ __forceinline__ uint32_t lane_xor_get(uint32_t v, int lane) {
    if constexpr (X < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (X << 10));   // bit-mask mode: xor X
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ X) << 2, (int)v);
}
__device__ __forceinline__ unsigned long long lane_xor_get64(unsigned long long k, int x, int lane) {
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    switch (x) {                         // x is a compile-time constant after unrolling
        case 1: lo = lane_xor_get<1>(lo, lane); hi = lane_xor_get<1>(hi, lane); break;
        case 2: lo = lane_xor_get<2>(lo, lane); hi = lane_xor_get<2>(hi, lane); break;
        case 3: lo = lane_xor_get<3>(lo, lane); hi = lane_xor_get<3>(hi, lane); break;
        case 4: lo = lane_xor_get<4>(lo, lane); hi = lane_xor_get<4>(hi, lane); break;
        case 7: lo = lane_xor_get<7>(lo, lane); hi = lane_xor_get<7>(hi, lane); break;
        case 8: lo = lane_xor_get<8>(lo, lane); hi = lane_xor_get<8>(hi, lane); break;
        case 15: lo = lane_xor_get<15>(lo, lane); hi = lane_xor_get<15>(hi, lane); break;
        case 16: lo = lane_xor_get<16>(lo, lane); hi = lane_xor_get<16>(hi, lane); break;
        case 31: lo = lane_xor_get<31>(lo, lane); hi = lane_xor_get<31>(hi, lane); break;
        case 32: lo = lane_xor_get<32>(lo, lane); hi = lane_xor_get<32>(hi, lane); break;
        default: lo = lane_xor_get<63>(lo, lane); hi = lane_xor_get<63>(hi, lane); break;
    }
    return ((unsigned long long)hi << 32) | lo;
}
// lanes whose bit `b` is clear: the lower partner of a pair (lane, lane ^ x) with b the top bit of x
__device__ __forceinline__ unsigned long long lower_lanes(int b) {
    return b == 1 ? 0x5555555555555555ull : b == 2 ? 0x3333333333333333ull : b == 4 ? 0x0F0F0F0F0F0F0F0Full
         : b == 8 ? 0x00FF00FF00FF00FFull : b == 16 ? 0x0000FFFF0000FFFFull : 0x00000000FFFFFFFFull;
}
__device__ __forceinline__ uint32_t sel_u32(unsigned long long m, uint32_t a, uint32_t b) {           // m ? a : b
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

// one sub-step whose partner lives in lane ^ X: register r meets register (MIRROR ? E - 1 - r : r) of that lane;
// B = top bit of X decides which lane of the pair is the lower one
template <int E, int X, int B, bool MIRROR>
__device__ __forceinline__ void sort_cross(unsigned long long (&k)[E], int lane) {
    const unsigned long long lower = lower_lanes(B);
    unsigned long long o[E];
#pragma unroll
    for (int r = 0; r < E; ++r) o[r] = lane_xor_get64(k[MIRROR ? E - 1 - r : r], X, lane);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        // the lower lane keeps the smaller key: take the partner's iff (mine > other) == lower
        const unsigned long long take = ~(__builtin_amdgcn_ballot_w64(k[r] > o[r]) ^ lower);
        const uint32_t lo = sel_u32(take, (uint32_t)o[r], (uint32_t)k[r]);
        const uint32_t hi = sel_u32(take, (uint32_t)(o[r] >> 32), (uint32_t)(k[r] >> 32));
        k[r] = ((unsigned long long)hi << 32) | lo;
    }
}
// both partners in this lane: registers (r, r ^ XR) for the r whose bit BIT is clear; the lower index keeps the minimum
template <int E, int XR, int BIT>
__device__ __forceinline__ void sort_local(unsigned long long (&k)[E]) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        if ((r & BIT) != 0) continue;
        const unsigned long long x = k[r], y = k[r ^ XR];
        const bool gt = x > y;
        k[r] = gt ? y : x; k[r ^ XR] = gt ? x : y;
    }
}
template <int E, int J>
__device__ __forceinline__ void sort_half_cleaners(unsigned long long (&k)[E], int lane) {   // distances J, J/2, ..., 1
    if constexpr (J >= 1) {
        if constexpr (J < E) sort_local<E, J, J>(k);
        else sort_cross<E, J / E, J / E, false>(k, lane);
        sort_half_cleaners<E, J / 2>(k, lane);
    }
}
template <int E, int KK>
__device__ __forceinline__ void sort_phases(unsigned long long (&k)[E], int lane) {          // merge phases KK, 2 KK, ... 64 E
    if constexpr (KK <= 64 * E) {
        // flip step: mirror inside aligned blocks of KK keys, then the half cleaners
        if constexpr (KK <= E) sort_local<E, KK - 1, KK / 2>(k);
        else sort_cross<E, KK / E - 1, KK / (2 * E), true>(k, lane);
        sort_half_cleaners<E, KK / 4>(k, lane);
        sort_phases<E, KK * 2>(k, lane);
    }
}
template <int E>
__device__ __forceinline__ void wave_sort_regs(unsigned long long (&k)[E], int lane) { sort_phases<E, 2>(k, lane); }

template <int E>
__device__ __forceinline__ void sort_one_tile_regs(const unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids,
                                                   int L, int lane) {
    unsigned long long k[E];
#pragma unroll
    for (int r = 0; r < E; ++r) k[r] = lane * E + r < L ? keys[lane * E + r] : ~0ull;     // +inf padding sorts last
    wave_sort_regs<E>(k, lane);
#pragma unroll
    for (int r = 0; r < E; ++r)
        if (lane * E + r < L) ids[lane * E + r] = key_to_id_word(k[r]);
}

__device__ __forceinline__ void sort_tiles_wave_body(const SortArgs& a, uint32_t bid, uint32_t nblk) {
    // the wave index is uniform by construction; saying so keeps the item loop, its loads and the length dispatch scalar
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const NonEmptyItems it = ggs_nonempty_items(a.bucket_count, (uint32_t)a.n_items);
    // persistent waves over the non-empty work items only (the empty ~90 % of an image never cost a
    // workgroup launch); round-robin over a longest-first list keeps the waves evenly loaded
    for (uint32_t r = bid * 4 + wave; r < it.n; r += nblk * 4) {
        const uint32_t item = a.order[(size_t)r * it.stride];
        const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T);
        const int L = (int)a.tile_count[(size_t)v * a.T + t];
        if (L > GGS_SORT_WAVE_CAP) continue;                  // longer lists: the workgroup variant
        const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
        const unsigned long long* keys = a.keys + base;
        uint32_t* ids = a.ids + base;
        if (L <= 64) sort_one_tile_regs<1>(keys, ids, L, lane);
        else if (L <= 128) sort_one_tile_regs<2>(keys, ids, L, lane);
        else if (L <= 256) sort_one_tile_regs<4>(keys, ids, L, lane);
        else if (L <= 512) sort_one_tile_regs<8>(keys, ids, L, lane);
        else sort_one_tile_regs<16>(keys, ids, L, lane);
    }
}

// K4a-large: persistent grid over the work items with more than GGS_SORT_WAVE_CAP keys (first in order[]), block 256.
// Sorts the tile's key segment by (depth bits, id) and
// writes the id-word list (quadrant mask << 28 | id) the render kernels walk.
__device__ __forceinline__ void sort_tiles_block_body(const SortArgs& a, unsigned long long* s_key, uint32_t bid, uint32_t nblk) {
    const int tid = threadIdx.x;
    const NonEmptyItems it = ggs_nonempty_items(a.bucket_count, (uint32_t)a.n_items);
    for (uint32_t r = bid; r < it.n_long; r += nblk) {
    const uint32_t item = a.order[(size_t)r * it.stride];
    const int v = (int)(item / (uint32_t)a.T), t = (int)(item % (uint32_t)a.T);
    const int L = (int)a.tile_count[(size_t)v * a.T + t];
    __syncthreads();                                      // s_key reuse across iterations
    if (L <= GGS_SORT_WAVE_CAP) continue;                 // short lists: ggs_k_sort_tiles_wave
    const size_t base = (size_t)a.view_base[v] + a.tile_offset[(size_t)v * a.T + t];
    unsigned long long* keys = a.keys + base;
    uint32_t* ids = a.ids + base;
    int n2 = 2;
    while (n2 < L) n2 <<= 1;
    if (L <= GGS_SORT_CAP) {
        for (int i = tid; i < L; i += 256) s_key[i] = keys[i];
        __syncthreads();
        bitonic_sort<unsigned long long*, true>(s_key, L, n2, tid);
        __syncthreads();
        for (int i = tid; i < L; i += 256) ids[i] = key_to_id_word(s_key[i]);
    } else {
        // Oversized list (pathological: > 4096 splats on one 16x16 tile): same network on the
        // global segment.  Slow but exact; plain loads/stores are ordered by __syncthreads
        // inside one workgroup (same CU, same L1).
        bitonic_sort<unsigned long long*, false>(keys, L, n2, tid);
        for (int i = tid; i < L; i += 256) ids[i] = key_to_id_word(keys[i]);
    }
    }
}
}  // namespace

// K4a: ONE launch for both list classes -- workgroups [0, n_block) sort the long lists (256 threads per tile, first in
// the grid so they start first), the rest are 4 independent waves each sorting short lists.  As two back-to-back
// kernels the long-list pass (few, slow workgroups) and the short-list pass could not overlap: 57 -> ~30 us for a
// single view, where both are latency chains.
__global__ __launch_bounds__(256) void ggs_k_sort_tiles(SortArgs a, unsigned n_block) {
    if (a.header->overflow) return;
    __shared__ unsigned long long s_mem[GGS_SORT_CAP];
    if (blockIdx.x < n_block) sort_tiles_block_body(a, s_mem, blockIdx.x, n_block);
    else sort_tiles_wave_body(a, blockIdx.x - n_block, gridDim.x - n_block);
}
// Short lists only, no LDS: for launches large enough to fill the chip the short-list waves are throughput bound and
// want the occupancy the 32 KB of the merged kernel would cap.
__global__ __launch_bounds__(256) void ggs_k_sort_tiles_wave(SortArgs a) {
    if (a.header->overflow) return;
    sort_tiles_wave_body(a, blockIdx.x, gridDim.x);
}
