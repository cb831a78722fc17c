// ggs_knn.hip -- mean squared distance to the 3 nearest neighbours of every point.
//
// Replaces `simple_knn._C.distCUDA2`, the reference's second external CUDA dependency
// (scene/gaussian_model.py:20,135 and scene/mesh_gaussian_model.py:23,233; built by setup.sh:35-37 from
// graphdeco-inria/gaussian-splatting/submodules/simple-knn).  It is called once, at model initialisation,
// to size the initial Gaussians: scales = log(sqrt(clamp_min(distCUDA2(xyz), 1e-7))).
//
// Upstream sorts the points along a Morton curve and searches neighbouring boxes; the result is the exact
// 3-NN mean (self excluded).  Two exact implementations here:
//   * ggs_dist2_3nn      -- brute force, no workspace: one lane per query, all P candidates streamed through LDS in
//                           1024-point tiles (wave-uniform ds_read_b128).  O(P^2): 4.2 ms at 100k points, ~110 ms at 500k.
//   * ggs_dist2_3nn_grid -- uniform grid over the bounding box (about two cells per point, so a surface-bound cloud -- the
//                           garment -- has ~10 points per occupied cell), counting sort of the points by cell, one lane per
//                           point in sorted order (a wave's lanes share a cell neighbourhood: coherent L1 traffic) searching
//                           the (2r+1)^3 cells around its own, r growing until the third-best distance is provably inside
//                           the searched box.  Same fp32 distance expression as the brute force, so the two agree bit for
//                           bit; no host sync (grid parameters are derived on the device from the bounding box).
#include "ggs_kernels.h"

namespace {

#define KNN_TILE 1024

__device__ __forceinline__ void insert3(float d, float& b0, float& b1, float& b2) {
    // keep the three smallest of {b0 <= b1 <= b2, d}
    const float t0 = fminf(b0, d), u0 = fmaxf(b0, d);
    const float t1 = fminf(b1, u0), u1 = fmaxf(b1, u0);
    b0 = t0; b1 = t1; b2 = fminf(b2, u1);
}

__global__ __launch_bounds__(256) void k_dist2_3nn(int P, const float* __restrict__ pts, float* __restrict__ out) {
    __shared__ float4 s_p[KNN_TILE];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ii = i < P ? i : P - 1;
    const float qx = pts[3 * (size_t)ii], qy = pts[3 * (size_t)ii + 1], qz = pts[3 * (size_t)ii + 2];
    float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;
    for (int base = 0; base < P; base += KNN_TILE) {
        const int n = min(KNN_TILE, P - base);
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += 256) {
            const size_t j = (size_t)(base + t);
            s_p[t] = make_float4(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], 0.f);
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < n; ++t) {
            const float4 c = s_p[t];
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            float d = dx * dx + dy * dy + dz * dz;
            d = (base + t == i) ? 3.402823466e+38f : d;       // a point is not its own neighbour
            insert3(d, b0, b1, b2);
        }
    }
    if (i < P) out[i] = (b0 + b1 + b2) / 3.0f;
}

// ---- grid search ----------------------------------------------------------------------------------------------------
struct KnnGrid {                  // lives at the head of the workspace, written by k_knn_setup
    float lo[3], inv_h, h;
    int n[3], ncell;
};
#define KNN_HDR_BYTES 256
#define KNN_MAX_PARTIAL 1024

struct KnnWs {
    KnnGrid* grid; float* partial; uint32_t* offset; uint32_t* cursor; uint32_t* cell_of; float4* sorted;
    unsigned ncell_max;
};

__device__ __forceinline__ int knn_cell_axis(float v, float lo, float inv_h, int n) {
    const int c = (int)floorf((v - lo) * inv_h);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

// per-block bounding boxes (no initialisation needed: every block writes its own slot)
__global__ __launch_bounds__(256) void k_knn_bbox(int P, const float* __restrict__ pts, float* __restrict__ partial) {
    float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)P; i += (size_t)gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = pts[3 * i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    __shared__ float red[6][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d)); }
        if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const float* r = red[threadIdx.x];
        partial[blockIdx.x * 6 + threadIdx.x] = threadIdx.x < 3 ? fminf(fminf(r[0], r[1]), fminf(r[2], r[3]))
                                                                : fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
    }
}

// one wave: reduce the block boxes, choose the cell size (about two cells per point, at most ncell_max cells, at most 1024
// per axis; a degenerate axis gets one cell)
__global__ __launch_bounds__(64) void k_knn_setup(int P, int n_partial, KnnWs w) {
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = 3.402823466e+38f, h = -3.402823466e+38f;
        for (int b = threadIdx.x; b < n_partial; b += 64) { l = fminf(l, w.partial[b * 6 + a]); h = fmaxf(h, w.partial[b * 6 + 3 + a]); }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { l = fminf(l, __shfl_xor(l, d)); h = fmaxf(h, __shfl_xor(h, d)); }
        lo[a] = l; hi[a] = h;
    }
    if (threadIdx.x != 0) return;
    float e[3], emax = 0.f;
    for (int a = 0; a < 3; ++a) { e[a] = hi[a] - lo[a]; emax = fmaxf(emax, e[a]); }
    // all points equal, or a non-finite coordinate (inf - inf = NaN, inf extents): ONE cell -- the search then degenerates to the
    // exhaustive scan, which returns garbage for garbage input like the brute-force kernel does, instead of an index fault
    bool finite = true;
    for (int a = 0; a < 3; ++a) finite = finite && (e[a] >= 0.f) && (e[a] < 3.0e38f) && (lo[a] > -3.0e38f) && (hi[a] < 3.0e38f);
    if (!finite) { for (int a = 0; a < 3; ++a) { e[a] = 0.f; lo[a] = 0.f; } emax = 0.f; }
    if (!(emax > 0.f)) emax = 1.f;
    for (int a = 0; a < 3; ++a) e[a] = fmaxf(e[a], emax * 1e-6f);
    float target = fminf(2.f * (float)P, (float)w.ncell_max);
    float h = cbrtf(e[0] * e[1] * e[2] / target);
    int n[3];
    for (int it = 0; it < 4; ++it) {
        double prod = 1.0;
        // !(f >= 1): also catches NaN (a NaN would pass both clamps and convert to 0 cells)
        for (int a = 0; a < 3; ++a) { float f = floorf(e[a] / h); n[a] = !(f >= 1.f) ? 1 : (f > 1024.f ? 1024 : (int)f); prod *= n[a]; }
        if (prod <= (double)w.ncell_max) break;
        h *= cbrtf((float)(prod / (double)w.ncell_max)) * 1.02f;
    }
    // Degenerate extents (planar / linear clouds: an axis pinned at the 1024-cell clamp keeps the product from shrinking with
    // h): grow h until the clamped grid fits, recomputing n after EVERY rescale, instead of giving up on the grid (one cell =
    // an O(P^2) scan per lane).
    for (int it = 0; it < 64 && (double)n[0] * n[1] * n[2] > (double)w.ncell_max; ++it) {
        h *= 1.26f;
        for (int a = 0; a < 3; ++a) { float f = floorf(e[a] / h); n[a] = !(f >= 1.f) ? 1 : (f > 1024.f ? 1024 : (int)f); }
    }
    if ((double)n[0] * n[1] * n[2] > (double)w.ncell_max) { n[0] = n[1] = n[2] = 1; }
    // cells of size h starting at lo; the last cell of an axis absorbs the remainder (cell index is clamped)
    KnnGrid g;
    for (int a = 0; a < 3; ++a) { g.lo[a] = lo[a]; g.n[a] = n[a]; }
    g.h = h; g.inv_h = 1.f / h; g.ncell = n[0] * n[1] * n[2];
    *w.grid = g;
}

__global__ __launch_bounds__(256) void k_knn_count(int P, const float* __restrict__ pts, KnnWs w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const KnnGrid g = *w.grid;
    const int cx = knn_cell_axis(pts[3 * (size_t)i], g.lo[0], g.inv_h, g.n[0]);
    const int cy = knn_cell_axis(pts[3 * (size_t)i + 1], g.lo[1], g.inv_h, g.n[1]);
    const int cz = knn_cell_axis(pts[3 * (size_t)i + 2], g.lo[2], g.inv_h, g.n[2]);
    const uint32_t c = (uint32_t)((cz * g.n[1] + cy) * g.n[0] + cx);
    w.cell_of[i] = c;
    atomicAdd(&w.offset[c], 1u);
}

// Exclusive scan of the cell counts in place, cursor = copy of the offsets.  Three coalesced passes over tiles of
// KNN_SCAN_TILE cells (a single workgroup walking a million counters with a 4 KB stride per lane took 9.8 of the 10 ms of a
// 500k-point call): tile sums -> scan of the tile sums (one workgroup, <= 1024 tiles per round) -> per-tile scan + base.
#define KNN_SCAN_TILE 2048
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_wsum, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
    __syncthreads();
    if (lane == 63) s_wsum[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k < wave) base += s_wsum[k]; total += s_wsum[k]; }
    return base + x - v;
}
__global__ __launch_bounds__(256) void k_knn_scan_sums(KnnWs w, uint32_t* __restrict__ tile_sum) {
    const int ncell = w.grid->ncell;
    const int base = blockIdx.x * KNN_SCAN_TILE;
    if (base >= ncell) return;
    uint32_t v = 0;
    for (int k = threadIdx.x; k < KNN_SCAN_TILE; k += 256) if (base + k < ncell) v += w.offset[base + k];
    __shared__ uint32_t s_wsum[4];
    uint32_t total;
    block_exclusive_scan_256(v, s_wsum, total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_knn_scan_top(KnnWs w, uint32_t* __restrict__ tile_sum) {
    const int ncell = w.grid->ncell;
    const int ntile = (ncell + KNN_SCAN_TILE - 1) / KNN_SCAN_TILE;
    __shared__ uint32_t s_wsum[4];
    uint32_t carry = 0;
    for (int b0 = 0; b0 < ntile; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const uint32_t v = i < ntile ? tile_sum[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_256(v, s_wsum, total);
        if (i < ntile) tile_sum[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) w.offset[ncell] = carry;            // = P
}
__global__ __launch_bounds__(256) void k_knn_scan_apply(KnnWs w, const uint32_t* __restrict__ tile_sum) {
    const int ncell = w.grid->ncell;
    const int base = blockIdx.x * KNN_SCAN_TILE;
    if (base >= ncell) return;
    // thread t owns the 8 consecutive cells base + 8 t .. + 7 (two 16-byte loads per lane, contiguous across the wave)
    uint32_t c[8], v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int i = base + 8 * (int)threadIdx.x + k; c[k] = i < ncell ? w.offset[i] : 0; v += c[k]; }
    __shared__ uint32_t s_wsum[4];
    uint32_t total;
    uint32_t run = tile_sum[blockIdx.x] + block_exclusive_scan_256(v, s_wsum, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = base + 8 * (int)threadIdx.x + k;
        if (i < ncell) { w.offset[i] = run; w.cursor[i] = run; }
        run += c[k];
    }
}

__global__ __launch_bounds__(256) void k_knn_scatter(int P, const float* __restrict__ pts, KnnWs w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t pos = atomicAdd(&w.cursor[w.cell_of[i]], 1u);
    w.sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

// one lane per point, in cell order
__global__ __launch_bounds__(256) void k_knn_query(int P, KnnWs w, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P) return;
    const KnnGrid g = *w.grid;
    const float4 q = w.sorted[t];
    const int qi = __float_as_int(q.w);
    const int c0[3] = {knn_cell_axis(q.x, g.lo[0], g.inv_h, g.n[0]), knn_cell_axis(q.y, g.lo[1], g.inv_h, g.n[1]),
                       knn_cell_axis(q.z, g.lo[2], g.inv_h, g.n[2])};
    const float qv[3] = {q.x, q.y, q.z};
    float b0 = 3.402823466e+38f, b1 = b0, b2 = b0;
    int r = 0;
    for (;;) {
        // the shell of cells at Chebyshev distance exactly r from the query's cell (r = 0: the cell itself)
        const int z0 = max(c0[2] - r, 0), z1 = min(c0[2] + r, g.n[2] - 1);
        const int y0 = max(c0[1] - r, 0), y1 = min(c0[1] + r, g.n[1] - 1);
        const int x0 = max(c0[0] - r, 0), x1 = min(c0[0] + r, g.n[0] - 1);
        for (int cz = z0; cz <= z1; ++cz)
            for (int cy = y0; cy <= y1; ++cy) {
                const bool edge_row = abs(cz - c0[2]) == r || abs(cy - c0[1]) == r;
                // inside an interior row only the two end cells belong to the shell; the rows of a face are whole.  Cells of
                // one row are contiguous in the sorted array: one candidate range per row (or two single cells)
                const uint32_t rowbase = (uint32_t)((cz * g.n[1] + cy) * g.n[0]);
                for (int part = 0; part < (edge_row ? 1 : 2); ++part) {
                    int xa, xb;
                    if (edge_row) { xa = x0; xb = x1; }
                    else {
                        const int xc = part == 0 ? c0[0] - r : c0[0] + r;
                        if (xc < 0 || xc >= g.n[0] || (part == 1 && r == 0)) continue;
                        xa = xb = xc;
                    }
                    const uint32_t j0 = w.offset[rowbase + xa], j1 = w.offset[rowbase + xb + 1];
                    for (uint32_t j = j0; j < j1; ++j) {
                        const float4 c = w.sorted[j];
                        const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
                        float d = dx * dx + dy * dy + dz * dz;
                        d = __float_as_int(c.w) == qi ? 3.402823466e+38f : d;       // a point is not its own neighbour
                        insert3(d, b0, b1, b2);
                    }
                }
            }
        // everything within the box of cells [c0 - r, c0 + r] has been seen.  Distance from the query to the nearest face
        // of that box that has unexplored cells behind it:
        float reach = 3.402823466e+38f;
        bool all = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // (the cell of a point is floor((v - lo) / h) in fp32; the face coordinate lo + k h is rounded differently, by at
            // most a few ulp of the coordinates involved: slack, so that the test stays conservative)
            const float slack = 2e-6f * (fabsf(qv[a]) + fabsf(g.lo[a]) + (float)(c0[a] + r + 1) * g.h);
            if (c0[a] - r > 0) { reach = fminf(reach, qv[a] - (g.lo[a] + (float)(c0[a] - r) * g.h) - slack); all = false; }
            if (c0[a] + r < g.n[a] - 1) { reach = fminf(reach, (g.lo[a] + (float)(c0[a] + r + 1) * g.h) - qv[a] - slack); all = false; }
        }
        if (all) break;                                   // the whole grid has been searched
        reach = fmaxf(reach, 0.f);
        if (b2 <= reach * reach) break;
        ++r;
    }
    out[qi] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

static size_t knn_partial_bytes(size_t ncell_max) {      // block bounding boxes, later the scan's tile sums
    const size_t a = KNN_MAX_PARTIAL * 6 * sizeof(float), b = ((ncell_max + KNN_SCAN_TILE - 1) / KNN_SCAN_TILE + 1) * 4;
    return ggs_align(a > b ? a : b);
}

extern "C" size_t ggs_dist2_3nn_scratch_bytes(int P) {
    if (P <= 0) return 0;
    const size_t ncell_max = (size_t)2 * P < 64 ? 64 : (size_t)2 * P;
    return KNN_HDR_BYTES + knn_partial_bytes(ncell_max) + 2 * ggs_align((ncell_max + 1) * 4) +
           ggs_align((size_t)P * 4) + ggs_align((size_t)P * 16);
}

extern "C" int ggs_dist2_3nn_grid(int P, const float* points, float* out, void* scratch, void* stream) {
    ggs_clear_error_();
    if (P < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_dist2_3nn_grid: bad size");
    if (P == 0) return GGS_OK;
    if (!points || !out || !scratch) return ggs_fail_(GGS_ERR_ARG, "ggs_dist2_3nn_grid: NULL pointer argument");
    if (P >= (1 << 30)) return ggs_fail_(GGS_ERR_SIZE, "ggs_dist2_3nn_grid: too many points");
    hipStream_t s = (hipStream_t)stream;
    const size_t ncell_max = (size_t)2 * P < 64 ? 64 : (size_t)2 * P;
    char* b = (char*)scratch;
    KnnWs w;
    w.grid = (KnnGrid*)b; b += KNN_HDR_BYTES;
    w.partial = (float*)b; b += knn_partial_bytes(ncell_max);
    w.offset = (uint32_t*)b; b += ggs_align((ncell_max + 1) * 4);
    w.cursor = (uint32_t*)b; b += ggs_align((ncell_max + 1) * 4);
    w.cell_of = (uint32_t*)b; b += ggs_align((size_t)P * 4);
    w.sorted = (float4*)b;
    w.ncell_max = (unsigned)ncell_max;
    const int nblk = (P + 255) / 256, npart = nblk < KNN_MAX_PARTIAL ? nblk : KNN_MAX_PARTIAL;
    if (ggs_zero_async(w.offset, (ncell_max + 1) * 4, s) != hipSuccess)
        return ggs_fail_(GGS_ERR_HIP, "ggs_dist2_3nn_grid: clearing the cell counters failed");
    hipLaunchKernelGGL(k_knn_bbox, dim3((unsigned)npart), dim3(256), 0, s, P, points, w.partial);
    hipLaunchKernelGGL(k_knn_setup, dim3(1), dim3(64), 0, s, P, npart, w);
    hipLaunchKernelGGL(k_knn_count, dim3((unsigned)nblk), dim3(256), 0, s, P, points, w);
    // (the per-block bounding boxes are dead after k_knn_setup: their region holds the scan's tile sums)
    uint32_t* tile_sum = (uint32_t*)w.partial;
    const unsigned ntile_max = (unsigned)((ncell_max + KNN_SCAN_TILE - 1) / KNN_SCAN_TILE);
    hipLaunchKernelGGL(k_knn_scan_sums, dim3(ntile_max), dim3(256), 0, s, w, tile_sum);
    hipLaunchKernelGGL(k_knn_scan_top, dim3(1), dim3(256), 0, s, w, tile_sum);
    hipLaunchKernelGGL(k_knn_scan_apply, dim3(ntile_max), dim3(256), 0, s, w, tile_sum);
    hipLaunchKernelGGL(k_knn_scatter, dim3((unsigned)nblk), dim3(256), 0, s, P, points, w);
    hipLaunchKernelGGL(k_knn_query, dim3((unsigned)nblk), dim3(256), 0, s, P, w, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "dist2_3nn_grid launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

extern "C" int ggs_dist2_3nn(int P, const float* points, float* out, void* stream) {
    ggs_clear_error_();
    if (P < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_dist2_3nn: bad size");
    if (P == 0) return GGS_OK;
    if (!points || !out) return ggs_fail_(GGS_ERR_ARG, "ggs_dist2_3nn: NULL pointer argument");
    hipLaunchKernelGGL(k_dist2_3nn, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, points, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "dist2_3nn launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}
