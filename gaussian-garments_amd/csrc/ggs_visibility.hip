// ggs_visibility.hip -- which Gaussians does the camera see?  First-hit ray casting of P rays
// (camera -> the Gaussian's anchor point on its bound face) against the garment mesh, on the GPU.
//
// Replaces AvatarGaussianModel.get_visible_mask (scene/avatar_gaussian_model.py:227-263), which every
// iteration of the appearance loop moves the mesh to the CPU, builds an open3d / Embree RaycastingScene,
// casts num_gs rays and moves the mask back ("the per-iteration GPU->CPU->GPU round trip that
// dominates s3").  Semantics kept literally: visible  <=>  id of the first triangle hit by the ray == binding.
//
// All rays leave one point, so the mesh is binned ONCE per call in a perspective grid seen from the camera
// (axis = camera -> mesh centroid): a triangle lands in the cells its projected bounding box overlaps, a ray
// only tests the triangles of the single cell its target projects to -- exact, because a ray through (u, w)
// can only hit triangles whose projection contains (u, w).  Triangles touching the camera plane (z <= eps)
// cannot be projected and go to a short "test always" list.  ~10 small kernels, all HBM/latency bound;
// P = F = 100k takes well under a millisecond instead of a host round trip.
#include "ggs_kernels.h"

namespace {

struct VisHeader {
    float sum[3]; unsigned n_sum;        // centroid accumulation
    float fwd[3], right[3], up[3];       // projection frame
    int bmin[2], bmax[2];                // projected bounds, order-preserving int encoding of floats
    float u0, w0, inv_cell_u, inv_cell_w;
    unsigned n_global, overflow, total;
};

struct VisArgs {
    int P, F, Vn, G;
    unsigned cap;                         // capacity of tri_ids
    const float* verts; const int64_t* faces; const float* cam; const float* targets; const int64_t* binding;
    VisHeader* hdr;
    float* proj;                          // [Vn][3] (u, w, z)
    unsigned* cell_count;                 // [G*G]
    unsigned* cell_offset;                // [G*G + 1]
    unsigned* cell_cursor;                // [G*G]
    unsigned* tri_ids;                    // [cap]
    unsigned* global_ids;                 // [F] triangles that cannot be projected
    unsigned char* mask; int* first_hit;
};

__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void k_vis_sum(VisArgs a) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.Vn; i += gridDim.x * 256) {
        s0 += a.verts[3 * (size_t)i]; s1 += a.verts[3 * (size_t)i + 1]; s2 += a.verts[3 * (size_t)i + 2];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s0 += __shfl_xor(s0, d); s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&a.hdr->sum[0], s0); atomicAdd(&a.hdr->sum[1], s1); atomicAdd(&a.hdr->sum[2], s2); }
}

__global__ void k_vis_frame(VisArgs a) {
    if (threadIdx.x || blockIdx.x) return;
    VisHeader* h = a.hdr;
    float f[3] = {h->sum[0] / a.Vn - a.cam[0], h->sum[1] / a.Vn - a.cam[1], h->sum[2] / a.Vn - a.cam[2]};
    float n = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (!(n > 0.f)) { f[0] = 0.f; f[1] = 0.f; f[2] = 1.f; n = 1.f; }
    for (int k = 0; k < 3; ++k) f[k] /= n;
    // any vector not parallel to f
    float t[3] = {0.f, 0.f, 0.f};
    const int ax = fabsf(f[0]) < fabsf(f[1]) ? (fabsf(f[0]) < fabsf(f[2]) ? 0 : 2) : (fabsf(f[1]) < fabsf(f[2]) ? 1 : 2);
    t[ax] = 1.f;
    float r[3] = {f[1] * t[2] - f[2] * t[1], f[2] * t[0] - f[0] * t[2], f[0] * t[1] - f[1] * t[0]};
    n = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; ++k) r[k] /= n;
    const float u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
    for (int k = 0; k < 3; ++k) { h->fwd[k] = f[k]; h->right[k] = r[k]; h->up[k] = u[k]; }
    h->bmin[0] = h->bmin[1] = 0x7fffffff; h->bmax[0] = h->bmax[1] = (int)0x80000000;
}

#define VIS_Z_EPS 1e-6f

__device__ __forceinline__ void vis_project(const VisHeader* h, const float* cam, const float* p, float& u, float& w, float& z) {
    const float d0 = p[0] - cam[0], d1 = p[1] - cam[1], d2 = p[2] - cam[2];
    z = d0 * h->fwd[0] + d1 * h->fwd[1] + d2 * h->fwd[2];
    const float iz = 1.f / z;
    u = (d0 * h->right[0] + d1 * h->right[1] + d2 * h->right[2]) * iz;
    w = (d0 * h->up[0] + d1 * h->up[1] + d2 * h->up[2]) * iz;
}

__global__ void k_vis_project(VisArgs a) {
    int mn0 = 0x7fffffff, mn1 = 0x7fffffff, mx0 = (int)0x80000000, mx1 = (int)0x80000000;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.Vn; i += gridDim.x * 256) {
        float u, w, z;
        vis_project(a.hdr, a.cam, a.verts + 3 * (size_t)i, u, w, z);
        a.proj[3 * (size_t)i] = u; a.proj[3 * (size_t)i + 1] = w; a.proj[3 * (size_t)i + 2] = z;
        if (z > VIS_Z_EPS) {
            mn0 = min(mn0, f2ord(u)); mx0 = max(mx0, f2ord(u));
            mn1 = min(mn1, f2ord(w)); mx1 = max(mx1, f2ord(w));
        }
    }
    atomicMin(&a.hdr->bmin[0], mn0); atomicMin(&a.hdr->bmin[1], mn1);
    atomicMax(&a.hdr->bmax[0], mx0); atomicMax(&a.hdr->bmax[1], mx1);
}

__global__ void k_vis_grid(VisArgs a) {
    if (threadIdx.x || blockIdx.x) return;
    VisHeader* h = a.hdr;
    float u0 = ord2f(h->bmin[0]), u1 = ord2f(h->bmax[0]), w0 = ord2f(h->bmin[1]), w1 = ord2f(h->bmax[1]);
    if (!(u1 >= u0) || !(w1 >= w0)) { u0 = w0 = 0.f; u1 = w1 = 1.f; }
    const float du = fmaxf(u1 - u0, 1e-12f), dw = fmaxf(w1 - w0, 1e-12f);
    h->u0 = u0; h->w0 = w0;
    h->inv_cell_u = (float)a.G / (du * 1.0001f);
    h->inv_cell_w = (float)a.G / (dw * 1.0001f);
}

__device__ __forceinline__ int vis_cell(float x, float x0, float inv, int G) {
    const int c = (int)floorf((x - x0) * inv);
    return c < 0 ? 0 : (c >= G ? G - 1 : c);
}

// COUNT = true: histogram; false: scatter ids (same traversal, so the two agree exactly)
template <bool COUNT>
__global__ void k_vis_bin(VisArgs a) {
    const VisHeader* h = a.hdr;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < a.F; f += gridDim.x * 256) {
        const int64_t i0 = a.faces[3 * (size_t)f], i1 = a.faces[3 * (size_t)f + 1], i2 = a.faces[3 * (size_t)f + 2];
        const float* p0 = a.proj + 3 * i0; const float* p1 = a.proj + 3 * i1; const float* p2 = a.proj + 3 * i2;
        if (!(p0[2] > VIS_Z_EPS && p1[2] > VIS_Z_EPS && p2[2] > VIS_Z_EPS)) {
            if (COUNT) a.global_ids[atomicAdd(&a.hdr->n_global, 1u)] = (unsigned)f;
            continue;
        }
        const int cx0 = vis_cell(fminf(p0[0], fminf(p1[0], p2[0])), h->u0, h->inv_cell_u, a.G);
        const int cx1 = vis_cell(fmaxf(p0[0], fmaxf(p1[0], p2[0])), h->u0, h->inv_cell_u, a.G);
        const int cy0 = vis_cell(fminf(p0[1], fminf(p1[1], p2[1])), h->w0, h->inv_cell_w, a.G);
        const int cy1 = vis_cell(fmaxf(p0[1], fmaxf(p1[1], p2[1])), h->w0, h->inv_cell_w, a.G);
        for (int y = cy0; y <= cy1; ++y)
            for (int x = cx0; x <= cx1; ++x) {
                const int c = y * a.G + x;
                if (COUNT) atomicAdd(&a.cell_count[c], 1u);
                else {
                    const unsigned pos = a.cell_offset[c] + atomicAdd(&a.cell_cursor[c], 1u);
                    if (pos < a.cap) a.tri_ids[pos] = (unsigned)f;
                }
            }
    }
}

__global__ __launch_bounds__(1024) void k_vis_scan(VisArgs a) {
    const int n = a.G * a.G, tid = threadIdx.x;
    const int per = (n + 1023) / 1024, t0 = tid * per;
    unsigned local = 0;
    for (int i = 0; i < per; ++i) if (t0 + i < n) local += a.cell_count[t0 + i];
    const int lane = tid & 63, wave = tid >> 6;
    unsigned x = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned y = __shfl_up(x, d); if (lane >= d) x += y; }
    __shared__ unsigned wsum[16];
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) base += wsum[w]; total += wsum[w]; }
    unsigned run = base + x - local;
    for (int i = 0; i < per; ++i) if (t0 + i < n) { a.cell_offset[t0 + i] = run; run += a.cell_count[t0 + i]; }
    if (tid == 0) { a.cell_offset[n] = total; a.hdr->total = total; if (total > a.cap) a.hdr->overflow = 1; }
}

// Moller-Trumbore, both faces, t > 0; returns t or -1.
__device__ __forceinline__ float ray_tri(const float* o, const float* d, const float* v0, const float* v1, const float* v2) {
    const float e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (det == 0.f) return -1.f;
    const float inv = 1.f / det;
    const float tv[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
    const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (u < 0.f || u > 1.f) return -1.f;
    const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) * inv;
    if (v < 0.f || u + v > 1.f) return -1.f;
    const float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    return t > 0.f ? t : -1.f;
}

__device__ __forceinline__ void vis_test(const VisArgs& a, unsigned f, const float* o, const float* d, float& best_t, int& best_f) {
    const int64_t i0 = a.faces[3 * (size_t)f], i1 = a.faces[3 * (size_t)f + 1], i2 = a.faces[3 * (size_t)f + 2];
    const float t = ray_tri(o, d, a.verts + 3 * i0, a.verts + 3 * i1, a.verts + 3 * i2);
    if (t > 0.f && (t < best_t || (t == best_t && (int)f < best_f))) { best_t = t; best_f = (int)f; }
}

// BRUTE: test every triangle (fallback when the grid lists overflowed `cap`)
template <bool BRUTE>
__global__ void k_vis_rays(VisArgs a) {
    const VisHeader* h = a.hdr;
    if (!BRUTE && h->overflow) return;
    if (BRUTE && !h->overflow) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.P; i += gridDim.x * 256) {
        const float o[3] = {a.cam[0], a.cam[1], a.cam[2]};
        const float* tg = a.targets + 3 * (size_t)i;
        float d[3] = {tg[0] - o[0], tg[1] - o[1], tg[2] - o[2]};
        const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] /= n; d[1] /= n; d[2] /= n;
        float best_t = 3.402823466e+38f;
        int best_f = -1;
        if (BRUTE) {
            for (int f = 0; f < a.F; ++f) vis_test(a, (unsigned)f, o, d, best_t, best_f);
        } else {
            float u, w, z;
            vis_project(h, a.cam, tg, u, w, z);
            if (z > VIS_Z_EPS) {
                const int c = vis_cell(w, h->w0, h->inv_cell_w, a.G) * a.G + vis_cell(u, h->u0, h->inv_cell_u, a.G);
                const unsigned b0 = a.cell_offset[c], b1 = a.cell_offset[c + 1];
                for (unsigned k = b0; k < b1; ++k) vis_test(a, a.tri_ids[k], o, d, best_t, best_f);
                for (unsigned k = 0; k < h->n_global; ++k) vis_test(a, a.global_ids[k], o, d, best_t, best_f);
            } else {                              // target behind the projection plane: no grid cell, test everything
                for (int f = 0; f < a.F; ++f) vis_test(a, (unsigned)f, o, d, best_t, best_f);
            }
        }
        if (a.first_hit) a.first_hit[i] = best_f;
        a.mask[i] = (best_f >= 0 && (int64_t)best_f == a.binding[i]) ? 1 : 0;
    }
}

struct VisLayout { size_t hdr, proj, count, cursor, offset, global_ids, tri_ids, total; };
VisLayout vis_layout(int F, int Vn, int G, size_t cap) {
    VisLayout L; size_t o = 0;
    L.hdr = o; o += ggs_align(sizeof(VisHeader));
    L.count = o; o += ggs_align((size_t)G * G * 4);
    L.cursor = o; o += ggs_align((size_t)G * G * 4);
    L.proj = o; o += ggs_align((size_t)Vn * 12);
    L.offset = o; o += ggs_align(((size_t)G * G + 1) * 4);
    L.global_ids = o; o += ggs_align((size_t)F * 4);
    L.tri_ids = o; o += ggs_align(cap * 4);
    L.total = o;
    return L;
}
int vis_grid_res(int F) {
    int g = (int)sqrtf((float)F * 0.5f);
    return g < 32 ? 32 : (g > 1024 ? 1024 : g);
}

}  // namespace

extern "C" {

size_t ggs_visibility_scratch_bytes(int F, int n_verts, size_t ids_capacity) {
    if (F < 0 || n_verts < 0) return 0;
    return vis_layout(F, n_verts, vis_grid_res(F), ids_capacity).total;
}

int ggs_visibility(int P, int F, int n_verts, const float* verts, const int64_t* faces, const float* cam,
                   const float* targets, const int64_t* binding, void* scratch, size_t ids_capacity,
                   unsigned char* mask, int* first_hit, void* stream) {
    ggs_clear_error_();
    if (P < 0 || F < 0 || n_verts < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_visibility: bad sizes");
    if (P == 0) return GGS_OK;
    if (!cam || !targets || !binding || !mask || !scratch || (F > 0 && (!verts || !faces)))
        return ggs_fail_(GGS_ERR_ARG, "ggs_visibility: NULL pointer argument");
    if (ids_capacity > 0xffffffffu) return ggs_fail_(GGS_ERR_SIZE, "ggs_visibility: ids_capacity too large");
    hipStream_t s = (hipStream_t)stream;
    const int G = vis_grid_res(F);
    const VisLayout L = vis_layout(F, n_verts, G, ids_capacity);
    char* b = (char*)scratch;
    VisArgs a;
    a.P = P; a.F = F; a.Vn = n_verts; a.G = G; a.cap = (unsigned)ids_capacity;
    a.verts = verts; a.faces = faces; a.cam = cam; a.targets = targets; a.binding = binding;
    a.hdr = (VisHeader*)(b + L.hdr); a.proj = (float*)(b + L.proj); a.cell_count = (unsigned*)(b + L.count);
    a.cell_cursor = (unsigned*)(b + L.cursor); a.cell_offset = (unsigned*)(b + L.offset);
    a.global_ids = (unsigned*)(b + L.global_ids); a.tri_ids = (unsigned*)(b + L.tri_ids);
    a.mask = mask; a.first_hit = first_hit;
    if (ggs_zero_async(scratch, L.proj, s) != hipSuccess)             // header + count + cursor
        return ggs_fail_(GGS_ERR_HIP, "ggs_visibility: clearing the grid counters failed");
    const int gv = n_verts > 0 ? (n_verts + 255) / 256 : 1, gf = F > 0 ? (F + 255) / 256 : 1;
    if (n_verts > 0 && F > 0) {
        hipLaunchKernelGGL(k_vis_sum, dim3(gv < 512 ? gv : 512), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_vis_frame, dim3(1), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_vis_project, dim3(gv < 1024 ? gv : 1024), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_vis_grid, dim3(1), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_vis_bin<true>, dim3(gf < 2048 ? gf : 2048), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_vis_scan, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(k_vis_bin<false>, dim3(gf < 2048 ? gf : 2048), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_vis_scan, dim3(1), dim3(1024), 0, s, a);
    }
    const int gp = (P + 255) / 256;
    hipLaunchKernelGGL(k_vis_rays<false>, dim3(gp < 4096 ? gp : 4096), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_vis_rays<true>, dim3(gp < 4096 ? gp : 4096), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "ggs_visibility launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

}  // extern "C"
