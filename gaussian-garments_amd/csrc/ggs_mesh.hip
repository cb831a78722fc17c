// ggs_mesh.hip -- fused mesh binding: garment-mesh vertices -> per-face frame -> world-space
// Gaussian position / scale / rotation, and its backward down to the vertices.
//
// Replaces ~25 small PyTorch kernels + the roma quaternion chain per optimisation step:
//   MeshGaussianModel.update_face_coor      scene/mesh_gaussian_model.py:90-95
//   compute_face_orientation                utils/graphics_utils.py:118-137 (safe_normalize eps 1e-20)
//   get_xyz / get_scaling / get_rotation    scene/mesh_gaussian_model.py:105-128
//   AvatarGaussianModel barycentric origin  scene/avatar_gaussian_model.py:140-159
// rotmat -> unit quaternion follows the branch-on-largest-of(diagonal, trace) construction
// roma.rotmat_to_unitquat documents (xyzw, then reordered to wxyz).
//
// One lane per Gaussian; HBM-bound (forward: reads 40 B params + 3 gathered vertices,
// writes 40 B; backward: + 40 B of incoming gradients, 9 float atomics into the vertex
// gradient).  Vertices / faces are tiny (V ~ 50k) and stay in L2.
#include "ggs_kernels.h"

namespace {

#define MESH_EPS 1e-20f

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// y = x / sqrt(max(x.x, eps)); returns the length used and whether the clamp was active.
__device__ __forceinline__ V3 safe_normalize(V3 x, float& len, bool& clamped) {
    const float d = dot(x, x);
    clamped = d < MESH_EPS;
    len = sqrtf(clamped ? MESH_EPS : d);
    return x * (1.f / len);
}
// vjp of safe_normalize: dx from dy.
__device__ __forceinline__ V3 safe_normalize_vjp(V3 y, float len, bool clamped, V3 dy) {
    if (clamped) return dy * (1.f / len);
    return (dy - y * dot(y, dy)) * (1.f / len);
}

struct Frame {
    V3 v0, v1, v2, e1, e2, a0, a1, a2, n, c;
    float l1, ln, lc, tdot, s;
    bool k1, kn, kc;
};

__device__ __forceinline__ void face_frame(const float* verts, const int64_t* faces, int64_t f, Frame& F, int64_t idx[3]) {
    idx[0] = faces[3 * f]; idx[1] = faces[3 * f + 1]; idx[2] = faces[3 * f + 2];
    F.v0 = ld3(verts + 3 * idx[0]); F.v1 = ld3(verts + 3 * idx[1]); F.v2 = ld3(verts + 3 * idx[2]);
    F.e1 = F.v1 - F.v0; F.e2 = F.v2 - F.v0;
    F.a0 = safe_normalize(F.e1, F.l1, F.k1);
    F.n = cross(F.a0, F.e2);
    F.a1 = safe_normalize(F.n, F.ln, F.kn);
    F.c = cross(F.a1, F.a0);
    V3 u2 = safe_normalize(F.c, F.lc, F.kc);
    F.a2 = u2 * -1.f;
    F.tdot = dot(F.a2, F.e2);
    F.s = (F.l1 + fabsf(F.tdot)) * 0.5f;
}

// Unnormalised quaternion (x,y,z,w) of a rotation matrix R[r][c]; `choice` = branch taken.
// The diagonal branches are instantiated per index: with run-time indices the 3x3 / 4-vectors live in scratch
// memory (120 B per thread) and the kernels, which are pure latency chains at 100k threads, run ~2x longer.
template <int I>
__device__ __forceinline__ void quat_diag_branch(const float R[3][3], float tr, float u[4]) {
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    u[I] = 1.f - tr + 2.f * R[I][I];
    u[J] = R[J][I] + R[I][J];
    u[K] = R[K][I] + R[I][K];
    u[3] = R[K][J] - R[J][K];
}
__device__ __forceinline__ void rotmat_to_quat_raw(const float R[3][3], float u[4], int& choice) {
    const float tr = R[0][0] + R[1][1] + R[2][2];
    choice = 0;
    float best = R[0][0];
    if (R[1][1] > best) { best = R[1][1]; choice = 1; }
    if (R[2][2] > best) { best = R[2][2]; choice = 2; }
    if (tr > best) { choice = 3; }
    if (choice == 3) {
        u[0] = R[2][1] - R[1][2]; u[1] = R[0][2] - R[2][0]; u[2] = R[1][0] - R[0][1]; u[3] = 1.f + tr;
    } else if (choice == 0) {
        quat_diag_branch<0>(R, tr, u);
    } else if (choice == 1) {
        quat_diag_branch<1>(R, tr, u);
    } else {
        quat_diag_branch<2>(R, tr, u);
    }
}
template <int CI>
__device__ __forceinline__ void quat_diag_branch_vjp(const float du[4], float dR[3][3]) {
    constexpr int CJ = (CI + 1) % 3, CK = (CJ + 1) % 3;
    dR[0][0] -= du[CI]; dR[1][1] -= du[CI]; dR[2][2] -= du[CI];
    dR[CI][CI] += 2.f * du[CI];
    dR[CJ][CI] += du[CJ]; dR[CI][CJ] += du[CJ];
    dR[CK][CI] += du[CK]; dR[CI][CK] += du[CK];
    dR[CK][CJ] += du[3]; dR[CJ][CK] -= du[3];
}

// Hamilton product of (w,x,y,z) quaternions.
__device__ __forceinline__ void qmul(const float a[4], const float b[4], float p[4]) {
    p[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    p[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    p[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    p[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

__device__ __forceinline__ float norm4(const float q[4]) { return sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); }

// F.normalize (eps 1e-12) forward and vjp.
__device__ __forceinline__ float normalize4(const float q[4], float y[4]) {
    float n = norm4(q);
    n = n < 1e-12f ? 1e-12f : n;
    for (int k = 0; k < 4; ++k) y[k] = q[k] / n;
    return n;
}
__device__ __forceinline__ void normalize4_vjp(const float y[4], float n, const float dy[4], float dq[4]) {
    const float d = y[0] * dy[0] + y[1] * dy[1] + y[2] * dy[2] + y[3] * dy[3];
    for (int k = 0; k < 4; ++k) dq[k] = (dy[k] - y[k] * d) / n;
}

struct MeshArgs {
    int P;
    const float* verts; const int64_t* faces; const int64_t* binding;
    const float *local_xyz, *log_scaling, *raw_rot, *bary;
    float *xyz, *scaling, *rotation;
    const float *dL_dxyz, *dL_dscaling, *dL_drotation;
    float *dL_dverts, *dL_dlocal, *dL_dlog_scaling, *dL_draw_rot;
};

// Shared forward pieces of the quaternion chain.
struct QuatChain { float u[4]; int choice; float un; float qf0[4]; float qf[4]; float qfn; float qr[4]; float qrn; float qw[4]; float rot[4]; float qwn; };

__device__ __forceinline__ void quat_chain(const Frame& F, const float raw[4], QuatChain& Q) {
    const float R[3][3] = {{F.a0.x, F.a1.x, F.a2.x}, {F.a0.y, F.a1.y, F.a2.y}, {F.a0.z, F.a1.z, F.a2.z}};
    rotmat_to_quat_raw(R, Q.u, Q.choice);
    Q.un = norm4(Q.u);
    // xyzw / |u|  ->  wxyz
    Q.qf0[0] = Q.u[3] / Q.un; Q.qf0[1] = Q.u[0] / Q.un; Q.qf0[2] = Q.u[1] / Q.un; Q.qf0[3] = Q.u[2] / Q.un;
    Q.qfn = normalize4(Q.qf0, Q.qf);            // rotation_activation(face_orien_quat[binding])
    Q.qrn = normalize4(raw, Q.qr);              // rotation_activation(_rotation)
    qmul(Q.qf, Q.qr, Q.qw);
    Q.qwn = normalize4(Q.qw, Q.rot);
}

__device__ __forceinline__ void mesh_fwd_one(const MeshArgs& a, int i) {
    Frame F;
    int64_t idx[3];
    face_frame(a.verts, a.faces, a.binding[i], F, idx);
    const V3 l = ld3(a.local_xyz + 3 * (size_t)i);
    V3 origin;
    if (a.bary) {
        const float* b = a.bary + 3 * (size_t)i;
        origin = F.v0 * b[0] + F.v1 * b[1] + F.v2 * b[2];
    } else {
        origin = (F.v0 + F.v1 + F.v2) * (1.f / 3.f);
    }
    const V3 w = F.a0 * l.x + F.a1 * l.y + F.a2 * l.z;
    const V3 p = w * F.s + origin;
    a.xyz[3 * (size_t)i] = p.x; a.xyz[3 * (size_t)i + 1] = p.y; a.xyz[3 * (size_t)i + 2] = p.z;
    for (int k = 0; k < 3; ++k) a.scaling[3 * (size_t)i + k] = expf(a.log_scaling[3 * (size_t)i + k]) * F.s;
    const float4 r4 = reinterpret_cast<const float4*>(a.raw_rot)[i];
    const float raw[4] = {r4.x, r4.y, r4.z, r4.w};
    QuatChain Q;
    quat_chain(F, raw, Q);
    reinterpret_cast<float4*>(a.rotation)[i] = make_float4(Q.rot[0], Q.rot[1], Q.rot[2], Q.rot[3]);
}
__global__ __launch_bounds__(256) void k_mesh_fwd(MeshArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.P) mesh_fwd_one(a, i);
}

// ggs_step_prologue: the jobs at the head of an optimisation step that depend on nothing computed in it -- zero fills, the
// per-iteration parameter block, the opacity activation, the mesh binding -- as ONE launch.  A graph-replayed s2 iteration
// spent ~45 us (of 415) in eight such launches of 1-5 us of work each: a dependent launch costs ~5 us whatever it does.
struct PrologueArgs {
    MeshArgs mesh;                                   // P = 0: no binding
    int n_op; const float* op_logit; float* op;
    int n_clear; uint32_t* clr[GGS_PROLOGUE_MAX_CLEAR]; size_t clr_words[GGS_PROLOGUE_MAX_CLEAR];
    const uint32_t* copy_src; uint32_t* copy_dst; int copy_words;
};
__global__ __launch_bounds__(256) void k_step_prologue(PrologueArgs a) {
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (blockIdx.x == gridDim.x - 1)
        for (int i = threadIdx.x; i < a.copy_words; i += 256) a.copy_dst[i] = a.copy_src[i];
    for (int c = 0; c < a.n_clear; ++c) {
        uint32_t* p = a.clr[c];
        const size_t n = a.clr_words[c];
        size_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4;     // words in front of the 16-byte boundary
        head = head < n ? head : n;
        const size_t n_vec = (n - head) / 4, tail0 = head + n_vec * 4;
        uint4* v = reinterpret_cast<uint4*>(p + head);
        for (size_t k = gi; k < n_vec; k += stride) v[k] = make_uint4(0, 0, 0, 0);
        if (gi < head) p[gi] = 0;
        if (gi < n - tail0) p[tail0 + gi] = 0;
    }
    if (gi < (size_t)a.n_op) a.op[gi] = 1.f / (1.f + expf(-a.op_logit[gi]));        // torch.sigmoid's expression
    if (gi < (size_t)a.mesh.P) mesh_fwd_one(a.mesh, (int)gi);
}

// Vertex gradients: the 256 Gaussians of a workgroup are neighbours on the mesh, so the vertices they touch usually span a
// short index range.  Their 9 contributions each are summed in an LDS window over that range (lds_add_f32 below) and the window is
// flushed with ONE global atomic per touched component: 100k Gaussians on a 50k-vertex grid send ~0.9 M global atomics with
// ~6 writers per address otherwise, which was this kernel's whole duration (30 us; the arithmetic is ~3 us).  A range
// wider than the window (unstructured bindings) falls back to the direct atomics.
#define MESH_WIN 2048            // vertices -> 24 KB of LDS
// LDS float add as a compare-and-swap loop: gfx950 executes ds_add_f32 one lane every ~3 cycles per CU (770 cycles per
// wave-instruction whatever the addresses; the integer ds_add_u32 takes 17), and a ds_cmpst loop measures 40-180 cycles per
// wave-instruction (tools/ubench/lds_atomics.hip, profiles/r04_lds_atomic_rates.md).
__device__ __forceinline__ void lds_add_f32(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = *u, assumed;
    do {
        assumed = old;
        old = atomicCAS(u, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
}
__global__ __launch_bounds__(256) void k_mesh_bwd(MeshArgs a) {
    __shared__ float s_win[MESH_WIN * 3];
    __shared__ int s_rng[2];
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const bool live = i0 < a.P;
    const int i = live ? i0 : a.P - 1;            // idle lanes shadow the last Gaussian and write nothing
    Frame F;
    int64_t idx[3];
    face_frame(a.verts, a.faces, a.binding[i], F, idx);
    if (threadIdx.x == 0) { s_rng[0] = 0x7fffffff; s_rng[1] = -1; }
    __syncthreads();
    {
        int lo = (int)min(idx[0], min(idx[1], idx[2])), hi = (int)max(idx[0], max(idx[1], idx[2]));
        if (!live) { lo = 0x7fffffff; hi = -1; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { lo = min(lo, __shfl_xor(lo, d)); hi = max(hi, __shfl_xor(hi, d)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_rng[0], lo); atomicMax(&s_rng[1], hi); }
    }
    __syncthreads();
    const int vbase = s_rng[0], vcount = s_rng[1] - s_rng[0] + 1;
    const bool windowed = vcount > 0 && vcount <= MESH_WIN;
    if (windowed)
        for (int j = threadIdx.x; j < 3 * vcount; j += 256) s_win[j] = 0.f;
    __syncthreads();
    const V3 l = ld3(a.local_xyz + 3 * (size_t)i);
    const V3 dxyz = a.dL_dxyz ? ld3(a.dL_dxyz + 3 * (size_t)i) : V3{0.f, 0.f, 0.f};
    const V3 dsc = a.dL_dscaling ? ld3(a.dL_dscaling + 3 * (size_t)i) : V3{0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.dL_drotation) {
        const float4 d4 = reinterpret_cast<const float4*>(a.dL_drotation)[i];
        drot[0] = d4.x; drot[1] = d4.y; drot[2] = d4.z; drot[3] = d4.w;
    }
    const float4 r4 = reinterpret_cast<const float4*>(a.raw_rot)[i];
    const float raw[4] = {r4.x, r4.y, r4.z, r4.w};
    QuatChain Q;
    quat_chain(F, raw, Q);

    // ---- rotation chain, reverse ----
    float dqw[4], dqf[4], dqr[4], draw[4], dqf0[4];
    normalize4_vjp(Q.rot, Q.qwn, drot, dqw);
    const float* A = Q.qf; const float* B = Q.qr;
    dqf[0] = dqw[0] * B[0] + dqw[1] * B[1] + dqw[2] * B[2] + dqw[3] * B[3];
    dqf[1] = -dqw[0] * B[1] + dqw[1] * B[0] - dqw[2] * B[3] + dqw[3] * B[2];
    dqf[2] = -dqw[0] * B[2] + dqw[1] * B[3] + dqw[2] * B[0] - dqw[3] * B[1];
    dqf[3] = -dqw[0] * B[3] - dqw[1] * B[2] + dqw[2] * B[1] + dqw[3] * B[0];
    dqr[0] = dqw[0] * A[0] + dqw[1] * A[1] + dqw[2] * A[2] + dqw[3] * A[3];
    dqr[1] = -dqw[0] * A[1] + dqw[1] * A[0] + dqw[2] * A[3] - dqw[3] * A[2];
    dqr[2] = -dqw[0] * A[2] - dqw[1] * A[3] + dqw[2] * A[0] + dqw[3] * A[1];
    dqr[3] = -dqw[0] * A[3] + dqw[1] * A[2] - dqw[2] * A[1] + dqw[3] * A[0];
    normalize4_vjp(Q.qr, Q.qrn, dqr, draw);
    if (live) reinterpret_cast<float4*>(a.dL_draw_rot)[i] = make_float4(draw[0], draw[1], draw[2], draw[3]);
    normalize4_vjp(Q.qf, Q.qfn, dqf, dqf0);
    // qf0 (wxyz) = u (xyzw) / |u|
    const float dq_xyzw[4] = {dqf0[1], dqf0[2], dqf0[3], dqf0[0]};
    const float y_xyzw[4] = {Q.qf0[1], Q.qf0[2], Q.qf0[3], Q.qf0[0]};
    float du[4];
    normalize4_vjp(y_xyzw, Q.un, dq_xyzw, du);
    float dR[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (Q.choice == 3) {
        dR[2][1] += du[0]; dR[1][2] -= du[0];
        dR[0][2] += du[1]; dR[2][0] -= du[1];
        dR[1][0] += du[2]; dR[0][1] -= du[2];
        dR[0][0] += du[3]; dR[1][1] += du[3]; dR[2][2] += du[3];
    } else if (Q.choice == 0) {
        quat_diag_branch_vjp<0>(du, dR);
    } else if (Q.choice == 1) {
        quat_diag_branch_vjp<1>(du, dR);
    } else {
        quat_diag_branch_vjp<2>(du, dR);
    }
    V3 da0 = {dR[0][0], dR[1][0], dR[2][0]};
    V3 da1 = {dR[0][1], dR[1][1], dR[2][1]};
    V3 da2 = {dR[0][2], dR[1][2], dR[2][2]};

    // ---- position / scale ----
    const V3 w = F.a0 * l.x + F.a1 * l.y + F.a2 * l.z;
    const V3 dw = dxyz * F.s;
    float ds = dot(dxyz, w);
    da0 = da0 + dw * l.x; da1 = da1 + dw * l.y; da2 = da2 + dw * l.z;
    if (live) {
        a.dL_dlocal[3 * (size_t)i] = dot(dw, F.a0);
        a.dL_dlocal[3 * (size_t)i + 1] = dot(dw, F.a1);
        a.dL_dlocal[3 * (size_t)i + 2] = dot(dw, F.a2);
    }
    const float dscv[3] = {dsc.x, dsc.y, dsc.z};
    for (int k = 0; k < 3; ++k) {
        const float ex = expf(a.log_scaling[3 * (size_t)i + k]);
        if (live) a.dL_dlog_scaling[3 * (size_t)i + k] = dscv[k] * ex * F.s;
        ds += dscv[k] * ex;
    }
    // s = (l1 + |a2.e2|) / 2
    float dl1 = 0.5f * ds;
    const float sg = F.tdot > 0.f ? 1.f : (F.tdot < 0.f ? -1.f : 0.f);
    const float dt = 0.5f * ds * sg;
    da2 = da2 + F.e2 * dt;
    V3 de2 = F.a2 * dt;
    // a2 = -normalize(c), c = a1 x a0
    const V3 u2 = F.a2 * -1.f;
    const V3 dc = safe_normalize_vjp(u2, F.lc, F.kc, da2 * -1.f);
    da1 = da1 + cross(F.a0, dc);
    da0 = da0 + cross(dc, F.a1);
    // a1 = normalize(n), n = a0 x e2
    const V3 dn = safe_normalize_vjp(F.a1, F.ln, F.kn, da1);
    da0 = da0 + cross(F.e2, dn);
    de2 = de2 + cross(dn, F.a0);
    // a0 = normalize(e1), l1 = |e1|
    V3 de1 = safe_normalize_vjp(F.a0, F.l1, F.k1, da0);
    if (!F.k1) de1 = de1 + F.a0 * dl1;
    // vertices
    V3 g0 = (de1 + de2) * -1.f, g1 = de1, g2 = de2;
    if (a.bary) {
        const float* b = a.bary + 3 * (size_t)i;
        g0 = g0 + dxyz * b[0]; g1 = g1 + dxyz * b[1]; g2 = g2 + dxyz * b[2];
    } else {
        const V3 t = dxyz * (1.f / 3.f);
        g0 = g0 + t; g1 = g1 + t; g2 = g2 + t;
    }
    if (windowed) {
        if (live) {
            float* o0 = s_win + 3 * (int)(idx[0] - vbase);
            float* o1 = s_win + 3 * (int)(idx[1] - vbase);
            float* o2 = s_win + 3 * (int)(idx[2] - vbase);
            lds_add_f32(o0, g0.x); lds_add_f32(o0 + 1, g0.y); lds_add_f32(o0 + 2, g0.z);
            lds_add_f32(o1, g1.x); lds_add_f32(o1 + 1, g1.y); lds_add_f32(o1 + 2, g1.z);
            lds_add_f32(o2, g2.x); lds_add_f32(o2 + 1, g2.y); lds_add_f32(o2 + 2, g2.z);
        }
        __syncthreads();
        float* out = a.dL_dverts + 3 * (size_t)vbase;
        for (int j = threadIdx.x; j < 3 * vcount; j += 256) {
            const float v = s_win[j];
            if (v != 0.f) atomicAdd(out + j, v);
        }
    } else if (live) {
        float* o0 = a.dL_dverts + 3 * idx[0];
        float* o1 = a.dL_dverts + 3 * idx[1];
        float* o2 = a.dL_dverts + 3 * idx[2];
        atomicAdd(o0, g0.x); atomicAdd(o0 + 1, g0.y); atomicAdd(o0 + 2, g0.z);
        atomicAdd(o1, g1.x); atomicAdd(o1 + 1, g1.y); atomicAdd(o1 + 2, g1.z);
        atomicAdd(o2, g2.x); atomicAdd(o2 + 1, g2.y); atomicAdd(o2 + 2, g2.z);
    }
}

}  // namespace

extern "C" {

int ggs_mesh_bind_forward(int P, int F, const float* verts, const int64_t* faces, const int64_t* binding,
                          const float* local_xyz, const float* log_scaling, const float* raw_rot,
                          const float* bary, float* xyz, float* scaling, float* rotation, void* stream) {
    ggs_clear_error_();
    if (P < 0 || F < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_mesh_bind_forward: bad sizes");
    if (P == 0) return GGS_OK;
    if (!verts || !faces || !binding || !local_xyz || !log_scaling || !raw_rot || !xyz || !scaling || !rotation) {
        return ggs_fail_(GGS_ERR_ARG, "ggs_mesh_bind_forward: NULL pointer argument");
    }
    MeshArgs a = {};
    a.P = P; a.verts = verts; a.faces = faces; a.binding = binding; a.local_xyz = local_xyz;
    a.log_scaling = log_scaling; a.raw_rot = raw_rot; a.bary = bary; a.xyz = xyz; a.scaling = scaling; a.rotation = rotation;
    hipLaunchKernelGGL(k_mesh_fwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "mesh_fwd launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

int ggs_step_prologue(const GgsStepPrologue* d, void* stream) {
    ggs_clear_error_();
    if (!d) return ggs_fail_(GGS_ERR_ARG, "ggs_step_prologue: NULL descriptor");
    if (d->n_clear < 0 || d->n_clear > GGS_PROLOGUE_MAX_CLEAR || d->P < 0 || d->F < 0 || d->n_opacity < 0)
        return ggs_fail_(GGS_ERR_ARG, "ggs_step_prologue: bad sizes");
    PrologueArgs a = {};
    size_t work = 1;
    for (int c = 0; c < d->n_clear; ++c) {
        if (!d->clear_ptr[c] || (reinterpret_cast<uintptr_t>(d->clear_ptr[c]) & 3) || (d->clear_bytes[c] & 3))
            return ggs_fail_(GGS_ERR_ARG, "ggs_step_prologue: clear range %d is NULL or not 4-byte aligned", c);
        a.clr[c] = static_cast<uint32_t*>(d->clear_ptr[c]);
        a.clr_words[c] = d->clear_bytes[c] / 4;
        work = work > d->clear_bytes[c] / 16 ? work : d->clear_bytes[c] / 16;
    }
    a.n_clear = d->n_clear;
    if (d->copy_bytes) {
        if (!d->copy_src || !d->copy_dst || d->copy_bytes > 1024 || (d->copy_bytes & 3) ||
            ((reinterpret_cast<uintptr_t>(d->copy_src) | reinterpret_cast<uintptr_t>(d->copy_dst)) & 3))
            return ggs_fail_(GGS_ERR_ARG, "ggs_step_prologue: the block copy needs two 4-byte aligned pointers and <= 1024 bytes");
        a.copy_src = static_cast<const uint32_t*>(d->copy_src); a.copy_dst = static_cast<uint32_t*>(d->copy_dst);
        a.copy_words = (int)(d->copy_bytes / 4);
    }
    if (d->n_opacity) {
        if (!d->opacity_logit || !d->opacity) return ggs_fail_(GGS_ERR_ARG, "ggs_step_prologue: NULL opacity pointer");
        a.n_op = d->n_opacity; a.op_logit = d->opacity_logit; a.op = d->opacity;
    }
    if (d->P) {
        if (!d->verts || !d->faces || !d->binding || !d->local_xyz || !d->log_scaling || !d->raw_rot || !d->xyz ||
            !d->scaling || !d->rotation)
            return ggs_fail_(GGS_ERR_ARG, "ggs_step_prologue: NULL mesh-binding pointer");
        a.mesh.P = d->P; a.mesh.verts = d->verts; a.mesh.faces = d->faces; a.mesh.binding = d->binding;
        a.mesh.local_xyz = d->local_xyz; a.mesh.log_scaling = d->log_scaling; a.mesh.raw_rot = d->raw_rot; a.mesh.bary = d->bary;
        a.mesh.xyz = d->xyz; a.mesh.scaling = d->scaling; a.mesh.rotation = d->rotation;
    }
    size_t blocks = (work + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    const size_t per_item = ((size_t)(d->P > d->n_opacity ? d->P : d->n_opacity) + 255) / 256;
    blocks = blocks > per_item ? blocks : per_item;
    blocks = blocks < 1 ? 1 : blocks;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_step_prologue, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "step_prologue launch failed: %s", hipGetErrorString(e));
    void* marked[GGS_PROLOGUE_MAX_CLEAR];
    for (int c = 0; c < d->n_clear; ++c) marked[c] = ((d->consumer_mask >> c) & 1u) ? d->clear_ptr[c] : nullptr;
    ggs_set_clear_marks_(d->n_clear, marked, d->clear_bytes, s);
    return GGS_OK;
}

int ggs_mesh_bind_backward(int P, int F, const float* verts, const int64_t* faces, const int64_t* binding,
                           const float* local_xyz, const float* log_scaling, const float* raw_rot,
                           const float* bary, const float* dL_dxyz, const float* dL_dscaling,
                           const float* dL_drotation, float* dL_dverts, float* dL_dlocal_xyz,
                           float* dL_dlog_scaling, float* dL_draw_rot, void* stream) {
    ggs_clear_error_();
    if (P < 0 || F < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_mesh_bind_backward: bad sizes");
    if (P == 0) return GGS_OK;
    if (!verts || !faces || !binding || !local_xyz || !log_scaling || !raw_rot || !dL_dverts || !dL_dlocal_xyz ||
        !dL_dlog_scaling || !dL_draw_rot) {
        return ggs_fail_(GGS_ERR_ARG, "ggs_mesh_bind_backward: NULL pointer argument");
    }
    MeshArgs a = {};
    a.P = P; a.verts = verts; a.faces = faces; a.binding = binding; a.local_xyz = local_xyz;
    a.log_scaling = log_scaling; a.raw_rot = raw_rot; a.bary = bary;
    a.dL_dxyz = dL_dxyz; a.dL_dscaling = dL_dscaling; a.dL_drotation = dL_drotation;
    a.dL_dverts = dL_dverts; a.dL_dlocal = dL_dlocal_xyz; a.dL_dlog_scaling = dL_dlog_scaling; a.dL_draw_rot = dL_draw_rot;
    hipLaunchKernelGGL(k_mesh_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "mesh_bwd launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

}  // extern "C"
