/*
 * splat_oracle.c -- CPU restatement (plain C, fp32) of the differentiable tile
 * rasterizer behind gaussian_renderer.render().
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker
 * (or as the timed CPU baseline).  The product (libggsplat.so) never links,
 * loads or calls it.
 *
 * PARITY UNPINNED.  The algorithm lives in a third-party CUDA extension that is
 * absent from /root/reference: diff_gaussian_rasterization_depth_alpha from
 * lizhe00/AnimatableGaussians (un-pinned clone, reference setup.sh:26-28).  The
 * reference holds no tests / golden vectors for it.  This file restates the
 * published algorithm (Kerbl et al. 2023 tile rasterizer + the depth/alpha
 * outputs of that fork; behavioural spec in SURVEY.md Appendix A) and is
 * anchored on the reference's call site gaussian_renderer/__init__.py:39-54 and
 * :103-111 (argument meaning, layouts, return order color/radii/depth/alpha).
 * The two stages that do exist in the reference tree -- SH evaluation
 * (utils/sh_utils.py:56-111) and scale/rotation -> cov3D
 * (utils/general_utils.py:91-120) -- are checked against golden vectors
 * generated from those files (tests/golden/).  The hand-derived backward below
 * is checked against autograd through oracle/torch_oracle.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  The
 * per-Gaussian stage is written so that every fp32 operation is an IEEE
 * correctly-rounded +,-,*,/,sqrt in a fixed order; the HIP preprocess kernel
 * follows the same order with contraction off, so radii / tile rectangles /
 * sort keys compare bit-exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16
#define NEAR_Z 0.2f
#define W_EPS 0.0000001f
#define FOV_CLAMP 1.3f
#define LOWPASS 0.3f
#define LAMBDA_FLOOR 0.1f
#define ALPHA_MAX 0.99f
#define ALPHA_MIN (1.0f / 255.0f)
#define T_MIN 0.0001f
#define DET_EPS 0.0000001f

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    int P, K, deg, W, H;
    float tanfovx, tanfovy, scale_modifier;
} ggo_params;

typedef struct {
    ggo_params prm;
    int gx, gy;
    /* per Gaussian */
    float *xy, *depth, *conic_op, *rgb, *cov3d;
    int *radii, *rect; /* rect: x0,y0,x1,y1 */
    uint8_t *clamped;
    /* binning */
    int64_t N;
    int64_t n_blended;   /* (Gaussian, pixel) pairs blended by the forward */
    int64_t *tile_start; /* gx*gy+1 */
    uint32_t *list;      /* sorted ids, N */
    /* per pixel */
    float *final_T;
    uint32_t *n_contrib;
} ggo_state;

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

static inline void xf43(const float* m, const float* p, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}

/* R (row-major, standard) from (w,x,y,z); not normalised (Appendix A.0). */
static inline void quat_R(const float* q, float* R) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = R S^2 R^T via M[k][i] = s_k R[i][k]; out (xx,xy,xz,yy,yz,zz) (A.1 step 3). */
static inline void cov3d_of(const float* scale, float mod, const float* q, float* c6) {
    float R[9], M[9];
    quat_R(q, R);
    for (int k = 0; k < 3; ++k) {
        float s = mod * scale[k];
        for (int i = 0; i < 3; ++i) M[k * 3 + i] = s * R[i * 3 + k];
    }
#define SIG(i, j) (M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j])
    c6[0] = SIG(0, 0); c6[1] = SIG(0, 1); c6[2] = SIG(0, 2);
    c6[3] = SIG(1, 1); c6[4] = SIG(1, 2); c6[5] = SIG(2, 2);
#undef SIG
}

/* SH basis value per coefficient (degree <= 3) for unit direction d.  Same
 * polynomials as utils/sh_utils.py:56-111. */
static inline void sh_basis(int deg, const float* d, float* b) {
    b[0] = SH_C0;
    if (deg > 0) {
        float x = d[0], y = d[1], z = d[2];
        b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.f * zz - xx - yy);
            b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.f * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

/* d(basis_k)/d(dir) for the backward. db[k*3 + axis]. */
static inline void sh_basis_grad(int deg, const float* d, float* db) {
    memset(db, 0, sizeof(float) * 48);
    if (deg > 0) {
        float x = d[0], y = d[1], z = d[2];
        db[1 * 3 + 1] = -SH_C1; db[2 * 3 + 2] = SH_C1; db[3 * 3 + 0] = -SH_C1;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            db[4 * 3 + 0] = SH_C2[0] * y; db[4 * 3 + 1] = SH_C2[0] * x;
            db[5 * 3 + 1] = SH_C2[1] * z; db[5 * 3 + 2] = SH_C2[1] * y;
            db[6 * 3 + 0] = SH_C2[2] * -2.f * x; db[6 * 3 + 1] = SH_C2[2] * -2.f * y; db[6 * 3 + 2] = SH_C2[2] * 4.f * z;
            db[7 * 3 + 0] = SH_C2[3] * z; db[7 * 3 + 2] = SH_C2[3] * x;
            db[8 * 3 + 0] = SH_C2[4] * 2.f * x; db[8 * 3 + 1] = SH_C2[4] * -2.f * y;
            if (deg > 2) {
                db[9 * 3 + 0] = SH_C3[0] * 6.f * x * y; db[9 * 3 + 1] = SH_C3[0] * (3.f * xx - 3.f * yy);
                db[10 * 3 + 0] = SH_C3[1] * y * z; db[10 * 3 + 1] = SH_C3[1] * x * z; db[10 * 3 + 2] = SH_C3[1] * x * y;
                db[11 * 3 + 0] = SH_C3[2] * -2.f * x * y; db[11 * 3 + 1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy);
                db[11 * 3 + 2] = SH_C3[2] * 8.f * y * z;
                db[12 * 3 + 0] = SH_C3[3] * -6.f * x * z; db[12 * 3 + 1] = SH_C3[3] * -6.f * y * z;
                db[12 * 3 + 2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                db[13 * 3 + 0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); db[13 * 3 + 1] = SH_C3[4] * -2.f * x * y;
                db[13 * 3 + 2] = SH_C3[4] * 8.f * x * z;
                db[14 * 3 + 0] = SH_C3[5] * 2.f * x * z; db[14 * 3 + 1] = SH_C3[5] * -2.f * y * z; db[14 * 3 + 2] = SH_C3[5] * (xx - yy);
                db[15 * 3 + 0] = SH_C3[6] * (3.f * xx - 3.f * yy); db[15 * 3 + 1] = SH_C3[6] * -6.f * x * y;
            }
        }
    }
}

/* Shared EWA pieces (A.1 step 4): clamped t, M = J * Wrot (2x3). */
typedef struct { float tx, ty, tz, fx, fy, xmul, ymul, M[6]; } ewa_t;

static inline void ewa_setup(const ggo_params* p, const float* view, const float* mean, ewa_t* e) {
    float t[3];
    xf43(view, mean, t);
    float limx = FOV_CLAMP * p->tanfovx, limy = FOV_CLAMP * p->tanfovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    e->xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e->ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    e->tx = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    e->ty = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    e->tz = t[2];
    e->fx = (float)p->W / (2.f * p->tanfovx);
    e->fy = (float)p->H / (2.f * p->tanfovy);
    float j00 = e->fx / e->tz, j02 = -(e->fx * e->tx) / (e->tz * e->tz);
    float j11 = e->fy / e->tz, j12 = -(e->fy * e->ty) / (e->tz * e->tz);
    /* Wrot row i = (view[i], view[4+i], view[8+i]) */
    for (int j = 0; j < 3; ++j) {
        e->M[j] = j00 * view[4 * j + 0] + j02 * view[4 * j + 2];
        e->M[3 + j] = j11 * view[4 * j + 1] + j12 * view[4 * j + 2];
    }
}

static inline void sym6_mul(const float* c6, const float* v, float* o) { /* Sigma * v */
    o[0] = c6[0] * v[0] + c6[1] * v[1] + c6[2] * v[2];
    o[1] = c6[1] * v[0] + c6[3] * v[1] + c6[4] * v[2];
    o[2] = c6[2] * v[0] + c6[4] * v[1] + c6[5] * v[2];
}

static inline void tile_rect(float px, float py, float r, int gx, int gy, int* rc) {
    int x0 = (int)((px - r) / (float)TILE), y0 = (int)((py - r) / (float)TILE);
    int x1 = (int)((px + r + (float)(TILE - 1)) / (float)TILE), y1 = (int)((py + r + (float)(TILE - 1)) / (float)TILE);
    rc[0] = x0 < 0 ? 0 : (x0 > gx ? gx : x0);
    rc[1] = y0 < 0 ? 0 : (y0 > gy ? gy : y0);
    rc[2] = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
    rc[3] = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
}

typedef struct { uint64_t key; } sort_item;
static int cmp_item(const void* a, const void* b) {
    uint64_t x = ((const sort_item*)a)->key, y = ((const sort_item*)b)->key;
    return (x > y) - (x < y);
}

void ggo_free(void* h) {
    ggo_state* s = (ggo_state*)h;
    if (!s) return;
    free(s->xy); free(s->depth); free(s->conic_op); free(s->rgb); free(s->cov3d);
    free(s->radii); free(s->rect); free(s->clamped); free(s->tile_start); free(s->list);
    free(s->final_T); free(s->n_contrib); free(s);
}

/* Forward (A.1).  shs [P,K,3] or NULL; colors [P,3] or NULL; (scales [P,3], rots [P,4]) or cov3d [P,6].
 * out_color [3,H,W], out_depth [H,W], out_alpha [H,W], radii [P].  Returns a state handle for ggo_backward. */
void* ggo_forward(const ggo_params* prm, const float* bg, const float* means3D, const float* shs,
                  const float* colors, const float* opac, const float* scales, const float* rots,
                  const float* cov3d_pre, const float* view, const float* proj, const float* campos,
                  float* out_color, float* out_depth, float* out_alpha, int* radii) {
    ggo_state* s = (ggo_state*)calloc(1, sizeof(ggo_state));
    s->prm = *prm;
    const int P = prm->P, W = prm->W, H = prm->H, K = prm->K;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, T = gx * gy;
    s->gx = gx; s->gy = gy;
    s->xy = (float*)calloc((size_t)P * 2 + 1, 4); s->depth = (float*)calloc((size_t)P + 1, 4);
    s->conic_op = (float*)calloc((size_t)P * 4 + 1, 4); s->rgb = (float*)calloc((size_t)P * 3 + 1, 4);
    s->cov3d = (float*)calloc((size_t)P * 6 + 1, 4); s->radii = (int*)calloc((size_t)P + 1, 4);
    s->rect = (int*)calloc((size_t)P * 4 + 1, 4); s->clamped = (uint8_t*)calloc((size_t)P * 3 + 1, 1);
    s->tile_start = (int64_t*)calloc((size_t)T + 1, 8);
    s->final_T = (float*)calloc((size_t)W * H + 1, 4); s->n_contrib = (uint32_t*)calloc((size_t)W * H + 1, 4);
    int64_t* tcount = (int64_t*)calloc((size_t)T + 1, 8);

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        radii[i] = 0;
        const float* m = means3D + 3 * (size_t)i;
        float pv[3];
        xf43(view, m, pv);
        if (pv[2] <= NEAR_Z) continue;
        float hx = proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12];
        float hy = proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13];
        float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
        float pw = 1.0f / (hw + W_EPS);
        float ndx = hx * pw, ndy = hy * pw;
        float* c6 = s->cov3d + 6 * (size_t)i;
        if (cov3d_pre) memcpy(c6, cov3d_pre + 6 * (size_t)i, 24);
        else cov3d_of(scales + 3 * (size_t)i, prm->scale_modifier, rots + 4 * (size_t)i, c6);
        ewa_t e;
        ewa_setup(prm, view, m, &e);
        float s0[3], s1[3];
        sym6_mul(c6, e.M, s0);
        sym6_mul(c6, e.M + 3, s1);
        float a = (e.M[0] * s0[0] + e.M[1] * s0[1] + e.M[2] * s0[2]) + LOWPASS;
        float b = e.M[0] * s1[0] + e.M[1] * s1[1] + e.M[2] * s1[2];
        float c = (e.M[3] * s1[0] + e.M[4] * s1[1] + e.M[5] * s1[2]) + LOWPASS;
        float det = a * c - b * b;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float mid = 0.5f * (a + c);
        float root = sqrtf(fmaxf_(LAMBDA_FLOOR, mid * mid - det));
        float l1 = mid + root, l2 = mid - root;
        float rad = ceilf(3.f * sqrtf(fmaxf_(l1, l2)));
        float px = ((ndx + 1.0f) * (float)W - 1.0f) * 0.5f;
        float py = ((ndy + 1.0f) * (float)H - 1.0f) * 0.5f;
        int rc[4];
        tile_rect(px, py, rad, gx, gy, rc);
        if ((rc[2] - rc[0]) * (rc[3] - rc[1]) == 0) continue;
        float* rgb = s->rgb + 3 * (size_t)i;
        if (colors) { rgb[0] = colors[3 * (size_t)i]; rgb[1] = colors[3 * (size_t)i + 1]; rgb[2] = colors[3 * (size_t)i + 2]; }
        else {
            float d[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
            float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] = d[0] / len; d[1] = d[1] / len; d[2] = d[2] / len;
            float bas[16];
            sh_basis(prm->deg, d, bas);
            int nk = (prm->deg + 1) * (prm->deg + 1);
            const float* sh = shs + (size_t)i * K * 3;
            for (int ch = 0; ch < 3; ++ch) {
                float r = 0.f;
                for (int k = 0; k < nk; ++k) r = r + bas[k] * sh[k * 3 + ch];
                r = r + 0.5f;
                s->clamped[3 * (size_t)i + ch] = r < 0.f;
                rgb[ch] = r < 0.f ? 0.f : r;
            }
        }
        s->depth[i] = pv[2];
        s->radii[i] = radii[i] = (int)rad;
        s->xy[2 * (size_t)i] = px; s->xy[2 * (size_t)i + 1] = py;
        float* co = s->conic_op + 4 * (size_t)i;
        co[0] = c * det_inv; co[1] = -b * det_inv; co[2] = a * det_inv; co[3] = opac[i];
        memcpy(s->rect + 4 * (size_t)i, rc, 16);
    }
    /* binning (A.1 step 9): per tile, ascending (fp32 depth bits, Gaussian index) */
    for (int i = 0; i < P; ++i) {
        if (s->radii[i] <= 0) continue;
        const int* rc = s->rect + 4 * (size_t)i;
        for (int y = rc[1]; y < rc[3]; ++y) for (int x = rc[0]; x < rc[2]; ++x) tcount[y * gx + x]++;
    }
    for (int t = 0; t < T; ++t) s->tile_start[t + 1] = s->tile_start[t] + tcount[t];
    s->N = s->tile_start[T];
    sort_item* items = (sort_item*)malloc(sizeof(sort_item) * (size_t)(s->N + 1));
    memset(tcount, 0, sizeof(int64_t) * (size_t)T);
    for (int i = 0; i < P; ++i) {
        if (s->radii[i] <= 0) continue;
        const int* rc = s->rect + 4 * (size_t)i;
        uint32_t dbits; memcpy(&dbits, &s->depth[i], 4);
        for (int y = rc[1]; y < rc[3]; ++y) for (int x = rc[0]; x < rc[2]; ++x) {
            int t = y * gx + x;
            items[s->tile_start[t] + tcount[t]++].key = ((uint64_t)dbits << 32) | (uint32_t)i;
        }
    }
    s->list = (uint32_t*)malloc(4 * (size_t)(s->N + 1));
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < T; ++t) {
        int64_t b0 = s->tile_start[t], b1 = s->tile_start[t + 1];
        if (b1 - b0 > 1) qsort(items + b0, (size_t)(b1 - b0), sizeof(sort_item), cmp_item);
        for (int64_t k = b0; k < b1; ++k) s->list[k] = (uint32_t)(items[k].key & 0xffffffffu);
    }
    free(items); free(tcount);

    /* compositing (A.1 step 10) */
    int64_t n_blended = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_blended)
    for (int t = 0; t < T; ++t) {
        int tx = t % gx, ty = t / gx;
        int64_t b0 = s->tile_start[t], b1 = s->tile_start[t + 1];
        for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
            for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                float Tr = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
                uint32_t contributor = 0, last = 0;
                float pxf = (float)px, pyf = (float)py;
                for (int64_t k = b0; k < b1; ++k) {
                    contributor++;
                    uint32_t id = s->list[k];
                    float dx = s->xy[2 * (size_t)id] - pxf, dy = s->xy[2 * (size_t)id + 1] - pyf;
                    const float* co = s->conic_op + 4 * (size_t)id;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.f) continue;
                    float alpha = fminf_(ALPHA_MAX, co[3] * expf(power));
                    if (alpha < ALPHA_MIN) continue;
                    float test_T = Tr * (1.f - alpha);
                    if (test_T < T_MIN) break;
                    float w = alpha * Tr;
                    const float* rgb = s->rgb + 3 * (size_t)id;
                    C0 += rgb[0] * w; C1 += rgb[1] * w; C2 += rgb[2] * w;
                    D += s->depth[id] * w; A += w;
                    Tr = test_T;
                    last = contributor;
                    n_blended++;
                }
                size_t pix = (size_t)py * W + px;
                s->final_T[pix] = Tr; s->n_contrib[pix] = last;
                out_color[pix] = C0 + Tr * bg[0];
                out_color[(size_t)H * W + pix] = C1 + Tr * bg[1];
                out_color[2 * (size_t)H * W + pix] = C2 + Tr * bg[2];
                out_depth[pix] = D; out_alpha[pix] = A;
            }
    }
    s->n_blended = n_blended;
    return s;
}

int64_t ggo_num_rendered(const void* h) { return ((const ggo_state*)h)->N; }
int64_t ggo_num_blended(const void* h) { return ((const ggo_state*)h)->n_blended; }

/* Copy internals out for tests.  Any pointer may be NULL. */
void ggo_get_internals(const void* h, float* xy, float* depth, float* conic_op, float* rgb, float* cov3d,
                       int64_t* tile_start, uint32_t* list, float* final_T, uint32_t* n_contrib) {
    const ggo_state* s = (const ggo_state*)h;
    size_t P = (size_t)s->prm.P, HW = (size_t)s->prm.W * s->prm.H, T = (size_t)s->gx * s->gy;
    if (xy) memcpy(xy, s->xy, P * 8);
    if (depth) memcpy(depth, s->depth, P * 4);
    if (conic_op) memcpy(conic_op, s->conic_op, P * 16);
    if (rgb) memcpy(rgb, s->rgb, P * 12);
    if (cov3d) memcpy(cov3d, s->cov3d, P * 24);
    if (tile_start) memcpy(tile_start, s->tile_start, (T + 1) * 8);
    if (list) memcpy(list, s->list, (size_t)s->N * 4);
    if (final_T) memcpy(final_T, s->final_T, HW * 4);
    if (n_contrib) memcpy(n_contrib, s->n_contrib, HW * 4);
}

static inline void atomic_addf(float* p, float v) {
#pragma omp atomic
    *p += v;
}

/* Backward (A.2).  dL_ddepth / dL_dalpha may be NULL (treated as zero).  Outputs are overwritten.
 * dL_dmeans2D [P,3] is w.r.t. NDC xy (pixel gradient * 0.5W / 0.5H), z = 0. */
void ggo_backward(const void* h, const float* bg, const float* means3D, const float* shs, const float* colors,
                  const float* scales, const float* rots, const float* cov3d_pre, const float* view,
                  const float* proj, const float* campos, const float* dL_dcolor, const float* dL_ddepth,
                  const float* dL_dalpha_img, float* dL_dmeans2D, float* dL_dcolors, float* dL_dopac,
                  float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots) {
    const ggo_state* s = (const ggo_state*)h;
    const ggo_params* prm = &s->prm;
    const int P = prm->P, W = prm->W, H = prm->H, K = prm->K, gx = s->gx, gy = s->gy, T = gx * gy;
    const size_t HW = (size_t)H * W;
    float* g_xy = (float*)calloc((size_t)P * 2 + 1, 4);   /* pixel units */
    float* g_con = (float*)calloc((size_t)P * 3 + 1, 4);  /* true d/d(conic.x,y,z) */
    float* g_dep = (float*)calloc((size_t)P + 1, 4);
    memset(dL_dcolors, 0, (size_t)P * 12);
    memset(dL_dopac, 0, (size_t)P * 4);

#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < T; ++t) {
        int tx = t % gx, ty = t / gx;
        int64_t b0 = s->tile_start[t];
        for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
            for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
                size_t pix = (size_t)py * W + px;
                const float Tf = s->final_T[pix];
                float Tr = Tf;
                const float dC[3] = {dL_dcolor[pix], dL_dcolor[HW + pix], dL_dcolor[2 * HW + pix]};
                const float dD = dL_ddepth ? dL_ddepth[pix] : 0.f;
                const float dA = dL_dalpha_img ? dL_dalpha_img[pix] : 0.f;
                const float bgdot = bg[0] * dC[0] + bg[1] * dC[1] + bg[2] * dC[2];
                float rec[3] = {0.f, 0.f, 0.f}, recD = 0.f, recA = 0.f;
                float last_a = 0.f, last_c[3] = {0.f, 0.f, 0.f}, last_d = 0.f;
                float pxf = (float)px, pyf = (float)py;
                for (int64_t k = b0 + (int64_t)s->n_contrib[pix] - 1; k >= b0; --k) {
                    uint32_t id = s->list[k];
                    float dx = s->xy[2 * (size_t)id] - pxf, dy = s->xy[2 * (size_t)id + 1] - pyf;
                    const float* co = s->conic_op + 4 * (size_t)id;
                    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.f) continue;
                    float G = expf(power);
                    float alpha = fminf_(ALPHA_MAX, co[3] * G);
                    if (alpha < ALPHA_MIN) continue;
                    Tr = Tr / (1.f - alpha);
                    float w = alpha * Tr;
                    const float* rgb = s->rgb + 3 * (size_t)id;
                    float dL_da = 0.f;
                    for (int ch = 0; ch < 3; ++ch) {
                        rec[ch] = last_a * last_c[ch] + (1.f - last_a) * rec[ch];
                        last_c[ch] = rgb[ch];
                        dL_da += (rgb[ch] - rec[ch]) * dC[ch];
                        atomic_addf(&dL_dcolors[3 * (size_t)id + ch], w * dC[ch]);
                    }
                    recD = last_a * last_d + (1.f - last_a) * recD;
                    last_d = s->depth[id];
                    dL_da += (last_d - recD) * dD;
                    recA = last_a + (1.f - last_a) * recA;
                    dL_da += (1.f - recA) * dA;
                    if (dD != 0.f) atomic_addf(&g_dep[id], w * dD);
                    dL_da *= Tr;
                    last_a = alpha;
                    dL_da += (-Tf / (1.f - alpha)) * bgdot;
                    /* straight-through the 0.99 clamp, like the upstream kernel */
                    float dL_dG = co[3] * dL_da;
                    float gdx = G * dx, gdy = G * dy;
                    atomic_addf(&g_xy[2 * (size_t)id], dL_dG * (-gdx * co[0] - gdy * co[1]));
                    atomic_addf(&g_xy[2 * (size_t)id + 1], dL_dG * (-gdy * co[2] - gdx * co[1]));
                    atomic_addf(&g_con[3 * (size_t)id], -0.5f * gdx * dx * dL_dG);
                    atomic_addf(&g_con[3 * (size_t)id + 1], -gdx * dy * dL_dG);
                    atomic_addf(&g_con[3 * (size_t)id + 2], -0.5f * gdy * dy * dL_dG);
                    atomic_addf(&dL_dopac[id], G * dL_da);
                }
            }
    }

    const int nk = (prm->deg + 1) * (prm->deg + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        float* o2 = dL_dmeans2D + 3 * (size_t)i;
        float* o3 = dL_dmeans3D + 3 * (size_t)i;
        float* oc = dL_dcov3D + 6 * (size_t)i;
        o2[0] = o2[1] = o2[2] = 0.f; o3[0] = o3[1] = o3[2] = 0.f;
        for (int k = 0; k < 6; ++k) oc[k] = 0.f;
        if (dL_dsh) memset(dL_dsh + (size_t)i * K * 3, 0, sizeof(float) * (size_t)K * 3);
        if (dL_dscales) { float* q = dL_dscales + 3 * (size_t)i; q[0] = q[1] = q[2] = 0.f; }
        if (dL_drots) { float* q = dL_drots + 4 * (size_t)i; q[0] = q[1] = q[2] = q[3] = 0.f; }
        if (s->radii[i] <= 0) continue;
        const float* m = means3D + 3 * (size_t)i;
        const float* c6 = s->cov3d + 6 * (size_t)i;
        float dmean[3] = {0.f, 0.f, 0.f};
        /* --- conic -> cov2D -> (cov3D, t) --- */
        ewa_t e;
        ewa_setup(prm, view, m, &e);
        float s0[3], s1[3];
        sym6_mul(c6, e.M, s0);
        sym6_mul(c6, e.M + 3, s1);
        float a = (e.M[0] * s0[0] + e.M[1] * s0[1] + e.M[2] * s0[2]) + LOWPASS;
        float b = e.M[0] * s1[0] + e.M[1] * s1[1] + e.M[2] * s1[2];
        float c = (e.M[3] * s1[0] + e.M[4] * s1[1] + e.M[5] * s1[2]) + LOWPASS;
        float det = a * c - b * b;
        float d2i = 1.f / (det * det + DET_EPS);
        const float* q = g_con + 3 * (size_t)i;
        float da = d2i * (-c * c * q[0] + b * c * q[1] - b * b * q[2]);
        float dc = d2i * (-b * b * q[0] + a * b * q[1] - a * a * q[2]);
        float db = d2i * (2.f * b * c * q[0] - (det + 2.f * b * b) * q[1] + 2.f * a * b * q[2]);
        const float* M0 = e.M; const float* M1 = e.M + 3;
        /* dL/dSigma (6-vector, off-diagonals carry both symmetric entries) = M^T G M */
        oc[0] = M0[0] * M0[0] * da + M0[0] * M1[0] * db + M1[0] * M1[0] * dc;
        oc[3] = M0[1] * M0[1] * da + M0[1] * M1[1] * db + M1[1] * M1[1] * dc;
        oc[5] = M0[2] * M0[2] * da + M0[2] * M1[2] * db + M1[2] * M1[2] * dc;
        oc[1] = 2.f * M0[0] * M0[1] * da + (M0[0] * M1[1] + M0[1] * M1[0]) * db + 2.f * M1[0] * M1[1] * dc;
        oc[2] = 2.f * M0[0] * M0[2] * da + (M0[0] * M1[2] + M0[2] * M1[0]) * db + 2.f * M1[0] * M1[2] * dc;
        oc[4] = 2.f * M0[1] * M0[2] * da + (M0[1] * M1[2] + M0[2] * M1[1]) * db + 2.f * M1[1] * M1[2] * dc;
        /* dL/dM = 2 G (M Sigma), G = [[da, db/2],[db/2, dc]] */
        float dM0[3], dM1[3];
        for (int j = 0; j < 3; ++j) {
            dM0[j] = 2.f * da * s0[j] + db * s1[j];
            dM1[j] = db * s0[j] + 2.f * dc * s1[j];
        }
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
        for (int j = 0; j < 3; ++j) {
            dJ00 += dM0[j] * view[4 * j + 0]; dJ02 += dM0[j] * view[4 * j + 2];
            dJ11 += dM1[j] * view[4 * j + 1]; dJ12 += dM1[j] * view[4 * j + 2];
        }
        float tz = 1.f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        float dtx = e.xmul * -e.fx * tz2 * dJ02;
        float dty = e.ymul * -e.fy * tz2 * dJ12;
        float dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + 2.f * e.fx * e.tx * tz3 * dJ02 + 2.f * e.fy * e.ty * tz3 * dJ12;
        dtz += g_dep[i]; /* depth = t.z */
        dmean[0] += view[0] * dtx + view[1] * dty + view[2] * dtz;
        dmean[1] += view[4] * dtx + view[5] * dty + view[6] * dtz;
        dmean[2] += view[8] * dtx + view[9] * dty + view[10] * dtz;
        /* --- pixel mean -> NDC -> mean3D --- */
        float gnx = g_xy[2 * (size_t)i] * 0.5f * (float)W, gny = g_xy[2 * (size_t)i + 1] * 0.5f * (float)H;
        o2[0] = gnx; o2[1] = gny;
        float hx = proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12];
        float hy = proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13];
        float hw = proj[3] * m[0] + proj[7] * m[1] + proj[11] * m[2] + proj[15];
        float pw = 1.0f / (hw + W_EPS);
        float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
        for (int j = 0; j < 3; ++j)
            dmean[j] += (proj[4 * j] * pw - proj[4 * j + 3] * mul1) * gnx + (proj[4 * j + 1] * pw - proj[4 * j + 3] * mul2) * gny;
        /* --- colour --- */
        if (!colors && dL_dsh) {
            float v[3] = {m[0] - campos[0], m[1] - campos[1], m[2] - campos[2]};
            float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            float d[3] = {v[0] / len, v[1] / len, v[2] / len};
            float bas[16], dbas[48];
            sh_basis(prm->deg, d, bas);
            sh_basis_grad(prm->deg, d, dbas);
            const float* sh = shs + (size_t)i * K * 3;
            float* osh = dL_dsh + (size_t)i * K * 3;
            float ddir[3] = {0.f, 0.f, 0.f};
            for (int ch = 0; ch < 3; ++ch) {
                float g = s->clamped[3 * (size_t)i + ch] ? 0.f : dL_dcolors[3 * (size_t)i + ch];
                for (int k = 0; k < nk; ++k) {
                    osh[k * 3 + ch] = bas[k] * g;
                    float sg = sh[k * 3 + ch] * g;
                    ddir[0] += dbas[k * 3] * sg; ddir[1] += dbas[k * 3 + 1] * sg; ddir[2] += dbas[k * 3 + 2] * sg;
                }
            }
            float dot = d[0] * ddir[0] + d[1] * ddir[1] + d[2] * ddir[2];
            for (int j = 0; j < 3; ++j) dmean[j] += (ddir[j] - d[j] * dot) / len;
        }
        o3[0] = dmean[0]; o3[1] = dmean[1]; o3[2] = dmean[2];
        /* --- cov3D -> scale, rotation --- */
        if (!cov3d_pre && dL_dscales && dL_drots) {
            const float* sc = scales + 3 * (size_t)i;
            const float* qq = rots + 4 * (size_t)i;
            float R[9], Mm[9], sv[3];
            quat_R(qq, R);
            for (int k = 0; k < 3; ++k) { sv[k] = prm->scale_modifier * sc[k]; for (int j = 0; j < 3; ++j) Mm[k * 3 + j] = sv[k] * R[j * 3 + k]; }
            float Gs[9] = {oc[0], 0.5f * oc[1], 0.5f * oc[2], 0.5f * oc[1], oc[3], 0.5f * oc[4], 0.5f * oc[2], 0.5f * oc[4], oc[5]};
            float dMm[9]; /* 2 M Gs */
            for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j)
                dMm[k * 3 + j] = 2.f * (Mm[k * 3] * Gs[j] + Mm[k * 3 + 1] * Gs[3 + j] + Mm[k * 3 + 2] * Gs[6 + j]);
            float dR[9];
            float* osc = dL_dscales + 3 * (size_t)i;
            for (int k = 0; k < 3; ++k) {
                float acc = 0.f;
                for (int j = 0; j < 3; ++j) { acc += dMm[k * 3 + j] * R[j * 3 + k]; dR[j * 3 + k] = sv[k] * dMm[k * 3 + j]; }
                /* upstream returns dL/d(modifier*scale) as dL/dscale (no chain-rule factor `modifier`): reproduced */
                osc[k] = acc;
            }
            float r = qq[0], x = qq[1], y = qq[2], z = qq[3];
            float* oq = dL_drots + 4 * (size_t)i;
            oq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            oq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
            oq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
            oq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
    free(g_xy); free(g_con); free(g_dep);
}
