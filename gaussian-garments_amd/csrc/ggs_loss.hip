// ggs_loss.hip -- fused photometric loss of the inner steps: masked L1 + 11x11 Gaussian-window SSIM,
// value AND gradient w.r.t. the rendered image in two tile passes.
//
// Replaces ("next" #1) the PyTorch chain of utils/loss_utils.py:17-68 as composed at
// s2_registration.py:259-260 / s3_appearance.py:132-133:
//     loss_img  = mean(|(img - gt) * mask|) * (1 - lambda)
//     loss_ssim = 1 - mean(ssim_map(img * mask, gt * mask)) * lambda
// which costs 5 grouped conv2d forward + their backward (~10x the image bytes, milliseconds at 1080p --
// several times the HIP rasterizer's forward+backward).  Here:
//   pass A (ggs_k_loss_stats): per 32x32 tile and channel, x = img*mask and y = gt*mask (+5 px halo, zero
//           padded like conv2d(padding=5)) go to LDS; separable 11-tap window (same fp32 taps as
//           create_window) gives mu1, mu2, E[xx], E[yy], E[xy]; the SSIM map value and its three partial
//           derivatives (d/dmu1 total, d/dE[xx], d/dE[xy]) are formed per pixel; block-reduced sums of
//           |x - y| and of the map go to sums[v] (one atomic pair per block).
//   pass B (ggs_k_loss_grad): the three derivative maps are filtered with the same (symmetric) window:
//           dSSIM/dx = G*dmu1 + 2 x (G*dExx) + y (G*dExy); combined with the L1 sign term and the mask.
// Roofline: HBM (A: reads 24 B/px(+mask) writes 36 B/px; B: reads 60 B/px writes 12 B/px ~ 270 MB per 1080p
// view); ~530 MAC per pixel, far below the fp32 peak, so LDS traffic of the separable passes is what is tuned.
#include "ggs_kernels.h"

namespace {

#define LT 32                 // output tile edge
#define LH 5                  // window half width
#define LI (LT + 2 * LH)      // input tile edge incl. halo = 42
#define SSIM_C1 0.0001f       // 0.01^2
#define SSIM_C2 0.0009f       // 0.03^2

// fp32 taps of gaussian(11, 1.5) / sum, bit patterns as utils/loss_utils.py:26-28 produces them
__device__ __constant__ const float G11[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                               2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                               3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};

struct LossArgs {
    int V, H, W;
    const float *img, *gt, *mask;     // [V][3][H][W], [V][3][H][W], [V][1][H][W] or null
    const float* w;                   // [V][2] device: weights of mean|x-y| and of mean ssim_map in the loss
    float inv_n;                      // 1 / (3 H W)
    float* sums;                      // [V][2] = {sum |x - y|, sum ssim_map}
    float* dmap;                      // [V][3 channels][3 maps][H][W] scratch
    float* dL_dimg;                   // [V][3][H][W]
};

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    v = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return v;
}

}  // namespace

// Pass A: grid (ceil(W/32), ceil(H/32), V), block 256.  A workgroup walks the three colour channels of its tile: the
// mask and the index arithmetic are shared, and the global loads of channel c + 1 are in flight while channel c is
// filtered (with 3 workgroups per CU -- LDS -- nothing else hides the HBM latency of the tile + halo reads).
__global__ __launch_bounds__(256) void ggs_k_loss_stats(LossArgs a) {
    __shared__ float sx[LI][LI + 1], sy[LI][LI + 1];
    __shared__ float hh[5][LI][LT + 1];
    __shared__ float s_red[4];
    const int tid = threadIdx.x;
    const int v = blockIdx.z;
    const int ox = blockIdx.x * LT, oy = blockIdx.y * LT;
    const size_t HW = (size_t)a.H * a.W;
    const float* mask = a.mask ? a.mask + (size_t)v * HW : nullptr;

    constexpr int NL = (LI * LI + 255) / 256;
    float xr[NL], yr[NL], mr[NL];
    size_t pp[NL];
    bool in[NL];
    {
        const float* img = a.img + (size_t)v * 3 * HW;
        const float* gt = a.gt + (size_t)v * 3 * HW;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = tid + j * 256;
            const int r = i / LI, c = i % LI;
            const int y = oy + r - LH, x = ox + c - LH;
            in[j] = i < LI * LI && x >= 0 && x < a.W && y >= 0 && y < a.H;
            pp[j] = in[j] ? (size_t)y * a.W + x : 0;
            xr[j] = in[j] ? img[pp[j]] : 0.f;
            yr[j] = in[j] ? gt[pp[j]] : 0.f;
            mr[j] = (in[j] && mask) ? mask[pp[j]] : 1.f;
        }
    }
    float l1 = 0.f, ssum = 0.f;
    for (int ch = 0; ch < 3; ++ch) {
    {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = tid + j * 256;
            if (i >= LI * LI) continue;
            const int r = i / LI, c = i % LI;
            const float xv = xr[j] * mr[j], yv = yr[j] * mr[j];
            // the L1 term is summed over the tile's own pixels only (not the halo)
            if (in[j] && r >= LH && r < LH + LT && c >= LH && c < LH + LT) l1 += fabsf(xv - yv);
            sx[r][c] = xv; sy[r][c] = yv;
        }
    }
    __syncthreads();
    if (ch < 2) {                      // next channel's tile + halo: consumed after this channel's two filter passes
        const float* img = a.img + ((size_t)v * 3 + ch + 1) * HW;
        const float* gt = a.gt + ((size_t)v * 3 + ch + 1) * HW;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            xr[j] = in[j] ? img[pp[j]] : 0.f;
            yr[j] = in[j] ? gt[pp[j]] : 0.f;
        }
    }
    // horizontal pass, register blocked: thread = (row, 8 adjacent output columns) reads its 18 inputs of x and y once
    // (36 LDS reads for 8 outputs x 5 maps instead of 176) -- the kernel is bound by LDS latency, not by the FMAs.
    // Accumulation order per output is unchanged (taps 0..10).
    {
        const int r = tid >> 2, c0 = (tid & 3) * 8;
        if (r < LI) {
            float xv[18], yv[18];
#pragma unroll
            for (int j = 0; j < 18; ++j) { xv[j] = sx[r][c0 + j]; yv[j] = sy[r][c0 + j]; }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float g = G11[k], x = xv[o + k], y = yv[o + k];
                    m1 = fmaf(g, x, m1); m2 = fmaf(g, y, m2);
                    e11 = fmaf(g, x * x, e11); e22 = fmaf(g, y * y, e22); e12 = fmaf(g, x * y, e12);
                }
                hh[0][r][c0 + o] = m1; hh[1][r][c0 + o] = m2; hh[2][r][c0 + o] = e11; hh[3][r][c0 + o] = e22;
                hh[4][r][c0 + o] = e12;
            }
        }
    }
    __syncthreads();
    // vertical pass + SSIM map + derivative maps: thread = (column, 4 adjacent output rows), 14 reads per map
    float* dm = a.dmap + ((size_t)v * 3 + ch) * 3 * HW;
    {
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        const int x = ox + c;
        float col[5][14];
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int j = 0; j < 14; ++j) col[m][j] = hh[m][r0 + j][c];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int y = oy + r0 + o;
            if (x >= a.W || y >= a.H) continue;
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = G11[k];
                m1 = fmaf(g, col[0][o + k], m1); m2 = fmaf(g, col[1][o + k], m2);
                e11 = fmaf(g, col[2][o + k], e11); e22 = fmaf(g, col[3][o + k], e22);
                e12 = fmaf(g, col[4][o + k], e12);
            }
            const float v1 = e11 - m1 * m1, v2 = e22 - m2 * m2, cv = e12 - m1 * m2;
            const float A1 = 2.f * m1 * m2 + SSIM_C1, A2 = 2.f * cv + SSIM_C2;
            const float B1 = m1 * m1 + m2 * m2 + SSIM_C1, B2 = v1 + v2 + SSIM_C2;
            // two v_rcp_f32 (1 ulp) instead of five IEEE divisions (~10 instructions each); B1, B2 >= C1, C2 > 0
            const float iB1 = __builtin_amdgcn_rcpf(B1), iB2 = __builtin_amdgcn_rcpf(B2);
            const float inv = iB1 * iB2;
            const float S = A1 * A2 * inv;
            ssum += S;
            const size_t p = (size_t)y * a.W + x;
            // total derivative w.r.t. mu1 (through A1, A2 = 2(E12 - m1 m2) + C2, B1, B2 = E11 - m1^2 + ...)
            dm[p] = 2.f * m2 * (A2 - A1) * inv - 2.f * m1 * S * iB1 + 2.f * m1 * S * iB2;
            dm[HW + p] = -S * iB2;                // d/dE[xx]
            dm[2 * HW + p] = 2.f * A1 * inv;      // d/dE[xy]
        }
    }
    __syncthreads();                   // hh and sx / sy are rewritten by the next channel
    }  // ch
    l1 = block_sum(l1, s_red);
    ssum = block_sum(ssum, s_red);
    if (tid == 0) {
        atomicAdd(&a.sums[2 * v], l1);
        atomicAdd(&a.sums[2 * v + 1], ssum);
    }
}

// Pass B: grid (ceil(W/32), ceil(H/32), V*3), block 256.
__global__ __launch_bounds__(256) void ggs_k_loss_grad(LossArgs a) {
    __shared__ float sd[3][LI][LI + 1];
    __shared__ float hh[3][LI][LT + 1];
    const int tid = threadIdx.x;
    const int v = blockIdx.z / 3, ch = blockIdx.z % 3;
    const int ox = blockIdx.x * LT, oy = blockIdx.y * LT;
    const size_t HW = (size_t)a.H * a.W;
    const float* dm = a.dmap + ((size_t)v * 3 + ch) * 3 * HW;
    {   // all global loads of a thread in flight at once (see pass A)
        constexpr int NL = (LI * LI + 255) / 256;
        float d0[NL], d1[NL], d2[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = tid + j * 256;
            const int r = i / LI, c = i % LI;
            const int y = oy + r - LH, x = ox + c - LH;
            const bool in = i < LI * LI && x >= 0 && x < a.W && y >= 0 && y < a.H;
            const size_t p = in ? (size_t)y * a.W + x : 0;
            d0[j] = in ? dm[p] : 0.f; d1[j] = in ? dm[HW + p] : 0.f; d2[j] = in ? dm[2 * HW + p] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = tid + j * 256;
            if (i >= LI * LI) continue;
            const int r = i / LI, c = i % LI;
            sd[0][r][c] = d0[j]; sd[1][r][c] = d1[j]; sd[2][r][c] = d2[j];
        }
    }
    __syncthreads();
    {   // horizontal, register blocked like pass A
        const int r = tid >> 2, c0 = (tid & 3) * 8;
        if (r < LI) {
            float d[3][18];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int j = 0; j < 18; ++j) d[m][j] = sd[m][r][c0 + j];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float g = G11[k];
                    h0 = fmaf(g, d[0][o + k], h0); h1 = fmaf(g, d[1][o + k], h1); h2 = fmaf(g, d[2][o + k], h2);
                }
                hh[0][r][c0 + o] = h0; hh[1][r][c0 + o] = h1; hh[2][r][c0 + o] = h2;
            }
        }
    }
    __syncthreads();
    const float* img = a.img + ((size_t)v * 3 + ch) * HW;
    const float* gt = a.gt + ((size_t)v * 3 + ch) * HW;
    const float* mask = a.mask ? a.mask + (size_t)v * HW : nullptr;
    float* out = a.dL_dimg + ((size_t)v * 3 + ch) * HW;
    const float w_l1 = a.w[2 * v] * a.inv_n, w_ssim = a.w[2 * v + 1] * a.inv_n;
    {
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        const int x = ox + c;
        float col[3][14];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int j = 0; j < 14; ++j) col[m][j] = hh[m][r0 + j][c];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int y = oy + r0 + o;
            if (x >= a.W || y >= a.H) continue;
            float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = G11[k];
                f0 = fmaf(g, col[0][o + k], f0); f1 = fmaf(g, col[1][o + k], f1); f2 = fmaf(g, col[2][o + k], f2);
            }
            const size_t p = (size_t)y * a.W + x;
            const float m = mask ? mask[p] : 1.f;
            const float xv = img[p] * m, yv = gt[p] * m;
            const float dssim = f0 + 2.f * xv * f1 + yv * f2;
            const float df = xv - yv;
            const float dl1 = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            out[p] = m * (w_ssim * dssim + w_l1 * dl1);
        }
    }
}

extern "C" {

size_t ggs_photometric_scratch_bytes(int n_views, int H, int W) {
    if (n_views <= 0 || H <= 0 || W <= 0) return 0;
    return ggs_align((size_t)n_views * 9 * (size_t)H * W * sizeof(float));
}

static int loss_args(LossArgs& a, int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                     void* scratch, const char* who) {
    ggs_clear_error_();
    if (n_views <= 0 || H <= 0 || W <= 0) return ggs_fail_(GGS_ERR_ARG, "%s: bad sizes", who);
    if (!img || !gt || !scratch) return ggs_fail_(GGS_ERR_ARG, "%s: NULL pointer argument", who);
    if ((size_t)n_views * 3 > 65535) return ggs_fail_(GGS_ERR_SIZE, "%s: n_views too large", who);
    a.V = n_views; a.H = H; a.W = W; a.img = img; a.gt = gt; a.mask = mask;
    a.inv_n = 1.f / (3.f * (float)H * (float)W);
    a.w = nullptr; a.sums = nullptr; a.dL_dimg = nullptr; a.dmap = (float*)scratch;
    return GGS_OK;
}

int ggs_photometric_forward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                            float* sums, void* scratch, void* stream_) {
    LossArgs a;
    int rc = loss_args(a, n_views, H, W, img, gt, mask, scratch, "ggs_photometric_forward");
    if (rc != GGS_OK) return rc;
    if (!sums) return ggs_fail_(GGS_ERR_ARG, "ggs_photometric_forward: sums is NULL");
    hipStream_t s = (hipStream_t)stream_;
    a.sums = sums;
    if (ggs_zero_async(sums, (size_t)n_views * 2 * sizeof(float), s) != hipSuccess)
        return ggs_fail_(GGS_ERR_HIP, "ggs_photometric_forward: clearing the sums failed");
    const dim3 grid((unsigned)((W + LT - 1) / LT), (unsigned)((H + LT - 1) / LT), (unsigned)n_views);
    hipLaunchKernelGGL(ggs_k_loss_stats, grid, dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "loss_stats launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

int ggs_photometric_backward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                             const void* scratch, const float* weights, float* dL_dimg, void* stream_) {
    LossArgs a;
    int rc = loss_args(a, n_views, H, W, img, gt, mask, (void*)scratch, "ggs_photometric_backward");
    if (rc != GGS_OK) return rc;
    if (!weights || !dL_dimg) return ggs_fail_(GGS_ERR_ARG, "ggs_photometric_backward: NULL pointer argument");
    a.w = weights; a.dL_dimg = dL_dimg;
    const dim3 grid((unsigned)((W + LT - 1) / LT), (unsigned)((H + LT - 1) / LT), (unsigned)(n_views * 3));
    hipLaunchKernelGGL(ggs_k_loss_grad, grid, dim3(256), 0, (hipStream_t)stream_, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "loss_grad launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

}  // extern "C"
