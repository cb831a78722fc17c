// ggs_stylegan.hip -- the two StyleGAN2 elementwise/resampling ops the appearance network (StyleUNet,
// scene/styleunet/styleunet.py) calls through the extension modules `fused` and `upfirdn2d`
// (SURVEY.md section 8f #3).  Written from the operator semantics (scene/styleunet/fused_act.py:33-130,
// scene/styleunet/upfirdn2d.py:98-227 incl. the in-tree PyTorch reference path upfirdn2d_native); the
// autograd wrappers stay the reference's own Python, which builds both backward passes out of these
// same two forward ops.
//
//   fused_bias_act : y = act(x + b[(i / step_b) % size_b]) * scale     act 1 = linear, 3 = leaky ReLU(alpha);
//                    grad 0 = forward, 1 = first derivative gated by the sign of `ref` (the saved forward
//                    output), 2 = second derivative (identically zero for these piecewise-linear acts).
//   upfirdn2d      : zero-insert upsample by (up_x, up_y) -> pad / crop -> true 2-D convolution with a small
//                    FIR kernel -> keep every (down_x, down_y)-th sample.  Layout [major][h][w][minor].
// Both are HBM-bound: bias_act streams 8 B per element (float4 per lane); upfirdn2d reads each input
// sample ceil(kw/up_x) * ceil(kh/up_y) times, served from L1/L2 (the taps of neighbouring outputs overlap).
#include "ggs_kernels.h"

namespace {

struct BiasActArgs {
    size_t n;
    const float *x, *b, *ref;
    float* y;
    int step_b, size_b, act, grad;
    float alpha, scale;
};

__device__ __forceinline__ float bias_act_one(float x, float ref, int mode, float alpha) {
    switch (mode) {
        case 30: return x > 0.f ? x : x * alpha;
        case 31: return ref > 0.f ? x : x * alpha;
        case 12:
        case 32: return 0.f;
        default: return x;                     // 10, 11 and anything unknown: linear
    }
}

__global__ __launch_bounds__(256) void k_bias_act(BiasActArgs a) {
    const int mode = a.act * 10 + a.grad;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t n4 = a.n / 4;
    const bool vec_bias = !a.b || (a.step_b % 4 == 0);       // a float4 never straddles two bias entries
    if (vec_bias) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(a.x)[i];
            if (a.b) {
                const float bb = a.b[((i * 4) / (size_t)a.step_b) % (size_t)a.size_b];
                v.x += bb; v.y += bb; v.z += bb; v.w += bb;
            }
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.ref) r = reinterpret_cast<const float4*>(a.ref)[i];
            float4 o;
            o.x = bias_act_one(v.x, r.x, mode, a.alpha) * a.scale;
            o.y = bias_act_one(v.y, r.y, mode, a.alpha) * a.scale;
            o.z = bias_act_one(v.z, r.z, mode, a.alpha) * a.scale;
            o.w = bias_act_one(v.w, r.w, mode, a.alpha) * a.scale;
            reinterpret_cast<float4*>(a.y)[i] = o;
        }
    }
    const size_t first = vec_bias ? n4 * 4 : 0;
    for (size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        float v = a.x[i];
        if (a.b) v += a.b[(i / (size_t)a.step_b) % (size_t)a.size_b];
        a.y[i] = bias_act_one(v, a.ref ? a.ref[i] : 0.f, mode, a.alpha) * a.scale;
    }
}

struct UpfirdnArgs {
    const float *in, *kernel;
    float* out;
    int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

// One lane per output sample (minor fastest, then x): for tap (i, j) the sample of the zero-inserted, padded
// signal at (oy * down_y + i - pad_y0, ox * down_x + j - pad_x0) is non-zero only on the up-sampling lattice.
__global__ __launch_bounds__(256) void k_upfirdn2d(UpfirdnArgs a) {
    const size_t total = (size_t)a.major * a.out_h * a.out_w * a.minor;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % a.minor);
        size_t t = idx / a.minor;
        const int ox = (int)(t % a.out_w); t /= a.out_w;
        const int oy = (int)(t % a.out_h);
        const int m = (int)(t / a.out_h);
        const float* src = a.in + (size_t)m * a.in_h * a.in_w * a.minor + c;
        float acc = 0.f;
        for (int i = 0; i < a.kh; ++i) {
            const int py = oy * a.down_y + i - a.pad_y0;
            if (py < 0 || py % a.up_y) continue;
            const int iy = py / a.up_y;
            if (iy >= a.in_h) continue;
            for (int j = 0; j < a.kw; ++j) {
                const int px = ox * a.down_x + j - a.pad_x0;
                if (px < 0 || px % a.up_x) continue;
                const int ix = px / a.up_x;
                if (ix >= a.in_w) continue;
                acc = fmaf(src[((size_t)iy * a.in_w + ix) * a.minor], a.kernel[(a.kh - 1 - i) * a.kw + (a.kw - 1 - j)], acc);
            }
        }
        a.out[idx] = acc;
    }
}

}  // namespace

extern "C" {

int ggs_fused_bias_act(size_t n, const float* x, const float* bias, const float* ref, int step_b, int size_b,
                       int act, int grad, float alpha, float scale, float* y, void* stream) {
    ggs_clear_error_();
    if (n == 0) return GGS_OK;
    if (!x || !y) return ggs_fail_(GGS_ERR_ARG, "ggs_fused_bias_act: NULL pointer argument");
    if (bias && (step_b <= 0 || size_b <= 0)) return ggs_fail_(GGS_ERR_ARG, "ggs_fused_bias_act: bad bias geometry");
    BiasActArgs a;
    a.n = n; a.x = x; a.b = bias; a.ref = ref; a.y = y; a.step_b = step_b; a.size_b = size_b; a.act = act; a.grad = grad;
    a.alpha = alpha; a.scale = scale;
    size_t blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    hipLaunchKernelGGL(k_bias_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "bias_act launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

int ggs_upfirdn2d_out_size(int in_h, int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                           int pad_x1, int pad_y0, int pad_y1, int* out_h, int* out_w) {
    ggs_clear_error_();
    if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0 || kh <= 0 || kw <= 0 || !out_h || !out_w)
        return ggs_fail_(GGS_ERR_ARG, "ggs_upfirdn2d: bad factors / kernel size");
    *out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    *out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    return GGS_OK;
}

int ggs_upfirdn2d(int major, int in_h, int in_w, int minor, const float* input, const float* kernel, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                  float* out, void* stream) {
    UpfirdnArgs a;
    int rc = ggs_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                                    &a.out_h, &a.out_w);
    if (rc != GGS_OK) return rc;
    if (major < 0 || in_h < 0 || in_w < 0 || minor < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_upfirdn2d: bad sizes");
    const size_t total = (size_t)major * (size_t)(a.out_h > 0 ? a.out_h : 0) * (size_t)(a.out_w > 0 ? a.out_w : 0) * (size_t)minor;
    if (total == 0) return GGS_OK;
    if (!input || !kernel || !out) return ggs_fail_(GGS_ERR_ARG, "ggs_upfirdn2d: NULL pointer argument");
    a.in = input; a.kernel = kernel; a.out = out; a.major = major; a.in_h = in_h; a.in_w = in_w; a.minor = minor;
    a.kh = kh; a.kw = kw; a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
    a.pad_x0 = pad_x0; a.pad_y0 = pad_y0;
    size_t blocks = (total + 255) / 256;
    blocks = blocks > 16384 ? 16384 : blocks;
    hipLaunchKernelGGL(k_upfirdn2d, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "upfirdn2d launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

}  // extern "C"
