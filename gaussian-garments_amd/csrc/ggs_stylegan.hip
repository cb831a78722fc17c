// ggs_stylegan.hip -- the two StyleGAN2 elementwise / resampling ops the appearance network (StyleUNet,
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
//
// Element types: float, half and double, native I/O (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// fused_bias_act_kernel.cu:96, upfirdn2d_kernel.cu:340); half computes in float like upstream's accumulators, double
// in double.  Nothing is converted on the host.
//
// Roofline: both ops are HBM streams (read the input once, write the output once).
//   * bias_act moves 2 (3 with `ref`) elements per output, 16 B per lane per access.
//   * upfirdn2d, the shapes StyleUNet uses (minor = 1, major = batch x channels; styleunet.py:32-91, 387-425):
//       blur 4x4 (up 1, down 1), upsample 4x4 (up 2), downsample 4x4 (down 2), Haar 2x2 (down 2), inverse Haar 2x2 (up 2)
//     and their backward passes (the same ops with up <-> down and the flipped kernel, upfirdn2d.py:128-141) run in
//     k_upfirdn2d_tile: a workgroup owns a 64 x 16 output tile of one image plane, stages the input patch it needs
//     (tile + FIR halo, zero outside the image) in LDS with coalesced row reads, and every lane evaluates 4 vertically
//     adjacent outputs (overlapping windows: each patch sample is read from LDS once per lane) with compile-time
//     up / down / tap counts: no integer division or modulo per tap, only the taps on the
//     up-sampling lattice are visited (polyphase), each input sample crosses HBM once.
//   * every other configuration (minor > 1, odd factors, other FIR sizes) takes the generic gather kernel.
#include <hip/hip_fp16.h>

#include "ggs_kernels.h"

namespace {

template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename Acc<T>::type ld(const T* p) { return (typename Acc<T>::type)(*p); }
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <typename T, typename A> __device__ __forceinline__ void st(T* p, A v) { *p = (T)v; }
template <> __device__ __forceinline__ void st<__half, float>(__half* p, float v) { *p = __float2half(v); }

// VW adjacent elements as ONE access of up to 8 bytes (float: 2, half: 2 or 4); the address is a multiple of VW elements
template <typename T, typename A, int VW> struct VecIO {
    static __device__ __forceinline__ void load(const T* p, A (&v)[VW]) {
#pragma unroll
        for (int k = 0; k < VW; ++k) v[k] = ld(p + k);
    }
    static __device__ __forceinline__ void store(T* p, const A (&v)[VW]) {
#pragma unroll
        for (int k = 0; k < VW; ++k) st(p + k, v[k]);
    }
};
template <> struct VecIO<float, float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[2]) { const float2 f = *reinterpret_cast<const float2*>(p); v[0] = f.x; v[1] = f.y; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <> struct VecIO<__half, float, 2> {
    static __device__ __forceinline__ void load(const __half* p, float (&v)[2]) { const __half2 h = *reinterpret_cast<const __half2*>(p); v[0] = __low2float(h); v[1] = __high2float(h); }
    static __device__ __forceinline__ void store(__half* p, const float (&v)[2]) { *reinterpret_cast<__half2*>(p) = __floats2half2_rn(v[0], v[1]); }
};
template <> struct VecIO<__half, float, 4> {
    static __device__ __forceinline__ void load(const __half* p, float (&v)[4]) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        const __half2 a = *reinterpret_cast<const __half2*>(&u.x), b = *reinterpret_cast<const __half2*>(&u.y);
        v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
    }
    static __device__ __forceinline__ void store(__half* p, const float (&v)[4]) {
        const __half2 a = __floats2half2_rn(v[0], v[1]), b = __floats2half2_rn(v[2], v[3]);
        uint2 u;
        u.x = *reinterpret_cast<const unsigned*>(&a); u.y = *reinterpret_cast<const unsigned*>(&b);
        *reinterpret_cast<uint2*>(p) = u;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct BiasActArgs {
    size_t n;
    const T *x, *b, *ref;
    T* y;
    int step_b, size_b, act, grad;
    float alpha, scale;
};

template <typename A>
__device__ __forceinline__ A bias_act_one(A x, A ref, int mode, A alpha) {
    switch (mode) {
        case 30: return x > (A)0 ? x : x * alpha;
        case 31: return ref > (A)0 ? x : x * alpha;
        case 12:
        case 32: return (A)0;
        default: return x;                     // 10, 11 and anything unknown: linear
    }
}

// VEC elements = 16 bytes per lane and access
template <typename T>
__global__ __launch_bounds__(256) void k_bias_act(BiasActArgs<T> a) {
    typedef typename Acc<T>::type A;
    constexpr int VEC = 16 / sizeof(T);
    struct alignas(16) Pack { T v[VEC]; };
    const int mode = a.act * 10 + a.grad;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t nv = a.n / VEC;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.y) |
                          reinterpret_cast<uintptr_t>(a.ref)) & 15) == 0 &&
                        (!a.b || a.step_b % VEC == 0);      // a pack never straddles two bias entries
    const A alpha = (A)a.alpha, scale = (A)a.scale;
    if (vec_ok) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
            Pack v = reinterpret_cast<const Pack*>(a.x)[i], r, o;
            if (a.ref) r = reinterpret_cast<const Pack*>(a.ref)[i];
            const A bb = a.b ? ld(a.b + ((i * VEC) / (size_t)a.step_b) % (size_t)a.size_b) : (A)0;
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                st(&o.v[k], bias_act_one<A>(ld(&v.v[k]) + bb, a.ref ? ld(&r.v[k]) : (A)0, mode, alpha) * scale);
            reinterpret_cast<Pack*>(a.y)[i] = o;
        }
    }
    const size_t first = vec_ok ? nv * VEC : 0;
    for (size_t i = first + (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
        A v = ld(a.x + i);
        if (a.b) v += ld(a.b + (i / (size_t)a.step_b) % (size_t)a.size_b);
        st(a.y + i, bias_act_one<A>(v, a.ref ? ld(a.ref + i) : (A)0, mode, alpha) * scale);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct UpfirdnArgs {
    const T *in, *kernel;
    T* out;
    int major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

// Generic: one lane per output sample (minor fastest, then x): for tap (i, j) the sample of the zero-inserted, padded
// signal at (oy * down_y + i - pad_y0, ox * down_x + j - pad_x0) is non-zero only on the up-sampling lattice.
template <typename T>
__global__ __launch_bounds__(256) void k_upfirdn2d(UpfirdnArgs<T> a) {
    typedef typename Acc<T>::type A;
    const size_t total = (size_t)a.major * a.out_h * a.out_w * a.minor;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % a.minor);
        size_t t = idx / a.minor;
        const int ox = (int)(t % a.out_w); t /= a.out_w;
        const int oy = (int)(t % a.out_h);
        const int m = (int)(t / a.out_h);
        const T* src = a.in + (size_t)m * a.in_h * a.in_w * a.minor + c;
        A acc = (A)0;
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
                acc += ld(src + ((size_t)iy * a.in_w + ix) * a.minor) * ld(a.kernel + (a.kh - 1 - i) * a.kw + (a.kw - 1 - j));
            }
        }
        st(a.out + idx, acc);
    }
}

// Tiled, minor == 1, UP / DOWN in {1, 2} (same factor in x and y), KH x KW taps known at compile time.
//   grid (ceil(out_w / 64), ceil(out_h / TH), major), TH = 32 (16 where the patch would not fit the LDS), block 256: lane
//   (tx = t & 63, ty = t >> 6) owns the outputs (ox0 + tx, oy0 + 4 (ty + 4 g) + r), r = 0..3, g = 0..TH/16-1.
// Coordinates: output (ox, oy) reads the zero-inserted signal at px = ox DOWN + j - pad_x0 (j = 0..KW-1), which holds
// input sample px / UP where px is a multiple of UP.  The tile needs px in [px_lo, px_lo + (63 DOWN + KW - 1)], i.e.
// input columns ix_lo = ceil(px_lo / UP) ... ; the LDS patch stores them densely, zero where the image ends.
#define UFD_TW 64
// tile height: 8 outputs per lane (two groups of 4 rows) where the patch stays under 16 KB of LDS, else 4.
// COLS = output columns per lane: 1, or 2 for half (a 128-column tile: the two outputs leave as ONE 4-byte store and the
// patch is staged with 4-byte loads -- a half tile of 64 columns keeps half the bytes in flight per workgroup that a float
// tile does, and the kernel is bound by exactly that: 2.0 against 3.0 TB/s for the same op in round 2).
template <typename A, int UP, int DOWN, int K, int COLS, int VW> struct UfdTile {
    static constexpr int W = ((UFD_TW * COLS - 1) * DOWN + K - 1) / UP + 2 + (VW - 1);          // (+VW-1: a patch staged VW columns at a time starts on a multiple of VW)
    static constexpr int H32 = ((32 - 1) * DOWN + K - 1) / UP + 2;
    static constexpr int TH = (size_t)H32 * (W + 1) * sizeof(A) <= 16 * 1024 ? 32 : 16;      // measured: the 35 KB patches of down = 2 lose more in occupancy (3.2 -> 1.9 TB/s) than the shorter halo gains
    static constexpr int H = ((TH - 1) * DOWN + K - 1) / UP + 2;
};

__device__ __forceinline__ int ceil_div_up(int a, int up) { return up == 1 ? a : (a + 1) >> 1; }    // ceil(a / 2), any sign

// The 4 outputs of a lane are 4 CONSECUTIVE rows (oy0 + 4 ty + r): their FIR windows overlap, and with every LDS row offset
// a compile-time constant the compiler loads each patch sample once per lane (4x4 blur: 7 rows x 4 columns = 28 LDS reads
// for 4 outputs instead of 64).  UP == 2: the row phase of output r is (P0 + r) & 1 with P0 = pad_y0 & 1 -- the tile
// origin and 4 ty are even -- so it is wave-uniform and a template parameter here; the column phase stays per lane.
template <typename A, int UP, int DOWN, int KH, int KW, int P0, int LD>
__device__ __forceinline__ void ufd_rows(const A* __restrict__ base, const A (&w)[KH][KW], int j0, A (&acc)[4]) {
    // base = &patch[row of (r = 0, ii = 0)][column of jj = 0]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i0 = UP == 1 ? 0 : ((P0 + r) & 1);
        const int row0 = UP == 1 ? r * DOWN : ((P0 + r + 1) >> 1);          // patch row of tap ii = 0, relative to base
        A sum = (A)0;
#pragma unroll
        for (int ii = 0; ii < (KH + UP - 1) / UP; ++ii) {
#pragma unroll
            for (int jj = 0; jj < (KW + UP - 1) / UP; ++jj) {
                const A wt = UP == 1 ? w[ii][jj] : (j0 ? w[i0 + 2 * ii][2 * jj + 1] : w[i0 + 2 * ii][2 * jj]);
                sum += base[(row0 + ii) * LD + jj] * wt;
            }
        }
        acc[r] = sum;
    }
}

// VW > 1 (half / float, in_w a multiple of VW): the patch is staged VW input columns per access (4 or 8 bytes).
template <typename T, int UP, int DOWN, int KH, int KW, int COLS, int VW>
__global__ __launch_bounds__(256) void k_upfirdn2d_tile(UpfirdnArgs<T> a) {
    typedef typename Acc<T>::type A;
    typedef UfdTile<A, UP, DOWN, KH, COLS, VW> Tile;
    constexpr bool VEC = VW > 1;
    constexpr int PW = Tile::W, PH = Tile::H, LD = PW + 1, TH = Tile::TH, G = TH / 16;      // G groups of 4 rows per lane
    constexpr int TW = UFD_TW * COLS;
    __shared__ A patch[PH * LD];
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH, m = blockIdx.z;
    const int px_lo = ox0 * DOWN - a.pad_x0, py_lo = oy0 * DOWN - a.pad_y0;
    const int ix_lo = VEC ? (ceil_div_up(px_lo, UP) & ~(VW - 1)) : ceil_div_up(px_lo, UP), iy_lo = ceil_div_up(py_lo, UP);
    const T* src = a.in + (size_t)m * a.in_h * a.in_w;
    if constexpr (VEC) {
        // groups of VW input columns as one load: in_w is a multiple of VW (launch condition), so a group is inside the image or
        // outside it as a whole, and every group is aligned
        constexpr int PWV = (PW + VW - 1) / VW;
        for (int i = threadIdx.x; i < PWV * PH; i += 256) {
            const int r = i / PWV, c = VW * (i - r * PWV);
            const int ix = ix_lo + c, iy = iy_lo + r;
            A v[VW];
#pragma unroll
            for (int k = 0; k < VW; ++k) v[k] = (A)0;
            if (ix >= 0 && ix < a.in_w && iy >= 0 && iy < a.in_h) VecIO<T, A, VW>::load(src + (size_t)iy * a.in_w + ix, v);
#pragma unroll
            for (int k = 0; k < VW; ++k)
                if (c + k < PW) patch[r * LD + c + k] = v[k];
        }
    } else {
        for (int i = threadIdx.x; i < PW * PH; i += 256) {
            const int r = i / PW, c = i - r * PW;              // PW is a compile-time constant: multiply-shift, once per sample
            const int ix = ix_lo + c, iy = iy_lo + r;
            A v = (A)0;
            if (ix >= 0 && ix < a.in_w && iy >= 0 && iy < a.in_h) v = ld(src + (size_t)iy * a.in_w + ix);
            patch[r * LD + c] = v;
        }
    }
    A w[KH][KW];                                            // flipped FIR: tap (i, j) multiplies kernel[KH-1-i][KW-1-j]
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j) w[i][j] = ld(a.kernel + (KH - 1 - i) * KW + (KW - 1 - j));
    __syncthreads();
    const int tx = threadIdx.x & (UFD_TW - 1), ty = threadIdx.x >> 6;
    const int ox = ox0 + tx * COLS;
    if (ox >= a.out_w) return;
    // first tap on the up-sampling lattice and its patch column: px = ox DOWN + j - pad_x0 must be a multiple of UP
    int j0[COLS], cx[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const int pxb = (ox + c) * DOWN - a.pad_x0;
        j0[c] = UP == 1 ? 0 : (pxb & 1);                    // pxb + j even  <=>  j has the parity of pxb
        cx[c] = (UP == 1 ? pxb : (pxb + j0[c]) >> 1) - ix_lo;
    }
    const int p0 = UP == 1 ? 0 : (a.pad_y0 & 1);            // = pyb & 1 for every lane (oyb is a multiple of 4)
    T* dst = a.out + (size_t)m * a.out_h * a.out_w + ox;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int oyb = oy0 + 4 * (ty + 4 * g);             // 4 consecutive rows; the groups of a lane are 16 rows apart
        if (oyb >= a.out_h) break;
        const int pyb = oyb * DOWN - a.pad_y0;
        const int cy = (UP == 1 ? pyb : (pyb - p0) >> 1) - iy_lo;
        A acc[COLS][4];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const A* base = patch + cy * LD + cx[c];
            if (p0) ufd_rows<A, UP, DOWN, KH, KW, 1, LD>(base, w, j0[c], acc[c]);
            else ufd_rows<A, UP, DOWN, KH, KW, 0, LD>(base, w, j0[c], acc[c]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (oyb + r >= a.out_h) break;
            if constexpr (COLS > 1) {                        // out_w is a multiple of COLS (launch condition): one aligned store
                A o[COLS];
#pragma unroll
                for (int c = 0; c < COLS; ++c) o[c] = acc[c][r];
                VecIO<T, A, COLS>::store(dst + (size_t)(oyb + r) * a.out_w, o);
            } else {
                st(dst + (size_t)(oyb + r) * a.out_w, acc[0][r]);
            }
        }
    }
}

template <typename T>
int launch_upfirdn(const UpfirdnArgs<T>& a, hipStream_t s) {
    const size_t total = (size_t)a.major * a.out_h * a.out_w * a.minor;
    const bool sq = a.up_x == a.up_y && a.down_x == a.down_y && a.kh == a.kw && a.minor == 1 && a.major <= 65535;
    typedef typename Acc<T>::type A;
    const int key = sq && a.out_h / 16 + 1 <= 65535 ? a.up_x * 100 + a.down_x * 10 + a.kh : -1;
    // vector accesses of n elements need row pitches and plane sizes that are multiples of n and n-element aligned bases:
    // in_al / out_al = the largest such n in {1, 2, 4}
    auto align_of = [](const void* base, int w, size_t plane) {
        int n = 1;
        for (int c : {2, 4})
            if (w % c == 0 && plane % c == 0 && (reinterpret_cast<uintptr_t>(base) % (c * sizeof(T))) == 0) n = c;
        return n;
    };
    const int in_al = align_of(a.in, a.in_w, (size_t)a.in_h * a.in_w), out_al = align_of(a.out, a.out_w, (size_t)a.out_h * a.out_w);
    (void)in_al; (void)out_al;
    switch (key) {
#define UFD_LAUNCH(UP, DOWN, K, COLS, VW)                                                                                        \
        {                                                                                                                         \
            constexpr int TH_ = UfdTile<A, UP, DOWN, K, COLS, VW>::TH;                                                            \
            const dim3 grid_((unsigned)((a.out_w + COLS * UFD_TW - 1) / (COLS * UFD_TW)), (unsigned)((a.out_h + TH_ - 1) / TH_), (unsigned)a.major); \
            hipLaunchKernelGGL((k_upfirdn2d_tile<T, UP, DOWN, K, K, COLS, VW>), grid_, dim3(256), 0, s, a);                       \
        }
#define UFD_CASE(UP, DOWN, K) \
        case UP * 100 + DOWN * 10 + K: {                                                                                         \
            /* half and float: several output columns per lane (up-sampling and same-size filters; a down-sampling tile that   */ \
            /* wide needs a > 32 KB patch and measured slower) and / or the patch staged several input columns per load        */ \
            if constexpr (sizeof(T) == 2) {                                                                                       \
                if (in_al >= 4 && out_al >= 4 && DOWN == 1) { UFD_LAUNCH(UP, DOWN, K, 4, 4) break; }                               \
                if (in_al >= 2 && out_al >= 2 && DOWN == 1) { UFD_LAUNCH(UP, DOWN, K, 2, 2) break; }                               \
                if (in_al >= 4) { UFD_LAUNCH(UP, DOWN, K, 1, 4) break; }                                                           \
                if (in_al >= 2) { UFD_LAUNCH(UP, DOWN, K, 1, 2) break; }                                                           \
            }                                                                                                                     \
            if constexpr (sizeof(T) == 4) {                                                                                       \
                if (in_al >= 2 && out_al >= 2 && DOWN == 1) { UFD_LAUNCH(UP, DOWN, K, 2, 2) break; }                               \
                if (in_al >= 2) { UFD_LAUNCH(UP, DOWN, K, 1, 2) break; }                                                           \
            }                                                                                                                     \
            UFD_LAUNCH(UP, DOWN, K, 1, 1)                                                                                         \
        } break;
        UFD_CASE(1, 1, 4) UFD_CASE(2, 1, 4) UFD_CASE(1, 2, 4)      // Blur, Upsample, Downsample (+ their backward passes)
        UFD_CASE(1, 2, 2) UFD_CASE(2, 1, 2)                        // Haar, inverse Haar
        UFD_CASE(1, 1, 2)
#undef UFD_CASE
#undef UFD_LAUNCH
        default: {
            size_t blocks = (total + 255) / 256;
            blocks = blocks > 16384 ? 16384 : blocks;
            hipLaunchKernelGGL(k_upfirdn2d<T>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "upfirdn2d launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

template <typename T>
int run_bias_act(size_t n, const void* x, const void* bias, const void* ref, int step_b, int size_b, int act, int grad,
                 float alpha, float scale, void* y, hipStream_t s) {
    BiasActArgs<T> a;
    a.n = n; a.x = (const T*)x; a.b = (const T*)bias; a.ref = (const T*)ref; a.y = (T*)y; a.step_b = step_b; a.size_b = size_b;
    a.act = act; a.grad = grad; a.alpha = alpha; a.scale = scale;
    size_t blocks = (n / (16 / sizeof(T)) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    hipLaunchKernelGGL(k_bias_act<T>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ggs_fail_(GGS_ERR_HIP, "bias_act launch failed: %s", hipGetErrorString(e));
    return GGS_OK;
}

template <typename T>
int run_upfirdn(int major, int in_h, int in_w, int minor, const void* input, const void* kernel, int kh, int kw, int up_x,
                int up_y, int down_x, int down_y, int pad_x0, int pad_y0, int out_h, int out_w, void* out, hipStream_t s) {
    UpfirdnArgs<T> a;
    a.in = (const T*)input; a.kernel = (const T*)kernel; a.out = (T*)out; a.major = major; a.in_h = in_h; a.in_w = in_w;
    a.minor = minor; a.kh = kh; a.kw = kw; a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
    a.pad_x0 = pad_x0; a.pad_y0 = pad_y0; a.out_h = out_h; a.out_w = out_w;
    return launch_upfirdn<T>(a, s);
}

}  // namespace

extern "C" {

int ggs_fused_bias_act_t(int dtype, size_t n, const void* x, const void* bias, const void* ref, int step_b, int size_b,
                         int act, int grad, float alpha, float scale, void* y, void* stream) {
    ggs_clear_error_();
    if (n == 0) return GGS_OK;
    if (!x || !y) return ggs_fail_(GGS_ERR_ARG, "ggs_fused_bias_act: NULL pointer argument");
    if (bias && (step_b <= 0 || size_b <= 0)) return ggs_fail_(GGS_ERR_ARG, "ggs_fused_bias_act: bad bias geometry");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case GGS_DTYPE_F32: return run_bias_act<float>(n, x, bias, ref, step_b, size_b, act, grad, alpha, scale, y, s);
        case GGS_DTYPE_F16: return run_bias_act<__half>(n, x, bias, ref, step_b, size_b, act, grad, alpha, scale, y, s);
        case GGS_DTYPE_F64: return run_bias_act<double>(n, x, bias, ref, step_b, size_b, act, grad, alpha, scale, y, s);
        default: return ggs_fail_(GGS_ERR_ARG, "ggs_fused_bias_act: dtype %d is not float / half / double", dtype);
    }
}

int ggs_fused_bias_act(size_t n, const float* x, const float* bias, const float* ref, int step_b, int size_b,
                       int act, int grad, float alpha, float scale, float* y, void* stream) {
    return ggs_fused_bias_act_t(GGS_DTYPE_F32, n, x, bias, ref, step_b, size_b, act, grad, alpha, scale, y, stream);
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

int ggs_upfirdn2d_t(int dtype, int major, int in_h, int in_w, int minor, const void* input, const void* kernel, int kh,
                    int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                    void* out, void* stream) {
    int out_h = 0, out_w = 0;
    int rc = ggs_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                                    &out_h, &out_w);
    if (rc != GGS_OK) return rc;
    if (major < 0 || in_h < 0 || in_w < 0 || minor < 0) return ggs_fail_(GGS_ERR_ARG, "ggs_upfirdn2d: bad sizes");
    const size_t total = (size_t)major * (size_t)(out_h > 0 ? out_h : 0) * (size_t)(out_w > 0 ? out_w : 0) * (size_t)minor;
    if (total == 0) return GGS_OK;
    if (!input || !kernel || !out) return ggs_fail_(GGS_ERR_ARG, "ggs_upfirdn2d: NULL pointer argument");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case GGS_DTYPE_F32: return run_upfirdn<float>(major, in_h, in_w, minor, input, kernel, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, out, s);
        case GGS_DTYPE_F16: return run_upfirdn<__half>(major, in_h, in_w, minor, input, kernel, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, out, s);
        case GGS_DTYPE_F64: return run_upfirdn<double>(major, in_h, in_w, minor, input, kernel, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w, out, s);
        default: return ggs_fail_(GGS_ERR_ARG, "ggs_upfirdn2d: dtype %d is not float / half / double", dtype);
    }
}

int ggs_upfirdn2d(int major, int in_h, int in_w, int minor, const float* input, const float* kernel, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                  float* out, void* stream) {
    return ggs_upfirdn2d_t(GGS_DTYPE_F32, major, in_h, in_w, minor, input, kernel, kh, kw, up_x, up_y, down_x, down_y,
                           pad_x0, pad_x1, pad_y0, pad_y1, out, stream);
}

}  // extern "C"
