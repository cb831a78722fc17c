// ggs_api.hip -- the extern "C" boundary of libggsplat.so (see include/ggsplat.h).
// Host-side only: argument checks, workspace carving, kernel launches on the caller's
// stream.  No allocation, no synchronisation (unless GgsParams.debug), no exceptions.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ggs_kernels.h"

namespace {

thread_local char g_err[512] = "";

}  // namespace

// Records the thread-local error message; shared by every translation unit of the library.
int ggs_fail_(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
void ggs_clear_error_() { g_err[0] = 0; }

namespace {
#define fail ggs_fail_

// ---- optional per-kernel timing (bench.py roofline leg) ------------------------------------
enum { K_PRE = 0, K_SCAN, K_SCATTER, K_SORT, K_RENDER_FWD, K_RENDER_BWD, K_PRE_BWD, K_ORDER, K_COUNT,
       K_ZERO = K_COUNT };       // (the two zero fills: bracketed in the timestamp mode only -- ggs_profile_read keeps its eight values)
// Second mode (ggs_profile_stamps): instead of host events, a one-lane kernel in front of and behind every bracketed kernel writes the
// device's constant-rate clock into the caller's buffer.  Those are ordinary launches: they are CAPTURED with the step and replayed
// with it, so the per-kernel intervals come from the replayed graph itself, not from eager launches with host events between them.
#define GGS_MAX_STAMPS 4096
struct Profile {
    bool on = false, have = false;
    hipEvent_t ev[K_COUNT][2];
    bool used[K_COUNT] = {};
    float ms[K_COUNT] = {};
    unsigned long long* stamps = nullptr;       // device buffer, one slot per stamp
    int stamp_cap = 0, n_stamps = 0, n_dropped = 0;
    short stamp_id[GGS_MAX_STAMPS];             // 2 kernel + (0 start | 1 stop)
};
thread_local Profile g_prof;

__global__ void k_stamp(unsigned long long* slot) { *slot = wall_clock64(); }

void prof_stamp(int id, hipStream_t s) {
    Profile& p = g_prof;
    if (p.n_stamps >= p.stamp_cap || p.n_stamps >= GGS_MAX_STAMPS) { ++p.n_dropped; return; }      // full: counted, not written
    p.stamp_id[p.n_stamps] = (short)id;
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, s, p.stamps + p.n_stamps);
    ++p.n_stamps;
}

void prof_start(int k, hipStream_t s) {
    if (g_prof.stamps) { prof_stamp(2 * k, s); return; }
    if (!g_prof.on) return;
    if (!g_prof.have) {
        for (int i = 0; i < K_COUNT; ++i) { hipEventCreate(&g_prof.ev[i][0]); hipEventCreate(&g_prof.ev[i][1]); }
        g_prof.have = true;
    }
    hipEventRecord(g_prof.ev[k][0], s);
}
void prof_stop(int k, hipStream_t s) {
    if (g_prof.stamps) { prof_stamp(2 * k + 1, s); return; }
    if (!g_prof.on) return;
    hipEventRecord(g_prof.ev[k][1], s);
    g_prof.used[k] = true;
}
void prof_collect(hipStream_t s) {
    if (!g_prof.on || g_prof.stamps) return;
    hipStreamSynchronize(s);
    for (int i = 0; i < K_COUNT; ++i)
        if (g_prof.used[i]) { hipEventElapsedTime(&g_prof.ms[i], g_prof.ev[i][0], g_prof.ev[i][1]); g_prof.used[i] = false; }
}

// After each launch: always catch launch errors; in debug mode also sync and catch execution errors.
int check(const char* what, hipStream_t s, int debug) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GGS_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
    if (debug) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(GGS_ERR_HIP, "%s: execution failed: %s", what, hipGetErrorString(e));
    }
    return GGS_OK;
}

#define GGS_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != GGS_OK) return rc__; \
    } while (0)

int check_params(const GgsParams* p) {
    if (!p) return fail(GGS_ERR_ARG, "params is NULL");
    if (p->P < 0 || p->W <= 0 || p->H <= 0 || p->n_views <= 0)
        return fail(GGS_ERR_ARG, "bad sizes P=%d W=%d H=%d n_views=%d", p->P, p->W, p->H, p->n_views);
    if (p->n_views > 65535) return fail(GGS_ERR_SIZE, "n_views=%d exceeds the grid.y limit 65535", p->n_views);
    if ((size_t)p->P * (size_t)p->n_views > (size_t)1 << 40) return fail(GGS_ERR_SIZE, "P*n_views too large");
    if (p->P >= (1 << GGS_ID_BITS)) return fail(GGS_ERR_SIZE, "P=%d exceeds the 2^%d id space of the tile lists", p->P, (int)GGS_ID_BITS);
    if (p->W > 32767 || p->H > 32767) return fail(GGS_ERR_SIZE, "image %dx%d exceeds the 16-bit pixel boxes of the records", p->W, p->H);
    const size_t tiles = (size_t)((p->W + GGS_TILE_W - 1) / GGS_TILE_W) * (size_t)((p->H + GGS_TILE - 1) / GGS_TILE);
    if (tiles * (size_t)p->n_views >= ((size_t)1 << 31))
        return fail(GGS_ERR_SIZE, "n_views * tiles = %zu work items exceed the launch grid", tiles * (size_t)p->n_views);
    return GGS_OK;
}

int check_modes(const GgsParams* p, const void* shs, const void* colors, const void* scales, const void* rots,
                const void* cov) {
    if (p->P == 0) return GGS_OK;   // nothing to draw: empty inputs carry no mode
    if ((shs != nullptr) == (colors != nullptr))
        return fail(GGS_ERR_ARG, "Please provide excatly one of either SHs or precomputed colors!");
    if (((scales != nullptr) || (rots != nullptr)) == (cov != nullptr) || ((scales != nullptr) != (rots != nullptr)))
        return fail(GGS_ERR_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (shs) {
        if (p->sh_degree < 0 || p->sh_degree > 3) return fail(GGS_ERR_ARG, "sh_degree=%d not in 0..3", p->sh_degree);
        if ((p->sh_degree + 1) * (p->sh_degree + 1) > p->K)
            return fail(GGS_ERR_ARG, "sh_degree=%d needs %d coefficients, shs has K=%d", p->sh_degree,
                        (p->sh_degree + 1) * (p->sh_degree + 1), p->K);
    }
    return GGS_OK;
}

// View-loop splits of the per-Gaussian backward: ~4k waves (4 per SIMD) keep 256 CUs busy.  (Tuning knob: GGS_BWD_WAVES.)
int bwd_splits(const GgsParams* p) {
    if (p->P <= 0) return 1;
    static const int target = [] {
        const char* e = getenv("GGS_BWD_WAVES");
        return e && *e ? atoi(e) : 4096;    // measured: 4096 beats 8192 and 2048 at K = 1 and K = 16 (tools/dbg/sweep_bwd_waves.sh)
    }();
    const int waves = (p->P + 63) / 64;
    int s = (target + waves - 1) / waves;
    return s < 1 ? 1 : (s > p->n_views ? p->n_views : s);
}

// launches with fewer (view, tile) work items than this use the one-wave-per-quadrant render kernels
#define GGS_QUAD_ITEMS_DEFAULT (57344 / GGS_TS)       // measured crossover at config 2: ~7 views of 8160 tiles (tools/dbg/quad_threshold.py)
// (tuning knob: the environment variable GGS_QUAD_ITEMS overrides the threshold; read once)
static int quad_items() {
    static const int v = [] {
        const char* e = getenv("GGS_QUAD_ITEMS");
        return e && *e ? atoi(e) : GGS_QUAD_ITEMS_DEFAULT;
    }();
    return v;
}
#define GGS_QUAD_ITEMS quad_items()

struct Dims { int gx, gy, T; };
Dims dims(const GgsParams* p) {
    Dims d;
    d.gx = (p->W + GGS_TILE_W - 1) / GGS_TILE_W;
    d.gy = (p->H + GGS_TILE - 1) / GGS_TILE;
    d.T = d.gx * d.gy;
    return d;
}

}  // namespace

extern "C" {

const char* ggs_last_error(void) { return g_err; }
const char* ggs_version(void) { return "ggsplat 0.2 gfx950"; }
#ifndef GGS_SRC_HASH
#define GGS_SRC_HASH "unknown"
#endif
const char* ggs_build_id(void) { return GGS_SRC_HASH; }
int ggs_tile_size(int* width, int* height) {
    if (width) *width = GGS_TILE_W;
    if (height) *height = GGS_TILE;
    return GGS_OK;
}

int ggs_profile_enable(int on) { g_prof.on = on != 0; return GGS_OK; }

int ggs_profile_stamps(void* device_slots, int capacity) {
    g_err[0] = 0;
    if (device_slots && capacity <= 0) return fail(GGS_ERR_ARG, "ggs_profile_stamps: capacity=%d", capacity);
    g_prof.stamps = (unsigned long long*)device_slots;
    g_prof.stamp_cap = device_slots ? capacity : 0;
    g_prof.n_stamps = g_prof.n_dropped = 0;
    return GGS_OK;
}
int ggs_profile_stamp_log(int* ids, int capacity, int* clock_khz) {
    g_err[0] = 0;
    if (clock_khz) {
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess)
            return fail(GGS_ERR_HIP, "ggs_profile_stamp_log: no wall clock rate: %s", hipGetErrorString(hipGetLastError()));
        *clock_khz = khz;
    }
    for (int i = 0; ids && i < g_prof.n_stamps && i < capacity; ++i) ids[i] = g_prof.stamp_id[i];
    return g_prof.n_stamps + g_prof.n_dropped;        // stamps ATTEMPTED: more than the slot capacity means the tail was dropped
}

int ggs_profile_read(float* ms, int n) {
    if (!ms || n < K_COUNT) return fail(GGS_ERR_ARG, "ggs_profile_read: need room for %d floats", (int)K_COUNT);
    for (int i = 0; i < K_COUNT; ++i) ms[i] = g_prof.ms[i];
    return K_COUNT;
}

int ggs_workspace_sizes(const GgsParams* p, size_t bin_capacity, size_t* geom_bytes, size_t* img_bytes,
                        size_t* bin_bytes) {
    GGS_TRY(check_params(p));
    const Dims d = dims(p);
    const size_t V = (size_t)p->n_views;
    if (geom_bytes) *geom_bytes = ggs_align(V * (size_t)p->P * sizeof(SplatRec)) + ggs_align(V * (size_t)p->P * sizeof(SplatAux));
    const BinLayout L = ggs_bin_layout(p->n_views, d.T, bin_capacity);
    // img: final_T | n_contrib
    // + the checkpoints of the segmented backward when this shape runs the latency mapping (ggs_common.h GGS_SEG)
    if (img_bytes) *img_bytes = ggs_align(V * (size_t)p->W * p->H * 4) * 2 +
                                ((long long)V * d.T < (long long)GGS_QUAD_ITEMS ? ggs_align(ggs_ckpt_bytes(bin_capacity)) : 0);
    if (bin_bytes) *bin_bytes = L.total;
    return GGS_OK;
}

int ggs_bin_layout(const GgsParams* p, size_t bin_capacity, size_t offsets[8]) {
    GGS_TRY(check_params(p));
    if (!offsets) return fail(GGS_ERR_ARG, "offsets is NULL");
    const BinLayout L = ggs_bin_layout(p->n_views, dims(p).T, bin_capacity);
    offsets[0] = L.header; offsets[1] = L.tile_count; offsets[2] = L.tile_cursor; offsets[3] = L.tile_offset;
    offsets[4] = L.view_base; offsets[5] = L.keys; offsets[6] = L.ids; offsets[7] = L.total;
    return GGS_OK;
}

}  // extern "C"

namespace {
int count_pairs_impl(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity, const void* img,
                     unsigned long long* count, int n_out, void* stream_) {
    g_err[0] = 0;
    GGS_TRY(check_params(p));
    if (!geom || !bin || !img || !count) return fail(GGS_ERR_ARG, "ggs_count_blends: NULL pointer argument");
    hipStream_t s = (hipStream_t)stream_;
    if (ggs_zero_async(count, 8 * (size_t)n_out, s) != hipSuccess) return fail(GGS_ERR_HIP, "ggs_count_blends: clearing the counter failed");
    if (p->P == 0) return GGS_OK;
    const Dims d = dims(p);
    const int V = p->n_views;
    const BinLayout L = ggs_bin_layout(V, d.T, bin_capacity);
    const char* b = (const char*)bin;
    RenderBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.P = p->P; a.W = p->W; a.H = p->H; a.gx = d.gx; a.gy = d.gy; a.T = d.T; a.n_items = V * d.T;
    a.tile_count = (const uint32_t*)(b + L.tile_count);
    a.tile_offset = (const uint32_t*)(b + L.tile_offset);
    a.view_base = (const unsigned long long*)(b + L.view_base);
    a.ids = (const uint32_t*)(b + L.ids);
    a.rec = (const SplatRec*)geom;
    a.n_contrib = (const uint32_t*)((const char*)img + ggs_align((size_t)V * p->W * p->H * 4));
    a.header = (const GgsBinHeader*)(b + L.header);
    hipLaunchKernelGGL(ggs_k_count_blends, dim3((unsigned)a.n_items), dim3(64), 0, s, a, count, n_out);
    return check("count_blends", s, p->debug);
}
}  // namespace

extern "C" {
int ggs_count_blends(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity, const void* img,
                     unsigned long long* count, void* stream_) {
    return count_pairs_impl(p, geom, bin, bin_capacity, img, count, 1, stream_);
}
int ggs_count_pairs(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity, const void* img,
                    unsigned long long* counts4, void* stream_) {
    return count_pairs_impl(p, geom, bin, bin_capacity, img, counts4, 4, stream_);
}
int ggs_count_forward_visits(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity,
                             unsigned long long* counts3, void* stream_) {
    g_err[0] = 0;
    GGS_TRY(check_params(p));
    if (!geom || !bin || !counts3) return fail(GGS_ERR_ARG, "ggs_count_forward_visits: NULL pointer argument");
    hipStream_t s = (hipStream_t)stream_;
    if (ggs_zero_async(counts3, 24, s) != hipSuccess) return fail(GGS_ERR_HIP, "ggs_count_forward_visits: clearing the counters failed");
    if (p->P == 0) return GGS_OK;
    const Dims d = dims(p);
    const int V = p->n_views;
    const BinLayout L = ggs_bin_layout(V, d.T, bin_capacity);
    char* b = (char*)bin;
    RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.P = p->P; a.W = p->W; a.H = p->H; a.gx = d.gx; a.gy = d.gy; a.T = d.T; a.n_items = V * d.T;
    a.header = (const GgsBinHeader*)(b + L.header);
    a.order = (const uint32_t*)(b + L.order);
    a.tile_count = (const uint32_t*)(b + L.tile_count);
    a.tile_offset = (const uint32_t*)(b + L.tile_offset);
    a.view_base = (const unsigned long long*)(b + L.view_base);
    a.ids = (uint32_t*)(b + L.ids);
    a.rec = (const SplatRec*)geom;
    hipLaunchKernelGGL(ggs_k_count_forward_visits, dim3((unsigned)a.n_items), dim3(64), 0, s, a, counts3);
    return check("count_forward_visits", s, p->debug);
}

// Leading bytes of `bin` / of the backward's scratch that ggs_forward* / ggs_backward zero-fill first (what a
// ggs_step_prologue must list to take those two launches over).
int ggs_step_clear_plan(const GgsParams* p, size_t bin_capacity, size_t* bin_bytes, size_t* backward_scratch_bytes) {
    g_err[0] = 0;
    GGS_TRY(check_params(p));
    if (!bin_bytes || !backward_scratch_bytes) return fail(GGS_ERR_ARG, "ggs_step_clear_plan: NULL output pointer");
    const Dims d = dims(p);
    *bin_bytes = ggs_bin_layout(p->n_views, d.T, bin_capacity).zero_bytes;
    *backward_scratch_bytes = (size_t)p->n_views * (size_t)p->P * sizeof(GradRec);
    return GGS_OK;
}

// Device address of page-locked host memory (hipHostMalloc / hipHostRegister with the mapped flag: what PyTorch's pinned
// tensors are), for kernels that read or write a small per-iteration block in place instead of behind a copy launch.
int ggs_host_mapped_pointer(void* host_ptr, void** device_ptr) {
    g_err[0] = 0;
    if (!host_ptr || !device_ptr) return fail(GGS_ERR_ARG, "ggs_host_mapped_pointer: NULL argument");
    *device_ptr = nullptr;
    const hipError_t e = hipHostGetDevicePointer(device_ptr, host_ptr, 0);
    if (e != hipSuccess || !*device_ptr) {
        (void)hipGetLastError();
        return fail(GGS_ERR_HIP, "ggs_host_mapped_pointer: not mapped page-locked memory: %s", hipGetErrorString(e));
    }
    return GGS_OK;
}

size_t ggs_backward_scratch_bytes(const GgsParams* p) {
    if (!p || p->P < 0 || p->n_views <= 0) return 0;
    const int splits = bwd_splits(p);
    const size_t part = splits > 1 ? (size_t)splits * (14 + 3 * (size_t)p->K) * (size_t)p->P * 4 : 0;
    return ggs_align((size_t)p->n_views * (size_t)p->P * sizeof(GradRec)) + ggs_align(part);
}

}  // extern "C"

namespace {
__global__ __launch_bounds__(256) void k_zero(uint32_t* p, size_t n_words, size_t head, size_t n_vec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint4* v = reinterpret_cast<uint4*>(p + head);
    for (size_t k = i; k < n_vec; k += stride) v[k] = make_uint4(0, 0, 0, 0);
    if (i < head) p[i] = 0;                                             // words in front of the 16-byte boundary
    const size_t tail0 = head + n_vec * 4;
    if (i < n_words - tail0) p[tail0 + i] = 0;                          // < 4 words behind the vector part
}
}  // namespace

// Pre-clear marks (ggs_step_prologue, include/ggsplat.h): ranges the prologue launch of this host thread has already
// zero-filled on a stream.  A fill of exactly such a range (same stream, same start, no longer) consumes the mark instead of
// launching -- every zero fill of the library goes through ggs_zero_async, so the consumers need no flag.
namespace {
struct ClearMark { hipStream_t s; char* p; size_t bytes; };
thread_local ClearMark g_marks[GGS_PROLOGUE_MAX_CLEAR];
thread_local int g_n_marks = 0;
bool take_mark(void* ptr, size_t bytes, hipStream_t s) {
    for (int i = 0; i < g_n_marks; ++i)
        if (g_marks[i].s == s && g_marks[i].p == (char*)ptr && bytes <= g_marks[i].bytes) {
            g_marks[i] = g_marks[--g_n_marks];
            return true;
        }
    return false;
}
}  // namespace
void ggs_set_clear_marks_(int n, void* const* ptrs, const size_t* bytes, hipStream_t s) {
    g_n_marks = 0;                                                      // marks nobody consumed are dropped here
    for (int i = 0; i < n && i < GGS_PROLOGUE_MAX_CLEAR; ++i)
        if (ptrs[i] && bytes[i]) g_marks[g_n_marks++] = ClearMark{s, (char*)ptrs[i], bytes[i]};
}

void ggs_drop_clear_marks_() { g_n_marks = 0; }
extern "C" int ggs_step_end(void) { g_n_marks = 0; return GGS_OK; }

hipError_t ggs_zero_async(void* ptr, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (g_n_marks && take_mark(ptr, bytes, s)) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(ptr) & 3) || (bytes & 3)) return hipErrorInvalidValue;
    const size_t n_words = bytes / 4;
    size_t head = ((16 - (reinterpret_cast<uintptr_t>(ptr) & 15)) & 15) / 4;
    if (head > n_words) head = n_words;
    const size_t n_vec = (n_words - head) / 4;
    size_t blocks = (n_vec + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(k_zero, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint32_t*>(ptr), n_words, head, n_vec);
    return hipGetLastError();
}

namespace {
enum { PHASE_COUNT = GGS_STAGE_COUNT, PHASE_BIN = GGS_STAGE_BIN, PHASE_COMPOSITE = GGS_STAGE_COMPOSITE,
       PHASE_RENDER = GGS_STAGE_BIN | GGS_STAGE_COMPOSITE };

// PHASE_COUNT    : clear counters, preprocess (+ tile histogram), scan, work-item order  -> header.num_rendered
// PHASE_BIN      : scatter keys, per-tile sort
// PHASE_COMPOSITE: composite (the only stage that writes the outputs and the per-pixel workspace)
int forward_impl(int phases, const GgsParams* p, const float* bg, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                 const float* tanfov, void* geom, void* bin, size_t bin_capacity, void* img, float* out_color,
                 float* out_depth, float* out_alpha, int* radii, void* stream_) {
    g_err[0] = 0;
    GGS_TRY(check_params(p));
    GGS_TRY(check_modes(p, shs, colors_precomp, scales, rotations, cov3D_precomp));
    if (!bg || !view || !proj || !campos || !tanfov || !bin || !img || !out_color || !out_depth ||
        !out_alpha || (p->P > 0 && (!means3D || !opacities || !radii || !geom)))
        return fail(GGS_ERR_ARG, "ggs_forward: NULL pointer argument");
    hipStream_t s = (hipStream_t)stream_;
    const Dims d = dims(p);
    const int V = p->n_views;
    const BinLayout L = ggs_bin_layout(V, d.T, bin_capacity);
    char* b = (char*)bin;
    GgsBinHeader* header = (GgsBinHeader*)(b + L.header);
    uint32_t* tile_count = (uint32_t*)(b + L.tile_count);
    uint32_t* tile_cursor = (uint32_t*)(b + L.tile_cursor);
    uint32_t* tile_offset = (uint32_t*)(b + L.tile_offset);
    unsigned long long* view_base = (unsigned long long*)(b + L.view_base);
    unsigned long long* keys = (unsigned long long*)(b + L.keys);
    uint32_t* ids = (uint32_t*)(b + L.ids);
    uint32_t* order = (uint32_t*)(b + L.order);
    uint32_t* bucket_count = (uint32_t*)(b + L.header + GGS_BUCKET_COUNT_OFF);
    uint32_t* bucket_cursor = (uint32_t*)(b + L.header + GGS_BUCKET_CURSOR_OFF);
    uint32_t* region_count = (uint32_t*)(b + L.header + GGS_REGION_COUNT_OFF);
    uint32_t* region_cursor = (uint32_t*)(b + L.header + GGS_REGION_CURSOR_OFF);
    const int n_items = V * d.T;
    const size_t HW = (size_t)p->W * p->H;
    float* final_T = (float*)img;
    uint32_t* n_contrib = (uint32_t*)((char*)img + ggs_align((size_t)V * HW * 4));

    const dim3 gridP((unsigned)((p->P + 255) / 256), (unsigned)V);
    if (phases & PHASE_COUNT) {
    if (g_prof.stamps) prof_stamp(2 * K_ZERO, s);
    if (ggs_zero_async(bin, L.zero_bytes, s) != hipSuccess)
        return fail(GGS_ERR_HIP, "ggs_forward: clearing the binning counters failed: %s", hipGetErrorString(hipGetLastError()));
    if (g_prof.stamps) prof_stamp(2 * K_ZERO + 1, s);
    if (p->P > 0) {
        PreArgs a;
        a.P = p->P; a.K = p->K; a.deg = p->sh_degree; a.W = p->W; a.H = p->H; a.gx = d.gx; a.gy = d.gy; a.T = d.T;
        a.scale_modifier = p->scale_modifier;
        a.means3D = means3D; a.shs = shs; a.colors = colors_precomp; a.opacities = opacities;
        a.scales = scales; a.rots = rotations; a.cov3d = cov3D_precomp;
        a.view = view; a.proj = proj; a.campos = campos; a.tanfov = tanfov;
        a.rec = (SplatRec*)geom; a.aux = (SplatAux*)((char*)geom + ggs_align((size_t)V * p->P * sizeof(SplatRec)));
        a.radii = radii; a.tile_count = tile_count;
        prof_start(K_PRE, s);
        // precomputed colours / SH degree 0 (the s2 setting): the specialisation without the degree 1-3 colour paths
        if (colors_precomp || p->sh_degree == 0) hipLaunchKernelGGL(ggs_k_preprocess_deg0, gridP, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(ggs_k_preprocess, gridP, dim3(256), 0, s, a);
        prof_stop(K_PRE, s);
        GGS_TRY(check("preprocess", s, p->debug));
    }
    {
        ScanArgs a;
        a.T = d.T; a.gx = d.gx; a.region_count = region_count; a.capacity = (unsigned long long)bin_capacity; a.tile_count = tile_count;
        a.tile_offset = tile_offset; a.view_base = view_base; a.header = header; a.bucket_count = bucket_count;
        if (V == 1) {             // one view: scan and work-item order in one launch (ggs_k_scan_order_one)
            prof_start(K_SCAN, s);
            hipLaunchKernelGGL(ggs_k_scan_order_one, dim3(1), dim3(1024), 0, s, a, order);
            prof_stop(K_SCAN, s);
            GGS_TRY(check("scan_order_one", s, p->debug));
        } else {
        prof_start(K_SCAN, s);
        hipLaunchKernelGGL(ggs_k_scan_tiles, dim3((unsigned)V), dim3(1024), 0, s, a);
        prof_stop(K_SCAN, s);
        GGS_TRY(check("scan_tiles", s, p->debug));
        OrderArgs o;
        o.n_items = n_items; o.tile_count = tile_count; o.bucket_count = bucket_count;
        o.bucket_cursor = bucket_cursor; o.order = order;
        o.T = d.T; o.gx = d.gx; o.region_count = region_count; o.region_cursor = region_cursor;
        prof_start(K_ORDER, s);
        hipLaunchKernelGGL(ggs_k_order_tiles, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, o);
        prof_stop(K_ORDER, s);
        GGS_TRY(check("order_tiles", s, p->debug));
        }
    }
    }  // PHASE_COUNT
    if (!(phases & PHASE_RENDER)) { prof_collect(s); return GGS_OK; }
    if (phases & PHASE_BIN) {
    if (p->P > 0) {
        ScatterArgs a;
        a.P = p->P; a.gx = d.gx; a.gy = d.gy; a.T = d.T; a.gx16 = (p->W + 15) / 16; a.rec = (const SplatRec*)geom; a.header = header;
        a.aux = (const SplatAux*)((const char*)geom + ggs_align((size_t)V * p->P * sizeof(SplatRec)));
        a.tile_cursor = tile_cursor; a.tile_offset = tile_offset; a.view_base = view_base; a.keys = keys;
        prof_start(K_SCATTER, s);
        hipLaunchKernelGGL(ggs_k_scatter, gridP, dim3(256), 0, s, a);
        prof_stop(K_SCATTER, s);
        GGS_TRY(check("scatter", s, p->debug));
    }
    {
        SortArgs a;
        a.T = d.T; a.n_items = n_items; a.order = order; a.bucket_count = bucket_count; a.header = header;
        a.tile_count = tile_count; a.tile_offset = tile_offset; a.view_base = view_base; a.keys = keys; a.ids = ids;
        prof_start(K_SORT, s);
        // persistent grids sized for the chip (256 CUs), not for the item count: most items are empty tiles
        const unsigned n_block = (unsigned)(n_items < 1280 ? n_items : 1280);
        const unsigned n_wave = (unsigned)((n_items + 3) / 4 < 2048 ? (n_items + 3) / 4 : 2048);
        if (n_items < GGS_QUAD_ITEMS) {
            // small launch (latency bound): both list classes in one grid so the two passes overlap
            hipLaunchKernelGGL(ggs_k_sort_tiles, dim3(n_block + n_wave), dim3(256), 0, s, a, n_block);
        } else {
            hipLaunchKernelGGL(ggs_k_sort_tiles, dim3(n_block), dim3(256), 0, s, a, n_block);
            hipLaunchKernelGGL(ggs_k_sort_tiles_wave, dim3(n_wave), dim3(256), 0, s, a);
        }
        prof_stop(K_SORT, s);
        GGS_TRY(check("sort_tiles", s, p->debug));
    }
    }  // PHASE_BIN
    if (phases & PHASE_COMPOSITE) {
        RenderArgs a;
        a.P = p->P; a.W = p->W; a.H = p->H; a.gx = d.gx; a.gy = d.gy; a.T = d.T; a.header = header; a.poison = p->debug;
        a.n_items = n_items; a.order = order; a.tile_count = tile_count; a.tile_offset = tile_offset; a.view_base = view_base; a.ids = ids;
        a.rec = (const SplatRec*)geom; a.bg = bg; a.out_color = out_color; a.out_depth = out_depth;
        a.out_alpha = out_alpha; a.final_T = final_T; a.n_contrib = n_contrib;
        a.ckpt = n_items < GGS_QUAD_ITEMS ? (float*)((char*)img + 2 * ggs_align((size_t)V * HW * 4)) : nullptr;
        a.ckpt_slots = (unsigned)ggs_ckpt_slots(bin_capacity);
        prof_start(K_RENDER_FWD, s);
        if (n_items < GGS_QUAD_ITEMS) hipLaunchKernelGGL(ggs_k_render_fwd_quad, dim3((unsigned)n_items * GGS_NQ), dim3(64), 0, s, a);
        else hipLaunchKernelGGL(ggs_k_render_fwd, dim3((unsigned)n_items), dim3(64), 0, s, a);
        prof_stop(K_RENDER_FWD, s);
        GGS_TRY(check("render_fwd", s, p->debug));
    }
    prof_collect(s);
    return GGS_OK;
}
}  // namespace

extern "C" {

#define GGS_FWD_PARAMS                                                                                              \
    const GgsParams *p, const float *bg, const float *means3D, const float *shs, const float *colors_precomp,      \
        const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,           \
        const float *view, const float *proj, const float *campos, const float *tanfov, void *geom, void *bin,     \
        size_t bin_capacity, void *img, float *out_color, float *out_depth, float *out_alpha, int *radii,          \
        void *stream_
#define GGS_FWD_ARGS                                                                                                \
    p, bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view, proj, campos, tanfov,   \
        geom, bin, bin_capacity, img, out_color, out_depth, out_alpha, radii, stream_

int ggs_forward(GGS_FWD_PARAMS) { return forward_impl(PHASE_COUNT | PHASE_RENDER, GGS_FWD_ARGS); }
int ggs_forward_count(GGS_FWD_PARAMS) { return forward_impl(PHASE_COUNT, GGS_FWD_ARGS); }
int ggs_forward_render(GGS_FWD_PARAMS) { return forward_impl(PHASE_RENDER, GGS_FWD_ARGS); }
int ggs_forward_stages(int stages, GGS_FWD_PARAMS) {
    if (stages <= 0 || (stages & ~(PHASE_COUNT | PHASE_RENDER)))
        return fail(GGS_ERR_ARG, "ggs_forward_stages: stages=%d is not a combination of GGS_STAGE_COUNT | _BIN | _COMPOSITE", stages);
    // a contiguous run only: COUNT | COMPOSITE would re-clear the counters and then composite against the lists of an earlier call
    if (stages == (GGS_STAGE_COUNT | GGS_STAGE_COMPOSITE))
        return fail(GGS_ERR_ARG, "ggs_forward_stages: GGS_STAGE_COUNT | GGS_STAGE_COMPOSITE skips GGS_STAGE_BIN (stages must be a contiguous run)");
    return forward_impl(stages, GGS_FWD_ARGS);
}
int ggs_forward_spec(GGS_FWD_PARAMS, void* host_header, void* header_event) {
    if (!host_header || !header_event) return fail(GGS_ERR_ARG, "ggs_forward_spec: NULL host_header / header_event");
    GGS_TRY(forward_impl(PHASE_COUNT, GGS_FWD_ARGS));
    hipStream_t s = (hipStream_t)stream_;
    // the header is the first 16 bytes of the binning buffer (ggs_bin_layout section 0)
    if (hipMemcpyAsync(host_header, bin, 16, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipEventRecord((hipEvent_t)header_event, s) != hipSuccess)
        return fail(GGS_ERR_HIP, "ggs_forward_spec: header copy / event record failed: %s", hipGetErrorString(hipGetLastError()));
    return forward_impl(PHASE_RENDER, GGS_FWD_ARGS);
}

int ggs_backward(const GgsParams* p, const float* bg, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                 const float* tanfov, const void* geom, const void* bin, size_t bin_capacity,
                 const void* img, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch,
                 float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs, float* dL_dcolors,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D, int accumulate, void* stream_) {
    g_err[0] = 0;
    GGS_TRY(check_params(p));
    GGS_TRY(check_modes(p, shs, colors_precomp, scales, rotations, cov3D_precomp));
    if (p->P == 0) return GGS_OK;
    if (!bg || !view || !proj || !campos || !tanfov || !geom || !bin || !img || !dL_dcolor || !scratch)
        return fail(GGS_ERR_ARG, "ggs_backward: NULL pointer argument");
    if (!means3D || !dL_dmeans3D || !dL_dopacities) return fail(GGS_ERR_ARG, "ggs_backward: NULL gradient output");
    if (shs && !dL_dshs) return fail(GGS_ERR_ARG, "ggs_backward: dL_dshs is NULL but shs given");
    if (colors_precomp && !dL_dcolors) return fail(GGS_ERR_ARG, "ggs_backward: dL_dcolors is NULL but colors given");
    if (cov3D_precomp && !dL_dcov3D) return fail(GGS_ERR_ARG, "ggs_backward: dL_dcov3D is NULL but cov3D given");
    if (scales && (!dL_dscales || !dL_drotations))
        return fail(GGS_ERR_ARG, "ggs_backward: dL_dscales / dL_drotations is NULL but scales given");
    hipStream_t s = (hipStream_t)stream_;
    const Dims d = dims(p);
    const int V = p->n_views;
    const BinLayout L = ggs_bin_layout(V, d.T, bin_capacity);
    const char* b = (const char*)bin;
    const size_t HW = (size_t)p->W * p->H;

    if (g_prof.stamps) prof_stamp(2 * K_ZERO, s);
    if (ggs_zero_async(scratch, (size_t)V * p->P * sizeof(GradRec), s) != hipSuccess)
        return fail(GGS_ERR_HIP, "ggs_backward: clearing the gradient records failed: %s", hipGetErrorString(hipGetLastError()));
    if (g_prof.stamps) prof_stamp(2 * K_ZERO + 1, s);
    {
        RenderBwdArgs a;
        a.P = p->P; a.W = p->W; a.H = p->H; a.gx = d.gx; a.gy = d.gy; a.T = d.T; a.poison = p->debug;
        a.n_items = V * d.T; a.order = (const uint32_t*)(b + L.order);
        a.bucket_count = (const uint32_t*)(b + L.header + GGS_BUCKET_COUNT_OFF);
        a.tile_count = (const uint32_t*)(b + L.tile_count);
        a.tile_offset = (const uint32_t*)(b + L.tile_offset);
        a.view_base = (const unsigned long long*)(b + L.view_base);
        a.ids = (const uint32_t*)(b + L.ids);
        a.rec = (const SplatRec*)geom; a.bg = bg;
        a.final_T = (const float*)img;
        a.n_contrib = (const uint32_t*)((const char*)img + ggs_align((size_t)V * HW * 4));
        a.dL_dcolor = dL_dcolor; a.dL_ddepth = dL_ddepth; a.dL_dalpha = dL_dalpha;
        a.acc = (GradRec*)scratch;
        a.header = (const GgsBinHeader*)(b + L.header);
        a.ckpt = a.n_items < GGS_QUAD_ITEMS ? (const float*)((const char*)img + 2 * ggs_align((size_t)V * HW * 4)) : nullptr;
        a.ckpt_slots = (unsigned)ggs_ckpt_slots(bin_capacity);
        const dim3 gridT((unsigned)(V * d.T));   // one wave64 per (view, tile) work item, longest lists first
        prof_start(K_RENDER_BWD, s);
        const bool da = dL_ddepth || dL_dalpha;
        if (a.n_items < GGS_QUAD_ITEMS && da) {      // latency mapping with depth / alpha gradients: unsegmented per-quadrant walks
            a.ckpt = nullptr;
            hipLaunchKernelGGL(ggs_k_render_bwd_da_quad, dim3((unsigned)(a.n_items * GGS_NQ)), dim3(64), 0, s, a);
        } else if (da) hipLaunchKernelGGL(ggs_k_render_bwd_da, gridT, dim3(64), 0, s, a);
        else hipLaunchKernelGGL(ggs_k_render_bwd, gridT, dim3(64), 0, s, a);
        prof_stop(K_RENDER_BWD, s);
        GGS_TRY(check("render_bwd", s, p->debug));
    }
    {
        PreBwdArgs a;
        a.P = p->P; a.K = p->K; a.deg = p->sh_degree; a.W = p->W; a.H = p->H; a.V = V; a.accumulate = accumulate;
        a.scale_modifier = p->scale_modifier;
        a.means3D = means3D; a.shs = shs; a.colors = colors_precomp; a.scales = scales; a.rots = rotations;
        a.cov3d = cov3D_precomp; a.view = view; a.proj = proj; a.campos = campos; a.tanfov = tanfov;
        a.rec = (const SplatRec*)geom; a.acc = (const GradRec*)scratch;
        a.aux = (const SplatAux*)((const char*)geom + ggs_align((size_t)V * p->P * sizeof(SplatRec)));
        a.dL_dmeans2D = dL_dmeans2D; a.dL_dmeans3D = dL_dmeans3D; a.dL_dopac = dL_dopacities; a.dL_dsh = dL_dshs;
        a.dL_dcolors = dL_dcolors; a.dL_dscales = dL_dscales; a.dL_drots = dL_drotations; a.dL_dcov3D = dL_dcov3D;
        prof_start(K_PRE_BWD, s);
        // enough lanes to fill 256 CUs: split the view loop when P alone gives < ~8k waves
        const int splits = bwd_splits(p);
        a.part = splits > 1 ? (float*)((char*)scratch + ggs_align((size_t)V * p->P * sizeof(GradRec))) : nullptr;
        const dim3 grid((unsigned)((p->P + 255) / 256), (unsigned)splits);
        switch (colors_precomp ? 0 : p->sh_degree) {
            case 0: hipLaunchKernelGGL(ggs_k_preprocess_bwd_sh0, grid, dim3(256), 0, s, a); break;
            case 1: hipLaunchKernelGGL(ggs_k_preprocess_bwd_sh1, grid, dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL(ggs_k_preprocess_bwd_sh2, grid, dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL(ggs_k_preprocess_bwd_sh3, grid, dim3(256), 0, s, a); break;
        }
        if (splits > 1) {
            const size_t lds = (size_t)64 * (14 + 3 * p->K + 1) * sizeof(float);      // 64 Gaussians x all components
            hipLaunchKernelGGL(ggs_k_reduce_partials, dim3((unsigned)((p->P + 63) / 64)), dim3(256), lds, s, a, splits);
        }
        prof_stop(K_PRE_BWD, s);
        GGS_TRY(check("preprocess_bwd", s, p->debug));
    }
    prof_collect(s);
    return GGS_OK;
}

}  // extern "C"
