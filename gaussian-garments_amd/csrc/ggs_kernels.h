// ggs_kernels.h -- kernel argument blocks and declarations (internal).
#pragma once
#include "ggs_common.h"

struct PreArgs {
    int P, K, deg, W, H, gx, gy, T;
    float scale_modifier;
    const float *means3D, *shs, *colors, *opacities, *scales, *rots, *cov3d;
    const float *view, *proj, *campos, *tanfov;
    SplatRec* rec;
    SplatAux* aux;
    int* radii;
    uint32_t* tile_count;
};

struct ScanArgs {
    int T, gx;
    uint32_t* region_count;   // [GGS_NBUCKET][GGS_NREGION]
    unsigned long long capacity;
    const uint32_t* tile_count;
    uint32_t* tile_offset;
    unsigned long long* view_base;
    GgsBinHeader* header;
    uint32_t* bucket_count;   // [GGS_NBUCKET]
};

struct OrderArgs {
    int n_items;              // V * T
    int T, gx;
    const uint32_t* region_count;     // [GGS_NBUCKET][GGS_NREGION]
    uint32_t* region_cursor;          // [GGS_NBUCKET][GGS_NREGION]
    const uint32_t* tile_count;
    const uint32_t* bucket_count;
    uint32_t* bucket_cursor;
    uint32_t* order;
};

struct ScatterArgs {
    int P, gx, gy, T, gx16;
    const SplatRec* rec;
    const SplatAux* aux;
    const GgsBinHeader* header;
    uint32_t* tile_cursor;
    const uint32_t* tile_offset;
    const unsigned long long* view_base;
    unsigned long long* keys;
};

struct SortArgs {
    int T, n_items;
    const uint32_t* order;
    const uint32_t* bucket_count;
    const GgsBinHeader* header;
    const uint32_t* tile_count;
    const uint32_t* tile_offset;
    const unsigned long long* view_base;
    unsigned long long* keys;
    uint32_t* ids;
};

struct RenderArgs {
    int P, W, H, gx, gy, T, n_items;
    int poison;               // GgsParams.debug: the per-quadrant walks fill their LDS record slice with NaNs before every round
    const uint32_t* order;
    const GgsBinHeader* header;
    const uint32_t* tile_count;
    const uint32_t* tile_offset;
    const unsigned long long* view_base;
    uint32_t* ids;            // forward narrows the quadrant masks to "actually blended" in place
    const SplatRec* rec;
    const float* bg;          // [V][3]
    float* out_color;         // [V][3][H][W]
    float* out_depth;         // [V][H][W]
    float* out_alpha;         // [V][H][W]
    float* final_T;           // [V][H][W]
    uint32_t* n_contrib;      // [V][H][W]
    float* ckpt;              // latency mapping: checkpoints for the segmented backward (ggs_common.h GGS_SEG), else null
    unsigned ckpt_slots;
};

struct RenderBwdArgs {
    int P, W, H, gx, gy, T, n_items;
    int poison;               // as RenderArgs.poison
    const uint32_t* order;
    const uint32_t* bucket_count;     // [GGS_NBUCKET] list-length class histogram (position of the non-empty items)
    const uint32_t* tile_count;
    const uint32_t* tile_offset;
    const unsigned long long* view_base;
    const uint32_t* ids;
    const SplatRec* rec;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dcolor;   // [V][3][H][W]
    const float* dL_ddepth;   // [V][H][W] or null
    const float* dL_dalpha;   // [V][H][W] or null
    GradRec* acc;             // [V][P]
    const GgsBinHeader* header;   // overflow != 0: the forward did not composite -> the backward does nothing
    const float* ckpt;        // latency mapping: the forward's checkpoints (null: unsegmented walk)
    unsigned ckpt_slots;
};

struct PreBwdArgs {
    int P, K, deg, W, H, V, accumulate;
    float scale_modifier;
    const float *means3D, *shs, *colors, *scales, *rots, *cov3d;
    const float *view, *proj, *campos, *tanfov;
    const SplatRec* rec;
    const SplatAux* aux;
    const GradRec* acc;
    float *dL_dmeans2D, *dL_dmeans3D, *dL_dopac, *dL_dsh, *dL_dcolors, *dL_dscales, *dL_drots, *dL_dcov3D;
    float* part;              // [splits][14 + 3K][P] partial sums (only when the view loop is split)
};

__global__ void ggs_k_preprocess(PreArgs a);
__global__ void ggs_k_preprocess_deg0(PreArgs a);
__global__ void ggs_k_scan_tiles(ScanArgs a);
__global__ void ggs_k_scatter(ScatterArgs a);
__global__ void ggs_k_order_tiles(OrderArgs a);
__global__ void ggs_k_scan_order_one(ScanArgs a, uint32_t* order);
__global__ void ggs_k_sort_tiles(SortArgs a, unsigned n_block);
__global__ void ggs_k_sort_tiles_wave(SortArgs a);
__global__ void ggs_k_render_fwd(RenderArgs a);
__global__ void ggs_k_render_fwd_quad(RenderArgs a);
__global__ void ggs_k_render_bwd(RenderBwdArgs a);
__global__ void ggs_k_render_bwd_da(RenderBwdArgs a);
__global__ void ggs_k_render_bwd_da_quad(RenderBwdArgs a);
__global__ void ggs_k_count_blends(RenderBwdArgs a, unsigned long long* out, int n_out);
__global__ void ggs_k_count_forward_visits(RenderArgs a, unsigned long long* out);
__global__ void ggs_k_preprocess_bwd_sh0(PreBwdArgs a);
__global__ void ggs_k_preprocess_bwd_sh1(PreBwdArgs a);
__global__ void ggs_k_preprocess_bwd_sh2(PreBwdArgs a);
__global__ void ggs_k_preprocess_bwd_sh3(PreBwdArgs a);
__global__ void ggs_k_reduce_partials(PreBwdArgs a, int splits);

// Optimiser state of one tensor (ggs_adam.hip; advanced by its tick kernels, by the update kernel that ticks itself, or by the
// last kernel of ggs_registration_aux_tail).
// 40 bytes, device.  `ticket` counts the workgroups of ggs_adam_tick_step_multi that are done with this tensor (back to 0
// when the launch ends); everything in front of it is the state a checkpoint keeps.
struct AdamState { long long step; float bias1; float bias2_sqrt; double pow1; double pow2; unsigned ticket; unsigned pad; };

__device__ __forceinline__ void ggs_adam_tick_one(AdamState* s, double beta1, double beta2) {
    const long long t = s->step + 1;
    s->step = t;
    // beta^t as a running double product (torch evaluates beta ** step in Python floats each step; the products agree to
    // ~t * 2^-53, far below the fp32 the corrections are used in) -- a device-side pow() made this one-thread kernel
    // the slowest launch of a graph-replayed iteration after the rasterizer and the loss
    const double p1 = t == 1 ? beta1 : s->pow1 * beta1, p2 = t == 1 ? beta2 : s->pow2 * beta2;
    s->pow1 = p1; s->pow2 = p2;
    s->bias1 = (float)(1.0 - p1);
    s->bias2_sqrt = (float)sqrt(1.0 - p2);
}


// host-side error plumbing (ggs_api.hip)
int ggs_fail_(int code, const char* fmt, ...);
void ggs_clear_error_();

// Zero-fill as a KERNEL, not hipMemsetAsync: memset nodes of a captured hipGraph were observed not to re-execute
// faithfully on replay (counters kept their values from the previous replay -> list overflow, memory fault), and
// every entry point must behave identically whether it is launched eagerly or replayed from a graph.
// p must be 4-byte aligned, bytes a multiple of 4.  Returns hipSuccess / the launch error.
hipError_t ggs_zero_async(void* p, size_t bytes, hipStream_t s);
// ggs_step_prologue's marks: ranges already zero-filled on `s` by this host thread; ggs_zero_async consumes them (ggs_api.hip)
void ggs_set_clear_marks_(int n, void* const* ptrs, const size_t* bytes, hipStream_t s);
void ggs_drop_clear_marks_();        // end of a step: marks nobody consumed must not outlive it
