/*
 * ggsplat.h -- C ABI of libggsplat.so, the MI355X (gfx950) differentiable
 * Gaussian-splat rasterizer.  This is the drop-in boundary of the hot path:
 * exactly what the reference's FFI for this path binds.
 *
 * Reference interface each entry point replaces (paths relative to the
 * reference repo eth-ait/Gaussian-Garments):
 *
 *   ggs_forward   <- `_C.rasterize_gaussians` of the external CUDA extension
 *                    diff_gaussian_rasterization_depth_alpha, reached through
 *                    GaussianRasterizer.forward at gaussian_renderer/__init__.py:54,
 *                    :103-111 (settings built at :39-52).  Argument meaning,
 *                    layouts and return order (color, radii, depth, alpha) follow
 *                    that call site.
 *   ggs_backward  <- `_C.rasterize_gaussians_backward`, reached implicitly through
 *                    loss.backward() at s2_registration.py:306 / s3_appearance.py:141.
 *   ggs_workspace_sizes
 *                 <- the geomBuffer / binningBuffer / imgBuffer byte tensors the
 *                    upstream extension grows through a resize callback; here the
 *                    caller allocates (no callback, no host sync inside the ABI).
 *   ggs_mesh_bind_forward / ggs_mesh_bind_backward
 *                 <- MeshGaussianModel.update_face_coor + get_xyz / get_scaling /
 *                    get_rotation (scene/mesh_gaussian_model.py:90-95, 105-128) with
 *                    compute_face_orientation (utils/graphics_utils.py:118-137) and
 *                    the AvatarGaussianModel barycentric origin
 *                    (scene/avatar_gaussian_model.py:140-159).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch); the
 *     library never allocates, frees or retains them.  fp32 unless stated.
 *   - `n_views` >= 1 batches V cameras over the same Gaussians in one call
 *     (grid dimension = view).  Per-view arrays are [V][...] contiguous.
 *     V = 1 is exactly the reference's per-camera call.
 *   - Matrices arrive transposed and flat like the reference passes them
 *     (scene/cameras.py:59-61): x' = m[0]x + m[4]y + m[8]z + m[12].
 *   - Quaternions are (w,x,y,z), assumed unit.  SH layout [P][K][3].
 *   - All work is enqueued on `stream`; nothing synchronises unless
 *     GgsParams.debug != 0 (mirrors pipe.debug: sync + error check per kernel).
 *   - Return value: 0 on success, negative error code otherwise; the message is
 *     in a thread-local buffer (ggs_last_error).  No C++ exception crosses the ABI.
 *   - Re-entrant; no global mutable state besides the thread-local error string.
 *     One process per GPU for multi-GPU use.
 */
#ifndef GGSPLAT_H
#define GGSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GgsParams {
    int P;                /* number of Gaussians                                            */
    int K;                /* SH coefficients per Gaussian stored in `shs` (0 if colors given) */
    int sh_degree;        /* active SH degree, 0..3, (sh_degree+1)^2 <= K                    */
    int W, H;             /* image size in pixels                                           */
    int n_views;          /* V >= 1                                                         */
    float scale_modifier; /* GaussianRasterizationSettings.scale_modifier                   */
    int prefiltered;      /* accepted for API parity; the reference always passes False      */
    int debug;            /* != 0: hipStreamSynchronize + error check after every kernel     */
} GgsParams;

/* Layout of the first 16 bytes of the binning buffer: read it back (async copy)
 * to learn num_rendered and whether `bin_capacity` was too small. */
typedef struct GgsBinHeader {
    unsigned long long num_rendered; /* sum over views of the (splat, tile) list entries the library KEEPS: the
                                      * reference's 3-sigma-square rule minus the pairs whose alpha can never reach
                                      * 1/255 inside the tile (output-invariant culling; e.g. 337k vs 462k per view
                                      * at config 2).  It sizes `bin_capacity`; it is NOT the upstream num_rendered. */
    unsigned long long overflow;     /* != 0: N > bin_capacity, outputs are invalid; retry     */
} GgsBinHeader;

/* Error codes */
#define GGS_OK 0
#define GGS_ERR_ARG (-1)    /* bad argument combination (both/neither of shs|colors, scales+rots|cov3D, ...) */
#define GGS_ERR_HIP (-2)    /* a HIP call or kernel launch failed                                            */
#define GGS_ERR_SIZE (-3)   /* size does not fit the index types                                             */

/* Bytes the caller must provide for the three opaque workspaces.
 * bin_capacity = max number of (Gaussian, tile) instances over all views.
 * The workspaces are OPAQUE state for ggs_backward (upstream's geomBuffer / binningBuffer / imgBuffer).  In particular the
 * image workspace (final_T [V][H][W] f32 | n_contrib [V][H][W] u32) is DEFINED ONLY ON THE PIXELS OF TILES THAT HAVE A SPLAT
 * LIST: the forward does not store it for empty 16x16 tiles (nothing on the device reads it there; final_T = 1 and
 * n_contrib = 0 by definition), and a forward that overflowed bin_capacity leaves all of it undefined.  A caller that wants
 * per-pixel transmittance / contributor counts must patch empty tiles from tile_count (ggs_bin_layout section 1), as
 * ggsplat.rasterizer.img_sections() does -- or use out_alpha (1 - final_T up to the un-applied terminating splat). */
int ggs_workspace_sizes(const GgsParams* prm, size_t bin_capacity, size_t* geom_bytes, size_t* img_bytes,
                        size_t* bin_bytes);

/* Introspection (tests / debugging): byte offsets of the sections of the binning buffer, in order
 * {header, tile_count [V][T] u32, tile_cursor [V][T] u32, tile_offset [V][T] u32, view_base [V] u64,
 *  keys [cap] u64, ids [cap] u32, total}.  Not part of the reference's interface. */
int ggs_bin_layout(const GgsParams* prm, size_t bin_capacity, size_t offsets[8]);

/* Bytes of the scratch ggs_backward needs: V*P*48 for the per-(view, Gaussian) gradient accumulators, plus
 * the per-split partial sums of the per-Gaussian stage when its view loop is split (small P). */
size_t ggs_backward_scratch_bytes(const GgsParams* prm);

/*
 * Forward.  Exactly one of shs [P][K][3] / colors_precomp [P][3] and exactly one of
 * (scales [P][3], rotations [P][4]) / cov3D_precomp [P][6] must be non-NULL.
 *   bg [V][3], view [V][16], proj [V][16], campos [V][3], tanfov [V][2] = (tanfovx, tanfovy)
 *   out_color [V][3][H][W], out_depth [V][H][W], out_alpha [V][H][W], radii [V][P] int32
 * geom / bin / img: workspaces of ggs_workspace_sizes(); they carry the state backward needs.
 */
int ggs_forward(const GgsParams* prm, const float* bg, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                const float* tanfov, void* geom, void* bin, size_t bin_capacity, void* img, float* out_color,
                float* out_depth, float* out_alpha, int* radii, void* stream);

/*
 * Two-phase forward with the SAME argument list, for callers that want to size the binning buffer
 * exactly (what the upstream extension does through its resize callback after the tile scan):
 *   ggs_forward_count  : preprocess + tile histogram + scan -> header.num_rendered is final once the
 *                        stream reaches this point (read the header back; ~10 us of GPU work);
 *   ggs_forward_render : key scatter + per-tile sort + compositing, on the same geom/bin/img buffers.
 * If header.overflow is set after the count phase, call it again with a binning buffer of at least
 * num_rendered entries before rendering.  ggs_forward == count followed by render.
 */
int ggs_forward_count(const GgsParams* prm, const float* bg, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales,
                      const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                      const float* campos, const float* tanfov, void* geom, void* bin, size_t bin_capacity,
                      void* img, float* out_color, float* out_depth, float* out_alpha, int* radii, void* stream);
/*
 * The eager caller's form of the two phases in ONE call: count phase, then an asynchronous copy of the 16-byte header
 * {num_rendered, overflow} to `host_header` (pinned host memory) and `hipEventRecord(header_event, stream)`, then the render
 * phase queued right behind WITHOUT waiting (every kernel of it is guarded by the overflow word on the device).  The caller waits
 * for the event only (the GPU is already compositing), reads the two words, and repeats the call with a larger buffer in the rare
 * overflow case.  Same results as ggs_forward; same argument list plus the two handles.  (The upstream extension synchronises at
 * the same point to size its binning buffer through the resize callback.)
 */
int ggs_forward_spec(const GgsParams* prm, const float* bg, const float* means3D, const float* shs,
                     const float* colors_precomp, const float* opacities, const float* scales,
                     const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                     const float* campos, const float* tanfov, void* geom, void* bin, size_t bin_capacity,
                     void* img, float* out_color, float* out_depth, float* out_alpha, int* radii, void* stream,
                     void* host_header, void* header_event);
int ggs_forward_render(const GgsParams* prm, const float* bg, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales,
                       const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                       const float* campos, const float* tanfov, void* geom, void* bin, size_t bin_capacity,
                       void* img, float* out_color, float* out_depth, float* out_alpha, int* radii, void* stream);
/*
 * Any contiguous run of the forward's three stages, same argument list behind the stage mask:
 *   GGS_STAGE_COUNT     = ggs_forward_count;
 *   GGS_STAGE_BIN       = key scatter + per-tile sort (reads the records and counters of the count stage, writes the lists);
 *   GGS_STAGE_COMPOSITE = compositing (reads records + lists; the only stage that writes out_color / out_depth / out_alpha and
 *                         the per-pixel workspace `img`).
 * ggs_forward_render == GGS_STAGE_BIN | GGS_STAGE_COMPOSITE.  For callers that software-pipeline launch sets: the binning of
 * set i + 1 can be queued in front of the backward of set i and its compositing beside it on another stream
 * (ggsplat.batch.fwd_bwd_views(pipeline=2); measurements: profiles/r05_pipeline_overlap.md).  The stages of one forward must run in
 * order on streams ordered by the caller.  No reference counterpart (the upstream forward is one call).
 */
#define GGS_STAGE_COUNT 1
#define GGS_STAGE_BIN 2
#define GGS_STAGE_COMPOSITE 4
int ggs_forward_stages(int stages, const GgsParams* prm, const float* bg, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* opacities, const float* scales,
                       const float* rotations, const float* cov3D_precomp, const float* view, const float* proj,
                       const float* campos, const float* tanfov, void* geom, void* bin, size_t bin_capacity,
                       void* img, float* out_color, float* out_depth, float* out_alpha, int* radii, void* stream);

/*
 * Backward.  dL_dcolor [V][3][H][W]; dL_ddepth / dL_dalpha [V][H][W] or NULL (zero).
 * geom / bin / img / bin_capacity: exactly what the matching ggs_forward call was given.
 * scratch: ggs_backward_scratch_bytes() bytes.
 * Outputs (summed over the V views; overwritten unless accumulate != 0):
 *   dL_dmeans3D [P][3], dL_dopacities [P], and by input mode
 *   dL_dshs [P][K][3] | dL_dcolors [P][3];  (dL_dscales [P][3], dL_drotations [P][4]) | dL_dcov3D [P][6].
 *   Pointers of the unused mode may be NULL.
 * dL_dmeans2D [V][P][3] (per view, never summed; may be NULL): gradient w.r.t. the NDC xy of the
 *   projected mean (pixel gradient * 0.5W / 0.5H), z = 0 -- what lands in render()'s
 *   screenspace_points.grad (gaussian_renderer/__init__.py:29-33).
 */
int ggs_backward(const GgsParams* prm, const float* bg, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const float* view, const float* proj, const float* campos,
                 const float* tanfov, const void* geom, const void* bin, size_t bin_capacity,
                 const void* img, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha, void* scratch,
                 float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs, float* dL_dcolors,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D, int accumulate, void* stream);

/*
 * Mesh binding (local Gaussian frame -> world), forward and backward.
 *   verts [V_m][3], faces [F][3] int64, binding [P] int64 (face of each Gaussian)
 *   local_xyz [P][3], log_scaling [P][3] (pre-exp), raw_rot [P][4] (pre-normalise, wxyz)
 *   bary [P][3] or NULL: barycentric origin (AvatarGaussianModel); NULL = face centre.
 * Outputs: xyz [P][3], scaling [P][3], rotation [P][4] (unit, wxyz).
 * Backward adds into dL_dverts [V_m][3] (atomics; caller zeroes) and overwrites
 * dL_dlocal_xyz / dL_dlog_scaling / dL_draw_rot.
 */
int ggs_mesh_bind_forward(int P, int F, const float* verts, const int64_t* faces, const int64_t* binding,
                          const float* local_xyz, const float* log_scaling, const float* raw_rot,
                          const float* bary, float* xyz, float* scaling, float* rotation, void* stream);

int ggs_mesh_bind_backward(int P, int F, const float* verts, const int64_t* faces, const int64_t* binding,
                           const float* local_xyz, const float* log_scaling, const float* raw_rot,
                           const float* bary, const float* dL_dxyz, const float* dL_dscaling,
                           const float* dL_drotation, float* dL_dverts, float* dL_dlocal_xyz,
                           float* dL_dlog_scaling, float* dL_draw_rot, void* stream);

/*
 * Fused photometric loss of the inner steps ("next" #1): masked L1 + 11x11 Gaussian-window
 * SSIM, value and gradient w.r.t. the rendered image, replacing utils/loss_utils.py:17-68 (l1_loss, ssim) as
 * composed at s2_registration.py:259-260 and s3_appearance.py:132-133.
 *   img, gt [V][3][H][W]; mask [V][1][H][W] or NULL (x = img*mask, y = gt*mask like the reference's in-place masking)
 * forward : sums [V][2] (device) <- { sum |x - y| , sum ssim_map(x, y) }  (means = sums / (3 H W)); the per-pixel
 *           derivative maps of the SSIM map stay in `scratch` (ggs_photometric_scratch_bytes() bytes).
 * backward: weights [V][2] (device) = { dLoss/d mean|x-y| , dLoss/d mean ssim_map } per view
 *           (the reference's loss (1-lambda) L1 + 1 - lambda SSIM gives {1-lambda, -lambda} times the upstream grad);
 *           dL_dimg [V][3][H][W] <- d Loss / d img.  Same img / gt / mask / scratch as the forward call.
 */
size_t ggs_photometric_scratch_bytes(int n_views, int H, int W);
int ggs_photometric_forward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                            float* sums, void* scratch, void* stream);
int ggs_photometric_backward(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                             const void* scratch, const float* weights, float* dL_dimg, void* stream);
/* Table form of the two calls above: `gt_tab` / `mask_tab` are DEVICE arrays of n_views device pointers (to one [3][H][W]
 * image / one [H][W] mask each; mask_tab NULL = no mask) which the kernels read at run time.  The loop at
 * s2_registration.py:241-260 fetches another camera's image every iteration (`viewpoint_cam.original_image.cuda()`); with
 * the iteration captured in a hipGraph the table lets a replay name that image by rewriting 8 bytes instead of copying it
 * into a static buffer.  No reference counterpart. */
int ggs_photometric_forward_tab(int n_views, int H, int W, const float* img, const float* const* gt_tab,
                                const float* const* mask_tab, float* sums, void* scratch, void* stream);
int ggs_photometric_backward_tab(int n_views, int H, int W, const float* img, const float* const* gt_tab,
                                 const float* const* mask_tab, const void* scratch, const float* weights, float* dL_dimg,
                                 void* stream);
/* Region-of-interest form.  In the loops at s2_registration.py:252-267 / s3_appearance.py:125-140 the loss gradient exists
 * only to be handed to the rasterizer's backward, and that reads dL/dimage on the pixels of tiles that have a splat list and
 * nowhere else (for one 1080p view of the 100k-Gaussian garment: ~1 tile in 10).  `tile_count` = the list lengths of the
 * forward that rendered `img` (section 1 of ggs_bin_layout: uint32 [n_views][ceil(H/16) * ceil(W/16)], device memory; NULL =
 * the plain form).  With it the backward pass runs only the 64 x 12-pixel boxes that overlap a non-empty tile and leaves
 * dL_dimg UNTOUCHED elsewhere, and the forward pass keeps the derivative maps of the boxes within reach of those; `sums` are the sums
 * over all pixels, as in the plain form.  Images by plain pointers (gt, mask; tables NULL) or by tables (gt_tab, mask_tab;
 * gt NULL), as above.  No reference counterpart (upstream composes the loss from PyTorch ops, utils/loss_utils.py:17-67). */
int ggs_photometric_forward_roi(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                float* sums, void* scratch, void* stream);
int ggs_photometric_backward_roi(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                 const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                 const void* scratch, const float* weights, float* dL_dimg, void* stream);
/* Sparse-mask form of the forward pass.  The masks of the loops are garment silhouettes (`gt_mask` of
 * s2_registration.py:246-258: the garment's label in the segmentation): most of a frame is masked out, and there x = y = 0,
 * every window statistic is zero and the SSIM map is one constant.  `mask_tiles` = the tile occupancy of the masks
 * (ggs_mask_tiles below: uint32 [n_views][ceil(H/16) * ceil(W/16)]; or `mask_tiles_tab`, a DEVICE array of n_views device
 * pointers to one such table each, like mask_tab -- the masks of a capture are static, one table per mask, computed once).
 * A 64-column box without a mask pixel in its input window, whose maps the backward pass will not read (tile_count is
 * REQUIRED), is skipped; the constant part of the SSIM sum is added once per view.  `sums` are those of the plain form up to the
 * fp32 rounding of a different summation order; ggs_photometric_backward_roi follows unchanged.  On a dense mask nothing is
 * skipped and a one- or two-view launch pays ~1.6x for its short bands: use the _roi form there.  No reference counterpart. */
int ggs_photometric_forward_sparse(int n_views, int H, int W, const float* img, const float* gt, const float* mask,
                                   const float* const* gt_tab, const float* const* mask_tab, const uint32_t* tile_count,
                                   const uint32_t* mask_tiles, const uint32_t* const* mask_tiles_tab,
                                   float* sums, void* scratch, void* stream);
/* tiles[v][ty * ceil(W/16) + tx] <- number of non-zero pixels of mask v (mask [n_views][H][W] floats) in the 16x16 tile (tx, ty). */
int ggs_mask_tiles(int n_views, int H, int W, const float* mask, uint32_t* tiles, void* stream);

/*
 * Mean squared distance of every point to its 3 nearest neighbours (self excluded) -- replaces
 * `simple_knn._C.distCUDA2` (scene/gaussian_model.py:135, scene/mesh_gaussian_model.py:233).
 * points [P][3], out [P].  Fewer than 4 points: the missing neighbours count as FLT_MAX like upstream.
 */
int ggs_dist2_3nn(int P, const float* points, float* out, void* stream);
/* The same quantity, bit for bit, through a uniform grid over the bounding box (upstream simple_knn searches Morton-sorted
 * boxes: sub-quadratic, and so is this): scratch = ggs_dist2_3nn_scratch_bytes(P) bytes of device memory.  0.3 ms instead of
 * 4.2 ms at 100k points, and what makes config 5's 500k points affordable.  No host sync. */
size_t ggs_dist2_3nn_scratch_bytes(int P);
int ggs_dist2_3nn_grid(int P, const float* points, float* out, void* scratch, void* stream);

/*
 * StyleGAN2 ops of the appearance network -- what the reference's extension modules `fused` and
 * `upfirdn2d` (scene/styleunet/fused_act.py:30, upfirdn2d.py:30; CUDA sources under scene/styleunet/) export.
 *   ggs_fused_bias_act: y[i] = act(x[i] + bias[(i / step_b) % size_b]) * scale, n elements, bias / ref may be NULL;
 *       act 1 linear, 3 leaky ReLU(alpha); grad 0 forward, 1 derivative gated by sign(ref), 2 zero.
 *   ggs_upfirdn2d: input [major][in_h][in_w][minor], kernel [kh][kw] -> out [major][out_h][out_w][minor]
 *       (upsample by zero insertion, pad (negative = crop), convolve, decimate);
 *       out_h = (in_h up_y + pad_y0 + pad_y1 - kh + down_y) / down_y, likewise out_w (ggs_upfirdn2d_out_size).
 * The `_t` entry points take the element type of every buffer (x, bias, ref, y / input, kernel, out): the reference
 * dispatches float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF: fused_bias_act_kernel.cu:96,
 * upfirdn2d_kernel.cu:340-369); half accumulates in float, double in double, nothing is converted on the way in or out.
 * The un-suffixed entry points are the float forms.
 */
#define GGS_DTYPE_F32 0
#define GGS_DTYPE_F16 1
#define GGS_DTYPE_F64 2
int ggs_fused_bias_act_t(int dtype, size_t n, const void* x, const void* bias, const void* ref, int step_b, int size_b,
                         int act, int grad, float alpha, float scale, void* y, void* stream);
int ggs_upfirdn2d_t(int dtype, int major, int in_h, int in_w, int minor, const void* input, const void* kernel, int kh,
                    int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                    void* out, void* stream);
int ggs_fused_bias_act(size_t n, const float* x, const float* bias, const float* ref, int step_b, int size_b,
                       int act, int grad, float alpha, float scale, float* y, void* stream);
int ggs_upfirdn2d_out_size(int in_h, int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                           int pad_x1, int pad_y0, int pad_y1, int* out_h, int* out_w);
int ggs_upfirdn2d(int major, int in_h, int in_w, int minor, const float* input, const float* kernel, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                  float* out, void* stream);

/*
 * Optimiser update of the inner loops as graph-capturable kernels -- what the reference does with
 * torch.optim.Adam(l, lr=0.0, eps=1e-15) (scene/mesh_gaussian_model.py:375, gaussian_model.py:165, avatar_net.py:50;
 * stepped at s2_registration.py:316-318, s3_appearance.py:143-145).  Same arithmetic as torch's single-tensor Adam
 * (no amsgrad, no weight decay), but step count / bias corrections (`state`, ggs_adam_state_bytes() bytes, zeroed by
 * the caller) and the learning rate (`lr`, one float) live in DEVICE memory, so a whole optimisation step can be
 * replayed as a hipGraph while the host moves the xyz schedule.  `guard` (may be NULL) points at a device u64 --
 * GgsBinHeader.overflow of the forward of the same step: non-zero => tick and step do nothing.
 *   ggs_adam_tick: once per optimiser step, before the per-tensor updates: step += 1, refresh bias corrections.
 *   ggs_adam_step: one tensor of n floats (param / grad / exp_avg / exp_avg_sq 16-byte aligned).
 * betas / eps are doubles because torch forms 1 - beta in double before rounding to fp32 (1 - 0.999f is off by 1.3e-5).
 */
size_t ggs_adam_state_bytes(void);
int ggs_adam_tick(void* state, double beta1, double beta2, const void* guard, void* stream);
int ggs_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lr,
                  double beta1, double beta2, double eps, const void* state, const void* guard, void* stream);
/* The same for up to 16 tensors per launch.  torch keeps a step count per PARAMETER (one without gradient skips the step and
 * its bias corrections lag), so every tensor has its own `state`: states[t], lrs[t] (device float), host arrays of
 * n device pointers / sizes.  Empty tensors are skipped. */
int ggs_adam_tick_multi(int n_states, void* const* states, double beta1, double beta2, const void* guard, void* stream);
int ggs_adam_step_multi(int n_tensors, const size_t* numel, float* const* params, const float* const* grads,
                        float* const* exp_avgs, float* const* exp_avg_sqs, const float* const* lrs,
                        const void* const* states, double beta1, double beta2, double eps, const void* guard,
                        void* stream);
/* ggs_adam_tick_multi + ggs_adam_step_multi in ONE launch (one launch less per optimisation step: ~5 us of a graph-replayed
 * iteration).  Every tensor is listed once; its state is advanced once.  The last 8 bytes of a state are a workgroup ticket
 * that is zero whenever no launch is in flight; a checkpoint keeps the first 32 bytes. */
int ggs_adam_tick_step_multi(int n_tensors, const size_t* numel, float* const* params, const float* const* grads,
                             float* const* exp_avgs, float* const* exp_avg_sqs, const float* const* lrs,
                             void* const* states, double beta1, double beta2, double eps, const void* guard,
                             void* stream);

/*
 * The rest of one s2 registration iteration in two kernels: hinge regularisers of the first-frame
 * template with their gradients (s2_registration.py:262-265), sigmoid chain rule of the opacity
 * (scene/gaussian_model.py:107-108) and the densification statistics (s2_registration.py:301-303,
 * scene/gaussian_model.py:410-412).  vis_i = radii[i] > 0, means over the visible Gaussians.
 *   xyz, log_scaling [P][3]: the LOCAL parameters _xyz / _scaling;  dL_dmeans2D [P][3] from ggs_backward.
 *   dL_dxyz / dL_dlog_scaling [P][3]: gradient buffers the hinge gradients are ADDED to (both NULL: no hinge terms).
 *   dL_dopacity_logit [P] (NULL: skip) = dL_dopacity * opacity (1 - opacity).
 *   max_radii2D, xyz_gradient_accum, denom [P] (max_radii2D NULL: no statistics; left untouched when *guard != 0).
 *   out_losses [3] = {loss_xyz, loss_scale, n_visible} (may be NULL);  scratch: 16 bytes, 16-byte aligned.
 */
int ggs_registration_aux(int P, const float* xyz, const float* log_scaling, const int* radii, const float* dL_dmeans2D,
                         const float* opacity, const float* dL_dopacity, float* dL_dopacity_logit, float threshold_xyz,
                         float lambda_xyz, float threshold_scale, float lambda_scale, float* dL_dxyz,
                         float* dL_dlog_scaling, float* max_radii2D, float* xyz_gradient_accum, float* denom,
                         float* out_losses, void* scratch, const void* guard, void* stream);
/* The same, and thread 0 of its last kernel also advances the optimiser states (ggs_adam_tick_multi) and assembles the step's
 * 48-byte result block -- what a graph-replayed step otherwise does with two small kernels and a copy launch (~5 us each):
 *   out_block: float[8] {sum|x - y|, sum ssim_map, loss_xyz, loss_scale, n_visible, 0, 0, 0} | u64[2] GgsBinHeader.
 * out_block may be host-mapped page-locked memory (ggs_host_mapped_pointer): the host then reads the block after synchronising
 * with the stream, without any copy.  loss_sums / header may be NULL (zeros are stored). */
typedef struct GgsStepTail {
    const float* loss_sums;     /* [2] device: the sums ggs_photometric_forward* produced in this step */
    const void* header;         /* device: GgsBinHeader of this step's forward (the first 16 bytes of `bin`) */
    void* out_block;            /* 48 bytes, 8-byte aligned; NULL: no result block */
    /* ggs_adam_tick_multi's arguments (n_adam_states = 0: none): the optimiser states are advanced here, under `guard`, so
     * that the update that follows is one ggs_adam_step_multi launch and nothing else */
    int n_adam_states;          /* <= 16 */
    void* adam_states[16];
    double beta1, beta2;
} GgsStepTail;
int ggs_registration_aux_tail(int P, const float* xyz, const float* log_scaling, const int* radii, const float* dL_dmeans2D,
                              const float* opacity, const float* dL_dopacity, float* dL_dopacity_logit, float threshold_xyz,
                              float lambda_xyz, float threshold_scale, float lambda_scale, float* dL_dxyz,
                              float* dL_dlog_scaling, float* max_radii2D, float* xyz_gradient_accum, float* denom,
                              float* out_losses, void* scratch, const void* guard, const GgsStepTail* tail, void* stream);

/*
 * Step prologue: the jobs at the head of one optimisation step that depend on nothing computed in it, as ONE launch.  Inside a
 * captured hipGraph a dependent launch costs ~5 us whatever it does; the s2 iteration had eight such launches (four zero fills,
 * the 176-byte parameter copy, sigmoid, a gradient fill, the mesh binding) in front of ~400 us of real work.
 *   clear_ptr / clear_bytes [n_clear <= 8]: ranges to zero-fill (4-byte aligned).  Every range is then MARKED for the calling
 *     host thread and `stream`: the next library call on that thread and stream that would itself zero-fill a range starting
 *     at the same address and no longer -- the binning counters of ggs_forward* (ggs_step_clear_plan: bin_bytes at `bin`), the
 *     gradient records of ggs_backward (backward_scratch_bytes at `scratch`), the sums of ggs_photometric_forward* (8 n_views
 *     bytes), the 16-byte scratch of ggs_registration_aux -- consumes the mark and skips its own fill launch.  Only the ranges
 *     flagged in `consumer_mask` are marked; the others are simply zero-filled (e.g. dL_dverts of ggs_mesh_bind_backward, which
 *     no library call clears).  Marks END WITH THE STEP: ggs_registration_aux* (the step's last library call but the optimiser
 *     update) drops whatever is left after consuming its own, ggs_step_end() does the same for steps shaped differently or cut
 *     short by an error, and so does the next ggs_step_prologue of the thread -- a mark can never meet a zero fill of a later,
 *     unrelated call that happens to start at the same address.  The caller must not write to a marked range
 *     before its consumer ran.
 *   copy_src -> copy_dst, copy_bytes <= 1024 (4-byte aligned; 0: none): a small block copy; copy_src may be host-mapped
 *     page-locked memory, read in place (the per-iteration camera / pointer block without a copy launch).
 *   P, F, verts ... rotation: the arguments of ggs_mesh_bind_forward (P = 0: no binding).
 *   opacity[i] = 1 / (1 + exp(-opacity_logit[i])), i < n_opacity (scene/gaussian_model.py:107-108; 0: none).
 */
#define GGS_PROLOGUE_MAX_CLEAR 8
typedef struct GgsStepPrologue {
    int n_clear;
    void* clear_ptr[GGS_PROLOGUE_MAX_CLEAR];
    size_t clear_bytes[GGS_PROLOGUE_MAX_CLEAR];
    const void* copy_src; void* copy_dst; size_t copy_bytes;
    int P, F;
    const float* verts; const int64_t* faces; const int64_t* binding;
    const float *local_xyz, *log_scaling, *raw_rot, *bary;
    float *xyz, *scaling, *rotation;
    int n_opacity; const float* opacity_logit; float* opacity;
    unsigned consumer_mask;      /* bit i: range i is zero-filled by a later library call of this step -> mark it (see above) */
} GgsStepPrologue;
int ggs_step_prologue(const GgsStepPrologue* d, void* stream);
/* Drops the marks the calling thread still holds (a step that raised half way, a step without ggs_registration_aux*). */
int ggs_step_end(void);
/* Leading bytes of `bin` that ggs_forward* and of `scratch` that ggs_backward zero-fill first. */
int ggs_step_clear_plan(const GgsParams* p, size_t bin_capacity, size_t* bin_bytes, size_t* backward_scratch_bytes);
/* Device address of mapped page-locked host memory (hipHostMalloc: PyTorch's pinned tensors); GGS_ERR_HIP if it is not. */
int ggs_host_mapped_pointer(void* host_ptr, void** device_ptr);

/*
 * Visibility of mesh-bound Gaussians from one camera -- replaces the per-iteration open3d / Embree
 * ray cast of AvatarGaussianModel.get_visible_mask (scene/avatar_gaussian_model.py:227-263): ray i goes from
 * `cam` [3] (device) to targets[i] (the Gaussian's anchor on its face); mask[i] = 1 iff the FIRST triangle the
 * ray hits is binding[i].  first_hit [P] (or NULL) receives that triangle id (-1: no hit).
 * verts [n_verts][3], faces [F][3] int64, binding [P] int64.  scratch: ggs_visibility_scratch_bytes(F, n_verts,
 * ids_capacity); ids_capacity bounds the triangle-in-cell lists of the camera-space grid (32 F is plenty); if it
 * is exceeded the call falls back to testing every triangle on the device -- slower, same answer, no host sync.
 */
size_t ggs_visibility_scratch_bytes(int F, int n_verts, size_t ids_capacity);
int ggs_visibility(int P, int F, int n_verts, const float* verts, const int64_t* faces, const float* cam,
                   const float* targets, const int64_t* binding, void* scratch, size_t ids_capacity,
                   unsigned char* mask, int* first_hit, void* stream);

/* Profiling aid (bench.py roofline leg; not part of the reference's interface).  While enabled on the
 * calling thread, ggs_forward / ggs_backward bracket each kernel with hipEvents on `stream`, synchronise
 * once at the end of the call, and keep the per-kernel milliseconds of that call.  ggs_profile_read copies
 * them out in the order {preprocess, scan_tiles, scatter, sort_tiles, render_fwd, render_bwd,
 * preprocess_bwd, order_tiles} and returns the count (8). */
int ggs_profile_enable(int on);
int ggs_profile_read(float* ms, int n);
/* Timestamp mode of the same aid (round 6): while `device_slots` (device memory, `capacity` x 8 bytes) is set on the calling thread,
 * ggs_forward* / ggs_backward launch a one-lane kernel in front of and behind every kernel they bracket (the eight above + their two
 * zero fills) that stores the device's constant-rate clock (hipDeviceAttributeWallClockRate) into the next slot.  These are ordinary
 * launches on `stream`: a caller that captures its step into a hipGraph captures them too, and every replay refreshes the slots -- the
 * per-kernel intervals then come from the replayed graph itself, launch gaps included, instead of from eager launches with host events
 * between them.  ggs_profile_stamps(NULL, 0) ends the mode; it also restarts the slot numbering.  ggs_profile_stamp_log copies out what
 * each slot written so far is -- 2 k for the start, 2 k + 1 for the end of kernel k in the order of ggs_profile_read, k = 8: a zero
 * fill --, stores the clock rate in kHz, and returns the number of stamps ATTEMPTED since the mode was set (more than the capacity: the
 * tail was not written).  Not part of the reference's interface. */
int ggs_profile_stamps(void* device_slots, int capacity);
int ggs_profile_stamp_log(int* ids, int capacity, int* clock_khz);

/* Introspection (bench.py's compute-side roofline; not part of the reference's interface): *count (device, 8 bytes) <-
 * the number of (Gaussian, pixel) pairs the forward that filled geom / bin / img blended = the pairs its backward
 * differentiates.  Walks the forward's lists with the backward's own tests; a few hundred microseconds, never on a timed path. */
int ggs_count_blends(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity, const void* img,
                     unsigned long long* count, void* stream);
/* Evaluated against blended work of the two compositing kernels (bench.py reports the ratio per kernel; VERDICT r5 #1c).
 * ggs_count_pairs: counts4 (device, 32 bytes) <- {blended pairs as ggs_count_blends, quadrant passes of the backward (64 pixels
 * evaluated each), list entries it reduces, list entries it walks}, from a COMPLETED forward.
 * ggs_count_forward_visits: counts3 (device, 24 bytes) <- {quadrant passes of the forward (64 alpha tests each), passes in which
 * some pixel passed the test (those run the blend), list entries walked}; it repeats the forward's walk on the lists as the binning
 * left them, so it must run BETWEEN GGS_STAGE_BIN and GGS_STAGE_COMPOSITE of a ggs_forward_stages sequence (the compositing narrows
 * the quadrant masks in place).  Neither is on a timed path. */
int ggs_count_pairs(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity, const void* img,
                    unsigned long long* counts4, void* stream);
int ggs_count_forward_visits(const GgsParams* p, const void* geom, const void* bin, size_t bin_capacity,
                             unsigned long long* counts3, void* stream);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* ggs_last_error(void);

/* Library version / build target string, e.g. "ggsplat 0.1 gfx950". */
const char* ggs_version(void);
/* First 16 hex digits of the sha256 over the library's sources (csrc Makefile order): identifies the build a profile or a
 * counter collection belongs to.  No reference counterpart (the upstream extension carries no build id). */
const char* ggs_build_id(void);
/* Pixel size of the tiles this build bins into (16 x 16 in the product; a 32 x 16 A/B build exists): the tile grid of
 * tile_count / ggs_bin_layout is ceil(W / width) x ceil(H / height).  Upstream's BLOCK_X / BLOCK_Y (compile-time constants
 * of the CUDA extension, no call). */
int ggs_tile_size(int* width, int* height);

#ifdef __cplusplus
}
#endif
#endif /* GGSPLAT_H */
