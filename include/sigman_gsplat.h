/*
 * sigman_gsplat.h -- C ABI of libsigman_gsplat.so: the MI355X (gfx950) differentiable Gaussian-
 * splatting rasterizer that replaces the third-party CUDA extension bound by the reference at
 *
 *     /root/reference/core/gaussians/gs.py:8-11    (import of GaussianRasterizationSettings / GaussianRasterizer)
 *     /root/reference/core/gaussians/gs.py:82-106  (settings construction + rasterizer(...) call, once per view)
 *     /root/reference/train_vae.py:166             (autograd backward through that call)
 *
 * The upstream package exposes three native entry points through pybind (`_C.rasterize_gaussians`,
 * `_C.rasterize_gaussians_backward`, `_C.mark_visible`).  This header declares
 *   (1) their one-call equivalents with allocator callbacks (sgr_rasterize_forward / _backward /
 *       sgr_mark_visible) -- what a maintainer binds to keep gs.py unchanged, and
 *   (2) the staged, view-BATCHED entry points the fast path uses (one launch chain for all B*V views,
 *       replacing the Python double loop at gs.py:62,75).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; fp32, contiguous, row-major
 *   - `stream` is a hipStream_t passed as void* (0 = legacy default stream)
 *   - all functions return 0 on success, non-zero on failure; sgr_last_error() gives the message.
 *     Nothing throws or aborts across this boundary (the reference caller swallows Python
 *     exceptions at core/modules/autoencoder.py:349-361, so errors must surface as return codes).
 *   - all buffers are owned by the caller (PyTorch); the library keeps no pointer after a call returns
 *   - matrices use the memory order of the reference's tensors: flat[4*c + r] = M[r][c]
 *     (cam_view = w2c^T, cam_view_proj = cam_view @ P^T; core/dataset/dataloader_VAE.py:207-208)
 *
 * Batched layout: `n_views` view slots; slot v renders subject s = v / views_per_subject; each subject
 * has P Gaussians.  Per-(view,Gaussian) arrays are indexed q = v*P + i.
 */
#ifndef SIGMAN_GSPLAT_H
#define SIGMAN_GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_ABI_VERSION 9
#define SGR_TILE 16                 /* 16x16 pixel tiles, as the published algorithm */
#define SGR_PART_FLOATS 10          /* floats of a partial gradient record (bucket-parallel backward): 40 B, 8-byte aligned */
#define SGR_REC_STRIDE 16           /* floats of the packed per-(view,Gaussian) record `rec` (64 B, one cache line) */

/* Problem description shared by every staged call. */
typedef struct SgrProblem {
    int32_t P;                      /* Gaussians per subject */
    int32_t n_views;                /* total view slots (B*V) */
    int32_t views_per_subject;      /* V; subject of slot v is v / V */
    int32_t H, W;                   /* image size */
    int32_t sh_degree;              /* 0..3 (only read when shs != NULL) */
    int32_t M;                      /* SH coefficients per Gaussian in `shs` */
    float tanfovx, tanfovy;
    float scale_modifier;           /* applied ONLY on the scales/rotations path (as upstream) */
    /* per-subject inputs [S,P,...] */
    const float *means3D;           /* [S,P,3] */
    const float *opacities;         /* [S,P]   */
    const float *colors_precomp;    /* [S,P,3] or NULL */
    const float *shs;               /* [S,P,M,3] or NULL (exactly one of colors_precomp / shs) */
    const float *cov3D_precomp;     /* [S,P,6] xx,xy,xz,yy,yz,zz or NULL */
    const float *scales;            /* [S,P,3] or NULL (exactly one of cov3D_precomp / scales+rotations) */
    const float *rotations;         /* [S,P,4] (r,x,y,z), used un-normalised as upstream */
    /* per-view cameras */
    const float *viewmatrix;        /* [n_views,16] */
    const float *projmatrix;        /* [n_views,16] */
    const float *campos;            /* [n_views,3]  */
    const float *bg;                /* [3] */
    /* optional (ABI v6): the clamp of the reference's caller (rendered_image.clamp(0, 1), gs.py:107) folded into the compositing kernels */
    float *color_clamped;           /* [n_views,3,H,W] or NULL.  Forward: clamp(colour, 0, 1) is ALSO written here (out_color stays unclamped:
                                       the backward starts from it) */
    int32_t clamp_grad;             /* backward: != 0 -> grad_color is dL/d(clamped colour): it passes where 0 <= colour <= 1 (torch.clamp's
                                       inclusive mask) and is dropped elsewhere; needs the bucket backward (with_aux) */
    int32_t reserved0;
} SgrProblem;

/*
 * Packed per-(view,Gaussian) record written by sgr_preprocess_forward and gathered by the render
 * kernels (4 x float4 = 64 B = one cache line per tile instance):
 *   rec[0..3]   = pixel x, pixel y, conic.xx, conic.xy
 *   rec[4..7]   = conic.yy, opacity, view depth, r
 *   rec[8..11]  = g, b, bf16(hx) | bf16(hy) << 16, p*
 *                 hx, hy: half extents of the exact alpha >= 1/255 bound, as the two bf16 halves of ONE word, rounded up (0xBF80 = -1: never
 *                 visible); p* (ABI v7): the published alpha test `min(0.99, opacity * exp(power)) >= 1/255` as a threshold on the exponent in the
 *                 exp2 domain -- the smallest float power2 <= 0 that passes, computed with a correctly rounded exp2 exactly like the CPU oracle
 *                 (+inf: never passes); the compositing kernels test p* <= power2 <= 0 (csrc/render.hip, header)
 *   rec[12..15] = 0 (padding: the record is one 64-byte line)
 */

int sgr_abi_version(void);
const char *sgr_last_error(void);

/* ---- one-call API (upstream-shaped: allocator callbacks, all views of a batch) --------------- */

/*
 * Allocator callback, the C-ABI form of upstream's `std::function<char*(size_t)>` resize functors: must return a DEVICE
 * pointer to at least `bytes` bytes (256-byte aligned) that stays valid until the matching backward has run.
 * which: 0 = geometry blob, 1 = binning blob, 2 = image blob, 3 = backward scratch.
 */
typedef char *(*sgr_alloc_fn)(void *user, int32_t which, size_t bytes);

/* Filled by sgr_rasterize_forward; the caller keeps it (and the blobs) alive for sgr_rasterize_backward. */
typedef struct SgrForwardState {
    uint64_t R_alloc;            /* size of the binning buffers in tile instances: exact num_rendered, or the capacity */
    uint64_t true_rendered;      /* exact mode: num_rendered; sync-free mode: ~0 (read nr_pinned_host after nr_event) */
    uint64_t NS;                 /* bucket slots per quadrant */
    int32_t with_aux /* 0 none, 2 row checkpoints (1 was the compact layout of ABI <= 6) */, result_in_b, flags_cleared;
    int32_t aux_no_da;           /* 1: the forward left the (depth, alpha) checkpoints out (with_aux & 2); sgr_rasterize_backward adds them on demand */
    int32_t fwd_kind;            /* the compositing kernel the forward used (2 segment-parallel, 3 one wave per quadrant) */
    int32_t nr_by_copy;          /* sync-free mode: 1 = the count reaches nr_pinned_host through an async device-to-host COPY (not byte-atomic: wait for
                                    nr_event before reading it); 0 = through one 8-byte store of a kernel (the word may be polled) */
    void *geom, *binning, *image;
    uint64_t geom_bytes, binning_bytes, image_bytes;
    uint64_t off_rec, off_rect, off_clamped, off_block_offsets, off_num_rendered;               /* in geom   */
    uint64_t off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_sort_ws;                        /* in binning */
    uint64_t off_ranges, off_final_T, off_n_contrib, off_compact, off_ckpt_tc, off_ckpt_da, off_desc, off_order, off_flags;   /* in image */
    /* ABI v8: the single-view fused step (sgr_rasterize_forward_l1 with SgrL1Epilogue.fuse_backward) */
    uint64_t off_part, off_loss_part;   /* in image: the backward's partial records [4*R_alloc*10] f32, the per-(tile, quadrant) loss shares */
    int32_t fused_bwd;           /* 1: the forward call also produced the loss, dL/dcolor and the partial records of the loss's own backward:
                                    sgr_rasterize_backward may be called with grad_color = NULL */
    int32_t order_kind;          /* ABI v9: form of the segment-parallel forward's work order at off_order: 0 = one uint4 per slot (render.hip's own
                                    prepare step), 1 = class-major, written by the single-view path's per-tile sort (empty tiles not listed) */
    uint64_t off_flags_fused;    /* ABI v9, in image: the record flags of the fused step's OWN backward (off_flags: of an ordinary backward's scratch records) */
} SgrForwardState;

/*
 * == upstream _C.rasterize_gaussians (gs.py:98-106), for all n_views view slots at once.
 * capacity = 0: exact mode, ONE blocking read of num_rendered per call (upstream: one per view).
 * capacity > 0: sync-free mode; binning buffers sized for `capacity` instances; the true count reaches nr_pinned_host[0]
 *               asynchronously as ONE 8-byte word, count | overflow << 63 (the host can never see the count without its flag), and
 *               state->nr_by_copy tells how the word travels: 0 = a kernel's single 8-byte store (poll the word; nr_event is NOT
 *               recorded), 1 = an async copy (a copy engine may write it piecewise: `nr_event`, a hipEvent_t, may be NULL, is recorded
 *               behind the launch chain: wait for it before reading).
 *               (exact mode fills nr_pinned_host[0] = count, [1] = overflow flag before it returns.)
 * with_aux != 0 also records what the bucket-parallel backward needs; with_aux = 3 (bit 1) leaves the per-pixel (depth, alpha) checkpoints --
 *               a third of the checkpoint stream, read only by a backward that is handed dL/ddepth or dL/dalpha -- to that backward, which
 *               then produces them with a second compositing pass (never needed on the reference's call paths).
 * alloc may be NULL if state->geom / binning / image and their *_bytes capacities are pre-filled by the caller (sizes as reported in
 * the state of an earlier call with the same shapes): no callbacks; returns 2 (nothing useful launched) if a blob is too small.
 * caller_clear / caller_clear_bytes: optional device buffer (multiple of 4 bytes) that the call zeroes on the side of its own
 * kernels (no extra launch): the accumulators of a loss kernel the caller runs right behind the forward.
 * Outputs: out_color [n_views,3,H,W], out_depth/out_alpha [n_views,1,H,W], out_radii i32 [n_views,P].
 */
int sgr_rasterize_forward(const SgrProblem *pb, uint64_t capacity, int32_t with_aux, sgr_alloc_fn alloc, void *user,
                          float *out_color, float *out_depth, float *out_alpha, int32_t *out_radii, uint64_t *nr_pinned_host,
                          void *nr_event, void *caller_clear, uint64_t caller_clear_bytes, SgrForwardState *state, void *stream);

/*
 * sgr_rasterize_forward followed by sgr_clamped_l1_loss on the rendered colour, in ONE call (the fused rasterize + image-loss node,
 * gs.py:98-107 + whole_loss.py:126-131): the loss kernel is queued right behind the compositing kernel instead of after a trip back
 * through the caller (at one 512^2 view the host, not the GPU, paces the step).  Return codes as sgr_rasterize_forward; on 2 nothing of
 * the epilogue ran.
 *
 * fuse_backward != 0 (ABI v8) allows the FUSED single-view step: when the launch composites with the segment-parallel kernel (<= 2048
 * tiles: one or two 512^2 views) and records auxiliary outputs (with_aux = 3: no depth/alpha checkpoints), the clamp + masked L1 and its
 * gradient are evaluated by the compositing workgroups themselves (the loss is pixel-local; the background's share is pre-filled by an
 * early launch) -- no loss launch -- and the bucket backward of dL/dloss = 1 is queued right behind by the same call, its spare workgroup
 * adding up the loss shares: the forward call leaves the partial records behind (state->fused_bwd = 1, state->off_part).
 * sgr_rasterize_backward with grad_color = NULL then only gathers them, multiplied by *grad_color_scale.  grad_color (dL/dcolor) is still
 * written, so a backward that is handed another upstream gradient (grad_color != NULL) works as after any forward.  Results: dL/dcolor and
 * the partial records are bit-identical to the unfused path's; the loss sums are added in a fixed order (no atomics: reproducible) and so
 * differ from the unfused path's in the last bits only.
 */
typedef struct SgrL1Epilogue {
    const float *target;          /* [n_views,3,H,W] */
    const float *mask;            /* [n_views,1,H,W] or NULL */
    float *grad_color;            /* [n_views,3,H,W]  d loss / d color */
    float *loss_per_view;         /* [n_views] */
    float *loss_total;            /* [1] or NULL */
    float weight;
    int32_t sums_already_zero;    /* 1: the accumulators are cleared by this very call (caller_clear) or were cleared by the caller */
    int32_t fuse_backward;        /* see above; 0 = never */
    int32_t reserved0;
} SgrL1Epilogue;
int sgr_rasterize_forward_l1(const SgrProblem *pb, uint64_t capacity, int32_t with_aux, sgr_alloc_fn alloc, void *user,
                             float *out_color, float *out_depth, float *out_alpha, int32_t *out_radii, uint64_t *nr_pinned_host,
                             void *nr_event, void *caller_clear, uint64_t caller_clear_bytes, SgrForwardState *state,
                             const SgrL1Epilogue *l1, void *stream);

/*
 * == upstream _C.rasterize_gaussians_backward (reached from train_vae.py:166).  Gradient outputs as in
 * sgr_preprocess_backward.  Needs a forward that ran with with_aux != 0 -- or one that rendered nothing (no visible tile instance:
 * every gradient is then written as zero).  out_color/out_depth/out_alpha are the forward's outputs.  grad_color_scale: optional DEVICE scalar
 * multiplied onto grad_color (the upstream gradient of a fused image loss; saves the caller an elementwise kernel), or NULL.
 * grad_color = NULL (ABI v8): only after a forward with state->fused_bwd = 1 -- the gradient of that forward's own loss, times
 * *grad_color_scale (grad_depth / grad_alpha must be NULL).
 */
int sgr_rasterize_backward(const SgrProblem *pb, const SgrForwardState *state, const int32_t *radii, const float *out_color,
                           const float *out_depth, const float *out_alpha, const float *grad_color, const float *grad_depth,
                           const float *grad_alpha, const float *grad_color_scale, sgr_alloc_fn alloc, void *user, float *dL_dmeans3D, float *dL_dmeans2D,
                           float *dL_dopacity, float *dL_dcolors, float *dL_dsh, float *dL_dcov3D, float *dL_dscales,
                           float *dL_drotations, void *stream);

/* upstream's `debug=True` (error conventions): while enabled, every kernel launch of the calling thread is followed by a
 * device synchronise, and a fault is reported as a failure of the entry point with the name of the kernel in sgr_last_error();
 * Thread-local; returns the previous value. */
int sgr_set_debug(int enable);

/* ---- staged, batched API -------------------------------------------------------------------- */

/* number of preprocess thread blocks per view (block_offsets needs n_views*that + 1 entries) */
int32_t sgr_preprocess_blocks_per_view(int32_t P);

/*
 * F1 + F2: cull/project/cov2D/conic/radius/rect per (view,Gaussian), block-wise tile counts and their
 * exclusive scan.  Outputs: rec [n_views*P*16], radii i32 [n_views*P], rect u32 [n_views*P*4]
 * (minx | miny<<16, maxx | maxy<<16, depth key bits, first tile-instance index -- written by sgr_bin: one 16-byte record for the emission
 * kernel and the backward's gathers), clamped u8 [n_views*P] (SH clamp bits, may be NULL without shs),
 * block_offsets u32 [2*(n_views*blocks_per_view + 1)] (first half: exclusive offsets, entry n = R; second half:
 * scratch for the un-scanned sums), num_rendered u64 [4] ([0] = R, [1] = 1 if R overflows the 32-bit instance index or
 * `capacity`, [2] = R | overflow << 63: the word the sync-free mode publishes to the host, [3] unused).  capacity = 0: none (the caller reads R back and sizes the binning buffers exactly, like upstream);
 * capacity > 0: the caller pre-sized its binning buffers for `capacity` instances and never reads R on the critical path.
 */
int sgr_preprocess_forward(const SgrProblem *pb, float *rec, int32_t *radii, uint32_t *rect, uint8_t *clamped,
                           uint32_t *block_offsets, uint64_t *num_rendered, uint64_t capacity, void *stream);

/* sort flavour: 3 = automatic (default); 5 = the single-view path (ABI 9): the emission kernel writes tile-ordered runs of (depth bits, value) composites
 * and a run matrix, ONE launch then gathers and sorts every tile's list in LDS (no tile pass): one or two 512^2 views (<= 2048 tiles, <= 2^19
 * instances), else like 3; 4 = view-segmented: the emission is view-major, so ONE order-free counting pass per view over the tile id (<= 4096 tiles per
 * view) + a register sort of the composites per tile; 1 = three kernels per 8-bit digit over the whole key (the fallback beyond 4096 tiles
 * per view).  All give bit-identical sorted keys, values and ranges (for finite depths).  Anything else is refused.  (ABI <= 6 also had
 * 0 = onesweep and 2 = LDS-segmented.) */
int sgr_set_sort_mode(int mode);
/* deep tile lists in the view-segmented flavour: instead of the register comparison network, a long tile is sorted by DISTRIBUTION in LDS
 * by one workgroup (per window of 15 232 entries): adaptive depth bins from the tile's own histogram, bin-ordered placement, rank inside
 * the (tiny) bin -- O(n).  mode 0 = automatic (launches with more than 1024 instances per tile on average, e.g. 1M Gaussians at 512^2:
 * they used to fall back to six whole-key radix passes), 1 = whenever flavour 4 runs, 2 = never (flavour 4 only: flavour 5 always uses it).  Same bits out; a tile with massive
 * exact depth ties (> 128 in one bin) takes the generic path.  Bits 8..15 of `mode` (tests; 0 = default 64): on the single-view path
 * (flavour 5: one or two 512^2 views) a tile with more windows of 3968 entries than this is sorted whole by its workgroup's stable radix passes --
 * what a tile beyond 64 windows gets in production.  Bits 16..19 (tests): 1 / 2 = deep launches of one or two views with the five-launch tile pass /
 * the one collect launch (default), 0 = the environment's SIGMAN_SORT_COLLECT. */
int sgr_set_sort_deep(int mode);

/* bytes of scratch sgr_bin needs for R tile instances */
size_t sgr_bin_workspace_bytes(uint64_t R, uint64_t tiles_total /* n_views * tiles per view */);

/*
 * F3 + F4 + F5: emit (key,value) per touched tile, stable LSD radix sort on the significant key bits,
 * per-tile ranges.  key = ((view*tiles + tile) << 32) | float_bits(depth); value = v*P + i.
 * R = exact instance count (num_rendered_dev = NULL), or the buffer CAPACITY with num_rendered_dev pointing at the device
 * counter written by sgr_preprocess_forward (sync-free mode: grids are sized by the capacity, kernels read the true count).
 * keys/vals: two buffers of R entries each (ping-pong).  On return *result_in_b_host tells which
 * buffer holds the sorted list.  ranges u32 [n_views*tiles*2] (start,end) into the sorted list.
 */
int sgr_bin(const SgrProblem *pb, const int32_t *radii, uint32_t *rect /* [3] of every record is written */,
            const uint32_t *block_offsets, uint64_t R, const uint64_t *num_rendered_dev, uint64_t *keys_a, uint64_t *keys_b,
            uint32_t *vals_a, uint32_t *vals_b, void *workspace, size_t workspace_bytes, uint32_t *ranges,
            int32_t *result_in_b_host, void *stream);

/* number of bucket slots per quadrant for the auxiliary forward outputs: (R >> 6) + tiles_total + 1 */
uint64_t sgr_bucket_slots(uint64_t R, uint64_t tiles_total);

/* forward compositing kernel choice: 0 = automatic (segment-parallel for launches of <= 2048 tiles, one wave per (tile, quadrant)
 * otherwise), 2 = segment-parallel, 3 = one wave per (tile, quadrant); anything else is refused.  Both produce the same integer
 * artefacts and the same alpha-test decisions. */
int sgr_set_forward_mode(int mode);

/* B2 + B3 kernel choice: 0 = automatic (default: on the colors_precomp path with views_per_subject in {2, 4, .., 256} one thread per
 * (view, Gaussian), the per-view contributions added in view order by one thread per Gaussian; otherwise one thread per Gaussian that
 * loops over the views), 1 = always the loop.  (Tests: 2 / 3 = the loop with 64 / 16 Gaussians per wave whatever the launch holds; launches of at
 * most 32 768 Gaussians take the latter on their own.)  All give bit-identical gradients. */
int sgr_set_backward_gather(int mode);

/* the fused single-view step of sgr_rasterize_forward_l1 when the epilogue allows it and the launch qualifies (environment SIGMAN_FUSED_STEP):
 * 1 (default) = taken; 0 = never (A/B measurements, parity tests against the unfused chain).  Thread-local; returns the previous value. */
int sgr_set_fused_step(int on);

/* sgr_rasterize_forward*: 0 (default) = when the binning ends in the register per-tile sort, only the point list is stored -- the sorted keys
 * have no reader behind that sort (the tile ranges come from the tile pass); 1 = keep the sorted keys in the binning blob as well
 * (debug / parity tests).  Returns the previous setting.  sgr_bin itself always returns sorted keys. */
int sgr_set_keep_sorted_keys(int keep);

/* F1: views one workgroup of sgr_preprocess_forward walks with its Gaussians held in registers: 0 = automatic (default: up to 8, as many
 * as leave >= 4096 workgroups), n >= 1 = exactly n (dev/test override).  The outputs do not depend on it. */
int sgr_set_preprocess_view_group(int n);

/*
 * F6: per-tile front-to-back compositing.  out_color [n_views,3,H,W], out_depth [n_views,1,H,W],
 * out_alpha [n_views,1,H,W], final_T f32 [n_views,H,W], n_contrib u32 [n_views,H,W].
 * Optional auxiliary outputs for the bucket-parallel backward (pass all four or none; NS = sgr_bucket_slots(R, n_views*tiles)):
 *   aux_compact  u32 [4][R][2]   per (tile, 8x8 quadrant) culled list: (record id, index in the tile list)
 *   aux_ckpt_tc  f32 [4*NS][4][64][4], aux_ckpt_da f32 [4*NS][4][64][2]   per-pixel (T,C) / (D,A) checkpoints of each <=64-survivor bucket:
 *                one record before each 16-survivor row -- T absolute; sums absolute on rows that start a forward segment, else relative
 *                to that row
 *   aux_desc     u32 [4*NS][2]   bucket descriptors (tile | (rows per segment - 1) << 30, (start << 7) | count); zeroed by this call
 * aux_order (optional, u32 [4 * n_views*tiles]) receives the work order of the segment-parallel kernel: (tile, first, end of its list, 0) per
 * slot, longest tile lists first, empty tiles last; NULL = tiles in index order.
 */
int sgr_render_forward(const SgrProblem *pb, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                       float *out_color, float *out_depth, float *out_alpha, float *final_T, uint32_t *n_contrib,
                       uint64_t R, void *aux_compact, void *aux_ckpt_tc, void *aux_ckpt_da, void *aux_desc, uint32_t *aux_order,
                       void *stream);

/*
 * B1: gradients of the compositing from the image gradients.  grad_depth / grad_alpha may be NULL (treated as zero; with either given,
 * aux_ckpt_da must hold the forward's depth/alpha checkpoints).  Needs the forward's auxiliary outputs and output images: one wave per
 * <=64-Gaussian bucket, lanes own Gaussians, pixel states move through the lanes (no reductions, no LDS atomics, NO GLOBAL ATOMICS):
 * each lane writes one partial record part[(4*instance + quadrant)*10 .. +10] and sets flags byte [4*instance + quadrant]
 * (part f32 [4*R*SGR_PART_FLOATS] = [4*R*10], flags u32 [R], flags zeroed by this call); sgr_preprocess_backward gathers them in a fixed order,
 * so gradients are bitwise reproducible.  (ABI <= 6 also had the published pixel-parallel reverse walk with float atomics as a second
 * path -- `grec`, `final_T`, `point_list` arguments; removed in v7.)
 */
int sgr_render_backward(const SgrProblem *pb, const uint32_t *ranges, const float *rec, const uint32_t *rect,
                        const uint32_t *n_contrib, const float *out_color, const float *out_depth,
                        const float *out_alpha, const float *grad_color, const float *grad_depth, const float *grad_alpha,
                        const float *grad_color_scale /* optional device scalar on grad_color, or NULL */,
                        uint64_t R, const void *aux_compact, const void *aux_ckpt_tc, const void *aux_ckpt_da,
                        const void *aux_desc, float *part, uint32_t *flags, void *stream);

/*
 * B2 + B3: the bucket backward's partial records (`rect` + `part` + `flags`) -> per-subject parameter gradients, summed over the
 * subject's views in a fixed order (no atomics).  Outputs are fully written (no pre-zeroing needed):
 *   dL_dmeans3D [S,P,3], dL_dmeans2D [n_views,P,3] (NDC units like upstream, z = 0; may be NULL: not written),
 *   dL_dopacity [S,P], dL_dcolors [S,P,3] (or dL_dsh [S,P,M,3] when shs), dL_dcov3D [S,P,6],
 *   dL_dscales [S,P,3] / dL_drotations [S,P,4] (only when scales given; else may be NULL)
 */
int sgr_preprocess_backward(const SgrProblem *pb, const int32_t *radii, const uint8_t *clamped,
                            const uint32_t *rect, const float *part, const uint32_t *flags, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors,
                            float *dL_dsh, float *dL_dcov3D, float *dL_dscales, float *dL_drotations,
                            void *stream);

/* ---- callers on either side of the rasterizer --------------------------------- */

/*
 * simple_knn.distCUDA2 replacement (gs.py:70): out_dist2[i] = mean of the squared distances from point i to its 3
 * nearest OTHER points (exact).  points [P,3].  workspace: sgr_knn_workspace_bytes(P, max_cells) bytes;
 * max_cells bounds the uniform grid (e.g. 1<<21).
 */
size_t sgr_knn_workspace_bytes(int32_t P, int32_t max_cells);
int sgr_knn_dist2(int32_t P, const float *points, float *out_dist2, void *workspace, size_t workspace_bytes,
                  int32_t max_cells, void *stream);
/* the same for n_sets point sets [n_sets,P,3] -> [n_sets,P] in one launch sequence (the B subjects of a step, gs.py:62);
 * workspace: n_sets * round_up(sgr_knn_workspace_bytes(P, max_cells), 256) bytes */
int sgr_knn_dist2_batched(int32_t n_sets, int32_t P, const float *points, float *out_dist2, void *workspace,
                          size_t workspace_bytes, int32_t max_cells, void *stream);

/*
 * Fused covariance build (gs.py:71-73 + get_covariance/strip_lowerdiag, gs.py:17-38), n = total Gaussians:
 *   scale = (scale_raw + 1) * sqrt(max(dist2, 1e-7));  Sigma = R diag(scale^2) R^T;  cov6 = xx,xy,xz,yy,yz,zz
 * scale_raw [n,3], rotation [n,3,3] (any 3x3 matrix, as the reference passes R_def), dist2 [n] (no gradient).
 */
int sgr_cov3d_forward(int32_t n, const float *scale_raw, const float *rotation, const float *dist2, float *cov6, void *stream);
int sgr_cov3d_backward(int32_t n, const float *scale_raw, const float *rotation, const float *dist2, const float *grad_cov6,
                       float *grad_scale_raw, float *grad_rotation, void *stream);

/*
 * Fused image-space loss epilogue (gs.py:107 clamp + whole_loss.py:126-131 masked L1), one pass:
 *   loss_per_view[v] = weight * sum_{c,p} mask * |clamp(color,0,1) - target|      (zeroed by the call)
 *   grad_color       = weight * mask * sign(clamp(color) - target) * 1[0 <= color <= 1]   (inclusive, like torch.clamp's backward)
 *   loss_total       = sum_v loss_per_view[v]   (optional, may be NULL; zeroed by the call; saves the caller a reduction launch)
 * sums_already_zero != 0: the caller guarantees both accumulators are zero (e.g. cleared by sgr_rasterize_forward's caller_clear)
 * color/target/grad_color [n_views,3,H,W]; mask [n_views,1,H,W] or NULL.
 */
int sgr_clamped_l1_loss(int32_t n_views, int32_t H, int32_t W, const float *color, const float *target, const float *mask,
                        float weight, float *grad_color, float *loss_per_view, float *loss_total, int32_t sums_already_zero,
                        void *stream);

/* ---- optional per-kernel profiler (HIP events on the launch stream; used by bench.py) -------- */
enum {
    SGR_K_PREPROCESS_FWD = 0, SGR_K_SCAN = 1, SGR_K_DUPLICATE = 2, SGR_K_SORT = 3, SGR_K_RANGES = 4,
    SGR_K_RENDER_FWD = 5, SGR_K_RENDER_BWD = 6, SGR_K_PREPROCESS_BWD = 7, SGR_K_KNN = 8, SGR_K_COV3D = 9,
    SGR_K_LOSS = 10, SGR_K_COUNT = 16
};
/* bit k of kernel_mask enables event pairs around kernel id k; 0 disables (default) */
int sgr_prof_configure(uint32_t kernel_mask);
/* after a device/stream synchronise: per-kernel-id summed milliseconds and launch counts; clears the log */
int sgr_prof_collect(double *total_ms /*[SGR_K_COUNT]*/, uint32_t *counts /*[SGR_K_COUNT]*/);

/* the shader clock (MHz) one wave ran at during a ~0.5-ms probe kernel on `stream`: ratio of the chip's own cycle counter to its constant 100-MHz
 * counter (bench.py reports it around the timed region: the box pool has a slow level the amdgpu sysfs node does not show).  Synchronises the stream. */
int sgr_clock_probe(double *mhz_host, void *stream);

/* upstream `mark_visible`: present[i] = (view-space z > 0.2) */
int sgr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *present, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGMAN_GSPLAT_H */
