// rasterize.hip -- the one-call entry points: what upstream's pybind `_C.rasterize_gaussians` /
// `_C.rasterize_gaussians_backward` are for /root/reference/core/gaussians/gs.py:98-106 and train_vae.py:166.
// Like upstream's CudaRasterizer::Rasterizer::forward(geomFunc, binningFunc, imageFunc, ...) the caller passes
// ALLOCATOR CALLBACKS (PyTorch owns every byte: the callback resizes a uint8 tensor and returns its data_ptr), the
// library lays its buffers out inside the three blobs and launches the whole chain
//     preprocess -> scan -> [num_rendered] -> duplicate -> radix sort -> ranges -> composite
// from ONE host call, for all views of the batch.  Host-side cost matters here: at the reference's sizes the
// GPU needs ~0.35 ms for a 512^2 view fwd+bwd, which a Python-driven launch sequence cannot feed.
#include <string.h>

#include <cstdlib>
#include "common.h"

namespace {
inline uint64_t align_up(uint64_t v, uint64_t a = 256) { return (v + a - 1) / a * a; }
}

// 0 (default): a fused forward whose binning ends in the register per-tile sort stores only the point list -- the sorted keys have no
// reader behind that sort (the tile ranges come from the tile pass); 1: keep them (forward_debug / the parity tests look at them)
// the single-view fused step (SgrL1Epilogue.fuse_backward) when the launch qualifies: 1 (default) = taken, 0 = never (A/B, tests): SIGMAN_FUSED_STEP
static thread_local int g_fused_step = sgr_env_knob("SIGMAN_FUSED_STEP", 0, 1, 1);
extern "C" int sgr_set_fused_step(int on) { const int old = g_fused_step; g_fused_step = on ? 1 : 0; return old; }
static int sgr_fused_step_enabled() { return g_fused_step; }
static thread_local int g_keep_sorted_keys = 0;     // thread-local like sgr_set_debug: forward_debug() toggles it around ONE call
extern "C" int sgr_set_keep_sorted_keys(int keep) { const int old = g_keep_sorted_keys; g_keep_sorted_keys = keep ? 1 : 0; return old; }
int sgr_preprocess_forward_ex(const SgrProblem *pb, float *rec, int32_t *radii, uint32_t *rect, uint8_t *clamped, uint32_t *block_offsets,
                              uint64_t *num_rendered, uint64_t capacity, bool skip_scan, const SgrBgJob *bg, void *stream_);
int sgr_bin_ex(const SgrProblem *pb, const int32_t *radii, uint32_t *rect, const uint32_t *block_offsets, uint64_t R,
               const uint64_t *num_rendered_dev, uint64_t *keys_a, uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *workspace,
               size_t workspace_bytes, uint32_t *ranges, int32_t *result_in_b_host, bool self_scan, uint64_t *nr_host, uint32_t *fwd_order,
               int *order_kind_out, const SgrBgJob *bg, bool occ_zeroed, uint32_t *const *clear_ptr, const uint64_t *clear_words,
               int *clear_done, bool first_index, bool sorted_keys, void *stream_);
int sgr_render_backward_ex(const SgrProblem *pb, const uint32_t *ranges, const float *rec, const uint32_t *rect,
                           const uint32_t *n_contrib, const float *out_color, const float *out_depth, const float *out_alpha,
                           const float *grad_color, const float *grad_depth, const float *grad_alpha, const float *grad_color_scale,
                           uint64_t R, const void *aux_compact, const void *aux_ckpt_tc, const void *aux_ckpt_da, const void *aux_desc,
                           float *part, uint32_t *flags, bool flags_cleared, const SgrFusedL1Args *loss_reduce, void *stream_);
int sgr_preprocess_backward_ex(const SgrProblem *pb, const int32_t *radii, const uint8_t *clamped, const uint32_t *rect,
                               const float *part, const uint32_t *flags, uint64_t n_inst, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity,
                               float *dL_dcolors, float *dL_dsh, float *dL_dcov3D, float *dL_dscales, float *dL_drotations,
                               const float *part_scale, void *stream_);
int sgr_render_forward_wants_prepare(const SgrProblem *pb, uint64_t R, bool use_aux, size_t *n_desc_out);
int sgr_render_forward_ex(const SgrProblem *pb, const uint32_t *ranges, const uint32_t *point_list, const float *rec, float *out_color,
                          float *out_depth, float *out_alpha, float *final_T, uint32_t *n_contrib, uint64_t R, void *aux_compact,
                          void *aux_ckpt_tc, void *aux_ckpt_da, void *aux_desc, uint32_t *aux_order, int prepared, int kind,
                          const SgrFusedL1Args *fused, bool bg_done, void *stream_);
int sgr_render_forward_kind(const SgrProblem *pb);
int sgr_get_forward_mode();

static int forward_launches(const SgrProblem *pb, uint64_t capacity, uint64_t R, SgrForwardState *st, float *out_color, float *out_depth,
                            float *out_alpha, int32_t *out_radii, uint64_t *nr_pinned_host, bool preprocess_done, void *caller_clear,
                            uint64_t caller_clear_bytes, const SgrL1Epilogue *l1 /* st->fused_bwd: the epilogue the compositing kernel absorbs */,
                            hipStream_t stream) {
    char *geom = (char *)st->geom, *binning = (char *)st->binning, *image = (char *)st->image;
    float *rec = (float *)(geom + st->off_rec);
    uint32_t *rect = (uint32_t *)(geom + st->off_rect);
    uint8_t *clamped = pb->shs ? (uint8_t *)(geom + st->off_clamped) : nullptr;
    uint32_t *block_offsets = (uint32_t *)(geom + st->off_block_offsets);
    uint64_t *num_rendered = (uint64_t *)(geom + st->off_num_rendered);
    // small sync-free launches: the block-sum scan (F2), the clear of the tile ranges and the copy of the instance count to the
    // caller's pinned slot are folded into the duplicate kernel (three launches fewer)
    const uint64_t nblk = (uint64_t)sgr_preprocess_blocks_per_view(pb->P) * pb->n_views;
    const bool self_scan = !preprocess_done && capacity > 0 && nblk <= 4096;
    // the fused step, sync-free mode (preprocess has not run yet): every tile is pre-filled with the background, its loss shares and dL/dcolor by
    // extra workgroups of the preprocess launch (common.h SgrBgJob); the compositing kernel's empty tiles then have nothing to do (bg_done)
    SgrBgJob bgj;
    memset(&bgj, 0, sizeof(bgj));
    int bg_done = 0;
    if (st->fused_bwd && !preprocess_done) {
        bgj.enabled = 1; bgj.W = pb->W; bgj.H = pb->H; bgj.Tx = (pb->W + SGR_TILE - 1) / SGR_TILE;
        bgj.tiles_per_view = (uint32_t)bgj.Tx * (uint32_t)((pb->H + SGR_TILE - 1) / SGR_TILE);
        bgj.tiles_total = bgj.tiles_per_view * (uint32_t)pb->n_views;
        bgj.bg = pb->bg; bgj.out_color = out_color; bgj.out_depth = out_depth; bgj.out_alpha = out_alpha;
        bgj.final_T = (float *)(image + st->off_final_T); bgj.n_contrib = (uint32_t *)(image + st->off_n_contrib); bgj.clamped = pb->color_clamped;
        if (st->fused_bwd) { bgj.target = l1->target; bgj.mask = l1->mask; bgj.weight = l1->weight; bgj.gimg = l1->grad_color;
                             bgj.loss_part = (float *)(image + st->off_loss_part); }
        bg_done = 1;
    }
    bool occ_zeroed = false;
    if (!preprocess_done) {
        // (the single-view path's tile-occupancy flags sit at the head of the sort workspace and must be zero when the emission kernel starts)
        bgj.zero_ptr = (uint32_t *)(binning + st->off_sort_ws); bgj.zero_words = SGR_BIN_OCC_WORDS;
        occ_zeroed = true;
        if (sgr_preprocess_forward_ex(pb, rec, out_radii, rect, clamped, block_offsets, num_rendered, capacity, self_scan, &bgj, stream)) return 1;
        if (nr_pinned_host && !self_scan) SGR_CHECK_HIP(hipMemcpyAsync(nr_pinned_host, num_rendered + 2, 8, hipMemcpyDeviceToHost, stream));   // count | overflow << 63
        st->nr_by_copy = self_scan ? 0 : 1;
    }
    int32_t in_b = 0;
    const bool aux_on = st->with_aux != 0;
    // the segment-parallel forward wants a work order (longest tile lists first) and cleared bucket descriptors: on the single-view path the
    // per-tile sort launch writes the order (class-major form) and the emission kernel clears the descriptors; else render.hip's own prepare kernel
    size_t n_desc = 0;
    const bool want_prep = sgr_render_forward_wants_prepare(pb, R, aux_on, &n_desc) != 0;
    int order_kind = 0;
    // buffers the later stages expect zeroed are cleared on the side by the duplicate kernel: the backward's flags (a fused step's own backward
    // has its own: off_flags_fused), an optional caller buffer (the fused loss node's accumulators), the bucket descriptors
    uint32_t *const own_flags = !aux_on ? nullptr : (uint32_t *)(image + (st->fused_bwd ? st->off_flags_fused : st->off_flags));
    uint32_t *clear_ptr[3] = {own_flags, (uint32_t *)caller_clear, aux_on ? (uint32_t *)(image + st->off_desc) : nullptr};
    const uint64_t clear_words[3] = {aux_on ? (R + 0) : 0, (caller_clear_bytes + 3) / 4, (uint64_t)n_desc * 2};
    int clear_done[3] = {0, 0, 0};
    // the empty tiles' outputs, unless the preprocess launch pre-filled every tile (fused step): written by the empty tiles' own workgroups of
    // the single-view path's sort launch
    SgrBgJob bgs;
    memset(&bgs, 0, sizeof(bgs));
    if (!bg_done) {
        bgs.enabled = 1; bgs.W = pb->W; bgs.H = pb->H; bgs.Tx = (pb->W + SGR_TILE - 1) / SGR_TILE;
        bgs.tiles_per_view = (uint32_t)bgs.Tx * (uint32_t)((pb->H + SGR_TILE - 1) / SGR_TILE);
        bgs.tiles_total = bgs.tiles_per_view * (uint32_t)pb->n_views;
        bgs.bg = pb->bg; bgs.out_color = out_color; bgs.out_depth = out_depth; bgs.out_alpha = out_alpha;
        bgs.final_T = (float *)(image + st->off_final_T); bgs.n_contrib = (uint32_t *)(image + st->off_n_contrib); bgs.clamped = pb->color_clamped;
        if (st->fused_bwd) { bgs.target = l1->target; bgs.mask = l1->mask; bgs.weight = l1->weight; bgs.gimg = l1->grad_color;       // (exact mode)
                             bgs.loss_part = (float *)(image + st->off_loss_part); }
    }
    if (sgr_bin_ex(pb, out_radii, rect, block_offsets, R, capacity > 0 ? num_rendered : nullptr, (uint64_t *)(binning + st->off_keys_a),
                   (uint64_t *)(binning + st->off_keys_b), (uint32_t *)(binning + st->off_vals_a), (uint32_t *)(binning + st->off_vals_b),
                   binning + st->off_sort_ws, (size_t)sgr_bin_workspace_bytes(R, (uint64_t)((pb->W + SGR_TILE - 1) / SGR_TILE) * ((pb->H + SGR_TILE - 1) / SGR_TILE) * pb->n_views),
                   (uint32_t *)(image + st->off_ranges), &in_b, self_scan, self_scan ? nr_pinned_host : nullptr,
                   want_prep ? (uint32_t *)(image + st->off_order) : nullptr, &order_kind, bg_done ? nullptr : &bgs, occ_zeroed,
                   clear_ptr, clear_words, clear_done, /*first_index=*/aux_on, /*sorted_keys=*/g_keep_sorted_keys != 0, stream)) return 1;
    st->order_kind = order_kind == 1 ? 1 : 0;                        // (the form of the order: what a later depth/alpha checkpoint pass must know)
    const int prepared = (order_kind == 1 ? 2 : (order_kind == 2 ? 1 : 0)) | (clear_done[2] ? 4 : 0);
    st->flags_cleared = st->fused_bwd ? 0 : clear_done[0];          // (of off_flags: what an ordinary backward writes)
    int fused_flags_cleared = st->fused_bwd ? clear_done[0] : 0;
    if (caller_clear && caller_clear_bytes && !clear_done[1]) SGR_CHECK_HIP(hipMemsetAsync(caller_clear, 0, (size_t)caller_clear_bytes, stream));
    st->result_in_b = in_b;
    const uint32_t *point_list = (const uint32_t *)(binning + (in_b ? st->off_vals_b : st->off_vals_a));
    st->fwd_kind = sgr_render_forward_kind(pb);
    SgrFusedL1Args fa;
    if (st->fused_bwd) {
        // the fused kernel's backward writes the flags of the survivors it looks at: the rest must be clear before it starts.  These flags are
        // the fused records' OWN (off_flags_fused): an ordinary backward on the same forward state (retain_graph, a second upstream gradient)
        // rewrites off_flags for ITS scratch records and must not touch what a later gather-only backward reads
        if (!fused_flags_cleared) { SGR_CHECK_HIP(hipMemsetAsync(image + st->off_flags_fused, 0, (size_t)R * 4, stream)); fused_flags_cleared = 1; }
        fa.target = l1->target; fa.mask = l1->mask; fa.weight = l1->weight; fa.gimg = l1->grad_color;
        fa.loss_part = (float *)(image + st->off_loss_part); fa.loss_per_view = l1->loss_per_view; fa.loss_total = l1->loss_total;
        fa.rect = rect; fa.part = (float *)(image + st->off_part); fa.flags = (uint32_t *)(image + st->off_flags_fused);
    }
    const int rc = sgr_render_forward_ex(pb, (const uint32_t *)(image + st->off_ranges), point_list, rec, out_color, out_depth, out_alpha,
                              (float *)(image + st->off_final_T), (uint32_t *)(image + st->off_n_contrib), R,
                              aux_on ? image + st->off_compact : nullptr, aux_on ? image + st->off_ckpt_tc : nullptr,
                              (aux_on && !st->aux_no_da) ? image + st->off_ckpt_da : nullptr, aux_on ? image + st->off_desc : nullptr,
                              (uint32_t *)(image + st->off_order), prepared, st->fwd_kind, st->fused_bwd ? &fa : nullptr, bg_done != 0, stream);
    if (rc || !st->fused_bwd) return rc;
    // fused step: the compositing kernel left the loss shares and dL/dcolor; the bucket backward of dL/dloss = 1 follows at once
    // (its spare workgroup sums the loss shares), so that the caller's backward only gathers
    SgrProblem pbb = *pb;
    pbb.clamp_grad = 0;                    // (the loss has its own clamp; dL/dcolor is w.r.t. the unclamped colour)
    return sgr_render_backward_ex(&pbb, (const uint32_t *)(image + st->off_ranges), rec, rect, (const uint32_t *)(image + st->off_n_contrib), out_color,
                                  out_depth, out_alpha, l1->grad_color, nullptr, nullptr, nullptr, R, image + st->off_compact, image + st->off_ckpt_tc, nullptr,
                                  image + st->off_desc, fa.part, fa.flags, /*flags_cleared=*/true, &fa, stream);
}

// l1 != NULL with l1->fuse_backward: the caller is sgr_rasterize_forward_l1 and allows the single-view FUSED step (render.hip, FusedL1) --
// taken when the launch uses the segment-parallel kernel and records auxiliary outputs without depth/alpha checkpoints; st->fused_bwd tells
static int rasterize_forward_impl(const SgrProblem *pb, uint64_t capacity, int32_t with_aux, sgr_alloc_fn alloc, void *user,
                                  float *out_color, float *out_depth, float *out_alpha, int32_t *out_radii,
                                  uint64_t *nr_pinned_host, void *nr_event, void *caller_clear, uint64_t caller_clear_bytes,
                                  SgrForwardState *st, const SgrL1Epilogue *l1, void *stream_) {
    if (!pb || !st || !out_color || !out_depth || !out_alpha || (!out_radii && pb->P > 0)) { sgr_set_error("sgr_rasterize_forward: NULL argument"); return 1; }
    // alloc == NULL: the caller pre-allocated the three blobs (state->geom / binning / image with their *_bytes capacities, e.g. from the
    // sizes a previous call with the same shapes reported): no callbacks.  A blob that is too small -> return 2 with the needed sizes in
    // state->*_bytes (nothing launched yet for the geometry blob; the caller retries with the allocator).
    const SgrForwardState pre = *st;
    if (!alloc && !(pre.geom && pre.binning && pre.image)) { sgr_set_error("sgr_rasterize_forward: no allocator and no pre-allocated blobs"); return 1; }
    auto get_blob = [&](int which, uint64_t bytes) -> char * {
        if (alloc) return alloc(user, which, (size_t)bytes);
        const uint64_t have = which == 0 ? pre.geom_bytes : (which == 1 ? pre.binning_bytes : pre.image_bytes);
        return have >= bytes ? (char *)(which == 0 ? pre.geom : (which == 1 ? pre.binning : pre.image)) : nullptr;
    };
    hipStream_t stream = (hipStream_t)stream_;
    const uint64_t nq = (uint64_t)pb->n_views * (uint64_t)(pb->P > 0 ? pb->P : 0);
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint64_t tiles_total = (uint64_t)Tx * Ty * pb->n_views;
    const uint64_t hw = (uint64_t)pb->H * pb->W * pb->n_views;
    const uint64_t nbo = (uint64_t)sgr_preprocess_blocks_per_view(pb->P) * pb->n_views + 1;
    memset(st, 0, sizeof(*st));
    // ---- geometry blob
    uint64_t o = 0;
    st->off_rec = o; o = align_up(o + (nq ? nq : 1) * SGR_REC_STRIDE * 4);
    st->off_rect = o; o = align_up(o + (nq ? nq : 1) * 16);
    st->off_clamped = o; o = align_up(o + (pb->shs ? (nq ? nq : 1) : 0));
    st->off_block_offsets = o; o = align_up(o + 2 * nbo * 4);
    st->off_num_rendered = o; o = align_up(o + 32);
    st->geom_bytes = o;
    char *geom = get_blob(0, o);
    if (!geom) { sgr_set_error(alloc ? "geometry allocator returned NULL" : "pre-allocated geometry blob too small"); return alloc ? 1 : 2; }
    st->geom = geom;
    uint64_t *num_rendered = (uint64_t *)(geom + st->off_num_rendered);
    uint64_t R = 0;
    bool preprocess_done = false;
    if (pb->P > 0 && capacity == 0) {
        // exact mode: preprocess first, then the one blocking read of num_rendered per batched forward (upstream: one per VIEW)
        if (sgr_preprocess_forward(pb, (float *)(geom + st->off_rec), out_radii, (uint32_t *)(geom + st->off_rect),
                                   pb->shs ? (uint8_t *)(geom + st->off_clamped) : nullptr, (uint32_t *)(geom + st->off_block_offsets),
                                   num_rendered, 0, stream)) return 1;
        uint64_t host2[2] = {0, 0};
        SGR_CHECK_HIP(hipMemcpyAsync(nr_pinned_host ? nr_pinned_host : host2, num_rendered, 16, hipMemcpyDeviceToHost, stream));
        SGR_CHECK_HIP(hipStreamSynchronize(stream));
        const uint64_t *h = nr_pinned_host ? nr_pinned_host : host2;
        if (h[1]) { sgr_set_error("num_rendered %llu exceeds the 32-bit instance index", (unsigned long long)h[0]); return 1; }
        R = h[0];
        preprocess_done = true;
    } else if (pb->P > 0) {
        R = capacity;                             // sync-free: buffers pre-sized, the true count goes to the host asynchronously
    } else {
        SGR_CHECK_HIP(hipMemsetAsync(num_rendered, 0, 32, stream));
        preprocess_done = true;
    }
    st->R_alloc = R;
    st->true_rendered = capacity > 0 ? ~0ull : R;
    // ---- binning blob
    const uint64_t Rn = R ? R : 1;
    o = 0;
    st->off_keys_a = o; o = align_up(o + Rn * 8);
    st->off_keys_b = o; o = align_up(o + Rn * 8);
    st->off_vals_a = o; o = align_up(o + Rn * 4);
    st->off_vals_b = o; o = align_up(o + Rn * 4);
    st->off_sort_ws = o; o = align_up(o + sgr_bin_workspace_bytes(R, tiles_total));
    st->binning_bytes = o;
    char *binning = get_blob(1, o);
    if (!binning) { sgr_set_error(alloc ? "binning allocator returned NULL" : "pre-allocated binning blob too small"); return alloc ? 1 : 2; }
    st->binning = binning;
    // ---- image blob
    const bool aux_on = with_aux && R > 0;
    const uint64_t NS = sgr_bucket_slots(R, tiles_total);
    st->NS = NS; st->with_aux = aux_on ? 2 : 0;                             // 0 none, 2 one checkpoint per 16-survivor row (1 was round 2's compact layout)
    // with_aux & 2: the (depth, alpha) checkpoints -- a third of the checkpoint stream -- are not written now: only a backward that is
    // handed dL/ddepth or dL/dalpha reads them (no call path of the reference does), and it then produces them with a second compositing pass
    st->aux_no_da = (aux_on && (with_aux & 2)) ? 1 : 0;
    o = 0;
    st->off_ranges = o; o = align_up(o + tiles_total * 8);
    st->off_final_T = o; o = align_up(o + hw * 4);
    st->off_n_contrib = o; o = align_up(o + hw * 4);
    // (the segment-parallel forward's work order; class-major form on the single-view path: header + 33 class regions, common.h)
    st->off_order = o; o = align_up(o + (tiles_total <= 2048 ? SGR_ORDER_HDR_WORDS * 4 + 33 * tiles_total * 16 : tiles_total * 16));
    if (aux_on) {
        st->off_flags = o; o = align_up(o + R * 4);                // one byte per (tile instance, quadrant): partial record written by the backward
        st->off_compact = o; o = align_up(o + 4 * R * 8);
        const uint64_t rows = 4;                                            // checkpoint records per pixel and 64-survivor bucket
        st->off_ckpt_tc = o; o = align_up(o + 4 * NS * rows * 64 * 16);
        st->off_ckpt_da = o; o = align_up(o + (st->aux_no_da ? 0 : 4 * NS * rows * 64 * 8));
        st->off_desc = o; o = align_up(o + 4 * NS * 8);
        if (l1 && l1->fuse_backward && st->aux_no_da && sgr_render_forward_kind(pb) == 2 && sgr_fused_step_enabled()) {
            st->fused_bwd = 1;
            st->off_part = o; o = align_up(o + R * 4 * SGR_PART_FLOATS * 4);        // the backward's partial records are written by the forward's launch
            st->off_loss_part = o; o = align_up(o + tiles_total * 4 * 4);
            st->off_flags_fused = o; o = align_up(o + R * 4);          // the flags of THOSE records (off_flags: of an ordinary backward's scratch records)
        }
    }
    st->image_bytes = o;
    char *image = get_blob(2, o);
    if (!image) { sgr_set_error(alloc ? "image allocator returned NULL" : "pre-allocated image blob too small"); return alloc ? 1 : 2; }
    st->image = image;

    if (forward_launches(pb, capacity, R, st, out_color, out_depth, out_alpha, out_radii, capacity > 0 ? nr_pinned_host : nullptr,
                         preprocess_done, caller_clear, caller_clear_bytes, l1, stream)) return 1;
    // (the event is only needed when the count travels by an asynchronous copy; a kernel's own 8-byte store is polled, and an event
    // record between the compositing kernel and whatever the caller queues next is a ~5 us bubble on the GPU)
    if (capacity > 0 && nr_event && st->nr_by_copy) SGR_CHECK_HIP(hipEventRecord((hipEvent_t)nr_event, stream));
    return 0;
}

extern "C" int sgr_rasterize_forward(const SgrProblem *pb, uint64_t capacity, int32_t with_aux, sgr_alloc_fn alloc, void *user,
                                     float *out_color, float *out_depth, float *out_alpha, int32_t *out_radii,
                                     uint64_t *nr_pinned_host, void *nr_event, void *caller_clear, uint64_t caller_clear_bytes,
                                     SgrForwardState *st, void *stream_) {
    return rasterize_forward_impl(pb, capacity, with_aux, alloc, user, out_color, out_depth, out_alpha, out_radii, nr_pinned_host, nr_event,
                                  caller_clear, caller_clear_bytes, st, nullptr, stream_);
}

extern "C" int sgr_clamped_l1_loss(int32_t n_views, int32_t H, int32_t W, const float *color, const float *target, const float *mask, float weight,
                                   float *grad_color, float *loss_per_view, float *loss_total, int32_t sums_already_zero, void *stream_);

extern "C" int sgr_rasterize_forward_l1(const SgrProblem *pb, uint64_t capacity, int32_t with_aux, sgr_alloc_fn alloc, void *user,
                                        float *out_color, float *out_depth, float *out_alpha, int32_t *out_radii,
                                        uint64_t *nr_pinned_host, void *nr_event, void *caller_clear, uint64_t caller_clear_bytes,
                                        SgrForwardState *st, const SgrL1Epilogue *l1, void *stream_) {
    if (!l1 || !l1->target || !l1->grad_color || !l1->loss_per_view) { sgr_set_error("sgr_rasterize_forward_l1: NULL epilogue argument"); return 1; }
    const int rc = rasterize_forward_impl(pb, capacity, with_aux, alloc, user, out_color, out_depth, out_alpha, out_radii, nr_pinned_host, nr_event,
                                          caller_clear, caller_clear_bytes, st, l1, stream_);
    if (rc) return rc;
    if (st->fused_bwd) return 0;               // loss, dL/dcolor and the compositing backward ran inside the compositing kernel
    return sgr_clamped_l1_loss(pb->n_views, pb->H, pb->W, out_color, l1->target, l1->mask, l1->weight, l1->grad_color, l1->loss_per_view,
                               l1->loss_total, l1->sums_already_zero, stream_);
}

extern "C" int sgr_rasterize_backward(const SgrProblem *pb, const SgrForwardState *st, const int32_t *radii, const float *out_color,
                                      const float *out_depth, const float *out_alpha, const float *grad_color,
                                      const float *grad_depth, const float *grad_alpha, const float *grad_color_scale,
                                      sgr_alloc_fn alloc, void *user,
                                      float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors, float *dL_dsh,
                                      float *dL_dcov3D, float *dL_dscales, float *dL_drotations, void *stream_) {
    if (!pb || !st || !alloc) { sgr_set_error("sgr_rasterize_backward: NULL argument"); return 1; }
    if (pb->P <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const uint64_t nq = (uint64_t)pb->n_views * (uint64_t)pb->P;
    if (st->with_aux == 0) {
        // A forward without auxiliary outputs: either nothing was rendered (R == 0: every Gaussian culled or off screen -- sgr_rasterize_forward
        // then records no lists or checkpoints whatever the caller asked for), and every gradient is zero; or the caller said it would not
        // differentiate (with_aux = 0).  (Until round 4 the published pixel-parallel backward ran here.)
        if (st->R_alloc != 0) { sgr_set_error("sgr_rasterize_backward: the forward ran without auxiliary outputs (with_aux = 0): nothing to differentiate through"); return 1; }
        const uint64_t ns = (uint64_t)(pb->n_views / pb->views_per_subject) * (uint64_t)pb->P;
        if (dL_dmeans3D) SGR_CHECK_HIP(hipMemsetAsync(dL_dmeans3D, 0, ns * 3 * 4, stream));
        if (dL_dmeans2D) SGR_CHECK_HIP(hipMemsetAsync(dL_dmeans2D, 0, nq * 3 * 4, stream));
        if (dL_dopacity) SGR_CHECK_HIP(hipMemsetAsync(dL_dopacity, 0, ns * 4, stream));
        if (dL_dcolors) SGR_CHECK_HIP(hipMemsetAsync(dL_dcolors, 0, ns * 3 * 4, stream));
        if (dL_dsh && pb->shs) SGR_CHECK_HIP(hipMemsetAsync(dL_dsh, 0, ns * (uint64_t)pb->M * 3 * 4, stream));
        if (dL_dcov3D) SGR_CHECK_HIP(hipMemsetAsync(dL_dcov3D, 0, ns * 6 * 4, stream));
        if (dL_dscales && pb->scales) SGR_CHECK_HIP(hipMemsetAsync(dL_dscales, 0, ns * 3 * 4, stream));
        if (dL_drotations && pb->scales) SGR_CHECK_HIP(hipMemsetAsync(dL_drotations, 0, ns * 4 * 4, stream));
        return 0;
    }
    const char *geom = (const char *)st->geom, *binning = (const char *)st->binning, *image = (const char *)st->image;
    const float *rec = (const float *)(geom + st->off_rec);
    const uint32_t *rect = (const uint32_t *)(geom + st->off_rect);
    if (!grad_color) {
        // a FUSED forward (st->fused_bwd) already ran the compositing backward for ITS OWN loss: what is left is the gather, with the upstream
        // scalar dL/dloss (grad_color_scale, NULL = 1) multiplied onto the gathered sums -- every gradient is linear in them
        if (!st->fused_bwd) { sgr_set_error("sgr_rasterize_backward: grad_color is NULL but the forward was not a fused rasterize + L1 step"); return 1; }
        if (grad_depth || grad_alpha) { sgr_set_error("sgr_rasterize_backward: the fused step's own backward has no dL/ddepth / dL/dalpha: pass grad_color"); return 1; }
        return sgr_preprocess_backward_ex(pb, radii, pb->shs ? (const uint8_t *)(geom + st->off_clamped) : nullptr, rect, (const float *)(image + st->off_part),
                                          (const uint32_t *)(image + st->off_flags_fused), st->R_alloc, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh,
                                          dL_dcov3D, dL_dscales, dL_drotations, grad_color_scale, stream_);
    }
    // scratch: partial records [4*R][10] f32 (the flags [R] u32 live in the forward's image blob)
    const uint64_t part_bytes = align_up(st->R_alloc * 4 * SGR_PART_FLOATS * 4);
    // depth / alpha gradients on a forward that left their checkpoints out: room for them in the scratch blob, filled by a second
    // compositing pass below (same kernels, same lists: the same values the forward would have stored)
    const bool da_refill = st->aux_no_da && (grad_depth || grad_alpha);
    const uint64_t da_bytes = da_refill ? align_up(4 * st->NS * 4 * 64 * 8) : 0;
    char *scratch = alloc(user, 3, (size_t)(part_bytes + da_bytes));
    if (!scratch) { sgr_set_error("scratch allocator returned NULL"); return 1; }
    float *part = (float *)scratch;
    uint32_t *flags = (uint32_t *)((char *)st->image + st->off_flags);
    const void *ckpt_da = !st->aux_no_da ? image + st->off_ckpt_da : nullptr;
    if (da_refill) {
        char *im = (char *)st->image;
        void *da = scratch + part_bytes;
        const uint32_t *point_list = (const uint32_t *)(binning + (st->result_in_b ? st->off_vals_b : st->off_vals_a));
        if (sgr_render_forward_ex(pb, (const uint32_t *)(image + st->off_ranges), point_list, rec, (float *)out_color, (float *)out_depth, (float *)out_alpha,
                                  (float *)(im + st->off_final_T), (uint32_t *)(im + st->off_n_contrib), st->R_alloc, im + st->off_compact, nullptr, da,
                                  im + st->off_desc, (uint32_t *)(im + st->off_order), /*prepared=*/(st->order_kind ? 2 : 1) | 4, st->fwd_kind, nullptr, false, stream_)) return 1;
        ckpt_da = da;
    }
    if (sgr_render_backward_ex(pb, (const uint32_t *)(image + st->off_ranges), rec, rect, (const uint32_t *)(image + st->off_n_contrib),
                               out_color, out_depth, out_alpha, grad_color, grad_depth, grad_alpha, grad_color_scale, st->R_alloc,
                               image + st->off_compact, image + st->off_ckpt_tc, ckpt_da, image + st->off_desc, part, flags,
                               st->flags_cleared != 0, nullptr, stream_))
        return 1;
    return sgr_preprocess_backward_ex(pb, radii, pb->shs ? (const uint8_t *)(geom + st->off_clamped) : nullptr, rect, part, flags, st->R_alloc,
                                      dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh, dL_dcov3D, dL_dscales, dL_drotations,
                                      nullptr, stream_);
}
