// rasterize.hip -- the one-call entry points: what upstream's pybind `_C.rasterize_gaussians` /
// `_C.rasterize_gaussians_backward` are for /root/reference/core/gaussians/gs.py:98-106 and train_vae.py:166.
// Like upstream's CudaRasterizer::Rasterizer::forward(geomFunc, binningFunc, imageFunc, ...) the caller passes
// ALLOCATOR CALLBACKS (PyTorch owns every byte: the callback resizes a uint8 tensor and returns its data_ptr), the
// library lays its buffers out inside the three blobs and launches the whole chain
//     preprocess -> scan -> [num_rendered] -> duplicate -> radix sort -> ranges -> composite
// from ONE host call, for all views of the batch.  Host-side cost matters here: at the reference's sizes the
// GPU needs ~0.35 ms for a 512^2 view fwd+bwd, which a Python-driven launch sequence cannot feed.
#include <string.h>

#include "common.h"

namespace {
inline uint64_t align_up(uint64_t v, uint64_t a = 256) { return (v + a - 1) / a * a; }
}

extern "C" int sgr_rasterize_forward(const SgrProblem *pb, uint64_t capacity, int32_t with_aux, sgr_alloc_fn alloc, void *user,
                                     float *out_color, float *out_depth, float *out_alpha, int32_t *out_radii,
                                     uint64_t *nr_pinned_host, void *nr_event, SgrForwardState *st, void *stream_) {
    if (!pb || !alloc || !st || !out_color || !out_depth || !out_alpha || (!out_radii && pb->P > 0)) { sgr_set_error("sgr_rasterize_forward: NULL argument"); return 1; }
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
    st->off_rect = o; o = align_up(o + (nq ? nq : 1) * 8);
    st->off_clamped = o; o = align_up(o + (pb->shs ? (nq ? nq : 1) : 0));
    st->off_block_offsets = o; o = align_up(o + 2 * nbo * 4);
    st->off_num_rendered = o; o = align_up(o + 16);
    st->geom_bytes = o;
    char *geom = alloc(user, 0, (size_t)o);
    if (!geom) { sgr_set_error("geometry allocator returned NULL"); return 1; }
    st->geom = geom;
    float *rec = (float *)(geom + st->off_rec);
    uint32_t *rect = (uint32_t *)(geom + st->off_rect);
    uint8_t *clamped = pb->shs ? (uint8_t *)(geom + st->off_clamped) : nullptr;
    uint32_t *block_offsets = (uint32_t *)(geom + st->off_block_offsets);
    uint64_t *num_rendered = (uint64_t *)(geom + st->off_num_rendered);
    uint64_t R = 0;
    if (pb->P > 0) {
        if (sgr_preprocess_forward(pb, rec, out_radii, rect, clamped, block_offsets, num_rendered, capacity, stream)) return 1;
        if (capacity > 0) {
            R = capacity;                         // sync-free: buffers pre-sized, the true count goes to the host asynchronously
            if (nr_pinned_host) {
                SGR_CHECK_HIP(hipMemcpyAsync(nr_pinned_host, num_rendered, 16, hipMemcpyDeviceToHost, stream));
                if (nr_event) SGR_CHECK_HIP(hipEventRecord((hipEvent_t)nr_event, stream));
            }
        } else {
            uint64_t host2[2] = {0, 0};           // exact mode: the one blocking read per batched forward (upstream: one per VIEW)
            SGR_CHECK_HIP(hipMemcpyAsync(nr_pinned_host ? nr_pinned_host : host2, num_rendered, 16, hipMemcpyDeviceToHost, stream));
            SGR_CHECK_HIP(hipStreamSynchronize(stream));
            const uint64_t *h = nr_pinned_host ? nr_pinned_host : host2;
            if (h[1]) { sgr_set_error("num_rendered %llu exceeds the 32-bit instance index", (unsigned long long)h[0]); return 1; }
            R = h[0];
        }
    } else {
        SGR_CHECK_HIP(hipMemsetAsync(num_rendered, 0, 16, stream));
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
    const uint64_t sort_ws = sgr_bin_workspace_bytes(R);
    st->off_sort_ws = o; o = align_up(o + sort_ws);
    st->binning_bytes = o;
    char *binning = alloc(user, 1, (size_t)o);
    if (!binning) { sgr_set_error("binning allocator returned NULL"); return 1; }
    st->binning = binning;
    // ---- image blob
    const bool aux_on = with_aux && R > 0;
    const uint64_t NS = sgr_bucket_slots(R, tiles_total);
    st->NS = NS; st->with_aux = aux_on ? 1 : 0;
    o = 0;
    st->off_ranges = o; o = align_up(o + tiles_total * 8);
    st->off_final_T = o; o = align_up(o + hw * 4);
    st->off_n_contrib = o; o = align_up(o + hw * 4);
    if (aux_on) {
        st->off_compact = o; o = align_up(o + 4 * R * 8);
        st->off_ckpt_tc = o; o = align_up(o + 4 * NS * 64 * 16);
        st->off_ckpt_da = o; o = align_up(o + 4 * NS * 64 * 8);
        st->off_desc = o; o = align_up(o + 4 * NS * 8);
    }
    st->image_bytes = o;
    char *image = alloc(user, 2, (size_t)o);
    if (!image) { sgr_set_error("image allocator returned NULL"); return 1; }
    st->image = image;
    int32_t in_b = 0;
    if (sgr_bin(pb, rec, out_radii, rect, block_offsets, R, capacity > 0 ? num_rendered : nullptr, (uint64_t *)(binning + st->off_keys_a),
                (uint64_t *)(binning + st->off_keys_b), (uint32_t *)(binning + st->off_vals_a), (uint32_t *)(binning + st->off_vals_b),
                binning + st->off_sort_ws, (size_t)sort_ws, (uint32_t *)(image + st->off_ranges), &in_b, stream)) return 1;
    st->result_in_b = in_b;
    const uint32_t *point_list = (const uint32_t *)(binning + (in_b ? st->off_vals_b : st->off_vals_a));
    return sgr_render_forward(pb, (const uint32_t *)(image + st->off_ranges), point_list, rec, out_color, out_depth, out_alpha,
                              (float *)(image + st->off_final_T), (uint32_t *)(image + st->off_n_contrib), R,
                              aux_on ? image + st->off_compact : nullptr, aux_on ? image + st->off_ckpt_tc : nullptr,
                              aux_on ? image + st->off_ckpt_da : nullptr, aux_on ? image + st->off_desc : nullptr, stream);
}

extern "C" int sgr_rasterize_backward(const SgrProblem *pb, const SgrForwardState *st, const int32_t *radii, const float *out_color,
                                      const float *out_depth, const float *out_alpha, const float *grad_color,
                                      const float *grad_depth, const float *grad_alpha, sgr_alloc_fn alloc, void *user,
                                      float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors, float *dL_dsh,
                                      float *dL_dcov3D, float *dL_dscales, float *dL_drotations, void *stream_) {
    if (!pb || !st || !alloc) { sgr_set_error("sgr_rasterize_backward: NULL argument"); return 1; }
    if (pb->P <= 0) return 0;
    const uint64_t nq = (uint64_t)pb->n_views * (uint64_t)pb->P;
    const bool aux_on = st->with_aux != 0;
    // scratch: bucket-parallel path = partial records [4*R][12] f32 + flags [R] u32; pixel-parallel path = grec [nq][12] f32
    const uint64_t part_bytes = align_up(st->R_alloc * 4 * SGR_REC_FLOATS * 4);
    const uint64_t scratch_bytes = aux_on ? part_bytes + align_up(st->R_alloc * 4) : nq * SGR_REC_FLOATS * 4;
    char *scratch = alloc(user, 3, (size_t)scratch_bytes);
    if (!scratch) { sgr_set_error("scratch allocator returned NULL"); return 1; }
    float *part = aux_on ? (float *)scratch : nullptr;
    uint32_t *flags = aux_on ? (uint32_t *)(scratch + part_bytes) : nullptr;
    float *grec = aux_on ? nullptr : (float *)scratch;
    const char *geom = (const char *)st->geom, *binning = (const char *)st->binning, *image = (const char *)st->image;
    const float *rec = (const float *)(geom + st->off_rec);
    const uint32_t *point_list = (const uint32_t *)(binning + (st->result_in_b ? st->off_vals_b : st->off_vals_a));
    if (sgr_render_backward(pb, (const uint32_t *)(image + st->off_ranges), point_list, rec, (const float *)(image + st->off_final_T),
                            (const uint32_t *)(image + st->off_n_contrib), out_color, out_depth, out_alpha, grad_color, grad_depth,
                            grad_alpha, st->R_alloc, aux_on ? image + st->off_compact : nullptr, aux_on ? image + st->off_ckpt_tc : nullptr,
                            aux_on ? image + st->off_ckpt_da : nullptr, aux_on ? image + st->off_desc : nullptr, grec, part, flags, stream_))
        return 1;
    return sgr_preprocess_backward(pb, radii, pb->shs ? (const uint8_t *)(geom + st->off_clamped) : nullptr, grec, rec, part, flags,
                                   dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh, dL_dcov3D, dL_dscales, dL_drotations,
                                   stream_);
}
