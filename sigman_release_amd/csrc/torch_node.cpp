// torch_node.cpp -- the upstream-signature single-view op (`rasterize_gaussians`, what /root/reference/core/gaussians/gs.py:98-106
// reaches once per view through GaussianRasterizer.forward, and train_vae.py:166 through its backward) as a C++ autograd node
// above the C ABI of libsigman_gsplat.so.
//
// Why it exists: the reference calls the rasterizer B*V = 64 times per step from a Python loop.  With the node written in Python
// (sigman_release_amd/rasterizer.py, _RasterizeGaussians) the host needs ~150 us to issue one forward and ~65 us for one backward
// -- more than the ~170 us the single-view kernels take -- so that loop was host-bound ("the reference's own
// call pattern").  This file does the same bookkeeping natively: fp32 / contiguity normalisation, the automatic sync-free capacity
// (exact the first time a shape is seen, then pre-sized buffers with the instance count polled from a pinned word that the
// emission kernel stores, transparent exact re-run on overflow), PyTorch-owned output / workspace tensors handed to the library
// through the allocator callback of include/sigman_gsplat.h, saved state for the backward.
//
// Count check policy.  The automatic mode has to know whether the forward fitted its pre-sized buffers.  DEFAULT = INLINE: the count is
// looked at inside the call, after every kernel of the chain has been queued (the emission kernel publishes it early in the chain, so the
// wait is short), and a forward that did not fit is re-run exactly before anything is returned: the caller NEVER receives a truncated
// image and never sees an error for it.  The price: the host cannot run further ahead of the GPU than one view.
// OPT-IN SIGMAN_COUNT_CHECK=deferred (or set_count_check("deferred")): once a shape's capacity has been stable for 8 inline-checked calls
// it is raised to 2x the largest count seen and the check is DEFERRED -- the host runs ahead; the count of a forward is looked at without
// blocking before the call returns (already visible and too large -> exact re-run, transparent), else by the thread's next forwards or
// by its own backward.  An overflow found that late cannot be repaired (the truncated image has been handed out): the forward's own
// backward and check_pending() raise RuntimeError for it, an unrelated later forward only warns (the reference's caller swallows
// exceptions of the render call, core/modules/autoencoder.py:349-361, and would zero the wrong batch item); the capacity is re-learned.
//
// PyTorch is plumbing here (tensors, streams, autograd graph); every kernel launch happens inside sgr_rasterize_forward /
// sgr_rasterize_backward.  Built by csrc/Makefile with g++ against the installed torch headers (no device code in this file).
// The Python node stays as the reference implementation of the same logic (debug / prefiltered calls and builds without this
// module use it); tests/test_gpu_reference_calls.py runs both.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <array>
#include <chrono>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/sigman_gsplat.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

#define SGR_TORCH_CHECK_HIP(expr) do { hipError_t e_ = (expr); TORCH_CHECK(e_ == hipSuccess, #expr, " failed: ", hipGetErrorString(e_)); } while (0)

// ---- pinned words + events for the asynchronous instance count: one per forward whose count nobody has looked at yet
struct CountSlot { uint64_t *host = nullptr; hipEvent_t ev = nullptr; int dev = 0; };
struct Pending { CountSlot slot; std::tuple<int, int64_t, int64_t, int64_t> key; uint64_t capacity; bool by_copy; std::atomic<bool> checked{false}; uint64_t count = 0; bool overflow = false, reported = false, warned = false; int64_t id = 0; std::mutex mu; };
struct ThreadState {
    std::vector<std::shared_ptr<Pending>> pending;       // deferred checks of this thread, oldest first
};
ThreadState &tstate() { thread_local ThreadState t; return t; }
// The slot pool is PROCESS-global (one mutex): a slot acquired by the forward thread is usually released by the autograd thread (the
// backward resolves the count), so per-thread free lists never saw their slots again and every step allocated fresh pinned memory.
// Bounded: at most kMaxFreeSlots idle slots are kept, the rest is handed back to the runtime.
constexpr size_t kMaxFreeSlots = 512;
std::mutex g_slot_mu;
std::vector<CountSlot> g_free_slots;
uint64_t g_slots_created = 0;
CountSlot acquire_slot(int dev) {
    {
        std::lock_guard<std::mutex> l(g_slot_mu);
        for (size_t i = g_free_slots.size(); i-- > 0;)
            if (g_free_slots[i].dev == dev) { CountSlot s = g_free_slots[i]; g_free_slots.erase(g_free_slots.begin() + (long)i); return s; }
    }
    CountSlot s;
    s.dev = dev;
    SGR_TORCH_CHECK_HIP(hipHostMalloc((void **)&s.host, 16, hipHostMallocDefault));
    SGR_TORCH_CHECK_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    { std::lock_guard<std::mutex> l(g_slot_mu); g_slots_created++; }
    return s;
}
void release_slot(const CountSlot &s) {
    {
        std::lock_guard<std::mutex> l(g_slot_mu);
        if (g_free_slots.size() < kMaxFreeSlots) { g_free_slots.push_back(s); return; }
    }
    (void)hipEventDestroy(s.ev);
    (void)hipHostFree(s.host);
}

struct KeyState { uint64_t capacity = 0, max_count = 0; int stable = 0; bool deferred = false; };
std::mutex g_mu, g_pend_mu;            // g_pend_mu: a Pending is shared by the issuing thread's list and the autograd thread's backward
std::map<int64_t, std::shared_ptr<Pending>> g_by_id;     // deferred checks a backward may still want to look at
int64_t g_next_id = 1;
std::map<std::tuple<int, int64_t, int64_t, int64_t>, KeyState> g_keys;                            // (device, P, H, W) -> learned capacity and check policy
std::map<std::tuple<int, int64_t, int64_t, int64_t, uint64_t, int, int>, std::array<uint64_t, 3>> g_blob_sizes;

struct AllocCtx { c10::Device dev{c10::kCUDA, 0}; Tensor blob[4]; };
thread_local std::string g_alloc_error;        // why the last allocator callback of this thread returned NULL
char *alloc_cb(void *user, int32_t which, size_t bytes) {
    // no exception may cross the C ABI: an out-of-memory error of the caching allocator becomes a NULL blob, on which the library returns an error
    // before it launches anything (the nodes then release their count slot and raise with this message)
    AllocCtx *a = (AllocCtx *)user;
    try {
        a->blob[which] = at::empty({(int64_t)(bytes < 256 ? 256 : bytes)}, at::TensorOptions().dtype(at::kByte).device(a->dev));
    } catch (const std::exception &e) {
        g_alloc_error = std::string("allocating ") + std::to_string((double)bytes / (double)(1ull << 30)) + " GiB for blob " + std::to_string(which) + ": " + e.what();
        return nullptr;
    }
    return (char *)a->blob[which].data_ptr();
}

// Waits for the pinned count word: polls for up to 2 s of wall time (the store comes from a kernel early in a chain that is already queued,
// possibly behind a step's worth of earlier work), so that the callers' fallback -- a device synchronise, which drains everything queued
// and costs the host its run-ahead -- only ever runs when something is wrong.  (A fixed 200 000 polls were over in ~60 us: with the GPU a
// step behind, every wait ended in that synchronise: C2 0.154 ms per step instead of 0.138.)
inline void spin_for_count(volatile uint64_t *w, const std::atomic<bool> &resolved) {
    if (*w != ~0ull) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < 2048; k++) { if (*w != ~0ull || resolved.load(std::memory_order_acquire)) return; __builtin_ia32_pause(); }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return;
    }
}
// The wait for a count (poll, event, in the worst case a device synchronise) runs WITHOUT g_pend_mu: the mutex is process-wide, and a thread
// waiting up to 2 s for its own view's count under it would stall every other thread's forward and backward bookkeeping (other streams, other
// GPUs).  true = the word is there (or somebody else resolved the entry meanwhile).
template <class P>
inline bool wait_for_count(P &p, bool block) {
    volatile uint64_t *w = p.slot.host;
    const bool there = p.by_copy ? hipEventQuery(p.slot.ev) == hipSuccess : *w != ~0ull;
    if (there) return true;
    if (!block) return false;
    if (p.by_copy) SGR_TORCH_CHECK_HIP(hipEventSynchronize(p.slot.ev));
    else {                                                         // stored by the emission kernel itself, no event recorded: poll, then drain
        spin_for_count(w, p.checked);
        if (*w == ~0ull && !p.checked.load(std::memory_order_acquire)) {
            // drain the device the entry belongs to -- the calling thread may sit on another one (a thread driving two GPUs polls its
            // older entries under the new forward's guard)
            int cur = 0;
            SGR_TORCH_CHECK_HIP(hipGetDevice(&cur));
            if (cur != p.slot.dev) SGR_TORCH_CHECK_HIP(hipSetDevice(p.slot.dev));
            const hipError_t e = hipDeviceSynchronize();
            if (cur != p.slot.dev) SGR_TORCH_CHECK_HIP(hipSetDevice(cur));
            SGR_TORCH_CHECK_HIP(e);
            TORCH_CHECK(*w != ~0ull || p.checked.load(std::memory_order_acquire),
                        "sigman_release_amd: the forward's instance count never arrived (device ", p.slot.dev, ")");
        }
    }
    return true;
}

inline Tensor f32c(const Tensor &t) { return (t.scalar_type() == at::kFloat && t.is_contiguous()) ? t : t.to(at::kFloat).contiguous(); }
inline const float *fptr(const Tensor &t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }

// 0 = inline (default), 1 = deferred once stable (opt-in)
std::atomic<int> g_count_check{-1};
bool deferral_allowed() {
    int v = g_count_check.load();
    if (v < 0) { const char *e = getenv("SIGMAN_COUNT_CHECK"); v = (e && std::string(e) == "deferred") ? 1 : 0; g_count_check.store(v); }
    return v == 1;
}

// reads a slot's word if it is there (block: wait for it); true = resolved
bool resolve(Pending &p, bool block) {
    if (p.checked.load(std::memory_order_acquire)) return true;
    // one thread at a time looks at an entry (its own mutex: only a thread that wants THIS count waits here); the slot stays ours meanwhile
    std::unique_lock<std::mutex> own(p.mu, std::defer_lock);
    if (block) own.lock(); else if (!own.try_lock()) return false;
    if (p.checked.load(std::memory_order_acquire)) return true;
    if (!wait_for_count(p, block)) return false;
    std::lock_guard<std::mutex> pl(g_pend_mu);
    if (p.checked.load(std::memory_order_relaxed)) return true;        // the other thread was first (it also released the slot)
    const uint64_t word = *(volatile uint64_t *)p.slot.host;
    p.count = word & ~(1ull << 63); p.overflow = (word >> 63) != 0; p.checked.store(true, std::memory_order_release);
    release_slot(p.slot);
    // (an overflowed forward stays findable by its own backward, which is the call that raises for it; bounded)
    if (!p.overflow) g_by_id.erase(p.id);
    else while (g_by_id.size() > 4096) g_by_id.erase(g_by_id.begin());
    std::lock_guard<std::mutex> l(g_mu);
    KeyState &k = g_keys[p.key];
    if (p.overflow) { k = KeyState(); return true; }            // back to the learning phase (exact next call)
    if (p.count > k.max_count) k.max_count = p.count;
    if (k.deferred && p.count * 10 > k.capacity * 6) {            // getting close: more head-room for the calls to come
        uint64_t c = 2 * p.count + 4096;
        k.capacity = c > 0xFFFFFFE0ull ? 0xFFFFFFE0ull : c;
    }
    return true;
}
[[noreturn]] void raise_deferred(Pending &p) {
    { std::lock_guard<std::mutex> pl(g_pend_mu); p.reported = true; g_by_id.erase(p.id); }
    TORCH_CHECK(false, "num_rendered ", p.count, " exceeded the automatic capacity ", p.capacity, " of an EARLIER forward of this thread, whose image is "
                "therefore truncated (SIGMAN_COUNT_CHECK=deferred: the count is checked after the fact once a shape's capacity has been stable; the "
                "capacity is being re-learned now; the default inline check re-renders such a forward before it returns)");
}
// deferred checks of earlier forwards: non-blocking unless too many are outstanding.  raise_now: check_pending() -- an unrelated forward
// only warns about an earlier forward's overflow (that forward's own backward raises)
void poll_pending(bool raise_now) {
    ThreadState &t = tstate();
    std::shared_ptr<Pending> bad;
    size_t keep = 0;
    for (size_t i = 0; i < t.pending.size(); i++) {
        Pending &p = *t.pending[i];
        const bool done = p.checked || resolve(p, t.pending.size() - i > 128);
        if (done && p.overflow && !p.reported) {
            if (raise_now) { if (!bad) bad = t.pending[i]; }
            else if (!p.warned) {
                p.warned = true;
                const std::string msg = "sigman rasterizer: an earlier deferred forward needed " + std::to_string(p.count) + " tile instances but its automatic "
                                        "capacity was " + std::to_string(p.capacity) + ": its image is truncated (its backward will raise; the capacity is being re-learned)";
                // a real Python UserWarning when this runs on a Python thread (a forward always does); c10's handler otherwise
                if (PyGILState_Check()) { if (PyErr_WarnEx(PyExc_UserWarning, msg.c_str(), 1) < 0) throw pybind11::error_already_set(); }
                else TORCH_WARN(msg);
            }
        }
        if (!done) t.pending[keep++] = t.pending[i];
    }
    t.pending.resize(keep);
    if (bad) raise_deferred(*bad);
}

void check_status(int status, const char *what) {
    if (status != 0 && !g_alloc_error.empty()) { const std::string why = g_alloc_error; g_alloc_error.clear(); TORCH_CHECK(false, what, " failed: ", sgr_last_error(), " (", why, ")"); }
    TORCH_CHECK(status == 0, what, " failed: ", sgr_last_error());
}

SgrProblem make_problem(int64_t P, int64_t H, int64_t W, int64_t sh_degree, int64_t M, double tfx, double tfy, double smod, const Tensor &means3D,
                        const Tensor &opac, const Tensor &colors, const Tensor &sh, const Tensor &cov, const Tensor &scales, const Tensor &rot,
                        const Tensor &vm, const Tensor &pm, const Tensor &campos, const Tensor &bg) {
    SgrProblem pb;
    memset(&pb, 0, sizeof(pb));                                     // (optional fields: color_clamped / clamp_grad off)
    pb.P = (int32_t)P; pb.n_views = 1; pb.views_per_subject = 1; pb.H = (int32_t)H; pb.W = (int32_t)W; pb.sh_degree = (int32_t)sh_degree; pb.M = (int32_t)M;
    pb.tanfovx = (float)tfx; pb.tanfovy = (float)tfy; pb.scale_modifier = (float)smod;
    pb.means3D = fptr(means3D); pb.opacities = fptr(opac); pb.colors_precomp = fptr(colors); pb.shs = fptr(sh); pb.cov3D_precomp = fptr(cov);
    pb.scales = fptr(scales); pb.rotations = fptr(rot); pb.viewmatrix = fptr(vm); pb.projmatrix = fptr(pm); pb.campos = fptr(campos); pb.bg = fptr(bg);
    return pb;
}

struct RasterizeGaussiansNode : public torch::autograd::Function<RasterizeGaussiansNode> {
    static variable_list forward(AutogradContext *ctx, Tensor means3D_, Tensor means2D, Tensor sh_, Tensor colors_, Tensor opac_, Tensor scales_,
                                 Tensor rot_, Tensor cov_, int64_t H, int64_t W, double tfx, double tfy, Tensor bg_, double smod, Tensor vm_,
                                 Tensor pm_, int64_t sh_degree, Tensor campos_, bool grad_mode) {
        TORCH_CHECK(means3D_.dim() == 2 && means3D_.size(1) == 3, "means3D must have dimensions (num_points, 3)");
        TORCH_CHECK(means3D_.is_cuda(), "sigman_release_amd rasterizer needs tensors on a ROCm device (there is no CPU fallback)");
        const c10::Device dev = means3D_.device();
        c10::DeviceGuard guard(dev);
        const int didx = dev.index();
        const int64_t P = means3D_.size(0);
        auto opt = [](const Tensor &t) { return (t.defined() && t.numel() > 0) ? f32c(t) : Tensor(); };
        // fp32-only op: inputs are cast here, so an enclosing autocast region (gs.py:98) cannot downcast them
        const Tensor means3D = f32c(means3D_), opac = f32c(opac_).reshape({P}), sh = opt(sh_), colors = opt(colors_), scales = opt(scales_),
                     rot = opt(rot_), cov = opt(cov_);
        const Tensor vm = f32c(vm_), pm = f32c(pm_), campos = f32c(campos_), bg = f32c(bg_);
        const int64_t M = sh.defined() ? sh.size(1) : 0;
        // (grad_mode: torch.is_grad_enabled() of the CALLER -- inside a Function's forward it is always off)
        const bool wants_grad = grad_mode && (means3D_.requires_grad() || opac_.requires_grad() || (sh_.defined() && sh_.requires_grad()) ||
                                (colors_.defined() && colors_.requires_grad()) || (scales_.defined() && scales_.requires_grad()) ||
                                (rot_.defined() && rot_.requires_grad()) || (cov_.defined() && cov_.requires_grad()) ||
                                (means2D.defined() && means2D.requires_grad()));
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor color = at::empty({3, H, W}, f32), depth = at::empty({1, H, W}, f32), alpha = at::empty({1, H, W}, f32);
        Tensor radii = at::empty({P}, f32.dtype(at::kInt));
        const SgrProblem pb = make_problem(P, H, W, sh_degree, M, tfx, tfy, smod, means3D, opac, colors, sh, cov, scales, rot, vm, pm, campos, bg);
        hipStream_t stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)didx).stream();
        const int with_aux = wants_grad ? 3 : 0;          // 3: the (depth, alpha) checkpoints are produced by the backward if it is ever handed such gradients
        const auto cap_key = std::make_tuple(didx, P, H, W);
        poll_pending(false);                                                 // earlier forwards whose count nobody has looked at yet
        SgrForwardState st;
        AllocCtx ac;
        ac.dev = dev;
        std::shared_ptr<Pending> mine;                                        // set when this forward's check is deferred
        for (int attempt = 0;; attempt++) {
            uint64_t capacity = 0;
            bool deferred = false;
            { std::lock_guard<std::mutex> l(g_mu); const KeyState &k = g_keys[cap_key]; capacity = k.capacity; deferred = k.deferred && capacity > 0; }
            CountSlot slot = acquire_slot(didx);
            slot.host[0] = ~0ull; slot.host[1] = 0;                          // sentinel: "the count has not arrived yet"
            memset(&st, 0, sizeof(st));
            int status = 2;
            const auto size_key = std::make_tuple(didx, P, H, W, capacity, with_aux, sh.defined() ? 1 : 0);
            if (capacity > 0) {
                std::array<uint64_t, 3> sizes{0, 0, 0};
                bool have = false;
                { std::lock_guard<std::mutex> l(g_mu); auto it = g_blob_sizes.find(size_key); if (it != g_blob_sizes.end()) { sizes = it->second; have = true; } }
                if (have) {
                    // sizes only depend on the shapes: from the second call on the three blobs are allocated here (no allocator callbacks)
                    for (int k = 0; k < 3; k++) ac.blob[k] = at::empty({(int64_t)sizes[k]}, f32.dtype(at::kByte));
                    st.geom = ac.blob[0].data_ptr(); st.binning = ac.blob[1].data_ptr(); st.image = ac.blob[2].data_ptr();
                    st.geom_bytes = sizes[0]; st.binning_bytes = sizes[1]; st.image_bytes = sizes[2];
                    status = sgr_rasterize_forward(&pb, capacity, with_aux, nullptr, nullptr, color.data_ptr<float>(), depth.data_ptr<float>(),
                                                   alpha.data_ptr<float>(), radii.data_ptr<int32_t>(), slot.host, slot.ev, nullptr, 0, &st, stream);
                }
            }
            if (status == 2) {                                               // first call with these shapes: the library asks for memory through the callback
                memset(&st, 0, sizeof(st));
                status = sgr_rasterize_forward(&pb, capacity, with_aux, alloc_cb, &ac, color.data_ptr<float>(), depth.data_ptr<float>(),
                                               alpha.data_ptr<float>(), radii.data_ptr<int32_t>(), slot.host, slot.ev, nullptr, 0, &st, stream);
                if (status == 0 && capacity > 0) {
                    std::lock_guard<std::mutex> l(g_mu);
                    g_blob_sizes[size_key] = {st.geom_bytes < 256 ? 256 : st.geom_bytes, st.binning_bytes < 256 ? 256 : st.binning_bytes,
                                              st.image_bytes < 256 ? 256 : st.image_bytes};
                }
            }
            if (status != 0) release_slot(slot);
            check_status(status, "sgr_rasterize_forward");
            if (P == 0) { release_slot(slot); break; }
            if (capacity > 0 && deferred) {
                // opt-in steady state: the host does not wait.  One look without blocking: a count that is already visible and does not fit
                // is repaired right here (exact re-run, nothing has been handed out yet); else this forward's backward or the thread's next
                // forwards look at it
                {
                    volatile uint64_t *w = slot.host;
                    const bool there = st.nr_by_copy ? hipEventQuery(slot.ev) == hipSuccess : *w != ~0ull;
                    if (there && (*w >> 63) && attempt == 0) {
                        release_slot(slot);
                        std::lock_guard<std::mutex> l(g_mu);
                        g_keys[cap_key] = KeyState();
                        continue;
                    }
                }
                mine = std::make_shared<Pending>();
                mine->slot = slot; mine->key = cap_key; mine->capacity = capacity; mine->by_copy = st.nr_by_copy != 0;
                { std::lock_guard<std::mutex> pl(g_pend_mu); mine->id = g_next_id++; g_by_id[mine->id] = mine; }
                tstate().pending.push_back(mine);
                break;
            }
            uint64_t count = st.true_rendered, overflow = 0;
            if (capacity > 0) {
                // learning phase: everything is queued; the emission kernel stores the count into the pinned word right after it has summed the
                // block counts (large launches: an async copy behind the scan kernel -- then only the event says "complete")
                volatile uint64_t *w = slot.host;
                static const std::atomic<bool> never{false};
                if (!st.nr_by_copy) spin_for_count(w, never);
                if (st.nr_by_copy) SGR_TORCH_CHECK_HIP(hipEventSynchronize(slot.ev));
                else if (*w == ~0ull) SGR_TORCH_CHECK_HIP(hipDeviceSynchronize());
                const uint64_t word = *w;
                count = word & ~(1ull << 63); overflow = word >> 63;
            }
            release_slot(slot);
            if (overflow && attempt == 0) {                                  // does not fit the remembered capacity: re-run exactly (and re-learn it)
                std::lock_guard<std::mutex> l(g_mu);
                g_keys[cap_key] = KeyState();
                continue;
            }
            TORCH_CHECK(!overflow, "num_rendered ", count, " exceeds the 32-bit instance index");
            {
                std::lock_guard<std::mutex> l(g_mu);
                KeyState &k = g_keys[cap_key];
                if (count > k.max_count) k.max_count = count;
                if (capacity == 0 || count * 11 > capacity * 10) {          // (re-)learn: 1.3x the count, and the stability counter starts over
                    uint64_t c = count + count * 3 / 10 + 4096;
                    k.capacity = c > 0xFFFFFFE0ull ? 0xFFFFFFE0ull : c;
                    k.stable = 0;
                } else if (++k.stable >= 8 && deferral_allowed()) {            // stable: 2x the largest count seen, checks deferred from now on
                    uint64_t c = 2 * k.max_count + 4096;
                    k.capacity = c > 0xFFFFFFE0ull ? 0xFFFFFFE0ull : c;
                    k.deferred = true;
                }
            }
            break;
        }
        // ---- what the backward needs
        ctx->set_materialize_grads(false);       // unused outputs (depth / alpha on the reference path) then arrive undefined: no fill kernels
        ctx->mark_non_differentiable({radii});
        Tensor st_bytes = at::empty({(int64_t)sizeof(SgrForwardState)}, at::TensorOptions().dtype(at::kByte));
        memcpy(st_bytes.data_ptr(), &st, sizeof(st));
        ctx->save_for_backward({means3D, opac, colors, sh, cov, scales, rot, color, depth, alpha, radii, ac.blob[0], ac.blob[1], ac.blob[2], vm, pm,
                                campos, bg});
        ctx->saved_data["st"] = st_bytes;
        if (mine) ctx->saved_data["pending"] = mine->id;
        ctx->saved_data["dims"] = std::vector<int64_t>{P, H, W, sh_degree, M, means2D.defined() && means2D.requires_grad() ? 1 : 0};
        ctx->saved_data["scal"] = std::vector<double>{tfx, tfy, smod};
        return {color, radii, depth, alpha};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &means3D = saved[0], &opac = saved[1], &colors = saved[2], &sh = saved[3], &cov = saved[4], &scales = saved[5], &rot = saved[6],
                     &color = saved[7], &depth = saved[8], &alpha = saved[9], &radii = saved[10], &vm = saved[14], &pm = saved[15], &campos = saved[16],
                     &bg = saved[17];
        const auto dims = ctx->saved_data["dims"].toIntVector();
        const auto scal = ctx->saved_data["scal"].toDoubleVector();
        const int64_t P = dims[0], H = dims[1], W = dims[2], sh_degree = dims[3], M = dims[4];
        const bool want_means2D = dims[5] != 0;
        const c10::Device dev = means3D.device();
        c10::DeviceGuard guard(dev);
        SgrForwardState st;
        memcpy(&st, ctx->saved_data["st"].toTensor().data_ptr(), sizeof(st));
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor gC = grads[0].defined() ? f32c(grads[0]) : at::zeros({3, H, W}, f32);
        Tensor gD = grads[2].defined() ? f32c(grads[2]) : Tensor(), gA = grads[3].defined() ? f32c(grads[3]) : Tensor();
        Tensor d_means3D = at::empty({P, 3}, f32), d_op = at::empty({P, 1}, f32), d_cov = at::empty({P, 6}, f32);
        Tensor d_means2D = want_means2D ? at::empty({P, 3}, f32) : Tensor();
        Tensor d_col = sh.defined() ? Tensor() : at::empty({P, 3}, f32);
        Tensor d_sh = sh.defined() ? at::empty_like(sh) : Tensor();
        Tensor d_sc = scales.defined() ? at::empty({P, 3}, f32) : Tensor(), d_rot = scales.defined() ? at::empty({P, 4}, f32) : Tensor();
        const SgrProblem pb = make_problem(P, H, W, sh_degree, M, scal[0], scal[1], scal[2], means3D, opac, colors, sh, cov, scales, rot, vm, pm, campos, bg);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        AllocCtx ac;
        ac.dev = dev;
        auto mp = [](const Tensor &t) -> float * { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; };
        std::shared_ptr<Pending> mine;
        if (ctx->saved_data.count("pending")) {                               // still unchecked? (else the issuing thread has looked at it already)
            std::lock_guard<std::mutex> pl(g_pend_mu);
            auto it = g_by_id.find(ctx->saved_data["pending"].toInt());
            if (it != g_by_id.end()) mine = it->second;
        }
        if (P > 0)
            check_status(sgr_rasterize_backward(&pb, &st, radii.data_ptr<int32_t>(), color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                                gC.data_ptr<float>(), mp(gD), mp(gA), nullptr, alloc_cb, &ac, mp(d_means3D), mp(d_means2D), mp(d_op),
                                                mp(d_col), mp(d_sh), mp(d_cov), mp(d_sc), mp(d_rot), stream),
                         "sgr_rasterize_backward");
        if (mine) {                                                           // after the backward is queued: the count arrived long ago
            resolve(*mine, true);
            if (mine->overflow && !mine->reported) raise_deferred(*mine);
        }
        // gradients in forward-argument order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, then the 10 settings
        variable_list out = {d_means3D, d_means2D, d_sh, colors.defined() ? d_col : Tensor(), d_op, d_sc, d_rot, cov.defined() ? d_cov : Tensor()};
        for (int k = 0; k < 11; k++) out.push_back(Tensor());
        return out;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// Batched rasterizer + fused clamp/L1 image loss (rasterizer.py: _RasterizeL1Batched, what a training step of the batched path and
// bench.py's step call) as a C++ node, for the reference's input flavour (colors_precomp + cov3D_precomp, no mask) in the explicit
// sync-free mode (max_rendered > 0).  Why: at one view the Python node needs 110-135 us of host time per step (depending on the boost
// state of the CPU core) against 137 us of kernels -- the host was co-limiting the step (tools/long_run.py, tools/trace_vs_time.sh:
// constant kernel durations, step period 137.6 -> 153 us in discrete levels).  Same semantics as the Python node: the count is looked
// at by the forward's own backward after its kernels are queued (or by a later forward / check_pending() if no backward ever runs),
// and a forward that did not fit its capacity raises -- the caller chose the capacity.
struct BPending { CountSlot slot; uint64_t capacity = 0; bool by_copy = false; std::atomic<bool> checked{false}; bool overflow = false, reported = false; uint64_t count = 0; int64_t id = 0; std::mutex mu; };
std::map<int64_t, std::shared_ptr<BPending>> g_b_by_id;
std::map<std::tuple<int, int64_t, int64_t, int64_t, int64_t, uint64_t, int>, std::array<uint64_t, 3>> g_b_blob_sizes;
std::vector<std::shared_ptr<BPending>> &b_pending() { thread_local std::vector<std::shared_ptr<BPending>> v; return v; }

bool b_resolve(BPending &p, bool block) {
    if (p.checked.load(std::memory_order_acquire)) return true;
    // one thread at a time looks at an entry (its own mutex: only a thread that wants THIS count waits here); the slot stays ours meanwhile
    std::unique_lock<std::mutex> own(p.mu, std::defer_lock);
    if (block) own.lock(); else if (!own.try_lock()) return false;
    if (p.checked.load(std::memory_order_acquire)) return true;
    if (!wait_for_count(p, block)) return false;
    std::lock_guard<std::mutex> pl(g_pend_mu);
    if (p.checked.load(std::memory_order_relaxed)) return true;
    const uint64_t word = *(volatile uint64_t *)p.slot.host;
    p.count = word & ~(1ull << 63); p.overflow = (word >> 63) != 0; p.checked.store(true, std::memory_order_release);
    release_slot(p.slot);
    // (an overflowed forward stays findable by its own backward -- the rasterizer node raises there even if a later forward has reported it
    // already: the data IS truncated; bounded)
    if (!p.overflow) g_b_by_id.erase(p.id);
    else while (g_b_by_id.size() > 4096) g_b_by_id.erase(g_b_by_id.begin());
    return true;
}
[[noreturn]] void b_raise(BPending &p, bool earlier, size_t more = 0) {
    p.reported = true;
    TORCH_CHECK(false, "num_rendered ", p.count, " exceeds max_rendered ", p.capacity, ": results of ",
                earlier ? "an EARLIER forward of this thread (reported now: nobody had looked at its count yet) are" : "this forward are",
                " truncated", more ? " (and those of " + std::to_string(more) + " more earlier forward(s) that did not fit either)" : std::string(),
                "; raise BatchedRasterizationSettings.max_rendered (or use 0 = exact mode)");
}
// Explicit capacity: WHEN the backward looks at its forward's count.  0 = "own" (default): it waits for it -- the count arrives ~15 us into the
// forward's own kernels, so the host can never be more than one step ahead of the GPU, and with ~100 us of host work per 133-us step (C2) every
// host hiccup longer than the slack becomes a bubble on the GPU.  1 = "lazy" (set_count_wait("lazy") / SIGMAN_COUNT_WAIT=lazy; bench.py opts in):
// the backward only looks (no wait); a count that has not arrived stays pending and is waited for by the thread's forward after next -- the host
// may run two steps ahead.  An overflow is then reported one step later ("EARLIER forward"), still before anything else of that thread runs.
// "lazy:N" (N = 1..16): the newest N forwards may stay unlooked-at (the host may run N + 1 steps ahead, an overflow is reported at most N steps
// later) -- the slack that rides out a host thread that is preempted for a few hundred microseconds on a shared machine.
std::atomic<int> g_count_wait{-1};               // -1 = not read yet; 0 = own; N >= 1 = lazy, depth N
int parse_count_wait(const std::string &m) {     // -1 = not a mode
    if (m == "own") return 0;
    if (m == "lazy") return 1;
    if (m.rfind("lazy:", 0) == 0 && m.size() > 5 && m.size() <= 7 && m.find_first_not_of("0123456789", 5) == std::string::npos) {
        const int n = atoi(m.c_str() + 5);
        if (n >= 1 && n <= 16) return n;
    }
    return -1;
}
int count_wait_depth() {
    int v = g_count_wait.load();
    if (v < 0) { const char *e = getenv("SIGMAN_COUNT_WAIT"); v = e ? parse_count_wait(e) : 0; if (v < 0) v = 0; g_count_wait.store(v); }
    return v;
}
bool count_wait_lazy() { return count_wait_depth() >= 1; }

void b_poll(bool block) {            // forwards of this thread whose backward never ran (or, lazy: ran before the count had arrived)
    auto &v = b_pending();
    std::shared_ptr<BPending> bad;
    size_t keep = 0, more = 0;
    const size_t in_flight = count_wait_lazy() ? (size_t)count_wait_depth() : 128;          // newest entries that may stay unresolved
    for (size_t i = 0; i < v.size(); i++) {
        BPending &p = *v[i];
        const bool done = p.checked || b_resolve(p, block || v.size() - i > in_flight);
        if (done && p.overflow && !p.reported) {               // one error per call: it names the oldest and counts the others (none is dropped unsaid)
            if (!bad) bad = v[i];
            else { p.reported = true; more++; }
        }
        if (!done) v[keep++] = v[i];
    }
    v.resize(keep);
    if (bad) b_raise(*bad, true, more);
}

struct RasterizeL1BatchedNode : public torch::autograd::Function<RasterizeL1BatchedNode> {
    static variable_list forward(AutogradContext *ctx, Tensor means3D_, Tensor colors_, Tensor opac_, Tensor cov_, Tensor vm_, Tensor pm_, Tensor campos_,
                                 Tensor bg_, Tensor target_, Tensor mask_, int64_t H, int64_t W, double tfx, double tfy, double smod, int64_t vps, int64_t capacity_,
                                 double weight, bool da_grads, bool grad_mode) {
        TORCH_CHECK(means3D_.dim() == 3 && means3D_.size(2) == 3, "means3D must have dimensions (subjects, num_points, 3)");
        TORCH_CHECK(means3D_.is_cuda(), "sigman_release_amd rasterizer needs tensors on a ROCm device (there is no CPU fallback)");
        TORCH_CHECK(capacity_ > 0, "the C++ batched node handles the explicit sync-free mode (max_rendered > 0) only");
        const c10::Device dev = means3D_.device();
        c10::DeviceGuard guard(dev);
        const int didx = dev.index();
        const int64_t S = means3D_.size(0), P = means3D_.size(1);
        const Tensor means3D = f32c(means3D_), opac = f32c(opac_).reshape({S, P}), colors = f32c(colors_), cov = f32c(cov_);
        const Tensor vm = f32c(vm_), pm = f32c(pm_), campos = f32c(campos_), bg = f32c(bg_), target = f32c(target_);
        const int64_t nv = vm.size(0);
        TORCH_CHECK(nv == S * vps, "viewmatrix has ", nv, " views but inputs describe ", S, " subjects x ", vps, " views");
        // the loss mask of the reference (whole_loss.py:126-131: gt_masks multiplies prediction and target alike), [n_views,1,H,W]; an empty tensor = none
        const bool has_mask = mask_.defined() && mask_.numel() > 0;
        const Tensor mask = has_mask ? f32c(mask_.device() == dev ? mask_ : mask_.to(dev)) : Tensor();
        TORCH_CHECK(!has_mask || mask.numel() == nv * H * W, "mask must have ", nv * H * W, " elements ([n_views,1,H,W]), got ", mask.numel());
        TORCH_CHECK(target.numel() == nv * 3 * H * W, "target must have ", nv * 3 * H * W, " elements ([n_views,3,H,W]), got ", target.numel());
        // (under torch.no_grad() nothing will ever come back for a gradient, whatever the leaves say: the lighter forward, the count looked at now)
        // (grad_mode: torch.is_grad_enabled() of the CALLER -- inside a Function's forward it is always off)
        const bool wants_grad = grad_mode && (means3D_.requires_grad() || opac_.requires_grad() || colors_.requires_grad() || cov_.requires_grad());
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor color = at::empty({nv, 3, H, W}, f32), depth = at::empty({nv, 1, H, W}, f32), alpha = at::empty({nv, 1, H, W}, f32);
        Tensor radii = at::empty({nv, P}, f32.dtype(at::kInt));
        Tensor sums = at::empty({nv + 1}, f32), gimg = at::empty({nv, 3, H, W}, f32);      // [per-view partial sums | total]: cleared by the rasterizer's own kernels
        SgrProblem pb = make_problem(P, H, W, 0, 0, tfx, tfy, smod, means3D, opac, colors, Tensor(), cov, Tensor(), Tensor(), vm, pm, campos, bg);
        pb.n_views = (int32_t)nv; pb.views_per_subject = (int32_t)vps;
        hipStream_t stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)didx).stream();
        const int with_aux = wants_grad ? (da_grads ? 1 : 3) : 0;
        const uint64_t capacity = (uint64_t)capacity_;
        b_poll(false);
        SgrL1Epilogue ep;
        ep.target = target.data_ptr<float>(); ep.mask = has_mask ? mask.data_ptr<float>() : nullptr; ep.grad_color = gimg.data_ptr<float>(); ep.loss_per_view = sums.data_ptr<float>();
        ep.loss_total = sums.data_ptr<float>() + nv; ep.weight = (float)weight; ep.sums_already_zero = 1;
        // one or two views: loss, dL/dcolor and the compositing backward of the loss run inside the compositing kernel (st.fused_bwd tells whether
        // the launch qualified); the backward below then only gathers
        ep.fuse_backward = (wants_grad && !da_grads) ? 1 : 0; ep.reserved0 = 0;
        SgrForwardState st;
        memset(&st, 0, sizeof(st));
        AllocCtx ac;
        ac.dev = dev;
        CountSlot slot = acquire_slot(didx);
        slot.host[0] = ~0ull; slot.host[1] = 0;
        // (the fused single-view step keeps its partial records in the image blob: the calling thread's switch is part of the key, so toggling it
        // -- sgr_set_fused_step, parallel.py -- neither makes the pre-allocated attempt fail nor leaves unfused calls with the larger blob)
        const int fused_on = sgr_set_fused_step(1);
        sgr_set_fused_step(fused_on);
        const auto size_key = std::make_tuple(didx, P, nv, H, W, capacity, with_aux | (fused_on ? 256 : 0));
        std::array<uint64_t, 3> sizes{0, 0, 0};
        bool have = false;
        { std::lock_guard<std::mutex> l(g_mu); auto it = g_b_blob_sizes.find(size_key); if (it != g_b_blob_sizes.end()) { sizes = it->second; have = true; } }
        int status = 2;
        if (have) {
            for (int k = 0; k < 3; k++) ac.blob[k] = at::empty({(int64_t)sizes[k]}, f32.dtype(at::kByte));
            st.geom = ac.blob[0].data_ptr(); st.binning = ac.blob[1].data_ptr(); st.image = ac.blob[2].data_ptr();
            st.geom_bytes = sizes[0]; st.binning_bytes = sizes[1]; st.image_bytes = sizes[2];
            status = sgr_rasterize_forward_l1(&pb, capacity, with_aux, nullptr, nullptr, color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                              radii.data_ptr<int32_t>(), slot.host, slot.ev, sums.data_ptr(), (uint64_t)(nv + 1) * 4, &st, &ep, stream);
        }
        if (status == 2) {
            memset(&st, 0, sizeof(st));
            status = sgr_rasterize_forward_l1(&pb, capacity, with_aux, alloc_cb, &ac, color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                              radii.data_ptr<int32_t>(), slot.host, slot.ev, sums.data_ptr(), (uint64_t)(nv + 1) * 4, &st, &ep, stream);
            if (status == 0) {
                std::lock_guard<std::mutex> l(g_mu);
                g_b_blob_sizes[size_key] = {st.geom_bytes < 256 ? 256 : st.geom_bytes, st.binning_bytes < 256 ? 256 : st.binning_bytes,
                                            st.image_bytes < 256 ? 256 : st.image_bytes};
            }
        }
        if (status != 0) release_slot(slot);
        check_status(status, "sgr_rasterize_forward_l1");
        auto mine = std::make_shared<BPending>();
        mine->slot = slot; mine->capacity = capacity; mine->by_copy = st.nr_by_copy != 0;
        { std::lock_guard<std::mutex> pl(g_pend_mu); mine->id = g_next_id++; g_b_by_id[mine->id] = mine; }
        if (wants_grad) b_pending().push_back(mine);              // found again by the backward, or by a later forward if no backward ever runs
        else {                                                     // nobody will come back for it: look now (everything is queued, the count is published early)
            b_resolve(*mine, true);
            if (mine->overflow) b_raise(*mine, false);
        }
        ctx->set_materialize_grads(false);
        Tensor per_view = sums.narrow(0, 0, nv), loss = sums.select(0, nv);
        ctx->mark_non_differentiable({radii, per_view});
        Tensor st_bytes = at::empty({(int64_t)sizeof(SgrForwardState)}, at::TensorOptions().dtype(at::kByte));
        memcpy(st_bytes.data_ptr(), &st, sizeof(st));
        ctx->save_for_backward({means3D, opac, colors, cov, color, depth, alpha, radii, ac.blob[0], ac.blob[1], ac.blob[2], vm, pm, campos, bg, gimg});
        ctx->saved_data["st"] = st_bytes;
        ctx->saved_data["pending"] = mine->id;
        ctx->saved_data["dims"] = std::vector<int64_t>{S, P, nv, H, W, vps, opac_.dim()};
        ctx->saved_data["scal"] = std::vector<double>{tfx, tfy, smod};
        return {loss, per_view, color, radii, depth, alpha};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &means3D = saved[0], &opac = saved[1], &colors = saved[2], &cov = saved[3], &color = saved[4], &depth = saved[5], &alpha = saved[6],
                     &radii = saved[7], &vm = saved[11], &pm = saved[12], &campos = saved[13], &bg = saved[14], &gimg = saved[15];
        const auto dims = ctx->saved_data["dims"].toIntVector();
        const auto scal = ctx->saved_data["scal"].toDoubleVector();
        const int64_t S = dims[0], P = dims[1], nv = dims[2], H = dims[3], W = dims[4], vps = dims[5];
        const c10::Device dev = means3D.device();
        c10::DeviceGuard guard(dev);
        SgrForwardState st;
        memcpy(&st, ctx->saved_data["st"].toTensor().data_ptr(), sizeof(st));
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        const Tensor &g_loss = grads[0], &g_color = grads[2];
        Tensor gC, scale;
        Tensor gD = grads[4].defined() ? f32c(grads[4]) : Tensor(), gA = grads[5].defined() ? f32c(grads[5]) : Tensor();
        // a fused forward has run the compositing backward of its own loss already: with nothing but dL/dloss coming back, gather and scale
        const bool gather_only = st.fused_bwd && g_loss.defined() && !g_color.defined() && !gD.defined() && !gA.defined();
        if (!g_loss.defined() && !g_color.defined()) gC = at::zeros_like(gimg);
        else if (!g_color.defined()) { gC = gimg; scale = g_loss.reshape({1}).to(at::kFloat); }        // the common case: only the loss is used
        else gC = g_loss.defined() ? f32c(g_color) + gimg * g_loss : f32c(g_color);
        Tensor d_means3D = at::empty({S, P, 3}, f32), d_op = at::empty({S, P}, f32), d_cov = at::empty({S, P, 6}, f32), d_col = at::empty({S, P, 3}, f32);
        SgrProblem pb = make_problem(P, H, W, 0, 0, scal[0], scal[1], scal[2], means3D, opac, colors, Tensor(), cov, Tensor(), Tensor(), vm, pm, campos, bg);
        pb.n_views = (int32_t)nv; pb.views_per_subject = (int32_t)vps;
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        AllocCtx ac;
        ac.dev = dev;
        auto mp = [](const Tensor &t) -> float * { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; };
        check_status(sgr_rasterize_backward(&pb, &st, radii.data_ptr<int32_t>(), color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                            gather_only ? nullptr : gC.data_ptr<float>(), mp(gD), mp(gA), mp(scale), alloc_cb, &ac, mp(d_means3D), nullptr, mp(d_op), mp(d_col), nullptr,
                                            mp(d_cov), nullptr, nullptr, stream),
                     "sgr_rasterize_backward");
        std::shared_ptr<BPending> mine;
        {
            std::lock_guard<std::mutex> pl(g_pend_mu);
            auto it = g_b_by_id.find(ctx->saved_data["pending"].toInt());
            if (it != g_b_by_id.end()) mine = it->second;
        }
        if (mine) {                    // after the backward is queued: the host never idles the GPU while it waits for the forward's counter
            if (b_resolve(*mine, !count_wait_lazy()) && mine->overflow && !mine->reported) b_raise(*mine, false);
        }
        variable_list out = {d_means3D, d_col, dims[6] == 3 ? d_op.unsqueeze(-1) : d_op, d_cov};
        for (int k = 0; k < 16; k++) out.push_back(Tensor());
        return out;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// The batched rasterizer (rasterizer.py: rasterize_gaussians_batched) and GaussianRenderer.render (renderer.py; gs.py:49-117 of the
// reference: distCUDA2 + get_covariance per subject, every view of every subject, clamp) as ONE C++ autograd node each, for the reference's
// input flavour (colours + covariances, no SH).  Why: render() is the boundary the reference really calls (autoencoder.py:350,426), and
// through the Python nodes it cost 0.25 ms at one view where the rasterizer step itself needs 0.14 -- four autograd nodes (3-NN wrapper,
// covariance, rasterizer, clamp) issued by the interpreter.  Capacity policy (max_rendered):
//   < 0  automatic, what render() uses: the first call with a shape runs exactly and remembers 1.3x its count; later calls are sync-free, the
//        count is looked at INSIDE the call once everything is queued (it arrives right after the emission kernel), and a forward that did not
//        fit is re-run exactly before anything is returned -- the caller never sees a truncated image or an error for it;
//   > 0  the caller's explicit capacity: checked by the forward's own backward (or at once when no gradient is wanted); overflow raises;
//   = 0  exact (upstream's blocking read of num_rendered, once per batch).
struct BKeyState { uint64_t capacity = 0; };
std::map<std::tuple<int, int64_t, int64_t, int64_t, int64_t>, BKeyState> g_bkeys;          // (device, P, views, H, W) -> learned capacity
std::map<std::tuple<int, int64_t, int64_t, int64_t, int64_t, uint64_t, int>, std::array<uint64_t, 3>> g_r_blob_sizes;

struct BatchedFwd { SgrForwardState st; AllocCtx ac; std::shared_ptr<BPending> pending; };

void batched_forward(const SgrProblem &pb, const c10::Device &dev, int64_t P, int64_t nv, int64_t H, int64_t W, int with_aux, int64_t capacity_req,
                     bool wants_grad, Tensor &color, Tensor &depth, Tensor &alpha, Tensor &radii, hipStream_t stream, BatchedFwd &out) {
    const int didx = dev.index();
    const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
    const auto cap_key = std::make_tuple(didx, P, nv, H, W);
    b_poll(false);
    for (int attempt = 0;; attempt++) {
        uint64_t capacity = capacity_req > 0 ? (uint64_t)capacity_req : 0;
        // automatic mode: the remembered capacity -- but the re-run after an overflow is exact whatever another thread rendering the same
        // shape has re-learned in between (a second sync-free attempt could overflow again and surface as an error instead of a re-run)
        if (capacity_req < 0 && attempt == 0) { std::lock_guard<std::mutex> l(g_mu); capacity = g_bkeys[cap_key].capacity; }
        CountSlot slot = acquire_slot(didx);
        slot.host[0] = ~0ull; slot.host[1] = 0;
        SgrForwardState &st = out.st;
        memset(&st, 0, sizeof(st));
        out.ac = AllocCtx();
        out.ac.dev = dev;
        int status = 2;
        const auto size_key = std::make_tuple(didx, P, nv, H, W, capacity, with_aux);
        if (capacity > 0) {
            std::array<uint64_t, 3> sizes{0, 0, 0};
            bool have = false;
            { std::lock_guard<std::mutex> l(g_mu); auto it = g_r_blob_sizes.find(size_key); if (it != g_r_blob_sizes.end()) { sizes = it->second; have = true; } }
            if (have) {
                for (int k = 0; k < 3; k++) out.ac.blob[k] = at::empty({(int64_t)sizes[k]}, f32.dtype(at::kByte));
                st.geom = out.ac.blob[0].data_ptr(); st.binning = out.ac.blob[1].data_ptr(); st.image = out.ac.blob[2].data_ptr();
                st.geom_bytes = sizes[0]; st.binning_bytes = sizes[1]; st.image_bytes = sizes[2];
                status = sgr_rasterize_forward(&pb, capacity, with_aux, nullptr, nullptr, color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                               radii.data_ptr<int32_t>(), slot.host, slot.ev, nullptr, 0, &st, stream);
            }
        }
        if (status == 2) {
            memset(&st, 0, sizeof(st));
            status = sgr_rasterize_forward(&pb, capacity, with_aux, alloc_cb, &out.ac, color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                           radii.data_ptr<int32_t>(), slot.host, slot.ev, nullptr, 0, &st, stream);
            if (status == 0 && capacity > 0) {
                std::lock_guard<std::mutex> l(g_mu);
                g_r_blob_sizes[size_key] = {st.geom_bytes < 256 ? 256 : st.geom_bytes, st.binning_bytes < 256 ? 256 : st.binning_bytes,
                                            st.image_bytes < 256 ? 256 : st.image_bytes};
            }
        }
        if (status != 0) release_slot(slot);
        check_status(status, "sgr_rasterize_forward");
        if (P == 0) { release_slot(slot); return; }
        if (capacity_req > 0) {                                    // the caller's capacity: like the fused-loss node
            auto mine = std::make_shared<BPending>();
            mine->slot = slot; mine->capacity = capacity; mine->by_copy = st.nr_by_copy != 0;
            { std::lock_guard<std::mutex> pl(g_pend_mu); mine->id = g_next_id++; g_b_by_id[mine->id] = mine; }
            if (wants_grad) { b_pending().push_back(mine); out.pending = mine; }
            else { b_resolve(*mine, true); if (mine->overflow) b_raise(*mine, false); }
            return;
        }
        uint64_t count = st.true_rendered, overflow = 0;
        if (capacity > 0) {                                        // automatic, sync-free: everything is queued, the count was published early
            volatile uint64_t *w = slot.host;
            static const std::atomic<bool> never{false};
            if (!st.nr_by_copy) spin_for_count(w, never);
            if (st.nr_by_copy) SGR_TORCH_CHECK_HIP(hipEventSynchronize(slot.ev));
            else if (*w == ~0ull) SGR_TORCH_CHECK_HIP(hipDeviceSynchronize());      // (the caller's DeviceGuard is on slot.dev)
            const uint64_t word = *w;
            if (word == ~0ull) { release_slot(slot); TORCH_CHECK(false, "sigman_release_amd: the forward's instance count never arrived (device ", didx, ")"); }
            count = word & ~(1ull << 63); overflow = word >> 63;
        }
        release_slot(slot);
        if (overflow && attempt == 0) {                            // did not fit the remembered capacity: exact re-run, capacity re-learned
            std::lock_guard<std::mutex> l(g_mu);
            g_bkeys[cap_key] = BKeyState();
            continue;
        }
        TORCH_CHECK(!overflow, "num_rendered ", count, " exceeds the 32-bit instance index");      // (only an exact run gets here with the flag set)
        if (capacity_req < 0) {
            std::lock_guard<std::mutex> l(g_mu);
            BKeyState &k = g_bkeys[cap_key];
            if (capacity == 0 || count * 11 > capacity * 10) {
                const uint64_t c = count + count * 3 / 10 + 4096;
                k.capacity = c > 0xFFFFFFE0ull ? 0xFFFFFFE0ull : c;
            }
        }
        return;
    }
}

// mode 0: rasterize_gaussians_batched(means3D [S,P,3], colours [S,P,3], opacities [S,P(,1)], cov3D [S,P,6])   -> colour (unclamped), radii, depth, alpha
// mode 1: GaussianRenderer.render: (position, rgb, opacity, scale_raw [S,P,3], rotation [S,P,3,3]) -> image = clamp(colour, 0, 1), radii, depth, alpha
struct RenderBatchedNode : public torch::autograd::Function<RenderBatchedNode> {
    static variable_list forward(AutogradContext *ctx, Tensor means3D_, Tensor colors_, Tensor opac_, Tensor cov_or_scale_, Tensor rotation_, Tensor vm_, Tensor pm_,
                                 Tensor campos_, Tensor bg_, int64_t H, int64_t W, double tfx, double tfy, double smod, int64_t vps, int64_t capacity_req,
                                 bool da_grads, int64_t mode, bool grad_mode) {
        TORCH_CHECK(means3D_.dim() == 3 && means3D_.size(2) == 3, "means3D must have dimensions (subjects, num_points, 3)");
        TORCH_CHECK(means3D_.is_cuda(), "sigman_release_amd rasterizer needs tensors on a ROCm device (there is no CPU fallback)");
        const c10::Device dev = means3D_.device();
        c10::DeviceGuard guard(dev);
        const int didx = dev.index();
        const int64_t S = means3D_.size(0), P = means3D_.size(1);
        const Tensor means3D = f32c(means3D_), opac = f32c(opac_).reshape({S, P}), colors = f32c(colors_);
        const Tensor vm = f32c(vm_), pm = f32c(pm_), campos = f32c(campos_), bg = f32c(bg_);
        const int64_t nv = vm.size(0);
        TORCH_CHECK(nv == S * vps, "viewmatrix has ", nv, " views but inputs describe ", S, " subjects x ", vps, " views");
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        hipStream_t stream = c10::hip::getCurrentHIPStream((c10::DeviceIndex)didx).stream();
        Tensor cov, scale_raw, rotation, dist2;
        if (mode == 1) {
            // gs.py:70-73: distCUDA2 (detached) + get_covariance for every subject, one launch sequence
            scale_raw = f32c(cov_or_scale_); rotation = f32c(rotation_);
            TORCH_CHECK(scale_raw.numel() == S * P * 3 && rotation.numel() == S * P * 9, "scale [S,P,3] and rotation [S,P,3,3] expected");
            dist2 = at::empty({S, P}, f32);
            cov = at::empty({S, P, 6}, f32);
            if (P > 0) {
                const int64_t mc = std::min<int64_t>((1 << 22) - 1, std::max<int64_t>(16 * P, 4096));      // (renderer.py: _KNN_MAX_CELLS)
                const uint64_t stride = ((uint64_t)sgr_knn_workspace_bytes((int32_t)P, (int32_t)mc) + 255) / 256 * 256;
                Tensor ws = at::empty({(int64_t)(stride * (uint64_t)S)}, f32.dtype(at::kByte));
                check_status(sgr_knn_dist2_batched((int32_t)S, (int32_t)P, means3D.data_ptr<float>(), dist2.data_ptr<float>(), ws.data_ptr(), stride * (uint64_t)S,
                                                   (int32_t)mc, stream), "sgr_knn_dist2_batched");
                check_status(sgr_cov3d_forward((int32_t)(S * P), scale_raw.data_ptr<float>(), rotation.data_ptr<float>(), dist2.data_ptr<float>(),
                                               cov.data_ptr<float>(), stream), "sgr_cov3d_forward");
            }
        } else {
            cov = f32c(cov_or_scale_);
        }
        const bool wants_grad = grad_mode && (means3D_.requires_grad() || opac_.requires_grad() || colors_.requires_grad() || cov_or_scale_.requires_grad() ||
                                              (mode == 1 && rotation_.requires_grad()));
        Tensor color = at::empty({nv, 3, H, W}, f32), depth = at::empty({nv, 1, H, W}, f32), alpha = at::empty({nv, 1, H, W}, f32);
        Tensor radii = at::empty({nv, P}, f32.dtype(at::kInt));
        SgrProblem pb = make_problem(P, H, W, 0, 0, tfx, tfy, smod, means3D, opac, colors, Tensor(), cov, Tensor(), Tensor(), vm, pm, campos, bg);
        pb.n_views = (int32_t)nv; pb.views_per_subject = (int32_t)vps;
        const int with_aux = wants_grad ? (da_grads ? 1 : 3) : 0;
        BatchedFwd fw;
        Tensor image = color;
        if (mode == 1) {                 // gs.py:107: the compositing kernel writes clamp(colour, 0, 1) next to the unclamped colours, which stay behind for the backward
            image = at::empty_like(color);
            pb.color_clamped = image.data_ptr<float>();
        }
        batched_forward(pb, dev, P, nv, H, W, with_aux, capacity_req, wants_grad, color, depth, alpha, radii, stream, fw);
        ctx->set_materialize_grads(false);
        ctx->mark_non_differentiable({radii});
        Tensor st_bytes = at::empty({(int64_t)sizeof(SgrForwardState)}, at::TensorOptions().dtype(at::kByte));
        memcpy(st_bytes.data_ptr(), &fw.st, sizeof(fw.st));
        ctx->save_for_backward({means3D, opac, colors, cov, color, depth, alpha, radii, fw.ac.blob[0], fw.ac.blob[1], fw.ac.blob[2], vm, pm, campos, bg,
                                scale_raw, rotation, dist2});
        ctx->saved_data["st"] = st_bytes;
        if (fw.pending) ctx->saved_data["pending"] = fw.pending->id;
        ctx->saved_data["dims"] = std::vector<int64_t>{S, P, nv, H, W, vps, opac_.dim(), mode};
        ctx->saved_data["scal"] = std::vector<double>{tfx, tfy, smod};
        return {image, radii, depth, alpha};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &means3D = saved[0], &opac = saved[1], &colors = saved[2], &cov = saved[3], &color = saved[4], &depth = saved[5], &alpha = saved[6],
                     &radii = saved[7], &vm = saved[11], &pm = saved[12], &campos = saved[13], &bg = saved[14], &scale_raw = saved[15], &rotation = saved[16],
                     &dist2 = saved[17];
        const auto dims = ctx->saved_data["dims"].toIntVector();
        const auto scal = ctx->saved_data["scal"].toDoubleVector();
        const int64_t S = dims[0], P = dims[1], nv = dims[2], H = dims[3], W = dims[4], vps = dims[5], mode = dims[7];
        const c10::Device dev = means3D.device();
        c10::DeviceGuard guard(dev);
        SgrForwardState st;
        memcpy(&st, ctx->saved_data["st"].toTensor().data_ptr(), sizeof(st));
        const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
        Tensor gC = grads[0].defined() ? f32c(grads[0]) : at::zeros({nv, 3, H, W}, f32);
        Tensor gD = grads[2].defined() ? f32c(grads[2]) : Tensor(), gA = grads[3].defined() ? f32c(grads[3]) : Tensor();
        Tensor d_means3D = at::empty({S, P, 3}, f32), d_op = at::empty({S, P}, f32), d_cov = at::empty({S, P, 6}, f32), d_col = at::empty({S, P, 3}, f32);
        SgrProblem pb = make_problem(P, H, W, 0, 0, scal[0], scal[1], scal[2], means3D, opac, colors, Tensor(), cov, Tensor(), Tensor(), vm, pm, campos, bg);
        pb.n_views = (int32_t)nv; pb.views_per_subject = (int32_t)vps;
        pb.clamp_grad = mode == 1 ? 1 : 0;     // render(): the upstream gradient is w.r.t. the clamped image; the compositing backward applies clamp's mask itself
        AllocCtx ac;
        ac.dev = dev;
        auto mp = [](const Tensor &t) -> float * { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; };
        if (P > 0)
            check_status(sgr_rasterize_backward(&pb, &st, radii.data_ptr<int32_t>(), color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(),
                                                gC.data_ptr<float>(), mp(gD), mp(gA), nullptr, alloc_cb, &ac, mp(d_means3D), nullptr, mp(d_op), mp(d_col), nullptr,
                                                mp(d_cov), nullptr, nullptr, stream),
                         "sgr_rasterize_backward");
        else { d_means3D.zero_(); d_op.zero_(); d_cov.zero_(); d_col.zero_(); }
        Tensor d_fourth = d_cov, d_rot;
        if (mode == 1) {                                             // get_covariance's backward: dL/dcov3D -> dL/dscale, dL/drotation (nn_dist is detached, gs.py:71)
            d_fourth = at::empty({S, P, 3}, f32); d_rot = at::empty({S, P, 3, 3}, f32);
            if (P > 0)
                check_status(sgr_cov3d_backward((int32_t)(S * P), scale_raw.data_ptr<float>(), rotation.data_ptr<float>(), dist2.data_ptr<float>(), d_cov.data_ptr<float>(),
                                                d_fourth.data_ptr<float>(), d_rot.data_ptr<float>(), stream), "sgr_cov3d_backward");
            else { d_fourth.zero_(); d_rot.zero_(); }
        }
        if (ctx->saved_data.count("pending")) {
            std::shared_ptr<BPending> mine;
            {
                std::lock_guard<std::mutex> pl(g_pend_mu);
                auto it = g_b_by_id.find(ctx->saved_data["pending"].toInt());
                if (it != g_b_by_id.end()) mine = it->second;
            }
            if (mine) {
                if (b_resolve(*mine, !count_wait_lazy()) && mine->overflow) b_raise(*mine, false);         // (again, if a later forward has reported it already: like the Python node)
            }
        }
        variable_list out = {d_means3D, d_col, dims[6] == 3 ? d_op.unsqueeze(-1) : d_op, d_fourth, d_rot};
        for (int k = 0; k < 14; k++) out.push_back(Tensor());
        return out;
    }
};

std::vector<Tensor> rasterize_batched(Tensor means3D, Tensor colors, Tensor opac, Tensor cov, Tensor vm, Tensor pm, Tensor campos, Tensor bg, int64_t H, int64_t W,
                                      double tfx, double tfy, double smod, int64_t vps, int64_t capacity, bool da_grads) {
    // (an empty tensor in the rotation slot: autograd's apply wants every tensor argument to have a device)
    return RenderBatchedNode::apply(means3D, colors, opac, cov, at::empty({0}, means3D.options()), vm, pm, campos, bg, H, W, tfx, tfy, smod, vps, capacity, da_grads,
                                    (int64_t)0, at::GradMode::is_enabled());
}

std::vector<Tensor> render_batched(Tensor position, Tensor rgb, Tensor opacity, Tensor scale, Tensor rotation, Tensor vm, Tensor pm, Tensor campos, Tensor bg,
                                   int64_t H, int64_t W, double tfx, double tfy, double smod, int64_t vps, int64_t capacity) {
    return RenderBatchedNode::apply(position, rgb, opacity, scale, rotation, vm, pm, campos, bg, H, W, tfx, tfy, smod, vps, capacity, false, (int64_t)1,
                                    at::GradMode::is_enabled());
}

std::vector<Tensor> rasterize_l1_batched(Tensor means3D, Tensor colors, Tensor opac, Tensor cov, Tensor vm, Tensor pm, Tensor campos, Tensor bg, Tensor target,
                                         Tensor mask /* [n_views,1,H,W], or an empty tensor */, int64_t H, int64_t W, double tfx, double tfy, double smod, int64_t vps,
                                         int64_t capacity, double weight, bool da_grads) {
    return RasterizeL1BatchedNode::apply(means3D, colors, opac, cov, vm, pm, campos, bg, target, mask, H, W, tfx, tfy, smod, vps, capacity, weight, da_grads,
                                         at::GradMode::is_enabled());
}

std::vector<Tensor> rasterize_gaussians(Tensor means3D, Tensor means2D, Tensor sh, Tensor colors, Tensor opac, Tensor scales, Tensor rot, Tensor cov,
                                        int64_t H, int64_t W, double tfx, double tfy, Tensor bg, double smod, Tensor vm, Tensor pm, int64_t sh_degree,
                                        Tensor campos) {
    return RasterizeGaussiansNode::apply(means3D, means2D, sh, colors, opac, scales, rot, cov, H, W, tfx, tfy, bg, smod, vm, pm, sh_degree, campos,
                                         at::GradMode::is_enabled());
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "C++ autograd node of the single-view upstream-signature rasterizer op above the C ABI of libsigman_gsplat.so";
    m.def("rasterize_gaussians", &rasterize_gaussians, "== diff_gaussian_rasterization.rasterize_gaussians for one view (automatic sync-free capacity)");
    m.def("rasterize_l1_batched", &rasterize_l1_batched, "batched rasterizer + fused clamp/L1 loss, explicit sync-free capacity (== rasterizer.rasterize_l1_loss_batched)");
    m.def("rasterize_batched", &rasterize_batched, "batched rasterizer, colours + covariances; max_rendered < 0 automatic (inline check, exact re-run), > 0 explicit, 0 exact (== rasterizer.rasterize_gaussians_batched)");
    m.def("render_batched", &render_batched, "GaussianRenderer.render as one node: 3-NN + covariance build + batched rasterizer + clamp (== renderer.GaussianRenderer.render)");
    m.def("reset_batched", []() { std::lock_guard<std::mutex> l(g_mu); g_bkeys.clear(); }, "forget the learned batched capacities (tests)");
    m.def("batched_capacity", [](int dev, int64_t P, int64_t views, int64_t H, int64_t W) { std::lock_guard<std::mutex> l(g_mu);
                                                                                          return g_bkeys[std::make_tuple(dev, P, views, H, W)].capacity; },
          "the automatic mode's learned capacity for a shape (0: none yet)");
    m.def("check_pending_batched", []() { b_poll(true); }, "look at the counts of this thread's batched forwards whose backward never ran; raises if one was truncated");
    m.def("abi_version", []() { return sgr_abi_version(); });
    m.def("check_pending", []() { ThreadState &t = tstate(); for (auto &p : t.pending) resolve(*p, true); poll_pending(true); },
          "wait for and check the deferred instance counts of this thread's earlier forwards (raises if one overflowed)");
    m.def("reset", []() { std::lock_guard<std::mutex> l(g_mu); g_keys.clear(); }, "forget the learned capacities (tests)");
    m.def("set_count_check", [](const std::string &mode) {
              TORCH_CHECK(mode == "inline" || mode == "deferred", "set_count_check: 'inline' or 'deferred'");
              g_count_check.store(mode == "deferred" ? 1 : 0);
              if (mode == "inline") { std::lock_guard<std::mutex> l(g_mu); for (auto &kv : g_keys) kv.second.deferred = false; }
          }, "'inline' (default): the instance count is checked inside every call, a forward that does not fit is re-rendered exactly before it "
             "returns; 'deferred': the check moves behind the call once a shape's capacity has been stable (== SIGMAN_COUNT_CHECK=deferred)");
    m.def("set_count_wait", [](const std::string &mode) {
              const int d = parse_count_wait(mode);
              TORCH_CHECK(d >= 0, "set_count_wait: 'own', 'lazy' or 'lazy:N' (N = 1..16)");
              g_count_wait.store(d);
          }, "explicit-capacity batched nodes: 'own' (default): a backward waits for its forward's instance count; 'lazy': it only looks, a count that "
             "has not arrived is waited for by the thread's forward after next (the host may run two steps ahead; an overflow is reported one step later); "
             "'lazy:N': the newest N forwards may stay unlooked-at (N + 1 steps ahead, reported at most N steps later)");
    m.def("slot_stats", []() { std::lock_guard<std::mutex> l(g_slot_mu); return std::make_tuple((uint64_t)g_slots_created, (uint64_t)g_free_slots.size()); },
          "(pinned count slots ever created, idle slots in the pool)");
    m.def("key_state", [](int dev, int64_t P, int64_t H, int64_t W) { std::lock_guard<std::mutex> l(g_mu); const KeyState &k = g_keys[std::make_tuple(dev, P, H, W)];
                                                                     return std::make_tuple(k.capacity, k.max_count, k.stable, k.deferred); });
}
