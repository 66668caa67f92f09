// common.h -- shared device/host helpers for the gfx950 Gaussian-splatting kernels.
// Written for CDNA4 only: wave64, DPP cross-lane ops, 160 KB LDS.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/sigman_gsplat.h"

#define SGR_WAVE 64
// The segment-parallel forward's work order, CLASS-MAJOR form (written by the single-view path's per-tile sort, binning.hip GatherFront): words
// [0, 32) = number of occupied tiles per length class (class c: 31 - min(31, n >> 7) == c, i.e. longest lists first), then from word
// SGR_ORDER_HDR_WORDS on uint4 entries (tile, first, end, 0), class c's at [c * tiles_total, c * tiles_total + count[c]).  Empty tiles are not listed.
#define SGR_ORDER_HDR_WORDS 64
#define SGR_BIN_OCC_WORDS 2048     // binning.hip: the single-view path's tile-occupancy flags, the first words of the sort workspace

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void sgr_set_error(const char *fmt, ...);
#define SGR_CHECK_HIP(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            sgr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)
int sgr_debug_enabled();        // api.hip: upstream's debug=True (thread-local)
#define SGR_CHECK_LAUNCH(name)                                                                   \
    do {                                                                                         \
        hipError_t _e = hipGetLastError();                                                       \
        if (_e != hipSuccess) {                                                                  \
            sgr_set_error("launch of %s failed: %s", name, hipGetErrorString(_e));               \
            return 1;                                                                            \
        }                                                                                        \
        if (sgr_debug_enabled()) {                                                               \
            if (getenv("SIGMAN_TRACE_LAUNCH")) { fprintf(stderr, "[sgr] %s\n", name); fflush(stderr); }  \
            _e = hipDeviceSynchronize();                                                         \
            if (_e != hipSuccess) {                                                              \
                sgr_set_error("debug: %s failed on the device: %s", name, hipGetErrorString(_e)); \
                return 1;                                                                        \
            }                                                                                    \
        }                                                                                        \
    } while (0)

// initial value of a per-thread dev / A-B knob: the environment variable when it holds one of the accepted values, else `dflt`.  The knobs
// (sgr_set_sort_mode, sgr_set_sort_deep, sgr_set_forward_mode) are thread_local -- a setter changes the CALLING thread's flavour only --
// and every thread, whenever it is created, starts from the same environment: a profile taken through SIGMAN_SORT_MODE=... measures that
// flavour on the autograd / DataLoader threads as well, not just on the thread that loaded the library.
inline int sgr_env_knob(const char *name, int lo, int hi, int dflt) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return (v >= lo && v <= hi) ? v : dflt;
}

// profiler hooks (api.hip): slot = sgr_prof_begin(kernel id, stream); launch...; sgr_prof_end(slot, stream)
int sgr_prof_begin(int kid, hipStream_t s);
void sgr_prof_end(int slot, hipStream_t s);
struct SgrProfScope {
    int slot; hipStream_t s;
    SgrProfScope(int kid, hipStream_t st) : slot(sgr_prof_begin(kid, st)), s(st) {}
    ~SgrProfScope() { sgr_prof_end(slot, s); }
};

// the single-view fused step (render.hip: FusedL1; rasterize.hip fills it from SgrL1Epilogue + the image blob)
struct SgrFusedL1Args {
    const float *target, *mask;
    float weight;
    float *gimg, *loss_part, *loss_per_view, *loss_total;
    const uint32_t *rect;
    float *part;
    uint32_t *flags;
};

// The EMPTY tiles of a one- or two-view launch (812 of the 1024 tiles of a 512^2 humanoid view) only receive the background -- and, in the
// fused step, contribute the background's loss share and dL/dcolor.  None of that depends on the Gaussians, so in the fused step (sync-free
// mode: the image blob exists before anything is launched) EVERY tile is pre-filled with it by extra workgroups of the preprocess launch, the
// first kernel of the chain (7.4 -> 8.6 us at C2); the compositing kernel overwrites the occupied tiles and its empty-tile workgroups leave
// at once -- instead of 812 of them trickling through its tail, each living through a load round trip (C2: 3.8 us).  Measured alternatives:
// idle workgroups of the per-tile sort launch (knows the ranges, fills only the empty tiles): that launch 10.3 -> 13.7 us; extra workgroups
// of the wide row scan (32 working workgroups, 5 us): 5.1 -> 6.9 us -- the fill is the loss kernel's traffic (14 MB) and hides nowhere
// completely; without the loss the compositing kernel's own empty-tile path costs 0.2 us, so the pre-fill is only used by the fused step.
struct SgrBgJob {
    int enabled, W, H, Tx;
    uint32_t tiles_per_view, tiles_total;
    const float *bg;
    float *out_color, *out_depth, *out_alpha, *final_T;
    uint32_t *n_contrib;
    float *clamped;                 // optional (SgrProblem.color_clamped)
    const float *target, *mask;     // target == NULL: no loss
    float weight;
    float *gimg, *loss_part;
    // independent of `enabled`: a few words the FIRST kernel of the chain (preprocess) zeroes on the side for a later one -- the single-view
    // path's tile-occupancy flags (binning.hip), which the emission kernel sets with plain stores
    uint32_t *zero_ptr;
    uint32_t zero_words;
};

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// MI355X dispatches workgroup b to XCD b % 8, each XCD with a private 4 MiB L2.  Remap the linear
// block id so that CONSECUTIVE logical ids (neighbouring tiles / neighbouring key chunks, which share
// Gaussians) land on the SAME XCD.  Pure speed hint: any placement gives the same results.
__device__ __forceinline__ uint32_t sgr_xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t per = n >> 3;                 // blocks per XCD for the divisible part
    const uint32_t main = per << 3;
    if (b >= main) return b;                     // ragged tail keeps its id
    return (b & 7u) * per + (b >> 3);
}

// DPP controls (gfx9 family)
#define SGR_DPP_QUAD_XOR1 0xB1      // quad_perm:[1,0,3,2]
#define SGR_DPP_QUAD_XOR2 0x4E      // quad_perm:[2,3,0,1]
#define SGR_DPP_ROW_HALF_MIRROR 0x141
#define SGR_DPP_ROW_MIRROR 0x140
#define SGR_DPP_ROW_BCAST15 0x142
#define SGR_DPP_ROW_BCAST31 0x143
#define SGR_DPP_WAVE_SHR1 0x138
#define SGR_DPP_WAVE_ROR1 0x13C
#define SGR_DPP_ROW_SHR(n) (0x110 + (n))

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float sgr_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// wave64 sum; the total is valid in lane 63 (and, after the first four steps, every lane holds its
// 16-lane row total).  6 v_add_f32_dpp, no LDS traffic.
__device__ __forceinline__ float sgr_wave_sum_to_lane63(float v) {
    v += sgr_dpp<SGR_DPP_QUAD_XOR1>(v);
    v += sgr_dpp<SGR_DPP_QUAD_XOR2>(v);
    v += sgr_dpp<SGR_DPP_ROW_HALF_MIRROR>(v);
    v += sgr_dpp<SGR_DPP_ROW_MIRROR>(v);
    v += sgr_dpp<SGR_DPP_ROW_BCAST15, 0xA>(v);
    v += sgr_dpp<SGR_DPP_ROW_BCAST31, 0xC>(v);
    return v;
}
__device__ __forceinline__ float sgr_readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float sgr_wave_sum(float v) { return sgr_readlane(sgr_wave_sum_to_lane63(v), 63); }

__device__ __forceinline__ uint32_t sgr_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// hardware fp32 atomic add (global_atomic_add_f32 / ds_add_f32); buffers are coarse-grained hipMalloc memory
__device__ __forceinline__ void sgr_atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }


// Work order for the segment-parallel forward + clear of the bucket descriptors, executed by ONE workgroup of any size (its own
// tiny kernel in render.hip, or the spare last workgroup of the tile-sort launch in binning.hip).  tmp: 66 words of LDS.
// order = uint4 per work-order slot: (tile id, first, end of its list, 0), longest list first (32 classes by n >> 7), empty tiles last -- the
// compositing workgroup of slot k reads its tile AND its range in one load (one dependent round trip less at the start of every workgroup).
__device__ __forceinline__ void sgr_fwd_prepare(const uint2 *__restrict__ ranges, uint32_t tiles_total, uint2 *__restrict__ desc, size_t n_desc,
                                                uint32_t *__restrict__ order, uint32_t *tmp) {
    uint32_t *sHist = tmp, *sCur = tmp + 33;
    const uint32_t t = threadIdx.x, nt = blockDim.x;
    if (desc) for (size_t i = t; i < n_desc; i += nt) desc[i] = make_uint2(0u, 0u);
    if (!order) return;
    if (t < 33) sHist[t] = 0;
    __syncthreads();
    for (uint32_t tile = t; tile < tiles_total; tile += nt) {
        const uint2 r = ranges[tile];
        atomicAdd(&sHist[r.y > r.x ? 31u - min(31u, (r.y - r.x) >> 7) : 32u], 1u);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int c = 0; c < 33; c++) { sCur[c] = run; run += sHist[c]; }
    }
    __syncthreads();
    for (uint32_t tile = t; tile < tiles_total; tile += nt) {
        const uint2 r = ranges[tile];
        reinterpret_cast<uint4 *>(order)[atomicAdd(&sCur[r.y > r.x ? 31u - min(31u, (r.y - r.x) >> 7) : 32u], 1u)] = make_uint4(tile, r.x, r.y, 0u);
    }
}

// Background (and, fused step, loss share + dL/dcolor) of tile `bid`; called by whole groups of 256 threads (4 waves; a tile's 256 pixels;
// wave w's loss sum goes to the loss slot of quadrant w -- render.hip's own empty-tile path writes the same).
__device__ __forceinline__ void sgr_bg_fill_tile(const SgrBgJob &j, uint32_t bid) {
    const uint32_t t = threadIdx.x & 255u, wv = t >> 6;
    const size_t hw = (size_t)j.H * j.W;
    const uint32_t view = bid / j.tiles_per_view, tile = bid - view * j.tiles_per_view;
    const int bx = (int)(tile % (uint32_t)j.Tx) * 16 + (int)(t & 15u), by = (int)(tile / (uint32_t)j.Tx) * 16 + (int)(t >> 4);
    const bool in = bx < j.W && by < j.H;
    const size_t pix = (size_t)by * j.W + bx, vb = (size_t)view * hw;
    float lsum = 0.f;
    if (in) {
        const float b[3] = {j.bg[0], j.bg[1], j.bg[2]};
        j.final_T[vb + pix] = 1.f;
        j.n_contrib[vb + pix] = 0u;
        j.out_depth[vb + pix] = 0.f;
        j.out_alpha[vb + pix] = 0.f;
        const float m = (j.target && j.mask) ? j.mask[vb + pix] : 1.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            j.out_color[vb * 3 + (size_t)c * hw + pix] = b[c];
            if (j.clamped) j.clamped[vb * 3 + (size_t)c * hw + pix] = fminf(fmaxf(b[c], 0.f), 1.f);
            if (j.target) {
                const float d = (fminf(fmaxf(b[c], 0.f), 1.f) - j.target[vb * 3 + (size_t)c * hw + pix]) * m;
                lsum += fabsf(d);
                const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                j.gimg[vb * 3 + (size_t)c * hw + pix] = (b[c] >= 0.f && b[c] <= 1.f) ? j.weight * m * sg : 0.f;
            }
        }
    }
    if (j.target) {
        lsum = sgr_wave_sum(lsum);
        if ((t & 63u) == 0u) j.loss_part[(size_t)bid * 4 + wv] = j.weight * lsum;
    }
}
// the tiles bid = group, group + n_groups, ...
__device__ __forceinline__ void sgr_bg_fill(const SgrBgJob &j, uint32_t group, uint32_t n_groups) {
    for (uint32_t bid = group; bid < j.tiles_total; bid += n_groups) sgr_bg_fill_tile(j, bid);
}

#endif  // __HIPCC__
