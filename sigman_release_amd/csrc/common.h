// common.h -- shared device/host helpers for the gfx950 Gaussian-splatting kernels.
// Written for CDNA4 only: wave64, DPP cross-lane ops, 160 KB LDS.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sigman_gsplat.h"

#define SGR_WAVE 64

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
// order[0] = number of tiles, order[1..] = tile ids, longest list first (32 classes by n >> 7), empty tiles last.
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
        order[0] = run;                                          // == tiles_total
    }
    __syncthreads();
    for (uint32_t tile = t; tile < tiles_total; tile += nt) {
        const uint2 r = ranges[tile];
        order[1u + atomicAdd(&sCur[r.y > r.x ? 31u - min(31u, (r.y - r.x) >> 7) : 32u], 1u)] = tile;
    }
}

#endif  // __HIPCC__
