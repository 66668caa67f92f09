// tile_sort.hip -- the per-tile depth sorts behind the tile passes of binning.hip (F4, second half): every tile's list of (depth bits << 32 | value)
// composites is ordered by that composite -- a total order, equal to the published stable sort by (tile, depth) with ties in emission order
// (/root/reference/core/gaussians/gs.py:98-106 -> cub::DeviceRadixSort in the third-party rasterizer).  Three engines:
//   * tile_sort_regs_kernel   a bitonic network in registers (batches: tens of thousands of short lists),
//   * deep_tile_kernel        an O(n) distribution sort in LDS (long lists; and, FB instantiation, the whole single-view path incl. its gather front end),
//   * seg_sort_passes         stable radix passes through global memory (lists beyond every LDS capacity, lists with massive exact depth ties).
#include <string.h>
#include "binning_internal.h"

namespace {

// ---- stable LSD radix passes over ONE tile's segment, through the global ping-pong pair ------------------------------
// The generic path of the per-tile sorts below: tiles beyond the register network's 16 384 entries, and tiles the LDS distribution
// sort declines (massive exact depth ties).  Digits on which a whole segment agrees are skipped -- the exponent byte of the depths
// inside one tile almost always is.  (Rounds 1-4 also ran these passes in LDS as a sort flavour of its own -- global passes over the
// tile bits, then one workgroup per tile over the depth bits -- next to a whole-key onesweep with decoupled look-back; both were
// removed in round 5: the automatic choice reached them only for <= 256 tiles / > 2^23 instances beyond 4096 tiles per view.)

// BY_VAL: the digits are taken from the VALUE instead of the depth bits -- the first half of a sort by the (depth, value) composite
// for segments whose entries arrive in arbitrary order
template <int NT, bool BY_VAL = false>
__device__ __forceinline__ void seg_sort_passes(uint32_t n, uint64_t *gka, uint32_t *gva, uint64_t *gkb, uint32_t *gvb, bool &in_b,
                                                uint32_t *hist, uint32_t *digit_base, uint32_t (*wave_cnt)[kRadix], uint32_t *wtot) {
    constexpr int NW = NT / 64;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int shift = 0; shift < 32; shift += kRadixBits) {
        // ---- histogram of this digit over the segment
        if (t < kRadix) hist[t] = 0;
        __syncthreads();
        for (uint32_t k = t; k < n; k += NT) {
            const uint32_t key = BY_VAL ? (in_b ? gvb[k] : gva[k]) : (uint32_t)(in_b ? gkb[k] : gka[k]);
            atomicAdd(&hist[(key >> shift) & (kRadix - 1)], 1u);
        }
        __syncthreads();
        const uint32_t cnt = t < kRadix ? hist[t] : 0u;
        if (__syncthreads_or(cnt == n)) continue;                 // every key has the same digit: nothing to do
        // ---- exclusive scan of the 256 bins (threads 0..255 own one bin each)
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nb = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += nb;
        }
        if (lane == 63 && wave < 4) wtot[wave] = inc;
        __syncthreads();
        if (t < kRadix) {
            uint32_t pre = inc - cnt;
            for (uint32_t w = 0; w < wave; w++) pre += wtot[w];
            digit_base[t] = pre;
        }
        // ---- stable scatter, NT keys per round in segment order.  16-wave workgroups use a two-level cross-wave prefix (4 groups of
        // 4 waves) and touch only the counters a round used: the first version had three 16-deep LDS loops per round (clear,
        // prefix, digit-base update) -- worth 1 us of the 24 us the 2 900-entry tiles of C2 take
        constexpr bool TWO_LEVEL = NW > 4;
        constexpr int NG = TWO_LEVEL ? NW / 4 : 1;                  // groups of 4 waves (8- and 16-wave workgroups)
        static_assert(!TWO_LEVEL || NW % 4 == 0, "two-level prefix: whole groups of 4 waves");
        __shared__ uint32_t gsum[NG][kRadix];
        const uint32_t pg = t >> 8, pd = t & (kRadix - 1);           // TWO_LEVEL: my (group of 4 waves, digit) in the prefix step
        if (TWO_LEVEL) {
#pragma unroll
            for (int k = 0; k < 4; k++) wave_cnt[4 * pg + k][pd] = 0;
            __syncthreads();
        }
        for (uint32_t r0 = 0; r0 < n; r0 += NT) {
            const uint32_t k = r0 + t;
            const bool valid = k < n;
            uint32_t key = 0, val = 0;
            uint64_t key64 = 0;
            if (valid) {
                key64 = in_b ? gkb[k] : gka[k]; key = (uint32_t)key64; val = in_b ? gvb[k] : gva[k];
            }
            const uint32_t d = ((BY_VAL ? val : key) >> shift) & (kRadix - 1);
            if (!TWO_LEVEL) {
                if (t < kRadix) {
#pragma unroll
                    for (int w = 0; w < NW; w++) wave_cnt[w][t] = 0;
                }
                __syncthreads();
            }
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < kRadixBits; b++) {
                const bool bit = (d >> b) & 1;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            if (valid && rank == 0) wave_cnt[wave][d] = (uint32_t)__popcll(peers);
            __syncthreads();
            if (TWO_LEVEL) {
                const uint32_t c0 = wave_cnt[4 * pg][pd], c1 = wave_cnt[4 * pg + 1][pd], c2 = wave_cnt[4 * pg + 2][pd], c3 = wave_cnt[4 * pg + 3][pd];
                wave_cnt[4 * pg][pd] = 0; wave_cnt[4 * pg + 1][pd] = c0; wave_cnt[4 * pg + 2][pd] = c0 + c1; wave_cnt[4 * pg + 3][pd] = c0 + c1 + c2;
                gsum[pg][pd] = (c0 + c1) + (c2 + c3);
                __syncthreads();
            }
            if (valid) {
                uint32_t pos = digit_base[d] + rank;
                if (TWO_LEVEL) {
                    pos += wave_cnt[wave][d];
                    for (uint32_t g = 0; g < (wave >> 2); g++) pos += gsum[g][d];
                } else {
                    for (uint32_t w = 0; w < wave; w++) pos += wave_cnt[w][d];
                }
                (in_b ? gka : gkb)[pos] = key64; (in_b ? gva : gvb)[pos] = val;
            }
            __syncthreads();
            if (TWO_LEVEL) {
#pragma unroll
                for (int k = 0; k < 4; k++) wave_cnt[4 * pg + k][pd] = 0;
                if (pg == 0) {
                    uint32_t add = 0;
#pragma unroll
                    for (int g = 0; g < NG; g++) add += gsum[g][pd];
                    digit_base[pd] += add;
                }
            } else if (t < kRadix) {
                uint32_t add = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) add += wave_cnt[w][t];
                digit_base[t] += add;
            }
            __syncthreads();
        }
        in_b = !in_b;
        __threadfence_block(); __syncthreads();
    }
}


// sorts ONE tile's segment of (depth bits << 32 | value) composites by that composite (stable passes over the value bits first, then over
// the depth bits), src -> dst as (tile | depth) keys and values; all NT threads of the workgroup take part.
template <int NT>
__device__ __forceinline__ void sort_one_tile(const uint2 range, uint64_t *__restrict__ src_keys, uint32_t *__restrict__ src_vals,
                                              uint64_t *__restrict__ dst_keys, uint32_t *__restrict__ dst_vals, uint32_t *hist, uint32_t *digit_base,
                                              uint32_t (*wave_cnt)[kRadix], uint32_t *wtot, uint32_t tile) {
    const uint32_t t = threadIdx.x;
    const uint32_t n = range.y - range.x;
    uint64_t *gsrc_k = src_keys + range.x, *gdst_k = dst_keys + range.x;
    uint32_t *gsrc_v = src_vals + range.x, *gdst_v = dst_vals + range.x;
    bool in_b = false;
    // the scatter pass left (depth bits << 32 | value) composites in src_keys: back to (tile | depth) keys and values first
    for (uint32_t k = t; k < n; k += NT) { const uint64_t c = gsrc_k[k]; gsrc_k[k] = ((uint64_t)tile << 32) | (c >> 32); gsrc_v[k] = (uint32_t)c; }
    __threadfence_block();
    __syncthreads();
    seg_sort_passes<NT, true>(n, gsrc_k, gsrc_v, gdst_k, gdst_v, in_b, hist, digit_base, wave_cnt, wtot);
    seg_sort_passes<NT, false>(n, gsrc_k, gsrc_v, gdst_k, gdst_v, in_b, hist, digit_base, wave_cnt, wtot);
    __threadfence_block();
    __syncthreads();
    if (!in_b)
        for (uint32_t k = t; k < n; k += NT) { gdst_k[k] = gsrc_k[k]; gdst_v[k] = gsrc_v[k]; }
}

// (out of line: inlined into the single-view path's per-tile sort, the rare fallback's registers spill the main path)
template <int NT>
__device__ __attribute__((noinline)) void sort_one_tile_ool(const uint2 range, uint64_t *src_keys, uint32_t *src_vals, uint64_t *dst_keys, uint32_t *dst_vals,
                                                            uint32_t *lds, uint32_t tile) {
    uint32_t *hist = lds, *digit_base = lds + kRadix, *wtot = lds + 2 * kRadix;
    uint32_t (*wave_cnt)[kRadix] = (uint32_t (*)[kRadix])(lds + 2 * kRadix + 64);
    sort_one_tile<NT>(range, src_keys, src_vals, dst_keys, dst_vals, hist, digit_base, wave_cnt, wtot, tile);
}


// ---- per-tile depth sort IN REGISTERS ----------------------------------------------------------------------------------
// One wave per tile, the tile's entries held as 64-bit composites (depth bits << 32 | value) in IPT registers per lane.  The value
// (view * P + Gaussian index) grows with the emission order, so ordering the composites IS the stable sort by depth -- any comparison
// network will do, and a bitonic network runs entirely in VGPRs: compare-exchanges between registers of a lane for partner distances
// < IPT, and for the larger distances a lane exchange (DPP quad / row permutes, ds_swizzle, v_permlane32_swap: no LDS memory, no
// barriers, nothing to wait for but the ALU).  "Flip" formulation: every merge of two sorted halves first pairs e with e ^ (k - 1),
// then e with e ^ j for j = k/4 .. 1, so every compare-exchange puts the smaller composite at the lower index and no direction
// flags are needed.  Element e lives in lane e / IPT, register e % IPT.  Tiles shorter than 64 * IPT are padded with all-ones.
// n log^2 n compare-exchanges instead of the radix sort's 3 passes, but nothing waits on LDS round trips or workgroup barriers and no
// LDS capacity limits the number of tiles in flight (measured on MI355X, tools/micro/bench_tile_sort.hip, 12 000 tiles: 250-entry
// tiles 98 us vs 176 us for the LDS radix sort, 1000-entry tiles 224 vs 215 us, 3000 x 2000 entries 164 vs 208 us; tiles beyond the
// LDS radix sort's 4096-entry capacity -- which went through global memory, ~100 us per tile -- stay in registers up to 16384).
// Composites are built as bit patterns of POSITIVE FINITE doubles (high word = the bits of a depth > 0.2, clamped below the
// all-ones exponent; low word = the value), for which the order as doubles IS the order as 64-bit integers: a compare-exchange is one
// v_min_f64 + one v_max_f64 (full rate on CDNA4) instead of v_cmp_u64 + 4 v_cndmask + the SGPR wait states between them.
// (binning.o is compiled with -fno-honor-nans so that no canonicalising v_max_f64 x, x is put in front of every operand.)
constexpr uint32_t kCompositeHiMax = 0x7FEFFFFFu;
constexpr uint64_t kCompositePad = 0x7FEFFFFFFFFFFFFFull;
template <typename T> __device__ __forceinline__ T sgr_min_t(T x, T y) {
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(uint64_t, __builtin_fmin(__builtin_bit_cast(double, x), __builtin_bit_cast(double, y)));
    else return x < y ? x : y;
}
template <typename T> __device__ __forceinline__ T sgr_max_t(T x, T y) {
    if constexpr (sizeof(T) == 8) return __builtin_bit_cast(uint64_t, __builtin_fmax(__builtin_bit_cast(double, x), __builtin_bit_cast(double, y)));
    else return x < y ? y : x;
}

template <int M>
__device__ __forceinline__ uint32_t sgr_xlane(uint32_t v) {                       // value of lane (l ^ M), M a compile-time constant
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, true);     // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);    // row_half_mirror
    else if constexpr (M == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);   // row_mirror
    else if constexpr (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (M << 10));             // bit-mask mode: xor within 32 lanes
    else return (uint32_t)__shfl_xor((int)v, M, 64);
}

// value of lane (l ^ M) for a 32- or 64-bit composite
template <int M, typename T>
__device__ __forceinline__ T sgr_xlane_t(T v) {
    if constexpr (sizeof(T) == 8) return ((uint64_t)sgr_xlane<M>((uint32_t)(v >> 32)) << 32) | sgr_xlane<M>((uint32_t)v);
    else return sgr_xlane<M>(v);
}

// Stages whose partner sits in ANOTHER lane (or wave) keep ONE of the two composites: the holder with the partner bit clear the smaller.
// POLARITY form makes that a single instruction: while a stage is running, holders with the bit set store their composites NEGATED
// (sign bit of the double flipped).  With own' = own ^ S_me and other' = other ^ S_partner (S_partner = opposite polarity),
//     v_min_f64(own', -other')  =  min(own, other)                 in the holder with the bit clear
//                               = -max(own, other)                 in the holder with the bit set: already the polarity form of its result
// (the negation of the fetched operand is a free VOP3 source modifier).  One v_xor on the high word then moves the result into the NEXT
// stage's polarity (xhi = S_me ^ S_next, a per-lane mask computed once per stage; 0 for the in-lane stages, which need plain values).
// Per composite: 2 lane moves + v_min_f64 + v_xor instead of 2 lane moves + v_min_f64 + v_max_f64 + 2 v_cndmask.
__device__ __forceinline__ uint64_t sgr_take_lower(uint64_t own, uint64_t other, uint32_t xhi) {
    const double r = __builtin_fmin(__builtin_bit_cast(double, own), -__builtin_bit_cast(double, other));
    return __builtin_bit_cast(uint64_t, r) ^ ((uint64_t)xhi << 32);
}

// Every register index and every lane permutation must be a compile-time constant, but the network must NOT be unrolled into one
// straight line of code: a 1024-entry sort is 55 stages of ~100 instructions, executed once per tile -- as straight-line code
// (32 KB for IPT = 16, 200 KB for IPT = 64) every wave streams its instructions from L2 and the kernel is instruction-fetch bound
// (measured: 385 us for the <= 1024-entry tiles of C3 instead of 134 us).  So: ONE copy of each distinct stage body (flip K, shift J),
// selected by a switch inside rolled loops over K and J.
template <typename T, int IPT, int K>
__device__ __forceinline__ void sgr_bitonic_flip(T (&a)[IPT], uint32_t xhi) {
    if constexpr (K <= IPT) {
#pragma unroll
        for (int r = 0; r < IPT; r++) {
            const int p = r ^ (K - 1);
            if (p > r) { const T x = a[r], y = a[p]; a[r] = sgr_min_t(x, y); a[p] = sgr_max_t(x, y); }
        }
    } else if constexpr (K <= 64 * IPT) {
        // partner: lane ^ (K / IPT - 1), register IPT - 1 - r (polarity bit: the top bit of that lane mask)
        constexpr int M = K / IPT - 1;
#pragma unroll
        for (int r = 0; r < IPT / 2; r++) {                                      // registers r and IPT - 1 - r trade partners
            const int q = IPT - 1 - r;
            const T br = sgr_xlane_t<M, T>(a[q]), bq = sgr_xlane_t<M, T>(a[r]);
            a[r] = sgr_take_lower(a[r], br, xhi); a[q] = sgr_take_lower(a[q], bq, xhi);
        }
    }
}

template <typename T, int IPT, int J>
__device__ __forceinline__ void sgr_bitonic_shift(T (&a)[IPT], uint32_t xhi) {
    if constexpr (J < IPT) {
#pragma unroll
        for (int r = 0; r < IPT; r++)
            if ((r & J) == 0) { const T x = a[r], y = a[r | J]; a[r] = sgr_min_t(x, y); a[r | J] = sgr_max_t(x, y); }
    } else if constexpr (J < 64 * IPT) {
#pragma unroll
        for (int r = 0; r < IPT; r++) a[r] = sgr_take_lower(a[r], sgr_xlane_t<J / IPT, T>(a[r]), xhi);      // (polarity bit: J / IPT)
    }
}

// lane-crossing stages, selected by the LANE mask (element distance / IPT): one copy of each body behind a switch
template <typename T, int IPT>
__device__ __forceinline__ void sgr_stage_flip_lanes(T (&a)[IPT], uint32_t xhi, int lanes /* K / IPT: 2 .. 64 */) {
    switch (lanes) {
        case 2: sgr_bitonic_flip<T, IPT, 2 * IPT>(a, xhi); break;
        case 4: sgr_bitonic_flip<T, IPT, 4 * IPT>(a, xhi); break;
        case 8: sgr_bitonic_flip<T, IPT, 8 * IPT>(a, xhi); break;
        case 16: sgr_bitonic_flip<T, IPT, 16 * IPT>(a, xhi); break;
        case 32: sgr_bitonic_flip<T, IPT, 32 * IPT>(a, xhi); break;
        default: sgr_bitonic_flip<T, IPT, 64 * IPT>(a, xhi); break;
    }
}
template <typename T, int IPT>
__device__ __forceinline__ void sgr_stage_shift_lanes(T (&a)[IPT], uint32_t xhi, int lanes /* J / IPT: 1 .. 32 */) {
    switch (lanes) {
        case 1: sgr_bitonic_shift<T, IPT, IPT>(a, xhi); break;
        case 2: sgr_bitonic_shift<T, IPT, 2 * IPT>(a, xhi); break;
        case 4: sgr_bitonic_shift<T, IPT, 4 * IPT>(a, xhi); break;
        case 8: sgr_bitonic_shift<T, IPT, 8 * IPT>(a, xhi); break;
        case 16: sgr_bitonic_shift<T, IPT, 16 * IPT>(a, xhi); break;
        default: sgr_bitonic_shift<T, IPT, 32 * IPT>(a, xhi); break;
    }
}

// the stages that stay inside a lane's IPT registers, as straight-line code (two instructions per compare-exchange, and only the
// final assignment of a fused run has to land in the loop-carried registers):
//   head = levels K = 2 .. IPT (every lane sorts its own registers);  tail = the shifts IPT/2 .. 1 that end every later level
template <typename T, int IPT, int J>
__device__ __forceinline__ void sgr_tail_from(T (&a)[IPT]) {
    if constexpr (J >= 1) { sgr_bitonic_shift<T, IPT, J>(a, 0u); sgr_tail_from<T, IPT, J / 2>(a); }
}
template <typename T, int IPT, int K>
__device__ __forceinline__ void sgr_head_from(T (&a)[IPT]) {
    if constexpr (K <= IPT) { sgr_bitonic_flip<T, IPT, K>(a, 0u); sgr_tail_from<T, IPT, K / 4>(a); sgr_head_from<T, IPT, K * 2>(a); }
}

// `nw` waves (a sub-group of the workgroup, nw a RUNTIME power of two) sort nw * 64 * IPT composites: wave `sub` holds elements
// [sub * 64 * IPT, (sub + 1) * 64 * IPT) in registers; merge steps whose partner distance reaches into another wave exchange whole
// register sets through LDS (gx: [nw][IPT * 64] composites, stored register-major so that lanes hit consecutive banks): flip = partner
// wave sub ^ (K / (64 IPT) - 1), mirrored position; shift = partner wave sub ^ (j / (64 IPT)), same position; the lower wave keeps the
// smaller composite.  ONE call site per stage kind for all modes, so the code exists once whatever nw is (the instruction cache holds
// 64 KB for two CUs; three inlined copies of the IPT = 16 network made the unified kernel twice as slow as its parts).
// Every wave of the WORKGROUP executes the same number of __syncthreads() for a given nw, whatever its tile holds.
template <typename T, int IPT>
__device__ __forceinline__ void sgr_bitonic_sort_group(T (&a)[IPT], uint32_t lane, uint32_t sub, int nw, T *gx) {
    static_assert(IPT <= 16, "one wave sorts at most 1024 composites; longer tiles use several waves");
    constexpr int WAVE_ELEMS = 64 * IPT;
    T *mine = gx + sub * (uint32_t)WAVE_ELEMS;
    const int Kmax = WAVE_ELEMS * nw;
    const uint32_t pid = sub * 64u + lane;                                       // holder id inside the sub-group: bit b <-> element bit IPT * b
    sgr_head_from<T, IPT, 2>(a);
#pragma nounroll
    for (int K = 2 * IPT; K <= Kmax; K <<= 1) {
        // one merge level: the flip (partner e ^ (K - 1)), then shifts (partner e ^ j) for j = K/4, K/8, .., IPT, then the in-lane tail.
        // Polarity bit of a stage = the holder bit its partner differs in: K / IPT / 2 for the flip, j / IPT for a shift, none in the tail.
        uint32_t s_cur = (pid & (uint32_t)(K / IPT / 2)) ? 0x80000000u : 0u;
#pragma unroll
        for (int r = 0; r < IPT; r++) a[r] ^= (uint64_t)s_cur << 32;
#pragma nounroll
        for (int j = 0;;) {                                                      // j == 0: the flip
            const bool first = j == 0;
            const int jn = first ? K / 4 : j / 2;                                // the stage after this one
            const uint32_t s_nxt = (jn >= IPT && (pid & (uint32_t)(jn / IPT))) ? 0x80000000u : 0u;
            const uint32_t xhi = s_cur ^ s_nxt;
            const bool cross = first ? (K > WAVE_ELEMS) : (j >= WAVE_ELEMS);
            if (cross) {
                // ---- partner in another wave of the sub-group
                const int jw = first ? (K / WAVE_ELEMS - 1) : (j / WAVE_ELEMS);
                const T *theirs = gx + (sub ^ (uint32_t)jw) * (uint32_t)WAVE_ELEMS;
                __syncthreads();                                                 // everyone is done reading the previous exchange
#pragma unroll
                for (int r = 0; r < IPT; r++) mine[r * 64 + lane] = a[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < IPT; r++) {
                    const T b = first ? theirs[(IPT - 1 - r) * 64 + (63 - lane)] : theirs[r * 64 + lane];
                    a[r] = sgr_take_lower(a[r], b, xhi);
                }
            } else if (first) {
                sgr_stage_flip_lanes<T, IPT>(a, xhi, K / IPT);
            } else {
                sgr_stage_shift_lanes<T, IPT>(a, xhi, j / IPT);
            }
            s_cur = s_nxt;
            j = jn;
            if (j < IPT) break;
        }
        sgr_tail_from<T, IPT, IPT / 2>(a);
    }
}

// after the sort, element e sits in lane e / IPT, register e % IPT: a lane would store IPT consecutive entries (64 partial lines per
// store instruction).  One trip through LDS (row stride IPT + 1: conflict-free both ways) re-deals the elements as e = r * 64 + lane,
// so that every global store of the caller is one contiguous run across the wave.  tb: this wave's 64 * (IPT + 1) composites.
template <typename T, int IPT>
__device__ __forceinline__ void sgr_redeal_coalesced(T (&a)[IPT], uint32_t lane, T *tb) {
#pragma unroll
    for (int r = 0; r < IPT; r++) tb[lane * (IPT + 1) + r] = a[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < IPT; r++) a[r] = tb[((uint32_t)r * (64u / IPT) + lane / IPT) * (IPT + 1) + lane % IPT];
}

// gc: the tile's (depth bits << 32 | value) composites as the scatter pass left them (any order).  n == 0: nothing is read or written,
// the network runs on padding (barrier parity).  gx: the sub-group's [nw][64 * 17] composites of LDS.
template <int IPT>
__device__ __forceinline__ void sgr_sort_tile_regs64(const uint64_t *__restrict__ gc, uint64_t *__restrict__ ok, uint32_t *__restrict__ ov, uint32_t n,
                                                     uint32_t tile, uint32_t lane, uint32_t sub, int nw, uint64_t *gx) {
    uint64_t a[IPT];
    const uint32_t wbase = sub * (64u * IPT);
#pragma unroll
    for (int r = 0; r < IPT; r++) {                                            // coalesced; the input order is irrelevant to the result
        const uint32_t k = wbase + (uint32_t)r * 64u + lane;
        const uint64_t c = k < n ? gc[k] : kCompositePad;
        a[r] = ((uint64_t)min((uint32_t)(c >> 32), kCompositeHiMax) << 32) | (uint32_t)c;
    }
    sgr_bitonic_sort_group<uint64_t, IPT>(a, lane, sub, nw, gx);
    if (nw > 1) __syncthreads();                                               // the other waves are done with the last exchange
    sgr_redeal_coalesced<uint64_t, IPT>(a, lane, gx + sub * (64u * 17u));
#pragma unroll
    for (int r = 0; r < IPT; r++) {
        const uint32_t e = wbase + (uint32_t)r * 64u + lane;
        if (e < n) { if (ok) ok[e] = ((uint64_t)tile << 32) | (a[r] >> 32); ov[e] = (uint32_t)a[r]; }
    }
}

// ONE launch for every tile of <= NW * 1024 entries: a fixed grid of NW-wave workgroups drains the worklists class by class, longest
// tiles first: class m (tiles of <= 1024 << m entries) is sorted by sub-groups of 2^m waves, NW >> m tiles per workgroup at a time.
// (Separate launches per class cost a ramp-up and a tail each -- with four classes that was more than the sorting itself at C3.)
// Tickets are drawn for a workgroup's worth of tiles at a time (and four rounds' worth for the single-wave class): returning atomics
// on one address complete one every ~9 ns on this part, so one ticket per tile made the 41 000 short tiles of C4 a 0.37 ms serial section.

template <int NW>
__global__ __launch_bounds__(64 * NW) void tile_sort_regs_kernel(const uint2 *__restrict__ ranges, uint64_t *__restrict__ src_comp,
                                                                 uint32_t *__restrict__ src_scratch, uint64_t *__restrict__ dst_keys,
                                                                 uint32_t *__restrict__ dst_vals, TileWork4 tw, int m_hi, int m_lo, int keep_keys,
                                                                 uint32_t *__restrict__ prep_order, uint32_t prep_tiles) {
    // keep_keys == 0: only the point list is stored (the sorted keys have no reader behind the per-tile sort: the ranges come from the
    // tile pass); the global-memory fallback for oversize tiles writes both regardless
    __shared__ uint64_t xbuf[NW * 64 * 17];                                      // per wave 64 x (16 + 1) composites: exchange + final re-deal
    __shared__ uint32_t s_item, s_next;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t nsort = gridDim.x;                                                  // workgroups that sort
    if (prep_order) {                                                            // the spare last workgroup orders the tiles for the segment-parallel forward
        nsort = gridDim.x - 1;
        if (blockIdx.x == nsort) { sgr_fwd_prepare(ranges, prep_tiles, nullptr, 0, prep_order, (uint32_t *)xbuf); return; }
    }
    // all class sizes up front (independent loads: one memory latency instead of one per class on the single-view critical path)
    uint32_t cnt[6];
#pragma unroll
    for (int m = 0; m < 6; m++) cnt[m] = *tw.w[m].count;
    if (NW == 16 && m_hi >= 4 && cnt[5]) {
        // ---- tiles beyond 16 384 entries (none in the bench configs): a whole workgroup per tile, LDS-free radix passes over the value
        // bits and then the depth bits through the global pair (src_comp / src_scratch <-> dst); first: they are the longest jobs
        const TileWork w = tw.w[5];
        const uint32_t nwork = cnt[5];
        uint32_t *l32 = (uint32_t *)xbuf;
        uint32_t *hist = l32, *digit_base = l32 + kRadix, *wtot = l32 + 2 * kRadix;
        uint32_t (*wave_cnt)[kRadix] = (uint32_t (*)[kRadix])(l32 + 2 * kRadix + 64);
        for (;;) {
            __syncthreads();
            if (threadIdx.x == 0) s_item = atomicAdd(w.ticket, 1u);
            __syncthreads();
            const uint32_t wi = s_item;
            if (wi >= nwork) break;
            const uint32_t tile = w.list[wi];
            const uint2 range = ranges[tile];
            if (range.y > range.x)
                sort_one_tile<64 * NW>(range, src_comp, src_scratch, dst_keys, dst_vals, hist, digit_base,
                                                wave_cnt, wtot, tile);
        }
        __syncthreads();
    }
    // Every class deals its FIRST round statically (workgroup -> slot, no atomics) and draws tickets only for what is left, so a launch
    // with fewer tiles than workgroups (one view) never waits for an atomic round trip.  The slots of a class are rotated by the number
    // of tiles in the longer classes: the workgroups that just sorted a longer tile get the last slots (beyond the list for one view).
    uint32_t longer = (NW == 16 && m_hi >= 4) ? cnt[5] : 0u;
#pragma nounroll
    for (int m = m_hi; m >= 1 && m >= m_lo; m--) {
        // ---- tiles of 1024 << (m - 1) < n <= 1024 << m entries: sub-groups of 2^m waves in lock step (the network has workgroup barriers)
        const int nw = 1 << m;
        const uint32_t groups = (uint32_t)(NW >> m), grp = wave >> m, sub = wave & (uint32_t)(nw - 1);
        const TileWork w = tw.w[m];
        const uint32_t nwork = cnt[m];
        if (nwork == 0u) continue;                                               // workgroup-uniform
        uint64_t *gx = xbuf + grp * (uint32_t)(nw * 64 * 17);
        // workgroup barriers a sub-group executes per tile: two per wave-crossing stage (level l of the log2(nw) upper levels has l of
        // them) + the one before the re-deal; sub-groups without a tile only keep that count (they must not compete for the SIMDs)
        const int n_barriers = m * (m + 1) + 1;
        // few tiles (one view): spread them over the workgroups, one sub-group each, instead of filling every sub-group of a few
        const uint32_t take0 = min(groups, max(1u, (nwork + nsort - 1u) / nsort)), dealt = nsort * take0;
        const uint32_t rot = longer % nsort;
        uint32_t take = take0, base = (blockIdx.x >= rot ? blockIdx.x - rot : blockIdx.x + nsort - rot) * take0;
        longer += nwork;
        for (;;) {
            if (base >= nwork) break;                                            // workgroup-uniform
            const uint32_t wi = base + grp;
            if (grp < take && wi < nwork) {
                const uint32_t tile = w.list[wi];
                const uint2 range = ranges[tile];
                sgr_sort_tile_regs64<16>(src_comp + range.x, keep_keys ? dst_keys + range.x : nullptr, dst_vals + range.x, range.y - range.x, tile, lane, sub, nw, gx);
            } else {
#pragma nounroll
                for (int b = 0; b < n_barriers; b++) __syncthreads();
            }
            if (dealt >= nwork) break;                                           // the static round covered the class
            const uint32_t left = nwork - min(nwork, base + take);
            take = min(groups, max(1u, (left + nsort - 1u) / nsort));
            __syncthreads();
            if (threadIdx.x == 0) s_item = dealt + atomicAdd(w.ticket, take);
            __syncthreads();
            base = s_item;
        }
    }
    if (m_lo > 0) return;
    // ---- tiles of <= 1024 entries: one wave each, no workgroup barriers inside the sort.  The workgroup takes BATCHES of tiles (the first
    // one statically, then one returning global atomic per batch: they complete one every ~9 ns on one address -- a ticket per tile was
    // a 0.37-ms serial section for the 41 000 short tiles of C4), its waves draw single tiles from the batch through an LDS counter, so
    // a wave with short tiles takes more of them; batches shrink towards the end of the list (guided self-scheduling): short tail.
    const TileWork w = tw.w[0];
    const uint32_t nwork = cnt[0];
    if (nwork == 0u) return;
    uint64_t *gx = xbuf + wave * (uint32_t)(64 * 17);
    const uint32_t want0 = min((uint32_t)(4 * NW), max(1u, (nwork + 2u * nsort - 1u) / (2u * nsort))), dealt = nsort * want0;
    const uint32_t rot = longer % nsort;
    uint32_t want = want0, base = (blockIdx.x >= rot ? blockIdx.x - rot : blockIdx.x + nsort - rot) * want0;
    __syncthreads();                                                             // (s_next: the classes above are done with LDS)
    if (threadIdx.x == 0) s_next = 0u;
    __syncthreads();
    for (;;) {
        if (base >= nwork) break;                                                // workgroup-uniform
        const uint32_t cntb = min(want, nwork - base);
        for (;;) {
            uint32_t i = 0;
            if (lane == 0) i = atomicAdd(&s_next, 1u);
            i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
            if (i >= cntb) break;
            const uint32_t tile = w.list[base + i];
            const uint2 range = ranges[tile];
            const uint32_t n = range.y - range.x;
            if (n == 0u) continue;
            uint64_t *okp = keep_keys ? dst_keys + range.x : nullptr;
            if (n <= 256u) sgr_sort_tile_regs64<4>(src_comp + range.x, okp, dst_vals + range.x, n, tile, lane, 0u, 1, gx);
            else if (n <= 512u) sgr_sort_tile_regs64<8>(src_comp + range.x, okp, dst_vals + range.x, n, tile, lane, 0u, 1, gx);
            else sgr_sort_tile_regs64<16>(src_comp + range.x, okp, dst_vals + range.x, n, tile, lane, 0u, 1, gx);
        }
        if (dealt >= nwork) break;                                               // the static round covered the class
        const uint32_t left = nwork - min(nwork, base + want);
        want = min((uint32_t)(4 * NW), max(1u, left / (2u * nsort)));
        __syncthreads();                                                         // every wave is done with the previous batch
        if (threadIdx.x == 0) { s_item = dealt + atomicAdd(w.ticket, want); s_next = 0u; }
        __syncthreads();
        base = s_item;
    }
}

// ---- deep tiles: an O(n) bucket sort of a whole tile in LDS, one workgroup per tile (window) ------------------------------------------
// The composites of a tile are (depth bits << 32 | value) with depths of ONE 16 x 16-pixel tile: a few surfaces, i.e. a smooth density
// over a narrow range.  So instead of a comparison network (n log^2 n compare-exchanges: a 16 384-entry tile keeps sixteen waves busy
// for ~80 us) the tile is sorted by DISTRIBUTION, entirely in LDS:
//   1. range [lo, hi] of the tile's depth bits, a coarse 256-bin histogram over it (every 4th composite: a density estimate);
//   2. the fine bins (4096) are dealt to the coarse bins in proportion to their share of the samples, so the fine bins are narrow where
//      the tile is dense (a depth outlier that stretches the range, or thin surfaces, cost resolution only where nothing is);
//      fine bin = monotone function of the depth bits (fp32 arithmetic, monotone by construction);
//   3. fine histogram, exclusive scan, every composite stored at its bin's cursor in LDS (one returning LDS atomic): ordered by bin;
//   4. a bin holds a handful of composites: each counts the smaller ones of its own bin (broadcast reads of neighbouring LDS words) and
//      that rank is its final place -- the point list (and keys) leave in nearly contiguous runs.
// Exactly the order of every other flavour: bins are ordered by depth, ranks by the full composite, composites are unique.
// Tiles of more than CAP - 128 entries are walked in windows of whole bins (re-reading the segment, L2-warm).  A tile in which some fine
// bin holds more than 128 composites (massive exact depth ties) goes to the generic per-tile sort's worklists instead.
#ifdef SGR_DEEP_TIMING          /* tools/micro/bench_tile_sort.hip: phase stamps of the first tile of every workgroup (100 MHz clock) */
__device__ unsigned long long sgr_deep_dbg[1024 * 16];
#define SGR_STAMP(P) if (t == 0 && i == blockIdx.x && blockIdx.x < 1024u) sgr_deep_dbg[blockIdx.x * 16 + (P)] = __builtin_amdgcn_s_memrealtime();
extern "C" int sgr_debug_deep_stamps(unsigned long long *host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(sgr_deep_dbg), sizeof(sgr_deep_dbg)) == hipSuccess ? 0 : 1; }
#else
#define SGR_STAMP(P)
#endif
// worklist entry of the deep kernels (FB = false): tile id | window << 26 (one workgroup per WINDOW of a tile: a tile of more than CAP - 128
// entries is shared by several workgroups, each of which builds the tile's histogram for itself and then places / ranks its own window); a tile
// the distribution sort declines goes to the register sort's worklists (lists / plan).
// FB = true (the single-view path: one or two views, <= 2048 tiles): ONE WORKGROUP PER TILE and no tile pass in front of it.  The workgroup
// reads its tile's columns of the run matrix the emission kernel left (GatherFront, kRunRow): the pieces of its list in the emission
// workgroups' tile-ordered runs and -- summed -- its range in the sorted list; it writes that range (F5), enters the tile into the compositing
// kernel's work order (a class-major list: one returning atomic per OCCUPIED tile on one of 32 counters), gathers its composites straight into
// registers and sorts them.  An empty tile's workgroup writes the tile's background on the spot (gf.bg), so the compositing kernel never looks at
// empty tiles.  A tile beyond one window (> CAP - 128 entries) is first copied to a contiguous scratch segment and its windows are then
// walked one after the other by this workgroup; a tile the distribution sort declines (massive depth ties) or with more windows than
// gf.max_windows is sorted by the stable radix passes through global memory.
template <int NT, int CAP, int NBF, bool FB = false>
__global__ __launch_bounds__(NT, (FB && NT == 512) ? 4 : 1) void deep_tile_kernel(uint64_t *comp, uint32_t *scratch, uint64_t *__restrict__ dst_keys, uint32_t *__restrict__ dst_vals,
                                                       const uint32_t *__restrict__ count_ptr, uint32_t *deep_list,
                                                       const uint2 *__restrict__ ranges, int keep_keys, VsegPlan *__restrict__ plan,
                                                       uint32_t *__restrict__ lists, uint32_t list_stride, GatherFront gf) {
    constexpr uint32_t RI = (CAP + NT - 1) / NT <= 4 ? 4 : ((CAP + NT - 1) / NT <= 8 ? 8 : 16), REG = NT * RI;   // the first REG composites of a tile live in registers (RI per thread) for all passes
    constexpr uint32_t ITEMS = 8, ROUND = NT * ITEMS;     // the rest (tiles beyond REG entries) is re-read from the segment in every pass
    constexpr uint32_t NW = NT / 64, NBC = 256, WIN = CAP - kDeepBinMax, PER = NBF / NT;
    static_assert(NBF % NT == 0 && NT >= (int)NBC && NT % 64 == 0 && (NBF & (NBF - 1)) == 0, "layout");
    static_assert(!FB || (NBF >= 513 && NT >= 512 && WIN <= REG), "FB: the piece tables of <= 512 runs live in s_pre / s_cur; a one-window tile fits the registers");
    __shared__ uint64_t s_comp[CAP];
    __shared__ uint32_t s_pre[NBF], s_cur[NBF];
    __shared__ uint32_t s_ccnt[NBC], s_fstart[NBC], s_fcnt[NBC];
    __shared__ uint32_t s_wave[NW], s_wave2[NW], s_lo, s_hi, s_bad;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t nsort = gridDim.x;
    const uint32_t ndeep = FB ? gf.tiles_total : *count_ptr;
#define SGR_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
    for (uint32_t i = blockIdx.x; i < ndeep; i += nsort) {
        __syncthreads();                                                        // (LDS reuse between tiles)
        SGR_STAMP(0)
        uint32_t tile, w0;
        uint2 range;
        uint64_t c[RI];
        uint32_t rest_lo = 0xFFFFFFFFu, rest_hi = 0u;                           // FB: depth range of the composites beyond the registers
        const uint64_t *seg;                                                    // the tile's composites in one piece (FB: only for tiles beyond one window)
        if constexpr (FB) {
            // ---- my columns of the run matrix: piece of run b = [rows[b][tile], rows[b][tile + 1]); sum_b (rows[b][tile] - base[b]) instances
            // sit in tiles before mine
            // workgroup i <-> tile i.  An empty tile's workgroup (three of four at a humanoid view) leaves after one load: its range, its background
            w0 = 0u;
            tile = sgr_tile_of_workgroup(i, gf.tx, gf.ty);                        // (XCD-local columns, engines on diagonals, centre first: binning_internal.h)
            if (SGR_UNIFORM(gf.occ[tile]) == 0u) {
                if (t == 0) gf.ranges[tile] = make_uint2(0u, 0u);
                if (gf.bg.enabled && t < 256u) sgr_bg_fill_tile(gf.bg, tile);
                continue;
            }
            SGR_STAMP(8)
            uint32_t a = 0u, e = 0u, bs = 0u;
            if (t < gf.nblk) { const uint32_t *r = gf.rows + (size_t)t * kRunRow + tile; a = r[0]; e = r[1]; bs = gf.base[t]; }
#ifdef SGR_DEBUG_BOUNDS
            if (t < gf.nblk && (a < bs || e < a || e > gf.cap_dbg)) printf("bad piece tile %u run %u a %u e %u bs %u cap %u from %p\n", tile, t, a, e, bs, gf.cap_dbg, (const void *)(gf.base + t));
#endif
            const uint32_t cnt = e - a;
            uint32_t inc = cnt, before = a - bs;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t nbv = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += nbv; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
            if (lane == 63u) s_wave[wave] = inc;
            if (lane == 0u) s_wave2[wave] = before;
            __syncthreads();
            uint32_t pre = 0u, n_all = 0u, first = 0u;
#pragma unroll
            for (uint32_t w = 0; w < NW; w++) { const uint32_t x = s_wave[w]; if (w < wave) pre += x; n_all += x; first += s_wave2[w]; }
            s_pre[t] = pre + inc - cnt; s_cur[t] = a;                            // piece prefix / piece start (runs beyond nblk: prefix = n, never chosen)
            static_assert(NT >= 512 && NT <= NBF, "one run per thread (<= NT emission runs); the piece tables live in s_pre / s_cur");
            const uint32_t n_t = SGR_UNIFORM(n_all), first_t = SGR_UNIFORM(first);
            range = make_uint2(first_t, first_t + n_t);
            if (t == 0) {
                gf.ranges[tile] = n_t ? range : make_uint2(0u, 0u);
                if (gf.order && n_t) {
                    const uint32_t cls = 31u - min(31u, n_t >> 7);
                    gf.order[(size_t)cls * gf.tiles_total + atomicAdd(&gf.cls_count[cls], 1u)] = make_uint4(tile, range.x, range.y, 0u);
                }
            }
            if (n_t == 0u) continue;                                            // (cannot happen: the tile is marked occupied)
            __syncthreads();
            SGR_STAMP(9)
#ifdef SGR_DEEP_TIMING
            if (t == 0 && blockIdx.x < 1024u) sgr_deep_dbg[blockIdx.x * 16 + 15] = n_t;
#endif
            // composite j of the tile: in the piece of the last run whose prefix is <= j.  K composites at a time, their binary searches in
            // lock-step (GK independent LDS reads per level instead of GK x levels dependent LDS round trips)
            constexpr int GK = 8;
            static_assert(RI % GK == 0, "register composites are gathered GK at a time");
            auto gather_k = [&](const uint32_t (&j)[GK], uint64_t (&out)[GK]) {
                uint32_t lo[GK];
#pragma unroll
                for (int k = 0; k < GK; k++) lo[k] = 0u;
                for (uint32_t step = gf.search_top; step > 0u; step >>= 1) {
#pragma unroll
                    for (int k = 0; k < GK; k++) if (s_pre[lo[k] + step] <= j[k]) lo[k] += step;
                }
#pragma unroll
                for (int k = 0; k < GK; k++) {
#ifdef SGR_DEBUG_BOUNDS
                    if (s_cur[lo[k]] + (j[k] - s_pre[lo[k]]) >= gf.cap_dbg) { printf("gather oob tile %u j %u lo %u pre %u cur %u n %u nblk %u\n", tile, j[k], lo[k], s_pre[lo[k]], s_cur[lo[k]], n_t, gf.nblk); out[k] = 0; continue; }
#endif
                    out[k] = comp[s_cur[lo[k]] + (j[k] - s_pre[lo[k]])];
                }
            };
            const uint32_t last_j = n_t - 1u;
#pragma unroll
            for (uint32_t it0 = 0; it0 < RI; it0 += GK) {
                if (it0 == 0u || it0 * NT < n_t) {                               // (uniform)
                    uint32_t j[GK];
                    uint64_t v[GK];
#pragma unroll
                    for (int k = 0; k < GK; k++) j[k] = min((it0 + k) * NT + t, last_j);
                    gather_k(j, v);
#pragma unroll
                    for (int k = 0; k < GK; k++) c[it0 + k] = v[k];
                } else {
                    // registers wholly beyond the list: any entry of the list will do (they only meet the range's min / max; every other use
                    // checks the index)
#pragma unroll
                    for (int k = 0; k < GK; k++) c[it0 + k] = c[0];
                }
            }
            if (n_t <= WIN) {
                seg = nullptr;                                                   // (n <= WIN <= REG: nothing below reads it)
            } else {
                // a tile beyond one window: the composites beyond the registers are re-read by every pass -- they go to the tile's stretch of the
                // scratch buffer (their depth range is taken on the way: one pass over them less; GK at a time: one at a time a long tile spent 1.7 us per trip here)
                uint64_t *sk = gf.scratch_k + range.x;
                for (uint32_t j0 = REG + t; j0 < n_t; j0 += (uint32_t)GK * NT) {
                    uint32_t j[GK];
                    uint64_t v[GK];
#pragma unroll
                    for (int k = 0; k < GK; k++) j[k] = min(j0 + (uint32_t)k * NT, last_j);
                    gather_k(j, v);
#pragma unroll
                    for (uint32_t u = 0; u < (uint32_t)GK; u++) {
                        const uint32_t z = (uint32_t)(v[u] >> 32);
                        rest_lo = min(rest_lo, z); rest_hi = max(rest_hi, z);       // (an index past the end repeats the last entry)
                        if (j0 + u * NT < n_t) sk[j0 + u * NT] = v[u];
                    }
                }
                __threadfence_block();
                seg = sk;
            }
            __syncthreads();                                                    // (s_pre / s_cur change roles below)
            SGR_STAMP(10)
        } else {
            const uint32_t entry = SGR_UNIFORM(deep_list[i]);
            tile = entry & kDeepTileMask; w0 = (entry >> 26) * WIN;              // this workgroup's window: sorted positions of bins starting in [w0, w0 + WIN)
            const uint2 range_v = ranges[tile];
            range = make_uint2(SGR_UNIFORM(range_v.x), SGR_UNIFORM(range_v.y));
            seg = comp + range.x;
            const uint32_t last_j = range.y - range.x - 1u;
#pragma unroll
            for (uint32_t it = 0; it < RI; it++) c[it] = seg[min(it * NT + t, last_j)];
        }
        const uint32_t n = range.y - range.x, last = n - 1u;
        // FB: stable LSD passes over the value bits, then the depth bits, through global memory (scratch_k / scratch <-> dst): the whole tile, by
        // this workgroup alone.  The composites must sit in scratch_k: the first REG of them are in registers only
        auto sort_here = [&]() {
#pragma unroll
            for (uint32_t it = 0; it < RI; it++) if (it * NT + t < n) gf.scratch_k[range.x + it * NT + t] = c[it];
            __threadfence_block();
            __syncthreads();
            sort_one_tile_ool<NT>(range, gf.scratch_k, scratch, dst_keys, dst_vals, (uint32_t *)s_comp, tile);
        };
        if constexpr (FB) {
            // tiny tiles (the rim of a subject's silhouette): every composite counts the smaller
            // ones -- n broadcast reads of LDS -- instead of the distribution sort's passes over its bin tables (6 us whatever the tile holds)
            if (n <= kDeepTiny) {
                if (t < n) s_comp[t] = c[0];
                __syncthreads();
                if (t < n) {
                    const uint64_t v = c[0];
                    uint32_t rank = 0;
                    for (uint32_t k = 0; k < n; k += 4u) {
                        const uint64_t x0 = s_comp[k], x1 = s_comp[min(k + 1u, last)], x2 = s_comp[min(k + 2u, last)], x3 = s_comp[min(k + 3u, last)];
                        rank += (x0 < v ? 1u : 0u) + ((k + 1u < n && x1 < v) ? 1u : 0u) + ((k + 2u < n && x2 < v) ? 1u : 0u) + ((k + 3u < n && x3 < v) ? 1u : 0u);
                    }
                    if (keep_keys) dst_keys[range.x + rank] = ((uint64_t)tile << 32) | (v >> 32);
                    dst_vals[range.x + rank] = (uint32_t)v;
                }
                SGR_STAMP(7)
                continue;
            }
        }
        if constexpr (FB) { if ((n + WIN - 1u) / WIN > gf.max_windows) { sort_here(); continue; } }
        // f(composite) for the composites beyond the registers
        auto for_each_rest = [&](auto f) {
            for (uint32_t r0 = REG; r0 < n; r0 += ROUND) {
                uint64_t d[ITEMS];
#pragma unroll
                for (uint32_t it = 0; it < ITEMS; it++) d[it] = seg[min(r0 + it * NT + t, last)];
#pragma unroll
                for (uint32_t it = 0; it < ITEMS; it++) if (r0 + it * NT + t < n) f(d[it]);
            }
        };
        // ---- 1. depth range, coarse histogram
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
        for (uint32_t it = 0; it < RI; it++) { const uint32_t z = (uint32_t)(c[it] >> 32); lo = min(lo, z); hi = max(hi, z); }      // (indices past the end repeat the last entry)
        if constexpr (FB) { lo = min(lo, rest_lo); hi = max(hi, rest_hi); }
        else for_each_rest([&](uint64_t v) { const uint32_t z = (uint32_t)(v >> 32); lo = min(lo, z); hi = max(hi, z); });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64)); }
        if (lane == 0) { s_wave[wave] = lo; s_wave2[wave] = hi; }
        if (t < NBC) s_ccnt[t] = 0u;
        __syncthreads();
        if (t == 0) {
            uint32_t a = s_wave[0], z = s_wave2[0];
            for (uint32_t w = 1; w < NW; w++) { a = min(a, s_wave[w]); z = max(z, s_wave2[w]); }
            s_lo = a; s_hi = z; s_bad = 0u;
        }
        __syncthreads();
        SGR_STAMP(1)
        lo = SGR_UNIFORM(s_lo);
        // coarse position of depth bits z: (z - lo) * 256 / (range + 1) in fp32 -- monotone in z; its integer part is the coarse bin
        const float scc = __builtin_bit_cast(float, SGR_UNIFORM(__builtin_bit_cast(uint32_t, (float)NBC / ((float)(SGR_UNIFORM(s_hi) - lo) + 1.0f))));
        // a density estimate is all the coarse histogram is: every 4th composite of the LIST (register composite j = it * NT + t with j % 4 == t % 4:
        // the threads with t % 4 == 0 enter all of theirs).  Not "every 4th register": the list arrives run by run, and with Gaussians in a
        // spatially coherent order its first NT entries are one patch of one surface -- the bins were dealt by that patch's depths, other
        // depths overflowed their bins and every long tile was declined (C2 in template order: per-tile sort 20 -> 73 us)
        uint32_t n_samples = 0;
        if ((t & 3u) == 0u) {
#pragma unroll
            for (uint32_t it = 0; it < RI; it++)
                if (it * NT + t < n) atomicAdd(&s_ccnt[min(NBC - 1u, (uint32_t)((float)((uint32_t)(c[it] >> 32) - lo) * scc))], 1u);
        }
        for (uint32_t k = REG + t * 4u; k < n; k += NT * 4u) atomicAdd(&s_ccnt[min(NBC - 1u, (uint32_t)((float)((uint32_t)(seg[k] >> 32) - lo) * scc))], 1u);
        n_samples = (min(n, REG) + 3u) / 4u + (n > REG ? (n - REG + 3u) / 4u : 0u);      // (uniform arithmetic; REG % 4 == 0)
        __syncthreads();
        SGR_STAMP(2)
        // ---- 2. fine bins per coarse bin: one for every occupied coarse bin + the rest in proportion to the samples
        if (t < NBC) {
            const uint32_t cnt = s_ccnt[t];
            const uint32_t fc = cnt ? 1u + (uint32_t)(((uint64_t)cnt * (uint64_t)(NBF - NBC)) / n_samples) : 0u;
            uint32_t inc = fc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t nbv = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += nbv; }
            if (lane == 63u) s_wave[wave] = inc;
            s_fcnt[t] = fc; s_fstart[t] = inc - fc;
        }
        __syncthreads();
        if (t < NBC) { uint32_t add = 0; for (uint32_t w = 0; w < wave; w++) add += s_wave[w]; s_fstart[t] += add; }
        for (uint32_t b = t; b < (uint32_t)NBF; b += NT) s_cur[b] = 0u;
        __syncthreads();
        SGR_STAMP(3)
        // fine bin of a composite: coarse bin cb + the fraction of the way through it, scaled to cb's share of the fine bins (a coarse bin
        // that the sampling missed has no fine bins of its own: its composites join the last bin of the nearest occupied one below)
        auto fine_bin = [&](uint64_t v) -> uint32_t {
            const float cf = (float)((uint32_t)(v >> 32) - lo) * scc;
            const uint32_t cb = min(NBC - 1u, (uint32_t)cf);
            const uint32_t fs = s_fstart[cb], fn = s_fcnt[cb];
            const float fr = cf - (float)cb;
            return fn ? fs + min(fn - 1u, (uint32_t)(fr * (float)fn)) : (fs ? fs - 1u : 0u);
        };
        // ---- 3. fine histogram.  The bins of the register composites are kept (2 x 16 bits per register); all table reads of a stage are
        // issued before the first atomic of the next (LDS operations complete in order: read, atomic, read, atomic.. would wait 16 times)
        uint32_t fbr[RI / 2];
        {
            uint32_t fb[RI];
#pragma unroll
            for (uint32_t it = 0; it < RI; it++) fb[it] = fine_bin(c[it]);
#pragma unroll
            for (uint32_t it = 0; it < RI; it += 2) fbr[it / 2] = fb[it] | (fb[it + 1] << 16);
#pragma unroll
            for (uint32_t it = 0; it < RI; it++) if (it * NT + t < n) atomicAdd(&s_cur[fb[it]], 1u);
        }
        for_each_rest([&](uint64_t v) { atomicAdd(&s_cur[fine_bin(v)], 1u); });
        __syncthreads();
        SGR_STAMP(4)
        {   // exclusive scan (PER consecutive bins per thread) -> s_pre; cursors = s_cur; fat bins -> generic path
            uint32_t h[PER], sum = 0, fat = 0;
#pragma unroll
            for (uint32_t j = 0; j < PER; j++) { h[j] = s_cur[t * PER + j]; sum += h[j]; fat |= h[j] > kDeepBinMax ? 1u : 0u; }
            uint32_t inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t nbv = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += nbv; }
            if (lane == 63u) s_wave[wave] = inc;
            if (fat) s_bad = 1u;
            __syncthreads();
            uint32_t p = inc - sum;
            for (uint32_t w = 0; w < wave; w++) p += s_wave[w];
#pragma unroll
            for (uint32_t j = 0; j < PER; j++) { s_pre[t * PER + j] = p; s_cur[t * PER + j] = p; p += h[j]; }
        }
        __syncthreads();
        SGR_STAMP(5)
        if (s_bad) {
            if constexpr (FB) {
                sort_here();                // massive depth ties: the stable radix passes
            } else if (t == 0 && w0 == 0u) {       // -> the generic per-tile sort (register classes up to 16 384 entries, global-memory passes beyond); once per tile
                const uint32_t cls = n <= 1024u ? 0u : (n <= 2048u ? 1u : (n <= 4096u ? 2u : (n <= 8192u ? 3u : (n <= 16384u ? 4u : 5u))));
                lists[(size_t)cls * list_stride + atomicAdd(&plan->count[cls], 1u)] = tile;
            }
            continue;
        }
        // ---- 4. my window: the bins that start in [w0, w0 + WIN) cover the sorted positions [wbeg, wend) (s_pre is monotone: first bin at
        // or beyond a position by binary search, the same for every thread).  FB: the tile's windows one after the other
        auto first_at = [&](uint32_t x) -> uint32_t {
            uint32_t b = 0;
            for (uint32_t step = NBF / 2; step > 0; step >>= 1) if (s_pre[b + step - 1u] < x) b += step;       // b = number of bins with s_pre < x (<= NBF - 1 probed)
            return (b == (uint32_t)NBF - 1u && s_pre[b] < x) ? n : s_pre[b];
        };
        for (;;) {
        const uint32_t wbeg = SGR_UNIFORM(first_at(w0)), wend = SGR_UNIFORM(first_at(w0 + WIN));
        {   // placement in LDS, bin-ordered (rank inside the bin = one returning LDS atomic; the order inside a bin is settled below)
            uint32_t pp[RI];
#pragma unroll
            for (uint32_t it = 0; it < RI; it++) pp[it] = s_pre[(fbr[it / 2] >> (16u * (it & 1u))) & 0xFFFFu];
#pragma unroll
            for (uint32_t it = 0; it < RI; it++)
                if (it * NT + t < n && pp[it] >= w0 && pp[it] < w0 + WIN) s_comp[atomicAdd(&s_cur[(fbr[it / 2] >> (16u * (it & 1u))) & 0xFFFFu], 1u) - w0] = c[it];
            for_each_rest([&](uint64_t v) { const uint32_t fb = fine_bin(v), p = s_pre[fb]; if (p >= w0 && p < w0 + WIN) s_comp[atomicAdd(&s_cur[fb], 1u) - w0] = v; });
        }
        __syncthreads();
        SGR_STAMP(6)
        // every composite counts the smaller ones of its own bin: its final place (two composites per trip: the chains of dependent LDS reads overlap)
        auto place_of = [&](uint32_t q, uint64_t &v, uint32_t &fb_out) -> uint32_t {
            v = s_comp[q - w0];
            const uint32_t fb = fine_bin(v);
            const uint32_t st = s_pre[fb] - w0, en = s_cur[fb] - w0, el = en - 1u;
            uint32_t rank = 0;
            for (uint32_t k = st; k < en; k += 4u) {                           // four neighbours per trip (reads clamped to the bin, the surplus not counted)
                const uint64_t x0 = s_comp[k], x1 = s_comp[min(k + 1u, el)], x2 = s_comp[min(k + 2u, el)], x3 = s_comp[min(k + 3u, el)];
                rank += (x0 < v ? 1u : 0u) + ((k + 1u < en && x1 < v) ? 1u : 0u) + ((k + 2u < en && x2 < v) ? 1u : 0u) + ((k + 3u < en && x3 < v) ? 1u : 0u);
            }
            fb_out = fb;
            return range.x + s_pre[fb] + rank;
        };
        for (uint32_t q = wbeg + t; q < wend; q += 2u * NT) {
            uint64_t v0, v1 = 0;
            uint32_t f0, f1;
            const bool two = q + NT < wend;
            const uint32_t g0 = place_of(q, v0, f0);
            const uint32_t g1 = two ? place_of(q + NT, v1, f1) : 0u;
            if (keep_keys) { dst_keys[g0] = ((uint64_t)tile << 32) | (v0 >> 32); if (two) dst_keys[g1] = ((uint64_t)tile << 32) | (v1 >> 32); }
            dst_vals[g0] = (uint32_t)v0;
            if (two) dst_vals[g1] = (uint32_t)v1;
        }
        SGR_STAMP(7)
        if (!FB || wend >= n) break;
        w0 += WIN;
        __syncthreads();                                                        // (s_comp is refilled by the next window)
        }
    }
#undef SGR_UNIFORM
}

}  // namespace

int sgr_tile_sort_regs_launch(uint32_t grid, hipStream_t stream, const uint2 *ranges, uint64_t *src_comp, uint32_t *src_scratch, uint64_t *dst_keys,
                              uint32_t *dst_vals, const TileWork4 &tw, int m_hi, int m_lo, int keep_keys, uint32_t *prep_order, uint32_t prep_tiles) {
    hipLaunchKernelGGL(tile_sort_regs_kernel<16>, dim3(grid), dim3(1024), 0, stream, ranges, src_comp, src_scratch, dst_keys, dst_vals, tw, m_hi, m_lo, keep_keys,
                       prep_order, prep_tiles);
    SGR_CHECK_LAUNCH("tile_sort_regs_kernel");
    return 0;
}

int sgr_deep_tile_launch(int which, uint32_t grid, hipStream_t stream, uint64_t *comp, uint32_t *scratch, uint64_t *dst_keys, uint32_t *dst_vals,
                         const uint32_t *count_ptr, uint32_t *deep_list, const uint2 *ranges, int keep_keys, VsegPlan *plan, uint32_t *lists,
                         uint32_t list_stride, const GatherFront &gf) {
    if (which == 0)
        hipLaunchKernelGGL((deep_tile_kernel<1024, kDeepBigCap, 4096>), dim3(grid), dim3(1024), 0, stream, comp, scratch, dst_keys, dst_vals, count_ptr, deep_list, ranges,
                           keep_keys, plan, lists, list_stride, gf);
    else if (which == 1)
        hipLaunchKernelGGL((deep_tile_kernel<512, kDeepSmallCap, 1024>), dim3(grid), dim3(512), 0, stream, comp, scratch, dst_keys, dst_vals, count_ptr, deep_list, ranges,
                           keep_keys, plan, lists, list_stride, gf);
    else
        hipLaunchKernelGGL((deep_tile_kernel<512, kDeepSmallCap, 1024, true>), dim3(grid), dim3(512), 0, stream, comp, scratch, dst_keys, dst_vals, count_ptr, deep_list, ranges,
                           keep_keys, plan, lists, list_stride, gf);
    SGR_CHECK_LAUNCH("deep_tile_kernel");
    return 0;
}
