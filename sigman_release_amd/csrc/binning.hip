// binning.hip -- tile binning for gfx950:
//   F3  emit one (key,value) per touched tile            (duplicate_keys_kernel)
//   F4  stable LSD radix sort, 8-bit digits, wave64 ballot ranking
//   F5  per-tile [start,end) ranges                       (tile_ranges_kernel)
// Replaces duplicateWithKeys + cub::DeviceRadixSort::SortPairs + identifyTileRanges of the third-party
// rasterizer behind /root/reference/core/gaussians/gs.py:98-106, for ALL views of a batch at once:
//   key   = ((view * tiles + tile) << 32) | float_bits(depth)      (depth > 0.2, so float order == uint order)
//   value = view * P + gaussian   (index of the packed record the render kernels gather)
// The sort is stable, so equal (tile, depth) keys keep emission order = ascending Gaussian index,
// exactly like the published algorithm; the result is therefore the unique total order on
// (tile, depth bits, index) and is bit-exact against the CPU oracle.
//
// HBM traffic: emission 12 B/instance written; each radix pass reads 8 B (upsweep) + 12 B (downsweep)
// and writes 12 B per instance; only ceil((32 + bits(n_views*tiles)) / 8) passes are run.
// No inter-workgroup communication inside a launch (upsweep / scan / downsweep are separate launches),
// so there is nothing placement- or dispatch-order-dependent here.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
// keys per thread per workgroup: small inputs (one 512^2 view: R ~ 2e5) want many small workgroups to fill 256 CUs,
// large batches want fewer, longer ones (less histogram traffic)
constexpr int kItemsSmall = 4, kItemsLarge = 16;

struct DupExtra {
    const uint32_t *self_sums;          // un-scanned per-workgroup tile counts (NULL: block_offsets already holds the scan)
    uint64_t *num_rendered;             // [2] device counter + overflow flag (self-scan mode)
    uint64_t *nr_host;                  // optional pinned host copy of the same
    unsigned long long capacity;
    uint32_t *zero_ptr[3];              // optional buffers to clear on the side: tile ranges, backward flags, a caller buffer
    uint32_t zero_words[3];
    uint32_t *zero_one;                 // optional single word to clear (the tile-sort worklist counter)
};

// ---- F3 -----------------------------------------------------------------------------------------
// Same grid as preprocess (blockIdx.y = view).  The block re-scans its 256 tile counts in LDS and adds
// the block offset from F2, so the per-Gaussian offsets array of the published algorithm is never
// materialised in HBM.
__global__ __launch_bounds__(kThreads) void duplicate_keys_kernel(int P, int Tx, int tiles_per_view,
                                                                  float4 *__restrict__ rec,
                                                                  const int32_t *__restrict__ radii,
                                                                  const uint2 *__restrict__ rect,
                                                                  const uint32_t *__restrict__ block_offsets, uint32_t cap,
                                                                  uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                                  DupExtra ex) {
    __shared__ uint32_t wave_tot[4];
    __shared__ unsigned long long red64[4];
    const int view = blockIdx.y;
    // piggy-backed clear of a small buffer the later kernels expect zeroed (the tile ranges): replaces a memset launch
#pragma unroll
    for (int c = 0; c < 3; c++)
        for (uint32_t z = (blockIdx.y * gridDim.x + blockIdx.x) * kThreads + threadIdx.x; z < ex.zero_words[c]; z += gridDim.x * gridDim.y * kThreads)
            ex.zero_ptr[c][z] = 0u;
    if (ex.zero_one && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *ex.zero_one = 0u;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cnt = 0;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    size_t q = 0;
    if (i < P) {
        q = (size_t)view * P + i;
        if (radii[q] > 0) {
            const uint2 r = rect[q];
            minx = r.x & 0xFFFF; miny = r.x >> 16; maxx = r.y & 0xFFFF; maxy = r.y >> 16;
            cnt = (uint32_t)((maxx - minx) * (maxy - miny));
        }
    }
    // exclusive scan of cnt across the block: wave-level inclusive scan via shuffles, then wave totals
    uint32_t inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(inc, off, 64);
        if (lane >= off) inc += n;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base;
    if (ex.self_sums) {
        // F2 folded in (small launches, sync-free mode): every workgroup sums the un-scanned counts of the workgroups before it
        // (at most a few thousand values) instead of waiting for a separate one-workgroup scan kernel; workgroup 0 also
        // publishes the total (device counter, overflow flag and -- if given -- the caller's pinned host slot)
        const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x, nb = gridDim.x * gridDim.y;
        const uint32_t upto = b == 0 ? nb : b;
        unsigned long long acc = 0;
        for (uint32_t k = threadIdx.x; k < upto; k += kThreads) acc += ex.self_sums[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) red64[wave] = acc;
        __syncthreads();
        const unsigned long long tot = (red64[0] + red64[1]) + (red64[2] + red64[3]);
        base = b == 0 ? 0u : (uint32_t)tot;
        if (b == 0 && threadIdx.x == 0) {
            const unsigned long long ovf = (tot > 0xFFFFFFF0ull || tot > ex.capacity) ? 1ull : 0ull;
            ex.num_rendered[0] = tot; ex.num_rendered[1] = ovf; ex.num_rendered[2] = tot | (ovf << 63);
            // ONE 8-byte store: the host can never observe the count without its overflow flag
            if (ex.nr_host) { __hip_atomic_store(ex.nr_host, tot | (ovf << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); }
        }
    } else {
        base = block_offsets[(size_t)view * gridDim.x + blockIdx.x];
    }
    const uint32_t block_base = base;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    const uint32_t off = base + inc - cnt;
    // ---- cooperative emission: the workgroup's keys form one contiguous run [block_base, block_base + total); output slot j is
    // written by thread j % 256 (perfectly coalesced 8-byte / 4-byte stores, no divergence between small and huge splats), which
    // finds the owning Gaussian by binary search over the 256 offsets in LDS.  (One thread per Gaussian looping over its own rect
    // wrote runs of 2-4 keys per lane: 0.7 TB/s at 64 views.)
    __shared__ uint32_t s_off[kThreads + 1], s_geo[kThreads], s_w[kThreads], s_dep[kThreads];
    s_off[threadIdx.x] = off - block_base;
    s_geo[threadIdx.x] = (uint32_t)minx | ((uint32_t)miny << 16);
    s_w[threadIdx.x] = (uint32_t)(maxx - minx);
    if (threadIdx.x == kThreads - 1) s_off[kThreads] = off - block_base + cnt;
    if (cnt) {
        s_dep[threadIdx.x] = __float_as_uint(rec[q * 4 + 1].z);
        rec[q * 4 + 3].x = __uint_as_float(off);           // first tile-instance index of this Gaussian (backward gather)
    }
    __syncthreads();
    const uint32_t total = s_off[kThreads];
    const uint32_t tbase = (uint32_t)view * (uint32_t)tiles_per_view;
    const uint32_t q0 = (uint32_t)view * (uint32_t)P + blockIdx.x * kThreads;
    for (uint32_t j = threadIdx.x; j < total; j += kThreads) {
        uint32_t lo = 0;                                       // largest o with s_off[o] <= j (runs of equal offsets end in the owner)
#pragma unroll
        for (uint32_t step = kThreads / 2; step > 0; step >>= 1)
            if (s_off[lo + step] <= j) lo += step;
        const uint32_t local = j - s_off[lo], w = s_w[lo], g = s_geo[lo];
        uint32_t y = (uint32_t)((float)local / (float)w);
        if (y * w > local) y--;
        if ((y + 1u) * w <= local) y++;
        const uint32_t x = local - y * w;
        const uint32_t dst = block_base + j;
        if (dst < cap) {                                       // capacity mode: never write past the caller's buffers
            keys[dst] = ((uint64_t)(tbase + ((g >> 16) + y) * (uint32_t)Tx + (g & 0xFFFFu) + x) << 32) | s_dep[lo];
            vals[dst] = q0 + lo;
        }
    }
}

// ---- F4: radix sort -------------------------------------------------------------------------------
// upsweep: per-block digit histogram, stored digit-major hist[d * nblocks + b] so that one linear
// exclusive scan yields, for every (digit, block), the global output offset of that block's first key
// with that digit.
template <int ITEMS>
__global__ __launch_bounds__(kThreads) void radix_upsweep_kernel(const uint64_t *__restrict__ keys, uint32_t n_host, const uint64_t *__restrict__ n_dev,
                                                                 int shift, uint32_t nblocks, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[kRadix];
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (kThreads * ITEMS);
#pragma unroll 4
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + threadIdx.x;
        if (k < n) atomicAdd(&h[(uint32_t)(keys[k] >> shift) & (kRadix - 1)], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// per-digit row scan: workgroup d turns hist[d][0..nblocks) into its exclusive prefix (in place) and writes the digit total.
// 256 workgroups run in parallel; the 256 digit totals are scanned by every downsweep workgroup itself (in LDS).
__global__ __launch_bounds__(kThreads) void radix_rowscan_kernel(uint32_t *__restrict__ hist, uint32_t nblocks,
                                                                 uint32_t *__restrict__ totals) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t carry_s;
    uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kThreads) {
        const uint32_t idx = base + t;
        const uint32_t v = idx < nblocks ? row[idx] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nb = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += nb;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t pre = carry_s;
        for (uint32_t w = 0; w < wave; w++) pre += wave_tot[w];
        if (idx < nblocks) row[idx] = pre + inc - v;
        __syncthreads();
        if (t == kThreads - 1) carry_s = pre + inc;
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = carry_s;
}

// downsweep: stable scatter.  Keys are consumed in rounds of 256 in memory order; inside a round the rank
// of a key among equal digits is (keys of earlier waves) + (earlier lanes of its own wave), the latter
// from a wave64 "match-any" built out of 8 ballots.
template <int ITEMS>
__global__ __launch_bounds__(kThreads) void radix_downsweep_kernel(const uint64_t *__restrict__ keys_in,
                                                                   const uint32_t *__restrict__ vals_in,
                                                                   uint64_t *__restrict__ keys_out,
                                                                   uint32_t *__restrict__ vals_out, uint32_t n_host,
                                                                   const uint64_t *__restrict__ n_dev, int shift,
                                                                   uint32_t nblocks, const uint32_t *__restrict__ hist,
                                                                   const uint32_t *__restrict__ totals) {
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    __shared__ uint32_t digit_base[kRadix];
    __shared__ uint32_t wave_cnt[4][kRadix];
    __shared__ uint32_t wtot[4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {   // exclusive scan of the 256 digit totals (one per thread) + this block's offset inside its digit
        const uint32_t v = totals[t];
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nb = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += nb;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t w = 0; w < wave; w++) pre += wtot[w];
        digit_base[t] = pre + inc - v + hist[(size_t)t * nblocks + blockIdx.x];
    }
    const uint32_t base = blockIdx.x * (kThreads * ITEMS);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // all of the tile's keys/values are fetched up front (ITEMS independent loads in flight): loading inside the ranking loop
    // exposes one full memory latency per round (SQ counters: 9 % VALU-active, 74 % waiting)
    uint64_t keys_r[ITEMS];
    uint32_t vals_r[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        keys_r[it] = 0; vals_r[it] = 0;
        if (k < n) { keys_r[it] = keys_in[k]; vals_r[it] = vals_in[k]; }
    }
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        const bool valid = k < n;
        const uint64_t key = keys_r[it];
        const uint32_t val = vals_r[it];
        const uint32_t d = (uint32_t)(key >> shift) & (kRadix - 1);
#pragma unroll
        for (int w = 0; w < 4; w++) wave_cnt[w][t] = 0;
        __syncthreads();
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
        if (valid) {
            uint32_t pos = digit_base[d] + rank;
            for (uint32_t w = 0; w < wave; w++) pos += wave_cnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        digit_base[t] += wave_cnt[0][t] + wave_cnt[1][t] + wave_cnt[2][t] + wave_cnt[3][t];
        __syncthreads();
    }
}

// ---- F4, wide-digit variant for the tile bits of small launches ---------------------------------------
// With <= 2048 tiles (one 512^2 view = 1024) the whole tile id fits ONE 11-bit digit: a single stable pass (three
// kernels) instead of two 8-bit passes (six).  Same structure as above; the per-round bookkeeping only touches the
// digits that occur in the round (leaders of each wave's match-any groups), never the whole 2048-entry tables.
constexpr int kWideBits = 11, kWide = 1 << kWideBits;

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void wide_upsweep_kernel(const uint64_t *__restrict__ keys, uint32_t n_host, const uint64_t *__restrict__ n_dev,
                                                                int shift, uint32_t nblocks, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[kWide];
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    for (int d = threadIdx.x; d < kWide; d += kThreads) h[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (kThreads * ITEMS);
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + threadIdx.x;
        if (k < n) atomicAdd(&h[(uint32_t)(keys[k] >> shift) & (kWide - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < kWide; d += kThreads) hist[(size_t)d * nblocks + blockIdx.x] = h[d];
}

// one wave per digit row (rows are ~200 entries at C2): 4 rows per workgroup, exclusive prefix in place + digit total
__global__ __launch_bounds__(kThreads) void wide_rowscan_kernel(uint32_t *__restrict__ hist, uint32_t nblocks, uint32_t *__restrict__ totals) {
    const uint32_t lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t *row = hist + (size_t)d * nblocks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 64) {
        const uint32_t idx = base + lane;
        const uint32_t v = idx < nblocks ? row[idx] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nb = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += nb;
        }
        if (idx < nblocks) row[idx] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) totals[d] = carry;
}

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void wide_downsweep_kernel(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                  uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n_host,
                                                                  const uint64_t *__restrict__ n_dev, int shift, uint32_t nblocks,
                                                                  const uint32_t *__restrict__ hist, const uint32_t *__restrict__ totals,
                                                                  uint2 *__restrict__ ranges, uint32_t tiles_total,
                                                                  uint32_t *__restrict__ worklist) {
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    __shared__ uint32_t digit_base[kWide];
    __shared__ uint32_t wave_cnt[4][kWide];
    __shared__ uint32_t wtot[4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int PER = kWide / kThreads;                        // 8 consecutive digits per thread
    {   // exclusive scan of the 2048 digit totals + this workgroup's offset inside each digit; wave_cnt starts zeroed
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { v[j] = totals[t * PER + j]; sum += v[j]; }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nb = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += nb;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (uint32_t w = 0; w < wave; w++) run += wtot[w];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const uint32_t d = t * PER + j;
            digit_base[d] = run + hist[(size_t)d * nblocks + blockIdx.x];
            // F5 for free: the digit IS the tile id, so the scanned totals are the tile ranges (and the occupied tiles the
            // per-tile sort's worklist, whose counter the duplicate kernel cleared)
            if (ranges && blockIdx.x == 0 && d < tiles_total) {
                ranges[d] = v[j] ? make_uint2(run, run + v[j]) : make_uint2(0u, 0u);
                if (v[j] && worklist) worklist[1 + atomicAdd(&worklist[0], 1u)] = d;
            }
            run += v[j];
#pragma unroll
            for (int w = 0; w < 4; w++) wave_cnt[w][d] = 0;
        }
    }
    const uint32_t base = blockIdx.x * (kThreads * ITEMS);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint64_t keys_r[ITEMS];
    uint32_t vals_r[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        keys_r[it] = 0; vals_r[it] = 0;
        if (k < n) { keys_r[it] = keys_in[k]; vals_r[it] = vals_in[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        const bool valid = k < n;
        const uint64_t key = keys_r[it];
        const uint32_t val = vals_r[it];
        const uint32_t d = (uint32_t)(key >> shift) & (kWide - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kWideBits; b++) {
            const bool bit = (d >> b) & 1;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t grp = (uint32_t)__popcll(peers);
        const bool leader = valid && rank == 0;
        if (leader) wave_cnt[wave][d] = grp;
        __syncthreads();
        if (valid) {
            uint32_t pos = digit_base[d] + rank;
            for (uint32_t w = 0; w < wave; w++) pos += wave_cnt[w][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        if (leader) { atomicAdd(&digit_base[d], grp); wave_cnt[wave][d] = 0; }
        __syncthreads();
    }
}

// ---- F4, onesweep variant: ONE kernel per digit instead of three --------------------------------------
// (a) radix_hist_all_kernel reads the keys once and builds the global histograms of every pass (the multiset of keys
//     does not change between passes, so all digit histograms can be taken up front);
// (b) per pass, radix_onesweep_kernel's workgroups draw tile ids from an atomic ticket (so every lower-numbered tile is
//     owned by a workgroup that is already running), publish their per-digit counts in a status word
//     (2 flag bits | 30-bit value) and resolve their exclusive prefix by decoupled look-back over the earlier tiles.
// The status word carries its own payload and only moves 0 -> AGGREGATE -> INCLUSIVE, so a stale read is merely an
// older valid state: relaxed agent-scope atomics suffice, no fences, no placement assumptions (guide G16).  Spins are
// bounded: if a status never shows up the kernel raises an error flag and returns instead of hanging.
constexpr uint32_t kFlagAgg = 1u << 30, kFlagInc = 2u << 30, kValMask = (1u << 30) - 1u;
constexpr int kMaxPasses = 8;

__global__ __launch_bounds__(kThreads) void radix_hist_all_kernel(const uint64_t *__restrict__ keys, uint32_t n_host,
                                                                  const uint64_t *__restrict__ n_dev, int passes,
                                                                  uint32_t *__restrict__ ghist /*[passes][256]*/) {
    __shared__ uint32_t h[kMaxPasses][kRadix];
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    for (int p = 0; p < passes; p++) h[p][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    // (uniform trip count per wave, so the ballots below are well defined)
    for (uint32_t k0 = blockIdx.x * kThreads + (threadIdx.x & ~63u); k0 < n; k0 += gridDim.x * kThreads) {
        const uint32_t k = k0 + lane;
        const bool valid = k < n;
        const uint64_t key = valid ? keys[k] : 0ull;
        const uint64_t vmask = __ballot(valid);
        for (int p = 0; p < passes; p++) {
            const uint32_t d = (uint32_t)(key >> (p * kRadixBits)) & (kRadix - 1);
            // keys arrive grouped by view and tile row, so in the upper digits a whole wave usually agrees: one atomic instead of a
            // 64-way same-address LDS conflict
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            if (__ballot(valid && d != d0) == 0ull) {
                if (lane == (uint32_t)__builtin_ctzll(vmask | (1ull << 63)) && vmask) atomicAdd(&h[p][d0], (uint32_t)__popcll(vmask));
            } else if (valid) {
                atomicAdd(&h[p][d], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; p++) {
        const uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&ghist[p * kRadix + threadIdx.x], c);
    }
}

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void radix_onesweep_kernel(const uint64_t *__restrict__ keys_in,
                                                                  const uint32_t *__restrict__ vals_in,
                                                                  uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                                  uint32_t n_host, const uint64_t *__restrict__ n_dev, int shift,
                                                                  const uint32_t *__restrict__ ghist /*[256] of this pass*/,
                                                                  uint32_t *__restrict__ status /*[tiles][256] of this pass*/,
                                                                  uint32_t *__restrict__ ticket, uint32_t *__restrict__ err) {
    __shared__ uint32_t digit_base[kRadix];
    __shared__ uint32_t hist[kRadix];
    __shared__ uint32_t wave_cnt[4][kRadix];
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t s_tile;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    hist[t] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * (kThreads * ITEMS);
    if (base >= n) return;                                     // workgroup-uniform
    uint64_t key[ITEMS];
    uint32_t val[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        key[it] = 0; val[it] = 0;
        if (k < n) { key[it] = keys_in[k]; val[it] = vals_in[k]; atomicAdd(&hist[(uint32_t)(key[it] >> shift) & (kRadix - 1)], 1u); }
    }
    __syncthreads();
    {   // thread t owns digit t: global start of the digit (scan of the pass histogram) + tiles before mine (look-back)
        const uint32_t g = ghist[t];
        uint32_t inc = g;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t nb = __shfl_up(inc, off, 64);
            if (lane >= (uint32_t)off) inc += nb;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t gstart = inc - g;
        for (uint32_t w = 0; w < wave; w++) gstart += wtot[w];
        const uint32_t cnt = hist[t];
        uint32_t excl = 0;
        uint32_t *mine = status + (size_t)tile * kRadix + t;
        if (tile == 0) {
            __hip_atomic_store(mine, cnt | kFlagInc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(mine, cnt | kFlagAgg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int tt = (int)tile - 1; tt >= 0; --tt) {
                const uint32_t *p = status + (size_t)tt * kRadix + t;
                uint32_t v = 0;
                uint32_t spins = 0;
                do {
                    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((v >> 30) == 0u && ++spins < (1u << 22));
                if ((v >> 30) == 0u) { atomicOr(err, 1u); break; }          // bounded spin: report instead of hanging
                excl += v & kValMask;
                if ((v >> 30) == 2u) break;
            }
            __hip_atomic_store(mine, ((excl + cnt) & kValMask) | kFlagInc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        digit_base[t] = gstart + excl;
    }
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        const bool valid = k < n;
        const uint32_t d = (uint32_t)(key[it] >> shift) & (kRadix - 1);
#pragma unroll
        for (int w = 0; w < 4; w++) wave_cnt[w][t] = 0;
        __syncthreads();
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
        if (valid) {
            uint32_t pos = digit_base[d] + rank;
            for (uint32_t w = 0; w < wave; w++) pos += wave_cnt[w][d];
            keys_out[pos] = key[it];
            vals_out[pos] = val[it];
        }
        __syncthreads();
        digit_base[t] += wave_cnt[0][t] + wave_cnt[1][t] + wave_cnt[2][t] + wave_cnt[3][t];
        __syncthreads();
    }
}

// ---- F4, segmented variant: the key is (tile, depth), so sort the TILE bits globally and the DEPTH bits per tile ----
// Global LSD passes are run over the tile-id bits only (2 passes for up to 65536 tiles instead of 6-7 over the
// whole key); they are stable, so afterwards every tile owns a contiguous segment whose entries are still in
// emission order (ascending Gaussian index).  tile_ranges then finds the segments and tile_sort_kernel sorts each
// one by its 32 depth bits with a stable LSD radix sort that lives entirely in LDS (160 KB/CU on MI355X: a 4096-entry
// segment needs 64 KB for two key/value ping-pong pairs).  Digits on which a whole segment agrees are skipped -- the
// exponent byte of the depths inside one tile almost always is.  Segments longer than the LDS capacity are sorted by
// the same code through the global ping-pong buffers.  The result is bit-identical to one stable sort of the full key.
constexpr int kSegCapSmall = 1024, kSegCapLarge = 4096;

template <int NT, bool IN_LDS>
__device__ __forceinline__ void seg_sort_passes(uint32_t n, uint32_t *ka, uint32_t *va, uint32_t *kb, uint32_t *vb,
                                                uint64_t *gka, uint32_t *gva, uint64_t *gkb, uint32_t *gvb, bool &in_b,
                                                uint32_t *hist, uint32_t *digit_base, uint32_t (*wave_cnt)[kRadix], uint32_t *wtot) {
    constexpr int NW = NT / 64;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int shift = 0; shift < 32; shift += kRadixBits) {
        // ---- histogram of this digit over the segment
        if (t < kRadix) hist[t] = 0;
        __syncthreads();
        for (uint32_t k = t; k < n; k += NT) {
            const uint32_t key = IN_LDS ? (in_b ? kb[k] : ka[k]) : (uint32_t)(in_b ? gkb[k] : gka[k]);
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
        __shared__ uint32_t gsum[TWO_LEVEL ? 4 : 1][kRadix];
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
                if (IN_LDS) { key = in_b ? kb[k] : ka[k]; val = in_b ? vb[k] : va[k]; }
                else { key64 = in_b ? gkb[k] : gka[k]; key = (uint32_t)key64; val = in_b ? gvb[k] : gva[k]; }
            }
            const uint32_t d = (key >> shift) & (kRadix - 1);
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
                if (IN_LDS) { (in_b ? ka : kb)[pos] = key; (in_b ? va : vb)[pos] = val; }
                else { (in_b ? gka : gkb)[pos] = key64; (in_b ? gva : gvb)[pos] = val; }
            }
            __syncthreads();
            if (TWO_LEVEL) {
#pragma unroll
                for (int k = 0; k < 4; k++) wave_cnt[4 * pg + k][pd] = 0;
                if (pg == 0) digit_base[pd] += (gsum[0][pd] + gsum[1][pd]) + (gsum[2][pd] + gsum[3][pd]);
            } else if (t < kRadix) {
                uint32_t add = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) add += wave_cnt[w][t];
                digit_base[t] += add;
            }
            __syncthreads();
        }
        in_b = !in_b;
        if (!IN_LDS) { __threadfence_block(); __syncthreads(); }
    }
}

// src = buffers holding the tile-bucketed data, dst = the other pair; the sorted segment always ends up in dst.
// Two size classes share the code: CAP=1024 / 256 threads (16 KB of LDS: full occupancy for the many short lists) handles
// segments of <= 1024 entries, CAP=4096 / 1024 threads handles the rest (and oversize segments through global memory).
// (launching a 1024-thread / 80-KB-LDS workgroup per tile just to have it exit costs ~10 us per CU slot, so the small-class
// launch, one workgroup per tile, appends the long tiles to a worklist that a fixed-size large-class grid then drains)
// (a launch with few tiles -- one 512^2 view -- skips the small class: tile_ranges_kernel puts every occupied tile on the worklist
// and the large-class workgroups take one each, all resident at once, so the two classes no longer run back to back)
struct SortPrep {               // optional piggy-back job of the large-class launch's spare last workgroup (sgr_fwd_prepare)
    uint2 *desc; size_t n_desc; uint32_t *order; uint32_t tiles_total; int enabled;
};

template <int NT, int CAP, bool SMALL_CLASS>
__global__ __launch_bounds__(NT) void tile_sort_kernel(const uint2 *__restrict__ ranges, uint64_t *__restrict__ src_keys,
                                                       uint32_t *__restrict__ src_vals, uint64_t *__restrict__ dst_keys,
                                                       uint32_t *__restrict__ dst_vals, uint32_t *__restrict__ worklist /*[0]=count, [1..]=tiles*/,
                                                       SortPrep prep) {
    __shared__ uint32_t ka[CAP], va[CAP], kb[CAP], vb[CAP];
    __shared__ uint32_t hist[kRadix], digit_base[kRadix], wave_cnt[NT / 64][kRadix], wtot[4];
    const uint32_t t = threadIdx.x;
    uint32_t nsort = gridDim.x;
    if (!SMALL_CLASS && NT == 1024 && prep.enabled) {
        nsort = gridDim.x - 1;
        if (blockIdx.x == nsort) { sgr_fwd_prepare(ranges, prep.tiles_total, prep.desc, prep.n_desc, prep.order, hist); return; }
    }
    const uint32_t nwork = SMALL_CLASS ? 1u : worklist[0];
    for (uint32_t wi = SMALL_CLASS ? 0u : blockIdx.x; wi < nwork; wi += nsort) {
    const uint32_t tile_id = SMALL_CLASS ? blockIdx.x : worklist[1 + wi];
    const uint2 range = ranges[tile_id];
    const uint32_t n = range.y - range.x;
    if (n == 0) return;
    if (SMALL_CLASS && n > (uint32_t)kSegCapSmall) {
        if (t == 0) worklist[1 + atomicAdd(&worklist[0], 1u)] = tile_id;
        return;
    }
    __syncthreads();                                               // (large class: LDS reuse between worklist items)
    uint64_t *gsrc_k = src_keys + range.x, *gdst_k = dst_keys + range.x;
    uint32_t *gsrc_v = src_vals + range.x, *gdst_v = dst_vals + range.x;
    bool in_b = false;
    if (n <= (uint32_t)CAP) {
        const uint32_t hi = (uint32_t)(gsrc_k[0] >> 32);                      // tile id, identical for the whole segment
        for (uint32_t k = t; k < n; k += NT) { ka[k] = (uint32_t)gsrc_k[k]; va[k] = gsrc_v[k]; }
        __syncthreads();
        if (n > 1) seg_sort_passes<NT, true>(n, ka, va, kb, vb, nullptr, nullptr, nullptr, nullptr, in_b, hist, digit_base, wave_cnt, wtot);
        __syncthreads();
        const uint32_t *fk = in_b ? kb : ka, *fv = in_b ? vb : va;
        for (uint32_t k = t; k < n; k += NT) { gdst_k[k] = ((uint64_t)hi << 32) | fk[k]; gdst_v[k] = fv[k]; }
    } else {
        // oversize segment: same algorithm through the global ping-pong pair (a = src, b = dst)
        seg_sort_passes<NT, false>(n, nullptr, nullptr, nullptr, nullptr, gsrc_k, gsrc_v, gdst_k, gdst_v, in_b, hist, digit_base, wave_cnt, wtot);
        __threadfence_block();
        __syncthreads();
        if (!in_b)                                               // result sits in src: move it to dst
            for (uint32_t k = t; k < n; k += NT) { gdst_k[k] = gsrc_k[k]; gdst_v[k] = gsrc_v[k]; }
    }
    }
}

// ---- F5 -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void tile_ranges_kernel(const uint64_t *__restrict__ keys, uint32_t n_host,
                                                               const uint64_t *__restrict__ n_dev, uint2 *__restrict__ ranges,
                                                               uint32_t *__restrict__ worklist, int append_all) {
    // worklist[0] = counter: append_all == 0 -> reset here for the small-class tile sort that fills it afterwards;
    // append_all != 0 -> (already cleared by the duplicate kernel) every occupied tile is appended by the thread at its segment start
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    if (worklist && !append_all && blockIdx.x == 0 && threadIdx.x == 0) worklist[0] = 0;
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= n) return;
    const uint32_t tile = (uint32_t)(keys[r] >> 32);
    bool first = r == 0;
    if (r == 0) ranges[tile].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[r - 1] >> 32);
        if (tile != prev) { ranges[prev].y = r; ranges[tile].x = r; first = true; }
    }
    if (r == n - 1) ranges[tile].y = n;
    if (append_all && first) worklist[1 + atomicAdd(&worklist[0], 1u)] = tile;
}

// 3 = automatic (default, by instance count: measured crossovers on MI355X), 2 = segmented (tile bits globally, depth bits per tile in LDS),
// 0 = onesweep, 1 = three kernels per pass
int sgr_sort_mode = 3;

inline int bits_for(uint64_t v) { int b = 0; while ((1ull << b) < v) b++; return b; }   // smallest b with 2^b >= v

}  // namespace

int sgr_validate_problem(const SgrProblem *pb);

extern "C" int sgr_set_sort_mode(int mode) { sgr_sort_mode = mode; return 0; }

extern "C" size_t sgr_bin_workspace_bytes(uint64_t R, uint64_t tiles_total) {
    const uint64_t nblocks = (R + kThreads * kItemsSmall - 1) / (kThreads * kItemsSmall);
    // onesweep: [ghist 8x256][tickets 8][err][pad] + status [8 passes][tiles][256]; three-kernel path: [hist tiles x 256][totals 256]
    return (size_t)((kMaxPasses * (nblocks > 0 ? nblocks : 1) + kMaxPasses + 2) * kRadix * sizeof(uint32_t) + 1024 +
                    (tiles_total ? (tiles_total + 64) * sizeof(uint32_t) : 0));
}

// self_scan: the caller skipped the F2 scan kernel (sgr_preprocess_forward_ex) and block_offsets + n + 1 holds the un-scanned
// counts; num_rendered_dev is then WRITTEN by the duplicate kernel (capacity mode only).  nr_host: optional pinned host slot.
int sgr_bin_ex(const SgrProblem *pb, float *rec, const int32_t *radii, const uint32_t *rect,
               const uint32_t *block_offsets, uint64_t R, const uint64_t *num_rendered_dev, uint64_t *keys_a,
               uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *workspace, size_t workspace_bytes,
               uint32_t *ranges, int32_t *result_in_b_host, bool self_scan, uint64_t *nr_host, void *prep_desc, size_t prep_n_desc,
               uint32_t *prep_order, int *prep_done, uint32_t *const *clear_ptr /*[2] or NULL*/, const uint64_t *clear_words /*[2]*/,
               int *clear_done /*[2]*/, void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint64_t tiles_total = (uint64_t)Tx * Ty * pb->n_views;
    if (tiles_total >= (1ull << 32)) { sgr_set_error("too many tiles (%llu)", (unsigned long long)tiles_total); return 1; }
    if (R > 0xFFFFFFF0ull) { sgr_set_error("num_rendered %llu exceeds the 32-bit instance index", (unsigned long long)R); return 1; }
    if (result_in_b_host) *result_in_b_host = 0;
    if (prep_done) *prep_done = 0;
    if (clear_done) clear_done[0] = clear_done[1] = 0;
    const bool fold_clear = R > 0 && pb->P > 0 && tiles_total * 2 <= (1u << 20);     // small: cleared by the duplicate kernel
    if (!fold_clear) SGR_CHECK_HIP(hipMemsetAsync(ranges, 0, tiles_total * 2 * sizeof(uint32_t), stream));
    if (R == 0 || pb->P == 0) return 0;
    if (workspace_bytes < sgr_bin_workspace_bytes(R, tiles_total)) { sgr_set_error("sgr_bin: workspace too small"); return 1; }
    const uint32_t n = (uint32_t)R;
    const int nbx = sgr_preprocess_blocks_per_view(pb->P);
    // few tiles + segmented sort: every occupied tile goes on the large-class worklist (counter cleared by the duplicate kernel,
    // filled by tile_ranges)
    const bool all_large = tiles_total <= 2048;
    { SgrProfScope _p(SGR_K_DUPLICATE, stream);
    DupExtra ex;
    const uint32_t nblk = (uint32_t)nbx * (uint32_t)pb->n_views;
    ex.self_sums = self_scan ? block_offsets + (nblk + 1) : nullptr;
    ex.num_rendered = const_cast<uint64_t *>(num_rendered_dev); ex.nr_host = self_scan ? nr_host : nullptr; ex.capacity = R;
    ex.zero_ptr[0] = fold_clear ? ranges : nullptr; ex.zero_words[0] = fold_clear ? (uint32_t)(tiles_total * 2) : 0u;
    for (int c = 0; c < 2; c++) {
        const bool ok = clear_ptr && clear_ptr[c] && clear_words && clear_words[c] <= (1ull << 26);
        ex.zero_ptr[1 + c] = ok ? clear_ptr[c] : nullptr; ex.zero_words[1 + c] = ok ? (uint32_t)clear_words[c] : 0u;
        if (clear_done) clear_done[c] = ok ? 1 : 0;
    }
    ex.zero_one = all_large ? (uint32_t *)((char *)workspace + sgr_bin_workspace_bytes(R, 0)) : nullptr;
    if (self_scan && !num_rendered_dev) { sgr_set_error("sgr_bin: self-scan needs the device counter"); return 1; }
    hipLaunchKernelGGL(duplicate_keys_kernel, dim3(nbx, pb->n_views), dim3(kThreads), 0, stream, pb->P, Tx, Tx * Ty,
                       (float4 *)rec, radii, (const uint2 *)rect, block_offsets, n, keys_a, vals_a, ex);
    SGR_CHECK_LAUNCH("duplicate_keys_kernel");
    }
    const bool small = n <= (1u << 19);
    const uint32_t tile_keys = kThreads * (small ? kItemsSmall : kItemsLarge);
    const uint32_t nblocks = (n + tile_keys - 1) / tile_keys;
    uint32_t *hist = (uint32_t *)workspace;
    uint32_t *totals = hist + (size_t)nblocks * kRadix;
    const int total_bits = 32 + bits_for(tiles_total);
    const int passes = (total_bits + kRadixBits - 1) / kRadixBits;
    uint64_t *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    // sort flavour: 2 (default) = tile bits globally + depth bits per tile in LDS; 0 = onesweep over the whole key; 1 = three kernels
    // automatic: segmented up to 2^19 instances (one 512^2 view: 74 vs 82 vs 100 us), three-kernel up to 2^23 (16 views: 285 vs
    // 323 vs 360 us), onesweep beyond (64 views: 1041 vs 1176 vs 1186 us; 90 views at 1024^2: 3.57 vs 3.82 vs 3.87 ms)
    const int mode = sgr_sort_mode != 3 ? sgr_sort_mode : (R <= (1ull << 19) ? 2 : (R <= (1ull << 23) ? 1 : 0));
    const bool segmented = mode == 2;
    if (segmented) {
        uint32_t *worklist = (uint32_t *)((char *)workspace + sgr_bin_workspace_bytes(R, 0));   // [1 + tiles_total] behind the radix scratch
        const int tile_bits = bits_for(tiles_total);
        const int tpasses = (tile_bits + kRadixBits - 1) / kRadixBits;
        const bool wide = small && tile_bits > kRadixBits && tile_bits <= kWideBits;      // one 11-bit pass instead of two 8-bit passes
        { SgrProfScope _ps(SGR_K_SORT, stream);
        if (wide) {
            hipLaunchKernelGGL(wide_upsweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, 32, nblocks, hist);
            hipLaunchKernelGGL(wide_rowscan_kernel, dim3(kWide / 4), dim3(kThreads), 0, stream, hist, nblocks, hist + (size_t)nblocks * kWide);
            hipLaunchKernelGGL(wide_downsweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, 32, nblocks, hist, hist + (size_t)nblocks * kWide,
                               (uint2 *)ranges, (uint32_t)tiles_total, worklist);
            SGR_CHECK_LAUNCH("wide tile-bit pass");
            uint64_t *tk = kin; kin = kout; kout = tk;
            uint32_t *tv = vin; vin = vout; vout = tv;
        } else
        for (int p = 0; p < tpasses; p++) {
            const int shift = 32 + p * kRadixBits;
            if (small) hipLaunchKernelGGL(radix_upsweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, shift, nblocks, hist);
            else hipLaunchKernelGGL(radix_upsweep_kernel<kItemsLarge>, dim3(nblocks), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, shift, nblocks, hist);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix), dim3(kThreads), 0, stream, hist, nblocks, totals);
            if (small) hipLaunchKernelGGL(radix_downsweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, shift, nblocks, hist, totals);
            else hipLaunchKernelGGL(radix_downsweep_kernel<kItemsLarge>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, shift, nblocks, hist, totals);
            SGR_CHECK_LAUNCH("radix tile-bit pass");
            uint64_t *tk = kin; kin = kout; kout = tk;
            uint32_t *tv = vin; vin = vout; vout = tv;
        }
        }
        if (!wide) {                                            // (the wide pass wrote the ranges and the worklist itself)
        SgrProfScope _pr(SGR_K_RANGES, stream);
        hipLaunchKernelGGL(tile_ranges_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, (uint2 *)ranges, worklist, all_large ? 1 : 0);
        SGR_CHECK_LAUNCH("tile_ranges_kernel");
        }
        { SgrProfScope _ps(SGR_K_SORT, stream);
        const uint32_t big_grid = (uint32_t)(tiles_total < 512 ? tiles_total : 512);
        SortPrep sp;
        sp.desc = (uint2 *)prep_desc; sp.n_desc = prep_n_desc; sp.order = prep_order; sp.tiles_total = (uint32_t)tiles_total;
        sp.enabled = (prep_order || prep_desc) ? 1 : 0;
        SortPrep none = sp; none.enabled = 0;
        if (!all_large)
            hipLaunchKernelGGL((tile_sort_kernel<256, kSegCapSmall, true>), dim3((uint32_t)tiles_total), dim3(256), 0, stream,
                               (const uint2 *)ranges, kin, vin, kout, vout, worklist, none);
        hipLaunchKernelGGL((tile_sort_kernel<1024, kSegCapLarge, false>), dim3(big_grid + (sp.enabled ? 1u : 0u)), dim3(1024), 0, stream,
                           (const uint2 *)ranges, kin, vin, kout, vout, worklist, sp);
        if (sp.enabled && prep_done) *prep_done = 1;
        SGR_CHECK_LAUNCH("tile_sort_kernel");
        }
        if (result_in_b_host) *result_in_b_host = (kout == keys_b) ? 1 : 0;
        return 0;
    }
    const bool onesweep = mode == 0 && passes <= kMaxPasses && R < (1ull << 30);
    { SgrProfScope _ps(SGR_K_SORT, stream);
    if (onesweep) {
        uint32_t *ws32 = (uint32_t *)workspace;
        uint32_t *ghist = ws32;                                   // [kMaxPasses][256]
        uint32_t *tickets = ghist + kMaxPasses * kRadix;          // [kMaxPasses]
        uint32_t *err = tickets + kMaxPasses;                     // [1] (+ padding to 256 entries)
        uint32_t *status = tickets + kRadix;                      // [passes][nblocks][256]
        SGR_CHECK_HIP(hipMemsetAsync(ws32, 0, ((size_t)(kMaxPasses + 1) * kRadix + (size_t)passes * nblocks * kRadix) * sizeof(uint32_t), stream));
        const uint32_t hist_blocks = nblocks < 1024u ? (nblocks ? nblocks : 1u) : 1024u;
        hipLaunchKernelGGL(radix_hist_all_kernel, dim3(hist_blocks), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, passes, ghist);
        SGR_CHECK_LAUNCH("radix_hist_all_kernel");
        for (int p = 0; p < passes; p++) {
            const int shift = p * kRadixBits;
            uint32_t *st_p = status + (size_t)p * nblocks * kRadix;
            if (small) hipLaunchKernelGGL(radix_onesweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, shift, ghist + p * kRadix, st_p, tickets + p, err);
            else hipLaunchKernelGGL(radix_onesweep_kernel<kItemsLarge>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, shift, ghist + p * kRadix, st_p, tickets + p, err);
            SGR_CHECK_LAUNCH("radix_onesweep_kernel");
            uint64_t *tk = kin; kin = kout; kout = tk;
            uint32_t *tv = vin; vin = vout; vout = tv;
        }
    } else
    for (int p = 0; p < passes; p++) {
        const int shift = p * kRadixBits;
        if (small) hipLaunchKernelGGL(radix_upsweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, shift, nblocks, hist);
        else hipLaunchKernelGGL(radix_upsweep_kernel<kItemsLarge>, dim3(nblocks), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, shift, nblocks, hist);
        SGR_CHECK_LAUNCH("radix_upsweep_kernel");
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix), dim3(kThreads), 0, stream, hist, nblocks, totals);
        SGR_CHECK_LAUNCH("radix_rowscan_kernel");
        if (small) hipLaunchKernelGGL(radix_downsweep_kernel<kItemsSmall>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, shift, nblocks, hist, totals);
        else hipLaunchKernelGGL(radix_downsweep_kernel<kItemsLarge>, dim3(nblocks), dim3(kThreads), 0, stream, kin, vin, kout, vout, n, num_rendered_dev, shift, nblocks, hist, totals);
        SGR_CHECK_LAUNCH("radix_downsweep_kernel");
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
    }
    }
    if (result_in_b_host) *result_in_b_host = (kin == keys_b) ? 1 : 0;
    { SgrProfScope _p(SGR_K_RANGES, stream);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, kin, n,
                       num_rendered_dev, (uint2 *)ranges, (uint32_t *)nullptr, 0);
    SGR_CHECK_LAUNCH("tile_ranges_kernel");
    }
    return 0;
}

extern "C" int sgr_bin(const SgrProblem *pb, float *rec, const int32_t *radii, const uint32_t *rect,
                       const uint32_t *block_offsets, uint64_t R, const uint64_t *num_rendered_dev, uint64_t *keys_a,
                       uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *workspace, size_t workspace_bytes,
                       uint32_t *ranges, int32_t *result_in_b_host, void *stream_) {
    return sgr_bin_ex(pb, rec, radii, rect, block_offsets, R, num_rendered_dev, keys_a, keys_b, vals_a, vals_b, workspace,
                      workspace_bytes, ranges, result_in_b_host, false, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, stream_);
}
