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
#include <algorithm>
#include <string.h>
#include "binning_internal.h"

namespace {

struct DupExtra {
    const uint32_t *self_sums;          // un-scanned per-workgroup tile counts (NULL: block_offsets already holds the scan)
    uint64_t *num_rendered;             // [2] device counter + overflow flag (self-scan mode)
    uint64_t *nr_host;                  // optional pinned host copy of the same
    unsigned long long capacity;
    uint32_t *zero_ptr[4];              // optional buffers to clear on the side: tile ranges, backward flags, a caller buffer, bucket descriptors
    uint32_t zero_words[4];
    uint32_t *zero_small; uint32_t zero_small_n;      // optional few words (<= 256) to clear: the tile-sort worklist counter(s)
    uint32_t *run_rows;                 // RUNS: [workgroups][kRunRow] the run matrix (above)
    uint32_t *run_base;                 // RUNS: [workgroups] first position of the workgroup's run
    uint32_t *occ;                      // RUNS: [kTileBins] tile-occupancy flags (zeroed before this launch)
    uint32_t write_first;               // store every Gaussian's first tile-instance index into rect[q].w (only the bucket backward's gathers read it;
                                        // a forward-only launch skips the 4-byte stores that dirty every line of the rect array: 0.29 GB of
                                        // write-back for the 90 views of C4)
};

// ---- F3 -----------------------------------------------------------------------------------------
// Same grid as preprocess (blockIdx.y = view).  The block re-scans its 256 tile counts in LDS and adds
// the block offset from F2, so the per-Gaussian offsets array of the published algorithm is never
// materialised in HBM.
// RUNS (the single-view path): the run is written ordered by tile, as composites, and its row of the run matrix with it (kRunRow above); one
// workgroup of NT = 1024 threads covers four of preprocess's 256-Gaussian blocks (a quarter of the rows for the per-tile sort to read, four
// times longer pieces to gather; the work per thread is the same).
template <bool RUNS, int NT>
__global__ __launch_bounds__(NT) void duplicate_keys_kernel(int P, int Tx, int tiles_per_view, int nbx /* preprocess blocks per view */,
                                                            const int32_t *__restrict__ radii,
                                                            uint4 *__restrict__ rect,
                                                            const uint32_t *__restrict__ block_offsets, uint32_t cap,
                                                            uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                            DupExtra ex) {
    constexpr int NWV = NT / 64, BPW = NT / kThreads;                  // waves; preprocess blocks per workgroup
    static_assert(NT % kThreads == 0 && (NT & (NT - 1)) == 0, "whole preprocess blocks");
    __shared__ uint32_t wave_tot[NWV];
    __shared__ unsigned long long red64[NWV];
    __shared__ __attribute__((aligned(16))) uint32_t s_th[RUNS ? kTileBins : 4];      // RUNS: this workgroup's keys per tile, then the tiles' cursors
    if (RUNS) for (int d = threadIdx.x; d < kTileBins / 4; d += NT) reinterpret_cast<uint4 *>(s_th)[d] = make_uint4(0u, 0u, 0u, 0u);
    const int view = blockIdx.y;
    // piggy-backed clear of small buffers the later kernels expect zeroed: replaces memset launches
#pragma unroll
    for (int c = 0; c < 4; c++)
        for (uint32_t z = (blockIdx.y * gridDim.x + blockIdx.x) * NT + threadIdx.x; z < ex.zero_words[c]; z += gridDim.x * gridDim.y * NT)
            ex.zero_ptr[c][z] = 0u;
    if (ex.zero_small && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < ex.zero_small_n) ex.zero_small[threadIdx.x] = 0u;
    const int i = blockIdx.x * NT + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t pb0 = (uint32_t)view * (uint32_t)nbx + blockIdx.x * (uint32_t)BPW;      // my first preprocess block
    // F2 folded in (small launches, sync-free mode): every workgroup sums the un-scanned counts of the preprocess blocks before its own (at most a
    // few thousand values) instead of waiting for a separate one-workgroup scan kernel.  Requested first: the loads travel with the rect loads
    unsigned long long acc = 0;
    if (ex.self_sums) {
        const uint32_t nb = (uint32_t)nbx * gridDim.y;
        const uint32_t upto = pb0 == 0 ? nb : pb0;
        for (uint32_t k = threadIdx.x; k < upto; k += NT) acc += ex.self_sums[k];
    }
    uint32_t cnt = 0, depth_bits = 0;
    int minx = 0, miny = 0, maxx = 0, maxy = 0;
    size_t q = 0;
    if (i < P) {
        q = (size_t)view * P + i;
        if (radii[q] > 0) {
            const uint4 r = rect[q];
            depth_bits = r.z;
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
        // workgroup 0 also publishes the total (device counter, overflow flag and -- if given -- the caller's pinned host slot)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) red64[wave] = acc;
        __syncthreads();
        unsigned long long tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) tot += red64[w];
        base = pb0 == 0 ? 0u : (uint32_t)tot;
        if (pb0 == 0 && threadIdx.x == 0) {
            const unsigned long long ovf = (tot > 0xFFFFFFF0ull || tot > ex.capacity) ? 1ull : 0ull;
            ex.num_rendered[0] = tot; ex.num_rendered[1] = ovf; ex.num_rendered[2] = tot | (ovf << 63);
            // ONE 8-byte store: the host can never observe the count without its overflow flag
            if (ex.nr_host) { __hip_atomic_store(ex.nr_host, tot | (ovf << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); }
        }
    } else {
        base = block_offsets[pb0];
    }
    const uint32_t block_base = base;
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    const uint32_t off = base + inc - cnt;
    // ---- cooperative emission: the workgroup's keys form one contiguous run [block_base, block_base + total); output slot j is
    // written by thread j % NT (perfectly coalesced 8-byte / 4-byte stores, no divergence between small and huge splats), which
    // finds the owning Gaussian by binary search over the NT offsets in LDS.  (One thread per Gaussian looping over its own rect
    // wrote runs of 2-4 keys per lane: 0.7 TB/s at 64 views.)
    __shared__ uint32_t s_off[NT + 1], s_geo[NT], s_w[NT], s_dep[NT];
    s_off[threadIdx.x] = off - block_base;
    s_geo[threadIdx.x] = (uint32_t)minx | ((uint32_t)miny << 16);
    s_w[threadIdx.x] = (uint32_t)(maxx - minx);
    if (threadIdx.x == NT - 1) s_off[NT] = off - block_base + cnt;
    if (cnt) {
        s_dep[threadIdx.x] = depth_bits;
        if (ex.write_first) rect[q].w = off;               // first tile-instance index of this Gaussian (backward gathers); the 16-byte record was just read
    }
    __syncthreads();
    const uint32_t total = s_off[NT];
    const uint32_t tbase = (uint32_t)view * (uint32_t)tiles_per_view;
    const uint32_t q0 = (uint32_t)view * (uint32_t)P + blockIdx.x * NT;
    // slot j of the run -> (owning Gaussian, tile id)
    auto slot_of = [&](uint32_t j, uint32_t &lo, uint32_t &tile_id) {
        lo = 0;                                                // largest o with s_off[o] <= j (runs of equal offsets end in the owner)
#pragma unroll
        for (uint32_t step = NT / 2; step > 0; step >>= 1)
            if (s_off[lo + step] <= j) lo += step;
        const uint32_t local = j - s_off[lo], w = s_w[lo], g = s_geo[lo];
        uint32_t y = (uint32_t)((float)local / (float)w);
        if (y * w > local) y--;
        if ((y + 1u) * w <= local) y++;
        const uint32_t x = local - y * w;
        tile_id = tbase + ((g >> 16) + y) * (uint32_t)Tx + (g & 0xFFFFu) + x;
    };
    if constexpr (!RUNS) {
        for (uint32_t j = threadIdx.x; j < total; j += NT) {
            uint32_t lo, tile_id;
            slot_of(j, lo, tile_id);
            const uint32_t dst = block_base + j;
            if (dst < cap) {                                       // capacity mode: never write past the caller's buffers
                keys[dst] = ((uint64_t)tile_id << 32) | s_dep[lo];
                vals[dst] = q0 + lo;
            }
        }
    } else {
        // (capacity mode: only the keys whose EMISSION index fits the caller's buffers exist -- the backward addresses its per-instance records
        // by that index, whatever place the key takes inside the run)
        const uint32_t total_all = total;
        const uint32_t total = block_base < cap ? min(total_all, cap - block_base) : 0u;
        // pass 1: keys per tile (the first four slots of a thread -- a run of <= 4 NT keys: all of them -- stay in registers for pass 2)
        constexpr int KEEP = 4;
        uint32_t k_lo[KEEP], k_tile[KEEP];
#pragma unroll
        for (int it = 0; it < KEEP; it++) {
            const uint32_t j = threadIdx.x + (uint32_t)it * NT;
            k_lo[it] = 0u; k_tile[it] = 0u;
            if (j < total) { slot_of(j, k_lo[it], k_tile[it]); atomicAdd(&s_th[k_tile[it]], 1u); }
        }
        for (uint32_t j = threadIdx.x + KEEP * NT; j < total; j += NT) {
            uint32_t lo, tile_id;
            slot_of(j, lo, tile_id);
            atomicAdd(&s_th[tile_id], 1u);
        }
        __syncthreads();
        // exclusive scan over the tiles (PER consecutive per thread): the row of the run matrix, and the cursors of pass 2.  Positions are
        // clamped to the capacity (sync-free mode): what the row promises is what pass 2 writes.  Tiles with keys are marked occupied.
        {
            constexpr int PER = kTileBins / NT;
            static_assert(PER >= 2 && PER % 2 == 0, "whole uint2 per thread");
            uint32_t hv[PER], sum = 0;
#pragma unroll
            for (int k = 0; k < PER; k += 2) {
                const uint2 h2 = reinterpret_cast<const uint2 *>(s_th)[(threadIdx.x * PER + k) / 2];
                hv[k] = h2.x; hv[k + 1] = h2.y; sum += h2.x + h2.y;
            }
            uint32_t sc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t n = __shfl_up(sc, o, 64);
                if (lane >= o) sc += n;
            }
            __syncthreads();                                   // (wave_tot is reused; everybody has read s_th)
            if (lane == 63) wave_tot[wave] = sc;
            __syncthreads();
            uint32_t run = sc - sum;
            for (int w = 0; w < wave; w++) run += wave_tot[w];
            const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
            uint32_t *row = ex.run_rows + (size_t)b * kRunRow;
            // (block_base + x <= 2^32 - 16: the count is checked against that)
            // total == 0 beyond the capacity: every position of the row is base_c = min(block_base, cap).  (The optimiser folded that minimum to
            // block_base -- ROCm 7.2 clang, as if block_base < cap were known here -- and the sort launch then read row positions beyond the
            // buffers: the capacity is hidden from it for this one comparison.)
            uint32_t cap_opaque = cap;
            asm volatile("" : "+s"(cap_opaque));
            const uint32_t base_c = block_base < cap_opaque ? block_base : cap_opaque;
#pragma unroll
            for (int k = 0; k < PER; k += 2) {
                const uint32_t e0 = run, e1 = run + hv[k];
                run = e1 + hv[k + 1];
                reinterpret_cast<uint2 *>(s_th)[(threadIdx.x * PER + k) / 2] = make_uint2(e0, e1);
                if (hv[k]) ex.occ[threadIdx.x * PER + k] = 1u;
                if (hv[k + 1]) ex.occ[threadIdx.x * PER + k + 1] = 1u;
                reinterpret_cast<uint2 *>(row)[(threadIdx.x * PER + k) / 2] = make_uint2(base_c + e0, base_c + e1);
            }
            if (threadIdx.x == 0) {
                reinterpret_cast<uint2 *>(row)[kTileBins / 2] = make_uint2(base_c + total, base_c + total);
                ex.run_base[b] = base_c;
            }
        }
        __syncthreads();
        // pass 2: every key takes the next free slot of its tile's piece (one returning LDS atomic; the order inside a piece is arbitrary --
        // the per-tile sort orders (depth, value) composites, a total order)
#pragma unroll
        for (int it = 0; it < KEEP; it++) {
            const uint32_t j = threadIdx.x + (uint32_t)it * NT;
            if (j < total) {
                const uint32_t dst = block_base + atomicAdd(&s_th[k_tile[it]], 1u);
                if (dst < cap) keys[dst] = ((uint64_t)s_dep[k_lo[it]] << 32) | (q0 + k_lo[it]);
            }
        }
        for (uint32_t j = threadIdx.x + KEEP * NT; j < total; j += NT) {
            uint32_t lo, tile_id;
            slot_of(j, lo, tile_id);
            const uint32_t dst = block_base + atomicAdd(&s_th[tile_id], 1u);
            if (dst < cap) keys[dst] = ((uint64_t)s_dep[lo] << 32) | (q0 + lo);
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

// ---- F4 + F5, VIEW-SEGMENTED flavour for multi-view batches and large launches -----------------------------------
// The emission is view-major (duplicate_keys: blockIdx.y = view, offsets from the scan of the per-block counts), so the view bits of
// the key are sorted before the sort starts: what remains is, per view, a sort by (tile-in-view, depth).  ONE counting pass per view
// over the tile id (<= 4096 tiles per view: 1024^2 images) puts every tile's instances into one contiguous segment -- as
// (depth bits << 32 | value) composites, in any order -- and yields the tile ranges (F5) and the worklists of occupied tiles as
// by-products of its scan; the composites are then sorted per tile in registers (tile_sort_regs_kernel).  Per key: 8 B (histogram)
// + 20 B (scatter) + 20 B (per-tile sort) of HBM traffic instead of 6-7 whole-key passes of 32 B.
//   vseg_view_totals -> vseg_plan   per-view key ranges from the per-block emission counts; the keys are cut into CHUNKS of
//                                   256 * ITEMS keys that never straddle a view: chunk_map[c] = (view, first key, count)
//   vseg_upsweep                    per-chunk tile histogram, hist[c][tile]
//   vseg_scan (one workgroup/view)  hist[c][tile] <- keys of that tile in earlier chunks of the view; tile totals -> ranges, worklists
//   vseg_scatter                    every key claims the next slot of its tile with one returning LDS atomic (no stability needed: the
//                                   per-tile sort orders (depth, value) composites, and the value grows with the emission order)
// All sizes come from device memory (sync-free mode: the host only knows the capacity).
constexpr int kVsegMaxViews = 4096, kVsegMaxBins = 4096;

__global__ __launch_bounds__(kThreads) void vseg_view_totals_kernel(const uint32_t *__restrict__ sums, uint32_t nbx,
                                                                    unsigned long long *__restrict__ view_total) {
    __shared__ unsigned long long red[4];
    const uint32_t *row = sums + (size_t)blockIdx.x * nbx;
    unsigned long long acc = 0;
    for (uint32_t k = threadIdx.x; k < nbx; k += kThreads) acc += row[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) view_total[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup of 1024 threads; n_views <= 4096
__global__ __launch_bounds__(1024) void vseg_plan_kernel(const unsigned long long *__restrict__ view_total, uint32_t n_views, uint32_t cap,
                                                         const uint64_t *__restrict__ n_dev, uint32_t chunk_keys, uint32_t max_chunks,
                                                         VsegPlan *__restrict__ plan, uint32_t *__restrict__ view_key_start,
                                                         uint32_t *__restrict__ view_chunk_start, uint4 *__restrict__ chunk_map,
                                                         const uint32_t *__restrict__ sums /* NULL, or the per-block emission counts [n_views][nbx] of
                                                         a launch with <= kVsegFoldViews views and <= 65 536 counts: the view totals are then summed here (one launch fewer) */,
                                                         uint32_t nbx) {
    __shared__ uint32_t s_key[kVsegMaxViews + 1], s_chunk[kVsegMaxViews + 1];
    __shared__ unsigned long long s_wave[16], s_vtot[kVsegFoldViews];
    __shared__ uint32_t s_wave32[16];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (sums) {
        // one wave per view (views w, w + 16, ..): no workgroup barrier per view
        for (uint32_t v = wave; v < n_views; v += 16u) {
            const uint32_t *row = sums + (size_t)v * nbx;
            unsigned long long acc = 0;
            for (uint32_t k = lane; k < nbx; k += 64u) acc += row[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) s_vtot[v] = acc;
        }
        __syncthreads();
    }
    const unsigned long long n_true = n_dev ? min((unsigned long long)cap, (unsigned long long)*n_dev) : (unsigned long long)cap;
    // ---- exclusive scan of the view totals (4 consecutive views per thread), clamped to the keys that exist in the buffers
    unsigned long long tot[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { const uint32_t v = t * 4 + j; tot[j] = v < n_views ? (sums ? s_vtot[v] : view_total[v]) : 0ull; sum += tot[j]; }
    unsigned long long inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned long long nb = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += nb; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long run = inc - sum;
    for (uint32_t w = 0; w < wave; w++) run += s_wave[w];
    uint32_t nch[4], csum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t v = t * 4 + j;
        const unsigned long long a = min(run, n_true), b = min(run + tot[j], n_true);
        run += tot[j];
        nch[j] = (uint32_t)((b - a + chunk_keys - 1) / chunk_keys);
        csum += nch[j];
        if (v < n_views) s_key[v] = (uint32_t)a;
        if (v + 1 == n_views) s_key[n_views] = (uint32_t)b;
    }
    uint32_t cinc = csum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t nb = __shfl_up(cinc, off, 64); if (lane >= (uint32_t)off) cinc += nb; }
    if (lane == 63) s_wave32[wave] = cinc;
    __syncthreads();
    uint32_t crun = cinc - csum;
    for (uint32_t w = 0; w < wave; w++) crun += s_wave32[w];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t v = t * 4 + j;
        if (v < n_views) s_chunk[v] = crun;
        crun += nch[j];
        if (v + 1 == n_views) s_chunk[n_views] = crun;
    }
    __syncthreads();
    const uint32_t n_chunks = min(s_chunk[n_views], max_chunks);
    if (t == 0) plan->n_chunks = n_chunks;
    if (t < 8) { plan->count[t] = 0; plan->ticket[t] = 0; }
    for (uint32_t v = t; v <= n_views; v += 1024) { view_key_start[v] = s_key[v]; view_chunk_start[v] = min(s_chunk[v], max_chunks); }
    // ---- chunk map: chunk c belongs to the last view whose first chunk is <= c
    for (uint32_t c = t; c < n_chunks; c += 1024) {
        uint32_t lo = 0, hi = n_views;                       // invariant: s_chunk[lo] <= c < s_chunk[hi]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_chunk[mid] <= c) lo = mid; else hi = mid; }
        const uint32_t k0 = s_key[lo] + (c - s_chunk[lo]) * chunk_keys;
        chunk_map[c] = make_uint4(lo, k0, min(chunk_keys, s_key[lo + 1] - k0), 0u);
    }
}

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void vseg_upsweep_kernel(const uint64_t *__restrict__ keys, const VsegPlan *__restrict__ plan,
                                                                const uint4 *__restrict__ chunk_map, uint32_t tiles_per_view,
                                                                uint32_t *__restrict__ hist) {
    __shared__ __attribute__((aligned(16))) uint32_t h[kVsegMaxBins];
    const uint32_t c = blockIdx.x;
    if (c >= plan->n_chunks) return;
    const uint4 cm = chunk_map[c];
    // the chunk's tile ids first, ALL in flight at once (indices past the end re-read the last key): with
    // the load inside the guarded loop below every one of the 32 was a branch, a load and a wait of its own -- 32 round trips per thread
    uint32_t tl[ITEMS];
    {
        const uint32_t last = cm.z ? cm.z - 1u : 0u;
        if (cm.z) {                                                // (workgroup-uniform)
#pragma unroll
            for (int it = 0; it < ITEMS; it++) tl[it] = (uint32_t)(keys[cm.y + min((uint32_t)it * kThreads + threadIdx.x, last)] >> 32);
        }
    }
    const bool vec = (tiles_per_view & 3u) == 0u;                 // rows of the histogram are 16-byte aligned: 16-byte LDS and global accesses
    if (vec) for (uint32_t d = threadIdx.x; d < tiles_per_view / 4u; d += kThreads) reinterpret_cast<uint4 *>(h)[d] = make_uint4(0u, 0u, 0u, 0u);
    else for (uint32_t d = threadIdx.x; d < tiles_per_view; d += kThreads) h[d] = 0;
    __syncthreads();
    const uint32_t tbase = cm.x * tiles_per_view;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = it * kThreads + threadIdx.x;
        if (k < cm.z) atomicAdd(&h[tl[it] - tbase], 1u);
    }
    __syncthreads();
    uint32_t *out = hist + (size_t)c * tiles_per_view;
    // (a quarter of the store requests: the write path prices requests, not bytes)
    if (vec) for (uint32_t d = threadIdx.x; d < tiles_per_view / 4u; d += kThreads) reinterpret_cast<uint4 *>(out)[d] = reinterpret_cast<const uint4 *>(h)[d];
    else for (uint32_t d = threadIdx.x; d < tiles_per_view; d += kThreads) out[d] = h[d];
}

// column scan: workgroup (view, slab of 256 tiles): per tile, exclusive prefix over the view's chunks (in place) and the tile total
__global__ __launch_bounds__(kThreads) void vseg_colscan_kernel(uint32_t *__restrict__ hist, const uint32_t *__restrict__ view_chunk_start,
                                                                uint32_t tiles_per_view, uint32_t *__restrict__ tile_total) {
    const uint32_t v = blockIdx.y, d = blockIdx.x * kThreads + threadIdx.x;
    if (d >= tiles_per_view) return;
    const uint32_t c0 = view_chunk_start[v], c1 = view_chunk_start[v + 1];
    uint32_t *col = hist + (size_t)c0 * tiles_per_view + d;
    uint32_t run = 0, c = c0;
    for (; c + 8 <= c1; c += 8, col += 8 * (size_t)tiles_per_view) {          // 8 independent loads in flight per thread
        uint32_t x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = col[(size_t)j * tiles_per_view];
#pragma unroll
        for (int j = 0; j < 8; j++) { col[(size_t)j * tiles_per_view] = run; run += x[j]; }
    }
    for (; c < c1; c++, col += tiles_per_view) { const uint32_t x = *col; *col = run; run += x; }
    tile_total[(size_t)v * tiles_per_view + d] = run;
}

// one workgroup (1024 threads) per view: exclusive scan of the tile totals in tile order -> ranges (F5) + the worklists of occupied
// tiles by size class for the per-tile depth sort
__global__ __launch_bounds__(1024) void vseg_scan_kernel(const uint32_t *__restrict__ tile_total, const uint32_t *__restrict__ view_key_start,
                                                         uint32_t tiles_per_view, uint2 *__restrict__ ranges, VsegPlan *__restrict__ plan,
                                                         uint32_t *__restrict__ lists /*[8][list_stride]: one worklist per size class + the deep kernels' two*/,
                                                         uint32_t list_stride, uint32_t deep_min /* tiles of more entries go to the deep kernels' lists (6, 7); 0xFFFFFFFF: none */,
                                                         uint32_t small_max /* ... of which those up to this many to the small instantiation's (7) */) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry, s_cnt[8], s_base[8];
    const uint32_t v = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = view_key_start[v];
    for (uint32_t d0 = 0; d0 < tiles_per_view; d0 += 1024) {
        const uint32_t d = d0 + t;
        const uint32_t run = d < tiles_per_view ? tile_total[(size_t)v * tiles_per_view + d] : 0u;
        uint32_t inc = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t nb = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += nb; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t start = s_carry + inc - run;
        uint32_t all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++) { const uint32_t x = s_wave[w]; if (w < wave) start += x; all += x; }
        // worklists: ranks inside the workgroup from LDS counters, ONE global atomic per class and 1024 tiles (a returning global atomic
        // per tile serialises on four addresses: 52 000 of them cost 0.46 ms at C4)
        if (t < 8) s_cnt[t] = 0;
        __syncthreads();
        uint32_t cls = 8, local = 0;
        if (d < tiles_per_view) {
            ranges[v * tiles_per_view + d] = run ? make_uint2(start, start + run) : make_uint2(0u, 0u);
            if (run) {                                                   // (single-key tiles too: the sort kernel moves them to the destination buffer)
                cls = run <= 1024u ? 0u : (run <= 2048u ? 1u : (run <= 4096u ? 2u : (run <= 8192u ? 3u : (run <= 16384u ? 4u : 5u))));
                uint32_t take = 1u;
                if (run > deep_min && run <= kDeepMaxN) {                            // -> the deep kernels: small tiles on list 7, big ones on list 6,
                    if (run <= small_max) cls = 7u;
                    else { cls = 6u; take = (run + (kDeepBigCap - kDeepBinMax) - 1u) / (kDeepBigCap - kDeepBinMax); }
                }
                local = atomicAdd(&s_cnt[cls], take);
            }
        }
        __syncthreads();
        if (t < 8 && s_cnt[t]) s_base[t] = atomicAdd(&plan->count[t], s_cnt[t]);
        __syncthreads();
        if (cls < 8) {
            const uint32_t id = v * tiles_per_view + d;
            uint32_t *dst = lists + (size_t)cls * list_stride + s_base[cls] + local;
            dst[0] = id;
            if (cls == 6u) { const uint32_t nw = (run + (kDeepBigCap - kDeepBinMax) - 1u) / (kDeepBigCap - kDeepBinMax); for (uint32_t w = 1; w < nw; w++) dst[w] = id | (w << 26); }
        }
        __syncthreads();
        if (t == 0) s_carry += all;
        __syncthreads();
    }
}

// scatter: the per-tile sort orders (depth, value) composites, so the tile pass needs NO stability -- a key can take ANY free slot of its
// tile's segment (the stable version kept a counter array per wave and match-any ballots: 64 KB of LDS at 4096 tiles per view, 0.85 ms
// at C4).  The first order-free version stored every composite straight into its segment (one returning LDS atomic on a 16-KB
// counter array per key, then an 8-byte store): fine for 1024 tiles per view, but a chunk of 8192 keys of a 4096-tile view touches ~600
// segments with ~14 keys each, and those scattered stores cost 2.5x their bytes in HBM write requests (the per-XCD L2s cannot keep
// that many partially written lines open).  So the workgroup first ORDERS its chunk by tile in LDS -- count per tile, exclusive scan, one
// returning LDS atomic per key for its place in the staged chunk -- and then copies the staged chunk out: consecutive threads store
// consecutive composites of a tile's run, a wave's store covers a handful of runs instead of 64 unrelated lines (C4: 0.40 -> 0.27 ms,
// C3: 0.11 -> 0.09 ms).  Chunks are dealt to the XCDs in contiguous eighths (neighbouring chunks' runs meet in one L2).
template <int MAXB, int ITEMS>
__global__ __launch_bounds__(1024) void vseg_scatter_staged_kernel(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                   uint64_t *__restrict__ keys_out, const VsegPlan *__restrict__ plan,
                                                                   const uint4 *__restrict__ chunk_map, uint32_t tiles_per_view,
                                                                   const uint32_t *__restrict__ hist, const uint2 *__restrict__ ranges) {
    constexpr int NT = 1024, PER = MAXB / NT;                        // chunk = 1024 * ITEMS keys; PER consecutive tiles per thread in the scan
    __shared__ uint64_t comp[NT * ITEMS];                            // the chunk, ordered by tile
    __shared__ uint32_t run[MAXB];                                   // per tile: count -> next free staged slot
    __shared__ uint32_t delta[MAXB];                                 // per tile: (first output slot of this chunk's keys) - (first staged slot)
    __shared__ uint16_t tl[NT * ITEMS];                              // tile (in view) of staged entry j
    __shared__ uint32_t wsum[16];
    const uint32_t n_chunks = plan->n_chunks, span = (n_chunks + 7u) >> 3;
    const uint32_t c = (blockIdx.x & 7u) * span + (blockIdx.x >> 3);          // XCD x takes the x-th eighth of the chunk list
    if ((blockIdx.x >> 3) >= span || c >= n_chunks) return;
    const uint4 cm = chunk_map[c];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t tbase = cm.x * tiles_per_view;
    uint64_t key[ITEMS];
    uint32_t val[ITEMS];
    const uint32_t last = cm.z - 1u;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = min((uint32_t)it * NT + t, last);
        key[it] = keys_in[cm.y + k]; val[it] = vals_in[cm.y + k];
    }
    for (uint32_t d = t; d < (uint32_t)MAXB; d += NT) run[d] = 0u;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++)
        if ((uint32_t)it * NT + t < cm.z) atomicAdd(&run[(uint32_t)(key[it] >> 32) - tbase], 1u);
    __syncthreads();
    // exclusive scan of the MAXB counts (PER consecutive tiles per thread)
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { v[j] = run[t * PER + j]; sum += v[j]; }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t nb = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += nb; }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - sum;
    for (uint32_t w = 0; w < wave; w++) base += wsum[w];
    const uint32_t *hrow = hist + (size_t)c * tiles_per_view;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t d = t * PER + j;
        run[d] = base;
        if (d < tiles_per_view && v[j]) delta[d] = ranges[tbase + d].x + hrow[d] - base;
        base += v[j];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITEMS; it++)
        if ((uint32_t)it * NT + t < cm.z) {
            const uint32_t d = (uint32_t)(key[it] >> 32) - tbase;
            const uint32_t ls = atomicAdd(&run[d], 1u);
            comp[ls] = (key[it] << 32) | val[it];
            tl[ls] = (uint16_t)d;
        }
    __syncthreads();
    for (uint32_t j = t; j < cm.z; j += NT) keys_out[j + delta[tl[j]]] = comp[j];
}

// column scan for views with MANY chunks (one view of a million Gaussians = 540 chunks: the kernel above walks them one thread per tile,
// 70 dependent round trips): workgroup (view, slab of 64 tiles), lane = tile (a chunk's 64 counts are one 256-byte run), wave w owns the
// chunks [w * per, (w + 1) * per) of the view: sums them, the 16 partial sums per tile meet in LDS, then it rewrites its chunks with the
// running prefix.  Two reads + one write per count, ~2 x (per / 8) dependent round trips.
__global__ __launch_bounds__(1024) void vseg_colscan_par_kernel(uint32_t *__restrict__ hist, const uint32_t *__restrict__ view_chunk_start,
                                                                uint32_t tiles_per_view, uint32_t *__restrict__ tile_total) {
    __shared__ uint32_t part[16][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, v = blockIdx.y;
    const uint32_t d = min(blockIdx.x * 64u + lane, tiles_per_view - 1u);       // (clamped: lanes past the last tile redo it and store nothing)
    const bool live = blockIdx.x * 64u + lane < tiles_per_view;
    const uint32_t c0 = view_chunk_start[v], c1 = view_chunk_start[v + 1], nc = c1 - c0;
    const uint32_t per = (nc + 15u) / 16u, a = c0 + min(nc, wave * per), b = min(c1, a + per);
    uint32_t *col = hist + (size_t)a * tiles_per_view + d;
    uint32_t sum = 0, c = a;
    for (; c + 8 <= b; c += 8) {
        uint32_t x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = col[(size_t)(c - a + j) * tiles_per_view];
#pragma unroll
        for (int j = 0; j < 8; j++) sum += x[j];
    }
    for (; c < b; c++) sum += col[(size_t)(c - a) * tiles_per_view];
    part[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) { const uint32_t x = part[w][lane]; if (w < wave) run += x; all += x; }
    if (live) {
        for (c = a; c + 8 <= b; c += 8) {
            uint32_t x[8];
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = col[(size_t)(c - a + j) * tiles_per_view];
#pragma unroll
            for (int j = 0; j < 8; j++) { col[(size_t)(c - a + j) * tiles_per_view] = run; run += x[j]; }
        }
        for (; c < b; c++) { const uint32_t x = col[(size_t)(c - a) * tiles_per_view]; col[(size_t)(c - a) * tiles_per_view] = run; run += x; }
        if (wave == 0) tile_total[(size_t)v * tiles_per_view + d] = all;
    }
}

// ---- deep launches of one or two views (C5: a million Gaussians on 1 024 tiles): the tile pass as ONE launch --------------------------------
// The emission kernel writes tile-ordered runs and the run matrix as on the single-view path (duplicate_keys_kernel<true, ..>, kRunRow); one
// workgroup per tile then reads its columns of the matrix -- the pieces of its list in the runs and, summed, its range --, copies the pieces
// into the tile's contiguous segment, writes the range (F5) and enters the tile into the per-tile sorts' worklists (exactly what
// vseg_scan_kernel enters: register-sort classes 0..5, the deep kernels' lists 6 / 7, one entry per window of a big tile).  Replaces the
// view-segmented flavour's plan / upsweep / column scan / tile scan / staged scatter launches (five dependent launches, 39 us of C5's 360).
template <int NT>
__global__ __launch_bounds__(NT) void tile_collect_kernel(const uint64_t *__restrict__ runs, uint64_t *__restrict__ dst, GatherFront gf, VsegPlan *__restrict__ plan,
                                                          uint32_t *__restrict__ lists, uint32_t list_stride, uint32_t deep_min, uint32_t small_max) {
    constexpr uint32_t NW = NT / 64;
    __shared__ uint32_t s_pre[NT], s_cur[NT], s_wave[NW], s_wave2[NW];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, i = blockIdx.x;
    uint32_t tile;
    tile = sgr_tile_of_workgroup(i, gf.tx, gf.ty);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)gf.occ[tile]) == 0u) { if (t == 0) gf.ranges[tile] = make_uint2(0u, 0u); return; }
    uint32_t a = 0u, e = 0u, bs = 0u;
    if (t < gf.nblk) { const uint32_t *r = gf.rows + (size_t)t * kRunRow + tile; a = r[0]; e = r[1]; bs = gf.base[t]; }
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
    s_pre[t] = pre + inc - cnt; s_cur[t] = a;                                    // piece prefix / piece start (runs beyond nblk: prefix = n)
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_all), x0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
    if (t == 0) {
        gf.ranges[tile] = n ? make_uint2(x0, x0 + n) : make_uint2(0u, 0u);
        if (n) {                                                                 // (single-key tiles too: the sort kernels move them to the destination buffer)
            uint32_t cls = n <= 1024u ? 0u : (n <= 2048u ? 1u : (n <= 4096u ? 2u : (n <= 8192u ? 3u : (n <= 16384u ? 4u : 5u))));
            uint32_t take = 1u;
            if (n > deep_min && n <= kDeepMaxN) {
                if (n <= small_max) cls = 7u;
                else { cls = 6u; take = (n + (kDeepBigCap - kDeepBinMax) - 1u) / (kDeepBigCap - kDeepBinMax); }
            }
            uint32_t *d = lists + (size_t)cls * list_stride + atomicAdd(&plan->count[cls], take);
            d[0] = tile;
            for (uint32_t w = 1; w < take; w++) d[w] = tile | (w << 26);
        }
    }
    __syncthreads();
    // the copy: one thread per composite, which finds its piece by binary search over the piece prefixes -- eight composites per trip, their
    // searches in lock-step (eight independent LDS reads per level) and their loads in flight together.  (Piece by piece -- a wave per piece,
    // contiguous loads and stores, no search -- was SLOWER: 42 against 31 us at C5; a piece holds ~30 composites, half a wave's lanes idle and
    // twice the load instructions.)
    uint64_t *out = dst + x0;
    constexpr uint32_t GK = 8;
    for (uint32_t j0 = t; j0 < n; j0 += GK * NT) {
        uint32_t j[GK], lo[GK];
        uint64_t v[GK];
#pragma unroll
        for (uint32_t u = 0; u < GK; u++) { j[u] = min(j0 + u * NT, n - 1u); lo[u] = 0u; }
        for (uint32_t step = gf.search_top; step > 0u; step >>= 1) {
#pragma unroll
            for (uint32_t u = 0; u < GK; u++) if (s_pre[lo[u] + step] <= j[u]) lo[u] += step;
        }
#pragma unroll
        for (uint32_t u = 0; u < GK; u++) v[u] = runs[s_cur[lo[u]] + (j[u] - s_pre[lo[u]])];
#pragma unroll
        for (uint32_t u = 0; u < GK; u++) if (j0 + u * NT < n) out[j0 + u * NT] = v[u];
    }
}

// ---- F5 -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void tile_ranges_kernel(const uint64_t *__restrict__ keys, uint32_t n_host,
                                                               const uint64_t *__restrict__ n_dev, uint2 *__restrict__ ranges) {
    const uint32_t n = n_dev ? (uint32_t)min((uint64_t)n_host, *n_dev) : n_host;
    const uint32_t r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= n) return;
    const uint32_t tile = (uint32_t)(keys[r] >> 32);
    if (r == 0) ranges[tile].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[r - 1] >> 32);
        if (tile != prev) { ranges[prev].y = r; ranges[tile].x = r; }
    }
    if (r == n - 1) ranges[tile].y = n;
}

// 3 = automatic (default), 5 = the single-view path: tile-ordered emission runs + one workgroup per tile that gathers and sorts, no tile pass (one or two 512^2 views; else like 3),
// 4 = view-segmented (per-view tile pass + per-tile depth sort), 1 = three kernels per 8-bit digit over the whole key (the fallback)
static int sgr_sort_mode_from_env() { const int v = sgr_env_knob("SIGMAN_SORT_MODE", 1, 5, 3); return v == 2 ? 3 : v; }     // (2 was removed in round 5: as refused as by the setter)
thread_local int sgr_sort_mode = sgr_sort_mode_from_env();        // (per thread; every thread starts from the environment)

struct VsegLayout {
    size_t plan, totals, key_start, chunk_start, chunk_map, hist, tile_total, lists, end; uint32_t chunk_keys, max_chunks;
    uint32_t list_stride;
};
inline VsegLayout vseg_layout(uint64_t R, uint64_t tiles_total, uint32_t n_views, uint32_t tiles_per_view, bool split = false) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    VsegLayout L;
    L.chunk_keys = tiles_per_view > 1024 ? 8192u : 4096u;                    // = 256 * ITEMS of the kernels instantiated below
    L.max_chunks = (uint32_t)(R / L.chunk_keys + n_views + 1);
    size_t o = 0;
    L.plan = o; o = al(o + sizeof(VsegPlan));
    L.totals = o; o = al(o + (size_t)n_views * 8);
    L.key_start = o; o = al(o + ((size_t)n_views + 1) * 4);
    L.chunk_start = o; o = al(o + ((size_t)n_views + 1) * 4);
    L.chunk_map = o; o = al(o + (size_t)L.max_chunks * 16);
    L.hist = o; o = al(o + (size_t)L.max_chunks * tiles_per_view * 4);
    L.tile_total = o; o = al(o + (size_t)tiles_total * 4);
    // deep mode: list 6 holds one entry per window of a big tile (<= R / (kDeepBigCap - kDeepBinMax) + tiles_total entries)
    L.list_stride = (uint32_t)tiles_total + (split ? (uint32_t)(R / (kDeepBigCap - kDeepBinMax)) + 1u : 0u);
    L.lists = o; o = al(o + (size_t)L.list_stride * 4 * 8);
    L.end = o;
    return L;
}

inline int bits_for(uint64_t v) { int b = 0; while ((1ull << b) < v) b++; return b; }   // smallest b with 2^b >= v

}  // namespace

int sgr_validate_problem(const SgrProblem *pb);

extern "C" int sgr_set_sort_mode(int mode) {
    if (mode != 1 && mode != 3 && mode != 4 && mode != 5) { sgr_set_error("sgr_set_sort_mode: %d is not a sort flavour (3 automatic, 5 single-view path, 4 view-segmented, 1 whole-key passes)", mode); return 1; }
    sgr_sort_mode = mode;
    return 0;
}
// deep tile lists in the view-segmented flavour (the LDS distribution sort, deep_tile_kernel): 0 = automatic (launches whose tile lists are
// deep on average), 1 = whenever that flavour runs, 2 = never (deep launches then keep the whole-key passes)
static thread_local int g_deep_mode = sgr_env_knob("SIGMAN_SORT_DEEP", 0, 2, 0);
// bits 8..15 of `mode` (tests): the most windows a tile may have on the single-view path before it is sorted whole by the stable radix passes
// (0 = the window field's 64)
static thread_local uint32_t g_deep_max_windows = kDeepMaxWindows;
static thread_local int g_collect_mode = sgr_env_knob("SIGMAN_SORT_COLLECT", 0, 1, 1);      // A/B: 0 = deep launches of one or two views keep the five-launch tile pass
extern "C" int sgr_set_sort_deep(int mode) {
    const int m = mode & 0xFF, cap = (mode >> 8) & 0xFF;
    g_deep_mode = (m >= 0 && m <= 2) ? m : 0;
    g_deep_max_windows = (cap >= 1 && cap <= (int)kDeepMaxWindows) ? (uint32_t)cap : kDeepMaxWindows;
    const int col = (mode >> 16) & 0xF;                       // bits 16..19 (tests): SIGMAN_SORT_COLLECT's value + 1; 0 = the environment's
    g_collect_mode = (col >= 1 && col <= 2) ? col - 1 : sgr_env_knob("SIGMAN_SORT_COLLECT", 0, 1, 1);
    return 0;
}

// the single-view path's share of the workspace: the run matrix, the runs' bases, the scratch composites of tiles that go through global memory
struct RunsLayout { size_t occ, rows, base, scratch_k, end; };
inline RunsLayout runs_layout(uint64_t R, uint32_t nblk) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    RunsLayout L;
    L.occ = 0;                                   // (at the head of the workspace: rasterize.hip has the preprocess launch zero it)
    L.rows = al((size_t)kTileBins * sizeof(uint32_t));
    L.base = al(L.rows + (size_t)nblk * kRunRow * sizeof(uint32_t));
    L.scratch_k = al(L.base + (size_t)nblk * sizeof(uint32_t));
    L.end = al(L.scratch_k + (size_t)R * sizeof(uint64_t));
    return L;
}

extern "C" size_t sgr_bin_workspace_bytes(uint64_t R, uint64_t tiles_total) {
    const uint64_t nblocks = (R + kThreads * kItemsSmall - 1) / (kThreads * kItemsSmall);
    // whole-key passes: [hist blocks x 256][totals 256]; the view-segmented flavour lays its plan / chunk map / histograms / worklists
    // over the whole area from the start (<= 2 R + R / 256 + 100 B per tile + 64 KB: see vseg_layout); the single-view path its run matrix
    // (<= 512 rows of 8 KB) and 8 B per instance of scratch (runs_layout)
    const size_t a = (size_t)((8 * (nblocks > 0 ? nblocks : 1) + 8 + 2) * kRadix * sizeof(uint32_t) + 1024 + ((size_t)4 << 20) +
                              (tiles_total ? (tiles_total + 64) * sizeof(uint32_t) + tiles_total * 128 + 65536 : 0));
    const size_t b = (tiles_total && tiles_total <= (uint64_t)kTileBins && R <= (1u << 19)) ? runs_layout(R, 512u).end : 0;
    // (deep launches of one or two views: the run matrix of <= 1024 emission workgroups in front of the view-segmented flavour's own layout)
    const size_t c = (tiles_total && tiles_total <= (uint64_t)kTileBins) ? runs_layout(0, 1024u).end : 0;
    return std::max(a + c, b);
}

// self_scan: the caller skipped the F2 scan kernel (sgr_preprocess_forward_ex) and block_offsets + n + 1 holds the un-scanned
// counts; num_rendered_dev is then WRITTEN by the duplicate kernel (capacity mode only).  nr_host: optional pinned host slot.
// fwd_order (optional, the segment-parallel forward's work order, SGR_ORDER_HDR_WORDS + 33 * tiles_total * 4 words): filled by the single-view
// path in its class-major form (*order_kind_out = 1; the workgroups of empty tiles then also write those tiles' background, `bg`); every other
// flavour with a per-tile register sort launch (view-segmented) fills it in the plain form -- one uint4 per slot, longest lists first, empty tiles
// last -- by that launch's spare workgroup (*order_kind_out = 2); else *order_kind_out = 0.  clear_ptr / clear_words / clear_done [3]: buffers
// to zero on the side of the emission kernel.
int sgr_bin_ex(const SgrProblem *pb, const int32_t *radii, uint32_t *rect,
               const uint32_t *block_offsets, uint64_t R, const uint64_t *num_rendered_dev, uint64_t *keys_a,
               uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *workspace, size_t workspace_bytes,
               uint32_t *ranges, int32_t *result_in_b_host, bool self_scan, uint64_t *nr_host, uint32_t *fwd_order, int *order_kind_out,
               const SgrBgJob *bg, bool occ_zeroed /* the first SGR_BIN_OCC_WORDS words of the workspace are zero (else: a memset launch on the
               single-view path) */, uint32_t *const *clear_ptr /*[3] or NULL*/, const uint64_t *clear_words /*[3]*/,
               int *clear_done /*[3]*/, bool first_index, bool sorted_keys, void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint64_t tiles_total = (uint64_t)Tx * Ty * pb->n_views;
    if (tiles_total >= (1ull << 32)) { sgr_set_error("too many tiles (%llu)", (unsigned long long)tiles_total); return 1; }
    if (R > 0xFFFFFFF0ull) { sgr_set_error("num_rendered %llu exceeds the 32-bit instance index", (unsigned long long)R); return 1; }
    if (result_in_b_host) *result_in_b_host = 0;
    if (order_kind_out) *order_kind_out = 0;
    if (clear_done) clear_done[0] = clear_done[1] = clear_done[2] = 0;
    if (R == 0 || pb->P == 0) { SGR_CHECK_HIP(hipMemsetAsync(ranges, 0, tiles_total * 2 * sizeof(uint32_t), stream)); return 0; }
    if (workspace_bytes < sgr_bin_workspace_bytes(R, tiles_total)) { sgr_set_error("sgr_bin: workspace too small"); return 1; }
    const uint32_t n = (uint32_t)R;
    const int nbx = sgr_preprocess_blocks_per_view(pb->P);
    const bool small = n <= (1u << 19);
    const uint32_t tile_keys = kThreads * (small ? kItemsSmall : kItemsLarge);
    const uint32_t nblocks = (n + tile_keys - 1) / tile_keys;
    uint32_t *hist = (uint32_t *)workspace;
    uint32_t *totals = hist + (size_t)nblocks * kRadix;
    const int total_bits = 32 + bits_for(tiles_total);
    const int passes = (total_bits + kRadixBits - 1) / kRadixBits;
    uint64_t *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    // automatic: one or two 512^2 views (<= 2048 tiles, <= 2^19 instances, <= 512 emission workgroups): the single-view path -- tile-ordered
    // emission runs + one workgroup per tile that gathers and sorts (no tile pass); everything else with <= 4096 tiles per view:
    // view-segmented; beyond that the whole-key passes
    const uint32_t tpv = (uint32_t)Tx * (uint32_t)Ty;
    // launches whose tile lists are deep on average (C5: 1M Gaussians on 1024 tiles): the view-segmented flavour hands its long tiles to the
    // LDS distribution sort (deep_tile_kernel) instead of the register network; without room for its worklists they keep the whole-key passes
    const bool deep = R > tiles_total * 1024ull;
    const bool want_deep = g_deep_mode == 1 || (g_deep_mode == 0 && deep);
    VsegLayout VL = vseg_layout(R, tiles_total, (uint32_t)pb->n_views, tpv, want_deep);
    bool split = want_deep;
    if (split && VL.end > workspace_bytes) { split = false; VL = vseg_layout(R, tiles_total, (uint32_t)pb->n_views, tpv, false); }
    const bool vseg_ok = tpv <= (uint32_t)kVsegMaxBins && pb->n_views <= kVsegMaxViews && VL.end <= workspace_bytes;
    // (launches of few Gaussians -- C1: 10 000 large splats -- emit with 256-thread workgroups: ten 1024-thread ones leave 246 CUs idle while
    // each walks 30 000 keys through one CU's LDS)
    const int run_nt = ((int64_t)pb->P * pb->n_views <= 32768) ? kThreads : kRunThreads;
    const uint32_t nbx_e = (uint32_t)(nbx + run_nt / kThreads - 1) / (uint32_t)(run_nt / kThreads);     // emission workgroups per view
    const uint32_t nblk_e = nbx_e * (uint32_t)pb->n_views;
    const RunsLayout RL = runs_layout(R, nblk_e);
    const bool runs_ok = tiles_total <= (uint64_t)kTileBins && small && nblk_e <= 512u && RL.end <= workspace_bytes;
    int mode = sgr_sort_mode;
    if (mode == 3 || mode == 5) mode = runs_ok ? 5 : 4;
    if (mode == 4 && !(vseg_ok && (!deep || split || sgr_sort_mode == 4))) mode = 1;
    if (mode == 4 && !vseg_ok) mode = 1;
    // deep launches of one or two views (C5) in the view-segmented flavour: the collect form of the tile pass (tile_collect_kernel)
    const RunsLayout RC = runs_layout(0, nblk_e);
    const bool collect = mode == 4 && split && tiles_total <= (uint64_t)kTileBins && nblk_e <= 1024u && RC.end + VL.end <= workspace_bytes && g_collect_mode != 0;
    const bool runs = mode == 5 || collect;
    // (every tile's range is written by its own workgroup on the single-view path; the other flavours write the occupied tiles' only)
    const bool fold_clear = !runs && tiles_total * 2 <= (1u << 20);     // small: cleared by the duplicate kernel
    if (!runs && !fold_clear) SGR_CHECK_HIP(hipMemsetAsync(ranges, 0, tiles_total * 2 * sizeof(uint32_t), stream));
    { SgrProfScope _p(SGR_K_DUPLICATE, stream);
    DupExtra ex;
    const RunsLayout &RU = collect ? RC : RL;
    ex.run_rows = runs ? (uint32_t *)((char *)workspace + RU.rows) : nullptr; ex.run_base = runs ? (uint32_t *)((char *)workspace + RU.base) : nullptr;
    ex.occ = runs ? (uint32_t *)((char *)workspace + RU.occ) : nullptr;
    if (runs && !occ_zeroed) SGR_CHECK_HIP(hipMemsetAsync(ex.occ, 0, (size_t)kTileBins * sizeof(uint32_t), stream));
    ex.write_first = first_index ? 1u : 0u;
    const uint32_t nblk = (uint32_t)nbx * (uint32_t)pb->n_views;
    ex.self_sums = self_scan ? block_offsets + (nblk + 1) : nullptr;
    ex.num_rendered = const_cast<uint64_t *>(num_rendered_dev); ex.nr_host = self_scan ? nr_host : nullptr; ex.capacity = R;
    ex.zero_ptr[0] = fold_clear ? ranges : nullptr; ex.zero_words[0] = fold_clear ? (uint32_t)(tiles_total * 2) : 0u;
    for (int c = 0; c < 3; c++) {
        const bool ok = clear_ptr && clear_ptr[c] && clear_words && clear_words[c] <= (1ull << 26);
        ex.zero_ptr[1 + c] = ok ? clear_ptr[c] : nullptr; ex.zero_words[1 + c] = ok ? (uint32_t)clear_words[c] : 0u;
        if (clear_done) clear_done[c] = ok ? 1 : 0;
    }
    ex.zero_small = (mode == 5 && fwd_order) ? fwd_order : nullptr; ex.zero_small_n = 48u;       // the work order's class counters
    if (collect) { ex.zero_small = (uint32_t *)((char *)workspace + RC.end + VL.plan); ex.zero_small_n = (uint32_t)(sizeof(VsegPlan) / 4); }   // the worklist counters and tickets
    if (self_scan && !num_rendered_dev) { sgr_set_error("sgr_bin: self-scan needs the device counter"); return 1; }
    if (runs && run_nt == kRunThreads) hipLaunchKernelGGL((duplicate_keys_kernel<true, kRunThreads>), dim3(nbx_e, pb->n_views), dim3(kRunThreads), 0, stream, pb->P, Tx, Tx * Ty, nbx,
                                 radii, (uint4 *)rect, block_offsets, n, collect ? kin : kout, (uint32_t *)nullptr, ex);
    else if (runs) hipLaunchKernelGGL((duplicate_keys_kernel<true, kThreads>), dim3(nbx_e, pb->n_views), dim3(kThreads), 0, stream, pb->P, Tx, Tx * Ty, nbx,
                                 radii, (uint4 *)rect, block_offsets, n, collect ? kin : kout, (uint32_t *)nullptr, ex);
    else hipLaunchKernelGGL((duplicate_keys_kernel<false, kThreads>), dim3(nbx, pb->n_views), dim3(kThreads), 0, stream, pb->P, Tx, Tx * Ty, nbx,
                            radii, (uint4 *)rect, block_offsets, n, keys_a, vals_a, ex);
    SGR_CHECK_LAUNCH("duplicate_keys_kernel");
    }
    GatherFront no_gf;
    memset(&no_gf, 0, sizeof(no_gf));
    if (mode == 4 && collect) {
        // deep launch of one or two views: tile-ordered emission runs (in kin, as composites) -> one collect launch (ranges, worklists, the tiles'
        // contiguous segments in kout) -> the per-tile sorts as below
        char *ws = (char *)workspace + RC.end;
        VsegPlan *plan = (VsegPlan *)(ws + VL.plan);
        uint32_t *lists = (uint32_t *)(ws + VL.lists);
        { SgrProfScope _ps(SGR_K_SORT, stream);
        GatherFront gf = no_gf;
        gf.rows = (const uint32_t *)((char *)workspace + RC.rows); gf.base = (const uint32_t *)((char *)workspace + RC.base);
        gf.occ = (const uint32_t *)((char *)workspace + RC.occ);
        gf.nblk = nblk_e; gf.tiles_total = (uint32_t)tiles_total; gf.tx = (uint32_t)Tx; gf.ty = (uint32_t)Ty; gf.ranges = (uint2 *)ranges;
        gf.search_top = 1u; while (gf.search_top * 2u < nblk_e) gf.search_top *= 2u;
        hipLaunchKernelGGL(tile_collect_kernel<1024>, dim3((uint32_t)tiles_total), dim3(1024), 0, stream, (const uint64_t *)kin, kout, gf, plan, lists,
                           VL.list_stride, 0u, kDeepSmallCap - kDeepBinMax);
        SGR_CHECK_LAUNCH("tile_collect_kernel");
        const uint32_t gbig = std::max(1u, (uint32_t)std::min<uint64_t>(R / (kDeepSmallCap - kDeepBinMax) + 1, 256u));
        const uint32_t gsmall = std::max(1u, (uint32_t)std::min<uint64_t>(std::min<uint64_t>(tiles_total, R / 64 + 1), 768u));
        if (sgr_deep_tile_launch(0, gbig, stream, kout, vout, kin, vin, &plan->count[6], lists + (size_t)6 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists,
                           VL.list_stride, no_gf)) return 1;
        if (sgr_deep_tile_launch(1, gsmall, stream, kout, vout, kin, vin, &plan->count[7], lists + (size_t)7 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists,
                           VL.list_stride, no_gf)) return 1;
        auto work = [&](int cls) { TileWork w = {lists + (size_t)cls * VL.list_stride, &plan->ticket[cls], &plan->count[cls]}; return w; };
        TileWork4 tw4;
        for (int c = 0; c < 6; c++) tw4.w[c] = work(c);
        if (sgr_tile_sort_regs_launch(std::min<uint32_t>((uint32_t)tiles_total, 64u) + (fwd_order ? 1u : 0u), stream, (const uint2 *)ranges, kout, vout, kin, vin, tw4, 4, 0,
                                      sorted_keys ? 1 : 0, fwd_order, (uint32_t)tiles_total)) return 1;
        if (fwd_order && order_kind_out) *order_kind_out = 2;
        }
        if (result_in_b_host) *result_in_b_host = (kin == keys_b) ? 1 : 0;
        return 0;
    }
    if (mode == 4) {
        char *ws = (char *)workspace;
        VsegPlan *plan = (VsegPlan *)(ws + VL.plan);
        unsigned long long *totals = (unsigned long long *)(ws + VL.totals);
        uint32_t *key_start = (uint32_t *)(ws + VL.key_start), *chunk_start = (uint32_t *)(ws + VL.chunk_start);
        uint4 *chunk_map = (uint4 *)(ws + VL.chunk_map);
        uint32_t *vhist = (uint32_t *)(ws + VL.hist), *tile_total = (uint32_t *)(ws + VL.tile_total), *lists = (uint32_t *)(ws + VL.lists);
        const uint32_t nblk = (uint32_t)nbx * (uint32_t)pb->n_views;
        const uint32_t *sums = block_offsets + (nblk + 1);                   // un-scanned per-block emission counts, [view][block]
        { SgrProfScope _ps(SGR_K_SORT, stream);
        const bool fold_totals = pb->n_views <= (int)kVsegFoldViews && (uint64_t)pb->n_views * (uint64_t)nbx <= 65536ull;      // (C3: 64 views x 391 counts)
        if (!fold_totals) hipLaunchKernelGGL(vseg_view_totals_kernel, dim3(pb->n_views), dim3(kThreads), 0, stream, sums, (uint32_t)nbx, totals);
        hipLaunchKernelGGL(vseg_plan_kernel, dim3(1), dim3(1024), 0, stream, totals, (uint32_t)pb->n_views, n, num_rendered_dev, VL.chunk_keys,
                           VL.max_chunks, plan, key_start, chunk_start, chunk_map, fold_totals ? sums : (const uint32_t *)nullptr, (uint32_t)nbx);
        if (VL.chunk_keys == 4096u) hipLaunchKernelGGL(vseg_upsweep_kernel<16>, dim3(VL.max_chunks), dim3(kThreads), 0, stream, kin, plan, chunk_map, tpv, vhist);
        else hipLaunchKernelGGL(vseg_upsweep_kernel<32>, dim3(VL.max_chunks), dim3(kThreads), 0, stream, kin, plan, chunk_map, tpv, vhist);
        // views with many chunks (one or a few views of a deep launch): the chunk-parallel column scan.  Its cost grows with tiles x views
        // (C5, 1 view x 1024 tiles, 540 chunks: 7 us against 70 us of dependent round trips), the serial scan's with the chunks per view only:
        // at 76 chunks per view the serial scan is the faster one (C4, 90 views x 4096 tiles: 31 us against 45 us)
        if ((uint64_t)VL.max_chunks >= 192ull * (uint64_t)pb->n_views)
            hipLaunchKernelGGL(vseg_colscan_par_kernel, dim3((tpv + 63u) / 64u, pb->n_views), dim3(1024), 0, stream, vhist, chunk_start, tpv, tile_total);
        else
        hipLaunchKernelGGL(vseg_colscan_kernel, dim3((tpv + kThreads - 1) / kThreads, pb->n_views), dim3(kThreads), 0, stream, vhist, chunk_start, tpv, tile_total);
        // deep mode: one or two views -> EVERY tile goes through the LDS distribution sort (the register sort's floor is one wave's
        // 1024-entry network, ~15 us, whatever the launch holds); more views -> the tiles beyond the single-wave class
        const uint32_t deep_min = !split ? 0xFFFFFFFFu : (tiles_total <= 2048 ? 0u : 1024u);                  // 0xFFFFFFFF: no tile
        hipLaunchKernelGGL(vseg_scan_kernel, dim3(pb->n_views), dim3(1024), 0, stream, tile_total, key_start, tpv, (uint2 *)ranges, plan, lists,
                           VL.list_stride, deep_min, kDeepSmallCap - kDeepBinMax);
        if (VL.chunk_keys == 4096u)
            hipLaunchKernelGGL((vseg_scatter_staged_kernel<1024, 4>), dim3(VL.max_chunks + 8), dim3(1024), 0, stream, kin, vin, kout, plan, chunk_map, tpv, vhist,
                               (const uint2 *)ranges);
        else
            hipLaunchKernelGGL((vseg_scatter_staged_kernel<4096, 8>), dim3(VL.max_chunks + 8), dim3(1024), 0, stream, kin, vin, kout, plan, chunk_map, tpv, vhist,
                               (const uint2 *)ranges);
        SGR_CHECK_LAUNCH("view-segmented tile pass");
        if (split) {
            // one workgroup per (window of a) deep tile sorts it by distribution in LDS; the final list goes straight to (kin, vin)
            // (grids of resident workgroups that stride over their lists: the counts are only known on the device, and a 1024-thread workgroup
            // with 155 KB of LDS that starts only to find nothing to do still holds a CU for microseconds)
            const uint32_t gbig = std::max(1u, (uint32_t)std::min<uint64_t>(R / (kDeepSmallCap - kDeepBinMax) + 1, 256u));
            const uint32_t gsmall = std::max(1u, (uint32_t)std::min<uint64_t>(std::min<uint64_t>(tiles_total, R / 64 + 1), 768u));
            if (sgr_deep_tile_launch(0, gbig, stream, kout, vout, kin, vin, &plan->count[6], lists + (size_t)6 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists,
                               VL.list_stride, no_gf)) return 1;
            if (sgr_deep_tile_launch(1, gsmall, stream, kout, vout, kin, vin, &plan->count[7], lists + (size_t)7 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists,
                               VL.list_stride, no_gf)) return 1;
        }
        {
        // depth bits per tile: keys now sit tile-bucketed in (kout, vout); the sorted list goes back into (kin, vin)
        auto work = [&](int cls) { TileWork w = {lists + (size_t)cls * VL.list_stride, &plan->ticket[cls], &plan->count[cls]}; return w; };
        auto grid = [&](uint32_t per_cu) { const uint64_t g = (uint64_t)per_cu * 256u; return (uint32_t)(tiles_total < g ? tiles_total : g); };
        // longest tiles first; tiles beyond the LDS capacity go through the global ping-pong buffers, a whole workgroup per tile
        const uint2 *rg = (const uint2 *)ranges;
        TileWork4 tw4;
        for (int c = 0; c < 6; c++) tw4.w[c] = work(c);
        // (deep mode with every tile on the deep lists: only the tiles those kernels declined -- massive depth ties -- are left: a small grid,
        // it usually just exits)
        if (sgr_tile_sort_regs_launch((deep_min == 0u ? std::min(grid(1), 64u) : grid(1)) + (fwd_order ? 1u : 0u), stream, rg, kout, vout, kin, vin, tw4, 4, 0,
                                      sorted_keys ? 1 : 0, fwd_order, (uint32_t)tiles_total)) return 1;
        if (fwd_order && order_kind_out) *order_kind_out = 2;
        }
        }
        if (result_in_b_host) *result_in_b_host = (kin == keys_b) ? 1 : 0;
        return 0;
    }
    if (mode == 5) {
        // the tile-ordered runs sit in kout as composites; the sorted list goes into (kin, vin).  One launch, one 512-thread workgroup per tile, two
        // per CU: the occupied tiles (212 of 1 024 at a humanoid view) all start at once and the empty tiles' workgroups -- one load and gone -- pass
        // through the slots beside them.  (1024-thread workgroups sort a tile ~1 us faster, but only ONE fits a CU (123 VGPRs): 812 empty-tile
        // workgroups then queued for the 44 CUs the working ones left free and the launch lasted 24 us instead of 14.)
        { SgrProfScope _ps(SGR_K_SORT, stream);
        GatherFront gf = no_gf;
        gf.rows = (const uint32_t *)((char *)workspace + RL.rows); gf.base = (const uint32_t *)((char *)workspace + RL.base);
        gf.occ = (const uint32_t *)((char *)workspace + RL.occ);
        gf.nblk = nblk_e; gf.tiles_total = (uint32_t)tiles_total; gf.tx = (uint32_t)Tx; gf.ty = (uint32_t)Ty; gf.ranges = (uint2 *)ranges;
        gf.search_top = 1u; while (gf.search_top * 2u < nblk_e) gf.search_top *= 2u;
        gf.order = fwd_order ? (uint4 *)(fwd_order + SGR_ORDER_HDR_WORDS) : nullptr; gf.cls_count = fwd_order;
        gf.scratch_k = (uint64_t *)((char *)workspace + RL.scratch_k);
        gf.max_windows = g_deep_max_windows; gf.cap_dbg = n;
        if (bg && fwd_order) gf.bg = *bg;
        if (sgr_deep_tile_launch(2, (uint32_t)tiles_total, stream, kout, vout, kin, vin, nullptr, nullptr, nullptr, sorted_keys ? 1 : 0, nullptr, nullptr, 0u, gf)) return 1;
        if (order_kind_out && fwd_order) *order_kind_out = 1;
        }
        if (result_in_b_host) *result_in_b_host = (kin == keys_b) ? 1 : 0;
        return 0;
    }
    // ---- the fallback: three kernels per 8-bit digit over the whole key (> 4096 tiles per view, or no room for the view-segmented plan)
    { SgrProfScope _ps(SGR_K_SORT, stream);
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
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, kin, n, num_rendered_dev, (uint2 *)ranges);
    SGR_CHECK_LAUNCH("tile_ranges_kernel");
    }
    return 0;
}

extern "C" int sgr_bin(const SgrProblem *pb, const int32_t *radii, uint32_t *rect,
                       const uint32_t *block_offsets, uint64_t R, const uint64_t *num_rendered_dev, uint64_t *keys_a,
                       uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *workspace, size_t workspace_bytes,
                       uint32_t *ranges, int32_t *result_in_b_host, void *stream_) {
    return sgr_bin_ex(pb, radii, rect, block_offsets, R, num_rendered_dev, keys_a, keys_b, vals_a, vals_b, workspace,
                      workspace_bytes, ranges, result_in_b_host, false, nullptr, nullptr, nullptr, nullptr, false, nullptr, nullptr, nullptr, /*first_index=*/true, /*sorted_keys=*/true, stream_);
}
