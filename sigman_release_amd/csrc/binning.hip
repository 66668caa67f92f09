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
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
// keys per thread per workgroup: small inputs (one 512^2 view: R ~ 2e5) want many small workgroups to fill 256 CUs,
// large batches want fewer, longer ones (less histogram traffic)
constexpr int kItemsSmall = 4, kItemsLarge = 16;

// the LDS distribution sort of long tile lists (deep_tile_kernel): a fine bin holds at most kDeepBinMax composites; LDS composites per
// workgroup of the big (15 360-composite) / small (4 096-composite) instantiation; worklist entry = tile id | window << 26
constexpr uint32_t kDeepBinMax = 128, kDeepBigCap = 15360, kDeepSmallCap = 4096, kDeepTileMask = 0x03FFFFFFu;
constexpr uint32_t kDeepMaxWindows = 64;           // 6-bit window field
constexpr int kRunThreads = 1024;                  // threads of an emission workgroup on the single-view path (duplicate_keys_kernel<true, ..>)
constexpr int kTileBins = 2048;                    // most tiles of a launch that takes the single-view path (one or two 512^2 views)
// The single-view path (<= kTileBins tiles, <= 512 emission workgroups, <= 2^19 instances): the emission workgroup b writes its key run ORDERED BY
// TILE (composites, depth bits << 32 | value) and one row of the run matrix: rows[b][T] = position of the first composite of tile T in its run,
// rows[b][T + 1] = the end of that piece; run_base[b] = where the run starts.  The per-tile sort then needs no tile pass at all: the workgroup
// of tile T reads column T (and T + 1) of the matrix -- one strided round trip --, which gives it the pieces of its list in every run AND,
// summed, the number of instances in all tiles before T (sum_b rows[b][T] - run_base[b]), i.e. the tile's range in the sorted list.
// The emission workgroups also mark the tiles they touch in occ[kTileBins] (plain stores of 1 into words the preprocess launch zeroed): the
// workgroup of an EMPTY tile leaves the sort launch after one load.
constexpr uint32_t kRunRow = kTileBins + 32;       // row stride in words: a multiple of four (16-byte stores) and NOT a multiple of the L2 channel
                                                   // interleave -- a column read walks 8 320-byte strides (8 KiB + one line): with 8 208 every row of a
                                                   // tile's column met in the same channel
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
            static_assert(PER == 2, "one uint2 per thread");
            const uint2 hv = reinterpret_cast<const uint2 *>(s_th)[threadIdx.x];
            const uint32_t sum = hv.x + hv.y;
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
            const uint32_t e0 = run, e1 = run + hv.x;
            reinterpret_cast<uint2 *>(s_th)[threadIdx.x] = make_uint2(e0, e1);
            if (hv.x) ex.occ[threadIdx.x * 2] = 1u;
            if (hv.y) ex.occ[threadIdx.x * 2 + 1] = 1u;
            const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
            uint32_t *row = ex.run_rows + (size_t)b * kRunRow;
            // (block_base + x <= 2^32 - 16: the count is checked against that)
            // total == 0 beyond the capacity: every position of the row is base_c = min(block_base, cap).  (The optimiser folded that minimum to
            // block_base -- ROCm 7.2 clang, as if block_base < cap were known here -- and the sort launch then read row positions beyond the
            // buffers: the capacity is hidden from it for this one comparison.)
            uint32_t cap_opaque = cap;
            asm volatile("" : "+s"(cap_opaque));
            const uint32_t base_c = block_base < cap_opaque ? block_base : cap_opaque;
            reinterpret_cast<uint2 *>(row)[threadIdx.x] = make_uint2(base_c + e0, base_c + e1);
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

// The single-view path's front end of deep_tile_kernel<.., FB = true> (one workgroup per TILE, no worklist; kRunRow above)
struct GatherFront {
    const uint32_t *rows, *base;        // the run matrix [nblk][kRunRow] and the runs' first positions [nblk]
    const uint32_t *occ;                // [kTileBins] != 0: some run holds a key of the tile
    uint32_t nblk, tiles_total, tx, ty /* tiles per image row / column */, search_top /* largest power of two < max(nblk, 2) */;
    uint2 *ranges;                      // [tiles_total], written here (F5): every tile's workgroup computes its own range from its columns
    uint4 *order;                       // optional work order of the segment-parallel forward, class-major: order[cls * tiles_total + k] = (tile, first,
    uint32_t *cls_count;                //   end, 0) of the k-th OCCUPIED tile with 31 - min(31, n >> 7) == cls; cls_count[32] (zeroed by the emission kernel)
    uint64_t *scratch_k;                // [R] composites of the tiles that go through global memory (several windows / massive depth ties)
    uint32_t cap_dbg;                   // (capacity of the buffers: -DSGR_DEBUG_BOUNDS checks)
    uint32_t max_windows;               // a tile of more windows is sorted whole by the stable radix passes (<= 64; tests lower it)
    SgrBgJob bg;                        // bg.enabled: the workgroup of an EMPTY tile writes the tile's background (the compositing kernel then never
};                                      //   looks at empty tiles)

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

struct TileWork { const uint32_t *list; uint32_t *ticket; const uint32_t *count; };

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
struct TileWork4 { TileWork w[6]; };        // [m], m = 0..4: tiles with <= 1024 << m entries; [5]: longer ones (global-memory fallback)

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
struct VsegPlan { uint32_t n_chunks, pad[7], count[8], ticket[8]; };      // worklists by tile size: <= 1024, <= 2048, <= 4096, <= 8192, <= 16384, longer; [6], [7]: the deep kernels' lists
constexpr uint32_t kDeepMaxN = 1u << 19;                          // longer tiles (a pathological half a million entries in one 16 x 16 tile) keep the generic path
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
                                                         a launch with <= 16 views: the view totals are then summed here (one launch fewer) */,
                                                         uint32_t nbx) {
    __shared__ uint32_t s_key[kVsegMaxViews + 1], s_chunk[kVsegMaxViews + 1];
    __shared__ unsigned long long s_wave[16], s_vtot[16];
    __shared__ uint32_t s_wave32[16];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (sums) {
        for (uint32_t v = 0; v < n_views; v++) {
            const uint32_t *row = sums + (size_t)v * nbx;
            unsigned long long acc = 0;
            for (uint32_t k = t; k < nbx; k += 1024) acc += row[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (lane == 0) s_wave[wave] = acc;
            __syncthreads();
            if (t == 0) { unsigned long long a = 0; for (int w = 0; w < 16; w++) a += s_wave[w]; s_vtot[v] = a; }
            __syncthreads();
        }
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
        const uint64_t *seg;                                                    // the tile's composites in one piece (FB: only for tiles beyond one window)
        if constexpr (FB) {
            // ---- my columns of the run matrix: piece of run b = [rows[b][tile], rows[b][tile + 1]); sum_b (rows[b][tile] - base[b]) instances
            // sit in tiles before mine
            // workgroup i <-> tile i.  An empty tile's workgroup (three of four at a humanoid view) leaves after one load: its range, its background
            w0 = 0u;
            // Workgroup ids go round the XCDs and, inside an XCD, round its shader engines: with 32 tiles per image row, id mod 32 -- one image
            // COLUMN -- would meet in one engine, and a humanoid's centre columns hold more occupied tiles (24) than an engine has slots for these
            // workgroups (16): the last ones started when the first had finished, 14 us late.  Ids walk down the image columns instead, every
            // column rotated by 5 rows more than the one before, so an engine's tiles lie on a diagonal.
            // And the columns are visited from the image centre outwards (the workgroups that do not fit the chip at once -- the second half of
            // the ids -- are the image's outer columns: a centred subject's tiles all start at t = 0).
            {
                const uint32_t tpv = gf.tx * gf.ty, vw = i / tpv, r = i - vw * tpv;
                const uint32_t ci = r / gf.ty, ri = r - ci * gf.ty, mid = gf.tx >> 1;
                const uint32_t tcol = (ci & 1u) ? mid - 1u - (ci >> 1) : mid + (ci >> 1);
                tile = vw * tpv + ((ri + 5u * ci) % gf.ty) * gf.tx + tcol;
            }
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
            if (t < 512u) { s_pre[t] = pre + inc - cnt; s_cur[t] = a; }          // piece prefix / piece start (runs beyond nblk: prefix = n, never chosen)
            static_assert(NT >= 512, "one run per thread");
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
            // composite j of the tile: in the piece of the last run whose prefix is <= j
            auto gather = [&](uint32_t j) -> uint64_t {
                uint32_t lo = 0u;
                for (uint32_t step = gf.search_top; step > 0u; step >>= 1) if (s_pre[lo + step] <= j) lo += step;
#ifdef SGR_DEBUG_BOUNDS
                if (s_cur[lo] + (j - s_pre[lo]) >= gf.cap_dbg) { printf("gather oob tile %u j %u lo %u pre %u cur %u n %u nblk %u\n", tile, j, lo, s_pre[lo], s_cur[lo], n_t, gf.nblk); return 0; }
#endif
                return comp[s_cur[lo] + (j - s_pre[lo])];
            };
            const uint32_t last_j = n_t - 1u;
            if (n_t <= WIN) {
#pragma unroll
                for (uint32_t it = 0; it < RI; it++) c[it] = gather(min(it * NT + t, last_j));
                seg = nullptr;                                                   // (n <= WIN <= REG: nothing below reads it)
            } else {
                uint64_t *sk = gf.scratch_k + range.x;
                for (uint32_t j = t; j < n_t; j += NT) sk[j] = gather(j);
                __threadfence_block();
                __syncthreads();
                seg = sk;
#pragma unroll
                for (uint32_t it = 0; it < RI; it++) c[it] = seg[min(it * NT + t, last_j)];
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
        // this workgroup alone.  The composites must sit in scratch_k: a one-window tile's are still in registers only
        auto sort_here = [&]() {
            if (!seg) {
#pragma unroll
                for (uint32_t it = 0; it < RI; it++) if (it * NT + t < n) gf.scratch_k[range.x + it * NT + t] = c[it];
                __threadfence_block();
            }
            __syncthreads();
            sort_one_tile_ool<NT>(range, gf.scratch_k, scratch, dst_keys, dst_vals, (uint32_t *)s_comp, tile);
        };
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
        for_each_rest([&](uint64_t v) { const uint32_t z = (uint32_t)(v >> 32); lo = min(lo, z); hi = max(hi, z); });
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
    {   // (the dispatch order of deep_tile_kernel<.., FB>: image columns from the centre outwards, rotated rows -- see there)
        const uint32_t tpv = gf.tx * gf.ty, vw = i / tpv, r = i - vw * tpv;
        const uint32_t ci = r / gf.ty, ri = r - ci * gf.ty, mid = gf.tx >> 1;
        const uint32_t tcol = (ci & 1u) ? mid - 1u - (ci >> 1) : mid + (ci >> 1);
        tile = vw * tpv + ((ri + 5u * ci) % gf.ty) * gf.tx + tcol;
    }
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
    // the copy: one thread per composite, which finds its piece by binary search over the piece prefixes (four composites per trip: their
    // searches and loads overlap).  (Piece by piece -- a wave per piece, contiguous loads and stores, no search -- was SLOWER: 42 against 31 us
    // at C5; a piece holds ~30 composites, half a wave's lanes idle and twice the load instructions.)
    uint64_t *out = dst + x0;
    for (uint32_t j0 = t; j0 < n; j0 += 4u * NT) {
        uint64_t v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const uint32_t j = min(j0 + u * NT, n - 1u);
            uint32_t lo = 0u;
            for (uint32_t step = gf.search_top; step > 0u; step >>= 1) if (s_pre[lo + step] <= j) lo += step;
            v[u] = runs[s_cur[lo] + (j - s_pre[lo])];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) if (j0 + u * NT < n) out[j0 + u * NT] = v[u];
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

// 3 = automatic (default), 5 = single wide tile pass + LDS distribution sort per tile (one or two 512^2 views; else like 3),
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
    if (mode != 1 && mode != 3 && mode != 4 && mode != 5) { sgr_set_error("sgr_set_sort_mode: %d is not a sort flavour (3 automatic, 5 wide pass, 4 view-segmented, 1 whole-key passes)", mode); return 1; }
    sgr_sort_mode = mode;
    return 0;
}
// deep tile lists in the view-segmented flavour (the LDS distribution sort, deep_tile_kernel): 0 = automatic (launches whose tile lists are
// deep on average), 1 = whenever that flavour runs, 2 = never (deep launches then keep the whole-key passes)
static thread_local int g_deep_mode = sgr_env_knob("SIGMAN_SORT_DEEP", 0, 2, 0);
// bits 8..15 of `mode` (tests): the most windows a tile may have behind the single wide tile pass before it is listed once and sorted whole
// (0 = the window field's 64)
static thread_local uint32_t g_deep_max_windows = kDeepMaxWindows;
static thread_local int g_collect_mode = sgr_env_knob("SIGMAN_SORT_COLLECT", 0, 1, 1);      // A/B: 0 = deep launches of one or two views keep the five-launch tile pass
extern "C" int sgr_set_sort_deep(int mode) {
    const int m = mode & 0xFF, cap = (mode >> 8) & 0xFF;
    g_deep_mode = (m >= 0 && m <= 2) ? m : 0;
    g_deep_max_windows = (cap >= 1 && cap <= (int)kDeepMaxWindows) ? (uint32_t)cap : kDeepMaxWindows;
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
    const uint32_t nbx_e = (uint32_t)(nbx + kRunThreads / kThreads - 1) / (uint32_t)(kRunThreads / kThreads);     // emission workgroups per view
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
    if (runs) hipLaunchKernelGGL((duplicate_keys_kernel<true, kRunThreads>), dim3(nbx_e, pb->n_views), dim3(kRunThreads), 0, stream, pb->P, Tx, Tx * Ty, nbx,
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
        hipLaunchKernelGGL((deep_tile_kernel<1024, kDeepBigCap, 4096>), dim3(gbig), dim3(1024), 0, stream, kout, vout, kin, vin, &plan->count[6],
                           lists + (size_t)6 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists, VL.list_stride, no_gf);
        hipLaunchKernelGGL((deep_tile_kernel<512, kDeepSmallCap, 1024>), dim3(gsmall), dim3(512), 0, stream, kout, vout, kin, vin, &plan->count[7],
                           lists + (size_t)7 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists, VL.list_stride, no_gf);
        SGR_CHECK_LAUNCH("deep_tile_kernel");
        auto work = [&](int cls) { TileWork w = {lists + (size_t)cls * VL.list_stride, &plan->ticket[cls], &plan->count[cls]}; return w; };
        TileWork4 tw4;
        for (int c = 0; c < 6; c++) tw4.w[c] = work(c);
        hipLaunchKernelGGL(tile_sort_regs_kernel<16>, dim3(std::min<uint32_t>((uint32_t)tiles_total, 64u) + (fwd_order ? 1u : 0u)), dim3(1024), 0, stream, (const uint2 *)ranges, kout, vout,
                           kin, vin, tw4, 4, 0, sorted_keys ? 1 : 0, fwd_order, (uint32_t)tiles_total);
        SGR_CHECK_LAUNCH("tile_sort_regs_kernel");
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
        const bool fold_totals = pb->n_views <= 16;
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
            hipLaunchKernelGGL((deep_tile_kernel<1024, kDeepBigCap, 4096>), dim3(gbig), dim3(1024), 0, stream, kout, vout, kin, vin, &plan->count[6],
                               lists + (size_t)6 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists, VL.list_stride, no_gf);
            hipLaunchKernelGGL((deep_tile_kernel<512, kDeepSmallCap, 1024>), dim3(gsmall), dim3(512), 0, stream, kout, vout, kin, vin, &plan->count[7],
                               lists + (size_t)7 * VL.list_stride, (const uint2 *)ranges, sorted_keys ? 1 : 0, plan, lists, VL.list_stride, no_gf);
            SGR_CHECK_LAUNCH("deep_tile_kernel");
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
        hipLaunchKernelGGL(tile_sort_regs_kernel<16>, dim3((deep_min == 0u ? std::min(grid(1), 64u) : grid(1)) + (fwd_order ? 1u : 0u)), dim3(1024), 0, stream, rg, kout, vout,
                           kin, vin, tw4, 4, 0, sorted_keys ? 1 : 0, fwd_order, (uint32_t)tiles_total);
        SGR_CHECK_LAUNCH("tile_sort_regs_kernel");
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
        hipLaunchKernelGGL((deep_tile_kernel<512, kDeepSmallCap, 1024, true>), dim3((uint32_t)tiles_total), dim3(512), 0, stream, kout, vout, kin, vin,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint2 *)nullptr, sorted_keys ? 1 : 0, (VsegPlan *)nullptr, (uint32_t *)nullptr, 0u, gf);
        SGR_CHECK_LAUNCH("deep_tile_kernel (single-view path)");
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
