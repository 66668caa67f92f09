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

// ---- F3 -----------------------------------------------------------------------------------------
// Same grid as preprocess (blockIdx.y = view).  The block re-scans its 256 tile counts in LDS and adds
// the block offset from F2, so the per-Gaussian offsets array of the published algorithm is never
// materialised in HBM.
__global__ __launch_bounds__(kThreads) void duplicate_keys_kernel(int P, int Tx, int tiles_per_view,
                                                                  float4 *__restrict__ rec,
                                                                  const int32_t *__restrict__ radii,
                                                                  const uint2 *__restrict__ rect,
                                                                  const uint32_t *__restrict__ block_offsets, uint32_t cap,
                                                                  uint64_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    __shared__ uint32_t wave_tot[4];
    const int view = blockIdx.y;
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
    uint32_t base = block_offsets[(size_t)view * gridDim.x + blockIdx.x];
    for (int w = 0; w < wave; w++) base += wave_tot[w];
    uint32_t off = base + inc - cnt;
    if (cnt) {
        const uint32_t dbits = __float_as_uint(rec[q * 4 + 1].z);
        rec[q * 4 + 3].x = __uint_as_float(off);           // first tile-instance index of this Gaussian (backward gather)
        const uint32_t tbase = (uint32_t)view * (uint32_t)tiles_per_view;
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                if (off < cap) {                                   // capacity mode: never write past the caller's buffers
                    keys[off] = ((uint64_t)(tbase + (uint32_t)(y * Tx + x)) << 32) | dbits;
                    vals[off] = (uint32_t)q;
                }
                off++;
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
    for (int it = 0; it < ITEMS; it++) {
        const uint32_t k = base + it * kThreads + t;
        const bool valid = k < n;
        uint64_t key = 0;
        uint32_t val = 0;
        if (valid) { key = keys_in[k]; val = vals_in[k]; }
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

inline int bits_for(uint64_t v) { int b = 0; while ((1ull << b) < v) b++; return b; }   // smallest b with 2^b >= v

}  // namespace

int sgr_validate_problem(const SgrProblem *pb);

extern "C" size_t sgr_bin_workspace_bytes(uint64_t R) {
    const uint64_t nblocks = (R + kThreads * kItemsSmall - 1) / (kThreads * kItemsSmall);
    return (size_t)(((nblocks > 0 ? nblocks : 1) + 1) * kRadix * sizeof(uint32_t) + 256);
}

extern "C" int sgr_bin(const SgrProblem *pb, float *rec, const int32_t *radii, const uint32_t *rect,
                       const uint32_t *block_offsets, uint64_t R, const uint64_t *num_rendered_dev, uint64_t *keys_a,
                       uint64_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, void *workspace, size_t workspace_bytes,
                       uint32_t *ranges, int32_t *result_in_b_host, void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint64_t tiles_total = (uint64_t)Tx * Ty * pb->n_views;
    if (tiles_total >= (1ull << 32)) { sgr_set_error("too many tiles (%llu)", (unsigned long long)tiles_total); return 1; }
    if (R > 0xFFFFFFF0ull) { sgr_set_error("num_rendered %llu exceeds the 32-bit instance index", (unsigned long long)R); return 1; }
    SGR_CHECK_HIP(hipMemsetAsync(ranges, 0, tiles_total * 2 * sizeof(uint32_t), stream));
    if (result_in_b_host) *result_in_b_host = 0;
    if (R == 0 || pb->P == 0) return 0;
    if (workspace_bytes < sgr_bin_workspace_bytes(R)) { sgr_set_error("sgr_bin: workspace too small"); return 1; }
    const uint32_t n = (uint32_t)R;
    const int nbx = sgr_preprocess_blocks_per_view(pb->P);
    { SgrProfScope _p(SGR_K_DUPLICATE, stream);
    hipLaunchKernelGGL(duplicate_keys_kernel, dim3(nbx, pb->n_views), dim3(kThreads), 0, stream, pb->P, Tx, Tx * Ty,
                       (float4 *)rec, radii, (const uint2 *)rect, block_offsets, n, keys_a, vals_a);
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
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, kin, n,
                       num_rendered_dev, (uint2 *)ranges);
    SGR_CHECK_LAUNCH("tile_ranges_kernel");
    }
    return 0;
}
