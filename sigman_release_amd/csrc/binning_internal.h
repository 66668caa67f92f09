// binning_internal.h -- what binning.hip (emission, tile passes, host orchestration) and tile_sort.hip (the per-tile depth sorts) share:
// constants, the worklist / plan structures and the launchers of the per-tile sort kernels.  Not part of the C ABI.
#pragma once
#include "common.h"

constexpr int kThreads = 256;
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
// keys per thread per workgroup: small inputs (one 512^2 view: R ~ 2e5) want many small workgroups to fill 256 CUs,
// large batches want fewer, longer ones (less histogram traffic)
constexpr int kItemsSmall = 4, kItemsLarge = 16;

// the LDS distribution sort of long tile lists (deep_tile_kernel): a fine bin holds at most kDeepBinMax composites; LDS composites per
// workgroup of the big (15 360-composite) / small (4 096-composite) instantiation; worklist entry = tile id | window << 26
constexpr uint32_t kDeepBinMax = 128, kDeepBigCap = 15360, kDeepSmallCap = 4096, kDeepTileMask = 0x03FFFFFFu;
constexpr uint32_t kVsegFoldViews = 256;          // the plan kernel sums the view totals itself for launches of up to this many views
constexpr uint32_t kDeepTiny = 256;               // single-view path: tiles of <= this many entries are ranked by all-pairs counting
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

// Workgroup id -> tile of the single-view path's per-tile launches (deep_tile_kernel<.., FB>, tile_collect_kernel).  Three things decide it:
//  * workgroup ids go round the XCDs, and a tile's workgroup reads its columns of the run matrix -- 128-byte lines of 32 neighbouring tiles, one
//    image row of a 512^2 view: with image row r handled by XCD r % 8 an XCD pulls only ITS eighth of the matrix over the fabric (C5's collect
//    launch: 977 rows, 8.1 MB -- every XCD fetching all of it was 8-10 us at the head of every workgroup, 5 us now; C2: 2.2 -> 1.8 us);
//  * inside an XCD ids go round the shader engines, which place workgroups IN ORDER: an engine whose next tile finds no free CU holds up
//    everything behind it.  The XCD's image rows are therefore rotated from column to column (an engine's tiles lie on a diagonal) and
//  * the columns are visited from the image centre outwards: the workgroups that do not fit the chip at once are a centred subject's empty margin.
// (tiles per image column not a multiple of 8: the same without the XCD part -- columns centre-out, rows rotated by 5 per column.)
__device__ __forceinline__ uint32_t sgr_tile_of_workgroup(uint32_t i, uint32_t tx, uint32_t ty) {
    const uint32_t tpv = tx * ty, vw = i / tpv, r = i - vw * tpv, mid = tx >> 1;
    uint32_t ci, row;
    if ((ty & 7u) == 0u) {
        const uint32_t k = r & 7u, l = r >> 3, nr = ty >> 3;       // XCD, index within the XCD, image rows per XCD
        ci = l / nr;
        row = k + 8u * ((l - ci * nr + ci) % nr);
    } else {
        ci = r / ty;
        row = (r - ci * ty + 5u * ci) % ty;
    }
    const uint32_t col = (ci & 1u) ? mid - 1u - (ci >> 1) : mid + (ci >> 1);
    return vw * tpv + row * tx + col;
}

struct TileWork { const uint32_t *list; uint32_t *ticket; const uint32_t *count; };
struct TileWork4 { TileWork w[6]; };        // [m], m = 0..4: tiles with <= 1024 << m entries; [5]: longer ones (global-memory fallback)

struct VsegPlan { uint32_t n_chunks, pad[7], count[8], ticket[8]; };      // worklists by tile size: <= 1024, <= 2048, <= 4096, <= 8192, <= 16384, longer; [6], [7]: the deep kernels' lists
constexpr uint32_t kDeepMaxN = 1u << 19;                          // longer tiles (a pathological half a million entries in one 16 x 16 tile) keep the generic path

// ---- launchers of the per-tile sort kernels (tile_sort.hip) -------------------------------------------------------------------------------
// register comparison network (tiles of up to 16 384 entries per workgroup of 16 waves; longer ones: global-memory radix passes), one launch for
// every length class; prep_order != NULL: one spare workgroup also writes the segment-parallel forward's plain work order
int sgr_tile_sort_regs_launch(uint32_t grid, hipStream_t stream, const uint2 *ranges, uint64_t *src_comp, uint32_t *src_scratch, uint64_t *dst_keys,
                              uint32_t *dst_vals, const TileWork4 &tw, int m_hi, int m_lo, int keep_keys, uint32_t *prep_order, uint32_t prep_tiles);
// LDS distribution sort: which = 0 the long-tile instantiation (1024 threads, 15 360 composites of LDS), 1 the short-tile one (512 threads, 4 096),
// 2 the single-view path (one workgroup per tile, GatherFront)
int sgr_deep_tile_launch(int which, uint32_t grid, hipStream_t stream, uint64_t *comp, uint32_t *scratch, uint64_t *dst_keys, uint32_t *dst_vals,
                         const uint32_t *count_ptr, uint32_t *deep_list, const uint2 *ranges, int keep_keys, VsegPlan *plan, uint32_t *lists,
                         uint32_t list_stride, const GatherFront &gf);
