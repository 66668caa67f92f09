// knn.hip -- exact mean squared distance to the 3 nearest OTHER points, and the fused covariance build.
// Replaces, on the call path /root/reference/core/gaussians/gs.py:70-73,
//   * simple_knn._C.distCUDA2(means3D)                         (third-party CUDA, un-vendored)
//   * get_covariance / strip_lowerdiag (gs.py:17-38) + the ~10 PyTorch kernels around them (row A4)
//
// distCUDA2's result is implementation independent (exact 3-NN), so instead of upstream's Morton sort +
// box pruning this uses what suits MI355X: a uniform grid hash built with a counting sort (LDS-free,
// a few HBM-streaming passes over P points) and a ring-by-ring cell search with a provable stop test:
// after all cells within Chebyshev ring r of the point's cell are searched, every unsearched point is
// at least r*cell away, so the search stops as soon as the 3rd-best squared distance <= (r*cell)^2.
// The three distances are summed in sorted order, so the result does not depend on the (atomic) order
// in which points landed in their cells.
#include "common.h"

namespace {

constexpr int kT = 256;

struct Grid {
    float minx, miny, minz;
    float inv_cell, cell;
    int gx, gy, gz;
};

__device__ __forceinline__ int cell_of(const Grid &g, float x, float y, float z, int &cx, int &cy, int &cz) {
    cx = min(g.gx - 1, max(0, (int)((x - g.minx) * g.inv_cell)));
    cy = min(g.gy - 1, max(0, (int)((y - g.miny) * g.inv_cell)));
    cz = min(g.gz - 1, max(0, (int)((z - g.minz) * g.inv_cell)));
    return (cz * g.gy + cy) * g.gx + cx;
}

// All kernels of the 3-NN run over a BATCH of point sets (blockIdx.y = set): the reference calls distCUDA2 once per subject of
// the batch (gs.py:62,70); one launch sequence for all subjects keeps 256 CUs busy where a single 100k-point set cannot.
struct KnnBatch {
    int P, max_cells;
    const float *points;      // [n_sets, P, 3]
    float *out;               // [n_sets, P]
    char *ws;                 // n_sets workspaces of ws_stride bytes
    size_t ws_stride;
};
struct KnnSet { const float *pts; float *out; int *bb; Grid *grid; uint32_t *bsum, *cell_start, *cell_fill, *pt_cell; float4 *sorted; };
__device__ __forceinline__ KnnSet knn_set(const KnnBatch &kb, int set) {
    KnnSet k;
    k.pts = kb.points + (size_t)set * kb.P * 3;
    k.out = kb.out + (size_t)set * kb.P;
    uint32_t *w = (uint32_t *)(kb.ws + (size_t)set * kb.ws_stride);
    k.bb = (int *)w; k.grid = (Grid *)(w + 8); k.bsum = w + 24; k.cell_start = w + 24 + 1024;
    k.cell_fill = k.cell_start + (size_t)kb.max_cells + 1; k.pt_cell = k.cell_fill + max(kb.max_cells, kb.P);      // (cell_fill: one rank per POINT)
    k.sorted = (float4 *)(((uintptr_t)(k.pt_cell + kb.P) + 15) & ~(uintptr_t)15);
    return k;
}

// bbox: per-block min/max -> atomics on ordered-int encodings
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

constexpr int kBoxBlocks = 64;       // workgroups of the prepare kernel that also reduce the bounding box (one partial each)

// One launch for everything the grid build needs before it can count: the cell counters / fill cursors are
// cleared, and the first kBoxBlocks workgroups leave one bounding-box partial each (6 floats at the head of the not-yet-used `sorted`
// array).  No atomics, no ticket: nothing here needs initialised memory (the workspace is whatever the caller's allocator returned).
__global__ __launch_bounds__(kT) void knn_prep_kernel(KnnBatch kb, int box_blocks) {
    const KnnSet ks = knn_set(kb, blockIdx.y);
    const size_t n = (size_t)kb.max_cells + 1;                // the cell counters (the array behind them holds every point's rank in its cell: no clear)
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) ks.cell_start[i] = 0u;
    if ((int)blockIdx.x >= box_blocks) return;
    const int P = kb.P; const float *pts = ks.pts;
    __shared__ float red[4][6];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * kT + threadIdx.x; i < P; i += box_blocks * kT)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // (only coordinates of a sane magnitude shape the grid: one Inf / 1e30 from a diverged decoder would otherwise stretch it until every
            // other point shares ONE cell and the search is all pairs; such points -- and NaN ones -- are binned into the border cells by cell_of's
            // clamps, which keeps the stop test of the search valid: clamping only ever moves a point's cell towards the rest)
            const float v = pts[3 * (size_t)i + k];
            if (fabsf(v) <= 1.0e15f) { mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64)); }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { red[wave][k] = mn[k]; red[wave][3 + k] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const float a = red[0][k], b = red[1][k], c = red[2][k], d = red[3][k];
        reinterpret_cast<float *>(ks.sorted)[blockIdx.x * 6 + k] = k < 3 ? fminf(fminf(a, b), fminf(c, d)) : fmaxf(fmaxf(a, b), fmaxf(c, d));
    }
}

// derive the grid from the bounding box on device (no host round trip): cell = cbrt(volume * 2 / P), clamped so the
// grid has at most max_cells cells; degenerate extents are padded.
__device__ void grid_setup(const KnnBatch &kb, const float (&mn)[3], const float (&mxv)[3], Grid *g) {
    const int P = kb.P, max_cells = kb.max_cells;
    float ex[3];
    for (int k = 0; k < 3; k++) ex[k] = fmaxf(mxv[k] - mn[k], 1e-6f);
    // ~1 point per cell if the cloud fills its bounding volume, ~5 per occupied cell if it is a surface (the reference's
    // inputs are points on the SMPL-X surface): take the finer of the two estimates.  (The search is bound by its per-lane candidate
    // loads -- one lane per cycle and CU through the texture addresser -- so the cells are as fine as the stop test allows: with
    // 0.8 x this edge 100 000 surface points take 71 instead of 93 us, with 0.65 x more queries need a second shell and the scan over
    // the cells grows: 73 us.)
    float vol = ex[0] * ex[1] * ex[2];
    const float area = 2.f * (ex[0] * ex[1] + ex[1] * ex[2] + ex[0] * ex[2]);
    float cell = fminf(cbrtf(vol * 1.024f / (float)max(P, 1)), sqrtf(area * 1.28f / (float)max(P, 1)));
    const float longest = fmaxf(ex[0], fmaxf(ex[1], ex[2]));
    cell = fmaxf(cell, longest / 1024.f);
    for (int it = 0; it < 64; it++) {
        const double n = ceil((double)ex[0] / cell + 1e-3) * ceil((double)ex[1] / cell + 1e-3) * ceil((double)ex[2] / cell + 1e-3);
        if (n <= (double)max_cells) break;
        cell *= 1.26f;
    }
    g->minx = mn[0]; g->miny = mn[1]; g->minz = mn[2];
    g->cell = cell; g->inv_cell = 1.0f / cell;
    g->gx = max(1, (int)ceilf(ex[0] / cell + 1e-3f)); g->gy = max(1, (int)ceilf(ex[1] / cell + 1e-3f));
    g->gz = max(1, (int)ceilf(ex[2] / cell + 1e-3f));
}

__global__ __launch_bounds__(kT) void cell_count_kernel(KnnBatch kb, int box_blocks) {
    const KnnSet ks = knn_set(kb, blockIdx.y);
    const int P = kb.P; const float *pts = ks.pts; uint32_t *cell_cnt = ks.cell_start, *pt_cell = ks.pt_cell;
    // every workgroup reduces the (<= 64) bounding-box partials and derives the grid for itself (the same few hundred bytes out of L2,
    // the same arithmetic: the same grid in every workgroup); workgroup 0 also stores it for the kernels that follow.  Replaces the
    // bounding-box kernel's atomics + last-workgroup ticket and the launch that initialised them.
    __shared__ Grid s_grid;
    if (threadIdx.x < 64) {
        const float *part = reinterpret_cast<const float *>(ks.sorted);
        const int b = min((int)threadIdx.x, box_blocks - 1);                 // (lanes beyond the partials repeat the last one)
        float mn[3], mx[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = part[b * 6 + k]; mx[k] = part[b * 6 + 3 + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64)); }
        }
        if (threadIdx.x == 0) {
            grid_setup(kb, mn, mx, &s_grid);
            if (blockIdx.x == 0) *ks.grid = s_grid;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= P) return;
    const Grid g = s_grid;
    int cx, cy, cz;
    const int c = cell_of(g, pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], cx, cy, cz);
    pt_cell[i] = (uint32_t)c;
    // the point's rank among the points of its cell (arrival order: the result does not depend on it, see the header) -- the scatter kernel
    // then needs neither cursors nor atomics of its own, and nobody has to clear a second per-cell array
    ks.cell_fill[i] = atomicAdd(&cell_cnt[c], 1u);
}

// exclusive scan over the cells in two parallel steps (a fine grid has ~1e6 cells: one workgroup walking them serially
// was the single most expensive kernel of the 3-NN): per-4096-cell block sums; local rescan + the sum of the block sums in front (every workgroup adds those <= 1024 values itself).
// After it cell_start[c] = first slot of cell c and cell_start[ncell] = P.
constexpr int kScanTile = 4096;

__global__ __launch_bounds__(kT) void cell_blocksum_kernel(KnnBatch kb) {
    const KnnSet ks = knn_set(kb, blockIdx.y);
    const Grid *gp = ks.grid; const uint32_t *data = ks.cell_start; uint32_t *bsum = ks.bsum;
    __shared__ uint32_t red[4];
    const uint32_t n = (uint32_t)(gp->gx * gp->gy * gp->gz) + 1u;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanTile / kT; k++) { const uint32_t i = base + k * kT + threadIdx.x; const uint32_t x = data[min(i, n - 1u)]; s += i < n ? x : 0u; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kT) void cell_scan_kernel(KnnBatch kb) {
    const KnnSet ks = knn_set(kb, blockIdx.y);
    const Grid *gp = ks.grid; uint32_t *data = ks.cell_start; const uint32_t *bsum = ks.bsum;
    __shared__ uint32_t wave_tot[4], wave_pre[4];
    const uint32_t n = (uint32_t)(gp->gx * gp->gy * gp->gz) + 1u;
    const uint32_t base = blockIdx.x * kScanTile;
    if (base >= n) return;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int PER = kScanTile / kT;                           // 16 consecutive cells per thread
    const uint32_t i0 = base + t * PER;
    uint32_t v[PER];
    uint32_t s = 0;
    // (unconditional loads: cells past the end re-read the last one and are zeroed -- a predicated load is a branch and a wait of its own)
#pragma unroll
    for (int k = 0; k < PER; k++) { const uint32_t x = data[min(i0 + (uint32_t)k, n - 1u)]; v[k] = (i0 + k < n) ? x : 0u; s += v[k]; }
    // everything in front of this tile: the workgroup sums the (<= 1024) block sums before its own -- no scan kernel in between
    uint32_t pre = 0;
    for (uint32_t k = t; k < blockIdx.x; k += kT) pre += bsum[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t x = __shfl_up(inc, off, 64); if (lane >= (uint32_t)off) inc += x; }
    if (lane == 63) wave_tot[wave] = inc;
    if (lane == 0) wave_pre[wave] = pre;
    __syncthreads();
    uint32_t e = ((wave_pre[0] + wave_pre[1]) + (wave_pre[2] + wave_pre[3])) + inc - s;
    for (uint32_t w = 0; w < wave; w++) e += wave_tot[w];
#pragma unroll
    for (int k = 0; k < PER; k++) { if (i0 + k < n) data[i0 + k] = e; e += v[k]; }
}

__global__ __launch_bounds__(kT) void cell_scatter_kernel(KnnBatch kb) {
    const KnnSet ks = knn_set(kb, blockIdx.y);
    const int P = kb.P; const float *pts = ks.pts; const uint32_t *pt_cell = ks.pt_cell, *cell_start = ks.cell_start;
    const uint32_t *cell_fill = ks.cell_fill; float4 *sorted = ks.sorted;      // xyz + bitcast original index
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= P) return;
    const uint32_t c = pt_cell[i];
    const uint32_t slot = cell_start[c] + cell_fill[i];           // (rank in the cell, from the count kernel)
    sorted[slot] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __uint_as_float((uint32_t)i));
}

__device__ __forceinline__ void push3(float d, float &b0, float &b1, float &b2) {
    if (d < b2) {
        if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; }
        else b2 = d;
    }
}

// branch-free sorted insert (the same three smallest as push3: nested branches cost more than five min/max in the divergent loops below)
__device__ __forceinline__ void offer3(float d, float &b0, float &b1, float &b2) {
    const float t0 = fminf(b0, d); d = fmaxf(b0, d); b0 = t0;
    const float t1 = fminf(b1, d); d = fmaxf(b1, d); b1 = t1;
    b2 = fminf(b2, d);
}

constexpr int kKnnLanes = 4;                       // lanes per query point

__global__ __launch_bounds__(kT) void knn3_kernel(KnnBatch kb) {
    // kKnnLanes lanes per query point, each taking every kKnnLanes-th x-row of the cube shell (the walk is a serial chain of short
    // divergent loops: what it lacks at 100 000 points is waves, 1.5 per SIMD with one lane per point).  Every lane keeps its own three
    // smallest distances; after a shell the group's three smallest are found by butterfly exchanges (disjoint candidate sets: no
    // distance is counted twice) and lane 0 carries them on.  The three smallest of a set do not depend on the order of insertion.
    const KnnSet ks = knn_set(kb, blockIdx.y);
    const int P = kb.P; const Grid *gp = ks.grid; const uint32_t *cell_start = ks.cell_start; const float4 *sorted = ks.sorted; float *out = ks.out;
    const int s = (blockIdx.x * kT + threadIdx.x) / kKnnLanes;      // points in CELL order: neighbouring groups search the same cells
    const int j = threadIdx.x % kKnnLanes;
    if (s >= P) return;                                            // (whole groups leave together)
    const Grid g = *gp;
    const float4 me = sorted[s];
    int cx, cy, cz;
    cell_of(g, me.x, me.y, me.z, cx, cy, cz);
    float b0 = 3.0e38f, b1 = 3.0e38f, b2 = 3.0e38f;               // this lane's candidates (lane 0: plus everything found in earlier shells)
    const int rmax = max(g.gx, max(g.gy, g.gz));
    // Growing cubes of Chebyshev radius r = 1, 2, ... around the point's cell.  Cells are stored x-fastest, so a whole x-row of
    // the cube is ONE contiguous run of `sorted` (2 boundary loads, then a streaming loop); rows that were already covered by
    // the previous cube only contribute their two new end cells.
    for (int r = 1; r <= rmax; r++) {
        const int z0 = max(0, cz - r), z1 = min(g.gz - 1, cz + r);
        const int y0 = max(0, cy - r), y1 = min(g.gy - 1, cy + r);
        const int xa = max(0, cx - r), xb = min(g.gx - 1, cx + r);
        const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        for (int ri = j; ri < nrows; ri += kKnnLanes) {
            const int z = z0 + ri / ny, y = y0 + ri % ny;
            const int row = (z * g.gy + y) * g.gx;
            const bool inner = r > 1 && abs(y - cy) < r && abs(z - cz) < r;
            if (!inner) {
                const uint32_t lo = cell_start[row + xa], hi = cell_start[row + xb + 1];
                for (uint32_t k = lo; k < hi; k++) {
                    const float4 o = sorted[k];
                    const float dx = o.x - me.x, dy = o.y - me.y, dz = o.z - me.z;
                    const float d = dx * dx + dy * dy + dz * dz;
                    offer3((int)k == s ? 3.0e38f : fminf(d, 3.0e38f), b0, b1, b2);     // (fminf: a NaN / Inf distance is "no neighbour" -- offer3's min / max would duplicate b0 for a NaN, order-dependently)
                }
            } else {
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const int x = side ? cx + r : cx - r;
                    if (x < 0 || x >= g.gx) continue;
                    const uint32_t lo = cell_start[row + x], hi = cell_start[row + x + 1];
                    for (uint32_t k = lo; k < hi; k++) {
                        const float4 o = sorted[k];
                        const float dx = o.x - me.x, dy = o.y - me.y, dz = o.z - me.z;
                        offer3(fminf(dx * dx + dy * dy + dz * dz, 3.0e38f), b0, b1, b2);
                    }
                }
            }
        }
        // the group's three smallest: butterfly over its lanes (every lane ends up with them)
        float m0 = b0, m1 = b1, m2 = b2;
#pragma unroll
        for (int off = 1; off < kKnnLanes; off <<= 1) {
            const float o0 = __shfl_xor(m0, off, 64), o1 = __shfl_xor(m1, off, 64), o2 = __shfl_xor(m2, off, 64);
            offer3(o0, m0, m1, m2); offer3(o1, m0, m1, m2); offer3(o2, m0, m1, m2);
        }
        if (j == 0) { b0 = m0; b1 = m1; b2 = m2; } else { b0 = b1 = b2 = 3.0e38f; }
        const float safe = (float)r * g.cell * 0.9999f;            // every unsearched point is at least this far away
        if (m2 <= safe * safe) break;                              // (group-uniform)
    }
    if (j != 0) return;
    // fewer than 4 points in total: missing neighbours count as distance 0 (upstream initialises its best[] to FLT_MAX
    // and would return garbage; P < 4 never happens on the reference path)
    if (b2 > 1.0e38f) b2 = 0.f;
    if (b1 > 1.0e38f) b1 = 0.f;
    if (b0 > 1.0e38f) b0 = 0.f;
    out[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

// ---- fused covariance build (gs.py:70-73 + gs.py:17-38) -------------------------------------------------
// scale = (scale_raw + 1) * sqrt(max(dist2, 1e-7)) (dist2 detached); Sigma = R diag(scale^2) R^T; pack xx,xy,xz,yy,yz,zz
__global__ __launch_bounds__(kT) void cov3d_fwd_kernel(int n, const float *__restrict__ scale_raw, const float *__restrict__ rot,
                                                       const float *__restrict__ dist2, float *__restrict__ cov6) {
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    const float nn = sqrtf(fmaxf(dist2[i], 0.0000001f));
    float s2[3], R[9];
#pragma unroll
    for (int k = 0; k < 3; k++) { const float s = (scale_raw[3 * (size_t)i + k] + 1.f) * nn; s2[k] = s * s; }
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = rot[9 * (size_t)i + k];
    float S[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = a; b < 3; b++) S[a][b] = R[3 * a] * s2[0] * R[3 * b] + R[3 * a + 1] * s2[1] * R[3 * b + 1] + R[3 * a + 2] * s2[2] * R[3 * b + 2];
    float *o = cov6 + 6 * (size_t)i;
    o[0] = S[0][0]; o[1] = S[0][1]; o[2] = S[0][2]; o[3] = S[1][1]; o[4] = S[1][2]; o[5] = S[2][2];
}

__global__ __launch_bounds__(kT) void cov3d_bwd_kernel(int n, const float *__restrict__ scale_raw, const float *__restrict__ rot,
                                                       const float *__restrict__ dist2, const float *__restrict__ g6,
                                                       float *__restrict__ g_scale_raw, float *__restrict__ g_rot) {
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    const float nn = sqrtf(fmaxf(dist2[i], 0.0000001f));
    float s[3], R[9];
#pragma unroll
    for (int k = 0; k < 3; k++) s[k] = (scale_raw[3 * (size_t)i + k] + 1.f) * nn;
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = rot[9 * (size_t)i + k];
    const float *g = g6 + 6 * (size_t)i;
    // the packed entries are read from the UPPER triangle only (strip_lowerdiag), so dL/dSigma_full is upper-triangular:
    // G[a][b] = g for a <= b, 0 below the diagonal (exactly what autograd gives the reference's code)
    const float G[3][3] = {{g[0], g[1], g[2]}, {0.f, g[3], g[4]}, {0.f, 0.f, g[5]}};
    // Sigma = R D R^T (D = diag(s^2)):  dL/dR = G R D + G^T R D ;  dL/dD_k = (R^T G R)_kk
    float GR[3][3], GtR[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            GR[a][k] = G[a][0] * R[k] + G[a][1] * R[3 + k] + G[a][2] * R[6 + k];
            GtR[a][k] = G[0][a] * R[k] + G[1][a] * R[3 + k] + G[2][a] * R[6 + k];
        }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int k = 0; k < 3; k++) g_rot[9 * (size_t)i + 3 * a + k] = (GR[a][k] + GtR[a][k]) * s[k] * s[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float dD = R[k] * GR[0][k] + R[3 + k] * GR[1][k] + R[6 + k] * GR[2][k];
        g_scale_raw[3 * (size_t)i + k] = dD * 2.f * s[k] * nn;
    }
}

}  // namespace

extern "C" size_t sgr_knn_workspace_bytes(int32_t P, int32_t max_cells) {
    // [bbox 8 u32][Grid 16 u32][cell_cnt/start max_cells+1][point ranks: max(max_cells, P) words][pt_cell P][sorted max(P, 2) float4]
    // (the head of `sorted` first holds the bounding-box partials: 6 floats per prepare workgroup, min(ceil(P / 256), 64) of them --
    // 24 * ceil(P / 256) <= 16 * P bytes from P = 2 on; a single point gets the room of two)
    return (size_t)(8 + 16 + 1024 + (size_t)max_cells + 1 + (size_t)(max_cells > P ? max_cells : P) + (size_t)P) * 4 + (size_t)(P < 2 ? 2 : P) * 16 + 64;
}

extern "C" int sgr_knn_dist2_batched(int32_t n_sets, int32_t P, const float *points, float *out_dist2, void *workspace,
                                     size_t workspace_bytes, int32_t max_cells, void *stream_) {
    if (P <= 0 || n_sets <= 0) return 0;
    if (!points || !out_dist2 || !workspace) { sgr_set_error("sgr_knn_dist2: NULL pointer"); return 1; }
    if (max_cells < 1) max_cells = 1;
    if (max_cells > (1 << 22) - 1) max_cells = (1 << 22) - 1;        // the block-sum scan handles up to 1024 tiles of 4096 cells
    const size_t stride = (sgr_knn_workspace_bytes(P, max_cells) + 255) & ~(size_t)255;
    if (workspace_bytes < stride * (size_t)n_sets) { sgr_set_error("sgr_knn_dist2: workspace too small"); return 1; }
    hipStream_t stream = (hipStream_t)stream_;
    KnnBatch kb;
    kb.P = P; kb.max_cells = max_cells; kb.points = points; kb.out = out_dist2; kb.ws = (char *)workspace; kb.ws_stride = stride;
    SgrProfScope _p(SGR_K_KNN, stream);
    const int nb = (P + kT - 1) / kT;
    const int scan_blocks = (max_cells + 1 + kScanTile - 1) / kScanTile;
    const size_t init_want = ((size_t)max_cells + 1 + kT * 4 - 1) / (kT * 4);
    const int init_blocks = (int)(init_want < 1024 ? init_want : 1024);
    const int box_blocks = min(nb, kBoxBlocks);                       // (6 floats each at the head of `sorted`: sgr_knn_workspace_bytes leaves room for them)
    hipLaunchKernelGGL(knn_prep_kernel, dim3(max(init_blocks, box_blocks), n_sets), dim3(kT), 0, stream, kb, box_blocks);
    hipLaunchKernelGGL(cell_count_kernel, dim3(nb, n_sets), dim3(kT), 0, stream, kb, box_blocks);
    hipLaunchKernelGGL(cell_blocksum_kernel, dim3(scan_blocks, n_sets), dim3(kT), 0, stream, kb);
    hipLaunchKernelGGL(cell_scan_kernel, dim3(scan_blocks, n_sets), dim3(kT), 0, stream, kb);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(nb, n_sets), dim3(kT), 0, stream, kb);
    hipLaunchKernelGGL(knn3_kernel, dim3((unsigned)(((size_t)P * kKnnLanes + kT - 1) / kT), n_sets), dim3(kT), 0, stream, kb);
    SGR_CHECK_LAUNCH("knn kernels");
    return 0;
}

extern "C" int sgr_knn_dist2(int32_t P, const float *points, float *out_dist2, void *workspace, size_t workspace_bytes,
                             int32_t max_cells, void *stream_) {
    return sgr_knn_dist2_batched(1, P, points, out_dist2, workspace, workspace_bytes, max_cells, stream_);
}

extern "C" int sgr_cov3d_forward(int32_t n, const float *scale_raw, const float *rotation, const float *dist2, float *cov6,
                                 void *stream_) {
    if (n <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    SgrProfScope _p(SGR_K_COV3D, stream);
    hipLaunchKernelGGL(cov3d_fwd_kernel, dim3((n + kT - 1) / kT), dim3(kT), 0, stream, n, scale_raw, rotation, dist2, cov6);
    SGR_CHECK_LAUNCH("cov3d_fwd_kernel");
    return 0;
}

extern "C" int sgr_cov3d_backward(int32_t n, const float *scale_raw, const float *rotation, const float *dist2,
                                  const float *grad_cov6, float *grad_scale_raw, float *grad_rotation, void *stream_) {
    if (n <= 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    SgrProfScope _p(SGR_K_COV3D, stream);
    hipLaunchKernelGGL(cov3d_bwd_kernel, dim3((n + kT - 1) / kT), dim3(kT), 0, stream, n, scale_raw, rotation, dist2, grad_cov6,
                       grad_scale_raw, grad_rotation);
    SGR_CHECK_LAUNCH("cov3d_bwd_kernel");
    return 0;
}
