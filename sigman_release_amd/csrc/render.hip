// render.hip -- per-tile compositing kernels for gfx950 (CDNA4, wave64):
//   F6  front-to-back alpha compositing -> color, depth, alpha (+ final_T, n_contrib)   (render_fwd_kernel)
//   B1  per-pixel reverse walk -> per-(view,Gaussian) gradient records                  (render_bwd_kernel)
// Replaces renderCUDA forward/backward of the third-party rasterizer behind
// /root/reference/core/gaussians/gs.py:98-106 and train_vae.py:166, for all views of a batch in ONE launch
// (grid = n_views * tiles; the reference issues B*V separate launch chains, gs.py:62,75).
//
// Mapping: one workgroup = one 16x16 tile = 4 waves; wave w owns the 8x8 quadrant (w&1, w>>1) so that a
// wave's 64 pixels are spatially compact (early termination and sub-tile culling are decided per wave).
// Per 256-Gaussian batch the workgroup gathers the packed 48-B records (3 coalesced 16-B loads per thread)
// into LDS once; each wave then builds, with __ballot over an exact per-Gaussian bounding test, the list of
// Gaussians that can reach alpha >= 1/255 anywhere in ITS quadrant and walks only those (scalar bit loop,
// LDS broadcast reads).  Culled Gaussians would have hit the published `alpha < 1/255 -> continue` rule for
// every pixel of the quadrant, so results (including n_contrib) are unchanged by the cull.
//
// Roofline: algorithmic HBM bytes are 44 B per tile instance + 24 B (fwd) / 28 B (bwd) per pixel
// (SURVEY.md 8d); with 3-4 px splats the inner loop is VALU-bound, not HBM-bound -- see DESIGN.md.
#include "common.h"

namespace {

constexpr int kBlock = 256;

struct Quad {
    uint32_t view, tile, tx, ty;
    int px, py;
    bool inside;
};

__device__ __forceinline__ uint32_t cull_mask(const float4 &a, const float4 &c, float x0, float y0) {
    const float gx = a.x, gy = a.y, hx = c.z, hy = c.w;
    if (hx < 0.f) return 0u;                       // opacity <= 1/255: can never pass the alpha floor
    const float lox = gx - hx, hix = gx + hx, loy = gy - hy, hiy = gy + hy;
    const bool xl = (hix >= x0) && (lox <= x0 + 7.f);
    const bool xr = (hix >= x0 + 8.f) && (lox <= x0 + 15.f);
    const bool yt = (hiy >= y0) && (loy <= y0 + 7.f);
    const bool yb = (hiy >= y0 + 8.f) && (loy <= y0 + 15.f);
    return (uint32_t)(xl && yt) | ((uint32_t)(xr && yt) << 1) | ((uint32_t)(xl && yb) << 2) | ((uint32_t)(xr && yb) << 3);
}

// -------------------------------------------------------------------------------------------------
// F6
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void render_fwd_kernel(int W, int H, int Tx, uint32_t tiles_per_view,
                                                            const uint2 *__restrict__ ranges,
                                                            const uint32_t *__restrict__ point_list,
                                                            const float4 *__restrict__ rec, const float *__restrict__ bg,
                                                            float *__restrict__ out_color, float *__restrict__ out_depth,
                                                            float *__restrict__ out_alpha, float *__restrict__ final_T,
                                                            uint32_t *__restrict__ n_contrib) {
    __shared__ float4 sA[kBlock], sB[kBlock], sC[kBlock];
    __shared__ uint32_t sMask[kBlock];
    const uint32_t bid = sgr_xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t view = bid / tiles_per_view, tile = bid - view * tiles_per_view;
    const uint32_t tx = tile % Tx, ty = tile / Tx;
    const uint2 range = ranges[bid];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int px = (int)tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = (int)ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)(tx * 16), y0 = (float)(ty * 16);
    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    const int n = (int)(range.y - range.x);
    const int rounds = (n + kBlock - 1) / kBlock;
    for (int r = 0; r < rounds; r++) {
        if (__syncthreads_count(done) == kBlock) break;      // also the barrier that protects LDS reuse
        const int idx = r * kBlock + t;
        uint32_t m = 0;
        if (idx < n) {
            const size_t id = point_list[range.x + idx];
            const float4 a = rec[id * 3 + 0], b = rec[id * 3 + 1], c = rec[id * 3 + 2];
            sA[t] = a; sB[t] = b; sC[t] = c;
            m = cull_mask(a, c, x0, y0);
        }
        sMask[t] = m;
        __syncthreads();
        uint64_t active = __ballot(!done);
        for (int ch = 0; ch < 4 && active; ch++) {
            uint64_t bal = __ballot((sMask[ch * 64 + lane] >> wave) & 1u);
            while (bal && active) {
                const int bit = __builtin_ctzll(bal);
                bal &= bal - 1;
                const int j = ch * 64 + bit;
                const float4 a = sA[j], b = sB[j];
                const float4 c = sC[j];
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float alpha = fminf(0.99f, b.y * __expf(power));
                if (!done && power <= 0.f && alpha >= (1.0f / 255.0f)) {
                    const float test_T = T * (1.f - alpha);
                    if (test_T < 0.0001f) {
                        done = true;                          // the crossing Gaussian is NOT composited
                    } else {
                        const float w = alpha * T;
                        C0 += b.w * w; C1 += c.x * w; C2 += c.y * w;
                        D += b.z * w;
                        A += w;
                        T = test_T;
                        last = (uint32_t)(r * kBlock + j + 1);
                    }
                }
                active = __ballot(!done);
            }
        }
    }
    if (inside) {
        const size_t hw = (size_t)H * W;
        const size_t pix = (size_t)py * W + px;
        const size_t vb = (size_t)view * hw;
        final_T[vb + pix] = T;
        n_contrib[vb + pix] = last;
        out_color[(vb * 3) + pix] = C0 + T * bg[0];
        out_color[(vb * 3) + hw + pix] = C1 + T * bg[1];
        out_color[(vb * 3) + 2 * hw + pix] = C2 + T * bg[2];
        out_depth[vb + pix] = D;
        out_alpha[vb + pix] = A;
    }
}

// -------------------------------------------------------------------------------------------------
// B1 (v1): pixel-parallel reverse walk; per visited Gaussian the 10 partials are reduced across the wave
// with DPP adds (no LDS traffic), accumulated per batch in LDS, and flushed with ONE set of hardware float
// atomics per (tile, Gaussian) instead of one per (pixel, Gaussian) as in the published kernel.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void render_bwd_kernel(int W, int H, int Tx, uint32_t tiles_per_view,
                                                            const uint2 *__restrict__ ranges,
                                                            const uint32_t *__restrict__ point_list,
                                                            const float4 *__restrict__ rec, const float *__restrict__ bg,
                                                            const float *__restrict__ final_T,
                                                            const uint32_t *__restrict__ n_contrib,
                                                            const float *__restrict__ gC, const float *__restrict__ gD,
                                                            const float *__restrict__ gA, float *__restrict__ grec) {
    __shared__ float4 sA[kBlock], sB[kBlock], sC[kBlock];
    __shared__ uint32_t sMask[kBlock];
    __shared__ uint32_t sId[kBlock];
    __shared__ float sGrad[10][kBlock];
    __shared__ uint32_t sMax[4];
    const uint32_t bid = sgr_xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t view = bid / tiles_per_view, tile = bid - view * tiles_per_view;
    const uint32_t tx = tile % Tx, ty = tile / Tx;
    const uint2 range = ranges[bid];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int px = (int)tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = (int)ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float x0 = (float)(tx * 16), y0 = (float)(ty * 16);
    const size_t hw = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const size_t vb = (size_t)view * hw;
    const float Tf = inside ? final_T[vb + pix] : 0.f;
    const uint32_t last = inside ? n_contrib[vb + pix] : 0u;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f, ga = 0.f;
    if (inside) {
        g0 = gC[vb * 3 + pix]; g1 = gC[vb * 3 + hw + pix]; g2 = gC[vb * 3 + 2 * hw + pix];
        if (gD) gd = gD[vb + pix];
        if (gA) ga = gA[vb + pix];
    }
    const float bg_dot = (bg[0] * g0 + bg[1] * g1) + bg[2] * g2;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    // only the first max(n_contrib) entries of the tile list can receive gradient
    uint32_t wmax = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, off, 64));
    if (lane == 0) sMax[wave] = wmax;
    __syncthreads();
    const uint32_t bmax = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));
    const int n = (int)min(range.y - range.x, bmax);
    const int rounds = (n + kBlock - 1) / kBlock;
    float T = Tf;
    float accC0 = 0.f, accC1 = 0.f, accC2 = 0.f, accD = 0.f, accA = 0.f;
    float last_alpha = 0.f, lastC0 = 0.f, lastC1 = 0.f, lastC2 = 0.f, lastD = 0.f;
    for (int r = rounds - 1; r >= 0; r--) {
        __syncthreads();
        const int idx = r * kBlock + t;
        uint32_t m = 0;
        if (idx < n) {
            const uint32_t id = point_list[range.x + idx];
            const float4 a = rec[(size_t)id * 3 + 0], b = rec[(size_t)id * 3 + 1], c = rec[(size_t)id * 3 + 2];
            sA[t] = a; sB[t] = b; sC[t] = c;
            sId[t] = id;
            m = cull_mask(a, c, x0, y0);
        }
        sMask[t] = m;
#pragma unroll
        for (int k = 0; k < 10; k++) sGrad[k][t] = 0.f;
        __syncthreads();
        for (int ch = 3; ch >= 0; ch--) {
            if ((uint32_t)(r * kBlock + ch * 64) >= wmax) continue;      // nothing in this chunk precedes any pixel's last contributor
            uint64_t bal = __ballot((sMask[ch * 64 + lane] >> wave) & 1u);
            while (bal) {
                const int bit = 63 - __builtin_clzll(bal);
                bal &= ~(1ull << bit);
                const int j = ch * 64 + bit;
                const uint32_t contributor = (uint32_t)(r * kBlock + j);    // 0-based position in the tile list
                const float4 a = sA[j], b = sB[j];
                const float4 c = sC[j];
                const float dx = a.x - pxf, dy = a.y - pyf;
                const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, b.y * G);
                const bool valid = (contributor < last) && power <= 0.f && alpha >= (1.0f / 255.0f);
                if (!__ballot(valid)) continue;
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f, v9 = 0.f;
                if (valid) {
                    T = T / (1.f - alpha);
                    const float w = alpha * T;
                    float dL_dalpha;
                    accC0 = last_alpha * lastC0 + (1.f - last_alpha) * accC0; lastC0 = b.w;
                    dL_dalpha = (b.w - accC0) * g0;
                    accC1 = last_alpha * lastC1 + (1.f - last_alpha) * accC1; lastC1 = c.x;
                    dL_dalpha += (c.x - accC1) * g1;
                    accC2 = last_alpha * lastC2 + (1.f - last_alpha) * accC2; lastC2 = c.y;
                    dL_dalpha += (c.y - accC2) * g2;
                    accD = last_alpha * lastD + (1.f - last_alpha) * accD; lastD = b.z;
                    dL_dalpha += (b.z - accD) * gd;
                    accA = last_alpha + (1.f - last_alpha) * accA;
                    dL_dalpha += (1.f - accA) * ga;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-Tf / (1.f - alpha)) * bg_dot;
                    const float dL_dG = b.y * dL_dalpha;           // differentiates through op*G even when capped (as upstream)
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * a.z - gdy * a.w;
                    const float dG_ddely = -gdy * b.x - gdx * a.w;
                    v0 = dL_dG * dG_ddelx * ddelx_dx;
                    v1 = dL_dG * dG_ddely * ddely_dy;
                    v2 = -0.5f * gdx * dx * dL_dG;
                    v3 = -0.5f * gdx * dy * dL_dG;
                    v4 = -0.5f * gdy * dy * dL_dG;
                    v5 = G * dL_dalpha;
                    v6 = w * gd;
                    v7 = w * g0; v8 = w * g1; v9 = w * g2;
                }
                v0 = sgr_wave_sum_to_lane63(v0); v1 = sgr_wave_sum_to_lane63(v1); v2 = sgr_wave_sum_to_lane63(v2);
                v3 = sgr_wave_sum_to_lane63(v3); v4 = sgr_wave_sum_to_lane63(v4); v5 = sgr_wave_sum_to_lane63(v5);
                v6 = sgr_wave_sum_to_lane63(v6); v7 = sgr_wave_sum_to_lane63(v7); v8 = sgr_wave_sum_to_lane63(v8);
                v9 = sgr_wave_sum_to_lane63(v9);
                if (lane == 63) {
                    sgr_atomic_add(&sGrad[0][j], v0); sgr_atomic_add(&sGrad[1][j], v1); sgr_atomic_add(&sGrad[2][j], v2);
                    sgr_atomic_add(&sGrad[3][j], v3); sgr_atomic_add(&sGrad[4][j], v4); sgr_atomic_add(&sGrad[5][j], v5);
                    sgr_atomic_add(&sGrad[6][j], v6); sgr_atomic_add(&sGrad[7][j], v7); sgr_atomic_add(&sGrad[8][j], v8);
                    sgr_atomic_add(&sGrad[9][j], v9);
                }
            }
        }
        __syncthreads();
        if (idx < n && sMask[t]) {
            float *g = grec + (size_t)sId[t] * SGR_REC_FLOATS;
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const float v = sGrad[k][t];
                if (v != 0.f) sgr_atomic_add(g + k, v);
            }
        }
    }
}

}  // namespace

int sgr_validate_problem(const SgrProblem *pb);

extern "C" int sgr_render_forward(const SgrProblem *pb, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                                  float *out_color, float *out_depth, float *out_alpha, float *final_T, uint32_t *n_contrib,
                                  void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint32_t tiles = (uint32_t)Tx * Ty;
    hipStream_t stream = (hipStream_t)stream_;
    SgrProfScope _p(SGR_K_RENDER_FWD, stream);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(tiles * pb->n_views), dim3(kBlock), 0, stream, pb->W, pb->H, Tx, tiles,
                       (const uint2 *)ranges, point_list, (const float4 *)rec, pb->bg, out_color, out_depth, out_alpha, final_T,
                       n_contrib);
    SGR_CHECK_LAUNCH("render_fwd_kernel");
    return 0;
}

extern "C" int sgr_render_backward(const SgrProblem *pb, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                                   const float *final_T, const uint32_t *n_contrib, const float *grad_color,
                                   const float *grad_depth, const float *grad_alpha, float *grec, void *stream_) {
    if (sgr_validate_problem(pb)) return 1;
    hipStream_t stream = (hipStream_t)stream_;
    const int Tx = (pb->W + SGR_TILE - 1) / SGR_TILE, Ty = (pb->H + SGR_TILE - 1) / SGR_TILE;
    const uint32_t tiles = (uint32_t)Tx * Ty;
    if (pb->P > 0)
        SGR_CHECK_HIP(hipMemsetAsync(grec, 0, (size_t)pb->n_views * pb->P * SGR_REC_FLOATS * sizeof(float), stream));
    SgrProfScope _p(SGR_K_RENDER_BWD, stream);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(tiles * pb->n_views), dim3(kBlock), 0, stream, pb->W, pb->H, Tx, tiles,
                       (const uint2 *)ranges, point_list, (const float4 *)rec, pb->bg, final_T, n_contrib, grad_color, grad_depth,
                       grad_alpha, grec);
    SGR_CHECK_LAUNCH("render_bwd_kernel");
    return 0;
}
