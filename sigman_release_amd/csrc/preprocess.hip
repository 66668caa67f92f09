// preprocess.hip -- per-Gaussian stages of the rasterizer for gfx950:
//   F1  cull / project / 2D covariance / conic / radius / tile rectangle   (sgr_preprocess_forward)
//   F2  tile-count scan (block partial sums + single-block exclusive scan)
//   B2+B3  gradient records -> parameter gradients, summed over views in a fixed order (no atomics)
//   mark_visible
// Replaces the per-view preprocess of the third-party rasterizer the reference calls at
// /root/reference/core/gaussians/gs.py:98-106 (and its backward reached from train_vae.py:166).
//
// THIS TRANSLATION UNIT IS COMPILED WITH -ffp-contract=off.  Every expression that feeds an integer
// artefact (depth key bits, radius, tile rectangle) is written in the canonical left-to-right fp32
// order documented in DESIGN.md, with correctly rounded div/sqrt (hipcc default), so those artefacts
// are bit-exact against the CPU oracle.  These kernels are HBM-streaming (~76 B read+written per
// Gaussian per view forward, ~150 B backward); one thread per Gaussian, view index on blockIdx.y so
// the camera matrices are wave-uniform scalar loads.
#include <atomic>
#include <string.h>
#include "common.h"

namespace {

constexpr int kPreThreads = 256;


struct Cov2D {
    float t[3];
    float xmul, ymul;
    float j00, j02, j11, j12;
    float m0[3], m1[3];
    float v0[3], v1[3];
    float a, b, c;
};

__device__ __forceinline__ void xform4x3(const float *m, const float *p, float *o) {
    o[0] = ((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2]) + m[12];
    o[1] = ((m[1] * p[0] + m[5] * p[1]) + m[9] * p[2]) + m[13];
    o[2] = ((m[2] * p[0] + m[6] * p[1]) + m[10] * p[2]) + m[14];
}
__device__ __forceinline__ void xform4x4(const float *m, const float *p, float *o) {
    o[0] = ((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2]) + m[12];
    o[1] = ((m[1] * p[0] + m[5] * p[1]) + m[9] * p[2]) + m[13];
    o[2] = ((m[2] * p[0] + m[6] * p[1]) + m[10] * p[2]) + m[14];
    o[3] = ((m[3] * p[0] + m[7] * p[1]) + m[11] * p[2]) + m[15];
}
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

// -------------------------------------------------------------------------------------------------
// The alpha test as a threshold on the exponent (render.hip, file header).
// sgr_exp2_cr: 2^x rounded to fp32 from an fp64 evaluation whose error (~1e-16) is far below half an fp32 ulp: n = rint(x),
// e^((x - n) ln 2) by a degree-13 Taylor polynomial in Horner form.  Every step is ONE correctly rounded IEEE operation (cvt, rint, sub,
// mul, fma, ldexp, cvt) and the CPU oracle (its ref_exp2_cr) performs the same steps in the same order: bit-identical on both sides.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgr_exp2_cr(float x) {
    const double xd = (double)x;
    const double n = rint(xd);
    const double t = (xd - n) * 0.693147180559945309417232121458;
    double p = 1.0 / 6227020800.0;
    p = fma(p, t, 1.0 / 479001600.0);
    p = fma(p, t, 1.0 / 39916800.0);
    p = fma(p, t, 1.0 / 3628800.0);
    p = fma(p, t, 1.0 / 362880.0);
    p = fma(p, t, 1.0 / 40320.0);
    p = fma(p, t, 1.0 / 5040.0);
    p = fma(p, t, 1.0 / 720.0);
    p = fma(p, t, 1.0 / 120.0);
    p = fma(p, t, 1.0 / 24.0);
    p = fma(p, t, 1.0 / 6.0);
    p = fma(p, t, 0.5);
    p = fma(p, t, 1.0);
    p = fma(p, t, 1.0);
    return (float)ldexp(p, (int)n);
}
// the published test at exponent -|p| (mag = bits of |p|), evaluated like the CPU oracle does: power2 <= 0 and min(0.99, op * 2^power2) >= 1/255
__device__ __forceinline__ bool alpha_passes(float op, uint32_t mag) {
    return fminf(0.99f, op * sgr_exp2_cr(-__uint_as_float(mag))) >= (1.0f / 255.0f);
}
// p* = the SMALLEST exponent (exp2 domain, <= 0) at which a Gaussian of opacity `op` passes the alpha test; +inf if it never does.
// alpha is monotone in the exponent, so  alpha >= 1/255  <=>  power2 >= p*  for every power2 <= 0.  Bisection over the bit patterns of
// |p| between a bracket around log2(255 op) (hardware log: only an estimate) -- widened to all of [0, 127] if the estimate is off.
__device__ __forceinline__ float sgr_alpha_threshold(float op) {
    if (!(op >= (1.0f / 255.0f))) return __builtin_inff();      // (NaN included) op * G <= op < 1/255 for every G <= 1
    uint32_t lo = 0u, hi = 0x42FE0000u;                         // |p| = 0 passes (alpha = min(0.99, op)); |p| = 127 never does
    const float est = fminf(fmaxf(__log2f(op * 255.0f), 0.f), 126.f);
    const uint32_t e = __float_as_uint(est), a = e > 8u ? e - 8u : 0u, b = e + 8u;
    if (alpha_passes(op, a)) { lo = a; if (!alpha_passes(op, b)) hi = b; else lo = b; } else hi = a;
    while (hi - lo > 1u) {                                      // invariant: lo passes, hi does not
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (alpha_passes(op, mid)) lo = mid; else hi = mid;
    }
    return -__uint_as_float(lo);
}

__device__ __forceinline__ void quat_to_rot(const float *q, float R[3][3]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = R diag(mod*s)^2 R^T, packed xx,xy,xz,yy,yz,zz
__device__ __forceinline__ void cov3d_from_scale_rot(const float *s, float mod, const float *q, float *cov) {
    float R[3][3];
    quat_to_rot(q, R);
    const float sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float Mx[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mx[i][k] = R[i][k] * sv[k];
    float S[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) S[i][j] = (Mx[i][0] * Mx[j][0] + Mx[i][1] * Mx[j][1]) + Mx[i][2] * Mx[j][2];
    cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2];
    cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
}

__device__ __forceinline__ void load_cov3d(const SgrProblem &pb, size_t sp /* s*P+i */, float *c6) {
    if (pb.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = pb.cov3D_precomp[sp * 6 + k];   // scale_modifier NOT applied (as upstream)
    } else {
        float s[3], q[4];
#pragma unroll
        for (int k = 0; k < 3; k++) s[k] = pb.scales[sp * 3 + k];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = pb.rotations[sp * 4 + k];
        cov3d_from_scale_rot(s, pb.scale_modifier, q, c6);
    }
}

__device__ __forceinline__ void cov2d_eval(const float *pview, const float *V, const float *cov6, float fx, float fy,
                                           float tanfovx, float tanfovy, Cov2D &o) {
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float tz = pview[2];
    const float txtz = pview[0] / tz, tytz = pview[1] / tz;
    o.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    o.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    o.t[0] = tx; o.t[1] = ty; o.t[2] = tz;
    o.j00 = fx / tz; o.j02 = -(fx * tx) / (tz * tz);
    o.j11 = fy / tz; o.j12 = -(fy * ty) / (tz * tz);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float r0 = V[4 * k + 0], r1 = V[4 * k + 1], r2 = V[4 * k + 2];
        o.m0[k] = o.j00 * r0 + o.j02 * r2;
        o.m1[k] = o.j11 * r1 + o.j12 * r2;
    }
    const float S[3][3] = {{cov6[0], cov6[1], cov6[2]}, {cov6[1], cov6[3], cov6[4]}, {cov6[2], cov6[4], cov6[5]}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        o.v0[i] = (S[i][0] * o.m0[0] + S[i][1] * o.m0[1]) + S[i][2] * o.m0[2];
        o.v1[i] = (S[i][0] * o.m1[0] + S[i][1] * o.m1[1]) + S[i][2] * o.m1[2];
    }
    o.a = ((o.m0[0] * o.v0[0] + o.m0[1] * o.v0[1]) + o.m0[2] * o.v0[2]) + 0.3f;
    o.b = (o.m0[0] * o.v1[0] + o.m0[1] * o.v1[1]) + o.m0[2] * o.v1[2];
    o.c = ((o.m1[0] * o.v1[0] + o.m1[1] * o.v1[1]) + o.m1[2] * o.v1[2]) + 0.3f;
}

// real SH basis (degree <= 3) and its gradient w.r.t. the unit direction
__device__ __forceinline__ int sh_basis(int deg, const float *d, float *B) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const float x = d[0], y = d[1], z = d[2];
    B[0] = C0;
    if (deg < 1) return 1;
    B[1] = -C1 * y; B[2] = C1 * z; B[3] = -C1 * x;
    if (deg < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = C2[0] * xy; B[5] = C2[1] * yz; B[6] = C2[2] * (2.f * zz - xx - yy);
    B[7] = C2[3] * xz; B[8] = C2[4] * (xx - yy);
    if (deg < 3) return 9;
    B[9] = C3[0] * y * (3.f * xx - yy); B[10] = C3[1] * xy * z;
    B[11] = C3[2] * y * (4.f * zz - xx - yy); B[12] = C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    B[13] = C3[4] * x * (4.f * zz - xx - yy); B[14] = C3[5] * z * (xx - yy);
    B[15] = C3[6] * x * (xx - 3.f * yy);
    return 16;
}
__device__ __forceinline__ void sh_basis_grad(int deg, const float *d, float G[16][3]) {
    const float C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) G[k][0] = G[k][1] = G[k][2] = 0.f;
    if (deg < 1) return;
    G[1][1] = -C1; G[2][2] = C1; G[3][0] = -C1;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z;
    G[4][0] = C2[0] * y; G[4][1] = C2[0] * x;
    G[5][1] = C2[1] * z; G[5][2] = C2[1] * y;
    G[6][0] = C2[2] * (-2.f * x); G[6][1] = C2[2] * (-2.f * y); G[6][2] = C2[2] * (4.f * z);
    G[7][0] = C2[3] * z; G[7][2] = C2[3] * x;
    G[8][0] = C2[4] * (2.f * x); G[8][1] = C2[4] * (-2.f * y);
    if (deg < 3) return;
    G[9][0] = C3[0] * (6.f * x * y); G[9][1] = C3[0] * (3.f * xx - 3.f * yy);
    G[10][0] = C3[1] * y * z; G[10][1] = C3[1] * x * z; G[10][2] = C3[1] * x * y;
    G[11][0] = C3[2] * (-2.f * x * y); G[11][1] = C3[2] * (4.f * zz - xx - 3.f * yy); G[11][2] = C3[2] * (8.f * y * z);
    G[12][0] = C3[3] * (-6.f * x * z); G[12][1] = C3[3] * (-6.f * y * z); G[12][2] = C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
    G[13][0] = C3[4] * (4.f * zz - 3.f * xx - yy); G[13][1] = C3[4] * (-2.f * x * y); G[13][2] = C3[4] * (8.f * x * z);
    G[14][0] = C3[5] * (2.f * x * z); G[14][1] = C3[5] * (-2.f * y * z); G[14][2] = C3[5] * (xx - yy);
    G[15][0] = C3[6] * (3.f * xx - 3.f * yy); G[15][1] = C3[6] * (-6.f * x * y);
}

// block-wide sum of one u32 per thread (256 threads = 4 waves); result valid in thread 0
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// -------------------------------------------------------------------------------------------------
// F1
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kPreThreads) void preprocess_fwd_kernel(SgrProblem pb, int views_per_wg, float4 *__restrict__ rec,
                                                                     int32_t *__restrict__ radii,
                                                                     uint4 *__restrict__ rect,
                                                                     uint8_t *__restrict__ clamped,
                                                                     uint32_t *__restrict__ block_sums, int nbx, SgrBgJob bg) {
    if (bg.zero_words && blockIdx.y == 0)
        for (uint32_t z = blockIdx.x * kPreThreads + threadIdx.x; z < bg.zero_words; z += gridDim.x * kPreThreads) bg.zero_ptr[z] = 0u;
    // (workgroups behind the nbx that own Gaussians: the background pre-fill of the fused single-view step, common.h SgrBgJob)
    if ((int)blockIdx.x >= nbx) { if (blockIdx.y == 0 && bg.enabled) sgr_bg_fill(bg, blockIdx.x - (uint32_t)nbx, gridDim.x - (uint32_t)nbx); return; }
    // One thread per Gaussian, looping over `views_per_wg` consecutive views: the per-Gaussian inputs (52 B: mean, covariance, opacity,
    // colour) stay in registers across the views of a subject.  With one view per workgroup a 90-view launch re-read them from HBM
    // 90 times (PMC at C4: 0.94 GB of reads beside 1.5 GB of writes).
    __shared__ uint32_t red[2][4];
    __shared__ float4 stage[kPreThreads / 64][256];
    const int i = blockIdx.x * kPreThreads + threadIdx.x;
    const int W = pb.W, H = pb.H;
    const int Tx = (W + SGR_TILE - 1) / SGR_TILE, Ty = (H + SGR_TILE - 1) / SGR_TILE;
    const int v0 = blockIdx.y * views_per_wg, v1 = min(v0 + views_per_wg, pb.n_views);
    const bool live = i < pb.P;
    int cur_subj = -1;
    float p[3] = {0.f, 0.f, 0.f}, c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, op = 0.f, rgb_in[3] = {0.f, 0.f, 0.f};
    float pstar = __builtin_inff();                           // alpha-test threshold on the exponent: a function of the opacity alone
    for (int view = v0; view < v1; view++) {
    const int subj = view / pb.views_per_subject;
    const float *V = pb.viewmatrix + 16 * (size_t)view;
    const float *M = pb.projmatrix + 16 * (size_t)view;
    uint32_t tiles = 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = make_float4(0.f, 0.f, __uint_as_float(0xBF80BF80u), __builtin_inff());   // (half extents -1 | -1: never visible; p* = +inf: never valid)
    if (live) {
        const size_t q = (size_t)view * pb.P + i, sp = (size_t)subj * pb.P + i;
        if (subj != cur_subj) {                              // uniform: once per subject
            cur_subj = subj;
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] = pb.means3D[sp * 3 + k];
            load_cov3d(pb, sp, c6);
            op = pb.opacities[sp];
            pstar = sgr_alpha_threshold(op);
            if (pb.colors_precomp) {
#pragma unroll
                for (int k = 0; k < 3; k++) rgb_in[k] = pb.colors_precomp[sp * 3 + k];   // untouched, no clamp
            }
        }
        int32_t rad_out = 0;
        uint2 rect_out = make_uint2(0u, 0u);
        uint8_t clamp_bits = 0;
        float pv[3];
        xform4x3(V, p, pv);
        if (pv[2] > 0.2f) {                                  // near cull; the lateral test is disabled upstream
            float ph[4];
            xform4x4(M, p, ph);
            const float pw = 1.0f / (ph[3] + 0.0000001f);
            const float projx = ph[0] * pw, projy = ph[1] * pw;
            const float fx = (float)W / (2.0f * pb.tanfovx), fy = (float)H / (2.0f * pb.tanfovy);
            Cov2D cq;
            cov2d_eval(pv, V, c6, fx, fy, pb.tanfovx, pb.tanfovy, cq);
            const float det = cq.a * cq.c - cq.b * cq.b;
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float cx = cq.c * det_inv, cy = -cq.b * det_inv, cz = cq.a * det_inv;
                const float mid = 0.5f * (cq.a + cq.c);
                const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lam1 = mid + disc, lam2 = mid - disc;
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
                const float px = ndc2pix(projx, W), py = ndc2pix(projy, H);
                const int rad = (int)my_radius;
                int minx = (int)((px - (float)rad) / (float)SGR_TILE); minx = min(Tx, max(0, minx));
                int miny = (int)((py - (float)rad) / (float)SGR_TILE); miny = min(Ty, max(0, miny));
                int maxx = (int)((px + (float)rad + (float)(SGR_TILE - 1)) / (float)SGR_TILE); maxx = min(Tx, max(0, maxx));
                int maxy = (int)((py + (float)rad + (float)(SGR_TILE - 1)) / (float)SGR_TILE); maxy = min(Ty, max(0, maxy));
                const int area = (maxx - minx) * (maxy - miny);
                // rad > 0: a non-finite covariance (a diverged decoder; a lone point's infinite 3-NN distance) gives a NaN radius, which converts to 0:
                // the emission kernel skips radius-0 splats (`radii > 0`, as upstream's duplicateWithKeys does), so such a splat must not count
                // tiles either -- upstream's preprocess does count them, and the key slots nobody writes then reach its sort uninitialised; here
                // they sent a garbage tile id into the binning and, every few runs, a memory fault (found by tools/fuzz_determinism.py)
                if (area != 0 && rad > 0) {
                    float rgb[3];
                    if (pb.colors_precomp) {
#pragma unroll
                        for (int k = 0; k < 3; k++) rgb[k] = rgb_in[k];
                    } else {
                        const float *cp = pb.campos + 3 * (size_t)view;
                        float d[3] = {p[0] - cp[0], p[1] - cp[1], p[2] - cp[2]};
                        const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
                        d[0] /= len; d[1] /= len; d[2] /= len;
                        float B[16];
                        const int nb = sh_basis(pb.sh_degree, d, B);
                        const float *sh = pb.shs + sp * (size_t)pb.M * 3;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            float r = 0.f;
                            for (int k = 0; k < nb; k++) r += B[k] * sh[3 * k + ch];
                            r += 0.5f;
                            if (r < 0.f) clamp_bits |= (uint8_t)(1u << ch);
                            rgb[ch] = fmaxf(r, 0.f);
                        }
                    }
                    // exact sub-tile cull bound: alpha = op*exp(power) >= 1/255  <=>  d^T Q d <= 2 ln(255 op);
                    // the bounding box of that ellipse has half extents sqrt(tau*cov_xx), sqrt(tau*cov_yy).
                    // Inflated by 1e-3 relative + 0.02 px so fp32 rounding can never cull a contributing pixel.
                    float hx = -1.f, hy = -1.f;
                    if (!(det > 0.f) || !(cq.a > 0.f) || !(cq.c > 0.f)) {
                        hx = hy = 3.0e38f;                      // not an ellipse: never cull
                    } else if (op * 255.f > 1.f) {
                        const float tau = 2.f * logf(op * 255.f) * 1.001f + 1e-3f;
                        hx = sqrtf(tau * cq.a) * 1.001f + 0.02f;
                        hy = sqrtf(tau * cq.c) * 1.001f + 0.02f;
                    } else if (!(op * 255.f <= 1.f)) {
                        hx = hy = 3.0e38f;                      // NaN opacity: keep upstream behaviour
                    }
                    r0 = make_float4(px, py, cx, cy);
                    r1 = make_float4(cz, op, pv[2], rgb[0]);
                    // the two half extents travel as bf16 halves of ONE word, rounded UP (a cull bound may only grow: + <= 0.8 %), so that the
                    // record's twelfth float is free for p*: no fourth 16-byte load per tile instance in the compositing kernels
                    const uint32_t hxb = hx < 0.f ? 0xBF80u : (uint32_t)min(0x7F80u, (__float_as_uint(hx) + 0xFFFFu) >> 16);
                    const uint32_t hyb = hy < 0.f ? 0xBF80u : (uint32_t)min(0x7F80u, (__float_as_uint(hy) + 0xFFFFu) >> 16);
                    r2 = make_float4(rgb[1], rgb[2], __uint_as_float(hxb | (hyb << 16)), pstar);
                    rad_out = rad;
                    rect_out = make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)maxx | ((uint32_t)maxy << 16));
                    tiles = (uint32_t)area;
                }
            }
        }
        radii[q] = rad_out;
        // (rect min, rect max, depth key bits, first tile-instance index -- filled in by the emission kernel): everything the emission kernel and
        // the backward's gathers need besides the compositing record, in one coalesced 16-byte record (the emission kernel used to fetch the depth from
        // the 64-byte `rec` line of every visible Gaussian and to write the instance index into it: 128 B of traffic for 8 useful bytes)
        rect[q] = make_uint4(rect_out.x, rect_out.y, __float_as_uint(r1.z), 0u);
        if (clamped) clamped[q] = clamp_bits;
    }
    {   // the wave's 64 compositing records (4 KB, padding included) leave as four fully coalesced 1-KB stores: transposed through LDS
        // (slot swizzle keeps both the 64-B-strided writes and the contiguous reads conflict-free)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float4 *sw = stage[wave];
        const int sz = (lane >> 2) & 3;
        sw[lane * 4 + (0 ^ sz)] = r0; sw[lane * 4 + (1 ^ sz)] = r1; sw[lane * 4 + (2 ^ sz)] = r2;
        sw[lane * 4 + (3 ^ sz)] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int first = blockIdx.x * kPreThreads + wave * 64;
        const int nrec = pb.P - first;                       // records of this wave that exist (may be <= 0 or > 64)
        float4 *out = rec + ((size_t)view * pb.P + first) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = k * 64 + lane, r = j >> 2;
            if (r < nrec) out[j] = sw[r * 4 + ((j & 3) ^ ((r >> 2) & 3))];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t tot = block_sum_u32(tiles, red[(view - v0) & 1]);   // alternating slots: one barrier per view is enough
    if (threadIdx.x == 0) block_sums[(size_t)view * nbx + blockIdx.x] = tot;
    }
}

// -------------------------------------------------------------------------------------------------
// F2: exclusive scan of the per-block tile counts (n = n_views * ceil(P/256)), one workgroup per 8192 counts.
// out[0..n) = exclusive prefix, out[n] = total (also stored as u64 in *num_rendered).
// -------------------------------------------------------------------------------------------------
constexpr uint32_t kScanTile = 8192;                              // counts per workgroup of the scan (8 consecutive per thread)

__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                               uint32_t n, uint64_t *__restrict__ num_rendered,
                                                               unsigned long long capacity) {
    // One workgroup per tile of 8192 counts, no chaining: every workgroup first SUMS all counts before its tile (coalesced 16-byte
    // loads, independent, at most a few hundred KB out of L2), then scans its own tile -- the workgroups never wait for each other.
    // (One workgroup walking the tiles with a barrier each -- __syncthreads drains the vector-memory counter, so every tile paid the
    // load and the store latency -- took 77 us for the 70k counts of a 90-view launch; the first version, one long private chunk
    // per thread + a 20-barrier LDS scan, 155 us.)
    struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };
    __shared__ unsigned long long wbase[16], wtot[16];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t lo = blockIdx.x * kScanTile;
    // ---- own tile: loads first, they fly during the sum below
    uint32_t v[8];
    {
        const uint32_t k = lo + t * 8u;
        if (k + 8u <= n) {
            const U4 a = *reinterpret_cast<const U4 *>(in + k), b = *reinterpret_cast<const U4 *>(in + k + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = k + j < n ? in[k + j] : 0u;
        }
    }
    // ---- sum of everything before the tile (lo is a multiple of 8192: whole groups of four)
    unsigned long long acc = 0;
    const uint32_t groups = lo / 4u;
    uint32_t g = t;
    for (; g + 3u * 1024u < groups; g += 4u * 1024u) {
        const U4 a = *reinterpret_cast<const U4 *>(in + 4u * g), b = *reinterpret_cast<const U4 *>(in + 4u * (g + 1024u));
        const U4 c = *reinterpret_cast<const U4 *>(in + 4u * (g + 2048u)), d = *reinterpret_cast<const U4 *>(in + 4u * (g + 3072u));
        acc += ((unsigned long long)a.x + a.y) + ((unsigned long long)a.z + a.w);
        acc += ((unsigned long long)b.x + b.y) + ((unsigned long long)b.z + b.w);
        acc += ((unsigned long long)c.x + c.y) + ((unsigned long long)c.z + c.w);
        acc += ((unsigned long long)d.x + d.y) + ((unsigned long long)d.z + d.w);
    }
    for (; g < groups; g += 1024u) {
        const U4 a = *reinterpret_cast<const U4 *>(in + 4u * g);
        acc += ((unsigned long long)a.x + a.y) + ((unsigned long long)a.z + a.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    // ---- scan of the tile
    unsigned long long sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) sum += v[j];
    unsigned long long inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long nb = __shfl_up(inc, off, 64);
        if (lane >= (uint32_t)off) inc += nb;
    }
    if (lane == 63) wtot[wave] = inc;
    if (lane == 0) wbase[wave] = acc;
    __syncthreads();
    unsigned long long pre = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const unsigned long long x = wtot[w]; if (w < (int)wave) pre += x; all += x; pre += wbase[w]; all += wbase[w]; }
    unsigned long long run = pre + inc - sum;
    {
        const uint32_t k = lo + t * 8u;
        if (k + 8u <= n) {
            U4 a, b;
            a.x = (uint32_t)run; run += v[0]; a.y = (uint32_t)run; run += v[1]; a.z = (uint32_t)run; run += v[2]; a.w = (uint32_t)run; run += v[3];
            b.x = (uint32_t)run; run += v[4]; b.y = (uint32_t)run; run += v[5]; b.z = (uint32_t)run; run += v[6]; b.w = (uint32_t)run;
            *reinterpret_cast<U4 *>(out + k) = a; *reinterpret_cast<U4 *>(out + k + 4) = b;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) { if (k + j < n) out[k + j] = (uint32_t)run; run += v[j]; }
        }
    }
    if (t == 0 && blockIdx.x == gridDim.x - 1) {                  // the last tile knows the total
        out[n] = (uint32_t)all;
        const unsigned long long ovf = (all > 0xFFFFFFF0ull || all > capacity) ? 1ull : 0ull;
        num_rendered[0] = all;
        num_rendered[1] = ovf;
        num_rendered[2] = all | (ovf << 63);            // the one word the sync-free mode copies to the host
    }
}

// -------------------------------------------------------------------------------------------------
// B2 + B3.  The contribution of ONE view to the gradients of one Gaussian (bwd_view), then the sum over the subject's views in view
// order and the covariance -> (scale, rotation) chain (bwd_finish).  Two kernels share them:
//   preprocess_bwd_kernel        one thread per (subject, Gaussian), loops over the views (SH path; any views_per_subject)
//   preprocess_bwd_lanes_kernel  one thread per (view, Gaussian): the gather's dependent loads (rect -> flags -> partial records) of
//                                the 8 views of a training subject run side by side instead of one after the other; the per-view
//                                contributions meet in LDS and ONE thread per Gaussian adds them in view order -- the same additions
//                                in the same order as the loop, so both kernels give bit-identical gradients
// -------------------------------------------------------------------------------------------------
struct ViewGrad { float mean[3], cov[6], op, col[3]; };
// bwd_view<SH, CH>: CH items per round of the wave-wide gather (128: two per lane and round trip; 64 for the kernel whose occupancy LDS bounds); LDS per wave
constexpr uint32_t gather_lds_floats(uint32_t ch) { return ch * 10 + 128; }

// SH=false (the reference's colors_precomp path) compiles without the spherical-harmonics tables: no scratch, half the VGPRs
template <bool SH, uint32_t kGatherChunk>
__device__ __forceinline__ void bwd_view(const SgrProblem &pb, int view, int i, size_t sp, const float (&p)[3], const float (&c6)[6], float fx, float fy,
                                         const int32_t *__restrict__ radii, const uint8_t *__restrict__ clamped,
                                         const uint4 *__restrict__ rect, const float4 *__restrict__ part, const uint32_t *__restrict__ flags,
                                         uint32_t n_inst, const float *__restrict__ part_scale, float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dsh,
                                         ViewGrad &out, bool live /* false: a lane without a Gaussian -- EVERY lane of a wave must make this call (the
                                         gather is done by the whole wave) */, float *gather_lds /* this WAVE's gather_lds_floats(kGatherChunk) floats of LDS */) {
#pragma unroll
    for (int k = 0; k < 3; k++) { out.mean[k] = 0.f; out.col[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 6; k++) out.cov[k] = 0.f;
    out.op = 0.f;
    const size_t q = live ? (size_t)view * pb.P + i : 0;
    float *g2out = (dL_dmeans2D && live) ? dL_dmeans2D + q * 3 : nullptr;      // (NULL: nobody wants dL/dNDC)
    uint4 r3 = make_uint4(0u, 0u, 0u, 0u);
    bool vis = false;
    if (live) { r3 = rect[q]; vis = radii[q] > 0; }                 // (the rect is requested beside the radius, not behind it: one round trip less)
    float4 g0, g1, g2;
    {
        // deterministic gather of the bucket-parallel backward's partial records: one per (tile instance, quadrant),
        // summed in tile order then quadrant order -- no atomics anywhere in the backward
        const uint32_t off = r3.w, rmin = r3.x, rmax = r3.y;
        const uint32_t ntile_all = vis ? ((rmax & 0xFFFFu) - (rmin & 0xFFFFu)) * ((rmax >> 16) - (rmin >> 16)) : 0u;
        // ---- the gather, by the whole WAVE.  One thread walking its own n tile instances is a chain of n dependent round trips, as long as the
        // largest splat among 64 lanes (C1, 10 000 Gaussians of up to 7 x 7 tiles: 23 us, twice C2's time for a tenth of the Gaussians).  The
        // instances of the wave's 64 splats are numbered through (prefix over the lanes) and dealt to the lanes kGatherChunk at a time: lane l
        // fetches the flag word and the present quadrant records of items l, l + 64 of the chunk -- no dummy loads, two round trips per chunk --,
        // adds the quadrants in quadrant order and leaves the ten sums in LDS; the owner then adds its items in instance order.  The same
        // additions in the same order whatever the kernel, the lane or the run: bitwise reproducible, and one order for both gather kernels.
        g0 = make_float4(0.f, 0.f, 0.f, 0.f); g1 = g0; g2 = g0;
        struct __attribute__((packed, aligned(8))) Rec40 { float2 v[5]; };   // 40 bytes, 8-byte aligned: 16 + 16 + 8-byte loads
        {
            const uint32_t lane = sgr_lane_id();
            float *s_item = gather_lds;                                      // [kGatherChunk][10] sums of one item
            uint32_t *s_end = reinterpret_cast<uint32_t *>(gather_lds + kGatherChunk * 10), *s_off = s_end + 64;
            uint32_t inc = ntile_all;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(inc, o, 64); if (lane >= (uint32_t)o) inc += nb; }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63), my_begin = inc - ntile_all;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();        // (the previous view's reads of the tables are over)
            s_end[lane] = inc; s_off[lane] = off;
            float acc[10];
#pragma unroll
            for (int j = 0; j < 10; j++) acc[j] = 0.f;
            for (uint32_t base = 0; base < total; base += kGatherChunk) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                uint32_t inst[kGatherChunk / 64], fw[kGatherChunk / 64];
#pragma unroll
                for (uint32_t r = 0; r < kGatherChunk / 64; r++) {
                    const uint32_t j = base + r * 64u + lane;
                    inst[r] = 0xFFFFFFFFu; fw[r] = 0u;
                    if (j < total) {
                        uint32_t lo = 0;                                     // the owner: the number of lanes whose items end at or before j
#pragma unroll
                        for (uint32_t step = 32; step > 0; step >>= 1) if (s_end[lo + step - 1u] <= j) lo += step;
                        const uint32_t i2 = s_off[lo] + (j - (lo ? s_end[lo - 1u] : 0u));
                        if (i2 < n_inst) { inst[r] = i2; fw[r] = flags[i2]; }        // (instances beyond the buffers do not exist: sync-free mode after an overflow)
                    }
                }
#pragma unroll
                for (uint32_t r = 0; r < kGatherChunk / 64; r++) {
                    float it[10];
#pragma unroll
                    for (int j = 0; j < 10; j++) it[j] = 0.f;
                    if (fw[r]) {
                        const Rec40 *rb = reinterpret_cast<const Rec40 *>(part) + (size_t)inst[r] * 4;
                        Rec40 rr[4];
#pragma unroll
                        for (uint32_t qd = 0; qd < 4; qd++) {
#pragma unroll
                            for (int j = 0; j < 5; j++) rr[qd].v[j] = make_float2(0.f, 0.f);
                            if ((fw[r] >> (8 * qd)) & 0xFFu) rr[qd] = rb[qd];
                        }
#pragma unroll
                        for (uint32_t qd = 0; qd < 4; qd++)
#pragma unroll
                            for (int j = 0; j < 5; j++) { it[2 * j] += rr[qd].v[j].x; it[2 * j + 1] += rr[qd].v[j].y; }
                    }
                    float *dst = s_item + (r * 64u + lane) * 10u;
#pragma unroll
                    for (int j = 0; j < 10; j += 2) *reinterpret_cast<float2 *>(dst + j) = make_float2(it[j], it[j + 1]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t j0 = max(my_begin, base), j1 = min(inc, base + kGatherChunk);
                for (uint32_t j = j0; j < j1; j++) {
                    const float *src = s_item + (j - base) * 10u;
#pragma unroll
                    for (int k = 0; k < 10; k += 2) { const float2 v2 = *reinterpret_cast<const float2 *>(src + k); acc[k] += v2.x; acc[k + 1] += v2.y; }
                }
            }
            g0 = make_float4(acc[0], acc[1], acc[2], acc[3]); g1 = make_float4(acc[4], acc[5], acc[6], acc[7]); g2.x = acc[8]; g2.y = acc[9];
        }
        if (part_scale) {
            // the partial records of a fused rasterize + loss step are for dL/dloss = 1: everything below is linear in the ten sums
            const float sc = *part_scale;
            g0.x *= sc; g0.y *= sc; g0.z *= sc; g0.w *= sc; g1.x *= sc; g1.y *= sc; g1.z *= sc; g1.w *= sc; g2.x *= sc; g2.y *= sc;
        }
    }
    if (!vis) { if (g2out) { g2out[0] = g2out[1] = g2out[2] = 0.f; } return; }
    const float *V = pb.viewmatrix + 16 * (size_t)view;
    const float *M = pb.projmatrix + 16 * (size_t)view;
    float pv[3];
    xform4x3(V, p, pv);
    Cov2D cq;
    cov2d_eval(pv, V, c6, fx, fy, pb.tanfovx, pb.tanfovy, cq);
    // ---- B2: conic -> (a,b,c) -> Sigma, projection Jacobian
    const float a = cq.a, b = cq.b, c = cq.c;
    const float gx = g0.z, gy = g0.w, gz = g1.x;
    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (denom2inv != 0.f) {
        dL_da = denom2inv * (-c * c * gx + 2.f * b * c * gy + (denom - a * c) * gz);
        dL_dc = denom2inv * (-a * a * gz + 2.f * a * b * gy + (denom - a * c) * gx);
        dL_db = denom2inv * 2.f * (b * c * gx - (denom + 2.f * b * b) * gy + a * b * gz);
        const float *m0 = cq.m0, *m1 = cq.m1;
        out.cov[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
        out.cov[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
        out.cov[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
        out.cov[1] = 2.f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.f * m1[0] * m1[1] * dL_dc;
        out.cov[2] = 2.f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.f * m1[0] * m1[2] * dL_dc;
        out.cov[4] = 2.f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.f * m1[1] * m1[2] * dL_dc;
    }
    float gm0[3], gm1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        gm0[k] = 2.f * cq.v0[k] * dL_da + cq.v1[k] * dL_db;
        gm1[k] = 2.f * cq.v1[k] * dL_dc + cq.v0[k] * dL_db;
    }
    float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dJ00 += V[4 * k + 0] * gm0[k]; dJ02 += V[4 * k + 2] * gm0[k];
        dJ11 += V[4 * k + 1] * gm1[k]; dJ12 += V[4 * k + 2] * gm1[k];
    }
    const float tz = 1.f / cq.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = cq.xmul * -fx * tz2 * dJ02;
    const float dty = cq.ymul * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * cq.t[0]) * tz3 * dJ02 + (2.f * fy * cq.t[1]) * tz3 * dJ12;
    float gm[3];
#pragma unroll
    for (int k = 0; k < 3; k++) gm[k] = (V[4 * k + 0] * dtx + V[4 * k + 1] * dty) + V[4 * k + 2] * dtz;
    // ---- B3: NDC mean -> 3D mean through the full projection
    float ph[4];
    xform4x4(M, p, ph);
    const float mw = 1.0f / (ph[3] + 0.0000001f);
    const float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
    const float g2x = g0.x, g2y = g0.y;
    gm[0] += (M[0] * mw - M[3] * mul1) * g2x + (M[1] * mw - M[3] * mul2) * g2y;
    gm[1] += (M[4] * mw - M[7] * mul1) * g2x + (M[5] * mw - M[7] * mul2) * g2y;
    gm[2] += (M[8] * mw - M[11] * mul1) * g2x + (M[9] * mw - M[11] * mul2) * g2y;
    const float gdep = g1.z;
    gm[0] += V[2] * gdep; gm[1] += V[6] * gdep; gm[2] += V[10] * gdep;
    const float gc3[3] = {g1.w, g2.x, g2.y};
    if constexpr (SH) {
        const float *cp = pb.campos + 3 * (size_t)view;
        const float d[3] = {p[0] - cp[0], p[1] - cp[1], p[2] - cp[2]};
        const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        const float u[3] = {d[0] / len, d[1] / len, d[2] / len};
        float B[16], Gb[16][3];
        const int nb = sh_basis(pb.sh_degree, u, B);
        sh_basis_grad(pb.sh_degree, u, Gb);
        const float *sh = pb.shs + sp * (size_t)pb.M * 3;
        float *gsh = dL_dsh + sp * (size_t)pb.M * 3;           // accumulated straight into global memory across the caller's view loop
        const uint8_t cb = clamped[q];
        float gdir[3] = {0.f, 0.f, 0.f};
        for (int ch = 0; ch < 3; ch++) {
            const float gcl = ((cb >> ch) & 1) ? 0.f : gc3[ch];
            for (int k = 0; k < nb; k++) {
                gsh[3 * k + ch] += B[k] * gcl;
                const float sg = sh[3 * k + ch] * gcl;
                gdir[0] += Gb[k][0] * sg; gdir[1] += Gb[k][1] * sg; gdir[2] += Gb[k][2] * sg;
            }
        }
        const float udot = (u[0] * gdir[0] + u[1] * gdir[1]) + u[2] * gdir[2];
#pragma unroll
        for (int k = 0; k < 3; k++) gm[k] += (gdir[k] - u[k] * udot) / len;
    } else {
        out.col[0] = gc3[0]; out.col[1] = gc3[1]; out.col[2] = gc3[2];
    }
    out.op = g1.y;
    out.mean[0] = gm[0]; out.mean[1] = gm[1]; out.mean[2] = gm[2];
    if (g2out) { g2out[0] = g2x; g2out[1] = g2y; g2out[2] = 0.f; }
}

template <bool SH>
__device__ __forceinline__ void bwd_finish(const SgrProblem &pb, size_t sp, const float (&gmean)[3], const float (&gcov)[6], float gop, const float (&gcol)[3],
                                           float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dopacity, float *__restrict__ dL_dcolors,
                                           float *__restrict__ dL_dcov3D, float *__restrict__ dL_dscales, float *__restrict__ dL_drot) {
#pragma unroll
    for (int k = 0; k < 3; k++) dL_dmeans3D[sp * 3 + k] = gmean[k];
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[sp * 6 + k] = gcov[k];
    dL_dopacity[sp] = gop;
    if (!SH) {
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dcolors[sp * 3 + k] = gcol[k];
    }
    if (pb.scales && !((gcov[0] != 0.f) | (gcov[1] != 0.f) | (gcov[2] != 0.f) | (gcov[3] != 0.f) | (gcov[4] != 0.f) | (gcov[5] != 0.f))) {
        // no gradient reached the covariance (culled in every view, or fully occluded): zeros, whatever the scales and the quaternion hold -- the
        // chain below would turn a non-finite scale of a splat nobody saw into NaN gradients (0 * NaN), where upstream's backward returns early
        // for a culled splat and leaves its zero-initialised outputs alone
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dscales[sp * 3 + k] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) dL_drot[sp * 4 + k] = 0.f;
    } else if (pb.scales) {
        float s[3], qv[4], Rm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++) s[k] = pb.scales[sp * 3 + k];
#pragma unroll
        for (int k = 0; k < 4; k++) qv[k] = pb.rotations[sp * 4 + k];
        quat_to_rot(qv, Rm);
        const float r = qv[0], x = qv[1], y = qv[2], z = qv[3];
        const float mod = pb.scale_modifier;
        const float sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
        const float Gs[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                                {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                                {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
        float dMx[3][3], dR[3][3];
#pragma unroll
        for (int a3 = 0; a3 < 3; a3++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 3; j++) acc += Gs[a3][j] * (Rm[j][k] * sv[k]);
                dMx[a3][k] = 2.f * acc;
            }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float acc = 0.f;
#pragma unroll
            for (int a3 = 0; a3 < 3; a3++) { acc += dMx[a3][k] * Rm[a3][k]; dR[a3][k] = dMx[a3][k] * sv[k]; }
            dL_dscales[sp * 3 + k] = mod * acc;
        }
        float *gq = dL_drot + sp * 4;
        gq[0] = 2.f * (z * (dR[1][0] - dR[0][1]) + y * (dR[0][2] - dR[2][0]) + x * (dR[2][1] - dR[1][2]));
        gq[1] = 2.f * (y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + r * (dR[2][1] - dR[1][2])) - 4.f * x * (dR[1][1] + dR[2][2]);
        gq[2] = 2.f * (x * (dR[0][1] + dR[1][0]) + r * (dR[0][2] - dR[2][0]) + z * (dR[1][2] + dR[2][1])) - 4.f * y * (dR[0][0] + dR[2][2]);
        gq[3] = 2.f * (r * (dR[1][0] - dR[0][1]) + x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1])) - 4.f * z * (dR[0][0] + dR[1][1]);
    }
}

#define SGR_BWD_ARGS                                                                                                              \
    SgrProblem pb, const int32_t *__restrict__ radii, const uint8_t *__restrict__ clamped,                                          \
    const uint4 *__restrict__ rect, const float4 *__restrict__ part, const uint32_t *__restrict__ flags, uint32_t n_inst,          \
    const float *__restrict__ part_scale,                                                                                           \
    float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D, float *__restrict__ dL_dopacity,                              \
    float *__restrict__ dL_dcolors, float *__restrict__ dL_dsh, float *__restrict__ dL_dcov3D, float *__restrict__ dL_dscales,      \
    float *__restrict__ dL_drot

// GPW = Gaussians per WAVE: 64, or 16 for launches of few Gaussians (C1: 10 000 splats of up to 7 x 7 tiles are 157 full waves on 1 024 SIMDs,
// each walking ~15 rounds of the wave-wide gather one after the other; with 16 Gaussians on the first 16 lanes of a wave -- all 64 lanes
// still gather -- it is 625 waves of ~4 rounds).  The additions per Gaussian are the same in the same order: bit-identical gradients.
template <bool SH, int GPW = 64>
__global__ __launch_bounds__(kPreThreads) void preprocess_bwd_kernel(SGR_BWD_ARGS) {
    __shared__ __attribute__((aligned(16))) float gather_lds[kPreThreads / 64][gather_lds_floats(128)];
    const int subj = blockIdx.y;
    const int i_raw = GPW == 64 ? blockIdx.x * kPreThreads + threadIdx.x
                                : (blockIdx.x * (kPreThreads / 64) + (threadIdx.x >> 6)) * GPW + (threadIdx.x & 63);
    const bool live = i_raw < pb.P && (GPW == 64 || (int)(threadIdx.x & 63) < GPW);     // (no early exit: the gather of a large splat needs every lane of its wave)
    const int i = live ? i_raw : pb.P - 1;
    const size_t sp = (size_t)subj * pb.P + i;
    const float fx = (float)pb.W / (2.0f * pb.tanfovx), fy = (float)pb.H / (2.0f * pb.tanfovy);
    const float p[3] = {pb.means3D[sp * 3 + 0], pb.means3D[sp * 3 + 1], pb.means3D[sp * 3 + 2]};
    float c6[6];
    load_cov3d(pb, sp, c6);
    float gmean[3] = {0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gop = 0.f, gcol[3] = {0.f, 0.f, 0.f};
    if (SH && live) {
        float *gsh = dL_dsh + sp * (size_t)pb.M * 3;
        for (int k = 0; k < pb.M * 3; k++) gsh[k] = 0.f;
    }
    const int v0 = subj * pb.views_per_subject;
    for (int vv = 0; vv < pb.views_per_subject; vv++) {
        ViewGrad g;
        bwd_view<SH, 128>(pb, v0 + vv, i, sp, p, c6, fx, fy, radii, clamped, rect, part, flags, n_inst, part_scale, dL_dmeans2D, dL_dsh, g, live, gather_lds[threadIdx.x >> 6]);
#pragma unroll
        for (int k = 0; k < 6; k++) gcov[k] += g.cov[k];
        if (!SH) { gcol[0] += g.col[0]; gcol[1] += g.col[1]; gcol[2] += g.col[2]; }
        gop += g.op;
        gmean[0] += g.mean[0]; gmean[1] += g.mean[1]; gmean[2] += g.mean[2];
    }
    if (!live) return;
    bwd_finish<SH>(pb, sp, gmean, gcov, gop, gcol, dL_dmeans3D, dL_dopacity, dL_dcolors, dL_dcov3D, dL_dscales, dL_drot);
}

// colors_precomp path, views_per_subject = VPS in {2, 4, .., 256}: thread t of a workgroup = (view t / GPB, Gaussian t % GPB) with
// GPB = 256 / VPS Gaussians per workgroup (view-major, so the 16-byte rect records of a wave's lanes are contiguous per view)
// (five waves per SIMD: the gather lives on memory latency, 96 VGPRs instead of 98 buy a fifth wave -- 0.368 -> 0.348 ms at C3; six spill: 0.45)
__global__ __launch_bounds__(kPreThreads) __attribute__((amdgpu_waves_per_eu(5))) void preprocess_bwd_lanes_kernel(SGR_BWD_ARGS) {
    __shared__ float acc[13][kPreThreads];
    __shared__ __attribute__((aligned(16))) float gather_lds[kPreThreads / 64][gather_lds_floats(64)];
    const int subj = blockIdx.y, vps = pb.views_per_subject, gpb = kPreThreads / vps;
    const int t = threadIdx.x, vv = t / gpb, gl = t - vv * gpb;
    const int i = blockIdx.x * gpb + gl;
    const float fx = (float)pb.W / (2.0f * pb.tanfovx), fy = (float)pb.H / (2.0f * pb.tanfovy);
    ViewGrad g;
    {
        const bool live = i < pb.P;                                   // (every lane makes the call: the gather of a large splat is done by the whole wave)
        const int ic = live ? i : pb.P - 1;
        const size_t sp = (size_t)subj * pb.P + ic;
        const float p[3] = {pb.means3D[sp * 3 + 0], pb.means3D[sp * 3 + 1], pb.means3D[sp * 3 + 2]};
        float c6[6];
        load_cov3d(pb, sp, c6);
        bwd_view<false, 64>(pb, subj * vps + vv, ic, sp, p, c6, fx, fy, radii, clamped, rect, part, flags, n_inst, part_scale, dL_dmeans2D, dL_dsh, g, live, gather_lds[t >> 6]);
    }
    if (i < pb.P) {
#pragma unroll
        for (int k = 0; k < 3; k++) { acc[k][t] = g.mean[k]; acc[10 + k][t] = g.col[k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) acc[3 + k][t] = g.cov[k];
        acc[9][t] = g.op;
    }
    __syncthreads();
    if (t < gpb && i < pb.P) {                                         // (t < gpb: vv == 0, gl == t)
        float s13[13];
#pragma unroll
        for (int k = 0; k < 13; k++) s13[k] = 0.f;
        for (int w = 0; w < vps; w++)                                  // view order: the additions of preprocess_bwd_kernel's loop
#pragma unroll
            for (int k = 0; k < 13; k++) s13[k] += acc[k][w * gpb + t];
        const float gmean[3] = {s13[0], s13[1], s13[2]}, gcov[6] = {s13[3], s13[4], s13[5], s13[6], s13[7], s13[8]}, gcol[3] = {s13[10], s13[11], s13[12]};
        bwd_finish<false>(pb, (size_t)subj * pb.P + i, gmean, gcov, s13[9], gcol, dL_dmeans3D, dL_dopacity, dL_dcolors, dL_dcov3D, dL_dscales, dL_drot);
    }
}

__global__ __launch_bounds__(kPreThreads) void mark_visible_kernel(int P, const float *__restrict__ means3D,
                                                                   const float *__restrict__ V, uint8_t *__restrict__ present) {
    const int i = blockIdx.x * kPreThreads + threadIdx.x;
    if (i >= P) return;
    const float p[3] = {means3D[3 * (size_t)i], means3D[3 * (size_t)i + 1], means3D[3 * (size_t)i + 2]};
    float pv[3];
    xform4x3(V, p, pv);
    present[i] = pv[2] > 0.2f ? 1 : 0;
}

int validate_problem(const SgrProblem *pb) {
    if (!pb) { sgr_set_error("null SgrProblem"); return 1; }
    if (pb->P < 0 || pb->n_views <= 0 || pb->views_per_subject <= 0 || pb->n_views % pb->views_per_subject != 0) {
        sgr_set_error("bad batch shape: P=%d n_views=%d views_per_subject=%d", pb->P, pb->n_views, pb->views_per_subject);
        return 1;
    }
    if (pb->H <= 0 || pb->W <= 0 || pb->H > 65535 * SGR_TILE || pb->W > 65535 * SGR_TILE) {
        sgr_set_error("bad image size %dx%d", pb->H, pb->W);
        return 1;
    }
    if (pb->P > 0 && (!pb->means3D || !pb->opacities)) { sgr_set_error("means3D / opacities must not be NULL"); return 1; }
    if ((pb->colors_precomp != nullptr) == (pb->shs != nullptr) && pb->P > 0) {
        sgr_set_error("Please provide excatly one of either SHs or precomputed colors!");
        return 1;
    }
    const bool sr = pb->scales != nullptr && pb->rotations != nullptr;
    if (((pb->scales != nullptr) != (pb->rotations != nullptr)) || (sr == (pb->cov3D_precomp != nullptr) && pb->P > 0)) {
        sgr_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return 1;
    }
    if (pb->shs && (pb->sh_degree < 0 || pb->sh_degree > 3 || pb->M < (pb->sh_degree + 1) * (pb->sh_degree + 1))) {
        sgr_set_error("sh_degree %d needs M >= %d coefficients (got %d)", pb->sh_degree, (pb->sh_degree + 1) * (pb->sh_degree + 1), pb->M);
        return 1;
    }
    if (!pb->viewmatrix || !pb->projmatrix || !pb->bg || (pb->shs && !pb->campos)) { sgr_set_error("camera pointers must not be NULL"); return 1; }
    return 0;
}

}  // namespace

int sgr_validate_problem(const SgrProblem *pb) { return validate_problem(pb); }

static thread_local int g_view_group = 0;
extern "C" int sgr_set_preprocess_view_group(int n) { g_view_group = n > 0 ? n : 0; return 0; }

extern "C" int32_t sgr_preprocess_blocks_per_view(int32_t P) { return P <= 0 ? 1 : (P + kPreThreads - 1) / kPreThreads; }

// skip_scan: leave the per-workgroup counts un-scanned behind block_offsets; sgr_bin_ex(self_scan = true) folds F2 into F3
int sgr_preprocess_forward_ex(const SgrProblem *pb, float *rec, int32_t *radii, uint32_t *rect, uint8_t *clamped,
                              uint32_t *block_offsets, uint64_t *num_rendered, uint64_t capacity, bool skip_scan,
                              const SgrBgJob *bg /* optional: background pre-fill by extra workgroups of this launch (fused single-view step) and / or a few words to zero */, void *stream_) {
    if (validate_problem(pb)) return 1;
    if (capacity == 0) capacity = ~0ull;
    hipStream_t stream = (hipStream_t)stream_;
    const int nbx = sgr_preprocess_blocks_per_view(pb->P);
    const uint32_t n = (uint32_t)nbx * (uint32_t)pb->n_views;
    // the un-scanned block sums live in the upper half of a caller buffer? no: scan in place is unsafe with
    // chunked reads, so the sums are staged right behind the offsets (caller allocates 2*(n+1) entries).
    uint32_t *sums = block_offsets + (n + 1);
    // views per workgroup: as many as keep >= 4096 workgroups in flight, at most 8 and at most a subject's views (a group that straddles
    // two subjects reloads its Gaussians at the boundary)
    int vpw = 1;
    while (vpw < 8 && vpw * 2 <= pb->views_per_subject && (size_t)nbx * ((pb->n_views + vpw * 2 - 1) / (vpw * 2)) >= 4096) vpw *= 2;
    if (g_view_group > 0) vpw = g_view_group < pb->n_views ? g_view_group : pb->n_views;
    SgrBgJob bgj;
    memset(&bgj, 0, sizeof(bgj));
    if (bg) bgj = *bg;
    dim3 grid(nbx + (bgj.enabled ? (int)bgj.tiles_total : 0), (pb->n_views + vpw - 1) / vpw);
    { SgrProfScope _p(SGR_K_PREPROCESS_FWD, stream);
    hipLaunchKernelGGL(preprocess_fwd_kernel, grid, dim3(kPreThreads), 0, stream, *pb, vpw, (float4 *)rec, radii, (uint4 *)rect,
                       clamped, sums, nbx, bgj);
    SGR_CHECK_LAUNCH("preprocess_fwd_kernel");
    }
    if (skip_scan) return 0;
    { SgrProfScope _p(SGR_K_SCAN, stream);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(n == 0 ? 1u : (n + kScanTile - 1) / kScanTile), dim3(1024), 0, stream, sums, block_offsets, n, num_rendered,
                       (unsigned long long)capacity);
    SGR_CHECK_LAUNCH("scan_block_sums_kernel");
    }
    return 0;
}

extern "C" int sgr_preprocess_forward(const SgrProblem *pb, float *rec, int32_t *radii, uint32_t *rect, uint8_t *clamped,
                                      uint32_t *block_offsets, uint64_t *num_rendered, uint64_t capacity, void *stream_) {
    return sgr_preprocess_forward_ex(pb, rec, radii, rect, clamped, block_offsets, num_rendered, capacity, false, nullptr, stream_);
}

// 0 = automatic (lanes over views on the colors_precomp path when views_per_subject is a power of two in 2..256), 1 = always the
// one-thread-per-Gaussian kernel (A/B, and the bit-identity test of the kernels)
// (tests: 2 / 3 = the one-thread-per-Gaussian kernel with 64 / 16 Gaussians per wave whatever the launch holds)
static std::atomic<int> g_bwd_view_loop{0};     // read by the backward, i.e. on the autograd thread: process-wide, but an atomic
extern "C" int sgr_set_backward_gather(int mode) { g_bwd_view_loop.store((mode >= 1 && mode <= 3) ? mode : 0); return 0; }

// n_inst: number of tile instances part / flags were sized for (the gather never reads beyond it)
int sgr_preprocess_backward_ex(const SgrProblem *pb, const int32_t *radii, const uint8_t *clamped,
                               const uint32_t *rect, const float *part, const uint32_t *flags, uint64_t n_inst, float *dL_dmeans3D, float *dL_dmeans2D,
                               float *dL_dopacity, float *dL_dcolors, float *dL_dsh, float *dL_dcov3D, float *dL_dscales, float *dL_drotations,
                               const float *part_scale /* optional device scalar on the gathered sums (the fused step's upstream dL/dloss) */, void *stream_) {
    if (validate_problem(pb)) return 1;
    if (pb->P == 0) return 0;
    if (!(part && flags && rect)) { sgr_set_error("sgr_preprocess_backward: need rect + part + flags"); return 1; }
    if (pb->shs && (!dL_dsh || !clamped)) { sgr_set_error("dL_dsh / clamped required on the SH path"); return 1; }
    if (!pb->shs && !dL_dcolors) { sgr_set_error("dL_dcolors required on the colors_precomp path"); return 1; }
    if (pb->scales && (!dL_dscales || !dL_drotations)) { sgr_set_error("dL_dscales / dL_drotations required"); return 1; }
    hipStream_t stream = (hipStream_t)stream_;
    const int nbx = sgr_preprocess_blocks_per_view(pb->P);
    dim3 grid(nbx, pb->n_views / pb->views_per_subject);
    { SgrProfScope _p(SGR_K_PREPROCESS_BWD, stream);
    if (pb->shs)
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, grid, dim3(kPreThreads), 0, stream, *pb, radii, clamped,
                           (const uint4 *)rect, (const float4 *)part, flags, (uint32_t)(n_inst > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_inst), part_scale, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh, dL_dcov3D, dL_dscales, dL_drotations);
    else if (pb->views_per_subject > 1 && pb->views_per_subject <= kPreThreads && kPreThreads % pb->views_per_subject == 0 && !g_bwd_view_loop.load()) {
        const int gpb = kPreThreads / pb->views_per_subject;
        hipLaunchKernelGGL(preprocess_bwd_lanes_kernel, dim3((pb->P + gpb - 1) / gpb, pb->n_views / pb->views_per_subject), dim3(kPreThreads), 0, stream, *pb,
                           radii, clamped, (const uint4 *)rect, (const float4 *)part, flags,
                           (uint32_t)(n_inst > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_inst), part_scale, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh, dL_dcov3D,
                           dL_dscales, dL_drotations);
    } else {
        // few Gaussians: 16 per wave (four times the waves, a quarter of the gather rounds each)
        const int loop_mode = g_bwd_view_loop.load();
        const bool sparse = loop_mode == 3 || (loop_mode != 2 && (int64_t)pb->P * (pb->n_views / pb->views_per_subject) <= 32768);
        if (sparse)
            hipLaunchKernelGGL((preprocess_bwd_kernel<false, 16>), dim3((pb->P + 16 * (kPreThreads / 64) - 1) / (16 * (kPreThreads / 64)), grid.y), dim3(kPreThreads), 0, stream, *pb, radii, clamped,
                               (const uint4 *)rect, (const float4 *)part, flags, (uint32_t)(n_inst > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_inst), part_scale, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh, dL_dcov3D, dL_dscales, dL_drotations);
        else
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, grid, dim3(kPreThreads), 0, stream, *pb, radii, clamped,
                           (const uint4 *)rect, (const float4 *)part, flags, (uint32_t)(n_inst > 0xFFFFFFFFull ? 0xFFFFFFFFull : n_inst), part_scale, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh, dL_dcov3D, dL_dscales, dL_drotations);
    }
    SGR_CHECK_LAUNCH("preprocess_bwd_kernel");
    }
    return 0;
}

extern "C" int sgr_preprocess_backward(const SgrProblem *pb, const int32_t *radii, const uint8_t *clamped,
                                       const uint32_t *rect, const float *part, const uint32_t *flags, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dopacity, float *dL_dcolors,
                                       float *dL_dsh, float *dL_dcov3D, float *dL_dscales, float *dL_drotations,
                                       void *stream_) {
    return sgr_preprocess_backward_ex(pb, radii, clamped, rect, part, flags, ~0ull, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dsh,
                                      dL_dcov3D, dL_dscales, dL_drotations, nullptr, stream_);
}

extern "C" int sgr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, uint8_t *present, void *stream_) {
    if (P <= 0) return 0;
    if (!means3D || !viewmatrix || !present) { sgr_set_error("sgr_mark_visible: NULL pointer"); return 1; }
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + kPreThreads - 1) / kPreThreads), dim3(kPreThreads), 0, (hipStream_t)stream_,
                       P, means3D, viewmatrix, present);
    SGR_CHECK_LAUNCH("mark_visible_kernel");
    return 0;
}
