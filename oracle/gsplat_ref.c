/*
 * oracle/gsplat_ref.c -- CPU restatement of the tile-based differentiable 3D-Gaussian-splatting
 * rasterizer that /root/reference calls at core/gaussians/gs.py:82-106 (GaussianRasterizer.forward)
 * and differentiates through at train_vae.py:166.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it.  The product path (sigman_release_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" at the reference level.  The arithmetic of this path lives in
 * the third-party package `diff_gaussian_rasterization` (github.com/ashawkey/diff-gaussian-
 * rasterization, unpinned HEAD, README.md:17-19 of the reference), which is NOT under
 * /root/reference, ships no tests / golden vectors, and is CUDA-only (cannot run here).  This file
 * restates its published algorithm (SURVEY.md Appendix A lists every constant: 0.2 near cull,
 * 1.3*tanfov clamp, +0.3 px^2 dilation, 0.1 eigenvalue floor, ceil(3 sigma) radius, 16x16 tiles,
 * 0.99 alpha cap, 1/255 alpha floor, 1e-4 transmittance stop, 1e-7 epsilons).  It is pinned by
 * (1) oracle/dense_oracle.py -- an independent O(P*H*W) PyTorch-autograd evaluation (fp64) whose
 * gradients come from autograd, not from the hand-derived formulas below -- and (2) golden vectors
 * in tests/golden/ generated from this file after (1) agreed.
 *
 * Canonical floating-point order: every expression below is evaluated in IEEE fp32, left to right,
 * WITHOUT fused multiply-add (compile with -ffp-contract=off).  The HIP preprocess kernel uses the
 * same order with contraction disabled, which is what makes radii / tile rects / depth-key bits /
 * sorted lists / tile ranges bit-exact between the two.  ndc2Pix is evaluated in double exactly
 * like the published algorithm (its literals are doubles).
 *
 * The per-visit exponent and the alpha test (F6 / B1) exist in two arithmetic forms, selected by
 * ref_set_alpha_mode():
 *   0 "reproducible" (DEFAULT, what the HIP kernels are pinned against): the published
 *       power = -0.5 (con.x dx^2 + con.z dy^2) - con.y dx dy;  alpha = min(0.99, con.w * exp(power))
 *     is evaluated as exp(power) = 2^power2 with the conic pre-scaled by -0.5 log2(e) / -log2(e)
 *     (three fp32 products per Gaussian) and power2 as ONE explicit chain of fused multiply-adds
 *     (ref_power2), and 2^power2 is rounded to fp32 from an fp64 polynomial (ref_exp2_cr: every
 *     step one correctly rounded IEEE operation).  Both are reproducible bit for bit on any IEEE
 *     machine -- the HIP kernels evaluate the same chain (v_fma_f32), and preprocess.hip derives from
 *     the same ref_exp2_cr steps the per-Gaussian exponent threshold its alpha test uses -- so
 *     EVERY alpha-test decision (alpha >= 1/255, power > 0) of the HIP path equals this file's.
 *   1 "published order": the literal expression above, left to right without FMA, libm expf.
 *     Mathematically the same function; the exponents differ by a few ulp, which moves an alpha
 *     that sits within ~1e-6 (relative) of 1/255 across the threshold: about 2 pixels per million
 *     (tests/test_oracle_cpu.py counts them).  Upstream's own build is no better defined: nvcc
 *     contracts the expression into FMAs of its choosing and exp() is CUDA's 2-ulp expf.
 *
 * One deliberate deviation from the ashawkey fork, documented in DESIGN.md: the backward pass
 * starts its transmittance walk from a stored final_T (as the original Inria rasterizer does)
 * instead of recomputing 1 - out_alpha; the two differ by fp32 rounding of a sum and the stored
 * value is the more accurate one.
 *
 * Layout conventions (all row-major, fp32 unless noted):
 *   viewmatrix / projmatrix : 16 floats in the memory order of the reference's tensors
 *                             (gs.py:78-79): flat[4*c + r] = M[r][c]  (column-major w2c / full proj)
 *   cov3D                   : [P,6] = xx,xy,xz,yy,yz,zz   (gs.py:29-38 strip_lowerdiag)
 *   out_color [3,H,W], out_depth [H,W], out_alpha [H,W]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16

/* real spherical-harmonics basis constants (standard, as used by every 3DGS implementation) */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct {
    int P, H, W;
    int sh_degree, M;          /* M = number of SH coefficients per Gaussian (0 if unused) */
    float tanfovx, tanfovy;
    float scale_modifier;
    const float *viewmatrix;   /* [16] */
    const float *projmatrix;   /* [16] */
    const float *campos;       /* [3]  */
    const float *bg;           /* [3]  */
} RefCam;

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

/* Threads of the CALLING thread's later parallel regions (OpenMP's nthreads-var is per thread): lets a test harness run several oracle
 * calls side by side from a thread pool, each with a small team, instead of one call at a time on every core. */
void ref_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- per-visit exponent / alpha (see the header): 0 = reproducible (default), 1 = published order */
static int g_alpha_mode = 0;
void ref_set_alpha_mode(int mode) { g_alpha_mode = mode ? 1 : 0; }
int ref_get_alpha_mode(void) { return g_alpha_mode; }

/* 2^x rounded to fp32 from an fp64 evaluation with ~1e-16 relative error: n = rint(x), e^((x - n) ln 2) as a
 * degree-13 Taylor polynomial in Horner form.  One correctly rounded IEEE operation per step; the same steps in
 * the same order as sgr_exp2_cr in sigman_release_amd/csrc/preprocess.hip (restated there, not shared). */
static inline float ref_exp2_cr(float x) {
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
void ref_exp2_cr_array(const float *x, float *y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = ref_exp2_cr(x[i]); }

/* The alpha test as a threshold on the exponent (what the HIP kernels evaluate): alpha = min(0.99, op * 2^p) is monotone in p, so
 *     alpha >= 1/255   <=>   p >= p*(op),      p* = the smallest fp32 p <= 0 that passes (+inf: none does).
 * Restated here so that tests can (a) prove the equivalence on the CPU (tests/test_oracle_cpu.py) and (b) compare the p* the HIP
 * preprocess kernel stores in every record bit for bit.  Plain bisection over the bit patterns of |p| in [0, 127]. */
static inline int ref_alpha_passes(float op, uint32_t mag) {
    union { uint32_t u; float f; } c; c.u = mag;
    return fminf_(0.99f, op * ref_exp2_cr(-c.f)) >= 1.0f / 255.0f;
}
float ref_alpha_threshold(float op) {
    if (!(op >= 1.0f / 255.0f)) return INFINITY;
    uint32_t lo = 0u, hi = 0x42FE0000u;                     /* |p| = 0 passes, |p| = 127 does not */
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (ref_alpha_passes(op, mid)) lo = mid; else hi = mid;
    }
    union { uint32_t u; float f; } c; c.u = lo;
    return -c.f;
}
void ref_alpha_threshold_array(const float *op, float *pstar, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) pstar[i] = ref_alpha_threshold(op[i]);
}
/* min(0.99, op * 2^p) >= 1/255 with p given by its value (p <= 0): the direct evaluation the renderer below uses */
void ref_alpha_test_array(const float *op, const float *p, uint8_t *pass, int64_t n) {
    for (int64_t i = 0; i < n; i++) pass[i] = (p[i] <= 0.0f) && (fminf_(0.99f, op[i] * ref_exp2_cr(p[i])) >= 1.0f / 255.0f);
}

/* exponent in the exp2 domain: conic pre-scaled (kxx, kyy by -0.5 log2 e, kxy by -log2 e), one FMA chain */
static const float K_LOG2E = -1.4426950408889634f, K_HALF_LOG2E = -0.7213475204444817f;
static inline float ref_power2(const float *co, float dx, float dy) {
    const float kxx = K_HALF_LOG2E * co[0], kxy = K_LOG2E * co[1], kyy = K_HALF_LOG2E * co[2];
    return fmaf(dx, kxx * dx, fmaf(kxy * dx, dy, (kyy * dy) * dy));
}
/* G = exp(power) at one (pixel, Gaussian) pair, or -1 if the published `power > 0 -> continue` rule applies */
static inline float ref_gauss(const float *co, float dx, float dy) {
    if (g_alpha_mode == 0) {
        const float p2 = ref_power2(co, dx, dy);
        if (p2 > 0.0f) return -1.f;
        return ref_exp2_cr(p2);
    }
    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
    if (power > 0.0f) return -1.f;
    return expf(power);
}

/* p_view = w2c * p  (published transformPoint4x3) */
static inline void xform4x3(const float *m, const float *p, float *o) {
    o[0] = ((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2]) + m[12];
    o[1] = ((m[1] * p[0] + m[5] * p[1]) + m[9] * p[2]) + m[13];
    o[2] = ((m[2] * p[0] + m[6] * p[1]) + m[10] * p[2]) + m[14];
}
static inline void xform4x4(const float *m, const float *p, float *o) {
    o[0] = ((m[0] * p[0] + m[4] * p[1]) + m[8] * p[2]) + m[12];
    o[1] = ((m[1] * p[0] + m[5] * p[1]) + m[9] * p[2]) + m[13];
    o[2] = ((m[2] * p[0] + m[6] * p[1]) + m[10] * p[2]) + m[14];
    o[3] = ((m[3] * p[0] + m[7] * p[1]) + m[11] * p[2]) + m[15];
}

/* ndc2Pix of the published algorithm: evaluated in double, rounded once to float */
static inline float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* Sigma = R diag(mod*s)^2 R^T from an UNNORMALISED quaternion (r,x,y,z); packs xx,xy,xz,yy,yz,zz */
static void cov3d_from_scale_rot(const float *s, float mod, const float *q, float *cov) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                     {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                     {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    float sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float Mx[3][3]; /* Mx = R * diag(sv) */
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Mx[i][k] = R[i][k] * sv[k];
    float S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[i][j] = (Mx[i][0] * Mx[j][0] + Mx[i][1] * Mx[j][1]) + Mx[i][2] * Mx[j][2];
    cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2];
    cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
}

/* SH basis values for a unit direction d; returns number of coefficients for `deg` */
static int sh_basis(int deg, const float *d, float *B) {
    float x = d[0], y = d[1], z = d[2];
    B[0] = SH_C0;
    if (deg < 1) return 1;
    B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
    if (deg < 2) return 4;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = SH_C2[0] * xy; B[5] = SH_C2[1] * yz; B[6] = SH_C2[2] * (2.f * zz - xx - yy);
    B[7] = SH_C2[3] * xz; B[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return 9;
    B[9] = SH_C3[0] * y * (3.f * xx - yy); B[10] = SH_C3[1] * xy * z;
    B[11] = SH_C3[2] * y * (4.f * zz - xx - yy); B[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
    B[13] = SH_C3[4] * x * (4.f * zz - xx - yy); B[14] = SH_C3[5] * z * (xx - yy);
    B[15] = SH_C3[6] * x * (xx - 3.f * yy);
    return 16;
}
/* d(basis_k)/d(dir) for the same basis */
static void sh_basis_grad(int deg, const float *d, float G[16][3]) {
    float x = d[0], y = d[1], z = d[2];
    memset(G, 0, sizeof(float) * 48);
    if (deg < 1) return;
    G[1][1] = -SH_C1; G[2][2] = SH_C1; G[3][0] = -SH_C1;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z;
    G[4][0] = SH_C2[0] * y; G[4][1] = SH_C2[0] * x;
    G[5][1] = SH_C2[1] * z; G[5][2] = SH_C2[1] * y;
    G[6][0] = SH_C2[2] * (-2.f * x); G[6][1] = SH_C2[2] * (-2.f * y); G[6][2] = SH_C2[2] * (4.f * z);
    G[7][0] = SH_C2[3] * z; G[7][2] = SH_C2[3] * x;
    G[8][0] = SH_C2[4] * (2.f * x); G[8][1] = SH_C2[4] * (-2.f * y);
    if (deg < 3) return;
    G[9][0] = SH_C3[0] * (6.f * x * y); G[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
    G[10][0] = SH_C3[1] * y * z; G[10][1] = SH_C3[1] * x * z; G[10][2] = SH_C3[1] * x * y;
    G[11][0] = SH_C3[2] * (-2.f * x * y); G[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); G[11][2] = SH_C3[2] * (8.f * y * z);
    G[12][0] = SH_C3[3] * (-6.f * x * z); G[12][1] = SH_C3[3] * (-6.f * y * z); G[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
    G[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); G[13][1] = SH_C3[4] * (-2.f * x * y); G[13][2] = SH_C3[4] * (8.f * x * z);
    G[14][0] = SH_C3[5] * (2.f * x * z); G[14][1] = SH_C3[5] * (-2.f * y * z); G[14][2] = SH_C3[5] * (xx - yy);
    G[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); G[15][1] = SH_C3[6] * (-6.f * x * y);
}

/* shared by preprocess forward and backward: the 2x3 projection Jacobian rows m0,m1 and cov2D */
typedef struct {
    float t[3];            /* view-space mean, x/y clamped to the 1.3*tanfov cone */
    float xmul, ymul;      /* 0 where the clamp was active (published x_grad_mul / y_grad_mul) */
    float j00, j02, j11, j12;
    float m0[3], m1[3];    /* rows of J*Rw2c */
    float v0[3], v1[3];    /* Sigma*m0, Sigma*m1 */
    float a, b, c;         /* dilated 2D covariance */
} Cov2D;

static void cov2d_eval(const float *pview, const float *V, const float *cov6, float fx, float fy,
                       float tanfovx, float tanfovy, Cov2D *o) {
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float tz = pview[2];
    const float txtz = pview[0] / tz, tytz = pview[1] / tz;
    o->xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    o->ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float tx = fminf_(limx, fmaxf_(-limx, txtz)) * tz;
    const float ty = fminf_(limy, fmaxf_(-limy, tytz)) * tz;
    o->t[0] = tx; o->t[1] = ty; o->t[2] = tz;
    o->j00 = fx / tz; o->j02 = -(fx * tx) / (tz * tz);
    o->j11 = fy / tz; o->j12 = -(fy * ty) / (tz * tz);
    /* rows of the w2c rotation: Rw[r][k] = V[4k + r] */
    for (int k = 0; k < 3; k++) {
        float r0 = V[4 * k + 0], r1 = V[4 * k + 1], r2 = V[4 * k + 2];
        o->m0[k] = o->j00 * r0 + o->j02 * r2;
        o->m1[k] = o->j11 * r1 + o->j12 * r2;
    }
    const float S[3][3] = {{cov6[0], cov6[1], cov6[2]}, {cov6[1], cov6[3], cov6[4]}, {cov6[2], cov6[4], cov6[5]}};
    for (int i = 0; i < 3; i++) {
        o->v0[i] = (S[i][0] * o->m0[0] + S[i][1] * o->m0[1]) + S[i][2] * o->m0[2];
        o->v1[i] = (S[i][0] * o->m1[0] + S[i][1] * o->m1[1]) + S[i][2] * o->m1[2];
    }
    o->a = ((o->m0[0] * o->v0[0] + o->m0[1] * o->v0[1]) + o->m0[2] * o->v0[2]) + 0.3f;
    o->b = (o->m0[0] * o->v1[0] + o->m0[1] * o->v1[1]) + o->m0[2] * o->v1[2];
    o->c = ((o->m1[0] * o->v1[0] + o->m1[1] * o->v1[1]) + o->m1[2] * o->v1[2]) + 0.3f;
}

/* ------------------------------------------------------------------------------------------
 * F1  preprocess: cull, project, 2D covariance, conic, radius, tile rectangle.
 * Follows the published preprocessCUDA; called once per view at gs.py:98-106.
 * Outputs (all [P...]): depths, xy[2], conic_opacity[4], radii (i32), rect[4] (minx,miny,maxx,maxy,
 * i32), tiles_touched (u32), cov3D[6] (copy of precomp or computed), rgb[3], clamped[3] (u8).
 * ---------------------------------------------------------------------------------------- */
void ref_preprocess(const RefCam *cam, const float *means3D, const float *opacities,
                    const float *cov3D_precomp, const float *scales, const float *rotations,
                    const float *colors_precomp, const float *shs, float *depths, float *xy,
                    float *conic_opacity, int32_t *radii, int32_t *rect, uint32_t *tiles_touched,
                    float *cov3D, float *rgb, uint8_t *clamped) {
    const int P = cam->P, W = cam->W, H = cam->H;
    const int Tx = (W + TILE - 1) / TILE, Ty = (H + TILE - 1) / TILE;
    const float fx = (float)W / (2.0f * cam->tanfovx), fy = (float)H / (2.0f * cam->tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0;
        depths[i] = 0.f; xy[2 * i] = xy[2 * i + 1] = 0.f;
        for (int k = 0; k < 4; k++) { conic_opacity[4 * i + k] = 0.f; rect[4 * i + k] = 0; }
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = 0.f;
        const float *p = means3D + 3 * i;
        float pv[3];
        xform4x3(cam->viewmatrix, p, pv);
        if (pv[2] <= 0.2f) continue;                      /* near cull; lateral test is disabled upstream */
        float ph[4];
        xform4x4(cam->projmatrix, p, ph);
        const float pw = 1.0f / (ph[3] + 0.0000001f);
        const float projx = ph[0] * pw, projy = ph[1] * pw;
        float *c6 = cov3D + 6 * i;
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, 6 * sizeof(float)); /* scale_modifier NOT applied */
        else cov3d_from_scale_rot(scales + 3 * i, cam->scale_modifier, rotations + 4 * i, c6);
        Cov2D q;
        cov2d_eval(pv, cam->viewmatrix, c6, fx, fy, cam->tanfovx, cam->tanfovy, &q);
        const float det = q.a * q.c - q.b * q.b;
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float cx = q.c * det_inv, cy = -q.b * det_inv, cz = q.a * det_inv;
        const float mid = 0.5f * (q.a + q.c);
        const float disc = sqrtf(fmaxf_(0.1f, mid * mid - det));
        const float lam1 = mid + disc, lam2 = mid - disc;
        const float my_radius = ceilf(3.f * sqrtf(fmaxf_(lam1, lam2)));
        const float px = ndc2pix(projx, W), py = ndc2pix(projy, H);
        const int rad = (int)my_radius;
        int minx = (int)((px - (float)rad) / (float)TILE); minx = minx < 0 ? 0 : (minx > Tx ? Tx : minx);
        int miny = (int)((py - (float)rad) / (float)TILE); miny = miny < 0 ? 0 : (miny > Ty ? Ty : miny);
        int maxx = (int)((px + (float)rad + (float)(TILE - 1)) / (float)TILE); maxx = maxx < 0 ? 0 : (maxx > Tx ? Tx : maxx);
        int maxy = (int)((py + (float)rad + (float)(TILE - 1)) / (float)TILE); maxy = maxy < 0 ? 0 : (maxy > Ty ? Ty : maxy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;
        if (!(rad > 0)) continue;   /* a NaN radius ((int)NaN is not even defined in C): never emitted (radii > 0 below), so it must not count tiles either -- the HIP path's rule */
        if (colors_precomp) {
            for (int k = 0; k < 3; k++) rgb[3 * i + k] = colors_precomp[3 * i + k]; /* untouched, no clamp */
        } else {
            float d[3] = {p[0] - cam->campos[0], p[1] - cam->campos[1], p[2] - cam->campos[2]};
            float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            d[0] /= len; d[1] /= len; d[2] /= len;
            float B[16];
            int nb = sh_basis(cam->sh_degree, d, B);
            const float *sh = shs + (size_t)i * cam->M * 3;
            for (int ch = 0; ch < 3; ch++) {
                float r = 0.f;
                for (int k = 0; k < nb; k++) r += B[k] * sh[3 * k + ch];
                r += 0.5f;
                clamped[3 * i + ch] = r < 0.f;
                rgb[3 * i + ch] = fmaxf_(r, 0.f);
            }
        }
        depths[i] = pv[2];
        radii[i] = rad;
        xy[2 * i] = px; xy[2 * i + 1] = py;
        conic_opacity[4 * i + 0] = cx; conic_opacity[4 * i + 1] = cy;
        conic_opacity[4 * i + 2] = cz; conic_opacity[4 * i + 3] = opacities[i];
        rect[4 * i + 0] = minx; rect[4 * i + 1] = miny; rect[4 * i + 2] = maxx; rect[4 * i + 3] = maxy;
        tiles_touched[i] = (uint32_t)((maxx - minx) * (maxy - miny));
    }
}

/* published markVisible: only the near-plane test survives upstream */
void ref_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present) {
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform4x3(viewmatrix, means3D + 3 * i, pv);
        present[i] = pv[2] > 0.2f;
    }
}

/* ------------------------------------------------------------------------------------------
 * F2-F5 binning: inclusive scan of tiles_touched, key emission, stable sort, per-tile ranges.
 * key = (tile_id << 32) | bitcast<u32>(depth);  value = Gaussian index.
 * Returns R (number of tile instances).  keys/vals must hold at least sum(tiles_touched).
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t k; uint32_t v; } KV;
static int kv_cmp(const void *a, const void *b) {
    const KV *x = (const KV *)a, *y = (const KV *)b;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    return x->v < y->v ? -1 : (x->v > y->v ? 1 : 0);   /* == stability of the published radix sort */
}
int64_t ref_bin(int P, int H, int W, const int32_t *radii, const int32_t *rect, const float *depths,
                const uint32_t *tiles_touched, uint32_t *point_offsets, uint64_t *keys, uint32_t *vals,
                uint32_t *ranges /* [tiles][2] */) {
    const int Tx = (W + TILE - 1) / TILE, Ty = (H + TILE - 1) / TILE;
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += tiles_touched[i]; point_offsets[i] = run; }
    const int64_t R = run;
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)Tx * Ty);
    if (R == 0) return 0;
    KV *kv = (KV *)malloc(sizeof(KV) * (size_t)R);
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        uint32_t off = i == 0 ? 0 : point_offsets[i - 1];
        uint32_t dbits; memcpy(&dbits, depths + i, 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; y++)
            for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; x++) {
                kv[off].k = ((uint64_t)(uint32_t)(y * Tx + x) << 32) | dbits;
                kv[off].v = (uint32_t)i;
                off++;
            }
    }
    qsort(kv, (size_t)R, sizeof(KV), kv_cmp);
    for (int64_t r = 0; r < R; r++) { keys[r] = kv[r].k; vals[r] = kv[r].v; }
    free(kv);
    for (int64_t r = 0; r < R; r++) {
        uint32_t t = (uint32_t)(keys[r] >> 32);
        if (r == 0 || t != (uint32_t)(keys[r - 1] >> 32)) ranges[2 * t] = (uint32_t)r;
        if (r == R - 1 || t != (uint32_t)(keys[r + 1] >> 32)) ranges[2 * t + 1] = (uint32_t)(r + 1);
    }
    return R;
}

/* ------------------------------------------------------------------------------------------
 * F6 render forward: per pixel front-to-back compositing over the tile's sorted list.
 * ---------------------------------------------------------------------------------------- */
void ref_render_fwd(int H, int W, const uint32_t *ranges, const uint32_t *point_list, const float *xy,
                    const float *conic_opacity, const float *rgb, const float *depths, const float *bg,
                    float *out_color, float *out_depth, float *out_alpha, float *final_T,
                    uint32_t *n_contrib) {
    const int Tx = (W + TILE - 1) / TILE, Ty = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < Tx * Ty; tile++) {
        const int tx0 = (tile % Tx) * TILE, ty0 = (tile / Tx) * TILE;
        const uint32_t lo = ranges[2 * tile], hi = ranges[2 * tile + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                const float pfx = (float)px, pfy = (float)py;
                float T = 1.f, C[3] = {0, 0, 0}, D = 0.f, A = 0.f;
                uint32_t contributor = 0, last = 0;
                for (uint32_t r = lo; r < hi; r++) {
                    const uint32_t g = point_list[r];
                    contributor++;
                    const float dx = xy[2 * g] - pfx, dy = xy[2 * g + 1] - pfy;
                    const float *co = conic_opacity + 4 * g;
                    const float G = ref_gauss(co, dx, dy);
                    if (G < 0.0f) continue;                 /* power > 0 */
                    const float alpha = fminf_(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1.f - alpha);
                    if (test_T < 0.0001f) break;            /* the crossing Gaussian is NOT composited */
                    const float w = alpha * T;
                    for (int ch = 0; ch < 3; ch++) C[ch] += rgb[3 * g + ch] * w;
                    D += depths[g] * w;
                    A += w;
                    T = test_T;
                    last = contributor;
                }
                const size_t pix = (size_t)py * W + px;
                final_T[pix] = T;
                n_contrib[pix] = last;
                for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
                out_depth[pix] = D;
                out_alpha[pix] = A;
            }
    }
}

/* ------------------------------------------------------------------------------------------
 * B1 render backward: per pixel back-to-front walk.  Per-visit math is fp32 in the published
 * order; the across-pixel sums (fp32 atomics of unspecified order upstream) are accumulated here
 * in double per tile instance and then per Gaussian in sorted-list order, so the oracle itself is
 * deterministic.  Outputs are ZEROED here: dL_dmean2D [P,2] (NDC-scaled like upstream: includes
 * the 0.5*W / 0.5*H factors), dL_dconic [P,3] (xx,xy,yy), dL_dopacity [P], dL_dcolor [P,3],
 * dL_ddepth [P].
 * ---------------------------------------------------------------------------------------- */
void ref_render_bwd(int P, int H, int W, int64_t R, const uint32_t *ranges, const uint32_t *point_list,
                    const float *xy, const float *conic_opacity, const float *rgb, const float *depths,
                    const float *bg, const float *final_T, const uint32_t *n_contrib,
                    const float *gC /*[3,H,W]*/, const float *gD /*[H,W]*/, const float *gA /*[H,W]*/,
                    float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                    float *dL_ddepth) {
    const int Tx = (W + TILE - 1) / TILE, Ty = (H + TILE - 1) / TILE;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    double *inst = (double *)calloc((size_t)(R > 0 ? R : 1) * 10, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < Tx * Ty; tile++) {
        const int tx0 = (tile % Tx) * TILE, ty0 = (tile / Tx) * TILE;
        const uint32_t lo = ranges[2 * tile];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                const size_t pix = (size_t)py * W + px;
                const float pfx = (float)px, pfy = (float)py;
                const float Tf = final_T[pix];
                float T = Tf;
                const float g0 = gC[pix], g1 = gC[(size_t)H * W + pix], g2 = gC[2 * (size_t)H * W + pix];
                const float gd = gD ? gD[pix] : 0.f, ga = gA ? gA[pix] : 0.f;
                const float bg_dot = (bg[0] * g0 + bg[1] * g1) + bg[2] * g2;
                float accC[3] = {0, 0, 0}, accD = 0.f, accA = 0.f;
                float last_alpha = 0.f, lastC[3] = {0, 0, 0}, lastD = 0.f;
                for (int64_t k = (int64_t)n_contrib[pix] - 1; k >= 0; k--) {
                    const uint32_t r = lo + (uint32_t)k;
                    const uint32_t g = point_list[r];
                    const float dx = xy[2 * g] - pfx, dy = xy[2 * g + 1] - pfy;
                    const float *co = conic_opacity + 4 * g;
                    const float G = ref_gauss(co, dx, dy);
                    if (G < 0.0f) continue;                 /* power > 0 */
                    const float alpha = fminf_(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float w = alpha * T;
                    double *acc = inst + (size_t)r * 10;
                    float dL_dalpha = 0.f;
                    const float gch[3] = {g0, g1, g2};
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = rgb[3 * g + ch];
                        accC[ch] = last_alpha * lastC[ch] + (1.f - last_alpha) * accC[ch];
                        lastC[ch] = c;
                        dL_dalpha += (c - accC[ch]) * gch[ch];
                        acc[6 + ch] += (double)(w * gch[ch]);
                    }
                    const float cd = depths[g];
                    accD = last_alpha * lastD + (1.f - last_alpha) * accD;
                    lastD = cd;
                    dL_dalpha += (cd - accD) * gd;
                    acc[9] += (double)(w * gd);
                    accA = last_alpha * 1.0f + (1.f - last_alpha) * accA;
                    dL_dalpha += (1.f - accA) * ga;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-Tf / (1.f - alpha)) * bg_dot;
                    /* alpha was possibly capped at 0.99: upstream still differentiates through op*G */
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    acc[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    acc[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    acc[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    acc[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    acc[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    acc[5] += (double)(G * dL_dalpha);
                }
            }
    }
    double *tot = (double *)calloc((size_t)(P > 0 ? P : 1) * 10, sizeof(double));
    for (int64_t r = 0; r < R; r++) {
        const uint32_t g = point_list[r];
        for (int k = 0; k < 10; k++) tot[(size_t)g * 10 + k] += inst[(size_t)r * 10 + k];
    }
    for (int i = 0; i < P; i++) {
        const double *t = tot + (size_t)i * 10;
        dL_dmean2D[2 * i] = (float)t[0]; dL_dmean2D[2 * i + 1] = (float)t[1];
        dL_dconic[3 * i] = (float)t[2]; dL_dconic[3 * i + 1] = (float)t[3]; dL_dconic[3 * i + 2] = (float)t[4];
        dL_dopacity[i] = (float)t[5];
        dL_dcolor[3 * i] = (float)t[6]; dL_dcolor[3 * i + 1] = (float)t[7]; dL_dcolor[3 * i + 2] = (float)t[8];
        dL_ddepth[i] = (float)t[9];
    }
    free(tot);
    free(inst);
}

/* ------------------------------------------------------------------------------------------
 * B2 + B3 preprocess backward (published computeCov2DCUDA + preprocessCUDA backward):
 *   dL_dconic -> dL_dcov3D, dL_dmeans3D (via the projection Jacobian)
 *   dL_dmean2D -> dL_dmeans3D (via the full projection)
 *   dL_ddepth  -> dL_dmeans3D (z row of w2c)
 *   dL_dcolor  -> dL_dsh (+ dL_dmeans3D via view direction) when SHs are used
 *   dL_dcov3D  -> dL_dscales, dL_drotations when scales/rotations are used
 * Only Gaussians with radii > 0 receive gradient.  All outputs are ZEROED here.
 * ---------------------------------------------------------------------------------------- */
void ref_preprocess_bwd(const RefCam *cam, const float *means3D, const int32_t *radii,
                        const float *cov3D /* as used in forward [P,6] */, const float *scales,
                        const float *rotations, const float *shs, const uint8_t *clamped,
                        const float *dL_dmean2D, const float *dL_dconic, const float *dL_dcolor,
                        const float *dL_ddepth, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
                        float *dL_dscales, float *dL_drot) {
    const int P = cam->P, W = cam->W, H = cam->H;
    const float fx = (float)W / (2.0f * cam->tanfovx), fy = (float)H / (2.0f * cam->tanfovy);
    const float *V = cam->viewmatrix, *M = cam->projmatrix;
    memset(dL_dmeans3D, 0, sizeof(float) * 3 * (size_t)P);
    memset(dL_dcov3D, 0, sizeof(float) * 6 * (size_t)P);
    if (dL_dsh) memset(dL_dsh, 0, sizeof(float) * 3 * (size_t)cam->M * (size_t)P);
    if (dL_dscales) memset(dL_dscales, 0, sizeof(float) * 3 * (size_t)P);
    if (dL_drot) memset(dL_drot, 0, sizeof(float) * 4 * (size_t)P);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        const float *p = means3D + 3 * i;
        float pv[3];
        xform4x3(V, p, pv);
        Cov2D q;
        cov2d_eval(pv, V, cov3D + 6 * i, fx, fy, cam->tanfovx, cam->tanfovy, &q);
        /* ---- B2: conic -> cov2D (a,b,c) ---- */
        const float a = q.a, b = q.b, c = q.c;
        const float gx = dL_dconic[3 * i], gy = dL_dconic[3 * i + 1], gz = dL_dconic[3 * i + 2];
        const float denom = a * c - b * b;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        float *gcov = dL_dcov3D + 6 * i;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * gx + 2.f * b * c * gy + (denom - a * c) * gz);
            dL_dc = denom2inv * (-a * a * gz + 2.f * a * b * gy + (denom - a * c) * gx);
            dL_db = denom2inv * 2.f * (b * c * gx - (denom + 2.f * b * b) * gy + a * b * gz);
            /* a = m0^T S m0, b = m0^T S m1, c = m1^T S m1; packed off-diagonals count twice */
            const float *m0 = q.m0, *m1 = q.m1;
            gcov[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
            gcov[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
            gcov[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
            gcov[1] = 2.f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.f * m1[0] * m1[1] * dL_dc;
            gcov[2] = 2.f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.f * m1[0] * m1[2] * dL_dc;
            gcov[4] = 2.f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.f * m1[1] * m1[2] * dL_dc;
        }
        /* dL/dm0 = 2 dL_da S m0 + dL_db S m1 ; dL/dm1 = 2 dL_dc S m1 + dL_db S m0 */
        float gm0[3], gm1[3];
        for (int k = 0; k < 3; k++) {
            gm0[k] = 2.f * q.v0[k] * dL_da + q.v1[k] * dL_db;
            gm1[k] = 2.f * q.v1[k] * dL_dc + q.v0[k] * dL_db;
        }
        /* m0 = j00*Rw[0] + j02*Rw[2] ; m1 = j11*Rw[1] + j12*Rw[2] */
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
        for (int k = 0; k < 3; k++) {
            dJ00 += V[4 * k + 0] * gm0[k]; dJ02 += V[4 * k + 2] * gm0[k];
            dJ11 += V[4 * k + 1] * gm1[k]; dJ12 += V[4 * k + 2] * gm1[k];
        }
        const float tz = 1.f / q.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = q.xmul * -fx * tz2 * dJ02;
        const float dty = q.ymul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * q.t[0]) * tz3 * dJ02 + (2.f * fy * q.t[1]) * tz3 * dJ12;
        float gm[3]; /* = Rw^T (dtx,dty,dtz) */
        for (int k = 0; k < 3; k++) gm[k] = (V[4 * k + 0] * dtx + V[4 * k + 1] * dty) + V[4 * k + 2] * dtz;
        /* ---- B3: screen-space mean -> 3D mean through the full projection ---- */
        float ph[4];
        xform4x4(M, p, ph);
        const float mw = 1.0f / (ph[3] + 0.0000001f);
        const float mul1 = ph[0] * mw * mw, mul2 = ph[1] * mw * mw;
        const float g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
        gm[0] += (M[0] * mw - M[3] * mul1) * g2x + (M[1] * mw - M[3] * mul2) * g2y;
        gm[1] += (M[4] * mw - M[7] * mul1) * g2x + (M[5] * mw - M[7] * mul2) * g2y;
        gm[2] += (M[8] * mw - M[11] * mul1) * g2x + (M[9] * mw - M[11] * mul2) * g2y;
        /* depth = p_view.z = V[2]x + V[6]y + V[10]z + V[14] */
        const float gdep = dL_ddepth ? dL_ddepth[i] : 0.f;
        gm[0] += V[2] * gdep; gm[1] += V[6] * gdep; gm[2] += V[10] * gdep;
        /* ---- SH colour backward ---- */
        if (shs) {
            float d[3] = {p[0] - cam->campos[0], p[1] - cam->campos[1], p[2] - cam->campos[2]};
            const float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            float u[3] = {d[0] / len, d[1] / len, d[2] / len};
            float B[16], Gb[16][3];
            const int nb = sh_basis(cam->sh_degree, u, B);
            sh_basis_grad(cam->sh_degree, u, Gb);
            const float *sh = shs + (size_t)i * cam->M * 3;
            float *gsh = dL_dsh + (size_t)i * cam->M * 3;
            float gdir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
                const float gcol = clamped[3 * i + ch] ? 0.f : dL_dcolor[3 * i + ch];
                for (int k = 0; k < nb; k++) {
                    gsh[3 * k + ch] = B[k] * gcol;
                    for (int a3 = 0; a3 < 3; a3++) gdir[a3] += Gb[k][a3] * sh[3 * k + ch] * gcol;
                }
            }
            /* through u = d/|d| : (I - u u^T)/|d| */
            const float udot = (u[0] * gdir[0] + u[1] * gdir[1]) + u[2] * gdir[2];
            for (int a3 = 0; a3 < 3; a3++) gm[a3] += (gdir[a3] - u[a3] * udot) / len;
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = gm[k];
        /* ---- Sigma = R diag(s)^2 R^T backward ---- */
        if (scales) {
            const float *qv = rotations + 4 * i, *s = scales + 3 * i;
            const float r = qv[0], x = qv[1], y = qv[2], z = qv[3];
            const float Rm[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                                    {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                                    {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
            const float mod = cam->scale_modifier;
            const float sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
            /* full symmetric dL/dSigma (packed off-diagonal grads are for the pair, so halve them) */
            const float Gs[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                                    {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                                    {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
            /* Sigma = Mx Mx^T, Mx = R diag(sv):  dL/dMx = 2 Gs Mx */
            float dMx[3][3];
            for (int a3 = 0; a3 < 3; a3++)
                for (int k = 0; k < 3; k++) {
                    float acc = 0.f;
                    for (int j = 0; j < 3; j++) acc += Gs[a3][j] * (Rm[j][k] * sv[k]);
                    dMx[a3][k] = 2.f * acc;
                }
            float dR[3][3];
            for (int k = 0; k < 3; k++) {
                float acc = 0.f;
                for (int a3 = 0; a3 < 3; a3++) { acc += dMx[a3][k] * Rm[a3][k]; dR[a3][k] = dMx[a3][k] * sv[k]; }
                dL_dscales[3 * i + k] = mod * acc;
            }
            float *gq = dL_drot + 4 * i;
            gq[0] = 2.f * (z * (dR[1][0] - dR[0][1]) + y * (dR[0][2] - dR[2][0]) + x * (dR[2][1] - dR[1][2]));
            gq[1] = 2.f * (y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + r * (dR[2][1] - dR[1][2])) - 4.f * x * (dR[1][1] + dR[2][2]);
            gq[2] = 2.f * (x * (dR[0][1] + dR[1][0]) + r * (dR[0][2] - dR[2][0]) + z * (dR[1][2] + dR[2][1])) - 4.f * y * (dR[0][0] + dR[2][2]);
            gq[3] = 2.f * (r * (dR[1][0] - dR[0][1]) + x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1])) - 4.f * z * (dR[0][0] + dR[1][1]);
        }
    }
}

int ref_abi_version(void) { return 1; }
