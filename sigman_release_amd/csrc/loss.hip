// loss.hip -- fused image-space loss epilogue: the step right after the rasterizer.
// Reference ops being fused:  rendered_image.clamp(0, 1)                  (core/gaussians/gs.py:107)
//                             L1(pred * mask, gt * mask) summed / (B*V)    (core/loss/whole_loss.py:126-131)
// One pass over the image produces per-view loss partial sums (ready for the RCCL all-reduce of image-space
// losses) AND dL/dcolor for the rasterizer backward, instead of ~10 elementwise PyTorch kernels that each
// re-read [V,3,H,W].  HBM-streaming: reads 24 B (+4 B mask) and writes 12 B per pixel, float4-vectorised.
#include "common.h"

namespace {
constexpr int kT = 1024;     // few, fat workgroups: every workgroup ends in two same-address float atomics, which serialise in L2

// grid: (blocks over H*W/4, n_views).  loss_view[v] += sum_{c,p} w * mask * |clamp(color) - target|
// grad[v,c,p] = w * mask * sign(clamp(color) - target) * 1[0 <= color <= 1]    (torch.clamp's backward mask is INCLUSIVE: exactly 0 / 1 pass)
__global__ __launch_bounds__(kT) void clamped_l1_kernel(const float *__restrict__ color, const float *__restrict__ target,
                                                        const float *__restrict__ mask, float weight, int hw, int vec,
                                                        float *__restrict__ grad, float *__restrict__ loss_view, float *__restrict__ loss_total) {
    __shared__ float red[kT / 64];
    const int v = blockIdx.y;
    const size_t base = (size_t)v * 3 * hw;
    float acc = 0.f;
    const int n4 = vec ? (hw >> 2) : 0;
    for (int i = blockIdx.x * kT + threadIdx.x; i < n4; i += gridDim.x * kT) {
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mask) m = reinterpret_cast<const float4 *>(mask + (size_t)v * hw)[i];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float4 x = reinterpret_cast<const float4 *>(color + base + (size_t)c * hw)[i];
            const float4 t = reinterpret_cast<const float4 *>(target + base + (size_t)c * hw)[i];
            float4 g;
#define SGR_L1(X, T, M, G)                                                        \
    {                                                                             \
        const float xc = fminf(fmaxf(X, 0.f), 1.f);                               \
        const float d = (xc - T) * M;                                             \
        acc += fabsf(d);                                                          \
        const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);                   \
        G = (X >= 0.f && X <= 1.f) ? weight * M * s : 0.f;                          \
    }
            SGR_L1(x.x, t.x, m.x, g.x) SGR_L1(x.y, t.y, m.y, g.y) SGR_L1(x.z, t.z, m.z, g.z) SGR_L1(x.w, t.w, m.w, g.w)
#undef SGR_L1
            reinterpret_cast<float4 *>(grad + base + (size_t)c * hw)[i] = g;
        }
    }
    // tail pixels (hw not a multiple of 4): handled by the first block of each view
    if (blockIdx.x == 0) {
        for (int p = (n4 << 2) + threadIdx.x; p < hw; p += kT) {
            const float m = mask ? mask[(size_t)v * hw + p] : 1.f;
            for (int c = 0; c < 3; c++) {
                const float x = color[base + (size_t)c * hw + p], t = target[base + (size_t)c * hw + p];
                const float xc = fminf(fmaxf(x, 0.f), 1.f);
                const float d = (xc - t) * m;
                acc += fabsf(d);
                const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                grad[base + (size_t)c * hw + p] = (x >= 0.f && x <= 1.f) ? weight * m * s : 0.f;
            }
        }
    }
    acc = sgr_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kT / 64; w++) sum += red[w];
        const float part = weight * sum;
        sgr_atomic_add(&loss_view[v], part);
        if (loss_total) sgr_atomic_add(loss_total, part);
    }
}
}  // namespace


extern "C" int sgr_clamped_l1_loss(int32_t n_views, int32_t H, int32_t W, const float *color, const float *target,
                                   const float *mask, float weight, float *grad_color, float *loss_per_view, float *loss_total,
                                   int32_t sums_already_zero, void *stream_) {
    if (n_views <= 0 || H <= 0 || W <= 0) return 0;
    if (!color || !target || !grad_color || !loss_per_view) { sgr_set_error("sgr_clamped_l1_loss: NULL pointer"); return 1; }
    const int hw = H * W;
    const bool vec_ok = !((((uintptr_t)color | (uintptr_t)target | (uintptr_t)grad_color | (uintptr_t)mask) & 15) || (hw & 3));
    hipStream_t stream = (hipStream_t)stream_;
    const bool adjacent = loss_total == loss_per_view + n_views;            // the usual layout: [n_views | total], one memset
    if (!sums_already_zero) {
        SGR_CHECK_HIP(hipMemsetAsync(loss_per_view, 0, sizeof(float) * (n_views + (adjacent ? 1 : 0)), stream));
        if (loss_total && !adjacent) SGR_CHECK_HIP(hipMemsetAsync(loss_total, 0, sizeof(float), stream));
    }
    SgrProfScope _p(SGR_K_LOSS, stream);
    const int n4 = hw >> 2;
    int bx = (n4 + kT - 1) / kT;
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(clamped_l1_kernel, dim3(vec_ok ? bx : 1, n_views), dim3(kT), 0, stream, color, target, mask, weight, hw,
                       vec_ok ? 1 : 0, grad_color, loss_per_view, loss_total);
    SGR_CHECK_LAUNCH("clamped_l1_kernel");
    return 0;
}
