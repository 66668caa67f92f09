// Dev micro-benchmark: issue rate of plain (non-packed) wave64 VALU instructions on one SIMD, vs packed fp32.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float *out, int iters, float a, float b) {
    float x[16];
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    v2f y[8];
    for (int i = 0; i < 8; i++) y[i] = v2f{x[2 * i], x[2 * i + 1]};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);          // 16 independent v_fma_f32
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = (x[i] > b) ? x[i] - a : x[i] + a;     // compare + selects / adds
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(y[i], v2f{a, a}, v2f{b, b});   // 8 v_pk_fma_f32
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += x[i];
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int MODE> static int run(const char *name, int per_iter) {
    float *out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256 * 8, threads = 256;                   // 8 workgroups x 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = (double)blocks * (threads / 64) * iters * per_iter;
    printf("%-34s %8.3f ms  %.3e wave-instr/s  = %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, winstr / (ms * 1e-3),
           (ms * 1e-3 * 2.4e9) / (winstr / 1024.0));
    return 0;
}
int main() {
    run<0>("v_fma_f32 x16 independent", 16);
    run<1>("cmp + cndmask/add mix", 16 * 3);
    run<2>("v_pk_fma_f32 x8 independent", 8);
    return 0;
}
