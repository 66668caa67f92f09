// Dev micro-benchmark: what do N workgroups pay for ending in ONE float atomicAdd on the SAME address (the loss sum folded into a
// compositing kernel), against ending in a plain store?  Each workgroup does ~10 us of dependent arithmetic first, so the atomics arrive
// spread over the kernel like they would at the end of tile workgroups.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_same.hip -o tools/micro/atomic_same
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE> __global__ void k(float *sum, float *plain, int iters, float a, int naddr) {
    float x = threadIdx.x * 0.001f + blockIdx.x;
    for (int it = 0; it < iters; it++) x = __builtin_fmaf(x, a, 0.5f);
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    if ((threadIdx.x & 63) == 0) {
        if (MODE == 0) plain[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = x;
        else if (MODE == 1) atomicAdd(&sum[(blockIdx.x % naddr) * 64], x);                       // non-returning, device scope
        else { const float old = atomicAdd(&sum[(blockIdx.x % naddr) * 64], x); if (old == 12345.f) plain[0] = old; }   // returning
    }
}
template <int MODE> static int run(const char *name, int blocks, int threads, int iters, int naddr) {
    float *sum, *plain; CK(hipMalloc(&sum, 4 * 64 * 1024)); CK(hipMalloc(&plain, 4 << 20));
    CK(hipMemset(sum, 0, 4 * 64 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, sum, plain, iters, 1.0001f, naddr);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, sum, plain, iters, 1.0001f, naddr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("%-28s blocks %7d x %4d threads, %5d iters, %4d addresses: %9.2f us\n", name, blocks, threads, iters, naddr, best * 1e3f);
    CK(hipFree(sum)); CK(hipFree(plain));
    return 0;
}
int main() {
    const int cfg[][3] = {{1660, 512, 2000}, {4096, 512, 2000}, {4096, 512, 0}, {65536, 64, 2000}, {262144, 64, 2000}, {262144, 64, 0}};
    for (auto &c : cfg) {
        run<0>("plain store", c[0], c[1], c[2], 1);
        run<1>("atomic, 1 address", c[0], c[1], c[2], 1);
        run<1>("atomic, 64 addresses", c[0], c[1], c[2], 64);
        run<2>("returning atomic, 1 address", c[0], c[1], c[2], 1);
    }
    return 0;
}
