// dev micro-test: do raw buffer loads return zeros for out-of-range offsets on this box, and the data for in-range ones?  (yes: 0 mismatches)
// Pitfall: __builtin_bit_cast(float, a.y) on an ext-vector ELEMENT reads element 0 with ROCm 7.2's clang; __uint_as_float(a.y) is right.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/bufoob.hip -o tools/micro/bufoob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *p, unsigned nbytes, const unsigned *offs, float *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)nbytes, 0x00020000);
    const int o = (int)offs[threadIdx.x];
    u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0);
    u2 b = __builtin_amdgcn_raw_buffer_load_b64(r, o + 16, 0, 0);
    float *d = out + threadIdx.x * 6;
    d[0] = __uint_as_float(a.x); d[1] = __uint_as_float(a.y); d[2] = __uint_as_float(a.z); d[3] = __uint_as_float(a.w);
    d[4] = __uint_as_float(b.x); d[5] = __uint_as_float(b.y);
}
int main() {
    const int n = 4096;
    std::vector<float> h(n); for (int i = 0; i < n; i++) h[i] = (float)(i + 1);
    float *dp, *dout; unsigned *doffs;
    hipMalloc(&dp, n * 4 * 2); hipMemcpy(dp, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dp + n, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> offs(64);
    for (int i = 0; i < 64; i++) offs[i] = (i % 4 == 3) ? 0x80000000u : (i % 4 == 2 ? (unsigned)(n * 4 - 8) : (unsigned)(i * 40));
    hipMalloc(&doffs, 256); hipMemcpy(doffs, offs.data(), 256, hipMemcpyHostToDevice);
    hipMalloc(&dout, 64 * 6 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dp, (unsigned)(n * 4), doffs, dout);
    std::vector<float> o(64 * 6); hipMemcpy(o.data(), dout, 64 * 6 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) {
        for (int j = 0; j < 6; j++) {
            const unsigned byte = offs[i] + 4 * j;
            const float want = (offs[i] == 0x80000000u || byte + 4 > (unsigned)(n * 4)) ? 0.f : h[byte / 4];
            if (o[i * 6 + j] != want) { if (bad < 8) printf("lane %d elem %d off %u: got %g want %g\n", i, j, offs[i], o[i * 6 + j], want); bad++; }
        }
    }
    printf("bufoob: %d mismatches\n", bad);
    return bad != 0;
}
