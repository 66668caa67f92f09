// Dev micro-benchmark: the per-tile sort kernels of csrc/binning.hip on synthetic tile lists (hipcc, run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I sigman_release_amd/csrc tools/micro/bench_tile_sort.hip -o /tmp/bts
#include "../../sigman_release_amd/csrc/binning.hip"
#include "../../sigman_release_amd/csrc/api.hip"
#include <string.h>
#include <vector>
#include <random>
#include <algorithm>
#include <cstdio>
int sgr_validate_problem(const SgrProblem *) { return 0; }
int32_t sgr_preprocess_blocks_per_view(int32_t P) { return (P + 255) / 256; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F> static float time_it(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms); }
    return best * 1000.f;
}

int main(int argc, char **argv) {
    std::mt19937 rng(1);
    struct Case { int ntiles, n; };
    std::vector<Case> cases = {{1, 100}, {1, 1000}, {1, 2000}, {1, 4000}, {1, 8000}, {3000, 1000}, {12000, 1000}, {12000, 500}, {12000, 250}, {3000, 2000}, {1500, 4000}, {600, 8000}, {1, 16000}, {300, 16000}};
    for (auto c : cases) {
        const size_t R = (size_t)c.ntiles * c.n;
        std::vector<uint64_t> hk(R); std::vector<uint32_t> hv(R); std::vector<uint2> hr(c.ntiles); std::vector<uint32_t> hl(c.ntiles);
        for (int t = 0; t < c.ntiles; t++) {
            hr[t] = make_uint2((uint32_t)((size_t)t * c.n), (uint32_t)((size_t)(t + 1) * c.n)); hl[t] = t;
            for (int k = 0; k < c.n; k++) { float z = 2.3f + 0.2f * (rng() % 100000) / 100000.f; uint32_t zb; memcpy(&zb, &z, 4); hk[(size_t)t * c.n + k] = ((uint64_t)t << 32) | zb; hv[(size_t)t * c.n + k] = (uint32_t)((size_t)t * c.n + k); }
        }
        uint64_t *ka, *kb; uint32_t *va, *vb, *list, *cnt; uint2 *ranges;
        CK(hipMalloc(&ka, R * 8)); CK(hipMalloc(&kb, R * 8)); CK(hipMalloc(&va, R * 4)); CK(hipMalloc(&vb, R * 4)); CK(hipMalloc(&list, c.ntiles * 4)); CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&ranges, c.ntiles * 8));
        CK(hipMemcpy(ka, hk.data(), R * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(va, hv.data(), R * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(list, hl.data(), c.ntiles * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ranges, hr.data(), c.ntiles * 8, hipMemcpyHostToDevice));
        uint32_t h2[2] = {(uint32_t)c.ntiles, 0};
        auto reset = [&]() { CK(hipMemcpyAsync(cnt, h2, 8, hipMemcpyHostToDevice, 0)); };
        TileWork w = {list, cnt + 1, cnt};
        auto grid = [&](uint32_t per_cu) { return std::min<uint32_t>(c.ntiles, per_cu * 256u); };
        TileWork4 tw4; TileWork none = {list, cnt + 3, cnt + 2};     // cnt[2] = 0 tiles, cnt[3] ticket
        for (int k = 0; k < 5; k++) tw4.w[k] = none;
        const int cls = c.n <= 1024 ? 0 : (c.n <= 2048 ? 1 : (c.n <= 4096 ? 2 : (c.n <= 8192 ? 3 : 4)));
        tw4.w[cls] = w;
        uint32_t h4[4] = {(uint32_t)c.ntiles, 0, 0, 0};
        auto reset4 = [&]() { CK(hipMemcpyAsync(cnt, h4, 16, hipMemcpyHostToDevice, 0)); };
        float us = 0, us64 = 0; const char *which = "regs<16>"; uint32_t nwide = 0;
        us = time_it([&]() { reset4(); hipLaunchKernelGGL(tile_sort_regs_kernel<16>, dim3(grid(1)), dim3(1024), 0, 0, ranges, ka, va, kb, vb, tw4, 4, 0); });
        CK(hipDeviceSynchronize());
        // verify
        std::vector<uint64_t> ok(R); std::vector<uint32_t> ov(R);
        CK(hipMemcpy(ok.data(), kb, R * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), vb, R * 4, hipMemcpyDeviceToHost));
        bool good = true;
        for (int t = 0; t < std::min(c.ntiles, 50) && good; t++) {
            std::vector<std::pair<uint64_t, uint32_t>> ref(c.n);
            for (int k = 0; k < c.n; k++) ref[k] = {hk[(size_t)t * c.n + k], hv[(size_t)t * c.n + k]};
            std::stable_sort(ref.begin(), ref.end(), [](auto &x, auto &y) { return x.first < y.first; });
            for (int k = 0; k < c.n; k++) if (ok[(size_t)t * c.n + k] != ref[k].first || ov[(size_t)t * c.n + k] != ref[k].second) { good = false; break; }
        }
        // old kernels for comparison
        float us_old = 0;
        if (c.n <= 1024) us_old = time_it([&]() { reset(); hipLaunchKernelGGL((tile_sort_dyn_kernel<256, 1024>), dim3(grid(8)), dim3(256), 0, 0, ranges, ka, va, kb, vb, w); });
        else if (c.n <= 4096) us_old = time_it([&]() { reset(); hipLaunchKernelGGL((tile_sort_dyn_kernel<1024, 4096>), dim3(grid(1)), dim3(1024), 0, 0, ranges, ka, va, kb, vb, w); });
        float us_wave = 0;
        printf("tiles %6d x %5d keys: %-10s u32 %8.1f us (%s, %u wide)  u64 %8.1f us   lds-block %8.1f us   lds-wave %8.1f us   keys/us u32 %.0f\n", c.ntiles, c.n, which, us, good ? "ok" : "WRONG", nwide, us64, us_old, us_wave, R / us);
        hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(list); hipFree(cnt); hipFree(ranges);
    }
    return 0;
}
