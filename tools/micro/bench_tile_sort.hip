// Dev micro-benchmark: the per-tile register sort of csrc/binning.hip on synthetic tile lists (hipcc, run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-honor-nans tools/micro/bench_tile_sort.hip -o tools/micro/bts
//   tools/micro/bts [tools/micro/tile_hist_c3.txt ...]      (files: "<tile length> <count>" lines, e.g. from tools/tile_hist.py)
#include "../../sigman_release_amd/csrc/binning.hip"
#include "../../sigman_release_amd/csrc/api.hip"
#include <string.h>
#include <vector>
#include <random>
#include <algorithm>
#include <cstdio>
#include <string>
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

static void run_case(const std::string &name, std::vector<uint32_t> len) {
    std::mt19937 rng(1);
    std::shuffle(len.begin(), len.end(), rng);
    const int ntiles = (int)len.size();
    size_t R = 0;
    std::vector<uint2> hr(ntiles);
    for (int t = 0; t < ntiles; t++) { hr[t] = make_uint2((uint32_t)R, (uint32_t)(R + len[t])); R += len[t]; }
    std::vector<uint64_t> hk(R); std::vector<uint32_t> hv(R);
    for (int t = 0; t < ntiles; t++)
        for (uint32_t k = 0; k < len[t]; k++) {
            float z = 2.3f + 0.2f * (rng() % 4000) / 4000.f;                 // few distinct depths: plenty of ties (stability is checked)
            uint32_t zb; memcpy(&zb, &z, 4);
            hk[hr[t].x + k] = ((uint64_t)t << 32) | zb; hv[hr[t].x + k] = hr[t].x + k;
        }
    // scramble the input order inside each tile (the scatter pass is order-free): the sort must order by (depth, value)
    for (int t = 0; t < ntiles; t++) {
        std::vector<uint32_t> perm(len[t]);
        for (uint32_t k = 0; k < len[t]; k++) perm[k] = k;
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint64_t> tk(len[t]); std::vector<uint32_t> tv(len[t]);
        for (uint32_t k = 0; k < len[t]; k++) { tk[k] = hk[hr[t].x + perm[k]]; tv[k] = hv[hr[t].x + perm[k]]; }
        std::copy(tk.begin(), tk.end(), hk.begin() + hr[t].x); std::copy(tv.begin(), tv.end(), hv.begin() + hr[t].x);
    }
    // worklists by class, like vseg_scan: [m] = tiles of <= 1024 << m entries
    std::vector<uint32_t> lists[5];
    for (int t = 0; t < ntiles; t++) {
        if (!len[t]) continue;
        int m = 0; while (m < 4 && len[t] > (1024u << m)) m++;
        if (len[t] > 16384u) { printf("tile too long\n"); exit(1); }
        lists[m].push_back(t);
    }
    uint64_t *ka, *kb; uint32_t *va, *vb, *list, *cnt; uint2 *ranges;
    CK(hipMalloc(&ka, R * 8 + 64)); CK(hipMalloc(&kb, R * 8 + 64)); CK(hipMalloc(&va, R * 4 + 64)); CK(hipMalloc(&vb, R * 4 + 64));
    CK(hipMalloc(&list, (size_t)5 * ntiles * 4 + 64)); CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&ranges, (size_t)ntiles * 8));
    { std::vector<uint64_t> hc(R); for (size_t i = 0; i < R; i++) hc[i] = (hk[i] << 32) | hv[i];      // what the scatter pass leaves: (depth, value) composites
      CK(hipMemcpy(ka, hc.data(), R * 8, hipMemcpyHostToDevice)); }
    CK(hipMemcpy(ranges, hr.data(), (size_t)ntiles * 8, hipMemcpyHostToDevice));
    uint32_t h[16] = {0};
    TileWork4 tw4;
    tw4.w[5] = TileWork{list, cnt + 15, cnt + 14};                                     // no tile beyond 16384 entries here: cnt[14] = 0
    for (int m = 0; m < 5; m++) {
        if (!lists[m].empty()) CK(hipMemcpy(list + (size_t)m * ntiles, lists[m].data(), lists[m].size() * 4, hipMemcpyHostToDevice));
        h[m] = (uint32_t)lists[m].size();                                         // cnt[0..4] counts, cnt[8..12] tickets
        tw4.w[m] = TileWork{list + (size_t)m * ntiles, cnt + 8 + m, cnt + m};
    }
    auto reset = [&]() { CK(hipMemcpyAsync(cnt, h, 64, hipMemcpyHostToDevice, 0)); };
    const float us = time_it([&]() { reset(); hipLaunchKernelGGL(tile_sort_regs_kernel<16>, dim3(256), dim3(1024), 0, 0, ranges, ka, va, kb, vb, tw4, 4, 0, SortPrep{nullptr, 0, nullptr, 0, 0}, getenv("BTS_NOKEYS") ? 0 : 1); });
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> ok(R); std::vector<uint32_t> ov(R);
    CK(hipMemcpy(ok.data(), kb, R * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), vb, R * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int t = 0; t < ntiles; t++) {
        std::vector<std::pair<uint64_t, uint32_t>> ref(len[t]);
        for (uint32_t k = 0; k < len[t]; k++) ref[k] = {hk[hr[t].x + k], hv[hr[t].x + k]};
        std::sort(ref.begin(), ref.end());
        for (uint32_t k = 0; k < len[t]; k++) if ((!getenv("BTS_NOKEYS") && ok[hr[t].x + k] != ref[k].first) || ov[hr[t].x + k] != ref[k].second) { bad++; break; }
    }
    printf("%-28s tiles %6d keys %9zu classes [%zu %zu %zu %zu %zu]: %8.1f us  %6.1f keys/ns  %s\n", name.c_str(), ntiles, R, lists[0].size(), lists[1].size(),
           lists[2].size(), lists[3].size(), lists[4].size(), us, R / us / 1000.0, bad ? "WRONG" : "ok");
    if (bad) printf("   %zu tiles wrong\n", bad);
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(list); hipFree(cnt); hipFree(ranges);
}

int main(int argc, char **argv) {
    struct Case { int ntiles; uint32_t n; };
    const std::vector<Case> cases = {{1, 100}, {1, 1000}, {1, 2000}, {1, 4000}, {1, 8000}, {12000, 250}, {12000, 1000}, {6000, 2000}, {3000, 4000}, {1500, 8000}, {750, 16000}, {3000, 1100}, {3000, 2100}};
    if (const char *only = getenv("BTS_ONLY")) {                                      // e.g. BTS_ONLY=12000x1000 (for rocprofv3 counter runs)
        int nt; unsigned n;
        if (sscanf(only, "%dx%u", &nt, &n) == 2) run_case(only, std::vector<uint32_t>(nt, n));
        return 0;
    }
    for (auto c : cases) run_case(std::to_string(c.ntiles) + " x " + std::to_string(c.n), std::vector<uint32_t>(c.ntiles, c.n));
    for (int i = 1; i < argc; i++) {
        FILE *f = fopen(argv[i], "r");
        if (!f) { printf("cannot open %s\n", argv[i]); continue; }
        std::vector<uint32_t> len; unsigned n, c;
        while (fscanf(f, "%u %u", &n, &c) == 2) for (unsigned k = 0; k < c; k++) len.push_back(n);
        fclose(f);
        run_case(argv[i], len);
    }
    return 0;
}
