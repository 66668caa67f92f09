// Dev micro-benchmark: the per-tile register sort of csrc/binning.hip on synthetic tile lists (hipcc, run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-honor-nans tools/micro/bench_tile_sort.hip -o tools/micro/bts
//   tools/micro/bts [tools/micro/tile_hist_c3.txt ...]      (files: "<tile length> <count>" lines, e.g. from tools/tile_hist.py)
#define SGR_DEEP_TIMING 1
#include "../../sigman_release_amd/csrc/binning.hip"
#include "../../sigman_release_amd/csrc/tile_sort.hip"
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


// the LDS distribution sort (deep_tile_kernel, both instantiations) + register sort for the tiles of <= deep_min entries: timed on their
// own, the result checked.       BTS_DEEP=<deep_min> [BTS_STAMPS=1] [BTS_SORTGRID=64] tools/micro/bts file...
static void run_deep_case(const std::string &name, std::vector<uint32_t> len, uint32_t deep_min) {
    std::mt19937 rng(1);
    std::shuffle(len.begin(), len.end(), rng);
    const int ntiles = (int)len.size();
    size_t R = 0;
    std::vector<uint2> hr(ntiles);
    for (int t = 0; t < ntiles; t++) { hr[t] = make_uint2((uint32_t)R, (uint32_t)(R + len[t])); R += len[t]; }
    std::vector<uint64_t> hc(R);
    for (int t = 0; t < ntiles; t++)
        for (uint32_t k = 0; k < len[t]; k++) {
            float z = 2.3f + 0.2f * (rng() % 40000) / 40000.f;
            uint32_t zb; memcpy(&zb, &z, 4);
            // unique values per tile (the sort is by the full composite; the order inside the tile is arbitrary already)
            hc[hr[t].x + k] = ((uint64_t)zb << 32) | (hr[t].x + (uint32_t)(((uint64_t)k * 2654435761ull) % len[t]));
        }
    const uint32_t stride = ntiles + (uint32_t)(R / (kDeepBigCap - kDeepBinMax)) + 1;
    std::vector<uint32_t> lists[8];
    for (int t = 0; t < ntiles; t++) {
        if (!len[t]) continue;
        int m = 0; while (m < 5 && len[t] > (1024u << m)) m++;
        if (len[t] > deep_min) {
            if (len[t] <= kDeepSmallCap - kDeepBinMax) lists[7].push_back(t);
            else for (uint32_t w = 0; w < (len[t] + (kDeepBigCap - kDeepBinMax) - 1) / (kDeepBigCap - kDeepBinMax); w++) lists[6].push_back(t | (w << 26));
            continue;
        }
        lists[m].push_back(t);
    }
    uint64_t *ka, *kb; uint32_t *va, *vb, *list; uint2 *ranges; VsegPlan *plan;
    CK(hipMalloc(&ka, R * 8 + 64)); CK(hipMalloc(&kb, R * 8 + 64)); CK(hipMalloc(&va, R * 4 + 64)); CK(hipMalloc(&vb, R * 4 + 64));
    CK(hipMalloc(&list, (size_t)8 * stride * 4 + 64)); CK(hipMalloc(&plan, sizeof(VsegPlan))); CK(hipMalloc(&ranges, (size_t)ntiles * 8));
    CK(hipMemcpy(ka, hc.data(), R * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(ranges, hr.data(), (size_t)ntiles * 8, hipMemcpyHostToDevice));
    VsegPlan hp; memset(&hp, 0, sizeof(hp));
    TileWork4 tw4;
    for (int m = 0; m < 8; m++) {
        if (!lists[m].empty()) CK(hipMemcpy(list + (size_t)m * stride, lists[m].data(), lists[m].size() * 4, hipMemcpyHostToDevice));
        hp.count[m] = (uint32_t)lists[m].size();
        if (m < 6) tw4.w[m] = TileWork{list + (size_t)m * stride, &plan->ticket[m], &plan->count[m]};
    }
    auto reset = [&]() { CK(hipMemcpyAsync(plan, &hp, sizeof(hp), hipMemcpyHostToDevice, 0)); };
    auto deep = [&]() {
        hipLaunchKernelGGL((deep_tile_kernel<1024, kDeepBigCap, 4096>), dim3(std::max(1u, std::min((uint32_t)lists[6].size(), 256u))), dim3(1024), 0, 0, ka, va, kb, vb, &plan->count[6], list + (size_t)6 * stride, ranges, 1, plan, list, stride, SortPrep{nullptr, 0, nullptr, 0, 0});
        hipLaunchKernelGGL((deep_tile_kernel<256, kDeepSmallCap, 1024>), dim3(std::max(1u, std::min((uint32_t)lists[7].size(), 768u))), dim3(256), 0, 0, ka, va, kb, vb, &plan->count[7], list + (size_t)7 * stride, ranges, 1, plan, list, stride, SortPrep{nullptr, 0, nullptr, 0, 0}); };
    auto sort = [&]() { hipLaunchKernelGGL(tile_sort_regs_kernel<16>, dim3(getenv("BTS_SORTGRID") ? atoi(getenv("BTS_SORTGRID")) : 256), dim3(1024), 0, 0, ranges, ka, va, kb, vb, tw4, 4, 0, SortPrep{nullptr, 0, nullptr, 0, 0}, 1); };
    // events around each stage only (the counter reset is a host-to-device copy)
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float best_d = 1e30f, best_s = 1e30f;
    for (int i = 0; i < 6; i++) {
        reset(); CK(hipEventRecord(e0)); deep(); CK(hipEventRecord(e1)); sort(); CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
        float md, ms; CK(hipEventElapsedTime(&md, e0, e1)); CK(hipEventElapsedTime(&ms, e1, e2));
        if (i) { best_d = std::min(best_d, md); best_s = std::min(best_s, ms); }
    }
    VsegPlan after; CK(hipMemcpy(&after, plan, sizeof(after), hipMemcpyDeviceToHost));
    std::vector<uint64_t> ok(R); std::vector<uint32_t> ov(R);
    CK(hipMemcpy(ok.data(), kb, R * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), vb, R * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int t = 0; t < ntiles; t++) {
        std::vector<uint64_t> ref(hc.begin() + hr[t].x, hc.begin() + hr[t].y);
        std::sort(ref.begin(), ref.end());
        for (uint32_t k = 0; k < len[t]; k++) if (ov[hr[t].x + k] != (uint32_t)ref[k] || (uint32_t)ok[hr[t].x + k] != (uint32_t)(ref[k] >> 32)) { bad++; break; }
    }
    if (getenv("BTS_STAMPS")) {
        std::vector<unsigned long long> st(1024 * 16);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(sgr_deep_dbg), st.size() * 8));
        unsigned long long t0 = ~0ull; for (int b = 0; b < std::min(ntiles, 1024); b++) if (st[b * 16]) t0 = std::min(t0, st[b * 16]);
        const std::vector<uint32_t> &wl = lists[6].empty() ? lists[7] : lists[6];
        for (int b : {0, 1, 2, 50, 100, 171, 200, 255, 256, 300, 436}) if (b < 1024 && (size_t)b < wl.size()) {
            printf("  wg %3d n %6u start %6.2f us | phases (us):", b, len[wl[b] & kDeepTileMask], (st[b * 16] - t0) * 0.01);
            for (int p = 1; p <= 7; p++) printf(" %5.2f", (double)(st[b * 16 + p] - st[b * 16 + p - 1]) * 0.01);
            printf("\n");
        }
    }
    printf("DEEP %-28s min %5u tiles %6d (windows %zu, small %zu) keys %9zu: deep %7.1f us, register sort %7.1f us | classes left [%u %u %u %u %u %u] %s\n", name.c_str(), deep_min, ntiles,
           lists[6].size(), lists[7].size(), R, best_d * 1000.f, best_s * 1000.f, after.count[0], after.count[1], after.count[2], after.count[3], after.count[4], after.count[5], bad ? "WRONG" : "ok");
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(list); hipFree(plan); hipFree(ranges);
}

int main(int argc, char **argv) {
    struct Case { int ntiles; uint32_t n; };
    const std::vector<Case> cases = {{1, 100}, {1, 1000}, {1, 2000}, {1, 4000}, {1, 8000}, {12000, 250}, {12000, 1000}, {6000, 2000}, {3000, 4000}, {1500, 8000}, {750, 16000}, {3000, 1100}, {3000, 2100}};
    if (const char *only = getenv("BTS_ONLY")) {                                      // e.g. BTS_ONLY=12000x1000 (for rocprofv3 counter runs)
        int nt; unsigned n;
        if (sscanf(only, "%dx%u", &nt, &n) == 2) run_case(only, std::vector<uint32_t>(nt, n));
        return 0;
    }
    if (const char *dm = getenv("BTS_DEEP")) {
        for (int i = 1; i < argc; i++) {
            FILE *f = fopen(argv[i], "r");
            if (!f) { printf("cannot open %s\n", argv[i]); continue; }
            std::vector<uint32_t> len; unsigned n, c;
            while (fscanf(f, "%u %u", &n, &c) == 2) for (unsigned k = 0; k < c; k++) len.push_back(n);
            fclose(f);
            run_deep_case(argv[i], len, (uint32_t)atoi(dm));
        }
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
