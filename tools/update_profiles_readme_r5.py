#!/usr/bin/env python3
"""Append / refresh the "Round 5" section of profiles/README.md from the r05_* files."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")

def j(name):
    with open(os.path.join(P, name)) as f:
        return json.loads(f.readline() if name.startswith("r05_bench") else f.read())

def sq_row(cfg, key, label):
    d = j(f"r05_sq_{cfg}.json")["kernels"].get(key)
    if not d:
        return None
    us = d["avg_us"]; valu = d["SQ_INSTS_VALU"]; rate = valu / (us * 1e-6)
    return (f"| {cfg.upper()} `{label}` | {us:.0f} | {valu:.3g} | {d['SQ_INSTS_SALU']:.3g} | {rate:.3g} | "
            f"{100 * rate / 8.55e11:.0f} % | {d['SQ_INSTS_LDS']:.3g} | {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.2f} |")

out = ["## Round 5", ""]
out.append("Collected by `tools/gpu_evidence.sh` (= `tools/gpu_profile.sh <config>` per configuration: FOUR separate rocprofv3 runs of\n"
           "`python bench.py --config <c> --no-cpu-baseline --no-variants --steps 5 --warmup 3` -- `--kernel-trace --stats`, `--pmc FETCH_SIZE --kernel-trace`,\n"
           "`--pmc WRITE_SIZE --kernel-trace`, and `--pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-trace` --\n"
           "plus `tools/pmc_sq.sh` for the SQ counters), on the round's final kernels (commit in `r05_pmc_*.json`).")
out.append("")
out.append("| file | what |\n|---|---|")
out.append("| `r05_{c2,c3,c4,c5}_kernel_stats.csv`, `r05_pmc_{c2,c3,c4,c5}.json`, `r05_{c2,c3,c4,c5}_table.md` | kernel stats, FETCH / WRITE and DRAM-side summaries (`bench.py` reads `roofline.traffic` from the newest `r0N_pmc_<config>.json`), the tables below; the `preprocess_fwd+scan` row merges the scan of the tile counts with the kernel that produces them (84 B per Gaussian and view for the pair) |")
out.append("| `r05_sq_{c2,c3,c4}.json` | SQ counters of the compositing kernels and the backward gather; `bench.py` derives `roofline_valu_issue` from them |")
b = {c: j(f"r05_bench_{c}.json") for c in ("c1", "c2", "c3", "c4", "c5")}
out.append("| `r05_bench_{c1,c2,c3,c4,c5}.json` | full bench lines (variants, windows, cpu_baseline, masked L1 headline): "
           + ", ".join(f"{c.upper()} {b[c]['ms_per_step']:.4g}" for c in b) + " ms per step |")
out.append("| `r05_bench_c2_driver_cmd.json` | `python bench.py --gpus 1 --steps 20 --warmup 5`, the driver's command, on the round's last commit right behind the full GPU suite (214 passed) and `smoke()`: "
           + f"{j('r05_bench_c2_driver_cmd.json')['ms_per_step']:.4f} ms per step, {j('r05_bench_c2_driver_cmd.json')['value']:.0f} views/s |")
out.append("| `r05_{c2,c3,c4,c5}_bench.json`, `r05_*_bench_under_rocprof.json` | the bench line of the profiled command without / under rocprofv3 |")
out.append("| `r05_bench_*_n2gloo.json` | `SIGMAN_BENCH_BACKEND=gloo python bench.py --gpus 2 --config <c> [--exchange ...]`: the 2-rank path on ONE GPU, host-staged collectives -- plumbing, not measurements |")
out.append("| `r05_fused_step_ab.txt` | the fused single-view step against the unfused chain on one box (`SIGMAN_FUSED_STEP=0/1`, three alternating runs) |")
out.append("| `r05_count_wait_ab.txt`, `r05_bench_c2_lazy4_boxC.json` | who limits a slow C2 reading: `host_queue` of the bench line under the library's count-wait modes (own / lazy / lazy:4), on quiet hosts and with one busy loop per host core; a full line taken in a GPU-side slow phase |")
out.append("| `r05_fuzz_parity.txt` | the round's randomised bit-for-bit sweeps (fused step, determinism, per-view pattern, host threads, 3-NN) |")
out.append("| `r05_rocprofv3_counters_avail.txt` | `rocprofv3 --list-avail` of the box: there is no MALL / Infinity-Cache hit counter on gfx950 in this ROCm; the DRAM-side TCC_EA0 counters below are what exists |")
out.append("")
w = b["c2"].get("windows") or {}
out.append(f"C2 headline line: {b['c2']['value']:.0f} views/s, {b['c2']['ms_per_step']:.4f} ms per step wall, gpu_ms_per_step {b['c2'].get('gpu_ms_per_step')}, "
           f"windows {json.dumps(w)}, sclk {b['c2'].get('sclk_mhz')}, sclk_mhz_probe {json.dumps({k: v for k, v in (b['c2'].get('sclk_mhz_probe') or {}).items() if k != 'how'})}, "
           f"host_queue {json.dumps({k: v for k, v in (b['c2'].get('host_queue') or {}).items() if k != 'note'})}, count_check: {b['c2']['config'].get('count_check')}.")
out.append("")
out.append("DRAM-side check (`ea_dram_bytes` in `r05_pmc_*.json` = TCC_EA0_RDREQ_DRAM_32B x 32 B + write requests x 64 / 32 B): it equals 2 x FETCH_SIZE + WRITE_SIZE\n"
           "to within 1 % on every kernel group of C2-C5, i.e. the L2 counters already count only what leaves L2 towards the fabric; whether a request is then served\n"
           "by the 256 MB Infinity Cache or by HBM is not observable with these counters, so `roofline.traffic` is an UPPER bound of the HBM bytes.")
out.append("")
out.append("| config | kernel group | 2 x FETCH + WRITE (MB) | TCC_EA0 DRAM-side (MB) |\n|---|---|---|---|")
for c in ("c2", "c3", "c4", "c5"):
    d = j(f"r05_pmc_{c}.json")
    for k, v in d["kernels"].items():
        if "ea_dram_bytes" in v and k.startswith("render"):
            out.append(f"| {c.upper()} | {k} | {v['hbm_bytes_corrected'] / 1e6:.2f} | {v['ea_dram_bytes'] / 1e6:.2f} |")
out.append("")
out.append("SQ counters (median per launch; issue rate = VALU instructions per second against the 8.55e11/s of `tools/micro/valu_rate.hip`; last column: SQ_WAIT_ANY / SQ_WAVE_CYCLES):")
out.append("")
out.append("| kernel | µs | VALU instructions | scalar instructions | VALU per second | of the issue rate | LDS instructions | waiting |\n|---|---|---|---|---|---|---|---|")
for cfg, key, label in (("c2", "render_fwd", "render_fwd_seg"), ("c2", "render_bwd", "render_bwd_bucket"),
                        ("c3", "render_fwd", "render_fwd_wave"), ("c3", "render_bwd", "render_bwd_bucket"),
                        ("c3", "preprocess_bwd", "preprocess_bwd_lanes"), ("c4", "render_fwd", "render_fwd_wave")):
    r = sq_row(cfg, key, label)
    if r:
        out.append(r)
out.append("")
names = {"c2": "100 000 Gaussians, 1 view 512²", "c3": "100 000 Gaussians, 64 views 512²",
         "c4": "200 000 Gaussians, 90 views 1024², forward only", "c5": "1M Gaussians, 1 view 512², depth + alpha gradients"}
for c in ("c2", "c3", "c4", "c5"):
    d = j(f"r05_pmc_{c}.json")
    out.append(f"**{c.upper()}** ({names[c]}, R = {d['num_rendered']:.2g}; step {b[c]['ms_per_step']:.4g} ms):")
    out.append("")
    out.append(open(os.path.join(P, f"r05_{c}_table.md")).read().rstrip())
    out.append("")
text = "\n".join(out).rstrip() + "\n"
path = os.path.join(P, "README.md")
cur = open(path).read()
cur = re.sub(r"\n## Round 5\n.*\Z", "\n", cur, flags=re.S).rstrip() + "\n\n" + text
open(path, "w").write(cur)
print(text[:3000])
