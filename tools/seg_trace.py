"""Dev tool (GPU): phase timeline of the segment-parallel forward's heaviest workgroups at a bench config.  Needs a library built with
-DSGR_SEG_TRACE:   tools/build_ab.sh trace render.hip -DSGR_SEG_TRACE;  SIGMAN_PY_NODE=1 SIGMAN_GSPLAT_LIB=$PWD/tools/ab/trace.so python tools/seg_trace.py c2
Events (wave 0 of the workgroup): 1 start, 2 loop top, 3 fill done, 4 phase 1 done, 5 prefix barrier passed, 6 phase 2 done, 7 round bookkeeping done,
8 list done, 9 outputs written.  Prints, per traced workgroup, the time spent between consecutive event kinds (us, s_memtime at 100 MHz)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sigman_release_amd import _cabi, cameras, rasterizer as R

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
P, H = cfg["P"], cfg["size"]
sub = bench.build_subject(name, P, {"c1": 0, "c2": 1, "c5": 4}.get(name, 1), dev)[0]
subj = {k: sub[k][None] for k in ("means3D", "cov3D", "opacity", "rgb")}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cv, cvp, cp = cameras.make_cameras([bench.VIEWS[0]])
st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 0.5, t(cv), t(cvp), 0, t(cp), 1)
L = _cabi.lib()
L.sgr_debug_seg_trace.restype = C.c_int; L.sgr_debug_seg_trace.argtypes = [C.c_void_p]
NS, NE = 8, 512
TILES = (H // 16) ** 2
buf = torch.zeros(NS * 4 * NE + TILES * 32, dtype=torch.int64, device=dev)
leaves = [subj[k].clone().requires_grad_(True) for k in ("means3D", "rgb", "opacity", "cov3D")]
FUSED = os.environ.get("FUSED", "0") == "1"       # the fused single-view step through the C++ node (the traced library must be the one the node links: lib/libsigman_gsplat.so)
if FUSED:
    st = R.BatchedRasterizationSettings(H, H, cameras.TAN_HALF_FOV, cameras.TAN_HALF_FOV, torch.ones(3, device=dev), 0.5, t(cv), t(cvp), 0, t(cp), 1, False, 400000 if name != "c5" else 4000000)
    gt = torch.rand(1, 3, H, H, device=dev)
def fwd():
    if FUSED:
        out = R.rasterize_l1_loss_batched(leaves[0], None, None, leaves[1], leaves[2][..., None] if leaves[2].dim() == 2 else leaves[2], None, None, leaves[3], st, gt, None, 1e-3)
        out[0].backward()
        return out
    return R.rasterize_gaussians_batched(leaves[0], None, None, leaves[1], leaves[2][..., None] if leaves[2].dim() == 2 else leaves[2], None, None, leaves[3], st)
for _ in range(5): fwd()
torch.cuda.synchronize()
assert L.sgr_debug_seg_trace(C.c_void_p(buf.data_ptr())) == 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fwd(); torch.cuda.synchronize()
raw = buf.cpu().numpy().astype(np.uint64)
b = raw[:NS * 4 * NE].reshape(NS, 4, NE)
sch = raw[NS * 4 * NE:].reshape(TILES, 4, 8)
names = {1: "start", 2: "loop", 3: "fill", 4: "phase1", 5: "prefix", 6: "phase2", 7: "book", 8: "done", 9: "written"}
t00 = None
for s in range(NS):
    for q in range(4):
        n = int(b[s, q, 0])
        if n < 3: continue
        info = int(b[s, q, 1]); length, bid = info >> 32, info & 0xFFFFFFFF
        ev = [(int(x) >> 8, int(x) & 255) for x in b[s, q, 2:n]]
        if t00 is None: t00 = ev[0][0]
        tot = {}
        for (ta, _), (tb, ib) in zip(ev[:-1], ev[1:]):
            tot[ib] = tot.get(ib, 0) + (tb - ta)
        rounds = sum(1 for _, i in ev if i == 6)
        fills = sum(1 for _, i in ev if i == 3)
        dur = (ev[-1][0] - ev[0][0]) / 100.0
        print(f"slot {s} q{q} tile {bid:5d} n {length:5d} rounds {rounds:2d}  start +{(ev[0][0] - t00) / 100.0:6.2f} us  total {dur:6.2f} us | " +
              "  ".join(f"{names[i]} {v / 100.0:5.2f}" for i, v in sorted(tot.items())))

# ---- schedule of ALL workgroups (s_memrealtime, 100 MHz): start / end relative to the first start, XCD / SE / CU from HW_ID
rows = []
for sl in range(TILES):
    for q in range(4):
        w0, w1, hw, info, wb, nbk = (int(x) for x in sch[sl, q][:6])
        if w0 == 0: continue
        xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
        cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
        rows.append((w0, w1, sl, q, info >> 32, xcc, se, sh, cu, wb, nbk))
t0 = min(r[0] for r in rows)
work = [r for r in rows if r[4] > 0]
tend = max(r[1] for r in work)
print(f"workgroups recorded {len(rows)}, working {len(work)}, span of the working ones {(tend - t0) / 100.0:.2f} us")
print("running working-workgroups over time (2-us bins):", [sum(1 for r in work if r[0] - t0 <= 100 * k * 2 < r[1] - t0) for k in range(0, int((tend - t0) / 200) + 2)])
import collections
per_cu = collections.defaultdict(list)
for r in work: per_cu[(r[5], r[6], r[7], r[8])].append(r)
print("CUs with working workgroups:", len(per_cu))
ends = sorted(((max(x[1] for x in v) - t0) / 100.0, k) for k, v in per_cu.items())
print("last end per CU: min %.1f median %.1f max %.1f us" % (ends[0][0], ends[len(ends) // 2][0], ends[-1][0]))
for e, k in ends[-4:] + ends[:2]:
    v = sorted(per_cu[k])
    print(f"  CU xcc{k[0]} se{k[1]} sh{k[2]} cu{k[3]}: " + "  ".join(f"[slot {x[2]} q{x[3]} n {x[4]} {(x[0] - t0) / 100.0:.1f}-{(x[1] - t0) / 100.0:.1f}]" for x in v))
starts = sorted((r[0] - t0) / 100.0 for r in work)
print("start times of working workgroups (us), every 53rd:", [round(x, 1) for x in starts[::53]])
durs = sorted(((r[1] - r[0]) / 100.0, r[4]) for r in work)
print("durations (us, n): shortest", durs[:3], "median", durs[len(durs) // 2], "longest", durs[-3:])


# ---- where the slot time goes: sum of workgroup lifetimes by list length (the kernel's length is the work per slot: DESIGN dead ends (ak))
cls = [(0, 256), (256, 512), (512, 768), (768, 1024), (1024, 1536), (1536, 2048), (2048, 1 << 30)]
tot = sum(r[1] - r[0] for r in work) / 100.0
print("workgroup-time by list length [n0, n1): workgroups, sum of lifetimes us (share), mean lifetime us, mean start us")
for a, b in cls:
    w = [r for r in work if a <= r[4] < b]
    if w:
        s = sum(r[1] - r[0] for r in w) / 100.0
        print(f"  [{a:5d},{b if b < 1 << 30 else 99999:5d}): {len(w):4d} wgs  {s:8.1f} us ({100 * s / tot:4.1f} %)  mean {s / len(w):5.1f} us  mean start {sum(r[0] - t0 for r in w) / 100.0 / len(w):5.1f} us")
print(f"  all: {len(work)} wgs {tot:.1f} us = {tot / 512:.1f} us per slot of 512")

# ---- rounds per workgroup (a round composites up to 512 survivors; the last one of a list is usually small and pays the full set of barriers)
by_r = collections.defaultdict(list)
for r in work: by_r[min(r[9], 4)].append(r)
print("workgroups by number of rounds: rounds, workgroups, mean lifetime us, mean n, mean survivors, survivors of the LAST round (mean)")
for k in sorted(by_r):
    v = by_r[k]
    print(f"  {k}: {len(v):4d} wgs  mean {sum(x[1] - x[0] for x in v) / 100.0 / len(v):5.1f} us  n {sum(x[4] for x in v) / len(v):6.0f}  survivors {sum(x[10] for x in v) / len(v):6.0f}  last round {sum((x[10] - 512 * (k - 1)) if k >= 1 else 0 for x in v) / len(v):5.0f}")
two = [x for x in work if x[9] == 2 and x[10] <= 1020]
print(f"two-round workgroups whose survivors would fit ONE round of 16 segments: {len(two)}, sum of lifetimes {sum(x[1] - x[0] for x in two) / 100.0:.0f} us")
