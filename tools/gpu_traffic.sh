# HBM traffic + time of the compositing kernels for one bench config under a set of env settings (dev A/B)
# usage: gpu_traffic.sh <config> "<ENV=.. ENV=..>" ["<ENV..>" ...]        (each quoted group = one variant; "" = defaults)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; c=$1; shift
i=0
for envs in "$@"; do
  i=$((i+1)); D=/tmp/tr_$i; rm -rf $D
  B="python $R/bench.py --config $c --no-cpu-baseline --no-variants --steps 4 --warmup 2"
  (cd /tmp; env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o x -- $B > /dev/null 2>&1
   env $envs rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f -o f -- $B > /dev/null 2>&1
   env $envs rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w -o w -- $B > /dev/null 2>&1)
  python3 - "$D" "$envs" "${PATS:-render}" <<'PY'
import csv, glob, re, statistics, sys
D, envs, pats = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
def short(n):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n); return m.group(1) if m else n[:50]
def pmc(d, name):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return per
fe, wr = pmc(D + "/f", "FETCH_SIZE"), pmc(D + "/w", "WRITE_SIZE")
st = {}
for f in glob.glob(D + "/ks/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): st[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print("== [%s]" % envs)
for k in sorted(st, key=lambda k: -st[k][0] * st[k][1]):
    if any(p in k for p in pats) and st[k][1] > 3:
        f = statistics.median(fe.get(k, [0])) * 2 / 1024; w = statistics.median(wr.get(k, [0])) / 1024
        print("%-52s %8.1f us  fetch %8.1f MB  write %8.1f MB  -> %.2f TB/s" % (k, st[k][0], f, w, (f + w) / st[k][0] if st[k][0] else 0))
PY
done
