# round 3: full gpu suite + one bench line per config (+ optional rocprof kernel stats of the configs given after the tag)      usage: bash tools/gpu_r3a.sh <tag> [configs to profile...]
T=${1:-r3a}; shift; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=1200 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for c in c2 c3 c4 c5; do timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; tail -3 $O/bench_$c.err | grep -v amdgpu; done
export TMPDIR=/tmp; cd /tmp
for c in "$@"; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -o $c -- python $GRAFT_REPO_ROOT/bench.py --config $c --no-cpu-baseline --no-variants --steps 10 --warmup 3 > $O/bench_${c}_under_rocprof.json 2> $O/ks_$c.err
cp $(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1) $O/${c}_kernel_stats.csv
echo "== $c"; python $GRAFT_REPO_ROOT/tools/kstats.py $O/${c}_kernel_stats.csv 20 | head -24
done
