# rocprofv3 kernel stats of bench.py at one config for SIGMAN_FUSED_STEP = 0 / 1 / 2 (one box)     usage: gpu_fused_kstats.sh [config]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; c=${1:-c2}; cd /tmp
for f in ${MODES:-0 1 2}; do
SIGMAN_FUSED_STEP=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_${c}_$f -o x -- python $R/bench.py --config $c --no-variants --no-cpu-baseline > /tmp/ks_${c}_$f.json 2>/dev/null
echo "== fused step $f: $(python3 -c "import json; d=json.load(open('/tmp/ks_${c}_$f.json')); print(d['ms_per_step'], d['windows']['wall_ms_per_step_min_median_max'])")"
python3 $R/tools/kstats.py $(find /tmp/ks_${c}_$f -name '*kernel_stats.csv' | head -1) 1000
done
