set -x
O=gpurun_out/r2b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 -k "sort or multiview or full_size or batched or seeded" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for c in c3 c4 c5; do timeout 600 python bench.py --config $c --no-variants --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; tail -3 $O/bench_$c.err; done
export TMPDIR=/tmp
for c in c3 c4; do
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -o $c -- python $GRAFT_REPO_ROOT/bench.py --config $c --no-variants --no-cpu-baseline --steps 5 --warmup 2 > /tmp/prof_$c.log 2>&1)
f=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1); cp $f $O/${c}_kernel_stats.csv; head -14 $f | cut -c1-90,300-420
done
