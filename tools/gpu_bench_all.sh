# all bench configs on one GPU (+ the 2-rank plumbing check over gloo): JSON lines under gpurun_out/<tag>/      usage: gpu_bench_all.sh <tag>
set -x
O=gpurun_out/$1; mkdir -p $O
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 2500 $O/bench_c2.json
# (every full run ends with its CPU baseline on all host cores: let the box settle before the next config is timed -- C3 measured
# 3.07-3.14 ms right behind such a leg and 2.90-2.98 ms on its own)
for c in c1 c3 c4 c5; do sleep 20; timeout 900 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; python -c "
import json; d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline'], d['kernel_ms_per_step'], d.get('variants'))"; tail -3 $O/bench_$c.err; done
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 > $O/bench_c2_n2gloo.json 2> $O/bench_c2_n2gloo.err; tail -c 600 $O/bench_c2_n2gloo.json; tail -3 $O/bench_c2_n2gloo.err
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c3 --steps 5 --warmup 2 > $O/bench_c3_n2gloo.json 2> $O/bench_c3_n2gloo.err; tail -c 600 $O/bench_c3_n2gloo.json
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c5 --steps 10 --warmup 3 > $O/bench_c5_n2gloo.json 2> $O/bench_c5_n2gloo.err; tail -c 600 $O/bench_c5_n2gloo.json; tail -3 $O/bench_c5_n2gloo.err
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c4 --steps 5 --warmup 2 > $O/bench_c4_n2gloo.json 2> $O/bench_c4_n2gloo.err; tail -c 600 $O/bench_c4_n2gloo.json; tail -3 $O/bench_c4_n2gloo.err
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c3 --exchange full --steps 5 --warmup 2 > $O/bench_c3_full_n2gloo.json 2> $O/bench_c3_full_n2gloo.err; tail -c 600 $O/bench_c3_full_n2gloo.json
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c3 --exchange full-pipelined --steps 5 --warmup 2 > $O/bench_c3_fullpipe_n2gloo.json 2> $O/bench_c3_fullpipe_n2gloo.err; tail -c 600 $O/bench_c3_fullpipe_n2gloo.json
SIGMAN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config c3 --exchange full-pipelined --pipeline-chunks 2 --steps 5 --warmup 2 > $O/bench_c3_fullpipe2_n2gloo.json 2> $O/bench_c3_fullpipe2_n2gloo.err; tail -c 600 $O/bench_c3_fullpipe2_n2gloo.json
