# copy the evidence of gpu_final.sh from gpurun_out/ into profiles/ (tracked)    usage: collect_profiles.sh <gpu_final tag> <round prefix e.g. r03> [bench prefix e.g. r03a]
set -e
T=$1; RND=$2; PFX=${3:-$2}; H=$(git rev-parse --short HEAD)
for c in c1 c2 c3 c4 c5; do P=gpurun_out/prof_$c; [ -f $P/pmc_$c.json ] || continue
python3 - <<PY
import json
d=json.load(open("$P/pmc_$c.json")); d["commit"]="$H"; json.dump(d, open("profiles/${RND}_pmc_$c.json","w"), indent=1)
PY
cp $P/kernel_stats.csv profiles/${RND}_${c}_kernel_stats.csv; cp $P/bench_under_rocprof.json profiles/${RND}_${c}_bench_under_rocprof.json; cp $P/bench_plain.json profiles/${RND}_${c}_bench.json; cp $P/table.md profiles/${RND}_${c}_table.md; done
for f in gpurun_out/$T/bench_*.json; do grep '^{"metric"' $f > profiles/${PFX}_$(basename $f) || true; done
ls profiles | wc -l
