# copy the evidence of gpu_final.sh from gpurun_out/ into profiles/ (tracked)    usage: collect_profiles.sh <gpu_final tag> <profiles prefix e.g. r02i>
set -e
T=$1; PFX=$2; H=$(git rev-parse --short HEAD)
for c in c2 c3 c4 c5; do P=gpurun_out/prof_$c; [ -f $P/r02_pmc_$c.json ] || continue
python3 - <<PY
import json
d=json.load(open("$P/r02_pmc_$c.json")); d["commit"]="$H"; json.dump(d, open("profiles/r02_pmc_$c.json","w"), indent=1)
PY
cp $P/kernel_stats.csv profiles/r02_${c}_kernel_stats.csv; cp $P/bench_under_rocprof.json profiles/r02_${c}_bench_under_rocprof.json; cp $P/bench_plain.json profiles/r02_${c}_bench.json; cp $P/table.md profiles/r02_${c}_table.md; done
git rm -q --cached --ignore-unmatch profiles/r02?_bench_*.json; rm -f profiles/r02?_bench_*.json
for f in gpurun_out/$T/bench_*.json; do grep '^{"metric"' $f > profiles/${PFX}_$(basename $f); done
ls profiles | wc -l
