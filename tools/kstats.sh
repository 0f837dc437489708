#!/bin/bash
# rocprofv3 kernel stats of one bench workload: bash tools/kstats.sh <tag> [bench args]  -> gpurun_out/<tag>_kernel_stats.csv (+ top 30 printed)
TAG=${1:-ks}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 6 --warmup 2 "$@" > $O/bench.json 2> $O/stats.log
cp $O/stats/bench_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:32]:
    print("%-60s %5s calls  avg %9.3f ms  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
tail -1 $O/bench.json | cut -c1-300
