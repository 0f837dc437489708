#!/bin/bash
# kernel-time summary of one workload under rocprofv3: bash tools/gpu_stats3.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=$1; shift
O=$R/gpurun_out/$T
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o rep -- python $R/bench.py --steps 4 --warmup 1 --skip-extras --lanes 1 "$@" > $O/under_rocprof.json 2> $O/stats.log
cd $R
python - <<PY
import csv,glob
f=glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[1:]:
    n=r[0]
    if any(x in n for x in ('k_sy','k_gather','k_probe<','rocprim','k_rehash')): continue
    a=float(r[3])/1e3; c=int(r[1])
    if a*c/6 > 200: print('%-70s calls %4s avg %9.1f us' % (n[:70], r[1], a))
PY
find $O/stats -name "*kernel_trace.csv" -delete
