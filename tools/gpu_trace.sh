#!/bin/bash
# a kernel timeline of one workload with one lane (where does a step wait?): rocprofv3 --kernel-trace, csv
cd $GRAFT_REPO_ROOT
T=${1:-r04t}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --skip-extras --lanes 1 "$@" > $O/bench.json 2> $O/bench.log
ls -la $O/trace | head
python - <<PY
import csv, glob
f = glob.glob('$O/trace/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), 'kernel records')
PY
