#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04w}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
shift
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --steps 4 --warmup 1 --skip-extras "$@" > $O/under_rocprof.json 2> $O/stats.log
find $O/stats -name "*kernel_trace.csv" -delete
