#!/bin/bash
# kernel stats of one workload (rocprofv3 --kernel-trace --stats), then the rescue-search statistics
cd $GRAFT_REPO_ROOT
T=${1:-r04d}; W=${2:-profile:1}
O=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --steps 4 --warmup 1 --skip-extras --headline-repeats $W --lanes 1 > $O/under_rocprof.json 2> $O/stats.log
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep -v "k_sy\|k_gather\|k_probe<\|rocprim" {} | head -40' | cut -d, -f1-4 | cut -c1-150
find $O/stats -name "*kernel_trace.csv" -delete
timeout 300 python tools/coop_profile.py $W 2>&1 | tail -14
