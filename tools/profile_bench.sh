#!/bin/bash
# Profiles the default bench on the GPU box: kernel-time summary + HBM traffic (PMC) in
# separate passes (the PMC passes must not be combined with other trace domains).
# Usage (from the repo root on the GPU box): bash tools/profile_bench.sh <tag>
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --skip-cpu --repeats '' --steps 8 --warmup 2 > $O/bench_under_rocprof.json 2> $O/stats.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python $R/bench.py --skip-extras --steps 2 --warmup 0 --probe-repeat 2 > /dev/null 2> $O/fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python $R/bench.py --skip-extras --steps 2 --warmup 0 --probe-repeat 2 > /dev/null 2> $O/write.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds -o bench -- python $R/bench.py --skip-extras --steps 2 --warmup 0 --probe-repeat 2 > /dev/null 2> $O/lds.log
rm -f $O/*/bench_kernel_trace.csv.bak
python $R/tools/summarize_profile.py $O $R/gpurun_out/${TAG}_summary
ls -la $R/gpurun_out/${TAG}_summary
