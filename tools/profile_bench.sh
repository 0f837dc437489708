#!/bin/bash
# Profiles bench.py on the GPU box: kernel-time summary + HBM traffic (PMC) + LDS bank conflicts in separate passes (the PMC
# passes must not be combined with other trace domains).
# Usage (from the repo root on the GPU box): bash tools/profile_bench.sh <tag> [bench.py arguments of the workload, e.g.
#   --headline-repeats 32,600,3000,0.02 --lanes 1]
# GRAFT_HEAD=<commit> in the environment is written into profiles/probe_traffic.json / s3b_traffic.json next to the sha of the kernel
# sources (the GPU box has no .git); bench.py reports roofline.traffic from those files only while both still match
TAG=${1:-prof}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --skip-cpu --repeats '' --harsh '' --harsh2 '' --hic-workload '' --config3-pairs 0 --graded-probe-only --steps 8 --warmup 2 "$@" > $O/bench_under_rocprof.json 2> $O/stats.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 2 --warmup 0 --probe-repeat 2 "$@" > $O/fetch_bench.json 2> $O/fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 2 --warmup 0 --probe-repeat 2 "$@" > /dev/null 2> $O/write.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 2 --warmup 0 --probe-repeat 2 "$@" > /dev/null 2> $O/lds.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/valu -o bench -- python $R/bench.py --skip-extras --graded-probe-only --steps 2 --warmup 0 --probe-repeat 2 "$@" > /dev/null 2> $O/valu.log
rm -f $O/*/bench_kernel_trace.csv.bak
python $R/tools/summarize_profile.py $O $R/gpurun_out/${TAG}_summary
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $R/gpurun_out/${TAG}_summary
