#!/bin/bash
# kernel stats + PMC passes (FETCH_SIZE / WRITE_SIZE / LDS conflicts / VALU) of the five bench workloads at one commit
cd $GRAFT_REPO_ROOT
T=${1:-r04q}
bash tools/profile_bench.sh ${T}_headline > gpurun_out/${T}_headline.log 2>&1
bash tools/profile_bench.sh ${T}_repeat --headline-repeats 32,600,3000,0.02 > gpurun_out/${T}_repeat.log 2>&1
bash tools/profile_bench.sh ${T}_harsh --headline-repeats profile:1 > gpurun_out/${T}_harsh.log 2>&1
bash tools/profile_bench.sh ${T}_harsh2 --headline-repeats profile:2 > gpurun_out/${T}_harsh2.log 2>&1
bash tools/profile_bench.sh ${T}_hic --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000 > gpurun_out/${T}_hic.log 2>&1
for w in headline repeat harsh harsh2 hic; do echo "== $w"; head -8 gpurun_out/${T}_${w}_summary/kernel_stats.csv | cut -d, -f1-4 | cut -c1-110; done
