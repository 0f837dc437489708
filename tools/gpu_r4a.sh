#!/bin/bash
# round-4 session A: where the harsh (mammalian-like) and hic workloads spend their time at the round's starting commit --
# kernel stats + PMC passes for profile:1, kernel stats for hic.
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r04a_harsh --headline-repeats profile:1 --lanes 1 > gpurun_out/r04a_harsh.log 2>&1
bash tools/profile_bench.sh r04a_hic --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000 --lanes 1 > gpurun_out/r04a_hic.log 2>&1
head -40 gpurun_out/r04a_harsh_summary/kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
head -30 gpurun_out/r04a_hic_summary/kernel_stats.csv | cut -d, -f1-5 | cut -c1-160
