#!/bin/bash
# what the CLI's first batch spends on device memory: CM_ALLOC_TRACE + CM_CLI_TIMES of a BGZF -> BED run (tools/e2e_bench.py's files)
cd $GRAFT_REPO_ROOT
T=${1:-r06_cli_alloc}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
D=/tmp/chromap_amd_e2e
[ -f $D/r1.fq.bgz ] || timeout 600 python tools/e2e_bench.py --gz --reps 1 > $O/e2e.json 2> $O/e2e.log
for i in 1 2; do
  CM_ALLOC_TRACE=1 CM_CLI_TIMES=1 chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out.bed 2>&1 | grep -v "^Mapped [0-9]* read" > $O/run$i.log
done
grep -c alloc $O/run2.log
awk '/alloc/ {s += $(NF-1)} END {print "total ms in allocations:", s}' $O/run2.log
grep -v "^Number\|^Loaded\|^Kmer" $O/run2.log | awk '!/alloc/ || $(NF-1) > 0.3'
/tmp/none 2>/dev/null; g++ -O2 -pthread -o /tmp/write_probe tools/probes/write_probe.cpp && /tmp/write_probe $D/wp.bin
