#!/bin/bash
# chunk size / threads / decode-ahead of the several-thread gunzip inside a CLI run (.gz -> BED of tools/e2e_bench.py's files)
cd $GRAFT_REPO_ROOT
T=${1:-r06_pargz_chunk}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
D=/tmp/chromap_amd_e2e
[ -f $D/r1.fq.gz ] || timeout 600 python tools/e2e_bench.py --gz --reps 1 > $O/e2e.json 2> $O/e2e.log
run() {  # label, env...
  local label=$1; shift
  for i in 1 2; do
    rm -f $D/out_gz.bed
    env "$@" CM_PARGZ_DEBUG=1 chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.gz -2 $D/r2.fq.gz -o $D/out_gz.bed > $O/$label.log 2>&1
    echo "$label: $(grep 'Mapped all' $O/$label.log)"
  done
  grep "so far" $O/$label.log | tail -2
}
cat /sys/fs/cgroup/cpu.max
run default X=1
run blocking CM_BLOCKING_SYNC=1
run default_c1024 CM_PARGZ_CHUNK_KB=1024
run blocking_c1024 CM_PARGZ_CHUNK_KB=1024 CM_BLOCKING_SYNC=1
run budget32 CM_CPU_BUDGET=32
for i in 1 2 3; do rm -f $D/out_b.bed; chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out_b.bed 2>&1 | grep "Mapped all" | sed "s/^/bgz: /"; rm -f $D/out_b.bed; CM_BLOCKING_SYNC=1 chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out_b.bed 2>&1 | grep "Mapped all" | sed "s/^/bgz blocking: /"; done
cat /sys/fs/cgroup/cpu.stat | grep thrott

md5sum $D/out_gz.bed
