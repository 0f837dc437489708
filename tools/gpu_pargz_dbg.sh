#!/bin/bash
# where the several-thread gunzip's time goes inside a CLI run (.gz -> BED of tools/e2e_bench.py's files): CM_PARGZ_DEBUG timers, thread counts
cd $GRAFT_REPO_ROOT
T=${1:-r06_pargz_dbg}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
D=/tmp/chromap_amd_e2e
[ -f $D/r1.fq.gz ] || timeout 600 python tools/e2e_bench.py --gz --reps 1 > $O/e2e.json 2> $O/e2e.log
echo "nproc $(nproc) THP $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag $(cat /sys/kernel/mm/transparent_hugepage/defrag)"
for NT in default 16; do
  if [ $NT = default ]; then unset CM_PARGZ_THREADS; else export CM_PARGZ_THREADS=$NT; fi
  for i in 1 2; do
    rm -f $D/out_gz.bed
    CM_PARGZ_DEBUG=1 CM_CLI_TIMES=1 chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.gz -2 $D/r2.fq.gz -o $D/out_gz.bed > $O/gz_$NT.log 2>&1
    echo "threads $NT: $(grep 'Mapped all' $O/gz_$NT.log)"
  done
  grep "so far" $O/gz_$NT.log | tail -2
done
unset CM_PARGZ_THREADS
for i in 1 2; do
  rm -f $D/out_gz.bed; CM_PARGZ_NO_AHEAD=1 chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.gz -2 $D/r2.fq.gz -o $D/out_gz.bed 2>&1 | grep "Mapped all" | sed "s/^/no decode ahead: /"
  rm -f $D/out_gz.bed; $GRAFT_REPO_ROOT/_base/chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.gz -2 $D/r2.fq.gz -o $D/out_gz.bed 2>&1 | grep "Mapped all" | sed "s/^/base: /"
done
md5sum $D/out_gz.bed
for NT in; do
  /usr/bin/env time -f "inflate-only one file, $NT threads: %e s" env CM_PARGZ_THREADS=$NT chromap_amd/chromap-amd --inflate-only $D/r1.fq.gz > /dev/null 2> $O/io_$NT.log; tail -2 $O/io_$NT.log
done
