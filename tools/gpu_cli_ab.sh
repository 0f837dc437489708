#!/bin/bash
# the CLI of this tree against the one built from HEAD into _base/ (same box, runs interleaved): BGZF -> BED and .gz -> BED of tools/e2e_bench.py's files
cd $GRAFT_REPO_ROOT
T=${1:-r06_cli_ab}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
D=/tmp/chromap_amd_e2e
[ -f $D/r1.fq.bgz ] || timeout 600 python tools/e2e_bench.py --gz --reps 1 > $O/e2e.json 2> $O/e2e.log
run() {  # label, binary, inputs, env...
  local label=$1 bin=$2 x=$3; shift 3
  rm -f $D/out_$label.bed
  env "$@" CM_CLI_TIMES=1 $bin --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.$x -2 $D/r2.fq.$x -o $D/out_$label.bed 2>&1 | grep "times\|Mapped all\|Sorted" > $O/$label.log
  echo "$label $x $(grep 'Mapped all' $O/$label.log)"
}
for i in 1 2 3; do
  run base $GRAFT_REPO_ROOT/_base/chromap_amd/chromap-amd bgz X=1
  run new chromap_amd/chromap-amd bgz X=1
  run div1 chromap_amd/chromap-amd bgz CM_FIRST_PIECE_DIV=1
  run div4 chromap_amd/chromap-amd bgz CM_FIRST_PIECE_DIV=4
done
cat $O/div1.log; cat $O/new.log; cat $O/div4.log
for i in 1 2; do
  run basegz $GRAFT_REPO_ROOT/_base/chromap_amd/chromap-amd gz X=1
  run newgz chromap_amd/chromap-amd gz X=1
done
cat $O/newgz.log | tail -4
for i in 1 2 3; do /usr/bin/time -f "base wall %e s" $GRAFT_REPO_ROOT/_base/chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out_w.bed 2>&1 | grep wall; /usr/bin/time -f "new wall %e s" chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out_w.bed 2>&1 | grep wall; done
md5sum $D/out_*.bed
