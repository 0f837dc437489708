#!/bin/bash
# the CLI of this tree against the one built from HEAD into _base/ on a 32 M-pair job (same box, runs interleaved): text and BGZF -> BED
cd $GRAFT_REPO_ROOT
T=${1:-r06_cli_ab32}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
D=/tmp/chromap_amd_e2e32
[ -f $D/r1.fq.bgz ] || timeout 900 python tools/e2e_bench.py --gz --reps 1 --pairs 32000000 --skip-host-ingest --dir $D > $O/e2e.json 2> $O/e2e.log
run() {  # label, binary, suffix, env...
  local label=$1 bin=$2 x=$3; shift 3
  rm -f $D/out_$label.bed; sync
  echo "$label $x $(env "$@" $bin --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq$x -2 $D/r2.fq$x -o $D/out_$label.bed 2>&1 | grep 'Mapped all')"
}
for i in 1 2 3 4; do
  run new chromap_amd/chromap-amd .bgz X=1
  run hi chromap_amd/chromap-amd .bgz CM_FQ_PRIO=1
  run lo chromap_amd/chromap-amd .bgz CM_FQ_PRIO=-1
done
md5sum $D/out_*.bed
