#!/bin/bash
# debugging aid (GPU box): run the reference binary and chromap-amd with the same flags on gen_synth data, keep a diff.
# usage: bash tools/cmp_ref.sh <tag> "<gen_synth args>" "<mapping flags>"
cd $GRAFT_REPO_ROOT
T=$1; GENARGS=$2; FLAGS=$3
O=gpurun_out/$T; mkdir -p $O
D=/tmp/cmp_$T; mkdir -p $D
python tools/gen_synth.py --out $D/d $GENARGS
./chromap_amd/chromap-amd -i -r $D/d.fa -o $D/d.idx 2> $O/index.log
./oracle/_ref/chromap $FLAGS -x $D/d.idx -r $D/d.fa -1 $D/d_1.fq -2 $D/d_2.fq -o $D/ref.out -t 32 2> $O/ref.log
./chromap_amd/chromap-amd $FLAGS -x $D/d.idx -r $D/d.fa -1 $D/d_1.fq -2 $D/d_2.fq -o $D/gpu.out 2> $O/gpu.log
wc -l $D/ref.out $D/gpu.out > $O/summary.txt
diff $D/ref.out $D/gpu.out | head -60 >> $O/summary.txt
tail -12 $O/ref.log >> $O/summary.txt; tail -12 $O/gpu.log >> $O/summary.txt
cat $O/summary.txt
gzip -c $D/ref.out > $O/ref.out.gz; gzip -c $D/gpu.out > $O/gpu.out.gz
