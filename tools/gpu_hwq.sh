#!/bin/bash
# do the streams of one process share hardware queues?  The same runs with GPU_MAX_HW_QUEUES unset / 8 / 16 (ROCm maps HIP streams onto
# 4 hardware queues by default; kernels of two streams on one queue run one behind the other).  usage: bash tools/gpu_hwq.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-r06_hwq}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
D=/tmp/chromap_amd_e2e
[ -f $D/r1.fq.bgz ] || timeout 600 python tools/e2e_bench.py --gz --reps 1 > $O/e2e.json 2> $O/e2e.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j['value'], 'M pairs/s', j['ms_per_step'], 'ms')
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
for Q in default 8 16; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  echo "== GPU_MAX_HW_QUEUES=$Q"
  for i in 1 2 3; do
    CM_CLI_TIMES=1 chromap_amd/chromap-amd --preset atac -x $D/g.index -r $D/g.fa -1 $D/r1.fq.bgz -2 $D/r2.fq.bgz -o $D/out.bed 2>&1 | grep "times\|Mapped all\|Sorted" > $O/cli_$Q.$i.log
    grep "Mapped all" $O/cli_$Q.$i.log
  done
  grep times $O/cli_$Q.3.log
  for L in 1 3; do
    timeout 500 python bench.py --steps 6 --warmup 2 --skip-extras --lanes $L > $O/head_q$Q.l$L.json 2> $O/head_q$Q.l$L.log; show $O/head_q$Q.l$L.json
    timeout 500 python bench.py --steps 6 --warmup 2 --skip-extras --lanes $L --headline-repeats profile:2 > $O/p2_q$Q.l$L.json 2> $O/p2_q$Q.l$L.log; show $O/p2_q$Q.l$L.json
  done
done
