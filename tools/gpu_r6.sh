#!/bin/bash
# round-6 GPU session: A/B of an option on the bench workloads in ONE call (same box).  usage: bash tools/gpu_r6.sh <tag> "<opt A>" "<opt B>" <workload>...
# workloads as in tools/gpu_r5.sh (head rep harsh p2 hic, suffix 3: three lanes); an option is a bench.py --option string such as coop=0x100ff ("" = none)
cd $GRAFT_REPO_ROOT
T=${1:-r06x}; shift
A=$1; shift
B=$1; shift
O=gpurun_out/$T
mkdir -p $O
run() {  # name, args
  local name=$1; shift
  timeout 500 python bench.py --steps ${STEPS:-6} --warmup 2 --skip-extras "$@" > $O/$name.json 2> $O/$name.log
  python - <<PY
import json
try:
    j=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('%-14s' % '$name', j['value'], 'M pairs/s', j['ms_per_step'], 'ms', json.dumps({k[:3]: round(v, 1) for k, v in j['stage_ms_per_step'].items()}))
except Exception as e:
    print('$name', 'failed', e); print(open('$O/$name.log').read()[-1500:])
PY
}
for w in "$@"; do
  L=1; case $w in *3) L=3;; esac
  for v in A B A B; do
    if [ $v = A ]; then OPT=$A; else OPT=$B; fi
    EX=""; if [ -n "$OPT" ]; then EX="--option $OPT"; fi
    case $w in
      head*) run $w.$v --lanes $L $EX;;
      rep*) run $w.$v --lanes $L --headline-repeats 32,600,3000,0.02 $EX;;
      harsh*) run $w.$v --lanes $L --headline-repeats profile:1 $EX;;
      p2*) run $w.$v --lanes $L --headline-repeats profile:2 $EX;;
      hic*) run $w.$v --lanes $L --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000 $EX;;
    esac
  done
done
