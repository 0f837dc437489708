#!/bin/bash
# heavy_mid_max (the longest hit list a 16-lane group takes; default 64) on every bench workload, A/B/A/B
cd $GRAFT_REPO_ROOT
T=${1:-r06_mid}
O=gpurun_out/$T; mkdir -p $O
run() {  # name, workload args..., then -- options
  local name=$1; shift
  timeout 500 python bench.py --steps 6 --warmup 2 --skip-extras --lanes 3 "$@" > $O/$name.json 2> $O/$name.log
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-22s' % sys.argv[2], j['value'], 'M pairs/s', j['ms_per_step'], 'ms')
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
HIC="--preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000"
for v in 64 80 96 112 64 96; do run hic_mid$v $HIC --option heavy_mid_max=$v; done
for v in 64 96 64 96; do run head_mid$v --option heavy_mid_max=$v; done
for v in 64 96 64 96; do run rep_mid$v --headline-repeats 32,600,3000,0.02 --option heavy_mid_max=$v; done
for v in 64 96 64 96; do run p1_mid$v --headline-repeats profile:1 --option heavy_mid_max=$v; done
for v in 64 96; do run p2_mid$v --headline-repeats profile:2 --option heavy_mid_max=$v; done
