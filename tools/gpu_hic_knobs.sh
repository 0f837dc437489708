#!/bin/bash
# host-side class thresholds on the hic workload (2 x 150, ~37 hits per read): which lists a lane, a 16-lane group, a wave take
cd $GRAFT_REPO_ROOT
T=${1:-r06_hic_knobs}
O=gpurun_out/$T; mkdir -p $O
run() {  # name, options...
  local name=$1; shift
  local EX=""; for o in "$@"; do EX="$EX --option $o"; done
  timeout 500 python bench.py --steps 6 --warmup 2 --skip-extras --lanes 3 --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000 $EX > $O/$name.json 2> $O/$name.log
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-22s' % sys.argv[2], j['value'], 'M pairs/s', j['ms_per_step'], 'ms', json.dumps({k[:3]: round(v, 1) for k, v in j['stage_ms_per_step'].items()}))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run default
run cap24 s3b_lane_cap=24
run cap32 s3b_lane_cap=32
run cap48 s3b_lane_cap=48
run mid32 heavy_mid_max=32
run mid96 heavy_mid_max=96
run mid128 heavy_mid_max=128
run nomid heavy_mid_max=-1
run nomid_cap48 heavy_mid_max=-1 s3b_lane_cap=48
run default2
run chunks8 mm_chunks=8
run chunks2 mm_chunks=2
