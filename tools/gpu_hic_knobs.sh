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
run wave256 heavy_wave_max=256
run wave384 heavy_wave_max=384
run cap12 s3b_lane_cap=12
run cap24 s3b_lane_cap=24
run last1 heavy_last=1
run lastm1 heavy_last=-1
run default2
run lpl1 probe_lookups_per_lane=1
run lpl4 probe_lookups_per_lane=4
run tile16 prep_tile_reads=16
run prefetch0 probe_pair_prefetch=0
run prefetch1 probe_pair_prefetch=1
