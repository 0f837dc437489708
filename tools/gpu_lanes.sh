#!/bin/bash
# lanes sweep on one workload: bash tools/gpu_lanes.sh <tag> <workload args...> -- lanes...
cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for L in "$@"; do
  timeout 500 python bench.py --steps 6 --warmup 2 --skip-extras --lanes $L "${ARGS[@]}" > $O/l$L.json 2> $O/l$L.log
  python - <<PY
import json
try:
    j=json.loads(open('$O/l$L.json').read().strip().splitlines()[-1])
    print('lanes $L', j['value'], 'M pairs/s', j['ms_per_step'], 'ms', json.dumps({k[:3]: round(v, 1) for k, v in j['stage_ms_per_step'].items()}))
except Exception as e:
    print('lanes $L failed', e); print(open('$O/l$L.log').read()[-800:])
PY
done
