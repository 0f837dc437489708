#!/bin/bash
# headline workload with / without the record exchange on one GPU (--force-exchange), one / three lanes
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
run() {
  local name=$1; shift
  timeout 500 python bench.py --steps 20 --warmup 3 --skip-extras "$@" > $O/$name.json 2> $O/$name.log
  python - <<PY
import json
try:
    j=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-22s' % '$name', j['value'], 'M pairs/s', j['ms_per_step'], 'ms')
except Exception as e:
    print('$name failed', e); print(open('$O/$name.log').read()[-600:])
PY
}
for rep in 1 2; do
run plain_l3 --lanes 3
run plain_l1 --lanes 1
run exch_l1 --force-exchange --exchange-lanes 1
run exch_l3 --force-exchange --exchange-lanes 3
run exch_l3_overlap --force-exchange --exchange-lanes 3 --option exchange_overlap=1
done
