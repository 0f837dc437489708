#!/bin/bash
# experiment runner: bash tools/gpu_exp.sh <tag> "<ENV=val ...>" <bench args...> -> one line with value + stage times
T=$1; E=$2; shift; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp
env $E timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras "$@" > gpurun_out/exp/$T.json 2> gpurun_out/exp/$T.log
python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/exp/$T.json').read().strip().splitlines()[-1])
    s=j['stage_ms_per_step']
    print('$T', '[$E]', j['value'], 'M pairs/s', j['ms_per_step'], 'ms', ' '.join('%s=%.2f'%(k.split('_')[0],v) for k,v in s.items()))
except Exception as e:
    print('$T failed', e); print(open('gpurun_out/exp/$T.log').read()[-800:])
PY
