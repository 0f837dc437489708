#!/bin/bash
# round-3 GPU session: the -m gpu suite, the repeat-bearing workload with and without the cooperative kernels, kernel stats.
# usage (on the GPU box, from the repo root): bash tools/gpu_r3.sh <tag> [quick]
cd $GRAFT_REPO_ROOT
T=${1:-r03a}
MODE=${2:-full}
O=gpurun_out/$T
mkdir -p $O
if [ "$MODE" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -15 $O/pytest.log
fi
REP="32,600,3000,0.02"
run() {  # name, extra args
  local name=$1; shift
  timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras --headline-repeats $REP "$@" > $O/$name.json 2> $O/$name.log
  python - <<PY
import json
try:
    j=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', j['value'], 'M pairs/s', j['ms_per_step'], 'ms', json.dumps(j['stage_ms_per_step']))
except Exception as e:
    print('$name', 'failed', e); print(open('$O/$name.log').read()[-1500:])
PY
}
run rep_l1_coop0 --lanes 1 --option coop=0
run rep_l1_coop --lanes 1
run rep_l3_coop --lanes 3
run rep_l1_coop1 --lanes 1 --option coop=1
timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $O/head.json 2> $O/head.log
python -c "
import json
j=json.loads(open('$O/head.json').read().strip().splitlines()[-1]); print('headline', j['value'], j['ms_per_step'], j['stage_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o rep -- python $R/bench.py --steps 4 --warmup 1 --skip-extras --headline-repeats $REP --lanes 1 > $R/$O/rep_under_rocprof.json 2> $R/$O/stats.log
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}' | cut -c1-200
find $O/stats -name "*kernel_trace.csv" -delete
