#!/bin/bash
# round-5 GPU session: optional pytest selection, then the bench workloads one by one (quick: no reference runs).
# usage: bash tools/gpu_r5.sh <tag> "<pytest args or none>" <workload>...   workloads: head rep harsh p2 hic (suffix 3: three lanes)
cd $GRAFT_REPO_ROOT
T=${1:-r05x}; shift
TESTS=${1:-none}; shift
O=gpurun_out/$T
mkdir -p $O
if [ "$TESTS" != "none" ]; then
  eval "timeout 1500 python -m pytest $TESTS -q --maxfail=10 -p no:cacheprovider" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -8 $O/pytest.log
fi
run() {  # name, args
  local name=$1; shift
  timeout 500 python bench.py --steps ${STEPS:-6} --warmup 2 --skip-extras "$@" > $O/$name.json 2> $O/$name.log
  python - <<PY
import json
try:
    j=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    c=j['counters_per_step']
    print('$name', j['value'], 'M pairs/s', j['ms_per_step'], 'ms', json.dumps(j['stage_ms_per_step']))
    print('   cand/read %.2f mapped %d multi %d rescued %d occ %d' % (c['num_candidates']/2.0/j['config']['pairs_per_gpu_per_step'], j['mapped_pairs_per_step'], c['num_multi_mappers'], c['num_pairs_rescued'], c['occurrences_read']))
except Exception as e:
    print('$name', 'failed', e); print(open('$O/$name.log').read()[-1500:])
PY
}
for w in "$@"; do
  L=1; case $w in *3) L=3;; esac
  case $w in
    head*) run $w --lanes $L;;
    rep*) run $w --lanes $L --headline-repeats 32,600,3000,0.02;;
    harsh*) run $w --lanes $L --headline-repeats profile:1;;
    p2*) run $w --lanes $L --headline-repeats profile:2;;
    hic*) run $w --lanes $L --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000;;
  esac
done
