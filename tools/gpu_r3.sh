#!/bin/bash
# round-3 GPU session: the -m gpu suite, the side workloads one by one (quick, no reference runs), the default bench line
# with every reference check, kernel stats of the repeat workload.
# usage (on the GPU box, from the repo root): bash tools/gpu_r3.sh <tag> [tests|notests] [default|nodefault]
cd $GRAFT_REPO_ROOT
T=${1:-r03a}
TESTS=${2:-tests}
DEF=${3:-default}
O=gpurun_out/$T
mkdir -p $O
if [ "$TESTS" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
  tail -6 $O/pytest.log
fi
REP="32,600,3000,0.02"
run() {  # name, args
  local name=$1; shift
  timeout 500 python bench.py --steps 6 --warmup 2 --skip-extras "$@" > $O/$name.json 2> $O/$name.log
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
run rep_l1 --lanes 1 --headline-repeats $REP
run rep_l3 --lanes 3 --headline-repeats $REP
run harsh_l1 --lanes 1 --headline-repeats profile:1
run harsh_l3 --lanes 3 --headline-repeats profile:1
run hic_l3 --lanes 3 --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000
run hic_l1 --lanes 1 --preset hic --readlen 150 --indel-rate 0.001 --hic 0.35 --pairs 2000000
roof() {  # name, args: the headline quickly, roofline block printed
  local name=$1; shift
  timeout 500 python bench.py --steps 10 --warmup 3 --skip-extras "$@" > $O/$name.json 2> $O/$name.log
  python - <<PY
import json
try:
    j=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r=j['roofline']
    print('$name', j['value'], 'M pairs/s', j['ms_per_step'], 'ms | probe', r['launch_ms'], 'ms achieved', r['achieved'], 'frac', r['frac'], 'useful', r.get('useful_frac'))
    print('   graded launch', r['probe_only']); print('   file layout  ', r.get('probe_on_file_layout'))
    print('   stages', json.dumps(j['stage_ms_per_step']))
except Exception as e:
    print('$name', 'failed', e); print(open('$O/$name.log').read()[-1500:])
PY
}
roof head_shift0 --probe-table-shift 0
roof head_shift1 --probe-table-shift 1
roof head_shift2 --probe-table-shift 2
if [ "$DEF" = "default" ]; then
  timeout 1500 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.log; echo "bench default rc $?"
  python - <<PY
import json
try:
    j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
    print('default', j['value'], j['ms_per_step'], j['config']['lanes'])
    print(' cpu', j['cpu_baseline'])
    for k in ('repeat_workload','harsh_repeat_workload','hic_workload'):
        r=j.get(k) or {}
        cb=r.get('cpu_baseline') or {}
        print(' ', k, r.get('value'), r.get('ms_per_step'), r.get('error'), 'ref:', cb.get('value'), cb.get('bed_identical_to_reference'), cb.get('bed_lines'))
    print(' pcie', j['pcie_inclusive'])
except Exception as e:
    print('default failed', e); print(open('$O/bench_default.log').read()[-3000:])
PY
fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o rep -- python $R/bench.py --steps 4 --warmup 1 --skip-extras --headline-repeats $REP --lanes 1 > $R/$O/rep_under_rocprof.json 2> $R/$O/stats.log
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'grep -v "k_sy\|k_gather\|k_probe<\|rocprim" {} | head -32' | cut -d, -f1-4 | cut -c1-150
find $O/stats -name "*kernel_trace.csv" -delete
