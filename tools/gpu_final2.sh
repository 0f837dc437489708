#!/bin/bash
# refresh of the closing session's bench line and headline profile after a late kernel change: bash tools/gpu_final2.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-r03n}
O=gpurun_out/$T
mkdir -p $O
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1800 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; echo "bench default rc $?"
python - <<PY
import json
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('default', j['value'], j['ms_per_step'], 'lanes', j['config']['lanes'])
r=j['roofline']; print(' roofline', r['achieved'], r['frac'], r['launch_ms'])
print(' cpu', {k: j['cpu_baseline'][k] for k in ('value','cores','kind','bed_identical_to_reference','bed_lines')})
for k in ('repeat_workload','harsh_repeat_workload','hic_workload'):
    r=j.get(k) or {}
    cb=r.get('cpu_baseline') or {}
    print(' ', k, r.get('value'), r.get('ms_per_step'), r.get('error'), 'ref:', cb.get('value'), cb.get('bed_identical_to_reference'), cb.get('bed_lines'))
print(' pcie', j['pcie_inclusive']); print(' stages', j['stage_ms_per_step'])
PY
timeout 400 python bench.py --steps 20 --warmup 5 --skip-extras --lanes 1 > $O/head_l1.json 2> $O/head_l1.log
python -c "
import json
j=json.loads(open('$O/head_l1.json').read().strip().splitlines()[-1]); print('headline lanes 1', j['value'], j['ms_per_step'], j['stage_ms_per_step'])"
bash tools/profile_bench.sh ${T}_prof > $O/profile.log 2>&1; tail -3 $O/profile.log
