#!/bin/bash
# the whole -m gpu suite, then the default bench line (every reference check)
cd $GRAFT_REPO_ROOT
T=${1:-r04full}
O=gpurun_out/$T
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 1800 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.log; echo "bench default rc $?"
python - <<PY
import json
try:
    j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
    print('default', j['value'], j['ms_per_step'], j['config']['lanes'], 'roofline', j['roofline']['frac'], j['roofline'].get('frac_visited'))
    print(' cpu', j['cpu_baseline'])
    for k in ('repeat_workload','harsh_repeat_workload','harsh2_repeat_workload','hic_workload'):
        r=j.get(k) or {}
        cb=r.get('cpu_baseline') or {}
        print(' ', k, r.get('value'), r.get('ms_per_step'), r.get('error'), 'ref:', cb.get('value'), cb.get('bed_identical_to_reference'), cb.get('bed_lines'))
    print(' pcie', j['pcie_inclusive'])
except Exception as e:
    print('default failed', e); print(open('$O/bench_default.log').read()[-3000:])
PY
