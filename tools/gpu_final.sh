#!/bin/bash
# the round's closing GPU session: the whole -m gpu suite (slow tests included), smoke, the default bench line with every
# reference check, the exchange path on one rank, profiles (kernel stats + PMC) of the headline and of the repeat workload
cd $GRAFT_REPO_ROOT
T=${1:-r03z}
O=gpurun_out/$T
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --durations=12 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -22 $O/pytest.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout 1800 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; echo "bench default rc $?"
python - <<PY
import json
try:
    j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
    print('default', j['value'], j['ms_per_step'], 'lanes', j['config']['lanes'])
    r=j['roofline']; print(' roofline', r['achieved'], r['frac'], r['launch_ms'], r.get('probe_on_file_layout'))
    print(' cpu', {k: j['cpu_baseline'][k] for k in ('value','cores','kind','bed_identical_to_reference','bed_lines')})
    for k in ('repeat_workload','harsh_repeat_workload','hic_workload'):
        r=j.get(k) or {}
        cb=r.get('cpu_baseline') or {}
        print(' ', k, r.get('value'), r.get('ms_per_step'), r.get('error'), 'ref:', cb.get('value'), cb.get('bed_identical_to_reference'), cb.get('bed_lines'))
        print('     ', r.get('stage_ms_per_step'))
    print(' pcie', j['pcie_inclusive']); print(' post', j['postprocess_on_device'])
    print(' stages', j['stage_ms_per_step'])
except Exception as e:
    print('default failed', e); print(open('$O/bench_default.log').read()[-3000:])
PY
for l in 1 3; do
timeout 400 python bench.py --steps 20 --warmup 5 --skip-extras --lanes $l > $O/head_l$l.json 2> $O/head_l$l.log
python -c "
import json
j=json.loads(open('$O/head_l$l.json').read().strip().splitlines()[-1]); print('headline lanes $l', j['value'], j['ms_per_step'], j['stage_ms_per_step'])"
done
timeout 400 python bench.py --steps 20 --warmup 5 --force-exchange --skip-extras > $O/bench_exchange.json 2> $O/bench_exchange.log
python -c "
import json
j=json.loads(open('$O/bench_exchange.json').read().strip().splitlines()[-1]); print('exchange', j['value'], j['ms_per_step'], j.get('exchange'))"
bash tools/profile_bench.sh ${T}_prof > $O/profile.log 2>&1; tail -4 $O/profile.log
bash tools/profile_bench.sh ${T}_prof_repeat --headline-repeats 32,600,3000,0.02 --lanes 1 > $O/profile_repeat.log 2>&1; tail -4 $O/profile_repeat.log
timeout 600 python tools/e2e_bench.py --gz > $O/e2e_cli.json 2> $O/e2e_cli.log; cut -c1-800 $O/e2e_cli.json
