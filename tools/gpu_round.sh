set -x
cd $GRAFT_REPO_ROOT
T=${1:-r02f}
mkdir -p gpurun_out/$T
timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$T/pytest.log
tail -8 gpurun_out/$T/pytest.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.log; echo "bench rc $?"
tail -3 gpurun_out/$T/bench_default.log
python - <<PY
import json
j=json.loads(open('gpurun_out/$T/bench_default.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'])
r=j.get('repeat_workload') or {}
print('repeat', r.get('value'), r.get('ms_per_step'), r.get('stage_ms_per_step'), r.get('error'), (r.get('cpu_baseline') or {}).get('bed_identical_to_reference'))
print(j['pcie_inclusive']); print(j['cpu_baseline'])
print({k:j['roofline'][k] for k in ('achieved','frac','useful_frac','launch_ms') if k in j['roofline']})
PY
timeout 300 python bench.py --steps 20 --warmup 5 --force-exchange --skip-extras > gpurun_out/$T/bench_exchange.json 2> gpurun_out/$T/bench_exchange.log; echo "bench-ex rc $?"
python -c "
import json
j=json.loads(open('gpurun_out/$T/bench_exchange.json').read().strip().splitlines()[-1]); print('exchange', j['value'], j['ms_per_step'], j.get('exchange'))"
timeout 600 python tools/e2e_bench.py --gz > gpurun_out/$T/e2e_cli.json 2> gpurun_out/$T/e2e_cli.log; cat gpurun_out/$T/e2e_cli.json | cut -c1-1500
bash tools/profile_bench.sh ${T}_prof > gpurun_out/$T/profile.log 2>&1; tail -5 gpurun_out/$T/profile.log
find gpurun_out/${T}_prof -name "*kernel_trace.csv" -delete; find gpurun_out/${T}_prof -name "*counter_collection.csv" -size +20M -delete
