set -x
cd $GRAFT_REPO_ROOT
T=${1:-r02b}
mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_synthetic.py -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$T/pytest.log
tail -12 gpurun_out/$T/pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 --skip-cpu > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.log; echo "bench rc $?"
tail -3 gpurun_out/$T/bench_default.log
python - <<PY
import json
j=json.loads(open('gpurun_out/$T/bench_default.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['stage_ms_per_step'])
r=j.get('repeat_workload') or {}
print('repeat', r.get('value'), r.get('ms_per_step'), r.get('stage_ms_per_step'), r.get('error'))
print(j['pcie_inclusive'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$T/rep_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --skip-extras --headline-repeats 32,600,3000,0.02 --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_repeats_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/$T/rep_stats.log
cd $GRAFT_REPO_ROOT
find gpurun_out/$T/rep_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {} | cut -c1-200'
find gpurun_out/$T/rep_stats -name "*kernel_trace.csv" -delete
