cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_stages.py -m gpu -q -x -p no:cacheprovider -k "not four_and_eight" > gpurun_out/r03l/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03l/pytest.log
for cfg in "1 " "3 " "1 --force-exchange" "1 --option speculative_sizes=0"; do
set -- $cfg; l=$1; shift
timeout 400 python bench.py --steps 20 --warmup 5 --skip-extras --lanes $l "$@" > gpurun_out/r03l/h.json 2> gpurun_out/r03l/h.log
python -c "
import json
j=json.loads(open('gpurun_out/r03l/h.json').read().strip().splitlines()[-1]); print('headline lanes $cfg', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
done
for l in 1 3; do
timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras --lanes $l --headline-repeats 32,600,3000,0.02 > gpurun_out/r03l/rep_l$l.json 2> gpurun_out/r03l/rep_l$l.log
python -c "
import json
j=json.loads(open('gpurun_out/r03l/rep_l$l.json').read().strip().splitlines()[-1]); print('rep_l$l', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
done
