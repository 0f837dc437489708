# quick A/B after a kernel change: parity subset, then the headline and the repeat workloads, one and three lanes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synthetic.py tests/test_gpu_golden_modes.py -x -q -p no:cacheprovider 2>&1 | tail -4
for l in 1 3; do
timeout 300 python bench.py --steps 8 --warmup 2 --skip-extras --lanes $l > gpurun_out/q5/head_l$l.json 2> gpurun_out/q5/head_l$l.log
python -c "
import json
j=json.loads(open('gpurun_out/q5/head_l$l.json').read().strip().splitlines()[-1]); print('head_l$l', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras --lanes $l --headline-repeats 32,600,3000,0.02 > gpurun_out/q5/rep_l$l.json 2> gpurun_out/q5/rep_l$l.log
python -c "
import json
j=json.loads(open('gpurun_out/q5/rep_l$l.json').read().strip().splitlines()[-1]); print('rep_l$l', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
done
timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras --lanes 3 --headline-repeats profile:1 > gpurun_out/q5/harsh_l3.json 2> gpurun_out/q5/harsh_l3.log
python -c "
import json
j=json.loads(open('gpurun_out/q5/harsh_l3.json').read().strip().splitlines()[-1]); print('harsh_l3', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
