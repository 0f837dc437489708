cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
python tools/coop_profile.py 2>&1 | tail -9
for l in 1 3; do
timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras --lanes $l --headline-repeats 32,600,3000,0.02 > gpurun_out/r03f/rep_l$l.json 2> gpurun_out/r03f/rep_l$l.log
python -c "
import json
j=json.loads(open('gpurun_out/r03f/rep_l$l.json').read().strip().splitlines()[-1]); print('rep_l$l', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
done
timeout 400 python bench.py --steps 6 --warmup 2 --skip-extras --lanes 3 --headline-repeats profile:1 > gpurun_out/r03f/harsh_l3.json 2> gpurun_out/r03f/harsh_l3.log
python -c "
import json
j=json.loads(open('gpurun_out/r03f/harsh_l3.json').read().strip().splitlines()[-1]); print('harsh_l3', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
