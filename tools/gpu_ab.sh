# A/B of one cmgpu_set_option on the headline workload, alternating runs on the same box: bash tools/gpu_ab.sh name=value [rounds]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
OPT=$1; N=${2:-3}
for i in $(seq 1 $N); do
for v in base alt; do
  if [ $v = alt ]; then X="--option $OPT"; else X=""; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --skip-extras --lanes 3 $X > gpurun_out/ab/$v$i.json 2> gpurun_out/ab/$v$i.log
  python -c "
import json
j=json.loads(open('gpurun_out/ab/$v$i.json').read().strip().splitlines()[-1]); print('$v', '$X', j['value'], j['ms_per_step'])"
done
done
