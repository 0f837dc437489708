cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q7
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden_modes.py tests/test_gpu_synthetic.py tests/test_gpu_stages.py -x -q -p no:cacheprovider 2>&1 | tail -3
run() { # name, args...
  n=$1; shift
  timeout 400 python bench.py --skip-extras "$@" > gpurun_out/q7/$n.json 2> gpurun_out/q7/$n.log
  python -c "
import json
j=json.loads(open('gpurun_out/q7/$n.json').read().strip().splitlines()[-1]); print('$n', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
}
run hic_l1 --steps 5 --warmup 2 --lanes 1 --preset hic --readlen 150 --hic 0.35 --indel-rate 0.001 --pairs 2000000
run hic_l3 --steps 5 --warmup 2 --lanes 3 --preset hic --readlen 150 --hic 0.35 --indel-rate 0.001 --pairs 2000000
run chip_l3 --steps 5 --warmup 2 --lanes 3 --preset chip --readlen 100 --frag-min 150 --frag-max 700
