cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q6
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | tail -3
run() { # name, args...
  n=$1; shift
  timeout 400 python bench.py --skip-extras "$@" > gpurun_out/q6/$n.json 2> gpurun_out/q6/$n.log
  python -c "
import json
j=json.loads(open('gpurun_out/q6/$n.json').read().strip().splitlines()[-1]); print('$n', j['value'], j['ms_per_step'], json.dumps(j['stage_ms_per_step']))"
}
run head_l1 --steps 8 --warmup 2 --lanes 1
run rep_l1 --steps 6 --warmup 2 --lanes 1 --headline-repeats 32,600,3000,0.02
run rep_l3 --steps 6 --warmup 2 --lanes 3 --headline-repeats 32,600,3000,0.02
run hic_l1 --steps 5 --warmup 2 --lanes 1 --preset hic --readlen 150 --hic 0.35 --indel-rate 0.001 --pairs 2000000
run hic_l3 --steps 5 --warmup 2 --lanes 3 --preset hic --readlen 150 --hic 0.35 --indel-rate 0.001 --pairs 2000000
