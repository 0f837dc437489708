# quick measurement runs: bench.py --skip-extras with option sets given as arguments ("a=1,b=2" each)
cd $GRAFT_REPO_ROOT
T=${TAG:-quick}
mkdir -p gpurun_out/$T
i=0
for opts in "$@"; do
  i=$((i+1))
  args=""
  for o in $(echo $opts | tr ',' ' '); do case $o in --*) args="$args $o";; *=*) args="$args --option $o";; *) args="$args $o";; esac; done
  timeout 300 python bench.py --steps 10 --warmup 2 --skip-extras $args > gpurun_out/$T/b$i.json 2> gpurun_out/$T/b$i.log
  python - <<PY
import json
try:
    j=json.loads(open('gpurun_out/$T/b$i.json').read().strip().splitlines()[-1])
    print('$opts', j['value'], j['ms_per_step'], j['stage_ms_per_step'])
except Exception as e:
    print('$opts', 'failed', e); print(open('gpurun_out/$T/b$i.log').read()[-800:])
PY
done
