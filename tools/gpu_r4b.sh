#!/bin/bash
# round-4 session B: list-length distributions of the two mammalian-like genomes + first throughput of profile:2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
for p in 1 2; do
  timeout 300 python tools/list_hist.py --repeats profile:$p > gpurun_out/r04b/hist_profile$p.json 2> gpurun_out/r04b/hist_profile$p.txt
  tail -30 gpurun_out/r04b/hist_profile$p.txt
done
timeout 300 python bench.py --steps 4 --warmup 1 --skip-extras --lanes 1 --headline-repeats profile:2 > gpurun_out/r04b/bench_p2.json 2> gpurun_out/r04b/bench_p2.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04b/bench_p2.json').read().strip().splitlines()[-1])
print('profile2', j['value'], j['ms_per_step'], j['stage_ms_per_step']); print(j['counters_per_step'])
PY
