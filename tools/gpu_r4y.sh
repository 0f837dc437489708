#!/bin/bash
# round-4 GPU session: end-to-end CLI rates only (plain, .gz, .bgz; each three times)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04y}
mkdir -p $O
timeout 1200 python tools/e2e_bench.py --gz > $O/e2e.json 2> $O/e2e.log; echo "e2e rc $?"
tail -5 $O/e2e.log; cat $O/e2e.json
