#!/bin/bash
# round-4 GPU session: the device BGZF inflate -- ingest + CLI tests, the inflate kernels' times, the end-to-end CLI rates (plain, .gz, .bgz)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04z}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ingest.py ${PYTEST_MORE} -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
export TMPDIR=/tmp
for lv in 1 6; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_l$lv -o inflate -- python $GRAFT_REPO_ROOT/tools/inflate_bench.py --level $lv > $GRAFT_REPO_ROOT/$O/inflate_l$lv.json 2> $GRAFT_REPO_ROOT/$O/inflate_l$lv.log )
  cat $O/inflate_l$lv.json; tail -3 $O/inflate_l$lv.log
  f=$(find $O/prof_l$lv -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f"
done
if [ -n "$E2E" ]; then
  timeout 900 python tools/e2e_bench.py --gz > $O/e2e.json 2> $O/e2e.log; echo "e2e rc $?"
  tail -5 $O/e2e.log; cat $O/e2e.json
fi
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete
