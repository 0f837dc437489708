set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
tail -15 gpurun_out/r02a/pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.log; echo "bench rc $?"
tail -3 gpurun_out/r02a/bench_default.log; head -c 1500 gpurun_out/r02a/bench_default.json
timeout 300 python bench.py --steps 10 --warmup 2 --force-exchange --skip-extras > gpurun_out/r02a/bench_exchange.json 2> gpurun_out/r02a/bench_exchange.log; echo "bench-ex rc $?"
tail -3 gpurun_out/r02a/bench_exchange.log; head -c 600 gpurun_out/r02a/bench_exchange.json
