cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_exchange.py -x -q -p no:cacheprovider -k "pipelined" 2>&1 | tail -3
timeout 600 python tools/pcie_bench.py --settings "lanes=1" "lanes=3" 2>&1 | tail -4
