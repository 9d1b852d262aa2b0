#!/bin/bash
# GPU call J: the device-pool tests, the latency tool with its per-stage table
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02j}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py -x -q > $O/pool_tests.log 2>&1; rc=$?; echo "pool rc=$rc" | tee -a $O/steps.log; tail -15 $O/pool_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python tools/latency.py > $O/latency.json 2> $O/latency.err; rc=$?; echo "latency rc=$rc" | tee -a $O/steps.log; tail -c 900 $O/latency.json
exit 0
