#!/bin/bash
# GPU call Y: parity tests of the prover with the current build, then A/B of an environment switch ($2 set = variant) on the same box
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02y}; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_prover.py tests/test_gpu_pool.py tests/test_gpu_api.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log; [ $rc = 0 ] || exit 1
bash tools/gpu/r02_call_t.sh ${1:-r02y}/ab $2
