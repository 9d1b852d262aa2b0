#!/bin/bash
# GPU call Z: parity tests (prover, engines, pool) with the current build, then A/B against the library named by $2 (a build of the previous code)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02z}; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_prover.py tests/test_gpu_engines.py tests/test_gpu_pool.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log; [ $rc = 0 ] || exit 1
bash tools/gpu/r02_call_u.sh ${1:-r02z}/ab $2
