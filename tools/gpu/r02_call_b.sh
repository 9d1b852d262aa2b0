#!/bin/bash
# GPU call B: prover/engine parity tests, bench, kernel-trace stats
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02b}; mkdir -p $O
stop_if_killed() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "step $2 was killed (rc=$1): stopping" | tee -a $O/steps.log; exit 1; fi; }
timeout -k 10 900 python -m pytest tests/test_gpu_prover.py tests/test_gpu_engines.py -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/steps.log; tail -25 $O/gpu_tests.log; stop_if_killed $rc pytest
if [ $rc != 0 ]; then exit 1; fi
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 4 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc" | tee -a $O/steps.log; tail -c 900 $O/bench.json; tail -5 $O/bench.err; stop_if_killed $rc bench
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; rc=$?; echo "prof rc=$rc" | tee -a $GRAFT_REPO_ROOT/$O/steps.log
cd $GRAFT_REPO_ROOT/$O/prof && find . -name "*kernel_trace.csv" -size +20M -delete; find . -name "*_kernel_stats.csv" | head
exit 0
