#!/bin/bash
# GPU call E: witness parity first, then the rest of the suite, bench, single-proof latency
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02e2}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_witness.py -x -q > $O/gpu_witness.log 2>&1; rc=$?; echo "witness pytest rc=$rc" | tee -a $O/steps.log; tail -15 $O/gpu_witness.log
if [ $rc != 0 ]; then exit 1; fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/steps.log; tail -5 $O/gpu_tests.log
if [ $rc != 0 ]; then exit 1; fi
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 4 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc $(python3 -c "import json;j=json.load(open('$O/bench.json'));print(j['value'], j['ms_per_step'])")" | tee -a $O/steps.log
timeout -k 10 300 python tools/latency.py > $O/latency.json 2> $O/latency.err; rc=$?; echo "latency rc=$rc" | tee -a $O/steps.log; cat $O/latency.json
exit 0
