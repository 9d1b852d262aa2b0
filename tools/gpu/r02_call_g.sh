#!/bin/bash
# GPU call G: bench with environment variants given as arguments "NAME=VALUE" (one bench each), after a short parity check
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_prover.py -x -q -k "fullprove or batch_prove or example or each_msm" > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/steps.log; tail -3 $O/gpu_tests.log
if [ $rc != 0 ]; then exit 1; fi
for v in "$@"; do
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify --steps 4 > $O/bench_$v.json 2> $O/bench_$v.err; rc=$?
  echo "$v rc=$rc $(python3 -c "import json;j=json.load(open('$O/bench_$v.json'));print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['valu']['frac'])")" | tee -a $O/steps.log
  if [ $rc = 124 ] || [ $rc = 137 ]; then exit 1; fi
done
exit 0
