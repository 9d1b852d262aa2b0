#!/bin/bash
# GPU call D: quick parity subset + bench variants (environment knobs)
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02d2}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_prover.py -x -q -k "fullprove or batch_prove or example" > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/steps.log; tail -3 $O/gpu_tests.log
if [ $rc != 0 ]; then exit 1; fi
for v in "96" "128" "112" "64"; do
  ZKC_INFLIGHT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-verify --steps 4 > $O/bench_inf$v.json 2> $O/bench_inf$v.err; rc=$?
  echo "inflight $v rc=$rc $(python3 -c "import json;j=json.load(open('$O/bench_inf$v.json'));print(j['value'], j['ms_per_step'], j['roofline']['valu']['achieved'], j['roofline']['valu']['frac'], j['roofline']['valu'].get('madds_per_proof'))")" | tee -a $O/steps.log
  if [ $rc = 124 ] || [ $rc = 137 ]; then exit 1; fi
done
exit 0
