#!/bin/bash
# GPU call P: lane-per-product blinding kernels -- parity tests first, then A/B against the one-wave-per-task kernel on the same box
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02p}; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_prover.py tests/test_gpu_pool.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log; [ $rc = 0 ] || exit 1
for mode in new old new old; do
  if [ $mode = old ]; then export ZKC_FINALIZE_WAVES=1; else unset ZKC_FINALIZE_WAVES; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > $O/b_$mode.json 2> $O/b_$mode.err || { echo "bench failed ($mode)"; tail -3 $O/b_$mode.err; exit 1; }
  python - $O/b_$mode.json $mode <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], j['value'], j['ms_per_step'])
P
done
unset ZKC_FINALIZE_WAVES
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_verify.json 2> $O/b_verify.err; echo "verify run rc=$?"
python - $O/b_verify.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); v=j['verified']; print(j['value'], v['batch_verifier_all_valid'], v['oracle_verifier_all_valid'])
P
