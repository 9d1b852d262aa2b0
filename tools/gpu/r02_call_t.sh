#!/bin/bash
# GPU call T: A/B on one box of an environment switch given as $2 (set = variant, unset = default)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02t}; mkdir -p $O; VAR=$2
for mode in default variant default variant; do
  if [ $mode = variant ]; then export $VAR=1; else unset $VAR; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > $O/b_$mode.json 2> $O/b_$mode.err || { echo "bench failed ($mode)"; tail -3 $O/b_$mode.err; exit 1; }
  python - $O/b_$mode.json $mode <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], j['value'], j['ms_per_step'])
P
done
export $VAR=1
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_verify.json 2> $O/b_verify.err; echo "verify run rc=$?"
python - $O/b_verify.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); v=j['verified']; print(j['value'], v['batch_verifier_all_valid'], v['oracle_verifier_all_valid'])
P
