#!/bin/bash
# GPU call M: A/B on one box -- proofs per pipeline pass (ZKC_INFLIGHT)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02m}; mkdir -p $O
for v in 96 128 96 128 114; do
  ZKC_INFLIGHT=$v timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > $O/b_$v.json 2> $O/b_$v.err || { echo "failed at $v"; tail -3 $O/b_$v.err; exit 1; }
  python - $O/b_$v.json $v <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print('inflight', sys.argv[2], j['value'], j['ms_per_step'])
P
done
