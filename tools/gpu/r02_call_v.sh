#!/bin/bash
# GPU call V: accumulation at two waves per SIMD (variant build) with one and two pipeline lanes, against the default, on one box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02v}; mkdir -p $O; ALT=$GRAFT_REPO_ROOT/zk-franchise-proof-circuit_amd/libzkcensus_acc2w.so
run() { # name lib lanes
  if [ -n "$2" ]; then export ZKCENSUS_LIB=$2; else unset ZKCENSUS_LIB; fi
  if [ -n "$3" ]; then export ZKC_LANES=$3; else unset ZKC_LANES; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > $O/b_$1.json 2> $O/b_$1.err || { echo "bench failed ($1)"; tail -3 $O/b_$1.err; exit 1; }
  python - $O/b_$1.json $1 <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], j['value'], j['ms_per_step'])
P
}
run default "" ""
run acc2w_lanes1 $ALT ""
run acc2w_lanes2 $ALT 2
run default_b "" ""
run acc2w_lanes2_b $ALT 2
