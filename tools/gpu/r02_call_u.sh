#!/bin/bash
# GPU call U: A/B on one box of two builds of the library (default against the one named by $2, e.g. zk-franchise-proof-circuit_amd/libzkcensus_prio2.so)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02u}; mkdir -p $O; ALT=$GRAFT_REPO_ROOT/$2
for mode in default variant default variant; do
  if [ $mode = variant ]; then export ZKCENSUS_LIB=$ALT; else unset ZKCENSUS_LIB; fi
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > $O/b_$mode.json 2> $O/b_$mode.err || { echo "bench failed ($mode)"; tail -3 $O/b_$mode.err; exit 1; }
  python - $O/b_$mode.json $mode <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], j['value'], j['ms_per_step'])
P
done
export ZKCENSUS_LIB=$ALT
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b_verify.json 2> $O/b_verify.err; echo "verify run rc=$?"
python - $O/b_verify.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); v=j['verified']; print(j['value'], v['batch_verifier_all_valid'], v['oracle_verifier_all_valid'])
P
