#!/bin/bash
# window sweeps of the generic prover on one box (each gpurun call lands on a different box: only numbers of one call compare)
#   bash tools/gpu/csec_sweep.sh <ENV_NAME> <logn list> <value> [<value> ...]      value "default" = variable unset
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; LOGN=$2; shift 2
for c in "$@"; do
  if [ "$c" = default ]; then unset "$VAR"; else export "$VAR=$c"; fi
  timeout -k 10 300 python tools/generic_bench.py --logn "$LOGN" --no-check --out "gpurun_out/sweep_${VAR}_$c.json" > "gpurun_out/sweep_${VAR}_$c.log" 2>&1 || { tail -5 "gpurun_out/sweep_${VAR}_$c.log"; exit 1; }
  python - "gpurun_out/sweep_${VAR}_$c.json" "$VAR=$c" <<'P'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], [(s['logn'], s['wires'], s['proofs_per_s']) for s in j['sizes']])
P
done
