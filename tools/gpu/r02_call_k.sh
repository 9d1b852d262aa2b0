#!/bin/bash
# GPU call K: shader clock and package power while the bench runs (rocm-smi samples beside a 12-step bench) -- evidence for "power-limited"
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02k}; mkdir -p $O
rocm-smi --showclocks --showpower --showuse --json > $O/smi_idle.json 2> $O/smi_idle.err; echo "idle sample rc=$?"
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showuse --json 2>/dev/null; echo; sleep 0.4; done ) > $O/smi_samples.jsonl &
SMI=$!
timeout -k 10 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-verify > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
head -c 200 $O/bench.json; echo; wc -l $O/smi_samples.jsonl; head -c 600 $O/smi_idle.json
exit 0
