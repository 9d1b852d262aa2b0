#!/bin/bash
# GPU call O: how the blinding kernel's duration scales with the proofs of a pass (kernel stats of short runs with ZKC_INFLIGHT = 1, 8, 32, 94)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02o}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 8 32 94; do
  ZKC_INFLIGHT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fin_$v -- python3 $R/bench.py --batch $v --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/run_$v.log 2>&1 || { echo "failed at $v"; tail -3 $R/$O/run_$v.log; exit 1; }
  F=$(find /tmp/fin_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v proofs per pass"; grep -E "zkc_finalize|zkc_msm_window29|zkc_msm_final|zkc_witness_chains_wave|zkc_msm_accumulate29<" $F | awk -F, '{printf "%-60s calls %s avg_ns %s\n", substr($1,1,60), $2, $4}'
done
