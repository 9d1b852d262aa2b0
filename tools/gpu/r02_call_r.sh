#!/bin/bash
# GPU call R: kernel stats + step edges with the split reduction on and off (same box)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02r}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in new old; do
  if [ $mode = old ]; then export ZKC_REDUCE_INLINE=1; else unset ZKC_REDUCE_INLINE; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$mode -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/prof_$mode.log 2>&1; echo "prof $mode rc=$?"
  F=$(find /tmp/kt_$mode -name "*kernel_stats.csv" | head -1); cp $F $R/$O/stats_$mode.csv
  T=$(find /tmp/kt_$mode -name "*kernel_trace.csv" | head -1); python3 $R/tools/step_edges.py $T 8 > $R/$O/edges_$mode.txt 2>&1
  head -3 $R/$O/edges_$mode.txt; grep '^{' $R/$O/prof_$mode.log | head -c 200; echo
done
