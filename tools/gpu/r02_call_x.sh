#!/bin/bash
# GPU call X: kernel stats + median pass period of the current build (short profiled bench)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02x}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/prof.log 2>&1; echo "prof rc=$?"
F=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $F $R/$O/stats.csv
T=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python3 $R/tools/step_edges.py $T 6 > $R/$O/edges.txt 2>&1; head -2 $R/$O/edges.txt; grep '^{' $R/$O/prof.log | head -c 160; echo
