#!/bin/bash
# GPU call N: kernel trace of a short bench, reduced on the box to the edges of its last step (tools/step_edges.py)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02n}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/prof.log 2>&1; echo "prof rc=$?"
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); head -1 $F > $R/$O/trace_header.txt
python3 $R/tools/step_edges.py $F 20 > $R/$O/step_edges.txt 2>&1; echo "edges rc=$?"; cat $R/$O/step_edges.txt | head -120
