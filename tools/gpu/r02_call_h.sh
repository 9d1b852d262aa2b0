#!/bin/bash
# GPU call H: kernel trace of the single-proof latency tool
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02h2}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/latency.py > $GRAFT_REPO_ROOT/$O/latency.log 2>&1; echo "rc=$?"; tail -2 $GRAFT_REPO_ROOT/$O/latency.log | cut -c1-300
