#!/bin/bash
# GPU call F: kernel trace of two steps (to look at the start and the end of a step) 
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02f2}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "prof rc=$?"
tail -c 300 $GRAFT_REPO_ROOT/$O/prof.log
exit 0
