#!/bin/bash
# GPU call Q: lane utilisation per kernel (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)) over a one-pass bench: which kernels spend issue slots on idle lanes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02q}; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_lanes -- python3 $R/bench.py --batch 96 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/pmc_lanes.log 2>&1; echo "pmc rc=$?"; tail -2 $R/$O/pmc_lanes.log
