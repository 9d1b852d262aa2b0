#!/bin/bash
# round-2 GPU call A: full GPU test suite, default bench, instruction-rate probe, two-rank rehearsal on one GPU, SQ counters of the accumulation
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02a; mkdir -p $O
stop_if_killed() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "step $2 was killed (rc=$1): stopping" | tee -a $O/steps.log; exit 1; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/steps.log; tail -5 $O/gpu_tests.log; stop_if_killed $rc pytest
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc" | tee -a $O/steps.log; tail -c 1500 $O/bench_default.json; stop_if_killed $rc bench
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probe/rate_probe.hip -o /tmp/rate_probe > $O/probe_build.log 2>&1 && timeout -k 10 200 /tmp/rate_probe > $O/rate_probe.txt 2>&1; rc=$?; echo "probe rc=$rc" | tee -a $O/steps.log; cat $O/rate_probe.txt; stop_if_killed $rc probe
ZKC_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --batch 192 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_share2.json 2> $O/bench_share2.err; rc=$?; echo "share2 rc=$rc" | tee -a $O/steps.log; tail -c 600 $O/bench_share2.json; stop_if_killed $rc share2
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py --batch 96 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1; rc=$?; echo "pmc rc=$rc" | tee -a $GRAFT_REPO_ROOT/$O/steps.log
tail -3 $GRAFT_REPO_ROOT/$O/pmc_sq.log
exit 0
