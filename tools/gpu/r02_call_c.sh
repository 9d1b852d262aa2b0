#!/bin/bash
# GPU call C: the evidence run -- full GPU suite, default bench, probes, rocprofv3 kernel stats + PMC passes, config-2 latency, config-5 stress
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02c2}; mkdir -p $O
stop_if_killed() { if [ "$1" = 124 ] || [ "$1" = 137 ]; then echo "step $2 was killed (rc=$1): stopping" | tee -a $O/steps.log; exit 1; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/steps.log; tail -4 $O/gpu_tests.log; stop_if_killed $rc pytest
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; echo "bench rc=$rc" | tee -a $O/steps.log; head -c 300 $O/bench_default.json; echo; stop_if_killed $rc bench
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probe/rate_probe.hip -o /tmp/rate_probe > $O/probe_build.log 2>&1 && timeout -k 10 300 /tmp/rate_probe > $O/rate_probe.txt 2>&1; rc=$?; echo "probe rc=$rc" | tee -a $O/steps.log; head -8 $O/rate_probe.txt; stop_if_killed $rc probe
timeout -k 10 300 python tools/latency.py > $O/latency.json 2> $O/latency.err; rc=$?; echo "latency rc=$rc" | tee -a $O/steps.log; tail -c 600 $O/latency.json; stop_if_killed $rc latency
timeout -k 10 400 python tools/stress.py 20 5 > $O/stress.json 2> $O/stress.err; rc=$?; echo "stress rc=$rc" | tee -a $O/steps.log; tail -c 600 $O/stress.json; stop_if_killed $rc stress
timeout -k 10 300 python tools/verify_bench.py 1024 > $O/verify_bench.json 2> $O/verify_bench.err; rc=$?; echo "verify_bench rc=$rc" | tee -a $O/steps.log; tail -c 400 $O/verify_bench.json; stop_if_killed $rc verify_bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $R/$O/prof.log 2>&1; rc=$?; echo "prof rc=$rc" | tee -a $R/$O/steps.log; stop_if_killed $rc prof
find $R/$O/prof -name "*kernel_trace.csv" -delete
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --batch 96 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/pmc_fetch.log 2>&1; rc=$?; echo "pmc fetch rc=$rc" | tee -a $R/$O/steps.log; stop_if_killed $rc pmc_fetch
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --batch 96 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/pmc_write.log 2>&1; rc=$?; echo "pmc write rc=$rc" | tee -a $R/$O/steps.log; stop_if_killed $rc pmc_write
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_sq -- python3 $R/bench.py --batch 96 --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/pmc_sq.log 2>&1; rc=$?; echo "pmc sq rc=$rc" | tee -a $R/$O/steps.log
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result $R/tools/probe/gather_probe.hip -o /tmp/gather_probe > /dev/null 2>&1 && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_calib -- /tmp/gather_probe > $R/$O/gather_probe.txt 2>&1; echo "calibration rc=$?" | tee -a $R/$O/steps.log
timeout -k 10 300 python $R/bench.py --steps 200 --warmup 2 --no-cpu-baseline --no-verify > $R/$O/bench_200.json 2> $R/$O/bench_200.err; echo "bench200 rc=$?" | tee -a $R/$O/steps.log; head -c 260 $R/$O/bench_200.json; echo
exit 0
