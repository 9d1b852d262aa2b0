#!/bin/bash
# One parameterised GPU call (replaces the 26 one-off r02_call_*.sh of round 2):
#   gpurun --timeout 1200 -- 'bash tools/gpu/call.sh <run-name> <step> [<step> ...]'
# Outputs land in gpurun_out/<run-name>/ (merged back by gpurun); steps run in the order given and the call stops at the first step that was
# KILLED (timeout), so that no further GPU work starts behind a hang.  Steps:
#   tests            python -m pytest tests -m gpu -x -q                        tests:<expr>   the same restricted with -k <expr>
#   bench            python bench.py (default contract run)                     bench200       200 steps, no CPU leg, no verification
#   bench:<args>     python bench.py <args with , for spaces>                   ab:<lib.so>    default / variant / default / variant bench A/B on this box
#   rate_probe       tools/probe/rate_probe.hip                                 fieldmul       tools/probe/fieldmul_probe.hip (f29 / FP64 / MFMA products)
#   latprobe         tools/probe/latency_probe.hip (one wave, dependent chains)
#   latency          tools/latency.py                                           stress         tools/stress.py 20 5
#   verify_bench     tools/verify_bench.py 1024                                 node           napi/example.js on the test key (tests/test_00_gpu_node_addon.py)
#   prof             rocprofv3 --kernel-trace --stats over a 2-step bench       pmc:<name>:<counters,comma>   one PMC pass over a one-pass bench
#   calib            FETCH_SIZE calibration (tools/probe/gather_probe.hip)      py:<script,args>              python <script> <args>
#   trace1           rocprofv3 --kernel-trace over tools/latency.py -> timeline of one proof (tools/single_proof_trace.py)
#   env:NAME=VALUE   export NAME=VALUE for the steps that follow (A/B of a library switch on one box)
set -o pipefail
R="$GRAFT_REPO_ROOT"; [ -n "$R" ] || R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R"; RUN=${1:-run}; shift; O="$R/gpurun_out/$RUN"; mkdir -p "$O"
log() { echo "$*" | tee -a "$O/steps.log"; }
killed() { [ "$1" = 124 ] || [ "$1" = 137 ]; }
PMC_BENCH="python3 $R/bench.py --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --no-verify"
for step in "$@"; do
  name=${step%%:*}; arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
  cd "$R"
  case $name in
    tests)        if [ -n "$arg" ]; then timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "${arg//,/ }" > "$O/gpu_tests.log" 2>&1; else timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$O/gpu_tests.log" 2>&1; fi; rc=$?; tail -4 "$O/gpu_tests.log" ;;
    bench)        if [ -n "$arg" ]; then f="bench_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_')${ZKC_AB_TAG:+_$ZKC_AB_TAG}"; timeout -k 10 600 python bench.py ${arg//,/ } > "$O/$f.json" 2> "$O/$f.err"; rc=$?; head -c 400 "$O/$f.json"; echo
                  else timeout -k 10 500 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; rc=$?; head -c 400 "$O/bench_default.json"; echo; fi ;;
    bench200)     timeout -k 10 300 python bench.py --steps 200 --warmup 2 --no-cpu-baseline --no-verify > "$O/bench_200.json" 2> "$O/bench_200.err"; rc=$?; head -c 260 "$O/bench_200.json"; echo ;;
    ab)           rc=0
                  for mode in default variant default variant; do
                    if [ $mode = variant ]; then export ZKCENSUS_LIB="$R/$arg"; else unset ZKCENSUS_LIB; fi
                    timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > "$O/ab_$mode.json" 2> "$O/ab_$mode.err"; rc=$?
                    [ $rc = 0 ] || { tail -3 "$O/ab_$mode.err"; break; }
                    python - "$O/ab_$mode.json" $mode <<'P' | tee -a "$O/ab.txt"
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], j['value'], j['ms_per_step'])
P
                  done; unset ZKCENSUS_LIB ;;
    rate_probe)   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probe/rate_probe.hip -o /tmp/rate_probe > "$O/rate_probe_build.log" 2>&1 && timeout -k 10 300 /tmp/rate_probe > "$O/rate_probe.txt" 2>&1; rc=$?; head -8 "$O/rate_probe.txt" ;;
    fieldmul)     hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -I zk-franchise-proof-circuit_amd/csrc tools/probe/fieldmul_probe.hip -o /tmp/fieldmul_probe > "$O/fieldmul_build.log" 2>&1 && timeout -k 10 300 /tmp/fieldmul_probe > "$O/fieldmul_probe.txt" 2>&1; rc=$?; cat "$O/fieldmul_probe.txt" ;;
    latprobe)     hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -I zk-franchise-proof-circuit_amd/csrc tools/probe/latency_probe.hip -o /tmp/latency_probe > "$O/latprobe_build.log" 2>&1 && timeout -k 10 120 /tmp/latency_probe > "$O/latency_probe.txt" 2>&1; rc=$?; cat "$O/latency_probe.txt" ;;
    latency)      f="latency${ZKC_AB_TAG:+_$ZKC_AB_TAG}"; timeout -k 10 300 python tools/latency.py > "$O/$f.json" 2> "$O/$f.err"; rc=$?; tail -c 900 "$O/$f.json" ;;
    stress)       timeout -k 10 400 python tools/stress.py 20 5 > "$O/stress.json" 2> "$O/stress.err"; rc=$?; tail -c 600 "$O/stress.json" ;;
    verify_bench) timeout -k 10 300 python tools/verify_bench.py 1024 > "$O/verify_bench.json" 2> "$O/verify_bench.err"; rc=$?; tail -c 400 "$O/verify_bench.json" ;;
    node)         timeout -k 10 600 python -m pytest tests/test_00_gpu_node_addon.py -m gpu -x -q -s > "$O/node_example.log" 2>&1; rc=$?; tail -c 900 "$O/node_example.log" ;;
    py)           f="py_$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_')${ZKC_AB_TAG:+_$ZKC_AB_TAG}"; timeout -k 10 600 python ${arg//,/ } > "$O/$f.out" 2> "$O/$f.err"; rc=$?; tail -c 1200 "$O/$f.out" ;;
    prof)         cd /tmp && export TMPDIR=/tmp
                  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-verify > "$O/prof.log" 2>&1; rc=$?
                  find "$O/prof" -name "*kernel_trace.csv" -size +20M -delete ;;
    pmc)          cd /tmp && export TMPDIR=/tmp; pn=${arg%%:*}; ctrs=${arg#*:}
                  timeout -k 10 300 rocprofv3 --pmc ${ctrs//,/ } --output-format csv -d "$O/pmc_$pn" -- $PMC_BENCH > "$O/pmc_$pn.log" 2>&1; rc=$?; tail -2 "$O/pmc_$pn.log" ;;
    calib)        cd /tmp && export TMPDIR=/tmp
                  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result "$R/tools/probe/gather_probe.hip" -o /tmp/gather_probe > /dev/null 2>&1 && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_calib" -- /tmp/gather_probe > "$O/gather_probe.txt" 2>&1; rc=$? ;;
    trace1)       cd /tmp && export TMPDIR=/tmp
                  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/trace1" -- python3 "$R/tools/latency.py" > "$O/trace1.log" 2>&1; rc=$?
                  f=$(find "$O/trace1" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 "$R/tools/single_proof_trace.py" "$f" > "$O/single_proof_trace.txt" 2>&1; find "$O/trace1" -name "*.csv" -delete; tail -60 "$O/single_proof_trace.txt" ;;
    env)          export "$arg"; rc=0 ;;
    unset)        unset "$arg"; rc=0 ;;                         # env:NAME=VALUE for the steps that follow (env:NAME= clears it)
    *)            log "unknown step $step"; exit 2 ;;
  esac
  log "$step rc=$rc"
  if killed $rc; then log "step $step was killed (rc=$rc): stopping"; exit 1; fi
done
exit 0
