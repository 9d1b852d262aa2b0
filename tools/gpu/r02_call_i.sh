#!/bin/bash
# GPU call I: the driver's round-end sequence in small -- smoke(), the self-launched two-rank path on one GPU, a sustained 25-step bench
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02i}; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc" | tee -a $O/steps.log; tail -2 $O/smoke.log; [ $rc = 0 ] || exit 1
ZKC_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --batch 192 --steps 1 --warmup 1 --no-cpu-baseline > $O/two_rank.json 2> $O/two_rank.err; rc=$?; echo "two-rank rc=$rc" | tee -a $O/steps.log; head -c 400 $O/two_rank.json; echo; [ $rc = 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 25 --warmup 2 --no-cpu-baseline > $O/bench_25.json 2> $O/bench_25.err; rc=$?; echo "bench25 rc=$rc" | tee -a $O/steps.log; head -c 300 $O/bench_25.json; echo
exit 0
