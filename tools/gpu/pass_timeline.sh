cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_timeline; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/pass_timeline.py $f 15 > $O/pass_timeline.txt 2>&1
find $O/prof -type f -size +1M -delete
cat $O/pass_timeline.txt
