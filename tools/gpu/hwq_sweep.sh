# GPU_MAX_HW_QUEUES sweep: service under load + the batch headline.  usage: hwq_sweep.sh <out-name> <q>...
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O; shift
for q in "$@"; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python tools/service_load.py --threads 64,256 --calls 12 --configs "workers=2,pass=96;workers=4,pass=48;workers=3,pass=64" > $O/load_q$q.jsonl 2> $O/load_q$q.err || exit 1
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify > $O/bench_q$q.json 2> $O/bench_q$q.err || exit 1
  echo "q=$q"; head -c 200 $O/bench_q$q.json; echo
done
