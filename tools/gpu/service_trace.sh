# rocprofv3 kernel trace of the proving service under load, one trace per configuration given as "name|workers=..,pass=..".  usage: service_trace.sh <threads> <mode> <cfg>...
cd /tmp && export TMPDIR=/tmp
T=$1; M=$2; shift; shift
for c in "$@"; do
  name=${c%%|*}; cfg=${c#*|}
  O=$GRAFT_REPO_ROOT/gpurun_out/r05_svc_trace_$name; mkdir -p $O
  timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/service_load.py --threads $T --calls 6 --modes $M --configs "$cfg" > $O/load.jsonl 2> $O/prof.log || exit 1
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/service_timeline.py $f 0.6 > $O/timeline.txt 2>&1
  find $O/prof -type f -size +1M -delete
  cat $O/load.jsonl | tail -1 | cut -c1-300; cat $O/timeline.txt
done
