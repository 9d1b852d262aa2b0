cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_prof20; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/generic_bench.py --logn 20 --no-check > $O/prof.log 2>&1
s=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $s $O/kernel_stats.csv
find $O/prof -type f -size +3M -delete
python3 - $O/kernel_stats.csv <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print('%-50s calls %5s avg %10.3f ms total %9.1f ms' % (r['Name'].split('(')[0][-50:], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
P
