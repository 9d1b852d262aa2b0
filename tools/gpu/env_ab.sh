# alternating A/B of library switches on the batch headline, one box: env_ab.sh <out-name> "<VAR=VAL ...>" ["<VAR=VAL ...>" ...]   (an empty string is the default configuration)
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O; shift
for rep in 1 2; do
  for cfg in "" "$@"; do
    tag=$(echo "${cfg:-default}" | tr -c 'A-Za-z0-9=\n' '_')
    env $cfg timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify > $O/b.json 2>/dev/null || exit 1
    python3 -c "
import json
d=json.load(open('$O/b.json')); print('%-60s %8.1f  launch %.2f ms' % ('${cfg:-default}', d['value'], d['roofline']['avg_launch_ms']))" | tee -a $O/ab.txt
  done
done
