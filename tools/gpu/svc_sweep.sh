# service sweep: usage svc_sweep.sh <out-name> <threads> <calls> <modes> <configs> [ENV=VAL ...]
cd $GRAFT_REPO_ROOT; O=gpurun_out/$1; mkdir -p $O; T=$2; C=$3; M=$4; CFG=$5; shift 5
for e in "$@"; do export "$e"; done
timeout -k 10 900 python tools/service_load.py --threads $T --calls $C --modes $M --configs "$CFG" --rounds 2 > $O/load.jsonl 2> $O/load.err || exit 1
