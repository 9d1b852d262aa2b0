#!/bin/bash
# GPU call L: the Node example (before anything in this shell touches the GPU) and the latency tool
set -o pipefail
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r02l}; mkdir -p $O
Z=build/artifacts/zkcensus_160.zkey; V=build/artifacts/zkcensus_160_vkey.json
ls build/artifacts | head -20 > $O/artifacts.txt
python - <<'P' > $O/paths.txt
from zkcensus_amd import setup
print(*setup.ensure_test_artifacts(160)[1:])
P
read Z V < $O/paths.txt
timeout -k 10 300 node napi/example.js $Z $V > $O/node_example.json 2> $O/node_example.err; rc=$?; echo "node rc=$rc"; tail -c 700 $O/node_example.json; tail -3 $O/node_example.err
timeout -k 10 300 python tools/latency.py > $O/latency.json 2> $O/latency.err; rc=$?; echo "latency rc=$rc"; tail -c 900 $O/latency.json
exit 0
