#!/bin/bash
# GPU session 2: launch floor calibration; Finito batches eager vs captured graph; rerun of the test suite.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s2"
mkdir -p "$O"
cd "$R"
echo "== launch floor"; timeout -k 10 120 tools/micro/launch_floor > "$O/launch_floor.log" 2>&1; cat "$O/launch_floor.log"
echo "== batches eager"; timeout -k 10 300 python tools/finito_batch_time.py 16 64 256 1024 4096 > "$O/batch_eager.log" 2>&1; cat "$O/batch_eager.log"
echo "== batches graph"; CIAO_OPTS=graph_batches=1 timeout -k 10 300 python tools/finito_batch_time.py 16 64 256 1024 4096 > "$O/batch_graph.log" 2>&1; cat "$O/batch_graph.log"
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -5 "$O/tests.log"
