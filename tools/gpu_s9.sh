#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s9"
mkdir -p "$O"
cd "$R"
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -15 "$O/tests.log"
cp gpurun_out/parity_observed.json "$O/parity_observed.json" 2>/dev/null
echo "== chain speed check"; timeout -k 10 200 python tools/chain_time.py 2>&1 | tail -3
timeout -k 10 200 python tools/saga_time.py 2>&1 | tail -3
