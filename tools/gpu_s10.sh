#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s10"
mkdir -p "$O"
cd "$R"
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q -x > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -25 "$O/tests.log"
echo "== big-d chain speed"; for D in 9000 16384 32768; do CIAO_D=$D timeout -k 10 200 python tools/chain_time.py 2>&1 | tail -1; done
