#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s6"
mkdir -p "$O"
cd "$R"
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q -x > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -5 "$O/tests.log"
echo "== blocks vs lists"; CIAO_BLOCKS=1 timeout -k 10 300 python tools/finito_batch_time.py 16 64 256 1024 4096 2>&1 | grep "static" | sort -u | tee "$O/blocks.log"
