#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s7"
mkdir -p "$O"
cd "$R"
echo "== stream ceiling"; timeout -k 10 300 tools/micro/stream_copy > "$O/stream_copy.log" 2>&1; cat "$O/stream_copy.log"
echo "== table modes (same box)"; timeout -k 10 600 python tools/table_modes.py 2>/dev/null | tail -1 > "$O/table.json"; python -c "
import json; d=json.load(open('$O/table.json')); [print(k, round(v['alg_GBps']), v['kernel']) for k,v in d.items()]"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > "$O/counters.txt" 2>&1); grep -c . "$O/counters.txt"; grep -o "TCC_[A-Z0-9_]*" "$O/counters.txt" | sort -u | tr '\n' ' ' | head -c 4000; echo
