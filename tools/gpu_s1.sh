#!/bin/bash
# GPU session 1 (round 2): test suite, batch trace (durations + gaps), table-mode rates + PMC passes, bench line.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s1"
mkdir -p "$O"
cd "$R"
echo "== tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -3 "$O/tests.log"
export TMPDIR=/tmp
echo "== batch times (plain)"; timeout -k 10 300 python tools/finito_batch_time.py 16 64 256 1024 4096 16384 > "$O/batch_plain.log" 2>&1; cat "$O/batch_plain.log"
echo "== batch trace"; (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/prof_batches" -o fb -- python3 "$R/tools/finito_batch_time.py" 64 256 4096 > "$O/batch_prof.log" 2>&1); echo "rc=$?"
f=$(find "$O/prof_batches" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py "$f" rows_ finalize > "$O/batch_gaps.json" && cat "$O/batch_gaps.json"
echo "== table modes (plain)"; timeout -k 10 600 python tools/table_modes.py > "$O/table_plain.log" 2>&1; tail -1 "$O/table_plain.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); [print(k, round(v['alg_GBps']), v['kernel']) for k,v in d.items()]"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== table modes pmc $c"; (cd /tmp && TABLE_REPS=2 timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_table_$c" -o t -- python3 "$R/tools/table_modes.py" > "$O/pmc_table_$c.log" 2>&1); echo "rc=$?"
done
echo "== bench"; timeout -k 10 600 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; python -c "
import json;j=json.load(open('$O/bench.json'));print({k:j[k] for k in ('value','ms_per_step','sweeps_per_sec')}, j['roofline']['frac']);
print(j.get('svrg_updates_per_sec'),'\n',j.get('saga_updates_per_sec'),'\n',[ (k,v) for k,v in j.items() if k.startswith('svrg_epochs')], j.get('chain_figures_error'))"
