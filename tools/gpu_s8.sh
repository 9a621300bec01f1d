#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s8"
mkdir -p "$O"
cd "$R"
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q > "$O/tests.log" 2>&1; echo "tests rc=$?"; tail -5 "$O/tests.log"
cp gpurun_out/parity_observed.json "$O/parity_observed.json" 2>/dev/null
export TMPDIR=/tmp
C="TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL"
echo "== pmc stalls: table modes"; (cd /tmp && TABLE_REPS=2 timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_stall_table" -o t -- python3 "$R/tools/table_modes.py" > "$O/pmc_stall_table.log" 2>&1); echo "rc=$?"
echo "== pmc busy: table modes"; (cd /tmp && TABLE_REPS=2 timeout -k 10 600 rocprofv3 --pmc TCC_BUSY TCC_CYCLE TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace --output-format csv -d "$O/pmc_busy_table" -o t -- python3 "$R/tools/table_modes.py" > "$O/pmc_busy_table.log" 2>&1); echo "rc=$?"
echo "== pmc stalls: sweep"; (cd /tmp && timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_stall_sweep" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras --no-chains > "$O/pmc_stall_sweep.log" 2>&1); echo "rc=$?"
echo "== pmc busy: sweep"; (cd /tmp && timeout -k 10 600 rocprofv3 --pmc TCC_BUSY TCC_CYCLE TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace --output-format csv -d "$O/pmc_busy_sweep" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras --no-chains > "$O/pmc_busy_sweep.log" 2>&1); echo "rc=$?"
ls "$O"/pmc_*/ | head -30
