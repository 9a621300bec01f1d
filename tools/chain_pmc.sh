#!/bin/bash
# Instruction counters of the chain kernels (ON the GPU box): two rocprofv3 --pmc passes over tools/chain_time.py (SVRG: two dot
# products / cached row dots, fp64 and fp32, d = 1024) and tools/saga_time.py (SAGA at BASELINE config #3), summarised per wave and
# step by tools/chain_pmc_summary.py -> gpurun_out/chain_pmc/summary.json (copied to profiles/rNN_chain_pmc_instructions.json).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/chain_pmc"
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 500 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/svrg_$i" -o c -- python3 "$R/tools/chain_time.py" > "$O/svrg_$i.log" 2>&1); echo "svrg pass $i rc=$?"
  (cd /tmp && timeout -k 10 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/saga_$i" -o c -- python3 "$R/tools/saga_time.py" > "$O/saga_$i.log" 2>&1); echo "saga pass $i rc=$?"
done
python3 "$R/tools/chain_pmc_summary.py" "$O" > "$O/summary.json" && cat "$O/summary.json"
