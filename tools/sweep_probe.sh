#!/bin/bash
# Sweep-kernel experiments at the bench shape (N=10M, d=1024, fp64): pipelining flavour x blocks per CU, with plain and
# non-temporal row loads (experiment builds under build/; the product library is not touched).  Runs ON the GPU box.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
for nt in 0 1; do
  if [ $nt = 0 ]; then lib=$("$here/exp_build.sh" plain_loads "-DCIAO_PLAIN_LOADS"); else lib=$("$here/exp_build.sh" nt_loads "-DCIAO_NT_LOADS"); fi
  for pf in 1 0; do for bpc in 1 2 3 4; do
    echo "nt=$nt pf=$pf bpc=$bpc $(cd "$here/.." && CIAO_HIP_LIB=$lib python bench.py --no-cpu --no-extras --steps 12 --warmup 2 --prefetch $pf --blocks-per-cu $bpc 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["roofline"]["achieved"]), "GB/s", round(d["ms_per_step"],3), "ms/step")')"
  done; done
done
