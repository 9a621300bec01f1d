#!/bin/bash
# Sweep-kernel experiments at the bench shape (N=10M, d=1024, fp64): pipelining flavour x blocks per CU, with plain and
# non-temporal row loads.  Runs ON the GPU box; restores the real build at the end.
set -e
cd "$(dirname "$0")/../ciaoalgorithms.jl_amd/csrc"
for nt in 0 1; do
  rm -f rows_f64.o rows_f32.o
  if [ $nt = 0 ]; then make -s -j8 EXTRA="-DCIAO_PLAIN_LOADS" >/dev/null 2>&1; else make -s -j8 >/dev/null 2>&1; fi
  for pf in 1 0; do for bpc in 1 2 3 4; do
    echo "nt=$nt pf=$pf bpc=$bpc $(cd ../.. && python bench.py --no-cpu --no-extras --steps 12 --warmup 2 --prefetch $pf --blocks-per-cu $bpc 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["roofline"]["achieved"]), "GB/s", round(d["ms_per_step"],3), "ms/step")')"
  done; done
done
rm -f rows_f64.o rows_f32.o
make -s -j8 >/dev/null 2>&1
