#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/s5"
mkdir -p "$O"
cd "$R"
for rep in 1 2 3; do
for v in product st_t1p1 st_t0p2 st_t2p2; do
  if [ $v = product ]; then unset CIAO_HIP_LIB; else export CIAO_HIP_LIB="$R/build/$v/libciao_hip.so"; fi
  echo "== rep $rep $v"; timeout -k 10 300 python tools/finito_batch_time.py 256 4096 16384 2>&1 | grep "r=" | sort -u | awk '{print $1, $2}' | tr '\n' ' '; echo
done
done 2>&1 | tee "$O/ab.log"
