#!/bin/bash
# Timing experiments on the SVRG chain: rebuild libciao_hip.so with pieces of the step removed (results are WRONG in
# these builds; only the time per step matters), run the S3 probe, restore the real build.  Runs ON the GPU box.
set -e
cd "$(dirname "$0")/../ciaoalgorithms.jl_amd/csrc"
for dbg in ${CHAIN_DBG_LIST:-0 1 2 4 3 7}; do
  rm -f chain_f64.o chain_f32.o
  make -s -j8 EXTRA="-DCIAO_CHAIN_DBG=$dbg" >/dev/null 2>&1
  echo "DBG=$dbg $(cd ../.. && python tools/chain_time.py)"
done
rm -f chain_f64.o chain_f32.o
make -s -j8 >/dev/null 2>&1
