#!/bin/bash
# Timing experiments on the SVRG chain: experiment builds with pieces of the step removed (results are WRONG in these
# builds; only the time per step matters), each in its own build/ directory -- the product library is not touched.
# Runs ON the GPU box.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
for dbg in ${CHAIN_DBG_LIST:-0 1 2 4 3 7}; do
  lib=$("$here/exp_build.sh" chain_dbg$dbg "-DCIAO_CHAIN_DBG=$dbg")
  echo "DBG=$dbg $(cd "$here/.." && CIAO_HIP_LIB=$lib python tools/chain_time.py)"
done
