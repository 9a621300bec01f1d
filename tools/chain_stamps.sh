#!/bin/bash
# Experiment build of the chain kernels with s_memtime stamps (CIAO_CHAIN_DBG=8, in build/chain_dbg8/); prints the
# per-segment cycle averages of an SVRG inner cycle.  Runs ON the GPU box.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
lib=$("$here/exp_build.sh" chain_dbg8 "-DCIAO_CHAIN_DBG=8")
(cd "$here/.." && CIAO_HIP_LIB=$lib python tools/chain_stamps.py)
