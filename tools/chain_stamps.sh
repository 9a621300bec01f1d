#!/bin/bash
# Rebuilds the chain kernels with s_memtime stamps (CIAO_CHAIN_DBG=8), prints the per-segment cycle averages of an SVRG
# inner cycle, restores the real build.  Runs ON the GPU box.
set -e
cd "$(dirname "$0")/../ciaoalgorithms.jl_amd/csrc"
rm -f chain_f64.o chain_f32.o
make -s -j8 EXTRA="-DCIAO_CHAIN_DBG=8" ../libciao_hip.so >/dev/null 2>&1
(cd ../.. && python tools/chain_stamps.py)
rm -f chain_f64.o chain_f32.o
make -s -j8 ../libciao_hip.so >/dev/null 2>&1
