#!/bin/bash
# A/B of a chain-kernel change on ONE box: the product library against an experiment build of the alternative
# (tools/exp_build.sh NAME "FLAGS" ...; CIAO_AB_LIB=build/NAME/libciao_hip.so).  us per update: SVRG (two dot products / cached row
# dots, fp64 and fp32, d = 1024), SAGA at BASELINE config #3 on the wave-specialised kernel and on chain_dma_kernel.
set -o pipefail
mkdir -p gpurun_out
B="${CIAO_AB_LIB:?set CIAO_AB_LIB to the experiment library (tools/exp_build.sh prints its path)}"
for rep in 1 2 3; do
  echo "product : $(python tools/chain_time.py) || ws $(python tools/saga_time.py | tail -1 | cut -c1-60) || dma $(CIAO_OPTS=chain_no_ws=1 python tools/saga_time.py | tail -1 | cut -c1-60)"
  echo "other   : $(CIAO_HIP_LIB=$B python tools/chain_time.py) || ws $(CIAO_HIP_LIB=$B python tools/saga_time.py | tail -1 | cut -c1-60) || dma $(CIAO_HIP_LIB=$B CIAO_OPTS=chain_no_ws=1 python tools/saga_time.py | tail -1 | cut -c1-60)"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/chain_ab.txt
