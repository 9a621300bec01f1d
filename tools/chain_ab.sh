#!/bin/bash
# A/B of chain changes on ONE box: the product library against build/prev/libciao_hip.so (the previous library), after the chain
# parity tests.  us per update: SVRG (two dot products / cached row dots), fp64 and fp32, N=200k d=1024; SAGA at BASELINE config #3.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "svrg or saga or finito or chain or wave_spec" > gpurun_out/chain_tests.log 2>&1
rc=$?
tail -3 gpurun_out/chain_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  echo "product : $(python tools/chain_time.py) || $(python tools/saga_time.py | tail -1)"
  echo "prev    : $(CIAO_HIP_LIB=$PWD/build/prev/libciao_hip.so python tools/chain_time.py) || $(CIAO_HIP_LIB=$PWD/build/prev/libciao_hip.so python tools/saga_time.py | tail -1)"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/chain_ab.txt
