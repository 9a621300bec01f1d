#!/bin/bash
# A/B of chain micro-optimisations: the product library against build/prev (the previous product library)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "svrg or saga or finito or chain" > gpurun_out/s19_chain_tests.log 2>&1
rc=$?
tail -4 gpurun_out/s19_chain_tests.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  echo "product : $(python tools/chain_time.py) || $(python tools/saga_time.py | tail -1)"
  echo "prev    : $(CIAO_HIP_LIB=$PWD/build/prev/libciao_hip.so python tools/chain_time.py) || $(CIAO_HIP_LIB=$PWD/build/prev/libciao_hip.so python tools/saga_time.py | tail -1)"
done 2>&1 | tee gpurun_out/s19_ab.txt
