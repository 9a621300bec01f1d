#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py -q -m gpu -x -k "adaptive" > gpurun_out/s24_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s24_tests.log
[ $rc -eq 0 ] || exit $rc
for lib in "" "$PWD/build/prev/libciao_hip.so"; do
  echo "== ${lib:-product}"
  CIAO_HIP_LIB=$lib python tools/af_time.py 2>/dev/null | tail -1
done
