#!/bin/bash
# whole GPU suite, then batch / adaptive timings against the previous library
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/s21_all.log 2>&1
rc=$?
tail -3 gpurun_out/s21_all.log
[ $rc -eq 0 ] || exit $rc
for lib in "" "$PWD/build/prev/libciao_hip.so"; do
  echo "== ${lib:-product}"
  CIAO_HIP_LIB=$lib python tools/finito_batch_time.py 2>/dev/null | tail -8
  CIAO_HIP_LIB=$lib python tools/af_time.py 2>/dev/null | tail -2
done 2>&1 | tee gpurun_out/s21_ab.txt
