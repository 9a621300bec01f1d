#!/bin/bash
# adaptive Finito on short rows: single-wave kernel vs four waves
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py tests/test_gpu_complex.py -q -m gpu -x -k "adaptive" > gpurun_out/s31_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s31_tests.log
[ $rc -eq 0 ] || exit $rc
for d in 64 128 256; do
  echo "d=$d one wave  : $(CIAO_D=$d python tools/af_time.py 2>/dev/null | tail -1)"
  echo "d=$d four waves: $(CIAO_D=$d CIAO_OPTS=chain_four_waves=1 python tools/af_time.py 2>/dev/null | tail -1)"
done | tee gpurun_out/s31_ab.txt
