#!/bin/bash
# register-ring chain on one wave (rows of up to 512 elements with no 16-byte structure): tests, then timings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py tests/test_golden.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/s27_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s27_tests.log
[ $rc -eq 0 ] || exit $rc
for d in 50 129; do
  echo "d=$d one wave  : $(CIAO_D=$d CIAO_OPTS=chain_no_dma=1 python tools/chain_time.py 2>/dev/null)"
  echo "d=$d four waves: $(CIAO_D=$d CIAO_OPTS=chain_no_dma=1,chain_four_waves=1 python tools/chain_time.py 2>/dev/null)"
done | tee gpurun_out/s27_ab.txt
