#!/bin/bash
# short rows on the single-wave chain: chain tests with it (default) and with chain_four_waves=1, then timings at d = 50 .. 512
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py tests/test_gpu_configs.py tests/test_golden.py -q -m gpu -x > gpurun_out/s25_tests.log 2>&1
rc=$?
tail -3 gpurun_out/s25_tests.log
[ $rc -eq 0 ] || exit $rc
for d in 128 512; do
  echo "d=$d one wave  : $(CIAO_D=$d python tools/chain_time.py 2>/dev/null)"
  echo "d=$d four waves: $(CIAO_D=$d CIAO_OPTS=chain_four_waves=1 python tools/chain_time.py 2>/dev/null)"
done | tee gpurun_out/s25_ab.txt
