#!/bin/bash
# dense ProShI: the new tests first, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py -q -m gpu -x -k "proshi or Sharing or sharing" > gpurun_out/s12_proshi.log 2>&1
rc=$?
tail -25 gpurun_out/s12_proshi.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests -q -m gpu -x > gpurun_out/s12_all.log 2>&1
rc=$?
tail -8 gpurun_out/s12_all.log
exit $rc
