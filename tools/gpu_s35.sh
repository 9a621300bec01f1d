#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_solvers.py -q -m gpu -x -k "proshi or Sharing or sharing" > gpurun_out/s35.log 2>&1
rc=$?
tail -25 gpurun_out/s35.log | cut -c1-200
exit $rc
