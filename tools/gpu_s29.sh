#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/s29.log 2>&1
rc=$?
tail -15 gpurun_out/s29.log
exit $rc
