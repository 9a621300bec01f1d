#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_complex.py -q -m gpu -x > gpurun_out/s17_parity.log 2>&1
rc=$?
tail -5 gpurun_out/s17_parity.log
exit $rc
