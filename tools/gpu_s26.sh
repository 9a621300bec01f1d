#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "short_rows" > gpurun_out/s26.log 2>&1
rc=$?
tail -15 gpurun_out/s26.log
exit $rc
