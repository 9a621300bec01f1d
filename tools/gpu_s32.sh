#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_chain" > gpurun_out/s32.log 2>&1
rc=$?
tail -40 gpurun_out/s32.log | cut -c1-220
exit $rc
