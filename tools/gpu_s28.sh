#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "chain_row_length_boundaries" > gpurun_out/s28.log 2>&1
rc=$?
tail -15 gpurun_out/s28.log
exit $rc
