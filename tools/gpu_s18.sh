#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bench_launcher.py tests/test_gpu_multirank.py -q -m gpu -x > gpurun_out/s18_launch.log 2>&1
rc=$?
tail -15 gpurun_out/s18_launch.log
exit $rc
