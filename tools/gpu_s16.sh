#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/s16_all.log 2>&1
rc=$?
tail -20 gpurun_out/s16_all.log
exit $rc
