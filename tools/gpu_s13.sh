#!/bin/bash
# adaptive Finito: complex T and rows of any length; then the whole GPU suite with durations
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_complex.py tests/test_gpu_parity.py tests/test_gpu_solvers.py -q -m gpu -x -k "adaptive or Complex" > gpurun_out/s13_adaptive.log 2>&1
rc=$?
tail -30 gpurun_out/s13_adaptive.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/s13_all.log 2>&1
rc=$?
tail -30 gpurun_out/s13_all.log
exit $rc
