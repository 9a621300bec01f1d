#!/bin/bash
# complex T: the new parity file + the reference's complex lasso testsets, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_complex.py tests/test_gpu_solvers.py -q -m gpu -x > gpurun_out/s11_complex.log 2>&1
rc=$?
tail -25 gpurun_out/s11_complex.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 560 python -m pytest tests -q -m gpu -x > gpurun_out/s11_all.log 2>&1
rc=$?
tail -8 gpurun_out/s11_all.log
exit $rc
