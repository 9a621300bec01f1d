#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_complex.py tests/test_gpu_solvers.py -q -m gpu -x > gpurun_out/s15_complex.log 2>&1
rc=$?
tail -25 gpurun_out/s15_complex.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/run_extras.py > gpurun_out/s15_extras.json 2> gpurun_out/s15_extras.err
rc=$?
python - <<'PY'
import json
j = json.load(open("gpurun_out/s15_extras.json"))
for k, v in j.items():
    if "complex" in k or "dense" in k or "8192" in k:
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "N"})
PY
exit $rc
