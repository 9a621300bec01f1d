#!/bin/bash
# extras incl. the new kernels (dense ProShI, complex sweep / chain, adaptive beyond 4096)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/run_extras.py > gpurun_out/s14_extras.json 2> gpurun_out/s14_extras.err
rc=$?
tail -5 gpurun_out/s14_extras.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/s14_extras.json"))
for k, v in j.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "N"})
PY
exit $rc
