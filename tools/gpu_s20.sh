#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --no-extras --no-cpu > gpurun_out/s20_bench.json 2> gpurun_out/s20_bench.err
rc=$?
python - <<'PY'
import json
j = json.load(open("gpurun_out/s20_bench.json"))
print("value", j["value"], "frac", j["roofline"]["frac"])
for k in ("svrg_updates_per_sec", "saga_updates_per_sec", "svrg_epochs_per_sec_N10M"):
    print(k, {a: b for a, b in j[k].items() if a not in ("roofline", "what")})
PY
exit $rc
