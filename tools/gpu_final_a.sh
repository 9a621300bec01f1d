#!/bin/bash
# Part A of the round's record runs: the bench line (with chains + extras) and the fp32 line.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/final"
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
echo "== bench (default)"; timeout -k 10 900 python bench.py > "$O/bench_f64.json" 2> "$O/bench_f64.err"; echo "rc=$?"; python -c "
import json;j=json.load(open('$O/bench_f64.json'));print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_avg_ms']); print(j['svrg_updates_per_sec']['value'], j['saga_updates_per_sec']['value']); print(j['cpu_baseline']['value'], list(j['extra'].keys())[:3])"
echo "== bench f32"; timeout -k 10 600 python bench.py --dtype f32 --no-extras --no-chains > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "rc=$?"; python -c "
import json;j=json.load(open('$O/bench_f32.json'));print(j['value'], j['ms_per_step'], j['roofline']['frac'])"
ls "$O"
