#!/bin/bash
# The round's record runs: bench line (with chains + extras), rocprofv3 kernel stats of the same command, PMC traffic passes.
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
echo "== rocprof stats f64"; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_f64" -o b -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-extras --no-chains > "$O/prof_f64.log" 2>&1); echo "rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c f64"; (cd /tmp && timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_f64_$c" -o b -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-extras --no-chains > "$O/pmc_f64_$c.log" 2>&1); echo "rc=$?"
done
echo "== rocprof stats extras"; (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_extras" -o e -- python3 "$R/tools/run_extras.py" > "$O/extras.json" 2> "$O/extras.err"); echo "rc=$?"
ls "$O" "$O"/prof_f64 | head -40
