#!/bin/bash
# The round's record runs, ON the GPU box (gpurun -- tools/record_runs.sh [a|b|c ...]); everything lands in gpurun_out/final/
# and the summaries to be judged are copied from there into profiles/rNN_*.
#   a  the bench line (update figures; the extras go to gpurun_out/bench_extras.json) and the fp32 line
#   b  rocprofv3 --kernel-trace --stats of the same bench command; the separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   c  rocprofv3 --kernel-trace --stats of the update figures at their own sizes (one SVRG outer iteration m = N = 10M, SAGA at
#      BASELINE config #3 and in fp64, Finito batches at config #5's per-rank shape) and of the extras
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
O="$R/gpurun_out/final"
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
for part in "${@:-a}"; do
case "$part" in
a)
  echo "== bench (default)"; timeout -k 10 1000 python bench.py > "$O/bench_f64.json" 2> "$O/bench_f64.err"; echo "rc=$?"; python -c "
import json;j=json.load(open('$O/bench_f64.json'));print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_avg_ms']); print(j['svrg_updates_per_sec']['value'], j['saga_updates_per_sec']['value'], j['finito_samples_per_sec']['value']); print(j['cpu_baseline']['value'], j.get('extras_file'))"; cp -f "$R/gpurun_out/bench_extras.json" "$O/bench_extras.json" 2>/dev/null
  echo "== bench f32"; timeout -k 10 600 python bench.py --dtype f32 --no-extras --no-chains > "$O/bench_f32.json" 2> "$O/bench_f32.err"; echo "rc=$?"; python -c "
import json;j=json.load(open('$O/bench_f32.json'));print(j['value'], j['ms_per_step'], j['roofline']['frac'])"
  ;;
b)
  echo "== rocprof stats f64"; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_f64" -o b -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-extras --no-chains > "$O/prof_f64.log" 2>&1); echo "rc=$?"
  echo "== rocprof stats f32"; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_f32" -o b -- python3 "$R/bench.py" --dtype f32 --steps 10 --warmup 2 --no-cpu --no-extras --no-chains > "$O/prof_f32.log" 2>&1); echo "rc=$?"
  for t in f64 f32; do
    for c in FETCH_SIZE WRITE_SIZE; do
      echo "== pmc $c $t"; (cd /tmp && timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_${t}_$c" -o b -- python3 "$R/bench.py" --dtype $t --steps 3 --warmup 1 --no-cpu --no-extras --no-chains > "$O/pmc_${t}_$c.log" 2>&1); echo "rc=$?"
    done
  done
  F64F=$(find "$O/pmc_f64_FETCH_SIZE" -name '*counter_collection.csv' | head -1); F64W=$(find "$O/pmc_f64_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
  F32F=$(find "$O/pmc_f32_FETCH_SIZE" -name '*counter_collection.csv' | head -1); F32W=$(find "$O/pmc_f32_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
  python tools/pmc_sweep_traffic.py ls_f64_N10000000_d1024 "$F64F" "$F64W" > "$O/pmc_traffic_1.json" && python tools/pmc_sweep_traffic.py ls_f32_N10000000_d1024 "$F32F" "$F32W" "$O/pmc_traffic_1.json" > "$O/pmc_traffic.json"; cat "$O/pmc_traffic.json"
  ;;
c)
  echo "== rocprof stats chains at N=10M"; (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_chains" -o c -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-extras > "$O/prof_chains.log" 2>&1); echo "rc=$?"
  echo "== rocprof stats extras"; (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_extras" -o e -- python3 "$R/tools/run_extras.py" > "$O/extras.json" 2> "$O/extras.err"); echo "rc=$?"
  ;;
esac
done
ls "$O" | head -40
