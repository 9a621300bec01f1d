# HBM traffic of rows_long_kernel per launch (d = 32768 fp64, 8 GB): two separate --pmc passes of tools/long_rows_time.py, corrected by
# tools/pmc_sweep_traffic.py as MI355X_MICROARCH.md prescribes; on the GPU box: gpurun -- bash tools/exp/long_rows_pmc.sh
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/long"; mkdir -p "$O"
export TMPDIR=/tmp CIAO_D=32768 CIAO_GB=8
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -o l -- python3 "$R/tools/long_rows_time.py" > "$O/pmc_$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$O/pmc_$c.log"; exit 1; }
done
cd "$R"
python tools/pmc_sweep_traffic.py long_f64_N30517_d32768 "$(find $O/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)" "$(find $O/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" | tee "$O/pmc_traffic_long.json"
