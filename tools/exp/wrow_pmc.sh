#!/bin/bash
# Round 5: HBM traffic of an index-list Finito batch on rows of tabular size (rows_wrow_kernel) by the counters: separate --pmc FETCH_SIZE /
# WRITE_SIZE passes over tools/small_batch_time.py (one shape, r = 65 536), bytes = 2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024 (the gfx950
# corrections of tools/pmc_sweep_traffic.py), against the algorithmic 3 d s + 2 s + 8 bytes per sample.  (ON the GPU box.)
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/wrow_pmc"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for shape in "50 f64" "100 f64" "255 f64" "50 f32"; do
  set -- $shape
  for c in FETCH_SIZE WRITE_SIZE; do
    CIAO_DS=$1 CIAO_DT=$2 CIAO_RS=65536 timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/p_$1_$2_$c" -o p -- python3 "$R/tools/small_batch_time.py" > "$O/log_$1_$2_$c.txt" 2>&1 || exit 1
  done
  python3 - "$O" $1 $2 <<'PY'
import csv, glob, sys
O, d, dt = sys.argv[1], int(sys.argv[2]), sys.argv[3]
es = 8 if dt == "f64" else 4
def med(counter):
    f = glob.glob(f"{O}/p_{d}_{dt}_{counter}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "rows_wrow_kernel" in r["Kernel_Name"] and ", 4>" in r["Kernel_Name"].split("(")[0] and r["Counter_Name"] == counter]
    v.sort()
    return v[len(v) // 2], len(v)
f, n = med("FETCH_SIZE"); w, _ = med("WRITE_SIZE")
traffic = 2 * f * 1024 + w * 1024
alg = 65536 * (3 * d * es + 2 * es + 8)
print(f"d={d} {dt} r=65536 rows_wrow_kernel: FETCH_SIZE {f:.0f} KiB, WRITE_SIZE {w:.0f} KiB (median of {n} launches) -> {traffic / 1e6:.1f} MB of HBM traffic against {alg / 1e6:.1f} MB algorithmic = {traffic / alg:.2f}x")
PY
done
