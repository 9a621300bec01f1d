#!/bin/bash
# Round 5: where a Finito batch over an index list on rows of tabular size spends its time -- rocprofv3 --kernel-trace of
# tools/small_batch_time.py for one shape at a time, reduced to the rows kernel's and finalize's median durations and the batch period.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/wrow_trace"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for shape in "50 f64" "100 f64" "50 f32" "255 f64"; do
  set -- $shape
  CIAO_DS=$1 CIAO_DT=$2 CIAO_RS=65536 CIAO_LFINITO_ONLY= timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$O/t_$1_$2" -o t -- python3 "$R/tools/small_batch_time.py" > "$O/log_$1_$2.txt" 2>&1
  python3 - "$O/t_$1_$2" "$1 $2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = {}
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "rows_wrow_kernel" not in n and "rows_smallm_kernel" not in n and "rows_generic_kernel" not in n: continue
    if i + 1 >= len(rows) or "finalize" not in rows[i + 1]["Kernel_Name"]: continue
    fin = rows[i + 1]
    nxt = next((x for x in rows[i + 2:i + 4] if x["Kernel_Name"] == n), None)
    key = n.split("(")[0].replace("void ciao::", "") + " grid=" + r["Grid_Size"]
    out.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(fin["End_Timestamp"]) - int(fin["Start_Timestamp"]),
                                    int(fin["Start_Timestamp"]) - int(r["End_Timestamp"]), (int(nxt["Start_Timestamp"]) - int(r["Start_Timestamp"])) if nxt else 0))
med = lambda v: sorted(v)[len(v) // 2] / 1e3
for k, v in out.items():
    if len(v) < 6: continue
    per = [x[3] for x in v if x[3]]
    print("d=%s  %-70s n=%3d  rows kernel %6.1f us | gap %4.1f | finalize %5.1f us | period %6.1f us" % (sys.argv[2], k[:70], len(v), med([x[0] for x in v]), med([x[2] for x in v]), med([x[1] for x in v]), med(per) if per else 0))
PY
done
