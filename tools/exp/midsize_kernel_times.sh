#!/bin/bash
# Round 5, VERDICT r4 item 6: where do the mid-size batches (Finito r = 1024 .. 16384 at d = 4096 fp32, ProShI r = 1024 .. 16384 at d = 1024
# fp64) stand against the mixed-traffic ceiling ONCE THE FIXED COST OF A BATCH IS TAKEN OUT?  rocprofv3 --kernel-trace of the two timing
# scripts; per batch size: the rows kernel's own duration, the finalize kernel's, and the launch-to-launch period (ON the GPU box).
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/midsize"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/finito" -o f -- python3 "$R/tools/finito_batch_time.py" 1024 2048 4096 8192 16384 65536 > "$O/finito.log" 2>&1; echo "finito rc=$?"
CIAO_D=1024 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/proshi" -o p -- python3 "$R/tools/proshi_batch_time.py" > "$O/proshi.log" 2>&1; echo "proshi rc=$?"
python3 - <<PY
import csv, glob, collections
for what, per_row in (("finito", 3 * 4096 * 4 + 16), ("proshi", 4 * 1024 * 8)):
    f = glob.glob("$O/%s/**/*kernel_trace.csv" % what, recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = collections.defaultdict(list)
    prev_end = {}
    for i, r in enumerate(rows):
        name = r["Kernel_Name"].split("(")[0].replace("void ciao::", "")
        if not (name.startswith("rows_") or name.startswith("proshi_")):
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        nxt = next((x for x in rows[i + 1:i + 4] if x["Kernel_Name"].startswith("void ciao::finalize")), None)
        fin = int(nxt["End_Timestamp"]) - int(nxt["Start_Timestamp"]) if nxt else 0
        span = int(nxt["End_Timestamp"]) - int(r["Start_Timestamp"]) if nxt else dur
        agg[(name, r["Grid_Size"])].append((dur, fin, span))
    print("==", what, "(bytes per row %d)" % per_row)
    for (name, grid), v in agg.items():
        if len(v) < 4: continue
        d = sorted(x[0] for x in v)[len(v) // 2] / 1e3; fi = sorted(x[1] for x in v)[len(v) // 2] / 1e3; sp = sorted(x[2] for x in v)[len(v) // 2] / 1e3
        print(f"  {name[:60]:60s} grid={grid:>8s} n={len(v):4d}  rows kernel {d:7.1f} us  finalize {fi:5.1f} us  rows start -> finalize end {sp:7.1f} us")
PY
grep -v amdgpu "$O/finito.log" | grep "^r=" | cut -c1-120; grep -v amdgpu "$O/proshi.log" | grep "r=" | cut -c1-120
