#!/bin/bash
# Round 5, VERDICT r4 item 6: counter passes of the mid-size batch kernels (Finito r = 4096 at d = 4096 fp32: rows_split_kernel; ProShI
# d = 1024 fp64: proshi_vec_kernel) -- L2 busy and memory-credit stalls, waves, LDS bank conflicts.  Separate --pmc passes, kernel trace only.
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out/midsize_pmc"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_BUSY_sum TCC_CYCLE_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/finito_$i" -o f -- python3 "$R/tools/finito_batch_time.py" 4096 > "$O/finito_$i.log" 2>&1; echo "finito pass $i ($set) rc=$?"
  CIAO_D=1024 timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/proshi_$i" -o p -- python3 "$R/tools/proshi_batch_time.py" > "$O/proshi_$i.log" 2>&1; echo "proshi pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for what, pat in (("finito", "rows_split_kernel"), ("proshi", "proshi_vec_kernel<double, false")):
    for f in sorted(glob.glob("$O/%s_*/**/*counter_collection.csv" % what, recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        by = collections.defaultdict(list)
        for r in rows:
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in by.items():
            v.sort()
            out.setdefault(what, {})[c] = {"median_per_dispatch": v[len(v) // 2], "max_per_dispatch": v[-1], "dispatches": len(v)}
print(json.dumps(out, indent=1))
json.dump(out, open("$O/summary.json", "w"), indent=1)
PY
