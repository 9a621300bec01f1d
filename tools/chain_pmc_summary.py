#!/usr/bin/env python3
"""Sums the counter_collection CSVs of tools/chain_pmc.sh per chain kernel and divides by waves x steps.  Steps per kernel
are what tools/chain_time.py (402 000 per variant: 2 000 warm-up + 400 000) and tools/saga_time.py (3 x 1 002 000) run."""
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            if "chain_" not in name:
                continue
            key = name.split("(")[0]
            acc.setdefault(key, {}).setdefault(row["Counter_Name"], 0.0)
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
out = {"command": "tools/chain_pmc.sh: rocprofv3 --pmc <four counters> --kernel-trace -- python3 tools/chain_time.py | tools/saga_time.py, two passes",
       "kernels": {}}
for key, c in sorted(acc.items()):
    ws = "chain_ws_kernel" in key
    steps = 3 * 1_002_000 if ("chain_ws_kernel" in key or ", 1, 1, 1," in key.replace("<float, 1, 1, 1", ", 1, 1, 1,")) and "float" in key and ("ws" in key) else 402_000
    if "chain_dma_kernel" in key and "float, 1, 1, 1" in key:
        steps = 3 * 1_002_000
    waves = 7 if ws else 4
    per = {k: v / steps for k, v in c.items()}
    entry = {"counters": {k: int(v) for k, v in c.items()}, "steps": steps, "waves_in_the_workgroup": waves,
             "per_step_whole_workgroup": {k: round(v, 2) for k, v in per.items()}}
    if "SQ_WAVE_CYCLES" in per:
        entry["cycles_per_step_per_wave"] = round(per["SQ_WAVE_CYCLES"] * 4 / waves, 1)     # the counter is in units of four cycles
    if not ws:
        entry["per_wave_step"] = {k.replace("SQ_INSTS_", ""): round(v / waves, 1) for k, v in per.items() if k.startswith("SQ_INSTS_")}
    out["kernels"][key] = entry
print(json.dumps(out, indent=1))
