#!/usr/bin/env python3
"""pmc_sweep_traffic.py KEY FETCH_counter_collection.csv WRITE_counter_collection.csv [existing.json]

HBM traffic per launch of the bench's dominant kernel from two separate `rocprofv3 --pmc` passes over the same bench.py
command (FETCH_SIZE and WRITE_SIZE cannot share a pass with the timed run), corrected as MI355X_MICROARCH.md "HBM"
prescribes: the counters are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming
read, so  traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024.   The dominant kernel is the one with the largest summed
FETCH_SIZE.  Prints the updated JSON (bench.py reads profiles/pmc_traffic.json: value under KEY, plus `kernel` per key and
`measured_at`)."""
import csv
import json
import re
import sys
import time
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


key, fpath, wpath = sys.argv[1:4]
out = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
f, w = per_kernel(fpath, "FETCH_SIZE"), per_kernel(wpath, "WRITE_SIZE")
name = max(f, key=lambda k: sum(f[k]))
fv, wv = sum(f[name]) / len(f[name]), sum(w[name]) / len(w[name])
m = re.search(r"ciao::(\w+)<([^>]*)>", name)
short = f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else name
out[key] = 2 * fv * 1024 + wv * 1024
out.setdefault("kernels", {})[key] = {"rocprof_name": short, "launches": len(f[name]), "FETCH_SIZE_KiB": fv, "WRITE_SIZE_KiB": wv}
out["measured_at"] = time.strftime("%Y-%m-%d", time.gmtime())
print(json.dumps(out, indent=1))
