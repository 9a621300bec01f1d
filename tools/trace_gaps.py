#!/usr/bin/env python3
"""trace_gaps.py KERNEL_TRACE.csv [name-filter ...] -- per-kernel durations and the gaps between consecutive dispatches of a
rocprofv3 --kernel-trace CSV (Start/End timestamps in ns): where a launch-bound loop (Finito batches: rows kernel ->
finalize -> rows kernel ...) spends its time.  Prints a JSON summary; `--tail N` restricts to the last N dispatches."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"ciao::(\w+)<([^>]*)>", name)
    return f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else name[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tail = None
    for i, a in enumerate(sys.argv):
        if a == "--tail":
            tail = int(sys.argv[i + 1])
            args.remove(sys.argv[i + 1])
    rows = list(csv.DictReader(open(args[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if tail:
        rows = rows[-tail:]
    filt = args[1:]
    dur = defaultdict(list)
    gaps = defaultdict(list)
    prev = None
    for r in rows:
        n = short(r["Kernel_Name"])
        if filt and not any(f in n for f in filt):
            prev = None
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur[(n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))].append(e - s)
        if prev is not None:
            gaps[(prev[0], n)].append(s - prev[1])
        prev = (n, e)

    def stat(v):
        v = sorted(v)
        return {"n": len(v), "avg_us": sum(v) / len(v) / 1e3, "med_us": v[len(v) // 2] / 1e3, "min_us": v[0] / 1e3, "max_us": v[-1] / 1e3}
    out = {"durations": {f"{k[0]} grid={k[1]}": stat(v) for k, v in sorted(dur.items())},
           "gaps_prev_end_to_next_start": {f"{a} -> {b}": stat(v) for (a, b), v in sorted(gaps.items())}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
