#!/usr/bin/env python3
"""Print VGPR / AGPR / SGPR / scratch / LDS / occupancy per kernel of one .hip unit (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':90s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    print(f"{r['name']:90s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
