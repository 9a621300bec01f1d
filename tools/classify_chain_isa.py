#!/usr/bin/env python3
"""Opcode histogram of ONE steady-state step of a chain kernel, from the compiler's own assembly (hipcc -S of the unit): the
instructions between two consecutive s_barrier of the check-free, box-free step group.
usage: classify_chain_isa.py <unit.hip> <mangled-kernel-substring>   e.g. chain_dma3_f64.hip 'IdLi2ELi4ELi0ELb0ELi256ELb0E'"""
import collections, os, re, subprocess, sys, tempfile
unit, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ciaoalgorithms.jl_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "u.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-DCIAO_BUILD_FLAGS=\"\"",
                    "--offload-device-only", "-S", os.path.join(csrc, unit), "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(":") or (l.startswith("_Z") and pat in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
bars = [i for i, l in enumerate(body) if re.match(r"\s+s_barrier", l)]
# steady-state steps: the most common distance between consecutive barriers, taken from the LAST run of equally spaced ones
# (the check-free group without the IndBox clamp is emitted last)
steps = [(a, b) for a, b in zip(bars, bars[1:])]
def ops(a, b):
    return [l.split()[0] for l in body[a + 1:b + 1] if re.match(r"\s+[a-z]", l) and not l.strip().startswith(";")]
hist = collections.Counter(tuple(sorted(collections.Counter(ops(a, b)).items())) for a, b in steps)
best = max(hist.items(), key=lambda kv: kv[1])[0]
tot = collections.Counter(dict(best))
cls = collections.OrderedDict()
def take(name, pred):
    n = sum(v for k, v in tot.items() if pred(k))
    cls[name] = n
take("VALU", lambda k: k.startswith("v_"))
take("SALU (incl. s_nop / s_waitcnt / branches)", lambda k: k.startswith("s_"))
take("LDS", lambda k: k.startswith("ds_"))
take("VMEM (LDS-DMA / loads / stores)", lambda k: k.startswith("global_") or k.startswith("buffer_") or k.startswith("flat_"))
print(f"# one steady-state step of {pat} in {unit}: {sum(tot.values())} instructions")
for k, v in cls.items():
    print(f"{v:4d}  {k}")
print("# opcodes")
for k, v in sorted(tot.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{v:4d}  {k}")
