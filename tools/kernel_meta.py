#!/usr/bin/env python3
"""Per-kernel resources of the BUILT library, read from its gfx950 code objects (no recompilation).

    python tools/kernel_meta.py [libciao_hip.so] [--spills] [--json]

The library's `.hip_fatbin` section is a sequence of clang offload bundles (one per translation unit); each bundle's gfx950
entry is an ELF whose NT_AMDGPU_METADATA note (msgpack, printed as YAML by llvm-readelf) lists, per kernel, the register
counts, `.private_segment_fixed_size` (scratch bytes per lane), `.sgpr_spill_count` / `.vgpr_spill_count` and the LDS size.
tests/test_kernel_resources.py gates on this: a kernel of the hot list with scratch fails the CPU suite.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "..", "ciaoalgorithms.jl_amd", "libciao_hip.so")


def code_objects(lib_path, workdir):
    """Write every gfx950 code object of `lib_path` into `workdir`; return their paths."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(workdir, "stripped.tmp")],
                   check=True, capture_output=True)
    data = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        p = data.find(MAGIC, pos)
        if p < 0:
            break
        (nb,) = struct.unpack_from("<Q", data, p + 24)
        off = p + 32
        for _ in range(nb):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and s > 0:
                path = os.path.join(workdir, f"co_{len(out)}.elf")
                with open(path, "wb") as f:
                    f.write(data[p + o:p + o + s])
                out.append(path)
        pos = p + 24
    return out


_KEYS = {"name": ".name", "scratch": ".private_segment_fixed_size", "vgpr": ".vgpr_count", "agpr": ".agpr_count",
         "sgpr": ".sgpr_count", "vgpr_spill": ".vgpr_spill_count", "sgpr_spill": ".sgpr_spill_count",
         "lds": ".group_segment_fixed_size", "symbol": ".symbol"}


def kernels_of(elf):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
    rows, cur = [], None
    in_kernels = False
    for line in txt.splitlines():
        if line.startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if in_kernels and re.match(r"^amdhsa\.", line):
            in_kernels = False
        if not in_kernels:
            continue
        m = re.match(r"^  - (\.\w+):\s*(.*)$", line)
        if m:                               # first key of a new kernel record
            cur = {}
            rows.append(cur)
            k, v = m.group(1), m.group(2)
        else:
            m = re.match(r"^    (\.\w+):\s*(.*)$", line)
            if not m or cur is None:
                continue
            k, v = m.group(1), m.group(2)
        for short, key in _KEYS.items():
            if k == key:
                cur[short] = v.strip().strip("'")
    for r in rows:
        for k in ("scratch", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "lds"):
            r[k] = int(r.get(k, 0) or 0)
    return [r for r in rows if "name" in r]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def library_kernels(lib_path=DEFAULT_LIB):
    """[{name (demangled), scratch, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, lds}] for every kernel of the built library."""
    with tempfile.TemporaryDirectory(prefix="ciao_co_") as wd:
        rows = []
        for elf in code_objects(lib_path, wd):
            rows += kernels_of(elf)
    dem = demangle([r["name"] for r in rows])
    for r, dn in zip(rows, dem):
        r["mangled"], r["name"] = r["name"], dn
    return rows


def main(argv):
    args = [a for a in argv if not a.startswith("--")]
    rows = library_kernels(args[0] if args else DEFAULT_LIB)
    if "--spills" in argv:
        rows = [r for r in rows if r["scratch"] or r["vgpr_spill"]]
    if "--json" in argv:
        json.dump(rows, sys.stdout, indent=1)
        return
    print(f"{'kernel':110s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'vspill':>6s} {'sspill':>6s} {'LDS':>7s}")
    for r in sorted(rows, key=lambda r: r["name"]):
        print(f"{r['name'][:110]:110s} {r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['scratch']:8d} {r['vgpr_spill']:6d} "
              f"{r['sgpr_spill']:6d} {r['lds']:7d}")
    print(f"# {len(rows)} kernels")


if __name__ == "__main__":
    main(sys.argv[1:])
