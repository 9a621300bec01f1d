#!/usr/bin/env python3
"""Kernels the built library HOLDS against kernels a traced run LAUNCHED.

    python tools/reached_kernels.py launched.tsv [libciao_hip.so]

`launched.tsv` = "<launches>\\t<kernel symbol>" per line, made by tools/exp/reached_kernels.sh from a `rocprofv3 --kernel-trace` of the
GPU test suite; the library's kernels come from its code objects (tools/kernel_meta.py).  Prints per kernel family how many
instantiations the library holds and how many were launched, then every instantiation that was NOT -- the list to prune, or to cover
with a test."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_meta  # noqa: E402


def norm(name):
    """One spelling for a demangled kernel symbol: no argument list, no spaces, no '.kd', no 'void'."""
    name = name.strip()
    if name.endswith(".kd"):
        name = name[:-3]
    name = re.sub(r"^void\s+", "", name)
    depth = 0
    for i, c in enumerate(name):   # cut the argument list: the first '(' outside template brackets
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            name = name[:i]
            break
    return name.replace(" ", "")


def family(n):
    return re.sub(r"^ciao::", "", n.split("<")[0])


def main():
    launched_file = sys.argv[1]
    lib = sys.argv[2] if len(sys.argv) > 2 else kernel_meta.DEFAULT_LIB
    held = {norm(k["name"]): k for k in kernel_meta.library_kernels(lib)}
    launched = {}
    for line in open(launched_file):
        cnt, name = line.rstrip("\n").split("\t", 1)
        launched[norm(name)] = launched.get(norm(name), 0) + int(cnt)
    ours = {n: c for n, c in launched.items() if n in held}
    foreign = sorted(n for n in launched if n not in held)
    per = collections.defaultdict(lambda: [0, 0])
    for n in held:
        per[family(n)][0] += 1
        per[family(n)][1] += n in ours
    print("# %s: %d kernels held, %d launched by the traced run (%d launches); %d launched kernels are not the library's (torch / rocm)"
          % (os.path.basename(lib), len(held), len(ours), sum(ours.values()), len(foreign)))
    print("%-28s %6s %9s" % ("family", "held", "launched"))
    for f, (h, l) in sorted(per.items(), key=lambda x: (x[1][1] - x[1][0], x[0])):
        print("%-28s %6d %9d%s" % (f, h, l, "" if h == l else "   <- %d not launched" % (h - l)))
    missing = sorted(n for n in held if n not in ours)
    print("# not launched (%d):" % len(missing))
    for n in missing:
        print(n)


if __name__ == "__main__":
    main()
