#!/usr/bin/env python3
"""Spill summary of one or more built objects / libraries (tools/kernel_meta.py rows): python tools/obj_spills.py a.o b.o [--all]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_meta

def main(argv):
    show_all = "--all" in argv
    for path in [a for a in argv if not a.startswith("--")]:
        rows = kernel_meta.library_kernels(path)
        bad = [r for r in rows if r["sgpr_spill"] or r["vgpr_spill"] or r["scratch"]]
        print(f"== {path}: {len(rows)} kernels, {len(bad)} spilling; max sgpr_spill {max([r['sgpr_spill'] for r in rows] + [0])}, "
              f"max vgpr_spill {max([r['vgpr_spill'] for r in rows] + [0])}, max scratch {max([r['scratch'] for r in rows] + [0])}")
        for r in sorted(rows if show_all else bad, key=lambda r: r["name"]):
            print(f"  {r['name'][5:105]:100s} v{r['vgpr']:4d} a{r['agpr']:4d} s{r['sgpr']:4d} scr{r['scratch']:5d} vs{r['vgpr_spill']:4d} ss{r['sgpr_spill']:4d}")

if __name__ == "__main__":
    main(sys.argv[1:])
