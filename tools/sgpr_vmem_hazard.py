#!/usr/bin/env python3
"""Static check of one gfx9 hazard the compiler cannot see through inline asm:

    "VALU writes SGPR  ->  VMEM reads that SGPR: 5 wait states"   (CDNA3 / CDNA4 ISA, manually inserted wait states)

hipcc's hazard recognizer inserts the s_nops for the vector-memory instructions IT emits; the operands of an inline-asm
`global_load_lds_dwordx4 v, s[a:b]` / `global_store_dwordx4 v, v, s[a:b]` are opaque to it.  A row or table-row base that reaches
such an asm from a v_readfirstlane_b32 (a wave-uniform pointer read from LDS) or from a v_readlane_b32 (the RESTORE of a spilled
scalar register -- hipcc puts it right in front of the use) less than five wait states earlier makes the memory instruction read
the STALE register pair: a wild 64-bit address.

    python tools/sgpr_vmem_hazard.py file.s [...]            compiler assembly (--save-temps)
    python tools/sgpr_vmem_hazard.py --lib [libciao_hip.so]  disassembles every gfx950 code object of the built library

Prints every vector-memory instruction with a scalar base whose base register (either half) was written by a VALU instruction
fewer than 5 wait states before it; exit status 1 if there is one.  An instruction is one wait state, `s_nop N` is N + 1; a label or
a branch ends the look-back conservatively (the distance along another path is unknown: counted as satisfied only if the
straight-line distance already is)."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
NEED = 5
VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)\w+")
SBASE = re.compile(r"\bs\[(\d+):(\d+)\]")
VALU_SGPR_DEF = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)\b")
VALU_SGPR_DEF2 = re.compile(r"^\s*v_(cmp|cmpx|add_co|sub_co|subrev_co|addc_co|subb_co|div_scale|mad_u64_u32|mad_i64_i32)\w*\s+(?:v\S+,\s*)?s\[(\d+):(\d+)\]")


def instructions(lines):
    """(text, lineno) of machine instructions; labels and directives are kept as markers ('LABEL', n)"""
    for n, raw in enumerate(lines, 1):
        t = raw.split(";")[0].split("//")[0].rstrip()
        if not t.strip():
            continue
        s = t.strip()
        if s.startswith(".") and not s.endswith(":"):
            continue
        if s.endswith(":"):
            yield ("LABEL", n)
            continue
        if re.match(r"^[0-9a-fA-F]+ <", s):     # objdump symbol line
            yield ("LABEL", n)
            continue
        yield (s, n)


def check(lines, name):
    ins = list(instructions(lines))
    bad = []
    for i, (t, n) in enumerate(ins):
        if t == "LABEL" or not VMEM.match(t):
            continue
        regs = set()
        for m in SBASE.finditer(t):
            a, b = int(m.group(1)), int(m.group(2))
            if b - a == 1:          # a 64-bit scalar base (128-bit descriptors of buffer_ ops are checked as well, below)
                regs.update((a, b))
            elif t.lstrip().startswith("buffer_"):
                regs.update(range(a, b + 1))
        if not regs:
            continue
        ws, j = 0, i - 1
        while j >= 0 and ws < NEED:
            u, un = ins[j]
            if u == "LABEL" or u.startswith("s_cbranch") or u.startswith("s_branch"):
                break
            m = VALU_SGPR_DEF.match(u)
            hit = None
            if m and int(m.group(2)) in regs:
                hit = int(m.group(2))
            m2 = VALU_SGPR_DEF2.match(u)
            if m2 and regs & set(range(int(m2.group(2)), int(m2.group(3)) + 1)):
                hit = int(m2.group(2))
            if hit is not None:
                bad.append((name, n, t, un, u, ws))
                break
            # a scalar instruction that (re)defines the register -- e.g. the s_mov_b64 copy of the base inside the chain kernels' asm --
            # is the definition the memory instruction sees: SALU-written SGPRs have no such hazard
            md = re.match(r"^s_\w+\s+s(?:\[(\d+):(\d+)\]|(\d+)\b)", u)
            if md:
                lo_ = int(md.group(1) if md.group(1) is not None else md.group(3))
                hi_ = int(md.group(2) if md.group(2) is not None else md.group(3))
                regs -= set(range(lo_, hi_ + 1))
                if not regs:
                    break
            mn = re.match(r"^s_nop\s+(\d+)", u)
            ws += (int(mn.group(1)) + 1) if mn else 1
            j -= 1
    return bad


def main(argv):
    texts = []
    if "--lib" in argv:
        import kernel_meta
        rest = [a for a in argv if not a.startswith("--")]
        lib = rest[0] if rest else kernel_meta.DEFAULT_LIB
        with tempfile.TemporaryDirectory(prefix="ciao_hz_") as wd:
            for elf in kernel_meta.code_objects(lib, wd):
                out = subprocess.run([f"{kernel_meta.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", elf],
                                     capture_output=True, text=True, check=True).stdout
                texts.append((os.path.basename(elf), out.splitlines()))
    else:
        for p in argv:
            texts.append((p, open(p).read().splitlines()))
    bad, nvmem = [], 0
    for name, lines in texts:
        nvmem += sum(1 for l in lines if VMEM.match(l) and SBASE.search(l))
        bad += check(lines, name)
    for name, n, t, un, u, ws in bad:
        print(f"{name}:{n}: `{t}` reads a scalar base that `{u}` (line {un}) wrote {ws} wait state(s) earlier (need {NEED})")
    print(f"# {len(texts)} file(s), {nvmem} vector-memory instructions with a scalar base, {len(bad)} inside the hazard window")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
