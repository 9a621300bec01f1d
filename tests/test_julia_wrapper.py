"""Static check of the Julia host side against the C ABI.

The image has no `julia` binary, so ciaoalgorithms.jl_amd/julia/CIAOAlgorithmsAMD/src/CIAOAlgorithmsAMD.jl (the wrapper a
maintainer of the reference would ship) has never run.  What CAN be verified without Julia, and is here:
  * every `ccall((:ciao_x, libciao), Ret, (types...), ...)` in the module and in INTEGRATION.md names an exported symbol,
    has the arity of its declaration and, argument by argument, the C type the ctypes table (_lib.SIGNATURES, itself checked
    against the header and the built library in tests/test_abi.py) declares;
  * the Julia mirrors of the C structs have the size and field offsets gcc gives the header's structs;
  * the module binds every entry point of the drop-in surface it needs (no stale or missing names);
  * the wrapper checks the ABI version it was written against.
"""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "ciaoalgorithms.jl_amd", "julia", "CIAOAlgorithmsAMD", "src", "CIAOAlgorithmsAMD.jl")
MD = os.path.join(ROOT, "INTEGRATION.md")


def _balanced(text, start):
    """text[start] == '(' -> index just past its matching ')' (strings and comments are not nested in the ccall heads)."""
    depth = 0
    for i in range(start, len(text)):
        if text[i] in "({[":
            depth += 1
        elif text[i] in ")}]":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def ccalls(text):
    """[(symbol, return type, [argument types], number of actual arguments)] of every ccall in `text`."""
    found = []
    for m in re.finditer(r"ccall\(", text):
        end = _balanced(text, m.end() - 1)
        parts = _split_top(text[m.end():end - 1])
        head = re.match(r"\(\s*:(\w+)\s*,\s*[^)]+\)", parts[0])
        assert head, parts[0]
        if not head.group(1).startswith("ciao_"):      # a call into another library (RCCL) shown for context
            continue
        argt = parts[2].strip()
        assert argt.startswith("(") and argt.endswith(")"), argt
        types = _split_top(argt[1:-1])
        found.append((head.group(1), parts[1].strip(), types, len(parts) - 3))
    return found


def jl_kind(t):
    t = t.replace(" ", "")
    simple = {"Int32": "i32", "Int64": "i64", "UInt64": "u64", "Float64": "f64", "Cstring": "cstr", "Cint": "i32"}
    if t in simple:
        return simple[t]
    typed = {"Ref{CiaoProblem}": "ptr:problem", "Ref{CiaoProxDesc}": "ptr:prox", "Ref{CiaoSepQuad}": "ptr:sepquad",
             "Ref{CiaoShardTable}": "ptr:shards"}
    if t in typed:
        return typed[t]
    if t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    raise AssertionError(f"unmapped Julia type {t}")


def c_kind(t, L):
    if t is C.c_int32:
        return "i32"
    if t is C.c_int64:
        return "i64"
    if t is C.c_uint64:
        return "u64"
    if t is C.c_double:
        return "f64"
    if t is C.c_char_p:
        return "cstr"
    if t is C.c_void_p:
        return "ptr"
    if t is L.ALLREDUCE_FN:
        return "ptr"
    for struct, kind in ((L.Problem, "ptr:problem"), (L.ProxDesc, "ptr:prox"), (L.SepQuad, "ptr:sepquad"), (L.ShardTable, "ptr:shards")):
        if t is C.POINTER(struct):
            return kind
    if hasattr(t, "_type_") and isinstance(t._type_, type):   # POINTER(scalar) / POINTER(c_void_p)
        return "ptr"
    raise AssertionError(f"unmapped ctypes type {t}")


def _check(text, L, where):
    calls = ccalls(text)
    assert calls, f"no ccall found in {where}"
    for sym, ret, types, nact in calls:
        assert sym in L.SIGNATURES, f"{where}: ccall to :{sym}, which the library does not declare"
        res, args = L.SIGNATURES[sym]
        assert jl_kind(ret) == c_kind(res, L), f"{where}: :{sym} returns {ret}, C says {res}"
        assert len(types) == len(args) == nact, f"{where}: :{sym} has {len(types)} types / {nact} values, C declares {len(args)} parameters"
        for k, (jt, ct) in enumerate(zip(types, args)):
            jk, ck = jl_kind(jt), c_kind(ct, L)
            # an untyped Ptr{Cvoid} may stand for any pointer parameter (a NULL, a device buffer); typed Refs must match exactly
            ok = jk == ck or (jk == "ptr" and ck.startswith("ptr"))
            assert ok, f"{where}: :{sym} argument {k + 1} is {jt} ({jk}) but the C parameter is {ck}"
    return calls


def test_every_ccall_of_the_julia_module_matches_the_abi(ciao):
    calls = _check(open(JL).read(), ciao._lib, "CIAOAlgorithmsAMD.jl")
    bound = {c[0] for c in calls}
    # the module binds the whole drop-in surface except what only other hosts need
    not_needed = {"ciao_ctx_set_allreduce",      # a host callback: the Julia host hands RCCL to the library instead (set_rccl!)
                  "ciao_ctx_set_stream", "ciao_ctx_timing_enable", "ciao_ctx_timing_read",   # bench instrumentation
                  "ciao_svrg_inner",             # the inner cycle alone: Base.iterate always runs whole epochs
                  "ciao_lfinito_iterate",        # index-list form: LFinito's batches are always static row blocks (_blocks form)
                  "ciao_synth_normal", "ciao_synth_targets", "ciao_sample_batches", "ciao_sample_uniform", "ciao_peer_allreduce"}   # CIAO_BENCH_API section
    # (the Julia host draws with Julia's own RNG, as the reference does: the injected splitmix stream is the Python mirror's)
    missing = set(ciao._lib.SIGNATURES) - bound - not_needed
    assert not missing, f"the Julia module binds no ccall for {sorted(missing)}"
    assert not (bound & {"ciao_synth_normal", "ciao_synth_targets"}), "bench helpers are not part of the drop-in surface"


def test_every_ccall_shown_in_integration_md_matches_the_abi(ciao):
    _check(open(MD).read(), ciao._lib, "INTEGRATION.md")


def _jl_structs(text):
    out = {}
    for m in re.finditer(r"^struct (Ciao\w+)(.*?)^end", text, re.S | re.M):
        if "<:" in m.group(0).split("\n")[0]:
            continue
        fields = []
        for line in m.group(2).split("\n"):
            line = line.split("#")[0]
            for f in line.split(";"):
                f = f.strip()
                if "::" in f:
                    name, ty = f.split("::")
                    fields.append((name.strip(), ty.strip()))
        out[m.group(1)] = fields
    return out


def _jl_layout(fields):
    size_al = {"Int32": (4, 4), "Int64": (8, 8), "UInt64": (8, 8), "Float64": (8, 8), "Ptr{Cvoid}": (8, 8)}
    off, offsets, maxal = 0, {}, 1
    for name, ty in fields:
        m = re.match(r"NTuple\{(\d+),\s*(.+)\}", ty)
        n, base = (int(m.group(1)), m.group(2).strip()) if m else (1, ty)
        sz, al = size_al[base]
        off = (off + al - 1) // al * al
        offsets[name] = off
        off += sz * n
        maxal = max(maxal, al)
    return (off + maxal - 1) // maxal * maxal, offsets


def test_julia_struct_mirrors_have_the_c_layout(ciao):
    L = ciao._lib
    structs = _jl_structs(open(JL).read())
    for jl_name, ct in (("CiaoProblem", L.Problem), ("CiaoProxDesc", L.ProxDesc), ("CiaoSepQuad", L.SepQuad), ("CiaoShardTable", L.ShardTable)):
        assert jl_name in structs, f"{jl_name} is not declared in the Julia module"
        size, offsets = _jl_layout(structs[jl_name])
        assert size == C.sizeof(ct), f"{jl_name}: Julia layout is {size} bytes, C struct {C.sizeof(ct)}"
        assert [n for n, _ in structs[jl_name]] == [f[0] for f in ct._fields_], f"{jl_name}: field order differs"
        for name, _ in structs[jl_name]:
            assert offsets[name] == getattr(ct, name).offset, f"{jl_name}.{name}: offset {offsets[name]} vs C {getattr(ct, name).offset}"


def test_ctypes_struct_layouts_match_gcc(ciao, tmp_path):
    """... and the ctypes mirrors those layouts were compared with are what gcc makes of the header (sepquad, shard table;
    tests/test_abi.py does problem and prox_desc)."""
    import subprocess
    src = tmp_path / "l.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ciao_hip.h"\nint main(void){\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ciao_sepquad), offsetof(ciao_sepquad,N), offsetof(ciao_sepquad,ld),'
                   ' offsetof(ciao_sepquad,Q), offsetof(ciao_sepquad,eta), offsetof(ciao_sepquad,hi), sizeof(ciao_shard_table),'
                   ' offsetof(ciao_shard_table,row0), offsetof(ciao_shard_table,A), offsetof(ciao_shard_table,table));\nreturn 0;}\n')
    exe = tmp_path / "l"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    S, T = ciao._lib.SepQuad, ciao._lib.ShardTable
    assert got == [C.sizeof(S), S.N.offset, S.ld.offset, S.Q.offset, S.eta.offset, S.hi.offset, C.sizeof(T), T.row0.offset, T.A.offset,
                   T.table.offset]


def test_the_wrapper_checks_the_abi_version_it_binds(ciao):
    text = open(JL).read()
    m = re.search(r"const CIAO_ABI_VERSION = Int32\((\d+)\)", text)
    assert m and int(m.group(1)) == ciao._lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "ciao_hip.h")).read()
    assert int(re.search(r"#define CIAO_ABI_VERSION (\d+)", hdr).group(1)) == ciao._lib.ABI_VERSION
    assert ":ciao_abi_version" in text and "CIAO_ABI_VERSION ||" in text.replace("== CIAO_ABI_VERSION ||", "CIAO_ABI_VERSION ||")
