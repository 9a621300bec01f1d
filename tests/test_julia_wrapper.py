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


# ======================================================================================================================
# The Julia wrapper stays level with the Python mirror: every constructor keyword, every ABI call a solver's
# iterable makes and every host-side feature of solvers.py has its counterpart in the .jl -- or is listed below with the reason.
# The tables are the contract: a feature added on one side only fails here.
# ======================================================================================================================
SOLVERS_PY = os.path.join(ROOT, "ciaoalgorithms.jl_amd", "solvers.py")
DEVICE_PY = os.path.join(ROOT, "ciaoalgorithms.jl_amd", "device.py")

# the ABI entry points each solver's iterable reaches -- on BOTH sides (parsed from solvers.py + device.py, and from the .jl section)
SOLVER_ABI = {
    "SVRG": {"ciao_svrg_init", "ciao_svrg_iterate"},
    "SAGA": {"ciao_saga_init", "ciao_saga_steps"},
    "Finito": {"ciao_hat_gamma", "ciao_finito_init", "ciao_finito_steps", "ciao_finito_steps_blocks", "ciao_lfinito_init", "ciao_lfinito_iterate_blocks"},
    "adaptive": {"ciao_afinito_init", "ciao_afinito_probe", "ciao_afinito_steps"},
    "Proshi": {"ciao_proshi_init", "ciao_proshi_steps", "ciao_proshi_steps_blocks"},
}
# ... and what only ONE side calls there, with the reason
JULIA_ONLY_ABI = {
    "SVRG": {"ciao_svrg_inner", "ciao_svrg_epoch_tail", "ciao_svrg_epoch_tail_multi", "ciao_full_gradient_multi"},   # iterate_together!: in solvers.py these live in solve_together (below), outside the iterable's class
    "Proshi": {"ciao_proshi_solution"},   # `solution(state)`: the Python mirror calls it from its state class
}
PYTHON_ONLY_ABI = {
    "adaptive": {"ciao_ctx_synchronize"},   # the Python iterable reads the step counters back itself; the .jl does it in its generic `synchronize`
}
# constructor keywords: Julia spelling -> accepted Python spellings (the Python mirror takes the Greek names AND ascii aliases)
KEYWORDS = {
    "SVRG": ["γ", "maxit", "verbose", "freq", "m", "plus"],
    "SAGA": ["γ", "maxit", "verbose", "freq", "SAG_flag"],
    "Finito": ["γ", "sweeping", "LFinito", "adaptive", "minibatch", "maxit", "verbose", "freq", "α", "tol", "tol_b"],
    "Proshi": ["γ", "sweeping", "minibatch", "maxit", "verbose", "freq", "α"],
}
# host-side features: (what, a pattern solvers.py / operators.py must contain, a pattern the .jl must contain)
FEATURES = [
    ("feature padding switch", r"PAD_FEATURES = True", r"const PAD_FEATURES = Ref\(true\)"),
    ("padded packing of F", r"def pack_F\(.*pad_to", r"function pack_F\(.*pad_to::Int"),
    ("padded box bounds of g", r"def pack_g\(.*pad_to", r"function pack_g\(.*pad_to::Int"),
    ("state vectors as length-d views of padded buffers", r"base\[:self\.d\]", r"statevec\("),
    ("adaptive Finito keeps its d", r"_pads_features", r"adaptive Finito keeps its d"),
    ("K solves in lockstep", r"def solve_together\(iterables, maxit, one_pass=False\)", r"function solve_together\(iters::Vector, maxit::Int; one_pass::Bool = false"),
    ("K full passes as one", r"svrg_epoch_tail_multi|full_gradient_multi", r"ciao_svrg_epoch_tail_multi"),
    ("explicit host route", r"fallback", r"fallback = nothing"),
    ("unpackable operators are an error", r"UnpackableOperator", r"struct UnpackableOperator"),
    ("objective monitor", r"_monitor_on", r"set_monitor!"),
    ("SVRG row-dot reuse only when the state is vouched for", r"reuse_rowdots", r"TRUST_SVRG_STATE"),
    ("SVRG++ caps maxit at 25", r"maxit = 25|min\(.*25", r"maxit = 25"),
    ("chain batches", r"chain_batch", r"function chain_batch\(f\)"),
    ("shard table", r"set_shards", r"set_shards!"),
    ("peer mailboxes", r"set_peers|PeerGroup", r"peer_mailbox_create"),
    ("ABI version check", r"", r"CIAO_ABI_VERSION \|\|"),
]
# what the Python mirror has and the wrapper deliberately does not (and why)
PYTHON_ONLY_FEATURES = {
    "stream=": "explicit counter-based index streams: a Julia user keeps the reference's own RNG calls (rand / sample / randperm in the wrapper)",
    "ctx=": "the wrapper has one default context per process (AMDGPU.jl's device and stream)",
    "backend=": "test hook of the Python mirror (forces the host route)",
    "stop=": "IterationTools.halt composes with the wrapper's iterables in Julia itself (SVRG.jl:70)",
}


def _py_method_abi():
    """device.Context method -> the ABI symbols it reaches (through other methods too)"""
    dev = open(DEVICE_PY).read()
    meth = {}
    for m in re.finditer(r"\n    def (\w+)\(self[^\n]*\n((?:(?!\n    def ).)*)", dev, re.S):
        meth[m.group(1)] = (set(re.findall(r"self\.lib\.(ciao_\w+)", m.group(2))), set(re.findall(r"self\.(\w+)\(", m.group(2))))

    def closure(name, seen):
        if name in seen or name not in meth:
            return set()
        seen.add(name)
        out = set(meth[name][0])
        for c in meth[name][1]:
            out |= closure(c, seen)
        return out
    return {k: closure(k, set()) for k in meth}


def _py_solver_abi():
    abi = _py_method_abi()
    sol = open(SOLVERS_PY).read()
    per_class = {}
    for m in re.finditer(r"\nclass (\w+)\([^\n]*\n((?:(?!\nclass |\ndef ).)*)", sol, re.S):
        calls = set(re.findall(r"(?:self\.ctx|ctx|it\.ctx|c)\.(\w+)\(", m.group(2)))
        per_class[m.group(1)] = set().union(*[abi.get(c, set()) for c in calls]) if calls else set()
    return {"SVRG": per_class["SVRG_basic_iterable"], "SAGA": per_class["SAGA_basic_iterable"],
            "Finito": per_class["FINITO_basic_iterable"] | per_class["FINITO_LFinito_iterable"],
            "adaptive": per_class["FINITO_adaptive_iterable"], "Proshi": per_class["Proshi_basic_iterable"]}


def _jl_solver_abi():
    jl = open(JL).read()
    secs = re.split(r"\n# =+\n# ([^\n]+)\n# =+\n", jl)
    out = {}
    for i in range(1, len(secs), 2):
        title, body = secs[i], secs[i + 1]
        key = ("adaptive" if title.startswith("adaptive") else "SVRG" if title.startswith("SVRG") else "SAGA" if title.startswith("SAGA")
               else "Finito" if title.startswith("Finito") else "Proshi" if title.startswith("ProShI") else None)
        if key:
            out[key] = set(re.findall(r"ccall\(\(:(ciao_\w+)", body))
    return out


def test_every_abi_call_of_a_python_solver_has_its_julia_counterpart():
    py, jl = _py_solver_abi(), _jl_solver_abi()
    for solver, both in SOLVER_ABI.items():
        assert py[solver] - PYTHON_ONLY_ABI.get(solver, set()) == both, (solver, "solvers.py reaches", sorted(py[solver]), "the table says", sorted(both))
        assert jl[solver] - JULIA_ONLY_ABI.get(solver, set()) == both, (solver, "the .jl calls", sorted(jl[solver]), "the table says", sorted(both))


def test_every_constructor_keyword_exists_on_both_sides():
    jl, py = open(JL).read(), open(SOLVERS_PY).read()
    for solver, kws in KEYWORDS.items():
        mj = re.search(r"function " + solver + r"\{R\}\(;(.*?)\) where", jl, re.S)
        jkw = re.findall(r"(?:^\s*|[;,]\s*)(\w+)(?:::|\s*=)", mj.group(1).replace("\n", " "))
        mp = re.search(r"class " + solver + r"\(_Solver\):.*?def __init__\(self, R=np\.float64, \*,(.*?)\):\n", py, re.S)
        pkw = re.findall(r"(\w+)\s*=", mp.group(1))
        assert set(kws) <= set(jkw), (solver, "Julia constructor lacks", sorted(set(kws) - set(jkw)))
        assert set(kws) <= set(pkw), (solver, "Python constructor lacks", sorted(set(kws) - set(pkw)))
        alias = {"gamma", "alpha"}   # ascii spellings of γ, α the Python mirror also takes
        assert set(pkw) - alias == set(kws), (solver, "one-sided Python keywords", sorted(set(pkw) - alias - set(kws)))
        assert set(jkw) == set(kws), (solver, "one-sided Julia keywords", sorted(set(jkw) - set(kws)))


def test_every_host_side_feature_exists_on_both_sides():
    jl = open(JL).read()
    py = open(SOLVERS_PY).read() + open(os.path.join(ROOT, "ciaoalgorithms.jl_amd", "operators.py")).read() + open(DEVICE_PY).read()
    missing = []
    for what, ppat, jpat in FEATURES:
        if ppat and not re.search(ppat, py):
            missing.append(f"python lacks: {what}")
        if not re.search(jpat, jl):
            missing.append(f"julia lacks: {what}")
    assert not missing, missing
    # the call keywords of the Python functors: all but the documented Python-only ones appear in the wrapper's functors
    for m in re.finditer(r"def _iterable\(self, x0,(.*?)\):", open(SOLVERS_PY).read(), re.S):
        for kw in re.findall(r"(\w+)=", m.group(1)):
            if kw + "=" in PYTHON_ONLY_FEATURES or kw in ("mu",):
                continue
            if kw == "shards":     # the wrapper sets a shard table on the context (set_shards!), not per call
                assert "set_shards!" in jl
                continue
            assert re.search(r"\b" + kw + r"\b\s*=", jl), f"the wrapper's functors lack the keyword `{kw}`"
