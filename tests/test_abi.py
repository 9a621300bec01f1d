"""CPU-side checks of the drop-in boundary: libciao_hip.so loads and exports every symbol include/ciao_hip.h declares,
the ctypes prototype table covers exactly that set, the ctypes struct layouts match the C structs, and -- with no GPU in
this container -- creating a context fails LOUDLY (there is no CPU fallback on the product path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ciao_hip.h")


def _symbols(src):
    return sorted(set(re.findall(r"CIAO_API\s+[\w\s\*]+?\b(ciao_\w+)\s*\(", src)))


def declared_symbols():
    """Every entry point the header declares, the CIAO_BENCH_API section included."""
    return _symbols(open(HEADER).read())


def surface_symbols():
    """The drop-in surface only: what the header shows WITHOUT CIAO_BENCH_API."""
    src = open(HEADER).read()
    i0, i1 = src.index("#ifdef CIAO_BENCH_API"), src.index("#endif /* CIAO_BENCH_API */")
    return _symbols(src[:i0] + src[i1:])


def test_bench_helpers_are_outside_the_drop_in_surface():
    """ciao_synth_* and ciao_sample_* serve bench.py / the tests / the host mirror's sampler, not the reference's path:
    a binding that includes the header as is does not see them."""
    extra = sorted(set(declared_symbols()) - set(surface_symbols()))
    assert extra == ["ciao_peer_allreduce", "ciao_sample_batches", "ciao_sample_uniform", "ciao_synth_normal", "ciao_synth_targets"]
    for name in surface_symbols():
        assert not name.startswith("ciao_synth") and not name.startswith("ciao_sample")


def test_product_library_is_a_clean_build(ciao):
    """Experiment builds (tools/exp_build.sh: timing macros, some with WRONG results) report their flags; the product
    library must report none."""
    lib = ciao._lib.load()
    assert "CIAO_HIP_LIB" not in os.environ, "tests must run against the product library"
    assert lib.ciao_build_flags() == b""


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for must in ("ciao_ctx_create", "ciao_gradient", "ciao_prox", "ciao_full_gradient", "ciao_proxgrad_step", "ciao_svrg_init",
                 "ciao_svrg_inner", "ciao_svrg_iterate", "ciao_saga_init", "ciao_saga_steps", "ciao_finito_init",
                 "ciao_finito_steps", "ciao_lfinito_init", "ciao_lfinito_iterate", "ciao_ctx_set_allreduce", "ciao_last_error"):
        assert must in names
    assert len(names) >= 28


def test_library_exports_every_declared_symbol(ciao):
    lib = ciao._lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/ciao_hip.h but not exported by libciao_hip.so"
    assert lib.ciao_abi_version() == 3


def test_ctypes_table_matches_header(ciao):
    assert sorted(ciao._lib.SIGNATURES) == declared_symbols()


def test_no_unexpected_exports(ciao):
    """-fvisibility=hidden: only the C ABI is visible (no C++ symbols leak across the boundary)."""
    out = subprocess.run(["nm", "-D", "--defined-only", ciao._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if " T " in ln]
    assert sorted(syms) == declared_symbols()


def test_struct_layouts_match_the_header(ciao, tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirrors."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ciao_hip.h"\nint main(void){\n'
                   'printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ciao_problem), offsetof(ciao_problem,dtype),'
                   ' offsetof(ciao_problem,N), offsetof(ciao_problem,d), offsetof(ciao_problem,ld), offsetof(ciao_problem,N_total),'
                   ' offsetof(ciao_problem,A), offsetof(ciao_problem,b), offsetof(ciao_problem,lam), sizeof(ciao_prox_desc));\n'
                   'printf("%zu %zu %zu %zu %zu\\n", offsetof(ciao_prox_desc,lam), offsetof(ciao_prox_desc,lo), offsetof(ciao_prox_desc,hi),'
                   ' offsetof(ciao_prox_desc,lo_vec), offsetof(ciao_prox_desc,hi_vec));\nreturn 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    a, b = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")[:2]
    P, G = ciao._lib.Problem, ciao._lib.ProxDesc
    assert [int(v) for v in a.split()] == [C.sizeof(P), P.dtype.offset, P.N.offset, P.d.offset, P.ld.offset, P.N_total.offset,
                                           P.A.offset, P.b.offset, P.lam.offset, C.sizeof(G)]
    assert [int(v) for v in b.split()] == [G.lam.offset, G.lo.offset, G.hi.offset, G.lo_vec.offset, G.hi_vec.offset]


def test_header_is_plain_c(tmp_path):
    """The boundary must be consumable from C (and therefore from Julia's ccall): compile the header as C11, pedantic."""
    src = tmp_path / "inc.c"
    src.write_text('#define CIAO_BENCH_API 1\n#include "ciao_hip.h"\nint main(void){return CIAO_ABI_VERSION - 3;}\n')
    subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "inc.o")], check=True)


def test_context_creation_fails_loudly_without_a_gpu(ciao):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    lib = ciao._lib.load()
    h = C.c_void_p()
    st = lib.ciao_ctx_create(0, None, C.byref(h))
    assert st == ciao._lib.ERR_HIP and not h.value
    assert b"HIP error" in lib.ciao_last_error()
    from ciaoalgorithms_jl_amd.device import Context
    with pytest.raises(ciao._lib.CiaoError):
        Context(0)


def test_missing_extension_fails_loudly(ciao, monkeypatch):
    """No libciao_hip.so -> ImportError with build instructions; never a silent CPU path."""
    L = ciao._lib
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", os.path.join(ROOT, "ciaoalgorithms.jl_amd", "no_such_libciao_hip.so"))
    with pytest.raises(ImportError, match="HIP extension is not built"):
        L.load()


def test_null_arguments_are_rejected_before_any_launch(ciao):
    lib = ciao._lib.load()
    assert lib.ciao_ctx_create(0, None, None) == ciao._lib.ERR_ARG
    assert lib.ciao_full_gradient(None, None, None, None) == ciao._lib.ERR_ARG
    assert lib.ciao_ctx_synchronize(None) == ciao._lib.ERR_ARG
    assert lib.ciao_ctx_chain_batch_begin(None) == ciao._lib.ERR_ARG and lib.ciao_ctx_chain_batch_end(None, 1) == ciao._lib.ERR_ARG
    assert lib.ciao_svrg_epoch_tail(None, None, 1, 0, None, None, None, None) == ciao._lib.ERR_ARG
    assert lib.ciao_ctx_destroy(None) == ciao._lib.OK


def test_product_package_never_touches_the_oracle():
    """No file of the product package may import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "ciaoalgorithms.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".jl", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for bad in ("from oracle", "import oracle", "ciao_oracle", "libciao_oracle", "orc_"):
                    assert bad not in text, f"{os.path.join(dirpath, f)} references the oracle ({bad})"


def test_the_drivers_build_check_agrees_with_the_header():
    """__graft_entry__.build() ends with check_library(): ABI version of the header == of the binding == of the built library, every
    declared symbol exported (round 5: the version went to 3 and build() still asserted 2)."""
    import __graft_entry__ as G
    G.check_library()
