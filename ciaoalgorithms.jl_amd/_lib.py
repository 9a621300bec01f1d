"""ctypes binding of libciao_hip.so -- the same symbols a Julia `ccall` binds (include/ciao_hip.h).

There is NO CPU fallback: if the shared library is missing this module raises at import of the first symbol, and
if there is no GPU `ciao_ctx_create` fails with CIAO_ERR_HIP.  The CPU oracle lives in /oracle and is never imported
from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CIAO_HIP_LIB selects another build of the same sources: the experiment builds of tools/exp_build.sh (timing macros), which
# live under build/<name>/ and never replace the product library.  ciao_build_flags() tells the two apart.
LIB_PATH = os.environ.get("CIAO_HIP_LIB") or os.path.join(_HERE, "libciao_hip.so")
ABI_VERSION = 3

OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_ALLOC, ERR_HOOK = 0, -1, -2, -3, -4, -5
F32, F64 = 0, 1
LOSS_LS, LOSS_LOGISTIC, LOSS_ZERO, LOSS_LS_COMPLEX = 0, 1, 2, 3
PROX_ZERO, PROX_L1, PROX_BOX, PROX_L1_COMPLEX = 0, 1, 2, 3


class CiaoError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libciao_hip status {status}: {msg}")
        self.status = status


class Problem(C.Structure):
    """ciao_problem"""
    _fields_ = [("loss", C.c_int32), ("dtype", C.c_int32), ("N", C.c_int64), ("d", C.c_int64), ("ld", C.c_int64),
                ("N_total", C.c_int64), ("A", C.c_void_p), ("b", C.c_void_p), ("lam", C.c_double)]


class ProxDesc(C.Structure):
    """ciao_prox_desc"""
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("lam", C.c_double), ("lo", C.c_double), ("hi", C.c_double),
                ("lo_vec", C.c_void_p), ("hi_vec", C.c_void_p)]


class SepQuad(C.Structure):
    """ciao_sepquad"""
    _fields_ = [("dtype", C.c_int32), ("dense", C.c_int32), ("N", C.c_int64), ("d", C.c_int64), ("ld", C.c_int64),
                ("N_total", C.c_int64), ("Q", C.c_void_p), ("q", C.c_void_p), ("eta", C.c_double), ("lo", C.c_double),
                ("hi", C.c_double)]


MAX_SHARDS = 8


class ShardTable(C.Structure):
    """ciao_shard_table"""
    _fields_ = [("nshards", C.c_int32), ("owner", C.c_int32), ("row0", C.c_int64 * (MAX_SHARDS + 1)),
                ("A", C.c_void_p * MAX_SHARDS), ("b", C.c_void_p * MAX_SHARDS), ("table", C.c_void_p * MAX_SHARDS),
                ("meta", C.c_void_p * MAX_SHARDS)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)

_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_PP, _GP, _SP = C.POINTER(Problem), C.POINTER(ProxDesc), C.POINTER(SepQuad)

# name -> (restype, argtypes): one row per declaration in include/ciao_hip.h
SIGNATURES = {
    "ciao_abi_version": (_i32, []),
    "ciao_last_error": (C.c_char_p, []),
    "ciao_build_flags": (C.c_char_p, []),
    "ciao_ctx_create": (_i32, [_i32, _vp, C.POINTER(_vp)]),
    "ciao_ctx_destroy": (_i32, [_vp]),
    "ciao_ctx_set_stream": (_i32, [_vp, _vp]),
    "ciao_ctx_synchronize": (_i32, [_vp]),
    "ciao_ctx_set_allreduce": (_i32, [_vp, ALLREDUCE_FN, _vp]),
    "ciao_ctx_set_rccl": (_i32, [_vp, _vp, C.c_char_p]),
    "ciao_ctx_set_monitor": (_i32, [_vp, _GP, _vp]),
    "ciao_ctx_set_option": (_i32, [_vp, C.c_char_p, _i64]),
    "ciao_ctx_timing_enable": (_i32, [_vp, _i32]),
    "ciao_ctx_timing_read": (_i32, [_vp, C.POINTER(_f64), C.POINTER(_i64)]),
    "ciao_ctx_last_kernel": (C.c_char_p, [_vp]),
    "ciao_gradient": (_i32, [_vp, _PP, _i64, _vp, _vp, _vp]),
    "ciao_prox": (_i32, [_vp, _i32, _i64, _GP, _vp, _f64, _vp]),
    "ciao_full_gradient": (_i32, [_vp, _PP, _vp, _vp]),
    "ciao_proxgrad_step": (_i32, [_vp, _PP, _GP, _f64, _vp, _vp, _vp]),
    "ciao_objective": (_i32, [_vp, _PP, _GP, _vp, C.POINTER(_f64)]),
    "ciao_svrg_init": (_i32, [_vp, _PP, _vp, _vp, _vp, _vp, _vp]),
    "ciao_svrg_inner": (_i32, [_vp, _PP, _GP, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_svrg_iterate": (_i32, [_vp, _PP, _GP, _f64, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ciao_svrg_epoch_tail": (_i32, [_vp, _PP, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ciao_svrg_epoch_tail_multi": (_i32, [_vp, _PP, _i32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "ciao_full_gradient_multi": (_i32, [_vp, _PP, _i32, _vp, _vp]),
    "ciao_ctx_set_shards": (_i32, [_vp, C.POINTER(ShardTable)]),
    "ciao_ipc_export": (_i32, [_vp, _vp, C.POINTER(_i64)]),
    "ciao_ipc_open": (_i32, [_vp, _i64, C.POINTER(_vp)]),
    "ciao_ipc_close": (_i32, [_vp, _i64]),
    "ciao_peer_mailbox_create": (_i32, [_vp, _i64, C.POINTER(C.c_void_p), C.POINTER(_i64)]),
    "ciao_peer_mailbox_destroy": (_i32, [_vp, _vp]),
    "ciao_ctx_set_peers": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_void_p), _i64]),
    "ciao_ctx_chain_batch_begin": (_i32, [_vp]),
    "ciao_ctx_chain_batch_end": (_i32, [_vp, _i32]),
    "ciao_saga_init": (_i32, [_vp, _PP, _GP, _f64, _vp, _vp, _vp, _vp]),
    "ciao_saga_steps": (_i32, [_vp, _PP, _GP, _f64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "ciao_hat_gamma": (_i32, [_vp, _i32, _i64, _vp, C.POINTER(_f64)]),
    "ciao_finito_init": (_i32, [_vp, _PP, _GP, _vp, _f64, _vp, _vp, _vp, _vp]),
    "ciao_finito_steps": (_i32, [_vp, _PP, _GP, _vp, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_finito_steps_blocks": (_i32, [_vp, _PP, _GP, _vp, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_lfinito_init": (_i32, [_vp, _PP, _f64, _vp, _vp, _vp, _vp]),
    "ciao_lfinito_iterate": (_i32, [_vp, _PP, _GP, _vp, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_lfinito_iterate_blocks": (_i32, [_vp, _PP, _GP, _vp, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_afinito_init": (_i32, [_vp, _PP, _GP, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ciao_afinito_probe": (_i32, [_vp, _PP, _i64, _vp, _vp, _f64, C.POINTER(_f64)]),
    "ciao_afinito_steps": (_i32, [_vp, _PP, _GP, _f64, _f64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "ciao_proshi_init": (_i32, [_vp, _SP, _GP, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ciao_proshi_steps": (_i32, [_vp, _SP, _GP, _vp, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_proshi_steps_blocks": (_i32, [_vp, _SP, _GP, _vp, _f64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ciao_proshi_solution": (_i32, [_vp, _SP, _vp, _vp, _vp]),
    "ciao_synth_normal": (_i32, [_vp, _i32, _vp, _i64, _i64, _i64, _i64, C.c_uint64, _f64]),
    "ciao_synth_targets": (_i32, [_vp, _PP, _vp, _f64, _i32, _i64, C.c_uint64, _vp]),
    "ciao_sample_batches": (_i32, [C.c_uint64, C.c_uint64, _i64, _i64, _i64, _vp, C.POINTER(C.c_uint64)]),
    "ciao_sample_uniform": (_i32, [_vp, C.c_uint64, C.c_uint64, _i64, _i64, _vp]),
    "ciao_peer_allreduce": (_i32, [_vp, _i32, _i64, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libciao_hip.so and declare every prototype.  Raises (loudly) when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C ciaoalgorithms.jl_amd/csrc`).  There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ciao_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.ciao_abi_version()}, this binding needs {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != OK:
        raise CiaoError(status, load().ciao_last_error().decode("utf-8", "replace"))
