"""ctypes loader for the CPU oracle (libciao_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/ciao_oracle.h.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under ciaoalgorithms.jl_amd/ does.

All arrays are host numpy arrays (C-contiguous); indices are 0-based int64.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CIAO_ORACLE_LIB selects another build of the same sources (the AddressSanitizer/UBSan one from `make -C oracle asan`)
_LIB_PATH = os.environ.get("CIAO_ORACLE_LIB") or os.path.join(_HERE, "libciao_oracle.so")

LOSS_LS, LOSS_LOGISTIC, LOSS_ZERO, LOSS_LS_COMPLEX = 0, 1, 2, 3
PROX_ZERO, PROX_L1, PROX_BOX, PROX_L1_COMPLEX = 0, 1, 2, 3


def as_pairs(a):
    """A complex array as interleaved (re, im) pairs of its real type -- Julia's reinterpret(R, a); real arrays unchanged."""
    a = np.ascontiguousarray(a)
    return a.view(a.real.dtype) if np.iscomplexobj(a) else a


def as_complex(a):
    """The inverse view: (re, im) pairs -> complex."""
    a = np.ascontiguousarray(a)
    return a.view(np.complex128 if a.dtype == np.float64 else np.complex64)


class _Problem(C.Structure):
    _fields_ = [("loss", C.c_int32), ("_pad", C.c_int32), ("N", C.c_int64), ("d", C.c_int64),
                ("A", C.c_void_p), ("b", C.c_void_p), ("lam", C.c_double)]


class _ProxDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("lam", C.c_double), ("lo", C.c_double),
                ("hi", C.c_double), ("lo_vec", C.c_void_p), ("hi_vec", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("ciao_oracle.c", "ciao_oracle_impl.inc", "ciao_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if os.environ.get("CIAO_ORACLE_LIB"):
        return _LIB_PATH
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        for sfx, ct in (("f64", C.c_double), ("f32", C.c_float)):
            getattr(_lib, f"orc_gradient_{sfx}").restype = ct
            getattr(_lib, f"orc_gradient_{sfx}").argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, ct, C.c_void_p, C.c_void_p]
            getattr(_lib, f"orc_objective_{sfx}").restype = C.c_double
            getattr(_lib, f"orc_julia_sum_scalar_{sfx}").restype = ct
            getattr(_lib, f"orc_julia_sum_scalar_{sfx}").argtypes = [C.c_int64, C.c_void_p, C.c_int]
            getattr(_lib, f"orc_julia_sum_vec_{sfx}").restype = None
            getattr(_lib, f"orc_julia_sum_vec_{sfx}").argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(_lib, f"orc_full_pass_omp_{sfx}").restype = C.c_int
    return _lib


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _ct(dtype):
    return C.c_double if np.dtype(dtype) == np.float64 else C.c_float


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _chk(a, dtype, shape=None):
    assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.c_contiguous, (type(a), getattr(a, "dtype", None))
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _idx(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a


class Problem:
    """Packed F = [f_1 .. f_N]: loss kind, row-major A (N x d), b (targets / labels), LeastSquares λ."""

    def __init__(self, loss, A, b=None, lam=1.0, N_total=None):
        """N_total: the rows in `A` are a gathered SUBSET of a problem of N_total rows (tests that hold a device run at full size
        against the oracle on the rows it touched): the functions that take explicit row indices (svrg_inner, saga_steps, ...)
        then use 1/N_total where the algorithm says 1/N and never look at a row that is not there."""
        self.loss = {"ls": LOSS_LS, "logistic": LOSS_LOGISTIC, "zero": LOSS_ZERO, "ls_complex": LOSS_LS_COMPLEX}.get(loss, loss)
        if np.iscomplexobj(A):   # complex T: LeastSquares on (re, im) pairs; d counts reals, b holds N pairs
            assert self.loss in (LOSS_LS, LOSS_LS_COMPLEX)
            self.loss = LOSS_LS_COMPLEX
            A = as_pairs(A)
            b = as_pairs(np.asarray(b, dtype=np.result_type(A.dtype, np.complex64)))
        self.A = np.ascontiguousarray(A)
        self.dtype = self.A.dtype
        _sfx(self.dtype)
        self.N, self.d = self.A.shape
        self.b = None if b is None else np.ascontiguousarray(b, dtype=self.dtype)
        if self.loss == LOSS_LS_COMPLEX:
            assert self.d % 2 == 0 and self.b.shape == (2 * self.N,)
        elif self.loss != LOSS_ZERO:
            assert self.b is not None and self.b.shape == (self.N,)
        self.lam = float(lam)
        self.rows = self.N
        if N_total is not None:
            assert N_total >= self.N
            self.N = int(N_total)
        self._c = _Problem(self.loss, 0, self.N, self.d, self.A.ctypes.data,
                           self.b.ctypes.data if self.b is not None else None, self.lam)

    @property
    def ref(self):
        return C.byref(self._c)


class Prox:
    """g: ('zero',) | ('l1', lam) | ('box', lo, hi) with scalar or per-coordinate bounds."""

    def __init__(self, kind="zero", lam=0.0, lo=-np.inf, hi=np.inf, dtype=np.float64):
        self.kind = {"zero": PROX_ZERO, "l1": PROX_L1, "box": PROX_BOX, "l1_complex": PROX_L1_COMPLEX}.get(kind, kind)
        self.lam = float(lam)
        self._lo_vec = self._hi_vec = None
        lo_s, hi_s = -np.inf, np.inf
        if self.kind == PROX_BOX:
            if np.ndim(lo) > 0:
                self._lo_vec = np.ascontiguousarray(lo, dtype=dtype)
            else:
                lo_s = float(lo)
            if np.ndim(hi) > 0:
                self._hi_vec = np.ascontiguousarray(hi, dtype=dtype)
            else:
                hi_s = float(hi)
        self._c = _ProxDesc(self.kind, 0, self.lam, lo_s, hi_s,
                            self._lo_vec.ctypes.data if self._lo_vec is not None else None,
                            self._hi_vec.ctypes.data if self._hi_vec is not None else None)

    @property
    def ref(self):
        return C.byref(self._c)


# ---------------------------------------------------------------------------------------------------------
# thin typed wrappers (one per exported routine)
# ---------------------------------------------------------------------------------------------------------

def gradient(loss, a, bi, lam, x):
    """gradient(f_i, x) -> (grad, f_i(x)); for LOSS_LS_COMPLEX a, x are (re, im) pairs and bi the target's pair"""
    dt = a.dtype
    y = np.empty_like(x)
    bp = np.ascontiguousarray(np.atleast_1d(bi), dtype=dt)
    f = getattr(lib(), f"orc_gradient_{_sfx(dt)}")(int(loss), a.shape[0], _p(a), _p(bp), _ct(dt)(lam), _p(x), _p(y))
    return y, f


def prox(g: Prox, x, gamma):
    y = np.empty_like(x)
    getattr(lib(), f"orc_prox_{_sfx(x.dtype)}")(g.ref, C.c_int64(x.shape[0]), _p(x), _ct(x.dtype)(gamma), _p(y))
    return y


def full_pass(p: Problem, x):
    av = np.empty(p.d, p.dtype)
    tmp = np.empty(p.d, p.dtype)
    getattr(lib(), f"orc_full_pass_{_sfx(p.dtype)}")(p.ref, _p(_chk(x, p.dtype, (p.d,))), _p(av), _p(tmp))
    return av


def full_pass_omp(p: Problem, x):
    av = np.empty(p.d, p.dtype)
    nt = getattr(lib(), f"orc_full_pass_omp_{_sfx(p.dtype)}")(p.ref, _p(_chk(x, p.dtype, (p.d,))), _p(av))
    return av, nt


def shard_pass(p: Problem, x, av):
    """av += sum over the rows HELD by `p` of grad f_i(x) / p.N, sequentially (p built with N_total = the whole problem's
    row count): the reference's full pass restricted to a row range, continued in `av` (SVRG_basic.jl:88-92)."""
    tmp = np.empty(p.d, p.dtype)
    getattr(lib(), f"orc_shard_pass_{_sfx(p.dtype)}")(p.ref, C.c_int64(p.rows), _p(_chk(x, p.dtype, (p.d,))),
                                                    _p(_chk(av, p.dtype, (p.d,))), _p(tmp))
    return av


def shard_sums_omp(p: Problem, x, acc):
    """acc[k] += sum over the rows held by `p` of c_i(x) a_ik (raw, not divided by N) on all host cores, long double sums;
    `acc` is a contiguous np.longdouble d-vector that is continued (slabs chain).  Returns the thread count."""
    assert acc.dtype == np.longdouble and acc.shape == (p.d,) and acc.flags.c_contiguous
    fn = getattr(lib(), f"orc_shard_sums_omp_{_sfx(p.dtype)}")
    fn.restype = C.c_int
    return fn(p.ref, C.c_int64(p.rows), _p(_chk(x, p.dtype, (p.d,))), C.c_void_p(acc.ctypes.data))


def svrg_init(p: Problem, x0):
    av, z, z_full, w = (np.empty(p.d, p.dtype) for _ in range(4))
    getattr(lib(), f"orc_svrg_init_{_sfx(p.dtype)}")(p.ref, _p(_chk(x0, p.dtype, (p.d,))), _p(av), _p(z), _p(z_full), _p(w))
    return av, z, z_full, w


def svrg_inner(p: Problem, g: Prox, gamma, idx, av, z, z_full, w):
    idx = _idx(idx)
    getattr(lib(), f"orc_svrg_inner_{_sfx(p.dtype)}")(p.ref, g.ref, _ct(p.dtype)(gamma), C.c_int64(len(idx)), _p(idx),
                                                    _p(av), _p(z), _p(z_full), _p(w))


def svrg_iterate(p: Problem, g: Prox, gamma, idx, plus, av, z, z_full, w):
    idx = _idx(idx)
    getattr(lib(), f"orc_svrg_iterate_{_sfx(p.dtype)}")(p.ref, g.ref, _ct(p.dtype)(gamma), C.c_int64(len(idx)), _p(idx),
                                                      C.c_int(1 if plus else 0), _p(av), _p(z), _p(z_full), _p(w))


def saga_init(p: Problem, g: Prox, gamma, x0):
    table = np.empty((p.N, p.d), p.dtype)
    av, z = np.empty(p.d, p.dtype), np.empty(p.d, p.dtype)
    getattr(lib(), f"orc_saga_init_{_sfx(p.dtype)}")(p.ref, g.ref, _ct(p.dtype)(gamma), _p(_chk(x0, p.dtype, (p.d,))),
                                                   _p(table), _p(av), _p(z))
    return table, av, z


def saga_steps(p: Problem, g: Prox, gamma, sag, idx, table, av, z):
    idx = _idx(idx)
    getattr(lib(), f"orc_saga_steps_{_sfx(p.dtype)}")(p.ref, g.ref, _ct(p.dtype)(gamma), C.c_int(1 if sag else 0),
                                                    C.c_int64(len(idx)), _p(idx), _p(table), _p(av), _p(z))


def finito_init(p: Problem, g: Prox, gam, x0):
    gam = _chk(gam, p.dtype, (p.N,))
    table = np.empty((p.N, p.d), p.dtype)
    av, z = np.empty(p.d, p.dtype), np.empty(p.d, p.dtype)
    hg = _ct(p.dtype)(0)
    getattr(lib(), f"orc_finito_init_{_sfx(p.dtype)}")(p.ref, g.ref, _p(gam), _p(_chk(x0, p.dtype, (p.d,))), _p(table),
                                                     _p(av), _p(z), C.byref(hg))
    return table, av, z, p.dtype.type(hg.value)


def _csr(batches):
    bptr = np.zeros(len(batches) + 1, np.int64)
    for t, bt in enumerate(batches):
        bptr[t + 1] = bptr[t] + len(bt)
    bidx = np.concatenate([np.asarray(bt, np.int64) for bt in batches]) if batches else np.zeros(0, np.int64)
    return bptr, np.ascontiguousarray(bidx)


def finito_steps(p: Problem, g: Prox, gam, hat_gamma, batches, table, av, z):
    """One Base.iterate per entry of `batches` (each a list of 0-based sample indices)."""
    bptr, bidx = _csr(batches)
    getattr(lib(), f"orc_finito_steps_{_sfx(p.dtype)}")(p.ref, g.ref, _p(gam), _ct(p.dtype)(hat_gamma),
                                                      C.c_int64(len(batches)), _p(bptr), _p(bidx), _p(table), _p(av), _p(z))


def lfinito_init(p: Problem, gam, x0):
    gam = _chk(gam, p.dtype, (p.N,))
    av, z, z_full = (np.empty(p.d, p.dtype) for _ in range(3))
    hg = _ct(p.dtype)(0)
    getattr(lib(), f"orc_lfinito_init_{_sfx(p.dtype)}")(p.ref, _p(gam), _p(_chk(x0, p.dtype, (p.d,))), _p(av), _p(z),
                                                      _p(z_full), C.byref(hg))
    return av, z, z_full, p.dtype.type(hg.value)


def lfinito_iterate(p: Problem, g: Prox, gam, hat_gamma, batches, av, z, z_full):
    bptr, bidx = _csr(batches)
    getattr(lib(), f"orc_lfinito_iterate_{_sfx(p.dtype)}")(p.ref, g.ref, _p(gam), _ct(p.dtype)(hat_gamma),
                                                         C.c_int64(len(batches)), _p(bptr), _p(bidx), _p(av), _p(z), _p(z_full))


def afinito_init(p: Problem, g: Prox, alpha, x0, retry_signs=None):
    """Finito_adaptive.jl:59-98 -> (table, gtable, gam, fi_x, av, z, hat_gamma)."""
    table, gtable = np.empty((p.N, p.d), p.dtype), np.empty((p.N, p.d), p.dtype)
    gam, fi_x = np.empty(p.N, p.dtype), np.empty(p.N, p.dtype)
    av, z = np.empty(p.d, p.dtype), np.empty(p.d, p.dtype)
    hg = _ct(p.dtype)(0)
    nretry = C.c_int64(0 if retry_signs is None else len(retry_signs))
    rs = None if retry_signs is None else np.ascontiguousarray(retry_signs, dtype=p.dtype)
    rc = getattr(lib(), f"orc_afinito_init_{_sfx(p.dtype)}")(p.ref, g.ref, _ct(p.dtype)(alpha), _p(_chk(x0, p.dtype, (p.d,))),
                                                             _p(table), _p(gtable), _p(gam), _p(fi_x), _p(av), _p(z), C.byref(hg),
                                                             _p(rs), C.byref(nretry))
    if rc != 0:
        raise RuntimeError("adaptive Finito init: the Lipschitz probe needs random retries that were not supplied")
    return table, gtable, gam, fi_x, av, z, p.dtype.type(hg.value)


def afinito_steps(p: Problem, g: Prox, alpha, tol_b, idx, table, gtable, gam, fi_x, hat_gamma, av, z):
    """Finito_adaptive.jl:118-155 for the samples idx -> (steps completed, new hat_gamma, backtracking trials)."""
    idx = _idx(idx)
    hg = _ct(p.dtype)(hat_gamma)
    ntr = C.c_int64(0)
    fn = getattr(lib(), f"orc_afinito_steps_{_sfx(p.dtype)}")
    fn.restype = C.c_int64
    done = fn(p.ref, g.ref, _ct(p.dtype)(alpha), _ct(p.dtype)(tol_b), C.c_int64(len(idx)), _p(idx), _p(table), _p(gtable),
              _p(gam), _p(fi_x), C.byref(hg), _p(av), _p(z), C.byref(ntr))
    return int(done), p.dtype.type(hg.value), int(ntr.value)


class SepQuad:
    """F = [f_1..f_N], f_i = Sum(Quadratic(Q_i, q_i), SqrDistL2(IndBox(lo, hi), eta))  (test/test_sharing.jl:16-25).
    Q of shape N x d holds the diagonals (the test's diagm); N x d x d holds one dense matrix per agent."""

    def __init__(self, Q, q, eta, lo, hi):
        self.Q = np.ascontiguousarray(Q)
        self.dtype = self.Q.dtype
        self.q = np.ascontiguousarray(q, dtype=self.dtype)
        self.dense = self.Q.ndim == 3
        self.N, self.d = self.q.shape
        assert self.Q.shape == ((self.N, self.d, self.d) if self.dense else (self.N, self.d))
        self.eta, self.lo, self.hi = float(eta), float(lo), float(hi)

    def args(self):
        ct = _ct(self.dtype)
        return (C.c_int64(self.N), C.c_int64(self.d), C.c_int32(int(self.dense)), _p(self.Q), _p(self.q), ct(self.eta), ct(self.lo),
                ct(self.hi))


def proshi_init(f: SepQuad, g: Prox, gam, x0):
    gam = _chk(gam, f.dtype, (f.N,))
    table = np.empty((f.N, f.d), f.dtype)
    av, z = np.empty(f.d, f.dtype), np.empty(f.d, f.dtype)
    hg = _ct(f.dtype)(0)
    getattr(lib(), f"orc_proshi_init_{_sfx(f.dtype)}")(*f.args(), g.ref, _p(gam), _p(_chk(x0, f.dtype, (f.d,))), _p(table), _p(av),
                                                     _p(z), C.byref(hg))
    return table, av, z, f.dtype.type(hg.value)


def proshi_steps(f: SepQuad, g: Prox, gam, hat_gamma, batches, table, av, z):
    bptr, bidx = _csr(batches)
    getattr(lib(), f"orc_proshi_steps_{_sfx(f.dtype)}")(*f.args(), g.ref, _p(gam), _ct(f.dtype)(hat_gamma), C.c_int64(len(batches)),
                                                      _p(bptr), _p(bidx), _p(table), _p(av), _p(z))


def proshi_solution(f: SepQuad, gam, z, table):
    getattr(lib(), f"orc_proshi_solution_{_sfx(f.dtype)}")(C.c_int64(f.N), C.c_int64(f.d), _p(gam), _p(z), _p(table))
    return table


def objective(p: Problem, g: Prox, x):
    return getattr(lib(), f"orc_objective_{_sfx(p.dtype)}")(p.ref, g.ref, _p(_chk(x, p.dtype, (p.d,))))


def julia_sum_vec(rows, div=None):
    """sum(A) for A = the rows of a (N, d) array (sum(A ./ div) with div): Base.mapreduce_impl's association order."""
    rows = np.ascontiguousarray(rows)
    N, d = rows.shape
    out = np.empty(d, rows.dtype)
    if div is not None:
        div = np.ascontiguousarray(div, dtype=rows.dtype)
    getattr(lib(), f"orc_julia_sum_vec_{_sfx(rows.dtype)}")(N, d, _p(rows), _p(div), _p(out))
    return out


def julia_sum_scalar(x, inv=False):
    """sum(x) (or sum(1 ./ x)) for a vector of scalars, pairwise above 1024 elements with left-to-right leaves."""
    x = np.ascontiguousarray(x)
    return x.dtype.type(getattr(lib(), f"orc_julia_sum_scalar_{_sfx(x.dtype)}")(x.size, _p(x), 1 if inv else 0))
