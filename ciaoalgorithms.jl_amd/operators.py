"""Operator descriptions in the ProximalOperators.jl vocabulary, and their packing for the device.

The reference accepts any ProximalOperators object for f_i and g and calls `gradient!` / `prox!` on it per sample
(SVRG_basic.jl:74-80 ...).  A GPU kernel cannot call an opaque host closure per sample, so this path *recognises* the
families the reference's own tests use and packs them:

    f_i = LeastSquares(A_i (1 x d), b_i (1), λ)                       test/test_lasso.jl:52-54
    f_i = Precompose(LogisticLoss([y_i], 1.0), a_i' (1 x d), 1.0)     test/test_logistic_l1.jl:36
    f_i = Zero()                                                      SVRG.jl:58 (default F)
    g   = NormL1(λ) | Zero() | IndBox(lo, hi)                         test/test_lasso.jl:59, SVRG.jl:49, test_sharing.jl:16

Anything else raises UnpackableOperator (a TypeError) -- there is no SILENT host fallback; `fallback="host"` on a solver call
asks for the explicit one (host_route.py).  These classes only *describe* operators (they hold host data and constructor
arguments); no arithmetic happens here (host_ops.py has the host route's).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from .device import PackedF, ProxG, torch_dtype


class UnpackableOperator(TypeError):
    """F or g is not one of the families the device path packs.  The solvers' `fallback="host"` catches exactly this (and
    nothing else: a dtype mismatch, a wrong length ... stay errors) and runs the problem on the host route (host_route.py)."""


class Zero:
    """ProximalOperators.Zero(): f(x) = 0; gradient 0; prox = identity."""

    def __repr__(self):
        return "Zero()"


class NormL1:
    """ProximalOperators.NormL1(λ): g(x) = λ‖x‖₁ (default λ = 1)."""

    def __init__(self, lam=1.0):
        if not (np.ndim(lam) == 0 and float(lam) >= 0):
            raise ValueError("NormL1: λ must be a nonnegative scalar on the device path")
        self.lam = float(lam)

    def __repr__(self):
        return f"NormL1({self.lam})"


class IndBox:
    """ProximalOperators.IndBox(lo, hi): indicator of {lo <= x <= hi}; scalar or per-coordinate bounds."""

    def __init__(self, lo, hi):
        if np.any(np.asarray(lo) > np.asarray(hi)):
            raise ValueError("IndBox: lo must be <= hi")
        self.lo, self.hi = lo, hi

    def __repr__(self):
        return "IndBox(...)"


class LeastSquares:
    """ProximalOperators.LeastSquares(A, b, λ): f(x) = λ/2 ‖A x − b‖² (default λ = 1)."""

    def __init__(self, A, b, lam=1.0):
        self.A = np.atleast_2d(np.asarray(A))
        self.b = np.atleast_1d(np.asarray(b))
        if self.A.shape[0] != self.b.shape[0]:
            raise ValueError("LeastSquares: A and b have incompatible sizes")
        if float(lam) < 0:
            raise ValueError("LeastSquares: λ must be nonnegative")
        self.lam = float(lam)


class LogisticLoss:
    """ProximalOperators.LogisticLoss(y, μ): f(x) = μ Σ log(1 + exp(−y_i x_i))."""

    def __init__(self, y, mu=1.0):
        self.y = np.atleast_1d(np.asarray(y))
        if not np.all(np.abs(self.y) == 1):
            raise ValueError("LogisticLoss: labels must be ±1")
        self.mu = float(mu)


class Precompose:
    """ProximalOperators.Precompose(f, L, μ[, b]): x ↦ f(L x + b), with L Lᵀ = μ I claimed by the caller."""

    def __init__(self, f, Lmat, mu=1.0, b=0.0):
        self.f, self.L, self.mu, self.b = f, np.atleast_2d(np.asarray(Lmat)), float(mu), b


class Quadratic:
    """ProximalOperators.Quadratic(Q, q): f(x) = 1/2 <x, Q x> + <q, x>.  Only diagonal Q is packable (test_sharing.jl:21-22
    builds Q = diagm(d_i))."""

    def __init__(self, Q, q):
        self.Q, self.q = np.atleast_2d(np.asarray(Q)), np.atleast_1d(np.asarray(q))
        if self.Q.shape != (self.q.shape[0], self.q.shape[0]):
            raise ValueError("Quadratic: Q must be n x n and q of length n")


class SqrDistL2:
    """ProximalOperators.SqrDistL2(ind, λ): f(x) = λ/2 dist²(x, set of `ind`); here `ind` must be an IndBox with scalar bounds."""

    def __init__(self, ind, lam=1.0):
        if float(lam) < 0:
            raise ValueError("SqrDistL2: λ must be nonnegative")
        self.ind, self.lam = ind, float(lam)


class Sum:
    """ProximalOperators.Sum(f1, f2, ...): x ↦ Σ f_k(x)."""

    def __init__(self, *fs):
        self.fs = fs


# ------------------------------------------------------------------------------------------------------------------------
# packing
# ------------------------------------------------------------------------------------------------------------------------

def _dev(a: np.ndarray, dtype: torch.dtype, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype=dtype, device=device).contiguous()


def _mark_padded(p: PackedF, d: int, dp: int) -> PackedF:
    p.padded_from = d if dp != d else None
    return p


def pack_F(F, N: int, d: int, R, device=None, complex_pairs: bool = False, pad_to: int | None = None) -> PackedF:
    """Recognise and pack the finite-sum term.  F is None (Zero()), a PackedF, or a sequence of N one-row operators.
    d counts the REAL coordinates of x0; with complex_pairs (complex x0) they are (re, im) pairs and the operators' rows
    have d/2 complex entries.  pad_to > d (real problems built from host operators only): the packed rows get pad_to - d zero
    columns -- coordinates that multiply nothing, so the d real ones come out as without them (solvers._Iterable: rows of a
    whole number of 16-byte chunks run the LDS-DMA chains, 2-3.7x the register-ring chains' speed)."""
    dtype = torch_dtype(R)
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n = d // 2 if complex_pairs else d          # entries of one operator row
    dp = d if pad_to is None else int(pad_to)

    def padded(A):
        if dp == d:
            return A
        out = np.zeros((A.shape[0], dp), dtype=A.dtype)
        out[:, :d] = A
        return out
    if isinstance(F, PackedF):
        if F.dtype != dtype:
            raise TypeError(f"F is packed as {F.dtype} but the solver's real type is {dtype} (no silent promotion)")
        if F.d != d:
            raise ValueError(f"F has d={F.d} but x0 has {d} (real) coordinates")
        return F
    if F is None:
        return _mark_padded(PackedF.zero(N, dp, dtype), d, dp)
    F = list(F)
    if len(F) != N:
        raise ValueError(f"F has {len(F)} terms but N={N}")
    if N == 0:
        return _mark_padded(PackedF.zero(0, dp, dtype), d, dp)
    if all(isinstance(f, Zero) for f in F):
        return _mark_padded(PackedF.zero(N, dp, dtype), d, dp)
    if all(isinstance(f, LeastSquares) for f in F):
        lam = F[0].lam
        if any(f.lam != lam for f in F):
            raise UnpackableOperator("LeastSquares terms with different λ cannot be packed")
        if any(f.A.shape != (1, n) for f in F):
            raise UnpackableOperator("only one-row LeastSquares terms (A_i of size 1 x d) are packable")
        A = np.concatenate([f.A for f in F], axis=0)
        b = np.concatenate([f.b for f in F], axis=0)
        if complex_pairs or np.iscomplexobj(A) or np.iscomplexobj(b):
            # complex T: (re, im) pairs of R -- reinterpret(R, ...) in Julia terms (a complex problem needs a complex x0)
            if not complex_pairs:
                raise TypeError("complex LeastSquares terms need a complex x0 (one type T for the problem, CIAOAlgorithms.jl:3)")
            ct = np.complex128 if dtype == torch.float64 else np.complex64
            Ap = np.ascontiguousarray(A.astype(ct)).view(ct(0).real.dtype)
            bp = np.ascontiguousarray(b.astype(ct)).view(ct(0).real.dtype)
            return PackedF.least_squares_complex(_dev(Ap, dtype, device), _dev(bp, dtype, device), lam)
        out = PackedF.least_squares(_dev(padded(A), dtype, device), _dev(b, dtype, device), lam)
        out.padded_from = d if dp != d else None
        return out
    if complex_pairs:
        raise UnpackableOperator("with a complex x0 the device path packs LeastSquares rows and Zero only")
    if all(isinstance(f, Precompose) and isinstance(f.f, LogisticLoss) for f in F):
        for f in F:
            if f.L.shape != (1, d) or f.f.y.shape != (1,) or f.f.mu != 1.0 or np.any(np.asarray(f.b) != 0):
                raise UnpackableOperator("only Precompose(LogisticLoss([y_i], 1.0), a_i' (1 x d), mu) terms are packable")
        A = np.concatenate([f.L for f in F], axis=0)
        y = np.concatenate([f.f.y for f in F], axis=0)
        out = PackedF.logistic(_dev(padded(A), dtype, device), _dev(y, dtype, device))
        out.padded_from = d if dp != d else None
        return out
    kinds = sorted({type(f).__name__ for f in F})
    raise UnpackableOperator(f"F of kinds {kinds} is not a family the device path can pack (LeastSquares rows, "
                             f"Precompose(LogisticLoss) rows, Zero).  An opaque operator object cannot be called per sample from a GPU "
                             f"kernel, and there is no SILENT host route (it would make every parity and performance statement about "
                             f"this path void): give F as one of those families or as a device matrix (PackedF / pack_rows_from_host), "
                             f"or ask for the slow host route explicitly with fallback=\"host\"")


def require_packable(F, g, complex_x0: bool = False):
    """The family checks of pack_F / pack_g WITHOUT touching a device: raises UnpackableOperator exactly when they would.  The
    solvers call it first, so that `fallback="host"` never allocates on (or needs) a GPU for a problem that cannot run there."""
    if not (F is None or isinstance(F, PackedF)):
        F = list(F)
        ok = (all(isinstance(f, Zero) for f in F)
              or (all(isinstance(f, LeastSquares) for f in F) and all(f.lam == F[0].lam and f.A.shape[0] == 1 for f in F))
              or (not complex_x0 and all(isinstance(f, Precompose) and isinstance(f.f, LogisticLoss) and f.L.shape[0] == 1
                                         and f.f.y.shape == (1,) and f.f.mu == 1.0 and not np.any(np.asarray(f.b) != 0) for f in F)))
        if F and not ok:
            kinds = sorted({type(f).__name__ for f in F})
            raise UnpackableOperator(f"F of kinds {kinds} is not a family the device path can pack (one-row LeastSquares with one λ, "
                                     f"Precompose(LogisticLoss) rows, Zero).  There is no SILENT host route; ask for the slow one "
                                     f"explicitly with fallback=\"host\"")
    if not (g is None or isinstance(g, (ProxG, Zero, NormL1)) or (isinstance(g, IndBox) and not complex_x0)):
        raise UnpackableOperator(f"g of type {type(g).__name__} is not a family the device path supports (Zero, NormL1, IndBox); "
                                 f"fallback=\"host\" runs any g with a prox(x, gamma) method on the host route")


def pack_rows_from_host(chunks, N: int, d: int, R, loss: str = "ls", lam: float = 1.0, device=None, N_total=None, row0: int = 0) -> PackedF:
    """The step BEFORE the path at scale: at N = 10^7 nobody builds N one-row operator objects
    (test_lasso.jl:50-58) -- the rows arrive as blocks of a host matrix (numpy arrays, np.memmap slices of a file on disk, a
    generator that reads them).  `chunks` yields (A_k, b_k) in row order, A_k of shape (n_k, d); they are staged through two
    pinned host buffers and copied on a stream of their own, so chunk k+1 is being read / converted on the host while chunk
    k is on the wire, and never more than two chunks of host memory are held.  loss: "ls" (targets b, LeastSquares λ = lam)
    or "logistic" (labels b in {-1, +1}).  Returns the PackedF the solvers take as F."""
    dtype = torch_dtype(R)
    np_R = np.float64 if dtype == torch.float64 else np.float32
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    A = torch.empty((N, d), dtype=dtype, device=device)
    b = torch.empty((N,), dtype=dtype, device=device)
    copy = torch.cuda.Stream(device=device)
    stage, done = [None, None], [None, None]
    at = 0
    for k, (Ak, bk) in enumerate(chunks):
        Ak = np.asarray(Ak)
        bk = np.asarray(bk).reshape(-1)
        n = Ak.shape[0]
        if Ak.ndim != 2 or Ak.shape[1] != d or bk.shape[0] != n or at + n > N:
            raise ValueError(f"chunk {k}: expected (n, {d}) rows with n targets and at most {N - at} rows left, got {Ak.shape} / {bk.shape}")
        if np.iscomplexobj(Ak) or np.iscomplexobj(bk):
            raise TypeError("complex data is outside the device path")
        s = k & 1
        if done[s] is not None:
            done[s].synchronize()                        # the copy that last used this staging pair has left the host
        if stage[s] is None or stage[s][0].shape[0] < n:
            stage[s] = (torch.empty((n, d), dtype=dtype).pin_memory(), torch.empty((n,), dtype=dtype).pin_memory())
        sa, sb = stage[s]
        np.copyto(sa[:n].numpy(), Ak, casting="same_kind" if Ak.dtype.kind == "f" else "unsafe")   # converts to R on the way
        np.copyto(sb[:n].numpy(), bk.astype(np_R, copy=False))
        with torch.cuda.stream(copy):
            A[at:at + n].copy_(sa[:n], non_blocking=True)
            b[at:at + n].copy_(sb[:n], non_blocking=True)
            done[s] = torch.cuda.Event()
            done[s].record(copy)
        at += n
    if at != N:
        raise ValueError(f"the chunks held {at} rows, expected {N}")
    copy.synchronize()
    torch.cuda.current_stream(device).wait_stream(copy)
    kind = {"ls": L.LOSS_LS, "logistic": L.LOSS_LOGISTIC}[loss]
    return PackedF(kind, A, b, float(lam) if loss == "ls" else 1.0, N_total=N_total, row0=row0)


def pack_g(g, d: int, R, device=None, complex_pairs: bool = False, pad_to: int | None = None) -> ProxG:
    """complex_pairs: the coordinates are (re, im) pairs of a complex vector (complex T): NormL1 is then the complex norm
    (modulus soft-threshold); IndBox has no complex meaning."""
    dtype = torch_dtype(R)
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    if isinstance(g, ProxG):
        return g
    if g is None or isinstance(g, Zero):
        return ProxG(L.PROX_ZERO)
    if isinstance(g, NormL1):
        return ProxG(L.PROX_L1_COMPLEX if complex_pairs else L.PROX_L1, lam=g.lam)
    if complex_pairs:
        raise UnpackableOperator(f"g of type {type(g).__name__} has no complex form on the device path (Zero, NormL1)")
    if isinstance(g, IndBox):
        lo_vec = hi_vec = None
        lo, hi = -float("inf"), float("inf")
        if np.ndim(g.lo) > 0:
            lo_vec = _dev(np.concatenate([np.asarray(g.lo, dtype=np.float64).reshape(d), np.full((pad_to or d) - d, -np.inf)]), dtype, device)
        else:
            lo = float(g.lo)
        if np.ndim(g.hi) > 0:
            hi_vec = _dev(np.concatenate([np.asarray(g.hi, dtype=np.float64).reshape(d), np.full((pad_to or d) - d, np.inf)]), dtype, device)
        else:
            hi = float(g.hi)
        return ProxG(L.PROX_BOX, lo=lo, hi=hi, lo_vec=lo_vec, hi_vec=hi_vec)
    raise UnpackableOperator(f"g of type {type(g).__name__} is not a family the device path supports (Zero, NormL1, IndBox); "
                             f"fallback=\"host\" runs any g with a prox(x, gamma) method on the host route")


def pack_sharing_F(F, N: int, d: int, R, device=None):
    """Recognise F::Vector of Sum(Quadratic(Q, q), SqrDistL2(IndBox(lo, hi), η)) (test/test_sharing.jl:16-25), or a
    lone Quadratic, and pack it as a PackedSepQuad.  Anything else raises TypeError."""
    from .device import PackedSepQuad
    dtype = torch_dtype(R)
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    if isinstance(F, PackedSepQuad):
        if F.dtype != dtype or F.d != d:
            raise TypeError("packed F does not match the solver's real type / x0")
        return F
    F = list(F)
    if len(F) != N:
        raise ValueError(f"F has {len(F)} terms but N={N}")
    Qs, qs, eta, lo, hi = [], [], None, None, None
    for f in F:
        parts = list(f.fs) if isinstance(f, Sum) else [f]
        quad = [t for t in parts if isinstance(t, Quadratic)]
        dist = [t for t in parts if isinstance(t, SqrDistL2)]
        if len(quad) != 1 or len(dist) > 1 or len(quad) + len(dist) != len(parts):
            raise TypeError("ProShI device path: each f_i must be Quadratic or Sum(Quadratic, SqrDistL2(IndBox, η))")
        Q = np.asarray(quad[0].Q)
        if Q.shape != (d, d):
            raise TypeError(f"ProShI device path: Quadratic terms must be {d} x {d} (got {Q.shape})")
        Qs.append(Q)
        qs.append(quad[0].q)
        e, l, h = (0.0, 0.0, 0.0)
        if dist:
            box = dist[0].ind
            if not isinstance(box, IndBox) or np.ndim(box.lo) or np.ndim(box.hi):
                raise TypeError("ProShI device path: SqrDistL2 must wrap an IndBox with scalar bounds")
            e, l, h = dist[0].lam, float(box.lo), float(box.hi)
        if eta is None:
            eta, lo, hi = e, l, h
        elif (eta, lo, hi) != (e, l, h):
            raise TypeError("ProShI device path: all agents must share the same soft box (η, lo, hi)")
    # all-diagonal Q_i (the reference test's diagm) pack as N x d and run element-wise; otherwise N dense d x d blocks
    if all(not np.any(Q - np.diag(np.diag(Q))) for Q in Qs):
        Qp = np.stack([np.diag(Q) for Q in Qs])
    else:
        Qp = np.stack(Qs)
    return PackedSepQuad(_dev(Qp, dtype, device), _dev(np.stack(qs), dtype, device), eta, lo, hi)
