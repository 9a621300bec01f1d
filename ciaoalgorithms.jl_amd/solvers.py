"""Host-side mirror of the reference's solver API (L3) and iterables (L2) for the finite-sum hot path.

Same names, keyword arguments, iteration protocol and error behaviour as
    src/algorithms/SVRG/SVRG.jl, SVRG_basic.jl
    src/algorithms/SAGA_SAG/SAGA.jl, SAGA_basic.jl
    src/algorithms/Finito/Finito.jl, Finito_basic.jl, Finito_LFinito.jl
but every `gradient!` / `prox!` / broadcast of the hot loops runs in libciao_hip.so (HIP, gfx950) through the C ABI.
Python identifiers may be Greek, so `γ`, `μ`, `α` work as keywords exactly as in Julia; ASCII aliases `gamma`, `mu`,
`alpha` are accepted too.

Differences forced by the device boundary (documented in DESIGN.md):
  * the sampling stream is an explicit input (`stream=IndexStream(seed)`), see sampling.py;
  * F must be one of the packable families (operators.py), or an already packed `PackedF` living on the device; anything else
    raises operators.UnpackableOperator unless the caller passes `fallback="host"`, which runs the same iterables on the host
    with one operator call per sample (host_route.py: slow, announced with a warning, `state.backend == "host"`);
  * state vectors are torch device tensors; `solution(state)` returns the state's own tensor (identity, as in
    test/test_lasso.jl:185), and the solver returns it as a numpy array when x0 was a numpy array;
  * adaptive Finito (Finito_adaptive.jl) keeps its per-sample scalars in an N x 4 device
    array instead of an N x d gradient table (grad f_i = c_i a_i for the packable families).

State ownership rule (SVRG).  The SVRG inner cycle may reuse the row dots a_i'z_full that the previous full pass computed
(one dot product per update instead of two).  That is only valid while A, b and z_full are untouched between two
iterations, and `state.z_full` IS the tensor `solution(state)` returns.  The iterable therefore records torch's in-place
version counters of those tensors after every call and vouches for the state (`reuse_rowdots`) only when they have not
moved; any torch in-place edit (warm restart, projection, rescaling A) makes the next epoch recompute.  Writes that
bypass torch (a raw pointer handed to another library) are not seen: call `state.invalidate()` after those.

Objective monitor / stopping.  The reference has `stop(state) = false` (SVRG.jl:55,70) and
its tests compute the cost outside (test_lasso.jl:45-47).  Here `solver(x0; ..., stop=callable)` ends the loop at the first
yielded state for which `stop(state)` is true (IterationTools.halt semantics), and `state.objective` is
(1/N) sum f_i(x) + g(x): for SVRG and LFinito it rides on the full pass the iteration makes anyway (the values
`gradient!` returns and the reference discards), taken at z_full; for the other iterables it costs one extra sweep.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch

from ._lib import ERR_UNSUPPORTED, CiaoError
from .device import Context, PackedF, default_context, torch_dtype
from . import host_route as HR
from .operators import UnpackableOperator, pack_F, pack_g, pack_sharing_F, require_packable
from .sampling import IndexStream

# Problems packed from host operators get their rows padded with zero columns to whole 16-byte chunks (see _Iterable.__init__)
PAD_FEATURES = True

__all__ = ["SVRG", "SAGA", "SAG", "Finito", "Proshi", "iterator", "solution", "solve_together"]


def _pick(greek, ascii_, name):
    if greek is not None and ascii_ is not None:
        raise TypeError(f"give {name} once")
    return greek if greek is not None else ascii_


_CPLX = {torch.float32: torch.complex64, torch.float64: torch.complex128}


def _x0_to_device(x0, dtype):
    """Return (device vector of the real type, was_numpy, is_complex).  A real device tensor of the right dtype is used as is
    (no copy).  Complex x0 (the reference's C <: RealOrComplex{R}, CIAOAlgorithms.jl:3) becomes its (re, im) pairs -- a view,
    reinterpret(R, x0) in Julia terms -- and the solution is handed back as complex."""
    if isinstance(x0, torch.Tensor):
        if x0.is_complex():
            if x0.dtype != _CPLX[dtype]:
                raise TypeError(f"x0 has dtype {x0.dtype} but the solver's real type is {dtype} (no silent promotion)")
            t = x0 if x0.is_cuda else x0.cuda()
            return torch.view_as_real(t.contiguous().view(-1)).reshape(-1), False, True
        if x0.dtype != dtype:
            raise TypeError(f"x0 has dtype {x0.dtype} but the solver's real type is {dtype} (no silent promotion)")
        t = x0 if x0.is_cuda else x0.cuda()
        return t.contiguous().view(-1), False, False
    a = np.asarray(x0)
    if np.iscomplexobj(a):
        if torch_dtype(a.real.dtype) != dtype:
            raise TypeError(f"x0 has dtype {a.dtype} but the solver's real type is {dtype} (no silent promotion)")
        return torch.from_numpy(np.ascontiguousarray(a).reshape(-1).view(a.real.dtype)).cuda(), True, True
    if torch_dtype(a.dtype if a.dtype.kind == "f" else np.float64) != dtype:
        raise TypeError(f"x0 has dtype {a.dtype} but the solver's real type is {dtype} (no silent promotion)")
    return torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).cuda(), True, False


def _maxL(Lc):
    return float(np.max(np.asarray(Lc.detach().cpu() if isinstance(Lc, torch.Tensor) else Lc)))


class _Iterable:
    """Common part of the four iterables: problem packing and the Python iteration protocol."""

    def __init__(self, R, F, g, x0, N, ctx, stream):
        if N is None:
            raise TypeError("N (number of terms in the finite sum) is required")
        self.R = torch_dtype(R)
        self.x0 = x0  # NOT copied: `iter.x0 === x0` (test/test_lasso.jl:182)
        self.N = int(N)
        self._x0_dev, self._numpy, self._complex = _x0_to_device(x0, self.R)
        self.d = self._x0_dev.numel()                   # reals: twice the length of a complex x0
        self.ctx = ctx if ctx is not None else default_context()
        # Feature padding.  The fast chain kernels stream rows of whole 16-byte chunks from 16-byte aligned addresses; a row of, say,
        # 1001 Float64 is neither, and runs the register-ring chains at 2-3.7x the time per update (d = 51 / 785 / 2049 fp64: 0.28 /
        # 0.57 / 1.69 us against 0.13-0.17 / 0.27-0.31 / 0.45).  When the problem is PACKED HERE from host operators (not a device
        # matrix the caller laid out), it is packed with dp - d zero columns, dp = d rounded up to whole chunks: the extra
        # coordinates multiply nothing and start at zero, so the d real ones evolve as without them (to rounding: another kernel
        # adds them in another order).  Every state vector is a length-d VIEW of a dp-long buffer -- what the caller sees has the
        # reference's shapes, identities and aliasing -- and the device calls take the view (device.Context._vec reads on to dp).
        self.dp = self.d
        vec = 16 // (8 if self.R == torch.float64 else 4)
        if PAD_FEATURES and not self._complex and not isinstance(F, PackedF) and self.d >= 1 and self.d % vec != 0 and self._pads_features:
            self.dp = (self.d + vec - 1) // vec * vec
            base = torch.zeros(self.dp, dtype=self.R, device=self._x0_dev.device)
            base[:self.d] = self._x0_dev
            self._x0_dev = base[:self.d]
        self.F = pack_F(F, self.N, self.d, self.R, self._x0_dev.device, complex_pairs=self._complex, pad_to=self.dp)
        if self.F.N_total != self.N:
            raise ValueError(f"F holds N_total={self.F.N_total} terms but N={self.N}")
        if self._complex != bool(getattr(self.F, "complex", False)) and self.F.loss != 2:
            raise TypeError("x0 and F must both be complex or both be real (CIAOAlgorithms.jl:3: one type T for the problem)")
        self.g = pack_g(g, self.d, self.R, self._x0_dev.device, complex_pairs=self._complex, pad_to=self.dp)
        self.stream = stream if stream is not None else IndexStream(0)
        self._state = None
        self._started = False
        self.monitor = False      # set by the functor when `verbose` or a `stop` callback wants objective values
        self._obj = None          # device float64[3]: {F, (1/N) sum f_i, g} of the last full pass

    # objective monitor: the ctx may be shared between iterables, so it is armed around each call that makes a full pass
    def _monitor_on(self):
        if not self.monitor:
            return
        if self._obj is None:
            self._obj = torch.full((3,), float("nan"), dtype=torch.float64, device=self._x0_dev.device)
        self.ctx.set_monitor(self.g, self._obj)

    def _monitor_off(self):
        if self.monitor:
            self.ctx.set_monitor(None, None)

    def objective(self, state):
        """(1/N) sum_i f_i(x) + g(x).  With the monitor armed, SVRG / LFinito read what their own full pass left (x = z_full);
        otherwise one extra sweep at solution(state)."""
        if self.monitor and self._obj is not None and self._rides_on_full_pass:
            self.ctx.synchronize()
            return float(self._obj[0].item())
        if isinstance(state, Proshi_basic_state):
            raise NotImplementedError("objective of a sharing problem (1/N) sum f_i(x_i) + g(sum x_i) is not on the device path")
        return self.ctx.objective(self.F, self.g, solution(state))

    _rides_on_full_pass = False

    def _draw(self, N, m):
        """m uniform indices of the injected stream -> (indices, last index on the host).  Long runs are generated ON THE DEVICE
        when the stream can (sampling.IndexStream: an epoch of 10^7 draws costs the host about a second and 80 MB of PCIe);
        short ones and replayed streams (FixedStream) on the host."""
        if m >= 4096 and hasattr(self.stream, "rand_indices_device") and 0 < N < (1 << 32):
            return self.stream.rand_indices_device(self.ctx, N, m)
        idx = self.stream.rand_indices(N, m)
        return idx, (int(idx[-1]) if m > 0 else None)

    _pads_features = True    # (the adaptive iterable keeps the caller's d: its per-sample probes and meta are laid out for it)

    def _new(self):
        """A state d-vector: with feature padding a length-d view of a zeroed dp-long buffer (the padding coordinates start at zero)."""
        if self.dp == self.d:
            return torch.empty(self.d, dtype=self.R, device=self._x0_dev.device)
        return torch.zeros(self.dp, dtype=self.R, device=self._x0_dev.device)[:self.d]

    def _new_table(self):
        """The N x d table of SAGA / Finito (this rank's rows): dp columns with feature padding (the kernels write them: zeros)."""
        return torch.empty((self.F.N, self.dp), dtype=self.R, device=self._x0_dev.device)

    # Python iteration protocol = Base.iterate(iter) then Base.iterate(iter, state)
    def __iter__(self):
        self._state, self._started = None, False
        return self

    def __next__(self):
        if not self._started:
            self._started = True
            self._state = self._init()
            if self._state is None:  # invalid configuration: `return nothing` ends the iteration
                raise StopIteration
            return self._state
        if self._state is None:
            raise StopIteration
        done = self._step(self._state, 1)   # None = all done; adaptive Finito may end early (`return nothing`)
        if done is not None and done < 1:
            self._state = None
            raise StopIteration
        return self._state


# ======================================================================================================================
# SVRG  (SVRG_basic.jl)
# ======================================================================================================================
class _State:
    _it = None

    @property
    def objective(self):
        """(1/N) sum_i f_i(x) + g(x) -- see the module docstring (objective monitor)."""
        return self._it.objective(self)


class SVRG_basic_state(_State):
    def __init__(self, γ, m, av, z, z_full, w):
        self.γ, self.m, self.av, self.z, self.z_full, self.w = γ, m, av, z, z_full, w
        self._tok = None   # torch version counters of (z_full, A, b) after the library last wrote the state

    gamma = property(lambda self: self.γ)

    def invalidate(self):
        """Forget that the library knows a_i'z_full: call after editing z_full / A / b behind torch's back."""
        self._tok = None


class SVRG_basic_iterable(_Iterable):
    def __init__(self, R, F, g, x0, N, L, μ, γ, m, plus, ctx=None, stream=None, shards=None):
        super().__init__(R, F, g, x0, N, ctx, stream)
        self.L, self.μ, self.γ, self.m, self.plus = L, μ, γ, m, plus
        self.shards = shards   # parallel.ShardGroup: the one chain runs on its owner and pulls the other shards' rows over xGMI
        if (self.F.row0 != 0 or self.F.N != self.N) and shards is None:
            raise ValueError("the SVRG inner cycle is a sequential chain: it needs the whole problem on one device, or a "
                             "parallel.ShardGroup (shards=...) through which one rank reads the other shards' rows")

    def _init(self):                                                       # SVRG_basic.jl:30-69
        N = self.N
        m = N if self.m is None else self.m
        if self.γ is None:
            if self.plus:
                warnings.warn("provide a stepsize γ")
                return None
            if self.L is None or self.μ is None:
                warnings.warn("smoothness or convexity parameter absent")
                return None
            L_M, μ_M = _maxL(self.L), _maxL(self.μ)
            γ = 1 / (10 * L_M)
            rho = (1 + 4 * L_M * γ ** 2 * μ_M * (N + 1)) / (μ_M * γ * N * (1 - 4 * L_M * γ))
            if rho >= 1:
                warnings.warn("convergence condition violated...provide a stepsize!")
        else:
            γ = self.γ
        av, z, z_full, w = self._new(), self._new(), self._new(), self._new()
        if self.shards is not None:
            self.shards.install(self.F)
        self._monitor_on()
        self.ctx.svrg_init(self.F, self._x0_dev, av, z, z_full, w)         # :57-66
        self._monitor_off()
        st = SVRG_basic_state(float(γ), int(m), av, z, z_full, w)
        st._it = self
        st._tok = self._versions(st)
        return st

    _rides_on_full_pass = True

    def _versions(self, st):
        """torch's in-place version counters of everything the cached a_i'z_full depend on (module docstring)."""
        return (st.z_full._version, st.z_full.data_ptr(),
                -1 if self.F.A is None else self.F.A._version, -1 if self.F.b is None else self.F.b._version)

    def _step(self, st, n):                                                # SVRG_basic.jl:71-96
        for _ in range(n):
            idx, _ = self._draw(self.N, st.m)                              # :73
            fresh = self.shards is None and st._tok is not None and st._tok == self._versions(st)
            self._monitor_on()
            self.ctx.svrg_iterate(self.F, self.g, st.γ, idx, self.plus, st.av, st.z, st.z_full, st.w, reuse_rowdots=fresh)
            self._monitor_off()
            st._tok = self._versions(st)
            if self.plus:
                st.m *= 2                                                  # :93


# ======================================================================================================================
# SAGA / SAG  (SAGA_basic.jl)
# ======================================================================================================================
class SAGA_basic_state(_State):
    def __init__(self, s, γ, av, z):
        self.s, self.γ, self.av, self.z, self.ind = s, γ, av, z, 0

    gamma = property(lambda self: self.γ)


class SAGA_basic_iterable(_Iterable):
    def __init__(self, R, F, g, x0, N, L, γ, SAG, ctx=None, stream=None, shards=None):
        super().__init__(R, F, g, x0, N, ctx, stream)
        self.L, self.γ, self.SAG = L, γ, SAG
        self.shards = shards   # parallel.ShardGroup (see SVRG_basic_iterable): data rows AND table rows of the other shards are remote
        if (self.F.row0 != 0 or self.F.N != self.N) and shards is None:
            raise ValueError("SAGA steps are a sequential chain: they need the whole problem on one device, or a "
                             "parallel.ShardGroup (shards=...)")

    def _init(self):                                                       # SAGA_basic.jl:26-51
        if self.γ is None:
            if self.L is None:
                warnings.warn("smoothness parameter absent")
                return None
            L_M = _maxL(self.L)
            γ = 1 / (16 * L_M) if self.SAG else 1 / (3 * L_M)
        else:
            γ = self.γ
        s = self._new_table()                                              # this rank's rows of the N x d table
        av, z = self._new(), self._new()
        if self.shards is not None:
            self.shards.install(self.F, table=s)
        self.ctx.saga_init(self.F, self.g, γ, self._x0_dev, s, av, z)      # :41-48
        st = SAGA_basic_state(s, float(γ), av, z)
        st._it = self
        return st

    def _step(self, st, n):                                                # SAGA_basic.jl:53-68, n consecutive calls
        idx, last = self._draw(self.N, n)                                  # :55 (one draw per iteration)
        self.ctx.saga_steps(self.F, self.g, st.γ, self.SAG, idx, st.s, st.av, st.z)
        st.ind = last + 1 if n > 0 else st.ind                             # 1-based like the reference's state.ind


# ======================================================================================================================
# Finito / MISO  (Finito_basic.jl) and LFinito (Finito_LFinito.jl)
# ======================================================================================================================
def _finito_gammas(it):
    """Finito_basic.jl:61-74 / Finito_LFinito.jl:51-63 -> device N-vector of stepsizes, or None (with the @warn)."""
    R, N, dev = it.R, it.N, it._x0_dev.device
    np_R = np.float64 if R == torch.float64 else np.float32
    if it.γ is None:
        if it.L is None:
            warnings.warn("--> smoothness parameter absent")
            return None
        if np.ndim(it.L) == 0 and not isinstance(it.L, torch.Tensor):
            val = np_R(it.α) * np_R(N) / np_R(it.L)
            return torch.full((it.F.N,), float(val), dtype=R, device=dev)
        Lh = it.L.to(device=dev, dtype=R) if isinstance(it.L, torch.Tensor) else torch.from_numpy(np.asarray(it.L, dtype=np_R)).to(dev)
        Lh = it.F.local_slice(Lh) if Lh.numel() == N and it.F.N != N else Lh
        return ((np_R(it.α) * np_R(N)) / Lh).contiguous()
    if np.ndim(it.γ) == 0 and not isinstance(it.γ, torch.Tensor):
        return torch.full((it.F.N,), float(np_R(it.γ)), dtype=R, device=dev)
    gh = it.γ.to(device=dev, dtype=R) if isinstance(it.γ, torch.Tensor) else torch.from_numpy(np.asarray(it.γ, dtype=np_R)).to(dev)
    gh = it.F.local_slice(gh) if gh.numel() == N and it.F.N != N else gh
    return gh.contiguous()


def _static_batch(N, r, j):
    """j-th (0-based) static batch of Finito_basic.jl:52-58: contiguous block of r, the last one may be shorter."""
    lo = r * j
    return np.arange(lo, min(lo + r, N), dtype=np.int64)


def _localise(it, batch):
    """Keep the members of a (global-index) batch that this rank owns, as local row indices (parallel.py)."""
    return it.F.localise(batch)


def _pack_batches(batches):
    bptr = np.zeros(len(batches) + 1, np.int64)
    np.cumsum([len(b) for b in batches], out=bptr[1:])
    bidx = np.concatenate(batches) if batches else np.zeros(0, np.int64)
    return bptr, bidx


def _static_batches_packed(N, r, js):
    """(bptr, bidx) of the static batches number js[0], js[1], ... (Finito_basic.jl:52-58), without a Python loop."""
    js = np.asarray(js, dtype=np.int64)
    starts = r * js
    lens = np.minimum(r, N - starts)
    bptr = np.zeros(js.size + 1, np.int64)
    np.cumsum(lens, out=bptr[1:])
    bidx = np.arange(bptr[-1], dtype=np.int64) + np.repeat(starts - bptr[:-1], lens)
    return bptr, bidx


def _static_blocks(it, js):
    """The static batches number js[0], js[1], ... (Finito_basic.jl:52-58) as this rank's contiguous LOCAL row blocks
    (first, length): index-free, and arithmetic on whole arrays whatever the sharding."""
    js = np.asarray(js, dtype=np.int64)
    lo = it.batch * js
    hi = np.minimum(lo + it.batch, it.N)
    return it.F.local_blocks(lo, hi)


def _next_static_numbers(it, st, n):
    """The 0-based static batch numbers of the next n iterations for sweeping 2 (cyclic, Finito_basic.jl:99: the first
    step uses batch 2) and 3 (a fresh randperm(d) whenever a pass is complete, :100-108; the first pass is the identity
    order); advances st.idxr / st.idx / st.inds exactly as n single steps would."""
    d = st.d
    if it.sweeping == 2:
        js = (st.idxr + np.arange(n, dtype=np.int64)) % d              # idxr is 1-based: next 0-based number = idxr % d
        st.idxr = int(js[-1]) + 1
        return js
    out, left = [], n
    while left > 0:
        if st.idx == d:
            st.inds = it.stream.randperm(d)
            st.idx = 0
        take = min(left, d - st.idx)
        out.append(np.asarray(st.inds[st.idx:st.idx + take], dtype=np.int64))
        st.idx += take
        left -= take
    js = np.concatenate(out)
    st.idxr = int(js[-1]) + 1
    return js


def _next_batches_packed(it, st, n):
    """(bptr, bidx) of the next n iterations' batches (Finito_basic.jl:95-108 / ProShI_basic.jl:95-107 applied n times).
    On one device nothing here loops in Python per iteration: a device batch takes 0.5-50 us."""
    N, r = it.N, it.batch
    whole = it.F.N == it.F.N_total and getattr(it.F, "cyclic", None) is None
    if it.sweeping == 1:
        if whole and hasattr(it.stream, "sample_batches"):
            return np.arange(n + 1, dtype=np.int64) * r, it.stream.sample_batches(N, r, n).reshape(-1)   # :97
        return _pack_batches([_localise(it, it.stream.sample_without_replacement(N, r)) for _ in range(n)])
    js = _next_static_numbers(it, st, n)
    if whole:
        return _static_batches_packed(N, r, js)
    return _pack_batches([_localise(it, _static_batch(N, r, int(j))) for j in js])


class FINITO_basic_state(_State):
    def __init__(self, s, γ, hat_γ, av, z, d):
        self.s, self.γ, self.hat_γ, self.av, self.z, self.d = s, γ, hat_γ, av, z, d
        self.ind = None
        self.idxr, self.idx, self.inds = 1, 0, np.arange(d, dtype=np.int64)   # Finito_basic.jl:37-40 (1-based idxr)

    hat_gamma = property(lambda self: self.hat_γ)


class FINITO_basic_iterable(_Iterable):
    def __init__(self, R, F, g, x0, N, L, γ, sweeping, batch, α, ctx=None, stream=None):
        super().__init__(R, F, g, x0, N, ctx, stream)
        self.L, self.γ, self.sweeping, self.batch, self.α = L, γ, int(sweeping), int(batch), α
        if self.batch < 1:
            raise ValueError("batch size must be >= 1")

    def _init(self):                                                       # Finito_basic.jl:44-89
        N, r = self.N, self.batch
        d_b = -(-N // r) if N > 0 else 0                                   # cld(N, r)  :59
        gam = _finito_gammas(self)                                         # :61-74
        if gam is None:
            return None
        hat_γ = self.ctx.hat_gamma(gam)                                    # :82
        s = self._new_table()
        av, z = self._new(), self._new()
        self.ctx.finito_init(self.F, self.g, gam, hat_γ, self._x0_dev, s, av, z)   # :76-84
        st = FINITO_basic_state(s, gam, hat_γ, av, z, d_b)
        st._it = self
        return st

    def _step(self, st, n):                                                # Finito_basic.jl:109-118, n iterations
        if self.sweeping != 1:   # static batches are contiguous row blocks: no index array at all (:99-108)
            first, length = _static_blocks(self, _next_static_numbers(self, st, n))
            self.ctx.finito_steps_blocks(self.F, self.g, st.γ, st.hat_γ, first, length, st.s, st.av, st.z)
            return
        bptr, bidx = _next_batches_packed(self, st, n)
        self.ctx.finito_steps(self.F, self.g, st.γ, st.hat_γ, bptr, bidx, st.s, st.av, st.z)


class FINITO_LFinito_state(_State):
    def __init__(self, γ, hat_γ, av, d, z, z_full):
        self.γ, self.hat_γ, self.av, self.d, self.z, self.z_full = γ, hat_γ, av, d, z, z_full
        self.inds = np.arange(d, dtype=np.int64)

    hat_gamma = property(lambda self: self.hat_γ)


class FINITO_LFinito_iterable(_Iterable):
    def __init__(self, R, F, g, x0, N, L, γ, sweeping, batch, α, ctx=None, stream=None):
        super().__init__(R, F, g, x0, N, ctx, stream)
        self.L, self.γ, self.sweeping, self.batch, self.α = L, γ, int(sweeping), int(batch), α
        if self.batch < 1:
            raise ValueError("batch size must be >= 1")

    def _init(self):                                                       # Finito_LFinito.jl:40-76
        N, r = self.N, self.batch
        gam = _finito_gammas(self)                                         # :51-63
        if gam is None:
            return None
        hat_γ = self.ctx.hat_gamma(gam)                                    # :66
        av, z, z_full = self._new(), self._new(), self._new()
        self._monitor_on()
        self.ctx.lfinito_init(self.F, hat_γ, self._x0_dev, av, z, z_full)  # :67-72
        self._monitor_off()
        st = FINITO_LFinito_state(gam, hat_γ, av, -(-N // r) if N > 0 else 0, z, z_full)
        st._it = self
        return st

    _rides_on_full_pass = True

    def _step(self, st, n):                                                # Finito_LFinito.jl:78-103
        for _ in range(n):
            if self.sweeping == 3:
                st.inds = self.stream.randperm(st.d)                       # :89
            # the batches are always the static contiguous blocks (:44-49), visited in the order st.inds (:90)
            first, length = _static_blocks(self, st.inds)
            self._monitor_on()
            self.ctx.lfinito_iterate_blocks(self.F, self.g, st.γ, st.hat_γ, first, length, st.av, st.z, st.z_full)
            self._monitor_off()


class FINITO_adaptive_state(_State):
    """Finito_adaptive.jl:13-29.  For the row-structured f_i of this path grad f_i = c_i a_i, so the reference's N x d
    gradient table `∇f` is the column c_i of `meta` (N x 4 copies x 4: c_i, f_i(x_i), γ_i, a_i'x_i); γ and fi_x are views."""

    def __init__(self, s, meta, hat_γ_dev, av, z, N):
        self.s, self.meta, self.hat_γ_dev, self.av, self.z = s, meta, hat_γ_dev, av, z
        self.ind = np.arange(N, dtype=np.int64)
        self.idx, self.idxr = 0, 0
        self.trials = 0

    γ = property(lambda self: self.meta[:, 0, 2])
    fi_x = property(lambda self: self.meta[:, 0, 1])
    hat_γ = property(lambda self: float(self.hat_γ_dev.item()))
    hat_gamma = hat_γ


class FINITO_adaptive_iterable(_Iterable):
    """FINITO_adaptive_iterable (Finito_adaptive.jl): per-sample backtracking on γ_i; no minibatch (:162)."""

    _pads_features = False   # its Lipschitz probe (x0 .+ 1, divided by sqrt(d), :86-88) counts the coordinates

    def __init__(self, R, F, g, x0, N, L, tol, tol_b, sweeping, α, ctx=None, stream=None, shards=None):
        super().__init__(R, F, g, x0, N, ctx, stream)
        self.L, self.tol, self.tol_b, self.sweeping, self.α = L, tol, tol_b, int(sweeping), α
        self.shards = shards   # parallel.ShardGroup: the owner's chain reads and writes the other shards' rows, table rows and scalars
        if (self.F.row0 != 0 or self.F.N != self.N) and shards is None:
            raise ValueError("adaptive Finito is a sequential chain: on a row-sharded problem it needs a parallel.ShardGroup (shards=...)")

    def _init(self):                                                       # Finito_adaptive.jl:59-98
        dev = self._x0_dev.device
        s = torch.empty((self.F.N, self.d), dtype=self.R, device=dev)      # (a rank's own rows on a row-sharded problem)
        meta = torch.empty((self.F.N, 4, 4), dtype=self.R, device=dev)
        hg = torch.empty(1, dtype=self.R, device=dev)
        av, z = self._new(), self._new()
        if self.shards is not None:
            return self._init_sharded(s, meta, hg, av, z)
        self.ctx.afinito_init(self.F, self.g, self.α, self._x0_dev, s, meta, av, z, hg)
        try:
            self.ctx.synchronize()
        except CiaoError as e:
            if e.status != ERR_UNSUPPORTED:
                raise
            # Finito_adaptive.jl:78-85: for samples whose probe gradient at x0 .+ 1 equals the one at x0 the reference draws
            # random probe points x0 .+ rand(t*[-1,1]), t = 1, 2, 4, ...  The draws come from the injected stream, sample by
            # sample in increasing i (the order in which the reference's loop meets them); then the init pass is repeated
            # with every stepsize known.
            np_R = np.float64 if self.R == torch.float64 else np.float32
            gam = meta[:, 0, 2].clone()
            eps = float(np.finfo(np_R).eps)
            n_x0 = self.d // 2 if self._complex else self.d     # length(x0): complex entries; the draws are real and go to the real parts
            for i in torch.nonzero(gam < 0).flatten().tolist():
                t = 1
                while True:
                    print("initial upper bound for L too small")                                   # :79
                    signs = torch.from_numpy(self.stream.rand_signs(n_x0).astype(np_R)).to(dev)     # :80
                    nmg = self.ctx.afinito_probe(self.F, i, self._x0_dev, signs, float(t))          # :81-82
                    t *= 2                                                                          # :83
                    if not nmg < eps:
                        break
                L_int = np.float64(np_R(nmg)) / (np.float64(t) * np.sqrt(np.float64(n_x0)))        # :86 (Float64 whatever R, :73)
                L_int /= np.float64(self.N)                                                         # :87
                gam[i] = float(np_R(np.float64(np_R(self.α)) / L_int))                              # :88
            self.ctx.afinito_init(self.F, self.g, self.α, self._x0_dev, s, meta, av, z, hg, gam_override=gam.contiguous())
            self.ctx.synchronize()
        st = FINITO_adaptive_state(s, meta, hg, av, z, self.N)
        st._it = self
        return st

    def _init_sharded(self, s, meta, hg, av, z):
        """The same init on a row-sharded problem: every rank sweeps its own rows (table rows, scalars; sum and hat_gamma all-reduced),
        the re-probes of :78-85 go through the ranks' replicated streams in increasing GLOBAL i -- every rank draws, the rank that
        holds the sample probes, its result decides for all how many draws the sample takes."""
        import torch.distributed as dist
        grp = self.shards.group
        rank = dist.get_rank(grp)
        self.shards.install(self.F, table=s, meta=meta)
        self.ctx.afinito_init(self.F, self.g, self.α, self._x0_dev, s, meta, av, z, hg)
        try:
            self.ctx.synchronize()
        except CiaoError as e:
            if e.status != ERR_UNSUPPORTED:
                raise
        gam = meta[:, 0, 2].clone()
        mine = [self.F.row0 + i for i in torch.nonzero(gam < 0).flatten().tolist()]
        every = [None] * dist.get_world_size(grp)
        dist.all_gather_object(every, mine, group=grp)
        todo = sorted((gi, r) for r, rows in enumerate(every) for gi in rows)
        if todo:
            np_R = np.float64 if self.R == torch.float64 else np.float32
            eps = float(np.finfo(np_R).eps)
            dev = self._x0_dev.device
            for gi, holder in todo:
                t = 1
                while True:
                    if rank == 0:
                        print("initial upper bound for L too small")                                # :79
                    signs = self.stream.rand_signs(self.d).astype(np_R)                             # :80 (every rank: the streams stay in step)
                    box = [None]
                    if rank == holder:
                        box[0] = self.ctx.afinito_probe(self.F, gi - self.F.row0, self._x0_dev, torch.from_numpy(signs).to(dev), float(t))
                    dist.broadcast_object_list(box, src=dist.get_global_rank(grp, holder) if grp is not None else holder, group=grp)
                    nmg = box[0]
                    t *= 2                                                                          # :83
                    if not nmg < eps:
                        break
                if rank == holder:
                    L_int = np.float64(np_R(nmg)) / (np.float64(t) * np.sqrt(np.float64(self.d)))   # :86
                    L_int /= np.float64(self.N)                                                     # :87
                    gam[gi - self.F.row0] = float(np_R(np.float64(np_R(self.α)) / L_int))           # :88
            self.ctx.afinito_init(self.F, self.g, self.α, self._x0_dev, s, meta, av, z, hg, gam_override=gam.contiguous())
            self.ctx.synchronize()
        st = FINITO_adaptive_state(s, meta, hg, av, z, self.N)
        st._it = self
        return st

    def _next_indices(self, st, n):                                        # :104-116 applied n times (0-based result)
        N = self.N
        if self.sweeping == 1:
            idx = self.stream.rand_indices(N, n)
        elif self.sweeping == 2:
            idx = (st.idxr + np.arange(n, dtype=np.int64)) % N             # idxr is 1-based: the next 0-based index is idxr % N
        else:
            out, left = [], n
            while left > 0:
                if st.idx == N:                                            # a pass is complete: fresh randperm (:109-112)
                    st.ind = self.stream.randperm(N)
                    st.idx = 0
                take = min(left, N - st.idx)
                out.append(np.asarray(st.ind[st.idx:st.idx + take], dtype=np.int64))
                st.idx += take
                left -= take
            idx = np.concatenate(out)
        st.idxr = int(idx[-1]) + 1
        return idx

    def _step(self, st, n):                                                # :118-150, n iterations in one launch
        idx = self._next_indices(st, n)
        done, trials = self.ctx.afinito_steps(self.F, self.g, self.α, self.tol_b, idx, st.s, st.meta, st.av, st.z, st.hat_γ_dev)
        st.trials += trials
        if done < n:                                                       # :121-124  @warn + return nothing
            warnings.warn("parameter `γ` became too small")
        return done


# ======================================================================================================================
# ProShI  (src/algorithms/ProShI/ProShI_basic.jl) -- sharing problems: the solution is the whole N x d table
# ======================================================================================================================
class Proshi_basic_state(_State):
    def __init__(self, it, s, γ, hat_γ, av, z, d):
        self._it, self.s, self.γ, self.hat_γ, self.av, self.z, self.d = it, s, γ, hat_γ, av, z, d
        self.idxr, self.idx, self.inds = 1, 0, np.arange(d, dtype=np.int64)   # ProShI_basic.jl:37-39

    hat_gamma = property(lambda self: self.hat_γ)


class Proshi_basic_iterable(_Iterable):
    def __init__(self, R, F, g, x0, N, L, γ, sweeping, batch, α, ctx=None, stream=None):
        if N is None:
            raise TypeError("N (number of agents) is required")
        self.R = torch_dtype(R)
        self.x0, self.N = x0, int(N)
        self._x0_dev, self._numpy, self._complex = _x0_to_device(x0, self.R)
        if self._complex:
            raise TypeError("complex agents are outside the ProShI device path")
        self.d = self._x0_dev.numel()
        self.dp = self.d
        self.ctx = ctx if ctx is not None else default_context()
        self.F = pack_sharing_F(F, self.N, self.d, self.R, self._x0_dev.device)
        self.g = pack_g(g, self.d, self.R, self._x0_dev.device)
        self.stream = stream if stream is not None else IndexStream(0)
        self._state, self._started = None, False
        self.monitor, self._obj = False, None
        self.L, self.γ, self.sweeping, self.batch, self.α = L, γ, int(sweeping), int(batch), α

    def _init(self):                                                       # ProShI_basic.jl:44-89
        N, r = self.N, self.batch
        gam = _finito_gammas(self)                                         # :61-74 (the same rule as Finito)
        if gam is None:
            return None
        dev = self._x0_dev.device
        s = torch.empty((self.F.N, self.d), dtype=self.R, device=dev)
        av, z = self._new(), self._new()
        hg = torch.empty(1, dtype=self.R, device=dev)
        self.ctx.proshi_init(self.F, self.g, gam, self._x0_dev, s, av, z, hg)   # :76-87
        return Proshi_basic_state(self, s, gam, float(hg.item()), av, z, -(-N // r) if N > 0 else 0)

    def _step(self, st, n):                                                # :109-121
        if self.sweeping != 1:   # static batches: contiguous blocks of agents (:50-57)
            first, length = _static_blocks(self, _next_static_numbers(self, st, n))
            self.ctx.proshi_steps_blocks(self.F, self.g, st.γ, st.hat_γ, first, length, st.s, st.av, st.z)
            return
        bptr, bidx = _next_batches_packed(self, st, n)
        self.ctx.proshi_steps(self.F, self.g, st.γ, st.hat_γ, bptr, bidx, st.s, st.av, st.z)


# ======================================================================================================================
# solution(state)   -- SVRG_basic.jl:99, SAGA_basic.jl:71, Finito_basic.jl:123, Finito_LFinito.jl:105
# ======================================================================================================================
def solution(state):
    if state is None:
        raise TypeError("solution(nothing): no method matching solution(::Nothing) -- the iterable ended before yielding "
                        "a state (invalid configuration, see the warning above)")
    if getattr(state, "backend", None) == "host":                          # the explicit host route (host_route.py)
        return HR.host_solution(state)
    if isinstance(state, Proshi_basic_state):                              # ProShI_basic.jl:127-132: shifts s IN PLACE
        state._it.ctx.proshi_solution(state._it.F, state.γ, state.z, state.s)
        return state.s
    return state.z_full if isinstance(state, SVRG_basic_state) else state.z


# ======================================================================================================================
# L3: solver structs + functors + iterator()
# ======================================================================================================================
def _route(device_factory, host_factory, fallback, backend, F=None, g=None, x0=None):
    """The device iterable; or the HOST route when the caller forces it (backend="host") or allows it (fallback="host") and
    F / g are not a family the device path packs (operators.UnpackableOperator -- nothing else is caught).  Never silently:
    host_route.announce() warns, and the states carry backend == "host"."""
    if backend not in (None, "device", "host") or fallback not in (None, "host"):
        raise ValueError('backend is "device" (default) or "host"; fallback is None (default) or "host"')
    if backend == "host":
        HR.announce('backend="host" was requested')
        return host_factory()
    try:
        require_packable(F, g, complex_x0=bool(np.iscomplexobj(x0.detach().cpu().numpy() if isinstance(x0, torch.Tensor) else x0)))
        return device_factory()
    except UnpackableOperator as e:
        if fallback != "host":
            raise
        HR.announce(f'fallback="host" was given and the device path cannot pack the problem ({str(e).split(".")[0]})')
        return host_factory()


class _Solver:
    _chunk = 1 << 20        # iterations per device launch in the functor's fast path ...
    _chunk_samples = 1 << 24   # ... capped so that one chunk's batch indices stay within 128 MiB (host scratch + HBM)

    def _drive(self, it, maxit, disp, stop=None, check_every=1):
        """The functor's loop (SVRG.jl:69-83): take(halt(iter, stop), maxit), optional printing, return (solution, num_iters).

        The reference's `stop(state) = false` (SVRG.jl:55); a callable `stop` ends the loop at the first yielded state for
        which it is true (IterationTools.halt yields that state and then stops).  `check_every=k` evaluates it only at every
        k-th state, so that iterables whose iteration is one device step (SAGA, Finito) still run k steps per launch.
        Nothing observes the intermediate states unless `verbose` / `stop`, so those iterations are issued `chunk` at a
        time; the yielded-state semantics of `iterator(...)` are unchanged.
        """
        it.monitor = bool(self.verbose or stop is not None)
        it_obj = iter(it)
        num_iters, state = 0, None
        try:
            state = next(it_obj)
            num_iters = 1
        except StopIteration:
            pass
        if state is not None:
            if self.verbose and num_iters % self.freq == 0:
                disp(num_iters, state)
            halted = stop is not None and check_every == 1 and bool(stop(state))
            while num_iters < maxit and not halted:
                n = maxit - num_iters
                if self.verbose:
                    n = min(n, self.freq - num_iters % self.freq)
                if stop is not None:
                    n = min(n, check_every - num_iters % check_every)
                chunk = self._chunk
                r = getattr(it, "batch", 1)
                if r > 1:
                    chunk = max(1, min(chunk, self._chunk_samples // r))
                n = min(n, chunk if it._chunkable else 1)
                done = it._step(state, n)
                done = n if done is None else done
                num_iters += done
                if self.verbose and num_iters % self.freq == 0:
                    disp(num_iters, state)
                if done < n:   # the iterable ended early (adaptive Finito: stepsize collapsed, Finito_adaptive.jl:121-124)
                    break
                if stop is not None and num_iters % check_every == 0:
                    halted = bool(stop(state))
            if self.verbose and num_iters % self.freq != 0:
                disp(num_iters, state)
        sol = solution(state)
        it.ctx.synchronize()
        if getattr(it, "backend", None) == "host":   # host route: numpy state; a torch x0 gets a (CPU) tensor back
            out = sol.reshape(np.shape(it.x0))
            return (torch.from_numpy(out) if isinstance(it.x0, torch.Tensor) else out), num_iters
        if isinstance(state, Proshi_basic_state):   # Array{Array{R,1}}: one x_i per agent (test_sharing.jl:45)
            return ([row for row in sol.cpu().numpy()] if it._numpy else sol), num_iters
        if it._complex:   # hand the (re, im) pairs back as the complex type x0 came in
            if it._numpy:
                h = sol.cpu().numpy()
                return h.view(np.complex128 if h.dtype == np.float64 else np.complex64).reshape(np.shape(it.x0)), num_iters
            return torch.view_as_complex(sol.view(-1, 2)).reshape(it.x0.shape), num_iters
        return (sol.cpu().numpy().reshape(np.shape(it.x0)) if it._numpy else sol), num_iters


def _split_drive_kw(kw):
    """Functor keywords that steer the loop rather than the iterable: stop=callable(state)->bool, check_every=k."""
    return kw.pop("stop", None), int(kw.pop("check_every", 1))


def _disp(fmt_attr):
    """`@printf("%5d | %.3e\n", it, γ)` of the reference (SVRG.jl:56); with the monitor armed the objective is appended."""
    def show(k, st):
        line = "%5d | %.3e  " % (k, getattr(st, fmt_attr))
        it = st._it
        if it is not None and it.monitor and not isinstance(st, Proshi_basic_state):
            line += "| F = %.9e" % st.objective
        print(line)
    return show


SVRG_basic_iterable._chunkable = False
SAGA_basic_iterable._chunkable = True
FINITO_basic_iterable._chunkable = True
FINITO_LFinito_iterable._chunkable = False
FINITO_adaptive_iterable._chunkable = True


class SVRG(_Solver):
    """SVRG{R}(; γ, maxit=10000, verbose=false, freq=1000, m=nothing, plus=false)   (SVRG.jl:24-44, :112-113)"""

    def __init__(self, R=np.float64, *, γ=None, gamma=None, maxit=10000, verbose=False, freq=1000, m=None, plus=False):
        γ = _pick(γ, gamma, "γ")
        assert γ is None or γ > 0
        assert maxit > 0
        assert freq > 0
        self.R, self.γ, self.maxit, self.verbose, self.freq, self.m, self.plus = R, γ, int(maxit), verbose, int(freq), m, plus

    def _iterable(self, x0, F=None, g=None, L=None, μ=None, mu=None, N=None, ctx=None, stream=None, shards=None,
                  fallback=None, backend=None):
        μ = _pick(μ, mu, "μ")
        m = self.m if self.m is not None else N                            # SVRG.jl:59
        return _route(lambda: SVRG_basic_iterable(self.R, F, g, x0, N, L, μ, self.γ, m, self.plus, ctx=ctx, stream=stream, shards=shards),
                      lambda: HR.HostSVRG(self.R, F, g, x0, N, L, μ, self.γ, m, self.plus, stream=stream), fallback, backend, F, g, x0)

    def __call__(self, x0, **kw):                                          # SVRG.jl:46-84
        maxit = self.maxit
        if self.plus and self.maxit > 25:                                  # :61-65
            maxit = 25
            warnings.warn("exponential number of inner updates...reverted to 25 maximum iterations")
        stop, every = _split_drive_kw(kw)
        return self._drive(self._iterable(x0, **kw), maxit, _disp("γ"), stop, every)


class SAGA(_Solver):
    """SAGA{R}(; γ, maxit=10000, verbose=false, freq=1000, SAG_flag=false)   (SAGA.jl:24-42, :108-109)"""

    def __init__(self, R=np.float64, *, γ=None, gamma=None, maxit=10000, verbose=False, freq=1000, SAG_flag=False):
        γ = _pick(γ, gamma, "γ")
        assert γ is None or γ > 0
        assert maxit > 0
        assert freq > 0
        self.R, self.γ, self.maxit, self.verbose, self.freq, self.SAG_flag = R, γ, int(maxit), verbose, int(freq), SAG_flag

    def _iterable(self, x0, F=None, g=None, L=None, N=None, ctx=None, stream=None, shards=None, fallback=None, backend=None):
        return _route(lambda: SAGA_basic_iterable(self.R, F, g, x0, N, L, self.γ, self.SAG_flag, ctx=ctx, stream=stream, shards=shards),
                      lambda: HR.HostSAGA(self.R, F, g, x0, N, L, self.γ, self.SAG_flag, stream=stream), fallback, backend, F, g, x0)

    def __call__(self, x0, **kw):                                          # SAGA.jl:44-73
        stop, every = _split_drive_kw(kw)
        return self._drive(self._iterable(x0, **kw), self.maxit, _disp("γ"), stop, every)


def SAG(R=np.float64, **kw):
    """SAG(R; kwargs...) = SAGA{R}(; kwargs..., SAG_flag=true)   (SAGA.jl:190-191)"""
    return SAGA(R, SAG_flag=True, **kw)


class Finito(_Solver):
    """Finito{R}(; γ, sweeping=1, LFinito=false, adaptive=false, minibatch=(false,1), maxit=10000, verbose=false,
    freq=10000, α=0.999, tol=1e-8, tol_b=1e-9)   (Finito.jl:32-64, :165-166)"""

    def __init__(self, R=np.float64, *, γ=None, gamma=None, sweeping=1, LFinito=False, adaptive=False, minibatch=(False, 1),
                 maxit=10000, verbose=False, freq=10000, α=None, alpha=None, tol=1e-8, tol_b=1e-9):
        γ = _pick(γ, gamma, "γ")
        α = _pick(α, alpha, "α")
        α = 0.999 if α is None else α
        assert γ is None or np.min(np.asarray(γ.cpu() if isinstance(γ, torch.Tensor) else γ)) > 0
        assert maxit > 0
        assert tol > 0
        assert tol_b > 0
        assert freq > 0
        self.R, self.γ, self.sweeping, self.LFinito, self.adaptive = R, γ, sweeping, LFinito, adaptive
        self.minibatch, self.maxit, self.verbose, self.freq, self.α, self.tol, self.tol_b = (
            tuple(minibatch), int(maxit), verbose, int(freq), α, tol, tol_b)

    def _iterable(self, x0, F=None, g=None, L=None, N=None, ctx=None, stream=None, fallback=None, backend=None, shards=None):   # Finito.jl:80-116
        if shards is not None and not self.adaptive:
            raise ValueError("shards= is for the sequential chains (adaptive Finito here); Finito / LFinito batches shard by rows alone")
        if self.LFinito:
            return _route(lambda: FINITO_LFinito_iterable(self.R, F, g, x0, N, L, self.γ, self.sweeping, self.minibatch[1], self.α,
                                                          ctx=ctx, stream=stream),
                          lambda: HR.HostLFinito(self.R, F, g, x0, N, L, self.γ, self.sweeping, self.minibatch[1], self.α, stream=stream),
                          fallback, backend, F, g, x0)
        if self.adaptive:
            if backend == "host":
                raise NotImplementedError("the host route has no adaptive Finito")
            return FINITO_adaptive_iterable(self.R, F, g, x0, N, L, self.tol, self.tol_b, self.sweeping, self.α,
                                            ctx=ctx, stream=stream, shards=shards)
        return _route(lambda: FINITO_basic_iterable(self.R, F, g, x0, N, L, self.γ, self.sweeping, self.minibatch[1], self.α,
                                                    ctx=ctx, stream=stream),
                      lambda: HR.HostFinito(self.R, F, g, x0, N, L, self.γ, self.sweeping, self.minibatch[1], self.α, stream=stream),
                      fallback, backend, F, g, x0)

    def __call__(self, x0, **kw):                                          # Finito.jl:66-133
        stop, every = _split_drive_kw(kw)
        return self._drive(self._iterable(x0, **kw), self.maxit, _disp("hat_γ"), stop, every)


class Proshi(_Solver):
    """Proshi{R}(; γ, sweeping=1, minibatch=(false,1), maxit=10000, verbose=false, freq=10000, α=0.999)   (ProShI.jl:18-40)"""

    def __init__(self, R=np.float64, *, γ=None, gamma=None, sweeping=1, minibatch=(False, 1), maxit=10000, verbose=False,
                 freq=10000, α=None, alpha=None):
        γ = _pick(γ, gamma, "γ")
        α = _pick(α, alpha, "α")
        α = 0.999 if α is None else α
        assert γ is None or np.min(np.asarray(γ.cpu() if isinstance(γ, torch.Tensor) else γ)) > 0
        assert maxit > 0
        assert freq > 0
        self.R, self.γ, self.sweeping, self.minibatch = R, γ, sweeping, tuple(minibatch)
        self.maxit, self.verbose, self.freq, self.α = int(maxit), verbose, int(freq), α

    def _iterable(self, x0, F=None, g=None, L=None, N=None, ctx=None, stream=None):
        return Proshi_basic_iterable(self.R, F, g, x0, N, L, self.γ, self.sweeping, self.minibatch[1], self.α, ctx=ctx,
                                     stream=stream)

    def __call__(self, x0, **kw):                                          # ProShI.jl:42-83
        stop, every = _split_drive_kw(kw)
        return self._drive(self._iterable(x0, **kw), self.maxit, _disp("hat_γ"), stop, every)


Proshi_basic_iterable._chunkable = True


def solve_together(iterables, maxit, one_pass=False):
    """K independent SVRG, SAGA / SAG or small-batch Finito solves over device-resident rows, advanced in LOCKSTEP: `iterables` are what
    `iterator(solver, x0, F=..., g=..., N=..., ctx=ctx)` returns, all on the same ctx (typically the same packed F with a g -- a
    lambda of the regularisation path -- and a sampling stream each).  The reference solves one problem per call and a chain is
    one workgroup on one of the GPU's 256 compute units; here the K sequential chains of every outer step are recorded and
    launched as ONE chain batch (Context.chain_batch; include/ciao_hip.h: ciao_ctx_chain_batch_begin), one workgroup per solve.
    Returns ([solution_k], num_iters) with the functors' counting (the init state is iteration 1, SVRG.jl:69-83).

    Each solve ends bitwise as its own `solver(maxit=maxit)(x0, ...)` would -- for SVRG with the row-dot cache off
    (`ctx.set_option("svrg_cache_rowdots", 0)`: a batch's inner cycles recompute a_i'z_full, include/ciao_hip.h:
    ciao_svrg_epoch_tail), for SAGA / SAG and Finito as is (Finito: batches small enough to run as a chain, i.e. the default
    minibatch of one sample up to option chain_max_batch; larger batches are batch-parallel kernels that fill the GPU alone).

    `one_pass=True` (SVRG over ONE packed F): the K full passes of every outer step (SVRG_basic.jl:87-92 per solve) run as ONE pass
    over the rows with K right-hand sides on the matrix cores (ciao_svrg_epoch_tail_multi; d = 256 / 512 / 768 / 1024, fp64 also 128) instead of K
    sweeps -- at N = 10M, K = 256 that is most of what an outer step costs beside the batched inner cycles.  The pass sums in another
    order, so each solve then follows its own functor call to ROUNDING, not bitwise."""
    its = list(iterables)
    if not its:
        return [], 0
    kinds = {type(it) for it in its}
    if len(kinds) != 1 or not kinds <= {SVRG_basic_iterable, SAGA_basic_iterable, FINITO_basic_iterable}:
        raise TypeError("solve_together takes SVRG iterables, SAGA / SAG iterables or basic Finito iterables (one kind), built by "
                        "iterator(solver, x0, ...)")
    ctx = its[0].ctx
    if any(it.ctx is not ctx for it in its) or any(getattr(it, "shards", None) is not None for it in its):
        raise ValueError("the solves of a batch share one ctx and are unsharded")
    states = []
    for it in its:
        iter(it)
        it._started = True
        it._state = it._init()
        if it._state is None:
            raise ValueError("an iterable of the batch has an invalid configuration (see the warning above)")
        states.append(it._state)
    num_iters = 1
    svrg = kinds == {SVRG_basic_iterable}
    chunk = _Solver._chunk
    while num_iters < maxit:
        if svrg:                                                           # SVRG_basic.jl:71-96, once per solve
            drawn = [it._draw(it.N, st.m)[0] for it, st in zip(its, states)]                       # :73
            with ctx.chain_batch():
                for it, st, idx in zip(its, states, drawn):
                    ctx.svrg_inner(it.F, it.g, st.γ, idx, st.av, st.z, st.z_full, st.w)            # :74-82
            same = (one_pass and all(it.F is its[0].F for it in its) and len({st.m for st in states}) == 1
                    and len({bool(it.plus) for it in its}) == 1)
            if same:                                                                               # :84-92 of all K solves, one pass over A
                ctx.svrg_epoch_tail_multi(its[0].F, states[0].m, its[0].plus, [st.av for st in states], [st.z for st in states],
                                          [st.z_full for st in states], [st.w for st in states])
            for it, st in zip(its, states):
                if not same:
                    ctx.svrg_epoch_tail(it.F, st.m, it.plus, st.av, st.z, st.z_full, st.w)         # :84-92
                st._tok = None
                if it.plus:
                    st.m *= 2                                                                      # :93
            num_iters += 1
        elif kinds == {FINITO_basic_iterable}:                             # Finito_basic.jl:91-121, n iterations per solve and launch
            n = min(maxit - num_iters, max(1, min(chunk, _Solver._chunk_samples // max(it.batch for it in its))))
            drawn = [_next_batches_packed(it, st, n) for it, st in zip(its, states)]               # :95-108 (index lists, whatever the sweeping)
            with ctx.chain_batch():
                for it, st, (bptr, bidx) in zip(its, states, drawn):
                    ctx.finito_steps(it.F, it.g, st.γ, st.hat_γ, bptr, bidx, st.s, st.av, st.z)    # :109-118
            num_iters += n
        else:                                                              # SAGA_basic.jl:53-68, n iterations per solve and launch
            n = min(maxit - num_iters, chunk)
            drawn = [it._draw(it.N, n) for it in its]                                              # :55
            with ctx.chain_batch():
                for it, st, (idx, _) in zip(its, states, drawn):
                    ctx.saga_steps(it.F, it.g, st.γ, it.SAG, idx, st.s, st.av, st.z)
            for st, (_, last) in zip(states, drawn):
                st.ind = last + 1
            num_iters += n
    ctx.synchronize()
    outs = []
    for it, st in zip(its, states):
        sol = solution(st)
        outs.append(sol.cpu().numpy().reshape(np.shape(it.x0)) if it._numpy else sol)
    return outs, num_iters


def iterator(solver, x0, **kw):
    """iterator(solver, x0; F, g, L, [μ], N): the raw iterable; maxit/verbose/freq of the solver are ignored
    (SVRG.jl:132-147, SAGA.jl:128-142, Finito.jl:186-234)."""
    return solver._iterable(x0, **kw)
