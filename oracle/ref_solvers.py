"""CPU restatement of the reference's iterables + solver functors (L2/L3), driving the C oracle kernels.

TEST INFRASTRUCTURE ONLY (see oracle/ciao_oracle.h).  Used by tests/ to pin the oracle against the reference's
known-answer tests and as the per-step checker for the HIP path.

What is restated (citations relative to /root/reference/):
  * SVRG_basic_iterable   src/algorithms/SVRG/SVRG_basic.jl:30-99   + driver SVRG.jl:46-84
  * SAGA_basic_iterable   src/algorithms/SAGA_SAG/SAGA_basic.jl:26-71 + driver SAGA.jl:44-73
  * FINITO_basic_iterable src/algorithms/Finito/Finito_basic.jl:44-123
  * FINITO_LFinito_iterable src/algorithms/Finito/Finito_LFinito.jl:40-105 + dispatcher Finito.jl:66-133

Julia's global RNG stream is not reproducible outside Julia, so every sampling decision is drawn from an injected
`stream` object with four methods (the four random calls the reference makes):
    rand_indices(N, m)                 <->  rand(state.ind, m)            SVRG_basic.jl:73
    rand_indices(N, 1)[0]              <->  rand(1:N)                     SAGA_basic.jl:55
    sample_without_replacement(N, r)   <->  sample(1:N, r, replace=false) Finito_basic.jl:97
    randperm(n)                        <->  randperm(n)                   Finito_basic.jl:102, Finito_LFinito.jl:89
All indices are 0-based here.  Each iterable is a Python generator-style object: iter() yields the init state
first (SVRG_basic.jl:68), then one state per Base.iterate(iter, state).
"""
from __future__ import annotations

import warnings

import numpy as np

from . import oracle as O


def _static_batches(N, r):
    """Finito_basic.jl:52-58 / Finito_LFinito.jl:44-49: contiguous blocks of r, plus the remainder."""
    ind = []
    dfl = N // r
    for i in range(dfl):
        ind.append(np.arange(r * i, r * (i + 1), dtype=np.int64))
    if r * dfl < N:
        ind.append(np.arange(r * dfl, N, dtype=np.int64))
    return ind


def _finito_gammas(dtype, N, L, gamma, alpha):
    """Finito_basic.jl:61-74 (identical in Finito_LFinito.jl:51-63)."""
    R = np.dtype(dtype).type
    if gamma is None:
        if L is None:
            warnings.warn("--> smoothness parameter absent")
            return None
        if np.ndim(L) == 0:
            return np.full(N, R(alpha) * R(N) / R(L), dtype)
        L = np.asarray(L, dtype)
        return (R(alpha) * R(N) / L).astype(dtype)
    if np.ndim(gamma) == 0:
        return np.full(N, R(gamma), dtype)
    return np.ascontiguousarray(gamma, dtype)


class SVRGState:
    def __init__(self, gamma, m, av, z, z_full, w):
        self.gamma, self.m, self.av, self.z, self.z_full, self.w = gamma, m, av, z, z_full, w


class SVRGIterable:
    def __init__(self, problem, g, x0, L=None, mu=None, gamma=None, m=None, plus=False, stream=None):
        self.p, self.g, self.x0, self.L, self.mu = problem, g, x0, L, mu
        self.gamma, self.m, self.plus, self.stream = gamma, m, plus, stream

    def __iter__(self):
        p = self.p
        R = p.dtype.type
        N = p.N
        m = N if self.m is None else self.m
        if self.gamma is None:                                            # SVRG_basic.jl:35-56
            if self.plus:
                warnings.warn("provide a stepsize γ")
                return
            if self.L is None or self.mu is None:
                warnings.warn("smoothness or convexity parameter absent")
                return
            L_M, mu_M = float(np.max(self.L)), float(np.max(self.mu))
            gamma = 1 / (10 * L_M)
            rho = (1 + 4 * L_M * gamma ** 2 * mu_M * (N + 1)) / (mu_M * gamma * N * (1 - 4 * L_M * gamma))
            if rho >= 1:
                warnings.warn("convergence condition violated...provide a stepsize!")
        else:
            gamma = self.gamma
        gamma = R(gamma)
        av, z, z_full, w = O.svrg_init(p, self.x0)                        # :57-66
        st = SVRGState(gamma, m, av, z, z_full, w)
        yield st                                                          # :68 init state is item #1
        while True:
            idx = self.stream.rand_indices(N, st.m)                       # :73
            O.svrg_iterate(p, self.g, gamma, idx, self.plus, st.av, st.z, st.z_full, st.w)   # :73-92
            if self.plus:
                st.m *= 2                                                 # :93
            yield st


class SAGAState:
    def __init__(self, s, gamma, av, z):
        self.s, self.gamma, self.av, self.z, self.ind = s, gamma, av, z, 0


class SAGAIterable:
    def __init__(self, problem, g, x0, L=None, gamma=None, sag=False, stream=None):
        self.p, self.g, self.x0, self.L, self.gamma, self.sag, self.stream = problem, g, x0, L, gamma, sag, stream

    def __iter__(self):
        p = self.p
        R = p.dtype.type
        if self.gamma is None:                                            # SAGA_basic.jl:29-39
            if self.L is None:
                warnings.warn("smoothness parameter absent")
                return
            L_M = float(np.max(self.L))
            gamma = 1 / (16 * L_M) if self.sag else 1 / (3 * L_M)
        else:
            gamma = self.gamma
        gamma = R(gamma)
        s, av, z = O.saga_init(p, self.g, gamma, self.x0)                 # :41-48
        st = SAGAState(s, gamma, av, z)
        yield st
        while True:
            st.ind = int(self.stream.rand_indices(p.N, 1)[0])             # :55
            O.saga_steps(p, self.g, gamma, self.sag, [st.ind], st.s, st.av, st.z)   # :56-65
            yield st


class FinitoState:
    def __init__(self, s, gam, hat_gamma, av, z, ind, d):
        self.s, self.gamma, self.hat_gamma, self.av, self.z, self.ind, self.d = s, gam, hat_gamma, av, z, ind, d
        self.idxr, self.idx, self.inds = 0, 0, np.arange(d, dtype=np.int64)   # Finito_basic.jl:37-40 (0-based idxr)


class FinitoIterable:
    def __init__(self, problem, g, x0, L=None, gamma=None, sweeping=1, batch=1, alpha=0.999, stream=None):
        self.p, self.g, self.x0, self.L, self.gamma = problem, g, x0, L, gamma
        self.sweeping, self.batch, self.alpha, self.stream = sweeping, batch, alpha, stream

    def __iter__(self):
        p = self.p
        N, r = p.N, self.batch
        if self.sweeping == 1:
            ind = [np.arange(r, dtype=np.int64)]                          # placeholder, Finito_basic.jl:50
        else:
            ind = _static_batches(N, r)                                   # :52-58
        d = -(-N // r)                                                    # cld(N, r) :59
        gam = _finito_gammas(p.dtype, N, self.L, self.gamma, self.alpha)  # :61-74
        if gam is None:
            return
        s, av, z, hg = O.finito_init(p, self.g, gam, self.x0)             # :76-84
        st = FinitoState(s, gam, hg, av, z, ind, d)
        yield st
        while True:
            if self.sweeping == 1:                                        # :96-97
                st.ind = [self.stream.sample_without_replacement(N, r)]
            elif self.sweeping == 2:                                      # :98-99  idxr = mod(idxr, d) + 1
                st.idxr = (st.idxr + 1) % st.d                            #   (1-based 1 -> 2 first; 0-based 0 -> 1)
            elif self.sweeping == 3:                                      # :100-108
                if st.idx == st.d:
                    st.inds = self.stream.randperm(st.d)
                    st.idx = 1
                else:
                    st.idx += 1
                st.idxr = int(st.inds[st.idx - 1])
            O.finito_steps(p, self.g, st.gamma, st.hat_gamma, [st.ind[st.idxr]], st.s, st.av, st.z)   # :110-118
            yield st


class LFinitoState:
    def __init__(self, gam, hat_gamma, av, ind, d, z, z_full):
        self.gamma, self.hat_gamma, self.av, self.ind, self.d, self.z, self.z_full = gam, hat_gamma, av, ind, d, z, z_full
        self.inds = np.arange(d, dtype=np.int64)


class LFinitoIterable:
    def __init__(self, problem, g, x0, L=None, gamma=None, sweeping=1, batch=1, alpha=0.999, stream=None):
        self.p, self.g, self.x0, self.L, self.gamma = problem, g, x0, L, gamma
        self.sweeping, self.batch, self.alpha, self.stream = sweeping, batch, alpha, stream

    def __iter__(self):
        p = self.p
        N, r = p.N, self.batch
        ind = _static_batches(N, r)                                       # Finito_LFinito.jl:44-49
        gam = _finito_gammas(p.dtype, N, self.L, self.gamma, self.alpha)  # :51-63
        if gam is None:
            return
        av, z, z_full, hg = O.lfinito_init(p, gam, self.x0)               # :66-72
        st = LFinitoState(gam, hg, av, ind, -(-N // r), z, z_full)
        yield st
        while True:
            if self.sweeping == 3:                                        # :89 (drawn after the full pass; the
                st.inds = self.stream.randperm(st.d)                      #      order of RNG calls is unchanged)
            batches = [st.ind[int(j)] for j in st.inds]
            O.lfinito_iterate(p, self.g, st.gamma, st.hat_gamma, batches, st.av, st.z, st.z_full)   # :82-100
            yield st


class AdaptiveFinitoState:
    def __init__(self, s, gf, gam, hat_gamma, fi_x, av, z, N):
        self.s, self.grad_f, self.gamma, self.hat_gamma, self.fi_x, self.av, self.z = s, gf, gam, hat_gamma, fi_x, av, z
        self.ind = np.arange(N, dtype=np.int64)   # Finito_adaptive.jl:52 (copy(indr)): identity order at first
        self.idx, self.idxr = 0, 0                # :53-54 (idxr is 1-based in the reference; 0 = "none yet")
        self.trials = 0


class AdaptiveFinitoIterable:
    """FINITO_adaptive_iterable (Finito_adaptive.jl).  minibatch is not supported by the reference (:162)."""

    def __init__(self, problem, g, x0, L=None, tol=1e-8, tol_b=1e-9, sweeping=1, alpha=0.999, stream=None):
        self.p, self.g, self.x0, self.L = problem, g, x0, L
        self.tol, self.tol_b, self.sweeping, self.alpha, self.stream = tol, tol_b, sweeping, alpha, stream

    def __iter__(self):
        p = self.p
        N = p.N
        R = p.dtype.type
        s, gf, gam, fi_x, av, z, hg = O.afinito_init(p, self.g, R(self.alpha), self.x0)     # :59-98
        st = AdaptiveFinitoState(s, gf, gam, hg, fi_x, av, z, N)
        yield st
        while True:
            if self.sweeping == 1:                                        # :105-106
                st.idxr = int(self.stream.rand_indices(N, 1)[0]) + 1
            elif self.sweeping == 2:                                      # :107-108
                st.idxr = st.idxr % N + 1
            elif self.sweeping == 3:                                      # :109-116
                if st.idx == N:
                    st.ind = self.stream.randperm(N)
                    st.idx = 1
                else:
                    st.idx += 1
                st.idxr = int(st.ind[st.idx - 1]) + 1
            done, st.hat_gamma, tr = O.afinito_steps(p, self.g, R(self.alpha), R(self.tol_b), [st.idxr - 1], st.s, st.grad_f,
                                                     st.gamma, st.fi_x, st.hat_gamma, st.av, st.z)
            st.trials += tr
            if done < 1:
                warnings.warn("parameter `γ` became too small")          # :122-123: return nothing
                return
            yield st


class ProshiState:
    def __init__(self, s, gam, hat_gamma, av, z, ind, d):
        self.s, self.gamma, self.hat_gamma, self.av, self.z, self.ind, self.d = s, gam, hat_gamma, av, z, ind, d
        self.idxr, self.idx, self.inds = 0, 0, np.arange(d, dtype=np.int64)   # ProShI_basic.jl:37-39 (0-based idxr)


class ProshiIterable:
    """Proshi_basic_iterable (src/algorithms/ProShI/ProShI_basic.jl); batch logic identical to Finito's (:47-59, :95-107)."""

    def __init__(self, f, g, x0, L=None, gamma=None, sweeping=1, batch=1, alpha=0.999, stream=None):
        self.f, self.g, self.x0, self.L, self.gamma = f, g, x0, L, gamma
        self.sweeping, self.batch, self.alpha, self.stream = sweeping, batch, alpha, stream

    def __iter__(self):
        f = self.f
        N, r = f.N, self.batch
        ind = [np.arange(r, dtype=np.int64)] if self.sweeping == 1 else _static_batches(N, r)
        d = -(-N // r)
        gam = _finito_gammas(f.dtype, N, self.L, self.gamma, self.alpha)      # :61-74 (same rule as Finito)
        if gam is None:
            return
        s, av, z, hg = O.proshi_init(f, self.g, gam, self.x0)                 # :76-87
        st = ProshiState(s, gam, hg, av, z, ind, d)
        yield st
        while True:
            if self.sweeping == 1:
                st.ind = [self.stream.sample_without_replacement(N, r)]
            elif self.sweeping == 2:
                st.idxr = (st.idxr + 1) % st.d
            elif self.sweeping == 3:
                if st.idx == st.d:
                    st.inds = self.stream.randperm(st.d)
                    st.idx = 1
                else:
                    st.idx += 1
                st.idxr = int(st.inds[st.idx - 1])
            O.proshi_steps(f, self.g, st.gamma, st.hat_gamma, [st.ind[st.idxr]], st.s, st.av, st.z)   # :109-121
            yield st


def proshi_solution(iterable, state):
    """solution(state::Proshi_basic_state), ProShI_basic.jl:127-132: shifts the table IN PLACE and returns it."""
    return O.proshi_solution(iterable.f, state.gamma, state.z, state.s)


def proshi(f, g, x0, maxit=10000, gamma=None, sweeping=1, batch=1, alpha=0.999, L=None, stream=None):
    """Proshi functor (ProShI.jl:42-83): returns (solution(state_final), num_iters)."""
    itb = ProshiIterable(f, g, x0, L, gamma, sweeping, batch, alpha, stream)
    it, last = 0, None
    for st in itb:
        it, last = it + 1, st
        if it >= maxit:
            break
    if last is None:
        raise TypeError("solution(nothing)")
    return proshi_solution(itb, last), it


def solution(state):
    """SVRG_basic.jl:99, SAGA_basic.jl:71, Finito_basic.jl:123, Finito_LFinito.jl:105."""
    return state.z_full if isinstance(state, SVRGState) else state.z


def run(iterable, maxit):
    """The solver functor's loop (SVRG.jl:69-83): take(iter, maxit), return (solution(last), count)."""
    it, last = 0, None
    for st in iterable:
        it, last = it + 1, st
        if it >= maxit:
            break
    if last is None:
        raise TypeError("solution(nothing): the iterable ended before yielding a state (invalid configuration)")
    return solution(last), it


def svrg(problem, g, x0, maxit=10000, gamma=None, m=None, plus=False, L=None, mu=None, stream=None):
    if plus and maxit > 25:                                               # SVRG.jl:61-65
        maxit = 25
        warnings.warn("exponential number of inner updates...reverted to 25 maximum iterations")
    return run(SVRGIterable(problem, g, x0, L, mu, gamma, m, plus, stream), maxit)


def saga(problem, g, x0, maxit=10000, gamma=None, sag=False, L=None, stream=None):
    return run(SAGAIterable(problem, g, x0, L, gamma, sag, stream), maxit)


def finito(problem, g, x0, maxit=10000, gamma=None, sweeping=1, lfinito=False, batch=1, alpha=0.999, L=None,
           stream=None, adaptive=False, tol=1e-8, tol_b=1e-9):
    if lfinito:                                                           # Finito.jl:80-116 (LFinito wins over adaptive)
        it = LFinitoIterable(problem, g, x0, L, gamma, sweeping, batch, alpha, stream)
    elif adaptive:
        it = AdaptiveFinitoIterable(problem, g, x0, L, tol, tol_b, sweeping, alpha, stream)
    else:
        it = FinitoIterable(problem, g, x0, L, gamma, sweeping, batch, alpha, stream)
    return run(it, maxit)
