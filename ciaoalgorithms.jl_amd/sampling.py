"""Injected index streams for the finite-sum solvers.

The reference draws every sample from Julia's *global* RNG (`rand(state.ind, m)` SVRG_basic.jl:73, `rand(1:N)`
SAGA_basic.jl:55, `sample(1:N, r, replace=false)` Finito_basic.jl:97, `randperm(d)` Finito_basic.jl:102 and
Finito_LFinito.jl:89).  That stream depends on the Julia version and cannot be reproduced outside Julia, so here the
sampling decisions are an explicit *input*: a counter-based splitmix64 stream whose k-th output depends only on
(seed, k).  Drawing m indices in one call or one at a time gives the same sequence, which is what lets the device path
consume indices in large chunks while the CPU oracle consumes them one by one.

All indices are 0-based int64 (the C ABI is 0-based; the Julia wrapper subtracts 1).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(seed: int, start: int, n: int) -> np.ndarray:
    """Outputs start .. start+n-1 of the splitmix64 sequence seeded with `seed` (vectorised, wrap-around uint64)."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class IndexStream:
    """Deterministic, chunking-independent source of the four kinds of random draws the reference makes."""

    def __init__(self, seed: int = 0):
        self.seed = int(seed)
        self.pos = 0  # number of 64-bit outputs consumed so far

    def _take(self, n: int) -> np.ndarray:
        out = _splitmix64(self.seed, self.pos, n)
        self.pos += n
        return out

    def rand_indices(self, N: int, m: int) -> np.ndarray:
        """m i.i.d. uniform draws from 0..N-1 (with replacement); N < 2**32."""
        assert 0 < N < (1 << 32)
        u = self._take(m) >> np.uint64(32)
        return ((u * np.uint64(N)) >> np.uint64(32)).astype(np.int64)

    def rand_indices_device(self, ctx, N: int, m: int):
        """The same m draws, generated on the device by `ctx` (device.Context.sample_uniform): returns (device int64 tensor,
        last index as a host int or None when m == 0).  The stream position advances exactly as rand_indices does, so host
        and device draws can be mixed freely."""
        assert 0 < N < (1 << 32)
        t = ctx.sample_uniform(self.seed, self.pos, N, m)
        last = None
        if m > 0:
            u = _splitmix64(self.seed, self.pos + m - 1, 1) >> np.uint64(32)
            last = int(((u * np.uint64(N)) >> np.uint64(32))[0])
        self.pos += m
        return t, last

    def rand_signs(self, n: int) -> np.ndarray:
        """n i.i.d. uniform draws from {-1, +1} as int8 (the reference's rand([-1, 1], size(x0)), Finito_adaptive.jl:80)."""
        return ((self._take(n) >> np.uint64(63)).astype(np.int8) * 2 - 1).astype(np.int8)

    def randperm(self, n: int) -> np.ndarray:
        """Uniform random permutation of 0..n-1 (argsort of n stream outputs; ties have probability ~n^2/2^64)."""
        return np.argsort(self._take(n), kind="stable").astype(np.int64)

    def sample_without_replacement(self, N: int, r: int) -> np.ndarray:
        """r distinct uniform indices from 0..N-1, in draw order (rejection of repeats)."""
        assert 0 < r <= N
        if 2 * r > N:
            return self.randperm(N)[:r].copy()
        out = np.empty(0, np.int64)
        while out.size < r:
            cand = self.rand_indices(N, max(r - out.size, 1))
            out = np.concatenate([out, cand])
            _, first = np.unique(out, return_index=True)
            out = out[np.sort(first)]
        return np.ascontiguousarray(out[:r])


    def sample_batches(self, N: int, r: int, n: int) -> np.ndarray:
        """n consecutive sample_without_replacement(N, r) as an (n, r) array -- the same values and the same stream
        position as n separate calls, without n trips through the interpreter (a device batch takes 10-50 us, the loop
        above 25-850 us).  r = 1 is a plain uniform draw; the general case runs in libciao_hip.so's host helper
        ciao_sample_batches, which restates the rule above with a hash set.  The returned array may be a view of a
        scratch buffer owned by the stream: consume or copy it before the next call."""
        assert 0 < r <= N and n >= 0
        if n == 0:
            return np.empty((0, r), np.int64)
        if r == 1:
            return self.rand_indices(N, n).reshape(n, 1)
        if 2 * r <= N and N < (1 << 32):
            try:
                from . import _lib as L
                lib = L.load()
            except Exception:   # the library is not built (CPU-only checkout): the interpreted rule, same result
                lib = None
            if lib is not None:
                import ctypes as C
                # the result lives in a scratch buffer that is reused (fresh pages cost more than the draws themselves):
                # it is valid until the next call on this stream
                if getattr(self, "_scratch", None) is None or self._scratch.size < n * r:
                    self._scratch = np.empty(max(n * r, 1 << 16), np.int64)
                out = self._scratch[:n * r].reshape(n, r)
                pos = C.c_uint64(0)
                L.check(lib.ciao_sample_batches(C.c_uint64(self.seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(self.pos), N, r, n,
                                                out.ctypes.data_as(C.c_void_p), C.byref(pos)))
                self.pos = int(pos.value)
                return out
        return np.stack([self.sample_without_replacement(N, r) for _ in range(n)])


class FixedStream:
    """Replays explicit index arrays (for golden-vector tests): rand_indices pops from `indices`, randperm and
    sample_without_replacement pop whole arrays from their queues."""

    def __init__(self, indices=(), perms=(), samples=()):
        self._idx = np.asarray(indices, np.int64)
        self._ipos = 0
        self._perms = [np.asarray(p, np.int64) for p in perms]
        self._samples = [np.asarray(s, np.int64) for s in samples]

    def rand_indices(self, N, m):
        out = self._idx[self._ipos:self._ipos + m]
        if out.size != m:
            raise IndexError("FixedStream: index stream exhausted")
        self._ipos += m
        assert out.size == 0 or (out.min() >= 0 and out.max() < N)
        return out.copy()

    def rand_signs(self, n):
        raise IndexError("FixedStream holds no sign draws")

    def randperm(self, n):
        p = self._perms.pop(0)
        assert p.size == n
        return p

    def sample_without_replacement(self, N, r):
        s = self._samples.pop(0)
        assert s.size == r
        return s
