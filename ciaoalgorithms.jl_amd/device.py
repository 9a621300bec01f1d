"""Device-side plumbing: a `Context` (ciao_ctx handle bound to a torch HIP stream) and typed wrappers that pass torch
device tensors to the C ABI.  PyTorch is used here only for device memory, streams and (in parallel.py)
torch.distributed -- all arithmetic happens in libciao_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.float64: L.F64}
_NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def torch_dtype(R) -> torch.dtype:
    """Map the reference's real type parameter `R` (numpy/torch dtype or python float) to a torch dtype."""
    if isinstance(R, torch.dtype):
        if R not in _DT:
            raise TypeError(f"only float32/float64 are supported on the device path, got {R}")
        return R
    if R is float:
        return torch.float64
    return _NP2T[np.dtype(R)]


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class PackedF:
    """F = [f_1 .. f_N] packed for the device: row-major A (N x d), b (targets or labels), loss kind, LeastSquares λ.

    Replaces the reference's Vector of N one-row operator objects (test/test_lasso.jl:50-58).  `N_total` is the global
    sample count when A holds only this rank's row shard (parallel.py); `row0` is the global index of local row 0.
    """

    def __init__(self, loss: int, A: torch.Tensor | None, b: torch.Tensor | None, lam: float = 1.0, N_total: int | None = None,
                 row0: int = 0, d: int | None = None, dtype: torch.dtype | None = None, N: int | None = None,
                 cyclic: tuple[int, int] | None = None):
        self.loss = int(loss)
        # row ownership of a shard: contiguous block [row0, row0+N) (default), or cyclic=(rank, world): this rank owns the
        # global rows i with i % world == rank, stored in order (local row = i // world).  Cyclic ownership spreads every
        # static contiguous Finito batch (Finito_basic.jl:52-58) evenly over the ranks.
        self.cyclic = cyclic
        if A is not None:
            assert A.is_cuda and A.dim() == 2 and A.stride(1) == 1, "A must be a row-major device matrix"
            assert A.dtype in _DT, f"unsupported dtype {A.dtype}"
            self.N, self.d = int(A.shape[0]), int(A.shape[1])
            self.ld = int(A.stride(0)) if self.N > 1 else self.d
            self.dtype = A.dtype
            self.device = A.device
        else:  # F = fill(Zero(), N): no data at all (SVRG.jl:58)
            assert loss == L.LOSS_ZERO and d is not None and dtype is not None and N is not None
            self.N, self.d, self.ld, self.dtype = int(N), int(d), int(d), dtype
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.complex = (self.loss == L.LOSS_LS_COMPLEX)   # vectors are (re, im) pairs of the real dtype: d counts reals
        if self.complex:
            assert self.d % 2 == 0 and self.ld % 2 == 0, "complex rows are (re, im) pairs: an even number of reals"
        if b is not None:
            assert b.is_cuda and b.dtype == self.dtype and b.is_contiguous() and b.shape == ((2 if self.complex else 1) * self.N,)
        elif loss != L.LOSS_ZERO:
            raise ValueError("b (targets / labels) is required for LeastSquares and logistic F")
        self.A, self.b, self.lam = A, b, float(lam)
        # Feature padding (operators.pack_F(pad_to=...)): the rows carry self.d - padded_from zero columns, and the solver's state
        # vectors are length-padded_from VIEWS of length-self.d buffers.  None: every vector must have exactly self.d elements.
        self.padded_from: int | None = None
        self.N_total = int(N_total) if N_total is not None else self.N
        self.row0 = int(row0)
        self._c = L.Problem(self.loss, _DT[self.dtype], self.N, self.d, max(self.ld, self.d), self.N_total,
                            A.data_ptr() if (A is not None and self.N > 0) else None,
                            b.data_ptr() if (b is not None and self.N > 0) else None, self.lam)

    @property
    def ref(self):
        return C.byref(self._c)

    def localise(self, idx):
        """Global sample indices -> the local row indices of the members this shard owns (order preserved)."""
        import numpy as _np
        idx = _np.asarray(idx, dtype=_np.int64)
        if self.N == self.N_total and self.cyclic is None:
            return idx
        if self.cyclic is not None:
            rank, world = self.cyclic
            return idx[idx % world == rank] // world
        sel = idx[(idx >= self.row0) & (idx < self.row0 + self.N)]
        return sel - self.row0

    def local_blocks(self, lo, hi):
        """Global row blocks [lo[t], hi[t]) -> (first, length) of the LOCAL rows this shard owns of each: again contiguous
        blocks, under both ownership rules (block: clip to [row0, row0+N); cyclic: local row = i // world for i % world == rank)."""
        import numpy as _np
        lo = _np.asarray(lo, dtype=_np.int64)
        hi = _np.asarray(hi, dtype=_np.int64)
        if self.N == self.N_total and self.cyclic is None:
            return lo, hi - lo
        if self.cyclic is not None:
            rank, world = self.cyclic
            f = (lo - rank + world - 1) // world          # first local row whose global index is >= lo
            e = (hi - rank + world - 1) // world          # first local row whose global index is >= hi
            f = _np.maximum(f, 0)
            return f, _np.maximum(e - f, 0)
        f = _np.clip(lo, self.row0, self.row0 + self.N) - self.row0
        e = _np.clip(hi, self.row0, self.row0 + self.N) - self.row0
        return f, e - f

    def local_slice(self, vec):
        """The entries of a global N_total-vector (L, gamma) that belong to this shard's rows."""
        if self.N == self.N_total and self.cyclic is None:
            return vec
        if self.cyclic is not None:
            rank, world = self.cyclic
            return vec[rank::world]
        return vec[self.row0:self.row0 + self.N]

    # reference-style constructors --------------------------------------------------------------------------------
    @staticmethod
    def least_squares(A, b, lam=1.0, **kw):
        """f_i = LeastSquares(A[i:i,:], b[i:i], lam)  (test/test_lasso.jl:52-54)"""
        return PackedF(L.LOSS_LS, A, b, lam, **kw)

    @staticmethod
    def least_squares_complex(A_pairs, b_pairs, lam=1.0, **kw):
        """Complex T (CIAOAlgorithms.jl:3; test_lasso.jl:3): f_i = LeastSquares(A[i:i,:], b[i:i], lam) with complex A, b given as
        interleaved (re, im) pairs of the real dtype: A_pairs is N x 2n, b_pairs has 2N entries (torch.view_as_real)."""
        return PackedF(L.LOSS_LS_COMPLEX, A_pairs, b_pairs, lam, **kw)

    @staticmethod
    def logistic(A, y, **kw):
        """f_i = Precompose(LogisticLoss([y_i], 1.0), A[i:i,:], 1.0)  (test/test_logistic_l1.jl:36)"""
        return PackedF(L.LOSS_LOGISTIC, A, y, 1.0, **kw)

    @staticmethod
    def zero(N, d, dtype):
        return PackedF(L.LOSS_ZERO, None, None, 0.0, d=d, dtype=dtype, N=N)


class PackedSepQuad:
    """F = [f_1..f_N] with f_i = Sum(Quadratic(diagm(Q_i), q_i), SqrDistL2(IndBox(lo, hi), eta)) packed for the device
    (the operator family of test/test_sharing.jl:16-25): q row-major N x d; Q the N x d diagonals, or N x d x d dense blocks."""

    def __init__(self, Q: torch.Tensor, q: torch.Tensor, eta: float = 0.0, lo: float = 0.0, hi: float = 0.0,
                 N_total: int | None = None, row0: int = 0):
        assert Q.is_cuda and q.is_cuda and q.dim() == 2 and Q.dtype == q.dtype and Q.dtype in _DT
        assert Q.is_contiguous() and q.is_contiguous()
        # Q of shape N x d: the diagonals (the reference test's diagm); N x d x d: one dense matrix per agent
        self.dense = Q.dim() == 3
        assert (Q.shape == (q.shape[0], q.shape[1], q.shape[1])) if self.dense else (Q.shape == q.shape)
        self.Q, self.q, self.eta, self.lo, self.hi = Q, q, float(eta), float(lo), float(hi)
        self.N, self.d = int(q.shape[0]), int(q.shape[1])
        self.dtype, self.device = Q.dtype, Q.device
        self.N_total = int(N_total) if N_total is not None else self.N
        self.row0, self.cyclic = int(row0), None
        self._c = L.SepQuad(_DT[self.dtype], int(self.dense), self.N, self.d, self.d, self.N_total, Q.data_ptr() if self.N else None,
                            q.data_ptr() if self.N else None, self.eta, self.lo, self.hi)

    @property
    def ref(self):
        return C.byref(self._c)

    localise = PackedF.localise
    local_slice = PackedF.local_slice
    local_blocks = PackedF.local_blocks


class ProxG:
    """g packed for the device: Zero / NormL1(λ) / IndBox(lo, hi)."""

    def __init__(self, kind=L.PROX_ZERO, lam=0.0, lo=-float("inf"), hi=float("inf"), lo_vec=None, hi_vec=None):
        self.kind, self.lam, self.lo, self.hi = int(kind), float(lam), float(lo), float(hi)
        self.lo_vec, self.hi_vec = lo_vec, hi_vec  # keep the device tensors alive
        self._c = L.ProxDesc(self.kind, 0, self.lam, self.lo, self.hi,
                             lo_vec.data_ptr() if lo_vec is not None else None,
                             hi_vec.data_ptr() if hi_vec is not None else None)

    @property
    def ref(self):
        return C.byref(self._c)


class _ChainBatch:
    """Context manager of Context.chain_batch()."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        L.check(self.ctx.lib.ciao_ctx_chain_batch_begin(self.ctx._h))
        self.ctx._batch_keep = []
        return self.ctx

    def __exit__(self, et, ev, tb):
        keep, self.ctx._batch_keep = self.ctx._batch_keep, None
        st = self.ctx.lib.ciao_ctx_chain_batch_end(self.ctx._h, 1 if et is None else 0)
        self.ctx._batch_last_idx = keep      # (the launch reads them: alive until the next batch or the ctx goes)
        if et is None:
            L.check(st)
        return False


class Context:
    """One device + one HIP stream + the library's private workspace (ciao_ctx).  Not thread-safe."""

    def __init__(self, device: int | None = None, stream: "torch.cuda.Stream | None" = None):
        self.lib = L.load()  # raises if the extension is not built
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.device = int(device)
        self._h = C.c_void_p()
        if stream is None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(self.device)
        self.stream = stream
        handle = C.c_void_p(stream.cuda_stream) if stream is not None else None
        L.check(self.lib.ciao_ctx_create(self.device, handle, C.byref(self._h)))
        self._hook_keepalive = None
        self._batch_keep = None        # index tensors of an open chain batch (kept alive until its launch is enqueued)
        self.shards = None

    # -- lifetime ----------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ciao_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        """Wait for the stream and surface a sticky device-side error (an out-of-range sample index)."""
        L.check(self.lib.ciao_ctx_synchronize(self._h))

    def set_option(self, key: str, value: int):
        L.check(self.lib.ciao_ctx_set_option(self._h, key.encode(), int(value)))

    def timing_enable(self, on: bool = True):
        L.check(self.lib.ciao_ctx_timing_enable(self._h, 1 if on else 0))

    def timing_read(self) -> tuple[float, int]:
        """(summed device ms of the dominant-kernel launches since the last read, their count); synchronises."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        L.check(self.lib.ciao_ctx_timing_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_kernel(self) -> str:
        return self.lib.ciao_ctx_last_kernel(self._h).decode()

    def set_allreduce(self, fn):
        """fn(buf_ptr:int, count:int, dtype:int, stream_ptr:int) -> int (0 = ok); None clears the hook."""
        if fn is None:
            self._hook_keepalive = None
            L.check(self.lib.ciao_ctx_set_allreduce(self._h, L.ALLREDUCE_FN(0), None))
            return

        def _tramp(_user, buf, count, dtype, stream):
            try:
                return int(fn(buf, count, dtype, stream))
            except Exception as e:  # never let a Python exception unwind through C
                import sys
                print(f"[ciao] all-reduce hook raised: {e!r}", file=sys.stderr)
                return 1

        cb = L.ALLREDUCE_FN(_tramp)
        self._hook_keepalive = cb
        L.check(self.lib.ciao_ctx_set_allreduce(self._h, cb, None))

    def set_monitor(self, g: "ProxG | None", obj: "torch.Tensor | None"):
        """Objective monitor (include/ciao_hip.h: ciao_ctx_set_monitor): `obj` is a device float64[3] that every full pass
        fills with {F(x), (1/N) sum f_i(x), g(x)}; obj=None switches it off."""
        if obj is None:
            self._monitor_keepalive = None
            L.check(self.lib.ciao_ctx_set_monitor(self._h, None, None))
            return
        assert obj.is_cuda and obj.dtype == torch.float64 and obj.is_contiguous() and obj.numel() >= 3
        self._monitor_keepalive = (g, obj)
        L.check(self.lib.ciao_ctx_set_monitor(self._h, g.ref if g is not None else None, _ptr(obj)))

    def set_shards(self, table: "L.ShardTable | None"):
        """Row-sharded problem for the sequential chains (include/ciao_hip.h: ciao_ctx_set_shards); None removes it."""
        L.check(self.lib.ciao_ctx_set_shards(self._h, C.byref(table) if table is not None else None))
        self.shards = table

    def set_peers(self, group: "parallel.PeerGroup | None"):
        """One-shot peer all-reduce (include/ciao_hip.h: ciao_ctx_set_peers): `group` is a parallel.PeerGroup whose mailboxes are
        exchanged and mapped; None turns it off."""
        if group is None:
            self._peers_keepalive = None
            L.check(self.lib.ciao_ctx_set_peers(self._h, 0, 0, None, 0))
            return
        self._peers_keepalive = group
        arr = (C.c_void_p * group.world)(*group.mailboxes)
        L.check(self.lib.ciao_ctx_set_peers(self._h, group.rank, group.world, arr, group.max_elems))

    def peer_allreduce(self, t: torch.Tensor):
        """In-place sum over the ranks of a float32 / float64 device tensor through the peer mailboxes (bench / tests)."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        L.check(self.lib.ciao_peer_allreduce(self._h, L.F64 if t.dtype == torch.float64 else L.F32, t.numel(), C.c_void_p(t.data_ptr())))

    def set_rccl(self, comm):
        """Native all-reduce: `comm` is a parallel.RcclComm (or None to clear).  The library then calls ncclAllReduce itself on
        its stream -- no Python callback per reduction (include/ciao_hip.h: ciao_ctx_set_rccl)."""
        if comm is None:
            self._rccl_keepalive = None
            L.check(self.lib.ciao_ctx_set_rccl(self._h, None, None))
            return
        self._rccl_keepalive = comm
        self._hook_keepalive = None
        L.check(self.lib.ciao_ctx_set_rccl(self._h, comm.handle, comm.lib_path.encode()))

    # -- helpers -------------------------------------------------------------------------------------------------------
    def _vec(self, t: torch.Tensor, p: PackedF, name: str, n: int | None = None):
        n = p.d if n is None else n
        # A length-padded_from VIEW of a longer buffer is taken for the p.d-vector it heads -- only for a problem whose rows the host
        # mirror padded itself (PackedF.padded_from, set by operators.pack_F(pad_to=...): solvers._Iterable owns those buffers and
        # made them p.d long).  For every other problem the length must be exact: the kernels read and write all p.d coordinates,
        # and a caller's big[:d-1] would be written one element past its end.
        ok = isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == p.dtype and t.is_contiguous()
        if ok and t.numel() != n:
            es = t.element_size()
            ok = (n == p.d and getattr(p, "padded_from", None) is not None and t.numel() == p.padded_from and
                  t.untyped_storage().nbytes() >= (t.storage_offset() + n) * es)
        if not ok:
            raise ValueError(f"{name}: need a contiguous {p.dtype} device vector of length {n}, got "
                             f"{getattr(t, 'dtype', type(t))} {tuple(getattr(t, 'shape', ()))} on {getattr(t, 'device', '?')}")
        return _ptr(t)

    def _idx(self, idx) -> torch.Tensor:
        if isinstance(idx, torch.Tensor):
            assert idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()
            return idx
        t = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(f"cuda:{self.device}", non_blocking=False)
        # The (synchronous) upload ran on torch's current stream; the kernels that read `t` run on the ctx's stream and may
        # still be running when the caller drops `t`: tell the caching allocator, or it hands the block out again too early.
        if self.stream is not None:
            t.record_stream(self.stream)
        return t

    def sample_uniform(self, seed: int, pos: int, N: int, m: int) -> torch.Tensor:
        """m uniform draws from 0..N-1 of the injected index stream (outputs pos .. pos+m-1), generated on the device
        (ciao_sample_uniform): the values sampling.IndexStream.rand_indices produces on the host."""
        out = torch.empty(int(m), dtype=torch.int64, device=f"cuda:{self.device}")
        L.check(self.lib.ciao_sample_uniform(self._h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(int(pos)), int(N), int(m),
                                             C.c_void_p(out.data_ptr())))
        if self.stream is not None:
            out.record_stream(self.stream)
        return out

    # -- L1 plugin API ----------------------------------------------------------------------------------------------
    def gradient(self, p: PackedF, i: int, x, y, fval=None):
        L.check(self.lib.ciao_gradient(self._h, p.ref, int(i), self._vec(x, p, "x"), self._vec(y, p, "y"), _ptr(fval)))

    def prox(self, g: ProxG, x: torch.Tensor, gamma: float, y: torch.Tensor):
        assert x.is_cuda and y.is_cuda and x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous()
        L.check(self.lib.ciao_prox(self._h, _DT[x.dtype], x.numel(), g.ref, _ptr(x), float(gamma), _ptr(y)))

    # -- sweep -------------------------------------------------------------------------------------------------------------
    def full_gradient(self, p: PackedF, x, av):
        L.check(self.lib.ciao_full_gradient(self._h, p.ref, self._vec(x, p, "x"), self._vec(av, p, "av")))

    def proxgrad_step(self, p: PackedF, g: ProxG, gamma: float, x, av, y):
        L.check(self.lib.ciao_proxgrad_step(self._h, p.ref, g.ref, float(gamma), self._vec(x, p, "x"), self._vec(av, p, "av"),
                                            self._vec(y, p, "y")))

    def objective(self, p: PackedF, g: ProxG, x) -> float:
        out = C.c_double(0.0)
        L.check(self.lib.ciao_objective(self._h, p.ref, g.ref, self._vec(x, p, "x"), C.byref(out)))
        return out.value

    # -- SVRG ------------------------------------------------------------------------------------------------------------
    def svrg_init(self, p, x0, av, z, z_full, w):
        L.check(self.lib.ciao_svrg_init(self._h, p.ref, self._vec(x0, p, "x0"), self._vec(av, p, "av"), self._vec(z, p, "z"),
                                        self._vec(z_full, p, "z_full"), self._vec(w, p, "w")))

    def chain_batch(self):
        """`with ctx.chain_batch(): ...` -- the svrg_inner / saga_steps / small-batch finito_steps calls inside are recorded and launched TOGETHER on leaving
        the block, one workgroup per chain (include/ciao_hip.h: ciao_ctx_chain_batch_begin / _end): several independent solves
        over the same rows (a regularisation path, folds, restarts) on as many compute units.  The chains must own their state
        vectors / tables; each one's results are bitwise those of the same call made alone.  An exception inside the block
        drops the records."""
        return _ChainBatch(self)

    def svrg_inner(self, p, g, gamma, idx, av, z, z_full, w):
        idx = self._idx(idx)
        if self._batch_keep is not None:
            self._batch_keep.append(idx)
        L.check(self.lib.ciao_svrg_inner(self._h, p.ref, g.ref, float(gamma), idx.numel(), _ptr(idx), self._vec(av, p, "av"),
                                         self._vec(z, p, "z"), self._vec(z_full, p, "z_full"), self._vec(w, p, "w")))
        return idx

    def svrg_iterate(self, p, g, gamma, idx, plus, av, z, z_full, w, reuse_rowdots: bool = False):
        """reuse_rowdots=True: the caller vouches that A, b and z_full are exactly what the previous svrg_init /
        svrg_iterate on this context left (include/ciao_hip.h: ciao_svrg_iterate); the inner cycle then reuses the
        a_i'z_full of that full pass.  The default never reads anything cached."""
        idx = self._idx(idx)
        L.check(self.lib.ciao_svrg_iterate(self._h, p.ref, g.ref, float(gamma), idx.numel(), _ptr(idx), 1 if plus else 0,
                                           1 if reuse_rowdots else 0, self._vec(av, p, "av"), self._vec(z, p, "z"),
                                           self._vec(z_full, p, "z_full"), self._vec(w, p, "w")))
        return idx

    def svrg_epoch_tail(self, p, m, plus, av, z, z_full, w):
        """SVRG_basic.jl:84-92 on their own: z_full = z / m; w = z_full unless plus; z = 0; av = full gradient at z_full."""
        L.check(self.lib.ciao_svrg_epoch_tail(self._h, p.ref, int(m), 1 if plus else 0, self._vec(av, p, "av"), self._vec(z, p, "z"),
                                              self._vec(z_full, p, "z_full"), self._vec(w, p, "w")))

    def _ptr_table(self, vecs, p, name):
        arr = (C.c_void_p * len(vecs))(*[self._vec(v, p, f"{name}[{k}]") for k, v in enumerate(vecs)])
        return arr

    def full_gradient_multi(self, p, xs, avs):
        """av_k = (1/N) sum_i grad f_i(x_k) for K iterates in ONE pass over A (ciao_full_gradient_multi: matrix cores where the shape
        allows, K single sweeps otherwise).  xs, avs: lists of K device d-vectors."""
        assert len(xs) == len(avs) and len(xs) >= 1
        L.check(self.lib.ciao_full_gradient_multi(self._h, p.ref, len(xs), self._ptr_table(xs, p, "x"), self._ptr_table(avs, p, "av")))

    def svrg_epoch_tail_multi(self, p, m, plus, avs, zs, z_fulls, ws):
        """SVRG_basic.jl:84-92 for K solves over the same rows: every solve's tail, then the K full passes as one pass over A."""
        K = len(avs)
        assert K >= 1 and len(zs) == K and len(z_fulls) == K and len(ws) == K
        L.check(self.lib.ciao_svrg_epoch_tail_multi(self._h, p.ref, K, int(m), 1 if plus else 0, self._ptr_table(avs, p, "av"),
                                                    self._ptr_table(zs, p, "z"), self._ptr_table(z_fulls, p, "z_full"),
                                                    self._ptr_table(ws, p, "w")))

    # -- SAGA / SAG ----------------------------------------------------------------------------------------------------
    def saga_init(self, p, g, gamma, x0, table, av, z):
        L.check(self.lib.ciao_saga_init(self._h, p.ref, g.ref, float(gamma), self._vec(x0, p, "x0"),
                                        self._vec(table, p, "table", p.N * p.d), self._vec(av, p, "av"), self._vec(z, p, "z")))

    def saga_steps(self, p, g, gamma, sag, idx, table, av, z):
        idx = self._idx(idx)
        if self._batch_keep is not None:
            self._batch_keep.append(idx)
        L.check(self.lib.ciao_saga_steps(self._h, p.ref, g.ref, float(gamma), 1 if sag else 0, idx.numel(), _ptr(idx),
                                         self._vec(table, p, "table", p.N * p.d), self._vec(av, p, "av"), self._vec(z, p, "z")))
        return idx

    # -- Finito / LFinito -------------------------------------------------------------------------------------------
    def hat_gamma(self, gam: torch.Tensor) -> float:
        out = C.c_double(0.0)
        L.check(self.lib.ciao_hat_gamma(self._h, _DT[gam.dtype], gam.numel(), _ptr(gam), C.byref(out)))
        return out.value

    def finito_init(self, p, g, gam, hat_gamma, x0, table, av, z):
        L.check(self.lib.ciao_finito_init(self._h, p.ref, g.ref, self._vec(gam, p, "gam", p.N), float(hat_gamma),
                                          self._vec(x0, p, "x0"), self._vec(table, p, "table", p.N * p.d),
                                          self._vec(av, p, "av"), self._vec(z, p, "z")))

    def finito_steps(self, p, g, gam, hat_gamma, bptr: np.ndarray, bidx, table, av, z):
        bptr = np.ascontiguousarray(bptr, dtype=np.int64)
        bidx = self._idx(bidx)
        if self._batch_keep is not None:
            self._batch_keep.append(bidx)
        assert bptr[0] == 0 and bptr[-1] == bidx.numel()
        L.check(self.lib.ciao_finito_steps(self._h, p.ref, g.ref, self._vec(gam, p, "gam", p.N), float(hat_gamma),
                                           len(bptr) - 1, C.c_void_p(bptr.ctypes.data), _ptr(bidx),
                                           self._vec(table, p, "table", p.N * p.d), self._vec(av, p, "av"),
                                           self._vec(z, p, "z")))
        return bidx

    @staticmethod
    def _blocks(first, length):
        first = np.ascontiguousarray(first, dtype=np.int64)
        length = np.ascontiguousarray(length, dtype=np.int64)
        assert first.shape == length.shape and first.ndim == 1
        return first, length

    def finito_steps_blocks(self, p, g, gam, hat_gamma, first, length, table, av, z):
        """Batches given as contiguous local row blocks [first[t], first[t] + length[t]) -- no index array (ciao_finito_steps_blocks)."""
        first, length = self._blocks(first, length)
        L.check(self.lib.ciao_finito_steps_blocks(self._h, p.ref, g.ref, self._vec(gam, p, "gam", p.N), float(hat_gamma), first.size,
                                                  C.c_void_p(first.ctypes.data), C.c_void_p(length.ctypes.data),
                                                  self._vec(table, p, "table", p.N * p.d), self._vec(av, p, "av"), self._vec(z, p, "z")))

    def lfinito_iterate_blocks(self, p, g, gam, hat_gamma, first, length, av, z, z_full):
        first, length = self._blocks(first, length)
        L.check(self.lib.ciao_lfinito_iterate_blocks(self._h, p.ref, g.ref, self._vec(gam, p, "gam", p.N), float(hat_gamma), first.size,
                                                     C.c_void_p(first.ctypes.data), C.c_void_p(length.ctypes.data),
                                                     self._vec(av, p, "av"), self._vec(z, p, "z"), self._vec(z_full, p, "z_full")))

    def proshi_steps_blocks(self, f, g, gam, hat_gamma, first, length, table, av, z):
        first, length = self._blocks(first, length)
        L.check(self.lib.ciao_proshi_steps_blocks(self._h, f.ref, g.ref, self._vec(gam, f, "gam", f.N), float(hat_gamma), first.size,
                                                  C.c_void_p(first.ctypes.data), C.c_void_p(length.ctypes.data),
                                                  self._vec(table, f, "table", f.N * f.d), self._vec(av, f, "av"), self._vec(z, f, "z")))

    def lfinito_init(self, p, hat_gamma, x0, av, z, z_full):
        L.check(self.lib.ciao_lfinito_init(self._h, p.ref, float(hat_gamma), self._vec(x0, p, "x0"), self._vec(av, p, "av"),
                                           self._vec(z, p, "z"), self._vec(z_full, p, "z_full")))

    def lfinito_iterate(self, p, g, gam, hat_gamma, bptr: np.ndarray, bidx, av, z, z_full):
        bptr = np.ascontiguousarray(bptr, dtype=np.int64)
        bidx = self._idx(bidx)
        assert bptr[0] == 0 and bptr[-1] == bidx.numel()
        L.check(self.lib.ciao_lfinito_iterate(self._h, p.ref, g.ref, self._vec(gam, p, "gam", p.N), float(hat_gamma),
                                              len(bptr) - 1, C.c_void_p(bptr.ctypes.data), _ptr(bidx), self._vec(av, p, "av"),
                                              self._vec(z, p, "z"), self._vec(z_full, p, "z_full")))
        return bidx

    # -- adaptive Finito -------------------------------------------------------------------------------------------------
    def afinito_init(self, p, g, alpha, x0, table, meta, av, z, hat_gamma_dev, gam_override=None):
        L.check(self.lib.ciao_afinito_init(self._h, p.ref, g.ref, float(alpha), self._vec(x0, p, "x0"),
                                           self._vec(table, p, "table", p.N * p.d), self._vec(meta, p, "meta", p.N * 16),
                                           self._vec(av, p, "av"), self._vec(z, p, "z"), self._vec(hat_gamma_dev, p, "hat_gamma", 1),
                                           None if gam_override is None else self._vec(gam_override, p, "gam_override", p.N)))

    def afinito_probe(self, p, i: int, x0, signs, t: float) -> float:
        """One retry of the Lipschitz probe for sample i at x0 + t*signs (Finito_adaptive.jl:80-82); synchronises."""
        out = C.c_double(0.0)
        n_signs = p.d // 2 if getattr(p, "complex", False) else p.d     # complex T: one real draw per complex entry
        L.check(self.lib.ciao_afinito_probe(self._h, p.ref, int(i), self._vec(x0, p, "x0"), self._vec(signs, p, "signs", n_signs),
                                            float(t), C.byref(out)))
        return out.value

    def afinito_steps(self, p, g, alpha, tol_b, idx, table, meta, av, z, hat_gamma_dev) -> tuple[int, int]:
        """-> (steps completed, backtracking trials); synchronises."""
        idx = self._idx(idx)
        done, trials = C.c_int64(0), C.c_int64(0)
        L.check(self.lib.ciao_afinito_steps(self._h, p.ref, g.ref, float(alpha), float(tol_b), idx.numel(), _ptr(idx),
                                            self._vec(table, p, "table", p.N * p.d), self._vec(meta, p, "meta", p.N * 16),
                                            self._vec(av, p, "av"), self._vec(z, p, "z"), self._vec(hat_gamma_dev, p, "hat_gamma", 1),
                                            C.byref(done), C.byref(trials)))
        return done.value, trials.value

    # -- ProShI ------------------------------------------------------------------------------------------------------------
    def proshi_init(self, f, g, gam, x0, table, av, z, hat_gamma_dev):
        L.check(self.lib.ciao_proshi_init(self._h, f.ref, g.ref, self._vec(gam, f, "gam", f.N), self._vec(x0, f, "x0"),
                                          self._vec(table, f, "table", f.N * f.d), self._vec(av, f, "av"), self._vec(z, f, "z"),
                                          self._vec(hat_gamma_dev, f, "hat_gamma", 1)))

    def proshi_steps(self, f, g, gam, hat_gamma, bptr: np.ndarray, bidx, table, av, z):
        bptr = np.ascontiguousarray(bptr, dtype=np.int64)
        bidx = self._idx(bidx)
        assert bptr[0] == 0 and bptr[-1] == bidx.numel()
        L.check(self.lib.ciao_proshi_steps(self._h, f.ref, g.ref, self._vec(gam, f, "gam", f.N), float(hat_gamma), len(bptr) - 1,
                                           C.c_void_p(bptr.ctypes.data), _ptr(bidx), self._vec(table, f, "table", f.N * f.d),
                                           self._vec(av, f, "av"), self._vec(z, f, "z")))

    def proshi_solution(self, f, gam, z, table):
        L.check(self.lib.ciao_proshi_solution(self._h, f.ref, self._vec(gam, f, "gam", f.N), self._vec(z, f, "z"),
                                              self._vec(table, f, "table", f.N * f.d)))

    # -- synthetic data -------------------------------------------------------------------------------------------------
    def synth_normal(self, out: torch.Tensor, row0: int, seed: int, scale: float):
        assert out.is_cuda and out.dim() == 2 and out.stride(1) == 1
        L.check(self.lib.ciao_synth_normal(self._h, _DT[out.dtype], _ptr(out), out.shape[0], out.shape[1],
                                           out.stride(0) if out.shape[0] > 1 else out.shape[1], int(row0), int(seed), float(scale)))

    def synth_targets(self, p: PackedF, x_true, noise: float, labels: bool, seed: int, b_out):
        L.check(self.lib.ciao_synth_targets(self._h, p.ref, self._vec(x_true, p, "x_true"), float(noise), 1 if labels else 0,
                                            int(p.row0), int(seed), self._vec(b_out, p, "b_out", p.N)))


_default_ctx: dict[int, Context] = {}


def _close_default_contexts():
    # destroy the library workspaces while the HIP runtime is still alive (not from __del__ during interpreter teardown)
    for c in list(_default_ctx.values()):
        try:
            c.close()
        except Exception:
            pass
    _default_ctx.clear()


import atexit  # noqa: E402

atexit.register(_close_default_contexts)


def default_context() -> Context:
    """A per-device Context bound to torch's current stream at first use."""
    dev = torch.cuda.current_device()
    ctx = _default_ctx.get(dev)
    if ctx is None:
        ctx = Context(dev)
        _default_ctx[dev] = ctx
    return ctx
