"""Multi-GPU: one process per GPU, rows of the finite sum sharded, one all-reduce of the d-vector per sweep / batch.

What shards: the full-gradient sweeps (SVRG_basic.jl:58-63,:87-92; Finito_LFinito.jl:85-88) and
the Finito / LFinito batches (Finito_basic.jl:110-117) -- rows are independent, the only coupling is the sum of the
rank-1 terms into one d-vector.  Rank p owns the contiguous rows [row0, row0 + n_local) of A, b, gam and of the
SAGA/Finito table; every d-vector (x, z, w, av, z_full) is replicated and stays bitwise identical across ranks because
every rank applies the same epilogue to the same all-reduced sum.

What does NOT shard: the sequential inner chains (SVRG_basic.jl:73-82, SAGA_basic.jl:53-68): replicas only.

The collective is RCCL (`torch.distributed` backend "nccl" on ROCm) over xGMI.  The message is d+1 scalars (8 KB at
d=1024 fp64): latency-bound, so it is issued once per sweep on the compute stream, never per row block.
libciao_hip.so calls back into `AllReduceHook` between its reduce and epilogue kernels (ciao_ctx_set_allreduce).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def shard_rows(N_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition: returns (row0, n_local); the first N_total % world ranks get one extra row."""
    assert 0 <= rank < world and N_total >= 0
    base, rem = divmod(N_total, world)
    n_local = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, n_local


def shard_rows_cyclic(N_total: int, rank: int, world: int) -> int:
    """Round-robin ownership (global row i belongs to rank i % world, local row i // world): number of local rows.
    Use with PackedF(..., cyclic=(rank, world)) when static contiguous Finito batches should span every rank."""
    assert 0 <= rank < world and N_total >= 0
    return (N_total - rank + world - 1) // world


class _DeviceBuffer:
    """Zero-copy view of library-owned device memory through the CUDA array interface (HIP pointers on ROCm)."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def _wrap(ptr: int, count: int, dtype: int, device: torch.device) -> torch.Tensor:
    np_dt = np.float64 if dtype == L.F64 else np.float32
    if device.type == "cpu":
        ctype = C.c_double if dtype == L.F64 else C.c_float
        arr = np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), C.POINTER(ctype)), shape=(count,))
        return torch.from_numpy(arr)
    return torch.as_tensor(_DeviceBuffer(ptr, count, np.dtype(np_dt).str), device=device)


class AllReduceHook:
    """Callable handed to Context.set_allreduce: sums `count` scalars at `buf` in place over the process group.

    backend "nccl" (= RCCL): the collective is enqueued on RCCL's stream behind the compute stream.
    backend "gloo" with device tensors (single-box rehearsals): staged through host memory.
    """

    def __init__(self, device: torch.device | str, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.device = torch.device(device)
        self.group = group
        self.calls = 0
        self.bytes = 0
        self._views = {}      # (ptr, count, dtype) -> tensor view of the library's buffer (it is reused call after call)
        self._backend = None

    def __call__(self, buf: int, count: int, dtype: int, stream: int) -> int:
        dist = self.dist
        key = (buf, count, dtype)
        t = self._views.get(key)
        if t is None:
            if len(self._views) > 64:
                self._views.clear()
            t = self._views[key] = _wrap(buf, count, dtype, self.device)
        if self._backend is None:
            self._backend = dist.get_backend(self.group)
        backend = self._backend
        if self.device.type == "cuda" and backend != "nccl":
            if stream:
                torch.cuda.ExternalStream(stream, device=self.device).synchronize()
            else:
                torch.cuda.current_stream(self.device).synchronize()
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        elif self.device.type == "cuda" and stream and stream != torch.cuda.current_stream(self.device).cuda_stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        self.bytes += count * (8 if dtype == L.F64 else 4)
        return 0


class ShardGroup:
    """The sequential chains (SVRG inner cycle, SAGA steps) on a row-sharded problem: ONE rank -- the owner -- runs the chain
    and reads the other ranks' rows through peer-mapped pointers over xGMI (include/ciao_hip.h:
    ciao_ctx_set_shards).  `install` exchanges HIP IPC handles of every rank's A, b (and SAGA table shard) through the process
    group (adaptive Finito: also of its s-table and per-sample scalars), opens them on the owner and hands the library the shard table; the other ranks install the same partition
    without pointers.  Contiguous block partition only (parallel.shard_rows); an all-reduce hook must be installed too."""

    def __init__(self, ctx, owner: int = 0, group=None):
        self.ctx, self.owner, self.group = ctx, int(owner), group
        self._opened = []     # (ptr, offset) mappings to close
        self._keep = None

    def _export(self, t):
        if t is None:
            return None
        h = (C.c_char * 64)()
        off = C.c_int64(0)
        L.check(self.ctx.lib.ciao_ipc_export(C.c_void_p(t.data_ptr()), h, C.byref(off)))
        return bytes(h), int(off.value)

    def _open(self, exported):
        if exported is None:
            return None
        h, off = exported
        buf = (C.c_char * 64).from_buffer_copy(h)
        out = C.c_void_p()
        L.check(self.ctx.lib.ciao_ipc_open(buf, off, C.byref(out)))
        self._opened.append((out.value, off))
        return out.value

    def install(self, F, table=None, meta=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        if world > L.MAX_SHARDS:
            raise ValueError(f"at most {L.MAX_SHARDS} shards (one node)")
        if F.cyclic is not None:
            raise ValueError("sharded chains need the contiguous block partition (parallel.shard_rows)")
        self.close()
        mine = {"n": F.N, "row0": F.row0, "ld": F.ld, "A": self._export(F.A) if F.N else None,
                "b": self._export(F.b) if F.N else None, "table": self._export(table) if (table is not None and F.N) else None,
                "meta": self._export(meta) if (meta is not None and F.N) else None}
        every = [None] * world
        dist.all_gather_object(every, mine, group=self.group)
        row0 = 0
        for k, e in enumerate(every):
            if e["row0"] != row0 or (e["n"] and e["ld"] != F.ld):
                raise ValueError(f"shard {k} starts at row {e['row0']} (expected {row0}) or has another row stride")
            row0 += e["n"]
        if row0 != F.N_total:
            raise ValueError(f"the shards hold {row0} rows, the problem has N_total = {F.N_total}")
        tbl = L.ShardTable()
        tbl.nshards, tbl.owner = world, 1 if rank == self.owner else 0
        acc = 0
        for k, e in enumerate(every):
            tbl.row0[k] = acc
            acc += e["n"]
            if rank != self.owner or e["n"] == 0:
                continue
            if k == rank:   # the owner's own shard: its local pointers
                tbl.A[k], tbl.b[k] = F.A.data_ptr(), F.b.data_ptr()
                tbl.table[k] = table.data_ptr() if table is not None else None
                tbl.meta[k] = meta.data_ptr() if meta is not None else None
            else:
                tbl.A[k], tbl.b[k], tbl.table[k] = self._open(e["A"]), self._open(e["b"]), self._open(e["table"])
                tbl.meta[k] = self._open(e["meta"])
        tbl.row0[world] = acc
        self._keep = (F, table, meta)
        self.row0 = [int(tbl.row0[k]) for k in range(world + 1)]
        self.ctx.set_shards(tbl)
        dist.barrier(group=self.group)   # nobody may free / reuse an exported allocation before the owner has opened it
        return tbl

    def close(self):
        if self.ctx.shards is not None:
            self.ctx.synchronize()
            self.ctx.set_shards(None)
        for ptr, off in self._opened:
            self.ctx.lib.ciao_ipc_close(C.c_void_p(ptr), off)
        self._opened = []
        self._keep = None


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]   # rccl.h: NCCL_UNIQUE_ID_BYTES


def default_rccl_path() -> str:
    """The RCCL this process already uses: the one bundled with torch, else the ROCm one."""
    import os
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else "librccl.so"


class PeerGroup:
    """The one-shot peer all-reduce (csrc/peer_kernels.h; SURVEY.md 8e "tuned alternative"): every rank creates a mailbox in its
    own HBM, the HIP IPC handles travel through the torch.distributed group (any backend -- control plane only), every rank maps
    the others' and hands the library the table (Context.set_peers).  From then on the d-vector sums of the sharded sweeps and
    batches are exchanged by the kernels themselves: no collective call per reduction.  max_elems >= d + 1 (2 d for the sharded
    chains' hand-over).  One node (world <= 8)."""

    def __init__(self, ctx, max_elems: int, group=None):
        import torch.distributed as dist
        self.ctx, self.max_elems, self.group = ctx, int(max_elems), group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise ValueError("peer mailboxes serve one node (at most 8 ranks)")
        # Every rank takes part in every collective below whatever happened to it locally: a rank that failed (allocation, export,
        # mapping) says so in the object it contributes, and ALL ranks raise together -- none is left waiting for another.
        self.own, self._opened, self.mailboxes = None, [], []
        mine, err = None, None
        try:
            own = C.c_void_p()
            nbytes = C.c_int64(0)
            L.check(ctx.lib.ciao_peer_mailbox_create(ctx._h, self.max_elems, C.byref(own), C.byref(nbytes)))
            self.own = own.value
            h = (C.c_char * 64)()
            off = C.c_int64(0)
            L.check(ctx.lib.ciao_ipc_export(C.c_void_p(self.own), h, C.byref(off)))
            mine = (bytes(h), int(off.value))
        except Exception as e:   # noqa: BLE001 -- reported to every rank below
            err = repr(e)
        every = [None] * self.world
        dist.all_gather_object(every, (mine, err), group=group)
        if all(e[1] is None for e in every):
            try:
                for r, ((hr, offr), _) in enumerate(every):
                    if r == self.rank:
                        self.mailboxes.append(self.own)
                        continue
                    out = C.c_void_p()
                    L.check(ctx.lib.ciao_ipc_open((C.c_char * 64).from_buffer_copy(hr), offr, C.byref(out)))
                    self._opened.append((out.value, offr))
                    self.mailboxes.append(out.value)
            except Exception as e:   # noqa: BLE001
                err = repr(e)
        # second round: every mailbox is mapped everywhere (or somebody failed) before the first reduction writes into one
        votes = [None] * self.world
        dist.all_gather_object(votes, err if err is not None else next((e[1] for e in every if e[1] is not None), None), group=group)
        bad = [(r, v) for r, v in enumerate(votes) if v is not None]
        if bad:
            self._release()
            raise RuntimeError(f"peer mailboxes could not be set up on every rank: {bad}")

    def _release(self):
        for ptr, off in self._opened:
            self.ctx.lib.ciao_ipc_close(C.c_void_p(ptr), off)
        self._opened = []
        if self.own is not None:
            self.ctx.lib.ciao_peer_mailbox_destroy(self.ctx._h, C.c_void_p(self.own))
            self.own = None

    def close(self):
        import torch.distributed as dist
        if getattr(self, "own", None) is None:
            return
        self.ctx.synchronize()
        if dist.is_initialized():
            dist.barrier(group=self.group)   # nobody is still writing into a mailbox that is about to be unmapped / freed
        self._release()


class RcclComm:
    """An RCCL communicator of our own (one rank per process / GPU), for Context.set_rccl: the d-vector all-reduce is then
    issued by libciao_hip.so itself on its stream.  The unique id travels through the existing torch.distributed group
    (any backend); nothing else of torch is involved.  `handle` is the ncclComm_t."""

    SYMBOLS = ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclAllReduce", "ncclGetErrorString")

    @staticmethod
    def probe(lib_path: str | None = None) -> str | None:
        """Everything that can fail on ONE rank alone -- loading librccl, resolving its symbols -- without any communication:
        None if this rank could build a communicator, else the reason.  Ranks vote on this BEFORE the collective
        ncclCommInitRank (a rank that failed here would otherwise leave the others waiting inside it)."""
        try:
            lib = C.CDLL(lib_path or default_rccl_path(), mode=C.RTLD_LOCAL)
            for name in RcclComm.SYMBOLS:
                getattr(lib, name)
        except (OSError, AttributeError) as e:
            return repr(e)
        return None

    def __init__(self, rank: int, world: int, device: int, group=None, lib_path: str | None = None):
        import torch.distributed as dist
        self.lib_path = lib_path or default_rccl_path()
        self._lib = C.CDLL(self.lib_path, mode=C.RTLD_LOCAL)
        self._lib.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
        self._lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self._lib.ncclGetErrorString.restype = C.c_char_p
        uid = _NcclUniqueId()
        err = None
        if rank == 0:
            r = self._lib.ncclGetUniqueId(C.byref(uid))
            if r != 0:
                err = f"ncclGetUniqueId failed: {self._lib.ncclGetErrorString(r).decode()}"
        if world > 1:
            # rank 0 always broadcasts -- the id, or its failure -- so that no rank is left waiting for the other
            box = [(err, C.string_at(C.addressof(uid), 128)) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            err = box[0][0]
            C.memmove(C.addressof(uid), box[0][1], 128)
        if err:
            raise RuntimeError(err)
        torch.cuda.set_device(device)
        comm = C.c_void_p()
        self._check(self._lib.ncclCommInitRank(C.byref(comm), world, uid, rank), "ncclCommInitRank")
        self.handle, self.rank, self.world = comm, rank, world

    def _check(self, r, what):
        if r != 0:
            raise RuntimeError(f"{what} failed: {self._lib.ncclGetErrorString(r).decode()}")

    def count(self) -> int:
        """ncclCommCount: the number of ranks RCCL itself says this communicator spans."""
        n = C.c_int(0)
        self._check(self._lib.ncclCommCount(self.handle, C.byref(n)), "ncclCommCount")
        return int(n.value)

    def all_reduce(self, t: torch.Tensor, stream=None):
        """In-place sum of a float32/float64 device tensor on `stream` (a torch stream; None = the current one): the very call
        libciao_hip.so makes for the d-vector sums (csrc/api.hip rccl_hook); used by bench.py to time the collective alone."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        s = stream if stream is not None else torch.cuda.current_stream(t.device)
        self._check(self._lib.ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(),
                                            8 if t.dtype == torch.float64 else 7, 0, self.handle, C.c_void_p(s.cuda_stream)),
                    "ncclAllReduce")

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self._lib.ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()


def init_process_group_from_env(backend: str | None = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK from the environment)."""
    import os

    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl" and torch.cuda.is_available():   # bind the communicator to this rank's GPU (no guessing by global rank)
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local
