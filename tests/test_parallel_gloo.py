"""N > 1 path on CPU: two processes, `gloo` backend, rendezvous on 127.0.0.1.

Each rank owns a contiguous row shard (parallel.shard_rows), computes its partial d-vector sum (here with the CPU
oracle, since this container has no GPU -- on the GPU the rows kernel produces exactly this partial), then the library's
all-reduce hook object (parallel.AllReduceHook, the same object bench.py installs with the RCCL backend) sums the
partials in place.  Every rank must end with the full-problem gradient, bitwise identical across ranks.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.parallel import AllReduceHook, shard_rows
    from oracle import oracle as O
    import problems as P
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, d = 103, 37
        A, b, x = P.synthetic("logistic", N, d, np.float64, seed=5)
        row0, n = shard_rows(N, rank, world)
        # local partial: sum over my rows of grad f_i(x)  (= n_local * orc_full_pass on the shard)
        shard = O.Problem("logistic", A[row0:row0 + n], b[row0:row0 + n], 1.0)
        partial = O.full_pass(shard, x) * n
        buf = np.concatenate([partial, [float(n)]])   # d + 1 scalars, like the library's sumbuf
        hook = AllReduceHook(torch.device("cpu"))
        assert hook(buf.ctypes.data, d + 1, L.F64, 0) == 0
        full = O.full_pass(O.Problem("logistic", A, b, 1.0), x)
        ok = np.allclose(buf[:d] / N, full, rtol=1e-13, atol=1e-15) and buf[d] == N and hook.calls == 1
        gathered = [torch.zeros(d + 1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(buf))
        same = all(torch.equal(gathered[0], g) for g in gathered)
        q.put((rank, bool(ok), bool(same)))
    finally:
        dist.destroy_process_group()


def test_sharded_sum_with_gloo_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(ok for _, ok, _ in res), "all-reduced sharded sum != full-problem gradient"
    assert all(same for _, _, same in res), "ranks disagree bitwise after the all-reduce"
