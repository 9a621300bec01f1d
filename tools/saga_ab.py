#!/usr/bin/env python3
"""SAGA step time, one box, every route a SAGA chain can take at d = CIAO_D (default 1024), N = CIAO_N (default 1M):

    dtype in {fp32, fp64}  x  {chain_ws_kernel (default), chain_dma_kernel (chain_no_ws=1)}  x  {one allocation, shard table}

The shard table is made of two slices of the same matrix / table (ciao_ctx_set_shards in process: same addresses, so the
sharded run is the unsharded run plus the address resolution through the table), and the results of all four routes of a dtype
are compared BITWISE.  us per update over CIAO_M steps (default 400k), indices on the device beforehand.
VERDICT r3 item 3: fp64 SAGA had never been timed; the sharded chain_dma SAGA carried 520 bytes of scratch."""
import hashlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
d = int(os.environ.get("CIAO_D", "1024"))
N = int(os.environ.get("CIAO_N", "1000000"))
m = int(os.environ.get("CIAO_M", "400000"))
loss = os.environ.get("CIAO_LOSS", "logistic")
reps = int(os.environ.get("CIAO_REPS", "1"))
dtypes = {"f32": (torch.float32,), "f64": (torch.float64,)}.get(os.environ.get("CIAO_DTYPE", ""), (torch.float32, torch.float64))
for tdt in dtypes:
    A = torch.empty((N, d), dtype=tdt, device="cuda"); y = torch.empty((N,), dtype=tdt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    logistic = loss == "logistic"
    F = PackedF(L.LOSS_LOGISTIC if logistic else L.LOSS_LS, A, y, 1.0 if logistic else float(N))
    ctx.synth_targets(F, torch.ones(d, dtype=tdt, device="cuda"), 0.1, logistic, 1, y)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    gamma = 1.0 if logistic else 1.0 / (3.0 * 1.3 * N)
    x0 = torch.ones(d, dtype=tdt, device="cuda")
    table = torch.empty((N, d), dtype=tdt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    idx = ctx._idx(IndexStream(0).rand_indices(N, m))
    cut = (N // 3) // 64 * 64

    def shard_table():
        t = L.ShardTable()
        t.nshards, t.owner = 2, 1
        for k, r0 in enumerate((0, cut)):
            t.row0[k] = r0
            t.A[k], t.b[k], t.table[k] = A[r0:].data_ptr(), y[r0:].data_ptr(), table[r0:].data_ptr()
        t.row0[2] = N
        return t

    results = {}
    for route, opts in (("ws", {}), ("dma", {"chain_no_ws": 1})) * reps:
        for sharded in (False, True):
            ctx.set_option("chain_no_ws", 0)
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.saga_init(F, g, gamma, x0, table, av, z)
            if sharded:
                ctx.set_shards(shard_table())
            try:
                ctx.saga_steps(F, g, gamma, False, idx[:2000], table, av, z); ctx.synchronize()
                t0 = time.perf_counter(); ctx.saga_steps(F, g, gamma, False, idx[2000:], table, av, z); ctx.synchronize()
                t = time.perf_counter() - t0
                kern = ctx.last_kernel()
            finally:
                ctx.set_shards(None)
            results[(route, sharded)] = (z.clone(), av.clone(), table[idx[:4096]].clone())
            print(f"{str(tdt).split('.')[-1]} d={d} N={N} {loss:8s} {route:3s} {'sharded  ' if sharded else 'unsharded'} "
                  f"{t / (m - 2000) * 1e6:.3f} us/update  ({kern.split(' grid')[0]})", flush=True)
    ctx.set_option("chain_no_ws", 0)
    ref = results[("ws", False)]
    same = all(all(torch.equal(a, b) for a, b in zip(ref, r)) for r in results.values())
    hh = hashlib.sha1()
    for tt in ref:
        hh.update(tt.detach().cpu().numpy().tobytes())
    print(f"{str(tdt).split('.')[-1]}: all four routes bitwise equal (z, av, 4096 table rows): {same} [{hh.hexdigest()[:10]}]", flush=True)
    del A, y, table, F, results
    torch.cuda.empty_cache()
