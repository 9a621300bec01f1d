#!/usr/bin/env python3
"""Soak of rows_long_kernel (a cluster of workgroups per row, rows beyond 64 KiB) on cuda:0: the same sweep launched hundreds of times
must give the same bits every time (the exchange adds the segments' partial dot products in one fixed tree whatever the timing), for
several row lengths, row counts (fewer rows than clusters, one row more than a multiple of the clusters, thousands of rows per cluster),
both segment sizes and both types; Finito batches of every size 1 .. 64 in a row against the table invariant.  A timeout of the
exchange (error word 6) or an out-of-order sum shows up as an exception or a mismatch.  Prints one line per shape."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
torch.cuda.set_device(0)
ctx = Context(0)
dev = torch.device("cuda", 0)
REPS = int(os.environ.get("CIAO_SOAK_REPS", "300"))
t_all = time.time()
for dt, d, N, opts in ((torch.float64, 8194, 20000, {}), (torch.float64, 32768, 63, {}), (torch.float64, 32768, 65, {}), (torch.float64, 32768, 30000, {}),
                       (torch.float64, 32768, 30000, {"long_j": 4}), (torch.float64, 131072, 1000, {}), (torch.float64, 262144, 9, {}),
                       (torch.float32, 65536, 20000, {}), (torch.float32, 20004, 50000, {"long_j": 4}), (torch.float32, 262144, 300, {"split_blocks_per_cu": 1})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    A = torch.empty((N, d), dtype=dt, device=dev)
    b = torch.empty(N, dtype=dt, device=dev)
    ctx.synth_normal(A, 0, 1, 1.0 / np.sqrt(d))
    ctx.synth_normal(b.view(N, 1), 0, 2, 1.0)
    F = PackedF(L.LOSS_LS, A, b, float(N))
    x = torch.full((d,), 0.01, dtype=dt, device=dev)
    av0, av = torch.empty_like(x), torch.empty_like(x)
    ctx.full_gradient(F, x, av0)
    name = ctx.last_kernel()
    assert "rows_long_kernel" in name, name
    bad = 0
    t0 = time.time()
    for _ in range(REPS):
        ctx.full_gradient(F, x, av)
        bad += int(not torch.equal(av, av0))
    ctx.synchronize()
    print(f"{'f64' if dt == torch.float64 else 'f32'} d={d} N={N} {opts}: {REPS} sweeps, {bad} differ from the first, {time.time() - t0:.1f} s  [{name}]", flush=True)
    assert bad == 0
    for k in opts:
        ctx.set_option(k, 0)
    del A, b, F
    torch.cuda.empty_cache()
# Finito batches of every size 1 .. 64 over row blocks, d = 20000 fp64: the invariant av == hat_gamma * sum_i s_i / gamma_i after all of them
N, d, dt = 4000, 20000, torch.float64
A = torch.empty((N, d), dtype=dt, device=dev)
b = torch.empty(N, dtype=dt, device=dev)
ctx.synth_normal(A, 0, 3, 1.0 / np.sqrt(d))
ctx.synth_normal(b.view(N, 1), 0, 4, 1.0)
F = PackedF(L.LOSS_LS, A, b, float(N))
g = ProxG(L.PROX_L1, lam=0.01)
gam = torch.full((N,), 0.4, dtype=dt, device=dev)
hg = ctx.hat_gamma(gam)
x0 = torch.zeros(d, dtype=dt, device=dev)
table = torch.empty((N, d), dtype=dt, device=dev)
av, z = torch.empty_like(x0), torch.empty_like(x0)
ctx.finito_init(F, g, gam, hg, x0, table, av, z)
ctx.set_option("chain_max_batch", 0)
first, nb = 0, 0
for r in list(range(1, 65)) * 2:
    if first + r > N:
        first = 0
    ctx.finito_steps_blocks(F, g, gam, hg, np.array([first], np.int64), np.array([r], np.int64), table, av, z)
    first += r
    nb += 1
ctx.synchronize()
inv = (table / gam[:, None]).sum(dim=0) * hg
err = float((av - inv).abs().max() / inv.abs().max())
print(f"f64 d={d}: {nb} Finito batches of 1 .. 64 rows [{ctx.last_kernel()}]: |av - hat_gamma sum s_i/gamma_i| / |.| = {err:.2e}", flush=True)
assert err < 1e-12
print(f"soak passed in {time.time() - t_all:.0f} s")
