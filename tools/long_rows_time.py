#!/usr/bin/env python3
"""Times the full-gradient sweep (SVRG_basic.jl:87-92), the SAGA init (SAGA_basic.jl:42-47) and Finito batches (Finito_basic.jl:109-118)
on rows beyond 64 KiB on cuda:0: the cluster kernel (rows_long_kernel, a cluster of workgroups per row) against the generic kernel it
replaces (long_rows=0).  CIAO_D, CIAO_F32, CIAO_GB (size of A, default 8) from the environment, options through CIAO_OPTS=key=value,...
Prints ms per sweep and algorithmic TB/s (N * d * s bytes per sweep; the table modes 2x / 3x that)."""
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
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
d = int(os.environ.get("CIAO_D", "32768"))
dt = torch.float32 if os.environ.get("CIAO_F32") else torch.float64
s = 4 if dt == torch.float32 else 8
N = int(float(os.environ.get("CIAO_GB", "8")) * 1e9 / (d * s))
dev = torch.device("cuda", 0)
A = torch.empty((N, d), dtype=dt, device=dev)
b = torch.empty(N, dtype=dt, device=dev)
ctx.synth_normal(A, 0, seed=7, scale=1.0 / np.sqrt(d))
ctx.synth_normal(b.view(N, 1), 0, seed=8, scale=1.0)
f = PackedF(L.LOSS_LS, A, b, float(N))
g = ProxG(L.PROX_L1, lam=0.01)
x0 = torch.full((d,), 0.01, dtype=dt, device=dev)
av, z = torch.empty_like(x0), torch.empty_like(x0)
def best_of(fn, n=3):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
out = []
t = best_of(lambda: ctx.full_gradient(f, x0, av))
name = ctx.last_kernel()
out.append(f"sweep {t * 1e3:.2f} ms {N * d * s / t / 1e12:.2f} TB/s")
if os.environ.get("CIAO_TABLE"):
    table = torch.empty((N, d), dtype=dt, device=dev)
    t = best_of(lambda: ctx.saga_init(f, g, 1e-3, x0, table, av, z))
    out.append(f"saga_init {t * 1e3:.2f} ms {2 * N * d * s / t / 1e12:.2f} TB/s (R+W)")
    gam = torch.full((N,), 0.5, dtype=dt, device=dev)
    hg = ctx.hat_gamma(gam)
    ctx.finito_init(f, g, gam, hg, x0, table, av, z)
    ctx.set_option("chain_max_batch", 0)
    for r in (256, 4096):
        if r > N:
            continue
        nb = min(8, N // r)
        first = np.arange(nb, dtype=np.int64) * r
        length = np.full(nb, r, np.int64)
        t = best_of(lambda: ctx.finito_steps_blocks(f, g, gam, hg, first, length, table, av, z))
        out.append(f"finito r={r} {t / nb * 1e6:.1f} us/batch {3 * nb * r * d * s / t / 1e12:.2f} TB/s (2R+W) [{ctx.last_kernel().split(' ')[0]}]")
print(f"{'f32' if s == 4 else 'f64'} d={d} N={N} [{os.environ.get('CIAO_OPTS', '')}] {name} | " + " | ".join(out), flush=True)
