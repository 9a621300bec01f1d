#!/usr/bin/env python3
"""Times Finito batch steps (d = 4096 f32 unless CIAO_D / CIAO_F64 say otherwise, N = 250k) for the batch sizes given on the command
line; prints us per batch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
own = torch.cuda.Stream() if os.environ.get("CIAO_OWN_STREAM") else None   # a capturable (non-default) stream
ctx = Context(0, stream=own)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
rs = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]
N = 250_000
d = int(os.environ.get("CIAO_D", "4096"))
dt = torch.float64 if os.environ.get("CIAO_F64") else torch.float32
A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(L.LOSS_LS, A, b, float(N))
ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
g = ProxG(L.PROX_L1, lam=1e-3)
gam = torch.full((N,), 0.999 * N / (1.3 * N), dtype=dt, device="cuda")
hg = ctx.hat_gamma(gam)
x0 = torch.zeros(d, dtype=dt, device="cuda")
table = torch.empty((N, d), dtype=dt, device="cuda")
av, z = torch.empty_like(x0), torch.empty_like(x0)
torch.cuda.synchronize()
ctx.finito_init(F, g, gam, hg, x0, table, av, z)
st = IndexStream(0)
out = []
for r in rs:
    nit = max(8, min(2000, (1 << 20) // r))
    bidx = ctx._idx(np.concatenate([st.sample_without_replacement(N, r) for _ in range(nit)]))
    bptr = np.arange(nit + 1, dtype=np.int64) * r
    ctx.finito_steps(F, g, gam, hg, bptr[:3], bidx[:2 * r], table, av, z); ctx.synchronize()
    t0 = time.perf_counter(); ctx.finito_steps(F, g, gam, hg, bptr, bidx, table, av, z); ctx.synchronize()
    t = time.perf_counter() - t0
    out.append(f"r={r}: {t / nit * 1e6:.2f} us/batch, {nit * r * (3 * d * A.element_size() + 16) / t / 1e9:.0f} GB/s [{ctx.last_kernel()}]")
    print(out[-1], flush=True)
if os.environ.get("CIAO_BLOCKS"):   # static contiguous batches: index lists vs the index-free *_blocks entry point
    for r in rs:
        nit = max(8, min(2000, N // r))
        first = (np.random.default_rng(0).permutation(N // r)[:nit] * r).astype(np.int64)
        length = np.full(nit, r, np.int64)
        bptr = np.arange(nit + 1, dtype=np.int64) * r
        bidx = ctx._idx((first[:, None] + np.arange(r)[None, :]).reshape(-1))
        for name, fn in (("index lists", lambda: ctx.finito_steps(F, g, gam, hg, bptr, bidx, table, av, z)),
                         ("row blocks ", lambda: ctx.finito_steps_blocks(F, g, gam, hg, first, length, table, av, z))):
            fn(); ctx.synchronize()
            t0 = time.perf_counter(); fn(); ctx.synchronize()
            t = time.perf_counter() - t0
            out.append(f"static r={r} {name}: {t / nit * 1e6:.2f} us/batch [{ctx.last_kernel()}]")
            print(out[-1], flush=True)
if os.environ.get("CIAO_LFINITO"):
    zf = torch.empty_like(x0)
    ctx.lfinito_init(F, hg, x0, av, z, zf)
    for r in rs:
        nb = N // r
        bidx = ctx._idx(np.arange(nb * r, dtype=np.int64))
        bptr = np.arange(nb + 1, dtype=np.int64) * r
        ctx.lfinito_iterate(F, g, gam, hg, bptr[:3], bidx[:2 * r], av, z, zf); ctx.synchronize()
        t0 = time.perf_counter(); ctx.lfinito_iterate(F, g, gam, hg, bptr, bidx, av, z, zf); ctx.synchronize()
        t = time.perf_counter() - t0
        out.append(f"LFinito r={r}: {t * 1e3:.2f} ms per iteration ({nb} batches + full sweep), {2 * N * d * 4 / t / 1e9:.0f} GB/s [{ctx.last_kernel()}]")
print("\n".join(out))
