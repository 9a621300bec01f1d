#!/usr/bin/env python3
"""Times ProShI agent batches (ProShI_basic.jl:109-121: 3 reads and 1 write of a d-vector per agent) of r agents on cuda:0,
r = 256 ... 65536, d and type from the environment (CIAO_D, CIAO_F32); options through CIAO_OPTS=key=value,...
Prints us per batch and algorithmic TB/s (r * 4 * d * s bytes per batch)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedSepQuad, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
d = int(os.environ.get("CIAO_D", "1024"))
dt = torch.float32 if os.environ.get("CIAO_F32") else torch.float64
s = 4 if dt == torch.float32 else 8
N = int(os.environ.get("CIAO_N", "1000000"))
dev = torch.device("cuda", 0)
Q = torch.empty((N, d), dtype=dt, device=dev)
q = torch.empty((N, d), dtype=dt, device=dev)
ctx.synth_normal(Q, 0, seed=7, scale=1.0)
ctx.synth_normal(q, 0, seed=8, scale=1.0)
Q.abs_()
f = PackedSepQuad(Q, q, eta=30.0, lo=-2.0, hi=2.0)
gbox = ProxG(L.PROX_BOX, lo=-float("inf"), hi=1.0)
gam = torch.full((N,), 0.999 * N / 40.0, dtype=dt, device=dev)
x0 = torch.zeros(d, dtype=dt, device=dev)
table = torch.empty((N, d), dtype=dt, device=dev)
av, z = torch.empty_like(x0), torch.empty_like(x0)
hgd = torch.empty(1, dtype=dt, device=dev)
ctx.proshi_init(f, gbox, gam, x0, table, av, z, hgd)
hg = float(hgd.item())
st = IndexStream(3)
out = []
for r in (256, 512, 1024, 2048, 4096, 8192, 16384, 65536):
    nit = 32 if r <= 4096 else 8
    batches = [st.sample_without_replacement(N, r) for _ in range(nit)]
    bptr = np.arange(nit + 1, dtype=np.int64) * r
    bidx = ctx._idx(np.concatenate(batches))
    ctx.proshi_steps(f, gbox, gam, hg, bptr[:2], bidx[:r], table, av, z)
    ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.proshi_steps(f, gbox, gam, hg, bptr, bidx, table, av, z)
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    out.append(f"r={r}: {best / nit * 1e6:.1f} us/batch {nit * r * 4 * d * s / best / 1e12:.2f} TB/s")
print(f"{'f32' if s == 4 else 'f64'} d={d} [{os.environ.get('CIAO_OPTS', '')}] {ctx.last_kernel()} | " + " | ".join(out))
