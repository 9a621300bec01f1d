#!/usr/bin/env python3
"""Times the adaptive Finito chain on cuda:0 for a few (dtype, N, d); prints us per step and trials per step."""
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
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
out = []
cases = [(torch.float64, 200_000, 1024), (torch.float64, 2_000, 1024), (torch.float32, 200_000, 1024), (torch.float64, 100_000, 4096)]
if os.environ.get("CIAO_D"):
    cases = [(torch.float64, 200_000, int(os.environ["CIAO_D"])), (torch.float32, 200_000, int(os.environ["CIAO_D"]))]
for dt, N, d in cases:
    k = 100_000
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(N))
    xt = torch.from_numpy(np.random.default_rng(1).standard_normal(d) * (np.random.default_rng(2).random(d) < 0.05)).to("cuda", dt)
    ctx.synth_targets(F, xt, 0.1, False, 1, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    table = torch.empty((N, d), dtype=dt, device="cuda")
    meta = torch.empty((N, 4, 4), dtype=dt, device="cuda")
    hg = torch.empty(1, dtype=dt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.afinito_init(F, g, 0.999, x0, table, meta, av, z, hg)
    idx = ctx._idx(IndexStream(0).rand_indices(N, k))
    ctx.afinito_steps(F, g, 0.999, 1e-9, idx[:2000], table, meta, av, z, hg)
    t0 = time.perf_counter(); done, trials = ctx.afinito_steps(F, g, 0.999, 1e-9, idx, table, meta, av, z, hg)
    t = time.perf_counter() - t0
    hh = hashlib.sha1()
    for tt in (z, av, hg, table[idx[:2048]]):
        hh.update(tt.detach().cpu().numpy().tobytes())
    out.append(f"{'f64' if dt == torch.float64 else 'f32'} N={N} d={d}: {t / max(done, 1) * 1e6:.3f} us/step, {trials / max(done, 1):.3f} trials/step [{hh.hexdigest()[:10]}]")
    del A, table
print(" | ".join(out))
