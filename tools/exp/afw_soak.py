#!/usr/bin/env python3
"""Long runs of adaptive Finito on the several-workgroup chain (afinito_wide_kernel): d = 131072 (64 workgroups) and d = 20001 (10,
the last one a partial slice), 300k steps over 2048 / 4096 samples, twice from the same state -- bitwise repeatable, no timeout word --
and us per step."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
for dt, N, d in ((torch.float64, 2048, 131072), (torch.float32, 2048, 131072), (torch.float64, 4096, 20001)):
    k = 300_000
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(N))
    xt = torch.from_numpy(np.random.default_rng(1).standard_normal(d) * (np.random.default_rng(2).random(d) < 0.05)).to("cuda", dt)
    ctx.synth_targets(F, xt, 0.1, False, 1, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    idx = ctx._idx(IndexStream(0).rand_indices(N, k))
    outs = []
    for rep in range(2):
        table = torch.empty((N, d), dtype=dt, device="cuda")
        meta = torch.empty((N, 4, 4), dtype=dt, device="cuda")
        hg = torch.empty(1, dtype=dt, device="cuda")
        av, z = torch.empty_like(x0), torch.empty_like(x0)
        ctx.afinito_init(F, g, 0.999, x0, table, meta, av, z, hg)
        ctx.synchronize()
        t0 = time.perf_counter(); done, trials = ctx.afinito_steps(F, g, 0.999, 1e-9, idx, table, meta, av, z, hg); ctx.synchronize()
        t = time.perf_counter() - t0
        outs.append((done, trials, z.clone(), av.clone(), hg.clone(), meta.clone(), table[:64].clone()))
        kern = ctx.last_kernel()
        del table
    same = outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and all(torch.equal(u, v) for u, v in zip(outs[0][2:], outs[1][2:]))
    print(f"{'f64' if dt == torch.float64 else 'f32'} N={N} d={d}: {t / max(done, 1) * 1e6:.3f} us/step, {trials / max(done, 1):.3f} trials/step, "
          f"done {done}/{k}, finite {bool(torch.isfinite(outs[0][2]).all())}, bitwise repeatable: {same}  [{kern}]", flush=True)
    del A, outs
    torch.cuda.empty_cache()
