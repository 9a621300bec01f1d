#!/usr/bin/env python3
"""Where a step of chain_ws_kernel spends its cycles: a CIAO_WS_DBG=1 experiment build (tools/exp_build.sh) sums, per consumer
wave, the cycles from arrival to a successful poll (exchange) and from there to the next arrival (compute), and counts poll
retries / landed spins; per issuer wave, how often it ran into the consumers and what issuing a step's DMA costs."""
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
ctx = Context(0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for dt in (torch.float32,):
    N, d, m = 2_000_000, int(os.environ.get("CIAO_D", "1024")), 200_000
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LOGISTIC, A, b, 1.0)
    ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, True, 1, b)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    x0 = torch.ones(d, dtype=dt, device="cuda")
    table = torch.empty((N, d), dtype=dt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, 1.0, x0, table, av, z)
    idx = ctx._idx(IndexStream(0).rand_indices(N, m))
    dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
    ctx.set_option("chain_dbg_ptr", dbg.data_ptr())
    ctx.saga_steps(F, g, 1.0, False, idx[:2000], table, av, z); ctx.synchronize()
    dbg.zero_()
    t0 = time.perf_counter(); ctx.saga_steps(F, g, 1.0, False, idx, table, av, z); ctx.synchronize()
    us = (time.perf_counter() - t0) / m * 1e6
    v = dbg.cpu().numpy().reshape(8, 8)
    print(f"{'f64' if dt == torch.float64 else 'f32'} d={d}: {us:.3f} us/update with stamps; {ctx.last_kernel()}")
    st = np.maximum(v[:4, 4], 1)
    print(f"   consumers: exchange cycles/step {np.round(v[:4, 0] / st, 1).tolist()}  compute {np.round(v[:4, 1] / st, 1).tolist()}  "
          f"poll retries/step {np.round(v[:4, 2] / st, 3).tolist()}  landed spins {v[:4, 3].tolist()}")
    for q in range(2):
        r = v[5 + q]
        print(f"   issuer {q}: blocked {r[0]} times ({r[1]} spins), lifetime {r[2]} cycles = {r[2] / m:.1f}/step, issue {r[3] / m:.1f} cycles/step")
