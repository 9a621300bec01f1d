#!/usr/bin/env python3
"""Times SAGA steps alone (no init, indices already on the device) at several N; d = 1024 fp32 logistic (config C3)."""
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
d = int(os.environ.get("CIAO_D", "1024"))
for N in (100_000, 1_000_000, 10_000_000):
    A = torch.empty((N, d), dtype=torch.float32, device="cuda"); y = torch.empty((N,), dtype=torch.float32, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LOGISTIC, A, y, 1.0)
    ctx.synth_targets(F, torch.ones(d, dtype=torch.float32, device="cuda"), 0.1, True, 1, y)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    x0 = torch.ones(d, dtype=torch.float32, device="cuda")
    table = torch.empty((N, d), dtype=torch.float32, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, 1.0, x0, table, av, z)
    m = 500_000
    idx = ctx._idx(IndexStream(0).rand_indices(N, m))
    ctx.saga_steps(F, g, 1.0, False, idx[:2000], table, av, z); ctx.synchronize()
    t0 = time.perf_counter(); ctx.saga_steps(F, g, 1.0, False, idx, table, av, z); ctx.synchronize()
    t = time.perf_counter() - t0
    # sequential (cache/TLB friendly) indices for comparison
    seq = ctx._idx(np.arange(m, dtype=np.int64) % N)
    t0 = time.perf_counter(); ctx.saga_steps(F, g, 1.0, False, seq, table, av, z); ctx.synchronize()
    t2 = time.perf_counter() - t0
    print(f"N={N}: random {t / m * 1e6:.3f} us/update, sequential {t2 / m * 1e6:.3f} us/update  ({ctx.last_kernel()})", flush=True)
    del A, y, table, F
    torch.cuda.empty_cache()
