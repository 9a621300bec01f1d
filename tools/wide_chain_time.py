#!/usr/bin/env python3
"""SVRG inner cycle and SAGA steps on rows beyond 8192 elements: the several-workgroup chain (chain_wide_kernel) against the
one-workgroup any-length kernel (option chain_no_wide=1), us per update; CIAO_DS = comma-separated row lengths."""
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
for d in [int(x) for x in os.environ.get("CIAO_DS", "9000,16384,32768,65536,131072").split(",")]:
    for dt in (torch.float64, torch.float32):
        N = max(64, min(20000, (2 << 30) // (d * 8)))
        A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
        ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
        F = PackedF(L.LOSS_LS, A, b, float(N))
        ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
        g = ProxG(L.PROX_L1, lam=1e-3)
        x0 = torch.zeros(d, dtype=dt, device="cuda")
        av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
        table = torch.empty((N, d), dtype=dt, device="cuda")
        row = [f"d={d:6d} {'f64' if dt == torch.float64 else 'f32'} N={N}:"]
        for no_wide in (0, 1):
            ctx.set_option("chain_no_wide", no_wide)
            m = 20000 if not no_wide else 2000
            idx = ctx._idx(IndexStream(0).rand_indices(N, m))
            ctx.svrg_init(F, x0, av, z, zf, w)
            ctx.svrg_inner(F, g, 1e-7, idx[:200], av, z, zf, w); ctx.synchronize()
            t0 = time.perf_counter(); ctx.svrg_inner(F, g, 1e-7, idx, av, z, zf, w); ctx.synchronize()
            ts = (time.perf_counter() - t0) / m * 1e6
            ks = ctx.last_kernel().split(" block")[0]
            ctx.saga_init(F, g, 1e-7, x0, table, av, z)
            ctx.saga_steps(F, g, 1e-7, False, idx[:200], table, av, z); ctx.synchronize()
            t0 = time.perf_counter(); ctx.saga_steps(F, g, 1e-7, False, idx, table, av, z); ctx.synchronize()
            tg = (time.perf_counter() - t0) / m * 1e6
            row.append(f"{'one workgroup' if no_wide else 'several     '} svrg {ts:7.2f} saga {tg:7.2f} us/update [{ks}]")
        ctx.set_option("chain_no_wide", 0)
        print(" | ".join(row), flush=True)
        del A, table, F
        torch.cuda.empty_cache()
