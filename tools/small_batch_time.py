#!/usr/bin/env python3
"""Finito batches and LFinito's batch sweep on rows of tabular size (d = 50, 100, 255; fp64 and fp32): the default kernels (round 4:
rows_smallm_kernel for dense row blocks where its tiles fit LDS, rows_smallb_kernel for index lists and the rest) against the scalar
generic kernel they used to run (option force_generic=1), index lists and row blocks.
TB/s of algorithmic bytes: Finito batch 3*d*s + 2*s + 8 per sample, LFinito batch sweep d*s + 2*s + 8."""
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
ctx.set_option("chain_max_batch", 0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
N = int(os.environ.get("CIAO_N", "4000000"))
st = IndexStream(0)
DS = tuple(int(x) for x in os.environ.get("CIAO_DS", "50,100,255").split(","))           # CIAO_DS=50 CIAO_DT=f64: one shape (profiling)
DTS = {"f64": (torch.float64,), "f32": (torch.float32,)}.get(os.environ.get("CIAO_DT", ""), (torch.float64, torch.float32))
for d in DS:
    for dt in DTS:
        es = 8 if dt == torch.float64 else 4
        A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
        ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
        F = PackedF(L.LOSS_LS, A, b, float(N))
        ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
        g = ProxG(L.PROX_L1, lam=1e-3)
        gam = torch.full((N,), 0.999 / 1.3, dtype=dt, device="cuda")
        hg = ctx.hat_gamma(gam)
        x0 = torch.zeros(d, dtype=dt, device="cuda")
        table = torch.empty((N, d), dtype=dt, device="cuda")
        av, z, zf = (torch.empty_like(x0) for _ in range(3))
        ctx.finito_init(F, g, gam, hg, x0, table, av, z)
        for r in (() if os.environ.get("CIAO_LFINITO_ONLY") else tuple(int(x) for x in os.environ.get("CIAO_RS", "4096,65536").split(","))):
            nit = 16 if r <= 65536 else 4
            idx = ctx._idx(np.concatenate([st.sample_without_replacement(N, r) for _ in range(nit)]))
            bptr = np.arange(nit + 1, dtype=np.int64) * r
            first = (np.arange(1, nit + 1, dtype=np.int64) % (N // r)) * r
            ln = np.full(nit, r, np.int64)
            row = [f"d={d:3d} {'f64' if es == 8 else 'f32'} r={r:5d}:"]
            for generic in (0, 1):
                ctx.set_option("force_generic", generic)
                for what, fn in (("lists", lambda: ctx.finito_steps(F, g, gam, hg, bptr, idx, table, av, z)),
                                 ("blocks", lambda: ctx.finito_steps_blocks(F, g, gam, hg, first, ln, table, av, z))):
                    fn(); ctx.synchronize()
                    t0 = time.perf_counter(); fn(); ctx.synchronize()
                    t = time.perf_counter() - t0
                    kname = ctx.last_kernel().split("<")[0].replace("rows_", "").replace("_kernel", "")
                    row.append(f"{kname:7s} finito {what:6s} {nit * r * (3 * d * es + 2 * es + 8) / t / 1e12:5.2f} TB/s ({t / nit * 1e6:7.1f} us/batch)")
                kern = ctx.last_kernel().split(" grid")[0]
            ctx.set_option("force_generic", 0)
            print(" | ".join(row) + f"  [{kern}]", flush=True)
        # LFinito: one iteration = full pass + batch sweep over all rows in static batches of 65536
        r = int(os.environ.get("CIAO_LFINITO_R", "65536"))   # (CIAO_LFINITO_R=4096: the mid-size batches)
        nb = N // r
        first = np.arange(nb, dtype=np.int64) * r
        ln = np.full(nb, r, np.int64)
        ctx.lfinito_init(F, hg, x0, av, z, zf)
        for tag, opt, val in (("matrix cores", "small_mfma", -1), ("smallb      ", "small_mfma", 0), ("generic     ", "force_generic", 1)):
            ctx.set_option(opt, val)
            ctx.lfinito_iterate_blocks(F, g, gam, hg, first, ln, av, z, zf); ctx.synchronize()
            t0 = time.perf_counter(); ctx.lfinito_iterate_blocks(F, g, gam, hg, first, ln, av, z, zf); ctx.synchronize()
            t = time.perf_counter() - t0
            print(f"d={d:3d} {'f64' if es == 8 else 'f32'} lfinito iteration (full pass + {nb} batches of {r}) {tag}: "
                  f"{t * 1e3:7.2f} ms = {2 * nb * r * (d * es + 2 * es) / t / 1e12:5.2f} TB/s  [{ctx.last_kernel().split(' grid')[0]}]", flush=True)
            ctx.set_option("small_mfma", -1)
            ctx.set_option("force_generic", 0)
        del A, b, table, F
        torch.cuda.empty_cache()
