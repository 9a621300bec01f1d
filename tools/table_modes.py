#!/usr/bin/env python3
"""The read+write table modes of the rows kernels at scale (SAGA init, Finito init, Finito batch r = 65536, ProShI init and
batch): HIP-event GB/s of each, a few launches apiece.  Run plain for the rates, or under `rocprofv3 --pmc X --kernel-trace`
for the per-kernel counters.  TABLE_REPS launches per mode (default 4); TABLE_ONLY=mode,... restricts."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, PackedSepQuad, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
reps = int(os.environ.get("TABLE_REPS", "4"))
only = [s for s in os.environ.get("TABLE_ONLY", "").split(",") if s]
res = {}


def timed(name, fn, nbytes):
    if only and name not in only:
        return
    fn()
    ctx.timing_enable(True); ctx.timing_read()
    for _ in range(reps):
        fn()
    ms, n = ctx.timing_read(); ctx.timing_enable(False)
    res[name] = {"kernel_ms": ms / n, "alg_GBps": nbytes / (ms / n * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (ms / n * 1e-3) / 8e12,
                 "kernel": ctx.last_kernel()}
    print(json.dumps({name: res[name]}), flush=True)


def lasso(N, d, dt, seed=1):
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, seed, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(N))
    ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, seed, b)
    return F


for tag, dt, d, N in (("f32_d1024", torch.float32, 1024, 4_000_000), ("f32_d4096", torch.float32, 4096, 1_000_000),
                      ("f64_d1024", torch.float64, 1024, 2_000_000)):
    es = 8 if dt == torch.float64 else 4
    F = lasso(N, d, dt)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    table = torch.empty((N, d), dtype=dt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    gam = torch.full((N,), 0.7, dtype=dt, device="cuda")
    hg = ctx.hat_gamma(gam)
    timed(f"saga_init_{tag}", lambda: ctx.saga_init(F, g, 0.5, x0, table, av, z), 2 * N * d * es)
    timed(f"finito_init_{tag}", lambda: ctx.finito_init(F, g, gam, hg, x0, table, av, z), 2 * N * d * es)
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    for r in (65536, 4096):
        bidx = ctx._idx(IndexStream(0).sample_without_replacement(N, r))
        bptr = np.array([0, r], np.int64)
        timed(f"finito_batch_r{r}_{tag}", lambda: ctx.finito_steps(F, g, gam, hg, bptr, bidx, table, av, z), r * (3 * d * es + 2 * es + 8))
    del F, table
    torch.cuda.empty_cache()

N, d = 1_000_000, 1024
Q = torch.empty((N, d), dtype=torch.float64, device="cuda"); q = torch.empty_like(Q)
ctx.synth_normal(Q, 0, 7, 1.0); ctx.synth_normal(q, 0, 8, 1.0); Q.abs_()
f = PackedSepQuad(Q, q, eta=30.0, lo=-2.0, hi=2.0)
gbox = ProxG(L.PROX_BOX, lo=-float("inf"), hi=1.0)
gam = torch.full((N,), 0.999 * N / 40.0, dtype=torch.float64, device="cuda")
x0 = torch.zeros(d, dtype=torch.float64, device="cuda")
table = torch.empty((N, d), dtype=torch.float64, device="cuda")
av, z = torch.empty_like(x0), torch.empty_like(x0)
hgd = torch.empty(1, dtype=torch.float64, device="cuda")
timed("proshi_init_f64_d1024", lambda: ctx.proshi_init(f, gbox, gam, x0, table, av, z, hgd), N * 3 * d * 8)
hg = float(hgd.item())
for r in (65536, 4096):
    bidx = ctx._idx(IndexStream(0).sample_without_replacement(N, r))
    bptr = np.array([0, r], np.int64)
    timed(f"proshi_batch_r{r}_f64_d1024", lambda: ctx.proshi_steps(f, gbox, gam, hg, bptr, bidx, table, av, z), r * 4 * d * 8)
ctx.synchronize()
print(json.dumps(res))
