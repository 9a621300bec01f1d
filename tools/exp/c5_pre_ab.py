#!/usr/bin/env python3
"""Short Finito batches (static row blocks, d = 4096 fp32 -- BASELINE config #5's shape): one row at a time per workgroup
(split_pre=1, round 3) against all of a workgroup's rows in flight at once (split_pre automatic, round 4), same box, and the
results of the two compared BITWISE (z, av, every table row).  us per batch over CIAO_NB batches, rows kernel by HIP events.
The option split_pre and the kernel's pre-issue loop were an experiment of round 4 (neutral, profiles/r04_c5_pre_ab.txt) and are
not in the product library; without them this script times the product kernel twice."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
torch.cuda.set_device(0)
ctx = Context(0)
N, d = 200_000, int(os.environ.get("CIAO_D", "4096"))
dt = torch.float64 if os.environ.get("CIAO_F64") else torch.float32
NB = int(os.environ.get("CIAO_NB", "1000"))
A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(L.LOSS_LS, A, b, float(N))
ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
g = ProxG(L.PROX_L1, lam=1e-3)
gam = (0.999 / 1.3 * (1.0 + 0.1 * torch.frac(torch.arange(N, device="cuda", dtype=torch.float64) * 0.6180339887498949))).to(dt)
hg = ctx.hat_gamma(gam)
x0 = torch.zeros(d, dtype=dt, device="cuda")
table = torch.empty((N, d), dtype=dt, device="cuda")
av, z = torch.empty_like(x0), torch.empty_like(x0)
rs = [int(a) for a in sys.argv[1:]] or [256, 384, 512, 768, 1024, 2048, 4096]
es = A.element_size()
for r in rs:
    nb = min(NB, 4 * (N // r))
    first = (np.arange(1, nb + 1, dtype=np.int64) % (N // r)) * r
    ln = np.full(nb, r, np.int64)
    res = {}
    for pre in (1, -1):
        try:
            ctx.set_option("split_pre", pre)
        except Exception:
            pass
        ctx.finito_init(F, g, gam, hg, x0, table, av, z)
        ctx.finito_steps_blocks(F, g, gam, hg, first[:10], ln[:10], table, av, z); ctx.synchronize()
        t0 = time.perf_counter(); ctx.finito_steps_blocks(F, g, gam, hg, first, ln, table, av, z); ctx.synchronize()
        t = time.perf_counter() - t0
        kern = ctx.last_kernel()
        state = (z.clone(), av.clone(), table[: min(N, nb * r)].clone())
        ctx.timing_enable(True); ctx.timing_read()
        ctx.finito_steps_blocks(F, g, gam, hg, first[:100], ln[:100], table, av, z); ctx.synchronize()
        k_ms, k_n = ctx.timing_read(); ctx.timing_enable(False)
        res[pre] = (t / nb * 1e6, k_ms / max(k_n, 1) * 1e3, kern, state)
    same = all(torch.equal(u, v) for u, v in zip(res[1][3], res[-1][3]))
    gbs = lambda us: r * (3 * d * es + es + 8) / us / 1e3
    print(f"r={r:5d}: one row at a time {res[1][0]:6.2f} us/batch (rows kernel {res[1][1]:5.2f}) = {gbs(res[1][0]):5.0f} GB/s | "
          f"all rows in flight {res[-1][0]:6.2f} us/batch (rows kernel {res[-1][1]:5.2f}) = {gbs(res[-1][0]):5.0f} GB/s | bitwise equal: {same} "
          f"[{res[-1][2].split('<')[1]}]", flush=True)
    del res
