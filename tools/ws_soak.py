#!/usr/bin/env python3
"""Soak of chain_ws_kernel: one SAGA launch of CIAO_SOAK_STEPS (default 10^8) steps at N = 2M, d = 1024 fp32 -- no watchdog trip,
rank-1 table rows, finite iterate (the incrementally maintained av drifts from mean(table) in fp32 over 10^8 updates, in the
reference too: reported, not asserted) -- and then 2 x 10^7 steps on chain_ws_kernel AND on chain_dma_kernel from the same state:
z, av and the whole 8 GB table must be BITWISE equal."""
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
steps = int(float(os.environ.get("CIAO_SOAK_STEPS", "1e8")))
N, d = 2_000_000, 1024
A = torch.empty((N, d), dtype=torch.float32, device="cuda"); y = torch.empty((N,), dtype=torch.float32, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(L.LOSS_LOGISTIC, A, y, 1.0)
ctx.synth_targets(F, torch.ones(d, dtype=torch.float32, device="cuda"), 0.1, True, 1, y)
g = ProxG(L.PROX_L1, lam=1.0 / N)
x0 = torch.ones(d, dtype=torch.float32, device="cuda")
table = torch.empty((N, d), dtype=torch.float32, device="cuda")
av, z = torch.empty_like(x0), torch.empty_like(x0)
ctx.saga_init(F, g, 1.0, x0, table, av, z)
idx, _ = IndexStream(0).rand_indices_device(ctx, N, steps)
t0 = time.perf_counter()
ctx.saga_steps(F, g, 1.0, False, idx, table, av, z)
ctx.synchronize()                       # raises if the kernel's spin watchdog or an index check tripped
t = time.perf_counter() - t0
print(f"{steps} steps in {t:.1f} s = {t / steps * 1e6:.4f} us per update [{ctx.last_kernel()}]", flush=True)
mean = torch.zeros(d, dtype=torch.float64, device="cuda")
for lo in range(0, N, 250_000):
    mean += table[lo:lo + 250_000].double().sum(dim=0)
mean /= N
err = float((av.double() - mean).abs().max() / mean.abs().max())
rows = torch.randint(0, N, (4000,), device="cuda")
t_, a_ = table[rows].double(), A[rows].double()
cos = (t_ * a_).sum(dim=1).abs() / (t_.norm(dim=1) * a_.norm(dim=1) + 1e-300)
print(f"av vs mean(table): rel err {err:.2e}; rank-1 rows: min |cos| {float(cos.min()):.9f}; z finite: {bool(torch.isfinite(z).all())}")
assert float(cos.min()) > 1 - 1e-5 and bool(torch.isfinite(z).all())
m2 = 20_000_000
idx2 = idx[:m2]
res = []
for no_ws in (0, 1):
    ctx.set_option("chain_no_ws", no_ws)
    t2 = torch.empty_like(table) if no_ws else table
    a2, z2 = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, 1.0, x0, t2, a2, z2)
    ctx.saga_steps(F, g, 1.0, False, idx2, t2, a2, z2)
    ctx.synchronize()
    res.append((t2, a2, z2, ctx.last_kernel().split(" ")[0]))
ctx.set_option("chain_no_ws", 0)
same = torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and all(
    torch.equal(res[0][0][lo:lo + 250_000], res[1][0][lo:lo + 250_000]) for lo in range(0, N, 250_000))
print(f"{m2} steps: {res[0][3]} vs {res[1][3]}: {'BITWISE equal (z, av, table)' if same else 'DIFFERENT'}")
assert same
print("soak ok")
