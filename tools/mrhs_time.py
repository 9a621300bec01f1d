#!/usr/bin/env python3
"""The full pass for K iterates in ONE pass over A (ciao_full_gradient_multi, csrc/mrhs_kernels.h) against K single sweeps,
at the metric's own size (N = CIAO_N, default 10M; d = 1024; fp64 unless CIAO_F32): seconds per call, TFLOP/s of the 4 N d K flops,
and what K single sweeps of the same rows cost on the same box."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF
torch.cuda.set_device(0)
ctx = Context(0)
N = int(os.environ.get("CIAO_N", "10000000")); d = int(os.environ.get("CIAO_D", "1024"))
dt = torch.float32 if os.environ.get("CIAO_F32") else torch.float64
loss = L.LOSS_LOGISTIC if os.environ.get("CIAO_LOGISTIC") else L.LOSS_LS
A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(loss, A, b, float(N) if loss == L.LOSS_LS else 1.0)
ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, loss == L.LOSS_LOGISTIC, 1, b)
x1 = torch.zeros(d, dtype=dt, device="cuda"); a1 = torch.empty_like(x1)
ctx.full_gradient(F, x1, a1); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): ctx.full_gradient(F, x1, a1)
ctx.synchronize()
t_sweep = (time.perf_counter() - t0) / 5
print(f"{'f32' if dt == torch.float32 else 'f64'} N={N} d={d}: one single sweep {t_sweep * 1e3:.2f} ms  [{ctx.last_kernel().split(' grid')[0]}]", flush=True)
for K in [int(k) for k in os.environ.get("CIAO_KS", "16,64,256").split(",")]:
    xs = [torch.randn(d, dtype=dt, device="cuda") * 0.1 for _ in range(K)]
    avs = [torch.empty_like(x) for x in xs]
    ctx.full_gradient_multi(F, xs, avs); ctx.synchronize()
    reps = 3 if K <= 64 else 2
    t0 = time.perf_counter()
    for _ in range(reps): ctx.full_gradient_multi(F, xs, avs)
    ctx.synchronize()
    t = (time.perf_counter() - t0) / reps
    err = 0.0
    for k in (0, K - 1):
        ctx.full_gradient(F, xs[k], a1)
        err = max(err, float((avs[k] - a1).abs().max() / a1.abs().max()))
    print(f"K={K:4d}: one pass {t * 1e3:8.2f} ms = {4.0 * N * d * K / t / 1e12:6.1f} TFLOP/s; K single sweeps {K * t_sweep * 1e3:8.1f} ms ({K * t_sweep / t:5.1f}x); "
          f"max rel. difference from the single sweep {err:.1e}  [{ctx.last_kernel()}]", flush=True)
