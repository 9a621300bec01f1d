#!/usr/bin/env python3
"""Times the SVRG inner chain (d = 1024 f64 and f32) on cuda:0; prints us per update.
CIAO_SHARDED=1: over a two-slice shard table of the same matrix (chain_dma_kernel<..., SHARDED>); every figure is followed by a
digest of the resulting state (w, z), so that two libraries can be compared bitwise on one box (tools/spill_ab.sh)."""
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
def digest(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:10]
for dt in (torch.float64, torch.float32):
    N, d, m = 200_000, int(os.environ.get("CIAO_D", "1024")), 400_000
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(N))
    ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    ctx.svrg_init(F, x0, av, z, zf, w)
    idx = ctx._idx(IndexStream(0).rand_indices(N, m))
    if os.environ.get("CIAO_SHARDED"):
        cut = (N // 3) // 64 * 64
        t = L.ShardTable()
        t.nshards, t.owner = 2, 1
        for k, r0 in enumerate((0, cut)):
            t.row0[k] = r0
            t.A[k], t.b[k], t.table[k] = A[r0:].data_ptr(), b[r0:].data_ptr(), 0
        t.row0[2] = N
        ctx.set_shards(t)
    ctx.svrg_inner(F, g, 1e-7, idx[:2000], av, z, zf, w); ctx.synchronize()
    t0 = time.perf_counter(); ctx.svrg_inner(F, g, 1e-7, idx, av, z, zf, w); ctx.synchronize()
    t2 = (time.perf_counter() - t0) / m * 1e6
    k2, d2 = ctx.last_kernel().split(" grid")[0], digest(w, z)
    if os.environ.get("CIAO_SHARDED"):
        ctx.set_shards(None)
        out.append(f"{'f64' if dt == torch.float64 else 'f32'} sharded two dots {t2:.3f} us/update [{d2}] ({k2})")
        continue
    # the same with the row dots of the last full pass reused (one dot product per step): a whole outer iteration, so the
    # sweep over the N rows is in the time (N / m of a sweep per update: about 1 ns here)
    ctx.svrg_init(F, x0, av, z, zf, w)
    ctx.svrg_iterate(F, g, 1e-7, idx[:2000], False, av, z, zf, w, reuse_rowdots=True); ctx.synchronize()
    t0 = time.perf_counter(); ctx.svrg_iterate(F, g, 1e-7, idx, False, av, z, zf, w, reuse_rowdots=True); ctx.synchronize()
    t1 = (time.perf_counter() - t0) / m * 1e6
    out.append(f"{'f64' if dt == torch.float64 else 'f32'} two dots {t2:.3f} [{d2}] one dot {t1:.3f} [{digest(w, z)}] us/update")
print(" | ".join(out))
