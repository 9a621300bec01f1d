import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ciao_loader; ciao = ciao_loader.load()
import problems as P
from oracle import oracle as O
import test_gpu_parity as TP
from ciaoalgorithms_jl_amd.device import Context
ctx = Context()
N, d, dtype = 200, 2048, np.float64
A, b, x0 = P.synthetic("ls", N, d, dtype, seed=21)
op, dp = TP.make("ls", A, b, float(N), dtype)
og, dg = TP.make_g("l1", dtype, d, lam=0.02)
st = ciao.IndexStream(4)
idx = st.rand_indices(N, 3 * N)
idx[5:8] = idx[5]; idx[10] = idx[8]; idx[20:30:2] = idx[20]; idx[21:31:2] = idx[21]
def run(no_dma, nsteps):
    table = torch.empty((N, d), dtype=torch.float64, device="cuda")
    meta4 = torch.empty((N, 4, 4), dtype=torch.float64, device="cuda")
    hg = torch.empty(1, dtype=torch.float64, device="cuda")
    av, z = torch.empty(d, dtype=torch.float64, device="cuda"), torch.empty(d, dtype=torch.float64, device="cuda")
    ctx.set_option("chain_no_dma", no_dma)
    ctx.afinito_init(dp, dg, 0.999, TP.dev(x0), table, meta4, av, z, hg)
    done, trials = ctx.afinito_steps(dp, dg, 0.999, 1e-9, idx[:nsteps], table, meta4, av, z, hg)
    ctx.set_option("chain_no_dma", 0)
    return trials, z.cpu().numpy(), meta4[:, 0, 2].cpu().numpy(), ctx.last_kernel()
def orc(nsteps):
    rt, rg, rgam, rfi, rav, rz, rhg = O.afinito_init(op, og, dtype(0.999), x0)
    rdone, rhg, rtrials = O.afinito_steps(op, og, dtype(0.999), dtype(1e-9), idx[:nsteps], rt, rg, rgam, rfi, rhg, rav, rz)
    return rtrials, rz, rgam
lo, hi = 0, len(idx)
for n in (len(idx), 300, 150, 75, 40, 20, 10):
    t1, z1, g1, k1 = run(0, n); t0, z0, g0, k0 = run(1, n); tr, zr, gr = orc(n)
    print(n, "trials dma/reg/oracle", t1, t0, tr, "max|z-zr| dma", np.abs(z1 - zr).max(), "reg", np.abs(z0 - zr).max(),
          "gamma ratio dma", np.unique(np.round(g1 / gr, 6)), "reg", np.unique(np.round(g0 / gr, 6)))
