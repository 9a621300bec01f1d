"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the CPU restatement --
see its PROVENANCE note): the oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _tol(dtype):
    return (1e-11, 1e-13) if dtype == np.float64 else (2e-5, 1e-7)


def _close(a, b, dtype, scale=1.0):
    rel, ab = _tol(dtype)
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() <= scale * (rel * max(np.abs(b).max(), 1e-30) + ab)


def test_fixtures_exist():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(path):
    from oracle import oracle as O
    G = np.load(path)
    dt = G["A"].dtype.type
    p = O.Problem(str(G["loss"]), G["A"], G["b"], float(G["lam_f"]))
    g = O.Prox("l1", lam=float(G["lam_g"]))
    N = p.N
    av, z, zf, w = O.svrg_init(p, G["x0"])
    assert _close(av, G["svrg_av0"], dt)
    for e in range(3):
        O.svrg_iterate(p, g, dt(G["svrg_gamma"]), G["svrg_idx"][e], False, av, z, zf, w)
    assert _close(zf, G["svrg_zfull"], dt) and _close(w, G["svrg_w"], dt) and _close(av, G["svrg_av"], dt)
    for key, sag in (("saga", False), ("sag", True)):
        table, av, z = O.saga_init(p, g, dt(G[f"{key}_gamma"]), G["x0"])
        assert _close(z, G[f"{key}_z0"], dt)
        O.saga_steps(p, g, dt(G[f"{key}_gamma"]), sag, G[f"{key}_idx"], table, av, z)
        assert _close(z, G[f"{key}_z"], dt) and _close(table, G[f"{key}_table"], dt)
    nb = -(-N // 2)
    static = [np.arange(2 * j, min(2 * j + 2, N), dtype=np.int64) for j in range(nb)]
    table, av, z, hg = O.finito_init(p, g, G["finito_gam"], G["x0"])
    assert _close(z, G["finito_z0"], dt)
    O.finito_steps(p, g, G["finito_gam"], hg, [static[(t + 1) % nb] for t in range(3 * nb)], table, av, z)
    assert _close(z, G["finito_z"], dt) and _close(table, G["finito_table"], dt)
    av, z, zf, hg = O.lfinito_init(p, G["finito_gam"], G["x0"])
    for _ in range(3):
        O.lfinito_iterate(p, g, G["finito_gam"], hg, static, av, z, zf)
    assert _close(z, G["lfinito_z"], dt) and _close(zf, G["lfinito_zfull"], dt)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_device_matches_golden(ctx, path):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    G = np.load(path)
    dt = G["A"].dtype.type
    N, d = G["A"].shape
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    F = PackedF(L.LOSS_LOGISTIC if str(G["loss"]) == "logistic" else L.LOSS_LS, dev(G["A"]), dev(G["b"]), float(G["lam_f"]))
    g = ProxG(L.PROX_L1, lam=float(G["lam_g"]))
    x0 = dev(G["x0"])
    new = lambda: torch.empty_like(x0)  # noqa: E731
    S = 200.0 if dt == np.float64 else 20.0
    av, z, zf, w = new(), new(), new(), new()
    ctx.svrg_init(F, x0, av, z, zf, w)
    assert _close(av.cpu().numpy(), G["svrg_av0"], dt, 10)
    for e in range(3):
        ctx.svrg_iterate(F, g, float(G["svrg_gamma"]), G["svrg_idx"][e], False, av, z, zf, w)
    assert _close(zf.cpu().numpy(), G["svrg_zfull"], dt, S) and _close(w.cpu().numpy(), G["svrg_w"], dt, S)
    for key, sag in (("saga", False), ("sag", True)):
        table = torch.empty((N, d), dtype=x0.dtype, device="cuda")
        av, z = new(), new()
        ctx.saga_init(F, g, float(G[f"{key}_gamma"]), x0, table, av, z)
        assert _close(z.cpu().numpy(), G[f"{key}_z0"], dt, 10)
        ctx.saga_steps(F, g, float(G[f"{key}_gamma"]), sag, G[f"{key}_idx"], table, av, z)
        assert _close(z.cpu().numpy(), G[f"{key}_z"], dt, S) and _close(table.cpu().numpy(), G[f"{key}_table"], dt, S)
    nb = -(-N // 2)
    static = [np.arange(2 * j, min(2 * j + 2, N), dtype=np.int64) for j in range(nb)]
    gam = dev(G["finito_gam"])
    hg = ctx.hat_gamma(gam)
    assert abs(hg - float(G["finito_hat_gamma"])) <= (1e-12 if dt == np.float64 else 1e-5) * hg
    table = torch.empty((N, d), dtype=x0.dtype, device="cuda")
    av, z = new(), new()
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    assert _close(z.cpu().numpy(), G["finito_z0"], dt, 50)
    batches = [static[(t + 1) % nb] for t in range(3 * nb)]
    bptr = np.zeros(len(batches) + 1, np.int64)
    np.cumsum([len(x) for x in batches], out=bptr[1:])
    ctx.finito_steps(F, g, gam, hg, bptr, np.concatenate(batches), table, av, z)
    assert _close(z.cpu().numpy(), G["finito_z"], dt, S) and _close(table.cpu().numpy(), G["finito_table"], dt, S)
    av, z, zf = new(), new(), new()
    ctx.lfinito_init(F, hg, x0, av, z, zf)
    bptr = np.zeros(nb + 1, np.int64)
    np.cumsum([len(x) for x in static], out=bptr[1:])
    for _ in range(3):
        ctx.lfinito_iterate(F, g, gam, hg, bptr, np.concatenate(static), av, z, zf)
    assert _close(z.cpu().numpy(), G["lfinito_z"], dt, S) and _close(zf.cpu().numpy(), G["lfinito_zfull"], dt, S)
    ctx.synchronize()
