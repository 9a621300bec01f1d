"""The sequential SVRG / SAGA chains on rows beyond 8192 elements as ONE chain shared by several workgroups
(chain_wide_kernel, csrc/chain_wide_kernels.h): each workgroup keeps its columns of the state in registers, the partial dot
products travel through a mailbox of self-validating words.  Against the oracle (SVRG_basic.jl:73-82, SAGA_basic.jl:53-68
restated), against the one-workgroup kernel it replaces on these shapes (option chain_no_wide) to rounding, and bitwise
against itself."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import close, dev, make, make_g

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("d", [8193, 9000, 20480, 33000, 70001])
def test_wide_chains_against_the_oracle(ctx, ciao, dtype, d):
    """5 .. 35 workgroups (slices of 2048 columns), a last slice that is short or a single column, both losses, l1 and per-coordinate box prox, SAG, a sample repeated inside the one-step prefetch window."""
    import torch
    from oracle import twin as O
    N = 24
    loss = "logistic" if d % 2 else "ls"
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=d)
    op, dp = make(loss, A, b, 1.0 if loss == "logistic" else float(N), dtype)
    tdt = dev(x0).dtype
    gamma = 0.3 if loss == "logistic" else 1.0 / (7 * N * float(np.max(np.sum(A.astype(np.float64) ** 2, axis=1))))
    idx = ciao.IndexStream(d).rand_indices(N, 200)
    idx[4:8] = idx[4]
    idx[50] = idx[48]
    G = -(-d // 2048)
    for gk in ("l1", "boxvec"):
        og, dg = make_g(gk, dtype, d, lam=0.01)
        av, z, zf, w = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(4))
        ctx.svrg_init(dp, dev(x0), av, z, zf, w)
        ctx.svrg_inner(dp, dg, gamma, idx, av, z, zf, w)
        assert "chain_wide_kernel" in ctx.last_kernel() and f"grid={G} " in ctx.last_kernel(), ctx.last_kernel()
        rav, rz, rzf, rw = O.svrg_init(op, x0)
        O.svrg_inner(op, og, dtype(gamma), idx, rav, rz, rzf, rw)
        close(w, rw, dtype, scale={64: 56, 32: 43}, what=f"svrg_inner w, {G} workgroups, g={gk}", scale64=820)
        close(z, rz, dtype, scale={64: 44, 32: 48}, what=f"svrg_inner z, {G} workgroups, g={gk}", scale64=430)
        # the one-workgroup kernel on the same chain: to rounding; itself again: bitwise
        av1, z1, zf1, w1 = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(4))
        ctx.svrg_init(dp, dev(x0), av1, z1, zf1, w1)
        ctx.set_option("chain_no_wide", 1)
        try:
            ctx.svrg_inner(dp, dg, gamma, idx, av1, z1, zf1, w1)
            assert "chain_big_kernel" in ctx.last_kernel(), ctx.last_kernel()
        finally:
            ctx.set_option("chain_no_wide", 0)
        close(w, w1.cpu().numpy(), dtype, scale=18, what="several workgroups vs one")
        av2, z2, zf2, w2 = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(4))
        ctx.svrg_init(dp, dev(x0), av2, z2, zf2, w2)
        ctx.svrg_inner(dp, dg, gamma, idx, av2, z2, zf2, w2)
        assert torch.equal(w, w2) and torch.equal(z, z2), "not reproducible"
        for sag in (False, True):
            table = torch.empty((N, d), dtype=tdt, device="cuda")
            sav, sz = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
            ctx.saga_init(dp, dg, gamma, dev(x0), table, sav, sz)
            ctx.saga_steps(dp, dg, gamma, sag, idx, table, sav, sz)
            assert "chain_wide_kernel" in ctx.last_kernel() and "alg1" in ctx.last_kernel(), ctx.last_kernel()
            rt, rsav, rsz = O.saga_init(op, og, dtype(gamma), x0)
            O.saga_steps(op, og, dtype(gamma), sag, idx, rt, rsav, rsz)
            close(sz, rsz, dtype, scale={64: 50, 32: 72}, what=f"saga z sag={sag}, {G} workgroups, g={gk}", scale64=840)
            close(sav, rsav, dtype, scale={64: 590, 32: 680}, what=f"saga av sag={sag}", scale64=630)
            close(table, rt, dtype, scale={64: 880, 32: 790}, what="saga table", scale64=800)
            close(sav, table.double().mean(dim=0).cpu().numpy(), dtype, scale={64: 150, 32: 160}, what="av invariant")
    ctx.synchronize()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("d", [8193, 20481])
def test_wide_finito_and_lfinito_chains(ctx, ciao, dtype, d):
    """Small-batch Finito (batches of 1, 2 and 3, per-sample stepsizes, a sample again in the next batch: the table row it reads is
    the one just written) and LFinito iterations on the several-workgroup chain against the oracle and the one-workgroup kernel."""
    import torch
    from oracle import twin as O
    N = 30
    loss = "logistic" if d % 2 else "ls"
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=d + 1)
    lam_f = 1.0 if loss == "logistic" else float(N)
    op, dp = make(loss, A, b, lam_f, dtype)
    og, dg = make_g("l1", dtype, d, lam=0.01)
    tdt = dev(x0).dtype
    Li = (lam_f if loss == "ls" else 0.25) * np.sum(A.astype(np.float64) ** 2, axis=1)
    gamh = (0.999 * N / np.maximum(Li, 1e-3 * Li.max())).astype(dtype)
    gam = dev(gamh)
    hg = ctx.hat_gamma(gam)
    for r in (1, 2, 3):
        table = torch.empty((N, d), dtype=tdt, device="cuda")
        av, z = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
        ctx.finito_init(dp, dg, gam, hg, dev(x0), table, av, z)
        rt, rav, rz, rhg = O.finito_init(op, og, gamh, x0)
        st = ciao.IndexStream(r + d)
        batches = [st.sample_without_replacement(N, r) for _ in range(40)]
        batches[5][0] = batches[4][0]
        batches[6][0] = batches[4][0]
        bptr = np.arange(len(batches) + 1, dtype=np.int64) * r
        ctx.finito_steps(dp, dg, gam, hg, bptr, np.concatenate(batches), table, av, z)
        assert "chain_wide_kernel" in ctx.last_kernel() and "alg2" in ctx.last_kernel(), ctx.last_kernel()
        O.finito_steps(op, og, gamh, rhg, batches, rt, rav, rz)
        close(z, rz, dtype, scale={64: 240, 32: 210}, what=f"finito z, batches of {r}, several workgroups", scale64=130)
        close(av, rav, dtype, scale={64: 230, 32: 200}, what=f"finito av, batches of {r}", scale64=120)
        close(table, rt, dtype, scale={64: 200, 32: 170}, what=f"finito table, batches of {r}", scale64=100)
    lav, lz, lzf = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(3))
    rav, rz, rzf, rhg = O.lfinito_init(op, gamh, x0)
    ctx.lfinito_init(dp, hg, dev(x0), lav, lz, lzf)
    blocks = [np.arange(2 * k, 2 * k + 2, dtype=np.int64) for k in range(N // 2)]
    bp = np.arange(0, N + 1, 2, dtype=np.int64)
    for it in range(3):
        ctx.lfinito_iterate(dp, dg, gam, hg, bp, np.concatenate(blocks), lav, lz, lzf)
        assert "chain_wide_kernel" in ctx.last_kernel() and "alg3" in ctx.last_kernel(), ctx.last_kernel()
        O.lfinito_iterate(op, og, gamh, rhg, blocks, rav, rz, rzf)
        close(lz, rz, dtype, scale=170, what=f"lfinito z it {it}, several workgroups", scale64=150)
        close(lav, rav, dtype, scale={64: 160, 32: 170}, what=f"lfinito av it {it}", scale64=140)
    ctx.synchronize()


def test_wide_chain_epochs_through_the_solver(ctx, ciao):
    """SVRG epochs on d = 10 000 through svrg_iterate (inner cycle on chain_wide_kernel, full passes in between) against the oracle;
    the step numbers of consecutive launches start over (the mailbox is cleared per launch)."""
    import torch
    from oracle import twin as O
    dtype, N, d = np.float64, 60, 10000
    A, b, x0 = P.synthetic("ls", N, d, dtype, seed=5)
    op, dp = make("ls", A, b, float(N), dtype)
    og, dg = make_g("l1", dtype, d, lam=0.01)
    gamma = 1.0 / (7 * N * float(np.max(np.sum(A ** 2, axis=1))))
    av, z, zf, w = (torch.empty(d, dtype=torch.float64, device="cuda") for _ in range(4))
    ctx.svrg_init(dp, dev(x0), av, z, zf, w)
    rav, rz, rzf, rw = O.svrg_init(op, x0)
    st = ciao.IndexStream(3)
    for ep in range(4):
        idx = st.rand_indices(N, 2 * N)
        ctx.svrg_iterate(dp, dg, gamma, idx, False, av, z, zf, w)
        O.svrg_iterate(op, og, dtype(gamma), idx, False, rav, rz, rzf, rw)
        close(zf, rzf, dtype, scale={64: 46}, what=f"svrg epoch {ep} z_full on the several-workgroup chain")
    ctx.synchronize()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_adaptive_finito_on_several_workgroups_is_repeatable(ctx, ciao, dtype):
    """afinito_wide_kernel (rows beyond 32 KiB): 3000 steps over 40 samples -- every sample recurs inside the look-ahead window many
    times, backtracking trials included -- twice from the same state: iterates, table, per-sample scalars (all four copies) and the
    two counters bitwise equal (every workgroup adds the partials in the same order and takes the same decisions), and the whole run
    within rounding of the one-workgroup kernel's."""
    import torch
    from test_gpu_parity import close, dev, make, make_g
    import problems as P
    N, d = 40, 9001
    A, b, x0 = P.synthetic("ls", N, d, dtype, seed=5)
    op, dp = make("ls", A, b, float(N), dtype)
    og, dg = make_g("l1", dtype, d, lam=0.01)
    tdt = dev(x0).dtype
    idx = ciao.IndexStream(12).rand_indices(N, 3000)
    idx[100:104] = idx[100]
    outs = []
    for route in ("wide", "wide", "big"):
        table = torch.empty((N, d), dtype=tdt, device="cuda")
        meta = torch.empty((N, 4, 4), dtype=tdt, device="cuda")
        hg = torch.empty(1, dtype=tdt, device="cuda")
        av, z = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
        ctx.afinito_init(dp, dg, 0.999, dev(x0), table, meta, av, z, hg)
        ctx.set_option("chain_no_wide", int(route == "big"))
        try:
            done, trials = ctx.afinito_steps(dp, dg, 0.999, 1e-9, idx, table, meta, av, z, hg)
            assert ("afinito_wide_kernel" if route == "wide" else "afinito_big_kernel") in ctx.last_kernel(), ctx.last_kernel()
            ctx.synchronize()
        finally:
            ctx.set_option("chain_no_wide", 0)
        assert done == len(idx)
        assert torch.equal(meta[:, 0], meta[:, 1]) and torch.equal(meta[:, 0], meta[:, 2]) and torch.equal(meta[:, 0], meta[:, 3])
        outs.append((trials, z.clone(), av.clone(), hg.clone(), table.clone(), meta.clone()))
    assert outs[0][0] == outs[1][0]
    for u, v in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(u, v), "not repeatable"
    # (device against device, eps32: 3000 dependent backtracking steps whose every dot product is summed in another order -- 64 partial sums
    # against one running sum; the Float32 oracle stands as far from either.  Step-for-step against the oracle: test_adaptive_finito_on_rows_of_any_length)
    if outs[0][0] == outs[2][0]:   # (a backtracking test on the boundary may fall the other way with another summation order)
        close(outs[0][1], outs[2][1].cpu().numpy(), dtype, scale={64: 8100, 32: 9700}, what="several workgroups vs one, z")
        close(outs[0][2], outs[2][2].cpu().numpy(), dtype, scale={64: 8100, 32: 9700}, what="several workgroups vs one, av")
    assert abs(outs[0][0] - outs[2][0]) <= max(2, outs[2][0] // 50)


@pytest.mark.parametrize("gk", ["zero", "box", "boxvec"])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
def test_adaptive_finito_on_several_workgroups_other_g(ctx, ciao, gk, loss):
    """The other prox families on afinito_wide_kernel (Zero; IndBox with scalar and with vector bounds, read per step), both losses,
    against the oracle on the same sample sequence."""
    import torch
    from oracle import twin as O
    N, d, dtype = 24, 8500, np.float64
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=9)
    lam_f = float(N) if loss == "ls" else 1.0
    op, dp = make(loss, A, b, lam_f, dtype)
    og, dg = make_g(gk, dtype, d, lam=0.02)
    tdt = dev(x0).dtype
    table = torch.empty((N, d), dtype=tdt, device="cuda")
    meta = torch.empty((N, 4, 4), dtype=tdt, device="cuda")
    hg = torch.empty(1, dtype=tdt, device="cuda")
    av, z = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
    ctx.afinito_init(dp, dg, 0.999, dev(x0), table, meta, av, z, hg)
    rt, rg, rgam, rfi, rav, rz, rhg = O.afinito_init(op, og, dtype(0.999), x0)
    idx = ciao.IndexStream(3).rand_indices(N, 4 * N)
    idx[7:10] = idx[7]
    done, trials = ctx.afinito_steps(dp, dg, 0.999, 1e-9, idx, table, meta, av, z, hg)
    assert "afinito_wide_kernel" in ctx.last_kernel(), ctx.last_kernel()
    rdone, rhg, rtrials = O.afinito_steps(op, og, dtype(0.999), dtype(1e-9), idx, rt, rg, rgam, rfi, rhg, rav, rz)
    assert done == rdone == len(idx) and trials == rtrials
    close(z, rz, dtype, scale={64: 840}, what=f"adaptive z, g = {gk}")
    close(av, rav, dtype, scale={64: 810}, what=f"adaptive av, g = {gk}")
    close(table, rt, dtype, scale={64: 370}, what=f"adaptive table, g = {gk}")
    close(meta[:, 0, 2], rgam, dtype, scale={64: 2400}, what=f"adaptive gamma_i, g = {gk}")
    ctx.synchronize()
