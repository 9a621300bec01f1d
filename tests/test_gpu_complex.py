"""GPU parity for complex T (CIAOAlgorithms.jl:3 `RealOrComplex`; test_lasso.jl:3 runs ComplexF32 / ComplexF64): every entry
point that accepts CIAO_LOSS_LS_COMPLEX / CIAO_PROX_L1_COMPLEX against the CPU oracle on GENUINELY complex data
(independent real and imaginary parts), vectors as interleaved (re, im) pairs of the real type.

The reference's own complex tests hold real data in complex containers; those run in test_gpu_solvers.py::TestLassoComplex.
Tolerances as in test_gpu_parity.py: scale * eps(R) * |oracle|_inf, scale written at each comparison.
"""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import close, dev, _batches

pytestmark = pytest.mark.gpu

RTYPE = {np.complex128: np.float64, np.complex64: np.float32}
# complex entries per row n (d = 2n reals): below / at / across the wave-per-row and LDS-resident shapes
CSHAPES = [(6, 3), (8, 5), (50, 50), (40, 128), (30, 300), (12, 512), (33, 1000), (9, 2048), (7, 4097)]


def cmake(A, b, lam_f):
    """(oracle Problem, device PackedF) over the same complex data."""
    import torch
    from oracle import twin as O
    from ciaoalgorithms_jl_amd.device import PackedF
    op = O.Problem("ls", A, b, lam_f)
    N = A.shape[0]
    tA = torch.from_numpy(O.as_pairs(A).reshape(N, -1)).cuda()
    tb = torch.from_numpy(O.as_pairs(b)).cuda()
    return op, PackedF.least_squares_complex(tA, tb, lam_f)


def cg(lam):
    from oracle import twin as O
    from ciaoalgorithms_jl_amd.device import ProxG
    import ciaoalgorithms_jl_amd._lib as L
    return O.Prox("l1_complex", lam=lam), ProxG(L.PROX_L1_COMPLEX, lam=lam)


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
def test_complex_gradient_prox_objective(ctx, ctype):
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    for N, n in ((9, 35), (5, 700)):
        A, b, x = P.synthetic_complex(N, n, ctype, seed=n)
        op, dp = cmake(A, b, 9.0)
        xp = O.as_pairs(x)
        y = torch.empty(2 * n, dtype=dev(xp).dtype, device="cuda")
        fv = torch.empty(1, dtype=y.dtype, device="cuda")
        for i in (0, N // 2, N - 1):
            ctx.gradient(dp, i, dev(xp), y, fv)
            gy, f = O.gradient(op.loss, O.as_pairs(A[i]), O.as_pairs(b[i:i + 1]), 9.0, xp)
            close(y, gy, R, scale={64: 38, 32: 34}, what=f"complex gradient i={i}", scale64=15)
            close(fv, [f], R, scale=62, what="complex f_i value")
        og, dg = cg(0.3)
        ctx.prox(dg, dev(xp), 0.5, y)
        # the modulus goes through hypot: libm's and the device's may differ in the last place
        close(y, O.prox(og, xp, R(0.5)), R, scale=8, what="complex prox", scale64=8)
        yh = y.cpu().numpy()
        ref = O.prox(og, xp, R(0.5))
        assert np.array_equal(yh == 0, ref == 0) or np.abs(np.hypot(xp[0::2], xp[1::2]) - 0.15).min() < 1e-6
        obj = ctx.objective(dp, dg, dev(xp))
        robj = O.objective(op, og, xp)
        assert abs(obj - robj) <= (1e-9 if R == np.float64 else 2e-4) * max(1.0, abs(robj))


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("shape", CSHAPES)
def test_complex_full_pass_and_proxgrad(ctx, ctype, shape):
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N, n = shape
    A, b, x = P.synthetic_complex(N, n, ctype, seed=N + n)
    op, dp = cmake(A, b, float(N))
    og, dg = cg(0.01)
    xp = O.as_pairs(x)
    av = torch.empty(2 * n, dtype=dev(xp).dtype, device="cuda")
    ctx.full_gradient(dp, dev(xp), av)
    assert "cplx" in ctx.last_kernel(), ctx.last_kernel()
    rav = O.full_pass(op, xp)
    # independent statement of the sum in numpy complex arithmetic
    A128, x128 = A.astype(np.complex128), x.astype(np.complex128)
    want = (float(N) * (A128.conj().T @ (A128 @ x128 - b.astype(np.complex128)))) / N
    close(av, O.as_pairs(want), R, scale={64: 30, 32: 16}, what="complex full pass vs numpy complex")
    close(av, rav, R, scale={64: 54, 32: 62}, what=f"complex full pass ({ctx.last_kernel()})", scale64=16)
    y = torch.empty_like(av)
    ctx.proxgrad_step(dp, dg, 0.05, dev(xp), av, y)
    ry = O.prox(og, (xp - R(0.05) * rav).astype(R), R(0.05))
    close(y, ry, R, scale={64: 10, 32: 9.8}, what="complex proxgrad y", scale64=8.5)
    # the monitor rides on the same pass
    obj = torch.full((3,), float("nan"), dtype=torch.float64, device="cuda")
    ctx.set_monitor(dg, obj)
    try:
        ctx.full_gradient(dp, dev(xp), av)
        ctx.synchronize()
        ref = O.objective(op, og, xp)
        assert abs(obj[0].item() - ref) <= (1e-11 if R == np.float64 else 5e-5) * max(1.0, abs(ref))
    finally:
        ctx.set_monitor(None, None)


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("n", [64, 300, 512, 1000, 4096])
def test_complex_streaming_kernel_equals_the_plain_one(ctx, ciao, ctype, n):
    """Rows of whole 16-byte chunks take rows_csplit_kernel (one workgroup per row, registers); force_generic routes the same
    calls through rows_cplx_kernel.  Every mode (sweep, two-point batch, SAGA init, Finito init and batch) on both."""
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N = 70
    A, b, x = P.synthetic_complex(N, n, ctype, seed=n)
    _, dp = cmake(A, b, float(N))
    _, dg = cg(0.02)
    xp = dev(O.as_pairs(x))
    gam = dev(np.linspace(0.5, 1.5, N).astype(R))
    hg = ctx.hat_gamma(gam)
    idx = np.sort(ciao.IndexStream(1).sample_without_replacement(N, 20))
    bptr = np.array([0, 20], np.int64)
    res = {}
    for generic in (0, 1):
        ctx.set_option("force_generic", generic)
        ctx.set_option("chain_max_batch", 0)
        try:
            names = []
            av = torch.empty_like(xp)
            ctx.full_gradient(dp, xp, av)
            names.append(ctx.last_kernel())
            table = torch.empty((N, 2 * n), dtype=xp.dtype, device="cuda")
            sav, sz = torch.empty_like(xp), torch.empty_like(xp)
            ctx.saga_init(dp, dg, 0.01, xp, table, sav, sz)
            names.append(ctx.last_kernel())
            ftab = torch.empty_like(table)
            fav, fz = torch.empty_like(xp), torch.empty_like(xp)
            ctx.finito_init(dp, dg, gam, hg, xp, ftab, fav, fz)
            names.append(ctx.last_kernel())
            ctx.finito_steps(dp, dg, gam, hg, bptr, idx, ftab, fav, fz)
            names.append(ctx.last_kernel())
            lav, lz, lzf = (torch.empty_like(xp) for _ in range(3))
            ctx.lfinito_init(dp, hg, xp, lav, lz, lzf)
            ctx.lfinito_iterate(dp, dg, gam, hg, bptr, idx, lav, lz, lzf)
            names.append(ctx.last_kernel())
        finally:
            ctx.set_option("force_generic", 0)
            ctx.set_option("chain_max_batch", -1)
        rowb = 2 * n * np.dtype(R).itemsize
        streaming = not generic and rowb % 16 == 0 and 64 <= rowb // 16 <= 4096       # the planner's rule (rows_launch.inc)
        want = "rows_csplit_kernel" if streaming else "rows_cplx_kernel"
        assert all(want in k for k in names), names
        res[generic] = [t.cpu().numpy() for t in (av, table, sav, sz, ftab, fav, fz, lav, lz, lzf)]
    for u, v, what in zip(res[0], res[1], ("av", "saga table", "saga av", "saga z", "finito table", "finito av", "finito z", "lfinito av",
                                           "lfinito z", "lfinito z_full")):
        close(u, v, R, scale={64: 17, 32: 14}, what=f"streaming vs plain complex kernel: {what}")
    ctx.synchronize()


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("shape", CSHAPES)
def test_complex_svrg_epochs(ctx, ciao, ctype, shape):
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N, n = shape
    A, b, x0 = P.synthetic_complex(N, n, ctype, seed=7)
    op, dp = cmake(A, b, float(N))
    og, dg = cg(0.01)
    gamma = 1.0 / (7 * float(N) * np.max(np.sum(np.abs(A.astype(np.complex128)) ** 2, axis=1)))
    xp = O.as_pairs(x0)
    tdt = dev(xp).dtype
    av, z, zf, w = (torch.empty(2 * n, dtype=tdt, device="cuda") for _ in range(4))
    ctx.svrg_init(dp, dev(xp), av, z, zf, w)
    rav, rz, rzf, rw = O.svrg_init(op, xp)
    close(av, rav, R, scale={64: 56, 32: 72}, what="complex svrg_init av", scale64=14)
    st = ciao.IndexStream(5)
    for ep in range(3):
        idx = st.rand_indices(N, 2 * N)
        ctx.svrg_iterate(dp, dg, gamma, idx, ep == 1, av, z, zf, w)
        O.svrg_iterate(op, og, R(gamma), idx, ep == 1, rav, rz, rzf, rw)
        close(zf, rzf, R, scale={64: 400, 32: 410}, what=f"complex svrg epoch {ep} z_full ({ctx.last_kernel()})", scale64=120)
        close(w, rw, R, scale={64: 400, 32: 410}, what=f"complex svrg epoch {ep} w", scale64=130)
        close(av, rav, R, scale={64: 220, 32: 390}, what=f"complex svrg epoch {ep} av", scale64=69)
    ctx.synchronize()


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("n", [128, 300, 512, 1000, 2048])
def test_complex_lds_dma_chain_is_dispatched_and_bitwise_the_register_ring(ctx, ciao, ctype, n):
    """aligned complex rows of up to 16 KiB run chain_cdma_kernel (the LDS-DMA ring of the real chains with complex
    arithmetic); option chain_no_dma=1 keeps them on chain_cplx_reg_kernel / chain_cplx_kernel.  Same formulas in the same
    order: SVRG inner cycle, SAGA and SAG steps end BITWISE equal on both in fp64 (one entry per 16-byte chunk: the same entries per
    thread) and to rounding in fp32 (two entries per chunk: other partial sums), unmasked and masked shapes; the dispatch is
    asserted through ciao_ctx_last_kernel."""
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N = 40
    A, b, x0 = P.synthetic_complex(N, n, ctype, seed=n)
    op, dp = cmake(A, b, float(N))
    og, dg = cg(0.01)
    gamma = 1.0 / (7 * float(N) * np.max(np.sum(np.abs(A.astype(np.complex128)) ** 2, axis=1)))
    xp = O.as_pairs(x0)
    tdt = dev(xp).dtype
    rowb = 2 * n * np.dtype(R).itemsize
    idx = ciao.IndexStream(n).rand_indices(N, 300)
    idx[50:53] = idx[49]
    outs = {}
    for no_dma in (0, 1):
        ctx.set_option("chain_no_dma", no_dma)
        try:
            av, z, zf, w = (torch.empty(2 * n, dtype=tdt, device="cuda") for _ in range(4))
            ctx.svrg_init(dp, dev(xp), av, z, zf, w)
            ctx.svrg_inner(dp, dg, gamma, idx, av, z, zf, w)
            k_svrg = ctx.last_kernel()
            res = [w.clone(), z.clone()]
            for sag in (False, True):
                table = torch.empty((N, 2 * n), dtype=tdt, device="cuda")
                sav, sz = torch.empty(2 * n, dtype=tdt, device="cuda"), torch.empty(2 * n, dtype=tdt, device="cuda")
                ctx.saga_init(dp, dg, gamma, dev(xp), table, sav, sz)
                ctx.saga_steps(dp, dg, gamma, sag, idx, table, sav, sz)
                res += [sz.clone(), sav.clone(), table.clone()]
            k_saga = ctx.last_kernel()
            ctx.synchronize()
        finally:
            ctx.set_option("chain_no_dma", 0)
        on_ring = (no_dma == 0 and rowb <= 16384)
        for k in (k_svrg, k_saga):
            assert ("chain_cdma_kernel" in k) == on_ring, (no_dma, rowb, k)
            if on_ring:
                assert ("masked" in k) == (rowb % 4096 != 0), k
            else:
                assert "chain_cplx" in k, k
        outs[no_dma] = res
    for u, v, what in zip(outs[0], outs[1], ("svrg w", "svrg z", "saga z", "saga av", "saga table", "sag z", "sag av", "sag table")):
        if ctype == np.complex128:   # one complex entry per 16-byte chunk: both kernels give a thread the same entries, in the same order
            assert torch.equal(u, v), f"chain_cdma_kernel differs from the complex register-ring chain in {what} (n={n}, {ctype.__name__})"
        else:                        # fp32: a chunk holds TWO entries, so the threads' partial dot products group differently
            close(u, v.cpu().numpy(), R, scale={32: 140}, what=f"chain_cdma_kernel vs the complex register-ring chain: {what} (n={n})")
    rav, rz, rzf, rw = O.svrg_init(op, xp)
    O.svrg_inner(op, og, R(gamma), idx, rav, rz, rzf, rw)
    close(outs[0][0], rw, R, scale={64: 1000, 32: 880}, what="complex LDS-DMA chain svrg w vs oracle", scale64=200)


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("sag", [False, True])
@pytest.mark.parametrize("shape", CSHAPES)
def test_complex_saga_steps(ctx, ciao, ctype, sag, shape):
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N, n = shape
    A, b, x0 = P.synthetic_complex(N, n, ctype, seed=9)
    op, dp = cmake(A, b, float(N))
    og, dg = cg(0.02)
    gamma = 1.0 / ((16 if sag else 3) * float(N) * np.max(np.sum(np.abs(A.astype(np.complex128)) ** 2, axis=1)))
    xp = O.as_pairs(x0)
    tdt = dev(xp).dtype
    table = torch.empty((N, 2 * n), dtype=tdt, device="cuda")
    av, z = torch.empty(2 * n, dtype=tdt, device="cuda"), torch.empty(2 * n, dtype=tdt, device="cuda")
    ctx.saga_init(dp, dg, gamma, dev(xp), table, av, z)
    rt, rav, rz = O.saga_init(op, og, R(gamma), xp)
    close(table, rt, R, scale={64: 86, 32: 74}, what="complex saga_init table", scale64=11)
    close(av, rav, R, scale={64: 58, 32: 62}, what="complex saga_init av", scale64=14)
    close(z, rz, R, scale={64: 8, 32: 9}, what="complex saga_init z", scale64=8.9)
    st = ciao.IndexStream(21)
    for chunk in (1, 2, 4 * N, 7):
        idx = st.rand_indices(N, chunk)
        if chunk == 7:
            idx[:] = idx[0]
        ctx.saga_steps(dp, dg, gamma, sag, idx, table, av, z)
        O.saga_steps(op, og, R(gamma), sag, idx, rt, rav, rz)
        close(z, rz, R, scale={64: 260, 32: 410}, what=f"complex saga z after chunk {chunk} ({ctx.last_kernel()})", scale64=180)
        close(av, rav, R, scale={64: 200, 32: 260}, what=f"complex saga av after chunk {chunk}", scale64=120)
        close(table, rt, R, scale={64: 180, 32: 250}, what=f"complex saga table after chunk {chunk}", scale64=55)
    close(av, table.double().mean(dim=0).cpu().numpy(), R, scale={64: 160, 32: 130}, what="complex av invariant")
    ctx.synchronize()


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("shape,r", [((6, 3), 1), ((8, 5), 3), ((50, 50), 7), ((64, 512), 16), ((300, 32), 100), ((20, 750), 4),
                                     ((700, 512), 300), ((12, 4097), 5)])
@pytest.mark.parametrize("path", ["chain", "rows"])
def test_complex_finito_and_lfinito(ctx, ciao, ctype, shape, r, path):
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N, n = shape
    A, b, x0 = P.synthetic_complex(N, n, ctype, seed=4)
    op, dp = cmake(A, b, float(N))
    og, dg = cg(0.02)
    Li = float(N) * np.sum(np.abs(A.astype(np.complex128)) ** 2, axis=1)
    gam = (0.999 * N / Li).astype(R)
    xp = O.as_pairs(x0)
    tdt = dev(xp).dtype
    table = torch.empty((N, 2 * n), dtype=tdt, device="cuda")
    av, z, zf = (torch.empty(2 * n, dtype=tdt, device="cuda") for _ in range(3))
    dgam = dev(gam)
    hg = ctx.hat_gamma(dgam)
    rt, rav, rz, rhg = O.finito_init(op, og, gam, xp)
    ctx.finito_init(dp, dg, dgam, hg, dev(xp), table, av, z)
    close(table, rt, R, scale={64: 10, 32: 13}, what="complex finito_init table", scale64=11)
    close(av, rav, R, scale={64: 56, 32: 65}, what="complex finito_init av", scale64=10)
    close(z, rz, R, scale={64: 56, 32: 65}, what="complex finito_init z", scale64=13)
    ctx.set_option("chain_max_batch", 64 if path == "chain" else 0)
    try:
        st = ciao.IndexStream(33)
        for mode, nit in (("random", 5), ("cyclic", 2 * (-(-N // r)) + 1)):
            batches = _batches(st, N, r, nit, mode)
            bptr = np.zeros(nit + 1, np.int64)
            np.cumsum([len(x) for x in batches], out=bptr[1:])
            ctx.finito_steps(dp, dg, dgam, hg, bptr, np.concatenate(batches), table, av, z)
            O.finito_steps(op, og, gam, rhg, batches, rt, rav, rz)
            close(z, rz, R, scale={64: 900, 32: 1200}, what=f"complex finito z {mode} ({ctx.last_kernel()})", scale64=190)
            close(av, rav, R, scale={64: 900, 32: 1200}, what=f"complex finito av {mode}", scale64=180)
            close(table, rt, R, scale={64: 720, 32: 900}, what=f"complex finito table {mode}", scale64=160)
        # LFinito over the same problem
        rav, rz, rzf, rhg = O.lfinito_init(op, gam, xp)
        ctx.lfinito_init(dp, hg, dev(xp), av, z, zf)
        close(av, rav, R, scale=110, what="complex lfinito_init av", scale64=8)
        nb = -(-N // r)
        static = [np.arange(r * j, min(r * j + r, N), dtype=np.int64) for j in range(nb)]
        for it in range(2):
            order = np.arange(nb) if it == 0 else st.randperm(nb)
            batches = [static[j] for j in order]
            bptr = np.zeros(nb + 1, np.int64)
            np.cumsum([len(x) for x in batches], out=bptr[1:])
            ctx.lfinito_iterate(dp, dg, dgam, hg, bptr, np.concatenate(batches), av, z, zf)
            O.lfinito_iterate(op, og, gam, rhg, batches, rav, rz, rzf)
            close(zf, rzf, R, scale={64: 250, 32: 1100}, what=f"complex lfinito z_full it {it}", scale64=100)
            close(z, rz, R, scale={64: 340, 32: 1500}, what=f"complex lfinito z it {it} ({ctx.last_kernel()})", scale64=150)
            close(av, rav, R, scale={64: 330, 32: 1900}, what=f"complex lfinito av it {it}", scale64=180)
    finally:
        ctx.set_option("chain_max_batch", -1)
    ctx.synchronize()


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
@pytest.mark.parametrize("shape", [(6, 3), (8, 5), (50, 50), (30, 300), (24, 512), (10, 2500)])
def test_complex_adaptive_finito(ctx, ciao, ctype, shape):
    """Finito_adaptive.jl for complex T: init (probe at x0 .+ one(R): real parts only; sqrt(length(x0)) = complex entries) and
    the backtracking steps (afinito_big_kernel<cplx>) against the oracle; invariants of the state as in the real test."""
    import torch
    from oracle import twin as O
    R = RTYPE[ctype]
    N, n = shape
    A, b, x0 = P.synthetic_complex(N, n, ctype, seed=12)
    op, dp = cmake(A, b, float(N))
    og, dg = cg(0.02)
    xp = O.as_pairs(x0)
    tdt = dev(xp).dtype
    alpha, tol_b = 0.999, 1e-9
    table = torch.empty((N, 2 * n), dtype=tdt, device="cuda")
    meta = torch.empty((N, 4, 4), dtype=tdt, device="cuda")
    hg = torch.empty(1, dtype=tdt, device="cuda")
    av, z = torch.empty(2 * n, dtype=tdt, device="cuda"), torch.empty(2 * n, dtype=tdt, device="cuda")
    ctx.afinito_init(dp, dg, alpha, dev(xp), table, meta, av, z, hg)
    assert "rows_cplx_kernel" in ctx.last_kernel()
    rt, rg, rgam, rfi, rav, rz, rhg = O.afinito_init(op, og, R(alpha), xp)
    close(meta[:, 0, 2], rgam, R, scale={64: 300, 32: 200}, what="complex adaptive init gamma_i", scale64=67)
    close(meta[:, 0, 1], rfi, R, scale={64: 48, 32: 43}, what="complex adaptive init f_i(x0)", scale64=8.3)
    close(hg, [rhg], R, scale={64: 14, 32: 18}, what="complex adaptive init hat_gamma")
    close(av, rav, R, scale={64: 22, 32: 27}, what="complex adaptive init av", scale64=15)
    close(z, rz, R, scale=23, what="complex adaptive init z", scale64=17)
    assert torch.equal(table, dev(xp).expand(N, -1))
    # independent statement of gamma_i: L_i = || conj(a_i) lam sum_k a_ik || / sqrt(n) / N
    A128 = A.astype(np.complex128)
    Lint = float(N) * np.abs(A128.sum(axis=1)) * np.linalg.norm(A128, axis=1) / np.sqrt(n) / N
    close(meta[:, 0, 2], alpha / Lint, R, scale={64: 99, 32: 66}, what="complex adaptive gamma_i vs numpy")
    st = ciao.IndexStream(3)
    idx = np.concatenate([st.rand_indices(N, 3 * N), np.arange(N, dtype=np.int64), np.full(4, 1, np.int64)])
    done, trials = ctx.afinito_steps(dp, dg, alpha, tol_b, idx, table, meta, av, z, hg)
    assert "afinito_big_kernel" in ctx.last_kernel() and "cplx" in ctx.last_kernel()
    rdone, rhg, rtrials = O.afinito_steps(op, og, R(alpha), R(tol_b), idx, rt, rg, rgam, rfi, rhg, rav, rz)
    assert done == rdone == len(idx)
    assert abs(trials - rtrials) <= max(2, 0.02 * rtrials), (trials, rtrials)
    if trials == rtrials:
        close(z, rz, R, scale={64: 350, 32: 500}, what=f"complex adaptive z ({ctx.last_kernel()})")
        close(av, rav, R, scale={64: 350, 32: 510}, what="complex adaptive av")
        close(hg, [rhg], R, scale={64: 39, 32: 83}, what="complex adaptive hat_gamma")
        close(meta[:, 0, 2], rgam, R, scale={64: 53, 32: 100}, what="complex adaptive gamma_i")
        close(table, rt, R, scale={64: 350, 32: 500}, what="complex adaptive table")
        cdev = (meta[:, 0, 0].double() + 1j * meta[:, 1, 0].double()).cpu().numpy()          # c_i = lam res_i
        close(O.as_pairs((np.conj(A128) * cdev[:, None])).reshape(N, -1), rg, R, scale={64: 260, 32: 520}, what="complex gradient table conj(a_i) c_i")
    # invariants: av == hat_gamma (sum_i x_i/gamma_i - (1/N) sum_i grad f_i); stored scalars consistent with stored points
    md = meta.double().cpu().numpy()
    gam = md[:, 0, 2]
    hgd = float(hg.item())
    assert abs(hgd - 1.0 / (1.0 / gam).sum()) <= (1e-10 if R == np.float64 else 2e-4) * hgd
    tab = O.as_complex(table.double().cpu().numpy().reshape(-1)).reshape(N, n)
    c = md[:, 0, 0] + 1j * md[:, 1, 0]
    inv = hgd * ((tab / gam[:, None]).sum(axis=0) - (np.conj(A128) * c[:, None]).sum(axis=0) / N)
    close(av, O.as_pairs(inv), R, scale={64: 57, 32: 76}, what="complex adaptive invariant av")
    dots = (A128 * tab).sum(axis=1)
    close(md[:, 0, 3] + 0 * md[:, 1, 3], dots.real, R, scale={64: 14, 32: 8}, what="Re a_i.x_i")
    close(md[:, 1, 3], dots.imag, R, scale=13, what="Im a_i.x_i")
    close(np.stack([c.real, c.imag]), np.stack([(float(N) * (dots - b)).real, (float(N) * (dots - b)).imag]), R, scale={64: 28, 32: 18}, what="c_i = lam res_i")
    assert np.array_equal(md[:, 0], md[:, 2]) and np.array_equal(md[:, 1], md[:, 3])
    ctx.synchronize()


def test_complex_argument_validation(ctx, ciao):
    """Complex rows pair only with Zero / the complex NormL1; odd d, IndBox and the real NormL1 are refused."""
    import torch
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    import ciaoalgorithms_jl_amd._lib as L
    from oracle import twin as O
    A, b, x = P.synthetic_complex(6, 8, np.complex128)
    _, dp = cmake(A, b, 6.0)
    xp = dev(O.as_pairs(x))
    av = torch.empty_like(xp)
    y = torch.empty_like(xp)
    for bad in (ProxG(L.PROX_L1, lam=0.1), ProxG(L.PROX_BOX, lo=-1.0, hi=1.0)):
        with pytest.raises(ciao._lib.CiaoError):
            ctx.proxgrad_step(dp, bad, 0.1, xp, av, y)
    # the complex prox on a real problem is refused too
    Ar, br, xr = P.synthetic("ls", 6, 16, np.float64)
    dpr = PackedF.least_squares(dev(Ar), dev(br), 6.0)
    with pytest.raises(ciao._lib.CiaoError):
        ctx.proxgrad_step(dpr, ProxG(L.PROX_L1_COMPLEX, lam=0.1), 0.1, dev(xr), torch.empty(16, dtype=torch.float64, device="cuda"),
                          torch.empty(16, dtype=torch.float64, device="cuda"))
    # an odd number of reals is not a vector of pairs
    with pytest.raises(ciao._lib.CiaoError):
        ctx.prox(ProxG(L.PROX_L1_COMPLEX, lam=0.1), xp[:7], 0.5, y[:7])
    # the SVRG entry points check the pairing as every other solver entry point does (they accepted a complex problem
    # with the real NormL1 -- thresholded per scalar -- and a real one with the complex NormL1 -- no prox at all)
    idx = np.arange(6, dtype=np.int64)
    z, zf, w = (torch.empty_like(xp) for _ in range(3))
    ctx.svrg_init(dp, xp, av, z, zf, w)
    for call in (lambda g: ctx.svrg_inner(dp, g, 0.1, idx, av, z, zf, w), lambda g: ctx.svrg_iterate(dp, g, 0.1, idx, False, av, z, zf, w)):
        with pytest.raises(ciao._lib.CiaoError) as ei:
            call(ProxG(L.PROX_L1, lam=0.1))
        assert ei.value.status == -1
    xr_d = dev(xr)
    avr, zr, zfr, wr = (torch.empty_like(xr_d) for _ in range(4))
    ctx.svrg_init(dpr, xr_d, avr, zr, zfr, wr)
    for call in (lambda g: ctx.svrg_inner(dpr, g, 0.1, idx, avr, zr, zfr, wr), lambda g: ctx.svrg_iterate(dpr, g, 0.1, idx, False, avr, zr, zfr, wr)):
        with pytest.raises(ciao._lib.CiaoError) as ei:
            call(ProxG(L.PROX_L1_COMPLEX, lam=0.1))
        assert ei.value.status == -1
    # the sharing problem has real iterates: the complex NormL1 is refused (CIAO_ERR_UNSUPPORTED), not silently ignored
    from ciaoalgorithms_jl_amd.device import PackedSepQuad
    Q = torch.ones((3, 4), dtype=torch.float64, device="cuda")
    f = PackedSepQuad(Q, Q.clone(), 1.0, -1.0, 1.0)
    gam = torch.ones(3, dtype=torch.float64, device="cuda")
    table = torch.empty((3, 4), dtype=torch.float64, device="cuda")
    v4 = [torch.zeros(4, dtype=torch.float64, device="cuda") for _ in range(3)]
    hg = torch.empty(1, dtype=torch.float64, device="cuda")
    bad = ProxG(L.PROX_L1_COMPLEX, lam=0.1)
    with pytest.raises(ciao._lib.CiaoError) as ei:
        ctx.proshi_init(f, bad, gam, v4[0], table, v4[1], v4[2], hg)
    assert ei.value.status == -3
    with pytest.raises(ciao._lib.CiaoError) as ei:
        ctx.proshi_steps(f, bad, gam, 0.5, np.array([0, 1], dtype=np.int64), np.array([0], dtype=np.int64), table, v4[1], v4[2])
    assert ei.value.status == -3
    with pytest.raises(ciao._lib.CiaoError) as ei:
        ctx.proshi_steps_blocks(f, bad, gam, 0.5, np.array([0], dtype=np.int64), np.array([1], dtype=np.int64), table, v4[1], v4[2])
    assert ei.value.status == -3
