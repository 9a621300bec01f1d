"""Full-gradient sweeps over rows of tabular size on the matrix cores (rows_smallm_kernel, csrc/rowsm_kernels.h):
tiles of 16 dense rows staged by LDS-DMA, the row dots (A x) and the rank-1 accumulation (A' c) as v_mfma_*_16x16x4.

Held against the oracle's full pass (SVRG_basic.jl:58-63 / :87-92 restated) to a stated multiple of eps(R), and against the
several-rows-per-wave kernel it replaces on these shapes (option small_mfma = 0) to rounding; the objective monitor and the cached
row dots of the SVRG chain ride on it as on the other sweeps."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import close, dev, make, make_g

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
@pytest.mark.parametrize("d", [17, 50, 64, 100, 130, 200, 255])
def test_sweep_on_the_matrix_cores(ctx, ciao, dtype, loss, d):
    """Row counts around the 16-row tile (1, 15, 16, 17), a few tiles, and enough for every wave of the grid to take several;
    d with every remainder modulo 4 (the last MFMA step) and modulo 16 (the last chunk of columns)."""
    import torch
    from oracle import twin as O
    for N in (1, 15, 16, 17, 100, 5000, 70001):
        A, b, x0 = P.synthetic(loss, N, d, dtype, seed=N + d)
        lam_f = float(N) if loss == "ls" else 1.0
        op, dp = make(loss, A, b, lam_f, dtype)
        av = torch.full((d,), float("nan"), dtype=dev(x0).dtype, device="cuda")
        ctx.full_gradient(dp, dev(x0), av)
        kern = ctx.last_kernel()
        rowb = d * np.dtype(dtype).itemsize
        if rowb % 16 == 0 and rowb >= 1024:
            assert "rows_split_kernel" in kern or "rows_fast_kernel" in kern, kern     # those shapes keep their kernels
            continue
        if dtype == np.float64 and d > 144:   # two 16-row tile buffers per wave no longer fit LDS: the several-rows-per-wave kernel
            assert "rows_small_kernel" in kern, kern
            continue
        assert "rows_smallm_kernel" in kern, kern
        close(av, O.full_pass(op, x0), dtype, scale={64: 530, 32: 800}, what=f"matrix-core sweep d={d} N={N}", scale64=250)
        av0 = torch.empty_like(av)
        ctx.set_option("small_mfma", 0)
        try:
            ctx.full_gradient(dp, dev(x0), av0)
            assert "rows_small_kernel" in ctx.last_kernel(), ctx.last_kernel()
        finally:
            ctx.set_option("small_mfma", -1)
        close(av, av0.cpu().numpy(), dtype, scale={64: 20, 32: 400}, what=f"matrix-core sweep vs the several-rows-per-wave kernel d={d} N={N}")
        av2 = torch.empty_like(av)
        ctx.full_gradient(dp, dev(x0), av2)
        assert torch.equal(av, av2), "not reproducible"
    ctx.synchronize()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_row_dots_and_objective_ride_on_the_matrix_core_sweep(ctx, ciao, dtype):
    """SVRG epochs whose full passes run on rows_smallm_kernel: the inner cycle reuses the row dots the pass cached
    (reuse_rowdots), so a wrong dot of any row shows in the next epoch; and the objective monitor's sum of f_i."""
    import torch
    from oracle import twin as O
    N, d = 3001, 50
    for loss in ("ls", "logistic"):
        A, b, x0 = P.synthetic(loss, N, d, dtype, seed=11)
        lam_f = float(N) if loss == "ls" else 1.0
        op, dp = make(loss, A, b, lam_f, dtype)
        og, dg = make_g("l1", dtype, d, lam=0.01)
        Lmax = (lam_f if loss == "ls" else 0.25) * np.max(np.sum(A.astype(np.float64) ** 2, axis=1))
        gamma = 1.0 / (7 * Lmax)
        tdt = dev(x0).dtype
        eps = np.finfo(dtype).eps

        def epochs(mfma):
            st = ciao.IndexStream(5)
            av, z, zf, w = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(4))
            ctx.set_option("small_mfma", -1 if mfma else 0)
            try:
                ctx.svrg_init(dp, dev(x0), av, z, zf, w)
                assert ("rows_smallm_kernel" if mfma else "rows_small_kernel") in ctx.last_kernel(), ctx.last_kernel()
                for _ in range(3):
                    ctx.svrg_iterate(dp, dg, gamma, st.rand_indices(N, N), False, av, z, zf, w, reuse_rowdots=True)
            finally:
                ctx.set_option("small_mfma", -1)
            return av, zf

        av, zf = epochs(True)
        av0, zf0 = epochs(False)
        rav, rz, rzf, rw = O.svrg_init(op, x0)
        scale_av = float(np.abs(rav).max())       # av shrinks as the solve converges: its error is held against the first pass's size
        st = ciao.IndexStream(5)
        for _ in range(3):
            O.svrg_iterate(op, og, dtype(gamma), st.rand_indices(N, N), False, rav, rz, rzf, rw)
        # (a dependent chain's error grows about linearly in its steps: 3 x 3001 here)
        close(zf, rzf, dtype, scale={64: 1500, 32: 98}, what=f"3 svrg epochs over matrix-core passes, z_full vs the oracle ({loss})", scale64=840)
        close(zf, zf0.cpu().numpy(), dtype, scale={64: 38, 32: 11}, what=f"3 svrg epochs, z_full: matrix-core passes vs the several-rows-per-wave kernel ({loss})")
        assert np.abs(av.cpu().numpy() - rav).max() <= 5000 * eps * scale_av
        assert np.abs(av.cpu().numpy() - av0.cpu().numpy()).max() <= 5000 * eps * scale_av
        obj = ctx.objective(dp, dg, zf)
        ref = O.objective(op, og, zf.cpu().numpy())
        assert abs(obj - ref) <= (1e-11 if dtype == np.float64 else 2e-5) * abs(ref), (obj, ref)
    ctx.synchronize()


def test_shapes_the_matrix_core_sweep_leaves_alone(ctx, ciao):
    """Padded rows, rows shorter than 17 elements and fp64 rows beyond 144 elements (two tile buffers per wave no longer fit LDS)
    stay on the several-rows-per-wave kernel; the ring depth option gives the same sums bit for bit."""
    import torch
    for (N, d, pad, dtype, want) in ((500, 50, 3, np.float32, "rows_small_kernel"), (500, 16, 0, np.float32, "rows_small_kernel"),
                                     (500, 5, 0, np.float32, "rows_small_kernel"), (500, 149, 0, np.float64, "rows_small_kernel"),
                                     (500, 50, 0, np.float64, "rows_smallm_kernel"), (500, 50, 0, np.float32, "rows_smallm_kernel")):
        A, b, x0 = P.synthetic("ls", N, d, dtype, seed=N)
        op, dp = make("ls", A, b, float(N), dtype, pad=pad)
        av = torch.empty(d, dtype=dev(x0).dtype, device="cuda")
        ctx.full_gradient(dp, dev(x0), av)
        assert want in ctx.last_kernel(), ctx.last_kernel()
    A, b, x0 = P.synthetic("logistic", 9001, 77, np.float32, seed=1)
    op, dp = make("logistic", A, b, 1.0, np.float32)
    ref = None
    for nb in (0, 2, 3, 4):
        ctx.set_option("small_nb", nb)
        try:
            av = torch.empty(77, dtype=torch.float32, device="cuda")
            ctx.full_gradient(dp, dev(x0), av)
            assert f"ring{nb if nb else 2}" in ctx.last_kernel(), ctx.last_kernel()
        finally:
            ctx.set_option("small_nb", 0)
        ref = av if ref is None else ref
        assert torch.equal(av, ref), nb
    ctx.synchronize()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
def test_table_modes_on_the_matrix_core_kernel(ctx, ciao, dtype, loss):
    """SAGA init, Finito init and Finito batches over row blocks on rows_smallm_kernel with row counts that end in a short tile (1, 17,
    1003 rows; blocks of 333): every table row written, nothing written past the table's end (a sentinel region behind it), and
    each result against the oracle."""
    import torch
    from oracle import twin as O
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    for d in (17, 50, 92):
        for N in (1, 17, 1003):
            A, b, x0 = P.synthetic(loss, N, d, dtype, seed=N + d)
            lam_f = float(N) if loss == "ls" else 1.0
            op, dp = make(loss, A, b, lam_f, dtype)
            og, dg = make_g("l1", dtype, d, lam=0.02)
            Li = (lam_f if loss == "ls" else 0.25) * np.sum(A.astype(np.float64) ** 2, axis=1)
            gam = (0.999 * N / np.maximum(Li, 1e-3 * Li.max())).astype(dtype)
            dgam = dev(gam)
            hg = ctx.hat_gamma(dgam)
            guard = 64
            big = torch.full((N + guard, d), float("nan"), dtype=tdt, device="cuda")
            table = big[:N]
            av, z = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
            gamma = 0.1 / max(Li.max(), 1.0)
            ctx.saga_init(dp, dg, gamma, dev(x0), table, av, z)
            rt, rav, rz = O.saga_init(op, og, dtype(gamma), x0)
            assert torch.isnan(big[N:]).all() and not torch.isnan(table).any(), (d, N)
            close(table, rt, dtype, scale={64: 22, 32: 23}, what=f"saga_init table d={d} N={N}", scale64=19)
            close(av, rav, dtype, scale={64: 98, 32: 65}, what=f"saga_init av d={d} N={N}", scale64=19)
            big.fill_(float("nan"))
            ctx.finito_init(dp, dg, dgam, hg, dev(x0), table, av, z)
            assert "rows_smallm_kernel" in ctx.last_kernel() and "mode3" in ctx.last_kernel(), ctx.last_kernel()
            rt, rav, rz, rhg = O.finito_init(op, og, gam, x0)
            assert torch.isnan(big[N:]).all() and not torch.isnan(table).any(), (d, N)
            close(table, rt, dtype, scale=18, what=f"finito_init table d={d} N={N}", scale64=12)
            close(av, rav, dtype, scale={64: 100, 32: 97}, what=f"finito_init av d={d} N={N}", scale64=10)
            if N < 1003:
                continue
            # static blocks of 333 rows (the last one shorter), cyclic order, every block starting where 16-byte alignment allows or not
            r = 336 if (336 * d * np.dtype(dtype).itemsize) % 16 == 0 else 333
            nbk = -(-N // r)
            order = [(t + 1) % nbk for t in range(nbk + 1)]
            static = [np.arange(r * j, min(r * j + r, N), dtype=np.int64) for j in order]
            ctx.set_option("chain_max_batch", 0)
            try:
                ctx.finito_steps_blocks(dp, dg, dgam, hg, np.array([x[0] for x in static]), np.array([len(x) for x in static]), table, av, z)
            finally:
                ctx.set_option("chain_max_batch", -1)
            O.finito_steps(op, og, gam, rhg, static, rt, rav, rz)
            assert torch.isnan(big[N:]).all() and not torch.isnan(table).any(), (d, N)
            close(z, rz, dtype, scale={64: 300, 32: 510}, what=f"finito blocks z d={d} ({ctx.last_kernel()})", scale64=32)
            close(table, rt, dtype, scale={64: 150, 32: 180}, what=f"finito blocks table d={d}", scale64=19)
    ctx.synchronize()
