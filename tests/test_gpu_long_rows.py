"""Rows beyond 64 KiB (more than 8192 fp64 / 16 384 fp32 elements) in the batch-parallel modes: rows_long_kernel (rowsl_kernels.h), a
CLUSTER of workgroups per row -- every workgroup a column segment, the partial dot products exchanged through the mailbox of the
several-workgroup chains.  All five modes (SVRG_basic.jl:87-92, SAGA_basic.jl:42-47, Finito_basic.jl:77-83 and :109-118,
Finito_LFinito.jl:93-98) against the oracle; against the generic kernel it replaces; run-to-run bitwise; both segment sizes; one and
several rows per cluster; padded rows, row blocks off row 0, index lists, the objective monitor and the cached row dots."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import close, make, make_g, dev

pytestmark = pytest.mark.gpu


def _gam(A, lam_f, loss, N, dtype):
    Li = (lam_f if loss == "ls" else 0.25) * np.sum(A.astype(np.float64) ** 2, axis=1)
    return (0.999 * N / np.maximum(Li, 1e-3 * Li.max())).astype(dtype)


@pytest.mark.parametrize("dtype,d", [(np.float64, 8194), (np.float64, 16384), (np.float64, 40000), (np.float32, 16388), (np.float32, 65536),
                                     (np.float32, 100000)])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
@pytest.mark.parametrize("opts", [{}, {"long_j": 4}, {"long_j": 8}, {"split_blocks_per_cu": 1}])
def test_long_rows_in_all_five_modes(ctx, ciao, dtype, d, loss, opts):
    import torch
    from oracle import twin as O
    N, r = 150, 40
    if opts.get("split_blocks_per_cu"):
        N = 400   # 256 resident workgroups: several rows per cluster
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=d)
    lam_f = float(N) if loss == "ls" else 1.0
    op, dp = make(loss, A, b, lam_f, dtype)
    og, dg = make_g("l1", dtype, d, lam=0.02)
    gam = _gam(A, lam_f, loss, N, dtype)
    tdt = dev(x0).dtype
    dgam = dev(gam)
    hg = ctx.hat_gamma(dgam)
    ctx.set_option("chain_max_batch", 0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        av, av2, z = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(3))
        ctx.full_gradient(dp, dev(x0), av)
        name = ctx.last_kernel()
        assert "rows_long_kernel" in name and f"J{opts.get('long_j', 8)}," in name, name
        ref = O.full_pass(op, x0)
        close(av, ref, dtype, scale={64: 160, 32: 240}, what=f"long rows full gradient d={d} ({name})", scale64=15)
        ctx.full_gradient(dp, dev(x0), av2)
        assert torch.equal(av, av2), "the cluster sweep is not reproducible run to run"
        ctx.set_option("long_rows", 0)
        ctx.full_gradient(dp, dev(x0), av2)
        assert "rows_long_kernel" not in ctx.last_kernel()
        ctx.set_option("long_rows", 1)
        close(av2, ref, dtype, scale={64: 170, 32: 240}, what="the generic kernel on the same rows", scale64=34)
        # SAGA init, Finito init
        table = torch.empty((N, d), dtype=tdt, device="cuda")
        sav, sz = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
        gs = dtype(0.1 / max(float(np.sum(A.astype(np.float64) ** 2, axis=1).max()), 1.0))
        ctx.saga_init(dp, dg, gs, dev(x0), table, sav, sz)
        rt, rav, rz = O.saga_init(op, og, gs, x0)
        close(table, rt, dtype, scale={64: 270, 32: 370}, what="long rows saga_init table", scale64=12)
        close(sav, rav, dtype, scale={64: 160, 32: 250}, what="long rows saga_init av", scale64=15)
        close(sz, rz, dtype, scale=8, what="long rows saga_init z", scale64=8)
        rt, rav, rz, rhg = O.finito_init(op, og, gam, x0)
        ctx.finito_init(dp, dg, dgam, hg, dev(x0), table, av, z)
        assert "rows_long_kernel" in ctx.last_kernel() and "mode3" in ctx.last_kernel(), ctx.last_kernel()
        close(table, rt, dtype, scale={64: 9.200000000000001, 32: 8.8}, what="long rows finito_init table", scale64=8)
        close(av, rav, dtype, scale={64: 92, 32: 67}, what="long rows finito_init av", scale64=14)
        # Finito batches: random index lists, then static blocks as index lists AND as row blocks (bitwise the same)
        st = ciao.IndexStream(d)
        rnd = [st.sample_without_replacement(N, r) for _ in range(3)]
        bptr = np.arange(len(rnd) + 1, dtype=np.int64) * r
        ctx.finito_steps(dp, dg, dgam, hg, bptr, np.concatenate(rnd), table, av, z)
        assert "rows_long_kernel" in ctx.last_kernel() and f"J{opts.get('long_j', 4)},mode4" in ctx.last_kernel(), ctx.last_kernel()
        O.finito_steps(op, og, gam, rhg, rnd, rt, rav, rz)
        close(z, rz, dtype, scale={64: 150, 32: 290}, what=f"long rows finito z, index lists ({ctx.last_kernel()})", scale64=34)
        close(table, rt, dtype, scale={64: 120, 32: 210}, what="long rows finito table, index lists", scale64=28)
        nb = -(-N // r)
        static = [np.arange(r * j, min(r * j + r, N), dtype=np.int64) for j in [(t + 1) % nb for t in range(nb + 1)]]
        t2, a2, z2 = table.clone(), av.clone(), z.clone()
        bp = np.zeros(len(static) + 1, np.int64)
        np.cumsum([len(x) for x in static], out=bp[1:])
        ctx.finito_steps(dp, dg, dgam, hg, bp, np.concatenate(static), table, av, z)
        ctx.finito_steps_blocks(dp, dg, dgam, hg, np.array([x[0] for x in static]), np.array([len(x) for x in static]), t2, a2, z2)
        assert torch.equal(z, z2) and torch.equal(av, a2) and torch.equal(table, t2), "row blocks and the same batches as index lists differ"
        O.finito_steps(op, og, gam, rhg, static, rt, rav, rz)
        close(z, rz, dtype, scale={64: 400, 32: 1400}, what="long rows finito z, row blocks", scale64=69)
        close(table, rt, dtype, scale={64: 370, 32: 1300}, what="long rows finito table, row blocks", scale64=59)
        inv = (table.double() / dgam.double()[:, None]).sum(dim=0).cpu().numpy() * hg
        close(av, inv, dtype, scale=26, what="long rows finito av invariant")
        # LFinito: the full pass + the batch sweep with two dot products per row
        lav, lz, lzf = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(3))
        rav, rz, rzf, rhg = O.lfinito_init(op, gam, x0)
        ctx.lfinito_init(dp, hg, dev(x0), lav, lz, lzf)
        blocks = [np.arange(r * j, min(r * j + r, N), dtype=np.int64) for j in range(nb)]
        bp = np.zeros(nb + 1, np.int64)
        np.cumsum([len(x) for x in blocks], out=bp[1:])
        for it in range(2):
            ctx.lfinito_iterate(dp, dg, dgam, hg, bp, np.concatenate(blocks), lav, lz, lzf)
            assert "rows_long_kernel" in ctx.last_kernel() and "mode1" in ctx.last_kernel(), ctx.last_kernel()
            O.lfinito_iterate(op, og, gam, rhg, blocks, rav, rz, rzf)
            close(lz, rz, dtype, scale={64: 450, 32: 4000}, what=f"long rows lfinito z it {it}", scale64=120)
            close(lav, rav, dtype, scale={64: 410, 32: 4300}, what=f"long rows lfinito av it {it}", scale64=130)
    finally:
        ctx.set_option("chain_max_batch", -1)
        ctx.set_option("long_rows", 1)
        for k in opts:
            ctx.set_option(k, 0)
    ctx.synchronize()


@pytest.mark.parametrize("dtype,d", [(np.float64, 9000), (np.float32, 20000)])
def test_long_rows_padded_stride_monitor_and_an_svrg_epoch(ctx, ciao, dtype, d):
    """A padded row stride (whole 16-byte chunks), the objective riding on the sweep (one cluster workgroup contributes it), and
    SVRG epochs whose inner cycle (the several-workgroup chain) uses the row dots the cluster sweep cached."""
    import torch
    from oracle import twin as O
    N = 90
    A, b, x0 = P.synthetic("ls", N, d, dtype, seed=d + 1)
    pad = 16 // np.dtype(dtype).itemsize * 3
    op, dp = make("ls", A, b, float(N), dtype, pad=pad)
    og, dg = make_g("l1", dtype, d, lam=0.01)
    tdt = dev(x0).dtype
    av = torch.empty(d, dtype=tdt, device="cuda")
    ctx.full_gradient(dp, dev(x0), av)
    assert "rows_long_kernel" in ctx.last_kernel(), ctx.last_kernel()
    close(av, O.full_pass(op, x0), dtype, scale={64: 62, 32: 120}, what="long padded rows full gradient", scale64=10)
    fv = ctx.objective(dp, dg, dev(x0))
    rf = O.objective(op, og, x0)
    assert abs(fv - rf) <= 200 * np.finfo(dtype).eps * abs(rf), (fv, rf)
    # SVRG: init + 2 epochs of m = N against the oracle (the row dots of the full pass reused by the inner cycle)
    Lmax = float(N) * float(np.sum(A.astype(np.float64) ** 2, axis=1).max())
    gamma = 1.0 / (7 * Lmax)
    z, zf, w = (torch.empty(d, dtype=tdt, device="cuda") for _ in range(3))
    ctx.svrg_init(dp, dev(x0), av, z, zf, w)
    rav, rz, rzf, rw = O.svrg_init(op, x0)
    close(av, rav, dtype, scale={64: 62, 32: 120}, what="svrg_init av on long rows", scale64=10)
    st = ciao.IndexStream(11)
    for ep in range(2):
        idx = st.rand_indices(N, N)
        ctx.svrg_iterate(dp, dg, gamma, idx, False, av, z, zf, w, reuse_rowdots=True)
        O.svrg_iterate(op, og, dtype(gamma), idx, False, rav, rz, rzf, rw)
        close(zf, rzf, dtype, scale={64: 18, 32: 28}, what=f"SVRG epoch {ep} on long rows, z_full", scale64=230)
        close(av, rav, dtype, scale={64: 75, 32: 100}, what=f"SVRG epoch {ep} on long rows, av", scale64=110)
    ctx.synchronize()


def test_long_rows_on_few_rows_and_at_the_segment_limit(ctx, ciao):
    """One row, two rows (fewer rows than clusters), and the longest row the exchange's one-wave tree takes (64 segments of 32 KiB:
    262 144 fp64 elements); one chunk more falls back to the generic kernel."""
    import torch
    from oracle import twin as O
    dtype = np.float64
    for N, d, long in ((1, 20000, True), (2, 262144, True), (3, 262146, False)):
        A, b, x0 = P.synthetic("ls", N, d, dtype, seed=N)
        op, dp = make("ls", A, b, float(N), dtype)
        av = torch.empty(d, dtype=torch.float64, device="cuda")
        ctx.full_gradient(dp, dev(x0), av)
        assert ("rows_long_kernel" in ctx.last_kernel()) == long, ctx.last_kernel()
        close(av, O.full_pass(op, x0), dtype, scale={64: 330}, what=f"long rows N={N} d={d}")
    ctx.synchronize()
