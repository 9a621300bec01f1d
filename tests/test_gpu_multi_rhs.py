"""The full-gradient pass for K iterates in ONE pass over A (ciao_full_gradient_multi / ciao_svrg_epoch_tail_multi;
csrc/mrhs_kernels.h: the one place on this path where the gradients are a dense A.x contraction, on the matrix cores).

Not in the reference, which solves one problem per call: each of the K results is held against (a) the oracle's full pass
(SVRG_basic.jl:87-92 restated) at that iterate and (b) the library's own single sweep at that iterate -- to a stated multiple of
eps(R) |av|_inf, not bitwise: the pass adds its rows in another order (MFMA tiles of 16 rows, partitions in order)."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import close, dev, make

pytestmark = pytest.mark.gpu


def _iterates(d, K, dtype, seed):
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal(d) * (0.3 + 0.1 * k)).astype(dtype) for k in range(K)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
@pytest.mark.parametrize("d,K,N", [(256, 2, 37), (256, 16, 1000), (512, 21, 3001), (1024, 48, 2050), (1024, 5, 16), (1024, 33, 40000), (768, 19, 5000), (128, 17, 3000)])
def test_one_pass_over_the_rows_for_K_iterates(ctx, ciao, dtype, loss, d, K, N):
    """K iterates (asymmetric: every solve its own scale), K not a multiple of the 16-solve column block, N not a multiple of the
    16-row tile and smaller than the partition count; against the oracle and against K single sweeps."""
    import torch
    from oracle import oracle as O
    A, b, _ = P.synthetic(loss, N, d, dtype, seed=d + K)
    lam_f = float(N) if loss == "ls" else 1.0
    op, dp = make(loss, A, b, lam_f, dtype)
    xs_h = _iterates(d, K, dtype, seed=K)
    xs = [dev(x) for x in xs_h]
    avs = [torch.full_like(x, float("nan")) for x in xs]
    ctx.full_gradient_multi(dp, xs, avs)
    if d == 128 and dtype == np.float32:   # (a wave's share of a 128-column fp32 tile is under one 256-byte strip: K single sweeps)
        assert "mrhs_kernel" not in ctx.last_kernel(), ctx.last_kernel()
    else:
        assert f"mrhs_kernel<{'f64' if dtype == np.float64 else 'f32'},SL{d // 16}>" in ctx.last_kernel() and f"K={K}" in ctx.last_kernel(), ctx.last_kernel()
    solo = torch.empty_like(xs[0])
    for k in range(K):
        ref = O.full_pass(op, xs_h[k])
        close(avs[k], ref, dtype, scale={64: 670, 32: 650}, what=f"multi-rhs pass, solve {k} of {K} vs the oracle (d={d}, N={N})")
        ctx.full_gradient(dp, xs[k], solo)
        close(avs[k], solo.cpu().numpy(), dtype, scale={64: 44, 32: 48}, what=f"multi-rhs pass, solve {k} vs its own single sweep")
    # deterministic: the same call again is bitwise the same
    avs2 = [torch.empty_like(x) for x in xs]
    ctx.full_gradient_multi(dp, xs, avs2)
    assert all(torch.equal(u, v) for u, v in zip(avs, avs2))
    ctx.synchronize()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_shapes_the_kernel_does_not_take_run_as_single_sweeps(ctx, ciao, dtype):
    """d outside {256, 512, 1024}, a padded row stride that is not 16-byte aligned, one iterate, the option switched off: K single
    sweeps inside the call -- bitwise what ciao_full_gradient gives."""
    import torch
    for (N, d, pad, K, off) in ((500, 100, 0, 3, 0), (300, 1000, 0, 4, 0), (200, 256, 1, 3, 0), (400, 1024, 0, 1, 0), (400, 1024, 0, 4, 1)):
        A, b, _ = P.synthetic("ls", N, d, dtype, seed=N)
        op, dp = make("ls", A, b, float(N), dtype, pad=pad)
        xs = [dev(x) for x in _iterates(d, K, dtype, seed=d)]
        avs = [torch.empty_like(x) for x in xs]
        ctx.set_option("multi_rhs_off", off)
        try:
            ctx.full_gradient_multi(dp, xs, avs)
        finally:
            ctx.set_option("multi_rhs_off", 0)
        assert "mrhs_kernel" not in ctx.last_kernel(), ctx.last_kernel()
        solo = torch.empty_like(xs[0])
        for k in range(K):
            ctx.full_gradient(dp, xs[k], solo)
            assert torch.equal(solo, avs[k]), (N, d, pad, K, k)
    ctx.synchronize()


def test_svrg_epoch_tails_of_K_solves_in_one_pass(ctx, ciao):
    """ciao_svrg_epoch_tail_multi against K calls of ciao_svrg_epoch_tail: z_full, w and z bitwise (the tail is elementwise), av to
    rounding; then solvers.solve_together(one_pass=True) against the default (K sweeps per outer step) on a regularisation path."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    from ciaoalgorithms_jl_amd import solvers as S
    dtype, N, d, K = np.float64, 3000, 1024, 20
    A, b, _ = P.synthetic("ls", N, d, dtype, seed=3)
    op, dp = make("ls", A, b, float(N), dtype)
    st = [[dev(v) for v in _iterates(d, 4, dtype, seed=10 + k)] for k in range(K)]        # (av, z, z_full, w) per solve
    st2 = [[t.clone() for t in s] for s in st]
    ctx.svrg_epoch_tail_multi(dp, 17, False, [s[0] for s in st], [s[1] for s in st], [s[2] for s in st], [s[3] for s in st])
    assert "mrhs_kernel" in ctx.last_kernel()
    for k in range(K):
        ctx.svrg_epoch_tail(dp, 17, False, *st2[k])
        assert torch.equal(st[k][1], st2[k][1]) and torch.equal(st[k][2], st2[k][2]) and torch.equal(st[k][3], st2[k][3])
        close(st[k][0], st2[k][0].cpu().numpy(), dtype, scale={64: 34}, what=f"epoch tail of solve {k}: av")
    # the lockstep driver, one pass per outer step against K sweeps per outer step
    F = PackedF.least_squares(dp.A, dp.b, float(N))
    Li = float(N) * np.sum(A * A, axis=1)
    gamma = 1.0 / (7 * float(Li.max()))
    lams = [0.01 * (1 + k) for k in range(6)]

    def solve(one_pass):
        ctx.set_option("svrg_cache_rowdots", 0)
        try:
            its = [S.iterator(S.SVRG(np.float64, γ=gamma), np.zeros(d), F=F, g=ProxG(L.PROX_L1, lam=lam), N=N, ctx=ctx,
                              stream=ciao.IndexStream(100 + k)) for k, lam in enumerate(lams)]
            return S.solve_together(its, 5, one_pass=one_pass)
        finally:
            ctx.set_option("svrg_cache_rowdots", 1)

    xs0, it0 = solve(False)
    xs1, it1 = solve(True)
    assert it0 == it1 == 5
    for u, v in zip(xs0, xs1):
        assert np.abs(u - v).max() <= 1e-11 * max(np.abs(u).max(), 1e-30)
    ctx.synchronize()
