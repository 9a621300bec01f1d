"""Several independent chains in one launch (include/ciao_hip.h: ciao_ctx_chain_batch_begin / _end).  The reference solves one
problem per call (SVRG_basic.jl:73-82, SAGA_basic.jl:53-68 are one sequential chain each); a batch is that many of those calls made
at once, one workgroup per chain -- so the bar is: each chain's results are BITWISE those of the same call made alone (which the
parity tests compare with the oracle), whatever else is in the batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(N, d, dtype, loss="ls", seed=0):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((N, d)) / np.sqrt(d)).astype(dtype)
    if loss == "ls":
        b = rng.standard_normal(N).astype(dtype)
        return PackedF(L.LOSS_LS, torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), float(N))
    y = np.sign(rng.standard_normal(N)).astype(dtype)
    return PackedF(L.LOSS_LOGISTIC, torch.from_numpy(A).cuda(), torch.from_numpy(y).cuda(), 1.0)


def _svrg_chains(ctx, F, K, m, seed):
    """K SVRG inner cycles over the same rows: own lambda, own index stream, own state (w0 differs too)."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    rng = np.random.default_rng(seed)
    tdt = F.A.dtype
    d = F.d
    chains = []
    for k in range(K):
        x0 = torch.from_numpy(0.1 * rng.standard_normal(d)).to("cuda", tdt)
        av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
        ctx.svrg_init(F, x0, av, z, zf, w)
        g = ProxG(L.PROX_L1, lam=1e-3 * (k + 1)) if k % 3 != 2 else ProxG(L.PROX_BOX, lo=-0.05 * (k + 1), hi=0.04 * (k + 1))
        idx = torch.from_numpy(rng.integers(0, F.N, size=m + 17 * k)).cuda()
        chains.append(dict(g=g, gamma=0.3 / (F.N * (k + 1)), idx=idx, av=av, z=z, zf=zf, w=w))   # L_i = lam_f |a_i|^2 ~ N
    ctx.synchronize()
    return chains


def _snapshot(chains, keys):
    return [{k: c[k].clone() for k in keys} for c in chains]


def _restore(chains, snap):
    for c, s in zip(chains, snap):
        for k, v in s.items():
            c[k].copy_(v)


@pytest.mark.parametrize("dtype,d", [(np.float64, 1024), (np.float32, 1024), (np.float64, 96), (np.float32, 200), (np.float32, 3000)],
                         ids=["f64-d1024", "f32-d1024", "f64-d96-one-wave", "f32-d200-one-wave", "f32-d3000-masked"])
def test_svrg_chains_in_one_launch_are_bitwise_the_chains_alone(ctx, dtype, d):
    import torch
    F = _problem(2500, d, dtype)
    chains = _svrg_chains(ctx, F, K=7, m=1500, seed=d)
    start = _snapshot(chains, ("z", "w"))
    for c in chains:                                  # one by one: the calls the parity tests compare with the oracle
        ctx.svrg_inner(F, c["g"], c["gamma"], c["idx"], c["av"], c["z"], c["zf"], c["w"])
    ctx.synchronize()
    assert "grid=1" in ctx.last_kernel()
    alone = _snapshot(chains, ("z", "w"))
    _restore(chains, start)
    with ctx.chain_batch():
        for c in chains:
            ctx.svrg_inner(F, c["g"], c["gamma"], c["idx"], c["av"], c["z"], c["zf"], c["w"])
    ctx.synchronize()
    assert ctx.last_kernel().startswith("chain batch: 7 chains in 1 launch(es)") and "grid=7" in ctx.last_kernel(), ctx.last_kernel()
    for k, (c, s) in enumerate(zip(chains, alone)):
        assert torch.isfinite(s["w"]).all() and torch.isfinite(s["z"]).all() and not torch.equal(s["w"], start[k]["w"])
        assert torch.equal(c["w"], s["w"]) and torch.equal(c["z"], s["z"]), f"chain {k}"


@pytest.mark.parametrize("dtype,d,opts", [(np.float32, 1024, {}), (np.float64, 512, {}), (np.float32, 1024, {"chain_no_ws": 1}), (np.float32, 100, {})],
                         ids=["f32-d1024-ws", "f64-d512-ws", "f32-d1024-dma", "f32-d100-one-wave"])
def test_saga_chains_in_one_launch_are_bitwise_the_chains_alone(ctx, dtype, d, opts):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        F = _problem(1200, d, dtype, loss="logistic")
        rng = np.random.default_rng(7)
        tdt = F.A.dtype
        chains = []
        for k in range(5):
            x0 = torch.from_numpy(0.1 * rng.standard_normal(d)).to("cuda", tdt)
            table = torch.empty((F.N, d), dtype=tdt, device="cuda")
            av, z = torch.empty_like(x0), torch.empty_like(x0)
            g = ProxG(L.PROX_L1, lam=2e-3 * (k + 1))
            gamma = 0.3 / (k + 1)
            ctx.saga_init(F, g, gamma, x0, table, av, z)
            idx = torch.from_numpy(rng.integers(0, F.N, size=2000 + 33 * k)).cuda()
            chains.append(dict(g=g, gamma=gamma, sag=(k == 3), idx=idx, table=table, av=av, z=z))
        ctx.synchronize()
        start = _snapshot(chains, ("table", "av", "z"))
        for c in chains:
            ctx.saga_steps(F, c["g"], c["gamma"], c["sag"], c["idx"], c["table"], c["av"], c["z"])
        ctx.synchronize()
        single = ctx.last_kernel().split("<")[0]
        alone = _snapshot(chains, ("table", "av", "z"))
        _restore(chains, start)
        with ctx.chain_batch():
            for c in chains:
                ctx.saga_steps(F, c["g"], c["gamma"], c["sag"], c["idx"], c["table"], c["av"], c["z"])
        ctx.synchronize()
        # (alone, rows of 2-8 KiB take the wave-specialised kernel; in a batch chain_dma_kernel, whose results are bitwise the same)
        assert single in ("chain_ws_kernel", "chain_dma_kernel")
        assert ctx.last_kernel().startswith("chain batch: 5 chains in 1 launch(es)") and "chain_dma_kernel" in ctx.last_kernel(), ctx.last_kernel()
        for k, (c, s) in enumerate(zip(chains, alone)):
            for key in ("table", "av", "z"):
                assert torch.isfinite(s[key]).all()
                assert torch.equal(c[key], s[key]), f"chain {k}: {key}"
    finally:
        for k in opts:
            ctx.set_option(k, 0)


def test_a_batch_of_different_kernels_takes_one_launch_each(ctx):
    """chains of different shapes / algorithms in one batch: grouped by kernel, every chain still bitwise its solo run"""
    import torch
    Fa, Fb = _problem(900, 1024, np.float64, seed=1), _problem(700, 256, np.float32, seed=2)
    ca, cb = _svrg_chains(ctx, Fa, K=3, m=800, seed=1), _svrg_chains(ctx, Fb, K=4, m=600, seed=2)
    sa, sb = _snapshot(ca, ("z", "w")), _snapshot(cb, ("z", "w"))
    for F, cs in ((Fa, ca), (Fb, cb)):
        for c in cs:
            ctx.svrg_inner(F, c["g"], c["gamma"], c["idx"], c["av"], c["z"], c["zf"], c["w"])
    ctx.synchronize()
    aa, ab = _snapshot(ca, ("z", "w")), _snapshot(cb, ("z", "w"))
    _restore(ca, sa)
    _restore(cb, sb)
    with ctx.chain_batch():
        for i in range(4):                       # interleaved
            if i < 3:
                c = ca[i]
                ctx.svrg_inner(Fa, c["g"], c["gamma"], c["idx"], c["av"], c["z"], c["zf"], c["w"])
            c = cb[i]
            ctx.svrg_inner(Fb, c["g"], c["gamma"], c["idx"], c["av"], c["z"], c["zf"], c["w"])
    ctx.synchronize()
    assert ctx.last_kernel().startswith("chain batch: 7 chains in 2 launch(es)"), ctx.last_kernel()
    for cs, al in ((ca, aa), (cb, ab)):
        for c, s in zip(cs, al):
            assert torch.equal(c["w"], s["w"]) and torch.equal(c["z"], s["z"])


def test_chain_batch_refuses_what_it_cannot_run(ctx):
    import torch
    from ciaoalgorithms_jl_amd._lib import CiaoError
    F = _problem(500, 256, np.float64)
    chains = _svrg_chains(ctx, F, K=2, m=100, seed=3)
    c0, c1 = chains
    start = _snapshot(chains, ("z", "w"))
    # two chains writing the same vector
    with pytest.raises(CiaoError, match="overlap"):
        with ctx.chain_batch():
            ctx.svrg_inner(F, c0["g"], c0["gamma"], c0["idx"], c0["av"], c0["z"], c0["zf"], c0["w"])
            ctx.svrg_inner(F, c1["g"], c1["gamma"], c1["idx"], c1["av"], c0["z"], c1["zf"], c1["w"])
    ctx.synchronize()
    for c, s in zip(chains, start):                   # nothing was launched
        assert torch.equal(c["w"], s["w"]) and torch.equal(c["z"], s["z"])
    # any other entry point while the batch is open; the exception drops the batch and the ctx works again
    with pytest.raises(CiaoError, match="chain batch is open"):
        with ctx.chain_batch():
            ctx.svrg_inner(F, c0["g"], c0["gamma"], c0["idx"], c0["av"], c0["z"], c0["zf"], c0["w"])
            ctx.full_gradient(F, c0["w"], c0["av"])
    ctx.synchronize()
    assert torch.equal(c0["w"], start[0]["w"])
    # a shape the ring kernels do not take (odd d: no 16-byte structure)
    Fo = _problem(300, 77, np.float64)
    co = _svrg_chains(ctx, Fo, K=1, m=50, seed=4)[0]
    with pytest.raises(CiaoError, match="chain batch takes"):
        with ctx.chain_batch():
            ctx.svrg_inner(Fo, co["g"], co["gamma"], co["idx"], co["av"], co["z"], co["zf"], co["w"])
    # _end without _begin
    with pytest.raises(CiaoError, match="no chain batch is open"):
        from ciaoalgorithms_jl_amd import _lib as L
        L.check(ctx.lib.ciao_ctx_chain_batch_end(ctx._h, 1))
    # an empty batch is fine, and the ctx is usable afterwards
    with ctx.chain_batch():
        pass
    ctx.svrg_inner(F, c0["g"], c0["gamma"], c0["idx"], c0["av"], c0["z"], c0["zf"], c0["w"])
    ctx.synchronize()
    assert not torch.equal(c0["w"], start[0]["w"])


def test_many_more_chains_than_compute_units(ctx):
    """600 one-wave chains and 300 four-wave ones (more workgroups than the 256 CUs hold at once: the rest queue)"""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    for d, K, dtype in ((64, 600, np.float32), (1024, 300, np.float32)):
        F = _problem(400, d, dtype, seed=d)
        tdt = F.A.dtype
        rng = np.random.default_rng(d)
        W = torch.from_numpy(0.1 * rng.standard_normal((K, d))).to("cuda", tdt)
        Z = torch.zeros_like(W)
        x0 = torch.zeros(d, dtype=tdt, device="cuda")
        av, z_, zf, w_ = (torch.empty_like(x0) for _ in range(4))
        ctx.svrg_init(F, x0, av, z_, zf, w_)
        idx = torch.from_numpy(rng.integers(0, F.N, size=(K, 300))).cuda()
        gs = [ProxG(L.PROX_L1, lam=1e-4 * (1 + k % 11)) for k in range(K)]
        W0 = W.clone()
        with ctx.chain_batch():
            for k in range(K):                   # av and z_full shared (read-only), w and z per chain
                ctx.svrg_inner(F, gs[k], 0.2 / F.N, idx[k], av, Z[k], zf, W[k])
        ctx.synchronize()
        assert f"chain batch: {K} chains in 1 launch(es)" in ctx.last_kernel()
        for k in (0, 1, K // 2, K - 1):
            wk, zk = W0[k].clone(), torch.zeros(d, dtype=tdt, device="cuda")
            ctx.svrg_inner(F, gs[k], 0.2 / F.N, idx[k], av, zk, zf, wk)
            ctx.synchronize()
            assert torch.isfinite(wk).all() and torch.equal(wk, W[k]) and torch.equal(zk, Z[k]), f"d={d} chain {k}"


# ======================================================================================================================
# the host mirror's lockstep driver: K solver runs through the reference-style API, their chains batched
# ======================================================================================================================
def _path_problem(ops, dtype, N=600, d=256, seed=5):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((N, d)) / np.sqrt(d)).astype(dtype)
    xt = rng.standard_normal(d) * (rng.random(d) < 0.1)
    b = (A @ xt + 0.01 * rng.standard_normal(N)).astype(dtype)
    F = [ops.LeastSquares(A[i:i + 1, :], b[i:i + 1], float(N)) for i in range(N)]     # test_lasso.jl:52-54
    Lc = N * (A.astype(np.float64) ** 2).sum(1)
    return A, b, F, Lc


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_svrg_regularisation_path_solved_together_is_bitwise_the_solves_alone(ctx, dtype):
    """SVRG (SVRG.jl:46-84) for six values of lambda over the same rows: solve_together == six functor calls, bit for bit"""
    import ciaoalgorithms_jl_amd.operators as ops
    import ciaoalgorithms_jl_amd.solvers as S
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    A, b, F, Lc = _path_problem(ops, dtype)
    N, d = A.shape
    lams = [0.3, 0.1, 0.03, 0.01, 0.003, 0.001]
    gam = dtype(1 / (7 * Lc.max()))                                                   # test_lasso.jl:164
    x0 = np.zeros(d, dtype)
    ctx.set_option("svrg_cache_rowdots", 0)        # a batch's inner cycles recompute a_i'z_full: compare with solves that do too
    try:
        alone = [S.SVRG(dtype, γ=gam, maxit=6)(x0, F=F, g=ops.NormL1(l), N=N, ctx=ctx, stream=IndexStream(k))
                 for k, l in enumerate(lams)]
        its = [S.iterator(S.SVRG(dtype, γ=gam), x0, F=F, g=ops.NormL1(l), N=N, ctx=ctx, stream=IndexStream(k)) for k, l in enumerate(lams)]
        xs, n = S.solve_together(its, maxit=6)
    finally:
        ctx.set_option("svrg_cache_rowdots", 1)
    assert n == 6 and all(a[1] == 6 for a in alone)
    for k, (x, (xa, _)) in enumerate(zip(xs, alone)):
        assert x.dtype == dtype and np.isfinite(x).all() and np.array_equal(x, xa), f"lambda #{k}"
    # more regularisation, smaller solution: the path is a path
    l1 = [float(np.abs(x).sum()) for x in xs]
    assert l1[0] < l1[-1]
    # ... and with the row-dot cache ON the separate solves agree to rounding (a different, equally valid summation order)
    xa, _ = S.SVRG(dtype, γ=gam, maxit=6)(x0, F=F, g=ops.NormL1(lams[2]), N=N, ctx=ctx, stream=IndexStream(2))
    eps = np.finfo(dtype).eps
    assert np.abs(xa - xs[2]).max() <= 2000 * eps * max(np.abs(xa).max(), 1e-30)


@pytest.mark.parametrize("sag", [False, True], ids=["saga", "sag"])
def test_saga_solves_together_are_bitwise_the_solves_alone(ctx, sag):
    """SAGA / SAG (SAGA.jl:44-73) for four lambdas, 3000 iterations each: the lockstep driver == the functor"""
    import ciaoalgorithms_jl_amd.operators as ops
    import ciaoalgorithms_jl_amd.solvers as S
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    dtype = np.float64
    A, b, F, Lc = _path_problem(ops, dtype, N=400, d=128, seed=9)
    N, d = A.shape
    lams = [0.1, 0.03, 0.01, 0.003]
    x0 = np.zeros(d, dtype)
    mk = lambda: S.SAGA(dtype, maxit=3000, SAG_flag=sag)
    alone = [mk()(x0, F=F, g=ops.NormL1(l), N=N, L=Lc, ctx=ctx, stream=IndexStream(10 + k)) for k, l in enumerate(lams)]
    its = [S.iterator(mk(), x0, F=F, g=ops.NormL1(l), N=N, L=Lc, ctx=ctx, stream=IndexStream(10 + k)) for k, l in enumerate(lams)]
    xs, n = S.solve_together(its, maxit=3000)
    assert n == 3000
    for k, (x, (xa, na)) in enumerate(zip(xs, alone)):
        assert na == 3000 and np.isfinite(x).all() and np.array_equal(x, xa), f"lambda #{k}"
    assert its[0]._state.ind >= 1


def test_solve_together_refuses_mixed_and_foreign_iterables(ctx):
    import ciaoalgorithms_jl_amd.operators as ops
    import ciaoalgorithms_jl_amd.solvers as S
    A, b, F, Lc = _path_problem(ops, np.float64, N=50, d=32)
    x0 = np.zeros(32)
    a = S.iterator(S.SVRG(γ=1e-3), x0, F=F, g=ops.NormL1(0.1), N=50, ctx=ctx)
    c = S.iterator(S.SAGA(γ=1e-3), x0, F=F, g=ops.NormL1(0.1), N=50, ctx=ctx)
    f = S.iterator(S.Finito(γ=1e-3, LFinito=True), x0, F=F, g=ops.NormL1(0.1), N=50, L=Lc, ctx=ctx)
    with pytest.raises(TypeError, match="one kind"):
        S.solve_together([a, c], maxit=3)
    with pytest.raises(TypeError, match="one kind"):
        S.solve_together([f], maxit=3)
    assert S.solve_together([], maxit=3) == ([], 0)


@pytest.mark.parametrize("sweeping,r", [(1, 1), (2, 1), (3, 4), (1, 3)], ids=["random-r1", "cyclic-r1", "shuffled-r4", "random-r3"])
def test_finito_solves_together_are_bitwise_the_solves_alone(ctx, sweeping, r):
    """Finito (Finito.jl:66-133; the default minibatch of one sample and small minibatches: sequential chains) for four lambdas"""
    import ciaoalgorithms_jl_amd.operators as ops
    import ciaoalgorithms_jl_amd.solvers as S
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    dtype = np.float64
    # (static batches, Finito_basic.jl:52-58, must divide N here: a shorter last batch would make a solve's steps two dependent
    #  launches, which a chain batch refuses)
    A, b, F, Lc = _path_problem(ops, dtype, N=404 if sweeping != 1 else 403, d=128, seed=11)
    N, d = A.shape
    lams = [0.1, 0.03, 0.01, 0.003]
    x0 = np.zeros(d, dtype)
    mk = lambda: S.Finito(dtype, maxit=2500, sweeping=sweeping, minibatch=(r > 1, r))
    alone = [mk()(x0, F=F, g=ops.NormL1(l), N=N, L=Lc, ctx=ctx, stream=IndexStream(20 + k)) for k, l in enumerate(lams)]
    assert "chain_dma_kernel" in ctx.last_kernel()
    its = [S.iterator(mk(), x0, F=F, g=ops.NormL1(l), N=N, L=Lc, ctx=ctx, stream=IndexStream(20 + k)) for k, l in enumerate(lams)]
    xs, n = S.solve_together(its, maxit=2500)
    assert n == 2500 and "chain batch: 4 chains in 1 launch(es)" in ctx.last_kernel()
    for k, (x, (xa, na)) in enumerate(zip(xs, alone)):
        assert na == 2500 and np.isfinite(x).all() and np.array_equal(x, xa), f"lambda #{k}"


def test_finito_steps_with_large_or_ragged_batches_are_refused_by_a_chain_batch(ctx):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd._lib import CiaoError
    from ciaoalgorithms_jl_amd.device import ProxG
    F = _problem(600, 256, np.float64)
    g = ProxG(L.PROX_L1, lam=1e-3)
    gam = torch.full((F.N,), 0.5, dtype=torch.float64, device="cuda")
    hg = ctx.hat_gamma(gam)
    x0 = torch.zeros(F.d, dtype=torch.float64, device="cuda")
    table = torch.empty((F.N, F.d), dtype=torch.float64, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    big = np.arange(0, 401, 200, dtype=np.int64)                 # two batches of 200: batch-parallel kernels, not a chain
    with pytest.raises(CiaoError, match="same size, small enough"):
        with ctx.chain_batch():
            ctx.finito_steps(F, g, gam, hg, big, np.arange(400, dtype=np.int64), table, av, z)
    ragged = np.array([0, 2, 3], dtype=np.int64)                 # a batch of 2, then one of 1: two dependent launches
    with pytest.raises(CiaoError, match="same size, small enough"):
        with ctx.chain_batch():
            ctx.finito_steps(F, g, gam, hg, ragged, np.arange(3, dtype=np.int64), table, av, z)
    with ctx.chain_batch():                                      # a single solve in a batch is fine
        ctx.finito_steps(F, g, gam, hg, np.arange(0, 41, 2, dtype=np.int64), np.arange(40, dtype=np.int64), table, av, z)
    ctx.synchronize()
    assert torch.isfinite(z).all() and "chain batch: 1 chains" in ctx.last_kernel()


def test_svrg_plus_plus_solved_together_doubles_m_like_the_solves_alone(ctx):
    """SVRG++ (SVRG.jl:31-38 plus=true: no w = z_full at the epoch's end, m doubles, SVRG_basic.jl:85,93) through the lockstep driver"""
    import ciaoalgorithms_jl_amd.operators as ops
    import ciaoalgorithms_jl_amd.solvers as S
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    dtype = np.float64
    A, b, F, Lc = _path_problem(ops, dtype, N=300, d=64, seed=21)
    N, d = A.shape
    lams = [0.05, 0.01, 0.002]
    gam = 1 / (7 * Lc.max())
    x0 = np.zeros(d, dtype)
    mk = lambda: S.SVRG(dtype, γ=gam, m=40, plus=True, maxit=6)
    ctx.set_option("svrg_cache_rowdots", 0)
    try:
        alone = [mk()(x0, F=F, g=ops.NormL1(l), N=N, ctx=ctx, stream=IndexStream(30 + k)) for k, l in enumerate(lams)]
        its = [S.iterator(mk(), x0, F=F, g=ops.NormL1(l), N=N, ctx=ctx, stream=IndexStream(30 + k)) for k, l in enumerate(lams)]
        xs, n = S.solve_together(its, maxit=6)
    finally:
        ctx.set_option("svrg_cache_rowdots", 1)
    assert n == 6 and all(it._state.m == 40 * 2 ** 5 for it in its)
    for k, (x, (xa, na)) in enumerate(zip(xs, alone)):
        assert na == 6 and np.isfinite(x).all() and np.array_equal(x, xa), f"lambda #{k}"
