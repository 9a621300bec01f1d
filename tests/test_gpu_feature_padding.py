"""Feature padding in the host mirror (solvers._Iterable): a problem packed from HOST operators whose rows are not a whole number of
16-byte chunks (d = 1001 Float64, d = 50 Float32 ...) is packed with a few zero columns, so that the chains run the LDS-DMA kernels
instead of the register ring (2-3.7x per update).  The caller must not notice: shapes, identities and aliasing of the reference's
protocol (`iter.x0 === x0`, `solution(state) === state.z`, vectors of length d), and results equal to the unpadded run to rounding."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(ciao, ctx):
    import ciaoalgorithms_jl_amd.operators as ops
    import ciaoalgorithms_jl_amd.solvers as S
    return S, ops


def _lasso(ops, T, N, d, seed):
    A, b, x0 = P.synthetic("ls", N, d, T, seed=seed)
    F = [ops.LeastSquares(A[i:i + 1], b[i:i + 1], float(N)) for i in range(N)]
    L = float(N) * np.sum(A.astype(np.float64) ** 2, axis=1)
    return F, A, L, np.zeros(d, dtype=T)


@pytest.mark.parametrize("T,d", [(np.float64, 1001), (np.float32, 50), (np.float64, 51), (np.float32, 2049)])
def test_padded_problem_looks_and_solves_like_the_unpadded_one(api, ctx, ciao, T, d):
    S, ops = api
    N = 60
    F, A, L, x0 = _lasso(ops, T, N, d, seed=d)
    g = ops.NormL1(0.01)
    gamma = 1.0 / (7 * float(L.max()))
    vec = 16 // np.dtype(T).itemsize
    dp = -(-d // vec) * vec
    results = {}
    for pad in (True, False):
        S.PAD_FEATURES = pad
        try:
            for name, solver, kw in (("svrg", S.SVRG(T, γ=gamma, maxit=4), {}), ("saga", S.SAGA(T, γ=gamma / 3, maxit=200), {}),
                                     ("finito", S.Finito(T, maxit=150), {"L": L}), ("lfinito", S.Finito(T, maxit=4, LFinito=True, minibatch=(True, 7)), {"L": L})):
                it = S.iterator(solver, x0, F=F, g=g, N=N, ctx=ctx, stream=ciao.IndexStream(3), **kw)
                assert it.x0 is x0 and it.d == d and it.dp == (dp if pad else d) and it.F.d == it.dp
                state = None
                for k, state in zip(range(solver.maxit), it):
                    pass
                x = S.solution(state)
                assert x is (state.z_full if name == "svrg" else state.z) and tuple(x.shape) == (d,) and x.dtype == it.R
                assert tuple(state.av.shape) == (d,) and tuple(state.z.shape) == (d,)
                if pad and name in ("svrg", "saga", "finito"):
                    assert "chain_dma_kernel" in ctx.last_kernel() or "chain_ws_kernel" in ctx.last_kernel() or "rows_" in ctx.last_kernel(), ctx.last_kernel()
                results[(name, pad)] = x.cpu().numpy().copy()
                xs, iters = solver(x0, F=F, g=g, N=N, ctx=ctx, stream=ciao.IndexStream(3), **kw)       # the functor: a host vector of length d
                assert isinstance(xs, np.ndarray) and xs.shape == (d,) and xs.dtype == T and iters == solver.maxit
                assert np.array_equal(xs, results[(name, pad)])
        finally:
            S.PAD_FEATURES = True
    # BOTH runs against the ORACLE's iterables on the same index streams (the padded path used to be held against
    # the unpadded device path only) -- Float32 also against the oracle run in Float64 on the same Float32 data (form (i) of
    # test_gpu_parity.close: the accuracy statement, inside BASELINE's 1e-4 = 840 eps32)
    from oracle import oracle as O
    from oracle import ref_solvers as RS
    from test_gpu_parity import close
    for wide_oracle in ((False, True) if T == np.float32 else (False,)):
        OT = np.float64 if wide_oracle else T
        op, og = O.Problem("ls", A.astype(OT), np.concatenate([f.b for f in F]).astype(OT), float(N)), O.Prox("l1", lam=0.01)
        refs = {"svrg": RS.svrg(op, og, x0.astype(OT), maxit=4, gamma=OT(T(gamma)), stream=ciao.IndexStream(3))[0],
                "saga": RS.saga(op, og, x0.astype(OT), maxit=200, gamma=OT(T(gamma / 3)), stream=ciao.IndexStream(3))[0],
                "finito": RS.finito(op, og, x0.astype(OT), maxit=150, L=L.astype(OT), stream=ciao.IndexStream(3))[0],
                "lfinito": RS.finito(op, og, x0.astype(OT), maxit=4, lfinito=True, sweeping=1, batch=7, L=L.astype(OT), stream=ciao.IndexStream(3))[0]}
        for name, ref in refs.items():
            for pad in (True, False):
                if wide_oracle:
                    close(results[(name, pad)], ref.astype(T), T, scale={32: 180}, what=f"{name} d={d} padded={pad}", ref64=ref, scale64=180)
                else:
                    close(results[(name, pad)], ref, T, scale={64: 110, 32: 100}, what=f"{name} d={d} padded={pad} vs the oracle's iterable")


def test_padding_is_not_applied_where_it_must_not_be(api, ctx, ciao):
    """A device matrix the caller laid out (PackedF), complex problems and adaptive Finito keep their d; per-coordinate box bounds are
    padded with (-inf, inf) and the padding coordinates stay out of the result."""
    import torch
    from ciaoalgorithms_jl_amd.device import PackedF
    S, ops = api
    T, N, d = np.float64, 40, 33
    F, A, L, x0 = _lasso(ops, T, N, d, seed=9)
    packed = PackedF.least_squares(torch.from_numpy(A).cuda(), torch.from_numpy(np.zeros(N)).cuda(), float(N))
    it = S.iterator(S.SVRG(T, γ=1e-3), x0, F=packed, g=ops.NormL1(0.1), N=N, ctx=ctx)
    assert it.dp == d and it.F is packed
    it = S.iterator(S.Finito(T, adaptive=True), x0, F=F, g=ops.NormL1(0.1), L=L, N=N, ctx=ctx)
    assert it.dp == d
    lo, hi = -np.arange(1, d + 1, dtype=T), np.arange(1, d + 1, dtype=T) * 0.5
    it = S.iterator(S.SAGA(T, γ=1.0 / (3 * float(L.max()))), np.full(d, 100.0), F=F, g=ops.IndBox(lo, hi), N=N, ctx=ctx, stream=ciao.IndexStream(1))
    assert it.dp == 34
    st = None
    for k, st in zip(range(50), it):
        pass
    z = S.solution(st).cpu().numpy()
    assert z.shape == (d,) and np.all(z <= hi + 1e-12) and np.all(z >= lo - 1e-12)


def test_a_short_view_is_refused_unless_the_host_mirror_padded_the_problem(api, ctx, ciao):
    """the entry points accept a length-d view of a longer buffer for the padded problem's d'-vector ONLY when the host
    mirror padded the problem itself (PackedF.padded_from); a caller's `big[:d-1]` on a matrix it laid out is an error, not a write past
    the end of the view."""
    import torch
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    T, N, d = np.float64, 24, 34
    A, b, _ = P.synthetic("ls", N, d, T, seed=2)
    packed = PackedF.least_squares(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), float(N))
    assert packed.padded_from is None
    big = torch.zeros(d + 8, dtype=torch.float64, device="cuda")
    x, av = torch.zeros(d, dtype=torch.float64, device="cuda"), torch.zeros(d, dtype=torch.float64, device="cuda")
    ctx.full_gradient(packed, x, av)
    with pytest.raises(ValueError, match="length 34"):
        ctx.full_gradient(packed, x, big[:d - 1])
    with pytest.raises(ValueError, match="length 34"):
        ctx.full_gradient(packed, big[:d - 1], av)
    assert float(big.abs().max()) == 0.0
    S, ops = api
    F = [ops.LeastSquares(A[i:i + 1, :d - 1].copy(), b[i:i + 1], float(N)) for i in range(N)]      # d - 1 = 33 columns: padded to 34
    it = S.iterator(S.SVRG(T, γ=1e-3), np.zeros(d - 1), F=F, g=ops.NormL1(0.1), N=N, ctx=ctx)
    assert it.F.padded_from == d - 1 and it.F.d == d
    state = next(iter(it))
    assert tuple(state.av.shape) == (d - 1,)
