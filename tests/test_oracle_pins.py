"""Pins the CPU oracle (oracle/) against the reference's OWN known-answer tests -- the oracle must pass these before
any GPU-vs-oracle parity claim means anything.

  (1) literal fixture  test/test_logistic_l1.jl:12-29 (8 x 5 data, labels, hard-coded x_star), every asserted testset
  (2) constructed known answer  test/test_lasso.jl:15-47 (generator restated in tests/problems.py), every algorithm
  (3) structural pins: solver(maxit=1) == first state; deterministic cyclic iterator == solver; eltype preserved
  (4) the invariants av == (1/N) sum grad f_i(z_full) (SVRG), av == mean(table) (SAGA), av == hg sum s_i/g_i (Finito)
"""
import numpy as np
import pytest

import problems as P
from oracle import oracle as O
from oracle import ref_solvers as RS


@pytest.fixture(scope="module")
def Stream(ciao):
    return ciao.IndexStream


def logistic(dtype=np.float64):
    A, y, L, lam, x0, x_star = P.logistic_fixture(dtype)
    return O.Problem("logistic", A, y, 1.0), O.Prox("l1", lam=lam), L, x0, x_star


def lasso(dtype):
    A, b, L, lam, x0, x_star, f_star = P.lasso_known_answer(dtype=dtype)
    N = A.shape[0]
    return O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=lam), L, x0, (lambda x: P.lasso_cost(A, b, lam, x)), f_star


# ---- (1) test_logistic_l1.jl ---------------------------------------------------------------------------------------------
class TestLogisticFixture:
    maxit, tol = 9000, 1e-4

    def test_xstar_is_the_fixed_point(self):
        """x_star satisfies x = prox_{t g}(x - t * (1/N) sum grad f_i(x)) for the restated operators."""
        p, g, L, x0, x_star = logistic()
        grad = O.full_pass(p, x_star)
        t = 0.1
        assert np.abs(O.prox(g, x_star - t * grad, t) - x_star).max() < 1e-8

    @pytest.mark.parametrize("sweeping", [1, 2, 3])
    def test_nominal_finito(self, Stream, sweeping):                        # :55-59
        p, g, L, x0, x_star = logistic()
        x, it = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, L=L, stream=Stream(0))
        assert np.abs(x - x_star).max() < self.tol and it == self.maxit

    @pytest.mark.parametrize("sweeping", [2, 3])
    def test_lfinito(self, Stream, sweeping):                               # :62-68
        p, g, L, x0, x_star = logistic()
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, lfinito=True, L=L, stream=Stream(0))
        assert np.abs(x - x_star).max() < self.tol

    @pytest.mark.parametrize("sweeping,batch", [(1, 2), (2, 2), (3, 3)])
    def test_finito_minibatch(self, Stream, sweeping, batch):               # :71-81
        p, g, L, x0, x_star = logistic()
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, batch=batch, L=L, stream=Stream(0))
        assert np.abs(x - x_star).max() < self.tol

    @pytest.mark.parametrize("sweeping,batch", [(2, 1), (2, 2), (3, 3)])
    def test_lfinito_minibatch(self, Stream, sweeping, batch):              # :84-93
        p, g, L, x0, x_star = logistic()
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, batch=batch, lfinito=True, L=L, stream=Stream(0))
        assert np.abs(x - x_star).max() < self.tol

    def test_scalar_gamma_and_L(self, Stream):                              # :96-108
        p, g, L, x0, x_star = logistic()
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, gamma=p.N / L.max(), L=L, stream=Stream(0))
        assert np.abs(x - x_star).max() < self.tol
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, L=float(L.max()), stream=Stream(0))
        assert np.abs(x - x_star).max() < self.tol

    def test_svrg_and_svrg_plus(self, Stream):                              # :125-137
        p, g, L, x0, x_star = logistic()
        gamma = 1 / (10 * L.max())
        x, _ = RS.svrg(p, g, x0, maxit=self.maxit, gamma=gamma, stream=Stream(0))
        assert np.linalg.norm(x - x_star) < self.tol
        x, it = RS.svrg(p, g, x0, maxit=16, gamma=gamma, m=p.N, plus=True, stream=Stream(0))
        assert np.linalg.norm(x - x_star) < self.tol and it == 16

    def test_saga_converges_sag_does_not_reach_tol(self, Stream):           # :158-205 (lines without @test)
        p, g, L, x0, x_star = logistic()
        x, _ = RS.saga(p, g, x0, maxit=self.maxit, L=L, stream=Stream(0))
        assert np.linalg.norm(x - x_star) < self.tol
        x, _ = RS.saga(p, g, x0, maxit=self.maxit, L=L, sag=True, stream=Stream(0))
        assert 1e-4 < np.linalg.norm(x - x_star) < 5e-2   # consistent with the reference not asserting it

    @pytest.mark.parametrize("lf", [True, False])
    def test_cyclic_iterator_equals_solver_exactly(self, Stream, lf):       # :111-122
        p, g, L, x0, x_star = logistic()
        x_f, n = RS.finito(p, g, x0, maxit=10, sweeping=2, lfinito=lf, L=L, stream=Stream(0))
        cls = RS.LFinitoIterable if lf else RS.FinitoIterable
        last = None
        for k, st in zip(range(10), cls(p, g, x0, L, None, 2, 1, 0.999, Stream(0))):
            last = st
            assert RS.solution(st) is st.z
        assert np.array_equal(RS.solution(last), x_f) and n == 10


# ---- (2) test_lasso.jl ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T", [np.float32, np.float64])
class TestLassoKnownAnswer:
    maxit, tol = 1000, 1e-4

    def test_generator_gives_the_minimiser(self, T):
        p, g, L, x0, cost, f_star = lasso(np.float64)
        A, b, _, lam, _, x_star, _ = P.lasso_known_answer()
        grad = O.full_pass(p, x_star)   # (1/N) sum N a_i (a_i'x - b_i) = A'(Ax - b)
        assert np.abs(O.prox(g, x_star - 0.05 * grad, 0.05) - x_star).max() < 1e-12
        assert abs(cost(x_star) - f_star) < 1e-12

    @pytest.mark.parametrize("sweeping", [1, 2, 3])
    def test_finito(self, Stream, T, sweeping):                             # :70-75
        p, g, L, x0, cost, f_star = lasso(T)
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, L=L, stream=Stream(0))
        assert cost(x) - f_star < self.tol and x.dtype == T

    @pytest.mark.parametrize("sweeping,batch,lf", [(2, 1, True), (3, 1, True), (1, 2, False), (2, 2, False), (3, 3, False),
                                                    (2, 2, True), (3, 3, True)])
    def test_finito_variants(self, Stream, T, sweeping, batch, lf):         # :78-125
        p, g, L, x0, cost, f_star = lasso(T)
        x, _ = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, batch=batch, lfinito=lf, L=L, stream=Stream(0))
        assert cost(x) - f_star < self.tol and x.dtype == T

    @pytest.mark.parametrize("sweeping", [1, 2, 3])
    def test_adaptive_finito(self, Stream, T, sweeping):                    # :88-98
        p, g, L, x0, cost, f_star = lasso(T)
        x, it = RS.finito(p, g, x0, maxit=self.maxit, sweeping=sweeping, adaptive=True, tol=T(1e-5), stream=Stream(0))
        assert cost(x) - f_star < self.tol and x.dtype == T and it == self.maxit

    def test_svrg(self, Stream, T):                                         # :164-176
        p, g, L, x0, cost, f_star = lasso(T)
        gamma = 1 / (7 * L.max())
        x, _ = RS.svrg(p, g, x0, maxit=self.maxit, gamma=gamma, stream=Stream(0))
        assert cost(x) - f_star < self.tol and x.dtype == T
        x, _ = RS.svrg(p, g, x0, maxit=16, gamma=gamma, m=1, plus=True, stream=Stream(0))
        assert cost(x) - f_star < self.tol and x.dtype == T

    @pytest.mark.parametrize("sag", [False, True])
    def test_saga_sag(self, Stream, T, sag):                                # :199-248
        p, g, L, x0, cost, f_star = lasso(T)
        x, _ = RS.saga(p, g, x0, maxit=10000 if sag else self.maxit, L=L, sag=sag, stream=Stream(0))
        assert cost(x) - f_star < self.tol and x.dtype == T

    def test_maxit_one_returns_the_init_state(self, Stream, T):             # :188-192, :224-228
        p, g, L, x0, cost, f_star = lasso(T)
        gamma = T(1 / (3 * L.max()))
        x, n = RS.svrg(p, g, x0, maxit=1, gamma=gamma, stream=Stream(0))
        assert np.array_equal(x, x0) and n == 1
        x, n = RS.saga(p, g, x0, maxit=1, gamma=gamma, stream=Stream(0))
        assert np.array_equal(x, O.prox(g, ((T(1) - gamma) * x0).astype(T), gamma)) and n == 1


# ---- (4) invariants ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("loss", ["ls", "logistic"])
def test_invariants(Stream, loss):
    A, b, x0 = P.synthetic(loss, 40, 17, np.float64)
    p = O.Problem(loss, A, b, 40.0 if loss == "ls" else 1.0)
    g = O.Prox("l1", lam=0.01)
    st = Stream(3)
    av, z, zf, w = O.svrg_init(p, x0)
    O.svrg_iterate(p, g, 0.01, st.rand_indices(40, 80), False, av, z, zf, w)
    assert np.allclose(av, O.full_pass(p, zf), rtol=0, atol=1e-15)
    table, av, z = O.saga_init(p, g, 0.01, x0)
    O.saga_steps(p, g, 0.01, False, st.rand_indices(40, 300), table, av, z)
    assert np.abs(av - table.mean(axis=0)).max() < 1e-13
    gam = np.linspace(0.5, 2.0, 40)
    table, av, z, hg = O.finito_init(p, g, gam, x0)
    O.finito_steps(p, g, gam, hg, [st.sample_without_replacement(40, 5) for _ in range(30)], table, av, z)
    assert np.abs(av - hg * (table / gam[:, None]).sum(axis=0)).max() < 1e-12


def test_oracle_operators_against_closed_forms():
    """O1-O4 formulas checked against numpy one-liners and finite differences."""
    rng = np.random.default_rng(0)
    a, x = rng.standard_normal(9), rng.standard_normal(9)
    gy, f = O.gradient(O.LOSS_LS, a, 0.3, 2.5, x)
    assert np.allclose(gy, 2.5 * (a @ x - 0.3) * a) and np.isclose(f, 1.25 * (a @ x - 0.3) ** 2)
    for yv in (1.0, -1.0):
        gy, f = O.gradient(O.LOSS_LOGISTIC, a, yv, 1.0, x)
        assert np.allclose(gy, -yv * a / (1 + np.exp(yv * (a @ x)))) and np.isclose(f, np.log1p(np.exp(-yv * (a @ x))))
        eps = 1e-6
        fd = [(O.gradient(O.LOSS_LOGISTIC, a, yv, 1.0, x + eps * e)[1] - O.gradient(O.LOSS_LOGISTIC, a, yv, 1.0, x - eps * e)[1])
              / (2 * eps) for e in np.eye(9)]
        assert np.allclose(gy, fd, atol=1e-8)
    v = np.array([-2.0, -0.5, 0.0, 0.4, 3.0])
    assert np.array_equal(O.prox(O.Prox("l1", lam=2.0), v, 0.25), np.sign(v) * np.maximum(np.abs(v) - 0.5, 0))
    assert np.array_equal(O.prox(O.Prox("zero"), v, 0.25), v)
    assert np.array_equal(O.prox(O.Prox("box", lo=-1.0, hi=0.5), v, 9.0), np.clip(v, -1.0, 0.5))
    lo, hi = np.full(5, -0.1), np.array([0.0, 0.1, 0.2, 0.3, 0.4])
    assert np.array_equal(O.prox(O.Prox("box", lo=lo, hi=hi), v, 9.0), np.clip(v, lo, hi))


# ---- test_sharing.jl (ProShI) -------------------------------------------------------------------
class TestSharingFixture:
    maxit, tol = 1000, 1e-4

    def _setup(self):
        Q, q, eta, lo, hi, L, g_hi, x0, sum_star = P.sharing_fixture()
        return O.SepQuad(Q, q, eta, lo, hi), O.Prox("box", lo=-np.inf, hi=g_hi), L, x0, sum_star

    @pytest.mark.parametrize("sweeping", [1, 2, 3])
    def test_basic_proshi(self, Stream, sweeping):                          # test_sharing.jl:41-46
        f, g, L, x0, sum_star = self._setup()
        x, it = RS.proshi(f, g, x0, maxit=self.maxit, sweeping=sweeping, L=L, stream=Stream(0))
        assert np.abs(x.sum(axis=0) - sum_star).max() < self.tol and x.shape == (3, 2) and x.dtype == np.float64

    @pytest.mark.parametrize("sweeping,batch", [(1, 2), (2, 2), (3, 3)])
    def test_proshi_minibatch(self, Stream, sweeping, batch):               # :49-59
        f, g, L, x0, sum_star = self._setup()
        x, it = RS.proshi(f, g, x0, maxit=self.maxit, sweeping=sweeping, batch=batch, L=L, stream=Stream(0))
        assert np.abs(x.sum(axis=0) - sum_star).max() < self.tol

    def test_scalar_gamma_and_L(self, Stream):                              # :62-74
        f, g, L, x0, sum_star = self._setup()
        x, _ = RS.proshi(f, g, x0, maxit=self.maxit, gamma=f.N / L.max(), L=L, stream=Stream(0))
        assert np.abs(x.sum(axis=0) - sum_star).max() < self.tol
        x, _ = RS.proshi(f, g, x0, maxit=self.maxit, L=float(L.max()), stream=Stream(0))
        assert np.abs(x.sum(axis=0) - sum_star).max() < self.tol


# ---- (5) Julia's `sum` over a Vector: Base.mapreduce_impl (pairwise above 1024 elements) -------------------------------------
def _julia_mapreduce(vals, ifirst, ilast, blksize=1024):
    """Line-by-line restatement of Base.mapreduce_impl(identity, +, A, ifirst, ilast, blksize) of Julia 1.0-1.6
    (base/reduce.jl), 1-based inclusive indices; `vals[i-1]` is a numpy scalar or a numpy vector (so + rounds in R)."""
    if ifirst == ilast:
        return vals[ifirst - 1]
    if ifirst + blksize > ilast:                       # sequential portion
        v = vals[ifirst - 1] + vals[ifirst]
        for i in range(ifirst + 2, ilast + 1):
            v = v + vals[i - 1]
        return v
    imid = (ifirst + ilast) >> 1                       # pairwise portion
    return _julia_mapreduce(vals, ifirst, imid, blksize) + _julia_mapreduce(vals, imid + 1, ilast, blksize)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N", [1, 2, 15, 16, 1023, 1024, 1025, 2048, 2049, 5000])
def test_julia_sum_rule(dtype, N):
    """The six init sums of the reference (SAGA_basic.jl:47; Finito_basic.jl:82-83; Finito_LFinito.jl:66;
    Finito_adaptive.jl:91-92; ProShI_basic.jl:82-83) are `sum` over a Vector: a left fold only up to 1024 elements.  The C
    oracle's julia_sum_* must reproduce the rule bit for bit (sums of d-vectors; scalar sums with left-to-right leaves)."""
    rng = np.random.default_rng(N)
    rows = rng.standard_normal((N, 7)).astype(dtype)
    gam = (0.5 + rng.random(N)).astype(dtype)
    want = _julia_mapreduce([rows[i] for i in range(N)], 1, N)
    assert np.array_equal(O.julia_sum_vec(rows), want)
    want = _julia_mapreduce([rows[i] / gam[i] for i in range(N)], 1, N)
    assert np.array_equal(O.julia_sum_vec(rows, gam), want)
    assert O.julia_sum_scalar(gam) == _julia_mapreduce(list(gam), 1, N)
    assert O.julia_sum_scalar(gam, inv=True) == _julia_mapreduce([dtype(1) / g for g in gam], 1, N)
    if N >= 1025:   # and it is NOT the left fold any more
        left = rows[0].copy()
        for i in range(1, N):
            left = left + rows[i]
        assert not np.array_equal(left, O.julia_sum_vec(rows))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_init_passes_use_the_julia_sum(dtype):
    """SAGA / Finito / LFinito / ProShI / adaptive-Finito init at N = 2049: av and hat_γ are the pairwise sums, bit for bit."""
    N, d = 2049, 5
    A, b, x0 = P.synthetic("ls", N, d, dtype, seed=3)
    p, g = O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=0.01)
    table, av, z = O.saga_init(p, g, dtype(0.1), x0)
    assert np.array_equal(av, O.julia_sum_vec(table) / dtype(N))
    gam = (0.5 + np.random.default_rng(0).random(N)).astype(dtype)
    table, av, z, hg = O.finito_init(p, g, gam, x0)
    hg_want = dtype(1) / O.julia_sum_scalar(gam, inv=True)
    assert hg == hg_want and np.array_equal(av, hg_want * O.julia_sum_vec(table, gam))
    av, z, zf, hg = O.lfinito_init(p, gam, x0)
    assert hg == hg_want
    f = O.SepQuad(np.abs(A) + dtype(0.1), A.copy(), eta=1.0, lo=-2.0, hi=2.0)
    table, av, z, hg = O.proshi_init(f, O.Prox("box", lo=-np.inf, hi=1.0, dtype=dtype), gam, x0)
    assert hg == O.julia_sum_scalar(gam) and np.array_equal(av, O.julia_sum_vec(table))


def test_adaptive_finito_keeps_julias_float64_promotions():
    """R = Float32: `L_int = zeros(N)` is a Float64 array (Finito_adaptive.jl:73), so γ_i = Float32(Float64(α) / L_i); `γ *= 0.8`
    is a Float64 product rounded back (:136) -- Float32(γ) * Float32(0.8) differs from it in the last place for some γ."""
    N, d = 40, 6
    A, b, x0 = P.synthetic("ls", N, d, np.float32, seed=8)
    p, g = O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=0.01)
    table, gtable, gam, fi_x, av, z, hg = O.afinito_init(p, g, np.float32(0.999), x0)
    for i in range(N):
        ge = O.gradient(p.loss, A[i], b[i], float(N), (x0 + np.float32(1)).astype(np.float32))[0]
        nmg = np.sqrt(np.float32(np.sum((ge - gtable[i]) ** 2, dtype=np.float32)))
        want = np.float32(np.float64(np.float32(0.999)) / (np.float64(nmg) / np.sqrt(np.float64(d)) / np.float64(N)))
        assert abs(float(gam[i]) - float(want)) <= 2 * np.spacing(want), (i, gam[i], want)   # nmg's own summation order aside
    g32 = np.float32(0.3)
    differ = [g for g in (np.float32(0.1) * k for k in range(1, 200)) if np.float32(np.float64(g) * 0.8) != g * np.float32(0.8)]
    assert differ, "the two roundings must be distinguishable, or the promotion would not matter"


# ---- (6) complex T (CIAOAlgorithms.jl:3; test_lasso.jl:3 runs ComplexF32 / ComplexF64) --------------------------------------
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_complex_operators_against_numpy_complex_arithmetic(dtype):
    """LeastSquares with a complex row (grad = lam * conj(a) * (a.x - b), value lam/2 |res|^2) and the complex NormL1 prox
    (sign(x) max(|x| - gl, 0)) on (re, im) pairs, against numpy's complex arithmetic."""
    A, b, x = P.synthetic_complex(7, 33, dtype, seed=4)
    lam = 7.0
    tol = 50 * np.finfo(A.real.dtype).eps
    for i in (0, 3, 6):
        y, f = O.gradient(O.LOSS_LS_COMPLEX, O.as_pairs(A[i]), O.as_pairs(b[i:i + 1]), lam, O.as_pairs(x))
        res = A[i].astype(np.complex128) @ x.astype(np.complex128) - complex(b[i])
        want = lam * np.conj(A[i].astype(np.complex128)) * res
        assert np.abs(O.as_complex(y) - want).max() <= tol * np.abs(want).max()
        assert abs(f - lam / 2 * abs(res) ** 2) <= tol * abs(f)
    g = O.Prox("l1_complex", lam=0.3)
    y = O.as_complex(O.prox(g, O.as_pairs(x), x.real.dtype.type(0.5)))
    x128 = x.astype(np.complex128)
    want = x128 / np.abs(x128) * np.maximum(np.abs(x128) - 0.15, 0)
    assert np.abs(y - want).max() <= tol and np.count_nonzero(y == 0) > 0
    p = O.Problem("ls", A, b, lam)
    assert p.loss == O.LOSS_LS_COMPLEX and p.d == 66
    obj = O.objective(p, g, O.as_pairs(x))
    want = 0.5 * lam * np.mean(np.abs(A.astype(np.complex128) @ x128 - b) ** 2) + 0.3 * np.abs(x128).sum()
    assert abs(obj - want) <= 1e3 * tol * abs(want)


@pytest.mark.parametrize("ctype", [np.complex128, np.complex64])
def test_reference_complex_lasso_is_the_real_lasso_in_complex_containers(ctype, Stream):
    """test_lasso.jl for T = ComplexF32 / ComplexF64 builds REAL data in complex arrays (rand(R, ...), zeros(T, n)).  With zero
    imaginary parts every complex operation of the restatement reduces to the real one exactly, so the complex run must
    reproduce the real run BIT FOR BIT (real parts) with imaginary parts exactly zero -- for every algorithm -- and meet the
    reference's own assertion (cost gap < 1e-4, eltype preserved)."""
    rtype = np.zeros(1, ctype).real.dtype.type
    A, b, L, lam, x0, x_star, f_star = P.lasso_known_answer(dtype=rtype)
    N = A.shape[0]
    Ac, bc, Lc, _, x0c, _, _ = P.lasso_known_answer(dtype=ctype)
    pr, gr = O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=lam)
    pc, gc = O.Problem("ls", Ac, bc, float(N)), O.Prox("l1_complex", lam=lam)
    runs = [("svrg", lambda p, g, x: RS.svrg(p, g, x, maxit=300, gamma=1 / (7 * L.max()), stream=Stream(0))),
            ("saga", lambda p, g, x: RS.saga(p, g, x, maxit=1000, L=L, stream=Stream(0))),
            ("finito", lambda p, g, x: RS.finito(p, g, x, maxit=1000, sweeping=2, L=L, stream=Stream(0))),
            ("lfinito", lambda p, g, x: RS.finito(p, g, x, maxit=300, sweeping=3, lfinito=True, batch=2, L=L, stream=Stream(0))),
            # adaptive: x0 .+ one(R) touches the real parts only and sqrt(length(x0)) counts complex entries (Finito_adaptive.jl:74,86)
            ("adaptive", lambda p, g, x: RS.finito(p, g, x, maxit=1000, sweeping=2, adaptive=True, tol=1e-5, L=L, stream=Stream(0)))]
    for name, run in runs:
        xr, _ = run(pr, gr, x0)
        xc, _ = run(pc, gc, O.as_pairs(x0c))
        xc = O.as_complex(xc)
        assert xc.dtype == ctype
        assert np.array_equal(xc.real, xr) and not np.any(xc.imag), name
        assert P.lasso_cost(Ac, bc, lam, xc) - f_star < 1e-4, name
