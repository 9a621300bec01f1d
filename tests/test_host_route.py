"""The explicit host route (`fallback="host"` / `backend="host"`, ciaoalgorithms.jl_amd/host_route.py): operator objects the
device path cannot pack still solve -- on the host, announced, labelled.
These tests need no GPU: an unpackable problem with fallback="host" must not touch one.

What the reference accepts and the device path does not: any ProximalOperators object (SVRG.jl:46-58, Finito.jl:66-116), e.g.
test/test_sharing.jl:16-25 builds Sum(Quadratic, SqrDistL2)."""
import warnings

import numpy as np
import pytest

import problems as P


@pytest.fixture(scope="module")
def pkg():
    import ciao_loader
    return ciao_loader.load()


def lasso(N=6, n=3):
    A, b, Lc, lam, x0, x_star, f_star = P.lasso_known_answer(N=N, n=n, p=2, seed=0)
    return A, b, Lc, lam, x0, f_star


class SmoothAbs:
    """A user-defined smooth term, not an operator class of this package: f(x) = sum sqrt(1 + (a.x - b)^2) (pseudo-Huber),
    with the ProximalOperators calling convention gradient(x) -> (grad, value)."""

    def __init__(self, a, b):
        self.a, self.b = np.asarray(a, dtype=np.float64), float(b)

    def gradient(self, x):
        r = self.a @ x - self.b
        s = np.sqrt(1 + r * r)
        return (r / s) * self.a, s


class ElasticNet:
    """A user-defined g (no device form): g(x) = l1 |x|_1 + l2/2 |x|^2 with prox(x, gamma) -> (y, g(y))."""

    def __init__(self, l1, l2):
        self.l1, self.l2 = l1, l2

    def prox(self, x, gamma):
        y = np.sign(x) * np.maximum(np.abs(x) - gamma * self.l1, 0) / (1 + gamma * self.l2)
        return y, self.value(y)

    def value(self, x):
        return self.l1 * np.sum(np.abs(x)) + 0.5 * self.l2 * np.dot(x, x)


def prox_gradient_reference(F, g, x0, L, iters=20000):
    """A plain proximal-gradient solve of (1/N) sum f_i + g to high accuracy with the same operator arithmetic: the answer the
    stochastic solvers must approach."""
    from ciaoalgorithms_jl_amd import host_ops as H
    x, N, t = x0.copy(), len(F), 1.0 / L
    for _ in range(iters):
        grad = sum(H.gradient(f, x)[0] for f in F) / N
        x = H.prox(g, x - t * grad, t)[0]
    return x


def test_fallback_is_explicit_announced_and_labelled(pkg):
    from ciaoalgorithms_jl_amd import solvers as S, operators as Op
    rng = np.random.default_rng(0)
    N, d = 5, 4
    F = [SmoothAbs(rng.standard_normal(d), rng.standard_normal()) for _ in range(N)]
    x0 = np.zeros(d)
    # (1) without the keyword: the same refusal as before, now a subclass of TypeError -- and before any device is touched
    with pytest.raises(Op.UnpackableOperator):
        S.SVRG(np.float64, γ=0.1, maxit=3)(x0, F=F, g=Op.NormL1(0.1), N=N)
    with pytest.raises(TypeError):
        S.iterator(S.SAGA(np.float64, γ=0.1), x0, F=F, g=Op.NormL1(0.1), N=N)
    # a g without a device form is refused the same way
    with pytest.raises(Op.UnpackableOperator):
        S.Finito(np.float64, maxit=3)(x0, F=[Op.LeastSquares(np.ones((1, d)), [1.0])] * N, g=ElasticNet(0.1, 0.1), L=np.ones(N), N=N)
    # (2) with it: a RuntimeWarning that says HOST, states labelled, the reference's identities kept
    with pytest.warns(RuntimeWarning, match="HOST route"):
        it = S.iterator(S.SAGA(np.float64, γ=0.05), x0, F=F, g=Op.NormL1(0.1), N=N, fallback="host")
    assert it.backend == "host" and it.x0 is x0                       # iter.x0 === x0, test/test_lasso.jl:182
    st = next(iter(it))
    assert st.backend == "host" and S.solution(st) is st.z            # test/test_lasso.jl:185
    # (3) other errors are NOT swallowed by the fallback
    with pytest.raises(ValueError):
        S.SVRG(np.float64, γ=0.1)(x0, F=F[:-1], g=None, N=N, fallback="host")
    with pytest.raises(ValueError):
        S.SVRG(np.float64, γ=0.1)(x0, F=F, g=None, N=N, fallback="cpu")
    # (4) maxit = 1 returns the init state's solution (test_lasso.jl:188-192): x0 for SVRG, prox((1 - γ) x0) for SAGA
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        x, n = S.SVRG(np.float64, γ=0.1, maxit=1)(np.ones(d), F=F, g=Op.NormL1(0.1), N=N, fallback="host")
        assert n == 1 and np.array_equal(x, np.ones(d))
        x, n = S.SAGA(np.float64, γ=0.1, maxit=1)(np.ones(d), F=F, g=Op.NormL1(0.1), N=N, fallback="host")
        assert n == 1 and np.allclose(x, 0.9 - 0.01)                  # soft threshold of 0.9 at γ λ = 0.01


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
def test_reference_lasso_testsets_through_the_host_route(pkg):
    """test/test_lasso.jl on the host route (backend="host" forces it for this packable problem): cost gap < 1e-4."""
    from ciaoalgorithms_jl_amd import solvers as S, operators as Op
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    A, b, Lc, lam, x0, f_star = lasso()
    N = A.shape[0]
    F = [Op.LeastSquares(A[i:i + 1], b[i:i + 1], float(N)) for i in range(N)]
    g = Op.NormL1(lam)
    kw = dict(F=F, g=g, N=N, backend="host")
    runs = [S.SVRG(np.float64, γ=1 / (7 * Lc.max()), maxit=1000)(x0, stream=IndexStream(1), **kw),
            S.SVRG(np.float64, γ=1 / (7 * Lc.max()), maxit=12, m=1, plus=True)(x0, stream=IndexStream(1), **kw),
            S.SAGA(np.float64, maxit=3000)(x0, L=Lc, stream=IndexStream(2), **kw),
            S.SAG(np.float64, maxit=10000)(x0, L=Lc, stream=IndexStream(3), **kw)]
    for sweeping, batch in ((1, 1), (2, 1), (3, 1), (1, 2), (2, 2), (3, 3)):
        runs.append(S.Finito(np.float64, sweeping=sweeping, minibatch=(True, batch), maxit=1000)(x0, L=Lc, stream=IndexStream(4), **kw))
    for sweeping, batch in ((2, 1), (3, 1), (2, 2), (3, 3)):
        runs.append(S.Finito(np.float64, LFinito=True, sweeping=sweeping, minibatch=(True, batch), maxit=400)(x0, L=Lc, stream=IndexStream(5), **kw))
    for k, (x, _) in enumerate(runs):
        assert x.dtype == np.float64 and P.lasso_cost(A, b, lam, x) - f_star < 1e-4, f"run {k}"
    # Float32 stays Float32 (test_lasso.jl:74)
    F32 = [Op.LeastSquares(A[i:i + 1].astype(np.float32), b[i:i + 1].astype(np.float32), float(N)) for i in range(N)]
    x, _ = S.SAGA(np.float32, maxit=3000)(x0.astype(np.float32), F=F32, g=g, L=Lc, N=N, backend="host", stream=IndexStream(2))
    assert x.dtype == np.float32 and P.lasso_cost(A, b, lam, x.astype(np.float64)) - f_star < 1e-4


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
def test_unpackable_problems_solve_through_svrg_saga_finito(pkg):
    """(a) the operator family of test/test_sharing.jl:16-25 with a DENSE Quadratic, Sum(Quadratic(Q_i, q_i), SqrDistL2(IndBox, η)),
    as the f_i of a finite sum with a non-box g (NormL1); (b) user-defined smooth f_i with a user-defined g.  Every solver's
    result is held against a deterministic proximal-gradient solve of the same problem."""
    from ciaoalgorithms_jl_amd import solvers as S, operators as Op
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    rng = np.random.default_rng(3)
    N, d = 6, 4
    # (a)
    F, Ls = [], []
    for _ in range(N):
        M = rng.standard_normal((d, d))
        Q = M @ M.T / d + 0.5 * np.eye(d)
        F.append(Op.Sum(Op.Quadratic(Q, rng.standard_normal(d)), Op.SqrDistL2(Op.IndBox(-0.3, 0.3), 2.0)))
        Ls.append(np.linalg.norm(Q, 2) + 2.0)
    Ls = np.array(Ls)
    g = Op.NormL1(0.05)
    x0 = np.zeros(d)
    ref = prox_gradient_reference(F, g, x0, Ls.max())
    kw = dict(F=F, g=g, N=N, fallback="host")
    sols = {"svrg": S.SVRG(np.float64, γ=1 / (7 * Ls.max()), maxit=400)(x0, stream=IndexStream(1), **kw)[0],
            "saga": S.SAGA(np.float64, maxit=6000)(x0, L=Ls, stream=IndexStream(2), **kw)[0],
            "finito": S.Finito(np.float64, sweeping=2, maxit=3000)(x0, L=Ls, stream=IndexStream(3), **kw)[0],
            "lfinito": S.Finito(np.float64, LFinito=True, sweeping=3, maxit=600)(x0, L=Ls, stream=IndexStream(4), **kw)[0]}
    for name, x in sols.items():
        assert np.abs(x - ref).max() < 1e-4, (name, np.abs(x - ref).max())
    # (b) pseudo-Huber terms + elastic net: L_i = |a_i|^2
    F = [SmoothAbs(rng.standard_normal(d), rng.standard_normal()) for _ in range(N)]
    Ls = np.array([np.dot(f.a, f.a) for f in F])
    g = ElasticNet(0.05, 0.2)
    ref = prox_gradient_reference(F, g, x0, Ls.max())
    kw = dict(F=F, g=g, N=N, fallback="host")
    sols = {"svrg": S.SVRG(np.float64, γ=1 / (7 * Ls.max()), maxit=300)(x0, stream=IndexStream(1), **kw)[0],
            "saga": S.SAGA(np.float64, maxit=5000)(x0, L=Ls, stream=IndexStream(2), **kw)[0],
            "finito": S.Finito(np.float64, sweeping=1, minibatch=(True, 2), maxit=3000)(x0, L=Ls, stream=IndexStream(3), **kw)[0]}
    for name, x in sols.items():
        assert np.abs(x - ref).max() < 1e-4, (name, np.abs(x - ref).max())
    # the objective monitor / stop callback work on the host route too
    seen = []
    x, n = S.SAGA(np.float64, maxit=5000)(x0, L=Ls, stream=IndexStream(2), stop=lambda st: seen.append(st.objective) or len(seen) >= 3,
                                          check_every=100, **kw)
    assert n == 300 and len(seen) == 3 and seen[2] <= seen[0]


def test_host_route_stays_out_of_the_product_path_and_the_oracle(pkg):
    """No file of the host route mentions the oracle, and neither host module is imported by the device modules' hot path
    (solvers.py only reaches it through _route)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("host_route.py", "host_ops.py"):
        src = open(os.path.join(root, "ciaoalgorithms.jl_amd", name)).read()
        assert "import oracle" not in src and "from oracle" not in src and "ciao_oracle" not in src
    dev_src = open(os.path.join(root, "ciaoalgorithms.jl_amd", "device.py")).read()
    assert "host_route" not in dev_src and "host_ops" not in dev_src
