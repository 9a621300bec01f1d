"""Test problems: the reference's literal l1-logistic fixture and its lasso known-answer generator, plus synthetic
problems for parity tests.  Data only -- no algorithm code."""
import numpy as np

# ---- test/test_logistic_l1.jl:12-29 (literal) ------------------------------------------------------------------------
LOGISTIC_XS = np.array([
    [5.1, 3.5, 1.4, 0.2, 1.0],
    [4.9, 3.0, 1.4, 0.2, 1.0],
    [4.7, 3.2, 1.3, 0.2, 1.0],
    [4.6, 3.1, 1.5, 0.2, 1.0],
    [5.7, 3.0, 4.2, 1.2, 1.0],
    [5.7, 2.9, 4.2, 1.3, 1.0],
    [6.2, 2.9, 4.3, 1.3, 1.0],
    [5.1, 2.5, 3.0, 1.1, 1.0],
])
LOGISTIC_YS = np.array([1.0] * 4 + [-1.0] * 4)
LOGISTIC_XSTAR = np.array([0.0, 0.924160995722576, -1.1343956493097298, 0.0, 0.0])


def logistic_fixture(dtype=np.float64):
    """(A, y, L, lam_g, x0, x_star): N=8, n=5, L_i = 0.25 ||a_i||^2 (:39), g = NormL1(1/N) (:44), x0 = ones (:46)."""
    A = LOGISTIC_XS.astype(dtype)
    y = LOGISTIC_YS.astype(dtype)
    L = (0.25 * np.sum(LOGISTIC_XS ** 2, axis=1)).astype(dtype)
    return A, y, L, 1.0 / A.shape[0], np.ones(A.shape[1], dtype), LOGISTIC_XSTAR.copy()


def lasso_known_answer(N=6, n=3, p=2, seed=0, dtype=np.float64, rho=10.0, lam=1.0):
    """The generator of test/test_lasso.jl:15-47 with our own RNG (the construction is RNG-agnostic): builds A, b such
    that the chosen sparse x_star is the exact minimiser of 1/2 ||Ax-b||^2 + lam ||x||_1.
    Returns (A, b, L, lam, x0, x_star, f_star) with f_i = LeastSquares(A[i], b[i], N), L_i = N ||a_i||^2 (:52-56)."""
    rng = np.random.default_rng(seed)
    y_star = rng.random(N)
    y_star /= np.linalg.norm(y_star)
    Cm = rng.random((N, n)) * 2 - 1
    CTy = np.abs(Cm.T @ y_star)
    perm = np.argsort(-CTy, kind="stable")
    alpha = np.zeros(n)
    for i in range(n):
        if i < p:
            alpha[perm[i]] = lam / CTy[perm[i]]
        else:
            alpha[perm[i]] = lam if CTy[perm[i]] < 0.1 * lam else lam * rng.random() / CTy[perm[i]]
    A = Cm * alpha[None, :]
    x_star = np.zeros(n)
    for i in range(p):
        x_star[perm[i]] = rng.random() * rho / np.sqrt(p) * np.sign(A[:, perm[i]] @ y_star)
    b = A @ x_star + y_star
    f_star = 0.5 * np.linalg.norm(A @ x_star - b) ** 2 + lam * np.abs(x_star).sum()
    L = N * np.sum(A ** 2, axis=1)
    if np.dtype(dtype).kind == "c":   # T = ComplexF32 / ComplexF64 (test_lasso.jl:3): the same REAL data in complex containers,
        R = np.zeros(1, dtype).real.dtype     # exactly as the reference builds it (rand(R, ...), zeros(T, n)); L stays real
        return (A.astype(dtype), b.astype(dtype), L.astype(R), lam, np.zeros(n, dtype), x_star.astype(dtype), f_star)
    return (A.astype(dtype), b.astype(dtype), L.astype(dtype), lam, np.zeros(n, dtype), x_star, f_star)


def lasso_cost(A, b, lam, x):
    """cost_lasso of test/test_lasso.jl:45 (evaluated in float64 / complex128)."""
    wide = np.complex128 if any(np.iscomplexobj(v) for v in (A, b, x)) else np.float64
    A, b, x = (np.asarray(v, wide) for v in (A, b, x))
    return float(0.5 * np.linalg.norm(A @ x - b) ** 2 + lam * np.abs(x).sum())


def synthetic_complex(N, n, dtype=np.complex128, seed=1):
    """A genuinely complex least-squares problem: A (N x n), b (N), x (n) with independent real and imaginary parts."""
    rng = np.random.default_rng(seed)
    A = ((rng.standard_normal((N, n)) + 1j * rng.standard_normal((N, n))) / np.sqrt(2 * n)).astype(dtype)
    x_true = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * (rng.random(n) < 0.2)
    b = (A.astype(np.complex128) @ x_true + 0.01 * (rng.standard_normal(N) + 1j * rng.standard_normal(N))).astype(dtype)
    x = (0.3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(dtype)
    return A, b, x


def synthetic(loss, N, d, dtype=np.float64, seed=1):
    """Random problem of the BASELINE shape: A ~ N(0,1)/sqrt(d); LS targets from a sparse x_true, or +-1 labels."""
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((N, d)) / np.sqrt(d)).astype(dtype)
    x_true = rng.standard_normal(d) * (rng.random(d) < 0.2)
    t = A.astype(np.float64) @ x_true + 0.01 * rng.standard_normal(N)
    if loss == "ls":
        b = t.astype(dtype)
    else:
        b = np.where(t >= 0, 1.0, -1.0).astype(dtype)
    x = (0.3 * rng.standard_normal(d)).astype(dtype)
    return A, b, x


# ---- test/test_sharing.jl:11-31 (literal) -------------------------------------------------------------------------------
def sharing_fixture(dtype=np.float64):
    """(Q, q, eta, lo, hi, L, g_hi, x0, sum_star): N = 3 agents, n = 2;  f_i = Sum(Quadratic(diagm(d_i), ones), SqrDistL2(IndBox(-2,2), eta)),
    eta = N*10, g = IndBox(-Inf, ones): sum_i x_i <= 1.  L_i = opnorm(Q[i]) + eta where Q[i] is the i-th ENTRY (linear index)
    of the 2 x 2 matrix diagm(d_i) -- a quirk of the reference test (:23): entries 1, 0, 0  ->  L = [31, 30, 30]."""
    dd = np.array([[1.0, 2.0], [-1.0, 3.0], [0.0, 10.0]])
    N, n = dd.shape
    eta = N * 10.0
    L = np.array([abs(dd[0, 0]) + eta, 0.0 + eta, 0.0 + eta])
    return (dd.astype(dtype), np.ones((N, n), dtype), eta, -2.0, 2.0, L.astype(dtype), np.ones(n, dtype), np.zeros(n, dtype),
            np.array([-5.136781609195401, -0.9333333333333327]))


# every GPU-vs-oracle comparison of tests/test_gpu_parity.py appends {test, line, what, dtype, ratio, scale} here
PARITY_LOG = []
