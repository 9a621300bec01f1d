"""Size-independent properties at sizes the CPU oracle cannot check in seconds (the BASELINE shapes, d = 1024):
affinity of the least-squares gradient, shard additivity, run-to-run determinism, objective/gradient consistency,
and the aggregate/table invariants after 10^5 dependent chain steps.  Data is generated on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D = 1_000_000, 1024   # 8.2 GB fp64: BASELINE config #2


@pytest.fixture(scope="module")
def big(ctx):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF
    dev = torch.device("cuda", 0)
    A = torch.empty((N, D), dtype=torch.float64, device=dev)
    b = torch.empty((N,), dtype=torch.float64, device=dev)
    ctx.synth_normal(A, 0, seed=11, scale=1.0 / np.sqrt(D))
    rng = np.random.default_rng(11)
    xt = torch.from_numpy(rng.standard_normal(D) * (rng.random(D) < 0.05)).to(dev)
    F = PackedF(L.LOSS_LS, A, b, float(N))
    ctx.synth_targets(F, xt, 0.01, False, 11, b)
    ctx.synchronize()
    yield F
    del A, b
    torch.cuda.empty_cache()


def _rand(seed, scale=1.0):
    import torch
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(D) * scale).cuda()


def test_synthetic_data_is_what_it_says(ctx, big):
    """Counter-based generator: unit-variance/sqrt(d) entries, reproducible per (seed,row,col) whatever the row offset."""
    import torch
    A = big.A
    sub = A[:4096]
    assert abs(float(sub.mean())) < 2e-3 and abs(float(sub.var()) * D - 1.0) < 2e-2
    B = torch.empty((16, D), dtype=torch.float64, device="cuda")
    ctx.synth_normal(B, 1000, seed=11, scale=1.0 / np.sqrt(D))   # rows 1000..1015 generated as a different "shard"
    assert torch.equal(B, A[1000:1016])


def test_ls_gradient_is_affine_in_x(ctx, big):
    """grad(x) = (1/N) sum N a_i (a_i'x - b_i) is affine:  G(x1 + x2) = G(x1) + G(x2) - G(0)."""
    import torch
    x1, x2 = _rand(1), _rand(2)
    g = [torch.empty(D, dtype=torch.float64, device="cuda") for _ in range(4)]
    ctx.full_gradient(big, x1, g[0])
    ctx.full_gradient(big, x2, g[1])
    ctx.full_gradient(big, x1 + x2, g[2])
    ctx.full_gradient(big, torch.zeros_like(x1), g[3])
    lhs, rhs = g[2], g[0] + g[1] - g[3]
    scale = float(torch.max(torch.abs(g[0])) + torch.max(torch.abs(g[1])))
    assert float(torch.max(torch.abs(lhs - rhs))) <= 1e-11 * scale


def test_full_size_sweep_is_deterministic_and_shard_additive(ctx, big):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF
    x = _rand(3, 0.1)
    a1, a2 = (torch.empty(D, dtype=torch.float64, device="cuda") for _ in range(2))
    ctx.full_gradient(big, x, a1)
    ctx.full_gradient(big, x, a2)
    assert torch.equal(a1, a2), "fixed summation order: bitwise reproducible"
    # the same rows as 3 uneven shards with N_total = N: the shard sums add up to the whole
    cuts = [0, 333_333, 700_001, N]
    acc = torch.zeros(D, dtype=torch.float64, device="cuda")
    part = torch.empty_like(acc)
    for lo, hi in zip(cuts, cuts[1:]):
        Fs = PackedF(L.LOSS_LS, big.A[lo:hi], big.b[lo:hi], float(N), N_total=N, row0=lo)
        ctx.full_gradient(Fs, x, part)
        acc += part
    assert float(torch.max(torch.abs(acc - a1))) <= 1e-12 * float(torch.max(torch.abs(a1)))


def test_objective_decreases_along_proximal_gradient_steps(ctx, big):
    """F(x+) <= F(x) for x+ = prox_{γg}(x - γ grad f(x)) with γ <= 1/L  (L = λ_max(A'A) ~ (1 + sqrt(N/d))^2 d/N ... < 2 here)."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    g = ProxG(L.PROX_L1, lam=1e-3)
    gamma = 1.0 / (1.2 * N / D * (1 + np.sqrt(D / N)) ** 2)   # 1/L for (1/N)·N·A'A = A'A, ||A'A|| ~ (N/d)(1+sqrt(d/N))^2
    x = torch.zeros(D, dtype=torch.float64, device="cuda")
    av, y = torch.empty_like(x), torch.empty_like(x)
    vals = [ctx.objective(big, g, x)]
    for _ in range(5):
        ctx.proxgrad_step(big, g, gamma, x, av, y)
        x, y = y, x
        vals.append(ctx.objective(big, g, x))
    assert all(b <= a * (1 + 1e-12) for a, b in zip(vals, vals[1:])), vals
    assert vals[-1] < 0.9 * vals[0]


def test_objective_matches_gradient_by_finite_differences(ctx, big):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    g0 = ProxG(L.PROX_ZERO)
    x, dx = _rand(5, 0.1), _rand(6, 1.0)
    grad = torch.empty(D, dtype=torch.float64, device="cuda")
    ctx.full_gradient(big, x, grad)
    eps = 1e-4
    fd = (ctx.objective(big, g0, x + eps * dx) - ctx.objective(big, g0, x - eps * dx)) / (2 * eps)
    an = float(torch.dot(grad, dx))
    assert abs(fd - an) <= 1e-7 * max(abs(an), 1.0)


def test_saga_invariant_after_many_chain_steps(ctx, ciao):
    """After 10^5 dependent SAGA steps (wave-specialised chain: table rows prefetched a ring ahead by the issuer waves, hazard re-reads):
    av == mean of the table rows, and every touched row equals grad f_i at SOME earlier iterate => row_i is a multiple of a_i."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    n, d = 20_000, 1024
    dev = torch.device("cuda", 0)
    A = torch.empty((n, d), dtype=torch.float32, device=dev)
    y = torch.empty((n,), dtype=torch.float32, device=dev)
    ctx.synth_normal(A, 0, seed=5, scale=1.0 / np.sqrt(d))
    F = PackedF(L.LOSS_LOGISTIC, A, y, 1.0)
    ctx.synth_targets(F, torch.ones(d, dtype=torch.float32, device=dev), 0.1, True, 5, y)
    g = ProxG(L.PROX_L1, lam=1.0 / n)
    x0 = torch.ones(d, dtype=torch.float32, device=dev)
    table = torch.empty((n, d), dtype=torch.float32, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    gamma = 1.0 / (3 * 0.25 * 1.4)
    ctx.saga_init(F, g, gamma, x0, table, av, z)
    idx = ciao.IndexStream(9).rand_indices(n, 100_000)
    ctx.saga_steps(F, g, gamma, False, idx, table, av, z)
    ctx.synchronize()
    assert "chain_ws_kernel" in ctx.last_kernel()      # the wave-specialised chain (4 KiB rows, SAGA)
    mean = table.double().mean(dim=0)
    assert float(torch.max(torch.abs(av.double() - mean))) <= 2e-4 * float(torch.max(torch.abs(mean))) + 1e-7
    # rank-1 structure: table_i = c_i * a_i  =>  |<table_i, a_i>| == ||table_i|| * ||a_i||
    rows = torch.from_numpy(np.unique(idx[:2000])).to(dev)
    t, a = table[rows].double(), A[rows].double()
    cos = (t * a).sum(1).abs() / (t.norm(dim=1) * a.norm(dim=1))
    assert float(cos.min()) > 1 - 1e-6
    assert bool(torch.isfinite(z).all())


def test_finito_invariant_after_batches(ctx, ciao):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    n, d = 50_000, 1024
    dev = torch.device("cuda", 0)
    A = torch.empty((n, d), dtype=torch.float64, device=dev)
    b = torch.empty((n,), dtype=torch.float64, device=dev)
    ctx.synth_normal(A, 0, seed=6, scale=1.0 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(n))
    ctx.synth_targets(F, torch.ones(d, dtype=torch.float64, device=dev), 0.01, False, 6, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    gam = (0.999 * n / (float(n) * (A * A).sum(1))).contiguous()
    hg = ctx.hat_gamma(gam)
    assert abs(hg - 1.0 / float((1.0 / gam).sum())) <= 1e-12 * hg
    x0 = torch.zeros(d, dtype=torch.float64, device=dev)
    table = torch.empty((n, d), dtype=torch.float64, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    st = ciao.IndexStream(4)
    for r, nit in ((1, 5000), (32, 200), (4096, 6)):   # chain, chain, batch-parallel
        batches = [st.sample_without_replacement(n, r) for _ in range(nit)]
        bptr = np.arange(nit + 1, dtype=np.int64) * r
        ctx.finito_steps(F, g, gam, hg, bptr, np.concatenate(batches), table, av, z)
    ctx.synchronize()
    inv = hg * (table / gam[:, None]).sum(dim=0)
    assert float(torch.max(torch.abs(av - inv))) <= 1e-9 * float(torch.max(torch.abs(inv)))


@pytest.mark.parametrize("tdt,d", [("float64", 1000), ("float64", 3000), ("float32", 1500), ("float64", 4096)])
def test_odd_row_lengths_at_scale(ctx, ciao, tdt, d):
    """Rows that are not 64*VEC*{1,2,4,8,16} elements run the masked workgroup-per-row kernel and the masked LDS-DMA chain:
    the sweep against a dense fp64 matrix product, then the SAGA and Finito aggregate invariants after chains and batches."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    dt = getattr(torch, tdt)
    n = 200_000
    dev = torch.device("cuda", 0)
    A = torch.empty((n, d), dtype=dt, device=dev)
    b = torch.empty((n,), dtype=dt, device=dev)
    ctx.synth_normal(A, 0, seed=9, scale=1.0 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(n))
    xt = torch.from_numpy(np.random.default_rng(9).standard_normal(d) * 0.1).to(dev, dt)
    ctx.synth_targets(F, xt, 0.01, False, 9, b)
    x = torch.from_numpy(np.random.default_rng(10).standard_normal(d) * 0.05).to(dev, dt)
    av = torch.empty(d, dtype=dt, device=dev)
    ctx.full_gradient(F, x, av)
    assert "rows_split_kernel" in ctx.last_kernel() and "masked" in ctx.last_kernel(), ctx.last_kernel()
    ref = torch.zeros(d, dtype=torch.float64, device=dev)
    for lo in range(0, n, 50_000):   # dense reference in fp64, in slabs
        Ad = A[lo:lo + 50_000].double()
        ref += Ad.T @ (Ad @ x.double() - b[lo:lo + 50_000].double())
    tol = 1e-11 if dt == torch.float64 else 2e-4
    assert float(torch.max(torch.abs(av.double() - ref))) <= tol * float(torch.max(torch.abs(ref)))

    g = ProxG(L.PROX_L1, lam=1e-3)
    table = torch.empty((n, d), dtype=dt, device=dev)
    z = torch.empty(d, dtype=dt, device=dev)
    st = ciao.IndexStream(12)
    # SAGA: av == mean of the table after 20 000 dependent steps
    gamma = 1.0 / (3.0 * 1.5 * n)
    ctx.saga_init(F, g, gamma, x, table, av, z)
    ctx.saga_steps(F, g, gamma, False, st.rand_indices(n, 20_000), table, av, z)
    assert ("chain_ws_kernel" in ctx.last_kernel() or "chain_dma_kernel" in ctx.last_kernel()) and \
        ("masked" in ctx.last_kernel()) == ((d * A.element_size()) % 4096 != 0), ctx.last_kernel()
    mean = table.double().mean(dim=0)
    rtol = 1e-9 if dt == torch.float64 else 5e-3
    assert float(torch.max(torch.abs(av.double() - mean))) <= rtol * float(torch.max(torch.abs(mean)))
    # Finito: av == hat_gamma * sum_i s_i / gamma_i after chains and batch-parallel steps
    gam = (0.999 * n / (float(n) * (A.double() ** 2).sum(1))).to(dt).contiguous()
    hg = ctx.hat_gamma(gam)
    ctx.finito_init(F, g, gam, hg, x, table, av, z)
    for r, nit in ((1, 3000), (300, 20), (5000, 4)):
        batches = [st.sample_without_replacement(n, r) for _ in range(nit)]
        ctx.finito_steps(F, g, gam, hg, np.arange(nit + 1, dtype=np.int64) * r, np.concatenate(batches), table, av, z)
    ctx.synchronize()
    inv = hg * (table.double() / gam.double()[:, None]).sum(dim=0)
    assert float(torch.max(torch.abs(av.double() - inv))) <= rtol * float(torch.max(torch.abs(inv)))
    assert bool(torch.isfinite(z).all())
    del A, table
    torch.cuda.empty_cache()
