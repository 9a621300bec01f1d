"""Every BASELINE.json config at its stated size, inside `pytest -m gpu`.

  C1  Lasso SVRG N=1000 d=50 fp64: the reference's generator (test/test_lasso.jl:15-47) at exactly that size, 30 SVRG
      epochs (29 000 dependent updates + 30 sweeps) through the solver functor, iterate by iterate against the oracle.
  C2  Lasso SVRG N=1M d=1024 fp64: 4096 chain updates at the real N against the ORACLE run on the rows they visit (both chain
      variants), then one real epoch (objective decreases, av == full gradient at z_full); tests/test_gpu_properties.py has
      the same shape.
  C3  l1-logistic SAGA N=10M d=1024 fp32, 41 GB gradient table in HBM: init, 4096 steps against the ORACLE run on the rows and
      table rows they visit (state and every visited table row), then 2*10^6 steps; the invariant
      av == mean(table) is checked on the touched rows (av_now - av_init == (1/N) sum_touched (s_i - s_i^init)) and every
      touched table row is a multiple of its data row (grad f_i = c_i a_i: rank-1 structure).
  C4  Lasso SVRG N=80M over 8 GPUs -> this rank's share is the bench shape, 10M x 1024 fp64 (rank 3's row0, the global 1/N):
      the shard's sweep against the ORACLE at the real size, slab by slab (500k rows to the host at a time; all-cores long-double
      sums of every slab chained, plus the reference's own sequential order on the first slab), then shard additivity of the
      sweep (two half shards, each scaled by the global N, add up to the whole) -- what the all-reduce relies on.
  C5  Finito N=10M d=4096 fp32 over 8 GPUs -> this rank's share 1.25M x 4096 fp32 (20.5 GB + 20.5 GB table, rank 3's row0, the
      global 1/N, per-sample gamma_i): three batches of 4096 against the ORACLE run on the 12 288 rows and table rows they visit
      (z, av, every visited table row), then 60 batches: av == hat_gamma * sum_i s_i / gamma_i, and the objective decreases.
The 8-GPU legs of C4 / C5 need an 8-GPU node (the driver's SCALE run); their single-rank work is what runs here.
"""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def synth(ctx, N, d, tdt, logistic, seed, N_total=None, row0=0):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF
    dev = torch.device("cuda", 0)
    A = torch.empty((N, d), dtype=tdt, device=dev)
    b = torch.empty((N,), dtype=tdt, device=dev)
    ctx.synth_normal(A, row0, seed=seed, scale=1.0 / np.sqrt(d))
    rng = np.random.default_rng(seed)
    xt = torch.from_numpy(rng.standard_normal(d) * (rng.random(d) < 0.05)).to(dev, tdt)
    Nt = N if N_total is None else N_total
    F = PackedF(L.LOSS_LOGISTIC if logistic else L.LOSS_LS, A, b, 1.0 if logistic else float(Nt), N_total=Nt, row0=row0)
    ctx.synth_targets(F, xt, 0.1 if logistic else 0.01, logistic, seed, b)
    ctx.synchronize()
    return F


@pytest.fixture(autouse=True)
def _free_hbm():
    yield
    import torch
    torch.cuda.empty_cache()


def gather_rows(F, idx):
    """The rows a chain visits, on the host: (touched rows sorted, A[touched], b[touched], idx remapped into them)."""
    import torch
    touched, remap = np.unique(idx, return_inverse=True)
    t = torch.from_numpy(touched).cuda()
    return touched, F.A[t].cpu().numpy(), F.b[t].cpu().numpy(), remap.astype(np.int64), t


def rel_eps(dev, ref, dtype):
    ref = np.asarray(ref)
    return float(np.abs(dev.cpu().numpy().astype(np.float64) - ref.astype(np.float64)).max() /
                 (np.finfo(dtype).eps * max(np.abs(ref).max(), 1e-300)))


def test_C1_lasso_svrg_N1000_d50_fp64_against_the_oracle(ciao, ctx):
    """30 epochs x 1000 updates: the device iterate against the oracle's at every yielded state, 1e-11 relative (observed
    7e-14 after all 29 000 dependent updates), and the known answer of the generator is approached."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    from ciaoalgorithms_jl_amd import solvers as S
    from oracle import oracle as O
    from oracle import ref_solvers as RS
    A, b, Lc, lam, x0, x_star, f_star = P.lasso_known_answer(N=1000, n=50, p=5, seed=0)
    N = A.shape[0]
    assert A.shape == (1000, 50) and A.dtype == np.float64
    gamma = 1.0 / (7 * Lc.max())                                            # test_lasso.jl:164
    F = PackedF.least_squares(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), float(N))
    g = ProxG(L.PROX_L1, lam=lam)
    it_dev = iter(S.iterator(S.SVRG(np.float64, γ=gamma), x0, F=F, g=g, N=N, ctx=ctx, stream=ciao.IndexStream(0)))
    it_ref = iter(RS.SVRGIterable(O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=lam), x0, gamma=gamma, stream=ciao.IndexStream(0)))
    worst = 0.0
    for k in range(30):
        sd, sr = next(it_dev), next(it_ref)
        err = np.abs(sd.z_full.cpu().numpy() - sr.z_full).max() / max(np.abs(sr.z_full).max(), 1e-300)
        worst = max(worst, err)
        assert err <= 1e-11, f"state {k + 1}: relative error {err:.2e}"
    x, it = S.SVRG(np.float64, γ=gamma, maxit=30)(x0, F=F, g=g, N=N, ctx=ctx, stream=ciao.IndexStream(0))
    assert it == 30 and np.array_equal(x, sd.z_full.cpu().numpy()), "the functor is the iterator driven 30 times"
    gap0, gap = P.lasso_cost(A, b, lam, x0) - f_star, P.lasso_cost(A, b, lam, x) - f_star
    assert 0 <= gap < 0.05 * gap0
    P.PARITY_LOG.append({"test": "test_C1", "line": 0, "what": "C1 worst relative iterate error over 30 states", "dtype": "float64",
                         "ratio": worst / np.finfo(np.float64).eps, "scale": 1e-11 / np.finfo(np.float64).eps})
    # the two paths' wall time for the same 29 epochs (recorded, not asserted: a 400 KB problem is one workgroup's worth of work;
    # VERDICT r2 "weak" 3 -- the device needed 11.9 ms against the oracle's 8.4 in round 2)
    import time
    def timed(make, n=30):
        it = iter(make())
        next(it)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n - 1):
            next(it)
        ctx.synchronize()
        return time.perf_counter() - t0
    t_dev = min(timed(lambda: S.iterator(S.SVRG(np.float64, γ=gamma), x0, F=F, g=g, N=N, ctx=ctx, stream=ciao.IndexStream(0))) for _ in range(3))
    t_ref = min(timed(lambda: RS.SVRGIterable(O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=lam), x0, gamma=gamma,
                                              stream=ciao.IndexStream(0))) for _ in range(3))
    P.PARITY_LOG.append({"test": "test_C1", "line": 1, "what": f"C1 wall time of 29 epochs: device {t_dev * 1e3:.2f} ms, oracle on one host core "
                                                              f"{t_ref * 1e3:.2f} ms", "dtype": "float64", "ratio": t_dev / t_ref, "scale": 1e9})


def test_C2_lasso_svrg_N1M_d1024_fp64_one_epoch(ciao, ctx):
    """One real epoch (m = N = 10^6 dependent updates + tail + sweep): av is the full gradient at the new z_full (checked
    with an independent sweep), w == z_full, z == 0, the objective went down, and the monitored objective of the tail pass
    is the objective at z_full."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    N, d = 1_000_000, 1024
    F = synth(ctx, N, d, torch.float64, False, 2)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=torch.float64, device="cuda")
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    obj = torch.zeros(3, dtype=torch.float64, device="cuda")
    f0 = ctx.objective(F, g, x0)
    gamma = 1.0 / (7 * 1.3 * N)
    # ---- the chain at the config's real N, d and row addresses against the ORACLE: 4096 updates from the
    # init state on the device; the <= 4096 rows they visit are gathered to the host and the oracle runs the same updates on
    # that submatrix from the same (av, z, z_full, w); tolerance 2000 eps (observed 194: the linear growth law of DESIGN section 5).  Both chain variants: two dot products per update (ciao_svrg_inner) and
    # the row dots of the full pass reused (ciao_svrg_iterate with reuse_rowdots, whose tail gives z_full = z / m).
    from oracle import oracle as O
    x1 = torch.from_numpy(np.random.default_rng(8).standard_normal(d) * 0.01).cuda()    # not the all-zero start: every term is live
    ctx.svrg_init(F, x1, av, z, zf, w)
    idx4 = ciao.IndexStream(5).rand_indices(N, 4096)
    _, A_t, b_t, remap, _ = gather_rows(F, idx4)
    op = O.Problem("ls", A_t, b_t, float(N), N_total=N)
    og = O.Prox("l1", lam=1e-3)
    h = [t.cpu().numpy().copy() for t in (av, z, zf, w)]
    O.svrg_inner(op, og, gamma, remap, *h)
    d2 = [t.clone() for t in (av, z, zf, w)]
    ctx.svrg_inner(F, g, gamma, idx4, *d2)
    assert "chain_dma_kernel<f64,J2,alg0>" in ctx.last_kernel(), ctx.last_kernel()
    e_w, e_z = rel_eps(d2[3], h[3], np.float64), rel_eps(d2[1], h[1], np.float64)
    assert e_w <= 2000 and e_z <= 2000, f"C2 chain (two dots) vs oracle after 4096 updates: w {e_w:.0f} eps, z {e_z:.0f} eps"
    d1 = [t.clone() for t in (av, z, zf, w)]
    ctx.svrg_iterate(F, g, gamma, idx4, False, *d1, reuse_rowdots=True)
    e_zf = rel_eps(d1[2], h[1] / 4096.0, np.float64)
    assert e_zf <= 2000, f"C2 chain (cached row dots) vs oracle after 4096 updates: z_full {e_zf:.0f} eps"
    P.PARITY_LOG.append({"test": "test_C2", "line": 0, "what": "C2 4096 chain updates at N=1M vs oracle (w, z, z_full; eps)", "dtype": "float64",
                         "ratio": max(e_w, e_z, e_zf), "scale": 2000.0})
    ctx.svrg_init(F, x0, av, z, zf, w)
    ctx.set_monitor(g, obj)
    ctx.svrg_iterate(F, g, gamma, ciao.IndexStream(0).rand_indices(N, N), False, av, z, zf, w, reuse_rowdots=True)
    ctx.set_monitor(None, None)
    ctx.synchronize()
    f1 = ctx.objective(F, g, zf)
    assert f1 < f0 and abs(obj[0].item() - f1) <= 1e-12 * abs(f1)
    chk = torch.empty_like(x0)
    ctx.full_gradient(F, zf, chk)
    assert torch.equal(chk, av), "the epoch's tail pass and a separate sweep at z_full are the same deterministic computation"
    assert torch.equal(w, zf) and torch.count_nonzero(z).item() == 0


def test_C3_l1logistic_saga_N10M_d1024_fp32_table_in_hbm(ciao, ctx):
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    N, d = 10_000_000, 1024
    F = synth(ctx, N, d, torch.float32, True, 3)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    gamma = 1.0 / (3 * 0.25 * 1.3)                                        # 1/(3 max L), test_logistic_l1.jl:39
    x0 = torch.ones(d, dtype=torch.float32, device="cuda")
    table = torch.empty((N, d), dtype=torch.float32, device="cuda")        # 40.96 GB
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, gamma, x0, table, av, z)
    av0 = av.double().clone()
    # ---- the chain at the config's real N, table size and addresses against the ORACLE: 4096 steps from the
    # init state (with repeats inside the prefetch window); the rows and table rows they visit are gathered to the host and the
    # oracle runs the same steps on them with 1/N of the whole problem; state and every visited table row are compared to
    # 50 eps (observed 3.6).
    from oracle import oracle as O
    idx4 = ciao.IndexStream(7).rand_indices(N, 4096)
    idx4[100:103] = idx4[100]
    idx4[2000] = idx4[1990]
    touched4, A_t, b_t, remap, t4 = gather_rows(F, idx4)
    op = O.Problem("logistic", A_t, b_t, 1.0, N_total=N)
    og = O.Prox("l1", lam=1.0 / N)
    h_tab, h_av, h_z = table[t4].cpu().numpy(), av.cpu().numpy().copy(), z.cpu().numpy().copy()
    O.saga_steps(op, og, np.float32(gamma), False, remap, h_tab, h_av, h_z)
    ctx.saga_steps(F, g, gamma, False, idx4, table, av, z)
    assert "chain_ws_kernel<f32,J1,alg1" in ctx.last_kernel() or "chain_dma_kernel<f32,J1,alg1>" in ctx.last_kernel(), ctx.last_kernel()
    e_z, e_av, e_t = rel_eps(z, h_z, np.float32), rel_eps(av, h_av, np.float32), rel_eps(table[t4], h_tab, np.float32)
    assert e_z <= 50 and e_av <= 50 and e_t <= 50, f"C3 chain vs oracle after 4096 steps: z {e_z:.0f}, av {e_av:.0f}, table {e_t:.0f} eps"
    P.PARITY_LOG.append({"test": "test_C3", "line": 0, "what": "C3 4096 SAGA steps at N=10M (41 GB table) vs oracle (z, av, table; eps)",
                         "dtype": "float32", "ratio": max(e_z, e_av, e_t), "scale": 50.0})
    ctx.saga_init(F, g, gamma, x0, table, av, z)        # back to the init state for the long run below
    steps = 2_000_000
    idx = ciao.IndexStream(0).rand_indices(N, steps)
    touched = np.unique(idx)
    tdev = torch.from_numpy(touched).cuda()
    # the init rows of the touched samples: grad f_i(x0) = c_i a_i, recomputed from the data in float64
    a_t = F.A[tdev].double()
    c0 = -F.b[tdev].double() / (1.0 + torch.exp(F.b[tdev].double() * (a_t @ x0.double())))
    for k in range(0, steps, 500_000):
        ctx.saga_steps(F, g, gamma, False, idx[k:k + 500_000], table, av, z)
    kern = ctx.last_kernel()
    ctx.synchronize()
    rows = table[tdev].double()
    # (1) rank-1 structure: every touched row is c_i * a_i for one scalar c_i
    c = (rows * a_t).sum(dim=1) / (a_t * a_t).sum(dim=1)
    resid = (rows - c[:, None] * a_t).abs().max().item()
    assert resid <= 5e-6 * rows.abs().max().item(), resid
    # (2) av == mean(table): only touched rows changed, so av - av_init == (1/N) sum_touched (s_i - s_i_init)
    delta = ((c - c0)[:, None] * a_t).sum(dim=0) / N
    err = (av.double() - av0 - delta).abs().max().item()
    assert err <= 2e-4 * max(av.double().abs().max().item(), av0.abs().max().item()), err
    # (3) untouched rows are still the init gradients (spot check) and the objective went down
    untouched = np.setdiff1d(np.arange(0, N, 997_001), touched)[:4]
    for i in untouched:
        a_i = F.A[int(i)].double()
        ci = -F.b[int(i)].double() / (1.0 + torch.exp(F.b[int(i)].double() * (a_i @ x0.double())))
        assert (table[int(i)].double() - ci * a_i).abs().max().item() <= 1e-6
    assert ctx.objective(F, g, z) < ctx.objective(F, g, x0)
    assert "chain_ws_kernel<f32,J1,alg1" in kern, kern


def test_C4_share_sweep_against_the_oracle_and_shard_additivity_at_10M_rows(ctx):
    """The per-GPU share of N = 80M (10M x 1024 fp64, 82 GB; rank 3's rows, lambda_f = N = 80M as in test_lasso.jl:54).
    (1) The shard's sweep against the oracle at the REAL size (SVRG_basic.jl:87-92): the rows travel to the
        host 500k at a time; `orc_shard_sums_omp` adds every slab's sum_i c_i a_i in long double (all cores) -- the value of the
        shard's sum in exact arithmetic to << 1 eps -- and the device's av = sum / N_total must agree to 16 eps of |av|_inf
        (observed 1.1); on the first slab alone the device (that slab as a shard of the same problem) is also held against
        `orc_shard_pass`, the reference's own sequential `av += grad/N` order, to 2000 eps (observed 112: a left fold over
        5*10^5 cancelling terms is itself a hundred eps from the exact sum).
    (2) The sweep over the whole shard equals the sum of the sweeps over its two halves, each computed as a shard of the same
        global problem (N_total kept) -- the identity the RCCL all-reduce of the d-vector relies on; the same result twice."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import PackedF
    n, d, N_total = 10_000_000, 1024, 80_000_000
    F = synth(ctx, n, d, torch.float64, False, 0, N_total=N_total, row0=3 * n)     # rank 3's shard of the 80M-row problem
    x = torch.from_numpy(np.random.default_rng(1).standard_normal(d) * 0.1).cuda()
    whole, again, lo, hi = (torch.empty(d, dtype=torch.float64, device="cuda") for _ in range(4))
    ctx.full_gradient(F, x, whole)
    ctx.full_gradient(F, x, again)
    assert torch.equal(whole, again)
    assert "rows_fast_kernel<f64,K8" in ctx.last_kernel(), ctx.last_kernel()
    from oracle import oracle as O
    SL = 500_000
    xh = x.cpu().numpy()
    acc = np.zeros(d, np.longdouble)
    pinA = torch.empty((SL, d), dtype=torch.float64).pin_memory()
    pinb = torch.empty((SL,), dtype=torch.float64).pin_memory()
    for k in range(0, n, SL):
        pinA.copy_(F.A[k:k + SL])
        pinb.copy_(F.b[k:k + SL])
        slab = O.Problem("ls", pinA.numpy(), pinb.numpy(), float(N_total), N_total=N_total)
        O.shard_sums_omp(slab, xh, acc)
        if k == 0:      # the reference's sequential order on the first slab, against the device's sweep over exactly those rows
            seq = O.shard_pass(slab, xh, np.zeros(d))
            F0 = PackedF(L.LOSS_LS, F.A[:SL], F.b[:SL], float(N_total), N_total=N_total, row0=3 * n)
            ctx.full_gradient(F0, x, again)
            e_seq = rel_eps(again, seq, np.float64)
            assert e_seq <= 2000, f"C4 first slab vs the sequential oracle: {e_seq:.0f} eps"
    ref = np.asarray(acc / N_total, dtype=np.float64)
    e_av = rel_eps(whole, ref, np.float64)
    assert e_av <= 16, f"C4 shard sweep (10M x 1024 fp64, row0 = 30M, 1/N of 80M) vs the oracle's long-double sums: {e_av:.1f} eps"
    P.PARITY_LOG.append({"test": "test_C4", "line": 0, "what": "C4 shard sweep at 10M x 1024 fp64 vs oracle long-double slab sums (eps of |av|inf)",
                         "dtype": "float64", "ratio": e_av, "scale": 16.0})
    P.PARITY_LOG.append({"test": "test_C4", "line": 1, "what": "C4 first 500k-row slab vs the oracle's sequential order (eps)",
                         "dtype": "float64", "ratio": e_seq, "scale": 2000.0})
    del pinA, pinb
    h = n // 2
    Flo = PackedF(L.LOSS_LS, F.A[:h], F.b[:h], float(N_total), N_total=N_total, row0=3 * n)
    Fhi = PackedF(L.LOSS_LS, F.A[h:], F.b[h:], float(N_total), N_total=N_total, row0=3 * n + h)
    ctx.full_gradient(Flo, x, lo)
    ctx.full_gradient(Fhi, x, hi)
    err = (whole - (lo + hi)).abs().max().item()
    assert err <= 1e-12 * whole.abs().max().item(), err
    # the shard's contribution carries the GLOBAL 1/N: a shard of an 80M-row problem, not a 10M-row problem of its own
    Fown = PackedF(L.LOSS_LS, F.A, F.b, float(N_total), N_total=n)
    ctx.full_gradient(Fown, x, lo)
    assert (lo / 8 - whole).abs().max().item() <= 1e-12 * whole.abs().max().item()


def test_C5_share_finito_N1p25M_d4096_fp32_batches_of_4096(ciao, ctx):
    """The per-GPU share of config #5 (Finito N = 10M, d = 4096 fp32 over 8 GPUs): rank 3's 1.25M rows + its 20.5 GB table
    shard, 1/N of the WHOLE problem, per-sample stepsizes.
    (1) VERDICT r3 item 1b (Finito_basic.jl:109-118): from the init state three static batches of 4096 (`ciao_finito_steps_blocks`);
        the 12 288 rows and table rows they visit are gathered to the host and `orc_finito_steps` runs the same three iterations on
        that submatrix with N_total; z, av and EVERY visited table row are compared -- with the oracle run in Float64 on the same
        Float32 data (16 eps32 on z / av, 8 on the table rows) AND in Float32 (table rows 20 eps; z / av within the bound of the
        reference's own left fold, which at 1/N = 1e-7 loses up to half an eps of av per sample: see the comment in the body).
    (2) the same rows as a problem of their own: 60 batches, av == hat_gamma * sum_i s_i / gamma_i over the whole table, the
        objective decreases."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    from oracle import oracle as O
    N, d, r, N_total = 1_250_000, 4096, 4096, 10_000_000
    row0 = 3 * N
    F = synth(ctx, N, d, torch.float32, False, 5, N_total=N_total, row0=row0)
    g = ProxG(L.PROX_L1, lam=1e-3)
    # gamma_i = alpha N / L_i (Finito_basic.jl:69) with L_i = lambda_f |a_i|^2 ~ N (1 +- 10 %): per-sample, not uniform
    wob = 1.0 + 0.1 * torch.frac(torch.arange(N, device="cuda", dtype=torch.float64) * 0.6180339887498949)
    gam = (0.999 * N_total / (1.3 * N_total) * wob).float()
    hg = ctx.hat_gamma(gam) / 8.0                                          # 1 / sum over ALL ranks of 1/gamma_i (8 shards alike)
    x0 = torch.zeros(d, dtype=torch.float32, device="cuda")
    table = torch.empty((N, d), dtype=torch.float32, device="cuda")        # 20.48 GB
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    # ---- (1) three batches at the real size against the oracle on the rows they visit
    nb0 = 3
    first0 = np.array([1, 2, 150], dtype=np.int64) * r                     # cyclic order starts at batch 2 (Finito_basic.jl:99); one far block
    vis = np.concatenate([np.arange(f, f + r, dtype=np.int64) for f in first0])
    tv = torch.from_numpy(vis).cuda()
    A_t, b_t, gam_t = F.A[tv].cpu().numpy(), F.b[tv].cpu().numpy(), gam[tv].cpu().numpy()
    h_tab, h_av, h_z = table[tv].cpu().numpy(), av.cpu().numpy().copy(), z.cpu().numpy().copy()
    op = O.Problem("ls", A_t, b_t, float(N_total), N_total=N_total)
    og = O.Prox("l1", lam=1e-3)
    batches0 = [np.arange(k * r, (k + 1) * r) for k in range(nb0)]
    # (a) the oracle in the problem's own precision: the reference's Float32 left fold `av .+= (t - s_i) * (hat_gamma / gamma_i)`
    O.finito_steps(op, og, gam_t, np.float32(hg), batches0, h_tab, h_av, h_z)
    # (b) the same iterations by the oracle in Float64 on the SAME Float32 data and start state: what the three batches compute,
    #     free of the fold's own rounding
    op64 = O.Problem("ls", A_t.astype(np.float64), b_t.astype(np.float64), float(N_total), N_total=N_total)
    w_tab, w_av, w_z = (table[tv].cpu().numpy().astype(np.float64), av.cpu().numpy().astype(np.float64), z.cpu().numpy().astype(np.float64))
    O.finito_steps(op64, og, gam_t.astype(np.float64), float(np.float32(hg)), batches0, w_tab, w_av, w_z)
    ctx.finito_steps_blocks(F, g, gam, hg, first0, np.full(nb0, r, np.int64), table, av, z)
    ctx.synchronize()
    assert "rows_split_kernel<f32,J4,mode4>" in ctx.last_kernel(), ctx.last_kernel()
    e_z64, e_av64, e_t64 = rel_eps(z, w_z, np.float32), rel_eps(av, w_av, np.float32), rel_eps(table[tv], w_tab, np.float32)
    assert e_z64 <= 16 and e_av64 <= 16 and e_t64 <= 8, \
        f"C5 share, 3 batches of 4096 vs the Float64 oracle on the same data: z {e_z64:.2f}, av {e_av64:.2f}, table rows {e_t64:.2f} eps(fp32)"
    # Against the Float32 oracle the table rows (elementwise) agree to a few eps, and z / av differ by what the reference's fold loses:
    # one increment is (t - s_i) hat_gamma / gamma_i ~ |z| / N_total ~ 1e-7 |av|, the size of HALF AN ULP of av in Float32, so the fold
    # `av += increment` rounds most of every increment away (up to 0.5 eps per sample, one-sided); the device adds the 4096
    # increments of a batch in a tree first and then once to av.  Bound stated: 0.5 eps per folded sample = 6144 eps for 12 288
    # (observed ~2900); the Float64 comparison above is the accuracy statement.
    e_z, e_av, e_t = rel_eps(z, h_z, np.float32), rel_eps(av, h_av, np.float32), rel_eps(table[tv], h_tab, np.float32)
    assert e_t <= 20 and e_z <= 0.5 * nb0 * r and e_av <= 0.5 * nb0 * r, \
        f"C5 share vs the Float32 oracle: z {e_z:.1f}, av {e_av:.1f}, table rows {e_t:.1f} eps"
    fold32 = rel_eps(torch.from_numpy(h_av), w_av, np.float32)            # the Float32 fold against the Float64 one: the reference's own loss
    assert fold32 > 4 * e_av64, "the Float32 left fold is expected to be the less accurate of the two here"
    P.PARITY_LOG.append({"test": "test_C5", "line": 0, "what": "C5 share 3 x 4096-row Finito batches at 1.25M x 4096 fp32 vs the oracle in Float64 on the same data (z, av; eps32)",
                         "dtype": "float32", "ratio": max(e_z64, e_av64), "scale": 16.0})
    P.PARITY_LOG.append({"test": "test_C5", "line": 1, "what": "C5 share visited table rows vs oracle (Float32 and Float64; eps32)", "dtype": "float32",
                         "ratio": max(e_t, e_t64), "scale": 8.0})
    P.PARITY_LOG.append({"test": "test_C5", "line": 2, "what": "C5 share z/av vs the Float32 oracle's left fold (eps32; bound 0.5 per folded sample; "
                                                              f"the fold itself is {fold32:.0f} eps from the Float64 value)",
                         "dtype": "float32", "ratio": max(e_z, e_av), "scale": 0.5 * nb0 * r})
    # ---- (2) the same rows as a problem of their own (N = 1.25M: what one rank's kernels see, with a 1/N under which 60 batches
    # move z visibly): 60 batches, the invariant over the whole table, and descent
    from ciaoalgorithms_jl_amd.device import PackedF
    Fown = PackedF(L.LOSS_LS, F.A, F.b, float(N))
    hg = ctx.hat_gamma(gam)
    ctx.finito_init(Fown, g, gam, hg, x0, table, av, z)
    f0 = ctx.objective(Fown, g, z)
    nb = 60
    first = (np.arange(1, nb + 1, dtype=np.int64) % (N // r)) * r          # cyclic: the first step uses batch 2 (Finito_basic.jl:99)
    ctx.finito_steps_blocks(Fown, g, gam, hg, first, np.full(nb, r, np.int64), table, av, z)
    ctx.synchronize()
    assert "rows_split_kernel<f32,J4,mode4>" in ctx.last_kernel()
    # invariant av == hat_gamma * sum_i s_i / gamma_i over the WHOLE 1.25M-row table (float64 column sums in slabs)
    acc = torch.zeros(d, dtype=torch.float64, device="cuda")
    for k in range(0, N, 125_000):
        acc += (table[k:k + 125_000].double() / gam[k:k + 125_000].double()[:, None]).sum(dim=0)
    inv = hg * acc
    err = (av.double() - inv).abs().max().item()
    assert err <= 2e-4 * inv.abs().max().item(), err
    assert ctx.objective(Fown, g, z) < f0
