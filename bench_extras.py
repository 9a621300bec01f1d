"""Secondary measurements reported under "extra" by `bench.py --extras` (and by tools/): the sequential chains
(SURVEY.md section 8a rows S3, G3, F3 with r = 1) and the batch-parallel table kernels (G2, F2, F3).  These are reported
honestly against the HBM bound although the chains are latency-bound by construction (one dependent step at a time).
"""
import time

import numpy as np
import torch


def _problem(ctx, dev, N, d, tdt, logistic, seed=1):
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import PackedF
    A = torch.empty((N, d), dtype=tdt, device=dev)
    b = torch.empty((N,), dtype=tdt, device=dev)
    ctx.synth_normal(A, 0, seed=seed, scale=1.0 / np.sqrt(d))
    rng = np.random.default_rng(seed)
    x_true = torch.from_numpy(rng.standard_normal(d) * (rng.random(d) < 0.05)).to(dev, tdt)
    F = PackedF(L.LOSS_LOGISTIC if logistic else L.LOSS_LS, A, b, 1.0 if logistic else float(N))
    ctx.synth_targets(F, x_true, noise=0.1, labels=logistic, seed=seed, b_out=b)
    return F


def _timed(ctx, fn, reps=1):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


def lambda_path(dev, Ks=(1, 2, 4, 8, 16, 32, 64, 128, 256), N=1_000_000, d=1024, m=40_000, alg="svrg", mode="streams"):
    """K independent chains over the SAME data matrix: a regularisation path (K values of lambda, an index stream each) of SVRG
    inner cycles, or of SAGA solves with a table each.  A chain is one workgroup on one CU and is latency-bound (25-30 GB/s).
    mode "streams": one context (= one HIP stream) per chain -- they run side by side as far as the HIP runtime gives the streams
    hardware queues of their own (GPU_MAX_HW_QUEUES, read when the runtime starts).  mode "batch": one context, the K calls
    recorded and launched as ONE grid of K workgroups (Context.chain_batch; include/ciao_hip.h ciao_ctx_chain_batch_begin).
    Returns the aggregate updates/s per K with its fraction of the HBM bound (d*s+8 bytes per SVRG update,
    3*d*s+8 per SAGA update)."""
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import Context, ProxG
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    tdt = torch.float64 if alg == "svrg" else torch.float32
    es = 8 if alg == "svrg" else 4
    boot = Context(dev.index)
    F = _problem(boot, dev, N, d, tdt, alg != "svrg")
    boot.synchronize()
    x0 = torch.zeros(d, dtype=tdt, device=dev) if alg == "svrg" else torch.ones(d, dtype=tdt, device=dev)
    Kmax = max(Ks)
    idxs = [boot._idx(IndexStream(k).rand_indices(N, m)) for k in range(Kmax)]
    gamma = 1.0 / (7 * 1.3 * N) if alg == "svrg" else 1.0 / (3 * 0.25 * 1.3)
    chains = []
    for k in range(Kmax):
        c = Context(dev.index, stream=torch.cuda.Stream(device=dev)) if mode == "streams" else boot
        g = ProxG(L.PROX_L1, lam=(1e-3 if alg == "svrg" else 1.0 / N) * (1.0 + k / Kmax))
        if alg == "svrg":
            st = tuple(torch.empty_like(x0) for _ in range(4))
            boot.svrg_init(F, x0, *st)
            tab = None
        else:
            st = (torch.empty_like(x0), torch.empty_like(x0))
            tab = torch.empty((N, d), dtype=tdt, device=dev)
            boot.saga_init(F, g, gamma, x0, tab, *st)
        chains.append((c, g, st, tab))
    boot.synchronize()

    def launch(c, g, st, tab, ix):
        if alg == "svrg":
            c.svrg_inner(F, g, gamma, ix, *st)
        else:
            c.saga_steps(F, g, gamma, False, ix, tab, *st)

    for k, (c, g, st, tab) in enumerate(chains):
        launch(c, g, st, tab, idxs[k][:512])
    torch.cuda.synchronize(dev)
    bytes_per = (d * es + 8) if alg == "svrg" else (3 * d * es + 8)
    curve = []
    for K in Ks:
        t0 = time.perf_counter()
        if mode == "batch":
            with boot.chain_batch():
                for k, (c, g, st, tab) in enumerate(chains[:K]):
                    launch(c, g, st, tab, idxs[k])
        else:
            for k, (c, g, st, tab) in enumerate(chains[:K]):
                launch(c, g, st, tab, idxs[k])
        torch.cuda.synchronize(dev)
        t = time.perf_counter() - t0
        agg = K * m / t
        curve.append({"K": K, "updates_per_s": agg, "us_per_update_per_chain": t / m * 1e6, "alg_GBps": agg * bytes_per / 1e9,
                      "frac_of_hbm_bound": agg * bytes_per / 8e12})
    kern = chains[0][0].last_kernel()
    if mode == "streams":
        for c, _, _, _ in chains:
            c.close()
    boot.close()
    import os
    return {"alg": alg, "mode": mode, "N": N, "d": d, "dtype": "f64" if es == 8 else "f32", "m_per_chain": m, "kernel": kern,
            "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "curve": curve}


def run(ctx, dev, quick=False):
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import ProxG
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    out = {}
    st = IndexStream(0)
    scale = 4 if quick else 1

    # ---- S3: SVRG inner cycle (sequential chain), Lasso N=1M d=1024 fp64 ------------------------------------------
    N, d = 1_000_000 // scale, 1024
    F = _problem(ctx, dev, N, d, torch.float64, False)
    g = ProxG(L.PROX_L1, lam=1e-3)
    gamma = 1.0 / (7 * 1.3 * N)
    x0 = torch.zeros(d, dtype=torch.float64, device=dev)
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    ctx.svrg_init(F, x0, av, z, zf, w)
    m = 200_000 // scale
    idx = ctx._idx(st.rand_indices(N, m))
    ctx.svrg_inner(F, g, gamma, idx[:1000], av, z, zf, w)   # warm
    t = _timed(ctx, lambda: ctx.svrg_inner(F, g, gamma, idx, av, z, zf, w))
    out["svrg_inner_f64_d1024"] = {"updates_per_s": m / t, "us_per_update": t / m * 1e6, "m": m, "N": N,
                                   "alg_GBps": m * (d * 8 + 8) / t / 1e9, "kernel": ctx.last_kernel()}
    # one full SVRG epoch with m = N (inner chain + tail + sweep)
    if not quick:
        idxN = ctx._idx(st.rand_indices(N, N))
        t = _timed(ctx, lambda: ctx.svrg_iterate(F, g, gamma, idxN, False, av, z, zf, w, reuse_rowdots=True))
        out["svrg_epoch_m=N_f64_N1M_d1024"] = {"epochs_per_s": 1.0 / t, "seconds": t}
    del F, idx
    torch.cuda.empty_cache()

    # ---- S3 on short rows (single-wave chain, no cross-wave exchange): d = 128 fp64, and BASELINE config #1's shape d = 50 -------
    for dd in (128, 50):
        N = 1_000_000 // scale
        F = _problem(ctx, dev, N, dd, torch.float64, False)
        x0 = torch.zeros(dd, dtype=torch.float64, device=dev)
        av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
        ctx.svrg_init(F, x0, av, z, zf, w)
        m = 200_000 // scale
        idx = ctx._idx(st.rand_indices(N, m))
        ctx.svrg_inner(F, g, 1.0 / (7 * 1.3 * N), idx[:1000], av, z, zf, w)
        t = _timed(ctx, lambda: ctx.svrg_inner(F, g, 1.0 / (7 * 1.3 * N), idx, av, z, zf, w))
        out[f"svrg_inner_f64_d{dd}"] = {"updates_per_s": m / t, "us_per_update": t / m * 1e6, "m": m, "N": N, "kernel": ctx.last_kernel()}
        del F, idx
        torch.cuda.empty_cache()

    # ---- G2 / G3: SAGA init (table write sweep) and SAGA steps, l1-logistic d=1024 fp32 -----------------------------
    N, d = 2_000_000 // scale, 1024
    F = _problem(ctx, dev, N, d, torch.float32, True)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    gamma = 1.0 / (3 * 0.25 * 1.3)
    x0 = torch.ones(d, dtype=torch.float32, device=dev)
    table = torch.empty((N, d), dtype=torch.float32, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, gamma, x0, table, av, z)
    ctx.timing_enable(True)
    ctx.timing_read()
    ctx.saga_init(F, g, gamma, x0, table, av, z)
    ms, n = ctx.timing_read()
    ctx.timing_enable(False)
    out["saga_init_f32_d1024"] = {"kernel_ms": ms / max(n, 1), "alg_GBps": 2 * N * d * 4 / (ms / max(n, 1) * 1e-3) / 1e9,
                                  "N": N, "kernel": ctx.last_kernel()}
    k = 200_000 // scale
    idx = ctx._idx(st.rand_indices(N, k))
    ctx.saga_steps(F, g, gamma, False, idx[:1000], table, av, z)
    t = _timed(ctx, lambda: ctx.saga_steps(F, g, gamma, False, idx, table, av, z))
    out["saga_steps_f32_d1024"] = {"updates_per_s": k / t, "us_per_update": t / k * 1e6, "steps": k, "N": N,
                                   "alg_GBps": k * (3 * d * 4 + 8) / t / 1e9, "kernel": ctx.last_kernel()}
    del F, table, idx
    torch.cuda.empty_cache()

    # ---- F2 / F3: Finito init and batches, d=4096 fp32 ---------------------------------------------------------------------
    N, d = 1_000_000 // scale, 4096
    F = _problem(ctx, dev, N, d, torch.float32, False)
    g = ProxG(L.PROX_L1, lam=1e-3)
    gam = torch.full((N,), 0.999 * N / (1.3 * N), dtype=torch.float32, device=dev)
    hg = ctx.hat_gamma(gam)
    x0 = torch.zeros(d, dtype=torch.float32, device=dev)
    table = torch.empty((N, d), dtype=torch.float32, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    for r in (1, 256, 4096, 65536 // scale):
        nit = max(4, min(20000 // scale, (1 << 18) // r))
        batches = [st.sample_without_replacement(N, r) for _ in range(nit)]
        bptr = np.arange(nit + 1, dtype=np.int64) * r
        bidx = ctx._idx(np.concatenate(batches))
        ctx.finito_steps(F, g, gam, hg, bptr[:3], bidx[:2 * r], table, av, z)
        t = _timed(ctx, lambda: ctx.finito_steps(F, g, gam, hg, bptr, bidx, table, av, z))
        out[f"finito_batch_r{r}_f32_d4096"] = {"samples_per_s": nit * r / t, "iterations_per_s": nit / t,
                                               "alg_GBps": nit * r * (3 * d * 4 + 16) / t / 1e9, "kernel": ctx.last_kernel()}
    del F, table
    torch.cuda.empty_cache()

    # ---- the sweep on row lengths outside the wave-per-row shapes: workgroup per row (masked), several rows per wave ------------
    for d_odd, tdt, tag in ((1000, torch.float64, "f64"), (50, torch.float64, "f64"), (3000, torch.float32, "f32")):
        Nn = (2_000_000 if d_odd >= 1000 else 8_000_000) // scale
        Fo = _problem(ctx, dev, Nn, d_odd, tdt, False)
        xo = torch.zeros(d_odd, dtype=tdt, device=dev)
        avo = torch.empty_like(xo)
        ctx.full_gradient(Fo, xo, avo)
        ctx.timing_enable(True)
        ctx.timing_read()
        for _ in range(5):
            ctx.full_gradient(Fo, xo, avo)
        ms, n = ctx.timing_read()
        ctx.timing_enable(False)
        es = 8 if tdt == torch.float64 else 4
        out[f"sweep_{tag}_d{d_odd}"] = {"kernel_ms": ms / max(n, 1), "alg_GBps": Nn * (d_odd * es + es) / (ms / max(n, 1) * 1e-3) / 1e9, "N": Nn,
                                        "kernel": ctx.last_kernel()}
        del Fo
        torch.cuda.empty_cache()

    # ---- adaptive Finito steps, Lasso d=1024 fp64 ------------------------------------------------------
    N, d = 200_000 // scale, 1024
    F = _problem(ctx, dev, N, d, torch.float64, False)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=torch.float64, device=dev)
    table = torch.empty((N, d), dtype=torch.float64, device=dev)
    meta4 = torch.empty((N, 4, 4), dtype=torch.float64, device=dev)
    hgd = torch.empty(1, dtype=torch.float64, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.afinito_init(F, g, 0.999, x0, table, meta4, av, z, hgd)
    k = 100_000 // scale
    idx = ctx._idx(st.rand_indices(N, k))
    ctx.afinito_steps(F, g, 0.999, 1e-9, idx[:1000], table, meta4, av, z, hgd)
    t0 = time.perf_counter()
    done, trials = ctx.afinito_steps(F, g, 0.999, 1e-9, idx, table, meta4, av, z, hgd)
    t = time.perf_counter() - t0
    out["adaptive_finito_steps_f64_d1024"] = {"updates_per_s": done / t, "us_per_update": t / max(done, 1) * 1e6, "steps": done,
                                              "trials_per_step": trials / max(done, 1), "kernel": ctx.last_kernel()}
    del F, table
    torch.cuda.empty_cache()

    # ---- ProShI agent batches: separable quadratic + soft box, d=1024 fp64 -----------------------------
    from ciaoalgorithms_jl_amd.device import PackedSepQuad
    N, d = 1_000_000 // scale, 1024
    Q = torch.empty((N, d), dtype=torch.float64, device=dev)
    q = torch.empty((N, d), dtype=torch.float64, device=dev)
    ctx.synth_normal(Q, 0, seed=7, scale=1.0)
    ctx.synth_normal(q, 0, seed=8, scale=1.0)
    Q.abs_()
    f = PackedSepQuad(Q, q, eta=30.0, lo=-2.0, hi=2.0)
    gbox = ProxG(L.PROX_BOX, lo=-float("inf"), hi=1.0)
    gam = torch.full((N,), 0.999 * N / 40.0, dtype=torch.float64, device=dev)
    x0 = torch.zeros(d, dtype=torch.float64, device=dev)
    table = torch.empty((N, d), dtype=torch.float64, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    hgd = torch.empty(1, dtype=torch.float64, device=dev)
    t = _timed(ctx, lambda: ctx.proshi_init(f, gbox, gam, x0, table, av, z, hgd))
    out["proshi_init_f64_d1024"] = {"seconds": t, "alg_GBps": N * 3 * d * 8 / t / 1e9, "N": N, "kernel": ctx.last_kernel()}
    hg = float(hgd.item())
    # small batches (the reference's default is one agent per iteration): one coordinate-parallel chain launch for the whole run;
    # the same through the batch-parallel launch pair for comparison
    for r in (1, 16):
        nit = 20000 // r // scale
        bidx = ctx._idx(st.sample_batches(N, r, nit).reshape(-1).copy())
        bptr = np.arange(nit + 1, dtype=np.int64) * r
        for lim, tag in ((-1, ""), (0, "_launch_pair")):
            if lim == 0 and r == 1:
                nn = nit // 10
            else:
                nn = nit
            ctx.set_option("proshi_chain_max_batch", lim)
            ctx.proshi_steps(f, gbox, gam, hg, bptr[:3], bidx[:2 * r], table, av, z)
            t = _timed(ctx, lambda: ctx.proshi_steps(f, gbox, gam, hg, bptr[:nn + 1], bidx[:nn * r], table, av, z))
            ctx.set_option("proshi_chain_max_batch", -1)
            out[f"proshi_batch_r{r}_f64_d1024{tag}"] = {"iterations_per_s": nn / t, "us_per_iteration": t / nn * 1e6, "agents_per_s": nn * r / t,
                                                         "kernel": ctx.last_kernel()}
    for r in (4096, 65536 // scale):
        nit = 16
        batches = [st.sample_without_replacement(N, r) for _ in range(nit)]
        bptr = np.arange(nit + 1, dtype=np.int64) * r
        bidx = ctx._idx(np.concatenate(batches))
        ctx.proshi_steps(f, gbox, gam, hg, bptr[:2], bidx[:r], table, av, z)
        t = _timed(ctx, lambda: ctx.proshi_steps(f, gbox, gam, hg, bptr, bidx, table, av, z))
        out[f"proshi_batch_r{r}_f64_d1024"] = {"agents_per_s": nit * r / t, "alg_GBps": nit * r * 4 * d * 8 / t / 1e9,
                                               "kernel": ctx.last_kernel()}
    del Q, q, table, f
    torch.cuda.empty_cache()

    # ---- ProShI with dense Quadratic(Q_i) blocks (ciao_sepquad.dense): N agents x (d x d), d = 256 fp64 -------------------
    N, d = 8192 // scale, 256
    Q = torch.empty((N * d, d), dtype=torch.float64, device=dev)
    q = torch.empty((N, d), dtype=torch.float64, device=dev)
    ctx.synth_normal(Q, 0, seed=9, scale=1.0 / np.sqrt(d))
    ctx.synth_normal(q, 0, seed=10, scale=1.0)
    f = PackedSepQuad(Q.view(N, d, d), q, eta=30.0, lo=-2.0, hi=2.0)
    gam = torch.full((N,), 0.999 * N / 40.0, dtype=torch.float64, device=dev)
    x0 = torch.zeros(d, dtype=torch.float64, device=dev)
    table = torch.empty((N, d), dtype=torch.float64, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.proshi_init(f, gbox, gam, x0, table, av, z, hgd)
    t = _timed(ctx, lambda: ctx.proshi_init(f, gbox, gam, x0, table, av, z, hgd), reps=3)
    out["proshi_dense_init_f64_d256"] = {"seconds": t, "alg_GBps": N * d * d * 8 / t / 1e9, "N": N, "kernel": ctx.last_kernel()}
    hg = float(hgd.item())
    r, nit = 1024 // scale, 16
    batches = [st.sample_without_replacement(N, r) for _ in range(nit)]
    bptr = np.arange(nit + 1, dtype=np.int64) * r
    bidx = ctx._idx(np.concatenate(batches))
    ctx.proshi_steps(f, gbox, gam, hg, bptr[:2], bidx[:r], table, av, z)
    t = _timed(ctx, lambda: ctx.proshi_steps(f, gbox, gam, hg, bptr, bidx, table, av, z))
    out[f"proshi_dense_batch_r{r}_f64_d256"] = {"agents_per_s": nit * r / t, "alg_GBps": nit * r * d * d * 8 / t / 1e9,
                                                 "kernel": ctx.last_kernel()}
    del Q, q, table, f
    torch.cuda.empty_cache()

    # ---- complex T (LOSS_LS_COMPLEX): the sweep and the SVRG chain on 512 complex entries per row (= 1024 fp64 reals) -----------
    from ciaoalgorithms_jl_amd.device import PackedF
    N, n = 1_000_000 // scale, 512
    A = torch.empty((N, 2 * n), dtype=torch.float64, device=dev)
    b = torch.empty((2 * N,), dtype=torch.float64, device=dev)
    ctx.synth_normal(A, 0, seed=11, scale=1.0 / np.sqrt(2 * n))
    ctx.synth_normal(b.view(N, 2), 0, seed=12, scale=1.0)
    Fc = PackedF.least_squares_complex(A, b, float(N))
    gc = ProxG(L.PROX_L1_COMPLEX, lam=1e-3)
    x0 = torch.zeros(2 * n, dtype=torch.float64, device=dev)
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    ctx.full_gradient(Fc, x0, av)
    ctx.timing_enable(True)
    ctx.timing_read()
    for _ in range(3):
        ctx.full_gradient(Fc, x0, av)
    ms, k = ctx.timing_read()
    ctx.timing_enable(False)
    out["sweep_complex_f64_n512"] = {"kernel_ms": ms / max(k, 1), "alg_GBps": N * (2 * n * 8 + 16) / (ms / max(k, 1) * 1e-3) / 1e9, "N": N,
                                     "kernel": ctx.last_kernel()}
    ctx.svrg_init(Fc, x0, av, z, zf, w)
    m = 20_000 // scale
    idx = ctx._idx(st.rand_indices(N, m))
    ctx.svrg_inner(Fc, gc, 1.0 / (7 * 1.3 * N), idx[:100], av, z, zf, w)
    t = _timed(ctx, lambda: ctx.svrg_inner(Fc, gc, 1.0 / (7 * 1.3 * N), idx, av, z, zf, w))
    out["svrg_inner_complex_f64_n512"] = {"updates_per_s": m / t, "us_per_update": t / m * 1e6, "m": m, "kernel": ctx.last_kernel()}
    del Fc, A, b
    torch.cuda.empty_cache()

    # ---- SVRG / SAGA chains on rows beyond 8192 elements: one chain shared by several workgroups (chain_wide_kernel), d = 32768 fp64 ---
    try:
        N, d = 4096 // scale, 32768
        F = _problem(ctx, dev, N, d, torch.float64, False)
        g = ProxG(L.PROX_L1, lam=1e-3)
        x0 = torch.zeros(d, dtype=torch.float64, device=dev)
        av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
        ctx.svrg_init(F, x0, av, z, zf, w)
        m = 20_000 // scale
        idx = ctx._idx(st.rand_indices(N, m))
        for tag, opt in (("several_workgroups", 0), ("one_workgroup", 1)):
            ctx.set_option("chain_no_wide", opt)
            mm = m if not opt else m // 10
            ctx.svrg_inner(F, g, 1e-7, idx[:100], av, z, zf, w)
            t = _timed(ctx, lambda: ctx.svrg_inner(F, g, 1e-7, idx[:mm], av, z, zf, w), reps=1)
            out[f"svrg_inner_f64_d32768_{tag}"] = {"us_per_update": t / mm * 1e6, "m": mm, "N": N, "kernel": ctx.last_kernel()}
        ctx.set_option("chain_no_wide", 0)
        # ... and the full-gradient sweep over the same rows (SVRG_basic.jl:87-92): a cluster of workgroups per row (rows_long_kernel,
        # csrc/rowsl_kernels.h) against the generic kernel it replaced (1 GB here: part of it comes out of the memory-side cache;
        # tools/long_rows_time.py measures 8 GB, profiles/r04_long_rows.txt)
        for tag, opt in (("cluster_per_row", 1), ("generic", 0)):
            ctx.set_option("long_rows", opt)
            ctx.full_gradient(F, x0, av)
            t = _timed(ctx, lambda: ctx.full_gradient(F, x0, av), reps=3)
            out[f"sweep_f64_d32768_{tag}"] = {"seconds": t, "alg_GBps": N * d * 8 / t / 1e9, "N": N, "kernel": ctx.last_kernel()}
        ctx.set_option("long_rows", 1)
        del F, idx
        torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001
        out["svrg_inner_f64_d32768"] = {"error": repr(e)}

    # ---- adaptive Finito on rows beyond the register-resident shapes (afinito_big_kernel), d = 8192 fp64 --------------------
    N, d = 20_000 // scale, 8192
    F = _problem(ctx, dev, N, d, torch.float64, False)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=torch.float64, device=dev)
    table = torch.empty((N, d), dtype=torch.float64, device=dev)
    meta4 = torch.empty((N, 4, 4), dtype=torch.float64, device=dev)
    hgd = torch.empty(1, dtype=torch.float64, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.afinito_init(F, g, 0.999, x0, table, meta4, av, z, hgd)
    k = 5_000 // scale
    idx = ctx._idx(st.rand_indices(N, k))
    ctx.afinito_steps(F, g, 0.999, 1e-9, idx[:100], table, meta4, av, z, hgd)
    t0 = time.perf_counter()
    done, trials = ctx.afinito_steps(F, g, 0.999, 1e-9, idx, table, meta4, av, z, hgd)
    t = time.perf_counter() - t0
    out["adaptive_finito_steps_f64_d8192"] = {"updates_per_s": done / t, "us_per_update": t / max(done, 1) * 1e6, "steps": done,
                                              "trials_per_step": trials / max(done, 1), "kernel": ctx.last_kernel()}
    del F, table
    torch.cuda.empty_cache()

    # ---- the read+write table modes against the ceiling of their traffic mix, not only against the 8 TB/s read peak (VERDICT r2
    # weak 6): plain copy / update / triad kernels on the same kind of box reach 5.69 / 5.59 / 5.66 TB/s (1R:1W / 2R:1W / 3R:1W,
    # profiles/r02_stream_ceilings.txt; DRAM write-credit stalls)
    mixed = {"saga_init_f32_d1024": ("copy 1R:1W", 5690.0), "finito_batch_r65536_f32_d4096": ("update 2R:1W", 5590.0),
             "finito_batch_r4096_f32_d4096": ("update 2R:1W", 5590.0), "proshi_init_f64_d1024": ("triad 3R:1W (2R:1W here)", 5590.0),
             "proshi_batch_r65536_f64_d1024": ("triad 3R:1W", 5660.0), "proshi_batch_r4096_f64_d1024": ("triad 3R:1W", 5660.0)}
    for key, (mix, ceil) in mixed.items():
        if key in out and isinstance(out[key], dict) and "alg_GBps" in out[key]:
            out[key]["mixed_traffic_ceiling"] = {"mix": mix, "GBps": ceil, "frac": out[key]["alg_GBps"] / ceil,
                                                 "frac_of_8TBps_read_peak": out[key]["alg_GBps"] / 8000.0,
                                                 "source": "profiles/r02_stream_ceilings.txt (plain streaming kernels, same pool)"}

    # ---- the full passes of K solves as ONE pass over A with K right-hand sides on the matrix cores (ciao_full_gradient_multi,
    # csrc/mrhs_kernels.h) against K single sweeps: what the epoch tails of solvers.solve_together(one_pass=True) cost.  fp64 peak:
    # 78.6 TFLOP/s (AMD's MI355X figure: matrix = vector rate in fp64); fp32 157.3 (MI355X_MICROARCH.md)
    try:
        Nm, dm = 1_000_000 // scale, 1024
        for tdt, tag, peak in ((torch.float64, "f64", 78.6), (torch.float32, "f32", 157.3)):
            Fm = _problem(ctx, dev, Nm, dm, tdt, False)
            x1 = torch.zeros(dm, dtype=tdt, device=dev)
            a1 = torch.empty_like(x1)
            ctx.full_gradient(Fm, x1, a1)
            t1 = _timed(ctx, lambda: ctx.full_gradient(Fm, x1, a1), reps=5)
            for K in (16, 256):
                xs = [torch.randn(dm, dtype=tdt, device=dev) * 0.1 for _ in range(K)]
                avs = [torch.empty_like(x) for x in xs]
                ctx.full_gradient_multi(Fm, xs, avs)
                kern = ctx.last_kernel()
                t = _timed(ctx, lambda: ctx.full_gradient_multi(Fm, xs, avs), reps=3)
                out[f"multi_rhs_full_pass_{tag}_K{K}_d1024"] = {
                    "seconds": t, "N": Nm, "TFLOPs": 4.0 * Nm * dm * K / t / 1e12, "K_single_sweeps_seconds": K * t1, "speedup": K * t1 / t,
                    "roofline": {"bound": "mfma", "achieved": 4.0 * Nm * dm * K / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                                 "frac": 4.0 * Nm * dm * K / t / 1e12 / peak}, "kernel": kern}
            if tdt == torch.float64 and not quick:
                # one whole outer step of K = 256 lockstep SVRG solves (a regularisation path) over these rows, m = N: the batched inner
                # cycles (one launch, a workgroup per solve) + the 256 epoch tails as K sweeps / as ONE pass (solve_together(one_pass=))
                K = 256
                sts = [tuple(torch.zeros(dm, dtype=tdt, device=dev) for _ in range(4)) for _ in range(K)]
                gs = [ProxG(L.PROX_L1, lam=1e-3 * (1.0 + k / K)) for k in range(K)]
                ixs = [ctx._idx(IndexStream(500 + k).rand_indices(Nm, Nm)) for k in range(K)]
                for k in range(K):
                    ctx.svrg_init(Fm, x1, *sts[k])
                ctx.set_option("svrg_cache_rowdots", 0)
                try:
                    def chains():
                        with ctx.chain_batch():
                            for k in range(K):
                                ctx.svrg_inner(Fm, gs[k], 1.0 / (7 * 1.3 * Nm), ixs[k], *sts[k])
                    t_ch = _timed(ctx, chains, reps=1)
                finally:
                    ctx.set_option("svrg_cache_rowdots", 1)
                cols = [[s[i] for s in sts] for i in range(4)]
                t_k = _timed(ctx, lambda: [ctx.svrg_epoch_tail(Fm, Nm, False, *s) for s in sts], reps=1)
                t_1 = _timed(ctx, lambda: ctx.svrg_epoch_tail_multi(Fm, Nm, False, *cols), reps=1)
                out["lambda_path_outer_step_K256_f64_d1024"] = {
                    "N": Nm, "m": Nm, "K": K, "inner_cycles_s": t_ch, "tails_as_K_sweeps_s": t_k, "tails_as_one_pass_s": t_1,
                    "outer_step_s": {"K_sweeps": t_ch + t_k, "one_pass": t_ch + t_1}, "updates_per_s_one_pass": K * Nm / (t_ch + t_1),
                    "note": "an extension beyond the reference's one-problem-per-call API (ciao_ctx_chain_batch_begin, ciao_svrg_epoch_tail_multi)"}
                del sts, ixs
            del Fm
            torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001
        out["multi_rhs_full_pass"] = {"error": repr(e)}

    # ---- K independent SVRG chains over the same A on K streams (a regularisation path): what the idle 255 CUs give through the
    # existing API.  The HIP runtime's hardware queues bound the concurrency (4 by default; tools/lambda_path.py with
    # GPU_MAX_HW_QUEUES=64 reaches 16x one chain: profiles/r03_lambda_path_streams.txt)
    if not quick:
        try:
            out["lambda_path_svrg_K_streams"] = lambda_path(dev, Ks=(1, 2, 4, 8, 16, 32), N=500_000, m=20_000)
        except Exception as e:
            out["lambda_path_svrg_K_streams"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        # ... and as ONE launch of K workgroups (Context.chain_batch): every CU runs a chain
        try:
            out["lambda_path_svrg_K_batched"] = lambda_path(dev, Ks=(1, 16, 64, 256, 1024), N=500_000, m=20_000, mode="batch")
        except Exception as e:
            out["lambda_path_svrg_K_batched"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out
