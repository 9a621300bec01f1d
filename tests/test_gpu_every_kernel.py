"""Every instantiation a dispatch path can reach is LAUNCHED at least once and held against the oracle.

The other GPU test files are written around behaviour (an algorithm, an edge case, a route) and left 476 of the library's 1282
kernels unlaunched (profiles/r05_reached_kernels_before.txt: a `rocprofv3 --kernel-trace` of the whole suite set against the
library's code objects by tools/reached_kernels.py).  This file is written around the DISPATCH TABLES instead: for every class a
launcher distinguishes -- row bytes (the J / wave-count classes), exact or masked last chunk, loss, algorithm, one allocation or a
shard table, and the options that force a fallback route (chain_four_waves, chain_no_ws, chain_no_dma, chain_big) -- one small
problem, the same index stream through the device and the oracle (SVRG_basic.jl:73-82, SAGA_basic.jl:53-68, Finito_basic.jl:97-117,
Finito_LFinito.jl:77-100, Finito_adaptive.jl:120-152 restated in oracle/), and the kernel the launcher reports is checked against
the class.  tools/exp/reached_kernels.sh re-runs the trace (every process, child ranks too): with this file the suite launches every kernel of the
library (profiles/r05_reached_kernels.txt); tests/test_kernel_reach_record.py holds the built library against that record on the CPU."""
import numpy as np
import pytest

import problems as P
from test_gpu_parity import close, dev, make, make_g

pytestmark = pytest.mark.gpu

F64, F32 = np.float64, np.float32
ALGS = ("svrg", "svrg_cached", "saga", "sag", "finito", "lfinito")
ALG_NO = {"svrg": 0, "saga": 1, "sag": 1, "finito": 2, "lfinito": 3, "svrg_cached": 4}   # chain_common.h CA_*


def _shard_table(L, dp, N, cuts, table=None):
    t = L.ShardTable()
    t.nshards, t.owner = len(cuts) - 1, 1
    for k in range(len(cuts) - 1):
        t.row0[k] = cuts[k]
        t.A[k] = dp.A[cuts[k]:].data_ptr() if cuts[k] < N else None
        t.b[k] = dp.b[cuts[k]:].data_ptr() if cuts[k] < N else None
        t.table[k] = table[cuts[k]:].data_ptr() if (table is not None and cuts[k] < N) else None
    t.row0[len(cuts) - 1] = N
    return t


def run_chain(ctx, ciao, alg, N, d, dtype, loss, gk, sharded=False, r=1):
    """One chain of `alg` on an N x d problem, device against oracle on the same index stream; -> the kernel the launcher reports."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from oracle import twin as O
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=N * 31 + d)
    lam_f = float(N) if loss == "ls" else 1.0
    op, dp = make(loss, A, b, lam_f, dtype)
    og, dg = make_g(gk, dtype, d, lam=0.02)
    Li = (lam_f if loss == "ls" else 0.25) * np.sum(A.astype(np.float64) ** 2, axis=1) + 1e-12
    tdt = dev(x0).dtype
    st = ciao.IndexStream(N * 1000 + d)
    new = lambda: torch.empty(d, dtype=tdt, device="cuda")
    cuts = [0, N // 3, N // 3, N]                     # three shards, the middle one empty
    tag = f"{alg} N={N} d={d} {np.dtype(dtype).name} {loss} {gk}{' sharded' if sharded else ''}"
    if alg in ("svrg", "svrg_cached"):
        gamma = 1.0 / (7 * Li.max())
        av, z, zf, w = new(), new(), new(), new()
        ctx.svrg_init(dp, dev(x0), av, z, zf, w)
        rav, rz, rzf, rw = O.svrg_init(op, x0)
        idx = st.rand_indices(N, 3 * N + 5)
        idx[4:7] = idx[4]                              # the same row three times in a row
        if sharded:
            ctx.set_shards(_shard_table(L, dp, N, cuts))
            try:
                ctx.svrg_inner(dp, dg, gamma, idx, av, z, zf, w)
                name = ctx.last_kernel()
                ctx.synchronize()
            finally:
                ctx.set_shards(None)
            O.svrg_inner(op, og, dtype(gamma), idx, rav, rz, rzf, rw)
            close(w, rw, dtype, scale={64: 220, 32: 200}, scale64=250, what=f"dispatch sweep {tag} w ({name})")
        elif alg == "svrg":
            ctx.svrg_inner(dp, dg, gamma, idx, av, z, zf, w)
            name = ctx.last_kernel()
            O.svrg_inner(op, og, dtype(gamma), idx, rav, rz, rzf, rw)
            close(w, rw, dtype, scale={64: 220, 32: 200}, scale64=250, what=f"dispatch sweep {tag} w ({name})")
        else:
            # whole outer iterations: the inner cycle reuses the a_i'z_full of the full pass before it (CA_SVRGC); the launcher's last
            # kernel is the full pass that closes the iteration, so no name here -- the kernel trace says what ran
            for ep in range(2):
                ctx.svrg_iterate(dp, dg, gamma, idx, False, av, z, zf, w, reuse_rowdots=True)
                O.svrg_iterate(op, og, dtype(gamma), idx, False, rav, rz, rzf, rw)
            name = None
            close(zf, rzf, dtype, scale=160, scale64=230, what=f"dispatch sweep {tag} z_full")
            close(av, rav, dtype, scale={64: 110, 32: 120}, scale64=160, what=f"dispatch sweep {tag} av")
    elif alg in ("saga", "sag"):
        gamma = 1.0 / ((16 if alg == "sag" else 3) * Li.max())
        table = torch.empty((N, d), dtype=tdt, device="cuda")
        av, z = new(), new()
        ctx.saga_init(dp, dg, gamma, dev(x0), table, av, z)
        rt, rav, rz = O.saga_init(op, og, dtype(gamma), x0)
        idx = st.rand_indices(N, 6 * N + 3)
        idx[4:7] = idx[4]
        if sharded:
            ctx.set_shards(_shard_table(L, dp, N, cuts, table))
        try:
            ctx.saga_steps(dp, dg, gamma, alg == "sag", idx, table, av, z)
            name = ctx.last_kernel()
            ctx.synchronize()
        finally:
            if sharded:
                ctx.set_shards(None)
        O.saga_steps(op, og, dtype(gamma), alg == "sag", idx, rt, rav, rz)
        close(z, rz, dtype, scale={64: 79, 32: 73}, scale64=360, what=f"dispatch sweep {tag} z ({name})")
        close(table, rt, dtype, scale={64: 400, 32: 790}, scale64=450, what=f"dispatch sweep {tag} table")
    else:
        assert not sharded
        gam = (0.999 * N / Li).astype(dtype)
        dgam = dev(gam)
        hg = ctx.hat_gamma(dgam)
        nb = -(-N // r)
        ctx.set_option("chain_max_batch", 64)
        try:
            if alg == "finito":
                table = torch.empty((N, d), dtype=tdt, device="cuda")
                av, z = new(), new()
                rt, rav, rz, rhg = O.finito_init(op, og, gam, x0)
                ctx.finito_init(dp, dg, dgam, hg, dev(x0), table, av, z)
                batches = [st.sample_without_replacement(N, r) for _ in range(2 * nb + 3)]
                batches[3] = batches[2].copy()         # a batch met again one step later
                bptr = np.zeros(len(batches) + 1, np.int64)
                np.cumsum([len(x) for x in batches], out=bptr[1:])
                ctx.finito_steps(dp, dg, dgam, hg, bptr, np.concatenate(batches), table, av, z)
                name = ctx.last_kernel()
                O.finito_steps(op, og, gam, rhg, batches, rt, rav, rz)
                close(z, rz, dtype, scale={64: 110, 32: 160}, scale64=95, what=f"dispatch sweep {tag} z ({name})")
                close(table, rt, dtype, scale={64: 110, 32: 140}, scale64=90, what=f"dispatch sweep {tag} table")
            else:
                static = [np.arange(r * j, min(r * j + r, N), dtype=np.int64) for j in range(nb)]
                av, z, zf = new(), new(), new()
                rav, rz, rzf, rhg = O.lfinito_init(op, gam, x0)
                ctx.lfinito_init(dp, hg, dev(x0), av, z, zf)
                for it in range(2):
                    order = np.arange(nb) if it == 0 else st.randperm(nb)
                    batches = [static[j] for j in order]
                    bptr = np.zeros(nb + 1, np.int64)
                    np.cumsum([len(x) for x in batches], out=bptr[1:])
                    ctx.lfinito_iterate(dp, dg, dgam, hg, bptr, np.concatenate(batches), av, z, zf)
                    O.lfinito_iterate(op, og, gam, rhg, batches, rav, rz, rzf)
                name = ctx.last_kernel()
                close(zf, rzf, dtype, scale={64: 97, 32: 93}, scale64=68, what=f"dispatch sweep {tag} z_full ({name})")
                close(av, rav, dtype, scale={64: 100, 32: 120}, scale64=100, what=f"dispatch sweep {tag} av")
        finally:
            ctx.set_option("chain_max_batch", -1)
    ctx.synchronize()
    return name


def _with(ctx, opts, fn):
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        return fn()
    finally:
        for k in opts:
            ctx.set_option(k, 0)


# ---- chain_dma_kernel / chain_ws_kernel: rows of whole 16-byte chunks up to 32 KiB ----------------------------------------------
# classes of chain_dma_launch.inc: one wave (<= 1 KiB: J1, <= 2 KiB: J2; exact = no masked chunk), four waves J1 / J2 / J4, eight waves
# (32 KiB); chain_launch.inc: SAGA on four-wave J1 rows goes to chain_ws_kernel unless chain_no_ws
ROWB = [1024, 1008, 2048, 2000, 4096, 4080, 8192, 8176, 16384, 16000, 32768, 32752]


def _dma_class(rowb):
    """(J as the launcher reports it = 4 KiB units of the row class, waves, masked) of the default route"""
    if rowb <= 1024:
        return 1, 1, rowb != 1024
    if rowb <= 2048:
        return 1, 1, rowb != 2048                      # (two chunks per lane of the one wave; reported by row class: J1)
    j = 1
    while j * 4096 < rowb:
        j *= 2
    return j, (8 if j == 8 else 4), rowb != j * 4096   # (32 KiB rows: reported as J8, eight waves of four chunks per thread)


@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
@pytest.mark.parametrize("rowb", ROWB)
def test_every_lds_dma_chain_class(ctx, ciao, dtype, loss, rowb):
    d = rowb // np.dtype(dtype).itemsize
    N = 13
    J, waves, masked = _dma_class(rowb)
    ty = "f64" if dtype == F64 else "f32"
    gk = "l1" if loss == "ls" else "boxvec"
    for alg in ALGS:
        name = run_chain(ctx, ciao, alg, N, d, dtype, loss, gk)
        if name is None:
            continue
        if alg in ("saga", "sag") and waves == 4 and J == 1:
            assert f"chain_ws_kernel<{ty},J1,alg1" in name and ("masked" in name) == masked, name
        else:
            assert f"chain_dma_kernel<{ty},J{J},alg{ALG_NO[alg]}" in name and f"block={64 * waves}" in name and ("masked" in name) == masked, name
    # the routes an option or a shard table selects: short rows on four waves, SAGA without the wave-specialised chain, shard tables
    if waves == 1:
        for alg in ALGS:
            name = _with(ctx, {"chain_four_waves": 1}, lambda: run_chain(ctx, ciao, alg, N, d, dtype, loss, gk))
            if name is None:
                continue
            want = "chain_ws_kernel" if alg in ("saga", "sag") else "chain_dma_kernel"
            assert want in name and "block=" in name and "block=64 " not in name, name
    if J == 1:
        for alg in ("saga", "sag"):
            name = _with(ctx, {"chain_no_ws": 1, "chain_four_waves": 1}, lambda: run_chain(ctx, ciao, alg, N, d, dtype, loss, gk))
            assert f"chain_dma_kernel<{ty},J1,alg1" in name and "block=256" in name, name
    for alg in ("svrg", "saga", "sag"):
        name = run_chain(ctx, ciao, alg, N, d, dtype, loss, gk, sharded=True)
        assert "sharded" in name and ("chain_ws_kernel" in name or "chain_dma_kernel" in name), name
        if alg != "svrg" and rowb <= 4096:
            name = _with(ctx, {"chain_no_ws": 1}, lambda: run_chain(ctx, ciao, alg, N, d, dtype, loss, gk, sharded=True))
            assert "chain_dma_kernel" in name and "sharded" in name, name


# ---- chain_kernel: the register-ring fallback (rows that are not whole aligned 16-byte chunks; forced here by chain_no_dma) --------
# classes of chain_launch.inc launch_chain_e: one wave (d <= 64), E = 1 / 4 / 8 / 16 / 32 elements per thread of 256 (fp64: up to 16);
# "full" = d is exactly E x threads
@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
@pytest.mark.parametrize("d", [64, 50, 256, 201, 1024, 999, 2048, 2001, 4096, 4001, 8192, 8001])
def test_every_register_ring_chain_class(ctx, ciao, dtype, loss, d):
    if dtype == F64 and d > 4096:
        pytest.skip("fp64 rows beyond 4096 elements are the several-workgroup chain's (chain_wide_kernel)")
    N = 11
    ty = "f64" if dtype == F64 else "f32"
    for alg in ALGS:
        name = _with(ctx, {"chain_no_dma": 1}, lambda: run_chain(ctx, ciao, alg, N, d, dtype, loss, "l1"))
        assert name is None or f"chain_kernel<{ty}" in name, name
        # the same shape through the any-length kernel (state in the caller's vectors)
        if d in (256, 999):
            name = _with(ctx, {"chain_big": 1}, lambda: run_chain(ctx, ciao, alg, N, d, dtype, loss, "box"))
            assert name is None or "chain_big_kernel" in name, name


# ---- adaptive Finito: afinito_dma_kernel (one wave / four waves, J = 1 ... 8, masked, over a shard table), afinito_chain_kernel ------
def run_afinito(ctx, ciao, N, d, dtype, loss, sharded=False):
    """Init + 3N backtracking steps against the oracle on the same sample sequence (unsharded), or -- over a shard table -- bitwise
    against the unsharded run of the same route; -> kernel name."""
    import torch
    import ciaoalgorithms_jl_amd._lib as L
    from oracle import twin as O
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=21 + d)
    lam_f = float(N) if loss == "ls" else 1.0
    op, dp = make(loss, A, b, lam_f, dtype)
    og, dg = make_g("l1", dtype, d, lam=0.02)
    alpha, tol_b = 0.999, 1e-9
    tdt = dev(x0).dtype
    idx = ciao.IndexStream(4 + d).rand_indices(N, 3 * N)
    idx[5:8] = idx[5]
    idx[10] = idx[8]

    def one(shards):
        table = torch.empty((N, d), dtype=tdt, device="cuda")
        meta4 = torch.empty((N, 4, 4), dtype=tdt, device="cuda")
        hg = torch.empty(1, dtype=tdt, device="cuda")
        av, z = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
        ctx.afinito_init(dp, dg, alpha, dev(x0), table, meta4, av, z, hg)
        ctx.synchronize()
        parts = None
        if shards:
            cuts = [0, N // 3, N // 3, N]
            parts = [(dp.A[c0:c1].clone(), dp.b[c0:c1].clone(), table[c0:c1].clone(), meta4[c0:c1].clone()) for c0, c1 in zip(cuts, cuts[1:])]
            t = L.ShardTable()
            t.nshards, t.owner = len(cuts) - 1, 1
            for k, (pa, pb, pt, pm) in enumerate(parts):
                t.row0[k] = cuts[k]
                if pa.shape[0]:
                    t.A[k], t.b[k], t.table[k], t.meta[k] = pa.data_ptr(), pb.data_ptr(), pt.data_ptr(), pm.data_ptr()
            t.row0[len(cuts) - 1] = N
            ctx.set_shards(t)
        try:
            done, trials = ctx.afinito_steps(dp, dg, alpha, tol_b, idx, table, meta4, av, z, hg)
            name = ctx.last_kernel()
            ctx.synchronize()
        finally:
            if shards:
                ctx.set_shards(None)
        if shards:
            table = torch.cat([p_[2] for p_ in parts])
            meta4 = torch.cat([p_[3] for p_ in parts])
        return done, trials, z, av, hg, table, meta4, name

    if sharded:
        # (a sharded chain runs on four waves whatever the row length: so does its twin)
        ctx.set_option("chain_four_waves", 1)
        try:
            u, v = one(False), one(True)
        finally:
            ctx.set_option("chain_four_waves", 0)
        assert u[0] == v[0] == len(idx) and u[1] == v[1], (u[:2], v[:2])
        for x, y in zip(u[2:7], v[2:7]):
            assert torch.equal(x, y), "the chain over a shard table is not bitwise the unsharded one"
        assert "sharded" in v[7], v[7]
        return v[7]
    done, trials, z, av, hg, table, meta4, name = one(False)
    rt, rg, rgam, rfi, rav, rz, rhg = O.afinito_init(op, og, dtype(alpha), x0)
    rdone, rhg, rtrials = O.afinito_steps(op, og, dtype(alpha), dtype(tol_b), idx, rt, rg, rgam, rfi, rhg, rav, rz)
    assert done == rdone == len(idx)
    # Float32: a backtracking test f_i(z) <= model + tol that sits on its boundary to within the rounding of the dot products may be
    # decided differently, and every later decision of these 39 steps with it (test_adaptive_finito_steps: how rarely); Float64: never
    assert abs(trials - rtrials) <= (0 if dtype == F64 else max(2, rtrials // 10)), (trials, rtrials)
    if trials == rtrials:
        close(z, rz, dtype, scale={64: 360, 32: 340}, what=f"dispatch sweep adaptive z N={N} d={d} {loss} ({name})")
        close(table, rt, dtype, scale=300, what=f"dispatch sweep adaptive table d={d} {loss}")
        close(hg, [rhg], dtype, scale={64: 160, 32: 130}, what=f"dispatch sweep adaptive hat_gamma d={d} {loss}")
    return name


@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
@pytest.mark.parametrize("loss", ["ls", "logistic"])
@pytest.mark.parametrize("rowb", ROWB)
def test_every_adaptive_finito_class(ctx, ciao, dtype, loss, rowb):
    d = rowb // np.dtype(dtype).itemsize
    N = 13
    ty = "f64" if dtype == F64 else "f32"
    j = 1
    while j * 4096 < rowb:
        j *= 2
    masked = rowb != j * 4096
    name = run_afinito(ctx, ciao, N, d, dtype, loss)
    assert f"afinito_dma_kernel<{ty},J{j}" in name and ("masked" in name) == masked, name
    if "block=64 " in name:   # short rows ran on one wave: the four-wave kernel of the same class too
        name = _with(ctx, {"chain_four_waves": 1}, lambda: run_afinito(ctx, ciao, N, d, dtype, loss))
        assert "afinito_dma_kernel" in name and "block=256" in name, name
    name = run_afinito(ctx, ciao, N, d, dtype, loss, sharded=True)
    assert f"afinito_dma_kernel<{ty},J{j}" in name, name
    # the compiler-scheduled fallback for rows that are not whole aligned chunks (forced): E = 1 / 4 / 8 / 16 elements per thread
    if rowb // np.dtype(dtype).itemsize <= 4096:
        name = _with(ctx, {"chain_no_dma": 1}, lambda: run_afinito(ctx, ciao, N, d, dtype, loss))
        assert "afinito_chain_kernel" in name, name


# ---- the batch-parallel kernels: every mode on every row class --------------------------------------------------------------------
def run_modes(ctx, ciao, N, d, dtype, loss, r, pad=0):
    """All six modes of the rows kernels on an N x d problem (plan_rows, rows_launch.inc): the full gradient, SAGA init, Finito init,
    the adaptive init, Finito batches and LFinito iterations with batches of r rows given as index lists and as row blocks; each
    against the oracle.  -> {mode: kernel name}."""
    import torch
    from oracle import twin as O
    A, b, x0 = P.synthetic(loss, N, d, dtype, seed=d + N)
    lam_f = float(N) if loss == "ls" else 1.0
    op, dp = make(loss, A, b, lam_f, dtype, pad=pad)
    og, dg = make_g("l1", dtype, d, lam=0.02)
    Li = (lam_f if loss == "ls" else 0.25) * np.sum(A.astype(np.float64) ** 2, axis=1)
    gam = (0.999 * N / np.maximum(Li, 1e-3 * Li.max())).astype(dtype)
    tdt = dev(x0).dtype
    dgam = dev(gam)
    hg = ctx.hat_gamma(dgam)
    new = lambda: torch.empty(d, dtype=tdt, device="cuda")
    names = {}
    tag = f"N={N} d={d} {np.dtype(dtype).name} {loss} pad={pad}"
    ctx.set_option("chain_max_batch", 0)              # batches of any size on the rows kernels, none as a chain
    try:
        av = new()
        ctx.full_gradient(dp, dev(x0), av)
        names["grad"] = ctx.last_kernel()
        close(av, O.full_pass(op, x0), dtype, scale={64: 450, 32: 390}, scale64=16, what=f"every mode {tag}: full gradient ({names['grad']})")
        table = torch.empty((N, d), dtype=tdt, device="cuda")
        sav, sz = new(), new()
        g0 = 0.1 / max(Li.max(), 1.0)
        ctx.saga_init(dp, dg, g0, dev(x0), table, sav, sz)
        names["saga_init"] = ctx.last_kernel()
        rt, rav, rz = O.saga_init(op, og, dtype(g0), x0)
        close(table, rt, dtype, scale={64: 87, 32: 160}, scale64=13, what=f"every mode {tag}: saga_init table")
        close(sav, rav, dtype, scale={64: 67, 32: 100}, scale64=15, what=f"every mode {tag}: saga_init av")
        z = new()
        rt, rav, rz, rhg = O.finito_init(op, og, gam, x0)
        ctx.finito_init(dp, dg, dgam, hg, dev(x0), table, av, z)
        names["finito_init"] = ctx.last_kernel()
        close(table, rt, dtype, scale=17, scale64=11, what=f"every mode {tag}: finito_init table")
        close(av, rav, dtype, scale={64: 110, 32: 80}, scale64=11, what=f"every mode {tag}: finito_init av")
        # Finito batches: random lists, then static blocks (the last one short) as row blocks
        st = ciao.IndexStream(d)
        rnd = [st.sample_without_replacement(N, r) for _ in range(3)] + [st.sample_without_replacement(N, max(1, r // 3))]
        bptr = np.zeros(len(rnd) + 1, np.int64)
        np.cumsum([len(x) for x in rnd], out=bptr[1:])
        ctx.finito_steps(dp, dg, dgam, hg, bptr, np.concatenate(rnd), table, av, z)
        names["finito_lists"] = ctx.last_kernel()
        O.finito_steps(op, og, gam, rhg, rnd, rt, rav, rz)
        close(z, rz, dtype, scale={64: 1300, 32: 15000}, scale64=32, what=f"every mode {tag}: finito z, lists ({names['finito_lists']})")
        close(table, rt, dtype, scale={64: 710, 32: 10000}, scale64=31, what=f"every mode {tag}: finito table, lists")
        nb = -(-N // r)
        blocks = [np.arange(r * j, min(r * j + r, N), dtype=np.int64) for j in range(nb)]
        ctx.finito_steps_blocks(dp, dg, dgam, hg, np.array([x[0] for x in blocks]), np.array([len(x) for x in blocks]), table, av, z)
        names["finito_blocks"] = ctx.last_kernel()
        O.finito_steps(op, og, gam, rhg, blocks, rt, rav, rz)
        close(z, rz, dtype, scale={64: 2500, 32: 28000}, scale64=64, what=f"every mode {tag}: finito z, blocks ({names['finito_blocks']})")
        close(table, rt, dtype, scale={64: 1100, 32: 16000}, scale64=55, what=f"every mode {tag}: finito table, blocks")
        # LFinito: the full pass + the batch sweep with two dot products per row, lists then blocks
        lav, lz, lzf = new(), new(), new()
        rav, rz, rzf, rhg = O.lfinito_init(op, gam, x0)
        ctx.lfinito_init(dp, hg, dev(x0), lav, lz, lzf)
        bp = np.zeros(nb + 1, np.int64)
        np.cumsum([len(x) for x in blocks], out=bp[1:])
        ctx.lfinito_iterate(dp, dg, dgam, hg, bp, np.concatenate(blocks), lav, lz, lzf)
        names["lfinito_lists"] = ctx.last_kernel()
        O.lfinito_iterate(op, og, gam, rhg, blocks, rav, rz, rzf)
        ctx.lfinito_iterate_blocks(dp, dg, dgam, hg, np.array([x[0] for x in blocks]), np.array([len(x) for x in blocks]), lav, lz, lzf)
        names["lfinito_blocks"] = ctx.last_kernel()
        O.lfinito_iterate(op, og, gam, rhg, blocks, rav, rz, rzf)
        close(lz, rz, dtype, scale={64: 1700, 32: 19000}, scale64=100, what=f"every mode {tag}: lfinito z ({names['lfinito_blocks']})")
        close(lav, rav, dtype, scale={64: 1600, 32: 28000}, scale64=86, what=f"every mode {tag}: lfinito av")
        # the adaptive init (Finito_adaptive.jl:59-93)
        meta4 = torch.empty((N, 4, 4), dtype=tdt, device="cuda")
        hgd = torch.empty(1, dtype=tdt, device="cuda")
        ctx.afinito_init(dp, dg, 0.999, dev(x0), table, meta4, av, z, hgd)
        names["afinito_init"] = ctx.last_kernel()
        rt, rg, rgam, rfi, rav, rz, rhg = O.afinito_init(op, og, dtype(0.999), x0)
        close(av, rav, dtype, scale={64: 100, 32: 96}, scale64=20, what=f"every mode {tag}: adaptive init av ({names['afinito_init']})")
        close(meta4[:, 0, 1], rfi, dtype, scale={64: 150, 32: 190}, scale64=18, what=f"every mode {tag}: adaptive init f_i(x0)")
    finally:
        ctx.set_option("chain_max_batch", -1)
    ctx.synchronize()
    return names


@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
@pytest.mark.parametrize("K", [1, 2, 4, 8, 16])
def test_every_wave_per_row_and_workgroup_per_row_class(ctx, ciao, dtype, K):
    """Rows of exactly K x 64 16-byte chunks: rows_fast_kernel / rows_multi_kernel (one wave per row) with and without the prefetch,
    the batches on rows_split_kernel (a workgroup per row, the default for batches up to 16 384 rows) and on the wave-per-row kernels
    (split_max_rows = 0); the workgroup-per-row kernel for every mode (split_all); the any-shape kernel (force_generic)."""
    d = K * 64 * (16 // np.dtype(dtype).itemsize)
    N, r = 700, 90
    loss = "logistic" if K in (2, 8) else "ls"
    n = run_modes(ctx, ciao, N, d, dtype, loss, r)
    assert "rows_fast_kernel" in n["grad"] or "rows_multi_kernel" in n["grad"], n
    # (batches up to 16 384 rows of exactly J x 256 chunks take a workgroup per row; 64 and 128 chunks stay one wave per row)
    assert ("rows_split_kernel" if K >= 4 else "rows_fast_kernel") in n["finito_lists"], n
    for opts in ({"sweep_prefetch": 1}, {"sweep_multi": 0, "split_max_rows": 0}, {"sweep_multi": 0, "sweep_prefetch": 1, "split_max_rows": 0}):
        try:
            n = _with(ctx, opts, lambda: run_modes(ctx, ciao, N, d, dtype, loss, r))
        finally:
            ctx.set_option("sweep_multi", 1), ctx.set_option("split_max_rows", -1), ctx.set_option("sweep_prefetch", -1)
        assert "rows_fast_kernel" in n["grad"] or "rows_multi_kernel" in n["grad"], (opts, n)
        if "split_max_rows" in opts:
            assert "rows_fast_kernel" in n["finito_lists"] and "rows_fast_kernel" in n["lfinito_blocks"], (opts, n)
    n = _with(ctx, {"split_all": 1}, lambda: run_modes(ctx, ciao, N, d, dtype, loss, r))
    assert "rows_split_kernel" in n["grad"] and "rows_split_kernel" in n["finito_init"], n
    n = _with(ctx, {"force_generic": 1}, lambda: run_modes(ctx, ciao, 300, d, dtype, loss, 40))
    assert "rows_generic_kernel" in n["grad"] and "rows_generic_kernel" in n["finito_blocks"], n


@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
@pytest.mark.parametrize("J", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("kind", ["chunks", "elements", "padded"])
def test_every_masked_workgroup_per_row_class(ctx, ciao, dtype, J, kind):
    """Rows that are not K x 64 chunks: whole 16-byte chunks with a masked tail, single elements (an odd d, or a row stride that
    breaks the alignment), J = 1 ... 16 chunks or elements per thread of the workgroup."""
    vec = 16 // np.dtype(dtype).itemsize
    if kind == "chunks":
        d = (J * 256 - 37) * vec                       # whole chunks, masked tail
        if J == 1:
            d = 150 * vec
        pad = 0
    else:
        d = J * 256 - 5 if J > 1 else 257              # single elements: 256 < d <= 4096
        d |= 1
        pad = 0 if kind == "elements" else 3
    N, r = 260, 50
    n = run_modes(ctx, ciao, N, d, dtype, "ls" if J % 4 else "logistic", r, pad=pad)
    assert "rows_split_kernel" in n["grad"] and "rows_split_kernel" in n["finito_lists"] and "rows_split_kernel" in n["lfinito_blocks"], n
    assert ("scalar" in n["grad"]) == (kind != "chunks"), n


@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
def test_remaining_rows_classes(ctx, ciao, dtype):
    """What the two sweeps above leave: exact J x 256 chunks beyond the wave-per-row shapes (J = 8, 16: 32 and 64 x 64 chunks), 256 single
    elements (J = 1 of the element form), the wave-per-row kernels without the prefetch and the two-row form on the batch sweep, the
    any-shape kernel with one wave per workgroup (rows of 24-48 KiB with no 16-byte structure), padded short rows at 16 elements
    per lane, the several-rows-per-wave batch kernel that rows_wrow_kernel replaced as the default (small_wrow = 0)."""
    vec = 16 // np.dtype(dtype).itemsize
    ty = "f64" if dtype == F64 else "f32"
    for J in (8, 16):
        n = run_modes(ctx, ciao, 120, J * 256 * vec, dtype, "ls", 30)
        assert f"rows_split_kernel<{ty},J{J}," in n["grad"] and f"rows_split_kernel<{ty},J{J}," in n["finito_lists"], n
    n = run_modes(ctx, ciao, 260, 256, dtype, "logistic", 50, pad=3)
    assert f"rows_split_kernel<{ty},J1," in n["grad"] and "scalar" in n["grad"], n
    d2 = 2 * 64 * vec
    for opts in ({"sweep_multi": 0, "sweep_prefetch": 0}, {"split_max_rows": 0}):
        try:
            n = _with(ctx, opts, lambda: run_modes(ctx, ciao, 700, d2 * (2 if "split_max_rows" in opts else 1), dtype, "ls", 90))
        finally:
            ctx.set_option("sweep_multi", 1), ctx.set_option("split_max_rows", -1), ctx.set_option("sweep_prefetch", -1)
        assert ("rows_fast_kernel" in n["grad"]) == ("sweep_multi" in opts), (opts, n)
    d_nw1 = 5001 if dtype == F64 else 9001
    n = run_modes(ctx, ciao, 40, d_nw1, dtype, "ls", 7)
    assert "rows_generic_kernel" in n["grad"] and "NW1" in n["lfinito_lists"], n
    n = _with(ctx, {"small_i": 16}, lambda: run_modes(ctx, ciao, 500, 50, dtype, "ls", 60, pad=3))
    assert "rows_small_kernel" in n["grad"] and "I16" in n["grad"], n
    for pad in (0, 3):
        try:
            n = _with(ctx, {"small_wrow": 0}, lambda: run_modes(ctx, ciao, 900, 50, dtype, "logistic", 130, pad=pad))
        finally:
            ctx.set_option("small_wrow", -1)
        assert "rows_smallb_kernel" in n["finito_lists"] and ("rows_smallb_kernel" in n["lfinito_lists"] or "rows_smallm_kernel" in n["lfinito_lists"]), n
    try:   # ... and on dense blocks of rows too short for the matrix-core tiles
        n = _with(ctx, {"small_wrow": 0}, lambda: run_modes(ctx, ciao, 900, 5, dtype, "ls", 130))
    finally:
        ctx.set_option("small_wrow", -1)
    assert "rows_smallb_kernel" in n["finito_blocks"] and "rows_smallb_kernel" in n["lfinito_blocks"], n


@pytest.mark.parametrize("dtype", [F64, F32], ids=["f64", "f32"])
@pytest.mark.parametrize("nc2", [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_matrix_core_tile_class(ctx, ciao, dtype, nc2):
    """rows_smallm_kernel: dense rows of 32 (nc2 - 1) < d <= 32 nc2 elements, every mode (fp64 tiles reach 144 elements)."""
    d = 32 * nc2 - 3 if nc2 > 1 else 20
    if dtype == F64 and nc2 > 5:
        pytest.skip("fp64 rows beyond 144 elements do not fit LDS with two tile buffers per wave")
    # the longest rows whose Finito batches still run here (three tile buffers per wave: d <= 98 fp64 / 196 fp32, plan_rows), and the
    # longest fp64 rows at all
    d = {(F64, 4): 98, (F64, 5): 141, (F32, 7): 196}.get((dtype, nc2), d)   # (141: whole chunks of 64 and more would be the workgroup-per-row kernel's)
    # (N a multiple of the batch and the batch of 16: every block starts on a 16-byte boundary whatever d; Finito batches of up to 8192
    # rows are rows_wrow_kernel's even as row blocks: 9008 rows per batch)
    n = run_modes(ctx, ciao, 18016, d, dtype, "ls" if nc2 % 2 else "logistic", 9008)
    assert "rows_smallm_kernel" in n["grad"] and "rows_smallm_kernel" in n["lfinito_blocks"], n
    if d <= (98 if dtype == F64 else 196):
        assert "rows_smallm_kernel" in n["finito_blocks"], n


def test_the_other_files_cases_on_the_classes_they_leave(ctx, ciao):
    """Complex chains (chain_cdma_kernel J = 1 / 2 / 4 exact and masked, chain_cplx_reg_kernel with 1 ... 8 entries per thread for the
    Finito chains), the complex workgroup-per-row kernel's J = 4 / 8 / 16 classes, ProShI's vector kernel at J = 4 / 8 and its scalar kernel
    with one wave per workgroup: the test bodies of tests/test_gpu_complex.py and tests/test_gpu_parity.py, run on the shapes those
    files' own parameter lists do not hold."""
    import test_gpu_complex as TC
    import test_gpu_parity as TP
    C128, C64 = np.complex128, np.complex64
    for ctype, n in ((C128, 256), (C128, 1024), (C64, 1024), (C64, 1500)):
        TC.test_complex_lds_dma_chain_is_dispatched_and_bitwise_the_register_ring(ctx, ciao, ctype, n)
    for ctype, n in ((C128, 256), (C128, 400), (C128, 1024), (C64, 1024), (C64, 1500), (C64, 2048)):
        TC.test_complex_finito_and_lfinito(ctx, ciao, ctype, (14, n), 1, "chain")
    for ctype in (C128, C64):
        for n in (200, 500, 1000, 2000):
            _with(ctx, {"chain_no_dma": 1}, lambda: TC.test_complex_finito_and_lfinito(ctx, ciao, ctype, (14, n), 1, "chain"))
    for ctype, n in ((C128, 2000), (C64, 1500), (C64, 5000)):
        TC.test_complex_finito_and_lfinito(ctx, ciao, ctype, (60, n), 20, "rows")
    TC.test_complex_full_pass_and_proxgrad(ctx, C64, (60, 5000))
    TC.test_complex_saga_steps(ctx, ciao, C64, False, (60, 5000))
    for dtype, shape, r, generic in ((F64, (600, 4096), 500, 0), (F32, (600, 3000), 500, 0), (F32, (300, 8192), 200, 0), (F32, (60, 8001), 50, 1)):
        TP.test_proshi_steps(ctx, ciao, dtype, shape, r, generic)


def test_the_all_reduce_hook_path_in_one_process(ctx, ciao):
    """finalize -> hook -> epilogue_kernel: the native-collective path of config #4 (tests/test_gpu_multirank.py runs it in child
    processes, which a kernel trace of the suite does not see).  A hook that adds nothing -- a world of one rank -- is called once per
    step with the d + 1 raw sums and leaves the fused path's result."""
    import torch
    from oracle import twin as O
    for dtype in (F64, F32):
        N, d = 500, 1024
        A, b, x0 = P.synthetic("ls", N, d, dtype, seed=5)
        op, dp = make("ls", A, b, float(N), dtype)
        og, dg = make_g("l1", dtype, d, lam=0.01)
        tdt = dev(x0).dtype
        av0, y0 = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
        ctx.proxgrad_step(dp, dg, 0.3, dev(x0), av0, y0)
        calls = []

        def hook(ptr, count, dt, stream):
            calls.append(count)
            return 0
        ctx.set_allreduce(hook)
        try:
            av1, y1 = torch.empty(d, dtype=tdt, device="cuda"), torch.empty(d, dtype=tdt, device="cuda")
            ctx.proxgrad_step(dp, dg, 0.3, dev(x0), av1, y1)
            ctx.synchronize()
        finally:
            ctx.set_allreduce(None)
        assert calls == [d + 1], calls
        close(av1, av0.cpu().numpy(), dtype, scale=8, what="hook path against the fused finalize: av")
        close(y1, y0.cpu().numpy(), dtype, scale=8, what="hook path against the fused finalize: y")
        close(av1, O.full_pass(op, x0), dtype, scale={64: 67, 32: 57}, scale64=10, what="hook path: full gradient")
