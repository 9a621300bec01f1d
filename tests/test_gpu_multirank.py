"""Row-sharded path on real hardware: two ranks share the one GPU of the test box (gloo rendezvous on 127.0.0.1; the
all-reduce hook stages through host memory because RCCL refuses two ranks on one device).  Everything else -- the
sharded rows kernels, the raw-sum hand-off to the hook, the epilogue after the reduce -- is the product path that the
8-GPU run uses with backend "nccl".  Checked against the CPU oracle on the WHOLE problem."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
    from ciaoalgorithms_jl_amd.parallel import AllReduceHook, shard_rows
    from ciaoalgorithms_jl_amd.solvers import Finito
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    from oracle import oracle as O
    from oracle import ref_solvers as RS
    import problems as P
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = {}
    try:
        dev = torch.device("cuda", 0)
        ctx = Context(0)
        hook = AllReduceHook(dev)
        ctx.set_allreduce(hook)
        N, d = 403, 1024
        A, b, x = P.synthetic("ls", N, d, np.float64, seed=12)
        row0, n = shard_rows(N, rank, world)
        F = PackedF(L.LOSS_LS, torch.from_numpy(A[row0:row0 + n]).to(dev), torch.from_numpy(b[row0:row0 + n]).to(dev), float(N),
                    N_total=N, row0=row0)
        g = ProxG(L.PROX_L1, lam=0.01)
        og, op = O.Prox("l1", lam=0.01), O.Problem("ls", A, b, float(N))
        xd = torch.from_numpy(x).to(dev)
        av, y = torch.empty_like(xd), torch.empty_like(xd)
        ctx.proxgrad_step(F, g, 0.05 / N, xd, av, y)
        rav = O.full_pass(op, x)
        ry = O.prox(og, x - (0.05 / N) * rav, 0.05 / N)
        ok["av"] = bool(np.abs(av.cpu().numpy() - rav).max() <= 1e-10 * np.abs(rav).max())
        ok["y"] = bool(np.abs(y.cpu().numpy() - ry).max() <= 1e-10 * max(np.abs(ry).max(), 1e-30) + 1e-14)
        ok["hook_calls"] = hook.calls == 1
        obj = ctx.objective(F, g, xd)
        ok["objective"] = bool(abs(obj - O.objective(op, og, x)) <= 1e-9 * abs(obj))
        # sharded Finito (batch-parallel: each rank updates the table rows it owns) and LFinito, through the solver API
        Li = float(N) * np.sum(A * A, axis=1)
        for lf in (False, True):
            solver = Finito(np.float64, maxit=6, sweeping=2, minibatch=(True, 64), LFinito=lf)
            xs, it = solver(np.zeros(d), F=F, g=g, L=Li, N=N, ctx=ctx, stream=IndexStream(0))
            xr, _ = RS.finito(op, og, np.zeros(d), maxit=6, sweeping=2, batch=64, lfinito=lf, L=Li, stream=IndexStream(0))
            ok[f"finito_lf{int(lf)}"] = bool(np.abs(xs - xr).max() <= 1e-9 * max(np.abs(xr).max(), 1e-30) + 1e-13)
        # the same with round-robin row ownership: every contiguous batch of 64 now has members on both ranks
        Fc = PackedF(L.LOSS_LS, torch.from_numpy(np.ascontiguousarray(A[rank::world])).to(dev),
                     torch.from_numpy(np.ascontiguousarray(b[rank::world])).to(dev), float(N), N_total=N, cyclic=(rank, world))
        for lf in (False, True):
            solver = Finito(np.float64, maxit=6, sweeping=3, minibatch=(True, 64), LFinito=lf)
            xs, it = solver(np.zeros(d), F=Fc, g=g, L=Li, N=N, ctx=ctx, stream=IndexStream(1))
            xr, _ = RS.finito(op, og, np.zeros(d), maxit=6, sweeping=3, batch=64, lfinito=lf, L=Li, stream=IndexStream(1))
            ok[f"finito_cyclic_lf{int(lf)}"] = bool(np.abs(xs - xr).max() <= 1e-9 * max(np.abs(xr).max(), 1e-30) + 1e-13)
        # replicas stay bitwise identical across ranks
        gathered = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, y.cpu())
        ok["bitwise_replicas"] = all(torch.equal(gathered[0], t) for t in gathered)
        # the sequential chains refuse a row-sharded problem (replicas only)
        try:
            ctx.svrg_inner(F, g, 0.1, np.zeros(3, np.int64), av, y, xd, torch.empty_like(xd))
            ok["chain_refused"] = False
        except L.CiaoError:
            ok["chain_refused"] = True
        # ---- ... unless a shard table says where every rank's rows live: then ONE chain runs on the owner (rank 0) and reads
        # rank 1's rows through an IPC-mapped pointer (here: the same GPU; on a node: a peer GPU over xGMI).  BASELINE config #4
        # as a real SVRG solve; SURVEY.md 8e option (a).
        from ciaoalgorithms_jl_amd.parallel import ShardGroup
        from ciaoalgorithms_jl_amd import solvers as S
        gamma = 1.0 / (7 * float(Li.max()))
        grp = ShardGroup(ctx, owner=0)
        it = iter(S.iterator(S.SVRG(np.float64, γ=gamma), np.zeros(d), F=F, g=g, N=N, ctx=ctx, stream=IndexStream(3), shards=grp))
        rit = iter(RS.SVRGIterable(op, og, np.zeros(d), gamma=gamma, stream=IndexStream(3)))
        worst = 0.0
        for _ in range(4):
            sd, sr = next(it), next(rit)
            worst = max(worst, float(np.abs(sd.z_full.cpu().numpy() - sr.z_full).max() / max(np.abs(sr.z_full).max(), 1e-30)))
        ok["sharded_svrg_vs_oracle"] = bool(worst <= 1e-11)
        ok["sharded_svrg_kernel"] = ("chain_dma_kernel" in ctx.last_kernel() or "rows_" in ctx.last_kernel())
        gathered = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, sd.z_full.cpu())
        ok["sharded_svrg_replicas_bitwise"] = all(torch.equal(gathered[0], t) for t in gathered)
        # the chain itself is BITWISE the single-GPU chain: rank 0 also holds the whole problem and runs the same inner cycle alone
        idxg = IndexStream(9).rand_indices(N, 2 * N)
        avs, zs, zfs, ws = (torch.empty(d, dtype=torch.float64, device=dev) for _ in range(4))
        ctx.svrg_init(F, xd, avs, zs, zfs, ws)                      # sharded full pass (all-reduced): the same av on both ranks
        ctx.svrg_inner(F, g, gamma, idxg, avs, zs, zfs, ws)         # sharded: owner's chain over both shards + broadcast
        if rank == 0:
            solo = Context(0)
            Fw = PackedF(L.LOSS_LS, torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev), float(N))
            z1, zf1, w1 = torch.zeros_like(xd), xd.clone(), xd.clone()
            solo.svrg_inner(Fw, g, gamma, idxg, avs, z1, zf1, w1)   # the SAME av, the whole matrix on one device
            solo.synchronize()
            ok["sharded_chain_bitwise_equals_single_gpu_chain"] = bool(torch.equal(w1, ws) and torch.equal(z1, zs))
            solo.close()
        wall = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(wall, ws.cpu())
        ok["sharded_chain_result_on_every_rank"] = all(torch.equal(wall[0], t) for t in wall)
        # SAGA: the gradient table is sharded too; rank 0's chain reads and writes rank 1's table rows
        grp.close()
        grp2 = ShardGroup(ctx, owner=0)
        sit = iter(S.iterator(S.SAGA(np.float64, γ=gamma), np.zeros(d), F=F, g=g, N=N, ctx=ctx, stream=IndexStream(5), shards=grp2))
        rsit = iter(RS.SAGAIterable(op, og, np.zeros(d), gamma=gamma, stream=IndexStream(5)))
        ss = sr = None
        for _ in range(300):
            ss, sr = next(sit), next(rsit)
        ok["sharded_saga_vs_oracle"] = bool(np.abs(ss.z.cpu().numpy() - sr.z).max() <= 1e-11 * max(np.abs(sr.z).max(), 1e-30))
        row0, n = shard_rows(N, rank, world)
        ok["sharded_saga_table_shard"] = bool(np.abs(ss.s.cpu().numpy() - sr.s[row0:row0 + n]).max() <= 1e-11 * np.abs(sr.s).max())
        grp2.close()
        # ---- adaptive Finito (Finito_adaptive.jl:59-155) on the row-sharded problem: every rank initialises its own rows' table rows and
        # scalars (one all-reduced sweep), rank 0's chain then reads and writes rank 1's table rows and scalars through IPC-mapped pointers
        for sweeping in (1, 3):
            grp3 = ShardGroup(ctx, owner=0)
            xs, nit = S.Finito(np.float64, maxit=3 * N, sweeping=sweeping, adaptive=True)(np.zeros(d), F=F, g=g, L=Li, N=N, ctx=ctx,
                                                                                       stream=IndexStream(30 + sweeping), shards=grp3)
            ok[f"sharded_afinito_kernel_sw{sweeping}"] = (rank != 0) or ("afinito_dma_kernel" in ctx.last_kernel() and "sharded" in ctx.last_kernel())
            xr, rit_n = RS.finito(op, og, np.zeros(d), maxit=3 * N, sweeping=sweeping, adaptive=True, L=Li, stream=IndexStream(30 + sweeping))
            ok[f"sharded_afinito_sw{sweeping}"] = bool(nit == rit_n and np.abs(xs - xr).max() <= 1e-9 * max(np.abs(xr).max(), 1e-30))
            gathered = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(gathered, torch.from_numpy(np.ascontiguousarray(xs)))
            ok[f"sharded_afinito_replicas_bitwise_sw{sweeping}"] = all(torch.equal(gathered[0], t) for t in gathered)
            grp3.close()
        # ... with samples whose Lipschitz probe at x0 .+ 1 is degenerate on BOTH ranks (:78-85): every rank draws the +-1 vectors from
        # its replicated stream in increasing global i, the rank that holds the sample probes and its result decides for all
        Ad = A.copy()
        for i in (3, 57, 250, 402):
            Ad[i] = 0
            Ad[i, 5], Ad[i, 6] = 0.7, -0.7
        opd = O.Problem("ls", Ad, b, float(N))
        row0, n = shard_rows(N, rank, world)
        Fd = PackedF(L.LOSS_LS, torch.from_numpy(Ad[row0:row0 + n]).to(dev), torch.from_numpy(b[row0:row0 + n]).to(dev), float(N),
                     N_total=N, row0=row0)
        draws = IndexStream(123)
        signs = np.stack([draws.rand_signs(d).astype(np.float64) for _ in range(64)])
        rt, rg, rgam, rfi, rav, rz, rhg = O.afinito_init(opd, og, np.float64(0.999), np.zeros(d), retry_signs=signs)
        grp4 = ShardGroup(ctx, owner=0)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            sd = next(iter(S.iterator(S.Finito(np.float64, adaptive=True), np.zeros(d), F=Fd, g=g, N=N, ctx=ctx, stream=IndexStream(123), shards=grp4)))
        ok["sharded_afinito_reprobe_gamma"] = bool(np.abs(sd.γ.cpu().numpy() - rgam[row0:row0 + n]).max() <= 1e-10 * np.abs(rgam).max()
                                                   and float(sd.γ.min()) > 0)
        ok["sharded_afinito_reprobe_av"] = bool(np.abs(sd.av.cpu().numpy() - rav).max() <= 1e-11 * np.abs(rav).max() + 1e-14)
        ok["sharded_afinito_reprobe_hg"] = bool(abs(sd.hat_γ - float(rhg)) <= 1e-12 * float(rhg))
        idxa = IndexStream(5).rand_indices(N, 500)
        done, trials = ctx.afinito_steps(Fd, g, 0.999, 1e-9, idxa, sd.s, sd.meta, sd.av, sd.z, sd.hat_γ_dev)
        rdone, rhg2, rtrials = O.afinito_steps(opd, og, np.float64(0.999), np.float64(1e-9), idxa, rt, rg, rgam, rfi, rhg, rav, rz)
        ok["sharded_afinito_steps_counts"] = bool(done == rdone == 500 and trials == rtrials)
        ok["sharded_afinito_steps_z"] = bool(np.abs(sd.z.cpu().numpy() - rz).max() <= 1e-10 * max(np.abs(rz).max(), 1e-30))
        ok["sharded_afinito_steps_table_shard"] = bool(np.abs(sd.s.cpu().numpy() - rt[row0:row0 + n]).max() <= 1e-10 * max(np.abs(rt).max(), 1e-30))
        ok["sharded_afinito_steps_hg"] = bool(abs(sd.hat_γ - float(rhg2)) <= 1e-11 * float(rhg2))
        grp4.close()
        # ---- ProShI on row-sharded agents (ProShI_basic.jl:76-87, :109-121): every rank owns a block (or every world-th) of the agents and
        # their table rows -- the table IS the solution, so it stays sharded; a batch updates the members a rank owns and one
        # all-reduce of d + 1 scalars gives every rank the same av / z (hat_gamma = sum_i gamma_i is all-reduced in the init)
        from ciaoalgorithms_jl_amd.device import PackedSepQuad
        Na, da = 203, 256
        rng = np.random.default_rng(77)
        Qa = rng.uniform(0.5, 3.0, (Na, da))
        qa = rng.standard_normal((Na, da))
        eta, lo, hi = 10.0, -2.0, 2.0
        La = Qa.max(axis=1) + eta
        osq, obox = O.SepQuad(Qa, qa, eta, lo, hi), O.Prox("box", lo=-np.inf, hi=1.0)
        gbox = ProxG(L.PROX_BOX, lo=-float("inf"), hi=1.0)
        row0, n = shard_rows(Na, rank, world)
        for name, sl, kw in (("block", slice(row0, row0 + n), dict(row0=row0)), ("cyclic", slice(rank, None, world), {})):
            Fs = PackedSepQuad(torch.from_numpy(np.ascontiguousarray(Qa[sl])).to(dev), torch.from_numpy(np.ascontiguousarray(qa[sl])).to(dev),
                               eta, lo, hi, N_total=Na, **kw)
            if name == "cyclic":
                Fs.cyclic = (rank, world)
            for sweeping, batch in ((1, 24), (2, 32), (3, 40)):
                xs, it = S.Proshi(np.float64, maxit=9, sweeping=sweeping, minibatch=(True, batch))(np.zeros(da), F=Fs, g=gbox, L=La, N=Na,
                                                                                                    ctx=ctx, stream=IndexStream(20 + sweeping))
                xr, _ = RS.proshi(osq, obox, np.zeros(da), maxit=9, sweeping=sweeping, batch=batch, L=La, stream=IndexStream(20 + sweeping))
                mine = np.stack([np.asarray(v) for v in xs]) if not isinstance(xs, np.ndarray) else xs
                ok[f"proshi_{name}_sw{sweeping}"] = bool(mine.shape == xr[sl].shape and
                                                          np.abs(mine - xr[sl]).max() <= 1e-10 * max(np.abs(xr).max(), 1e-30))
        ctx.synchronize()
        ctx.close()
        q.put((rank, ok))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, {"exception": repr(e) + traceback.format_exc()}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok in res:
        assert all(v is True for v in ok.values()), f"rank {rank}: {ok}"


def _rccl_worker(q):
    """One rank, a communicator of our own, the library calling ncclAllReduce itself (ciao_ctx_set_rccl)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import torch
        import ciao_loader
        ciao_loader.load()
        from ciaoalgorithms_jl_amd import _lib as L
        from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
        from ciaoalgorithms_jl_amd.parallel import RcclComm
        from ciaoalgorithms_jl_amd.solvers import Finito
        from ciaoalgorithms_jl_amd.sampling import IndexStream
        from oracle import oracle as O
        from oracle import ref_solvers as RS
        import problems as P
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ok = {}
        ctx = Context(0)
        comm = RcclComm(0, 1, 0)
        ctx.set_rccl(comm)
        N, d = 300, 1024
        A, b, x = P.synthetic("ls", N, d, np.float64, seed=3)
        F = PackedF(L.LOSS_LS, torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev), float(N))
        g = ProxG(L.PROX_L1, lam=0.01)
        og, op = O.Prox("l1", lam=0.01), O.Problem("ls", A, b, float(N))
        xd = torch.from_numpy(x).to(dev)
        av, y = torch.empty_like(xd), torch.empty_like(xd)
        ctx.proxgrad_step(F, g, 0.05 / N, xd, av, y)
        rav = O.full_pass(op, x)
        ok["av"] = bool(np.abs(av.cpu().numpy() - rav).max() <= 1e-10 * np.abs(rav).max())
        Li = float(N) * np.sum(A * A, axis=1)
        xs, _ = Finito(np.float64, maxit=6, sweeping=2, minibatch=(True, 64))(np.zeros(d), F=F, g=g, L=Li, N=N, ctx=ctx, stream=IndexStream(0))
        xr, _ = RS.finito(op, og, np.zeros(d), maxit=6, sweeping=2, batch=64, lfinito=False, L=Li, stream=IndexStream(0))
        ok["finito"] = bool(np.abs(xs - xr).max() <= 1e-9 * max(np.abs(xr).max(), 1e-30) + 1e-13)
        try:   # with a communicator set the problem counts as row-sharded: the sequential chains refuse it
            ctx.svrg_inner(F, g, 0.1, np.zeros(3, np.int64), av, y, xd, torch.empty_like(xd))
            ok["chain_refused"] = False
        except L.CiaoError:
            ok["chain_refused"] = True
        ctx.set_rccl(None)
        ctx.svrg_inner(F, g, 0.1 / N, np.zeros(3, np.int64), av, y, xd, torch.empty_like(xd))   # accepted again
        ctx.synchronize()
        ctx.close()
        comm.close()
        q.put(ok)
    except Exception as e:  # pragma: no cover
        import traceback
        q.put({"exception": repr(e) + traceback.format_exc()})


def test_native_rccl_allreduce_one_rank():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_rccl_worker, args=(q,))
    p.start()
    ok = q.get(timeout=300)
    p.join(timeout=60)
    assert all(v is True for v in ok.values()), ok


def _peer_worker(rank, world, port, q):
    """The one-shot peer all-reduce (csrc/peer_kernels.h) between two processes sharing the test box's GPU: real HIP IPC mappings
    of each other's mailboxes, the kernels of both processes running side by side.  Every result is held BITWISE against the
    host-staged hook path (the sum of two terms is the same in either order) and against the oracle."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
    from ciaoalgorithms_jl_amd.parallel import AllReduceHook, PeerGroup, shard_rows
    from ciaoalgorithms_jl_amd.solvers import Finito
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    from oracle import oracle as O
    import problems as P
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = {}
    try:
        dev = torch.device("cuda", 0)
        ctx = Context(0)
        # (the ranks do host work of different length between reductions -- the oracle runs beside them -- and on a box whose CPU share is
        # contended one rank was once 30 s behind: the wall-clock bound of a peer wait is a setting, 30 s by default)
        ctx.set_option("peer_timeout_s", 150)
        N, d = 403, 1024
        A, b, x = P.synthetic("ls", N, d, np.float64, seed=12)
        row0, n = shard_rows(N, rank, world)
        F = PackedF(L.LOSS_LS, torch.from_numpy(A[row0:row0 + n]).to(dev), torch.from_numpy(b[row0:row0 + n]).to(dev), float(N),
                    N_total=N, row0=row0)
        g = ProxG(L.PROX_L1, lam=0.01)
        og, op = O.Prox("l1", lam=0.01), O.Problem("ls", A, b, float(N))
        xd = torch.from_numpy(x).to(dev)
        Li = float(N) * np.sum(A * A, axis=1)

        def workload():
            out = {}
            av, y = torch.empty_like(xd), torch.empty_like(xd)
            xk = xd.clone()
            for _ in range(5):                                    # five chained sweeps: both parities, increasing sequence numbers
                ctx.proxgrad_step(F, g, 0.05 / N, xk, av, y)
                xk, y = y, xk
            out["sweeps"] = xk.cpu().numpy().copy()
            out["objective"] = ctx.objective(F, g, xk)            # the raw-sum form of the reduction
            for lf in (False, True):
                solver = Finito(np.float64, maxit=8, sweeping=2, minibatch=(True, 64), LFinito=lf)
                out[f"finito{int(lf)}"] = solver(np.zeros(d), F=F, g=g, L=Li, N=N, ctx=ctx, stream=IndexStream(0))[0]
            ctx.synchronize()
            return out

        hook = AllReduceHook(dev)
        ctx.set_allreduce(hook)
        ref = workload()
        ctx.set_allreduce(None)
        pg = PeerGroup(ctx, max_elems=2 * d)
        ctx.set_peers(pg)
        # (1) the primitive on a buffer of its own, thirty times in a row
        good = True
        for k in range(30):
            t = torch.full((d + 1,), float(rank + 1) * (k + 1), dtype=torch.float64, device=dev)
            t[7] = 0.5 * (rank + 1)
            ctx.peer_allreduce(t)
            ctx.synchronize()
            want = sum(float(r + 1) * (k + 1) for r in range(world))
            good = good and float(t[0]) == want and float(t[d]) == want and float(t[7]) == 0.5 * sum(r + 1 for r in range(world))
        ok["primitive"] = bool(good)
        t32 = torch.full((100,), 1.5 + rank, dtype=torch.float32, device=dev)
        ctx.peer_allreduce(t32)
        ctx.synchronize()
        ok["primitive_f32"] = bool(torch.all(t32 == sum(1.5 + r for r in range(world))))
        # (2) the fused form inside sweeps / batches / the objective: bitwise the hook path's results
        got = workload()
        for key in ref:
            ok[f"bitwise_{key}"] = bool(np.array_equal(np.asarray(ref[key]), np.asarray(got[key])))
        rav = O.full_pass(op, x)
        av1, y1 = torch.empty_like(xd), torch.empty_like(xd)
        ctx.proxgrad_step(F, g, 0.05 / N, xd, av1, y1)
        ok["av_vs_oracle"] = bool(np.abs(av1.cpu().numpy() - rav).max() <= 1e-10 * np.abs(rav).max())
        # (3) replicas stay bitwise identical across ranks
        gathered = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(got["sweeps"]))
        ok["bitwise_replicas"] = all(torch.equal(gathered[0], t) for t in gathered)
        # (4) a reduction longer than the mailboxes is refused, not truncated
        try:
            big = torch.zeros(2 * d + 5, dtype=torch.float64, device=dev)
            ctx.peer_allreduce(big)
            ok["oversize_refused"] = False
        except L.CiaoError:
            ok["oversize_refused"] = True
        # (5) ADVICE r3 (medium): the ONE chain of a row-sharded SVRG / SAGA solve with the peers set.  The owner runs the chain
        # over both shards; the other rank enqueues its half of the owner's broadcast at once and waits IN the mailbox for the whole
        # chain kernel (a wall-clock bound now, lengthened for this one reduction by the chain's length; it used to be a poll
        # count).  Against the oracle on the whole problem, and bitwise the owner's state on every rank.
        from ciaoalgorithms_jl_amd.parallel import ShardGroup
        from ciaoalgorithms_jl_amd import solvers as S
        from oracle import ref_solvers as RS
        gamma = 1.0 / (7 * float(Li.max()))
        grp = ShardGroup(ctx, owner=0)
        it = iter(S.iterator(S.SVRG(np.float64, γ=gamma), np.zeros(d), F=F, g=g, N=N, ctx=ctx, stream=IndexStream(3), shards=grp))
        rit = iter(RS.SVRGIterable(op, og, np.zeros(d), gamma=gamma, stream=IndexStream(3)))
        worst = 0.0
        for _ in range(4):
            sd, sr = next(it), next(rit)
            worst = max(worst, float(np.abs(sd.z_full.cpu().numpy() - sr.z_full).max() / max(np.abs(sr.z_full).max(), 1e-30)))
        ctx.synchronize()
        ok["sharded_svrg_over_peers_vs_oracle"] = bool(worst <= 1e-11)
        gathered = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, sd.z_full.cpu())
        ok["sharded_svrg_over_peers_replicas_bitwise"] = all(torch.equal(gathered[0], t) for t in gathered)
        grp.close()
        grp2 = ShardGroup(ctx, owner=1)          # the OTHER rank owns the SAGA chain: rank 0 is the one that waits
        sit = iter(S.iterator(S.SAGA(np.float64, γ=gamma), np.zeros(d), F=F, g=g, N=N, ctx=ctx, stream=IndexStream(5), shards=grp2))
        rsit = iter(RS.SAGAIterable(op, og, np.zeros(d), gamma=gamma, stream=IndexStream(5)))
        ss = sr = None
        for _ in range(300):
            ss, sr = next(sit), next(rsit)
        ctx.synchronize()
        ok["sharded_saga_over_peers_vs_oracle"] = bool(np.abs(ss.z.cpu().numpy() - sr.z).max() <= 1e-11 * max(np.abs(sr.z).max(), 1e-30))
        ok["sharded_saga_over_peers_table_shard"] = bool(np.abs(ss.s.cpu().numpy() - sr.s[row0:row0 + n]).max() <= 1e-11 * np.abs(sr.s).max())
        grp2.close()
        # adaptive Finito's chain through the mailboxes: av and z as the chains' hand-over, hat_gamma and the two counters as a second,
        # 3-element Float64 reduction (rank 1 owns the chain, rank 0 waits)
        grp2b = ShardGroup(ctx, owner=1)
        xs, nit = S.Finito(np.float64, maxit=2 * N, sweeping=1, adaptive=True)(np.zeros(d), F=F, g=g, L=Li, N=N, ctx=ctx,
                                                                                 stream=IndexStream(41), shards=grp2b)
        ctx.synchronize()
        xr, rit_n = RS.finito(op, og, np.zeros(d), maxit=2 * N, sweeping=1, adaptive=True, L=Li, stream=IndexStream(41))
        ok["sharded_afinito_over_peers_vs_oracle"] = bool(nit == rit_n and np.abs(xs - xr).max() <= 1e-9 * max(np.abs(xr).max(), 1e-30))
        gathered = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(np.ascontiguousarray(xs)))
        ok["sharded_afinito_over_peers_replicas_bitwise"] = all(torch.equal(gathered[0], t) for t in gathered)
        grp2b.close()
        # a long chain (1.5M steps: about half a second of kernel) that the waiting rank has to sit out
        idxl = IndexStream(11).rand_indices(N, 1_500_000)
        grp3 = ShardGroup(ctx, owner=0)
        avs, zs, zfs, ws = (torch.empty(d, dtype=torch.float64, device=dev) for _ in range(4))
        ctx.svrg_init(F, xd, avs, zs, zfs, ws)
        grp3.install(F)
        ctx.svrg_inner(F, g, gamma, idxl, avs, zs, zfs, ws)
        ctx.synchronize()                                            # raises if a rank gave up waiting (error word 4)
        wall = [torch.zeros(d, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(wall, ws.cpu())
        ok["long_sharded_chain_over_peers_waited_out"] = all(torch.equal(wall[0], t) for t in wall) and bool(torch.isfinite(ws).all())
        grp3.close()
        # (6) config #5's batch unit through the mailboxes: Finito, d = 4096 fp32, static batches of 1024 = 512
        # rows per rank, per-sample stepsizes, against the oracle on the whole problem -- in Float64 on the same Float32 data (the
        # accuracy statement) and in Float32 (the table rows; z / av within the reference fold's own loss, tests/test_gpu_configs.py C5)
        N5, d5, r5, nb5 = 8192, 4096, 1024, 6
        A5, b5, _ = P.synthetic("ls", N5, d5, np.float32, seed=55)
        # round-robin row ownership (rank owns the global rows i with i % world == rank): every contiguous batch of 1024 has 512 members
        # on each rank, as DESIGN section 7 prescribes for config #5's static batches
        n5 = N5 // world
        F5 = PackedF(L.LOSS_LS, torch.from_numpy(np.ascontiguousarray(A5[rank::world])).to(dev),
                     torch.from_numpy(np.ascontiguousarray(b5[rank::world])).to(dev), float(N5), N_total=N5, cyclic=(rank, world))
        g5 = ProxG(L.PROX_L1, lam=1e-3)
        gam5 = (0.999 / 1.3 * (1.0 + 0.1 * ((np.arange(N5) * 0.6180339887498949) % 1.0))).astype(np.float32)
        hg5 = np.float32(1.0 / np.sum(1.0 / gam5.astype(np.float64)))
        gam5d = torch.from_numpy(np.ascontiguousarray(gam5[rank::world])).to(dev)
        x05 = torch.zeros(d5, dtype=torch.float32, device=dev)
        tab5 = torch.empty((n5, d5), dtype=torch.float32, device=dev)
        av5, z5 = torch.empty_like(x05), torch.empty_like(x05)
        pg5 = PeerGroup(ctx, max_elems=d5 + 1)
        ctx.set_peers(pg5)
        ctx.finito_init(F5, g5, gam5d, float(hg5), x05, tab5, av5, z5)
        ctx.synchronize()
        st_tab = [torch.zeros((n5, d5), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(st_tab, tab5.cpu())
        h_tab = np.empty((N5, d5), np.float32)
        for r in range(world):
            h_tab[r::world] = st_tab[r].numpy()
        h_av, h_z = av5.cpu().numpy().copy(), z5.cpu().numpy().copy()
        w_tab, w_av, w_z = h_tab.astype(np.float64), h_av.astype(np.float64), h_z.astype(np.float64)
        og5 = O.Prox("l1", lam=1e-3)
        first5 = (np.arange(1, nb5 + 1, dtype=np.int64) % (N5 // r5)) * r5       # cyclic order starts at batch 2 (Finito_basic.jl:99)
        batches5 = [np.arange(f, f + r5) for f in first5]
        O.finito_steps(O.Problem("ls", A5, b5, float(N5)), og5, gam5, hg5, batches5, h_tab, h_av, h_z)
        O.finito_steps(O.Problem("ls", A5.astype(np.float64), b5.astype(np.float64), float(N5)), og5, gam5.astype(np.float64), float(hg5),
                       batches5, w_tab, w_av, w_z)
        lo5, ln5 = F5.local_blocks(first5, first5 + r5)
        ctx.finito_steps_blocks(F5, g5, gam5d, float(hg5), lo5, ln5, tab5, av5, z5)
        kern5 = ctx.last_kernel()
        ctx.synchronize()
        e32 = np.finfo(np.float32).eps
        def eps_of(dev_t, ref):
            return float(np.abs(dev_t.cpu().numpy().astype(np.float64) - ref).max() / (e32 * max(np.abs(ref).max(), 1e-300)))
        ok["c5_batch_kernel"] = "rows_split_kernel<f32,J4,mode4>" in kern5 and bool((ln5 == r5 // world).all())
        ok["c5_batches_over_peers_vs_f64_oracle"] = eps_of(z5, w_z) <= 16 and eps_of(av5, w_av) <= 16 and eps_of(tab5, w_tab[rank::world]) <= 8
        ok["c5_batches_over_peers_vs_f32_oracle"] = eps_of(tab5, h_tab[rank::world].astype(np.float64)) <= 20 and \
            eps_of(z5, h_z.astype(np.float64)) <= 0.5 * nb5 * r5
        zs5 = [torch.zeros(d5, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(zs5, z5.cpu())
        ok["c5_replicas_bitwise"] = all(torch.equal(zs5[0], t) for t in zs5)
        # (7) ADVICE r3 (low): the same group turned off and on again continues its sequence numbers (flags of the earlier
        # reductions are still in the mailboxes and would match a restarted count), and what the peers displaced is back after "off"
        # ... and ONE RANK IS LATE: rank 0 has set the group again and published its first reduction -- its flag, one number
        # further, is in rank 1's mailbox -- before rank 1 reads where to continue.  Each rank continues from its OWN flag, which no
        # peer writes (until round 5 it took the largest flag in its mailbox: the late rank then started one number ahead and both
        # timed out)
        ctx.set_peers(None)
        if rank == 1:
            import time
            time.sleep(1.5)
        ctx.set_peers(pg5)
        t7 = torch.full((d5 + 1,), float(rank + 1), dtype=torch.float32, device=dev)
        for _ in range(3):
            t7.fill_(float(rank + 1))
            ctx.peer_allreduce(t7)
        ctx.synchronize()
        ok["peers_reused_after_off_on"] = bool(torch.all(t7 == sum(float(r + 1) for r in range(world))))
        ctx.set_peers(None)
        pg5.close()
        ctx.set_allreduce(hook)
        ctx.set_peers(pg)
        ctx.set_peers(None)                                          # "off" puts the displaced hook back
        calls0 = hook.calls
        ctx.proxgrad_step(F, g, 0.05 / N, xd, av1, y1)
        ctx.synchronize()
        ok["displaced_hook_restored"] = hook.calls == calls0 + 1 and bool(np.abs(av1.cpu().numpy() - rav).max() <= 1e-10 * np.abs(rav).max())
        ctx.set_allreduce(None)
        # (8) off again: the context is a single-device context as before
        pg.close()
        solo = PackedF(L.LOSS_LS, torch.from_numpy(A).to(dev), torch.from_numpy(b).to(dev), float(N))
        ctx.proxgrad_step(solo, g, 0.05 / N, xd, av1, y1)
        ctx.synchronize()
        ok["single_again"] = bool(np.abs(av1.cpu().numpy() - rav).max() <= 1e-10 * np.abs(rav).max())
        ctx.close()
        q.put((rank, ok))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, {"exception": repr(e) + traceback.format_exc()}))
    finally:
        dist.destroy_process_group()


def test_peer_allreduce_two_ranks_on_one_gpu():
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok in res:
        assert all(v is True for v in ok.values()), f"rank {rank}: {ok}"
