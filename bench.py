#!/usr/bin/env python3
"""bench.py -- the SVRG full-gradient + prox sweep (the roofline row of BASELINE.md).

One "step" = one pass of the hot path over the resident batch of synthetic input:
        av = (1/N) sum_i grad f_i(x)  over ALL N rows of the row-major N x d matrix (each row read from HBM once)
        x+ = prox_{gamma g}(x - gamma * av)                      (soft threshold, fused into the reduce epilogue)
and the next step starts from x+ (a proximal-gradient iteration, so no step can be cached).

Workload at N=1: Lasso (f_i = LeastSquares(a_i', b_i, N), g = NormL1) with N = 10M rows, d = 1024, fp64 (81.92 GB of A):
the configuration BASELINE.json's metric is quoted on.  With --gpus P every rank holds its own 10M-row shard (weak
scaling; P = 8 is BASELINE config #4, N = 80M) and the d-vector sum is all-reduced over xGMI once per step.

Launching.  `python bench.py --gpus P` with P > 1 and no WORLD_SIZE in the environment starts the P ranks ITSELF: the
parent process (which never touches the GPU and does not even import torch) spawns P fresh children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits for them, relays rank 0's JSON line and exits
non-zero if any child failed.  Under `python -m torch.distributed.run --nproc-per-node P bench.py --gpus P` (WORLD_SIZE
set) each process is one rank.  Either way: one process per GPU.

The collective is RCCL over xGMI.  Default: the library calls ncclAllReduce itself on its stream (ciao_ctx_set_rccl; no
host callback per reduction).  CIAO_BENCH_COLLECTIVE=torch routes it through torch.distributed (backend "nccl" = RCCL)
instead; CIAO_BENCH_BACKEND=gloo exists only to rehearse several ranks on ONE GPU (host-staged).  `config.collective`
reports what actually ran.

Prints ONE JSON line (rank 0).  `value` = sample-gradient(+prox) evaluations of the SWEEP per second over all ranks,
inputs resident in HBM before the timed region.  An *update* in SURVEY.md section 8d's sense is one step of the
sequential SVRG / SAGA inner loops; those are latency-bound dependent chains and are reported separately at top level
(`svrg_updates_per_sec`, `saga_updates_per_sec`, `svrg_epochs_per_sec_N10M`), each with its own roofline fraction.
`roofline` is for the dominant kernel (the rows sweep), timed with HIP events on the stream it is launched on;
`cpu_baseline` is the single-threaded CPU oracle on a bounded sample of the same rows.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--loss", choices=["ls", "logistic"], default="ls")
    ap.add_argument("--prefetch", type=int, default=None, help="sweep_prefetch option (tuning)")
    ap.add_argument("--blocks-per-cu", type=int, default=None, help="sweep_blocks_per_cu option (tuning)")
    ap.add_argument("--grid", type=int, default=None, help="sweep_grid option (tuning)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary timings (sequential chains, table kernels; N=1 only)")
    ap.add_argument("--cpu-chain-updates", type=int, default=100_000, help="updates of each chain the CPU baseline replays")
    ap.add_argument("--cpu-chain-seconds", type=float, default=5.0, help="CPU time budget of each chain baseline")
    ap.add_argument("--no-chains", action="store_true", help="skip the top-level SVRG / SAGA chain figures (N=1 only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work at all: rendezvous, barrier and the JSON relay only (CPU test of the launcher)")
    return ap.parse_args(argv)


# ======================================================================================================================
# parent: spawn one fresh process per GPU (never touches the GPU itself)
# ======================================================================================================================
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    import subprocess
    world = args.gpus
    env0 = dict(os.environ)
    env0["WORLD_SIZE"] = str(world)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(_free_port()))
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env0["CIAO_BENCH_SPAWNED"] = "1"
    procs = []
    for r in range(world):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    out0 = ""
    deadline = time.time() + float(os.environ.get("CIAO_BENCH_TIMEOUT", "1500"))
    failed = None
    try:
        # rank 0's line is short: read it when the process ends; poll the others so that one dead rank cannot hang the rest
        pending = set(range(world))
        while pending and failed is None:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is not None:
                    pending.discard(r)
                    if r == 0:
                        out0 = procs[0].stdout.read()
                    if rc != 0:
                        failed = (r, rc)
                        break
            if time.time() > deadline:
                failed = (-1, 124)
            time.sleep(0.05)
    finally:
        for p in procs:           # exactly the processes started here, by PID
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                pass
    if failed is not None:
        print(f"[bench] rank {failed[0]} failed with exit code {failed[1]}; no result line", file=sys.stderr)
        sys.stdout.write(out0)
        return failed[1] if failed[1] not in (0, None) else 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not lines:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


# ======================================================================================================================
# one rank
# ======================================================================================================================
def dry_rank(args):
    """The launcher's plumbing without a GPU: process group (gloo), barrier, MAX over ranks, rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "sweep_sample_gradients_per_sec", "value": None, "unit": "sample-gradients/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_rank_seconds": float(el.item()),
                          "config": {"workload": "launcher_dry_run", "collective": "gloo(dry run)" if world > 1 else "none",
                                     "launcher": "bench.py spawned the ranks" if os.environ.get("CIAO_BENCH_SPAWNED") else
                                     ("external launcher (WORLD_SIZE set)" if world > 1 else "single process")}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def _timed_events(torch, stream, fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def run_rank(args):
    import numpy as np
    import torch
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
    from ciaoalgorithms_jl_amd.parallel import AllReduceHook, RcclComm, init_process_group_from_env, shard_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    dist = None
    force_dist = os.environ.get("CIAO_BENCH_FORCE_DIST") == "1"   # exercise the collective path even with one rank
    backend = os.environ.get("CIAO_BENCH_BACKEND", "nccl")        # "nccl" IS RCCL on ROCm; gloo = one-GPU rehearsal only
    if world > 1 or force_dist:
        import torch.distributed as dist
        # one process may see only ITS GPU (a launcher that masks devices per rank): then device_count is 1 and that is fine
        masked = any(os.environ.get(k) for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        if backend == "nccl" and world > torch.cuda.device_count() and not masked:
            print(f"[bench] {world} ranks need {world} GPUs (found {torch.cuda.device_count()}); RCCL takes one rank per device. "
                  f"CIAO_BENCH_BACKEND=gloo rehearses several ranks on one GPU.", file=sys.stderr)
            return 2
        rank, world, local = init_process_group_from_env(backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    es = 8 if args.dtype == "f64" else 4

    ctx = Context(dev.index)
    if args.prefetch is not None:
        ctx.set_option("sweep_prefetch", args.prefetch)
    if args.blocks_per_cu is not None:
        ctx.set_option("sweep_blocks_per_cu", args.blocks_per_cu)
        ctx.set_option("split_blocks_per_cu", min(args.blocks_per_cu, 16))
    if args.grid is not None:
        ctx.set_option("sweep_grid", args.grid)
    for kv in os.environ.get("CIAO_OPTS", "").split(","):   # any other tuning knob: CIAO_OPTS=key=value,key=value
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))

    # ---- synthetic problem, generated on the device, keyed by the GLOBAL (row, col): SURVEY.md section 8d ------------
    n_local, d = args.rows_per_gpu, args.d
    N_total = n_local * world
    row0 = rank * n_local
    assert shard_rows(N_total, rank, world) == (row0, n_local)
    A = torch.empty((n_local, d), dtype=tdt, device=dev)
    b = torch.empty((n_local,), dtype=tdt, device=dev)
    ctx.synth_normal(A, row0, seed=0, scale=1.0 / np.sqrt(d))            # ||a_i||^2 ~ 1
    rng = np.random.default_rng(0)
    x_true = rng.standard_normal(d) * (rng.random(d) < 0.05)
    x_true_d = torch.from_numpy(x_true).to(dev, tdt)
    logistic = args.loss == "logistic"
    lam_f = 1.0 if logistic else float(N_total)                          # LeastSquares(.., R(N)) test_lasso.jl:54
    F = PackedF(L.LOSS_LOGISTIC if logistic else L.LOSS_LS, A, b, lam_f, N_total=N_total, row0=row0)
    ctx.synth_targets(F, x_true_d, noise=0.01 if not logistic else 0.1, labels=logistic, seed=0, b_out=b)
    lam_g = 1e-3 if not logistic else 1.0 / N_total
    g = ProxG(L.PROX_L1, lam=lam_g)
    L_max = (lam_f if not logistic else 0.25) * 1.3                      # ||a_i||^2 <= ~1.3 for d = 1024
    gamma = 1.0 / (7.0 * L_max) if not logistic else 1.0 / (10.0 * L_max)   # test_lasso.jl:164 / test_logistic_l1.jl:126

    # ---- the collective: what is installed is what `config.collective` reports -----------------------------------------
    collective, rccl_ranks, comm, hook, peers = "none", None, None, None, None
    if world > 1 or force_dist:
        want = os.environ.get("CIAO_BENCH_COLLECTIVE", "rccl" if backend == "nccl" else "torch")
        if want == "peer":
            # the one-shot peer all-reduce (csrc/peer_kernels.h): the sweep's finalize kernel writes the raw sum into every rank's
            # mailbox, its epilogue waits for the flags -- no collective call, no extra launch.  Not the default until a node run
            # has measured it against ncclAllReduce.
            from ciaoalgorithms_jl_amd.parallel import PeerGroup
            peers = PeerGroup(ctx, max_elems=2 * d)
            ctx.set_peers(peers)
            collective = "peer mailboxes over HIP IPC (one direct write per rank + flags), fused into finalize / epilogue: no collective call"
        elif want == "rccl" and backend == "nccl":
            # Vote BEFORE the collective initialisation, on what can fail on one rank alone (loading librccl, its symbols): a rank
            # that failed there while the others went on into ncclCommInitRank would leave them waiting for it.  Every rank then
            # takes the same path; a failure INSIDE ncclCommInitRank is fatal (this rank exits non-zero and the launcher -- ours
            # or torch.distributed.run -- ends the others).
            why = RcclComm.probe()
            if why is not None:
                print(f"[bench] rank {rank}: native RCCL path unavailable ({why})", file=sys.stderr)
            ok = torch.tensor([1 if why is None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if rank == 0:
                    print("[bench] falling back to torch.distributed for the collective on every rank", file=sys.stderr)
            else:
                # ... and a second vote AFTER the collective initialisation: a rank whose ncclCommInitRank returned an error (the
                # others' then did too, or will time out inside it) must not leave the rest on a communicator it is not part of
                err = None
                try:
                    comm = RcclComm(rank, world, dev.index)
                    rccl_ranks = comm.count()
                except Exception as e:   # noqa: BLE001 -- whatever went wrong, every rank has to hear of it
                    err, comm = repr(e), None
                    print(f"[bench] rank {rank}: native RCCL communicator failed ({err})", file=sys.stderr)
                ok = torch.tensor([1 if err is None else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if comm is not None:
                        comm.close()
                    comm, rccl_ranks = None, None
                    if rank == 0:
                        print("[bench] falling back to torch.distributed for the collective on every rank", file=sys.stderr)
                else:
                    ctx.set_rccl(comm)
                    collective = "rccl ncclAllReduce(d+1) per step, issued by the library on its stream"
        if comm is None and peers is None:
            hook = AllReduceHook(dev)
            ctx.set_allreduce(hook)
            collective = (f"torch.distributed all_reduce(d+1) per step, backend {dist.get_backend()}"
                          + (" (= RCCL)" if dist.get_backend() == "nccl" else " (host-staged: one-GPU rehearsal, not xGMI)"))
    xa = torch.zeros(d, dtype=tdt, device=dev)                           # x0 = 0 (test_lasso.jl:60)
    xb = torch.empty_like(xa)
    av = torch.empty_like(xa)
    ctx.synchronize()

    def step(i):
        src, dst = (xa, xb) if i % 2 == 0 else (xb, xa)
        ctx.proxgrad_step(F, g, gamma, src, av, dst)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    ctx.timing_enable(True)
    ctx.timing_read()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = ctx.timing_read()
    ctx.timing_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctx.synchronize()
    kernel_name = ctx.last_kernel()

    # ---- the collective alone: K all-reduces of d+1 scalars back to back on the compute stream ---------------------------
    allreduce_us, allreduce_peer_us = None, None
    if world > 1 or force_dist:
        buf = torch.zeros(d + 1, dtype=tdt, device=dev)
        reps = 200
        try:
            if peers is not None:
                fn = lambda: ctx.peer_allreduce(buf)
            elif comm is not None:
                fn = lambda: comm.all_reduce(buf, ctx.stream)
            else:
                fn = lambda: hook(buf.data_ptr(), d + 1, L.F64 if es == 8 else L.F32, ctx.stream.cuda_stream if ctx.stream else 0)
            for _ in range(10):
                fn()
            fence()
            allreduce_us = _timed_events(torch, ctx.stream or torch.cuda.current_stream(), fn, reps) * 1e6
            fence()
        except Exception as e:
            print(f"[bench] all-reduce microbenchmark failed: {e!r}", file=sys.stderr)
        # ... and the peer exchange beside it, whatever collective the timed region used (as two kernels of its own here: the
        # sweep fuses them into finalize / epilogue and pays no launch for them).  Set up and torn down around the measurement.
        # On real peers (backend nccl, several GPUs) this is the FIRST time the mailboxes cross xGMI: it runs after the timed region,
        # but a fault in it would still cost the whole line, so there it needs CIAO_BENCH_PEER_PROBE=1 (or CIAO_BENCH_COLLECTIVE=peer);
        # the one-GPU rehearsals (gloo backend, CIAO_BENCH_FORCE_DIST) always run it.
        probe = os.environ.get("CIAO_BENCH_PEER_PROBE", "1" if (backend != "nccl" or world == 1) else "0") == "1"
        if peers is not None:
            allreduce_peer_us = allreduce_us
        elif probe:
            try:
                from ciaoalgorithms_jl_amd.parallel import PeerGroup
                saved_rccl, saved_hook = comm, hook
                pg = PeerGroup(ctx, max_elems=2 * d)
                ctx.set_peers(pg)
                for _ in range(10):
                    ctx.peer_allreduce(buf)
                fence()
                allreduce_peer_us = _timed_events(torch, ctx.stream or torch.cuda.current_stream(), lambda: ctx.peer_allreduce(buf), reps) * 1e6
                fence()
                ctx.set_peers(None)
                pg.close()
                if saved_rccl is not None:
                    ctx.set_rccl(saved_rccl)
                elif saved_hook is not None:
                    ctx.set_allreduce(saved_hook)
            except Exception as e:
                print(f"[bench] peer all-reduce microbenchmark failed: {e!r}", file=sys.stderr)

    units = float(N_total) * args.steps                                  # sample-gradients processed by all ranks
    value = units / elapsed
    alg_bytes = n_local * (d * es + es)                                  # per launch: rows + b_i  
    k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
    achieved = alg_bytes / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
    # HBM traffic per launch from the PMC counters: NOT measured in this run (a counter pass cannot share a run with the
    # timed region); it is the committed value of an earlier `rocprofv3 --pmc` pass of this same command
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            tj = json.load(fh)
        tkey = f"{args.loss}_{args.dtype}_N{n_local}_d{d}"
        traffic = tj.get(tkey)
        counted = tj.get("kernels", {}).get(tkey, {}).get("rocprof_name")
        if traffic is not None and counted and counted.split("<")[0] != kernel_name.split("<")[0]:
            traffic = None   # the dominant kernel has changed since the counter pass: do not present a stale figure
        traffic_src = {"file": "profiles/pmc_traffic.json", "measured_at": tj.get("measured_at"), "kernel_counted": counted,
                       "static": True} if traffic else None
    except Exception:
        traffic = None

    out = {
        "metric": "sweep_sample_gradients_per_sec",
        "metric_definition": "sample-gradient(+fused prox) evaluations per second of the SVRG full-gradient + prox sweep "
                             "(SURVEY.md 8a row S4; BASELINE.md's roofline row).  One SVRG/SAGA inner-loop *update* (SURVEY.md "
                             "8d) is a step of a dependent chain: see svrg_updates_per_sec / saga_updates_per_sec.",
        "value": value,
        "unit": "sample-gradients/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "sweeps_per_sec": args.steps / elapsed,
        # everything in a step that is not the sweep kernel: finalize + epilogue, launch gaps and -- with several ranks -- the
        # all-reduce of the d+1 scalars (BASELINE.md section 2 asks for that figure in microseconds)
        "step_overhead_us_beyond_sweep_kernel": (elapsed / args.steps - k_avg_s) * 1e6,
        "allreduce_us_per_step": allreduce_us,
        "allreduce_us_per_step_peer_mailboxes": allreduce_peer_us,
        "rccl_ranks": rccl_ranks,
        "config": {"workload": f"{'l1_logistic' if logistic else 'lasso'}_svrg_fullgrad_prox_sweep",
                   "N_total": N_total, "rows_per_gpu": n_local, "d": d, "f": "LeastSquares(a_i,b_i,N)" if not logistic else "Precompose(LogisticLoss)",
                   "g": f"NormL1({lam_g:g})", "gamma": gamma, "parallelism": f"rows_sharded_x{world}",
                   "collective": collective, "launcher": "bench.py spawned the ranks" if os.environ.get("CIAO_BENCH_SPAWNED") else
                   ("external launcher (WORLD_SIZE set)" if world > 1 else "single process")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel_name,
                     "kernel_avg_ms": k_avg_s * 1e3, "kernel_launches": k_n, "algorithmic_bytes_per_launch": alg_bytes},
    }

    # ---- cpu_baseline: the single-threaded oracle (reference-shaped sequential pass) on a bounded sample -------------
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            from oracle import oracle as O
            n_s = min(args.cpu_rows, n_local)
            A_h = A[:n_s].cpu().numpy()
            b_h = b[:n_s].cpu().numpy()
            x_h = (xa if args.steps % 2 == 0 else xb).cpu().numpy()   # whichever; any point works
            op = O.Problem("logistic" if logistic else "ls", A_h, b_h, lam_f)
            reps, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds and reps < 50:
                t1 = time.perf_counter()
                O.full_pass(op, x_h)
                t_cpu += time.perf_counter() - t1
                reps += 1
            t2 = time.perf_counter()
            _, nt = O.full_pass_omp(op, x_h)
            t_omp = time.perf_counter() - t2
            out["cpu_baseline"] = {"value": n_s * reps / t_cpu, "unit": "sample-gradients/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_s} rows of the same A (d={d}, {args.dtype}), {reps} sequential full passes, "
                                             f"{t_cpu:.1f} s; oracle/ciao_oracle.c orc_full_pass (SVRG_basic.jl:87-92 restated)",
                                   "all_cores": {"value": n_s / t_omp, "cores": int(nt), "kind": "openmp sweep (not the reference's shape)"},
                                   "host_cores": os.cpu_count()}
            del A_h, b_h, op
        except Exception as e:  # the baseline must never cost us the bench line
            out["cpu_baseline"] = {"value": None, "unit": "sample-gradients/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}

    # ---- the updates SURVEY.md 8d defines: sequential SVRG / SAGA chains at the metric's own size (N=1 only) -------------
    args.cpu_sweep_rate = (out.get("cpu_baseline") or {}).get("value")
    if rank == 0 and world == 1 and not args.no_chains:
        try:
            out.update(chain_figures(ctx, dev, F, g, gamma, A, b, n_local, d, args, L, np, torch))
        except Exception as e:
            out["chain_figures_error"] = repr(e)

    if not args.no_extras and world == 1 and rank == 0:
        try:
            import bench_extras
            del A, b, F
            torch.cuda.empty_cache()
            out["extra"] = bench_extras.run(ctx, dev)
        except Exception as e:
            out["extra"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if peers is not None:
        ctx.set_peers(None)
        peers.close()
    ctx.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def chain_figures(ctx, dev, F, g, gamma, A, b, N, d, args, L, np, torch):
    """One REAL SVRG outer iteration (SVRG_basic.jl:71-96: m = N dependent updates + the full pass) on the resident
    problem, and SAGA steps at BASELINE config #3's size (N rows x d fp32 + the N x d gradient table).  Each figure carries
    the bandwidth bound it is measured against (S3 d*s+8, G3 3*d*s+8 bytes per update); the chains are
    dependent (step k+1 reads the iterate step k wrote), so the bound is not reachable by construction."""
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    from ciaoalgorithms_jl_amd.sampling import IndexStream
    res = {}
    tdt = A.dtype
    es = 8 if tdt == torch.float64 else 4
    st = IndexStream(0)
    x0 = torch.zeros(d, dtype=tdt, device=dev)
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    ctx.svrg_init(F, x0, av, z, zf, w)
    hidx = st.rand_indices(N, N)
    idx = ctx._idx(hidx)
    cpu_svrg = None
    if not args.no_cpu:
        # cpu_baseline of the chain: the single-threaded oracle (orc_svrg_inner: SVRG_basic.jl:73-82 restated,
        # two gradient! calls + the four broadcasts + prox! per update) on the FIRST updates of this very epoch -- the rows they visit
        # are gathered to the host, the state is the device's init state -- for a bounded number of updates
        try:
            cpu_svrg = cpu_chain_baseline("svrg", ctx, F, g, gamma, hidx[:args.cpu_chain_updates], N, (av, z, zf, w), None, args, np, torch)
        except Exception as e:
            cpu_svrg = {"value": None, "unit": "updates/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
    del hidx
    ctx.svrg_iterate(F, g, gamma, idx[:4096], False, av, z, zf, w)                 # warm (code objects, workspace)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.svrg_iterate(F, g, gamma, idx, False, av, z, zf, w, reuse_rowdots=True)      # ONE epoch, m = N
    ctx.synchronize()
    t_epoch = time.perf_counter() - t0
    t1 = time.perf_counter()
    ctx.full_gradient(F, zf, av)
    ctx.synchronize()
    t_sweep = time.perf_counter() - t1
    t_chain = max(t_epoch - t_sweep, 1e-9)
    upd = N / t_chain
    bound = HBM_PEAK_GBS * 1e9 / (d * es + 8)
    res["svrg_updates_per_sec"] = {"value": upd, "us_per_update": 1e6 / upd, "m": N, "N": N, "d": d, "dtype": args.dtype,
                                   "what": "SVRG inner cycle (SVRG_basic.jl:73-82), one dependent chain on one workgroup",
                                   "roofline": {"bound": "hbm", "bytes_per_update": d * es + 8, "bound_updates_per_sec": bound,
                                                "achieved": upd * (d * es + 8) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": upd / bound, "note": "latency-bound dependent chain (SURVEY.md section 7)"}}
    if cpu_svrg is not None:
        res["svrg_updates_per_sec"]["cpu_baseline"] = cpu_svrg
    ep_bytes = N * (d * es + 8) + N * (d * es + es)
    res[f"svrg_epochs_per_sec_N{N // 1_000_000}M" if N % 1_000_000 == 0 else f"svrg_epochs_per_sec_N{N}"] = {
        "value": 1.0 / t_epoch, "seconds_per_epoch": t_epoch, "m": N, "inner_cycle_s": t_chain, "full_pass_s": t_sweep,
        "what": "one SVRG outer iteration = m = N updates + tail + full-gradient sweep (SVRG_basic.jl:71-96), measured once",
        "roofline": {"bound": "hbm", "bytes_per_epoch": ep_bytes, "achieved": ep_bytes / t_epoch / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ep_bytes / t_epoch / 1e9 / HBM_PEAK_GBS}}
    if cpu_svrg is not None and cpu_svrg.get("value") and args.cpu_sweep_rate:
        # an epoch on one host core = N updates + N sample-gradients of the full pass, at the two rates measured in this run
        t_cpu_epoch = N / cpu_svrg["value"] + N / args.cpu_sweep_rate
        res[f"svrg_epochs_per_sec_N{N // 1_000_000}M" if N % 1_000_000 == 0 else f"svrg_epochs_per_sec_N{N}"]["cpu_baseline"] = {
            "value": 1.0 / t_cpu_epoch, "unit": "epochs/s", "cores": 1, "kind": "port",
            "sample": f"extrapolated: N / (oracle updates/s on {cpu_svrg.get('updates')} updates) + N / (oracle sample-gradients/s of "
                      f"cpu_baseline above) = {t_cpu_epoch:.0f} s per epoch; a whole epoch on one core would take that long"}
    del idx
    # ---- the same inner cycle for 256 INDEPENDENT solves at once (a regularisation path: a lambda, an index stream and a state
    # each) over the same resident rows, recorded and launched as one chain batch -- one workgroup per solve.  Not the figure
    # above (that is one solve, sequential by definition): what the GPU's other 255 compute units are worth to a host that has
    # several solves.
    try:
        K, mk = 256, 40_000
        W = torch.zeros((K, d), dtype=tdt, device=dev)
        Z = torch.zeros((K, d), dtype=tdt, device=dev)
        gs_k = [ProxG(L.PROX_L1, lam=g.lam * (1.0 + k / K)) for k in range(K)]
        idxs = [ctx._idx(IndexStream(1000 + k).rand_indices(N, mk)) for k in range(K)]

        def batch(m_):
            with ctx.chain_batch():
                for k in range(K):                       # av and z_full shared (read-only), w and z per solve
                    ctx.svrg_inner(F, gs_k[k], gamma, idxs[k][:m_], av, Z[k], zf, W[k])
            ctx.synchronize()

        batch(512)
        t0 = time.perf_counter()
        batch(mk)
        tb = time.perf_counter() - t0
        updb = K * mk / tb
        res["svrg_updates_per_sec_256_solves"] = {
            "value": updb, "us_per_update_per_solve": tb / mk * 1e6, "solves": K, "m_per_solve": mk, "N": N, "d": d, "dtype": args.dtype,
            "what": "256 independent SVRG inner cycles (SVRG_basic.jl:73-82 each) over the same rows in ONE launch (ciao_ctx_chain_batch_begin "
                    "/ _end): aggregate updates/s; each solve bitwise what its own call computes (tests/test_gpu_chain_batch.py)",
            "kernel": ctx.last_kernel(),
            "roofline": {"bound": "hbm", "bytes_per_update": d * es + 8, "achieved": updb * (d * es + 8) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": updb * (d * es + 8) / 1e9 / HBM_PEAK_GBS}}
        del W, Z, idxs
    except Exception as e:   # noqa: BLE001 -- a secondary figure must not cost the line
        res["svrg_updates_per_sec_256_solves"] = {"error": repr(e)}
    # ---- SAGA at config #3: l1-logistic, fp32, N x d data + N x d table -------------------------------------------------
    if d * 4 * N * 2 + (A.numel() * A.element_size() if A.dtype != torch.float32 else 0) < 250e9:
        A32 = torch.empty((N, d), dtype=torch.float32, device=dev)
        y32 = torch.empty((N,), dtype=torch.float32, device=dev)
        ctx.synth_normal(A32, 0, seed=1, scale=1.0 / np.sqrt(d))
        rng = np.random.default_rng(1)
        xt = torch.from_numpy(rng.standard_normal(d) * (rng.random(d) < 0.05)).to(dev, torch.float32)
        Fs = PackedF(L.LOSS_LOGISTIC, A32, y32, 1.0)
        ctx.synth_targets(Fs, xt, noise=0.1, labels=True, seed=1, b_out=y32)
        gs = ProxG(L.PROX_L1, lam=1.0 / N)
        gam = 1.0 / (3 * 0.25 * 1.3)
        table = torch.empty((N, d), dtype=torch.float32, device=dev)
        x1 = torch.ones(d, dtype=torch.float32, device=dev)
        sav, sz = torch.empty_like(x1), torch.empty_like(x1)
        ctx.saga_init(Fs, gs, gam, x1, table, sav, sz)
        k = 400_000
        hsidx = st.rand_indices(N, k)
        sidx = ctx._idx(hsidx)
        cpu_saga = None
        if not args.no_cpu:
            try:
                cpu_saga = cpu_chain_baseline("saga", ctx, Fs, gs, gam, hsidx[:args.cpu_chain_updates], N, (sav, sz), table, args, np, torch)
            except Exception as e:
                cpu_saga = {"value": None, "unit": "updates/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        ctx.saga_steps(Fs, gs, gam, False, sidx[:4096], table, sav, sz)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.saga_steps(Fs, gs, gam, False, sidx, table, sav, sz)
        ctx.synchronize()
        ts = time.perf_counter() - t0
        upd = k / ts
        bound = HBM_PEAK_GBS * 1e9 / (3 * d * 4 + 8)
        res["saga_updates_per_sec"] = {"value": upd, "us_per_update": 1e6 / upd, "steps": k, "N": N, "d": d, "dtype": "f32",
                                       "what": "SAGA step (SAGA_basic.jl:53-68) at BASELINE config #3 (l1-logistic, table in HBM)",
                                       "kernel": ctx.last_kernel(),
                                       "roofline": {"bound": "hbm", "bytes_per_update": 3 * d * 4 + 8, "bound_updates_per_sec": bound,
                                                    "achieved": upd * (3 * d * 4 + 8) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": upd / bound, "note": "latency-bound dependent chain (SURVEY.md section 7)"}}
        if cpu_saga is not None:
            res["saga_updates_per_sec"]["cpu_baseline"] = cpu_saga
        del A32, y32, table, Fs, sidx
        torch.cuda.empty_cache()
    return res


def cpu_chain_baseline(alg, ctx, F, g, gamma, hidx, N, state, table, args, np, torch):
    """The oracle's sequential chain (single thread) on the rows the first len(hidx) updates of the device's run visit: rows (and
    SAGA table rows) gathered to the host, indices remapped, 1/N of the whole problem, the device's own start state.  The
    device state is not touched.  Repeats the same updates until cpu_chain_seconds have passed (the state keeps moving)."""
    from oracle import oracle as O
    touched, remap = np.unique(hidx, return_inverse=True)
    t = torch.from_numpy(touched).to(F.A.device)
    A_t, b_t = F.A[t].cpu().numpy(), F.b[t].cpu().numpy()
    logistic = (alg == "saga")
    op = O.Problem("logistic" if logistic else "ls", A_t, b_t, F.lam, N_total=N)
    og = O.Prox("l1", lam=g.lam)
    rdt = A_t.dtype.type
    host = [v.cpu().numpy().copy() for v in state]
    h_tab = table[t].cpu().numpy() if table is not None else None
    remap = remap.astype(np.int64)
    reps, t_cpu = 0, 0.0
    while t_cpu < args.cpu_chain_seconds and reps < 200:
        t0 = time.perf_counter()
        if alg == "svrg":
            O.svrg_inner(op, og, rdt(gamma), remap, *host)
        else:
            O.saga_steps(op, og, rdt(gamma), False, remap, h_tab, *host)
        t_cpu += time.perf_counter() - t0
        reps += 1
    n = len(remap) * reps
    what = ("orc_svrg_inner (SVRG_basic.jl:73-82 restated: two gradient! + four broadcasts + prox! per update)" if alg == "svrg"
            else "orc_saga_steps (SAGA_basic.jl:53-68 restated: gradient!, two broadcasts, prox!, table row copy per update)")
    return {"value": n / t_cpu, "unit": "updates/s", "us_per_update": t_cpu / n * 1e6, "cores": 1, "kind": "port", "updates": n,
            "sample": f"the first {len(remap)} updates of the same run ({len(touched)} distinct rows of the same A"
                      f"{' and table' if table is not None else ''} gathered to the host, d={A_t.shape[1]}, {A_t.dtype.name}), "
                      f"{reps} time(s) over, {t_cpu:.1f} s; oracle/ciao_oracle.c {what}", "host_cores": os.cpu_count()}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)          # parent: no torch, no HIP, no GPU
    if args.dry_run:
        return dry_rank(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
