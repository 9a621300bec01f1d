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

Prints ONE JSON line (rank 0), kept under 6 KB so that the tail of a driver's log holds all of it.  `value` = per-sample
grad+prox updates of the SWEEP per second over all ranks (one update = one row's gradient into the aggregate + its share of the
fused prox), inputs resident in HBM before the timed region; `epochs_per_sec` = sweeps per second.  The other three kinds of
update SURVEY.md section 8d defines are reported at top level of the same line, measured after the timed region on rank 0, each
as {value, roofline{...frac}, cpu_baseline{...}}:
    svrg_updates_per_sec       one step of the SVRG inner cycle (S3), from ONE real outer iteration m = N at the metric's size
    svrg_epochs_per_sec_N10M   that outer iteration as a whole (m = N updates + tail + full-gradient sweep)
    saga_updates_per_sec       one SAGA step (G3) at BASELINE config #3 (N x 1024 fp32 + the N x 1024 table), and in fp64
    finito_samples_per_sec     one sample of a Finito batch (F3) at BASELINE config #5's per-rank shape (1.25M x 4096 fp32 + table):
                               batches of 4096, and the 512-row share one rank of eight has of such a batch
`roofline` is for the dominant kernel (the rows sweep), timed with HIP events on the stream it is launched on;
`cpu_baseline` is the single-threaded CPU oracle on a bounded sample of the same rows (also with --gpus N > 1: rank 0 times it
after the timed region while the other ranks wait at the final barrier).  The secondary timings (`extra`) go to a side file
(gpurun_out/bench_extras.json when that directory exists, else bench_extras.json beside this script) and to stderr.

Before a rank allocates its 82 GB, `collective_preflight` sends 1 KB through every collective path that is going to be used
(torch.distributed, the library's own ncclAllReduce, the peer mailboxes when asked for), each stage under a watchdog: a path that
fails is recorded and the next one is taken on every rank alike; if none works the rank exits non-zero with the record on stderr.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
METRIC = "grad+prox updates/sec"          # BASELINE.json's metric; the timed region is the full-gradient + prox sweep (its roofline row)
UNIT = "updates/s"


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
    ap.add_argument("--no-chains", action="store_true", help="skip the top-level SVRG / SAGA / Finito update figures")
    ap.add_argument("--chain-rows", type=int, default=None, help="rows of the chain figures' SAGA / Finito problems (default: the configs' sizes)")
    ap.add_argument("--preflight-seconds", type=float, default=120.0, help="watchdog of each collective preflight stage")
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
class Preflight:
    """1 KB through each collective path BEFORE the big allocation, one stage at a time, each under a watchdog.

    A stage that raises is recorded ("failed: ...") and the caller takes the next path; a stage that HANGS cannot be recovered
    from inside the process (a collective has no timeout of its own here), so its watchdog prints the record to stderr and ends
    the process with exit code 3 -- the launcher (ours, or torch.distributed.run) then ends the other ranks.  Nothing here
    re-executes a process that has touched the GPU: a dead rank is a dead run, reported."""

    def __init__(self, rank, seconds):
        self.rank, self.seconds, self.rec = rank, float(seconds), {}

    def stage(self, name, fn):
        done = threading.Event()

        def dog():
            if not done.wait(self.seconds):
                self.rec[name] = f"timeout after {self.seconds:.0f} s"
                print(f"[bench] rank {self.rank}: collective_preflight " + json.dumps(self.rec), file=sys.stderr, flush=True)
                os._exit(3)

        threading.Thread(target=dog, daemon=True).start()
        try:
            note = fn()
            self.rec[name] = "ok" if not note else f"ok ({note})"
        except Exception as e:   # noqa: BLE001 -- whatever went wrong is the record
            self.rec[name] = ("failed: " + repr(e))[:160]
        finally:
            done.set()
        return self.rec[name].startswith("ok")

    def give_up(self, why):
        self.rec["fatal"] = why
        print(f"[bench] rank {self.rank}: collective_preflight " + json.dumps(self.rec), file=sys.stderr, flush=True)
        return 3


def _vote(dist, torch, ok, dev=None):
    """Every rank learns whether EVERY rank succeeded (MIN over the ranks), through the control plane that stage 1 has shown to work."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dev is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def dry_rank(args):
    """The launcher's plumbing without a GPU: process group (gloo), the collective preflight's control flow, barrier, MAX over
    ranks, rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    pf = Preflight(rank, args.preflight_seconds)
    collective = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

        def torch_stage():
            t = torch.ones(256, dtype=torch.float32)
            dist.all_reduce(t)
            assert float(t[0]) == world, (float(t[0]), world)
            return "gloo, dry run"

        if not pf.stage("torch", torch_stage):
            return pf.give_up("the control plane itself does not work")
        # the native-RCCL stage as the real run takes it: a rank-local failure (here: forced), a vote, the same fallback on every rank
        forced = os.environ.get("CIAO_BENCH_FORCE_RCCL_FAIL")      # "all", or the number of the one rank that fails (tests)
        mine = forced is None or (forced != "all" and str(rank) != forced)

        def rccl_stage():
            if not mine:
                raise RuntimeError("forced by CIAO_BENCH_FORCE_RCCL_FAIL")
            return "not attempted: dry run"

        ok = pf.stage("rccl", rccl_stage)
        if not _vote(dist, torch, ok):
            if ok:
                pf.rec["rccl"] = "ok here, failed on another rank"
            collective = "gloo(dry run); native RCCL failed the preflight: every rank fell back to torch.distributed"
        else:
            collective = "gloo(dry run)"
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": UNIT, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_rank_seconds": float(el.item()),
                          "collective_preflight": pf.rec,
                          "config": {"workload": "launcher_dry_run", "collective": collective,
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

    n_local, d = args.rows_per_gpu, args.d
    N_total = n_local * world
    row0 = rank * n_local
    assert shard_rows(N_total, rank, world) == (row0, n_local)
    logistic = args.loss == "logistic"

    # ---- collective preflight (before the big allocation) and the choice of the collective: what is installed is what
    # `config.collective` reports ------------------------------------------------------------------------------------------
    pf = Preflight(rank, args.preflight_seconds)
    collective, rccl_ranks, comm, hook, peers = "none", None, None, None, None
    if world > 1 or force_dist:
        from ciaoalgorithms_jl_amd.parallel import PeerGroup
        want = os.environ.get("CIAO_BENCH_COLLECTIVE", "rccl" if backend == "nccl" else "torch")
        small = torch.ones(256, dtype=torch.float32, device=dev)

        def torch_stage():
            small.fill_(1.0)
            dist.all_reduce(small)
            torch.cuda.synchronize()
            assert float(small[0]) == world, (float(small[0]), world)
            return f"backend {dist.get_backend()}"

        if not pf.stage("torch", torch_stage):
            return pf.give_up("torch.distributed, the control plane of every other stage, does not work")
        if want == "rccl" and backend == "nccl":
            # what can fail on one rank alone (loading librccl, its symbols; ncclCommInitRank returning an error) is voted on, so
            # that every rank takes the same path; a rank that HANGS inside ncclCommInitRank is ended by the stage's watchdog
            def rccl_stage():
                nonlocal comm, rccl_ranks
                if os.environ.get("CIAO_BENCH_FORCE_RCCL_FAIL") in ("all", str(rank)):
                    raise RuntimeError("forced by CIAO_BENCH_FORCE_RCCL_FAIL")
                why = RcclComm.probe()
                if why is not None:
                    raise RuntimeError(why)
                comm = RcclComm(rank, world, dev.index)
                rccl_ranks = comm.count()
                small.fill_(1.0)
                comm.all_reduce(small, ctx.stream)
                ctx.synchronize()
                assert float(small[0]) == world, (float(small[0]), world)
                return f"{rccl_ranks} ranks"

            ok = pf.stage("rccl", rccl_stage)
            if _vote(dist, torch, ok, dev):
                ctx.set_rccl(comm)
                collective = "rccl ncclAllReduce(d+1) per step, issued by the library on its stream"
            else:
                if ok:
                    pf.rec["rccl"] = "ok here, failed on another rank"
                if comm is not None:
                    comm.close()
                comm, rccl_ranks = None, None
                if rank == 0:
                    print("[bench] native RCCL failed the preflight: torch.distributed carries the collective on every rank", file=sys.stderr)
        elif want == "peer":
            # the one-shot peer all-reduce (csrc/peer_kernels.h): the sweep's finalize kernel writes the raw sum into every rank's
            # mailbox, its epilogue waits for the flags -- no collective call, no extra launch.  Not the default until a node run
            # has measured it against ncclAllReduce.
            def peer_stage():
                nonlocal peers
                peers = PeerGroup(ctx, max_elems=2 * d)
                ctx.set_peers(peers)
                small.fill_(1.0)
                ctx.peer_allreduce(small)
                ctx.synchronize()
                assert float(small[0]) == world, (float(small[0]), world)

            ok = pf.stage("peer", peer_stage)
            if _vote(dist, torch, ok, dev):
                collective = "peer mailboxes over HIP IPC (one direct write per rank + flags), fused into finalize / epilogue: no collective call"
            else:
                if ok:
                    pf.rec["peer"] = "ok here, failed on another rank"
                try:
                    ctx.set_peers(None)
                    if peers is not None:
                        peers.close()
                except Exception:   # noqa: BLE001
                    pass
                peers = None
        if comm is None and peers is None:
            hook = AllReduceHook(dev)
            ctx.set_allreduce(hook)
            collective = (f"torch.distributed all_reduce(d+1) per step, backend {dist.get_backend()}"
                          + (" (= RCCL)" if dist.get_backend() == "nccl" else " (host-staged: one-GPU rehearsal, not xGMI)"))

    # ---- synthetic problem, generated on the device, keyed by the GLOBAL (row, col): SURVEY.md section 8d ------------
    A = torch.empty((n_local, d), dtype=tdt, device=dev)
    b = torch.empty((n_local,), dtype=tdt, device=dev)
    ctx.synth_normal(A, row0, seed=0, scale=1.0 / np.sqrt(d))            # ||a_i||^2 ~ 1
    rng = np.random.default_rng(0)
    x_true = rng.standard_normal(d) * (rng.random(d) < 0.05)
    x_true_d = torch.from_numpy(x_true).to(dev, tdt)
    lam_f = 1.0 if logistic else float(N_total)                          # LeastSquares(.., R(N)) test_lasso.jl:54
    F = PackedF(L.LOSS_LOGISTIC if logistic else L.LOSS_LS, A, b, lam_f, N_total=N_total, row0=row0)
    ctx.synth_targets(F, x_true_d, noise=0.01 if not logistic else 0.1, labels=logistic, seed=0, b_out=b)
    lam_g = 1e-3 if not logistic else 1.0 / N_total
    g = ProxG(L.PROX_L1, lam=lam_g)
    L_max = (lam_f if not logistic else 0.25) * 1.3                      # ||a_i||^2 <= ~1.3 for d = 1024
    gamma = 1.0 / (7.0 * L_max) if not logistic else 1.0 / (10.0 * L_max)   # test_lasso.jl:164 / test_logistic_l1.jl:126
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
        # sweep fuses them into finalize / epilogue and pays no launch for them).  On real peers (backend nccl, several GPUs) this is
        # the FIRST time the mailboxes cross xGMI: it runs after the timed region, but a fault in it would still cost the whole line,
        # so there it needs CIAO_BENCH_PEER_PROBE=1 (or CIAO_BENCH_COLLECTIVE=peer); the one-GPU rehearsals always run it.
        probe = os.environ.get("CIAO_BENCH_PEER_PROBE", "1" if (backend != "nccl" or world == 1) else "0") == "1"
        if peers is not None:
            allreduce_peer_us = allreduce_us
        elif probe:
            pg, good = None, False

            def peer_probe():
                nonlocal pg, allreduce_peer_us
                pg = PeerGroup(ctx, max_elems=2 * d)
                ctx.set_peers(pg)           # displaces the RCCL communicator / the hook; "off" below puts it back (ciao_ctx_set_peers)
                for _ in range(10):
                    ctx.peer_allreduce(buf)
                fence()
                allreduce_peer_us = _timed_events(torch, ctx.stream or torch.cuda.current_stream(), lambda: ctx.peer_allreduce(buf), reps) * 1e6
                fence()

            try:
                good = pf.stage("peer_probe", peer_probe)
            finally:   # whatever happened in between: peers off (the displaced collective is back in place), mailboxes unmapped
                try:
                    ctx.set_peers(None)
                    if pg is not None:
                        pg.close()
                except Exception as e:   # noqa: BLE001
                    print(f"[bench] peer probe teardown failed: {e!r}", file=sys.stderr)
            # the ranks agree on whether the probe worked before anything else uses the collective again
            if not _vote(dist, torch, good, dev):
                allreduce_peer_us = None

    units = float(N_total) * args.steps                                  # per-sample grad+prox updates processed by all ranks
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
        traffic_src = f"static: profiles/pmc_traffic.json ({tj.get('measured_at')}, {counted})" if traffic else None
    except Exception:
        traffic = None

    out = {
        "metric": METRIC,
        "metric_definition": "the SVRG full-gradient + prox sweep (SVRG_basic.jl:87-92): one update = one row's gradient into the aggregate "
                             "+ its share of the fused prox; value = N_total * steps / time.  The dependent-chain updates of SURVEY 8d "
                             "are the *_per_sec entries below.",
        "value": value,
        "unit": UNIT,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "epochs_per_sec": args.steps / elapsed,
        # everything in a step that is not the sweep kernel: finalize + epilogue, launch gaps and -- with several ranks -- the
        # all-reduce of the d+1 scalars (BASELINE.md section 2 asks for that figure in microseconds)
        "step_overhead_us_beyond_sweep_kernel": (elapsed / args.steps - k_avg_s) * 1e6,
        "allreduce_us_per_step": allreduce_us,
        "allreduce_us_per_step_peer_mailboxes": allreduce_peer_us,
        "rccl_ranks": rccl_ranks,
        "collective_preflight": pf.rec if (world > 1 or force_dist) else None,
        "config": {"workload": f"{'l1_logistic' if logistic else 'lasso'}_svrg_fullgrad_prox_sweep",
                   "N_total": N_total, "rows_per_gpu": n_local, "d": d, "f": "LeastSquares(a_i,b_i,N)" if not logistic else "Precompose(LogisticLoss)",
                   "g": f"NormL1({lam_g:g})", "gamma": gamma, "parallelism": f"rows_sharded_x{world}",
                   "collective": collective, "launcher": "bench.py spawned the ranks" if os.environ.get("CIAO_BENCH_SPAWNED") else
                   ("external launcher (WORLD_SIZE set)" if world > 1 else "single process")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel_name,
                     "kernel_avg_ms": k_avg_s * 1e3, "kernel_launches": k_n, "algorithmic_bytes_per_launch": alg_bytes},
    }

    # ---- everything below runs on rank 0 only, after the timed region; with several ranks the others wait at the final barrier ----
    # cpu_baseline: the single-threaded oracle (reference-shaped sequential pass) on a bounded sample of this rank's rows
    if rank == 0 and not args.no_cpu:
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
            out["cpu_baseline"] = {"value": n_s * reps / t_cpu, "unit": UNIT, "cores": 1, "kind": "port",
                                   "sample": f"first {n_s} rows of the same A (d={d}, {args.dtype}), {reps} sequential full passes, "
                                             f"{t_cpu:.1f} s; orc_full_pass (SVRG_basic.jl:87-92 restated)",
                                   "all_cores": {"value": n_s / t_omp, "cores": int(nt), "kind": "openmp sweep (not the reference's shape)"},
                                   "host_cores": os.cpu_count()}
            del A_h, b_h, op
        except Exception as e:  # the baseline must never cost us the bench line
            out["cpu_baseline"] = {"value": None, "unit": UNIT, "cores": 1, "kind": "port", "sample": f"failed: {e!r}"[:200]}

    # the updates SURVEY.md 8d defines: the sequential SVRG / SAGA chains and the Finito batch, at the configs' own sizes
    args.cpu_sweep_rate = (out.get("cpu_baseline") or {}).get("value")
    if rank == 0 and not args.no_chains:
        cctx = ctx
        try:
            if world > 1 or force_dist:
                # this rank's shard as a problem of its own (N = the shard's rows), on a context without a collective: the chains
                # are one dependent chain and do not shard
                cctx = Context(dev.index)
                Fc = PackedF(F.loss, A, b, 1.0 if logistic else float(n_local))
                gam_c = gamma * (N_total / n_local) if not logistic else gamma
            else:
                Fc, gam_c = F, gamma
            out.update(chain_figures(cctx, dev, Fc, g, gam_c, A, b, n_local, d, args, L, np, torch))
        except Exception as e:
            out["chain_figures_error"] = repr(e)[:300]
        finally:
            if cctx is not ctx:
                cctx.close()

    if not args.no_extras and world == 1 and rank == 0:
        ex_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
        ex_path = os.path.join(ex_dir, "bench_extras.json")
        try:
            import bench_extras
            del A, b, F
            torch.cuda.empty_cache()
            extra = bench_extras.run(ctx, dev)
        except Exception as e:
            extra = {"error": repr(e)}
        try:
            with open(ex_path, "w") as fh:
                json.dump(extra, fh, indent=1)
            out["extras_file"] = os.path.relpath(ex_path, ROOT)
        except Exception as e:   # noqa: BLE001
            out["extras_file"] = f"not written: {e!r}"[:120]
        print("[bench] extra " + json.dumps(extra), file=sys.stderr, flush=True)

    if rank == 0:
        print(json.dumps(_compact(out), separators=(",", ":")), flush=True)
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


def _compact(o, nd=6):
    """Floats to `nd` significant digits: the line has to fit the tail of a log."""
    if isinstance(o, float):
        return float(f"{o:.{nd}g}")
    if isinstance(o, dict):
        return {k: _compact(v, nd) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_compact(v, nd) for v in o]
    return o


def _roof(rate, bytes_per_unit):
    return {"bound": "hbm", "achieved": rate * bytes_per_unit / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": rate * bytes_per_unit / 1e9 / HBM_PEAK_GBS, "bytes_per_unit": bytes_per_unit}


def chain_figures(ctx, dev, F, g, gamma, A, b, N, d, args, L, np, torch):
    """The three other kinds of update of SURVEY.md 8d, each {value, roofline, cpu_baseline}:
    * ONE real SVRG outer iteration (SVRG_basic.jl:71-96: m = N dependent updates + the full pass) on the resident problem;
    * SAGA steps (SAGA_basic.jl:53-68) in fp64 on the resident rows + an N x d table, and at BASELINE config #3's size (N x d fp32 + table);
    * Finito batches (Finito_basic.jl:109-118) at BASELINE config #5's per-rank shape (1.25M x 4096 fp32 + table): r = 4096 and the
      512-row share one rank of eight has of such a batch.
    Bytes per update from SURVEY.md 8d (S3 d*s+8, G3 3*d*s+8, F3 3*d*s+s+8); the chains are dependent (step k+1 reads the iterate step k
    wrote), so their bound is not reachable by construction."""
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
        # cpu_baseline of the chain: the single-threaded oracle (orc_svrg_inner: SVRG_basic.jl:73-82 restated) on the FIRST updates of
        # this very epoch -- the rows they visit gathered to the host, the device's init state -- for a bounded number of updates
        try:
            cpu_svrg = cpu_chain_baseline("svrg", ctx, F, g, gamma, hidx[:args.cpu_chain_updates], N, (av, z, zf, w), None, args, np, torch)
        except Exception as e:
            cpu_svrg = {"value": None, "unit": "updates/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"[:200]}
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
    res["svrg_updates_per_sec"] = {"value": upd, "us_per_update": 1e6 / upd, "m": N, "d": d, "dtype": args.dtype,
                                   "what": "SVRG inner cycle (SVRG_basic.jl:73-82): one dependent chain on one workgroup, latency-bound",
                                   "roofline": _roof(upd, d * es + 8)}
    if cpu_svrg is not None:
        res["svrg_updates_per_sec"]["cpu_baseline"] = cpu_svrg
    ep_bytes = N * (d * es + 8) + N * (d * es + es)
    ekey = f"svrg_epochs_per_sec_N{N // 1_000_000}M" if N % 1_000_000 == 0 else f"svrg_epochs_per_sec_N{N}"
    res[ekey] = {"value": 1.0 / t_epoch, "seconds_per_epoch": t_epoch, "inner_cycle_s": t_chain, "full_pass_s": t_sweep,
                 "what": "one SVRG outer iteration: m = N updates + tail + full-gradient sweep (SVRG_basic.jl:71-96), measured once",
                 "roofline": _roof(1.0 / t_epoch, ep_bytes)}
    if cpu_svrg is not None and cpu_svrg.get("value") and args.cpu_sweep_rate:
        # an epoch on one host core = N updates + N sample-gradients of the full pass, at the two rates measured in this run
        t_cpu_epoch = N / cpu_svrg["value"] + N / args.cpu_sweep_rate
        res[ekey]["cpu_baseline"] = {"value": 1.0 / t_cpu_epoch, "unit": "epochs/s", "cores": 1, "kind": "port",
                                     "sample": f"extrapolated from the two oracle rates of this run: {t_cpu_epoch:.0f} s per epoch on one core"}
    del idx
    # ---- the full passes of K lockstep solves over these rows as ONE pass on the matrix cores (ciao_full_gradient_multi; an EXTENSION
    # beyond the reference's one-problem-per-call API, labelled so): 4 N d K flops against the dense MFMA peak of the dtype
    try:
        if d in (256, 512, 1024):
            Km = 64
            xs = [torch.randn(d, dtype=tdt, device=dev) * 0.1 for _ in range(Km)]
            avs = [torch.empty_like(x) for x in xs]
            ctx.full_gradient_multi(F, xs, avs)
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.full_gradient_multi(F, xs, avs)
            ctx.synchronize()
            tm = time.perf_counter() - t0
            peak = 78.6 if es == 8 else 157.3
            tf = 4.0 * N * d * Km / tm / 1e12
            res["full_pass_K_solves"] = {"K": Km, "seconds": tm, "vs_K_sweeps": Km * t_sweep / tm,
                                         "roofline": {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak},
                                         "what": "EXTENSION beyond the reference: K solves' full passes as one pass over A (mrhs_kernel)"}
            del xs, avs
    except Exception as e:   # noqa: BLE001
        res["full_pass_K_solves"] = {"error": repr(e)[:200]}
    Ns = args.chain_rows or N

    def saga_figure(Fs, gs, gam, x1, tag, kern_note):
        sdt = Fs.A.dtype
        ses = 8 if sdt == torch.float64 else 4
        table = torch.empty((Fs.N, d), dtype=sdt, device=dev)
        sav, sz = torch.empty_like(x1), torch.empty_like(x1)
        ctx.saga_init(Fs, gs, gam, x1, table, sav, sz)
        k = 400_000
        hsidx = st.rand_indices(Fs.N, k)
        sidx = ctx._idx(hsidx)
        cpu_saga = None
        if not args.no_cpu:
            try:
                cpu_saga = cpu_chain_baseline("saga", ctx, Fs, gs, gam, hsidx[:args.cpu_chain_updates], Fs.N, (sav, sz), table, args, np, torch)
            except Exception as e:
                cpu_saga = {"value": None, "unit": "updates/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"[:200]}
        ctx.saga_steps(Fs, gs, gam, False, sidx[:4096], table, sav, sz)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.saga_steps(Fs, gs, gam, False, sidx, table, sav, sz)
        ctx.synchronize()
        upd = k / (time.perf_counter() - t0)
        fig = {"value": upd, "us_per_update": 1e6 / upd, "steps": k, "N": Fs.N, "d": d, "dtype": tag, "what": kern_note,
               "kernel": ctx.last_kernel().split(" grid")[0], "roofline": _roof(upd, 3 * d * ses + 8)}
        if cpu_saga is not None:
            fig["cpu_baseline"] = cpu_saga
        del table, sidx
        torch.cuda.empty_cache()
        return fig

    # ---- SAGA in fp64 (the reference's default R, SAGA.jl:108-109) on the resident rows: Lasso, an N x d fp64 table beside A ---------
    saga64 = None
    if tdt == torch.float64 and 2 * N * d * 8 < 200e9:
        try:
            g64 = 1.0 / (3.0 * 1.3 * F.lam) if F.loss == L.LOSS_LS else 1.0 / (3 * 0.25 * 1.3)
            saga64 = saga_figure(F, g, g64, torch.zeros(d, dtype=tdt, device=dev), "f64",
                                 "SAGA step (SAGA_basic.jl:53-68) in fp64 on the metric's own rows + an N x d table")
        except Exception as e:   # noqa: BLE001
            saga64 = {"error": repr(e)[:200]}
    # ---- SAGA at config #3: l1-logistic, fp32, N x d data + N x d table -------------------------------------------------
    if d * 4 * Ns * 2 + A.numel() * A.element_size() < 250e9:
        A32 = torch.empty((Ns, d), dtype=torch.float32, device=dev)
        y32 = torch.empty((Ns,), dtype=torch.float32, device=dev)
        ctx.synth_normal(A32, 0, seed=1, scale=1.0 / np.sqrt(d))
        rng = np.random.default_rng(1)
        xt = torch.from_numpy(rng.standard_normal(d) * (rng.random(d) < 0.05)).to(dev, torch.float32)
        Fs = PackedF(L.LOSS_LOGISTIC, A32, y32, 1.0)
        ctx.synth_targets(Fs, xt, noise=0.1, labels=True, seed=1, b_out=y32)
        res["saga_updates_per_sec"] = saga_figure(Fs, ProxG(L.PROX_L1, lam=1.0 / Ns), 1.0 / (3 * 0.25 * 1.3), torch.ones(d, dtype=torch.float32, device=dev),
                                                  "f32", "SAGA step (SAGA_basic.jl:53-68) at BASELINE config #3 (l1-logistic, table in HBM)")
        if saga64 is not None:
            res["saga_updates_per_sec"]["f64"] = saga64
        del A32, y32, Fs
        torch.cuda.empty_cache()
    elif saga64 is not None:
        res["saga_updates_per_sec"] = saga64

    # ---- Finito at config #5's per-rank shape: 1.25M x 4096 fp32 + the table; batches of 4096 and the 512-row share ---------------
    try:
        res["finito_samples_per_sec"] = finito_figure(ctx, dev, args, L, np, torch)
    except Exception as e:   # noqa: BLE001 -- a figure must not cost the line
        res["finito_samples_per_sec"] = {"error": repr(e)[:200]}
    return res


def finito_figure(ctx, dev, args, L, np, torch):
    """One sample of a Finito batch (Finito_basic.jl:109-118; SURVEY 8d: 3*d*s + s + 8 bytes per sample): static cyclic batches
    (`ciao_finito_steps_blocks`) at BASELINE config #5's per-rank shape."""
    from ciaoalgorithms_jl_amd.device import PackedF, ProxG
    N5, d5 = (args.chain_rows or 1_250_000), 4096
    A5 = torch.empty((N5, d5), dtype=torch.float32, device=dev)
    b5 = torch.empty((N5,), dtype=torch.float32, device=dev)
    ctx.synth_normal(A5, 0, seed=5, scale=1.0 / np.sqrt(d5))
    rng = np.random.default_rng(5)
    xt = torch.from_numpy(rng.standard_normal(d5) * (rng.random(d5) < 0.05)).to(dev, torch.float32)
    F5 = PackedF(L.LOSS_LS, A5, b5, float(N5))
    ctx.synth_targets(F5, xt, noise=0.01, labels=False, seed=5, b_out=b5)
    g5 = ProxG(L.PROX_L1, lam=1e-3)
    gam = (0.999 / 1.3 * (1.0 + 0.1 * torch.frac(torch.arange(N5, device=dev, dtype=torch.float64) * 0.6180339887498949))).float()
    hg = ctx.hat_gamma(gam)                                                   # gamma_i = alpha N / L_i, per sample (Finito_basic.jl:69)
    x0 = torch.zeros(d5, dtype=torch.float32, device=dev)
    table = torch.empty((N5, d5), dtype=torch.float32, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.finito_init(F5, g5, gam, hg, x0, table, av, z)
    bytes_per = 3 * d5 * 4 + 4 + 8
    fig = {}
    for r, nb, key in ((4096, 200, None), (512, 600, "share_512_rows")):
        nblk = N5 // r
        first = (np.arange(1, nb + 1, dtype=np.int64) % nblk) * r            # cyclic: the first step uses batch 2 (Finito_basic.jl:99)
        length = np.full(nb, r, np.int64)
        ctx.finito_steps_blocks(F5, g5, gam, hg, first[:8], length[:8], table, av, z)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.finito_steps_blocks(F5, g5, gam, hg, first, length, table, av, z)
        ctx.synchronize()
        t = time.perf_counter() - t0
        # the rows kernel's own duration from HIP events, in a pass of its own (an event pair per launch costs the batches 3-6 us each)
        ctx.timing_enable(True)
        ctx.timing_read()
        ctx.finito_steps_blocks(F5, g5, gam, hg, first[:50], length[:50], table, av, z)
        k_ms, k_n = ctx.timing_read()
        ctx.timing_enable(False)
        rate = nb * r / t
        e = {"value": rate, "batch": r, "us_per_batch": t / nb * 1e6, "rows_kernel_us": k_ms / max(k_n, 1) * 1e3,
             "kernel": ctx.last_kernel().split(" grid")[0], "roofline": _roof(rate, bytes_per)}
        if key is None:
            fig.update(e)
            fig.update({"N": N5, "d": d5, "dtype": "f32",
                        "what": "one sample of a Finito batch (Finito_basic.jl:109-118) at BASELINE config #5's per-rank shape, static batches"})
        else:
            e["what"] = "the 512 rows one rank of eight holds of a 4096-batch (no collective in this figure)"
            fig[key] = e
    if not args.no_cpu:
        try:
            from oracle import oracle as O
            r, nbc = 4096, 2
            rows = torch.arange(r, (nbc + 1) * r, device=dev)
            A_t, b_t, gam_t = A5[rows].cpu().numpy(), b5[rows].cpu().numpy(), gam[rows].cpu().numpy()
            h_tab, h_av, h_z = table[rows].cpu().numpy(), av.cpu().numpy().copy(), z.cpu().numpy().copy()
            op = O.Problem("ls", A_t, b_t, float(N5), N_total=N5)
            og = O.Prox("l1", lam=1e-3)
            batches = [np.arange(k * r, (k + 1) * r) for k in range(nbc)]
            reps, t_cpu = 0, 0.0
            while t_cpu < args.cpu_chain_seconds and reps < 200:
                t1 = time.perf_counter()
                O.finito_steps(op, og, gam_t, np.float32(hg), batches, h_tab, h_av, h_z)
                t_cpu += time.perf_counter() - t1
                reps += 1
            n = reps * nbc * r
            fig["cpu_baseline"] = {"value": n / t_cpu, "unit": "samples/s", "cores": 1, "kind": "port", "samples": n,
                                   "sample": f"the first {nbc} batches of 4096 of the same run (rows and table rows gathered to the host, d=4096, float32), "
                                             f"{reps} time(s) over, {t_cpu:.1f} s; orc_finito_steps (Finito_basic.jl:109-118 restated)",
                                   "host_cores": os.cpu_count()}
        except Exception as e:   # noqa: BLE001
            fig["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"[:200]}
    del A5, b5, table, F5
    torch.cuda.empty_cache()
    return fig


def cpu_chain_baseline(alg, ctx, F, g, gamma, hidx, N, state, table, args, np, torch):
    """The oracle's sequential chain (single thread) on the rows the first len(hidx) updates of the device's run visit: rows (and
    SAGA table rows) gathered to the host, indices remapped, 1/N of the whole problem, the device's own start state.  The
    device state is not touched.  Repeats the same updates until cpu_chain_seconds have passed (the state keeps moving)."""
    from oracle import oracle as O
    touched, remap = np.unique(hidx, return_inverse=True)
    t = torch.from_numpy(touched).to(F.A.device)
    A_t, b_t = F.A[t].cpu().numpy(), F.b[t].cpu().numpy()
    logistic = (F.loss == O.LOSS_LOGISTIC)
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
    what = "orc_svrg_inner (SVRG_basic.jl:73-82 restated)" if alg == "svrg" else "orc_saga_steps (SAGA_basic.jl:53-68 restated)"
    return {"value": n / t_cpu, "unit": "updates/s", "us_per_update": t_cpu / n * 1e6, "cores": 1, "kind": "port", "updates": n,
            "sample": f"the first {len(remap)} updates of the same run ({len(touched)} rows{' + table rows' if table is not None else ''} "
                      f"gathered to the host, d={A_t.shape[1]}, {A_t.dtype.name}), {reps}x, {t_cpu:.1f} s; {what}",
            "host_cores": os.cpu_count()}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)          # parent: no torch, no HIP, no GPU
    if args.dry_run:
        return dry_rank(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
