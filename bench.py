#!/usr/bin/env python3
"""bench.py -- the SVRG full-gradient + prox sweep (the roofline row of BASELINE.md).

One "step" = one pass of the hot path over the resident batch of synthetic input:
        av = (1/N) sum_i grad f_i(x)  over ALL N rows of the row-major N x d matrix (each row read from HBM once)
        x+ = prox_{gamma g}(x - gamma * av)                      (soft threshold, fused into the reduce epilogue)
and the next step starts from x+ (a proximal-gradient iteration, so no step can be cached).

Workload at N=1: Lasso (f_i = LeastSquares(a_i', b_i, N), g = NormL1) with N = 10M rows, d = 1024, fp64 (81.92 GB of A):
the configuration BASELINE.json's metric is quoted on.  With --gpus P every rank holds its own 10M-row shard (weak
scaling; P = 8 is BASELINE config #4, N = 80M) and the d-vector sum is all-reduced with RCCL once per step.

Prints ONE JSON line (rank 0).  `value` = sample-gradient(+prox) updates per second over all ranks, inputs resident in
HBM before the timed region.  `roofline` is for the dominant kernel (rows_fast_kernel), timed with HIP events on the
stream it is launched on; `cpu_baseline` is the single-threaded CPU oracle on a bounded sample of the same rows.
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


def parse():
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
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
    from ciaoalgorithms_jl_amd.parallel import AllReduceHook, init_process_group_from_env, shard_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    dist = None
    force_dist = os.environ.get("CIAO_BENCH_FORCE_DIST") == "1"   # exercise the RCCL hook even with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        # RCCL ("nccl") is the product path; CIAO_BENCH_BACKEND=gloo exists only to rehearse several ranks on ONE GPU
        rank, world, local = init_process_group_from_env(os.environ.get("CIAO_BENCH_BACKEND", "nccl"))
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
    if world > 1 or force_dist:
        if os.environ.get("CIAO_BENCH_COLLECTIVE", "torch") == "rccl":   # the library calls ncclAllReduce itself
            from ciaoalgorithms_jl_amd.parallel import RcclComm
            ctx.set_rccl(RcclComm(rank, world, dev.index))
        else:                                                            # default: torch.distributed (backend nccl = RCCL)
            ctx.set_allreduce(AllReduceHook(dev))
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

    units = float(N_total) * args.steps                                  # sample-gradients processed by all ranks
    value = units / elapsed
    alg_bytes = n_local * (d * es + es)                                  # per launch: rows + b_i  
    k_avg_s = (k_ms / max(k_n, 1)) * 1e-3
    achieved = alg_bytes / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            traffic = json.load(fh).get(f"{args.loss}_{args.dtype}_N{n_local}_d{d}")
    except Exception:
        traffic = None

    out = {
        "metric": "sample_gradient_prox_updates_per_sec",
        "value": value,
        "unit": "updates/s",
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
        "config": {"workload": f"{'l1_logistic' if logistic else 'lasso'}_svrg_fullgrad_prox_sweep",
                   "N_total": N_total, "rows_per_gpu": n_local, "d": d, "f": "LeastSquares(a_i,b_i,N)" if not logistic else "Precompose(LogisticLoss)",
                   "g": f"NormL1({lam_g:g})", "gamma": gamma, "parallelism": f"rows_sharded_x{world}",
                   "collective": "rccl_allreduce(d+1)/step" if (world > 1 or force_dist) else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
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
            out["cpu_baseline"] = {"value": n_s * reps / t_cpu, "unit": "updates/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n_s} rows of the same A (d={d}, {args.dtype}), {reps} sequential full passes, "
                                             f"{t_cpu:.1f} s; oracle/ciao_oracle.c orc_full_pass (SVRG_basic.jl:87-92 restated)",
                                   "all_cores": {"value": n_s / t_omp, "cores": int(nt), "kind": "openmp sweep (not the reference's shape)"},
                                   "host_cores": os.cpu_count()}
        except Exception as e:  # the baseline must never cost us the bench line
            out["cpu_baseline"] = {"value": None, "unit": "updates/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}

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
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
