#!/usr/bin/env python3
"""Latency budget of ONE batch of BASELINE config #5 as a rank of the 8-GPU run sees it: Finito, d = 4096
fp32, batches of 4096 samples = 512 rows per rank.  Measured on the one GPU of the test box with TWO ranks (two processes,
HIP IPC mailboxes): per batch -- the rows kernel (HIP events of the library), the whole batch, and the collective on its own
-- for (a) one rank without a collective, (b) two ranks through the peer mailboxes, (c) two ranks through a host-staged
torch.distributed all-reduce (the rehearsal stand-in for RCCL).  Writes gpurun_out/c5_share_batch_budget.json.
The two ranks SHARE one GPU here (their kernels interleave on it): on a node each rank has a GPU of its own and the mailbox
writes cross xGMI instead of staying in one HBM."""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_SHARE, D, N_LOCAL, NB = 512, 4096, 200_000, 1500


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_main(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import ciao_loader
    ciao_loader.load()
    from ciaoalgorithms_jl_amd import _lib as L
    from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
    from ciaoalgorithms_jl_amd.parallel import AllReduceHook, PeerGroup
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = Context(0)
    N_total = N_LOCAL * world
    A = torch.empty((N_LOCAL, D), dtype=torch.float32, device=dev)
    b = torch.empty((N_LOCAL,), dtype=torch.float32, device=dev)
    ctx.synth_normal(A, rank * N_LOCAL, 1, 1 / np.sqrt(D))
    F = PackedF(L.LOSS_LS, A, b, float(N_total), N_total=N_total, row0=rank * N_LOCAL)
    ctx.synth_targets(F, torch.ones(D, dtype=torch.float32, device=dev), 0.1, False, 1, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    gam = torch.full((N_LOCAL,), 0.999 * N_total / (1.3 * N_total), dtype=torch.float32, device=dev)
    hg = 1.0 / (N_total / float(gam[0]))
    x0 = torch.zeros(D, dtype=torch.float32, device=dev)
    table = torch.empty((N_LOCAL, D), dtype=torch.float32, device=dev)
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    pg = hook = None
    if mode == "peer":
        pg = PeerGroup(ctx, max_elems=2 * D)
        ctx.set_peers(pg)
    elif mode == "hook":
        hook = AllReduceHook(dev)
        ctx.set_allreduce(hook)
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    first = (np.arange(NB, dtype=np.int64) % (N_LOCAL // R_SHARE)) * R_SHARE      # this rank's 512 rows of each batch of the sweep
    ln = np.full(NB, R_SHARE, np.int64)
    ctx.finito_steps_blocks(F, g, gam, hg, first[:20], ln[:20], table, av, z)
    ctx.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    ctx.finito_steps_blocks(F, g, gam, hg, first, ln, table, av, z)      # the batches as the solver issues them: no events in between
    ctx.synchronize()
    t = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    ctx.timing_enable(True)                                                 # a second pass with the rows kernel bracketed by events
    ctx.timing_read()
    ctx.finito_steps_blocks(F, g, gam, hg, first, ln, table, av, z)
    ctx.synchronize()
    k_ms, k_n = ctx.timing_read()
    ctx.timing_enable(False)
    res = {"mode": mode, "world": world, "us_per_batch": t / NB * 1e6, "rows_kernel_us": (k_ms / max(k_n, 1)) * 1e3 if k_n else None,
           "kernel": ctx.last_kernel()}
    if mode == "peer":
        buf = torch.zeros(D + 1, dtype=torch.float32, device=dev)
        for _ in range(20):
            ctx.peer_allreduce(buf)
        ctx.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            ctx.peer_allreduce(buf)
        e1.record()
        torch.cuda.synchronize()
        res["collective_alone_us"] = e0.elapsed_time(e1) / 500 * 1e3
        ctx.set_peers(None)
        pg.close()
    ctx.close()
    q.put((rank, res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run(world, mode):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=rank_main, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)[0][1]


if __name__ == "__main__":
    out = {"what": "BASELINE config #5 (Finito N=10M d=4096 fp32, batches of 4096 over 8 GPUs): one rank's share of a batch = 512 rows; "
                   "measured on ONE MI355X, two ranks sharing it", "rows_per_rank_and_batch": R_SHARE, "d": D, "batches": NB,
           "single_rank_no_collective": run(1, "none"), "two_ranks_peer_mailboxes": run(2, "peer"),
           "two_ranks_host_staged_hook": run(2, "hook")}
    a, p = out["single_rank_no_collective"], out["two_ranks_peer_mailboxes"]
    # the 8-GPU expectation: the single-rank batch (rows kernel + finalize + epilogue, measured with the GPU to itself) plus what the
    # exchange adds -- seven remote 16 KB slot writes and a flag per rank instead of local ones, one xGMI hop of latency for the
    # flags to arrive (about 1-2 us per hop between MI355X peers) -- with the fused form paying no launch for it
    xgmi_hop_us = 2.0
    exp = a["us_per_batch"] + xgmi_hop_us
    out["expected_8gpu"] = {"us_per_batch": exp, "samples_per_s": 4096 / exp * 1e6,
                            "arithmetic": f"single-rank batch {a['us_per_batch']:.1f} us (rows kernel {a['rows_kernel_us']:.1f} us) + one xGMI hop for the "
                                          f"flags ~{xgmi_hop_us} us; the two-rank one-GPU run measured {p['us_per_batch']:.1f} us per batch with both "
                                          f"ranks' kernels sharing the GPU, and {p.get('collective_alone_us', float('nan')):.1f} us for the exchange as two "
                                          "kernels of its own"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "c5_share_batch_budget.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))
