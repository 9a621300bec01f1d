#!/usr/bin/env python3
"""GPU tuning sweep for the rows kernels: options x dtypes -> HIP-event kernel time and algorithmic GB/s."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader  # noqa: E402

ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L  # noqa: E402
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    rows = []
    for dt, d, N in ((torch.float64, 1024, 4_000_000), (torch.float32, 1024, 8_000_000), (torch.float32, 4096, 2_000_000),
                     (torch.float64, 2048, 2_000_000), (torch.float32, 512, 16_000_000), (torch.float32, 256, 16_000_000),
                     (torch.float64, 128, 16_000_000)):
        es = 8 if dt == torch.float64 else 4
        A = torch.empty((N, d), dtype=dt, device=dev)
        b = torch.empty((N,), dtype=dt, device=dev)
        ctx.synth_normal(A, 0, 0, 1.0 / np.sqrt(d))
        F = PackedF(L.LOSS_LS, A, b, float(N))
        ctx.synth_targets(F, torch.ones(d, dtype=dt, device=dev), 0.01, False, 0, b)
        g = ProxG(L.PROX_L1, lam=1e-3)
        x, av, y = (torch.zeros(d, dtype=dt, device=dev) for _ in range(3))
        for pf in (0, 2):   # 2 = multi-row kernel (short rows only)
            if pf == 2 and d * es > 4096:
                continue
            ctx.set_option("sweep_multi", 1 if pf == 2 else 0)
            for grid in (128, 192, 256, 384, 512, 768, 1024, 2048):
                bpc = grid
                ctx.set_option("sweep_prefetch", 0 if pf == 2 else pf)
                ctx.set_option("sweep_blocks_per_cu", 16)
                ctx.set_option("sweep_grid", grid)
                for _ in range(2):
                    ctx.proxgrad_step(F, g, 1e-9, x, av, y)
                ctx.timing_enable(True)
                ctx.timing_read()
                for _ in range(8):
                    ctx.proxgrad_step(F, g, 1e-9, x, av, y)
                ms, n = ctx.timing_read()
                ctx.timing_enable(False)
                gbps = N * (d * es + es) / (ms / n * 1e-3) / 1e9
                rows.append({"dtype": str(dt), "d": d, "N": N, "prefetch": pf, "blocks_per_cu": bpc, "ms": ms / n, "GBps": gbps,
                             "kernel": ctx.last_kernel()})
                print(json.dumps(rows[-1]), flush=True)
        del A, b, F
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
