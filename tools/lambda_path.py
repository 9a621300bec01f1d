#!/usr/bin/env python3
"""K independent chains over the same A (bench_extras.lambda_path): aggregate updates/s for K = 1 ... 256, on K streams
(`GPU_MAX_HW_QUEUES=32 python tools/lambda_path.py [svrg|saga]`: the variable must be set before the HIP runtime starts) or as one
batched launch of K workgroups (`python tools/lambda_path.py svrg batch`; CIAO_KS=..., CIAO_N=... override the sizes)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ciao_loader
ciao_loader.load()
import bench_extras
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
alg = sys.argv[1] if len(sys.argv) > 1 else "svrg"
Ks = tuple(int(v) for v in os.environ.get("CIAO_KS", "1,2,4,8,16,32,64,128,256").split(","))
mode = sys.argv[2] if len(sys.argv) > 2 else "streams"
N = int(os.environ.get("CIAO_N", "1000000"))
if alg == "saga":
    Ks = tuple(k for k in Ks if k * N * 4096 <= 200e9)   # a table per chain
r = bench_extras.lambda_path(dev, Ks=Ks, N=N, alg=alg, mode=mode)
print(json.dumps(r))
for c in r["curve"]:
    print(f"K={c['K']:4d}: {c['updates_per_s'] / 1e6:8.2f} M updates/s  ({c['us_per_update_per_chain']:.3f} us per update and chain, "
          f"{c['alg_GBps']:.0f} GB/s = {c['frac_of_hbm_bound']:.4f} of the HBM bound)", file=sys.stderr)
