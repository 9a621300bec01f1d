#!/usr/bin/env python3
"""K independent chains on K streams over the same A (bench_extras.lambda_path): aggregate updates/s for K = 1 ... 256.
GPU_MAX_HW_QUEUES must be set before the HIP runtime starts: `GPU_MAX_HW_QUEUES=32 python tools/lambda_path.py [svrg|saga]`."""
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
if alg == "saga":
    Ks = tuple(k for k in Ks if k <= 32)          # a 4 GB table per chain at N = 1M
r = bench_extras.lambda_path(dev, Ks=Ks, alg=alg)
print(json.dumps(r))
for c in r["curve"]:
    print(f"K={c['K']:4d}: {c['updates_per_s'] / 1e6:8.2f} M updates/s  ({c['us_per_update_per_chain']:.3f} us per update and chain, "
          f"{c['alg_GBps']:.0f} GB/s = {c['frac_of_hbm_bound']:.4f} of the HBM bound)", file=sys.stderr)
