#!/usr/bin/env python3
"""Runs bench_extras on cuda:0 and prints one JSON object."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader  # noqa: E402

ciao_loader.load()
import bench_extras  # noqa: E402
from ciaoalgorithms_jl_amd.device import Context  # noqa: E402

torch.cuda.set_device(0)
ctx = Context(0)
print(json.dumps(bench_extras.run(ctx, torch.device("cuda", 0), quick="--quick" in sys.argv), indent=1))
ctx.close()
