#!/usr/bin/env python
"""c3 layer wall-clock under a few kernel_flags (A/B): python tools/lab/c3_flags.py [flags ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyhgt_amd import HGTConv, GraphPlan
from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout
dev = "cuda:0"
batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
x, nt, tm, ei, et, _, ed = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
T, R = 4, len(ed)
plan = GraphPlan(nt, ei, et, tm, T, R)
for prec in ("f16x3", "bf16x3"):
    for flags in [int(v) for v in sys.argv[1:]] or [0]:
        layer = HGTConv(256, 256, T, R, 8, 0.2, True, True, precision=prec).eval().to(dev)
        layer.kernel_flags = flags
        best = 1e9
        with torch.no_grad():
            for _ in range(30):
                layer(x, nt, ei, et, tm, plan=plan)
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(300):
                    layer(x, nt, ei, et, tm, plan=plan)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
        print("c3 %s flags=%d: %.2f us" % (prec, flags, best))
