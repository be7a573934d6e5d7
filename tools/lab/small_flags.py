#!/usr/bin/env python
"""Latency regime A/B: kernel_flags (VALU aggregation, unfused update) and precisions on the c3 / c5 layer shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyhgt_amd import HGTConv, GraphPlan
from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout


def timeit(fn, iters=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


dev = "cuda:0"
for name, schema, kw, d, H in (("c3", "mag", dict(n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3), 256, 8),
                               ("c5", "oag", dict(n_seed=256, width=128, depth=6, feat_dim=400, mean_degree=1.2, seed=5), 400, 8)):
    batch = synthetic_sampled_batch(schema, **kw)
    x, nt, tm, ei, et, _, edge_dict = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    T, R = int(nt.max()) + 1, len(edge_dict)
    plan = GraphPlan(nt, ei, et, tm, T, R)
    for prec in ("bf16x3", "f16x3"):
        for flags in (0, 16, 32):
            layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=prec).eval().to(dev)
            layer.kernel_flags = flags
            with torch.no_grad():
                us = timeit(lambda: layer(x, nt, ei, et, tm, plan=plan))
            print("%s N=%d E=%d d=%d %s flags=%d: %.1f us" % (name, nt.numel(), et.numel(), d, prec, flags, us))
