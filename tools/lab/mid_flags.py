#!/usr/bin/env python
"""Where does the item-parallel aggregation stop paying?  One layer, d=256 / 8 heads and d=512 / 8 heads, growing graphs."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pyhgt_amd import HGTConv, GraphPlan
from pyhgt_amd.synth import synthetic_typed_graph


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


dev = "cuda:0"
SIZES = [int(v) for v in os.environ.get("MID_SIZES", "8000,16000,24000,32000,48000,64000").split(",")]
FLAGS = [int(v) for v in os.environ.get("MID_FLAGS", "16,32").split(",")]
for d, H, T, R in ((256, 8, 4, 8), (512, 8, 4, 9)):
    for N in SIZES:
        E = 10 * N
        x, nt, ei, et, tm = [t.to(dev) for t in synthetic_typed_graph(N, E, d, T, R, seed=N)]
        plan = GraphPlan(nt, ei, et, tm, T, R)
        res = []
        for flags in FLAGS:
            layer = HGTConv(d, d, T, R, H, 0.2, True, True).eval().to(dev)
            layer.kernel_flags = flags
            with torch.no_grad():
                res.append(timeit(lambda: layer(x, nt, ei, et, tm, plan=plan)))
        print("d=%d N=%d E=%d:" % (d, N, E), ", ".join("flags %d: %.1f us" % (f, r) for f, r in zip(FLAGS, res)))
