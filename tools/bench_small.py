#!/usr/bin/env python
"""Latency-regime timings (c3 / c5 sized sampled subgraphs; development aid, not the judged bench)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhgt_amd import HGTConv, GNN, GraphPlan  # noqa: E402
from pyhgt_amd.synth import synthetic_typed_graph  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    dev = "cuda:0"
    # c3 surrogate: ogbn-mag sampled subgraph, T=4 R=9 (incl. self), N=2548 E=31324, d=256 H=8, RTE on
    N, E, d, H, T, R = 2548, 31324, 256, 8, 4, 9
    x, nt, ei, et, tm = [t.to(dev) for t in synthetic_typed_graph(N, E, d, T, R, seed=3, schema=True)]
    for prec in ("bf16x3", "fp32"):
        layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=prec).eval().to(dev)
        plan = GraphPlan(nt, ei, et, tm, T, R)
        with torch.no_grad():
            us = timeit(lambda: layer(x, nt, ei, et, tm, plan=plan))
            us_plan = timeit(lambda: GraphPlan(nt, ei, et, tm, T, R), iters=50, warm=5)
        print("c3 surrogate  N=%d E=%d d=%d %-6s: %.1f us / layer (plan cached), plan build %.1f us, %.1f M edges/s" % (
            N, E, d, prec, us, us_plan, E / us))
    # c5 surrogate: OAG sampled batch, T=5 R=33, N=4096, d=400 H=8, in_dim 1169, 2-layer GNN
    N, E, d, H, T, R, din = 4096, 40000, 400, 8, 5, 33, 1169
    x, nt, ei, et, tm = [t.to(dev) for t in synthetic_typed_graph(N, E, din, T, R, seed=5, schema=True)]
    for prec in ("bf16x3", "fp32"):
        gnn = GNN(din, d, T, R, H, 2, prev_norm=True, last_norm=True, use_RTE=True).eval().to(dev)
        for gc in gnn.gcs:
            gc.base_conv.precision = prec
        with torch.no_grad():
            us = timeit(lambda: gnn(x, nt, tm, ei, et), iters=100, warm=10)
        print("c5 surrogate  N=%d E=%d in=%d d=%d 2-layer GNN %-6s: %.1f us / forward (plan cached)" % (N, E, din, d, prec, us))


if __name__ == "__main__":
    main()
