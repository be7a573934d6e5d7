#!/usr/bin/env python
"""Latency-regime timings (BASELINE.json configs[2] / configs[4], the reference's real workloads --
sampled sub-graphs).  The datasets are not available offline, so the inputs are sampler-shaped synthetic batches with the
layout facts of the reference pipeline (pyhgt_amd.sampled.synthetic_sampled_batch: type-contiguous ids, `self` runs first,
target-sorted runs, edge_time in [111, 129], min in-degree 1), handed over once through `to_torch`'s wire format
(hgt_plan_build: radix sorts) and once through the device-side hand-off (to_device_graph -> hgt_plan_from_sorted)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhgt_amd import HGTConv, GNN, GraphPlan  # noqa: E402
from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout, to_device_graph  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


PRECS = tuple(os.environ.get("HGT_SMALL_PRECS", "bf16x3,f16x3,fp32").split(","))      # e.g. HGT_SMALL_PRECS=f16x3 under rocprofv3


def main():
    dev = "cuda:0"
    res = {}
    # configs[2] surrogate: ogbn-mag sampled sub-graph (sample_depth 6, sample_width 128), T=4 R=9 (incl. self), d=256 H=8
    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    x, nt, tm, ei, et, _, edge_dict = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    T, R, d, H = 4, len(edge_dict), 256, 8
    N, E = nt.numel(), et.numel()
    dg = to_device_graph(*batch, device=dev)
    src32, dst32 = dg[3][0].int().contiguous(), dg[3][1].int().contiguous()
    time32 = dg[2].int().contiguous()
    rel_ptr = torch.searchsorted(dg[4], torch.arange(R + 1, device=dev)).int()
    type_off = torch.searchsorted(dg[1], torch.arange(T + 1, device=dev)).int()
    us_build = timeit(lambda: GraphPlan(nt, ei, et, tm, T, R), iters=100, warm=10)
    us_sorted = timeit(lambda: GraphPlan.from_sorted(dg[1], dg[3], dg[4], dg[2], src32, dst32, time32, rel_ptr, type_off, T, R),
                       iters=100, warm=10)
    res["c3"] = {"N": N, "E": E, "plan_build_us": us_build, "plan_from_sorted_us": us_sorted}
    for prec in PRECS:
        layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=prec).eval().to(dev)
        plan = GraphPlan(nt, ei, et, tm, T, R)
        with torch.no_grad():
            us = timeit(lambda: layer(x, nt, ei, et, tm, plan=plan))
        res["c3"][prec + "_layer_us"] = us
        print("c3 surrogate  N=%d E=%d d=%d %-6s: %.1f us / layer (plan cached), plan build %.1f us (radix) / %.1f us (from sorted), "
              "%.1f M edges/s" % (N, E, d, prec, us, us_build, us_sorted, E / us))
    # configs[4] surrogate: OAG sampled batch, T=5 R=33, batch 256, d=400 H=8, in_dim 1169, 2-layer GNN
    batch = synthetic_sampled_batch("oag", n_seed=256, width=128, depth=6, feat_dim=1169, mean_degree=1.2, seed=5)
    x, nt, tm, ei, et, _, edge_dict = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    T, R, d, H, din = 5, len(edge_dict), 400, 8, 1169
    N, E = nt.numel(), et.numel()
    res["c5"] = {"N": N, "E": E}
    for prec in PRECS:
        gnn = GNN(din, d, T, R, H, 2, prev_norm=True, last_norm=True, use_RTE=True).eval().to(dev)
        for gc in gnn.gcs:
            gc.base_conv.precision = prec
        with torch.no_grad():
            us = timeit(lambda: gnn(x, nt, tm, ei, et), iters=100, warm=10)
        res["c5"][prec + "_gnn2_us"] = us
        print("c5 surrogate  N=%d E=%d in=%d d=%d 2-layer GNN %-6s: %.1f us / forward (plan cached)" % (N, E, din, d, prec, us))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
