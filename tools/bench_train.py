#!/usr/bin/env python
"""Training-step timing of one HGTConv at BASELINE.json configs[1]: forward + backward through
pyhgt_amd/autograd.py against the inference forward, plan (and transposed plan) cached."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhgt_amd import HGTConv, GraphPlan  # noqa: E402
from pyhgt_amd.synth import synthetic_typed_graph  # noqa: E402


def main():
    dev = "cuda:0"
    N, E, d, T, R, H = (int(os.environ.get("HGT_TRAIN_N", 1000000)), int(os.environ.get("HGT_TRAIN_E", 10000000)), 256, 4, 8, 8)
    x, nt, ei, et, tm = [t.to(dev) for t in synthetic_typed_graph(N, E, d, T, R, seed=1)]
    layer = HGTConv(d, d, T, R, H, 0.2, True, False).to(dev)
    plan = GraphPlan(nt, ei, et, None, T, R)
    res = {}
    layer.eval()
    with torch.no_grad():
        for _ in range(3):
            layer(x, nt, ei, et, None, plan=plan)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            layer(x, nt, ei, et, None, plan=plan)
        torch.cuda.synchronize()
        res["inference_forward_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    layer.train()
    xg = x.clone().requires_grad_(True)
    g = torch.randn(N, d, device=dev)
    for phase in ("forward_ms", "forward_backward_ms"):
        for it in range(2 + 5):
            if it == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            out = layer(xg, nt, ei, et, None, plan=plan)
            if phase == "forward_backward_ms":
                out.backward(g)
                layer.zero_grad(set_to_none=True)
                xg.grad = None
        torch.cuda.synchronize()
        res["training_" + phase] = (time.perf_counter() - t0) / 5 * 1e3
    res["N"], res["E"] = N, E
    res["peak_mem_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps(res))


if __name__ == "__main__":
    main()
