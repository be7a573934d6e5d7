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


def training_step_bytes(N, E, d, H):
    """Algorithmic HBM bytes of one training step of HGTConv (4-argument form, LayerNorm on) in the same minimal-traffic convention
    as the forward's SURVEY 8(d) model: every kernel reads its inputs and writes its outputs once, every edge gathers one 4d-byte
    row per gather pass (no cache-reuse credit), ids cost 12 B per edge and pass, weights are ignored.  Kernel by kernel, in the
    order pyhgt_amd/autograd.py enqueues them."""
    Nd, EH, Eg = N * 4 * d, E * H * 4, E * (4 * d + 12)          # one fp32 feature array / one per-edge-per-head array / one gather pass
    fwd = {"project_qkv": 4 * Nd, "edge_logits": Eg + Nd + EH, "edge_softmax": 2 * EH, "edge_aggregate": Eg + EH + Nd,
           "a_linear": 2 * Nd, "node_update": 3 * Nd}
    bwd = {"node_update_bwd": 5 * Nd,                              # read grad_out, trans, x; write d_trans, dx_skip
           "gelu(agg)": 2 * Nd, "wgrad_a": 2 * Nd, "d_gelu = d_trans W_a": 2 * Nd, "gelu_bwd": 3 * Nd,
           "d_att (logits kernel on dagg, V, M^T)": Eg + Nd + EH, "head_dot rho": 2 * Nd, "softmax_bwd": 3 * EH,
           "spmm dQ": Eg + EH + Nd, "re-sort ds, att to the transposed plan": 8 * EH, "spmm dK": Eg + EH + Nd, "spmm dV": Eg + EH + Nd,
           "outer d relation_msg": Eg + EH + Nd, "outer d relation_att": Eg + EH + Nd,
           "wgrad_qkv": 4 * Nd, "dx = dqkv W_qkv": 4 * Nd, "dx += dx_skip": 3 * Nd}
    return fwd, bwd


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
    fwd_b, bwd_b = training_step_bytes(N, E, d, H)
    bf, bb = sum(fwd_b.values()), sum(bwd_b.values())
    t_f, t_s = res["training_forward_ms"], res["training_forward_backward_ms"]
    res["roofline"] = {
        "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
        "algorithmic_bytes": {"training_forward": bf, "backward": bb, "step": bf + bb},
        "achieved": round((bf + bb) / (t_s * 1e-3) / 1e9, 1), "frac": round((bf + bb) / (t_s * 1e-3) / 1e9 / 8000.0, 4),
        "training_forward_frac": round(bf / (t_f * 1e-3) / 1e9 / 8000.0, 4),
        "backward_frac": round(bb / ((t_s - t_f) * 1e-3) / 1e9 / 8000.0, 4),
        "backward_bytes_by_kernel": bwd_b, "training_forward_bytes_by_kernel": fwd_b,
        "note": "per-kernel times: rocprofv3 kernel statistics of this command (tools/profile_train.sh -> profiles/<tag>_train_kernel_stats.txt)"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
