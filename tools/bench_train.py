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


def minimal_backward_bytes(N, E, d, H):
    """A MINIMAL-traffic model of the backward (round-5 review: the figure above counts the kernels as built -- seven E x d gather passes).
    What the arithmetic needs at least: ONE walk in target order that gathers K_j and V_j per edge (d att = <d agg' M^T, v>, the softmax
    backward in registers, dQ_i summed in place -- Q_i and d agg_i are read once per target -- and d relation_att / d relation_msg as
    outer products of rows already in flight), which leaves d s [E, H] behind, and ONE walk in source order that gathers the target-side
    rows q~_i and d agg'_i per edge to sum dK_j and dV_j: four gather passes, two per-edge-per-head arrays written and read once, every
    node-level array read or written once per GEMM that needs it."""
    Nd, EH, Eg = N * 4 * d, E * H * 4, E * (4 * d + 12)
    return {"node_update_bwd + a_linear bwd (read grad_out, trans, x, agg; write d agg, dx_skip; W_a gradient reads agg, d_trans)": 9 * Nd,
            "target-order walk (gather K, V; read Q, d agg; write dQ, d s)": 2 * Eg + 3 * Nd + 2 * EH,
            "source-order walk (gather q~, d agg'; read d s, att; write dK, dV)": 2 * Eg + 2 * Nd + 2 * EH,
            "Q|K|V weight gradients + dx (read x, dQ|dK|dV; write dx)": 8 * Nd}


def sampled_batches(dev):
    """Training step (forward + backward, dropout on) at the sizes the reference's scripts actually step on (round-5 review, task 8):
    the c3 surrogate (one layer, d = 256) and the c5 / published ogbn-mag surrogates (whole GNN), wall-clock us per step."""
    from pyhgt_amd import GNN
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout
    res = {}

    def step_us(fn, params, iters=30):
        for it in range(5 + iters):
            if it == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            out = fn()
            out.backward(torch.ones_like(out))
            for p_ in params:
                p_.grad = None
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6

    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    x, nt, tm, ei, et, _, ed = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    layer = HGTConv(256, 256, 4, len(ed), 8, 0.2, True, True).to(dev).train()
    plan = GraphPlan(nt, ei, et, tm, 4, len(ed))
    xg = x.clone().requires_grad_(True)
    res["c3_layer"] = {"N": int(nt.numel()), "E": int(et.numel()), "d": 256,
                       "fwd_bwd_us": step_us(lambda: layer(xg, nt, ei, et, tm, plan=plan), list(layer.parameters()) + [xg])}
    for key, c in (("c5_gnn2", dict(schema="oag", n_seed=256, width=128, depth=6, feat_dim=1169, mean_degree=1.2, seed=5, in_dim=1169, n_hid=400,
                                    T=5, H=8, L=2, norm=False)),
                   ("mag4_gnn4", dict(schema="mag", n_seed=128, width=128, depth=6, feat_dim=129, mean_degree=4.0, seed=3, in_dim=129, n_hid=512,
                                      T=4, H=8, L=4, norm=True))):
        batch = synthetic_sampled_batch(c["schema"], n_seed=c["n_seed"], width=c["width"], depth=c["depth"], feat_dim=c["feat_dim"],
                                        mean_degree=c["mean_degree"], seed=c["seed"])
        x, nt, tm, ei, et, _, ed = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
        gnn = GNN(c["in_dim"], c["n_hid"], c["T"], len(ed), c["H"], c["L"], 0.2, "hgt", c["norm"], c["norm"], True).to(dev).train()
        res[key] = {"N": int(nt.numel()), "E": int(et.numel()), "n_hid": c["n_hid"], "layers": c["L"],
                    "fwd_bwd_us": step_us(lambda: gnn(x, nt, tm, ei, et), list(gnn.parameters()))}
    return res


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
    mb = minimal_backward_bytes(N, E, d, H)
    res["roofline"] = {
        "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
        "algorithmic_bytes": {"training_forward": bf, "backward": bb, "step": bf + bb},
        # against a minimal-traffic backward (four gather passes) instead of the kernels as built (seven): the stricter figure
        "minimal_backward_bytes": sum(mb.values()), "minimal_backward_bytes_by_part": mb,
        "backward_frac_of_minimal_model": round(sum(mb.values()) / ((res["training_forward_backward_ms"] - res["training_forward_ms"]) * 1e-3) / 1e9 / 8000.0, 4),
        "achieved": round((bf + bb) / (t_s * 1e-3) / 1e9, 1), "frac": round((bf + bb) / (t_s * 1e-3) / 1e9 / 8000.0, 4),
        "training_forward_frac": round(bf / (t_f * 1e-3) / 1e9 / 8000.0, 4),
        "backward_frac": round(bb / ((t_s - t_f) * 1e-3) / 1e9 / 8000.0, 4),
        "backward_bytes_by_kernel": bwd_b, "training_forward_bytes_by_kernel": fwd_b,
        "note": "per-kernel times: rocprofv3 kernel statistics of this command (tools/profile_train.sh -> profiles/<tag>_train_kernel_stats.txt)"}
    if not os.environ.get("HGT_TRAIN_NO_SMALL"):
        del x, xg, g, layer, plan
        torch.cuda.empty_cache()
        res["sampled_batches"] = sampled_batches(dev)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
