#!/usr/bin/env python
"""Kernel timeline of ONE steady-state c3 layer forward (run under rocprofv3 --kernel-trace; the analysis half reads its CSV):
   python tools/lab/trace_small.py run        -> 60 forwards of the c3 surrogate layer
   python tools/lab/trace_small.py show DIR   -> durations and gaps of the kernels of the last forward"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    import torch
    from pyhgt_amd import HGTConv, GraphPlan
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout
    dev = "cuda:0"
    which = os.environ.get("HGT_TRACE_WORKLOAD", "c3")
    if which == "c3":
        batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
        d = 256
    else:      # the scripts' default batch (sample_width 520)
        batch = synthetic_sampled_batch("mag", n_seed=128, width=520, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
        d = 256
    x, nt, tm, ei, et, _, edge_dict = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    T, R, H = 4, len(edge_dict), 8
    layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=os.environ.get("HGT_TRACE_PREC", "bf16x3")).eval().to(dev)
    plan = GraphPlan(nt, ei, et, tm, T, R)
    print("N", nt.numel(), "E", et.numel())
    with torch.no_grad():
        for _ in range(60):
            layer(x, nt, ei, et, tm, plan=plan)
    torch.cuda.synchronize()


def show(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the layer's first kernel = the typed linear of Q|K|V; take the last complete forward
    starts = [i for i, n in enumerate(names) if "k_typed_linear" in n and ("<0, false" in n or "ILi0ELb0" in n)]
    # forwards begin at every other typed linear at most; find the period from the tail
    tail = rows[-40:]
    t0 = None
    prev_end = None
    out = []
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        out.append("%8.2f us  gap %6.2f  %s" % ((e - s) / 1e3, gap, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:110]))
        prev_end = e
    print("\n".join(out))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
