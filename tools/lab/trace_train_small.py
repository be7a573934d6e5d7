#!/usr/bin/env python
"""Kernel timeline of ONE training step (forward + backward) of the c3 surrogate layer, under rocprofv3 --kernel-trace:
   python tools/lab/trace_train_small.py run      python tools/lab/trace_train_small.py show DIR"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    import torch
    from pyhgt_amd import HGTConv, GraphPlan
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout
    dev = "cuda:0"
    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    x, nt, tm, ei, et, _, ed = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    layer = HGTConv(256, 256, 4, len(ed), 8, 0.2, True, True).to(dev).train()
    plan = GraphPlan(nt, ei, et, tm, 4, len(ed))
    xg = x.clone().requires_grad_(True)
    for _ in range(30):
        out = layer(xg, nt, ei, et, tm, plan=plan)
        out.backward(torch.ones_like(out))
        for p in list(layer.parameters()) + [xg]:
            p.grad = None
    torch.cuda.synchronize()


def show(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    period = None
    for p in range(5, 400):
        span = max(2 * p, 200)
        if len(names) >= span + p and all(names[-i] == names[-i - p] for i in range(1, span + 1)):
            period = p
            break
    print("kernels per step:", period)
    step = rows[-period:]
    tot = 0.0
    prev = None
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot += (e - s) / 1e3
        print("%8.2f us  gap %7.2f  %s" % ((e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]))
        prev = e
    print("sum of kernel durations %.1f us; first start -> last end %.1f us" % (tot, (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
