#!/usr/bin/env python
"""Kernel timeline of one graph-plan build on the c3 batch (radix build and sorted hand-off), under rocprofv3 --kernel-trace."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    import torch
    from pyhgt_amd import GraphPlan
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout, to_device_graph
    dev = "cuda:0"
    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    x, nt, tm, ei, et, _, ed = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
    T, R = 4, len(ed)
    dg = to_device_graph(*batch, device=dev)
    src32, dst32, time32 = dg[3][0].int().contiguous(), dg[3][1].int().contiguous(), dg[2].int().contiguous()
    rel_ptr = torch.searchsorted(dg[4], torch.arange(R + 1, device=dev)).int()
    type_off = torch.searchsorted(dg[1], torch.arange(T + 1, device=dev)).int()
    big = torch.empty(64 << 20, device=dev)
    for it in range(20):
        big.add_(1.0)
        GraphPlan(nt, ei, et, tm, T, R)
        big.add_(1.0)
        GraphPlan.from_sorted(dg[1], dg[3], dg[4], dg[2], src32, dst32, time32, rel_ptr, type_off, T, R)
    torch.cuda.synchronize()


def show(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last iteration: from the second-to-last big add_ on
    idx = [i for i, r in enumerate(rows) if "CUDAFunctorOnSelf_add" in r["Kernel_Name"]]
    seg = rows[idx[-2]:]
    prev = None
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%8.2f us  gap %7.2f  %s" % ((e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:110]))
        prev = e


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
