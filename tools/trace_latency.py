#!/usr/bin/env python
"""Kernel timelines of the latency-regime workloads of the judged line (bench.py: secondary.latency_regime), one steady-state forward each:
   python tools/trace_latency.py run  c3|c3w520|c5|mag4|c1 [precision]   -> 60 forwards (run under rocprofv3 --kernel-trace)
   python tools/trace_latency.py show DIR [n_layers]                     -> durations and gaps of the kernels of the LAST forward,
                                                                            the sum of the durations, and the span first start -> last end
c3 / c3w520: one HGTConv layer on the sampler-shaped ogbn-mag batch (sample_width 128 / 520); c5: the 2-layer OAG GNN (in 1169 -> 400,
33 relations); mag4: the published 4-layer n_hid = 512 ogbn-mag model; c1: BASELINE.json configs[0] (tests/golden/c1_full.npz).
tools/gpu.sh latency <tag> runs all of them and writes profiles-ready text files to gpurun_out/."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(which, prec):
    import numpy as np
    import torch
    from pyhgt_amd import HGTConv, GNN, GraphPlan
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout
    dev = "cuda:0"
    if which in ("c3", "c3w520"):
        batch = synthetic_sampled_batch("mag", n_seed=128, width=128 if which == "c3" else 520, depth=6, feat_dim=256, mean_degree=4.0,
                                        seed=3 if which == "c3" else 5)
        x, nt, tm, ei, et, _, edge_dict = [t.to(dev) if torch.is_tensor(t) else t for t in to_torch_layout(*batch)]
        T, R, H, d = 4, len(edge_dict), 8, 256
        layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=prec).eval().to(dev)
        plan = GraphPlan(nt, ei, et, tm, T, R)
        print("N", nt.numel(), "E", et.numel())
        fn = lambda: layer(x, nt, ei, et, tm, plan=plan)
    elif which == "c1":
        z = np.load(os.path.join(ROOT, "tests", "golden", "c1_full.npz"))
        N, E, d, H, T, R, use_norm, use_rte = [int(v) for v in z["meta"]][:8]
        x = torch.from_numpy(z["node_feature"]).to(dev)
        nt = torch.from_numpy(z["node_type"]).long().to(dev)
        ei = torch.from_numpy(z["edge_index"]).long().to(dev)
        et = torch.from_numpy(z["edge_type"]).long().to(dev)
        tm = torch.from_numpy(z["edge_time"]).long().to(dev)
        layer = HGTConv(d, d, T, R, H, 0.2, bool(use_norm), bool(use_rte), precision=prec).eval().to(dev)
        plan = GraphPlan(nt, ei, et, tm if use_rte else None, T, R)
        fn = lambda: layer(x, nt, ei, et, tm, plan=plan)
    else:
        # (the batches of tests/golden/gnn_oag2.npz / gnn_mag4.npz, random parameters: timing only)
        c = (dict(schema="oag", n_seed=256, width=128, depth=6, feat_dim=1169, mean_degree=1.2, batch_seed=5, in_dim=1169, n_hid=400, T=5,
                  R=33, H=8, n_layers=2, prev_norm=False, last_norm=False, use_RTE=True) if which == "c5" else
             dict(schema="mag", n_seed=128, width=128, depth=6, feat_dim=129, mean_degree=4.0, batch_seed=3, in_dim=129, n_hid=512, T=4,
                  R=9, H=8, n_layers=4, prev_norm=True, last_norm=True, use_RTE=True))
        batch = synthetic_sampled_batch(c["schema"], n_seed=c["n_seed"], width=c["width"], depth=c["depth"], feat_dim=c["feat_dim"],
                                        mean_degree=c["mean_degree"], seed=c["batch_seed"])
        xc, ntc, tmc, eic, etc_, _, _ = to_torch_layout(*batch)
        args = [t.to(dev) for t in (xc, ntc, tmc, eic, etc_)]
        gnn = GNN(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], 0.2, "hgt", c["prev_norm"], c["last_norm"],
                  c["use_RTE"]).eval().to(dev)
        for gc in gnn.gcs:
            gc.base_conv.precision = prec
        print("N", ntc.numel(), "E", etc_.numel(), "layers", c["n_layers"])
        fn = lambda: gnn(*args)
    with torch.no_grad():
        for _ in range(60):
            fn()
    torch.cuda.synchronize()


def show(d, n_layers=1):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # period of the steady state = one FORWARD (not one layer of a multi-layer model): the shortest p whose repetition explains the
    # last max(3 p, 150) kernel names
    period = None
    for p in range(1, 200):
        span = max(3 * p, 150)
        if len(names) >= span + p and all(names[-i] == names[-i - p] for i in range(1, span + 1)):
            period = p
            break
    if period is None:
        period = 40
    # mean over the last 20 forwards of every position of the period
    nf = min(20, len(rows) // period - 1)
    tail = rows[-period:]
    dur = [0.0] * period
    gap = [0.0] * period
    span = 0.0
    for k in range(nf):
        fw = rows[len(rows) - (k + 1) * period: len(rows) - k * period]
        prev_end = None
        for i, r in enumerate(fw):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            dur[i] += (e - s) / 1e3 / nf
            if prev_end is not None:
                gap[i] += (s - prev_end) / 1e3 / nf
            prev_end = e
        span += (int(fw[-1]["End_Timestamp"]) - int(fw[0]["Start_Timestamp"])) / 1e3 / nf
    print("kernels per forward: %d   (mean of the last %d forwards; durations and the gap to the previous kernel, us)" % (period, nf))
    for i, r in enumerate(tail):
        print("%8.2f us  gap %6.2f  %s" % (dur[i], gap[i], r["Kernel_Name"].replace("(anonymous namespace)::", "")[:120]))
    print("sum of kernel durations %.2f us; first start -> last end %.2f us; per layer %.2f / %.2f us" % (
        sum(dur), span, sum(dur) / n_layers, span / n_layers))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        if os.environ.get("HGT_TRACE_FLAGS"):      # A/B runs: HGT_FLAG_* bits OR-ed into every layer
            from pyhgt_amd import HGTConv
            HGTConv.EXTRA_KERNEL_FLAGS = int(os.environ["HGT_TRACE_FLAGS"])
        run(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "bf16x3")
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
