#!/usr/bin/env python
"""End-to-end training loop in the shape of the reference's scripts (ogbn-mag/train_ogbn_mag.py:141-198,
OAG/train_paper_field.py:218-279): sampled batch -> to_device_graph (instead of to_torch + .to(device)) -> GNN -> Classifier
-> nll_loss -> backward -> optimizer step, on sampler-shaped synthetic batches (the datasets are not available offline).

    python examples/train_synthetic.py [--schema mag|oag] [--steps 30] [--conv hgt|dense_hgt]

Everything on the hot path runs on the HIP kernels of pyhgt_amd (forward and backward); torch supplies the optimizer, the loss
and the autograd boundary."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhgt_amd import GNN, Classifier  # noqa: E402
from pyhgt_amd.sampled import synthetic_sampled_batch, to_device_graph  # noqa: E402


def run(schema="mag", steps=30, conv="hgt", n_hid=128, n_heads=8, n_layers=2, n_classes=16, batch_size=128, lr=2e-3, seed=0,
        device="cuda:0", verbose=True):
    torch.manual_seed(seed)
    feat_dim = 129 if schema == "mag" else 256
    batches = []
    for b in range(4):      # a small pool of sampled batches, cycled like an epoch of pre-sampled jobs (train_ogbn_mag.py:82-104)
        fe, ti, el, graph = synthetic_sampled_batch(schema, n_seed=batch_size, width=64, depth=4, feat_dim=feat_dim, mean_degree=6.0,
                                                    seed=seed * 100 + b)
        dg = to_device_graph(fe, ti, el, graph, device=device)
        g = torch.Generator().manual_seed(b)
        proj = torch.randn(feat_dim, n_classes, generator=g)
        labels = (dg[0][:batch_size].cpu() @ proj).argmax(dim=1).to(device)     # a learnable synthetic task on the seed papers
        batches.append((dg, labels))
    T, R = len(batches[0][0][5]), len(batches[0][0][6])
    gnn = GNN(feat_dim, n_hid, T, R, n_heads, n_layers, dropout=0.2, conv_name=conv, prev_norm=True, last_norm=True, use_RTE=True).to(device)
    head = Classifier(n_hid, n_classes).to(device)
    opt = torch.optim.AdamW(list(gnn.parameters()) + list(head.parameters()), lr=lr)
    losses, t0 = [], time.perf_counter()
    for step in range(steps):
        (x, nt, tm, ei, et, _, _), y = batches[step % len(batches)]
        gnn.train(), head.train()
        rep = gnn(x, nt, tm, ei, et)
        loss = torch.nn.functional.nll_loss(head(rep[:batch_size]), y)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(gnn.parameters(), 0.5)
        opt.step()
        losses.append(loss.item())
        if verbose and (step % 5 == 0 or step == steps - 1):
            print("step %3d  loss %.4f" % (step, losses[-1]))
    torch.cuda.synchronize()
    if verbose:
        print("%.1f ms per training step (%d nodes, %d edges, %d layers, conv=%s)" % ((time.perf_counter() - t0) / steps * 1e3, nt.numel(),
                                                                                   et.numel(), n_layers, conv))
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--schema", default="mag", choices=["mag", "oag"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--conv", default="hgt", choices=["hgt", "dense_hgt"])
    a = ap.parse_args()
    run(a.schema, a.steps, a.conv)
