"""Generate tests/golden/gnn_*.npz by running the REFERENCE's own `pyHGT.model.GNN` (build container only).

TEST INFRASTRUCTURE.  The wrapper `GNN.forward` (model.py:66-80) is pinned to outputs of the verbatim reference on
sampler-shaped batches (pyhgt_amd.sampled.synthetic_sampled_batch through the reference's wire format):

  gnn_oag2  BASELINE.json configs[4] shape: OAG schema (T=5, R=33), in_dim 1169 -> n_hid 400, 8 heads, 2 layers, the
            script's defaults (no LayerNorm, RTE on: OAG/train_paper_field.py:30-40,190-192)
  gnn_mag4  the published ogbn-mag model: T=4, R=9, in_dim 129 -> n_hid 512, 8 heads, 4 layers, prev_norm / last_norm / RTE
            (ogbn-mag/train_ogbn_mag.py:36-46,108-111; 21,173,389 parameters with its classifier, README.md:30)

The parameters are NOT stored (85 MB): `oracle.hgt_oracle.make_gnn_state_dict(seed)` rebuilds them on both sides, and the
batch is rebuilt from its seed.  Stored: 192 sampled rows of the adapter output and of EVERY layer's output (forward hooks
on the reference's GeneralConv modules), so that the tests can print the error growth per layer.

    python oracle/gen_golden_gnn.py        # rewrites tests/golden/gnn_*.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import hgt_oracle as O                        # noqa: E402
from oracle.reference_loader import load_reference_model  # noqa: E402
from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout  # noqa: E402

GNN_CASES = {
    "gnn_oag2": dict(schema="oag", n_seed=256, width=128, depth=6, feat_dim=1169, mean_degree=1.2, batch_seed=5,
                     in_dim=1169, n_hid=400, T=5, R=33, H=8, n_layers=2, prev_norm=False, last_norm=False, use_RTE=True, seed=11),
    "gnn_mag4": dict(schema="mag", n_seed=128, width=128, depth=6, feat_dim=129, mean_degree=4.0, batch_seed=3,
                     in_dim=129, n_hid=512, T=4, R=9, H=8, n_layers=4, prev_norm=True, last_norm=True, use_RTE=True, seed=12),
}
N_ROWS = 192


def build_batch(c):
    batch = synthetic_sampled_batch(c["schema"], n_seed=c["n_seed"], width=c["width"], depth=c["depth"], feat_dim=c["feat_dim"],
                                    mean_degree=c["mean_degree"], seed=c["batch_seed"])
    return batch, to_torch_layout(*batch)


def pick_rows(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(n, generator=g)[:N_ROWS].sort().values


def main():
    model = load_reference_model()
    outdir = os.path.join(ROOT, "tests", "golden")
    for name, c in sorted(GNN_CASES.items()):
        _, (x, nt, tm, ei, et, _, edge_dict) = build_batch(c)
        assert len(edge_dict) == c["R"]
        gnn = model.GNN(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], 0.2, "hgt", c["prev_norm"], c["last_norm"],
                        c["use_RTE"]).eval()
        sd = O.make_gnn_state_dict(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"], c["last_norm"],
                                   c["use_RTE"], seed=c["seed"])
        missing = gnn.load_state_dict(sd, strict=True)
        captured = []
        hooks = [gnn.gcs[0].register_forward_pre_hook(lambda m, a: captured.append(a[0].detach().clone()))]
        hooks += [gc.register_forward_hook(lambda m, a, o: captured.append(o.detach().clone())) for gc in gnn.gcs]
        with torch.no_grad():
            out = gnn(x, nt, tm, ei, et)
        for h in hooks:
            h.remove()
        assert len(captured) == c["n_layers"] + 1 and torch.equal(captured[-1], out)
        rows = pick_rows(x.size(0), c["seed"])
        # the fp64 restatement on the same inputs, for the record (the tests compare against the REFERENCE rows below)
        ref64, lay64 = O.gnn_forward(sd, c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"],
                                     c["last_norm"], c["use_RTE"], x, nt, tm, ei, et, return_layers=True)
        per_layer = [float((captured[i].double() - lay64[i]).abs().max()) for i in range(len(captured))]
        blob = dict(rows=rows.numpy().astype(np.int32), layers=np.stack([t[rows].numpy() for t in captured]),
                    n_nodes=np.array([x.size(0)]), n_edges=np.array([et.numel()]),
                    fp32_reference_vs_fp64_restatement=np.array(per_layer))
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **blob)
        print("%-9s N=%d E=%d  |out|max=%.3f  reference(fp32) vs fp64 restatement per layer: %s  %.0f KB" % (
            name, x.size(0), et.numel(), float(out.abs().max()), ["%.1e" % e for e in per_layer], os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
