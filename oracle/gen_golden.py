"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE.  The reference has no golden vectors of its own for
HGTConv, so known answers are produced here by importing
/root/reference/pyHGT/conv.py verbatim (oracle/reference_loader.py), loading a
seeded state_dict into the reference HGTConv, running forward in eval mode on
seeded synthetic graphs and storing inputs, parameters, `out` and `.att`.

    python oracle/gen_golden.py        # rewrites tests/golden/*.npz

The fixtures travel to the GPU box; /root/reference does not.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import hgt_oracle as O                       # noqa: E402
from oracle.reference_loader import load_reference_conv  # noqa: E402
from pyhgt_amd.synth import synthetic_typed_graph        # noqa: E402

# name -> (N, E, d, H, T, R, use_norm, use_RTE, graph kwargs)
CASES = {
    # BASELINE.json configs[0] at full size: T3 R4 N10k E50k d64 H4
    "c1_full": dict(N=10000, E=50000, d=64, H=4, T=3, R=4, use_norm=True, use_RTE=True, gk={}),
    # small cases covering SURVEY appendix D
    "small_rte_norm": dict(N=600, E=3000, d=64, H=4, T=3, R=4, use_norm=True, use_RTE=True, gk={}),
    "small_plain": dict(N=300, E=1500, d=32, H=2, T=2, R=3, use_norm=False, use_RTE=False, gk={}),
    "dk50": dict(N=200, E=1200, d=100, H=2, T=2, R=3, use_norm=True, use_RTE=True, gk={}),
    "d256_h8": dict(N=400, E=4000, d=256, H=8, T=2, R=8, use_norm=True, use_RTE=False, gk={}),
    "hubs_unsorted": dict(N=500, E=6000, d=64, H=8, T=4, R=5, use_norm=True, use_RTE=True,
                          gk=dict(sorted_types=False, dst_skew=1.2, strided_edge_index=False)),
    "isolated_empty_type": dict(N=800, E=400, d=32, H=4, T=4, R=2, use_norm=True, use_RTE=False,
                                gk={}, drop_type=2),
    "schema": dict(N=700, E=5000, d=64, H=4, T=4, R=9, use_norm=True, use_RTE=True, gk=dict(schema=True)),
    # DenseHGTConv (conv.py:143-280): same message(), dense update
    "dense_rte_norm": dict(N=500, E=4000, d=64, H=4, T=3, R=4, use_norm=True, use_RTE=True, gk={}, dense=True),
    "dense_plain_d256": dict(N=300, E=2500, d=256, H=8, T=2, R=3, use_norm=False, use_RTE=False, gk={}, dense=True),
}


# fixed per-case seeds (the first eight are "100 + index in sorted order" of the original case list)
SEEDS = {"c1_full": 100, "d256_h8": 101, "dk50": 102, "hubs_unsorted": 103, "isolated_empty_type": 104, "schema": 105,
         "small_plain": 106, "small_rte_norm": 107, "dense_rte_norm": 108, "dense_plain_d256": 109}


def build_case(name, c, seed):
    x, nt, ei, et, tm = synthetic_typed_graph(c["N"], c["E"], c["d"], c["T"], c["R"], seed=seed, **c["gk"])
    if "drop_type" in c:  # a node type with zero nodes (conv.py:83,123 skip paths)
        nt = torch.where(nt == c["drop_type"], torch.full_like(nt, c["drop_type"] + 1), nt)
    sd = O.make_state_dict(c["d"], c["d"], c["T"], c["R"], c["H"], c["use_norm"], c["use_RTE"], seed=seed + 1000,
                           dense=c.get("dense", False))
    return sd, x, nt, ei, et, tm


def main():
    conv = load_reference_conv()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = set(sys.argv[1:])          # optional: regenerate just the named cases
    for name, c in sorted(CASES.items()):
        if only and name not in only:
            continue
        sd, x, nt, ei, et, tm = build_case(name, c, seed=SEEDS[name])
        cls = conv.DenseHGTConv if c.get("dense", False) else conv.HGTConv
        layer = cls(c["d"], c["d"], c["T"], c["R"], c["H"], 0.2, c["use_norm"], c["use_RTE"]).eval()
        layer.load_state_dict(sd)
        with torch.no_grad():
            out = layer(x, nt, ei, et, tm)
            att = layer.att
        blob = {"param::" + kk: vv.numpy() for kk, vv in sd.items()}
        blob.update(
            node_feature=x.numpy(), node_type=nt.numpy().astype(np.int32),
            edge_index=ei.contiguous().numpy().astype(np.int32), edge_type=et.numpy().astype(np.int16),
            edge_time=tm.numpy().astype(np.int16), out=out.numpy(), att=att.numpy(),
            meta=np.array([c["N"], c["E"], c["d"], c["H"], c["T"], c["R"], int(c["use_norm"]), int(c["use_RTE"]),
                           int(c.get("dense", False))], dtype=np.int64),
            strided=np.array([int(c["gk"].get("strided_edge_index", True))]),
        )
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **blob)
        print("%-22s N=%d E=%d d=%d H=%d  |out|max=%.3f  %.1f KB" % (
            name, c["N"], c["E"], c["d"], c["H"], float(out.abs().max()), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
