import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # HGT_TEST_KERNEL_FLAGS=<HGT_FLAG_* bits>: the whole suite once more with a kernel forced onto every layer, e.g. 1024 = the
    # x-stationary split GEMM on the small graphs its size threshold keeps it off, 512 = the LDS-ring aggregation (tools/gpu.sh).
    # A switch of the TEST SUITE: the library itself reads no environment variable.
    extra = int(os.environ.get("HGT_TEST_KERNEL_FLAGS", "0"))
    if extra:
        from pyhgt_amd import HGTConv
        HGTConv.EXTRA_KERNEL_FLAGS = extra


def golden_names():
    """Single-layer fixtures (oracle/gen_golden.py); the gnn_* files of oracle/gen_golden_gnn.py have their own tests."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith("gnn_")]


def load_golden(name):
    """Fixture written by oracle/gen_golden.py from the verbatim reference."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = [int(v) for v in z["meta"]]
    N, E, d, H, T, R, use_norm, use_RTE = meta[:8]
    dense = bool(meta[8]) if len(meta) > 8 else False       # DenseHGTConv fixture (conv.py:143-280)
    sd = {k[len("param::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param::")}
    ei = torch.from_numpy(z["edge_index"].astype(np.int64))
    if int(z["strided"][0]):
        ei = ei.t().contiguous().t()      # the (1,2)-strided view data.py:254 delivers
    return dict(
        name=name, N=N, E=E, d=d, H=H, T=T, R=R, use_norm=bool(use_norm), use_RTE=bool(use_RTE), dense=dense, sd=sd,
        x=torch.from_numpy(z["node_feature"]), node_type=torch.from_numpy(z["node_type"].astype(np.int64)),
        edge_index=ei, edge_type=torch.from_numpy(z["edge_type"].astype(np.int64)),
        edge_time=torch.from_numpy(z["edge_time"].astype(np.int64)),
        out=torch.from_numpy(z["out"]), att=torch.from_numpy(z["att"]),
    )


@pytest.fixture(params=golden_names())
def golden(request):
    return load_golden(request.param)
