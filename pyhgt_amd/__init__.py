"""pyhgt_amd -- MI355X (gfx950) native forward pass of pyHGT's HGTConv.

Drop-in for the reference layers pyHGT/conv.py::HGTConv / DenseHGTConv (same constructor, parameter
names, forward signature) backed by hand-written HIP kernels behind a C ABI (include/hgt_hip.h).
"""
from .conv import HGTConv, DenseHGTConv, GeneralConv, RelTemporalEncoding, GraphPlan, install_into  # noqa: F401
from .model import GNN, Classifier, Matcher  # noqa: F401

__all__ = ["HGTConv", "DenseHGTConv", "GeneralConv", "RelTemporalEncoding", "GraphPlan", "install_into", "GNN", "Classifier", "Matcher"]
