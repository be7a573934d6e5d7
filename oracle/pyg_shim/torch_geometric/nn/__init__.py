"""torch_geometric.nn stand-in: MessagePassing + placeholder baselines.

conv.py:5 does `from torch_geometric.nn import GCNConv, GATConv`; those two
baselines are third-party layers outside the HGT hot path, so they are
placeholders that refuse to run.
"""
import torch.nn as _nn
from .conv import MessagePassing  # noqa: F401


class _OutOfScopeBaseline(_nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            "GCNConv/GATConv are PyG baselines, not part of the HGTConv hot path")


class GCNConv(_OutOfScopeBaseline):
    pass


class GATConv(_OutOfScopeBaseline):
    pass
