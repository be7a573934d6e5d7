"""torch_geometric.utils stand-in: segment softmax (conv.py:8, used at conv.py:108).

PyG 1.3.x definition: num_nodes = index.max()+1;
out = exp(src - scatter_max(src, index)[index]);
out = out / (scatter_add(out, index)[index] + 1e-16).
"""
import torch


def softmax(src, index, num_nodes=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    seg_max = torch.full(shape, float("-inf"), dtype=src.dtype, device=src.device)
    seg_max.scatter_reduce_(0, idx, src, reduce="amax", include_self=True)
    out = (src - seg_max.index_select(0, index)).exp()
    seg_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).index_add_(0, index, out)
    return out / (seg_sum.index_select(0, index) + 1e-16)
