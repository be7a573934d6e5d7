"""torch_geometric.nn.inits stand-in (conv.py:7, used at conv.py:53-54)."""
import math


def uniform(size, tensor):
    bound = 1.0 / math.sqrt(size)
    if tensor is not None:
        tensor.data.uniform_(-bound, bound)


def glorot(tensor):
    # U(-a, a), a = sqrt(6 / (fan_in + fan_out)) over the two trailing dims
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)
