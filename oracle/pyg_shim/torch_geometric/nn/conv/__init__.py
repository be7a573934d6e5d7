"""MessagePassing stand-in (flow source_to_target, aggr 'add').

Published PyG semantics that conv.py relies on (conv.py:13 ctor kwargs,
conv.py:57 propagate call, conv.py:60 message signature, conv.py:114 update
signature):
  * for every argument of message() named `<name>_j` the tensor kwargs[<name>]
    is gathered along node_dim with edge_index[0] (source j), `<name>_i` with
    edge_index[1] (target i); `edge_index_i` is edge_index[1]; every other
    argument name is passed through from the propagate kwargs;
  * message output [E, F] is scatter-ADDED by edge_index[1] into zeros [N, F],
    N = size of the node tensors along node_dim;
  * update(aggr_out, <named kwargs>) produces the layer output.
"""
import inspect

import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=0, **kwargs):
        super().__init__()
        if aggr != "add" or flow != "source_to_target" or node_dim != 0:
            raise NotImplementedError("shim covers only what pyHGT/conv.py uses")
        self._msg_args = [p for p in inspect.signature(self.message).parameters]
        self._upd_args = [p for p in inspect.signature(self.update).parameters][1:]

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        n_nodes = None
        feed = []
        for name in self._msg_args:
            if name == "edge_index_i":
                feed.append(dst)
            elif name == "edge_index_j":
                feed.append(src)
            elif name.endswith("_i") or name.endswith("_j"):
                base = kwargs[name[:-2]]
                n_nodes = base.size(0) if n_nodes is None else n_nodes
                feed.append(base.index_select(0, dst if name.endswith("_i") else src))
            else:
                feed.append(kwargs[name])
        msg = self.message(*feed)
        if n_nodes is None:
            n_nodes = int(dst.max()) + 1
        out = torch.zeros((n_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, dst, msg)
        return self.update(out, *[kwargs[a] for a in self._upd_args])

    def message(self, *args):  # pragma: no cover - overridden by HGTConv
        raise NotImplementedError

    def update(self, aggr_out):  # pragma: no cover - overridden by HGTConv
        return aggr_out
