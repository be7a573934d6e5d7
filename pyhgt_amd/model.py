"""Host-side mirror of the reference's GNN wrapper ("next" row 1).

`GNN` keeps the constructor, attribute / state_dict names and forward signature of
/root/reference/pyHGT/model.py:54-80: a typed input adapter (`adapt_ws[t]` Linear + tanh, model.py:70-76)
followed by `n_layers` GeneralConv('hgt') layers that all receive the same graph tensors (model.py:78-79).
The adapter runs on the same typed-linear kernels as the layers (no per-type boolean masks, no host sync --
the reference syncs once per type at model.py:73), and one GraphPlan is built for the whole stack.
Inference runs without autograd; with grad enabled the adapter and the layers take their differentiable paths
(pyhgt_amd/autograd.py).  Classifier / Matcher (model.py:3-49) run on the same kernels.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from .conv import GeneralConv, GraphPlan, _ptr, _stream

__all__ = ["GNN", "Classifier", "Matcher"]


class GNN(nn.Module):
    def __init__(self, in_dim, n_hid, num_types, num_relations, n_heads, n_layers, dropout=0.2, conv_name='hgt',
                 prev_norm=False, last_norm=False, use_RTE=True):
        super().__init__()
        self.gcs = nn.ModuleList()
        self.num_types = num_types
        self.in_dim = in_dim
        self.n_hid = n_hid
        self.adapt_ws = nn.ModuleList(nn.Linear(in_dim, n_hid) for _ in range(num_types))
        self.drop = nn.Dropout(dropout)
        for _ in range(n_layers - 1):
            self.gcs.append(GeneralConv(conv_name, n_hid, n_hid, num_types, num_relations, n_heads, dropout,
                                        use_norm=prev_norm, use_RTE=use_RTE))
        self.gcs.append(GeneralConv(conv_name, n_hid, n_hid, num_types, num_relations, n_heads, dropout,
                                    use_norm=last_norm, use_RTE=use_RTE))
        self._packed = None
        self._packed_key = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_packed"] = st["_packed_key"] = None
        st.pop("_plist", None)
        st.pop("_tiles", None)
        st.pop("_tiles_key", None)
        return st

    def __setstate__(self, state):
        """Modules pickled by the reference class (torch.save(model), train_paper_field.py:279) lack the cache attributes."""
        super().__setstate__(state)
        self.__dict__.setdefault("_packed", None)
        self.__dict__.setdefault("_packed_key", None)

    def invalidate(self):
        """Forget the packed adapter weights and every layer's packed / prepared images (needed after writes through
        `.data`, which do not bump the parameter versions the caches are keyed on; see HGTConv.invalidate)."""
        self._packed = self._packed_key = None
        self.__dict__.pop("_tiles_key", None)
        self.__dict__.pop("_plist", None)
        for gc in self.gcs:
            if hasattr(gc.base_conv, "invalidate"):
                gc.base_conv.invalidate()

    def _load_from_state_dict(self, *args, **kwargs):
        self._packed = self._packed_key = None
        self.__dict__.pop("_plist", None)
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._packed = self._packed_key = None
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, *args, **kwargs)

    def _pack_adapter(self):
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = [p for lin in self.adapt_ws for p in lin.parameters()]
        key = 0                                  # sum of version counters (see HGTConv._pack_parameters)
        for p in plist:
            key += p._version
        if self._packed is None or self._packed_key != key or self.training:
            with torch.no_grad():
                w = torch.stack([lin.weight for lin in self.adapt_ws]).float().contiguous()   # [T, n_hid, in_dim]
                b = torch.stack([lin.bias for lin in self.adapt_ws]).float().contiguous()     # [T, n_hid]
            self._packed, self._packed_key = (w, b), key
        return self._packed

    def _adapter_tiles(self, w, in_dim, n_hid, st, f16=False):
        """Split MFMA tiles of the adapter weights (hgt_split_weights[_f16]), kept until the weights change."""
        key = (w.data_ptr(), self._packed_key, f16)
        if getattr(self, "_tiles_key", None) != key:
            lib = _lib.load()
            nb = C.c_uint64()
            _lib.check(lib.hgt_split_weights_bytes(self.num_types, in_dim, n_hid, C.byref(nb)), "hgt_split_weights_bytes")
            self._tiles = torch.empty(int(nb.value), dtype=torch.uint8, device=w.device)
            _lib.check((lib.hgt_split_weights_f16 if f16 else lib.hgt_split_weights)(_ptr(w), n_hid * in_dim, self.num_types, in_dim, n_hid,
                                                                                      _ptr(self._tiles), st), "hgt_split_weights(adapter)")
            self._tiles_key = key
        return self._tiles

    def forward(self, node_feature, node_type, edge_time, edge_index, edge_type):
        """Same argument order as the reference (model.py:69): note edge_time comes third."""
        lib = _lib.load()
        if not node_feature.is_cuda:
            raise RuntimeError("pyhgt_amd.GNN runs only on a ROCm GPU tensor; there is no CPU fallback")
        conv0 = self.gcs[0].base_conv
        plan = GraphPlan.cached(node_type, edge_index, edge_type, edge_time if conv0.use_RTE else None,
                                conv0.num_types, conv0.num_relations)
        rows = plan.row_lists()
        if torch.is_grad_enabled() and (node_feature.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training / differentiable path: the adapter through TypedLinearFunction (typed weight
            # gradient kernels), tanh + dropout as in model.py:70-76, then the layers' own autograd path
            from .autograd import TypedLinearFunction
            w = torch.stack([lin.weight for lin in self.adapt_ws]).float()
            b = torch.stack([lin.bias for lin in self.adapt_ws]).float()
            h = TypedLinearFunction.apply((rows.rows_all, rows.off_all, plan), self.num_types, conv0.precision,
                                          node_feature.float(), w, b)
            h = self.drop(torch.tanh(h))        # rows of a type no adapter claims stay 0 (model.py:70)
            for gc in self.gcs:
                h = gc.base_conv(h, node_type, edge_index, edge_type, edge_time, plan=plan)
            return h
        x = node_feature.detach().float().contiguous()
        N = x.size(0)
        w, b = self._pack_adapter()
        T, n_hid, in_dim = self.num_types, self.n_hid, self.in_dim
        h = torch.empty(N, n_hid, dtype=torch.float32, device=x.device)
        st = _stream()
        # typed adapter (in_dim is arbitrary, e.g. 129 or 1169): the precision of the layers -- split-bf16 x3 MFMA (its row loader
        # takes any K) or the exact fp32 MFMA kernel
        need_tanh = True
        if conv0.precision in ("bf16x3", "f16x3") and n_hid % 4 == 0:
            f16 = conv0.precision == "f16x3"
            tiles = self._adapter_tiles(w, in_dim, n_hid, st, f16)
            fn = lib.hgt_typed_linear_f16x3 if f16 else lib.hgt_typed_linear_bf16x3
            # adapter + tanh in one kernel where the kernel that takes the shape has the activation epilogue (sampled batches;
            # HGT_ERR_UNSUPPORTED = nothing was launched)
            # (asked for on sampled batches only: with the bit set a large input never takes the x-stationary kernel, which has no
            #  activation epilogue)
            rc = -2 if N >= 65536 else fn(_ptr(x), in_dim, rows.rows_all, rows.off_all, T, N, in_dim, n_hid, _ptr(tiles), _ptr(b), n_hid,
                                          _ptr(h), 0, 0, n_hid, 0, _lib.HGT_LINEAR_TANH, st)
            if rc == 0:
                need_tanh = False
            else:
                _lib.check(fn(_ptr(x), in_dim, rows.rows_all, rows.off_all, T, N, in_dim, n_hid, _ptr(tiles), _ptr(b), n_hid, _ptr(h), 0, 0, n_hid,
                              0, 0, st), "hgt_typed_linear_bf16x3(adapter)")
        else:
            _lib.check(lib.hgt_typed_linear(_ptr(x), in_dim, rows.rows_all, rows.off_all, T, N, in_dim, n_hid, _ptr(w), n_hid * in_dim,
                                            _ptr(b), n_hid, _ptr(h), 0, 0, n_hid, 0, 0, 0, st), "hgt_typed_linear(adapter)")
        # nodes whose type no adapter claims stay zero like the reference's zero-initialised `res` (model.py:70);
        # rows_all[off_all[T] .. off_all[T+1]) are exactly those nodes
        if not (plan.NQ == plan.N and plan.no_unknown_rows):      # (skipped once the plan header says every row has a valid type)
            _lib.check(lib.hgt_zero_rows(rows.rows_all, rows.off_all + 4 * T, n_hid, _ptr(h), st), "hgt_zero_rows")
        if need_tanh:
            _lib.check(lib.hgt_tanh_inplace(_ptr(h), N * n_hid, st), "hgt_tanh_inplace")
        for gc in self.gcs:
            h = gc.base_conv(h, node_type, edge_index, edge_type, edge_time, plan=plan)
        return h


def _dense_linear(x, weight, bias, scale=1.0):
    """y = (x @ weight^T + bias) * scale on the exact fp32 MFMA typed-linear kernel (one group = every row)."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("pyhgt_amd heads run only on a ROCm GPU tensor; there is no CPU fallback")
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        # differentiable path: same kernel forward, typed weight-gradient kernels backward (pyhgt_amd/autograd.py)
        from .autograd import TypedLinearFunction
        xx = x.float().contiguous()
        n = xx.size(0)
        rows = torch.arange(n, dtype=torch.int32, device=x.device)
        off = torch.tensor([0, n], dtype=torch.int32, device=x.device)
        w = (weight.float() * scale).unsqueeze(0)
        b = (bias.float() * scale).unsqueeze(0) if bias is not None else None
        return TypedLinearFunction.apply((rows.data_ptr(), off.data_ptr(), (rows, off)), 1, "fp32", xx, w, b)
    x = x.detach().float().contiguous()
    n, k = x.shape
    n_out = weight.size(0)
    w = (weight.detach().float() * scale).contiguous()
    b = (bias.detach().float() * scale).contiguous() if bias is not None else None
    rows = torch.arange(n, dtype=torch.int32, device=x.device)
    off = torch.tensor([0, n], dtype=torch.int32, device=x.device)
    y = torch.empty(n, n_out, dtype=torch.float32, device=x.device)
    _lib.check(lib.hgt_typed_linear(_ptr(x), k, _ptr(rows), _ptr(off), 1, n, k, n_out, _ptr(w), 0, _ptr(b), 0, _ptr(y), 0, 0, n_out,
                                    0, 0, 0, _stream()), "hgt_typed_linear(head)")
    return y


class Classifier(nn.Module):
    """model.py:3-14: log_softmax(linear(x).squeeze(), dim=-1) on the seed rows."""

    def __init__(self, n_hid, n_out):
        super().__init__()
        self.n_hid, self.n_out = n_hid, n_out
        self.linear = nn.Linear(n_hid, n_out)

    def forward(self, x):
        tx = _dense_linear(x.reshape(-1, self.n_hid), self.linear.weight, self.linear.bias)
        if tx.requires_grad:
            return torch.log_softmax(tx, dim=-1).reshape(*x.shape[:-1], self.n_out).squeeze()      # model.py:11, under autograd
        out = torch.empty_like(tx)
        _lib.check(_lib.load().hgt_log_softmax_rows(_ptr(tx), tx.size(0), tx.size(1), _ptr(out), _stream()), "hgt_log_softmax_rows")
        return out.reshape(*x.shape[:-1], self.n_out).squeeze()

    def __repr__(self):
        return '{}(n_hid={}, n_out={})'.format(self.__class__.__name__, self.n_hid, self.n_out)


class Matcher(nn.Module):
    """model.py:16-49: scaled dot product between projected node pairs (link prediction), with the reference's
    inference-time cache of the projected candidates."""

    def __init__(self, n_hid):
        super().__init__()
        self.left_linear = nn.Linear(n_hid, n_hid)
        self.right_linear = nn.Linear(n_hid, n_hid)
        self.sqrt_hd = math.sqrt(n_hid)
        self.n_hid = n_hid
        self.cache = None

    def forward(self, x, y, infer=False, pair=False):
        ty = _dense_linear(y, self.right_linear.weight, self.right_linear.bias)
        if infer and self.cache is not None:
            tx = self.cache
        else:
            tx = _dense_linear(x, self.left_linear.weight, self.left_linear.bias)
            if infer:
                self.cache = tx
        if pair and (tx.requires_grad or ty.requires_grad):
            return (tx * ty).sum(dim=-1) / self.sqrt_hd                                            # model.py:41,44, under autograd
        if pair:
            out = torch.empty(tx.size(0), dtype=torch.float32, device=tx.device)
            _lib.check(_lib.load().hgt_row_dot(_ptr(tx), _ptr(ty), tx.size(0), self.n_hid, 1.0 / self.sqrt_hd, _ptr(out), _stream()),
                       "hgt_row_dot")
            return out
        # tx @ ty^T / sqrt(n_hid): the typed-linear kernel with ty as the "weight" and no bias
        return _dense_linear(tx, ty / self.sqrt_hd, None)

    def __repr__(self):
        return '{}(n_hid={})'.format(self.__class__.__name__, self.n_hid)
