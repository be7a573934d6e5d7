"""Device-side hand-off of sampled sub-graphs.

The reference turns a sampled batch into tensors with `to_torch(feature, time, edge_list, graph)`
(/root/reference/pyHGT/data.py:212-256): Python lists of int64 [source, target] pairs appended edge by edge, then
`LongTensor(...).t()`.  What the sampler hands it is already almost what the GPU plan needs:

  * node ids are type-contiguous, types ascending (data.py:227-235);
  * `edge_list[target_type][source_type][relation]` is a list of [target_serial, source_serial] pairs with the targets in
    ascending order inside every such run (data.py:199-209), the `self` run of a type first (data.py:183-186);
  * edge_time = year(target) - year(source) + 120 (data.py:250).

`to_device_graph` is a sibling of `to_torch` with the SAME inputs and the same 7-tuple result (the model code does not
change), but it (a) builds the arrays with numpy instead of per-edge list appends, (b) orders the runs relation-major --
inside one relation id the concatenated runs are then target-sorted, because a relation's runs belong to ascending target
types -- and (c) hands the int32 form of exactly that order to `hgt_plan_from_sorted`, which builds the GraphPlan without
the radix sorts of `hgt_plan_build`, and registers the plan for the tensors it returns: the first layer's
`GraphPlan.cached(...)` lookup hits.

`synthetic_sampled_batch` builds sampler OUTPUT (feature / time / edge_list dictionaries + a graph-like object) for the
ogbn-mag and OAG schemas -- the datasets themselves are not available offline -- with the layout facts above, for the
latency-regime benchmarks (BASELINE.json configs[2] and configs[4]) and the parity tests.
"""
from collections import OrderedDict

import numpy as np
import torch

from .conv import GraphPlan

__all__ = ["to_device_graph", "to_torch_layout", "synthetic_sampled_batch", "SchemaGraph", "MAG_META", "OAG_META"]

# (target_type, source_type, relation) triples in get_meta_graph() order; `rev_` twins as data.py:61 adds them
MAG_META = [("paper", "paper", "PP_cite"), ("paper", "paper", "rev_PP_cite"), ("paper", "author", "AP_write"),
            ("author", "paper", "rev_AP_write"), ("paper", "field_of_study", "PF_in"), ("field_of_study", "paper", "rev_PF_in"),
            ("author", "institution", "AI_in"), ("institution", "author", "rev_AI_in")]
_OAG_FWD = [("paper", "paper", "PP_cite"), ("paper", "field", "PF_in_L0"), ("paper", "field", "PF_in_L1"), ("paper", "field", "PF_in_L2"),
            ("paper", "field", "PF_in_L3"), ("paper", "field", "PF_in_L4"), ("paper", "field", "PF_in_L5"), ("paper", "venue", "PV_Conference"),
            ("paper", "venue", "PV_Journal"), ("paper", "venue", "PV_Repository"), ("paper", "venue", "PV_Patent"),
            ("paper", "author", "AP_write_first"), ("paper", "author", "AP_write_last"), ("paper", "author", "AP_write_other"),
            ("field", "field", "FF_in"), ("author", "affiliation", "in")]
OAG_META = _OAG_FWD + [(s, t, "rev_" + r) for (t, s, r) in _OAG_FWD]


class SchemaGraph:
    """The two methods of the reference's `Graph` that to_torch uses (data.py:72-83): get_types(), get_meta_graph()."""

    def __init__(self, types, meta):
        self._types, self._meta = list(types), list(meta)

    def get_types(self):
        return list(self._types)

    def get_meta_graph(self):
        return list(self._meta)


def synthetic_sampled_batch(schema="mag", n_seed=128, width=128, depth=6, feat_dim=None, mean_degree=6.0, seed=0):
    """Sampler-shaped output for a MAG-like (T=4, R=9 incl. `self`) or OAG-like (T=5, R=33) random graph:
    returns (feature, time, edge_list, graph) exactly as `sample_subgraph` would hand them to `to_torch`
    (data.py:86-210): per-type node budgets of n_seed + depth * width for the seed type and depth * width for the others,
    dict insertion order = seed type first, `self` run first, targets ascending inside a run, years in [2010, 2019]."""
    rng = np.random.default_rng(seed)
    if schema == "mag":
        types, meta, seed_type = ["paper", "author", "field_of_study", "institution"], MAG_META, "paper"
        feat_dim = feat_dim or 129
    elif schema == "oag":
        types, meta, seed_type = ["paper", "author", "field", "venue", "affiliation"], OAG_META, "paper"
        feat_dim = feat_dim or 1169
    else:
        raise ValueError("schema must be 'mag' or 'oag'")
    graph = SchemaGraph(types, meta)
    counts = {t: depth * width + (n_seed if t == seed_type else 0) for t in types}
    feature = {t: rng.standard_normal((counts[t], feat_dim)).astype(np.float32) for t in types}
    time = {t: rng.integers(2010, 2020, size=counts[t]) for t in types}
    order = [seed_type] + [t for t in types if t != seed_type]            # layer_data insertion order: seeds first
    edge_list = OrderedDict()
    for t in order:
        edge_list[t] = OrderedDict()
        edge_list[t][t] = OrderedDict()
        edge_list[t][t]["self"] = [[i, i] for i in range(counts[t])]      # data.py:183-186
    for (tt, st, rel) in meta:
        n_t, n_s = counts[tt], counts[st]
        deg = rng.poisson(mean_degree * n_s / max(1, sum(counts.values())) * len(types), size=n_t)
        deg = np.minimum(deg, n_s)
        pairs = []
        for ti in np.nonzero(deg)[0]:                                     # targets ascending (data.py:199-209)
            for si in rng.choice(n_s, size=deg[ti], replace=False):
                pairs.append([int(ti), int(si)])
        if pairs:
            edge_list[tt].setdefault(st, OrderedDict())[rel] = pairs
    return feature, time, edge_list, graph


def _runs(edge_list, node_off, edge_dict):
    """-> list of (relation id, target type offset, [n,2] int array of [target, source] GLOBAL ids) in dict order."""
    out = []
    for tt in edge_list:
        for st in edge_list[tt]:
            for rel in edge_list[tt][st]:
                pairs = np.asarray(edge_list[tt][st][rel], dtype=np.int64).reshape(-1, 2)
                if pairs.shape[0] == 0:
                    continue
                g = np.stack([pairs[:, 0] + node_off[tt], pairs[:, 1] + node_off[st]], axis=1)
                out.append((edge_dict[rel], node_off[tt], g))
    return out


def _node_arrays(feature, time, graph):
    types = graph.get_types()
    node_dict, n = {}, 0
    for t in types:
        node_dict[t] = [n, len(node_dict)]
        n += len(feature[t])
    edge_dict = {e[2]: i for i, e in enumerate(graph.get_meta_graph())}
    edge_dict["self"] = len(edge_dict)
    feat = np.concatenate([np.asarray(feature[t], dtype=np.float32).reshape(len(feature[t]), -1) for t in types], axis=0)
    ntime = np.concatenate([np.asarray(time[t], dtype=np.int64).reshape(-1) for t in types])
    ntype = np.concatenate([np.full(len(feature[t]), node_dict[t][1], dtype=np.int64) for t in types])
    type_off = np.array([node_dict[t][0] for t in types] + [n], dtype=np.int32)
    return types, node_dict, edge_dict, feat, ntime, ntype, type_off


def to_torch_layout(feature, time, edge_list, graph):
    """The tensors `to_torch` returns (same order of nodes AND edges, data.py:212-256), built with numpy -- used to check
    that the synthetic batches and `to_device_graph` agree with the reference's wire format."""
    types, node_dict, edge_dict, feat, ntime, ntype, _ = _node_arrays(feature, time, graph)
    node_off = {t: node_dict[t][0] for t in types}
    runs = _runs(edge_list, node_off, edge_dict)
    tgt = np.concatenate([g[:, 0] for _, _, g in runs]) if runs else np.zeros(0, np.int64)
    src = np.concatenate([g[:, 1] for _, _, g in runs]) if runs else np.zeros(0, np.int64)
    et = np.concatenate([np.full(len(g), r, dtype=np.int64) for r, _, g in runs]) if runs else np.zeros(0, np.int64)
    etime = ntime[tgt] - ntime[src] + 120
    ei = torch.from_numpy(np.stack([src, tgt], axis=1)).t()                # [2, E] view with strides (1, 2), like data.py:254
    return (torch.from_numpy(feat), torch.from_numpy(ntype), torch.from_numpy(etime), ei, torch.from_numpy(et), node_dict, edge_dict)


def to_device_graph(feature, time, edge_list, graph, device="cuda"):
    """Sibling of `to_torch` (data.py:212-256): same arguments, same 7-tuple (node_feature, node_type, edge_time, edge_index,
    edge_type, node_dict, edge_dict) -- tensors already on `device` -- plus the GraphPlan, built from the sorted int32 form
    and registered for those tensors.  Edges are ordered relation-major (stable inside a relation); HGTConv's output does not
    depend on the edge order.  Returns the 7-tuple; the plan is `GraphPlan.cached(...)` away (or `.plan` of the result)."""
    types, node_dict, edge_dict, feat, ntime, ntype, type_off = _node_arrays(feature, time, graph)
    node_off = {t: node_dict[t][0] for t in types}
    R = len(edge_dict)
    runs = _runs(edge_list, node_off, edge_dict)
    # relation-major; a relation's runs in ascending order of their target type offset -> targets non-decreasing per relation
    runs.sort(key=lambda r: (r[0], r[1]))
    if runs:
        tgt = np.concatenate([g[:, 0] for _, _, g in runs])
        src = np.concatenate([g[:, 1] for _, _, g in runs])
        rel = np.concatenate([np.full(len(g), r, dtype=np.int64) for r, _, g in runs])
    else:
        tgt = src = rel = np.zeros(0, np.int64)
    for r in range(R):      # the sampler guarantees ascending targets inside a run; verify once per relation (cheap, vectorised)
        seg = tgt[rel == r]
        if seg.size > 1 and np.any(seg[1:] < seg[:-1]):
            order = np.argsort(rel * (len(ntype) + 1) + tgt, kind="stable")   # generic fallback: one stable host sort
            tgt, src, rel = tgt[order], src[order], rel[order]
            break
    etime = ntime[tgt] - ntime[src] + 120
    rel_ptr = np.searchsorted(rel, np.arange(R + 1)).astype(np.int32)
    dev = torch.device(device)
    node_feature = torch.from_numpy(feat).to(dev)
    node_type = torch.from_numpy(ntype).to(dev)
    # int32 form for the plan (12 B / edge over PCIe instead of 40 B of int64 wire format) ...
    src32 = torch.from_numpy(src.astype(np.int32)).to(dev)
    dst32 = torch.from_numpy(tgt.astype(np.int32)).to(dev)
    time32 = torch.from_numpy(etime.astype(np.int32)).to(dev)
    # ... and the reference's int64 tensors, derived on the device (forward() signature, keep_att order, backward plans)
    edge_index = torch.stack([src32.long(), dst32.long()], dim=1).t()
    edge_type = torch.repeat_interleave(torch.arange(R, device=dev), torch.from_numpy(np.diff(rel_ptr).astype(np.int64)).to(dev))
    edge_time = time32.long()
    T = len(types)
    plan = GraphPlan.from_sorted(node_type, edge_index, edge_type, edge_time, src32, dst32, time32, torch.from_numpy(rel_ptr).to(dev),
                                 torch.from_numpy(type_off).to(dev), T, R)
    GraphPlan.register(plan, node_type, edge_index, edge_type, edge_time, T, R)
    out = _DeviceGraph((node_feature, node_type, edge_time, edge_index, edge_type, node_dict, edge_dict))
    out.plan = plan
    return out


class _DeviceGraph(tuple):
    """The 7-tuple of to_torch with the prebuilt plan attached (`.plan`)."""
    plan = None
