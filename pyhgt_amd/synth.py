"""Seeded synthetic typed graphs in the wire format HGTConv.forward consumes.

The format is what the reference's sampler emits (/root/reference/pyHGT/data.py:212-256):
node_feature f32[N,d], node_type i64[N] (type-contiguous, ascending), edge_index
i64[2,E] delivered as the (1,2)-strided transpose of an [E,2] list (data.py:245,254),
edge_type i64[E], edge_time i64[E] in [0,240) (data.py:250).  The recipe follows
SURVEY.md section 8(d): uniform sources/targets (Poisson in-degree), relation
ids independent of node types (every (src_type, rel, dst_type) combination occurs).
"""
import torch


def synthetic_typed_graph(num_nodes, num_edges, dim, num_types, num_relations, seed=0,
                          sorted_types=True, strided_edge_index=True, dst_skew=0.0,
                          with_time=True, schema=False, device="cpu"):
    """Returns (node_feature, node_type, edge_index, edge_type, edge_time) on `device`.

    dst_skew > 0 draws targets from a Pareto-like law (hub nodes) instead of uniform;
    schema=True ties every relation to one fixed (src_type, dst_type) pair like real
    heterogeneous data (requires sorted_types).
    """
    g = torch.Generator().manual_seed(seed)
    N, E, T, R = num_nodes, num_edges, num_types, num_relations
    node_type = torch.randint(0, T, (N,), generator=g)
    if sorted_types:
        node_type = node_type.sort().values
    node_feature = torch.randn(N, dim, generator=g)
    edge_type = torch.randint(0, R, (E,), generator=g)
    if schema:
        if not sorted_types:
            raise ValueError("schema graphs need type-contiguous nodes")
        counts = torch.bincount(node_type, minlength=T)
        offs = torch.cumsum(counts, 0) - counts
        rel_src = torch.randint(0, T, (R,), generator=g)
        rel_dst = torch.randint(0, T, (R,), generator=g)
        u = torch.rand(E, generator=g)
        w = torch.rand(E, generator=g)
        st, dt = rel_src[edge_type], rel_dst[edge_type]
        src = offs[st] + (u * counts[st].clamp(min=1)).long().clamp(max=N - 1)
        dst = offs[dt] + (w * counts[dt].clamp(min=1)).long().clamp(max=N - 1)
    else:
        src = torch.randint(0, N, (E,), generator=g)
        if dst_skew > 0.0:
            u = torch.rand(E, generator=g).clamp_min(1e-9)
            dst = ((u ** (-1.0 / dst_skew) - 1.0) % N).long().clamp(0, N - 1)
            dst = torch.randperm(N, generator=g)[dst]
        else:
            dst = torch.randint(0, N, (E,), generator=g)
    if strided_edge_index:
        edge_index = torch.stack([src, dst], dim=1).contiguous().t()   # [2,E] view, strides (1,2)
    else:
        edge_index = torch.stack([src, dst], dim=0).contiguous()
    if with_time:
        edge_time = torch.randint(0, 240, (E,), generator=g)
    else:
        edge_time = torch.zeros(E, dtype=torch.long)
    out = (node_feature, node_type, edge_index, edge_type, edge_time)
    if device != "cpu":
        # .to() keeps the strides of edge_index
        out = tuple(t.to(device) for t in out)
    return out


def pick_check_targets(node_type, dst, n_random=1600, tile=64, seed=0):
    """Target rows for a sampled parity check of a LARGE graph: rows around every node-type boundary (mixed-type tiles of the
    fused update), the first and the last (ragged) tile, the rows of maximum / minimum in-degree, and `n_random` random rows.
    node_type / dst may live on any device; returns a sorted unique int64 tensor on that device."""
    N = int(node_type.numel())
    dev = node_type.device
    picks = [torch.arange(0, min(N, tile), device=dev), torch.arange(max(0, N - tile - 7), N, device=dev)]
    change = (node_type[1:] != node_type[:-1]).nonzero().flatten()[:16]          # type-sorted graphs: T-1 boundaries
    for b in change.tolist():
        picks.append(torch.arange(max(0, b - tile // 2), min(N, b + tile // 2), device=dev))
    deg = torch.bincount(dst, minlength=N)
    picks.append(deg.argmax().reshape(1))
    picks.append(deg.argmin().reshape(1))
    picks.append(torch.topk(deg, min(4, N)).indices)
    g = torch.Generator(device="cpu").manual_seed(seed)
    picks.append(torch.randint(0, N, (n_random,), generator=g).to(dev))
    return torch.unique(torch.cat([p.to(torch.int64) for p in picks]))


def induced_in_neighbourhood(node_feature, node_type, edge_index, edge_type, edge_time, targets):
    """The sub-graph that determines HGTConv's output rows `targets` exactly: ALL in-edges of those targets (original order)
    and the source nodes they reference, ids remapped to [0, n_sub).  A target's output depends on nothing else (softmax and
    aggregation are per target, conv.py:108-111; update is per node, conv.py:114-134), so an oracle run on this graph is
    exact for those rows -- the other nodes of the sub-graph lose in-edges and their outputs are meaningless.
    Returns CPU tensors (x_sub, node_type_sub, edge_index_sub [2,E_sub], edge_type_sub, edge_time_sub or None,
    pos) where pos[i] is the row of targets[i] in the sub-graph."""
    N = int(node_type.numel())
    dev = node_type.device
    src, dst = edge_index[0], edge_index[1]
    flag = torch.zeros(N, dtype=torch.bool, device=dev)
    flag[targets] = True
    eids = flag[dst].nonzero().flatten()
    s, t = src[eids], dst[eids]
    nodes = torch.unique(torch.cat([targets, s]))
    ei_sub = torch.stack([torch.searchsorted(nodes, s), torch.searchsorted(nodes, t)], dim=0)
    pos = torch.searchsorted(nodes, targets)
    tm = edge_time[eids].cpu() if edge_time is not None else None
    return (node_feature[nodes].cpu(), node_type[nodes].cpu(), ei_sub.cpu(), edge_type[eids].cpu(), tm, pos.cpu())
