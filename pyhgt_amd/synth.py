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
