"""Destination-partitioned HGTConv across GPUs (one process per GPU, torch.distributed over RCCL).

The reference is single-GPU (no collective anywhere), so this is new design,
not a translation: softmax and aggregation are per TARGET node (conv.py:108, aggr='add'
conv.py:13) and update() is per node, so a rank that owns a contiguous range of target nodes and
ALL their in-edges needs no reduction with other ranks -- only the source rows x_j of in-edges
whose source lives on another rank (halo).  Per layer there is exactly one exchange step:

    pack rows peers need (hgt_gather_rows)  ->  all_to_all_single (RCCL; on the xGMI full mesh every
    peer pair has its own link, so the 7 transfers of a rank run concurrently)  ->  HGTConv on
    [own rows ; halo rows] with n_q_rows = own (halo rows only get K/V projections).

Halo ids are deduplicated per rank, so a source referenced by many local edges crosses the link
once.  Everything that depends only on the graph (halo id lists, split sizes, halo node types,
local edge ids, the GraphPlan) is built once in __init__.
"""
import torch
import torch.distributed as dist

from . import _lib


def _all_to_all_int64(send, send_splits, recv_splits, group):
    recv = send.new_empty(int(sum(recv_splits)))
    dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group)
    return recv


class HaloPlan:
    """Graph-only part of the exchange: which of my rows every peer needs, and where the rows I
    receive go.  Pure index arithmetic + three small all-to-alls; backend-agnostic (the CPU tests
    run it over gloo)."""

    def __init__(self, node_type_own, src_global, node_offsets, rank, world, group=None):
        dev = src_global.device
        self.rank, self.world, self.group = rank, world, group
        self.offsets = torch.as_tensor(node_offsets, dtype=torch.int64, device=dev)       # [world+1]
        lo, hi = int(self.offsets[rank]), int(self.offsets[rank + 1])
        self.n_own = hi - lo
        remote_mask = (src_global < lo) | (src_global >= hi)
        need = torch.unique(src_global[remote_mask])                                       # sorted, deduplicated
        self.n_halo = int(need.numel())
        owner_bounds = torch.searchsorted(need, self.offsets)                              # need is sorted by owner
        self.recv_splits = (owner_bounds[1:] - owner_bounds[:-1]).tolist()
        counts = torch.tensor(self.recv_splits, dtype=torch.int64, device=dev)
        got = torch.empty_like(counts)
        dist.all_to_all_single(got, counts, group=group)
        self.send_splits = got.tolist()
        # tell every owner which of its rows I need; receive which of my rows the peers need
        asked = _all_to_all_int64(need, self.recv_splits, self.send_splits, group)
        self.send_rows = (asked - lo).to(torch.int32)                                      # local row ids, grouped by peer
        # node types of my halo rows (owners answer in the order I asked)
        types_for_peers = node_type_own[(asked - lo)]
        self.halo_types = _all_to_all_int64(types_for_peers.contiguous(), self.send_splits, self.recv_splits, group)
        # local id of every edge source: own rows first, then halo rows in `need` order
        pos = torch.searchsorted(need, src_global.clamp(min=0)) if self.n_halo > 0 else torch.zeros_like(src_global)
        self.src_local = torch.where(remote_mask, self.n_own + pos, src_global - lo)
        self.node_type_local = torch.cat([node_type_own, self.halo_types])
        self.n_local = self.n_own + self.n_halo

    def exchange(self, x_own, x_local, pack=None):
        """Fill x_local[n_own:] with the halo rows (x_local[:n_own] must already hold x_own).
        `pack(x_own, rows_int32) -> [len(rows), d]`; defaults to the HIP gather kernel on GPU."""
        d = x_own.size(1)
        if pack is None:
            if not x_own.is_cuda:
                raise RuntimeError("pyhgt_amd.dist: halo packing runs the HIP gather kernel; CPU tensors need an explicit pack fn")
            send = torch.empty(self.send_rows.numel(), d, dtype=x_own.dtype, device=x_own.device)
            _lib.check(_lib.load().hgt_gather_rows(x_own.data_ptr(), x_own.stride(0), self.send_rows.data_ptr(),
                                                   self.send_rows.numel(), d, send.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream), "hgt_gather_rows")
        else:
            send = pack(x_own, self.send_rows)
        recv = x_local[self.n_own:]
        dist.all_to_all_single(recv, send, [s for s in self.recv_splits], [s for s in self.send_splits], group=self.group)
        return x_local


class PartitionedGraph:
    """One rank's share of a destination-partitioned typed graph + the per-layer forward."""

    def __init__(self, node_type_own, src_global, dst_local, edge_type, edge_time, num_types, num_relations,
                 nodes_per_rank, rank, world, group=None, node_offsets=None):
        from .conv import GraphPlan
        if node_offsets is None:
            node_offsets = [nodes_per_rank * r for r in range(world + 1)]
        self.halo = HaloPlan(node_type_own, src_global, node_offsets, rank, world, group)
        self.n_own, self.n_local = self.halo.n_own, self.halo.n_local
        self.edge_index = torch.stack([self.halo.src_local, dst_local], dim=0).contiguous()
        self.edge_type, self.edge_time = edge_type, edge_time
        self.node_type_local = self.halo.node_type_local
        self.plan = GraphPlan(self.node_type_local, self.edge_index, edge_type, edge_time, num_types, num_relations,
                              n_q_rows=self.n_own)
        self.x_local = None

    def forward(self, layer, x_own, phase_events=None):
        d = x_own.size(1)
        if self.x_local is None or self.x_local.size(1) != d:
            self.x_local = torch.empty(self.n_local, d, dtype=x_own.dtype, device=x_own.device)
        if x_own.data_ptr() != self.x_local.data_ptr():
            self.x_local[:self.n_own].copy_(x_own)
        self.halo.exchange(self.x_local[:self.n_own], self.x_local)
        return layer(self.x_local, self.node_type_local, self.edge_index, self.edge_type, self.edge_time,
                     plan=self.plan, n_q_rows=self.n_own, phase_events=phase_events)
