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

Overlap (n_chunks > 1, the default on GPUs): every peer's rows are cut into n_chunks slices and the
halo part of the local buffer is ordered (chunk, peer, row).  A step then runs

    pack(c0), all-to-all(c0, async) | Q|K|V of the own rows | pack(c1), all-to-all(c1, async) |
    wait(c0), K|V of chunk 0 | pack(c2), ... | wait(c_last), K|V of the last chunk | edge phase + update

through hgt_conv_forward's stages 1/2/3, so the exchange (RCCL's own stream) overlaps the own-row
projections, the packing and the halo projections of the earlier chunks; only the edge phase needs
every source row.

Source-bucketed edge phase (bucketed=True, the default where it applies: split-bf16 precision and
(n_chunks + 1) * num_relations < 64).  The edge phase does not have to wait for the last chunk either: a rank's edges are
bucketed by where their SOURCE row comes from -- bucket 0 = own rows, bucket c + 1 = halo chunk c -- by numbering relations
`bucket * R + relation` in the plan (the relation parameters are repeated per bucket, a few MB).  The plan then keeps, inside
every target tile, the edges of one bucket together, and hgt_conv_forward's stage 4 runs logits + aggregation over ONE bucket,
carrying the online-softmax state (reference, exp-sum, un-normalised rows) from bucket to bucket in the workspace:

    pack(all chunks), all-to-all(c0 .. c_last, async, back to back on RCCL's stream) |
    Q|K|V of the own rows | edge phase of bucket 0 (own sources) |
    wait(c0), K|V of chunk 0, edge phase of bucket 1 | ... | wait(c_last), K|V, edge phase of the last bucket + update

so that only the last chunk's projections, its bucket of edges and the node update are behind the exchange.  Partial softmax
results combine exactly (m = max, rescale both sides): the result equals the one-call layer up to fp32 rounding
(tests/test_hgt_gpu.py::test_bucketed_edge_phase_matches_one_call_layer).
"""
import torch
import torch.distributed as dist

from . import _lib


def _host_staged(t, group):
    """gloo has no device all-to-all: device tensors are staged through the host (only used to run the multi-GPU code path
    on a box without several GPUs; RCCL takes the device tensors as they are)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all(recv, send, recv_splits, send_splits, group, async_op=False):
    """dist.all_to_all_single, host-staged under gloo.  Returns an object with wait() when async_op."""
    if not _host_staged(send, group):
        return dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group, async_op=async_op)
    recv_h = torch.empty(recv.shape, dtype=recv.dtype)
    work = dist.all_to_all_single(recv_h, send.cpu(), recv_splits, send_splits, group=group, async_op=async_op)

    class _Staged:
        def wait(self):
            if work is not None:
                work.wait()
            recv.copy_(recv_h)
            return True
    if async_op:
        return _Staged()
    recv.copy_(recv_h)
    return None


def _all_to_all_int64(send, send_splits, recv_splits, group):
    recv = send.new_empty(int(sum(recv_splits)))
    _all_to_all(recv, send, recv_splits, send_splits, group)
    return recv


class HaloPlan:
    """Graph-only part of the exchange: which of my rows every peer needs, and where the rows I
    receive go.  Pure index arithmetic + three small all-to-alls; backend-agnostic (the CPU tests
    run it over gloo)."""

    def __init__(self, node_type_own, src_global, node_offsets, rank, world, group=None, n_chunks=1):
        dev = src_global.device
        self.rank, self.world, self.group = rank, world, group
        self.n_chunks = C = max(1, int(n_chunks))
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
        _all_to_all(got, counts, None, None, group)
        self.send_splits = got.tolist()
        # tell every owner which of its rows I need; receive which of my rows the peers need
        asked = _all_to_all_int64(need, self.recv_splits, self.send_splits, group)
        send_rows = (asked - lo).to(torch.int32)                                           # local row ids, grouped by peer
        # node types of my halo rows (owners answer in the order I asked)
        types_for_peers = node_type_own[(asked - lo)]
        halo_types = _all_to_all_int64(types_for_peers.contiguous(), self.send_splits, self.recv_splits, group)
        # Chunk-major order of the halo rows / of the rows I send: slice c of a peer's list of length L is
        # [c*L//C, (c+1)*L//C) -- both sides derive the same slices from the same per-peer lengths.
        def chunk_major(splits):
            starts = [0]
            for n in splits:
                starts.append(starts[-1] + n)
            pieces, per_chunk = [], []
            for c in range(C):
                sizes = []
                for p, n in enumerate(splits):
                    a, b = starts[p] + (c * n) // C, starts[p] + ((c + 1) * n) // C
                    pieces.append(torch.arange(a, b, device=dev))
                    sizes.append(b - a)
                per_chunk.append(sizes)
            order = torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.int64, device=dev)
            return order, per_chunk
        recv_order, self.recv_chunk_splits = chunk_major(self.recv_splits)     # new halo position -> position in `need`
        send_order, self.send_chunk_splits = chunk_major(self.send_splits)
        self.send_rows = send_rows[send_order].contiguous()
        self.halo_types = halo_types[recv_order].contiguous()
        inv = torch.empty_like(recv_order)
        inv[recv_order] = torch.arange(recv_order.numel(), device=dev)                     # position in `need` -> new halo position
        # local id of every edge source: own rows first, then halo rows (chunk, peer, id order)
        pos = torch.searchsorted(need, src_global.clamp(min=0)) if self.n_halo > 0 else torch.zeros_like(src_global)
        pos = inv[pos.clamp(max=max(self.n_halo - 1, 0))] if self.n_halo > 0 else pos
        self.src_local = torch.where(remote_mask, self.n_own + pos, src_global - lo)
        self.node_type_local = torch.cat([node_type_own, self.halo_types])
        self.n_local = self.n_own + self.n_halo
        self.halo_order = recv_order      # halo row i holds global node need[halo_order[i]]
        self.need = need
        # per chunk: offsets into the send list / the halo rows
        self.send_chunk_off = [0]
        self.recv_chunk_off = [0]
        for c in range(C):
            self.send_chunk_off.append(self.send_chunk_off[-1] + sum(self.send_chunk_splits[c]))
            self.recv_chunk_off.append(self.recv_chunk_off[-1] + sum(self.recv_chunk_splits[c]))

    def to(self, device):
        """Move the index tensors of the plan to `device` (the plan itself can be built on any backend)."""
        for k, v in list(vars(self).items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def chunk_row_lists(self, num_types):
        """Typed row lists (int32 rows grouped by node type, int32 offsets[T+1]) of the halo rows of every chunk,
        in the form hgt_conv_forward stage 2 takes.  Rows of a type outside [0, T) get no projection (their edges are
        unclaimed, conv.py:68-69)."""
        lists = []
        for c in range(self.n_chunks):
            a, b = self.recv_chunk_off[c], self.recv_chunk_off[c + 1]
            t = self.halo_types[a:b]
            valid = (t >= 0) & (t < num_types)
            key = torch.where(valid, t, torch.full_like(t, num_types))
            order = torch.argsort(key, stable=True)
            counts = torch.bincount(key, minlength=num_types + 1)[:num_types]
            off = torch.zeros(num_types + 1, dtype=torch.int64, device=t.device)
            off[1:] = torch.cumsum(counts, 0)
            n_valid = int(valid.sum())
            rows = (self.n_own + a + order[:n_valid]).to(torch.int32).contiguous()
            lists.append((rows, off.to(torch.int32).contiguous()))
        return lists

    def edge_buckets(self):
        """Source bucket of every edge (module docstring): 0 = the source is an own row, 1 + c = it arrives with halo chunk c
        (halo rows are stored in chunk order)."""
        bounds = torch.tensor(self.recv_chunk_off[1:], dtype=torch.int64, device=self.src_local.device)
        halo_pos = self.src_local - self.n_own
        in_chunk = torch.searchsorted(bounds, halo_pos.clamp(min=0), right=True)
        return torch.where(halo_pos >= 0, 1 + in_chunk, torch.zeros_like(halo_pos))

    def bucketed_edge_types(self, edge_type, num_relations):
        """Relation ids of the source-bucketed plan: bucket * R + relation for the edges a relation claims, (n_chunks + 1) * R
        (= the plan's unclaimed bucket) for relation ids outside [0, R) (conv.py:68-69: logit 0, no message, still in the softmax)."""
        claimed = (edge_type >= 0) & (edge_type < num_relations)
        return torch.where(claimed, self.edge_buckets() * num_relations + edge_type,
                           torch.full_like(edge_type, (self.n_chunks + 1) * num_relations))

    def exchange_chunk(self, c, x_own, x_local, pack=None, async_op=False, compress=False):
        """One slice of the exchange: pack the rows of chunk c the peers need, all-to-all them into the halo rows of
        chunk c.  Returns (work, buffers) when async_op: work.wait() makes the current stream wait for the rows (and, with
        compress, expands them).  compress: rows travel in the 24-bit format of hgt_gather_rows_c24 (3/4 of the bytes;
        relative error <= 2^-16 on halo features, which only feed the K/V projections)."""
        d = x_own.size(1)
        rows = self.send_rows[self.send_chunk_off[c]:self.send_chunk_off[c + 1]]
        recv = x_local[self.n_own + self.recv_chunk_off[c]:self.n_own + self.recv_chunk_off[c + 1]]
        lib = _lib.load() if x_own.is_cuda else None
        st = torch.cuda.current_stream().cuda_stream if x_own.is_cuda else None
        if compress and pack is None and x_own.is_cuda and d % 4 == 0:
            send = torch.empty(rows.numel(), 3 * d, dtype=torch.uint8, device=x_own.device)
            _lib.check(lib.hgt_gather_rows_c24(x_own.data_ptr(), x_own.stride(0), rows.data_ptr(), rows.numel(), d,
                                               send.data_ptr(), st), "hgt_gather_rows_c24")
            wire = torch.empty(recv.size(0), 3 * d, dtype=torch.uint8, device=x_own.device)
            work = _all_to_all(wire, send, list(self.recv_chunk_splits[c]), list(self.send_chunk_splits[c]), self.group,
                               async_op=async_op)
            n_recv, ld = recv.size(0), x_local.stride(0)

            def expand():
                _lib.check(lib.hgt_unpack_rows_c24(wire.data_ptr(), n_recv, d, recv.data_ptr(), ld,
                                                   torch.cuda.current_stream().cuda_stream), "hgt_unpack_rows_c24")

            class _Expanding:
                def wait(self):
                    if work is not None:
                        work.wait()
                    expand()
                    return True
            if async_op:
                return _Expanding(), (send, wire)
            expand()
            return None
        if pack is None:
            if not x_own.is_cuda:
                raise RuntimeError("pyhgt_amd.dist: halo packing runs the HIP gather kernel; CPU tensors need an explicit pack fn")
            send = torch.empty(rows.numel(), d, dtype=x_own.dtype, device=x_own.device)
            _lib.check(lib.hgt_gather_rows(x_own.data_ptr(), x_own.stride(0), rows.data_ptr(), rows.numel(), d,
                                           send.data_ptr(), st), "hgt_gather_rows")
        else:
            send = pack(x_own, rows)
        work = _all_to_all(recv, send, list(self.recv_chunk_splits[c]), list(self.send_chunk_splits[c]), self.group,
                           async_op=async_op)
        return (work, (send,)) if async_op else None

    def exchange(self, x_own, x_local, pack=None):
        """Fill x_local[n_own:] with the halo rows (x_local[:n_own] must already hold x_own), one slice after the other.
        `pack(x_own, rows_int32) -> [len(rows), d]`; defaults to the HIP gather kernel on GPU."""
        for c in range(self.n_chunks):
            self.exchange_chunk(c, x_own, x_local, pack=pack)
        return x_local


class PartitionedGraph:
    """One rank's share of a destination-partitioned typed graph + the per-layer forward."""

    def __init__(self, node_type_own, src_global, dst_local, edge_type, edge_time, num_types, num_relations,
                 nodes_per_rank, rank, world, group=None, node_offsets=None, n_chunks=4, halo=None, compress=False, bucketed=None):
        """halo: a prebuilt HaloPlan for this rank (tests build it on CPU over gloo and move it to the device with
        HaloPlan.to); otherwise it is negotiated here with three small all-to-alls.
        bucketed: source-bucketed edge phase (module docstring); None = wherever it applies (decided per layer in forward:
        it needs the split-bf16 precision), False = the edge phase waits for the last chunk (stages 1/2/3)."""
        from .conv import GraphPlan
        if node_offsets is None:
            node_offsets = [nodes_per_rank * r for r in range(world + 1)]
        self.halo = halo if halo is not None else HaloPlan(node_type_own, src_global, node_offsets, rank, world, group,
                                                            n_chunks=n_chunks)
        self.compress = bool(compress)     # 24-bit halo rows on the links (exchange_chunk); off: exact fp32 rows
        C = self.halo.n_chunks
        self.n_buckets = C + 1
        can_bucket = self.n_buckets * num_relations < 64        # the streaming walk keeps one range per relation id in a lane
        self.bucketed = can_bucket if bucketed is None else bool(bucketed)
        if self.bucketed and not can_bucket:
            raise ValueError("bucketed edge phase needs (n_chunks + 1) * num_relations < 64")
        self.chunk_lists = self.halo.chunk_row_lists(num_types) if (C > 1 or self.bucketed) else None
        self.n_own, self.n_local = self.halo.n_own, self.halo.n_local
        self.edge_index = torch.stack([self.halo.src_local, dst_local], dim=0).contiguous()
        self.edge_type, self.edge_time = edge_type, edge_time
        self.node_type_local = self.halo.node_type_local
        self.num_relations = num_relations
        self.plan = GraphPlan(self.node_type_local, self.edge_index, edge_type, edge_time, num_types, num_relations,
                              n_q_rows=self.n_own)
        self.bucket_plan = None
        if self.bucketed:
            # bucket of an edge = where its source row comes from: 0 own, 1 + c halo chunk c (halo rows are in chunk order)
            self.edge_type_bucketed = self.halo.bucketed_edge_types(edge_type, num_relations)
            self.bucket_plan = GraphPlan(self.node_type_local, self.edge_index, self.edge_type_bucketed, edge_time, num_types,
                                         self.n_buckets * num_relations, n_q_rows=self.n_own)
        self.x_local = None
        self.workspace = None      # owned here: Q/K/V stay in it between the stages of one step

    def forward(self, layer, x_own, phase_events=None):
        d = x_own.size(1)
        bucketed = self.bucketed and getattr(layer, "precision", None) in ("bf16x3", "f16x3") and getattr(layer, "_UPDATE_MODE", 0) == 0
        need = layer.workspace_bytes(self.n_local, self.plan.E, self.n_buckets if bucketed else 1)
        if self.workspace is None or self.workspace.numel() < need or self.workspace.device != x_own.device:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=x_own.device)
        if self.x_local is None or self.x_local.size(1) != d:
            self.x_local = torch.empty(self.n_local, d, dtype=x_own.dtype, device=x_own.device)
        if x_own.data_ptr() != self.x_local.data_ptr():
            self.x_local[:self.n_own].copy_(x_own)
        x_own_v = self.x_local[:self.n_own]
        if self.chunk_lists is None:
            for c in range(self.halo.n_chunks):
                self.halo.exchange_chunk(c, x_own_v, self.x_local, compress=self.compress)
            return layer(self.x_local, self.node_type_local, self.edge_index, self.edge_type, self.edge_time,
                         plan=self.plan, n_q_rows=self.n_own, phase_events=phase_events, workspace=self.workspace)
        if bucketed:
            S, C = self.n_buckets, self.halo.n_chunks
            args = (self.x_local, self.node_type_local, self.edge_index, self.edge_type_bucketed, self.edge_time)
            kw = dict(plan=self.bucket_plan, n_q_rows=self.n_own, workspace=self.workspace)
            # every chunk is packed and queued on the links up front: the transfers run back to back on RCCL's stream
            pending = [self.halo.exchange_chunk(c, x_own_v, self.x_local, async_op=True, compress=self.compress) for c in range(C)]
            layer(*args, stage=1, slices=(0, S), phase_events=phase_events, **kw)     # Q|K|V of the own rows
            out = layer(*args, stage=4, slices=(0, S), **kw)                           # edges whose source is an own row
            for c in range(C):
                work, bufs = pending[c]
                work.wait()
                for b in bufs:
                    b.record_stream(torch.cuda.current_stream())
                layer(*args, stage=2, proj=self.chunk_lists[c], slices=(0, S), **kw)   # K|V of the halo rows of chunk c
                out = layer(*args, stage=4, slices=(c + 1, S), phase_events=phase_events if c == C - 1 else None, **kw)
            return out
        # pipelined: chunk c+1 is packed and put on the links while chunk c's halo rows are projected
        args = (self.x_local, self.node_type_local, self.edge_index, self.edge_type, self.edge_time)
        kw = dict(plan=self.plan, n_q_rows=self.n_own, workspace=self.workspace)
        C = self.halo.n_chunks
        pending = [self.halo.exchange_chunk(0, x_own_v, self.x_local, async_op=True, compress=self.compress)]
        layer(*args, stage=1, phase_events=phase_events, **kw)            # Q|K|V of the own rows
        for c in range(C):
            if c + 1 < C:
                pending.append(self.halo.exchange_chunk(c + 1, x_own_v, self.x_local, async_op=True, compress=self.compress))
            work, bufs = pending[c]
            work.wait()                                                   # current stream waits for chunk c (and expands it)
            for b in bufs:
                b.record_stream(torch.cuda.current_stream())
            layer(*args, stage=2, proj=self.chunk_lists[c], **kw)         # K|V of the halo rows of chunk c
        return layer(*args, stage=3, phase_events=phase_events, **kw)
