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
once.  Everything that depends only on the graph (partition, halo id lists, split sizes, halo node
types, local edge ids, the GraphPlan) is built once.

Partitioner ("contiguous dst ranges balanced by in-edge count").  `partition()` cuts a GLOBAL typed graph
into `world` contiguous target ranges whose in-edge counts are equal up to one plan tile of targets (prefix sums of the
in-degrees, cut points rounded to the tile so that no tile of the kernels' plan straddles two ranks), and hands every rank its
share: its rows of node_type, all in-edges of its targets (global source ids, local target ids), types and times.

Schedules of a layer (PartitionedGraph(mode=...)):

"blocked" (round 4, the default where it applies: split precision, padded row <= 256 columns, HGTConv update).  The rank's own
targets are cut into B contiguous TARGET BLOCKS of equal in-edge count (tile aligned) and the halo rows are ordered by the FIRST
block that needs them: chunk b = the remote rows whose first use is in block b.  Block b's in-edges then only reference own rows and
chunks 0..b, so a step is

    pack + all-to-all(chunk 0 .. B-1, async, back to back on RCCL's stream)  |  Q|K|V of the own rows  |
    wait(chunk 0), K|V of chunk 0, edge phase + fused update of block 0  |  wait(chunk 1), K|V of chunk 1, block 1  | ...

and every block runs the SINGLE-GPU kernel pair (logits, aggregation with the node update fused in) on a range of destination
tiles (hgt_conv_forward stage 5): no softmax state is carried between launches, nothing is written and re-read between them
(round 2/3's source-bucketed schedule paid a 1 KB row + 64 B of state per target and bucket boundary and an unfused update), and
only the LAST block -- 1/B of the edge phase, plus the K|V of the smallest chunk -- sits behind the exchange.  First-use chunks
shrink (uniform sources, 8 ranks, B = 8: 20 / 17 / 15 / 13 / 11 / 9 / 8 / 7 % of the halo rows), which front-loads the links
exactly when nothing else competes for them.  With the 24-bit wire format the halo rows are projected straight off the wire
buffer (hgt_conv_args.proj_c24): no expansion pass, 3/4 of the bytes read.

"bucketed" (round 2): source-bucketed edge phase, hgt_conv_forward stage 4 -- a rank's edges are bucketed by where their SOURCE row
comes from (bucket 0 = own rows, bucket c + 1 = halo chunk c; equal slices of every peer's list) by numbering relations
`bucket * R + relation` in the plan, and the online-softmax state (reference, exp-sum, un-normalised rows) is carried from bucket to
bucket in the workspace.  Kept for layers the blocked schedule does not cover and as a cross-check (tests).

"pipelined" (round 1): the exchange overlaps the projections only; the edge phase waits for the last chunk (stages 1/2/3).  The
fallback for every layer (exact fp32 precision, DenseHGTConv, rows wider than 256 columns).
"""
import torch
import torch.distributed as dist

from . import _lib

PLAN_TILE = 256      # destination tile of the kernels' plan (hgt_plan_constants; checked against the library on first GPU use)


# ------------------------------------------------------------------------------------------------------------
# partitioner
# ------------------------------------------------------------------------------------------------------------
def partition_offsets(dst, n_nodes, world, align=PLAN_TILE):
    """Cut [0, n_nodes) into `world` contiguous target ranges with (nearly) equal in-edge counts: the r-th cut is the multiple of
    `align` whose in-edge prefix sum is closest to r * E / world.  Returns a list of world + 1 offsets
    (non-decreasing; the last one is n_nodes; a rank may be empty on tiny graphs)."""
    n_nodes, world, align = int(n_nodes), int(world), max(1, int(align))
    E = int(dst.numel())
    if world <= 1 or n_nodes == 0:
        return [0] + [n_nodes] * max(world, 1)
    n_cand = (n_nodes + align - 1) // align + 1                       # candidate cut points 0, align, 2 align, ..., n_nodes
    deg_tile = torch.bincount(torch.div(dst, align, rounding_mode="floor"), minlength=n_cand - 1).to(torch.int64)
    prefix = torch.zeros(n_cand, dtype=torch.int64, device=dst.device)
    prefix[1:] = torch.cumsum(deg_tile, 0)                            # prefix[c] = in-edges of targets < min(c * align, n_nodes)
    targets = torch.tensor([(r * E) // world for r in range(1, world)], dtype=torch.int64, device=dst.device)
    hi = torch.searchsorted(prefix, targets).clamp(1, n_cand - 1)     # first candidate with prefix >= target
    lo = hi - 1
    pick = torch.where((targets - prefix[lo]) <= (prefix[hi] - targets), lo, hi)
    cuts = [0] + [min(int(c) * align, n_nodes) for c in pick.tolist()] + [n_nodes]
    for i in range(1, len(cuts)):                                     # monotone (ties on tiny graphs)
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def partition(node_type, edge_index, edge_type, edge_time, world, rank, align=PLAN_TILE, node_offsets=None):
    """One rank's share of a GLOBAL typed graph under the in-edge-balanced destination partition.

    node_type i64[N], edge_index i64[2, E] (row 0 = source, row 1 = target), edge_type i64[E], edge_time i64[E] or None --
    the tensors HGTConv.forward takes (conv.py:56).  Returns a dict: node_offsets (world + 1 cut points, shared by all ranks
    because every rank derives them from the same graph), node_type_own, src_global, dst_local, edge_type, edge_time (the rank's
    in-edges in their original relative order) and edge_ids (their positions in the global edge list)."""
    N = int(node_type.numel())
    if node_offsets is None:
        node_offsets = partition_offsets(edge_index[1], N, world, align)
    lo, hi = int(node_offsets[rank]), int(node_offsets[rank + 1])
    dst = edge_index[1]
    mine = ((dst >= lo) & (dst < hi)).nonzero(as_tuple=True)[0]
    return dict(node_offsets=list(node_offsets), node_type_own=node_type[lo:hi].contiguous(),
                src_global=edge_index[0][mine].contiguous(), dst_local=(dst[mine] - lo).contiguous(),
                edge_type=edge_type[mine].contiguous(), edge_time=None if edge_time is None else edge_time[mine].contiguous(),
                edge_ids=mine)


def partition_offsets_weighted(dst, n_nodes, weights, align=PLAN_TILE):
    """partition_offsets with given relative in-edge shares per part (weights need not be normalised)."""
    n_nodes, align = int(n_nodes), max(1, int(align))
    parts = len(weights)
    E = int(dst.numel())
    if parts <= 1 or n_nodes == 0:
        return [0] + [n_nodes] * max(parts, 1)
    n_cand = (n_nodes + align - 1) // align + 1
    deg_tile = torch.bincount(torch.div(dst, align, rounding_mode="floor"), minlength=n_cand - 1).to(torch.int64)
    prefix = torch.zeros(n_cand, dtype=torch.int64, device=dst.device)
    prefix[1:] = torch.cumsum(deg_tile, 0)
    tot = float(sum(weights))
    acc, tg = 0.0, []
    for w in weights[:-1]:
        acc += float(w)
        tg.append(int(round(E * acc / tot)))
    targets = torch.tensor(tg, dtype=torch.int64, device=dst.device)
    hi = torch.searchsorted(prefix, targets).clamp(1, n_cand - 1)
    lo = hi - 1
    pick = torch.where((targets - prefix[lo]) <= (prefix[hi] - targets), lo, hi)
    cuts = [0] + [min(int(c) * align, n_nodes) for c in pick.tolist()] + [n_nodes]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def target_blocks(dst_local, n_own, n_blocks, align=PLAN_TILE, shape="equal"):
    """Bounds (n_blocks + 1 offsets, multiples of `align` except the last) of contiguous target blocks.  shape "equal": equal
    in-edge counts (the smallest tail behind the exchange: 1 / n_blocks of the edge phase -- the choice when the links bound the
    step); "geometric": shares 1, 1, 2, 4, ... (tiny first blocks start as soon as the first rows arrive, the large last ones run
    at the one-call layer's efficiency -- the choice when the GPU side bounds the step; the tail is half the edge phase)."""
    if shape == "geometric" and n_blocks > 1:
        return partition_offsets_weighted(dst_local, n_own, [1.0] + [2.0 ** i for i in range(n_blocks - 1)], align)
    return partition_offsets(dst_local, n_own, n_blocks, align)


# ------------------------------------------------------------------------------------------------------------
# exchange
# ------------------------------------------------------------------------------------------------------------
def _host_staged(t, group):
    """gloo has no device all-to-all: device tensors are staged through the host (only used to run the multi-GPU code path
    on a box without several GPUs; RCCL takes the device tensors as they are)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Done:
    def wait(self):
        return True


def _all_to_all(recv, send, recv_splits, send_splits, group, async_op=False):
    """dist.all_to_all_single, host-staged under gloo.  Returns an object with wait() when async_op."""
    if not _host_staged(send, group):
        work = dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group, async_op=async_op)
        return work if async_op else None
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
    """Graph-only part of the exchange: which of my rows every peer needs, in which chunk, and where the rows I receive go.
    Pure index arithmetic + four small all-to-alls; backend-agnostic (the CPU tests run it over gloo).

    Chunks.  The halo rows of a rank are stored, and exchanged, in n_chunks chunks; the local halo order is (chunk, owner, id).
      * edge_block is None: every owner's (id-sorted) list is cut into n_chunks equal slices -- chunks of the pipelined and the
        source-bucketed schedules;
      * edge_block = int64[E], the target block of every edge (n_chunks = number of blocks): chunk c holds the remote rows whose
        FIRST use is in block c, so that the in-edges of block b only reference own rows and chunks 0..b (blocked schedule)."""

    def __init__(self, node_type_own, src_global, node_offsets, rank, world, group=None, n_chunks=1, edge_block=None, emulate=None):
        """emulate = {"node_type_global": i64[N_global]}: ONE process plays `rank` of a `world`-rank partition without a process
        group (bench.py --emulate-world: the per-rank GPU work of a multi-GPU step measured on one GPU).  The receive side is
        exact (ids, chunks, types from the global node_type); the send side mirrors it -- every peer is assumed to ask for as many
        of my rows, per chunk, as I ask of it, drawn uniformly from my rows -- and exchange_chunk replaces the all-to-all by a
        device copy of the packed rows (same bytes written, read and moved through HBM as the link transfer would)."""
        dev = src_global.device
        self.emulate = emulate
        self._bufs = {}          # (chunk, format, width) -> persistent send / wire buffers of the exchange (sizes are graph constants)
        self.rank, self.world, self.group = rank, world, group
        self.n_chunks = C = max(1, int(n_chunks))
        self.offsets = torch.as_tensor(node_offsets, dtype=torch.int64, device=dev)       # [world+1]
        lo, hi = int(self.offsets[rank]), int(self.offsets[rank + 1])
        self.n_own = hi - lo
        remote_mask = (src_global < lo) | (src_global >= hi)
        need, inverse = torch.unique(src_global[remote_mask], return_inverse=True)         # sorted, deduplicated
        self.n_halo = int(need.numel())
        owner = (torch.searchsorted(self.offsets, need, right=True) - 1).clamp(0, world - 1)      # need is sorted => owner is monotone
        owner_bounds = torch.searchsorted(need, self.offsets)
        self.recv_splits = (owner_bounds[1:] - owner_bounds[:-1]).tolist()
        # chunk of every needed row
        if self.n_halo == 0:
            chunk = torch.zeros(0, dtype=torch.int64, device=dev)
        elif edge_block is None:
            # equal slices of each owner's (id-sorted) list: row j of n is in slice floor(((j + 1) C - 1) / n)
            n_of = torch.tensor(self.recv_splits, dtype=torch.int64, device=dev)[owner]
            j = torch.arange(self.n_halo, device=dev) - owner_bounds[:-1][owner]
            chunk = torch.div((j + 1) * C - 1, n_of, rounding_mode="floor")
        else:
            if edge_block.shape != src_global.shape:
                raise ValueError("edge_block must have one entry per edge")
            chunk = torch.full((self.n_halo,), C, dtype=torch.int64, device=dev)
            chunk.scatter_reduce_(0, inverse, edge_block[remote_mask].to(torch.int64), reduce="amin", include_self=True)
            if int(chunk.min()) < 0 or int(chunk.max()) >= C:
                raise ValueError("edge_block values must lie in [0, n_chunks)")
        # ask order = (owner, chunk, id); local halo order = (chunk, owner, id)
        key_ask = owner * C + chunk
        ask_order = torch.sort(key_ask, stable=True).indices                               # position in `need` per ask slot
        counts = torch.bincount(key_ask, minlength=world * C).reshape(world, C)
        # per-peer totals, then the per-(peer, chunk) counts: both sides need them to cut the chunked all-to-alls
        tot = counts.sum(1)
        if emulate is not None:
            self.send_splits = tot.tolist()
            send_counts = counts.clone()
            g = torch.Generator(device=dev).manual_seed(977 + rank)
            asked = lo + torch.randint(0, max(self.n_own, 1), (int(tot.sum()),), generator=g, device=dev)
            halo_types_ask = emulate["node_type_global"].to(dev)[need[ask_order]]
        else:
            got = torch.empty_like(tot)
            _all_to_all(got, tot, None, None, group)
            self.send_splits = got.tolist()
            send_counts = torch.empty_like(counts)
            _all_to_all(send_counts.view(-1), counts.reshape(-1).contiguous(), [C] * world, [C] * world, group)   # [peer q][chunk] rows q wants
            # tell every owner which of its rows I need (owner, chunk, id order); receive which of my rows the peers need
            asked = _all_to_all_int64(need[ask_order].contiguous(), self.recv_splits, self.send_splits, group)
            if asked.numel() and (int(asked.min()) < lo or int(asked.max()) >= hi):      # (setup time: one synchronisation)
                raise RuntimeError("pyhgt_amd.dist: rank %d was asked for rows outside its range [%d, %d): the ranks do not agree on "
                                   "node_offsets (every rank must pass the same partition)" % (rank, lo, hi))
            # node types of my halo rows (owners answer in the order I asked)
            types_for_peers = node_type_own[(asked - lo)]
            halo_types_ask = _all_to_all_int64(types_for_peers.contiguous(), self.send_splits, self.recv_splits, group)
        send_rows = (asked - lo).to(torch.int32)                                           # local row ids, (peer, chunk, id) order

        def chunk_major(cnt):
            """(peer, chunk, id) -> (chunk, peer, id): gather order + per-chunk split sizes, from a [world, C] count matrix."""
            cnt_l = cnt.tolist()
            starts = [[0] * C for _ in range(world)]
            run = 0
            for p in range(world):
                for c in range(C):
                    starts[p][c] = run
                    run += cnt_l[p][c]
            pieces, per_chunk = [], []
            for c in range(C):
                sizes = []
                for p in range(world):
                    pieces.append(torch.arange(starts[p][c], starts[p][c] + cnt_l[p][c], device=dev))
                    sizes.append(cnt_l[p][c])
                per_chunk.append(sizes)
            order = torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.int64, device=dev)
            return order, per_chunk
        recv_order, self.recv_chunk_splits = chunk_major(counts)               # new halo position -> ask slot
        send_order, self.send_chunk_splits = chunk_major(send_counts)
        self.send_rows = send_rows[send_order].contiguous()
        self.halo_types = halo_types_ask[recv_order].contiguous()
        halo_order = ask_order[recv_order]                                                 # halo row i holds global node need[halo_order[i]]
        inv = torch.empty_like(halo_order)
        inv[halo_order] = torch.arange(halo_order.numel(), device=dev)                      # position in `need` -> halo position
        # local id of every edge source: own rows first, then halo rows (chunk, owner, id order)
        pos = torch.zeros_like(src_global)
        if self.n_halo > 0:
            pos[remote_mask] = inv[inverse]
        self.src_local = torch.where(remote_mask, self.n_own + pos, src_global - lo)
        self.node_type_local = torch.cat([node_type_own, self.halo_types])
        self.n_local = self.n_own + self.n_halo
        self.halo_order = halo_order
        self.halo_chunk = chunk[halo_order]                                                # chunk of every halo row (non-decreasing)
        self.need = need
        # per chunk: offsets into the send list / the halo rows
        self.send_chunk_off = [0]
        self.recv_chunk_off = [0]
        for c in range(C):
            self.send_chunk_off.append(self.send_chunk_off[-1] + sum(self.send_chunk_splits[c]))
            self.recv_chunk_off.append(self.recv_chunk_off[-1] + sum(self.recv_chunk_splits[c]))
        # chunks in which NO rank sends or receives anything are skipped by every rank (the collective is not even entered)
        vol = torch.tensor([sum(self.recv_chunk_splits[c]) + sum(self.send_chunk_splits[c]) for c in range(C)], dtype=torch.int64,
                           device=dev)
        if world > 1 and emulate is None:
            vol = vol.cpu() if _host_staged(vol, group) else vol
            dist.all_reduce(vol, group=group)
        self.chunk_live = [bool(v > 0) for v in vol.tolist()]

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

    def _transfer(self, recv, send, c, async_op):
        """The all-to-all of chunk c (or, emulating, a device copy of as many bytes)."""
        if getattr(self, "emulate", None) is not None:
            n = min(recv.size(0), send.size(0))
            if n:
                # an elementwise KERNEL on the compute stream, not Tensor.copy_: the runtime hands large device-to-device copies to
                # the copy engines, and the compute queue then idles ~0.6 ms per chunk on the cross-engine hand-off (measured with
                # rocprofv3: 4.8 ms of idle per step that no real run has -- RCCL moves the rows on its own stream)
                a, b = recv[:n].reshape(-1), send[:n].reshape(-1)
                if a.dtype == torch.uint8 and a.numel() % 4 == 0:
                    a, b = a.view(torch.int32), b.view(torch.int32)
                torch.bitwise_or(b, 0, out=a) if a.dtype in (torch.int32, torch.uint8) else torch.add(b, 0, out=a)
            return _Done() if async_op else None
        if not self.chunk_live[c]:
            return None
        return _all_to_all(recv, send, list(self.recv_chunk_splits[c]), list(self.send_chunk_splits[c]), self.group, async_op=async_op)

    def exchange_chunk(self, c, x_own, x_local, pack=None, async_op=False, compress=False, expand=True):
        """One chunk of the exchange: pack the rows of chunk c the peers need, all-to-all them into the halo rows of chunk c.
        Returns (work, buffers) when async_op: work.wait() makes the current stream wait for the rows (and, with compress and
        expand, expands them into x_local).  compress: rows travel in the 24-bit format of hgt_gather_rows_c24 (3/4 of the bytes;
        relative error <= 2^-16 on halo features, which only feed the K/V projections); expand=False leaves them in the wire
        buffer (buffers[1], [rows, 3 d] uint8) for a projection that reads the wire format directly.
        A chunk in which no rank of the group moves a row is skipped by every rank (chunk_live); a rank whose own part is empty
        still enters the collective with zero-length splits."""
        d = x_own.size(1)
        rows = self.send_rows[self.send_chunk_off[c]:self.send_chunk_off[c + 1]]
        recv = x_local[self.n_own + self.recv_chunk_off[c]:self.n_own + self.recv_chunk_off[c + 1]]
        lib = _lib.load() if x_own.is_cuda else None
        st = torch.cuda.current_stream().cuda_stream if x_own.is_cuda else None
        # the pack / wire buffers are kept across steps: their sizes depend on the graph only, and allocating them per step costs
        # milliseconds once the caching allocator has to wait for record_stream'd blocks of the previous step (measured: 4 - 13 ms
        # of "pack" per step with 8 - 16 chunks).  Reuse is safe: the compute stream waited for chunk c's transfer of step i before
        # it projected the chunk, so both buffers are idle again when step i + 1 packs into them.
        def persistent(tag, shape, dtype):
            key = (c, tag, d, str(x_own.device))
            buf = self.__dict__.setdefault("_bufs", {}).get(key)
            if buf is None or buf.shape != torch.Size(shape) or buf.dtype != dtype:
                buf = self._bufs[key] = torch.empty(shape, dtype=dtype, device=x_own.device)
            return buf
        if compress and pack is None and x_own.is_cuda and d % 4 == 0:
            send = persistent("c24s", (rows.numel(), 3 * d), torch.uint8)
            if rows.numel():
                _lib.check(lib.hgt_gather_rows_c24(x_own.data_ptr(), x_own.stride(0), rows.data_ptr(), rows.numel(), d,
                                                   send.data_ptr(), st), "hgt_gather_rows_c24")
            wire = persistent("c24r", (recv.size(0), 3 * d), torch.uint8)
            work = self._transfer(wire, send, c, async_op)
            n_recv, ld = recv.size(0), x_local.stride(0)

            def expand_rows():
                if expand and n_recv:
                    _lib.check(lib.hgt_unpack_rows_c24(wire.data_ptr(), n_recv, d, recv.data_ptr(), ld,
                                                       torch.cuda.current_stream().cuda_stream), "hgt_unpack_rows_c24")

            class _Expanding:
                def wait(self):
                    if work is not None:
                        work.wait()
                    expand_rows()
                    return True
            if async_op:
                return _Expanding(), (send, wire)
            expand_rows()
            return None
        if pack is None:
            if not x_own.is_cuda:
                raise RuntimeError("pyhgt_amd.dist: halo packing runs the HIP gather kernel; CPU tensors need an explicit pack fn")
            send = persistent("f32s", (rows.numel(), d), x_own.dtype)
            if rows.numel():
                _lib.check(lib.hgt_gather_rows(x_own.data_ptr(), x_own.stride(0), rows.data_ptr(), rows.numel(), d,
                                               send.data_ptr(), st), "hgt_gather_rows")
        else:
            send = pack(x_own, rows)
        work = self._transfer(recv, send, c, async_op)
        if async_op:
            return (work if work is not None else _Done()), (send,)
        return None

    def exchange(self, x_own, x_local, pack=None):
        """Fill x_local[n_own:] with the halo rows (x_local[:n_own] must already hold x_own), one chunk after the other.
        `pack(x_own, rows_int32) -> [len(rows), d]`; defaults to the HIP gather kernel on GPU."""
        for c in range(self.n_chunks):
            self.exchange_chunk(c, x_own, x_local, pack=pack)
        return x_local


MODES = ("blocked", "bucketed", "pipelined")


class PartitionedGraph:
    """One rank's share of a destination-partitioned typed graph + the per-layer forward (module docstring: schedules)."""

    def __init__(self, node_type_own, src_global, dst_local, edge_type, edge_time, num_types, num_relations,
                 nodes_per_rank, rank, world, group=None, node_offsets=None, n_chunks=None, halo=None, compress=False, bucketed=None,
                 mode=None, overlap_blocks=False, block_shape="equal"):
        """mode: "blocked" (default) / "bucketed" / "pipelined"; bucketed=True/False is the round-2 spelling of the last two.
        n_chunks: halo chunks = target blocks of the blocked schedule (default 8), equal slices otherwise (default 4).
        halo: a prebuilt HaloPlan for this rank (tests build it on CPU over gloo and move it to the device with HaloPlan.to);
        otherwise it is negotiated here with four small all-to-alls.  A layer the chosen schedule does not cover (exact fp32
        precision, DenseHGTConv, padded rows wider than 256 columns) runs the pipelined schedule."""
        from .conv import GraphPlan
        if mode is None:
            mode = "blocked" if bucketed is None else ("bucketed" if bucketed else "pipelined")
        if mode not in MODES:
            raise ValueError("mode must be one of %s" % (MODES,))
        if node_offsets is None:
            node_offsets = [nodes_per_rank * r for r in range(world + 1)]
        n_own = int(node_offsets[rank + 1]) - int(node_offsets[rank])
        if n_chunks is None:
            n_chunks = 8 if mode == "blocked" else 4
        self.mode = mode
        # overlap_blocks (experiment, OFF): the edge phases of consecutive target blocks on two side streams (block b on stream b % 2)
        # while the main stream keeps projecting the next halo chunks.  Blocks are independent (disjoint targets, logits positions,
        # pending entries, hub slots), and the 1/B-size launches have long grid tails (8 blocks: 6.5 ms of edge kernels against
        # 5.6 ms for the one-call layer) -- but measured (round 4, emulated rank of 8): 23.5 ms per step instead of 15.9: the
        # PERSISTENT projection kernel (one workgroup per CU, static tile partition) next to the edge kernels leaves half of its
        # workgroups waiting for a CU while the rest finish early.  Kept for a projection kernel that can share a CU.
        self.overlap_blocks = bool(overlap_blocks)
        self._side = None
        self.block_bounds = None
        edge_block = None
        if mode == "blocked":
            self.block_bounds = target_blocks(dst_local, n_own, n_chunks, shape=block_shape)
            edge_block = torch.searchsorted(torch.tensor(self.block_bounds[1:], dtype=torch.int64, device=dst_local.device),
                                            dst_local, right=True).clamp(max=n_chunks - 1)
        if halo is not None:
            self.halo = halo
            if mode == "blocked" and halo.n_chunks != n_chunks:
                raise ValueError("a prebuilt HaloPlan for the blocked schedule must be built with edge_block / n_chunks = blocks")
            if mode == "blocked" and halo.n_halo > 0:
                # ... and with THESE blocks: block b's in-edges may only reference own rows and halo chunks 0..b (the K|V of a later
                # chunk are not projected yet when the block runs -- the kernels would read stale workspace rows without any error)
                # (one pass over the edges and ONE host synchronisation per PartitionedGraph built on a prebuilt plan; everything on the
                #  plan's device -- a plan built on another device than dst_local is compared there, not a device-mismatch RuntimeError)
                hdev = halo.src_local.device
                off = torch.tensor(list(halo.recv_chunk_off[1:]), dtype=torch.int64, device=hdev)
                src_chunk = torch.searchsorted(off, (halo.src_local.to(torch.int64) - halo.n_own).clamp(min=0), right=True)
                late = (halo.src_local >= halo.n_own) & (src_chunk > edge_block.to(hdev))
                if bool(late.any()):
                    raise ValueError("the prebuilt HaloPlan was built for other target blocks (edge_block / block_shape / alignment): "
                                     "%d edges reference a halo chunk later than their own block" % int(late.sum()))
        else:
            self.halo = HaloPlan(node_type_own, src_global, node_offsets, rank, world, group, n_chunks=n_chunks, edge_block=edge_block)
        self.compress = bool(compress)     # 24-bit halo rows on the links (exchange_chunk); off: exact fp32 rows
        C = self.halo.n_chunks
        self.n_buckets = C + 1
        if mode == "bucketed" and not self.n_buckets * num_relations < 64:      # the streaming walk keeps one range per relation id in a lane
            raise ValueError("bucketed edge phase needs (n_chunks + 1) * num_relations < 64")
        self.bucketed = (mode == "bucketed")
        self.chunk_lists = self.halo.chunk_row_lists(num_types) if (C > 1 or mode != "pipelined") else None
        self.n_own, self.n_local = self.halo.n_own, self.halo.n_local
        self.edge_index = torch.stack([self.halo.src_local, dst_local], dim=0).contiguous()
        self.edge_type, self.edge_time = edge_type, edge_time
        self.node_type_local = self.halo.node_type_local
        self.num_relations = num_relations
        self.plan = GraphPlan(self.node_type_local, self.edge_index, edge_type, edge_time, num_types, num_relations,
                              n_q_rows=self.n_own)
        self.bucket_plan = None
        self.blocks = None
        if mode == "bucketed":
            # bucket of an edge = where its source row comes from: 0 own, 1 + c halo chunk c (halo rows are in chunk order)
            self.edge_type_bucketed = self.halo.bucketed_edge_types(edge_type, num_relations)
            self.bucket_plan = GraphPlan(self.node_type_local, self.edge_index, self.edge_type_bucketed, edge_time, num_types,
                                         self.n_buckets * num_relations, n_q_rows=self.n_own)
        if mode == "blocked":
            # (q_begin, q_end, item_begin, item_end) of every target block: the plan's items are ordered by destination tile
            tab, tile = self.plan.tile_items()
            if tile != PLAN_TILE:
                raise RuntimeError("libhgt_hip was built with a plan tile of %d targets, pyhgt_amd.dist assumes %d" % (tile, PLAN_TILE))
            self.blocks = []
            for b in range(C):
                q0, q1 = self.block_bounds[b], self.block_bounds[b + 1]
                t0, t1 = q0 // tile, (q1 + tile - 1) // tile
                self.blocks.append((q0, q1, int(tab[t0]), int(tab[t1])))
        self.x_local = None
        self.workspace = None      # owned here: Q/K/V stay in it between the stages of one step
        self.timeline = None       # set to a list: forward() appends (label, torch.cuda.Event) marks on the compute stream (bench.py)

    def _mark(self, label):
        if self.timeline is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.timeline.append((label, ev))

    def layer_mode(self, layer):
        """The schedule this layer runs (a layer outside a schedule's coverage takes the pipelined one)."""
        split = getattr(layer, "precision", None) in ("bf16x3", "f16x3")
        hgt = getattr(layer, "_UPDATE_MODE", 0) == 0
        if self.mode == "blocked" and split and hgt and not getattr(layer, "keep_att", False):
            lay = _lib.layout_for(layer.out_dim, layer.n_heads)
            # (everything hgt_conv_forward stage 5 / the fused matrix-core aggregation require: a layer outside it would fail with
            #  HGT_ERR_UNSUPPORTED after stages 1 and 2 and the all-to-alls are in flight)
            if (lay.d_pad <= 256 and layer.out_dim % 4 == 0 and layer.in_dim % 4 == 0 and self.num_relations < 64 and
                    not (layer.kernel_flags & _lib.HGT_FLAG_VALU_AGGREGATE)):
                return "blocked"
        if self.mode == "bucketed" and split and hgt:
            return "bucketed"
        return "pipelined"

    def forward(self, layer, x_own, phase_events=None):
        d = x_own.size(1)
        mode = self.layer_mode(layer)
        bucketed = (mode == "bucketed")
        need = layer.workspace_bytes(self.n_local, self.plan.E, self.n_buckets if bucketed else 1)
        if self.workspace is None or self.workspace.numel() < need or self.workspace.device != x_own.device:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=x_own.device)
        if self.x_local is None or self.x_local.size(1) != d:
            self.x_local = torch.empty(self.n_local, d, dtype=x_own.dtype, device=x_own.device)
        if x_own.data_ptr() != self.x_local.data_ptr():
            self.x_local[:self.n_own].copy_(x_own)
        x_own_v = self.x_local[:self.n_own]
        C = self.halo.n_chunks
        cur = torch.cuda.current_stream
        if self.chunk_lists is None:
            for c in range(C):
                self.halo.exchange_chunk(c, x_own_v, self.x_local, compress=self.compress)
            return layer(self.x_local, self.node_type_local, self.edge_index, self.edge_type, self.edge_time,
                         plan=self.plan, n_q_rows=self.n_own, phase_events=phase_events, workspace=self.workspace)
        if mode == "blocked":
            args = (self.x_local, self.node_type_local, self.edge_index, self.edge_type, self.edge_time)
            kw = dict(plan=self.plan, n_q_rows=self.n_own, workspace=self.workspace)
            # the halo rows are projected straight off the 24-bit wire buffer where the projection kernel can read it
            direct = self.compress and d % 4 == 0 and d <= 256
            self._mark("start")
            # every chunk is packed and queued on the links up front: the transfers run back to back on RCCL's stream
            pending = [self.halo.exchange_chunk(c, x_own_v, self.x_local, async_op=True, compress=self.compress, expand=not direct)
                       for c in range(C)]
            self._mark("pack")
            layer(*args, stage=1, phase_events=phase_events, **kw)                         # Q|K|V of the own rows
            self._mark("own_qkv")
            out = torch.empty(self.n_own, layer.out_dim, dtype=torch.float32, device=x_own.device)
            overlap = self.overlap_blocks and C > 1 and x_own.is_cuda
            main = cur()
            if overlap and self._side is None:
                self._side = [torch.cuda.Stream(device=x_own.device), torch.cuda.Stream(device=x_own.device)]
            for b in range(C):
                work, bufs = pending[b]
                work.wait()
                for t in bufs:
                    t.record_stream(cur())
                self._mark("wait")
                rows, off = self.chunk_lists[b]
                if rows.numel():
                    c24 = (bufs[1], self.n_own + self.halo.recv_chunk_off[b]) if direct else None
                    layer(*args, stage=2, proj=(rows, off), proj_c24=c24, **kw)            # K|V of the halo rows of chunk b
                self._mark("halo_kv")
                if overlap:      # block b on a side stream, behind everything the main stream has enqueued so far (K|V of chunks <= b)
                    side = self._side[b % 2]
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        layer(*args, stage=5, block=self.blocks[b], out=out, **kw)
                else:
                    layer(*args, stage=5, block=self.blocks[b], out=out, phase_events=phase_events if b == C - 1 else None, **kw)
                    self._mark("edge_blocks")
            if overlap:
                main.wait_stream(self._side[0])
                main.wait_stream(self._side[1])
                self._mark("edge_blocks")     # (with overlap: what the edge phases add BEHIND the last projection)
            layer._prepared_valid = True      # (stage 5 is not a "final" call of HGTConv.forward: the images were all written)
            return out
        if bucketed:
            S = self.n_buckets
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
                    b.record_stream(cur())
                layer(*args, stage=2, proj=self.chunk_lists[c], slices=(0, S), **kw)   # K|V of the halo rows of chunk c
                out = layer(*args, stage=4, slices=(c + 1, S), phase_events=phase_events if c == C - 1 else None, **kw)
            return out
        # pipelined: chunk c+1 is packed and put on the links while chunk c's halo rows are projected
        args = (self.x_local, self.node_type_local, self.edge_index, self.edge_type, self.edge_time)
        kw = dict(plan=self.plan, n_q_rows=self.n_own, workspace=self.workspace)
        pending = [self.halo.exchange_chunk(0, x_own_v, self.x_local, async_op=True, compress=self.compress)]
        layer(*args, stage=1, phase_events=phase_events, **kw)            # Q|K|V of the own rows
        for c in range(C):
            if c + 1 < C:
                pending.append(self.halo.exchange_chunk(c + 1, x_own_v, self.x_local, async_op=True, compress=self.compress))
            work, bufs = pending[c]
            work.wait()                                                   # current stream waits for chunk c (and expands it)
            for b in bufs:
                b.record_stream(cur())
            layer(*args, stage=2, proj=self.chunk_lists[c], **kw)         # K|V of the halo rows of chunk c
        return layer(*args, stage=3, phase_events=phase_events, **kw)
