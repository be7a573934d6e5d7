"""world_size-2/3 CPU tests (gloo) of the destination partition + halo exchange (pyhgt_amd/dist.py).
The exchange logic is backend-agnostic; the layer compute of each rank is done here by the CPU
oracle (tests may use it), and the stitched result must equal the oracle on the whole graph."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import hgt_oracle as O
from pyhgt_amd.synth import synthetic_typed_graph


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, E, d, T, R, H, offsets, tmpdir, n_chunks=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyhgt_amd.dist import HaloPlan
        torch.set_num_threads(2)
        x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=77, sorted_types=False)
        sd = O.make_state_dict(d, d, T, R, H, True, True, seed=78)
        lo, hi = offsets[rank], offsets[rank + 1]
        mine = (ei[1] >= lo) & (ei[1] < hi)                       # a rank owns ALL in-edges of its targets
        src_g, dst_l = ei[0][mine], ei[1][mine] - lo
        hp = HaloPlan(nt[lo:hi], src_g, offsets, rank, world, n_chunks=n_chunks)
        # structural checks
        assert hp.n_own == hi - lo and hp.n_local == hp.n_own + hp.n_halo
        assert sum(hp.recv_splits) == hp.n_halo and sum(hp.send_splits) == hp.send_rows.numel()
        x_local = torch.empty(hp.n_local, d)
        x_local[:hp.n_own] = x[lo:hi]
        hp.exchange(x[lo:hi], x_local, pack=lambda xo, rows: xo.index_select(0, rows.long()))
        # halo rows are exactly the remote sources, deduplicated, with their features and types
        remote = torch.unique(src_g[(src_g < lo) | (src_g >= hi)])
        assert torch.equal(hp.need, remote)
        assert torch.equal(torch.sort(hp.halo_order).values, torch.arange(remote.numel()))     # a permutation
        if n_chunks == 1:
            assert torch.equal(hp.halo_order, torch.arange(remote.numel()))
        remote = remote[hp.halo_order]                            # (chunk, peer, id) order of the halo rows
        assert torch.equal(x_local[hp.n_own:], x[remote])
        assert torch.equal(hp.node_type_local[hp.n_own:], nt[remote])
        assert torch.equal(x_local[hp.src_local], x[src_g])
        # stage-2 row lists: every halo row of a valid type exactly once, grouped by type, chunk by chunk
        seen = []
        for c, (rows, off) in enumerate(hp.chunk_row_lists(T)):
            a, b = hp.n_own + hp.recv_chunk_off[c], hp.n_own + hp.recv_chunk_off[c + 1]
            assert off[0] == 0 and off[-1] == rows.numel() and ((rows >= a) & (rows < b)).all()
            for t in range(T):
                assert (hp.node_type_local[rows[off[t]:off[t + 1]].long()] == t).all()
            seen.append(rows.long())
        # source buckets of the bucketed edge phase: 0 <=> own source, 1 + c <=> the source row arrives with chunk c
        bucket = hp.edge_buckets()
        assert bucket.shape == hp.src_local.shape and int(bucket.min()) >= 0 and int(bucket.max()) <= n_chunks
        assert torch.equal(bucket == 0, hp.src_local < hp.n_own)
        for c in range(n_chunks):
            a, b = hp.n_own + hp.recv_chunk_off[c], hp.n_own + hp.recv_chunk_off[c + 1]
            assert torch.equal(bucket == c + 1, (hp.src_local >= a) & (hp.src_local < b))
        # relation ids of the bucketed plan: bucket * R + relation, unclaimed relation ids -> (n_chunks + 1) * R; and the
        # oracle on the re-numbered graph (relation parameters repeated per bucket) equals the oracle on the original one
        et_m = et[mine].clone()
        et_m[::9] = R + 3
        et_b = hp.bucketed_edge_types(et_m, R)
        ok = (et_m >= 0) & (et_m < R)
        assert torch.equal(et_b[ok], bucket[ok] * R + et_m[ok]) and (et_b[~ok] == (n_chunks + 1) * R).all()
        B = n_chunks + 1
        sd_b = dict(sd)
        for k in ("relation_att", "relation_msg"):
            sd_b[k] = sd[k].repeat(B, 1, 1, 1)
        sd_b["relation_pri"] = sd["relation_pri"].repeat(B, 1)
        out_b = O.forward_closed_form(sd_b, T, B * R, H, x_local, hp.node_type_local, torch.stack([hp.src_local, dst_l]), et_b, tm[mine])
        out_o = O.forward_closed_form(sd, T, R, H, x_local, hp.node_type_local, torch.stack([hp.src_local, dst_l]), et_m, tm[mine])
        assert (out_b - out_o).abs().max().item() < 1e-12
        seen = torch.cat(seen)
        valid = (hp.node_type_local[hp.n_own:] >= 0) & (hp.node_type_local[hp.n_own:] < T)
        assert torch.equal(torch.sort(seen).values, hp.n_own + valid.nonzero(as_tuple=True)[0])
        # local layer (oracle) on [own; halo] == rows [lo,hi) of the global layer
        ei_local = torch.stack([hp.src_local, dst_l])
        out_local = O.forward_closed_form(sd, T, R, H, x_local, hp.node_type_local, ei_local, et[mine], tm[mine])
        out_global = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm)
        err = (out_local[:hp.n_own] - out_global[lo:hi]).abs().max().item()
        assert err < 1e-10, err
        torch.save(torch.tensor([hp.n_halo, err]), os.path.join(tmpdir, "ok%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,offsets,n_chunks", [(2, [0, 300, 600], 1), (3, [0, 150, 380, 600], 1), (3, [0, 150, 380, 600], 4),
                                                    (2, [0, 300, 600], 7)])
def test_partitioned_forward_equals_global(world, offsets, n_chunks, tmp_path):
    N, E, d, T, R, H = 600, 5000, 32, 3, 4, 4
    port = _free_port()
    mp.spawn(_worker, args=(world, port, N, E, d, T, R, H, offsets, str(tmp_path), n_chunks), nprocs=world, join=True)
    for r in range(world):
        assert os.path.isfile(os.path.join(str(tmp_path), "ok%d.pt" % r))


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the in-edge-balanced partitioner and the target-blocked schedule (first-use halo chunks), pyhgt_amd/dist.py
# ---------------------------------------------------------------------------------------------------------------------
def _zipf_targets(N, E, a, g, permute):
    u = torch.rand(E, generator=g)
    dst = (N * u ** (1.0 / (1.0 - a))).long().clamp(0, N - 1)
    return torch.randperm(N, generator=g)[dst] if permute else dst


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partitioner_balances_in_edges_on_a_zipf_graph(world):
    """contiguous dst ranges balanced by in-edge count.  Zipf(0.5) in-degrees over randomly numbered nodes: every rank's
    edge count within 2 % of E / world, cut points on plan tiles; degree-SORTED Zipf(0.8) ids (bench.py's hub generator: a quarter
    of all edges in the first tile): as balanced as whole tiles allow."""
    from pyhgt_amd.dist import partition_offsets, PLAN_TILE
    g = torch.Generator().manual_seed(5)
    N, E = 200_000, 2_000_000
    dst = _zipf_targets(N, E, 0.5, g, permute=True)
    offs = partition_offsets(dst, N, world)
    assert offs[0] == 0 and offs[-1] == N and all(o % PLAN_TILE == 0 for o in offs[:-1]) and offs == sorted(offs)
    counts = [int(((dst >= offs[r]) & (dst < offs[r + 1])).sum()) for r in range(world)]
    assert sum(counts) == E
    assert max(abs(c - E / world) for c in counts) <= 0.02 * E / world, counts
    # the uniform recipe of bench.py: equal node ranges come out (to one tile)
    dst_u = torch.randint(0, N, (E,), generator=g)
    offs_u = partition_offsets(dst_u, N, world)
    assert max(abs(offs_u[r] - r * N // world) for r in range(world + 1)) <= 2 * PLAN_TILE
    # hubs first: a cut can only move by whole tiles, so the bound is the heaviest tile next to a cut
    dst_h = _zipf_targets(N, E, 0.8, g, permute=False)
    offs_h = partition_offsets(dst_h, N, world)
    tile_deg = torch.bincount(dst_h // PLAN_TILE)
    counts_h = [int(((dst_h >= offs_h[r]) & (dst_h < offs_h[r + 1])).sum()) for r in range(world)]
    assert sum(counts_h) == E and max(abs(c - E / world) for c in counts_h) <= int(tile_deg.max())


def test_partition_shares_cover_the_graph_exactly_once():
    from pyhgt_amd.dist import partition
    x, nt, ei, et, tm = synthetic_typed_graph(3000, 20000, 8, 3, 4, seed=3, sorted_types=False)
    world, seen = 5, []
    for r in range(world):
        sh = partition(nt, ei, et, tm, world, r, align=64)
        lo, hi = sh["node_offsets"][r], sh["node_offsets"][r + 1]
        assert torch.equal(sh["node_type_own"], nt[lo:hi])
        assert torch.equal(sh["src_global"], ei[0][sh["edge_ids"]]) and torch.equal(sh["dst_local"] + lo, ei[1][sh["edge_ids"]])
        assert torch.equal(sh["edge_type"], et[sh["edge_ids"]]) and torch.equal(sh["edge_time"], tm[sh["edge_ids"]])
        assert int(sh["dst_local"].min()) >= 0 and int(sh["dst_local"].max()) < hi - lo
        seen.append(sh["edge_ids"])
    assert torch.equal(torch.sort(torch.cat(seen)).values, torch.arange(ei.size(1)))
    # degenerate inputs: no edges, more ranks than tiles
    empty = partition(nt, ei[:, :0], et[:0], tm[:0], 3, 1, align=64)
    assert empty["src_global"].numel() == 0 and empty["node_offsets"][-1] == 3000


def _blocked_graph(case, N, E, d, T, R, world):
    """Global graph of the blocked-schedule tests.  case "uniform": sources anywhere.  "island": the LAST rank's targets only
    have sources inside its own range (a rank with no halo at all).  "front": remote sources only point at the first targets
    of every rank (later blocks introduce no new halo row: dead chunks on every rank)."""
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=91, sorted_types=False)
    per = N // world
    src, dst = ei[0].clone(), ei[1].clone()
    if case == "island":
        last = dst >= (world - 1) * per
        src[last] = (world - 1) * per + src[last] % (N - (world - 1) * per)
    if case == "front":
        owner_d, owner_s = (dst // per).clamp(max=world - 1), (src // per).clamp(max=world - 1)
        remote = owner_d != owner_s
        dst[remote] = owner_d[remote] * per + dst[remote] % 16
    return x, nt, torch.stack([src, dst]), et, tm


def _worker_blocked(rank, world, port, case, n_blocks, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyhgt_amd.dist import HaloPlan, partition, target_blocks
        torch.set_num_threads(2)
        N, E, d, T, R, H = 960, 9000, 16, 3, 4, 4
        x, nt, ei, et, tm = _blocked_graph(case, N, E, d, T, R, world)
        sd = O.make_state_dict(d, d, T, R, H, True, True, seed=92)
        offsets = [r * (N // world) for r in range(world)] + [N]
        sh = partition(nt, ei, et, tm, world, rank, node_offsets=offsets)
        lo, hi = offsets[rank], offsets[rank + 1]
        src_g, dst_l = sh["src_global"], sh["dst_local"]
        bounds = target_blocks(dst_l, hi - lo, n_blocks, align=16)
        assert bounds[0] == 0 and bounds[-1] == hi - lo and bounds == sorted(bounds)
        eblock = torch.searchsorted(torch.tensor(bounds[1:]), dst_l, right=True).clamp(max=n_blocks - 1)
        hp = HaloPlan(nt[lo:hi], src_g, offsets, rank, world, n_chunks=n_blocks, edge_block=eblock)
        remote_mask = (src_g < lo) | (src_g >= hi)
        remote = torch.unique(src_g[remote_mask])
        assert torch.equal(hp.need, remote) and hp.n_halo == remote.numel()
        if case == "island" and rank == world - 1:
            assert hp.n_halo == 0 and sum(hp.recv_splits) == 0
        # halo rows are ordered by chunk; chunk of a row = the FIRST block that uses it
        assert torch.equal(torch.sort(hp.halo_order).values, torch.arange(remote.numel()))
        assert (hp.halo_chunk[1:] >= hp.halo_chunk[:-1]).all()
        src_chunk = torch.full_like(src_g, -1)
        halo_pos = hp.src_local - hp.n_own
        src_chunk[remote_mask] = hp.halo_chunk[halo_pos[remote_mask]]
        assert (src_chunk <= eblock).all()                                    # block b only needs chunks 0..b
        for c in range(n_blocks):                                             # ... and every row of chunk c IS needed by block c
            rows_c = torch.arange(hp.recv_chunk_off[c], hp.recv_chunk_off[c + 1])
            used = torch.unique(halo_pos[remote_mask & (eblock == c)])
            assert torch.isin(rows_c, used).all()
        for c in range(n_blocks):                                             # a chunk with rows here is live for everybody
            if hp.recv_chunk_off[c + 1] > hp.recv_chunk_off[c] or hp.send_chunk_off[c + 1] > hp.send_chunk_off[c]:
                assert hp.chunk_live[c]
        if case == "front":      # all remote edges point into ONE block per rank: at most `world` live chunks, the rest is skipped
            assert 1 <= sum(hp.chunk_live) <= min(world, n_blocks) and (n_blocks <= world or not all(hp.chunk_live)), hp.chunk_live
        # the chunked exchange delivers features and types (a rank without halo still enters every live collective)
        x_local = torch.full((hp.n_local, d), float("nan"))
        x_local[:hp.n_own] = x[lo:hi]
        ei_local = torch.stack([hp.src_local, dst_l])
        out_global = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm)
        worst = 0.0
        for b in range(n_blocks):
            hp.exchange_chunk(b, x[lo:hi], x_local, pack=lambda xo, rows: xo.index_select(0, rows.long()))
            # block b with ONLY chunks 0..b delivered (later halo rows are still NaN): the oracle on the block's in-edges
            eb = eblock == b
            q0, q1 = bounds[b], bounds[b + 1]
            if q1 > q0:
                out_b = O.forward_closed_form(sd, T, R, H, x_local, hp.node_type_local, ei_local[:, eb], sh["edge_type"][eb],
                                              sh["edge_time"][eb])
                assert torch.isfinite(out_b[q0:q1]).all()
                worst = max(worst, (out_b[q0:q1] - out_global[lo + q0:lo + q1]).abs().max().item())
        halo_ids = remote[hp.halo_order]
        assert torch.equal(x_local[hp.n_own:], x[halo_ids]) and torch.equal(hp.node_type_local[hp.n_own:], nt[halo_ids])
        assert worst < 1e-10, worst
        # stage-2 row lists cover every halo row of a valid type once, chunk by chunk
        rows_all = torch.cat([rows.long() for rows, _ in hp.chunk_row_lists(T)]) if n_blocks else torch.zeros(0, dtype=torch.long)
        valid = (hp.node_type_local[hp.n_own:] >= 0) & (hp.node_type_local[hp.n_own:] < T)
        assert torch.equal(torch.sort(rows_all).values, hp.n_own + valid.nonzero(as_tuple=True)[0])
        torch.save(torch.tensor([hp.n_halo, worst]), os.path.join(tmpdir, "ok%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case,n_blocks", [(2, "uniform", 3), (3, "uniform", 5), (4, "island", 4), (2, "front", 3), (4, "front", 8)])
def test_target_blocked_halo_chunks_over_gloo(world, case, n_blocks, tmp_path):
    """First-use halo chunks: block b of every rank is computable once chunks 0..b have arrived; ranks without halo rows and chunks
    no rank has rows in (zero-length splits / skipped collectives) are handled; stitched outputs equal the global oracle."""
    port = _free_port()
    mp.spawn(_worker_blocked, args=(world, port, case, n_blocks, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.isfile(os.path.join(str(tmp_path), "ok%d.pt" % r))
