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
