"""the device-side hand-off of sampled batches.  CPU part: the synthetic sampler output has the layout
facts of the reference pipeline (appendix C) and `to_torch_layout` reproduces the reference's own `to_torch` bit for bit."""
import numpy as np
import pytest
import torch

from oracle.reference_loader import reference_available, load_reference_data
from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout

needs_ref = pytest.mark.skipif(not reference_available(), reason="reference tree not present (GPU box)")


@pytest.mark.parametrize("schema,T,R", [("mag", 4, 9), ("oag", 5, 33)])
def test_synthetic_batch_has_the_layout_of_the_reference_pipeline(schema, T, R):
    feature, time, edge_list, graph = synthetic_sampled_batch(schema, n_seed=32, width=24, depth=3, feat_dim=16, seed=1)
    x, nt, tm, ei, et, node_dict, edge_dict = to_torch_layout(feature, time, edge_list, graph)
    assert len(graph.get_types()) == T and len(edge_dict) == R and edge_dict["self"] == R - 1      # data.py:237-238
    assert torch.equal(nt, nt.sort().values)                                  # type-contiguous, ascending (data.py:227-235)
    assert ei.shape[0] == 2 and ei.stride() == (1, 2)                         # the .t() view of data.py:254
    assert int(tm.min()) >= 111 and int(tm.max()) <= 129                      # year differences within +-9 (appendix C)
    deg = torch.bincount(ei[1], minlength=nt.numel())
    assert int(deg.min()) >= 1                                                # every node has its self loop
    # runs: maximal stretches of one (relation, target type); targets ascend inside a run; a type's `self` run comes first
    key = et * T + nt[ei[1]]
    starts = torch.cat([torch.tensor([0]), (key[1:] != key[:-1]).nonzero().flatten() + 1, torch.tensor([key.numel()])])
    seen_types = set()
    for a, b in zip(starts[:-1].tolist(), starts[1:].tolist()):
        tgt = ei[1, a:b]
        assert torch.all(tgt[1:] >= tgt[:-1])
        t = int(nt[tgt[0]])
        if t not in seen_types:
            assert int(et[a]) == edge_dict["self"]
            seen_types.add(t)


@needs_ref
@pytest.mark.parametrize("schema", ["mag", "oag"])
def test_to_torch_layout_equals_the_reference_to_torch(schema):
    data = load_reference_data()
    feature, time, edge_list, graph = synthetic_sampled_batch(schema, n_seed=16, width=12, depth=2, feat_dim=8, seed=3)
    ref = data.to_torch(feature, time, edge_list, graph)                       # data.py:212-256, verbatim
    mine = to_torch_layout(feature, time, edge_list, graph)
    for a, b in zip(ref[:5], mine[:5]):
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)
    assert ref[3].stride() == mine[3].stride()
    assert ref[5] == mine[5] and ref[6] == mine[6]
