"""CPU tests: the oracle against the reference's own outputs (golden fixtures) and,
when /root/reference is present, against the reference executed live."""
import pytest
import torch

from oracle import hgt_oracle as O
from oracle.reference_loader import reference_available, load_reference_conv, load_reference_model
from pyhgt_amd.synth import synthetic_typed_graph

TOL = 1e-4   # BASELINE.json north_star: within 1e-4 fp32


def test_closed_form_matches_golden(golden):
    g = golden
    out, att = O.forward_closed_form(g["sd"], g["T"], g["R"], g["H"], g["x"], g["node_type"], g["edge_index"],
                                     g["edge_type"], g["edge_time"], use_norm=g["use_norm"], use_RTE=g["use_RTE"],
                                     dtype=torch.float64, return_att=True, dense=g["dense"])
    assert (out.float() - g["out"]).abs().max().item() < 2e-5
    assert (att.float() - g["att"]).abs().max().item() < 2e-6
    out32 = O.forward_closed_form(g["sd"], g["T"], g["R"], g["H"], g["x"], g["node_type"], g["edge_index"],
                                  g["edge_type"], g["edge_time"], use_norm=g["use_norm"], use_RTE=g["use_RTE"],
                                  dtype=torch.float32, dense=g["dense"])
    assert (out32 - g["out"]).abs().max().item() < 2e-5


def test_reference_cost_port_matches_golden(golden):
    g = golden
    if g["E"] > 10000:
        pytest.skip("port is the slow path; covered on the small fixtures")
    if g["dense"]:
        pytest.skip("the timed CPU port restates HGTConv (the benchmarked layer) only")
    out, att = O.forward_meta_relation_port(g["sd"], g["T"], g["R"], g["H"], g["x"], g["node_type"],
                                            g["edge_index"], g["edge_type"], g["edge_time"],
                                            use_norm=g["use_norm"], use_RTE=g["use_RTE"], return_att=True)
    assert (out - g["out"]).abs().max().item() < 2e-5
    assert (att - g["att"]).abs().max().item() < 2e-6


def test_attention_rows_sum_to_one(golden):
    g = golden
    _, att = O.forward_closed_form(g["sd"], g["T"], g["R"], g["H"], g["x"], g["node_type"], g["edge_index"],
                                   g["edge_type"], g["edge_time"], use_norm=g["use_norm"], use_RTE=g["use_RTE"],
                                   return_att=True, dense=g["dense"])
    dst = g["edge_index"][1]
    s = torch.zeros(g["N"], g["H"], dtype=att.dtype).index_add_(0, dst, att)
    has_in = torch.zeros(g["N"], dtype=torch.bool)
    has_in[dst] = True
    assert (s[has_in] - 1.0).abs().max().item() < 1e-9
    assert s[~has_in].abs().max().item() == 0.0 if (~has_in).any() else True


def test_edge_permutation_invariance():
    T, R, H, d = 3, 4, 4, 32
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=5)
    x, nt, ei, et, tm = synthetic_typed_graph(300, 2000, d, T, R, seed=6)
    a = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm)
    p = torch.randperm(2000, generator=torch.Generator().manual_seed(1))
    b = O.forward_closed_form(sd, T, R, H, x, nt, ei[:, p], et[p], tm[p])
    assert (a - b).abs().max().item() < 1e-10


def test_out_of_range_relation_is_zero_logit():
    """conv.py:68-69: edges no meta relation claims keep logit 0 / message 0 but stay in the softmax."""
    T, R, H, d = 2, 3, 2, 16
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=2)
    x, nt, ei, et, tm = synthetic_typed_graph(50, 300, d, T, R, seed=3)
    et2 = et.clone()
    et2[::7] = R + 1
    o_cf = O.forward_closed_form(sd, T, R, H, x, nt, ei, et2, None, use_RTE=False, dtype=torch.float32)
    o_pt = O.forward_meta_relation_port(sd, T, R, H, x, nt, ei, et2, None, use_RTE=False)
    assert (o_cf - o_pt).abs().max().item() < 2e-5


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
def test_live_reference_agrees_with_oracle_and_param_count():
    conv = load_reference_conv()
    T, R, H, d = 4, 8, 8, 64
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=11)
    layer = conv.HGTConv(d, d, T, R, H, 0.2, True, True).eval()
    layer.load_state_dict(sd)
    x, nt, ei, et, tm = synthetic_typed_graph(1500, 12000, d, T, R, seed=12)
    with torch.no_grad():
        ref = layer(x, nt, ei, et, tm)
    got = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, dtype=torch.float32)
    assert (ref - got).abs().max().item() < 2e-5
    # structural known answer published by the reference: ogbn-mag/README.md:30
    model = load_reference_model()
    net = torch.nn.Sequential(model.GNN(129, 512, 4, 9, 8, 4, prev_norm=True, last_norm=True, use_RTE=True),
                              model.Classifier(512, 349))
    assert sum(p.numel() for p in net.parameters()) == 21173389


@pytest.mark.skipif(not reference_available(), reason="/root/reference only exists in the build container")
@pytest.mark.parametrize("dense", [False, True])
def test_backward_oracle_matches_autograd_through_the_live_reference(dense):
    """Oracle for the (not yet built) backward pass, SURVEY 8f-2: gradients of <out, g> with respect to the input features
    and every parameter, from oracle.backward_reference, against torch.autograd through the verbatim reference layer
    (eval mode, fp64 vs the reference's fp32 -> 2e-4 relative)."""
    conv = load_reference_conv()
    T, R, H, d, N, E = 3, 4, 4, 32, 300, 2500
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=21, dense=dense)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=22)
    g = torch.randn(N, d, generator=torch.Generator().manual_seed(23))
    layer = (conv.DenseHGTConv if dense else conv.HGTConv)(d, d, T, R, H, 0.2, True, True).eval()
    layer.load_state_dict(sd)
    xr = x.clone().requires_grad_(True)
    (layer(xr, nt, ei, et, tm) * g).sum().backward()
    got = O.backward_reference(sd, T, R, H, x, nt, ei, et, tm, g, dense=dense)

    def close(a, b):
        return (a.double() - b.double()).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())
    assert close(xr.grad, got["x"])
    for name, p in layer.named_parameters():
        if p.grad is None:                      # the sinusoid table is detached by nn.Embedding? (it is not: it has a grad)
            continue
        assert close(p.grad, got[name]), name


@pytest.mark.parametrize("name", ["gnn_oag2", "gnn_mag4"])
def test_gnn_restatement_matches_reference_gnn_goldens(name):
    """oracle.gnn_forward (model.py:66-80 restated) against rows of every layer's output of the VERBATIM reference GNN
    (oracle/gen_golden_gnn.py: 2-layer OAG shape, published 4-layer n_hid=512 ogbn-mag model)."""
    import os
    import numpy as np
    from oracle.gen_golden_gnn import GNN_CASES, build_batch
    c = GNN_CASES[name]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    _, (x, nt, tm, ei, et, _, _) = build_batch(c)
    assert x.size(0) == int(z["n_nodes"][0]) and et.numel() == int(z["n_edges"][0])
    sd = O.make_gnn_state_dict(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"], c["last_norm"],
                               c["use_RTE"], seed=c["seed"])
    _, layers = O.gnn_forward(sd, c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"], c["last_norm"],
                              c["use_RTE"], x, nt, tm, ei, et, return_layers=True)
    rows = torch.from_numpy(z["rows"]).long()
    for i, h in enumerate(layers):
        err = (h[rows].float() - torch.from_numpy(z["layers"][i])).abs().max().item()
        assert err < 1e-5, (name, i, err)
