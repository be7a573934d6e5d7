"""GPU tests of the backward pass: gradients of pyhgt_amd.HGTConv / GNN w.r.t. the input and EVERY
parameter of state_dict against oracle.backward_reference (reverse mode through the fp64 closed form, itself pinned against
autograd through the verbatim reference in tests/test_oracle.py).  Tolerance: 2e-4 relative to the largest entry of each
gradient tensor AND a per-entry atol + rtol bound (_grads_close)."""
import ctypes as C

import pytest
import torch

from oracle import hgt_oracle as O
from pyhgt_amd import HGTConv, GNN, GraphPlan, _lib
from pyhgt_amd.synth import synthetic_typed_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 2e-4

CASES = [
    # name, T, R, H, d, N, E, use_norm, use_RTE, graph kwargs, tweaks
    ("c1_like", 3, 4, 4, 64, 2000, 10000, True, True, {}, {}),
    ("c2_shape", 4, 8, 8, 256, 4000, 40000, True, False, {}, {}),
    ("dk50_unsorted", 2, 3, 4, 200, 1500, 9000, True, True, dict(sorted_types=False), dict(unknown=True)),
    ("no_norm_hubs", 3, 5, 2, 32, 3000, 30000, False, False, dict(dst_skew=1.1), dict(hub=True, unclaimed=True)),
    ("single_head", 2, 2, 1, 16, 300, 2500, True, True, {}, {}),
    ("three_heads", 2, 3, 3, 96, 1200, 9000, True, True, {}, {}),          # 64 % H != 0: padded head layout
]


ENTRY_RTOL, ENTRY_ATOL = 2e-3, 1e-3      # per entry: |got - ref| <= ENTRY_ATOL * rms(ref) + ENTRY_RTOL * |ref|


def _grads_close(name, got, ref, rtol=RTOL):
    """Two bounds: (a) max |got - ref| <= rtol * max |ref| (the round-2 form: one number per tensor), and (b) PER ENTRY
    |got - ref| <= ENTRY_ATOL * rms(ref) + ENTRY_RTOL * |ref| -- an entry of ordinary size (>= the tensor's rms) must be right
    to 0.3 %, a small one to 1e-3 of the rms, so that a wrong small-magnitude entry (one relation's prior, one type's gate)
    cannot hide behind a large one."""
    ref = ref.to(torch.float64)
    got = got.detach().cpu().to(torch.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-12)
    diff = (got - ref).abs()
    err = diff.max().item() / scale
    assert err < rtol, "%s: max |grad - oracle| = %.3e of the largest entry (%.3e)" % (name, err, scale)
    rms = max(ref.pow(2).mean().sqrt().item(), 1e-12)
    excess = (diff - (ENTRY_ATOL * rms + ENTRY_RTOL * ref.abs())).max().item()
    assert excess <= 0.0, "%s: an entry misses atol %.0e * rms (%.3e) + rtol %.0e by %.3e" % (name, ENTRY_ATOL, rms, ENTRY_RTOL, excess)
    return err


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hgtconv_backward_matches_oracle(case, precision):
    name, T, R, H, d, N, E, use_norm, use_RTE, gk, tw = case
    sd = O.make_state_dict(d, d, T, R, H, use_norm, use_RTE, seed=11)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=12, **gk)
    nt, et, ei = nt.clone(), et.clone(), ei.clone()
    if tw.get("unknown"):
        nt[::13] = T + 1                    # nodes no typed layer claims: zero output rows, no gradient
    if tw.get("unclaimed"):
        et[::7] = R                         # edges no meta relation claims: constant logit 0, no message
    if tw.get("hub"):
        ei[1, :4000] = 17                   # > 1024 in-edges: hub path of the aggregations (forward and transposed)
        ei[0, 4000:7000] = 23               # ... and a node with > 1024 OUT-edges: a hub of the transposed plan
    g = torch.Generator().manual_seed(5)
    gout = torch.randn(N, d, generator=g)
    ref = O.backward_reference(sd, T, R, H, x, nt, ei, et, tm if use_RTE else None, gout, use_norm=use_norm, use_RTE=use_RTE)
    layer = HGTConv(d, d, T, R, H, 0.2, use_norm, use_RTE, precision=precision).eval()     # eval: no dropout, like the oracle
    layer.load_state_dict(sd)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    GraphPlan.clear_cache()
    out = layer(xd, nt.to(DEV), ei.to(DEV), et.to(DEV), tm.to(DEV) if use_RTE else None)
    fwd = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm if use_RTE else None, use_norm=use_norm, use_RTE=use_RTE)
    assert (out.detach().cpu().double() - fwd).abs().max().item() < 1e-4
    out.backward(gout.to(DEV))
    torch.cuda.synchronize()
    worst = _grads_close("x", xd.grad, ref["x"])
    for k, p in layer.named_parameters():
        if k == "emb.emb.weight" and p.grad is None:
            continue
        assert p.grad is not None, k
        worst = max(worst, _grads_close(k, p.grad, ref[k]))
    print("backward %s / %s: worst relative gradient error %.2e over %d tensors" % (name, precision, worst,
                                                                                     1 + len(list(layer.parameters()))))


DENSE_CASES = [
    ("dense_c1_like", 3, 4, 4, 64, 2000, 10000, True, True, {}, {}),
    ("dense_no_norm_unknown", 2, 3, 4, 128, 1500, 9000, False, False, dict(sorted_types=False), dict(unknown=True, unclaimed=True)),
    ("dense_c2_shape", 4, 8, 8, 256, 3000, 30000, True, False, {}, dict(hub=True)),
]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", DENSE_CASES, ids=[c[0] for c in DENSE_CASES])
def test_dense_hgtconv_backward_matches_oracle(case, precision):
    """DenseHGTConv (conv.py:143-280): gradients of the input and of every parameter (a_linears, norms, mid_linear, out_linear,
    out_norm, relation_*, q/k/v_linears, emb) against reverse mode through the fp64 closed form with the dense update."""
    from pyhgt_amd import DenseHGTConv
    name, T, R, H, d, N, E, use_norm, use_RTE, gk, tw = case
    sd = O.make_state_dict(d, d, T, R, H, use_norm, use_RTE, seed=31, dense=True)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=32, **gk)
    nt, et, ei = nt.clone(), et.clone(), ei.clone()
    if tw.get("unknown"):
        nt[::13] = T + 1
    if tw.get("unclaimed"):
        et[::7] = R
    if tw.get("hub"):
        ei[1, :4000] = 17
    g = torch.Generator().manual_seed(6)
    gout = torch.randn(N, d, generator=g)
    ref = O.backward_reference(sd, T, R, H, x, nt, ei, et, tm if use_RTE else None, gout, use_norm=use_norm, use_RTE=use_RTE, dense=True)
    layer = DenseHGTConv(d, d, T, R, H, 0.2, use_norm, use_RTE, precision=precision).eval()
    layer.load_state_dict(sd)
    layer = layer.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    GraphPlan.clear_cache()
    out = layer(xd, nt.to(DEV), ei.to(DEV), et.to(DEV), tm.to(DEV) if use_RTE else None)
    fwd = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm if use_RTE else None, use_norm=use_norm, use_RTE=use_RTE, dense=True)
    assert (out.detach().cpu().double() - fwd).abs().max().item() < 1e-4
    out.backward(gout.to(DEV))
    torch.cuda.synchronize()
    worst = _grads_close("x", xd.grad, ref["x"])
    for k, p in layer.named_parameters():
        if k == "emb.emb.weight" and p.grad is None:
            continue
        assert p.grad is not None, k
        worst = max(worst, _grads_close(k, p.grad, ref[k]))
    print("dense backward %s / %s: worst relative gradient error %.2e" % (name, precision, worst))
    # training mode: two dropouts (conv.py:259,271), finite gradients
    layer.train()
    layer.zero_grad()
    o2 = layer(xd, nt.to(DEV), ei.to(DEV), et.to(DEV), tm.to(DEV) if use_RTE else None)
    o2.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for k, p in layer.named_parameters() if k != "emb.emb.weight")


def test_training_mode_runs_with_dropout_and_eval_matches():
    T, R, H, d, N, E = 3, 4, 4, 64, 1500, 9000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=3)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=4)
    layer = HGTConv(d, d, T, R, H, 0.2, True, True).to(DEV)
    layer.load_state_dict(sd)
    args = [t.to(DEV) for t in (x, nt, ei, et, tm)]
    layer.train()                                          # the reference's training scripts: model.train() + loss.backward()
    out = layer(*args)
    assert out.requires_grad and torch.isfinite(out).all()
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())
    layer.eval()
    with torch.no_grad():
        inf = layer(*args)                                 # inference path (fused kernels)
    diff = layer(*args)                                    # differentiable path in eval mode: no dropout -> same numbers
    assert (inf - diff.detach()).abs().max().item() < 5e-5
    layer.train()
    layer.drop.p = 0.0
    assert (layer(*args).detach() - inf).abs().max().item() < 5e-5


def test_gnn_training_step_matches_autograd_through_the_oracle():
    """Two-layer GNN (adapter + tanh + 2 x HGTConv, model.py:54-80) + Classifier head: gradients of every parameter against
    torch autograd through the fp64 closed-form oracle composed the same way."""
    from pyhgt_amd import Classifier
    T, R, H, in_dim, d, N, E, n_cls = 3, 4, 4, 37, 64, 1200, 8000, 5
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, in_dim, T, R, seed=21)
    torch.manual_seed(1)
    gnn = GNN(in_dim, d, T, R, H, 2, dropout=0.0, prev_norm=True, last_norm=True, use_RTE=True).to(DEV).train()
    head = Classifier(d, n_cls).to(DEV).train()
    y = torch.randint(0, n_cls, (200,))
    rep = gnn(x.to(DEV), nt.to(DEV), tm.to(DEV), ei.to(DEV), et.to(DEV))
    loss = torch.nn.functional.nll_loss(head(rep[:200]), y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # the same computation in fp64 torch on the CPU
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in list(gnn.named_parameters()) + [("head." + k, v) for k, v in
                                                                                                     head.named_parameters()]}
    h = torch.zeros(N, d, dtype=torch.float64)
    for t in range(T):
        idx = (nt == t).nonzero().flatten()
        h = h.index_add(0, idx, torch.tanh(x[idx].double() @ P["adapt_ws.%d.weight" % t].T + P["adapt_ws.%d.bias" % t]))
    for li in range(2):
        sd = {k[len("gcs.%d.base_conv." % li):]: v for k, v in P.items() if k.startswith("gcs.%d.base_conv." % li)}
        h = O.forward_closed_form(sd, T, R, H, h, nt, ei, et, tm, use_norm=True, use_RTE=True)
    logp = torch.log_softmax(h[:200] @ P["head.linear.weight"].T + P["head.linear.bias"], dim=-1)
    ref_loss = torch.nn.functional.nll_loss(logp, y)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4
    worst = 0.0
    for k, v in list(gnn.named_parameters()) + [("head." + k, v) for k, v in head.named_parameters()]:
        if P[k].grad is None:
            continue
        worst = max(worst, _grads_close(k, v.grad, P[k].grad, rtol=5e-4))
    print("GNN training step: worst relative gradient error %.2e" % worst)


@pytest.mark.parametrize("m,n_cols,n", [(256, 256, 5000), (768, 256, 3000), (64, 37, 1000), (100, 400, 700)])
def test_typed_weight_gradient_kernels(m, n_cols, n):
    lib = _lib.load()
    T = 3
    g = torch.Generator().manual_seed(m + n)
    A = torch.randn(n, m, generator=g)
    B = torch.randn(n, n_cols, generator=g)
    types = torch.randint(0, T + 1, (n,), generator=g)          # type T = rows of no group
    order = torch.argsort(types, stable=True).to(torch.int32)
    off = torch.zeros(T + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.bincount(types, minlength=T + 1)[:T], 0)
    Ad, Bd, od, fd = A.to(DEV), B.to(DEV), order.to(DEV), off.to(DEV)
    out = torch.zeros(T, m, n_cols, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.hgt_typed_wgrad(Ad.data_ptr(), m, Bd.data_ptr(), n_cols, od.data_ptr(), fd.data_ptr(), T, n, m, n_cols, out.data_ptr(),
                               m * n_cols, st) == 0
    cs = torch.zeros(T, m, device=DEV)
    assert lib.hgt_typed_colsum(Ad.data_ptr(), m, od.data_ptr(), fd.data_ptr(), T, n, m, cs.data_ptr(), m, st) == 0
    # the split-bf16 x3 form (128 x 128 tiles, transposed LDS staging) with the column sums from the same pass
    out3 = torch.zeros(T, m, n_cols, device=DEV)
    cs3 = torch.zeros(T, m, device=DEV)
    assert lib.hgt_typed_wgrad_bf16x3(Ad.data_ptr(), m, Bd.data_ptr(), n_cols, od.data_ptr(), fd.data_ptr(), T, n, m, n_cols, out3.data_ptr(),
                                      m * n_cols, cs3.data_ptr(), m, st) == 0
    torch.cuda.synchronize()
    for t in range(T):
        idx = (types == t).nonzero().flatten()
        ref = A[idx].double().T @ B[idx].double()
        scale = max(1.0, ref.abs().max().item())
        assert (out[t].cpu().double() - ref).abs().max().item() < 1e-4 * scale
        assert (out3[t].cpu().double() - ref).abs().max().item() < 1e-4 * scale
        assert (cs[t].cpu().double() - A[idx].double().sum(0)).abs().max().item() < 1e-3
        assert (cs3[t].cpu().double() - A[idx].double().sum(0)).abs().max().item() < 1e-3


@pytest.mark.parametrize("conv", ["hgt", "dense_hgt"])
def test_training_loop_reduces_the_loss(conv):
    """examples/train_synthetic.py: the reference's training-loop shape (sampled batch -> to_device_graph -> GNN -> Classifier ->
    nll_loss -> backward -> AdamW) on the HIP forward + backward; the loss of a learnable synthetic task must fall."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                 "examples", "train_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    losses = mod.run("mag", steps=40, conv=conv, verbose=False)
    assert all(l == l for l in losses)                              # finite
    assert sum(losses[-8:]) / 8 < 0.8 * sum(losses[:4]) / 4, (losses[:4], losses[-8:])


@pytest.mark.parametrize("precision", ["bf16x3"])
def test_training_path_at_the_benchmark_size_sampled(precision):
    """The training forward + backward at BASELINE.json configs[1] ITSELF (1M nodes / 10M edges, d=256, H=8: the unfused
    forward kernels with kept intermediates, the transposed 10M-edge plan, the hub-free spmm passes at full occupancy --
    sizes no small case reaches).  Checked exactly on a sample: the loss weights only ~300 target rows (type-boundary tiles,
    first / last tile, max in-degree, random), so every gradient depends only on the sub-graph induced by ALL in-edges of
    those rows, where oracle.backward_reference (fp64) is run; the sampled OUTPUT rows are compared the same way."""
    from pyhgt_amd.synth import pick_check_targets, induced_in_neighbourhood
    T, R, H, d, N, E = 4, 8, 8, 256, 1_000_000, 10_000_000
    g = torch.Generator(device=DEV).manual_seed(2024)
    nt = torch.randint(0, T, (N,), generator=g, device=DEV).sort().values
    x = torch.randn(N, d, generator=g, device=DEV)
    src = torch.randint(0, N, (E,), generator=g, device=DEV)
    dst = torch.randint(0, N, (E,), generator=g, device=DEV)
    et = torch.randint(0, R, (E,), generator=g, device=DEV)
    ei = torch.stack([src, dst], dim=1).t()
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=9)
    layer = HGTConv(d, d, T, R, H, 0.2, True, False, precision=precision).eval()
    layer.load_state_dict(sd)
    layer = layer.to(DEV)
    tg = pick_check_targets(nt, dst, n_random=150, tile=16, seed=4)
    gout_rows = torch.randn(tg.numel(), d, generator=torch.Generator().manual_seed(6))
    gout = torch.zeros(N, d, device=DEV)
    gout[tg] = gout_rows.to(DEV)
    GraphPlan.clear_cache()
    xd = x.clone().requires_grad_(True)
    out = layer(xd, nt, ei, et)                      # grad enabled: the training path (pyhgt_amd/autograd.py)
    out.backward(gout)
    torch.cuda.synchronize()
    xs, nts, eis, ets, _, pos = induced_in_neighbourhood(x, nt, ei, et, None, tg)
    nodes = torch.unique(torch.cat([tg, src[torch.isin(dst, tg)]]))       # the sub-graph's node ids (sorted, like `pos`)
    gsub = torch.zeros(xs.size(0), d)
    gsub[pos] = gout_rows
    fwd = O.forward_closed_form(sd, T, R, H, xs, nts, eis, ets, None, use_norm=True, use_RTE=False, dtype=torch.float64)
    ferr = (out.detach()[tg].cpu().double() - fwd[pos]).abs().max().item()
    assert ferr < 1e-4, ferr
    ref = O.backward_reference(sd, T, R, H, xs, nts, eis, ets, None, gsub, use_norm=True, use_RTE=False)
    worst = _grads_close("x[sub-graph]", xd.grad[nodes], ref["x"])
    outside = torch.ones(N, dtype=torch.bool, device=DEV)
    outside[nodes] = False
    assert xd.grad[outside].abs().max().item() == 0.0                     # nothing else can receive gradient
    for k, p in layer.named_parameters():
        assert p.grad is not None, k
        worst = max(worst, _grads_close(k, p.grad, ref[k]))
    print("training path at c2 size (%s): %d sampled rows, %d sub-graph edges, forward err %.2e, worst gradient error %.2e" % (
        precision, tg.numel(), eis.size(1), ferr, worst))
    del out, xd, gout
    GraphPlan.clear_cache()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("d,H,R,rte", [(256, 8, 9, True), (512, 8, 5, False), (64, 4, 4, True), (400, 8, 33, True)])
def test_item_parallel_spmm_matches_the_sub_tile_kernel(d, H, R, rte):
    """hgt_edge_spmm_items (round 6: the gather passes of the backward on the item-parallel kernels, what autograd takes for sampled
    batches) against hgt_edge_spmm on the same plan, weights and rows -- the same sums in another order (fp32 rounding), hub targets and
    unclaimed relations included; two calls are bit-identical (no atomics)."""
    lib = _lib.load()
    T, N, E = 3, 3000, 30000
    x, nt, ei, et, tm = [t.to(DEV) for t in synthetic_typed_graph(N, E, d, T, R, seed=d + R, dst_skew=1.1)]
    et = et.clone()
    et[::41] = R
    plan = GraphPlan(nt, ei, et, tm if rte else None, T, R)
    lay = _lib.HgtLayout()
    assert lib.hgt_layout_for(d, H, C.byref(lay)) == 0
    Hl, dkp, dp = lay.heads, lay.dk_pad, lay.d_pad
    g = torch.Generator().manual_seed(7)
    w = torch.randn(E, Hl, generator=g).to(DEV)
    rows = torch.randn(N, dp, generator=g).to(DEV)
    rte_rows = torch.randn(T * 240, dp, generator=g).to(DEV) if rte else None
    f_p = (torch.randn(R, Hl, dkp, dkp, generator=g) / dkp ** 0.5).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = C.c_uint64()
    assert lib.hgt_relation_frag_bytes(R, Hl, dkp, C.byref(nb)) == 0
    frag = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    assert lib.hgt_relation_frag_pack(f_p.data_ptr(), R, Hl, dkp, frag.data_ptr(), st) == 0
    assert lib.hgt_hub_workspace_bytes(E, Hl, dkp, C.byref(nb)) == 0
    hub = torch.empty(max(int(nb.value), 256), dtype=torch.uint8, device=DEV)
    assert lib.hgt_edge_aggregate_items_bytes(E, Hl, dkp, C.byref(nb)) == 0
    scratch = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    rp = 0 if rte_rows is None else rte_rows.data_ptr()
    ref = torch.zeros(N, dp, device=DEV)
    assert lib.hgt_edge_spmm(plan.ptr, N, E, T, R, Hl, dkp, w.data_ptr(), rows.data_ptr(), rp, f_p.data_ptr(), frag.data_ptr(), ref.data_ptr(),
                             dp, N, hub.data_ptr(), st) == 0
    outs = []
    for _ in range(2):
        out = torch.full((N, 2 * dp), 5.0, device=DEV)      # (written into the second column block of a wider array: ld_out = 2 dp)
        assert lib.hgt_edge_spmm_items(plan.ptr, N, E, T, R, Hl, dkp, w.data_ptr(), rows.data_ptr(), rp, frag.data_ptr(),
                                       out.data_ptr() + 4 * dp, 2 * dp, N, scratch.data_ptr(), scratch.numel(), st) == 0
        torch.cuda.synchronize()
        assert bool((out[:, :dp] == 5.0).all())
        outs.append(out[:, dp:].clone())
    assert torch.equal(outs[0], outs[1])
    scale = ref.abs().max().item()
    err = (outs[0] - ref).abs().max().item() / scale
    print("spmm_items d=%d R=%d: max|items - sub-tile| = %.2e of the largest entry" % (d, R, err))
    assert err < 2e-5
