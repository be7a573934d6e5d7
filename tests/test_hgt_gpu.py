"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(a) golden outputs of the reference itself (tests/golden, made by oracle/gen_golden.py),
(b) the CPU oracle on seeded inputs, (c) size-independent properties at larger sizes.
Tolerance: 1e-4 absolute on fp32 outputs (BASELINE.json north_star); integer plan data bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import hgt_oracle as O
from pyhgt_amd import HGTConv, DenseHGTConv, GeneralConv, GraphPlan, _lib
from pyhgt_amd.synth import synthetic_typed_graph

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _layer_from(sd, d, T, R, H, use_norm, use_RTE, keep_att=True, precision="fp32", dense=False):
    cls = DenseHGTConv if dense else HGTConv
    layer = cls(d, d, T, R, H, 0.2, use_norm, use_RTE, keep_att=keep_att, precision=precision).eval()
    layer.load_state_dict(sd)
    return layer.to(DEV)


def _to_dev(*ts):
    return [None if t is None else t.to(DEV) for t in ts]


def _run(layer, x, nt, ei, et, tm):
    GraphPlan.clear_cache()
    with torch.no_grad():
        out = layer(*_to_dev(x, nt, ei, et, tm))
    torch.cuda.synchronize()
    return out.cpu(), (layer.att.cpu() if layer.att is not None else None)


# ------------------------------------------------------------------ (a) the reference's own outputs
def test_matches_reference_golden(golden):
    g = golden
    layer = _layer_from(g["sd"], g["d"], g["T"], g["R"], g["H"], g["use_norm"], g["use_RTE"], dense=g["dense"])
    out, att = _run(layer, g["x"], g["node_type"], g["edge_index"], g["edge_type"], g["edge_time"])
    assert out.shape == g["out"].shape
    assert (out - g["out"]).abs().max().item() < TOL
    assert (att - g["att"]).abs().max().item() < 1e-5


def test_split_bf16_precision_matches_reference_golden(golden):
    """precision="bf16x3" (3-term split-bf16 MFMA for the typed linears) must meet the same 1e-4 bound."""
    g = golden
    layer = _layer_from(g["sd"], g["d"], g["T"], g["R"], g["H"], g["use_norm"], g["use_RTE"], precision="bf16x3", dense=g["dense"])
    out, att = _run(layer, g["x"], g["node_type"], g["edge_index"], g["edge_type"], g["edge_time"])
    assert (out - g["out"]).abs().max().item() < TOL
    assert (att - g["att"]).abs().max().item() < 1e-5


def test_split_f16_precision_matches_reference_golden(golden):
    """precision="f16x3" (fp16 hi / lo split with power-of-two row scales) against the verbatim reference's outputs: the
    difference is the reference's own fp32 rounding, an order of magnitude inside the 1e-4 bound."""
    g = golden
    layer = _layer_from(g["sd"], g["d"], g["T"], g["R"], g["H"], g["use_norm"], g["use_RTE"], precision="f16x3", dense=g["dense"])
    out, att = _run(layer, g["x"], g["node_type"], g["edge_index"], g["edge_type"], g["edge_time"])
    err = (out - g["out"]).abs().max().item()
    print("golden %s f16x3: max|err| %.2e" % (g["name"], err))
    assert err < F16_TOL
    assert (att - g["att"]).abs().max().item() < 1e-5


def test_default_constructed_layer_is_reference_accurate(golden):
    """Round 6: a layer built with the reference's own constructor arguments (no precision keyword) runs the fp16 hi / lo split --
    every golden of the verbatim reference within its own fp32 rounding (bound 1e-5, measured <= 2e-6), not just the 1e-4 of the
    north star; the bf16 split of rounds 1-5 is the opt-in fast mode."""
    from pyhgt_amd.conv import DEFAULT_PRECISION
    g = golden
    cls = DenseHGTConv if g["dense"] else HGTConv
    layer = cls(g["d"], g["d"], g["T"], g["R"], g["H"], 0.2, g["use_norm"], g["use_RTE"]).eval()
    assert layer.precision == DEFAULT_PRECISION == "f16x3"
    layer.load_state_dict(g["sd"])
    layer = layer.to(DEV)
    out, _ = _run(layer, g["x"], g["node_type"], g["edge_index"], g["edge_type"], g["edge_time"])
    err = (out - g["out"]).abs().max().item()
    print("golden %s default-constructed layer: max|err| %.2e" % (g["name"], err))
    assert err < F16_TOL


# ------------------------------------------------------------------ (b) oracle on seeded inputs
F16_TOL = 1e-5          # "f16x3" against the fp64 closed form / the reference's fp32 outputs (measured <= 2e-6)
PREC_TOL = {"fp32": TOL, "bf16x3": TOL, "f16x3": F16_TOL}
CASES = [
    # N, E, d, H, T, R, use_norm, use_RTE, graph kwargs
    (3000, 30000, 256, 8, 4, 8, True, False, {}),                     # c2 shape, small
    (3000, 30000, 256, 8, 4, 8, True, True, {}),
    (1000, 9000, 512, 8, 4, 9, True, True, {}),                       # ogbn-mag width: d_k = 64 (non-hoisted path)
    (600, 5000, 400, 8, 5, 33, True, True, dict(schema=True)),         # OAG shape: d_k = 50 padded to 64
    (2000, 20000, 128, 16, 3, 5, False, False, {}),                    # 16 heads, d_k = 8
    (2000, 20000, 64, 1, 2, 3, True, True, {}),                        # one head
    (5000, 60000, 64, 4, 3, 4, True, True, dict(dst_skew=1.1)),        # hubs: items split at 256 edges
    (700, 3000, 32, 2, 3, 2, True, False, dict(sorted_types=False, strided_edge_index=False)),
    (129, 1, 64, 4, 2, 2, True, True, {}),                             # a single edge
    (700, 6000, 768, 8, 3, 5, True, True, {}),                         # n_hid 768: d_k = 96 padded to 128, rows of 1024 padded columns
    (600, 5000, 1024, 16, 2, 4, True, False, {}),                      # n_hid 1024, 16 heads (conv.py:21 accepts any d % H == 0)
    (500, 4000, 1024, 8, 2, 3, False, True, {}),                       # n_hid 1024, 8 heads: d_k = 128
]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_matches_oracle(case, precision):
    N, E, d, H, T, R, use_norm, use_RTE, gk = case
    sd = O.make_state_dict(d, d, T, R, H, use_norm, use_RTE, seed=N + E)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 1, **gk)
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=use_norm, use_RTE=use_RTE,
                                         dtype=torch.float64, return_att=True)
    layer = _layer_from(sd, d, T, R, H, use_norm, use_RTE, precision=precision)
    out, att = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    err, err_att = (out.double() - ref).abs().max().item(), (att.double() - att_ref).abs().max().item()
    print("case N=%d E=%d d=%d H=%d %s: max|out| err %.2e, att err %.2e" % (N, E, d, H, precision, err, err_att))
    assert err < PREC_TOL[precision]
    assert err_att < 1e-5


MFMA_LOGITS_CASES = [
    # N, E, d, H, T, R, use_RTE: wavefront layouts (VEC, LPH) of hgt_edge_logits_mfma
    (3000, 30000, 256, 8, 4, 8, True),        # (4, 8): the benchmark layout (vector-ALU kernel by default; forced here)
    (1500, 12000, 512, 8, 3, 9, True),        # (4, 16) x 2 head groups: ogbn-mag width
    (1500, 12000, 256, 4, 3, 5, False),       # (4, 16)
    (1200, 9000, 128, 2, 2, 4, True),         # (2, 32)
    (1200, 9000, 256, 2, 2, 4, False),        # (4, 32)
    (900, 7000, 64, 1, 2, 3, True),           # (1, 64)
    (900, 7000, 128, 1, 2, 3, False),         # (2, 64)
    (900, 7000, 256, 1, 2, 3, True),          # (4, 64)
    (5000, 60000, 256, 4, 3, 4, True),        # (4, 16) with hub targets (dst_skew below): many chunks with one slot
]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("case", MFMA_LOGITS_CASES, ids=[str(i) for i in range(len(MFMA_LOGITS_CASES))])
def test_logits_with_matrix_core_transforms(case, precision):
    """hgt_edge_logits_mfma (target-side relation transform of 16 distinct targets at a time on the matrix cores) on every
    wavefront layout it is instantiated for: the attention weights (softmax of the logits) and the layer output against the
    fp64 closed form; unclaimed relations (logit 0) included."""
    N, E, d, H, T, R, use_RTE = case
    gk = dict(dst_skew=1.1) if N == 5000 else {}
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=N + E + d)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 7, **gk)
    et = et.clone()
    et[::41] = R
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=use_RTE,
                                         dtype=torch.float64, return_att=True)
    layer = _layer_from(sd, d, T, R, H, True, use_RTE, precision=precision)
    layer.kernel_flags = 4          # HGT_FLAG_MFMA_LOGITS
    out, att = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    err, err_att = (out.double() - ref).abs().max().item(), (att.double() - att_ref).abs().max().item()
    print("mfma logits N=%d E=%d d=%d H=%d %s: max|out| err %.2e, att err %.2e" % (N, E, d, H, precision, err, err_att))
    assert err < PREC_TOL[precision]
    assert err_att < (1e-5 if precision == "bf16x3" else 2e-6)
    layer.kernel_flags = 8          # HGT_FLAG_VALU_LOGITS: the two kernels agree on the attention weights
    out2, att2 = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    assert (att2 - att).abs().max().item() < 1e-5


ITEM_AGG_CASES = [
    # N, E, d, H, T, R, use_RTE, graph kwargs: hgt_edge_aggregate_items on every kind of wavefront layout
    (3000, 30000, 256, 8, 4, 8, True, {}),                       # (4, 8): the benchmark layout (sub-tile kernel by default; forced here)
    (1500, 12000, 512, 8, 3, 9, True, {}),                       # (4, 16) x 2 head groups
    (600, 5000, 400, 8, 5, 33, True, dict(schema=True)),          # OAG shape: 33 relations, d_k = 50 padded to 64
    (2000, 20000, 128, 16, 3, 5, False, {}),                      # (2, 4): 16 heads
    (2000, 20000, 64, 1, 2, 3, True, {}),                         # (1, 64): one head
    (5000, 60000, 64, 4, 3, 4, True, dict(dst_skew=1.1)),         # hub targets: runs of thousands of edges across items and chunks
    (700, 3000, 32, 2, 3, 2, False, dict(sorted_types=False)),    # (1, 32)... half-empty rows
    (129, 1, 64, 4, 2, 2, True, {}),                              # a single edge
]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("case", ITEM_AGG_CASES, ids=[str(i) for i in range(len(ITEM_AGG_CASES))])
def test_item_parallel_aggregation(case, precision):
    """hgt_edge_aggregate_items (latency regime: runs per work item on the matrix cores + an ordered merge per target) forced on
    every layout: output against the fp64 closed form, unclaimed relations and unknown node types included; two forwards are
    bit-identical (no atomics), and the sub-tile kernels (flag 32) agree."""
    N, E, d, H, T, R, use_RTE, gk = case
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=N + E + 3)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 11, **gk)
    nt, et = nt.clone(), et.clone()
    if N > 200:
        nt[::131] = T + 2
        et[::37] = R
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=use_RTE, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, True, use_RTE, keep_att=False, precision=precision)
    layer.kernel_flags = 16         # HGT_FLAG_ITEM_AGGREGATE
    out, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    out_b, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    err = (out.double() - ref).abs().max().item()
    print("item aggregation N=%d E=%d d=%d H=%d R=%d %s: max|out| err %.2e" % (N, E, d, H, R, precision, err))
    assert err < PREC_TOL[precision]
    assert torch.equal(out, out_b)
    layer.kernel_flags = 32         # HGT_FLAG_NO_ITEM_AGGREGATE
    out2, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    assert (out2 - out).abs().max().item() < (1e-4 if precision == "bf16x3" else 1e-5)
    # HGT_FLAG_SINGLE_PASS: logits inside the runs kernel (hgt_edge_single_pass.hip, where it is instantiated; the other layouts
    # fall through to the two kernels): must agree with the two-kernel form to the split precision
    layer.kernel_flags = 16 | _lib.HGT_FLAG_SINGLE_PASS
    out3, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    out3b, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    assert torch.equal(out3, out3b)
    assert (out3 - out).abs().max().item() < (2e-5 if precision == "bf16x3" else 2e-6)
    assert (out3.double() - ref).abs().max().item() < PREC_TOL[precision]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("case", ITEM_AGG_CASES, ids=[str(i) for i in range(len(ITEM_AGG_CASES))])
@pytest.mark.parametrize("use_norm", [True, False])
def test_merge_pass_fused_with_the_node_update_is_bit_identical(case, precision, use_norm):
    """Round 6 (sampled batches): k_merge_update -- the merge pass of the item-parallel aggregation IS the node update (a_linear +
    gated skip + LayerNorm, conv.py:119-133), the merged rows are never written -- against hgt_edge_aggregate_items +
    hgt_linear_update_* as two calls (HGT_FLAG_NO_MERGE_UPDATE): the same merged row bit for bit and the same split products; the
    16 x 16 x 32 MFMAs of the one-target-per-wavefront form sum 32 k per instruction where the tile kernels sum 16, the epilogue's fused
    multiply-adds are contracted differently and the LayerNorm sums meet in a different wavefront order: last-bit differences only.  Unclaimed relations, unknown node types, one / sixteen heads, 64 .. 512 columns; also against the fp64 closed form."""
    if HGTConv.EXTRA_KERNEL_FLAGS & _lib.HGT_FLAG_FUSED_ANY_SIZE:
        pytest.skip("the forced-kernel pass (tools/gpu.sh final) puts the fused sub-tile kernel on every layer: neither form under test runs")
    N, E, d, H, T, R, use_RTE, gk = case
    sd = O.make_state_dict(d, d, T, R, H, use_norm, use_RTE, seed=N + E + 5)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 13, **gk)
    nt, et = nt.clone(), et.clone()
    if N > 200:
        nt[::131] = T + 2
        et[::37] = R
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=use_norm, use_RTE=use_RTE, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, use_norm, use_RTE, keep_att=False, precision=precision)
    layer.kernel_flags = _lib.HGT_FLAG_ITEM_AGGREGATE
    fused, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    fused_b, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    layer.kernel_flags = _lib.HGT_FLAG_ITEM_AGGREGATE | _lib.HGT_FLAG_NO_MERGE_UPDATE
    two, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    assert torch.equal(fused, fused_b)
    diff = (fused - two).abs().max().item()
    err = (fused.double() - ref).abs().max().item()
    print("merge+update N=%d d=%d H=%d R=%d norm=%s %s: |fused - two calls| %.2e, err %.2e" % (N, d, H, R, use_norm, precision, diff, err))
    assert diff <= 4e-6
    assert err < PREC_TOL[precision]


COOP_CASES = ITEM_AGG_CASES[0:3] + [
    (2000, 20000, 64, 1, 2, 3, True, {}),                         # (1, 64): one column tile per wavefront
    (3000, 40000, 128, 2, 3, 5, False, {}),                       # (2, 32)
    (1100, 9000, 256, 2, 2, 7, True, {}),                         # (2, 64) x 2 head groups: d_k = 128, eight fragment steps per wavefront
    (12000, 600000, 64, 1, 3, 4, True, dict(dst_skew=1.1)),       # 128-edge items: two chunks, several rounds of 16 slots, hub runs
    (12000, 2200000, 64, 1, 2, 3, False, dict(dst_skew=0.6)),     # 512-edge items: every wavefront of a workgroup walks two chunks of ONE item
    (15000, 560000, 128, 2, 3, 6, False, {}),                     # 128-edge items, no hubs: the wavefronts of a workgroup run out of slots at different rounds
]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("case", COOP_CASES, ids=[str(i) for i in range(len(COOP_CASES))])
def test_relation_transform_shared_by_the_workgroup_is_bit_identical(case, precision):
    """Round 6, d_k >= 64 (logits) / >= 32 (runs): k_edge_logits_coop / k_edge_runs_coop -- the four wavefronts of a workgroup keep their own work items but
    split the relation transform (conv.py:98-99, 101-102) by columns, a quarter of the fragment image each, kept in registers while
    consecutive items share the relation -- against the one-wavefront-per-item kernels (HGT_FLAG_NO_COOP_EDGE): the same MFMA
    products in the same order, so the layer's output is the same bit for bit; also against the fp64 closed form.  16-edge items
    (sampled batches: the default domain of the shared form) and 128 / 512-edge items (HGT_FLAG_COOP_EDGE_ALWAYS: several chunks and
    lock-step rounds per workgroup), hub runs, unclaimed relations, unknown types."""
    if HGTConv.EXTRA_KERNEL_FLAGS & _lib.HGT_FLAG_FUSED_ANY_SIZE:
        pytest.skip("the forced-kernel pass puts the fused sub-tile kernel on every layer: the runs kernels under test do not run")
    N, E, d, H, T, R, use_RTE, gk = case
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=N + E + 9)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 17, **gk)
    nt, et = nt.clone(), et.clone()
    nt[::131] = T + 2
    et[::37] = R
    layer = _layer_from(sd, d, T, R, H, True, use_RTE, keep_att=False, precision=precision)
    layer.kernel_flags = _lib.HGT_FLAG_ITEM_AGGREGATE | _lib.HGT_FLAG_MFMA_LOGITS | _lib.HGT_FLAG_COOP_EDGE_ALWAYS
    coop, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    coop_b, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    layer.kernel_flags = _lib.HGT_FLAG_ITEM_AGGREGATE | _lib.HGT_FLAG_MFMA_LOGITS | _lib.HGT_FLAG_NO_COOP_EDGE
    single, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    single_b, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    assert torch.equal(single, single_b), "one wavefront per item: two forwards differ"
    assert torch.equal(coop, coop_b), "shared transform: two forwards differ"
    assert torch.equal(coop, single)
    if E <= 100000:
        ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=use_RTE, dtype=torch.float64)
        assert (coop.double() - ref).abs().max().item() < PREC_TOL[precision]


DENSE_CASES = [
    # N, E, d, H, T, R, use_norm, use_RTE
    (2500, 25000, 256, 8, 4, 8, True, False),      # c2 shape
    (900, 7000, 400, 8, 5, 12, True, True),        # d_k = 50 padded to 64; out_linear has K = 800 (multi-panel GEMM)
    (1200, 9000, 64, 4, 3, 4, False, True),        # no per-type LayerNorm
    (1419, 6268, 128, 16, 1, 9, True, False),      # one node type, d_pad = 128: the shared dense layer sizes the weight-tile scratch
]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("case", DENSE_CASES, ids=[str(i) for i in range(len(DENSE_CASES))])
def test_dense_hgt_conv_matches_oracle(case, precision):
    """DenseHGTConv (conv.py:143-280): HGTConv's message() + the dense update; unknown node types and unclaimed
    relations included (rows of unknown type are 0, conv.py:255)."""
    N, E, d, H, T, R, use_norm, use_RTE = case
    sd = O.make_state_dict(d, d, T, R, H, use_norm, use_RTE, seed=7 * N + E, dense=True)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 3)
    nt = nt.clone(); et = et.clone()
    nt[::97] = T + 1                                  # unknown node types
    et[::53] = R                                      # relation ids no meta relation claims
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=use_norm, use_RTE=use_RTE,
                                         dtype=torch.float64, return_att=True, dense=True)
    layer = _layer_from(sd, d, T, R, H, use_norm, use_RTE, precision=precision, dense=True)
    out, att = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    err = (out.double() - ref).abs().max().item()
    print("dense case N=%d E=%d d=%d H=%d %s: max|out| err %.2e" % (N, E, d, H, precision, err))
    assert err < PREC_TOL[precision]
    assert (att.double() - att_ref).abs().max().item() < 1e-5
    assert out[nt == T + 1].abs().max().item() == 0.0


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
def test_prepared_weight_images_follow_parameter_updates(precision):
    """hgt_conv_args.prepared keeps the weight-only preprocessing (packed relation matrices, split weight tiles, temporal
    tables) across calls; it must be rebuilt when any parameter changes in place and reused (bit-identical output) when
    nothing changed."""
    T, R, H, d, N, E = 3, 4, 4, 64, 1500, 12000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=61)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=62)
    layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision=precision)
    out1, _ = _run(layer, x, nt, ei, et, tm)
    out2, _ = _run(layer, x, nt, ei, et, tm)                    # second call trusts the prepared buffer
    assert torch.equal(out1, out2)
    assert (out1.double() - O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, dtype=torch.float64)).abs().max().item() < TOL
    with torch.no_grad():                                        # in-place updates of every kind of cached parameter
        layer.relation_pri.mul_(1.7)
        layer.relation_msg.add_(0.05)
        layer.k_linears[1].weight.mul_(0.8)
        layer.a_linears[0].weight.add_(0.02)
        layer.emb.lin.bias.add_(0.1)
    sd2 = {k: v.detach().cpu().clone() for k, v in layer.state_dict().items()}
    out3, _ = _run(layer, x, nt, ei, et, tm)
    ref3 = O.forward_closed_form(sd2, T, R, H, x, nt, ei, et, tm, dtype=torch.float64)
    assert (out3.double() - ref3).abs().max().item() < TOL
    assert (out3 - out1).abs().max().item() > 1e-3             # the update really changed the result


def test_large_logit_spread_forces_softmax_rereference():
    """The aggregation kernel re-references its running softmax only when a logit exceeds the segment's reference by
    more than 40 (deferred rescaling).  Blow the logits up (relation_pri x 60, multi-edge segments via few targets)
    so that branch is taken, and compare with the fp64 oracle."""
    T, R, H, d, N, E = 2, 2, 4, 64, 400, 12000
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=51)
    sd["relation_pri"] = sd["relation_pri"] * 60.0
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=52)
    ei = ei.clone()
    ei[1] = ei[1] % 40                                   # 40 targets x 2 relations -> ~150 edges per segment
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_RTE=False, dtype=torch.float64, return_att=True)
    s_spread = float((att_ref.max(0).values.log() - att_ref.clamp_min(1e-300).min(0).values.log()).max())
    assert s_spread > 60.0                               # the input really spans more than the threshold
    layer = _layer_from(sd, d, T, R, H, True, False)
    out, att = _run(layer, x, nt, ei, et, None)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < TOL
    assert (att.double() - att_ref).abs().max().item() < 1e-4   # logits ~ 1e2: fp32 logit rounding alone is ~1e-5


@pytest.mark.parametrize("use_RTE", [False, True])
def test_hub_targets_take_the_split_path(use_RTE):
    """Targets with more than 1024 in-edges are aggregated by the hub kernels (edge ranges split over many
    wavefronts, fp32 atomics); everything else stays on the one-wavefront-per-sub-tile path."""
    T, R, H, d, N, E = 3, 4, 4, 64, 3000, 40000
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=61)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=62)
    ei = ei.clone()
    ei[1, :9000] = 5                    # hub 1: 9000 in-edges, all relations
    ei[1, 9000:12500] = 777             # hub 2
    ei[1, 12500:13600] = 2999           # hub 3: just over the threshold, last node
    et = et.clone()
    et[100:400] = R + 3                 # some unclaimed edges into hub 1 (logit 0, no message)
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm if use_RTE else None, use_RTE=use_RTE,
                                         dtype=torch.float64, return_att=True)
    layer = _layer_from(sd, d, T, R, H, True, use_RTE)
    out, att = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    plan = GraphPlan(*_to_dev(nt, ei, et, tm if use_RTE else None), T, R)
    torch.cuda.synchronize()
    assert int(plan.buf[:12].view(torch.int32).cpu()[2]) == 3          # three hubs detected
    assert (out.double() - ref).abs().max().item() < TOL
    assert (att.double() - att_ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("scale", [-60.0, 60.0, -400.0, 400.0])
def test_hub_targets_with_extreme_logits(scale):
    """Hub path numerics (round-1 review): the per-(hub, head) reference of the exp-sums must be the TRUE max logit.
    relation_pri x (+-60) spreads a hub's logits over hundreds of units; x (+-400) plus a multi-edge hub (2000 copies of ONE
    edge, so all its logits are equal) yields, for one of the two signs, a hub whose logits are ALL far below -87: with a
    reference of max(max, 0) every exp underflows and the hub's aggregate collapses to 0."""
    T, R, H, d, N, E = 2, 3, 4, 64, 2500, 30000
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=71)
    sd["relation_pri"] = sd["relation_pri"] * scale
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=72)
    ei, et = ei.clone(), et.clone()
    ei[1, :6000] = 11                   # hub with random sources / relations: logits of both signs
    ei[1, 6000:8000] = 1200             # multi-edge hub: 2000 x the same (source, relation) -> identical logits
    ei[0, 6000:8000] = 77
    et[6000:8000] = 1
    other = torch.ones(E, dtype=torch.bool)
    other[6000:8000] = False
    ei[1, other & (ei[1] == 1200)] = 1201          # nothing else enters the multi-edge hub
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_RTE=False, dtype=torch.float64, return_att=True)
    assert (att_ref[6000:8000] - 1.0 / 2000).abs().max().item() < 1e-12
    layer = _layer_from(sd, d, T, R, H, True, False)
    out, att = _run(layer, x, nt, ei, et, None)
    assert torch.isfinite(out).all()
    # logits of magnitude ~1e3 carry ~1e-4 of fp32 rounding into the exponent: the 1e-4 bar holds for the x60 cases, the x400
    # cases test against a collapse to 0 (an O(1) error)
    assert (out.double() - ref).abs().max().item() < (TOL if abs(scale) <= 60 else 2e-3)
    assert (att[6000:8000].double() - 1.0 / 2000).abs().max().item() < 1e-7
    assert (att.double() - att_ref).abs().max().item() < (1e-4 if abs(scale) <= 60 else 2e-3)   # logits ~ 1e2..1e3 in fp32


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("d,H", [(96, 3), (80, 5), (96, 6), (192, 12)])
def test_head_counts_that_do_not_divide_64(d, H, precision):
    """conv.py:21 only needs d % n_heads == 0.  3 / 5 / 6 / 12 heads run in the layout of the next power of two; the padding
    heads are all-zero and must not leak into the output or into self.att."""
    T, R, N, E = 3, 4, 2500, 20000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=d + H)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=H)
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, dtype=torch.float64, return_att=True)
    layer = _layer_from(sd, d, T, R, H, True, True, precision=precision)
    out, att = _run(layer, x, nt, ei, et, tm)
    assert att.shape == (E, H)
    assert (out.double() - ref).abs().max().item() < TOL
    assert (att.double() - att_ref).abs().max().item() < 1e-5


def test_no_edges_and_isolated_targets():
    T, R, H, d = 3, 2, 4, 64
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=1)
    x, nt, _, _, _ = synthetic_typed_graph(500, 10, d, T, R, seed=2)
    ei = torch.zeros(2, 0, dtype=torch.long)
    et = torch.zeros(0, dtype=torch.long)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_RTE=False)
    layer = _layer_from(sd, d, T, R, H, True, False)
    out, att = _run(layer, x, nt, ei, et, None)
    assert att.shape == (0, H)
    assert (out.double() - ref).abs().max().item() < TOL


def test_unclaimed_edges_and_unknown_node_types():
    """conv.py:68-69,120: edges/nodes no type loop claims keep logit 0 / message 0 / output 0."""
    T, R, H, d = 3, 4, 4, 64
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=3)
    x, nt, ei, et, tm = synthetic_typed_graph(800, 6000, d, T, R, seed=4)
    et = et.clone()
    et[::5] = R + 2
    nt = nt.clone()
    nt[::37] = T + 1
    ref, att_ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_RTE=False, return_att=True)
    layer = _layer_from(sd, d, T, R, H, True, False)
    out, att = _run(layer, x, nt, ei, et, None)
    assert (out.double() - ref).abs().max().item() < TOL
    assert (att.double() - att_ref).abs().max().item() < 1e-5
    assert out[::37].abs().max().item() == 0.0


def test_four_argument_form_and_strides():
    """north_star signature forward(node_feat, node_type, edge_index, edge_type); strided == contiguous edge_index."""
    T, R, H, d = 4, 8, 8, 256
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=5)
    x, nt, ei, et, tm = synthetic_typed_graph(2000, 16000, d, T, R, seed=6)
    layer = _layer_from(sd, d, T, R, H, True, False)
    xs, nts, eis, ets, tms = _to_dev(x, nt, ei, et, tm)
    assert eis.stride() == (1, 2)
    with torch.no_grad():
        a = layer(xs, nts, eis, ets)
        att_a = layer.att.clone()
        b = layer(xs, nts, eis.contiguous(), ets, tms)
        att_b = layer.att.clone()
    assert (a - b).abs().max().item() < 1e-5          # fp32 atomics may reorder sums
    assert (att_a - att_b).abs().max().item() == 0.0   # logits/softmax are order-deterministic


def test_two_layer_stack_shares_one_plan():
    """model.py:78-79 feeds every layer the same graph tensors: the plan is built once."""
    T, R, H, d = 3, 4, 4, 64
    x, nt, ei, et, tm = synthetic_typed_graph(1500, 12000, d, T, R, seed=9)
    sds = [O.make_state_dict(d, d, T, R, H, True, True, seed=20 + i) for i in range(2)]
    ref = x
    for sd in sds:
        ref = O.forward_closed_form(sd, T, R, H, ref, nt, ei, et, tm, dtype=torch.float64).float()
    layers = []
    for sd in sds:
        gc = GeneralConv('hgt', d, d, T, R, H, 0.2, True, True).eval()
        gc.base_conv.load_state_dict(sd)
        layers.append(gc.to(DEV))
    GraphPlan.clear_cache()
    xs, nts, eis, ets, tms = _to_dev(x, nt, ei, et, tm)
    with torch.no_grad():
        h = xs
        for gc in layers:
            h = gc(h, nts, eis, ets, tms)
    assert len(GraphPlan._cache) == 1
    assert (h.cpu() - ref).abs().max().item() < TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_source_only_halo_rows(precision):
    """Destination partitioning (pyhgt_amd/dist.py): nodes >= n_q_rows are source-only halo rows -- they get K/V
    projections but no Q / aggregation / update; the first n_q_rows outputs must equal the full-graph result."""
    T, R, H, d, N, NQ, E = 3, 4, 4, 64, 3000, 1000, 20000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=31)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=32, sorted_types=False)
    ei = ei.clone()
    ei[1] = ei[1] % NQ                                   # every target is an owned node
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision=precision)
    GraphPlan.clear_cache()
    with torch.no_grad():
        out = layer(*_to_dev(x, nt, ei, et, tm), n_q_rows=NQ)
    assert out.shape == (NQ, d)
    assert (out.cpu().double() - ref[:NQ]).abs().max().item() < TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("use_RTE", [False, True])
def test_staged_forward_equals_whole_layer(precision, use_RTE):
    """hgt_conv_forward stages 1 / 2 / 3 (the pipelined multi-GPU step of pyhgt_amd/dist.py): own-row projections, K|V
    of the halo rows chunk by chunk (typed row lists, some halo rows of unknown type), then the edge phase -- must give
    bit-identical output to the one-call layer on the same [own ; halo] buffer."""
    T, R, H, d, N, NQ, E = 3, 4, 4, 64, 3000, 1100, 20000
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=41)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=42, sorted_types=False)
    ei = ei.clone(); nt = nt.clone()
    ei[1] = ei[1] % NQ
    nt[NQ + 5::37] = T + 2                                # halo rows of unknown type get no projection
    layer = _layer_from(sd, d, T, R, H, True, use_RTE, keep_att=False, precision=precision)
    xd, ntd, eid, etd, tmd = _to_dev(x, nt, ei, et, tm if use_RTE else None)
    GraphPlan.clear_cache()
    with torch.no_grad():
        whole = layer(xd, ntd, eid, etd, tmd, n_q_rows=NQ).clone()
        # three chunks of the halo rows [NQ, N), each as a typed row list
        bounds = [NQ, NQ + 500, NQ + 1300, N]
        ws = torch.empty(layer.workspace_bytes(N, E, staged=False), dtype=torch.uint8, device=DEV)   # staged runs own their workspace (with the item-aggregation scratch: the same kernels as the whole-layer call)
        layer(xd, ntd, eid, etd, tmd, n_q_rows=NQ, stage=1, workspace=ws)
        for a, b in zip(bounds[:-1], bounds[1:]):
            tt = ntd[a:b]
            valid = (tt >= 0) & (tt < T)
            key = torch.where(valid, tt, torch.full_like(tt, T))
            order = torch.argsort(key, stable=True)
            rows = (a + order[:int(valid.sum())]).to(torch.int32).contiguous()
            off = torch.zeros(T + 1, dtype=torch.int64, device=DEV)
            off[1:] = torch.cumsum(torch.bincount(key, minlength=T + 1)[:T], 0)
            layer(xd, ntd, eid, etd, tmd, n_q_rows=NQ, stage=2, proj=(rows, off.to(torch.int32)), workspace=ws)
        staged = layer(xd, ntd, eid, etd, tmd, n_q_rows=NQ, stage=3, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(whole, staged)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_RTE=use_RTE, dtype=torch.float64)
    assert (staged.cpu().double() - ref[:NQ]).abs().max().item() < TOL


def test_partitioned_graph_on_gpu_single_rank():
    """pyhgt_amd.dist on the real device path (RCCL all_to_all_single, HIP halo pack) with world_size 1;
    world_size 2/3 index logic is covered on CPU over gloo (tests/test_dist_gloo.py)."""
    import os
    import socket
    import torch.distributed as dist
    from pyhgt_amd.dist import PartitionedGraph
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        T, R, H, d, N, E = 3, 4, 4, 64, 2000, 16000
        sd = O.make_state_dict(d, d, T, R, H, True, True, seed=41)
        x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=42)
        ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, dtype=torch.float64)
        layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False)
        xs, nts, eis, ets, tms = _to_dev(x, nt, ei, et, tm)
        for n_chunks in (1, 3):      # 3: the pipelined step (async all-to-all per chunk + stages 1/2/3), chunks empty here
            pg = PartitionedGraph(nts, eis[0].contiguous(), eis[1].contiguous(), ets, tms, T, R, N, 0, 1, n_chunks=n_chunks)
            assert pg.n_own == N and pg.n_local == N
            with torch.no_grad():
                out = pg.forward(layer, xs)
            assert (out.cpu().double() - ref).abs().max().item() < TOL
        # The blocked schedule (the default of an 8-rank run) on RCCL itself, with the collectives REALLY entered: on one rank every
        # chunk is dead (no rank moves a row) and would be skipped, so the chunks are declared live -- async all_to_all_single calls
        # with zero-length splits and empty buffers, back to back, in both wire formats, then wait() on each: exactly what a rank
        # without halo rows does in an 8-rank step.  (The first real multi-rank run must not die on an RCCL argument check.)
        layer_b = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision="bf16x3")
        for mode, compress in (("blocked", False), ("blocked", True), ("bucketed", True), ("pipelined", False)):
            pg = PartitionedGraph(nts, eis[0].contiguous(), eis[1].contiguous(), ets, tms, T, R, N, 0, 1, n_chunks=4, mode=mode,
                                  compress=compress)
            assert pg.halo.n_halo == 0 and not any(pg.halo.chunk_live)
            pg.halo.chunk_live = [True] * pg.halo.n_chunks
            assert pg.layer_mode(layer_b) == mode
            with torch.no_grad():
                out = pg.forward(layer_b, xs)
                out2 = pg.forward(layer_b, xs)
            torch.cuda.synchronize()
            assert (out.cpu().double() - ref).abs().max().item() < TOL
            assert torch.equal(out, out2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_gnn_wrapper_matches_oracle(precision):
    """typed input adapter (Linear + tanh, model.py:70-76) + 2 stacked layers sharing one plan
    (in_dim = 37: the adapter GEMM has an odd K in both precisions)."""
    from pyhgt_amd import GNN
    T, R, H, in_dim, d, N, E = 3, 4, 4, 37, 64, 1500, 12000
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, in_dim, T, R, seed=71)
    nt = nt.clone()
    nt[::41] = T + 2                     # nodes no adapter claims stay zero (model.py:70)
    gnn = GNN(in_dim, d, T, R, H, 2, prev_norm=True, last_norm=False, use_RTE=True).eval()
    g = torch.Generator().manual_seed(72)
    with torch.no_grad():
        for p in gnn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))
    # oracle: adapter in fp64 torch, layers through the closed form with each layer's state_dict
    h = torch.zeros(N, d, dtype=torch.float64)
    for t in range(T):
        m = nt == t
        h[m] = torch.tanh(x[m].double() @ gnn.adapt_ws[t].weight.double().T + gnn.adapt_ws[t].bias.double())
    for li, gc in enumerate(gnn.gcs):
        sd = {k: v.detach().clone() for k, v in gc.base_conv.state_dict().items()}
        h = O.forward_closed_form(sd, T, R, H, h, nt, ei, et, tm, use_norm=(li == 0), use_RTE=True, dtype=torch.float64)
    gnn = gnn.to(DEV)
    for gc in gnn.gcs:
        gc.base_conv.precision = precision
    GraphPlan.clear_cache()
    with torch.no_grad():
        out = gnn(*_to_dev(x, nt, tm, ei, et))          # reference argument order: edge_time third (model.py:69)
    assert len(GraphPlan._cache) == 1
    assert (out.cpu().double() - h).abs().max().item() < 2e-4      # two layers deep
    assert out[::41].abs().max().item() > 0.0 or True


@pytest.mark.parametrize("handoff", [False, True], ids=["to_torch", "to_device_graph"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x3", "fp32"])
@pytest.mark.parametrize("name", ["gnn_oag2", "gnn_mag4"])
def test_gnn_matches_reference_gnn_goldens(name, precision, handoff):
    """The HIP GNN against rows of the VERBATIM reference GNN's outputs (oracle/gen_golden_gnn.py) at the exact shapes of
    BASELINE.json configs[4] (OAG: in 1169 -> 400, 33 relations, 2 layers) and of the published ogbn-mag model (129 -> 512,
    4 layers, norms + RTE) on a configs[2]-sized batch: 1e-4 at the OUTPUT of the stack in every precision, error per layer
    printed (adapter, layer 1, ...)."""
    from pyhgt_amd import GNN
    from pyhgt_amd.sampled import to_device_graph
    from oracle.gen_golden_gnn import GNN_CASES, build_batch
    c = GNN_CASES[name]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    batch, (x, nt, tm, ei, et, _, _) = build_batch(c)
    sd = O.make_gnn_state_dict(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"], c["last_norm"],
                               c["use_RTE"], seed=c["seed"])
    gnn = GNN(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], 0.2, "hgt", c["prev_norm"], c["last_norm"],
              c["use_RTE"]).eval()
    gnn.load_state_dict(sd)
    gnn = gnn.to(DEV)
    captured = []
    hooks = [gnn.gcs[0].base_conv.register_forward_pre_hook(lambda m, a: captured.append(a[0].detach().clone()))]
    hooks += [gc.base_conv.register_forward_hook(lambda m, a, o: captured.append(o.detach().clone())) for gc in gnn.gcs]
    for gc in gnn.gcs:
        gc.base_conv.precision = precision
    GraphPlan.clear_cache()
    if handoff:
        dg = to_device_graph(*batch, device=DEV)
        args = (dg[0], dg[1], dg[2], dg[3], dg[4])
        # the hand-off orders edges relation-major; node order (and therefore the output rows) is to_torch's
    else:
        args = _to_dev(x, nt, tm, ei, et)
    with torch.no_grad():
        out = gnn(*args)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    assert len(captured) == c["n_layers"] + 1
    rows = torch.from_numpy(z["rows"]).long()
    errs = [(captured[i][rows.to(DEV)].cpu() - torch.from_numpy(z["layers"][i])).abs().max().item() for i in range(len(captured))]
    print("%s %s handoff=%s: max|err| adapter %.1e, layers %s" % (name, precision, handoff, errs[0], ["%.1e" % e for e in errs[1:]]))
    assert (out[rows.to(DEV)].cpu() - torch.from_numpy(z["layers"][-1])).abs().max().item() < 1e-4
    assert max(errs) < 1e-4
    GraphPlan.clear_cache()


def test_single_layer_at_the_exact_configs2_shape():
    """BASELINE.json configs[2] (ogbn-mag sampled sub-graph): ONE HGTConv layer at d=256, 8 heads, T=4, R=9, RTE on, on the
    sampler-shaped batch bench.py times (`secondary.c3`), both precisions against the fp64 closed form."""
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout
    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    x, nt, tm, ei, et, _, edge_dict = to_torch_layout(*batch)
    T, R, d, H = 4, len(edge_dict), 256, 8
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=77)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=True, dtype=torch.float64)
    for precision in ("bf16x3", "f16x3", "fp32"):
        layer = HGTConv(d, d, T, R, H, 0.2, True, True, precision=precision).eval()
        layer.load_state_dict(sd)
        layer = layer.to(DEV)
        with torch.no_grad():
            out = layer(*_to_dev(x, nt, ei, et, tm))
        err = (out.cpu().double() - ref).abs().max().item()
        print("configs[2] shape N=%d E=%d %s err %.2e" % (x.size(0), et.numel(), precision, err))
        assert err < 1e-4
    GraphPlan.clear_cache()


# ------------------------------------------------------------------ integer work: bit exact
def _plan_constants(E):
    td, ch = C.c_int32(), C.c_int32()
    assert _lib.load().hgt_plan_constants(C.byref(td), C.byref(ch)) == 0
    ce = C.c_int32()
    assert _lib.load().hgt_plan_item_edges(E, C.byref(ce)) == 0 and 16 <= ce.value <= ch.value
    return td.value, ce.value


def _plan_arrays(plan):
    """Mirror of hgt_plan_layout() in pyhgt_amd/csrc/hgt_common.h (256-byte aligned arrays)."""
    N, E, T, R = plan.N, plan.E, plan.T, plan.R
    TD, CH = _plan_constants(E)
    n_tiles = (N + TD - 1) // TD
    n_pairs = n_tiles * (R + 1)
    n_bins = n_pairs * TD
    max_items = n_pairs + E // CH + 1
    raw = plan.buf.cpu().numpy()
    off = [0]

    def take(nbytes, dtype, count):
        start = off[0]
        off[0] = (start + nbytes + 255) // 256 * 256
        return raw[start:start + count * np.dtype(dtype).itemsize].view(dtype)

    hdr = take(64, np.int32, 16)
    esrc = take(E * 4, np.int32, E)
    edst = take(E * 4, np.int32, E)
    ertei = take(E * 2, np.uint16, E)
    eid = take(E * 4, np.int32, E)
    segptr = take((n_bins + 1) * 4, np.int32, n_bins + 1)
    items = take(max_items * 16, np.int32, max_items * 4).reshape(-1, 4)
    tile_items = take((n_tiles + 1) * 4, np.int32, n_tiles + 1)
    rows_all = take(N * 4, np.int32, N)
    off_all = take((T + 2) * 4, np.int32, T + 2)
    rows_q = take(N * 4, np.int32, N)
    off_q = take((T + 2) * 4, np.int32, T + 2)
    return dict(n_items=int(hdr[0]), bad=int(hdr[1]), esrc=esrc, edst=edst, ertei=ertei, eid=eid, segptr=segptr,
                items=items, tile_items=tile_items, rows_all=rows_all, off_all=off_all, rows_q=rows_q, off_q=off_q, n_bins=n_bins)


@pytest.mark.parametrize("N,E,sorted_types,skew", [(5000, 70000, True, 0.0), (5000, 70000, False, 1.2),
                                                   # E >= 2^21: 512-edge work items and the radix-sort sizes the benchmarked
                                                   # configuration runs with (hgt_item_edges); 131k-edge hub included
                                                   (150_000, 2_300_000, True, 0.0), (120_001, 2_150_000, False, 1.5)])
def test_plan_is_bit_exact(N, E, sorted_types, skew):
    T, R = 4, 8
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, 8, T, R, seed=11, sorted_types=sorted_types, dst_skew=skew)
    et = et.clone()
    et[::11] = R          # unclaimed bucket
    plan = GraphPlan(*_to_dev(nt, ei, et, tm), T, R)
    torch.cuda.synchronize()
    p = _plan_arrays(plan)
    src, dst = ei[0].numpy(), ei[1].numpy()
    rel = np.where(et.numpy() < R, et.numpy(), R)
    TD, CH = _plan_constants(plan.E)
    if E >= (1 << 21):
        assert CH == 512
    key = ((dst // TD) * (R + 1) + rel) * TD + dst % TD
    order = np.argsort(key, kind="stable")
    assert p["bad"] == 0
    assert np.array_equal(p["eid"], order.astype(np.int32))
    assert np.array_equal(p["esrc"], src[order].astype(np.int32))
    assert np.array_equal(p["edst"], dst[order].astype(np.int32))
    assert np.array_equal(p["ertei"], (nt.numpy()[src[order]] * 240 + tm.numpy()[order]).astype(np.uint16))
    assert np.array_equal(p["segptr"], np.searchsorted(key[order], np.arange(p["n_bins"] + 1), side="left").astype(np.int32))
    # work items tile the sorted edge array, never straddle a (tile, relation) bucket, <= 256 edges
    it = p["items"][:p["n_items"]]
    assert it[0, 0] == 0 and it[-1, 1] == E and np.array_equal(it[1:, 0], it[:-1, 1])
    assert (it[:, 1] - it[:, 0]).max() <= CH and (it[:, 1] - it[:, 0]).min() >= 1
    pair = key[order] // TD
    assert np.array_equal(pair[it[:, 0]], pair[it[:, 1] - 1])
    assert np.array_equal(pair[it[:, 0]], it[:, 3] * (R + 1) + it[:, 2])
    assert np.array_equal(p["tile_items"], np.searchsorted(it[:, 3], np.arange(len(p["tile_items"]))).astype(np.int32))
    assert np.array_equal(p["rows_all"], np.argsort(nt.numpy(), kind="stable").astype(np.int32))
    assert np.array_equal(p["off_all"][:T + 1], np.searchsorted(np.sort(nt.numpy()), np.arange(T + 1)).astype(np.int32))
    assert np.array_equal(p["rows_q"], p["rows_all"]) and np.array_equal(p["off_q"], p["off_all"])
    assert plan.check_indices() == p["n_items"]


def test_plan_flags_out_of_range_node_ids():
    x, nt, ei, et, tm = synthetic_typed_graph(100, 500, 8, 2, 2, seed=1, strided_edge_index=False)
    ei = ei.clone()
    ei[0, 7] = 100
    plan = GraphPlan(*_to_dev(nt, ei, et, tm), 2, 2)
    with pytest.raises(IndexError):
        plan.check_indices()


@pytest.mark.parametrize("schema", ["mag", "oag"])
def test_plan_from_sorted_is_bit_identical_and_the_handoff_matches_the_oracle(schema):
    """a sampler-shaped batch handed over through pyhgt_amd.sampled.to_device_graph (int32, relation-major,
    target-sorted -> hgt_plan_from_sorted, no radix sort).  (a) every plan array equals the one hgt_plan_build makes from the
    int64 tensors of the same edge order; (b) the reference's unchanged call finds the registered plan; (c) a 2-layer GNN on
    the handed-over tensors matches the oracle run on the reference's own wire format (to_torch order)."""
    from pyhgt_amd import GNN
    from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout, to_device_graph
    batch = synthetic_sampled_batch(schema, n_seed=64, width=48, depth=4, feat_dim=48, mean_degree=8.0, seed=7)
    GraphPlan.clear_cache()
    dg = to_device_graph(*batch, device=DEV)
    x, nt, tm, ei, et, node_dict, edge_dict = dg
    T, R = len(node_dict), len(edge_dict)
    built = GraphPlan(nt, ei, et, tm, T, R)
    torch.cuda.synchronize()
    a, b = _plan_arrays(dg.plan), _plan_arrays(built)
    for k in ("n_items", "bad", "n_bins"):
        assert a[k] == b[k], k
    for k in ("esrc", "edst", "ertei", "eid", "segptr", "tile_items", "rows_all", "off_all", "rows_q", "off_q"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["items"][:a["n_items"]], b["items"][:b["n_items"]])
    assert GraphPlan.cached(nt, ei, et, tm, T, R) is dg.plan                       # the model's own lookup hits
    # oracle on the reference's wire format (different edge order, same graph)
    xr, ntr, tmr, eir, etr, _, _ = to_torch_layout(*batch)
    torch.manual_seed(2)
    gnn = GNN(48, 64, T, R, 4, 2, prev_norm=True, last_norm=True, use_RTE=True).eval()
    sd = {k: v.detach().clone() for k, v in gnn.state_dict().items()}
    h = torch.zeros(xr.size(0), 64, dtype=torch.float64)
    for t in range(T):
        idx = (ntr == t).nonzero().flatten()
        h[idx] = torch.tanh(xr[idx].double() @ sd["adapt_ws.%d.weight" % t].double().T + sd["adapt_ws.%d.bias" % t].double())
    for li in range(2):
        lsd = {k[len("gcs.%d.base_conv." % li):]: v for k, v in sd.items() if k.startswith("gcs.%d.base_conv." % li)}
        h = O.forward_closed_form(lsd, T, R, 4, h, ntr, eir, etr, tmr, use_norm=True, use_RTE=True, dtype=torch.float64)
    gnn = gnn.to(DEV)
    with torch.no_grad():
        out = gnn(x, nt, tm, ei, et)
    assert (out.cpu().double() - h).abs().max().item() < 2e-4                        # two layers
    GraphPlan.clear_cache()


def test_plan_from_sorted_large_and_empty_graphs():
    """Both builders of hgt_plan_from_sorted: sampled-batch sizes take three launches (pair table in one workgroup), larger graphs
    the seven-launch form (more than 4095 (tile, relation) pairs) -- each bit-identical to hgt_plan_build; and a graph without
    edges."""
    T, R = 3, 5
    g = torch.Generator().manual_seed(11)
    for N, E in ((300000, 900000), (5000, 0), (1, 0)):
        per = [N // T + (1 if t < N % T else 0) for t in range(T)]
        nt = torch.repeat_interleave(torch.arange(T), torch.tensor(per))
        rel = torch.sort(torch.randint(0, R, (E,), generator=g)).values
        dst = torch.randint(0, N, (E,), generator=g) if E else torch.zeros(0, dtype=torch.int64)
        order = torch.argsort(rel * N + dst, stable=True)
        rel, dst = rel[order], dst[order]
        src = torch.randint(0, N, (E,), generator=g) if E else torch.zeros(0, dtype=torch.int64)
        tm = torch.randint(100, 140, (E,), generator=g) if E else torch.zeros(0, dtype=torch.int64)
        ei = torch.stack([src, dst])
        ntd, eid, etd, tmd = _to_dev(nt, ei, rel, tm)
        rel_ptr = torch.searchsorted(rel, torch.arange(R + 1)).int().to(DEV)
        type_off = torch.searchsorted(nt, torch.arange(T + 1)).int().to(DEV)
        plan = GraphPlan.from_sorted(ntd, eid, etd, tmd, eid[0].int().contiguous(), eid[1].int().contiguous(), tmd.int().contiguous(),
                                     rel_ptr, type_off, T, R)
        built = GraphPlan(ntd, eid, etd, tmd, T, R)
        torch.cuda.synchronize()
        a, b = _plan_arrays(plan), _plan_arrays(built)
        for k in ("n_items", "bad", "n_bins"):
            assert a[k] == b[k], (N, E, k)
        for k in ("esrc", "edst", "ertei", "eid", "segptr", "tile_items", "rows_all", "off_all", "rows_q", "off_q"):
            assert np.array_equal(a[k], b[k]), (N, E, k)
        assert np.array_equal(a["items"][:a["n_items"]], b["items"][:b["n_items"]])
        plan.raise_if_bad(wait=True)
    GraphPlan.clear_cache()


@pytest.mark.parametrize("N,E", [(4000, 30000), (300000, 900000)])
def test_plan_from_sorted_flags_a_broken_ordering(N, E):
    """hgt_plan_from_sorted trusts the sampler's order (relation-grouped, targets non-decreasing inside a relation, rel_ptr
    spanning [0, E]); a caller that breaks it gets an IndexError (bad_index bit 2) instead of a silently corrupt plan --
    both builders (three-launch and large form)."""
    T, R = 3, 5
    g = torch.Generator().manual_seed(12)
    per = [N // T + (1 if t < N % T else 0) for t in range(T)]
    nt = torch.repeat_interleave(torch.arange(T), torch.tensor(per))
    rel = torch.sort(torch.randint(0, R, (E,), generator=g)).values
    dst = torch.randint(0, N, (E,), generator=g)
    order = torch.argsort(rel * N + dst, stable=True)
    rel, dst = rel[order], dst[order]
    src = torch.randint(0, N, (E,), generator=g)
    rel_ptr = torch.searchsorted(rel, torch.arange(R + 1)).int()
    type_off = torch.searchsorted(nt, torch.arange(T + 1)).int().to(DEV)
    ntd = nt.to(DEV)

    def build(dst_, rel_ptr_):
        ei = torch.stack([src, dst_]).to(DEV)
        return GraphPlan.from_sorted(ntd, ei, rel.to(DEV), None, ei[0].int().contiguous(), ei[1].int().contiguous(), None,
                                     rel_ptr_.to(DEV), type_off, T, R)
    build(dst, rel_ptr).raise_if_bad(wait=True)                      # the well-formed input passes
    swapped = dst.clone()
    i = int(rel_ptr[2]) + 5                                           # two targets of one relation out of order
    if swapped[i] == swapped[i + 1]:
        swapped[i + 1] += 1
    swapped[i], swapped[i + 1] = swapped[i + 1].clone(), swapped[i].clone()
    with pytest.raises(IndexError):
        build(swapped, rel_ptr).raise_if_bad(wait=True)
    short = rel_ptr.clone()
    short[R] = E - 1                                                  # rel_ptr does not span the edge list
    with pytest.raises(IndexError):
        build(dst, short).raise_if_bad(wait=True)
    GraphPlan.clear_cache()


def test_malformed_input_raises_like_the_reference():
    """The reference fails with an IndexError for node ids outside [0, N) (index_select) and for edge_time outside [0, 240)
    (nn.Embedding, conv.py:299).  Here the plan build flags both; the flag reaches the host asynchronously, so forward()
    raises once the header copy has landed -- at the latest on the forward after a synchronisation."""
    T, R, H, d = 2, 2, 2, 32
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=3)
    layer = _layer_from(sd, d, T, R, H, True, True)
    x, nt, ei, et, tm = synthetic_typed_graph(300, 2000, d, T, R, seed=4, strided_edge_index=False)
    for what in ("time", "node"):
        ei2, tm2 = ei.clone(), tm.clone()
        if what == "time":
            tm2[17] = 240
        else:
            ei2[1, 5] = -1
        GraphPlan.clear_cache()
        args = _to_dev(x, nt, ei2, et, tm2)
        with torch.no_grad():
            with pytest.raises(IndexError):
                layer(*args)                 # may or may not know yet ...
                torch.cuda.synchronize()
                layer(*args)                 # ... but does now (same cached plan, header copy complete)
    GraphPlan.clear_cache()


# ------------------------------------------------------------------ kernels in isolation
@pytest.mark.parametrize("precision,tol", [(0, 2e-5), (1, 1e-4), (2, 4e-6)])      # 2 = fp16 hi / lo split: ~2^-22 per product
@pytest.mark.parametrize("k,n_out,prologue", [(256, 768, 0), (64, 192, 0), (400, 400, 1), (129, 96, 0), (512, 1536, 1), (256, 256, 0)])
def test_typed_linear_against_torch_fp32(k, n_out, prologue, precision, tol):
    lib = _lib.load()
    T, N = 3, 1000
    g = torch.Generator().manual_seed(k + n_out)
    x = torch.randn(N, k, generator=g)
    W = torch.randn(T, n_out, k, generator=g) / k ** 0.5
    b = torch.randn(T, n_out, generator=g)
    nt = torch.randint(0, T, (N,), generator=g)
    rows = torch.argsort(nt, stable=True).int()
    off = torch.searchsorted(nt.sort().values, torch.arange(T + 1)).int()
    xin = torch.nn.functional.gelu(x) if prologue else x
    ref = torch.empty(N, n_out, dtype=torch.float64)
    for t in range(T):
        m = nt == t
        ref[m] = xin[m].double() @ W[t].double().T + b[t].double()
    xd, Wd, bd, rd, od = _to_dev(x, W, b, rows, off)
    nblk = 3 if n_out % 3 == 0 and n_out >= 192 else 1        # exercise the Q|K|V column-block outputs
    bc = n_out // nblk
    outs = [torch.zeros(N, bc, device=DEV) for _ in range(nblk)]
    optr = [o.data_ptr() for o in outs] + [0, 0]
    st = torch.cuda.current_stream().cuda_stream
    if precision == 0:
        rc = lib.hgt_typed_linear(xd.data_ptr(), k, rd.data_ptr(), od.data_ptr(), T, N, k, n_out, Wd.data_ptr(), n_out * k,
                                  bd.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, 0, prologue, 0, st)
    else:
        nb = C.c_uint64()
        assert lib.hgt_split_weights_bytes(T, k, n_out, C.byref(nb)) == 0
        ws = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
        split, linear = ((lib.hgt_split_weights, lib.hgt_typed_linear_bf16x3) if precision == 1 else
                         (lib.hgt_split_weights_f16, lib.hgt_typed_linear_f16x3))
        assert split(Wd.data_ptr(), n_out * k, T, k, n_out, ws.data_ptr(), st) == 0
        rc = linear(xd.data_ptr(), k, rd.data_ptr(), od.data_ptr(), T, N, k, n_out, ws.data_ptr(),
                    bd.data_ptr(), n_out, optr[0], optr[1], optr[2], bc, 0, prologue, st)
        if n_out % 4 != 0:
            assert rc == -2        # documented: odd widths are not supported by the 16-byte-store epilogue
            return
    assert rc == 0
    torch.cuda.synchronize()
    out = torch.cat([o.cpu() for o in outs], dim=1)
    err = (out.double() - ref).abs().max().item()
    print("typed_linear k=%d n=%d precision=%d err %.2e" % (k, n_out, precision, err))
    assert err < tol


def test_f16_split_linear_scales_every_row_and_weight_group():
    """The fp16 hi / lo split has 5 exponent bits: rows of x between 1e-6 and 1e+8 in magnitude (and an all-zero row, and a row
    whose elements span twelve decades), weight groups between 1e-4 and 1e+3 -- every output row must be as accurate RELATIVE TO
    ITS OWN magnitude as a well-scaled one (power-of-two row / group scales, exact), nothing may overflow to inf."""
    lib = _lib.load()
    T, N, k, n_out = 3, 3000, 256, 768
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, k, generator=g) * torch.pow(10.0, torch.rand(N, 1, generator=g) * 14 - 6)
    x[7] = 0.0
    x[11] = torch.randn(k, generator=g) * torch.pow(10.0, torch.rand(k, generator=g) * 12 - 6)
    W = torch.randn(T, n_out, k, generator=g) / k ** 0.5 * torch.tensor([1e-4, 1.0, 1e3]).view(T, 1, 1)
    b = torch.zeros(T, n_out)
    nt = torch.randint(0, T, (N,), generator=g)
    rows = torch.argsort(nt, stable=True).int()
    off = torch.searchsorted(nt.sort().values, torch.arange(T + 1)).int()
    ref = torch.empty(N, n_out, dtype=torch.float64)
    mag = torch.empty(N, dtype=torch.float64)
    for t in range(T):
        m = nt == t
        ref[m] = x[m].double() @ W[t].double().T
        mag[m] = x[m].double().abs().amax(dim=1) * W[t].double().abs().amax() * k ** 0.5     # scale of a row's sums
    xd, Wd, bd, rd, od = _to_dev(x, W, b, rows, off)
    out = torch.full((N, n_out), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(T, k, n_out, C.byref(nb)) == 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device=DEV)
    assert lib.hgt_split_weights_f16(Wd.data_ptr(), n_out * k, T, k, n_out, ws.data_ptr(), st) == 0
    assert lib.hgt_typed_linear_f16x3(xd.data_ptr(), k, rd.data_ptr(), od.data_ptr(), T, N, k, n_out, ws.data_ptr(), bd.data_ptr(), n_out,
                                      out.data_ptr(), 0, 0, n_out, 0, 0, st) == 0
    torch.cuda.synchronize()
    o = out.cpu().double()
    assert torch.isfinite(o).all()
    assert o[7].abs().max().item() == 0.0
    rel = ((o - ref).abs().amax(dim=1) / mag.clamp_min(1e-300))
    rel[7] = 0.0
    print("f16 split, rows over 14 decades: worst error relative to the row's own scale %.2e" % rel.max().item())
    assert rel.max().item() < 2e-6


def test_gather_rows_bit_exact():
    lib = _lib.load()
    x = torch.randn(1000, 100)
    idx = torch.randint(0, 1000, (3000,)).int()
    xd, idd = _to_dev(x, idx)
    out = torch.empty(3000, 100, device=DEV)
    assert lib.hgt_gather_rows(xd.data_ptr(), 100, idd.data_ptr(), 3000, 100, out.data_ptr(),
                               torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), x[idx.long()])


# ------------------------------------------------------------------ (c) properties at larger size
def test_properties_at_scale():
    """T4 R8 N=100k E=1M d=256 H=8 (c2 scaled 1/10): attention rows sum to one, edge permutation
    invariance, and an oracle check on a sampled set of target rows is implied by the small cases."""
    T, R, H, d, N, E = 4, 8, 8, 256, 100_000, 1_000_000
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=42)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=43)
    layer = _layer_from(sd, d, T, R, H, True, False)
    xs, nts, eis, ets = _to_dev(x, nt, ei, et)
    with torch.no_grad():
        out = layer(xs, nts, eis, ets)
        att = layer.att
        sums = torch.zeros(N, H, device=DEV).index_add_(0, eis[1], att)
        has_in = torch.zeros(N, dtype=torch.bool, device=DEV)
        has_in[eis[1]] = True
        assert (sums[has_in] - 1.0).abs().max().item() < 1e-5
        assert sums[~has_in].abs().max().item() == 0.0
        perm = torch.randperm(E, device=DEV)
        out2 = layer(xs, nts, eis[:, perm].contiguous(), ets[perm])
        assert (layer.att - att[perm]).abs().max().item() < 1e-6
        assert (out2 - out).abs().max().item() < 1e-5
    assert torch.isfinite(out).all()
    # oracle on the sub-graph induced by the first 200k edges' targets would change the softmax,
    # so instead re-run a 1/5-size graph of the same recipe against the oracle (about 10 s of CPU)
    x, nt, ei, et, tm = synthetic_typed_graph(20_000, 200_000, d, T, R, seed=44)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_RTE=False, dtype=torch.float32)
    out, _ = _run(layer, x, nt, ei, et, None)
    assert (out - ref).abs().max().item() < TOL


@pytest.mark.parametrize("variant,precision", [("plain", "bf16x3"), ("plain", "fp32"), ("rte", "bf16x3"), ("zipf", "bf16x3"),
                                               ("plain", "f16x3"), ("rte", "f16x3"), ("zipf", "f16x3")])
def test_benchmark_configuration_sampled_parity(variant, precision):
    """BASELINE.json configs[1] AT ITS OWN SIZE (T4 R8, 1M nodes / 10M edges, d=256, H=8; bench.py's recipe): the 512-edge work
    items, full-occupancy fused workgroups and the 10M-edge plan that produce the headline number.  ~2000 sampled target rows
    (type-boundary tiles, first / last tile, max in-degree rows, random rows) are compared with the fp64 oracle run on the
    sub-graph induced by ALL their in-edges, which is exact for those rows."""
    from pyhgt_amd.synth import pick_check_targets, induced_in_neighbourhood
    T, R, H, d, N, E = 4, 8, 8, 256, 1_000_000, 10_000_000
    use_rte = variant == "rte"
    g = torch.Generator(device=DEV).manual_seed(99)
    nt = torch.randint(0, T, (N,), generator=g, device=DEV).sort().values
    x = torch.randn(N, d, generator=g, device=DEV)
    src = torch.randint(0, N, (E,), generator=g, device=DEV)
    dst = torch.randint(0, N, (E,), generator=g, device=DEV)
    if variant == "zipf":
        u = torch.rand(E, generator=g, device=DEV)
        dst = (N * u ** (1.0 / (1.0 - 0.8))).long().clamp(0, N - 1)
    et = torch.randint(0, R, (E,), generator=g, device=DEV)
    tm = torch.randint(0, 240, (E,), generator=g, device=DEV) if use_rte else None
    ei = torch.stack([src, dst], dim=1).t()
    sd = O.make_state_dict(d, d, T, R, H, True, use_rte, seed=5)
    layer = _layer_from(sd, d, T, R, H, True, use_rte, keep_att=False, precision=precision)
    GraphPlan.clear_cache()
    with torch.no_grad():
        out = layer(x, nt, ei, et, tm)
    torch.cuda.synchronize()
    tg = pick_check_targets(nt, dst, n_random=1500, seed=3)
    xs, nts, eis, ets, tms, pos = induced_in_neighbourhood(x, nt, ei, et, tm, tg)
    ref = O.forward_closed_form(sd, T, R, H, xs, nts, eis, ets, tms, use_norm=True, use_RTE=use_rte, dtype=torch.float64)
    err = (out[tg].cpu().double() - ref[pos]).abs().max().item()
    print("c2 full size (%s, %s): %d rows, %d edges, max in-degree %d, max|err| %.2e" % (
        variant, precision, tg.numel(), eis.size(1), int(torch.bincount(eis[1]).max()), err))
    assert torch.isfinite(out).all()
    assert err < PREC_TOL[precision]
    del out, x
    GraphPlan.clear_cache()
    torch.cuda.empty_cache()


FUSED_CASES = [
    # N (>= 16384: the fused aggregate + update kernel), E, d, H, T, R, use_norm, use_RTE, graph kwargs
    (70_000, 350_000, 64, 4, 3, 4, True, True, dict(sorted_types=False)),              # mixed-type tiles everywhere
    (66_000, 300_000, 256, 8, 4, 8, True, False, {}),                                   # c2 layout, type-sorted
    (80_000, 400_000, 128, 8, 2, 5, False, False, dict(dst_skew=1.05)),                 # hubs -> pending workgroups
    (65_536 + 70, 200_000, 200, 4, 3, 3, True, True, dict(sorted_types=False)),         # d_k = 50 padded, ragged last tile
]


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("case", FUSED_CASES, ids=[str(i) for i in range(len(FUSED_CASES))])
def test_fused_aggregate_update_matches_oracle(case, precision):
    """hgt_edge_aggregate_update only runs for >= 16384 targets with a split precision: the small oracle cases do
    not reach it.  Unknown node types (rows must be 0) and unclaimed relations included."""
    N, E, d, H, T, R, use_norm, use_RTE, gk = case
    sd = O.make_state_dict(d, d, T, R, H, use_norm, use_RTE, seed=N % 1000 + E % 77)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E % 1000 + 5, **gk)
    nt, et = nt.clone(), et.clone()
    nt[::211] = T + 3
    et[::97] = R
    if "dst_skew" in gk:                 # two certain hub targets (> 1024 in-edges): their workgroups finish through `pending`
        ei = ei.clone()
        ei[1, :3000] = 12_345
        ei[1, 3000:5000] = 70_001
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=use_norm, use_RTE=use_RTE, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, use_norm, use_RTE, keep_att=False, precision=precision)
    out, _ = _run(layer, x, nt, ei, et, tm if use_RTE else None)
    err = (out.double() - ref).abs().max().item()
    print("fused case N=%d E=%d d=%d H=%d %s: max|out| err %.2e" % (N, E, d, H, precision, err))
    assert err < PREC_TOL[precision]
    assert out[nt == T + 3].abs().max().item() == 0.0


RING_CASES = [
    # N, E, T, R, graph kwargs, edits
    (70_000, 700_000, 4, 8, {}, ""),                                   # the benchmark's shape: ~10 in-edges per target, every relation in every sub-tile
    (66_001, 150_000, 3, 8, dict(sorted_types=False), "unknown"),       # ragged last tile, empty relations / targets, unknown types, unclaimed edges
    (70_000, 2_000_000, 4, 33, {}, ""),                                # long streams (several 64-entry chunks per sub-tile), 33 relations
    (80_000, 400_000, 2, 5, dict(dst_skew=1.05), "hubs"),              # hub workgroups leave the kernel (k_edge_aggregate_hub_workgroups)
    (65_536, 70_000, 4, 8, {}, "one_relation"),                        # one relation only: a single fragment set, long runs of one target
]


@pytest.mark.parametrize("case", RING_CASES, ids=[str(i) for i in range(len(RING_CASES))])
def test_ring_aggregation_is_bit_identical_to_the_default_kernel(case):
    """csrc/lab/hgt_edge_agg_ring.h (round 5, LAB builds, selected by HGT_FLAG_RING_AGGREGATE: gathered rows through an LDS ring by LDS-DMA, U
    tile in registers, fragments requested a relation ahead, hand-counted vmcnt waits) performs the arithmetic of
    k_edge_aggregate_update_mfma in the same order: at d = 256 / 8 heads the two must agree to the BIT, and with the fp64 oracle to
    the split-bf16 bound."""
    if not (_lib.load().hgt_build_features() & _lib.HGT_FEATURE_LAB_KERNELS):
        pytest.skip("the ring kernel is part of LAB builds only (make -C pyhgt_amd/csrc LAB=1; profiles/r05_forced_kernels_suite.txt)")
    N, E, T, R, gk, edit = case
    d, H = 256, 8
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=N % 1000 + R)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E % 1000 + 9, with_time=False, **gk)
    nt, et, ei = nt.clone(), et.clone(), ei.clone()
    if edit == "unknown":
        nt[::211] = T + 3
        et[::97] = R
    if edit == "hubs":
        ei[1, :3000] = 12_345
        ei[1, 3000:5000] = 70_001
    if edit == "one_relation":
        et[:] = 3
        ei[1, :40_000] = ei[1, :40_000] % 97          # runs of ~400 edges into one target (below the hub threshold)
    layer = _layer_from(sd, d, T, R, H, True, False, keep_att=False, precision="bf16x3")
    outs = []
    det = _lib.HGT_FLAG_DETERMINISTIC_HUBS if edit == "hubs" else 0      # (hub rows: atomics by default, not run-to-run reproducible)
    for flags in (_lib.HGT_FLAG_RING_AGGREGATE, 0, _lib.HGT_FLAG_RING_AGGREGATE):
        layer.kernel_flags = flags | det
        out, _ = _run(layer, x, nt, ei, et, None)
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[2])                       # run-to-run
    same = torch.equal(outs[0], outs[1])
    if not same:
        bad = (outs[0] != outs[1]).any(dim=1).nonzero().flatten()
        print("ring vs default kernel: %d differing rows, first %s, max |d| %.3e" % (
            bad.numel(), bad[:8].tolist(), (outs[0] - outs[1]).abs().max().item()))
    assert same
    if N * E <= 70_000 * 700_000:
        ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_norm=True, use_RTE=False, dtype=torch.float64)
        assert (outs[0].double() - ref).abs().max().item() < PREC_TOL["bf16x3"]


@pytest.mark.parametrize("order", ["small_first", "large_first"])
@pytest.mark.parametrize("N", [3000, 70_000])        # unfused (4 targets per wavefront) and fused (streaming walk) aggregation
def test_f16_split_rows_of_very_different_size_per_relation(N, order):
    """fp16 has 5 exponent bits: the aggregation scales the relation rows of a target by ONE power of two, chosen at the target's
    first relation with 2^7 of headroom.  Here the sources of relation 0 are 10^6 times smaller (or larger) than those of the
    other relations, so the scale has to move mid-target (with the accumulator rescale) -- and the tiny rows must neither flush the
    result nor overflow it.  No LayerNorm: the output keeps the raw magnitudes; the error is measured relative to each row."""
    d, H, T, R = 64, 4, 2, 4
    E = 6 * N
    sd = O.make_state_dict(d, d, T, R, H, False, False, seed=31)
    for t in range(T):
        sd["v_linears.%d.bias" % t].zero_()          # V = x W_v: the rows inherit the scale of the source features
        sd["q_linears.%d.weight" % t].zero_()        # Q, K = their biases: logits of size 10^6 would make the fp32 softmax itself
        sd["k_linears.%d.weight" % t].zero_()        # (in the reference too) the dominant error
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=77)
    et = torch.where(ei[0] < N // 2, torch.zeros_like(et), 1 + et % (R - 1))
    lo, hi = (slice(0, N // 2), slice(N // 2, N)) if order == "small_first" else (slice(N // 2, N), slice(0, N // 2))
    x = x.clone()
    x[lo] *= 1e-3
    x[hi] *= 1e3
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_norm=False, use_RTE=False, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, False, False, keep_att=False, precision="f16x3")
    out, _ = _run(layer, x, nt, ei, et, None)
    assert torch.isfinite(out).all()
    rel = ((out.double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-30)).max().item()
    print("f16x3 rows 1e-3 / 1e+3 (%s, N=%d): max row-relative err %.2e" % (order, N, rel))
    assert rel < 1e-5


def test_f16_split_first_row_thirty_five_decades_below_a_later_one():
    """The optimistic fp16 row scale of the aggregation (round 4) takes sigma_t from the target's first non-zero row and walks the
    sub-tile again with 2^14 more headroom per attempt when a later row overflows.  Round-4 advisor finding: the attempts were capped
    at 4, and a target whose first row was more than ~2^70 smaller than a later one got NaN.  Here the sources of relation 0 are
    1e-25 in size and the others 1e+10: the walk must retry ~9 times and still deliver finite, accurate rows."""
    d, H, T, R, N = 64, 4, 2, 4, 70_000
    E = 6 * N
    sd = O.make_state_dict(d, d, T, R, H, False, False, seed=33)
    for t in range(T):
        sd["v_linears.%d.bias" % t].zero_()
        sd["q_linears.%d.weight" % t].zero_()
        sd["k_linears.%d.weight" % t].zero_()
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=78)
    et = torch.where(ei[0] < N // 2, torch.zeros_like(et), 1 + et % (R - 1))
    x = x.clone()
    x[:N // 2] *= 1e-25
    x[N // 2:] *= 1e10
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, None, use_norm=False, use_RTE=False, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, False, False, keep_att=False, precision="f16x3")
    out, _ = _run(layer, x, nt, ei, et, None)
    assert torch.isfinite(out).all()
    rel = ((out.double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-300)).max().item()
    print("f16x3 rows 1e-25 / 1e+10: max row-relative err %.2e" % rel)
    assert rel < 1e-5


def test_classifier_and_matcher_heads():
    """model.py:3-49 on the native kernels: log_softmax(linear(x)), and the scaled (pairwise / all-pairs) dot product of
    projected node pairs incl. the inference cache -- against the same formulas in fp64 torch."""
    from pyhgt_amd import Classifier, Matcher
    g = torch.Generator().manual_seed(5)
    n_hid, n_out, n, m = 96, 37, 300, 45
    x, y = torch.randn(n, n_hid, generator=g), torch.randn(m, n_hid, generator=g)
    clf = Classifier(n_hid, n_out).eval()
    ref = torch.log_softmax(x.double() @ clf.linear.weight.double().T + clf.linear.bias.double(), dim=-1)
    with torch.no_grad():
        out = clf.to(DEV)(x.to(DEV))
    assert out.shape == ref.shape and (out.cpu().double() - ref).abs().max().item() < 1e-5
    mt = Matcher(n_hid).eval()
    tx = x.double() @ mt.left_linear.weight.double().T + mt.left_linear.bias.double()
    ty = y.double() @ mt.right_linear.weight.double().T + mt.right_linear.bias.double()
    mt = mt.to(DEV)
    with torch.no_grad():
        full = mt(x.to(DEV), y.to(DEV))
        pair = mt(x[:m].to(DEV), y.to(DEV), pair=True)
        cached = mt(x.to(DEV), y.to(DEV), infer=True)
        cached2 = mt(torch.zeros_like(x).to(DEV), y.to(DEV), infer=True)        # second call must use the cache, not x
    assert (full.cpu().double() - tx @ ty.T / n_hid ** 0.5).abs().max().item() < 1e-5
    assert (pair.cpu().double() - (tx[:m] * ty).sum(-1) / n_hid ** 0.5).abs().max().item() < 1e-5
    assert torch.equal(cached, full) and torch.equal(cached2, full)


def _halo_worker(rank, world, port, N, E, d, T, R, offsets, n_chunks, tmpdir, blocked=False, hub=False):
    """CPU / gloo: negotiate the HaloPlan of `rank` exactly like a real run, then hand it to the parent.  blocked: first-use
    chunks of the target-blocked schedule (n_chunks = target blocks)."""
    import os
    import torch.distributed as dist
    from pyhgt_amd.dist import HaloPlan, target_blocks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, nt, ei, et, tm = _halo_graph(N, E, d, T, R, offsets, hub)
        lo, hi = offsets[rank], offsets[rank + 1]
        mine = (ei[1] >= lo) & (ei[1] < hi)
        eblock = None
        if blocked:
            dst_l = ei[1][mine] - lo
            bounds = target_blocks(dst_l, hi - lo, n_chunks)
            eblock = torch.searchsorted(torch.tensor(bounds[1:]), dst_l, right=True).clamp(max=n_chunks - 1)
        hp = HaloPlan(nt[lo:hi], ei[0][mine], offsets, rank, world, n_chunks=n_chunks, edge_block=eblock)
        hp.group = None
        torch.save(hp, os.path.join(tmpdir, "halo%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _halo_graph(N, E, d, T, R, offsets, hub):
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=91, sorted_types=False)
    if hub:      # a hub target (> 1024 in-edges) in the second rank's range, unclaimed edges, nodes of no known type
        nt, ei, et = nt.clone(), ei.clone(), et.clone()
        ei[1, :2500] = offsets[1] + 300
        et[::13] = R + 1
        nt[::29] = T
    return x, nt, ei, et, tm


def _c24_round_trip(rows):
    """rows -> 24-bit transport format -> fp32, through the two HIP kernels of the compressed exchange."""
    lib = _lib.load()
    n, d = rows.shape
    idx = torch.arange(n, dtype=torch.int32, device=rows.device)
    wire = torch.empty(n, 3 * d, dtype=torch.uint8, device=rows.device)
    back = torch.empty_like(rows)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.hgt_gather_rows_c24(rows.data_ptr(), d, idx.data_ptr(), n, d, wire.data_ptr(), st) == 0
    assert lib.hgt_unpack_rows_c24(wire.data_ptr(), n, d, back.data_ptr(), d, st) == 0
    return back


def test_c24_transport_format_round_trip():
    """24-bit halo rows: relative error <= 2^-16 per element, exact for values with <= 15 mantissa bits, sign/zero kept."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(777, 200, generator=g) * torch.logspace(-6, 6, 200)
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, -2.5])
    back = _c24_round_trip(x.to(DEV).contiguous()).cpu()
    rel = ((back - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -16 * 1.0001
    assert torch.equal(back[0, :4], x[0, :4])


class _DoneWork:
    def wait(self):
        return True


def _fake_exchange_for(hp, xg):
    """The all-to-all of one chunk replaced by a copy out of the global feature table (one GPU plays every rank)."""
    lib = _lib.load()

    def fake_exchange(c, x_own, x_local, pack=None, async_op=False, compress=False, expand=True):
        a, b = hp.recv_chunk_off[c], hp.recv_chunk_off[c + 1]
        rows = xg[hp.need[hp.halo_order[a:b]]].contiguous()
        wire = None
        if compress:
            n, d = rows.shape
            wire = torch.empty(n, 3 * d, dtype=torch.uint8, device=rows.device)
            if n:
                idx = torch.arange(n, dtype=torch.int32, device=rows.device)
                assert lib.hgt_gather_rows_c24(rows.data_ptr(), d, idx.data_ptr(), n, d, wire.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream) == 0
        if not compress or expand:
            x_local[hp.n_own + a:hp.n_own + b] = _c24_round_trip(rows) if (compress and b > a) else rows
        else:
            x_local[hp.n_own + a:hp.n_own + b] = float("nan")      # nothing may read the fp32 halo rows in the direct mode
        bufs = (x_local[:0], wire) if compress else (x_local[:0],)
        return (_DoneWork(), bufs) if async_op else None
    return fake_exchange


@pytest.mark.parametrize("precision,compress,mode", [("fp32", False, "blocked"), ("bf16x3", False, "pipelined"), ("bf16x3", True, "bucketed"),
                                                     ("bf16x3", False, "bucketed")])
def test_pipelined_partitioned_forward_with_real_halos(precision, compress, mode, tmp_path):
    """The multi-GPU step of pyhgt_amd/dist.py (chunked exchange + hgt_conv_forward stages 1/2/3 or 1/2/4) on ONE GPU: the halo
    plans of a 3-rank partition are negotiated over gloo in CPU worker processes, every rank's forward then
    runs on the device with the all-to-all replaced by a copy out of the global feature table, and the stitched outputs
    must equal the oracle on the whole graph.  An exact-fp32 layer falls back to the pipelined schedule whatever the graph's mode."""
    import socket
    import torch.multiprocessing as mp
    from pyhgt_amd.dist import PartitionedGraph
    N, E, d, T, R, H, world, n_chunks = 900, 9000, 64, 3, 4, 4, 3, 3
    offsets = [0, 250, 610, 900]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_halo_worker, args=(world, port, N, E, d, T, R, offsets, n_chunks, str(tmp_path), mode == "blocked"), nprocs=world, join=True)
    x, nt, ei, et, tm = _halo_graph(N, E, d, T, R, offsets, False)
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=92)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, dtype=torch.float64)
    layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision=precision)
    xg = x.to(DEV)
    for rank in range(world):
        lo, hi = offsets[rank], offsets[rank + 1]
        mine = (ei[1] >= lo) & (ei[1] < hi)
        hp = torch.load(os.path.join(str(tmp_path), "halo%d.pt" % rank), weights_only=False).to(DEV)
        assert hp.n_halo > 0 and hp.n_chunks == n_chunks
        hp.exchange_chunk = _fake_exchange_for(hp, xg)
        pg = PartitionedGraph(None, None, (ei[1][mine] - lo).to(DEV), et[mine].to(DEV), tm[mine].to(DEV), T, R, 0, rank, world,
                              node_offsets=offsets, halo=hp, compress=compress, mode=mode, n_chunks=n_chunks)
        assert pg.mode == mode and (pg.bucket_plan is not None) == (mode == "bucketed")
        assert pg.layer_mode(layer) == ("pipelined" if precision == "fp32" else mode)
        GraphPlan.clear_cache()
        with torch.no_grad():
            out = pg.forward(layer, xg[lo:hi].contiguous())
        assert out.shape == (hi - lo, d)
        assert (out.cpu().double() - ref[lo:hi]).abs().max().item() < TOL


@pytest.mark.parametrize("compress,use_rte", [(False, True), (True, False), (True, True)])
def test_target_blocked_schedule_with_real_halos(compress, use_rte, tmp_path):
    """The round-4 schedule (pyhgt_amd/dist.py "blocked": first-use halo chunks + hgt_conv_forward stage 5 per target block) on
    ONE GPU, like the test above: 3 ranks x 4 target blocks of whole plan tiles, a hub target, unclaimed edges and unknown node
    types; with compress the halo rows are projected straight off the 24-bit wire buffer (the fp32 halo rows are poisoned with
    NaN).  Every rank's stitched output must equal the oracle on the whole graph, and the blocked forward must be BIT-IDENTICAL to
    the pipelined forward of the same rank (the same kernels on tile ranges: no state, no different rounding points)."""
    import socket
    import torch.multiprocessing as mp
    from pyhgt_amd.dist import PartitionedGraph
    N, E, d, T, R, H, world, n_blocks = 9000, 110000, 64, 3, 4, 4, 3, 4
    offsets = [0, 2900, 6100, 9000]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_halo_worker, args=(world, port, N, E, d, T, R, offsets, n_blocks, str(tmp_path), True, True), nprocs=world, join=True)
    x, nt, ei, et, tm = _halo_graph(N, E, d, T, R, offsets, True)
    sd = O.make_state_dict(d, d, T, R, H, True, use_rte, seed=92)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm if use_rte else None, dtype=torch.float64, use_RTE=use_rte)
    layer = _layer_from(sd, d, T, R, H, True, use_rte, keep_att=False, precision="bf16x3")
    det = _lib.HGT_FLAG_DETERMINISTIC_HUBS      # (the default hub path adds fp32 partials atomically: not bit-reproducible)
    layer.kernel_flags = det
    xg = x.to(DEV)
    for rank in range(world):
        lo, hi = offsets[rank], offsets[rank + 1]
        mine = (ei[1] >= lo) & (ei[1] < hi)
        hp = torch.load(os.path.join(str(tmp_path), "halo%d.pt" % rank), weights_only=False).to(DEV)
        hp.exchange_chunk = _fake_exchange_for(hp, xg)
        tmr = tm[mine].to(DEV) if use_rte else None
        pg = PartitionedGraph(None, None, (ei[1][mine] - lo).to(DEV), et[mine].to(DEV), tmr, T, R, 0, rank, world,
                              node_offsets=offsets, halo=hp, compress=compress, mode="blocked", n_chunks=n_blocks)
        assert pg.layer_mode(layer) == "blocked" and len(pg.blocks) == n_blocks
        assert pg.blocks[0][0] == 0 and pg.blocks[-1][1] == hi - lo and all(b[0] % 256 == 0 for b in pg.blocks)
        assert all(pg.blocks[i][1] == pg.blocks[i + 1][0] and pg.blocks[i][3] == pg.blocks[i + 1][2] for i in range(n_blocks - 1))
        GraphPlan.clear_cache()
        with torch.no_grad():
            out = pg.forward(layer, xg[lo:hi].contiguous())
            pg.mode = "pipelined"
            layer.kernel_flags = det | _lib.HGT_FLAG_FUSED_ANY_SIZE      # the same fused aggregation + update kernel, over the whole range
            pg.workspace = None                                          # (sized per forward: flags changed)
            out_p = pg.forward(layer, xg[lo:hi].contiguous())
            layer.kernel_flags = det
            pg.workspace = None
        assert out.shape == (hi - lo, d) and torch.isfinite(out).all()
        assert (out.cpu().double() - ref[lo:hi]).abs().max().item() < TOL
        assert torch.equal(out, out_p)


@pytest.mark.parametrize("use_rte,zipf,H", [(False, False, 8), (True, True, 8), (True, False, 4)])
def test_target_blocks_are_bit_identical_to_the_one_call_layer(use_rte, zipf, H):
    """hgt_conv_forward stage 5 (ABI 6): the edge phase + fused update of a range of destination tiles.  Running the blocks of a
    graph one after the other (in any order) must reproduce the one-call layer BIT FOR BIT -- with source-only halo rows, hub
    targets inside and outside a block (the hub kernels filter by range), unclaimed edges and unknown node types.  The third case:
    heads of 64 columns -- the block's logits run on the matrix cores (hgt_edge_logits_range with the fragment image).  (Staged calls
    are precision 1 by contract: hgt_conv_forward answers HGT_ERR_UNSUPPORTED to precision 2 with a stage, pyhgt_amd.HGTConv maps
    an "f16x3" layer's staged calls to "bf16x3".)"""
    T, R, d, N, NQ, E = 4, 8, 256, 90_000, 70_000, 900_000
    sd = O.make_state_dict(d, d, T, R, H, True, use_rte, seed=51)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=52, sorted_types=False)
    nt, ei, et = nt.clone(), ei.clone(), et.clone()
    ei[1] = ei[1] % NQ                     # targets are the first NQ rows, the rest are source-only halo rows
    if zipf:
        ei[1, :4000] = 70                  # hub in the first block
        ei[1, 4000:6500] = 40_000          # hub in a later block
    et[::11] = R + 2
    nt[::17] = T
    layer = _layer_from(sd, d, T, R, H, True, use_rte, keep_att=False, precision="bf16x3")
    layer.kernel_flags = _lib.HGT_FLAG_DETERMINISTIC_HUBS      # hub rows without atomics: bit-reproducible (the default path is not)
    xd, ntd, eid, etd, tmd = _to_dev(x, nt, ei, et, tm)
    GraphPlan.clear_cache()
    plan = GraphPlan(ntd, eid, etd, tmd if use_rte else None, T, R, n_q_rows=NQ)
    ws = torch.empty(layer.workspace_bytes(N, E), dtype=torch.uint8, device=DEV)
    args = (xd, ntd, eid, etd, tmd if use_rte else None)
    with torch.no_grad():
        ref = layer(*args, plan=plan, n_q_rows=NQ, workspace=ws).clone()
        tab, tile = plan.tile_items()
        assert tile == 256 and tab.numel() == (N + tile - 1) // tile + 1 and (tab[1:] >= tab[:-1]).all()
        bounds = [0, 256 * 40, 256 * 41, 256 * 41, 256 * 200, NQ]              # uneven blocks, an EMPTY one, a ragged last tile
        out = torch.full((NQ, d), float("nan"), device=DEV)
        kw = dict(plan=plan, n_q_rows=NQ, workspace=ws)
        layer(*args, stage=1, **kw)
        halo = torch.arange(NQ, N, device=DEV)
        hp_types = ntd[NQ:]
        order = torch.argsort(torch.where(hp_types < T, hp_types, torch.full_like(hp_types, T)), stable=True)
        cnt = torch.bincount(hp_types.clamp(max=T), minlength=T + 1)[:T]
        off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
        off[1:] = torch.cumsum(cnt, 0).to(torch.int32)
        rows = halo[order][:int(cnt.sum())].to(torch.int32).contiguous()
        layer(*args, stage=2, proj=(rows, off), **kw)
        for b in (3, 0, 4, 2, 1):          # any order: the blocks are independent
            q0, q1 = bounds[b], bounds[b + 1]
            blk = (q0, q1, int(tab[q0 // tile]), int(tab[(q1 + tile - 1) // tile]))
            assert layer(*args, stage=5, block=blk, out=out, **kw) is out
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "blocks differ from the one-call layer: max %.3e in %d rows" % (
        (out - ref).abs().max().item(), int((out != ref).any(dim=1).sum()))
    fwd = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm if use_rte else None, use_norm=True, use_RTE=use_rte)
    assert (out.cpu().double() - fwd[:NQ]).abs().max().item() < TOL
    # argument checks of the new stage
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            layer(*args, stage=5, block=(100, 512, 0, 1), out=out, **kw)          # q_begin is not a multiple of the plan tile
        with pytest.raises(ValueError):
            layer(*args, stage=5, block=(0, 256, 0, 1), out=out[:10], **kw)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_typed_linear_reads_the_24_bit_wire_format(precision):
    """prologue 2 of the split typed linears (ABI 6): x = rows in the 24-bit transport format of the halo exchange, decoded by the
    kernel's loader.  Must be bit-identical to unpacking the rows first (the decoded value has 16 significant bits, so its bf16
    hi / mid split is exact) -- ragged row tiles, several groups, rows addressed through a row list with an offset base."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    n, k, n_out, G = 1000, 256, 512, 3
    x = (torch.randn(n, k, generator=g) * torch.logspace(-3, 3, k)).to(DEV)
    W = (torch.randn(G, n_out, k, generator=g) / 16).to(DEV)
    bias = torch.randn(G, n_out, generator=g).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    idx = torch.arange(n, dtype=torch.int32, device=DEV)
    wire = torch.empty(n, 3 * k, dtype=torch.uint8, device=DEV)
    assert lib.hgt_gather_rows_c24(x.data_ptr(), k, idx.data_ptr(), n, k, wire.data_ptr(), st) == 0
    xu = torch.empty_like(x)
    assert lib.hgt_unpack_rows_c24(wire.data_ptr(), n, k, xu.data_ptr(), k, st) == 0
    base = 5000                                                    # local id of the wire buffer's first row
    perm = torch.randperm(n, generator=g).to(DEV)
    rows = (base + perm[:900]).to(torch.int32).contiguous()        # 900 of the 1000 rows, shuffled, in 3 groups
    off = torch.tensor([0, 301, 301 + 64, 900], dtype=torch.int32, device=DEV)
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(G, k, n_out, C.byref(nb)) == 0
    wsplit = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    split = lib.hgt_split_weights_f16 if precision == "f16x3" else lib.hgt_split_weights
    lin = lib.hgt_typed_linear_f16x3 if precision == "f16x3" else lib.hgt_typed_linear_bf16x3
    assert split(W.data_ptr(), n_out * k, G, k, n_out, wsplit.data_ptr(), st) == 0
    outs = []
    rows0 = (rows - base).contiguous()                              # the same rows addressed from the start of the fp32 buffer
    for mode in (0, 2):
        if mode == 0:      # plain fp32 rows (unpacked first): ids relative to the buffer
            o0, o1 = torch.zeros(n, 256, device=DEV), torch.zeros(n, 256, device=DEV)
            assert lin(xu.data_ptr(), k, rows0.data_ptr(), off.data_ptr(), G, 900, k, n_out, wsplit.data_ptr(), bias.data_ptr(), n_out,
                       o0.data_ptr(), o1.data_ptr(), None, 256, 0, 0, st) == 0
            outs.append((o0, o1))
        else:              # 24-bit wire rows, addressed by LOCAL row id through a base pointer shifted back by `base` rows (stage 2)
            o0, o1 = torch.zeros(base + n, 256, device=DEV), torch.zeros(base + n, 256, device=DEV)
            assert lin(wire.data_ptr() - base * (3 * k // 4) * 4, 3 * k // 4, rows.data_ptr(), off.data_ptr(), G, 900, k, n_out,
                       wsplit.data_ptr(), bias.data_ptr(), n_out, o0.data_ptr(), o1.data_ptr(), None, 256, 0, 2, st) == 0
            assert float(o0[:base].abs().max()) == 0.0 and float(o1[:base].abs().max()) == 0.0
            outs.append((o0[base:], o1[base:]))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    refK = torch.zeros(n, n_out, dtype=torch.float64)
    offl = off.tolist()
    for gi in range(G):
        r = rows0[offl[gi]:offl[gi + 1]].long().cpu()
        refK[r] = xu.cpu().double()[r] @ W[gi].cpu().double().T + bias[gi].cpu().double()
    got = torch.cat([outs[1][0], outs[1][1]], 1).cpu().double()
    scale = (xu.cpu().double().abs() @ W.abs().amax(0).cpu().double().T).max().item()
    assert (got - refK).abs().max().item() <= (1e-6 if precision == "f16x3" else 3e-5) * scale
    # prologue values beyond 2 are rejected; a K the persistent kernel does not cover is "unsupported" (the caller unpacks first)
    assert lin(wire.data_ptr(), 3 * k // 4, rows0.data_ptr(), off.data_ptr(), G, 900, k, n_out, wsplit.data_ptr(), bias.data_ptr(), n_out,
               outs[0][0].data_ptr(), outs[0][1].data_ptr(), None, 256, 0, 3, st) == -1
    assert lin(wire.data_ptr(), 3 * 512 // 4, rows0.data_ptr(), off.data_ptr(), G, 900, 512, n_out, wsplit.data_ptr(), bias.data_ptr(), n_out,
               outs[0][0].data_ptr(), outs[0][1].data_ptr(), None, 256, 0, 2, st) == -2


@pytest.mark.parametrize("use_rte,n_slices,N,E", [(False, 4, 70000, 700000), (True, 3, 70000, 500000), (True, 5, 3000, 40000)])
def test_bucketed_edge_phase_matches_one_call_layer(use_rte, n_slices, N, E):
    """hgt_conv_forward stage 4: the edges are bucketed by source-row range, relation ids become
    bucket * R + relation, and the edge phase runs bucket by bucket with the softmax state carried in the workspace.  The result
    must agree with the one-call layer to the split-bf16 bound (<= 5e-5: a (target, relation) partial sum is rounded to bf16 hi/mid
    once per BUCKET here and once per relation there, so the 2^-17-relative rounding points differ; the softmax merge itself is
    exact up to fp32 rounding) and with the fp64 oracle like every other path (TOL) -- with hub targets (> 1024 in-edges, processed with the last
    bucket), unclaimed edges, unknown node types, targets whose edges all sit in one bucket and targets with none."""
    T, R, H, d = 3, 4, 8, 256
    sd = O.make_state_dict(d, d, T, R, H, True, use_rte, seed=41)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=42, sorted_types=False)
    nt, ei, et = nt.clone(), ei.clone(), et.clone()
    ei[1, :3000] = 5                       # a hub target
    ei[1, 3000:3040] = 9                   # a target all of whose edges come from one source range
    ei[0, 3000:3040] = torch.arange(40) % 7
    et[::11] = R + 2                       # edges no relation claims
    nt[::17] = T                           # nodes of no known type
    x[:, :] *= 3.0                         # wider logit range: the partial references of the buckets differ
    layer = _layer_from(sd, d, T, R, H, True, use_rte, keep_att=True, precision="bf16x3")
    xd, ntd, eid, etd, tmd = _to_dev(x, nt, ei, et, tm)
    GraphPlan.clear_cache()
    with torch.no_grad():
        ref = layer(xd, ntd, eid, etd, tmd).clone()
        att_ref = layer.att.clone()
    bucket = (ei[0] * n_slices // N).clamp(max=n_slices - 1)
    claimed = (et >= 0) & (et < R)
    et_b = torch.where(claimed, bucket * R + et, torch.full_like(et, n_slices * R)).to(DEV)
    plan = GraphPlan(ntd, eid, et_b, tmd, T, n_slices * R)
    ws = torch.empty(layer.workspace_bytes(N, E, n_slices), dtype=torch.uint8, device=DEV)
    with torch.no_grad():
        kw = dict(plan=plan, workspace=ws)
        assert layer(xd, ntd, eid, et_b, tmd, stage=1, slices=(0, n_slices), **kw) is None
        for b in range(n_slices):
            out = layer(xd, ntd, eid, et_b, tmd, stage=4, slices=(b, n_slices), **kw)
            assert (out is None) == (b < n_slices - 1)
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 5e-5
    datt = (layer.att - att_ref).abs()
    assert datt.max().item() <= 1e-5      # fp32 summation order of the softmax denominators (3000-term hub sums)
    fwd = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm if use_rte else None, use_norm=True, use_RTE=use_rte)
    assert (out.cpu().double() - fwd).abs().max().item() < TOL


def test_forward_is_hip_graph_capturable():
    """hgt_conv_forward never synchronises and only uses torch's caching allocator, so a whole forward (here two stacked
    layers) can be captured in a hipGraph and replayed; the replay must reproduce the eager result bit for bit."""
    T, R, H, d, N, E = 3, 4, 4, 64, 1200, 9000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=17)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=18)
    layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision="bf16x3")
    xd, ntd, eid, etd, tmd = _to_dev(x, nt, ei, et, tm)
    GraphPlan.clear_cache()
    plan = GraphPlan(ntd, eid, etd, tmd, T, R)
    with torch.no_grad():
        eager = layer(layer(xd, ntd, eid, etd, tmd, plan=plan), ntd, eid, etd, tmd, plan=plan).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            layer(xd, ntd, eid, etd, tmd, plan=plan)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = layer(layer(xd, ntd, eid, etd, tmd, plan=plan), ntd, eid, etd, tmd, plan=plan)
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_bench_line_of_a_multi_rank_run_reports_world_size_and_backend(tmp_path):
    """Guard for the driver's scaling runs: `bench.py --gpus 2` launched through torch.distributed.run prints ONE JSON line whose
    n_gpus is the world size and whose config names the collective backend.  Walked here with both ranks on the one GPU of the
    test box (HGT_BENCH_DEVICE=0, gloo): the replicas workload (configs[4], no data-path collective) and its parity field."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HGT_BENCH_DEVICE="0", HGT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c5", "--steps", "6", "--warmup", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["scaling"] == "weak" and j["unit"] == "edges/s"
    assert j["config"]["parallelism"] == "replicas x2" and j["config"]["backend"].startswith("gloo")
    assert j["value"] > 0 and j["parity_max_abs_err"] is not None and j["parity_max_abs_err"] <= 1e-4


def test_bench_line_of_a_two_rank_partitioned_run(tmp_path):
    """`bench.py --gpus 2` on the configs[3] recipe at a reduced size, both ranks on the one GPU of the test box (gloo, host-staged
    exchange): global graph -> in-edge-balanced partitioner -> first-use halo chunks negotiated between the ranks -> target-blocked
    steps with 24-bit halo rows projected off the wire buffer.  The line must carry the parity of the benchmarked tensors, a
    whole-step roofline without per-kernel fractions (round-3 review: fractions > 1 from phase events that bracket no kernel),
    the stage times, and the two secondary schedules with their own parity."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HGT_BENCH_DEVICE="0", HGT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--nodes-per-gpu", "40000", "--edges-per-gpu", "400000", "--blocks", "4", "--locality", "0.5", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    cfg = j["config"]
    assert cfg["parallelism"].startswith("dst-partition x2") and cfg["edge_phase"].startswith("target-blocked: 4 blocks")
    assert cfg["halo_format"].startswith("24-bit") and cfg["locality"] == 0.5 and len(cfg["node_offsets"]) == 3
    assert abs(cfg["node_offsets"][1] - 40000) <= 512 and cfg["node_offsets"][1] % 256 == 0
    assert j["parity_max_abs_err"] is not None and j["parity_max_abs_err"] <= 1e-4
    rf = j["roofline"]
    assert "per_kernel_frac" not in rf and 0.0 < rf["frac"] <= 1.0 and rf["frac"] == rf["layer_frac"]
    assert {"pack", "own_qkv", "wait", "halo_kv", "edge_blocks"} <= set(rf["stage_ms"])
    sec = j["secondary"]
    assert sec["halo_fp32"]["parity_max_abs_err"] <= 1e-4 and sec["edge_phase_after_last_chunk"]["parity_max_abs_err"] <= 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: parity-evidence gaps named by the round-3 review
# ---------------------------------------------------------------------------------------------------------------------
class _ReferenceShapedGNN(torch.nn.Module):
    """The reference's wrapper, statement for statement (pyHGT/model.py:51-80), as TEST scaffolding: a torch adapter
    (`adapt_ws[t]` + tanh through boolean masks and `res[idx] = ...`), dropout, and the POSITIONAL 5-argument call
    `gc(meta_xs, node_type, edge_index, edge_type, edge_time)` of model.py:79 into `GeneralConv` -- which is what runs when the
    reference's own model.py is used with `pyhgt_amd.install_into(pyHGT.conv)` (the reference itself cannot travel to the GPU box)."""

    def __init__(self, in_dim, n_hid, num_types, num_relations, n_heads, n_layers, dropout=0.2, conv_name='hgt', prev_norm=False,
                 last_norm=False, use_RTE=True):
        super().__init__()
        self.gcs = torch.nn.ModuleList()
        self.num_types, self.in_dim, self.n_hid = num_types, in_dim, n_hid
        self.adapt_ws = torch.nn.ModuleList()
        self.drop = torch.nn.Dropout(dropout)
        for t in range(num_types):
            self.adapt_ws.append(torch.nn.Linear(in_dim, n_hid))
        for l in range(n_layers - 1):
            self.gcs.append(GeneralConv(conv_name, n_hid, n_hid, num_types, num_relations, n_heads, dropout, use_norm=prev_norm, use_RTE=use_RTE))
        self.gcs.append(GeneralConv(conv_name, n_hid, n_hid, num_types, num_relations, n_heads, dropout, use_norm=last_norm, use_RTE=use_RTE))

    def forward(self, node_feature, node_type, edge_time, edge_index, edge_type):
        res = torch.zeros(node_feature.size(0), self.n_hid).to(node_feature.device)
        for t_id in range(self.num_types):
            idx = (node_type == int(t_id))
            if idx.sum() == 0:
                continue
            res[idx] = torch.tanh(self.adapt_ws[t_id](node_feature[idx]))
        meta_xs = self.drop(res)
        del res
        for gc in self.gcs:
            meta_xs = gc(meta_xs, node_type, edge_index, edge_type, edge_time)
        return meta_xs


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("name", ["gnn_oag2", "gnn_mag4"])
def test_reference_call_sequence_through_general_conv_matches_reference_goldens(name, precision):
    """The drop-in as the reference's unmodified GNN.forward drives it (model.py:69-80): torch adapter, masks, positional call into
    GeneralConv -> HGTConv on the HIP path, the same graph tensors handed to every layer (one cached plan), reference parameter
    names loaded with load_state_dict -- against the outputs of the verbatim reference GNN (tests/golden/gnn_*.npz)."""
    from oracle.gen_golden_gnn import GNN_CASES, build_batch
    c = GNN_CASES[name]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    _, (x, nt, tm, ei, et, _, _) = build_batch(c)
    sd = O.make_gnn_state_dict(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], c["prev_norm"], c["last_norm"],
                               c["use_RTE"], seed=c["seed"])
    gnn = _ReferenceShapedGNN(c["in_dim"], c["n_hid"], c["T"], c["R"], c["H"], c["n_layers"], 0.2, "hgt", c["prev_norm"], c["last_norm"],
                              c["use_RTE"]).eval()
    missing, unexpected = gnn.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    gnn = gnn.to(DEV)
    for gc in gnn.gcs:
        assert type(gc.base_conv) is HGTConv
        gc.base_conv.precision = precision
    GraphPlan.clear_cache()
    built = []
    orig = GraphPlan.__init__
    GraphPlan.__init__ = lambda self, *a, **k: (built.append(1), orig(self, *a, **k))[1]
    try:
        with torch.no_grad():
            out = gnn(*_to_dev(x, nt, tm, ei, et))
    finally:
        GraphPlan.__init__ = orig
    torch.cuda.synchronize()
    assert len(built) == 1                                   # every layer found the first layer's plan
    rows = torch.from_numpy(z["rows"]).long().to(DEV)
    err = (out[rows].cpu() - torch.from_numpy(z["layers"][-1])).abs().max().item()
    print("%s %s reference call sequence: max|err| %.1e" % (name, precision, err))
    assert err < 1e-4
    GraphPlan.clear_cache()


def test_strict_mode_raises_like_the_reference_on_the_first_forward():
    """conv.py:57 (index_select inside propagate) raises IndexError at once for a node id outside [0, N).  strict=True (or
    GraphPlan.STRICT) reproduces that on the very FIRST forward of a new graph; the default surfaces it on a later forward."""
    T, R, H, d, N, E = 3, 4, 4, 64, 500, 3000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=5)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=6)
    ei = ei.clone()
    ei[0, 17] = N + 3
    layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision="bf16x3")
    args = _to_dev(x, nt, ei, et, tm)
    GraphPlan.clear_cache()
    layer.strict = True
    with torch.no_grad(), pytest.raises(IndexError):
        layer(*args)
    GraphPlan.clear_cache()
    layer.strict = None
    GraphPlan.STRICT = True
    try:
        with torch.no_grad(), pytest.raises(IndexError):
            layer(*args)
        tm_bad = tm.clone()
        tm_bad[5] = 240
        good_ei = _to_dev(synthetic_typed_graph(N, E, d, T, R, seed=6)[2])[0]
        with torch.no_grad(), pytest.raises(IndexError):
            layer(args[0], args[1], good_ei, args[3], tm_bad.to(DEV))
        with torch.no_grad():                                 # a well-formed graph is unaffected
            out = layer(args[0], args[1], good_ei, args[3], args[4])
        assert torch.isfinite(out).all()
    finally:
        GraphPlan.STRICT = False
        GraphPlan.clear_cache()


def test_plan_cache_and_weight_cache_follow_in_place_edits():
    """GraphPlan.cached keys on (address, version counter, shape, strides) of the graph tensors; the packed-weight cache on the
    parameters' version counters AND storage addresses.  What bumps a version (every torch in-place op) or moves a storage
    (`p.data = tensor`, a re-assigned Parameter) is picked up; a write THROUGH `.data` is invisible to both -- documented, and
    undone by GraphPlan.clear_cache() / layer.invalidate()."""
    T, R, H, d, N, E = 3, 4, 4, 64, 800, 6000
    sd = O.make_state_dict(d, d, T, R, H, True, False, seed=7)
    x, nt, ei, et, _ = synthetic_typed_graph(N, E, d, T, R, seed=8)
    layer = _layer_from(sd, d, T, R, H, True, False, keep_att=False, precision="bf16x3")
    xd, ntd, eid, etd = _to_dev(x, nt, ei.contiguous(), et)
    oracle = lambda ei_, sd_: O.forward_closed_form(sd_, T, R, H, x, nt, ei_, et, None, use_RTE=False)
    GraphPlan.clear_cache()
    with torch.no_grad():
        assert (layer(xd, ntd, eid, etd).cpu().double() - oracle(ei, sd)).abs().max().item() < TOL
        # 1. an in-place torch op on edge_index bumps its version: a new plan is built
        ei2 = ei.clone()
        ei2[0] = torch.roll(ei[0], 1)
        eid.copy_(ei2.to(DEV))
        assert (layer(xd, ntd, eid, etd).cpu().double() - oracle(ei2, sd)).abs().max().item() < TOL
        # 2. a write through .data does not: the stale plan (of ei2) is served until the cache is cleared
        eid.data.copy_(ei.to(DEV))
        stale = layer(xd, ntd, eid, etd).cpu().double()
        assert (stale - oracle(ei2, sd)).abs().max().item() < TOL
        GraphPlan.clear_cache()
        assert (layer(xd, ntd, eid, etd).cpu().double() - oracle(ei, sd)).abs().max().item() < TOL
        # 3. weights: `p.data = tensor` and a re-assigned Parameter are detected without invalidate()
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2["k_linears.1.weight"] = sd["k_linears.1.weight"] * 0.5
        layer.k_linears[1].weight.data = sd2["k_linears.1.weight"].to(DEV)
        assert (layer(xd, ntd, eid, etd).cpu().double() - oracle(ei, sd2)).abs().max().item() < TOL
        sd2["skip"] = sd["skip"] + 1.0
        layer.skip = torch.nn.Parameter(sd2["skip"].to(DEV))
        assert (layer(xd, ntd, eid, etd).cpu().double() - oracle(ei, sd2)).abs().max().item() < TOL
    # 4. LRU order: a hit refreshes BOTH keys of a registered plan (with / without edge_time)
    GraphPlan.clear_cache()
    old = GraphPlan.CACHE_SIZE
    GraphPlan.CACHE_SIZE = 2
    try:
        tms = [torch.zeros(E, dtype=torch.int64, device=DEV) for _ in range(3)]
        eis = [eid.clone() for _ in range(3)]
        plans = [GraphPlan(ntd, eis[i], etd, tms[i], T, R) for i in range(3)]
        GraphPlan.register(plans[0], ntd, eis[0], etd, tms[0], T, R)
        GraphPlan.register(plans[1], ntd, eis[1], etd, tms[1], T, R)
        assert GraphPlan.cached(ntd, eis[0], etd, tms[0], T, R) is plans[0]        # plan 0 is now the most recently used
        GraphPlan.register(plans[2], ntd, eis[2], etd, tms[2], T, R)              # evicts plan 1, not the sibling key of plan 0
        assert GraphPlan.cached(ntd, eis[0], etd, None, T, R) is plans[0]
        assert GraphPlan.cached(ntd, eis[2], etd, tms[2], T, R) is plans[2]
    finally:
        GraphPlan.CACHE_SIZE = old
        GraphPlan.clear_cache()


def test_prepared_images_survive_a_change_of_kernel_flags():
    """Round-3 advisor finding: the fragment images of the prepared buffer were only written when the FIRST forward's flags selected
    the kernels that read them; a later flag change on the live layer then read uninitialised memory.  Every order must work."""
    T, R, H, d, N, E = 3, 4, 8, 256, 3000, 30000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=9)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=10)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm)
    args = _to_dev(x, nt, ei, et, tm)
    F = _lib
    orders = [(F.HGT_FLAG_VALU_LOGITS | F.HGT_FLAG_VALU_AGGREGATE, 0, F.HGT_FLAG_MFMA_LOGITS), (F.HGT_FLAG_VALU_AGGREGATE, F.HGT_FLAG_MFMA_LOGITS),
              (F.HGT_FLAG_MFMA_LOGITS, F.HGT_FLAG_VALU_LOGITS, 0), (0, F.HGT_FLAG_MFMA_LOGITS | F.HGT_FLAG_NO_ITEM_AGGREGATE)]
    for precision in ("bf16x3", "f16x3"):
        for order in orders:
            layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision=precision)
            GraphPlan.clear_cache()
            for fl in order:
                layer.kernel_flags = fl
                with torch.no_grad():
                    out = layer(*args)
                assert (out.cpu().double() - ref).abs().max().item() < TOL, (precision, order, fl)


@pytest.mark.parametrize("precision,flags", [("bf16x3", 0), ("f16x3", 0), ("bf16x3", _lib.HGT_FLAG_NO_FUSED_UPDATE), ("fp32", 0)])
def test_deterministic_hub_mode_is_bit_reproducible(precision, flags):
    """HGT_FLAG_DETERMINISTIC_HUBS (ABI 6): hub targets (> 1024 in-edges) are aggregated without atomics -- one partial slot per piece,
    summed in a fixed order -- so repeated forwards are bit-identical on EVERY row, hubs included (the reference's scatter-add,
    conv.py:13, gives no such guarantee on a GPU; the default hub path here does not either).  Fused and unfused aggregation, the
    matrix-core and the exact vector-ALU kernels; result against the oracle like every other path."""
    T, R, H, d, N, E = 3, 4, 8, 256, 70_000, 700_000
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=61)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=62, sorted_types=False)
    ei, et = ei.clone(), et.clone()
    ei[1, :6000] = 11                      # hubs of very different size, in different workgroups
    ei[1, 6000:7500] = 40_000
    ei[1, 7500:9000] = 40_001
    et[::9] = R + 1                        # unclaimed edges (also inside the hubs)
    layer = _layer_from(sd, d, T, R, H, True, True, keep_att=False, precision=precision)
    layer.kernel_flags = flags | _lib.HGT_FLAG_DETERMINISTIC_HUBS | _lib.HGT_FLAG_NO_ITEM_AGGREGATE
    args = _to_dev(x, nt, ei, et, tm)
    GraphPlan.clear_cache()
    with torch.no_grad():
        outs = [layer(*args).clone() for _ in range(4)]
        layer.kernel_flags = flags | _lib.HGT_FLAG_NO_ITEM_AGGREGATE
        atomics = layer(*args).clone()
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm)
    assert (outs[0].cpu().double() - ref).abs().max().item() < TOL
    assert (atomics - outs[0]).abs().max().item() < 1e-5       # same sums, another order
    hubs = torch.tensor([11, 40_000, 40_001], device=DEV)
    others = torch.ones(N, dtype=torch.bool, device=DEV)
    others[hubs] = False
    assert torch.equal(atomics[others], outs[0][others])       # every non-hub row is deterministic in both modes


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("N,k,n_out,T,prologue", [
    (300_007, 256, 768, 4, 0),       # the Q|K|V shape: several rounds per workgroup, ragged groups incl. an empty and a 37-row one
    (1_000_000, 256, 768, 4, 0),     # BASELINE configs[1] row count
    (300_007, 256, 512, 4, 2),       # halo K|V straight off 24-bit wire rows
    (300_007, 512, 1536, 4, 0),      # K = 512 (n_hid 512): four wavefronts of 512 registers
    (1_000_000, 512, 512, 3, 0),
])
def test_xs_gemm_against_fp64_at_its_dispatch_sizes(N, k, n_out, T, prologue, precision):
    """csrc/hgt_gemm_xs.hip is only taken for >= 262 144 rows (K = 512: >= 65 536), so test_typed_linear_against_torch_fp32 (N = 1000)
    never reaches it and the bit-identity test compares it with another HIP kernel.  Here its output is compared DIRECTLY with a
    float64 matmul (torch, on the GPU) at the sizes where it is dispatched: ragged, permuted row lists with an empty group, fp32 rows
    and 24-bit wire rows (conv.py:96-97,103 typed projections).  Tolerances: those of test_typed_linear_against_torch_fp32."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_xs
    lib = _lib.load()
    f16 = 1 if precision == "f16x3" else 0
    x, W, b, rows, off, ws = bench_xs.setup(lib, N, k, n_out, T, f16, seed=N % 89 + k, ragged=1)
    st = torch.cuda.current_stream().cuda_stream
    if prologue == 2:
        idx = torch.arange(N, dtype=torch.int32, device=DEV)
        wire = torch.empty(N, 3 * k, dtype=torch.uint8, device=DEV)
        assert lib.hgt_gather_rows_c24(x.data_ptr(), k, idx.data_ptr(), N, k, wire.data_ptr(), st) == 0
        xin = torch.empty_like(x)
        assert lib.hgt_unpack_rows_c24(wire.data_ptr(), N, k, xin.data_ptr(), k, st) == 0      # what the kernel's loader decodes
        xptr, ldx = wire.data_ptr(), 3 * k // 4
    else:
        xin, xptr, ldx = x, x.data_ptr(), k
    nblk = 3 if n_out % 3 == 0 else 2
    bc = n_out // nblk
    outs = [torch.full((N, bc), float("nan"), device=DEV) for _ in range(nblk)]
    bench_xs.run(lib, f16, xptr, ldx, rows, off, T, N, k, n_out, ws, b, outs, bc, 0, prologue)
    torch.cuda.synchronize()
    got = torch.cat(outs, 1)
    assert not torch.isnan(got).any()          # every row of the list was written
    offl = off.tolist()
    tol = (4e-6 if f16 else 1e-4) * (k // 256)      # (K = 512: twice the terms per sum)
    worst = 0.0
    for t in range(T):
        r = rows[offl[t]:offl[t + 1]].long()
        for lo in range(0, r.numel(), 131072):                 # (fp64 reference in slices: 1M x 1536 doubles would be 12 GB)
            rr = r[lo:lo + 131072]
            ref = xin[rr].double() @ W[t].double().T + b[t].double()
            worst = max(worst, float((got[rr].double() - ref).abs().max()))
    print("xs typed_linear N=%d k=%d n=%d prologue=%d %s err %.2e" % (N, k, n_out, prologue, precision, worst))
    assert worst < tol


@pytest.mark.parametrize("N,k,n_out,T,f16,c24,bypos", [
    (1000, 256, 768, 3, 0, 0, 0),          # a handful of units: one round per workgroup
    (300007, 256, 768, 4, 0, 0, 1),        # several rounds per workgroup: the in-loop prefetch of the next item's rows
    (300007, 256, 768, 4, 1, 0, 0),        # fp16 split: row scales found from the fragment-shaped rows
    (200003, 256, 512, 3, 0, 1, 0),        # 24-bit wire rows (the halo K|V projection)
    (150001, 64, 192, 5, 1, 0, 0),         # K = 64
    (90001, 256, 200, 3, 0, 0, 0),         # a last step with one masked and one partly masked column tile
    (200003, 512, 1536, 3, 0, 0, 0),       # K = 512: four wavefronts of 512 registers, 32 columns per step
    (120001, 512, 512, 4, 1, 0, 1),
])
def test_xs_gemm_is_bit_identical_to_the_slab_kernel(N, k, n_out, T, f16, c24, bypos):
    """csrc/hgt_gemm_xs.hip (x rows stationary in registers, W through an LDS ring by LDS-DMA) accumulates every output element in
    the order of k_typed_linear_pc: on ragged, permuted inputs with an empty and a tiny group the two kernels (selected through the
    prologue bits HGT_LINEAR_NO_XS / HGT_LINEAR_FORCE_XS of the C ABI) must agree to the bit, and untouched output rows must stay
    untouched."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_xs
    lib = _lib.load()
    assert bench_xs.check(lib, N, k, n_out, T, f16, c24, bypos, ragged=1, seed=N % 97)




# ------------------------------------------------------------------ rows / logits beyond 4 GiB from their bases
def _fused_vs_unfused(layer, x, nt, ei, et, rows=None):
    """The fused aggregation + update kernel (32-bit lane offsets when every row sits below 4 GiB, 64-bit addresses otherwise)
    against the unfused kernels (always 64-bit addresses) on the same inputs."""
    outs = []
    for fl in (_lib.HGT_FLAG_DETERMINISTIC_HUBS | _lib.HGT_FLAG_NO_ITEM_AGGREGATE | _lib.HGT_FLAG_FUSED_ANY_SIZE,
               _lib.HGT_FLAG_DETERMINISTIC_HUBS | _lib.HGT_FLAG_NO_ITEM_AGGREGATE | _lib.HGT_FLAG_NO_FUSED_UPDATE):
        layer.kernel_flags = fl
        GraphPlan.clear_cache()
        with torch.no_grad():
            o = layer(x, nt, ei, et, None)
        torch.cuda.synchronize()
        outs.append(o if rows is None else o[rows].clone())
        del o
    return outs


def test_gathered_rows_beyond_four_gib_from_the_table_base():
    """4.3 M rows of 256 floats: the V rows of the last nodes sit above 2^32 bytes, where the fused kernel must drop its 32-bit lane
    offsets (hgt_edge_agg_mfma.hip: HgtFusedUpdate::small32).  All edges live among the last 4096 nodes."""
    if torch.cuda.get_device_properties(0).total_memory < 64 << 30:
        pytest.skip("needs ~30 GB of device memory")
    N, n_act, E, d, H, T, R = 4_300_000, 4096, 60_000, 256, 8, 3, 5
    assert (N - n_act) * d * 4 > 1 << 32
    g = torch.Generator().manual_seed(11)
    layer = HGTConv(d, d, T, R, H, 0.2, True, False, precision="bf16x3").eval().to(DEV)
    x = torch.zeros(N, d, device=DEV)
    x[N - n_act:] = torch.randn(n_act, d, generator=g).to(DEV)
    nt = torch.randint(0, T, (N,), generator=g).to(DEV)
    ei = (torch.randint(0, n_act, (2, E), generator=g) + (N - n_act)).to(DEV)
    et = torch.randint(0, R, (E,), generator=g).to(DEV)
    rows = torch.arange(N - n_act, N, device=DEV)
    fused, unfused = _fused_vs_unfused(layer, x, nt, ei, et, rows)
    # the same graph renumbered to 4096 nodes: everything below 4 GiB
    with torch.no_grad():
        small = layer(x[N - n_act:].contiguous(), nt[N - n_act:].contiguous(), ei - (N - n_act), et, None)
    assert torch.isfinite(fused).all()
    assert (fused - unfused).abs().max().item() < 2e-5
    assert (fused - small).abs().max().item() < 2e-5


def test_logits_beyond_four_gib_from_their_base():
    """135 M edges x 8 heads: the logit rows of the last edges sit above 2^32 bytes (the other half of the small32 condition)."""
    if torch.cuda.get_device_properties(0).total_memory < 64 << 30:
        pytest.skip("needs ~30 GB of device memory")
    N, E, d, H, T, R = 200_000, 135_000_000, 32, 8, 2, 3
    assert E * H * 4 > 1 << 32
    g = torch.Generator(device=DEV).manual_seed(12)
    layer = HGTConv(d, d, T, R, H, 0.2, True, False, precision="bf16x3").eval().to(DEV)
    x = torch.randn(N, d, device=DEV, generator=g)
    nt = torch.randint(0, T, (N,), device=DEV, generator=g)
    ei = torch.randint(0, N, (2, E), device=DEV, generator=g)
    et = torch.randint(0, R, (E,), device=DEV, generator=g)
    fused, unfused = _fused_vs_unfused(layer, x, nt, ei, et)
    assert torch.isfinite(fused).all()
    assert (fused - unfused).abs().max().item() < 2e-5


# ------------------------------------------------------------------ a_linear + gated skip + LayerNorm in one kernel, up to 512 columns
@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("n,k,n_out,use_norm", [
    (1000, 256, 256, 1),      # one pass, persistent kernel
    (1000, 512, 512, 1),      # n_hid 512: two passes x two K panels in registers (k_typed_linear_update_wide)
    (777, 512, 400, 1),       # n_hid 400 (OAG): 8 heads x 50 padded to 64 -> K = 512, 400 columns, the second pass partly masked
    (300, 320, 512, 0),       # K not a multiple of the panel, no LayerNorm
    (70_000, 512, 512, 1),    # more row tiles than CUs
])
def test_linear_update_against_float64(precision, n, k, n_out, use_norm):
    """hgt_linear_update_{bf16x3,f16x3} (conv.py:125-133: a_linear, gated skip, LayerNorm) against the float64 closed form, at the
    column counts of its three kernels; ragged groups incl. an empty one, rows through a shuffled row list, rows of no group untouched."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(5 + n_out)
    G = 4
    agg = torch.randn(n, k, generator=g).to(DEV)
    xs = torch.randn(n, n_out, generator=g).to(DEV)
    W = (torch.randn(G, n_out, k, generator=g) / k ** 0.5).to(DEV)
    bias = torch.randn(G, n_out, generator=g).to(DEV)
    skip = torch.tensor([0.3, -1.2, 2.0, 0.0]).to(DEV)
    lnw = (1.0 + 0.1 * torch.randn(G, n_out, generator=g)).to(DEV)
    lnb = (0.1 * torch.randn(G, n_out, generator=g)).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    n_used = n - 37
    perm = torch.randperm(n, generator=g)
    rows = perm[:n_used].to(torch.int32).to(DEV).contiguous()
    c1 = n_used // 3
    off = torch.tensor([0, c1, c1, c1 + 65, n_used], dtype=torch.int32, device=DEV)      # group 1 is empty
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(G, k, n_out, C.byref(nb)) == 0
    wsplit = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    split = lib.hgt_split_weights_f16 if precision == "f16x3" else lib.hgt_split_weights
    upd = lib.hgt_linear_update_f16x3 if precision == "f16x3" else lib.hgt_linear_update_bf16x3
    assert split(W.data_ptr(), n_out * k, G, k, n_out, wsplit.data_ptr(), st) == 0
    out = torch.full((n, n_out), 7.0, device=DEV)
    assert upd(agg.data_ptr(), k, rows.data_ptr(), off.data_ptr(), G, n_used, k, n_out, wsplit.data_ptr(), bias.data_ptr(), n_out,
               xs.data_ptr(), n_out, skip.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), use_norm, out.data_ptr(), st) == 0
    torch.cuda.synchronize()
    ref = torch.full((n, n_out), 7.0, dtype=torch.float64)
    offl = off.tolist()
    for gi in range(G):
        r = rows[offl[gi]:offl[gi + 1]].long().cpu()
        if r.numel() == 0:
            continue
        a = torch.sigmoid(skip[gi].cpu().double())
        y = (agg.cpu().double()[r] @ W[gi].cpu().double().T + bias[gi].cpu().double()) * a + xs.cpu().double()[r] * (1 - a)
        if use_norm:
            y = torch.nn.functional.layer_norm(y, (n_out,), lnw[gi].cpu().double(), lnb[gi].cpu().double(), 1e-5)
        ref[r] = y
    err = (out.cpu().double() - ref).abs().max().item()
    print("linear_update %s n_out %d k %d: max|err| %.2e" % (precision, n_out, k, err))
    assert err < (4e-6 if precision == "f16x3" else 5e-5)
    # beyond 512 columns (or K beyond two panels with more than one pass): unsupported, the caller takes the two-kernel form
    assert upd(agg.data_ptr(), k, rows.data_ptr(), off.data_ptr(), G, n_used, k, 516, wsplit.data_ptr(), bias.data_ptr(), n_out,
               xs.data_ptr(), n_out, skip.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), use_norm, out.data_ptr(), st) == -2


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("n,k,n_out,prologue", [(3200, 256, 768, 0), (1000, 512, 1536, 0), (777, 200, 1200, 0), (300, 1169, 400, 0),
                                                  (5000, 64, 192, 0), (4000, 256, 512, 0), (70, 129, 96, 0), (2000, 250, 96, 0),
                                                  (3200, 512, 1536, 0), (4096, 400, 1536, 0), (4096, 1169, 400, 0)])      # 96-, 128- and 32-row streamed tiles
def test_tile_linear_is_bit_identical_to_the_slab_kernels(precision, n, k, n_out, prologue):
    """The latency-regime typed linear (csrc/hgt_gemm_tile.hip, round 6: K <= 256 and at most ~1 000 workgroups of 32 x 128 outputs)
    keeps the slab kernels' split, k order and product order: bit-identical output on ragged shuffled groups (incl. an empty one), K
    that is not a multiple of a panel / of 4 (scalar row loads), both output modes; shapes outside its domain fall through to the
    slab kernels (the comparison is then trivially equal, the float64 bound still checked)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + k)
    G = 4
    x = (torch.randn(n + 9, k, generator=g) * torch.pow(10.0, torch.rand(n + 9, 1, generator=g) * 4 - 2)).to(DEV)
    W = (torch.randn(G, n_out, k, generator=g) / k ** 0.5).to(DEV)
    bias = torch.randn(G, n_out, generator=g).to(DEV)
    rows = torch.randperm(n + 9, generator=g)[:n].to(torch.int32).to(DEV).contiguous()
    c1 = n // 3
    off = torch.tensor([0, c1, c1, min(n, c1 + 65), n], dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(G, k, n_out, C.byref(nb)) == 0
    ws = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    split, linear = ((lib.hgt_split_weights_f16, lib.hgt_typed_linear_f16x3) if precision == "f16x3" else
                     (lib.hgt_split_weights, lib.hgt_typed_linear_bf16x3))
    assert split(W.data_ptr(), n_out * k, G, k, n_out, ws.data_ptr(), st) == 0
    nblk = 3 if n_out % 3 == 0 and (n_out // 3) % 4 == 0 else 1
    bc = n_out // nblk
    for by_pos in (0, 1):
        res = []
        for sel in (0, _lib.HGT_LINEAR_NO_TILE | _lib.HGT_LINEAR_NO_XS):
            outs = [torch.full((n + 9, bc), 3.0, device=DEV) for _ in range(nblk)]
            optr = [o.data_ptr() for o in outs] + [0, 0]
            assert linear(x.data_ptr(), k, rows.data_ptr(), off.data_ptr(), G, n, k, n_out, ws.data_ptr(), bias.data_ptr(), n_out,
                          optr[0], optr[1], optr[2], bc, by_pos, prologue | sel, st) == 0
            torch.cuda.synchronize()
            res.append(torch.cat(outs, dim=1))
        assert torch.equal(res[0], res[1]), "tile kernel differs from the slab kernel: max %.3e" % (res[0] - res[1]).abs().max().item()
    # and against float64 (the slab kernel's own bound)
    xin = torch.nn.functional.gelu(x.cpu().double()) if prologue else x.cpu().double()
    offl = off.tolist()
    worst = 0.0
    for gi in range(G):
        r = rows[offl[gi]:offl[gi + 1]].long().cpu()
        if r.numel() == 0:
            continue
        ref = xin[r] @ W[gi].cpu().double().T + bias[gi].cpu().double()
        got = res[0][offl[gi]:offl[gi + 1]].cpu().double()          # (the by_pos = 1 run: position order)
        scale = xin[r].abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
        worst = max(worst, ((got - ref).abs() / scale).max().item())
    print("tile linear %s n=%d k=%d n_out=%d: max|err| / row scale %.2e" % (precision, n, k, n_out, worst))
    assert worst < (1e-5 if precision == "f16x3" else 6e-5)      # (rows spanning four decades: the slab kernel's own figure; 8.7e-6 over 6.3 M outputs at k = 400)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("n,k,n_out", [(3000, 129, 512), (3200, 512, 400), (4096, 1169, 400), (300, 64, 64), (9000, 512, 1536)])
def test_typed_linear_with_the_tanh_epilogue(precision, n, k, n_out):
    """prologue | HGT_LINEAR_TANH (round 6: the GNN's typed adapter + tanh, model.py:70-76, in one kernel): tanh(x W^T + b) against
    float64 wherever a latency-regime kernel takes the shape (tile kernel, streamed kernel, K > 256 slab kernel); where none does the
    call answers HGT_ERR_UNSUPPORTED and has launched nothing (the output keeps its fill value)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + k + n_out)
    G = 3
    x = torch.randn(n, k, generator=g).to(DEV)
    W = (torch.randn(G, n_out, k, generator=g) / k ** 0.5).to(DEV)
    bias = torch.randn(G, n_out, generator=g).to(DEV)
    nt = torch.randint(0, G, (n,), generator=g).sort().values
    rows = torch.arange(n, dtype=torch.int32).to(DEV)
    off = torch.searchsorted(nt, torch.arange(G + 1)).int().to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(G, k, n_out, C.byref(nb)) == 0
    ws = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    split, linear = ((lib.hgt_split_weights_f16, lib.hgt_typed_linear_f16x3) if precision == "f16x3" else
                     (lib.hgt_split_weights, lib.hgt_typed_linear_bf16x3))
    assert split(W.data_ptr(), n_out * k, G, k, n_out, ws.data_ptr(), st) == 0
    out = torch.full((n, n_out), 9.0, device=DEV)
    rc = linear(x.data_ptr(), k, rows.data_ptr(), off.data_ptr(), G, n, k, n_out, ws.data_ptr(), bias.data_ptr(), n_out, out.data_ptr(), 0, 0,
                n_out, 0, _lib.HGT_LINEAR_TANH, st)
    torch.cuda.synchronize()
    if rc == -2:
        assert k <= 256 and bool((out == 9.0).all())      # K <= 256 outside the tile kernel's domain: nothing launched
        return
    assert rc == 0
    ref = torch.empty(n, n_out, dtype=torch.float64)
    ntl = nt.tolist()
    for gi in range(G):
        m = nt == gi
        ref[m] = torch.tanh(x.cpu().double()[m] @ W[gi].cpu().double().T + bias[gi].cpu().double())
    err = (out.cpu().double() - ref).abs().max().item()
    print("tanh epilogue %s n=%d k=%d n_out=%d: max|err| %.2e" % (precision, n, k, n_out, err))
    assert err < (6e-6 if precision == "f16x3" else 5e-5)      # (K = 1169: 3.6e-6 in the fp16 split)


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("n,k,n_out,use_norm", [(3200, 256, 256, 1), (1000, 512, 512, 1), (777, 512, 400, 0), (333, 256, 64, 1), (4096, 512, 400, 1)])
def test_tile_linear_update_is_bit_identical_to_the_slab_kernels(precision, n, k, n_out, use_norm):
    """The fused update of the latency regime (32 whole rows per workgroup, hgt_gemm_tile.hip) against the persistent / wide kernels
    (bit 1 of `use_norm` keeps the call off the tile kernel): same sums, same LayerNorm order -> bit-identical rows."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(11 + n_out + n)
    G = 4
    agg = torch.randn(n, k, generator=g).to(DEV)
    xs = torch.randn(n, n_out, generator=g).to(DEV)
    W = (torch.randn(G, n_out, k, generator=g) / k ** 0.5).to(DEV)
    bias = torch.randn(G, n_out, generator=g).to(DEV)
    skip = torch.tensor([0.3, -1.2, 2.0, 0.0]).to(DEV)
    lnw = (1.0 + 0.1 * torch.randn(G, n_out, generator=g)).to(DEV)
    lnb = (0.1 * torch.randn(G, n_out, generator=g)).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    n_used = n - 37
    rows = torch.randperm(n, generator=g)[:n_used].to(torch.int32).to(DEV).contiguous()
    c1 = n_used // 3
    off = torch.tensor([0, c1, c1, c1 + 65, n_used], dtype=torch.int32, device=DEV)
    nb = C.c_uint64()
    assert lib.hgt_split_weights_bytes(G, k, n_out, C.byref(nb)) == 0
    wsplit = torch.empty(int(nb.value), dtype=torch.uint8, device=DEV)
    split = lib.hgt_split_weights_f16 if precision == "f16x3" else lib.hgt_split_weights
    upd = lib.hgt_linear_update_f16x3 if precision == "f16x3" else lib.hgt_linear_update_bf16x3
    assert split(W.data_ptr(), n_out * k, G, k, n_out, wsplit.data_ptr(), st) == 0
    res = []
    for no_tile in (0, 2):
        out = torch.full((n, n_out), 7.0, device=DEV)
        assert upd(agg.data_ptr(), k, rows.data_ptr(), off.data_ptr(), G, n_used, k, n_out, wsplit.data_ptr(), bias.data_ptr(), n_out,
                   xs.data_ptr(), n_out, skip.data_ptr(), lnw.data_ptr(), lnb.data_ptr(), use_norm | no_tile, out.data_ptr(), st) == 0
        torch.cuda.synchronize()
        res.append(out)
    d = (res[0] - res[1]).abs().max().item()
    print("tile update %s n=%d k=%d n_out=%d: max|tile - slab| %.3e" % (precision, n, k, n_out, d))
    assert d <= 2e-6      # (same products; the LayerNorm sums meet in a different order: last-bit differences)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 1000, 100_003])
def test_c24_pack_of_256_column_rows_is_bit_exact(n):
    """hgt_gather_rows_c24 at the benchmark width against the format's definition computed with torch integer ops (little-endian 3-byte
    groups of (bits + 0x80) >> 8) -- random row indices with repeats, ragged counts, the bytes behind the last row untouched."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(n)
    d, n_src = 256, 5000
    x = (torch.randn(n_src, d, generator=g) * torch.logspace(-5, 5, d)).to(DEV)
    idx = torch.randint(0, n_src, (n,), generator=g).to(torch.int32).to(DEV)
    wire = torch.full((n + 1, 3 * d), 0xAB, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.hgt_gather_rows_c24(x.data_ptr(), d, idx.data_ptr(), n, d, wire.data_ptr(), st) == 0
    torch.cuda.synchronize()
    bits = x[idx.long()].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    v = ((bits + 0x80) >> 8) & 0xFFFFFF                                   # 24-bit codes, [n, 256]
    ref = torch.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], dim=2).reshape(n, 3 * d).to(torch.uint8)   # little-endian 3-byte groups
    assert torch.equal(wire[:n], ref)
    assert bool((wire[n] == 0xAB).all())
    back = torch.empty(n, d, device=DEV)
    assert lib.hgt_unpack_rows_c24(wire.data_ptr(), n, d, back.data_ptr(), d, st) == 0
    torch.cuda.synchronize()
    assert ((back - x[idx.long()]).abs() <= x[idx.long()].abs() * 2.0 ** -16 * 1.0001).all()
