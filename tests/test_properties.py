"""Size-independent properties of HGTConv.forward (conv.py:56-134), driven by hypothesis:

  * the order of the edge list does not matter (PyG's scatter / softmax are order-free, conv.py:57,108);
  * relabelling the nodes permutes the output rows;
  * listing every edge twice changes nothing (a softmax-weighted mean over in-edges);
  * the exported attention sums to 1 over the in-edges of every target that has any (conv.py:108);
  * a target without in-edges gets the update of a zero aggregate (conv.py:119-133 with agg = 0).

The CPU half pins the oracle itself (closed form against the line-by-line meta-relation port: these are the two restatements
every GPU test leans on); the GPU half (-m gpu) runs the same properties through the C ABI, where they exercise what the
oracle cannot: the plan's sort, the work-item cuts, the run / slot bookkeeping of the item-parallel kernels (small graphs take
hgt_edge_aggregate_items, graphs of >= 16384 targets the fused streaming kernel)."""
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import hgt_oracle as O
from pyhgt_amd.synth import synthetic_typed_graph

CASE = st.tuples(st.integers(5, 60),            # nodes
                 st.integers(0, 300),           # edges
                 st.sampled_from([(16, 2), (32, 4), (24, 3), (64, 1)]),     # (d, heads)
                 st.integers(1, 3),             # node types
                 st.integers(1, 4),             # relations
                 st.booleans(),                 # use_RTE
                 st.integers(0, 10_000))        # seed


def _graph(N, E, d, T, R, seed):
    x, nt, ei, et, tm = synthetic_typed_graph(N, max(E, 1), d, T, R, seed=seed, sorted_types=False, strided_edge_index=False)
    if E == 0:
        ei, et, tm = ei[:, :0], et[:0], tm[:0]
    return x, nt, ei.contiguous(), et, tm


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(CASE)
def test_oracle_restatements_agree_and_are_edge_order_free(case):
    N, E, (d, H), T, R, use_RTE, seed = case
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=seed)
    x, nt, ei, et, tm = _graph(N, E, d, T, R, seed)
    a, att = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=use_RTE, dtype=torch.float64, return_att=True)
    b = O.forward_meta_relation_port(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=use_RTE)      # (fp32 like the reference)
    assert (a - b.double()).abs().max().item() < 5e-5
    if E > 0:
        perm = torch.randperm(E, generator=torch.Generator().manual_seed(seed))
        c = O.forward_closed_form(sd, T, R, H, x, nt, ei[:, perm], et[perm], tm[perm], use_norm=True, use_RTE=use_RTE, dtype=torch.float64)
        assert (a - c).abs().max().item() < 1e-9
        sums = torch.zeros(N, H, dtype=torch.float64).index_add_(0, ei[1], att)
        has = torch.zeros(N, dtype=torch.bool)
        has[ei[1]] = True
        assert (sums[has] - 1.0).abs().max().item() < 1e-9
        # every edge twice
        e2 = O.forward_closed_form(sd, T, R, H, x, nt, torch.cat([ei, ei], 1), torch.cat([et, et]), torch.cat([tm, tm]),
                                   use_norm=True, use_RTE=use_RTE, dtype=torch.float64)
        assert (a - e2).abs().max().item() < 1e-9


# ------------------------------------------------------------------------------------------------ the HIP path
def _gpu_layer(sd, d, T, R, H, use_RTE, precision):
    from pyhgt_amd import HGTConv
    layer = HGTConv(d, d, T, R, H, 0.2, True, use_RTE, keep_att=True, precision=precision).eval()
    layer.load_state_dict(sd)
    return layer.to("cuda:0")


def _gpu_run(layer, x, nt, ei, et, tm, use_RTE):
    from pyhgt_amd import GraphPlan
    GraphPlan.clear_cache()
    dev = "cuda:0"
    with torch.no_grad():
        out = layer(x.to(dev), nt.to(dev), ei.to(dev), et.to(dev), tm.to(dev) if use_RTE else None)
    torch.cuda.synchronize()
    return out.cpu(), layer.att.cpu()


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(CASE, st.sampled_from(["bf16x3", "f16x3", "fp32"]))
def test_hip_path_properties_small_graphs(case, precision):
    N, E, (d, H), T, R, use_RTE, seed = case
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=seed)
    x, nt, ei, et, tm = _graph(N, E, d, T, R, seed)
    layer = _gpu_layer(sd, d, T, R, H, use_RTE, precision)
    tol = 1e-4 if precision == "bf16x3" else 1e-5
    out, att = _gpu_run(layer, x, nt, ei, et, tm, use_RTE)
    ref = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=use_RTE, dtype=torch.float64)
    assert (out.double() - ref).abs().max().item() < tol
    if E == 0:
        return
    g = torch.Generator().manual_seed(seed + 1)
    # edge order
    perm = torch.randperm(E, generator=g)
    out_p, att_p = _gpu_run(layer, x, nt, ei[:, perm], et[perm], tm[perm], use_RTE)
    assert (out_p - out).abs().max().item() < tol
    assert (att_p - att[perm]).abs().max().item() < 1e-5           # self.att follows the caller's edge order
    # node relabelling: new id of node i is pi[i]
    pi = torch.randperm(N, generator=g)
    inv = torch.empty_like(pi)
    inv[pi] = torch.arange(N)
    out_r, _ = _gpu_run(layer, x[inv], nt[inv], pi[ei], et, tm, use_RTE)
    assert (out_r[pi] - out).abs().max().item() < tol
    # every edge twice
    out_2, att_2 = _gpu_run(layer, x, nt, torch.cat([ei, ei], 1), torch.cat([et, et]), torch.cat([tm, tm]), use_RTE)
    assert (out_2 - out).abs().max().item() < tol
    assert (att_2[:E] + att_2[E:] - att).abs().max().item() < 1e-5
    # attention sums to one over the in-edges of every target that has any
    sums = torch.zeros(N, H).index_add_(0, ei[1], att)
    has = torch.zeros(N, dtype=torch.bool)
    has[ei[1]] = True
    assert (sums[has] - 1.0).abs().max().item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("N", [20_000, 70_000])          # fused streaming kernel (>= 16384 targets); 20 000 is not a multiple of 256
def test_hip_path_properties_fused_kernel(N, precision):
    """The same properties where the aggregation + update run as one kernel: edge order, node relabelling, duplicated edges."""
    d, H, T, R, E = 64, 4, 3, 4, 6 * N
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=N)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=N + 1, sorted_types=False, strided_edge_index=False)
    layer = _gpu_layer(sd, d, T, R, H, True, precision)
    tol = 1e-4 if precision == "bf16x3" else 1e-5
    out, att = _gpu_run(layer, x, nt, ei, et, tm, True)
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(E, generator=g)
    out_p, att_p = _gpu_run(layer, x, nt, ei[:, perm], et[perm], tm[perm], True)
    assert (out_p - out).abs().max().item() < tol
    assert (att_p - att[perm]).abs().max().item() < 1e-5
    pi = torch.randperm(N, generator=g)
    inv = torch.empty_like(pi)
    inv[pi] = torch.arange(N)
    out_r, _ = _gpu_run(layer, x[inv], nt[inv], pi[ei], et, tm, True)
    assert (out_r[pi] - out).abs().max().item() < tol
    out_2, _ = _gpu_run(layer, x, nt, torch.cat([ei, ei], 1), torch.cat([et, et]), torch.cat([tm, tm]), True)
    assert (out_2 - out).abs().max().item() < tol
    sums = torch.zeros(N, H).index_add_(0, ei[1], att)
    has = torch.zeros(N, dtype=torch.bool)
    has[ei[1]] = True
    assert (sums[has] - 1.0).abs().max().item() < 1e-5
