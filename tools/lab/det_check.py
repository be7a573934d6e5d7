"""tools/lab/det_check.py: two forwards of one layer under several kernel-flag sets: where do they differ (rows, in-degree of the rows)."""
import sys, torch
sys.path.insert(0, ".")
from oracle import hgt_oracle as O
from pyhgt_amd import HGTConv, GraphPlan, _lib
from pyhgt_amd.synth import synthetic_typed_graph
sys.path.insert(0, "tests")
import test_hgt_gpu as TG

def main():
    N, E, d, H, T, R, use_RTE = 20000, int(sys.argv[1]) if len(sys.argv) > 1 else 600000, 64, 1, 3, 4, True
    skew = float(sys.argv[2]) if len(sys.argv) > 2 else 1.1
    sd = O.make_state_dict(d, d, T, R, H, True, use_RTE, seed=N + E + 9)
    x, nt, ei, et, tm = synthetic_typed_graph(N, E, d, T, R, seed=E + 17, dst_skew=skew)
    deg = torch.bincount(ei[1], minlength=N)
    print("E", E, "max in-degree", int(deg.max()), "targets > 1024:", int((deg > 1024).sum()))
    F = _lib
    sets = {"items": F.HGT_FLAG_ITEM_AGGREGATE, "items+nocoop": F.HGT_FLAG_ITEM_AGGREGATE | F.HGT_FLAG_NO_COOP_EDGE,
            "items+nocoop+valu_logits": F.HGT_FLAG_ITEM_AGGREGATE | F.HGT_FLAG_NO_COOP_EDGE | F.HGT_FLAG_VALU_LOGITS,
            "items+nocoop+two_calls": F.HGT_FLAG_ITEM_AGGREGATE | F.HGT_FLAG_NO_COOP_EDGE | F.HGT_FLAG_NO_MERGE_UPDATE,
            "no_items": F.HGT_FLAG_NO_ITEM_AGGREGATE, "no_items+det_hubs": F.HGT_FLAG_NO_ITEM_AGGREGATE | F.HGT_FLAG_DETERMINISTIC_HUBS}
    for prec in ("bf16x3",):
        for name, fl in sets.items():
            layer = TG._layer_from(sd, d, T, R, H, True, use_RTE, keep_att=False, precision=prec)
            layer.kernel_flags = fl
            a, _ = TG._run(layer, x, nt, ei, et, tm)
            b, _ = TG._run(layer, x, nt, ei, et, tm)
            bad = ((a != b).any(dim=1)).nonzero().flatten()
            print("%-28s %s rows that differ: %d  max|diff| %.2e  in-degrees of the first: %s" % (
                name, prec, bad.numel(), (a - b).abs().max().item(), deg[bad[:8]].tolist()))
main()
