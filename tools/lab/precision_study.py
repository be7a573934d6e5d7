#!/usr/bin/env python
"""Which stage of the layer carries the split-bf16 error?  (CPU, fp64; tools/lab: not part of the library.)

Every MFMA stage of the default precision evaluates a product as a_mid*b_hi + a_hi*b_mid + a_hi*b_hi with bf16 hi / mid terms.
This script replays one layer in fp64 and applies that operand rounding to ONE stage at a time (typed Q|K|V projections, relation
message transforms U_r M_r, a_linear), everything else exact, and prints the resulting max |out - exact|.
Also: the same with an fp16 hi / lo split (11 + 11 mantissa bits, row-scaled into range), and with a third bf16 term (x6)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hgt_oracle as O  # noqa: E402
from pyhgt_amd.sampled import synthetic_sampled_batch, to_torch_layout  # noqa: E402


def split_bf16(a, terms=2):
    parts, r = [], a.float().double()
    for _ in range(terms):
        p = r.float().to(torch.bfloat16).double()
        parts.append(p)
        r = r - p
    return parts


def split_f16(a):
    # row-scaled so that the row maximum sits at 2^14 (exact power-of-two scaling), hi + lo in fp16
    m = a.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    sc = torch.exp2(14 - torch.ceil(torch.log2(m)))
    s = a * sc
    hi = s.float().to(torch.float16).double()
    lo = (s - hi).float().to(torch.float16).double()
    return [hi / sc, lo / sc]


def mm(a, bt, mode):
    """a [n,k] @ bt[k,m] under an operand-splitting mode."""
    if mode == "exact":
        return a @ bt
    if mode == "bf16x3":
        (ah, am), (bh, bm) = split_bf16(a), split_bf16(bt.T)
        return am @ bh.T + ah @ bm.T + ah @ bh.T
    if mode == "bf16x6":
        (a0, a1, a2), (b0, b1, b2) = split_bf16(a, 3), split_bf16(bt.T, 3)
        return a2 @ b0.T + a1 @ b1.T + a0 @ b2.T + a1 @ b0.T + a0 @ b1.T + a0 @ b0.T
    if mode == "f16x3":
        (ah, al), (bh, bl) = split_f16(a), split_f16(bt.T)
        return al @ bh.T + ah @ bl.T + ah @ bh.T
    raise ValueError(mode)


def layer(sd, T, R, H, x, nt, ei, et, tm, modes):
    dt = torch.float64
    x = x.double()
    N, d = x.shape
    dk = d // H
    src, dst = ei[0], ei[1]
    Q = torch.zeros(N, d, dtype=dt); K = torch.zeros(N, d, dtype=dt); V = torch.zeros(N, d, dtype=dt)
    for t in range(T):
        m = nt == t
        for name, out in (("q", Q), ("k", K), ("v", V)):
            out[m] = mm(x[m], sd["%s_linears.%d.weight" % (name, t)].double().T, modes["qkv"]) + sd["%s_linears.%d.bias" % (name, t)].double()
    rte = sd["emb.emb.weight"].double() @ sd["emb.lin.weight"].double().T + sd["emb.lin.bias"].double()
    tj = nt[src]
    k_e = K[src].clone(); v_e = V[src].clone()
    for t in range(T):
        m = tj == t
        k_e[m] += mm(rte, sd["k_linears.%d.weight" % t].double().T, modes["qkv"])[tm[m]]
        v_e[m] += mm(rte, sd["v_linears.%d.weight" % t].double().T, modes["qkv"])[tm[m]]
    s = torch.zeros(et.numel(), H, dtype=dt)
    for r in range(R):
        sel = (et == r).nonzero(as_tuple=True)[0]
        kp = torch.einsum("ehk,hkc->ehc", k_e[sel].view(-1, H, dk), sd["relation_att"][r].double())
        s[sel] = (Q[dst[sel]].view(-1, H, dk) * kp).sum(-1) * sd["relation_pri"][r].double() / math.sqrt(dk)
    att = O._segment_softmax(s, dst, N)
    agg = torch.zeros(N, d, dtype=dt)
    for r in range(R):
        sel = (et == r).nonzero(as_tuple=True)[0]
        U = torch.zeros(N, H, dk, dtype=dt).index_add_(0, dst[sel], v_e[sel].view(-1, H, dk) * att[sel].unsqueeze(-1))
        for h in range(H):
            agg.view(N, H, dk)[:, h] += mm(U[:, h], sd["relation_msg"][r, h].double(), modes["msg"])
    g = 0.5 * agg * (1 + torch.erf(agg / math.sqrt(2)))
    out = torch.zeros(N, d, dtype=dt)
    for t in range(T):
        m = nt == t
        tr = mm(g[m], sd["a_linears.%d.weight" % t].double().T, modes["a"]) + sd["a_linears.%d.bias" % t].double()
        al = torch.sigmoid(sd["skip"][t].double())
        y = tr * al + x[m] * (1 - al)
        mu = y.mean(-1, keepdim=True); var = ((y - mu) ** 2).mean(-1, keepdim=True)
        out[m] = (y - mu) / torch.sqrt(var + 1e-5) * sd["norms.%d.weight" % t].double() + sd["norms.%d.bias" % t].double()
    return out, s


def main():
    batch = synthetic_sampled_batch("mag", n_seed=128, width=128, depth=6, feat_dim=256, mean_degree=4.0, seed=3)
    x, nt, tm, ei, et, _, ed = to_torch_layout(*batch)
    T, R, d, H = 4, len(ed), 256, 8
    sd = O.make_state_dict(d, d, T, R, H, True, True, seed=77)
    base = dict(qkv="exact", msg="exact", a="exact")
    ref, sref = layer(sd, T, R, H, x, nt, ei, et, tm, base)
    chk = O.forward_closed_form(sd, T, R, H, x, nt, ei, et, tm, use_norm=True, use_RTE=True, dtype=torch.float64)
    print("restatement vs oracle: %.1e   max |logit| %.1f" % ((ref - chk).abs().max(), sref.abs().max()))
    for mode in ("bf16x3", "bf16x6", "f16x3"):
        for stage in ("qkv", "msg", "a", "all"):
            m = dict(base)
            for k in (("qkv", "msg", "a") if stage == "all" else (stage,)):
                m[k] = mode
            out, s = layer(sd, T, R, H, x, nt, ei, et, tm, m)
            print("%-7s in %-4s: max|out err| %.2e   max|logit err| %.2e" % (mode, stage, (out - ref).abs().max(), (s - sref).abs().max()))


if __name__ == "__main__":
    main()
