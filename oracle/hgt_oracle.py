"""CPU oracle for one HGTConv layer -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this file; pyhgt_amd/ never does (the product path fails loudly when
the HIP library is missing instead of falling back to anything in here).

It restates, in plain torch-on-CPU, the algorithm of the reference layer
    /root/reference/pyHGT/conv.py:11-139   (HGTConv)
    /root/reference/pyHGT/conv.py:283-299  (RelTemporalEncoding)
plus the three PyTorch-Geometric 1.3.2 symbols the reference delegates to
(MessagePassing.propagate, utils.softmax, nn.inits.glorot -- not vendored in
/root/reference; restated from their published semantics, see
oracle/pyg_shim/).

Pinning: the reference ships NO tests and NO golden vectors for this path
(SURVEY.md section 8c), so the oracle is pinned against outputs of the
reference itself: oracle/gen_golden.py imports the reference's conv.py
verbatim (through oracle/pyg_shim) in the build container, runs it on seeded
inputs and commits inputs+outputs under tests/golden/*.npz;
tests/test_oracle.py checks both entry points below against those fixtures.

Two entry points over the same math:
  * forward_closed_form -- node-level restatement (projections once per node,
    relation transforms per edge); runs in fp64 or fp32, fast enough for
    10^5..10^6 edges.  This is the parity checker.
  * forward_meta_relation_port -- same cost structure as the reference
    (per-meta-relation edge groups, projections recomputed per EDGE, E x d
    intermediates); this is what bench.py times as `cpu_baseline` ("port"),
    because the real reference cannot travel to the GPU box.

Parameter names/shapes are the reference module's state_dict
(conv.py:28-51): {k,q,v,a}_linears.{t}.{weight,bias}, norms.{t}.{weight,bias},
relation_pri [R,H], relation_att/relation_msg [R,H,dk,dk], skip [T],
emb.emb.weight [240,in], emb.lin.{weight,bias}.
"""
import math

import torch

RTE_MAX_LEN = 240  # conv.py:287


# --------------------------------------------------------------------------
# parameter construction (independent re-statement of conv.py:28-54,289-297)
# --------------------------------------------------------------------------
def sinusoid_table(n_hid, max_len=RTE_MAX_LEN, dtype=torch.float32):
    """emb[p,2c]=sin(p*w_c)/sqrt(n), emb[p,2c+1]=cos(p*w_c)/sqrt(n) (conv.py:289-294)."""
    pos = torch.arange(0.0, max_len).unsqueeze(1)
    freq = torch.exp(torch.arange(0, n_hid, 2) * -(math.log(10000.0) / n_hid))
    tab = torch.empty(max_len, n_hid)
    tab[:, 0::2] = torch.sin(pos * freq) / math.sqrt(n_hid)
    tab[:, 1::2] = torch.cos(pos * freq) / math.sqrt(n_hid)
    return tab.to(dtype)


def make_state_dict(in_dim, out_dim, num_types, num_relations, n_heads,
                    use_norm=True, use_RTE=True, seed=0, randomize_gates=True, dense=False):
    """Random parameters in the reference's state_dict layout.

    Distributions mirror the reference init (Linear: U(+-1/sqrt(fan_in));
    relation_att/msg: glorot U(+-sqrt(6/(2*dk)))), but relation_pri and skip are
    randomised (the reference defaults, all ones, would leave those code
    paths untested -- SURVEY.md appendix D.7).
    """
    g = torch.Generator().manual_seed(seed)
    dk = out_dim // n_heads

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    sd = {}
    for t in range(num_types):
        for name, fin in (("k", in_dim), ("q", in_dim), ("v", in_dim), ("a", out_dim)):
            b = 1.0 / math.sqrt(fin)
            sd["%s_linears.%d.weight" % (name, t)] = uni((out_dim, fin), b)
            sd["%s_linears.%d.bias" % (name, t)] = uni((out_dim,), b)
        if use_norm:
            sd["norms.%d.weight" % t] = 1.0 + 0.1 * uni((out_dim,), 1.0)
            sd["norms.%d.bias" % t] = 0.1 * uni((out_dim,), 1.0)
    ga = math.sqrt(6.0 / (2 * dk))
    sd["relation_att"] = uni((num_relations, n_heads, dk, dk), ga)
    sd["relation_msg"] = uni((num_relations, n_heads, dk, dk), ga)
    if randomize_gates:
        sd["relation_pri"] = 0.5 + torch.rand((num_relations, n_heads), generator=g)
        sd["skip"] = torch.randn((num_types,), generator=g)
    else:
        sd["relation_pri"] = torch.ones(num_relations, n_heads)
        sd["skip"] = torch.ones(num_types)
    if use_RTE:
        sd["emb.emb.weight"] = sinusoid_table(in_dim)
        b = 1.0 / math.sqrt(in_dim)
        sd["emb.lin.weight"] = uni((in_dim, in_dim), b)
        sd["emb.lin.bias"] = uni((in_dim,), b)
    if dense:   # DenseHGTConv (conv.py:143-191): no `skip`; shared mid/out linears + out_norm
        del sd["skip"]
        b = 1.0 / math.sqrt(out_dim)
        sd["mid_linear.weight"] = uni((2 * out_dim, out_dim), b)
        sd["mid_linear.bias"] = uni((2 * out_dim,), b)
        b = 1.0 / math.sqrt(2 * out_dim)
        sd["out_linear.weight"] = uni((out_dim, 2 * out_dim), b)
        sd["out_linear.bias"] = uni((out_dim,), b)
        sd["out_norm.weight"] = 1.0 + 0.1 * uni((out_dim,), 1.0)
        sd["out_norm.bias"] = 0.1 * uni((out_dim,), 1.0)
    return sd


def _stack(sd, fmt, n, dtype):
    return torch.stack([sd[fmt % t] for t in range(n)]).to(dtype)


def _gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _layer_norm(y, w, b, eps=1e-5):
    mu = y.mean(dim=-1, keepdim=True)
    var = ((y - mu) ** 2).mean(dim=-1, keepdim=True)
    return (y - mu) / torch.sqrt(var + eps) * w + b


def _segment_softmax(s, dst, n_nodes):
    """Per (target, head) softmax over incoming edges; PyG 1.3.2 utils.softmax
    (conv.py:108): exp(s-max)/(sum exp(s-max) + 1e-16)."""
    m = torch.full((n_nodes, s.shape[1]), float("-inf"), dtype=s.dtype)
    m.scatter_reduce_(0, dst.view(-1, 1).expand_as(s), s, reduce="amax", include_self=True)
    p = torch.exp(s - m[dst])
    z = torch.zeros((n_nodes, s.shape[1]), dtype=s.dtype).index_add_(0, dst, p)
    return p / (z[dst] + 1e-16)


def _update(sd, agg, x, node_type, num_types, use_norm, dtype, library_ops=False):
    """conv.py:114-134 (eval mode: dropout is the identity).  library_ops=True uses the same
    torch.nn.functional calls as the reference (F.gelu, LayerNorm) -- for the timed port; the
    checker keeps the explicit formulas."""
    N, d = agg.shape
    Wa = _stack(sd, "a_linears.%d.weight", num_types, dtype)
    ba = _stack(sd, "a_linears.%d.bias", num_types, dtype)
    g = torch.nn.functional.gelu(agg) if library_ops else _gelu_erf(agg)   # conv.py:119
    out = torch.zeros(N, d, dtype=dtype)                            # conv.py:120
    alpha = torch.sigmoid(sd["skip"].to(dtype))                     # conv.py:129
    for t in range(num_types):
        rows = (node_type == t).nonzero(as_tuple=True)[0]
        if rows.numel() == 0:
            continue
        o = g[rows] @ Wa[t].T + ba[t]                               # conv.py:125
        y = o * alpha[t] + x[rows] * (1.0 - alpha[t])               # conv.py:131/133
        if use_norm:
            w, b = sd["norms.%d.weight" % t].to(dtype), sd["norms.%d.bias" % t].to(dtype)
            y = torch.nn.functional.layer_norm(y, (d,), w, b, 1e-5) if library_ops else _layer_norm(y, w, b)
        out[rows] = y
    return out


def _update_dense(sd, agg, x, node_type, num_types, use_norm, dtype):
    """DenseHGTConv.update, conv.py:250-274 (eval mode): per type  y1 = LN_t(a_linear_t(agg) + x)  (no gelu on
    the aggregate, no gate), then the shared dense layer  out = out_norm(out_linear(gelu(mid_linear(y1))) + y1)."""
    N, d = agg.shape
    Wa = _stack(sd, "a_linears.%d.weight", num_types, dtype)
    ba = _stack(sd, "a_linears.%d.bias", num_types, dtype)
    Wm, bm = sd["mid_linear.weight"].to(dtype), sd["mid_linear.bias"].to(dtype)
    Wo, bo = sd["out_linear.weight"].to(dtype), sd["out_linear.bias"].to(dtype)
    wn, bn = sd["out_norm.weight"].to(dtype), sd["out_norm.bias"].to(dtype)
    out = torch.zeros(N, d, dtype=dtype)                            # conv.py:255
    for t in range(num_types):
        rows = (node_type == t).nonzero(as_tuple=True)[0]
        if rows.numel() == 0:
            continue
        y1 = agg[rows] @ Wa[t].T + ba[t] + x[rows]                  # conv.py:259
        if use_norm:
            y1 = _layer_norm(y1, sd["norms.%d.weight" % t].to(dtype), sd["norms.%d.bias" % t].to(dtype))   # conv.py:264
        y2 = _gelu_erf(y1 @ Wm.T + bm) @ Wo.T + bo + y1             # conv.py:271
        out[rows] = _layer_norm(y2, wn, bn)                         # conv.py:272
    return out


# --------------------------------------------------------------------------
# entry point 1: node-level closed form (the parity checker)
# --------------------------------------------------------------------------
def forward_closed_form(sd, num_types, num_relations, n_heads, x, node_type, edge_index,
                        edge_type, edge_time=None, use_norm=True, use_RTE=True,
                        dtype=torch.float64, return_att=False, return_agg=False, dense=False):
    """One HGTConv (or, with dense=True, DenseHGTConv: same message(), conv.py:197-248, update of conv.py:250-274)
    forward (eval mode).  Math follows conv.py:60-134:

      q_e = W_q[tau(i)] x_i + b          (conv.py:73-77,96)
      k_e = W_k[tau(j)] (x_j + RTE(dt_e)) + b   (conv.py:91-92,97)
      k'_e,h = k_e,h . A[phi(e),h]       (conv.py:98)
      s_e,h = <q_e,h, k'_e,h> * pri[phi(e),h] / sqrt(dk)   (conv.py:99)
      att = softmax over incoming edges of i  (conv.py:108)
      msg_e,h = (v_e,h . M[phi(e),h]) * att_e,h           (conv.py:104,109)
      agg_i = sum_e msg_e ; out = update(agg)             (conv.py:13,114-134)

    Linear maps are applied once per NODE (x -> Q,K,V) plus an additive
    per-(type, dt) table for the temporal term; that is algebraically the
    per-edge form of the reference because nn.Linear is affine.
    Edges whose node/relation ids fall outside [0,T)/[0,R) keep logit 0 and
    message 0 but stay in the softmax, like the reference's zero-initialised
    res_att/res_msg (conv.py:68-69).
    """
    T, R, H = num_types, num_relations, n_heads
    x = x.to(dtype)
    N, d_in = x.shape
    d = sd["relation_att"].shape[1] * sd["relation_att"].shape[2]
    dk = d // H
    src, dst = edge_index[0].long(), edge_index[1].long()
    E = src.numel()
    etype = edge_type.long()
    ntype = node_type.long()

    Wq = _stack(sd, "q_linears.%d.weight", T, dtype); bq = _stack(sd, "q_linears.%d.bias", T, dtype)
    Wk = _stack(sd, "k_linears.%d.weight", T, dtype); bk = _stack(sd, "k_linears.%d.bias", T, dtype)
    Wv = _stack(sd, "v_linears.%d.weight", T, dtype); bv = _stack(sd, "v_linears.%d.bias", T, dtype)

    Q = torch.zeros(N, d, dtype=dtype); K = torch.zeros(N, d, dtype=dtype); V = torch.zeros(N, d, dtype=dtype)
    for t in range(T):
        rows = (ntype == t).nonzero(as_tuple=True)[0]
        if rows.numel() == 0:
            continue
        xt = x[rows]
        Q[rows] = xt @ Wq[t].T + bq[t]
        K[rows] = xt @ Wk[t].T + bk[t]
        V[rows] = xt @ Wv[t].T + bv[t]

    tj = ntype[src]
    ti = ntype[dst]
    valid = (tj >= 0) & (tj < T) & (ti >= 0) & (ti < T) & (etype >= 0) & (etype < R)
    k_e = K[src]
    v_e = V[src]
    if use_RTE:
        if edge_time is None:
            raise ValueError("use_RTE=True needs edge_time")
        rte = sd["emb.emb.weight"].to(dtype) @ sd["emb.lin.weight"].to(dtype).T + sd["emb.lin.bias"].to(dtype)
        rte_k = torch.einsum("pd,tod->tpo", rte, Wk)   # [T, 240, d] = rte @ Wk[t].T
        rte_v = torch.einsum("pd,tod->tpo", rte, Wv)
        tjc = tj.clamp(0, T - 1)
        et = edge_time.long()
        k_e = k_e + rte_k[tjc, et]
        v_e = v_e + rte_v[tjc, et]
    q_e = Q[dst]

    # relation transforms, one relation at a time (keeps memory at O(E*d))
    A_all = sd["relation_att"].to(dtype)
    M_all = sd["relation_msg"].to(dtype)
    pri_all = sd["relation_pri"].to(dtype)
    s = torch.zeros(E, H, dtype=dtype)                 # unclaimed edges keep logit 0 (conv.py:68)
    vp = torch.zeros(E, H, dk, dtype=dtype)            # ... and message 0 (conv.py:69)
    for r in range(R):
        sel = ((etype == r) & valid).nonzero(as_tuple=True)[0]
        if sel.numel() == 0:
            continue
        kp = torch.einsum("ehk,hkc->ehc", k_e[sel].view(-1, H, dk), A_all[r])
        s[sel] = (q_e[sel].view(-1, H, dk) * kp).sum(-1) * pri_all[r] / math.sqrt(dk)
        vp[sel] = torch.einsum("ehk,hkc->ehc", v_e[sel].view(-1, H, dk), M_all[r])

    if E > 0:
        att = _segment_softmax(s, dst, N)
    else:
        att = s
    msg = (vp * att.unsqueeze(-1)).reshape(E, d)
    agg = torch.zeros(N, d, dtype=dtype).index_add_(0, dst, msg)
    out = (_update_dense if dense else _update)(sd, agg, x, ntype, T, use_norm, dtype)
    res = [out]
    if return_att:
        res.append(att)
    if return_agg:
        res.append(agg)
    return res[0] if len(res) == 1 else tuple(res)


# --------------------------------------------------------------------------
# entry point 2: reference-cost port (what bench.py times on the host cores)
# --------------------------------------------------------------------------
def forward_meta_relation_port(sd, num_types, num_relations, n_heads, x, node_type, edge_index,
                               edge_type, edge_time=None, use_norm=True, use_RTE=True,
                               return_att=False):
    """fp32 port with the reference's cost structure (conv.py:56-134 + PyG
    propagate): materialise x_i/x_j per edge, walk the T x T x R meta-relation
    cube, and for every non-empty meta relation run the three Linear layers
    and the two per-head d_k x d_k transforms on the EDGE rows, writing into
    E x H / E x d scratch; then segment softmax, scatter-add and update."""
    T, R, H = num_types, num_relations, n_heads
    dtype = torch.float32
    x = x.to(dtype)
    N = x.shape[0]
    d = sd["relation_att"].shape[1] * sd["relation_att"].shape[2]
    dk = d // H
    src, dst = edge_index[0].long(), edge_index[1].long()
    E = src.numel()
    x_i = x.index_select(0, dst)                        # propagate(): *_i gathers
    x_j = x.index_select(0, src)                        # propagate(): *_j gathers
    t_i = node_type.long().index_select(0, dst)
    t_j = node_type.long().index_select(0, src)
    logits = torch.zeros(E, H, dtype=dtype)             # conv.py:68
    msg = torch.zeros(E, H, dk, dtype=dtype)            # conv.py:69
    inv = 1.0 / math.sqrt(dk)
    for st in range(T):                                 # conv.py:71
        from_st = t_j == st
        for tt in range(T):                             # conv.py:75
            st_tt = from_st & (t_i == tt)
            for r in range(R):                          # conv.py:78
                sel = st_tt & (edge_type == r)
                if not bool(sel.any()):                 # conv.py:83
                    continue
                xs = x_j[sel]
                if use_RTE:                             # conv.py:91-92
                    tvec = sd["emb.emb.weight"][edge_time[sel].long()]
                    xs = xs + torch.addmm(sd["emb.lin.bias"], tvec, sd["emb.lin.weight"].T)
                qm = torch.addmm(sd["q_linears.%d.bias" % tt], x_i[sel], sd["q_linears.%d.weight" % tt].T)
                km = torch.addmm(sd["k_linears.%d.bias" % st], xs, sd["k_linears.%d.weight" % st].T)
                vm = torch.addmm(sd["v_linears.%d.bias" % st], xs, sd["v_linears.%d.weight" % st].T)
                km = torch.bmm(km.view(-1, H, dk).transpose(0, 1), sd["relation_att"][r]).transpose(0, 1)
                logits[sel] = (qm.view(-1, H, dk) * km).sum(-1) * sd["relation_pri"][r] * inv
                msg[sel] = torch.bmm(vm.view(-1, H, dk).transpose(0, 1), sd["relation_msg"][r]).transpose(0, 1)
    att = _segment_softmax(logits, dst, N) if E > 0 else logits
    res = (msg * att.unsqueeze(-1)).reshape(E, d)
    agg = torch.zeros(N, d, dtype=dtype).index_add_(0, dst, res)
    out = _update(sd, agg, x, node_type.long(), T, use_norm, dtype, library_ops=True)
    return (out, att) if return_att else out


# --------------------------------------------------------------------------
# entry point 3: gradients (oracle for the backward pass, SURVEY.md section 8f-2 -- not built on the GPU yet)
# --------------------------------------------------------------------------
def backward_reference(sd, num_types, num_relations, n_heads, x, node_type, edge_index, edge_type, edge_time, grad_out,
                       use_norm=True, use_RTE=True, dtype=torch.float64, dense=False):
    """d<out, grad_out> / d(x) and / d(every floating parameter of sd), by reverse-mode differentiation of
    forward_closed_form (every step of it is a differentiable torch op, so this is the exact backward of the math of
    conv.py:60-134 in eval mode).  Pinned against autograd through the verbatim reference in tests/test_oracle.py."""
    leaf = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    xg = x.detach().to(dtype).requires_grad_(True)
    out = forward_closed_form(leaf, num_types, num_relations, n_heads, xg, node_type, edge_index, edge_type, edge_time,
                              use_norm=use_norm, use_RTE=use_RTE, dtype=dtype, dense=dense)
    names = list(leaf.keys())
    grads = torch.autograd.grad((out * grad_out.to(dtype)).sum(), [xg] + [leaf[k] for k in names], allow_unused=True)
    res = {"x": grads[0]}
    for k, g in zip(names, grads[1:]):
        res[k] = g if g is not None else torch.zeros_like(leaf[k])
    return res


# ---------------------------------------------------------------------------------------------------------------
# GNN wrapper (model.py:54-80): seeded parameters in the reference's state_dict layout, and the wrapper restated
# ---------------------------------------------------------------------------------------------------------------
def make_gnn_state_dict(in_dim, n_hid, num_types, num_relations, n_heads, n_layers, prev_norm, last_norm, use_RTE, seed=0):
    """state_dict of `GNN(in_dim, n_hid, ...)` (model.py:51-64): `adapt_ws.t.{weight,bias}` + `gcs.l.base_conv.<layer key>`.
    A pure function of its arguments (torch CPU generator), so that the GNN goldens (oracle/gen_golden_gnn.py) need not
    store 21 M parameters: the generator and the tests rebuild the same tensors from the seed."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    b = 1.0 / math.sqrt(in_dim)
    for t in range(num_types):
        sd["adapt_ws.%d.weight" % t] = (torch.rand((n_hid, in_dim), generator=g) * 2 - 1) * b
        sd["adapt_ws.%d.bias" % t] = (torch.rand((n_hid,), generator=g) * 2 - 1) * b
    for l in range(n_layers):
        norm = last_norm if l == n_layers - 1 else prev_norm
        layer = make_state_dict(n_hid, n_hid, num_types, num_relations, n_heads, norm, use_RTE, seed=seed * 1000 + 17 * l + 1)
        for k, v in layer.items():
            sd["gcs.%d.base_conv.%s" % (l, k)] = v
    return sd


def gnn_forward(sd, in_dim, n_hid, num_types, num_relations, n_heads, n_layers, prev_norm, last_norm, use_RTE,
                node_feature, node_type, edge_time, edge_index, edge_type, dtype=torch.float64, return_layers=False):
    """model.py:66-80 restated (eval mode: dropout is the identity): typed adapter + tanh, then the stacked layers through
    forward_closed_form.  Returns the output (and the list [adapter output, layer 1 output, ...] with return_layers)."""
    x = node_feature.to(dtype)
    h = torch.zeros(x.size(0), n_hid, dtype=dtype)
    for t in range(num_types):                                    # model.py:70-75
        m = node_type == t
        if m.any():
            h[m] = torch.tanh(x[m] @ sd["adapt_ws.%d.weight" % t].to(dtype).T + sd["adapt_ws.%d.bias" % t].to(dtype))
    layers = [h]
    for l in range(n_layers):                                     # model.py:78-79
        pre = "gcs.%d.base_conv." % l
        lsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        norm = last_norm if l == n_layers - 1 else prev_norm
        h = forward_closed_form(lsd, num_types, num_relations, n_heads, h, node_type, edge_index, edge_type, edge_time,
                                use_norm=norm, use_RTE=use_RTE, dtype=dtype)
        layers.append(h)
    return (h, layers) if return_layers else h
