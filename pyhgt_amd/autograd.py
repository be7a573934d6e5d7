"""Training path of HGTConv / DenseHGTConv: forward that keeps its intermediates + the hand-written backward.

The reference trains through autograd over conv.py:60-134 (`loss.backward()`, OAG/train_paper_field.py:249,
ogbn-mag/train_ogbn_mag.py:172).  Here a `torch.autograd.Function` wraps the HIP kernels: the forward is the same
node-level algebra as the inference path (typed Q|K|V projections once per node, target-side relation transforms, softmax
over in-edges, relation-wise aggregation on the matrix cores), run kernel by kernel so that Q, K, V, the attention weights,
the aggregate and the a_linear output stay available; the backward strings together the kernels of
csrc/hgt_backward.hip + the forward kernels on the transposed graph (the derivation is in that file's header).

PyTorch's role: the Function boundary, parameter packing with differentiable stack/cat/pad ops (so the gradients of the
packed arrays flow back to the reference-named parameters by themselves), tiny O(R H d_k^2) / O(T 240 d) chain-rule steps on
relation matrices and temporal tables, and the dropout mask (torch RNG, like the reference's nn.Dropout at conv.py:125).
There is no CPU or eager fallback: CPU tensors raise.
"""
import ctypes as C
import math

import torch

from . import _lib

__all__ = ["hgt_conv_train", "TypedLinearFunction"]


def _p(t):
    return 0 if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _chk(name, rc):
    _lib.check(rc, name)


class _Ops:
    """Thin typed wrappers around the C ABI calls the training path needs (all on the current stream)."""

    def __init__(self, plan, lay, T, R, H, precision):
        self.lib = _lib.load()
        self.plan, self.lay = plan, lay
        self.T, self.R, self.Hreal = T, R, H
        self.H = lay.heads                                   # kernels run with the layout's head count (extra heads: zero)
        self.dk, self.dkp, self.dp = lay.d_k, lay.dk_pad, lay.d_pad
        self.split = precision in ("bf16x3", "f16x3")      # (the differentiable path evaluates "f16x3" layers with the bf16 split)
        self.N, self.E = plan.N, plan.E
        self.dev = plan.device
        self._scratch = {}

    # -- dense ---------------------------------------------------------------------------------------------
    def typed_linear(self, x, ldx, rows, off, n_groups, n_rows, k, n_out, W, w_off, wgs, bias, b_off, bgs, outs, block_cols,
                     by_pos=0, prologue=0):
        """y = prologue(x[rows]) W[g]^T + b[g]; W / bias given as (tensor, element offset).  outs: up to 3 column blocks."""
        lib = self.lib
        o = [_p(t) for t in outs] + [0] * (3 - len(outs))
        wp = W.data_ptr() + 4 * w_off
        bp = 0 if bias is None else bias.data_ptr() + 4 * b_off
        if self.split and (n_out % 4 == 0) and (block_cols % 4 == 0):
            nb = C.c_uint64()
            _chk("hgt_split_weights_bytes", lib.hgt_split_weights_bytes(n_groups, k, n_out, C.byref(nb)))
            tiles = torch.empty(int(nb.value), dtype=torch.uint8, device=self.dev)
            _chk("hgt_split_weights", lib.hgt_split_weights(wp, wgs, n_groups, k, n_out, _p(tiles), _st()))
            _chk("hgt_typed_linear_bf16x3", lib.hgt_typed_linear_bf16x3(_p(x), ldx, rows, off, n_groups, n_rows, k, n_out, _p(tiles), bp, bgs,
                                                                      o[0], o[1], o[2], block_cols, by_pos, prologue, _st()))
            tiles.record_stream(torch.cuda.current_stream())
        else:
            _chk("hgt_typed_linear", lib.hgt_typed_linear(_p(x), ldx, rows, off, n_groups, n_rows, k, n_out, wp, wgs, bp, bgs,
                                                        o[0], o[1], o[2], block_cols, by_pos, prologue, 0, _st()))

    def wgrad(self, A, lda, B, ldb, rows, off, n_groups, n_rows, m, n_cols, with_colsum=False):
        """dW[g] = A_g^T B_g (and, with_colsum, db[g] = column sums of A_g from the same pass): split-bf16 x3 MFMA kernel for
        bf16x3 layers, the exact fp32 MFMA kernel (+ the column-sum kernel) otherwise."""
        out = torch.zeros(n_groups, m, n_cols, dtype=torch.float32, device=self.dev)
        cs = torch.zeros(n_groups, m, dtype=torch.float32, device=self.dev) if with_colsum else None
        if self.split:
            _chk("hgt_typed_wgrad_bf16x3", self.lib.hgt_typed_wgrad_bf16x3(_p(A), lda, _p(B), ldb, rows, off, n_groups, n_rows, m, n_cols,
                                                                         _p(out), m * n_cols, _p(cs), m, _st()))
        else:
            _chk("hgt_typed_wgrad", self.lib.hgt_typed_wgrad(_p(A), lda, _p(B), ldb, rows, off, n_groups, n_rows, m, n_cols, _p(out),
                                                           m * n_cols, _st()))
            if with_colsum:
                _chk("hgt_typed_colsum", self.lib.hgt_typed_colsum(_p(A), lda, rows, off, n_groups, n_rows, m, _p(cs), m, _st()))
        return (out, cs) if with_colsum else out

    def colsum(self, A, lda, rows, off, n_groups, n_rows, m):
        out = torch.zeros(n_groups, m, dtype=torch.float32, device=self.dev)
        _chk("hgt_typed_colsum", self.lib.hgt_typed_colsum(_p(A), lda, rows, off, n_groups, n_rows, m, _p(out), m, _st()))
        return out

    # -- relation matrices ------------------------------------------------------------------------------------
    def pack(self, att_like, msg_like, pri):
        """hgt_relation_pack: (att_t[r,h,c,k] = att_like[r,h,k,c] * pri / sqrt(dk), msg_p = msg_like), zero padded to dk_pad."""
        R, H, dkp = self.R, self.H, self.dkp
        att_t = torch.empty(R, H, dkp, dkp, dtype=torch.float32, device=self.dev)
        msg_p = torch.empty(R, H, dkp, dkp, dtype=torch.float32, device=self.dev)
        _chk("hgt_relation_pack", self.lib.hgt_relation_pack(_p(att_like.contiguous()), _p(msg_like.contiguous()), _p(pri.contiguous()), R,
                                                           self.Hreal, H, self.dk, dkp, _p(att_t), _p(msg_p), _st()))
        return att_t, msg_p

    def frags(self, msg_p):
        nb = C.c_uint64()
        _chk("hgt_relation_frag_bytes", self.lib.hgt_relation_frag_bytes(self.R, self.H, self.dkp, C.byref(nb)))
        if nb.value == 0:
            raise RuntimeError("pyhgt_amd: training needs a layout the matrix-core aggregation covers (a head of at most 256 columns)")
        f = torch.empty(int(nb.value), dtype=torch.uint8, device=self.dev)
        _chk("hgt_relation_frag_pack", self.lib.hgt_relation_frag_pack(_p(msg_p), self.R, self.H, self.dkp, _p(f), _st()))
        return f

    # -- edge phase -------------------------------------------------------------------------------------------
    def _hub_ws(self, plan):
        nb = C.c_uint64()
        _chk("hgt_hub_workspace_bytes", self.lib.hgt_hub_workspace_bytes(plan.E, self.H, self.dkp, C.byref(nb)))
        return torch.empty(max(int(nb.value), 256), dtype=torch.uint8, device=self.dev)

    def logits(self, plan, Q, K, rte_k, att_t):
        out = torch.empty(plan.E, self.H, dtype=torch.float32, device=self.dev)
        _chk("hgt_edge_logits", self.lib.hgt_edge_logits(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, self.dkp, _p(Q), _p(K), _p(rte_k),
                                                       _p(att_t), _p(out), _st()))
        return out

    def softmax_(self, plan, logits):
        _chk("hgt_edge_softmax", self.lib.hgt_edge_softmax(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, _p(logits), _st()))
        return logits

    def _items_scratch(self, plan):
        """Scratch of the item-parallel gather passes (sampled batches), one buffer per plan size, reused by every call of a step."""
        key = ("items", plan.E)
        buf = self._scratch.get(key)
        if buf is None:
            nb = C.c_uint64()
            _chk("hgt_edge_aggregate_items_bytes", self.lib.hgt_edge_aggregate_items_bytes(plan.E, self.H, self.dkp, C.byref(nb)))
            buf = self._scratch[key] = torch.empty(max(int(nb.value), 16), dtype=torch.uint8, device=self.dev)
        return buf

    def spmm(self, plan, w, rows_ptr, rte_rows, f_p, f_frag, out, out_col, ld_out, n_q_rows):
        # sampled batches: the item-parallel form (one wavefront per <= 16-edge item + an ordered merge) -- the sub-tile kernel's
        # wavefronts each walk sixteen targets' edges one after the other (285 vs ~25 us per call at c3, round 6)
        if plan.N < 65536 and plan.E > 0 and self.R < 64 and ld_out % 4 == 0 and out_col % 4 == 0:
            sc = self._items_scratch(plan)
            rc = self.lib.hgt_edge_spmm_items(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, self.dkp, _p(w), rows_ptr, _p(rte_rows),
                                              _p(f_frag), out.data_ptr() + 4 * out_col, ld_out, n_q_rows, _p(sc), sc.numel(), _st())
            if rc == 0:
                sc.record_stream(torch.cuda.current_stream())
                return
            if rc != -2:
                _chk("hgt_edge_spmm_items", rc)
        hub = self._hub_ws(plan)
        _chk("hgt_edge_spmm", self.lib.hgt_edge_spmm(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, self.dkp, _p(w), rows_ptr,
                                                   _p(rte_rows), _p(f_p), _p(f_frag), out.data_ptr() + 4 * out_col, ld_out, n_q_rows,
                                                   _p(hub), _st()))
        hub.record_stream(torch.cuda.current_stream())

    def to_edge_ids(self, plan, sorted_vals):
        out = torch.empty_like(sorted_vals)
        _chk("hgt_att_export", self.lib.hgt_att_export(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, _p(sorted_vals), _p(out), self.H,
                                                     _st()))
        return out

    def to_sorted(self, plan, by_id):
        out = torch.empty_like(by_id)
        _chk("hgt_edge_gather_sorted", self.lib.hgt_edge_gather_sorted(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, _p(by_id), _p(out),
                                                                     _st()))
        return out

    def outer(self, plan, w, a, rte_a, b):
        out = torch.zeros(self.R, self.H, self.dkp, self.dkp, dtype=torch.float32, device=self.dev)
        _chk("hgt_relation_outer", self.lib.hgt_relation_outer(plan.ptr, plan.N, plan.E, self.T, self.R, self.H, self.dkp, _p(w), _p(a),
                                                             _p(rte_a), _p(b), _p(out), _st()))
        return out


def _rte_row_lists(T, dev):
    rows = (torch.arange(T * _lib.HGT_RTE_LEN, device=dev) % _lib.HGT_RTE_LEN).to(torch.int32)
    off = (torch.arange(T + 1, device=dev) * _lib.HGT_RTE_LEN).to(torch.int32)
    return rows, off


class _HGTConvTrain(torch.autograd.Function):
    """HGTConv / DenseHGTConv forward + backward on the HIP kernels.  The message path (conv.py:60-111) is shared; the update is
    conv.py:114-134 (HGTConv: gelu, a_linear, dropout, gated skip, LayerNorm) or conv.py:250-274 (DenseHGTConv: a_linear, dropout,
    plain residual, LayerNorm, then the shared dense layer with its own dropout and out_norm)."""

    @staticmethod
    def forward(ctx, layer, plan, drop_masks, x, w_qkv, b_qkv, w_a, b_a, ratt, rmsg, rpri, skip, ln_w, ln_b, rte_emb, rte_w, rte_b,
                mid_w, mid_b, out_w, out_b, oln_w, oln_b):
        lib = _lib.load()
        lay = _lib.layout_for(layer.out_dim, layer.n_heads)
        T, R, H = layer.num_types, layer.num_relations, layer.n_heads
        ops = _Ops(plan, lay, T, R, H, layer.precision)
        N, din, dout, dp = plan.N, layer.in_dim, layer.out_dim, lay.d_pad
        dev = x.device
        use_rte, use_norm = bool(layer.use_RTE), bool(layer.use_norm)
        dense = mid_w is not None
        rows = plan.row_lists()
        x = x.contiguous()
        att_t, msg_p = ops.pack(ratt, rmsg, rpri)
        msg_f = ops.frags(msg_p)
        # projections (conv.py:96-97,103 once per node)
        qkv = torch.empty(3, N, dp, dtype=torch.float32, device=dev)
        ops.typed_linear(x, din, rows.rows_all, rows.off_all, T, N, din, 3 * dp, w_qkv, 0, 3 * dp * din, b_qkv, 0, 3 * dp,
                         [qkv[0], qkv[1], qkv[2]], dp)
        rte_k = rte_v = rte_lin = None
        if use_rte:     # temporal tables (conv.py:91-92,298-299 hoisted off the edges)
            rr, ro = _rte_row_lists(T, dev)
            rte_lin = torch.empty(_lib.HGT_RTE_LEN, din, dtype=torch.float32, device=dev)
            ops.typed_linear(rte_emb, din, rr.data_ptr(), ro.data_ptr(), 1, _lib.HGT_RTE_LEN, din, din, rte_w, 0, 0, rte_b, 0, 0, [rte_lin], din,
                             by_pos=1)
            rte_kv = torch.empty(2, T * _lib.HGT_RTE_LEN, dp, dtype=torch.float32, device=dev)
            ops.typed_linear(rte_lin, din, rr.data_ptr(), ro.data_ptr(), T, T * _lib.HGT_RTE_LEN, din, 2 * dp, w_qkv, dp * din, 3 * dp * din,
                             None, 0, 0, [rte_kv[0], rte_kv[1]], dp, by_pos=1)
            rte_k, rte_v = rte_kv[0], rte_kv[1]
        # attention (conv.py:98-99,108): logits in sorted edge order, normalised in place
        att = ops.softmax_(plan, ops.logits(plan, qkv[0], qkv[1], rte_k, att_t))
        if layer.keep_att and E > 0:      # self.att (conv.py:108) in the caller's edge order, like the inference path
            a_out = torch.empty(E, layer.n_heads, dtype=torch.float32, device=dev)
            _chk("hgt_att_export", lib.hgt_att_export(plan.ptr, N, E, T, R, ops.H, _p(att), _p(a_out), layer.n_heads, _st()))
            layer.att = a_out
        else:
            layer.att = None
        # aggregation (conv.py:104,109-111 + scatter-add): agg = sum_r (sum_e att_e v_e) M_r
        agg = torch.empty(N, dp, dtype=torch.float32, device=dev)
        ops.spmm(plan, att, qkv[2].data_ptr(), rte_v, msg_p, msg_f, agg, 0, dp, N)
        m1, m2 = drop_masks if drop_masks is not None else (None, None)
        empty = x.new_empty(0)
        trans = torch.empty(N, dout, dtype=torch.float32, device=dev)
        out = torch.empty(N, dout, dtype=torch.float32, device=dev)
        if not dense:
            # update (conv.py:119-133): a_linear(gelu(agg)) -> dropout -> gated skip -> LayerNorm
            ops.typed_linear(agg, dp, rows.rows_q, rows.off_q, T, N, dp, dout, w_a, 0, dout * dp, b_a, 0, dout, [trans], dout, prologue=1)
            if m1 is not None:
                _chk("hgt_mul_inplace", lib.hgt_mul_inplace(_p(trans), _p(m1), trans.numel(), _st()))
            _chk("hgt_node_update", lib.hgt_node_update(_p(trans), _p(x), din, _p(plan.node_type), _p(skip), _p(ln_w), _p(ln_b), int(use_norm),
                                                      N, dout, T, _p(out), _st()))
            y1 = mid = trans2 = off2 = empty
        else:
            # DenseHGTConv.update (conv.py:250-274): y1 = LN_t(drop(a_linear(agg)) + x); out = out_norm(drop(out_linear(gelu(mid_linear(y1)))) + y1)
            ops.typed_linear(agg, dp, rows.rows_q, rows.off_q, T, N, dp, dout, w_a, 0, dout * dp, b_a, 0, dout, [trans], dout)
            if m1 is not None:
                _chk("hgt_mul_inplace", lib.hgt_mul_inplace(_p(trans), _p(m1), trans.numel(), _st()))
            y1 = torch.empty(N, dout, dtype=torch.float32, device=dev)
            _chk("hgt_node_update_ex", lib.hgt_node_update_ex(_p(trans), _p(x), din, _p(plan.node_type), None, _p(ln_w), _p(ln_b),
                                                            int(use_norm), 0, N, dout, T, _p(y1), _st()))
            off2 = torch.empty(2, dtype=torch.int32, device=dev)
            _chk("hgt_single_group_offsets", lib.hgt_single_group_offsets(rows.off_q, T, _p(off2), _st()))
            mid = torch.zeros(N, 2 * dout, dtype=torch.float32, device=dev)
            ops.typed_linear(y1, dout, rows.rows_q, off2.data_ptr(), 1, N, dout, 2 * dout, mid_w, 0, 0, mid_b, 0, 0, [mid], 2 * dout)
            trans2 = torch.zeros(N, dout, dtype=torch.float32, device=dev)
            ops.typed_linear(mid, 2 * dout, rows.rows_q, off2.data_ptr(), 1, N, 2 * dout, dout, out_w, 0, 0, out_b, 0, 0, [trans2], dout,
                             prologue=1)
            if m2 is not None:
                _chk("hgt_mul_inplace", lib.hgt_mul_inplace(_p(trans2), _p(m2), trans2.numel(), _st()))
            _chk("hgt_node_update_ex", lib.hgt_node_update_ex(_p(trans2), _p(y1), dout, _p(plan.node_type), None, _p(oln_w), _p(oln_b), 1, 1,
                                                            N, dout, T, _p(out), _st()))
        ctx.layer, ctx.plan, ctx.lay = layer, plan, lay
        ctx.use_rte, ctx.use_norm, ctx.dense = use_rte, use_norm, dense
        ctx.save_for_backward(x, w_qkv, w_a, ratt, rmsg, rpri, skip if skip is not None else empty, ln_w if ln_w is not None else empty,
                              rte_emb, rte_w, rte_b, qkv, att, agg, trans,
                              m1 if m1 is not None else empty, m2 if m2 is not None else empty, rte_k if use_rte else empty,
                              rte_v if use_rte else empty, y1, mid, trans2, off2,
                              mid_w if dense else empty, out_w if dense else empty, oln_w if dense else empty)
        return out

    @staticmethod
    def backward(ctx, gout):
        layer, plan, lay = ctx.layer, ctx.plan, ctx.lay
        (x, w_qkv, w_a, ratt, rmsg, rpri, skip, ln_w, rte_emb, rte_w, rte_b, qkv, att, agg, trans, m1, m2, rte_k, rte_v, y1, mid, trans2,
         off2, mid_w, out_w, oln_w) = ctx.saved_tensors
        lib = _lib.load()
        T, R, H = layer.num_types, layer.num_relations, layer.n_heads
        ops = _Ops(plan, lay, T, R, H, layer.precision)
        # (data gradients d gelu(agg), dx run on the layer's own typed-linear kernels: split-bf16 x3 by default, relative error
        #  ~1e-5, two orders below the gradient tolerance; precision='fp32' layers keep the exact typed-linear kernel.  The
        #  relation transforms of every hgt_edge_spmm -- the training forward's aggregation included -- are ALWAYS split-bf16 MFMA
        #  products under grad, also for precision='fp32': ~1e-5 away from the exact VALU aggregation of the inference path)
        N, E, din, dout, dp, dk, dkp = plan.N, plan.E, layer.in_dim, layer.out_dim, lay.d_pad, lay.d_k, lay.dk_pad
        Hr, H = H, lay.heads                                 # model heads / layout heads
        dev = x.device
        use_rte, use_norm, dense = ctx.use_rte, ctx.use_norm, ctx.dense
        rows = plan.row_lists()
        gout = gout.contiguous().float()
        m1 = None if m1.numel() == 0 else m1
        m2 = None if m2.numel() == 0 else m2
        if not use_rte:
            rte_k = rte_v = None
        Q, K, V = qkv[0], qkv[1], qkv[2]
        d_skip = d_mid_w = d_mid_b = d_out_w = d_out_b = d_oln_w = d_oln_b = None
        d_lnw = torch.zeros(T, dout, dtype=torch.float32, device=dev) if use_norm else None
        d_lnb = torch.zeros(T, dout, dtype=torch.float32, device=dev) if use_norm else None
        d_trans = torch.empty(N, dout, dtype=torch.float32, device=dev)
        dx_skip = torch.empty(N, din, dtype=torch.float32, device=dev)
        dagg = torch.empty(N, dp, dtype=torch.float32, device=dev)
        w_a_t = w_a.transpose(1, 2).contiguous()                                     # [T][dp][dout]
        if not dense:
            # ---- update backward (conv.py:125-133)
            d_alpha = torch.zeros(T, dtype=torch.float32, device=dev)
            _chk("hgt_node_update_bwd", lib.hgt_node_update_bwd(_p(gout), _p(trans), _p(x), din, _p(plan.node_type), _p(skip), _p(ln_w),
                                                              int(use_norm), _p(m1), N, dout, T, _p(d_trans), _p(dx_skip), din,
                                                              _p(d_alpha), _p(d_lnw), _p(d_lnb), _st()))
            alpha = torch.sigmoid(skip)
            d_skip = d_alpha * alpha * (1.0 - alpha)
            # a_linear: trans = gelu(agg) W_a^T + b_a
            g = torch.nn.functional.gelu(agg)                                       # exact erf form, conv.py:119
            d_w_a, d_b_a = ops.wgrad(d_trans, dout, g, dp, rows.rows_q, rows.off_q, T, N, dout, dp, with_colsum=True)
            del g
            dg = torch.empty(N, dp, dtype=torch.float32, device=dev)
            ops.typed_linear(d_trans, dout, rows.rows_q, rows.off_q, T, N, dout, dp, w_a_t, 0, dp * dout, None, 0, 0, [dg], dp)
            _chk("hgt_gelu_bwd", lib.hgt_gelu_bwd(_p(dg), _p(agg), _p(dagg), dagg.numel(), _st()))
            del dg
        else:
            # ---- DenseHGTConv.update backward (conv.py:250-274 in reverse)
            d_oln_w = torch.zeros(1, dout, dtype=torch.float32, device=dev)
            d_oln_b = torch.zeros(1, dout, dtype=torch.float32, device=dev)
            d_t2 = torch.empty(N, dout, dtype=torch.float32, device=dev)           # gradient of out_linear's (dropped) output
            d_y1 = torch.empty(N, dout, dtype=torch.float32, device=dev)           # residual branch of y1
            _chk("hgt_node_update_bwd_ex", lib.hgt_node_update_bwd_ex(_p(gout), _p(trans2), _p(y1), dout, _p(plan.node_type), None, _p(oln_w),
                                                                    1, 1, _p(m2), N, dout, T, _p(d_t2), _p(d_y1), dout, None, _p(d_oln_w),
                                                                    _p(d_oln_b), _st()))
            g2 = torch.nn.functional.gelu(mid)
            d_out_w, d_out_b = ops.wgrad(d_t2, dout, g2, 2 * dout, rows.rows_q, off2.data_ptr(), 1, N, dout, 2 * dout, with_colsum=True)
            del g2
            out_w_t = out_w.t().contiguous()                                           # [2 dout][dout]
            d_g2 = torch.zeros(N, 2 * dout, dtype=torch.float32, device=dev)
            ops.typed_linear(d_t2, dout, rows.rows_q, off2.data_ptr(), 1, N, dout, 2 * dout, out_w_t, 0, 0, None, 0, 0, [d_g2], 2 * dout)
            d_mid = torch.empty_like(d_g2)
            _chk("hgt_gelu_bwd", lib.hgt_gelu_bwd(_p(d_g2), _p(mid), _p(d_mid), d_mid.numel(), _st()))
            del d_g2
            d_mid_w, d_mid_b = ops.wgrad(d_mid, 2 * dout, y1, dout, rows.rows_q, off2.data_ptr(), 1, N, 2 * dout, dout, with_colsum=True)
            mid_w_t = mid_w.t().contiguous()                                           # [dout][2 dout]
            d_y1b = torch.zeros(N, dout, dtype=torch.float32, device=dev)
            ops.typed_linear(d_mid, 2 * dout, rows.rows_q, off2.data_ptr(), 1, N, 2 * dout, dout, mid_w_t, 0, 0, None, 0, 0, [d_y1b], dout)
            d_y1 += d_y1b
            del d_mid, d_y1b
            _chk("hgt_node_update_bwd_ex", lib.hgt_node_update_bwd_ex(_p(d_y1), _p(trans), _p(x), din, _p(plan.node_type), None, _p(ln_w),
                                                                    int(use_norm), 0, _p(m1), N, dout, T, _p(d_trans), _p(dx_skip), din,
                                                                    None, _p(d_lnw), _p(d_lnb), _st()))
            d_w_a, d_b_a = ops.wgrad(d_trans, dout, agg, dp, rows.rows_q, rows.off_q, T, N, dout, dp, with_colsum=True)
            ops.typed_linear(d_trans, dout, rows.rows_q, rows.off_q, T, N, dout, dp, w_a_t, 0, dp * dout, None, 0, 0, [dagg], dp)
            d_oln_w, d_oln_b = d_oln_w[0], d_oln_b[0]
            d_out_w, d_out_b, d_mid_w, d_mid_b = d_out_w[0], d_out_b[0], d_mid_w[0], d_mid_b[0]
        # rows of an unknown type get no a_linear (their agg gradient is zero): typed_linear leaves them unwritten
        _chk("hgt_zero_rows", lib.hgt_zero_rows(rows.rows_q, rows.off_q + 4 * T, dp, _p(dagg), _st()))

        # ---- aggregation / attention backward (conv.py:98-111)
        sqrt_dk = math.sqrt(dk)
        ones_pri = torch.full((R, Hr), sqrt_dk, dtype=torch.float32, device=dev)    # pri / sqrt(dk) == 1
        m_t, _ = ops.pack(rmsg, rmsg, ones_pri)                                      # m_t[r,h,c,k] = M[r,h,k,c]
        d_att = ops.logits(plan, dagg, V, rte_v, m_t)                                # <dagg_i M^T, v_e>
        rho = torch.empty(N, H, dtype=torch.float32, device=dev)
        _chk("hgt_head_dot", lib.hgt_head_dot(_p(dagg), _p(agg), N, H, dkp, _p(rho), _st()))
        ds = torch.empty(E, H, dtype=torch.float32, device=dev)
        _chk("hgt_edge_softmax_bwd", lib.hgt_edge_softmax_bwd(plan.ptr, N, E, T, R, H, _p(att), _p(d_att), _p(rho), H, _p(ds), _st()))
        del d_att
        scale = (rpri / sqrt_dk).view(R, Hr, 1, 1)
        a_s = ratt * scale                                                           # A[k][c] * pri / sqrt(dk)
        dqkv = torch.zeros(N, 3 * dp, dtype=torch.float32, device=dev)
        # dQ_i = sum_r (sum_e ds_e k_e) . (A s)           [out = in . F, F[k][c] = A[k][c] s]
        _, f_q = ops.pack(ratt, a_s, rpri)
        ops.spmm(plan, ds, K.data_ptr(), rte_k, f_q, ops.frags(f_q), dqkv, 0, 3 * dp, N)
        # transposed graph: dK_j = sum_r (sum_e ds_e q_i) . (A s)^T,  dV_j = sum_r (sum_e att_e dagg_i) . M^T
        plan_t = plan.transposed()
        ds_t = ops.to_sorted(plan_t, ops.to_edge_ids(plan, ds))
        att_tr = ops.to_sorted(plan_t, ops.to_edge_ids(plan, att))
        _, f_k = ops.pack(ratt, a_s.transpose(2, 3), rpri)
        ops.spmm(plan_t, ds_t, Q.data_ptr(), None, f_k, ops.frags(f_k), dqkv, dp, 3 * dp, N)
        _, f_v = ops.pack(ratt, rmsg.transpose(2, 3), rpri)
        ops.spmm(plan_t, att_tr, dagg.data_ptr(), None, f_v, ops.frags(f_v), dqkv, 2 * dp, 3 * dp, N)
        # relation parameters
        d_msg = ops.outer(plan, att, V, rte_v, dagg)[:, :Hr, :dk, :dk]               # d relation_msg[r,h,k,c]
        o_att = ops.outer(plan, ds, K, rte_k, Q)[:, :Hr, :dk, :dk]                   # sum ds_e k_e[k] q_i[c]
        d_ratt = o_att * scale
        d_rpri = (o_att * ratt).sum(dim=(2, 3)) / sqrt_dk

        # ---- temporal tables (use_RTE): their gradient is the same two spmm's grouped by (source type, dt) instead of by source
        d_rte_emb = d_rte_w = d_rte_b = None
        d_w_qkv_extra = None
        if use_rte:
            plan_r, tab = plan.rte_plan(T, R)
            ds_r = ops.to_sorted(plan_r, ops.to_edge_ids(plan, ds))
            att_r = ops.to_sorted(plan_r, ops.to_edge_ids(plan, att))
            d_tab = torch.zeros(tab, 2 * dp, dtype=torch.float32, device=dev)
            # sources of plan_r are the original TARGETS, shifted by `tab` ids: the row pointer is shifted back
            ops.spmm(plan_r, ds_r, Q.data_ptr() - 4 * tab * dp, None, f_k, ops.frags(f_k), d_tab, 0, 2 * dp, tab)
            ops.spmm(plan_r, att_r, dagg.data_ptr() - 4 * tab * dp, None, f_v, ops.frags(f_v), d_tab, dp, 2 * dp, tab)
            # tables = (emb W_rte^T + b_rte) W_{k|v}[t]^T: chain rule on [T*240, d] arrays with torch ops (tiny)
            with torch.enable_grad():
                e_, w_, b_ = (t.detach().requires_grad_(True) for t in (rte_emb, rte_w, rte_b))
                wkv = w_qkv.detach()[:, dp:3 * dp, :].requires_grad_(True)           # [T][2dp][din]
                lin = e_ @ w_.t() + b_                                               # [240, din]
                tabs = torch.einsum("pd,tod->tpo", lin, wkv).reshape(T * _lib.HGT_RTE_LEN, 2 * dp)
                ge, gw, gb, gkv = torch.autograd.grad(tabs, [e_, w_, b_, wkv], d_tab)
            d_rte_emb, d_rte_w, d_rte_b = ge, gw, gb
            d_w_qkv_extra = gkv

        # ---- projections backward (conv.py:96-97,103)
        d_w_qkv, d_b_qkv = ops.wgrad(dqkv, 3 * dp, x, din, rows.rows_all, rows.off_all, T, N, 3 * dp, din, with_colsum=True)
        if d_w_qkv_extra is not None:
            d_w_qkv[:, dp:3 * dp, :] += d_w_qkv_extra
        dx = None
        if ctx.needs_input_grad[3]:
            w_qkv_t = w_qkv.transpose(1, 2).contiguous()                             # [T][din][3dp]
            dx = torch.zeros(N, din, dtype=torch.float32, device=dev)
            ops.typed_linear(dqkv, 3 * dp, rows.rows_all, rows.off_all, T, N, 3 * dp, din, w_qkv_t, 0, din * 3 * dp, None, 0, 0, [dx], din)
            dx += dx_skip
        return (None, None, None, dx, d_w_qkv, d_b_qkv, d_w_a, d_b_a, d_ratt, d_msg.contiguous(), d_rpri, d_skip, d_lnw, d_lnb,
                d_rte_emb, d_rte_w, d_rte_b, d_mid_w, d_mid_b, d_out_w, d_out_b, d_oln_w, d_oln_b)


def hgt_conv_train(layer, plan, x, packed, drop_p):
    """Training-mode forward of `layer` (HGTConv or DenseHGTConv) through the autograd Function.  `packed` =
    layer._pack_parameters(grad=True); drop_p = dropout probability of conv.py:125 / 259,271 (0 in eval mode)."""
    if plan.NQ != plan.N:
        raise NotImplementedError("pyhgt_amd: the backward pass covers single-GPU graphs (n_q_rows == n_nodes)")
    lay = packed["lay"]
    if lay.dk_pad > 64:
        # hgt_relation_outer (the relation gradients) splits a head over at most 64 lanes x registers, and the MFMA form of
        # hgt_edge_spmm needs heads of at most 256 columns: say so HERE instead of failing inside loss.backward()
        raise NotImplementedError("pyhgt_amd: the backward pass supports heads of at most 64 (padded) columns; out_dim=%d / n_heads=%d "
                                  "gives %d.  Inference has no such limit (INTEGRATION.md, training limits)" % (layer.out_dim, layer.n_heads,
                                                                                                              lay.dk_pad))
    dense = "mid_w" in packed
    masks = None
    if drop_p > 0.0:
        keep = 1.0 - drop_p
        if keep <= 0.0:      # nn.Dropout(p=1) yields zeros (not 0/0)
            draw = lambda: torch.zeros((plan.N, layer.out_dim), dtype=torch.float32, device=x.device)
        else:
            draw = lambda: torch.bernoulli(torch.full((plan.N, layer.out_dim), keep, dtype=torch.float32, device=x.device)) / keep
        masks = (draw(), draw() if dense else None)          # DenseHGTConv drops twice (conv.py:259 and conv.py:271)
    return _HGTConvTrain.apply(layer, plan, masks, x, packed["w_qkv"], packed["b_qkv"], packed["w_a"], packed["b_a"], packed["ratt"],
                               packed["rmsg"], packed["rpri"], packed.get("skip"), packed.get("ln_w"), packed.get("ln_b"),
                               packed.get("rte_emb"), packed.get("rte_w"), packed.get("rte_b"), packed.get("mid_w"), packed.get("mid_b"),
                               packed.get("out_w"), packed.get("out_b"), packed.get("out_ln_w"), packed.get("out_ln_b"))


class TypedLinearFunction(torch.autograd.Function):
    """y[n] = x[n] W[type(n)]^T + b[type(n)] on the typed-linear kernels with its backward (typed weight gradient, column
    sums, optional input gradient): the input adapter of model.GNN (model.py:70-76) and the Linear layers of the heads."""

    @staticmethod
    def forward(ctx, plan_rows, n_groups, precision, x, w, b):
        # plan_rows = (rows ptr, off ptr, keep-alive object); w [G][n_out][k], b [G][n_out]
        lib = _lib.load()
        rows, off, _keep = plan_rows
        x = x.contiguous()
        n, k = x.shape
        n_out = w.shape[1]
        y = torch.zeros(n, n_out, dtype=torch.float32, device=x.device)
        wc, bc = w.contiguous(), (b.contiguous() if b is not None else None)
        if precision in ("bf16x3", "f16x3") and n_out % 4 == 0:
            nb = C.c_uint64()
            _chk("hgt_split_weights_bytes", lib.hgt_split_weights_bytes(n_groups, k, n_out, C.byref(nb)))
            tiles = torch.empty(int(nb.value), dtype=torch.uint8, device=x.device)
            _chk("hgt_split_weights", lib.hgt_split_weights(_p(wc), n_out * k, n_groups, k, n_out, _p(tiles), _st()))
            _chk("hgt_typed_linear_bf16x3", lib.hgt_typed_linear_bf16x3(_p(x), k, rows, off, n_groups, n, k, n_out, _p(tiles), _p(bc), n_out,
                                                                      _p(y), 0, 0, n_out, 0, 0, _st()))
            tiles.record_stream(torch.cuda.current_stream())
        else:
            _chk("hgt_typed_linear", lib.hgt_typed_linear(_p(x), k, rows, off, n_groups, n, k, n_out, _p(wc), n_out * k, _p(bc), n_out, _p(y), 0, 0,
                                                        n_out, 0, 0, 0, _st()))
        ctx.plan_rows, ctx.n_groups, ctx.has_bias, ctx.precision = plan_rows, n_groups, b is not None, precision
        ctx.save_for_backward(x, wc)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        rows, off, _keep = ctx.plan_rows
        x, w = ctx.saved_tensors
        G = ctx.n_groups
        gy = gy.contiguous().float()
        n, k = x.shape
        n_out = w.shape[1]
        dev = x.device
        dw = torch.zeros(G, n_out, k, dtype=torch.float32, device=dev)
        db = torch.zeros(G, n_out, dtype=torch.float32, device=dev) if ctx.has_bias else None
        if ctx.precision in ("bf16x3", "f16x3"):
            _chk("hgt_typed_wgrad_bf16x3", lib.hgt_typed_wgrad_bf16x3(_p(gy), n_out, _p(x), k, rows, off, G, n, n_out, k, _p(dw), n_out * k,
                                                                    _p(db), n_out, _st()))
        else:
            _chk("hgt_typed_wgrad", lib.hgt_typed_wgrad(_p(gy), n_out, _p(x), k, rows, off, G, n, n_out, k, _p(dw), n_out * k, _st()))
            if ctx.has_bias:
                _chk("hgt_typed_colsum", lib.hgt_typed_colsum(_p(gy), n_out, rows, off, G, n, n_out, _p(db), n_out, _st()))
        dx = None
        if ctx.needs_input_grad[3]:
            wt = w.transpose(1, 2).contiguous()
            dx = torch.zeros(n, k, dtype=torch.float32, device=dev)
            _chk("hgt_typed_linear", lib.hgt_typed_linear(_p(gy), n_out, rows, off, G, n, n_out, k, _p(wt), k * n_out, 0, 0, _p(dx), 0, 0, k, 0, 0,
                                                        0, _st()))
        return None, None, None, dx, dw, db
