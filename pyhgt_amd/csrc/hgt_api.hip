// C-ABI glue: error strings, layout helper and the whole-layer entry point that enqueues every
// kernel of one HGTConv.forward (conv.py:56-134, eval mode) on the caller's stream.
#include "hgt_common.h"

#ifndef HGT_FUSED_MIN_NODES
#define HGT_FUSED_MIN_NODES 16384         // targets from which the aggregation + update run as ONE kernel (hgt_edge_aggregate_update):
                                          // 10 edges per node, d = 256: 154 vs 125 us (item-parallel + update kernel) at 8k nodes, 181 vs
                                          // 191 at 16k, 229 vs 285 at 24k, 252 vs 365 at 32k, 474 vs 677 at 60k
#endif
#ifndef HGT_ITEM_AGG_MAX_NODES
#define HGT_ITEM_AGG_MAX_NODES 65536      // graphs below this CAN take the item-parallel aggregation (its scratch is part of the workspace)
#endif
#ifndef HGT_ITEM_AGG_DEFAULT_NODES
#define HGT_ITEM_AGG_DEFAULT_NODES 65536  // ... and below this they do by default (hgt_edge_agg_items.hip; measured with 10 edges per
                                          // node: 127 vs 173 us at 8k nodes, 191 vs 194 at 16k, 369 vs 383 at 32k, 708 vs 745 at 64k, d = 256 (unfused);
                                          // 330 vs 356 at 8k ... 1409 vs 1501 at 48k, d = 512)
#endif

namespace {

struct ConvWorkspace {
    uint64_t off_q, off_k, off_v, off_logits, off_agg, off_trans, off_att_t, off_msg_p, off_msg_f, off_att_f, off_hub;
    uint64_t off_rte_lin, off_rte_k, off_rte_v, off_rte_rows, off_rte_off, off_ws_qkv, off_ws_a, off_ws_rte, off_off2, off_pending, off_state, off_zitems, zitems_bytes, total;
};

static ConvWorkspace conv_workspace(int64_t N, int64_t NQ, int64_t E, int in_dim, int out_dim, int T, int R, int /*n_heads*/, int use_rte,
                                    const hgt_layout& lay, bool item_scratch = true, bool det_hubs = false) {
    const int H = lay.heads;     // layout heads (n_heads rounded up to a power of two)
    ConvWorkspace w;
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { uint64_t r = o; o = hgt_align_up(o + bytes, 256); return r; };
    const uint64_t dp = (uint64_t)lay.d_pad;
    w.off_q = take((uint64_t)NQ * dp * 4);
    w.off_k = take((uint64_t)N * dp * 4);
    w.off_v = take((uint64_t)N * dp * 4);
    w.off_logits = take((uint64_t)E * H * 4);
    w.off_agg = take((uint64_t)NQ * dp * 4);
    w.off_trans = take((uint64_t)NQ * out_dim * 4);
    w.off_att_t = take((uint64_t)R * H * lay.dk_pad * lay.dk_pad * 4);
    w.off_msg_p = take((uint64_t)R * H * lay.dk_pad * lay.dk_pad * 4);
    uint64_t fb = 0;
    hgt_relation_frag_bytes(R, H, lay.dk_pad, &fb);
    w.off_msg_f = take(fb);
    w.off_att_f = take(fb);
    uint64_t hb = 0;
    hgt_hub_workspace_bytes_ex(E, H, lay.dk_pad, R, det_hubs ? 1 : 0, &hb);
    w.off_hub = take(hb);
    if (use_rte) {
        w.off_rte_lin = take((uint64_t)HGT_RTE_LEN * in_dim * 4);
        w.off_rte_k = take((uint64_t)T * HGT_RTE_LEN * dp * 4);
        w.off_rte_v = take((uint64_t)T * HGT_RTE_LEN * dp * 4);
        w.off_rte_rows = take((uint64_t)T * HGT_RTE_LEN * 4);
        w.off_rte_off = take((uint64_t)(T + 1) * 4);
    } else {
        w.off_rte_lin = w.off_rte_k = w.off_rte_v = w.off_rte_rows = w.off_rte_off = 0;
    }
    // split-bf16 weight tiles (precision = 1); sized unconditionally, they are small
    uint64_t b = 0;
    hgt_split_weights_bytes(T, in_dim, 3 * lay.d_pad, &b);
    w.off_ws_qkv = take(b);
    // shared by the a_linear / Q-only / temporal K|V splits (used one after the other on the stream)
    hgt_split_weights_bytes(T, in_dim > lay.d_pad ? in_dim : lay.d_pad, 2 * lay.d_pad > out_dim ? 2 * lay.d_pad : out_dim, &b);
    {   // ... and by the shared dense layer of DenseHGTConv (one group; its out_linear has K = 2*out_dim, which the typed
        // shapes above do not cover when T == 1 and d_pad <= 128 -- found by tools/fuzz_parity.py)
        uint64_t b2 = 0;
        hgt_split_weights_bytes(1, 2 * out_dim, out_dim, &b2);
        if (b2 > b) b = b2;
        hgt_split_weights_bytes(1, out_dim, 2 * out_dim, &b2);
        if (b2 > b) b = b2;
    }
    w.off_ws_a = take(b);
    hgt_split_weights_bytes(1, in_dim, in_dim, &b);
    w.off_ws_rte = take(use_rte ? b : 0);
    w.off_off2 = take(256);
    w.off_pending = take((uint64_t)(NQ / 64 + 1) * 4);
    w.off_state = take((uint64_t)NQ * H * 2 * 4);   // softmax state carried between relation slices (stage 4)
    // scratch of the item-parallel aggregation (hgt_edge_aggregate_items): only for graphs in the latency regime
    w.zitems_bytes = 0;
    if (item_scratch && N < HGT_ITEM_AGG_MAX_NODES) {
        uint64_t zb = 0;
        hgt_edge_aggregate_items_bytes(E, H, lay.dk_pad, &zb);
        if (zb <= ((uint64_t)1 << 30)) w.zitems_bytes = zb;
    }
    w.off_zitems = take(w.zitems_bytes);
    w.total = o;
    return w;
}

struct PreparedLayout {
    uint64_t off_att_t, off_msg_p, off_msg_f, off_att_f, off_ws_qkv, off_ws_upd, off_rte_k, off_rte_v, total;
};

static PreparedLayout prepared_layout(int in_dim, int out_dim, int T, int R, int /*n_heads*/, int use_rte, const hgt_layout& lay) {
    const int H = lay.heads;
    PreparedLayout p;
    uint64_t o = 0, b = 0;
    auto take = [&](uint64_t bytes) { uint64_t r = o; o = hgt_align_up(o + bytes, 256); return r; };
    p.off_att_t = take((uint64_t)R * H * lay.dk_pad * lay.dk_pad * 4);
    p.off_msg_p = take((uint64_t)R * H * lay.dk_pad * lay.dk_pad * 4);
    hgt_relation_frag_bytes(R, H, lay.dk_pad, &b);
    p.off_msg_f = take(b);
    p.off_att_f = take(b);
    hgt_split_weights_bytes(T, in_dim, 3 * lay.d_pad, &b);
    p.off_ws_qkv = take(b);
    hgt_split_weights_bytes(T, lay.d_pad, out_dim, &b);
    p.off_ws_upd = take(b);
    p.off_rte_k = take(use_rte ? (uint64_t)T * HGT_RTE_LEN * lay.d_pad * 4 : 0);
    p.off_rte_v = take(use_rte ? (uint64_t)T * HGT_RTE_LEN * lay.d_pad * 4 : 0);
    p.total = o;
    return p;
}

// off2 = {0, off_q[T]}: all rows of a valid type as ONE group (the shared dense layer of DenseHGTConv)
__global__ void k_single_group(const int32_t* __restrict__ off_q, int T, int32_t* __restrict__ off2) {
    if (threadIdx.x == 0) { off2[0] = 0; off2[1] = off_q[T]; }
}

// rows[i] = i % 240 for i < T*240 ; off[g] = g*240
__global__ void k_rte_row_lists(int T, int32_t* __restrict__ rows, int32_t* __restrict__ off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T * HGT_RTE_LEN) rows[i] = i % HGT_RTE_LEN;
    if (i <= T) off[i] = i * HGT_RTE_LEN;
}

}  // namespace

extern "C" const char* hgt_strerror(int code) {
    switch (code) {
        case HGT_OK: return "ok";
        case HGT_ERR_INVALID_ARG: return "invalid argument";
        case HGT_ERR_UNSUPPORTED: return "unsupported shape (need d % n_heads == 0, n_heads <= 16, padded row width <= 1024)";
        case HGT_ERR_WORKSPACE: return "workspace too small";
        case HGT_ERR_TOO_LARGE: return "problem exceeds 32-bit plan indices";
        case HGT_ERR_LAUNCH: return "HIP launch/runtime error";
        default: return "unknown error";
    }
}

extern "C" int hgt_abi_version(void) { return HGT_ABI_VERSION; }
extern "C" int hgt_build_features(void) {
#ifdef HGT_LAB_KERNELS
    return HGT_FEATURE_LAB_KERNELS;
#else
    return 0;
#endif
}

extern "C" int hgt_layout_for(int32_t d_out, int32_t n_heads, hgt_layout* out) {
    if (!out) return HGT_ERR_INVALID_ARG;
    int rc = hgt_layout_compute(d_out, n_heads, out);
    if (rc != HGT_OK) return rc;
    // rows of up to 1024 padded columns (vec 16: n_hid 768 / 1024; the edge kernels split them into head groups of <= 256 or 512
    // columns per wavefront), at least 4 lanes per head (n_heads <= 16)
    if (out->vec > 16 || 64 / out->heads < 4) return HGT_ERR_UNSUPPORTED;
    return HGT_OK;
}

extern "C" int hgt_conv_workspace_bytes(int64_t n_nodes, int64_t n_edges, int32_t in_dim, int32_t out_dim, int32_t n_types,
                                        int32_t n_relations, int32_t n_heads, int32_t use_rte, uint64_t* out) {
    if (!out || n_nodes < 0 || n_edges < 0 || in_dim <= 0) return HGT_ERR_INVALID_ARG;
    hgt_layout lay;
    int rc = hgt_layout_for(out_dim, n_heads, &lay);
    if (rc != HGT_OK) return rc;
    *out = conv_workspace(n_nodes, n_nodes, n_edges, in_dim, out_dim, n_types, n_relations, n_heads, use_rte, lay).total;
    return HGT_OK;
}

// ABI 6: the same without the scratch of the item-parallel aggregation (up to 1 GiB on graphs below 65536 nodes) for callers that
// know the call cannot take it: exact fp32 precision, HGT_FLAG_NO_ITEM_AGGREGATE, staged multi-GPU calls.  hgt_conv_forward
// accepts either size (a workspace without the scratch simply rules the item-parallel kernel out).
extern "C" int hgt_conv_workspace_bytes_ex(int64_t n_nodes, int64_t n_edges, int32_t in_dim, int32_t out_dim, int32_t n_types,
                                           int32_t n_relations, int32_t n_heads, int32_t use_rte, int32_t options, uint64_t* out) {
    if (!out || n_nodes < 0 || n_edges < 0 || in_dim <= 0) return HGT_ERR_INVALID_ARG;
    hgt_layout lay;
    int rc = hgt_layout_for(out_dim, n_heads, &lay);
    if (rc != HGT_OK) return rc;
    *out = conv_workspace(n_nodes, n_nodes, n_edges, in_dim, out_dim, n_types, n_relations, n_heads, use_rte, lay, (options & 1) != 0, (options & 2) != 0).total;
    return HGT_OK;
}

extern "C" int hgt_conv_prepared_bytes(int32_t in_dim, int32_t out_dim, int32_t n_types, int32_t n_relations, int32_t n_heads,
                                       int32_t use_rte, uint64_t* out) {
    if (!out || in_dim <= 0 || n_types <= 0 || n_relations <= 0) return HGT_ERR_INVALID_ARG;
    hgt_layout lay;
    int rc = hgt_layout_for(out_dim, n_heads, &lay);
    if (rc != HGT_OK) return rc;
    *out = prepared_layout(in_dim, out_dim, n_types, n_relations, n_heads, use_rte, lay).total;
    return HGT_OK;
}

extern "C" int hgt_conv_forward(const hgt_conv_args* a, void* stream_) {
    if (!a) return HGT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t N = a->n_nodes, E = a->n_edges;
    const int64_t NQ = (a->n_q_rows > 0 && a->n_q_rows <= N) ? a->n_q_rows : N;
    const int T = a->n_types, R = a->n_relations, Hreal = a->n_heads, din = a->in_dim, dout = a->out_dim;
    if (!a->x || !a->node_type || !a->plan || !a->w_qkv || !a->b_qkv || !a->w_a || !a->b_a || !a->relation_att ||
        !a->relation_msg || !a->relation_pri || (!a->skip && a->update_mode == 0) || !a->workspace || !a->out)
        return HGT_ERR_INVALID_ARG;
    if (din != dout) return HGT_ERR_INVALID_ARG;   // the skip connection of conv.py:131 needs in_dim == out_dim
    if (a->use_norm && (!a->ln_w || !a->ln_b)) return HGT_ERR_INVALID_ARG;
    if (a->use_rte && (!a->rte_emb || !a->rte_w || !a->rte_b)) return HGT_ERR_INVALID_ARG;
    if (a->want_att && E > 0 && !a->att_out) return HGT_ERR_INVALID_ARG;
    const bool dense = (a->update_mode == 1);
    if (a->update_mode != 0 && a->update_mode != 1) return HGT_ERR_INVALID_ARG;
    if (a->precision < 0 || a->precision > 2) return HGT_ERR_INVALID_ARG;
    // precision 2 (fp16 hi/lo split, include/hgt_hip.h): whole-layer calls only -- the staged multi-GPU calls share one prepared
    // image between a sliced edge phase (whose state does not carry the fp16 row scales) and the rest, and stay on precision 1
    const bool f16 = (a->precision == 2);
    const int fmode = (f16 ? 1 : 0) | ((a->flags & HGT_FLAG_NO_COOP_EDGE) ? 2 : 0) | ((a->flags & HGT_FLAG_COOP_EDGE_ALWAYS) ? 4 : 0);      // `frag_f16` of the item kernels
    if (f16 && a->stage != 0) return HGT_ERR_UNSUPPORTED;
    if (dense && (!a->mid_w || !a->mid_b || !a->out_w || !a->out_b || !a->out_ln_w || !a->out_ln_b)) return HGT_ERR_INVALID_ARG;
    hgt_layout lay;
    int rc = hgt_layout_for(dout, Hreal, &lay);
    if (rc != HGT_OK) return rc;
    const int H = lay.heads;     // the kernels run with the layout's head count (extra heads are all-zero)
    const int dp = lay.d_pad;
    const bool det_hubs = (a->flags & HGT_FLAG_DETERMINISTIC_HUBS) != 0;       // needs the larger hub region: options bit 1
    ConvWorkspace w = conv_workspace(N, N, E, din, dout, T, R, H, a->use_rte, lay, true, det_hubs);   // sized for NQ == N (upper bound)
    if (a->workspace_bytes < w.total) {      // a workspace sized without the item-aggregation scratch (hgt_conv_workspace_bytes_ex)
        w = conv_workspace(N, N, E, din, dout, T, R, H, a->use_rte, lay, false, det_hubs);
        if (a->workspace_bytes < w.total) return HGT_ERR_WORKSPACE;
    }
    if (N == 0) return HGT_OK;
    char* wb = (char*)a->workspace;
    float* Q = (float*)(wb + w.off_q);
    float* K = (float*)(wb + w.off_k);
    float* V = (float*)(wb + w.off_v);
    float* logits = (float*)(wb + w.off_logits);
    float* agg = (float*)(wb + w.off_agg);
    float* trans = (float*)(wb + w.off_trans);
    float* att_t = (float*)(wb + w.off_att_t);
    float* msg_p = (float*)(wb + w.off_msg_p);
    float* rte_k = nullptr;
    float* rte_v = nullptr;
    // weight-only preprocessing: in the caller's `prepared` buffer (kept across calls) or in the workspace (every call)
    char* pb = (char*)a->prepared;
    PreparedLayout pl = prepared_layout(din, dout, T, R, H, a->use_rte, lay);
    if (pb && a->prepared_bytes < pl.total) return HGT_ERR_WORKSPACE;
    const bool fresh = !(pb && a->prepared_valid);          // derive the weight images in this call
    void* hub_ws = (a->plan_no_hubs & 1) ? nullptr : (void*)(wb + w.off_hub);
    const bool no_unknown_rows = (a->plan_no_hubs & 2) != 0;     // the caller knows that every target row has a valid type
    void* msg_f = wb + w.off_msg_f;
    void* att_f = wb + w.off_att_f;
    if (pb) {
        att_t = (float*)(pb + pl.off_att_t);
        msg_p = (float*)(pb + pl.off_msg_p);
        msg_f = pb + pl.off_msg_f;
        att_f = pb + pl.off_att_f;
    }
    // relation transforms of the aggregation: matrix cores (split-bf16 x3) with the split precision, exact fp32 mat-vecs otherwise
    uint64_t frag_bytes = 0;
    hgt_relation_frag_bytes(R, H, lay.dk_pad, &frag_bytes);
    const bool have_frags = (a->precision >= 1) && frag_bytes > 0;
    void* const msg_f_buf = msg_f;      // where the fragment images live (the kernel arguments below may be nulled by the flags)
    void* const att_f_buf = att_f;
    const bool mfma_agg = have_frags && !(a->flags & HGT_FLAG_VALU_AGGREGATE);
    if (!mfma_agg) msg_f = nullptr;
    // logits: the target-side transforms on the matrix cores where the vector-ALU kernel is instruction-bound (d_k >= 64)
    const bool mfma_logits = have_frags && !(a->flags & HGT_FLAG_VALU_LOGITS) && a->stage != 4 &&
                             (lay.dk_pad >= 64 || (a->flags & HGT_FLAG_MFMA_LOGITS));

    auto mark = [&](int i) {
        if (a->phase_events && a->phase_events[i]) (void)hipEventRecord((hipEvent_t)a->phase_events[i], stream);
    };
    const int stage = a->stage;
    if (stage < 0 || stage > 5) return HGT_ERR_INVALID_ARG;
    // stage 5: edge phase + fused update of ONE target block (multi-GPU path: the block's in-edges only reference source rows of the
    // halo chunks that have arrived); every block is the single-GPU kernel pair on a range of destination tiles
    const bool blocked = (stage == 5);
    if (blocked) {
        if (a->q_begin < 0 || a->q_end < a->q_begin || a->q_end > NQ || (a->q_begin % HGT_TD) != 0 || a->item_begin < 0 ||
            a->item_end < a->item_begin)
            return HGT_ERR_INVALID_ARG;
        if (!mfma_agg || dense || a->want_att) return HGT_ERR_UNSUPPORTED;
    }
    // stage 4: the edge phase over ONE slice of the relation buckets (multi-GPU path: relation id = source bucket * R' + relation)
    const bool sliced = (stage == 4);
    int sl_lo = 0, sl_hi = R + 1, sl_more = 0;
    if (sliced) {
        const int S = a->slice_count, si = a->slice_index;
        if (S <= 0 || R % S != 0 || si < 0 || si >= S) return HGT_ERR_INVALID_ARG;
        if (!mfma_agg || dense) return HGT_ERR_UNSUPPORTED;      // the slice merge lives in the matrix-core aggregation kernel
        sl_lo = si * (R / S);
        sl_hi = (si + 1) * (R / S) + (si == S - 1 ? 1 : 0);       // the last slice also takes the bucket of unclaimed edges
        sl_more = (si < S - 1);
    }
    if (stage == 2 && (a->proj_n < 0 || (a->proj_n > 0 && (!a->proj_rows || !a->proj_off)))) return HGT_ERR_INVALID_ARG;
    if (stage == 0 || stage == 1) mark(0);
    hgt_plan_rows pr;
    rc = hgt_plan_row_lists(a->plan, N, E, T, R, &pr);
    if (rc != HGT_OK) return rc;

    // (1) relation matrices: fold pri/sqrt(dk), transpose att, zero-pad heads (conv.py:98-99,104).  BOTH fragment images are made
    // whenever the split precision has them, whatever kernels this call's flags select: a `prepared` buffer outlives the call and a
    // later call with other flags trusts it (round-3 advisor finding: a flag change on a live layer read an image that was never written)
    if ((stage == 0 || stage == 1) && fresh) {
        rc = hgt_relation_pack(a->relation_att, a->relation_msg, a->relation_pri, R, Hreal, H, lay.d_k, lay.dk_pad, att_t, msg_p, stream);
        if (rc != HGT_OK) return rc;
        if (have_frags) {
            rc = f16 ? hgt_relation_frag_pack_f16(msg_p, R, H, lay.dk_pad, msg_f_buf, stream)
                     : hgt_relation_frag_pack(msg_p, R, H, lay.dk_pad, msg_f_buf, stream);
            if (rc != HGT_OK) return rc;
            rc = f16 ? hgt_relation_frag_pack_f16(att_t, R, H, lay.dk_pad, att_f_buf, stream)
                     : hgt_relation_frag_pack(att_t, R, H, lay.dk_pad, att_f_buf, stream);
            if (rc != HGT_OK) return rc;
        }
    }

    // typed linear dispatch: exact fp32 MFMA, or split-bf16 x3 with weights split+tiled into the workspace
    const bool split = (a->precision >= 1);
    auto split_weights = f16 ? hgt_split_weights_f16 : hgt_split_weights;
    auto linear = [&](const float* xin, int64_t ldx, const int32_t* rws, const int32_t* goff, int ng, int64_t nrows, int kk, int nout,
                      const float* Wp, int64_t wgs, const float* bp, int64_t bgs, float* o0, float* o1, float* o2, int bcols,
                      int by_pos, void* wsplit, int prologue = 0, bool tiles_ready = false) -> int {
        if (!split)
            return hgt_typed_linear(xin, ldx, rws, goff, ng, nrows, kk, nout, Wp, wgs, bp, bgs, o0, o1, o2, bcols, by_pos, prologue, 0, stream);
        if (((nout | bcols) & 3) != 0)   // the split kernel stores 16 B per lane: odd widths take the exact fp32 kernel
            return hgt_typed_linear(xin, ldx, rws, goff, ng, nrows, kk, nout, Wp, wgs, bp, bgs, o0, o1, o2, bcols, by_pos, prologue, 0, stream);
        if (!tiles_ready) {
            int r2 = split_weights(Wp, wgs, ng, kk, nout, wsplit, stream);
            if (r2 != HGT_OK) return r2;
        }
        // (kernel-selection bits of the split linears: HGT_FLAG_XS_GEMM_ALWAYS / _NEVER -- tests and A/B runs)
        const int sel = ((a->flags & HGT_FLAG_XS_GEMM_NEVER) ? HGT_LINEAR_NO_XS : ((a->flags & HGT_FLAG_XS_GEMM_ALWAYS) ? HGT_LINEAR_FORCE_XS : 0)) |
                        ((a->flags & HGT_FLAG_NO_TILE_GEMM) ? HGT_LINEAR_NO_TILE : 0);
        return f16 ? hgt_typed_linear_f16x3(xin, ldx, rws, goff, ng, nrows, kk, nout, wsplit, bp, bgs, o0, o1, o2, bcols, by_pos, prologue | sel, stream)
                   : hgt_typed_linear_bf16x3(xin, ldx, rws, goff, ng, nrows, kk, nout, wsplit, bp, bgs, o0, o1, o2, bcols, by_pos, prologue | sel,
                                             stream);
    };
    void* ws_qkv_scratch = wb + w.off_ws_qkv;                       // K|V-only tiles of the halo branch
    void* ws_qkv = pb ? (void*)(pb + pl.off_ws_qkv) : ws_qkv_scratch; // tiles of the full [Q|K|V] weight
    void* ws_a = wb + w.off_ws_a;                                     // scratch tiles (Q-only, K|V, temporal, dense layer)
    void* ws_upd = pb ? (void*)(pb + pl.off_ws_upd) : ws_a;           // tiles of W_a

    // (2) typed projections once per NODE (conv.py:96-97,103 did them per edge)
    const int64_t wstride = (int64_t)3 * dp * din;
    if (stage == 2) {   // K|V of one received chunk of halo rows (the K|V split tiles are re-made each time: tiny)
        if (a->proj_n == 0) return HGT_OK;
        if (a->proj_c24) {   // straight off the wire buffer (24-bit rows): no expansion pass, 3/4 of the bytes read
            if (!split) return HGT_ERR_UNSUPPORTED;
            const int64_t ldw = 3 * (int64_t)(din / 4);      // dwords per wire row
            const float* xw = reinterpret_cast<const float*>(a->proj_c24) - a->proj_c24_row0 * ldw;   // indexed by the LOCAL row id
            return linear(xw, ldw, a->proj_rows, a->proj_off, T, a->proj_n, din, 2 * dp, a->w_qkv + (int64_t)dp * din, wstride,
                          a->b_qkv + dp, 3 * dp, K, V, nullptr, dp, 0, ws_a, 2);
        }
        return linear(a->x, din, a->proj_rows, a->proj_off, T, a->proj_n, din, 2 * dp, a->w_qkv + (int64_t)dp * din, wstride, a->b_qkv + dp,
                      3 * dp, K, V, nullptr, dp, 0, ws_a);
    }
    if (stage == 3 || stage == 4 || stage == 5) goto edge_phase;
    if (stage == 1) {   // own rows only: one fused Q|K|V launch, exactly like the single-GPU layer
        rc = linear(a->x, din, pr.rows_q, pr.off_q, T, NQ, din, 3 * dp, a->w_qkv, wstride, a->b_qkv, 3 * dp, Q, K, V, dp, 0, ws_qkv, 0, !fresh);
        if (rc != HGT_OK) return rc;
    } else if (NQ == N) {
        rc = linear(a->x, din, pr.rows_all, pr.off_all, T, N, din, 3 * dp, a->w_qkv, wstride, a->b_qkv, 3 * dp, Q, K, V, dp, 0, ws_qkv, 0, !fresh);
        if (rc != HGT_OK) return rc;
    } else {
        // halo rows (>= NQ) only need K and V.  The split tiles of the full [Q|K|V] weight serve both launches:
        // Q = columns [0,dp) -> its own split; K|V = columns [dp,3dp)
        rc = linear(a->x, din, pr.rows_q, pr.off_q, T, NQ, din, dp, a->w_qkv, wstride, a->b_qkv, 3 * dp, Q, nullptr, nullptr, dp, 0, ws_a);
        if (rc != HGT_OK) return rc;
        rc = linear(a->x, din, pr.rows_all, pr.off_all, T, N, din, 2 * dp, a->w_qkv + (int64_t)dp * din, wstride, a->b_qkv + dp, 3 * dp,
                    K, V, nullptr, dp, 0, ws_qkv_scratch);
        if (rc != HGT_OK) return rc;
        if (pb && fresh && split) {   // keep the prepared buffer complete: a later call may be a whole-graph or staged one
            rc = split_weights(a->w_qkv, wstride, T, din, 3 * dp, ws_qkv, stream);
            if (rc != HGT_OK) return rc;
        }
    }

    // (3) temporal tables: rte_k[t][p] = (emb[p] W_rte^T + b_rte) W_k[t]^T  (conv.py:91-92,298-299 hoisted off the edges)
    if (a->use_rte && fresh) {
        float* rte_lin = (float*)(wb + w.off_rte_lin);
        rte_k = pb ? (float*)(pb + pl.off_rte_k) : (float*)(wb + w.off_rte_k);
        rte_v = pb ? (float*)(pb + pl.off_rte_v) : (float*)(wb + w.off_rte_v);
        int32_t* rrows = (int32_t*)(wb + w.off_rte_rows);
        int32_t* roff = (int32_t*)(wb + w.off_rte_off);
        const int nthr = T * HGT_RTE_LEN;
        k_rte_row_lists<<<(nthr + 255) / 256, 256, 0, stream>>>(T, rrows, roff);
        rc = linear(a->rte_emb, din, rrows, roff, 1, HGT_RTE_LEN, din, din, a->rte_w, 0, a->rte_b, 0, rte_lin, nullptr, nullptr, din, 1,
                    wb + w.off_ws_rte);
        if (rc != HGT_OK) return rc;
        // K|V part of the weight, split again for this 2-output launch (tiny)
        rc = linear(rte_lin, din, rrows, roff, T, (int64_t)T * HGT_RTE_LEN, din, 2 * dp, a->w_qkv + (int64_t)dp * din, wstride, nullptr, 0,
                    rte_k, rte_v, nullptr, dp, 1, ws_a);
        if (rc != HGT_OK) return rc;
    }

    if (stage == 1) {
        // staged forwards whose edge phase runs per target block (stage 5) find the image of W_a in the prepared buffer
        if (pb && fresh && split && !dense && dp <= 256 && dout <= dp && (dout & 3) == 0) {
            rc = split_weights(a->w_a, (int64_t)dout * dp, T, dp, dout, ws_upd, stream);
            if (rc != HGT_OK) return rc;
        }
        return HGT_OK;
    }
edge_phase:
    if (a->use_rte) {   // (also for stage 3 and for calls that trust the prepared tables)
        rte_k = pb ? (float*)(pb + pl.off_rte_k) : (float*)(wb + w.off_rte_k);
        rte_v = pb ? (float*)(pb + pl.off_rte_v) : (float*)(wb + w.off_rte_v);
    }
    mark(1);
    if (blocked) {
        const bool can_fuse = split && dp <= 256 && dout <= dp && (dout & 3) == 0 && (din & 3) == 0;
        if (!can_fuse) return HGT_ERR_UNSUPPORTED;
        if (a->q_begin == a->q_end) return HGT_OK;
        if (E > 0 && a->item_end > a->item_begin) {
            rc = hgt_edge_logits_range(a->plan, N, E, T, R, H, lay.dk_pad, Q, K, rte_k, att_t, mfma_logits ? att_f : nullptr, fmode, logits,
                                       a->item_begin, a->item_end, stream);
            if (rc != HGT_OK) return rc;
        }
        mark(2);
        mark(3);
        // (the image of W_a was written by stage 1 when the prepared buffer is fresh -- see below -- or is made here)
        if (!pb) {
            rc = split_weights(a->w_a, (int64_t)dout * dp, T, dp, dout, ws_upd, stream);
            if (rc != HGT_OK) return rc;
        }
        if (a->q_end < 0) return HGT_ERR_INVALID_ARG;
        rc = hgt_edge_aggregate_update_sel(a->plan, N, E, T, R, H, lay.dk_pad, logits, V, rte_v, msg_p, msg_f, agg, NQ, hub_ws,
                                           (int32_t*)(wb + w.off_pending), a->node_type, ws_upd, a->b_a, a->x, din, a->skip, a->ln_w,
                                           a->ln_b, a->use_norm, dout, a->out, stream, a->q_begin, a->q_end, 0, det_hubs ? 1 : 0,
                                           (a->flags & HGT_FLAG_RING_AGGREGATE) ? 1 : 0);
        mark(4);
        mark(5);
        mark(6);
        return rc;
    }
    // (5') which aggregation form runs is decided BEFORE the logits: the single-pass form of the latency regime computes them itself
    // (graphs below HGT_FUSED_MIN_NODES targets take the unfused kernels: the latency regime, see items_agg)
    const bool fuse_all = split && !dense && dp <= 256 && dout <= dp && (dout & 3) == 0 && (din & 3) == 0 &&
                          (NQ >= HGT_FUSED_MIN_NODES || (a->flags & HGT_FLAG_FUSED_ANY_SIZE)) &&
                          !(a->flags & HGT_FLAG_NO_FUSED_UPDATE) && !sliced &&
                          !(f16 && !mfma_agg);   // the vector-ALU kernel's fused epilogue only reads the bf16 image of W_a
    // latency regime: the item-parallel form (hgt_edge_agg_items.hip), where a sub-tile wavefront's chain of edge batches and
    // relation ends is the kernel time (c3: 80 -> 66 us per layer, c5: 195 -> 147 us); flags force / forbid it
    const bool items_agg = mfma_agg && !sliced && w.zitems_bytes > 0 && NQ < HGT_ITEM_AGG_MAX_NODES && R < 64 &&
                           !(a->flags & HGT_FLAG_NO_ITEM_AGGREGATE) &&
                           (NQ < HGT_ITEM_AGG_DEFAULT_NODES || (a->flags & HGT_FLAG_ITEM_AGGREGATE));
    // ... and, on request (HGT_FLAG_SINGLE_PASS) and when nobody asks for the attention weights, logits + runs in ONE walk
    // (lab/hgt_edge_single_pass.hip, LAB builds only): one kernel and the [E][H] logits array less; measured equal at c3, 5 % slower at c5
    bool agg_done = false;
    if (items_agg && !fuse_all && !a->want_att && have_frags && E > 0 && (a->flags & HGT_FLAG_SINGLE_PASS)) {
        rc = hgt_edge_single_pass_items(a->plan, N, E, T, R, H, lay.dk_pad, Q, K, V, rte_k, rte_v, att_f_buf, msg_f, f16 ? 1 : 0, agg, NQ,
                                        dense ? 0 : 1, wb + w.off_zitems, w.zitems_bytes, stream);
        if (rc == HGT_OK) agg_done = true;
        else if (rc != HGT_ERR_UNSUPPORTED) return rc;      // (a layout it is not instantiated for: the two-kernel form below)
    }
    // (4) edge phase: logits, then softmax fused into the aggregation (online, per target sub-tile)
    if (E > 0 && !agg_done) {
        rc = sliced ? hgt_edge_logits_slice(a->plan, N, E, T, R, H, lay.dk_pad, Q, K, rte_k, att_t, logits, sl_lo, sl_hi, stream)
             : mfma_logits ? hgt_edge_logits_mfma(a->plan, N, E, T, R, H, lay.dk_pad, Q, K, rte_k, att_t, att_f, fmode, logits, stream)
                           : hgt_edge_logits(a->plan, N, E, T, R, H, lay.dk_pad, Q, K, rte_k, att_t, logits, stream);
        if (rc != HGT_OK) return rc;
    }
    mark(2);
    mark(3);
    // (5) aggregation + update.  Preferred form: one kernel that never writes agg (hgt_edge_aggregate_update).
    if (fuse_all) {
        if (fresh || !pb) {
            rc = split_weights(a->w_a, (int64_t)dout * dp, T, dp, dout, ws_upd, stream);
            if (rc != HGT_OK) return rc;
        }
        if (det_hubs && hub_ws && !msg_f)
            rc = HGT_ERR_UNSUPPORTED;      // (vector-ALU aggregation: the unfused kernels below carry the deterministic hub mode)
        else
            rc = hgt_edge_aggregate_update_sel(a->plan, N, E, T, R, H, lay.dk_pad, logits, V, rte_v, msg_p, msg_f, agg, NQ, hub_ws,
                                               (int32_t*)(wb + w.off_pending), a->node_type, ws_upd, a->b_a, a->x, din, a->skip, a->ln_w,
                                               a->ln_b, a->use_norm, dout, a->out, stream, 0, (det_hubs && hub_ws) ? NQ : -1, f16 ? 1 : 0,
                                               (det_hubs && hub_ws) ? 1 : 0, (a->flags & HGT_FLAG_RING_AGGREGATE) ? 1 : 0);
        if (rc == HGT_OK) {
            if (a->want_att && E > 0) {
                rc = hgt_edge_softmax(a->plan, N, E, T, R, H, logits, stream);
                if (rc != HGT_OK) return rc;
                rc = hgt_att_export(a->plan, N, E, T, R, H, logits, a->att_out, Hreal, stream);
                if (rc != HGT_OK) return rc;
            }
            mark(4);
            mark(5);
            mark(6);
            return HGT_OK;
        }
        if (rc != HGT_ERR_UNSUPPORTED) return rc;   // unsupported layout (head-group split): the unfused kernels below
    }
    // runs for E == 0 too: it writes the zero rows of isolated targets; HGTConv stores gelu(agg) (conv.py:119), DenseHGTConv agg
    // latency regime: the item-parallel form (hgt_edge_agg_items.hip), where a sub-tile wavefront's chain of edge batches and
    // relation ends is the kernel time (c3: 80 -> 66 us per layer, c5: 195 -> 147 us); flags force / forbid it
    rc = agg_done ? HGT_OK : HGT_ERR_UNSUPPORTED;
    // sampled batches (round 6): the merge pass of the item-parallel aggregation IS the node update (k_merge_update) -- two of the
    // layer's five dependent kernels become one and `agg` is never written
    if (items_agg && !agg_done && !dense && !(a->flags & HGT_FLAG_NO_MERGE_UPDATE) && dp <= 512 && dout <= 512 && dout <= dp &&
        (dout & 3) == 0 && (din & 3) == 0) {
        if (fresh || !pb) {
            rc = split_weights(a->w_a, (int64_t)dout * dp, T, dp, dout, ws_upd, stream);
            if (rc != HGT_OK) return rc;
        }
        rc = hgt_edge_aggregate_items_update(a->plan, N, E, T, R, H, lay.dk_pad, logits, V, rte_v, msg_f, fmode, NQ, wb + w.off_zitems,
                                             w.zitems_bytes, pr.rows_q, pr.off_q, T, ws_upd, a->b_a, a->x, din, a->skip, a->ln_w, a->ln_b,
                                             a->use_norm, dout, a->out, stream);
        if (rc == HGT_OK) {
            if (a->want_att && E > 0) {
                rc = hgt_edge_softmax(a->plan, N, E, T, R, H, logits, stream);
                if (rc != HGT_OK) return rc;
                rc = hgt_att_export(a->plan, N, E, T, R, H, logits, a->att_out, Hreal, stream);
                if (rc != HGT_OK) return rc;
            }
            mark(4);
            mark(5);
            if (!no_unknown_rows) rc = hgt_zero_rows(pr.rows_q, pr.off_q + T, dout, a->out, stream);   // nodes of unknown type -> 0 (conv.py:120)
            mark(6);
            return rc;
        }
        if (rc != HGT_ERR_UNSUPPORTED) return rc;
    }
    if (items_agg && !agg_done)
        rc = hgt_edge_aggregate_items(a->plan, N, E, T, R, H, lay.dk_pad, logits, V, rte_v, msg_f, fmode, agg, NQ, dense ? 0 : 1,
                                      wb + w.off_zitems, w.zitems_bytes, stream);
    if (rc != HGT_ERR_UNSUPPORTED) {
        // (done, or a real error)
    } else if (sliced) {
        rc = hgt_edge_aggregate_slice(a->plan, N, E, T, R, H, lay.dk_pad, logits, V, rte_v, msg_p, msg_f, agg, NQ, 1, hub_ws, sl_lo, sl_hi,
                                      (float*)(wb + w.off_state), sl_lo > 0, sl_more, stream);
        if (rc != HGT_OK || sl_more) return rc;      // state + un-normalised rows stay in the workspace for the next slice
    } else {
        rc = hgt_edge_aggregate_ex(a->plan, N, E, T, R, H, lay.dk_pad, logits, V, rte_v, msg_p, msg_f, (f16 && msg_f) ? 1 : 0, agg, NQ,
                                   dense ? 0 : 1, hub_ws, det_hubs ? 1 : 0, stream);
    }
    if (rc != HGT_OK) return rc;
    if (a->want_att && E > 0) {   // self.att (conv.py:108): normalise the logits in place and un-sort them
        rc = hgt_edge_softmax(a->plan, N, E, T, R, H, logits, stream);
        if (rc != HGT_OK) return rc;
        rc = hgt_att_export(a->plan, N, E, T, R, H, logits, a->att_out, Hreal, stream);
        if (rc != HGT_OK) return rc;
    }
    mark(4);
    // (6) update: a_linear(gelu(agg)) -> gated skip -> LayerNorm (conv.py:119-133)
    // (257..512 columns, e.g. n_hid 400 / 512: k_typed_linear_update_wide, round 5)
    const bool fuse_update = !dense && split && (dout <= 256 || (dout <= 512 && dp <= 512)) && (dout & 3) == 0 && (din & 3) == 0;
    if (dense) {
        // DenseHGTConv.update (conv.py:250-274): no gelu on the aggregate, plain residual, then the shared dense layer
        rc = linear(agg, dp, pr.rows_q, pr.off_q, T, NQ, dp, dout, a->w_a, (int64_t)dout * dp, a->b_a, dout, trans, nullptr, nullptr, dout,
                    0, ws_a);
        if (rc != HGT_OK) return rc;
        mark(5);
        // y1 = LN_t(a_linear(agg) + x), kept in `out`
        rc = hgt_node_update_ex(trans, a->x, din, a->node_type, nullptr, a->ln_w, a->ln_b, a->use_norm, 0, NQ, dout, T, a->out, stream);
        if (rc != HGT_OK) return rc;
        int32_t* off2 = (int32_t*)(wb + w.off_off2);
        k_single_group<<<1, 64, 0, stream>>>(pr.off_q, T, off2);
        // mid = mid_linear(y1): [NQ][2*dout] in the (dead) Q|K region; gelu is applied where out_linear loads it
        float* mid = Q;
        if (w.off_k != w.off_q + (uint64_t)N * dp * 4 || 2 * dout > 2 * dp) return HGT_ERR_WORKSPACE;
        rc = linear(a->out, dout, pr.rows_q, off2, 1, NQ, dout, 2 * dout, a->mid_w, 0, a->mid_b, 0, mid, nullptr, nullptr, 2 * dout, 0,
                    ws_a);
        if (rc != HGT_OK) return rc;
        rc = linear(mid, 2 * dout, pr.rows_q, off2, 1, NQ, 2 * dout, dout, a->out_w, 0, a->out_b, 0, trans, nullptr, nullptr, dout, 0,
                    ws_a, 1);
        if (rc != HGT_OK) return rc;
        // out = out_norm(out_linear(...) + y1), in place over y1
        rc = hgt_node_update_ex(trans, a->out, dout, a->node_type, nullptr, a->out_ln_w, a->out_ln_b, 1, 1, NQ, dout, T, a->out, stream);
    } else if (fuse_update) {
        if (fresh || !pb) {
            rc = split_weights(a->w_a, (int64_t)dout * dp, T, dp, dout, ws_upd, stream);
            if (rc != HGT_OK) return rc;
        }
        rc = (f16 ? hgt_linear_update_f16x3 : hgt_linear_update_bf16x3)(agg, dp, pr.rows_q, pr.off_q, T, NQ, dp, dout, ws_upd, a->b_a, dout, a->x, din, a->skip, a->ln_w,
                                      a->ln_b, (a->use_norm ? 1 : 0) | ((a->flags & HGT_FLAG_NO_TILE_GEMM) ? 2 : 0), a->out, stream);
        if (rc != HGT_OK) return rc;
        mark(5);
        if (!no_unknown_rows) rc = hgt_zero_rows(pr.rows_q, pr.off_q + T, dout, a->out, stream);   // nodes of unknown type -> 0 (conv.py:120)
    } else {
        // (rows wider than 256 columns, e.g. n_hid = 400: the image of W_a is kept with the prepared weights like the fused forms')
        rc = linear(agg, dp, pr.rows_q, pr.off_q, T, NQ, dp, dout, a->w_a, (int64_t)dout * dp, a->b_a, dout, trans, nullptr, nullptr, dout,
                    0, ws_upd, 0, pb && !fresh);
        if (rc != HGT_OK) return rc;
        mark(5);
        rc = hgt_node_update(trans, a->x, din, a->node_type, a->skip, a->ln_w, a->ln_b, a->use_norm, NQ, dout, T, a->out, stream);
    }
    mark(6);
    return rc;
}
