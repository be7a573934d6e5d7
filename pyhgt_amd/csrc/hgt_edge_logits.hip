// Edge phase of HGTConv.forward: relation-aware attention logits, per-target softmax, attention-
// weighted aggregation.  Replaces conv.py:98-99,104,108-111 + PyG's gathers / scatter-add
// (conv.py:13,57) without ever materialising an E x d tensor.
//
// Algebra (validated against the reference):
//     s_e,h   = <q_i,h , k_j,h A[r,h]> pri[r,h]/sqrt(dk)  =  <A'[r,h] q_i,h , k_j,h>       (target-side transform)
//     agg_i,h = sum_e att_e (v_j,h M[r,h])                =  sum_r (sum_{e in (i,r)} att_e v_j,h) M[r,h]
// so the d_k x d_k relation matrices are applied once per (target, relation) SEGMENT, not per
// edge, and the per-edge work is one gathered row + a dot (pass 1) or an axpy (pass 2): HBM-bound.
//
// Work decomposition: edges are sorted by (dst tile of 64, relation, dst); one 64-lane wavefront
// takes one work item = <= 256 consecutive edges of one (tile, relation).  All its segments share
// the relation, so the wave keeps its slice of the relation matrix in REGISTERS (dk_pad*vec floats
// per lane, 128 for d=256/H=8) for the whole item instead of re-reading 32 KB per segment.
// Lane l owns `VEC` contiguous floats of a row (one coalesced 64*VEC*4-byte row read per
// wavefront instruction); head h = l / LPH; per-head dot products are reduced over LPH adjacent
// lanes with DPP.  Rows for the next UN edges are requested before the current ones are consumed.
#include "hgt_edge_common.h"

#ifndef HGT_LOGITS_XCD
#define HGT_LOGITS_XCD 1
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// pass 1: logits
// ---------------------------------------------------------------------------------------------
// S32 (chosen by the launcher: one head group and every Q / K / temporal row below 4 GiB from its base): compile-time row strides and
// 32-bit unsigned lane offsets -- as run-time 64-bit products every gathered row paid seven scalar multiplies / adds (r05 ISA audit,
// like the aggregation kernel's)
template <int VEC, int LPH, bool RTE, bool S32 = false>
__global__ __launch_bounds__(256) void k_edge_logits(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ Q,
    const float* __restrict__ K, const float* __restrict__ rteK, const float* __restrict__ attT, float* __restrict__ logits, int R,
    int HT, int rel_lo, int rel_hi, int item_lo, int item_hi, int items_cap) {
    // A wave covers DP = 64*VEC consecutive floats of a row = H = 64/LPH heads.  When the row has more heads (HT > H),
    // blockIdx.y selects the head group: used when the full-width relation fragment (dk_pad*vec floats per lane) would
    // not fit in registers (d = 512: 512 floats) -- narrower slices keep it register-resident.
    // With temporal encoding the table rows get their own slots (added at use), so the batch is 3/4 as deep (full depth needs 270 registers).
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH, UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>();
    const int hg = S32 ? 0 : blockIdx.y;
    if constexpr (S32) HT = H;
    const int64_t ld = S32 ? (int64_t)DP : (int64_t)HT * DKP;   // row stride of Q/K/V/rte tables in floats
    const int co = hg * DP;                 // first column of this head group
    constexpr bool HOIST = (DKP * VEC <= 128);
    __shared__ __attribute__((aligned(16))) float s_bounce[4][DP + 4 * (64 / LPH)];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware item order: workgroups are dealt to the 8 XCDs round-robin (linear id % 8), and the R + 1 items of a target tile
    // (adjacent in the item list) all read the tile's Q rows.  The XCDs take runs of 16 consecutive item groups in turn, so a
    // tile's items meet in one L2 at about the same time instead of pulling the tile through eight of them (HGT_LOGITS_XCD=0:
    // the plain order; gridDim.x is a multiple of 128)
    // (the item is requested TOGETHER with the header's item count -- index clamped to the table -- not behind it: one dependent
    //  round trip less in front of every wavefront's first row; sampled batches are a chain of such round trips, r6)
#if HGT_LOGITS_XCD
    // (chunks of HGT_XCD_CHUNK workgroups, dealt to the XCDs in turn: contiguous EIGHTHS of the list put all the heavy items of a
    //  skewed graph -- its hub tiles come first -- on one XCD: Zipf(0.8) logits 2.2 -> 3.2 ms)
    constexpr int XC = 16;
    const int q8 = (int)(blockIdx.x >> 3), vblock = (q8 / XC) * (8 * XC) + (int)(blockIdx.x & 7u) * XC + (q8 % XC);
#else
    const int vblock = blockIdx.x;
#endif
    const int item = item_lo + vblock * 4 + wib;
    const HgtItem it = items[min(item, items_cap - 1)];
    const int n_items = item_hi >= 0 ? item_hi : hdr->n_items;      // (item_lo, item_hi): the items of a target block, or (0, -1)
    if (item >= n_items) return;
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    const int h = lane / LPH, p = lane % LPH;
    if (rel < rel_lo || rel >= rel_hi) return;   // (multi-GPU path: only the relation buckets of the source rows that have arrived)

    if (rel >= R) {   // edges no meta relation claims: logit 0 (conv.py:68)
        for (int64_t i = (int64_t)beg * H + lane; i < (int64_t)end * H; i += 64) logits[(i / H) * HT + hg * H + (i % H)] = 0.0f;
        return;
    }

    float* bounce = s_bounce[wib];
    const float* __restrict__ fglob = attT + ((int64_t)(rel * HT + hg * H + h) * DKP) * DKP + p * VEC;
    float frag[HOIST ? DKP : 1][VEC];
    if constexpr (HOIST) {
#pragma unroll
        for (int j = 0; j < DKP; ++j) load_vec<VEC>(fglob + j * DKP, frag[j]);
    }

    int cur_dst = -1;
    float qt[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) qt[i] = 0.0f;

    // Software pipeline: two half-batches (A, B) of HB edges.  The loads of the NEXT half-batch are issued before the
    // current one is consumed, so the mat-vec / dot work of one half overlaps the gather latency of the other.  Every
    // issue is unconditional and of fixed size (indices clamped to the chunk, the Q row is fetched for every edge, not
    // only at segment starts) so that hipcc can keep counted s_waitcnt vmcnt(N): a conditional load inside the
    // pipeline makes it fall back to vmcnt(0), which serialises everything (measured: +45 %).
    constexpr int HB = UN / 2;
    for (int base = beg; base < end; base += 64) {
        const int nb = min(64, end - base);
        const int li = base + min(lane, nb - 1);
        const int my_src = esrc[li], my_dst = edst[li];
        const int my_rte = RTE ? (int)ertei[li] : 0;
        float krA[HB][VEC], qrA[HB][VEC], trA[RTE ? HB : 1][VEC];
        float krB[HB][VEC], qrB[HB][VEC], trB[RTE ? HB : 1][VEC];
#define HGT_ISSUE(KR, QR, TR, I0)                                                                  \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        const int idx = min((I0) + u, nb - 1);                                                     \
        const int s_ = __builtin_amdgcn_readlane(my_src, idx);                                     \
        const int d_ = __builtin_amdgcn_readlane(my_dst, idx);                                     \
        if constexpr (S32) {                                                                       \
            load_vec<VEC>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(K) + (unsigned)((unsigned)s_ * (unsigned)(DP * 4) + (unsigned)(lane * VEC * 4))), KR[u]); \
            if constexpr (RTE) {                                                                   \
                const int ri = __builtin_amdgcn_readlane(my_rte, idx);                             \
                load_vec<VEC>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(rteK) + (unsigned)((unsigned)ri * (unsigned)(DP * 4) + (unsigned)(lane * VEC * 4))), TR[u]); \
            }                                                                                      \
            load_vec<VEC>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(Q) + (unsigned)((unsigned)d_ * (unsigned)(DP * 4) + (unsigned)(lane * VEC * 4))), QR[u]); \
        } else {                                                                                   \
            load_vec<VEC>(K + (int64_t)s_ * ld + co + lane * VEC, KR[u]);                          \
            if constexpr (RTE) {                                                                   \
                const int ri = __builtin_amdgcn_readlane(my_rte, idx);                             \
                load_vec<VEC>(rteK + (int64_t)ri * ld + co + lane * VEC, TR[u]);                   \
            }                                                                                      \
            load_vec<VEC>(Q + (int64_t)d_ * ld + co + lane * VEC, QR[u]);                          \
        }                                                                                          \
    }
#define HGT_PROCESS(KR, QR, TR, I0)                                                                \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        if ((I0) + u < nb) {                                                                       \
            const int d_ = __builtin_amdgcn_readlane(my_dst, (I0) + u);                            \
            if (d_ != cur_dst) {                                                                   \
                head_matvec<VEC, DKP, HOIST>(QR[u], bounce, lane, h, frag, fglob, qt);             \
                cur_dst = d_;                                                                      \
            }                                                                                      \
            float part = 0.0f;                                                                     \
            _Pragma("unroll") for (int i = 0; i < VEC; ++i) {                                      \
                float kv = KR[u][i];                                                               \
                if constexpr (RTE) kv += TR[u][i];                                                 \
                part = fmaf(qt[i], kv, part);                                                      \
            }                                                                                      \
            part = head_allreduce<LPH>(part);                                                      \
            if (p == 0) logits[(int64_t)(base + (I0) + u) * HT + hg * H + h] = part;               \
        }                                                                                          \
    }
        cur_dst = -1;   // the first edge of a chunk always recomputes q~ (its Q row is loaded anyway)
        HGT_ISSUE(krA, qrA, trA, 0)
        for (int i0 = 0; i0 < nb; i0 += 2 * HB) {
            HGT_ISSUE(krB, qrB, trB, i0 + HB)
            HGT_PROCESS(krA, qrA, trA, i0)
            HGT_ISSUE(krA, qrA, trA, i0 + 2 * HB)
            HGT_PROCESS(krB, qrB, trB, i0 + HB)
        }
#undef HGT_ISSUE
#undef HGT_PROCESS
    }
}

// ---------------------------------------------------------------------------------------------
// softmax over the in-edges of each target, per head (PyG utils.softmax, conv.py:108); in place
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_softmax(const int32_t* __restrict__ segptr, float* __restrict__ s, int64_t NQ,
                                                      int H, int R) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t dst = idx / H;
    const int h = (int)(idx % H);
    if (dst >= NQ) return;
    const int64_t tile = dst / HGT_TD, dl = dst % HGT_TD;
    float m = -INFINITY;
    for (int r = 0; r <= R; ++r) {
        const int64_t b = (tile * (R + 1) + r) * HGT_TD + dl;
        const int e0 = segptr[b], e1 = segptr[b + 1];
        for (int e = e0; e < e1; ++e) m = fmaxf(m, s[(int64_t)e * H + h]);
    }
    float z = 0.0f;
    for (int r = 0; r <= R; ++r) {
        const int64_t b = (tile * (R + 1) + r) * HGT_TD + dl;
        const int e0 = segptr[b], e1 = segptr[b + 1];
        for (int e = e0; e < e1; ++e) z += expf(s[(int64_t)e * H + h] - m);
    }
    const float inv = 1.0f / (z + 1e-16f);
    for (int r = 0; r <= R; ++r) {
        const int64_t b = (tile * (R + 1) + r) * HGT_TD + dl;
        const int e0 = segptr[b], e1 = segptr[b + 1];
        for (int e = e0; e < e1; ++e) {
            const int64_t o = (int64_t)e * H + h;
            s[o] = expf(s[o] - m) * inv;
        }
    }
}

__global__ void k_att_export(const int32_t* __restrict__ eid, const float* __restrict__ att, float* __restrict__ out, int64_t E, int H,
                             int Hout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * H) return;
    const int64_t p = i / H;
    const int h = (int)(i % H);
    if (h < Hout) out[(int64_t)eid[p] * Hout + h] = att[i];
}

__global__ void k_relation_pack(const float* __restrict__ ratt, const float* __restrict__ rmsg, const float* __restrict__ rpri,
                                int R, int H, int HL, int dk, int dkp, float* __restrict__ attT, float* __restrict__ msgP) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)R * HL * dkp * dkp;
    if (i >= total) return;
    const int b = (int)(i % dkp), a = (int)((i / dkp) % dkp);
    const int64_t rhl = i / ((int64_t)dkp * dkp);
    const int hh = (int)(rhl % HL);
    const int64_t rh = (rhl / HL) * H + hh;        // index into the [R][H] parameter tensors
    float va = 0.0f, vm = 0.0f;
    if (a < dk && b < dk && hh < H) {
        // attT[r][h][c=a][k=b] = att[r][h][k=b][c=a] * pri / sqrt(dk)
        va = ratt[(rh * dk + b) * dk + a] * rpri[rh] / sqrtf((float)dk);
        vm = rmsg[(rh * dk + a) * dk + b];
    }
    attT[i] = va;
    msgP[i] = vm;
}

template <int VEC, int LPH>
struct LaunchLogits {
    static int run(const HgtPlanView& pv, const float* Q, const float* K, const float* rteK, const float* attT, float* logits,
                   int R, int HT, int rel_lo, int rel_hi, int item_lo, int item_hi, int small32, hipStream_t stream) {
        const int64_t n_launch = item_hi >= 0 ? (int64_t)(item_hi - item_lo) : pv.L.max_items;
        const unsigned blocks = ((unsigned)((n_launch + 3) / 4) + 127u) & ~127u;      // (a multiple of 8 XCDs x 16: XCD-aware item order)
        dim3 grid(blocks, (unsigned)(HT / (64 / LPH)));
        if (small32 && grid.y == 1) {
            if (rteK)
                k_edge_logits<VEC, LPH, true, true><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attT, logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, (int)pv.L.max_items);
            else
                k_edge_logits<VEC, LPH, false, true><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attT, logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, (int)pv.L.max_items);
        } else if (rteK)
            k_edge_logits<VEC, LPH, true><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attT, logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, (int)pv.L.max_items);
        else
            k_edge_logits<VEC, LPH, false><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attT, logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, (int)pv.L.max_items);
        return HGT_OK;
    }
};

}  // namespace

extern "C" int hgt_relation_pack(const float* relation_att, const float* relation_msg, const float* relation_pri,
                                 int32_t R, int32_t H, int32_t HL, int32_t d_k, int32_t dk_pad, float* att_t, float* msg_p, void* stream) {
    if (!relation_att || !relation_msg || !relation_pri || !att_t || !msg_p || R <= 0 || H <= 0 || HL < H || d_k <= 0 || dk_pad < d_k)
        return HGT_ERR_INVALID_ARG;
    const int64_t total = (int64_t)R * HL * dk_pad * dk_pad;
    k_relation_pack<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(relation_att, relation_msg, relation_pri, R, H, HL,
                                                                                      d_k, dk_pad, att_t, msg_p);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

// hgt_edge_logits_mfma.hip
int hgt_launch_logits_mfma(int vec, int lph, int mode, const HgtPlanView& pv, const float* Q, const float* K, const float* rteK,
                           const unsigned short* attF, float* logits, int R, int HT, int rel_lo, int rel_hi, int item_lo, int item_hi, hipStream_t stream);

// head-group split of the matrix-core kernels (hgt_edge_agg_mfma.hip): the wave's slice must be <= 256 columns
static int mfma_logits_split_for(int vec_full, int lph_full) {
    int s = 1;
    while (vec_full / s > 4 && lph_full * s * 2 <= 64) s *= 2;
    return (vec_full / s <= 4) ? s : 0;
}

static int edge_logits_impl(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad, const float* Q,
                            const float* K, const float* rte_k, const float* att_t, float* logits, int rel_lo, int rel_hi, void* stream,
                            const void* att_frag = nullptr, int frag_f16 = 0, int item_lo = 0, int item_hi = -1) {
    if (!plan || !Q || !K || !att_t || !logits || H <= 0 || 64 % H != 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    if (rel_lo < 0 || rel_hi > R + 1 || rel_lo > rel_hi) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    if (att_frag) {      // target-side transforms on the matrix cores; layouts it does not cover take the vector-ALU kernel below
        const int spm = mfma_logits_split_for(dk_pad / lph, lph);
        if (spm != 0) {
            int rc = hgt_launch_logits_mfma(dk_pad / lph / spm, lph * spm, (frag_f16 & 7) | (hgt_item_edges(E) << 8), pv, Q, K, rte_k, (const unsigned short*)att_frag,
                                            logits, (int)R, (int)H, rel_lo, rel_hi, item_lo, item_hi, (hipStream_t)stream);
            if (rc == HGT_OK) HGT_CHECK_LAUNCH();
            if (rc != HGT_ERR_UNSUPPORTED) return rc;
        }
    }
    const int sp = head_split_for(dk_pad / lph, lph, dk_pad);
    // (every row of Q / K -- and of the 240-row temporal table -- below 4 GiB from its base: the 32-bit-offset instantiation)
    const int small32 = ((uint64_t)N * (uint64_t)H * (uint64_t)dk_pad * 4u < (1ull << 32)) ? 1 : 0;
    int rc = dispatch_layout<LaunchLogits>(dk_pad / lph / sp, lph * sp, pv, Q, K, rte_k, att_t, logits, (int)R, (int)H, rel_lo, rel_hi,
                                           item_lo, item_hi, small32, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_logits(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                               const float* Q, const float* K, const float* rte_k, const float* att_t, float* logits, void* stream) {
    return edge_logits_impl(plan, N, E, T, R, H, dk_pad, Q, K, rte_k, att_t, logits, 0, R + 1, stream);
}

extern "C" int hgt_edge_logits_mfma(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                    const float* Q, const float* K, const float* rte_k, const float* att_t, const void* att_frag,
                                    int32_t frag_f16, float* logits, void* stream) {
    if (!att_frag) return HGT_ERR_INVALID_ARG;
    return edge_logits_impl(plan, N, E, T, R, H, dk_pad, Q, K, rte_k, att_t, logits, 0, R + 1, stream, att_frag, frag_f16);
}

extern "C" int hgt_edge_logits_slice(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                     const float* Q, const float* K, const float* rte_k, const float* att_t, float* logits,
                                     int32_t rel_lo, int32_t rel_hi, void* stream) {
    return edge_logits_impl(plan, N, E, T, R, H, dk_pad, Q, K, rte_k, att_t, logits, rel_lo, rel_hi, stream);
}

// ABI 6: the work items [item_begin, item_end) only (the items of a range of target tiles: hgt_plan_tile_items_offset) -- one
// target block of the multi-GPU path.  att_frag may be NULL (vector-ALU kernel).
extern "C" int hgt_edge_logits_range(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                     const float* Q, const float* K, const float* rte_k, const float* att_t, const void* att_frag,
                                     int32_t frag_f16, float* logits, int32_t item_begin, int32_t item_end, void* stream) {
    if (item_begin < 0 || item_end < item_begin) return HGT_ERR_INVALID_ARG;
    if (item_begin == item_end) return HGT_OK;
    return edge_logits_impl(plan, N, E, T, R, H, dk_pad, Q, K, rte_k, att_t, logits, 0, R + 1, stream, att_frag, frag_f16, item_begin, item_end);
}

extern "C" int hgt_edge_softmax(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, float* logits_att,
                                void* stream) {
    if (!plan || !logits_att || H <= 0) return HGT_ERR_INVALID_ARG;
    if (E == 0 || N == 0) return HGT_OK;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    const int64_t threads = N * H;
    k_edge_softmax<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(pv.segptr, logits_att, N, H, R);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}


extern "C" int hgt_att_export(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, const float* att_sorted,
                              float* att_out, int32_t H_out, void* stream) {
    if (!plan || !att_sorted || !att_out || H <= 0 || H_out <= 0 || H_out > H) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    k_att_export<<<(unsigned)((E * H + 255) / 256), 256, 0, (hipStream_t)stream>>>(pv.eid, att_sorted, att_out, E, H, H_out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
