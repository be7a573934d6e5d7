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
#include "hgt_common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over the LPH adjacent lanes that hold one head; every lane of the group gets the total
template <int LPH>
__device__ __forceinline__ float head_allreduce(float v) {
    if (LPH >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    if (LPH >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    if (LPH >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
    if (LPH >= 16) v += dpp_f<0x140>(v);  // row_mirror
    if (LPH >= 32) v += __shfl_xor(v, 16);
    if (LPH >= 64) v += __shfl_xor(v, 32);
    return v;
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&o)[VEC]) {
    if constexpr (VEC == 1) {
        o[0] = p[0];
    } else if constexpr (VEC == 2) {
        float2 t = *reinterpret_cast<const float2*>(p);
        o[0] = t.x; o[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) {
            float4 t = *reinterpret_cast<const float4*>(p + 4 * i);
            o[4 * i] = t.x; o[4 * i + 1] = t.y; o[4 * i + 2] = t.z; o[4 * i + 3] = t.w;
        }
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec_lds(float* p, const float (&o)[VEC]) {
    if constexpr (VEC == 1) {
        p[0] = o[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
    } else {
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) *reinterpret_cast<float4*>(p + 4 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
}

// y[i] = sum_j xin[j] * F[j][i]  over one head (DKP inputs, this lane's VEC outputs).
// xin is spread over the LPH lanes of the head -> bounced through a wave-private LDS row.
// F comes from registers (HOIST) or from the packed relation matrix in global memory (L1/L2).
template <int VEC, int DKP, bool HOIST>
__device__ __forceinline__ void head_matvec(const float (&xv)[VEC], float* bounce, int lane, int h,
                                            const float (&frag)[HOIST ? DKP : 1][VEC], const float* __restrict__ fglob,
                                            float (&y)[VEC]) {
    // head h's DKP inputs live at h*(DKP+4): the +4 floats of padding put the (up to 4) heads that one
    // ds_read_b128 lane group reads on different banks (unpadded, heads 0/2 and 1/3 collided: 2-way conflict)
    store_vec_lds<VEC>(bounce + lane * VEC + (lane / (DKP / VEC)) * 4, xv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = 0.0f;
    const float* xb = bounce + h * (DKP + 4);
#pragma unroll
    for (int j4 = 0; j4 < DKP / 4; ++j4) {
        const float4 xx = *reinterpret_cast<const float4*>(xb + 4 * j4);
        const float xs[4] = {xx.x, xx.y, xx.z, xx.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if constexpr (HOIST) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) y[i] = fmaf(xs[jj], frag[4 * j4 + jj][i], y[i]);
            } else {
                float f[VEC];
                load_vec<VEC>(fglob + (4 * j4 + jj) * DKP, f);
#pragma unroll
                for (int i = 0; i < VEC; ++i) y[i] = fmaf(xs[jj], f[i], y[i]);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();   // the bounce row may be rewritten only after every lane has read it
}

template <int VEC>
constexpr int unroll_for() { return VEC <= 4 ? 8 : (VEC == 8 ? 4 : 2); }

// ---------------------------------------------------------------------------------------------
// pass 1: logits
// ---------------------------------------------------------------------------------------------
template <int VEC, int LPH, bool RTE>
__global__ __launch_bounds__(256) void k_edge_logits(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ Q,
    const float* __restrict__ K, const float* __restrict__ rteK, const float* __restrict__ attT, float* __restrict__ logits, int R,
    int HT) {
    // A wave covers DP = 64*VEC consecutive floats of a row = H = 64/LPH heads.  When the row has more heads (HT > H),
    // blockIdx.y selects the head group: used when the full-width relation fragment (dk_pad*vec floats per lane) would
    // not fit in registers (d = 512: 512 floats) -- narrower slices keep it register-resident.
    // With temporal encoding the table rows get their own slots (added at use), so the batch is half as deep.
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH, UN = RTE ? unroll_for<VEC>() / 2 : unroll_for<VEC>();
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;   // row stride of Q/K/V/rte tables in floats
    const int co = hg * DP;                 // first column of this head group
    constexpr bool HOIST = (DKP * VEC <= 128);
    __shared__ __attribute__((aligned(16))) float s_bounce[4][DP + 4 * (64 / LPH)];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wib;
    if (item >= hdr->n_items) return;
    const HgtItem it = items[item];
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    const int h = lane / LPH, p = lane % LPH;

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
        load_vec<VEC>(K + (int64_t)s_ * ld + co + lane * VEC, KR[u]);                              \
        if constexpr (RTE) {                                                                       \
            const int ri = __builtin_amdgcn_readlane(my_rte, idx);                                 \
            load_vec<VEC>(rteK + (int64_t)ri * ld + co + lane * VEC, TR[u]);                       \
        }                                                                                          \
        load_vec<VEC>(Q + (int64_t)d_ * ld + co + lane * VEC, QR[u]);                              \
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

// ---------------------------------------------------------------------------------------------
// pass 2: softmax + aggregation.  One WAVEFRONT owns a sub-tile of 16 consecutive target nodes and
// walks all relations of it in ascending order: for relation r its edges are the contiguous sorted
// range segptr[(tile, r, 16*sub)] .. segptr[(tile, r, 16*sub+16)].  Because the wave sees EVERY
// in-edge of its targets, the per-target softmax (conv.py:108) is evaluated online, with no separate
// normalisation pass over the logits:
//   * inside a (target, relation) segment: running max m, running sum l and U = sum exp(s-m) v in
//     registers (rescaled when the max grows);
//   * at the segment end: z = U M[r] (register-resident slice of M[r], re-read from L2 once per
//     (sub-tile, relation)), then merged into the target's state kept in WAVE-PRIVATE LDS:
//     acc = acc*exp(m_t - m') + z*exp(m - m'), l_t likewise ("planar" [row][i][lane] layout: the 64
//     lanes of an LDS access hit 64 distinct banks);
//   * at the end: agg = acc / (l_t + 1e-16)  -- identical to PyG's exp(s-max)/(sum exp(s-max)+1e-16)
//     because the final running max is the true max.  Unclaimed edges (bucket R) count with logit 0
//     and no message (conv.py:68-69).
// No atomics, no barriers, no zero-fill of agg; fixed summation order -> bitwise reproducible.
// Rows are written once with plain coalesced stores (optionally through gelu, conv.py:119).
// ---------------------------------------------------------------------------------------------
constexpr int HGT_SUB = 16;       // targets per wavefront
constexpr float HGT_NEG = -1.0e30f;

template <int VEC, int LPH, bool RTE, bool HUBS>
__device__ __forceinline__ void aggregate_subtile(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgP, float* __restrict__ agg, int R, int64_t NQ, int apply_gelu,
    int HT, unsigned hub_mask, float (&s_acc)[4][16 * 64 * VEC], float (&s_bounce)[4][64 * VEC + 4 * (64 / LPH)], float (&s_ml)[4][2][16 * 16]) {
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH, UN = RTE ? unroll_for<VEC>() / 2 : unroll_for<VEC>();
    const int hg = blockIdx.y;              // head group (see k_edge_logits)
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP;
    constexpr bool HOIST = (DKP * VEC <= 128);

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a workgroup covers 64 consecutive targets (4 waves x 16); the plan's destination tile may be larger
    const int64_t row0 = (int64_t)blockIdx.x * 64 + wib * HGT_SUB;
    if (row0 >= NQ) return;
    const int tile = (int)(row0 / HGT_TD);
    const int within = (int)(row0 % HGT_TD);
    const int h = lane / LPH, p = lane % LPH;
    float* acc = s_acc[wib];
    float* bounce = s_bounce[wib];
    float* s_m = s_ml[wib][0];
    float* s_l = s_ml[wib][1];

#pragma unroll
    for (int j = 0; j < HGT_SUB * VEC / 4; ++j) *reinterpret_cast<float4*>(&acc[j * 256 + lane * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < HGT_SUB * 16 / 64; ++j) { s_m[j * 64 + lane] = HGT_NEG; s_l[j * 64 + lane] = 0.0f; }

    for (int rel = 0; rel <= R; ++rel) {
      const int64_t b0 = ((int64_t)tile * (R + 1) + rel) * HGT_TD + within;
      // maximal runs [dl0, dl1) of non-hub targets: one run covering the whole sub-tile unless it contains a hub
      for (int dl0 = 0; dl0 < HGT_SUB;) {
        int dl1 = HGT_SUB;
        if constexpr (HUBS) {
            if ((hub_mask >> dl0) & 1u) { ++dl0; continue; }
            dl1 = dl0 + 1;
            while (dl1 < HGT_SUB && !((hub_mask >> dl1) & 1u)) ++dl1;
        }
        const int beg = __builtin_amdgcn_readfirstlane(segptr[b0 + dl0]);
        const int end = __builtin_amdgcn_readfirstlane(segptr[b0 + dl1]);
        dl0 = dl1;
        if (beg == end) continue;
        const bool claimed = rel < R;   // bucket R: logit 0, no message

        const float* __restrict__ fglob = msgP + ((int64_t)((claimed ? rel : 0) * HT + hg * H + h) * DKP) * DKP + p * VEC;
        float frag[HOIST ? DKP : 1][VEC];
        if constexpr (HOIST) {
            if (claimed) {
#pragma unroll
                for (int j = 0; j < DKP; ++j) load_vec<VEC>(fglob + j * DKP, frag[j]);
            }
        }

        int cur_dst = -1;
        float U[VEC], m_seg = HGT_NEG, l_seg = 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) U[i] = 0.0f;

        auto flush = [&]() {
            if (cur_dst >= 0) {
                float z[VEC];
                if (claimed) {
                    head_matvec<VEC, DKP, HOIST>(U, bounce, lane, h, frag, fglob, z);
                } else {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) z[i] = 0.0f;
                }
                const int dl = cur_dst - (int)row0;
                const float m_t = s_m[dl * 16 + h], l_t = s_l[dl * 16 + h];
                const float m_new = fmaxf(m_t, m_seg);
                const float ca = __expf(m_t - m_new), cb = __expf(m_seg - m_new);
                float* o = acc + dl * DP + lane;
#pragma unroll
                for (int i = 0; i < VEC; ++i) o[i * 64] = o[i * 64] * ca + z[i] * cb;   // wave-private read-modify-write
                if (p == 0) { s_m[dl * 16 + h] = m_new; s_l[dl * 16 + h] = l_t * ca + l_seg * cb; }
            }
        };

        for (int base = beg; base < end; base += 64) {
            const int nb = min(64, end - base);
            const int li = base + min(lane, nb - 1);
            const int my_src = esrc[li], my_dst = edst[li];
            const int my_rte = RTE ? (int)ertei[li] : 0;
            for (int i0 = 0; i0 < nb; i0 += UN) {
                float vr[UN][VEC], sl[UN], tr[RTE ? UN : 1][VEC];
                int dsts[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int idx = min(i0 + u, nb - 1);
                    const int s = __builtin_amdgcn_readlane(my_src, idx);
                    dsts[u] = __builtin_amdgcn_readlane(my_dst, idx);
                    if (claimed) {
                        load_vec<VEC>(V + (int64_t)s * ld + co + lane * VEC, vr[u]);
                        sl[u] = logits[(int64_t)(base + idx) * HT + hg * H + h];
                        if constexpr (RTE) {
                            const int ri = __builtin_amdgcn_readlane(my_rte, idx);
                            load_vec<VEC>(rteV + (int64_t)ri * ld + co + lane * VEC, tr[u]);
                        }
                    } else {
                        sl[u] = 0.0f;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) vr[u][i] = 0.0f;
                        if constexpr (RTE) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) tr[u][i] = 0.0f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (i0 + u < nb) {
                        if (dsts[u] != cur_dst) {
                            flush();
#pragma unroll
                            for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
                            m_seg = sl[u];   // reference point of the segment: its first logit (not necessarily the max)
                            l_seg = 0.0f;
                            cur_dst = dsts[u];
                        }
                        // Deferred rescaling: weights are exp(s - m_seg) relative to the segment's reference; the reference
                        // is only moved (and U, l rescaled) when a logit exceeds it by more than 40 (exp(40) ~ 2e17 is far
                        // from fp32 overflow), which is a wave-uniform rare branch.  Any reference gives the same softmax:
                        // the merge below and the final division are invariant to it.
                        float dlt = sl[u] - m_seg;
                        if (__builtin_amdgcn_ballot_w64(dlt > 40.0f) != 0) {
                            const float m_new = fmaxf(m_seg, sl[u]);
                            const float sc = __expf(m_seg - m_new);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) U[i] *= sc;
                            l_seg *= sc;
                            m_seg = m_new;
                            dlt = sl[u] - m_seg;
                        }
                        const float pe = __expf(dlt);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            float vv = vr[u][i];
                            if constexpr (RTE) vv += tr[u][i];
                            U[i] = fmaf(pe, vv, U[i]);
                        }
                        l_seg += pe;
                    }
                }
            }
        }
        flush();
      }
    }

    // write-out: normalise, un-permute the planar layout, one coalesced row store per wave instruction
    for (int r = 0; r < HGT_SUB; ++r) {
        const int64_t row = row0 + r;
        if (row >= NQ) break;
        if (HUBS && ((hub_mask >> r) & 1u)) continue;   // written by k_hub_finalize
        const float inv = 1.0f / (s_l[r * 16 + h] + 1e-16f);
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            o[i] = acc[r * DP + i * 64 + lane] * inv;
            if (apply_gelu) o[i] = 0.5f * o[i] * (1.0f + erff(o[i] * 0.70710678118654752440f));
        }
        float* g = agg + row * ld + co + lane * VEC;
        if constexpr (VEC == 1) {
            g[0] = o[0];
        } else if constexpr (VEC == 2) {
            *reinterpret_cast<float2*>(g) = make_float2(o[0], o[1]);
        } else {
#pragma unroll
            for (int i = 0; i < VEC / 4; ++i) *reinterpret_cast<float4*>(g + 4 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
        }
    }
}

// Sub-tiles without a hub target (all of them on c2) take the HUBS = false instantiation: its loop nest is the plain
// "one range per relation" walk (the run logic costs ~4 % when it is compiled into the hot path).
template <int VEC, int LPH, bool RTE>
__global__ __launch_bounds__(256) void k_edge_aggregate(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgP, float* __restrict__ agg, int R, int64_t NQ, int apply_gelu,
    int HT, const int32_t* __restrict__ hub_slot) {
    __shared__ __attribute__((aligned(16))) float s_acc[4][16 * 64 * VEC];
    __shared__ __attribute__((aligned(16))) float s_bounce[4][64 * VEC + 4 * (64 / LPH)];
    __shared__ float s_ml[4][2][16 * 16];   // running max / sum per (target, head); H <= 16
    // hub targets (in-degree > HGT_HUB_DEG, plan) are aggregated by the hub kernels below; this wave skips them
    unsigned hub_mask = 0;
    if (hub_slot) {
        const int lane = threadIdx.x & 63;
        const int64_t rr = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + (lane & 15);
        const bool is_hub = (lane < 16) && (rr < NQ) && (hub_slot[rr] >= 0);
        hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    }
    if (hub_mask == 0)
        aggregate_subtile<VEC, LPH, RTE, false>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, 0u, s_acc,
                                                s_bounce, s_ml);
    else
        aggregate_subtile<VEC, LPH, RTE, true>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, hub_mask,
                                               s_acc, s_bounce, s_ml);
}

// ---------------------------------------------------------------------------------------------
// Hub path.  A target with more than HGT_HUB_DEG in-edges would make the single wavefront that owns its sub-tile
// walk all of them (Zipf targets: 8 -> 190 ms at c2 size).  Hubs are therefore skipped by k_edge_aggregate and
// handled by a fixed grid of wavefronts that split every (hub, relation) edge range into HUB_CHUNKS pieces:
//   k_hub_max        max logit per (hub, head)  (wave reduce + one atomicMax per head and piece)
//   k_hub_accumulate sum exp(s - m) and (sum exp(s - m) V[src]) M[rel] per piece, atomically added to the hub's
//                    fp32 accumulators (any reference m gives the same softmax; m = max(max logit, 0) covers the
//                    unclaimed bucket, whose logits are 0)
//   k_hub_finalize   agg[hub] = gelu(acc / (l + 1e-16))
// All three exit immediately when the plan found no hub (hdr->n_hubs == 0).
// ---------------------------------------------------------------------------------------------
constexpr int HUB_CHUNKS = 64;
constexpr int HUB_GRID_WAVES = 8192;

__device__ __forceinline__ int f2ord(float f) { const int b = __builtin_bit_cast(int, f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float ord2f(int o) { return __builtin_bit_cast(float, o ^ ((o >> 31) & 0x7fffffff)); }

struct HubBuffers {
    int* mx;      // [max_hubs][HT] ordered-int max logit
    float* l;     // [max_hubs][HT]
    float* acc;   // [max_hubs][HT * DKP]
};

__global__ void k_hub_init(const HgtPlanHeader* __restrict__ hdr, HubBuffers hb, int HT, int dfull) {
    const int n = hdr->n_hubs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = 2 * HT + dfull;
    if (i >= (int64_t)n * per) return;
    const int slot = (int)(i / per), r = (int)(i % per);
    if (r < HT) hb.mx[slot * HT + r] = f2ord(-1.0e30f);
    else if (r < 2 * HT) hb.l[slot * HT + r - HT] = 0.0f;
    else hb.acc[(int64_t)slot * dfull + r - 2 * HT] = 0.0f;
}

// work id w -> (hub slot, relation bucket, piece); edges [pb, pe) of that piece
__device__ __forceinline__ bool hub_piece(int w, int n_hubs, int R, const int32_t* __restrict__ hub_list,
                                          const int32_t* __restrict__ segptr, int& slot, int& rel, int& pb, int& pe) {
    const int per_hub = (R + 1) * HUB_CHUNKS;
    slot = w / per_hub;
    if (slot >= n_hubs) return false;
    const int r2 = w - slot * per_hub;
    rel = r2 / HUB_CHUNKS;
    const int c = r2 - rel * HUB_CHUNKS;
    const int64_t dst = hub_list[slot];
    const int64_t b = ((dst / HGT_TD) * (R + 1) + rel) * HGT_TD + dst % HGT_TD;
    const int beg = segptr[b], end = segptr[b + 1];
    const int len = end - beg, piece = (len + HUB_CHUNKS - 1) / HUB_CHUNKS;
    pb = beg + c * piece;
    pe = min(end, pb + piece);
    return pb < pe;
}

__global__ __launch_bounds__(256) void k_hub_max(const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list,
                                                 const int32_t* __restrict__ segptr, const float* __restrict__ logits, int R, int HT,
                                                 HubBuffers hb) {
    const int n_hubs = hdr->n_hubs;
    if (n_hubs == 0) return;
    const int lane = threadIdx.x & 63;
    const int epw = 64 / HT;                       // edges per wave iteration
    const int hh = lane % HT, eo = lane / HT;
    const int total = n_hubs * (R + 1) * HUB_CHUNKS;
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += gridDim.x * 4) {
        int slot, rel, pb, pe;
        if (!hub_piece(w, n_hubs, R, hub_list, segptr, slot, rel, pb, pe)) continue;
        float m = -1.0e30f;
        if (rel < R) {
            for (int e = pb + eo; e < pe; e += epw) m = fmaxf(m, logits[(int64_t)e * HT + hh]);
        } else {
            m = 0.0f;                              // unclaimed bucket: logits are 0
        }
        for (int sft = HT; sft < 64; sft <<= 1) m = fmaxf(m, __shfl_xor(m, sft));
        if (lane < HT) atomicMax(&hb.mx[slot * HT + hh], f2ord(m));
    }
}

template <int VEC, int LPH, bool RTE>
__global__ __launch_bounds__(256) void k_hub_accumulate(
    const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list, const int32_t* __restrict__ segptr,
    const int32_t* __restrict__ esrc, const uint16_t* __restrict__ ertei, const float* __restrict__ logits,
    const float* __restrict__ V, const float* __restrict__ rteV, const float* __restrict__ msgP, int R, int HT, HubBuffers hb) {
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH, UN = RTE ? unroll_for<VEC>() / 2 : unroll_for<VEC>();
    constexpr bool HOIST = (DKP * VEC <= 128);
    __shared__ __attribute__((aligned(16))) float s_bounce[4][DP + 4 * (64 / LPH)];
    const int n_hubs = hdr->n_hubs;
    if (n_hubs == 0) return;
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane / LPH, p = lane % LPH;
    float* bounce = s_bounce[wib];
    const int total = n_hubs * (R + 1) * HUB_CHUNKS;
    for (int w = blockIdx.x * 4 + wib; w < total; w += gridDim.x * 4) {
        int slot, rel, pb, pe;
        if (!hub_piece(w, n_hubs, R, hub_list, segptr, slot, rel, pb, pe)) continue;
        slot = __builtin_amdgcn_readfirstlane(slot);
        rel = __builtin_amdgcn_readfirstlane(rel);
        pb = __builtin_amdgcn_readfirstlane(pb);
        pe = __builtin_amdgcn_readfirstlane(pe);
        const float mref = fmaxf(ord2f(hb.mx[slot * HT + hg * H + h]), 0.0f);
        float l_part = 0.0f;
        if (rel >= R) {                            // unclaimed: logit 0, no message
            l_part = (float)(pe - pb) * __expf(0.0f - mref);
            if (p == 0) atomicAdd(&hb.l[slot * HT + hg * H + h], l_part);
            continue;
        }
        const float* __restrict__ fglob = msgP + ((int64_t)(rel * HT + hg * H + h) * DKP) * DKP + p * VEC;
        float frag[HOIST ? DKP : 1][VEC];
        if constexpr (HOIST) {
#pragma unroll
            for (int j = 0; j < DKP; ++j) load_vec<VEC>(fglob + j * DKP, frag[j]);
        }
        float U[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
        for (int base = pb; base < pe; base += 64) {
            const int nb = min(64, pe - base);
            const int li = base + min(lane, nb - 1);
            const int my_src = esrc[li];
            const int my_rte = RTE ? (int)ertei[li] : 0;
            for (int i0 = 0; i0 < nb; i0 += UN) {
                float vr[UN][VEC], sl[UN], tr[RTE ? UN : 1][VEC];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int idx = min(i0 + u, nb - 1);
                    const int s = __builtin_amdgcn_readlane(my_src, idx);
                    load_vec<VEC>(V + (int64_t)s * ld + co + lane * VEC, vr[u]);
                    sl[u] = logits[(int64_t)(base + idx) * HT + hg * H + h];
                    if constexpr (RTE) {
                        const int ri = __builtin_amdgcn_readlane(my_rte, idx);
                        load_vec<VEC>(rteV + (int64_t)ri * ld + co + lane * VEC, tr[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (i0 + u < nb) {
                        const float pe_ = __expf(sl[u] - mref);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            float vv = vr[u][i];
                            if constexpr (RTE) vv += tr[u][i];
                            U[i] = fmaf(pe_, vv, U[i]);
                        }
                        l_part += pe_;
                    }
                }
            }
        }
        float z[VEC];
        head_matvec<VEC, DKP, HOIST>(U, bounce, lane, h, frag, fglob, z);
        float* o = hb.acc + (int64_t)slot * ld + co + lane * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(o + i, z[i]);
        if (p == 0) unsafeAtomicAdd(&hb.l[slot * HT + hg * H + h], l_part);
    }
}

__global__ __launch_bounds__(256) void k_hub_finalize(const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list,
                                                      HubBuffers hb, float* __restrict__ agg, int HT, int dkp, int64_t NQ,
                                                      int apply_gelu) {
    const int n_hubs = hdr->n_hubs;
    const int lane = threadIdx.x & 63;
    const int dfull = HT * dkp;
    for (int slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < n_hubs; slot += gridDim.x * 4) {
        const int64_t row = hub_list[slot];
        if (row >= NQ) continue;
        for (int c = lane; c < dfull; c += 64) {
            float v = hb.acc[(int64_t)slot * dfull + c] / (hb.l[slot * HT + c / dkp] + 1e-16f);
            if (apply_gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            agg[row * dfull + c] = v;
        }
    }
}

__global__ void k_att_export(const int32_t* __restrict__ eid, const float* __restrict__ att, float* __restrict__ out, int64_t E, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * H) return;
    const int64_t p = i / H;
    const int h = (int)(i % H);
    out[(int64_t)eid[p] * H + h] = att[i];
}

__global__ void k_relation_pack(const float* __restrict__ ratt, const float* __restrict__ rmsg, const float* __restrict__ rpri,
                                int R, int H, int dk, int dkp, float* __restrict__ attT, float* __restrict__ msgP) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)R * H * dkp * dkp;
    if (i >= total) return;
    const int b = (int)(i % dkp), a = (int)((i / dkp) % dkp);
    const int64_t rh = i / ((int64_t)dkp * dkp);
    float va = 0.0f, vm = 0.0f;
    if (a < dk && b < dk) {
        // attT[r][h][c=a][k=b] = att[r][h][k=b][c=a] * pri / sqrt(dk)
        va = ratt[(rh * dk + b) * dk + a] * rpri[rh] / sqrtf((float)dk);
        vm = rmsg[(rh * dk + a) * dk + b];
    }
    attT[i] = va;
    msgP[i] = vm;
}

template <template <int, int> class Launcher, typename... Args>
int dispatch_layout(int vec, int lph, Args... args) {
#define HGT_CASE(V, L) \
    if (vec == V && lph == L) return Launcher<V, L>::run(args...);
    HGT_CASE(1, 4) HGT_CASE(2, 4) HGT_CASE(4, 4) HGT_CASE(8, 4)
    HGT_CASE(1, 8) HGT_CASE(2, 8) HGT_CASE(4, 8) HGT_CASE(8, 8)
    HGT_CASE(1, 16) HGT_CASE(2, 16) HGT_CASE(4, 16) HGT_CASE(8, 16)
    HGT_CASE(1, 32) HGT_CASE(2, 32) HGT_CASE(4, 32) HGT_CASE(8, 32)
    HGT_CASE(1, 64) HGT_CASE(2, 64) HGT_CASE(4, 64) HGT_CASE(8, 64)
#undef HGT_CASE
    return HGT_ERR_UNSUPPORTED;
}

template <int VEC, int LPH>
struct LaunchLogits {
    static int run(const HgtPlanView& pv, const float* Q, const float* K, const float* rteK, const float* attT, float* logits,
                   int R, int HT, hipStream_t stream) {
        const unsigned blocks = (unsigned)((pv.L.max_items + 3) / 4);
        dim3 grid(blocks, (unsigned)(HT / (64 / LPH)));
        if (rteK)
            k_edge_logits<VEC, LPH, true><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attT, logits, R, HT);
        else
            k_edge_logits<VEC, LPH, false><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attT, logits, R, HT);
        return HGT_OK;
    }
};

template <int VEC, int LPH>
struct LaunchAggregate {
    static int run(const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP, float* agg,
                   int R, int64_t NQ, int apply_gelu, int HT, HubBuffers hb, hipStream_t stream) {
        const int64_t tiles = (NQ + 63) / 64;
        const unsigned ny = (unsigned)(HT / (64 / LPH));
        dim3 grid((unsigned)tiles, ny);
        const int32_t* hub_slot = hb.mx ? pv.hub_slot : nullptr;
        if (rteV)
            k_edge_aggregate<VEC, LPH, true><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgP, agg, R,
                                                                       NQ, apply_gelu, HT, hub_slot);
        else
            k_edge_aggregate<VEC, LPH, false><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgP, agg, R,
                                                                        NQ, apply_gelu, HT, hub_slot);
        if (hb.mx) {   // hub path: a fixed grid, every wave returns at once when the plan has no hub
            const int dkp = VEC * LPH;
            const int64_t cells = (int64_t)pv.L.max_hubs * (2 * HT + HT * dkp);
            k_hub_init<<<(unsigned)((cells + 255) / 256), 256, 0, stream>>>(pv.hdr, hb, HT, HT * dkp);
            k_hub_max<<<HUB_GRID_WAVES / 4, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, logits, R, HT, hb);
            dim3 hgrid(HUB_GRID_WAVES / 4, ny);
            if (rteV)
                k_hub_accumulate<VEC, LPH, true><<<hgrid, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, pv.esrc, pv.ertei, logits, V,
                                                                            rteV, msgP, R, HT, hb);
            else
                k_hub_accumulate<VEC, LPH, false><<<hgrid, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, pv.esrc, pv.ertei, logits, V,
                                                                             rteV, msgP, R, HT, hb);
            k_hub_finalize<<<256, 256, 0, stream>>>(pv.hdr, pv.hub_list, hb, agg, HT, dkp, NQ, apply_gelu);
        }
        return HGT_OK;
    }
};

// Head-group split: the smallest power of two that makes the per-lane relation fragment (dk_pad * vec / split floats)
// fit in 128 registers; 1 for every layout up to d = 256 / 8 heads.  Measured at c2 (d=256): a split of 2 is slower
// (logits 2.86 vs 2.75 ms, aggregate 4.19 vs 3.42 ms), so it is only used when the fragment cannot be hoisted
// (d = 512 / d_k = 64: 37 ms -> see DESIGN.md).
static int head_split_for(int vec_full, int lph_full, int dk_pad) {
    int s = 1;
    while (dk_pad * (vec_full / s) > 128 && (vec_full / s) > 1 && lph_full * s * 2 <= 64) s *= 2;
    return s;
}

}  // namespace

extern "C" int hgt_relation_pack(const float* relation_att, const float* relation_msg, const float* relation_pri,
                                 int32_t R, int32_t H, int32_t d_k, int32_t dk_pad, float* att_t, float* msg_p, void* stream) {
    if (!relation_att || !relation_msg || !relation_pri || !att_t || !msg_p || R <= 0 || H <= 0 || d_k <= 0 || dk_pad < d_k)
        return HGT_ERR_INVALID_ARG;
    const int64_t total = (int64_t)R * H * dk_pad * dk_pad;
    k_relation_pack<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(relation_att, relation_msg, relation_pri, R, H,
                                                                                      d_k, dk_pad, att_t, msg_p);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_logits(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                               const float* Q, const float* K, const float* rte_k, const float* att_t, float* logits, void* stream) {
    if (!plan || !Q || !K || !att_t || !logits || H <= 0 || 64 % H != 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    const int sp = head_split_for(dk_pad / lph, lph, dk_pad);
    int rc = dispatch_layout<LaunchLogits>(dk_pad / lph / sp, lph * sp, pv, Q, K, rte_k, att_t, logits, (int)R, (int)H, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
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

extern "C" int hgt_hub_workspace_bytes(int64_t n_edges, int32_t n_heads, int32_t dk_pad, uint64_t* out) {
    if (!out || n_edges < 0 || n_heads <= 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const uint64_t max_hubs = (uint64_t)(n_edges / HGT_HUB_DEG + 1);
    *out = hgt_align_up(max_hubs * n_heads * 4, 256) * 2 + hgt_align_up(max_hubs * n_heads * dk_pad * 4, 256);
    return HGT_OK;
}

extern "C" int hgt_edge_aggregate(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                  const float* logits, const float* V, const float* rte_v, const float* msg_p, float* agg,
                                  int64_t n_q_rows, int32_t apply_gelu, void* hub_ws, void* stream) {
    if (!plan || !V || !msg_p || !agg || (E > 0 && !logits) || H <= 0 || 64 % H != 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    HubBuffers hb = {nullptr, nullptr, nullptr};
    if (hub_ws && E > 0) {
        const uint64_t max_hubs = (uint64_t)pv.L.max_hubs;
        char* b = (char*)hub_ws;
        hb.mx = (int*)b;
        b += hgt_align_up(max_hubs * H * 4, 256);
        hb.l = (float*)b;
        b += hgt_align_up(max_hubs * H * 4, 256);
        hb.acc = (float*)b;
    }
    const int sp = head_split_for(dk_pad / lph, lph, dk_pad);
    int rc = dispatch_layout<LaunchAggregate>(dk_pad / lph / sp, lph * sp, pv, logits, V, rte_v, msg_p, agg, (int)R, NQ, (int)apply_gelu,
                                              (int)H, hb, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_att_export(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, const float* att_sorted,
                              float* att_out, void* stream) {
    if (!plan || !att_sorted || !att_out || H <= 0) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    k_att_export<<<(unsigned)((E * H + 255) / 256), 256, 0, (hipStream_t)stream>>>(pv.eid, att_sorted, att_out, E, H);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
