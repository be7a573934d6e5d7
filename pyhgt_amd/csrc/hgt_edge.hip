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
#include "hgt_split_common.h"

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
        // d_k = 64 (16 bounce reads): without a fence hipcc issues all of them up front -- 64 live registers on top of the
        // 128-register fragment push the aggregation kernel past 256 and to one wave per SIMD
        if constexpr (DKP > 32) {
            if ((j4 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
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
    // With temporal encoding the table rows get their own slots (added at use), so the batch is 3/4 as deep (full depth needs 270 registers).
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH, UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>();
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

// FUSE: instead of writing agg rows, the 16 finished rows (normalised, through gelu) stay in registers (`rowvals`) for
// the fused a_linear + node-update epilogue of k_edge_aggregate_update.
template <int VEC, int LPH, bool RTE, bool HUBS, bool FUSE = false>
__device__ __forceinline__ void aggregate_subtile(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgP, float* __restrict__ agg, int R, int64_t NQ, int apply_gelu,
    int HT, unsigned hub_mask, float (&s_acc)[4][16 * 64 * VEC], float (&s_bounce)[4][64 * VEC + 4 * (64 / LPH)], float (&s_ml)[4][2][16 * 16],
    float (&rowvals)[FUSE ? 16 : 1][VEC], int sub_rt = HGT_SUB) {
    // targets per wavefront: HGT_SUB, or fewer (a divisor of it) on small graphs so that more wavefronts exist
    const int SUBR = FUSE ? HGT_SUB : sub_rt;
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH, UN = RTE ? unroll_for<VEC>() / 2 : unroll_for<VEC>();
    const int hg = blockIdx.y;              // head group (see k_edge_logits)
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP;
    constexpr bool HOIST = (DKP * VEC <= 128);

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a workgroup covers 64 consecutive targets (4 waves x 16); the plan's destination tile may be larger
    const int64_t row0 = (int64_t)blockIdx.x * (4 * SUBR) + wib * SUBR;
    if (row0 >= NQ) {
        if constexpr (FUSE) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < VEC; ++i) rowvals[r][i] = 0.0f;
        }
        return;
    }
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

    // hub-free sub-tiles: the edge range of every relation bucket is read up front (lane r = bucket r), so that empty
    // relations cost a v_readlane instead of a dependent load (c5: 33 relations, most of them empty for a given target type)
    const bool ranges_ready = !HUBS && R < 64;
    int my_beg = 0, my_end = 0;
    if (ranges_ready) {
        const int64_t bb = ((int64_t)tile * (R + 1) + min(lane, R)) * HGT_TD + within;
        my_beg = segptr[bb];
        my_end = segptr[bb + SUBR];
    }
    for (int rel = 0; rel <= R; ++rel) {
      const int64_t b0 = ((int64_t)tile * (R + 1) + rel) * HGT_TD + within;
      // maximal runs [dl0, dl1) of non-hub targets: one run covering the whole sub-tile unless it contains a hub
      for (int dl0 = 0; dl0 < SUBR;) {
        int dl1 = SUBR;
        if constexpr (HUBS) {
            if ((hub_mask >> dl0) & 1u) { ++dl0; continue; }
            dl1 = dl0 + 1;
            while (dl1 < SUBR && !((hub_mask >> dl1) & 1u)) ++dl1;
        }
        const int beg = ranges_ready ? __builtin_amdgcn_readlane(my_beg, rel) : __builtin_amdgcn_readfirstlane(segptr[b0 + dl0]);
        const int end = ranges_ready ? __builtin_amdgcn_readlane(my_end, rel) : __builtin_amdgcn_readfirstlane(segptr[b0 + dl1]);
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

    if constexpr (FUSE) {
#pragma unroll
        for (int r = 0; r < HGT_SUB; ++r) {
            // rows beyond NQ (last tile only) have an all-zero accumulator: gelu(0) = 0, no branch needed
            const float inv = 1.0f / (s_l[r * 16 + h] + 1e-16f);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float o = acc[r * DP + i * 64 + lane] * inv;
                rowvals[r][i] = 0.5f * o * (1.0f + erff(o * 0.70710678118654752440f));   // conv.py:119
            }
        }
        return;
    }
    // write-out: normalise, un-permute the planar layout, one coalesced row store per wave instruction
    for (int r = 0; r < SUBR; ++r) {
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
__global__ __launch_bounds__(256, 2) void k_edge_aggregate(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgP, float* __restrict__ agg, int R, int64_t NQ, int apply_gelu,
    int HT, const int32_t* __restrict__ hub_slot, int sub) {
    __shared__ __attribute__((aligned(16))) float s_acc[4][16 * 64 * VEC];
    __shared__ __attribute__((aligned(16))) float s_bounce[4][64 * VEC + 4 * (64 / LPH)];
    __shared__ float s_ml[4][2][16 * 16];   // running max / sum per (target, head); H <= 16
    // hub targets (in-degree > HGT_HUB_DEG, plan) are aggregated by the hub kernels below; this wave skips them
    unsigned hub_mask = 0;
    if (hub_slot) {
        const int lane = threadIdx.x & 63;
        const int64_t rr = (int64_t)blockIdx.x * (4 * sub) + (threadIdx.x >> 6) * sub + (lane & 15);
        const bool is_hub = (lane < sub) && (rr < NQ) && (hub_slot[rr] >= 0);
        hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    }
    float no_rowvals[1][VEC];
    if (hub_mask == 0)
        aggregate_subtile<VEC, LPH, RTE, false>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, 0u, s_acc,
                                                s_bounce, s_ml, no_rowvals, sub);
    else
        aggregate_subtile<VEC, LPH, RTE, true>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, hub_mask,
                                               s_acc, s_bounce, s_ml, no_rowvals, sub);
}

// ---------------------------------------------------------------------------------------------
// Pass 2 with the node update fused in (HGTConv, split-bf16 precision, d_pad <= 256, plan without hubs):
//     out[i] = LN_t( (gelu(agg_i) W_a[t]^T + b_a[t]) * sigmoid(skip[t]) + x_i * (1 - sigmoid(skip[t])) )   conv.py:119-133
// The workgroup that aggregated 64 targets already holds their finished rows; writing them to HBM only for a second
// kernel to read them back costs 2 x 4d bytes per node and a kernel that is latency bound on its own (1.2 ms at c2).
// Here the rows go registers -> LDS as the bf16 hi/mid A operand (the 66 KB slab overlays the accumulator, which is
// dead by then), the four wavefronts run the 64 x d x d split-bf16 MFMA product against the L2-resident fragment-ordered
// W_a (hgt_split_weights), and the gated skip + LayerNorm epilogue writes `out` directly.  agg never touches HBM --
// which is what the minimal-traffic model of SURVEY.md 8(d) assumes.
// A tile whose rows have several node types (only at the T-1 type boundaries of a type-sorted graph) repeats the
// product per type present; rows of unknown type are written as 0 (conv.py:120).
// ---------------------------------------------------------------------------------------------

struct FusedUpdate {
    const int64_t* node_type;
    const unsigned short* w_split;   // hgt_split_weights(W_a): [T][1][n_kc][2][8][64][8] bf16
    const float* bias;               // [T][n_out]
    const float* xs;                 // skip input rows [*][ldxs]
    int64_t ldxs;
    const float* skip;               // [T]
    const float* lnw;                // [T][n_out] or nullptr
    const float* lnb;
    int use_norm, n_types, n_out;
    float* out;                      // [NQ][n_out]
};

// The epilogue proper: `vals` = this wavefront's 16 finished rows (gelu applied), `slab` = >= 2*A_PLANE bytes of LDS
// every wavefront is done with, `tables` = 2.5 KB of LDS for the row types and the LayerNorm partial sums.
template <int VEC>
__device__ __forceinline__ void fused_update_epilogue(const float (&vals)[16][VEC], unsigned char* slab, unsigned char* tables,
                                                      int64_t row0, int64_t NQ, const FusedUpdate& fu) {
    constexpr int DP = 64 * VEC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- A operand: hi/mid bf16 planes of the 64 finished rows, [plane][row][k], 528 B row stride
    int* s_type = reinterpret_cast<int*>(tables);                // [64]
    float* s_sum = reinterpret_cast<float*>(tables + 256);         // [64][4]
    float* s_var = s_sum + 256;                                    // [64][4]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned char* prow = slab + (wave * 16 + r) * A_STRIDE + lane * VEC * 2;
        unsigned short hi[VEC], mid[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            hi[i] = bf16_rne(vals[r][i]);
            mid[i] = bf16_rne(vals[r][i] - bf16_to_f32(hi[i]));
        }
        if constexpr (VEC == 1) {
            *reinterpret_cast<unsigned short*>(prow) = hi[0];
            *reinterpret_cast<unsigned short*>(prow + A_PLANE) = mid[0];
        } else if constexpr (VEC == 2) {
            *reinterpret_cast<unsigned*>(prow) = (unsigned)hi[0] | ((unsigned)hi[1] << 16);
            *reinterpret_cast<unsigned*>(prow + A_PLANE) = (unsigned)mid[0] | ((unsigned)mid[1] << 16);
        } else {
            *reinterpret_cast<uint2*>(prow) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
            *reinterpret_cast<uint2*>(prow + A_PLANE) =
                make_uint2((unsigned)mid[0] | ((unsigned)mid[1] << 16), (unsigned)mid[2] | ((unsigned)mid[3] << 16));
        }
    }
    if (tid < 64) {
        const int64_t row = row0 + tid;
        int64_t t = (row < NQ) ? fu.node_type[row] : -1;
        s_type[tid] = (t >= 0 && t < fu.n_types) ? (int)t : -1;
    }
    __syncthreads();

    const int my_t = s_type[lane];
    int tmin = my_t < 0 ? 0x7fffffff : my_t, tmax = my_t;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        tmin = min(tmin, __shfl_xor(tmin, sft));
        tmax = max(tmax, __shfl_xor(tmax, sft));
    }
    tmin = __builtin_amdgcn_readfirstlane(tmin);
    tmax = __builtin_amdgcn_readfirstlane(tmax);

    constexpr int NKC = DP / KC;                       // k-chunks (a multiple of 4, like split_dims)
    const int n_out = fu.n_out;
    const int frow = lane & 31, khalf = lane >> 5;
    const unsigned char* abase = slab + frow * A_STRIDE + khalf * 16;
    const bool o1 = lane & 1, o2 = lane & 2;
    const float inv_n = 1.0f / (float)n_out;
    const int rt0 = (lane & 3) + 4 * (lane >> 5);      // row of register group (j, q): rt0 + 32 j + 8 q

    for (int g = tmin; g <= tmax; ++g) {               // empty range when no row has a valid type
        if (__builtin_amdgcn_ballot_w64(my_t == g) == 0) continue;
        // ---- 64 x n_out x DP product; this wavefront owns columns [64 wave, 64 wave + 64) = column tiles 2 wave, 2 wave + 1
        f32x16 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.0f;
        if (wave * 64 < n_out) {
            const unsigned short* __restrict__ wf = fu.w_split + (int64_t)g * NKC * 2 * W_PLANE_ELEMS + ((2 * wave) * 64 + lane) * 8;
            // B fragments of two column tiles, NSTG k-chunks ahead in named register stages; A fragments one chunk ahead.
            // The sched barriers pin "next A -> 12 MFMAs -> refill of the consumed stage" (see k_typed_linear_pc: left alone,
            // hipcc sinks the loads next to their uses and every wait becomes a wait for a load that was just issued).
            bf16x8 s0h0, s0h1, s0m0, s0m1, s1h0, s1h1, s1m0, s1m1, s2h0, s2h1, s2m0, s2m1, s3h0, s3h1, s3m0, s3m1;
            bf16x8 e_h0, e_m0, e_h1, e_m1, o_h0, o_m0, o_h1, o_m1;
#define FU_LOAD_B(S, KCI)                                                                            \
    {                                                                                                \
        const unsigned short* t_ = wf + (int64_t)min((KCI), NKC - 1) * 2 * W_PLANE_ELEMS;            \
        s##S##h0 = *reinterpret_cast<const bf16x8*>(t_);                                             \
        s##S##h1 = *reinterpret_cast<const bf16x8*>(t_ + 64 * 8);                                    \
        s##S##m0 = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                             \
        s##S##m1 = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS + 64 * 8);                    \
    }
#define FU_LOAD_A(P, KCI)                                                                            \
    {                                                                                                \
        const unsigned char* a_ = abase + min((KCI), NKC - 1) * (KC * 2);                            \
        P##_h0 = *reinterpret_cast<const bf16x8*>(a_);                                               \
        P##_m0 = *reinterpret_cast<const bf16x8*>(a_ + A_PLANE);                                     \
        P##_h1 = *reinterpret_cast<const bf16x8*>(a_ + 32 * A_STRIDE);                               \
        P##_m1 = *reinterpret_cast<const bf16x8*>(a_ + A_PLANE + 32 * A_STRIDE);                     \
    }
#define FU_STEP(S, KCI, P, PN)                                                                                     \
    {                                                                                                              \
        FU_LOAD_A(PN, (KCI) + 1)                                                                                   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_m0, s##S##h0, acc[0][0], 0, 0, 0);                 \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_m1, s##S##h0, acc[0][1], 0, 0, 0);                 \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_m0, s##S##h1, acc[1][0], 0, 0, 0);                 \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_m1, s##S##h1, acc[1][1], 0, 0, 0);                 \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h0, s##S##m0, acc[0][0], 0, 0, 0);                 \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h1, s##S##m0, acc[0][1], 0, 0, 0);                 \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h0, s##S##m1, acc[1][0], 0, 0, 0);                 \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h1, s##S##m1, acc[1][1], 0, 0, 0);                 \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h0, s##S##h0, acc[0][0], 0, 0, 0);                 \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h1, s##S##h0, acc[0][1], 0, 0, 0);                 \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h0, s##S##h1, acc[1][0], 0, 0, 0);                 \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P##_h1, s##S##h1, acc[1][1], 0, 0, 0);                 \
        FU_LOAD_B(S, (KCI) + 4)                                                                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  /* 4 DS reads   */                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); /* 12 MFMAs     */                                     \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);  /* 4 VMEM reads */                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
            FU_LOAD_B(0, 0)
            FU_LOAD_B(1, 1)
            FU_LOAD_B(2, 2)
            FU_LOAD_B(3, 3)
            FU_LOAD_A(e, 0)
            for (int kq = 0; kq < NKC; kq += 4) {
                FU_STEP(0, kq, e, o)
                FU_STEP(1, kq + 1, o, e)
                FU_STEP(2, kq + 2, e, o)
                FU_STEP(3, kq + 3, o, e)
            }
#undef FU_STEP
#undef FU_LOAD_A
#undef FU_LOAD_B
        }
        // ---- epilogue for the rows of type g: bias, gated skip, LayerNorm, store
        const float alpha = 1.0f / (1.0f + expf(-fu.skip[g]));
        float y[16][4];                                // [c*8 + j*4 + q][4 consecutive columns]
        int orow[8];                                   // row of group (j, q); -1 = not a row of this type
#pragma unroll
        for (int jq = 0; jq < 8; ++jq) {
            const int rt = rt0 + 32 * (jq >> 2) + 8 * (jq & 3);
            orow[jq] = (s_type[rt] == g) ? rt : -1;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = wave * 64 + c * 32 + ((lane & 31) >> 2) * 4;
            const bool col_ok = col < n_out;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok) b4 = *reinterpret_cast<const float4*>(fu.bias + (int64_t)g * n_out + col);
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                const int j = jq >> 2, q = jq & 3;
                float v0 = acc[c][j][4 * q], v1 = acc[c][j][4 * q + 1], v2 = acc[c][j][4 * q + 2], v3 = acc[c][j][4 * q + 3];
                quad_transpose(v0, v1, v2, v3, o1, o2);
                // (requesting these rows before the workgroup barrier was measured slower: the loads only queue behind the
                // gathers of the workgroup sharing the CU, and 64 more live registers spill)
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col_ok && orow[jq] >= 0) xv = *reinterpret_cast<const float4*>(fu.xs + (row0 + orow[jq]) * fu.ldxs + col);
                y[c * 8 + jq][0] = col_ok ? (v0 + b4.x) * alpha + xv.x * (1.0f - alpha) : 0.0f;
                y[c * 8 + jq][1] = col_ok ? (v1 + b4.y) * alpha + xv.y * (1.0f - alpha) : 0.0f;
                y[c * 8 + jq][2] = col_ok ? (v2 + b4.z) * alpha + xv.z * (1.0f - alpha) : 0.0f;
                y[c * 8 + jq][3] = col_ok ? (v3 + b4.w) * alpha + xv.w * (1.0f - alpha) : 0.0f;
            }
        }
        if (fu.use_norm) {
            // a row's columns live in 4 wavefronts x 2 column tiles x 8 lanes: lane-strided sums, one table entry per (row, wave)
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                const int rt = rt0 + 32 * (jq >> 2) + 8 * (jq & 3);
                float ps = y[jq][0] + y[jq][1] + y[jq][2] + y[jq][3] + y[8 + jq][0] + y[8 + jq][1] + y[8 + jq][2] + y[8 + jq][3];
                ps = strided8_sum(ps);
                if (((lane & 31) >> 2) == 0) s_sum[rt * 4 + wave] = ps;
            }
            __syncthreads();
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                const int rt = rt0 + 32 * (jq >> 2) + 8 * (jq & 3);
                const float4 a4 = *reinterpret_cast<const float4*>(&s_sum[rt * 4]);
                const float mean = (a4.x + a4.y + a4.z + a4.w) * inv_n;
                float ps = 0.0f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool col_ok = wave * 64 + c * 32 + ((lane & 31) >> 2) * 4 < n_out;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[c * 8 + jq][e] -= mean;                                  // centred from here on
                        if (col_ok) ps += y[c * 8 + jq][e] * y[c * 8 + jq][e];
                    }
                }
                ps = strided8_sum(ps);
                if (((lane & 31) >> 2) == 0) s_var[rt * 4 + wave] = ps;
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = wave * 64 + c * 32 + ((lane & 31) >> 2) * 4;
            const bool col_ok = col < n_out;
            float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fu.use_norm && col_ok) {
                w4 = *reinterpret_cast<const float4*>(fu.lnw + (int64_t)g * n_out + col);
                c4 = *reinterpret_cast<const float4*>(fu.lnb + (int64_t)g * n_out + col);
            }
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                if (!col_ok || orow[jq] < 0) continue;
                float rstd = 1.0f;
                if (fu.use_norm) {
                    const float4 a4 = *reinterpret_cast<const float4*>(&s_var[orow[jq] * 4]);
                    rstd = rsqrtf((a4.x + a4.y + a4.z + a4.w) * inv_n + 1e-5f);
                }
                const float* yy = y[c * 8 + jq];
                *reinterpret_cast<float4*>(fu.out + (row0 + orow[jq]) * n_out + col) =
                    make_float4(yy[0] * rstd * w4.x + c4.x, yy[1] * rstd * w4.y + c4.y, yy[2] * rstd * w4.z + c4.z, yy[3] * rstd * w4.w + c4.w);
            }
        }
        if (fu.use_norm) __syncthreads();   // the tables are rewritten by the next type of a mixed tile
    }
    // rows of unknown type -> 0 (conv.py:120)
    for (int r = wave * 16; r < wave * 16 + 16; ++r) {
        if (row0 + r < NQ && s_type[r] < 0) {
            for (int cidx = lane; cidx < n_out; cidx += 64) fu.out[(row0 + r) * n_out + cidx] = 0.0f;
        }
    }
}

// Workgroups that contain a hub target cannot finish their rows here (the hub kernels write those rows of agg later):
// they take the unfused path, raise pending[workgroup], and k_update_pending runs the same epilogue from agg afterwards.
template <int VEC, int LPH, bool RTE>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_update(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgP, float* __restrict__ agg, int R, int64_t NQ, int HT,
    const int32_t* __restrict__ hub_slot, int32_t* __restrict__ pending, FusedUpdate fu) {
    constexpr int DP = 64 * VEC;
    static_assert(DP <= KP, "the fused epilogue keeps the whole K extent in one LDS slab");
    constexpr int AGG_PART = 4 * 16 * 64 * VEC * 4 + 4 * (64 * VEC + 4 * (64 / LPH)) * 4;
    constexpr int FRONT = AGG_PART > 2 * A_PLANE ? AGG_PART : 2 * A_PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[FRONT + 4 * 2 * 256 * 4];
    auto& s_acc = *reinterpret_cast<float(*)[4][16 * 64 * VEC]>(smem);
    auto& s_bounce = *reinterpret_cast<float(*)[4][64 * VEC + 4 * (64 / LPH)]>(smem + 4 * 16 * 64 * VEC * 4);
    auto& s_ml = *reinterpret_cast<float(*)[4][2][16 * 16]>(smem + FRONT);
    const int64_t row0 = (int64_t)blockIdx.x * 64;

    unsigned hub_mask = 0;
    if (hub_slot) {
        const int lane = threadIdx.x & 63;
        const int64_t rr = row0 + (threadIdx.x >> 6) * 16 + (lane & 15);
        const bool is_hub = (lane < 16) && (rr < NQ) && (hub_slot[rr] >= 0);
        hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    }
    const bool any_hub = hub_slot ? (__syncthreads_or(hub_mask != 0) != 0) : false;
    if (threadIdx.x == 0) pending[blockIdx.x] = any_hub ? 1 : 0;
    if (any_hub) {
        float no_rowvals[1][VEC];
        if (hub_mask == 0)
            aggregate_subtile<VEC, LPH, RTE, false>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, agg, R, NQ, 1, HT, 0u, s_acc,
                                                    s_bounce, s_ml, no_rowvals);
        else
            aggregate_subtile<VEC, LPH, RTE, true>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, agg, R, NQ, 1, HT, hub_mask, s_acc,
                                                   s_bounce, s_ml, no_rowvals);
        return;
    }
    float vals[16][VEC];
    aggregate_subtile<VEC, LPH, RTE, false, true>(segptr, esrc, edst, ertei, logits, V, rteV, msgP, nullptr, R, NQ, 1, HT, 0u, s_acc,
                                                  s_bounce, s_ml, vals);
    __syncthreads();   // every wavefront is done with the accumulators, bounce rows and softmax state
    fused_update_epilogue<VEC>(vals, smem, smem + FRONT, row0, NQ, fu);
}

// the node update of the workgroups k_edge_aggregate_update left pending (their agg rows are complete by now)
template <int VEC>
__global__ __launch_bounds__(256, 2) void k_update_pending(const float* __restrict__ agg, int64_t ld_agg, int64_t NQ,
                                                           const int32_t* __restrict__ pending, FusedUpdate fu) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_PLANE + 4096];
    if (pending[blockIdx.x] == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    float vals[16][VEC];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 16 + r;
#pragma unroll
        for (int i = 0; i < VEC; ++i) vals[r][i] = 0.0f;
        if (row < NQ) load_vec<VEC>(agg + row * ld_agg + lane * VEC, vals[r]);
    }
    fused_update_epilogue<VEC>(vals, smem, smem + 2 * A_PLANE, row0, NQ, fu);
}

// ---------------------------------------------------------------------------------------------
// Hub path.  A target with more than HGT_HUB_DEG in-edges would make the single wavefront that owns its sub-tile
// walk all of them (Zipf targets: 8 -> 190 ms at c2 size).  Hubs are therefore skipped by k_edge_aggregate and
// handled by a fixed grid of wavefronts that split every (hub, relation) edge range into HUB_CHUNKS pieces:
//   k_hub_max        max logit per (hub, head)  (wave reduce + one atomicMax per head and piece)
//   k_hub_accumulate sum exp(s - m) and (sum exp(s - m) V[src]) M[rel] per piece, atomically added to the hub's
//                    fp32 accumulators, m = the hub's true max logit per head (unclaimed edges count with logit 0)
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
        // true max over ALL in-edges of the hub (k_hub_max folds a 0 in for a non-empty unclaimed bucket), like PyG's softmax
        const float mref = ord2f(hb.mx[slot * HT + hg * H + h]);
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
        // small graphs (the reference's sampled subgraphs): 4 instead of 16 targets per wavefront -> 4x the wavefronts, each
        // with a quarter of the serial edge walk (c3 surrogate, N = 2.5k: 150 -> 60 us)
        const int sub = (NQ < 65536) ? 4 : HGT_SUB;
        const int64_t tiles = (NQ + 4 * sub - 1) / (4 * sub);
        const unsigned ny = (unsigned)(HT / (64 / LPH));
        dim3 grid((unsigned)tiles, ny);
        const int32_t* hub_slot = hb.mx ? pv.hub_slot : nullptr;
        if (rteV)
            k_edge_aggregate<VEC, LPH, true><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgP, agg, R,
                                                                       NQ, apply_gelu, HT, hub_slot, sub);
        else
            k_edge_aggregate<VEC, LPH, false><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgP, agg, R,
                                                                        NQ, apply_gelu, HT, hub_slot, sub);
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

template <int VEC, int LPH>
struct LaunchAggregateUpdate {
    static int run(const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP, float* agg, int R,
                   int64_t NQ, int HT, HubBuffers hb, int32_t* pending, FusedUpdate fu, hipStream_t stream) {
        if constexpr (64 * VEC <= KP) {
            if (HT != 64 / LPH) return HGT_ERR_UNSUPPORTED;   // a head-group split leaves a workgroup with part of the row
            const int64_t tiles = (NQ + 63) / 64;
            dim3 grid((unsigned)tiles, 1);
            const int32_t* hub_slot = hb.mx ? pv.hub_slot : nullptr;
            if (rteV)
                k_edge_aggregate_update<VEC, LPH, true><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgP,
                                                                                 agg, R, NQ, HT, hub_slot, pending, fu);
            else
                k_edge_aggregate_update<VEC, LPH, false><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV,
                                                                                  msgP, agg, R, NQ, HT, hub_slot, pending, fu);
            if (hb.mx) {   // hub path (see LaunchAggregate) + the update of the workgroups that had to wait for it
                const int dkp = VEC * LPH;
                const int64_t cells = (int64_t)pv.L.max_hubs * (2 * HT + HT * dkp);
                k_hub_init<<<(unsigned)((cells + 255) / 256), 256, 0, stream>>>(pv.hdr, hb, HT, HT * dkp);
                k_hub_max<<<HUB_GRID_WAVES / 4, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, logits, R, HT, hb);
                dim3 hgrid(HUB_GRID_WAVES / 4, 1);
                if (rteV)
                    k_hub_accumulate<VEC, LPH, true><<<hgrid, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, pv.esrc, pv.ertei, logits, V,
                                                                                rteV, msgP, R, HT, hb);
                else
                    k_hub_accumulate<VEC, LPH, false><<<hgrid, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, pv.esrc, pv.ertei, logits, V,
                                                                                 rteV, msgP, R, HT, hb);
                k_hub_finalize<<<256, 256, 0, stream>>>(pv.hdr, pv.hub_list, hb, agg, HT, dkp, NQ, 1);
                k_update_pending<VEC><<<grid, 256, 0, stream>>>(agg, (int64_t)HT * dkp, NQ, pending, fu);
            }
            return HGT_OK;
        } else {
            return HGT_ERR_UNSUPPORTED;
        }
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


extern "C" int hgt_edge_aggregate_update(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                         const float* logits, const float* V, const float* rte_v, const float* msg_p, float* agg,
                                         int64_t n_q_rows, void* hub_ws, int32_t* pending, const int64_t* node_type,
                                         const void* w_a_split, const float* b_a, const float* x_skip, int64_t ld_skip,
                                         const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm, int32_t n_out,
                                         float* out, void* stream) {
    if (!plan || !V || !msg_p || !agg || !pending || !node_type || !w_a_split || !b_a || !x_skip || !skip || !out || H <= 0 ||
        64 % H != 0 || dk_pad <= 0 || n_out <= 0)
        return HGT_ERR_INVALID_ARG;
    if (E > 0 && !logits) return HGT_ERR_INVALID_ARG;
    if (use_norm && (!ln_w || !ln_b)) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    const int dp = H * dk_pad;
    if (dp > KP || n_out > dp || (n_out & 3) != 0 || (ld_skip & 3) != 0 || ((uintptr_t)x_skip & 15) != 0) return HGT_ERR_UNSUPPORTED;
    if (head_split_for(dk_pad / lph, lph, dk_pad) != 1) return HGT_ERR_UNSUPPORTED;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    HubBuffers hb = {nullptr, nullptr, nullptr};
    if (hub_ws && E > 0) {   // same carving as hgt_edge_aggregate
        const uint64_t per = hgt_align_up((uint64_t)pv.L.max_hubs * H * 4, 256);
        hb.mx = (int*)hub_ws;
        hb.l = (float*)((char*)hub_ws + per);
        hb.acc = (float*)((char*)hub_ws + 2 * per);
    }
    FusedUpdate fu = {node_type, (const unsigned short*)w_a_split, b_a, x_skip, ld_skip, skip, ln_w, ln_b, use_norm, T, n_out, out};
    int rc = dispatch_layout<LaunchAggregateUpdate>(dk_pad / lph, lph, pv, logits, V, rte_v, msg_p, agg, (int)R, NQ, (int)H, hb, pending, fu,
                                                    (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
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
