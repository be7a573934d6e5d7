// Aggregation kernels with the relation transforms on the vector ALU (round-1 design, kept for the exact-fp32 precision,
// for layouts the matrix-core kernel does not cover and for A/B runs: HGT_FLAG_VALU_AGGREGATE).
#include "hgt_edge_common.h"
#include "hgt_fused_update.h"

namespace {

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

template <int VEC, int LPH, bool RTE>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_update(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgP, float* __restrict__ agg, int R, int64_t NQ, int HT,
    const int32_t* __restrict__ hub_slot, int32_t* __restrict__ pending, HgtFusedUpdate fu) {
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

template <int VEC, int LPH>
struct LaunchAggregate {
    static int run(const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP, float* agg,
                   int R, int64_t NQ, int apply_gelu, int HT, HgtHubBuffers hb, hipStream_t stream) {
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
            int rc = hgt_launch_hub(VEC, LPH, pv, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, hb, ny, (int64_t)HT * (VEC * LPH), stream);
            if (rc != HGT_OK) return rc;
        }
        return HGT_OK;
    }
};

template <int VEC, int LPH>
struct LaunchAggregateUpdate {
    static int run(const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP, float* agg, int R,
                   int64_t NQ, int HT, HgtHubBuffers hb, int32_t* pending, HgtFusedUpdate fu, hipStream_t stream) {
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
                int rc = hgt_launch_hub(VEC, LPH, pv, logits, V, rteV, msgP, agg, R, NQ, 1, HT, hb, 1u, (int64_t)HT * (VEC * LPH), stream);
                if (rc != HGT_OK) return rc;
                k_update_pending<VEC><<<grid, 256, 0, stream>>>(agg, (int64_t)HT * (VEC * LPH), NQ, pending, fu);
            }
            return HGT_OK;
        } else {
            return HGT_ERR_UNSUPPORTED;
        }
    }
};

}  // namespace

int hgt_valu_aggregate(const HgtPlanView& pv, int dk_pad, const float* logits, const float* V, const float* rteV, const float* msgP,
                       float* agg, int R, int64_t NQ, int apply_gelu, int H, HgtHubBuffers hb, hipStream_t stream) {
    const int lph = 64 / H;
    const int sp = head_split_for(dk_pad / lph, lph, dk_pad);
    return dispatch_layout<LaunchAggregate>(dk_pad / lph / sp, lph * sp, pv, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, H, hb, stream);
}

int hgt_valu_aggregate_update(const HgtPlanView& pv, int dk_pad, const float* logits, const float* V, const float* rteV,
                              const float* msgP, float* agg, int R, int64_t NQ, int H, HgtHubBuffers hb, int32_t* pending,
                              HgtFusedUpdate fu, hipStream_t stream) {
    const int lph = 64 / H;
    if (head_split_for(dk_pad / lph, lph, dk_pad) != 1) return HGT_ERR_UNSUPPORTED;
    return dispatch_layout<LaunchAggregateUpdate>(dk_pad / lph, lph, pv, logits, V, rteV, msgP, agg, R, NQ, H, hb, pending, fu, stream);
}
