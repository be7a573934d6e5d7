// Edge phase, matrix-core variant: the per-(target, relation) d_k x d_k transforms of conv.py:98,104 are
// BATCHED over the 16 targets of a sub-tile and run on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains) instead of
// one 128-FMA VALU mat-vec per segment.
//
// Why: on c2 (random relations, 1.75 edges per (target, relation) segment) the VALU mat-vec was ~60 % of the
// instructions of k_edge_logits / k_edge_aggregate and pinned 128 VGPRs per lane for the relation-matrix slice
// (rocprofv3: SQ_ACTIVE_INST_VALU = 39 % / 54 % of the kernels' time, 2 waves per SIMD).  fp32 MFMA has the same
// flop rate as the fp32 VALU, but it is a separate pipe that runs beside other waves' VALU/memory work, the B
// operand is streamed from L2 right when it is needed (registers are free during the edge loops), and the whole
// sub-tile costs 128 MFMAs per relation regardless of how the edges split into segments.
//
// One wavefront = one sub-tile of 16 consecutive targets, all relations ascending (edges of (tile, rel, sub) are
// the contiguous sorted range segptr[(tile, rel, 16*sub)] .. segptr[(tile, rel, 16*sub+16)]).
//   aggregate: per relation, U[16][d] = sum over the segment of exp(s - m_seg) V[src] is built row by row in a
//              wave-private LDS tile, Z = U . M[rel] is ONE batched MFMA per head, and Z is merged into the
//              per-target online-softmax accumulator kept in MFMA C-layout REGISTERS (64 VGPRs at d=256):
//              acc = acc*exp(m_t-m') + Z*exp(m_seg-m').  agg = acc / (l_t + 1e-16) at the end (PyG softmax).
//   logits:    Q[16][d] sits in registers in MFMA A-layout, Qt = Q . A'[rel]^T is one batched MFMA per head,
//              written to a wave-private LDS tile; per edge: one q~ row from LDS, one gathered K row, one dot.
// Relation matrices come in MFMA B-fragment order (hgt_relation_frag) so a fragment is one coalesced 1 KB load.
#include "hgt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SUB = 16;            // targets per wavefront
constexpr float NEG = -1.0e30f;

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int LPH>
__device__ __forceinline__ float head_sum(float v) {
    if (LPH >= 2) v += dppf<0xB1>(v);
    if (LPH >= 4) v += dppf<0x4E>(v);
    if (LPH >= 8) v += dppf<0x141>(v);
    if (LPH >= 16) v += dppf<0x140>(v);
    if (LPH >= 32) v += __shfl_xor(v, 16);
    if (LPH >= 64) v += __shfl_xor(v, 32);
    return v;
}

template <int VEC>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&o)[VEC]) {
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
__device__ __forceinline__ void stv(float* p, const float (&o)[VEC]) {
    if constexpr (VEC == 1) {
        p[0] = o[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
    } else {
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i) *reinterpret_cast<float4*>(p + 4 * i) = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
}

// X [RH][DKP][DKP] row-major (row = contraction index) -> fragment order [RH][ct][kq][lane][4]:
//   frag[...][l][t] = X[kq*16 + (l>>4)*4 + t][ct*16 + (l&15)]
__global__ void k_relation_frag(const float* __restrict__ X, int64_t n_mat, int dkp, float* __restrict__ F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nt = dkp / 16;
    const int64_t per = (int64_t)dkp * dkp;
    if (i >= n_mat * per) return;
    const int64_t m = i / per;
    int r = (int)(i - m * per);
    const int t = r & 3;
    r >>= 2;
    const int l = r & 63;
    r >>= 6;
    const int kq = r % nt, ct = r / nt;
    const int row = kq * 16 + (l >> 4) * 4 + t, col = ct * 16 + (l & 15);
    F[i] = X[m * per + (int64_t)row * dkp + col];
}

// ------------------------------------------------------------------------------------------------
// aggregate
// ------------------------------------------------------------------------------------------------
template <int DKP, int H>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_mfma(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const float* __restrict__ msgF, float* __restrict__ agg, int R, int64_t NQ, int apply_gelu) {
    constexpr int DP = DKP * H, VEC = DP / 64, LPH = 64 / H, NT = DKP / 16, LD = DP + 4;
    constexpr int UN = (VEC <= 4) ? 8 : 4;
    __shared__ __attribute__((aligned(16))) float s_u[4][SUB * LD];        // U tile / final transpose buffer, per wave
    __shared__ __attribute__((aligned(16))) float s_st[4][4][H * SUB];     // per wave: m_t, l_t, (m_seg -> ca), (l_seg -> cb), [h][row]

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int64_t row0 = (int64_t)tile * HGT_TD + wib * SUB;
    if (row0 >= NQ) return;
    const int h = lane / LPH, p = lane % LPH;
    const int fi = lane & 15, fg = lane >> 4;       // MFMA fragment coordinates of this lane
    float* u = s_u[wib];
    float* st_m = s_st[wib][0];
    float* st_l = s_st[wib][1];
    float* st_a = s_st[wib][2];
    float* st_b = s_st[wib][3];

    f32x4 acc[H][NT];                                // C layout: rows 4*fg + i, column h*DKP + ct*16 + fi
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[hh][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    {
        float z[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) z[i] = 0.0f;
#pragma unroll
        for (int r = 0; r < SUB; ++r) stv<VEC>(u + r * LD + lane * VEC, z);
        for (int i = lane; i < H * SUB; i += 64) { st_m[i] = NEG; st_l[i] = 0.0f; st_a[i] = NEG; st_b[i] = 0.0f; }
    }

    for (int rel = 0; rel <= R; ++rel) {
        const int64_t b0 = ((int64_t)tile * (R + 1) + rel) * HGT_TD + wib * SUB;
        const int beg = __builtin_amdgcn_readfirstlane(segptr[b0]);
        const int end = __builtin_amdgcn_readfirstlane(segptr[b0 + SUB]);
        if (beg == end) continue;
        const bool claimed = rel < R;   // bucket R: logit 0, no message (conv.py:68-69)

        // ---- edge loop: online softmax inside each (target, relation) segment, U rows into the LDS tile
        int cur_dst = -1;
        unsigned touched = 0;
        float U[VEC], m_seg = NEG, l_seg = 0.0f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) U[i] = 0.0f;

        auto flush = [&]() {
            if (cur_dst >= 0) {
                const int dl = cur_dst - (int)row0;
                stv<VEC>(u + dl * LD + lane * VEC, U);
                if (p == 0) { st_a[h * SUB + dl] = m_seg; st_b[h * SUB + dl] = l_seg; }
                touched |= 1u << dl;
            }
        };

        for (int base = beg; base < end; base += 64) {
            const int nb = min(64, end - base);
            const int li = base + min(lane, nb - 1);
            const int my_src = esrc[li], my_dst = edst[li];
            const int my_rte = rteV ? (int)ertei[li] : 0;
            for (int i0 = 0; i0 < nb; i0 += UN) {
                float vr[UN][VEC], sl[UN];
                int dsts[UN];
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) {
                    const int idx = min(i0 + uu, nb - 1);
                    const int s = __builtin_amdgcn_readlane(my_src, idx);
                    dsts[uu] = __builtin_amdgcn_readlane(my_dst, idx);
                    if (claimed) {
                        ldv<VEC>(V + (int64_t)s * DP + lane * VEC, vr[uu]);
                        sl[uu] = logits[(int64_t)(base + idx) * H + h];
                        if (rteV) {
                            const int ri = __builtin_amdgcn_readlane(my_rte, idx);
                            float t[VEC];
                            ldv<VEC>(rteV + (int64_t)ri * DP + lane * VEC, t);
#pragma unroll
                            for (int i = 0; i < VEC; ++i) vr[uu][i] += t[i];
                        }
                    } else {
                        sl[uu] = 0.0f;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) vr[uu][i] = 0.0f;
                    }
                }
#pragma unroll
                for (int uu = 0; uu < UN; ++uu) {
                    if (i0 + uu < nb) {
                        if (dsts[uu] != cur_dst) {
                            flush();
#pragma unroll
                            for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
                            m_seg = NEG;
                            l_seg = 0.0f;
                            cur_dst = dsts[uu];
                        }
                        const float m_new = fmaxf(m_seg, sl[uu]);
                        const float sc = __expf(m_seg - m_new), pe = __expf(sl[uu] - m_new);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) U[i] = fmaf(U[i], sc, pe * vr[uu][i]);
                        l_seg = fmaf(l_seg, sc, pe);
                        m_seg = m_new;
                    }
                }
            }
        }
        flush();
        touched = __builtin_amdgcn_readfirstlane(touched);

        // ---- B fragments of M[rel] for every head: all loads in flight at once (the edge-loop registers are dead here)
        f32x4 bf[H][NT][NT];
        if (claimed) {
            const float* __restrict__ mf = msgF + (int64_t)rel * H * DKP * DKP + lane * 4;
#pragma unroll
            for (int hh = 0; hh < H; ++hh)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int kq = 0; kq < NT; ++kq)
                        bf[hh][ct][kq] = *reinterpret_cast<const f32x4*>(mf + ((hh * NT + ct) * NT + kq) * 256);
        }

        // ---- merge factors per (target, head): m' = max(m_t, m_seg), ca = exp(m_t - m'), cb = exp(m_seg - m')
        for (int i = lane; i < H * SUB; i += 64) {
            const float mt = st_m[i], lt = st_l[i], ms = st_a[i], ls = st_b[i];
            const float mn = fmaxf(mt, ms);
            const float ca = __expf(mt - mn), cb = __expf(ms - mn);
            st_m[i] = mn;
            st_l[i] = lt * ca + ls * cb;
            st_a[i] = ca;
            st_b[i] = cb;
        }

        // ---- Z = U . M[rel] per head on the matrix cores, merged into the accumulators
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            const f32x4 ca4 = *reinterpret_cast<const f32x4*>(st_a + hh * SUB + 4 * fg);
            const f32x4 cb4 = *reinterpret_cast<const f32x4*>(st_b + hh * SUB + 4 * fg);
            if (claimed) {
                f32x4 af[NT];
#pragma unroll
                for (int kq = 0; kq < NT; ++kq) af[kq] = *reinterpret_cast<const f32x4*>(u + fi * LD + hh * DKP + kq * 16 + fg * 4);
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kq = 0; kq < NT; ++kq) {
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kq][0], bf[hh][ct][kq][0], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kq][1], bf[hh][ct][kq][1], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kq][2], bf[hh][ct][kq][2], d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kq][3], bf[hh][ct][kq][3], d, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[hh][ct][i] = acc[hh][ct][i] * ca4[i] + d[i] * cb4[i];
                }
            } else {
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[hh][ct][i] *= ca4[i];
            }
        }

        // ---- reset the rows this relation touched
        {
            float z[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) z[i] = 0.0f;
            for (unsigned mk = touched; mk != 0; mk &= mk - 1) {
                const int dl = __builtin_ctz(mk);
                stv<VEC>(u + dl * LD + lane * VEC, z);
            }
            for (int i = lane; i < H * SUB; i += 64) { st_a[i] = NEG; st_b[i] = 0.0f; }
        }
    }

    // ---- write-out: normalise in C layout, transpose through the LDS tile, one coalesced row store per instruction
#pragma unroll
    for (int hh = 0; hh < H; ++hh) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(st_l + hh * SUB + 4 * fg);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) u[(4 * fg + i) * LD + hh * DKP + ct * 16 + fi] = acc[hh][ct][i] / (l4[i] + 1e-16f);
    }
    for (int r = 0; r < SUB; ++r) {
        const int64_t row = row0 + r;
        if (row >= NQ) break;
        float o[VEC];
        ldv<VEC>(u + r * LD + lane * VEC, o);
        if (apply_gelu) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = 0.5f * o[i] * (1.0f + erff(o[i] * 0.70710678118654752440f));
        }
        float* g = agg + row * DP + lane * VEC;
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

template <int DKP, int H>
int launch_aggregate(const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgF, float* agg, int R,
                     int64_t NQ, int apply_gelu, hipStream_t stream) {
    const int64_t tiles = (NQ + HGT_TD - 1) / HGT_TD;
    k_edge_aggregate_mfma<DKP, H><<<(unsigned)tiles, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgF, agg, R,
                                                                     NQ, apply_gelu);
    return HGT_OK;
}

}  // namespace

extern "C" int hgt_relation_frag(const float* att_t, const float* msg_p, int32_t R, int32_t H, int32_t dk_pad, float* att_f, float* msg_f,
                                 void* stream) {
    if (!att_t || !msg_p || !att_f || !msg_f || R <= 0 || H <= 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    if (dk_pad % 16 != 0) return HGT_ERR_UNSUPPORTED;
    const int64_t n_mat = (int64_t)R * H, total = n_mat * dk_pad * dk_pad;
    k_relation_frag<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(att_t, n_mat, dk_pad, att_f);
    k_relation_frag<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(msg_p, n_mat, dk_pad, msg_f);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_aggregate_mfma(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                       const float* logits, const float* V, const float* rte_v, const float* msg_f, float* agg,
                                       int64_t n_q_rows, int32_t apply_gelu, void* stream_) {
    if (!plan || !V || !msg_f || !agg || (E > 0 && !logits) || H <= 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    int rc = HGT_ERR_UNSUPPORTED;
#define HGT_AGG_CASE(D, HH) \
    if (dk_pad == D && H == HH) rc = launch_aggregate<D, HH>(pv, logits, V, rte_v, msg_f, agg, (int)R, NQ, (int)apply_gelu, stream);
    // instantiated where the B fragments of one relation fit in registers (H * (dk_pad/16)^2 * 4 <= 128 VGPRs)
    HGT_AGG_CASE(16, 4) HGT_AGG_CASE(16, 8) HGT_AGG_CASE(16, 16)
    HGT_AGG_CASE(32, 2) HGT_AGG_CASE(32, 4) HGT_AGG_CASE(32, 8)
    HGT_AGG_CASE(64, 1) HGT_AGG_CASE(64, 2)
#undef HGT_AGG_CASE
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
