// Pass 1 (attention logits), split-bf16 matrix-core variant for the d_k = 32 / 8-head layout (d = 256).
//
// Same contract and work decomposition as k_edge_logits (hgt_edge.hip): one wavefront = one work item = a run of
// sorted edges of one (destination tile, relation).  What changes is how q~ = A'[rel] q is obtained.  The VALU kernel
// does one 128-FMA mat-vec per (target, relation) segment -- 256 SIMD cycles per segment, 0.68 ms of pure VALU time
// at c2 (5.7 M segments), 39 % of that kernel's instructions.  But EVERY segment of an item shares the relation, so
// the transforms of 16 segments at a time are one small GEMM  Qt[16 x 256] = Q[16 x 256] . blockdiag_h(A'[rel,h]^T)
// on v_mfma_f32_16x16x32_bf16 with the operands split into bf16 hi+mid terms (3 MFMAs per tile, fp32 accumulate,
// product error <= ~3*2^-18): 48 MFMAs (768 matrix-pipe cycles) per 16 segments instead of 4096 VALU cycles, on a
// pipe that is otherwise idle in this kernel.
//   * B operand: A'[rel]^T pre-split to bf16 hi/mid and stored in MFMA-fragment order by hgt_relation_pack_bf16;
//     all 8 heads x 2 column tiles stay in registers for the whole item (128 VGPRs, the same budget as the fp32 slice).
//   * A operand: the 16 targets' Q rows are gathered in fragment shape (8 consecutive floats per lane), split once.
//   * D (MFMA C layout) is written to a wave-private LDS tile [16][260]; an edge loop then reads one q~ row per
//     segment (ds_read_b128) and does gathered-K-row dots exactly like the VALU kernel.
// Used by hgt_conv_forward when precision = split-bf16 and the layout is (d_k_pad = 32, 8 heads); everything else
// (and the exact fp32 mode) keeps the VALU kernel.
#include "hgt_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int H = 8, DKP = 32, DP = 256, LPH = 8, UN = 8, LD = DP + 4;

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float head8_sum(float v) {
    v += dppf<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dppf<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dppf<0x141>(v);   // row_half_mirror
    return v;
}

// attT [R*H][32 (c)][32 (k)] fp32 -> fragments [R*H][ct 2][plane 2][lane 64][8] bf16:
//   frag[..][l][e] = split(attT[c = (l>>4)*8 + e][k = ct*16 + (l&15)])
__global__ void k_pack_att_bf16(const float* __restrict__ attT, int64_t n_mat, unsigned short* __restrict__ F) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mat * 2 * 64 * 8) return;
    int r = (int)(i % (2 * 64 * 8));
    const int64_t m = i / (2 * 64 * 8);
    const int e = r & 7;
    r >>= 3;
    const int l = r & 63, ct = r >> 6;
    const float v = attT[m * 1024 + ((l >> 4) * 8 + e) * 32 + ct * 16 + (l & 15)];
    const unsigned short hi = bf16_rne(v);
    const unsigned short mid = bf16_rne(v - bf16_to_f32(hi));
    F[((m * 2 + ct) * 2 + 0) * 512 + l * 8 + e] = hi;
    F[((m * 2 + ct) * 2 + 1) * 512 + l * 8 + e] = mid;
}

__device__ __forceinline__ void split8(const float4 a, const float4 b, bf16x8& hi, bf16x8& mid) {
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned short hh = bf16_rne(f[i]);
        hi[i] = (short)hh;
        mid[i] = (short)bf16_rne(f[i] - bf16_to_f32(hh));
    }
}

template <bool RTE>
__global__ __launch_bounds__(256, 2) void k_edge_logits_mfma(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ Q,
    const float* __restrict__ K, const float* __restrict__ rteK, const unsigned short* __restrict__ attF,
    float* __restrict__ logits, int R) {
    __shared__ __attribute__((aligned(16))) float s_qt[4][16 * LD];   // q~ tile of the current 16 segments, per wave
    constexpr int BU = RTE ? UN / 2 : UN;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wib;
    if (item >= hdr->n_items) return;
    const HgtItem it = items[item];
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    const int h = lane / LPH, p = lane % LPH;
    const int fi = lane & 15, fg = lane >> 4;

    if (rel >= R) {   // edges no meta relation claims: logit 0 (conv.py:68)
        for (int64_t i = (int64_t)beg * H + lane; i < (int64_t)end * H; i += 64) logits[i] = 0.0f;
        return;
    }

    // B fragments of A'[rel]^T: [h][ct] hi and mid, 16 B per lane each, resident for the whole item
    bf16x8 bh[H][2], bm[H][2];
    {
        const unsigned short* __restrict__ f = attF + (int64_t)rel * H * 2 * 2 * 512 + lane * 8;
#pragma unroll
        for (int hh = 0; hh < H; ++hh)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                bh[hh][ct] = *reinterpret_cast<const bf16x8*>(f + ((hh * 2 + ct) * 2 + 0) * 512);
                bm[hh][ct] = *reinterpret_cast<const bf16x8*>(f + ((hh * 2 + ct) * 2 + 1) * 512);
            }
    }
    float* qt_tile = s_qt[wib];

    for (int base = beg; base < end; base += 64) {
        const int nb = min(64, end - base);
        const int li = base + min(lane, nb - 1);
        const int my_src = esrc[li], my_dst = edst[li];
        const int my_rte = RTE ? (int)ertei[li] : 0;
        // segment starts inside this chunk (lane 0 always starts one: the previous chunk's q~ tile is gone)
        const int prev_dst = __shfl_up(my_dst, 1);
        const bool is_start = (lane < nb) && (lane == 0 || my_dst != prev_dst);
        const unsigned long long smask = __builtin_amdgcn_ballot_w64(is_start);
        const int nseg = __builtin_popcountll(smask);

        for (int g0 = 0; g0 < nseg; g0 += 16) {
            // ---- the (up to) 16 targets of this group: lane -> position of the (g0 + fi)-th segment start
            unsigned long long mk = smask;
            const int want = g0 + fi;
            for (int c = 0; c < want; ++c) mk &= mk - 1;              // drop the `want` lowest set bits
            const bool row_ok = want < nseg;
            const int spos = row_ok ? __builtin_ctzll(mk) : 0;
            const int row_dst = __shfl(my_dst, spos);
            // edge range of the group inside the chunk: [e_lo, e_hi)
            unsigned long long mg = smask;
            for (int c = 0; c < g0; ++c) mg &= mg - 1;
            const int e_lo = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mg));
            unsigned long long mh = mg;
            for (int c = 0; c < 16 && mh; ++c) mh &= mh - 1;
            const int e_hi = __builtin_amdgcn_readfirstlane(mh ? __builtin_ctzll(mh) : nb);

            // ---- first batch of K rows in flight before the matrix phase
            float kr[BU][4], tr[RTE ? BU : 1][4];
            auto issue = [&](int e0) {
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int idx = min(e0 + u, e_hi - 1);
                    const int s = __builtin_amdgcn_readlane(my_src, idx);
                    const float4 t = *reinterpret_cast<const float4*>(K + (int64_t)s * DP + lane * 4);
                    kr[u][0] = t.x; kr[u][1] = t.y; kr[u][2] = t.z; kr[u][3] = t.w;
                    if constexpr (RTE) {
                        const int ri = __builtin_amdgcn_readlane(my_rte, idx);
                        const float4 w = *reinterpret_cast<const float4*>(rteK + (int64_t)ri * DP + lane * 4);
                        tr[u][0] = w.x; tr[u][1] = w.y; tr[u][2] = w.z; tr[u][3] = w.w;
                    }
                }
            };
            issue(e_lo);

            // ---- Qt tile = Q[16 targets] . A'[rel]^T, head by head on the matrix cores (split-bf16, 3 MFMAs per tile)
#pragma unroll
            for (int hh = 0; hh < H; ++hh) {
                float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
                if (row_ok) {
                    const float* qp = Q + (int64_t)row_dst * DP + hh * DKP + fg * 8;
                    qa = *reinterpret_cast<const float4*>(qp);
                    qb = *reinterpret_cast<const float4*>(qp + 4);
                }
                bf16x8 ah, am;
                split8(qa, qb, ah, am);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[hh][ct], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[hh][ct], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[hh][ct], d, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) qt_tile[(4 * fg + r) * LD + hh * DKP + ct * 16 + fi] = d[r];
                }
            }

            // ---- edges of the group: one q~ row per segment from LDS, gathered K rows, per-head dots
            int cur_row = -1;
            float qt[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e0 = e_lo; e0 < e_hi; e0 += BU) {
                if (e0 != e_lo) issue(e0);
#pragma unroll
                for (int u = 0; u < BU; ++u) {
                    const int e = e0 + u;
                    if (e < e_hi) {
                        // segment ordinal of edge e inside the chunk = (#starts at positions <= e) - 1
                        const int row = __builtin_popcountll(smask & ((2ull << e) - 1ull)) - 1 - g0;
                        if (row != cur_row) {
                            const float4 t = *reinterpret_cast<const float4*>(qt_tile + row * LD + lane * 4);
                            qt[0] = t.x; qt[1] = t.y; qt[2] = t.z; qt[3] = t.w;
                            cur_row = row;
                        }
                        float part = 0.0f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float kv = kr[u][i];
                            if constexpr (RTE) kv += tr[u][i];
                            part = fmaf(qt[i], kv, part);
                        }
                        part = head8_sum(part);
                        if (p == 0) logits[(int64_t)(base + e) * H + h] = part;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int hgt_relation_pack_bf16(const float* att_t, int32_t R, int32_t H_, int32_t dk_pad, void* att_bf, void* stream) {
    if (!att_t || !att_bf || R <= 0) return HGT_ERR_INVALID_ARG;
    if (H_ != H || dk_pad != DKP) return HGT_ERR_UNSUPPORTED;
    const int64_t n_mat = (int64_t)R * H, total = n_mat * 2 * 64 * 8;
    k_pack_att_bf16<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(att_t, n_mat, (unsigned short*)att_bf);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_logits_bf16x3(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H_, int32_t dk_pad,
                                      const float* Q, const float* K, const float* rte_k, const void* att_bf, float* logits,
                                      void* stream_) {
    if (!plan || !Q || !K || !att_bf || !logits) return HGT_ERR_INVALID_ARG;
    if (H_ != H || dk_pad != DKP) return HGT_ERR_UNSUPPORTED;
    if (E == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    const unsigned blocks = (unsigned)((pv.L.max_items + 3) / 4);
    if (rte_k)
        k_edge_logits_mfma<true><<<blocks, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rte_k,
                                                             (const unsigned short*)att_bf, logits, (int)R);
    else
        k_edge_logits_mfma<false><<<blocks, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rte_k,
                                                              (const unsigned short*)att_bf, logits, (int)R);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
