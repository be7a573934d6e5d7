// Pass 2 of the edge phase (softmax + attention-weighted aggregation, conv.py:104,108-111 + PyG's scatter-add) with the
// relation transforms on the MATRIX CORES.
//
//     agg_i,h = sum_r ( sum_{e in (i,r)} att_e v_j,h ) M[r,h]                                 
//
// The vector-ALU kernel (hgt_edge_agg_valu.hip) evaluates the d_k x d_k product once per (target, relation) SEGMENT as a
// mat-vec out of a 128-register fragment: 5.9 M segments x 64 dependent v_pk_fma_f32 + an LDS bounce + an accumulator
// read-modify-write at c2 -- it is bound by VALU issue at 2 waves per SIMD, not by HBM (round-1 review: 0.36 of the roofline).
// Here a wavefront still owns 16 consecutive targets and walks their relations in ascending order, but
//   * per relation r it only accumulates  U_r[t] = sum_e exp(s_e - m_t) v_j  per target (per edge: one gathered row, one exp,
//     VEC fmas); a finished segment is split into bf16 hi/mid and parked as ONE ROW of a wave-private LDS tile U_r[16 x DP]
//     (XOR-swizzled 16-byte slots: conflict-free ds_write_b64 on the way in, conflict-free ds_read_b128 on the way out);
//   * after the relation's edges,  Z^T += blockdiag(M_r)^T . U_r^T  runs on v_mfma_f32_16x16x32_bf16 (3 products per tile:
//     mid*hi + hi*mid + hi*hi, fp32 accumulate): first operand = fragment-ordered hi/mid image of M_r (L2-resident, written
//     once per parameter set by hgt_relation_frag_pack), second operand = U_r rows straight out of the tile.  Computing the
//     TRANSPOSE puts all of a lane's accumulators on ONE target (column = lane & 15), so
//   * the accumulators Z stay in MFMA registers across ALL relations (64 VGPRs at d = 256) -- no per-segment accumulator
//     traffic at all -- and the softmax can stay online: weights are taken relative to a per-(target, head) reference that is
//     only moved when a logit exceeds it by more than 40 (rare); moving it rescales that target's column = a per-lane multiply.
// No barriers, no atomics, fixed summation order.  The node update (a_linear + gated skip + LayerNorm, conv.py:119-133) is
// fused behind it exactly as before (hgt_fused_update.h): the finished rows go from the accumulators through gelu into the
// bf16 hi/mid A slab of the epilogue.
#include "hgt_edge_common.h"
#include "hgt_fused_update.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef HGT_FU_NSTG
#define HGT_FU_NSTG 2     // W_a fragment stages of the fused epilogue (hgt_fused_update.h)
#endif
#ifndef HGT_AGG_GS
#define HGT_AGG_GS 8      // column-tile steps whose fragments are requested together at a relation end (16 loads in flight)
#endif
template <int VEC> struct RowT;
template <> struct RowT<4> { typedef f32x4 type; };
template <> struct RowT<2> { typedef f32x2 type; };
template <> struct RowT<1> { typedef float type; };
__device__ __forceinline__ float row_elem(const f32x4& v, int i) { return v[i]; }
__device__ __forceinline__ float row_elem(const f32x2& v, int i) { return v[i]; }
__device__ __forceinline__ float row_elem(const float& v, int) { return v; }

template <int VEC, int LPH>
struct MG {   // geometry of one wavefront's slice: DP columns = 64 lanes x VEC floats
    static constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH;
    static constexpr int NCT = DP / 16;                 // 16-column output tiles
    static constexpr int KW = DKP > 32 ? DKP : 32;      // k window of a column tile (a head, or 32 columns holding several heads)
    static constexpr int NKS = KW / 32;                 // MFMA k-steps per column tile
    static constexpr int ROWB = DP * 2;                 // bytes of one bf16 row of the U tile
    static constexpr int NS = DP / 8;                   // 16-byte slots per row
    static constexpr int PLANE = 16 * ROWB;             // one bf16 plane of the tile (16 targets)
};

// msg_p [R][HT][DKP][DKP] fp32 (hgt_relation_pack: msg_p[r][h][k][c] = relation_msg[r][h][k][c], zero padded) ->
// fragments [R][head group][col tile c][k-step s][plane][lane 64][8] bf16 of blockdiag_h(M[r,h]) restricted to the k window
// of the column tile:  frag[..][l][e] = split( Mfull[kbase(c) + 32 s + (l>>4)*8 + e][16 c + (l&15)] )
// = the FIRST operand of v_mfma_f32_16x16x32_bf16 (row = output column l&15, k = (l>>4)*8 + e).
// F16: fp16 hi / lo fragments of M * scale, ONE power-of-two scale for all relations and heads (k_msg_scale: it has to be the same
// for everything that is summed into one accumulator; its inverse sits behind the fragments and is applied when the rows are
// normalised)
template <bool F16>
__global__ void k_msg_frag_pack(const float* __restrict__ msgP, int R, int HT, int DKP, int DP, unsigned short* __restrict__ out,
                                const float* __restrict__ gscale) {
    const int KW = DKP > 32 ? DKP : 32, NKS = KW / 32, NCT = DP / 16, NY = HT * DKP / DP;
    const int64_t total = (int64_t)R * NY * NCT * NKS * 512;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t r = i;
    const int e = (int)(r % 8); r /= 8;
    const int l = (int)(r % 64); r /= 64;
    const int s = (int)(r % NKS); r /= NKS;
    const int c = (int)(r % NCT); r /= NCT;
    const int hg = (int)(r % NY);
    const int rel = (int)(r / NY);
    const int kbase = (16 * c / KW) * KW;
    const int kf = hg * DP + kbase + 32 * s + (l >> 4) * 8 + e, nf = hg * DP + 16 * c + (l & 15);
    const int hk = kf / DKP, hn = nf / DKP;
    float v = 0.0f;
    if (hk == hn) v = msgP[(((int64_t)rel * HT + hn) * DKP + kf % DKP) * DKP + nf % DKP];
    unsigned short hi, mid;
    split1_t<F16>(v, F16 ? gscale[1] : 1.0f, hi, mid);
    const int64_t tile = ((((int64_t)rel * NY + hg) * NCT + c) * NKS + s) * 2;
    out[(tile + 0) * 512 + l * 8 + e] = hi;
    out[(tile + 1) * 512 + l * 8 + e] = mid;
}

// max |M| over all relations -> tail = {inverse scale, scale} (one workgroup)
__global__ __launch_bounds__(1024) void k_msg_scale(const float* __restrict__ msgP, int64_t n, float* __restrict__ tail) {
    __shared__ unsigned s_m[16];
    unsigned m = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = max(m, __builtin_bit_cast(unsigned, fabsf(msgP[i])));
    m = wave_max_bits(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = max(m, s_m[w]);
        f16_row_scale(m, tail[1], tail[0]);
    }
}

// fp16 split of the U rows (F16): every TARGET of a sub-tile has one power-of-two scale sigma_t for all its relations (its rows
// of all relations are summed into one accumulator column).  The wavefront keeps the 16 scales in ONE register, lane i = target
// i (sigv; thrv = 2^15 / sigma_t, 0 = not set yet), read with v_readlane -- an LDS table cost a dependent LDS round trip per
// parked row, +0.55 ms at the benchmark size.  sigma_t is chosen when the target's first row is parked -- row maximum ->
// [2^8, 2^9): 2^7 of headroom for the rows of its other relations -- and the common case afterwards is ONE compare of the lane's
// own maximum with the threshold (no cross-lane reduction); only a row that would leave the fp16 range takes the slow path
// again, which moves sigma_t and rescales the target's accumulator column (exactly like a move of the softmax reference).
// 1 / sigma_t joins the softmax normalisation at the end.
template <int VEC>
__device__ __forceinline__ float f16_target_scale(const float (&U)[VEC], float& sigv, float& thrv, float& rescv, int& resc_any, int dl_) {
    const int dl = __builtin_amdgcn_readfirstlane(dl_);
    float m;
    if constexpr (VEC == 4) {     // two instructions (fmaxf(fabsf()) chains cost four: hipcc canonicalises every operand)
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32 %0, %0, |%4|" : "=&v"(m) : "v"(U[0]), "v"(U[1]), "v"(U[2]), "v"(U[3]));
    } else {
        m = fabsf(U[0]);
#pragma unroll
        for (int i = 1; i < VEC; ++i) m = fmaxf(m, fabsf(U[i]));
    }
    const float thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thrv), dl));
    const unsigned sb = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, sigv), dl);
    if (__builtin_amdgcn_ballot_w64(m >= thr) == 0) return __builtin_bit_cast(float, sb);      // fits under the target's scale
    unsigned e = wave_max_bits(__builtin_bit_cast(unsigned, m)) >> 23;       // biased exponent of the row maximum, wave-uniform
    const unsigned se = sb >> 23;                                              // biased exponent of sigma_t (0 = unset)
    if (e < 40u) return se != 0u ? __builtin_bit_cast(float, sb) : 1.0f;       // an all-zero (or vanishing) row decides nothing
    e = e > 220u ? 220u : e;
    const unsigned ne = 127u + 8u + 127u - e;                                 // sigma = 2^(8 - (e - 127))
    const float sig = __builtin_bit_cast(float, ne << 23);
    const bool mine = (int)(threadIdx.x & 63) == dl;
    if (se != 0u) {      // sigma moves: the target's accumulator column follows at the relation's end (f16_apply_rescale) -- done
                         // here, the structurizer merged this rare branch into the common paths through 64 accumulator copies
        const int de = 127 + (int)ne - (int)se;                                    // sigma_new / sigma_old (< 1) = 2^(ne - se)
        const float rescale = de > 0 ? __builtin_bit_cast(float, (unsigned)de << 23) : 0.0f;   // (below 2^-126: what was summed is nothing)
        rescv = mine ? rescv * rescale : rescv;
        resc_any = 1;
    }
    sigv = mine ? sig : sigv;
    thrv = mine ? __builtin_bit_cast(float, (127u + 15u + 127u - ne) << 23) : thrv;      // 2^15 / sigma
    return sig;
}

// the pending moves of sigma_t (f16_target_scale) applied to the accumulator columns, before the relation's products are added
template <int NCT>
__device__ __forceinline__ void f16_apply_rescale(float& rescv, int& resc_any, f32x4 (&acc)[NCT]) {
    if (resc_any) {
        const float f = __shfl(rescv, (int)(threadIdx.x & 15));
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] *= f;
        rescv = 1.0f;
        resc_any = 0;
    }
}

template <int VEC>
constexpr int agg_unroll(bool rte) { return rte ? 4 : 8; }

// One wavefront: targets [row0, row0 + SUBR) of the destination tile, all relations.  On return acc[c] holds, for target
// (lane & 15) and columns 16 c + 4 (lane >> 4) .. + 3, the UN-normalised aggregate sum_e exp(s_e - m) v'_e; s_l (LDS) holds
// the matching exp-sums per (target, head).
template <int VEC, int LPH, bool RTE, bool HUBS, bool F16>
__device__ __forceinline__ void agg_mfma_subtile(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, int R, int rel_lo, int rel_hi, int HT, unsigned hub_mask,
    int SUBR, int64_t row0, unsigned char* utile, float* s_m, float* s_l, float* s_sc, float* s_sig, int raw,
    f32x4 (&acc)[MG<VEC, LPH>::NCT]) {
    using G = MG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS;
    constexpr int UN = agg_unroll<VEC>(RTE);
    float sigv = 0.0f, thrv = 0.0f, rescv = 1.0f;   // fp16 split: the targets' scales / thresholds / pending moves, lane i = target i
    int resc_any = 0;
    if constexpr (F16) { if ((threadIdx.x & 63) < 16) s_sig[threadIdx.x & 63] = 0.0f; }
    const int hg = blockIdx.y;              // head group (head-group split: the wave covers DP of the HT * DKP columns)
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP;
    const int NY = HT / H;

    const int lane = threadIdx.x & 63;
    const int h = lane / LPH, p = lane % LPH;
    const int fi = lane & 15, fg = lane >> 4;
    const int tile = (int)(row0 / HGT_TD);
    const int within = (int)(row0 % HGT_TD);

#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_m[j * 64 + lane] = HGT_NEG; s_l[j * 64 + lane] = 0.0f; }

    // byte offset of this lane's VEC columns inside a tile row (gather layout), before the row swizzle
    const int wb = lane * VEC * 2;
    // read offset of the (row = fi, k block fg) fragment of column tile 0 / k-step 0, before the swizzle
    const int rrow = fi * ROWB;

    // hub-free sub-tiles: the edge range of every relation bucket is read up front (lane r = bucket r)
    const bool ranges_ready = !HUBS && R < 64;
    int my_beg = 0, my_end = 0;
    if (ranges_ready) {
        const int64_t bb = ((int64_t)tile * (R + 1) + min(lane, R)) * HGT_TD + within;
        my_beg = segptr[bb];
        my_end = segptr[bb + SUBR];
    }
    for (int rel = rel_lo; rel < min(rel_hi, R + 1); ++rel) {   // (the whole layer: [0, R]; a source bucket of the multi-GPU path: a slice)
        const int64_t b0 = ((int64_t)tile * (R + 1) + rel) * HGT_TD + within;
        const bool claimed = rel < R;   // bucket R: logit 0, no message (conv.py:68-69)
        unsigned rowmask = 0;           // rows of the U tile written for this relation (wave-uniform)
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

            int cur_dst = -1;
            float U[VEC], m_ref = 0.0f, l_seg = 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) U[i] = 0.0f;

            auto flush = [&]() {
                if (cur_dst >= 0) {
                    const int dl = cur_dst - (int)row0;
                    if (p == 0) {
                        s_l[dl * 16 + h] += l_seg;
                        s_m[dl * 16 + h] = m_ref;
                    }
                    if (claimed) {
                        unsigned char* w = utile + dl * ROWB + ((((wb >> 4) ^ (dl & (NS - 1)))) << 4) + (wb & 15);
                        float sig = 1.0f;
                        if constexpr (F16) {
                            sig = f16_target_scale<VEC>(U, sigv, thrv, rescv, resc_any, dl);
                        }
                        if constexpr (VEC == 1) {
                            unsigned short hi, mid;
                            split1_t<F16>(U[0], sig, hi, mid);
                            *reinterpret_cast<unsigned short*>(w) = hi;
                            *reinterpret_cast<unsigned short*>(w + G::PLANE) = mid;
                        } else if constexpr (VEC == 2) {
                            unsigned hi, mid;
                            split2_t<F16>(U[0], U[1], sig, hi, mid);
                            *reinterpret_cast<unsigned*>(w) = hi;
                            *reinterpret_cast<unsigned*>(w + G::PLANE) = mid;
                        } else {
                            uint2 hi, mid;
                            split4_t<F16>(make_float4(U[0], U[1], U[2], U[3]), sig, hi, mid);
                            *reinterpret_cast<uint2*>(w) = hi;
                            *reinterpret_cast<uint2*>(w + G::PLANE) = mid;
                        }
                        rowmask |= 1u << dl;
                    }
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
                                l_seg = 0.0f;
                                cur_dst = dsts[u];
                                // reference of the target's softmax weights: the first logit the target ever saw (any
                                // relation); kept in LDS between the relations.  Any reference gives the same softmax.
                                const float m_t = s_m[(cur_dst - (int)row0) * 16 + h];
                                m_ref = (m_t == HGT_NEG) ? sl[u] : m_t;
                            }
                            float dlt = sl[u] - m_ref;
                            if (!raw && __builtin_amdgcn_ballot_w64(dlt > 40.0f) != 0) {
                                // Rare: move the reference of the heads that were exceeded; everything this target has
                                // accumulated so far is rescaled: U / l of the running segment, its exp-sum in LDS and its
                                // COLUMN of the MFMA accumulators (lanes with (lane & 15) == target: a per-lane multiply).
                                const float m_new = (dlt > 40.0f) ? sl[u] : m_ref;
                                const float sc = __expf(m_ref - m_new);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) U[i] *= sc;
                                l_seg *= sc;
                                const int dl = cur_dst - (int)row0;
                                if (p == 0) {
                                    s_l[dl * 16 + h] *= sc;
                                    s_sc[h] = sc;
                                }
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                                for (int c = 0; c < NCT; ++c) {
                                    const float f = (fi == dl) ? s_sc[(16 * c + 4 * fg) / DKP] : 1.0f;
                                    acc[c] *= f;
                                }
                                __builtin_amdgcn_wave_barrier();
                                m_ref = m_new;
                                dlt = sl[u] - m_ref;
                            }
                            const float pe = raw ? sl[u] : __expf(dlt);
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
        if (rowmask == 0) continue;   // no claimed edge in this relation: nothing to transform
        if constexpr (F16) f16_apply_rescale<NCT>(rescv, resc_any, acc);

        // rows without an edge in this relation contribute nothing: zero them (rows >= SUBR are never read back: every
        // column of the transposed product depends on its own row only)
        for (int r = 0; r < SUBR; ++r) {
            if ((rowmask >> r) & 1u) continue;
            unsigned char* w = utile + r * ROWB + ((((wb >> 4) ^ (r & (NS - 1)))) << 4) + (wb & 15);
            if constexpr (VEC == 1) {
                *reinterpret_cast<unsigned short*>(w) = 0;
                *reinterpret_cast<unsigned short*>(w + G::PLANE) = 0;
            } else if constexpr (VEC == 2) {
                *reinterpret_cast<unsigned*>(w) = 0u;
                *reinterpret_cast<unsigned*>(w + G::PLANE) = 0u;
            } else {
                *reinterpret_cast<uint2*>(w) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(w + G::PLANE) = make_uint2(0u, 0u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // Z^T[16 c .. +16][target] += M_r^T[cols of tile c][k window] . U_r^T[k window][target]
        const unsigned short* __restrict__ mf = msgF + (((int64_t)rel * NY + hg) * NCT) * NKS * 2 * 512 + lane * 8;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int kbase = (16 * c / KW) * KW;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const int slot = (kbase + 32 * s) / 8 + fg;
                const unsigned char* up = utile + rrow + ((slot ^ (fi & (NS - 1))) << 4);
                const bf16x8 uh = *reinterpret_cast<const bf16x8*>(up);
                const bf16x8 um = *reinterpret_cast<const bf16x8*>(up + G::PLANE);
                const unsigned short* t = mf + (int64_t)((c * NKS + s) * 2) * 512;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(t);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(t + 512);
                acc[c] = mfma16_t<F16>(am, uh, acc[c]);
                acc[c] = mfma16_t<F16>(ah, um, acc[c]);
                acc[c] = mfma16_t<F16>(ah, uh, acc[c]);
            }
        }
        __builtin_amdgcn_wave_barrier();   // the tile is rewritten by the next relation only after every lane has read it
    }
    if constexpr (F16) { if ((threadIdx.x & 63) < 16) s_sig[threadIdx.x & 63] = sigv; }      // for agg_mfma_finish
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the sub-tile walk (hub-free sub-tiles, R < 64): the kernel is bound by the LATENCY of its gathers, not
// by arithmetic (rocprofv3, c2: VALU 27 % busy, 64 % of the wave cycles parked in s_waitcnt), so what matters is how many
// rows a wavefront keeps in flight.  The walk above restarts its load pipeline for every (sub-tile, relation) range
// (~20 edges: three dependent round trips -- edge ids, rows, fragments -- per range).  Here
//   * the R + 1 ranges of the sub-tile are concatenated into ONE virtual edge stream; edge ids / relation / sorted position
//     of 64 stream entries live in lane registers (`meta`), the next 64 are fetched one chunk ahead;
//   * the stream is cut into BATCHES of <= UN edges that never straddle a relation; the rows of batch k + 1 are requested
//     before batch k is consumed (two register buffers A / B), across relation boundaries -- every request is unconditional
//     and of fixed size, so hipcc keeps counted s_waitcnt vmcnt(N);
//   * the relation-end work (zero rows, fragment loads, 48 MFMAs) runs BETWEEN batches, with the next batch's rows already
//     on their way.
// ---------------------------------------------------------------------------------------------
// FULL: the wavefront covers whole rows (HT == H, no head-group split: the fused kernels) -- row and logit strides are then compile-time
// powers of two; as run-time values every gathered row paid a 64-bit scalar multiply (7 SALU instructions, round-5 ISA audit of a
// kernel that is bound by instruction issue: profiles/r05_agg_counters.txt)
// S32 (fused kernels, chosen by the launcher): every gathered row / logit / temporal row lies below 4 GiB from its base -> 32-bit
// unsigned lane offsets, one VALU instruction per address instead of a 64-bit scalar shift + a 64-bit vector add (r05: 3.22 -> 3.14 ms)
template <int VEC, int LPH, bool RTE, bool F16, bool FULL = false, bool S32 = false>
__device__ __forceinline__ void agg_mfma_stream(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, int R, int rel_lo, int rel_hi, int HT_, int SUBR,
    int64_t row0, unsigned char* utile, float* s_m, float* s_l, float* s_sc, float* s_sig, int raw,
    f32x4 (&acc)[MG<VEC, LPH>::NCT]) {
    using G = MG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS;
    // fp16 split, OPTIMISTIC row scales (round 4): sigma_t (lane i = target i) is chosen when the target's first non-zero row is
    // parked -- row maximum -> [1, 2), i.e. 2^14 of headroom for the rows of its other relations and still ~2^-25 of absolute
    // resolution through the subnormal lo terms -- and later rows are NOT checked against it (round 3's per-row check: two
    // v_readlane, a compare and a branch per parked row cost +0.26 ms at the benchmark size).  A row that leaves the fp16 range
    // turns into inf in the U tile and NaN in the accumulators; that is looked for ONCE, after the walk, and the sub-tile is
    // then walked again with 2^14 more headroom FOR THE TARGETS THAT OVERFLOWED (shiftv; a column of the transposed product only
    // depends on its own target's rows, so the other targets of the sub-tile keep their scales and reproduce their values).
    // Rows within 2^14 of their target's first row -- a logit spread of ~9.7 above the first logit seen -- never retry.
    // The retries are not capped below the fp32 range (round-4 advisor finding: a cap of 4 published NaN for a target whose first row
    // was 2^70 smaller than a later one).
    float sigv = 0.0f;
    int shiftv = 0;      // extra headroom (powers of two) per target, lane i = target i: grows by 14 for a target whose rows overflowed
#ifndef HGT_AGG_UN
#define HGT_AGG_UN 4
#endif
#ifndef HGT_AGG_UN_RTE
#define HGT_AGG_UN_RTE 3
#endif
    constexpr int UN = RTE ? HGT_AGG_UN_RTE : HGT_AGG_UN;    // rows per batch; two batches (register buffers A / B) are in flight
    const int hg = FULL ? 0 : blockIdx.y;
    const int HTx = FULL ? H : HT_;      // heads per row
    const int64_t ld = FULL ? (int64_t)DP : (int64_t)HT_ * DKP;
    const int co = hg * DP;
    const int NY = FULL ? 1 : HT_ / H;

    const int lane = threadIdx.x & 63;
    const int h = lane / LPH, p = lane % LPH;
    const int fi = lane & 15, fg = lane >> 4;
    const int tile = (int)(row0 / HGT_TD);
    const int within = (int)(row0 % HGT_TD);

    const int wb = lane * VEC * 2;
    const int rrow = fi * ROWB;

    // ranges of the R + 1 relation buckets (lane r = bucket r) and their exclusive prefix = position in the virtual stream
    int my_beg = 0, my_len = 0;
    {
        const int64_t bb = ((int64_t)tile * (R + 1) + min(lane, R)) * HGT_TD + within;
        my_beg = segptr[bb];
        my_len = (lane <= R && lane >= rel_lo && lane < rel_hi) ? segptr[bb + SUBR] - my_beg : 0;   // relations outside the slice: empty ranges
    }
    int incl = my_len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const int my_pre = incl - my_len;
    const int total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_m[j * 64 + lane] = HGT_NEG; s_l[j * 64 + lane] = 0.0f; }
    if constexpr (F16) { if (lane < 16) s_sig[lane] = 0.0f; }
    if (total == 0) return;
  for (int attempt = 0;; ++attempt) {      // (one pass; the fp16 split may walk the sub-tile again with more headroom, see above)
    if (attempt > 0) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_m[j * 64 + lane] = HGT_NEG; s_l[j * 64 + lane] = 0.0f; }
        sigv = 0.0f;
        __builtin_amdgcn_wave_barrier();
    }

    // stream entries [vbase, vbase + 64) -> lane registers; entries beyond the end replicate the last edge
    // (the relation and the target's position inside the sub-tile travel in ONE register, key = relation << 8 | target - row0:
    //  one v_readlane and one SGPR per batch slot instead of two -- the kernel is short of both register files)
    auto load_meta = [&](int vbase, int& m_src, int& m_key, int& m_pos, int& m_rte) {
        const int v = min(vbase + lane, total - 1);
        int rsel = 0;
        for (int r = 0; r <= R; ++r) {
            const int pr = __builtin_amdgcn_readlane(my_pre, r), ln = __builtin_amdgcn_readlane(my_len, r);
            if (ln > 0 && v >= pr) rsel = r;
        }
        m_pos = __shfl(my_beg, rsel) + (v - __shfl(my_pre, rsel));
        m_src = esrc[m_pos];
        m_key = (edst[m_pos] - (int)row0) | (rsel << 8);
        m_rte = RTE ? (int)ertei[m_pos] : 0;
    };

#define load_meta_next load_meta
#define AGG_META_WAIT
    int c_src, c_key, c_pos, c_rte;          // current chunk
    int n_src = 0, n_key = 0, n_pos = 0, n_rte = 0;   // next chunk (prefetched)
    int vbase = 0, nb = min(64, total);
    load_meta(0, c_src, c_key, c_pos, c_rte);
    if (total > 64) load_meta(64, n_src, n_key, n_pos, n_rte);

    int cur_dl = -1, cur_rel = -1;       // running segment: target position inside the sub-tile / relation
    unsigned rowmask = 0;
    float U[VEC], m_ref = 0.0f, l_seg = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
    bool seg_claimed = false;      // relation of the running segment is a real one (its rows go into the U tile)

    // INVARIANTS of flush() and of the row path (round-5 advisor):
    //  * the exp-sum write-back `s_l[..] += l_seg` is a plain read-modify-write executed by EVERY lane of a head with the SAME l_seg
    //    (l_seg is updated from head-uniform logits only): it must stay non-atomic and lane-uniform -- a ds_add_f32 here would add
    //    l_seg once per lane of the head (LPH times);
    //  * U is parked in the tile ONLY under seg_claimed, and U is zeroed at every segment change: the rows of the bucket of unclaimed
    //    edges (rel == R) are accumulated into U unconditionally by AGG_ROW_ACC (no branch on the row path) and then dropped here --
    //    an unclaimed segment's U must never reach the tile.
    auto flush = [&]() {
        if (cur_dl >= 0) {
            const int dl = cur_dl;
            {      // (every lane of the head writes the same pair: no exec mask around it -- r05: 3.22 -> 3.19 ms)
                s_l[dl * 16 + h] += l_seg;      // (as ds_add_f32 without return -- one instruction, no round trip -- measured the same: r05)
                s_m[dl * 16 + h] = m_ref;
            }
            if (seg_claimed) {
                unsigned char* w = utile + dl * ROWB + ((((wb >> 4) ^ (dl & (NS - 1)))) << 4) + (wb & 15);
                float sig = 1.0f;
                if constexpr (F16) {
                    const int dls = __builtin_amdgcn_readfirstlane(dl);
                    unsigned sb = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, sigv), dls);
                    if (__builtin_expect(sb == 0u, 0)) {      // the target's first row: its maximum decides sigma_t (an all-zero row decides nothing)
                        float m = fabsf(U[0]);
#pragma unroll
                        for (int i = 1; i < VEC; ++i) m = fmaxf(m, fabsf(U[i]));
                        unsigned e = wave_max_bits(__builtin_bit_cast(unsigned, m)) >> 23;
                        sb = 0x3f800000u;
                        if (e >= 40u) {
                            e = e > 220u ? 220u : e;
                            const int be = 254 - (int)e - __builtin_amdgcn_readlane(shiftv, dls);   // sigma = 2^(-(e - 127) - shift): row maximum -> [1, 2) / 2^shift
                            sb = (unsigned)(be < 1 ? 1 : be) << 23;
                            sigv = (lane == dls) ? __builtin_bit_cast(float, sb) : sigv;
                        }
                    }
                    sig = __builtin_bit_cast(float, sb);
                }
                if constexpr (VEC == 1) {
                    unsigned short hi, mid;
                    split1_t<F16>(U[0], sig, hi, mid);
                    *reinterpret_cast<unsigned short*>(w) = hi;
                    *reinterpret_cast<unsigned short*>(w + G::PLANE) = mid;
                } else if constexpr (VEC == 2) {
                    unsigned hi, mid;
                    split2_t<F16>(U[0], U[1], sig, hi, mid);
                    *reinterpret_cast<unsigned*>(w) = hi;
                    *reinterpret_cast<unsigned*>(w + G::PLANE) = mid;
                } else {
                    uint2 hi, mid;
                    split4_t<F16>(make_float4(U[0], U[1], U[2], U[3]), sig, hi, mid);
                    *reinterpret_cast<uint2*>(w) = hi;
                    *reinterpret_cast<uint2*>(w + G::PLANE) = mid;
                }
                rowmask |= 1u << dl;
            }
        }
        cur_dl = -1;
    };

    // end of a relation: Z^T += M_r^T . U_r^T for the rows parked in the tile.  The fragments of M_r come from L2 in groups
    // of GS column-tile steps, eight 1 KB loads in flight at a time, pinned with sched_barrier: left alone, hipcc issued the 32
    // fragment loads of a relation two at a time and waited for each pair with vmcnt(0) (ISA audit of the first version) --
    // sixteen dependent L2 round trips per relation end, the largest single cost of that version.
    constexpr int STEPS = NCT * NKS, GS = HGT_AGG_GS < STEPS ? HGT_AGG_GS : STEPS, NG = STEPS / GS;
    static_assert(STEPS % GS == 0, "column-tile steps come in multiples of 4 (DP is a multiple of 64)");
    auto relation_end = [&](int rel) {
        if (rowmask == 0) return;
        for (unsigned zm_ = ~rowmask & ((1u << SUBR) - 1u); zm_ != 0; zm_ &= zm_ - 1) {      // (only the absent rows are visited)
            const int r = __builtin_ctz(zm_);
            unsigned char* w = utile + r * ROWB + ((((wb >> 4) ^ (r & (NS - 1)))) << 4) + (wb & 15);
            if constexpr (VEC == 1) {
                *reinterpret_cast<unsigned short*>(w) = 0;
                *reinterpret_cast<unsigned short*>(w + G::PLANE) = 0;
            } else if constexpr (VEC == 2) {
                *reinterpret_cast<unsigned*>(w) = 0u;
                *reinterpret_cast<unsigned*>(w + G::PLANE) = 0u;
            } else {
                *reinterpret_cast<uint2*>(w) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(w + G::PLANE) = make_uint2(0u, 0u);
            }
        }
        rowmask = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned short* __restrict__ mf = msgF + (((int64_t)rel * NY + hg) * NCT) * NKS * 2 * 512 + lane * 8;
        bf16x8 f0h[GS], f0m[GS];     // one group in flight: a second register buffer would push the kernel over 256 VGPRs
#define FRAG_ISSUE(FH, FM, G_)                                                                     \
    _Pragma("unroll") for (int j = 0; j < GS; ++j) {                                               \
        const unsigned short* t_ = mf + (int64_t)(((G_) * GS + j) * 2) * 512;            \
        FH[j] = *reinterpret_cast<const bf16x8*>(t_);                                              \
        FM[j] = *reinterpret_cast<const bf16x8*>(t_ + 512);                                        \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);   /* all 2 GS loads of the group are issued before the first MFMA waits */
#define FRAG_MUL(FH, FM, G_)                                                                       \
    _Pragma("unroll") for (int j = 0; j < GS; ++j) {                                               \
        const int step = (G_) * GS + j, c = step / NKS, ks = step % NKS;                           \
        const int kbase = (16 * c / KW) * KW;                                                      \
        const int slot = (kbase + 32 * ks) / 8 + fg;                                               \
        const unsigned char* up = utile + rrow + ((slot ^ (fi & (NS - 1))) << 4);                  \
        const bf16x8 uh = *reinterpret_cast<const bf16x8*>(up);                                    \
        const bf16x8 um = *reinterpret_cast<const bf16x8*>(up + G::PLANE);                         \
        acc[c] = mfma16_t<F16>(FM[j], uh, acc[c]);                                                 \
        acc[c] = mfma16_t<F16>(FH[j], um, acc[c]);                                                 \
        acc[c] = mfma16_t<F16>(FH[j], uh, acc[c]);                                                 \
    }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            FRAG_ISSUE(f0h, f0m, g)
            FRAG_MUL(f0h, f0m, g)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef FRAG_ISSUE
#undef FRAG_MUL
        __builtin_amdgcn_wave_barrier();
    };

    typedef typename RowT<VEC>::type Row;
    Row vrA[UN], trA[RTE ? UN : 1], vrB[UN], trB[RTE ? UN : 1];
    float slA[UN], slB[UN];
    int keyA[UN], keyB[UN];

    // A batch = the next UN entries of the stream (fewer only at a chunk end), whatever their relations.
#define AGG_ISSUE(VR, SL, TR, KY, I0, CNT)                                                         \
    _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                               \
        /* slots beyond the batch re-request its last edge (same address: served by the L1) */     \
        const int idx = min((I0) + u, (I0) + max((CNT), 1) - 1);                                   \
        const int s_ = __builtin_amdgcn_readlane(c_src, idx);                                      \
        const int p_ = __builtin_amdgcn_readlane(c_pos, idx);                                      \
        KY[u] = __builtin_amdgcn_readlane(c_key, idx);                                             \
        if constexpr (FULL && S32) {                                                               \
            AGG_LOAD(VR[u], reinterpret_cast<const char*>(V) + (unsigned)((unsigned)s_ * (unsigned)(DP * 4) + (unsigned)(lane * VEC * 4))) \
            AGG_LOAD(SL[u], reinterpret_cast<const char*>(logits) + (unsigned)((unsigned)p_ * (unsigned)(H * 4) + (unsigned)(h * 4)))       \
            if constexpr (RTE) {                                                                   \
                const int ri = __builtin_amdgcn_readlane(c_rte, idx);                              \
                AGG_LOAD(TR[u], reinterpret_cast<const char*>(rteV) + (unsigned)((unsigned)ri * (unsigned)(DP * 4) + (unsigned)(lane * VEC * 4))) \
            }                                                                                      \
        } else {                                                                                   \
            AGG_LOAD(VR[u], V + (int64_t)s_ * ld + co + lane * VEC)                                \
            AGG_LOAD(SL[u], logits + (int64_t)p_ * HTx + hg * H + h)                               \
            if constexpr (RTE) {                                                                   \
                const int ri = __builtin_amdgcn_readlane(c_rte, idx);                              \
                AGG_LOAD(TR[u], rteV + (int64_t)ri * ld + co + lane * VEC)                         \
            }                                                                                      \
        }                                                                                          \
    }
#define AGG_LOAD(D, P) D = *reinterpret_cast<const __typeof__(D)*>(P);
#define AGG_WAIT(VR, SL, TR)
    // One gathered row of the current relation in three pieces: segment change (parking of the previous segment, state of the new
    // one), the rare re-reference of the softmax, the accumulation
#define AGG_ROW_SEG(SL, KY, u)                                                                     \
                if ((KY[u] & 255) != cur_dl) {                                                     \
                    flush();                                                                       \
                    _Pragma("unroll") for (int i = 0; i < VEC; ++i) U[i] = 0.0f;                   \
                    l_seg = 0.0f;                                                                  \
                    cur_dl = KY[u] & 255;                                                          \
                    seg_claimed = claimed_;                                                        \
                    /* (requested here, behind the parking's own exp-sum round trip: requested before it -- one wait for both -- */ \
                    /*  it measured 0.8 % slower, r05) */                                           \
                    const float m_t = s_m[cur_dl * 16 + h];                                        \
                    m_ref = (m_t == HGT_NEG) ? SL[u] : m_t;                                        \
                }
#define AGG_ROW_RESCALE(SL, u)                                                                     \
                    const float m_new = (dlt > 40.0f) ? SL[u] : m_ref;                             \
                    const float sc = __expf(m_ref - m_new);                                        \
                    _Pragma("unroll") for (int i = 0; i < VEC; ++i) U[i] *= sc;                    \
                    l_seg *= sc;                                                                   \
                    int dl = cur_dl;                                                               \
                    asm volatile("" : "+s"(dl));   /* (keeps the lane masks of this rare path out of every batch: hipcc hoisted four v_cmp per batch) */ \
                    if (p == 0) {                                                                  \
                        s_l[dl * 16 + h] *= sc;                                                    \
                        s_sc[h] = sc;                                                              \
                    }                                                                              \
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                         \
                    __builtin_amdgcn_wave_barrier();                                               \
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                         \
                    _Pragma("unroll") for (int c = 0; c < NCT; ++c) {                              \
                        const float f = (fi == dl) ? s_sc[(16 * c + 4 * fg) / DKP] : 1.0f;         \
                        acc[c] *= f;                                                               \
                    }                                                                              \
                    __builtin_amdgcn_wave_barrier();                                               \
                    m_ref = m_new;                                                                 \
                    dlt = SL[u] - m_ref;
#define AGG_ROW_ACC(VR, SL, TR, u)                                                                 \
                const float pe = raw ? SL[u] : __expf(dlt);   /* raw: the array holds the edge weights themselves */ \
                /* (no `if (claimed_)`: the rows of the unclaimed bucket are gathered too -- valid addresses -- and their sums are */ \
                /*  never parked (seg_claimed); as a condition it compiled to four v_cndmask per row, round-5 ISA audit) */          \
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) {                                  \
                    float vv = row_elem(VR[u], i);                                                 \
                    if constexpr (RTE) vv += row_elem(TR[u], i);                                   \
                    U[i] = fmaf(pe, vv, U[i]);                                                     \
                }                                                                                  \
                l_seg += pe;
    // Consume a batch.  Ordinary case -- a full batch inside the current relation (at config 2 four of five batches): the rows one after
    // the other, nothing to decide (r05 ISA audit: the run bookkeeping of the general form was ~35 scalar instructions per batch in a
    // kernel bound by instruction issue).  General case: runs [lo, hi) of one relation, the relation-end work between the runs.
#define AGG_PROCESS(VR, SL, TR, KY, CNT)                                                           \
    {                                                                                              \
        int lo = 0;                                                                                \
        bool plain_ = (CNT) == UN && (KY[UN - 1] >> 8) == cur_rel;                                 \
        const bool claimed_p = cur_rel < R;                                                        \
        _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                           \
            if (plain_) {                                                                          \
                const bool claimed_ = claimed_p;                                                   \
                AGG_ROW_SEG(SL, KY, u)                                                             \
                float dlt = SL[u] - m_ref;                                                         \
                if (!raw && __builtin_amdgcn_ballot_w64(dlt > 40.0f) != 0) {                       \
                    plain_ = false;      /* (the re-reference writes the accumulators: the general form takes over at this row) */ \
                } else {                                                                           \
                    AGG_ROW_ACC(VR, SL, TR, u)                                                     \
                    lo = u + 1;                                                                    \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        while (lo < (CNT)) {                                                                       \
            int rel_run = KY[0] >> 8, hi = (CNT);                                                  \
            _Pragma("unroll") for (int u = 1; u < UN; ++u) if (u <= lo) rel_run = KY[u] >> 8;      \
            _Pragma("unroll") for (int u = UN - 1; u >= 1; --u) if (u > lo && u < (CNT) && (KY[u] >> 8) != rel_run) hi = u; \
            if (rel_run != cur_rel) {                                                              \
                flush();                                                                           \
                relation_end(cur_rel);                                                             \
                cur_rel = rel_run;                                                                 \
            }                                                                                      \
            const bool claimed_ = rel_run < R;                                                     \
            _Pragma("unroll") for (int u = 0; u < UN; ++u) {                                       \
                if (u >= lo && u < hi) {                                                           \
                    AGG_ROW_SEG(SL, KY, u)                                                         \
                    float dlt = SL[u] - m_ref;                                                     \
                    if (!raw && __builtin_amdgcn_ballot_w64(dlt > 40.0f) != 0) {                   \
                        AGG_ROW_RESCALE(SL, u)                                                     \
                    }                                                                              \
                    AGG_ROW_ACC(VR, SL, TR, u)                                                     \
                }                                                                                  \
            }                                                                                      \
            lo = hi;                                                                               \
        }                                                                                          \
    }
    // batch after (I0, CNT); rotates the chunk registers at a chunk end; CNTN = 0: the stream is over (the issue that
    // follows then re-requests the last edge: unconditional, fixed size)
#define AGG_NEXT(I0, CNT, I0N, CNTN)                                                               \
    {                                                                                              \
        I0N = (I0) + (CNT);                                                                        \
        if (I0N >= nb) {                                                                           \
            if (vbase + 64 >= total) {                                                             \
                I0N = nb - 1;                                                                      \
                CNTN = 0;                                                                          \
            } else {                                                                               \
                vbase += 64;                                                                       \
                nb = min(64, total - vbase);                                                       \
                AGG_META_WAIT                                                                      \
                c_src = n_src; c_key = n_key; c_pos = n_pos; c_rte = n_rte;                        \
                if (vbase + 64 < total) load_meta_next(vbase + 64, n_src, n_key, n_pos, n_rte);     \
                I0N = 0;                                                                           \
                CNTN = min(UN, nb);                                                                \
            }                                                                                      \
        } else {                                                                                   \
            CNTN = min(UN, nb - I0N);                                                              \
        }                                                                                          \
    }

    // Two register buffers: the rows of batch k + 1 are requested before batch k is consumed.
    int i0A = 0, cntA = min(UN, nb), i0B, cntB;
    AGG_ISSUE(vrA, slA, trA, keyA, 0, cntA)
    for (;;) {
        AGG_NEXT(i0A, cntA, i0B, cntB)
        AGG_ISSUE(vrB, slB, trB, keyB, i0B, cntB)
        AGG_WAIT(vrA, slA, trA)
        AGG_PROCESS(vrA, slA, trA, keyA, cntA)
        if (cntB == 0) break;
        AGG_NEXT(i0B, cntB, i0A, cntA)
        AGG_ISSUE(vrA, slA, trA, keyA, i0A, cntA)
        AGG_WAIT(vrB, slB, trB)
        AGG_PROCESS(vrB, slB, trB, keyB, cntB)
        if (cntA == 0) break;
    }
    flush();
    relation_end(cur_rel);
    if constexpr (F16) {
        // did a row leave the fp16 range?  (inf in the tile -> NaN / inf in an accumulator; x * 0 keeps both as NaN)
        float chk = 0.0f;
#pragma unroll
        for (int c = 0; c < NCT; ++c) chk = fmaf(acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3], 0.0f, chk);
        const unsigned long long bad = __builtin_amdgcn_ballot_w64(chk != chk);      // lane -> its target is (lane & 15)
        if (bad == 0 || attempt >= 20) {      // (every retry adds 2^14 of headroom: 18 cover the whole fp32 range -- finite rows always end with bad == 0)
            if (lane < 16) s_sig[lane] = sigv;      // for agg_mfma_finish
            break;
        }
        const unsigned bad_t = (unsigned)((bad | (bad >> 16) | (bad >> 32) | (bad >> 48)) & 0xFFFFull);
        shiftv += ((bad_t >> (lane & 15)) & 1u) ? 14 : 0;
    } else {
        break;
    }
  }
#undef AGG_ISSUE
#undef AGG_LOAD
#undef AGG_WAIT
#undef AGG_META_WAIT
#undef load_meta_next
#undef AGG_PROCESS
#undef AGG_ROW_SEG
#undef AGG_ROW_RESCALE
#undef AGG_ROW_ACC
#undef AGG_NEXT
}

// normalise (PyG softmax denominator, conv.py:108) + optional exact-erf gelu (conv.py:119), in the accumulator layout;
// apply_gelu: 0 = normalise, 1 = normalise + gelu, 2 = raw weighted sum (no softmax: hgt_edge_spmm)
// s_sig / minv: the fp16 split's operand scales (1 / sigma_t per target, inverse fragment scale); nullptr / 1 for the bf16 split
template <int VEC, int LPH>
__device__ __forceinline__ void agg_mfma_finish(const float* s_l, int apply_gelu, f32x4 (&acc)[MG<VEC, LPH>::NCT],
                                                const float* s_sig = nullptr, float minv = 1.0f) {
    using G = MG<VEC, LPH>;
    const int lane = threadIdx.x & 63;
    const int fi = lane & 15, fg = lane >> 4;
    float unscale = minv;
    if (s_sig != nullptr) {
        const float sg = s_sig[fi];
        unscale = (sg != 0.0f) ? minv / sg : 0.0f;       // (a target without a claimed edge never set its scale: its column is 0)
    }
#pragma unroll
    for (int c = 0; c < G::NCT; ++c) {
        const float inv = unscale * ((apply_gelu == 2) ? 1.0f : 1.0f / (s_l[fi * 16 + (16 * c + 4 * fg) / G::DKP] + 1e-16f));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o = acc[c][r] * inv;
            if (apply_gelu == 1) o = 0.5f * o * (1.0f + erff(o * 0.70710678118654752440f));
            acc[c][r] = o;
        }
    }
}

template <int VEC, int LPH>
__device__ __forceinline__ void agg_mfma_store(float* __restrict__ agg, int64_t row0, int SUBR, int64_t NQ, int64_t ld, int co,
                                               unsigned hub_mask, const f32x4 (&acc)[MG<VEC, LPH>::NCT]) {
    using G = MG<VEC, LPH>;
    const int lane = threadIdx.x & 63;
    const int fi = lane & 15, fg = lane >> 4;
    const int64_t row = row0 + fi;
    if (fi >= SUBR || row >= NQ || ((hub_mask >> fi) & 1u)) return;   // hub rows are written by k_hub_finalize
    float* g = agg + row * ld + co + 4 * fg;
#pragma unroll
    for (int c = 0; c < G::NCT; ++c)
        *reinterpret_cast<float4*>(g + 16 * c) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
}

template <int VEC, int LPH>
constexpr int agg_mfma_lds_bytes() { return 4 * 2 * MG<VEC, LPH>::PLANE; }

// inverse fragment scale of the fp16 image: one float behind the fragments (hgt_relation_frag_pack_f16)
template <int VEC, int LPH>
__device__ __forceinline__ float msg_frag_inv_scale(const unsigned short* __restrict__ msgF, int R, int HT) {
    using G = MG<VEC, LPH>;
    return reinterpret_cast<const float*>(msgF + (int64_t)R * (HT / G::H) * G::NCT * G::NKS * 2 * 512)[0];
}

template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_mfma(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, float* __restrict__ agg, int R, int64_t NQ, int apply_gelu,
    int HT, const int32_t* __restrict__ hub_slot, int sub, int64_t ld_out, HgtRelSlice sl) {
    using G = MG<VEC, LPH>;
    const int raw = (apply_gelu == 2);
    __shared__ __attribute__((aligned(16))) unsigned char s_u[4][2 * G::PLANE];
    __shared__ float s_ml[4][2][16 * 16];   // reference / exp-sum per (target, head); H <= 16
    __shared__ float s_scale[4][16];
    __shared__ float s_sig[F16 ? 4 : 1][16];          // fp16 split: the targets' row scales
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* my_sig = s_sig[F16 ? wib : 0];
    const int64_t row0 = sl.q_lo + (int64_t)blockIdx.x * (4 * sub) + wib * sub;
    if (row0 >= NQ) return;
    unsigned hub_mask = 0;
    if (hub_slot) {
        const int64_t rr = row0 + (lane & 15);
        const bool is_hub = (lane < sub) && (rr < NQ) && (hub_slot[rr] >= 0);
        hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    }
    f32x4 acc[G::NCT];
    if (hub_mask == 0 && R < 64)
        agg_mfma_stream<VEC, LPH, RTE, F16>(segptr, esrc, edst, ertei, logits, V, rteV, msgF, R, sl.lo, sl.hi, HT, sub, row0, s_u[wib],
                                            s_ml[wib][0], s_ml[wib][1], s_scale[wib], my_sig, raw, acc);
    else if (hub_mask == 0)
        agg_mfma_subtile<VEC, LPH, RTE, false, F16>(segptr, esrc, edst, ertei, logits, V, rteV, msgF, R, sl.lo, sl.hi, HT, 0u, sub, row0,
                                                    s_u[wib], s_ml[wib][0], s_ml[wib][1], s_scale[wib], my_sig, raw, acc);
    else
        agg_mfma_subtile<VEC, LPH, RTE, true, F16>(segptr, esrc, edst, ertei, logits, V, rteV, msgF, R, sl.lo, sl.hi, HT, hub_mask, sub, row0,
                                                   s_u[wib], s_ml[wib][0], s_ml[wib][1], s_scale[wib], my_sig, raw, acc);
    if (sl.state) {
        // Source-bucketed multi-GPU edge phase: this launch covered one slice of the relation buckets.  Merge with what the
        // earlier slices left -- (reference m, exp-sum l) per (target, head) in sl.state, un-normalised rows in agg -- the way
        // two online-softmax partials combine: m = max(m_a, m_b), x = x_a e^(m_a - m) + x_b e^(m_b - m).
        float* s_m = s_ml[wib][0];
        float* s_l = s_ml[wib][1];
        float* s_sb = reinterpret_cast<float*>(s_u[wib]);      // the U tile is dead: scale of THIS slice's partial
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int hg = blockIdx.y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = j * 64 + lane, t = idx >> 4, hh = idx & 15;
            const int64_t row = row0 + t;
            float m_b = s_m[idx], l_b = s_l[idx], m_p = HGT_NEG, l_p = 0.0f;
            const bool live = (t < sub) && (row < NQ) && (hh < G::H);
            float* st = sl.state + ((int64_t)row * HT + hg * G::H + hh) * 2;
            if (live && sl.has_prev) { m_p = st[0]; l_p = st[1]; }
            const float m = fmaxf(m_b, m_p);
            const float sp = __expf(m_p - m), sb = __expf(m_b - m);
            const float l = l_p * sp + l_b * sb;
            if (live && sl.more) { st[0] = m; st[1] = l; }
            s_m[idx] = sp;
            s_l[idx] = l;
            s_sb[idx] = sb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int fi = lane & 15, fg = lane >> 4;
        const int64_t row = row0 + fi;
        const bool mine = (fi < sub) && (row < NQ) && !((hub_mask >> fi) & 1u);
        const float* g = agg + row * ld_out + (int64_t)hg * G::DP + 4 * fg;
#pragma unroll
        for (int c = 0; c < G::NCT; ++c) {
            const int hh = (16 * c + 4 * fg) / G::DKP;
            const float sb = s_sb[fi * 16 + hh], sp = s_m[fi * 16 + hh];
            float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mine && sl.has_prev) prev = *reinterpret_cast<const float4*>(g + 16 * c);
            acc[c][0] = acc[c][0] * sb + prev.x * sp;
            acc[c][1] = acc[c][1] * sb + prev.y * sp;
            acc[c][2] = acc[c][2] * sb + prev.z * sp;
            acc[c][3] = acc[c][3] * sb + prev.w * sp;
        }
    }
    if (!sl.more) {
        if constexpr (F16) agg_mfma_finish<VEC, LPH>(s_ml[wib][1], apply_gelu, acc, my_sig, msg_frag_inv_scale<VEC, LPH>(msgF, R, HT));
        else agg_mfma_finish<VEC, LPH>(s_ml[wib][1], apply_gelu, acc);
    }
    agg_mfma_store<VEC, LPH>(agg, row0, sub, NQ, ld_out, (int)blockIdx.y * G::DP, hub_mask, acc);
}

// The sub-tile walk of a workgroup that contains a hub target, kept OUT OF LINE (a real call, rare path): inlined next to the
// streaming walk + fused epilogue it costs the common path 22 spilled VGPRs (92 B of scratch per thread = 0.8 GB of extra HBM
// traffic per launch at c2, rocprofv3 WRITE_SIZE).
template <int VEC, int LPH, bool RTE, bool F16>
__device__ __attribute__((noinline)) void agg_mfma_hub_workgroup(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, float* __restrict__ agg, int R, int64_t NQ, int HT,
    unsigned hub_mask, int64_t wrow0, int sub, unsigned char* utile, float* s_m, float* s_l, float* s_sc, float* s_sig) {
    using G = MG<VEC, LPH>;
    f32x4 acc[G::NCT];
    agg_mfma_subtile<VEC, LPH, RTE, true, F16>(segptr, esrc, edst, ertei, logits, V, rteV, msgF, R, 0, R + 1, HT, hub_mask, sub, wrow0, utile,
                                               s_m, s_l, s_sc, s_sig, 0, acc);
    if constexpr (F16) agg_mfma_finish<VEC, LPH>(s_l, 1, acc, s_sig, msg_frag_inv_scale<VEC, LPH>(msgF, R, HT));
    else agg_mfma_finish<VEC, LPH>(s_l, 1, acc);
    agg_mfma_store<VEC, LPH>(agg, wrow0, sub, NQ, (int64_t)HT * G::DKP, 0, hub_mask, acc);
}

// Aggregation + fused node update (see hgt_fused_update.h).  Workgroups that contain a hub target cannot finish their rows
// here: they write agg, raise pending[workgroup], and k_update_pending runs the same epilogue from agg after the hub kernels.
template <int VEC, int LPH, bool RTE, bool F16, bool S32>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_update_mfma(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, float* __restrict__ agg, int R, int64_t NQ, int HT,
    const int32_t* __restrict__ hub_slot, int32_t* __restrict__ pending, HgtFusedUpdate fu) {
    using G = MG<VEC, LPH>;
    static_assert(G::DP <= KP, "the fused epilogue keeps the whole K extent in one LDS slab");
    constexpr int TILES = 4 * 2 * G::PLANE;
    constexpr int FRONT = TILES > 2 * A_PLANE ? TILES : 2 * A_PLANE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[FRONT + 4 * 2 * 256 * 4 + 4 * 16 * 4 + (F16 ? 4 * 16 * 4 : 0)];
    float* s_ml = reinterpret_cast<float*>(smem + FRONT);                 // [4][2][256]; later: the epilogue's tables
    float* s_scale = reinterpret_cast<float*>(smem + FRONT + 4 * 2 * 256 * 4);
    float* s_sig = s_scale + (F16 ? 4 * 16 : 0);                           // fp16 split: row scales of the targets, [4][16]
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = fu.q_lo + (int64_t)blockIdx.x * 64;
    const int64_t wrow0 = row0 + wib * 16;

    unsigned hub_mask = 0;
    if (hub_slot) {
        const int64_t rr = wrow0 + (lane & 15);
        const bool is_hub = (lane < 16) && (rr < NQ) && (hub_slot[rr] >= 0);
        hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    }
    const bool any_hub = hub_slot ? (__syncthreads_or(hub_mask != 0) != 0) : false;
    if (threadIdx.x == 0) pending[fu.q_lo / 64 + blockIdx.x] = any_hub ? 1 : 0;      // (absolute workgroup index: target blocks may run concurrently)

    unsigned char* utile = smem + wib * 2 * G::PLANE;
    float* s_m = s_ml + wib * 512;
    float* s_l = s_m + 256;
    if (any_hub) return;      // k_edge_aggregate_hub_workgroups walks the 64 targets of such a workgroup (same launcher)
    f32x4 acc[G::NCT];
    if (wrow0 < NQ) {
        agg_mfma_stream<VEC, LPH, RTE, F16, true, S32>(segptr, esrc, edst, ertei, logits, V, rteV, msgF, R, 0, R + 1, HT, 16, wrow0, utile, s_m, s_l,
                                                       s_scale + wib * 16, s_sig + wib * 16, 0, acc);
        if constexpr (F16) agg_mfma_finish<VEC, LPH>(s_l, 1, acc, s_sig + wib * 16, msg_frag_inv_scale<VEC, LPH>(msgF, R, HT));
        else agg_mfma_finish<VEC, LPH>(s_l, 1, acc);
    } else {
#pragma unroll
        for (int c = 0; c < G::NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};   // rows beyond NQ: gelu(0) = 0
    }
    // row types for the epilogue, requested before the barrier: the round trip overlaps the wait for the slowest wavefront
    int type_pre = -1;
    if (wib == 0 && row0 + lane < NQ) {
        const int64_t t = fu.node_type[row0 + lane];
        type_pre = (t >= 0 && t < fu.n_types) ? (int)t : -1;
    }
    __syncthreads();   // every wavefront is done with its U tile and its softmax state: the A slab overlays them
    {
        // accumulator layout -> A slab: target (lane & 15) of this wavefront, columns 16 c + 4 (lane >> 4) .. + 3
        const int fi = lane & 15, fg = lane >> 4;
        unsigned char* prow = smem + (wib * 16 + fi) * A_STRIDE + fg * 8;
        float scale = 1.0f;
        if constexpr (F16) {
            // a row (target fi) lives in the four lanes fi, fi + 16, fi + 32, fi + 48: maximum over the lane's 4 * NCT values,
            // then over the four lanes; its inverse scale goes into the epilogue's table (FU_RINV_OFF)
            float m = 0.0f;
#pragma unroll
            for (int c = 0; c < G::NCT; ++c) m = fmaxf(fmaxf(m, fmaxf(fabsf(acc[c][0]), fabsf(acc[c][1]))), fmaxf(fabsf(acc[c][2]), fabsf(acc[c][3])));
            unsigned mb = __builtin_bit_cast(unsigned, m);
            mb = max(mb, (unsigned)__shfl_xor((int)mb, 16));
            mb = max(mb, (unsigned)__shfl_xor((int)mb, 32));
            float inv;
            f16_row_scale(mb, scale, inv);
            if (fg == 0) reinterpret_cast<float*>(smem + FRONT + FU_RINV_OFF)[wib * 16 + fi] = inv;
        }
#pragma unroll
        for (int c = 0; c < G::NCT; ++c) {
            uint2 hi, mid;
            split4_t<F16>(make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]), scale, hi, mid);
            *reinterpret_cast<uint2*>(prow + c * 32) = hi;
            *reinterpret_cast<uint2*>(prow + A_PLANE + c * 32) = mid;
        }
    }
    fused_update_tail<VEC, HGT_FU_NSTG, true, F16>(smem, smem + FRONT, row0, NQ, fu, type_pre);
}

// The 64-target tiles of the fused kernels that contain a hub target (pending[tile] != 0): their non-hub targets are walked run by
// run here, agg is written, and k_update_pending finishes the rows after the hub kernels.  Its own kernel since round 5: as an
// out-of-line call inside the fused kernels it gave every launch of them a 504-byte scratch frame (rocprofv3 Scratch_Size) -- and the
// neighbours of hubs are themselves heavy targets (Zipf(0.8): ~1000 in-edges each), so a tile's walk is cut into wavefronts of
// HGT_HUBWG_SUB targets (32 wavefronts per tile instead of 4: 16 targets of 1000 edges per wavefront were 2.8 ms on their own).
#ifndef HGT_HUBWG_SUB
#define HGT_HUBWG_SUB 2
#endif
template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_hub_workgroups(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const uint16_t* __restrict__ ertei, const float* __restrict__ logits, const float* __restrict__ V,
    const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, float* __restrict__ agg, int R, int64_t NQ, int HT,
    const int32_t* __restrict__ hub_slot, const int32_t* __restrict__ pending, int64_t q_lo) {
    using G = MG<VEC, LPH>;
    constexpr int SUB = HGT_HUBWG_SUB, BPT = 16 / SUB;      // blocks (of 4 wavefronts) per 64-target tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 2 * G::PLANE + 4 * 2 * 256 * 4 + 4 * 16 * 4 + (F16 ? 4 * 16 * 4 : 0)];
    const int64_t tile = blockIdx.x / BPT;
    if (pending[q_lo / 64 + tile] == 0) return;
    float* s_ml = reinterpret_cast<float*>(smem + 4 * 2 * G::PLANE);
    float* s_scale = s_ml + 4 * 2 * 256;
    float* s_sig = s_scale + (F16 ? 4 * 16 : 0);
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wrow0 = q_lo + tile * 64 + (int64_t)(blockIdx.x % BPT) * (4 * SUB) + wib * SUB;
    if (wrow0 >= NQ) return;
    const int64_t rr = wrow0 + (lane & 15);
    const bool is_hub = (lane < SUB) && (rr < NQ) && (hub_slot[rr] >= 0);
    const unsigned hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    float* s_m = s_ml + wib * 512;
    agg_mfma_hub_workgroup<VEC, LPH, RTE, F16>(segptr, esrc, edst, ertei, logits, V, rteV, msgF, agg, R, NQ, HT, hub_mask, wrow0, SUB,
                                               smem + wib * 2 * G::PLANE, s_m, s_m + 256, s_scale + wib * 16, s_sig + wib * 16);
}

// the LDS-ring form of the fused kernel (lab/hgt_edge_agg_ring.h: bit-identical, measured slower -- DESIGN.md section 10): LAB builds only
#if defined(HGT_LAB_KERNELS) && defined(HGT_MFMA_PART_VEC) && HGT_MFMA_PART_VEC == 4 && HGT_MFMA_PART_RTE == 0 && HGT_MFMA_PART_F16 == 0
#define HGT_HAVE_RING 1
#include "lab/hgt_edge_agg_ring.h"
#endif

// ---------------------------------------------------------------------------------------------
// Launchers.  This file is compiled THIRTEEN times (csrc/Makefile): once as the main translation unit (C entry points, dispatch) and
// once per (VEC, RTE, F16) PART, which instantiates the kernels of its five lane layouts -- the 60 x 2 kernel instantiations are then
// built in parallel instead of in one eight-minute hipcc run.  -DHGT_MFMA_PART_VEC=v -DHGT_MFMA_PART_RTE=r -DHGT_MFMA_PART_F16=f selects a part.
// ---------------------------------------------------------------------------------------------
#define HGT_MFMA_AGG_ARGS                                                                                                         \
    const HgtPlanView &pv, const float *logits, const float *V, const float *rteV, const float *msgP, const unsigned short *msgF,  \
        float *agg, int R, int64_t NQ, int apply_gelu, int HT, HgtHubBuffers hb, int64_t ld_out, HgtRelSlice sl, hipStream_t stream
#define HGT_MFMA_AGGUPD_ARGS                                                                                                      \
    const HgtPlanView &pv, const float *logits, const float *V, const float *rteV, const float *msgP, const unsigned short *msgF,  \
        float *agg, int R, int64_t NQ, int HT, HgtHubBuffers hb, int32_t *pending, HgtFusedUpdate fu, hipStream_t stream

#ifdef HGT_MFMA_PART_VEC
template <int VEC, int LPH, bool RTE, bool F16>
static int launch_agg_mfma(HGT_MFMA_AGG_ARGS) {
    // fp16 split: the row scales of the targets are not part of the slice state and the backward's spmm keeps bf16
    if (F16 && (sl.state != nullptr || apply_gelu == 2)) return HGT_ERR_UNSUPPORTED;
    // small graphs (the reference's sampled subgraphs): 4 instead of 16 targets per wavefront -> 4x the wavefronts
    // (round 3: 2 targets per wavefront for sampled-batch sizes when the row is not split over head groups and the schema has
    //  few relations -- c3: 92 -> 82 us per layer; with 33 relations and a head-group split (c5) 4 stays faster: 519 vs 546 us)
#ifndef HGT_SMALL_SUB
#define HGT_SMALL_SUB 2
#endif
    const unsigned ny_ = (unsigned)(HT / (64 / LPH));
    // (2 targets per wavefront only up to a few thousand targets: at 16 000 the 8 000 wavefronts re-read 2 GB of fragments -- 937 us
    //  per layer against 320 us with 4; since round 3 these sizes take hgt_edge_aggregate_items unless it is switched off)
    const int sub = (NQ < 65536) ? ((NQ < 6144 && ny_ == 1 && R <= 16) ? HGT_SMALL_SUB : 4) : HGT_SUB;
    const int64_t tiles = (NQ - sl.q_lo + 4 * sub - 1) / (4 * sub);      // NQ = end of the launch's target range
    const unsigned ny = (unsigned)(HT / (64 / LPH));
    dim3 grid((unsigned)tiles, ny);
    const int32_t* hub_slot = hb.mx ? pv.hub_slot : nullptr;
    k_edge_aggregate_mfma<VEC, LPH, RTE, F16><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgF, agg, R,
                                                                       NQ, apply_gelu, HT, hub_slot, sub, ld_out, sl);
    if (hb.mx && !sl.more)   // hub targets take all their relations at once, after the last slice
        return hgt_launch_hub(VEC, LPH, pv, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, hb, ny, ld_out, stream);
    return HGT_OK;
}

template <int VEC, int LPH, bool RTE, bool F16>
static int launch_aggupd_mfma(HGT_MFMA_AGGUPD_ARGS) {
    if (HT != 64 / LPH) return HGT_ERR_UNSUPPORTED;   // a head-group split leaves a workgroup with part of the row
    if (R >= 64) return HGT_ERR_UNSUPPORTED;          // the streaming walk keeps the R + 1 ranges in lane registers
    const int64_t tiles = (NQ - fu.q_lo + 63) / 64;      // NQ = end of the launch's target range
    dim3 grid((unsigned)tiles, 1);
    const int32_t* hub_slot = hb.mx ? pv.hub_slot : nullptr;
    bool ring = false;
#ifdef HGT_HAVE_RING
    // d = 256 / 8 heads, no temporal rows, bf16 split, on request (HGT_FLAG_RING_AGGREGATE): the ring form (hgt_edge_agg_ring.h)
    if constexpr (VEC == 4 && LPH == 8 && !RTE && !F16) {
      if (fu.ring) {
        ring = true;
        k_edge_aggregate_update_ring<false><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, logits, V, msgF, R, NQ, hub_slot, pending, fu);
      }
    }
#endif
    if (!ring) {
        if (fu.small32)
            k_edge_aggregate_update_mfma<VEC, LPH, RTE, F16, true><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgF,
                                                                                            agg, R, NQ, HT, hub_slot, pending, fu);
        else
            k_edge_aggregate_update_mfma<VEC, LPH, RTE, F16, false><<<grid, 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgF,
                                                                                             agg, R, NQ, HT, hub_slot, pending, fu);
    }
    if (hb.mx) {   // workgroups with a hub target: their other targets, then the hub path + the update of those workgroups
        k_edge_aggregate_hub_workgroups<VEC, LPH, RTE, F16><<<dim3((unsigned)tiles * (16 / HGT_HUBWG_SUB), 1), 256, 0, stream>>>(pv.segptr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgF,
                                                                                     agg, R, NQ, HT, hub_slot, pending, fu.q_lo);
        int rc = hgt_launch_hub(VEC, LPH, pv, logits, V, rteV, msgP, agg, R, NQ, 1, HT, hb, 1u, (int64_t)HT * (VEC * LPH), stream);
        if (rc != HGT_OK) return rc;
        k_update_pending<VEC, F16><<<grid, 256, 0, stream>>>(agg, (int64_t)HT * (VEC * LPH), NQ, pending, fu);
    }
    return HGT_OK;
}
#endif

// head-group split of the matrix-core kernels: the wave's slice must be <= 256 columns (VEC <= 4): 16 KB of U tile per wave
static int mfma_split_for(int vec_full, int lph_full) {
    int s = 1;
    while (vec_full / s > 4 && lph_full * s * 2 <= 64) s *= 2;
    return (vec_full / s <= 4) ? s : 0;   // 0: not covered (one head wider than 256 columns)
}

// det: the deterministic hub mode (one partial slot per piece behind the atomic accumulators; hgt_hub_workspace_bytes_ex)
static HgtHubBuffers carve_hub(void* hub_ws, const HgtPlanView& pv, int H, int64_t E, int dk_pad = 0, int R = 0, bool det = false) {
    HgtHubBuffers hb = {nullptr, nullptr, nullptr};
    if (hub_ws && E > 0) {
        const uint64_t per = hgt_align_up((uint64_t)pv.L.max_hubs * H * 4, 256);
        hb.mx = (int*)hub_ws;
        hb.l = (float*)((char*)hub_ws + per);
        hb.acc = (float*)((char*)hub_ws + 2 * per);
        if (det) {
            const uint64_t acc_bytes = hgt_align_up((uint64_t)pv.L.max_hubs * H * dk_pad * 4, 256);
            const uint64_t np = (uint64_t)(R + 1) * HGT_HUB_PIECES;
            hb.part = (float*)((char*)hub_ws + 2 * per + acc_bytes);
            hb.lpart = (float*)((char*)hb.part + hgt_align_up((uint64_t)pv.L.max_hubs * np * H * dk_pad * 4, 256));
        }
    }
    return hb;
}

}  // namespace

#define HGT_MFMA_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define HGT_MFMA_NAME(base, v, r, f) HGT_MFMA_CAT2(base, v, r_, r, f_, f)
// hgt_mfma_agg_v<VEC>r_<RTE>f_<F16>(lph, ...), hgt_mfma_aggupd_v<VEC>r_<RTE>f_<F16>(lph, ...): one pair per part
#define HGT_MFMA_DECL(v, r, f)                                             \
    __attribute__((visibility("hidden"))) int HGT_MFMA_NAME(hgt_mfma_agg_v, v, r, f)(int lph, HGT_MFMA_AGG_ARGS); \
    __attribute__((visibility("hidden"))) int HGT_MFMA_NAME(hgt_mfma_aggupd_v, v, r, f)(int lph, HGT_MFMA_AGGUPD_ARGS);
#define HGT_MFMA_DECL4(v) HGT_MFMA_DECL(v, 0, 0) HGT_MFMA_DECL(v, 1, 0) HGT_MFMA_DECL(v, 0, 1) HGT_MFMA_DECL(v, 1, 1)
HGT_MFMA_DECL4(1) HGT_MFMA_DECL4(2) HGT_MFMA_DECL4(4)

#ifdef HGT_MFMA_PART_VEC
// ------------------------------------------------------------------------------------------------ a (VEC, RTE, F16) part
#ifdef HGT_DEV_LAYOUTS   // development builds: d = 256 / 8 heads and d = 64 / 4 heads only
#define HGT_MFMA_LPH_CASES(F) \
    if ((HGT_MFMA_PART_VEC == 4 && lph == 8)) return F(8); \
    if ((HGT_MFMA_PART_VEC == 1 && lph == 16)) return F(16);
#else
#define HGT_MFMA_LPH_CASES(F) \
    if (lph == 4) return F(4);  \
    if (lph == 8) return F(8);  \
    if (lph == 16) return F(16); \
    if (lph == 32) return F(32); \
    if (lph == 64) return F(64);
#endif
int HGT_MFMA_NAME(hgt_mfma_agg_v, HGT_MFMA_PART_VEC, HGT_MFMA_PART_RTE, HGT_MFMA_PART_F16)(int lph, HGT_MFMA_AGG_ARGS) {
#define HGT_MFMA_F(L) \
    launch_agg_mfma<HGT_MFMA_PART_VEC, L, (HGT_MFMA_PART_RTE != 0), (HGT_MFMA_PART_F16 != 0)>(pv, logits, V, rteV, msgP, msgF, agg, R, NQ, apply_gelu, HT, hb, ld_out, sl, stream)
    HGT_MFMA_LPH_CASES(HGT_MFMA_F)
#undef HGT_MFMA_F
    return HGT_ERR_UNSUPPORTED;
}
int HGT_MFMA_NAME(hgt_mfma_aggupd_v, HGT_MFMA_PART_VEC, HGT_MFMA_PART_RTE, HGT_MFMA_PART_F16)(int lph, HGT_MFMA_AGGUPD_ARGS) {
#define HGT_MFMA_F(L) \
    launch_aggupd_mfma<HGT_MFMA_PART_VEC, L, (HGT_MFMA_PART_RTE != 0), (HGT_MFMA_PART_F16 != 0)>(pv, logits, V, rteV, msgP, msgF, agg, R, NQ, HT, hb, pending, fu, stream)
    HGT_MFMA_LPH_CASES(HGT_MFMA_F)
#undef HGT_MFMA_F
    return HGT_ERR_UNSUPPORTED;
}
#else
// ------------------------------------------------------------------------------------------------ the main translation unit
namespace {
int mfma_agg_dispatch(int vec, int lph, bool f16, HGT_MFMA_AGG_ARGS) {
#define HGT_MFMA_ARGS_ lph, pv, logits, V, rteV, msgP, msgF, agg, R, NQ, apply_gelu, HT, hb, ld_out, sl, stream
#define HGT_MFMA_CALL(v)                                                                                        \
    return f16 ? (rteV ? hgt_mfma_agg_v##v##r_1f_1(HGT_MFMA_ARGS_) : hgt_mfma_agg_v##v##r_0f_1(HGT_MFMA_ARGS_)) \
               : (rteV ? hgt_mfma_agg_v##v##r_1f_0(HGT_MFMA_ARGS_) : hgt_mfma_agg_v##v##r_0f_0(HGT_MFMA_ARGS_));
    if (vec == 1) { HGT_MFMA_CALL(1) }
    if (vec == 2) { HGT_MFMA_CALL(2) }
    if (vec == 4) { HGT_MFMA_CALL(4) }
#undef HGT_MFMA_CALL
#undef HGT_MFMA_ARGS_
    return HGT_ERR_UNSUPPORTED;      // a wave's slice wider than 256 columns: the vector-ALU kernel
}
int mfma_aggupd_dispatch(int vec, int lph, bool f16, HGT_MFMA_AGGUPD_ARGS) {
#define HGT_MFMA_ARGS_ lph, pv, logits, V, rteV, msgP, msgF, agg, R, NQ, HT, hb, pending, fu, stream
#define HGT_MFMA_CALL(v)                                                                                              \
    return f16 ? (rteV ? hgt_mfma_aggupd_v##v##r_1f_1(HGT_MFMA_ARGS_) : hgt_mfma_aggupd_v##v##r_0f_1(HGT_MFMA_ARGS_)) \
               : (rteV ? hgt_mfma_aggupd_v##v##r_1f_0(HGT_MFMA_ARGS_) : hgt_mfma_aggupd_v##v##r_0f_0(HGT_MFMA_ARGS_));
    if (vec == 1) { HGT_MFMA_CALL(1) }
    if (vec == 2) { HGT_MFMA_CALL(2) }
    if (vec == 4) { HGT_MFMA_CALL(4) }
#undef HGT_MFMA_CALL
#undef HGT_MFMA_ARGS_
    return HGT_ERR_UNSUPPORTED;
}

template <bool F16>
int relation_frag_pack_impl(const float* msg_p, int32_t R, int32_t H, int32_t dk_pad, void* msg_frag, void* stream) {
    if (!msg_p || !msg_frag || R <= 0 || H <= 0 || dk_pad <= 0 || 64 % H != 0) return HGT_ERR_INVALID_ARG;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    const int sp = mfma_split_for(dk_pad / lph, lph);
    if (sp == 0) return HGT_ERR_UNSUPPORTED;
    const int vec = dk_pad / lph / sp;
    const int dp = 64 * vec;
    // after a head-group split a "head" of the kernel is still a real head (heads are never cut): the block structure of
    // blockdiag(M) is described by the REAL head width dk_pad
    const int kw = dk_pad > 32 ? dk_pad : 32;
    const int64_t total = (int64_t)R * (H * dk_pad / dp) * (dp / 16) * (kw / 32) * 512;
    float* tail = reinterpret_cast<float*>((unsigned short*)msg_frag + total * 2);      // [inverse scale, scale]
    if (F16) k_msg_scale<<<1, 1024, 0, (hipStream_t)stream>>>(msg_p, (int64_t)R * H * dk_pad * dk_pad, tail);
    k_msg_frag_pack<F16><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(msg_p, R, H, dk_pad, dp, (unsigned short*)msg_frag, tail);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
}  // namespace

extern "C" int hgt_relation_frag_bytes(int32_t R, int32_t H, int32_t dk_pad, uint64_t* out) {
    if (!out || R <= 0 || H <= 0 || dk_pad <= 0 || 64 % H != 0) return HGT_ERR_INVALID_ARG;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    const int sp = mfma_split_for(dk_pad / lph, lph);
    if (sp == 0) { *out = 0; return HGT_OK; }      // layout not covered by the matrix-core kernel: no image needed
    const int vec = dk_pad / lph / sp, lphs = lph * sp, dkp = vec * lphs, dp = 64 * vec;
    const int kw = dkp > 32 ? dkp : 32;
    *out = (uint64_t)R * (H * dk_pad / dp) * (dp / 16) * (kw / 32) * 2 * 512 * 2 + 256;   // + the fp16 image's scale pair
    return HGT_OK;
}

extern "C" int hgt_relation_frag_pack(const float* msg_p, int32_t R, int32_t H, int32_t dk_pad, void* msg_frag, void* stream) {
    return relation_frag_pack_impl<false>(msg_p, R, H, dk_pad, msg_frag, stream);
}
extern "C" int hgt_relation_frag_pack_f16(const float* msg_p, int32_t R, int32_t H, int32_t dk_pad, void* msg_frag, void* stream) {
    return relation_frag_pack_impl<true>(msg_p, R, H, dk_pad, msg_frag, stream);
}

extern "C" int hgt_hub_workspace_bytes(int64_t n_edges, int32_t n_heads, int32_t dk_pad, uint64_t* out) {
    if (!out || n_edges < 0 || n_heads <= 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const uint64_t max_hubs = (uint64_t)(n_edges / HGT_HUB_DEG + 1);
    *out = hgt_align_up(max_hubs * n_heads * 4, 256) * 2 + hgt_align_up(max_hubs * n_heads * dk_pad * 4, 256);
    return HGT_OK;
}
// ABI 6: + the per-piece partial slots of the deterministic hub mode (n_relations + 1 buckets x 32 pieces per hub)
extern "C" int hgt_hub_workspace_bytes_ex(int64_t n_edges, int32_t n_heads, int32_t dk_pad, int32_t n_relations, int32_t deterministic,
                                          uint64_t* out) {
    int rc = hgt_hub_workspace_bytes(n_edges, n_heads, dk_pad, out);
    if (rc != HGT_OK || !deterministic) return rc;
    if (n_relations <= 0) return HGT_ERR_INVALID_ARG;
    const uint64_t max_hubs = (uint64_t)(n_edges / HGT_HUB_DEG + 1), np = (uint64_t)(n_relations + 1) * HGT_HUB_PIECES;
    *out += hgt_align_up(max_hubs * np * n_heads * dk_pad * 4, 256) + hgt_align_up(max_hubs * np * n_heads * 4, 256);
    return HGT_OK;
}

static int edge_aggregate_impl(bool f16, const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                               const float* logits, const float* V, const float* rte_v, const float* msg_p, const void* msg_frag,
                               float* agg, int64_t n_q_rows, int32_t apply_gelu, void* hub_ws, void* stream, bool det_hubs = false) {
    if (f16 && !msg_frag) return HGT_ERR_INVALID_ARG;
    if (!plan || !V || !msg_p || !agg || (E > 0 && !logits) || H <= 0 || 64 % H != 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    HgtHubBuffers hb = carve_hub(hub_ws, pv, H, E, dk_pad, R, det_hubs);
    const HgtRelSlice whole = {0, (int)R + 1, nullptr, 0, 0};
    int rc = HGT_ERR_UNSUPPORTED;
    const int sp = msg_frag ? mfma_split_for(dk_pad / lph, lph) : 0;
    if (sp != 0)
        rc = mfma_agg_dispatch(dk_pad / lph / sp, lph * sp, f16, pv, logits, V, rte_v, msg_p, (const unsigned short*)msg_frag, agg,
                                            (int)R, NQ, (int)(apply_gelu ? 1 : 0), (int)H, hb, (int64_t)H * dk_pad, whole,
                                            (hipStream_t)stream);
    if (rc == HGT_ERR_UNSUPPORTED)   // exact-fp32 request (msg_frag == NULL) or a layout only the vector-ALU kernel covers
        rc = hgt_valu_aggregate(pv, dk_pad, logits, V, rte_v, msg_p, agg, (int)R, NQ, (int)(apply_gelu ? 1 : 0), (int)H, hb, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
extern "C" int hgt_edge_aggregate(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                  const float* logits, const float* V, const float* rte_v, const float* msg_p, const void* msg_frag,
                                  float* agg, int64_t n_q_rows, int32_t apply_gelu, void* hub_ws, void* stream) {
    return edge_aggregate_impl(false, plan, N, E, T, R, H, dk_pad, logits, V, rte_v, msg_p, msg_frag, agg, n_q_rows, apply_gelu, hub_ws, stream);
}
// msg_frag = the hgt_relation_frag_pack_f16 image (required); layouts only the vector-ALU kernel covers run in exact fp32
extern "C" int hgt_edge_aggregate_f16x3(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                        const float* logits, const float* V, const float* rte_v, const float* msg_p,
                                        const void* msg_frag, float* agg, int64_t n_q_rows, int32_t apply_gelu, void* hub_ws,
                                        void* stream) {
    return edge_aggregate_impl(true, plan, N, E, T, R, H, dk_pad, logits, V, rte_v, msg_p, msg_frag, agg, n_q_rows, apply_gelu, hub_ws, stream);
}

// ABI 6: hub_deterministic != 0: hub targets are accumulated without atomics (hub_ws of hgt_hub_workspace_bytes_ex(.., 1) bytes)
extern "C" int hgt_edge_aggregate_ex(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                     const float* logits, const float* V, const float* rte_v, const float* msg_p, const void* msg_frag,
                                     int32_t frag_f16, float* agg, int64_t n_q_rows, int32_t apply_gelu, void* hub_ws,
                                     int32_t hub_deterministic, void* stream) {
    return edge_aggregate_impl(frag_f16 != 0, plan, N, E, T, R, H, dk_pad, logits, V, rte_v, msg_p, msg_frag, agg, n_q_rows, apply_gelu, hub_ws,
                               stream, hub_deterministic != 0);
}

// One slice [rel_lo, rel_hi) of the relation buckets (include/hgt_hip.h): matrix-core kernel only (msg_frag required).
extern "C" int hgt_edge_aggregate_slice(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                        const float* logits, const float* V, const float* rte_v, const float* msg_p,
                                        const void* msg_frag, float* agg, int64_t n_q_rows, int32_t apply_gelu, void* hub_ws,
                                        int32_t rel_lo, int32_t rel_hi, float* state, int32_t has_prev, int32_t more, void* stream) {
    if (!plan || !V || !msg_p || !msg_frag || !agg || !state || (E > 0 && !logits) || H <= 0 || 64 % H != 0 || dk_pad <= 0)
        return HGT_ERR_INVALID_ARG;
    if (rel_lo < 0 || rel_hi > R + 1 || rel_lo > rel_hi) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    HgtHubBuffers hb = carve_hub(hub_ws, pv, H, E);
    const HgtRelSlice sl = {(int)rel_lo, (int)rel_hi, state, has_prev ? 1 : 0, more ? 1 : 0};
    const int sp = mfma_split_for(dk_pad / lph, lph);
    if (sp == 0) return HGT_ERR_UNSUPPORTED;
    int rc = mfma_agg_dispatch(dk_pad / lph / sp, lph * sp, false, pv, logits, V, rte_v, msg_p, (const unsigned short*)msg_frag, agg,
                                            (int)R, NQ, (int)(apply_gelu ? 1 : 0), (int)H, hb, (int64_t)H * dk_pad, sl, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

static int edge_aggregate_update_impl(bool f16, const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                      const float* logits, const float* V, const float* rte_v, const float* msg_p,
                                      const void* msg_frag, float* agg, int64_t n_q_rows, void* hub_ws, int32_t* pending,
                                      const int64_t* node_type, const void* w_a_split, const float* b_a, const float* x_skip,
                                      int64_t ld_skip, const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm,
                                      int32_t n_out, float* out, void* stream, int64_t q_begin = 0, int64_t q_end = -1,
                                      bool det_hubs = false, bool use_ring = false) {
    if (f16 && !msg_frag) return HGT_ERR_INVALID_ARG;
    if (!plan || !V || !msg_p || !agg || !pending || !node_type || !w_a_split || !b_a || !x_skip || !skip || !out || H <= 0 ||
        64 % H != 0 || dk_pad <= 0 || n_out <= 0)
        return HGT_ERR_INVALID_ARG;
    if (E > 0 && !logits) return HGT_ERR_INVALID_ARG;
    if (use_norm && (!ln_w || !ln_b)) return HGT_ERR_INVALID_ARG;
    int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    const bool ranged = (q_end >= 0);      // a target block [q_begin, q_end) of the multi-GPU path (include/hgt_hip.h, ABI 6)
    if (ranged) {
        if (q_begin < 0 || q_begin > q_end || q_end > NQ || (q_begin % HGT_TD) != 0) return HGT_ERR_INVALID_ARG;
        if (!msg_frag) return HGT_ERR_UNSUPPORTED;      // the vector-ALU kernels always walk the whole graph
        NQ = q_end;
        if (q_begin == q_end) return HGT_OK;
    }
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    const int dp = H * dk_pad;
    if (dp > KP || n_out > dp || (n_out & 3) != 0 || (ld_skip & 3) != 0 || ((uintptr_t)x_skip & 15) != 0) return HGT_ERR_UNSUPPORTED;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    HgtHubBuffers hb = carve_hub(hub_ws, pv, H, E, dk_pad, R, det_hubs);
    HgtFusedUpdate fu = {node_type, (const unsigned short*)w_a_split, b_a, x_skip, ld_skip, skip, ln_w, ln_b, use_norm, T, n_out, out,
                         ranged ? q_begin : 0, use_ring ? 1 : 0,
                         // every gathered row / logit / temporal row sits below 4 GiB from its base: the fused kernels address them
                         // with 32-bit lane offsets
                         ((uint64_t)N * (uint64_t)dp * 4u < (1ull << 32) && (uint64_t)E * (uint64_t)H * 4u < (1ull << 32)) ? 1 : 0};
    if (ranged) { hb.q_lo = q_begin; hb.q_hi = q_end; }
    int rc = HGT_ERR_UNSUPPORTED;
    if (msg_frag && mfma_split_for(dk_pad / lph, lph) == 1)
        rc = mfma_aggupd_dispatch(dk_pad / lph, lph, f16, pv, logits, V, rte_v, msg_p, (const unsigned short*)msg_frag, agg,
                                                  (int)R, NQ, (int)H, hb, pending, fu, (hipStream_t)stream);
    if (rc == HGT_ERR_UNSUPPORTED && !msg_frag)
        rc = hgt_valu_aggregate_update(pv, dk_pad, logits, V, rte_v, msg_p, agg, (int)R, NQ, (int)H, hb, pending, fu, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
#define HGT_AGGUPD_PARAMS                                                                                                            \
    const void *plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad, const float *logits, const float *V,     \
        const float *rte_v, const float *msg_p, const void *msg_frag, float *agg, int64_t n_q_rows, void *hub_ws, int32_t *pending,   \
        const int64_t *node_type, const void *w_a_split, const float *b_a, const float *x_skip, int64_t ld_skip, const float *skip,   \
        const float *ln_w, const float *ln_b, int32_t use_norm, int32_t n_out, float *out, void *stream
#define HGT_AGGUPD_PASS                                                                                                              \
    plan, N, E, T, R, H, dk_pad, logits, V, rte_v, msg_p, msg_frag, agg, n_q_rows, hub_ws, pending, node_type, w_a_split, b_a, x_skip, \
        ld_skip, skip, ln_w, ln_b, use_norm, n_out, out, stream
extern "C" int hgt_edge_aggregate_update(HGT_AGGUPD_PARAMS) { return edge_aggregate_update_impl(false, HGT_AGGUPD_PASS); }
// msg_frag = hgt_relation_frag_pack_f16 image, w_a_split = hgt_split_weights_f16 image (both required)
extern "C" int hgt_edge_aggregate_update_f16x3(HGT_AGGUPD_PARAMS) { return edge_aggregate_update_impl(true, HGT_AGGUPD_PASS); }
// ABI 6: the targets [q_begin, q_end) only (q_begin a multiple of the plan tile): one target block of the multi-GPU path
// frag_f16: msg_frag / w_a_split are the fp16 images; hub_deterministic: hubs without atomics (hgt_hub_workspace_bytes_ex(.., 1))
extern "C" int hgt_edge_aggregate_update_range(HGT_AGGUPD_PARAMS, int64_t q_begin, int64_t q_end, int32_t frag_f16,
                                               int32_t hub_deterministic) {
    if (q_end < 0) return HGT_ERR_INVALID_ARG;
    return edge_aggregate_update_impl(frag_f16 != 0, HGT_AGGUPD_PASS, q_begin, q_end, hub_deterministic != 0);
}

int hgt_edge_aggregate_update_sel(HGT_AGGUPD_PARAMS, int64_t q_begin, int64_t q_end, int32_t frag_f16, int32_t hub_deterministic, int32_t use_ring) {
    return edge_aggregate_update_impl(frag_f16 != 0, HGT_AGGUPD_PASS, q_begin, q_end, hub_deterministic != 0, use_ring != 0);
}

// out[i][ld_out] = sum_rel ( sum_{e in (i,rel)} w_e rows[src_e] ) F[rel]  -- the aggregation kernel without the softmax: the edge
// weights are given.  The backward pass is three of these (include/hgt_hip.h).
extern "C" int hgt_edge_spmm(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                             const float* weights, const float* rows, const float* rte_rows, const float* f_p, const void* f_frag,
                             float* out, int64_t ld_out, int64_t n_q_rows, void* hub_ws, void* stream) {
    if (!plan || !rows || !f_p || !f_frag || !out || (E > 0 && !weights) || H <= 0 || 64 % H != 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0 || ld_out < (int64_t)H * dk_pad || (ld_out & 3) != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    HgtHubBuffers hb = carve_hub(hub_ws, pv, H, E);
    const int sp = mfma_split_for(dk_pad / lph, lph);
    if (sp == 0) return HGT_ERR_UNSUPPORTED;
    int rc = mfma_agg_dispatch(dk_pad / lph / sp, lph * sp, false, pv, weights, rows, rte_rows, f_p, (const unsigned short*)f_frag, out,
                                            (int)R, NQ, 2, (int)H, hb, ld_out, HgtRelSlice{0, (int)R + 1, nullptr, 0, 0}, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
#endif   // main translation unit
