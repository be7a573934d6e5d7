// Attention logits (conv.py:98-99,104) with the target-side relation transform on the matrix cores.
//
//     s_e,h = < q~_(i,r),h , k_j,h >,   q~_(i,r) = q_i blockdiag_h(A'[r,h])        (hgt_edge_logits.hip: algebra, A' = att_t)
//
// k_edge_logits evaluates q~ once per (target, relation) segment as a d_k x d_k mat-vec on the vector ALU with the relation's
// matrix held in registers.  That is the right form for d_k <= 32 (128 floats per lane: at the benchmark size the kernel runs at
// the gather rate of the K rows), but for d_k = 64 -- the width of the reference's own training configurations (n_hid 400 or 512
// with 8 heads) -- the matrix only fits after a 4-way head-group split (512-byte row slices per wavefront) and the mat-vecs cost
// 4x the vector instructions per edge: the kernel is instruction-bound at half the gather rate.
//
// Here a wavefront takes the same work item (<= 256 edges of one (target tile, relation), sorted by target), numbers the
// distinct targets it meets ("slots", in order of appearance: the items are target-sorted, so a slot is a run of edges) and
// processes them 16 at a time, like a relation end of the aggregation kernel (hgt_edge_agg_mfma.hip) run backwards:
//   A. the 16 Q rows -> split hi/mid (bf16, or fp16 with a power-of-two row scale) -> wave-private LDS tile
//   B. q~^T = A'^T-fragments x Q^T on v_mfma_f32_16x16x32 (3 products per step, fragments of hgt_relation_frag_pack(att_t)
//      streamed from L2), accumulators -> the same LDS bytes as an fp32 [16][DP + 4] tile
//   C. the edges of the 16 slots: gather K (+ temporal row), read q~ of the edge's slot from LDS, dot, reduce over the head's
//      lanes, store -- the per-edge part of k_edge_logits without the per-edge Q row.
// The wavefront covers DP = 64 * VEC <= 256 columns (1 KB row slices); wider rows use blockIdx.y head groups.
#include "hgt_edge_common.h"
#include "hgt_split_common.h"

#ifndef HGT_LOGITS_XCD
#define HGT_LOGITS_XCD 1
#endif
#ifndef HGT_LGM_GS
#define HGT_LGM_GS 8      // column-tile steps whose fragments are requested together (16 loads in flight)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// workgroup barrier that waits for LDS traffic only (rows requested before it stay in flight)
__device__ __forceinline__ void coop_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int VEC, int LPH>
struct LG {   // geometry of one wavefront's slice (same tile layout as MG in hgt_edge_agg_mfma.hip)
    static constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH;
    static constexpr int NCT = DP / 16;                 // 16-column output tiles
    static constexpr int KW = DKP > 32 ? DKP : 32;      // k window of a column tile
    static constexpr int NKS = KW / 32;                 // MFMA k-steps per column tile
    static constexpr int ROWB = DP * 2;                 // bytes of one 16-bit row of the Q tile
    static constexpr int NS = DP / 8;                   // 16-byte slots per row
    static constexpr int PLANE = 16 * ROWB;             // one 16-bit plane of the tile (16 targets)
    static constexpr int QS = DP + 4;                   // floats per row of the fp32 q~ tile (+4: the 16 rows start on different banks)
    static constexpr int WAVE_LDS = 16 * QS * 4;        // bytes per wavefront (>= 2 * PLANE)
};

template <int VEC>
__device__ __forceinline__ unsigned abs_bits_vec(const float (&v)[VEC]) {
    float m = fabsf(v[0]);
#pragma unroll
    for (int i = 1; i < VEC; ++i) m = fmaxf(m, fabsf(v[i]));
    return __builtin_bit_cast(unsigned, m);
}

template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 2) void k_edge_logits_mfma(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ Q,
    const float* __restrict__ K, const float* __restrict__ rteK, const unsigned short* __restrict__ attF, float* __restrict__ logits,
    int R, int HT, int rel_lo, int rel_hi, int item_lo, int item_hi, int items_cap) {
    using G = LG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS, QS = G::QS;
    constexpr int UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>(), HB = UN / 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][G::WAVE_LDS];
    __shared__ __attribute__((aligned(16))) float s_qinv[F16 ? 4 : 1][16], s_qscale[F16 ? 4 : 1][16];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if HGT_LOGITS_XCD     // XCD-aware item order, see k_edge_logits
    // (chunks of HGT_XCD_CHUNK workgroups, dealt to the XCDs in turn: contiguous EIGHTHS of the list put all the heavy items of a
    //  skewed graph -- its hub tiles come first -- on one XCD: Zipf(0.8) logits 2.2 -> 3.2 ms)
    constexpr int XC = 16;
    const int q8 = (int)(blockIdx.x >> 3), vblock = (q8 / XC) * (8 * XC) + (int)(blockIdx.x & 7u) * XC + (q8 % XC);
#else
    const int vblock = blockIdx.x;
#endif
    const int item = item_lo + vblock * 4 + wib;
    const HgtItem it = items[min(item, items_cap - 1)];              // (requested together with the header's item count: see k_edge_logits)
    const int n_items = item_hi >= 0 ? item_hi : hdr->n_items;      // (item_lo, item_hi): the items of a target block, or (0, -1)
    if (item >= n_items) return;
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    if (rel < rel_lo || rel >= rel_hi) return;
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP, NY = HT / H;
    const int h = lane / LPH, p = lane % LPH;
    if (rel >= R) {   // edges no meta relation claims: logit 0 (conv.py:68)
        for (int64_t i = (int64_t)beg * H + lane; i < (int64_t)end * H; i += 64) logits[(i / H) * HT + hg * H + (i % H)] = 0.0f;
        return;
    }

    unsigned char* tile = smem[wib];
    float* qtile = reinterpret_cast<float*>(tile);
    const int fi = lane & 15, fg = lane >> 4;
    const int wb = lane * VEC * 2;            // byte offset of this lane's 16-bit elements inside a tile row
    const int rrow = fi * ROWB;
    const unsigned short* __restrict__ mf = attF + (((int64_t)rel * NY + hg) * NCT) * NKS * 2 * 512 + lane * 8;
    float ainv = 1.0f;                         // inverse scale of the fp16 fragment image (behind the fragments)
    if constexpr (F16) ainv = reinterpret_cast<const float*>(attF + (int64_t)R * NY * NCT * NKS * 2 * 512)[0];

    constexpr int STEPS = NCT * NKS, GSW = (STEPS >= 32 ? 2 : 1) * HGT_LGM_GS, GS = GSW < STEPS ? GSW : STEPS, NG = STEPS / GS;
    static_assert(STEPS % GS == 0, "column-tile steps come in multiples of 4");
    // (d_k >= 64, 32 steps: groups of 16 -- two fragment round trips per transform instead of four; the LDS tile caps the kernel at
    //  two wavefronts per SIMD anyway, so the 64 extra registers are free)

    for (int base = beg; base < end; base += 64) {
        const int nb = min(64, end - base);
        const int li = base + min(lane, nb - 1);
        const int my_src = esrc[li], my_dst = edst[li];
        const int my_rte = RTE ? (int)ertei[li] : 0;
        // slots: distinct targets of the chunk in order of appearance (lanes beyond the chunk replicate its last edge)
        const int prev_dst = __shfl_up(my_dst, 1);
        const bool lead = (lane == 0) || (my_dst != prev_dst);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(lead);
        const int my_slot = __builtin_popcountll(mask & (~0ull >> (63 - lane))) - 1;
        const int nd = __builtin_popcountll(mask);
        unsigned long long mrem = mask;       // leaders of the slots not yet loaded
        int lead_idx = 0;

        for (int t0 = 0; t0 < nd; t0 += 16) {
            // ---- A. Q rows of slots [t0, t0 + 16) (slots beyond the chunk's last re-read its last leader: rows nobody reads)
            float qrow[16][VEC];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (mrem != 0ull) {
                    lead_idx = __builtin_ctzll(mrem);
                    mrem &= mrem - 1ull;
                }
                const int d_ = __builtin_amdgcn_readlane(my_dst, lead_idx);
                load_vec<VEC>(Q + (int64_t)d_ * ld + co + lane * VEC, qrow[r]);
            }
            // the edges of the group: lanes [e_lo, e_lo + e_cnt)
            const unsigned long long in_g = __builtin_amdgcn_ballot_w64(my_slot >= t0 && my_slot < t0 + 16 && lane < nb);
            const int e_lo = __builtin_ctzll(in_g), e_end = e_lo + __builtin_popcountll(in_g);
            float krA[HB][VEC], trA[RTE ? HB : 1][VEC], krB[HB][VEC], trB[RTE ? HB : 1][VEC];
#define LGM_ISSUE(KR, TR, I0)                                                                      \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        const int idx = min((I0) + u, e_end - 1);                                                  \
        const int s_ = __builtin_amdgcn_readlane(my_src, idx);                                     \
        load_vec<VEC>(K + (int64_t)s_ * ld + co + lane * VEC, KR[u]);                              \
        if constexpr (RTE) {                                                                       \
            const int ri = __builtin_amdgcn_readlane(my_rte, idx);                                 \
            load_vec<VEC>(rteK + (int64_t)ri * ld + co + lane * VEC, TR[u]);                       \
        }                                                                                          \
    }
#define LGM_PROCESS(KR, TR, I0)                                                                    \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        if ((I0) + u < e_end) {                                                                    \
            const int r_ = __builtin_amdgcn_readlane(my_slot, (I0) + u) - t0;                      \
            float qt[VEC];                                                                         \
            load_vec<VEC>(qtile + r_ * QS + lane * VEC, qt);                                       \
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
            // the first K rows of the group are requested before the transform (behind the Q rows in the load queue: the split
            // below waits for the Q rows only), so their latency is covered by phases A and B
            LGM_ISSUE(krA, trA, e_lo)

            // fp16 split: the 16 row maxima together through the (still unused) tile: see k_edge_logits_coop
            float scl[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) scl[r] = 1.0f;
            if constexpr (F16) {
                unsigned* mx_ = reinterpret_cast<unsigned*>(tile);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx_[r * 64 + ((((lane >> 2) ^ r) << 2) | (lane & 3))] = abs_bits_vec<VEC>(qrow[r]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int r_ = lane & 15, q_ = lane >> 4;
                unsigned m = 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint4 v4 = *reinterpret_cast<const uint4*>(mx_ + r_ * 64 + (((q_ * 4 + c) ^ r_) << 2));
                    m = max(max(m, max(v4.x, v4.y)), max(v4.z, v4.w));
                }
                m = max(m, (unsigned)__shfl_xor((int)m, 16));
                m = max(m, (unsigned)__shfl_xor((int)m, 32));
                float sc_l, inv_l;
                f16_row_scale(m, sc_l, inv_l);
                if (lane < 16) {
                    s_qinv[wib][lane] = inv_l;
                    s_qscale[wib][lane] = sc_l;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();         // (also: every lane has read the maxima before the split rows overwrite them)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 s4 = *reinterpret_cast<const float4*>(&s_qscale[wib][4 * c]);
                    scl[4 * c] = s4.x; scl[4 * c + 1] = s4.y; scl[4 * c + 2] = s4.z; scl[4 * c + 3] = s4.w;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float scale = scl[r];
                unsigned char* w = tile + r * ROWB + ((((wb >> 4) ^ (r & (NS - 1)))) << 4) + (wb & 15);
                if constexpr (VEC == 1) {
                    unsigned short hi, mid;
                    split1_t<F16>(qrow[r][0], scale, hi, mid);
                    *reinterpret_cast<unsigned short*>(w) = hi;
                    *reinterpret_cast<unsigned short*>(w + G::PLANE) = mid;
                } else if constexpr (VEC == 2) {
                    unsigned hi, mid;
                    split2_t<F16>(qrow[r][0], qrow[r][1], scale, hi, mid);
                    *reinterpret_cast<unsigned*>(w) = hi;
                    *reinterpret_cast<unsigned*>(w + G::PLANE) = mid;
                } else {
                    uint2 hi, mid;
                    split4_t<F16>(make_float4(qrow[r][0], qrow[r][1], qrow[r][2], qrow[r][3]), scale, hi, mid);
                    *reinterpret_cast<uint2*>(w) = hi;
                    *reinterpret_cast<uint2*>(w + G::PLANE) = mid;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- B. q~^T = fragments x Q^T
            f32x4 acc[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 fh[GS], fm[GS];
            // the fragments do not depend on the group: left visible, hipcc hoists all 2 * STEPS loads out of both loops (256
            // registers for d_k = 64 -> 245 spilled).  They are meant to be re-read from L2 per group, like a relation end of
            // the aggregation kernel.
            const unsigned short* mfg = mf;
            asm volatile("" : "+v"(mfg));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    const unsigned short* t_ = mfg + (int64_t)((g * GS + j) * 2) * 512;
                    fh[j] = *reinterpret_cast<const bf16x8*>(t_);
                    fm[j] = *reinterpret_cast<const bf16x8*>(t_ + 512);
                }
                __builtin_amdgcn_sched_barrier(0);   // all 2 GS loads of the group are issued before the first MFMA waits
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    const int step = g * GS + j, c = step / NKS, ks = step % NKS;
                    const int kbase = (16 * c / KW) * KW;
                    const int slot = (kbase + 32 * ks) / 8 + fg;
                    const unsigned char* up = tile + rrow + ((slot ^ (fi & (NS - 1))) << 4);
                    const bf16x8 uh = *reinterpret_cast<const bf16x8*>(up);
                    const bf16x8 um = *reinterpret_cast<const bf16x8*>(up + G::PLANE);
                    acc[c] = mfma16_t<F16>(fm[j], uh, acc[c]);
                    acc[c] = mfma16_t<F16>(fh[j], um, acc[c]);
                    acc[c] = mfma16_t<F16>(fh[j], uh, acc[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();         // every lane has read the 16-bit tile: its bytes become the fp32 q~ tile
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                float sc = 1.0f;
                if constexpr (F16) sc = s_qinv[wib][fi] * ainv;
                // accumulator layout: lane = (target fi, row group fg): columns 16 c + 4 fg .. + 3 of target fi
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    *reinterpret_cast<float4*>(qtile + fi * QS + 16 * c + 4 * fg) =
                        make_float4(acc[c][0] * sc, acc[c][1] * sc, acc[c][2] * sc, acc[c][3] * sc);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- C. the edges of the group, two half-batches in flight (fixed-size unconditional issues: counted vmcnt)
            for (int i0 = e_lo; i0 < e_end; i0 += 2 * HB) {
                LGM_ISSUE(krB, trB, i0 + HB)
                LGM_PROCESS(krA, trA, i0)
                LGM_ISSUE(krA, trA, i0 + 2 * HB)
                LGM_PROCESS(krB, trB, i0 + HB)
            }
#undef LGM_ISSUE
#undef LGM_PROCESS
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();         // the q~ tile is rewritten by the next group only after every lane has read it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// The same kernel with the TRANSFORM shared by the workgroup (round 6).  Above, every wavefront streams the relation's whole fragment
// image of its head group per 16 slots (d_k = 64: 64 KB; with the 16-edge items of a sampled batch that is 4 KB of fragments per edge
// next to 2 KB of rows: redirected to one cache-resident tile the kernel is 10 us of 40 faster at c5, 6.7 of 27 for the 4-layer model).
// Here the four wavefronts of a workgroup still own one work item each for phases A and C, but phase B is split by COLUMNS:
// wavefront w computes column tiles [w NCT/4, (w + 1) NCT/4) of q~ for the slots of ALL four items -- a quarter of a relation's image
// (d_k = 64: one head, 16 KB = 64 registers), kept in registers while consecutive items share the relation (they are sorted by
// (tile, relation)): 16-64 KB of fragment loads per workgroup and round instead of 256.  Three workgroup barriers per round of 16 slots.
template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 2) void k_edge_logits_coop(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ Q,
    const float* __restrict__ K, const float* __restrict__ rteK, const unsigned short* __restrict__ attF, float* __restrict__ logits,
    int R, int HT, int rel_lo, int rel_hi, int item_lo, int item_hi, int items_cap) {
    using G = LG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS, QS = G::QS;
    constexpr int UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>(), HB = UN / 2;
    constexpr int CTW = NCT / 4, SW = CTW * NKS;      // column tiles / fragment steps of one wavefront's share of the transform
    static_assert(NCT % 4 == 0 && SW <= 8, "a wavefront's share of the fragment image stays in registers");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][G::WAVE_LDS];
    __shared__ int s_rel[4];
    __shared__ __attribute__((aligned(16))) float s_qinv[F16 ? 4 : 1][16], s_qscale[F16 ? 4 : 1][16];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if HGT_LOGITS_XCD
    constexpr int XC = 16;
    const int q8 = (int)(blockIdx.x >> 3), vblock = (q8 / XC) * (8 * XC) + (int)(blockIdx.x & 7u) * XC + (q8 % XC);
#else
    const int vblock = blockIdx.x;
#endif
    const int item = item_lo + vblock * 4 + wib;
    const HgtItem it = items[min(item, items_cap - 1)];
    const int n_items = item_hi >= 0 ? item_hi : hdr->n_items;
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP, NY = HT / H;
    const int h = lane / LPH, p = lane % LPH;
    bool more = item < n_items && rel >= rel_lo && rel < rel_hi && beg < end;
    if (more && rel >= R) {   // edges no meta relation claims: logit 0 (conv.py:68)
        for (int64_t i = (int64_t)beg * H + lane; i < (int64_t)end * H; i += 64) logits[(i / H) * HT + hg * H + (i % H)] = 0.0f;
        more = false;
    }

    unsigned char* tile = smem[wib];
    float* qtile = reinterpret_cast<float*>(tile);
    const int fi = lane & 15, fg = lane >> 4;
    const int wb = lane * VEC * 2;
    const int rrow = fi * ROWB;
    float ainv = 1.0f;
    if constexpr (F16) ainv = reinterpret_cast<const float*>(attF + (int64_t)R * NY * NCT * NKS * 2 * 512)[0];

    bf16x8 fh[SW], fm[SW];
#pragma unroll
    for (int j = 0; j < SW; ++j) fh[j] = fm[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    int have = -1;                    // the relation whose fragments fh / fm hold
    // (the first item's quarter requested up front, before the rows: measured slower -- 38.5 vs 37.8 us at c5 -- the requests delay the rows)
    int base = beg - 64, t0 = 0, nd = 0, nb = 0;
    int my_src = 0, my_dst = 0, my_rte = 0, my_slot = 0, lead_idx = 0;
    unsigned long long mrem = 0ull;

    for (;;) {
        const bool active = more;
        int e_lo = 0, e_end = 0;
        float krA[HB][VEC], trA[RTE ? HB : 1][VEC], krB[HB][VEC], trB[RTE ? HB : 1][VEC];
#define LGC_ISSUE(KR, TR, I0)                                                                      \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        const int idx = min((I0) + u, e_end - 1);                                                  \
        const int s_ = __builtin_amdgcn_readlane(my_src, idx);                                     \
        load_vec<VEC>(K + (int64_t)s_ * ld + co + lane * VEC, KR[u]);                              \
        if constexpr (RTE) {                                                                       \
            const int ri = __builtin_amdgcn_readlane(my_rte, idx);                                 \
            load_vec<VEC>(rteK + (int64_t)ri * ld + co + lane * VEC, TR[u]);                       \
        }                                                                                          \
    }
#define LGC_PROCESS(KR, TR, I0)                                                                    \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        if ((I0) + u < e_end) {                                                                    \
            const int r_ = __builtin_amdgcn_readlane(my_slot, (I0) + u) - t0;                      \
            float qt[VEC];                                                                         \
            load_vec<VEC>(qtile + r_ * QS + lane * VEC, qt);                                       \
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
        if (active) {
            if (t0 >= nd) {           // next 64-edge chunk of the item
                base += 64;
                nb = min(64, end - base);
                const int li = base + min(lane, nb - 1);
                my_src = esrc[li];
                my_dst = edst[li];
                my_rte = RTE ? (int)ertei[li] : 0;
                const int prev_dst = __shfl_up(my_dst, 1);
                const bool lead = (lane == 0) || (my_dst != prev_dst);
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(lead);
                my_slot = __builtin_popcountll(mask & (~0ull >> (63 - lane))) - 1;
                nd = __builtin_popcountll(mask);
                mrem = mask;
                lead_idx = 0;
                t0 = 0;
            }
            // ---- A. Q rows of slots [t0, t0 + 16) -> split -> this wavefront's 16-bit tile
            float qrow[16][VEC];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (mrem != 0ull) {
                    lead_idx = __builtin_ctzll(mrem);
                    mrem &= mrem - 1ull;
                }
                const int d_ = __builtin_amdgcn_readlane(my_dst, lead_idx);
                load_vec<VEC>(Q + (int64_t)d_ * ld + co + lane * VEC, qrow[r]);
            }
            const unsigned long long in_g = __builtin_amdgcn_ballot_w64(my_slot >= t0 && my_slot < t0 + 16 && lane < nb);
            e_lo = __builtin_ctzll(in_g);
            e_end = e_lo + __builtin_popcountll(in_g);
            LGC_ISSUE(krA, trA, e_lo)
            // fp16 split: the 16 row maxima TOGETHER (one wave reduction per row -- 8 DPP steps, 4 v_readlane and the scale on the scalar
            // unit, sixteen times -- was 5.6 of c5's 38 us): every lane parks its 16 per-row maxima in the tile (free between rounds,
            // 16-byte chunks XOR-swizzled by row), lane (r = l & 15, q = l >> 4) reduces the 16 lanes of quarter q for row r with four
            // 16-byte reads, two cross-quarter steps, and the scales come back through a 16-float table.
            float scl[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) scl[r] = 1.0f;
            if constexpr (F16) {
                unsigned* mx_ = reinterpret_cast<unsigned*>(tile);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx_[r * 64 + ((((lane >> 2) ^ r) << 2) | (lane & 3))] = abs_bits_vec<VEC>(qrow[r]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int r_ = lane & 15, q_ = lane >> 4;
                unsigned m = 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint4 v4 = *reinterpret_cast<const uint4*>(mx_ + r_ * 64 + (((q_ * 4 + c) ^ r_) << 2));
                    m = max(max(m, max(v4.x, v4.y)), max(v4.z, v4.w));
                }
                m = max(m, (unsigned)__shfl_xor((int)m, 16));
                m = max(m, (unsigned)__shfl_xor((int)m, 32));
                float sc_l, inv_l;
                f16_row_scale(m, sc_l, inv_l);
                if (lane < 16) {
                    s_qinv[wib][lane] = inv_l;
                    s_qscale[wib][lane] = sc_l;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();         // (also: every lane has read the maxima before the split rows overwrite them)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 s4 = *reinterpret_cast<const float4*>(&s_qscale[wib][4 * c]);
                    scl[4 * c] = s4.x; scl[4 * c + 1] = s4.y; scl[4 * c + 2] = s4.z; scl[4 * c + 3] = s4.w;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float scale = scl[r];
                unsigned char* w = tile + r * ROWB + ((((wb >> 4) ^ (r & (NS - 1)))) << 4) + (wb & 15);
                if constexpr (VEC == 1) {
                    unsigned short hi, mid;
                    split1_t<F16>(qrow[r][0], scale, hi, mid);
                    *reinterpret_cast<unsigned short*>(w) = hi;
                    *reinterpret_cast<unsigned short*>(w + G::PLANE) = mid;
                } else if constexpr (VEC == 2) {
                    unsigned hi, mid;
                    split2_t<F16>(qrow[r][0], qrow[r][1], scale, hi, mid);
                    *reinterpret_cast<unsigned*>(w) = hi;
                    *reinterpret_cast<unsigned*>(w + G::PLANE) = mid;
                } else {
                    uint2 hi, mid;
                    split4_t<F16>(make_float4(qrow[r][0], qrow[r][1], qrow[r][2], qrow[r][3]), scale, hi, mid);
                    *reinterpret_cast<uint2*>(w) = hi;
                    *reinterpret_cast<uint2*>(w + G::PLANE) = mid;
                }
            }
        }
        if (lane == 0) s_rel[wib] = active ? rel : -1;
        coop_barrier();
        int rk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rk[k] = __builtin_amdgcn_readfirstlane(s_rel[k]);
        if ((rk[0] & rk[1] & rk[2] & rk[3]) < 0) break;      // no wavefront of the workgroup has slots left

        // ---- B. this wavefront's column tiles of q~^T = fragments x Q^T, for the tiles of all four wavefronts
        f32x4 acc[4][CTW];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < CTW; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rk[k] < 0) continue;
            if (rk[k] != have) {
                have = rk[k];
                const unsigned short* mf = attF + ((((int64_t)have * NY + hg) * NCT + wib * CTW) * NKS) * 2 * 512 + lane * 8;
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    fh[j] = *reinterpret_cast<const bf16x8*>(mf + (int64_t)(j * 2) * 512);
                    fm[j] = *reinterpret_cast<const bf16x8*>(mf + (int64_t)(j * 2 + 1) * 512);
                }
            }
            const unsigned char* tk = smem[k] + rrow;
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int cc = j / NKS, ks = j % NKS;
                const int c = wib * CTW + cc;
                const int kbase = (16 * c / KW) * KW;
                const int slot = (kbase + 32 * ks) / 8 + fg;
                const unsigned char* up = tk + ((slot ^ (fi & (NS - 1))) << 4);
                const bf16x8 uh = *reinterpret_cast<const bf16x8*>(up);
                const bf16x8 um = *reinterpret_cast<const bf16x8*>(up + G::PLANE);
                acc[k][cc] = mfma16_t<F16>(fm[j], uh, acc[k][cc]);
                acc[k][cc] = mfma16_t<F16>(fh[j], um, acc[k][cc]);
                acc[k][cc] = mfma16_t<F16>(fh[j], uh, acc[k][cc]);
            }
        }
        coop_barrier();               // every wavefront has read the 16-bit tiles: their bytes become the fp32 q~ tiles
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rk[k] < 0) continue;
            float sc = 1.0f;
            if constexpr (F16) sc = s_qinv[k][fi] * ainv;
            float* qk = reinterpret_cast<float*>(smem[k]) + fi * QS + 16 * (wib * CTW) + 4 * fg;
#pragma unroll
            for (int c = 0; c < CTW; ++c)
                *reinterpret_cast<float4*>(qk + 16 * c) = make_float4(acc[k][c][0] * sc, acc[k][c][1] * sc, acc[k][c][2] * sc, acc[k][c][3] * sc);
        }
        coop_barrier();

        // ---- C. the edges of this wavefront's group
        if (active) {
            for (int i0 = e_lo; i0 < e_end; i0 += 2 * HB) {
                LGC_ISSUE(krB, trB, i0 + HB)
                LGC_PROCESS(krA, trA, i0)
                LGC_ISSUE(krA, trA, i0 + 2 * HB)
                LGC_PROCESS(krB, trB, i0 + HB)
            }
            t0 += 16;
            more = (t0 < nd) || (base + 64 < end);
        }
#undef LGC_ISSUE
#undef LGC_PROCESS
        // (the next round's phase A rewrites this wavefront's OWN tile only; s_rel / s_qinv are rewritten behind barriers every reader
        //  of this round has passed)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int VEC, int LPH>
static int launch_logits_mfma(int mode, const HgtPlanView& pv, const float* Q, const float* K, const float* rteK, const unsigned short* attF,
                              float* logits, int R, int HT, int rel_lo, int rel_hi, int item_lo, int item_hi, hipStream_t stream) {
    const bool f16 = (mode & 1) != 0;
    const int item_edges = mode >> 8;
    const int64_t n_launch = item_hi >= 0 ? (int64_t)(item_hi - item_lo) : pv.L.max_items;
    const unsigned blocks = ((unsigned)((n_launch + 3) / 4) + 127u) & ~127u;      // (a multiple of 8 XCDs x 16: XCD-aware item order)
    dim3 grid(blocks, (unsigned)(HT / (64 / LPH)));
    using G = LG<VEC, LPH>;
    // d_k >= 64 and the 16-edge items of a sampled batch: the transform shared by the workgroup (a wavefront's quarter of the fragment
    // image fits its registers).  Larger items run several lock-step rounds per workgroup and measured SLOWER than a wavefront each
    // (N = 500 k, d = 512: 3.61 vs 3.30 ms with four 256-edge items per workgroup, the same with the chunks of ONE item per workgroup,
    // i.e. one relation and one fragment request per item: at that size the fragment stream is not what bounds the kernel).
    if constexpr (G::DKP >= 64 && G::NCT % 4 == 0 && (G::NCT / 4) * G::NKS <= 8) {
        if (!(mode & 2) && ((mode & 4) || item_edges <= 16)) {
#define LGC_LAUNCH(RTE_, F16_)                                                                                                   \
    k_edge_logits_coop<VEC, LPH, RTE_, F16_><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attF, \
                                                                       logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, (int)pv.L.max_items)
            if (rteK) { if (f16) LGC_LAUNCH(true, true); else LGC_LAUNCH(true, false); }
            else      { if (f16) LGC_LAUNCH(false, true); else LGC_LAUNCH(false, false); }
#undef LGC_LAUNCH
            return HGT_OK;
        }
    }
#define LGM_LAUNCH(RTE_, F16_)                                                                                                   \
    k_edge_logits_mfma<VEC, LPH, RTE_, F16_><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, rteK, attF, \
                                                                       logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, (int)pv.L.max_items)
    if (rteK) { if (f16) LGM_LAUNCH(true, true); else LGM_LAUNCH(true, false); }
    else      { if (f16) LGM_LAUNCH(false, true); else LGM_LAUNCH(false, false); }
#undef LGM_LAUNCH
    return HGT_OK;
}

}  // namespace

// vec / lph: the wavefront's layout after the head-group split of the matrix-core kernels (<= 256 columns per wavefront)
__attribute__((visibility("hidden"))) int hgt_launch_logits_mfma(int vec, int lph, int mode, const HgtPlanView& pv, const float* Q,
                                                                 const float* K, const float* rteK, const unsigned short* attF,
                                                                 float* logits, int R, int HT, int rel_lo, int rel_hi, int item_lo, int item_hi,
                                                                 hipStream_t stream) {
#define LGM_CASE(V, L) \
    if (vec == V && lph == L) return launch_logits_mfma<V, L>(mode, pv, Q, K, rteK, attF, logits, R, HT, rel_lo, rel_hi, item_lo, item_hi, stream);
#ifdef HGT_DEV_LAYOUTS
    LGM_CASE(4, 8) LGM_CASE(4, 16)
#else
    // d_k >= 64 (the layouts the vector-ALU kernel has to split into narrow head groups) + the benchmark layout for A/B runs
    LGM_CASE(4, 8) LGM_CASE(4, 16) LGM_CASE(4, 32) LGM_CASE(4, 64) LGM_CASE(2, 32) LGM_CASE(2, 64) LGM_CASE(1, 64)
#endif
#undef LGM_CASE
    return HGT_ERR_UNSUPPORTED;
}
