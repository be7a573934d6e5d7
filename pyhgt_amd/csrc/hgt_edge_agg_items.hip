// Aggregation for the latency regime (sampled sub-graphs: a few thousand targets), item-parallel and deterministic.
//
// The sub-tile kernels (hgt_edge_agg_mfma.hip) give a wavefront a few targets and ALL their relations: one dependent chain of
// edge batches (~2.5 us per 6-8 edges under load) and relation ends (fragment loads from L2 + MFMAs, ~3.5 us each).  That chain
// is hidden behind 2 000 other wavefronts on a million-node graph, but it IS the kernel time on a 4 000-node batch: with the 33
// relations of the OAG schema a wavefront runs ~25 relation ends for ~40 edges (102 us per layer at c5, its largest kernel).
//
// Here the parallelism is the logits kernels': one wavefront per work item = <= 256 target-sorted edges of ONE (target tile,
// relation).  A "run" is the maximal stretch of consecutive edges of one target inside a 64-edge chunk of an item; the
// wavefront sums  u = sum_e e^(s_e - m_run) v_e  per run (online softmax with the run's own reference), parks the u rows of up
// to 16 runs in a wave-private LDS tile, transforms them with the relation's message fragments on the matrix cores (ONE
// round per 16 runs) and writes every transformed row, with the run's (m_run, l_run) per head, at the position of the run's
// first edge in a scratch [E][d] (k_edge_runs_mfma).  A second kernel (k_merge_runs) owns one target per wavefront, walks the
// target's segments in (relation, position) order and combines the runs exactly like two softmax partials combine
// (m = max, x = sum x_run e^(m_run - m)), adds the edges no meta relation claims (logit 0, no message), normalises, applies
// gelu and stores the row -- fixed order, no atomics: a forward is bit-reproducible.  Hub targets need no separate path.
#include "hgt_edge_common.h"
#include "hgt_split_common.h"
#include "hgt_wt_store.h"

#ifndef HGT_LOGITS_XCD
#define HGT_LOGITS_XCD 1
#endif
#ifndef HGT_AGI_GS
#define HGT_AGI_GS 8      // column-tile steps whose fragments are requested together (16 loads in flight)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC, int LPH>
struct AG {   // geometry of one wavefront's slice (tile layout of MG in hgt_edge_agg_mfma.hip)
    static constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH;
    static constexpr int NCT = DP / 16, KW = DKP > 32 ? DKP : 32, NKS = KW / 32;
    static constexpr int ROWB = DP * 2, NS = DP / 8, PLANE = 16 * ROWB;
};

template <int VEC>
__device__ __forceinline__ unsigned abs_bits_v(const float (&v)[VEC]) {
    float m = fabsf(v[0]);
#pragma unroll
    for (int i = 1; i < VEC; ++i) m = fmaxf(m, fabsf(v[i]));
    return __builtin_bit_cast(unsigned, m);
}

template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 2) void k_edge_runs_mfma(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ logits,
    const float* __restrict__ V, const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, float* __restrict__ zrows,
    float* __restrict__ zstat, unsigned char* __restrict__ zflag, int R, int HT, int raw, int items_cap) {      // raw: `logits` ARE the edge weights (hgt_edge_spmm_items)
    using G = AG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS;
    constexpr int UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>(), HB = UN / 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][2 * G::PLANE];
    __shared__ int s_pos[4][16];
    __shared__ __attribute__((aligned(16))) float s_rinv[F16 ? 4 : 1][16][4];      // [wavefront][run][quarter of the row (16 lanes)]

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if HGT_LOGITS_XCD     // XCD-aware item order, see k_edge_logits
    constexpr int XC = 16;
    const int q8 = (int)(blockIdx.x >> 3), vblock = (q8 / XC) * (8 * XC) + (int)(blockIdx.x & 7u) * XC + (q8 % XC);
#else
    const int vblock = blockIdx.x;
#endif
    const int item = vblock * 4 + wib;
    const HgtItem it = items[min(item, items_cap - 1)];      // (requested together with the header's item count: see k_edge_logits)
    const int n_items = hdr->n_items;
    if (item >= n_items) return;
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    if (rel >= R) return;                      // edges no meta relation claims carry no message: k_merge_runs counts them
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP, NY = HT / H;
    const int h = lane / LPH, p = lane % LPH;

    unsigned char* tile = smem[wib];
    const int fi = lane & 15, fg = lane >> 4;
    const int wb = lane * VEC * 2;
    const int rrow = fi * ROWB;
    const unsigned short* __restrict__ mf = msgF + (((int64_t)rel * NY + hg) * NCT) * NKS * 2 * 512 + lane * 8;
    float minv = 1.0f;
    if constexpr (F16) minv = reinterpret_cast<const float*>(msgF + (int64_t)R * NY * NCT * NKS * 2 * 512)[0];

    constexpr int STEPS = NCT * NKS, GSW = (STEPS >= 32 ? 2 : 1) * HGT_AGI_GS, GS = GSW < STEPS ? GSW : STEPS, NG = STEPS / GS;
    static_assert(STEPS % GS == 0, "column-tile steps come in multiples of 4");
    // (d_k >= 64, 32 steps: groups of 16 -- two fragment round trips per transform instead of four; the LDS tile caps the kernel at
    //  two wavefronts per SIMD anyway, so the 64 extra registers are free)

    for (int base = beg; base < end; base += 64) {
        const int nb = min(64, end - base);
        const int li = base + min(lane, nb - 1);
        const int my_src = esrc[li], my_dst = edst[li];
        const int my_rte = RTE ? (int)ertei[li] : 0;
        const int prev_dst = __shfl_up(my_dst, 1);
        const bool lead = (lane == 0) || (my_dst != prev_dst);      // (lanes beyond the chunk replicate its last edge: never leaders)
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(lead);
        const int my_slot = __builtin_popcountll(mask & (~0ull >> (63 - lane))) - 1;
        const int nd = __builtin_popcountll(mask);
        if (hg == 0 && lane < nb) zflag[base + lane] = lead ? 1 : 0;

        for (int t0 = 0; t0 < nd; t0 += 16) {
            const unsigned long long in_g = __builtin_amdgcn_ballot_w64(my_slot >= t0 && my_slot < t0 + 16 && lane < nb);
            const int e_lo = __builtin_ctzll(in_g), e_end = e_lo + __builtin_popcountll(in_g);
            const int nrows = min(16, nd - t0);
            if (lead && my_slot >= t0 && my_slot < t0 + 16) s_pos[wib][my_slot - t0] = base + lane;

            // ---- runs of the group: u = sum e^(s - m_run) v, parked in the tile (split hi / mid)
            int cur_r = -1, cur_pos = 0;
            float U[VEC], m_run = HGT_NEG, l_run = 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
            auto flush = [&]() {
                if (cur_r < 0) return;
                float scale = 1.0f;
                if constexpr (F16) {
                    // one power-of-two scale per QUARTER of the row where a head does not span quarters (LPH <= 16: the transform is
                    // block-diagonal by head, any scale uniform over a head's columns factors out of its products): four DPP steps
                    // and the rule in the vector ALU instead of a wave reduction through v_readlane and the scalar unit on the
                    // chain every run of a one-edge segment goes through
                    float inv;
                    if constexpr (LPH <= 16) {
                        f16_row_scale(row16_max_bits(abs_bits_v<VEC>(U)), scale, inv);
                        if ((lane & 15) == 0) s_rinv[wib][cur_r][lane >> 4] = inv;
                    } else {
                        f16_row_scale(wave_max_bits(abs_bits_v<VEC>(U)), scale, inv);
                        if (lane < 4) s_rinv[wib][cur_r][lane] = inv;
                    }
                }
                unsigned char* w = tile + cur_r * ROWB + ((((wb >> 4) ^ (cur_r & (NS - 1)))) << 4) + (wb & 15);
                if constexpr (VEC == 1) {
                    unsigned short hi, mid;
                    split1_t<F16>(U[0], scale, hi, mid);
                    *reinterpret_cast<unsigned short*>(w) = hi;
                    *reinterpret_cast<unsigned short*>(w + G::PLANE) = mid;
                } else if constexpr (VEC == 2) {
                    unsigned hi, mid;
                    split2_t<F16>(U[0], U[1], scale, hi, mid);
                    *reinterpret_cast<unsigned*>(w) = hi;
                    *reinterpret_cast<unsigned*>(w + G::PLANE) = mid;
                } else {
                    uint2 hi, mid;
                    split4_t<F16>(make_float4(U[0], U[1], U[2], U[3]), scale, hi, mid);
                    *reinterpret_cast<uint2*>(w) = hi;
                    *reinterpret_cast<uint2*>(w + G::PLANE) = mid;
                }
                if (p == 0) *reinterpret_cast<float2*>(zstat + ((int64_t)cur_pos * HT + hg * H + h) * 2) = make_float2(m_run, l_run);
            };
            float vrA[HB][VEC], trA[RTE ? HB : 1][VEC], slA[HB], vrB[HB][VEC], trB[RTE ? HB : 1][VEC], slB[HB];
#define AGI_ISSUE(VR, TR, SL, I0)                                                                  \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        const int idx = min((I0) + u, e_end - 1);                                                  \
        const int s_ = __builtin_amdgcn_readlane(my_src, idx);                                     \
        load_vec<VEC>(V + (int64_t)s_ * ld + co + lane * VEC, VR[u]);                              \
        SL[u] = logits[(int64_t)(base + idx) * HT + hg * H + h];                                   \
        if constexpr (RTE) {                                                                       \
            const int ri = __builtin_amdgcn_readlane(my_rte, idx);                                 \
            load_vec<VEC>(rteV + (int64_t)ri * ld + co + lane * VEC, TR[u]);                       \
        }                                                                                          \
    }
#define AGI_PROCESS(VR, TR, SL, I0)                                                                \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        if ((I0) + u < e_end) {                                                                    \
            const int r_ = __builtin_amdgcn_readlane(my_slot, (I0) + u) - t0;                      \
            if (r_ != cur_r) {                                                                     \
                flush();                                                                           \
                cur_r = r_;                                                                        \
                cur_pos = base + (I0) + u;                                                         \
                m_run = HGT_NEG;                                                                   \
                l_run = 0.0f;                                                                      \
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) U[i] = 0.0f;                       \
            }                                                                                      \
            const float m_new = raw ? 0.0f : fmaxf(m_run, SL[u]);                                  \
            const float sc = raw ? 1.0f : __expf(m_run - m_new), pe = raw ? SL[u] : __expf(SL[u] - m_new); \
            _Pragma("unroll") for (int i = 0; i < VEC; ++i) {                                      \
                float vv = VR[u][i];                                                               \
                if constexpr (RTE) vv += TR[u][i];                                                 \
                U[i] = fmaf(pe, vv, U[i] * sc);                                                    \
            }                                                                                      \
            l_run = fmaf(l_run, sc, pe);                                                           \
            m_run = m_new;                                                                         \
        }                                                                                          \
    }
            AGI_ISSUE(vrA, trA, slA, e_lo)
            for (int i0 = e_lo; i0 < e_end; i0 += 2 * HB) {
                AGI_ISSUE(vrB, trB, slB, i0 + HB)
                AGI_PROCESS(vrA, trA, slA, i0)
                AGI_ISSUE(vrA, trA, slA, i0 + 2 * HB)
                AGI_PROCESS(vrB, trB, slB, i0 + HB)
            }
#undef AGI_ISSUE
#undef AGI_PROCESS
            flush();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- z^T = message fragments x u^T (rows the group does not use hold stale bytes: every column of the transposed
            //      product depends on its own row only, and those columns are not stored)
            f32x4 acc[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 fh[GS], fm[GS];
            const unsigned short* mfg = mf;
            asm volatile("" : "+v"(mfg));          // (keeps hipcc from hoisting all 2 * STEPS fragment loads out of the loops)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    const unsigned short* t_ = mfg + (int64_t)((g * GS + j) * 2) * 512;
                    fh[j] = *reinterpret_cast<const bf16x8*>(t_);
                    fm[j] = *reinterpret_cast<const bf16x8*>(t_ + 512);
                }
                __builtin_amdgcn_sched_barrier(0);
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
            // ---- transformed rows -> scratch, at the position of the run's first edge
            if (fi < nrows) {
                float sc4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if constexpr (F16) {
                    const float4 q4 = *reinterpret_cast<const float4*>(s_rinv[wib][fi]);
                    sc4[0] = q4.x * minv; sc4[1] = q4.y * minv; sc4[2] = q4.z * minv; sc4[3] = q4.w * minv;
                }
                float* zr = zrows + (int64_t)s_pos[wib][fi] * ld + co + 4 * fg;
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    const float sc = sc4[c / (NCT / 4)];      // (column tile c lies in quarter c / VEC of the row)
                    store_wt16(zr + 16 * c, acc[c][0] * sc, acc[c][1] * sc, acc[c][2] * sc, acc[c][3] * sc);      // (read by the merge kernel only)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();         // the tile / position table are rewritten by the next group
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// k_edge_runs_mfma with the TRANSFORM shared by the workgroup (see k_edge_logits_coop: same reason, same split).  Every wavefront
// walks the runs of its own work item and parks them in its tile; then wavefront w is the column tiles [w NCT/4, (w + 1) NCT/4) of
// z^T = fragments x u^T for the tiles of all four -- a quarter of the relation's image in registers, re-requested only when the
// relation changes between consecutive items.  Rows, statistics and flags are the single-wavefront kernel's, bit for bit.
__device__ __forceinline__ void coop_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 2) void k_edge_runs_coop(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ logits,
    const float* __restrict__ V, const float* __restrict__ rteV, const unsigned short* __restrict__ msgF, float* __restrict__ zrows,
    float* __restrict__ zstat, unsigned char* __restrict__ zflag, int R, int HT, int raw, int items_cap) {
    using G = AG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS;
    constexpr int UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>(), HB = UN / 2;
    constexpr int CTW = NCT / 4, SW = CTW * NKS;
    static_assert(NCT % 4 == 0 && SW <= 8, "a wavefront's share of the fragment image stays in registers");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][2 * G::PLANE];
    __shared__ int s_pos[4][16];
    __shared__ int s_rel[4], s_nrows[4];
    __shared__ __attribute__((aligned(16))) float s_rinv[F16 ? 4 : 1][16][4];      // [wavefront][run][quarter of the row (16 lanes)]

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if HGT_LOGITS_XCD
    constexpr int XC = 16;
    const int q8 = (int)(blockIdx.x >> 3), vblock = (q8 / XC) * (8 * XC) + (int)(blockIdx.x & 7u) * XC + (q8 % XC);
#else
    const int vblock = blockIdx.x;
#endif
    const int item = vblock * 4 + wib;
    const HgtItem it = items[min(item, items_cap - 1)];
    const int n_items = hdr->n_items;
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    bool more = item < n_items && rel < R && beg < end;      // (edges no meta relation claims carry no message: k_merge_runs counts them)
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP, NY = HT / H;
    const int h = lane / LPH, p = lane % LPH;

    unsigned char* tile = smem[wib];
    const int fi = lane & 15, fg = lane >> 4;
    const int wb = lane * VEC * 2;
    const int rrow = fi * ROWB;
    float minv = 1.0f;
    if constexpr (F16) minv = reinterpret_cast<const float*>(msgF + (int64_t)R * NY * NCT * NKS * 2 * 512)[0];

    bf16x8 fh[SW], fm[SW];
#pragma unroll
    for (int j = 0; j < SW; ++j) fh[j] = fm[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    int have = -1;
    int base = beg - 64, t0 = 0, nd = 0, nb = 0;
    int my_src = 0, my_rte = 0, my_slot = 0;
    bool lead = false;

    for (;;) {
        const bool active = more;
        int nrows = 0;
        if (active) {
            if (t0 >= nd) {
                base += 64;
                nb = min(64, end - base);
                const int li = base + min(lane, nb - 1);
                my_src = esrc[li];
                const int my_dst = edst[li];
                my_rte = RTE ? (int)ertei[li] : 0;
                const int prev_dst = __shfl_up(my_dst, 1);
                lead = (lane == 0) || (my_dst != prev_dst);
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(lead);
                my_slot = __builtin_popcountll(mask & (~0ull >> (63 - lane))) - 1;
                nd = __builtin_popcountll(mask);
                if (hg == 0 && lane < nb) zflag[base + lane] = lead ? 1 : 0;
                t0 = 0;
            }
            const unsigned long long in_g = __builtin_amdgcn_ballot_w64(my_slot >= t0 && my_slot < t0 + 16 && lane < nb);
            const int e_lo = __builtin_ctzll(in_g), e_end = e_lo + __builtin_popcountll(in_g);
            nrows = min(16, nd - t0);
            if (lead && my_slot >= t0 && my_slot < t0 + 16) s_pos[wib][my_slot - t0] = base + lane;

            int cur_r = -1, cur_pos = 0;
            float U[VEC], m_run = HGT_NEG, l_run = 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
            auto flush = [&]() {
                if (cur_r < 0) return;
                float scale = 1.0f;
                if constexpr (F16) {
                    // one power-of-two scale per QUARTER of the row where a head does not span quarters (LPH <= 16: the transform is
                    // block-diagonal by head, any scale uniform over a head's columns factors out of its products): four DPP steps
                    // and the rule in the vector ALU instead of a wave reduction through v_readlane and the scalar unit on the
                    // chain every run of a one-edge segment goes through
                    float inv;
                    if constexpr (LPH <= 16) {
                        f16_row_scale(row16_max_bits(abs_bits_v<VEC>(U)), scale, inv);
                        if ((lane & 15) == 0) s_rinv[wib][cur_r][lane >> 4] = inv;
                    } else {
                        f16_row_scale(wave_max_bits(abs_bits_v<VEC>(U)), scale, inv);
                        if (lane < 4) s_rinv[wib][cur_r][lane] = inv;
                    }
                }
                unsigned char* w = tile + cur_r * ROWB + ((((wb >> 4) ^ (cur_r & (NS - 1)))) << 4) + (wb & 15);
                if constexpr (VEC == 1) {
                    unsigned short hi, mid;
                    split1_t<F16>(U[0], scale, hi, mid);
                    *reinterpret_cast<unsigned short*>(w) = hi;
                    *reinterpret_cast<unsigned short*>(w + G::PLANE) = mid;
                } else if constexpr (VEC == 2) {
                    unsigned hi, mid;
                    split2_t<F16>(U[0], U[1], scale, hi, mid);
                    *reinterpret_cast<unsigned*>(w) = hi;
                    *reinterpret_cast<unsigned*>(w + G::PLANE) = mid;
                } else {
                    uint2 hi, mid;
                    split4_t<F16>(make_float4(U[0], U[1], U[2], U[3]), scale, hi, mid);
                    *reinterpret_cast<uint2*>(w) = hi;
                    *reinterpret_cast<uint2*>(w + G::PLANE) = mid;
                }
                if (p == 0) *reinterpret_cast<float2*>(zstat + ((int64_t)cur_pos * HT + hg * H + h) * 2) = make_float2(m_run, l_run);
            };
            float vrA[HB][VEC], trA[RTE ? HB : 1][VEC], slA[HB], vrB[HB][VEC], trB[RTE ? HB : 1][VEC], slB[HB];
#define AGC_ISSUE(VR, TR, SL, I0)                                                                  \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        const int idx = min((I0) + u, e_end - 1);                                                  \
        const int s_ = __builtin_amdgcn_readlane(my_src, idx);                                     \
        load_vec<VEC>(V + (int64_t)s_ * ld + co + lane * VEC, VR[u]);                              \
        SL[u] = logits[(int64_t)(base + idx) * HT + hg * H + h];                                   \
        if constexpr (RTE) {                                                                       \
            const int ri = __builtin_amdgcn_readlane(my_rte, idx);                                 \
            load_vec<VEC>(rteV + (int64_t)ri * ld + co + lane * VEC, TR[u]);                       \
        }                                                                                          \
    }
#define AGC_PROCESS(VR, TR, SL, I0)                                                                \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        if ((I0) + u < e_end) {                                                                    \
            const int r_ = __builtin_amdgcn_readlane(my_slot, (I0) + u) - t0;                      \
            if (r_ != cur_r) {                                                                     \
                flush();                                                                           \
                cur_r = r_;                                                                        \
                cur_pos = base + (I0) + u;                                                         \
                m_run = HGT_NEG;                                                                   \
                l_run = 0.0f;                                                                      \
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) U[i] = 0.0f;                       \
            }                                                                                      \
            const float m_new = raw ? 0.0f : fmaxf(m_run, SL[u]);                                  \
            const float sc = raw ? 1.0f : __expf(m_run - m_new), pe = raw ? SL[u] : __expf(SL[u] - m_new); \
            _Pragma("unroll") for (int i = 0; i < VEC; ++i) {                                      \
                float vv = VR[u][i];                                                               \
                if constexpr (RTE) vv += TR[u][i];                                                 \
                U[i] = fmaf(pe, vv, U[i] * sc);                                                    \
            }                                                                                      \
            l_run = fmaf(l_run, sc, pe);                                                           \
            m_run = m_new;                                                                         \
        }                                                                                          \
    }
            AGC_ISSUE(vrA, trA, slA, e_lo)
            for (int i0 = e_lo; i0 < e_end; i0 += 2 * HB) {
                AGC_ISSUE(vrB, trB, slB, i0 + HB)
                AGC_PROCESS(vrA, trA, slA, i0)
                AGC_ISSUE(vrA, trA, slA, i0 + 2 * HB)
                AGC_PROCESS(vrB, trB, slB, i0 + HB)
            }
#undef AGC_ISSUE
#undef AGC_PROCESS
            flush();
            t0 += 16;
            more = (t0 < nd) || (base + 64 < end);
        }
        if (lane == 0) {
            s_rel[wib] = active ? rel : -1;
            s_nrows[wib] = nrows;
        }
        coop_barrier();
        int rk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rk[k] = __builtin_amdgcn_readfirstlane(s_rel[k]);
        if ((rk[0] & rk[1] & rk[2] & rk[3]) < 0) break;

        // ---- this wavefront's column tiles of z^T = message fragments x u^T, for the tiles of all four wavefronts
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (rk[k] < 0) continue;
            if (rk[k] != have) {
                have = rk[k];
                const unsigned short* mf = msgF + ((((int64_t)have * NY + hg) * NCT + wib * CTW) * NKS) * 2 * 512 + lane * 8;
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    fh[j] = *reinterpret_cast<const bf16x8*>(mf + (int64_t)(j * 2) * 512);
                    fm[j] = *reinterpret_cast<const bf16x8*>(mf + (int64_t)(j * 2 + 1) * 512);
                }
            }
            f32x4 acc[CTW];
#pragma unroll
            for (int c = 0; c < CTW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                acc[cc] = mfma16_t<F16>(fm[j], uh, acc[cc]);
                acc[cc] = mfma16_t<F16>(fh[j], um, acc[cc]);
                acc[cc] = mfma16_t<F16>(fh[j], uh, acc[cc]);
            }
            if (fi < s_nrows[k]) {
                float sc = 1.0f;
                if constexpr (F16) sc = s_rinv[k][fi][wib] * minv;      // (this wavefront's column tiles are quarter wib of the row)
                float* zr = zrows + (int64_t)s_pos[k][fi] * ld + co + 16 * (wib * CTW) + 4 * fg;
#pragma unroll
                for (int c = 0; c < CTW; ++c)
                    store_wt16(zr + 16 * c, acc[c][0] * sc, acc[c][1] * sc, acc[c][2] * sc, acc[c][3] * sc);      // (read by the merge kernel only)
            }
        }
        coop_barrier();               // tiles, positions and scales are rewritten by the next round
    }
}

// One wavefront per target: the runs of all its segments, in (relation, position) order.  Lane l holds VECF consecutive columns
// of the full row (d = 64 * VECF); DKP >= VECF, so a lane's columns belong to one head.  NB = run rows requested together.
// Returns the finished row of target i (normalised, gelu applied when asked for) in o[].
template <int VECF, int NB>
__device__ __forceinline__ void merge_target(int64_t i, int lane, const int32_t* __restrict__ segptr, const float* __restrict__ zrows,
                                             const float* __restrict__ zstat, const unsigned char* __restrict__ zflag, int R, int HT, int DKP,
                                             int apply_gelu, float (&o)[VECF]) {
    const int64_t ld = (int64_t)HT * DKP;
    const int hd = (lane * VECF) / DKP;
    const int64_t tile = i / HGT_TD;
    const int dl = (int)(i % HGT_TD);
    int e0 = 0, len = 0;
    if (lane <= R) {
        const int64_t b = (tile * (R + 1) + lane) * HGT_TD + dl;
        e0 = segptr[b];
        len = segptr[b + 1] - e0;
    }
    const int n_unclaimed = __builtin_amdgcn_readlane(len, R);      // R < 64 (checked by the launcher)
    if (lane >= R) len = 0;
    int incl = len;
#pragma unroll
    for (int o_ = 1; o_ < 64; o_ <<= 1) {
        const int t = __shfl_up(incl, o_);
        if (lane >= o_) incl += t;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    float M = HGT_NEG, L = 0.0f, X[VECF];
#pragma unroll
    for (int k = 0; k < VECF; ++k) X[k] = 0.0f;

    for (int j0 = 0; j0 < total; j0 += 64) {
        // edge j0 + lane of the target's concatenated segments -> (relation, position)
        const int j = j0 + lane;
        int rsel = 0;
        for (int r = 0; r < R; ++r) rsel += (j >= __builtin_amdgcn_readlane(incl, r)) ? 1 : 0;
        rsel = min(rsel, R - 1);
        const int pos = __shfl(e0, rsel) + (j - (__shfl(incl, rsel) - __shfl(len, rsel)));
        const bool live = j < total;
        const bool start = live && zflag[live ? pos : 0] != 0;
        unsigned long long runs = __builtin_amdgcn_ballot_w64(start);
        while (runs != 0ull) {
            // up to NB runs per round: their rows and statistics are requested together
            int pz[NB];
            int nr = 0;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (runs != 0ull) {
                    const int k = __builtin_ctzll(runs);
                    runs &= runs - 1ull;
                    pz[u] = __builtin_amdgcn_readlane(pos, k);
                    nr = u + 1;
                } else {
                    pz[u] = pz[u > 0 ? u - 1 : 0];
                }
            }
            float rows[NB][VECF];
            float2 st[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                load_vec<VECF>(zrows + (int64_t)pz[u] * ld + lane * VECF, rows[u]);
                st[u] = *reinterpret_cast<const float2*>(zstat + ((int64_t)pz[u] * HT + hd) * 2);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (u < nr) {
                    const float m_new = fmaxf(M, st[u].x);
                    const float sa = __expf(M - m_new), sb = __expf(st[u].x - m_new);
#pragma unroll
                    for (int k = 0; k < VECF; ++k) X[k] = fmaf(rows[u][k], sb, X[k] * sa);
                    L = fmaf(L, sa, st[u].y * sb);
                    M = m_new;
                }
            }
        }
    }
    if (n_unclaimed > 0) {                       // logit 0, no message (conv.py:68)
        const float m_new = fmaxf(M, 0.0f);
        const float sa = __expf(M - m_new), sb = __expf(0.0f - m_new);
#pragma unroll
        for (int k = 0; k < VECF; ++k) X[k] *= sa;
        L = fmaf(L, sa, (float)n_unclaimed * sb);
    }
    const float inv = apply_gelu == 2 ? 1.0f : 1.0f / (L + 1e-16f);      // (2: the plain weighted sum of hgt_edge_spmm_items)
#pragma unroll
    for (int k = 0; k < VECF; ++k) {
        float v = X[k] * inv;
        if (apply_gelu == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        o[k] = v;
    }
}

template <int VECF>
__global__ __launch_bounds__(256) void k_merge_runs(const int32_t* __restrict__ segptr, const float* __restrict__ zrows,
                                                    const float* __restrict__ zstat, const unsigned char* __restrict__ zflag,
                                                    float* __restrict__ agg, int R, int64_t NQ, int HT, int DKP, int apply_gelu,
                                                    int64_t ld_out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= NQ) return;
    float o[VECF];
    merge_target<VECF, 4>(i, lane, segptr, zrows, zstat, zflag, R, HT, DKP, apply_gelu, o);
    float* g = agg + i * ld_out + lane * VECF;
    if constexpr (VECF == 1) {
        g[0] = o[0];
    } else if constexpr (VECF == 2) {
        *reinterpret_cast<float2*>(g) = make_float2(o[0], o[1]);
    } else {
#pragma unroll
        for (int k = 0; k < VECF / 4; ++k) *reinterpret_cast<float4*>(g + 4 * k) = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Merge + node update in ONE kernel (round 6; sampled batches).  k_merge_runs + the update GEMM were two of the layer's five dependent
// kernels (6.7 + 12 us at c3, 14 + 24 us at c5), with the merged rows written to `agg` and read back in between.  Here a workgroup of
// SIXTEEN wavefronts owns 16 targets of one node type: every wavefront merges ONE target exactly like k_merge_runs (same order, same
// arithmetic: the same row bit for bit), splits the row into bf16 (fp16) hi / mid terms and parks it in an LDS slab; then the sixteen
// wavefronts are the 16 x 32 output columns of a_linear: the slab's rows 0..15 are the upper half of a 32-row MFMA tile (the lower half
// is never stored -- matrix-core time is not what a 3 000-target batch waits for), W_a's fragments stream through an eight-stage
// register ring that is already requested while the merge runs, and the gated skip + LayerNorm epilogue (conv.py:129-133) is the tile
// kernel's.  Same products in the same order as hgt_linear_update_*: identical output.
// ---------------------------------------------------------------------------------------------------------------------------------
struct MergeUpdateArgs {
    const int32_t* segptr; const float* zrows; const float* zstat; const unsigned char* zflag; int R; int HT; int DKP; int apply_gelu;
    const int32_t* rows; const int32_t* group_off; int n_groups; int n_out;
    const unsigned short* wsplit; const float* bias; const float* xs; int64_t ldxs; const float* skip; const float* lnw; const float* lnb;
    int use_norm; float* out;
};

constexpr int MU_ROWS = 16, MU_NW = 16, MU_STG = 8;

template <int VECF, bool F16, int TPW>
__global__ __launch_bounds__(64 * MU_NW, 4) void k_merge_update(const MergeUpdateArgs a) {
    constexpr int MU_R = MU_ROWS * TPW;      // targets per workgroup: wavefront w merges rows w (and w + 16)
    constexpr int DP = 64 * VECF, ASTRK = DP * 2 + 16, NKC = (DP / 16 + 3) & ~3, NB = VECF >= 8 ? 2 : 4;
    constexpr int STG = NKC < MU_STG ? NKC : MU_STG;
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][32 * ASTRK];      // [plane][row][k] (rows 16..31: never written, never stored)
    __shared__ int s_rid[MU_R];
    __shared__ float s_rinv[F16 ? MU_R : 1];
    __shared__ __attribute__((aligned(16))) float s_red[2][MU_R * MU_NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = 0, row0 = 0, nrows = 0;
    {
        // (an XCD-aware order -- row tiles in eight contiguous chunks, one per XCD -- measured neutral to slower here, r6: c5 30.9 vs 28.3 us)
        const int t = (int)blockIdx.x;
        int before = 0;
        bool found = false;
        for (g = 0; g < a.n_groups; ++g) {
            const int gb = a.group_off[g], ge = a.group_off[g + 1];
            const int nt = (ge - gb + MU_R - 1) / MU_R;
            if (t < before + nt) {
                row0 = gb + (t - before) * MU_R;
                nrows = min(MU_R, ge - row0);
                found = true;
                break;
            }
            before += nt;
        }
        if (!found) return;
    }
    const int n_out = a.n_out;
    const int n_pass = (n_out + BNP - 1) / BNP;
    const bool live = (wave >> 3) < n_pass;
    const unsigned short* wp = a.wsplit + ((int64_t)g * n_pass + (live ? (wave >> 3) : 0)) * NKC * 2 * W_PLANE_ELEMS + ((wave & 7) * 64 + lane) * 8;
    float winv = 1.0f;
    if constexpr (F16) winv = reinterpret_cast<const float*>(a.wsplit + (int64_t)a.n_groups * n_pass * NKC * 2 * W_PLANE_ELEMS)[g];
    // the first STG k-chunks of this wavefront's column block: in flight while the merge runs
    bf16x8 wh[STG], wm[STG];
    if (live) {      // (rows of at most 256 columns: wavefronts 8..15 only merge)
#pragma unroll
        for (int s_ = 0; s_ < STG; ++s_) {
            wh[s_] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)s_ * 2 * W_PLANE_ELEMS);
            wm[s_] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)s_ * 2 * W_PLANE_ELEMS + W_PLANE_ELEMS);
        }
    }
    // the skip rows of this lane's outputs (rows rt0 + 8 q, 4 columns): requested here -- their ids straight from the row list --
    // so that they travel while the merge runs (behind the slab barrier they cost the epilogue one more exposed round trip)
    constexpr int NQR = 2 * TPW;        // a lane's rows of the 32-row MFMA tile: rt0 + 8 q, q < NQR
    const int col_l = ((lane & 31) >> 2) * 4, rt0 = (lane & 3) + 4 * (lane >> 5);
    const int col = wave * 32 + col_l;
    const bool col_ok = live && col < n_out;
    float4 xv[NQR];
#pragma unroll
    for (int q = 0; q < NQR; ++q) {
        const int rt = rt0 + 8 * q;
        xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && rt < nrows) xv[q] = *reinterpret_cast<const float4*>(a.xs + (int64_t)a.rows[row0 + rt] * a.ldxs + col);
    }
    // ---- phase 1: wavefront w merges target rows[row0 + w] (and rows[row0 + 16 + w])
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
        const int r = tp * MU_ROWS + wave;
        const int rid = (r < nrows) ? a.rows[row0 + r] : -1;
        if (lane == 0) s_rid[r] = rid;
        float o[VECF];
        if (rid >= 0) {
            merge_target<VECF, NB>((int64_t)rid, lane, a.segptr, a.zrows, a.zstat, a.zflag, a.R, a.HT, a.DKP, a.apply_gelu, o);
        } else {
#pragma unroll
            for (int k = 0; k < VECF; ++k) o[k] = 0.0f;
        }
        float scale = 1.0f;
        if constexpr (F16) {
            float inv;
            f16_row_scale(wave_max_bits(abs_bits_v<VECF>(o)), scale, inv);
            if (lane == 0) s_rinv[r] = inv;
        }
        unsigned char* w_ = sA[0] + r * ASTRK + lane * VECF * 2;
        if constexpr (VECF == 1) {
            unsigned short hi, mid;
            split1_t<F16>(o[0], scale, hi, mid);
            *reinterpret_cast<unsigned short*>(w_) = hi;
            *reinterpret_cast<unsigned short*>(w_ + 32 * ASTRK) = mid;
        } else if constexpr (VECF == 2) {
            unsigned hi, mid;
            split2_t<F16>(o[0], o[1], scale, hi, mid);
            *reinterpret_cast<unsigned*>(w_) = hi;
            *reinterpret_cast<unsigned*>(w_ + 32 * ASTRK) = mid;
        } else {
#pragma unroll
            for (int q = 0; q < VECF / 4; ++q) {
                uint2 hi, mid;
                split4_t<F16>(make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]), scale, hi, mid);
                *reinterpret_cast<uint2*>(w_ + 8 * q) = hi;
                *reinterpret_cast<uint2*>(w_ + 8 * q + 32 * ASTRK) = mid;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: out[rows x n_out] = slab x W_a^T
    float y[NQR][4];
    if (live) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        const unsigned char* sl_ = sA[0] + (lane & 31) * ASTRK + (lane >> 5) * 16;
#pragma unroll
        for (int kc0 = 0; kc0 < NKC; kc0 += STG) {
#pragma unroll
            for (int s_ = 0; s_ < STG; ++s_) {
                const int kc = kc0 + s_;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sl_ + kc * 32);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(sl_ + kc * 32 + 32 * ASTRK);
                acc = mfma32_t<F16>(am, wh[s_], acc);
                acc = mfma32_t<F16>(ah, wm[s_], acc);
                acc = mfma32_t<F16>(ah, wh[s_], acc);
                if (kc + STG < NKC) {
                    wh[s_] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)(kc + STG) * 2 * W_PLANE_ELEMS);
                    wm[s_] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)(kc + STG) * 2 * W_PLANE_ELEMS + W_PLANE_ELEMS);
                }
            }
        }
        // ---- epilogue (hgt_gemm_tile.hip's)
        const bool o1 = lane & 1, o2 = lane & 2;
        const float alpha = 1.0f / (1.0f + expf(-a.skip[g]));
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + (int64_t)g * n_out + col);
#pragma unroll
        for (int q = 0; q < NQR; ++q) {
            float v0 = acc[4 * q], v1 = acc[4 * q + 1], v2 = acc[4 * q + 2], v3 = acc[4 * q + 3];
            quad_transpose(v0, v1, v2, v3, o1, o2);
            const float sc = F16 ? s_rinv[F16 ? (rt0 + 8 * q) : 0] * winv : 1.0f;
            y[q][0] = col_ok ? (v0 * sc + b4.x) * alpha + xv[q].x * (1.0f - alpha) : 0.0f;
            y[q][1] = col_ok ? (v1 * sc + b4.y) * alpha + xv[q].y * (1.0f - alpha) : 0.0f;
            y[q][2] = col_ok ? (v2 * sc + b4.z) * alpha + xv[q].z * (1.0f - alpha) : 0.0f;
            y[q][3] = col_ok ? (v3 * sc + b4.w) * alpha + xv[q].w * (1.0f - alpha) : 0.0f;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NQR; ++q) y[q][0] = y[q][1] = y[q][2] = y[q][3] = 0.0f;
    }
    const float inv_n = 1.0f / (float)n_out;
    float rstd[NQR];
#pragma unroll
    for (int q = 0; q < NQR; ++q) rstd[q] = 1.0f;
    if (a.use_norm) {
#pragma unroll
        for (int q = 0; q < NQR; ++q) {
            const float ps = strided8_sum(y[q][0] + y[q][1] + y[q][2] + y[q][3]);
            if (((lane & 31) >> 2) == 0) s_red[0][(rt0 + 8 * q) * MU_NW + wave] = ps;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQR; ++q) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < MU_NW; ++w) s += s_red[0][(rt0 + 8 * q) * MU_NW + w];
            const float mean = s * inv_n;
            float ps = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[q][i] -= mean;
                const float d = col_ok ? y[q][i] : 0.0f;
                ps = fmaf(d, d, ps);
            }
            ps = strided8_sum(ps);
            if (((lane & 31) >> 2) == 0) s_red[1][(rt0 + 8 * q) * MU_NW + wave] = ps;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQR; ++q) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < MU_NW; ++w) s += s_red[1][(rt0 + 8 * q) * MU_NW + w];
            rstd[q] = rsqrtf(s * inv_n + 1e-5f);
        }
    }
    if (col_ok) {
        float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.use_norm) {
            w4 = *reinterpret_cast<const float4*>(a.lnw + (int64_t)g * n_out + col);
            c4 = *reinterpret_cast<const float4*>(a.lnb + (int64_t)g * n_out + col);
        }
#pragma unroll
        for (int q = 0; q < NQR; ++q) {
            const int rt = rt0 + 8 * q;
            if (rt < nrows)      // (the layer's output: read by the next layer's projections -- another kernel)
                store_wt16(a.out + (int64_t)s_rid[rt] * n_out + col, y[q][0] * rstd[q] * w4.x + c4.x, y[q][1] * rstd[q] * w4.y + c4.y,
                           y[q][2] * rstd[q] * w4.z + c4.z, y[q][3] * rstd[q] * w4.w + c4.w);
        }
    }
}

// k_merge_update with one target per wavefront and the update GEMM on 16 x 16 x 32 MFMAs: the 16 merged rows are the B operand (16
// targets), W_a's fragments the A operand -- no wasted lower half of a 32-row tile (half the matrix-core time and half the LDS fragment
// reads of phase 2: 5.2 of c5's 29 us, 4.7 of the 4-layer model's 24, were MFMAs + fragment reads), a lane ends up with 4 consecutive
// output columns of ONE target (no quad transpose, one row scale per lane).  The SAME weight image: lane (m = l & 15, kg = l >> 4) of
// column tile t and 32-k step ks reads the 16 bytes the 32 x 32 form's lane (16 t + m, kg & 1) reads for k-chunk 2 ks + (kg >> 1).
// Products in the order  mid x hi, hi x mid, hi x hi  per 32-k step: the sums differ from the 32 x 32 form's in the last bits only.
template <int VECF, bool F16>
__global__ __launch_bounds__(64 * MU_NW, 4) void k_merge_update16(const MergeUpdateArgs a) {
    constexpr int DP = 64 * VECF, ASTRK = DP * 2 + 16, NKC = (DP / 16 + 3) & ~3, NKS = NKC / 2, NB = VECF >= 8 ? 2 : 4;
    constexpr int STG = NKS < 4 ? NKS : 4;                 // ring: 32-k steps whose fragments (2 column tiles x 2 planes) are in registers
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][MU_ROWS * ASTRK];      // [plane][row][k]
    __shared__ int s_rid[MU_ROWS];
    __shared__ float s_rinv[F16 ? MU_ROWS : 1];
    __shared__ __attribute__((aligned(16))) float s_red[2][MU_ROWS * MU_NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = 0, row0 = 0, nrows = 0;
    {
        const int t = (int)blockIdx.x;
        int before = 0;
        bool found = false;
        for (g = 0; g < a.n_groups; ++g) {
            const int gb = a.group_off[g], ge = a.group_off[g + 1];
            const int nt = (ge - gb + MU_ROWS - 1) / MU_ROWS;
            if (t < before + nt) {
                row0 = gb + (t - before) * MU_ROWS;
                nrows = min(MU_ROWS, ge - row0);
                found = true;
                break;
            }
            before += nt;
        }
        if (!found) return;
    }
    const int n_out = a.n_out;
    const int n_pass = (n_out + BNP - 1) / BNP;
    const bool live = (wave >> 3) < n_pass;
    const int m16 = lane & 15, kg = lane >> 4;
    const unsigned short* wp = a.wsplit + (((int64_t)g * n_pass + (live ? (wave >> 3) : 0)) * NKC + (kg >> 1)) * 2 * W_PLANE_ELEMS +
                               ((wave & 7) * 64 + (kg & 1) * 32 + m16) * 8;
    float winv = 1.0f;
    if constexpr (F16) winv = reinterpret_cast<const float*>(a.wsplit + (int64_t)a.n_groups * n_pass * NKC * 2 * W_PLANE_ELEMS)[g];
    // fragments of 32-k step ks: column tile t at + 128 elements, hi / mid planes; requested while the merge runs
    bf16x8 wh[STG][2], wm[STG][2];
#define MU16_LOAD_W(S, KS)                                                                                        \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                                                            \
        const unsigned short* q_ = wp + (int64_t)(KS) * 4 * W_PLANE_ELEMS + t_ * 128;                             \
        wh[S][t_] = *reinterpret_cast<const bf16x8*>(q_);                                                         \
        wm[S][t_] = *reinterpret_cast<const bf16x8*>(q_ + W_PLANE_ELEMS);                                         \
    }
    if (live) {
#pragma unroll
        for (int s_ = 0; s_ < STG; ++s_) { MU16_LOAD_W(s_, s_) }
    }
    // this lane's outputs: target m16, columns col0 + 16 t + 0..3; their skip rows travel while the merge runs
    const int col0 = wave * 32 + 4 * kg;
    const bool t_ok = m16 < nrows;
    bool c_ok[2];
    float4 xv[2];
#pragma unroll
    for (int t_ = 0; t_ < 2; ++t_) {
        c_ok[t_] = live && (col0 + 16 * t_) < n_out;
        xv[t_] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_ok[t_] && t_ok) xv[t_] = *reinterpret_cast<const float4*>(a.xs + (int64_t)a.rows[row0 + m16] * a.ldxs + col0 + 16 * t_);
    }
    // ---- phase 1: wavefront w merges target rows[row0 + w]  (k_merge_update's, TPW = 1)
    {
        const int r = wave;
        const int rid = (r < nrows) ? a.rows[row0 + r] : -1;
        if (lane == 0) s_rid[r] = rid;
        float o[VECF];
        if (rid >= 0) {
            merge_target<VECF, NB>((int64_t)rid, lane, a.segptr, a.zrows, a.zstat, a.zflag, a.R, a.HT, a.DKP, a.apply_gelu, o);
        } else {
#pragma unroll
            for (int k = 0; k < VECF; ++k) o[k] = 0.0f;
        }
        float scale = 1.0f;
        if constexpr (F16) {
            float inv;
            f16_row_scale(wave_max_bits(abs_bits_v<VECF>(o)), scale, inv);
            if (lane == 0) s_rinv[r] = inv;
        }
        unsigned char* w_ = sA[0] + r * ASTRK + lane * VECF * 2;
        if constexpr (VECF == 1) {
            unsigned short hi, mid;
            split1_t<F16>(o[0], scale, hi, mid);
            *reinterpret_cast<unsigned short*>(w_) = hi;
            *reinterpret_cast<unsigned short*>(w_ + MU_ROWS * ASTRK) = mid;
        } else if constexpr (VECF == 2) {
            unsigned hi, mid;
            split2_t<F16>(o[0], o[1], scale, hi, mid);
            *reinterpret_cast<unsigned*>(w_) = hi;
            *reinterpret_cast<unsigned*>(w_ + MU_ROWS * ASTRK) = mid;
        } else {
#pragma unroll
            for (int q = 0; q < VECF / 4; ++q) {
                uint2 hi, mid;
                split4_t<F16>(make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]), scale, hi, mid);
                *reinterpret_cast<uint2*>(w_ + 8 * q) = hi;
                *reinterpret_cast<uint2*>(w_ + 8 * q + MU_ROWS * ASTRK) = mid;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: out^T[32 columns x 16 targets] = W_a fragments x slab^T
    float y[2][4];
    if (live) {
        f32x4 acc[2];
        acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* sl_ = sA[0] + m16 * ASTRK + kg * 16;
#pragma unroll
        for (int ks0 = 0; ks0 < NKS; ks0 += STG) {
#pragma unroll
            for (int s_ = 0; s_ < STG; ++s_) {
                const int ks = ks0 + s_;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sl_ + ks * 64);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(sl_ + ks * 64 + MU_ROWS * ASTRK);
#pragma unroll
                for (int t_ = 0; t_ < 2; ++t_) {
                    acc[t_] = mfma16_t<F16>(wm[s_][t_], bh, acc[t_]);
                    acc[t_] = mfma16_t<F16>(wh[s_][t_], bm, acc[t_]);
                    acc[t_] = mfma16_t<F16>(wh[s_][t_], bh, acc[t_]);
                }
                if (ks + STG < NKS) { MU16_LOAD_W(s_, ks + STG) }
            }
        }
        const float alpha = 1.0f / (1.0f + expf(-a.skip[g]));
        const float sc = F16 ? s_rinv[F16 ? m16 : 0] * winv : 1.0f;
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c_ok[t_] && a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + (int64_t)g * n_out + col0 + 16 * t_);
            y[t_][0] = c_ok[t_] ? (acc[t_][0] * sc + b4.x) * alpha + xv[t_].x * (1.0f - alpha) : 0.0f;
            y[t_][1] = c_ok[t_] ? (acc[t_][1] * sc + b4.y) * alpha + xv[t_].y * (1.0f - alpha) : 0.0f;
            y[t_][2] = c_ok[t_] ? (acc[t_][2] * sc + b4.z) * alpha + xv[t_].z * (1.0f - alpha) : 0.0f;
            y[t_][3] = c_ok[t_] ? (acc[t_][3] * sc + b4.w) * alpha + xv[t_].w * (1.0f - alpha) : 0.0f;
        }
    } else {
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) y[t_][0] = y[t_][1] = y[t_][2] = y[t_][3] = 0.0f;
    }
#undef MU16_LOAD_W
    // ---- LayerNorm over the row of target m16: this lane's 8 columns, the 4 lanes of the target, the 16 wavefronts
    const float inv_n = 1.0f / (float)n_out;
    float rstd = 1.0f;
    if (a.use_norm) {
        float ps = (y[0][0] + y[0][1] + y[0][2] + y[0][3]) + (y[1][0] + y[1][1] + y[1][2] + y[1][3]);
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        if (kg == 0) s_red[0][m16 * MU_NW + wave] = ps;
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < MU_NW; ++w) s += s_red[0][m16 * MU_NW + w];
        const float mean = s * inv_n;
        ps = 0.0f;
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[t_][i] -= mean;
                const float d = c_ok[t_] ? y[t_][i] : 0.0f;
                ps = fmaf(d, d, ps);
            }
        ps += __shfl_xor(ps, 16);
        ps += __shfl_xor(ps, 32);
        if (kg == 0) s_red[1][m16 * MU_NW + wave] = ps;
        __syncthreads();
        s = 0.0f;
#pragma unroll
        for (int w = 0; w < MU_NW; ++w) s += s_red[1][m16 * MU_NW + w];
        rstd = rsqrtf(s * inv_n + 1e-5f);
    }
    if (t_ok) {
        const int64_t orow = (int64_t)s_rid[m16] * n_out;
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) {
            if (!c_ok[t_]) continue;
            float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.use_norm) {
                w4 = *reinterpret_cast<const float4*>(a.lnw + (int64_t)g * n_out + col0 + 16 * t_);
                c4 = *reinterpret_cast<const float4*>(a.lnb + (int64_t)g * n_out + col0 + 16 * t_);
            }
            // (the layer's output: read by the next layer's projections -- another kernel)
            store_wt16(a.out + orow + col0 + 16 * t_, y[t_][0] * rstd * w4.x + c4.x, y[t_][1] * rstd * w4.y + c4.y,
                       y[t_][2] * rstd * w4.z + c4.z, y[t_][3] * rstd * w4.w + c4.w);
        }
    }
}

template <int VEC, int LPH>
static int launch_runs(int mode, const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const unsigned short* msgF,
                       float* zrows, float* zstat, unsigned char* zflag, int R, int HT, hipStream_t stream, int raw = 0) {
    const bool f16 = (mode & 1) != 0;
    const int item_edges = mode >> 8;
    const unsigned blocks = ((unsigned)((pv.L.max_items + 3) / 4) + 127u) & ~127u;
    dim3 grid(blocks, (unsigned)(HT / (64 / LPH)));
#define AGI_LAUNCH(KERNEL, RTE_, F16_)                                                                                     \
    KERNEL<VEC, LPH, RTE_, F16_><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, logits, V, rteV, msgF, \
                                                           zrows, zstat, zflag, R, HT, raw, (int)pv.L.max_items)
    using G = AG<VEC, LPH>;
    // d_k >= 32 and 16-edge items: the transform shared by the workgroup (a wavefront's quarter of the fragment image fits its
    // registers); larger items: a wavefront each (see launch_logits_mfma).  d_k = 32 (c3): 14.25 -> 13.9 us (f16x3), 13.6 -> 12.7 (bf16x3)
    if constexpr (G::DKP >= 32 && G::NCT % 4 == 0 && (G::NCT / 4) * G::NKS <= 8) {
        if (!(mode & 2) && ((mode & 4) || item_edges <= 16)) {
            if (rteV) { if (f16) AGI_LAUNCH(k_edge_runs_coop, true, true); else AGI_LAUNCH(k_edge_runs_coop, true, false); }
            else      { if (f16) AGI_LAUNCH(k_edge_runs_coop, false, true); else AGI_LAUNCH(k_edge_runs_coop, false, false); }
            return HGT_OK;
        }
    }
    if (rteV) { if (f16) AGI_LAUNCH(k_edge_runs_mfma, true, true); else AGI_LAUNCH(k_edge_runs_mfma, true, false); }
    else      { if (f16) AGI_LAUNCH(k_edge_runs_mfma, false, true); else AGI_LAUNCH(k_edge_runs_mfma, false, false); }
#undef AGI_LAUNCH
    return HGT_OK;
}

}  // namespace

// scratch: [E][d] transformed rows | [E][H][2] run statistics | [E] run-start flags
extern "C" int hgt_edge_aggregate_items_bytes(int64_t n_edges, int32_t n_heads, int32_t dk_pad, uint64_t* out) {
    if (!out || n_edges < 0 || n_heads <= 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    const uint64_t E = (uint64_t)n_edges, d = (uint64_t)n_heads * dk_pad;
    *out = hgt_align_up(E * d * 4, 256) + hgt_align_up(E * n_heads * 8, 256) + hgt_align_up(E, 256);
    return HGT_OK;
}

// lab/hgt_edge_single_pass.hip (logits inside the runs kernel: correct, not faster -- DESIGN.md section 10): part of LAB builds only
// (make LAB=1); the shipped library answers HGT_ERR_UNSUPPORTED and the caller takes the two-kernel form
#ifdef HGT_LAB_KERNELS
int hgt_launch_single_pass_runs(int vec, int lph, bool f16, const HgtPlanView& pv, const float* Q, const float* K, const float* V,
                                const float* rteK, const float* rteV, const unsigned short* attF, const unsigned short* msgF, float* zrows,
                                float* zstat, unsigned char* zflag, int R, int HT, hipStream_t stream);
#else
static int hgt_launch_single_pass_runs(int, int, bool, const HgtPlanView&, const float*, const float*, const float*, const float*, const float*,
                                       const unsigned short*, const unsigned short*, float*, float*, unsigned char*, int, int, hipStream_t) {
    return HGT_ERR_UNSUPPORTED;
}
#endif

// sp != nullptr: the single-pass form (logits computed inside the runs kernel from Q, K and the attention fragments)
struct SinglePassArgs {
    const float* Q;
    const float* K;
    const float* rte_k;
    const void* att_frag;
};

static int aggregate_items_impl(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                const float* logits, const float* V, const float* rte_v, const void* msg_frag, int32_t frag_f16,
                                float* agg, int64_t n_q_rows, int32_t apply_gelu, void* scratch, uint64_t scratch_bytes,
                                void* stream_, const SinglePassArgs* spa, const MergeUpdateArgs* mu = nullptr, int64_t ld_out = 0) {
    if (!plan || !V || !msg_frag || (!agg && !mu) || (E > 0 && ((!spa && !logits) || !scratch)) || H <= 0 || 64 % H != 0 || dk_pad <= 0)
        return HGT_ERR_INVALID_ARG;
    if (spa && (!spa->Q || !spa->K || !spa->att_frag || ((spa->rte_k == nullptr) != (rte_v == nullptr)))) return HGT_ERR_INVALID_ARG;
    if (apply_gelu < 0 || apply_gelu > 2) return HGT_ERR_INVALID_ARG;
    const int64_t NQ = (n_q_rows > 0 && n_q_rows <= N) ? n_q_rows : N;
    if (NQ == 0) return HGT_OK;
    const int lph = 64 / H;
    const int64_t d = (int64_t)H * dk_pad;
    if (dk_pad % lph != 0 || R >= 64 || d % 64 != 0 || d / 64 > 8) return HGT_ERR_UNSUPPORTED;
    uint64_t need = 0;
    hgt_edge_aggregate_items_bytes(E, H, dk_pad, &need);
    if (E > 0 && scratch_bytes < need) return HGT_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    float* zrows = (float*)scratch;
    float* zstat = (float*)((char*)scratch + hgt_align_up((uint64_t)E * d * 4, 256));
    unsigned char* zflag = (unsigned char*)zstat + hgt_align_up((uint64_t)E * H * 8, 256);
    // the wavefront's layout after the head-group split of the matrix-core kernels (<= 256 columns per wavefront)
    int sp = 1;
    const int vec_full = dk_pad / lph;
    while (vec_full / sp > 4 && lph * sp * 2 <= 64) sp *= 2;
    if (vec_full / sp > 4) return HGT_ERR_UNSUPPORTED;
    const int vec = vec_full / sp, lphs = lph * sp;
    if (E > 0 && spa) {
        int rc = hgt_launch_single_pass_runs(vec, lphs, (frag_f16 & 1) != 0, pv, spa->Q, spa->K, V, spa->rte_k, rte_v, (const unsigned short*)spa->att_frag,
                                             (const unsigned short*)msg_frag, zrows, zstat, zflag, (int)R, (int)H, stream);
        if (rc != HGT_OK) return rc;
    } else if (E > 0) {
        int rc = HGT_ERR_UNSUPPORTED;
#define AGI_CASE(V_, L_) \
        if (vec == V_ && lphs == L_) rc = launch_runs<V_, L_>((frag_f16 & 7) | (hgt_item_edges(E) << 8), pv, logits, V, rte_v, (const unsigned short*)msg_frag, zrows, zstat, zflag, (int)R, (int)H, stream, apply_gelu == 2 ? 1 : 0);
#ifdef HGT_DEV_LAYOUTS
        AGI_CASE(4, 8) AGI_CASE(4, 16) AGI_CASE(1, 16)
#else
        AGI_CASE(1, 4) AGI_CASE(2, 4) AGI_CASE(4, 4) AGI_CASE(1, 8) AGI_CASE(2, 8) AGI_CASE(4, 8) AGI_CASE(1, 16) AGI_CASE(2, 16) AGI_CASE(4, 16)
        AGI_CASE(1, 32) AGI_CASE(2, 32) AGI_CASE(4, 32) AGI_CASE(1, 64) AGI_CASE(2, 64) AGI_CASE(4, 64)
#endif
#undef AGI_CASE
        if (rc != HGT_OK) return rc;
    }
    const int vf = (int)(d / 64);
    if (mu) {      // merge + a_linear + gated skip + LayerNorm in one kernel (k_merge_update)
        MergeUpdateArgs m = *mu;
        m.segptr = pv.segptr; m.zrows = zrows; m.zstat = zstat; m.zflag = zflag; m.R = (int)R; m.HT = (int)H; m.DKP = (int)dk_pad;
        m.apply_gelu = (int)apply_gelu;
        // 16 targets per workgroup (one per wavefront) while that is at most ~one workgroup per CU; 32 (two per wavefront: half the
        // passes over W_a) beyond
        // (two per wavefront also for 512-column rows from 2 048 targets on -- half the 1 MB fragment passes -- measured SLOWER: c5 36 vs
        //  31 us, published 4-layer model 32 vs 25 us per layer: the second merge's chain costs more than the passes it saves)
        const int tpw = NQ <= 16 * 288 ? 1 : 2;
        const unsigned ugrid = (unsigned)((NQ + MU_ROWS * tpw - 1) / (MU_ROWS * tpw) + m.n_groups);      // device-side group sizes: the upper bound
#define AGI_MU(VF)                                                                                      \
        do {                                                                                            \
            if (tpw == 1) {                                                                             \
                if (frag_f16 & 1) k_merge_update16<VF, true><<<ugrid, 64 * MU_NW, 0, stream>>>(m);          \
                else k_merge_update16<VF, false><<<ugrid, 64 * MU_NW, 0, stream>>>(m);                  \
            } else {                                                                                    \
                if (frag_f16 & 1) k_merge_update<VF, true, 2><<<ugrid, 64 * MU_NW, 0, stream>>>(m);         \
                else k_merge_update<VF, false, 2><<<ugrid, 64 * MU_NW, 0, stream>>>(m);                 \
            }                                                                                           \
        } while (0)
        if (vf == 1) AGI_MU(1); else if (vf == 2) AGI_MU(2); else if (vf == 4) AGI_MU(4); else AGI_MU(8);
#undef AGI_MU
        HGT_CHECK_LAUNCH();
        return HGT_OK;
    }
    const unsigned mgrid = (unsigned)((NQ + 3) / 4);
#define AGI_MERGE(VF) \
    k_merge_runs<VF><<<mgrid, 256, 0, stream>>>(pv.segptr, zrows, zstat, zflag, agg, (int)R, NQ, (int)H, (int)dk_pad, (int)apply_gelu, ld_out > 0 ? ld_out : d)
    if (vf == 1) AGI_MERGE(1); else if (vf == 2) AGI_MERGE(2); else if (vf == 4) AGI_MERGE(4); else AGI_MERGE(8);
#undef AGI_MERGE
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_aggregate_items(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                        const float* logits, const float* V, const float* rte_v, const void* msg_frag, int32_t frag_f16,
                                        float* agg, int64_t n_q_rows, int32_t apply_gelu, void* scratch, uint64_t scratch_bytes,
                                        void* stream_) {
    return aggregate_items_impl(plan, N, E, T, R, H, dk_pad, logits, V, rte_v, msg_frag, frag_f16, agg, n_q_rows, apply_gelu, scratch,
                                scratch_bytes, stream_, nullptr);
}

// ABI 6: logits + item-parallel aggregation in ONE walk (hgt_edge_single_pass.hip) for sampled sub-graphs: no [E][H] logits array.
// att_frag / msg_frag = hgt_relation_frag_pack[_f16] of att_t / msg_p; rte_k and rte_v both or neither; same scratch and the same
// result (up to fp32 rounding of the logits' summation order) as hgt_edge_logits_mfma + hgt_edge_aggregate_items.
// HGT_ERR_UNSUPPORTED for layouts the kernel is not instantiated for.
extern "C" int hgt_edge_single_pass_items(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                          const float* Q, const float* K, const float* V, const float* rte_k, const float* rte_v,
                                          const void* att_frag, const void* msg_frag, int32_t frag_f16, float* agg, int64_t n_q_rows,
                                          int32_t apply_gelu, void* scratch, uint64_t scratch_bytes, void* stream_) {
    const SinglePassArgs spa = {Q, K, rte_k, att_frag};
    return aggregate_items_impl(plan, N, E, T, R, H, dk_pad, nullptr, V, rte_v, msg_frag, frag_f16, agg, n_q_rows, apply_gelu, scratch,
                                scratch_bytes, stream_, &spa);
}

// ABI 7: item-parallel aggregation whose merge pass IS the node update: hgt_edge_aggregate_items followed by hgt_linear_update_* (a_linear +
// gated skip + LayerNorm, conv.py:119-133) with the merged rows never written to memory (k_merge_update: 16 targets per workgroup).
// rows / group_off: the target rows grouped by node type (hgt_plan_row_lists: rows_q / off_q); w_a_split: hgt_split_weights[_f16] of W_a
// with k = n_heads * dk_pad; frag_f16 selects the fp16 images of BOTH msg_frag and w_a_split.  Same output as the two-call form.
// HGT_ERR_UNSUPPORTED: rows wider than 512 padded columns, n_out > 512, n_out % 4 != 0, ld_skip % 4 != 0.
extern "C" int hgt_edge_aggregate_items_update(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                               const float* logits, const float* V, const float* rte_v, const void* msg_frag, int32_t frag_f16,
                                               int64_t n_q_rows, void* scratch, uint64_t scratch_bytes, const int32_t* rows,
                                               const int32_t* group_off, int32_t n_groups, const void* w_a_split, const float* b_a,
                                               const float* x_skip, int64_t ld_skip, const float* skip, const float* ln_w, const float* ln_b,
                                               int32_t use_norm, int32_t n_out, float* out, void* stream_) {
    if (!rows || !group_off || n_groups <= 0 || !w_a_split || !x_skip || !skip || !out || n_out <= 0) return HGT_ERR_INVALID_ARG;
    if (use_norm && (!ln_w || !ln_b)) return HGT_ERR_INVALID_ARG;
    const int64_t d = (int64_t)H * dk_pad;
    if (d > 512 || d % 64 != 0 || n_out > 512 || (n_out & 3) != 0 || (ld_skip & 3) != 0 || ((uintptr_t)x_skip & 15) != 0 || n_out > d)
        return HGT_ERR_UNSUPPORTED;
    MergeUpdateArgs m;
    m.segptr = nullptr; m.zrows = nullptr; m.zstat = nullptr; m.zflag = nullptr; m.R = 0; m.HT = 0; m.DKP = 0; m.apply_gelu = 1;
    m.rows = rows; m.group_off = group_off; m.n_groups = n_groups; m.n_out = n_out;
    m.wsplit = (const unsigned short*)w_a_split; m.bias = b_a; m.xs = x_skip; m.ldxs = ld_skip; m.skip = skip; m.lnw = ln_w; m.lnb = ln_b;
    m.use_norm = use_norm; m.out = out;
    return aggregate_items_impl(plan, N, E, T, R, H, dk_pad, logits, V, rte_v, msg_frag, frag_f16, nullptr, n_q_rows, 1, scratch, scratch_bytes,
                                stream_, nullptr, &m);
}

// ABI 7: hgt_edge_spmm (GIVEN edge weights, plain weighted sum: the gather passes of the backward) on the item-parallel kernels -- the
// form for sampled batches, where the sub-tile kernel's wavefronts each walk sixteen targets' edges one after the other (285 us per
// call on the transposed plan of the c3 batch against ~25 us here, r6).  scratch: hgt_edge_aggregate_items_bytes().  f_frag =
// hgt_relation_frag_pack(f_p) (bf16 image).  Deterministic (no atomics).  HGT_ERR_UNSUPPORTED: more than 63 relations, rows wider than
// 512 padded columns, ld_out % 4 != 0 -> hgt_edge_spmm.
extern "C" int hgt_edge_spmm_items(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                   const float* weights, const float* rows, const float* rte_rows, const void* f_frag, float* out,
                                   int64_t ld_out, int64_t n_q_rows, void* scratch, uint64_t scratch_bytes, void* stream_) {
    if (!out || ld_out < (int64_t)H * dk_pad || (ld_out & 3) != 0 || (((uintptr_t)out) & 15) != 0) return HGT_ERR_UNSUPPORTED;
    return aggregate_items_impl(plan, N, E, T, R, H, dk_pad, weights, rows, rte_rows, f_frag, 0, out, n_q_rows, 2, scratch, scratch_bytes, stream_,
                                nullptr, nullptr, ld_out);
}
