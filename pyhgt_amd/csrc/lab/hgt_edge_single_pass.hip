// Single-pass edge phase for the latency regime (sampled sub-graphs): attention logits, per-run online softmax and the
// attention-weighted, relation-transformed message sums of conv.py:98-111 in ONE walk over the edges -- no [E][H] logits array,
// one kernel launch less, K and V rows of an edge requested together.
//
// It is the item-parallel aggregation (hgt_edge_agg_items.hip: one wavefront per work item = <= 256 target-sorted edges of one
// (target tile, relation); runs of consecutive same-target edges; 16 runs per matrix-core round; k_merge_runs combines a target's
// runs in a fixed order) with the logits of hgt_edge_logits_mfma.hip folded in.  Per group of <= 16 runs of a chunk:
//   A. the Q rows of the runs' targets -> split hi / mid -> wave-private LDS tile;
//   B. q~^T = A'^T-fragments x Q^T on v_mfma_f32_16x16x32 (fragments of hgt_relation_frag_pack(att_t)), written back over the same
//      LDS bytes as an fp32 [16][DP + 4] tile -- the target-side transform of SURVEY appendix A.4, once per (target, relation) run;
//   C. the edges of the group: gather K AND V (+ temporal rows), s = <q~[run], k> reduced over the head's lanes, online softmax with
//      the run's own reference, u += e^(s - m) v; a finished run is split and parked as a row of a second LDS tile;
//   D. z^T = message fragments x u^T, rows + (m, l) per head stored at the position of the run's first edge (k_merge_runs' input).
// MEASURED (round 4, MI355X): not faster -- ogbn-mag batch 66.7 vs 66.9 us per layer, OAG 2-layer forward 372 vs 354 us.  The q~
// tile has to stay alive while the runs' u rows are parked, so a wavefront needs two 16 KB LDS tiles: 4 wavefronts per CU instead
// of the 8 each of the two separate kernels gets, and a work item stays one chain of dependent round trips (Q rows, fragments,
// K / V rows, fragments, store) that only co-resident wavefronts hide.  Hence opt-in (HGT_FLAG_SINGLE_PASS; it does save the
// [E][H] logits array), tested on every layout it is instantiated for, and the answer to "would one walk over the edges be
// faster?" for the latency regime; for the million-node graph the same question is answered on paper in DESIGN.md section 10.
// No atomics, fixed order: bit-reproducible like its two-kernel form.
#include "../hgt_edge_common.h"
#include "../hgt_split_common.h"

#ifndef HGT_LOGITS_XCD
#define HGT_LOGITS_XCD 1
#endif
#ifndef HGT_SP_GS
#define HGT_SP_GS 8      // column-tile steps whose fragments are requested together (16 loads in flight)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC, int LPH>
struct SG {   // geometry of one wavefront's slice (tile layout of MG in hgt_edge_agg_mfma.hip)
    static constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH;
    static constexpr int NCT = DP / 16, KW = DKP > 32 ? DKP : 32, NKS = KW / 32;
    static constexpr int ROWB = DP * 2, NS = DP / 8, PLANE = 16 * ROWB;
    static constexpr int QS = DP + 4;                   // floats per row of the fp32 q~ tile (+4: the 16 rows start on different banks)
    static constexpr int Q_LDS = 16 * QS * 4;           // bytes of the q~ tile (>= 2 * PLANE: it first holds the 16-bit planes of Q)
    static constexpr int WAVE_LDS = Q_LDS + 2 * PLANE;  // + the u tile
};

template <int VEC>
__device__ __forceinline__ unsigned sp_abs_bits(const float (&v)[VEC]) {
    float m = fabsf(v[0]);
#pragma unroll
    for (int i = 1; i < VEC; ++i) m = fmaxf(m, fabsf(v[i]));
    return __builtin_bit_cast(unsigned, m);
}

// one row of VEC floats -> hi / mid 16-bit planes of a swizzled [16][DP] tile (the layout both transforms read)
template <int VEC, bool F16, int ROWB, int NS, int PLANE>
__device__ __forceinline__ void park_row(unsigned char* tile, int r, int wb, const float (&v)[VEC], float scale) {
    unsigned char* w = tile + r * ROWB + ((((wb >> 4) ^ (r & (NS - 1)))) << 4) + (wb & 15);
    if constexpr (VEC == 1) {
        unsigned short hi, mid;
        split1_t<F16>(v[0], scale, hi, mid);
        *reinterpret_cast<unsigned short*>(w) = hi;
        *reinterpret_cast<unsigned short*>(w + PLANE) = mid;
    } else if constexpr (VEC == 2) {
        unsigned hi, mid;
        split2_t<F16>(v[0], v[1], scale, hi, mid);
        *reinterpret_cast<unsigned*>(w) = hi;
        *reinterpret_cast<unsigned*>(w + PLANE) = mid;
    } else {
        uint2 hi, mid;
        split4_t<F16>(make_float4(v[0], v[1], v[2], v[3]), scale, hi, mid);
        *reinterpret_cast<uint2*>(w) = hi;
        *reinterpret_cast<uint2*>(w + PLANE) = mid;
    }
}

template <int VEC, int LPH, bool RTE, bool F16>
__global__ __launch_bounds__(256, 1) void k_edge_single_pass(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ Q, const float* __restrict__ K,
    const float* __restrict__ V, const float* __restrict__ rteK, const float* __restrict__ rteV, const unsigned short* __restrict__ attF,
    const unsigned short* __restrict__ msgF, float* __restrict__ zrows, float* __restrict__ zstat, unsigned char* __restrict__ zflag,
    int R, int HT) {
    using G = SG<VEC, LPH>;
    constexpr int DKP = G::DKP, DP = G::DP, H = G::H, NCT = G::NCT, KW = G::KW, NKS = G::NKS, ROWB = G::ROWB, NS = G::NS, QS = G::QS;
    constexpr int UN = RTE ? (unroll_for<VEC>() * 3) / 4 : unroll_for<VEC>(), HB = UN / 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4][G::WAVE_LDS];
    __shared__ int s_pos[4][16];
    __shared__ float s_rinv[F16 ? 4 : 1][16];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_items = hdr->n_items;
#if HGT_LOGITS_XCD     // XCD-aware item order, see k_edge_logits
    constexpr int XC = 16;
    const int q8 = (int)(blockIdx.x >> 3), vblock = (q8 / XC) * (8 * XC) + (int)(blockIdx.x & 7u) * XC + (q8 % XC);
#else
    const int vblock = blockIdx.x;
#endif
    const int item = vblock * 4 + wib;
    if (item >= n_items) return;
    const HgtItem it = items[item];
    const int beg = __builtin_amdgcn_readfirstlane(it.beg), end = __builtin_amdgcn_readfirstlane(it.end);
    const int rel = __builtin_amdgcn_readfirstlane(it.rel);
    if (rel >= R) return;                      // edges no meta relation claims: logit 0, no message -- k_merge_runs counts them
    const int hg = blockIdx.y;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP, NY = HT / H;
    const int h = lane / LPH, p = lane % LPH;

    unsigned char* qbytes = smem[wib];                        // Q planes, then the fp32 q~ tile
    float* qtile = reinterpret_cast<float*>(qbytes);
    unsigned char* utile = smem[wib] + G::Q_LDS;
    const int fi = lane & 15, fg = lane >> 4;
    const int wb = lane * VEC * 2;
    const int rrow = fi * ROWB;
    const int64_t frag_rel = (((int64_t)rel * NY + hg) * NCT) * NKS * 2 * 512 + lane * 8;
    const unsigned short* __restrict__ af = attF + frag_rel;
    const unsigned short* __restrict__ mf = msgF + frag_rel;
    float ainv = 1.0f, minv = 1.0f;            // inverse scales of the fp16 fragment images (behind the fragments)
    if constexpr (F16) {
        ainv = reinterpret_cast<const float*>(attF + (int64_t)R * NY * NCT * NKS * 2 * 512)[0];
        minv = reinterpret_cast<const float*>(msgF + (int64_t)R * NY * NCT * NKS * 2 * 512)[0];
    }
    constexpr int STEPS = NCT * NKS, GSW = (STEPS >= 32 ? 2 : 1) * HGT_SP_GS, GS = GSW < STEPS ? GSW : STEPS, NG = STEPS / GS;
    static_assert(STEPS % GS == 0, "column-tile steps come in multiples of 4");

    // acc^T[16 c .. +16][row fi] = fragments x tile^T  (3-term split product; rows the group does not use hold stale bytes: every
    // column of the transposed product depends on its own row only)
    auto transform = [&](const unsigned short* frags, const unsigned char* tile, f32x4 (&acc)[NCT]) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 fh[GS], fm[GS];
        const unsigned short* fgp = frags;
        asm volatile("" : "+v"(fgp));          // (keeps hipcc from hoisting all 2 * STEPS fragment loads out of the loops)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const unsigned short* t_ = fgp + (int64_t)((g * GS + j) * 2) * 512;
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
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

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
        unsigned long long mrem = mask;       // leaders of the runs whose Q row is not loaded yet
        int lead_idx = 0;

        for (int t0 = 0; t0 < nd; t0 += 16) {
            // ---- A. Q rows of runs [t0, t0 + 16) (rows beyond the chunk's last run re-read its last leader: nobody reads them)
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
            const int e_lo = __builtin_ctzll(in_g), e_end = e_lo + __builtin_popcountll(in_g);
            const int nrows = min(16, nd - t0);
            if (lead && my_slot >= t0 && my_slot < t0 + 16) s_pos[wib][my_slot - t0] = base + lane;

            float krA[HB][VEC], vrA[HB][VEC], tkA[RTE ? HB : 1][VEC], tvA[RTE ? HB : 1][VEC];
            float krB[HB][VEC], vrB[HB][VEC], tkB[RTE ? HB : 1][VEC], tvB[RTE ? HB : 1][VEC];
#define SP_ISSUE(KR, VR, TK, TV, I0)                                                               \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        const int idx = min((I0) + u, e_end - 1);                                                  \
        const int s_ = __builtin_amdgcn_readlane(my_src, idx);                                     \
        load_vec<VEC>(K + (int64_t)s_ * ld + co + lane * VEC, KR[u]);                              \
        load_vec<VEC>(V + (int64_t)s_ * ld + co + lane * VEC, VR[u]);                              \
        if constexpr (RTE) {                                                                       \
            const int ri = __builtin_amdgcn_readlane(my_rte, idx);                                 \
            load_vec<VEC>(rteK + (int64_t)ri * ld + co + lane * VEC, TK[u]);                       \
            load_vec<VEC>(rteV + (int64_t)ri * ld + co + lane * VEC, TV[u]);                       \
        }                                                                                          \
    }
            // the first K / V rows of the group are requested before the transform (behind the Q rows in the load queue)
            SP_ISSUE(krA, vrA, tkA, tvA, e_lo)

            float qinv = 1.0f;                 // fp16 split: lane r = inverse scale of Q row r
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float scale = 1.0f;
                if constexpr (F16) {
                    float inv;
                    f16_row_scale(wave_max_bits(sp_abs_bits<VEC>(qrow[r])), scale, inv);
                    qinv = (lane == r) ? inv : qinv;
                }
                park_row<VEC, F16, ROWB, NS, G::PLANE>(qbytes, r, wb, qrow[r], scale);
            }
            wave_sync();

            // ---- B. q~^T = A'^T fragments x Q^T, then the tile's bytes become the fp32 q~ rows
            {
                f32x4 acc[NCT];
                transform(af, qbytes, acc);
                wave_sync();                   // every lane has read the 16-bit planes
                float sc = 1.0f;
                if constexpr (F16) sc = __shfl(qinv, fi) * ainv;
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    *reinterpret_cast<float4*>(qtile + fi * QS + 16 * c + 4 * fg) =
                        make_float4(acc[c][0] * sc, acc[c][1] * sc, acc[c][2] * sc, acc[c][3] * sc);
                wave_sync();
            }

            // ---- C. the edges of the group: logit, online softmax per run, u += e^(s - m) v; finished runs are parked in the u tile
            int cur_r = -1, cur_pos = 0;
            float U[VEC], m_run = HGT_NEG, l_run = 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) U[i] = 0.0f;
            auto flush = [&]() {
                if (cur_r < 0) return;
                float scale = 1.0f;
                if constexpr (F16) {
                    float inv;
                    f16_row_scale(wave_max_bits(sp_abs_bits<VEC>(U)), scale, inv);
                    if (lane == 0) s_rinv[wib][cur_r] = inv;
                }
                park_row<VEC, F16, ROWB, NS, G::PLANE>(utile, cur_r, wb, U, scale);
                if (p == 0) *reinterpret_cast<float2*>(zstat + ((int64_t)cur_pos * HT + hg * H + h) * 2) = make_float2(m_run, l_run);
            };
#define SP_PROCESS(KR, VR, TK, TV, I0)                                                             \
    _Pragma("unroll") for (int u = 0; u < HB; ++u) {                                               \
        if ((I0) + u < e_end) {                                                                    \
            const int r_ = __builtin_amdgcn_readlane(my_slot, (I0) + u) - t0;                      \
            float qt[VEC];                                                                         \
            load_vec<VEC>(qtile + r_ * QS + lane * VEC, qt);                                       \
            float s_e = 0.0f;                                                                      \
            _Pragma("unroll") for (int i = 0; i < VEC; ++i) {                                      \
                float kv = KR[u][i];                                                               \
                if constexpr (RTE) kv += TK[u][i];                                                 \
                s_e = fmaf(qt[i], kv, s_e);                                                        \
            }                                                                                      \
            s_e = head_allreduce<LPH>(s_e);                                                        \
            if (r_ != cur_r) {                                                                     \
                flush();                                                                           \
                cur_r = r_;                                                                        \
                cur_pos = base + (I0) + u;                                                         \
                m_run = HGT_NEG;                                                                   \
                l_run = 0.0f;                                                                      \
                _Pragma("unroll") for (int i = 0; i < VEC; ++i) U[i] = 0.0f;                       \
            }                                                                                      \
            const float m_new = fmaxf(m_run, s_e);                                                 \
            const float sc = __expf(m_run - m_new), pe = __expf(s_e - m_new);                      \
            _Pragma("unroll") for (int i = 0; i < VEC; ++i) {                                      \
                float vv = VR[u][i];                                                               \
                if constexpr (RTE) vv += TV[u][i];                                                 \
                U[i] = fmaf(pe, vv, U[i] * sc);                                                    \
            }                                                                                      \
            l_run = fmaf(l_run, sc, pe);                                                           \
            m_run = m_new;                                                                         \
        }                                                                                          \
    }
            for (int i0 = e_lo; i0 < e_end; i0 += 2 * HB) {
                SP_ISSUE(krB, vrB, tkB, tvB, i0 + HB)
                SP_PROCESS(krA, vrA, tkA, tvA, i0)
                SP_ISSUE(krA, vrA, tkA, tvA, i0 + 2 * HB)
                SP_PROCESS(krB, vrB, tkB, tvB, i0 + HB)
            }
#undef SP_ISSUE
#undef SP_PROCESS
            flush();
            wave_sync();

            // ---- D. z^T = message fragments x u^T; transformed rows -> scratch, at the position of the run's first edge
            {
                f32x4 acc[NCT];
                transform(mf, utile, acc);
                if (fi < nrows) {
                    float sc = 1.0f;
                    if constexpr (F16) sc = s_rinv[wib][fi] * minv;
                    float* zr = zrows + (int64_t)s_pos[wib][fi] * ld + co + 4 * fg;
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        *reinterpret_cast<float4*>(zr + 16 * c) = make_float4(acc[c][0] * sc, acc[c][1] * sc, acc[c][2] * sc, acc[c][3] * sc);
                }
            }
            wave_sync();                       // both tiles and the position table are rewritten by the next group
        }
    }
}

template <int VEC, int LPH>
static int launch_single_pass(bool f16, const HgtPlanView& pv, const float* Q, const float* K, const float* V, const float* rteK,
                              const float* rteV, const unsigned short* attF, const unsigned short* msgF, float* zrows, float* zstat,
                              unsigned char* zflag, int R, int HT, hipStream_t stream) {
    const unsigned blocks = ((unsigned)((pv.L.max_items + 3) / 4) + 127u) & ~127u;
    dim3 grid(blocks, (unsigned)(HT / (64 / LPH)));
#define SP_LAUNCH(RTE_, F16_)                                                                                                  \
    k_edge_single_pass<VEC, LPH, RTE_, F16_><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, Q, K, V, rteK, rteV, \
                                                                       attF, msgF, zrows, zstat, zflag, R, HT)
    if (rteV) { if (f16) SP_LAUNCH(true, true); else SP_LAUNCH(true, false); }
    else      { if (f16) SP_LAUNCH(false, true); else SP_LAUNCH(false, false); }
#undef SP_LAUNCH
    return HGT_OK;
}

}  // namespace

// Runs of the single-pass form for the layouts it is instantiated for (HGT_ERR_UNSUPPORTED otherwise: hgt_edge_logits +
// hgt_edge_aggregate_items); writes the scratch arrays k_merge_runs reads.  vec / lph: the wavefront's layout after the head-group
// split of the matrix-core kernels.
__attribute__((visibility("hidden"))) int hgt_launch_single_pass_runs(int vec, int lph, bool f16, const HgtPlanView& pv, const float* Q,
                                                                      const float* K, const float* V, const float* rteK, const float* rteV,
                                                                      const unsigned short* attF, const unsigned short* msgF, float* zrows,
                                                                      float* zstat, unsigned char* zflag, int R, int HT,
                                                                      hipStream_t stream) {
    if ((rteK == nullptr) != (rteV == nullptr)) return HGT_ERR_INVALID_ARG;
#define SP_CASE(V_, L_) \
    if (vec == V_ && lph == L_) return launch_single_pass<V_, L_>(f16, pv, Q, K, V, rteK, rteV, attF, msgF, zrows, zstat, zflag, R, HT, stream);
#ifdef HGT_DEV_LAYOUTS
    SP_CASE(4, 8) SP_CASE(4, 16) SP_CASE(1, 16)
#else
    // the reference's shapes: d = 256 / 8 heads (4, 8), n_hid 400 / 512 with 8 heads (4, 16), d = 64 / 4 heads (1, 16), d = 128 (2, 8), (2, 16)
    SP_CASE(4, 8) SP_CASE(4, 16) SP_CASE(1, 16) SP_CASE(2, 8) SP_CASE(2, 16) SP_CASE(4, 32)
#endif
#undef SP_CASE
    return HGT_ERR_UNSUPPORTED;
}
