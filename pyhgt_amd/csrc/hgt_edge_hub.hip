// Hub path of the aggregation (targets with more than HGT_HUB_DEG in-edges), shared by both aggregation kernels.
#include "hgt_edge_common.h"

namespace {

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
#ifndef HGT_HUB_CHUNKS
#define HGT_HUB_CHUNKS HGT_HUB_PIECES      // pieces per (hub, relation) range: 64 -> 8.8 ms, 32 -> 8.4 ms, 16 -> 8.7 ms at c2 with Zipf(0.8) targets (the longest pieces of
                               // the largest hub against the per-piece fixed cost)
#endif
constexpr int HUB_CHUNKS = HGT_HUB_CHUNKS;
constexpr int HUB_GRID_WAVES = 8192;

__device__ __forceinline__ int f2ord(float f) { const int b = __builtin_bit_cast(int, f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float ord2f(int o) { return __builtin_bit_cast(float, o ^ ((o >> 31) & 0x7fffffff)); }
// deterministic mode: the partial slots of pieces without an edge are never written, so they start at zero
__global__ void k_hub_init_parts(const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list, HgtHubBuffers hb, int HT,
                                 int dfull, int R) {
    const int64_t per_hub = (int64_t)(R + 1) * HUB_CHUNKS * (dfull + HT);
    const int64_t total = (int64_t)hdr->n_hubs * per_hub;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = i / per_hub, r = i % per_hub;
        if (hb.q_hi > 0) {
            const int64_t dst = hub_list[slot];
            if (dst < hb.q_lo || dst >= hb.q_hi) continue;
        }
        const int64_t n_acc = (int64_t)(R + 1) * HUB_CHUNKS * dfull;
        if (r < n_acc) hb.part[slot * n_acc + r] = 0.0f;
        else hb.lpart[slot * (int64_t)(R + 1) * HUB_CHUNKS * HT + (r - n_acc)] = 0.0f;
    }
}

__global__ void k_hub_init(const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list, HgtHubBuffers hb, int HT, int dfull) {
    const int n = hdr->n_hubs;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = 2 * HT + dfull;
    if (i >= (int64_t)n * per) return;
    const int slot = (int)(i / per), r = (int)(i % per);
    if (hb.q_hi > 0) {      // a target block: only ITS hubs (another block may be accumulating into the other slots right now)
        const int64_t dst = hub_list[slot];
        if (dst < hb.q_lo || dst >= hb.q_hi) return;
    }
    if (r < HT) hb.mx[slot * HT + r] = f2ord(-1.0e30f);
    else if (r < 2 * HT) hb.l[slot * HT + r - HT] = 0.0f;
    else hb.acc[(int64_t)slot * dfull + r - 2 * HT] = 0.0f;
}

// work id w -> (hub slot, relation bucket, piece); edges [pb, pe) of that piece
__device__ __forceinline__ bool hub_piece(int w, int n_hubs, int R, const int32_t* __restrict__ hub_list,
                                          const int32_t* __restrict__ segptr, int& slot, int& rel, int& pb, int& pe,
                                          int64_t q_lo, int64_t q_hi) {
    const int per_hub = (R + 1) * HUB_CHUNKS;
    slot = w / per_hub;
    if (slot >= n_hubs) return false;
    const int r2 = w - slot * per_hub;
    rel = r2 / HUB_CHUNKS;
    const int c = r2 - rel * HUB_CHUNKS;
    const int64_t dst = hub_list[slot];
    if (q_hi > 0 && (dst < q_lo || dst >= q_hi)) return false;      // a hub of another target block (multi-GPU path)
    const int64_t b = ((dst / HGT_TD) * (R + 1) + rel) * HGT_TD + dst % HGT_TD;
    const int beg = segptr[b], end = segptr[b + 1];
    const int len = end - beg, piece = (len + HUB_CHUNKS - 1) / HUB_CHUNKS;
    pb = beg + c * piece;
    pe = min(end, pb + piece);
    return pb < pe;
}

__global__ __launch_bounds__(256) void k_hub_max(const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list,
                                                 const int32_t* __restrict__ segptr, const float* __restrict__ logits, int R, int HT,
                                                 HgtHubBuffers hb) {
    const int n_hubs = hdr->n_hubs;
    if (n_hubs == 0) return;
    const int lane = threadIdx.x & 63;
    const int epw = 64 / HT;                       // edges per wave iteration
    const int hh = lane % HT, eo = lane / HT;
    const int total = n_hubs * (R + 1) * HUB_CHUNKS;
    for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < total; w += gridDim.x * 4) {
        int slot, rel, pb, pe;
        if (!hub_piece(w, n_hubs, R, hub_list, segptr, slot, rel, pb, pe, hb.q_lo, hb.q_hi)) continue;
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
    const float* __restrict__ V, const float* __restrict__ rteV, const float* __restrict__ msgP, int R, int HT, HgtHubBuffers hb, int raw) {
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
        if (!hub_piece(w, n_hubs, R, hub_list, segptr, slot, rel, pb, pe, hb.q_lo, hb.q_hi)) continue;
        slot = __builtin_amdgcn_readfirstlane(slot);
        rel = __builtin_amdgcn_readfirstlane(rel);
        pb = __builtin_amdgcn_readfirstlane(pb);
        pe = __builtin_amdgcn_readfirstlane(pe);
        // true max over ALL in-edges of the hub (k_hub_max folds a 0 in for a non-empty unclaimed bucket), like PyG's softmax
        const float mref = raw ? 0.0f : ord2f(hb.mx[slot * HT + hg * H + h]);   // raw: the array holds the weights (hgt_edge_spmm)
        float l_part = 0.0f;
        // deterministic mode: this piece's own slot (summed by k_hub_finalize in piece order) instead of fp32 atomics
        const int64_t pslot = w;      // (hub slot, relation bucket, piece) -> one slot per piece
        if (rel >= R) {                            // unclaimed: logit 0, no message
            l_part = (float)(pe - pb) * __expf(0.0f - mref);
            if (p == 0) {
                if (hb.part) hb.lpart[pslot * HT + hg * H + h] = l_part;
                else atomicAdd(&hb.l[slot * HT + hg * H + h], l_part);
            }
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
                        const float pe_ = raw ? sl[u] : __expf(sl[u] - mref);
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
        if (hb.part) {
            float* o = hb.part + pslot * ld + co + lane * VEC;
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = z[i];
            if (p == 0) hb.lpart[pslot * HT + hg * H + h] = l_part;
        } else {
            float* o = hb.acc + (int64_t)slot * ld + co + lane * VEC;
#pragma unroll
            for (int i = 0; i < VEC; ++i) unsafeAtomicAdd(o + i, z[i]);
            if (p == 0) unsafeAtomicAdd(&hb.l[slot * HT + hg * H + h], l_part);
        }
    }
}

__global__ __launch_bounds__(256) void k_hub_finalize(const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ hub_list,
                                                      HgtHubBuffers hb, float* __restrict__ agg, int HT, int dkp, int64_t NQ,
                                                      int apply_gelu, int64_t ld_out, int R) {
    const int n_hubs = hdr->n_hubs;
    const int lane = threadIdx.x & 63;
    const int dfull = HT * dkp;
    for (int slot = blockIdx.x * 4 + (threadIdx.x >> 6); slot < n_hubs; slot += gridDim.x * 4) {
        const int64_t row = hub_list[slot];
        if (row >= NQ || (hb.q_hi > 0 && (row < hb.q_lo || row >= hb.q_hi))) continue;
        for (int c = lane; c < dfull; c += 64) {
            float v, l;
            if (hb.part) {      // deterministic mode: the pieces' partial rows and exp-sums in (relation, piece) order
                const int np = (R + 1) * HUB_CHUNKS;
                v = 0.0f; l = 0.0f;
                for (int q = 0; q < np; ++q) {
                    v += hb.part[((int64_t)slot * np + q) * dfull + c];
                    l += hb.lpart[((int64_t)slot * np + q) * HT + c / dkp];
                }
            } else {
                v = hb.acc[(int64_t)slot * dfull + c];
                l = hb.l[slot * HT + c / dkp];
            }
            if (apply_gelu != 2) v /= (l + 1e-16f);
            if (apply_gelu == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            agg[row * ld_out + c] = v;
        }
    }
}

template <int VEC, int LPH>
struct LaunchHub {
    static int run(const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP, float* agg, int R,
                   int64_t NQ, int apply_gelu, int HT, HgtHubBuffers hb, unsigned ny, int64_t ld_out, hipStream_t stream) {
        const int raw = (apply_gelu == 2);
        const int dkp = VEC * LPH;
        const int64_t cells = (int64_t)pv.L.max_hubs * (2 * HT + HT * dkp);
        k_hub_init<<<(unsigned)((cells + 255) / 256), 256, 0, stream>>>(pv.hdr, pv.hub_list, hb, HT, HT * dkp);
        if (hb.part) k_hub_init_parts<<<2048, 256, 0, stream>>>(pv.hdr, pv.hub_list, hb, HT, HT * dkp, R);
        k_hub_max<<<HUB_GRID_WAVES / 4, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, logits, R, HT, hb);
        dim3 hgrid(HUB_GRID_WAVES / 4, ny);
        if (rteV)
            k_hub_accumulate<VEC, LPH, true><<<hgrid, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, pv.esrc, pv.ertei, logits, V, rteV,
                                                                        msgP, R, HT, hb, raw);
        else
            k_hub_accumulate<VEC, LPH, false><<<hgrid, 256, 0, stream>>>(pv.hdr, pv.hub_list, pv.segptr, pv.esrc, pv.ertei, logits, V, rteV,
                                                                         msgP, R, HT, hb, raw);
        k_hub_finalize<<<256, 256, 0, stream>>>(pv.hdr, pv.hub_list, hb, agg, HT, dkp, NQ, apply_gelu, ld_out, R);
        return HGT_OK;
    }
};

}  // namespace

int hgt_launch_hub(int vec, int lph, const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP,
                   float* agg, int R, int64_t NQ, int apply_gelu, int HT, HgtHubBuffers hb, unsigned ny, int64_t ld_out, hipStream_t stream) {
    return dispatch_layout<LaunchHub>(vec, lph, pv, logits, V, rteV, msgP, agg, R, NQ, apply_gelu, HT, hb, ny, ld_out, stream);
}
