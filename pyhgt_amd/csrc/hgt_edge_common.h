// Helpers shared by the edge-phase translation units (hgt_edge_logits.hip, hgt_edge_agg_valu.hip, hgt_edge_agg_mfma.hip,
// hgt_edge_hub.hip): lane-group reductions, row loads, the register-resident relation mat-vec of the vector-ALU kernels, the
// (vec, lanes-per-head) layout dispatch, and the structs that cross translation units.
#pragma once
#include "hgt_common.h"

// per-hub accumulators of the hub path (hgt_edge_hub.hip); global scope: passed between translation units
struct HgtHubBuffers {
    int* mx;      // [max_hubs][HT] ordered-int max logit
    float* l;     // [max_hubs][HT]
    float* acc;   // [max_hubs][HT * DKP]
    int64_t q_lo, q_hi;   // only hubs in [q_lo, q_hi) are processed (a target block of the multi-GPU path); q_hi <= 0: all
    float* part;  // deterministic mode (HGT_FLAG_DETERMINISTIC_HUBS): [max_hubs][(R+1) * pieces][HT * DKP] partial rows, one slot per
    float* lpart; // piece, and [max_hubs][(R+1) * pieces][HT] partial exp-sums -- summed by k_hub_finalize in piece order; NULL: atomics
};
constexpr int HGT_HUB_PIECES = 32;   // pieces per (hub, relation) range (hgt_edge_hub.hip)

// arguments of the fused node update (hgt_fused_update.h)
// A slice [lo, hi) of the plan's R + 1 relation buckets, and the softmax state carried from slice to slice (multi-GPU path:
// relation id = source bucket * R + relation, one slice per bucket of arrived source rows -- pyhgt_amd/dist.py).
struct HgtRelSlice {
    int lo, hi;
    float* state;     // f32[NQ][H][2] = (reference, exp-sum) per (target, head); NULL: the whole layer in one launch
    int has_prev;     // an earlier slice left state + un-normalised rows (in agg): merge with them
    int more;         // further slices follow: leave state + un-normalised rows instead of finishing
    int64_t q_lo;     // first target row of the launch (a multiple of 64; target blocks of the multi-GPU path), 0 = the whole graph
};

struct HgtFusedUpdate {
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
    int64_t q_lo;                    // first target row of the launch (a multiple of 64): workgroup b owns rows q_lo + 64 b ..
    int ring;                        // HGT_FLAG_RING_AGGREGATE: k_edge_aggregate_update_ring where it is instantiated
    int small32;                     // V / logits / temporal rows lie below 4 GiB from their bases (launcher: the S32 instantiation, 32-bit lane offsets)
};


// hub kernels (hgt_edge_hub.hip): max / exp-sum + weighted sum / finalize for the targets the plan marked as hubs.
// vec / lph = the (possibly head-group split) layout the calling aggregation kernel runs with, ny = number of head groups.
int hgt_launch_hub(int vec, int lph, const HgtPlanView& pv, const float* logits, const float* V, const float* rteV, const float* msgP,
                   float* agg, int R, int64_t NQ, int apply_gelu, int HT, HgtHubBuffers hb, unsigned ny, int64_t ld_out, hipStream_t stream);
// apply_gelu: 0 = softmax-normalised sum, 1 = + gelu, 2 = raw weighted sum (`logits` holds the edge weights; hgt_edge_spmm)

// vector-ALU aggregation kernels (hgt_edge_agg_valu.hip; relation transforms as per-segment mat-vecs out of registers)
int hgt_valu_aggregate(const HgtPlanView& pv, int dk_pad, const float* logits, const float* V, const float* rteV, const float* msgP,
                       float* agg, int R, int64_t NQ, int apply_gelu, int H, HgtHubBuffers hb, hipStream_t stream);
int hgt_valu_aggregate_update(const HgtPlanView& pv, int dk_pad, const float* logits, const float* V, const float* rteV,
                              const float* msgP, float* agg, int R, int64_t NQ, int H, HgtHubBuffers hb, int32_t* pending,
                              HgtFusedUpdate fu, hipStream_t stream);

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

constexpr int HGT_SUB = 16;       // targets per wavefront of the aggregation kernels
constexpr float HGT_NEG = -1.0e30f;

template <template <int, int> class Launcher, typename... Args>
int dispatch_layout(int vec, int lph, Args... args) {
#define HGT_CASE(V, L) \
    if (vec == V && lph == L) return Launcher<V, L>::run(args...);
#ifdef HGT_DEV_LAYOUTS   // development builds: d = 256 / 8 heads and d = 64 / 4 heads only
    HGT_CASE(4, 8) HGT_CASE(1, 16)
#else
    HGT_CASE(1, 4) HGT_CASE(2, 4) HGT_CASE(4, 4) HGT_CASE(8, 4)
    HGT_CASE(1, 8) HGT_CASE(2, 8) HGT_CASE(4, 8) HGT_CASE(8, 8)
    HGT_CASE(1, 16) HGT_CASE(2, 16) HGT_CASE(4, 16) HGT_CASE(8, 16)
    HGT_CASE(1, 32) HGT_CASE(2, 32) HGT_CASE(4, 32) HGT_CASE(8, 32)
    HGT_CASE(1, 64) HGT_CASE(2, 64) HGT_CASE(4, 64) HGT_CASE(8, 64)
#endif
#undef HGT_CASE
    return HGT_ERR_UNSUPPORTED;
}


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
