// Pieces shared by the split-bf16 MFMA kernels (hgt_gemm_bf16x3.hip) and the aggregate kernel's fused
// a_linear + node-update epilogue (hgt_edge.hip): LDS slab geometry, the weight fragment geometry of
// hgt_split_weights, the fp32 -> bf16 hi/mid split and the register shuffles of the epilogues.
#pragma once
#include "hgt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 64;          // rows per workgroup
constexpr int BNP = 256;        // output columns per pass (8 waves x 32)
constexpr int KC = 16;          // k per MFMA (v_mfma_f32_32x32x16_bf16)
constexpr int KP = 256;         // k panel kept in LDS
constexpr int A_STRIDE = KP * 2 + 16;   // bytes
constexpr int A_PLANE = BM * A_STRIDE;          // 33792
constexpr int W_PLANE_ELEMS = BNP * KC;         // bf16 elements per plane of one (pass, k-chunk) tile = 8 x 64 x 8

__device__ __forceinline__ float gelu_erf_(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// round-to-nearest-even fp32 -> bf16 (inputs are finite)
__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// fp32 pair -> packed bf16 hi / mid terms (a = hi + mid + O(2^-18 a)) on gfx950's hardware conversion: v_cvt_pk_bf16_f32
// (round to nearest even, identical to bf16_rne for finite inputs), 5 instructions per pair instead of ~20
typedef __bf16 hgt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hgt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid) {
    const hgt_f32x2 v = {a, b};
    const hgt_bf16x2 h = __builtin_convertvector(v, hgt_bf16x2);
    const hgt_f32x2 r = v - __builtin_convertvector(h, hgt_f32x2);
    const hgt_bf16x2 m = __builtin_convertvector(r, hgt_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
}

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid) {
    split2(v.x, v.y, hi.x, mid.x);
    split2(v.z, v.w, hi.y, mid.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16 hi / lo split (precision "f16x3", round 3).  The same three products per MFMA step -- lo*hi + hi*lo + hi*hi -- but with
// 11 + 11 mantissa bits per operand instead of 8 + 8: relative error of a product ~2^-22 instead of ~2^-17 (measured end to end:
// 2e-7 against 1.4e-5 on the sampled-batch layer, tools/lab/precision_study.py), at the same MFMA count, LDS and register
// footprint.  fp16 has 5 exponent bits, so every operand ROW is scaled by a power of two (exact) that puts its largest magnitude
// into [2^14, 2^15): no overflow, and the lo term of every element that matters stays a normal number (gfx950's f16 MFMA honours
// subnormal inputs -- tools/lab/probe_f16.hip -- so smaller elements degrade gracefully to an absolute error of 2^-25 of the row
// maximum).  The inverse scales are applied to the fp32 accumulators in the epilogues.
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 hgt_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hgt_f16x8 __attribute__((ext_vector_type(8)));

// bits of max |x| over a row -> (scale, inverse scale), both exact powers of two: scale * max in [2^14, 2^15)
__device__ __forceinline__ void f16_row_scale(unsigned max_abs_bits, float& scale, float& inv) {
    unsigned e = max_abs_bits >> 23;                     // biased exponent of the row maximum (sign bit is 0)
    e = e < 20u ? 20u : (e > 240u ? 240u : e);           // all-zero / denormal rows and infinities: clamped, results stay finite / propagate
    scale = __builtin_bit_cast(float, (268u - e) << 23); // 2^(14 - (e - 127))
    inv = __builtin_bit_cast(float, (e - 14u) << 23);    // 2^((e - 127) - 14)
}

// fp32 pair (already scaled) -> packed fp16 hi / lo terms: v_cvt_pk_f16_f32 (round to nearest even), 5 instructions per pair
__device__ __forceinline__ void split2_f16(float a, float b, unsigned& hi, unsigned& lo) {
    const hgt_f32x2 v = {a, b};
    const hgt_f16x2 h = __builtin_convertvector(v, hgt_f16x2);
    const hgt_f32x2 r = v - __builtin_convertvector(h, hgt_f32x2);
    const hgt_f16x2 l = __builtin_convertvector(r, hgt_f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// format-generic forms: F16 = false: bf16 hi / mid (scale ignored, pass 1.0f); F16 = true: fp16 hi / lo of v * scale
template <bool F16>
__device__ __forceinline__ void split4_t(const float4 v, float scale, uint2& hi, uint2& lo) {
    if constexpr (F16) {
        split2_f16(v.x * scale, v.y * scale, hi.x, lo.x);
        split2_f16(v.z * scale, v.w * scale, hi.y, lo.y);
    } else {
        split4(v, hi, lo);
    }
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma32_t(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hgt_f16x8, a), __builtin_bit_cast(hgt_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

typedef float hgt_f32x4v __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ hgt_f32x4v mfma16_t(bf16x8 a, bf16x8 b, hgt_f32x4v c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hgt_f16x8, a), __builtin_bit_cast(hgt_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ void split2_t(float a, float b, float scale, unsigned& hi, unsigned& lo) {
    if constexpr (F16) split2_f16(a * scale, b * scale, hi, lo);
    else split2(a, b, hi, lo);
}
template <bool F16>
__device__ __forceinline__ void split1_t(float a, float scale, unsigned short& hi, unsigned short& lo) {
    if constexpr (F16) {
        const _Float16 h = (_Float16)(a * scale);
        const _Float16 l = (_Float16)(a * scale - (float)h);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    } else {
        hi = bf16_rne(a);
        lo = bf16_rne(a - bf16_to_f32(hi));
    }
}

// max over the 64 lanes of a wavefront of a NON-NEGATIVE float's bits (compare as unsigned), wave-uniform result:
// four DPP steps inside the rows of 16 lanes, then the four row results through SGPRs
__device__ __forceinline__ unsigned wave_max_bits(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
// max over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15) of a NON-NEGATIVE float's bits: every lane of the row gets the result.
// Four DPP moves -- a __shfl_xor is a ds_bpermute: an LDS round trip per step (the per-panel row maxima of hgt_gemm_tile.hip).
__device__ __forceinline__ unsigned row16_max_bits(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}
__device__ __forceinline__ unsigned abs_bits4(const float4 v) {
    return __builtin_bit_cast(unsigned, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}

// tanh of the typed input adapter (model.py:74) as  1 - 2 / (e^(2x) + 1)  on the hardware exp2 / rcp (each ~1 ulp: |error| <= ~1.5e-7,
// saturates to +-1 without a branch; libm's tanhf is ~40 instructions with two of them) -- one definition for the fused epilogues and
// hgt_tanh_inplace, so that both forms of the adapter publish the same values
__device__ __forceinline__ float hgt_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // e^(2x) = 2^(2x log2 e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// 4x4 transpose between the 4 lanes of a quad and 4 registers: lane j reg i <- lane i reg j
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, bool o1, bool o2) {
    float t;
    t = dpp_mov_f<0xB1>(o1 ? v0 : v1); if (o1) v0 = t; else v1 = t;    // quad_perm [1,0,3,2]
    t = dpp_mov_f<0xB1>(o1 ? v2 : v3); if (o1) v2 = t; else v3 = t;
    t = dpp_mov_f<0x4E>(o2 ? v0 : v2); if (o2) v0 = t; else v2 = t;    // quad_perm [2,3,0,1]
    t = dpp_mov_f<0x4E>(o2 ? v1 : v3); if (o2) v1 = t; else v3 = t;
}

__device__ __forceinline__ float strided8_sum(float v) {   // sum over the 8 lanes {j, j+4, ..., j+28} of a 32-lane half
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

}  // namespace
