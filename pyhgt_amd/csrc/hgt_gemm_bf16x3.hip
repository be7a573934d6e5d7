// Typed (grouped) linear layer, split-bf16 x3 variant ("A-stationary").
//
//   y[n, :] = prologue(x[n, :]) @ W[type(n)]^T + b[type(n)]
//
// fp32 operands are split into two bf16 terms  a = a_hi + a_mid  (a_hi = bf16(a), a_mid = bf16(a - a_hi))
// and a product is evaluated as  a_mid*b_hi + a_hi*b_mid + a_hi*b_hi  on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate): 3 MFMAs at 16x the fp32-MFMA rate, relative error of
// a product <= ~3*2^-18.  gfx950 has no xf32/TF32 MFMA, so this is the only reduced-cost route for
// fp32 operands.  Opt-in (HGTConv(precision="bf16x3")), parity-tested at the same 1e-4 bound.
//
// Shape of the problem: M = millions of node rows, K = d (256), N = 3d: output-heavy and, with the
// MFMA cost cut 5x, bound by HBM and by latency/synchronisation -- a k-loop with LDS-staged tiles and
// one or two barriers per k-step (hgt_gemm.hip) spends most of its time waiting.  So here:
//   * a workgroup (8 waves) owns 64 rows and ALL output columns: the 64 x K slab of x is read from
//     HBM exactly once, as whole 1 KB rows, split to bf16 hi/mid ONCE and kept in LDS (66 KB, so two
//     workgroups share a CU and one's slab load / epilogue overlaps the other's MFMAs);
//   * W never goes through LDS: hgt_split_weights pre-splits it and stores it in MFMA-FRAGMENT order
//     ([pass][k-chunk][plane][32-column tile][lane][8 bf16]), so a wave's B fragment is one coalesced
//     1 KB load straight into registers from the L2-resident 1.5 MB image, prefetched four k-chunks
//     ahead in named register stages;
//   * every wave owns a 64-row x 32-column strip of the current 256-column pass: A fragments (2 row
//     tiles x hi/mid) come from the read-only LDS slab, 6 MFMAs per 4 ds_read_b128 + 2 global loads,
//     and -- because nothing is written to LDS in the main loop -- there is NO barrier in it.
// The 528 B LDS row stride puts the 16 lanes of a ds_read_b128 group on 16 distinct 4-bank slots.
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

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned short h[4], m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = bf16_rne(f[i]);
        m[i] = bf16_rne(f[i] - bf16_to_f32(h[i]));
    }
    hi = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    mid = make_uint2((unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16));
}

// W [n_groups][n_out][k] fp32 -> [g][pass][kchunk][plane][col tile 8][lane 64][8] bf16 (zero padded): the 8 bf16 of
// (col tile ct, lane l) are W[pass*256 + ct*32 + (l&31)][kchunk*16 + (l>>5)*8 .. +8] = one lane's B fragment.
__global__ void k_split_weights(const float* __restrict__ W, int64_t wgs, int n_groups, int k, int n_out, int n_pass, int n_kc,
                                unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_group = (int64_t)n_pass * n_kc * W_PLANE_ELEMS;
    if (i >= per_group * n_groups) return;
    const int g = (int)(i / per_group);
    int64_t r = i - (int64_t)g * per_group;
    const int e = (int)(r % 8);
    r /= 8;
    const int l = (int)(r % 64);
    r /= 64;
    const int ct = (int)(r % 8);
    r /= 8;
    const int kc = (int)(r % n_kc);
    const int pass = (int)(r / n_kc);
    const int n = pass * BNP + ct * 32 + (l & 31), kidx = kc * KC + (l >> 5) * 8 + e;
    float v = 0.0f;
    if (n < n_out && kidx < k) v = W[(int64_t)g * wgs + (int64_t)n * k + kidx];
    const unsigned short h = bf16_rne(v);
    const unsigned short m = bf16_rne(v - bf16_to_f32(h));
    const int64_t tile = (((int64_t)g * n_pass + pass) * n_kc + kc) * 2;
    const int within = (ct * 64 + l) * 8 + e;
    out[(tile + 0) * W_PLANE_ELEMS + within] = h;
    out[(tile + 1) * W_PLANE_ELEMS + within] = m;
}

// Load one K panel of the 64-row x slab: whole rows, split into bf16 hi/mid planes ONCE, stored to LDS.
// Done in two halves of 4 float4 per thread to keep the live register set small (the kernel is capped at 128 VGPRs
// so that two workgroups fit a CU; an 8-deep version spilled ~230 B per lane to scratch = +2.9 GB of HBM traffic at c2).
template <int PROLOGUE>
__device__ __forceinline__ void load_a_panel(int kp0, int tid, const int* s_rid, const float* __restrict__ x, int64_t ldx, int k,
                                             int vec_ok, unsigned char* sA, bool wait_readers) {
    if (wait_readers) __syncthreads();   // every wave is done reading the previous panel
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 512 * (half * 4 + j);
            const int kk = kp0 + (f & 63) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const int rid = s_rid[f >> 6];
            if (rid >= 0 && kk < k) {
                const float* px = x + (int64_t)rid * ldx + kk;
                if (vec_ok && kk + 3 < k) {
                    a = *reinterpret_cast<const float4*>(px);
                } else {
                    a.x = px[0];
                    if (kk + 1 < k) a.y = px[1];
                    if (kk + 2 < k) a.z = px[2];
                    if (kk + 3 < k) a.w = px[3];
                }
                if (PROLOGUE == 1) { a.x = gelu_erf_(a.x); a.y = gelu_erf_(a.y); a.z = gelu_erf_(a.z); a.w = gelu_erf_(a.w); }
            }
            av[j] = a;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 512 * (half * 4 + j);
            const int r = f >> 6, cb = (f & 63) * 8;
            uint2 hi, mid;
            split4(av[j], hi, mid);
            *reinterpret_cast<uint2*>(sA + r * A_STRIDE + cb) = hi;
            *reinterpret_cast<uint2*>(sA + A_PLANE + r * A_STRIDE + cb) = mid;
        }
    }
    __syncthreads();   // panel visible
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

// epilogue of one 256-column pass.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Narrow (4 B per lane) stores are issue-bound, so each group of 4 registers (4 consecutive rows, one column per lane)
// is transposed inside the lane quad: afterwards a lane holds 4 consecutive COLUMNS of one row and writes one 16 B
// store (one wave instruction = 8 rows x 128 B) -- 4x fewer store instructions.
__device__ __forceinline__ void store_pass(f32x16 (&acc)[2], int pass, int wave, int lane, int g, int n_out, int nrows, int row0,
                                           const int* s_rid, const float* __restrict__ bias, int64_t bgs, float* __restrict__ out0,
                                           float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos) {
    const int col = pass * BNP + wave * 32 + ((lane & 31) >> 2) * 4;      // first of this lane's 4 columns after the transpose
    const bool col_ok = col < n_out;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok && bias) b4 = *reinterpret_cast<const float4*>(bias + (int64_t)g * bgs + col);
    const int blk = col_ok ? col / block_cols : 0, cc = col - blk * block_cols;
    float* __restrict__ ob = (blk == 0) ? out0 : ((blk == 1) ? out1 : out2);
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v0 = acc[j][4 * q], v1 = acc[j][4 * q + 1], v2 = acc[j][4 * q + 2], v3 = acc[j][4 * q + 3];
            quad_transpose(v0, v1, v2, v3, o1, o2);
            const int rt = j * 32 + (lane & 3) + 8 * q + 4 * (lane >> 5);
            if (col_ok && rt < nrows) {
                const int64_t orow = by_pos ? (int64_t)(row0 + rt) : (int64_t)s_rid[rt];
                *reinterpret_cast<float4*>(ob + orow * block_cols + cc) = make_float4(v0 + b4.x, v1 + b4.y, v2 + b4.z, v3 + b4.w);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
}

struct UpdateArgs {
    const float* xs;      // skip-connection input rows [*, ldxs]
    int64_t ldxs;
    const float* skip;    // [n_groups]
    const float* lnw;     // [n_groups][n_out] or nullptr
    const float* lnb;
    int use_norm;
};

__device__ __forceinline__ float strided8_sum(float v) {   // sum over the 8 lanes {j, j+4, ..., j+28} of a 32-lane half
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}

// Fused node update (conv.py:129-133) as the epilogue of the a_linear GEMM, single 256-column pass:
//   y = (acc + b) * sigmoid(skip[t]) + x * (1 - sigmoid(skip[t]));  out = LayerNorm_t(y)  (two-pass mean/variance)
// A row's 256 columns live in 8 waves x 8 lanes; partial sums meet in a small LDS table (the A slab is dead by then).
__device__ __forceinline__ void store_pass_update(f32x16 (&acc)[2], int wave, int lane, int g, int n_out, int nrows, const int* s_rid,
                                                  const float* __restrict__ bias, int64_t bgs, float* __restrict__ out,
                                                  const UpdateArgs& u, float* s_red) {
    const int col = wave * 32 + ((lane & 31) >> 2) * 4;
    const bool col_ok = col < n_out;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok && bias) b4 = *reinterpret_cast<const float4*>(bias + (int64_t)g * bgs + col);
    const float alpha = 1.0f / (1.0f + expf(-u.skip[g]));
    const bool o1 = lane & 1, o2 = lane & 2;
    const float inv_n = 1.0f / (float)n_out;
    float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u.use_norm && col_ok) {
        w4 = *reinterpret_cast<const float4*>(u.lnw + (int64_t)g * n_out + col);
        c4 = *reinterpret_cast<const float4*>(u.lnb + (int64_t)g * n_out + col);
    }
    // the two 32-row halves one after the other: half the live registers (so two workgroups still fit a CU)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float y[4][4];
        int64_t orow[4];
        int rts[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v0 = acc[j][4 * q], v1 = acc[j][4 * q + 1], v2 = acc[j][4 * q + 2], v3 = acc[j][4 * q + 3];
            quad_transpose(v0, v1, v2, v3, o1, o2);
            const int rt = j * 32 + (lane & 3) + 8 * q + 4 * (lane >> 5);
            rts[q] = rt;
            orow[q] = (rt < nrows) ? (int64_t)s_rid[rt] : -1;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok && orow[q] >= 0) xv = *reinterpret_cast<const float4*>(u.xs + orow[q] * u.ldxs + col);
            y[q][0] = col_ok ? (v0 + b4.x) * alpha + xv.x * (1.0f - alpha) : 0.0f;
            y[q][1] = col_ok ? (v1 + b4.y) * alpha + xv.y * (1.0f - alpha) : 0.0f;
            y[q][2] = col_ok ? (v2 + b4.z) * alpha + xv.z * (1.0f - alpha) : 0.0f;
            y[q][3] = col_ok ? (v3 + b4.w) * alpha + xv.w * (1.0f - alpha) : 0.0f;
        }
        if (u.use_norm) {
            // pass 1: mean
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float ps = strided8_sum(y[q][0] + y[q][1] + y[q][2] + y[q][3]);
                if (((lane & 31) >> 2) == 0) s_red[rts[q] * 8 + wave] = ps;
            }
            __syncthreads();
            float mean[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(&s_red[rts[q] * 8]);
                const float4 b = *reinterpret_cast<const float4*>(&s_red[rts[q] * 8 + 4]);
                mean[q] = (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * inv_n;
            }
            __syncthreads();
            // pass 2: variance
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float d0 = y[q][0] - mean[q], d1 = y[q][1] - mean[q], d2 = y[q][2] - mean[q], d3 = y[q][3] - mean[q];
                if (!col_ok) d0 = d1 = d2 = d3 = 0.0f;
                const float ps = strided8_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
                if (((lane & 31) >> 2) == 0) s_red[rts[q] * 8 + wave] = ps;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(&s_red[rts[q] * 8]);
                const float4 b = *reinterpret_cast<const float4*>(&s_red[rts[q] * 8 + 4]);
                const float rstd = rsqrtf((a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * inv_n + 1e-5f);
                if (col_ok && orow[q] >= 0)
                    *reinterpret_cast<float4*>(out + orow[q] * n_out + col) =
                        make_float4((y[q][0] - mean[q]) * rstd * w4.x + c4.x, (y[q][1] - mean[q]) * rstd * w4.y + c4.y,
                                    (y[q][2] - mean[q]) * rstd * w4.z + c4.z, (y[q][3] - mean[q]) * rstd * w4.w + c4.w);
            }
            __syncthreads();   // s_red is reused by the second half
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col_ok && orow[q] >= 0)
                    *reinterpret_cast<float4*>(out + orow[q] * n_out + col) = make_float4(y[q][0], y[q][1], y[q][2], y[q][3]);
        }
    }
}

template <int PROLOGUE, bool UPD>
__global__ __launch_bounds__(512, UPD ? 2 : 4) void k_typed_linear_split(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
    int n_groups, int k, int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int vec_ok,
    UpdateArgs upd) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[2 * A_PLANE];        // [plane][64][528]
    __shared__ int s_rid[BM];

    const int slot = blockIdx.x;
    int g = 0, gbeg = 0, gend = 0, tiles_before = 0;
    for (; g < n_groups; ++g) {
        gbeg = group_off[g];
        gend = group_off[g + 1];
        int nt = (gend - gbeg + BM - 1) / BM;
        if (slot < tiles_before + nt) break;
        tiles_before += nt;
    }
    if (g >= n_groups) return;
    const int row0 = gbeg + (slot - tiles_before) * BM;
    const int nrows = min(BM, gend - row0);

    const int tid = threadIdx.x;
    if (tid < BM) s_rid[tid] = (tid < nrows) ? rows[row0 + tid] : -1;
    __syncthreads();

    const int n_pass = (n_out + BNP - 1) / BNP;
    const int n_kc = ((k + KC - 1) / KC + 3) & ~3;   // padded to a multiple of 4 k-chunks
    const int n_panel = (k + KP - 1) / KP;
    const int total = n_pass * n_kc;
    const int lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, khalf = lane >> 5;
    // this wave's B fragments: plane stride W_PLANE_ELEMS, tile stride 2 * W_PLANE_ELEMS (bf16 elements)
    const unsigned short* __restrict__ wfrag = wsplit + (int64_t)g * total * 2 * W_PLANE_ELEMS + (wave * 64 + lane) * 8;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // B fragment stream: k-chunk T of the flattened (pass, k-chunk) sequence, prefetched HGT_NSTAGE ahead in named register stages.
    // n_kc is a multiple of 4 (hgt_split_weights zero-pads K to a multiple of 64, the A slab is zero-filled
    // beyond k), so the 4-step body needs no guards and a pass boundary always falls between bodies.
#ifndef HGT_NSTAGE
#define HGT_NSTAGE 2   // measured: 2 stages (1.83 ms at c2) beat 4 (1.88 ms): fewer live registers -> fewer spills at the 128-VGPR cap
#endif
    bf16x8 s0h, s0m, s1h, s1m;
#if HGT_NSTAGE == 4
    bf16x8 s2h, s2m, s3h, s3m;
#endif
#define HGT_LOAD_STAGE(S, T)                                                                          \
    {                                                                                                 \
        const unsigned short* t_ = wfrag + (int64_t)min((T), total - 1) * 2 * W_PLANE_ELEMS;          \
        s##S##h = *reinterpret_cast<const bf16x8*>(t_);                                               \
        s##S##m = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                               \
    }
    HGT_LOAD_STAGE(0, 0)
    HGT_LOAD_STAGE(1, 1)
#if HGT_NSTAGE == 4
    HGT_LOAD_STAGE(2, 2)
    HGT_LOAD_STAGE(3, 3)
#endif

    load_a_panel<PROLOGUE>(0, tid, s_rid, x, ldx, k, vec_ok, sA, false);

#define HGT_STEP(S, T, KCP)                                                                                        \
    {                                                                                                              \
        const int ao = frow * A_STRIDE + ((KCP) * KC + khalf * 8) * 2;                                             \
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(sA + ao);                                              \
        const bf16x8 am0 = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao);                                    \
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(sA + ao + 32 * A_STRIDE);                              \
        const bf16x8 am1 = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao + 32 * A_STRIDE);                    \
        const bf16x8 bh = s##S##h, bm = s##S##m;                                                                   \
        HGT_LOAD_STAGE(S, (T) + HGT_NSTAGE)                                                                        \
        /* small terms first, hi*hi last; the two accumulators alternate */                                        \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am0, bh, acc[0], 0, 0, 0);                                \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am1, bh, acc[1], 0, 0, 0);                                \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bm, acc[0], 0, 0, 0);                                \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bm, acc[1], 0, 0, 0);                                \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc[0], 0, 0, 0);                                \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc[1], 0, 0, 0);                                \
    }

    for (int pass = 0; pass < n_pass; ++pass) {
        for (int panel = 0; panel < n_panel; ++panel) {
            if (n_panel > 1 && (pass | panel) != 0) load_a_panel<PROLOGUE>(panel * KP, tid, s_rid, x, ldx, k, vec_ok, sA, true);
            const int nkc_p = min(KP / KC, n_kc - panel * (KP / KC));
            const int tbase = pass * n_kc + panel * (KP / KC);
            for (int kq = 0; kq < nkc_p; kq += 4) {
#if HGT_NSTAGE == 4
                HGT_STEP(0, tbase + kq, kq)
                HGT_STEP(1, tbase + kq + 1, kq + 1)
                HGT_STEP(2, tbase + kq + 2, kq + 2)
                HGT_STEP(3, tbase + kq + 3, kq + 3)
#else
                HGT_STEP(0, tbase + kq, kq)
                HGT_STEP(1, tbase + kq + 1, kq + 1)
                HGT_STEP(0, tbase + kq + 2, kq + 2)
                HGT_STEP(1, tbase + kq + 3, kq + 3)
#endif
            }
        }
        if constexpr (UPD) {
            __syncthreads();   // every wave left the MFMA loop: the A slab can be reused as the reduction table
            store_pass_update(acc, wave, lane, g, n_out, nrows, s_rid, bias, bgs, out0, upd, reinterpret_cast<float*>(sA));
        } else {
            store_pass(acc, pass, wave, lane, g, n_out, nrows, row0, s_rid, bias, bgs, out0, out1, out2, block_cols, by_pos);
        }
    }
#undef HGT_STEP
#undef HGT_LOAD_STAGE
}

static inline void split_dims(int k, int n_out, int* n_pass, int* n_kc) {
    *n_pass = (n_out + BNP - 1) / BNP;
    *n_kc = ((k + KC - 1) / KC + 3) & ~3;   // k-chunks padded to a multiple of 4 (zero tiles)
}

}  // namespace

extern "C" int hgt_split_weights_bytes(int32_t n_groups, int32_t k, int32_t n_out, uint64_t* out) {
    if (!out || n_groups <= 0 || k <= 0 || n_out <= 0) return HGT_ERR_INVALID_ARG;
    int n_pass, n_kstep;
    split_dims(k, n_out, &n_pass, &n_kstep);
    *out = (uint64_t)n_groups * n_pass * n_kstep * 2 * W_PLANE_ELEMS * 2;
    return HGT_OK;
}

extern "C" int hgt_split_weights(const float* W, int64_t w_group_stride, int32_t n_groups, int32_t k, int32_t n_out, void* w_split,
                                 void* stream) {
    if (!W || !w_split || n_groups <= 0 || k <= 0 || n_out <= 0) return HGT_ERR_INVALID_ARG;
    int n_pass, n_kstep;
    split_dims(k, n_out, &n_pass, &n_kstep);
    const int64_t total = (int64_t)n_groups * n_pass * n_kstep * W_PLANE_ELEMS;
    k_split_weights<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, w_group_stride, n_groups, k, n_out, n_pass, n_kstep,
                                                                                     (unsigned short*)w_split);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_typed_linear_bf16x3(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                       int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                       int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                       int32_t out_by_position, int32_t prologue, void* stream_) {
    if (!x || !rows || !group_off || !w_split || !out0 || n_groups <= 0 || n_rows < 0 || k <= 0 || n_out <= 0 || block_cols <= 0)
        return HGT_ERR_INVALID_ARG;
    const int n_blocks_out = (n_out + block_cols - 1) / block_cols;
    if (n_blocks_out > 3 || (n_blocks_out > 1 && !out1) || (n_blocks_out > 2 && !out2)) return HGT_ERR_INVALID_ARG;
    if (prologue != 0 && prologue != 1) return HGT_ERR_INVALID_ARG;
    if (((n_out | block_cols) & 3) != 0) return HGT_ERR_UNSUPPORTED;   // 16-byte epilogue stores; use hgt_typed_linear (fp32) instead
    if (n_rows == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;   // device-side group sizes: launch the upper bound
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const int vec_ok = (ldx % 4 == 0) && (k % 4 == 0) && (((uintptr_t)x & 15) == 0);
    UpdateArgs noupd = {nullptr, 0, nullptr, nullptr, nullptr, 0};
    if (prologue == 0)
        k_typed_linear_split<0, false><<<(unsigned)row_tiles, 512, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                                (const unsigned short*)w_split, bias, b_group_stride,
                                                                                out0, out1, out2, block_cols, out_by_position, vec_ok, noupd);
    else
        k_typed_linear_split<1, false><<<(unsigned)row_tiles, 512, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                                (const unsigned short*)w_split, bias, b_group_stride,
                                                                                out0, out1, out2, block_cols, out_by_position, vec_ok, noupd);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

// a_linear + gated skip + LayerNorm in one kernel (n_out <= 256, n_out % 4 == 0): see store_pass_update.
extern "C" int hgt_linear_update_bf16x3(const float* agg, int64_t ld_agg, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                        int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                        int64_t b_group_stride, const float* x_skip, int64_t ld_skip, const float* skip,
                                        const float* ln_w, const float* ln_b, int32_t use_norm, float* out, void* stream_) {
    if (!agg || !rows || !group_off || !w_split || !x_skip || !skip || !out || n_groups <= 0 || n_rows < 0 || k <= 0 || n_out <= 0)
        return HGT_ERR_INVALID_ARG;
    if (use_norm && (!ln_w || !ln_b)) return HGT_ERR_INVALID_ARG;
    if (n_out > BNP || (n_out & 3) != 0 || (ld_skip & 3) != 0 || ((uintptr_t)x_skip & 15) != 0) return HGT_ERR_UNSUPPORTED;
    if (n_rows == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const int vec_ok = (ld_agg % 4 == 0) && (k % 4 == 0) && (((uintptr_t)agg & 15) == 0);
    UpdateArgs u = {x_skip, ld_skip, skip, ln_w, ln_b, use_norm};
    k_typed_linear_split<0, true><<<(unsigned)row_tiles, 512, 0, stream>>>(agg, ld_agg, rows, group_off, n_groups, k, n_out,
                                                                           (const unsigned short*)w_split, bias, b_group_stride, out,
                                                                           nullptr, nullptr, n_out, 0, vec_ok, u);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
