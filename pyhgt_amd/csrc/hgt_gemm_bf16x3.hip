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
// MFMA cost cut 5x, bound by HBM and by latency -- a k-loop that re-fetches a 128-B piece of every
// x row per step (the fp32 kernel in hgt_gemm.hip) spends its time waiting.  So here:
//   * a workgroup (8 waves) owns 64 rows and ALL output columns: the 64 x K slab of x is read from
//     HBM exactly once, as whole 1 KB rows, split to bf16 hi/mid ONCE and kept in LDS (66 KB);
//   * W is pre-split and pre-tiled by hgt_split_weights into contiguous 16 KB [256 cols][32 k] tiles
//     (hi and mid planes), L2-resident (1.5 MB), streamed through a double-buffered LDS ring with the
//     next tile's global loads in flight under the current tile's MFMAs; one barrier per k-step;
//   * waves are laid out 2 (rows) x 4 (cols): each wave owns a 32 x 64 strip of the 64 x 256 pass,
//     A fragments are shared by its two 32x32 tiles (6 MFMAs per 6 ds_read_b128).
// LDS row strides (528 B for A, 80 B for B) put the 16 lanes of a ds_read_b128 group on 16
// distinct 4-bank slots (conflict free).
#include "hgt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 64;          // rows per workgroup
constexpr int BNP = 256;        // output columns per pass
constexpr int BK = 32;          // k per streamed W tile
constexpr int KP = 256;         // k panel kept in LDS
constexpr int A_STRIDE = KP * 2 + 16;   // bytes
constexpr int B_STRIDE = BK * 2 + 16;   // bytes
constexpr int A_PLANE = BM * A_STRIDE;          // 33792
constexpr int B_PLANE = BNP * B_STRIDE;         // 20480
constexpr int W_TILE_ELEMS = BNP * BK;          // bf16 elements per (plane) tile

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

// W [n_groups][n_out][k] fp32 -> tiles [g][pass][kstep][plane][256][32] bf16, zero padded
__global__ void k_split_weights(const float* __restrict__ W, int64_t wgs, int n_groups, int k, int n_out, int n_pass, int n_kstep,
                                unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_group = (int64_t)n_pass * n_kstep * W_TILE_ELEMS;
    if (i >= per_group * n_groups) return;
    const int g = (int)(i / per_group);
    int64_t r = i - (int64_t)g * per_group;
    const int kk = (int)(r % BK);
    r /= BK;
    const int row = (int)(r % BNP);
    r /= BNP;
    const int ks = (int)(r % n_kstep);
    const int pass = (int)(r / n_kstep);
    const int n = pass * BNP + row, kidx = ks * BK + kk;
    float v = 0.0f;
    if (n < n_out && kidx < k) v = W[(int64_t)g * wgs + (int64_t)n * k + kidx];
    const unsigned short h = bf16_rne(v);
    const unsigned short m = bf16_rne(v - bf16_to_f32(h));
    const int64_t tile = (((int64_t)g * n_pass + pass) * n_kstep + ks) * 2;
    out[(tile + 0) * W_TILE_ELEMS + row * BK + kk] = h;
    out[(tile + 1) * W_TILE_ELEMS + row * BK + kk] = m;
}

template <int PROLOGUE>
__global__ __launch_bounds__(512, 2) void k_typed_linear_split(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
    int n_groups, int k, int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int vec_ok) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[2 * A_PLANE];        // [plane][64][528]
    __shared__ __attribute__((aligned(16))) unsigned char sB[2 * 2 * B_PLANE];    // [buf][plane][256][80]
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
    const int n_kstep = (k + BK - 1) / BK;
    const int n_panel = (k + KP - 1) / KP;
    const unsigned short* __restrict__ wg = wsplit + (int64_t)g * n_pass * n_kstep * 2 * W_TILE_ELEMS;

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, khalf = lane >> 5;

    // this thread's 8 float4 of the A slab: f = tid + 512*j -> row f>>6, float4 column f&63
    int a_rid[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_rid[j] = s_rid[(tid + 512 * j) >> 6];

    // The W tiles of this row tile form ONE linear stream of n_pass * n_kstep tiles (tile-major layout written
    // by hgt_split_weights).  They are prefetched FOUR k-steps ahead into four named register stages
    // (an L2 round trip is ~1 us, one k-step of MFMA work only ~0.4 us), staged through a 2-deep LDS ring.
    const int total = n_pass * n_kstep;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // W chunk handled by this thread: row slot q = tid>>2 is permuted inside each block of 8 rows
    // (0,4,1,5,2,6,3,7) so that the two rows written by one 8-lane ds_write_b128 group are 4 apart:
    // 4 * 80 B = 16 banks (mod 32) -> the group covers all 32 banks once (the natural order was 2-way).
    const int q_ = tid >> 2;
    const int brow = (q_ & ~7) + ((q_ & 1) << 2) + ((q_ >> 1) & 3);
    const int gch0 = brow * 4 + (tid & 3);                            // chunk index inside the 16 KB plane tile
    const int bo0 = brow * B_STRIDE + (tid & 3) * 16;                 // LDS offset of that chunk
    const int bo1 = (brow + 128) * B_STRIDE + (tid & 3) * 16;         // ... of chunk gch0 + 512

    uint4 s0a, s0b, s0c, s0d, s1a, s1b, s1c, s1d, s2a, s2b, s2c, s2d, s3a, s3b, s3c, s3d;
#define HGT_LOAD_STAGE(S, T)                                                                  \
    if ((T) < total) {                                                                        \
        const unsigned short* t_ = wg + (int64_t)(T) * 2 * W_TILE_ELEMS;                      \
        s##S##a = *reinterpret_cast<const uint4*>(t_ + gch0 * 8);                             \
        s##S##b = *reinterpret_cast<const uint4*>(t_ + (gch0 + 512) * 8);                     \
        s##S##c = *reinterpret_cast<const uint4*>(t_ + W_TILE_ELEMS + gch0 * 8);              \
        s##S##d = *reinterpret_cast<const uint4*>(t_ + W_TILE_ELEMS + (gch0 + 512) * 8);      \
    }
    HGT_LOAD_STAGE(0, 0)
    HGT_LOAD_STAGE(1, 1)
    HGT_LOAD_STAGE(2, 2)
    HGT_LOAD_STAGE(3, 3)

    auto load_a_panel = [&](int kp0) {
        float4 av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = tid + 512 * j;
            const int kk = kp0 + (f & 63) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const int rid = a_rid[j];
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
        __syncthreads();   // readers of the previous panel are done
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = tid + 512 * j;
            const int r = f >> 6, cb = (f & 63) * 8;
            uint2 hi, mid;
            split4(av[j], hi, mid);
            *reinterpret_cast<uint2*>(sA + r * A_STRIDE + cb) = hi;
            *reinterpret_cast<uint2*>(sA + A_PLANE + r * A_STRIDE + cb) = mid;
        }
    };

    auto epilogue = [&](int pass) {
        // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = pass * BNP + wn * 64 + j * 32 + (lane & 31);
            if (col < n_out) {
                const float bcol = bias ? bias[(int64_t)g * bgs + col] : 0.0f;
                const int blk = col / block_cols, cc = col - blk * block_cols;
                float* __restrict__ ob = (blk == 0) ? out0 : ((blk == 1) ? out1 : out2);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rt = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (rt < nrows && !(by_pos & 2 && acc[j][r] != 12345.0f)) {
                        const int64_t orow = (by_pos & 1) ? (int64_t)(row0 + rt) : (int64_t)s_rid[rt];
                        ob[orow * block_cols + cc] = acc[j][r] + bcol;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        }
    };

#define HGT_STEP(S, T)                                                                                              \
    if ((T) < total) {                                                                                              \
        const int pass_ = (T) / n_kstep, ksg_ = (T) - pass_ * n_kstep;        /* k-step inside the row of tiles */  \
        const int ksp_ = ksg_ & (KP / BK - 1);                                /* k-step inside the A panel */       \
        if (ksp_ == 0 && (n_panel > 1 || (T) == 0)) load_a_panel((ksg_ / (KP / BK)) * KP);                          \
        unsigned char* bb = sB + ((T) & 1) * 2 * B_PLANE;                                                           \
        *reinterpret_cast<uint4*>(bb + bo0) = s##S##a;                                                              \
        *reinterpret_cast<uint4*>(bb + bo1) = s##S##b;                                                              \
        *reinterpret_cast<uint4*>(bb + B_PLANE + bo0) = s##S##c;                                                    \
        *reinterpret_cast<uint4*>(bb + B_PLANE + bo1) = s##S##d;                                                    \
        __syncthreads(); /* tile T (and a fresh A panel) visible; ring slot (T+1)&1 is free again */                \
        HGT_LOAD_STAGE(S, (T) + 4)                                                                                  \
        _Pragma("unroll") for (int kc = 0; kc < BK / 16; ++kc) {                                                    \
            const int ao = (wm * 32 + frow) * A_STRIDE + (ksp_ * BK + kc * 16 + khalf * 8) * 2;                     \
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sA + ao);                                            \
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao);                                  \
            const int bo_ = (wn * 64 + frow) * B_STRIDE + (kc * 16 + khalf * 8) * 2;                                 \
            const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(bb + bo_);                                          \
            const bf16x8 bm0 = *reinterpret_cast<const bf16x8*>(bb + B_PLANE + bo_);                                \
            const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(bb + bo_ + 32 * B_STRIDE);                          \
            const bf16x8 bm1 = *reinterpret_cast<const bf16x8*>(bb + B_PLANE + bo_ + 32 * B_STRIDE);                \
            /* small terms first, hi*hi last; the two accumulators alternate so dependent MFMAs are 2 apart */      \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh0, acc[0], 0, 0, 0);                             \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh1, acc[1], 0, 0, 0);                             \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm0, acc[0], 0, 0, 0);                             \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm1, acc[1], 0, 0, 0);                             \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh0, acc[0], 0, 0, 0);                             \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh1, acc[1], 0, 0, 0);                             \
        }                                                                                                           \
        if (ksg_ == n_kstep - 1) epilogue(pass_);                                                                   \
    }

    for (int t0 = 0; t0 < total; t0 += 4) {
        HGT_STEP(0, t0)
        HGT_STEP(1, t0 + 1)
        HGT_STEP(2, t0 + 2)
        HGT_STEP(3, t0 + 3)
    }
#undef HGT_STEP
#undef HGT_LOAD_STAGE
}

static inline void split_dims(int k, int n_out, int* n_pass, int* n_kstep) {
    *n_pass = (n_out + BNP - 1) / BNP;
    *n_kstep = (k + BK - 1) / BK;
}

}  // namespace

extern "C" int hgt_split_weights_bytes(int32_t n_groups, int32_t k, int32_t n_out, uint64_t* out) {
    if (!out || n_groups <= 0 || k <= 0 || n_out <= 0) return HGT_ERR_INVALID_ARG;
    int n_pass, n_kstep;
    split_dims(k, n_out, &n_pass, &n_kstep);
    *out = (uint64_t)n_groups * n_pass * n_kstep * 2 * W_TILE_ELEMS * 2;
    return HGT_OK;
}

extern "C" int hgt_split_weights(const float* W, int64_t w_group_stride, int32_t n_groups, int32_t k, int32_t n_out, void* w_split,
                                 void* stream) {
    if (!W || !w_split || n_groups <= 0 || k <= 0 || n_out <= 0) return HGT_ERR_INVALID_ARG;
    int n_pass, n_kstep;
    split_dims(k, n_out, &n_pass, &n_kstep);
    const int64_t total = (int64_t)n_groups * n_pass * n_kstep * W_TILE_ELEMS;
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
    if (n_rows == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;   // device-side group sizes: launch the upper bound
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const int vec_ok = (ldx % 4 == 0) && (k % 4 == 0) && (((uintptr_t)x & 15) == 0);
    if (prologue == 0)
        k_typed_linear_split<0><<<(unsigned)row_tiles, 512, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                         (const unsigned short*)w_split, bias, b_group_stride, out0,
                                                                         out1, out2, block_cols, out_by_position, vec_ok);
    else
        k_typed_linear_split<1><<<(unsigned)row_tiles, 512, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                         (const unsigned short*)w_split, bias, b_group_stride, out0,
                                                                         out1, out2, block_cols, out_by_position, vec_ok);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
