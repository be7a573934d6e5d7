// Typed (grouped) linear layer, split-bf16 x3 variant (precision = 1 of hgt_typed_linear).
//
// fp32 inputs are split on the fly into two bf16 terms  a = a_hi + a_mid  (a_hi = bf16(a),
// a_mid = bf16(a - a_hi)); the product is evaluated as  a_hi*b_hi + a_hi*b_mid + a_mid*b_hi  on the
// bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate): 3 MFMAs at 16x the fp32 MFMA rate,
// relative error of a product <= ~3*2^-18 (the dropped a_mid*b_mid and third-split terms).  gfx950
// has no xf32/TF32 MFMA, so this is the only reduced-cost route for fp32 operands; it is opt-in
// (HGTConv(precision="bf16x3")) and parity-tested at the same 1e-4 bound as the exact fp32 path.
//
// Same 128x128 tile / 4 waves (2x2) / 64x64 per wave decomposition as hgt_gemm.hip; BK = 32.
// The split happens ONCE per element while staging (global fp32 -> registers -> hi/mid bf16 planes
// in LDS), not per wave.  LDS rows are 32 bf16 = 64 B, padded to an 80 B stride: for a
// ds_read_b128 the 16 lanes of a group (16 different rows, same k offset) land on 16 distinct
// 4-bank slots (20*row mod 64 words), i.e. conflict free.
#include "hgt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROW_B = 80;   // LDS row stride in bytes (64 B of data + 16 B pad)

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

template <int PROLOGUE>
__global__ __launch_bounds__(256) void k_typed_linear_bf16x3(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
    int n_groups, int k, int n_out, const float* __restrict__ W, int64_t wgs, const float* __restrict__ bias, int64_t bgs,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int vec_ok) {
    // [A_hi | A_mid | B_hi | B_mid], each 128 rows x 80 B
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BM * ROW_B];
    __shared__ int s_rid[BM];
    unsigned char* const sAh = smem;
    unsigned char* const sAm = smem + BM * ROW_B;
    unsigned char* const sBh = smem + 2 * BM * ROW_B;
    unsigned char* const sBm = smem + 3 * BM * ROW_B;

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
    const int col0 = blockIdx.y * BN;
    const float* __restrict__ Wg = W + (int64_t)g * wgs;

    const int tid = threadIdx.x;
    if (tid < BM) s_rid[tid] = (tid < nrows) ? rows[row0 + tid] : -1;
    __syncthreads();

    int a_rid[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_rid[j] = s_rid[(tid + 256 * j) >> 3];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fkb = (lane >> 5) * 16;   // byte offset of this lane's 8 bf16 inside a 16-wide k chunk

    auto load_tile = [&](int k0, float4 (&av)[4], float4 (&bv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            const int r = f >> 3, kk = k0 + (f & 7) * 4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
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
            const int n = col0 + r;
            if (n < n_out && kk < k) {
                const float* pw = Wg + (int64_t)n * k + kk;
                if (vec_ok && kk + 3 < k) {
                    b = *reinterpret_cast<const float4*>(pw);
                } else {
                    b.x = pw[0];
                    if (kk + 1 < k) b.y = pw[1];
                    if (kk + 2 < k) b.z = pw[2];
                    if (kk + 3 < k) b.w = pw[3];
                }
            }
            av[j] = a;
            bv[j] = b;
        }
    };

    float4 av[4], bv[4];
    load_tile(0, av, bv);
    for (int k0 = 0; k0 < k; k0 += BK) {
        __syncthreads();   // previous tile's fragment reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            const int r = f >> 3, cb = (f & 7) * 8;   // 4 bf16 = 8 bytes
            uint2 hi, mid;
            split4(av[j], hi, mid);
            *reinterpret_cast<uint2*>(sAh + r * ROW_B + cb) = hi;
            *reinterpret_cast<uint2*>(sAm + r * ROW_B + cb) = mid;
            split4(bv[j], hi, mid);
            *reinterpret_cast<uint2*>(sBh + r * ROW_B + cb) = hi;
            *reinterpret_cast<uint2*>(sBm + r * ROW_B + cb) = mid;
        }
        __syncthreads();
        if (k0 + BK < k) load_tile(k0 + BK, av, bv);   // next tile's global loads fly under this tile's MFMAs
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
            bf16x8 ah[2], am[2], bh[2], bm[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ao = (wm * 64 + i * 32 + frow) * ROW_B + kc * 32 + fkb;
                const int bo = (wn * 64 + i * 32 + frow) * ROW_B + kc * 32 + fkb;
                ah[i] = *reinterpret_cast<const bf16x8*>(sAh + ao);
                am[i] = *reinterpret_cast<const bf16x8*>(sAm + ao);
                bh[i] = *reinterpret_cast<const bf16x8*>(sBh + bo);
                bm[i] = *reinterpret_cast<const bf16x8*>(sBm + bo);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // small terms first, the dominant hi*hi term last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= n_out) continue;
        const float bcol = bias ? bias[(int64_t)g * bgs + col] : 0.0f;
        const int blk = col / block_cols, cc = col - blk * block_cols;
        float* __restrict__ ob = (blk == 0) ? out0 : ((blk == 1) ? out1 : out2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rt = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (rt < nrows) {
                    const int64_t orow = by_pos ? (int64_t)(row0 + rt) : (int64_t)s_rid[rt];
                    ob[orow * block_cols + cc] = acc[i][j][r] + bcol;
                }
            }
        }
    }
}

}  // namespace

// called by hgt_typed_linear (hgt_gemm.hip) for precision == 1
int hgt_typed_linear_bf16x3_launch(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                   int64_t n_rows, int32_t k, int32_t n_out, const float* W, int64_t wgs, const float* bias,
                                   int64_t bgs, float* out0, float* out1, float* out2, int32_t block_cols, int32_t by_pos,
                                   int32_t prologue, int vec_ok, hipStream_t stream) {
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    dim3 grid((unsigned)row_tiles, (unsigned)((n_out + BN - 1) / BN));
    if (prologue == 0)
        k_typed_linear_bf16x3<0><<<grid, 256, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out, W, wgs, bias, bgs, out0, out1,
                                                           out2, block_cols, by_pos, vec_ok);
    else
        k_typed_linear_bf16x3<1><<<grid, 256, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out, W, wgs, bias, bgs, out0, out1,
                                                           out2, block_cols, by_pos, vec_ok);
    if (hipGetLastError() != hipSuccess) return HGT_ERR_LAUNCH;
    return HGT_OK;
}
