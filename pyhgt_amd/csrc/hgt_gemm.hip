// Typed (grouped) linear layer on the gfx950 matrix cores.
//
//   y[n, :] = prologue(x[n, :]) @ W[type(n)]^T + b[type(n)]      for every row n in a typed row list
//
// Replaces the reference's per-meta-relation, per-EDGE nn.Linear calls (conv.py:96-97,103: 6*E*d^2
// flop) by one launch over NODES (6*N*d^2), the per-type a_linear of conv.py:125, and the Linear of
// RelTemporalEncoding (conv.py:297-299) when the temporal tables are built.
//
// Tiling (designed for 64-wide wavefronts / MFMA, not a warp tiling): 128x128 output tile per
// 256-thread workgroup, BK = 32, 4 waves as 2x2, each wave 64x64 = 2x2 tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain -> fp32 parity with the reference).  A rows are
// GATHERED through the typed row list while staging into LDS, so node types need not be
// contiguous; LDS row stride 36 floats makes the ds_read_b128 fragment reads conflict-free
// ((36*row) mod 64 is distinct for 16 rows that differ mod 16).  Each lane fetches 4 consecutive
// k's with one ds_read_b128 and feeds them to 4 successive MFMAs: lanes 0-31 take k = 8q..8q+3,
// lanes 32-63 take k = 8q+4..8q+7 for both operands, which is a legal re-association of the K sum.
#include "hgt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int PROLOGUE>
__global__ __launch_bounds__(256) void k_typed_linear_f32(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
    int n_groups, int k, int n_out, const float* __restrict__ W, int64_t wgs, const float* __restrict__ bias, int64_t bgs,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int vec_ok,
    int n_col_tiles) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];
    __shared__ int s_rid[BM];

    // which (group, row tile) is this workgroup?  group sizes live on the device (no host sync)
    // XCD-aware 1-D grid: hardware places block b on XCD b % 8.  Logical tile id l = (b % 8) * per + b / 8 gives
    // every XCD a contiguous run of logical tiles, and the column tiles of one row tile are consecutive l,
    // so the x rows of a row tile are fetched from HBM once and re-read from that XCD's L2 by its other
    // column tiles (without this each column tile streamed x from HBM again: 6x for the Q|K|V launch).
    const int per_xcd = gridDim.x >> 3;
    const int ltile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int slot = ltile / n_col_tiles;
    const int col_tile = ltile - slot * n_col_tiles;
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
    const int col0 = col_tile * BN;
    const float* __restrict__ Wg = W + (int64_t)g * wgs;

    const int tid = threadIdx.x;
    if (tid < BM) s_rid[tid] = (tid < nrows) ? rows[row0 + tid] : -1;
    __syncthreads();

    // staging assignment: float4 index f = tid + 256*j -> (row r = f>>3, k-quad c4 = f&7)
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
    const int frow = lane & 31, fk = (lane >> 5) * 4;

    for (int k0 = 0; k0 < k; k0 += BK) {
        float4 av[4], bv[4];
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
                if (PROLOGUE == 1) { a.x = gelu_erf(a.x); a.y = gelu_erf(a.y); a.z = gelu_erf(a.z); a.w = gelu_erf(a.w); }
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
        __syncthreads();   // previous tile's fragment reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            const int r = f >> 3, c = (f & 7) * 4;
            *reinterpret_cast<float4*>(&As[r * LDS_LD + c]) = av[j];
            *reinterpret_cast<float4*>(&Bs[r * LDS_LD + c]) = bv[j];
        }
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < BK / 8; ++kq) {
            float4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + frow) * LDS_LD + kq * 8 + fk]);
                bf[i] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + frow) * LDS_LD + kq * 8 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
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

extern "C" int hgt_typed_linear(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off,
                                int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_out,
                                const float* W, int64_t w_group_stride, const float* bias, int64_t b_group_stride,
                                float* out0, float* out1, float* out2, int32_t block_cols,
                                int32_t out_by_position, int32_t prologue, int32_t precision, void* stream_) {
    if (!x || !rows || !group_off || !W || !out0 || n_groups <= 0 || n_rows < 0 || k <= 0 || n_out <= 0 || block_cols <= 0)
        return HGT_ERR_INVALID_ARG;
    const int n_blocks_out = (n_out + block_cols - 1) / block_cols;
    if (n_blocks_out > 3 || (n_blocks_out > 1 && !out1) || (n_blocks_out > 2 && !out2)) return HGT_ERR_INVALID_ARG;
    if (prologue != 0 && prologue != 1) return HGT_ERR_INVALID_ARG;
    if (precision != 0) return HGT_ERR_INVALID_ARG;   // split-bf16: hgt_split_weights + hgt_typed_linear_bf16x3
    if (n_rows == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    // group sizes are device data: launch the upper bound on row tiles, surplus workgroups exit
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const int n_col_tiles = (n_out + BN - 1) / BN;
    const int64_t total_tiles = row_tiles * n_col_tiles;
    if (total_tiles > 0x7ffffff0LL) return HGT_ERR_TOO_LARGE;
    const unsigned grid = (unsigned)((total_tiles + 7) / 8 * 8);   // multiple of 8 for the XCD remap; surplus blocks exit
    const int vec_ok = (ldx % 4 == 0) && (k % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)W & 15) == 0) &&
                       (w_group_stride % 4 == 0);
    if (prologue == 0)
        k_typed_linear_f32<0><<<grid, 256, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out, W, w_group_stride, bias,
                                                        b_group_stride, out0, out1, out2, block_cols, out_by_position, vec_ok, n_col_tiles);
    else
        k_typed_linear_f32<1><<<grid, 256, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out, W, w_group_stride, bias,
                                                        b_group_stride, out0, out1, out2, block_cols, out_by_position, vec_ok, n_col_tiles);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
