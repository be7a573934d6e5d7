// Typed (grouped) linear layer for the LATENCY regime: sampled batches of a few thousand rows (BASELINE.json configs[2] / [4],
// the only workload the reference's scripts run: pyHGT/model.py:77-79, train_ogbn_mag.py:44-46, train_paper_field.py:35-41).
//
//   y[n, :] = x[n, :] @ W[type(n)]^T + b[type(n)]          (conv.py:96-97,103: Q | K | V;  conv.py:119-133: a_linear + update)
//
// The persistent kernels of hgt_gemm_bf16x3.hip are built for millions of rows: one workgroup per CU, a producer / consumer split
// that overlaps tile i + 1's rows with tile i's MFMAs.  On a 3 200-row batch a workgroup sees ONE tile and nothing overlaps: 14 us
// for Q | K | V and 19 us for the update at c3 (profiles/r06_latency_base_c3.txt) against 1 - 3 us of matrix-core work.  Here the
// same arithmetic (same split, same k order, same three products per k-chunk: results are bit-identical to the slab kernels) is
// laid out for many small workgroups and a short dependent chain:
//   * a workgroup owns one (row tile, column tile) pair: 32 x 128 outputs for the plain linear (a 3 200 x 768 problem is 624
//     workgroups of four wavefronts, up to four per CU), 32 whole rows for the update form (8 wavefronts x 32 or 64 columns);
//   * the tile's x rows are requested WHOLE (all K panels, K <= 256 / 512) together with a wavefront's B fragments of the first two
//     panels before anything is waited for; they go registers -> (split hi / mid) -> an LDS slab that holds every panel: ONE
//     workgroup barrier, then a k loop of ds_read + MFMA with the later panels' fragments requested one panel of MFMAs ahead;
//   * the update form (a_linear + gated skip + LayerNorm, conv.py:129-133) requests its skip rows while the MFMAs run.
// What the time of such a kernel is made of (r6 eliminations at c3, 64 x 128 tiles, 13.9 us warm: no stores 10.8, no fragment loads
// 11.4, no row loads 12.8, none of the three 8.6): dispatch + the id chain + one round of MFMAs + a store burst that a CU retires
// at ~10 B / clk -- every phase short, none overlapped with another inside ONE round of workgroups.  Smaller tiles (more, lighter
// workgroups per CU) were worth 2.7 us; the sampled-batch layer's update now runs inside the merge pass instead
// (hgt_edge_agg_items.hip: k_merge_update), this file's update form serves the two-call path.
// fp16 split (precision "f16x3"): the power-of-two row scale needs the maximum of the WHOLE row before its first panel is split --
// the row is in the registers of 16 or 32 lanes of one wavefront by then: four or five lane exchanges, no second pass over x.
#include "hgt_common.h"
#include "hgt_split_common.h"
#include "hgt_wt_store.h"
#include <algorithm>

#ifndef HGT_TILE_MAX_ROWS
#define HGT_TILE_MAX_ROWS 16384      // typed linears below this many rows take the tile kernels (measured against the persistent kernel)
#endif

namespace {

struct TileArgs {
    const float* x; int64_t ldx; const int32_t* rows; const int32_t* group_off; int n_groups; int k; int n_out;
    const unsigned short* wsplit; const float* bias; int64_t bgs; float* out0; float* out1; float* out2; int block_cols; int by_pos;
    int vec_ok; int prologue;
    // update form
    const float* xs; int64_t ldxs; const float* skip; const float* lnw; const float* lnb; int use_norm;
};

typedef float f32x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tile_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NPM = K panels the LDS slab holds (K <= NPM * KPAN: the launcher's domain check)
template <int NW, int RT, int CT, int KPAN, int NPM, bool UPD, bool F16>
__global__ __launch_bounds__(64 * NW, (KPAN == 64 && RT == 1 && CT == 1) ? 4 : 2) void k_tile_linear(const TileArgs a) {
    constexpr int BMT = 32 * RT, KCP = KPAN / 16, LPR = KPAN / 4, RPI = 64 / LPR, NL = BMT / (RPI * NW);
    constexpr int ASTR = KPAN * 2 + 16, APLANE = BMT * ASTR;
    static_assert(NL >= 1 && NL * RPI * NW == BMT, "row tile / wavefront geometry");
    __shared__ __attribute__((aligned(16))) unsigned char sA[NPM][2 * APLANE];      // the tile's rows, all K panels: [panel][plane][row][k]
    __shared__ int s_rid[BMT];
    __shared__ float s_rinv[F16 ? BMT : 1];
    __shared__ __attribute__((aligned(16))) float s_red[UPD ? 2 * BMT * NW : 1];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which (group, rows) is this row tile?  (device-side group sizes: the launcher starts an upper bound of row tiles)
    int g = 0, row0 = 0, nrows = 0;
    {
        int before = 0, t = (int)blockIdx.x;
        bool found = false;
        for (g = 0; g < a.n_groups; ++g) {
            const int gb = a.group_off[g], ge = a.group_off[g + 1];
            const int nt = (ge - gb + BMT - 1) / BMT;
            if (t < before + nt) {
                row0 = gb + (t - before) * BMT;
                nrows = min(BMT, ge - row0);
                found = true;
                break;
            }
            before += nt;
        }
        if (!found) return;
    }
    const int k = a.k, n_out = a.n_out;
    const int n_pass = (n_out + BNP - 1) / BNP;
    const int n_kc = ((k + KC - 1) / KC + 3) & ~3;
    const int total = n_pass * n_kc;
    const int n_pan = (n_kc + KCP - 1) / KCP;       // <= NPM

    // ---- this wavefront's column blocks (32 columns each) and their fragment streams
    int cb[CT];
    bool live[CT];
    const unsigned short* wp[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        cb[c] = ((int)blockIdx.y * NW + wave) * CT + c;
        live[c] = (cb[c] >> 3) < n_pass;
        const int pass = live[c] ? (cb[c] >> 3) : 0;
        wp[c] = a.wsplit + ((int64_t)g * total + (int64_t)pass * n_kc) * 2 * W_PLANE_ELEMS + ((cb[c] & 7) * 64 + lane) * 8;
    }
    float winv = 1.0f;
    if constexpr (F16) winv = reinterpret_cast<const float*>(a.wsplit + (int64_t)a.n_groups * total * 2 * W_PLANE_ELEMS)[g];

    // ---- row ids: every lane keeps the ids of the NL rows it loads; the table in LDS serves the epilogue
    const int lrow = lane / LPR, lk = (lane % LPR) * 4;
    int myrid[NL];
    const int rid_safe = a.rows[row0];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int r = (j * NW + wave) * RPI + lrow;
        myrid[j] = (r < nrows) ? a.rows[row0 + r] : -1;
    }
    if (tid < BMT) s_rid[tid] = (tid < nrows) ? a.rows[row0 + tid] : -1;

    bf16x8 wh[2][CT][KCP], wm[2][CT][KCP];
#define TILE_LOAD_W(B, P)                                                                                  \
    _Pragma("unroll") for (int c = 0; c < CT; ++c) {                                                       \
        _Pragma("unroll") for (int kc = 0; kc < KCP; ++kc) {                                               \
            const int kidx = min((P) * KCP + kc, n_kc - 1);                                                \
            const unsigned short* t_ = wp[c] + (int64_t)kidx * 2 * W_PLANE_ELEMS;                          \
            wh[B][c][kc] = *reinterpret_cast<const bf16x8*>(t_);                                           \
            wm[B][c][kc] = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                           \
        }                                                                                                  \
    }
    // ---- everything is requested before anything is waited for: the fragments of the first two panels, then ALL panels of the
    //      tile's rows (a lane: NL rows x NPM panels x 16 bytes; always an in-bounds address -- absent rows re-read the tile's
    //      first row, columns past k the row's last ones -- and zeroed below where they must read as zero)
    TILE_LOAD_W(0, 0)
    float4 areg[NPM][NL];
#pragma unroll
    for (int P = 0; P < NPM; ++P) {
        if (P < n_pan) {
            const int kk = P * KPAN + lk;
            if (a.vec_ok) {
                const int kc_ = min(kk, k - 4);
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const int rid = myrid[j] < 0 ? rid_safe : myrid[j];
                    areg[P][j] = *reinterpret_cast<const float4*>(a.x + (int64_t)rid * a.ldx + kc_);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const int rid = myrid[j] < 0 ? rid_safe : myrid[j];
                    const float* px = a.x + (int64_t)rid * a.ldx;
                    areg[P][j] = make_float4(px[min(kk, k - 1)], px[min(kk + 1, k - 1)], px[min(kk + 2, k - 1)], px[min(kk + 3, k - 1)]);
                }
            }
        }
    }
    if (n_pan > 1) TILE_LOAD_W(1, 1)
    // ---- zero what lies outside the matrix, fp16 split: the rows' power-of-two scales (the whole row is in this wavefront's registers)
    unsigned mb[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) mb[j] = 0u;
#pragma unroll
    for (int P = 0; P < NPM; ++P) {
        if (P < n_pan) {
            const int kk = P * KPAN + lk;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const bool row_ok = myrid[j] >= 0;
                float4 v = areg[P][j];
                v.x = (row_ok && kk < k) ? v.x : 0.f;
                v.y = (row_ok && kk + 1 < k) ? v.y : 0.f;
                v.z = (row_ok && kk + 2 < k) ? v.z : 0.f;
                v.w = (row_ok && kk + 3 < k) ? v.w : 0.f;
                areg[P][j] = v;
                if constexpr (F16) mb[j] = max(mb[j], abs_bits4(v));
            }
        }
    }
    float rscale[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        rscale[j] = 1.0f;
        if constexpr (F16) {
            mb[j] = row16_max_bits(mb[j]);      // (a row's panel slice = LPR = 16 or 32 lanes: one or two DPP rows)
            if constexpr (LPR == 32) mb[j] = max(mb[j], (unsigned)__shfl_xor((int)mb[j], 16));
            float inv;
            f16_row_scale(mb[j], rscale[j], inv);
            if ((lane % LPR) == 0) s_rinv[(j * NW + wave) * RPI + lrow] = inv;
        }
    }
#pragma unroll
    for (int P = 0; P < NPM; ++P) {
        if (P < n_pan) {
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int r = (j * NW + wave) * RPI + lrow;
                uint2 hi, mid;
                split4_t<F16>(areg[P][j], rscale[j], hi, mid);
                unsigned char* p_ = sA[P] + r * ASTR + lk * 2;
                *reinterpret_cast<uint2*>(p_) = hi;
                *reinterpret_cast<uint2*>(p_ + APLANE) = mid;
            }
        }
    }

    f32x16 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.0f;

    const int frow = lane & 31, khalf = lane >> 5;
    // One panel: KCP k-chunks x (RT x CT) x 3 MFMAs; the A fragments of k-chunk kc + 1 are read while kc's MFMAs run.  No branch
    // inside: the k-chunks past n_kc (K not a multiple of the panel) multiply zero columns of the slab by a finite fragment.
#define TILE_LOAD_AF(SET, KCI)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                                    \
        fah[SET][r] = *reinterpret_cast<const bf16x8*>(sl_ + r * 32 * ASTR + (KCI) * 32);                               \
        fam[SET][r] = *reinterpret_cast<const bf16x8*>(sl_ + r * 32 * ASTR + (KCI) * 32 + APLANE);                      \
    }
#define TILE_COMPUTE(B, P)                                                                                              \
    {                                                                                                                   \
        const unsigned char* sl_ = sA[P] + frow * ASTR + khalf * 16;                                                    \
        bf16x8 fah[2][RT], fam[2][RT];                                                                                  \
        TILE_LOAD_AF(0, 0)                                                                                              \
        _Pragma("unroll") for (int kc = 0; kc < KCP; ++kc) {                                                            \
            if (kc + 1 < KCP) { TILE_LOAD_AF((kc + 1) & 1, kc + 1) }                                                    \
            _Pragma("unroll") for (int c = 0; c < CT; ++c) {                                                            \
                _Pragma("unroll") for (int r = 0; r < RT; ++r) acc[r][c] = mfma32_t<F16>(fam[kc & 1][r], wh[B][c][kc], acc[r][c]); \
                _Pragma("unroll") for (int r = 0; r < RT; ++r) acc[r][c] = mfma32_t<F16>(fah[kc & 1][r], wm[B][c][kc], acc[r][c]); \
                _Pragma("unroll") for (int r = 0; r < RT; ++r) acc[r][c] = mfma32_t<F16>(fah[kc & 1][r], wh[B][c][kc], acc[r][c]); \
            }                                                                                                           \
        }                                                                                                               \
    }

    tile_barrier();        // the ONE barrier of the k loop: the slab is complete
    // ---- the skip rows of the update form: requested here, used after the last panel
    const int col_l = ((lane & 31) >> 2) * 4;           // this lane's 4 consecutive columns inside a column block (after the quad transpose)
    const int rt0 = (lane & 3) + 4 * (lane >> 5);       // ... and its rows: rt0 + 8 q (+ 32 per row tile)
    // (two column blocks per wavefront: 32 more registers than the file has next to both fragment buffers -- requested after the
    //  last panel instead, one exposed L2 round trip)
    float4 xv[UPD ? RT : 1][UPD ? CT : 1][4];
#define TILE_LOAD_SKIP()                                                                                                \
    _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                                      \
        _Pragma("unroll") for (int c = 0; c < CT; ++c)                                                                  \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                             \
                const int orow = s_rid[r * 32 + rt0 + 8 * q];                                                           \
                const int col = cb[c] * 32 + col_l;                                                                     \
                xv[r][c][q] = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
                if (orow >= 0 && col < n_out) xv[r][c][q] = *reinterpret_cast<const float4*>(a.xs + (int64_t)orow * a.ldxs + col); \
            }
    if constexpr (UPD && CT == 1) { TILE_LOAD_SKIP() }

    for (int P = 0; P < n_pan; P += 2) {
        TILE_COMPUTE(0, P)
        if (P + 2 < n_pan) TILE_LOAD_W(0, P + 2)
        if (P + 1 < n_pan) {
            TILE_COMPUTE(1, P + 1)
            if (P + 3 < n_pan) TILE_LOAD_W(1, P + 3)
        }
    }
#undef TILE_COMPUTE
#undef TILE_LOAD_AF
#undef TILE_LOAD_W
    if constexpr (UPD && CT > 1) { TILE_LOAD_SKIP() }
#undef TILE_LOAD_SKIP

    // ---- epilogue
    const bool o1 = lane & 1, o2 = lane & 2;
    if constexpr (!UPD) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = cb[c] * 32 + col_l;
            if (!live[c] || col >= n_out) continue;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + (int64_t)g * a.bgs + col);
            const int blk = col / a.block_cols, cc = col - blk * a.block_cols;
            float* __restrict__ ob = (blk == 0) ? a.out0 : ((blk == 1) ? a.out1 : a.out2);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[r][c][4 * q], v1 = acc[r][c][4 * q + 1], v2 = acc[r][c][4 * q + 2], v3 = acc[r][c][4 * q + 3];
                    quad_transpose(v0, v1, v2, v3, o1, o2);
                    const int rt = r * 32 + rt0 + 8 * q;
                    if (rt < nrows) {
                        const float sc = F16 ? s_rinv[F16 ? rt : 0] * winv : 1.0f;
                        const int64_t orow = a.by_pos ? (int64_t)(row0 + rt) : (int64_t)s_rid[rt];
                        float4 o4 = make_float4(v0 * sc + b4.x, v1 * sc + b4.y, v2 * sc + b4.z, v3 * sc + b4.w);
                        if (a.prologue) o4 = make_float4(hgt_tanh(o4.x), hgt_tanh(o4.y), hgt_tanh(o4.z), hgt_tanh(o4.w));
                        store_wt16(ob + orow * a.block_cols + cc, o4.x, o4.y, o4.z, o4.w);      // (read by the next kernel only)
                    }
                }
        }
    } else {
        // y = (acc + b) * sigmoid(skip[t]) + x * (1 - sigmoid(skip[t]));  out = LayerNorm_t(y), two-pass mean / variance (conv.py:129-133).
        // A row's columns live in NW wavefronts x CT column blocks x 8 lanes: lane-strided sums, then one LDS table entry per (row, wave).
        const float alpha = 1.0f / (1.0f + expf(-a.skip[g]));
        const float inv_n = 1.0f / (float)n_out;
        float y[RT][CT][4][4];
        float* s_sum = s_red;
        float* s_var = s_red + BMT * NW;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = cb[c] * 32 + col_l;
            const bool col_ok = live[c] && col < n_out;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok && a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + (int64_t)g * a.bgs + col);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[r][c][4 * q], v1 = acc[r][c][4 * q + 1], v2 = acc[r][c][4 * q + 2], v3 = acc[r][c][4 * q + 3];
                    quad_transpose(v0, v1, v2, v3, o1, o2);
                    const float sc = F16 ? s_rinv[F16 ? (r * 32 + rt0 + 8 * q) : 0] * winv : 1.0f;
                    const float4 x4 = xv[r][c][q];
                    y[r][c][q][0] = col_ok ? (v0 * sc + b4.x) * alpha + x4.x * (1.0f - alpha) : 0.0f;
                    y[r][c][q][1] = col_ok ? (v1 * sc + b4.y) * alpha + x4.y * (1.0f - alpha) : 0.0f;
                    y[r][c][q][2] = col_ok ? (v2 * sc + b4.z) * alpha + x4.z * (1.0f - alpha) : 0.0f;
                    y[r][c][q][3] = col_ok ? (v3 * sc + b4.w) * alpha + x4.w * (1.0f - alpha) : 0.0f;
                }
        }
        if (a.use_norm) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float ps = 0.0f;
#pragma unroll
                    for (int c = 0; c < CT; ++c) ps += y[r][c][q][0] + y[r][c][q][1] + y[r][c][q][2] + y[r][c][q][3];
                    ps = strided8_sum(ps);
                    if (((lane & 31) >> 2) == 0) s_sum[(r * 32 + rt0 + 8 * q) * NW + wave] = ps;
                }
            tile_barrier();
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rt = r * 32 + rt0 + 8 * q;
                    float s = 0.0f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) s += s_sum[rt * NW + w];
                    const float mean = s * inv_n;
                    float ps = 0.0f;
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const bool col_ok = live[c] && (cb[c] * 32 + col_l) < n_out;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            y[r][c][q][i] -= mean;      // y is centred from here on
                            const float d = col_ok ? y[r][c][q][i] : 0.0f;
                            ps = fmaf(d, d, ps);
                        }
                    }
                    ps = strided8_sum(ps);
                    if (((lane & 31) >> 2) == 0) s_var[rt * NW + wave] = ps;
                }
            tile_barrier();
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = cb[c] * 32 + col_l;
            if (!live[c] || col >= n_out) continue;
            float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.use_norm) {
                w4 = *reinterpret_cast<const float4*>(a.lnw + (int64_t)g * n_out + col);
                c4 = *reinterpret_cast<const float4*>(a.lnb + (int64_t)g * n_out + col);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rt = r * 32 + rt0 + 8 * q;
                    if (rt >= nrows) continue;
                    float rstd = 1.0f;
                    if (a.use_norm) {
                        float s = 0.0f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) s += s_var[rt * NW + w];
                        rstd = rsqrtf(s * inv_n + 1e-5f);
                    }
                    const int64_t orow = (int64_t)s_rid[rt];
                    *reinterpret_cast<float4*>(a.out0 + orow * n_out + col) =
                        make_float4(y[r][c][q][0] * rstd * w4.x + c4.x, y[r][c][q][1] * rstd * w4.y + c4.y, y[r][c][q][2] * rstd * w4.z + c4.z,
                                    y[r][c][q][3] * rstd * w4.w + c4.w);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K > 256 (n_hid 400 / 512, the 1169-column OAG adapter) on a few thousand rows: tiles of (32 RT) x 256 outputs, eight wavefronts,
// the x rows STREAMED through a double-buffered LDS panel of 64 columns (the whole-row slab above would be 130 - 330 KB).
// Why its own shape (r6): at 3 200 - 4 096 rows x 1 536 columns the 64 x 256 tiles of the slab kernel are 324 - 414 workgroups on 256
// CUs -- some CUs get two, i.e. the kernel takes two tiles' time -- and every workgroup pulls 640 KB through its CU's L1 four
// k-chunks at a time: 30 us against ~6 us of matrix-core work.  Here RT is chosen by the launcher so that the tiles fit the chip in ONE
// round where they can (3 200 rows: RT = 3 -> 216 tiles), a wavefront's B fragments of a whole panel (4 k-chunks) are requested
// together one panel of MFMAs (4 x RT x 3) ahead, and the next panel's rows are in flight during the current panel's MFMAs: one
// barrier per panel.  Same split, k order and product order as the slab kernels: bit-identical in the bf16 split.
// fp16 split: RUNNING row scales like k_typed_linear_split (a panel is split with the scale its row has so far, lowered first when
// the panel's own maximum needs it; accumulators of lowered rows are multiplied by the exact power-of-two ratio) -- no pass over x
// for the row maxima.
// ---------------------------------------------------------------------------------------------------------------------------------
// 32-row tiles (RT = 1) with a two-slot fragment ring: 124 registers -> TWO workgroups per CU, so up to 512 tiles are resident at once.
// The OAG adapter (4 096 x 1 169 -> 400: 262 tiles at RT = 1) took 49 us (f16x3) / 45 (bf16x3) as 134 tiles of 64 rows on 134 CUs;
// 31 us this way.  128-row tiles (RT = 4, two slots as well: 244 registers) where nothing smaller fits one round (c5 Q|K|V,
// 4 096 x 400 -> 1 536: 32 us against 38 on the slab kernel; RT = 3 at 270 tiles -- a second round of 14 -- 50 us).
#ifndef HGT_TS_R1_SMALL
#define HGT_TS_R1_SMALL 1
#endif
#ifndef HGT_TS_R1_TILES
#define HGT_TS_R1_TILES 512
#endif
#define HGT_TS_OCC(RT) ((HGT_TS_R1_SMALL && (RT) == 1) ? 4 : 2)
#define HGT_TS_WR(RT) (((RT) >= 4 || (HGT_TS_R1_SMALL && (RT) == 1)) ? 2 : 3)
template <int RT, bool F16>
__global__ __launch_bounds__(512, HGT_TS_OCC(RT)) void k_tile_linear_stream(const TileArgs a) {
    constexpr int NW = 8, KPAN = 64, BMT = 32 * RT, KCP = 4, LPR = 16, RPI = 4, NL = RT;      // a lane: RT rows x 16 bytes per panel
    constexpr int ASTR = KPAN * 2 + 16, APLANE = BMT * ASTR;
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][2 * APLANE];
    __shared__ int s_rid[BMT];
    __shared__ float s_rscale[F16 ? BMT : 1], s_rinv[F16 ? BMT : 1];       // (a row's scale is read and written by ONE wavefront)
    __shared__ __attribute__((aligned(16))) float s_ratio[2][F16 ? BMT : 1];      // by panel parity
    __shared__ int s_flag[2];                                                      // = P + 1 when panel P lowered a row's scale (no clearing)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- XCD-aware work order.  Workgroup b runs on XCD b % 8 and every XCD has its own 4 MB L2: with the plain (row tile, pass) order
    // every XCD walks ALL (type, pass) slabs of the fragment image -- 12.6 MB at n_hid = 512 -- and 74 % of its L2 requests miss
    // (r6 counters: TCC_MISS 1.15 M of 1.55 M requests, 147 MB per launch through the fabric: that, not the matrix cores, was the
    // 30 us).  Here the work list is ordered (type, pass, row tile) and cut into eight contiguous chunks, one per XCD: an XCD touches
    // ~3 slabs of 512 KB and the rows of about one type.
    int g = 0, row0 = 0, nrows = 0, pass = 0;
    {
        const int n_passes = (a.n_out + BNP - 1) / BNP;
        int t_all = 0;
        for (int gg = 0; gg < a.n_groups; ++gg) t_all += (a.group_off[gg + 1] - a.group_off[gg] + BMT - 1) / BMT;
        const int total_work = t_all * n_passes, chunk = (total_work + 7) / 8;
        const int b = (int)blockIdx.x, xcd = b & 7, j = b >> 3;
        const int v = xcd * chunk + j;
        if (j >= chunk || v >= total_work) return;
        int before = 0;
        bool found = false;
        for (g = 0; g < a.n_groups; ++g) {
            const int gb = a.group_off[g], ge = a.group_off[g + 1];
            const int nt = (ge - gb + BMT - 1) / BMT;
            if (v < (before + nt) * n_passes) {
                const int local = v - before * n_passes;
                pass = local / nt;
                const int t = local - pass * nt;
                row0 = gb + t * BMT;
                nrows = min(BMT, ge - row0);
                found = true;
                break;
            }
            before += nt;
        }
        if (!found) return;
    }
    const int k = a.k, n_out = a.n_out;
    const int n_pass = (n_out + BNP - 1) / BNP;
    const int n_kc = ((k + KC - 1) / KC + 3) & ~3;
    const int total = n_pass * n_kc;
    const int n_pan = n_kc / KCP;                    // n_kc is a multiple of 4
    const unsigned short* wp = a.wsplit + ((int64_t)g * total + (int64_t)pass * n_kc) * 2 * W_PLANE_ELEMS + (wave * 64 + lane) * 8;
    float winv = 1.0f;
    if constexpr (F16) winv = reinterpret_cast<const float*>(a.wsplit + (int64_t)a.n_groups * total * 2 * W_PLANE_ELEMS)[g];

    const int lrow = lane / LPR, lk = (lane % LPR) * 4;
    int myrid[NL];
    const int rid_safe = a.rows[row0];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int r = (j * NW + wave) * RPI + lrow;
        myrid[j] = (r < nrows) ? a.rows[row0 + r] : -1;
    }
    if (tid < BMT) s_rid[tid] = (tid < nrows) ? a.rows[row0 + tid] : -1;
    if (tid < 2) s_flag[tid] = 0;

    // Rings: the B fragments of THREE panels in registers (a panel is requested two panels of MFMAs before its use), the x rows of
    // two panels in registers (requested two panels before they are split into the LDS panel that is free by then), two LDS panels.
    // One panel ahead (r6, first form) left every wavefront waiting ~1.5 us per panel for its own requests: 26 us for 6 us of MFMAs.
    constexpr int WR = HGT_TS_WR(RT);      // (128-row tiles: two ring slots -- the third would not fit the register file -- and twice the MFMAs per panel to cover a request)
    bf16x8 wh[WR][KCP], wm[WR][KCP];
    float4 areg[2][NL];
#define TS_LOAD_W(B, P)                                                                                    \
    _Pragma("unroll") for (int kc = 0; kc < KCP; ++kc) {                                                   \
        const unsigned short* t_ = wp + (int64_t)((P) * KCP + kc) * 2 * W_PLANE_ELEMS;                     \
        wh[B][kc] = *reinterpret_cast<const bf16x8*>(t_);                                                  \
        wm[B][kc] = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                                  \
    }
#define TS_LOAD_A(S, P)      /* (always an in-bounds address; zeroed in TS_COMMIT_A where it must read as zero) */      \
    {                                                                                                                   \
        const int kk_ = (P) * KPAN + lk;                                                                                \
        if (a.vec_ok) {                                                                                                 \
            const int kc_ = min(kk_, k - 4);                                                                            \
            _Pragma("unroll") for (int j = 0; j < NL; ++j) {                                                            \
                const int rid = myrid[j] < 0 ? rid_safe : myrid[j];                                                     \
                areg[S][j] = *reinterpret_cast<const float4*>(a.x + (int64_t)rid * a.ldx + kc_);                        \
            }                                                                                                           \
        } else {                                                                                                        \
            _Pragma("unroll") for (int j = 0; j < NL; ++j) {                                                            \
                const int rid = myrid[j] < 0 ? rid_safe : myrid[j];                                                     \
                const float* px = a.x + (int64_t)rid * a.ldx;                                                           \
                areg[S][j] = make_float4(px[min(kk_, k - 1)], px[min(kk_ + 1, k - 1)], px[min(kk_ + 2, k - 1)], px[min(kk_ + 3, k - 1)]); \
            }                                                                                                           \
        }                                                                                                               \
    }
#define TS_COMMIT_A(S, P, BUF)                                                                                          \
    {                                                                                                                   \
        const int kk_ = (P) * KPAN + lk;                                                                                \
        _Pragma("unroll") for (int j = 0; j < NL; ++j) {                                                                \
            const int r = (j * NW + wave) * RPI + lrow;                                                                 \
            const bool row_ok = myrid[j] >= 0;                                                                          \
            float4 v = areg[S][j];                                                                                      \
            v.x = (row_ok && kk_ < k) ? v.x : 0.f;                                                                      \
            v.y = (row_ok && kk_ + 1 < k) ? v.y : 0.f;                                                                  \
            v.z = (row_ok && kk_ + 2 < k) ? v.z : 0.f;                                                                  \
            v.w = (row_ok && kk_ + 3 < k) ? v.w : 0.f;                                                                  \
            float scale = 1.0f;                                                                                         \
            if constexpr (F16) {                                                                                        \
                const unsigned mb = row16_max_bits(abs_bits4(v));      /* (LPR = 16: the row's panel sits in one DPP row) */ \
                float inv;                                                                                              \
                f16_row_scale(mb, scale, inv);                                                                          \
                if ((P) > 0) {                                                                                          \
                    const float cur = s_rscale[r];                                                                      \
                    if (scale < cur) {      /* this panel is larger than everything the row had so far */              \
                        if ((lane % LPR) == 0) { s_ratio[(P) & 1][r] = scale / cur; s_flag[(P) & 1] = (P) + 1; }        \
                    } else {                                                                                            \
                        scale = cur;                                                                                    \
                        inv = s_rinv[r];                                                                                \
                        if ((lane % LPR) == 0) s_ratio[(P) & 1][r] = 1.0f;                                              \
                    }                                                                                                   \
                }                                                                                                       \
                if ((lane % LPR) == 0) { s_rscale[r] = scale; s_rinv[r] = inv; }                                        \
            }                                                                                                           \
            uint2 hi, mid;                                                                                              \
            split4_t<F16>(v, scale, hi, mid);                                                                           \
            unsigned char* p_ = (BUF) + r * ASTR + lk * 2;                                                              \
            *reinterpret_cast<uint2*>(p_) = hi;                                                                         \
            *reinterpret_cast<uint2*>(p_ + APLANE) = mid;                                                               \
        }                                                                                                               \
    }

    TS_LOAD_W(0, 0)
    TS_LOAD_A(0, 0)
    if (n_pan > 1) { TS_LOAD_W(1, 1) TS_LOAD_A(1, 1) }
    if constexpr (WR > 2) {
        if (n_pan > 2) TS_LOAD_W(2, 2)
    }
    TS_COMMIT_A(0, 0, sA[0])
    if (n_pan > 2) TS_LOAD_A(0, 2)

    f32x16 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.0f;
    const int frow = lane & 31, khalf = lane >> 5;
#define TS_LOAD_AF(SET, KCI)                                                                                            \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                                    \
        fah[SET][r] = *reinterpret_cast<const bf16x8*>(sl_ + r * 32 * ASTR + (KCI) * 32);                               \
        fam[SET][r] = *reinterpret_cast<const bf16x8*>(sl_ + r * 32 * ASTR + (KCI) * 32 + APLANE);                      \
    }
#define TS_COMPUTE(WB, LB)                                                                                              \
    {                                                                                                                   \
        const unsigned char* sl_ = sA[LB] + frow * ASTR + khalf * 16;                                                   \
        bf16x8 fah[2][RT], fam[2][RT];                                                                                  \
        TS_LOAD_AF(0, 0)                                                                                                \
        _Pragma("unroll") for (int kc = 0; kc < KCP; ++kc) {                                                            \
            if (kc + 1 < KCP) { TS_LOAD_AF((kc + 1) & 1, kc + 1) }                                                      \
            _Pragma("unroll") for (int r = 0; r < RT; ++r) acc[r] = mfma32_t<F16>(fam[kc & 1][r], wh[WB][kc], acc[r]);  \
            _Pragma("unroll") for (int r = 0; r < RT; ++r) acc[r] = mfma32_t<F16>(fah[kc & 1][r], wm[WB][kc], acc[r]);  \
            _Pragma("unroll") for (int r = 0; r < RT; ++r) acc[r] = mfma32_t<F16>(fah[kc & 1][r], wh[WB][kc], acc[r]);  \
        }                                                                                                               \
    }
    // a panel whose rows were split with a LOWERED scale: the accumulators of those rows follow (C/D layout of the 32x32 MFMA:
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).  Flag = panel index + 1 and ratios by panel parity: nothing to clear, the
    // slot of panel P is rewritten by panel P + 2 two barriers after its last reader.
#define TS_RESCALE(P)                                                                                                   \
    if constexpr (F16) {                                                                                                \
        if ((P) > 0 && s_flag[(P) & 1] == (P) + 1) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                              \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
                    const float4 rv = *reinterpret_cast<const float4*>(&s_ratio[(P) & 1][r * 32 + 8 * q + 4 * khalf]);  \
                    acc[r][4 * q] *= rv.x; acc[r][4 * q + 1] *= rv.y; acc[r][4 * q + 2] *= rv.z; acc[r][4 * q + 3] *= rv.w; \
                }                                                                                                       \
        }                                                                                                               \
    }
    // panel P0 + I: fragments in ring slot I % 3, rows in LDS panel I & 1; afterwards panel + 1 is split into the other LDS panel out of
    // register set (I + 1) & 1, which then takes the rows of panel + 3, and ring slot I % 3 takes the fragments of panel + 3
#define TS_STEP(I)                                                                                                      \
    if (P0 + (I) < n_pan) {                                                                                             \
        const int P = P0 + (I);                                                                                         \
        TS_RESCALE(P)                                                                                                   \
        TS_COMPUTE((I) % WR, (I) & 1)                                                                                   \
        if (P + WR < n_pan) TS_LOAD_W((I) % WR, P + WR)                                                                 \
        if (P + 1 < n_pan) {                                                                                            \
            TS_COMMIT_A(((I) + 1) & 1, P + 1, sA[((I) + 1) & 1])                                                        \
            if (P + 3 < n_pan) TS_LOAD_A(((I) + 1) & 1, P + 3)                                                          \
            tile_barrier();                                                                                             \
        }                                                                                                               \
    }

    tile_barrier();
    for (int P0 = 0; P0 < n_pan; P0 += 6) {
        TS_STEP(0) TS_STEP(1) TS_STEP(2) TS_STEP(3) TS_STEP(4) TS_STEP(5)
    }
#undef TS_STEP
#undef TS_COMPUTE
#undef TS_LOAD_AF
#undef TS_LOAD_W
#undef TS_LOAD_A
#undef TS_COMMIT_A
#undef TS_RESCALE

    // ---- epilogue: bias, optional tanh, 16-byte stores after the quad transpose
    const bool o1 = lane & 1, o2 = lane & 2;
    const int col_l = ((lane & 31) >> 2) * 4, rt0 = (lane & 3) + 4 * (lane >> 5);
    const int col = pass * BNP + wave * 32 + col_l;
    if (col < n_out) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + (int64_t)g * a.bgs + col);
        const int blk = col / a.block_cols, cc = col - blk * a.block_cols;
        float* __restrict__ ob = (blk == 0) ? a.out0 : ((blk == 1) ? a.out1 : a.out2);
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = acc[r][4 * q], v1 = acc[r][4 * q + 1], v2 = acc[r][4 * q + 2], v3 = acc[r][4 * q + 3];
                quad_transpose(v0, v1, v2, v3, o1, o2);
                const int rt = r * 32 + rt0 + 8 * q;
                if (rt < nrows) {
                    const float sc = F16 ? s_rinv[F16 ? rt : 0] * winv : 1.0f;
                    const int64_t orow = a.by_pos ? (int64_t)(row0 + rt) : (int64_t)s_rid[rt];
                    float4 o4 = make_float4(v0 * sc + b4.x, v1 * sc + b4.y, v2 * sc + b4.z, v3 * sc + b4.w);
                    if (a.prologue) o4 = make_float4(hgt_tanh(o4.x), hgt_tanh(o4.y), hgt_tanh(o4.z), hgt_tanh(o4.w));
                    store_wt16(ob + orow * a.block_cols + cc, o4.x, o4.y, o4.z, o4.w);
                }
            }
    }
}

#ifndef HGT_TILE_MAX_TILES
#define HGT_TILE_MAX_TILES 256
#endif

template <int RT>
static void launch_stream(bool f16, const TileArgs& a, int64_t n_rows, hipStream_t stream) {
    const int64_t row_tiles = (n_rows + 32 * RT - 1) / (32 * RT) + a.n_groups;      // device-side group sizes: the upper bound
    const int64_t work = row_tiles * ((a.n_out + BNP - 1) / BNP);
    dim3 grid((unsigned)((work + 7) / 8 * 8));                                       // (a multiple of 8: one chunk of the work list per XCD)
    if (f16) k_tile_linear_stream<RT, true><<<grid, 512, 0, stream>>>(a);
    else k_tile_linear_stream<RT, false><<<grid, 512, 0, stream>>>(a);
}

template <int NW, int RT, int CT, int KPAN, int NPM, bool UPD>
static void launch_tile(bool f16, const TileArgs& a, int64_t n_rows, hipStream_t stream) {
    constexpr int BMT = 32 * RT, BNT = 32 * CT * NW;
    const int64_t row_tiles = (n_rows + BMT - 1) / BMT + a.n_groups;      // device-side group sizes: the upper bound
    dim3 grid((unsigned)row_tiles, (unsigned)((a.n_out + BNT - 1) / BNT));
    if (f16) k_tile_linear<NW, RT, CT, KPAN, NPM, UPD, true><<<grid, 64 * NW, 0, stream>>>(a);
    else k_tile_linear<NW, RT, CT, KPAN, NPM, UPD, false><<<grid, 64 * NW, 0, stream>>>(a);
}

}  // namespace

// 1 = launched, 0 = not this kernel's domain (the caller takes the persistent / slab kernels), < 0 = error.
// `upd` = nullptr: the plain typed linear; otherwise {x_skip, ld_skip, skip, ln_w, ln_b, use_norm} of the fused update (n_out <= 512).
int hgt_typed_linear_tile_try(bool f16, const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                              int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias, int64_t bgs, float* out0,
                              float* out1, float* out2, int32_t block_cols, int32_t by_pos, int32_t prologue, const HgtTileUpdate* upd,
                              void* stream_) {
    // (prologue 1 = gelu on load -- DenseHGTConv's out_linear, the unfused wide update -- stays on the slab kernels: the inlined erf next
    //  to both fragment buffers spills (4.7 KB of scratch per lane in the fp16 form))
    const int act_tanh = (prologue & HGT_LINEAR_TANH) ? 1 : 0;      // tanh on the output (the GNN's typed adapter, model.py:70-76)
    prologue &= ~HGT_LINEAR_TANH;
    if (n_rows > HGT_TILE_MAX_ROWS || prologue != 0 || n_groups > 64 || (upd && act_tanh)) return 0;
    if (((n_out | block_cols) & 3) != 0) return 0;
    if (upd && (n_out > 512 || k > (n_out <= 256 ? 256 : 512) || (upd->ld_skip & 3) != 0 || ((uintptr_t)upd->x_skip & 15) != 0)) return 0;
    TileArgs a;
    a.x = x; a.ldx = ldx; a.rows = rows; a.group_off = group_off; a.n_groups = n_groups; a.k = k; a.n_out = n_out;
    a.wsplit = (const unsigned short*)w_split; a.bias = bias; a.bgs = bgs; a.out0 = out0; a.out1 = out1; a.out2 = out2;
    a.block_cols = block_cols; a.by_pos = by_pos; a.prologue = act_tanh;      // (the field carries the output activation: gelu-on-load is not this kernel's)
    a.vec_ok = (ldx % 4 == 0) && (k % 4 == 0) && k >= 4 && (((uintptr_t)x & 15) == 0) ? 1 : 0;
    a.xs = nullptr; a.ldxs = 0; a.skip = nullptr; a.lnw = nullptr; a.lnb = nullptr; a.use_norm = 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (upd) {
        a.xs = upd->x_skip; a.ldxs = upd->ld_skip; a.skip = upd->skip; a.lnw = upd->ln_w; a.lnb = upd->ln_b; a.use_norm = upd->use_norm;
        if (n_out <= 256) launch_tile<8, 1, 1, 128, 2, true>(f16, a, n_rows, stream);      // (c3: 11.9 us against 19.2 on the persistent kernel)
        else launch_tile<8, 1, 2, 64, 8, true>(f16, a, n_rows, stream);
    } else {
        // plain linear: 32 x 128 tiles of four wavefronts, four workgroups per CU (K panels of 64: 120 registers) -- measured at c3
        // (3 200 x 256 -> 768): 11.3 us against 14.0 (64 x 128 tiles) and 13.9 (persistent kernel).  Beyond K = 256 or ~1 000 workgroups
        // the slab kernels are as fast or faster (K = 512: 36 vs 31 us at 3 200 rows): not this kernel's domain
        const int64_t wgs = ((n_rows + 31) / 32 + n_groups) * ((n_out + 127) / 128);
        if (k > 256) {
            // (32 RT) x 256 tiles, RT <= 4, when all of them are RESIDENT at once (RT = 1: two workgroups per CU, 512 tiles; otherwise one
            // per CU, 256); among those the RT with the shortest workgroup under a two-term model -- matrix-core time ~ 6 RT, bytes
            // through its CU's L1 ~ (RT + 8): whichever is larger -- ties: the larger tile (fewer passes over W).  Measured (r6, warm,
            // f16x3): 3 200 x 512 -> 1 536: RT = 3 (216 tiles) 28 us, slab kernel 37; 4 096 x 1 169 -> 400: RT = 1 (262 tiles) 31, RT = 2
            // 49, slab 52; 4 096 x 400 -> 1 536: RT = 4 (210 tiles) 32, slab 38, RT = 1 in 1.5 rounds 31 (not taken: slower at n_hid 512)
            const int64_t passes = (n_out + BNP - 1) / BNP;
            int best = 0;
            int64_t best_cost = 0;
            for (int rt = 1; rt <= 4; ++rt) {
                const int64_t tiles = ((n_rows + 32 * rt - 1) / (32 * rt) + n_groups / 2) * passes;      // (about half the groups end in a partial tile)
                if (tiles > ((HGT_TS_R1_SMALL && rt == 1) ? HGT_TS_R1_TILES : HGT_TILE_MAX_TILES)) continue;
                const int64_t cost = std::max<int64_t>(6 * rt, rt + 8);
                if (best == 0 || cost <= best_cost) { best = rt; best_cost = cost; }
            }
            if (best == 0) return 0;
            if (best == 1) launch_stream<1>(f16, a, n_rows, stream);
            else if (best == 2) launch_stream<2>(f16, a, n_rows, stream);
            else if (best == 3) launch_stream<3>(f16, a, n_rows, stream);
            else launch_stream<4>(f16, a, n_rows, stream);
        } else {
            if (wgs > 1024) return 0;
            launch_tile<4, 1, 1, 64, 4, false>(f16, a, n_rows, stream);
        }
    }
    if (hipGetLastError() != hipSuccess) return HGT_ERR_LAUNCH;
    return 1;
}
