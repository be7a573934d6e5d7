// Backward pass of HGTConv: the kernels that have no forward counterpart.  The reference gets its
// gradients from autograd through conv.py:60-134 (OAG/train_paper_field.py:249 `loss.backward()`); here the chain rule is
// written out on the node-level algebra of the forward so that the E x d tensors never exist either:
//
//   out = LN_t(y), y = o a + x (1 - a), a = sigmoid(skip_t), o = D * (gelu(agg) W_a^T + b_a)   (D = dropout mask / keep prob.)
//       -> hgt_node_update_bwd: d o, d x (skip path), d skip, d LN weight / bias
//       -> d gelu(agg) = d o W_a (typed linear with W_a^T), d agg = . * gelu'(agg) (hgt_gelu_bwd),
//          d W_a / d b_a = typed weight gradient (hgt_typed_wgrad / hgt_typed_colsum)
//   agg_i,h = sum_e att_e (v_e M_r):   d att_e = <dagg_i M_r^T, v_e>      = the LOGITS kernel with (Q, K, A') := (dagg, V, M^T)
//                                      d s_e   = att_e (d att_e - <dagg_i, agg_i>_h)                 (hgt_edge_softmax_bwd)
//   s_e = <A'_r q_i, k_e>:             d Q_i = sum_r (sum_e ds_e k_e) A'_r          = hgt_edge_spmm on the graph
//                                      d K_j = sum_r A'_r (sum_e ds_e q_i)          = hgt_edge_spmm on the TRANSPOSED graph
//                                      d V_j = sum_r (sum_e att_e dagg_i) M_r^T     = hgt_edge_spmm on the TRANSPOSED graph
//                                      d M_r = sum_e att_e v_e^T dagg_i,  d A'_r = sum_e ds_e k_e^T q_i   (hgt_relation_outer)
//   Q|K|V = x W_qkv^T + b:             d x += [dQ|dK|dV] W_qkv (typed linear), d W_qkv / d b_qkv = typed weight gradient.
// Everything is enqueued on the caller's stream; small parameter gradients are accumulated with fp32 atomics into
// caller-zeroed buffers (run-to-run differences of the summation order only).
#include "hgt_edge_common.h"
#include "hgt_split_common.h"

namespace {


__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// node update backward (conv.py:125-133 in reverse).  One wavefront per chunk of ROWS_PER_WAVE consecutive rows; the
// per-type parameter gradients are summed in registers while the type does not change and flushed with atomics.
// ---------------------------------------------------------------------------------------------
constexpr int NUB_ROWS = 32;
constexpr int NUB_MAXC = 8;     // columns per lane: d <= 512

__global__ __launch_bounds__(256) void k_node_update_bwd(
    const float* __restrict__ gout, const float* __restrict__ trans, const float* __restrict__ x, int64_t ldx,
    const int64_t* __restrict__ node_type, const float* __restrict__ skip, const float* __restrict__ lnw, int use_norm,
    const float* __restrict__ drop_mask, int64_t NQ, int d, int T, float* __restrict__ d_trans, float* __restrict__ dx, int64_t ld_dx,
    float* __restrict__ d_alpha, float* __restrict__ d_lnw, float* __restrict__ d_lnb, int shared_norm, int rows_per_wave) {
    // skip == NULL: plain residual y = o + x (DenseHGTConv.update, conv.py:259,271), no gate gradient;
    // shared_norm: ONE LayerNorm for every type (out_norm, conv.py:272): its parameters / gradients are row 0 of lnw / d_lnw / d_lnb
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r0 = wave * rows_per_wave;
    if (r0 >= NQ) return;
    const int nc = (d + 63) / 64;
    float gw[NUB_MAXC], gb[NUB_MAXC], ga = 0.0f;
#pragma unroll
    for (int c = 0; c < NUB_MAXC; ++c) gw[c] = gb[c] = 0.0f;
    int cur_t = -1;
    auto flush = [&]() {
        if (cur_t >= 0) {
            if (use_norm) {
#pragma unroll
                for (int c = 0; c < NUB_MAXC; ++c) {
                    const int col = c * 64 + lane;
                    if (c < nc && col < d) {
                        const int64_t lrow = shared_norm ? 0 : cur_t;
                        unsafeAtomicAdd(&d_lnw[lrow * d + col], gw[c]);
                        unsafeAtomicAdd(&d_lnb[lrow * d + col], gb[c]);
                    }
                    gw[c] = gb[c] = 0.0f;
                }
            }
            if (skip) {
                ga = wave_sum(ga);
                if (lane == 0) unsafeAtomicAdd(&d_alpha[cur_t], ga);
            }
            ga = 0.0f;
        }
    };
    for (int64_t r = r0; r < min(r0 + (int64_t)rows_per_wave, NQ); ++r) {
        const int64_t t64 = node_type[r];
        const int t = (t64 >= 0 && t64 < T) ? (int)t64 : -1;
        if (t != cur_t) { flush(); cur_t = t; }
        if (t < 0) {     // rows of unknown type: output 0, no gradient (conv.py:120)
#pragma unroll
            for (int c = 0; c < NUB_MAXC; ++c) {
                const int col = c * 64 + lane;
                if (c < nc && col < d) { d_trans[r * d + col] = 0.0f; dx[r * ld_dx + col] = 0.0f; }
            }
            continue;
        }
        const float alpha = skip ? 1.0f / (1.0f + expf(-skip[t])) : 1.0f;
        const float beta = skip ? 1.0f - alpha : 1.0f;           // weight of the residual row
        float o[NUB_MAXC], xv[NUB_MAXC], g[NUB_MAXC], y[NUB_MAXC];
        float s1 = 0.0f;
#pragma unroll
        for (int c = 0; c < NUB_MAXC; ++c) {
            const int col = c * 64 + lane;
            const bool ok = c < nc && col < d;
            o[c] = ok ? trans[r * d + col] : 0.0f;
            xv[c] = ok ? x[r * ldx + col] : 0.0f;
            g[c] = ok ? gout[r * d + col] : 0.0f;
            y[c] = o[c] * alpha + xv[c] * beta;
            s1 += y[c];
        }
        float dy[NUB_MAXC];
        if (use_norm) {
            const float mean = wave_sum(s1) / (float)d;
            float s2 = 0.0f;
#pragma unroll
            for (int c = 0; c < NUB_MAXC; ++c) {
                const int col = c * 64 + lane;
                const bool ok = c < nc && col < d;
                y[c] = ok ? y[c] - mean : 0.0f;
                s2 += y[c] * y[c];
            }
            const float rstd = rsqrtf(wave_sum(s2) / (float)d + 1e-5f);
            float a1 = 0.0f, a2 = 0.0f;
            float gh[NUB_MAXC];
#pragma unroll
            for (int c = 0; c < NUB_MAXC; ++c) {
                const int col = c * 64 + lane;
                const bool ok = c < nc && col < d;
                y[c] *= rstd;                                            // y = normalised row
                const float w = ok ? lnw[(int64_t)(shared_norm ? 0 : t) * d + col] : 0.0f;
                gw[c] += g[c] * y[c];
                gb[c] += g[c];
                gh[c] = g[c] * w;
                a1 += gh[c];
                a2 += gh[c] * y[c];
            }
            a1 = wave_sum(a1) / (float)d;
            a2 = wave_sum(a2) / (float)d;
#pragma unroll
            for (int c = 0; c < NUB_MAXC; ++c) dy[c] = rstd * (gh[c] - a1 - y[c] * a2);
        } else {
#pragma unroll
            for (int c = 0; c < NUB_MAXC; ++c) dy[c] = g[c];
        }
#pragma unroll
        for (int c = 0; c < NUB_MAXC; ++c) {
            const int col = c * 64 + lane;
            if (c < nc && col < d) {
                ga += dy[c] * (o[c] - xv[c]);
                float dt = dy[c] * alpha;
                if (drop_mask) dt *= drop_mask[r * d + col];            // o = mask * (a_linear output), conv.py:125
                d_trans[r * d + col] = dt;
                dx[r * ld_dx + col] = dy[c] * beta;
            }
        }
    }
    flush();
}

// dagg = dg * gelu'(agg), gelu = exact erf form (conv.py:119)
__global__ void k_gelu_bwd(const float* __restrict__ dg, const float* __restrict__ agg, float* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4*>(agg + i);
    const float4 g = *reinterpret_cast<const float4*>(dg + i);
    auto f = [](float v, float gg) {
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
        return gg * (cdf + v * pdf);
    };
    *reinterpret_cast<float4*>(out + i) = make_float4(f(a.x, g.x), f(a.y, g.y), f(a.z, g.z), f(a.w, g.w));
}

// x[i] *= m[i]  (dropout of the a_linear output, conv.py:125; the mask holds 0 or 1/(1-p))
__global__ void k_mul_inplace(float* __restrict__ x, const float* __restrict__ m, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= m[i];
}

// ds[p][h] = att[p][h] * (datt[p][h] - rho[dst[p]][h])   (softmax backward per target and head; sorted edge order)
__global__ void k_edge_softmax_bwd(const int32_t* __restrict__ edst, const float* __restrict__ att, const float* __restrict__ datt,
                                   const float* __restrict__ rho, int64_t ld_rho, float* __restrict__ ds, int64_t E, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * H) return;
    const int64_t p = i / H;
    const int h = (int)(i % H);
    ds[i] = att[i] * (datt[i] - rho[(int64_t)edst[p] * ld_rho + h]);
}

// out[p][h] = in[eid[p]][h]: values in ORIGINAL edge order -> the sorted order of a plan
__global__ void k_gather_sorted(const int32_t* __restrict__ eid, const float* __restrict__ in, float* __restrict__ out, int64_t E, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E * H) return;
    const int64_t p = i / H;
    out[i] = in[(int64_t)eid[p] * H + (i % H)];
}

// rho[n][h] = <a[n][h*dkp .. +dkp], b[n][...]>: a thread per 4 consecutive columns (coalesced 16 B loads of both rows), partial
// dots reduced over the dkp/4 consecutive threads of a head (dkp is a power of two)
__global__ void k_head_dot(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows, int H, int dkp, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
    const int64_t total = n_rows * H * (dkp / 4);
    float s = 0.0f;
    if (i < total) {
        const float4 u = *reinterpret_cast<const float4*>(a + 4 * i);
        const float4 v = *reinterpret_cast<const float4*>(b + 4 * i);
        s = u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
    const int tph = dkp / 4;                                               // threads per head: 1 .. 64
    for (int o = 1; o < tph; o <<= 1) s += __shfl_xor(s, o);
    if (i < total && (i % tph) == 0) out[i / tph] = s;
}

// ---------------------------------------------------------------------------------------------
// Typed weight gradient  dW[g][m][n] += sum_{rows p of group g} A[rows[p]][m] * B[rows[p]][n]   (= A_g^T B_g)
// on v_mfma_f32_32x32x2_f32 (exact fp32 products).  A workgroup owns a 64 x 64 tile of (m, n) and a chunk of WG_ROWS rows of
// one group: the 64-row slices of A and B go through LDS ([row][64] fp32), each wavefront owns a 32 x 32 quadrant, the
// partial tile is added to dW with fp32 atomics (one pass over every row per (m, n) tile: A is re-read n_out/64 times, B
// m/64 times -- fine for a one-off per step; the forward GEMMs are the optimised ones).
// ---------------------------------------------------------------------------------------------
constexpr int WG_ROWS = 2048;

__global__ __launch_bounds__(256) void k_typed_wgrad(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                     const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off, int n_groups,
                                                     int M, int Nc, float* __restrict__ out, int64_t out_group_stride, int vecA, int vecB) {
    __shared__ float sA[64][68];
    __shared__ float sB[64][68];
    // which (group, row chunk) is this block?
    int slot = blockIdx.x, g = 0, gbeg = 0, gend = 0, before = 0;
    for (; g < n_groups; ++g) {
        gbeg = group_off[g];
        gend = group_off[g + 1];
        const int nch = (gend - gbeg + WG_ROWS - 1) / WG_ROWS;
        if (slot < before + nch) break;
        before += nch;
    }
    if (g >= n_groups) return;
    const int p0 = gbeg + (slot - before) * WG_ROWS, p1 = min(p0 + WG_ROWS, gend);
    const int m0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int pb = p0; pb < p1; pb += 64) {
        __syncthreads();
        // 64 rows x 64 columns of A and of B: thread -> (row = tid / 4 .. , 16 columns)
        {
            const int r = tid >> 2, cq = (tid & 3) * 16;
            const int p = pb + r;
            const int64_t rid = (p < p1) ? (int64_t)rows[p] : -1;
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                if (rid >= 0) {
                    const int ma = m0 + cq + j, nb = n0 + cq + j;
                    if (vecA && ma + 3 < M) a = *reinterpret_cast<const float4*>(A + rid * lda + ma);
                    else { if (ma < M) a.x = A[rid * lda + ma]; if (ma + 1 < M) a.y = A[rid * lda + ma + 1]; if (ma + 2 < M) a.z = A[rid * lda + ma + 2]; if (ma + 3 < M) a.w = A[rid * lda + ma + 3]; }
                    if (vecB && nb + 3 < Nc) b = *reinterpret_cast<const float4*>(B + rid * ldb + nb);
                    else { if (nb < Nc) b.x = B[rid * ldb + nb]; if (nb + 1 < Nc) b.y = B[rid * ldb + nb + 1]; if (nb + 2 < Nc) b.z = B[rid * ldb + nb + 2]; if (nb + 3 < Nc) b.w = B[rid * ldb + nb + 3]; }
                }
                *reinterpret_cast<float4*>(&sA[r][cq + j]) = a;
                *reinterpret_cast<float4*>(&sB[r][cq + j]) = b;
            }
        }
        __syncthreads();
        // D[m][n] += sum_row A[row][m] B[row][n]: MFMA operand a = A^T[m = lane&31][k = row], b = B[k = row][n = lane&31]
#pragma unroll 8
        for (int k = 0; k < 64; k += 2) {
            const int kr = k + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[kr][wm + (lane & 31)], sB[kr][wn + (lane & 31)], acc, 0, 0, 0);
        }
    }
    // C layout: col (n) = lane & 31, row (m) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = out + (int64_t)g * out_group_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = n0 + wn + (lane & 31);
        if (m < M && n < Nc) unsafeAtomicAdd(&o[(int64_t)m * Nc + n], acc[r]);
    }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient as 3-term split-bf16 products on v_mfma_f32_32x32x16_bf16 (relative error of a product ~3 * 2^-18, like
// the forward typed linears): 8x the matrix-core rate of the fp32 instruction and 128 x 128 output tiles, so that A and B are
// re-read Nc/128 and M/128 times instead of Nc/64 and M/64 (the fp32 kernel above moved 33 GB per training step at c2 = 11 ms).
// Both MFMA operands need the ROW index along K, i.e. eight consecutive rows of one column per lane: the 32-row chunks of A and B
// are therefore staged through LDS TRANSPOSED -- a thread reads eight rows of one column (coalesced 256 B per wavefront and row),
// splits them into bf16 hi / mid and writes one 16 B fragment piece per plane; column stride 80 B: conflict-free writes and reads.
// The next chunk's rows are in flight (registers) while the current one is multiplied.  Optionally also the column sums of A
// (bias gradient), from the registers that pass through anyway.
// ---------------------------------------------------------------------------------------------
constexpr int WX_T = 128;            // tile edge (columns of A = rows of dW, columns of B)
constexpr int WX_KR = 32;            // rows per chunk
constexpr int WX_CS = 80;            // LDS bytes per column: 32 rows x 2 B + 16 B of padding
constexpr int WX_PLANE = WX_T * WX_CS;
constexpr int WX_ROWS = 4096;        // rows of one group per workgroup

__global__ __launch_bounds__(256) void k_typed_wgrad_x3(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                        const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
                                                        int n_groups, int M, int Nc, int n_mt, float* __restrict__ out,
                                                        int64_t out_group_stride, float* __restrict__ colsum, int64_t cs_group_stride) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WX_PLANE];   // A hi | A mid | B hi | B mid
    // tile index fastest: the workgroups that share a row chunk are neighbours in launch order (their rows meet in the L2)
    const int mt = blockIdx.x % n_mt, nt = blockIdx.x / n_mt;
    int slot = blockIdx.y, g = 0, gbeg = 0, gend = 0, before = 0;
    for (; g < n_groups; ++g) {
        gbeg = group_off[g];
        gend = group_off[g + 1];
        const int nch = (gend - gbeg + WX_ROWS - 1) / WX_ROWS;
        if (slot < before + nch) break;
        before += nch;
    }
    if (g >= n_groups) return;
    const int p0 = gbeg + (slot - before) * WX_ROWS, p1 = min(p0 + WX_ROWS, gend);
    const int m0 = mt * WX_T, n0 = nt * WX_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    // staging role: column c of the tile, row octets o0 and o0 + 2 of the chunk, for A and for B
    const int c = tid & 127, o0 = tid >> 7;
    const bool a_ok = m0 + c < M, b_ok = n0 + c < Nc;
    const float* __restrict__ pa = A + (a_ok ? m0 + c : 0);
    const float* __restrict__ pb = B + (b_ok ? n0 + c : 0);
    float va[16], vb[16];
    int rid[16], rid_next[16];       // row ids of the chunk in flight / of the one after it (no id -> row dependency inside the loop)
    auto load_ids = [&](int pbase, int (&ids)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = pbase + 8 * (o0 + 2 * (j >> 3)) + (j & 7);
            ids[j] = (p < p1) ? rows[p] : -1;
        }
    };
    auto load_rows = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t r = max(rid[j], 0);
            va[j] = pa[r * lda];
            vb[j] = pb[r * ldb];
        }
    };
    float csum = 0.0f;
    auto commit = [&]() {       // registers -> transposed bf16 hi / mid planes (rows beyond the chunk and columns beyond M / Nc: 0)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float fa[8], fb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool live = rid[half * 8 + j] >= 0;
                fa[j] = (live && a_ok) ? va[half * 8 + j] : 0.0f;
                fb[j] = (live && b_ok) ? vb[half * 8 + j] : 0.0f;
                csum += fa[j];
            }
            uint4 ah, am, bh, bm;
            split2(fa[0], fa[1], ah.x, am.x); split2(fa[2], fa[3], ah.y, am.y); split2(fa[4], fa[5], ah.z, am.z); split2(fa[6], fa[7], ah.w, am.w);
            split2(fb[0], fb[1], bh.x, bm.x); split2(fb[2], fb[3], bh.y, bm.y); split2(fb[4], fb[5], bh.z, bm.z); split2(fb[6], fb[7], bh.w, bm.w);
            unsigned char* w = smem + c * WX_CS + (o0 + 2 * half) * 16;
            *reinterpret_cast<uint4*>(w) = ah;
            *reinterpret_cast<uint4*>(w + WX_PLANE) = am;
            *reinterpret_cast<uint4*>(w + 2 * WX_PLANE) = bh;
            *reinterpret_cast<uint4*>(w + 3 * WX_PLANE) = bm;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_ids(p0, rid);
    load_rows();
    load_ids(p0 + WX_KR, rid_next);
    for (int pbase = p0; pbase < p1; pbase += WX_KR) {
        __syncthreads();                 // the previous chunk's fragments have been read
        commit();
        if (pbase + WX_KR < p1) {        // next chunk's rows (in flight during the products below), the ids of the one after it
#pragma unroll
            for (int j = 0; j < 16; ++j) rid[j] = rid_next[j];
            load_rows();
            load_ids(pbase + 2 * WX_KR, rid_next);
        }
        __syncthreads();
        const unsigned char* fa = smem + (wm + (lane & 31)) * WX_CS + (lane >> 5) * 16;
        const unsigned char* fb = smem + 2 * WX_PLANE + (wn + (lane & 31)) * WX_CS + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], am[2], bh[2], bm[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(fa + i * 32 * WX_CS + ks * 32);
                am[i] = *reinterpret_cast<const bf16x8*>(fa + WX_PLANE + i * 32 * WX_CS + ks * 32);
                bh[i] = *reinterpret_cast<const bf16x8*>(fb + i * 32 * WX_CS + ks * 32);
                bm[i] = *reinterpret_cast<const bf16x8*>(fb + WX_PLANE + i * 32 * WX_CS + ks * 32);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    // C layout of a 32 x 32 tile: col (n) = lane & 31, row (m) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = out + (int64_t)g * out_group_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = n0 + wn + 32 * j + (lane & 31);
                if (m < M && n < Nc) unsafeAtomicAdd(&o[(int64_t)m * Nc + n], acc[i][j][r]);
            }
    if (colsum && nt == 0 && a_ok) unsafeAtomicAdd(&colsum[(int64_t)g * cs_group_stride + m0 + c], csum);
}

// out[g][c] += sum_{rows p of group g} A[rows[p]][c]    (bias gradients)
__global__ __launch_bounds__(256) void k_typed_colsum(const float* __restrict__ A, int64_t lda, const int32_t* __restrict__ rows,
                                                      const int32_t* __restrict__ group_off, int n_groups, int M, float* __restrict__ out,
                                                      int64_t out_group_stride) {
    constexpr int CH = 256;
    int slot = blockIdx.x * 4 + (threadIdx.x >> 6), g = 0, gbeg = 0, gend = 0, before = 0;
    for (; g < n_groups; ++g) {
        gbeg = group_off[g];
        gend = group_off[g + 1];
        const int nch = (gend - gbeg + CH - 1) / CH;
        if (slot < before + nch) break;
        before += nch;
    }
    if (g >= n_groups) return;
    const int lane = threadIdx.x & 63;
    const int p0 = gbeg + (slot - before) * CH, p1 = min(p0 + CH, gend);
    for (int c0 = 0; c0 < M; c0 += 64 * 4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = p0; p < p1; ++p) {
            const int64_t rid = rows[p];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j * 64 + lane;
                if (c < M) s[j] += A[rid * lda + c];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j * 64 + lane;
            if (c < M) unsafeAtomicAdd(&out[(int64_t)g * out_group_stride + c], s[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Relation outer products:  out[r][h][k][c] += sum_{e of relation r} w_e,h * a[src_e][h][k] * b[dst_e][h][c]
// (d relation_msg with (w, a, b) = (att, V, dagg); d A' with (ds, K, Q)).  A wavefront takes the work items of ONE relation
// (blockIdx.z) inside its slice of the plan's item list (runs of <= 512 sorted edges of one (tile, relation)); lane = (head,
// VEC rows k of the head's block); the b row of the edge is broadcast inside the head's lanes through LDS; the dkp x dkp
// blocks accumulate in registers over ~64 items and are flushed once with atomics.
// ---------------------------------------------------------------------------------------------
template <int VEC, int LPH, bool RTE>
__global__ __launch_bounds__(256) void k_relation_outer(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ w, const float* __restrict__ a,
    const float* __restrict__ rte_a, const float* __restrict__ b, float* __restrict__ out, int R, int HT, int items_per_wave) {
    constexpr int DKP = VEC * LPH, DP = 64 * VEC, H = 64 / LPH;
    __shared__ __attribute__((aligned(16))) float s_b[4][DP + 4 * (64 / LPH)];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hg = blockIdx.y;
    const int rel_sel = blockIdx.z;            // this wavefront only takes the items of ONE relation: one flush per wavefront
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP;
    const int h = lane / LPH, p = lane % LPH;
    float* bounce = s_b[wib];
    const int n_items = hdr->n_items;
    const int first = (blockIdx.x * 4 + wib) * items_per_wave;
    if (first >= n_items) return;
    float acc[VEC][DKP];      // rows k = p*VEC + i of head h, all DKP columns
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int c = 0; c < DKP; ++c) acc[i][c] = 0.0f;
    bool any = false;
    for (int ib = first; ib < min(first + items_per_wave, n_items); ib += 64) {
        // 64 item headers at a time (lane i = item ib + i); the matching ones are walked one after the other
        const int my_i = min(ib + lane, n_items - 1);
        const HgtItem mine = items[my_i];
        const bool take = (ib + lane < min(first + items_per_wave, n_items)) && mine.rel == rel_sel;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(take);
        while (todo) {
            const int li_ = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int beg = __builtin_amdgcn_readlane(mine.beg, li_), end = __builtin_amdgcn_readlane(mine.end, li_);
            any = true;
            for (int base = beg; base < end; base += 64) {
                const int nb = min(64, end - base);
                const int li = base + min(lane, nb - 1);
                const int my_src = esrc[li], my_dst = edst[li];
                const int my_rte = RTE ? (int)ertei[li] : 0;
                // UB edges per batch: all their row / weight loads are issued (unconditionally: slots beyond the chunk re-read
                // its last edge) before the first one is consumed -- one memory round trip per batch instead of one per edge
                // (the per-edge form ran 15.7 ms at c2, the whole backward pass 68 ms)
                constexpr int UB = (VEC * DKP <= 128 && !RTE) ? 8 : 4;      // (more would push the kernel past 256 registers = one wavefront per SIMD)
                for (int e0 = 0; e0 < nb; e0 += UB) {
                    float av[UB][VEC], bv[UB][VEC], tv[RTE ? UB : 1][VEC], we[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int idx = min(e0 + u, nb - 1);
                        const int s = __builtin_amdgcn_readlane(my_src, idx), dd = __builtin_amdgcn_readlane(my_dst, idx);
                        load_vec<VEC>(a + (int64_t)s * ld + co + lane * VEC, av[u]);
                        if constexpr (RTE) {
                            const int ri = __builtin_amdgcn_readlane(my_rte, idx);
                            load_vec<VEC>(rte_a + (int64_t)ri * ld + co + lane * VEC, tv[u]);
                        }
                        load_vec<VEC>(b + (int64_t)dd * ld + co + lane * VEC, bv[u]);
                        we[u] = w[(int64_t)(base + idx) * HT + hg * H + h];
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (e0 + u < nb) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) {
                                if constexpr (RTE) av[u][i] += tv[u][i];
                                av[u][i] *= we[u];
                            }
                            store_vec_lds<VEC>(bounce + lane * VEC + (lane / LPH) * 4, bv[u]);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const float* xb = bounce + h * (DKP + 4);
#pragma unroll
                            for (int c4 = 0; c4 < DKP / 4; ++c4) {
                                const float4 bb = *reinterpret_cast<const float4*>(xb + 4 * c4);
#pragma unroll
                                for (int i = 0; i < VEC; ++i) {
                                    acc[i][4 * c4 + 0] = fmaf(av[u][i], bb.x, acc[i][4 * c4 + 0]);
                                    acc[i][4 * c4 + 1] = fmaf(av[u][i], bb.y, acc[i][4 * c4 + 1]);
                                    acc[i][4 * c4 + 2] = fmaf(av[u][i], bb.z, acc[i][4 * c4 + 2]);
                                    acc[i][4 * c4 + 3] = fmaf(av[u][i], bb.w, acc[i][4 * c4 + 3]);
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
        }
    }
    if (any) {
        float* o = out + (((int64_t)rel_sel * HT + hg * H + h) * DKP + p * VEC) * DKP;
#pragma unroll
        for (int i = 0; i < VEC; ++i)
#pragma unroll
            for (int c = 0; c < DKP; ++c) unsafeAtomicAdd(&o[i * DKP + c], acc[i][c]);
    }
}

// The same sums on the matrix cores for 32-wide heads (d_k = 32: c2, c3): the outer products of an edge batch are one
// v_mfma_f32_32x32x2_f32 per (head, pair of edges) -- operand A = the two scaled source rows' 32 head columns, B = the two target
// rows' -- exact fp32 products, 256 matrix-core cycles per edge instead of ~550 vector-ALU cycles (128 FMAs per lane, LDS bounce,
// two wave barriers per edge).  The rows of a batch are parked in LDS as [edge][column] (288-float stride: the two edges of a pair
// fall into different bank halves); the 8 head blocks accumulate in 128 registers and are flushed once per wavefront.
template <int VEC, bool RTE>
__global__ __launch_bounds__(256, 2) void k_relation_outer_mfma(
    const HgtItem* __restrict__ items, const HgtPlanHeader* __restrict__ hdr, const int32_t* __restrict__ esrc,
    const int32_t* __restrict__ edst, const uint16_t* __restrict__ ertei, const float* __restrict__ w, const float* __restrict__ a,
    const float* __restrict__ rte_a, const float* __restrict__ b, float* __restrict__ out, int R, int HT, int items_per_wave) {
    constexpr int DKP = 32, LPH = DKP / VEC, DP = 64 * VEC, H = 64 / LPH, UB = 8, RS = DP + 32;   // RS: LDS row stride in floats
    static_assert(DP % 32 == 0 && H * DKP == DP, "32-wide heads");
    __shared__ __attribute__((aligned(16))) float s_rows[4][2][UB][RS];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hg = blockIdx.y, rel_sel = blockIdx.z;
    const int64_t ld = (int64_t)HT * DKP;
    const int co = hg * DP;
    const int h = lane / LPH;
    float (*sa)[RS] = s_rows[wib][0];
    float (*sb)[RS] = s_rows[wib][1];
    const int n_items = hdr->n_items;
    const int first = (blockIdx.x * 4 + wib) * items_per_wave;
    if (first >= n_items) return;
    f32x16 acc[H];
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[hh][r] = 0.0f;
    bool any = false;
    for (int ib = first; ib < min(first + items_per_wave, n_items); ib += 64) {
        const int my_i = min(ib + lane, n_items - 1);
        const HgtItem mine = items[my_i];
        const bool take = (ib + lane < min(first + items_per_wave, n_items)) && mine.rel == rel_sel;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(take);
        while (todo) {
            const int li_ = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int beg = __builtin_amdgcn_readlane(mine.beg, li_), end = __builtin_amdgcn_readlane(mine.end, li_);
            any = true;
            for (int base = beg; base < end; base += 64) {
                const int nb = min(64, end - base);
                const int li = base + min(lane, nb - 1);
                const int my_src = esrc[li], my_dst = edst[li];
                const int my_rte = RTE ? (int)ertei[li] : 0;
                for (int e0 = 0; e0 < nb; e0 += UB) {
                    float av[UB][VEC], bv[UB][VEC], tv[RTE ? UB : 1][VEC], we[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int idx = min(e0 + u, nb - 1);
                        const int s = __builtin_amdgcn_readlane(my_src, idx), dd = __builtin_amdgcn_readlane(my_dst, idx);
                        load_vec<VEC>(a + (int64_t)s * ld + co + lane * VEC, av[u]);
                        if constexpr (RTE) {
                            const int ri = __builtin_amdgcn_readlane(my_rte, idx);
                            load_vec<VEC>(rte_a + (int64_t)ri * ld + co + lane * VEC, tv[u]);
                        }
                        load_vec<VEC>(b + (int64_t)dd * ld + co + lane * VEC, bv[u]);
                        we[u] = w[(int64_t)(base + idx) * HT + hg * H + h];
                    }
                    __builtin_amdgcn_wave_barrier();          // the previous batch's operands have been read
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const float sc = (e0 + u < nb) ? we[u] : 0.0f;      // slots beyond the chunk contribute nothing
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            if constexpr (RTE) av[u][i] += tv[u][i];
                            av[u][i] *= sc;
                        }
                        store_vec_lds<VEC>(&sa[u][lane * VEC], av[u]);
                        store_vec_lds<VEC>(&sb[u][lane * VEC], bv[u]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int er = lane >> 5, cc = lane & 31;
#pragma unroll
                    for (int hh = 0; hh < H; ++hh)
#pragma unroll
                        for (int pr = 0; pr < UB / 2; ++pr)
                            acc[hh] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[2 * pr + er][hh * 32 + cc], sb[2 * pr + er][hh * 32 + cc], acc[hh],
                                                                          0, 0, 0);
                }
            }
        }
    }
    if (any) {
        // C layout of a 32 x 32 block: column c = lane & 31, row k = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            float* o = out + ((int64_t)rel_sel * HT + hg * H + hh) * DKP * DKP;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                unsafeAtomicAdd(&o[k * DKP + (lane & 31)], acc[hh][r]);
            }
        }
    }
}

template <int VEC, int LPH>
struct LaunchOuter {
    static int run(const HgtPlanView& pv, const float* w, const float* a, const float* rte_a, const float* b, float* out, int R, int HT,
                   hipStream_t stream) {
        if constexpr (VEC * LPH * VEC <= 128 && VEC * LPH >= 4) {
            // ~16 items of the selected relation per wavefront (items are ordered (tile, relation)): at c2 3 000 wavefronts for
            // 1 024 SIMDs (64 items left the chip with fewer wavefronts than SIMDs) against 25 M flush atomics
            // (sampled batches -- a few thousand 16-edge items: 16 (R + 1) items per wavefront left 13 x R wavefronts walking ~250 edges
            //  each, 565 us per call at c3 (r6 timeline of a training step); 2 (R + 1) there)
            const int ipw = (pv.L.max_items < 16384 ? 2 : 16) * (R + 1);
            const int64_t waves = (pv.L.max_items + ipw - 1) / ipw;
            dim3 grid((unsigned)((waves + 3) / 4), (unsigned)(HT / (64 / LPH)), (unsigned)R);
            if constexpr (VEC * LPH == 32 && VEC <= 4) {          // 32-wide heads: matrix-core form
                if (rte_a)
                    k_relation_outer_mfma<VEC, true><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, w, a, rte_a, b, out,
                                                                              R, HT, ipw);
                else
                    k_relation_outer_mfma<VEC, false><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, w, a, rte_a, b, out,
                                                                               R, HT, ipw);
                return HGT_OK;
            }
            if (rte_a)
                k_relation_outer<VEC, LPH, true><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, w, a, rte_a, b, out, R,
                                                                          HT, ipw);
            else
                k_relation_outer<VEC, LPH, false><<<grid, 256, 0, stream>>>(pv.items, pv.hdr, pv.esrc, pv.edst, pv.ertei, w, a, rte_a, b, out, R,
                                                                           HT, ipw);
            return HGT_OK;
        } else {
            return HGT_ERR_UNSUPPORTED;
        }
    }
};

static inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

static int node_update_bwd_impl(const float* grad_out, const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                                const float* skip, const float* ln_w, int32_t use_norm, int32_t shared_norm, const float* drop_mask,
                                int64_t n_rows, int32_t d, int32_t n_types, float* d_trans, float* dx, int64_t ld_dx, float* d_alpha,
                                float* d_ln_w, float* d_ln_b, void* stream) {
    if (!grad_out || !trans || !x || !node_type || !d_trans || !dx || (skip && !d_alpha) || n_rows < 0 || d <= 0 || d > 64 * NUB_MAXC)
        return HGT_ERR_INVALID_ARG;
    if (use_norm && (!ln_w || !d_ln_w || !d_ln_b)) return HGT_ERR_INVALID_ARG;
    if (n_rows == 0) return HGT_OK;
    // rows per wavefront: 32 amortise the parameter-gradient atomics on a large graph; a sampled batch of a few thousand rows would be
    // ~100 wavefronts walking 32 rows one after the other (c3: 119 us) -- 2 rows there, 8 in between
    const int rpw = n_rows >= 65536 ? NUB_ROWS : (n_rows >= 16384 ? 8 : 2);
    const int64_t waves = (n_rows + rpw - 1) / rpw;
    k_node_update_bwd<<<nblk(waves, 4), 256, 0, (hipStream_t)stream>>>(grad_out, trans, x, ldx, node_type, skip, ln_w, use_norm, drop_mask,
                                                                       n_rows, d, n_types, d_trans, dx, ld_dx, d_alpha, d_ln_w, d_ln_b,
                                                                       shared_norm, rpw);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_node_update_bwd(const float* grad_out, const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                                   const float* skip, const float* ln_w, int32_t use_norm, const float* drop_mask, int64_t n_rows,
                                   int32_t d, int32_t n_types, float* d_trans, float* dx, int64_t ld_dx, float* d_alpha, float* d_ln_w,
                                   float* d_ln_b, void* stream) {
    if (!skip) return HGT_ERR_INVALID_ARG;
    return node_update_bwd_impl(grad_out, trans, x, ldx, node_type, skip, ln_w, use_norm, 0, drop_mask, n_rows, d, n_types, d_trans, dx,
                                ld_dx, d_alpha, d_ln_w, d_ln_b, stream);
}

// reverse of hgt_node_update_ex: skip == NULL = plain residual (no gate, d_alpha unused), shared_norm = one LayerNorm for all types
extern "C" int hgt_node_update_bwd_ex(const float* grad_out, const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                                      const float* skip, const float* ln_w, int32_t use_norm, int32_t shared_norm, const float* drop_mask,
                                      int64_t n_rows, int32_t d, int32_t n_types, float* d_trans, float* dx, int64_t ld_dx,
                                      float* d_alpha, float* d_ln_w, float* d_ln_b, void* stream) {
    return node_update_bwd_impl(grad_out, trans, x, ldx, node_type, skip, ln_w, use_norm, shared_norm, drop_mask, n_rows, d, n_types,
                                d_trans, dx, ld_dx, d_alpha, d_ln_w, d_ln_b, stream);
}

// off2 = {0, off[n_groups]}: every row of a valid group as ONE group (the shared dense layer of DenseHGTConv)
__global__ void k_single_group_offsets(const int32_t* __restrict__ off, int n_groups, int32_t* __restrict__ off2) {
    if (threadIdx.x == 0) { off2[0] = 0; off2[1] = off[n_groups]; }
}
extern "C" int hgt_single_group_offsets(const int32_t* group_off, int32_t n_groups, int32_t* off2, void* stream) {
    if (!group_off || !off2 || n_groups <= 0) return HGT_ERR_INVALID_ARG;
    k_single_group_offsets<<<1, 64, 0, (hipStream_t)stream>>>(group_off, n_groups, off2);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_gelu_bwd(const float* dg, const float* agg, float* out, int64_t n, void* stream) {
    if (!dg || !agg || !out || n < 0 || (n & 3) != 0) return HGT_ERR_INVALID_ARG;
    if (n == 0) return HGT_OK;
    k_gelu_bwd<<<nblk(n / 4, 256), 256, 0, (hipStream_t)stream>>>(dg, agg, out, n);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_mul_inplace(float* x, const float* m, int64_t n, void* stream) {
    if (!x || !m || n < 0) return HGT_ERR_INVALID_ARG;
    if (n == 0) return HGT_OK;
    k_mul_inplace<<<nblk(n, 256), 256, 0, (hipStream_t)stream>>>(x, m, n);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_softmax_bwd(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, const float* att,
                                    const float* d_att, const float* rho, int64_t ld_rho, float* d_logits, void* stream) {
    if (!plan || !att || !d_att || !rho || !d_logits || H <= 0) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    k_edge_softmax_bwd<<<nblk(E * H, 256), 256, 0, (hipStream_t)stream>>>(pv.edst, att, d_att, rho, ld_rho, d_logits, E, H);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_edge_gather_sorted(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, const float* by_edge_id,
                                      float* sorted, void* stream) {
    if (!plan || !by_edge_id || !sorted || H <= 0) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    k_gather_sorted<<<nblk(E * H, 256), 256, 0, (hipStream_t)stream>>>(pv.eid, by_edge_id, sorted, E, H);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_head_dot(const float* a, const float* b, int64_t n_rows, int32_t n_heads, int32_t dk_pad, float* out, void* stream) {
    if (!a || !b || !out || n_rows < 0 || n_heads <= 0 || dk_pad <= 0 || (dk_pad & 3) != 0) return HGT_ERR_INVALID_ARG;
    if (n_rows == 0) return HGT_OK;
    if (dk_pad > 256 || (dk_pad & (dk_pad - 1)) != 0) return HGT_ERR_INVALID_ARG;
    k_head_dot<<<nblk(n_rows * n_heads * (dk_pad / 4), 256), 256, 0, (hipStream_t)stream>>>(a, b, n_rows, n_heads, dk_pad, out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_typed_wgrad(const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* rows, const int32_t* group_off,
                               int32_t n_groups, int64_t n_rows, int32_t m, int32_t n_cols, float* out, int64_t out_group_stride,
                               void* stream) {
    if (!A || !B || !rows || !group_off || !out || n_groups <= 0 || n_rows < 0 || m <= 0 || n_cols <= 0) return HGT_ERR_INVALID_ARG;
    const int vecA = ((lda & 3) == 0 && ((uintptr_t)A & 15) == 0), vecB = ((ldb & 3) == 0 && ((uintptr_t)B & 15) == 0);   // 16 B row loads
    if (n_rows == 0) return HGT_OK;
    const int64_t chunks = (n_rows + WG_ROWS - 1) / WG_ROWS + n_groups;   // device-side group sizes: launch the upper bound
    dim3 grid((unsigned)chunks, (unsigned)((m + 63) / 64), (unsigned)((n_cols + 63) / 64));
    k_typed_wgrad<<<grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, rows, group_off, n_groups, m, n_cols, out, out_group_stride, vecA, vecB);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_typed_wgrad_bf16x3(const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* rows,
                                      const int32_t* group_off, int32_t n_groups, int64_t n_rows, int32_t m, int32_t n_cols, float* out,
                                      int64_t out_group_stride, float* colsum, int64_t colsum_group_stride, void* stream) {
    if (!A || !B || !rows || !group_off || !out || n_groups <= 0 || n_rows < 0 || m <= 0 || n_cols <= 0) return HGT_ERR_INVALID_ARG;
    if (n_rows == 0) return HGT_OK;
    const int n_mt = (m + WX_T - 1) / WX_T, n_nt = (n_cols + WX_T - 1) / WX_T;
    const int64_t chunks = (n_rows + WX_ROWS - 1) / WX_ROWS + n_groups;    // device-side group sizes: launch the upper bound
    if (chunks > 65535) return HGT_ERR_TOO_LARGE;
    dim3 grid((unsigned)(n_mt * n_nt), (unsigned)chunks);
    k_typed_wgrad_x3<<<grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, rows, group_off, n_groups, m, n_cols, n_mt, out, out_group_stride,
                                                          colsum, colsum_group_stride);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_typed_colsum(const float* A, int64_t lda, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                int64_t n_rows, int32_t m, float* out, int64_t out_group_stride, void* stream) {
    if (!A || !rows || !group_off || !out || n_groups <= 0 || n_rows < 0 || m <= 0) return HGT_ERR_INVALID_ARG;
    if (n_rows == 0) return HGT_OK;
    const int64_t waves = (n_rows + 255) / 256 + n_groups;
    k_typed_colsum<<<nblk(waves, 4), 256, 0, (hipStream_t)stream>>>(A, lda, rows, group_off, n_groups, m, out, out_group_stride);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_relation_outer(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad,
                                  const float* weights, const float* a_src, const float* rte_a, const float* b_dst, float* out,
                                  void* stream) {
    if (!plan || !a_src || !b_dst || !out || (E > 0 && !weights) || H <= 0 || 64 % H != 0 || dk_pad <= 0) return HGT_ERR_INVALID_ARG;
    if (E == 0) return HGT_OK;
    const int lph = 64 / H;
    if (dk_pad % lph != 0) return HGT_ERR_INVALID_ARG;
    HgtPlanView pv = hgt_plan_view(plan, N, E, T, R);
    // the per-lane accumulator block is VEC x DKP floats: split head groups until it fits 256 registers
    int vec = dk_pad / lph, l2 = lph;
    while (vec * dk_pad > 128 && vec > 1 && l2 * 2 <= 64) { vec /= 2; l2 *= 2; }
    int rc = dispatch_layout<LaunchOuter>(vec, l2, pv, weights, a_src, rte_a, b_dst, out, (int)R, (int)H, (hipStream_t)stream);
    if (rc != HGT_OK) return rc;
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
