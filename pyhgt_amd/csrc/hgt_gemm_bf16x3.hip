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
#include "hgt_split_common.h"
#include <algorithm>


namespace {

// W [n_groups][n_out][k] fp32 -> [g][pass][kchunk][plane][col tile 8][lane 64][8] bf16 (zero padded): the 8 bf16 of
// (col tile ct, lane l) are W[pass*256 + ct*32 + (l&31)][kchunk*16 + (l>>5)*8 .. +8] = one lane's B fragment.
// F16: fp16 hi / lo planes of W[g] * scale[g] (one power-of-two scale per group, k_group_scale; the image's tail holds its inverse)
template <bool F16>
__global__ void k_split_weights(const float* __restrict__ W, int64_t wgs, int n_groups, int k, int n_out, int n_pass, int n_kc,
                                unsigned short* __restrict__ out, const float* __restrict__ gscale) {
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
    unsigned short h, m;
    if constexpr (F16) {
        v *= gscale[g];
        const _Float16 hh = (_Float16)v;
        const _Float16 ll = (_Float16)(v - (float)hh);
        h = __builtin_bit_cast(unsigned short, hh);
        m = __builtin_bit_cast(unsigned short, ll);
    } else {
        h = bf16_rne(v);
        m = bf16_rne(v - bf16_to_f32(h));
    }
    const int64_t tile = (((int64_t)g * n_pass + pass) * n_kc + kc) * 2;
    const int within = (ct * 64 + l) * 8 + e;
    out[(tile + 0) * W_PLANE_ELEMS + within] = h;
    out[(tile + 1) * W_PLANE_ELEMS + within] = m;
}

// one workgroup per group: max |W[g]| -> tail[g] = inverse scale (read by the GEMM epilogues), tail[n_groups + g] = scale
__global__ __launch_bounds__(1024) void k_group_scale(const float* __restrict__ W, int64_t wgs, int64_t per_group, float* __restrict__ tail,
                                                       int n_groups) {
    __shared__ unsigned s_m[16];
    const int g = blockIdx.x;
    unsigned m = 0;
    for (int64_t i = threadIdx.x; i < per_group; i += 1024) m = max(m, __builtin_bit_cast(unsigned, fabsf(W[(int64_t)g * wgs + i])));
    m = wave_max_bits(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = max(m, s_m[w]);
        float sc, inv;
        f16_row_scale(m, sc, inv);
        tail[g] = inv;
        tail[n_groups + g] = sc;
    }
}

// Load one K panel of the 64-row x slab: whole rows, split into bf16 hi/mid planes ONCE, stored to LDS.
// Done in two halves of 4 float4 per thread to keep the live register set small (the kernel is capped at 128 VGPRs
// so that two workgroups fit a CU; an 8-deep version spilled ~230 B per lane to scratch = +2.9 GB of HBM traffic at c2).
template <int PROLOGUE, bool F16>
__device__ __forceinline__ void load_a_panel(int kp0, int tid, const int* s_rid, const float* __restrict__ x, int64_t ldx, int k,
                                             int vec_ok, unsigned char* sA, bool wait_readers, float* s_scale, float* s_inv,
                                             float* s_ratio, int* s_flag, bool first) {      // first: the row's first panel (F16: its scale starts here)
    if (wait_readers) __syncthreads();   // every wave is done reading the previous panel
    const int par = (kp0 / KP) & 1, epoch = kp0 / KP + 1;      // s_flag[par] == epoch: this panel lowered some row's scale (nothing to clear:
                                                              // a stale match -- the same panel of an earlier pass -- rescales by ratios of 1)
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
                } else {      // (four loads in flight, no branch between them: rows of 1169 floats are not 16-byte aligned)
                    const int rem = k - 1 - kk;
                    const float t1 = px[min(1, rem)], t2 = px[min(2, rem)], t3 = px[min(3, rem)];
                    a.x = px[0];
                    a.y = rem >= 1 ? t1 : 0.f;
                    a.z = rem >= 2 ? t2 : 0.f;
                    a.w = rem >= 3 ? t3 : 0.f;
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
            float scale = 1.0f;
            if constexpr (F16) {      // the wavefront holds the row's whole panel (64 lanes x 4 columns): running scale (see the kernel)
                float inv;
                f16_row_scale(wave_max_bits(abs_bits4(av[j])), scale, inv);
                if (!first) {
                    const float cur = s_scale[r];
                    if (scale < cur) {      // (wave-uniform) this panel is larger than everything before it
                        if ((tid & 63) == 0) { s_ratio[r] = scale / cur; s_flag[par] = epoch; }
                    } else {
                        scale = cur;
                        inv = s_inv[r];
                        if ((tid & 63) == 0) s_ratio[r] = 1.0f;
                    }
                }
                if ((tid & 63) == 0) { s_scale[r] = scale; s_inv[r] = inv; }
            }
            split4_t<F16>(av[j], scale, hi, mid);
            *reinterpret_cast<uint2*>(sA + r * A_STRIDE + cb) = hi;
            *reinterpret_cast<uint2*>(sA + A_PLANE + r * A_STRIDE + cb) = mid;
        }
    }
    __syncthreads();   // panel visible
}

// epilogue of one 256-column pass.  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Narrow (4 B per lane) stores are issue-bound, so each group of 4 registers (4 consecutive rows, one column per lane)
// is transposed inside the lane quad: afterwards a lane holds 4 consecutive COLUMNS of one row and writes one 16 B
// store (one wave instruction = 8 rows x 128 B) -- 4x fewer store instructions.
__device__ __forceinline__ void store_pass(f32x16 (&acc)[2], int pass, int wave, int lane, int g, int n_out, int nrows, int row0,
                                           const int* s_rid, const float* __restrict__ bias, int64_t bgs, float* __restrict__ out0,
                                           float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos,
                                           const float* s_inv, float winv, int act_tanh = 0) {      // s_inv == nullptr: no operand scales (bf16 split)
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
                const float sc = s_inv ? s_inv[rt] * winv : 1.0f;
                float4 o4 = make_float4(v0 * sc + b4.x, v1 * sc + b4.y, v2 * sc + b4.z, v3 * sc + b4.w);
                if (act_tanh) o4 = make_float4(hgt_tanh(o4.x), hgt_tanh(o4.y), hgt_tanh(o4.z), hgt_tanh(o4.w));
                *reinterpret_cast<float4*>(ob + orow * block_cols + cc) = o4;
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

// Fused node update (conv.py:129-133) as the epilogue of the a_linear GEMM, single 256-column pass:
//   y = (acc + b) * sigmoid(skip[t]) + x * (1 - sigmoid(skip[t]));  out = LayerNorm_t(y)  (two-pass mean/variance)
// A row's 256 columns live in 8 waves x 8 lanes; partial sums meet in a small LDS table (the A slab is dead by then).
__device__ __forceinline__ void store_pass_update(f32x16 (&acc)[2], int wave, int lane, int g, int n_out, int nrows, const int* s_rid,
                                                  const float* __restrict__ bias, int64_t bgs, float* __restrict__ out,
                                                  const UpdateArgs& u, float* s_red, const float* s_inv, float winv) {
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
            const float sc = s_inv ? s_inv[rt] * winv : 1.0f;
            y[q][0] = col_ok ? (v0 * sc + b4.x) * alpha + xv.x * (1.0f - alpha) : 0.0f;
            y[q][1] = col_ok ? (v1 * sc + b4.y) * alpha + xv.y * (1.0f - alpha) : 0.0f;
            y[q][2] = col_ok ? (v2 * sc + b4.z) * alpha + xv.z * (1.0f - alpha) : 0.0f;
            y[q][3] = col_ok ? (v3 * sc + b4.w) * alpha + xv.w * (1.0f - alpha) : 0.0f;
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

template <int PROLOGUE, bool UPD, bool F16, int NSTG = 2>
__global__ __launch_bounds__(512, UPD ? 2 : 4) void k_typed_linear_split(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
    int n_groups, int k, int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int vec_ok,
    UpdateArgs upd, int pass_split_act) {
    const int pass_split = pass_split_act & 0xffff, act_tanh = pass_split_act >> 16;      // (bit 16: tanh on the output, the GNN's adapter)
    __shared__ __attribute__((aligned(16))) unsigned char sA[2 * A_PLANE];        // [plane][64][528]
    __shared__ int s_rid[BM];
    __shared__ float s_scale[F16 ? BM : 1], s_inv[F16 ? BM : 1];                   // fp16 split: row scales and their inverses
    __shared__ __attribute__((aligned(16))) float s_ratio[F16 ? BM : 1];          // ... the factor a panel lowered a row's scale by
    __shared__ int s_flag[2];                                                      // ... "some row was lowered", by panel parity

    // pass_split = n_pass (small problems: fewer row tiles than CUs): a workgroup owns ONE 256-column pass of a row tile, so that
    // tiles x passes workgroups share the work (sampled sub-graphs of a few thousand nodes: 50-64 row tiles for 256 CUs);
    // pass_split = 1: a workgroup owns a row tile and walks all passes
    // (an XCD-aware work order like k_tile_linear_stream's -- (type, pass, row tile) in eight contiguous chunks -- measured neutral for
    //  this kernel's 64 x 256 tiles, r6: 37.1 - 38.4 vs 37.5 us at 4 096 x 400 -> 1 536)
    const int slot = blockIdx.x / pass_split;
    const int pass_only = (pass_split > 1) ? (int)(blockIdx.x % pass_split) : -1;
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
    float winv = 1.0f;                               // inverse of the group's weight scale: the image's tail (hgt_split_weights_f16)
    if constexpr (F16) {
        winv = reinterpret_cast<const float*>(wsplit + (int64_t)n_groups * total * 2 * W_PLANE_ELEMS)[g];
        // several K panels: ONE scale per row without a second pass over x (round 6; rounds 3-5 took the row maximum in a pre-pass:
        // +8 us at K = 512, +21 us at K = 1169 on a 4 000-row batch) -- RUNNING scales: a panel is split with the scale its row has
        // so far, lowered first when the panel's own maximum needs it; the accumulators of a row whose scale was lowered are multiplied
        // by the (power-of-two, exact) ratio before the panel's products are added.  Absolute resolution = that of the row's maximum so
        // far: the whole-row scale's accuracy when the largest panel comes first, better otherwise.
    }

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // B fragment stream: k-chunk T of the flattened (pass, k-chunk) sequence, prefetched HGT_NSTAGE ahead in named register stages.
    // n_kc is a multiple of 4 (hgt_split_weights zero-pads K to a multiple of 64, the A slab is zero-filled
    // beyond k), so the 4-step body needs no guards and a pass boundary always falls between bodies.
    // NSTG = 2: measured at c2 (1.83 ms) against 4 (1.88 ms): fewer live registers -> no spills at the 128-VGPR cap.
    // NSTG = 4 (latency regime, launched when the grid is smaller than the chip: one workgroup = one dependent chain of B-fragment
    // round trips to L2, registers are not what limits it): four k-chunks in flight, 256-VGPR budget.
    bf16x8 s0h, s0m, s1h, s1m, s2h, s2m, s3h, s3m;
#define HGT_LOAD_STAGE(S, T)                                                                          \
    {                                                                                                 \
        const unsigned short* t_ = wfrag + (int64_t)min((T), total - 1) * 2 * W_PLANE_ELEMS;          \
        s##S##h = *reinterpret_cast<const bf16x8*>(t_);                                               \
        s##S##m = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                               \
    }
    const int pass_lo = (pass_only >= 0) ? pass_only : 0, pass_hi = (pass_only >= 0) ? pass_only + 1 : n_pass;
    HGT_LOAD_STAGE(0, pass_lo * n_kc)
    HGT_LOAD_STAGE(1, pass_lo * n_kc + 1)
    if constexpr (NSTG == 4) {
        HGT_LOAD_STAGE(2, pass_lo * n_kc + 2)
        HGT_LOAD_STAGE(3, pass_lo * n_kc + 3)
    }

    load_a_panel<PROLOGUE, F16>(0, tid, s_rid, x, ldx, k, vec_ok, sA, false, s_scale, s_inv, s_ratio, s_flag, true);

#define HGT_STEP(S, T, KCP)                                                                                        \
    {                                                                                                              \
        const int ao = frow * A_STRIDE + ((KCP) * KC + khalf * 8) * 2;                                             \
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(sA + ao);                                              \
        const bf16x8 am0 = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao);                                    \
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(sA + ao + 32 * A_STRIDE);                              \
        const bf16x8 am1 = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao + 32 * A_STRIDE);                    \
        const bf16x8 bh = s##S##h, bm = s##S##m;                                                                   \
        HGT_LOAD_STAGE(S, (T) + NSTG)                                                                              \
        /* small terms first, hi*hi last; the two accumulators alternate */                                        \
        acc[0] = mfma32_t<F16>(am0, bh, acc[0]);                                \
        acc[1] = mfma32_t<F16>(am1, bh, acc[1]);                                \
        acc[0] = mfma32_t<F16>(ah0, bm, acc[0]);                                \
        acc[1] = mfma32_t<F16>(ah1, bm, acc[1]);                                \
        acc[0] = mfma32_t<F16>(ah0, bh, acc[0]);                                \
        acc[1] = mfma32_t<F16>(ah1, bh, acc[1]);                                \
    }

    for (int pass = pass_lo; pass < pass_hi; ++pass) {
        for (int panel = 0; panel < n_panel; ++panel) {
            if (n_panel > 1 && (pass != pass_lo || panel != 0)) load_a_panel<PROLOGUE, F16>(panel * KP, tid, s_rid, x, ldx, k, vec_ok, sA, true, s_scale, s_inv, s_ratio, s_flag, panel == 0);
            if constexpr (F16) {
                if (panel > 0 && s_flag[panel & 1] == panel + 1) {      // (workgroup-uniform) some row's scale was lowered by this panel
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {      // C/D layout: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
                            const float4 rv = *reinterpret_cast<const float4*>(&s_ratio[j * 32 + 8 * q + 4 * khalf]);
                            acc[j][4 * q] *= rv.x; acc[j][4 * q + 1] *= rv.y; acc[j][4 * q + 2] *= rv.z; acc[j][4 * q + 3] *= rv.w;
                        }
                }
            }
            const int nkc_p = min(KP / KC, n_kc - panel * (KP / KC));
            const int tbase = pass * n_kc + panel * (KP / KC);
            for (int kq = 0; kq < nkc_p; kq += 4) {
                if constexpr (NSTG == 4) {
                    HGT_STEP(0, tbase + kq, kq)
                    HGT_STEP(1, tbase + kq + 1, kq + 1)
                    HGT_STEP(2, tbase + kq + 2, kq + 2)
                    HGT_STEP(3, tbase + kq + 3, kq + 3)
                } else {
                    HGT_STEP(0, tbase + kq, kq)
                    HGT_STEP(1, tbase + kq + 1, kq + 1)
                    HGT_STEP(0, tbase + kq + 2, kq + 2)
                    HGT_STEP(1, tbase + kq + 3, kq + 3)
                }
            }
        }
        if constexpr (UPD) {
            __syncthreads();   // every wave left the MFMA loop: the A slab can be reused as the reduction table
            store_pass_update(acc, wave, lane, g, n_out, nrows, s_rid, bias, bgs, out0, upd, reinterpret_cast<float*>(sA),
                              F16 ? s_inv : nullptr, winv);
        } else {
            store_pass(acc, pass, wave, lane, g, n_out, nrows, row0, s_rid, bias, bgs, out0, out1, out2, block_cols, by_pos,
                       F16 ? s_inv : nullptr, winv, act_tanh);
        }
    }
#undef HGT_STEP
#undef HGT_LOAD_STAGE
}

// =============================================================================================
// a_linear + gated skip + LayerNorm for rows of 257..512 output columns and K <= 512 (n_hid 400 / 512: the reference's published
// ogbn-mag width, ogbn-mag/train_ogbn_mag.py:36-40) in ONE kernel (round 5).  Before: k_typed_linear_split wrote the 512-column
// product (2 passes x 2 K panels, the A panels loaded once per pass) and hgt_node_update read it back -- 4 d bytes per node written
// and read for nothing and a second K-panel load per tile (d512_h8: 0.98 + 0.78 ms).
// Here both 256-column passes keep their accumulators (64 registers), the K panels are the OUTER loop (each loaded once), and
// the LayerNorm statistics run over the two passes' columns together in the epilogue.  The B fragments of the (panel, pass) segments
// are not contiguous in the image: a prefetch cursor walks them in the order they are used, so a segment's first k-chunk is as
// much in flight as any other.
// =============================================================================================
__device__ __forceinline__ void store_update_wide(f32x16 (&acc)[2][2], int wave, int lane, int g, int n_out, int nrows, const int* s_rid,
                                                  const float* __restrict__ bias, int64_t bgs, float* __restrict__ out,
                                                  const UpdateArgs& u, float* s_red, const float* s_inv, float winv) {
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));      // (the address arithmetic of this section stays out of the MFMA loop's register budget)
    const int col0 = wave * 32 + ((lane_e & 31) >> 2) * 4;      // pass p: col0 + 256 p
    const float alpha = 1.0f / (1.0f + expf(-u.skip[g]));
    const bool o1 = lane_e & 1, o2 = lane_e & 2;
    const float inv_n = 1.0f / (float)n_out;
    float4 b4[2], w4[2], c4[2];
    bool col_ok[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int col = col0 + p * BNP;
        col_ok[p] = col < n_out;
        b4[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        w4[p] = make_float4(1.f, 1.f, 1.f, 1.f);
        c4[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok[p] && bias) b4[p] = *reinterpret_cast<const float4*>(bias + (int64_t)g * bgs + col);
        if (u.use_norm && col_ok[p]) {
            w4[p] = *reinterpret_cast<const float4*>(u.lnw + (int64_t)g * n_out + col);
            c4[p] = *reinterpret_cast<const float4*>(u.lnb + (int64_t)g * n_out + col);
        }
    }
    // the two 32-row halves one after the other (half the live registers)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float y[2][4][4];
        int64_t orow[4];
        int rts[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rt = j * 32 + (lane_e & 3) + 8 * q + 4 * (lane_e >> 5);
            rts[q] = rt;
            orow[q] = (rt < nrows) ? (int64_t)s_rid[rt] : -1;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = acc[p][j][4 * q], v1 = acc[p][j][4 * q + 1], v2 = acc[p][j][4 * q + 2], v3 = acc[p][j][4 * q + 3];
                quad_transpose(v0, v1, v2, v3, o1, o2);
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (col_ok[p] && orow[q] >= 0) xv = *reinterpret_cast<const float4*>(u.xs + orow[q] * u.ldxs + col0 + p * BNP);
                const float sc = s_inv ? s_inv[rts[q]] * winv : 1.0f;
                y[p][q][0] = col_ok[p] ? (v0 * sc + b4[p].x) * alpha + xv.x * (1.0f - alpha) : 0.0f;
                y[p][q][1] = col_ok[p] ? (v1 * sc + b4[p].y) * alpha + xv.y * (1.0f - alpha) : 0.0f;
                y[p][q][2] = col_ok[p] ? (v2 * sc + b4[p].z) * alpha + xv.z * (1.0f - alpha) : 0.0f;
                y[p][q][3] = col_ok[p] ? (v3 * sc + b4[p].w) * alpha + xv.w * (1.0f - alpha) : 0.0f;
            }
        }
        if (u.use_norm) {
            // mean: a row's columns live in 8 waves x 2 passes x 8 lanes
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float ps = strided8_sum(y[0][q][0] + y[0][q][1] + y[0][q][2] + y[0][q][3] + y[1][q][0] + y[1][q][1] + y[1][q][2] + y[1][q][3]);
                if (((lane_e & 31) >> 2) == 0) s_red[rts[q] * 8 + wave] = ps;
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
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float ss = 0.0f;
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[p][q][e] -= mean[q];      // centred from here on
                        if (col_ok[p]) ss += y[p][q][e] * y[p][q][e];
                    }
                const float ps = strided8_sum(ss);
                if (((lane_e & 31) >> 2) == 0) s_red[rts[q] * 8 + wave] = ps;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(&s_red[rts[q] * 8]);
                const float4 b = *reinterpret_cast<const float4*>(&s_red[rts[q] * 8 + 4]);
                const float rstd = rsqrtf((a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * inv_n + 1e-5f);
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    if (col_ok[p] && orow[q] >= 0)
                        *reinterpret_cast<float4*>(out + orow[q] * n_out + col0 + p * BNP) =
                            make_float4(y[p][q][0] * rstd * w4[p].x + c4[p].x, y[p][q][1] * rstd * w4[p].y + c4[p].y,
                                        y[p][q][2] * rstd * w4[p].z + c4[p].z, y[p][q][3] * rstd * w4[p].w + c4[p].w);
            }
            __syncthreads();   // s_red is reused by the second half
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    if (col_ok[p] && orow[q] >= 0)
                        *reinterpret_cast<float4*>(out + orow[q] * n_out + col0 + p * BNP) = make_float4(y[p][q][0], y[p][q][1], y[p][q][2], y[p][q][3]);
        }
    }
}

template <bool F16>
__global__ __launch_bounds__(512, 1) void k_typed_linear_update_wide(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off, int n_groups, int k,
    int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs, float* __restrict__ out, int vec_ok,
    UpdateArgs upd) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[2 * A_PLANE];        // [plane][64][528]
    __shared__ int s_rid[BM];
    __shared__ float s_scale[F16 ? BM : 1], s_inv[F16 ? BM : 1];
    __shared__ __attribute__((aligned(16))) float s_ratio[F16 ? BM : 1];          // running row scales: see k_typed_linear_split
    __shared__ int s_flag[2];

    // (an XCD-aware order -- the row tiles cut into eight contiguous chunks, one per XCD, so that an XCD streams about one type's
    //  1 MB image at a time -- measured slower at d512_h8: 1.34 vs 1.25 ms, r6)
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

    const int n_kc = ((k + KC - 1) / KC + 3) & ~3;      // k-chunks of the image (a multiple of 4), <= 32 here
    const int n_panel = (k + KP - 1) / KP;              // 1 or 2
    const int total = 2 * n_kc;                         // two passes
    const int lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, khalf = lane >> 5;
    const unsigned short* __restrict__ wfrag = wsplit + (int64_t)g * total * 2 * W_PLANE_ELEMS + (wave * 64 + lane) * 8;
    float winv = 1.0f;
    if constexpr (F16) {
        winv = reinterpret_cast<const float*>(wsplit + (int64_t)n_groups * total * 2 * W_PLANE_ELEMS)[g];
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][j][r] = 0.0f;

    // prefetch cursor over the (panel, pass, k-chunk) order of use; beyond the end it re-requests the last tile
    int pf_pass = 0, pf_kc = 0, pf_beg = 0, pf_end = min(KP / KC, n_kc);
    // (two k-chunks of B fragments in flight; four measured SLOWER, 1.36 vs 1.27 ms at d512_h8: every tile streams the whole 1 MB image
    //  of its type through the L2 -- 8 GB per launch next to 3 GB of HBM traffic -- and more requests in flight only queue)
    bf16x8 s0h, s0m, s1h, s1m;
#define HGT_WLOAD(S)                                                                                  \
    {                                                                                                 \
        const unsigned short* t_ = wfrag + (int64_t)(pf_pass * n_kc + min(pf_kc, n_kc - 1)) * 2 * W_PLANE_ELEMS; \
        s##S##h = *reinterpret_cast<const bf16x8*>(t_);                                               \
        s##S##m = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                               \
        /* (branches on purpose: as scalar selects the same cursor measured 1.45 instead of 1.25 ms at d512_h8, r05) */ \
        if (++pf_kc == pf_end) {                                                                      \
            if (pf_pass == 0) { pf_pass = 1; pf_kc = pf_beg; }                                        \
            else if (pf_end < n_kc) { pf_pass = 0; pf_beg = pf_end; pf_kc = pf_beg; pf_end = n_kc; }  \
            else pf_kc = n_kc;                                                                        \
        }                                                                                             \
    }
#define HGT_WSTEP(S, P, KCP)                                                                                       \
    {                                                                                                              \
        const int ao = frow * A_STRIDE + ((KCP) * KC + khalf * 8) * 2;                                             \
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(sA + ao);                                              \
        const bf16x8 am0 = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao);                                    \
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(sA + ao + 32 * A_STRIDE);                              \
        const bf16x8 am1 = *reinterpret_cast<const bf16x8*>(sA + A_PLANE + ao + 32 * A_STRIDE);                    \
        const bf16x8 bh = s##S##h, bm = s##S##m;                                                                   \
        HGT_WLOAD(S)                                                                                               \
        acc[P][0] = mfma32_t<F16>(am0, bh, acc[P][0]);                          \
        acc[P][1] = mfma32_t<F16>(am1, bh, acc[P][1]);                          \
        acc[P][0] = mfma32_t<F16>(ah0, bm, acc[P][0]);                          \
        acc[P][1] = mfma32_t<F16>(ah1, bm, acc[P][1]);                          \
        acc[P][0] = mfma32_t<F16>(ah0, bh, acc[P][0]);                          \
        acc[P][1] = mfma32_t<F16>(ah1, bh, acc[P][1]);                          \
    }
    HGT_WLOAD(0)
    HGT_WLOAD(1)
    for (int panel = 0; panel < n_panel; ++panel) {
        load_a_panel<0, F16>(panel * KP, tid, s_rid, x, ldx, k, vec_ok, sA, panel > 0, s_scale, s_inv, s_ratio, s_flag, panel == 0);
        if constexpr (F16) {
            if (panel > 0 && s_flag[panel & 1] == panel + 1) {      // (workgroup-uniform) the second panel lowered some row's scale
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 rv = *reinterpret_cast<const float4*>(&s_ratio[j * 32 + 8 * q + 4 * khalf]);
                            acc[p][j][4 * q] *= rv.x; acc[p][j][4 * q + 1] *= rv.y; acc[p][j][4 * q + 2] *= rv.z; acc[p][j][4 * q + 3] *= rv.w;
                        }
            }
        }
        const int nkc_p = min(KP / KC, n_kc - panel * (KP / KC));
        for (int kq = 0; kq < nkc_p; kq += 4) {
            HGT_WSTEP(0, 0, kq)
            HGT_WSTEP(1, 0, kq + 1)
            HGT_WSTEP(0, 0, kq + 2)
            HGT_WSTEP(1, 0, kq + 3)
        }
        for (int kq = 0; kq < nkc_p; kq += 4) {
            HGT_WSTEP(0, 1, kq)
            HGT_WSTEP(1, 1, kq + 1)
            HGT_WSTEP(0, 1, kq + 2)
            HGT_WSTEP(1, 1, kq + 3)
        }
    }
#undef HGT_WSTEP
#undef HGT_WLOAD
    __syncthreads();   // every wave left the MFMA loops: the A slab is reused as the reduction table
    store_update_wide(acc, wave, lane, g, n_out, nrows, s_rid, bias, bgs, out, upd, reinterpret_cast<float*>(sA), F16 ? s_inv : nullptr, winv);
}

// =============================================================================================
// Persistent producer / consumer variant (k <= 256: the whole K extent is one LDS slab).
//
// Why: the kernel above has at most two workgroups per CU and every one of them runs its phases
// back to back (row ids -> 64 KB slab from HBM -> MFMA loop -> epilogue), so HBM idles while a
// workgroup computes and the matrix cores idle while it loads.  A next-tile prefetch through
// registers does not help: s_waitcnt vmcnt retires IN ORDER, so the first B-fragment wait of the
// MFMA loop would also wait for the older slab loads.  The loads therefore move to their own waves:
//   * one workgroup per CU (grid = #CUs), looping over 64-row tiles  t = blockIdx.x, += gridDim.x;
//   * waves 0..7 (two per SIMD) are CONSUMERS: the MFMA loop + epilogue of tile i out of slab[i&1];
//   * waves 8..11 (one per SIMD) are PRODUCERS: they keep the 64 KB of tile i+2 in flight in
//     registers, and split + store tile i+1 into slab[(i+1)&1] while the consumers work on tile i;
//   * one workgroup barrier per tile hands slab[(i+1)&1] over (plus two inside the LayerNorm
//     epilogue of the fused update; the producers simply arrive at those as well).
// LDS: 2 x 66 KB slabs + row ids + the LayerNorm tables = 137 KB.  VGPR budget 168 (3 waves/SIMD).
// =============================================================================================
constexpr int PC_CONS = 8, PC_PROD = 4, PC_THREADS = 64 * (PC_CONS + PC_PROD);
constexpr int PC_AREGS = BM * (KP / 4) / (64 * PC_PROD);   // float4 per producer lane per tile = 16

__device__ __forceinline__ bool tile_lookup(int t, const int32_t* __restrict__ group_off, int n_groups, int& g, int& row0, int& nrows) {
    int before = 0;
    for (g = 0; g < n_groups; ++g) {
        const int gb = group_off[g], ge = group_off[g + 1];
        const int nt = (ge - gb + BM - 1) / BM;
        if (t < before + nt) {
            row0 = gb + (t - before) * BM;
            nrows = min(BM, ge - row0);
            return true;
        }
        before += nt;
    }
    return false;
}

// producer: request the rows of one tile (wave pw fetches rows pw, pw+4, ..., one coalesced 1 KB row per instruction).
// Fast path: 16 unconditional back-to-back 16 B loads (absent rows read row 0 and are zeroed in pc_commit); written
// with per-row conditions hipcc lowers the whole thing to 4 B loads behind branches (measured: 47k cycles per tile).
template <int PROLOGUE>
__device__ __forceinline__ void pc_issue(float4 (&a)[PC_AREGS], int& v_rid, int pw, int lane, int row0, int nrows,
                                         const int32_t* __restrict__ rows, const float* __restrict__ x, int64_t ldx, int k, int vec_ok) {
    const int myrow = pw + PC_PROD * lane;   // lanes 0..15 carry the 16 row ids of this wave
    v_rid = (lane < PC_AREGS && myrow < nrows) ? rows[row0 + myrow] : -1;
    const int kk = lane * 4;
    if constexpr (PROLOGUE == 2) {
        // x holds rows in the 24-bit transport format of the multi-GPU exchange (hgt_gather_rows_c24: 4 values = 3 dwords, ldx =
        // 3 k / 4 dwords per row; k % 4 == 0 checked by the launcher): one 12-byte load per lane, decoded in pc_commit
        // (rows beyond the tile re-read the tile's FIRST row -- always present -- not row 0: x is the wire buffer shifted back by the
        //  chunk's first local row id (hgt_conv_forward stage 2), so "row 0" lies far outside any allocation)
        const int rid_safe = rows[row0];
        if (kk < k) {
#pragma unroll
            for (int j = 0; j < PC_AREGS; ++j) {
                const int rj = __builtin_amdgcn_readlane(v_rid, j);
                const int rid = rj < 0 ? rid_safe : rj;
                const float* px = x + (int64_t)rid * ldx + lane * 3;
                a[j] = make_float4(px[0], px[1], px[2], 0.0f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PC_AREGS; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    if (vec_ok) {   // k % 4 == 0: a lane is entirely inside or entirely outside the row
        if (kk < k) {
#pragma unroll
            for (int j = 0; j < PC_AREGS; ++j) {
                const int rid = max(__builtin_amdgcn_readlane(v_rid, j), 0);
                a[j] = *reinterpret_cast<const float4*>(x + (int64_t)rid * ldx + kk);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PC_AREGS; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < PC_AREGS; ++j) {
        const int rid = __builtin_amdgcn_readlane(v_rid, j);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rid >= 0 && kk < k) {
            const float* px = x + (int64_t)rid * ldx + kk;
            v.x = px[0];
            if (kk + 1 < k) v.y = px[1];
            if (kk + 2 < k) v.z = px[2];
            if (kk + 3 < k) v.w = px[3];
        }
        a[j] = v;
    }
}

template <int PROLOGUE, bool F16>
__device__ __forceinline__ void pc_commit(const float4 (&a)[PC_AREGS], int v_rid, int pw, int lane, unsigned char* slab, int* rid_out,
                                          float* rinv_out) {
    if (lane < PC_AREGS) rid_out[pw + PC_PROD * lane] = v_rid;
#pragma unroll
    for (int j = 0; j < PC_AREGS; ++j) {
        float4 v = a[j];
        if constexpr (PROLOGUE == 2) {      // 3 dwords of the 24-bit format -> 4 floats (value = 24 bits << 8: hgt_update.hip)
            const unsigned w0 = __builtin_bit_cast(unsigned, v.x), w1 = __builtin_bit_cast(unsigned, v.y), w2 = __builtin_bit_cast(unsigned, v.z);
            v.x = __builtin_bit_cast(float, w0 << 8);
            v.y = __builtin_bit_cast(float, ((w0 >> 24) | (w1 << 8)) << 8);
            v.z = __builtin_bit_cast(float, ((w1 >> 16) | (w2 << 16)) << 8);
            v.w = __builtin_bit_cast(float, w2 & 0xFFFFFF00u);
        }
        if (__builtin_amdgcn_readlane(v_rid, j) < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);   // row beyond the tile
        if (PROLOGUE == 1) { v.x = gelu_erf_(v.x); v.y = gelu_erf_(v.y); v.z = gelu_erf_(v.z); v.w = gelu_erf_(v.w); }
        float scale = 1.0f;
        if constexpr (F16) {      // the wavefront holds the whole row: its maximum is a wave reduction, its scale wave-uniform
            float inv;
            f16_row_scale(wave_max_bits(abs_bits4(v)), scale, inv);
            if (lane == 0) rinv_out[pw + PC_PROD * j] = inv;
        }
        uint2 hi, mid;
        split4_t<F16>(v, scale, hi, mid);
        unsigned char* p = slab + (pw + PC_PROD * j) * A_STRIDE + lane * 8;
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + A_PLANE) = mid;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Output rows are written with stores the compiler does not see (inline asm).  Reason: a CU retires only ~10 B/clk of
// stores (measured: 64 KB per pass = 6.5-9k cycles of epilogue with the matrix cores idle), so the stores of pass p are
// spread over the MFMA loop of pass p+1, one every other k-chunk.  With visible stores pending, hipcc must treat
// vmcnt as unordered and turns every B-fragment wait of that loop into vmcnt(0).  With hidden stores it keeps emitting
// vmcnt(N), N = younger LOADS -- still sufficient: loads return in order, so a pending target load implies N+1
// pending loads, i.e. counter > N.  Pending stores only make the wait a little stricter than necessary.
__device__ __forceinline__ void hidden_store16(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" : : "v"(p), "v"(v));
}

// workgroup barrier that orders LDS only (a __syncthreads() would also drain every outstanding global load AND store)
__device__ __forceinline__ void pc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct PendingRows {     // one pass worth of finished output of this lane: 8 x (row, 4 consecutive columns)
    f32x4 v[8];
    float* base;         // column offset already applied; nullptr = this lane's columns are out of range
    int ld;              // row stride in floats
    // the output row of register group u is looked up again when the store is issued (8 fewer live registers):
    const int* rid;      // LDS row-id table of the tile the rows belong to (three tables rotate, see s_rid)
    int row0, nrows, by_pos;
    __device__ __forceinline__ int row(int u, int lane) const {
        const int rt = (u >> 2) * 32 + (lane & 3) + 8 * (u & 3) + 4 * (lane >> 5);
        return (base != nullptr && rt < nrows) ? (by_pos ? row0 + rt : rid[rt]) : -1;
    }
};

// Fused node update (conv.py:129-133), both 32-row halves at once, two barriers:
//   y = (acc + b) * sigmoid(skip[t]) + x * (1 - sigmoid(skip[t]));  out = LayerNorm_t(y)  (two-pass mean / variance)
__device__ __forceinline__ void pc_store_update(f32x16 (&acc)[2], int wave, int lane, int g, int n_out, int nrows, const int* s_rid,
                                                const float* __restrict__ bias, int64_t bgs, float* __restrict__ out,
                                                const UpdateArgs& u, float* s_sum, float* s_var, PendingRows& pr,
                                                const float* s_inv, float winv) {
    const int col = wave * 32 + ((lane & 31) >> 2) * 4;
    const bool col_ok = col < n_out;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok && bias) b4 = *reinterpret_cast<const float4*>(bias + (int64_t)g * bgs + col);
    const float alpha = 1.0f / (1.0f + expf(-u.skip[g]));
    const bool o1 = lane & 1, o2 = lane & 2;
    const float inv_n = 1.0f / (float)n_out;
    float y[8][4];
    // row of register group jq = (half, q): rt0 + 32*half + 8*q; its node id is re-read from LDS where needed
    // (keeping the eight ids in registers spills at the 168-VGPR cap)
    const int rt0 = (lane & 3) + 4 * (lane >> 5);
#define PC_OROW(JQ) ((rt0 + 32 * ((JQ) >> 2) + 8 * ((JQ)&3)) < nrows ? s_rid[rt0 + 32 * ((JQ) >> 2) + 8 * ((JQ)&3)] : -1)
    // the skip rows in two groups of four 16 B loads (all eight at once costs 16 more live registers: spills at the 168 cap)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float4 xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jq = half * 4 + q;
            xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int orow = PC_OROW(jq);
            if (col_ok && orow >= 0) xv[q] = *reinterpret_cast<const float4*>(u.xs + (int64_t)orow * u.ldxs + col);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jq = half * 4 + q;
            float v0 = acc[half][4 * q], v1 = acc[half][4 * q + 1], v2 = acc[half][4 * q + 2], v3 = acc[half][4 * q + 3];
            quad_transpose(v0, v1, v2, v3, o1, o2);
            const float sc = s_inv ? s_inv[rt0 + 32 * (jq >> 2) + 8 * (jq & 3)] * winv : 1.0f;
            y[jq][0] = col_ok ? (v0 * sc + b4.x) * alpha + xv[q].x * (1.0f - alpha) : 0.0f;
            y[jq][1] = col_ok ? (v1 * sc + b4.y) * alpha + xv[q].y * (1.0f - alpha) : 0.0f;
            y[jq][2] = col_ok ? (v2 * sc + b4.z) * alpha + xv[q].z * (1.0f - alpha) : 0.0f;
            y[jq][3] = col_ok ? (v3 * sc + b4.w) * alpha + xv[q].w * (1.0f - alpha) : 0.0f;
        }
    }
    if (!u.use_norm) {
#pragma unroll
        for (int jq = 0; jq < 8; ++jq) {
            pr.v[jq] = f32x4{y[jq][0], y[jq][1], y[jq][2], y[jq][3]};
        }
        pr.base = col_ok ? out + col : nullptr;
        pr.ld = n_out;
        pc_barrier();
        pc_barrier();
        return;
    }
    // a row's columns live in 8 waves x 8 lanes: lane-strided sums, then one table entry per (row, wave)
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
        const int rt = (jq >> 2) * 32 + (lane & 3) + 8 * (jq & 3) + 4 * (lane >> 5);
        const float ps = strided8_sum(y[jq][0] + y[jq][1] + y[jq][2] + y[jq][3]);
        if (((lane & 31) >> 2) == 0) s_sum[rt * 8 + wave] = ps;
    }
    pc_barrier();
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
        const int rt = (jq >> 2) * 32 + (lane & 3) + 8 * (jq & 3) + 4 * (lane >> 5);
        const float4 a = *reinterpret_cast<const float4*>(&s_sum[rt * 8]);
        const float4 b = *reinterpret_cast<const float4*>(&s_sum[rt * 8 + 4]);
        const float mean = (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * inv_n;
        y[jq][0] -= mean; y[jq][1] -= mean; y[jq][2] -= mean; y[jq][3] -= mean;   // y is centred from here on
        float d0 = y[jq][0], d1 = y[jq][1], d2 = y[jq][2], d3 = y[jq][3];
        if (!col_ok) d0 = d1 = d2 = d3 = 0.0f;
        const float ps = strided8_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
        if (((lane & 31) >> 2) == 0) s_var[rt * 8 + wave] = ps;
    }
    pc_barrier();
    float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok) {   // loaded this late on purpose: 8 fewer registers live across the two reductions
        w4 = *reinterpret_cast<const float4*>(u.lnw + (int64_t)g * n_out + col);
        c4 = *reinterpret_cast<const float4*>(u.lnb + (int64_t)g * n_out + col);
    }
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
        const int rt = (jq >> 2) * 32 + (lane & 3) + 8 * (jq & 3) + 4 * (lane >> 5);
        const float4 a = *reinterpret_cast<const float4*>(&s_var[rt * 8]);
        const float4 b = *reinterpret_cast<const float4*>(&s_var[rt * 8 + 4]);
        const float rstd = rsqrtf((a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w) * inv_n + 1e-5f);
        pr.v[jq] = f32x4{y[jq][0] * rstd * w4.x + c4.x, y[jq][1] * rstd * w4.y + c4.y, y[jq][2] * rstd * w4.z + c4.z,
                         y[jq][3] * rstd * w4.w + c4.w};
    }
    pr.base = col_ok ? out + col : nullptr;
    pr.ld = n_out;
}
#undef PC_OROW

// one 256-column pass of the plain linear layer: bias, 4x4 quad transpose (see store_pass), rows parked in PendingRows
__device__ __forceinline__ void pc_stage_pass(f32x16 (&acc)[2], int pass, int wave, int lane, int g, int n_out, int nrows, int row0,
                                              const int* s_rid, const float* __restrict__ bias, int64_t bgs, float* __restrict__ out0,
                                              float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos,
                                              PendingRows& pr, const float* s_inv, float winv) {
    const int col = pass * BNP + wave * 32 + ((lane & 31) >> 2) * 4;
    const bool col_ok = col < n_out;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok && bias) b4 = *reinterpret_cast<const float4*>(bias + (int64_t)g * bgs + col);
    const int blk = col_ok ? col / block_cols : 0, cc = col - blk * block_cols;
    float* __restrict__ ob = (blk == 0) ? out0 : ((blk == 1) ? out1 : out2);
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
        const int j = jq >> 2, q = jq & 3;
        float v0 = acc[j][4 * q], v1 = acc[j][4 * q + 1], v2 = acc[j][4 * q + 2], v3 = acc[j][4 * q + 3];
        quad_transpose(v0, v1, v2, v3, o1, o2);
        const float sc = s_inv ? s_inv[j * 32 + (lane & 3) + 8 * q + 4 * (lane >> 5)] * winv : 1.0f;
        pr.v[jq] = f32x4{v0 * sc + b4.x, v1 * sc + b4.y, v2 * sc + b4.z, v3 * sc + b4.w};
    }
    pr.base = col_ok ? ob + cc : nullptr;
    pr.ld = block_cols;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
}

template <int PROLOGUE, bool UPD, bool F16>
__global__ __launch_bounds__(PC_THREADS) void k_typed_linear_pc(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off,
    int n_groups, int k, int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int vec_ok,
    UpdateArgs upd, int pass_split) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][2 * A_PLANE];   // [slab][plane][64][528]
    __shared__ int s_rid[3][BM];   // three tables: the parked rows of tile i are written while tile i+2 is being loaded
    __shared__ __attribute__((aligned(16))) float s_red[2][2][BM * 8];          // [parity][sum|var][row][wave]
    __shared__ float s_rinv[F16 ? 3 : 1][F16 ? BM : 1];                         // fp16 split: inverse row scales, rotating like s_rid

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // number of tiles of this workgroup (the same value in every wave: barrier counts must agree)
    // work items: row tiles, or (row tile, 256-column pass) pairs when pass_split = n_pass > 1 (small problems: more workgroups
    // than row tiles -- a sampled sub-graph has ~50 row tiles for 256 CUs)
    int total_tiles = 0;
    for (int g = 0; g < n_groups; ++g) total_tiles += (group_off[g + 1] - group_off[g] + BM - 1) / BM;
    total_tiles *= pass_split;
    const int first = blockIdx.x, stride = gridDim.x;
    const int n_mine = (total_tiles > first) ? (total_tiles - first + stride - 1) / stride : 0;
    if (n_mine == 0) return;

    if (wave >= PC_CONS) {
        // ------------------------------------------------ producers
        const int pw = wave - PC_CONS;
        // lane id from mbcnt: reusing the consumers' `lane` keeps it live across their whole code and costs a spill
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        float4 a[PC_AREGS];
        int v_rid, g, row0, nrows;
        tile_lookup(first / pass_split, group_off, n_groups, g, row0, nrows);
        pc_issue<PROLOGUE>(a, v_rid, pw, lane, row0, nrows, rows, x, ldx, k, vec_ok);
        pc_commit<PROLOGUE, F16>(a, v_rid, pw, lane, sA[0], s_rid[0], s_rinv[0]);
        if (n_mine > 1) {
            tile_lookup((first + stride) / pass_split, group_off, n_groups, g, row0, nrows);
            pc_issue<PROLOGUE>(a, v_rid, pw, lane, row0, nrows, rows, x, ldx, k, vec_ok);
        }
        pc_barrier();                                   // B_0
        for (int i = 0; i < n_mine; ++i) {
            if (i + 1 < n_mine) {
                pc_commit<PROLOGUE, F16>(a, v_rid, pw, lane, sA[(i + 1) & 1], s_rid[(i + 1) % 3], s_rinv[F16 ? (i + 1) % 3 : 0]);
                if (i + 2 < n_mine) {
                    tile_lookup((first + (i + 2) * stride) / pass_split, group_off, n_groups, g, row0, nrows);
                    pc_issue<PROLOGUE>(a, v_rid, pw, lane, row0, nrows, rows, x, ldx, k, vec_ok);
                }
            }
            if constexpr (UPD) {
                pc_barrier();
                pc_barrier();
            }
            if (i + 1 < n_mine) pc_barrier();           // B_{i+1}
        }
        return;
    }

    // ---------------------------------------------------- consumers
    const int lane = tid & 63;
    const int n_pass = (n_out + BNP - 1) / BNP;
    const int n_kc = ((k + KC - 1) / KC + 3) & ~3;
    const int total = n_pass * n_kc;
    const int frow = lane & 31, khalf = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    PendingRows pr;
    bool have_pend = false;   // wave-uniform
#pragma unroll
    for (int u = 0; u < 8; ++u) pr.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    pr.base = nullptr;
    pr.ld = 0;
    pr.rid = s_rid[0];
    pr.row0 = pr.nrows = pr.by_pos = 0;
#define PC_STORE(U)                                                                                   \
    if (have_pend) {                                                                 \
        const int r_ = pr.row(U, lane);                                                               \
        if (r_ >= 0) hidden_store16(pr.base + (int64_t)r_ * pr.ld, pr.v[U]);                          \
    }

    for (int i = 0; i < n_mine; ++i) {
        int g, row0, nrows;
        const int vt = first + i * stride;
        tile_lookup(vt / pass_split, group_off, n_groups, g, row0, nrows);
        const int pass_lo = (pass_split > 1) ? vt % pass_split : 0, pass_hi = (pass_split > 1) ? pass_lo + 1 : n_pass;
        const unsigned short* __restrict__ wfrag = wsplit + (int64_t)g * total * 2 * W_PLANE_ELEMS + (wave * 64 + lane) * 8;
        const float winv = F16 ? reinterpret_cast<const float*>(wsplit + (int64_t)n_groups * total * 2 * W_PLANE_ELEMS)[g] : 1.0f;
        const float* s_inv = F16 ? s_rinv[F16 ? i % 3 : 0] : nullptr;
        bf16x8 s0h, s0m, s1h, s1m, s2h, s2m, s3h, s3m;
#define PC_LOAD_STAGE(S, T)                                                                           \
    {                                                                                                 \
        const unsigned short* t_ = wfrag + (int64_t)min((T), total - 1) * 2 * W_PLANE_ELEMS;          \
        s##S##h = *reinterpret_cast<const bf16x8*>(t_);                                               \
        s##S##m = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                               \
    }
        // B fragments are prefetched NST k-chunks ahead (the fused-update variant has fewer registers to spare)
        constexpr int NST = UPD ? 2 : 4;
        PC_LOAD_STAGE(0, pass_lo * n_kc)
        PC_LOAD_STAGE(1, pass_lo * n_kc + 1)
        if constexpr (NST >= 4) {
            PC_LOAD_STAGE(2, pass_lo * n_kc + 2)
            PC_LOAD_STAGE(3, pass_lo * n_kc + 3)
        }
        pc_barrier();                                      // B_i: slab[i&1] holds tile i
        const unsigned char* slab = sA[i & 1] + frow * A_STRIDE + khalf * 16;
        // A fragments (2 row tiles x hi/mid) are read one k-chunk ahead, into alternating register sets
        bf16x8 e_h0, e_m0, e_h1, e_m1, o_h0, o_m0, o_h1, o_m1;
#define PC_LOAD_A(P, KCP)                                                                              \
    {                                                                                                 \
        const unsigned char* a_ = slab + (KCP) * (KC * 2);                                            \
        P##_h0 = *reinterpret_cast<const bf16x8*>(a_);                                                \
        P##_m0 = *reinterpret_cast<const bf16x8*>(a_ + A_PLANE);                                      \
        P##_h1 = *reinterpret_cast<const bf16x8*>(a_ + 32 * A_STRIDE);                                \
        P##_m1 = *reinterpret_cast<const bf16x8*>(a_ + A_PLANE + 32 * A_STRIDE);                      \
    }
        // One k-chunk: 6 MFMAs (small terms first, hi*hi last; the two accumulators alternate).  The sched barriers pin
        // the order "next A fragments -> MFMAs -> refill of the consumed B stage": left alone, hipcc sinks all the B
        // loads of the 4-step body to its end (a stall at the top of every body), and a refill issued BEFORE the
        // stage's last use gets a fresh register plus a v_mov rotation of all stages that waits for every load.
#define PC_STEP(S, T, P, PN, KNEXT)                                                                                \
    {                                                                                                              \
        PC_LOAD_A(PN, KNEXT)                                                                                       \
        acc[0] = mfma32_t<F16>(P##_m0, s##S##h, acc[0]);                        \
        acc[1] = mfma32_t<F16>(P##_m1, s##S##h, acc[1]);                        \
        acc[0] = mfma32_t<F16>(P##_h0, s##S##m, acc[0]);                        \
        acc[1] = mfma32_t<F16>(P##_h1, s##S##m, acc[1]);                        \
        acc[0] = mfma32_t<F16>(P##_h0, s##S##h, acc[0]);                        \
        acc[1] = mfma32_t<F16>(P##_h1, s##S##h, acc[1]);                        \
        PC_LOAD_STAGE(S, (T) + NST)                                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); /* 4 DS reads    */                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); /* 6 MFMAs       */                                     \
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0); /* 2 VMEM reads  */                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
        // 4 k-chunks + two of the previous pass's parked output rows
#define PC_BODY(B)                                                                                                 \
    {                                                                                                              \
        const int kq = 4 * (B);                                                                                    \
        const int knext = (kq + 4 == n_kc) ? 0 : kq + 4; /* the next pass starts over on the same slab */          \
        if constexpr (NST == 4) {                                                                                  \
            PC_STEP(0, tbase + kq, e, o, kq + 1)                                                                   \
            PC_STEP(1, tbase + kq + 1, o, e, kq + 2)                                                               \
            PC_STORE(2 * (B))                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            PC_STEP(2, tbase + kq + 2, e, o, kq + 3)                                                               \
            PC_STEP(3, tbase + kq + 3, o, e, knext)                                                                \
            PC_STORE(2 * (B) + 1)                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        } else {                                                                                                   \
            PC_STEP(0, tbase + kq, e, o, kq + 1)                                                                   \
            PC_STEP(1, tbase + kq + 1, o, e, kq + 2)                                                               \
            PC_STORE(2 * (B))                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            PC_STEP(0, tbase + kq + 2, e, o, kq + 3)                                                               \
            PC_STEP(1, tbase + kq + 3, o, e, knext)                                                                \
            PC_STORE(2 * (B) + 1)                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                          \
    }
        PC_LOAD_A(e, 0)
        for (int pass = pass_lo; pass < pass_hi; ++pass) {
            const int tbase = pass * n_kc;
            PC_BODY(0)
            if (n_kc > 4) PC_BODY(1)
            if (n_kc > 8) PC_BODY(2)
            if (n_kc > 12) PC_BODY(3)
            // k < 256: the bodies that did not run leave their rows behind
            if (n_kc <= 12) { PC_STORE(6) PC_STORE(7) }
            if (n_kc <= 8) { PC_STORE(4) PC_STORE(5) }
            if (n_kc <= 4) { PC_STORE(2) PC_STORE(3) }
            if constexpr (UPD) {
                pc_store_update(acc, wave, lane, g, n_out, nrows, s_rid[i % 3], bias, bgs, out0, upd, s_red[i & 1][0], s_red[i & 1][1], pr,
                                s_inv, winv);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            } else {
                pc_stage_pass(acc, pass, wave, lane, g, n_out, nrows, row0, s_rid[i % 3], bias, bgs, out0, out1, out2, block_cols, by_pos, pr,
                              s_inv, winv);
            }
            have_pend = true;
            pr.rid = s_rid[i % 3];
            pr.row0 = row0;
            pr.nrows = nrows;
            pr.by_pos = UPD ? 0 : by_pos;
        }
#undef PC_LOAD_A
#undef PC_STEP
#undef PC_BODY
#undef PC_LOAD_STAGE
    }
    // the rows of the very last pass
    PC_STORE(0) PC_STORE(1) PC_STORE(2) PC_STORE(3) PC_STORE(4) PC_STORE(5) PC_STORE(6) PC_STORE(7)
#undef PC_STORE
}


static int pc_grid() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v;
    }
    return n_cu;
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
    // tiles + tail: [n_groups] inverse weight scales, [n_groups] scales (written by hgt_split_weights_f16; unused by the bf16 image)
    *out = (uint64_t)n_groups * n_pass * n_kstep * 2 * W_PLANE_ELEMS * 2 + hgt_align_up((uint64_t)n_groups * 8, 256);
    return HGT_OK;
}

extern "C" int hgt_split_weights(const float* W, int64_t w_group_stride, int32_t n_groups, int32_t k, int32_t n_out, void* w_split,
                                 void* stream) {
    if (!W || !w_split || n_groups <= 0 || k <= 0 || n_out <= 0) return HGT_ERR_INVALID_ARG;
    int n_pass, n_kstep;
    split_dims(k, n_out, &n_pass, &n_kstep);
    const int64_t total = (int64_t)n_groups * n_pass * n_kstep * W_PLANE_ELEMS;
    k_split_weights<false><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, w_group_stride, n_groups, k, n_out, n_pass,
                                                                                            n_kstep, (unsigned short*)w_split, nullptr);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_split_weights_f16(const float* W, int64_t w_group_stride, int32_t n_groups, int32_t k, int32_t n_out, void* w_split,
                                     void* stream) {
    if (!W || !w_split || n_groups <= 0 || k <= 0 || n_out <= 0) return HGT_ERR_INVALID_ARG;
    int n_pass, n_kstep;
    split_dims(k, n_out, &n_pass, &n_kstep);
    const int64_t total = (int64_t)n_groups * n_pass * n_kstep * W_PLANE_ELEMS;
    float* tail = reinterpret_cast<float*>((unsigned short*)w_split + total * 2);
    k_group_scale<<<(unsigned)n_groups, 1024, 0, (hipStream_t)stream>>>(W, w_group_stride, (int64_t)n_out * k, tail, n_groups);
    k_split_weights<true><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, w_group_stride, n_groups, k, n_out, n_pass,
                                                                                           n_kstep, (unsigned short*)w_split, tail + n_groups);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

template <bool F16>
static int typed_linear_split_impl(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                   int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                   int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                   int32_t out_by_position, int32_t prologue, void* stream_) {
    if (!x || !rows || !group_off || !w_split || !out0 || n_groups <= 0 || n_rows < 0 || k <= 0 || n_out <= 0 || block_cols <= 0)
        return HGT_ERR_INVALID_ARG;
    const int n_blocks_out = (n_out + block_cols - 1) / block_cols;
    if (n_blocks_out > 3 || (n_blocks_out > 1 && !out1) || (n_blocks_out > 2 && !out2)) return HGT_ERR_INVALID_ARG;
    const int32_t prologue_arg = prologue;      // (with the kernel-selection bits HGT_LINEAR_FORCE_XS / HGT_LINEAR_NO_XS)
    if (prologue < 0 || (prologue & ~(0xff | HGT_LINEAR_FORCE_XS | HGT_LINEAR_NO_XS | HGT_LINEAR_NO_TILE | HGT_LINEAR_TANH)) != 0) return HGT_ERR_INVALID_ARG;
    const int act_tanh = (prologue & HGT_LINEAR_TANH) ? 1 : 0;      // tanh on the output: the tile kernel / the K > 256 slab kernel only
    prologue &= 0xff;
    if (prologue > 2) return HGT_ERR_INVALID_ARG;
    if (prologue == 2 && (k > KP || (k & 3) != 0 || ldx != 3 * (int64_t)(k / 4))) return HGT_ERR_UNSUPPORTED;   // 24-bit wire rows: the persistent kernel only
    if (((n_out | block_cols) & 3) != 0) return HGT_ERR_UNSUPPORTED;   // 16-byte epilogue stores; use hgt_typed_linear (fp32) instead
    if (n_rows == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;   // device-side group sizes: launch the upper bound
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const int vec_ok = (ldx % 4 == 0) && (k % 4 == 0) && (((uintptr_t)x & 15) == 0);
    UpdateArgs noupd = {nullptr, 0, nullptr, nullptr, nullptr, 0};
    // latency regime (sampled sub-graphs: fewer row tiles than CUs): one workgroup per (row tile, 256-column pass)
    const int n_pass = (n_out + BNP - 1) / BNP;
    const int pass_split = (n_pass > 1 && row_tiles * 2 <= pc_grid()) ? n_pass : 1;
    if (!(prologue_arg & (HGT_LINEAR_NO_TILE | HGT_LINEAR_FORCE_XS)) && prologue <= 1) {
        // sampled batches (a few thousand rows): the tile kernel (hgt_gemm_tile.hip), laid out for a short dependent chain
        const int tl = hgt_typed_linear_tile_try(F16, x, ldx, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, out0, out1,
                                                 out2, block_cols, out_by_position, prologue | (act_tanh ? HGT_LINEAR_TANH : 0), nullptr, stream_);
        if (tl != 0) return tl < 0 ? tl : HGT_OK;
    }
    if (act_tanh && k <= KP) return HGT_ERR_UNSUPPORTED;      // (the persistent / x-stationary kernels have no activation epilogue: the caller runs hgt_tanh_inplace)
    if (!act_tanh) {   // millions of rows: the x-stationary kernel (hgt_gemm_xs.hip: W through LDS once per 256 rows; K = 64 / 128 / 256 / 512)
        const int xs = hgt_typed_linear_xs_try(F16, x, ldx, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, out0, out1,
                                               out2, block_cols, out_by_position, prologue_arg, stream_);
        if (xs != 0) return xs < 0 ? xs : HGT_OK;
    }
    if (k <= KP) {
        // persistent producer/consumer kernel, one workgroup per CU
        const unsigned grid = (unsigned)std::min<int64_t>(row_tiles * pass_split, pc_grid());
        if (prologue == 0)
            k_typed_linear_pc<0, false, F16><<<grid, PC_THREADS, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                         (const unsigned short*)w_split, bias, b_group_stride, out0, out1,
                                                                         out2, block_cols, out_by_position, vec_ok, noupd, pass_split);
        else if (prologue == 1)
            k_typed_linear_pc<1, false, F16><<<grid, PC_THREADS, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                         (const unsigned short*)w_split, bias, b_group_stride, out0, out1,
                                                                         out2, block_cols, out_by_position, vec_ok, noupd, pass_split);
        else
            k_typed_linear_pc<2, false, F16><<<grid, PC_THREADS, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,
                                                                         (const unsigned short*)w_split, bias, b_group_stride, out0, out1,
                                                                         out2, block_cols, out_by_position, 1, noupd, pass_split);
        HGT_CHECK_LAUNCH();
        return HGT_OK;
    }
    const unsigned grid_s = (unsigned)(row_tiles * pass_split);
    const bool deep = row_tiles * pass_split <= 2 * pc_grid();      // latency regime: four B-fragment stages (see the kernel)
#define HGT_SPLIT_LAUNCH(P, NS)                                                                                                        \
    k_typed_linear_split<P, false, F16, NS><<<grid_s, 512, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out,                   \
                                                                        (const unsigned short*)w_split, bias, b_group_stride, out0, out1, \
                                                                        out2, block_cols, out_by_position, vec_ok, noupd, pass_split | (act_tanh << 16))
    if (prologue == 0) { if (deep) HGT_SPLIT_LAUNCH(0, 4); else HGT_SPLIT_LAUNCH(0, 2); }
    else               { if (deep) HGT_SPLIT_LAUNCH(1, 4); else HGT_SPLIT_LAUNCH(1, 2); }
#undef HGT_SPLIT_LAUNCH
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_typed_linear_bf16x3(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                       int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                       int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                       int32_t out_by_position, int32_t prologue, void* stream) {
    return typed_linear_split_impl<false>(x, ldx, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, out0, out1, out2,
                                          block_cols, out_by_position, prologue, stream);
}

extern "C" int hgt_typed_linear_f16x3(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                      int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                      int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                      int32_t out_by_position, int32_t prologue, void* stream) {
    return typed_linear_split_impl<true>(x, ldx, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, out0, out1, out2,
                                         block_cols, out_by_position, prologue, stream);
}

// a_linear + gated skip + LayerNorm in one kernel (n_out <= 256, or n_out <= 512 with k <= 512; n_out % 4 == 0): see store_pass_update /
// store_update_wide.
template <bool F16>
static int linear_update_split_impl(const float* agg, int64_t ld_agg, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                    int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                    int64_t b_group_stride, const float* x_skip, int64_t ld_skip, const float* skip,
                                    const float* ln_w, const float* ln_b, int32_t use_norm, float* out, void* stream_) {
    if (!agg || !rows || !group_off || !w_split || !x_skip || !skip || !out || n_groups <= 0 || n_rows < 0 || k <= 0 || n_out <= 0)
        return HGT_ERR_INVALID_ARG;
    if ((use_norm & 1) && (!ln_w || !ln_b)) return HGT_ERR_INVALID_ARG;
    if (n_out > 2 * BNP || (n_out > BNP && k > 2 * KP) || (n_out & 3) != 0 || (ld_skip & 3) != 0 || ((uintptr_t)x_skip & 15) != 0)
        return HGT_ERR_UNSUPPORTED;
    if (n_rows == 0) return HGT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const int vec_ok = (ld_agg % 4 == 0) && (k % 4 == 0) && (((uintptr_t)agg & 15) == 0);
    UpdateArgs u = {x_skip, ld_skip, skip, ln_w, ln_b, use_norm & 1};
    if (!(use_norm & 2)) {      // (bit 1 of use_norm: HGT_LINEAR_NO_TILE of this entry point -- tests / A/B timings)
        const HgtTileUpdate tu = {x_skip, ld_skip, skip, ln_w, ln_b, use_norm & 1};
        const int tl = hgt_typed_linear_tile_try(F16, agg, ld_agg, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, out,
                                                 nullptr, nullptr, n_out, 0, 0, &tu, stream_);
        if (tl != 0) return tl < 0 ? tl : HGT_OK;
    }
    if (n_out > BNP) {      // 257..512 columns: both passes' accumulators stay in registers (k_typed_linear_update_wide)
        k_typed_linear_update_wide<F16><<<(unsigned)row_tiles, 512, 0, stream>>>(agg, ld_agg, rows, group_off, n_groups, k, n_out,
                                                                               (const unsigned short*)w_split, bias, b_group_stride, out,
                                                                               vec_ok, u);
        HGT_CHECK_LAUNCH();
        return HGT_OK;
    }
    if (k <= KP) {
        const unsigned grid = (unsigned)std::min<int64_t>(row_tiles, pc_grid());
        k_typed_linear_pc<0, true, F16><<<grid, PC_THREADS, 0, stream>>>(agg, ld_agg, rows, group_off, n_groups, k, n_out,
                                                                    (const unsigned short*)w_split, bias, b_group_stride, out, nullptr,
                                                                    nullptr, n_out, 0, vec_ok, u, 1);
        HGT_CHECK_LAUNCH();
        return HGT_OK;
    }
    k_typed_linear_split<0, true, F16><<<(unsigned)row_tiles, 512, 0, stream>>>(agg, ld_agg, rows, group_off, n_groups, k, n_out,
                                                                           (const unsigned short*)w_split, bias, b_group_stride, out,
                                                                           nullptr, nullptr, n_out, 0, vec_ok, u, 1);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_linear_update_bf16x3(const float* agg, int64_t ld_agg, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                        int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                        int64_t b_group_stride, const float* x_skip, int64_t ld_skip, const float* skip,
                                        const float* ln_w, const float* ln_b, int32_t use_norm, float* out, void* stream) {
    return linear_update_split_impl<false>(agg, ld_agg, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, x_skip,
                                           ld_skip, skip, ln_w, ln_b, use_norm, out, stream);
}

extern "C" int hgt_linear_update_f16x3(const float* agg, int64_t ld_agg, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                       int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                       int64_t b_group_stride, const float* x_skip, int64_t ld_skip, const float* skip,
                                       const float* ln_w, const float* ln_b, int32_t use_norm, float* out, void* stream) {
    return linear_update_split_impl<true>(agg, ld_agg, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, b_group_stride, x_skip,
                                          ld_skip, skip, ln_w, ln_b, use_norm, out, stream);
}
