// Fused node update of HGTConv (conv.py:119-133) as the epilogue of an aggregation kernel: shared by the vector-ALU and the
// matrix-core aggregation kernels and by k_update_pending.
#pragma once
#include "hgt_edge_common.h"
#include "hgt_split_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Pass 2 with the node update fused in (HGTConv, split-bf16 precision, d_pad <= 256, plan without hubs):
//     out[i] = LN_t( (gelu(agg_i) W_a[t]^T + b_a[t]) * sigmoid(skip[t]) + x_i * (1 - sigmoid(skip[t])) )   conv.py:119-133
// The workgroup that aggregated 64 targets already holds their finished rows; writing them to HBM only for a second
// kernel to read them back costs 2 x 4d bytes per node and a kernel that is latency bound on its own (1.2 ms at c2).
// Here the rows go registers -> LDS as the bf16 hi/mid A operand (the 66 KB slab overlays the accumulator, which is
// dead by then), the four wavefronts run the 64 x d x d split-bf16 MFMA product against the L2-resident fragment-ordered
// W_a (hgt_split_weights), and the gated skip + LayerNorm epilogue writes `out` directly.  agg never touches HBM --
// which is what the minimal-traffic model of SURVEY.md 8(d) assumes.
// A tile whose rows have several node types (only at the T-1 type boundaries of a type-sorted graph) repeats the
// product per type present; rows of unknown type are written as 0 (conv.py:120).
// ---------------------------------------------------------------------------------------------
// A operand of the epilogue: hi/mid bf16 planes of this wavefront's 16 finished rows (gather layout: lane = VEC consecutive
// columns of a row), [plane][row][k], 528 B row stride
constexpr int FU_RINV_OFF = 2560;      // byte offset of the fp16 split's inverse row scales ([64] floats) inside `tables`
template <int VEC, bool F16 = false>
__device__ __forceinline__ void fused_slab_from_rows(const float (&vals)[16][VEC], unsigned char* slab, float* s_rinv = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned char* prow = slab + (wave * 16 + r) * A_STRIDE + lane * VEC * 2;
        unsigned short hi[VEC], mid[VEC];
        float scale = 1.0f;
        if constexpr (F16) {      // the wavefront holds the whole row (gather layout)
            float m = fabsf(vals[r][0]);
#pragma unroll
            for (int i = 1; i < VEC; ++i) m = fmaxf(m, fabsf(vals[r][i]));
            float inv;
            f16_row_scale(wave_max_bits(__builtin_bit_cast(unsigned, m)), scale, inv);
            if (lane == 0) s_rinv[wave * 16 + r] = inv;
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) split1_t<F16>(vals[r][i], scale, hi[i], mid[i]);
        if constexpr (VEC == 1) {
            *reinterpret_cast<unsigned short*>(prow) = hi[0];
            *reinterpret_cast<unsigned short*>(prow + A_PLANE) = mid[0];
        } else if constexpr (VEC == 2) {
            *reinterpret_cast<unsigned*>(prow) = (unsigned)hi[0] | ((unsigned)hi[1] << 16);
            *reinterpret_cast<unsigned*>(prow + A_PLANE) = (unsigned)mid[0] | ((unsigned)mid[1] << 16);
        } else {
            *reinterpret_cast<uint2*>(prow) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
            *reinterpret_cast<uint2*>(prow + A_PLANE) =
                make_uint2((unsigned)mid[0] | ((unsigned)mid[1] << 16), (unsigned)mid[2] | ((unsigned)mid[3] << 16));
        }
    }
}

// The epilogue proper.  `slab` = the 2*A_PLANE bytes of LDS holding the A operand of the workgroup's 64 rows (gelu applied),
// written by every wavefront before this call (no barrier needed in between: the first thing here is one);
// `tables` = 2.5 KB of LDS for the row types and the LayerNorm partial sums.
// NSTG = k-chunks of W_a fragments prefetched ahead (4: fastest; 2: 32 registers fewer, for callers that would spill)
// XPRE: the 16 skip-connection loads of a lane (x rows of the tile) are requested BEFORE the product and consumed after it.
// Written next to their uses they are conditional loads inside an unrolled loop, and hipcc waits for each one with vmcnt(0)
// right before its use: 16 dependent HBM round trips per tile, ~30 us of a workgroup's ~125 us at c2 (measured by removing the
// epilogue: 3.61 -> 2.66 ms) -- the whole cost of the fused epilogue.  Unconditional (clamped) loads ahead of the MFMA loop take one.
constexpr int FU_NO_TYPE = -0x7fffffff;
template <int VEC, int NSTG = 4, bool XPRE = false, bool F16 = false>
__device__ __forceinline__ void fused_update_tail(unsigned char* slab, unsigned char* tables, int64_t row0, int64_t NQ,
                                                  const HgtFusedUpdate& fu, int type_pre = FU_NO_TYPE) {
    constexpr int DP = 64 * VEC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int* s_type = reinterpret_cast<int*>(tables);                // [64]
    float* s_sum = reinterpret_cast<float*>(tables + 256);         // [64][4]
    float* s_var = s_sum + 256;                                    // [64][4]
    const float* s_rinv = reinterpret_cast<const float*>(tables + FU_RINV_OFF);   // [64], fp16 split only (written by the caller)
    if (tid < 64) {
        // type_pre: the caller requested node_type[row0 + lane] before its own barrier (one round trip off the critical path)
        int64_t t = type_pre;
        if (type_pre == FU_NO_TYPE) {
            const int64_t row = row0 + tid;
            t = (row < NQ) ? fu.node_type[row] : -1;
        }
        s_type[tid] = (t >= 0 && t < fu.n_types) ? (int)t : -1;
    }
    __syncthreads();

    const int my_t = s_type[lane];
    int tmin = my_t < 0 ? 0x7fffffff : my_t, tmax = my_t;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        tmin = min(tmin, __shfl_xor(tmin, sft));
        tmax = max(tmax, __shfl_xor(tmax, sft));
    }
    tmin = __builtin_amdgcn_readfirstlane(tmin);
    tmax = __builtin_amdgcn_readfirstlane(tmax);

    constexpr int NKC = DP / KC;                       // k-chunks (a multiple of 4, like split_dims)
    const int n_out = fu.n_out;
    const int frow = lane & 31, khalf = lane >> 5;
    const unsigned char* abase = slab + frow * A_STRIDE + khalf * 16;
    const bool o1 = lane & 1, o2 = lane & 2;
    const float inv_n = 1.0f / (float)n_out;

    for (int g = tmin; g <= tmax; ++g) {               // empty range when no row has a valid type
        if (__builtin_amdgcn_ballot_w64(my_t == g) == 0) continue;
        // ---- 64 x n_out x DP product; this wavefront owns columns [64 wave, 64 wave + 64) = column tiles 2 wave, 2 wave + 1
        f32x16 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.0f;
        float4 xpre[XPRE ? 16 : 1], bpre[XPRE ? 2 : 1], wpre[XPRE ? 2 : 1], cpre[XPRE ? 2 : 1];
        float skip_pre = 0.0f;
        if constexpr (XPRE) {
            skip_pre = fu.skip[g];
            int lane_x = lane;
            asm volatile("" : "+v"(lane_x));            // (keeps the address arithmetic out of the walk, like lane_e below)
            const int rt0x = (lane_x & 3) + 4 * (lane_x >> 5);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = wave * 64 + c * 32 + ((lane_x & 31) >> 2) * 4;
                const int colc = col < n_out ? col : 0;
                // bias and LayerNorm rows of this lane's columns as well: loaded where they are used (behind `if (col_ok)`), the
                // waits for them land inside the store loop as vmcnt(0), which also waits for the previous STORE's acknowledge:
                // sixteen serialised write round trips per tile (ISA audit)
                bpre[c] = *reinterpret_cast<const float4*>(fu.bias + (int64_t)g * n_out + colc);
                const float* lw = fu.use_norm ? fu.lnw : fu.bias;      // (always a valid address; ignored without LayerNorm)
                const float* lb = fu.use_norm ? fu.lnb : fu.bias;
                wpre[c] = *reinterpret_cast<const float4*>(lw + (int64_t)g * n_out + colc);
                cpre[c] = *reinterpret_cast<const float4*>(lb + (int64_t)g * n_out + colc);
#pragma unroll
                for (int jq = 0; jq < 8; ++jq) {
                    const int64_t row = min(row0 + rt0x + 32 * (jq >> 2) + 8 * (jq & 3), NQ - 1);
                    xpre[c * 8 + jq] = *reinterpret_cast<const float4*>(fu.xs + row * fu.ldxs + colc);
                }
            }
        }
        if (wave * 64 < n_out) {
            const unsigned short* __restrict__ wf = fu.w_split + (int64_t)g * NKC * 2 * W_PLANE_ELEMS + ((2 * wave) * 64 + lane) * 8;
            // B fragments of two column tiles, NSTG k-chunks ahead in named register stages; A fragments one chunk ahead.
            // The sched barriers pin "next A -> 12 MFMAs -> refill of the consumed stage" (see k_typed_linear_pc: left alone,
            // hipcc sinks the loads next to their uses and every wait becomes a wait for a load that was just issued).
            bf16x8 s0h0, s0h1, s0m0, s0m1, s1h0, s1h1, s1m0, s1m1, s2h0, s2h1, s2m0, s2m1, s3h0, s3h1, s3m0, s3m1;
            bf16x8 s4h0, s4h1, s4m0, s4m1, s5h0, s5h1, s5m0, s5m1, s6h0, s6h1, s6m0, s6m1, s7h0, s7h1, s7m0, s7m1;   // (NSTG == 8 only)
            constexpr int NST = (NSTG == 8 && NKC % 8 != 0) ? 4 : NSTG;      // eight stages need a multiple of eight k-chunks
            bf16x8 e_h0, e_m0, e_h1, e_m1, o_h0, o_m0, o_h1, o_m1;
#define FU_LOAD_B(S, KCI)                                                                            \
    {                                                                                                \
        const unsigned short* t_ = wf + (int64_t)min((KCI), NKC - 1) * 2 * W_PLANE_ELEMS;            \
        s##S##h0 = *reinterpret_cast<const bf16x8*>(t_);                                             \
        s##S##h1 = *reinterpret_cast<const bf16x8*>(t_ + 64 * 8);                                    \
        s##S##m0 = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS);                             \
        s##S##m1 = *reinterpret_cast<const bf16x8*>(t_ + W_PLANE_ELEMS + 64 * 8);                    \
    }
#define FU_LOAD_A(P, KCI)                                                                            \
    {                                                                                                \
        const unsigned char* a_ = abase + min((KCI), NKC - 1) * (KC * 2);                            \
        P##_h0 = *reinterpret_cast<const bf16x8*>(a_);                                               \
        P##_m0 = *reinterpret_cast<const bf16x8*>(a_ + A_PLANE);                                     \
        P##_h1 = *reinterpret_cast<const bf16x8*>(a_ + 32 * A_STRIDE);                               \
        P##_m1 = *reinterpret_cast<const bf16x8*>(a_ + A_PLANE + 32 * A_STRIDE);                     \
    }
#define FU_STEP(S, KCI, P, PN)                                                                                     \
    {                                                                                                              \
        FU_LOAD_A(PN, (KCI) + 1)                                                                                   \
        acc[0][0] = mfma32_t<F16>(P##_m0, s##S##h0, acc[0][0]);                 \
        acc[0][1] = mfma32_t<F16>(P##_m1, s##S##h0, acc[0][1]);                 \
        acc[1][0] = mfma32_t<F16>(P##_m0, s##S##h1, acc[1][0]);                 \
        acc[1][1] = mfma32_t<F16>(P##_m1, s##S##h1, acc[1][1]);                 \
        acc[0][0] = mfma32_t<F16>(P##_h0, s##S##m0, acc[0][0]);                 \
        acc[0][1] = mfma32_t<F16>(P##_h1, s##S##m0, acc[0][1]);                 \
        acc[1][0] = mfma32_t<F16>(P##_h0, s##S##m1, acc[1][0]);                 \
        acc[1][1] = mfma32_t<F16>(P##_h1, s##S##m1, acc[1][1]);                 \
        acc[0][0] = mfma32_t<F16>(P##_h0, s##S##h0, acc[0][0]);                 \
        acc[0][1] = mfma32_t<F16>(P##_h1, s##S##h0, acc[0][1]);                 \
        acc[1][0] = mfma32_t<F16>(P##_h0, s##S##h1, acc[1][0]);                 \
        acc[1][1] = mfma32_t<F16>(P##_h1, s##S##h1, acc[1][1]);                 \
        FU_LOAD_B(S, (KCI) + NST)                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  /* 4 DS reads   */                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0); /* 12 MFMAs     */                                     \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);  /* 4 VMEM reads */                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    }
            FU_LOAD_B(0, 0)
            FU_LOAD_B(1, 1)
            if constexpr (NST >= 4) {
                FU_LOAD_B(2, 2)
                FU_LOAD_B(3, 3)
            }
            if constexpr (NST == 8) {
                FU_LOAD_B(4, 4)
                FU_LOAD_B(5, 5)
                FU_LOAD_B(6, 6)
                FU_LOAD_B(7, 7)
            }
            FU_LOAD_A(e, 0)
            for (int kq = 0; kq < NKC; kq += (NST == 8 ? 8 : 4)) {
                if constexpr (NST == 8) {
                    FU_STEP(0, kq, e, o)
                    FU_STEP(1, kq + 1, o, e)
                    FU_STEP(2, kq + 2, e, o)
                    FU_STEP(3, kq + 3, o, e)
                    FU_STEP(4, kq + 4, e, o)
                    FU_STEP(5, kq + 5, o, e)
                    FU_STEP(6, kq + 6, e, o)
                    FU_STEP(7, kq + 7, o, e)
                } else if constexpr (NST == 4) {
                    FU_STEP(0, kq, e, o)
                    FU_STEP(1, kq + 1, o, e)
                    FU_STEP(2, kq + 2, e, o)
                    FU_STEP(3, kq + 3, o, e)
                } else {
                    FU_STEP(0, kq, e, o)
                    FU_STEP(1, kq + 1, o, e)
                    FU_STEP(0, kq + 2, e, o)
                    FU_STEP(1, kq + 3, o, e)
                }
            }
#undef FU_STEP
#undef FU_LOAD_A
#undef FU_LOAD_B
        }
        // ---- epilogue for the rows of type g: bias, gated skip, LayerNorm, store
        // The lane id is laundered through an empty asm: otherwise hipcc hoists the ~50 LDS / global address computations of
        // this section above the type loop, keeps them live across the MFMA section (which needs the whole register file)
        // and spills them -- 236 B of scratch per thread = 1.9 GB of extra HBM traffic per launch at c2 (rocprofv3 WRITE_SIZE).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int rt0e = (lane_e & 3) + 4 * (lane_e >> 5);
        const float alpha = 1.0f / (1.0f + expf(-(XPRE ? skip_pre : fu.skip[g])));
        // fp16 split: inverse scale of the group's W_a image (tail of the image, hgt_split_weights_f16) x the rows' inverse scales
        const float winv = F16 ? reinterpret_cast<const float*>(fu.w_split + (int64_t)fu.n_types * NKC * 2 * W_PLANE_ELEMS)[g] : 1.0f;
        float y[16][4];                                // [c*8 + j*4 + q][4 consecutive columns]
        int orow[8];                                   // row of group (j, q); -1 = not a row of this type
#pragma unroll
        for (int jq = 0; jq < 8; ++jq) {
            const int rt = rt0e + 32 * (jq >> 2) + 8 * (jq & 3);
            orow[jq] = (s_type[rt] == g) ? rt : -1;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = wave * 64 + c * 32 + ((lane_e & 31) >> 2) * 4;
            const bool col_ok = col < n_out;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (XPRE) {
                if (col_ok) b4 = bpre[c];
            } else {
                if (col_ok) b4 = *reinterpret_cast<const float4*>(fu.bias + (int64_t)g * n_out + col);
            }
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                const int j = jq >> 2, q = jq & 3;
                float v0 = acc[c][j][4 * q], v1 = acc[c][j][4 * q + 1], v2 = acc[c][j][4 * q + 2], v3 = acc[c][j][4 * q + 3];
                quad_transpose(v0, v1, v2, v3, o1, o2);
                // (requesting these rows before the workgroup barrier was measured slower: the loads only queue behind the
                // gathers of the workgroup sharing the CU, and 64 more live registers spill)
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (XPRE) {
                    if (col_ok && orow[jq] >= 0) xv = xpre[c * 8 + jq];
                } else {
                    if (col_ok && orow[jq] >= 0) xv = *reinterpret_cast<const float4*>(fu.xs + (row0 + orow[jq]) * fu.ldxs + col);
                }
                const float sc = F16 ? s_rinv[rt0e + 32 * (jq >> 2) + 8 * (jq & 3)] * winv : 1.0f;
                y[c * 8 + jq][0] = col_ok ? (v0 * sc + b4.x) * alpha + xv.x * (1.0f - alpha) : 0.0f;
                y[c * 8 + jq][1] = col_ok ? (v1 * sc + b4.y) * alpha + xv.y * (1.0f - alpha) : 0.0f;
                y[c * 8 + jq][2] = col_ok ? (v2 * sc + b4.z) * alpha + xv.z * (1.0f - alpha) : 0.0f;
                y[c * 8 + jq][3] = col_ok ? (v3 * sc + b4.w) * alpha + xv.w * (1.0f - alpha) : 0.0f;
            }
        }
        if (fu.use_norm) {
            // a row's columns live in 4 wavefronts x 2 column tiles x 8 lanes: lane-strided sums, one table entry per (row, wave)
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                const int rt = rt0e + 32 * (jq >> 2) + 8 * (jq & 3);
                float ps = y[jq][0] + y[jq][1] + y[jq][2] + y[jq][3] + y[8 + jq][0] + y[8 + jq][1] + y[8 + jq][2] + y[8 + jq][3];
                ps = strided8_sum(ps);
                if (((lane_e & 31) >> 2) == 0) s_sum[rt * 4 + wave] = ps;
            }
            __syncthreads();
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                const int rt = rt0e + 32 * (jq >> 2) + 8 * (jq & 3);
                const float4 a4 = *reinterpret_cast<const float4*>(&s_sum[rt * 4]);
                const float mean = (a4.x + a4.y + a4.z + a4.w) * inv_n;
                float ps = 0.0f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool col_ok = wave * 64 + c * 32 + ((lane_e & 31) >> 2) * 4 < n_out;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[c * 8 + jq][e] -= mean;                                  // centred from here on
                        if (col_ok) ps += y[c * 8 + jq][e] * y[c * 8 + jq][e];
                    }
                }
                ps = strided8_sum(ps);
                if (((lane_e & 31) >> 2) == 0) s_var[rt * 4 + wave] = ps;
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = wave * 64 + c * 32 + ((lane_e & 31) >> 2) * 4;
            const bool col_ok = col < n_out;
            float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (XPRE) {
                if (fu.use_norm && col_ok) { w4 = wpre[c]; c4 = cpre[c]; }
            } else if (fu.use_norm && col_ok) {
                w4 = *reinterpret_cast<const float4*>(fu.lnw + (int64_t)g * n_out + col);
                c4 = *reinterpret_cast<const float4*>(fu.lnb + (int64_t)g * n_out + col);
            }
#pragma unroll
            for (int jq = 0; jq < 8; ++jq) {
                if (!col_ok || orow[jq] < 0) continue;
                float rstd = 1.0f;
                if (fu.use_norm) {
                    const float4 a4 = *reinterpret_cast<const float4*>(&s_var[orow[jq] * 4]);
                    rstd = rsqrtf((a4.x + a4.y + a4.z + a4.w) * inv_n + 1e-5f);
                }
                const float* yy = y[c * 8 + jq];
                *reinterpret_cast<float4*>(fu.out + (row0 + orow[jq]) * n_out + col) =
                    make_float4(yy[0] * rstd * w4.x + c4.x, yy[1] * rstd * w4.y + c4.y, yy[2] * rstd * w4.z + c4.z, yy[3] * rstd * w4.w + c4.w);
            }
        }
        if (fu.use_norm) __syncthreads();   // the tables are rewritten by the next type of a mixed tile
    }
    // rows of unknown type -> 0 (conv.py:120)
    for (int r = wave * 16; r < wave * 16 + 16; ++r) {
        if (row0 + r < NQ && s_type[r] < 0) {
            for (int cidx = lane; cidx < n_out; cidx += 64) fu.out[(row0 + r) * n_out + cidx] = 0.0f;
        }
    }
}

// `vals` = this wavefront's 16 finished rows (gelu applied) in the gather layout
template <int VEC, bool F16 = false>
__device__ __forceinline__ void fused_update_epilogue(const float (&vals)[16][VEC], unsigned char* slab, unsigned char* tables,
                                                      int64_t row0, int64_t NQ, const HgtFusedUpdate& fu) {
    fused_slab_from_rows<VEC, F16>(vals, slab, reinterpret_cast<float*>(tables + FU_RINV_OFF));
    fused_update_tail<VEC, 4, false, F16>(slab, tables, row0, NQ, fu);
}

// Workgroups that contain a hub target cannot finish their rows here (the hub kernels write those rows of agg later):


// the node update of the workgroups k_edge_aggregate_update left pending (their agg rows are complete by now)
template <int VEC, bool F16 = false>
__global__ __launch_bounds__(256, 2) void k_update_pending(const float* __restrict__ agg, int64_t ld_agg, int64_t NQ,
                                                           const int32_t* __restrict__ pending, HgtFusedUpdate fu) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_PLANE + 4096];
    if (pending[fu.q_lo / 64 + blockIdx.x] == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = fu.q_lo + (int64_t)blockIdx.x * 64;
    float vals[16][VEC];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wave * 16 + r;
#pragma unroll
        for (int i = 0; i < VEC; ++i) vals[r][i] = 0.0f;
        if (row < NQ) load_vec<VEC>(agg + row * ld_agg + lane * VEC, vals[r]);
    }
    fused_update_epilogue<VEC, F16>(vals, smem, smem + 2 * A_PLANE, row0, NQ, fu);
}

}  // namespace
