// Typed (grouped) linear layer, split-bf16 x3, "wide" persistent form (round 3).
//
//   y[n, :] = x[n, :] @ W[type(n)]^T + b[type(n)]          (conv.py:96-97,103: the typed Q|K|V projections)
//
// Why a second form next to k_typed_linear_pc (hgt_gemm_bf16x3.hip).  Counters and elimination runs of that kernel at c2
// (DESIGN.md section 4.2): 39 % MFMA-busy, bound by (a) LDS A-fragment reads -- eight consumer wavefronts of 64 rows x 32
// columns each re-read the tile's 4 KB of A fragments per k-chunk, once per 256-column pass: 1.5 MB of LDS reads per tile --
// and (b) output stores that share the in-order vmcnt queue with the W fragment loads.  Both follow from the decomposition
// (12 wavefronts = 8 consumers + 4 producers caps a wavefront at 168 registers = a 64 x 32 accumulator).  Here:
//   * 8 wavefronts per CU and NO producer wavefronts (256-register budget): a wavefront owns 64 rows x 64 columns
//     (4 accumulators), so a k-chunk is 4 A reads + 4 W loads for 12 MFMAs: half the LDS bytes per MFMA, the same W bytes;
//   * the x tile goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, one 1 KB row per instruction, no registers), as RAW
//     fp32 rows (row stride 1040 B: conflict-free ds_read_b128 of fragment-shaped pieces); the bf16 hi/mid split happens on
//     the fly in the consumers (5 VALU per pair, hidden under the MFMAs of the other wavefront of the SIMD).  One wavefront
//     per step has the DMA duty (rotating): its own W waits queue behind the DMA (vmcnt retires in order) while its SIMD
//     partner keeps the matrix pipe busy;
//   * work items are (64-row tile, 64-column pair); a step = 8 consecutive items, one per wavefront.  With 768 = 12 pairs a
//     tile spans 1.5 steps, so two tiles alternate between the two LDS slabs and a tile's DMA runs one step ahead;
//   * finished rows are parked (64 registers) and trickle out one 16 B store per lane per k-chunk of the next item, hidden
//     from the compiler so that its counted vmcnt waits for the W stages stay counted (see hgt_gemm_bf16x3.hip).
// Contract: k <= 256, k % 4 == 0, 16-byte aligned rows, no prologue, >= 8 column pairs (n_out >= 512: Q|K|V and the K|V of
// halo rows); everything else keeps the older kernels.  Same fragment-ordered W image (hgt_split_weights).
#include "hgt_common.h"
#include "hgt_split_common.h"
#include <algorithm>

// Experiment switches (tools/lab builds only; the product build uses the defaults)
#ifndef WD_NST
#define WD_NST 2             // W stages in registers = prefetch distance in k-chunks
#endif
#ifndef WD_X_NOSTORE
#define WD_X_NOSTORE 0       // timing only: no output stores
#endif
#ifndef WD_X_STORE_TOP
#define WD_X_STORE_TOP 0     // parked row stored at the top of a k-chunk instead of its end
#endif
#ifndef WD_X_DUMMY
#define WD_X_DUMMY 0         // n visible 4-byte loads after every hidden store (see WD_DUMMY)
#endif
#ifndef WD_X_NODMA
#define WD_X_NODMA 0         // timing only: only tile 0 is DMA'd
#endif
#ifndef WD_X_NOSPLIT
#define WD_X_NOSPLIT 0       // timing only: raw bits used as fragments
#endif
#ifndef WD_X_NOB
#define WD_X_NOB 0           // timing only: no W refills
#endif
#ifndef WD_X_PRIO
#define WD_X_PRIO 0          // 1: wavefronts 0-3 run at s_setprio 1
#endif
#ifndef WD_X_SKEW
#define WD_X_SKEW 0          // n: wavefronts 4-7 sleep n x 64 cycles after every step barrier (anti-phase with their SIMD partner)
#endif
#ifndef WD_X_STORE_FLAVOUR
#define WD_X_STORE_FLAVOUR 0 // 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
#endif
#ifndef WD_X_SCRATCH_STORE
#define WD_X_SCRATCH_STORE 0 // timing only: every store goes to the same few KB (no HBM write traffic)
#endif
#ifndef WD_X_TRACE
#define WD_X_TRACE 0         // s_memtime phase totals of wavefronts 0 and 4 of workgroup 0 into hgt_wd_trace
#endif
#ifndef WD_SUFFIX
#define WD_SUFFIX
#endif
#define WD_CAT2(a, b) a##b
#define WD_CAT(a, b) WD_CAT2(a, b)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WD_WAVES = 8, WD_THREADS = 64 * WD_WAVES;
constexpr int WD_ROW = KP * 4 + 16;             // LDS bytes per raw fp32 x row: 260 dwords -> 16 rows hit 16 distinct 4-bank slots
constexpr int WD_SLAB = BM * WD_ROW;            // 66560
constexpr int WD_RID = 2 * WD_SLAB;             // int row_id[4][BM]: row ids of the four most recent tiles
constexpr int WD_LDS = WD_RID + 4 * BM * 4 + 64;   // + slack: the k-chunk prefetch past the last chunk reads (unused) bytes behind a slab

struct WdRaw { f32x4 a0, b0, a1, b1; };         // raw fp32 A pieces of one k-chunk: row tile 0 (k 0-3, 4-7), row tile 1
struct WdFrag { bf16x8 h0, m0, h1, m1; };       // A fragments: row tiles 0/1, hi/mid
struct WdB { bf16x8 h0, m0, h1, m1; };          // B fragments: column tiles 0/1, hi/mid

struct WdItem { int valid, tile, g, row0, nrows, ct0; };

struct WdPending {        // one item's finished output of this lane: 16 x (row, 4 consecutive columns); u = c * 8 + j * 4 + q
    f32x4 v[16];
    float* base[2];       // per column tile: block pointer + column offset
    uint64_t cols[2];     // per column tile: lanes whose columns exist (0 while nothing is parked)
    unsigned ld;
    const int* rid;       // LDS table of the output rows of the tile these rows belong to
};

// stores the compiler does not see (see hidden_store16 in hgt_gemm_bf16x3.hip), predicated INSIDE the statement with the lane
// mask of the rows that exist: an `if` around the store would cut every k-chunk into its own basic block, and hipcc's machine
// sinking then moves the split of the next chunk's A fragments down into the block that uses them -- in front of its MFMAs,
// where the matrix pipe drains for ~45 VALU instructions (seen in the ISA).  s_nop 1: the data registers must not be
// overwritten by the next instruction before the store has read them.
__device__ __forceinline__ void wd_hidden_store16(float* p, f32x4 v, uint64_t lanes) {
    uint64_t saved;
#if WD_X_STORE_FLAVOUR == 1
#define WD_ST_FL " nt"
#elif WD_X_STORE_FLAVOUR == 2
#define WD_ST_FL " sc1"
#elif WD_X_STORE_FLAVOUR == 3
#define WD_ST_FL " sc0 sc1"
#else
#define WD_ST_FL ""
#endif
    asm volatile("s_and_saveexec_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, off" WD_ST_FL "\n\ts_nop 1\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(p), "v"(v), "s"(lanes)
                 : "memory");
}

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst + 16 * lane]; M0 is written and restored inside the
// statement (it is compiler-reserved).  Not counted by hipcc: the issuing wavefront waits vmcnt(0) itself before the hand-over.
__device__ __forceinline__ void wd_glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ void wd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bool wd_tile_lookup(int t, const int32_t* __restrict__ group_off, int n_groups, int& g, int& row0, int& nrows) {
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

// one wavefront: row ids of a tile into the LDS table, its 64 rows HBM -> slab (rows beyond the tile repeat its last row:
// their products are never stored).  Lanes beyond k leave their (zero-initialised) 16 bytes alone.
__device__ __forceinline__ void wd_dma_tile(unsigned char* smem, unsigned lds_base, int slab, int slot, int row0, int nrows,
                                            const int32_t* __restrict__ rows, const float* __restrict__ x, int64_t ldx, int k, int lane,
                                            int by_pos) {
    const int rid = rows[row0 + min(lane, nrows - 1)];
    // the table holds the OUTPUT row of every tile row (its position in the row list or its node id), -1 beyond the tile
    reinterpret_cast<int*>(smem + WD_RID)[slot * BM + lane] = (lane < nrows) ? (by_pos ? row0 + lane : rid) : -1;
    const unsigned dst0 = lds_base + (unsigned)slab * WD_SLAB;
    if (lane * 4 < k) {
#pragma unroll 8
        for (int r = 0; r < BM; ++r) {
            const int rr = __builtin_amdgcn_readlane(rid, r);
            wd_glds16(x + (int64_t)rr * ldx + lane * 4, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)r * WD_ROW));
        }
    }
}

template <int NKC>
__global__ __launch_bounds__(WD_THREADS, 2) void WD_CAT(k_typed_linear_wide, WD_SUFFIX)(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off, int n_groups,
    int k, int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs, float* __restrict__ out0,
    float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos) {
    // ONE shared object: a second one makes hipcc drain vmcnt before LDS reads (cdna_hip_programming.md, GEMM traps)
    __shared__ __attribute__((aligned(16))) unsigned char smem[WD_LDS];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int total_tiles = 0;
    for (int g = 0; g < n_groups; ++g) total_tiles += (group_off[g + 1] - group_off[g] + BM - 1) / BM;
    const int first = blockIdx.x, stride = gridDim.x;
    const int n_mine = (total_tiles > first) ? (total_tiles - first + stride - 1) / stride : 0;
    if (n_mine == 0) return;

    const int n_ct = (n_out + 31) / 32;           // 32-column tiles
    const int U = (n_ct + 1) / 2;                 // 64-column pairs per row tile (>= 8, checked by the launcher)
    const int n_items = n_mine * U;
    const int n_steps = (n_items + WD_WAVES - 1) / WD_WAVES;
    const int n_pass = (n_out + BNP - 1) / BNP;
    constexpr int n_kc = NKC;                     // k-chunks of 16, padded to a multiple of 4 (zero tiles in the W image)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // zero both slabs: columns >= k are never written by the DMA, and stale LDS bytes may be NaN patterns
    for (int o = tid * 16; o < 2 * WD_SLAB; o += WD_THREADS * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if (wave == 0) {                              // tile 0 (the only tile whose first step is 0)
        int g, row0, nrows;
        wd_tile_lookup(first, group_off, n_groups, g, row0, nrows);
        wd_dma_tile(smem, lds_base, 0, 0, row0, nrows, rows, x, ldx, k, lane, by_pos);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    auto get_item = [&](int s) {
        WdItem it;
        const int q = s * WD_WAVES + wave;
        it.valid = (s < n_steps) && (q < n_items);
        it.tile = it.valid ? q / U : 0;
        it.ct0 = it.valid ? 2 * (q - it.tile * U) : 0;
        it.g = it.row0 = it.nrows = 0;
        if (it.valid) wd_tile_lookup(first + it.tile * stride, group_off, n_groups, it.g, it.row0, it.nrows);
        return it;
    };
    // B fragments of (group g, column tile ct): [g][pass][k-chunk][plane][column tile 8][lane][8] (hgt_split_weights)
    // -> wave-uniform BYTE offset of the fragments of (g, ct) inside the W image (SGPRs); the lane's 16 bytes sit at + 16 * lane,
    // so that a load is "SGPR base + 32-bit VGPR offset" with no per-lane 64-bit address arithmetic
    auto wptr = [&](int g, int ct) {
        const uint64_t a = 2 * (((uint64_t)(g * n_pass + (ct >> 3)) * n_kc * 2) * W_PLANE_ELEMS + (uint64_t)((ct & 7) * 64) * 8);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<const unsigned char*>(wsplit) + (((uint64_t)hi << 32) | lo);
    };
    const unsigned wlane = (unsigned)lane * 16u;

    const int frow = lane & 31, khalf = lane >> 5;
    f32x16 acc[4];                                // [j * 2 + c]: row tile j, column tile c; the first k-chunk starts them from 0
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    WdPending pr;
#pragma unroll
    for (int u = 0; u < 16; ++u) pr.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    pr.base[0] = pr.base[1] = nullptr;
    pr.cols[0] = pr.cols[1] = 0;
    pr.ld = 0;
    pr.rid = reinterpret_cast<const int*>(smem + WD_RID);
    bool have_pend = false;                       // wave-uniform

    // parked row UU of the previous item: the row id comes out of the LDS table (read unconditionally and FIRST in a k-chunk, so
    // that the wait for it is a counted lgkmcnt behind the chunk's A reads, not a drain), the store is predicated
    const int rt0 = (lane & 3) + 4 * (lane >> 5);
#define WD_ROWOFF(UU) ((((UU) >> 2) & 1) * 32 + 8 * ((UU)&3))
#define WD_ROWID(UU) const int rid_c = pr.rid[rt0 + WD_ROWOFF(UU)];
#define WD_STORE_R(UU)                                                                                \
    {                                                                                                 \
        float* b_ = pr.base[(UU) >> 3];                                                               \
        const uint64_t m_ = __builtin_amdgcn_ballot_w64(rid_c >= 0) & pr.cols[(UU) >> 3];             \
        float* p_ = WD_X_SCRATCH_STORE ? out0 + (threadIdx.x & 511) * 4 : b_ + (uint64_t)(unsigned)rid_c * pr.ld; \
        wd_hidden_store16(p_, pr.v[UU], m_);                                                          \
    }
#define WD_STORE(UU) { WD_ROWID(UU) WD_STORE_R(UU) }
#define WD_LOAD_B(ST, KCX)                                                                            \
    {                                                                                                 \
        const unsigned o_ = (unsigned)min((KCX), n_kc - 1) * (2 * W_PLANE_ELEMS * 2);                 \
        bst[ST].h0 = *reinterpret_cast<const bf16x8*>(w0 + o_ + wlane);                               \
        bst[ST].m0 = *reinterpret_cast<const bf16x8*>(w0 + o_ + W_PLANE_ELEMS * 2 + wlane);           \
        bst[ST].h1 = *reinterpret_cast<const bf16x8*>(w1 + o_ + wlane);                               \
        bst[ST].m1 = *reinterpret_cast<const bf16x8*>(w1 + o_ + W_PLANE_ELEMS * 2 + wlane);           \
    }
#define WD_LOAD_RAW(KCX)                                                                              \
    {                                                                                                 \
        const unsigned char* p_ = aptr + min((KCX), n_kc - 1) * (KC * 4);                             \
        raw.a0 = *reinterpret_cast<const f32x4*>(p_);                                                 \
        raw.b0 = *reinterpret_cast<const f32x4*>(p_ + 16);                                            \
        raw.a1 = *reinterpret_cast<const f32x4*>(p_ + 32 * WD_ROW);                                   \
        raw.b1 = *reinterpret_cast<const f32x4*>(p_ + 32 * WD_ROW + 16);                              \
    }
#if WD_X_NOSPLIT
#define WD_SPLIT(F)                                                                                   \
    {                                                                                                 \
        fr[F].h0 = __builtin_bit_cast(bf16x8, raw.a0); fr[F].m0 = __builtin_bit_cast(bf16x8, raw.b0); \
        fr[F].h1 = __builtin_bit_cast(bf16x8, raw.a1); fr[F].m1 = __builtin_bit_cast(bf16x8, raw.b1); \
    }
#else
#define WD_SPLIT(F)                                                                                   \
    {                                                                                                 \
        uint2 h_, m_, h2_, m2_;                                                                       \
        split4(make_float4(raw.a0.x, raw.a0.y, raw.a0.z, raw.a0.w), h_, m_);                          \
        split4(make_float4(raw.b0.x, raw.b0.y, raw.b0.z, raw.b0.w), h2_, m2_);                        \
        fr[F].h0 = __builtin_bit_cast(bf16x8, make_uint4(h_.x, h_.y, h2_.x, h2_.y));                  \
        fr[F].m0 = __builtin_bit_cast(bf16x8, make_uint4(m_.x, m_.y, m2_.x, m2_.y));                  \
        split4(make_float4(raw.a1.x, raw.a1.y, raw.a1.z, raw.a1.w), h_, m_);                          \
        split4(make_float4(raw.b1.x, raw.b1.y, raw.b1.z, raw.b1.w), h2_, m2_);                        \
        fr[F].h1 = __builtin_bit_cast(bf16x8, make_uint4(h_.x, h_.y, h2_.x, h2_.y));                  \
        fr[F].m1 = __builtin_bit_cast(bf16x8, make_uint4(m_.x, m_.y, m2_.x, m2_.y));                  \
    }
#endif
    // 12 MFMAs of one k-chunk: small terms first, hi*hi last; every accumulator is touched once per group of four
#define WD_MFMA(F, ST, FIRST)                                                                                           \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].m0, bst[ST].h0, (FIRST) ? zero16 : acc[0], 0, 0, 0);          \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].m1, bst[ST].h0, (FIRST) ? zero16 : acc[2], 0, 0, 0);          \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].m0, bst[ST].h1, (FIRST) ? zero16 : acc[1], 0, 0, 0);          \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].m1, bst[ST].h1, (FIRST) ? zero16 : acc[3], 0, 0, 0);          \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h0, bst[ST].m0, acc[0], 0, 0, 0);                             \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h1, bst[ST].m0, acc[2], 0, 0, 0);                             \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h0, bst[ST].m1, acc[1], 0, 0, 0);                             \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h1, bst[ST].m1, acc[3], 0, 0, 0);                             \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h0, bst[ST].h0, acc[0], 0, 0, 0);                             \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h1, bst[ST].h0, acc[2], 0, 0, 0);                             \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h0, bst[ST].h1, acc[1], 0, 0, 0);                             \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[F].h1, bst[ST].h1, acc[3], 0, 0, 0);
    // k-chunk KCX (a literal): raw holds chunk KCX+1 -> split it into the other fragment set, request chunk KCX+2, the chunk's
    // MFMAs, refill of the W stage just consumed, one parked row of the previous item (its row id was requested a chunk ahead)
#if WD_X_DUMMY
    // a VISIBLE 4-byte load behind every hidden store: hipcc's counted vmcnt waits see only its own loads, so a pending hidden
    // store makes every wait stricter by one load -- it then waits for a W fragment requested moments ago.  The dummy load is
    // counted (one more younger load allowed) and, being older than the W refills, it retires right behind the stage waited for.
#define WD_DUMMY(KCX)                                                                                 \
    _Pragma("unroll") for (int d_ = 0; d_ < WD_X_DUMMY; ++d_) {                                       \
        dsink ^= dval[((KCX) + 1) & 1][d_];                                                           \
        dval[(KCX)&1][d_] = *reinterpret_cast<const volatile int*>(dptr + d_ * 64);                   \
    }
#else
#define WD_DUMMY(KCX)
#endif
#define WD_STORE_AT(KCX)                                                                              \
        WD_STORE_R(KCX)                                                                               \
        WD_DUMMY(KCX)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);
#if WD_X_STORE_TOP
#define WD_STORE_TOP(KCX) WD_STORE_AT(KCX)
#define WD_STORE_END(KCX)
#else
#define WD_STORE_TOP(KCX)
#define WD_STORE_END(KCX) WD_STORE_AT(KCX)
#endif
#if WD_X_NOB
#define WD_REFILL_B(KCX)
#else
#define WD_REFILL_B(KCX) WD_LOAD_B((KCX) % WD_NST, (KCX) + WD_NST)
#endif
#define WD_CHUNK(KCX)                                                                                 \
    {                                                                                                 \
        const int rid_c = rid_nx;                                                                     \
        rid_nx = pr.rid[rt0 + WD_ROWOFF(((KCX) + 1) & 15)];                                           \
        WD_STORE_TOP(KCX)                                                                             \
        /* the split's inputs are made opaque HERE: pure arithmetic has no chain, and instruction selection otherwise  */ \
        /* places it above the previous chunk's scheduling barrier, outside the region the groups below apply to       */ \
        asm volatile("" : "+v"(raw.a0), "+v"(raw.b0), "+v"(raw.a1), "+v"(raw.b1));                    \
        WD_SPLIT(((KCX) + 1) & 1)                                                                     \
        WD_LOAD_RAW((KCX) + 2)                                                                        \
        WD_MFMA((KCX)&1, (KCX) % WD_NST, (KCX) == 0)                                                           \
        WD_REFILL_B(KCX)                                                                              \
        /* pin: next row id, then the split's VALU work spread between the MFMAs (left alone, hipcc runs the ~45 VALU  */ \
        /* instructions of the split as one block in front of the MFMAs and the matrix pipe drains meanwhile), then    */ \
        /* the W refill and the next A reads                                                                           */ \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                            \
        WD_SCHED4 WD_SCHED4 WD_SCHED4                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        WD_STORE_END(KCX)                                                                             \
    }
#define WD_SCHED1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
#define WD_SCHED4 WD_SCHED1 WD_SCHED1 WD_SCHED1 WD_SCHED1
#define WD_BODY(B)                                                                                    \
    WD_CHUNK(4 * (B))                                                                                 \
    WD_CHUNK(4 * (B) + 1)                                                                             \
    WD_CHUNK(4 * (B) + 2)                                                                             \
    WD_CHUNK(4 * (B) + 3)

    WdRaw raw;
    WdFrag fr[2];
    WdB bst[WD_NST];
    int rid_nx = 0;
#if WD_X_DUMMY
    int dsink = 0, dval[2][WD_X_DUMMY] = {};
    const int* dptr = reinterpret_cast<const int*>(wsplit) + lane;      // any resident, readable address
#endif
    WdItem it = get_item(0);
    const unsigned char* w0 = wptr(it.g, it.ct0);
    const unsigned char* w1 = wptr(it.g, it.ct0 + 1);
#define WD_FIRST_STAGES                                                                               \
    if (it.valid) {                                                                                   \
        _Pragma("unroll") for (int st_ = 0; st_ < WD_NST; ++st_) WD_LOAD_B(st_, st_)                  \
    }
    WD_FIRST_STAGES

#if WD_X_PRIO
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int s = 0; s < n_steps; ++s) {
        wd_barrier();                             // everything requested during step s-1 has landed (its issuer waited)
#if WD_X_SKEW
        if (wave >= 4) __builtin_amdgcn_s_sleep(WD_X_SKEW);
#endif
        const bool duty = (wave == (s & (WD_WAVES - 1)));
        bool dma_out = false;
        if (duty) {                               // the tile whose first step is s+1 (its slab was released by the barrier above)
            const int jc = (WD_WAVES * (s + 1) + U - 1) / U;
            if (!WD_X_NODMA && jc < n_mine && jc * U < WD_WAVES * (s + 2)) {
                int g, row0, nrows;
                wd_tile_lookup(first + jc * stride, group_off, n_groups, g, row0, nrows);
                wd_dma_tile(smem, lds_base, jc & 1, jc & 3, row0, nrows, rows, x, ldx, k, lane, by_pos);
                dma_out = true;
            }
        }
        if (it.valid) {
            // this item's bias columns (consumed when the rows are parked)
            float4 b4[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = (it.ct0 + c) * 32 + ((lane & 31) >> 2) * 4;
                b4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias != nullptr && col < n_out) b4[c] = *reinterpret_cast<const float4*>(bias + (int64_t)it.g * bgs + col);
            }
            const unsigned char* aptr = smem + (it.tile & 1) * WD_SLAB + frow * WD_ROW + khalf * 32;
            rid_nx = pr.rid[rt0 + WD_ROWOFF(0)];
            WD_LOAD_RAW(0)
            WD_SPLIT(0)
            WD_LOAD_RAW(1)
            WD_BODY(0)
            if constexpr (NKC > 4) { WD_BODY(1) }
            if constexpr (NKC > 8) { WD_BODY(2) }
            if constexpr (NKC > 12) { WD_BODY(3) }
            // k < 256: the chunks that did not run leave their rows behind
            if constexpr (NKC <= 12) { WD_STORE(12) WD_STORE(13) WD_STORE(14) WD_STORE(15) }
            if constexpr (NKC <= 8) { WD_STORE(8) WD_STORE(9) WD_STORE(10) WD_STORE(11) }
            if constexpr (NKC <= 4) { WD_STORE(4) WD_STORE(5) WD_STORE(6) WD_STORE(7) }
            // park the finished rows: bias, 4x4 quad transpose (a lane ends up with 4 consecutive columns of one row)
            const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = (it.ct0 + c) * 32 + ((lane & 31) >> 2) * 4;
                const bool col_ok = col < n_out;
                const int blk = col_ok ? col / block_cols : 0, cc = col - blk * block_cols;
                float* ob = (blk == 0) ? out0 : ((blk == 1) ? out1 : out2);
                pr.base[c] = ob + cc;
                pr.cols[c] = WD_X_NOSTORE ? 0ull : __builtin_amdgcn_ballot_w64(col_ok);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v0 = acc[j * 2 + c][4 * q], v1 = acc[j * 2 + c][4 * q + 1], v2 = acc[j * 2 + c][4 * q + 2],
                              v3 = acc[j * 2 + c][4 * q + 3];
                        quad_transpose(v0, v1, v2, v3, o1, o2);
                        pr.v[c * 8 + j * 4 + q] = f32x4{v0 + b4[c].x, v1 + b4[c].y, v2 + b4[c].z, v3 + b4[c].w};
                    }
                }
            }
            pr.ld = (unsigned)block_cols;
            pr.rid = reinterpret_cast<const int*>(smem + WD_RID) + (it.tile & 3) * BM;
            have_pend = true;
        } else if (have_pend) {                   // no item in this step (tail): the parked rows leave now
            WD_STORE(0) WD_STORE(1) WD_STORE(2) WD_STORE(3) WD_STORE(4) WD_STORE(5) WD_STORE(6) WD_STORE(7)
            WD_STORE(8) WD_STORE(9) WD_STORE(10) WD_STORE(11) WD_STORE(12) WD_STORE(13) WD_STORE(14) WD_STORE(15)
            have_pend = false;
            pr.cols[0] = pr.cols[1] = 0;
        }
        // the next item's first W stages are requested BEFORE the barrier (they do not depend on the slab)
        it = get_item(s + 1);
        w0 = wptr(it.g, it.ct0);
        w1 = wptr(it.g, it.ct0 + 1);
        WD_FIRST_STAGES
        if (dma_out) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // the rows of the very last item
    WD_STORE(0) WD_STORE(1) WD_STORE(2) WD_STORE(3) WD_STORE(4) WD_STORE(5) WD_STORE(6) WD_STORE(7)
    WD_STORE(8) WD_STORE(9) WD_STORE(10) WD_STORE(11) WD_STORE(12) WD_STORE(13) WD_STORE(14) WD_STORE(15)
#if WD_X_DUMMY
    asm volatile("" ::"v"(dsink));
#endif
#undef WD_BODY
#undef WD_FIRST_STAGES
#undef WD_STORE_AT
#undef WD_STORE_TOP
#undef WD_STORE_END
#undef WD_REFILL_B
#undef WD_DUMMY
#undef WD_ROWOFF
#undef WD_CHUNK
#undef WD_MFMA
#undef WD_SPLIT
#undef WD_LOAD_RAW
#undef WD_LOAD_B
#undef WD_STORE
#undef WD_STORE_R
#undef WD_ROWID
#undef WD_SCHED1
#undef WD_SCHED4
}

}  // namespace

// Launcher used by hgt_typed_linear_bf16x3 (hgt_gemm_bf16x3.hip).  Returns HGT_ERR_UNSUPPORTED when the shape is outside
// this kernel's contract (the caller then takes the older kernel), HGT_OK after a launch.
int WD_CAT(hgt_launch_typed_linear_wide, WD_SUFFIX)(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                 int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                 int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                 int32_t out_by_position, int n_cu, hipStream_t stream) {
    if (k > KP || (k & 3) != 0 || (ldx & 3) != 0 || (((uintptr_t)x) & 15) != 0) return HGT_ERR_UNSUPPORTED;
    if ((n_out & 3) != 0 || (block_cols & 31) != 0) return HGT_ERR_UNSUPPORTED;
    if (bias != nullptr && ((((uintptr_t)bias) & 15) != 0 || (b_group_stride & 3) != 0)) return HGT_ERR_UNSUPPORTED;
    const int n_ct = (n_out + 31) / 32, U = (n_ct + 1) / 2;
    if (U < WD_WAVES) return HGT_ERR_UNSUPPORTED;
    // two slabs suffice iff tile j-2 is finished a full step before tile j's first step (its DMA runs during the step between)
    for (int j = 2; j < 2 + 2 * WD_WAVES; ++j)
        if (((j - 1) * U - 1) / WD_WAVES > (j * U) / WD_WAVES - 2) return HGT_ERR_UNSUPPORTED;
    const int64_t row_tiles = (n_rows + BM - 1) / BM + n_groups;
    if (row_tiles > 0x7fffffffLL) return HGT_ERR_TOO_LARGE;
    const unsigned grid = (unsigned)std::min<int64_t>(row_tiles, n_cu);
    const int n_kc = ((k + KC - 1) / KC + 3) & ~3;
#define WD_LAUNCH(NKC)                                                                                                       \
    WD_CAT(k_typed_linear_wide, WD_SUFFIX)<NKC><<<grid, WD_THREADS, 0, stream>>>(x, ldx, rows, group_off, n_groups, k, n_out, (const unsigned short*)w_split, \
                                                              bias, b_group_stride, out0, out1, out2, block_cols, out_by_position)
    switch (n_kc) {
        case 4: WD_LAUNCH(4); break;
        case 8: WD_LAUNCH(8); break;
        case 12: WD_LAUNCH(12); break;
        default: WD_LAUNCH(16); break;
    }
#undef WD_LAUNCH
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
