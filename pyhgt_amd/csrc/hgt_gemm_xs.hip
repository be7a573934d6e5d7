// Typed (grouped) linear layer, split x3 variant with the x rows STATIONARY in registers ("xs", round 4).
//
//   y[n, :] = x[n, :] @ W[type(n)]^T + b[type(n)]          K = 64 / 128 / 256 / 512, hundreds of thousands of rows and more
//                                                          (the Q|K|V projection of conv.py:96-97,103; the halo K|V projections)
//
// Why another form.  k_typed_linear_pc (hgt_gemm_bf16x3.hip) keeps the 64-row x slab in LDS and streams the WHOLE split image of W
// (786 KB at d = 256) from L2 into registers once per 64 rows: 12.3 GB per launch at c2 through the CUs' vector-memory pipes, next
// to the 3 GB of output stores and the slab loads, with the waits of all three sharing one in-order counter per wavefront
// (HISTORY.md section 4.2: matrix cores 38 % busy, 1.40 ms against 0.6 ms of MFMA work).  Here the roles are swapped:
//   * a wavefront owns 32 rows for ALL output columns.  Its x rows are loaded from HBM ONCE, straight into MFMA A-fragment
//     order (lane = row, 8 consecutive k per lane and k-chunk), split into hi / lo planes in registers (128 VGPRs at K = 256)
//     and stay there: no LDS slab, no producer wavefronts, no LDS traffic for x at all;
//   * W goes through LDS: the 8 wavefronts of a workgroup (256 rows, two per SIMD) walk the output columns together, 64 columns
//     per step, and the step's 64 KB of B fragments (2 column tiles x hi/lo x K) arrive by LDS-DMA (global_load_lds_dwordx4,
//     1 KB per instruction, no registers) in a two-slot ring, one step ahead: 786 KB of W per 256 rows instead of per 64, and
//     every wavefront's k-loop contains only ds_read_b128 + MFMA (4 reads per 6 MFMAs: a third of the LDS read rate);
//   * one workgroup barrier per step (3072 MFMA cycles per wavefront); the only vector-memory wait sits at the END of a
//     k-loop, when the DMA issued a whole step earlier has long landed, in front of the step's 8 output stores;
//   * the wavefronts are staggered: wavefronts 0-3 run  k-loop -> epilogue -> barrier,  wavefronts 4-7  k-loop -> barrier ->
//     epilogue,  so that one half's stores and bookkeeping fall into the other half's MFMA stream (measured: 2-3 % over lock-step;
//     the other orders behind the `stagger` argument -- other pairings, the LDS-DMA owned by half the wavefronts with counted
//     vmcnt waits, every store behind the step's DMA, non-temporal stores -- are all within +-3 % of it: DESIGN.md section 4);
//   * the rows of the NEXT item are requested inside the last step's k-loop, k-chunk by k-chunk into the registers the chunk
//     just released, and split when the loop is done.
// The accumulation order per output element (k-chunks ascending; lo*hi, hi*lo, hi*hi) is the one of k_typed_linear_pc, so the
// two kernels are BIT-IDENTICAL (tools/bench_xs.py, tests/test_hgt_gpu.py::test_xs_gemm_is_bit_identical_to_the_slab_kernel).
// K = 512 runs as <NKC 32, 4 wavefronts, 1 column tile per step>: the 256 fragment registers of 32 rows need the 512-register budget of
// one wavefront per SIMD.  The row-list partition (units of 128 rows, rounds of one or two units of ONE group, contiguous balanced unit
// ranges per workgroup) is shared with the host-side enumerator hgt_typed_linear_xs_schedule and tested on the CPU.
// Reads the unchanged image of hgt_split_weights[_f16]: a B fragment is one contiguous 1 KB piece of it.
#include "hgt_common.h"
#include "hgt_split_common.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int XS_ROWS = 32;                  // rows of a wavefront's item (one 32x32 MFMA row tile)
constexpr int XS_UNIT = 4 * XS_ROWS;         // rows of a unit: the items of wavefronts 0-3 or 4-7 (one wavefront per SIMD each)
constexpr int XS_PIECE = 1024;               // bytes of one B fragment (64 lanes x 8 bf16)
constexpr int XS_MAXCOL = 3072;              // widest output (3 blocks of 1024 padded columns)

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst + 16 * lane]; M0 is written and restored inside the
// statement (it is compiler-reserved).  Not counted by hipcc: the issuing wavefront waits for it itself (xs_wait_vm).
__device__ __forceinline__ void xs_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// workgroup barrier that orders LDS only (a __syncthreads() would also drain every outstanding global access)
__device__ __forceinline__ void xs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void xs_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// unit u of the concatenated per-group unit lists -> (group, first row position, rows); g = -1 past the end
__host__ __device__ __forceinline__ void xs_unit_lookup(int u, const int32_t* __restrict__ group_off, int n_groups, int& g, int& row0, int& cnt) {
    int before = 0;
    g = -1;
    row0 = 0;
    cnt = 0;
    for (int gg = 0; gg < n_groups; ++gg) {
        const int gb = group_off[gg], ge = group_off[gg + 1];
        const int nu = (ge - gb + XS_UNIT - 1) / XS_UNIT;
        if (g < 0 && u < before + nu) {
            g = gg;
            row0 = gb + (u - before) * XS_UNIT;
            cnt = (ge - row0 < XS_UNIT) ? ge - row0 : XS_UNIT;
        }
        before += nu;
    }
}

struct XsRound {      // up to two units of ONE group (the ring holds one group's W): wavefronts 0-3 take unit A, 4-7 unit B
    int valid, g, used;
    int row0A, cntA, row0B, cntB;      // (scalar fields: an indexed array inside this struct is demoted to LDS / scratch)
};
__host__ __device__ __forceinline__ XsRound xs_round(int u, int u_end, const int32_t* __restrict__ group_off, int n_groups, bool pair) {
    XsRound r;
    r.valid = 0; r.g = 0; r.used = 0;
    r.row0A = r.row0B = 0;
    r.cntA = r.cntB = 0;
    if (u >= u_end) return r;
    int g, row0, cnt;
    xs_unit_lookup(u, group_off, n_groups, g, row0, cnt);
    if (g < 0) return r;
    r.valid = 1; r.g = g; r.used = 1;
    r.row0A = row0; r.cntA = cnt;
    if (pair && u + 1 < u_end) {
        int g2, row2, cnt2;
        xs_unit_lookup(u + 1, group_off, n_groups, g2, row2, cnt2);
        if (g2 == g) { r.used = 2; r.row0B = row2; r.cntB = cnt2; }
    }
    return r;
}

// wavefront `wave` of a round: rows [irow0, irow0 + inrows) of the row list (inrows = 0: idle).  nw = 8: wavefronts 0-3 take unit A,
// 4-7 unit B; nw = 4: unit A only.  Shared by the kernel and the host-side enumerator (hgt_typed_linear_xs_schedule).
__host__ __device__ __forceinline__ void xs_item_of(const XsRound& r, int wave, int nw, int& irow0, int& inrows) {
    const int uh = nw == 8 ? wave >> 2 : 0, wi = wave & 3;
    const int r0 = uh ? r.row0B : r.row0A, c = uh ? r.cntB : r.cntA;
    irow0 = r0 + XS_ROWS * wi;
    int n = c - XS_ROWS * wi;
    n = n > XS_ROWS ? XS_ROWS : n;
    inrows = (r.valid && n > 0) ? n : 0;
}

// the 24-bit transport format of the multi-GPU exchange (hgt_gather_rows_c24): 4 values = 3 dwords, value = 24 bits << 8
__device__ __forceinline__ float4 xs_decode_c24(unsigned w0, unsigned w1, unsigned w2) {
    float4 v;
    v.x = __builtin_bit_cast(float, w0 << 8);
    v.y = __builtin_bit_cast(float, ((w0 >> 24) | (w1 << 8)) << 8);
    v.z = __builtin_bit_cast(float, ((w1 >> 16) | (w2 << 16)) << 8);
    v.w = __builtin_bit_cast(float, w2 & 0xFFFFFF00u);
    return v;
}

// request the 8 values (k0 = kc * 16 + half * 8 .. + 8) of this lane's row for k-chunk kc (K = NKC * 16 exactly: every chunk is
// inside the row, the chunk offsets are immediates of ONE address register pair -- with a per-lane range select the compiler
// precomputed sixteen addresses, spilled them, and every reload in the prefetch loop waited for vmcnt(0)).
// px: the lane's row, already advanced to its half (PROLOGUE 0: floats, + half * 8; PROLOGUE 2: dwords of the wire format, + half * 6)
template <int PROLOGUE>
__device__ __forceinline__ void xs_issue_chunk(float4 (&xr)[2], const float* __restrict__ px, int kc) {
    if constexpr (PROLOGUE == 2) {
        const unsigned* pd = reinterpret_cast<const unsigned*>(px) + kc * 12;
        const uint2 a = *reinterpret_cast<const uint2*>(pd), b = *reinterpret_cast<const uint2*>(pd + 2), c = *reinterpret_cast<const uint2*>(pd + 4);
        xr[0] = make_float4(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.x), __builtin_bit_cast(float, b.y));
        xr[1] = make_float4(__builtin_bit_cast(float, c.x), __builtin_bit_cast(float, c.y), 0.0f, 0.0f);
    } else {
        const float* p = px + kc * 16;
        xr[0] = *reinterpret_cast<const float4*>(p);
        xr[1] = *reinterpret_cast<const float4*>(p + 4);
    }
}

template <int PROLOGUE, int NKC>
__device__ __forceinline__ void xs_issue_all(float4 (&xr)[NKC][2], const float* __restrict__ px) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) xs_issue_chunk<PROLOGUE>(xr[kc], px, kc);
}

// raw rows -> A fragments (hi / lo planes).  F16: one power-of-two scale per row (row maximum -> [2^14, 2^15), the rule of
// k_typed_linear_pc: pc_commit), its inverse into the wavefront's LDS table for the epilogues.
template <int PROLOGUE, bool F16, int NKC>
__device__ __forceinline__ void xs_split(float4 (&xr)[NKC][2], bf16x8 (&ah)[NKC], bf16x8 (&am)[NKC], int lane, float* s_inv_w) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        if constexpr (PROLOGUE == 2) {
            const float4 a = xr[kc][0], b = xr[kc][1];
            xr[kc][0] = xs_decode_c24(__builtin_bit_cast(unsigned, a.x), __builtin_bit_cast(unsigned, a.y), __builtin_bit_cast(unsigned, a.z));
            xr[kc][1] = xs_decode_c24(__builtin_bit_cast(unsigned, a.w), __builtin_bit_cast(unsigned, b.x), __builtin_bit_cast(unsigned, b.y));
        }
    }
    float scale = 1.0f;
    if constexpr (F16) {
        unsigned mb = 0u;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) mb = max(mb, max(abs_bits4(xr[kc][0]), abs_bits4(xr[kc][1])));
        mb = max(mb, (unsigned)__shfl_xor((int)mb, 32));      // the row's other half lives in lane ^ 32
        float inv;
        f16_row_scale(mb, scale, inv);
        if (lane < 32) s_inv_w[lane] = inv;
    }
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        uint2 h0, m0, h1, m1;
        split4_t<F16>(xr[kc][0], scale, h0, m0);
        split4_t<F16>(xr[kc][1], scale, h1, m1);
        ah[kc] = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
        am[kc] = __builtin_bit_cast(bf16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
    }
}

// one step's k-loop: NT 32-column tiles x K out of the ring slot.  Piece order of a k-chunk in the slot: the NT tiles' hi planes,
// then their lo planes.  PREFETCH: the rows of the next item, chunk by chunk behind the MFMAs that consumed the chunk's fragments.
template <int PROLOGUE, bool F16, int NKC, int NT, bool PREFETCH>
__device__ __forceinline__ void xs_kloop(const unsigned char* slot_lane, const bf16x8 (&ah)[NKC], const bf16x8 (&am)[NKC], f32x16 (&acc)[NT],
                                         float4 (&xr)[NKC][2], const float* __restrict__ px_next) {
    bf16x8 bh[2][NT], bm[2][NT];
#define XS_LOAD_B(BUF, KCX)                                                                                          \
    {                                                                                                                \
        const unsigned char* b_ = slot_lane + (KCX) * (2 * NT * XS_PIECE);                                           \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) bh[BUF][j] = *reinterpret_cast<const bf16x8*>(b_ + j * XS_PIECE);          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) bm[BUF][j] = *reinterpret_cast<const bf16x8*>(b_ + (NT + j) * XS_PIECE);   \
    }
    XS_LOAD_B(0, 0)
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < NKC) XS_LOAD_B(cur ^ 1, kc + 1)
        // small terms first, hi*hi last; the accumulators alternate (the order of k_typed_linear_pc: bit-identical results)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mfma32_t<F16>(am[kc], bh[cur][j], acc[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mfma32_t<F16>(ah[kc], bm[cur][j], acc[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mfma32_t<F16>(ah[kc], bh[cur][j], acc[j]);
        if constexpr (PREFETCH) xs_issue_chunk<PROLOGUE>(xr[kc], px_next, kc);
        if (kc + 1 < NKC) __builtin_amdgcn_sched_group_barrier(0x100, 2 * NT, 0);      // the next chunk's DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);                        // the chunk's MFMAs
        if constexpr (PREFETCH) __builtin_amdgcn_sched_group_barrier(0x020, PROLOGUE == 2 ? 3 : 2, 0);   // the freed registers' loads
        __builtin_amdgcn_sched_barrier(0);
    }
#undef XS_LOAD_B
}

typedef float f32x4s __attribute__((ext_vector_type(4)));

// one finished 32 x 32 tile: bias, 4x4 quad transpose (a lane then holds 4 consecutive columns of one row: 16-byte stores,
// one wave instruction = 8 rows x 128 B), row scales of the fp16 split.  Output rows, inverse row scales and the bias come out of
// LDS tables (in registers they would be live across the k-loop: 12 more VGPRs in a kernel that sits at the 256 cap).
template <bool F16>
__device__ __forceinline__ void xs_store_tile(const f32x16& acc, int col, int lane, int n_out, const float* s_bias, float* __restrict__ out0,
                                              float* __restrict__ out1, float* __restrict__ out2, int block_cols, const int* s_orow_w,
                                              const float* s_inv_w, float winv, bool nt) {
    const bool col_ok = col < n_out;
    // at most three output blocks (Q | K | V): two compares instead of a division
    const bool b1 = col >= block_cols, b2 = col >= 2 * block_cols;
    const int cc = col - (b2 ? 2 * block_cols : (b1 ? block_cols : 0));
    float* __restrict__ ob = b2 ? out2 : (b1 ? out1 : out0);
    const bool o1 = lane & 1, o2 = lane & 2;
    const float4 b4 = *reinterpret_cast<const float4*>(s_bias + col);      // (the table is padded to whole 64-column steps)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v0 = acc[4 * q], v1 = acc[4 * q + 1], v2 = acc[4 * q + 2], v3 = acc[4 * q + 3];
        quad_transpose(v0, v1, v2, v3, o1, o2);
        const int rt = (lane & 3) + 8 * q + 4 * (lane >> 5);
        const int orow = s_orow_w[rt];
        float sc = 1.0f;
        if constexpr (F16) sc = s_inv_w[rt] * winv;
        if (col_ok && orow >= 0) {
            const f32x4s r = {v0 * sc + b4.x, v1 * sc + b4.y, v2 * sc + b4.z, v3 * sc + b4.w};
            f32x4s* p = reinterpret_cast<f32x4s*>(ob + (int64_t)orow * block_cols + cc);
            if (nt) __builtin_nontemporal_store(r, p);
            else *p = r;
        }
    }
}

// NW wavefronts of 32 rows each, NT column tiles per step:  <8, 2> for K <= 256 (two wavefronts per SIMD, 256 registers each, 64
// columns per step);  <4, 1> for K = 512 (the 256 fragment registers of a row tile need the 512-register budget of ONE wavefront per
// SIMD; 32 columns per step keep a ring slot at 64 KB).
template <int PROLOGUE, bool F16, int NKC, int NW, int NT>
__global__ __launch_bounds__(64 * NW) void k_typed_linear_xs(
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, const int32_t* __restrict__ group_off, int n_groups,
    int n_out, const unsigned short* __restrict__ wsplit, const float* __restrict__ bias, int64_t bgs, float* __restrict__ out0,
    float* __restrict__ out1, float* __restrict__ out2, int block_cols, int by_pos, int stagger) {
    // ONE shared object (a second one makes hipcc drain vmcnt before LDS reads):
    // [2 ring slots][NKC][2 NT pieces][1 KB] | inverse row scales [NW][32] | output rows [NW][32] | bias [2 round parities][XS_MAXCOL]
    constexpr int XS_THREADS = 64 * NW, CW = 32 * NT;      // columns of a step
    constexpr int PIECES = NKC * 2 * NT, SLOT = PIECES * XS_PIECE;
    constexpr int OFF_INV = 2 * SLOT, OFF_OROW = OFF_INV + NW * XS_ROWS * 4, OFF_BIAS = OFF_OROW + NW * XS_ROWS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_BIAS + 2 * XS_MAXCOL * 4];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int half = lane >> 5;

    int total_units = 0;
    for (int g = 0; g < n_groups; ++g) total_units += (group_off[g + 1] - group_off[g] + XS_UNIT - 1) / XS_UNIT;
    // contiguous, balanced unit ranges (a lone last unit costs half a round: its four wavefronts sit on four different SIMDs)
    int u = (int)((int64_t)blockIdx.x * total_units / gridDim.x);
    const int u_end = (int)((int64_t)(blockIdx.x + 1) * total_units / gridDim.x);
    if (u >= u_end) return;

    const int n_steps = (n_out + CW - 1) / CW;                           // >= 2 (launcher)
    const int n_pass = (n_out + BNP - 1) / BNP;
    const int64_t gimg = (int64_t)n_pass * NKC * 2 * W_PLANE_ELEMS;      // bf16 elements of one group's image
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    float* s_inv_w = reinterpret_cast<float*>(smem + OFF_INV) + wave * XS_ROWS;
    int* s_orow_w = reinterpret_cast<int*>(smem + OFF_OROW) + wave * XS_ROWS;
    if (lane < XS_ROWS) s_inv_w[lane] = 1.0f;
    // the bias of a round's group, zero-padded to whole steps.  Rounds alternate between two tables; the table of round r + 1 is
    // written behind the barrier that ends step 0 of round r: the last reader of that table -- a staggered wavefront's epilogue of
    // the last step of round r - 1 -- ran before that barrier, its first reader comes several barriers later.
    auto stage_bias = [&](int g, int parity) {
        float* tb = reinterpret_cast<float*>(smem + OFF_BIAS) + parity * XS_MAXCOL;
        const int padded = n_steps * CW;
        int tid_l = tid;
        asm volatile("" : "+v"(tid_l));      // (laundered: nothing derived from it is hoisted out of the step loop into live registers)
        for (int c = tid_l; c < padded; c += XS_THREADS) tb[c] = (bias && c < n_out) ? bias[(int64_t)g * bgs + c] : 0.0f;
    };
    const float* winv_tab = reinterpret_cast<const float*>(wsplit + (int64_t)n_groups * gimg);   // fp16 image only: inverse group scales

    // stagger bits 0-1: which wavefronts run  k-loop -> barrier -> epilogue  (0 none; n: bit n - 1 of the wavefront id);
    // bit 2: those wavefronts own the whole LDS-DMA and wait for it with a COUNTED vmcnt that leaves their own output stores (issued
    //        after the DMA) in flight; the other four never wait on the vector-memory counter at all;  bit 3: non-temporal stores;
    // bits 4-6 (timing experiments only, results invalid): no stores / no epilogue / no row loads
    const int pairing = stagger & 3;
    const bool defer = pairing && ((wave >> (pairing - 1)) & 1);
    const bool bdma = pairing && (stagger & 4);
    // 4 alone: EVERY wavefront runs  k-loop -> counted wait -> barrier -> DMA -> epilogue:  a step's LDS-DMA enters the CU's memory
    // pipe ahead of all of the step's output stores (64 KB that drain at ~10 B/clk) and no wait ever covers a store
    const bool defer_all = (stagger & 7) == 4;
    const int drank = pairing ? (((wave >> pairing) << (pairing - 1)) | (wave & ((1 << (pairing - 1)) - 1))) : 0;   // rank among the staggered
    const bool nt_store = stagger & 8;
    const bool dbg_nostore = stagger & 16, dbg_noepi = stagger & 32, dbg_noswitch = stagger & 64;
    // the B fragments of step s of group g -> ring slot: NKC * 4 pieces of 1 KB, NKC / 2 per wavefront
    // (bdma: only the four staggered wavefronts issue -- NKC pieces each -- so that the others never wait on a vector-memory counter)
    auto dma_step = [&](int g, int s, int slot) {
        if (bdma && !defer) return;
        const int col0 = s * CW, pass = col0 >> 8, ct0 = (col0 & 255) >> 5;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));     // (laundered, as in stage_bias)
        const unsigned short* wg = wsplit + (int64_t)g * gimg + (int64_t)pass * NKC * 2 * W_PLANE_ELEMS + lane_l * 8;
        constexpr int PER = PIECES / NW;
        const int first = bdma ? drank * (2 * PER) : wave * PER;
#pragma unroll
        for (int i = 0; i < 2 * PER; ++i) {
            if (i < PER || bdma) {
                const unsigned piece = (unsigned)(first + i);
                const unsigned kc = piece >> (NT == 2 ? 2 : 1), p = piece & (2 * NT - 1);      // (piece = kc * 2 NT + plane * NT + tile)
                const unsigned short* src = wg + (kc * 2 + (p >> (NT - 1))) * W_PLANE_ELEMS + (ct0 + (p & (NT - 1))) * 512;
                xs_glds16(src, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(slot * SLOT + piece * XS_PIECE)));
            }
        }
    };

    // this wavefront's item of a round: rows [irow0, irow0 + inrows) of the row list; inrows = 0: idle (barriers and DMA only)
    const int uh = NW == 8 ? wave >> 2 : 0, wi = wave & 3;
    auto item_of = [&](const XsRound& r, int& irow0, int& inrows) { xs_item_of(r, wave, NW, irow0, inrows); };
    // the lane's source row (advanced to its half); rows beyond the item repeat its last row (never stored).  An idle item reads
    // the first row of the round and is split like any other: no branches, no zero-initialised fragment registers meeting the
    // real ones in 128 phi nodes.
    auto row_ptr = [&](const XsRound& r, int irow0, int inrows) -> const float* {
        const int pos = inrows > 0 ? irow0 + min(lane & 31, inrows - 1) : r.row0A;
        return x + (int64_t)rows[pos] * ldx + half * (PROLOGUE == 2 ? 6 : 8);
    };
    // the output row of every row of the item (-1 = none) into the wavefront's LDS table (read by its own epilogues only)
    auto out_rows = [&](int irow0, int inrows) {
        if (lane < XS_ROWS) s_orow_w[lane] = (lane < inrows) ? (by_pos ? irow0 + lane : rows[irow0 + lane]) : -1;
    };

    XsRound cur = xs_round(u, u_end, group_off, n_groups, NW == 8);
    u += cur.used;
    int irow0, inrows;
    item_of(cur, irow0, inrows);

    // ONE definition site per register array (anything else -- a second split site, a k-loop variant chosen by a branch -- leaves
    // hipcc with 128-register phi webs it resolves through scratch): the raw rows xr are loop-carried (requested here for the first
    // item, inside the last step's k-loop for every later one), the fragments ah / am are made from them at the top of a round.
    float4 xr[NKC][2];
    xs_issue_all<PROLOGUE, NKC>(xr, row_ptr(cur, irow0, inrows));
    dma_step(cur.g, 0, 0);
    stage_bias(cur.g, 0);
    xs_wait_vm();
    xs_barrier();      // step 0 is in slot 0
    dma_step(cur.g, 1, 1);

    // Steady state, step t (global count; slot t & 1):   K(t): k-loop out of slot t & 1  ->  vmcnt(0): this wavefront's pieces of
    // step t + 1 (issued a whole k-loop ago) landed  ->  { barrier t + 1: every wavefront is done with slot t & 1 and step t + 1 is
    // complete; DMA of step t + 2 into slot t & 1 }  and  { epilogue of step t }  in the wavefront's order (defer).
    int t = 0;
    int rpar = 0;      // parity of the round: its bias table
    bool full8 = false;
    while (true) {
        bf16x8 ah[NKC], am[NKC];
        out_rows(irow0, inrows);
        xs_split<PROLOGUE, F16, NKC>(xr, ah, am, lane, s_inv_w);

        const XsRound nxt = xs_round(u, u_end, group_off, n_groups, NW == 8);
        u += nxt.used;
        int nrow0, nnrows;
        item_of(nxt, nrow0, nnrows);
        // the rows the last step's k-loop requests: the next item's; without one (idle next round, or nothing left) its own again
        const float* pn = (nxt.valid && nnrows > 0) ? row_ptr(nxt, nrow0, nnrows) : row_ptr(cur, irow0, inrows);
        const float winv = F16 ? winv_tab[cur.g] : 1.0f;
        const float* s_bias = reinterpret_cast<const float*>(smem + OFF_BIAS) + rpar * XS_MAXCOL;
        auto front = [&](int s) {
            if (s == n_steps - 1 && !nxt.valid) return;      // the very last step: nothing follows
            xs_barrier();
            if (s == 0 && nxt.valid) stage_bias(nxt.g, rpar ^ 1);
            if (s + 2 < n_steps) dma_step(cur.g, s + 2, t & 1);
            else if (nxt.valid) dma_step(nxt.g, s + 2 - n_steps, t & 1);
        };
        auto epilogue = [&](int s, const f32x16 (&acc)[NT]) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass rejects the constraint -- silently, together with the kernel's launch stub)
            if (dbg_noepi) {
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("" : : "v"(acc[j]));      // (a use: the k-loop stays)
            }
#endif
            if (inrows > 0 && !dbg_noepi) {
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));     // (laundered: the epilogue's lane-derived masks / addresses are recomputed here)
                const int colA = s * CW + ((lane_e & 31) >> 2) * 4;
                const int n_eff = dbg_nostore ? 0 : n_out;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    xs_store_tile<F16>(acc[j], colA + 32 * j, lane_e, n_eff, s_bias, out0, out1, out2, block_cols, s_orow_w, s_inv_w, winv, nt_store);
            }
            // exactly 4 NT store instructions were issued? (all four row groups of every column tile have an active lane)
            full8 = inrows > 24 && s * CW + 32 * (NT - 1) < n_out && !dbg_noepi && !dbg_nostore;
        };
        auto tail = [&](int s, const f32x16 (&acc)[NT], auto pf_tag) {
            constexpr bool PF = decltype(pf_tag)::value;
            constexpr int NX = NKC * (PROLOGUE == 2 ? 3 : 2);      // row loads of the prefetching k-loop
            if (defer_all || bdma) {
                if (defer_all || defer) {
                    // outstanding, oldest first: this wavefront's DMA pieces of step t + 1 (wanted), the stores of its last epilogue,
                    // the row loads of a prefetching k-loop.  vmcnt retires in order: leaving exactly the younger ones in flight
                    // proves the pieces have landed.
                    if (full8 && 4 * NT + NX <= 63) {
                        if (PF) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * NT + NX <= 63 ? 4 * NT + NX : 0) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * NT) : "memory");
                    } else {
                        xs_wait_vm();
                    }
                    front(s);
                    epilogue(s, acc);
                } else {
                    epilogue(s, acc);
                    front(s);
                }
            } else {
                xs_wait_vm();
                if (defer) front(s);
                epilogue(s, acc);
                if (!defer) front(s);
            }
            ++t;
        };
        for (int s = 0; s + 1 < n_steps; ++s) {
            f32x16 acc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            if (inrows > 0) xs_kloop<PROLOGUE, F16, NKC, NT, false>(smem + (t & 1) * SLOT + lane * 16, ah, am, acc, xr, nullptr);
            tail(s, acc, std::false_type{});
        }
        {   // the last step of the round: every wavefront runs it (an idle one for its loads only: one definition of xr)
            f32x16 acc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            xs_kloop<PROLOGUE, F16, NKC, NT, true>(smem + (t & 1) * SLOT + lane * 16, ah, am, acc, xr, dbg_noswitch ? x : pn);
            tail(n_steps - 1, acc, std::true_type{});
        }
        if (!nxt.valid) break;
        cur = nxt;
        irow0 = nrow0;
        inrows = nnrows;
        rpar ^= 1;
    }
}

static int xs_grid() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu = v;
    }
    return n_cu;
}

template <int PROLOGUE, bool F16>
static void xs_launch_nkc(int nkc, unsigned grid, hipStream_t stream, const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off,
                          int n_groups, int n_out, const unsigned short* w, const float* bias, int64_t bgs, float* out0, float* out1,
                          float* out2, int block_cols, int by_pos, int stagger) {
#define XS_ARGS x, ldx, rows, group_off, n_groups, n_out, w, bias, bgs, out0, out1, out2, block_cols, by_pos
    if (nkc == 32) k_typed_linear_xs<PROLOGUE, F16, 32, 4, 1><<<grid, 256, 0, stream>>>(XS_ARGS, stagger & ~7);      // (one wavefront per SIMD: no pairs)
    else if (nkc == 16) k_typed_linear_xs<PROLOGUE, F16, 16, 8, 2><<<grid, 512, 0, stream>>>(XS_ARGS, stagger);
    else if (nkc == 8) k_typed_linear_xs<PROLOGUE, F16, 8, 8, 2><<<grid, 512, 0, stream>>>(XS_ARGS, stagger);
    else k_typed_linear_xs<PROLOGUE, F16, 4, 8, 2><<<grid, 512, 0, stream>>>(XS_ARGS, stagger);
#undef XS_ARGS
}

}  // namespace

static unsigned xs_grid_for(int64_t n_rows, int32_t n_groups, int32_t k, int n_cu) {
    const int64_t units = (n_rows + XS_UNIT - 1) / XS_UNIT + n_groups;
    return (unsigned)std::min<int64_t>(std::max<int64_t>(k == 512 ? units : units / 2, 1), n_cu);
}

// The kernel's work decomposition, enumerated on the HOST with the kernel's own helpers (tests/test_xs_schedule.py: every position of
// the row list is covered exactly once, a round never mixes groups, workgroups get contiguous balanced unit ranges).  group_off is a
// HOST array here.  items[i] = {workgroup, round, wavefront, group, first position, rows}; returns HGT_ERR_TOO_LARGE if max_items is
// too small (n_items then holds the count needed).
extern "C" int hgt_typed_linear_xs_schedule(const int32_t* group_off, int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_cu,
                                            int32_t* items, int64_t max_items, int64_t* n_items) {
    if (!group_off || !n_items || n_groups <= 0 || n_cu <= 0 || (max_items > 0 && !items)) return HGT_ERR_INVALID_ARG;
    const int nw = k == 512 ? 4 : 8;
    const unsigned grid = xs_grid_for(n_rows, n_groups, k, n_cu);
    int total_units = 0;
    for (int g = 0; g < n_groups; ++g) total_units += (group_off[g + 1] - group_off[g] + XS_UNIT - 1) / XS_UNIT;
    int64_t n = 0;
    for (unsigned b = 0; b < grid; ++b) {
        int u = (int)((int64_t)b * total_units / grid);
        const int u_end = (int)((int64_t)(b + 1) * total_units / grid);
        for (int round = 0; u < u_end; ++round) {
            const XsRound r = xs_round(u, u_end, group_off, n_groups, nw == 8);
            if (!r.valid) break;
            u += r.used;
            for (int w = 0; w < nw; ++w) {
                int irow0, inrows;
                xs_item_of(r, w, nw, irow0, inrows);
                if (inrows <= 0) continue;
                if (n < max_items) {
                    int32_t* it = items + n * 6;
                    it[0] = (int32_t)b; it[1] = round; it[2] = w; it[3] = r.g; it[4] = irow0; it[5] = inrows;
                }
                ++n;
            }
        }
    }
    *n_items = n;
    return n > max_items ? HGT_ERR_TOO_LARGE : HGT_OK;
}

// 1 = launched, 0 = shape / size outside this kernel's domain (the caller falls back to k_typed_linear_pc), < 0 = error.
// In a wire-format call (prologue 2) ldx / x follow hgt_typed_linear_bf16x3: dwords per row, 8-byte aligned.
int hgt_typed_linear_xs_try(bool f16, const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                            int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias, int64_t bgs, float* out0,
                            float* out1, float* out2, int32_t block_cols, int32_t by_pos, int32_t prologue, void* stream_) {
    // kernel selection travels in the prologue argument like every other switch of the C ABI (include/hgt_hip.h): bit 8 takes this kernel
    // for every eligible shape whatever the row count (tests, tools/bench_xs.py), bit 9 never takes it
    const int mode = (prologue & HGT_LINEAR_NO_XS) ? 0 : ((prologue & HGT_LINEAR_FORCE_XS) ? 1 : -1);
    prologue &= 0xff;
    if (mode == 0) return 0;
    if (prologue != 0 && prologue != 2) return 0;
    if (prologue == 2 && k > KP) return 0;
    const int n_kc = k / KC;
    if (n_out > XS_MAXCOL || n_out <= 64 || (k != 64 && k != 128 && k != 256 && k != 512)) return 0;   // K = NKC * 16 exactly (see xs_issue_chunk)
    // wavefront order: wavefronts 4-7 staggered (3: see the kernel): Q|K|V at c2 1.28 ms against 1.31 in lock-step and 1.45 for the slab
    // kernel.  The other orders measured in round 4 (counted-wait forms, trickled stores: all within +-3 %) live in HISTORY.md section 10
    // and tools/lab/hgt_gemm_xs_trickled.txt, not in the library.
    const int stagger = 3;
    if (prologue == 0 && ((ldx & 3) != 0 || ((uintptr_t)x & 15) != 0)) return 0;
    if (prologue == 2 && (((uintptr_t)x & 7) != 0)) return 0;
    // measured crossover against the slab kernels (tools/bench_xs.py --threshold, profiles/r04_xs_threshold.txt): K <= 256 even at
    // 160-260 k rows, ahead from ~300 k (a full grid needs 65 536 rows per round; below a few rounds the tail costs what the smaller W
    // traffic gains); K = 512 ahead from 66 k rows on (the slab kernel re-splits x per 256-column pass there)
    if (mode != 1 && n_rows < (k == 512 ? 65536 : 262144)) return 0;
    const unsigned grid = xs_grid_for(n_rows, n_groups, k, xs_grid());
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned short* w = (const unsigned short*)w_split;
#define XS_GO(P, F) xs_launch_nkc<P, F>(n_kc, grid, stream, x, ldx, rows, group_off, n_groups, n_out, w, bias, bgs, out0, out1, out2, block_cols, by_pos, stagger)
    if (prologue == 0) { if (f16) XS_GO(0, true); else XS_GO(0, false); }
    else               { if (f16) XS_GO(2, true); else XS_GO(2, false); }
#undef XS_GO
    if (hipGetLastError() != hipSuccess) return HGT_ERR_LAUNCH;
    return 1;
}
