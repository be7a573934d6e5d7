// Ring form of the fused aggregation + node-update kernel (round 5): d = 256 / 8 heads (VEC = 4, LPH = 8), no temporal rows, bf16 split.
// Included by hgt_edge_agg_mfma.hip (part VEC = 4 / RTE = 0 / F16 = 0) inside its anonymous namespace in LAB builds only (make LAB=1;
// selected by HGT_FLAG_RING_AGGREGATE):
// bit-identical to k_edge_aggregate_update_mfma, measured 3.57 ms against 3.27 at the benchmark size -- kept as the record of the
// experiment the round-4 review asked for and as the A/B partner of the default kernel (counters: profiles/r05_agg_counters.txt).
//
// The idea.  A wavefront of the default kernel keeps 2 x 4 gathered rows in flight -- all its registers allow next to 64 accumulators, 64
// fragment registers and the row buffers -- and every relation end fetches 32 KB of fragments behind the rows in flight.  Here the on-chip
// capacity is re-assigned:
//   * gathered rows never touch registers before they are used: a wavefront owns an LDS RING of RG_D slots of 1 KB, a row arrives by
//     LDS-DMA (global_load_lds_dwordx4: scalar base + one lane offset, no VGPR) and is read back with one ds_read_b128 one row ahead of
//     its use; the slot is refilled at once -- RG_D - 1 rows in flight per wavefront across segment, relation and chunk boundaries, and
//     past the end of the stream (the last row is re-requested), so that every wait for a row is the SAME s_waitcnt immediate;
//   * the LDS for the ring comes from the U tile: finished segment rows (bf16 hi / mid) wait in a 4-row park buffer and reach the
//     REGISTERS of the lanes that hold their targets' columns of the MFMA B operand through exec-masked ds_read_b128 (64 VGPRs per lane
//     hold the 16 x 256 tile);
//   * half of a relation's fragments (64 VGPRs: all 32 would be 128, the whole file next to accumulators and U tile) is requested when
//     the walk ENTERS the relation; the other half after the first half's products: one exposed round trip that also drains the ring
//     (measured: hidden by the other wavefronts);
//   * edge ids and logits of 64 stream entries arrive by LDS-DMA as well (4 instructions per chunk).
// Nothing in the walk is a compiler-visible vector-memory access: hipcc emits no vmcnt wait inside it, the waits are the kernel's own.
//
// What was learnt (the reason it is not the default): neither kernel is bound by memory latency or by the fragment traffic.  Both are
// bound by INSTRUCTION ISSUE -- two wavefronts per SIMD retire about one instruction per 5 cycles whatever the mix, and the time follows
// the instruction count (ring v1: 190 instructions per row, 4.55 ms; this form: 127, 3.57 ms; default kernel: 113, 3.27 ms).  The ring
// removes the waits it was built to remove and pays for it with per-row bookkeeping: a single-row loop has no instruction-level
// parallelism across rows and ~60 instructions of fetch / issue / index work per row (1.85 ms with everything else switched off), where
// the default kernel's batches of four amortise theirs.
#pragma once

#ifndef RG_OFF
#define RG_OFF 0      // timing-only: bit mask of parts left out -- 1 node update, 2 register fills of parked rows, 4 parking (split + LDS
#endif                // writes + fills), 8 relation transforms, 16 waits for rows, 32 the row DMA instruction, 64 exp + fma, 128 gathers hit row 0
#ifndef RG_DEPTH
#define RG_DEPTH 8
#endif
constexpr int RG_D = RG_DEPTH;                            // ring slots: RG_D - 1 rows in flight per wavefront + the one being read back
constexpr int RG_RING = RG_D * 1024;
constexpr int RG_LOG = 2 * 2048;                          // logits of two 64-entry chunks ([2 halves of the heads][64 entries][4 floats])
constexpr int RG_META = 2 * 512;                          // source ids | target ids of two chunks
constexpr int RG_PARK = 4;                                // parked rows per fill of the U registers
constexpr int RG_BNC = RG_PARK * 1024;                    // park buffer: RG_PARK x (hi plane | mid plane)
constexpr int RG_PEND = 512;                              // pending scales of the accumulator columns, [16 targets][8 heads]
constexpr int RG_WAVE = RG_RING + RG_LOG + RG_META + RG_BNC + RG_PEND;
constexpr int RG_FRONT = 4 * RG_WAVE > 2 * A_PLANE ? 4 * RG_WAVE : 2 * A_PLANE;      // the epilogue's A slab overlays the wavefronts' regions
constexpr int RG_STATE = 4 * 1024;                        // softmax reference | exp-sum per (target, head), [16][8] floats each, per wavefront
constexpr int RG_SMEM = RG_FRONT + RG_STATE + 4 * 16 * 4;
static_assert(RG_STATE >= FU_RINV_OFF + 256, "the epilogue's tables overlay the softmax state");
static_assert(2 * RG_SMEM <= 160 * 1024, "two workgroups per CU");
static_assert(RG_D >= 4 && RG_D <= 16, "ring depth");

// LDS-DMA, 64 lanes x 16 B: global [sbase + voff(lane)] -> LDS [lds_dst + 16 lane] (voff: 32-bit unsigned byte offset per lane)
__device__ __forceinline__ void rg_dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// per-lane 64-bit addresses: 16 B / 4 B per lane
__device__ __forceinline__ void rg_dma16_v(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void rg_dma4_v(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// WIDE: the rows of V span 4 GB or more (row base = 64-bit scalar address); otherwise one 32-bit lane offset from V
template <bool WIDE>
__global__ __launch_bounds__(256, 2) void k_edge_aggregate_update_ring(
    const int32_t* __restrict__ segptr, const int32_t* __restrict__ esrc, const int32_t* __restrict__ edst,
    const float* __restrict__ logits, const float* __restrict__ V, const unsigned short* __restrict__ msgF, int R, int64_t NQ,
    const int32_t* __restrict__ hub_slot, int32_t* __restrict__ pending, HgtFusedUpdate fu) {
    constexpr int DKP = 32, NCT = 16;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RG_SMEM];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row0 = fu.q_lo + (int64_t)blockIdx.x * 64;
    const int64_t wrow0 = row0 + wib * 16;

    unsigned hub_mask = 0;
    if (hub_slot) {
        const int64_t rr = wrow0 + (lane & 15);
        const bool is_hub = (lane < 16) && (rr < NQ) && (hub_slot[rr] >= 0);
        hub_mask = (unsigned)(__builtin_amdgcn_ballot_w64(is_hub) & 0xFFFFull);
    }
    const bool any_hub = hub_slot ? (__syncthreads_or(hub_mask != 0) != 0) : false;
    if (threadIdx.x == 0) pending[fu.q_lo / 64 + blockIdx.x] = any_hub ? 1 : 0;      // (absolute workgroup index: target blocks may run concurrently)
    if (any_hub) return;      // k_edge_aggregate_hub_workgroups walks the 64 targets of such a workgroup (same launcher)

    unsigned char* const wbase = smem + wib * RG_WAVE;
    unsigned char* const ring = wbase;
    unsigned char* const logb = wbase + RG_RING;
    unsigned char* const metab = logb + RG_LOG;
    unsigned char* const bnc = metab + RG_META;
    float* const s_pend = reinterpret_cast<float*>(bnc + RG_BNC);
    float* const s_m = reinterpret_cast<float*>(smem + RG_FRONT + wib * 1024);      // [16][8] references, then [16][8] exp-sums
    float* const s_l = s_m + 128;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_byte*)wbase);

    const int h = lane >> 3;
    const int fi = lane & 15, fg = lane >> 4;
    f32x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wrow0 < NQ) {
        const int tile = (int)(wrow0 / HGT_TD), within = (int)(wrow0 % HGT_TD);
        // ranges of the R + 1 relation buckets (lane r = bucket r) and their exclusive prefix = position in the virtual stream
        int my_beg, my_len;
        {
            const int64_t bb = ((int64_t)tile * (R + 1) + min(lane, R)) * HGT_TD + within;
            my_beg = segptr[bb];
            my_len = (lane <= R) ? segptr[bb + 16] - my_beg : 0;
        }
        int incl = my_len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int my_pre = incl - my_len;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        s_m[lane] = HGT_NEG; s_m[64 + lane] = HGT_NEG;
        s_l[lane] = 0.0f; s_l[64 + lane] = 0.0f;
        s_pend[lane] = 1.0f; s_pend[64 + lane] = 1.0f;

        if (total > 0) {
            // The issue pointer runs RG_D entries ahead of the fetch pointer and never stops: beyond the end of the stream it re-requests
            // the last edge's row (entries are clamped: cache hits), so "one row issued per row fetched" holds from the first row to the
            // last and every wait for a row is the SAME immediate.
            const int total_iss = total + RG_D;
            // stream entries [vbase, vbase + 64) -> meta / logits buffers `par` (4 hidden operations)
            auto chunk_issue = [&](int vbase, int par) {
                const int v = min(vbase + lane, total - 1);
                int rsel = 0;
                for (int r = 0; r <= R; ++r) {
                    const int pr = __builtin_amdgcn_readlane(my_pre, r), ln = __builtin_amdgcn_readlane(my_len, r);
                    if (ln > 0 && v >= pr) rsel = r;
                }
                const int pos = __shfl(my_beg, rsel) + (v - __shfl(my_pre, rsel));
                const unsigned mb = lds_w + RG_RING + RG_LOG + par * 512, lb = lds_w + RG_RING + par * 2048;
                rg_dma4_v(esrc + pos, mb);
                rg_dma4_v(edst + pos, mb + 256);
                rg_dma16_v(logits + (int64_t)pos * 8, lb);
                rg_dma16_v(logits + (int64_t)pos * 8 + 4, lb + 1024);
            };
#pragma nounroll
            for (int c = 0; c < 2; ++c)
                if (c * 64 < total_iss) chunk_issue(c * 64, c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ids of the chunk the ISSUE pointer is in (sources) / the FETCH pointer is in (targets, relative to the sub-tile)
            int v_src = *reinterpret_cast<const int*>(metab + lane * 4);
            int v_dst = *reinterpret_cast<const int*>(metab + 256 + lane * 4) - (int)wrow0;

            const unsigned voff = (unsigned)lane * 16u;
            // B operand of the transforms: rows of U_r for target (lane & 15), k block (lane >> 4), head hh: hi / mid
            bf16x8 ubh[8], ubm[8];
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) {
                ubh[hh] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                ubm[hh] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
            // Fragments.  A relation's 32 fragments are 128 VGPRs per lane -- next to 64 accumulators and the 64 registers of the U tile
            // that is the whole file -- so the 64 registers of fq hold ONE HALF at a time: column tiles 0-7 (heads 0-3) are requested when
            // the walk enters a claimed relation and are there at its end; tiles 8-15 are requested after the first half's products:
            // the one exposed round trip of a relation end (it drains the ring: every row issued before it has landed afterwards, `skip`).
            // Written by hidden loads and never read before the first of them (rowmask != 0 implies a claimed relation): no initial value.
            bf16x8 fq[NCT];      // [2 t] = hi, [2 t + 1] = mid of column tile t (first half) / 8 + t (second half)

            // ---- scalar state
            int rows_left = total;                 // rows not yet processed
            int con_left = 0, con_rel = -1;        // rows left in the relation being processed / that relation
            int f_idx = 0, f_par = 0, f_chunk = 0; // fetch pointer: entry inside its chunk, the chunk's buffers, the chunk
            int f_slot = 0;                        // byte offset of the fetch pointer's ring slot
            int i_idx = 0, i_par = 0;              // issue pointer: entry inside its chunk, the chunk's buffers
            int i_slot = 0;
            int skip = 0;                          // rows known to have landed (no wait)
            bool started = false, pend_any = false, chunk_due = false;
            int cur_dl = -1;
            unsigned rowmask = 0;
            float U0 = 0.f, U1 = 0.f, U2 = 0.f, U3 = 0.f, m_ref = 0.0f, l_seg = 0.0f, l_old = 0.0f;
            // per-lane constants of the LDS addresses
            const unsigned char* const p_ring = ring + lane * 16;
            const unsigned char* const p_log = logb + (h >> 2) * 1024 + (h & 3) * 4;
            const float* const p_state = s_m + h;

            // one row issued: entry at the issue pointer -> slot i_slot; the pointer advances
            auto issue_row = [&]() {
                const int src = (RG_OFF & 128) ? 0 : __builtin_amdgcn_readlane(v_src, i_idx);      // (128: every gather reads row 0)
                if (RG_OFF & 32) asm volatile("" ::"v"(((unsigned)src << 10) + voff), "s"(lds_w + (unsigned)i_slot));
                else if constexpr (WIDE) rg_dma16_s(V + (int64_t)src * 256, voff, lds_w + (unsigned)i_slot);
                else rg_dma16_s(V, ((unsigned)src << 10) + voff, lds_w + (unsigned)i_slot);
                if (++i_idx == 64) {      // the issue pointer enters the next chunk: its ids landed long ago (>= 64 - RG_D rows consumed since)
                    i_idx = 0;
                    i_par ^= 1;
                    v_src = *reinterpret_cast<const int*>(metab + i_par * 512 + lane * 4);
                    // (waited for HERE, once per 64 rows: left pending, the loop head's merged scoreboard puts an lgkmcnt(0) -- a full
                    //  drain of the LDS queue, parked-row reads included -- in front of every v_readlane of it)
                    asm volatile("" : "+v"(v_src));
                }
            };
            // row at the fetch pointer -> registers (and its logit, its target); the pointer advances.  WAIT = the vmcnt immediate: RG_D - 1
            // when this iteration's row has been issued already, RG_D - 2 before it
#define RG_FETCH(row, sl, dl, WAIT)                                                                                                \
    {                                                                                                                              \
        if (skip > 0) --skip;                                                                                                      \
        else if (!(RG_OFF & 16)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");                                          \
        row = *reinterpret_cast<const float4*>(p_ring + f_slot);                                                                   \
        sl = *reinterpret_cast<const float*>(p_log + f_par * 2048 + f_idx * 16);                                                   \
        dl = __builtin_amdgcn_readlane(v_dst, f_idx);                                                                              \
        n_slot = f_slot;                                                                                                           \
        f_slot = (f_slot + 1024 == RG_RING) ? 0 : f_slot + 1024;                                                                   \
        if (++f_idx == 64) {      /* the fetch pointer enters the next chunk: the previous chunk's buffers are free (chunk_due) */ \
            f_idx = 0;                                                                                                             \
            f_par ^= 1;                                                                                                            \
            ++f_chunk;                                                                                                             \
            v_dst = *reinterpret_cast<const int*>(metab + f_par * 512 + 256 + lane * 4) - (int)wrow0;                              \
            asm volatile("" : "+v"(v_dst));                                                                                        \
            chunk_due = true;                                                                                                      \
        }                                                                                                                          \
    }
            int n_slot = 0;

            // Parked rows wait in the park buffer (RG_PARK slots of hi | mid) and reach the U registers RG_PARK at a time: the exec-masked
            // fill is 16 ds_read_b128 whatever the number of lanes -- per row it cost 0.3 ms of the launch (profiles/r05_*), now a quarter.
            //   pk = rows waiting, pmask = the lanes that hold their targets' columns, slotv = byte offset of each such lane's slot
            int pk = 0;
            unsigned long long pmask = 0;
            int slotv = 0;
            auto fill = [&]() {
                if (pk > 0) {
                    if ((pmask >> lane) & 1ull) {
                        const unsigned char* b = bnc + fg * 16 + slotv;
#pragma unroll
                        for (int hh = 0; hh < 8; ++hh) {
                            ubh[hh] = *reinterpret_cast<const bf16x8*>(b + 64 * hh);
                            ubm[hh] = *reinterpret_cast<const bf16x8*>(b + 512 + 64 * hh);
                        }
                    }
                    pk = 0;
                    pmask = 0;
                }
            };
            // park the running segment: its exp-sum joins the target's state, its row goes into the U tile.  Afterwards (m_ref, l_old) are
            // the state of target cur_dl as it stands in LDS.
            auto flush = [&]() {
                if (cur_dl >= 0) {
                    const int dl = cur_dl;
                    l_old = l_old + l_seg;
                    l_seg = 0.0f;
                    s_m[dl * 8 + h] = m_ref;      // (every lane of a head writes the same pair: no exec mask)
                    s_l[dl * 8 + h] = l_old;
                    if (con_rel < R && !(RG_OFF & 4)) {
                        uint2 hi, mid;
                        split4(make_float4(U0, U1, U2, U3), hi, mid);
                        unsigned char* w = bnc + pk * 1024 + lane * 8;
                        *reinterpret_cast<uint2*>(w) = hi;
                        *reinterpret_cast<uint2*>(w + 512) = mid;
                        const bool mine = (fi == dl);
                        slotv = mine ? pk * 1024 : slotv;
                        pmask |= __builtin_amdgcn_ballot_w64(mine);
                        rowmask |= 1u << dl;
                        if (++pk == RG_PARK && !(RG_OFF & 2)) fill();
                    }
                }
            };

#define RG_TIE8(O) asm volatile("" : "+v"(fq[O]), "+v"(fq[O + 1]), "+v"(fq[O + 2]), "+v"(fq[O + 3]), "+v"(fq[O + 4]), "+v"(fq[O + 5]), "+v"(fq[O + 6]), "+v"(fq[O + 7]));
#define RG_FRAG_HALF(HALF)                                                                                                         \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                                                \
        const unsigned short* t_ = fb_cur + ((HALF) * 8 + t) * 1024;                                                               \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fq[2 * t]) : "v"(voff), "s"(t_) : "memory");                           \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(fq[2 * t + 1]) : "v"(voff), "s"(t_) : "memory");           \
    }
#define RG_MUL_HALF(HALF)                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                                                \
        const int c = (HALF) * 8 + t;                                                                                              \
        acc[c] = mfma16_t<false>(fq[2 * t + 1], ubh[c >> 1], acc[c] * pf[c >> 1]);                                                 \
        acc[c] = mfma16_t<false>(fq[2 * t], ubm[c >> 1], acc[c]);                                                                  \
        acc[c] = mfma16_t<false>(fq[2 * t], ubh[c >> 1], acc[c]);                                                                  \
    }
            // Relation boundary (also the start and the end of the stream): park the running segment, transform the finished relation,
            // enter the next one.  ONE code site: the fragment registers have a single definition.
#define RG_BOUNDARY()                                                                                                              \
    {                                                                                                                              \
        flush();                                                                                                                   \
        if (!(RG_OFF & 2)) fill();                                                                                                  \
        cur_dl = -1;                                                                                                               \
        bool drained = false;                                                                                                      \
        if (rowmask != 0 && !(RG_OFF & 8)) {             /* Z^T += M_r^T . U_r^T for the rows parked during relation con_rel */                  \
            const unsigned short* fb_cur = msgF + (int64_t)con_rel * (NCT * 2 * 512);                                              \
            if (rel_len < RG_D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* (a relation shorter than the ring) */        \
            float pf[8];      /* pending scales of this lane's target, head by head (1.0 unless a softmax reference moved) */      \
            {                                                                                                                      \
                const float4 p0 = *reinterpret_cast<const float4*>(s_pend + fi * 8), p1 = *reinterpret_cast<const float4*>(s_pend + fi * 8 + 4); \
                pf[0] = p0.x; pf[1] = p0.y; pf[2] = p0.z; pf[3] = p0.w; pf[4] = p1.x; pf[5] = p1.y; pf[6] = p1.z; pf[7] = p1.w;    \
            }                                                                                                                      \
            RG_TIE8(0) RG_TIE8(8)                                                                                                  \
            RG_MUL_HALF(0)                                                                                                         \
            RG_FRAG_HALF(1)                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* (the youngest operations: everything issued has landed) */    \
            drained = true;                                                                                                        \
            RG_TIE8(0) RG_TIE8(8)                                                                                                  \
            RG_MUL_HALF(1)                                                                                                         \
            _Pragma("unroll") for (int hh = 0; hh < 8; ++hh) {                                                                     \
                ubh[hh] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};                                                                        \
                ubm[hh] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};                                                                        \
            }                                                                                                                      \
            if (pend_any) {      /* (rare) the factors are in the accumulators now */                                              \
                pend_any = false;                                                                                                  \
                s_pend[lane] = 1.0f;                                                                                               \
                s_pend[64 + lane] = 1.0f;                                                                                          \
            }                                                                                                                      \
        }                                                                                                                          \
        rowmask = 0;                                                                                                               \
        if (rows_left == 0) break;                                                                                                 \
        do { ++con_rel; } while (con_rel < 63 && __builtin_amdgcn_readlane(my_len, con_rel) == 0);                                 \
        con_left = __builtin_amdgcn_readlane(my_len, con_rel);                                                                     \
        rel_len = con_left;                                                                                                        \
        if (con_rel < R) {      /* the first half of this relation's fragments: consumed at its end */                            \
            if (started && !drained) {      /* (rare: a relation that parked nothing) rows in flight are older than these loads */ \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
                drained = true;                                                                                                    \
            }                                                                                                                      \
            const unsigned short* fb_cur = msgF + (int64_t)con_rel * (NCT * 2 * 512);                                              \
            RG_FRAG_HALF(0)                                                                                                        \
        }                                                                                                                          \
        if (drained) skip = RG_D - 1;                                                                                              \
        if (!started) {      /* the first rows of the stream (issued behind the first relation's fragments) */                    \
            started = true;                                                                                                        \
            _Pragma("nounroll") for (int k = 0; k < RG_D; ++k) { i_slot = k * 1024; issue_row(); }                                 \
            RG_FETCH(rowC, slC, dlC, RG_D - 1)                                                                                     \
            i_slot = n_slot;                                                                                                       \
            mtC = HGT_NEG;      /* (nothing has been parked yet: every target's state is the initial one) */                      \
            loC = 0.0f;                                                                                                            \
        }                                                                                                                          \
    }
            // The fragments of half 0 are older than every row issued after them, and a row issued after them has been waited for before
            // the relation ends unless the relation is shorter than the ring: then the boundary waits for everything (rel_len).
            int rel_len = 0;
            // One iteration per row: row r sits in the C registers (fetched one iteration ahead); the row behind it is fetched at the TOP of
            // the iteration, so that its LDS round trip runs behind the parking of the finished segment, the next row's issue and row r's
            // arithmetic.  Every helper has ONE expansion site in the loop (two copies of the row step, with the register sets swapped
            // instead of copied, cost 236 B of scratch per lane: the accumulators).
            int prev_dl = -1;
            float4 rowC = make_float4(0.f, 0.f, 0.f, 0.f);
            float slC = 0.0f, mtC = 0.0f, loC = 0.0f;
            int dlC = 0;
            for (;;) {
                if (con_left == 0) RG_BOUNDARY()
                if (chunk_due) {      // ids + logits of the chunk after the fetch pointer's, into the buffers it just left
                    chunk_due = false;
                    if ((f_chunk + 1) * 64 < total_iss) chunk_issue((f_chunk + 1) * 64, f_par ^ 1);
                }
                float4 rowN;
                float slN;
                int dlN;
                RG_FETCH(rowN, slN, dlN, RG_D - 2)
                // a new segment begins with row r: park the finished one BEFORE the next row's state is read (the row after r may belong
                // to the target whose segment is parked here -- in the next relation)
                if (dlC != cur_dl) {
                    const bool same_target = (dlC == prev_dl);
                    flush();
                    U0 = U1 = U2 = U3 = 0.0f;
                    cur_dl = dlC;
                    if (!same_target) {      // (same target in the next relation: its state is in m_ref / l_old already)
                        m_ref = (mtC == HGT_NEG) ? slC : mtC;
                        l_old = loC;
                    }
                    prev_dl = dlC;
                }
                const float mtN = p_state[dlN * 8], loN = p_state[dlN * 8 + 128];
                issue_row();      // (the slot of row r is free: its data sits in registers)
                i_slot = n_slot;
                {
                    float dlt = slC - m_ref;
                    if (__builtin_amdgcn_ballot_w64(dlt > 40.0f) != 0) {
                        // Rare: move the reference of the heads that were exceeded; everything this target has accumulated so far is
                        // rescaled (see k_edge_aggregate_update_mfma) ...
                        const float m_new = (dlt > 40.0f) ? slC : m_ref;
                        const float sc = __expf(m_ref - m_new);
                        U0 *= sc; U1 *= sc; U2 *= sc; U3 *= sc;
                        l_seg *= sc;
                        l_old *= sc;
                        // ... and its COLUMN of the accumulators -- not here: a second place that writes the 64 accumulator registers costs
                        // the walk ~60 registers (the allocator keeps both versions of the tuples: 188-236 B of scratch in every form
                        // tried).  The factor is left in the target's row of s_pend and multiplied in where the column is touched next:
                        // the relation's products (RG_MUL_HALF) or the final normalisation.  (x * 1.0f is exact: no move, no change.)
                        if ((lane & 7) == 0) s_pend[cur_dl * 8 + h] *= sc;
                        pend_any = true;
                        m_ref = m_new;
                        dlt = slC - m_ref;
                    }
                    const float pe = (RG_OFF & 64) ? dlt : __expf(dlt);
                    // (rows of the unclaimed bucket are gathered too and their sums never parked: no condition here)
                    if (!(RG_OFF & 64)) {
                        U0 = fmaf(pe, rowC.x, U0);
                        U1 = fmaf(pe, rowC.y, U1);
                        U2 = fmaf(pe, rowC.z, U2);
                        U3 = fmaf(pe, rowC.w, U3);
                    } else {
                        U0 += rowC.x + rowC.y + rowC.z + rowC.w;
                    }
                    l_seg += pe;
                    --con_left;
                    --rows_left;
                }
                rowC = rowN; slC = slN; dlC = dlN; mtC = mtN; loC = loN;
            }
#undef RG_FETCH
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the re-requested rows behind the end of the stream)
#undef RG_BOUNDARY
#undef RG_TIE8
#undef RG_FRAG_HALF
#undef RG_MUL_HALF
        }
        // normalise (PyG softmax denominator, conv.py:108) + exact-erf gelu (conv.py:119), in the accumulator layout
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const float inv = 1.0f * (1.0f / (s_l[fi * 8 + (16 * c + 4 * fg) / DKP] + 1e-16f));
            const float pfin = s_pend[fi * 8 + (c >> 1)];      // (a reference moved after the column's last products: still pending)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o = (acc[c][r] * pfin) * inv;
                o = 0.5f * o * (1.0f + erff(o * 0.70710678118654752440f));
                acc[c][r] = o;
            }
        }
    }
    // row types for the epilogue, requested before the barrier: the round trip overlaps the wait for the slowest wavefront
    int type_pre = -1;
    if (wib == 0 && row0 + lane < NQ) {
        const int64_t t = fu.node_type[row0 + lane];
        type_pre = (t >= 0 && t < fu.n_types) ? (int)t : -1;
    }
    __syncthreads();   // every wavefront is done with its ring and its softmax state: the A slab overlays them
    {
        // accumulator layout -> A slab: target (lane & 15) of this wavefront, columns 16 c + 4 (lane >> 4) .. + 3
        unsigned char* prow = smem + (wib * 16 + fi) * A_STRIDE + fg * 8;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            uint2 hi, mid;
            split4_t<false>(make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]), 1.0f, hi, mid);
            *reinterpret_cast<uint2*>(prow + c * 32) = hi;
            *reinterpret_cast<uint2*>(prow + A_PLANE + c * 32) = mid;
        }
    }
#if !(RG_OFF & 1)      // (no node update: the walk's own register need)
    fused_update_tail<4, HGT_FU_NSTG, true, false>(smem, smem + RG_FRONT, row0, NQ, fu, type_pre);
#endif
}
