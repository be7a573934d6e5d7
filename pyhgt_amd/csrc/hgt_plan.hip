// Graph plan build: (edge_index, edge_type, edge_time, node_type) -> sorted, compacted edge
// arrays + segment table + wavefront work items + typed row lists.  One-off per sampled
// subgraph (shared by every layer); see include/hgt_hip.h for what it replaces in the reference
// (PyG propagate gathers called from conv.py:57, the mask cube conv.py:71-84, update masks
// conv.py:121-123).
//
// The stable sorts use rocPRIM's device radix sort (ROCm-native primitive, header-only); the
// key/fill/segment/item kernels are hand-written.  Everything is enqueued on the caller's stream,
// nothing synchronises; the number of work items stays on the device (HgtPlanHeader::n_items)
// and consumers launch the host-side upper bound.
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "hgt_common.h"

namespace {

struct TmpLayout {
    uint64_t off_keys_in, off_keys_out, off_vals_in, off_vals_out, off_pair_cnt, off_pair_off;
    uint64_t off_nkeys_in, off_nkeys_out, off_nvals_in, off_sort_tmp, sort_tmp_bytes, total;
};

static TmpLayout tmp_layout(int64_t N, int64_t E, const HgtPlanLayout& L, uint64_t sort_tmp_bytes) {
    TmpLayout t;
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { uint64_t r = o; o = hgt_align_up(o + bytes, 256); return r; };
    t.off_keys_in = take((uint64_t)E * 4);
    t.off_keys_out = take((uint64_t)E * 4);
    t.off_vals_in = take((uint64_t)E * 4);
    t.off_vals_out = take((uint64_t)E * 4);
    t.off_pair_cnt = take((uint64_t)(L.n_pairs + 1) * 4);
    t.off_pair_off = take((uint64_t)(L.n_pairs + 1) * 4);
    t.off_nkeys_in = take((uint64_t)N * 4);
    t.off_nkeys_out = take((uint64_t)N * 4);
    t.off_nvals_in = take((uint64_t)N * 4);
    t.sort_tmp_bytes = sort_tmp_bytes;
    t.off_sort_tmp = take(sort_tmp_bytes);
    t.total = o;
    return t;
}

static int key_bits(uint64_t n_values) {
    int b = 1;
    while (b < 32 && (1ull << b) < n_values) ++b;
    return b;
}

// rocPRIM temp-storage requirement (host-side query, no launch).
static int sort_tmp_query(int64_t N, int64_t E, int32_t T, const HgtPlanLayout& L, uint64_t* bytes) {
    size_t a = 0, b = 0, c = 0;
    uint32_t* kp = nullptr;
    int32_t* vp = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, a, kp, kp, vp, vp, (size_t)(E > 0 ? E : 1), 0, key_bits((uint64_t)L.n_bins),
                                  (hipStream_t)0) != hipSuccess) return HGT_ERR_LAUNCH;
    if (rocprim::radix_sort_pairs(nullptr, b, kp, kp, vp, vp, (size_t)(N > 0 ? N : 1), 0, key_bits((uint64_t)T + 1),
                                  (hipStream_t)0) != hipSuccess) return HGT_ERR_LAUNCH;
    if (rocprim::exclusive_scan(nullptr, c, vp, vp, 0, (size_t)(L.n_pairs + 1), rocprim::plus<int32_t>(),
                                (hipStream_t)0) != hipSuccess) return HGT_ERR_LAUNCH;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    *bytes = (uint64_t)m + 256;
    return HGT_OK;
}

__global__ void k_init_header(HgtPlanHeader* hdr) {
    if (threadIdx.x == 0) { hdr->n_items = 0; hdr->bad_index = 0; hdr->n_hubs = 0; }
}

// key = ((dst / TD) * (R+1) + rel') * TD + dst % TD ; rel' = R for edges no meta relation claims
__global__ void k_edge_keys(const int64_t* __restrict__ ei, int64_t sr, int64_t sc, const int64_t* __restrict__ etype,
                            const int64_t* __restrict__ ntype, int64_t N, int64_t NQ, int64_t E, int T, int R,
                            uint32_t* __restrict__ keys, int32_t* __restrict__ vals, HgtPlanHeader* hdr) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t src = ei[e * sc];
    int64_t dst = ei[sr + e * sc];
    bool bad = false;
    if (src < 0 || src >= N) { src = 0; bad = true; }
    if (dst < 0 || dst >= NQ) { dst = 0; bad = true; }
    if (bad) atomicOr(&hdr->bad_index, 1);
    int64_t ts = ntype[src], td = ntype[dst], r = etype[e];
    bool claimed = !bad && ts >= 0 && ts < T && td >= 0 && td < T && r >= 0 && r < R;
    uint32_t rr = claimed ? (uint32_t)r : (uint32_t)R;
    uint32_t tile = (uint32_t)(dst / HGT_TD), dl = (uint32_t)(dst % HGT_TD);
    keys[e] = (tile * (uint32_t)(R + 1) + rr) * HGT_TD + dl;
    vals[e] = (int32_t)e;
}

__global__ void k_edge_fill(const int64_t* __restrict__ ei, int64_t sr, int64_t sc, const int64_t* __restrict__ etime,
                            const int64_t* __restrict__ ntype, int64_t N, int64_t NQ, int64_t E, int T,
                            const int32_t* __restrict__ order, int32_t* __restrict__ esrc, int32_t* __restrict__ edst,
                            uint16_t* __restrict__ ertei, HgtPlanHeader* hdr) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    int32_t e = order[p];
    int64_t src = ei[(int64_t)e * sc];
    int64_t dst = ei[sr + (int64_t)e * sc];
    if (src < 0 || src >= N) src = 0;
    if (dst < 0 || dst >= NQ) dst = 0;
    int64_t ts = ntype[src];
    if (ts < 0 || ts >= T) ts = 0;
    int64_t tm = etime ? etime[e] : 0;
    if (tm < 0 || tm >= HGT_RTE_LEN) {   // the reference's nn.Embedding lookup (conv.py:299) raises; here: flagged + clamped
        atomicOr(&hdr->bad_index, 2);
        tm = tm < 0 ? 0 : HGT_RTE_LEN - 1;
    }
    esrc[p] = (int32_t)src;
    edst[p] = (int32_t)dst;
    ertei[p] = (uint16_t)(ts * HGT_RTE_LEN + tm);
}

// segptr[b] = first sorted position whose key >= b (lower bound); segptr[n_bins] = E
__global__ void k_segptr(const uint32_t* __restrict__ keys_sorted, int64_t E, int64_t n_bins, int32_t* __restrict__ segptr) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_bins) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys_sorted[mid] < b) lo = mid + 1; else hi = mid;
    }
    segptr[b] = (int32_t)lo;
}

__global__ void k_pair_counts(const int32_t* __restrict__ segptr, int64_t n_pairs, int ch, int32_t* __restrict__ cnt) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_pairs) return;
    if (j == n_pairs) { cnt[j] = 0; return; }
    int32_t len = segptr[(j + 1) * HGT_TD] - segptr[j * HGT_TD];
    cnt[j] = (len + ch - 1) / ch;
}

__global__ void k_items(const int32_t* __restrict__ segptr, const int32_t* __restrict__ pair_off, int64_t n_pairs, int R, int ch,
                        HgtItem* __restrict__ items, int32_t* __restrict__ tile_items, HgtPlanHeader* hdr) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_pairs) return;
    if (j % (R + 1) == 0) tile_items[j / (R + 1)] = pair_off[j];   // includes the end sentinel at j == n_pairs
    if (j == n_pairs) { hdr->n_items = pair_off[n_pairs]; return; }
    int32_t beg = segptr[j * HGT_TD], end = segptr[(j + 1) * HGT_TD];
    int32_t o = pair_off[j];
    int32_t rel = (int32_t)(j % (R + 1)), tile = (int32_t)(j / (R + 1));
    for (int32_t b = beg; b < end; b += ch) {
        HgtItem it;
        it.beg = b;
        it.end = (b + ch < end) ? b + ch : end;
        it.rel = rel;
        it.tile = tile;
        items[o++] = it;
    }
}

// hub targets: in-degree (over all relation buckets) above HGT_HUB_DEG.  They get a slot in the hub buffers and are
// aggregated by many wavefronts (one per work item) instead of the single wavefront that owns their sub-tile.
__global__ void k_hub_detect(const int32_t* __restrict__ segptr, int64_t N, int R, int32_t* __restrict__ hub_slot,
                             int32_t* __restrict__ hub_list, HgtPlanHeader* hdr) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int64_t tile = n / HGT_TD, dl = n % HGT_TD;
    int deg = 0;
    for (int r = 0; r <= R; ++r) {
        const int64_t b = (tile * (R + 1) + r) * HGT_TD + dl;
        deg += segptr[b + 1] - segptr[b];
    }
    int slot = -1;
    if (deg > HGT_HUB_DEG) {
        slot = atomicAdd(&hdr->n_hubs, 1);
        hub_list[slot] = (int32_t)n;
    }
    hub_slot[n] = slot;
}

__global__ void k_node_keys(const int64_t* __restrict__ ntype, int64_t N, int T, uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int64_t t = ntype[n];
    keys[n] = (t >= 0 && t < T) ? (uint32_t)t : (uint32_t)T;
    vals[n] = (int32_t)n;
}

// off[g] = lower bound of g in the sorted type keys, g = 0..T+1
// (hdr != nullptr: also record how many of the N rows have no valid type -- the layer then knows whether hgt_zero_rows has work)
__global__ void k_type_offsets(const uint32_t* __restrict__ keys_sorted, int64_t N, int T, int32_t* __restrict__ off, HgtPlanHeader* hdr) {
    int g = threadIdx.x;
    if (g <= T + 1) {
        int64_t lo = 0, hi = N;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys_sorted[mid] < g) lo = mid + 1; else hi = mid;
        }
        off[g] = (int32_t)lo;
    }
    __syncthreads();
    if (hdr != nullptr && g == 0) hdr->n_unknown_q = off[T + 1] - off[T];
}


// ---------------------------------------------------------------------------------------------
// Plan from a PRE-SORTED graph (hgt_plan_from_sorted; SURVEY.md section 8f-3: the reference's sampler already emits
// type-contiguous nodes and, per relation, target-sorted edge runs -- data.py:183-209,227-246 -- so the radix sorts of
// hgt_plan_build are redundant for its batches).  Input: edges grouped by relation (rel_ptr[R+1]), target ids
// non-decreasing inside a relation.  The plan order (tile, relation, target) is then a MERGE of R sorted lists by tile:
//   lb[tile][r]  = first edge of relation r with target >= tile * TD              (binary search)
//   base[tile][r] = exclusive scan of (lb[tile+1][r] - lb[tile][r]) over (tile, r)  -> sorted position of that run
//   edge i of relation r in tile t lands at base[t][r] + (i - lb[t][r]);  segptr by one more binary search per bin.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ a, int lo, int hi, int key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// pair j = tile * (R+1) + r (r == R: the unclaimed bucket, always empty here).  cnt[j] = edges of the pair; icnt[j] = its items.
__global__ void k_sorted_pair_counts(const int32_t* __restrict__ dst, const int32_t* __restrict__ rel_ptr, int64_t n_pairs, int R, int ch,
                                     int32_t* __restrict__ lb, int32_t* __restrict__ cnt, int32_t* __restrict__ icnt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_pairs) return;
    if (j == n_pairs) { cnt[j] = 0; icnt[j] = 0; return; }
    const int r = (int)(j % (R + 1));
    const int64_t tile = j / (R + 1);
    int c = 0, l0 = 0;
    if (r < R) {
        const int b = rel_ptr[r], e = rel_ptr[r + 1];
        l0 = lower_bound_i32(dst, b, e, (int)(tile * HGT_TD));
        const int l1 = lower_bound_i32(dst, l0, e, (int)((tile + 1) * HGT_TD));
        c = l1 - l0;
    }
    lb[j] = l0;
    cnt[j] = c;
    icnt[j] = (c + ch - 1) / ch;
}

__global__ void k_sorted_segptr(const int32_t* __restrict__ dst, const int32_t* __restrict__ lb, const int32_t* __restrict__ cnt,
                                const int32_t* __restrict__ base, int64_t n_bins, int R, int64_t E, int32_t* __restrict__ segptr) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_bins) return;
    if (b == n_bins) { segptr[b] = (int32_t)E; return; }
    const int64_t j = b / HGT_TD;
    const int dl = (int)(b % HGT_TD);
    const int64_t tile = j / (R + 1);
    const int l0 = lb[j], c = cnt[j];
    const int pos = lower_bound_i32(dst, l0, l0 + c, (int)(tile * HGT_TD + dl));
    segptr[b] = base[j] + (pos - l0);
}

__global__ void k_sorted_scatter(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const int32_t* __restrict__ etime,
                                 const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ type_off, const int32_t* __restrict__ lb,
                                 const int32_t* __restrict__ base, int64_t E, int T, int R, int64_t N, int64_t NQ,
                                 int32_t* __restrict__ esrc, int32_t* __restrict__ edst, uint16_t* __restrict__ ertei,
                                 int32_t* __restrict__ eid, HgtPlanHeader* hdr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    int r = 0;
    while (r + 1 < R && rel_ptr[r + 1] <= i) ++r;       // R is small (<= 64 relations in the reference's schemas)
    int s = src[i], d = dst[i];
    // the caller's ordering is part of the contract: targets non-decreasing inside a relation, rel_ptr spanning [0, E] --
    // otherwise positions collide inside [0, E) and the plan would be silently corrupt (bit 2 of bad_index)
    if ((i > rel_ptr[r] && dst[i - 1] > d) || (i == 0 && (rel_ptr[0] != 0 || rel_ptr[R] != E))) atomicOr(&hdr->bad_index, 4);
    if (s < 0 || s >= N || d < 0 || d >= NQ) { atomicOr(&hdr->bad_index, 1); s = max(0, min(s, (int)N - 1)); d = max(0, min(d, (int)NQ - 1)); }
    const int64_t j = (int64_t)(d / HGT_TD) * (R + 1) + r;
    int p = base[j] + ((int)i - lb[j]);
    if (p < 0 || p >= E) { atomicOr(&hdr->bad_index, 1); p = (int)i; }   // (only with malformed input: unsorted targets / ids out of range)
    int ts = 0;
    while (ts + 1 < T && type_off[ts + 1] <= s) ++ts;
    int tm = etime ? etime[i] : 0;
    if (tm < 0 || tm >= HGT_RTE_LEN) { atomicOr(&hdr->bad_index, 2); tm = tm < 0 ? 0 : HGT_RTE_LEN - 1; }
    esrc[p] = s;
    edst[p] = d;
    ertei[p] = (uint16_t)(ts * HGT_RTE_LEN + tm);
    eid[p] = (int32_t)i;
}

// typed row lists of type-contiguous nodes: rows = identity, offsets given
__global__ void k_sorted_rows(const int32_t* __restrict__ type_off, int64_t N, int64_t NQ, int T, int32_t* __restrict__ rows_all,
                              int32_t* __restrict__ off_all, int32_t* __restrict__ rows_q, int32_t* __restrict__ off_q, HgtPlanHeader* hdr) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) {
        hdr->n_items = 0; hdr->bad_index = 0; hdr->n_hubs = 0;
        hdr->n_unknown_q = (int32_t)(min(N, NQ) - min((int64_t)type_off[T], NQ));
    }
    if (n < N) { rows_all[n] = (int32_t)n; rows_q[n] = (int32_t)n; }
    if (n <= T + 1) {
        const int32_t v = (n <= T) ? type_off[n] : (int32_t)N;          // group T (unknown types) is empty
        off_all[n] = v;
        off_q[n] = (int32_t)min((int64_t)v, NQ);
    }
}

// exclusive scans of the (small) per-pair counts in ONE workgroup: two arrays at once (edge positions, item positions)
__global__ __launch_bounds__(1024) void k_scan2_single(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int64_t n,
                                                        int32_t* __restrict__ oa, int32_t* __restrict__ ob) {
    __shared__ int sa[1024], sb[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t beg = tid * per, end = min(beg + per, n);
    int ta = 0, tb = 0;
    for (int64_t i = beg; i < end; ++i) { ta += a[i]; tb += b[i]; }
    sa[tid] = ta; sb[tid] = tb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int va = tid >= o ? sa[tid - o] : 0, vb = tid >= o ? sb[tid - o] : 0;
        __syncthreads();
        sa[tid] += va; sb[tid] += vb;
        __syncthreads();
    }
    int ra = sa[tid] - ta, rb = sb[tid] - tb;
    for (int64_t i = beg; i < end; ++i) { oa[i] = ra; ob[i] = rb; ra += a[i]; rb += b[i]; }
}

// ---- small graphs (the reference's sampled batches): the same plan in three launches instead of seven ----
// (a) ONE workgroup: typed row lists, per-pair lower bounds / counts, both exclusive scans (the pair table fits in LDS)
constexpr int SMALL_PAIRS = 4096;     // (n_pairs + 1) <= SMALL_PAIRS: 4 table entries per thread of a 1024-thread workgroup
__global__ __launch_bounds__(1024) void k_sorted_small_head(
    const int32_t* __restrict__ dst, const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ type_off, int64_t N, int64_t NQ,
    int T, int R, int n_pairs, int ch, int32_t* __restrict__ lb, int32_t* __restrict__ cnt, int32_t* __restrict__ base,
    int32_t* __restrict__ pair_off, int32_t* __restrict__ rows_all, int32_t* __restrict__ off_all, int32_t* __restrict__ rows_q,
    int32_t* __restrict__ off_q, HgtPlanHeader* hdr) {
    __shared__ int s_lb[SMALL_PAIRS + 1];
    __shared__ int s_a[1024], s_b[1024];
    const int tid = threadIdx.x;
    if (tid == 0) {
        hdr->n_items = 0; hdr->bad_index = 0; hdr->n_hubs = 0;
        hdr->n_unknown_q = (int32_t)(min(N, NQ) - min((int64_t)type_off[T], NQ));
    }
    for (int64_t n = tid; n < N; n += 1024) { rows_all[n] = (int32_t)n; rows_q[n] = (int32_t)n; }
    if (tid <= T + 1) {
        const int32_t v = (tid <= T) ? type_off[tid] : (int32_t)N;
        off_all[tid] = v;
        off_q[tid] = (int32_t)min((int64_t)v, NQ);
    }
    // lower bound of every (tile, relation) pair: ONE binary search per pair; its end is the next tile's lower bound
    for (int j = tid; j < n_pairs; j += 1024) {
        const int r = j % (R + 1), tile = j / (R + 1);
        int l0 = 0;
        if (r < R) l0 = lower_bound_i32(dst, rel_ptr[r], rel_ptr[r + 1], tile * HGT_TD);
        s_lb[j] = l0;
    }
    __syncthreads();
    // thread t owns the contiguous entries [4t, 4t + 4) of the pair table (entry n_pairs = the end sentinel, count 0)
    int c[4], ic[4], ta = 0, tb = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = 4 * tid + u;
        c[u] = 0;
        if (j < n_pairs) {
            const int r = j % (R + 1);
            if (r < R) {
                const int l1 = (j + R + 1 < n_pairs) ? s_lb[j + R + 1] : rel_ptr[r + 1];
                c[u] = l1 - s_lb[j];
            }
        }
        ic[u] = (c[u] + ch - 1) / ch;
        ta += c[u];
        tb += ic[u];
    }
    s_a[tid] = ta; s_b[tid] = tb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int va = tid >= o ? s_a[tid - o] : 0, vb = tid >= o ? s_b[tid - o] : 0;
        __syncthreads();
        s_a[tid] += va; s_b[tid] += vb;
        __syncthreads();
    }
    int ra = s_a[tid] - ta, rb = s_b[tid] - tb;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = 4 * tid + u;
        if (j <= n_pairs) {
            lb[j] = (j < n_pairs) ? s_lb[j] : 0;
            cnt[j] = c[u];
            base[j] = ra;
            pair_off[j] = rb;
        }
        ra += c[u];
        rb += ic[u];
    }
}

// (b) ONE wide launch: blocks [0, nb_e) scatter the edges, [nb_e, nb_e + nb_b) write segptr, the rest cut the work items
//     (an item range comes from base / cnt, not from segptr, so that the three roles only depend on launch (a))
__global__ void k_sorted_small_body(const int32_t* __restrict__ src, const int32_t* __restrict__ dst, const int32_t* __restrict__ etime,
                                    const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ type_off,
                                    const int32_t* __restrict__ lb, const int32_t* __restrict__ cnt, const int32_t* __restrict__ base,
                                    const int32_t* __restrict__ pair_off, int64_t E, int T, int R, int64_t N, int64_t NQ, int64_t n_bins,
                                    int n_pairs, int ch, unsigned nb_e, unsigned nb_b, int32_t* __restrict__ esrc,
                                    int32_t* __restrict__ edst, uint16_t* __restrict__ ertei, int32_t* __restrict__ eid,
                                    int32_t* __restrict__ segptr, HgtItem* __restrict__ items, int32_t* __restrict__ tile_items,
                                    HgtPlanHeader* hdr) {
    if (blockIdx.x < nb_e) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= E) return;
        int r = 0;
        while (r + 1 < R && rel_ptr[r + 1] <= i) ++r;
        int s = src[i], d = dst[i];
        if ((i > rel_ptr[r] && dst[i - 1] > d) || (i == 0 && (rel_ptr[0] != 0 || rel_ptr[R] != E))) atomicOr(&hdr->bad_index, 4);   // see k_sorted_scatter
        if (s < 0 || s >= N || d < 0 || d >= NQ) { atomicOr(&hdr->bad_index, 1); s = max(0, min(s, (int)N - 1)); d = max(0, min(d, (int)NQ - 1)); }
        const int64_t j = (int64_t)(d / HGT_TD) * (R + 1) + r;
        int p = base[j] + ((int)i - lb[j]);
        if (p < 0 || p >= E) { atomicOr(&hdr->bad_index, 1); p = (int)i; }   // (only with malformed input: unsorted targets / ids out of range)
        int ts = 0;
        while (ts + 1 < T && type_off[ts + 1] <= s) ++ts;
        int tm = etime ? etime[i] : 0;
        if (tm < 0 || tm >= HGT_RTE_LEN) { atomicOr(&hdr->bad_index, 2); tm = tm < 0 ? 0 : HGT_RTE_LEN - 1; }
        esrc[p] = s;
        edst[p] = d;
        ertei[p] = (uint16_t)(ts * HGT_RTE_LEN + tm);
        eid[p] = (int32_t)i;
    } else if (blockIdx.x < nb_e + nb_b) {
        const int64_t b = (int64_t)(blockIdx.x - nb_e) * blockDim.x + threadIdx.x;
        if (b > n_bins) return;
        if (b == n_bins) { segptr[b] = (int32_t)E; return; }
        const int64_t j = b / HGT_TD;
        const int dl = (int)(b % HGT_TD);
        const int64_t tile = j / (R + 1);
        const int l0 = lb[j], c = cnt[j];
        const int pos = (c == 0) ? l0 : lower_bound_i32(dst, l0, l0 + c, (int)(tile * HGT_TD + dl));
        segptr[b] = base[j] + (pos - l0);
    } else {
        const int64_t j = (int64_t)(blockIdx.x - nb_e - nb_b) * blockDim.x + threadIdx.x;
        if (j > n_pairs) return;
        if (j % (R + 1) == 0) tile_items[j / (R + 1)] = pair_off[j];
        if (j == n_pairs) { hdr->n_items = pair_off[n_pairs]; return; }
        const int32_t beg = base[j], end = beg + cnt[j];
        int32_t o = pair_off[j];
        const int32_t rel = (int32_t)(j % (R + 1)), tile = (int32_t)(j / (R + 1));
        for (int32_t b = beg; b < end; b += ch) {
            HgtItem it;
            it.beg = b;
            it.end = (b + ch < end) ? b + ch : end;
            it.rel = rel;
            it.tile = tile;
            items[o++] = it;
        }
    }
}

static inline unsigned nblk(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

extern "C" int hgt_plan_sizes_for(int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                                  hgt_plan_sizes* out) {
    if (!out || n_nodes < 0 || n_edges < 0 || n_types <= 0 || n_relations <= 0 || n_types > 250) return HGT_ERR_INVALID_ARG;
    HgtPlanLayout L = hgt_plan_layout(n_nodes, n_edges, n_types, n_relations);
    if (n_nodes >= (1ll << 31) - HGT_TD || n_edges >= (1ll << 31) - HGT_CH || (uint64_t)L.n_bins >= (1ull << 32) - 1 ||
        (int64_t)n_types * HGT_RTE_LEN > 65535)
        return HGT_ERR_TOO_LARGE;
    uint64_t st = 0;
    int rc = sort_tmp_query(n_nodes, n_edges, n_types, L, &st);
    if (rc != HGT_OK) return rc;
    TmpLayout t = tmp_layout(n_nodes, n_edges, L, st);
    out->plan_bytes = L.total;
    out->tmp_bytes = t.total;
    out->max_items = L.max_items;
    out->n_bins = L.n_bins;
    return HGT_OK;
}

extern "C" int hgt_plan_constants(int32_t* tile_nodes, int32_t* item_edges) {
    if (!tile_nodes || !item_edges) return HGT_ERR_INVALID_ARG;
    *tile_nodes = HGT_TD;
    *item_edges = HGT_CH;
    return HGT_OK;
}

extern "C" int hgt_plan_item_edges(int64_t n_edges, int32_t* item_edges) {
    if (!item_edges || n_edges < 0) return HGT_ERR_INVALID_ARG;
    *item_edges = hgt_item_edges(n_edges);
    return HGT_OK;
}

extern "C" int hgt_plan_row_lists(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types,
                                  int32_t n_relations, hgt_plan_rows* out) {
    if (!plan || !out) return HGT_ERR_INVALID_ARG;
    HgtPlanView v = hgt_plan_view(plan, n_nodes, n_edges, n_types, n_relations);
    out->rows_all = v.rows_all;
    out->off_all = v.off_all;
    out->rows_q = v.rows_q;
    out->off_q = v.off_q;
    return HGT_OK;
}

// ABI 6: where the per-tile item table lives inside a plan buffer: int32[n_tiles + 1] at byte offset *offset, the logits work
// items of destination tile t (tile_nodes targets, hgt_plan_constants) are [table[t], table[t + 1]).  The multi-GPU path reads it
// once per graph to launch the edge phase of one target block (hgt_conv_forward stage 5).
extern "C" int hgt_plan_tile_items_offset(int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, uint64_t* offset,
                                          int64_t* n_tiles) {
    if (!offset || !n_tiles || n_nodes < 0 || n_edges < 0) return HGT_ERR_INVALID_ARG;
    const HgtPlanLayout L = hgt_plan_layout(n_nodes, n_edges, n_types, n_relations);
    *offset = L.off_tile_items;
    *n_tiles = L.n_tiles;
    return HGT_OK;
}

// ---- small graphs through the wire format of the reference (hgt_plan_build on a sampled batch: every training batch is a NEW
// graph, data.py:212-256 -> model.py:69): the radix build was 25 launches of ~4.8 us each at c3 (r6 timeline, tools/lab/trace_plan.py).
// Two single-workgroup kernels take ten of them:
// (a) the typed row lists (all nodes / target nodes) as a STABLE counting sort by node type -- bucket by bucket, a block scan per
//     bucket: the same lists as the two stable radix sorts of (type, node id) pairs, for N <= 65 536 and T <= 30
constexpr int SMALL_ROWS_N = 65536, SMALL_ROWS_T = 30;
__global__ __launch_bounds__(1024) void k_node_rows_small(const int64_t* __restrict__ ntype, int64_t N, int64_t NQ, int T,
                                                           int32_t* __restrict__ rows_all, int32_t* __restrict__ off_all,
                                                           int32_t* __restrict__ rows_q, int32_t* __restrict__ off_q, HgtPlanHeader* hdr) {
    __shared__ int s_a[16], s_b[16];
    __shared__ unsigned char s_key[SMALL_ROWS_N];      // the nodes' buckets, read once (coalesced) from the int64 type array
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t n = tid; n < N; n += 1024) {
        const int64_t ty = ntype[n];
        s_key[n] = (unsigned char)((ty >= 0 && ty < T) ? (int)ty : T);
    }
    __syncthreads();
    const int64_t per = (N + 1023) / 1024;
    const int64_t beg = min((int64_t)tid * per, N), end = min(beg + per, N);
    int base_a = 0, base_q = 0;
    for (int t = 0; t <= T; ++t) {      // bucket T = nodes whose type no layer knows
        int ca = 0, cq = 0;
        for (int64_t n = beg; n < end; ++n) {
            const int key = s_key[n];
            if (key == t) { ++ca; if (n < NQ) ++cq; }
        }
        // block exclusive scan of (ca, cq): wavefront scans + the sixteen wavefront totals
        int ia = ca, iq = cq;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int va = __shfl_up(ia, o), vq = __shfl_up(iq, o);
            if (lane >= o) { ia += va; iq += vq; }
        }
        if (lane == 63) { s_a[wave] = ia; s_b[wave] = iq; }
        __syncthreads();
        int wa = 0, wq = 0, tot_a = 0, tot_q = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int va = s_a[w], vq = s_b[w];
            if (w < wave) { wa += va; wq += vq; }
            tot_a += va; tot_q += vq;
        }
        int pa = base_a + wa + ia - ca, pq = base_q + wq + iq - cq;
        for (int64_t n = beg; n < end; ++n) {
            const int key = s_key[n];
            if (key == t) { rows_all[pa++] = (int32_t)n; if (n < NQ) rows_q[pq++] = (int32_t)n; }
        }
        if (tid == 0) { off_all[t] = base_a; off_q[t] = base_q; }
        base_a += tot_a; base_q += tot_q;
        __syncthreads();
    }
    if (tid == 0) {
        off_all[T + 1] = base_a;
        off_q[T + 1] = base_q;
        hdr->n_unknown_q = off_q[T + 1] - off_q[T];
    }
}

// (b) per-pair item counts + their exclusive scan in one workgroup (k_pair_counts + rocprim::exclusive_scan: three launches)
__global__ __launch_bounds__(1024) void k_pair_scan_small(const int32_t* __restrict__ segptr, int64_t n_pairs, int ch,
                                                           int32_t* __restrict__ pair_off) {
    __shared__ int s_a[16];
    const int tid = threadIdx.x;
    const int64_t n = n_pairs + 1, per = (n + 1023) / 1024;
    const int64_t beg = min((int64_t)tid * per, n), end = min(beg + per, n);
    int tot = 0;
    for (int64_t j = beg; j < end; ++j) tot += (j < n_pairs) ? (segptr[(j + 1) * HGT_TD] - segptr[j * HGT_TD] + ch - 1) / ch : 0;
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if ((tid & 63) >= o) inc += v;
    }
    if ((tid & 63) == 63) s_a[tid >> 6] = inc;
    __syncthreads();
    int wsum = 0;
    for (int w = 0; w < (tid >> 6); ++w) wsum += s_a[w];
    int run = wsum + inc - tot;
    for (int64_t j = beg; j < end; ++j) {
        pair_off[j] = run;
        run += (j < n_pairs) ? (segptr[(j + 1) * HGT_TD] - segptr[j * HGT_TD] + ch - 1) / ch : 0;
    }
}

extern "C" int hgt_plan_build(const int64_t* edge_index, int64_t stride_row, int64_t stride_col,
                              const int64_t* edge_type, const int64_t* edge_time, const int64_t* node_type,
                              int64_t N, int64_t NQ, int64_t E, int32_t T, int32_t R,
                              void* plan, uint64_t plan_bytes, void* tmp, uint64_t tmp_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    hgt_plan_sizes sz;
    int rc = hgt_plan_sizes_for(N, E, T, R, &sz);
    if (rc != HGT_OK) return rc;
    if (!plan || !tmp || !node_type || (E > 0 && (!edge_index || !edge_type)) || NQ < 0 || NQ > N) return HGT_ERR_INVALID_ARG;
    if (plan_bytes < sz.plan_bytes || tmp_bytes < sz.tmp_bytes) return HGT_ERR_WORKSPACE;

    HgtPlanLayout L = hgt_plan_layout(N, E, T, R);
    uint64_t st = 0;
    sort_tmp_query(N, E, T, L, &st);
    TmpLayout tl = tmp_layout(N, E, L, st);
    char* pb = (char*)plan;
    char* tb = (char*)tmp;
    HgtPlanHeader* hdr = (HgtPlanHeader*)(pb + L.off_hdr);
    int32_t* esrc = (int32_t*)(pb + L.off_esrc);
    int32_t* edst = (int32_t*)(pb + L.off_edst);
    uint16_t* ertei = (uint16_t*)(pb + L.off_ertei);
    int32_t* eid = (int32_t*)(pb + L.off_eid);
    int32_t* segptr = (int32_t*)(pb + L.off_segptr);
    HgtItem* items = (HgtItem*)(pb + L.off_items);
    int32_t* tile_items = (int32_t*)(pb + L.off_tile_items);
    int32_t* rows_all = (int32_t*)(pb + L.off_rows_all);
    int32_t* off_all = (int32_t*)(pb + L.off_off_all);
    int32_t* rows_q = (int32_t*)(pb + L.off_rows_q);
    int32_t* off_q = (int32_t*)(pb + L.off_off_q);
    int32_t* hub_slot = (int32_t*)(pb + L.off_hub_slot);
    int32_t* hub_list = (int32_t*)(pb + L.off_hub_list);
    uint32_t* keys_in = (uint32_t*)(tb + tl.off_keys_in);
    uint32_t* keys_out = (uint32_t*)(tb + tl.off_keys_out);
    int32_t* vals_in = (int32_t*)(tb + tl.off_vals_in);
    int32_t* pair_cnt = (int32_t*)(tb + tl.off_pair_cnt);
    int32_t* pair_off = (int32_t*)(tb + tl.off_pair_off);
    uint32_t* nkeys_in = (uint32_t*)(tb + tl.off_nkeys_in);
    uint32_t* nkeys_out = (uint32_t*)(tb + tl.off_nkeys_out);
    int32_t* nvals_in = (int32_t*)(tb + tl.off_nvals_in);
    void* sort_tmp = (void*)(tb + tl.off_sort_tmp);
    size_t sort_bytes = (size_t)tl.sort_tmp_bytes;

    const int BS = 256;
    k_init_header<<<1, 64, 0, stream>>>(hdr);
    if (E > 0) {
        k_edge_keys<<<nblk(E, BS), BS, 0, stream>>>(edge_index, stride_row, stride_col, edge_type, node_type, N, NQ, E, T, R,
                                                   keys_in, vals_in, hdr);
        // stable: equal (tile, rel, dst) keep original edge order -> deterministic per-segment summation order
        if (rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys_out, vals_in, eid, (size_t)E, 0,
                                      key_bits((uint64_t)L.n_bins), stream) != hipSuccess) return HGT_ERR_LAUNCH;
        k_edge_fill<<<nblk(E, BS), BS, 0, stream>>>(edge_index, stride_row, stride_col, edge_time, node_type, N, NQ, E, T,
                                                   eid, esrc, edst, ertei, hdr);
    }
    k_segptr<<<nblk(L.n_bins + 1, BS), BS, 0, stream>>>(keys_out, E, L.n_bins, segptr);
    if (L.n_pairs + 1 <= 8192) {      // sampled batches: counts + scan in one workgroup (c2's 35 k pairs: 189 us this way, ~25 as three launches)
        k_pair_scan_small<<<1, 1024, 0, stream>>>(segptr, L.n_pairs, hgt_item_edges(E), pair_off);
    } else {
        k_pair_counts<<<nblk(L.n_pairs + 1, BS), BS, 0, stream>>>(segptr, L.n_pairs, hgt_item_edges(E), pair_cnt);
        sort_bytes = (size_t)tl.sort_tmp_bytes;
        if (rocprim::exclusive_scan(sort_tmp, sort_bytes, pair_cnt, pair_off, 0, (size_t)(L.n_pairs + 1),
                                    rocprim::plus<int32_t>(), stream) != hipSuccess) return HGT_ERR_LAUNCH;
    }
    k_items<<<nblk(L.n_pairs + 1, BS), BS, 0, stream>>>(segptr, pair_off, L.n_pairs, R, hgt_item_edges(E), items, tile_items, hdr);
    if (N > 0) k_hub_detect<<<nblk(N, BS), BS, 0, stream>>>(segptr, N, R, hub_slot, hub_list, hdr);

    // typed row lists: all nodes, and target nodes [0, NQ)
    if (N > 0 && N <= SMALL_ROWS_N && T <= SMALL_ROWS_T) {      // sampled batches: one workgroup (same lists as the stable sorts below)
        k_node_rows_small<<<1, 1024, 0, stream>>>(node_type, N, NQ, T, rows_all, off_all, rows_q, off_q, hdr);
    } else {
        if (N > 0) {
            k_node_keys<<<nblk(N, BS), BS, 0, stream>>>(node_type, N, T, nkeys_in, nvals_in);
            sort_bytes = (size_t)tl.sort_tmp_bytes;
            if (rocprim::radix_sort_pairs(sort_tmp, sort_bytes, nkeys_in, nkeys_out, nvals_in, rows_all, (size_t)N, 0,
                                          key_bits((uint64_t)T + 1), stream) != hipSuccess) return HGT_ERR_LAUNCH;
        }
        k_type_offsets<<<1, 256, 0, stream>>>(nkeys_out, N, T, off_all, nullptr);
        if (NQ > 0) {
            sort_bytes = (size_t)tl.sort_tmp_bytes;
            if (rocprim::radix_sort_pairs(sort_tmp, sort_bytes, nkeys_in, nkeys_out, nvals_in, rows_q, (size_t)NQ, 0,
                                          key_bits((uint64_t)T + 1), stream) != hipSuccess) return HGT_ERR_LAUNCH;
        }
        k_type_offsets<<<1, 256, 0, stream>>>(nkeys_out, NQ, T, off_q, hdr);
    }
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

// the 16-byte plan header (n_items, bad_index, n_hubs, n_unknown_q) -> pinned host memory, asynchronously on `stream`
extern "C" int hgt_plan_header_to_host(const void* plan, void* host_dst, void* stream) {
    if (!plan || !host_dst) return HGT_ERR_INVALID_ARG;
    if (hipMemcpyAsync(host_dst, plan, 16, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return HGT_ERR_LAUNCH;
    return HGT_OK;
}

extern "C" int hgt_plan_from_sorted(const int32_t* src, const int32_t* dst, const int32_t* edge_time, const int32_t* rel_ptr,
                                    const int32_t* type_off, int64_t N, int64_t NQ, int64_t E, int32_t T, int32_t R, void* plan,
                                    uint64_t plan_bytes, void* tmp, uint64_t tmp_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    hgt_plan_sizes sz;
    int rc = hgt_plan_sizes_for(N, E, T, R, &sz);
    if (rc != HGT_OK) return rc;
    if (!plan || !tmp || !rel_ptr || !type_off || (E > 0 && (!src || !dst)) || NQ < 0 || NQ > N) return HGT_ERR_INVALID_ARG;
    if (plan_bytes < sz.plan_bytes || tmp_bytes < sz.tmp_bytes) return HGT_ERR_WORKSPACE;
    HgtPlanLayout L = hgt_plan_layout(N, E, T, R);
    char* pb = (char*)plan;
    char* tb = (char*)tmp;
    HgtPlanHeader* hdr = (HgtPlanHeader*)(pb + L.off_hdr);
    int32_t* esrc = (int32_t*)(pb + L.off_esrc);
    int32_t* edst = (int32_t*)(pb + L.off_edst);
    uint16_t* ertei = (uint16_t*)(pb + L.off_ertei);
    int32_t* eid = (int32_t*)(pb + L.off_eid);
    int32_t* segptr = (int32_t*)(pb + L.off_segptr);
    HgtItem* items = (HgtItem*)(pb + L.off_items);
    int32_t* tile_items = (int32_t*)(pb + L.off_tile_items);
    // scratch: four int32 arrays of n_pairs + 1 entries inside the (much larger) sort scratch of hgt_plan_build
    const uint64_t stride = hgt_align_up((uint64_t)(L.n_pairs + 1) * 4, 256);
    if (tmp_bytes < 5 * stride) return HGT_ERR_WORKSPACE;
    int32_t* lb = (int32_t*)tb;
    int32_t* cnt = (int32_t*)(tb + stride);
    int32_t* icnt = (int32_t*)(tb + 2 * stride);
    int32_t* base = (int32_t*)(tb + 3 * stride);
    int32_t* pair_off = (int32_t*)(tb + 4 * stride);
    const int BS = 256;
    const int ch = hgt_item_edges(E);
    if (L.n_pairs + 1 <= SMALL_PAIRS && N <= 262144 && T + 2 <= 1024) {
        // sampled-batch sizes: three launches (pair table + scans in one workgroup; scatter + segptr + items side by side; hubs)
        k_sorted_small_head<<<1, 1024, 0, stream>>>(dst, rel_ptr, type_off, N, NQ, T, R, (int)L.n_pairs, ch, lb, cnt, base, pair_off,
                                                    (int32_t*)(pb + L.off_rows_all), (int32_t*)(pb + L.off_off_all),
                                                    (int32_t*)(pb + L.off_rows_q), (int32_t*)(pb + L.off_off_q), hdr);
        const unsigned nb_e = nblk(E, BS), nb_b = nblk(L.n_bins + 1, BS), nb_i = nblk(L.n_pairs + 1, BS);
        k_sorted_small_body<<<nb_e + nb_b + nb_i, BS, 0, stream>>>(src, dst, edge_time, rel_ptr, type_off, lb, cnt, base, pair_off, E, T, R, N,
                                                                  NQ, L.n_bins, (int)L.n_pairs, ch, nb_e, nb_b, esrc, edst, ertei, eid, segptr,
                                                                  items, tile_items, hdr);
        if (N > 0) k_hub_detect<<<nblk(N, BS), BS, 0, stream>>>(segptr, N, R, (int32_t*)(pb + L.off_hub_slot), (int32_t*)(pb + L.off_hub_list), hdr);
        HGT_CHECK_LAUNCH();
        return HGT_OK;
    }
    k_sorted_rows<<<nblk(std::max<int64_t>(N, T + 2), BS), BS, 0, stream>>>(type_off, N, NQ, T, (int32_t*)(pb + L.off_rows_all),
                                                                           (int32_t*)(pb + L.off_off_all), (int32_t*)(pb + L.off_rows_q),
                                                                           (int32_t*)(pb + L.off_off_q), hdr);
    k_sorted_pair_counts<<<nblk(L.n_pairs + 1, BS), BS, 0, stream>>>(dst, rel_ptr, L.n_pairs, R, ch, lb, cnt, icnt);
    k_scan2_single<<<1, 1024, 0, stream>>>(cnt, icnt, L.n_pairs + 1, base, pair_off);
    if (E > 0)
        k_sorted_scatter<<<nblk(E, BS), BS, 0, stream>>>(src, dst, edge_time, rel_ptr, type_off, lb, base, E, T, R, N, NQ, esrc, edst, ertei,
                                                        eid, hdr);
    k_sorted_segptr<<<nblk(L.n_bins + 1, BS), BS, 0, stream>>>(dst, lb, cnt, base, L.n_bins, R, E, segptr);
    k_items<<<nblk(L.n_pairs + 1, BS), BS, 0, stream>>>(segptr, pair_off, L.n_pairs, R, ch, items, tile_items, hdr);
    if (N > 0) k_hub_detect<<<nblk(N, BS), BS, 0, stream>>>(segptr, N, R, (int32_t*)(pb + L.off_hub_slot), (int32_t*)(pb + L.off_hub_list), hdr);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
