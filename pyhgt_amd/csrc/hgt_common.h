// Shared definitions of the libhgt_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hgt_hip.h"

#ifndef HGT_TD
#define HGT_TD 256       // destination nodes per tile (sort key = (dst/TD, relation, dst%TD)); multiple of 64.
                         // measured at c2: 64 -> 2.76 ms, 128 -> 2.69 ms, 256 -> 2.65 ms for the logits kernel (longer work items
                         // amortise the per-item relation-fragment load); the aggregation kernel walks 16-target sub-tiles either way
#endif
#ifndef HGT_CH
#define HGT_CH 512       // max edges per wavefront work item
#endif
#define HGT_WAVE 64
#ifndef HGT_MIN_ITEM
#define HGT_MIN_ITEM 16   // shortest logits work item (edges)
#endif
#ifndef HGT_HUB_DEG
#define HGT_HUB_DEG 1024  // targets with more in-edges are "hubs": aggregated by many wavefronts (hgt_edge.hip, hub path)
#endif

#define HGT_CHECK_LAUNCH()                          \
    do {                                            \
        if (hipGetLastError() != hipSuccess) return HGT_ERR_LAUNCH; \
    } while (0)

// hgt_gemm_xs.hip: the x-stationary form of the split typed linear (1 = launched, 0 = not its domain, < 0 = error)
int hgt_typed_linear_xs_try(bool f16, const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                            int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias, int64_t bgs, float* out0,
                            float* out1, float* out2, int32_t block_cols, int32_t by_pos, int32_t prologue, void* stream);

// hgt_gemm_tile.hip: the latency-regime form of the split typed linear and of the fused update (1 = launched, 0 = not its domain, < 0 = error)
struct HgtTileUpdate {
    const float* x_skip; int64_t ld_skip; const float* skip; const float* ln_w; const float* ln_b; int use_norm;
};
int hgt_typed_linear_tile_try(bool f16, const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                              int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias, int64_t bgs, float* out0,
                              float* out1, float* out2, int32_t block_cols, int32_t by_pos, int32_t prologue, const HgtTileUpdate* upd,
                              void* stream);

// hgt_edge_aggregate_update / _f16x3 / _range with the kernel-selection flags of hgt_conv_forward (q_end < 0: the whole graph)
int hgt_edge_aggregate_update_sel(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R, int32_t H, int32_t dk_pad, const float* logits,
                                  const float* V, const float* rte_v, const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows,
                                  void* hub_ws, int32_t* pending, const int64_t* node_type, const void* w_a_split, const float* b_a,
                                  const float* x_skip, int64_t ld_skip, const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm,
                                  int32_t n_out, float* out, void* stream, int64_t q_begin, int64_t q_end, int32_t frag_f16,
                                  int32_t hub_deterministic, int32_t use_ring);

static inline uint64_t hgt_align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Device-written header at the start of the plan buffer.
struct HgtPlanHeader {
    int32_t n_items;      // number of valid work items (written by the build)
    int32_t bad_index;    // bit 0: an edge endpoint was outside [0, n_nodes) / target >= n_q_rows; bit 1: edge_time outside [0, 240);
                          // bit 2: hgt_plan_from_sorted was handed edges that are not relation-grouped / target-sorted
    int32_t n_hubs;       // number of hub targets (in-degree > HGT_HUB_DEG), see hub_slot / hub_list
    int32_t n_unknown_q;  // target rows [0, n_q_rows) whose node type is outside [0, n_types): their output rows are zeroed
    int32_t pad[12];
};

// One wavefront work item: sorted edge positions [beg, end) all in one (dst tile, relation) bucket.
struct __attribute__((aligned(16))) HgtItem {
    int32_t beg, end, rel, tile;
};

// Edges per logits work item.  HGT_CH for large graphs; small graphs (the sampled subgraphs of the reference: E ~ 30k)
// get shorter items so that there are a few thousand wavefronts instead of ~100 walking 512 edges each one after the other
// (round 3: down to 16 edges -- at E = 31k the logits kernel ran 27 us with 64-edge items, one dependent batch chain per item).
static inline int hgt_item_edges(int64_t E) {
    int ch = HGT_CH;
    while (ch > HGT_MIN_ITEM && E / ch < 4096) ch >>= 1;   // (16-edge items at the sizes of the reference's sampled batches: E ~ 30-150k)
    return ch;
}

// Byte offsets of the arrays inside the plan buffer; a pure function of (N, E, T, R).
struct HgtPlanLayout {
    uint64_t off_hdr, off_esrc, off_edst, off_ertei, off_eid, off_segptr, off_items, off_tile_items;
    uint64_t off_rows_all, off_off_all, off_rows_q, off_off_q, off_hub_slot, off_hub_list, total;
    int64_t n_tiles, n_bins, n_pairs, max_items, max_hubs;
};

static inline HgtPlanLayout hgt_plan_layout(int64_t N, int64_t E, int32_t T, int32_t R) {
    HgtPlanLayout L;
    L.n_tiles = (N + HGT_TD - 1) / HGT_TD;
    L.n_pairs = L.n_tiles * (R + 1);
    L.n_bins = L.n_pairs * HGT_TD;
    L.max_items = L.n_pairs + E / hgt_item_edges(E) + 1;
    uint64_t o = 0;
    auto take = [&](uint64_t bytes) { uint64_t r = o; o = hgt_align_up(o + bytes, 256); return r; };
    L.off_hdr = take(sizeof(HgtPlanHeader));
    L.off_esrc = take((uint64_t)E * 4);
    L.off_edst = take((uint64_t)E * 4);
    L.off_ertei = take((uint64_t)E * 2);
    L.off_eid = take((uint64_t)E * 4);
    L.off_segptr = take((uint64_t)(L.n_bins + 1) * 4);
    L.off_items = take((uint64_t)L.max_items * sizeof(HgtItem));
    L.off_tile_items = take((uint64_t)(L.n_tiles + 1) * 4);
    L.off_rows_all = take((uint64_t)N * 4);
    L.off_off_all = take((uint64_t)(T + 2) * 4);
    L.off_rows_q = take((uint64_t)N * 4);
    L.off_off_q = take((uint64_t)(T + 2) * 4);
    L.max_hubs = E / HGT_HUB_DEG + 1;
    L.off_hub_slot = take((uint64_t)N * 4);
    L.off_hub_list = take((uint64_t)L.max_hubs * 4);
    L.total = o;
    return L;
}

struct HgtPlanView {
    const HgtPlanHeader* hdr;
    const int32_t* esrc;
    const int32_t* edst;
    const uint16_t* ertei;
    const int32_t* eid;
    const int32_t* segptr;
    const HgtItem* items;
    const int32_t* tile_items;   // items of dst tile t = items[tile_items[t] .. tile_items[t+1])
    const int32_t* rows_all;
    const int32_t* off_all;
    const int32_t* rows_q;
    const int32_t* off_q;
    const int32_t* hub_slot;   // [N]: index into the hub buffers or -1
    const int32_t* hub_list;   // [n_hubs]: target id of each hub
    HgtPlanLayout L;
};

static inline HgtPlanView hgt_plan_view(const void* plan, int64_t N, int64_t E, int32_t T, int32_t R) {
    HgtPlanView v;
    v.L = hgt_plan_layout(N, E, T, R);
    const char* b = (const char*)plan;
    v.hdr = (const HgtPlanHeader*)(b + v.L.off_hdr);
    v.esrc = (const int32_t*)(b + v.L.off_esrc);
    v.edst = (const int32_t*)(b + v.L.off_edst);
    v.ertei = (const uint16_t*)(b + v.L.off_ertei);
    v.eid = (const int32_t*)(b + v.L.off_eid);
    v.segptr = (const int32_t*)(b + v.L.off_segptr);
    v.items = (const HgtItem*)(b + v.L.off_items);
    v.tile_items = (const int32_t*)(b + v.L.off_tile_items);
    v.rows_all = (const int32_t*)(b + v.L.off_rows_all);
    v.off_all = (const int32_t*)(b + v.L.off_off_all);
    v.rows_q = (const int32_t*)(b + v.L.off_rows_q);
    v.off_q = (const int32_t*)(b + v.L.off_off_q);
    v.hub_slot = (const int32_t*)(b + v.L.off_hub_slot);
    v.hub_list = (const int32_t*)(b + v.L.off_hub_list);
    return v;
}

static inline int hgt_layout_compute(int32_t d_out, int32_t n_heads, hgt_layout* o) {
    if (d_out <= 0 || n_heads <= 0 || d_out % n_heads != 0) return HGT_ERR_INVALID_ARG;
    if (n_heads > 64) return HGT_ERR_UNSUPPORTED;
    // head counts that do not divide 64 (the reference accepts any d % n_heads == 0, conv.py:21: 3, 5, 6, 12 ...) run in the
    // layout of the next power of two: the extra heads are all-zero (zero weight rows / relation matrices), contribute
    // nothing to any real column and are cut away again at the boundary (hgt_relation_pack, hgt_att_export)
    int heads = 1;
    while (heads < n_heads) heads *= 2;
    int dk = d_out / n_heads;
    int lph = 64 / heads;                   // lanes per head
    int vec = 1;
    while (vec * lph < dk) vec *= 2;
    if (vec > 16) return HGT_ERR_UNSUPPORTED;
    o->d_k = dk;
    o->vec = vec;
    o->dk_pad = vec * lph;
    o->d_pad = 64 * vec;
    o->heads = heads;
    return HGT_OK;
}
