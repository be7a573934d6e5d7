/*
 * hgt_hip.h -- C ABI of libhgt_hip.so: MI355X (gfx950) kernels for the forward pass of the
 * Heterogeneous Graph Transformer convolution (acbull/pyHGT, pyHGT/conv.py::HGTConv).
 *
 * Boundary conventions:
 *   - extern "C", plain pointers and sizes, no torch / C++ types in any signature;
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - every call enqueues work on the given hipStream_t (passed as void*) and returns; nothing
 *     here synchronises the stream, allocates or frees device memory (the caller owns every
 *     buffer, sized with the *_sizes() helpers);
 *   - return value 0 = HGT_OK, negative = error (hgt_strerror()); nothing throws or exits;
 *   - re-entrant and thread-safe as long as concurrent calls use distinct workspaces.
 *
 * Each entry point cites the reference code (file:line under /root/reference) it replaces.
 * The Python binding that calls these through ctypes is pyhgt_amd/_lib.py; the binding a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 */
#ifndef HGT_HIP_H_
#define HGT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HGT_ABI_VERSION 7

/* error codes */
#define HGT_OK 0
#define HGT_ERR_INVALID_ARG (-1)   /* null pointer, negative size, d % n_heads != 0, ... */
#define HGT_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels are instantiated for */
#define HGT_ERR_WORKSPACE (-3)     /* caller-provided buffer too small */
#define HGT_ERR_TOO_LARGE (-4)     /* N or E beyond 32-bit plan indices */
#define HGT_ERR_LAUNCH (-5)        /* HIP runtime reported a launch error */

/* length of the relative-temporal-encoding table, conv.py:287 (max_len = 240) */
#define HGT_RTE_LEN 240

const char* hgt_strerror(int code);
int hgt_abi_version(void);
/* Bit mask of optional parts compiled into this library.  Bit 0 (HGT_FEATURE_LAB_KERNELS): the measured-and-not-adopted kernels of
 * csrc/lab/ (make LAB=1) -- only then do HGT_FLAG_RING_AGGREGATE / HGT_FLAG_SINGLE_PASS select anything; the shipped library is
 * built without them and ignores both flags. */
#define HGT_FEATURE_LAB_KERNELS 1
int hgt_build_features(void);

/* ----------------------------------------------------------------------------------------------
 * Internal "head-padded" feature layout.  Q/K/V/agg rows hold n_heads blocks of dk_pad floats
 * (dk_pad >= d_k = d_out / n_heads, extra columns are zero) so that one 64-lane wavefront covers
 * a row with `vec` contiguous floats per lane and every head maps to dk_pad / vec adjacent lanes.
 * Replaces the .view(-1, n_heads, d_k) of conv.py:96-97,103.
 * Requires: d_out % n_heads == 0, n_heads <= 16.  A head count that does not divide 64 (3, 5, 6, 12: legal in the reference,
 * conv.py:21) runs in the layout of the next power of two `heads`: the extra heads are all-zero and never reach an output.
 * Every kernel-level entry point below takes the LAYOUT head count (`heads`) as its n_heads; hgt_conv_forward, the workspace
 * helpers, hgt_relation_pack and hgt_att_export take the model's real head count as well.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hgt_layout {
    int32_t d_k;     /* d_out / n_heads                         */
    int32_t dk_pad;  /* padded head width                       */
    int32_t d_pad;   /* heads * dk_pad = 64 * vec               */
    int32_t vec;     /* floats per lane (1,2,4,8,16)            */
    int32_t heads;   /* head count of the layout: n_heads rounded up to a power of two (ABI 3) */
} hgt_layout;
int hgt_layout_for(int32_t d_out, int32_t n_heads, hgt_layout* out_host);

/* ----------------------------------------------------------------------------------------------
 * Graph plan: everything that depends only on (edge_index, edge_type, edge_time, node_type).
 * Built once per sampled subgraph and shared by all layers (the reference reuses the same graph
 * tensors for every layer, model.py:78-79).  Replaces, for the whole layer, PyG's per-call
 * index_select gathers (MessagePassing.propagate, called at conv.py:57), the T*T*R boolean mask
 * cube and its host syncs (conv.py:71-84) and the per-type masks of update() (conv.py:121-123):
 *   - edges stably sorted by (dst / TD, relation, dst % TD) (TD = tile size, hgt_plan_constants), ids int64 -> int32,
 *     source-type*240 + edge_time folded into one uint16 per edge;
 *   - a segment table over (dst tile, relation, dst) and a list of wavefront work items
 *     (a bounded run of consecutive edges of one (dst tile, relation));
 *   - nodes stably sorted by type (row lists for the typed linears) + per-type offsets.
 * Edges whose relation id or endpoint node types fall outside [0,R) / [0,T) go to an extra
 * "unclaimed" relation bucket: logit 0, message 0, still part of the softmax -- the behaviour of
 * the reference's zero-initialised res_att / res_msg (conv.py:68-69).
 * ---------------------------------------------------------------------------------------------- */
typedef struct hgt_plan_sizes {
    uint64_t plan_bytes;   /* persistent plan buffer                         */
    uint64_t tmp_bytes;    /* scratch needed only during hgt_plan_build      */
    int64_t  max_items;    /* upper bound on wavefront work items            */
    int64_t  n_bins;       /* (dst tiles) * (R+1) * tile size                */
} hgt_plan_sizes;

int hgt_plan_sizes_for(int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                       hgt_plan_sizes* out_host);
/* build-time constants of the plan: destination tile size and the edge cap of one work item */
int hgt_plan_constants(int32_t* tile_nodes_host, int32_t* item_edges_host);
/* edges per logits work item for a graph of n_edges edges (<= the maximum hgt_plan_constants reports; smaller for small
 * graphs so that enough wavefronts exist) */
int hgt_plan_item_edges(int64_t n_edges, int32_t* item_edges);

/* Row lists exported by a built plan (device pointers INTO the plan buffer), for hgt_typed_linear:
 *   rows_all / off_all : all n_nodes nodes stably sorted by type, int32[n_nodes] / int32[T+2]
 *   rows_q   / off_q   : the target nodes [0, n_q_rows) sorted by type (identical to the above
 *                        when n_q_rows == n_nodes); group T holds nodes with an out-of-range type. */
typedef struct hgt_plan_rows {
    const int32_t* rows_all; const int32_t* off_all;
    const int32_t* rows_q;   const int32_t* off_q;
} hgt_plan_rows;
int hgt_plan_row_lists(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types,
                       int32_t n_relations, hgt_plan_rows* out_host);

/* ABI 6: the per-tile item table of a plan: int32[*n_tiles + 1] at byte offset *offset of the plan buffer; the logits work items of
 * destination tile t (tile_nodes targets each, hgt_plan_constants) are [table[t], table[t + 1]).  pyhgt_amd.dist reads it once per
 * graph to launch the edge phase of ONE target block (hgt_conv_forward stage 5). */
int hgt_plan_tile_items_offset(int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, uint64_t* offset_host,
                               int64_t* n_tiles_host);

/* edge_index: int64, element (row, e) at edge_index[row * stride_row + e * stride_col]; row 0 =
 * source j, row 1 = target i (conv.py:60-63, data.py:245,254 -- the reference hands over the
 * (1,2)-strided transpose of an [E,2] tensor).  edge_time may be NULL (use_RTE = False).
 * n_q_rows: nodes [0, n_q_rows) are targets that get Q / aggregation / update; nodes beyond are
 * source-only halo rows (multi-GPU destination partitioning); pass n_nodes on a single GPU.
 * Every edge target must be < n_q_rows. */
int hgt_plan_build(const int64_t* edge_index, int64_t stride_row, int64_t stride_col,
                   const int64_t* edge_type, const int64_t* edge_time, const int64_t* node_type,
                   int64_t n_nodes, int64_t n_q_rows, int64_t n_edges, int32_t n_types, int32_t n_relations,
                   void* plan, uint64_t plan_bytes, void* tmp, uint64_t tmp_bytes, void* stream);

/* ABI 5: the first 16 bytes of a plan (int32 n_items, bad_index, n_hubs, n_unknown_q) copied to PINNED host memory without a
 * synchronisation (one hipMemcpyAsync on `stream`): how the host learns "no hub target / no unknown rows / malformed ids". */
int hgt_plan_header_to_host(const void* plan, void* host_dst_pinned, void* stream);

/* Plan from a graph that is ALREADY in the order the reference's sampler produces (data.py:183-209,
 * 227-246): nodes type-contiguous with ascending types (type t = ids [type_off[t], type_off[t+1])), edges grouped by relation
 * (relation r = positions [rel_ptr[r], rel_ptr[r+1]) of src / dst / edge_time) with NON-DECREASING target ids inside a
 * relation, ids already int32.  Builds the same plan as hgt_plan_build (bit-identical arrays; eid = position in the given
 * order) without its radix sorts: seven small launches, no 64-bit index traffic.  pyhgt_amd.sampled.to_device_graph is the
 * sibling of to_torch (data.py:212-256) that emits this form.  Same buffer sizes as hgt_plan_build (hgt_plan_sizes_for).
 * edge_time may be NULL.  All arrays are DEVICE arrays. */
int hgt_plan_from_sorted(const int32_t* src, const int32_t* dst, const int32_t* edge_time, const int32_t* rel_ptr,
                         const int32_t* type_off, int64_t n_nodes, int64_t n_q_rows, int64_t n_edges, int32_t n_types,
                         int32_t n_relations, void* plan, uint64_t plan_bytes, void* tmp, uint64_t tmp_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Typed (grouped) linear layer on MFMA: for every node type t and every row n of that type
 *     y[n, :] = prologue(x[n, :]) @ W[t]^T + b[t]
 * Replaces the per-meta-relation, per-EDGE nn.Linear calls of conv.py:96-97,103 (Q, K, V) and
 * the per-type a_linear of conv.py:125, plus RelTemporalEncoding.lin (conv.py:297-299) when
 * building the temporal tables.  Row lists come from the plan (or any caller-made row list).
 *   x         [*, ldx] fp32 rows, gathered through rows[] (node ids)
 *   rows      int32[n_rows] row ids grouped by type; group g = rows[group_off[g] .. group_off[g+1])
 *   group_off int32[n_groups+1] (device)
 *   W         fp32 [n_groups][n_out][k] with stride w_group_stride floats between groups
 *   bias      fp32 [n_groups][n_out] with stride b_group_stride, or NULL
 *   out0..2   the n_out columns are split into blocks of block_cols columns, block c is written
 *             to out{c}[row * block_cols + col]; row = rows[p] (out_by_position = 0) or p (= 1)
 *   prologue  0 = none, 1 = exact (erf) GELU applied to x on load (conv.py:119); split variants only (ABI 6): 2 = x holds rows in the
 *             24-bit transport format of the multi-GPU exchange (hgt_gather_rows_c24; ldx = 3 k / 4 dwords, k <= 256, k % 4 == 0),
 *             decoded by the kernel's loader -- the halo rows a rank receives are projected straight off the wire buffer.
 *             Split variants: | HGT_LINEAR_FORCE_XS takes the x-stationary kernel (hgt_gemm_xs.hip) for every shape it covers
 *             whatever the row count (default: from 262 144 rows, K = 512 from 65 536), | HGT_LINEAR_NO_XS never takes it -- the
 *             two kernels are bit-identical; the bits exist for tests and A/B timings (no environment variable is read)
 *   precision must be 0 (fp32 MFMA, exact fp32 FMA chain); the split-bf16 variant is below
 * ---------------------------------------------------------------------------------------------- */
#define HGT_LINEAR_FORCE_XS 0x100
#define HGT_LINEAR_NO_XS 0x200
#define HGT_LINEAR_TANH 0x1000     /* ABI 7, split variants: tanh applied to the output (model.py:70-76: the GNN's typed adapter + tanh in one kernel).
                                    * Only the latency-regime tile kernel and the K > 256 slab kernel carry it: HGT_ERR_UNSUPPORTED otherwise (and
                                    * nothing launched) -> plain call + hgt_tanh_inplace */
#define HGT_LINEAR_NO_TILE 0x400   /* ABI 7: never the latency-regime tile kernel (hgt_gemm_tile.hip; the default below 49 152 rows) -- it is
                                    * bit-identical to the slab kernels; the bit exists for tests and A/B timings */
int hgt_typed_linear(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off,
                     int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_out,
                     const float* W, int64_t w_group_stride, const float* bias, int64_t b_group_stride,
                     float* out0, float* out1, float* out2, int32_t block_cols,
                     int32_t out_by_position, int32_t prologue, int32_t precision, void* stream);

/* Split-bf16 x3 variant of the typed linear layer (same contract as hgt_typed_linear): operands are
 * split into bf16 hi+mid terms and a product is evaluated as mid*hi + hi*mid + hi*hi on the bf16 matrix
 * cores with fp32 accumulation (relative error of a product <= ~3*2^-18).  W is split and tiled once per
 * forward by hgt_split_weights into a caller-owned buffer of hgt_split_weights_bytes() bytes. */
int hgt_split_weights_bytes(int32_t n_groups, int32_t k, int32_t n_out, uint64_t* out_host);
int hgt_split_weights(const float* W, int64_t w_group_stride, int32_t n_groups, int32_t k, int32_t n_out,
                      void* w_split, void* stream);
int hgt_typed_linear_bf16x3(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off,
                            int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_out, const void* w_split,
                            const float* bias, int64_t b_group_stride, float* out0, float* out1, float* out2,
                            int32_t block_cols, int32_t out_by_position, int32_t prologue, void* stream);

/* fp16 hi / lo variant of the split products (ABI 5, precision "f16x3"): the same three MFMAs per step with 11 + 11 mantissa bits
 * per operand instead of 8 + 8 (relative error of a product ~2^-22: the layer's output is within ~3e-7 of the fp64 result), at the
 * same cost.  fp16 has 5 exponent bits, so every x row is scaled by a power of two (found on the fly) and every weight group by
 * one (found by hgt_split_weights_f16, kept in the image's tail); the inverse scales go onto the fp32 accumulators.  Same
 * arguments and the same image size (hgt_split_weights_bytes) as the bf16 functions; an image must be used with its own kind. */
int hgt_split_weights_f16(const float* W, int64_t w_group_stride, int32_t n_groups, int32_t k, int32_t n_out,
                          void* w_split, void* stream);
int hgt_typed_linear_f16x3(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off,
                           int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_out, const void* w_split,
                           const float* bias, int64_t b_group_stride, float* out0, float* out1, float* out2,
                           int32_t block_cols, int32_t out_by_position, int32_t prologue, void* stream);

/* Introspection (host only, no GPU): the work decomposition of the x-stationary kernel behind hgt_typed_linear_bf16x3 / _f16x3 on
 * large inputs (csrc/hgt_gemm_xs.hip), enumerated with the kernel's own scheduling helpers.  group_off_host = the [n_groups + 1]
 * offsets as a HOST array, n_cu = number of workgroups the launch may use (the device's CU count), k selects the form (512: four
 * wavefronts per workgroup, otherwise eight).  items[i] = {workgroup, round, wavefront, group, first position in the row list,
 * rows (1..32)}; *n_items = number of entries the schedule has; HGT_ERR_TOO_LARGE (with *n_items set) if max_items is smaller.
 * Replaces nothing in the reference (its Linear layers have no decomposition to inspect); it exists so that the partition of the row
 * list -- every position exactly once, no round mixing two groups -- is tested on the CPU (tests/test_xs_schedule.py). */
int hgt_typed_linear_xs_schedule(const int32_t* group_off_host, int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_cu,
                                 int32_t* items_host, int64_t max_items, int64_t* n_items_host);

/* a_linear (conv.py:125) with the node update (conv.py:129-133, see hgt_node_update) fused into its epilogue:
 *   out[n] = LN_t( (agg[n] @ W_a[t]^T + b_a[t]) * sigmoid(skip[t]) + x_skip[n] * (1 - sigmoid(skip[t])) )
 * for the rows of every group; split-bf16 x3 MFMA; needs n_out % 4 == 0 and n_out <= 256, or n_out <= 512 with k <= 512 (round 5:
 * both 256-column passes of a row tile stay in registers, LayerNorm over the two together) -- HGT_ERR_UNSUPPORTED otherwise -> use
 * hgt_typed_linear[_bf16x3] + hgt_node_update.  Rows of no group are not written: hgt_zero_rows. */
int hgt_linear_update_bf16x3(const float* agg, int64_t ld_agg, const int32_t* rows, const int32_t* group_off,
                             int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_out, const void* w_split,
                             const float* bias, int64_t b_group_stride, const float* x_skip, int64_t ld_skip,
                             const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm, float* out,
                             void* stream);
/* the same on an fp16 hi / lo image (hgt_split_weights_f16) */
int hgt_linear_update_f16x3(const float* agg, int64_t ld_agg, const int32_t* rows, const int32_t* group_off,
                            int32_t n_groups, int64_t n_rows, int32_t k, int32_t n_out, const void* w_split,
                            const float* bias, int64_t b_group_stride, const float* x_skip, int64_t ld_skip,
                            const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm, float* out,
                            void* stream);
/* out[rows[i]][0..d) = 0 for i in [range[0], range[1]) -- range is a DEVICE array of two int32 */
int hgt_zero_rows(const int32_t* rows, const int32_t* range, int32_t d, float* out, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Relation parameter packing (per forward, R*H*dk*dk elements):
 *   att_t[r][h][c][k] = relation_att[r][h][k][c] * relation_pri[r][h] / sqrt(d_k)   (zero padded; heads >= n_heads: zero)
 *   msg_p[r][h][k][c] = relation_msg[r][h][k][c]                                     (zero padded)
 * both [R][H][dk_pad][dk_pad].  Folds conv.py:99's "* relation_pri / sqrt_dk" into the matrix and
 * transposes relation_att so that  q . (k A) = (A q) . k  can be evaluated on the target side.
 * ---------------------------------------------------------------------------------------------- */
int hgt_relation_pack(const float* relation_att, const float* relation_msg, const float* relation_pri,
                      int32_t n_relations, int32_t n_heads, int32_t n_heads_layout, int32_t d_k, int32_t dk_pad,
                      float* att_t, float* msg_p, void* stream);
/* n_heads = heads of the parameter tensors, n_heads_layout >= n_heads = heads of att_t / msg_p (extra heads: zero matrices) */

/* ----------------------------------------------------------------------------------------------
 * Edge phase (replaces conv.py:98-99,104,108-111 and PyG's scatter-add, conv.py:13 aggr='add').
 * All feature rows are in the head-padded layout, row stride d_pad.
 *   hgt_edge_logits    s[p][h] = (A'[rel] q[dst])_h . (K[src] + rte_k[src_type, dt])_h  for every
 *                      sorted edge position p (conv.py:98-99); unclaimed edges get 0
 *   hgt_edge_softmax   in place: s -> exp(s - max_i) / (sum_i exp(s - max_i) + 1e-16) over all
 *                      in-edges of each target, per head (PyG softmax, conv.py:108)
 *   hgt_edge_aggregate takes the RAW logits of hgt_edge_logits and evaluates the per-target softmax
 *                      online (same formula as above) while aggregating:
 *                      agg[i] = sum_rel (sum_{e in (i,rel)} att_e (V[src] + rte_v[...])) M[rel]
 *                      (conv.py:104,108-111 + scatter-add) for every target i < n_q_rows (isolated
 *                      targets get 0); apply_gelu != 0 stores gelu(agg) instead (conv.py:119).
 *                      hgt_edge_softmax is only needed to materialise att for hgt_att_export.
 *   hgt_att_export     att_out[original edge id][h] = att[p][h]   (self.att, conv.py:108)
 * rte_k / rte_v: [n_types*240][d_pad] tables or NULL.
 * ---------------------------------------------------------------------------------------------- */
int hgt_edge_logits(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                    int32_t n_heads, int32_t dk_pad, const float* Q, const float* K, const float* rte_k,
                    const float* att_t, float* logits, void* stream);
/* ABI 5: the same logits with the target-side transform  q~ = q A'[r]  on the matrix cores: att_frag = hgt_relation_frag_pack(att_t)
 * (frag_f16 = 0) or hgt_relation_frag_pack_f16(att_t) (frag_f16 = 1), 16 distinct targets of a work item at a time, 3-term split
 * products like the aggregation's.  The form for d_k >= 64 (the reference's own widths: n_hid 400 / 512 with 8 heads), where the
 * vector-ALU kernel needs a 4-way head-group split and is instruction-bound; layouts it does not cover fall through to
 * hgt_edge_logits (att_t is required for that).
 * frag_f16 bits (ABI 7, here and in hgt_edge_aggregate_items[_update]): bit 0 = the fp16 images; for d_k >= 64 (the runs kernel of the
 * item-parallel aggregation: >= 32) and the 16-edge work items of a sampled batch the four wavefronts of a workgroup SHARE the relation transform (a quarter of the fragment image each, kept
 * in registers while consecutive items share the relation; same result bit for bit): bit 1 = never, bit 2 = for larger items too. */
int hgt_edge_logits_mfma(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                         int32_t n_heads, int32_t dk_pad, const float* Q, const float* K, const float* rte_k,
                         const float* att_t, const void* att_frag, int32_t frag_f16, float* logits, void* stream);
/* ABI 6: the work items [item_begin, item_end) only = the items of a range of destination tiles (hgt_plan_tile_items_offset): the
 * logits of ONE target block of the multi-GPU path.  att_frag may be NULL (vector-ALU kernel). */
int hgt_edge_logits_range(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                          int32_t n_heads, int32_t dk_pad, const float* Q, const float* K, const float* rte_k,
                          const float* att_t, const void* att_frag, int32_t frag_f16, float* logits, int32_t item_begin,
                          int32_t item_end, void* stream);
int hgt_edge_softmax(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                     int32_t n_heads, float* logits_att, void* stream);
int hgt_edge_aggregate(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                       int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                       const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows, int32_t apply_gelu,
                       void* hub_ws, void* stream);
/* ABI 5, latency regime (sampled sub-graphs): the same aggregate (after it agg equals hgt_edge_aggregate's up to fp32 rounding), item-
 * parallel: one wavefront per logits work item sums the runs of consecutive same-target edges, transforms 16 runs per matrix-core
 * round and writes them to `scratch` (hgt_edge_aggregate_items_bytes); a second kernel combines every target's runs in (relation,
 * position) order like softmax partials.  No atomics, fixed order: bit-reproducible; hub targets need no hub_ws.  msg_frag is
 * required (frag_f16: which of hgt_relation_frag_pack / _f16 made it); apply_gelu 0 / 1; n_relations < 64. */
int hgt_edge_aggregate_items_bytes(int64_t n_edges, int32_t n_heads, int32_t dk_pad, uint64_t* out_host);
int hgt_edge_aggregate_items(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                             int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                             const void* msg_frag, int32_t frag_f16, float* agg, int64_t n_q_rows, int32_t apply_gelu,
                             void* scratch, uint64_t scratch_bytes, void* stream);
/* ABI 7: hgt_edge_aggregate_items whose merge pass IS the node update (sampled batches: one of the layer's five dependent kernels less and the
 * merged rows never written): per target the runs are merged exactly like hgt_edge_aggregate_items merges them (apply_gelu = 1), the row is
 * multiplied with W_a[type] on the matrix cores (split x3 like hgt_linear_update_*; w_a_split = hgt_split_weights[_f16](W_a) with
 * k = n_heads * dk_pad; frag_f16 selects the fp16 images of BOTH msg_frag and w_a_split) and the gated skip + LayerNorm of conv.py:129-133
 * is applied: out[n] for every row n of rows[] (rows / group_off: target rows grouped by node type, hgt_plan_row_lists rows_q / off_q).
 * Output identical to hgt_edge_aggregate_items + hgt_linear_update_*.  HGT_ERR_UNSUPPORTED: rows wider than 512 padded columns,
 * n_out > 512 or > the padded row, n_out % 4 != 0, ld_skip % 4 != 0 -> use the two-call form. */
int hgt_edge_aggregate_items_update(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                                    int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                                    const void* msg_frag, int32_t frag_f16, int64_t n_q_rows, void* scratch, uint64_t scratch_bytes,
                                    const int32_t* rows, const int32_t* group_off, int32_t n_groups, const void* w_a_split,
                                    const float* b_a, const float* x_skip, int64_t ld_skip, const float* skip, const float* ln_w,
                                    const float* ln_b, int32_t use_norm, int32_t n_out, float* out, void* stream);
/* ABI 6: logits AND the item-parallel aggregation in one walk over the edges (sampled sub-graphs): per group of <= 16 runs the
 * target-side transform q~ = q A'[r] on the matrix cores (att_frag = hgt_relation_frag_pack[_f16](att_t)), then per edge K and V
 * gathered together, the logit, the run's online softmax and the weighted sum, then the message transform (msg_frag) -- no [E][H]
 * logits array, one launch less per layer.  Same scratch, same fixed-order merge and (up to the fp32 summation order of the
 * logits) the same result as hgt_edge_logits_mfma + hgt_edge_aggregate_items.  rte_k / rte_v: both tables or both NULL.
 * HGT_ERR_UNSUPPORTED for layouts it is not instantiated for (the caller takes the two-kernel form) -- and, in the shipped library,
 * for every call that has work to do (n_edges > 0 and target rows > 0; an empty call returns HGT_OK like every entry point): the
 * kernel (csrc/lab/hgt_edge_single_pass.hip, measured not faster) is compiled into LAB builds only (hgt_build_features). */
int hgt_edge_single_pass_items(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                               int32_t n_heads, int32_t dk_pad, const float* Q, const float* K, const float* V, const float* rte_k,
                               const float* rte_v, const void* att_frag, const void* msg_frag, int32_t frag_f16, float* agg,
                               int64_t n_q_rows, int32_t apply_gelu, void* scratch, uint64_t scratch_bytes, void* stream);
/* ABI 4: one slice [rel_lo, rel_hi) of the plan's n_relations + 1 relation buckets (bucket n_relations = unclaimed edges).
 * hgt_edge_logits_slice writes the logits of the slice's edges only.  hgt_edge_aggregate_slice (matrix-core kernel: msg_frag
 * required) aggregates the slice and combines it with what earlier slices left: state = f32[n_q_rows][n_heads][2] (softmax
 * reference, exp-sum) and the un-normalised rows in agg; has_prev = 0 for the first slice; more != 0: leave state + raw rows
 * for the next slice, more = 0: normalise (+ gelu) -- after it agg equals hgt_edge_aggregate's over all slices (partials merge as
 * m = max(m_a, m_b), x = x_a e^(m_a - m) + x_b e^(m_b - m): exact up to fp32 rounding).  Hub targets are processed with the
 * last slice, over all relations. */
int hgt_edge_logits_slice(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                          int32_t n_heads, int32_t dk_pad, const float* Q, const float* K, const float* rte_k,
                          const float* att_t, float* logits, int32_t rel_lo, int32_t rel_hi, void* stream);
int hgt_edge_aggregate_slice(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                             int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                             const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows, int32_t apply_gelu,
                             void* hub_ws, int32_t rel_lo, int32_t rel_hi, float* state, int32_t has_prev, int32_t more,
                             void* stream);
/* msg_frag (ABI 3): hgt_relation_frag_pack(msg_p) = the relation message matrices as bf16 hi/mid MFMA fragments.  Non-NULL:
 * the per-(target, relation) transforms  (sum att_e v_e) M[rel]  run on the matrix cores, 16 targets x one relation at a time,
 * as 3-term split-bf16 products with fp32 accumulation (relative error of a product <= ~3*2^-18, like the split-bf16 typed
 * linears).  NULL: exact fp32 mat-vecs on the vector ALU (the round-1 kernel; also taken for a head wider than 256 columns). */
int hgt_relation_frag_bytes(int32_t n_relations, int32_t n_heads, int32_t dk_pad, uint64_t* out_host);
int hgt_relation_frag_pack(const float* msg_p, int32_t n_relations, int32_t n_heads, int32_t dk_pad, void* msg_frag, void* stream);
/* fp16 hi / lo variant (ABI 5, precision "f16x3"; see hgt_split_weights_f16): the fragments hold M * s with ONE power-of-two s for
 * all relations (everything summed into one accumulator must share it; 1/s sits behind the fragments, hgt_relation_frag_bytes
 * counts it), the rows  sum att_e v_e  of a target are scaled by one power of two per target, chosen when its first row is
 * formed (2^7 of headroom, moved with an exact rescale of the accumulator column if a later relation needs more), and both
 * inverses join the softmax normalisation.  Relative error of a transform ~2^-22.  hgt_edge_aggregate_f16x3 takes the f16 image
 * (required); slices, hgt_edge_spmm and the hub path (exact fp32) are unchanged. */
int hgt_relation_frag_pack_f16(const float* msg_p, int32_t n_relations, int32_t n_heads, int32_t dk_pad, void* msg_frag, void* stream);
int hgt_edge_aggregate_f16x3(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                             int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                             const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows, int32_t apply_gelu,
                             void* hub_ws, void* stream);
/* hub_ws: scratch of hgt_hub_workspace_bytes() bytes for targets with more than 1024 in-edges ("hubs"): their edge
 * ranges are split over many wavefronts (max / sum-exp / weighted sum accumulated with atomics) instead of being
 * walked by the one wavefront that owns their 16-target sub-tile.  NULL = no hub path (correct, slow on hubs). */
int hgt_hub_workspace_bytes(int64_t n_edges, int32_t n_heads, int32_t dk_pad, uint64_t* out_host);
/* ABI 6: deterministic != 0 adds one partial slot per piece, (n_relations + 1) x 32 pieces per possible hub (n_edges / 1024 + 1 of them):
 * about 0.28 * n_edges * (H * dk_pad + H) * 4 bytes.  hgt_edge_aggregate_ex / hgt_edge_aggregate_update_range take such a buffer with
 * hub_deterministic = 1. */
int hgt_hub_workspace_bytes_ex(int64_t n_edges, int32_t n_heads, int32_t dk_pad, int32_t n_relations, int32_t deterministic,
                               uint64_t* out_host);
int hgt_edge_aggregate_ex(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                          int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                          const float* msg_p, const void* msg_frag, int32_t frag_f16, float* agg, int64_t n_q_rows, int32_t apply_gelu,
                          void* hub_ws, int32_t hub_deterministic, void* stream);
int hgt_att_export(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                   int32_t n_heads, const float* att_sorted, float* att_out, int32_t n_heads_out, void* stream);
/* att_sorted [E][n_heads] (layout heads) -> att_out [E][n_heads_out] (the first n_heads_out heads; the model's real heads) */

/* ----------------------------------------------------------------------------------------------
 * Node update epilogue (conv.py:129-133): per node n of type t (rows with a type outside [0,T)
 * are written as zeros, like the reference's zero-initialised `res`, conv.py:120):
 *     y = trans[n] * sigmoid(skip[t]) + x[n] * (1 - sigmoid(skip[t]));
 *     out[n] = use_norm ? LayerNorm(y; ln_w[t], ln_b[t], eps 1e-5) : y
 * ---------------------------------------------------------------------------------------------- */
int hgt_node_update(const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                    const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm,
                    int64_t n_nodes, int32_t d, int32_t n_types, float* out, void* stream);

/* Generalised form used by DenseHGTConv (conv.py:250-274): skip == NULL -> plain residual y = trans[n] + x[n]
 * (conv.py:259,271); ln_shared != 0 -> one LayerNorm for every type (out_norm, conv.py:272), ln_w/ln_b are [d].
 * out may alias x (every row is read completely before it is written). */
int hgt_node_update_ex(const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                       const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm, int32_t ln_shared,
                       int64_t n_nodes, int32_t d, int32_t n_types, float* out, void* stream);

/* x[i] = tanh(x[i]) in place: the activation of the typed input adapter of model.GNN (model.py:70-76) */
int hgt_tanh_inplace(float* x, int64_t n, void* stream);

/* Task heads of model.py, used by pyhgt_amd.Classifier / Matcher on the seed rows:
 *   hgt_log_softmax_rows: out[r] = log_softmax(x[r])   (torch.log_softmax(..., dim=-1), model.py:11)
 *   hgt_row_dot:          out[r] = scale * <x[r], y[r]> (Matcher pair=True, model.py:41,44) */
int hgt_log_softmax_rows(const float* x, int64_t n_rows, int32_t n_cols, float* out, void* stream);
int hgt_row_dot(const float* x, const float* y, int64_t n_rows, int32_t d, float scale, float* out, void* stream);

/* row gather used to pack halo rows for the multi-GPU exchange: out[i] = x[idx[i]] */
int hgt_gather_rows(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t d, float* out, void* stream);
/* The same rows in a 24-bit transport format (sign, 8 exponent, 15 mantissa bits, round to nearest: relative error <= 2^-16;
 * 3*d bytes per row, d % 4 == 0) and its inverse: the multi-GPU exchange is bound by the links, and halo rows only feed the
 * K/V projections. */
int hgt_gather_rows_c24(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t d, void* out, void* stream);
int hgt_unpack_rows_c24(const void* in, int64_t n, int32_t d, float* out, int64_t ld_out, void* stream);

/* hgt_edge_aggregate with HGTConv's node update (conv.py:119-133) fused in: the workgroup that aggregated 64 targets
 * multiplies their gelu'd rows with W_a on the matrix cores (split-bf16 x3, w_a_split = hgt_split_weights(W_a,
 * k = H*dk_pad, n_out)) and writes
 *     out[i] = LN_t( (gelu(agg_i) W_a[t]^T + b_a[t]) * sigmoid(skip[t]) + x_skip[i] * (1 - sigmoid(skip[t])) ),
 * rows of unknown type = 0, so agg never goes through HBM.  Needs H*dk_pad <= 256, n_out % 4 == 0 and a layout without
 * head-group split (HGT_ERR_UNSUPPORTED otherwise -> hgt_edge_aggregate + hgt_linear_update_bf16x3).
 * `agg` [n_q_rows][H*dk_pad] and `pending` [(n_q_rows+63)/64] int32 are scratch: workgroups that contain a hub target
 * (hub_ws != NULL) finish through agg and the hub kernels, exactly like hgt_edge_aggregate, and are updated last. */
int hgt_edge_aggregate_update(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                              int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                              const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows, void* hub_ws, int32_t* pending,
                              const int64_t* node_type, const void* w_a_split, const float* b_a, const float* x_skip,
                              int64_t ld_skip, const float* skip, const float* ln_w, const float* ln_b, int32_t use_norm,
                              int32_t n_out, float* out, void* stream);
/* the same with the fp16 hi / lo images: msg_frag = hgt_relation_frag_pack_f16, w_a_split = hgt_split_weights_f16 (both required) */
int hgt_edge_aggregate_update_f16x3(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                                    int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                                    const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows, void* hub_ws,
                                    int32_t* pending, const int64_t* node_type, const void* w_a_split, const float* b_a,
                                    const float* x_skip, int64_t ld_skip, const float* skip, const float* ln_w, const float* ln_b,
                                    int32_t use_norm, int32_t n_out, float* out, void* stream);

/* ABI 6: hgt_edge_aggregate_update for the targets [q_begin, q_end) only (q_begin a multiple of the plan tile, q_end <= n_q_rows):
 * one TARGET BLOCK of the multi-GPU path, whose in-edges only reference source rows that have already arrived.  `pending` must hold
 * (n_q_rows + 63) / 64 entries like in the whole-graph call: the kernels index it by ABSOLUTE 64-row tile (q_begin / 64 + workgroup),
 * so that target blocks of one layer may run concurrently.  Matrix-core kernel only (msg_frag required).  frag_f16: msg_frag / w_a_split are the fp16 images;
 * hub_deterministic: see HGT_FLAG_DETERMINISTIC_HUBS. */
int hgt_edge_aggregate_update_range(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations,
                                    int32_t n_heads, int32_t dk_pad, const float* logits, const float* V, const float* rte_v,
                                    const float* msg_p, const void* msg_frag, float* agg, int64_t n_q_rows, void* hub_ws,
                                    int32_t* pending, const int64_t* node_type, const void* w_a_split, const float* b_a,
                                    const float* x_skip, int64_t ld_skip, const float* skip, const float* ln_w, const float* ln_b,
                                    int32_t use_norm, int32_t n_out, float* out, void* stream, int64_t q_begin, int64_t q_end,
                                    int32_t frag_f16, int32_t hub_deterministic);

/* ----------------------------------------------------------------------------------------------
 * Backward pass (the reference gets it from autograd: OAG/train_paper_field.py:249,
 * ogbn-mag/train_ogbn_mag.py:172).  The chain rule on the node-level algebra of the forward needs three more edge
 * primitives, all of them "forward kernels in disguise" (pyhgt_amd/autograd.py strings them together):
 *   hgt_edge_spmm        out[i] = sum_rel ( sum_{e in (i,rel)} w_e (rows[src_e] + rte_rows[..]) ) F[rel]   -- the aggregation
 *                        kernel with GIVEN edge weights (sorted edge order of `plan`) instead of the softmax; F = f_p
 *                        (hgt_relation_pack layout, [R][H][dk_pad][dk_pad], out = in . F) and its MFMA image f_frag
 *                        (hgt_relation_frag_pack).  dQ = spmm(plan, ds, K, A'), dK = spmm(plan^T, ds, Q, A'^T),
 *                        dV = spmm(plan^T, att, dagg, M^T) where plan^T is the plan of the reversed edges.
 *   hgt_edge_logits      (existing) with (Q, K, att_t) := (dagg, V, M^T) yields d att_e = <dagg_i M^T, v_e>
 *   hgt_edge_softmax_bwd d s_e = att_e (d att_e - rho[dst_e]),  rho = hgt_head_dot(dagg, agg)
 *   hgt_relation_outer   out[r][h][k][c] += sum_{e of relation r} w_e a[src_e][h][k] b[dst_e][h][c]  (d relation_msg, d A')
 *   hgt_edge_gather_sorted  values given per ORIGINAL edge id -> sorted edge order of a plan (inverse of hgt_att_export)
 * and the dense pieces:
 *   hgt_node_update_bwd  reverse of hgt_node_update (+ the dropout mask of conv.py:125): d_trans, dx (skip path, overwritten),
 *                        d_alpha[t] += sum dy (o - x) (d skip = d_alpha * a (1 - a)), d_ln_w / d_ln_b [T][d] += ...
 *   hgt_gelu_bwd         out = dg * gelu'(agg)           hgt_mul_inplace   x *= m (dropout mask)
 *   hgt_typed_wgrad      out[g][m][n] += sum_{rows p of group g} A[rows[p]][m] B[rows[p]][n]   (exact fp32 MFMA; fp32 atomics)
 *   hgt_typed_colsum     out[g][c]    += sum_{rows p of group g} A[rows[p]][c]                  (bias gradients)
 * Accumulating outputs (+=) must be zeroed by the caller.
 * ---------------------------------------------------------------------------------------------- */
int hgt_edge_spmm(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, int32_t n_heads,
                  int32_t dk_pad, const float* weights, const float* rows, const float* rte_rows, const float* f_p,
                  const void* f_frag, float* out, int64_t ld_out, int64_t n_q_rows, void* hub_ws, void* stream);
/* ABI 7: the same sums on the item-parallel kernels of the latency regime (hgt_edge_aggregate_items with the edge weights given): one
 * wavefront per <= 16-edge work item + a fixed-order merge per target, no atomics; scratch = hgt_edge_aggregate_items_bytes().  What
 * pyhgt_amd/autograd.py takes below 65 536 nodes (the sub-tile kernel's wavefronts walk sixteen targets' edges one after the other:
 * 285 us per call on the transposed plan of a 3 200-node sampled batch against ~25 us).  HGT_ERR_UNSUPPORTED: > 63 relations, rows wider
 * than 512 padded columns, ld_out % 4 != 0, out not 16-byte aligned -> hgt_edge_spmm. */
int hgt_edge_spmm_items(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, int32_t n_heads,
                        int32_t dk_pad, const float* weights, const float* rows, const float* rte_rows, const void* f_frag, float* out,
                        int64_t ld_out, int64_t n_q_rows, void* scratch, uint64_t scratch_bytes, void* stream);
int hgt_edge_softmax_bwd(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, int32_t n_heads,
                         const float* att, const float* d_att, const float* rho, int64_t ld_rho, float* d_logits, void* stream);
int hgt_edge_gather_sorted(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, int32_t n_heads,
                           const float* by_edge_id, float* sorted, void* stream);
int hgt_head_dot(const float* a, const float* b, int64_t n_rows, int32_t n_heads, int32_t dk_pad, float* out, void* stream);
int hgt_relation_outer(const void* plan, int64_t n_nodes, int64_t n_edges, int32_t n_types, int32_t n_relations, int32_t n_heads,
                       int32_t dk_pad, const float* weights, const float* a_src, const float* rte_a, const float* b_dst, float* out,
                       void* stream);
int hgt_node_update_bwd(const float* grad_out, const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                        const float* skip, const float* ln_w, int32_t use_norm, const float* drop_mask, int64_t n_rows, int32_t d,
                        int32_t n_types, float* d_trans, float* dx, int64_t ld_dx, float* d_alpha, float* d_ln_w, float* d_ln_b,
                        void* stream);
/* the reverse of hgt_node_update_ex (DenseHGTConv.update, conv.py:250-274): skip == NULL = plain residual y = o + x (no gate;
 * d_alpha unused), shared_norm != 0 = one LayerNorm for every type (row 0 of ln_w / d_ln_w / d_ln_b) */
int hgt_node_update_bwd_ex(const float* grad_out, const float* trans, const float* x, int64_t ldx, const int64_t* node_type,
                           const float* skip, const float* ln_w, int32_t use_norm, int32_t shared_norm, const float* drop_mask,
                           int64_t n_rows, int32_t d, int32_t n_types, float* d_trans, float* dx, int64_t ld_dx, float* d_alpha,
                           float* d_ln_w, float* d_ln_b, void* stream);
/* off2[0..1] = {0, group_off[n_groups]}: the rows of every valid group as one group (shared dense layer of DenseHGTConv) */
int hgt_single_group_offsets(const int32_t* group_off, int32_t n_groups, int32_t* off2, void* stream);
int hgt_gelu_bwd(const float* dg, const float* agg, float* out, int64_t n, void* stream);
int hgt_mul_inplace(float* x, const float* m, int64_t n, void* stream);
int hgt_typed_wgrad(const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* rows, const int32_t* group_off,
                    int32_t n_groups, int64_t n_rows, int32_t m, int32_t n_cols, float* out, int64_t out_group_stride, void* stream);
/* hgt_typed_wgrad as 3-term split-bf16 products (relative error of a product ~3*2^-18, like the forward typed linears); colsum
 * (optional, [n_groups][colsum_group_stride]) += the column sums of A per group (the bias gradient) from the same pass. */
int hgt_typed_wgrad_bf16x3(const float* A, int64_t lda, const float* B, int64_t ldb, const int32_t* rows, const int32_t* group_off,
                           int32_t n_groups, int64_t n_rows, int32_t m, int32_t n_cols, float* out, int64_t out_group_stride,
                           float* colsum, int64_t colsum_group_stride, void* stream);
int hgt_typed_colsum(const float* A, int64_t lda, const int32_t* rows, const int32_t* group_off, int32_t n_groups, int64_t n_rows,
                     int32_t m, float* out, int64_t out_group_stride, void* stream);

/* ----------------------------------------------------------------------------------------------
 * One whole HGTConv.forward (conv.py:56-134, eval mode) as a single enqueue of the kernels above.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hgt_conv_args {
    /* problem */
    int64_t n_nodes, n_edges;
    int32_t in_dim, out_dim, n_types, n_relations, n_heads;
    int32_t use_norm, use_rte, precision, want_att;   /* precision: 0 exact fp32 MFMA chain, 1 split-bf16 x3, 2 fp16 hi/lo x3
                                                       * (every matrix-core product of the layer; stage 0 only: staged calls
                                                       * return HGT_ERR_UNSUPPORTED) */
    int64_t n_q_rows;            /* nodes [0, n_q_rows) get Q/aggregate/update; others only K,V (halo) */
    /* inputs */
    const float* x;              /* [n_nodes][in_dim]                          */
    const int64_t* node_type;    /* [n_nodes]                                  */
    const void* plan;            /* hgt_plan_build output                      */
    /* parameters (packed by the caller, see pyhgt_amd/conv.py::_pack_parameters) */
    const float* w_qkv;          /* [T][3*d_pad][in_dim] head-padded rows      */
    const float* b_qkv;          /* [T][3*d_pad]                               */
    const float* w_a;            /* [T][out_dim][d_pad] head-padded columns    */
    const float* b_a;            /* [T][out_dim]                               */
    const float* relation_att;   /* [R][H][d_k][d_k]                           */
    const float* relation_msg;   /* [R][H][d_k][d_k]                           */
    const float* relation_pri;   /* [R][H]                                     */
    const float* skip;           /* [T]                                        */
    const float* ln_w;           /* [T][out_dim] or NULL                       */
    const float* ln_b;           /* [T][out_dim] or NULL                       */
    const float* rte_emb;        /* [240][in_dim]   (use_rte)                  */
    const float* rte_w;          /* [in_dim][in_dim]                           */
    const float* rte_b;          /* [in_dim]                                   */
    /* workspace + outputs */
    void* workspace;             /* hgt_conv_workspace_bytes()                 */
    uint64_t workspace_bytes;
    float* out;                  /* [n_nodes][out_dim]                         */
    float* att_out;              /* [n_edges][n_heads] or NULL                 */
    /* optional instrumentation: HOST array of HGT_N_PHASE_EVENTS hipEvent_t handles, recorded on the
     * stream at the phase boundaries listed below (NULL = no events)           */
    void* const* phase_events;
    /* ABI 2: update_mode 0 = HGTConv.update (conv.py:114-134); 1 = DenseHGTConv.update (conv.py:250-274):
     *   y1 = LN_t(a_linear_t(agg) + x)   (no gelu on agg, no gate; `skip` is ignored)
     *   out = out_norm(out_linear(gelu(mid_linear(y1))) + y1)      (weights shared by all types)   */
    int32_t update_mode;
    const float* mid_w;          /* [2*out_dim][out_dim]                       */
    const float* mid_b;          /* [2*out_dim]                                */
    const float* out_w;          /* [out_dim][2*out_dim]                       */
    const float* out_b;          /* [out_dim]                                  */
    const float* out_ln_w;       /* [out_dim]                                  */
    const float* out_ln_b;       /* [out_dim]                                  */
    /* Staged execution for the multi-GPU path (pyhgt_amd/dist.py), so that the halo exchange overlaps the projections.
     * All stages of one forward use the same args (same workspace) on the same stream:
     *   0  the whole layer (default)
     *   1  parameter packing + Q|K|V projections of the rows [0, n_q_rows) only (+ temporal tables)
     *   2  K|V projections of the rows listed in proj_rows (typed row list: group t = proj_rows[proj_off[t] ..
     *      proj_off[t+1]), device arrays, n_types+1 offsets) -- call once per received chunk of halo rows
     *   3  edge phase + update (needs K,V of every source row: stages 1 and 2 done)
     *   4  (ABI 4) edge phase over slice `slice_index` of `slice_count` equal slices of the relation ids, softmax state carried
     *      from slice to slice in the workspace; the LAST slice also takes the unclaimed edges and runs the update.  For
     *      source-bucketed graphs: the caller numbers relations bucket * R' + relation (n_relations = slice_count * R', the
     *      relation parameters repeated slice_count times), so slice b = the edges whose source rows arrived with bucket b
     *      and the edge phase overlaps the exchange of the later buckets.  bf16x3 precision, HGTConv update only.
     *   5  (ABI 6) edge phase + fused node update of the TARGET BLOCK [q_begin, q_end) (q_begin a multiple of the plan tile; its
     *      logits work items are [item_begin, item_end), hgt_plan_tile_items_offset).  pyhgt_amd.dist orders a rank's halo rows by
     *      the first target block that needs them, so block b can run as soon as halo chunks 0..b have been projected (stage 2):
     *      no state is carried between blocks, every block is the single-GPU kernel pair on a tile range.  Split precisions, HGTConv
     *      update, padded row <= 256 columns (HGT_ERR_UNSUPPORTED otherwise: the caller falls back to stages 1/2/3).          */
    int32_t stage;
    const int32_t* proj_rows;
    const int32_t* proj_off;
    int64_t proj_n;              /* number of rows in proj_rows (host value)   */
    /* Weight-only preprocessing kept across calls (inference with fixed parameters: the reference re-derives nothing
     * per call either): packed relation matrices, split-bf16 tiles of W_qkv and W_a, temporal tables RTE_K / RTE_V.
     * `prepared` = caller-owned device buffer of hgt_conv_prepared_bytes() bytes that belongs to ONE parameter set;
     * prepared_valid = 0 -> this call fills it, 1 -> this call trusts it (the caller resets it to 0 whenever a
     * parameter, the precision or the buffer changed).  NULL = recompute into the workspace every call. */
    void* prepared;
    uint64_t prepared_bytes;
    int32_t prepared_valid;
    /* What the caller KNOWS about the plan (from the plan header, read back asynchronously -- pyhgt_amd.GraphPlan), as a bit set;
     * 0 = nothing known: every kernel is enqueued.
     *   bit 0: the plan found no hub target, so the hub kernels of the aggregation (4-5 launches that would exit at once)
     *          are not enqueued;
     *   bit 1: every target row has a valid node type, so hgt_zero_rows (output rows of unknown type := 0) is not enqueued. */
    int32_t plan_no_hubs;
    /* ABI 3: bit set of HGT_FLAG_* (0 = the default kernel selection) */
    int32_t flags;
    /* ABI 4: stage 4 only */
    int32_t slice_index;
    int32_t slice_count;
    /* ABI 6: stage 5 only */
    int64_t q_begin, q_end;
    int32_t item_begin, item_end;
    /* ABI 6: stage 2 only, optional: the rows of proj_rows are read from this buffer of 24-bit wire rows (hgt_gather_rows_c24 format,
     * 3 * in_dim bytes per row) instead of from x -- wire row = node row - proj_c24_row0 (the chunk's first local row).  Needs
     * in_dim <= 256, in_dim % 4 == 0 and a split precision (HGT_ERR_UNSUPPORTED otherwise -> hgt_unpack_rows_c24 + plain stage 2). */
    const void* proj_c24;
    int64_t proj_c24_row0;
} hgt_conv_args;

/* hgt_conv_args.flags: explicit kernel-selection switches (A/B measurements, tests); never read from the environment */
#define HGT_FLAG_NO_FUSED_UPDATE 1   /* aggregation writes agg, the node update runs as its own kernel(s) */
#define HGT_FLAG_VALU_AGGREGATE  2   /* relation transforms of the aggregation on the vector ALU (round-1 kernel) instead of MFMA */
#define HGT_FLAG_MFMA_LOGITS     4   /* hgt_edge_logits_mfma for every layout it covers (default: d_k >= 64 only) */
#define HGT_FLAG_VALU_LOGITS     8   /* never hgt_edge_logits_mfma */
#define HGT_FLAG_ITEM_AGGREGATE 16   /* hgt_edge_aggregate_items wherever it applies (the default below 65536 nodes when its scratch is at most 1 GB) */
#define HGT_FLAG_NO_ITEM_AGGREGATE 32 /* never hgt_edge_aggregate_items */
#define HGT_FLAG_FUSED_ANY_SIZE 64   /* hgt_edge_aggregate_update below its default size too (>= 16384 targets) */
#define HGT_FLAG_SINGLE_PASS 256     /* LAB builds only (ignored otherwise: the two-kernel form runs).  ABI 6: hgt_edge_single_pass_items instead of logits + item-parallel aggregation where it applies (sampled
                                      * sub-graphs, attention weights not exported).  Off by default: measured equal at c3 and 5 % slower at c5
                                      * (its two LDS tiles cap it at 4 wavefronts per CU) */
#define HGT_FLAG_XS_GEMM_ALWAYS 1024 /* ABI 6: the x-stationary split GEMM (hgt_gemm_xs.hip) for every typed linear of the layer it covers, whatever
                                      * the row count (HGT_LINEAR_FORCE_XS); */
#define HGT_FLAG_XS_GEMM_NEVER 2048  /* ... never (HGT_LINEAR_NO_XS): the slab kernels.  Bit-identical results either way: tests / A/B timings */
#define HGT_FLAG_NO_TILE_GEMM 4096   /* ABI 7: the typed linears of a small layer on the persistent / slab kernels instead of the latency-regime tile
                                      * kernel (HGT_LINEAR_NO_TILE): bit-identical results; tests / A/B timings */
#define HGT_FLAG_NO_MERGE_UPDATE 8192 /* ABI 7: sampled batches: hgt_edge_aggregate_items + hgt_linear_update_* as two calls instead of
                                      * hgt_edge_aggregate_items_update (identical output; tests / A/B timings) */
#define HGT_FLAG_NO_COOP_EDGE 16384  /* ABI 7: d_k >= 64 (logits) / >= 32 (runs): the one-wavefront-per-item forms of hgt_edge_logits_mfma / the runs kernel of
                                      * hgt_edge_aggregate_items instead of the forms that share the relation transform across the workgroup
                                      * (identical output bit for bit; tests / A/B timings).  The same switch is bit 1 of `frag_f16` of those calls */
#define HGT_FLAG_COOP_EDGE_ALWAYS 32768 /* ... the shared-transform forms for work items of more than 16 edges too (default: the 16-edge items
                                      * of sampled batches only; larger items measured slower in lock-step rounds).  Bit 2 of `frag_f16` */
#define HGT_FLAG_RING_AGGREGATE 512 /* LAB builds only (hgt_build_features() & HGT_FEATURE_LAB_KERNELS; ignored otherwise): the LDS-ring form of the fused
                                     * aggregation kernel (round 5, csrc/lab/hgt_edge_agg_ring.h: rows by LDS-DMA, U tile in registers) where it exists
                                     * (d = 256 / 8 heads, no temporal rows, bf16 split): bit-identical, measured 9 % slower at c2 */
#define HGT_FLAG_DETERMINISTIC_HUBS 128 /* ABI 6: targets with more than 1024 in-edges ("hubs") are aggregated WITHOUT atomics: every piece of a
                                      * (hub, relation) range writes its partial row / exp-sum to its own slot and the finalize kernel sums
                                      * the slots in (relation, piece) order, so two forwards are bit-identical on every row (the default
                                      * hub path adds fp32 partials atomically: hub rows then differ in the last bits from run to run).
                                      * Costs workspace: hgt_conv_workspace_bytes_ex(options bit 1) */

/* phase boundaries at which hgt_conv_forward records phase_events[i]:
 *   0 start | 1 relation pack + Q/K/V (+ temporal tables) done | 2 logits done | 3 softmax done |
 *   4 aggregate (+ att export) done | 5 a_linear done | 6 node update done                      */
#define HGT_N_PHASE_EVENTS 7

int hgt_conv_workspace_bytes(int64_t n_nodes, int64_t n_edges, int32_t in_dim, int32_t out_dim,
                             int32_t n_types, int32_t n_relations, int32_t n_heads, int32_t use_rte,
                             uint64_t* out_host);
/* ABI 6: options bit 0 = include the scratch of hgt_edge_aggregate_items (E*d*4 + E*H*8 + E bytes, up to 1 GiB, on graphs below 65536
 * nodes; calls that cannot take that kernel -- exact fp32, HGT_FLAG_NO_ITEM_AGGREGATE, staged multi-GPU calls -- leave it out, and
 * hgt_conv_forward accepts either size); bit 1 = the partial slots of HGT_FLAG_DETERMINISTIC_HUBS (required when that flag is set). */
int hgt_conv_workspace_bytes_ex(int64_t n_nodes, int64_t n_edges, int32_t in_dim, int32_t out_dim,
                                int32_t n_types, int32_t n_relations, int32_t n_heads, int32_t use_rte, int32_t options,
                                uint64_t* out_host);
int hgt_conv_prepared_bytes(int32_t in_dim, int32_t out_dim, int32_t n_types, int32_t n_relations, int32_t n_heads,
                            int32_t use_rte, uint64_t* out_host);
int hgt_conv_forward(const hgt_conv_args* args_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HGT_HIP_H_ */
