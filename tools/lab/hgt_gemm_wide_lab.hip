// tools/lab only: dispatches hgt_launch_typed_linear_wide to one of the experiment builds of hgt_gemm_wide.hip
// (HGT_WD_VARIANT=n, tools/lab/build_lab.sh).  Never part of the product library.
#include "hgt_common.h"   // -I pyhgt_amd/csrc, -DHGT_LAB_WIDE (tools/lab/build_lab.sh)
#include <cstdlib>
#define DECL(N) int hgt_launch_typed_linear_wide_v##N(const float*, int64_t, const int32_t*, const int32_t*, int32_t, int64_t, int32_t, int32_t, \
    const void*, const float*, int64_t, float*, float*, float*, int32_t, int32_t, int, hipStream_t);
WD_DECLS
int hgt_launch_typed_linear_wide(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                 int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                 int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                 int32_t out_by_position, int n_cu, hipStream_t stream) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("HGT_WD_VARIANT"); v = e ? atoi(e) : 14; }
#define CALL(N) case N: return hgt_launch_typed_linear_wide_v##N(x, ldx, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, \
    b_group_stride, out0, out1, out2, block_cols, out_by_position, n_cu, stream);
    switch (v) { WD_CALLS }
    return HGT_ERR_INVALID_ARG;
}
