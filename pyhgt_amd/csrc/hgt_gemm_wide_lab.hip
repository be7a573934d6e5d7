// tools/lab only: dispatches hgt_launch_typed_linear_wide to one of the experiment builds of hgt_gemm_wide.hip
// (HGT_WD_VARIANT=n, see the Makefile target `lab`).  Never part of the product library.
#include "hgt_common.h"
#include <cstdlib>
#define DECL(N) int hgt_launch_typed_linear_wide_v##N(const float*, int64_t, const int32_t*, const int32_t*, int32_t, int64_t, int32_t, int32_t, \
    const void*, const float*, int64_t, float*, float*, float*, int32_t, int32_t, int, hipStream_t);
DECL(0) DECL(1) DECL(2) DECL(3) DECL(4) DECL(5) DECL(6) DECL(7) DECL(8) DECL(9) DECL(10) DECL(11) DECL(12) DECL(13)
int hgt_launch_typed_linear_wide(const float* x, int64_t ldx, const int32_t* rows, const int32_t* group_off, int32_t n_groups,
                                 int64_t n_rows, int32_t k, int32_t n_out, const void* w_split, const float* bias,
                                 int64_t b_group_stride, float* out0, float* out1, float* out2, int32_t block_cols,
                                 int32_t out_by_position, int n_cu, hipStream_t stream) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("HGT_WD_VARIANT"); v = e ? atoi(e) : 0; }
#define CALL(N) case N: return hgt_launch_typed_linear_wide_v##N(x, ldx, rows, group_off, n_groups, n_rows, k, n_out, w_split, bias, \
    b_group_stride, out0, out1, out2, block_cols, out_by_position, n_cu, stream);
    switch (v) { CALL(0) CALL(1) CALL(2) CALL(3) CALL(4) CALL(5) CALL(6) CALL(7) CALL(8) CALL(9) CALL(10) CALL(11) CALL(12) CALL(13) }
    return HGT_ERR_INVALID_ARG;
}
