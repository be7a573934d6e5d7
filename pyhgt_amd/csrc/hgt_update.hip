// Node update epilogue (conv.py:129-133) and the halo row gather.
//
// One 64-lane wavefront per node row: gated skip connection  y = o*sigmoid(skip[t]) + x*(1-sigmoid(skip[t]))
// followed by the per-type LayerNorm (eps 1e-5, affine).  Rows whose type is outside [0,T) are
// written as zeros (the reference's zero-initialised `res`, conv.py:120).
#include "hgt_common.h"
#include "hgt_split_common.h"

namespace {

constexpr int MAX_PER_LANE = 16;   // d <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void k_node_update(const float* __restrict__ trans, const float* __restrict__ x, int64_t ldx,
                                                     const int64_t* __restrict__ ntype, const float* __restrict__ skip,
                                                     const float* __restrict__ lnw, const float* __restrict__ lnb, int use_norm,
                                                     int ln_shared, int64_t N, int d, int T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int64_t t = ntype[n];
    float* __restrict__ o = out + n * d;
    if (t < 0 || t >= T) {
        for (int c = lane; c < d; c += 64) o[c] = 0.0f;
        return;
    }
    // gated skip of HGTConv (conv.py:129-131) or, without a gate, the plain residual of DenseHGTConv (conv.py:259,271)
    const float a = skip ? 1.0f / (1.0f + expf(-skip[t])) : 1.0f;
    const float a1 = skip ? 1.0f - a : 1.0f;
    const float* __restrict__ tr = trans + n * d;
    const float* __restrict__ xr = x + n * ldx;
    float y[MAX_PER_LANE];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < MAX_PER_LANE; ++j) {
        const int c = lane + 64 * j;
        y[j] = 0.0f;
        if (c < d) {
            y[j] = tr[c] * a + xr[c] * a1;
            s += y[j];
        }
    }
    if (use_norm) {
        const float mean = wave_sum(s) / (float)d;
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            const int c = lane + 64 * j;
            if (c < d) { const float dlt = y[j] - mean; v += dlt * dlt; }
        }
        const float rstd = rsqrtf(wave_sum(v) / (float)d + 1e-5f);
        const float* __restrict__ w = lnw + (ln_shared ? 0 : t * d);
        const float* __restrict__ b = lnb + (ln_shared ? 0 : t * d);
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            const int c = lane + 64 * j;
            if (c < d) o[c] = (y[j] - mean) * rstd * w[c] + b[c];
        }
    } else {
#pragma unroll
        for (int j = 0; j < MAX_PER_LANE; ++j) {
            const int c = lane + 64 * j;
            if (c < d) o[c] = y[j];
        }
    }
}

__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ idx,
                                                     int64_t n, int d, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* __restrict__ src = x + (int64_t)idx[i] * ldx;
    float* __restrict__ dst = out + i * d;
    if ((d & 3) == 0 && (ldx & 3) == 0) {
        for (int c = lane * 4; c < d; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    } else {
        for (int c = lane; c < d; c += 64) dst[c] = src[c];
    }
}

// 24-bit transport format of halo rows (sign, 8 exponent bits, 15 mantissa bits, round to nearest; relative error <= 2^-16):
// 4 floats -> 3 dwords.  The multi-GPU exchange is bound by the xGMI links, so a quarter fewer bytes is a quarter less time;
// halo rows only feed the K/V projections, whose split-bf16 operands carry 16 mantissa bits anyway.
__device__ __forceinline__ unsigned f32_to_c24(float f) {
    return (__builtin_bit_cast(unsigned, f) + 0x80u) >> 8;
}
__device__ __forceinline__ float c24_to_f32(unsigned v) { return __builtin_bit_cast(float, v << 8); }

__global__ __launch_bounds__(256) void k_gather_rows_c24(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ idx,
                                                         int64_t n, int d, unsigned* __restrict__ out) {
    // one wavefront per row.  The 3 dwords a lane makes from its 4 floats go through a wave-private LDS row so that the row leaves
    // as 16-byte stores of consecutive lanes (round 4: three dword stores with a 12-byte lane stride ran at 3 TB/s of traffic; the
    // packing of 5 M halo rows is 2 - 3 ms of a multi-GPU step)
    __shared__ __attribute__((aligned(16))) unsigned s_row[4][3 * 64];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wib;
    if (i >= n) return;
    const float* __restrict__ src = x + (int64_t)idx[i] * ldx;
    unsigned* __restrict__ dst = out + i * (3 * (d / 4));
    for (int c0 = 0; c0 < d / 4; c0 += 64) {
        const int c = c0 + lane;
        if (c < d / 4) {
            const float4 f = *reinterpret_cast<const float4*>(src + 4 * c);
            const unsigned v0 = f32_to_c24(f.x), v1 = f32_to_c24(f.y), v2 = f32_to_c24(f.z), v3 = f32_to_c24(f.w);
            s_row[wib][3 * lane] = v0 | (v1 << 24);
            s_row[wib][3 * lane + 1] = (v1 >> 8) | (v2 << 16);
            s_row[wib][3 * lane + 2] = (v2 >> 16) | (v3 << 8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nd = 3 * min(64, d / 4 - c0);      // dwords of this pass (a multiple of 3; 192 for whole passes)
        unsigned* o = dst + 3 * c0;
        if (lane * 4 + 3 < nd && ((((uintptr_t)o) & 15) == 0)) {
            *reinterpret_cast<uint4*>(o + 4 * lane) = *reinterpret_cast<const uint4*>(&s_row[wib][4 * lane]);
        } else {
            for (int q = 4 * lane; q < min(4 * lane + 4, nd); ++q) o[q] = s_row[wib][q];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void k_unpack_rows_c24(const unsigned* __restrict__ in, int64_t n, int d, float* __restrict__ out,
                                                         int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const unsigned* __restrict__ src = in + i * (3 * (d / 4));
    float* __restrict__ dst = out + i * ld;
    for (int c = lane; c < d / 4; c += 64) {
        const unsigned w0 = src[3 * c], w1 = src[3 * c + 1], w2 = src[3 * c + 2];
        const unsigned v0 = w0 & 0xFFFFFFu, v1 = (w0 >> 24) | ((w1 & 0xFFFFu) << 8), v2 = (w1 >> 16) | ((w2 & 0xFFu) << 16), v3 = w2 >> 8;
        *reinterpret_cast<float4*>(dst + 4 * c) = make_float4(c24_to_f32(v0), c24_to_f32(v1), c24_to_f32(v2), c24_to_f32(v3));
    }
}

// out[rows[i]] = 0 for i in [off[0], off[1]) (device-side range): rows whose node type no group claims (conv.py:120)
__global__ __launch_bounds__(256) void k_zero_rows(const int32_t* __restrict__ rows, const int32_t* __restrict__ off, int d,
                                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int beg = off[0], end = off[1];
    for (int i = beg + blockIdx.x * 4 + (threadIdx.x >> 6); i < end; i += gridDim.x * 4) {
        float* o = out + (int64_t)rows[i] * d;
        for (int c = lane; c < d; c += 64) o[c] = 0.0f;
    }
}

__global__ void k_tanh_inplace(float* __restrict__ x, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<float4*>(x + i);
        v.x = hgt_tanh(v.x); v.y = hgt_tanh(v.y); v.z = hgt_tanh(v.z); v.w = hgt_tanh(v.w);
        *reinterpret_cast<float4*>(x + i) = v;
    } else {
        for (int64_t j = i; j < n; ++j) x[j] = hgt_tanh(x[j]);
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) v = fmaxf(v, __shfl_xor(v, sft));
    return v;
}

// out[r][c] = x[r][c] - max_r - log(sum_c exp(x[r][c] - max_r)); one wavefront per row, any width
__global__ __launch_bounds__(256) void k_log_softmax_rows(const float* __restrict__ x, int64_t n, int c, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* __restrict__ xr = x + r * c;
    float m = -3.0e38f;
    for (int j = lane; j < c; j += 64) m = fmaxf(m, xr[j]);
    m = wave_max(m);
    float s = 0.0f;
    for (int j = lane; j < c; j += 64) s += expf(xr[j] - m);
    const float lse = m + logf(wave_sum(s));
    float* __restrict__ o = out + r * c;
    for (int j = lane; j < c; j += 64) o[j] = xr[j] - lse;
}

// out[r] = scale * <x[r], y[r]>
__global__ __launch_bounds__(256) void k_row_dot(const float* __restrict__ x, const float* __restrict__ y, int64_t n, int d, float scale,
                                                 float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.0f;
    for (int j = lane; j < d; j += 64) s = fmaf(x[r * d + j], y[r * d + j], s);
    s = wave_sum(s);
    if (lane == 0) out[r] = s * scale;
}

}  // namespace

extern "C" int hgt_log_softmax_rows(const float* x, int64_t n_rows, int32_t n_cols, float* out, void* stream) {
    if (n_rows == 0) return HGT_OK;
    if (!x || !out || n_rows < 0 || n_cols <= 0) return HGT_ERR_INVALID_ARG;
    k_log_softmax_rows<<<(unsigned)((n_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, n_rows, n_cols, out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_row_dot(const float* x, const float* y, int64_t n_rows, int32_t d, float scale, float* out, void* stream) {
    if (n_rows == 0) return HGT_OK;
    if (!x || !y || !out || n_rows < 0 || d <= 0) return HGT_ERR_INVALID_ARG;
    k_row_dot<<<(unsigned)((n_rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, y, n_rows, d, scale, out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_tanh_inplace(float* x, int64_t n, void* stream) {
    if (n == 0) return HGT_OK;
    if (!x || n < 0 || ((uintptr_t)x & 15) != 0) return HGT_ERR_INVALID_ARG;
    k_tanh_inplace<<<(unsigned)((n / 4 + 256) / 256), 256, 0, (hipStream_t)stream>>>(x, n);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_zero_rows(const int32_t* rows, const int32_t* range, int32_t d, float* out, void* stream) {
    if (!rows || !range || !out || d <= 0) return HGT_ERR_INVALID_ARG;
    k_zero_rows<<<64, 256, 0, (hipStream_t)stream>>>(rows, range, d, out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_node_update_ex(const float* trans, const float* x, int64_t ldx, const int64_t* node_type, const float* skip,
                                  const float* ln_w, const float* ln_b, int32_t use_norm, int32_t ln_shared, int64_t n_nodes,
                                  int32_t d, int32_t n_types, float* out, void* stream) {
    if (!trans || !x || !node_type || !out || d <= 0 || n_nodes < 0 || (use_norm && (!ln_w || !ln_b))) return HGT_ERR_INVALID_ARG;
    if (d > 64 * MAX_PER_LANE) return HGT_ERR_UNSUPPORTED;
    if (n_nodes == 0) return HGT_OK;
    k_node_update<<<(unsigned)((n_nodes + 3) / 4), 256, 0, (hipStream_t)stream>>>(trans, x, ldx, node_type, skip, ln_w, ln_b, use_norm,
                                                                                 ln_shared, n_nodes, d, n_types, out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_node_update(const float* trans, const float* x, int64_t ldx, const int64_t* node_type, const float* skip,
                               const float* ln_w, const float* ln_b, int32_t use_norm, int64_t n_nodes, int32_t d,
                               int32_t n_types, float* out, void* stream) {
    if (!skip) return HGT_ERR_INVALID_ARG;
    return hgt_node_update_ex(trans, x, ldx, node_type, skip, ln_w, ln_b, use_norm, 0, n_nodes, d, n_types, out, stream);
}

extern "C" int hgt_gather_rows_c24(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t d, void* out, void* stream) {
    if (n == 0) return HGT_OK;
    if (!x || !idx || !out || d <= 0 || n < 0) return HGT_ERR_INVALID_ARG;
    if ((d & 3) != 0 || (ldx & 3) != 0 || ((uintptr_t)x & 15) != 0) return HGT_ERR_UNSUPPORTED;
    k_gather_rows_c24<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, ldx, idx, n, d, (unsigned*)out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_unpack_rows_c24(const void* in, int64_t n, int32_t d, float* out, int64_t ld_out, void* stream) {
    if (n == 0) return HGT_OK;
    if (!in || !out || d <= 0 || n < 0) return HGT_ERR_INVALID_ARG;
    if ((d & 3) != 0 || (ld_out & 3) != 0 || ((uintptr_t)out & 15) != 0) return HGT_ERR_UNSUPPORTED;
    k_unpack_rows_c24<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>((const unsigned*)in, n, d, out, ld_out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}

extern "C" int hgt_gather_rows(const float* x, int64_t ldx, const int32_t* idx, int64_t n, int32_t d, float* out, void* stream) {
    if (n == 0) return HGT_OK;
    if (!x || !idx || !out || d <= 0 || n < 0) return HGT_ERR_INVALID_ARG;
    k_gather_rows<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, ldx, idx, n, d, out);
    HGT_CHECK_LAUNCH();
    return HGT_OK;
}
