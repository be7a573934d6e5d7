// tools/lab/gather_ceiling.hip -- what a kernel that does NOTHING but gather random rows reaches on this part (the access pattern of the
// logits and aggregation kernels: one wavefront instruction = one contiguous row of ROWB bytes, rows picked by a random index list).
//   hipcc --offload-arch=gfx950 -O3 tools/lab/gather_ceiling.hip -o tools/lab/gather_ceiling.bin && tools/lab/gather_ceiling.bin
// Prints TB/s for 1 KB and 2 KB rows, 4..16 rows in flight per wavefront, 2 and 4 wavefronts per SIMD, random and sorted index lists.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VEC, int DEPTH>      // lane = VEC floats of a row (row = 64 * VEC * 4 bytes); DEPTH rows in flight per wavefront
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, const int* __restrict__ idx, long n_idx, int per_wave,
                                                float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long beg = wave * per_wave;
    if (beg >= n_idx) return;
    const int cnt = (int)min((long)per_wave, n_idx - beg);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;
    for (int base = 0; base < cnt; base += 64) {
        const int my = idx[beg + min(base + lane, cnt - 1)];
        const int nb = min(64, cnt - base);
        for (int i0 = 0; i0 < nb; i0 += DEPTH) {
            float v[DEPTH][VEC];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int r = __builtin_amdgcn_readlane(my, min(i0 + u, 63));
                const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(tab) + (unsigned)((unsigned)r * (unsigned)(64 * VEC * 4) + (unsigned)(lane * VEC * 4)));
                if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
                else { const float4 t0 = *reinterpret_cast<const float4*>(p), t1 = *reinterpret_cast<const float4*>(p + 4);
                       v[u][0] = t0.x; v[u][1] = t0.y; v[u][2] = t0.z; v[u][3] = t0.w; v[u][4] = t1.x; v[u][5] = t1.y; v[u][6] = t1.z; v[u][7] = t1.w; }
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] += v[u][i];
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += acc[i];
    out[wave * 64 + lane] = s;
}

template <int VEC, int DEPTH>
static void run(const char* what, const float* tab, const int* idx, long n_idx, int per_wave, float* out) {
    const long waves = (n_idx + per_wave - 1) / per_wave;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k_gather<VEC, DEPTH><<<grid, 256>>>(tab, idx, n_idx, per_wave, out);
    CK(hipEventRecord(a));
    const int it = 10;
    for (int i = 0; i < it; ++i) k_gather<VEC, DEPTH><<<grid, 256>>>(tab, idx, n_idx, per_wave, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= it;
    const double bytes = (double)n_idx * 64 * VEC * 4;
    printf("%-28s row %4d B depth %2d per_wave %4d: %.3f ms = %.2f TB/s\n", what, 64 * VEC * 4, DEPTH, per_wave, ms, bytes / ms / 1e9);
}

int main() {
    const long N = 1000000, E = 10000000;
    float* tab; int *idx_r, *idx_s; float* out;
    CK(hipMalloc(&tab, (size_t)N * 2048)); CK(hipMemset(tab, 0, (size_t)N * 2048));
    CK(hipMalloc(&idx_r, E * 4)); CK(hipMalloc(&idx_s, E * 4)); CK(hipMalloc(&out, (size_t)(E / 16 + 1024) * 64 * 4));
    std::vector<int> h(E);
    std::mt19937 g(1);
    for (long i = 0; i < E; ++i) h[i] = (int)(g() % N);
    CK(hipMemcpy(idx_r, h.data(), E * 4, hipMemcpyHostToDevice));
    std::sort(h.begin(), h.end());
    CK(hipMemcpy(idx_s, h.data(), E * 4, hipMemcpyHostToDevice));
    run<4, 4>("random", tab, idx_r, E, 160, out);
    run<4, 8>("random", tab, idx_r, E, 160, out);
    run<4, 16>("random", tab, idx_r, E, 160, out);
    run<4, 8>("random (long items)", tab, idx_r, E, 640, out);
    run<4, 16>("random (long items)", tab, idx_r, E, 640, out);
    run<8, 4>("random", tab, idx_r, E / 2, 160, out);
    run<8, 8>("random", tab, idx_r, E / 2, 160, out);
    run<4, 8>("sorted (each row ~10x)", tab, idx_s, E, 160, out);
    run<4, 16>("sorted (each row ~10x)", tab, idx_s, E, 160, out);
    return 0;
}
