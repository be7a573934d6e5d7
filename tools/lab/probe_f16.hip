// Hardware probe for an fp16 hi/lo operand split on the matrix cores (tools/lab, not part of the library):
//  (1) does v_mfma_f32_32x32x16_f16 honour SUBNORMAL fp16 inputs, or flush them to zero?
//  (2) is the fp32 -> fp16 conversion the compiler emits round-to-nearest-even?
//  (3) a 32x32x16 product of split operands (lo*hi + hi*lo + hi*hi) against double.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_probe(const float* A, const float* B, float* C, int mode) {
    // A [32][16], B [16][32] (row major); lane l: row/col = l & 31, k = (l >> 5) * 8 .. + 8
    const int l = threadIdx.x, r = l & 31, k0 = (l >> 5) * 8;
    f16x8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
        const float a = A[r * 16 + k0 + e], b = B[(k0 + e) * 32 + r];
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;
        ah[e] = ha; al[e] = (_Float16)(a - (float)ha);
        bh[e] = hb; bl[e] = (_Float16)(b - (float)hb);
    }
    f32x16 acc = {0};
    if (mode == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
        C[row * 32 + col] = acc[i];
    }
}

__global__ void k_cvt(const float* x, unsigned short* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { _Float16 h = (_Float16)x[i]; out[i] = __builtin_bit_cast(unsigned short, h); }
}

int main() {
    float *dA, *dB, *dC; unsigned short* dH;
    hipMalloc(&dA, 32 * 16 * 4); hipMalloc(&dB, 16 * 32 * 4); hipMalloc(&dC, 32 * 32 * 4); hipMalloc(&dH, 64 * 2);
    std::vector<float> A(32 * 16), B(16 * 32), C(32 * 32);
    // (1) subnormal inputs: A = 2^-20 (fp16 subnormal: min normal 2^-14), B = 2^10 -> product 2^-10 per k, sum over 16 = 2^-6
    for (auto& v : A) v = ldexpf(1.0f, -20);
    for (auto& v : B) v = 1024.0f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_probe<<<1, 64>>>(dA, dB, dC, 0);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    printf("(1) subnormal fp16 input 2^-20 x 2^10, k=16: got %g, expected %g  -> %s\n", C[0], ldexp(1.0, -6),
           C[0] == (float)ldexp(1.0, -6) ? "subnormals HONOURED" : (C[0] == 0.0f ? "FLUSHED to zero" : "other"));
    // (2) rounding of the conversion: 1 + 2^-11 (tie -> even = 1.0), 1 + 3*2^-11 (tie -> 1 + 2^-9), 1 + 2^-11 + 2^-20 (above tie -> 1 + 2^-10)
    std::vector<float> X = {1.0f + ldexpf(1.f, -11), 1.0f + 3 * ldexpf(1.f, -11), 1.0f + ldexpf(1.f, -11) + ldexpf(1.f, -20), 65519.0f, 65520.0f, 1e-8f};
    std::vector<unsigned short> Hh(X.size());
    hipMemcpy(dA, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(dA, dH, (int)X.size());
    hipMemcpy(Hh.data(), dH, X.size() * 2, hipMemcpyDeviceToHost);
    printf("(2) cvt: %04x %04x %04x %04x %04x %04x (RNE expects 3c00 3c02 3c01 7bff 7c00 0000)\n", Hh[0], Hh[1], Hh[2], Hh[3], Hh[4], Hh[5]);
    // (3) split product against double
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 8.0f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        k_probe<<<1, 64>>>(dA, dB, dC, mode);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < 16; ++k) { s += (double)A[i * 16 + k] * B[k * 32 + j]; sa += fabs((double)A[i * 16 + k] * B[k * 32 + j]); }
            worst = fmax(worst, fabs(C[i * 32 + j] - s)); scale = fmax(scale, sa);
        }
        printf("(3) %s: max |err| %.3e (sum |terms| up to %.2f -> relative %.2e)\n", mode ? "lo*hi + hi*lo + hi*hi" : "hi*hi only", worst, scale, worst / scale);
    }
    return 0;
}
