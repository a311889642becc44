// Probe: operand layout of v_mfma_f32_32x32x16_{f16,bf16} on gfx950 and whether f16 denormal inputs are honoured.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* A, const float* B, float* D, int mode) {
    // A [32][16] row-major (i,k), B [16][32] (k,n); assumed: lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31]
    const int l = threadIdx.x;
    f16v acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (mode == 0) {
        h8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = (_Float16)B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else {
        b8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = (__bf16)B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = acc[e];
    }
}

int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7 + 3) % 13 - 6) / 4.f; hB[i] = (float)((i * 5 + 1) % 11 - 5) / 8.f; }
        if (mode == 2) for (int i = 0; i < 512; ++i) hA[i] *= 1.0e-6f;   // f16 denormal range (min normal 6.1e-5)
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode == 1 ? 1 : 0);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
            double r = 0;
            for (int k = 0; k < 16; ++k) {
                float a = hA[i * 16 + k], b = hB[k * 32 + n];
                if (mode != 1) { a = (float)(_Float16)a; b = (float)(_Float16)b; }
                r += (double)a * b;
            }
            maxerr = fmax(maxerr, fabs(r - hD[i * 32 + n])); maxref = fmax(maxref, fabs(r));
        }
        printf("mode %d (%s): max err %.3g  max |ref| %.3g\n", mode, mode == 0 ? "f16 layout" : mode == 1 ? "bf16 layout" : "f16 denormal A", maxerr, maxref);
    }
    return 0;
}
