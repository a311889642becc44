#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16x8 __attribute__((aligned(2))) bf16x8_u;
__global__ void probe(const short* in, short* out, int shift, int iters, long long* clk) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    bf16x8 acc = {0,0,0,0,0,0,0,0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const short* p = lds + ((threadIdx.x * 8 + it * 64) & 4095) + shift;
        bf16x8 v = *reinterpret_cast<const bf16x8_u*>(p);
        acc += v;
    }
    long long t1 = clock64();
    *reinterpret_cast<bf16x8*>(out + threadIdx.x * 8) = acc;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    std::vector<short> h(8192); for (int i = 0; i < 8192; ++i) h[i] = (short)i;
    short *din, *dout; long long* dclk;
    hipMalloc(&din, 16384); hipMalloc(&dout, 256 * 16); hipMalloc(&dclk, 8);
    hipMemcpy(din, h.data(), 16384, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 9; ++shift) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, din, dout, shift, 1, dclk);
        std::vector<short> o(2048); hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int t = 0; t < 256; ++t) for (int e = 0; e < 8; ++e) if (o[t * 8 + e] != (short)(((t * 8) & 4095) + shift + e)) ok = false;
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, din, dout, shift, 4096, dclk);
        long long c; hipMemcpy(&c, dclk, 8, hipMemcpyDeviceToHost);
        printf("shift %d: %s, %.2f clk per wave-read\n", shift, ok ? "correct" : "WRONG", (double)c / 4096);
    }
    return 0;
}
