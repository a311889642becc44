// ds_read_b128 throughput of the 16-channel stage's operand fetch under three LDS layouts (lane = j + 16 kg: position column j, K group kg):
//   0: rows of 32 bytes (one position, 16 channels), lane reads half kg & 1 of row j (+ tap)           - the kernel as shipped
//   1: two planes of 16-byte rows (channels 0-7 | 8-15), lane reads row j of plane kg & 1
//   2: rows of 32 bytes at a 48-byte stride
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, int iters, long long* clk, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kg = lane >> 4;
    unsigned addr;
    if (mode == 0) addr = (128 * wave + j + (kg >> 1) * 3) * 32 + (kg & 1) * 16;
    else if (mode == 1) addr = (kg & 1) * 16384 + (128 * wave + j + (kg >> 1) * 3) * 16;
    else addr = (128 * wave + j + (kg >> 1) * 3) * 48 + (kg & 1) * 16;
    addr += (unsigned)reinterpret_cast<uintptr_t>(lds);
    u32x4 acc = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4 v0, v1, v2, v3;
        const unsigned a = addr + (it & 7) * (mode == 1 ? 16 : (mode == 2 ? 48 : 32));
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\tds_read_b128 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
        acc += v0 + v1 + v2 + v3;
    }
    long long t1 = clock64();
    if (lane == 0) clk[wave] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    long long* dclk; unsigned* sink;
    hipMalloc(&dclk, 64); hipMalloc(&sink, 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 65536, 0, mode, 2048, dclk, sink);
        long long c[4]; hipMemcpy(c, dclk, 32, hipMemcpyDeviceToHost);
        printf("layout %d: %.1f clk per ds_read_b128 of the workgroup's 4 waves (wave 0: %.1f per own read)\n", mode, (double)c[0] / 2048 / 4 , (double)c[0] / 2048 / 4);
    }
    return 0;
}
