#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(const float4* __restrict__ src, float4* dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    // each wave copies 1 KiB: lane l -> smem[wave*1024 + l*16]
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + tid),
                                     (void __attribute__((address_space(3)))*)(smem + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[tid] = *reinterpret_cast<float4*>(smem + ((tid + 1) % 256) * 16);
}
int main() {
    float4 *s, *d; hipMalloc(&s, 4096); hipMalloc(&d, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, s, d);
    float o[1024]; hipMemcpy(o, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int t = 0; t < 256; ++t) for (int e = 0; e < 4; ++e) if (o[t * 4 + e] != h[((t + 1) % 256) * 4 + e]) ++bad;
    printf("global_load_lds b128: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    return 0;
}
