// K0: weight-norm fold  w = g * v / ||v||  (torch.nn.utils.weight_norm, dim=0; reference call sites
// models.py:18-33,58-61,83,90-92,100) fused with the relayout into the [k][C_in][C_out] form the conv tiles read.
#include "v2w_common.h"

namespace {

// scale[row] = g[row] / ||v[row, :]||   (one block per row; fp64 accumulation; g == NULL -> 1)
__global__ void __launch_bounds__(256)
wn_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale, int inner) {
    __shared__ double red[16];
    const int row = blockIdx.x;
    const float* src = v + (size_t)row * inner;
    double acc = 0.0;
    for (int i = threadIdx.x; i < inner; i += 256) { const double x = (double)src[i]; acc += x * x; }
    const double n2 = v2w_block_sum(acc, red);
    if (threadIdx.x == 0) scale[row] = g ? (float)((double)g[row] / sqrt(n2)) : 1.f;
}

// conv: v (C_out, C_in*K) -> wf[(t*C_in + ci)*C_out + co]; 32x32 tiles through LDS so both sides are coalesced.
__global__ void __launch_bounds__(256)
relayout_conv_kernel(const float* __restrict__ v, const float* __restrict__ scale, float* __restrict__ wf,
                     int Cout, int Cin, int K) {
    __shared__ float tile[32][33];
    const int inner = Cin * K;
    const int j0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int y = ty; y < 32; y += 8) {
        const int co = co0 + y, j = j0 + tx;
        tile[y][tx] = (co < Cout && j < inner) ? v[(size_t)co * inner + j] * scale[co] : 0.f;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int j = j0 + y, co = co0 + tx;
        if (j < inner && co < Cout) {
            const int ci = j / K, t = j % K;
            wf[((size_t)t * Cin + ci) * Cout + co] = tile[tx][y];
        }
    }
}

// convT: v (C_in, C_out*K); one block per ci: norm over the row, then wf[(t*C_in + ci)*C_out + co] = s * v[ci][co][t]
__global__ void __launch_bounds__(256)
fold_convt_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ wf,
                  int Cin, int Cout, int K) {
    __shared__ double red[16];
    const int ci = blockIdx.x;
    const int inner = Cout * K;
    const float* src = v + (size_t)ci * inner;
    float s = 1.f;
    if (g) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < inner; i += 256) { const double x = (double)src[i]; acc += x * x; }
        const double n2 = v2w_block_sum(acc, red);
        s = (float)((double)g[ci] / sqrt(n2));
    }
    for (int idx = threadIdx.x; idx < inner; idx += 256) {
        const int t = idx / Cout, co = idx % Cout;
        wf[((size_t)t * Cin + ci) * Cout + co] = src[co * K + t] * s;
    }
}

}  // namespace

extern "C" int v2w_wn_fold_conv(const float* v, const float* g, float* wf, float* scratch,
                                int c_out, int c_in, int k, void* stream) {
    if (!v || !wf || !scratch || c_out <= 0 || c_in <= 0 || k <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int inner = c_in * k;
    hipLaunchKernelGGL(wn_scale_kernel, dim3(c_out), dim3(256), 0, st, v, g, scratch, inner);
    hipLaunchKernelGGL(relayout_conv_kernel, dim3((inner + 31) / 32, (c_out + 31) / 32), dim3(256), 0, st,
                       v, scratch, wf, c_out, c_in, k);
    return v2w_launch_status();
}

extern "C" int v2w_wn_fold_convt(const float* v, const float* g, float* wf, float* scratch,
                                 int c_in, int c_out, int k, void* stream) {
    (void)scratch;
    if (!v || !wf || c_out <= 0 || c_in <= 0 || k <= 0) return V2W_E_ARG;
    hipLaunchKernelGGL(fold_convt_kernel, dim3(c_in), dim3(256), 0, (hipStream_t)stream, v, g, wf, c_in, c_out, k);
    return v2w_launch_status();
}
