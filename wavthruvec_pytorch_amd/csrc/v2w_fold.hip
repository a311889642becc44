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

// out[m][co][ci] = wf[t_start + m*t_step][ci][co], m < n: the weights of the conv that back-propagates through a forward
// conv (t_start = K-1, t_step = -1: flipped taps) or through one phase of a transposed conv (t_start = t0_r, t_step = u)
__global__ void __launch_bounds__(256)
transpose_flip_kernel(const float* __restrict__ wf, float* __restrict__ out, int t_start, int t_step, int Cin, int Cout) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = wf + (size_t)(t_start + t * t_step) * Cin * Cout;
    for (int y = ty; y < 32; y += 8) {
        const int ci = ci0 + y, co = co0 + tx;
        tile[y][tx] = (ci < Cin && co < Cout) ? src[(size_t)ci * Cout + co] : 0.f;
    }
    __syncthreads();
    float* dst = out + (size_t)t * Cout * Cin;
    for (int y = ty; y < 32; y += 8) {
        const int co = co0 + y, ci = ci0 + tx;
        if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = tile[tx][y];
    }
}

}  // namespace

extern "C" int v2w_wf_transpose_flip(const float* wf, float* out, int k, int c_in, int c_out, void* stream) {
    if (!wf || !out || k <= 0 || c_in <= 0 || c_out <= 0) return V2W_E_ARG;
    V2W_LAUNCH(transpose_flip_kernel, dim3((c_out + 31) / 32, (c_in + 31) / 32, k), dim3(256), 0, (hipStream_t)stream,
                       wf, out, k - 1, -1, c_in, c_out);
    return v2w_launch_status();
}

extern "C" int v2w_wf_gather_transpose(const float* wf, float* out, int k, int c_in, int c_out, int t_start, int t_step, int n,
                                       void* stream) {
    if (!wf || !out || k <= 0 || c_in <= 0 || c_out <= 0 || n <= 0) return V2W_E_ARG;
    if (t_start < 0 || t_start >= k || t_start + (n - 1) * t_step < 0 || t_start + (n - 1) * t_step >= k) return V2W_E_ARG;
    V2W_LAUNCH(transpose_flip_kernel, dim3((c_out + 31) / 32, (c_in + 31) / 32, n), dim3(256), 0, (hipStream_t)stream,
                       wf, out, t_start, t_step, c_in, c_out);
    return v2w_launch_status();
}

extern "C" int v2w_wn_fold_conv(const float* v, const float* g, float* wf, float* scratch,
                                int c_out, int c_in, int k, void* stream) {
    if (!v || !wf || !scratch || c_out <= 0 || c_in <= 0 || k <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int inner = c_in * k;
    V2W_LAUNCH(wn_scale_kernel, dim3(c_out), dim3(256), 0, st, v, g, scratch, inner);
    V2W_LAUNCH(relayout_conv_kernel, dim3((inner + 31) / 32, (c_out + 31) / 32), dim3(256), 0, st,
                       v, scratch, wf, c_out, c_in, k);
    return v2w_launch_status();
}

extern "C" int v2w_wn_fold_convt(const float* v, const float* g, float* wf, float* scratch,
                                 int c_in, int c_out, int k, void* stream) {
    (void)scratch;
    if (!v || !wf || c_out <= 0 || c_in <= 0 || k <= 0) return V2W_E_ARG;
    V2W_LAUNCH(fold_convt_kernel, dim3(c_in), dim3(256), 0, (hipStream_t)stream, v, g, wf, c_in, c_out, k);
    return v2w_launch_status();
}
