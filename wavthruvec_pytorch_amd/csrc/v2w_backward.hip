// Small backward kernels of the generator path.  The heavy parts of the backward pass reuse the forward
// tile kernel (dgrad, v2w_conv_mfma.hip) and the MFMA weight-gradient kernel (v2w_wgrad.hip); this file holds the reductions
// and the parameter-space chain rules: Conditional BatchNorm, tanh / conv_post, weight norm, spectral-norm Linear and fcs.
#include "v2w_common.h"

#define V2W_TAIL_PARTS 512     // partial rows of v2w_tail_bwd's part_ws (ABI v32: C_in * k * 512 doubles)

namespace {

// ---- CondBN backward, step 1: per (b, c) row  S1 = sum_l dx, S2 = sum_l dx * xr      (one block per row)
__global__ void __launch_bounds__(256)
cbn_bwd_rowsums_kernel(const float* __restrict__ dx, const float* __restrict__ xr, float* __restrict__ s12, int L) {
    __shared__ double red[16];
    const int row = blockIdx.x;
    const float* d = dx + (size_t)row * L;
    const float* x = xr + (size_t)row * L;
    double a = 0.0, b = 0.0;
    float fa = 0.f, fb = 0.f;
    int n = 0;
    for (int l = threadIdx.x; l < L; l += 256) {
        const float dv = d[l];
        fa += dv; fb = fmaf(dv, x[l], fb);
        if (++n == 64) { a += fa; b += fb; fa = fb = 0.f; n = 0; }   // bound the fp32 chains
    }
    a += fa; b += fb;
    const double t1 = v2w_block_sum(a, red), t2 = v2w_block_sum(b, red);
    if (threadIdx.x == 0) { s12[2 * row] = (float)t1; s12[2 * row + 1] = (float)t2; }
}

// ---- step 2: dgb (B, 2C) = [dgamma | dbeta] and the per-channel sums  csum[c] = sum_b gamma*dbeta, csum[C+c] = sum_b gamma*dgamma
// (the two numbers a data-parallel run all-reduces); one thread per channel.
//   x = gamma*xhat + beta, xhat = (xr - mean)*rstd :  dbeta = S1, dgamma = rstd*(S2 - mean*S1)
__global__ void cbn_bwd_sums_kernel(const float* __restrict__ s12, const float* __restrict__ gb, const double* __restrict__ stats,
                                    const float* __restrict__ rmean, const float* __restrict__ rvar, float* __restrict__ dgb,
                                    double* __restrict__ csum, int B, int C, int training, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) { const double n = stats[2 * C]; mean = stats[c] / n; var = stats[C + c] / n - mean * mean; if (var < 0) var = 0; }
    else { mean = rmean[c]; var = rvar[c]; }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    double g1 = 0.0, g2 = 0.0;
    for (int b = 0; b < B; ++b) {
        const double S1 = s12[2 * (b * C + c)], S2 = s12[2 * (b * C + c) + 1];
        const double dbeta = S1, dgamma = rstd * (S2 - mean * S1);
        const double gamma = gb[(size_t)b * 2 * C + c];
        dgb[(size_t)b * 2 * C + c] = (float)dgamma;
        dgb[(size_t)b * 2 * C + C + c] = (float)dbeta;
        g1 += gamma * dbeta; g2 += gamma * dgamma;
    }
    csum[c] = g1; csum[C + c] = g2;
}

// ---- step 3: dxr = A[b,c]*dx + Bc[c]*xr + Cc[c]:  A = gamma*rstd ; train: Bc = -rstd^2*m2, Cc = -rstd*m1 + rstd^2*mean*m2,
// m1 = csum[c]/n, m2 = csum[C+c]/n (n = global element count) ; eval: Bc = Cc = 0.   One thread per (b, c).
__global__ void cbn_bwd_tables_kernel(const float* __restrict__ gb, const double* __restrict__ stats, const double* __restrict__ csum,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                      float* __restrict__ A, float* __restrict__ Bc, float* __restrict__ Cc,
                                      int B, int C, int training, float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    double mean, var, n = 1.0;
    if (training) { n = stats[2 * C]; mean = stats[c] / n; var = stats[C + c] / n - mean * mean; if (var < 0) var = 0; }
    else { mean = rmean[c]; var = rvar[c]; }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    A[idx] = (float)((double)gb[(size_t)b * 2 * C + c] * rstd);
    if (b == 0) {
        if (training) {
            const double m1 = csum[c] / n, m2 = csum[C + c] / n;
            Bc[c] = (float)(-rstd * rstd * m2);
            Cc[c] = (float)(-rstd * m1 + rstd * rstd * mean * m2);
        } else { Bc[c] = 0.f; Cc[c] = 0.f; }
    }
}

__global__ void __launch_bounds__(256)
affine3_kernel(const float* __restrict__ dx, const float* __restrict__ xr, const float* __restrict__ A, const float* __restrict__ Bc,
               const float* __restrict__ Cc, float* __restrict__ out, int C, int L) {
    const int row = blockIdx.y;   // b*C + c
    const float a = A[row], bc = Bc[row % C], cc = Cc[row % C];
    const float* d = dx + (size_t)row * L;
    const float* x = xr + (size_t)row * L;
    float* o = out + (size_t)row * L;
    for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) o[l] = fmaf(a, d[l], fmaf(bc, x[l], cc));
}

// ---- tail backward: dp = dy * (1 - y^2)
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dp, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dp[i] = dy[i] * (1.f - y[i] * y[i]);
}

// conv_post dgrad: dx[b,ci,l] = lrelu'(x[b,ci,l]) * sum_t w[t][ci] * dp[b, l - (t - (k-1)/2)]
__global__ void __launch_bounds__(256)
conv_post_dgrad_kernel(const float* __restrict__ dp, const float* __restrict__ wf, const float* __restrict__ x, float* __restrict__ dx,
                       int Cin, int L, int k, float slope) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    const int ci = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const int pad = (k - 1) / 2;
    const float* d = dp + (size_t)b * L;
    float acc = 0.f;
    for (int t = 0; t < k; ++t) {
        const int li = l - (t - pad);
        if (li >= 0 && li < L) acc = fmaf(wf[t * Cin + ci], d[li], acc);
    }
    const size_t o = ((size_t)b * Cin + ci) * L + l;
    dx[o] = x[o] > 0.f ? acc : acc * slope;
}

// conv_post wgrad partials: part[(ci*k + t)*nsplit + s] = sum over a slice of (b, l) of lrelu(x)[b,ci,l+(t-pad)] * dp[b,l]
__global__ void __launch_bounds__(256)
conv_post_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dp, double* __restrict__ part,
                       int B, int Cin, int L, int k, float slope, int nsplit) {
    __shared__ double red[16];
    const int ci = blockIdx.y, s = blockIdx.x;
    const int pad = (k - 1) / 2;
    const int per = (L + nsplit - 1) / nsplit, lo = s * per, hi = min(L, lo + per);
    for (int t = 0; t < k; ++t) {
        double acc = 0.0;
        for (int b = 0; b < B; ++b) {
            const float* xs = x + ((size_t)b * Cin + ci) * L;
            const float* d = dp + (size_t)b * L;
            float fa = 0.f;
            for (int l = lo + threadIdx.x; l < hi; l += 256) {
                const int li = l + t - pad;
                if (li >= 0 && li < L) fa = fmaf(v2w_lrelu(xs[li], slope), d[l], fa);
            }
            acc += fa;
        }
        const double tot = v2w_block_sum(acc, red);
        if (threadIdx.x == 0) part[((size_t)ci * k + t) * nsplit + s] = tot;
    }
}
// conv_post input AND weight gradient in one pass over x (round 5; the two kernels above read x once for the mask and once per tap - 0.28 + 0.48 ms at
// B = 32 x 81 920 samples, 13 x the time the 168 MB of x and the 168 MB of dx take).  Persistent workgroups walk tiles of TP positions of one
// batch item: x [C][TP + 8] and dp [TP + 8] staged in LDS (zeros outside the sequence), a thread owns 4 consecutive positions - per channel
// three float4 LDS reads give the 12 values both gradients need - stores dx as float4 and keeps dW[c][t] of its positions in C * K
// registers over all its tiles; one block reduction at the end, partials in the layout conv_post_wgrad_reduce_kernel adds up (fp64, fixed order).
template <int C, int K>
__global__ void __launch_bounds__(256)
tail_bwd_fused_kernel(const float* __restrict__ dp, const float* __restrict__ wf, const float* __restrict__ x, float* __restrict__ dx,
                      double* __restrict__ part, int L, float slope, int ntl, int ntiles) {
    constexpr int TP = 1024, XW = TP + 8, PAD = (K - 1) / 2;
    static_assert(PAD <= 4 && K <= 9, "the 12-value window of a position quad covers 4 positions to either side");
    extern __shared__ __attribute__((aligned(16))) float smem_t[];
    float* const xs = smem_t;                      // [C][XW]: positions l0 - 4 .. l0 + TP + 3
    float* const dps = xs + C * XW;                // [XW]
    float* const wl = dps + XW;                    // [K][C]
    float* const red = wl + K * C;                 // [4][C * K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K * C; i += 256) wl[i] = wf[i];
    float acc[C][K];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < K; ++t) acc[c][t] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / ntl, l0 = (tile - b * ntl) * TP;
        __syncthreads();                           // the previous tile's reads are done (and wl is written)
        for (int i = tid; i < (C + 1) * (XW / 4); i += 256) {
            const int row = i / (XW / 4), c4 = i - row * (XW / 4);
            const int pos = l0 - 4 + 4 * c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos >= 0 && pos < L) v = *reinterpret_cast<const f32x4*>(row < C ? x + ((size_t)b * C + row) * L + pos : dp + (size_t)b * L + pos);
            *reinterpret_cast<f32x4*>((row < C ? xs + row * XW : dps) + 4 * c4) = v;
        }
        __syncthreads();
        const int l = l0 + 4 * tid;
        if (l < L) {
            float dw[12];                          // dp at positions l - 4 .. l + 7
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(dps + 4 * tid + 4 * u);
                dw[4 * u] = q[0]; dw[4 * u + 1] = q[1]; dw[4 * u + 2] = q[2]; dw[4 * u + 3] = q[3];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float xw[12], xa[12];              // x and lrelu(x) at positions l - 4 .. l + 7
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(xs + c * XW + 4 * tid + 4 * u);
                    xw[4 * u] = q[0]; xw[4 * u + 1] = q[1]; xw[4 * u + 2] = q[2]; xw[4 * u + 3] = q[3];
                }
#pragma unroll
                for (int j = 0; j < 12; ++j) xa[j] = v2w_lrelu(xw[j], slope);
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) a = fmaf(wl[t * C + c], dw[4 + i - (t - PAD)], a);       // dp[l + i - (t - PAD)]
                    o[i] = xw[4 + i] > 0.f ? a : a * slope;
                }
                *reinterpret_cast<f32x4*>(dx + ((size_t)b * C + c) * L + l) = o;
#pragma unroll
                for (int t = 0; t < K; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[c][t] = fmaf(xa[4 + i + t - PAD], dw[4 + i], acc[c][t]);   // lrelu(x)[l + i + t - PAD] * dp[l + i]
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float sum = v2w_wave_sum(acc[c][t]);
            if (lane == 0) red[wave * C * K + c * K + t] = sum;
        }
    __syncthreads();
    for (int i = tid; i < C * K; i += 256)
        part[(size_t)i * gridDim.x + blockIdx.x] = ((double)red[i] + (double)red[C * K + i]) + ((double)red[2 * C * K + i] + (double)red[3 * C * K + i]);
}

__global__ void conv_post_wgrad_reduce_kernel(const double* __restrict__ part, float* __restrict__ dwf, int Cin, int k, int nsplit) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // ci*k + t
    if (idx >= Cin * k) return;
    double v = 0.0;
    for (int s = 0; s < nsplit; ++s) v += part[(size_t)idx * nsplit + s];
    const int ci = idx / k, t = idx - ci * k;
    dwf[t * Cin + ci] = (float)v;   // [k][Cin][1]
}

// ---- weight-norm backward: w = g*v/||v|| (norm over all dims but 0), dw given in the [k][C_in][C_out] layout.
//   dg[row] = <dw,v>_row / ||v|| ;  dv = (g/||v||) * (dw - v * <dw,v>_row / ||v||^2)        one block per row
__global__ void __launch_bounds__(256)
wn_bwd_kernel(const float* __restrict__ dwf, const float* __restrict__ v, const float* __restrict__ g,
              float* __restrict__ dv, float* __restrict__ dg, int Cin, int Cout, int K, int transposed) {
    __shared__ double red[16];
    const int row = blockIdx.x;                       // co (conv) or ci (transposed)
    const int inner = (transposed ? Cout : Cin) * K;
    const float* vr = v + (size_t)row * inner;
    auto dw_at = [&](int i) {                          // element i of the row in v's layout -> dwf index
        const int o = i / K, t = i - o * K;           // o = ci (conv) / co (transposed)
        return transposed ? dwf[((size_t)t * Cin + row) * Cout + o] : dwf[((size_t)t * Cin + o) * Cout + row];
    };
    double dot = 0.0, n2 = 0.0;
    for (int i = threadIdx.x; i < inner; i += 256) { const double x = vr[i]; dot += x * (double)dw_at(i); n2 += x * x; }
    dot = v2w_block_sum(dot, red);
    n2 = v2w_block_sum(n2, red);
    const double norm = sqrt(n2);
    if (!g) {                                          // weight norm removed: the parameter IS the weight
        for (int i = threadIdx.x; i < inner; i += 256) dv[(size_t)row * inner + i] = dw_at(i);
        return;
    }
    const double gs = (double)g[row] / norm, k2 = dot / n2;
    if (threadIdx.x == 0) dg[row] = (float)(dot / norm);
    for (int i = threadIdx.x; i < inner; i += 256)
        dv[(size_t)row * inner + i] = (float)(gs * ((double)dw_at(i) - (double)vr[i] * k2));
}

// ---- conditioning backward (one stage): gb = (W/sigma) z + b,  z = fc_w cat(spk,noise) + fc_b,  sigma = u^T W v (u, v constants)
// grid.x: 0..R-1 -> rows of dW_orig / d sn_bias ; then 128 rows of dfc ; threads over the inner dimension
__global__ void __launch_bounds__(128)
cond_bwd_kernel(const float* __restrict__ dgb, const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ u,
                const float* __restrict__ v, const float* __restrict__ sigma_p, const float* __restrict__ spk, const float* __restrict__ noise,
                float* __restrict__ dW, float* __restrict__ dsnb, float* __restrict__ dfc_w, float* __restrict__ dfc_b,
                float* __restrict__ dz_ws, int B, int R, int spk_dim, int noise_dim, int phase) {
    const float sigma = sigma_p[0];
    const int j = threadIdx.x;                       // 0..127
    if (phase == 0) {
        // dWhat[r][j] = sum_b dgb[b][r] z[b][j]  (written to dW, fixed up in phase 2) ; dsnb[r] = sum_b dgb[b][r]
        const int r = blockIdx.x;
        float acc = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) { const float d = dgb[(size_t)b * R + r]; acc = fmaf(d, z[b * 128 + j], acc); sb += d; }
        dW[(size_t)r * 128 + j] = acc;
        if (j == 0) dsnb[r] = sb;
    } else if (phase == 1) {
        // dz[b][j] = sum_r dgb[b][r] * W[r][j] / sigma
        const int b = blockIdx.x;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(dgb[(size_t)b * R + r], W[(size_t)r * 128 + j], acc);
        dz_ws[b * 128 + j] = acc / sigma;
    } else if (phase == 2) {
        // dW_orig = dWhat/sigma - (sum(dWhat .* W)/sigma^2) u v^T ; block 0 reduces the scalar first into dz_ws[B*128]
        __shared__ float red[16];
        float part = 0.f;
        for (int r = 0; r < R; ++r) part = fmaf(dW[(size_t)r * 128 + j], W[(size_t)r * 128 + j], part);
        const float tot = v2w_block_sum(part, red);
        if (j == 0) dz_ws[B * 128] = tot;
    } else if (phase == 3) {
        const int r = blockIdx.x;
        const float tot = dz_ws[B * 128];
        dW[(size_t)r * 128 + j] = dW[(size_t)r * 128 + j] / sigma - tot / (sigma * sigma) * u[r] * v[j];
    } else {
        // fc: dfc_w[j][i] = sum_b dz[b][j] * sn[b][i] ; dfc_b[j] = sum_b dz[b][j] ; block = output row j', threads over i
        const int jj = blockIdx.x, D = spk_dim + noise_dim;
        for (int i = threadIdx.x; i < D; i += 128) {
            float acc = 0.f;
            for (int b = 0; b < B; ++b) {
                const float sn = i < spk_dim ? spk[(size_t)b * spk_dim + i] : noise[(size_t)b * noise_dim + i - spk_dim];
                acc = fmaf(dz_ws[b * 128 + jj], sn, acc);
            }
            dfc_w[(size_t)jj * D + i] = acc;
        }
        if (threadIdx.x == 0) { float sb = 0.f; for (int b = 0; b < B; ++b) sb += dz_ws[b * 128 + jj]; dfc_b[jj] = sb; }
    }
}

}  // namespace

extern "C" int v2w_cbn_bwd_sums(const float* dx, const float* xr, const float* gb, const double* stats,
                                const float* running_mean, const float* running_var, float* s12_ws, float* dgb, double* csum,
                                int B, int C, int L, int training, float eps, void* stream) {
    if (!dx || !xr || !gb || !s12_ws || !dgb || !csum || B <= 0 || C <= 0 || L <= 0) return V2W_E_ARG;
    if (training ? !stats : (!running_mean || !running_var)) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    V2W_LAUNCH(cbn_bwd_rowsums_kernel, dim3(B * C), dim3(256), 0, st, dx, xr, s12_ws, L);
    V2W_LAUNCH(cbn_bwd_sums_kernel, dim3((C + 63) / 64), dim3(64), 0, st, s12_ws, gb, stats, running_mean, running_var, dgb, csum,
                       B, C, training, eps);
    return v2w_launch_status();
}

extern "C" int v2w_cbn_bwd_apply(const float* dx, const float* xr, const float* gb, const double* stats, const double* csum,
                                 const float* running_mean, const float* running_var, float* tab_ws, float* dxr,
                                 int B, int C, int L, int training, float eps, void* stream) {
    if (!dx || !xr || !gb || !csum || !tab_ws || !dxr || B <= 0 || C <= 0 || L <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* A = tab_ws; float* Bc = tab_ws + (size_t)B * C; float* Cc = Bc + C;
    V2W_LAUNCH(cbn_bwd_tables_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, gb, stats, csum, running_mean, running_var,
                       A, Bc, Cc, B, C, training, eps);
    int gx = (L + 255) / 256; if (gx > 64) gx = 64;
    V2W_LAUNCH(affine3_kernel, dim3(gx, B * C), dim3(256), 0, st, dx, xr, A, Bc, Cc, dxr, C, L);
    return v2w_launch_status();
}

extern "C" int v2w_tail_bwd(const float* dy, const float* y, const float* x, const float* wf, float* dp_ws, double* part_ws,
                            float* dx, float* dwf, int B, int C_in, int L, int k, float slope, void* stream) {
    if (!dy || !y || !x || !wf || !dp_ws || !part_ws || !dx || !dwf || B <= 0 || C_in <= 0 || L <= 0 || k <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * L;
    int g = (int)((n + 255) / 256); if (g > 4096) g = 4096;
    V2W_LAUNCH(tanh_bwd_kernel, dim3(g), dim3(256), 0, st, dy, y, dp_ws, n);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (C_in == 16 && k == 7 && L % 4 == 0 && al16(x) && al16(dx) && al16(dp_ws)) {       // the generator's tail: one pass over x for both gradients
        constexpr int TP = 1024;
        const int ntl = (L + TP - 1) / TP;
        const long long ntiles = (long long)B * ntl;
        if (ntiles <= 0x7fffffffll) {
            const int nwg = ntiles < V2W_TAIL_PARTS ? (int)ntiles : V2W_TAIL_PARTS;
            const size_t lds = ((size_t)17 * (TP + 8) + 7 * 16 + 4 * 16 * 7) * sizeof(float);
            auto kern = tail_bwd_fused_kernel<16, 7>;
            hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, st);
            if (e != hipSuccess) return (int)e;
            V2W_LAUNCH(kern, dim3(nwg), dim3(256), lds, st, dp_ws, wf, x, dx, part_ws, L, slope, ntl, (int)ntiles);
            V2W_LAUNCH(conv_post_wgrad_reduce_kernel, dim3((C_in * k + 63) / 64), dim3(64), 0, st, part_ws, dwf, C_in, k, nwg);
            return v2w_launch_status();
        }
    }
    V2W_LAUNCH(conv_post_dgrad_kernel, dim3((L + 255) / 256, C_in, B), dim3(256), 0, st, dp_ws, wf, x, dx, C_in, L, k, slope);
    const int nsplit = 64;
    V2W_LAUNCH(conv_post_wgrad_kernel, dim3(nsplit, C_in), dim3(256), 0, st, x, dp_ws, part_ws, B, C_in, L, k, slope, nsplit);
    V2W_LAUNCH(conv_post_wgrad_reduce_kernel, dim3((C_in * k + 63) / 64), dim3(64), 0, st, part_ws, dwf, C_in, k, nsplit);
    return v2w_launch_status();
}

extern "C" int v2w_wn_bwd(const float* dwf, const float* v, const float* g, float* dv, float* dg,
                          int c_in, int c_out, int k, int transposed, void* stream) {
    if (!dwf || !v || !dv || (g && !dg) || c_in <= 0 || c_out <= 0 || k <= 0) return V2W_E_ARG;
    V2W_LAUNCH(wn_bwd_kernel, dim3(transposed ? c_in : c_out), dim3(256), 0, (hipStream_t)stream, dwf, v, g, dv, dg,
                       c_in, c_out, k, transposed);
    return v2w_launch_status();
}

extern "C" int v2w_cond_bwd(const float* dgb, const float* z, const float* sn_w, const float* sn_u, const float* sn_v, const float* sigma,
                            const float* spk, const float* noise, float* d_sn_w, float* d_sn_b, float* d_fc_w, float* d_fc_b,
                            float* dz_ws, int B, int C, int spk_dim, int noise_dim, void* stream) {
    if (!dgb || !z || !sn_w || !sn_u || !sn_v || !sigma || !spk || !noise || !d_sn_w || !d_sn_b || !d_fc_w || !d_fc_b || !dz_ws)
        return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int R = 2 * C;
    for (int phase = 0; phase < 5; ++phase) {
        const int grid = phase == 0 || phase == 3 ? R : (phase == 1 ? B : (phase == 2 ? 1 : 128));
        V2W_LAUNCH(cond_bwd_kernel, dim3(grid), dim3(128), 0, st, dgb, z, sn_w, sn_u, sn_v, sigma, spk, noise,
                           d_sn_w, d_sn_b, d_fc_w, d_fc_b, dz_ws, B, R, spk_dim, noise_dim, phase);
    }
    return v2w_launch_status();
}
