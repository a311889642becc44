// Conditional BatchNorm pieces of the Vec2Wav generator (vec2wav/modules.py:5-30, models.py:120,131-134):
//   cond_*      : z = fcs[i](cat(spk, noise)); legacy spectral-norm power iteration; [gamma|beta] = (W/sigma) z + b
//   bn_stats    : per-channel sum / sum-of-squares over (B, L): wavefront shuffles -> block -> fixed-order fp64
//   bn_finalize : mean / var / running-stat update and the folded per-sample affine a[b,c], s[b,c]
// The normalised tensor itself is never written: consumers apply a*x + s while staging their tiles.
#include "v2w_common.h"

namespace {

// ---- z[s][b][j] = fc_b[j] + sum_i fc_w[j][i] * cat(spk,noise)[b][i];  grid (B, n_stages), 128 threads (one per j)
__global__ void __launch_bounds__(128)
cond_fc_kernel(const v2w_cond_args a) {
    extern __shared__ float sn[];   // cat(spk[b], noise[b])
    const int b = blockIdx.x, s = blockIdx.y, j = threadIdx.x;
    const int D = a.spk_dim + a.noise_dim;
    for (int i = threadIdx.x; i < D; i += blockDim.x)
        sn[i] = i < a.spk_dim ? a.spk[(size_t)b * a.spk_dim + i] : a.noise[(size_t)b * a.noise_dim + (i - a.spk_dim)];
    __syncthreads();
    if (!a.fc_w[s]) {   // no fcs layer in front (ConditionalBatchNorm1d used on its own): z is the input itself
        a.z_ws[((size_t)s * a.B + b) * 128 + j] = sn[j];
        return;
    }
    const float* w = a.fc_w[s] + (size_t)j * D;
    float acc = 0.f;
    for (int i = 0; i < D; ++i) acc = fmaf(w[i], sn[i], acc);
    a.z_ws[((size_t)s * a.B + b) * 128 + j] = acc + a.fc_b[s][j];
}

// ---- legacy torch.nn.utils.spectral_norm hook (n_power_iterations = 1, eps = 1e-12); one 1024-thread block per stage.
//   training: v <- normalize(W^T u); u <- normalize(W v)   (written back in place)
//   sigma = u . (W v)
__global__ void __launch_bounds__(1024)
cond_sn_kernel(const v2w_cond_args a) {
    __shared__ float v_s[128];
    __shared__ float part[8][128];
    __shared__ float red[16];
    extern __shared__ float u_s[];   // [R] then wv[R]
    const int s = blockIdx.x, R = 2 * a.C[s];
    float* wv_s = u_s + R;
    const float* W = a.sn_w[s];
    const int tid = threadIdx.x;
    for (int r = tid; r < R; r += 1024) u_s[r] = a.sn_u[s][r];
    if (tid < 128) v_s[tid] = a.sn_v[s][tid];
    __syncthreads();
    const float eps = 1e-12f;
    if (a.training) {
        // v = W^T u : column j = tid % 128, rows r = tid/128 (mod 8): each row read is one coalesced 512-B line
        const int j = tid & 127, q = tid >> 7;
        float acc = 0.f;
        for (int r = q; r < R; r += 8) acc = fmaf(W[(size_t)r * 128 + j], u_s[r], acc);
        part[q][j] = acc;
        __syncthreads();
        float col = 0.f;
        if (tid < 128) {
#pragma unroll
            for (int k = 0; k < 8; ++k) col += part[k][tid];   // fixed order
        }
        const float n2 = v2w_block_sum(tid < 128 ? col * col : 0.f, red);
        const float inv = 1.f / fmaxf(sqrtf(n2), eps);
        if (tid < 128) v_s[tid] = col * inv;
        __syncthreads();
    }
    // wv = W v : one wave per row, lanes over the 128 columns
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < R; r += 16) {
        float p = W[(size_t)r * 128 + lane] * v_s[lane] + W[(size_t)r * 128 + 64 + lane] * v_s[64 + lane];
        p = v2w_wave_sum(p);
        if (lane == 0) wv_s[r] = p;
    }
    __syncthreads();
    if (a.training) {
        float pp = 0.f;
        for (int r = tid; r < R; r += 1024) pp += wv_s[r] * wv_s[r];
        const float n2 = v2w_block_sum(pp, red);
        const float inv = 1.f / fmaxf(sqrtf(n2), eps);
        for (int r = tid; r < R; r += 1024) u_s[r] = wv_s[r] * inv;
        __syncthreads();
        for (int r = tid; r < R; r += 1024) a.sn_u[s][r] = u_s[r];
        if (tid < 128) a.sn_v[s][tid] = v_s[tid];
    }
    float pp = 0.f;
    for (int r = tid; r < R; r += 1024) pp += u_s[r] * wv_s[r];
    const float sigma = v2w_block_sum(pp, red);
    if (tid == 0) a.sigma_ws[s] = sigma;
}

// ---- gb[s][b][r] = sn_b[r] + (sum_j W[r][j] z[j]) / sigma ; grid (B, n_stages, row groups of 64), one wave per row
__global__ void __launch_bounds__(256)
cond_linear_kernel(const v2w_cond_args a) {
    __shared__ float z_s[128];
    const int b = blockIdx.x, s = blockIdx.y, R = 2 * a.C[s];
    const int r0 = blockIdx.z * 64;
    if (r0 >= R) return;
    if (threadIdx.x < 128) z_s[threadIdx.x] = a.z_ws[((size_t)s * a.B + b) * 128 + threadIdx.x];
    __syncthreads();
    const float sigma = a.sigma_ws[s];
    const float* W = a.sn_w[s];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float z0 = z_s[lane], z1 = z_s[64 + lane];
    for (int r = r0 + wave; r < min(R, r0 + 64); r += 4) {
        // (W / sigma) z == (W z) / sigma up to one rounding of the row sum
        float p = W[(size_t)r * 128 + lane] * z0 + W[(size_t)r * 128 + 64 + lane] * z1;
        p = v2w_wave_sum(p);
        if (lane == 0) a.gb[s][(size_t)b * R + r] = p / sigma + a.sn_b[s][r];
    }
}

// ---- eval mode, everything between (spk, noise) and the per-sample affine of EVERY stage in one launch (models.py:120,131 + modules.py:20-30
// with the BatchNorm in eval mode): z = fcs[s](cat(spk, noise)), [gamma | beta] = (W z) / sigma + b, a = gamma * rstd(running_var),
// s = beta - a * running_mean.  sigma_ws[s] = u . (W v) is a function of the parameters alone: v2w_cond_sigma computes it once per weight
// version.  Grid (B, n_stages, channel groups of 32): a block recomputes z (128 x D MACs) and takes rows c and C + c of its 32 channels.
__global__ void __launch_bounds__(256)
cond_affine_eval_kernel(const v2w_cond_eval_args e) {
    extern __shared__ float sn[];                  // cat(spk[b], noise[b])  [D]
    __shared__ float z_s[128];
    __shared__ float gbl[64];
    const v2w_cond_args& a = e.c;
    const int b = blockIdx.x, s = blockIdx.y, C = a.C[s];
    const int cg = blockIdx.z * 32;
    if (cg >= C) return;
    const int D = a.spk_dim + a.noise_dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < D; i += 256)
        sn[i] = i < a.spk_dim ? a.spk[(size_t)b * a.spk_dim + i] : a.noise[(size_t)b * a.noise_dim + (i - a.spk_dim)];
    __syncthreads();
    if (!a.fc_w[s]) {
        if (tid < 128) z_s[tid] = sn[tid];
    } else {
        // one wave per output row: coalesced reads of the row, fixed-order wave sum - four rows at a time, so that their loads are in flight
        // together (at B = 1 this kernel is on the critical path: 32 dependent row round trips per wave took 60 of its 96 us)
        for (int j0 = wave * 4; j0 < 128; j0 += 16) {
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = lane; i < D; i += 64) {
                const float x = sn[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = fmaf(a.fc_w[s][(size_t)(j0 + q) * D + i], x, p[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = v2w_wave_sum(p[q]);
                if (lane == 0) z_s[j0 + q] = t + a.fc_b[s][j0 + q];
            }
        }
    }
    __syncthreads();
    const float sigma = a.sigma_ws[s];
    const float* W = a.sn_w[s];
    const float z0 = z_s[lane], z1 = z_s[64 + lane];
    for (int r0 = wave * 4; r0 < 64; r0 += 16) {   // (four rows of loads in flight, as above)
        float p[4];
        int rows[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = r0 + q, c = cg + (rr & 31);
            rows[q] = c < C ? (rr >> 5) * C + c : -1;
            const int r = rows[q] < 0 ? 0 : rows[q];
            p[q] = W[(size_t)r * 128 + lane] * z0 + W[(size_t)r * 128 + 64 + lane] * z1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = v2w_wave_sum(p[q]);
            if (lane == 0 && rows[q] >= 0) gbl[r0 + q] = t / sigma + a.sn_b[s][rows[q]];
        }
    }
    __syncthreads();
    if (tid < 32 && cg + tid < C) {
        const int c = cg + tid;
        const double mean = (double)e.running_mean[s][c], var = (double)e.running_var[s][c];
        const float rstd = (float)(1.0 / sqrt(var + (double)e.eps[s]));
        const float av = gbl[tid] * rstd;
        e.a_out[s][(size_t)b * C + c] = av;
        e.s_out[s][(size_t)b * C + c] = fmaf(-av, (float)mean, gbl[32 + tid]);
    }
}

// ---- per-channel partial sums; grid (V2W_BN_SPLITS, C); each block covers a slice of L for every batch item: its four waves
// take batch items round-robin (a slice may be as short as 64 positions), 64 lanes along the slice
__global__ void __launch_bounds__(256)
bn_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, int B, int C, int L, int slice) {
    __shared__ double red[16];
    const int sp = blockIdx.x, c = blockIdx.y;
    const int lo = sp * slice, hi = min(L, lo + slice);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double d1 = 0.0, d2 = 0.0;
    for (int b = wv; b < B; b += 4) {
        const float* row = x + ((size_t)b * C + c) * L;
        float s1 = 0.f, s2 = 0.f, u1 = 0.f, u2 = 0.f;
        int l = lo + lane;
        for (; l + 64 < hi; l += 128) {            // two independent chains per lane
            const float v = row[l], w = row[l + 64];
            s1 += v; s2 = fmaf(v, v, s2);
            u1 += w; u2 = fmaf(w, w, u2);
        }
        if (l < hi) { const float v = row[l]; s1 += v; s2 = fmaf(v, v, s2); }
        // spill the fp32 running sums into fp64 once per batch item: bounds the fp32 chain length to slice/128
        d1 += (double)s1 + (double)u1; d2 += (double)s2 + (double)u2;
    }
    const double t1 = v2w_block_sum(d1, red);
    const double t2 = v2w_block_sum(d2, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * V2W_BN_SPLITS + sp) * 2 + 0] = t1;
        partial[((size_t)c * V2W_BN_SPLITS + sp) * 2 + 1] = t2;
    }
}

__global__ void bn_reduce_kernel(const double* __restrict__ partial, double* __restrict__ stats, int C, double count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0) stats[2 * C] = count;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < V2W_BN_SPLITS; ++i) {   // fixed order -> run-to-run deterministic
        s1 += partial[((size_t)c * V2W_BN_SPLITS + i) * 2 + 0];
        s2 += partial[((size_t)c * V2W_BN_SPLITS + i) * 2 + 1];
    }
    stats[c] = s1;
    stats[C + c] = s2;
}

// This thread's rows (threadIdx.x, + 256, ..) of channel c: the (sum, sumsq) pairs added in row order - the plain loop's order -, four 8-byte loads
// in flight (a layer with thousands of rows gives a thread a dozen: one exposed latency each when taken one at a time).  Round 6: with it the
// one-level kernels serve up to 4 096 rows (the 2 944 of the 64-channel stage at B = 32 x T = 256: one launch of ~6 us instead of the two-level pair's 12)
typedef float bn_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bn_rows_of_channel(const float* __restrict__ part, int ntiles, int C, int c, double& s1, double& s2) {
    const bn_f2* p = reinterpret_cast<const bn_f2*>(part) + c;
    int t = threadIdx.x;
    for (; t + 768 < ntiles; t += 1024) {
        const bn_f2 a = p[(size_t)t * C], b = p[(size_t)(t + 256) * C], d = p[(size_t)(t + 512) * C], e = p[(size_t)(t + 768) * C];
        s1 += (double)a[0]; s2 += (double)a[1];
        s1 += (double)b[0]; s2 += (double)b[1];
        s1 += (double)d[0]; s2 += (double)d[1];
        s1 += (double)e[0]; s2 += (double)e[1];
    }
    for (; t < ntiles; t += 256) {
        const bn_f2 a = p[(size_t)t * C];
        s1 += (double)a[0]; s2 += (double)a[1];
    }
}

// stats[c] = sum over tiles of part[tile][c][0..1] in fp64, fixed order; one block per channel
__global__ void __launch_bounds__(256)
bn_reduce_partials_kernel(const float* __restrict__ part, double* __restrict__ stats, int ntiles, int C, double count) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    bn_rows_of_channel(part, ntiles, C, c, s1, s2);
    const double t1 = v2w_block_sum(s1, red);
    const double t2 = v2w_block_sum(s2, red);
    if (threadIdx.x == 0) {
        stats[c] = t1;
        stats[C + c] = t2;
        if (c == 0) stats[2 * C] = count;
    }
}

// ---- the two-level form of bn_reduce_partials + bn_finalize for the many-tile layers (the fused stage kernels write one row of partial
// sums per 224 positions: 23 424 rows at BASELINE configs[2]; one block per channel read them with a stride of 2 C floats - 8 useful bytes
// per 128-byte line - in 10-30 us).  Level 1: slice s of the rows, whole rows at a time (coalesced, four rows in flight per thread), fp64
// sums in a fixed order -> slices[s][2 C].  Level 2 (bn_finalize_slices_kernel): 16 channels per block - the slices are added in a fixed
// order by 16 x 16 threads through LDS, then the block writes (a, s) of its channels for every sample.  Two short launches, bit-reproducible.
__global__ void __launch_bounds__(256)
bn_reduce_slices_kernel(const float* __restrict__ part, double* __restrict__ slices, int ntiles, int C) {
    __shared__ double red[256];
    const int ns = gridDim.x, s = blockIdx.x;
    const int lo = (int)((long long)ntiles * s / ns), hi = (int)((long long)ntiles * (s + 1) / ns);
    const int W = 2 * C;                                   // floats per row
    for (int j0 = 0; j0 < W; j0 += 256) {                  // (C = 256: two column passes)
        const int cols = W - j0 < 256 ? W - j0 : 256;      // a power of two for the generator's widths; any width works
        const int lanes = 256 / cols > 0 ? 256 / cols : 1; // rows in flight per step
        const int j = threadIdx.x % cols, tl = threadIdx.x / cols;
        double acc = 0.0;
        if (tl < lanes) {
            const float* p = part + j0 + j;
            int t = lo + tl;
            for (; t + 3 * lanes < hi; t += 4 * lanes) {   // four independent loads, added in row order
                const float v0 = p[(size_t)t * W], v1 = p[(size_t)(t + lanes) * W], v2 = p[(size_t)(t + 2 * lanes) * W], v3 = p[(size_t)(t + 3 * lanes) * W];
                acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
            }
            for (; t < hi; t += lanes) acc += (double)p[(size_t)t * W];
        }
        red[threadIdx.x] = tl < lanes ? acc : 0.0;
        __syncthreads();
        if (threadIdx.x < cols) {
            double tot = 0.0;
            for (int q = 0; q < lanes; ++q) tot += red[q * cols + threadIdx.x];     // fixed order
            // row layout [c][2] = (sum, sumsq) pairs -> slices[s][stat][c]
            const int col = j0 + threadIdx.x;
            slices[(size_t)s * W + (col & 1) * C + (col >> 1)] = tot;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
bn_finalize_slices_kernel(const double* __restrict__ slices, int nslices, double count, const float* __restrict__ gb,
                          float* running_mean, float* running_var, int64_t* nbt,
                          float* __restrict__ a_out, float* __restrict__ s_out, int B, int C, float momentum, float eps) {
    __shared__ double red[2][16][16];
    __shared__ double tot[2][16];
    __shared__ float mr[2][16];
    const int tid = threadIdx.x, cl = tid & 15, g = tid >> 4;
    const int c0 = blockIdx.x * 16, c = c0 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int q = g; q < nslices; q += 16) { s1 += slices[(size_t)q * 2 * C + c]; s2 += slices[(size_t)q * 2 * C + C + c]; }
    red[0][g][cl] = s1; red[1][g][cl] = s2;
    __syncthreads();
    if (tid < 32) {
        const int st = tid >> 4;
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += red[st][q][cl];
        tot[st][cl] = t;
    }
    __syncthreads();
    if (tid < 16 && c < C) {
        const double mean = tot[0][cl] / count;
        double var = tot[1][cl] / count - mean * mean;        // biased (normalisation) variance
        if (var < 0.0) var = 0.0;
        mr[0][cl] = (float)mean;
        mr[1][cl] = (float)(1.0 / sqrt(var + (double)eps));
        const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        if (c == 0 && nbt) *nbt += 1;
    }
    __syncthreads();
    if (c < C) {
        const float mean = mr[0][cl], rstd = mr[1][cl];
        for (int b = g; b < B; b += 16) {
            const float gamma = gb[(size_t)b * 2 * C + c], beta = gb[(size_t)b * 2 * C + C + c];
            const float av = gamma * rstd;
            a_out[(size_t)b * C + c] = av;
            s_out[(size_t)b * C + c] = fmaf(-av, mean, beta);
        }
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ gb,
                                   float* running_mean, float* running_var, int64_t* nbt,
                                   float* __restrict__ a_out, float* __restrict__ s_out,
                                   int B, int C, int training, float momentum, float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    double mean, var, count = 1.0;
    if (training) {
        count = stats[2 * C];
        mean = stats[c] / count;
        var = stats[C + c] / count - mean * mean;   // biased (normalisation) variance
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gamma = gb[(size_t)b * 2 * C + c], beta = gb[(size_t)b * 2 * C + C + c];
    const float av = gamma * rstd;
    a_out[idx] = av;
    s_out[idx] = fmaf(-av, (float)mean, beta);
    if (training && b == 0) {
        // F.batch_norm: running = (1-m)*running + m*stat, running_var with the UNBIASED batch variance
        const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        if (c == 0 && nbt) *nbt += 1;
    }
}

// bn_reduce_partials_kernel + bn_finalize_kernel(training) as ONE launch (ABI v34): block c adds channel c's partial rows (the same fp64
// additions in the same order), keeps the sums in stats[] for whoever reads them later, then writes the running statistics and (a, s) of its
// channel for every sample - nothing of the finalisation crosses channels.  One launch less on the chain producer -> statistics -> consumer that
// every stage boundary of a train-mode forward waits on (5 us each at B = 32 x T = 256, four of the five boundaries).
__global__ void __launch_bounds__(256)
bn_reduce_finalize_kernel(const float* __restrict__ part, double* __restrict__ stats, int ntiles, double count, const float* __restrict__ gb,
                          float* running_mean, float* running_var, int64_t* nbt, float* __restrict__ a_out, float* __restrict__ s_out,
                          int B, int C, float momentum, float eps) {
    __shared__ double red[16];
    __shared__ double tot[2];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    bn_rows_of_channel(part, ntiles, C, c, s1, s2);
    const double t1 = v2w_block_sum(s1, red);
    const double t2 = v2w_block_sum(s2, red);
    if (threadIdx.x == 0) {
        tot[0] = t1; tot[1] = t2;
        stats[c] = t1;
        stats[C + c] = t2;
        if (c == 0) stats[2 * C] = count;
    }
    __syncthreads();
    const double mean = tot[0] / count;
    double var = tot[1] / count - mean * mean;      // biased (normalisation) variance
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int b = threadIdx.x; b < B; b += 256) {
        const float gamma = gb[(size_t)b * 2 * C + c], beta = gb[(size_t)b * 2 * C + C + c];
        const float av = gamma * rstd;
        a_out[(size_t)b * C + c] = av;
        s_out[(size_t)b * C + c] = fmaf(-av, (float)mean, beta);
    }
    if (threadIdx.x == 0) {
        const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        if (c == 0 && nbt) *nbt += 1;
    }
}

__global__ void __launch_bounds__(256)
affine_apply_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ s,
                    float* __restrict__ out, int L) {
    const int row = blockIdx.y;   // b*C + c
    const float av = a[row], sv = s[row];
    const float* src = x + (size_t)row * L;
    float* dst = out + (size_t)row * L;
    for (int l = blockIdx.x * 256 + threadIdx.x; l < L; l += gridDim.x * 256) dst[l] = fmaf(av, src[l], sv);
}

}  // namespace

extern "C" int v2w_affine_apply(const float* x, const float* a, const float* s, float* out, int B, int C, int L, void* stream) {
    if (!x || !a || !s || !out || B <= 0 || C <= 0 || L <= 0) return V2W_E_ARG;
    int gx = (L + 255) / 256; if (gx > 64) gx = 64;
    V2W_LAUNCH(affine_apply_kernel, dim3(gx, B * C), dim3(256), 0, (hipStream_t)stream, x, a, s, out, L);
    return v2w_launch_status();
}

extern "C" int v2w_cond_gamma_beta(const v2w_cond_args* a, void* stream) {
    if (!a || !a->spk || (!a->noise && a->noise_dim > 0) || !a->z_ws || !a->sigma_ws) return V2W_E_ARG;
    if (a->spk_dim <= 0 || a->noise_dim < 0) return V2W_E_ARG;
    if (a->n_stages <= 0 || a->n_stages > V2W_MAX_STAGES || a->B <= 0) return V2W_E_ARG;
    int maxR = 0;
    for (int s = 0; s < a->n_stages; ++s) {
        if ((a->fc_w[s] == nullptr) != (a->fc_b[s] == nullptr)) return V2W_E_ARG;
        if (!a->fc_w[s] && a->spk_dim + a->noise_dim != 128) return V2W_E_SHAPE;
        if (!a->sn_w[s] || !a->sn_b[s] || !a->sn_u[s] || !a->sn_v[s] || !a->gb[s] || a->C[s] <= 0)
            return V2W_E_ARG;
        if (2 * a->C[s] > maxR) maxR = 2 * a->C[s];
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_fc = (size_t)(a->spk_dim + a->noise_dim) * sizeof(float);
    V2W_LAUNCH(cond_fc_kernel, dim3(a->B, a->n_stages), dim3(128), lds_fc, st, *a);
    V2W_LAUNCH(cond_sn_kernel, dim3(a->n_stages), dim3(1024), (size_t)2 * maxR * sizeof(float), st, *a);
    V2W_LAUNCH(cond_linear_kernel, dim3(a->B, a->n_stages, (maxR + 63) / 64), dim3(256), 0, st, *a);
    return v2w_launch_status();
}

// sigma_ws[s] = u_s . (W_s v_s) of every stage (training != 0: after one power iteration, u and v updated in place) - the spectral-norm half
// of v2w_cond_gamma_beta alone.  In eval mode sigma depends on the parameters only: call it once per weight version.
extern "C" int v2w_cond_sigma(const v2w_cond_args* a, void* stream) {
    if (!a || !a->sigma_ws || a->n_stages <= 0 || a->n_stages > V2W_MAX_STAGES) return V2W_E_ARG;
    int maxR = 0;
    for (int s = 0; s < a->n_stages; ++s) {
        if (!a->sn_w[s] || !a->sn_u[s] || !a->sn_v[s] || a->C[s] <= 0) return V2W_E_ARG;
        if (2 * a->C[s] > maxR) maxR = 2 * a->C[s];
    }
    V2W_LAUNCH(cond_sn_kernel, dim3(a->n_stages), dim3(1024), (size_t)2 * maxR * sizeof(float), (hipStream_t)stream, *a);
    return v2w_launch_status();
}

extern "C" int v2w_cond_affine_eval(const v2w_cond_eval_args* e, void* stream) {
    if (!e) return V2W_E_ARG;
    const v2w_cond_args* a = &e->c;
    if (!a->spk || (!a->noise && a->noise_dim > 0) || !a->sigma_ws) return V2W_E_ARG;
    if (a->spk_dim <= 0 || a->noise_dim < 0 || a->n_stages <= 0 || a->n_stages > V2W_MAX_STAGES || a->B <= 0) return V2W_E_ARG;
    int maxC = 0;
    for (int s = 0; s < a->n_stages; ++s) {
        if ((a->fc_w[s] == nullptr) != (a->fc_b[s] == nullptr)) return V2W_E_ARG;
        if (!a->fc_w[s] && a->spk_dim + a->noise_dim != 128) return V2W_E_SHAPE;
        if (!a->sn_w[s] || !a->sn_b[s] || a->C[s] <= 0) return V2W_E_ARG;
        if (!e->running_mean[s] || !e->running_var[s] || !e->a_out[s] || !e->s_out[s]) return V2W_E_ARG;
        if (a->C[s] > maxC) maxC = a->C[s];
    }
    const size_t lds = (size_t)(a->spk_dim + a->noise_dim) * sizeof(float);
    if (lds > 48 * 1024) return V2W_E_SHAPE;
    V2W_LAUNCH(cond_affine_eval_kernel, dim3(a->B, a->n_stages, (maxC + 31) / 32), dim3(256), lds, (hipStream_t)stream, *e);
    return v2w_launch_status();
}

extern "C" int v2w_bn_stats(const float* x, double* stats, double* partial_ws, int B, int C, int L, void* stream) {
    if (!x || !stats || !partial_ws || B <= 0 || C <= 0 || L <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int slice = (L + V2W_BN_SPLITS - 1) / V2W_BN_SPLITS;
    slice = (slice + 63) & ~63;   // keep wave-wide row reads aligned
    V2W_LAUNCH(bn_stats_kernel, dim3(V2W_BN_SPLITS, C), dim3(256), 0, st, x, partial_ws, B, C, L, slice);
    V2W_LAUNCH(bn_reduce_kernel, dim3((C + 63) / 64), dim3(64), 0, st, partial_ws, stats, C, (double)B * (double)L);
    return v2w_launch_status();
}

extern "C" int v2w_bn_reduce_partials(const float* part, int ntiles, int C, double count, double* stats, void* stream) {
    if (!part || !stats || ntiles <= 0 || C <= 0 || count <= 0.0) return V2W_E_ARG;
    V2W_LAUNCH(bn_reduce_partials_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, part, stats, ntiles, C, count);
    return v2w_launch_status();
}

extern "C" int v2w_bn_finalize(const double* stats, const float* gb,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* a_out, float* s_out, int B, int C, int training,
                               float momentum, float eps, void* stream) {
    if (!gb || !a_out || !s_out || !running_mean || !running_var || B <= 0 || C <= 0) return V2W_E_ARG;
    if (training && !stats) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    V2W_LAUNCH(bn_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, stats, gb,
                       running_mean, running_var, num_batches_tracked, a_out, s_out, B, C, training, momentum, eps);
    return v2w_launch_status();
}

// v2w_bn_reduce_partials + v2w_bn_finalize(training = 1) in one launch (ABI v34): same sums, same (a, s), same running statistics, bit for bit.
// Not for a data-parallel run (the all-reduce of `stats` sits between the two).
extern "C" int v2w_bn_reduce_finalize(const float* part, int ntiles, double count, const float* gb,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      double* stats, float* a_out, float* s_out, int B, int C, float momentum, float eps, void* stream) {
    if (!part || ntiles <= 0 || !(count > 0.0) || !gb || !running_mean || !running_var || !stats || !a_out || !s_out || B <= 0 || C <= 0)
        return V2W_E_ARG;
    V2W_LAUNCH(bn_reduce_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, part, stats, ntiles, count, gb,
                       running_mean, running_var, num_batches_tracked, a_out, s_out, B, C, momentum, eps);
    return v2w_launch_status();
}

// The two-level form for layers with thousands of partial rows (ABI v28): v2w_bn_reduce_slices sums slice s of the rows into
// slices[s][2 C] fp64 (nslices blocks, whole rows at a time); v2w_bn_finalize_slices is v2w_bn_finalize in train mode reading those slices
// (added in slice order) instead of the [sum | sumsq | count] array.  Same values as v2w_bn_reduce_partials + v2w_bn_finalize up to the
// order of the fp64 additions; a data-parallel run, which all-reduces the array, keeps the one-level form.
extern "C" int v2w_bn_reduce_slices(const float* part, int ntiles, int C, double* slices, int nslices, void* stream) {
    if (!part || !slices || ntiles <= 0 || C <= 0 || nslices <= 0 || nslices > 1024) return V2W_E_ARG;
    V2W_LAUNCH(bn_reduce_slices_kernel, dim3(nslices), dim3(256), 0, (hipStream_t)stream, part, slices, ntiles, C);
    return v2w_launch_status();
}
extern "C" int v2w_bn_finalize_slices(const double* slices, int nslices, double count, const float* gb,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      float* a_out, float* s_out, int B, int C, float momentum, float eps, void* stream) {
    if (!slices || nslices <= 0 || !(count > 0.0) || !gb || !a_out || !s_out || !running_mean || !running_var || B <= 0 || C <= 0) return V2W_E_ARG;
    V2W_LAUNCH(bn_finalize_slices_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, slices, nslices, count, gb,
                       running_mean, running_var, num_batches_tracked, a_out, s_out, B, C, momentum, eps);
    return v2w_launch_status();
}
