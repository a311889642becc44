// Memory-bound ends of the MPD / MSD discriminator forwards (reference vec2wav/models.py:158-275).
// The convolutions themselves run on the f32 MFMA tile kernel (v2w_conv1d_fwd) as stride-1 problems:
//   * a stride-s Conv1d (k taps, padding P) over x equals a stride-1 conv over the s phase-de-interleaved copies of x stacked as
//     channels:  j - P = s*q + r  ->  out[t] = sum_{q,r,c} w[c][s*q + r + P] * x_r[c][t + q],  x_r[c][u] = x[c][s*u + r];
//   * DiscriminatorP's (k, 1) Conv2d over the (B, C, H, p) view is that same conv along H for p independent columns: on the
//     flattened (H*p) axis it is a Conv1d with dilation p (zero padding in H and in H*p coincide since every shift is a
//     multiple of p), so feature maps stay in the reference's (B, C, H, p) layout with no transposes;
//   * the first layers (C_in = 1) are unfolded into k (padded to 16) shifted rows and run as 1-tap MFMA convs.
#include "v2w_common.h"

namespace {

// x (B, C, L, inner) -> out (B, s*C, U, inner), U = ceil(L / s):
//   out[b][((c / Cg)*s + r)*Cg + c % Cg][u][w] = x[b][c][s*u + r][w]  (0 past L); Cg = channels per conv group (C when ungrouped)
// Rows of x / out are ipitch / opitch floats apart (>= L*inner / U*inner; the out tail is zero-filled): the conv kernel's float4
// staging wants rows that are multiples of 4 floats, whatever the period.
__global__ void __launch_bounds__(256)
phase_split_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int Cg, int L, int inner, int s, int U,
                   int ipitch, int opitch) {
    const int b = blockIdx.y;
    const size_t total = (size_t)s * C * opitch;
    const float* xb = x + (size_t)b * C * ipitch;
    float* ob = out + (size_t)b * total;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int cs = (int)(idx / opitch);
        const int rem = (int)(idx - (size_t)cs * opitch);
        const int u = rem / inner, w = rem - u * inner;
        const int g = cs / (s * Cg), rr = cs - g * s * Cg;
        const int r = rr / Cg, c = g * Cg + (rr - r * Cg);
        const int l = s * u + r;
        ob[idx] = (u < U && l < L) ? xb[(size_t)c * ipitch + (size_t)l * inner + w] : 0.f;
    }
}

// The same, one OUTPUT ROW per blockIdx.x (the (group, phase, channel) arithmetic once per row instead of five integer divisions per
// element) and four consecutive output floats per thread: one division by `inner` per element (fp32 estimate + correction), a 16-byte
// store.  Rows of 4-float multiples at 16-byte aligned bases (what the discriminators' pitched buffers are).  The discriminators' phase
// splits were 37 ms of a 646 ms GAN iteration on the element-wise form.
__global__ void __launch_bounds__(256)
phase_split_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int Cg, int L, int inner, int s, int U,
                        int ipitch, int opitch, float inv_inner) {
    const int row = blockIdx.x;                            // b * s * C + cs
    const int b = row / (s * C), cs = row - b * s * C;
    const int g = cs / (s * Cg), rr = cs - g * s * Cg;
    const int r = rr / Cg, c = g * Cg + (rr - r * Cg);
    const float* src = x + ((size_t)b * C + c) * ipitch;
    float* dst = out + (size_t)row * opitch;
    const int n4 = opitch >> 2;
    for (int q4 = blockIdx.y * 256 + threadIdx.x; q4 < n4; q4 += gridDim.y * 256) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rem = 4 * q4 + e;
            int u = inner == 1 ? rem : (int)((float)rem * inv_inner);
            if (inner != 1) { if (u * inner > rem) --u; else if ((u + 1) * inner <= rem) ++u; }
            const int w = rem - u * inner;
            const int l = s * u + r;
            v[e] = (u < U && l < L) ? src[(size_t)l * inner + w] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dst + 4 * q4) = v;
    }
}

// The same for inner == 1, one group, dense rows whose phase length is a multiple of 4 (the generator's transposed-conv input gradients:
// B x C rows of S * U floats): a thread reads the 4 S consecutive floats of four output positions as S float4s (coalesced) and writes one
// float4 to each of the S phase rows (coalesced) - the element-wise form above reads with stride S and ran at 1.3 TB/s.
template <int S>
__global__ void __launch_bounds__(256)
phase_split_vec_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int U) {
    const int row = blockIdx.y;                        // b * C + c
    const int b = row / C, c = row - b * C;
    const float* src = x + (size_t)row * S * U;
    float* dst = out + ((size_t)b * S * C + c) * U;    // phase r of channel c: row r * C + c of batch item b
    for (int q = blockIdx.x * 256 + threadIdx.x; 4 * q < U; q += gridDim.x * 256) {
        float v[4 * S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + (size_t)4 * S * q + 4 * i);
            v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
        }
#pragma unroll
        for (int r = 0; r < S; ++r)
            *reinterpret_cast<f32x4*>(dst + (size_t)r * C * U + 4 * q) = f32x4{v[r], v[S + r], v[2 * S + r], v[3 * S + r]};
    }
}

// x (B, T) single-channel audio, read as (H, inner) rows with a reflect pad on the right up to H*inner samples
// (models.py:176-181) -> out (B, rows, U*inner): out[b][j][u*inner + w] = xpad[(s*u + j - pad)*inner + w] for j < k and
// 0 <= s*u + j - pad < H, else 0.  Turns the C_in = 1 first layers into rows-channel 1-tap convs.
__global__ void __launch_bounds__(256)
unfold1_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int H, int inner, int s, int k, int pad, int rows, int U,
               int opitch) {
    const int b = blockIdx.y;
    const size_t row = (size_t)opitch;
    const size_t total = (size_t)rows * row;
    const float* xb = x + (size_t)b * T;
    float* ob = out + (size_t)b * total;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx / row);
        const int rem = (int)(idx - (size_t)j * row);
        const int u = rem / inner, w = rem - u * inner;
        const int h = s * u + j - pad;
        float v = 0.f;
        if (j < k && u < U && h >= 0 && h < H) {
            const int i = h * inner + w;
            v = xb[i < T ? i : 2 * (T - 1) - i];
        }
        ob[idx] = v;
    }
}

// x (B, C, L, inner) -> out (B, k*C, U, inner), U = (L + 2 pad - k)/s + 1: out[b][j*C + c][u][w] = x[b][c][s*u + j - pad][w] (0 outside
// [0, L)): a short strided conv (DiscriminatorP's k = 5, stride 3) as ONE 1-tap conv over k*C channels: exactly the reference's
// k*C*C_out MACs per output (the phase-stacked form pads 5 taps to 6 slots) and no halo.
__global__ void __launch_bounds__(256)
unfold_taps_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int L, int inner, int s, int k, int pad, int U,
                   int ipitch, int opitch) {
    const int b = blockIdx.y;
    const size_t total = (size_t)k * C * opitch;
    const float* xb = x + (size_t)b * C * ipitch;
    float* ob = out + (size_t)b * total;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int jc = (int)(idx / opitch);
        const int rem = (int)(idx - (size_t)jc * opitch);
        const int u = rem / inner, w = rem - u * inner;
        const int j = jc / C, c = jc - j * C;
        const int l = s * u + j - pad;
        ob[idx] = (u < U && l >= 0 && l < L) ? xb[(size_t)c * ipitch + (size_t)l * inner + w] : 0.f;
    }
}

// AvgPool1d(4, 2, padding=2) (models.py:255-258; zero padding counted): out[t] = (x[2t-2] + x[2t-1] + x[2t] + x[2t+1]) / 4
__global__ void __launch_bounds__(256)
avgpool4_kernel(const float* __restrict__ x, float* __restrict__ out, int L, int Lo) {
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * L;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < Lo; t += gridDim.x * 256) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 2 * t + j - 2;
            if (i >= 0 && i < L) acc += xb[i];
        }
        out[(size_t)b * Lo + t] = acc * 0.25f;
    }
}

}  // namespace

extern "C" int v2w_phase_split(const float* x, float* out, int B, int C, int Cg, int L, int inner, int s, int ipitch, int opitch,
                               void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || Cg <= 0 || C % Cg || L <= 0 || inner <= 0 || s <= 0) return V2W_E_ARG;
    const int U = (L + s - 1) / s;
    if (ipitch <= 0) ipitch = L * inner;
    if (opitch <= 0) opitch = U * inner;
    if (ipitch < L * inner || opitch < U * inner) return V2W_E_ARG;
    const size_t total = (size_t)s * C * opitch;
    if (inner == 1 && Cg == C && L == s * U && U % 4 == 0 && ipitch == L && opitch == U && (long long)B * C <= 65535 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && (s == 2 || s == 4 || s == 5 || s == 8)) {
        int gx = (U / 4 + 255) / 256; if (gx > 64) gx = 64;
        const dim3 grid(gx, B * C);
        hipStream_t st = (hipStream_t)stream;
        switch (s) {
            case 2: V2W_LAUNCH(phase_split_vec_kernel<2>, grid, dim3(256), 0, st, x, out, C, U); break;
            case 4: V2W_LAUNCH(phase_split_vec_kernel<4>, grid, dim3(256), 0, st, x, out, C, U); break;
            case 5: V2W_LAUNCH(phase_split_vec_kernel<5>, grid, dim3(256), 0, st, x, out, C, U); break;
            default: V2W_LAUNCH(phase_split_vec_kernel<8>, grid, dim3(256), 0, st, x, out, C, U); break;
        }
        return v2w_launch_status();
    }
    if (opitch % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (long long)B * s * C < (1ll << 31) && opitch < (1 << 22)) {
        int gy = (opitch / 4 + 255) / 256; if (gy > 64) gy = 64;
        V2W_LAUNCH(phase_split_rows_kernel, dim3(B * s * C, gy), dim3(256), 0, (hipStream_t)stream, x, out, C, Cg, L, inner, s, U,
                           ipitch, opitch, 1.f / (float)inner);
        return v2w_launch_status();
    }
    int gx = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    V2W_LAUNCH(phase_split_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, out, C, Cg, L, inner, s, U, ipitch, opitch);
    return v2w_launch_status();
}

// x: `rows` rows of `pitch` floats: zero [valid, pitch) of every row (the conv kernel computed them as ordinary positions;
// a stride-1 consumer must see the reference's zero padding there).
__global__ void __launch_bounds__(256)
zero_tail_kernel(float* __restrict__ x, long rows, int pitch, int valid) {
    const int n = pitch - valid;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < rows * n; idx += (long)gridDim.x * 256)
        x[(idx / n) * pitch + valid + idx % n] = 0.f;
}

extern "C" int v2w_zero_tail(float* x, long long rows, int pitch, int valid, void* stream) {
    if (!x || rows <= 0 || pitch <= 0 || valid < 0 || valid > pitch) return V2W_E_ARG;
    if (valid == pitch) return 0;
    const long total = rows * (pitch - valid);
    int gx = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    V2W_LAUNCH(zero_tail_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, x, (long)rows, pitch, valid);
    return v2w_launch_status();
}

extern "C" int v2w_unfold1(const float* x, float* out, int B, int T, int H, int inner, int s, int k, int pad, int rows, int opitch,
                           void* stream) {
    if (!x || !out || B <= 0 || T <= 1 || H <= 0 || inner <= 0 || s <= 0 || k <= 0 || pad < 0 || rows < k) return V2W_E_ARG;
    if ((long long)H * inner < T || (long long)H * inner - T >= T) return V2W_E_ARG;     // reflect pad shorter than the signal
    if (H + 2 * pad < k) return V2W_E_SHAPE;
    const int U = (H + 2 * pad - k) / s + 1;
    if (opitch <= 0) opitch = U * inner;
    if (opitch < U * inner) return V2W_E_ARG;
    const size_t total = (size_t)rows * opitch;
    int gx = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    V2W_LAUNCH(unfold1_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, out, T, H, inner, s, k, pad, rows, U, opitch);
    return v2w_launch_status();
}

extern "C" int v2w_avgpool4(const float* x, float* out, int B, int L, void* stream) {
    if (!x || !out || B <= 0 || L <= 0) return V2W_E_ARG;
    const int Lo = L / 2 + 1;
    int gx = (Lo + 255) / 256; if (gx > 4096) gx = 4096;
    V2W_LAUNCH(avgpool4_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, out, L, Lo);
    return v2w_launch_status();
}


extern "C" int v2w_unfold_taps(const float* x, float* out, int B, int C, int L, int inner, int s, int k, int pad, int ipitch, int opitch,
                               void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || L <= 0 || inner <= 0 || s <= 0 || k <= 0 || pad < 0) return V2W_E_ARG;
    if (L + 2 * pad < k) return V2W_E_SHAPE;
    const int U = (L + 2 * pad - k) / s + 1;
    if (ipitch <= 0) ipitch = L * inner;
    if (opitch <= 0) opitch = U * inner;
    if (ipitch < L * inner || opitch < U * inner) return V2W_E_ARG;
    const size_t total = (size_t)k * C * opitch;
    int gx = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    V2W_LAUNCH(unfold_taps_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, out, C, L, inner, s, k, pad, U, ipitch, opitch);
    return v2w_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the discriminators (both training steps of train.py:188-215 differentiate through them).  The convolutions' input
// gradients run on the forward conv kernel (transposed, tap-flipped weights), the weight gradients on v2w_wgrad_slice; below are
// the memory-bound pieces between them.
namespace {

// dz = (g + d) * lrelu'(f): g = gradient that arrived on the returned feature map (dense (rows, valid) or NULL), d = input
// gradient of the next conv, f = the ACTIVATED map (sign(f) = sign of the pre-activation since slope > 0).  slope == 1: no
// activation (conv_post).  The pitch tail is written as 0: those columns are ordinary positions to the kernels.
// d comes either pitched like f (MERGE = false) or in the phase-stacked form dxs (B, s*C, dpitch) of the strided layer above
// (MERGE = true, v2w_phase_merge folded in): d[b][c][l*inner + w] = dxs[b][((c/Cg)*s + l%s)*Cg + c%Cg][(l/s)*inner + w].
// Row-structured: a block owns one (b, c) row (four rows, one wave each, when rows are short), so the per-row sum of dz - the bias
// gradient's partial - falls out of the same pass (rowsum, optional; summed over b in fixed order by rowsum_reduce_kernel).
template <bool MERGE>
__global__ void __launch_bounds__(256)
disc_dz_rows_kernel(const float* __restrict__ f, const float* __restrict__ g, const float* __restrict__ d, float* __restrict__ dz,
                    float* __restrict__ rowsum, long rows, int C, int Cg, int inner, int s, int dpitch, int pitch, int valid,
                    float slope, int rpb) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = rpb == 1 ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;
    const int t0 = rpb == 1 ? threadIdx.x : lane, tstep = rpb == 1 ? 256 : 64;
    float acc = 0.f;
    if (row < rows) {
        const long b = row / C;
        const int c = (int)(row - b * C);
        const float* fr = f + row * pitch;
        const float* gr = g ? g + row * valid : nullptr;
        float* zr = dz + row * pitch;
        const float* dr = d ? (MERGE ? d + (size_t)b * s * C * dpitch : d + row * pitch) : nullptr;
        const int csb = MERGE ? (c / Cg) * s * Cg + c % Cg : 0;
        for (int t = t0; t < pitch; t += tstep) {
            float v = 0.f;
            if (t < valid) {
                if (MERGE) {
                    const int l = t / inner, w = t - l * inner;
                    const int u = l / s, r = l - u * s;
                    v = dr[(size_t)(csb + r * Cg) * dpitch + (size_t)u * inner + w];
                } else if (dr) {
                    v = dr[t];
                }
                if (gr) v += gr[t];
                if (slope != 1.f && !(fr[t] > 0.f)) v *= slope;
            }
            zr[t] = v;
            acc += v;
        }
    }
    if (!rowsum) return;                                  // (uniform over the block)
    if (rpb == 1) {
        const float tot = v2w_block_sum(acc, red);
        if (threadIdx.x == 0) rowsum[row] = tot;
    } else {
        const float tot = v2w_wave_sum(acc);
        if (lane == 0 && row < rows) rowsum[row] = tot;
    }
}

// db[c] = sum_b rowsum[b][c], fp64 accumulation in fixed order
__global__ void rowsum_reduce_kernel(const float* __restrict__ rowsum, float* __restrict__ db, int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0;
    for (int b = 0; b < B; ++b) a += (double)rowsum[(size_t)b * C + c];
    db[c] = (float)a;
}

// inverse of phase_split_kernel: out[b][c][l][w] = dxs[b][((c/Cg)*s + r)*Cg + c%Cg][u][w], l = s*u + r < L
__global__ void __launch_bounds__(256)
phase_merge_kernel(const float* __restrict__ dxs, float* __restrict__ out, int C, int Cg, int L, int inner, int s, int ipitch, int opitch) {
    const int b = blockIdx.y;
    const size_t row = (size_t)L * inner;
    const size_t total = (size_t)C * row;
    const float* xb = dxs + (size_t)b * s * C * ipitch;
    float* ob = out + (size_t)b * C * opitch;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx / row);
        const int rem = (int)(idx - (size_t)c * row);
        const int l = rem / inner, w = rem - l * inner;
        const int u = l / s, r = l - u * s;
        const int cs = ((c / Cg) * s + r) * Cg + c % Cg;
        ob[(size_t)c * opitch + rem] = xb[(size_t)cs * ipitch + (size_t)u * inner + w];
    }
}

// backward of unfold1_kernel: dx[b][i] = sum of dxu over every (row j, column) that read sample i, reflected tail included
__global__ void __launch_bounds__(256)
fold1_kernel(const float* __restrict__ dxu, float* __restrict__ dx, int T, int H, int inner, int s, int k, int pad, int rows, int U, int ipitch) {
    const int b = blockIdx.y;
    const float* xb = dxu + (size_t)b * rows * ipitch;
    auto at = [&](int pos) {            // gradient that reached padded position pos = h*inner + w
        const int h = pos / inner, w = pos - h * inner;
        float v = 0.f;
        for (int j = 0; j < k; ++j) {
            const int n = h + pad - j;
            if (n < 0 || n % s) continue;
            const int u = n / s;
            if (u < U) v += xb[(size_t)j * ipitch + (size_t)u * inner + w];
        }
        return v;
    };
    const int Tp = H * inner;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) {
        float v = at(i);
        const int m = 2 * (T - 1) - i;       // the right reflect pad reads sample i at position m
        if (m >= T && m < Tp) v += at(m);
        dx[(size_t)b * T + i] = v;
    }
}

// backward of avgpool4_kernel: dx[i] = 0.25 * sum of dout[t] over 2t + j - 2 = i, j < 4
__global__ void __launch_bounds__(256)
avgpool4_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int L, int Lo) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = i + 2 - j;
            if (n >= 0 && (n & 1) == 0 && (n >> 1) < Lo) acc += dout[(size_t)b * Lo + (n >> 1)];
        }
        dx[(size_t)b * L + i] = acc * 0.25f;
    }
}

// weight gradient of a C_out = 1 conv (conv_post): dwf[t][ci] = sum_{b,l} x[b][ci][l + (t - tap0)*dil] * dz[b][l]; one block per ci
__global__ void __launch_bounds__(256)
cout1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dwf, int B, int C, int L, int k, int dil, int tap0) {
    __shared__ double red[16];
    const int ci = blockIdx.x;
    for (int t = 0; t < k; ++t) {
        const int off = (t - tap0) * dil;
        double acc = 0.0;
        for (int b = 0; b < B; ++b) {
            const float* xr = x + ((size_t)b * C + ci) * L;
            const float* dr = dz + (size_t)b * L;
            float a = 0.f;
            for (int l = threadIdx.x; l < L; l += 256) {
                const int li = l + off;
                if (li >= 0 && li < L) a = fmaf(xr[li], dr[l], a);
            }
            acc += (double)a;
        }
        const double tot = v2w_block_sum(acc, red);
        if (threadIdx.x == 0) dwf[(size_t)t * C + ci] = (float)tot;
        __syncthreads();
    }
}

}  // namespace

static int disc_dz_launch(bool merge, const float* f, const float* g, const float* d, float* dz, float* rowsum, long rows, int C, int Cg,
                          int inner, int s, int dpitch, int pitch, int valid, float slope, hipStream_t st) {
    const int rpb = pitch > 256 ? 1 : 4;
    const long blocks = (rows + rpb - 1) / rpb;
    if (blocks > 0x7fffffffL) return V2W_E_SHAPE;
    if (merge)
        V2W_LAUNCH(disc_dz_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, f, g, d, dz, rowsum, rows, C, Cg, inner, s,
                           dpitch, pitch, valid, slope, rpb);
    else
        V2W_LAUNCH(disc_dz_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, f, g, d, dz, rowsum, rows, C, Cg, inner, s,
                           dpitch, pitch, valid, slope, rpb);
    return v2w_launch_status();
}

// rowsum (optional, rows floats): the sum of every dz row - the bias gradient is v2w_rowsum_reduce over the batch items.
extern "C" int v2w_disc_dz(const float* f, const float* g, const float* d, float* dz, float* rowsum, long long rows, int pitch, int valid,
                           float slope, void* stream) {
    if (!f || !dz || rows <= 0 || pitch <= 0 || valid < 0 || valid > pitch || slope <= 0.f) return V2W_E_ARG;
    return disc_dz_launch(false, f, g, d, dz, rowsum, (long)rows, 1, 1, 1, 1, 0, pitch, valid, slope, (hipStream_t)stream);
}

extern "C" int v2w_phase_merge(const float* dxs, float* out, int B, int C, int Cg, int L, int inner, int s, int ipitch, int opitch,
                               void* stream) {
    if (!dxs || !out || B <= 0 || C <= 0 || Cg <= 0 || C % Cg || L <= 0 || inner <= 0 || s <= 0) return V2W_E_ARG;
    const int U = (L + s - 1) / s;
    if (ipitch <= 0) ipitch = U * inner;
    if (opitch <= 0) opitch = L * inner;
    if (ipitch < U * inner || opitch < L * inner) return V2W_E_ARG;
    const size_t total = (size_t)C * L * inner;
    int gx = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    V2W_LAUNCH(phase_merge_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dxs, out, C, Cg, L, inner, s, ipitch, opitch);
    return v2w_launch_status();
}

extern "C" int v2w_fold1(const float* dxu, float* dx, int B, int T, int H, int inner, int s, int k, int pad, int rows, int ipitch,
                         void* stream) {
    if (!dxu || !dx || B <= 0 || T <= 1 || H <= 0 || inner <= 0 || s <= 0 || k <= 0 || pad < 0 || rows < k) return V2W_E_ARG;
    if ((long long)H * inner < T || (long long)H * inner - T >= T || H + 2 * pad < k) return V2W_E_ARG;
    const int U = (H + 2 * pad - k) / s + 1;
    if (ipitch <= 0) ipitch = U * inner;
    if (ipitch < U * inner) return V2W_E_ARG;
    int gx = (T + 255) / 256; if (gx > 4096) gx = 4096;
    V2W_LAUNCH(fold1_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dxu, dx, T, H, inner, s, k, pad, rows, U, ipitch);
    return v2w_launch_status();
}

extern "C" int v2w_avgpool4_bwd(const float* dout, float* dx, int B, int L, void* stream) {
    if (!dout || !dx || B <= 0 || L <= 0) return V2W_E_ARG;
    int gx = (L + 255) / 256; if (gx > 4096) gx = 4096;
    V2W_LAUNCH(avgpool4_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dout, dx, L, L / 2 + 1);
    return v2w_launch_status();
}

extern "C" int v2w_cout1_wgrad(const float* x, const float* dz, float* dwf, int B, int C, int L, int k, int dil, int tap0, void* stream) {
    if (!x || !dz || !dwf || B <= 0 || C <= 0 || L <= 0 || k <= 0 || dil <= 0 || tap0 < 0 || tap0 >= k) return V2W_E_ARG;
    V2W_LAUNCH(cout1_wgrad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, dz, dwf, B, C, L, k, dil, tap0);
    return v2w_launch_status();
}


// v2w_disc_dz with d given in the phase-stacked form of the strided layer above (v2w_phase_merge folded in): f, dz (B, C, pitch),
// g dense (B, C, L*inner) or NULL, dxs (B, s*C, dpitch).
extern "C" int v2w_disc_dz_merge(const float* f, const float* g, const float* dxs, float* dz, float* rowsum, int B, int C, int Cg, int L,
                                 int inner, int s, int dpitch, int pitch, float slope, void* stream) {
    if (!f || !dxs || !dz || B <= 0 || C <= 0 || Cg <= 0 || C % Cg || L <= 0 || inner <= 0 || s <= 0 || slope <= 0.f) return V2W_E_ARG;
    const int U = (L + s - 1) / s;
    if (pitch < L * inner || dpitch < U * inner) return V2W_E_ARG;
    return disc_dz_launch(true, f, g, dxs, dz, rowsum, (long)B * C, C, Cg, inner, s, dpitch, pitch, L * inner, slope, (hipStream_t)stream);
}

// db[c] = sum over the B batch items of rowsum[b][c] (fp64, fixed order): the bias gradient from v2w_disc_dz's row sums
extern "C" int v2w_rowsum_reduce(const float* rowsum, float* db, int B, int C, void* stream) {
    if (!rowsum || !db || B <= 0 || C <= 0) return V2W_E_ARG;
    V2W_LAUNCH(rowsum_reduce_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, rowsum, db, B, C);
    return v2w_launch_status();
}
