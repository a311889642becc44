// Scalar-FMA kernels: the any-shape path of the conv entry points (V2W_ALGO_DIRECT) and the fused tail
// leaky_relu -> conv_post -> tanh (models.py:143-145).  One thread per output sample, lanes along the
// frame axis so every global access is coalesced.
#include "v2w_common.h"

namespace {

__global__ void __launch_bounds__(256)
conv1d_direct_kernel(const v2w_conv1d_args a) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int co = blockIdx.y, b = blockIdx.z;
    if (l >= a.L) return;
    const int pad = a.pad_left >= 0 ? a.pad_left : a.dil * (a.k - 1) / 2;
    const int istr = a.in_stride > 0 ? a.in_stride : 1;
    const int cit = a.in_ct > 0 ? a.in_ct : a.C_in, cot = a.out_ct > 0 ? a.out_ct : a.C_out;
    float acc = 0.f;
    for (int ci = 0; ci < a.C_in; ++ci) {
        const float* src = a.in + ((size_t)b * cit + ci) * a.L * istr + a.in_phase;
        const float av = a.in_a ? a.in_a[b * a.C_in + ci] : 1.f;
        const float sv = a.in_s ? a.in_s[b * a.C_in + ci] : 0.f;
        for (int t = 0; t < a.k; ++t) {
            const int li = l + t * a.dil - pad;
            if (li < 0 || li >= a.L) continue;
            const float x = v2w_lrelu(fmaf(av, src[(size_t)li * istr], sv), a.slope);
            acc = fmaf(a.wf[((size_t)t * a.C_in + ci) * a.C_out + co], x, acc);
        }
    }
    const size_t o = ((size_t)b * cot + co) * a.L + l;
    float v = acc;
    if (a.mask_src) {
        const float ma = a.mask_a ? a.mask_a[b * a.C_out + co] : 1.f, ms = a.mask_a ? a.mask_s[b * a.C_out + co] : 0.f;
        v = fmaf(ma, a.mask_src[o], ms) > 0.f ? v : v * a.mask_slope;
    }
    v += a.bias ? a.bias[co] : 0.f;
    if (a.res) {
        const float ra = a.res_a ? a.res_a[b * a.C_out + co] : 1.f;
        const float rs = a.res_s ? a.res_s[b * a.C_out + co] : 0.f;
        v += fmaf(ra, a.res[o], rs);
    }
    if (a.add1) v += a.add0[o] + a.add1[o];
    else if (a.add0) v += a.add0[o];
    else if (a.accumulate) v += a.out[o];
    if (a.out_div != 0.f) v = v / a.out_div;
    if (a.out_slope != 0.f && a.out_slope != 1.f) v = v > 0.f ? v : v * a.out_slope;
    a.out[o] = v;
}

// Few output channels (C_out <= 8: the last stage of a six-stage x640 generator has 8 channels, below every MFMA tile): one thread per
// position computes ALL CO output channels - the activated input value is loaded once per (channel, tap) and feeds CO FMAs, the CO weights
// of a (tap, channel) are one uniform (scalar) load.  conv1d_direct_kernel launches a block row per output channel instead: C_out times the
// input reads and a global weight load per FMA (0.9 - 1.6 ms per conv of the 8-channel stage at B = 16 x 163 840 positions, against 84 MB of
// tensor).  Same arguments and flags.
// EXACT: C_out == CO (no per-channel guards: the CO weights of a (tap, channel) are ONE s_load_dwordx8; with guards hipcc emitted a branch,
// a one-dword scalar load and a wait per FMA).
template <int CO, bool EXACT>
__global__ void __launch_bounds__(256)
conv1d_small_kernel(const v2w_conv1d_args a) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (l >= a.L) return;
    const int pad = a.pad_left >= 0 ? a.pad_left : a.dil * (a.k - 1) / 2;
    const int istr = a.in_stride > 0 ? a.in_stride : 1;
    const int cit = a.in_ct > 0 ? a.in_ct : a.C_in, cot = a.out_ct > 0 ? a.out_ct : a.C_out;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
    for (int ci = 0; ci < a.C_in; ++ci) {
        const float* src = a.in + ((size_t)b * cit + ci) * a.L * istr + a.in_phase;
        const float av = a.in_a ? a.in_a[b * a.C_in + ci] : 1.f;
        const float sv = a.in_s ? a.in_s[b * a.C_in + ci] : 0.f;
#pragma unroll 4
        for (int t = 0; t < a.k; ++t) {
            const int li = l + t * a.dil - pad;
            const bool in = li >= 0 && li < a.L;
            const float raw = src[(size_t)min(max(li, 0), a.L - 1) * istr];      // (clamped, unconditional: no branch around the load)
            const float x = in ? v2w_lrelu(fmaf(av, raw, sv), a.slope) : 0.f;
            const float* w = a.wf + ((size_t)t * a.C_in + ci) * a.C_out;          // (uniform: scalar loads)
#pragma unroll
            for (int co = 0; co < CO; ++co)
                if (EXACT || co < a.C_out) acc[co] = fmaf(w[co], x, acc[co]);
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (!EXACT && co >= a.C_out) break;
        const size_t o = ((size_t)b * cot + co) * a.L + l;
        float v = acc[co];
        if (a.mask_src) {
            const float ma = a.mask_a ? a.mask_a[b * a.C_out + co] : 1.f, ms = a.mask_a ? a.mask_s[b * a.C_out + co] : 0.f;
            v = fmaf(ma, a.mask_src[o], ms) > 0.f ? v : v * a.mask_slope;
        }
        v += a.bias ? a.bias[co] : 0.f;
        if (a.res) {
            const float ra = a.res_a ? a.res_a[b * a.C_out + co] : 1.f;
            const float rs = a.res_s ? a.res_s[b * a.C_out + co] : 0.f;
            v += fmaf(ra, a.res[o], rs);
        }
        if (a.add1) v += a.add0[o] + a.add1[o];
        else if (a.add0) v += a.add0[o];
        else if (a.accumulate) v += a.out[o];
        if (a.out_div != 0.f) v = v / a.out_div;
        if (a.out_slope != 0.f && a.out_slope != 1.f) v = v > 0.f ? v : v * a.out_slope;
        a.out[o] = v;
    }
}

// Transposed counterpart: one thread per INPUT position q computes its u output positions u q .. u q + u - 1 of all CO channels (u <= 4):
// the taps of a phase are uniform, the u outputs of a channel leave as one 8- / 16-byte store.
template <int CO, int U, bool EXACT>
__global__ void __launch_bounds__(256)
convt1d_small_kernel(const v2w_convt1d_args a) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= a.L) return;
    const int pad = (a.k - U) / 2;
    float acc[CO][U];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int r = 0; r < U; ++r) acc[co][r] = 0.f;
    for (int ci = 0; ci < a.C_in; ++ci) {
        const float* src = a.in + ((size_t)b * a.C_in + ci) * a.L;
        // out[U q + r] = sum_t in[(U q + r + pad - t) / U] w[t] over the taps t = (r + pad) mod U, + U, ...: input offset c - m, c = (r + pad) / U
#pragma unroll
        for (int r = 0; r < U; ++r) {
            const int t0 = (r + pad) % U, c = (r + pad) / U;
            for (int m = 0; t0 + m * U < a.k; ++m) {
                const int i = q + c - m;
                const bool in = i >= 0 && i < a.L;
                const float raw = src[min(max(i, 0), a.L - 1)];
                const float x = in ? v2w_lrelu(raw, a.slope) : 0.f;
                const float* w = a.wf + ((size_t)(t0 + m * U) * a.C_in + ci) * a.C_out;
#pragma unroll
                for (int co = 0; co < CO; ++co)
                    if (EXACT || co < a.C_out) acc[co][r] = fmaf(w[co], x, acc[co][r]);
            }
        }
    }
    const size_t Lout = (size_t)a.L * U;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (!EXACT && co >= a.C_out) break;
        const float bv = a.bias ? a.bias[co] : 0.f;
        float* dst = a.out + ((size_t)b * a.C_out + co) * Lout + (size_t)U * q;
        if constexpr (U == 2) *reinterpret_cast<f32x2*>(dst) = f32x2{acc[co][0] + bv, acc[co][1] + bv};
        else *reinterpret_cast<f32x4*>(dst) = f32x4{acc[co][0] + bv, acc[co][1] + bv, acc[co][2] + bv, acc[co][3] + bv};
    }
}

// C_out = 1 (the discriminators' conv_post, models.py:171,230): a reduction over C_in * k, HBM-bound on reading the input once.
// One block = 32 output positions of one batch item; 8 channel groups of 32 lanes split the input channels (4 independent
// accumulation chains each) and meet in LDS.
__global__ void __launch_bounds__(256)
conv1d_cout1_kernel(const v2w_conv1d_args a) {
    __shared__ float red[8][32];
    const int lane = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int l = blockIdx.x * 32 + lane, b = blockIdx.y;
    const int pad = a.pad_left >= 0 ? a.pad_left : a.dil * (a.k - 1) / 2;
    const int cit = a.in_ct > 0 ? a.in_ct : a.C_in;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (l < a.L) {
        for (int t = 0; t < a.k; ++t) {
            const int li = l + t * a.dil - pad;
            if (li < 0 || li >= a.L) continue;
            const float* src = a.in + (size_t)b * cit * a.L + li;
            const float* w = a.wf + (size_t)t * a.C_in;
            int ci = cg;
            for (; ci + 24 < a.C_in; ci += 32) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[u] = fmaf(w[ci + 8 * u], v2w_lrelu(src[(size_t)(ci + 8 * u) * a.L], a.slope), acc[u]);
            }
            for (; ci < a.C_in; ci += 8) acc[0] = fmaf(w[ci], v2w_lrelu(src[(size_t)ci * a.L], a.slope), acc[0]);
        }
    }
    red[cg][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (cg == 0 && l < a.L) {
        float v = a.bias ? a.bias[0] : 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += red[g][lane];
        if (a.out_slope != 0.f && a.out_slope != 1.f) v = v > 0.f ? v : v * a.out_slope;
        a.out[((size_t)b * (a.out_ct > 0 ? a.out_ct : 1)) * a.L + l] = v;
    }
}

__global__ void __launch_bounds__(256)
convt1d_direct_kernel(const v2w_convt1d_args a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;   // output position
    const int co = blockIdx.y, b = blockIdx.z;
    const int Lout = a.L * a.u;
    if (n >= Lout) return;
    const int pad = (a.k - a.u) / 2;
    // out[n] = sum_{i,t : i*u - pad + t = n} in[i] * w[t]   (torch ConvTranspose1d)
    const int t0 = (n + pad) % a.u;
    float acc = 0.f;
    for (int ci = 0; ci < a.C_in; ++ci) {
        const float* src = a.in + ((size_t)b * a.C_in + ci) * a.L;
        for (int t = t0; t < a.k; t += a.u) {
            const int i = (n + pad - t) / a.u;
            if (n + pad - t < 0 || i >= a.L) continue;
            const float x = v2w_lrelu(src[i], a.slope);
            acc = fmaf(a.wf[((size_t)t * a.C_in + ci) * a.C_out + co], x, acc);
        }
    }
    a.out[((size_t)b * a.C_out + co) * Lout + n] = acc + (a.bias ? a.bias[co] : 0.f);
}

// bf16 activation storage (BASELINE configs[2]): the tail can read a bf16 input tensor; everything after the load is fp32
struct InF32 { typedef float elem_t; static __device__ __forceinline__ float get(const float* p, size_t i) { return p[i]; }
               static __device__ __forceinline__ f32x4 get4(const float* p, size_t i) { return *reinterpret_cast<const f32x4*>(p + i); } };
struct InBf16 { typedef unsigned short elem_t;
                static __device__ __forceinline__ float get(const unsigned short* p, size_t i) { return __builtin_bit_cast(float, (unsigned)p[i] << 16); }
                static __device__ __forceinline__ f32x4 get4(const unsigned short* p, size_t i) {
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 w = *reinterpret_cast<const u32x2*>(p + i);
                    return f32x4{__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                                 __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)}; } };

// Tail: each thread produces 4 consecutive samples from a register window of 4 + (k-1) inputs per channel.
// HBM-bound (3.3 FLOP/B): the input is read once (neighbour overlap is served by L1/L2), the output written once.
template <int KMAX, typename IN = InF32>
__global__ void __launch_bounds__(256)
conv_post_tanh_kernel(const typename IN::elem_t* __restrict__ in, const float* __restrict__ wf, const float* __restrict__ bias,
                      float* __restrict__ out, int B, int Cin, int L, int k, float slope) {
    extern __shared__ float w_s[];   // [k][Cin]
    for (int i = threadIdx.x; i < k * Cin; i += blockDim.x) w_s[i] = wf[i];
    __syncthreads();
    const int pad = (k - 1) / 2;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (l0 >= L) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < Cin; ++ci) {
        const size_t src = ((size_t)b * Cin + ci) * L;
        float win[4 + KMAX - 1];
#pragma unroll
        for (int j = 0; j < 4 + KMAX - 1; ++j) {
            const int li = l0 - pad + j;
            win[j] = (j < 4 + k - 1 && li >= 0 && li < L) ? v2w_lrelu(IN::get(in, src + li), slope) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t < k) {
                const float w = w_s[t * Cin + ci];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = fmaf(w, win[o + t], acc[o]);
            }
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    float* dst = out + (size_t)b * L + l0;
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (l0 + o < L) dst[o] = tanhf(acc[o] + bv);
}

// Vectorised tail for L % 4 == 0 and k <= 9: per channel three aligned float4 loads cover the 4 outputs' window
// [l0-4, l0+8); every load is a full 16 B/lane coalesced access.
template <typename IN = InF32>
__global__ void __launch_bounds__(256)
conv_post_tanh_vec4_kernel(const typename IN::elem_t* __restrict__ in, const float* __restrict__ wf, const float* __restrict__ bias,
                           float* __restrict__ out, int B, int Cin, int L, int k, float slope) {
    extern __shared__ float w_s[];   // [k][Cin]
    for (int i = threadIdx.x; i < k * Cin; i += blockDim.x) w_s[i] = wf[i];
    __syncthreads();
    const int pad = (k - 1) / 2;
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (l0 >= L) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ci = 0; ci < Cin; ++ci) {
        const size_t src = ((size_t)b * Cin + ci) * L + l0;
        const f32x4 lo = l0 >= 4 ? IN::get4(in, src - 4) : zero4;
        const f32x4 mid = IN::get4(in, src);
        const f32x4 hi = l0 + 4 < L ? IN::get4(in, src + 4) : zero4;
        float win[12];
#pragma unroll
        for (int e = 0; e < 4; ++e) { win[e] = v2w_lrelu(lo[e], slope); win[4 + e] = v2w_lrelu(mid[e], slope); win[8 + e] = v2w_lrelu(hi[e], slope); }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < k) {
                const float w = w_s[t * Cin + ci];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = fmaf(w, win[4 + o + t - pad], acc[o]);   // pad <= 4
            }
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    f32x4 y;
#pragma unroll
    for (int o = 0; o < 4; ++o) y[o] = tanhf(acc[o] + bv);
    *reinterpret_cast<f32x4*>(out + (size_t)b * L + l0) = y;
}

}  // namespace

int v2w_conv1d_direct(const v2w_conv1d_args* a, hipStream_t stream) {
    const int istr = a->in_stride > 0 ? a->in_stride : 1;
    if (a->C_out == 1 && istr == 1 && !a->in_a && !a->res && !a->add0 && !a->mask_src && !a->accumulate && a->out_div == 0.f) {
        V2W_LAUNCH(conv1d_cout1_kernel, dim3((a->L + 31) / 32, a->B), dim3(256), 0, stream, *a);
        return v2w_launch_status();
    }
    if (a->C_out >= 2 && a->C_out <= 8 && a->B <= 65535) {
        if (a->C_out == 8) V2W_LAUNCH((conv1d_small_kernel<8, true>), dim3((a->L + 255) / 256, a->B), dim3(256), 0, stream, *a);
        else V2W_LAUNCH((conv1d_small_kernel<8, false>), dim3((a->L + 255) / 256, a->B), dim3(256), 0, stream, *a);
        return v2w_launch_status();
    }
    dim3 grid((a->L + 255) / 256, a->C_out, a->B);
    V2W_LAUNCH(conv1d_direct_kernel, grid, dim3(256), 0, stream, *a);
    return v2w_launch_status();
}

int v2w_convt1d_direct(const v2w_convt1d_args* a, hipStream_t stream) {
    if (a->C_out >= 2 && a->C_out <= 8 && a->B <= 65535 && (a->u == 2 || a->u == 4) && a->k >= a->u && ((a->k - a->u) & 1) == 0 &&
        (reinterpret_cast<uintptr_t>(a->out) & 15) == 0) {
        const dim3 grid((a->L + 255) / 256, a->B);
        if (a->u == 2 && a->C_out == 8) V2W_LAUNCH((convt1d_small_kernel<8, 2, true>), grid, dim3(256), 0, stream, *a);
        else if (a->u == 2) V2W_LAUNCH((convt1d_small_kernel<8, 2, false>), grid, dim3(256), 0, stream, *a);
        else if (a->C_out == 8) V2W_LAUNCH((convt1d_small_kernel<8, 4, true>), grid, dim3(256), 0, stream, *a);
        else V2W_LAUNCH((convt1d_small_kernel<8, 4, false>), grid, dim3(256), 0, stream, *a);
        return v2w_launch_status();
    }
    dim3 grid((a->L * a->u + 255) / 256, a->C_out, a->B);
    V2W_LAUNCH(convt1d_direct_kernel, grid, dim3(256), 0, stream, *a);
    return v2w_launch_status();
}

template <typename IN>
static int conv_post_tanh_impl(const typename IN::elem_t* in, const float* wf, const float* bias, float* out,
                               int B, int C_in, int L, int k, float slope, void* stream) {
    if (!in || !wf || !out || B <= 0 || C_in <= 0 || L <= 0 || k <= 0 || (k & 1) == 0) return V2W_E_ARG;
    if (k > 15) return V2W_E_SHAPE;
    dim3 grid((L + 1023) / 1024, B);
    const size_t lds = (size_t)k * C_in * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = (L % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (aligned && k <= 9) {
        V2W_LAUNCH(conv_post_tanh_vec4_kernel<IN>, grid, dim3(256), lds, s, in, wf, bias, out, B, C_in, L, k, slope);
        return v2w_launch_status();
    }
    if (k <= 7) V2W_LAUNCH((conv_post_tanh_kernel<7, IN>), grid, dim3(256), lds, s, in, wf, bias, out, B, C_in, L, k, slope);
    else V2W_LAUNCH((conv_post_tanh_kernel<15, IN>), grid, dim3(256), lds, s, in, wf, bias, out, B, C_in, L, k, slope);
    return v2w_launch_status();
}

extern "C" int v2w_conv_post_tanh(const float* in, const float* wf, const float* bias, float* out,
                                  int B, int C_in, int L, int k, float slope, void* stream) {
    return conv_post_tanh_impl<InF32>(in, wf, bias, out, B, C_in, L, k, slope, stream);
}

int v2w_conv_post_tanh_bf16_mfma(const unsigned short* in, const float* wf, const float* bias, float* out,
                                 int B, int C_in, int L, int k, float slope, hipStream_t stream);      // v2w_conv_post_bf16.hip

extern "C" int v2w_conv_post_tanh_bf16in(const void* x_bf16, const float* wf, const float* bias, float* out,
                                         int B, int C_in, int L, int k, float slope, void* stream) {
    if (x_bf16 && wf && out && B > 0) {          // 16 / 8 channels, aligned rows: the taps as a Toeplitz product on the matrix pipe
        const int rc = v2w_conv_post_tanh_bf16_mfma(static_cast<const unsigned short*>(x_bf16), wf, bias, out, B, C_in, L, k, slope, (hipStream_t)stream);
        if (rc != V2W_E_SHAPE) return rc;
    }
    return conv_post_tanh_impl<InBf16>(static_cast<const unsigned short*>(x_bf16), wf, bias, out, B, C_in, L, k, slope, stream);
}
