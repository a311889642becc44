// The whole ResBlock2 residual section of an 8-channel stage in one kernel, fp32, on the vector ALU (reference: vec2wav/models.py:135-141;
// the sixth stage of a x640 generator, `upsample_rates` (5, 4, 4, 2, 2, 2), has 512 / 2^6 = 8 channels - below every MFMA tile):
//   out = ( sum_j [ t1_j + conv_{k_j, dil2_j}(lrelu(t1_j)) + b2_j ] ) / out_div,   t1_j = x + conv_{k_j, dil1_j}(lrelu(x)) + b1_j,   x = a * in + s.
// Per layer (conv1d_small_kernel, v2w_direct.hip) the stage costs 12 launches and 1.13 ms at B = 16 x 163 840 positions for 84 MB tensors;
// here x is read once, every t1_j stays in LDS and the branch sum in registers.  A thread owns 2 window columns and all 8 channels of them:
// an activated value read from LDS feeds 8 FMAs, the 8 weights of a (tap, input channel) are one uniform s_load_dwordx8 shared by both
// columns.  Any odd kernel sizes / dilations with halos <= 32 (run-time loops), up to 4 branches.
#include "v2w_common.h"

namespace {

constexpr int SS_C = 8, SS_W = 512, SS_NTH = 256, SS_HMAX = 32;

struct SmallStageArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* w1[4]; const float* bias1[4]; const float* w2[4]; const float* bias2[4];
    int K[4], d1[4], d2[4];
    float* out;
    int nk, B, L, h1max, h2max, nto, ntl;
    float slope, out_div;
};

__global__ void __launch_bounds__(SS_NTH)
small_stage_kernel(const SmallStageArgs a) {
    constexpr int C = SS_C, W = SS_W;
    extern __shared__ float smem_s[];
    const int h1max = a.h1max, h2max = a.h2max, XR = W + 2 * h1max;
    float* const X = smem_s;                 // [C][XR]  lrelu(x): row r <-> position n0 - h2max - h1max + r
    float* const R = X + C * XR;             // [C][W]   x itself (the residual) at the window columns
    float* const T = R + C * W;              // [C][W]   lrelu(t1_j) at the window columns
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.ntl, n0 = (blockIdx.x % a.ntl) * a.nto;
    const int L = a.L;
    const float slope = a.slope;

    // ---- stage x: coalesced along positions, one channel row after the other
    const int pos0 = n0 - h2max - h1max;
    for (int c = 0; c < C; ++c) {
        const float av = a.in_a ? a.in_a[b * C + c] : 1.f, sv = a.in_a ? a.in_s[b * C + c] : 0.f;
        const float* src = a.in + ((size_t)b * C + c) * L;
        for (int r = tid; r < XR; r += SS_NTH) {
            const int pos = pos0 + r;
            const bool in = pos >= 0 && pos < L;
            const float x = in ? fmaf(av, src[in ? pos : 0], sv) : 0.f;        // (the padding of the ACTIVATED signal is exactly 0)
            X[c * XR + r] = v2w_lrelu(x, slope);
            const int col = r - h1max;
            if (col >= 0 && col < W) R[c * W + col] = x;
        }
    }
    __syncthreads();

    float oacc[2][C];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int co = 0; co < C; ++co) {
            float s = 0.f;
            for (int j = 0; j < a.nk; ++j) s += a.bias2[j] ? a.bias2[j][co] : 0.f;
            oacc[q][co] = s;
        }
    const int col0 = tid, col1 = tid + SS_NTH;               // this thread's two window columns
    for (int j = 0; j < a.nk; ++j) {
        const int K = a.K[j], d1 = a.d1[j], d2 = a.d2[j];
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        // ---- conv1_j at both columns (window column c <-> X row c + h1max)
        float acc[2][C];
#pragma unroll
        for (int co = 0; co < C; ++co) acc[0][co] = acc[1][co] = a.bias1[j] ? a.bias1[j][co] : 0.f;
        for (int t = 0; t < K; ++t) {
            const int r0 = col0 + h1max - h1 + t * d1;
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                const float x0 = X[ci * XR + r0], x1 = X[ci * XR + r0 + SS_NTH];
                const float* w = a.w1[j] + ((size_t)t * C + ci) * C;            // (uniform: one s_load_dwordx8)
#pragma unroll
                for (int co = 0; co < C; ++co) { acc[0][co] = fmaf(w[co], x0, acc[0][co]); acc[1][co] = fmaf(w[co], x1, acc[1][co]); }
            }
        }
        if (j > 0) __syncthreads();                          // conv2 of the previous branch has read T
        // ---- t1 = acc + x; the running output takes t1, the tile lrelu(t1) (0 outside the sequence: conv2 zero-pads t1)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = q ? col1 : col0;
            const int pos = n0 - h2max + col;
            const bool in = pos >= 0 && pos < L;
#pragma unroll
            for (int co = 0; co < C; ++co) {
                const float t1 = in ? acc[q][co] + R[co * W + col] : 0.f;
                oacc[q][co] += t1;
                T[co * W + col] = v2w_lrelu(t1, slope);
            }
        }
        __syncthreads();
        // ---- conv2_j (taps of columns outside the valid range are clamped to the tile: those columns are not stored)
        for (int t = 0; t < K; ++t) {
            const int o = t * d2 - h2;
            const int r0 = min(max(col0 + o, 0), W - 1), r1 = min(max(col1 + o, 0), W - 1);
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                const float x0 = T[ci * W + r0], x1 = T[ci * W + r1];
                const float* w = a.w2[j] + ((size_t)t * C + ci) * C;
#pragma unroll
                for (int co = 0; co < C; ++co) { oacc[0][co] = fmaf(w[co], x0, oacc[0][co]); oacc[1][co] = fmaf(w[co], x1, oacc[1][co]); }
            }
        }
    }
    // ---- the nto valid columns (h2max .. h2max + nto): consecutive lanes = consecutive positions of a channel row
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int col = q ? col1 : col0;
        const int v = col - h2max, pos = n0 + v;
        if (v < 0 || v >= a.nto || pos >= L) continue;
#pragma unroll
        for (int co = 0; co < C; ++co) {
            float y = oacc[q][co];
            if (a.out_div != 0.f) y = y / a.out_div;
            a.out[((size_t)b * C + co) * L + pos] = y;
        }
    }
}

}  // namespace

// wf1[j] / wf2[j]: the folded weights [k][C][C] (v2w_wn_fold_conv layout) of branch j's two convs - NOT a packed fragment stream.
// V2W_E_SHAPE: other channel counts, even kernel sizes, halos > 32.
extern "C" int v2w_resblock2_stage_small_fwd(const v2w_stage_args* q, void* stream) {
    if (!q || !q->in || !q->out || q->nk < 1 || q->nk > 4 || q->B <= 0 || q->L <= 0) return V2W_E_ARG;
    if ((q->in_a == nullptr) != (q->in_s == nullptr)) return V2W_E_ARG;
    if (q->C != SS_C) return V2W_E_SHAPE;
    SmallStageArgs p{};
    p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.out = q->out;
    p.nk = q->nk; p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div;
    for (int j = 0; j < q->nk; ++j) {
        if (!q->wp1[j] || !q->wp2[j] || q->k[j] <= 0 || q->dil1[j] <= 0 || q->dil2[j] <= 0) return V2W_E_ARG;
        if ((q->k[j] & 1) == 0) return V2W_E_SHAPE;
        p.w1[j] = q->wp1[j]; p.bias1[j] = q->bias1[j]; p.w2[j] = q->wp2[j]; p.bias2[j] = q->bias2[j];
        p.K[j] = q->k[j]; p.d1[j] = q->dil1[j]; p.d2[j] = q->dil2[j];
        const int h1 = q->dil1[j] * (q->k[j] - 1) / 2, h2 = q->dil2[j] * (q->k[j] - 1) / 2;
        if (h1 > p.h1max) p.h1max = h1;
        if (h2 > p.h2max) p.h2max = h2;
    }
    if (p.h1max > SS_HMAX || p.h2max > SS_HMAX) return V2W_E_SHAPE;
    p.nto = SS_W - 2 * p.h2max;
    p.ntl = (q->L + p.nto - 1) / p.nto;
    if ((long long)q->B * p.ntl > 0x7fffffffll) return V2W_E_SHAPE;
    const size_t lds = (size_t)SS_C * (SS_W + 2 * p.h1max + 2 * SS_W) * sizeof(float);
    V2W_LAUNCH(small_stage_kernel, dim3(q->B * p.ntl), dim3(SS_NTH), lds, (hipStream_t)stream, p);
    return v2w_launch_status();
}
