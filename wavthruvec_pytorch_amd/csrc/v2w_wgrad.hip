// Weight gradient of Conv1d / ConvTranspose1d on the f32 MFMA pipe (backward of the generator path).
//
//   dW[t][ci][co] = sum_{b,q}  act(x)[b, ci, q + off_t] * dy[b, co, u*q + r_t]
//     Conv1d            : u = 1, r_t = 0, off_t = (t - (k-1)/2) * dil                       (forward: models.py:37-44,65-70,123)
//     ConvTranspose1d   : r_t = (t - pad) mod u, off_t = (r_t + pad)/u - (t - (r_t + pad)%u)/u   (forward: models.py:128-129)
//   act(x) = leaky_relu(a*x + s): the same activated signal the forward conv consumed.
//
// GEMM view: the REDUCTION runs over positions.  Per workgroup: a (ci-tile x co-tile) block of dW for up to TG taps;
// A[i = ci][k = position] = act(x) tile (one A per tap: the tap is a column offset), B[k = position][j = co] = dy tile shared
// by the taps.  D[ci][co] keeps co on the lanes, i.e. contiguous in the [k][C_in][C_out] weight layout.
// The position axis is split over S workgroup columns (x WP waves); every split writes its partial block into its own
// slab and a second kernel sums the slabs in fixed order: deterministic, no atomics.
#include "v2w_common.h"

namespace {

#define V2W_WG_TG 6   // taps accumulated per launch (TG * 16 accumulator registers)

struct WgradArgs {
    const float* x; const float* x_a; const float* x_s;   // (B, Cin, Lq) and its per-(b,ci) affine
    const float* dy;                                       // (B, Cout, Ldy), Ldy = u * Lq
    float* slab;                                           // [S*WP][K][Cin][Cout]
    int B, Cin, Cout, Lq, K;
    int u, r;                 // dy position = u*q + r
    int ntap;                 // taps of this launch
    int tap[V2W_WG_TG];       // real tap index (slab row)
    int off[V2W_WG_TG];       // signal offset of the tap
    int hl, hr;               // max(0, -min off), max(0, max off)
    int wco, wci, wp;         // wave arrangement (wco*wci*wp == 4)
    int S;                    // position splits (grid.y)
    int nchunk;               // position chunks per batch item
    int ptw, xtw;             // LDS row strides
    float slope;
};

template <int MF>
__global__ void __launch_bounds__(256)
wgrad_kernel(const WgradArgs p) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int KSTEP = F::KSTEP;          // positions per MFMA: 2 (MF = 32) or 4 (MF = 16)
    constexpr int PT = 128;                  // positions per staged chunk
    extern __shared__ float smem[];
    const int CO_T = p.wco * MF, CI_T = p.wci * MF;
    float* const DYs = smem;                 // [CO_T][ptw]
    float* const Xas = smem + CO_T * p.ptw;  // [CI_T][xtw]

    const int cot = p.Cout / CO_T;
    const int co0 = (blockIdx.x % cot) * CO_T, ci0 = (blockIdx.x / cot) * CI_T;
    const int s = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int w_p = wave % p.wp, w_ci = (wave / p.wp) % p.wci, w_co = wave / (p.wp * p.wci);
    const int Lq = p.Lq, Ldy = p.Lq * p.u;

    acc_t acc[V2W_WG_TG];
#pragma unroll
    for (int t = 0; t < V2W_WG_TG; ++t)
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) acc[t][e] = 0.f;

    const int items = p.B * p.nchunk;
    const int per = (items + p.S - 1) / p.S;
    const int it0 = s * per, it1 = min(items, it0 + per);
    const int sub = PT / p.wp;               // positions of a chunk each wave reduces
    const int xcols = PT + p.hl + p.hr;

    for (int it = it0; it < it1; ++it) {
        const int b = it / p.nchunk, q0 = (it % p.nchunk) * PT;
        __syncthreads();
        // ---- stage dy (phase r of the stride-u grid) and the activated signal, zero outside [0, Lq)
        for (int idx = tid; idx < CO_T * PT; idx += 256) {
            const int row = idx / PT, col = idx - row * PT;
            const int q = q0 + col;
            DYs[row * p.ptw + col] = q < Lq ? p.dy[((size_t)b * p.Cout + co0 + row) * Ldy + (size_t)p.u * q + p.r] : 0.f;
        }
        for (int idx = tid; idx < CI_T * xcols; idx += 256) {
            const int row = idx / xcols, col = idx - row * xcols;
            const int q = q0 - p.hl + col;
            const int ch = b * p.Cin + ci0 + row;
            float v = 0.f;
            if (q >= 0 && q < Lq) {
                const float av = p.x_a ? p.x_a[ch] : 1.f, sv = p.x_a ? p.x_s[ch] : 0.f;
                v = v2w_lrelu(fmaf(av, p.x[(size_t)ch * Lq + q], sv), p.slope);
            }
            Xas[row * p.xtw + col] = v;
        }
        __syncthreads();
        const float* brow = DYs + (w_co * MF + lr) * p.ptw + w_p * sub + hk;
        const float* arow = Xas + (w_ci * MF + lr) * p.xtw + w_p * sub + hk + p.hl;
        for (int kq = 0; kq < sub; kq += KSTEP) {
            const float bv = brow[kq];
#pragma unroll
            for (int t = 0; t < V2W_WG_TG; ++t)
                if (t < p.ntap) acc[t] = F::mfma(arow[kq + p.off[t]], bv, acc[t]);
        }
    }

    // ---- partial block of this split -> its slab: rows = ci (accumulator rows), lanes = co (contiguous)
    const int slab_id = s * p.wp + w_p;
    float* dst = p.slab + (size_t)slab_id * p.K * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < V2W_WG_TG; ++t) {
        if (t >= p.ntap) continue;
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int ci = ci0 + w_ci * MF + F::row(e, hk);
            const int co = co0 + w_co * MF + lr;
            dst[((size_t)p.tap[t] * p.Cin + ci) * p.Cout + co] = acc[t][e];
        }
    }
}

// dwf[i] = sum_s slab[s][i]   (fixed order)
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dwf, size_t n, int nslab) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < nslab; ++s) v += slab[(size_t)s * n + i];
        dwf[i] = v;
    }
}

}  // namespace

// Number of partial slabs ([k][C_in][C_out] floats each) v2w_wgrad needs for this problem; 0 = shape not supported.
extern "C" int v2w_wgrad_slabs(int B, int c_in, int c_out, int Lq) {
    const int mf = (c_out % 32 == 0 && c_in % 32 == 0) ? 32 : ((c_out % 16 == 0 && c_in % 16 == 0) ? 16 : 0);
    if (!mf) return 0;
    int wco = 1, wci = 1;
    if (c_out % (2 * mf) == 0) wco = 2;
    if (c_in % (2 * mf) == 0 && wco * 2 <= 4) wci = 2;
    const int wp = 4 / (wco * wci);
    const int tiles = (c_out / (wco * mf)) * (c_in / (wci * mf));
    const int items = B * ((Lq + 127) / 128);
    int S = (3 * 256 + tiles - 1) / tiles;
    if (S > items) S = items;
    if (S < 1) S = 1;
    return S * wp;
}

// dwf [k][C_in][C_out] = weight gradient; u = 1 / pad ignored for Conv1d (dil used), stride u and pad = (k-u)/2 for ConvTranspose1d.
extern "C" int v2w_wgrad(const float* x, const float* x_a, const float* x_s, const float* dy, float* dwf, float* slab_ws,
                         int B, int c_in, int c_out, int Lq, int k, int dil, int u, float slope, void* stream) {
    if (!x || !dy || !dwf || !slab_ws || B <= 0 || c_in <= 0 || c_out <= 0 || Lq <= 0 || k <= 0 || dil <= 0 || u <= 0) return V2W_E_ARG;
    if ((x_a == nullptr) != (x_s == nullptr)) return V2W_E_ARG;
    const int nslab = v2w_wgrad_slabs(B, c_in, c_out, Lq);
    if (!nslab) return V2W_E_SHAPE;
    const int mf = (c_out % 32 == 0 && c_in % 32 == 0) ? 32 : 16;
    WgradArgs p{};
    p.x = x; p.x_a = x_a; p.x_s = x_s; p.dy = dy; p.slab = slab_ws;
    p.B = B; p.Cin = c_in; p.Cout = c_out; p.Lq = Lq; p.K = k; p.u = u; p.slope = slope;
    p.wco = (c_out % (2 * mf) == 0) ? 2 : 1;
    p.wci = (c_in % (2 * mf) == 0 && p.wco * 2 <= 4) ? 2 : 1;
    p.wp = 4 / (p.wco * p.wci);
    p.S = nslab / p.wp;
    p.nchunk = (Lq + 127) / 128;
    const int tiles = (c_out / (p.wco * mf)) * (c_in / (p.wci * mf));
    const int pad = u > 1 ? (k - u) / 2 : 0;
    hipStream_t st = (hipStream_t)stream;
    // taps grouped by dy phase (conv: one phase), at most V2W_WG_TG taps per launch
    for (int r = 0; r < u; ++r) {
        int taps[64], offs[64], n = 0;
        for (int t = 0; t < k && n < 64; ++t) {
            if (u == 1) { taps[n] = t; offs[n] = (t - (k - 1) / 2) * dil; ++n; }
            else if (((t - pad) % u + u) % u == r) {
                const int t0 = (r + pad) % u, c = (r + pad) / u, m = (t - t0) / u;
                taps[n] = t; offs[n] = c - m; ++n;
            }
        }
        for (int g0 = 0; g0 < n; g0 += V2W_WG_TG) {
            p.r = r;
            p.ntap = n - g0 < V2W_WG_TG ? n - g0 : V2W_WG_TG;
            int lo = 0, hi = 0;
            for (int i = 0; i < p.ntap; ++i) {
                p.tap[i] = taps[g0 + i]; p.off[i] = offs[g0 + i];
                if (p.off[i] < lo) lo = p.off[i];
                if (p.off[i] > hi) hi = p.off[i];
            }
            p.hl = -lo; p.hr = hi;
            int ptw = 128, xtw = 128 + p.hl + p.hr;
            if (mf == 32) { ptw |= 1; xtw |= 1; }                                     // odd stride: 32 rows hit 32 banks
            else { ptw += ((2 - ptw % 32) + 32) % 32; xtw += ((2 - xtw % 32) + 32) % 32; }   // stride = 2 mod 32 (four 16-lane k-groups)
            p.ptw = ptw; p.xtw = xtw;
            const size_t lds = ((size_t)p.wco * mf * ptw + (size_t)p.wci * mf * xtw) * sizeof(float);
            if (lds > 160 * 1024) return V2W_E_SHAPE;
            if (mf == 32) {
                if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(wgrad_kernel<32>, dim3(tiles, p.S), dim3(256), lds, st, p);
            } else {
                if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(wgrad_kernel<16>, dim3(tiles, p.S), dim3(256), lds, st, p);
            }
        }
    }
    const size_t nw = (size_t)k * c_in * c_out;
    int grid = (int)((nw + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, slab_ws, dwf, nw, nslab);
    return v2w_launch_status();
}
