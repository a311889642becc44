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

#define V2W_WG_TG 7   // taps accumulated per launch (TG * 16 accumulator registers)
#define V2W_WG_XC4_WIDE 56
#define V2W_WG_XC4 40 // float4 columns of the widest signal tile the pipelined kernel stages (128 + 2*16 positions)

struct WgradArgs {
    const float* x; const float* x_a; const float* x_s;   // (B, Cin, Lq) and its per-(b,ci) affine
    const float* dy;                                       // (B, Cout, Ldy), Ldy = u * Lq
    float* slab;                                           // [S*WP][K][Cin][Cout]
    int B, Cin, Cout, Lq, K;
    int CinT, CoutT;          // channels per batch item of the tensors x / dy live in (> Cin / Cout: one group of a grouped conv)
    int ngroups;              // grid.z: group g reads channels [g*Cin, ..) of x, [g*Cout, ..) of dy and owns slab region g
    int nslab;                // slabs per group
    int u, r;                 // dy position = u*q + r
    int ntap;                 // taps of this launch
    int tap[V2W_WG_TG];       // real tap index (slab row)
    int off[V2W_WG_TG];       // signal offset of the tap
    int rr[V2W_WG_TG];        // dy phase of the tap (pipelined kernel; the generic kernel takes one phase `r` per launch)
    int hl, hr;               // max(0, -min off), max(0, max off)
    int hla, xc4, vec4;       // hl rounded up to 4; float4 columns staged; 1 = aligned float4 staging (u == 1, Lq % 4 == 0)
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

    const int cot = (p.Cout + CO_T - 1) / CO_T;          // (channel counts below a multiple of the tile: the missing rows stage as 0)
    const int co0 = (blockIdx.x % cot) * CO_T, ci0 = (blockIdx.x / cot) * CI_T;
    const int s = blockIdx.y;
    const int cob = blockIdx.z * p.Cout + co0, cib = blockIdx.z * p.Cin + ci0;     // channel bases inside the x / dy tensors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int w_p = wave % p.wp, w_ci = (wave / p.wp) % p.wci, w_co = wave / (p.wp * p.wci);
    const int Lq = p.Lq, Ldy = p.Lq * p.u;
    const int co_n = min(CO_T, p.Cout - co0), ci_n = min(CI_T, p.Cin - ci0);       // real rows of this tile

    acc_t acc[V2W_WG_TG];
#pragma unroll
    for (int t = 0; t < V2W_WG_TG; ++t)
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) acc[t][e] = 0.f;

    const int items = p.B * p.nchunk;
    const int per = (items + p.S - 1) / p.S;
    const int it0 = s * per, it1 = min(items, it0 + per);
    const int sub = PT / p.wp;               // positions of a chunk each wave reduces
    const int xcols = p.xc4 * 4;             // staged signal columns: column 0 <-> position q0 - hla

    for (int it = it0; it < it1; ++it) {
        const int b = it / p.nchunk, q0 = (it % p.nchunk) * PT;
        __syncthreads();
        // ---- stage dy (phase r of the stride-u grid) and the activated signal, zero outside [0, Lq)
        if (p.vec4) {
            // aligned float4 global loads (16 B / lane, whole rows coalesced); LDS rows have an odd stride -> dword stores
            for (int idx = tid; idx < CO_T * (PT / 4); idx += 256) {
                const int row = idx >> 5, col = (idx & 31) * 4;           // PT / 4 == 32 float4 per row
                const int q = q0 + col;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < Lq && row < co_n) v = *reinterpret_cast<const f32x4*>(p.dy + ((size_t)b * p.CoutT + cob + row) * Ldy + q);
                float* d = DYs + row * p.ptw + col;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
            const int xc4 = p.xc4;                                         // float4 columns of the signal tile
            const int qa = q0 - p.hla;                                     // position of signal column 0 (multiple of 4)
            for (int idx = tid; idx < CI_T * xc4; idx += 256) {
                const int row = idx / xc4, col = (idx - row * xc4) * 4;
                const int q = qa + col;
                const int ch = b * p.CinT + cib + row;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q >= 0 && q < Lq && row < ci_n) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.x + (size_t)ch * Lq + q);
                    const float av = p.x_a ? p.x_a[ch] : 1.f, sv = p.x_a ? p.x_s[ch] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v2w_lrelu(fmaf(av, g[e], sv), p.slope);
                }
                float* d = Xas + row * p.xtw + col;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        } else {
            for (int idx = tid; idx < CO_T * PT; idx += 256) {
                const int row = idx / PT, col = idx - row * PT;
                const int q = q0 + col;
                DYs[row * p.ptw + col] = (q < Lq && row < co_n) ? p.dy[((size_t)b * p.CoutT + cob + row) * Ldy + (size_t)p.u * q + p.r] : 0.f;
            }
            for (int idx = tid; idx < CI_T * xcols; idx += 256) {
                const int row = idx / xcols, col = idx - row * xcols;
                const int q = q0 - p.hla + col;
                const int ch = b * p.CinT + cib + row;
                float v = 0.f;
                if (q >= 0 && q < Lq && row < ci_n) {
                    const float av = p.x_a ? p.x_a[ch] : 1.f, sv = p.x_a ? p.x_s[ch] : 0.f;
                    v = v2w_lrelu(fmaf(av, p.x[(size_t)ch * Lq + q], sv), p.slope);
                }
                Xas[row * p.xtw + col] = v;
            }
        }
        __syncthreads();
        const float* brow = DYs + (w_co * MF + lr) * p.ptw + w_p * sub + hk;
        const float* arow = Xas + (w_ci * MF + lr) * p.xtw + w_p * sub + hk + p.hla;
#pragma unroll 4
        for (int kq = 0; kq < sub; kq += KSTEP) {
            const float bv = brow[kq];
#pragma unroll
            for (int t = 0; t < V2W_WG_TG; ++t)
                if (t < p.ntap) acc[t] = F::mfma(arow[kq + p.off[t]], bv, acc[t]);
        }
    }

    // ---- partial block of this split -> its slab: rows = ci (accumulator rows), lanes = co (contiguous)
    const int slab_id = s * p.wp + w_p;
    float* dst = p.slab + ((size_t)blockIdx.z * p.nslab + slab_id) * p.K * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < V2W_WG_TG; ++t) {
        if (t >= p.ntap) continue;
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int ci = ci0 + w_ci * MF + F::row(e, hk);
            const int co = co0 + w_co * MF + lr;
            if (ci < p.Cin && co < p.Cout) dst[((size_t)p.tap[t] * p.Cin + ci) * p.Cout + co] = acc[t][e];
        }
    }
}


// Positions (q domain) one staged item covers in the pipelined kernel: the dy tile is PTQ*U <= 256 columns wide.
static constexpr int v2w_wg_ptq(int u) { return u == 1 ? 128 : (u == 2 ? 128 : (u == 4 ? 48 : (u == 5 ? 32 : 0))); }
// Widest signal tile (float4 columns) the pipelined kernel stages: dilated Conv1d halos reach 16 positions per side, the
// per-phase offsets of a transposed conv stay within a few positions.
static constexpr int v2w_wg_xc4(int u) { return u == 1 ? V2W_WG_XC4 : (v2w_wg_ptq(u) + 16) / 4; }

// Pipelined variant (every Conv1d / ConvTranspose1d of the generator at its training shapes; rows 16-byte aligned):
// compile-time wave arrangement, tap count and stride; float4 global loads for item i+1 are issued BEFORE the MFMA loop of
// item i and parked in registers (the HBM/L2 latency hides under ~NT*PTQ/KSTEP MFMAs per wave); LDS is written with the
// activation applied.  For U > 1 the dy tile holds all U phases (contiguous, coalesced) and every tap reads its own phase.
// WIDE: the signal tile may span V2W_WG_XC4_WIDE float4 columns (DiscriminatorP's dilation = period conv: 5 taps at +-2*19).
template <int MF, int WCO, int WCI, int NT, int U, bool WIDE = false>
__global__ void __launch_bounds__(256, (NT > 6 && Frag<MF>::NREG * WCO * WCI == 64) ? 1 : 2)
wgrad_pipe_kernel(const WgradArgs p) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int KSTEP = F::KSTEP, PTQ = v2w_wg_ptq(U), WP = 4 / (WCO * WCI);
    constexpr int CO_T = WCO * MF, CI_T = WCI * MF, SUB = PTQ / WP;
    constexpr int DYC4 = PTQ * U / 4;                               // float4 columns of the dy tile
    constexpr int NDY = (CO_T * DYC4 + 255) / 256;                  // float4 dy loads per thread and item
    constexpr int NXM = (CI_T * (WIDE ? V2W_WG_XC4_WIDE : v2w_wg_xc4(U)) + 255) / 256;         // upper bound of float4 signal loads per thread (= ceil(XC4 / TPR))
    static_assert(SUB % KSTEP == 0 && (PTQ * U) % 4 == 0, "tile shape");
    extern __shared__ float smem[];
    float* const DYs = smem;
    float* const Xas = smem + CO_T * p.ptw;

    const int cot = p.Cout / CO_T;
    const int co0 = (blockIdx.x % cot) * CO_T, ci0 = (blockIdx.x / cot) * CI_T;
    const int s = blockIdx.y;
    const int cob = blockIdx.z * p.Cout + co0, cib = blockIdx.z * p.Cin + ci0;     // channel bases inside the x / dy tensors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int w_p = wave % WP, w_ci = (wave / WP) % WCI, w_co = wave / (WP * WCI);
    const int Lq = p.Lq, Ldy = p.Lq * U;

    // signal-tile staging: TPR threads share one row (one affine pair per thread), float4 columns xl, xl + TPR, ...
    constexpr int TPR = 256 / CI_T;
    const int xrow = tid / TPR, xl = tid % TPR;

    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) acc[t][e] = 0.f;

    const int items = p.B * p.nchunk;
    const int per = (items + p.S - 1) / p.S;
    const int it0 = s * per, it1 = min(items, it0 + per);

    f32x4 dyv[NDY], xv[NXM];
    float xa = 1.f, xs = 0.f;
    auto issue = [&](int it) {
        const int b = it / p.nchunk, q0 = (it - b * p.nchunk) * PTQ;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / DYC4, pos = q0 * U + (idx - row * DYC4) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < CO_T && pos < Ldy) v = *reinterpret_cast<const f32x4*>(p.dy + ((size_t)b * p.CoutT + cob + row) * Ldy + pos);
            dyv[i] = v;
        }
        const int ch = b * p.CinT + cib + xrow;
        const float* xsrc = p.x + (size_t)ch * Lq + (q0 - p.hla);
        if (p.x_a) { xa = p.x_a[ch]; xs = p.x_s[ch]; }
#pragma unroll
        for (int i = 0; i < NXM; ++i) {
            const int c4 = xl + i * TPR;
            const int q = q0 - p.hla + c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c4 < p.xc4 && q >= 0 && q < Lq) v = *reinterpret_cast<const f32x4*>(xsrc + c4 * 4);
            xv[i] = v;
        }
        return q0;
    };

    // operand rows of this lane: B = dy (one pointer per tap when the taps sit on different phases), A = activated signal
    const float* brow = DYs + (w_co * MF + lr) * p.ptw + (w_p * SUB + hk) * U;
    const float* bt[NT];
    const float* at[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bt[t] = brow + p.rr[t];
        at[t] = Xas + (w_ci * MF + lr) * p.xtw + w_p * SUB + hk + p.hla + p.off[t];
    }

    int qnext = 0;
    if (it0 < it1) qnext = issue(it0);
    for (int it = it0; it < it1; ++it) {
        const int qcur = qnext;
        const float xa_cur = xa, xs_cur = xs;
        __syncthreads();                                            // the previous item's MFMA reads are done
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / DYC4;
            if (row >= CO_T) continue;
            float* d = DYs + row * p.ptw + (idx - row * DYC4) * 4;
            d[0] = dyv[i][0]; d[1] = dyv[i][1]; d[2] = dyv[i][2]; d[3] = dyv[i][3];
        }
        {
            // positions outside [0, Lq) must stage as 0 (zero padding of the ACTIVATED signal), not act(s)
            float* drow = Xas + xrow * p.xtw;
#pragma unroll
            for (int i = 0; i < NXM; ++i) {
                const int c4 = xl + i * TPR;
                if (c4 >= p.xc4) continue;
                const int q = qcur - p.hla + c4 * 4;
                const bool in = q >= 0 && q < Lq;
                float* d = drow + c4 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = in ? v2w_lrelu(fmaf(xa_cur, xv[i][e], xs_cur), p.slope) : 0.f;
            }
        }
        __syncthreads();
        if (it + 1 < it1) qnext = issue(it + 1);                    // in flight during the MFMA loop below
#pragma unroll (U == 1 ? 8 : 2)
        for (int kq = 0; kq < SUB; kq += KSTEP) {
            if constexpr (U == 1) {
                const float bv = brow[kq];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = F::mfma(at[t][kq], bv, acc[t]);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = F::mfma(at[t][kq], bt[t][kq * U], acc[t]);
            }
        }
    }

    const int slab_id = s * WP + w_p;
    float* dst = p.slab + ((size_t)blockIdx.z * p.nslab + slab_id) * p.K * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int ci = ci0 + w_ci * MF + F::row(e, hk);
            const int co = co0 + w_co * MF + lr;
            dst[((size_t)p.tap[t] * p.Cin + ci) * p.Cout + co] = acc[t][e];
        }
}

template <int MF, int WCO, int WCI, int NT, int U, bool WIDE = false>
static bool launch_pipe(const WgradArgs& p, int tiles, size_t lds, hipStream_t st) {
    if (!tiles) return true;                                       // dry run: "is this shape instantiated?"
    auto kern = wgrad_pipe_kernel<MF, WCO, WCI, NT, U, WIDE>;
    // the > 64 KiB LDS opt-in is per device: set on the current device at every launch (idempotent, host-side only; the library keeps no state)
    (void)v2w_max_lds(reinterpret_cast<const void*>(kern), 160 * 1024, st);
    V2W_LAUNCH(kern, dim3(tiles, p.S, p.ngroups), dim3(256), lds, st, p);
    return true;
}

template <int MF, int WCO, int WCI>
static bool launch_pipe_nt(const WgradArgs& p, int tiles, size_t lds, hipStream_t st) {
    switch (p.ntap) {
        case 1: return launch_pipe<MF, WCO, WCI, 1, 1>(p, tiles, lds, st);     // 1, 2: the discriminators' unfolded / phase-stacked layers
        case 2: return launch_pipe<MF, WCO, WCI, 2, 1>(p, tiles, lds, st);
        case 3: return launch_pipe<MF, WCO, WCI, 3, 1>(p, tiles, lds, st);
        case 4: return launch_pipe<MF, WCO, WCI, 4, 1>(p, tiles, lds, st);
        case 5: return launch_pipe<MF, WCO, WCI, 5, 1>(p, tiles, lds, st);
        case 6: return launch_pipe<MF, WCO, WCI, 6, 1>(p, tiles, lds, st);
        case 7: return launch_pipe<MF, WCO, WCI, 7, 1>(p, tiles, lds, st);
        default: return false;
    }
}

// true = launched on the pipelined kernel (the instantiated shapes are the generator's own layers; anything else takes
// the generic kernel above)
static bool try_pipe(const WgradArgs& p, int mf, int tiles, size_t lds, hipStream_t st) {
    const int cfg = mf * 100 + p.wco * 10 + p.wci;
    if (p.vec4 && p.u == 1 && cfg == 3222 && p.ntap == 5 && p.xc4 > V2W_WG_XC4 && p.xc4 <= V2W_WG_XC4_WIDE)
        return launch_pipe<32, 2, 2, 5, 1, true>(p, tiles, lds, st);
    if (!p.vec4 || p.xc4 > v2w_wg_xc4(p.u)) return false;
    if (p.u == 1) {
        if (cfg == 3222) return launch_pipe_nt<32, 2, 2>(p, tiles, lds, st);
        if (cfg == 3211) return launch_pipe_nt<32, 1, 1>(p, tiles, lds, st);
        if (cfg == 1611) return launch_pipe_nt<16, 1, 1>(p, tiles, lds, st);
        return false;
    }
    if (p.u == 5 && cfg == 3222 && p.ntap == 6) return launch_pipe<32, 2, 2, 6, 5>(p, tiles, lds, st);
    if (p.u == 5 && cfg == 3222 && p.ntap == 5) return launch_pipe<32, 2, 2, 5, 5>(p, tiles, lds, st);
    if (p.u == 4 && cfg == 3222 && p.ntap == 4) return launch_pipe<32, 2, 2, 4, 4>(p, tiles, lds, st);
    if (p.u == 2 && cfg == 3212 && p.ntap == 4) return launch_pipe<32, 1, 2, 4, 2>(p, tiles, lds, st);
    if (p.u == 2 && cfg == 1612 && p.ntap == 4) return launch_pipe<16, 1, 2, 4, 2>(p, tiles, lds, st);
    return false;
}

// dwf[i] = sum_s slab[s][i]: a block covers 256 consecutive outputs (one float4 per lane) x 4 slab lanes (one per wave);
// every wave sums each 4th slab with 4 loads in flight, the four partial sums are combined in fixed order (deterministic).
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* slab, float* dwf, size_t n, int nslab) {
    __shared__ f32x4 part[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + o) * 4;             // n % 4 == 0 (checked by the plan): weights are k * C_in * C_out
    slab += (size_t)blockIdx.y * nslab * n; dwf += (size_t)blockIdx.y * n;   // grid.y = group
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int s = g;
        f32x4 v1 = v, v2 = v, v3 = v;
        for (; s + 12 < nslab; s += 16) {
            v  += *reinterpret_cast<const f32x4*>(slab + (size_t)s * n + i);
            v1 += *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 4) * n + i);
            v2 += *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 8) * n + i);
            v3 += *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 12) * n + i);
        }
        for (; s < nslab; s += 4) v += *reinterpret_cast<const f32x4*>(slab + (size_t)s * n + i);
        v = (v + v1) + (v2 + v3);
    }
    part[g][o] = v;
    __syncthreads();
    if (g == 0 && i < n) *reinterpret_cast<f32x4*>(dwf + i) = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
}

// The same sum for many slabs and few outputs (the narrow layers: 2048 slabs of 3072 .. 11264 weights - the 4-lane form walks them in 3 - 11
// workgroups, 80 - 150 us of load latency): 16 slab lanes per block of 64 float4 outputs, combined in fixed order.
__global__ void __launch_bounds__(1024)
wgrad_reduce16_kernel(const float* slab, float* dwf, size_t n, int nslab) {
    __shared__ f32x4 part[16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + o) * 4;
    slab += (size_t)blockIdx.y * nslab * n; dwf += (size_t)blockIdx.y * n;   // grid.y = group
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int s = g;
        f32x4 v1 = v, v2 = v, v3 = v;
        for (; s + 48 < nslab; s += 64) {
            v  += *reinterpret_cast<const f32x4*>(slab + (size_t)s * n + i);
            v1 += *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 16) * n + i);
            v2 += *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 32) * n + i);
            v3 += *reinterpret_cast<const f32x4*>(slab + (size_t)(s + 48) * n + i);
        }
        for (; s < nslab; s += 16) v += *reinterpret_cast<const f32x4*>(slab + (size_t)s * n + i);
        v = (v + v1) + (v2 + v3);
    }
    part[g][o] = v;
    __syncthreads();
    if (g == 0 && i < n) {
        f32x4 r = part[0][o];
#pragma unroll
        for (int j = 1; j < 16; ++j) r += part[j][o];
        *reinterpret_cast<f32x4*>(dwf + i) = r;
    }
}

}  // namespace

// Wave arrangement of a workgroup's 4 waves over (C_out, C_in, positions) and the number of partial slabs.  Conv1d (u = 1):
// 2 x 2 x 1 when both channel counts allow two MFMA row blocks, else 1 x 1 x 4 - the shapes the pipelined kernel is instantiated
// for (a mixed 2 x 1 would take the scalar kernel).  Transposed convs keep the arrangement their instantiations were tuned with.
static int v2w_wgrad_plan(int B, int c_in, int c_out, int Lq, int u, int* mf_o, int* wco_o, int* wci_o, int ngroups = 1) {
    int mf = (c_out % 32 == 0 && c_in % 32 == 0) ? 32 : ((c_out % 16 == 0 && c_in % 16 == 0) ? 16 : 0);
    const bool padded = !mf;                   // any other channel counts (the 8-channel stage of a x640 generator): 16-row tiles, the missing rows staged
    if (padded) {                              // as 0 by the generic kernel; the slab reduce sums float4s of the k * c_in * c_out weights
        if (ngroups != 1 || (c_in * c_out) % 4 != 0) return 0;
        mf = 16;
    }
    int wco = 1, wci = 1;
    if (padded) {
    } else if (u == 1) {
        if (c_out % (2 * mf) == 0 && c_in % (2 * mf) == 0) wco = wci = 2;
    } else {
        if (c_out % (2 * mf) == 0) wco = 2;
        if (c_in % (2 * mf) == 0 && wco * 2 <= 4) wci = 2;
    }
    const int wp = 4 / (wco * wci);
    const int tiles = ((c_out + wco * mf - 1) / (wco * mf)) * ((c_in + wci * mf - 1) / (wci * mf));
    const int items = B * ((Lq + 127) / 128);
    int S = (2 * 256 + tiles * ngroups - 1) / (tiles * ngroups);       // ~2 workgroups per CU in flight (all groups of one launch together)
    if (S > items) S = items;
    if (S < 1) S = 1;
    if (mf_o) { *mf_o = mf; *wco_o = wco; *wci_o = wci; }
    return S * wp;
}

// Number of partial slabs ([k][C_in][C_out] floats each) v2w_wgrad / v2w_wgrad_slice need for this problem (an upper bound over the
// conv and transposed-conv arrangements); 0 = shape not supported.
extern "C" int v2w_wgrad_slabs(int B, int c_in, int c_out, int Lq) {
    const int a = v2w_wgrad_plan(B, c_in, c_out, Lq, 1, nullptr, nullptr, nullptr);
    const int b = v2w_wgrad_plan(B, c_in, c_out, Lq, 2, nullptr, nullptr, nullptr);
    return a > b ? a : b;
}

// dwf [k][C_in][C_out] = weight gradient; u = 1 / pad ignored for Conv1d (dil used), stride u and pad = (k-u)/2 for ConvTranspose1d.
static int wgrad_impl(const float* x, const float* x_a, const float* x_s, const float* dy, float* dwf, float* slab_ws,
                      int B, int c_in, int c_out, int Lq, int k, int dil, int u, float slope, int tap0, int x_ct, int dy_ct, int ngroups, void* stream);

extern "C" int v2w_wgrad(const float* x, const float* x_a, const float* x_s, const float* dy, float* dwf, float* slab_ws,
                         int B, int c_in, int c_out, int Lq, int k, int dil, int u, float slope, void* stream) {
    return wgrad_impl(x, x_a, x_s, dy, dwf, slab_ws, B, c_in, c_out, Lq, k, dil, u, slope, -1, 0, 0, 1, stream);
}

// Conv1d weight gradient with the taps at offsets (t - tap0) * dil (tap0 = -1: symmetric (k-1)/2) on channel slices: x / dy point at
// the first channel of a c_in / c_out slice of tensors with x_ct / dy_ct channels per batch item (0: dense) - one group of a
// grouped conv, or the asymmetric tap sets of the discriminators' phase-stacked strided convs.
extern "C" int v2w_wgrad_slice(const float* x, const float* dy, float* dwf, float* slab_ws, int B, int c_in, int c_out, int Lq,
                               int k, int dil, int tap0, int x_ct, int dy_ct, void* stream) {
    if (tap0 < -1 || tap0 >= k || x_ct < 0 || dy_ct < 0 || (x_ct > 0 && x_ct < c_in) || (dy_ct > 0 && dy_ct < c_out)) return V2W_E_ARG;
    return wgrad_impl(x, nullptr, nullptr, dy, dwf, slab_ws, B, c_in, c_out, Lq, k, dil, 1, 1.f, tap0, x_ct, dy_ct, 1, stream);
}

// Slabs PER GROUP of v2w_wgrad_groups (the position splits are shared out over the groups of the launch); 0 = unsupported shape.
extern "C" int v2w_wgrad_group_slabs(int B, int c_in, int c_out, int Lq, int ngroups) {
    if (ngroups < 1) return 0;
    return v2w_wgrad_plan(B, c_in, c_out, Lq, 1, nullptr, nullptr, nullptr, ngroups);
}

// All `ngroups` groups of a grouped Conv1d in ONE launch per tap group (grid.z = group): x (B, ngroups*c_in, Lq), dy (B, ngroups*c_out, Lq),
// dwf [ngroups][k][c_in][c_out]; slab_ws: ngroups * v2w_wgrad_group_slabs(...) * k*c_in*c_out floats.
extern "C" int v2w_wgrad_groups(const float* x, const float* dy, float* dwf, float* slab_ws, int B, int c_in, int c_out, int Lq,
                                int k, int dil, int tap0, int ngroups, void* stream) {
    if (tap0 < -1 || tap0 >= k || ngroups < 1 || ngroups > 65535) return V2W_E_ARG;
    return wgrad_impl(x, nullptr, nullptr, dy, dwf, slab_ws, B, c_in, c_out, Lq, k, dil, 1, 1.f, tap0, ngroups * c_in, ngroups * c_out, ngroups, stream);
}

static int wgrad_impl(const float* x, const float* x_a, const float* x_s, const float* dy, float* dwf, float* slab_ws,
                      int B, int c_in, int c_out, int Lq, int k, int dil, int u, float slope, int tap0, int x_ct, int dy_ct, int ngroups, void* stream) {
    if (!x || !dy || !dwf || !slab_ws || B <= 0 || c_in <= 0 || c_out <= 0 || Lq <= 0 || k <= 0 || dil <= 0 || u <= 0) return V2W_E_ARG;
    if ((x_a == nullptr) != (x_s == nullptr)) return V2W_E_ARG;
    int mf = 0, wco = 1, wci = 1;
    const int nslab = v2w_wgrad_plan(B, c_in, c_out, Lq, u, &mf, &wco, &wci, ngroups);
    if (!nslab) return V2W_E_SHAPE;
    WgradArgs p{};
    p.x = x; p.x_a = x_a; p.x_s = x_s; p.dy = dy; p.slab = slab_ws;
    p.B = B; p.Cin = c_in; p.Cout = c_out; p.Lq = Lq; p.K = k; p.u = u; p.slope = slope;
    p.CinT = x_ct > 0 ? x_ct : c_in; p.CoutT = dy_ct > 0 ? dy_ct : c_out;
    p.ngroups = ngroups;
    p.wco = wco; p.wci = wci;
    p.wp = 4 / (p.wco * p.wci);
    p.S = nslab / p.wp;
    p.nslab = nslab;
    const int tiles = ((c_out + p.wco * mf - 1) / (p.wco * mf)) * ((c_in + p.wci * mf - 1) / (p.wci * mf));
    const bool padded = c_out % 16 != 0 || c_in % 16 != 0;        // (generic kernel only: the pipelined ones stage whole tiles)
    const int pad = u > 1 ? (k - u) / 2 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (Lq % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);

    // LDS row strides: odd for the 32-wide MFMA (32 rows -> 32 banks), 2 mod 32 for the 16-wide one (four 16-lane k-groups)
    auto set_strides = [&](int dy_cols) {
        int ptw = dy_cols, xtw = p.xc4 * 4;
        if (mf == 32) { ptw |= 1; xtw |= 1; }
        else { ptw += ((2 - ptw % 32) + 32) % 32; xtw += ((2 - xtw % 32) + 32) % 32; }
        p.ptw = ptw; p.xtw = xtw;
        return ((size_t)p.wco * mf * ptw + (size_t)p.wci * mf * xtw) * sizeof(float);
    };
    auto set_group = [&](const int* taps, const int* offs, const int* phs, int n, int ptq) {
        p.ntap = n;
        int lo = 0, hi = 0;
        for (int i = 0; i < n; ++i) {
            p.tap[i] = taps[i]; p.off[i] = offs[i]; p.rr[i] = phs[i];
            if (offs[i] < lo) lo = offs[i];
            if (offs[i] > hi) hi = offs[i];
        }
        p.hl = -lo; p.hr = hi;
        p.hla = (p.hl + 3) & ~3;
        p.xc4 = (p.hla + ptq + p.hr + 3) / 4;
    };

    // every tap with its dy phase and signal offset
    int taps[64], offs[64], phs[64], n = 0;
    for (int t = 0; t < k && n < 64; ++t, ++n) {
        taps[n] = t;
        if (u == 1) { phs[n] = 0; offs[n] = (t - (tap0 >= 0 ? tap0 : (k - 1) / 2)) * dil; }
        else {
            const int r = ((t - pad) % u + u) % u;
            const int t0 = (r + pad) % u, c = (r + pad) / u, m = (t - t0) / u;
            phs[n] = r; offs[n] = c - m;
        }
    }
    if (n < k) return V2W_E_SHAPE;
    const int ngrp = (n + V2W_WG_TG - 1) / V2W_WG_TG;
    const int gsz = (n + ngrp - 1) / ngrp;                         // balanced groups: 11 taps -> 6 + 5, 8 -> 4 + 4

    // 1) pipelined kernel: groups of taps across phases, one staged dy tile holds all u phases
    const int ptq = v2w_wg_ptq(u);
    bool piped = aligned && ptq > 0 && !padded;
    if (piped) {
        p.vec4 = 1;
        p.nchunk = (Lq + ptq - 1) / ptq;
        for (int g0 = 0; g0 < n && piped; g0 += gsz) {
            set_group(taps + g0, offs + g0, phs + g0, n - g0 < gsz ? n - g0 : gsz, ptq);
            const size_t lds = set_strides(ptq * u);
            if (lds > 160 * 1024 || !try_pipe(p, mf, 0, lds, st)) piped = false;       // dry run over all groups first
        }
        for (int g0 = 0; g0 < n && piped; g0 += gsz) {
            set_group(taps + g0, offs + g0, phs + g0, n - g0 < gsz ? n - g0 : gsz, ptq);
            try_pipe(p, mf, tiles, set_strides(ptq * u), st);
        }
    }
    // 2) generic kernel: one dy phase per launch, scalar staging
    if (!piped) {
        p.vec4 = 0;
        p.nchunk = (Lq + 127) / 128;
        for (int r = 0; r < u; ++r) {
            int gt[64], go[64], gp[64], m = 0;
            for (int i = 0; i < n; ++i) if (phs[i] == r) { gt[m] = taps[i]; go[m] = offs[i]; gp[m] = r; ++m; }
            const int ng = (m + V2W_WG_TG - 1) / V2W_WG_TG;
            const int gs = ng ? (m + ng - 1) / ng : 1;
            for (int g0 = 0; g0 < m; g0 += gs) {
                p.r = r;
                set_group(gt + g0, go + g0, gp + g0, m - g0 < gs ? m - g0 : gs, 128);
                const size_t lds = set_strides(128);
                if (lds > 160 * 1024) return V2W_E_SHAPE;
                if (mf == 32) {
                    if (lds > 64 * 1024) (void)v2w_max_lds(reinterpret_cast<const void*>(wgrad_kernel<32>), (int)lds, st);
                    V2W_LAUNCH(wgrad_kernel<32>, dim3(tiles, p.S, p.ngroups), dim3(256), lds, st, p);
                } else {
                    if (lds > 64 * 1024) (void)v2w_max_lds(reinterpret_cast<const void*>(wgrad_kernel<16>), (int)lds, st);
                    V2W_LAUNCH(wgrad_kernel<16>, dim3(tiles, p.S, p.ngroups), dim3(256), lds, st, p);
                }
            }
        }
    }
    const size_t nw = (size_t)k * c_in * c_out;
    if (nslab >= 128) V2W_LAUNCH(wgrad_reduce16_kernel, dim3((unsigned)((nw + 255) / 256), ngroups), dim3(1024), 0, st, slab_ws, dwf, nw, nslab);
    else V2W_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((nw + 255) / 256), ngroups), dim3(256), 0, st, slab_ws, dwf, nw, nslab);
    return v2w_launch_status();
}
