// Fused residual pair for the narrow stages (C = 32 or 16: the whole channel dimension is one MFMA chunk):
//
//   ResBlock2 step pair (models.py:65-70):  t1 = x + conv_{k,d1}(lrelu(x)) + b1 ;  out = t1 + conv_{k,d2}(lrelu(t1)) + b2
//   ResBlock1 pair      (models.py:37-44):  t1 =     conv_{k,d1}(lrelu(x)) + b1 ;  out = x  + conv_{k,d2}(lrelu(t1)) + b2
//
// in ONE kernel: the intermediate t1 never leaves the CU.  These layers are HBM-bound as separate launches
// (12-35 FLOP/B at C = 16); fused, the tile of x is read once and only `out` is written.
//
//   LDS  X [C][xw] : x = a*in + s (CondBN affine folded in), exactly 0 outside [0, L); leaky_relu is applied when an
//                    operand is read, so the same tile also serves as the residual
//        T1[C][tw] : t1 on the W positions conv2 needs, exactly 0 outside [0, L) (conv2 zero-pads t1, not x)
//   Each wave owns all C output channels x (W / WN) positions.  conv1 runs on W = MF*NI*WN positions, conv2's valid
//   outputs are the NTO = W - 2*h2 (rounded down to a multiple of 4) central ones: both phases issue the same
//   number of MFMAs per wave and tile boundaries stay 16-byte aligned.
//   Weights: the same packed A-fragment streams the per-layer kernel uses (v2w_pack_mfma, single chunk), read straight
//   from L2 through a two-deep register ping-pong.
#include "v2w_common.h"

namespace {

struct PairArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* wp1; const float* bias1; const float* wp2; const float* bias2;
    const float* add0; const float* add1;
    float* out;
    int B, C, L, K, d1, d2;
    int h1, h2;       // halos of the two convs
    int xoff;         // X column of position (n0 - h2 - h1); X column 0 is 16-byte aligned in global memory
    int xw, tw;       // LDS row strides
    int xcols;        // staged X columns (multiple of 4)
    int nto;          // valid outputs per tile
    int ntl, ntiles;
    int vec4;
    int res_mode;     // 0: ResBlock2 (t1 += x, out += t1)   1: ResBlock1 (out += x)
    int toff, eoff;   // LDS offsets (floats) of T1 and of the bias table
    float slope, out_div;
};

#define V2W_PAIR_MULTI 4
struct PairMulti {
    PairArgs p[V2W_PAIR_MULTI];
    int start[V2W_PAIR_MULTI + 1];
};

template <int MF, int NI, int WN>
__global__ void __launch_bounds__(64 * WN)
resblock_pair_kernel(const PairMulti m) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int NTHREADS = 64 * WN;
    constexpr int C = MF;                   // all channels = one MFMA row block = one packed chunk
    constexpr int W = MF * NI * WN;         // positions computed per phase
    constexpr int KSTEP = F::KSTEP;
    constexpr int CKG = 4 * KSTEP;
    constexpr int GPC = C / CKG;            // A fragments per tap: 4 (MF = 32) or 1 (MF = 16)

    extern __shared__ __attribute__((aligned(16))) float smem[];

    int pq = 0;
#pragma unroll
    for (int i = 1; i < V2W_PAIR_MULTI; ++i) pq += (int)blockIdx.x >= m.start[i] ? 1 : 0;
    const PairArgs& p = m.p[pq];
    const int tile = blockIdx.x - m.start[pq];
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * p.nto;  // first output position of the tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int wn0 = wave * (MF * NI);
    const int L = p.L, K = p.K, xw = p.xw, tw = p.tw;
    const float slope = p.slope;
    float* const Xs = smem;                 // [C][xw]
    float* const Ts = smem + p.toff;        // [C][tw]; ResBlock2 (res_mode 0) overlays it on X: x is dead once t1 exists
    float* const etab = smem + p.eoff;      // bias1[C], bias2[C]

    // ---- stage x = a*in + s (0 outside the sequence); X column 0 <-> position pos0 (a multiple of 4)
    const int pos0 = n0 - p.h2 - p.h1 - p.xoff;
    if (tid < C) { etab[tid] = p.bias1 ? p.bias1[tid] : 0.f; etab[C + tid] = p.bias2 ? p.bias2[tid] : 0.f; }
    if (p.vec4) {
        const int xw4 = p.xcols >> 2;
        const unsigned magic = (unsigned)(((1ull << 32) + xw4 - 1) / xw4);
        for (int idx = tid; idx < C * xw4; idx += NTHREADS) {
            const int row = (int)__umulhi((unsigned)idx, magic);
            const int col = (idx - row * xw4) * 4;
            const int pos = pos0 + col;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos >= 0 && pos < L) {       // L % 4 == 0, pos % 4 == 0: whole float4 inside
                const int ch = b * C + row;
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.in + (size_t)ch * L + pos);
                const float av = p.in_a ? p.in_a[ch] : 1.f, sv = p.in_a ? p.in_s[ch] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(av, g[e], sv);
            }
            *reinterpret_cast<f32x4*>(Xs + row * xw + col) = v;
        }
    } else {
        for (int c = wave; c < C; c += WN) {
            const int ch = b * C + c;
            const float av = p.in_a ? p.in_a[ch] : 1.f, sv = p.in_a ? p.in_s[ch] : 0.f;
            for (int j = lane; j < p.xcols; j += 64) {
                const int pos = pos0 + j;
                Xs[c * xw + j] = (pos >= 0 && pos < L) ? fmaf(av, p.in[(size_t)ch * L + pos], sv) : 0.f;
            }
        }
    }

    // ---- weight streams: K*GPC fragments of 1 KiB each per conv, consumed in order
    const int nfrag = K * GPC;
    const f32x4* ap1 = reinterpret_cast<const f32x4*>(p.wp1) + lane;
    const f32x4* ap2 = reinterpret_cast<const f32x4*>(p.wp2) + lane;
    f32x4 a0, a1;
    a0 = ap1[0];
    __syncthreads();

    acc_t acc[NI];
    // one conv phase: acc[j] = sum over taps / channels of A * lrelu(src[..][col + t*dil])
    auto conv_phase = [&](const f32x4* ap, const f32x4* ap_next, const float* src, int sw, int colbase, int dil) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < F::NREG; ++e) acc[j][e] = 0.f;
        int f = 0;
        auto step = [&](const f32x4& use, f32x4& ld) {
            // next fragment of this conv, or the first one of the following conv (clamped re-read at the very end)
            const int fn = f + 1;
            ld = fn < nfrag ? ap[(size_t)fn * 64] : ap_next[0];
            __builtin_amdgcn_sched_barrier(0);
            const int t = f / GPC, gg = f - t * GPC;
            const float* xrow = src + (gg * CKG + hk) * sw + colbase + t * dil;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float bb[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) bb[j] = v2w_lrelu(xrow[kk * KSTEP * sw + j * MF], slope);
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[j] = F::mfma(use[kk], bb[j], acc[j]);
            }
            ++f;
        };
        int it = 0;
        for (; it + 1 < nfrag; it += 2) { step(a0, a1); step(a1, a0); }
        if (it < nfrag) { step(a0, a1); a0 = a1; }
    };

    // ---- conv1 -> t1 on positions [n0 - h2, n0 - h2 + W); X column of (position, tap 0) = col + xoff
    conv_phase(ap1, ap2, Xs, xw, wn0 + lr + p.xoff, p.d1);
#pragma unroll
    for (int e = 0; e < F::NREG; ++e) {
        const int co = F::row(e, hk);
        const float bias = etab[co];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn0 + j * MF + lr;
            const int pos = n0 - p.h2 + col;
            float v = acc[j][e] + bias;
            if (p.res_mode == 0) v += Xs[co * xw + col + p.xoff + p.h1];
            acc[j][e] = (pos >= 0 && pos < L) ? v : 0.f;
        }
    }
    if (p.toff == 0) __syncthreads();       // T1 overlays X: every wave must be done reading x (operands and residual)
#pragma unroll
    for (int e = 0; e < F::NREG; ++e) {
        const int co = F::row(e, hk);
#pragma unroll
        for (int j = 0; j < NI; ++j) Ts[co * tw + wn0 + j * MF + lr] = acc[j][e];
    }
    __syncthreads();

    // ---- conv2 on the first NTO columns' worth of outputs (all W computed, the rest masked); T1 column of tap 0 = col
    conv_phase(ap2, ap2 + (size_t)(nfrag - 1) * 64, Ts, tw, wn0 + lr, p.d2);
#pragma unroll
    for (int e = 0; e < F::NREG; ++e) {
        const int co = F::row(e, hk);
        const float bias = etab[C + co];
        const size_t orow = ((size_t)b * C + co) * L;
        float a0v[NI], a1v[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn0 + j * MF + lr, pos = n0 + col;
            const bool ok = col < p.nto && pos < L;
            a0v[j] = (p.add0 && ok) ? p.add0[orow + pos] : 0.f;
            a1v[j] = (p.add1 && ok) ? p.add1[orow + pos] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn0 + j * MF + lr, pos = n0 + col;
            if (col >= p.nto || pos >= L) continue;
            float v = acc[j][e] + bias;
            v += p.res_mode == 0 ? Ts[co * tw + col + p.h2] : Xs[co * xw + col + p.xoff + p.h1 + p.h2];
            if (p.add1) v += a0v[j] + a1v[j];
            else if (p.add0) v += a0v[j];
            if (p.out_div != 0.f) v = v / p.out_div;
            p.out[orow + pos] = v;
        }
    }
}

template <int MF, int NI, int WN>
int launch_pair(const v2w_pair_args* a, int n, hipStream_t stream) {
    constexpr int W = MF * NI * WN;
    PairMulti m{};
    size_t lds = 0;
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        const v2w_pair_args& q = a[i];
        PairArgs p{};
        p.in = q.in; p.in_a = q.in_a; p.in_s = q.in_s; p.wp1 = q.wp1; p.bias1 = q.bias1; p.wp2 = q.wp2; p.bias2 = q.bias2;
        p.add0 = q.add0; p.add1 = q.add1; p.out = q.out;
        p.B = q.B; p.C = q.C; p.L = q.L; p.K = q.k; p.d1 = q.dil1; p.d2 = q.dil2;
        p.h1 = q.dil1 * (q.k - 1) / 2; p.h2 = q.dil2 * (q.k - 1) / 2;
        p.nto = (W - 2 * p.h2) & ~3;
        if (p.nto < W / 2) return V2W_E_SHAPE;             // receptive field too wide for this tile: use the per-layer path
        const int hsum = p.h1 + p.h2;
        p.xoff = ((hsum + 3) & ~3) - hsum;
        p.xcols = (p.xoff + W + 2 * p.h1 + 3) & ~3;
        int xw = p.xcols, tw = W + 2 * p.h2;
        tw = (tw + 3) & ~3;
        if (MF == 16) { xw += ((16 - xw % 32) + 32) % 32; tw += ((16 - tw % 32) + 32) % 32; }
        p.xw = xw; p.tw = tw;
        p.ntl = (q.L + p.nto - 1) / p.nto;
        p.ntiles = q.B * p.ntl;
        p.vec4 = (q.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(q.in) & 15) == 0);
        p.res_mode = q.res_mode; p.slope = q.slope; p.out_div = q.out_div;
        p.toff = q.res_mode == 0 ? 0 : MF * xw;
        p.eoff = q.res_mode == 0 ? MF * (xw > tw ? xw : tw) : MF * (xw + tw);
        const size_t l = ((size_t)p.eoff + 2 * MF) * sizeof(float);
        if (l > lds) lds = l;
        m.p[i] = p;
        m.start[i] = grid;
        grid += p.ntiles;
    }
    m.start[n] = grid;
    for (int i = n + 1; i <= V2W_PAIR_MULTI; ++i) m.start[i] = 0x7fffffff;
    auto kern = resblock_pair_kernel<MF, NI, WN>;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return V2W_E_SHAPE;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN), lds, stream, m);
    return v2w_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------
// Whole residual section of a narrow ResBlock2 stage in ONE kernel (models.py:135-141 with ResBlock2.forward inlined):
//   out = ( sum_j [ t1_j + conv_{k_j,d2_j}(lrelu(t1_j)) + b2_j ] ) / nk ,   t1_j = x + conv_{k_j,d1_j}(lrelu(x)) + b1_j
// The x tile is staged ONCE for all nk branches, every t1_j lives only in LDS, the branch sum lives in registers and is
// added in the reference's order ((r0 + r1) + r2); HBM sees one read of x and one write of out per stage.
#define V2W_STAGE_MAXB 4
struct StageArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* wp1[V2W_STAGE_MAXB]; const float* bias1[V2W_STAGE_MAXB];
    const float* wp2[V2W_STAGE_MAXB]; const float* bias2[V2W_STAGE_MAXB];
    int K[V2W_STAGE_MAXB], d1[V2W_STAGE_MAXB], d2[V2W_STAGE_MAXB];
    float* out;
    int nk, B, L;
    int h1max, h2max;
    int xoff, xw, tw, xcols, nto, ntl;
    int vec4;
    float slope, out_div;
};

template <int MF, int NI, int WN>
__global__ void __launch_bounds__(64 * WN)
resblock2_stage_kernel(const StageArgs p) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int NTHREADS = 64 * WN;
    constexpr int C = MF;
    constexpr int KSTEP = F::KSTEP;
    constexpr int CKG = 4 * KSTEP;
    constexpr int GPC = C / CKG;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * p.nto;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int wn0 = wave * (MF * NI);
    const int L = p.L, xw = p.xw, tw = p.tw;
    const float slope = p.slope;
    float* const Xs = smem;                 // [C][xw]
    float* const Ts = smem + C * xw;        // [C][tw]
    float* const etab = Ts + C * tw;        // bias1[nk][C], bias2[nk][C]

    const int pos0 = n0 - p.h2max - p.h1max - p.xoff;
    for (int i = tid; i < p.nk * C; i += NTHREADS) {
        const int j = i / C, c = i - j * C;
        etab[i] = p.bias1[j] ? p.bias1[j][c] : 0.f;
        etab[V2W_STAGE_MAXB * C + i] = p.bias2[j] ? p.bias2[j][c] : 0.f;
    }
    if (p.vec4) {
        const int xw4 = p.xcols >> 2;
        const unsigned magic = (unsigned)(((1ull << 32) + xw4 - 1) / xw4);
        for (int idx = tid; idx < C * xw4; idx += NTHREADS) {
            const int row = (int)__umulhi((unsigned)idx, magic);
            const int col = (idx - row * xw4) * 4;
            const int pos = pos0 + col;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos >= 0 && pos < L) {
                const int ch = b * C + row;
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.in + (size_t)ch * L + pos);
                const float av = p.in_a ? p.in_a[ch] : 1.f, sv = p.in_a ? p.in_s[ch] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(av, g[e], sv);
            }
            *reinterpret_cast<f32x4*>(Xs + row * xw + col) = v;
        }
    } else {
        for (int c = wave; c < C; c += WN) {
            const int ch = b * C + c;
            const float av = p.in_a ? p.in_a[ch] : 1.f, sv = p.in_a ? p.in_s[ch] : 0.f;
            for (int j = lane; j < p.xcols; j += 64) {
                const int pos = pos0 + j;
                Xs[c * xw + j] = (pos >= 0 && pos < L) ? fmaf(av, p.in[(size_t)ch * L + pos], sv) : 0.f;
            }
        }
    }

    f32x4 a0, a1;
    a0 = (reinterpret_cast<const f32x4*>(p.wp1[0]) + lane)[0];
    __syncthreads();

    acc_t acc[NI], oacc[NI];
    auto conv_phase = [&](const f32x4* ap, const f32x4* ap_next, int nfrag, const float* src, int sw, int colbase, int dil) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < F::NREG; ++e) acc[j][e] = 0.f;
        int f = 0;
        auto step = [&](const f32x4& use, f32x4& ld) {
            const int fn = f + 1;
            ld = fn < nfrag ? ap[(size_t)fn * 64] : ap_next[0];
            __builtin_amdgcn_sched_barrier(0);
            const int t = f / GPC, gg = f - t * GPC;
            const float* xrow = src + (gg * CKG + hk) * sw + colbase + t * dil;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float bb[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) bb[j] = v2w_lrelu(xrow[kk * KSTEP * sw + j * MF], slope);
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[j] = F::mfma(use[kk], bb[j], acc[j]);
            }
            ++f;
        };
        int it = 0;
        for (; it + 1 < nfrag; it += 2) { step(a0, a1); step(a1, a0); }
        if (it < nfrag) { step(a0, a1); a0 = a1; }
    };

    for (int jb = 0; jb < p.nk; ++jb) {
        const int K = p.K[jb], d1 = p.d1[jb], d2 = p.d2[jb];
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        const int nfrag = K * GPC;
        const f32x4* ap1 = reinterpret_cast<const f32x4*>(p.wp1[jb]) + lane;
        const f32x4* ap2 = reinterpret_cast<const f32x4*>(p.wp2[jb]) + lane;
        const f32x4* ap_after = jb + 1 < p.nk ? reinterpret_cast<const f32x4*>(p.wp1[jb + 1]) + lane : ap2 + (size_t)(nfrag - 1) * 64;

        // ---- conv1_j -> t1_j on positions [n0 - h2max, n0 - h2max + W)
        conv_phase(ap1, ap2, nfrag, Xs, xw, wn0 + lr + p.xoff + (p.h1max - h1), d1);
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int co = F::row(e, hk);
            const float bias = etab[jb * C + co];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn0 + j * MF + lr;
                const int pos = n0 - p.h2max + col;
                const float v = acc[j][e] + bias + Xs[co * xw + col + p.xoff + p.h1max];
                acc[j][e] = (pos >= 0 && pos < L) ? v : 0.f;
            }
        }
        if (jb > 0) __syncthreads();          // conv2 of the previous branch has finished reading T1
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int co = F::row(e, hk);
#pragma unroll
            for (int j = 0; j < NI; ++j) Ts[co * tw + wn0 + j * MF + lr] = acc[j][e];
        }
        __syncthreads();

        // ---- conv2_j ; r_j = (acc + b2) + t1_j ; branch sum in the reference's order
        conv_phase(ap2, ap_after, nfrag, Ts, tw, wn0 + lr + (p.h2max - h2), d2);
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int co = F::row(e, hk);
            const float bias = etab[V2W_STAGE_MAXB * C + jb * C + co];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float r = (acc[j][e] + bias) + Ts[co * tw + wn0 + j * MF + lr + p.h2max];
                oacc[j][e] = jb == 0 ? r : oacc[j][e] + r;
            }
        }
    }

#pragma unroll
    for (int e = 0; e < F::NREG; ++e) {
        const int co = F::row(e, hk);
        const size_t orow = ((size_t)b * C + co) * L;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn0 + j * MF + lr, pos = n0 + col;
            if (col >= p.nto || pos >= L) continue;
            float v = oacc[j][e];
            if (p.out_div != 0.f) v = v / p.out_div;
            p.out[orow + pos] = v;
        }
    }
}

template <int MF, int NI, int WN>
int launch_stage(const v2w_stage_args* q, hipStream_t stream) {
    constexpr int W = MF * NI * WN;
    StageArgs p{};
    p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.out = q->out;
    p.nk = q->nk; p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div;
    for (int j = 0; j < q->nk; ++j) {
        p.wp1[j] = q->wp1[j]; p.bias1[j] = q->bias1[j]; p.wp2[j] = q->wp2[j]; p.bias2[j] = q->bias2[j];
        p.K[j] = q->k[j]; p.d1[j] = q->dil1[j]; p.d2[j] = q->dil2[j];
        const int h1 = q->dil1[j] * (q->k[j] - 1) / 2, h2 = q->dil2[j] * (q->k[j] - 1) / 2;
        if (h1 > p.h1max) p.h1max = h1;
        if (h2 > p.h2max) p.h2max = h2;
    }
    p.nto = (W - 2 * p.h2max) & ~3;
    if (p.nto < W / 2) return V2W_E_SHAPE;
    const int hsum = p.h1max + p.h2max;
    p.xoff = ((hsum + 3) & ~3) - hsum;
    p.xcols = (p.xoff + W + 2 * p.h1max + 3) & ~3;
    int xw = p.xcols, tw = (W + 2 * p.h2max + 3) & ~3;
    if (MF == 16) { xw += ((16 - xw % 32) + 32) % 32; tw += ((16 - tw % 32) + 32) % 32; }
    p.xw = xw; p.tw = tw;
    p.ntl = (q->L + p.nto - 1) / p.nto;
    p.vec4 = (q->L % 4 == 0) && ((reinterpret_cast<uintptr_t>(q->in) & 15) == 0);
    const size_t lds = ((size_t)MF * (xw + tw) + 2 * V2W_STAGE_MAXB * MF) * sizeof(float);
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    auto kern = resblock2_stage_kernel<MF, NI, WN>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(q->B * p.ntl), dim3(64 * WN), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

extern "C" int v2w_resblock_pair_fwd(const v2w_pair_args* a, int n, void* stream) {
    if (!a || n < 1 || n > V2W_PAIR_MULTI) return V2W_E_ARG;
    for (int i = 0; i < n; ++i) {
        const v2w_pair_args& q = a[i];
        if (!q.in || !q.wp1 || !q.wp2 || !q.out) return V2W_E_ARG;
        if (q.B <= 0 || q.C <= 0 || q.L <= 0 || q.k <= 0 || q.dil1 <= 0 || q.dil2 <= 0) return V2W_E_ARG;
        if ((q.k & 1) == 0) return V2W_E_SHAPE;
        if ((q.in_a == nullptr) != (q.in_s == nullptr)) return V2W_E_ARG;
        if (q.add1 && !q.add0) return V2W_E_ARG;
        if (q.res_mode != 0 && q.res_mode != 1) return V2W_E_ARG;
        if (q.C != a[0].C || q.B != a[0].B || q.L != a[0].L) return V2W_E_SHAPE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (a[0].C == 32) return launch_pair<32, 2, 4>(a, n, st);     // 32 channels x 256 positions per workgroup
    if (a[0].C == 16) return launch_pair<16, 4, 4>(a, n, st);     // 16 channels x 256 positions per workgroup
    return V2W_E_SHAPE;
}

extern "C" int v2w_resblock2_stage_fwd(const v2w_stage_args* a, void* stream) {
    if (!a || !a->in || !a->out || a->nk < 1 || a->nk > V2W_STAGE_MAXB) return V2W_E_ARG;
    if (a->B <= 0 || a->C <= 0 || a->L <= 0) return V2W_E_ARG;
    if ((a->in_a == nullptr) != (a->in_s == nullptr)) return V2W_E_ARG;
    for (int j = 0; j < a->nk; ++j) {
        if (!a->wp1[j] || !a->wp2[j] || a->k[j] <= 0 || a->dil1[j] <= 0 || a->dil2[j] <= 0) return V2W_E_ARG;
        if ((a->k[j] & 1) == 0) return V2W_E_SHAPE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (a->C == 32) return launch_stage<32, 2, 4>(a, st);
    if (a->C == 16) return launch_stage<16, 4, 4>(a, st);
    return V2W_E_SHAPE;
}
