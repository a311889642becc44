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


// ---------------------------------------------------------------------------------------------------------------
// Stage x = a*in + s (the folded CondBN affine; exactly 0 outside [0, L)) of one (batch item, window) into Xs[C][xw].
// float4 form: every thread issues ALL its global loads (NPF slots) back to back and only then converts and writes LDS - a plain
// `for (idx = tid; ...; idx += NTHREADS)` loop compiles to load -> wait -> write per trip, i.e. C*xcols/4/NTHREADS serialised HBM
// round trips per tile (9 at C = 32) during which the workgroup issues no MFMA.
template <int C, int NTHREADS>
struct XStage {
    static constexpr int HMAX = 32;                                   // largest conv1 halo the slots cover
    static constexpr int W = 256;
    static constexpr int NPF = (C * ((W + 2 * HMAX + 8) / 4) + NTHREADS - 1) / NTHREADS;
    static_assert(NPF * NTHREADS < 8192, "slot index range of the magic division");
    static bool fits(int xcols) { return C * (xcols >> 2) <= NPF * NTHREADS; }

    static __device__ __forceinline__ void vec4(const float* __restrict__ in, const float* __restrict__ in_a, const float* __restrict__ in_s,
                                                float* Xs, int b, int L, int pos0, int xcols, int xw, int tid) {
        const int xw4 = xcols >> 2;
        const unsigned magic = (unsigned)(((1ull << 32) + xw4 - 1) / xw4);
        f32x4 g[NPF];
        float av[NPF], sv[NPF];
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTHREADS;
            const int row = (int)__umulhi((unsigned)idx, magic);
            const int pos = pos0 + (idx - row * xw4) * 4;
            g[s] = f32x4{0.f, 0.f, 0.f, 0.f}; av[s] = 1.f; sv[s] = 0.f;
            if (idx < C * xw4 && pos >= 0 && pos < L) {       // L % 4 == 0, pos % 4 == 0: whole float4 inside
                const int ch = b * C + row;
                g[s] = *reinterpret_cast<const f32x4*>(in + (size_t)ch * L + pos);
                if (in_a) { av[s] = in_a[ch]; sv[s] = in_s[ch]; }
            }
        }
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTHREADS;
            if (idx >= C * xw4) continue;
            const int row = (int)__umulhi((unsigned)idx, magic);
            const int col = (idx - row * xw4) * 4;
            const int pos = pos0 + col;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pos >= 0 && pos < L) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(av[s], g[s][e], sv[s]);
            }
            *reinterpret_cast<f32x4*>(Xs + row * xw + col) = v;
        }
    }
    // any L / alignment: dword loads, one channel row per wave at a time
    static __device__ __forceinline__ void scalar(const float* __restrict__ in, const float* __restrict__ in_a, const float* __restrict__ in_s,
                                                  float* Xs, int b, int L, int pos0, int xcols, int xw, int wave, int lane) {
        for (int c = wave; c < C; c += NTHREADS / 64) {
            const int ch = b * C + c;
            const float av = in_a ? in_a[ch] : 1.f, sv = in_a ? in_s[ch] : 0.f;
            for (int j = lane; j < xcols; j += 64) {
                const int pos = pos0 + j;
                Xs[c * xw + j] = (pos >= 0 && pos < L) ? fmaf(av, in[(size_t)ch * L + pos], sv) : 0.f;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// One conv phase of the fused kernels: acc[j] += sum over taps t < K and channels c of W[t][c][.] * lrelu(src[c][col + t*dil]).
// All C channels are one packed chunk, so a tap is QT = 4*GPC k-steps and k-step q reads row hk + q*KSTEP of `src` at the tap's
// column: a uniformly strided walk.  Both operand streams are software-pipelined by hand (hipcc alone emits
// ds_read -> s_waitcnt lgkmcnt(0) -> MFMA with the LDS latency exposed at every k-step):
//   B (signal, LDS): the read of k-step q + LOOK is issued before the MFMAs of k-step q; slot q % NB of `bq` holds k-step q; the
//                    leaky_relu is applied in registers when a slot is consumed, so the tile also serves as the residual.
//   A (weights, L2): RING-deep register ring, RING - 1 fragments (1 KiB each) in flight; the stream runs on into the next
//                    conv's fragments (`ap_next`) so a phase change does not drain it.
template <int MF, int NI, int RING, int LOOK>
struct FusedConv {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    static constexpr int KSTEP = F::KSTEP, CKG = 4 * KSTEP, GPC = MF / CKG, QT = 4 * GPC, NB = LOOK + 1;
    static_assert(QT % NB == 0 && LOOK < QT, "operand slots must line up at every tap start");
    static_assert(GPC % RING == 0 || (GPC == 1 && RING == 2), "ring slot of a tap's first fragment must be static");

    f32x4 ar[RING];
    float bq[NB][NI];
    int rb;            // GPC == 1 only: ring slot of the next tap's fragment (alternates per tap)

    // fragment g of the running conv (g >= nfrag: the following conv's stream)
    static __device__ __forceinline__ f32x4 frag(const f32x4* ap, const f32x4* ap_next, int nfrag, int g) {
        return g < nfrag ? ap[(size_t)g * 64] : ap_next[(size_t)(g - nfrag) * 64];
    }
    // before the first phase: the first RING - 1 fragments of the first conv
    __device__ __forceinline__ void start(const f32x4* ap) {
#pragma unroll
        for (int g = 0; g + 1 < RING; ++g) ar[g] = ap[(size_t)g * 64];
        rb = 0;
    }
    __device__ __forceinline__ void prime(const float* xt, int sw) {
#pragma unroll
        for (int q = 0; q < LOOK; ++q)
#pragma unroll
            for (int j = 0; j < NI; ++j) bq[q][j] = xt[q * KSTEP * sw + j * MF];
    }
    template <int RB>
    __device__ __forceinline__ void tap(acc_t (&acc)[NI], const f32x4* ap, const f32x4* ap_next, int nfrag, int g0,
                                        const float* xt, const float* xn, int sw, float slope) {
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            if ((q & 3) == 0) ar[(RB + (q >> 2) + RING - 1) % RING] = frag(ap, ap_next, nfrag, g0 + (q >> 2) + RING - 1);
            const int qa = q + LOOK;
            const float* src = qa < QT ? xt + qa * KSTEP * sw : xn + (qa - QT) * KSTEP * sw;
#pragma unroll
            for (int j = 0; j < NI; ++j) bq[qa % NB][j] = src[j * MF];
            __builtin_amdgcn_sched_barrier(0);      // reads and weight prefetch stay AHEAD of this k-step's MFMAs
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[j] = F::mfma(ar[(RB + (q >> 2)) % RING][q & 3], v2w_lrelu(bq[q % NB][j], slope), acc[j]);
        }
    }
    // the whole phase; x0 = src + hk*sw + (this lane's column of tap 0).  Zeroes acc first.
    __device__ __forceinline__ void run(acc_t (&acc)[NI], const f32x4* ap, const f32x4* ap_next, int K, const float* x0, int sw,
                                        int dil, float slope) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < F::NREG; ++e) acc[j][e] = 0.f;
        const int nfrag = K * GPC;
        prime(x0, sw);
        const float* xt = x0;
        if constexpr (GPC % RING == 0) {
            for (int t = 0; t < K; ++t, xt += dil)
                tap<0>(acc, ap, ap_next, nfrag, t * GPC, xt, t + 1 < K ? xt + dil : xt, sw, slope);
        } else {                                    // GPC == 1, RING == 2: slot parity carried across taps and phases
            int t = 0;
            if (rb) { tap<1>(acc, ap, ap_next, nfrag, 0, xt, K > 1 ? xt + dil : xt, sw, slope); ++t; xt += dil; }
            for (; t + 1 < K; t += 2, xt += 2 * dil) {
                tap<0>(acc, ap, ap_next, nfrag, t, xt, xt + dil, sw, slope);
                tap<1>(acc, ap, ap_next, nfrag, t + 1, xt + dil, t + 2 < K ? xt + 2 * dil : xt, sw, slope);
            }
            rb = 0;
            if (t < K) { tap<0>(acc, ap, ap_next, nfrag, t, xt, xt, sw, slope); rb = 1; }
        }
    }
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
    if (p.vec4) XStage<C, NTHREADS>::vec4(p.in, p.in_a, p.in_s, Xs, b, L, pos0, p.xcols, xw, tid);
    else XStage<C, NTHREADS>::scalar(p.in, p.in_a, p.in_s, Xs, b, L, pos0, p.xcols, xw, wave, lane);

    // ---- weight streams: K*GPC fragments of 1 KiB each per conv, consumed in order
    const int nfrag = K * GPC;
    const f32x4* ap1 = reinterpret_cast<const f32x4*>(p.wp1) + lane;
    const f32x4* ap2 = reinterpret_cast<const f32x4*>(p.wp2) + lane;
    typedef FusedConv<MF, NI, MF == 32 ? 4 : 2, 3> FC;
    FC fc;
    fc.start(ap1);
    __syncthreads();

    acc_t acc[NI];
    // ---- conv1 -> t1 on positions [n0 - h2, n0 - h2 + W); X column of (position, tap 0) = col + xoff
    fc.run(acc, ap1, ap2, K, Xs + hk * xw + wn0 + lr + p.xoff, xw, p.d1, slope);
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
    fc.run(acc, ap2, ap2, K, Ts + hk * tw + wn0 + lr, tw, p.d2, slope);   // (runs on into a harmless re-read of its own head)
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
        if (!XStage<MF, 64 * WN>::fits(p.xcols)) return V2W_E_SHAPE;
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
    if (p.vec4) XStage<C, NTHREADS>::vec4(p.in, p.in_a, p.in_s, Xs, b, L, pos0, p.xcols, xw, tid);
    else XStage<C, NTHREADS>::scalar(p.in, p.in_a, p.in_s, Xs, b, L, pos0, p.xcols, xw, wave, lane);

    typedef FusedConv<MF, NI, MF == 32 ? 4 : 2, 3> FC;
    FC fc;
    fc.start(reinterpret_cast<const f32x4*>(p.wp1[0]) + lane);
    __syncthreads();

    acc_t acc[NI], oacc[NI];
    for (int jb = 0; jb < p.nk; ++jb) {
        const int K = p.K[jb], d1 = p.d1[jb], d2 = p.d2[jb];
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        const f32x4* ap1 = reinterpret_cast<const f32x4*>(p.wp1[jb]) + lane;
        const f32x4* ap2 = reinterpret_cast<const f32x4*>(p.wp2[jb]) + lane;
        const f32x4* ap_after = jb + 1 < p.nk ? reinterpret_cast<const f32x4*>(p.wp1[jb + 1]) + lane : ap2;   // last: a harmless re-read

        // ---- conv1_j -> t1_j on positions [n0 - h2max, n0 - h2max + W)
        fc.run(acc, ap1, ap2, K, Xs + hk * xw + wn0 + lr + p.xoff + (p.h1max - h1), xw, d1, slope);
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
        fc.run(acc, ap2, ap_after, K, Ts + hk * tw + wn0 + lr + (p.h2max - h2), tw, d2, slope);
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
    if (!XStage<MF, 64 * WN>::fits(p.xcols)) return V2W_E_SHAPE;
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
