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

#ifdef V2W_TIMELINE   // diagnostic build only (see v2w_common.h)
V2W_TL_SETTER(v2w_timeline_set)
#endif

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
                                        int dil, float slope, int tl = -1) {   // tl: first timeline slot of the per-tap stamps (diagnostic builds)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < F::NREG; ++e) acc[j][e] = 0.f;
        const int nfrag = K * GPC;
        prime(x0, sw);
        const float* xt = x0;
        if constexpr (GPC % RING == 0) {
            for (int t = 0; t < K; ++t, xt += dil) {
                tap<0>(acc, ap, ap_next, nfrag, t * GPC, xt, t + 1 < K ? xt + dil : xt, sw, slope);
                if (tl >= 0) V2W_STAMP(tl + t);
            }
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
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(64 * WN), lds, stream, m);
    return v2w_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------
// Whole residual section of a narrow ResBlock2 stage in ONE kernel (models.py:135-141 with ResBlock2.forward inlined):
//   out = ( sum_j [ t1_j + conv_{k_j,d2_j}(lrelu(t1_j)) + b2_j ] ) / nk ,   t1_j = x + conv_{k_j,d1_j}(lrelu(x)) + b1_j
// The x tile is staged ONCE for all nk branches, every t1_j lives only in LDS, the branch sum lives in registers and is
// added in the reference's order ((r0 + r1) + r2); HBM sees one read of x and one write of out per stage.
//
// What shapes this kernel (tools/stage_timeline.py, in-kernel stamps on MI355X): the f32 MFMA shares the SIMD's vector ALU, so
// EVERY vector instruction issued inside the MFMA loop costs matrix time - an operand leaky_relu in registers (3 VALU per
// product) took 26 cycles per 64-cycle MFMA, a v_add + ds_read2_b32 pair per k-step another 10.  Hence:
//   * LDS tiles hold the ACTIVATED operands lrelu(x), lrelu(t1); the residuals (raw x, raw t1) live in registers: conv1 and conv2
//     are computed on the SAME window of W positions, so a lane's conv2 accumulator meets its own t1 (valid outputs: the window
//     minus h2max columns on each side);
//   * tiles are position-major, [position][C + pad] with the channels of a row permuted so that the four k-steps of one packed
//     weight fragment are 16 contiguous bytes: ONE ds_read_b128 (immediate offsets, no address arithmetic) feeds four MFMAs per
//     column block; row stride 36 / 24 floats makes those reads bank-conflict free;
//   * the next unit's operands (LDS) and the weight fragments (L2, scalar base + lane offset) are requested one unit / three
//     fragments ahead of the MFMAs that use them.
#define V2W_STAGE_MAXB 4
struct StageArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* wp1[V2W_STAGE_MAXB]; const float* bias1[V2W_STAGE_MAXB];
    const float* wp2[V2W_STAGE_MAXB]; const float* bias2[V2W_STAGE_MAXB];
    int K[V2W_STAGE_MAXB], d1[V2W_STAGE_MAXB], d2[V2W_STAGE_MAXB];
    float* out;
    int nk, B, L;
    int h1max, h2max;
    int xoff;          // X row of position (n0 - h2max - h1max); X row 0 sits at a position that is a multiple of 4
    int xrows;         // staged X rows (multiple of 4)
    int nto, ntl;      // valid outputs per tile, tiles per batch item
    int vec4;
    float slope, out_div;
    // fused tail (POST instantiations; models.py:143-145): y = tanh(conv_post(leaky_relu(stage output, post_slope))), post_k <= 9 taps, written as
    // (B, 1, L); `out` is not written.  A tile then advances by nadv = (nto - 2 hout) & ~3 positions and starts hout = (post_k - 1) / 2 positions
    // early (they feed the taps only); without the tail nadv = nto, hout = 0.
    const float* post_w; const float* post_b; float* post_out;
    int post_k, nadv, hout;
    float post_slope;
    // input-gradient form (BWD instantiations; the backward of the section): see v2w_stage_args::bwd_*
    const float* mask1[V2W_STAGE_MAXB]; float* mid_out[V2W_STAGE_MAXB];
    float* rowsum[V2W_STAGE_MAXB];   // optional [tiles * WN][C][2]: per (tile, wave) channel sums of dt1_j over the positions the tile owns (, 0)
    const float* mask2; const float* mask2_a; const float* mask2_s;
    float mask_slope;
};

template <int MF> struct StageGeom {
    static constexpr int RS = MF == 32 ? 36 : 24;      // floats per position row: 16-byte aligned, conflict-free ds_read_b128
    static constexpr int HMAX = 32;                    // largest conv1 halo the staging slots cover
    // LDS slot of channel c inside a position row.  The packed weight fragment (v2w_pack_mfma) pairs k-step kk, lane half hk with
    // channel 8g + 2kk + hk (32x32x2) / 4kk + hk (16x16x4); the slot order puts a lane's four k-steps side by side.
    __host__ __device__ static constexpr int slot(int c) {
        return MF == 32 ? ((c & ~7) + 4 * (c & 1) + ((c & 7) >> 1)) : (4 * (c & 3) + (c >> 2));
    }
};

// One conv phase: acc[j] = sum over taps and channels of W * (activated tile).  `x0` = this lane's float4 of tap 0, column block 0,
// unit 0; a tap advances `dil` rows, a column block MF rows, a unit 8 floats.
template <int MF, int NI>
struct StageConv {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    static constexpr int KSTEP = F::KSTEP, GPC = MF / (4 * KSTEP), RING = MF == 32 ? 4 : 2, RS = StageGeom<MF>::RS;
    static_assert(GPC % RING == 0 || (GPC == 1 && RING == 2), "ring slot of a tap's first fragment must be static");
    f32x4 ar[RING];        // weight fragments, RING - 1 in flight
    f32x4 bb[2][NI];       // operand float4s of the running and of the next unit
    int rb;                // GPC == 1: ring / operand slot of the next tap (alternates per tap, carried across phases)

    __device__ __forceinline__ void start(const f32x4* w, int lane) {
#pragma unroll
        for (int g = 0; g + 1 < RING; ++g) ar[g] = w[(size_t)g * 64 + lane];
        rb = 0;
    }
    template <int RB>
    __device__ __forceinline__ void tap(acc_t (&acc)[NI], const f32x4* wa, const f32x4* wnext, int nfrag, int g0, int lane,
                                        const float* xt, const float* xn) {
#pragma unroll
        for (int gg = 0; gg < GPC; ++gg) {
            const int g = g0 + gg + RING - 1;                                   // uniform: the stream select stays scalar
            const f32x4* wsrc = g < nfrag ? wa + (size_t)g * 64 : wnext + (size_t)(g - nfrag) * 64;
            ar[(RB + gg + RING - 1) % RING] = wsrc[lane];
            const float* src = gg + 1 < GPC ? xt + 8 * (gg + 1) : xn;
#pragma unroll
            for (int j = 0; j < NI; ++j) bb[(RB + gg + 1) & 1][j] = *reinterpret_cast<const f32x4*>(src + j * MF * RS);
            __builtin_amdgcn_sched_barrier(0);      // operand requests stay AHEAD of this unit's MFMAs
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[j] = F::mfma(ar[(RB + gg) % RING][kk], bb[(RB + gg) & 1][j][kk], acc[j]);
        }
    }
    // `brow`: this lane's NREG bias values (bias[F::row(e, hk)]): the accumulators START at the bias, so no epilogue add is needed
    __device__ __forceinline__ void run(acc_t (&acc)[NI], const float (&brow)[F::NREG], const f32x4* wa, const f32x4* wnext, int K,
                                        int lane, const float* x0, int dil, int tl = -1) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < F::NREG; ++e) acc[j][e] = brow[e];
        const int nfrag = K * GPC, step = dil * RS;
        const float* xt = x0;
        if constexpr (GPC % RING == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j) bb[0][j] = *reinterpret_cast<const f32x4*>(x0 + j * MF * RS);
            for (int t = 0; t + 1 < K; ++t, xt += step) {
                tap<0>(acc, wa, wnext, nfrag, t * GPC, lane, xt, xt + step);
                if (tl >= 0) V2W_STAMP(tl + t);
            }
            tap<0>(acc, wa, wnext, nfrag, (K - 1) * GPC, lane, xt, xt);      // (the run-on reads of the last tap are never used)
        } else {                                    // GPC == 1, RING == 2: slot parity carried across taps and phases
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x0 + j * MF * RS);
                if (rb) bb[1][j] = v; else bb[0][j] = v;
            }
            int t = 0;
            if (rb) { tap<1>(acc, wa, wnext, nfrag, 0, lane, xt, K > 1 ? xt + step : xt); ++t; xt += step; }
            for (; t + 1 < K; t += 2, xt += 2 * step) {
                tap<0>(acc, wa, wnext, nfrag, t, lane, xt, xt + step);
                tap<1>(acc, wa, wnext, nfrag, t + 1, lane, xt + step, t + 2 < K ? xt + 2 * step : xt);
            }
            rb = 0;
            if (t < K) { tap<0>(acc, wa, wnext, nfrag, t, lane, xt, xt); rb = 1; }
        }
    }
};

// BWD: the section's INPUT GRADIENT on the same tiles (backward.py; the backward of models.py:135-141).  With dr = in_a * in (= dxs / nk):
//   dt1_j = dr + lrelu'(t1_j) * conv(dr; W2_j^T, taps reversed)   ->  mid_out[j] (the weight and bias gradients of conv1_j read it),
//   dx    = sum_j dt1_j + lrelu'(x) * sum_j conv(dt1_j; W1_j^T, taps reversed)   ->  out,        x = mask2_a * mask2 + mask2_s.
// The caller hands the transposed fragment streams as wp1 (conv2's) / wp2 (conv1's) with the dilations swapped, slope = 1 (the operands are
// gradients: no activation), no biases; mask1[j] = the forward's t1_j, mask2 = the forward's xr.  dr is read ONCE for the three branches, every
// dt1_j is written once and never read back by this kernel, the branch sum stays in registers: 9 tensor passes instead of the 18 of the
// three merged launches.
template <int MF, int NI, int WN, bool POST = false, bool BWD = false>
__global__ void __launch_bounds__(64 * WN)
resblock2_stage_kernel(const StageArgs p) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    typedef StageGeom<MF> G;
    constexpr int NTHREADS = 64 * WN;
    constexpr int C = MF;
    constexpr int W = MF * NI * WN;         // positions each conv phase computes
    constexpr int RS = G::RS;
    constexpr int NR = F::NREG;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = blockIdx.x;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * p.nadv - p.hout;  // first valid output position of the tile (multiple of 4)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int wn0 = wave * (MF * NI);
    const int L = p.L;
    const float slope = p.slope;
    float* const Xa = smem;                          // [xrows][RS]  lrelu(x), exactly 0 outside the sequence
    float* const Ta = smem + p.xrows * RS;           // [W][RS]      raw x of the window at first, then lrelu(t1_j); conv2's taps reach h2max rows
                                                     //              past either end (X's tail / the pad below): read-only, feeds discarded columns
    float* const etab = Ta + (W + p.h2max) * RS;     // bias1[nk][C], bias2[nk][C]
    const int pos0 = n0 - p.h2max - p.h1max - p.xoff;   // position of X row 0
    const int xc0 = p.xoff + p.h1max;                // X row of window column 0 (position n0 - h2max)
    V2W_STAMP(0);
    for (int i = tid; i < p.nk * C; i += NTHREADS) {
        const int j = i / C, c = i - j * C;
        etab[i] = p.bias1[j] ? p.bias1[j][c] : 0.f;
        etab[V2W_STAGE_MAXB * C + i] = p.bias2[j] ? p.bias2[j][c] : 0.f;
    }
    if constexpr (POST) {         // the tail's weights as wl[c][12] (see the store phase), behind the tables: never overlaid
        float* const wl0 = etab + 2 * V2W_STAGE_MAXB * C;
        for (int i = tid; i < C * 12; i += NTHREADS) {
            const int c = i / 12, t = i - 12 * c;
            wl0[i] = t < p.post_k ? p.post_w[t * C + c] : 0.f;
        }
    }

    // ---- stage x = a*in + s: lrelu(x) into Xa (all rows), raw x of the window into Ta (the residual of every branch).
    // A thread takes 4 channels x 4 positions at a time (all its loads are issued before the first is consumed).  The 4 channels are
    // the ones whose slots are ADJACENT in a position row - 8g + h + {0, 2, 4, 6} (32 channels) / b + {0, 4, 8, 12} (16) - so a block
    // is one 16-byte LDS store per position; consecutive lanes take the channel quads of one position quad first (a full row =
    // conflict-free stores, 128-byte pieces of 8 / 4 rows on the global side), then the next position quad.
    constexpr int NCQ = C / 4;                      // channel quads
    auto quad_ch = [&](int cq, int i) { return MF == 32 ? 8 * (cq >> 1) + (cq & 1) + 2 * i : cq + 4 * i; };   // i-th channel of quad cq
    if (p.vec4) {
        constexpr int NPF = (NCQ * ((W + 2 * G::HMAX + 8) / 4) + NTHREADS - 1) / NTHREADS;
        const int xr4 = p.xrows >> 2;
        f32x4 g[NPF][4];
        float av[NPF][4], sv[NPF][4];
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTHREADS;
            const int cq = idx % NCQ, pq = idx / NCQ;
            const int pos = pos0 + pq * 4;
            const bool ok = pq < xr4 && pos >= 0 && pos < L;      // L % 4 == 0, pos % 4 == 0: whole float4 inside
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g[s][i] = f32x4{0.f, 0.f, 0.f, 0.f}; av[s][i] = 1.f; sv[s][i] = 0.f;
                if (ok) {
                    const int ch = b * C + quad_ch(cq, i);
                    g[s][i] = *reinterpret_cast<const f32x4*>(p.in + (size_t)ch * L + pos);
                    if (p.in_a) { av[s][i] = p.in_a[ch]; sv[s][i] = p.in_s[ch]; }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTHREADS;
            const int cq = idx % NCQ, pq = idx / NCQ;
            if (pq >= xr4) continue;
            const int r0 = pq * 4;
            const bool ok = pos0 + r0 >= 0 && pos0 + r0 < L;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 raw, act;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    raw[i] = ok ? fmaf(av[s][i], g[s][i][e], sv[s][i]) : 0.f;
                    act[i] = v2w_lrelu(raw[i], slope);
                }
                const int r = r0 + e;
                *reinterpret_cast<f32x4*>(Xa + r * RS + 4 * cq) = act;       // slots 4cq .. 4cq + 3 = the quad's channels, in order
                if (r >= xc0 && r < xc0 + W) *reinterpret_cast<f32x4*>(Ta + (r - xc0) * RS + 4 * cq) = raw;
            }
        }
    } else {                                    // any L / alignment: one element at a time
        for (int i = tid; i < C * p.xrows; i += NTHREADS) {
            const int c = i / p.xrows, r = i - c * p.xrows, pos = pos0 + r;
            float raw = 0.f;
            if (pos >= 0 && pos < L) {
                const int ch = b * C + c;
                raw = fmaf(p.in_a ? p.in_a[ch] : 1.f, p.in[(size_t)ch * L + pos], p.in_s ? p.in_s[ch] : 0.f);
            }
            Xa[r * RS + G::slot(c)] = v2w_lrelu(raw, slope);
            if (r >= xc0 && r < xc0 + W) Ta[(r - xc0) * RS + G::slot(c)] = raw;
        }
    }

    typedef StageConv<MF, NI> SC;
    SC sc;
    sc.start(reinterpret_cast<const f32x4*>(p.wp1[0]), lane);
    V2W_STAMP(1);
    __syncthreads();
    V2W_STAMP(2);

    // accumulator row `reg` of this lane = channel co = F::row(reg, hk); its slot in a position row:
    //   32 channels: co = 8g + 4hk + r  ->  slot 8g + 2hk + (r >> 1) + 4(r & 1): (r0, r2) and (r1, r3) are 8-byte pairs
    //   16 channels: co = 4hk + reg     ->  slot 4reg + hk
    auto row_get = [&](const float* row, float (&v)[NR]) {
        if constexpr (MF == 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x2 a = *reinterpret_cast<const f32x2*>(row + 8 * g + 2 * hk);
                const f32x2 c = *reinterpret_cast<const f32x2*>(row + 8 * g + 2 * hk + 4);
                v[4 * g] = a[0]; v[4 * g + 2] = a[1]; v[4 * g + 1] = c[0]; v[4 * g + 3] = c[1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) v[r] = row[4 * r + hk];
        }
    };
    auto row_put = [&](float* row, const float (&v)[NR]) {
        if constexpr (MF == 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<f32x2*>(row + 8 * g + 2 * hk) = f32x2{v[4 * g], v[4 * g + 2]};
                *reinterpret_cast<f32x2*>(row + 8 * g + 2 * hk + 4) = f32x2{v[4 * g + 1], v[4 * g + 3]};
            }
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) row[4 * r + hk] = v[r];
        }
    };

    // bias of this lane's accumulator rows: F::row(e, hk) is 4 consecutive channels per register quad (32 ch.) / all of them (16)
    auto bias_rows = [&](const float* tab, float (&v)[NR]) {
        if constexpr (MF == 32) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(tab + 8 * g + 4 * hk);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * g + r] = q[r];
            }
        } else {
            const f32x4 q = *reinterpret_cast<const f32x4*>(tab + 4 * hk);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = q[r];
        }
    };
    float xres[NI][NR];                       // raw x at this lane's outputs: the residual of conv1 in every branch
#pragma unroll
    for (int j = 0; j < NI; ++j) row_get(Ta + (wn0 + j * MF + lr) * RS, xres[j]);

    // BWD: the sign of x = a * xr + s at this lane's outputs, one bit per accumulator register (lrelu'(x) of the final sum)
    unsigned m2bits = 0u;
    if constexpr (BWD) {
        static_assert(NI * NR <= 32, "one mask bit per accumulator register");
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int pos = n0 - p.h2max + wn0 + j * MF + lr;
            if (pos >= 0 && pos < L) {
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    const int ch = b * C + F::row(e, hk);
                    const float xv = fmaf(p.mask2_a ? p.mask2_a[ch] : 1.f, p.mask2[(size_t)ch * L + pos], p.mask2_a ? p.mask2_s[ch] : 0.f);
                    m2bits |= (xv > 0.f ? 1u : 0u) << (j * NR + e);
                }
            }
        }
    }
    acc_t acc[NI];
    float t1r[NI][NR], oacc[NI][NR];
    const float* const xl = Xa + (wn0 + lr) * RS + 4 * hk;     // this lane's float4 in X / T1 row (window column) 0
    const float* const tl = Ta + (wn0 + lr) * RS + 4 * hk;
    for (int jb = 0; jb < p.nk; ++jb) {
        const int K = p.K[jb], d1 = p.d1[jb], d2 = p.d2[jb];
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        const f32x4* w1 = reinterpret_cast<const f32x4*>(p.wp1[jb]);
        const f32x4* w2 = reinterpret_cast<const f32x4*>(p.wp2[jb]);
        const f32x4* wafter = jb + 1 < p.nk ? reinterpret_cast<const f32x4*>(p.wp1[jb + 1]) : w2;   // last: a harmless re-read

        // ---- conv1_j on the window: column col <-> position n0 - h2max + col
        float brow[NR];
        bias_rows(etab + jb * C, brow);
        if constexpr (BWD) {
            // the forward's t1_j at this lane's accumulator elements (the mask of the first epilogue): issued here, into the registers of
            // t1r - dead until that epilogue - so that the 2 x NR loads land under the conv's MFMA loop
            const float* const mk = p.mask1[jb] + (size_t)b * C * L;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pos = n0 - p.h2max + wn0 + j * MF + lr;
                const bool in_seq = pos >= 0 && pos < L;
#pragma unroll
                for (int e = 0; e < NR; ++e) t1r[j][e] = in_seq ? mk[(size_t)F::row(e, hk) * L + pos] : 1.f;
            }
        }
        sc.run(acc, brow, w1, w2, K, lane, xl + (xc0 - h1) * RS, d1, jb == 0 ? 16 : (jb == 2 ? 19 : -1));
        V2W_STAMP(3 + 4 * jb);
        if constexpr (BWD) {
            // dt1_j = dr + lrelu'(t1_j) * acc: the forward's t1_j comes from global memory at the accumulator's own (channel, position) - 128
            // contiguous bytes per register and lane half - and dt1_j goes back the same way, each position from the tile that owns it
            float* const mo = p.mid_out[jb] + (size_t)b * C * L;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn0 + j * MF + lr, pos = n0 - p.h2max + col;
                const bool in_seq = pos >= 0 && pos < L;
                const bool own = in_seq && col >= p.h2max && col < p.h2max + p.nto;
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    const float v = in_seq ? xres[j][e] + (t1r[j][e] > 0.f ? acc[j][e] : acc[j][e] * p.mask_slope) : 0.f;
                    t1r[j][e] = v;
                    if (own) mo[(size_t)F::row(e, hk) * L + pos] = v;
                }
            }
            if (p.rowsum[jb]) {
                // the bias gradient of conv1_j = the channel sums of dt1_j: this wave's share, one row per (tile, wave), reduced by the caller
                // (v2w_bn_reduce_partials) in a fixed order
                float* const rs = p.rowsum[jb] + ((size_t)(tile * WN + wave) * C) * 2;
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int col = wn0 + j * MF + lr, pos = n0 - p.h2max + col;
                        if (pos >= 0 && pos < L && col >= p.h2max && col < p.h2max + p.nto) sum += t1r[j][e];
                    }
#pragma unroll
                    for (int off = MF / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);      // over the MF lanes that share (e, hk)
                    if (lr == 0) *reinterpret_cast<f32x2*>(rs + 2 * F::row(e, hk)) = f32x2{sum, 0.f};
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int pos = n0 - p.h2max + wn0 + j * MF + lr;
            const bool in_seq = pos >= 0 && pos < L;            // conv2 zero-pads t1 outside the sequence
#pragma unroll
            for (int e = 0; e < NR; ++e) t1r[j][e] = in_seq ? acc[j][e] + xres[j][e] : 0.f;
        }
        }
        __syncthreads();                      // everyone is done reading Ta (raw x at jb == 0, the previous branch's t1 after)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            float a[NR];
#pragma unroll
            for (int e = 0; e < NR; ++e) a[e] = v2w_lrelu(t1r[j][e], slope);
            row_put(Ta + (wn0 + j * MF + lr) * RS, a);
        }
        __syncthreads();
        V2W_STAMP(4 + 4 * jb);

        // ---- conv2_j on the same window ; r_j = (acc + b2) + t1_j ; branch sum in the reference's order
        bias_rows(etab + V2W_STAGE_MAXB * C + jb * C, brow);
        sc.run(acc, brow, w2, wafter, K, lane, tl - h2 * RS, d2);
        V2W_STAMP(5 + 4 * jb);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                float r = acc[j][e];
                if constexpr (BWD) r = (m2bits >> (j * NR + e)) & 1u ? r : r * p.mask_slope;     // lrelu'(x): the same mask for every branch
                r += t1r[j][e];
                oacc[j][e] = jb == 0 ? r : oacc[j][e] + r;
            }
        V2W_STAMP(6 + 4 * jb);
    }

    // ---- store: the branch sums go through an LDS scratch [C][W + 4] (T1 is dead) laid out so that the first valid output
    // (window column h2max = position n0, a multiple of 4) sits at a 16-byte aligned scratch column; all threads then store float4s
    // along positions: a quarter of the store instructions of the row-per-register form, 256-byte segments.
    __syncthreads();                          // every wave is done reading T1
    {
        constexpr int SRS = W + 4;
        float* const scr = Ta;
        const int soff = (p.h2max + 3) & ~3;          // scratch column of window column h2max: 16-byte aligned
        const float dinv = p.out_div != 0.f ? 1.f / p.out_div : 1.f;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int sc = wn0 + j * MF + lr - p.h2max + soff;      // scratch column of this lane's window column
            if constexpr (POST) {
                // the tail's operand z = leaky_relu(out / nk, post_slope), exactly 0 outside the sequence (conv_post zero-pads)
                const int pos = n0 - p.h2max + wn0 + j * MF + lr;
                const bool in_seq = pos >= 0 && pos < L;
#pragma unroll
                for (int e = 0; e < NR; ++e) {
                    float v = oacc[j][e];
                    if (p.out_div != 0.f) v = v2w_div_by(v, p.out_div, dinv);
                    scr[F::row(e, hk) * SRS + sc] = in_seq ? v2w_lrelu(v, p.post_slope) : 0.f;
                }
            } else {
#pragma unroll
            for (int e = 0; e < NR; ++e) scr[F::row(e, hk) * SRS + sc] = oacc[j][e];
            }
        }
        __syncthreads();
        if constexpr (POST) {
            // y[p] = tanh(b + sum_c sum_t w[t][c] z[c][p + t - hout]) (hout = (post_k - 1) / 2) for the nadv positions p = n0 + hout + m: scratch
            // column of z[c][p + t - hout] is soff + m + t.  Thread (output quad q, channel group cg) adds the taps of 4 channels for 4
            // consecutive outputs - per channel the three aligned float4s from scratch column soff + 4 q hold the <= 12 values it needs, the
            // weights come from the table wl[c][12] the kernel's first instructions put behind the bias tables - the four groups meet through LDS
            // and the quad's first thread finishes: bias, tanh, one float4 store.  (One thread per quad walking all 16 channels with the
            // weights as scalar loads: 12 k cycles per tile, 100 us of a 585 us kernel.)
            const int PK = p.post_k, nq = p.nadv >> 2;
            float* const part = scr + C * SRS;                       // [4][nq] float4
            const float* const wl = etab + 2 * V2W_STAGE_MAXB * C;   // [C][12]: taps 0 .. post_k - 1, zeros behind them (filled at the kernel's start)
            const int qd = tid % nq, cg = tid / nq;
            const int p0 = n0 + p.hout + 4 * qd;
            if (cg < 4 && p0 < L) {
                f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int cc = 0; cc < C / 4; ++cc) {
                    const int c = (C / 4) * cg + cc;
                    const float* zr = scr + c * SRS + soff + 4 * qd;
                    float z[12], wv[12];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 4 * u);
                        const f32x4 ww = *reinterpret_cast<const f32x4*>(wl + 12 * c + 4 * u);
                        z[4 * u] = zz[0]; z[4 * u + 1] = zz[1]; z[4 * u + 2] = zz[2]; z[4 * u + 3] = zz[3];
                        wv[4 * u] = ww[0]; wv[4 * u + 1] = ww[1]; wv[4 * u + 2] = ww[2]; wv[4 * u + 3] = ww[3];
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int x = 0; x < 4; ++x) y[x] = fmaf(wv[t], z[t + x], y[x]);     // (taps past post_k: zero weights)
                }
                *reinterpret_cast<f32x4*>(part + (cg * nq + qd) * 4) = y;
            }
            __syncthreads();
            if (cg == 0 && p0 < L) {
                const float pb = p.post_b ? p.post_b[0] : 0.f;
                f32x4 y = *reinterpret_cast<const f32x4*>(part + qd * 4);
#pragma unroll
                for (int g2 = 1; g2 < 4; ++g2) y += *reinterpret_cast<const f32x4*>(part + (g2 * nq + qd) * 4);
#pragma unroll
                for (int x = 0; x < 4; ++x) y[x] = tanhf(y[x] + pb);
                float* dst = p.post_out + (size_t)b * L + p0;
                if (p.vec4) {
                    *reinterpret_cast<f32x4*>(dst) = y;
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (p0 + x < L) dst[x] = y[x];
                }
            }
            return;
        }
        const int nq = p.nto >> 2;
        const unsigned magic = (unsigned)(((1ull << 32) + nq - 1) / nq);
        for (int idx = tid; idx < C * nq; idx += NTHREADS) {
            const int row = (int)__umulhi((unsigned)idx, magic), q = idx - row * nq;
            const int pos = n0 + 4 * q;
            if (pos >= L) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * SRS + soff + 4 * q);
            if (p.out_div != 0.f) {
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], p.out_div, dinv);
            }
            float* dst = p.out + ((size_t)b * C + row) * L + pos;
            if (p.vec4) {
                *reinterpret_cast<f32x4*>(dst) = v;              // L % 4 == 0: whole float4 inside
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (pos + x < L) dst[x] = v[x];
            }
        }
    }
    V2W_STAMP(15);
}

template <int MF, int NI, int WN>
int launch_stage(const v2w_stage_args* q, hipStream_t stream) {
    const bool post = q->post_out != nullptr;
    typedef StageGeom<MF> G;
    constexpr int W = MF * NI * WN;
    StageArgs p{};
    p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.out = q->out;
    p.nk = q->nk; p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div;
    for (int j = 0; j < q->nk; ++j) {
        p.wp1[j] = q->wp1[j]; p.bias1[j] = q->bias1[j]; p.wp2[j] = q->wp2[j]; p.bias2[j] = q->bias2[j];
        p.K[j] = q->k[j]; p.d1[j] = q->dil1[j]; p.d2[j] = q->dil2[j];
        if (q->k[j] < 3) return V2W_E_SHAPE;             // the weight ring runs three fragments ahead into the next stream
        const int h1 = q->dil1[j] * (q->k[j] - 1) / 2, h2 = q->dil2[j] * (q->k[j] - 1) / 2;
        if (h1 > p.h1max) p.h1max = h1;
        if (h2 > p.h2max) p.h2max = h2;
    }
    p.nto = (W - 2 * p.h2max) & ~3;
    if (p.nto < W / 2 || p.h1max > G::HMAX) return V2W_E_SHAPE;
    p.nadv = p.nto; p.hout = 0;
    if (post) {                     // the generator's tail behind the stage: C -> 1 channels, an odd tap count <= 9, fp32 (B, 1, L) output
        if (!q->post_w || q->post_k < 1 || q->post_k > 9 || !(q->post_k & 1) || MF != 16) return V2W_E_SHAPE;
        p.post_w = q->post_w; p.post_b = q->post_b; p.post_out = q->post_out; p.post_k = q->post_k; p.post_slope = q->post_slope;
        p.hout = (q->post_k - 1) / 2; p.nadv = (p.nto - 2 * p.hout) & ~3;
        if (p.nadv < W / 4) return V2W_E_SHAPE;
    }
    const int hsum = p.h1max + p.h2max + p.hout;       // (X row 0 sits at a position that is a multiple of 4: n0 + hout is one)
    p.xoff = ((hsum + 3) & ~3) - hsum;
    p.xrows = (p.xoff + W + 2 * p.h1max + 3) & ~3;
    if (p.xrows < p.h2max) return V2W_E_SHAPE;
    if (MF * (W + 4) > (W + p.h2max) * G::RS) return V2W_E_SHAPE;    // the store scratch [C][W + 4] overlays the T1 tile
    p.ntl = (q->L + p.nadv - 1) / p.nadv;
    p.vec4 = (q->L % 4 == 0) && ((reinterpret_cast<uintptr_t>(q->in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(q->out) & 15) == 0) &&
             ((reinterpret_cast<uintptr_t>(q->post_out) & 15) == 0) &&
             ((reinterpret_cast<uintptr_t>(q->in_a) & 15) == 0) && ((reinterpret_cast<uintptr_t>(q->in_s) & 15) == 0);
    size_t lds = ((size_t)(p.xrows + W + p.h2max) * G::RS + 2 * V2W_STAGE_MAXB * MF + (post ? MF * 12 : 0)) * sizeof(float);
    if (post && (size_t)MF * (W + 4) + 4 * (p.nadv >> 2) * 4 > (size_t)(W + p.h2max) * G::RS) return V2W_E_SHAPE;     // the partial sums sit behind the scratch, inside the T1 tile
    if (post && 4 * (p.nadv >> 2) > 64 * WN) return V2W_E_SHAPE;                                                       // one thread per (output quad, channel group)
#ifdef V2W_TIMELINE
    if (const char* e = getenv("V2W_TL_LDSPAD")) lds += (size_t)atoi(e);      // fewer workgroups per CU: what does ONE wave per SIMD reach?
#endif
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    const bool bwd = q->bwd_mask2 != nullptr;
    if (bwd) {
        if (post || q->slope != 1.f) return V2W_E_ARG;
        for (int j = 0; j < q->nk; ++j) {
            if (!q->bwd_mask1[j] || !q->bwd_mid[j] || q->bias1[j] || q->bias2[j]) return V2W_E_ARG;
            p.mask1[j] = q->bwd_mask1[j]; p.mid_out[j] = q->bwd_mid[j]; p.rowsum[j] = q->bwd_rowsum[j];
        }
        p.mask2 = q->bwd_mask2; p.mask2_a = q->bwd_mask2_a; p.mask2_s = q->bwd_mask2_s; p.mask_slope = q->bwd_slope;
    }
    auto kern = bwd ? resblock2_stage_kernel<MF, NI, WN, false, true>
                    : (post ? resblock2_stage_kernel<MF, NI, WN, (MF == 16)> : resblock2_stage_kernel<MF, NI, WN, false>);
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(q->B * p.ntl), dim3(64 * WN), lds, stream, p);
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

// Rows of v2w_stage_args::bwd_rowsum the input-gradient form writes for this problem (one per (tile, wave)): the launcher's own geometry.
extern "C" int v2w_resblock2_stage_bwd_rows(const v2w_stage_args* a) {
    if (!a || a->nk < 1 || a->nk > V2W_STAGE_MAXB || a->B <= 0 || a->L <= 0 || (a->C != 16 && a->C != 32)) return 0;
    int h2max = 0;
    for (int j = 0; j < a->nk; ++j) { const int h2 = a->dil2[j] * (a->k[j] - 1) / 2; if (h2 > h2max) h2max = h2; }
    const int nto256 = (256 - 2 * h2max) & ~3;
    const bool small = nto256 > 0 && (long long)a->B * ((a->L + nto256 - 1) / nto256) < 224 && 128 - 2 * h2max >= 64;
    const int nto = ((small ? 128 : 256) - 2 * h2max) & ~3;
    if (nto <= 0) return 0;
    const long long rows = (long long)a->B * ((a->L + nto - 1) / nto) * 4;
    return rows > 0x7fffffffll ? 0 : (int)rows;
}

extern "C" int v2w_resblock2_stage_fwd(const v2w_stage_args* a, void* stream) {
    if (!a || !a->in || (!a->out && !a->post_out) || a->nk < 1 || a->nk > V2W_STAGE_MAXB) return V2W_E_ARG;
    if (a->B <= 0 || a->C <= 0 || a->L <= 0) return V2W_E_ARG;
    if ((a->in_a == nullptr) != (a->in_s == nullptr) || (a->bwd_mask2_a == nullptr) != (a->bwd_mask2_s == nullptr)) return V2W_E_ARG;
    if (a->bwd_mask2 && (!a->out || a->post_out)) return V2W_E_ARG;
    if (a->post_out && a->C != 16) return V2W_E_SHAPE;           // the fused tail: the last (16-channel) stage
    for (int j = 0; j < a->nk; ++j) {
        if (!a->wp1[j] || !a->wp2[j] || a->k[j] <= 0 || a->dil1[j] <= 0 || a->dil2[j] <= 0) return V2W_E_ARG;
        if ((a->k[j] & 1) == 0) return V2W_E_SHAPE;
    }
    hipStream_t st = (hipStream_t)stream;
    // latency sizes (inference at B = 1: fewer windows of 256 positions than CUs - a tile's six convs run one after the other on a
    // quarter of the chip): windows of 128 positions, twice the workgroups for 23 % more halo work
    int h2max = 0;
    for (int j = 0; j < a->nk; ++j) { const int h2 = a->dil2[j] * (a->k[j] - 1) / 2; if (h2 > h2max) h2max = h2; }
    const int nto = (256 - 2 * h2max) & ~3;
    const bool small = nto > 0 && (long long)a->B * ((a->L + nto - 1) / nto) < 224 && 128 - 2 * h2max >= 64;
    if (a->C == 32) return small ? launch_stage<32, 1, 4>(a, st) : launch_stage<32, 2, 4>(a, st);
    if (a->C == 16) return small ? launch_stage<16, 2, 4>(a, st) : launch_stage<16, 4, 4>(a, st);
    return V2W_E_SHAPE;
}
