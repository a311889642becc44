// Implicit-GEMM Conv1d / ConvTranspose1d on the gfx950 f32 MFMA pipe.
//
// One kernel template serves both ops of the Vec2Wav generator:
//   conv  (U = 1): out[b,co,l]     = bias[co] + sum_{ci,t} W[t][ci][co] * act(in[b,ci,l + (t-(k-1)/2)*dil])
//                  (models.py:37-44, 65-70, 123 of the reference, with the leaky_relu, the folded CondBN
//                  affine, the residual add, the `xs +=` accumulation and the `/num_kernels` fused in)
//   convT (U > 1): out[b,co,U*q+r] = bias[co] + sum_{ci,m} W[t0_r+m*U][ci][co] * act(in[b,ci,q + c_r - m])
//                  t0_r = (r+pad)%U, c_r = (r+pad)/U  - the polyphase form of ConvTranspose1d(k,U,pad)
//                  (models.py:128-129).
//
// GEMM view per workgroup: M = MT output channels, N = NT input-rate positions, K = C_in x taps.
// A (weights)  : never staged.  v2w_pack_mfma() stores them in MFMA A-fragment order, so one fully coalesced
//                global_load_dwordx4 per lane (1 KiB per wave) feeds four consecutive MFMA k-steps; the panel is
//                L2-resident (<= 2.9 MB per layer) and the next fragment is fetched while the current one computes.
// B (signal)   : LDS tile Xs[c][NT + halo], double buffered over C_in chunks.  Lanes read consecutive positions
//                (conflict-free ds_read_b32); every tap is a column offset into the SAME tile, so each input element
//                is fetched from HBM once per M-tile and the activation / CondBN affine is applied once, at staging
//                time.  The next chunk travels global -> registers while the current one computes (one barrier/chunk).
// MFMA         : v_mfma_f32_32x32x2_f32 (C_out >= 32) or v_mfma_f32_16x16x4_f32 (C_out == 16): exact fp32
//                (bit-identical to an fmaf chain), 64 FLOP/clk/SIMD.
// Waves        : WM x WN waves per workgroup, each owning (MF*MI) x (MF*NI) outputs for each of the U phases;
//                64-lane fragments: lane&(MF-1) = row/col inside the MFMA tile, lane/MF = k index.
#include "v2w_common.h"

namespace {

struct TileArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* wp; const float* bias;
    const float* res; const float* res_a; const float* res_s;
    float* out;
    int B, Cin, Cout, L, K, dil;
    int pad;      // convT only: (K-U)/2
    int hl, hr;   // halo (input positions) left / right of the tile
    int hla;      // hl rounded up to a multiple of 4: LDS column 0 <-> position n0 - hla (16-B aligned rows)
    int xw;       // LDS row stride of the input tile (floats)
    int xcols;    // columns actually staged (multiple of 4)
    int vec4;     // 1: L % 4 == 0 and 16-B aligned base -> float4 staging with register prefetch
    int ntl;      // position tiles per batch item
    int ntiles;   // B * ntl
    float slope;
    int accumulate;
    float out_div;
};

template <int MF> struct Frag;
template <> struct Frag<32> {
    typedef f32x16 acc_t;
    static constexpr int NREG = 16, KSTEP = 2;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    __device__ static __forceinline__ int row(int reg, int hk) { return (reg & 3) + 8 * (reg >> 2) + 4 * hk; }
};
template <> struct Frag<16> {
    typedef f32x4 acc_t;
    static constexpr int NREG = 4, KSTEP = 4;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane&15, row = (lane>>4)*4 + reg
    __device__ static __forceinline__ int row(int reg, int hk) { return hk * 4 + reg; }
};

template <int MF, int U, int MI, int NI, int WM, int WN, int CK, int NPF>
__global__ void __launch_bounds__(64 * WM * WN)
conv_tile_kernel(const TileArgs p) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int MT = MF * MI * WM;
    constexpr int NT = MF * NI * WN;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int KSTEP = F::KSTEP;
    constexpr int CKG = 4 * KSTEP;          // channels covered by one packed A fragment (4 k-steps)
    static_assert(CK % CKG == 0, "chunk must hold whole A fragments");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [CK][xw]

    // ---- which tile: ids that differ by a multiple of 8 tend to share an XCD (and its L2), so the M-tiles that
    // re-read the same input tile are placed 8 apart (speed only, never correctness).
    const int mtiles = p.Cout / MT;
    const int id = blockIdx.x;
    const int grp = id / (8 * mtiles), rem = id % (8 * mtiles);
    const int mt = rem >> 3;
    const int tile = grp * 8 + (rem & 7);
    if (tile >= p.ntiles) return;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * NT;   // first input-rate position of the tile
    const int m0 = mt * MT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & (MF - 1);       // row (A) / column (B, D) inside the MFMA tile
    const int hk = lane / MF;             // k index inside the MFMA k-step
    const int wm0 = (wave / WN) * (MF * MI);
    const int wn0 = (wave % WN) * (MF * NI);

    acc_t acc[U][MI][NI];
#pragma unroll
    for (int r = 0; r < U; ++r)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < F::NREG; ++e) acc[r][i][j][e] = 0.f;

    const int xw = p.xw;
    const int L = p.L, K = p.K;
    const float slope = p.slope;
    const int nch = p.Cin / CK;
    const int G = p.Cin / CKG;             // packed A fragments per tap
    const int xw4 = p.xcols >> 2;          // float4 columns staged per row
    const int pos0 = n0 - p.hla;           // position of LDS column 0

    // ---- staging slots of this thread (vec4 path): slot s covers float4 #(tid + s*NTHREADS) of the [CK][xw4] chunk image
    int s_row[NPF], s_col[NPF];
    bool s_ok[NPF];
#pragma unroll
    for (int s = 0; s < NPF; ++s) {
        const int idx = tid + s * NTHREADS;
        s_row[s] = idx / xw4;
        s_col[s] = (idx % xw4) * 4;
        s_ok[s] = idx < CK * xw4;
    }
    f32x4 pf[NPF];
    float pf_a[NPF], pf_s[NPF];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto prefetch = [&](int ci0) {       // global -> registers, no wait
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int pos = pos0 + s_col[s];
            pf[s] = zero4; pf_a[s] = 1.f; pf_s[s] = 0.f;
            if (s_ok[s] && pos >= 0 && pos < L) {
                const int ch = b * p.Cin + ci0 + s_row[s];
                pf[s] = *reinterpret_cast<const f32x4*>(p.in + (size_t)ch * L + pos);
                if (p.in_a) { pf_a[s] = p.in_a[ch]; pf_s[s] = p.in_s[ch]; }
            }
        }
    };
    auto commit = [&](float* Xs) {   // registers -> activation -> LDS
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            if (!s_ok[s]) continue;
            const float av = pf_a[s], sv = pf_s[s];
            const int pos = pos0 + s_col[s];
            f32x4 v = zero4;
            if (pos >= 0 && pos < L) {   // whole float4 inside (L % 4 == 0, pos % 4 == 0): padding stays exactly 0
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v2w_lrelu(fmaf(av, pf[s][e], sv), slope);
            }
            *reinterpret_cast<f32x4*>(Xs + s_row[s] * xw + s_col[s]) = v;
        }
    };
    auto stage_scalar = [&](int ci0, float* Xs) {   // any L / alignment: dword loads straight into LDS
        for (int c = wave; c < CK; c += WM * WN) {
            const int ch = b * p.Cin + ci0 + c;
            const float* src = p.in + (size_t)ch * L;
            const float av = p.in_a ? p.in_a[ch] : 1.f;
            const float sv = p.in_s ? p.in_s[ch] : 0.f;
            for (int j = lane; j < p.xcols; j += 64) {
                const int l = pos0 + j;
                float v = 0.f;
                if (l >= 0 && l < L) v = v2w_lrelu(fmaf(av, src[l], sv), slope);
                Xs[c * xw + j] = v;
            }
        }
    };

    // ---- packed weights: float4 index = ((mb*K + t)*G + g)*64 + lane, mb = 32- or 16-row block of output channels
    const f32x4* wp4 = reinterpret_cast<const f32x4*>(p.wp);
    const int mb0 = (m0 + wm0) / MF;
    auto load_a = [&](f32x4 (&a)[MI], int t, int g) {
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = wp4[((size_t)((mb0 + i) * K + t) * G + g) * 64 + lane];
    };

    // ---- prologue: chunk 0 into buffer 0
    if (p.vec4) { prefetch(0); commit(smem); }
    else stage_scalar(0, smem);
    __syncthreads();

    const int colbase = wn0 + lr + p.hla;   // LDS column of this lane's output position (tap offset added per tap)
    constexpr int GPC = CK / CKG;           // A fragments per chunk and tap
    auto first_tap = [&](int r) { return U == 1 ? 0 : (r + p.pad) % U; };
    f32x4 a_cur[MI], a_nxt[MI];
    load_a(a_cur, first_tap(0), 0);
    for (int ch = 0; ch < nch; ++ch) {
        float* Xs = smem + (ch & 1) * (CK * xw);
        float* Xn = smem + ((ch + 1) & 1) * (CK * xw);
        const bool more = ch + 1 < nch;
        if (more && p.vec4) prefetch((ch + 1) * CK);   // in flight during the MFMA phase below
        const int g0 = ch * GPC;

#pragma unroll
        for (int r = 0; r < U; ++r) {
            int t0, tstr, d0, dstr, nt;
            if (U == 1) { t0 = 0; tstr = 1; d0 = -p.hl; dstr = p.dil; nt = K; }
            else { const int rp = r + p.pad; t0 = rp % U; tstr = U; d0 = rp / U; dstr = -1; nt = (K - t0 + U - 1) / U; }
            // flattened (tap m, fragment gg) loop; the NEXT A fragment (next iteration, next phase or next chunk)
            // is always in flight while the current one feeds the MFMAs
            const int nit = nt * GPC;
            int m = 0, gg = 0;
            for (int it = 0; it < nit; ++it) {
                int m2 = m, gg2 = gg + 1;
                if (gg2 == GPC) { gg2 = 0; ++m2; }
                if (it + 1 < nit) load_a(a_nxt, t0 + m2 * tstr, g0 + gg2);
                else if (r + 1 < U) load_a(a_nxt, first_tap(r + 1), g0);
                else if (more) load_a(a_nxt, first_tap(0), g0 + GPC);
                const float* xrow = Xs + colbase + d0 + m * dstr + (gg * CKG + hk) * xw;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    float bb[NI];
#pragma unroll
                    for (int j = 0; j < NI; ++j) bb[j] = xrow[kk * KSTEP * xw + j * MF];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[r][i][j] = F::mfma(a_cur[i][kk], bb[j], acc[r][i][j]);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) a_cur[i] = a_nxt[i];
                m = m2; gg = gg2;
            }
        }

        if (more) {
            if (p.vec4) commit(Xn);
            else stage_scalar((ch + 1) * CK, Xn);
        }
        __syncthreads();   // Xn complete for the next iteration; everyone done with Xs before it is overwritten again
    }

    // ---- epilogue: + bias [+ residual] [+ out] [/ out_div]; the U phases of one (co, q) are U consecutive floats.
    const int Lout = L * U;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int co = m0 + wm0 + i * MF + F::row(e, hk);
            const float bias = p.bias ? p.bias[co] : 0.f;
            const size_t orow = ((size_t)b * p.Cout + co) * Lout;
            float ra = 1.f, rs = 0.f;
            if (U == 1 && p.res_a) { ra = p.res_a[b * p.Cout + co]; rs = p.res_s[b * p.Cout + co]; }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int q = n0 + wn0 + j * MF + lr;
                if (q >= L) continue;
                if constexpr (U == 1) {
                    float v = acc[0][i][j][e] + bias;
                    if (p.res) v += fmaf(ra, p.res[orow + q], rs);
                    if (p.accumulate) v += p.out[orow + q];
                    if (p.out_div != 0.f) v = v / p.out_div;
                    p.out[orow + q] = v;
                } else if constexpr (U == 2) {
                    f32x2 v; v[0] = acc[0][i][j][e] + bias; v[1] = acc[1][i][j][e] + bias;
                    *reinterpret_cast<f32x2*>(p.out + orow + (size_t)q * 2) = v;
                } else if constexpr (U == 4) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[r][i][j][e] + bias;
                    *reinterpret_cast<f32x4*>(p.out + orow + (size_t)q * 4) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < U; ++r) p.out[orow + (size_t)q * U + r] = acc[r][i][j][e] + bias;
                }
            }
        }
    }
}

template <int MF, int U, int MI, int NI, int WM, int WN, int CK>
int launch_tile(TileArgs p, hipStream_t stream) {
    constexpr int MT = MF * MI * WM, NT = MF * NI * WN, NTHREADS = 64 * WM * WN;
    constexpr int HMAX = 32;                                       // largest halo (each side) the slot count covers
    constexpr int NPF = (CK * ((NT + 2 * HMAX) / 4) + NTHREADS - 1) / NTHREADS;
    if (p.Cout % MT != 0 || p.Cin % CK != 0) return V2W_E_SHAPE;
    p.hla = (p.hl + 3) & ~3;
    if (p.hla > HMAX || p.hr > HMAX) return V2W_E_SHAPE;
    p.ntl = (p.L + NT - 1) / NT;
    p.ntiles = p.B * p.ntl;
    p.xcols = (p.hla + NT + p.hr + 3) & ~3;
    int xw = p.xcols;
    if (MF == 16) xw += ((16 - xw % 32) + 32) % 32;  // xw % 32 == 16: the two 16-lane k-groups of a half-wave hit disjoint banks
    p.xw = xw;
    p.vec4 = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0);
    const size_t lds = (size_t)2 * CK * xw * sizeof(float);
    auto kern = conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF>;
    const int mtiles = p.Cout / MT;
    const int grid = ((p.ntiles + 7) / 8) * 8 * mtiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, stream, p);
    return v2w_launch_status();
}

template <int U>
int launch_convt_u(TileArgs p, hipStream_t stream) {
    if (p.Cout % 64 == 0) return launch_tile<32, U, 1, 1, 2, 2, 16>(p, stream);
    if (p.Cout == 32) return launch_tile<32, U, 1, 1, 1, 4, 16>(p, stream);
    if (p.Cout == 16) return launch_tile<16, U, 1, 2, 1, 4, 16>(p, stream);
    return V2W_E_SHAPE;
}

// wp[((mb*K + t)*G + g)*64 + lane][j] = wf[t][g*CKG + j*KSTEP + lane/MF][mb*MF + lane%MF]
__global__ void __launch_bounds__(256)
pack_mfma_kernel(const float* __restrict__ wf, float* __restrict__ wp, int K, int Cin, int Cout, int MF) {
    const int KSTEP = MF == 32 ? 2 : 4, CKG = 4 * KSTEP, G = Cin / CKG;
    const size_t total = (size_t)K * Cin * Cout;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int j = o & 3;
        const int lane = (o >> 2) & 63;
        size_t rest = o >> 8;
        const int g = rest % G; rest /= G;
        const int t = rest % K;
        const int mb = rest / K;
        const int c = g * CKG + j * KSTEP + lane / MF;
        const int co = mb * MF + lane % MF;
        wp[o] = wf[((size_t)t * Cin + c) * Cout + co];
    }
}

}  // namespace

// Which MFMA fragment the packed weights of a (C_in, C_out) layer use: 32, 16, or 0 when no tile configuration fits.
int v2w_mfma_frag(int c_in, int c_out) {
    if (c_out == 16 && c_in % 16 == 0) return 16;
    if (c_out % 32 == 0 && c_in % 16 == 0) return 32;
    return 0;
}

extern "C" int v2w_pack_mfma(const float* wf, float* wp, int k, int c_in, int c_out, void* stream) {
    if (!wf || !wp || k <= 0 || c_in <= 0 || c_out <= 0) return V2W_E_ARG;
    const int mf = v2w_mfma_frag(c_in, c_out);
    if (!mf) return V2W_E_SHAPE;
    const size_t total = (size_t)k * c_in * c_out;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wf, wp, k, c_in, c_out, mf);
    return v2w_launch_status();
}

// Called by v2w_api.hip.  Returns V2W_E_SHAPE when no tile configuration fits (caller falls back to the direct kernel).
int v2w_conv1d_mfma(const v2w_conv1d_args* a, hipStream_t stream) {
    if (!a->wp || !v2w_mfma_frag(a->C_in, a->C_out)) return V2W_E_SHAPE;
    TileArgs p{};
    p.in = a->in; p.in_a = a->in_a; p.in_s = a->in_s; p.wp = a->wp; p.bias = a->bias;
    p.res = a->res; p.res_a = a->res_a; p.res_s = a->res_s; p.out = a->out;
    p.B = a->B; p.Cin = a->C_in; p.Cout = a->C_out; p.L = a->L; p.K = a->k; p.dil = a->dil;
    p.pad = 0; p.hl = p.hr = a->dil * (a->k - 1) / 2;
    p.slope = a->slope; p.accumulate = a->accumulate; p.out_div = a->out_div;
    if (p.Cout % 128 == 0) return launch_tile<32, 1, 2, 2, 2, 2, 16>(p, stream);
    if (p.Cout == 64) return launch_tile<32, 1, 2, 2, 1, 4, 16>(p, stream);
    if (p.Cout % 32 == 0) return launch_tile<32, 1, 1, 2, 1, 4, 16>(p, stream);
    if (p.Cout == 16) return launch_tile<16, 1, 1, 8, 1, 4, 16>(p, stream);
    return V2W_E_SHAPE;
}

int v2w_convt1d_mfma(const v2w_convt1d_args* a, hipStream_t stream) {
    if (!a->wp || !v2w_mfma_frag(a->C_in, a->C_out)) return V2W_E_SHAPE;
    TileArgs p{};
    p.in = a->in; p.wp = a->wp; p.bias = a->bias; p.out = a->out;
    p.B = a->B; p.Cin = a->C_in; p.Cout = a->C_out; p.L = a->L; p.K = a->k; p.dil = 1;
    p.pad = (a->k - a->u) / 2;
    p.slope = a->slope; p.accumulate = 0; p.out_div = 0.f;
    // halo over all phases: offsets c_r - m, m in [0, nt_r)
    int hl = 0, hr = 0;
    for (int r = 0; r < a->u; ++r) {
        const int rp = r + p.pad, t0 = rp % a->u, c = rp / a->u, nt = (a->k - t0 + a->u - 1) / a->u;
        if (c > hr) hr = c;
        if (nt - 1 - c > hl) hl = nt - 1 - c;
    }
    p.hl = hl; p.hr = hr;
    switch (a->u) {
        case 2: return launch_convt_u<2>(p, stream);
        case 4: return launch_convt_u<4>(p, stream);
        case 5: return launch_convt_u<5>(p, stream);
        case 8: return launch_convt_u<8>(p, stream);
        default: return V2W_E_SHAPE;
    }
}
