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
// GEMM view per workgroup: M = MT output channels, N = NT input-rate positions, K = (C_in chunk) x taps.
// A (weights)  : LDS tile Ws[tap][c][MT]      - lanes read consecutive co          -> conflict-free ds_read_b32
// B (signal)   : LDS tile Xs[c][NT + halo]    - lanes read consecutive positions, every tap is just a column
//                offset into the SAME tile, so each input element is fetched from HBM once per M-tile and the
//                activation / CondBN affine is applied once at staging time, not once per tap.
// MFMA         : v_mfma_f32_32x32x2_f32 (C_out >= 32) or v_mfma_f32_16x16x4_f32 (C_out == 16): exact fp32
//                (bit-identical to an fmaf chain), 64 FLOP/clk/SIMD.
// Waves        : WM x WN waves per workgroup, each owning (MF*MI) x (MF*NI) outputs for each of the U phases;
//                64-lane fragments: lane&(MF-1) = row/col inside the MFMA tile, lane/MF = k index.
#include "v2w_common.h"

namespace {

struct TileArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* wf; const float* bias;
    const float* res; const float* res_a; const float* res_s;
    float* out;
    int B, Cin, Cout, L, K, dil;
    int pad;      // convT only: (K-U)/2
    int hl, hr;   // halo (input positions) left / right of the tile
    int xw;       // LDS row stride of the input tile (floats)
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

template <int MF, int U, int MI, int NI, int WM, int WN, int CK>
__global__ void __launch_bounds__(64 * WM * WN)
conv_tile_kernel(const TileArgs p) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int MT = MF * MI * WM;
    constexpr int NT = MF * NI * WN;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int KSTEP = F::KSTEP;
    static_assert(CK % KSTEP == 0, "chunk must hold whole MFMA k-steps");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                 // [CK][xw]
    float* Ws = smem + CK * p.xw;     // [K][CK][MT]

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
    const int xw_used = NT + p.hl + p.hr;
    const int L = p.L, K = p.K;
    const float slope = p.slope;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += CK) {
        if (ci0) __syncthreads();   // everyone is done reading the previous chunk
        // ---- stage the input tile: activation (and the folded CondBN affine) applied once, zero padding outside [0,L)
        for (int c = wave; c < CK; c += WM * WN) {
            const int ch = b * p.Cin + ci0 + c;
            const float* src = p.in + (size_t)ch * L;
            const float av = p.in_a ? p.in_a[ch] : 1.f;
            const float sv = p.in_s ? p.in_s[ch] : 0.f;
            for (int j = lane; j < xw_used; j += 64) {
                const int l = n0 - p.hl + j;
                float v = 0.f;
                if (l >= 0 && l < L) v = v2w_lrelu(fmaf(av, src[l], sv), slope);
                Xs[c * xw + j] = v;
            }
        }
        // ---- stage the weight chunk Ws[t][c][0..MT) <- wf[t][ci0+c][m0..m0+MT)
        for (int idx = tid; idx < K * CK * MT; idx += NTHREADS) {
            const int x = idx % MT, row = idx / MT;
            const int c = row % CK, t = row / CK;
            Ws[idx] = p.wf[((size_t)(t * p.Cin + ci0 + c)) * p.Cout + m0 + x];
        }
        __syncthreads();

#pragma unroll
        for (int r = 0; r < U; ++r) {
            int t0, tstr, d0, dstr, nt;
            if (U == 1) { t0 = 0; tstr = 1; d0 = -p.hl; dstr = p.dil; nt = K; }
            else { const int rp = r + p.pad; t0 = rp % U; tstr = U; d0 = rp / U; dstr = -1; nt = (K - t0 + U - 1) / U; }
            for (int m = 0; m < nt; ++m) {
                const float* wrow = Ws + (t0 + m * tstr) * (CK * MT) + wm0 + lr;
                const float* xrow = Xs + wn0 + lr + p.hl + d0 + m * dstr;
#pragma unroll
                for (int kk = 0; kk < CK / KSTEP; ++kk) {
                    const int c = kk * KSTEP + hk;
                    float a[MI], bb[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[i] = wrow[c * MT + i * MF];
#pragma unroll
                    for (int j = 0; j < NI; ++j) bb[j] = xrow[c * xw + j * MF];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[r][i][j] = F::mfma(a[i], bb[j], acc[r][i][j]);
                }
            }
        }
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
    constexpr int MT = MF * MI * WM, NT = MF * NI * WN;
    if (p.Cout % MT != 0 || p.Cin % CK != 0) return V2W_E_SHAPE;
    p.ntl = (p.L + NT - 1) / NT;
    p.ntiles = p.B * p.ntl;
    int xw = NT + p.hl + p.hr;
    if (MF == 16) xw += ((16 - xw % 32) + 32) % 32;  // xw % 32 == 16: the two 16-lane k-groups of a half-wave hit disjoint banks
    else xw = (xw + 3) & ~3;
    p.xw = xw;
    const size_t lds = ((size_t)CK * xw + (size_t)p.K * CK * MT) * sizeof(float);
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    auto kern = conv_tile_kernel<MF, U, MI, NI, WM, WN, CK>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int mtiles = p.Cout / MT;
    const int grid = ((p.ntiles + 7) / 8) * 8 * mtiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, p);
    return v2w_launch_status();
}

template <int U>
int launch_convt_u(TileArgs p, hipStream_t stream) {
    if (p.Cout % 64 == 0) return launch_tile<32, U, 1, 1, 2, 2, 8>(p, stream);
    if (p.Cout == 32) return launch_tile<32, U, 1, 1, 1, 4, 8>(p, stream);
    if (p.Cout == 16) return launch_tile<16, U, 1, 2, 1, 4, 8>(p, stream);
    return V2W_E_SHAPE;
}

}  // namespace

// Called by v2w_api.hip.  Returns V2W_E_SHAPE when no tile configuration fits (caller falls back to the direct kernel).
int v2w_conv1d_mfma(const v2w_conv1d_args* a, hipStream_t stream) {
    TileArgs p{};
    p.in = a->in; p.in_a = a->in_a; p.in_s = a->in_s; p.wf = a->wf; p.bias = a->bias;
    p.res = a->res; p.res_a = a->res_a; p.res_s = a->res_s; p.out = a->out;
    p.B = a->B; p.Cin = a->C_in; p.Cout = a->C_out; p.L = a->L; p.K = a->k; p.dil = a->dil;
    p.pad = 0; p.hl = p.hr = a->dil * (a->k - 1) / 2;
    p.slope = a->slope; p.accumulate = a->accumulate; p.out_div = a->out_div;
    if (p.Cin % 8 != 0) return V2W_E_SHAPE;
    if (p.Cout % 128 == 0) return launch_tile<32, 1, 2, 2, 2, 2, 8>(p, stream);
    if (p.Cout == 64) return launch_tile<32, 1, 2, 2, 1, 4, 8>(p, stream);
    if (p.Cout == 32) return launch_tile<32, 1, 1, 2, 1, 4, 8>(p, stream);
    if (p.Cout == 16) return launch_tile<16, 1, 1, 8, 1, 4, 8>(p, stream);
    return V2W_E_SHAPE;
}

int v2w_convt1d_mfma(const v2w_convt1d_args* a, hipStream_t stream) {
    TileArgs p{};
    p.in = a->in; p.wf = a->wf; p.bias = a->bias; p.out = a->out;
    p.B = a->B; p.Cin = a->C_in; p.Cout = a->C_out; p.L = a->L; p.K = a->k; p.dil = 1;
    p.pad = (a->k - a->u) / 2;
    p.slope = a->slope; p.accumulate = 0; p.out_div = 0.f;
    if (p.Cin % 8 != 0) return V2W_E_SHAPE;
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
