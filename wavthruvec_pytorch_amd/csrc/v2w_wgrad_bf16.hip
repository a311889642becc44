// Weight gradient of a Conv1d on the gfx950 bf16 matrix pipe (the backward of a generator trained in the bf16 arithmetic:
// vec2wav/train.py:167,214 under torch.autocast - the reference's autocast backward computes the same products from bf16 operands).
//
//   dW[t][ci][co] = sum_{b,q}  bf16(act(x))[b, ci, q + off_t] * bf16(dy)[b, co, q],   off_t = (t - (k-1)/2) * dil,   fp32 accumulation
//   act(x) = leaky_relu(a*x + s): the activated signal the forward conv consumed (computed in fp32, rounded once).
//
// GEMM view as in v2w_wgrad.hip: the reduction runs over positions, A[ci][position] = act(x) tile (one A per tap: a column offset),
// B[position][co] = dy tile, D[ci][co] with co on the lanes.  What differs is the operand fetch: a lane of v_mfma_f32_32x32x16_bf16 /
// 16x16x32_bf16 feeds 8 CONSECUTIVE positions of one row (16 bytes).  The dy rows are read with one aligned ds_read_b128.  The signal rows
// are read at a tap offset, i.e. at any 2-byte alignment - a misaligned ds_read_b128 is correct on gfx950 but costs 64 clocks instead of 8
// (tools/exp/lds_unaligned_probe.hip) - so a lane reads the five dwords that cover its 8 elements (ds_read2_b32 x 2 + ds_read_b32, full
// rate: the signal rows have an odd dword stride) and shifts odd offsets into place with v_alignbit_b32.
//
// Two arrangements of the workgroup's 4 waves:
//   64 x 64 tile, 2 x 2 waves (C_in, C_out multiples of 64): every wave owns a 32 x 32 block for ALL taps of the launch (<= 7: 112
//       accumulator registers; k = 9 / 11 take two launches);
//   MF x MF tile, MF = 32 / 16 (the narrow stages): the waves share the TAPS (wave w: taps w*NT .. w*NT+NT-1) and every
//       wave reduces over all positions of the item - one launch per layer whatever k, a quarter of the partial slabs of a position split
//       (k <= 4: one tap per wave; above: three - two per wave measured slower).
// These layers are bound by the read of x and dy (C <= 64: one pass over both per launch), not by the matrix pipe; inputs are fp32
// tensors (converted while staging) or bf16 tensors (io_bf16 = 3).  Partials go to per-split slabs, summed in fixed order: deterministic.
#include <type_traits>
#include "v2w_common.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WgBfArgs {
    const void* x; const float* x_a; const float* x_s;   // (B, Cin, Lq) fp32 or bf16, and its per-(b, ci) affine (or null)
    const void* dy;                                       // (B, Cout, Lq) fp32 or bf16
    float* slab;                                          // [S][K][Cin][Cout]
    int B, Cin, Cout, Lq, K, dil;
    int tap0, ntap;           // 2 x 2 arrangement: taps [tap0, tap0 + ntap) in this launch
    int S, nchunk;            // position splits (grid.y); 128-position chunks per batch item
    int hla, xc8, xtw;        // halo columns staged on the left (multiple of 8), 8-element groups staged per signal row, signal row stride in dwords (odd)
    float slope;
};

__device__ __forceinline__ unsigned int wg_pack(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;          // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float wg_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float wg_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

template <int MF> struct BfFrag;
template <> struct BfFrag<32> {
    typedef f32x16 acc_t;
    static constexpr int NREG = 16, KSTEP = 16, KG = 2;
    __device__ static __forceinline__ acc_t mfma(u32x4 a, u32x4 b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row(int reg, int hk) { return (reg & 3) + 8 * (reg >> 2) + 4 * hk; }
};
template <> struct BfFrag<16> {
    typedef f32x4 acc_t;
    static constexpr int NREG = 4, KSTEP = 32, KG = 4;
    __device__ static __forceinline__ acc_t mfma(u32x4 a, u32x4 b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row(int reg, int hk) { return hk * 4 + reg; }
};

constexpr int WGB_PTQ = 128;              // positions per staged item
constexpr int WGB_DYW = WGB_PTQ + 8;      // dy row stride in elements: 272 bytes (16-byte aligned, 16 mod 256: b128 reads spread over the banks)

// MF, WCO, WCI: MFMA block and wave arrangement (2 x 2: every wave all NT taps; 1 x 1: the waves share the taps, NT per wave); BF: bf16 tensors.
template <int MF, int WCO, int WCI, int NT, bool BF>
__global__ void __launch_bounds__(256, 2)
wgrad_bf16_kernel(const WgBfArgs p) {
    typedef BfFrag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr bool TS = (WCO * WCI == 1);                       // taps shared out over the waves
    static_assert(TS || WCO * WCI == 4, "wave arrangement");
    constexpr int CO_T = WCO * MF, CI_T = WCI * MF, KSTEP = F::KSTEP;
    constexpr int NDG = CO_T * (WGB_PTQ / 8) / 256;             // 8-element dy groups per thread and item
    constexpr int TPR = 256 / CI_T;                             // threads sharing one signal row
    constexpr int NXG = (24 + TPR - 1) / TPR;                   // upper bound of signal groups per thread (xc8 <= 24)
    static_assert(NDG >= 1, "tile shape");
    extern __shared__ unsigned int smem[];
    unsigned short* const DYs = reinterpret_cast<unsigned short*>(smem);          // [CO_T][WGB_DYW] bf16
    unsigned int* const Xs = smem + CO_T * WGB_DYW / 2;                            // [CI_T][xtw] dwords (two bf16 each)

    const int cot = p.Cout / CO_T;
    const int co0 = (blockIdx.x % cot) * CO_T, ci0 = (blockIdx.x / cot) * CI_T;
    const int s = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1), hk = lane / MF;
    const int w_ci = TS ? 0 : (wave & 1), w_co = TS ? 0 : (wave >> 1);
    const int tbase = TS ? wave * NT : p.tap0;                  // first tap of this wave
    const int ntw = TS ? max(0, min(NT, p.K - tbase)) : NT;     // taps of this wave (wave-uniform)
    const int Lq = p.Lq;

    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) acc[t][e] = 0.f;

    const int items = p.B * p.nchunk;
    const int per = (items + p.S - 1) / p.S;
    const int it0 = s * per, it1 = min(items, it0 + per);

    const int xrow = tid / TPR, xl = tid % TPR;
    // staging registers: fp32 tensors park two float4 per group, bf16 tensors one 16-byte word
    typedef typename std::conditional<BF, u32x4, f32x4>::type ld_t;
    constexpr int LPG = BF ? 1 : 2;                             // loads per group
    ld_t dyv[NDG * LPG], xv[NXG * LPG];
    float xa = 1.f, xs = 0.f;
    const size_t esz = BF ? 2 : 4;
    auto issue = [&](int it) __attribute__((always_inline)) {
        const int b = it / p.nchunk, q0 = (it - b * p.nchunk) * WGB_PTQ;
#pragma unroll
        for (int i = 0; i < NDG; ++i) {
            const int g = tid + i * 256;
            const int row = g / (WGB_PTQ / 8), pos = q0 + (g % (WGB_PTQ / 8)) * 8;
            const char* src = reinterpret_cast<const char*>(p.dy) + (((size_t)b * p.Cout + co0 + row) * Lq + pos) * esz;
            const bool in = pos < Lq;
#pragma unroll
            for (int h = 0; h < LPG; ++h) {
                ld_t v = {};
                if (in) v = *reinterpret_cast<const ld_t*>(src + h * 16);
                dyv[i * LPG + h] = v;
            }
        }
        const int ch = b * p.Cin + ci0 + xrow;
        if (p.x_a) { xa = p.x_a[ch]; xs = p.x_s[ch]; }
        const char* xsrc = reinterpret_cast<const char*>(p.x) + (size_t)ch * Lq * esz;
#pragma unroll
        for (int i = 0; i < NXG; ++i) {
            const int c8 = xl + i * TPR;
            const int q = q0 - p.hla + c8 * 8;
            const bool in = c8 < p.xc8 && q >= 0 && q < Lq;
#pragma unroll
            for (int h = 0; h < LPG; ++h) {
                ld_t v = {};
                if (in) v = *reinterpret_cast<const ld_t*>(xsrc + (ptrdiff_t)q * (ptrdiff_t)esz + h * 16);
                xv[i * LPG + h] = v;
            }
        }
        return q0;
    };

    // operand rows of this lane
    const unsigned short* brow = DYs + (w_co * MF + lr) * WGB_DYW + hk * 8;
    const unsigned int* at[NT];
    int sh[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int e0 = p.hla + (tbase + t - (p.K - 1) / 2) * p.dil;          // element of the signal row under position 0 (>= 0 for a real tap)
        const int e = t < ntw ? e0 : p.hla;
        at[t] = Xs + (w_ci * MF + lr) * p.xtw + (e >> 1) + hk * 4;
        sh[t] = (e & 1) * 16;
    }

    int qnext = 0;
    if (it0 < it1) qnext = issue(it0);
    for (int it = it0; it < it1; ++it) {
        const int qcur = qnext;
        const float xa_cur = xa, xs_cur = xs;
        __syncthreads();                                            // the previous item's MFMA reads are done
#pragma unroll
        for (int i = 0; i < NDG; ++i) {
            const int g = tid + i * 256;
            const int row = g / (WGB_PTQ / 8), c8 = g % (WGB_PTQ / 8);
            u32x4 w;
            if constexpr (BF) w = dyv[i];
            else {
                const f32x4 v0 = dyv[2 * i], v1 = dyv[2 * i + 1];
                w = u32x4{wg_pack(v0[0], v0[1]), wg_pack(v0[2], v0[3]), wg_pack(v1[0], v1[1]), wg_pack(v1[2], v1[3])};
            }
            *reinterpret_cast<u32x4*>(DYs + row * WGB_DYW + c8 * 8) = w;
        }
        {
            // positions outside [0, Lq) stage as 0 (zero padding of the ACTIVATED signal), not act(s)
            unsigned int* drow = Xs + xrow * p.xtw;
#pragma unroll
            for (int i = 0; i < NXG; ++i) {
                const int c8 = xl + i * TPR;
                if (c8 >= p.xc8) continue;
                const int q = qcur - p.hla + c8 * 8;
                const bool in = q >= 0 && q < Lq;
                float v[8];
                if constexpr (BF) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] = wg_lo(xv[i][e]); v[2 * e + 1] = wg_hi(xv[i][e]); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = xv[2 * i][e]; v[4 + e] = xv[2 * i + 1][e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = in ? v2w_lrelu(fmaf(xa_cur, v[e], xs_cur), p.slope) : 0.f;
                unsigned int* d = drow + c8 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = wg_pack(v[2 * e], v[2 * e + 1]);
            }
        }
        __syncthreads();
        if (it + 1 < it1) qnext = issue(it + 1);                    // in flight during the MFMA loop below
#pragma unroll 2
        for (int kq = 0; kq < WGB_PTQ; kq += KSTEP) {
            const u32x4 bv = *reinterpret_cast<const u32x4*>(brow + kq);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (TS && t >= ntw) continue;
                const unsigned int* r = at[t] + kq / 2;
                const unsigned int w0 = r[0], w1 = r[1], w2 = r[2], w3 = r[3], w4 = r[4];
                const u32x4 av = {__builtin_amdgcn_alignbit(w1, w0, sh[t]), __builtin_amdgcn_alignbit(w2, w1, sh[t]),
                                  __builtin_amdgcn_alignbit(w3, w2, sh[t]), __builtin_amdgcn_alignbit(w4, w3, sh[t])};
                acc[t] = F::mfma(av, bv, acc[t]);
            }
        }
    }

    float* dst = p.slab + (size_t)s * p.K * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (TS && t >= ntw) continue;
#pragma unroll
        for (int e = 0; e < F::NREG; ++e) {
            const int ci = ci0 + w_ci * MF + F::row(e, hk);
            const int co = co0 + w_co * MF + lr;
            dst[((size_t)(tbase + t) * p.Cin + ci) * p.Cout + co] = acc[t][e];
        }
    }
}

// dwf[i] = sum_s slab[s][i]: a block covers 64 float4 outputs x 16 slab lanes (one per wave): every wave sums each 16th slab with 4 loads
// in flight, the sixteen partial sums are combined in fixed order (deterministic).  (The narrow layers have few outputs and many slabs:
// the 4-lane reduce of v2w_wgrad.hip walks 512 slabs in 3 workgroups.)
__global__ void __launch_bounds__(1024)
wgrad_bf16_reduce_kernel(const float* slab, float* dwf, size_t n, int nslab) {
    __shared__ f32x4 part[16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + o) * 4;             // n % 4 == 0: weights are k * C_in * C_out with C % 16 == 0
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

// Arrangement and number of position splits for a layer; 0 = no instantiation (the caller runs the exact fp32 kernel).
static int wgb_plan(int B, int c_in, int c_out, int Lq, int k, int* mf_o, int* ts_o) {
    if (c_in != c_out || Lq % 8 != 0 || !(k & 1) || k < 1 || k > 11) return 0;
    int mf, ts;
    if (c_in % 64 == 0) { mf = 32; ts = 0; }
    else if (c_in == 32) { mf = 32; ts = 1; }
    else if (c_in == 16) { mf = 16; ts = 1; }
    else return 0;
    const int tile = ts ? mf : 64;
    const int tiles = (c_in / tile) * (c_out / tile);
    const int items = B * ((Lq + WGB_PTQ - 1) / WGB_PTQ);
    int S = ((ts ? 4 : 2) * 256 + tiles - 1) / tiles;             // workgroups per CU the register / LDS budget lets stay resident
    if (S > items) S = items;
    if (S < 1) S = 1;
    if (mf_o) { *mf_o = mf; *ts_o = ts; }
    return S;
}

template <int MF, int WCO, int WCI, int NT, bool BF>
static void wgb_launch(const WgBfArgs& p, int tiles, size_t lds, hipStream_t st) {
    auto kern = wgrad_bf16_kernel<MF, WCO, WCI, NT, BF>;
    if (lds > 64 * 1024) (void)v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, st);
    V2W_LAUNCH(kern, dim3(tiles, p.S), dim3(256), lds, st, p);
}

template <int MF, int WCO, int WCI, bool BF>
static bool wgb_launch_nt(int nt, const WgBfArgs& p, int tiles, size_t lds, hipStream_t st) {
    switch (nt) {
        case 1: wgb_launch<MF, WCO, WCI, 1, BF>(p, tiles, lds, st); return true;
        case 2: wgb_launch<MF, WCO, WCI, 2, BF>(p, tiles, lds, st); return true;
        case 3: wgb_launch<MF, WCO, WCI, 3, BF>(p, tiles, lds, st); return true;
        default: break;
    }
    if constexpr (WCO * WCI == 4) {
        switch (nt) {
            case 4: wgb_launch<MF, WCO, WCI, 4, BF>(p, tiles, lds, st); return true;
            case 5: wgb_launch<MF, WCO, WCI, 5, BF>(p, tiles, lds, st); return true;
            case 6: wgb_launch<MF, WCO, WCI, 6, BF>(p, tiles, lds, st); return true;
            case 7: wgb_launch<MF, WCO, WCI, 7, BF>(p, tiles, lds, st); return true;
            default: break;
        }
    }
    return false;
}

}  // namespace

// Position splits (= partial slabs of k*C_in*C_out floats) v2w_wgrad_bf16 needs for this layer; 0 = no bf16 instantiation for the shape.
extern "C" int v2w_wgrad_bf16_slabs(int B, int c_in, int c_out, int Lq, int k) {
    return wgb_plan(B, c_in, c_out, Lq, k, nullptr, nullptr);
}

extern "C" int v2w_wgrad_bf16(const void* x, const float* x_a, const float* x_s, const void* dy, float* dwf, float* slab_ws,
                              int B, int c_in, int c_out, int Lq, int k, int dil, float slope, int io_bf16, void* stream) {
    if (!x || !dy || !dwf || !slab_ws || B <= 0 || c_in <= 0 || c_out <= 0 || Lq <= 0 || k <= 0 || dil <= 0) return V2W_E_ARG;
    if ((x_a == nullptr) != (x_s == nullptr) || (io_bf16 != 0 && io_bf16 != 3)) return V2W_E_ARG;
    int mf = 0, ts = 0;
    const int S = wgb_plan(B, c_in, c_out, Lq, k, &mf, &ts);
    if (!S) return V2W_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return V2W_E_ARG;
    const int hl = (k - 1) / 2 * dil;
    WgBfArgs p{};
    p.x = x; p.x_a = x_a; p.x_s = x_s; p.dy = dy; p.slab = slab_ws;
    p.B = B; p.Cin = c_in; p.Cout = c_out; p.Lq = Lq; p.K = k; p.dil = dil; p.slope = slope;
    p.S = S; p.nchunk = (Lq + WGB_PTQ - 1) / WGB_PTQ;
    p.hla = (hl + 7) & ~7;
    p.xc8 = (p.hla + WGB_PTQ + hl + 7) / 8;
    if (p.xc8 > 24) return V2W_E_SHAPE;                              // halo beyond the staged tile (k = 11 reaches dilation 5)
    p.xtw = (p.xc8 * 4 + 1) | 1;                                    // one dword past the last element is read by the odd-offset shift
    const int tile = ts ? mf : 64;
    const int tiles = (c_in / tile) * (c_out / tile);
    const size_t lds = (size_t)tile * WGB_DYW * 2 + (size_t)tile * p.xtw * 4;
    hipStream_t st = (hipStream_t)stream;
    const bool bf = io_bf16 == 3;
    bool ok = true;
    if (ts) {
        const int nt = k <= 4 ? 1 : 3;                                // (two taps per wave - k = 5 .. 8 - measured slower than three: 110 against 75 us at C = 32, k = 7)
        p.tap0 = 0; p.ntap = k;
        if (mf == 32) ok = bf ? wgb_launch_nt<32, 1, 1, true>(nt, p, tiles, lds, st) : wgb_launch_nt<32, 1, 1, false>(nt, p, tiles, lds, st);
        else ok = bf ? wgb_launch_nt<16, 1, 1, true>(nt, p, tiles, lds, st) : wgb_launch_nt<16, 1, 1, false>(nt, p, tiles, lds, st);
    } else {
        const int ngrp = (k + 6) / 7, gsz = (k + ngrp - 1) / ngrp;     // balanced tap groups: 11 -> 6 + 5, 9 -> 5 + 4
        for (int t0 = 0; t0 < k && ok; t0 += gsz) {
            p.tap0 = t0; p.ntap = k - t0 < gsz ? k - t0 : gsz;
            ok = bf ? wgb_launch_nt<32, 2, 2, true>(p.ntap, p, tiles, lds, st) : wgb_launch_nt<32, 2, 2, false>(p.ntap, p, tiles, lds, st);
        }
    }
    if (!ok) return V2W_E_SHAPE;
    const size_t nw = (size_t)k * c_in * c_out;
    V2W_LAUNCH(wgrad_bf16_reduce_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(1024), 0, st, slab_ws, dwf, nw, S);
    return v2w_launch_status();
}
