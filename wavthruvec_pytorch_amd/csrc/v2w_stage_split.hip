// Whole residual section of a narrow ResBlock2 stage (C = 32 or C = 16) in ONE kernel on the f16 / bf16 matrix pipe
// (models.py:135-141 with ResBlock2.forward, models.py:65-70, inlined):
//   out = ( sum_j [ t1_j + conv_{k_j,d2_j}(lrelu(t1_j)) + b2_j ] ) / nk ,   t1_j = x + conv_{k_j,d1_j}(lrelu(x)) + b1_j ,  x = a*in + s
//
// The split-operand counterpart of resblock2_stage_kernel (v2w_resblock_fused.hip): same tiling (x staged once for all nk
// branches, every t1_j only in LDS, the branch sum in registers in the reference's order ((r0 + r1) + r2)), with the
// arithmetic of v2w_conv_split.hip: operands x = x_hi + x_lo in f16 (weights pre-scaled by a per-layer power of two),
// x_hi*w_hi + x_hi*w_lo + x_lo*w_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (BF: bf16 operands, one MFMA).
//
//   LDS  X [position][32 ch hi | 32 ch lo | pad] (144-B rows: conflict-free ds_read_b128 of 8 channels per lane): the
//          ACTIVATED operand lrelu(x), exactly 0 outside [0, L); a tap is a row offset.
//        T  same layout: lrelu(t1_j) on the W positions conv2 needs.
//   Residuals: x and t1 exist in LDS only as the halves of their ACTIVATED values; the residual adds rebuild them,
//          v = unlrelu(hi + lo), accurate to 2^-22 |v| (the precision of the products themselves); in the bf16 form x is
//          re-read from global memory instead (its tile holds 8 bits), t1 is carried at bf16 precision like every operand.  (Re-reading x from
//          global memory in the conv1 epilogue is exact but cost ~5000 cycles per branch: 64 latency-exposed loads per thread.)
//   Weights: the (hi, lo) fragment stream of v2w_pack_split (row block 0: [chunk C/16][tap K][hi | lo][64 lanes][16 B]; for C = 16
//          the 16 output rows are zero-padded to the 32 rows of the MFMA: these stages are latency-, not MFMA-bound),
//          read straight from L2 into registers through a 4-slot ring: no weight barrier at all, two barriers per branch.
#include "v2w_common.h"
#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef unsigned int raw16 __attribute__((ext_vector_type(4)));

#define V2W_SS_MAXB 4

struct StageSplitArgs {
    const float* in; const float* in_a; const float* in_s;
    const unsigned char* w1[V2W_SS_MAXB]; const float* sc1[V2W_SS_MAXB]; const float* bias1[V2W_SS_MAXB];
    const unsigned char* w2[V2W_SS_MAXB]; const float* sc2[V2W_SS_MAXB]; const float* bias2[V2W_SS_MAXB];
    int K[V2W_SS_MAXB], d1[V2W_SS_MAXB], d2[V2W_SS_MAXB];
    const unsigned char* wbase;   // the 2*nk weight streams lie back to back in execution order (w1_0, w2_0, w1_1, ...): unit g at wbase + g*2048
    int ntot;                     // units of all streams
    float* out;
    int nk, B, L;
    int h1max, h2max;
    int xoff, xrows, nto, ntl;
    int vec4;
    float slope, out_div;
};

// C = 16: 43 KB of LDS -> three workgroups per CU (register budget 168); C = 32: 79 KB -> two
template <int NCH, int NI, int WN, bool BF>
__global__ void __launch_bounds__(64 * WN) __attribute__((amdgpu_waves_per_eu(NCH == 1 ? 3 : 2, NCH == 1 ? 3 : 2)))
stage_split_kernel(const StageSplitArgs p) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTHREADS = 64 * WN;
    constexpr int C = 16 * NCH;                     // channels: NCH chunks of one MFMA k-step
    constexpr int HB = C * 2;                       // bytes of the hi (or lo) half of a row
    constexpr int W = 32 * NI * WN;                 // positions computed per phase
    constexpr int ROWB = 2 * HB + 16;               // 144 (C = 32) / 80 (C = 16): conflict-free ds_read_b128 across rows

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * p.nto;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, hk = lane >> 5;
    const int wn0 = wave * (32 * NI);
    const int L = p.L;
    const float slope = p.slope;
    unsigned char* const Xs = smem;                              // [xrows][ROWB]
    unsigned char* const Ts = smem + p.xrows * ROWB;             // [W][ROWB]
    float* const etab = reinterpret_cast<float*>(Ts + W * ROWB); // bias1[nk][C], bias2[nk][C], then a[C], s[C] of this batch item

    auto act = [&](float v) __attribute__((always_inline)) {
        v = slope <= 1.f ? fmaxf(v, v * slope) : v2w_lrelu(v, slope);
        return BF ? v : __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    };
    // 4 consecutive channels of one position -> their (hi, lo) halves at `d` (hi) and `d + HB` (lo)
    auto put4 = [&](unsigned char* d, const float (&v)[4]) __attribute__((always_inline)) {
        if constexpr (BF) {
            b4 hi;
#pragma unroll
            for (int c = 0; c < 4; ++c) hi[c] = (__bf16)v[c];
            *reinterpret_cast<b4*>(d) = hi;
        } else {
            h4 hi, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const _Float16 h = (_Float16)v[c];
                hi[c] = h;
                lo[c] = (_Float16)(v[c] - (float)h);
            }
            *reinterpret_cast<h4*>(d) = hi;
            *reinterpret_cast<h4*>(d + HB) = lo;
        }
    };

    // ---- prologue: biases, the activated x tile (row r <-> position pos0 + r)
    const int pos0 = n0 - p.h2max - p.h1max - p.xoff;
    for (int i = tid; i < p.nk * C; i += NTHREADS) {
        const int j = i / C, c = i - j * C;
        etab[i] = p.bias1[j] ? p.bias1[j][c] : 0.f;
        etab[V2W_SS_MAXB * C + i] = p.bias2[j] ? p.bias2[j][c] : 0.f;
    }
    float* const aff = etab + 2 * V2W_SS_MAXB * C;
    for (int c = tid; c < C; c += NTHREADS) {
        aff[c] = p.in_a ? p.in_a[b * C + c] : 1.f;
        aff[C + c] = p.in_a ? p.in_s[b * C + c] : 0.f;
    }
    if (p.vec4) {
        // wave w stages channels (C/4)w .. (C/4)w + C/4 - 1 (NCH groups of 4), lane l the position groups l, l + 64, ...
        const int xp4 = p.xrows >> 2;
        for (int g = 0; g < NCH; ++g) {
            const int c0 = (wave * NCH + g) * 4;
            float av[4], sv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                av[c] = p.in_a ? p.in_a[b * C + c0 + c] : 1.f;
                sv[c] = p.in_a ? p.in_s[b * C + c0 + c] : 0.f;
            }
            // every global load of the group is issued before the first conversion: one exposed latency, not one per slice
            constexpr int NPG = (W + 64 + 4) / 4 / 64 + 1;      // position groups per lane (xrows <= W + 2*32 + 4)
            f32x4 x4[NPG][4];
#pragma unroll
            for (int s = 0; s < NPG; ++s) {
                const int pg = lane + s * 64;
                const int pos = pos0 + pg * 4;
                const bool in_seq = pg < xp4 && pos >= 0 && pos < L;      // L % 4 == 0, pos % 4 == 0: whole float4 in or out
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    x4[s][c] = in_seq ? *reinterpret_cast<const f32x4*>(p.in + (size_t)(b * C + c0 + c) * L + pos) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < NPG; ++s) {
                const int pg = lane + s * 64;
                if (pg >= xp4) continue;
                const int pos = pos0 + pg * 4;
                const bool in_seq = pos >= 0 && pos < L;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = in_seq ? act(fmaf(av[c], x4[s][c][e], sv[c])) : 0.f;
                    put4(Xs + (pg * 4 + e) * ROWB + c0 * 2, v);
                }
            }
        }
    } else {
        for (int c = wave; c < C; c += WN) {
            const int ch = b * C + c;
            const float av = p.in_a ? p.in_a[ch] : 1.f, sv = p.in_a ? p.in_s[ch] : 0.f;
            for (int r = lane; r < p.xrows; r += 64) {
                const int pos = pos0 + r;
                const float v = (pos >= 0 && pos < L) ? act(fmaf(av, p.in[(size_t)ch * L + pos], sv)) : 0.f;
                if constexpr (BF) {
                    reinterpret_cast<__bf16*>(Xs + r * ROWB)[c] = (__bf16)v;
                } else {
                    const _Float16 h = (_Float16)v;
                    _Float16* d = reinterpret_cast<_Float16*>(Xs + r * ROWB) + c;
                    d[0] = h;
                    d[C] = (_Float16)(v - (float)h);
                }
            }
        }
    }
    __syncthreads();

    auto mma = [&](acc_t c, raw16 a, raw16 bb) __attribute__((always_inline)) {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, bb), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, bb), c, 0, 0, 0);
    };

    acc_t acc[NI], oacc[NI];
    // Weight units (chunk, tap) travel L2 -> registers through a 4-slot ring, THREE units ahead of their use: a unit is only
    // 3 * NI MFMAs (~200 cycles) of work, one unit of lookahead exposes most of the L2 latency.  The ring runs on across the
    // conv phases (the tail of a phase prefetches the head of the next stream); a stream has 2K = 6 / 14 / 22 units, so a
    // phase starts at any slot: the phase body is instantiated per start slot (S0) to keep every slot index static.
    raw16 rh[4], rl[BF ? 1 : 4];
    auto ld = [&](auto slot_c, const unsigned char* ptr) __attribute__((always_inline)) {
        constexpr int sl = decltype(slot_c)::value;
        rh[sl] = *reinterpret_cast<const raw16*>(ptr);
        if constexpr (!BF) rl[sl] = *reinterpret_cast<const raw16*>(ptr + 1024);
    };
    auto conv_phase = [&](auto s0_c, int ug0, const unsigned char* src, int rowbase, int maxrow, int K, int dil) __attribute__((always_inline)) {
        constexpr int S0 = decltype(s0_c)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const int nu = NCH * K;
        // ring prefetch pointer: the streams are contiguous, so the unit three ahead is simply + 3 units (clamped at the very end)
        const unsigned char* wp_ = p.wbase + (size_t)(ug0 + 3) * 2048 + lane * 16;
        const unsigned char* const wend = p.wbase + (size_t)(p.ntot - 1) * 2048 + lane * 16;
        // signal fragments of unit u+1 are read while the MFMAs of unit u run (bn -> bc hand-over in registers)
        raw16 bh[NI], bl[BF ? 1 : NI];
        auto read_b = [&](int u) __attribute__((always_inline)) {
            const int chk = u >= K ? 1 : 0, t = u - chk * K;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                // columns past the valid outputs of conv2 would read beyond T: clamped (their results are discarded)
                const unsigned char* xr = src + min(rowbase + j * 32 + t * dil, maxrow) * ROWB + chk * 32 + hk * 16;
                bh[j] = *reinterpret_cast<const raw16*>(xr);
                if constexpr (!BF) bl[j] = *reinterpret_cast<const raw16*>(xr + HB);
            }
        };
        read_b(0);
        auto unit = [&](auto s_c, int u) __attribute__((always_inline)) {
            constexpr int sl = (S0 + decltype(s_c)::value) & 3, slp = (sl + 3) & 3;
            ld(std::integral_constant<int, slp>{}, wp_ < wend ? wp_ : wend);
            wp_ += 2048;
            __builtin_amdgcn_sched_barrier(0);
            raw16 ch_[NI], cl_[BF ? 1 : NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) { ch_[j] = bh[j]; if constexpr (!BF) cl_[j] = bl[j]; }
            if (u + 1 < nu) read_b(u + 1);
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[j] = mma(acc[j], rh[sl], ch_[j]);
            if constexpr (!BF) {
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[j] = mma(acc[j], rh[sl], cl_[j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[j] = mma(acc[j], rl[sl], ch_[j]);
            }
        };
        for (int u0 = 0; u0 < nu; u0 += 4) {
            unit(std::integral_constant<int, 0>{}, u0);
            if (u0 + 1 < nu) unit(std::integral_constant<int, 1>{}, u0 + 1);
            if (u0 + 2 < nu) unit(std::integral_constant<int, 2>{}, u0 + 2);
            if (u0 + 3 < nu) unit(std::integral_constant<int, 3>{}, u0 + 3);
        }
    };
    // C = 32: a stream has 2K units and K is odd, 2K = 2 (mod 4): conv1 of every branch starts at ring slot 0 and conv2 at
    // slot 2 (static).  C = 16: K units per stream, the start slot walks: the phase body is selected by a switch.
    int s0 = 0;
    int ug = 0;                                                  // global index of the next phase's first unit
    auto run_phase = [&](auto conv2_c, const unsigned char* src, int rowbase, int maxrow, int K, int dil) __attribute__((always_inline)) {
        if constexpr (NCH == 2) {
            conv_phase(std::integral_constant<int, decltype(conv2_c)::value ? 2 : 0>{}, ug, src, rowbase, maxrow, K, dil);
        } else {
            switch (s0) {
                case 0: conv_phase(std::integral_constant<int, 0>{}, ug, src, rowbase, maxrow, K, dil); break;
                case 1: conv_phase(std::integral_constant<int, 1>{}, ug, src, rowbase, maxrow, K, dil); break;
                case 2: conv_phase(std::integral_constant<int, 2>{}, ug, src, rowbase, maxrow, K, dil); break;
                default: conv_phase(std::integral_constant<int, 3>{}, ug, src, rowbase, maxrow, K, dil); break;
            }
            s0 = (s0 + K) & 3;
        }
        ug += NCH * K;
    };
    ld(std::integral_constant<int, 0>{}, p.wbase + lane * 16);
    ld(std::integral_constant<int, 1>{}, p.wbase + lane * 16 + 2048);
    ld(std::integral_constant<int, 2>{}, p.wbase + lane * 16 + 4096);

    for (int jb = 0; jb < p.nk; ++jb) {
        const int K = p.K[jb], d1 = p.d1[jb], d2 = p.d2[jb];
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        const float winv1 = p.sc1[jb][0], winv2 = p.sc2[jb][0];

        // ---- conv1_j -> t1_j on positions [n0 - h2max, n0 - h2max + W): X row of output column c, tap 0 = c + xoff + h1max - h1
        run_phase(std::false_type{}, Xs, wn0 + lr + p.xoff + (p.h1max - h1), p.xrows - 1, K, d1);
        if (jb > 0) __syncthreads();              // conv2 of the previous branch has finished reading T
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn0 + j * 32 + lr;
            const int pos = n0 - p.h2max + col;
            const bool in_seq = pos >= 0 && pos < L;
            const unsigned char* xrow_ = Xs + (col + p.xoff + p.h1max) * ROWB;     // the x tile row of this output position
#pragma unroll
            for (int q = 0; q < 2 * NCH; ++q) {                   // accumulator rows 8q + 4hk + c < C (C = 16: the padded rows are skipped)
                float xv[4], v[4];
                if constexpr (BF) {
                    // bf16 operands carry 8 bits: the residual path must not go through them.  x is re-read from global memory
                    // (L2) and affine-folded on the fly; xv holds the ACTIVATED value like the split branch below
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int co = 8 * q + 4 * hk + c;
                        xv[c] = in_seq ? v2w_lrelu(fmaf(aff[co], p.in[(size_t)(b * C + co) * L + pos], aff[C + co]), slope) : 0.f;
                    }
                } else {
                    const h4 hi = *reinterpret_cast<const h4*>(xrow_ + (8 * q + 4 * hk) * 2);
                    const h4 lo = *reinterpret_cast<const h4*>(xrow_ + (8 * q + 4 * hk) * 2 + HB);
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = (float)hi[c] + (float)lo[c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int co = 8 * q + 4 * hk + c;            // accumulator register 4*q + c
                    const float x = xv[c] >= 0.f ? xv[c] : xv[c] / slope;      // undo the leaky_relu applied at staging
                    const float t1 = acc[j][4 * q + c] * winv1 + etab[jb * C + co] + x;
                    v[c] = in_seq ? act(t1) : 0.f;                // conv2 zero-pads t1 (and lrelu(0) = 0)
                }
                put4(Ts + col * ROWB + (8 * q + 4 * hk) * 2, v);
            }
        }
        __syncthreads();

        // ---- conv2_j ; r_j = (acc + b2) + t1_j ; branch sum in the reference's order
        run_phase(std::true_type{}, Ts, wn0 + lr + (p.h2max - h2), W - 1, K, d2);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const unsigned char* trow = Ts + (wn0 + j * 32 + lr + p.h2max) * ROWB;
#pragma unroll
            for (int q = 0; q < 2 * NCH; ++q) {
                float y[4];
                if constexpr (BF) {
                    const b4 hi = *reinterpret_cast<const b4*>(trow + (8 * q + 4 * hk) * 2);
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[c] = (float)hi[c];
                } else {
                    const h4 hi = *reinterpret_cast<const h4*>(trow + (8 * q + 4 * hk) * 2);
                    const h4 lo = *reinterpret_cast<const h4*>(trow + (8 * q + 4 * hk) * 2 + HB);
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[c] = (float)hi[c] + (float)lo[c];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int co = 8 * q + 4 * hk + c;
                    const float t1 = y[c] >= 0.f ? y[c] : y[c] / slope;     // undo the leaky_relu applied at staging
                    const float r = (acc[j][4 * q + c] * winv2 + etab[V2W_SS_MAXB * C + jb * C + co]) + t1;
                    oacc[j][4 * q + c] = jb == 0 ? r : oacc[j][4 * q + c] + r;
                }
            }
        }
    }

#pragma unroll
    for (int e = 0; e < 8 * NCH; ++e) {                          // rows >= C are MFMA padding
        const int co = F::row(e, hk);
        const size_t orow = ((size_t)b * C + co) * L;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn0 + j * 32 + lr, pos = n0 + col;
            if (col >= p.nto || pos >= L) continue;
            float v = oacc[j][e];
            if (p.out_div != 0.f) v = v / p.out_div;
            p.out[orow + pos] = v;
        }
    }
}

template <int NCH, int NI, int WN>
int launch_stage_split(const v2w_stage_split_args* q, hipStream_t stream) {
    constexpr int W = 32 * NI * WN, C = 16 * NCH, ROWB = 4 * C + 16;
    StageSplitArgs p{};
    p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.out = q->out;
    p.nk = q->nk; p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div;
    for (int j = 0; j < q->nk; ++j) {
        p.w1[j] = reinterpret_cast<const unsigned char*>(q->wps1[j]); p.sc1[j] = q->sc1[j]; p.bias1[j] = q->bias1[j];
        p.w2[j] = reinterpret_cast<const unsigned char*>(q->wps2[j]); p.sc2[j] = q->sc2[j]; p.bias2[j] = q->bias2[j];
        p.K[j] = q->k[j]; p.d1[j] = q->dil1[j]; p.d2[j] = q->dil2[j];
        const int h1 = q->dil1[j] * (q->k[j] - 1) / 2, h2 = q->dil2[j] * (q->k[j] - 1) / 2;
        if (h1 > p.h1max) p.h1max = h1;
        if (h2 > p.h2max) p.h2max = h2;
    }
    // the kernel walks ONE weight stream: w1_0, w2_0, w1_1, ... must lie back to back (NCH*k units of 2 KiB each)
    p.wbase = p.w1[0];
    const unsigned char* expect = p.wbase;
    for (int j = 0; j < q->nk; ++j) {
        if (p.w1[j] != expect && !v2w_dry(stream)) return V2W_E_ARG;
        expect += (size_t)NCH * q->k[j] * 2048;
        if (p.w2[j] != expect && !v2w_dry(stream)) return V2W_E_ARG;
        expect += (size_t)NCH * q->k[j] * 2048;
        p.ntot += 2 * NCH * q->k[j];
    }
    p.nto = (W - 2 * p.h2max) & ~3;
    if (p.nto < W / 2) return V2W_E_SHAPE;
    const int hsum = p.h1max + p.h2max;
    p.xoff = ((hsum + 3) & ~3) - hsum;
    p.xrows = (p.xoff + W + 2 * p.h1max + 3) & ~3;
    p.ntl = (q->L + p.nto - 1) / p.nto;
    p.vec4 = (q->L % 4 == 0) && ((reinterpret_cast<uintptr_t>(q->in) & 15) == 0);
    const size_t lds = (size_t)(p.xrows + W) * ROWB + (size_t)(2 * V2W_SS_MAXB + 2) * C * sizeof(float);
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    if (v2w_dry(stream)) return 0;
    auto kern = q->bf16 ? stage_split_kernel<NCH, NI, WN, true> : stage_split_kernel<NCH, NI, WN, false>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(q->B * p.ntl), dim3(64 * WN), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

int v2w_resblock2_stage_bf16(const v2w_stage_split_args* a, hipStream_t stream, int* up_tiles_out = nullptr);   // v2w_stage_bf16.hip

static int stage_split_dispatch(const v2w_stage_split_args* a, void* stream, int* up_tiles_out) {
    if (!a || (!a->in && !a->rb1) || (!a->out && !a->post_out && !a->up_out && !a->rb1) || a->nk < 1 || a->nk > V2W_SS_MAXB) return V2W_E_ARG;
    if (a->rb1) {                                                                          // ResBlock1 pair mode: bf16 tensors only
        if (!(a->bf16 && a->io_bf16 == 3)) return V2W_E_SHAPE;
        for (int j = 0; j < a->nk; ++j) if (!a->in_b[j] && !a->in) return V2W_E_ARG;
    }
    if (a->post_out && !(a->bf16 && a->io_bf16 == 3)) return V2W_E_SHAPE;                 // the fused tail exists on the bf16-tensor path only
    if (a->up_out) {                                                                       // ... and so does the fused upsampler
        if (a->post_out) return V2W_E_ARG;
        if (!(a->bf16 && a->io_bf16 == 3)) return V2W_E_SHAPE;
    }
    if (a->B <= 0 || a->C <= 0 || a->L <= 0) return V2W_E_ARG;
    if ((a->in_a == nullptr) != (a->in_s == nullptr)) return V2W_E_ARG;
    for (int j = 0; j < a->nk; ++j) {
        if (!a->wps1[j] || !a->wps2[j] || !a->sc1[j] || !a->sc2[j] || a->k[j] <= 0 || a->dil1[j] <= 0 || a->dil2[j] <= 0) return V2W_E_ARG;
        if ((a->k[j] & 1) == 0) return V2W_E_SHAPE;
    }
    if (a->bf16) {     // bf16 operands: the weights-in-registers kernel of v2w_stage_bf16.hip; shapes it does not take fall through
        const int rc = v2w_resblock2_stage_bf16(a, (hipStream_t)stream, up_tiles_out);
        if (rc != V2W_E_SHAPE) return rc;
    }
    if (a->io_bf16 || a->up_out || a->rb1) return V2W_E_SHAPE;       // this file's kernels read and write fp32 only
    if (a->C == 32) return launch_stage_split<2, 2, 4>(a, (hipStream_t)stream);      // 32 channels x 256 positions per workgroup
    if (a->C == 16) return launch_stage_split<1, 2, 4>(a, (hipStream_t)stream);      // 16 channels (MFMA rows zero-padded) x 256 positions
    return V2W_E_SHAPE;
}
extern "C" int v2w_resblock2_stage_split_fwd(const v2w_stage_split_args* a, void* stream) { return stage_split_dispatch(a, stream, nullptr); }

// Shape query (ABI v28): 0 when v2w_resblock2_stage_split_fwd would run this stage as ONE kernel, V2W_E_SHAPE / V2W_E_ARG as the call itself
// would answer.  Only the sizes (B, C, L, nk, k, dilations), the mode flags, slope and the ALIGNMENT of the tensor pointers that are set
// are read; NULL tensor / weight pointers stand for "aligned".  Generator._bf16_storage_kernels_exist asks this before a forward starts
// instead of restating the kernels' limits (tap count, halo) in Python.
extern "C" int v2w_resblock2_stage_split_config(const v2w_stage_split_args* a) {
    if (!a) return V2W_E_ARG;
    v2w_stage_split_args q = *a;
    float* const dummy = reinterpret_cast<float*>(static_cast<uintptr_t>(4096));     // aligned, never dereferenced
    if (!q.in) q.in = dummy;
    if (!q.out && !q.post_out && !q.up_u && !q.rb1) q.out = dummy;
    if (q.rb1) for (int j = 0; j < q.nk && j < V2W_SS_MAXB; ++j) { if (!q.in_b[j]) q.in_b[j] = dummy; if (!q.out_b[j]) q.out_b[j] = dummy; }
    if (q.up_u) {                      // a fused-upsampler query: up_u / up_k / up_slope are read
        if (!q.up_out) q.up_out = dummy;
        if (!q.up_wps) q.up_wps = dummy;
    }
    for (int j = 0; j < q.nk && j < V2W_SS_MAXB; ++j) {
        if (!q.wps1[j]) q.wps1[j] = dummy;
        if (!q.wps2[j]) q.wps2[j] = dummy;
        if (!q.sc1[j]) q.sc1[j] = dummy;
        if (!q.sc2[j]) q.sc2[j] = dummy;
    }
    return stage_split_dispatch(&q, V2W_DRY_STREAM, nullptr);
}
// Rows of up_stats_part ([rows][C / 2][2]) a fused-upsampler call (up_u != 0) fills; <= 0: the call would not run fused (error code or 0)
extern "C" int v2w_resblock2_stage_up_tiles(const v2w_stage_split_args* a) {
    if (!a || !a->up_u) return 0;
    v2w_stage_split_args q = *a;
    float* const dummy = reinterpret_cast<float*>(static_cast<uintptr_t>(4096));
    if (!q.in) q.in = dummy;
    if (!q.up_out) q.up_out = dummy;
    if (!q.up_wps) q.up_wps = dummy;
    for (int j = 0; j < q.nk && j < V2W_SS_MAXB; ++j) {
        if (!q.wps1[j]) q.wps1[j] = dummy;
        if (!q.wps2[j]) q.wps2[j] = dummy;
        if (!q.sc1[j]) q.sc1[j] = dummy;
        if (!q.sc2[j]) q.sc2[j] = dummy;
    }
    int n = 0;
    const int rc = stage_split_dispatch(&q, V2W_DRY_STREAM, &n);
    return rc == 0 ? n : rc;
}
