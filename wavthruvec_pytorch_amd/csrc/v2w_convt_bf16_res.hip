// leaky_relu -> ConvTranspose1d(k, stride U, padding (k - U) / 2) -> + bias (+ the BatchNorm partial sums of its output) on bf16 tensors with
// the input tile RESIDENT in LDS (models.py:128-129; BASELINE configs[2]: bf16 compute / fp32 accumulate, bf16 activation storage).
//
// The transposed conv is the 3-tap Conv1d over UP * C_out virtual rows of v2w_conv_bf16.hip (row = co * UP + phase, the polyphase taps in the
// fragments of v2w_pack_bf16_convt).  That kernel walks C_in in 32-channel chunks with a barrier each - and with 3 taps a chunk holds
// ~1.5 k cycles of MFMA issue against one exposed memory latency (~4 us): 259 us for a layer whose bytes take 40 us.  Here, as in
// v2w_conv_bf16_res.hip:
//   * every input channel of the tile is staged in ONE burst (64-byte rows, slots XOR-swizzled: conflict-free operand reads), one barrier,
//     then the MFMA loop runs over all (plane, tap) without another;
//   * U == UP (2, 4, 8): the phases of an output channel are ADJACENT accumulator registers of one lane - a lane holds U consecutive
//     output positions of a channel (or 4 of the 8), so the output leaves straight from the accumulators as 4- / 8-byte bf16 stores,
//     consecutive lanes = consecutive addresses: no LDS scratch, no transposition;
//   * the BatchNorm partial sums (sum, sum of squares of the fp32 values, before rounding) are reduced across the lanes of a column block
//     with DPP adds in a fixed order, across the waves of a tile through LDS in a fixed order: one slot per (tile, channel) as before;
//   * U = 5 (the generator's first upsampler, 8 virtual phases of which 5 exist): no register layout puts 5 consecutive outputs in a lane,
//     so the accumulators pass through a wave-private fp32 scratch that IS the output block [channel][5 x 64 positions] (over the dead input
//     tile) and leave as 8-byte bf16 stores along positions - the epilogue of the chunked kernel (v2w_conv_bf16.hip), without its 16
//     barriers and exposed memory latencies per tile.
//   * U = UP = 5 (round 4): the EXACT five phases - virtual rows co * 5 + phase, the second region of v2w_pack_bf16_convt's buffer - instead of
//     8 of which 3 are zero (37.5 % of the MFMA work of the kernel above).  160 rows = 32 channels per wave, 4 waves = 128 channels x 64
//     positions per workgroup, ONE workgroup per CU (the 512-channel input tile is 74 KB) with the whole register file: 10 accumulator blocks
//     per wave, every weight fragment feeds two MFMAs, every staged input tile serves 128 channels instead of 32.  Accumulators start at the
//     bias; the epilogue passes 32 positions at a time through a wave-private scratch [32 channels][5 x 32 outputs] (over the dead tile).
#include <type_traits>
#include "v2w_tile.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define V2W_CT_UNIT 2048

struct CtArgs {
    const unsigned short* in; const unsigned char* wps; const float* bias; unsigned short* out; float* stats_part;
    int B, Cin, CoutR, L;      // CoutR: real output channels; the kernel's rows are CoutR * UP
    int KV, hl;                // virtual taps and the left halo (input positions)
    int hla, xrows, ntl, ntiles;
    float slope;
};

__device__ __forceinline__ unsigned int ct_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float ct_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float ct_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ int ct_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL> __device__ __forceinline__ float ct_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ct_readlane(float v, int l) {           // (the builtin is typed int: a float argument would be CONVERTED)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum over the 16 lanes of a DPP row, in every lane of the row (fixed order): quad xor 1, quad xor 2, half mirror, mirror
__device__ __forceinline__ float ct_row_sum(float v) {
    v += ct_dpp<0xB1>(v);
    v += ct_dpp<0x4E>(v);
    v += ct_dpp<0x141>(v);
    v += ct_dpp<0x140>(v);
    return v;
}

// workgroups (of 4 waves) per CU the register budget is cut for
constexpr int ct_wgs(int mi, int ni) { return mi * ni >= 8 ? 2 : (mi * ni >= 4 ? 3 : 4); }
#define V2W_CT5_SCRATCH (4 * 16 * 160 * 4)      // UP = 5: four waves x [16 channels][160 outputs] fp32

// U: the stride (= UP, or 5 with UP = 8: whole tiles only, NI even)
template <int MI, int NI, int WM, int WN, int UP, int U = UP>
__global__ void __launch_bounds__(64 * WM * WN, UP == 5 ? 1 : ct_wgs(MI, NI))
convt_bf16_res_kernel(const CtArgs a) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTH = 64 * WM * WN, MT = 32 * MI * WM, NT = 32 * NI * WN;
    constexpr int NPF = (8 * ((NT + 16) / 4) + NTH - 1) / NTH;                 // staging items of one 32-channel plane per thread
    static_assert(UP == 2 || UP == 4 || UP == 8 || (UP == 5 && MI == 5 && WM == 4 && WN == 1), "phases per channel");
    static_assert(U == UP || (U == 5 && UP == 8 && NI % 2 == 0), "stride");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];

    const int Cin = ct_uni(a.Cin), L = ct_uni(a.L), xrows = ct_uni(a.xrows), hla = ct_uni(a.hla), KV = ct_uni(a.KV);
    const int CoutR = ct_uni(a.CoutR);
    const int nch = Cin >> 5, psz = xrows * 64;
    // bias of this M-tile's channels [MT / UP]: behind the tile - and behind the waves' scratch of the U = 5 epilogue, which reads it (a tile
    // of few input channels is smaller than that scratch)
    constexpr int SCR8 = (U != UP) ? WM * WN * 2048 * 4 : 0;
    float* const btab = reinterpret_cast<float*>(smem_c + (nch * psz > SCR8 ? nch * psz : SCR8));
    float* const red = btab + MT / UP;                                          // [WN][MT / UP][2] partial sums of the waves
    const float slope = a.slope;

    const int mtiles = (CoutR * UP) / MT;
    const int id = blockIdx.x;
    const int grp = id / (8 * mtiles), rem = id % (8 * mtiles);
    const int mt = rem >> 3;
    const int tile = grp * 8 + (rem & 7);
    if (tile >= a.ntiles) return;
    const int b = tile / a.ntl;
    const int n0 = (tile % a.ntl) * NT;
    const int m0 = mt * MT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = ct_uni(tid >> 6);
    const int lr = lane & 31, hk = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MI);
    const int wn0 = (wave % WN) * (32 * NI);
    const int pos0 = n0 - hla;

    for (int c = tid; c < MT / UP; c += NTH) btab[c] = a.bias ? a.bias[m0 / UP + c] : 0.f;

    // ---- stage lrelu(in) as bf16: every plane of the tile, two planes of loads in flight (an item = 4 channels x 4 positions)
    {
        const int nq = xrows >> 2;
        unsigned poff[NPF];
        bool in_img[NPF], in_seq[NPF];
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH;
            const int cq = idx & 7, pq = idx >> 3;
            in_img[s] = pq < nq;
            const int pos = pos0 + pq * 4;
            in_seq[s] = in_img[s] && pos >= 0 && pos < L;
            poff[s] = (unsigned)(4 * cq * L + (in_seq[s] ? pos : 0)) * 2u;
            asm volatile("" : "+v"(poff[s]));
        }
        auto prefetch = [&](int ch, u32x2 (&pf)[NPF][4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned char* base = reinterpret_cast<const unsigned char*>(a.in) + (size_t)(b * Cin + 32 * ch + i) * L * 2;
#pragma unroll
                for (int s = 0; s < NPF; ++s) pf[s][i] = *gptr<const u32x2>(base + poff[s]);
            }
        };
        auto commit = [&](int ch, const u32x2 (&pf)[NPF][4]) {
            unsigned char* const plane = smem_c + ch * psz;
#pragma unroll
            for (int s = 0; s < NPF; ++s) {
                if (!in_img[s]) continue;
                const int idx = tid + s * NTH;
                const int cq = idx & 7, pq = idx >> 3;
                unsigned char* dst = plane + pq * 256 + ((((cq >> 1) ^ (pq & 3)) << 4) | ((cq & 1) << 3));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xv = (e & 1) ? ct_hi(pf[s][i][e >> 1]) : ct_lo(pf[s][i][e >> 1]);
                        v[i] = fmaxf(xv, xv * slope);
                    }
                    u32x2 w = {ct_pack2(v[0], v[1]), ct_pack2(v[2], v[3])};
                    if (!in_seq[s]) w = u32x2{0u, 0u};
                    *reinterpret_cast<u32x2*>(dst + e * 64) = w;
                }
            }
        };
        {
        u32x2 pf0[NPF][4], pf1[NPF][4];
        prefetch(0, pf0);
        if (nch > 1) prefetch(1, pf1);
        __builtin_amdgcn_sched_barrier(0);
        int ch = 0;
        for (; ch + 2 < nch; ch += 2) {
            commit(ch, pf0);
            prefetch(ch + 2, pf0);
            __builtin_amdgcn_sched_barrier(0);
            commit(ch + 1, pf1);
            if (ch + 3 < nch) prefetch(ch + 3, pf1);
            __builtin_amdgcn_sched_barrier(0);
        }
        commit(ch, pf0);
        if (ch + 1 < nch) commit(ch + 1, pf1);
        }
    }
    __syncthreads();

    // ---- the MFMA loop over the resident tile: nch * KV taps in pairs (ring slots 0 / 1 even taps, 2 / 3 odd taps)
    acc_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if constexpr (UP == 5) acc[i][j][e] = btab[32 * wave + (32 * i + 8 * (e >> 2) + 4 * hk + (e & 3)) / 5];    // (the bias)
                else acc[i][j][e] = 0.f;
            }
    {
        const unsigned lane16 = (unsigned)lane * 16u;
        auto mfma = [&](acc_t c, u32x4 av, u32x4 bv) {
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, av), __builtin_bit_cast(b8, bv), c, 0, 0, 0);
        };
        auto baddr = [&](int ch, int row) { return (unsigned)(ch * psz + row * 64 + ((hk ^ ((row >> 2) & 3)) << 4)); };
        const int K = KV;
        const int nst = 2 * nch * K;
        const unsigned char* ap[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) ap[i] = a.wps + (size_t)((m0 + wm0) / 32 + i) * nst * V2W_CT_UNIT;
        constexpr int RD = 2;                                                   // taps of weight fragments in flight (2 k-steps each)
        u32x4 ar[2 * RD][MI];
        auto load_frag = [&](u32x4 (&av)[MI], int ch, int s, int t) {
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            const int chc = ch < nch ? ch : nch - 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                av[i] = *gptr<const u32x4>(ap[i] + (size_t)((2 * chc + s) * K + t) * V2W_CT_UNIT + l16);
        };
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            int c1 = 0, t1 = d;
            while (t1 >= K) { t1 -= K; ++c1; }
            load_frag(ar[2 * d], c1, 0, t1);
            load_frag(ar[2 * d + 1], c1, 1, t1);
        }
        __builtin_amdgcn_sched_barrier(0);
        int ch = 0, t = 0, qc = 0, qt = RD;
        while (qt >= K) { qt -= K; ++qc; }
        const int r0 = hla - ct_uni(a.hl) + wn0 + lr;
        unsigned xt = baddr(0, r0);
        u32x4 bb[2][NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            bb[0][j] = *reinterpret_cast<const u32x4*>(smem_c + xt + j * 2048);
            bb[1][j] = *reinterpret_cast<const u32x4*>(smem_c + (xt ^ 32u) + j * 2048);
        }
        auto kstep = [&](auto bs_c, const u32x4 (&av)[MI], unsigned nxt) {
            constexpr int bs = decltype(bs_c)::value;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mfma(acc[i][j], av[i], bb[bs][j]);
                bb[bs][j] = *reinterpret_cast<const u32x4*>(smem_c + nxt + j * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto tap = [&](auto par_c) {
            constexpr int S0 = 2 * decltype(par_c)::value;
            int nch_ = ch, nt_ = t + 1;
            if (nt_ >= K) { nt_ = 0; ++nch_; }
            if (nch_ >= nch) { nch_ = ch; nt_ = t; }
            const unsigned xn = baddr(nch_, r0 + nt_);
            kstep(std::integral_constant<int, 0>{}, ar[S0], xn);
            load_frag(ar[S0], qc, 0, qt);
            __builtin_amdgcn_sched_barrier(0);
            kstep(std::integral_constant<int, 1>{}, ar[S0 + 1], xn ^ 32u);
            load_frag(ar[S0 + 1], qc, 1, qt);
            __builtin_amdgcn_sched_barrier(0);
            if (++qt >= K) { qt = 0; ++qc; }
            ch = nch_; t = nt_; xt = xn;
        };
        const int TT = nch * K;
        int g = 0;
        for (; g + 1 < TT; g += 2) { tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); }
        if (g < TT) tap(std::integral_constant<int, 0>{});
    }

    // ---- epilogue: + bias, partial sums, bf16 stores straight from the accumulators.  Block (i, j), register quad g, lane (lr, hk):
    // virtual rows 32 i + 8 g + 4 hk + {0..3} at input position q = n0 + wn0 + 32 j + lr, i.e.
    //   UP = 4: channel 8 i + 2 g + hk, outputs 4 q + {0..3};   UP = 8: channel 4 i + g, outputs 8 q + 4 hk + {0..3};
    //   UP = 2: channels 16 i + 4 g + 2 hk + {0, 1}, outputs 2 q + {0, 1} each
    constexpr int CPB = 32 / UP;                                                // channels per 32-row block
    const int Lout = L * U;
    const bool stats = a.stats_part != nullptr;
    if constexpr (UP == 5) {
        // ---- the exact five phases: block (i, j), register e, lane (lr, hk) = row 32 i + 8 (e / 4) + 4 hk + e % 4 of this wave's 160 =
        // (channel row / 5, phase row % 5) at input position 32 j + lr.  Sixteen channels (rows 0..79 / 80..159: whole register quads) of 32
        // positions at a time -> scratch[channel][5 lr + phase] (10 KB a wave: two workgroups' tiles fit a CU); float4 number lane + 64 g of
        // the scratch = 4 consecutive outputs of channel (lane + 64 g) / 40: an 8-byte bf16 store; lane (channel, quarter) adds a quarter row
        __syncthreads();                                                        // every wave has left the MFMA loop: the input tile is dead
        float* const scr = reinterpret_cast<float*>(smem_c) + wave * (16 * 160);
        const int co0 = (m0 + wm0) / 5;
        unsigned char* const ob0 = reinterpret_cast<unsigned char*>(a.out) + (size_t)b * CoutR * Lout * 2;
        const int cS = lane & 15, qS = lane >> 4;
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r0_ = 32 * i + 8 * (e >> 2) + (e & 3), r1_ = r0_ + 4;
                        if (r0_ / 80 != h) continue;                             // (r0_ and r0_ + 4 share the half: quads of 8 rows)
                        const int o0 = (r0_ / 5 - 16 * h) * 160 + r0_ % 5, o1 = (r1_ / 5 - 16 * h) * 160 + r1_ % 5;
                        scr[(hk ? o1 : o0) + 5 * lr] = acc[i][j][e];
                    }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned ub = (unsigned)((co0 + 16 * h) * Lout + 5 * (n0 + 32 * j)) * 2u;       // (uniform)
#pragma unroll
                for (int g = 0; g < 10; ++g) {
                    const int idx = lane + 64 * g;
                    const int c = idx / 40, q4 = idx - 40 * c;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(scr + 4 * idx);
                    *gptr<u32x2>(ob0 + ub + (unsigned)(c * Lout + 4 * q4) * 2u) = u32x2{ct_pack2(v[0], v[1]), ct_pack2(v[2], v[3])};
                }
                if (stats) {
                    const float* row = scr + cS * 160 + qS * 40;
#pragma unroll 8
                    for (int k = 0; k < 40; ++k) {
                        int kk = k + 2 * cS; kk = kk >= 40 ? kk - 40 : kk;     // rotated start: the 16 channels x 4 quarters read 64 banks
                        const float v = row[kk];
                        s1[h] += v; s2[h] = fmaf(v, v, s2[h]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (stats) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float t1 = s1[h] + __shfl_xor(s1[h], 16, 64), t2 = s2[h] + __shfl_xor(s2[h], 16, 64);
                t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
                if (lane < 16) {
                    gptr<float>(a.stats_part)[((size_t)tile * CoutR + co0 + 16 * h + lane) * 2 + 0] = t1;
                    gptr<float>(a.stats_part)[((size_t)tile * CoutR + co0 + 16 * h + lane) * 2 + 1] = t2;
                }
            }
        }
        return;
    } else if constexpr (U != UP) {
        // ---- U = 5: 64 input positions (two column blocks) of a 32-row block at a time -> scratch [4 channels][5 x 64 outputs] (the output's
        // own layout: block (i, j), register quad eg, lane (lr, hk) holds phases 4 hk + {0..3} of channel eg at input position 32 j + lr)
        // -> float4 number lane + 64 g of the scratch is 4 consecutive outputs of channel (lane + 64 g) / 80: + bias, sums, an 8-byte bf16 store.
        constexpr int RW = 16 * U, G = CPB * U / 4, ORS = U * 64;               // float4s per channel row of a pass; float4s per lane; scratch row
        constexpr int PARTS = 64 / (2 * CPB), NPER = RW / PARTS;
        static_assert(CPB == 4 && G == 5 && RW % PARTS == 0, "U = 5 on 8 virtual phases");
        __syncthreads();                                                        // every wave has left the MFMA loop: the input tile is dead
        float* const scr = reinterpret_cast<float*>(smem_c) + wave * 2048;
        unsigned char* const ob0 = reinterpret_cast<unsigned char*>(a.out) + (size_t)b * CoutR * Lout * 2;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f32x4 sa[G], sq[G];
#pragma unroll
            for (int g = 0; g < G; ++g) sa[g] = sq[g] = zero4;
            const int co0 = (m0 + wm0 + i * 32) / UP;
#pragma unroll
            for (int jh = 0; jh < NI; jh += 2) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const acc_t& ac = acc[i][jh + jj];
                    float* const sc = scr + U * (jj * 32 + lr);
#pragma unroll
                    for (int eg = 0; eg < 4; ++eg) {
                        sc[eg * ORS + 4 * hk] = ac[4 * eg];
                        if (hk == 0) { sc[eg * ORS + 1] = ac[4 * eg + 1]; sc[eg * ORS + 2] = ac[4 * eg + 2]; sc[eg * ORS + 3] = ac[4 * eg + 3]; }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned ub = (unsigned)(co0 * Lout + U * (n0 + wn0 + jh * 32)) * 2u;      // (uniform)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int idx = lane + 64 * g;
                    const int cl = (idx >= RW) + (idx >= 2 * RW) + (idx >= 3 * RW), c4 = idx - cl * RW;
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + 4 * idx);
                    const float bias = btab[(wm0 / UP) + CPB * i + cl];
                    v += f32x4{bias, bias, bias, bias};
                    if (stats) { sa[g] += v; sq[g] += v * v; }
                    *gptr<u32x2>(ob0 + ub + (unsigned)(cl * Lout + 4 * c4) * 2u) = u32x2{ct_pack2(v[0], v[1]), ct_pack2(v[2], v[3])};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (stats) {
                // per-lane sums -> scratch[g * 64 + lane] = (sum, sumsq): slot k belongs to channel k / RW; lane = (channel, part, stat) adds NPER
                // slots in a fixed order, the parts of a channel meet in a fixed shuffle tree
#pragma unroll
                for (int g = 0; g < G; ++g)
                    *reinterpret_cast<f32x2*>(scr + 2 * (g * 64 + lane)) =
                        f32x2{(sa[g][0] + sa[g][1]) + (sa[g][2] + sa[g][3]), (sq[g][0] + sq[g][1]) + (sq[g][2] + sq[g][3])};
                const int stat = lane & 1, part = (lane >> 1) % PARTS, c = lane / (2 * PARTS);
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < NPER; ++k) t += scr[2 * (c * RW + part * NPER + (k + (lane >> 1)) % NPER) + stat];
#pragma unroll
                for (int off = 2; off < 2 * PARTS; off <<= 1) t += __shfl_xor(t, off, 64);
                if (part == 0) red[((wave % WN) * (MT / UP) + (wm0 / UP) + CPB * i + c) * 2 + stat] = t;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
    unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + (size_t)b * CoutR * Lout * 2;
    const int cw0 = (m0 + wm0) / UP;                                            // first channel of this wave (inside the tensor)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            constexpr int NC = UP == 2 ? 2 : 1;                                 // channels of a register quad
            const int cl = UP == 4 ? 2 * g + hk : (UP == 8 ? g : 4 * g + 2 * hk);      // channel inside block i (+ 1 for the second of UP = 2)
            float bias[NC], s1[NC], s2[NC];
#pragma unroll
            for (int n = 0; n < NC; ++n) { bias[n] = btab[(wm0 / UP) + CPB * i + cl + n]; s1[n] = s2[n] = 0.f; }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int q = wn0 + lr;
                asm volatile("" : "+v"(q));
                q += n0 + 32 * j;
                const bool ok = q < L;
                float v[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    v[x] = acc[i][j][4 * g + x] + bias[UP == 2 ? (x >> 1) : 0];
                    if (!ok) v[x] = 0.f;
                    s1[UP == 2 ? (x >> 1) : 0] += v[x];
                    s2[UP == 2 ? (x >> 1) : 0] = fmaf(v[x], v[x], s2[UP == 2 ? (x >> 1) : 0]);
                }
                if (ok) {
                    const int c = cw0 + CPB * i + cl;
                    if constexpr (UP == 2) {
                        *gptr<unsigned>(obase + (unsigned)(c * Lout + 2 * q) * 2u) = ct_pack2(v[0], v[1]);
                        *gptr<unsigned>(obase + (unsigned)((c + 1) * Lout + 2 * q) * 2u) = ct_pack2(v[2], v[3]);
                    } else {
                        const int o = UP == 4 ? 4 * q : 8 * q + 4 * hk;
                        *gptr<u32x2>(obase + (unsigned)(c * Lout + o) * 2u) = u32x2{ct_pack2(v[0], v[1]), ct_pack2(v[2], v[3])};
                    }
                }
            }
            if (stats) {
                // the 32 lanes that share hk hold one channel's columns (UP = 8: both halves hold the same channel): DPP row sums, then the
                // rows of a half (and the halves) in a fixed order
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    const float r1 = ct_row_sum(s1[n]), r2 = ct_row_sum(s2[n]);
                    const float a1 = ct_readlane(r1, 0) + ct_readlane(r1, 16);
                    const float b1 = ct_readlane(r1, 32) + ct_readlane(r1, 48);
                    const float a2 = ct_readlane(r2, 0) + ct_readlane(r2, 16);
                    const float b2 = ct_readlane(r2, 32) + ct_readlane(r2, 48);
                    float* rd = red + ((wave % WN) * (MT / UP) + (wm0 / UP) + CPB * i) * 2;
                    if constexpr (UP == 8) {
                        if (lane == 0) { rd[2 * g] = a1 + b1; rd[2 * g + 1] = a2 + b2; }
                    } else if constexpr (UP == 4) {
                        if (lane == 0) { rd[2 * (2 * g)] = a1; rd[2 * (2 * g) + 1] = a2; rd[2 * (2 * g + 1)] = b1; rd[2 * (2 * g + 1) + 1] = b2; }
                    } else {
                        if (lane == 0) { rd[2 * (4 * g + n)] = a1; rd[2 * (4 * g + n) + 1] = a2; rd[2 * (4 * g + 2 + n)] = b1; rd[2 * (4 * g + 2 + n) + 1] = b2; }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }     // (U == UP)
    if (stats) {
        __syncthreads();
        for (int c = tid; c < MT / UP; c += NTH) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { t1 += red[(w * (MT / UP) + c) * 2]; t2 += red[(w * (MT / UP) + c) * 2 + 1]; }
            gptr<float>(a.stats_part)[((size_t)tile * CoutR + m0 / UP + c) * 2 + 0] = t1;
            gptr<float>(a.stats_part)[((size_t)tile * CoutR + m0 / UP + c) * 2 + 1] = t2;
        }
    }
}

template <int MI, int NI, int WM, int WN, int UP, int U = UP>
int launch_ct(CtArgs p, hipStream_t stream, int* ntiles_out, int32_t* cfg) {
    constexpr int NTH = 64 * WM * WN, MT = 32 * MI * WM, NT = 32 * NI * WN;
    if ((p.CoutR * UP) % MT != 0 || p.Cin % 32 != 0) return V2W_E_SHAPE;
    if ((U != UP || UP == 5) && p.L % NT != 0) return V2W_E_SHAPE;   // the scratch epilogues of U = 5 serve whole tiles
    const int nch = p.Cin / 32;
    if (nch > 1 && (nch & 1)) return V2W_E_SHAPE;                    // planes are staged in pairs
    p.hla = (p.hl + 3) & ~3;
    const int hr = p.KV - 1 - p.hl;
    if (p.hla > 8 || hr > 8) return V2W_E_SHAPE;
    p.ntl = (p.L + NT - 1) / NT;
    p.ntiles = p.B * p.ntl;
    p.xrows = (p.hla + NT + hr + 3) & ~3;
    size_t tile_b = (size_t)nch * p.xrows * 64;
    if (U != UP && tile_b < (size_t)WM * WN * 2048 * sizeof(float)) tile_b = (size_t)WM * WN * 2048 * sizeof(float);   // the waves' scratch overlays the tile
    size_t lds = tile_b + (size_t)(MT / UP) * (1 + 2 * WN) * sizeof(float);
    if (UP == 5 && lds < V2W_CT5_SCRATCH) lds = V2W_CT5_SCRATCH;     // (bias table and tile are dead by then)
    constexpr int WGS = UP == 5 ? 1 : ct_wgs(MI, NI);                // workgroups per CU the register budget allows (4-wave workgroups)
    if (lds * (WGS > 2 ? 2 : WGS) > 160 * 1024) return V2W_E_SHAPE;
    if (cfg) { const int32_t c[10] = {MI, NI, WM, WN, UP, 102 /* = this kernel */, 1, 1, 32, U}; for (int i = 0; i < 10; ++i) cfg[i] = c[i]; }
    if (ntiles_out) { *ntiles_out = p.ntiles; return 0; }
    const int grid = ((p.ntiles + 7) / 8) * 8 * ((p.CoutR * UP) / MT);
    auto kern = convt_bf16_res_kernel<MI, NI, WM, WN, UP, U>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(NTH), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

// Called by the dispatcher of v2w_conv_bf16.hip for bf16 tensors (io_bf16 == 3), aligned, L % 4 == 0, U == UP in {2, 4, 8}.
// hl / KV: the virtual conv's geometry (convt_geom).  V2W_E_SHAPE: the chunked kernel runs instead.
int v2w_convt1d_bf16_res(const v2w_convt1d_args* a, int UP, int hl, int KV, hipStream_t stream, int* ntiles_out, int32_t* cfg) {
    if (a->io_bf16 != 3 || (a->u != UP && !(a->u == 5 && UP == 8)) || a->L % 4 != 0) return V2W_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(a->in) & 15) || (reinterpret_cast<uintptr_t>(a->out) & 15)) return V2W_E_SHAPE;    // (queries carry the pointers too)
    if ((long long)a->C_out * a->L * UP * 2 >= (1ll << 31) || (long long)a->C_in * a->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;
    if (!(a->slope > 0.f && a->slope <= 1.f)) return V2W_E_SHAPE;
    CtArgs p{};
    p.in = reinterpret_cast<const unsigned short*>(a->in); p.wps = reinterpret_cast<const unsigned char*>(a->wp); p.bias = a->bias;
    p.out = reinterpret_cast<unsigned short*>(a->out); p.stats_part = a->stats_part;
    p.B = a->B; p.Cin = a->C_in; p.CoutR = a->C_out; p.L = a->L; p.KV = KV; p.hl = hl; p.slope = a->slope;
    const int rows = a->C_out * UP;
    if (a->u == 5) {
        // the exact five phases (fragments: the second region of v2w_pack_bf16_convt's buffer, behind the C_out * 8 virtual rows): 128 channels x 64
        // positions per workgroup; whole tiles, C_out in steps of 128
        if (a->C_out % 128 == 0 && a->L % 64 == 0 && a->C_in % 32 == 0) {
            CtArgs q = p;
            q.wps = p.wps + (size_t)(a->C_out * 8 / 32) * (a->C_in / 16) * KV * V2W_CT_UNIT;
            const int rc = launch_ct<5, 2, 4, 1, 5, 5>(q, stream, ntiles_out, cfg);
            if (rc != V2W_E_SHAPE) return rc;
        }
        // 256 virtual rows x 64 positions, 4 waves of 64 x 64: the 512-channel input tile of the generator's first upsampler fits twice per CU
        if (rows % 256 == 0) return launch_ct<2, 2, 4, 1, 8, 5>(p, stream, ntiles_out, cfg);
        return V2W_E_SHAPE;
    }
    if (UP == 4) {
        if (rows % 256 == 0) return launch_ct<4, 2, 2, 2, 4>(p, stream, ntiles_out, cfg);       // 256 rows x 128 positions
        if (rows % 128 == 0) return launch_ct<2, 4, 2, 2, 4>(p, stream, ntiles_out, cfg);       // 128 rows x 256 positions
        if (rows % 64 == 0) return launch_ct<2, 4, 1, 4, 4>(p, stream, ntiles_out, cfg);        // 64 rows x 512 positions
        return V2W_E_SHAPE;
    }
    if (UP == 2) {
        // (measured at configs[2], ups.3 / ups.4: tiles of 256 positions 228 / 190 us, of 1024 positions - / 188 us, against 213 / 172 us here)
        if (rows % 64 == 0) return launch_ct<2, 4, 1, 4, 2>(p, stream, ntiles_out, cfg);        // 64 rows x 512 positions
        if (rows % 32 == 0) return launch_ct<1, 4, 1, 4, 2>(p, stream, ntiles_out, cfg);        // 32 rows x 512 positions
        return V2W_E_SHAPE;
    }
    if (UP == 8) {
        if (rows % 256 == 0) return launch_ct<4, 2, 2, 2, 8>(p, stream, ntiles_out, cfg);
        if (rows % 128 == 0) return launch_ct<2, 4, 2, 2, 8>(p, stream, ntiles_out, cfg);
        return V2W_E_SHAPE;
    }
    return V2W_E_SHAPE;
}
