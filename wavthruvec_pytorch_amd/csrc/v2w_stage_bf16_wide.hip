// Whole residual section of a WIDE ResBlock2 stage (C = 64 / 128 / 256) in ONE kernel on the bf16 matrix pipe, bf16 tensors in and out
// (BASELINE configs[2]: bf16 compute / fp32 accumulate, bf16 activation storage); models.py:135-141 with ResBlock2.forward inlined:
//   out = ( sum_j [ t1_j + conv_{k_j,d2_j}(lrelu(t1_j)) + b2_j ] ) / nk ,   t1_j = x + conv_{k_j,d1_j}(lrelu(x)) + b1_j ,   x = a * in + s
//
// As six separate convolutions these stages move 15 activation tensors per stage through HBM (x read three times, three t1 written and
// read twice, two partial sums written and read) and each launch pays its own staging and epilogue; here x is read ONCE, the output
// written ONCE, and t1_j never leaves the chip:
//   * one workgroup of 8 waves per CU owns a window of W positions of all C channels.  LDS holds the activated x tile
//     [C / 32 planes][W + 2 h1max rows][64 B] and the activated t1 tile [C / 32][W + 2 h2max][64 B] (both position-major bf16, the 16-byte
//     slots of a row XOR-swizzled by (row >> 2) & 3: conflict-free ds_read_b128 operands at every tap offset - v2w_conv_bf16_res.hip);
//   * per branch: conv1_j on the window (MFMA over every (plane, tap) of the resident x tile, no barrier inside) -> t1 = acc + x
//     (the residual rebuilt from the tile itself, in the accumulator's layout) -> lrelu -> bf16 -> the t1 tile, 8-byte LDS stores straight
//     from the accumulator layout (no transposition: rows ARE positions) -> barrier -> conv2_j over the t1 tile, accumulating onto ONE
//     fp32 accumulator that runs over the branches and already holds sum_j (t1_j + b2_j) in fp32 (t1 is never rounded on that path);
//   * valid outputs = window - h2max columns per side (conv2's halo); they leave through an fp32 LDS scratch (the dead t1 tile) as
//     8-byte bf16 stores along positions;
//   * weights: the fragments of v2w_pack_bf16 / v2w_split_pack_batch from L2 through a four-slot register ring, three k-steps ahead.
#include <type_traits>
#include "v2w_tile.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define V2W_WS_MAXB 4
#define V2W_WS_UNIT 2048     // byte pitch of the packed fragments of one (32-row block, 16-channel k-step, tap)
#define V2W_WS_RING 2        // taps of weight fragments in flight per wave (2 or 4: measured the same, 1074 vs 1092 us at C = 128)
// Measured inside this kernel in rounds 3-4 and removed again ("measured and rejected"): a persistent tile loop with the next
// tile's x loads in flight under the stores (0-10 % slower: hardware dispatch of one workgroup per tile keeps the workgroups of a CU out of
// phase, which is what hides the epilogues); a conv's first fragments requested before the barrier / epilogue in front of it (within noise);
// s_setprio schemes for the younger half of the workgroup (no gain); 64 x 64 / 64 x 128 outputs per wave (slower).

struct WideArgs {
    const unsigned short* in; const float* in_a; const float* in_s;
    const unsigned char* w1[V2W_WS_MAXB]; const float* bias1[V2W_WS_MAXB];
    const unsigned char* w2[V2W_WS_MAXB]; const float* bias2[V2W_WS_MAXB];
    int K[V2W_WS_MAXB], d1[V2W_WS_MAXB], d2[V2W_WS_MAXB];
    unsigned short* out;
    int nk, B, C, L;
    int h1max, h2max, xoff, xrows, trows, nto, ntl, ntiles;
    float slope, inv_slope, out_div;
    // fused tail (C = 16 only): leaky_relu(post_slope) -> conv_post (post_k taps, C -> 1) -> tanh (models.py:143-145) of the stage's output;
    // `out` is not written.  hout = (post_k - 1) / 2 columns of the valid window per side feed the taps only (0 without the tail).
    const float* post_w; const float* post_b; float* post_out;
    int post_k, hout;
    float post_slope;
    // fused upsampler (UPF > 0 instantiations): leaky_relu(up_slope) -> ConvTranspose1d(2 UPF taps, stride UPF, C -> C / 2) + bias of the stage's
    // output (models.py:128-129 of the NEXT stage) with its BatchNorm partial sums; `out` is not written.  up_w: the virtual 3-tap fragments of
    // v2w_pack_bf16_convt; up_out (B, C / 2, UPF L) bf16; up_stats [ntiles][C / 2][2] or NULL.  hout = 1: a column of the valid window per
    // side feeds the taps only.
    const unsigned char* up_w; const float* up_bias; unsigned short* up_out; float* up_stats;
    float up_slope;
    // ResBlock1 pair mode (rb1 != 0; run-time block sets only; models.py:37-44): nk independent PROBLEMS, one conv pair each, in one launch -
    //   out_b[p] = ( x_p + conv_{k_p, d2_p}(lrelu(conv_{k_p, d1_p}(lrelu x_p) + b1_p)) + b2_p  [+ add0 + add1] ) / out_div ,  x_p = a * in_b[p] + s
    // - the p-th branch of a stage at one of its three (dilated conv, conv) pairs: the residual goes to the OUTPUT, not to the intermediate.
    // Tiles [p * ntiles1, (p + 1) * ntiles1) belong to problem p; add0 / add1 (the last pair of the last branch: the other branches' results).
    int rb1, ntiles1;
    const unsigned short* in_b[V2W_WS_MAXB]; unsigned short* out_b[V2W_WS_MAXB];
    const unsigned short* add0; const unsigned short* add1;
};

__device__ __forceinline__ unsigned int ws_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float ws_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float ws_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// min without the canonicalisation fminf() carries (v_max_f32 x, x, x in front of every v_min_f32: the operands here are bf16 bit patterns moved
// into a float, which the compiler cannot prove quiet): finite operands only
__device__ __forceinline__ float ws_min(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -__builtin_inff()); }   // (the median of (a, b, -inf))
__device__ __forceinline__ int ws_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <int CTRL> __device__ __forceinline__ float ws_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float ws_readlane(float v, int l) {           // (the builtin is typed int: a float argument would be CONVERTED)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum over the 16 lanes of a DPP row, in every lane of the row (fixed order): quad xor 1, quad xor 2, half mirror, mirror
__device__ __forceinline__ float ws_row_sum(float v) {
    v += ws_dpp<0xB1>(v);
    v += ws_dpp<0x4E>(v);
    v += ws_dpp<0x141>(v);
    v += ws_dpp<0x140>(v);
    return v;
}
template <typename T> __device__ __forceinline__ T* ws_uni(T* v) { pin_s(v); return v; }

// MI x NI blocks of 32 x 32 per wave, WM x WN waves: C = 32 MI WM channels, window W = 32 NI WN positions
// OCC = waves per SIMD the register budget is cut for: 8-wave workgroups (one per CU) and 4-wave workgroups at two per CU: 2 (256 registers);
// a 4-wave workgroup alone on its CU: 1 (the whole 512-register file)
// CH = channels of a plane that exist: 32, or 16 for the C = 16 stage (ONE plane of 32-byte rows, one k-step per tap, the MFMA's rows 16-31
// are the zero rows of the packed fragments: half of every MFMA is padding, on a stage that the vector ALU and the memory bound anyway)
// WLDS: the weight fragments reach the waves through an LDS ring filled by LDS-DMA (global_load_lds_dwordx4), each fragment fetched ONCE per
// workgroup and tap instead of once per wave that needs it.  Measured why: with the fragments loaded straight into registers the 8 waves of a
// C = 128 tile ask the CU's vector memory pipe for 2 KB per 4 MFMAs each - 128 B / clk at full matrix rate against the 64 B / clk it delivers;
// the conv phases ran at 57 % of the MFMA issue rate (71 % with the loads compiled out).
// STD: the kernel is compiled for the generator's own residual blocks - three branches of 3 / 7 / 11 taps, dilation 1 in the first convs and 3 in
// the second (hparams.py:42-43 with ResBlock2) - with every tap count and dilation a compile-time constant (conv_ct below); the launcher
// checks the arguments against it.  !STD: any branch count / kernel size / dilation at run time.
// UPF: 0, or the stride (2 / 4) of the NEXT stage's transposed conv, run here on the stage's output while it is still on chip (STD only)
// TSP (round 5; MI = 2, fused upsampler only): the t1 tile holds HALF the channels.  At C = 256 the x and t1 tiles of all channels bound the
// window to 128 positions - 96 valid outputs per tile, a third of the MFMA work spent on halo columns, 1.75 residencies of the chip at
// B = 32 x T = 256.  Here every branch runs in two passes: pass h computes the conv1 rows of the waves' row blocks 2 wm + h (t1 planes
// h, h + 2, ..: half a tile), conv2 accumulates that half of its reduction; the t1 of a pass is the wave's own accumulator block h, so the
// fp32 residual into the running output needs no exchange.  The window grows to 192 positions (160 valid: a fifth of halo work, ONE
// residency at B = 32 x T = 256); the fused upsampler runs its virtual rows in two halves (its accumulators would not fit otherwise), its
// operand tile z over the dead x tile, its scratch over the t1 half tile.
template <int MI, int NI, int WM, int WN, int OCC, int CH, bool WLDS, bool STD, int UPF = 0, bool TSP = false>
__global__ void __launch_bounds__(64 * WM * WN, OCC)
wide_stage_bf16_kernel(const WideArgs a) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    static_assert(CH == 32 || (CH == 16 && MI == 1 && WM == 1), "16 channels: one row block");
    static_assert(UPF == 0 || ((UPF == 2 || UPF == 4) && STD && CH == 32), "fused upsampler: the compile-time block set, strides 2 and 4");
    static_assert(!TSP || (MI == 2 && UPF > 0 && STD && CH == 32 && !WLDS), "half t1 tile: two row blocks per wave, the fused upsampler");
    constexpr int NTH = 64 * WM * WN, C = CH == 16 ? 16 : 32 * MI * WM, W = 32 * NI * WN, NCH = CH == 16 ? 1 : C / 32;
    constexpr int MI1 = TSP ? 1 : MI;                                           // conv1 row blocks per pass
    constexpr int NCHT = TSP ? NCH / 2 : NCH;                                   // planes of the t1 tile
    constexpr int RB = 2 * CH;                                                  // bytes of a tile row (one position of one plane)
    constexpr int KS = CH / 16;                                                 // k-steps per (plane, tap)
    constexpr int NG = CH / 8;                                                  // accumulator register quads that hold real channels
    constexpr int NQC = CH / 4;                                                 // channel quads of a plane
    constexpr int CB = 32 * RB;                                                 // bytes between the column blocks of a wave
    constexpr int XRMAX = W + 2 * 32 + 4;                                       // rows of the x tile at most (h1max <= 32)
    constexpr int NPF = (NCH * NQC * (XRMAX / 4) + NTH - 1) / NTH;              // staging items (4 channels x 4 positions) per thread
    // 16-byte slot swizzle of a row: 64-byte rows (4 slots) by (row >> 2) & 3, 32-byte rows (2 slots) by (row >> 3) & 1 - the 16 lanes of a
    // ds_read_b128 group then fall on 16 different slots of the 256-byte bank row at every tap offset
    auto swz = [](int row) { return CH == 32 ? ((row >> 2) & 3) : ((row >> 3) & 1); };

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];

    const int L = ws_uni(a.L), xrows = ws_uni(a.xrows), trows = ws_uni(a.trows), nk = ws_uni(a.nk);
    const int h1max = ws_uni(a.h1max), h2max = ws_uni(a.h2max), nto = ws_uni(a.nto);
    const int xpsz = xrows * RB, tpsz = trows * RB;                             // bytes per plane
    const unsigned xbase = 0, tbase = (unsigned)(NCH * xpsz);                   // LDS byte offsets of the two tiles
    float* const btab = reinterpret_cast<float*>(smem_w + tbase + NCHT * tpsz); // bias1[nk][C], then sum_j bias2_j [C], then a[C], s[C]
    float* const b2tab = btab + V2W_WS_MAXB * C;
    float* const atab = b2tab + C;
    const unsigned wring = tbase + (unsigned)(NCHT * tpsz) + (unsigned)((V2W_WS_MAXB + 3) * C * sizeof(float));     // WLDS: the fragment ring
    const float slope = a.slope, inv_slope = a.inv_slope;

    const int hout = ws_uni(a.hout);
    // one tile per workgroup
    int b = 0, n0 = 0, pos0 = 0;                                                // batch item, position of the first valid output column, position of x row 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = ws_uni(tid >> 6);
    int lr = lane & 31, hk = lane >> 5;                        // (not const: made opaque again at every tile, see the tile loop)
    const int wm0 = (wave / WN) * (32 * MI);
    const int wn0 = (wave % WN) * (32 * NI);
    const int xc0 = ws_uni(a.xoff) + h1max;                                     // x row of window column 0 (position n0 - h2max)
    auto tile_origin = [&](int tile, int& tb, int& tn0, int& tpos0) {
        if constexpr (!STD) { if (a.rb1) tile %= a.ntiles1; }
        tb = tile / a.ntl;
        tn0 = (tile % a.ntl) * (nto - 2 * hout) - hout;
        tpos0 = tn0 - h2max - h1max - ws_uni(a.xoff);                           // (a multiple of 4)
    };
    V2W_STAMP(0);

    // ---- stage lrelu(a * in + s) as bf16, every channel of the window + halo in ONE burst of loads.  An item = 4 channels x 4 positions:
    // four 8-byte loads (4 positions of one channel), four 8-byte LDS stores (the 4 channels of one position).  Loads (issue_x) and the
    // LDS writes (commit_x) are apart: the next tile's loads fly while the current tile is stored.
    const int nq = xrows >> 2;
    const int nitems = NCH * NQC * nq;
    u32x2 pf[NPF][4];
    auto issue_x = [&](int tile) {
        int tb, tn0, tpos0;
        tile_origin(tile, tb, tn0, tpos0);
        const unsigned short* src = a.in;
        if constexpr (!STD) { if (a.rb1) src = a.in_b[tile / a.ntiles1]; }
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(src) + (size_t)tb * C * L * 2;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH;
            const int cq = idx % NQC, rest = idx / NQC;
            const int pq = rest % nq, ch = rest / nq;
            const int pos = tpos0 + pq * 4;
            const bool ok = idx < nitems && pos >= 0 && pos < L;
            unsigned vo = (unsigned)((32 * (idx < nitems ? ch : 0) + 4 * cq) * L + (ok ? pos : 0)) * 2u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[s][i] = *gptr<const u32x2>(inb + (size_t)i * L * 2 + vo);
        }
    };
    auto commit_x = [&]() {                                  // (b, pos0: the tile the loads were issued for; the affine table of b is complete)
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH;
            if (idx >= nitems) continue;
            const int cq = idx % NQC, rest = idx / NQC;
            const int pq = rest % nq, ch = rest / nq;
            const int pos = pos0 + pq * 4;
            const bool ok = pos >= 0 && pos < L;           // L % 4 == 0 and pos % 4 == 0: a quad is inside or outside as a whole
            const f32x4 av = *reinterpret_cast<const f32x4*>(atab + 32 * ch + 4 * cq);
            const f32x4 sv = *reinterpret_cast<const f32x4*>(atab + C + 32 * ch + 4 * cq);
            const int row = pq * 4;
            unsigned char* dst = smem_w + xbase + ch * xpsz + row * RB + ((((cq >> 1) ^ swz(row)) << 4) | ((cq & 1) << 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (e & 1) ? ws_hi(pf[s][i][e >> 1]) : ws_lo(pf[s][i][e >> 1]);
                    v[i] = v2w_lrelu(fmaf(av[i], xv, sv[i]), slope);
                }
                u32x2 w = {ws_pack2(v[0], v[1]), ws_pack2(v[2], v[3])};
                if (!ok) w = u32x2{0u, 0u};                 // the padding of the ACTIVATED signal is exactly 0
                *reinterpret_cast<u32x2*>(dst + e * RB) = w;
            }
        }
    };

    const bool young = wave >= (WM * WN) / 2;                 // (uniform) the second-dispatched half of the workgroup's waves
    acc_t acc1[MI1][NI], oacc[MI][NI];
    unsigned lane16 = (unsigned)lane * 16u;
    auto mfma = [&](acc_t c, u32x4 av, u32x4 bv) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, av), __builtin_bit_cast(b8, bv), c, 0, 0, 0);
    };
    // this lane's 16 bytes of k-step 0 at (plane ch, row): slot hk, swizzled by the row; with 64-byte rows k-step 1 is the address ^ 32
    auto baddr = [&](unsigned base, int psz, int ch, int row) {
        return base + (unsigned)(ch * psz + row * RB + ((hk ^ swz(row)) << 4));
    };

    // ---- the MFMA loop of one conv over a resident tile (base, plane size psz): the NCH * K taps, a ring of V2W_WS_RING taps of weight
    // fragments (KS per row block and tap) refilled as their k-steps retire.  B operands: two sets, k-step q uses set q & 1 and refills it
    // with k-step q + 2 (64-byte rows: the same k-step of the next tap; 32-byte rows: the tap after the next).  r0 = tile row of (column lr
    // of this wave's block 0, tap 0).
    auto conv = [&](acc_t (&acc)[MI][NI], unsigned base, int psz, int r0, const unsigned char* wps, int K, int dil) {
        const int nst = KS * NCH * K;
        // The t1 tile has no halo rows of its own: a tap that reaches past it reads whatever lies h2max rows before / after the plane (the
        // neighbouring plane, the x tile, the tables, the slack the launcher allocates behind them).  An output column depends on the operand
        // rows of that column alone, so the garbage (NaN patterns included) stays in window columns that are never stored.
        auto rowaddr = [&](int ch, int row) { return baddr(base, psz, ch, row); };
        if constexpr (WLDS) {
            // ---- fragments through the LDS ring.  A tap = FT = (C / 32) * KS fragments of 1 KiB, fragment f = (row block f / KS, k-step f % KS);
            // wave w copies fragments w * FPW .. of every tap (one global_load_lds_dwordx4 each: 64 lanes x 16 bytes, the LDS address
            // in M0).  Three slots: at the top of tap g every wave has waited for its part of tap g + 1 (counted vmcnt: the copies
            // retire in order), the barrier makes the whole tap visible and frees the slot of tap g (its fragments sit in registers since
            // tap g - 1) for the copy of tap g + 3; tap g + 1's fragments are read into the other register set under tap g's MFMAs.
            static_assert(KS == 2, "ring of 64-byte-row tiles");
            constexpr int NW = WM * WN, FT = (C / 32) * KS, FPW = (FT + NW - 1) / NW, SLOT = FT * 1024;
            static_assert(FT % NW == 0 || FT < NW, "whole fragments per wave");
            const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_w) + wring;
            const unsigned lane16w = lane16;
            auto dma = [&](int slot, int ch, int t) {            // this wave's fragments of (plane ch, tap t) -> ring slot (uniform arguments)
                const int chc = ch < NCH ? ch : NCH - 1;
#pragma unroll
                for (int u = 0; u < FPW; ++u) {
                    const int f = wave * FPW + u;
                    if (FT < NW && f >= FT) break;               // (more waves than fragments: the first FT waves copy)
                    const int rb = f / KS, sq = f % KS;
                    const unsigned char* src = wps + ((size_t)rb * nst + (size_t)(KS * chc + sq) * K + t) * V2W_WS_UNIT + lane16w;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(slot * SLOT + f * 1024));
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
            };
            constexpr int MYF = FT < NW ? 1 : FPW;               // copies a wave has in flight per tap (waves beyond FT: none - they only wait less)
            int qc = 0, qt = 0;
            auto next_q = [&]() { if (++qt >= K) { qt = 0; ++qc; } };
            __syncthreads();                                     // every wave is done with the ring (the previous conv's last taps)
            dma(0, qc, qt); next_q();
            dma(1, qc, qt); next_q();
            dma(2, qc, qt); next_q();
            u32x4 areg[2][KS][MI];
            auto read_a = [&](u32x4 (&av)[KS][MI], int slot) {
#pragma unroll
                for (int sq = 0; sq < KS; ++sq)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        av[sq][i] = *reinterpret_cast<const u32x4*>(smem_w + wring + slot * SLOT + ((wm0 / 32 + i) * KS + sq) * 1024 + lane16);
            };
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * MYF) : "memory");
            __builtin_amdgcn_s_barrier();
            read_a(areg[0], 0);
            int lc = 0, lt = 0;
            auto advance = [&]() {
                int nc = lc, nt = lt + 1;
                if (nt >= K) { nt = 0; ++nc; }
                if (nc < NCH) { lc = nc; lt = nt; }
            };
            u32x4 bb[2][NI];
            {
                const unsigned x0 = rowaddr(0, r0);
                advance();
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    bb[0][j] = *reinterpret_cast<const u32x4*>(smem_w + x0 + j * CB);
                    bb[1][j] = *reinterpret_cast<const u32x4*>(smem_w + (x0 ^ 32u) + j * CB);
                }
            }
            int slot_n = 1, slot_q = 0;                          // slot of tap g + 1 (read next), slot the next copy goes to (= tap g's)
            auto tapw = [&](auto par_c) {
                constexpr int P = decltype(par_c)::value;
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" :: "n"(MYF) : "memory");      // tap g + 1 has landed; my reads of tap g are back
                __builtin_amdgcn_s_barrier();
                dma(slot_q, qc, qt); next_q();
                read_a(areg[P ^ 1], slot_n);
                slot_q = slot_q == 2 ? 0 : slot_q + 1;
                slot_n = slot_n == 2 ? 0 : slot_n + 1;
                const unsigned xn = rowaddr(lc, r0 + lt * dil);
                advance();
#pragma unroll
                for (int sq = 0; sq < KS; ++sq) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) acc[i][j] = mfma(acc[i][j], areg[P][sq][i], bb[sq][j]);
                        bb[sq][j] = *reinterpret_cast<const u32x4*>(smem_w + (sq ? (xn ^ 32u) : xn) + j * CB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            const int TT = NCH * K;
            int g = 0;
            for (; g + 2 <= TT; g += 2) { tapw(std::integral_constant<int, 0>{}); tapw(std::integral_constant<int, 1>{}); }
            if (g < TT) tapw(std::integral_constant<int, 0>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the copies past the last tap (clamped re-reads) have landed before the ring is reused
            return;
        }
        const unsigned char* ap[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) ap[i] = wps + (size_t)(wm0 / 32 + i) * nst * V2W_WS_UNIT;
        // (Measured at C = 128: 3 or 7 k-steps of lookahead time the same - the loop is not bound by the fragments' latency.)
        constexpr int RT = V2W_WS_RING;
        u32x4 ar[KS * RT][MI];
        auto load_frag = [&](u32x4 (&av)[MI], int ch, int s, int t) {     // k-step s of (plane ch, tap t); clamped past the end
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            const int chc = ch < NCH ? ch : NCH - 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                av[i] = *gptr<const u32x4>(ap[i] + (size_t)((KS * chc + s) * K + t) * V2W_WS_UNIT + l16);
        };
        int qc = 0, qt = 0;                                  // the tap whose fragments are requested next
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) load_frag(ar[KS * r + s2], qc, s2, qt);
            if (++qt >= K) { qt = 0; ++qc; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (lc, lt): the tap whose B operands are read next - one tap ahead of the running one (two with one k-step per tap)
        int lc = 0, lt = 0;
        auto advance = [&]() {
            int nc = lc, nt = lt + 1;
            if (nt >= K) { nt = 0; ++nc; }
            if (nc < NCH) { lc = nc; lt = nt; }              // (past the end: the last tap again - read, never used)
        };
        u32x4 bb[2][NI];
        {
            const unsigned x0 = rowaddr(0, r0);
            advance();
            const unsigned x1 = KS == 2 ? (x0 ^ 32u) : rowaddr(lc, r0 + lt * dil);
            if constexpr (KS == 1) advance();
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bb[0][j] = *reinterpret_cast<const u32x4*>(smem_w + x0 + j * CB);
                bb[1][j] = *reinterpret_cast<const u32x4*>(smem_w + x1 + j * CB);
            }
        }
        auto kstep = [&](auto bs_c, const u32x4 (&av)[MI], unsigned nxt) {
            constexpr int bs = decltype(bs_c)::value;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mfma(acc[i][j], av[i], bb[bs][j]);
                bb[bs][j] = *reinterpret_cast<const u32x4*>(smem_w + nxt + j * CB);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto tap = [&](auto par_c) {
            constexpr int P = decltype(par_c)::value, S0 = KS * P;
            const unsigned xn = rowaddr(lc, r0 + lt * dil);
            advance();
            if constexpr (KS == 2) {
                kstep(std::integral_constant<int, 0>{}, ar[S0], xn);
                load_frag(ar[S0], qc, 0, qt);
                __builtin_amdgcn_sched_barrier(0);
                kstep(std::integral_constant<int, 1>{}, ar[S0 + 1], xn ^ 32u);
                load_frag(ar[S0 + 1], qc, 1, qt);
            } else {
                kstep(std::integral_constant<int, P & 1>{}, ar[S0], xn);
                load_frag(ar[S0], qc, 0, qt);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (++qt >= K) { qt = 0; ++qc; }
        };
        const int TT = NCH * K;
        int g = 0;
        for (; g + RT <= TT; g += RT) {
            tap(std::integral_constant<int, 0>{});
            if constexpr (RT > 1) tap(std::integral_constant<int, 1>{});
            if constexpr (RT > 2) tap(std::integral_constant<int, 2>{});
            if constexpr (RT > 3) tap(std::integral_constant<int, 3>{});
        }
        if (g < TT) tap(std::integral_constant<int, 0>{});
        if constexpr (RT > 2) { if (g + 1 < TT) tap(std::integral_constant<int, 1>{}); }
        if constexpr (RT > 3) { if (g + 2 < TT) tap(std::integral_constant<int, 2>{}); }
    };

    // ---- the same loop with the tap count and dilation as COMPILE-TIME constants (the generator's residual blocks: 3 / 7 / 11 taps, dilation
    // 1 and 3).  Measured on the runtime form: a tap of 8 MFMAs (256 cycles of issue) costs 870 cycles with two waves per SIMD and 725 with
    // every load compiled out - the scalar bookkeeping of a tap (which plane, which tap, wrap-arounds, fragment index multiplies: ~40
    // dependent SALU instructions) stands between the MFMAs of a wave, and two waves that run the same code hit those stretches together.
    // Unrolled over the taps of a plane, every offset is an immediate or one add.
    // (A conv's loop starts cold - its first four fragments come from L2, ~1.5 k cycles in which the wave issues nothing; requesting them ahead of
    // the barrier / epilogue in front of the conv was measured and changed nothing: the other wave of the SIMD covers the cold start.)
    // (acc: [MIX][NI] blocks, row block i of this wave = block rb0 + i of the weight stream: MIX = MI, rb0 = wm0 / 32 for the stage's own
    // convs; the fused upsampler runs UPF / 2 times as many row blocks per wave)
    // nchp_c planes of the tile are walked; plane ch of the tile is reduction plane PMUL ch + padd of the weights (all of them: PMUL = 1,
    // padd = 0; the half t1 tile of a TSP pass: PMUL = 2, padd = the pass)
    auto conv_ct = [&](auto k_c, auto d_c, auto& acc, int rb0, unsigned base, int psz, int r0, const unsigned char* wps, auto nchp_c, int padd)
        __attribute__((always_inline)) {
        constexpr int K = decltype(k_c)::value, DIL = decltype(d_c)::value;
        constexpr int NCHP = decltype(nchp_c)::value, PMUL = NCH / NCHP;
        constexpr int MIX = (int)std::extent<std::remove_reference_t<decltype(acc)>, 0>::value;
        static_assert(KS == 2 && K >= 2, "64-byte rows");
        const unsigned char* ap[MIX];
#pragma unroll
        for (int i = 0; i < MIX; ++i) ap[i] = wps + (size_t)(rb0 + i) * (KS * NCH * K) * V2W_WS_UNIT;
        u32x4 ar[4][MIX];                                     // fragments of two taps: slots 2 (g & 1) + s for global tap g
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));
        auto frag = [&](u32x4 (&av)[MIX], int foff) {         // foff: (uniform) byte offset of the fragment inside a row block's stream
#pragma unroll
            for (int i = 0; i < MIX; ++i) av[i] = *gptr<const u32x4>(ap[i] + foff + l16);
        };
        auto addr = [&](unsigned pbase, int row) { return pbase + (unsigned)(row * RB + ((hk ^ swz(row)) << 4)); };
        // fragment (plane ch, k-step s, tap t) of a row block sits at ((2 ch + s) K + t) units
        const int f0 = 2 * padd * K * V2W_WS_UNIT;
        frag(ar[0], f0);
        frag(ar[1], f0 + K * V2W_WS_UNIT);
        frag(ar[2], f0 + 1 * V2W_WS_UNIT);
        frag(ar[3], f0 + (K + 1) * V2W_WS_UNIT);
        u32x4 bb[2][NI];
        {
            const unsigned x0 = addr(base, r0);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bb[0][j] = *reinterpret_cast<const u32x4*>(smem_w + x0 + j * CB);
                bb[1][j] = *reinterpret_cast<const u32x4*>(smem_w + (x0 ^ 32u) + j * CB);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto plane = [&](auto par_c, int ch) {
            constexpr int PAR = decltype(par_c)::value;       // parity of the plane's first tap in the global tap order (K is odd)
            const unsigned pbase = base + (unsigned)(ch * psz);
            const bool lastp = ch + 1 >= NCHP;
            const int fbase = 2 * (PMUL * ch + padd) * K * V2W_WS_UNIT;       // (uniform) the plane's k-step 0, tap 0
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int sl = 2 * ((PAR + t) & 1);
                // B operands of the next tap: this plane one dilation step on, or tap 0 of the next plane (past the end: this tap again)
                const bool wrapn = t + 1 >= K;
                const unsigned xn = wrapn ? (lastp ? addr(pbase, r0 + t * DIL) : addr(pbase + (unsigned)psz, r0)) : addr(pbase, r0 + (t + 1) * DIL);
                // fragments two taps on: this plane, or the next one (its blocks lie 2 K units further; past the end: this plane's again)
                const int t2 = (t + 2) % K;
                const int f2 = fbase + ((t + 2 >= K && !lastp) ? 2 * PMUL * K * V2W_WS_UNIT : 0) + t2 * V2W_WS_UNIT;
#pragma unroll
                for (int sq = 0; sq < 2; ++sq) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
#pragma unroll
                        for (int i = 0; i < MIX; ++i) acc[i][j] = mfma(acc[i][j], ar[sl + sq][i], bb[sq][j]);
                        bb[sq][j] = *reinterpret_cast<const u32x4*>(smem_w + (sq ? (xn ^ 32u) : xn) + j * CB);
                    }
                    frag(ar[sl + sq], f2 + sq * K * V2W_WS_UNIT);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if constexpr (NCHP == 1) {
            plane(std::integral_constant<int, 0>{}, 0);
        } else {
#pragma nounroll
            for (int ch = 0; ch < NCHP; ch += 2) {
                plane(std::integral_constant<int, 0>{}, ch);
                plane(std::integral_constant<int, 1>{}, ch + 1);
            }
        }
    };
    constexpr std::integral_constant<int, NCH> all_planes{};
    // ---- the running output accumulator starts at sum_j b2_j
    auto init_oacc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};              // (registers of the padding rows stay 0)
                if (g < NG) bv = *reinterpret_cast<const f32x4*>(b2tab + wm0 + 32 * i + 8 * g + 4 * hk);
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int j = 0; j < NI; ++j) oacc[i][j][4 * g + x] = bv[x];
            }
    };

    bool rb1 = false;
    if constexpr (!STD) rb1 = ws_uni(a.rb1) != 0;
    auto branch = [&](int jb, auto k_c) __attribute__((always_inline)) {
        constexpr int KC = decltype(k_c)::value;             // > 0: the tap count at compile time (dilations 1 and 3), 0: run-time arguments
        const int K = KC ? KC : ws_uni(a.K[jb]), d1 = KC ? 1 : ws_uni(a.d1[jb]), d2 = KC ? 3 : ws_uni(a.d2[jb]);
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        // ---- conv1_j on the window: column col <-> position n0 - h2max + col <-> x row xc0 + col.  (ab: one row block of conv1's accumulators,
        // rowblk: its index among the C / 32 row blocks)
        auto init_acc1 = [&](acc_t (&ab)[NI], int rowblk) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (g < NG) bv = *reinterpret_cast<const f32x4*>(btab + jb * C + 32 * rowblk + 8 * g + 4 * hk);
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int j = 0; j < NI; ++j) ab[j][4 * g + x] = bv[x];
            }
        };
        // ---- t1 = acc + x (x rebuilt from the activated tile: registers 4g .. 4g+3 of block j <-> channels 8g + 4hk + {0..3} of plane xplane
        // at this lane's position = 8 contiguous bytes); the running output block ob takes t1 in fp32, plane tplane of the t1 tile lrelu(t1)
        // as bf16
        auto t1_epi = [&](acc_t (&ab)[NI], acc_t (&ob)[NI], int xplane, int tplane) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int col = wn0 + lr;
                asm volatile("" : "+v"(col));               // (recomputed per branch: hoisted out of the loop these addresses spill)
                col += 32 * j;
                const int pos = n0 - h2max + col;
                const bool in_seq = pos >= 0 && pos < L;    // conv2 zero-pads t1 outside the sequence
                const int xrow = xc0 + col, trow = col;
                const unsigned xq = xbase + (unsigned)(xplane * xpsz + xrow * RB + 8 * hk);
                const unsigned tq = tbase + (unsigned)(tplane * tpsz + trow * RB + 8 * hk);
                const int xsw = swz(xrow), tsw = swz(trow);
                // (two elements per vector instruction where the ISA has a packed form - multiply, add; the position mask is a factor 0 / 1:
                // every value here is finite, x is exactly 0 outside the sequence.  6 vector instructions per element instead of 11.)
                const f32x2 msk = {in_seq ? 1.f : 0.f, in_seq ? 1.f : 0.f};
                const f32x2 isl2 = {inv_slope, inv_slope}, sl2 = {slope, slope};
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const u32x2 w = *reinterpret_cast<const u32x2*>(smem_w + xq + ((g ^ xsw) << 4));
                    u32x2 packed;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 xa = {ws_lo(w[h]), ws_hi(w[h])};
                        const f32x2 xi = xa * isl2;
                        const f32x2 xr = {ws_min(xa[0], xi[0]), ws_min(xa[1], xi[1])};       // lrelu undone (slope < 1)
                        f32x2 t = f32x2{ab[j][4 * g + 2 * h], ab[j][4 * g + 2 * h + 1]};
                        if (!rb1) t = t + xr;                // ResBlock2: t1 = x + conv1(..); ResBlock1: the intermediate carries no residual
                        t = t * msk;
                        const f32x2 radd = rb1 ? xr : t;     // ... its residual x (0 outside the sequence) joins the OUTPUT instead
                        ob[j][4 * g + 2 * h] += radd[0];
                        ob[j][4 * g + 2 * h + 1] += radd[1];
                        const f32x2 ts = t * sl2;
                        packed[h] = ws_pack2(fmaxf(t[0], ts[0]), fmaxf(t[1], ts[1]));
                    }
                    *reinterpret_cast<u32x2*>(smem_w + tq + ((g ^ tsw) << 4)) = packed;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (TSP) {
            // ---- two passes: conv1 rows of row block 2 wm + h -> the t1 half tile (plane wm of it) -> that half of conv2's reduction
            auto pass = [&](auto h_c) __attribute__((always_inline)) {
                constexpr int H = decltype(h_c)::value;
                init_acc1(acc1[0], wm0 / 32 + H);
                conv_ct(k_c, std::integral_constant<int, 1>{}, acc1, wm0 / 32 + H, xbase, xpsz, xc0 - h1 + wn0 + lr, ws_uni(a.w1[jb]), all_planes, 0);
                __syncthreads();          // the previous pass's conv2 has finished reading the t1 half tile
                t1_epi(acc1[0], oacc[H], wm0 / 32 + H, wave / WN);
                __syncthreads();
                conv_ct(k_c, std::integral_constant<int, 3>{}, oacc, wm0 / 32, tbase, tpsz, wn0 + lr - h2, ws_uni(a.w2[jb]),
                        std::integral_constant<int, NCHT>{}, H);
            };
            pass(std::integral_constant<int, 0>{});
            pass(std::integral_constant<int, 1>{});
        } else {
#pragma unroll
        for (int i = 0; i < MI1; ++i) init_acc1(acc1[i], wm0 / 32 + i);
        V2W_STAMP(3 + 6 * jb);
        if constexpr (KC > 0) conv_ct(k_c, std::integral_constant<int, 1>{}, acc1, wm0 / 32, xbase, xpsz, xc0 - h1 + wn0 + lr, ws_uni(a.w1[jb]), all_planes, 0);
        else conv(acc1, xbase, xpsz, xc0 - h1 + wn0 + lr, ws_uni(a.w1[jb]), K, d1);
        V2W_STAMP(4 + 6 * jb);
        __syncthreads();          // conv2 of the previous branch has finished reading the t1 tile
        V2W_STAMP(5 + 6 * jb);
#pragma unroll
        for (int i = 0; i < MI1; ++i) t1_epi(acc1[i], oacc[i], wm0 / 32 + i, wm0 / 32 + i);
        V2W_STAMP(6 + 6 * jb);
        __syncthreads();
        V2W_STAMP(7 + 6 * jb);
        // ---- conv2_j on the same window, onto the running accumulator
        if constexpr (KC > 0) conv_ct(k_c, std::integral_constant<int, 3>{}, oacc, wm0 / 32, tbase, tpsz, wn0 + lr - h2, ws_uni(a.w2[jb]), all_planes, 0);
        else conv(oacc, tbase, tpsz, wn0 + lr - h2, ws_uni(a.w2[jb]), K, d2);
        }
        V2W_STAMP(8 + 6 * jb);
    };
    issue_x(blockIdx.x);
    {
    const int tile = blockIdx.x;
    tile_origin(tile, b, n0, pos0);
    // every per-lane address term of the unrolled convs derives from these three: opaque per tile, or hipcc hoists ~100 registers of
    // loop-invariant offsets out of the tile loop and spills the accumulators
    asm volatile("" : "+v"(lr), "+v"(hk), "+v"(lane16));
    __syncthreads();              // the previous tile's stores have read the scratch (which overlays the tiles and tables)
    for (int i = tid; i < nk * C; i += NTH) {
        const int j = i / C, c = i - j * C;
        btab[i] = a.bias1[j] ? a.bias1[j][c] : 0.f;
    }
    const int prob = rb1 ? tile / ws_uni(a.ntiles1) : 0;          // ResBlock1 pair mode: this tile's problem = its one branch
    for (int c = tid; c < C; c += NTH) {
        float v = 0.f;
        for (int j = rb1 ? prob : 0; j < (rb1 ? prob + 1 : nk); ++j) v += a.bias2[j] ? a.bias2[j][c] : 0.f;
        b2tab[c] = v;
        atab[c] = a.in_a ? a.in_a[b * C + c] : 1.f;
        atab[C + c] = a.in_a ? a.in_s[b * C + c] : 0.f;
    }
    __syncthreads();
    commit_x();
    V2W_STAMP(1);
    __syncthreads();
    V2W_STAMP(2);
    init_oacc();
    if constexpr (STD) {
        branch(0, std::integral_constant<int, 3>{});
        branch(1, std::integral_constant<int, 7>{});
        branch(2, std::integral_constant<int, 11>{});
    } else {
        if (rb1) branch(prob, std::integral_constant<int, 0>{});
        else for (int jb = 0; jb < nk; ++jb) branch(jb, std::integral_constant<int, 0>{});
    }

    if constexpr (UPF > 0) {
        // ---- the NEXT stage's upsampler on this tile, while the stage's output is on chip (models.py:128-129: leaky_relu -> ConvTranspose1d
        // (2 UPF taps, stride UPF) + bias): z = lrelu(out / nk) goes into the dead t1 tile as bf16 rows - the transposed conv's operand, written
        // like a t1 epilogue, no transposition - and the polyphase 3-tap conv over the UPF * C / 2 virtual rows (v2w_convt_bf16_res.hip) runs on
        // it: the phases of a channel are adjacent accumulator registers of one lane, so the output leaves straight from the accumulators with
        // the BatchNorm partial sums of the fp32 values.  The stage's own output never exists in memory: one tensor written and one tensor
        // read less per stage, one launch less, no staging of the upsampler's input.
        constexpr int MIU = MI * UPF / 2;                     // 32-row blocks of virtual rows per wave
        constexpr int UH = TSP ? 2 : 1, MIUH = MIU / UH;      // ... run in UH passes of MIUH blocks (TSP: 192 accumulator registers would not fit)
        constexpr int CPB = 32 / UPF, CU = C / 2;             // channels per 32-row block; real output channels
        __syncthreads();                                      // conv2 of the last branch has read the t1 tile
        V2W_STAMP(27);
        // the operand tile z: the dead t1 tile - or (TSP: the t1 tile holds half the planes) the dead x tile, a slack of 256 bytes in front of
        // it for tap -1 of plane 0; the scratch (bias [C / 2], partial sums [WN][C / 2][2], the waves' strips): the other dead tile
        const unsigned zbase = TSP ? xbase + 256u : tbase;
        float* const ubias = reinterpret_cast<float*>(smem_w + (TSP ? tbase : xbase));
        float* const ured = ubias + CU;
        for (int c = tid; c < CU; c += NTH) ubias[c] = a.up_bias ? a.up_bias[c] : 0.f;
        {
            const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f;
            const float uslope = a.up_slope;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int col = wn0 + lr;
                    asm volatile("" : "+v"(col));
                    col += 32 * j;
                    const int pos = n0 - h2max + col;
                    const bool in_seq = pos >= 0 && pos < L;  // the transposed conv sees the L positions of the sequence only
                    const unsigned tq = zbase + (unsigned)((wm0 / 32 + i) * tpsz + col * RB + 8 * hk);
                    const int tsw = swz(col);
                    // (packed forms; the mask as a factor: the out-of-sequence columns a stored output can read - position -1, position L -
                    // lie inside the tile's valid window and are finite)
                    const f32x2 msk = {in_seq ? 1.f : 0.f, in_seq ? 1.f : 0.f};
                    const f32x2 d2 = {a.out_div, a.out_div}, r2 = {dinv, dinv}, us2 = {uslope, uslope};
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        u32x2 packed;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x2 v = {oacc[i][j][4 * g + 2 * h], oacc[i][j][4 * g + 2 * h + 1]};
                            if (a.out_div != 0.f) {          // v2w_div_by, two lanes at a time: the correctly rounded quotient
                                const f32x2 q = v * r2;
                                v = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, d2, v), r2, q);
                            }
                            v = v * msk;
                            const f32x2 vs = v * us2;
                            packed[h] = ws_pack2(fmaxf(v[0], vs[0]), fmaxf(v[1], vs[1]));
                        }
                        *reinterpret_cast<u32x2*>(smem_w + tq + ((g ^ tsw) << 4)) = packed;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __syncthreads();
        V2W_STAMP(21);
        // Block (i, j), register quad g, lane (lr, hk): virtual rows 32 i + 8 g + 4 hk + {0..3} at input position q.
        //   UPF = 4: channel 8 i + 2 g + hk, outputs 4 q + {0..3};   UPF = 2: channels 16 i + 4 g + 2 hk + {0, 1}, outputs 2 q + {0, 1}
        // The accumulators start at the bias of their channel (no add in the epilogue).
        constexpr int NC = UPF == 2 ? 2 : 1;                  // channels of a register quad
        constexpr int CHW = MIUH * CPB;                       // channels of this wave per pass
        const int Lout = L * UPF;
        const bool stats = a.up_stats != nullptr;
        unsigned char* const obase = reinterpret_cast<unsigned char*>(a.up_out) + (size_t)b * CU * Lout * 2;
        const int ncen = nto - 2;                             // input positions this tile is the centre of: window columns h2max + 1 ..
        float* const ssc = ured + WN * CU * 2 + wave * (CHW * 64);       // this wave's scratch, reused by its passes
        unsigned qo[NI];                                      // byte offset of output position U q inside a channel row; 0xffffffff: not stored
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) {
            const int col = wn0 + 32 * jj + lr;
            const int cc = col - h2max - 1;                   // index among the tile's centres
            const int q = n0 + 1 + cc;
            const bool ok = cc >= 0 && cc < ncen && q < L;
            qo[jj] = ok ? (unsigned)(UPF * q) * 2u : 0xffffffffu;
        }
#pragma unroll 1
        for (int uh = 0; uh < UH; ++uh) {
        const int cw0 = ((wm0 / 32) * (UPF / 2) + MIUH * uh) * CPB;       // the first channel of this wave's pass
        acc_t uacc[MIUH][NI];
#pragma unroll
        for (int i = 0; i < MIUH; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = CPB * i + (UPF == 4 ? 2 * g + hk : 4 * g + 2 * hk);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float bv = ubias[cw0 + cl + (UPF == 2 ? (x >> 1) : 0)];
#pragma unroll
                    for (int j = 0; j < NI; ++j) uacc[i][j][4 * g + x] = bv;
                }
            }
        // virtual tap tv reads input position q + tv - 1 (2 U taps at stride U: one input position of halo per side)
        conv_ct(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, uacc, (wm0 / 32) * (UPF / 2) + MIUH * uh, zbase, tpsz,
                wn0 + lr - 1, ws_uni(a.up_w), all_planes, 0);
        V2W_STAMP(22);
        // ---- epilogue: bf16 stores straight from the accumulators (the phases of a channel are adjacent registers of a lane, consecutive
        // lanes = consecutive output positions).  BatchNorm partial sums of the fp32 values: a lane adds up its columns of a channel, the
        // 32 lanes of a channel meet through a wave-private LDS scratch [channel][32][2] that ONE pass of the wave sums per (channel, stat)
        // in a fixed order - instead of a DPP tree + readlanes per register quad (30 of the ~100 vector instructions of a quad).
#pragma unroll
        for (int i = 0; i < MIUH; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = CPB * i + (UPF == 4 ? 2 * g + hk : 4 * g + 2 * hk);       // channel inside the wave's pass (+ 1: the second of UPF = 2)
                float s1[NC], s2[NC];
#pragma unroll
                for (int n = 0; n < NC; ++n) s1[n] = s2[n] = 0.f;
                const unsigned crow = (unsigned)((cw0 + cl) * Lout) * 2u;
#pragma unroll
                for (int jj = 0; jj < NI; ++jj) {
                    float v[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        v[x] = uacc[i][jj][4 * g + x];
                        // (columns outside the tile's centres / the sequence count 0 - a select, not a product: they may hold NaN patterns)
                        const float vm = qo[jj] != 0xffffffffu ? v[x] : 0.f;
                        s1[UPF == 2 ? (x >> 1) : 0] += vm;
                        s2[UPF == 2 ? (x >> 1) : 0] = fmaf(vm, vm, s2[UPF == 2 ? (x >> 1) : 0]);
                    }
                    if (qo[jj] != 0xffffffffu) {
                        if constexpr (UPF == 2) {
                            *gptr<unsigned>(obase + crow + qo[jj]) = ws_pack2(v[0], v[1]);
                            *gptr<unsigned>(obase + crow + (unsigned)Lout * 2u + qo[jj]) = ws_pack2(v[2], v[3]);
                        } else {
                            *gptr<u32x2>(obase + crow + qo[jj]) = u32x2{ws_pack2(v[0], v[1]), ws_pack2(v[2], v[3])};
                        }
                    }
                }
                if (stats) {
#pragma unroll
                    for (int n = 0; n < NC; ++n) *reinterpret_cast<f32x2*>(ssc + ((cl + n) * 32 + lr) * 2) = f32x2{s1[n], s2[n]};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (stats) {
            // lane = (channel, stat) of this wave's CHW channels: the 32 column sums in an order that starts at the lane's own slot (LDS banks)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (wave-private scratch: the wave's own writes have landed)
            if (lane < 2 * CHW) {
                const int chl = lane >> 1, st = lane & 1;
                float t = 0.f;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) t += ssc[(chl * 32 + ((k + lane) & 31)) * 2 + st];
                ured[((wave % WN) * CU + cw0 + chl) * 2 + st] = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the strip is read before the next pass rewrites it)
        }
        }     // (uh)
        if (stats) {
            __syncthreads();
            for (int c = tid; c < CU; c += NTH) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) { t1 += ured[(w * CU + c) * 2]; t2 += ured[(w * CU + c) * 2 + 1]; }
                gptr<float>(a.up_stats)[((size_t)tile * CU + c) * 2 + 0] = t1;
                gptr<float>(a.up_stats)[((size_t)tile * CU + c) * 2 + 1] = t2;
            }
        }
        V2W_STAMP(23);
    } else {
    // ---- the nto valid columns (window columns h2max .. h2max + nto) through an fp32 scratch [C][W + 8] in the dead tiles, shifted so that
    // output position quads are 16-byte aligned (all waves: quads cross the waves' columns), then either the stage's output - 8-byte bf16
    // stores along positions - or, fused, the generator's tail on it
    __syncthreads();
    V2W_STAMP(27);
    {
        constexpr int SRS = W + 8;
        float* const scr = reinterpret_cast<float*>(smem_w);
        const int soff = (h2max + 3) & ~3;
        const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f;
        const bool tail = a.post_out != nullptr;
        const float pslope = a.post_slope;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn0 + j * 32 + lr;
                const int sc = col - h2max + soff;
                const int pos = n0 - h2max + col;
                const bool in_seq = pos >= 0 && pos < L;
#pragma unroll
                for (int e = 0; e < 4 * NG; ++e) {
                    float v = oacc[i][j][e];
                    if (tail) {                              // conv_post reads leaky_relu(x / nk), zero-padded outside the sequence
                        if (a.out_div != 0.f) v = v2w_div_by(v, a.out_div, dinv);
                        v = in_seq ? v2w_lrelu(v, pslope) : 0.f;
                    }
                    scr[(wm0 + 32 * i + F::row(e, hk)) * SRS + sc] = v;
                }
            }
        __syncthreads();
        // the next tile's x: in flight under this tile's stores.  Unconditional (past the end: the last tile again, never committed) - under
        // a condition the old values stay live through the whole iteration as the other input of the join: 48 registers, 100 spills
        if (!tail) {
            const int nq = nto >> 2;
            const unsigned magic = (unsigned)(((1ull << 32) + nq - 1) / nq);
            unsigned short* dstp = a.out;
            if constexpr (!STD) { if (rb1) dstp = a.out_b[prob]; }
            unsigned char* const obase = reinterpret_cast<unsigned char*>(dstp) + (size_t)b * C * L * 2;
            const bool adds = !STD && rb1 && a.add0 != nullptr && prob == nk - 1;      // (the summing problem is the last one)
            for (int idx = tid; idx < C * nq; idx += NTH) {
                const int row = (int)__umulhi((unsigned)idx, magic), q = idx - row * nq;
                const int pos = n0 + 4 * q;
                if (pos >= L) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * SRS + soff + 4 * q);
                if (adds) {        // ((r0 + r1) + r2) / nk in the reference's order (models.py:135-141): the other branches' bf16 results first
                    const size_t eo = ((size_t)b * C * L + (size_t)row * L + pos) * 2;
                    const u32x2 w0 = *gptr<const u32x2>(reinterpret_cast<const unsigned char*>(a.add0) + eo);
                    f32x4 s4 = {ws_lo(w0[0]), ws_hi(w0[0]), ws_lo(w0[1]), ws_hi(w0[1])};
                    if (a.add1) {
                        const u32x2 w1 = *gptr<const u32x2>(reinterpret_cast<const unsigned char*>(a.add1) + eo);
                        s4 += f32x4{ws_lo(w1[0]), ws_hi(w1[0]), ws_lo(w1[1]), ws_hi(w1[1])};
                    }
                    v = s4 + v;
                }
                if (a.out_div != 0.f) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], a.out_div, dinv);
                }
                *gptr<u32x2>(obase + (unsigned)(row * L + pos) * 2u) = u32x2{ws_pack2(v[0], v[1]), ws_pack2(v[2], v[3])};
            }
        } else {
            // y[p] = tanh(b + sum_{t, c} w[t][c] * z[c][p + t - hout]) for the nto - 2 hout positions p = n0 + hout + m: scratch column of
            // z[c][p + t - hout] is soff + m + t.  A thread takes 4 consecutive outputs: per channel 3 aligned float4s of the scratch row.
            const int nty = nto - 2 * hout;
            const int PK = a.post_k;
            const float pb = a.post_b ? a.post_b[0] : 0.f;
            for (int qd = tid; qd * 4 < nty; qd += NTH) {
                const int m = 4 * qd;
                const int p0 = n0 + hout + m;
                if (p0 >= L) continue;
                f32x4 y = {pb, pb, pb, pb};
                for (int c = 0; c < C; ++c) {
                    const float* zr = scr + c * SRS + soff + m;
                    float z[12];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 4 * u);
                        z[4 * u] = zz[0]; z[4 * u + 1] = zz[1]; z[4 * u + 2] = zz[2]; z[4 * u + 3] = zz[3];
                    }
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        if (t < PK) {
                            const float wv = a.post_w[t * C + c];
#pragma unroll
                            for (int x = 0; x < 4; ++x) y[x] = fmaf(wv, z[t + x], y[x]);
                        }
                    }
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) y[x] = tanhf(y[x]);
                *gptr<f32x4>(a.post_out + (size_t)b * L + p0) = y;
            }
        }
    }
    V2W_STAMP(28);
    }     // (UPF == 0)
    }
}

template <int MI, int NI, int WM, int WN, int OCC = 2, int CH = 32, bool WLDS = false, int UPF = 0, bool TSP = false>
int launch_wide(const v2w_stage_split_args* q, hipStream_t stream, int* up_tiles_out = nullptr) {
    constexpr int NTH = 64 * WM * WN, C = CH == 16 ? 16 : 32 * MI * WM, W = 32 * NI * WN, NCH = CH == 16 ? 1 : C / 32, RB = 2 * CH;
    constexpr int NCHT = TSP ? NCH / 2 : NCH;                // planes of the t1 tile (TSP: half of them, see the kernel)
    WideArgs p{};
    p.in = reinterpret_cast<const unsigned short*>(q->in); p.in_a = q->in_a; p.in_s = q->in_s;
    p.out = reinterpret_cast<unsigned short*>(q->out);
    p.nk = q->nk; p.B = q->B; p.C = q->C; p.L = q->L; p.slope = q->slope; p.inv_slope = 1.f / q->slope; p.out_div = q->out_div;
    for (int j = 0; j < q->nk; ++j) {
        p.w1[j] = static_cast<const unsigned char*>(q->wps1[j]); p.bias1[j] = q->bias1[j];
        p.w2[j] = static_cast<const unsigned char*>(q->wps2[j]); p.bias2[j] = q->bias2[j];
        p.K[j] = q->k[j]; p.d1[j] = q->dil1[j]; p.d2[j] = q->dil2[j];
        const int h1 = q->dil1[j] * (q->k[j] - 1) / 2, h2 = q->dil2[j] * (q->k[j] - 1) / 2;
        if (h1 > p.h1max) p.h1max = h1;
        if (h2 > p.h2max) p.h2max = h2;
    }
    if (p.h1max > 32 || p.h2max > 32) return V2W_E_SHAPE;
    if (q->post_out) {                         // the fused tail: C = 16 (one plane), an odd tap count <= 9, fp32 (B, 1, L) output
        if (CH != 16 || !q->post_w || q->post_k < 1 || q->post_k > 9 || !(q->post_k & 1)) return V2W_E_SHAPE;
        if (reinterpret_cast<uintptr_t>(q->post_out) & 15) return V2W_E_SHAPE;
        p.post_w = q->post_w; p.post_b = q->post_b; p.post_out = q->post_out; p.post_k = q->post_k; p.post_slope = q->post_slope;
        p.hout = (q->post_k - 1) / 2;
    } else if (UPF > 0) {                      // the fused upsampler: 2 U taps at stride U = UPF (one input position of halo per side), C -> C / 2
        if (q->up_u != UPF || q->up_k != 2 * UPF || !q->up_wps || !q->up_out) return V2W_E_SHAPE;
        if ((reinterpret_cast<uintptr_t>(q->up_out) & 15) || (reinterpret_cast<uintptr_t>(q->up_wps) & 15)) return V2W_E_SHAPE;
        if (!(q->up_slope > 0.f && q->up_slope <= 1.f)) return V2W_E_SHAPE;
        if ((long long)(C / 2) * q->L * UPF * 2 >= (1ll << 31)) return V2W_E_SHAPE;      // 32-bit offsets inside one batch item
        p.up_w = static_cast<const unsigned char*>(q->up_wps); p.up_bias = q->up_bias; p.up_out = reinterpret_cast<unsigned short*>(q->up_out);
        p.up_stats = q->up_stats_part; p.up_slope = q->up_slope;
        p.hout = 1;
    } else if (q->rb1) {                       // ResBlock1 pair mode: one problem per branch, each with its own input / output tensor
        if (q->up_out || q->post_out || UPF > 0) return V2W_E_ARG;
        for (int j = 0; j < q->nk; ++j) {
            if (!q->out_b[j]) return V2W_E_ARG;
            const void* ib = q->in_b[j] ? q->in_b[j] : static_cast<const void*>(q->in);
            if ((reinterpret_cast<uintptr_t>(ib) & 15) || (reinterpret_cast<uintptr_t>(q->out_b[j]) & 15)) return V2W_E_SHAPE;
            p.in_b[j] = reinterpret_cast<const unsigned short*>(ib); p.out_b[j] = reinterpret_cast<unsigned short*>(q->out_b[j]);
        }
        if ((reinterpret_cast<uintptr_t>(q->add0) & 15) || (reinterpret_cast<uintptr_t>(q->add1) & 15) || (q->add1 && !q->add0)) return V2W_E_ARG;
        p.rb1 = 1; p.add0 = reinterpret_cast<const unsigned short*>(q->add0); p.add1 = reinterpret_cast<const unsigned short*>(q->add1);
    } else if (!q->out) return V2W_E_ARG;
    p.nto = (W - 2 * p.h2max) & ~3;
    if (p.hout) p.nto = ((W - 2 * p.h2max - 2 * p.hout) & ~3) + 2 * p.hout;     // the tile advances by nto - 2 hout: a multiple of 4
    if (p.nto - 2 * p.hout < W / 2) return V2W_E_SHAPE;
    const int hsum = p.h1max + p.h2max + p.hout;
    p.xoff = ((hsum + 3) & ~3) - hsum;
    p.xrows = (p.xoff + W + 2 * p.h1max + 3) & ~3;
    p.trows = W;                                      // (no halo rows: conv2's taps read up to h2max rows around the plane, see conv)
    p.ntl = (q->L + (p.nto - 2 * p.hout) - 1) / (p.nto - 2 * p.hout);
    p.ntiles = q->B * p.ntl;
    if (p.rb1) { p.ntiles1 = p.ntiles; p.ntiles = q->nk * p.ntiles1; }
    const size_t tiles = ((size_t)NCH * p.xrows + (size_t)NCHT * p.trows) * RB;
    // (+ 32 rows of slack: conv2's taps past the last plane of the t1 tile stay inside the allocation; TSP: the tables behind the tile - 7 KB
    // at C = 256 - are that slack, the budget has no room for another)
    const size_t lds = tiles + (size_t)(V2W_WS_MAXB + 3) * C * sizeof(float) + (WLDS ? (size_t)3 * (C / 32) * (CH / 16) * 1024 : 0) + (TSP ? 0 : 32 * RB);
    if (lds * ((OCC * 4) / (WM * WN)) > 160 * 1024) return V2W_E_SHAPE;       // (as many workgroups per CU as the configuration counts on)
    if (UPF == 0 && lds < (size_t)C * (W + 8) * sizeof(float)) return V2W_E_SHAPE;   // the store scratch [C][W + 8] overlays the tiles (and the dead tables / ring)
    if (TSP) {
        // z (all planes, W rows, 256 bytes of slack in front) over the x tile; the upsampler's scratch over the t1 half tile
        constexpr int CUc = C / 2, CHWc = (MI * UPF / 2 / 2) * (32 / (UPF ? UPF : 1));
        if ((size_t)NCH * W * RB + 256 + RB > (size_t)NCH * p.xrows * RB) return V2W_E_SHAPE;
        if ((size_t)(CUc + WN * CUc * 2 + WM * WN * CHWc * 64) * sizeof(float) > (size_t)NCHT * p.trows * RB) return V2W_E_SHAPE;
        if (p.h2max * RB > (int)((V2W_WS_MAXB + 3) * C * sizeof(float))) return V2W_E_SHAPE;
    }
    if (!(q->slope > 0.f && q->slope <= 1.f)) return V2W_E_SHAPE;            // lrelu as max(v, slope v), undone as min(a, a / slope)
    bool std_cfg = CH == 32 && !WLDS && q->nk == 3;          // the generator's own blocks: the compile-time form
    for (int j = 0; j < 3 && std_cfg; ++j) std_cfg = q->k[j] == 3 + 4 * j && q->dil1[j] == 1 && q->dil2[j] == 3;
    if (p.rb1) std_cfg = false;
    if (UPF > 0 && !std_cfg) return V2W_E_SHAPE;             // the fused upsampler exists for the compile-time block set only
    if (up_tiles_out) *up_tiles_out = p.ntiles;              // rows of up_stats_part
    constexpr bool STDK = CH == 32 && !WLDS;
    // (a TSP configuration exists in its compile-time form only: the run-time form of that tiling would need the full t1 tile)
    void (*kern)(const WideArgs) = nullptr;
    if constexpr (TSP) {
        if (!(std_cfg && STDK)) return V2W_E_SHAPE;
        kern = wide_stage_bf16_kernel<MI, NI, WM, WN, OCC, CH, WLDS, true, UPF, true>;
    } else {
        kern = (std_cfg && STDK) ? wide_stage_bf16_kernel<MI, NI, WM, WN, OCC, CH, WLDS, STDK, (STDK ? UPF : 0), false>
                                 : wide_stage_bf16_kernel<MI, NI, WM, WN, OCC, CH, WLDS, false, 0>;
    }
    if (v2w_dry(stream)) return 0;
    hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
    if (e != hipSuccess) return (int)e;
    V2W_LAUNCH(kern, dim3(p.ntiles), dim3(NTH), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_wide)
#endif

// Called by v2w_resblock2_stage_bf16 (v2w_stage_bf16.hip) for C >= 64 on bf16 tensors.  V2W_E_SHAPE: the caller issues the convs one by one.
// up_tiles_out: receives the rows of up_stats_part of a fused-upsampler call.
int v2w_resblock2_stage_bf16_wide(const v2w_stage_split_args* a, hipStream_t stream, int* up_tiles_out) {
    if (a->io_bf16 != 3 || a->nk > V2W_WS_MAXB) return V2W_E_SHAPE;
    if (a->post_out && a->C != 16) return V2W_E_SHAPE;
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    if (a->L % 4 != 0 || !al16(a->in) || !al16(a->out)) return V2W_E_SHAPE;
    if ((long long)a->C * a->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;           // 32-bit offsets inside one batch item
    if (a->rb1 && a->C == 16) return launch_wide<1, 4, 1, 2, 2, 16>(a, stream);  // ResBlock1 pairs on 16 channels: the resident-tile form (one k-step per tap)
    if (a->up_out) {
        // the stage with the next stage's upsampler behind it (stride 4 after the 256- and 128-channel stages, stride 2 after 64 and 32: the
        // generator's (5, 4, 4, 2, 2) and the x640 variant's (8, 5, 4, 2, 2) from the second / third upsampler on)
        if (a->post_out) return V2W_E_SHAPE;
        if (a->C == 256 && a->up_u == 4) {
            // 192-position windows (160 valid outputs) on a half t1 tile, 64 x 96 outputs per wave - 256 tiles = ONE residency of the chip at
            // B = 32 x T = 256 (96-output windows: 448 tiles, 1.75) and a fifth of the MFMA work on halo columns instead of a third
            const int rc = launch_wide<2, 3, 4, 2, 2, 32, false, 4, true>(a, stream, up_tiles_out);
            if (rc != V2W_E_SHAPE) return rc;
            return launch_wide<1, 4, 8, 1, 2, 32, false, 4>(a, stream, up_tiles_out);
        }
        if (a->C == 128 && a->up_u == 4) {
            // the same form at 128 channels: 384-position windows (352 valid) - measured 1.4 % of a configs[2] forward
            const int rc = launch_wide<2, 3, 2, 4, 2, 32, false, 4, true>(a, stream, up_tiles_out);
            if (rc != V2W_E_SHAPE) return rc;
            return launch_wide<1, 4, 4, 2, 2, 32, false, 4>(a, stream, up_tiles_out);
        }
        // (64 channels on that form - <2, 3, 1, 4, .., 2, true>: two workgroups of 384 positions per CU, or <2, 3, 1, 8, .., 2, true>: one of 768 -
        // measured within +-0.5 % of this configuration at configs[2] and 0.7 % slower at B = 32 x T = 256: not taken)
        if (a->C == 64 && a->up_u == 2) return launch_wide<2, 2, 1, 4, 2, 32, false, 2>(a, stream, up_tiles_out);
        if (a->C == 32 && a->up_u == 2) return launch_wide<1, 4, 1, 2, 2, 32, false, 2>(a, stream, up_tiles_out);
        return V2W_E_SHAPE;
    }
    // Measured (one MI355X, configs[2] shapes, us per stage): 8 waves of 64 x 64 outputs 1074 / 1225 / 1133 (C = 128 / 64 / 256) against 4 waves of
    // 64 x 128 with the whole register file 1227 / 1467 / 1199: the single wave per SIMD runs its bare MFMA loop at 89 % of the issue rate
    // but nothing covers its epilogues.  C = 64 fits twice per CU as 4-wave workgroups of 256 positions, which run out of phase.
    // 32 x 128 outputs per wave: a weight fragment feeds FOUR MFMAs (half the vector-memory traffic of the 64 x 64 form, the operand reads
    // from LDS double: 128 B / clk of its 256)
    if (a->C == 128) return launch_wide<1, 4, 4, 2>(a, stream);                   // 128 channels x 256 positions, 8 waves
    if (a->C == 64) return launch_wide<2, 2, 1, 4>(a, stream);                    // 64 channels x 256 positions, 4 waves of 64 x 64, two workgroups per CU (1135-1145 us; <1, 4, 2, 2>: 1160-1173)
    if (a->C == 256) return launch_wide<1, 4, 8, 1>(a, stream);                   // 256 channels x 128 positions, 8 waves
    // 32 channels x 256 positions, TWO waves of 32 x 128 outputs, four workgroups per CU (the workgroups run out of phase: 870 us against the
    // 940 us of stage_bf16_kernel<32>, whose four waves share every barrier)
    if (a->C == 32) return launch_wide<1, 4, 1, 2>(a, stream);
    if (a->C == 16) return launch_wide<1, 4, 1, 2, 2, 16>(a, stream);            // 16 channels x 256 positions, two waves (+ the fused tail)
    return V2W_E_SHAPE;
}
