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
// What shapes the loop (tools/stage_timeline.py, MI355X): the f32 MFMA shares the SIMD's vector ALU, so every vector instruction
// issued between two MFMAs costs matrix time (a v_add + ds_read2_b32 pair per k-step: ~10 cycles per 64-cycle MFMA pair).  The
// main loop therefore contains MFMAs, one ds_read_b128 per four k-steps and column block, one global_load_dwordx4 per four
// k-steps and row block - and no address arithmetic.
// A (weights)  : never staged.  v2w_pack_mfma() stores them in MFMA A-fragment order, so one fully coalesced
//                global_load_dwordx4 per lane (1 KiB per wave; scalar base + lane offset) feeds four consecutive MFMA k-steps;
//                the panel is L2-resident (<= 2.9 MB per layer); RING - 1 fragments stay in flight (vmcnt retires in order, so the
//                ring has to ride out the signal prefetch issued in front of it).
// B (signal)   : LDS tile Xs[NT + halo positions][CK + pad], double buffered over C_in chunks.  POSITION-major, and the channels of
//                a row are permuted (slot()) so that the four k-steps of one packed weight fragment are 16 contiguous bytes: a
//                lane reads its operand for four MFMA k-steps with ONE ds_read_b128 at an immediate offset; row strides of 36 /
//                20 / 24 floats keep those reads bank-conflict free.  Every tap is a row offset into the SAME tile, so each input
//                element is fetched from HBM once per M-tile and the activation / CondBN affine is applied once, at staging time.
//                The next chunk travels global -> registers during the MFMA phase (one barrier per chunk).
// MFMA         : v_mfma_f32_32x32x2_f32 (C_out >= 32) or v_mfma_f32_16x16x4_f32 (C_out == 16): exact fp32
//                (bit-identical to an fmaf chain), 64 FLOP/clk/SIMD.
// Waves        : WM x WN waves per workgroup, each owning (MF*MI) x (MF*NI) outputs for each of the U phases;
//                64-lane fragments: lane&(MF-1) = row/col inside the MFMA tile, lane/MF = k index.
#include <type_traits>
#include "v2w_tile.h"

#ifdef V2W_TIMELINE   // diagnostic build only (see v2w_common.h)
V2W_TL_SETTER(v2w_timeline_set_tile)
#endif

namespace {

// Geometry of the LDS signal tile: [positions][RS floats], the CK channels of a row permuted so that the four k-steps a lane feeds
// to one packed weight fragment (v2w_pack_mfma pairs k-step kk, lane half hk with channel 8g + 2kk + hk for the 32x32x2 MFMA and
// 4kk + hk for 16x16x4) are 16 contiguous bytes.  RS is the smallest 16-byte-aligned stride that keeps the wave's ds_read_b128
// (lane -> row, lane half -> slot quad) bank-conflict free.  Staging writes slot-adjacent channel PAIRS (c0, c0 + PAIR_DC).
template <int MF, int CK> struct TileGeom {
    static constexpr int RS = CK == 32 ? 36 : (MF == 32 ? 20 : 24);
    static constexpr int PAIR_DC = MF == 32 ? 2 : 4;
    __host__ __device__ static constexpr int slot(int c) {
        return MF == 32 ? ((c & ~7) + 4 * (c & 1) + ((c & 7) >> 1)) : (4 * (c & 3) + (c >> 2));
    }
    // first channel of the pair that occupies slots 2P, 2P + 1
    __host__ __device__ static constexpr int pair_c0(int P) {
        return MF == 32 ? (8 * (P >> 2) + 4 * (P & 1) + ((P >> 1) & 1)) : (8 * (P & 1) + (P >> 1));
    }
};

// EPI: which optional epilogue is compiled in.  0 = none (the generator's forward kernels).  1 = the backward-only leaky_relu-
// derivative mask (+ out_slope): it costs 8 VGPRs (one occupancy step on the 128 x 128 tile, 5-12 % of a layer's time), so it is
// its own instantiation.  2 = out_slope only (the discriminators' activated feature maps): no extra registers.
// (second launch bound = waves per SIMD the register allocation must leave room for: the 128 x 128 conv tile runs three
// workgroups per CU - its LDS footprint allows exactly that - and would otherwise drift to 2 through the epilogue's temporaries)
// VEC: every problem of the launch has float4-aligned, unit-stride input (the vector staging path).  A template parameter, not a
// run-time flag: with the element-wise fallback in the same kernel its (never taken) loads join the hot path, and hipcc then waits
// vmcnt(0) in front of every chunk's prefetch and again right behind it - the prefetch's memory latency in the open, once per chunk.
template <int MF, int U, int MI, int NI, int WM, int WN, int CK, int NPF, int RING, int EPI, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN, (U == 1 && MI * NI >= 4) ? 3 : 1)
conv_tile_kernel(const MultiArgs m) {
    typedef Frag<MF> F;
    typedef typename F::acc_t acc_t;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int MT = MF * MI * WM;
    constexpr int NT = MF * NI * WN;
    constexpr int KSTEP = F::KSTEP;
    constexpr int CKG = 4 * KSTEP;          // channels covered by one packed A fragment (4 k-steps)
    constexpr int GPC = CK / CKG;           // A fragments (units) per chunk and tap
    constexpr int RS = TileGeom<MF, CK>::RS;   // floats per position row of the LDS tile
    static_assert(CK % CKG == 0, "chunk must hold whole A fragments");
    static_assert(RING == 2 || (RING == 4 && GPC == 4), "the 4-deep ring walks exactly one tap per revolution");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [xrows][RS] signal tiles, then the tables

    // ---- which tile: ids that differ by a multiple of 8 tend to share an XCD (and its L2), so the M-tiles that
    // re-read the same input tile are placed 8 apart (speed only, never correctness).
    int pq = 0;
#pragma unroll
    for (int i = 1; i < V2W_MAX_MULTI; ++i) pq += (int)blockIdx.x >= m.start[i] ? 1 : 0;
    const TileArgs& p = m.p[pq];
    const int mtiles = p.Cout / MT;
    int id = blockIdx.x - m.start[pq];
    int slice = 0;                            // split over C_in chunks (launch_tile): the grid repeats the tiles once per slice
    if (p.ksplit > 1) {
        const int per = ((p.ntiles + 7) >> 3) * 8 * mtiles;
        slice = id / per;
        id -= slice * per;
    }
    const int grp = id / (8 * mtiles), rem = id % (8 * mtiles);
    const int mt = rem >> 3;
    const int tile = grp * 8 + (rem & 7);
    if (tile >= p.ntiles) return;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * NT;   // first input-rate position of the tile
    const int m0 = mt * MT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & (MF - 1);       // row (A) / column (B, D) inside the MFMA tile
    const int hk = lane / MF;             // k index inside the MFMA k-step
    const int wm0 = (wave / WN) * (MF * MI);
    const int wn0 = (wave % WN) * (MF * NI);
    const int L = p.L, K = p.K;
    const float slope = p.slope;
    const int nch = p.Cin / CK;
    const int nchs = p.ksplit > 1 ? nch / p.ksplit : nch;     // chunks of this workgroup: [ch0, ch1)
    const int ch0 = slice * nchs, ch1 = ch0 + nchs;
    float* const outp = p.out + (size_t)slice * p.B * p.CoutT * L * U;
    const int pos0 = n0 - p.hla;           // position of LDS row 0
    const int bufsz = p.xrows * RS;        // floats per signal buffer
    float* const etab = smem + p.atab_off;    // epilogue constants of this M-tile: bias, res_a, res_s, mask_a, mask_s [MT] each
    float* const atab = etab + 5 * MT;        // folded CondBN affine of this batch item: a[Cin] then s[Cin]

    V2W_STAMP(0);
    acc_t acc[U][MI][NI];
#pragma unroll
    for (int r = 0; r < U; ++r)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < F::NREG; ++e) acc[r][i][j][e] = 0.f;

    // ---- signal staging, all waves cooperate.  An item = one slot-adjacent channel pair x 4 positions: two float4 loads, four
    // 8-byte LDS stores.  Each 16-lane quarter of a wave takes 8 consecutive position quads of 2 pairs: on the global side that is two
    // full 128-byte lines per row pair (the same line count as a row-contiguous 256 bytes); on the LDS side the stores are 4-way bank
    // conflicted (rows 4 apart are 16 banks apart at every 16-byte aligned stride) - 48 wave-stores per chunk, noise against its MFMAs.
    // prefetch() only issues the global loads; commit() applies affine + leaky_relu and writes LDS a chunk later.
    const int nq = p.xrows >> 2;             // position quads per row of items
    const int nq8 = (nq + 7) >> 3;           // ... in groups of 8
    const unsigned magic = (unsigned)(((1ull << 32) + nq8 - 1) / nq8);   // g / nq8 == umulhi(g, magic) for g < 8192
    f32x4 pf[NPF][2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // item -> (channel c0 of the pair, first row) is recomputed where needed instead of living in registers
    auto item = [&](int s, int& c0, int& row, bool& in_img, bool& in_seq) {
        int t = tid;
        asm volatile("" : "+v"(t));     // opaque: keeps hipcc from hoisting registers per slot out of the chunk loop
        const int idx = t + s * NTHREADS;
        const int g = idx >> 4;                                   // group of 16 items: 8 quads x 2 pairs
        const int pg = (int)__umulhi((unsigned)g, magic);         // pair-of-pairs index
        const int quad = (g - pg * nq8) * 8 + (idx & 7);
        const int P = 2 * pg + ((idx >> 3) & 1);
        row = quad * 4;
        in_img = quad < nq && P < CK / 2;
        c0 = TileGeom<MF, CK>::pair_c0(in_img ? P : 0);
        const int pos = pos0 + row;
        // L % 4 == 0 and pos % 4 == 0: a float4 is entirely inside [0, L) or entirely padding
        in_seq = in_img && pos >= 0 && pos < L;
    };
    auto prefetch = [&](int ci0) {
        const float* src = p.in + (size_t)(b * p.CinT + ci0) * L + pos0;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            int c0, row; bool in_img, in_seq;
            item(s, c0, row, in_img, in_seq);
            pf[s][0] = zero4; pf[s][1] = zero4;
            if (in_seq) {
                pf[s][0] = *reinterpret_cast<const f32x4*>(src + (size_t)c0 * L + row);
                pf[s][1] = *reinterpret_cast<const f32x4*>(src + (size_t)(c0 + TileGeom<MF, CK>::PAIR_DC) * L + row);
            }
        }
    };
    auto commit = [&](int ci0, float* Xs) {
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            int c0, row; bool in_img, in_seq;
            item(s, c0, row, in_img, in_seq);
            if (!in_img) continue;
            float a0 = 1.f, s0 = 0.f, a1 = 1.f, s1 = 0.f;
            if (p.in_a) {
                a0 = atab[ci0 + c0]; s0 = atab[p.Cin + ci0 + c0];
                a1 = atab[ci0 + c0 + TileGeom<MF, CK>::PAIR_DC]; s1 = atab[p.Cin + ci0 + c0 + TileGeom<MF, CK>::PAIR_DC];
            }
            float* dst = Xs + row * RS + TileGeom<MF, CK>::slot(c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x2 v = {0.f, 0.f};        // padding stays exactly 0 (it pads the ACTIVATED signal)
                if (in_seq) { v[0] = v2w_lrelu(fmaf(a0, pf[s][0][e], s0), slope); v[1] = v2w_lrelu(fmaf(a1, pf[s][1][e], s1), slope); }
                *reinterpret_cast<f32x2*>(dst + e * RS) = v;
            }
        }
    };
    auto stage_scalar = [&](int ci0, float* Xs) {   // any L / alignment / input stride: dword loads straight into LDS
        for (int c = wave; c < CK; c += WM * WN) {
            const int ch = b * p.CinT + ci0 + c;
            const float* src = p.in + (size_t)ch * L * p.in_stride + p.in_phase;
            const float av = p.in_a ? p.in_a[ch] : 1.f;
            const float sv = p.in_s ? p.in_s[ch] : 0.f;
            float* dst = Xs + TileGeom<MF, CK>::slot(c);
            for (int j = lane; j < p.xrows; j += 64) {
                const int l = pos0 + j;
                float v = 0.f;
                if (l >= 0 && l < L) v = v2w_lrelu(fmaf(av, src[(size_t)l * p.in_stride], sv), slope);
                dst[j * RS] = v;
            }
        }
    };

    // ---- packed weights (v2w_pack_mfma): per 32-/16-row block mb the fragments lie in exactly the order this loop
    // consumes them - [chunk][phase-ordered tap][fragment] - 1 KiB (64 lanes x float4) each, so "next" is always +1 KiB.
    // The per-row-block base is wave-uniform (scalar registers); the lane only adds its 16-byte offset.
    const int nfrag = nch * K * GPC;        // fragments per row block
    const f32x4* ap[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
        ap[i] = reinterpret_cast<const f32x4*>(p.wp) + ((size_t)((m0 + wm0) / MF + i) * nfrag) * 64;
    int fidx = ch0 * K * GPC;               // fragment the NEXT load fetches (clamped at the end: a harmless re-read)
    f32x4 ar[RING][MI];                     // weight ring: RING - 1 fragments in flight (every index below is a compile-time constant)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_next = [&](f32x4 (&a)[MI]) {
        const int f = fidx < nfrag ? fidx : nfrag - 1;
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));     // opaque: otherwise hipcc folds base + lane into a loop-invariant VECTOR address and adds the
                                          // fragment offset with 64-bit vector adds; this way it is global_load_dwordx4 v, v_off, s[base]
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ap[i] + (size_t)f * 64) + l16);
        ++fidx;
    };

    // ---- B operands: one float4 (four k-steps) per column block and unit.  With few MFMAs per unit (MI*NI < 4) the next unit's
    // float4s are requested into a second register set at the start of the running unit; otherwise each column block's register is
    // refilled right after its last use in the running unit (MI*(NI-1) MFMAs before it is needed again).
    constexpr bool DB = MI * NI < 4;
    f32x4 bb[DB ? 2 : 1][NI];
    // RB: ring / operand-set slot of this tap's first unit (always 0 unless GPC == 1)
    auto tap = [&](auto rb_c, acc_t (&c)[MI][NI], const float* xt, const float* xn) {
        constexpr int RB = decltype(rb_c)::value;
#pragma unroll
        for (int gg = 0; gg < GPC; ++gg) {
            load_next(ar[(RB + gg + RING - 1) % RING]);
            const float* src = gg + 1 < GPC ? xt + CKG * (gg + 1) : xn;      // next unit: same rows, next 8 / 16 slots - or the next tap
            const int cur = DB ? (RB + gg) & 1 : 0;      // compile-time after unrolling
            if constexpr (DB) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bb[cur ^ 1][j] = *reinterpret_cast<const f32x4*>(src + j * MF * RS);
            }
            __builtin_amdgcn_sched_barrier(0);      // operand requests stay AHEAD of this unit's MFMAs
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) c[i][j] = F::mfma(ar[(RB + gg) % RING][i][kk], bb[cur][j][kk], c[i][j]);
                    if constexpr (!DB) {
                        if (kk == 3) bb[0][j] = *reinterpret_cast<const f32x4*>(src + j * MF * RS);
                    }
                }
                if constexpr (!DB) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto prime = [&](auto rb_c, const float* xt) {
        constexpr int RB = decltype(rb_c)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j) bb[DB ? (RB & 1) : 0][j] = *reinterpret_cast<const f32x4*>(xt + j * MF * RS);
    };

    // ---- prologue: epilogue constants, affine table, chunk 0, first fragments
    for (int c = tid; c < MT; c += NTHREADS) {
        etab[c] = p.bias ? p.bias[m0 + c] : 0.f;
        etab[MT + c] = p.res_a ? p.res_a[b * p.Cout + m0 + c] : 1.f;
        etab[2 * MT + c] = p.res_a ? p.res_s[b * p.Cout + m0 + c] : 0.f;
        etab[3 * MT + c] = p.mask_a ? p.mask_a[b * p.Cout + m0 + c] : 1.f;
        etab[4 * MT + c] = p.mask_a ? p.mask_s[b * p.Cout + m0 + c] : 0.f;
    }
    if (p.in_a) {
        for (int c = tid; c < p.Cin; c += NTHREADS) {
            atab[c] = p.in_a[b * p.Cin + c];
            atab[p.Cin + c] = p.in_s[b * p.Cin + c];
        }
        __syncthreads();
    }
    if constexpr (VEC) { prefetch(ch0 * CK); commit(ch0 * CK, smem); }
    else stage_scalar(ch0 * CK, smem);
#pragma unroll
    for (int g = 0; g + 1 < RING; ++g) load_next(ar[g]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    V2W_STAMP(1);

    const int lbase = (wn0 + lr + p.hla) * RS + 4 * hk;   // this lane's float4 in LDS row (position n0 + its column), unit 0
    typedef std::integral_constant<int, 0> RB0;
    typedef std::integral_constant<int, 1> RB1;
    for (int ch = ch0; ch < ch1; ++ch) {
        const float* Xs = smem + ((ch - ch0) & 1) * bufsz;
        float* Xn = smem + ((ch - ch0 + 1) & 1) * bufsz;
        const bool more = ch + 1 < ch1;
        if constexpr (VEC) { if (more) prefetch((ch + 1) * CK); }   // in flight during the MFMA phase below
        __builtin_amdgcn_sched_barrier(0);
        if (ch - ch0 < 6) V2W_STAMP(2 + 4 * (ch - ch0));

        // per phase r (one phase for a conv): taps at rows d0 + m*dstr, m < nt
        auto phase = [&](int r, int& d0, int& dstr, int& nt) {
            if (U == 1) { d0 = -p.hl; dstr = p.dil; nt = K; }
            else { const int rp = r + p.pad, t0 = rp % U; d0 = rp / U; dstr = -1; nt = (K - t0 + U - 1) / U; }
        };
        {
            int d0, dstr, nt;
            phase(0, d0, dstr, nt);
            prime(RB0{}, Xs + lbase + d0 * RS);
        }
#pragma unroll
        for (int r = 0; r < U; ++r) {
            int d0, dstr, nt;
            phase(r, d0, dstr, nt);
            const int step = dstr * RS;
            const float* xt = Xs + lbase + d0 * RS;
            const float* xlast = xt;                  // where the reads run on after this phase's last tap (last phase: never used)
            if (r + 1 < U) { int d1, s1, n1; phase(r + 1, d1, s1, n1); xlast = Xs + lbase + d1 * RS; }
            if constexpr (GPC > 1) {
                for (int t = 0; t + 1 < nt; ++t, xt += step) tap(RB0{}, acc[r], xt, xt + step);
                tap(RB0{}, acc[r], xt, xlast);
            } else {                                  // GPC == 1, RING == 2: ring and operand set alternate per tap -> taps in pairs
                int t = 0;
                for (; t + 1 < nt; t += 2, xt += 2 * step) {
                    tap(RB0{}, acc[r], xt, xt + step);
                    tap(RB1{}, acc[r], xt + step, t + 2 < nt ? xt + 2 * step : xlast);
                }
                if (t < nt) {                         // odd count: the in-flight fragment / operands sit in slot 1; hand them over
                    tap(RB0{}, acc[r], xt, xlast);
#pragma unroll
                    for (int i = 0; i < MI; ++i) ar[0][i] = ar[1][i];
                    if constexpr (DB) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) bb[0][j] = bb[1][j];
                    }
                }
            }
        }

        if (ch - ch0 < 6) V2W_STAMP(3 + 4 * (ch - ch0));
        if (more) {
            if constexpr (VEC) commit((ch + 1) * CK, Xn);
            else stage_scalar((ch + 1) * CK, Xn);
            if (ch - ch0 < 6) V2W_STAMP(4 + 4 * (ch - ch0));
            __syncthreads();   // Xn complete for the next iteration; everyone done with Xs before it is overwritten again
            if (ch - ch0 < 6) V2W_STAMP(5 + 4 * (ch - ch0));
        }
    }
    V2W_STAMP(26);

    // ---- epilogue: + bias [+ residual] [+ out] [/ out_div]; the U phases of one (co, q) are U consecutive floats.
    const int Lout = L * U;
    if constexpr (U == 1) {
        // The accumulator tile of a wave (one 32- / 16-row block at a time) goes through a wave-private LDS scratch (the signal
        // buffers are dead) and comes back as float4s ALONG positions: 16-byte loads of the residual / addends and 16-byte stores, a
        // quarter of the memory instructions of a row-per-register epilogue, all loads of a group in flight together, and - as
        // important - a ROLLED loop: the fully unrolled row-per-register form made this kernel 66 KB of code against a 64 KB
        // instruction cache shared by two CUs, and its once-per-tile straight-line code ran at instruction-fetch speed (15-45 % of a
        // tile's time in the tools/tile_timeline.py stamps).
        constexpr bool MASK = EPI == 1;
        constexpr int ERS = MF * NI;               // floats per scratch row (the float4 read-back is conflict-free unpadded)
        constexpr int C4 = ERS / 4;                // float4s per row
        constexpr int NIT = MF * C4 / 64;          // float4s per lane and row block
        constexpr int GV = NIT < 4 ? NIT : 4;      // float4s per lane processed together
        static_assert(NIT % GV == 0 && (MF * C4) % 64 == 0, "row block must split evenly over the wave");
        __syncthreads();                           // every wave is done with the signal tiles
        float* const scr = smem + wave * (MF * ERS);
        const float dinv = p.out_div != 0.f ? 1.f / p.out_div : 1.f;
        float* const rsum = atab + (p.in_a ? 2 * p.Cin : 0);      // MASK + stats_part: [WN][MT] row sums of the waves (behind the tables)
        // the common forward cases - no addend or only the residual - request ALL their residual float4s of a row block before the
        // block's LDS transposition, so one memory round trip per row block overlaps the LDS traffic; the rarer combinations
        // (running sum, two addends, mask) go GV float4s at a time
        const bool simple = p.evec && !p.accumulate && !p.add0 && !p.add1 && !(MASK && p.mask_src);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const size_t gbase = ((size_t)b * p.CoutT + m0 + wm0 + i * MF) * L + n0 + wn0;
            const int cbase = wm0 + i * MF;
            f32x4 rall[NIT];
            if (simple) {
#pragma unroll
                for (int g = 0; g < NIT; ++g) {
                    const int idx = lane + 64 * g;
                    const int row = idx / C4, c4 = idx - row * C4;
                    rall[g] = zero4;
                    if (p.res && n0 + wn0 + 4 * c4 < L) rall[g] = *reinterpret_cast<const f32x4*>(p.res + gbase + (size_t)row * L + 4 * c4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < F::NREG; ++e) scr[F::row(e, hk) * ERS + j * MF + lr] = acc[0][i][j][e];
            // (wave-private scratch: the wave's own LDS writes are ordered before its reads by the waitcnt hipcc emits)
            __builtin_amdgcn_sched_barrier(0);
            if (simple) {
#pragma unroll
                for (int g = 0; g < NIT; ++g) {
                    const int idx = lane + 64 * g;
                    const int row = idx / C4, c4 = idx - row * C4;
                    const int col = cbase + row;
                    const float bias = etab[col], ra = etab[MT + col], rs = etab[2 * MT + col];
                    f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * ERS + 4 * c4);
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        float t = v[x] + bias;
                        if (p.res) t += fmaf(ra, rall[g][x], rs);
                        if (p.out_div != 0.f) t = v2w_div_by(t, p.out_div, dinv);
                        if constexpr (EPI != 0)
                            if (p.out_slope != 1.f) t = t > 0.f ? t : t * p.out_slope;
                        v[x] = t;
                    }
                    if (n0 + wn0 + 4 * c4 < L) *reinterpret_cast<f32x4*>(outp + gbase + (size_t)row * L + 4 * c4) = v;
                }
            } else if (p.evec) {
#pragma unroll 1
                for (int g0 = 0; g0 < NIT; g0 += GV) {
                    f32x4 rv[GV], ov[GV], o2[GV], mv[MASK ? GV : 1];
#pragma unroll
                    for (int g = 0; g < GV; ++g) {
                        const int idx = lane + 64 * (g0 + g);
                        const int row = idx / C4, c4 = idx - row * C4;
                        const bool ok = n0 + wn0 + 4 * c4 < L;             // L % 4 == 0: a float4 is inside or outside as a whole
                        const size_t goff = gbase + (size_t)row * L + 4 * c4;
                        rv[g] = ov[g] = o2[g] = zero4;
                        if (ok) {
                            if (p.res) rv[g] = *reinterpret_cast<const f32x4*>(p.res + goff);
                            if (p.accumulate) ov[g] = *reinterpret_cast<const f32x4*>(outp + goff);
                            else if (p.add0) ov[g] = *reinterpret_cast<const f32x4*>(p.add0 + goff);
                            if (p.add1) o2[g] = *reinterpret_cast<const f32x4*>(p.add1 + goff);
                        }
                        if constexpr (MASK) {
                            mv[g] = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (ok && p.mask_src) mv[g] = *reinterpret_cast<const f32x4*>(p.mask_src + goff);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GV; ++g) {
                        const int idx = lane + 64 * (g0 + g);
                        const int row = idx / C4, c4 = idx - row * C4;
                        const int col = cbase + row;
                        const float bias = etab[col], ra = etab[MT + col], rs = etab[2 * MT + col];
                        f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * ERS + 4 * c4);
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            float t = v[x];
                            if constexpr (MASK)
                                if (p.mask_src) t = fmaf(etab[3 * MT + col], mv[g][x], etab[4 * MT + col]) > 0.f ? t : t * p.mask_slope;
                            t += bias;
                            if (p.res) t += fmaf(ra, rv[g][x], rs);
                            if (p.add1) t += ov[g][x] + o2[g][x];      // (add0 + add1) + value: the reference's `xs += ...` order
                            else if (p.accumulate || p.add0) t += ov[g][x];
                            if (p.out_div != 0.f) t = v2w_div_by(t, p.out_div, dinv);
                            if constexpr (EPI != 0)
                                if (p.out_slope != 1.f) t = t > 0.f ? t : t * p.out_slope;
                            v[x] = t;
                        }
                        if (n0 + wn0 + 4 * c4 < L) *reinterpret_cast<f32x4*>(outp + gbase + (size_t)row * L + 4 * c4) = v;
                        if constexpr (MASK) {
                            // per-channel sum of what this launch stores (the bias gradient of the layer whose output gradient it is):
                            // the C4 lanes that share a row meet in a shuffle tree, the row's first lane parks the sum in LDS
                            if (p.stats_part) {
                                float sr = n0 + wn0 + 4 * c4 < L ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
#pragma unroll
                                for (int off = 1; off < C4; off <<= 1) sr += __shfl_xor(sr, off, 64);
                                if (c4 == 0) rsum[(wave % WN) * MT + col] = sr;
                            }
                        }
                    }
                }
            } else {                               // ragged L / unaligned operands: the same walk, one element at a time
#pragma unroll 1
                for (int idx = lane; idx < MF * ERS; idx += 64) {
                    const int row = idx / ERS, c = idx - row * ERS;
                    if (n0 + wn0 + c >= L) continue;
                    const int col = cbase + row;
                    const size_t goff = gbase + (size_t)row * L + c;
                    float t = scr[idx];
                    if constexpr (MASK)
                        if (p.mask_src) t = fmaf(etab[3 * MT + col], p.mask_src[goff], etab[4 * MT + col]) > 0.f ? t : t * p.mask_slope;
                    t += etab[col];
                    if (p.res) t += fmaf(etab[MT + col], p.res[goff], etab[2 * MT + col]);
                    if (p.add1) t += p.add0[goff] + p.add1[goff];
                    else if (p.accumulate) t += outp[goff];
                    else if (p.add0) t += p.add0[goff];
                    if (p.out_div != 0.f) t = v2w_div_by(t, p.out_div, dinv);
                    if constexpr (EPI != 0)
                        if (p.out_slope != 1.f) t = t > 0.f ? t : t * p.out_slope;
                    outp[goff] = t;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MASK) {
            if (p.stats_part) {                   // [tile][Cout][2] = (sum of the stored values over the tile's positions, 0): bn_reduce_partials adds the tiles
                __syncthreads();
                for (int c = tid; c < MT; c += NTHREADS) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) t += rsum[w * MT + c];
                    p.stats_part[((size_t)tile * p.Cout + m0 + c) * 2 + 0] = t;
                    p.stats_part[((size_t)tile * p.Cout + m0 + c) * 2 + 1] = 0.f;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int e = 0; e < F::NREG; ++e) {
                const int co = m0 + wm0 + i * MF + F::row(e, hk);
                const float bias = etab[co - m0];
                const size_t orow = ((size_t)b * p.CoutT + co) * Lout;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int q = n0 + wn0 + j * MF + lr;
                    if (q >= L) continue;
                    if constexpr (U == 2) {
                        f32x2 v; v[0] = acc[0][i][j][e] + bias; v[1] = acc[1][i][j][e] + bias;
                        *reinterpret_cast<f32x2*>(outp + orow + (size_t)q * 2) = v;
                    } else if constexpr (U == 4) {
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[r][i][j][e] + bias;
                        *reinterpret_cast<f32x4*>(outp + orow + (size_t)q * 4) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < U; ++r) outp[orow + (size_t)q * U + r] = acc[r][i][j][e] + bias;
                    }
                }
            }
        }
    }
    V2W_STAMP(27);

    // ---- fused BatchNorm statistics of the transposed-conv output (modules.py:23): per tile and channel (sum, sumsq),
    // lanes -> wavefront shuffles -> the WN waves through LDS in fixed order -> one slot per (tile, channel); the slots are
    // summed in fp64 by bn_reduce_partials_kernel, so the result is bit-reproducible (no atomics).
    if constexpr (U > 1) {
        if (p.stats_part) {
            __syncthreads();                      // everyone is done with the signal tiles: reuse them as scratch
            float* red = smem;                    // [WN][MT][2]
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int e = 0; e < F::NREG; ++e) {
                    const int col = wm0 + i * MF + F::row(e, hk);
                    const float bias = etab[col];
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int q = n0 + wn0 + j * MF + lr;
                        if (q < L) {
#pragma unroll
                            for (int r = 0; r < U; ++r) { const float v = acc[r][i][j][e] + bias; s1 += v; s2 = fmaf(v, v, s2); }
                        }
                    }
#pragma unroll
                    for (int off = MF / 2; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
                    if (lr == 0) {
                        red[((wave % WN) * MT + col) * 2 + 0] = s1;
                        red[((wave % WN) * MT + col) * 2 + 1] = s2;
                    }
                }
            }
            __syncthreads();
            for (int c = tid; c < MT; c += NTHREADS) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) { t1 += red[(w * MT + c) * 2]; t2 += red[(w * MT + c) * 2 + 1]; }
                p.stats_part[((size_t)tile * p.Cout + m0 + c) * 2 + 0] = t1;
                p.stats_part[((size_t)tile * p.Cout + m0 + c) * 2 + 1] = t2;
            }
        }
    }
}

// ---- split over C_in for launches that cannot fill the chip (inference at B = 1: conv_pre at T = 50 is 64 workgroups walking
// 24 chunks x 7 taps one after the other, 176 us).  The launch is repeated over `ksplit` slices of the chunks, every slice writes its
// plain partial sums to its own slab of a per-(device, stream) workspace (same (B, C_out, L_out) layout as the output), and
// splitk_reduce_kernel adds the slabs in slice order and applies the epilogue - bias, residual, addends, division - in the main
// kernel's arithmetic order: deterministic, no atomics, visibility from the kernel boundary.
struct SplitEpi {
    const float* slab; float* out;
    const float* bias; const float* res; const float* res_a; const float* res_s; const float* add0; const float* add1;
    int accumulate; float out_div;
    int Cout, Lout;
    size_t slice_stride;      // floats between the slabs of consecutive slices
};
struct SplitEpiArgs {
    SplitEpi e[V2W_MAX_MULTI];
    long long start[V2W_MAX_MULTI + 1];     // first float4 of problem i in the flat index space
    int n, S;
};

// VEC: float4 per thread (every L_out a multiple of 4, every pointer 16-byte aligned); else one element per thread
template <bool VEC>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const SplitEpiArgs a) {
    constexpr int W = VEC ? 4 : 1;
    typedef typename std::conditional<VEC, f32x4, float>::type vec_t;
    const long long total = a.start[a.n];
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        int q = 0;
#pragma unroll
        for (int i = 1; i < V2W_MAX_MULTI; ++i) q += (i < a.n && idx >= a.start[i]) ? 1 : 0;
        const SplitEpi& e = a.e[q];
        const size_t off = (size_t)(idx - a.start[q]) * W;
        const size_t row = off / e.Lout;                  // b * Cout + co (VEC: Lout % 4 == 0, a float4 never crosses rows)
        const int co = (int)(row % e.Cout);
        float v[W], r[W], ov[W], o2[W];
        auto ld = [&](const float* src, float (&dst)[W]) {
            const vec_t t = *reinterpret_cast<const vec_t*>(src + off);
            if constexpr (VEC) { dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3]; } else dst[0] = t;
        };
        ld(e.slab, v);
        for (int s = 1; s < a.S; ++s) {
            float t[W];
            ld(e.slab + (size_t)s * e.slice_stride, t);
#pragma unroll
            for (int x = 0; x < W; ++x) v[x] += t[x];
        }
        const float bias = e.bias ? e.bias[co] : 0.f;
#pragma unroll
        for (int x = 0; x < W; ++x) r[x] = ov[x] = o2[x] = 0.f;
        float ra = 1.f, rs = 0.f;
        if (e.res) { ld(e.res, r); if (e.res_a) { ra = e.res_a[row]; rs = e.res_s[row]; } }
        if (e.accumulate) ld(e.out, ov);
        else if (e.add0) ld(e.add0, ov);
        if (e.add1) ld(e.add1, o2);
        const float dinv = e.out_div != 0.f ? 1.f / e.out_div : 1.f;
#pragma unroll
        for (int x = 0; x < W; ++x) {
            float t2 = v[x] + bias;
            if (e.res) t2 += fmaf(ra, r[x], rs);
            if (e.add1) t2 += ov[x] + o2[x];
            else if (e.accumulate || e.add0) t2 += ov[x];
            if (e.out_div != 0.f) t2 = v2w_div_by(t2, e.out_div, dinv);
            v[x] = t2;
        }
        if constexpr (VEC) *reinterpret_cast<f32x4*>(e.out + off) = f32x4{v[0], v[1], v[2], v[3]};
        else e.out[off] = v[0];
    }
}

// HMAX: largest halo (each side) the staging slots cover: 32 for the generator (k = 11, dilation 5 -> 25); the 48 variants serve
// DiscriminatorP's dilation = period convs (k = 5, period 19 -> 38) at one more prefetch slot per thread.
template <int MF, int U, int MI, int NI, int WM, int WN, int CK, int HMAX = 32>
int launch_tile(const TileArgs* ps, int nprob, hipStream_t stream) {
    constexpr int MT = MF * MI * WM, NT = MF * NI * WN, NTHREADS = 64 * WM * WN;
    constexpr int RS = TileGeom<MF, CK>::RS;
    constexpr int NPF = ((CK / 2) * (((NT + 2 * HMAX) / 4 + 7) / 8 * 8) + NTHREADS - 1) / NTHREADS;   // staging items (channel pair x 4 positions) per thread, quads padded to groups of 8
    constexpr int KSTEP = MF == 32 ? 2 : 4;
    constexpr int RING = (CK / (4 * KSTEP) == 4) ? 4 : 2;
    static_assert(NTHREADS * NPF < 8 * 8192, "item index range of the magic division");
    if (nprob < 1 || nprob > V2W_MAX_MULTI) return V2W_E_ARG;
    MultiArgs m{};
    size_t lds = 0;
    int grid = 0;
    for (int i = 0; i < nprob; ++i) {
        TileArgs p = ps[i];
        if (p.Cout % MT != 0 || p.Cin % CK != 0) return V2W_E_SHAPE;
        p.hla = (p.hl + 3) & ~3;
        if (p.hla > HMAX || p.hr > HMAX) return V2W_E_SHAPE;   // (before the configuration query answers: a probe must see it too)
        if (p.cfg_out) {
            const int c[10] = {MF, U, MI, NI, WM, WN, CK, NPF, RING, p.B * ((p.L + NT - 1) / NT)};
            for (int k = 0; k < 10; ++k) p.cfg_out[k] = c[k];   // [9] = number of position tiles (rows of stats_part)
            return 0;
        }
        p.ntl = (p.L + NT - 1) / NT;
        p.ntiles = p.B * p.ntl;
        p.xrows = (p.hla + NT + p.hr + 3) & ~3;
        p.xcols = p.xrows; p.xw = RS;
        if (p.in_stride < 1) p.in_stride = 1;
        p.vec4 = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0) && p.in_stride == 1;
        const int nbuf = p.Cin / CK > 1 ? 2 : 1;
        p.atab_off = nbuf * p.xrows * RS;
        // vector epilogue (conv only): float4 I/O along positions through a wave-private LDS scratch that overlays the signal buffers
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        p.evec = U == 1 && p.L % 4 == 0 && al16(p.out) && al16(p.res) && al16(p.add0) && al16(p.add1) && al16(p.mask_src);
        if (U == 1 && p.atab_off < WM * WN * MF * MF * NI) p.atab_off = WM * WN * MF * MF * NI;   // room for the epilogue scratch
        size_t l = ((size_t)p.atab_off + 5 * MT + (p.in_a ? 2 * p.Cin : 0)) * sizeof(float);
        if (U == 1 && p.stats_part) {                 // row sums of a masked launch (bias gradients): vector epilogue only, [WN][MT] floats behind the tables
            if (!p.evec || !p.mask_src || p.CoutT != p.Cout) return V2W_E_ARG;
            l += (size_t)WN * MT * sizeof(float);
        }
        if (U > 1 && p.stats_part) {                  // the fused BatchNorm partials reuse the signal buffers as [WN][MT][2] scratch
            const size_t need = (size_t)WN * MT * 2 * sizeof(float);
            if ((size_t)p.atab_off * sizeof(float) < need) return V2W_E_SHAPE;
        }
        if (l > lds) lds = l;
        m.p[i] = p;
        m.start[i] = grid;
        grid += ((p.ntiles + 7) / 8) * 8 * (p.Cout / MT);
    }
    for (int i = nprob; i <= V2W_MAX_MULTI; ++i) m.start[i] = i == nprob ? grid : 0x7fffffff;
    m.start[nprob] = grid;
    for (int i = nprob + 1; i <= V2W_MAX_MULTI; ++i) m.start[i] = 0x7fffffff;
    int epi = 0;
    for (int i = 0; i < nprob; ++i) {
        if (m.p[i].mask_src != nullptr) epi = 1;
        else if (m.p[i].out_slope != 1.f && epi == 0) epi = 2;
    }
    // split over C_in chunks (see splitk_reduce_kernel) when the launch is at most half a workgroup per CU and every problem has the
    // plain epilogue: S = the largest power of two that divides the chunk count and keeps the grid within two workgroups per CU
    SplitEpiArgs red{};
    int S = 1;
    bool red_vec = true;
    if (epi == 0 && grid <= 128) {
        const int nch = m.p[0].Cin / CK;
        int s2 = 1;
        // (a launch whose serial chain is short - fewer than 24 (chunk, tap) steps: the narrow upsamplers, 64-channel convs - gains less
        // from the split than the reduce launch behind it costs)
        while (s2 * 2 <= 8 && nch % (s2 * 2) == 0 && grid * s2 * 2 <= 512 && nch * m.p[0].K >= 24) s2 *= 2;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        size_t floats = 0;
        bool ok = s2 > 1, vec = true;
        for (int i = 0; i < nprob && ok; ++i) {
            const TileArgs& p = m.p[i];
            const size_t n = (size_t)p.B * p.Cout * p.L * U;
            ok = !p.stats_part && p.CoutT == p.Cout && p.Cin / CK == nch;
            vec = vec && (p.L * U) % 4 == 0 && al16(p.out) && al16(p.res) && al16(p.add0) && al16(p.add1);
            floats += n * s2;
        }
        if (m.p[0].ws_query) {                       // host-only query: bytes of caller scratch this launch would use
            *m.p[0].ws_query = ok ? (long long)(floats * sizeof(float)) : 0;
            return 0;
        }
        if (ok && m.p[0].splitk_ws && (long long)(floats * sizeof(float)) <= m.p[0].splitk_ws_bytes) {
            float* ws = m.p[0].splitk_ws;           // caller-owned (v2w_conv1d_args::splitk_ws): nothing is allocated or kept here
            {
                S = s2;
                red_vec = vec;
                size_t off = 0;
                long long f4 = 0;
                grid = 0;
                for (int i = 0; i < nprob; ++i) {
                    TileArgs& p = m.p[i];
                    const size_t n = (size_t)p.B * p.Cout * p.L * U;
                    red.e[i] = SplitEpi{ws + off, p.out, p.bias, p.res, p.res_a, p.res_s, p.add0, p.add1, p.accumulate, p.out_div, p.Cout, p.L * U, n};
                    red.start[i] = f4;
                    f4 += vec ? (long long)(n / 4) : (long long)p.B * p.Cout * p.L * U;
                    p.out = ws + off; p.bias = nullptr; p.res = p.res_a = p.res_s = nullptr; p.add0 = p.add1 = nullptr;
                    p.accumulate = 0; p.out_div = 0.f; p.ksplit = S;
                    p.evec = U == 1 && p.L % 4 == 0;
                    off += n * S;
                    m.start[i] = grid;
                    grid += ((p.ntiles + 7) / 8) * 8 * (p.Cout / MT) * S;
                }
                m.start[nprob] = grid;
                red.start[nprob] = f4;
                red.n = nprob; red.S = S;
            }
        }
    }
    if (m.p[0].ws_query) { *m.p[0].ws_query = 0; return 0; }
    bool vec = true;
    for (int i = 0; i < nprob; ++i) vec = vec && m.p[i].vec4;
    auto kern = vec ? (epi == 1 ? conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF, RING, 1, true>
                       : (epi == 2 ? conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF, RING, 2, true> : conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF, RING, 0, true>))
                    : (epi == 1 ? conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF, RING, 1, false>
                       : (epi == 2 ? conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF, RING, 2, false> : conv_tile_kernel<MF, U, MI, NI, WM, WN, CK, NPF, RING, 0, false>));
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return V2W_E_SHAPE;
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(NTHREADS), lds, stream, m);
    if (S > 1) {
        const int rc = v2w_launch_status();
        if (rc != 0) return rc;
        long long blocks = (red.start[nprob] + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        if (red_vec) V2W_LAUNCH(splitk_reduce_kernel<true>, dim3((int)blocks), dim3(256), 0, stream, red);
        else V2W_LAUNCH(splitk_reduce_kernel<false>, dim3((int)blocks), dim3(256), 0, stream, red);
    }
    return v2w_launch_status();
}

template <int U>
int launch_convt_u(TileArgs p, hipStream_t stream) {
    if (p.Cout % 64 == 0) return launch_tile<32, U, 1, 1, 2, 2, 16>(&p, 1, stream);
    if (p.Cout == 32) return launch_tile<32, U, 1, 1, 1, 4, 16>(&p, 1, stream);
    if (p.Cout == 16) return launch_tile<16, U, 1, 2, 1, 4, 16>(&p, 1, stream);
    return V2W_E_SHAPE;
}

// wp float index o = ((((mb*nch + ch)*K + ts)*GPC + gg)*64 + lane)*4 + j  holds
//   wf[tap(ts)][ch*CK + gg*CKG + j*KSTEP + lane/MF][mb*MF + lane%MF]
// where ts runs over the taps in the order the tile kernel consumes them: 0..K-1 for a conv, phase by phase
// (r = 0..U-1: t = (r+pad)%U, +U, ...) for a transposed conv.
__global__ void __launch_bounds__(256)
pack_mfma_kernel(const float* wf, float* wp, int K, int Cin, int Cout, int MF, int CK, int U, int tflip = 0) {
    const int KSTEP = MF == 32 ? 2 : 4, CKG = 4 * KSTEP, GPC = CK / CKG, nch = Cin / CK;
    const int pad = (K - U) / 2;
    const size_t total = (size_t)K * Cin * Cout;
    wf += (size_t)blockIdx.y * total; wp += (size_t)blockIdx.y * total;       // grid.y = matrix of a batch (v2w_pack_mfma_batch)
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int j = o & 3;
        const int lane = (o >> 2) & 63;
        size_t rest = o >> 8;
        const int gg = rest % GPC; rest /= GPC;
        const int ts = rest % K; rest /= K;
        const int ch = rest % nch;
        const int mb = rest / nch;
        int t = ts;
        if (U > 1) {   // ts-th tap in phase-major order
            int left = ts;
            for (int r = 0; r < U; ++r) {
                const int t0 = (r + pad) % U, nt = (K - t0 + U - 1) / U;
                if (left < nt) { t = t0 + left * U; break; }
                left -= nt;
            }
        }
        const int c = ch * CK + gg * CKG + j * KSTEP + lane / MF;
        const int co = mb * MF + lane % MF;
        // tflip: `wf` is the FORWARD layer's [K][Cout][Cin] and this stream serves its input gradient: tap-reversed, transposed
        wp[o] = tflip ? wf[((size_t)(K - 1 - t) * Cout + co) * Cin + c] : wf[((size_t)t * Cin + c) * Cout + co];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Batched weight-norm fold + pack: every MFMA layer of the generator in two launches (scale, pack) driven by a device
// array of descriptors, instead of three tiny launches per layer.
__device__ __forceinline__ int find_layer(const int32_t* __restrict__ starts, int n, int blk) {
    int lo = 0, hi = n;                       // starts[li] <= blk < starts[li+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= blk) lo = mid; else hi = mid; }
    return lo;
}

// scale[row] = g[row] / ||v[row,:]||  (1 when g == NULL); one block per (layer, row)
__global__ void __launch_bounds__(256)
fold_scale_batch_kernel(const v2w_fold_desc* __restrict__ descs, const int32_t* __restrict__ starts, int n) {
    __shared__ double red[16];
    const int li = find_layer(starts, n, blockIdx.x);
    const v2w_fold_desc d = descs[li];
    const int row = blockIdx.x - starts[li];
    const int inner = (d.transposed ? d.c_out : d.c_in) * d.k;
    const float* src = d.v + (size_t)row * inner;
    double acc = 0.0;
    if (d.g)
        for (int i = threadIdx.x; i < inner; i += 256) { const double x = (double)src[i]; acc += x * x; }
    const double n2 = v2w_block_sum(acc, red);
    if (threadIdx.x == 0) d.scale[row] = d.g ? (float)((double)d.g[row] / sqrt(n2)) : 1.f;
}

// one block per (layer, row block mb, channel chunk ch): coalesced read of the MF x CK x K sub-block of v into LDS,
// then the K*GPC fragments of that (mb, ch) are written out contiguously in consumption order
__global__ void __launch_bounds__(256)
fold_pack_batch_kernel(const v2w_fold_desc* __restrict__ descs, const int32_t* __restrict__ starts, int n) {
    extern __shared__ float tile[];
    const int li = find_layer(starts, n, blockIdx.x);
    const v2w_fold_desc d = descs[li];
    const int blk = blockIdx.x - starts[li];
    const int MF = d.mf, CK = d.ck, K = d.k, U = d.u;
    const int KSTEP = MF == 32 ? 2 : 4, CKG = 4 * KSTEP, GPC = CK / CKG, nch = d.c_in / CK;
    const int mb = blk / nch, ch = blk % nch;
    const int nrow = d.transposed ? CK : MF;            // LDS rows
    const int rlen = (d.transposed ? MF : CK) * K;      // contiguous floats per row in v
    const int rstride = rlen + 1;
    for (int idx = threadIdx.x; idx < nrow * rlen; idx += 256) {
        const int r = idx / rlen, x = idx - r * rlen;
        const float* src = d.transposed ? d.v + ((size_t)(ch * CK + r) * d.c_out + mb * MF) * K
                                        : d.v + ((size_t)(mb * MF + r) * d.c_in + ch * CK) * K;
        const float sc = d.scale[d.transposed ? ch * CK + r : mb * MF + r];
        tile[r * rstride + x] = src[x] * sc;
    }
    __syncthreads();
    const int pad = (K - U) / 2;
    float* dst = d.wp + (size_t)(mb * nch + ch) * K * GPC * 256;
    for (int o = threadIdx.x; o < K * GPC * 256; o += 256) {
        const int j = o & 3, lane = (o >> 2) & 63;
        const int rest = o >> 8;
        const int gg = rest % GPC, ts = rest / GPC;
        int t = ts;
        if (U > 1) {
            int left = ts;
            for (int r = 0; r < U; ++r) {
                const int t0 = (r + pad) % U, nt = (K - t0 + U - 1) / U;
                if (left < nt) { t = t0 + left * U; break; }
                left -= nt;
            }
        }
        const int c = gg * CKG + j * KSTEP + lane / MF, co = lane % MF;
        dst[o] = d.transposed ? tile[c * rstride + co * K + t] : tile[co * rstride + c * K + t];
    }
    // ---- the same sub-block in the plain layout wf [k][C_in][C_out] (a training forward: the gradient kernels read it), coalesced along C_out
    if (d.wf) {
        for (int idx = threadIdx.x; idx < K * CK * MF; idx += 256) {
            const int col = idx % MF, c = (idx / MF) % CK, t = idx / (MF * CK);
            d.wf[((size_t)t * d.c_in + ch * CK + c) * d.c_out + mb * MF + col] =
                d.transposed ? tile[c * rstride + col * K + t] : tile[col * rstride + c * K + t];
        }
    }
    // ---- ... and as block (row block ch, chunk mb) of the INPUT-GRADIENT conv's fragment stream (C -> C Conv1d layers, MF == CK): that conv
    // has rows = this layer's input channels, k-channels = its output channels and tap ts = tap K - 1 - ts of this layer
    if (d.wpd) {
        const int nchd = d.c_out / CK;                      // chunks of the gradient conv (over this layer's C_out)
        float* dd = d.wpd + (size_t)(ch * nchd + mb) * K * GPC * 256;
        for (int o = threadIdx.x; o < K * GPC * 256; o += 256) {
            const int j = o & 3, lane = (o >> 2) & 63;
            const int rest = o >> 8;
            const int gg = rest % GPC, ts = rest / GPC;
            const int kc = gg * CKG + j * KSTEP + lane / MF, row = lane % MF;      // k-channel (this layer's C_out index), row (its C_in index)
            dd[o] = tile[kc * rstride + row * K + (K - 1 - ts)];
        }
    }
}

}  // namespace

// Tile configuration of a layer: MFMA fragment (32 / 16, 0 = none) and channel chunk CK.  Shared by v2w_pack_mfma and
// the launchers so that the packed fragment order always matches the kernel instantiation that consumes it.
struct LayerCfg { int mf, ck; };
static LayerCfg v2w_layer_cfg(int c_in, int c_out, int u) {
    LayerCfg c{0, 0};
    if (c_in % 16 != 0) return c;
    if (u == 1) {
        if (c_out % 32 == 0 && c_in % 32 == 0) c = {32, 32};
        else if (c_out % 32 == 0) c = {32, 16};      // C_in = 16 (+32k): backward of the narrowest upsampler
        else if (c_out == 16) c = {16, 16};
    } else if (u == 2 || u == 4 || u == 5 || u == 8) {
        if (c_out % 64 == 0 || c_out == 32) c = {32, 16};
        else if (c_out == 16) c = {16, 16};
    }
    return c;
}

extern "C" int v2w_pack_mfma(const float* wf, float* wp, int k, int c_in, int c_out, int u, void* stream) {
    if (!wf || !wp || k <= 0 || c_in <= 0 || c_out <= 0 || u <= 0) return V2W_E_ARG;
    const LayerCfg cfg = v2w_layer_cfg(c_in, c_out, u);
    if (!cfg.mf) return V2W_E_SHAPE;
    const size_t total = (size_t)k * c_in * c_out;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    V2W_LAUNCH(pack_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wf, wp, k, c_in, c_out, cfg.mf, cfg.ck, u);
    return v2w_launch_status();
}

// The fragment stream of a conv layer's INPUT-GRADIENT convolution straight from the layer's own wf [k][C][D] (C = its C_in, D = its C_out):
// the stream v2w_pack_mfma(v2w_wf_transpose_flip(wf), k, D, C, 1) would give, without the transposed copy.  c_in / c_out are the
// gradient convolution's (c_in = D, c_out = C).
extern "C" int v2w_pack_mfma_dgrad(const float* wf, float* wp, int k, int c_in, int c_out, void* stream) {
    if (!wf || !wp || k <= 0 || c_in <= 0 || c_out <= 0) return V2W_E_ARG;
    const LayerCfg cfg = v2w_layer_cfg(c_in, c_out, 1);
    if (!cfg.mf) return V2W_E_SHAPE;
    const size_t total = (size_t)k * c_in * c_out;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    V2W_LAUNCH(pack_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wf, wp, k, c_in, c_out, cfg.mf, cfg.ck, 1, 1);
    return v2w_launch_status();
}

// n matrices [k][c_in][c_out] back to back -> n packed streams back to back, one launch (the groups of a grouped conv)
extern "C" int v2w_pack_mfma_batch(const float* wf, float* wp, int k, int c_in, int c_out, int u, int n, void* stream) {
    if (n <= 0 || n > 65535) return V2W_E_ARG;
    if (!wf || !wp || k <= 0 || c_in <= 0 || c_out <= 0 || u <= 0) return V2W_E_ARG;
    const LayerCfg cfg = v2w_layer_cfg(c_in, c_out, u);
    if (!cfg.mf) return V2W_E_SHAPE;
    const size_t total = (size_t)k * c_in * c_out;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    V2W_LAUNCH(pack_mfma_kernel, dim3(grid, n), dim3(256), 0, (hipStream_t)stream, wf, wp, k, c_in, c_out, cfg.mf, cfg.ck, u);
    return v2w_launch_status();
}

// Called by v2w_api.hip.  Returns V2W_E_SHAPE when no tile configuration fits (caller falls back to the direct kernel).
// n problems (1 <= n <= V2W_MAX_MULTI) that share B, C_in, C_out, L - hence the tile configuration - in ONE launch.
int v2w_conv1d_mfma(const v2w_conv1d_args* a, int n, hipStream_t stream, int* cfg_out, long long* ws_query) {
    if (n < 1 || n > V2W_MAX_MULTI) return V2W_E_ARG;
    const LayerCfg cfg = v2w_layer_cfg(a->C_in, a->C_out, 1);
    if (!cfg.mf) return V2W_E_SHAPE;
    TileArgs ps[V2W_MAX_MULTI];
    long tiles128 = 0;
    for (int i = 0; i < n; ++i) {
        const v2w_conv1d_args* q = a + i;
        if (q->B != a->B || q->C_in != a->C_in || q->C_out != a->C_out || q->L != a->L) return V2W_E_SHAPE;
        if (!q->wp && !cfg_out && !ws_query) return V2W_E_SHAPE;
        TileArgs p{};
        p.cfg_out = cfg_out;
        p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.wp = q->wp; p.bias = q->bias;
        p.res = q->res; p.res_a = q->res_a; p.res_s = q->res_s; p.out = q->out;
        p.add0 = q->add0; p.add1 = q->add1;
        p.mask_src = q->mask_src; p.mask_a = q->mask_a; p.mask_s = q->mask_s; p.mask_slope = q->mask_slope;
        p.stats_part = q->rowsum_part;
        p.B = q->B; p.Cin = q->C_in; p.Cout = q->C_out; p.L = q->L; p.K = q->k; p.dil = q->dil;
        p.CinT = q->in_ct > 0 ? q->in_ct : q->C_in; p.CoutT = q->out_ct > 0 ? q->out_ct : q->C_out;
        p.out_slope = q->out_slope > 0.f ? q->out_slope : 1.f;
        p.pad = 0; p.hl = p.hr = q->dil * (q->k - 1) / 2;
        if (q->pad_left >= 0) { p.hl = q->pad_left; p.hr = q->dil * (q->k - 1) - q->pad_left; if (p.hr < 0) return V2W_E_ARG; }
        p.in_stride = q->in_stride > 0 ? q->in_stride : 1; p.in_phase = q->in_phase;
        p.slope = q->slope; p.accumulate = q->accumulate; p.out_div = q->out_div;
        p.splitk_ws = a->splitk_ws; p.splitk_ws_bytes = a->splitk_ws ? a->splitk_ws_bytes : 0; p.ws_query = ws_query;
        ps[i] = p;
        tiles128 += (long)p.B * ((p.L + 127) / 128) * (p.Cout / 128);
    }
    int halo = 0;
    for (int i = 0; i < n; ++i) { const int h = ((ps[i].hl + 3) & ~3) > ps[i].hr ? ((ps[i].hl + 3) & ~3) : ps[i].hr; if (h > halo) halo = h; }
    if (halo > 32) {            // wide-halo variants (see launch_tile): dense layers of the period discriminators
        if (halo > 48 || cfg.mf != 32 || cfg.ck != 32 || a->C_out % 64 != 0) return V2W_E_SHAPE;
        if (a->C_out % 128 == 0 && tiles128 >= 4 * 256) return launch_tile<32, 1, 2, 2, 2, 2, 32, 48>(ps, n, stream);
        return launch_tile<32, 1, 1, 2, 2, 2, 32, 48>(ps, n, stream);
    }
    // latency-bound sizes (inference at B = 1): fewer than two 64 x 64 tiles per CU -> 64 x 64 tiles: the most workgroups and one
    // MFMA per k-step and wave, i.e. the shortest serial chain through the K loop (cfg1: 1.54 -> 1.23 ms per forward)
    if (cfg.mf == 32 && cfg.ck == 32 && a->C_out % 64 == 0 && (long)n * a->B * ((a->L + 63) / 64) * (a->C_out / 64) < 512)
        return launch_tile<32, 1, 1, 1, 2, 2, 32>(ps, n, stream);
    if (cfg.mf == 32 && cfg.ck == 32 && a->C_out % 128 == 0) {
        // 128 x 128 tiles unless that leaves fewer than ~4 tiles per CU: then 128 x 64 halves the tail imbalance
        if (tiles128 >= 4 * 256) return launch_tile<32, 1, 2, 2, 2, 2, 32>(ps, n, stream);
        return launch_tile<32, 1, 2, 1, 2, 2, 32>(ps, n, stream);
    }
    // (64 x 256 and 64 x 128 single-row-block variants measured slower on MI355X: 52-92 / 61-90 vs 64-93 TF)
    if (cfg.mf == 32 && cfg.ck == 32 && a->C_out % 64 == 0) return launch_tile<32, 1, 1, 2, 2, 2, 32>(ps, n, stream);
    if (cfg.mf == 32 && cfg.ck == 16) return launch_tile<32, 1, 1, 2, 1, 4, 16>(ps, n, stream);
    if (cfg.mf == 32) return launch_tile<32, 1, 1, 2, 1, 4, 32>(ps, n, stream);
    if (cfg.mf == 16) return launch_tile<16, 1, 1, 4, 1, 4, 16>(ps, n, stream);
    return V2W_E_SHAPE;
}

int v2w_convt1d_mfma(const v2w_convt1d_args* a, hipStream_t stream, int* cfg_out, long long* ws_query) {
    const LayerCfg cfg = v2w_layer_cfg(a->C_in, a->C_out, a->u);
    if ((!a->wp && !cfg_out && !ws_query) || !cfg.mf) return V2W_E_SHAPE;
    TileArgs p{};
    p.cfg_out = cfg_out;
    p.splitk_ws = a->splitk_ws; p.splitk_ws_bytes = a->splitk_ws ? a->splitk_ws_bytes : 0; p.ws_query = ws_query;
    p.in = a->in; p.wp = a->wp; p.bias = a->bias; p.out = a->out; p.stats_part = a->stats_part;
    p.B = a->B; p.Cin = a->C_in; p.Cout = a->C_out; p.L = a->L; p.K = a->k; p.dil = 1;
    p.CinT = p.Cin; p.CoutT = p.Cout; p.out_slope = 1.f;
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

// ---- batched fold + pack (see include/vec2wav_hip.h)
extern "C" int v2w_fold_plan(v2w_fold_desc* descs, int n, int32_t* starts) {
    if (!descs || !starts || n <= 0) return V2W_E_ARG;
    int bs = 0, bp = 0, lds = 0;
    for (int i = 0; i < n; ++i) {
        v2w_fold_desc& d = descs[i];
        if (d.c_in <= 0 || d.c_out <= 0 || d.k <= 0 || d.u <= 0) return V2W_E_ARG;
        const LayerCfg cfg = v2w_layer_cfg(d.c_in, d.c_out, d.transposed ? d.u : 1);
        if (!cfg.mf) return V2W_E_SHAPE;
        d.mf = cfg.mf; d.ck = cfg.ck;
        if (d.wpd && (d.transposed || d.c_in != d.c_out || cfg.mf != cfg.ck)) return V2W_E_ARG;   // the gradient stream comes from the same LDS block only then
        starts[i] = bs; starts[n + 1 + i] = bp;
        bs += d.transposed ? d.c_in : d.c_out;
        bp += (d.c_out / cfg.mf) * (d.c_in / cfg.ck);
        const int nrow = d.transposed ? cfg.ck : cfg.mf, rlen = (d.transposed ? cfg.mf : cfg.ck) * d.k;
        const int bytes = nrow * (rlen + 1) * (int)sizeof(float);
        if (bytes > lds) lds = bytes;
    }
    starts[n] = bs; starts[2 * n + 1] = bp;
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    return lds;   // dynamic LDS bytes the pack kernel needs (> 0)
}

extern "C" int v2w_fold_pack_batch(const v2w_fold_desc* descs_dev, const int32_t* starts_dev, int n,
                                   int nblk_scale, int nblk_pack, int lds_bytes, void* stream) {
    if (!descs_dev || !starts_dev || n <= 0 || nblk_scale <= 0 || nblk_pack <= 0 || lds_bytes <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(fold_pack_batch_kernel), lds_bytes, st);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(fold_scale_batch_kernel, dim3(nblk_scale), dim3(256), 0, st, descs_dev, starts_dev, n);
    V2W_LAUNCH(fold_pack_batch_kernel, dim3(nblk_pack), dim3(256), lds_bytes, st, descs_dev, starts_dev + n + 1, n);
    return v2w_launch_status();
}
