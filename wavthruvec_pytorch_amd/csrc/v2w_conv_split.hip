// Conv1d of the generator on the gfx950 f16 matrix pipe with fp32-equivalent products ("split f16x3").
//
//   out[b,co,l] = bias[co] + sum_{ci,t} W[t][ci][co] * act(in[b,ci,l + (t-(k-1)/2)*dil])      (models.py:37-44, 65-70, 123)
//
// Every fp32 operand is split into two halves-precision parts  x = x_hi + x_lo  (x_hi = rne_f16(x), x_lo = rne_f16(x - x_hi);
// weights are first scaled by a per-layer power of two so that both parts sit in the normal f16 range) and the product is
// accumulated in fp32 as  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi  - three v_mfma_f32_32x32x16_f16.  x_hi*w_hi is exact in fp32
// (11 x 11 bits), the dropped x_lo*w_lo term is <= 2^-22 |x w|: the result carries ~22 bits per product against fp32's 24,
// i.e. it differs from the exact-fp32 kernel by about as much as two fp32 summation orders differ from each other
// (measured through the whole generator: 1e-7, parity bar 1e-4).  The f16 pipe runs 16x the f32 MFMA rate, so three passes
// still leave >5x headroom: these layers stop being MFMA-bound and approach their HBM / LDS bounds.
//
// GEMM view per workgroup: M = MT output channels, N = NT positions, K = C_in x taps, consumed as (chunk of 32 channels) x tap.
// B (signal) : LDS tile [position][32 ch hi | 32 ch lo | pad] (144 B rows: conflict-free ds_read_b128 of 8 channels per
//              lane), double buffered over chunks; the fp32 -> (hi, lo) split, the CondBN affine, leaky_relu and zero
//              padding happen once at staging; a tap is a ROW offset into the same tile.
// A (weights): v2w_pack_split() stores, per 32-row block, chunk and tap, the four MFMA A fragments (2 k-steps x hi/lo,
//              1 KiB each) contiguously; a stage (= all row blocks of the workgroup for one chunk x tap) is copied
//              global -> LDS by global_load_lds_dwordx4 (no registers), two stages ahead of its use.
// Sync       : one workgroup barrier per tap; the async copies are fenced with explicit s_waitcnt vmcnt (see the loop).
#include "v2w_tile.h"


namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef unsigned int raw16 __attribute__((ext_vector_type(4)));   // one 16-byte MFMA operand fragment, type-agnostic

#define V2W_SPLIT_CK 16        // input channels per stage = one MFMA k-step
#define V2W_SPLIT_ROWB 80       // bytes per staged position: 16 ch hi (32 B) | 16 ch lo (32 B) | 16 B pad
#define V2W_SPLIT_UNIT 2048     // bytes of the A fragments of one (32-row block, chunk, tap): [hi, lo][64 lanes][16 B]
#define V2W_SPLIT_HMAX 32       // largest halo per side
#define V2W_SPLIT_NAB 4         // weight stages resident in LDS: one computing, one published for the next stage, two in flight
#define V2W_SPLIT_WPE 2         // waves per SIMD the register allocation targets (= workgroups per CU)
#define V2W_SPLIT_C64 0
#define V2W_SPLIT_FORCE 0       // experiments: 1 = 64 x 128 tiles, 2 = 128 x 128 tiles for every C_out % 128 == 0 layer

#define V2W_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define V2W_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// BF = false: split f16x3 (three MFMAs per product).  BF = true: plain bf16 operands, ONE v_mfma_f32_32x32x16_bf16 per product
// (BASELINE configs[2] 'bf16 compute / fp32 accumulate'): same tiles, same fragment layout with the lo halves unused.
// HM: the largest halo per side the staging slots cover (V2W_SPLIT_HMAX for the generator; 40 for the discriminators' dilation = period convs:
// 2 x 17 / 2 x 19 positions)
template <int MI, int NI, int WM, int WN, bool VEC, bool BF, int HM = V2W_SPLIT_HMAX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(V2W_SPLIT_WPE, V2W_SPLIT_WPE)))
conv_split_kernel(const MultiArgs m) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    static_assert(WM * WN == 4, "four waves: the staging maps one wave to 4 of the 16 chunk channels");
    constexpr int NTHREADS = 256;
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, CK = V2W_SPLIT_CK;
    constexpr int ROWB = V2W_SPLIT_ROWB;
    constexpr int NMT = MT / 32;                                // 32-row blocks per workgroup
    constexpr int ASTAGE = NMT * V2W_SPLIT_UNIT;                // bytes of one weight stage
    constexpr int NAB = V2W_SPLIT_NAB;
    // async 16-B copies per thread and stage; bf16 moves only the hi half of every unit (1 KiB = one wave copy per unit)
    constexpr int ADMA = BF ? (NMT + 3) / 4 : ASTAGE / (NTHREADS * 16);
    static_assert(ASTAGE % (NTHREADS * 16) == 0 && ADMA >= 1, "a stage is a whole number of workgroup copies");
    constexpr int NS = ((NT + 2 * HM) / 4 + 63) / 64;    // position groups (4 positions) per lane
    constexpr int NSIG = (CK * ((NT + 2 * HM) / 4) + NTHREADS - 1) / NTHREADS;   // async copies per thread of one raw signal chunk
    constexpr int RS = 72;                                      // floats per row of the epilogue transpose tile (32 x 64 per pass; 4*RS % 64 == 32)
    static_assert(NI % 2 == 0, "the epilogue works on pairs of 32-column blocks");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int pq = 0;
#pragma unroll
    for (int i = 1; i < V2W_MAX_MULTI; ++i) pq += (int)blockIdx.x >= m.start[i] ? 1 : 0;
    const TileArgs& p = m.p[pq];
    const int mtiles = p.Cout / MT;
    const int id = blockIdx.x - m.start[pq];
    const int grp = id / (8 * mtiles), rem = id % (8 * mtiles);
    const int mt = rem >> 3;
    const int tile = grp * 8 + (rem & 7);
    if (tile >= p.ntiles) return;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * NT;
    const int m0 = mt * MT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, hk = lane >> 5;
    const int wmi = wave / WN;                                   // row-block group of this wave
    const int wm0 = wmi * (32 * MI);
    const int wn0 = (wave % WN) * (32 * NI);
    const int L = p.L, K = p.K;
    const float slope = p.slope;
    const int nch = p.Cin / CK;
    const int nst = nch * K;                                     // (chunk, tap) stages = units per row block, contiguous in wps
    const int pos0 = n0 - p.hla;                                 // position of LDS row 0
    const int xbytes = p.xcols * ROWB;
    unsigned char* const Xs0 = smem;
    unsigned char* const As0 = smem + 2 * xbytes;
    unsigned char* const Rs = As0 + NAB * ASTAGE;                // raw fp32 image of the next chunk: [16 ch][xcols], filled by LDS-DMA
    float* const atab = reinterpret_cast<float*>(smem + p.atab_off);   // a[Cin], s[Cin] of this batch item
    // epilogue constants bias, res_a, res_s, mask_a, mask_s [MT] each: filled AFTER the stage loop, behind the transpose tiles, in
    // LDS the stage buffers no longer need (keeps the 64 x 256 tile at two workgroups per CU)
    float* const etab = reinterpret_cast<float*>(smem) + 4 * 32 * RS;

    acc_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- weight stages: async global -> LDS.  Copy #i of a thread moves 16 B, a wave instruction fills 1 KiB of LDS; the
    // (chunk, tap) units of a row block are contiguous, so the source of the next stage is simply + one unit, and the ring
    // slot advances by one: all bookkeeping is a few scalar adds per stage.
    const unsigned char* dsrc[ADMA];
#pragma unroll
    for (int i = 0; i < ADMA; ++i) {
        const int q = i * NTHREADS + tid;                        // 16-B element of the stage; 128 of them per unit (bf16: the first 64)
        const int unit = BF ? (q >> 6) % NMT : q >> 7, off = BF ? (q & 63) : (q & 127);     // bf16, NMT < 4: waves 2, 3 repeat 0, 1
        dsrc[i] = reinterpret_cast<const unsigned char*>(p.wps) + (size_t)(m0 / 32 + unit) * nst * V2W_SPLIT_UNIT + off * 16;
    }
    const unsigned a_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(As0 + wave * 1024));
    const unsigned a_lds_bf = __builtin_amdgcn_readfirstlane(    // bf16: wave w fills the hi half of unit (w % NMT)
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(As0 + (wave % NMT) * V2W_SPLIT_UNIT));
    int dma_slot = 0;                                            // ring slot the next copy fills
    auto dma_next = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < ADMA; ++i) {
            // Issued through inline asm ON PURPOSE: hipcc orders every later LDS read behind a builtin LDS-DMA with
            // s_waitcnt vmcnt(0) (it cannot prove the buffers distinct), which would serialise copy and compute.  Hidden
            // from its bookkeeping, the copy only makes the compiler's own vmcnt waits more conservative (vmcnt retires in
            // order); the waits THIS data needs are the explicit V2W_WAIT_VM below.  M0 = LDS byte address of lane 0.
            const unsigned lds = BF ? a_lds_bf + dma_slot * ASTAGE + i * (4 * V2W_SPLIT_UNIT)
                                    : a_lds + dma_slot * ASTAGE + i * (NTHREADS * 16);
            unsigned m0_save;                                    // M0 is a reserved register: hand it back as found
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_save) : "s"(lds), "v"(dsrc[i]) : "memory");
            dsrc[i] += V2W_SPLIT_UNIT;
        }
        dma_slot = dma_slot + 1 == NAB ? 0 : dma_slot + 1;
    };

    const int xp4 = p.xcols >> 2;
    // ---- signal staging: wave w owns channels 4w..4w+3 of the chunk, lane l the position groups l, l+64, ... (1 KiB
    // contiguous per row and load instruction); every thread issues exactly NSIG loads (addresses clamped).
    // The next chunk's fp32 signal also travels global -> LDS asynchronously (same hidden-from-hipcc copies as the weights:
    // with no compiler-visible VMEM load in the loop, hipcc adds no vmcnt waits of its own - a register prefetch made it wait
    // for the freshly issued weight copies at every commit, ~2900 cycles per chunk).  Float4 #lin of the image [16][xp4]
    // is copied by thread (lin % 256) in instruction lin / 256; global offsets are fixed per thread, only the chunk base moves.
    // The region holds exactly the image rounded up to 1 KiB (one wave copy); the wave copies of the last instruction that
    // would run past it are pulled back to end flush with it (they re-copy the same data to the same place).
    const int rawb = (CK * xp4 * 16 + 1023) & ~1023;
    int soff[NSIG], rbase[NSIG];
#pragma unroll
    for (int i = 0; i < NSIG; ++i) {
        int wb = i * (NTHREADS * 16) + wave * 1024;              // LDS byte offset of this wave's 1 KiB in instruction i
        wb = wb > rawb - 1024 ? rawb - 1024 : wb;
        rbase[i] = wb;
        const int lin = (wb >> 4) + lane;
        int row = lin / xp4, c4 = lin - row * xp4;
        if (row > CK - 1) { row = CK - 1; c4 = 0; }              // beyond the image: a harmless duplicate copy
        int pos = pos0 + c4 * 4;
        pos = pos < 0 ? 0 : (pos > L - 4 ? L - 4 : pos);          // out-of-sequence columns are zeroed at commit
        soff[i] = row * L + pos;
    }
    const unsigned r_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Rs);
    auto prefetch = [&](int ci0) __attribute__((always_inline)) {
        const float* src = p.in + (size_t)(b * p.CinT + ci0) * L;      // (CinT: channels of the tensor `in` points into - a group's slice)
#pragma unroll
        for (int i = 0; i < NSIG; ++i) {
            unsigned m0_save;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_save) : "s"(r_lds + __builtin_amdgcn_readfirstlane(rbase[i])), "v"(src + soff[i]) : "memory");
        }
    };
    auto act = [&](float v) __attribute__((always_inline)) {     // leaky_relu, then into the f16 range (see the header)
        v = slope <= 1.f ? fmaxf(v, v * slope) : v2w_lrelu(v, slope);
        return BF ? v : __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
    };
    auto commit = [&](int ci0, unsigned char* Xs) __attribute__((always_inline)) {
        float av[4], sv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            av[c] = p.in_a ? atab[ci0 + wave * 4 + c] : 1.f;
            sv[c] = p.in_a ? atab[p.Cin + ci0 + wave * 4 + c] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int pg = lane + s * 64;
            if (pg >= xp4) continue;
            const int pos = pos0 + pg * 4;
            const bool in_seq = pos >= 0 && pos < L;             // L % 4 == 0, pos % 4 == 0: whole float4 in or out
            f32x4 pf[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) pf[c] = *reinterpret_cast<const f32x4*>(Rs + ((wave * 4 + c) * xp4 + pg) * 16);
            // rows 4 apart alias 4-way in LDS (320 B = 16 banks mod 64): 8 stores per chunk and wave, not worth a rotation
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned char* d = Xs + (pg * 4 + e) * ROWB + wave * 8;
                if constexpr (BF) {
                    b4 hi;
#pragma unroll
                    for (int c = 0; c < 4; ++c) hi[c] = (__bf16)(in_seq ? act(fmaf(av[c], pf[c][e], sv[c])) : 0.f);
                    *reinterpret_cast<b4*>(d) = hi;
                } else {
                    h4 hi, lo;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = in_seq ? act(fmaf(av[c], pf[c][e], sv[c])) : 0.f;   // padding of the ACTIVATED signal
                        const _Float16 h = (_Float16)v;
                        hi[c] = h;
                        lo[c] = (_Float16)(v - (float)h);
                    }
                    *reinterpret_cast<h4*>(d) = hi;
                    *reinterpret_cast<h4*>(d + 32) = lo;
                }
            }
        }
    };
    auto stage_scalar = [&](int ci0, unsigned char* Xs) __attribute__((always_inline)) {        // any L / alignment / input stride
        for (int c = wave; c < CK; c += 4) {
            const int ch = b * p.Cin + ci0 + c;
            const float* src = p.in + (size_t)(b * p.CinT + ci0 + c) * L * p.in_stride + p.in_phase;
            const float av = p.in_a ? p.in_a[ch] : 1.f;
            const float sv = p.in_s ? p.in_s[ch] : 0.f;
            for (int j = lane; j < p.xcols; j += 64) {
                const int l = pos0 + j;
                float v = 0.f;
                if (l >= 0 && l < L) v = act(fmaf(av, src[(size_t)l * p.in_stride], sv));
                if constexpr (BF) {
                    reinterpret_cast<__bf16*>(Xs + j * ROWB)[c] = (__bf16)v;
                } else {
                    const _Float16 h = (_Float16)v;
                    _Float16* d = reinterpret_cast<_Float16*>(Xs + j * ROWB) + c;
                    d[0] = h;
                    d[16] = (_Float16)(v - (float)h);
                }
            }
        }
    };

    // ---- prologue
    if (p.in_a) {
        for (int c = tid; c < p.Cin; c += NTHREADS) {
            atab[c] = p.in_a[b * p.Cin + c];
            atab[p.Cin + c] = p.in_s[b * p.Cin + c];
        }
    }
    const float winv = p.winv[0];
    __syncthreads();
    for (int s0 = 0; s0 < NAB - 1; ++s0)
        if (s0 < nst) dma_next();
    if constexpr (VEC) {
        prefetch(0);
        V2W_WAIT_VM(0);
        V2W_BARRIER();                                           // raw chunk 0 (and the first weight stages) landed
        commit(0, Xs0);
    } else {
        stage_scalar(0, Xs0);
        V2W_WAIT_VM(0);
    }
    V2W_BARRIER();

    const int rowbase = wn0 + lr + p.hla - p.hl;                 // LDS row of this lane's column for tap 0
    int st = 0;
    // One stage = one (chunk, tap): 3 * MI * NI MFMAs per wave between two workgroup barriers.
    // Issue order inside a stage: the weight copy of stage st+NAB-1, then (first tap of a chunk) the signal prefetch of the
    // next chunk.  vmcnt retires in order; stage st+1 was issued NAB-2 stages ago, so at the end of the stage
    //   SIG stage : vmcnt((NAB-2)*ADMA + NSIG) leaves the NAB-2 younger copies and this stage's prefetch outstanding;
    //   otherwise : vmcnt((NAB-2)*ADMA) leaves the NAB-2 younger copies outstanding (a prefetch older than that has landed);
    //   tail      : once no copy is issued any more the counts no longer hold -> vmcnt(0).
    // SIG / COMMIT are compile-time so that hipcc's own (path-insensitive) vmcnt bookkeeping sees prefetch -> commit as a
    // straight line and adds no waits of its own inside the tap loop.
    // B (signal) fragments of a stage come from the chunk's tile, which does not change between the taps of a chunk: they
    // are read for tap t+1 BEFORE the barrier that ends tap t (their LDS latency hides under this tap's MFMAs); only the
    // A fragments, which the barrier publishes, are read after it.
    raw16 bh[NI], bl[BF ? 1 : NI];
    auto mma = [&](acc_t c, raw16 a, raw16 b) __attribute__((always_inline)) {
        if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    };
    auto read_b = [&](const unsigned char* Xs, int t) __attribute__((always_inline)) {
        const unsigned char* xr = Xs + (rowbase + t * p.dil) * ROWB + hk * 16;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            bh[j] = *reinterpret_cast<const raw16*>(xr + j * 32 * ROWB);
            if constexpr (!BF) bl[j] = *reinterpret_cast<const raw16*>(xr + j * 32 * ROWB + 32);
        }
    };
    // ---- software pipeline of the stage loop.  A weight stage is PUBLISHED (copy landed + workgroup barrier) two stages
    // before its use, so the A fragments of stage st+1 are read from LDS while stage st computes, like the B fragments of the
    // next tap: a stage starts its MFMAs right after the barrier with every operand already in registers.  The A fragments
    // alternate between two register sets (A0 / A1); K is odd, so a chunk starts and ends on the same set and the next
    // chunk starts on the other one: the two chunk bodies below keep every register index static.
    int ring = 0;                                                // ring slot of the stage being computed
    auto read_a = [&](raw16 (&h)[MI], raw16 (&l)[BF ? 1 : MI], int slot) __attribute__((always_inline)) {
        const unsigned char* Ab = As0 + slot * ASTAGE + (wmi * MI) * V2W_SPLIT_UNIT + lane * 16;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            h[i] = *reinterpret_cast<const raw16*>(Ab + i * V2W_SPLIT_UNIT);
            if constexpr (!BF) l[i] = *reinterpret_cast<const raw16*>(Ab + i * V2W_SPLIT_UNIT + 1024);
        }
    };
    auto stage = [&](raw16 (&ah)[MI], raw16 (&al)[BF ? 1 : MI], raw16 (&nh)[MI], raw16 (&nl)[BF ? 1 : MI],
                     const bool SIG, const bool COMMIT, int ch, int t) __attribute__((always_inline)) {
        const unsigned char* Xs = Xs0 + (ch & 1) * xbytes;
        const bool more = ch + 1 < nch;
        const bool dma = st + (NAB - 1) < nst;
        if (dma) dma_next();
        const bool sig = SIG && VEC && more;
        if (sig) prefetch((ch + 1) * CK);
        __builtin_amdgcn_sched_barrier(0);

        if (SIG) read_b(Xs, t);                                  // first tap of a chunk: its tile was committed just before the barrier
        const int nslot = ring + 1 == NAB ? 0 : ring + 1;
        if (st + 1 < nst) read_a(nh, nl, nslot);                 // next stage's weights: published by the previous barrier
        ring = nslot;
        __builtin_amdgcn_sched_barrier(0);
        // term order al*bh, ah*bh, ah*bl: bh has its last use after the second group and bl after the third, so the next tap's
        // fragments are read into the SAME registers right there (no copies) and their LDS latency hides under the
        // remaining MFMAs, the wait and the barrier
        const unsigned char* xn = Xs + (rowbase + (t + 1) * p.dil) * ROWB + hk * 16;
        if constexpr (!BF) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mma(acc[i][j], al[i], bh[j]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = mma(acc[i][j], ah[i], bh[j]);
        if (!COMMIT) {
#pragma unroll
            for (int j = 0; j < NI; ++j) bh[j] = *reinterpret_cast<const raw16*>(xn + j * 32 * ROWB);
        }
        if constexpr (!BF) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = mma(acc[i][j], ah[i], bl[j]);
            if (!COMMIT) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bl[j] = *reinterpret_cast<const raw16*>(xn + j * 32 * ROWB + 32);
            }
        }

        if (COMMIT && more) {
            unsigned char* Xn = Xs0 + ((ch + 1) & 1) * xbytes;
            if constexpr (VEC) commit((ch + 1) * CK, Xn);        // K >= 3: the tap-1 wait already covered the raw chunk
            else stage_scalar((ch + 1) * CK, Xn);
        }
        // vmcnt retires in order.  The stage two ahead must have landed before this barrier publishes it: it was issued a stage
        // ago, so only what THIS stage issued may stay outstanding (its copy, and the signal copies of a SIG stage; an older
        // signal copy has landed by then as well).  Once no copy is issued any more the counts no longer hold: vmcnt(0).
        if (!dma) V2W_WAIT_VM(0);
        else if (sig) V2W_WAIT_VM(ADMA + NSIG);
        else V2W_WAIT_VM(ADMA);
        // LDS reads in flight (next operands) need not drain before the barrier; the LDS WRITES of a commit must
        if (COMMIT) V2W_BARRIER(); else asm volatile("s_barrier" ::: "memory");
        ++st;
    };
    raw16 a0h[MI], a0l[BF ? 1 : MI], a1h[MI], a1l[BF ? 1 : MI];
    read_a(a0h, a0l, 0);
    auto chunk = [&](raw16 (&xh)[MI], raw16 (&xl)[BF ? 1 : MI], raw16 (&yh)[MI], raw16 (&yl)[BF ? 1 : MI], int ch) __attribute__((always_inline)) {
        stage(xh, xl, yh, yl, true, false, ch, 0);
        int t = 1;
        for (; t + 1 < K - 1; t += 2) {
            stage(yh, yl, xh, xl, false, false, ch, t);
            stage(xh, xl, yh, yl, false, false, ch, t + 1);
        }
        stage(yh, yl, xh, xl, false, false, ch, K - 2);
        stage(xh, xl, yh, yl, false, true, ch, K - 1);
    };
    for (int ch = 0; ch < nch; ch += 2) {
        chunk(a0h, a0l, a1h, a1l, ch);
        if (ch + 1 < nch) chunk(a1h, a1l, a0h, a0l, ch + 1);
    }

    // ---- epilogue (same contract as the f32 tile kernel): undo the weight scale, mask, + bias [+ residual] [+ addends]
    // [/ out_div].  The accumulators hold 4 consecutive ROWS per lane; each wave transposes 32 x (32*NI) blocks through its
    // own LDS region (all stages are consumed: the tile buffers are free) so that global traffic is float4 along positions.
    // (the last stage ended with a workgroup barrier: every wave is done with the stage buffers)
    for (int c = tid; c < MT; c += NTHREADS) {
        etab[c] = p.bias ? p.bias[m0 + c] : 0.f;
        etab[MT + c] = p.res_a ? p.res_a[b * p.Cout + m0 + c] : 1.f;
        etab[2 * MT + c] = p.res_a ? p.res_s[b * p.Cout + m0 + c] : 0.f;
        etab[3 * MT + c] = p.mask_a ? p.mask_a[b * p.Cout + m0 + c] : 1.f;
        etab[4 * MT + c] = p.mask_a ? p.mask_s[b * p.Cout + m0 + c] : 0.f;
    }
    __syncthreads();
    float* const T = reinterpret_cast<float*>(smem) + wave * (32 * RS);
    // one pass = 32 rows x 64 positions (two accumulator blocks): bounded registers for any NI.  The passes are spelled out
    // with STATIC accumulator indices below: left as a loop hipcc does not always unroll it, indexes `acc` dynamically and
    // parks the whole accumulator array in scratch.
    auto epass = [&](const acc_t& c0, const acc_t& c1, const int i, const int j0) __attribute__((always_inline)) {
        {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                T[F::row(e, hk) * RS + lr] = c0[e] * winv;       // power of two: exact
                T[F::row(e, hk) * RS + 32 + lr] = c1[e] * winv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private region: no workgroup barrier needed
            constexpr int C4 = 16;                               // float4 per row of the pass
            constexpr int NIT = 32 * C4 / 64;                    // float4 per lane
            constexpr int EG = MI * NI >= 8 ? 2 : 4;             // float4 gathers in flight per operand (register budget)
#pragma unroll
            for (int g0 = 0; g0 < NIT; g0 += EG) {
                f32x4 rv[EG], ov[EG], o2[EG], mv[EG];
#pragma unroll
                for (int g = 0; g < EG; ++g) {
                    const int f = (g0 + g) * 64 + lane;
                    const int row = f / C4, q = n0 + wn0 + j0 * 32 + (f % C4) * 4;
                    const size_t o = ((size_t)b * p.CoutT + m0 + wm0 + i * 32 + row) * L + q;
                    const bool in = p.evec && q < L;
                    rv[g] = (p.res && in) ? *reinterpret_cast<const f32x4*>(p.res + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    ov[g] = (p.accumulate && in) ? *reinterpret_cast<const f32x4*>(p.out + o)
                                                 : ((p.add0 && in) ? *reinterpret_cast<const f32x4*>(p.add0 + o) : f32x4{0.f, 0.f, 0.f, 0.f});
                    o2[g] = (p.add1 && in) ? *reinterpret_cast<const f32x4*>(p.add1 + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    mv[g] = (p.mask_src && in) ? *reinterpret_cast<const f32x4*>(p.mask_src + o) : f32x4{1.f, 1.f, 1.f, 1.f};
                }
#pragma unroll
                for (int g = 0; g < EG; ++g) {
                    const int f = (g0 + g) * 64 + lane;
                    const int row = f / C4, c4 = f % C4, q = n0 + wn0 + j0 * 32 + c4 * 4;
                    if (q >= L) continue;
                    const int col = wm0 + i * 32 + row;
                    const size_t o = ((size_t)b * p.CoutT + m0 + col) * L + q;
                    const float bias = etab[col], ra = etab[MT + col], rs = etab[2 * MT + col];
                    const float ma = etab[3 * MT + col], ms = etab[4 * MT + col];
                    f32x4 v = *reinterpret_cast<const f32x4*>(T + row * RS + c4 * 4);
                    if (!p.evec) {                               // ragged L / unaligned operands: element-wise tail path
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (q + e >= L) break;
                            float x = v[e];
                            if (p.mask_src) x = fmaf(ma, p.mask_src[o + e], ms) > 0.f ? x : x * p.mask_slope;
                            x += bias;
                            if (p.res) x += fmaf(ra, p.res[o + e], rs);
                            if (p.add1) x += p.add0[o + e] + p.add1[o + e];
                            else if (p.accumulate) x += p.out[o + e];
                            else if (p.add0) x += p.add0[o + e];
                            if (p.out_div != 0.f) x = x / p.out_div;
                            if (p.out_slope != 1.f) x = x > 0.f ? x : x * p.out_slope;
                            p.out[o + e] = x;
                        }
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = v[e];
                        if (p.mask_src) x = fmaf(ma, mv[g][e], ms) > 0.f ? x : x * p.mask_slope;
                        x += bias;
                        if (p.res) x += fmaf(ra, rv[g][e], rs);
                        if (p.add1) x += ov[g][e] + o2[g][e];    // (add0 + add1) + value: the reference's `xs += ...` order
                        else if (p.accumulate || p.add0) x += ov[g][e];
                        if (p.out_div != 0.f) x = x / p.out_div;
                        if (p.out_slope != 1.f) x = x > 0.f ? x : x * p.out_slope;      // (the discriminators' activated feature maps)
                        v[e] = x;
                    }
                    *reinterpret_cast<f32x4*>(p.out + o) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of T done before the next pass overwrites it
        }
    };
    epass(acc[0][0], acc[0][1], 0, 0);
    if constexpr (NI >= 4) epass(acc[0][2], acc[0][3], 0, 2);
    if constexpr (MI >= 2) {
        epass(acc[1][0], acc[1][1], 1, 0);
        if constexpr (NI >= 4) epass(acc[1][2], acc[1][3], 1, 2);
    }
    static_assert(MI <= 2 && NI <= 4, "epilogue passes are spelled out for MI <= 2, NI <= 4");
}

template <int MI, int NI, int WM, int WN, int HM = V2W_SPLIT_HMAX>
int launch_split(const TileArgs* ps, int nprob, hipStream_t stream, bool bf) {
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, NTHREADS = 64 * WM * WN, CK = V2W_SPLIT_CK;
    constexpr int RS = 72;
    if (nprob < 1 || nprob > V2W_MAX_MULTI) return V2W_E_ARG;
    MultiArgs m{};
    size_t lds = 0;
    int grid = 0;
    for (int i = 0; i < nprob; ++i) {
        TileArgs p = ps[i];
        if (p.Cout % MT != 0 || p.Cin % CK != 0 || !p.wps || !p.winv) return V2W_E_SHAPE;
        if (p.K < 3 || (p.K & 1) == 0) return V2W_E_SHAPE;            // the stage pipeline alternates register sets over an odd tap count
        p.hla = (p.hl + 3) & ~3;
        if (p.hla > HM || p.hr > HM) return V2W_E_SHAPE;
        p.ntl = (p.L + NT - 1) / NT;
        p.ntiles = p.B * p.ntl;
        p.xcols = (p.hla + NT + p.hr + 3) & ~3;
        p.xw = 0;
        if (p.in_stride < 1) p.in_stride = 1;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        p.vec4 = (p.L % 4 == 0) && p.L >= 4 && al16(p.in) && p.in_stride == 1;
        p.evec = (p.L % 4 == 0) && al16(p.out) && al16(p.res) && al16(p.add0) && al16(p.add1) && al16(p.mask_src);
        const size_t rawb = ((size_t)CK * (p.xcols / 4) * 16 + 1023) & ~(size_t)1023;
        size_t l = (size_t)2 * p.xcols * V2W_SPLIT_ROWB + (size_t)V2W_SPLIT_NAB * (MT / 32) * V2W_SPLIT_UNIT + rawb;
        const size_t tl = ((size_t)4 * 32 * RS + 5 * MT) * sizeof(float);   // epilogue: transpose tiles + constants overlay the stage buffers
        if (tl > l) l = tl;
        p.atab_off = (int)l;
        l += (p.in_a ? (size_t)2 * p.Cin : 0) * sizeof(float);
        if (l > lds) lds = l;
        m.p[i] = p;
        m.start[i] = grid;
        grid += ((p.ntiles + 7) / 8) * 8 * (p.Cout / MT);
    }
    m.start[nprob] = grid;
    for (int i = nprob + 1; i <= V2W_MAX_MULTI; ++i) m.start[i] = 0x7fffffff;
    // every problem of the launch shares one etab/atab offset (the largest), so that the kernel reads it from its own args
    bool vec = true;                                               // one staging flavour per launch: float4 only if every problem allows it
    for (int i = 0; i < nprob; ++i) vec = vec && m.p[i].vec4;
    for (int i = 0; i < nprob; ++i) m.p[i].vec4 = vec;
    if (bf && HM != V2W_SPLIT_HMAX) return V2W_E_SHAPE;             // (the wide-halo form exists for the split-f16 operands only)
    auto kern = bf ? (vec ? conv_split_kernel<MI, NI, WM, WN, true, true> : conv_split_kernel<MI, NI, WM, WN, false, true>)
                   : (vec ? conv_split_kernel<MI, NI, WM, WN, true, false, HM> : conv_split_kernel<MI, NI, WM, WN, false, false, HM>);
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(NTHREADS), lds, stream, m);
    return v2w_launch_status();
}

// ---- weight preparation: max |w| of the layer -> power-of-two scale -> (hi, lo) half fragments in consumption order
__global__ void __launch_bounds__(256)
split_absmax_kernel(const float* __restrict__ wf, size_t n, unsigned int* __restrict__ amax_bits) {
    __shared__ float red[16];
    float mx = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(wf[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(amax_bits, __float_as_uint(mx));              // non-negative floats order like their bit patterns: deterministic
    }
}

// scale = 2^(13 - exponent(max |w|)): the largest weight lands in [8192, 16384); sc[0] = 1/scale (epilogue), sc[1] = scale
__device__ __forceinline__ float split_scale_from_bits(unsigned int bits) {
    const float mx = __uint_as_float(bits);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.f;
    int ex;
    frexpf(mx, &ex);                                            // mx = f * 2^ex, f in [0.5, 1)
    return ldexpf(1.f, 14 - ex);
}

// wps 16-B element o = (((mb*nch + ch)*K + t)*2 + hl)*64 + lane  (nch = C_in / 16 chunks) holds, for j = 0..7,
//   part_hl( scale * wf[t][ch*16 + 8*(lane>>5) + j][mb*32 + (lane&31)] )
__global__ void __launch_bounds__(256)
pack_split_kernel(const float* __restrict__ wf, h8* __restrict__ wps, const unsigned int* __restrict__ amax_bits,
                  float* __restrict__ sc, int K, int Cin, int Cout) {
    const float scale = split_scale_from_bits(amax_bits[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[0] = 1.f / scale; sc[1] = scale; }
    const int nch = Cin / V2W_SPLIT_CK;
    const size_t total = (size_t)((Cout + 31) / 32) * nch * K * 64;     // lanes; each writes its hi and lo fragment element
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int lane = o & 63;
        size_t rest = o >> 6;
        const int t = rest % K; rest /= K;
        const int ch = rest % nch;
        const int mb = rest / nch;
        const int co = mb * 32 + (lane & 31);
        const int c0 = ch * V2W_SPLIT_CK + 8 * (lane >> 5);
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = co < Cout ? wf[((size_t)t * Cin + c0 + j) * Cout + co] * scale : 0.f;   // rows past C_out: MFMA padding
            const _Float16 h = (_Float16)v;
            hi[j] = h;
            lo[j] = (_Float16)(v - (float)h);
        }
        const size_t base = (((size_t)(mb * nch + ch) * K + t) * 2) * 64 + lane;
        wps[base] = hi;
        wps[base + 64] = lo;
    }
}

// bf16 operands (V2W_ALGO_BF16): same unit layout, hi slot = rne_bf16(w), lo slot unused; no scaling (bf16 has fp32's range)
__global__ void __launch_bounds__(256)
pack_bf16_kernel(const float* __restrict__ wf, b8* __restrict__ wps, float* __restrict__ sc, int K, int Cin, int Cout) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[0] = 1.f; sc[1] = 1.f; }
    const int nch = Cin / V2W_SPLIT_CK;
    const size_t total = (size_t)((Cout + 31) / 32) * nch * K * 64;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int lane = o & 63;
        size_t rest = o >> 6;
        const int t = rest % K; rest /= K;
        const int ch = rest % nch;
        const int mb = rest / nch;
        const int co = mb * 32 + (lane & 31);
        const int c0 = ch * V2W_SPLIT_CK + 8 * (lane >> 5);
        b8 hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) hi[j] = (__bf16)(co < Cout ? wf[((size_t)t * Cin + c0 + j) * Cout + co] : 0.f);
        wps[(((size_t)(mb * nch + ch) * K + t) * 2) * 64 + lane] = hi;
    }
}

// ---- batched weight-norm fold + split pack: every split layer of the generator in three launches driven by a device
// descriptor table (mirrors v2w_fold_pack_batch of the f32 path).
__device__ __forceinline__ int split_find_layer(const int32_t* __restrict__ starts, int n, int blk) {
    int lo = 0, hi = n;                       // starts[li] <= blk < starts[li+1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= blk) lo = mid; else hi = mid; }
    return lo;
}

__global__ void split_zero_batch_kernel(const v2w_split_desc* __restrict__ descs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<unsigned int*>(descs[i].sc)[2] = 0u;
}

// one block per (layer, output row): rowscale[row] = g[row] / ||v[row,:]|| (1 when g == NULL) and the layer's max |w|
__global__ void __launch_bounds__(256)
split_rowscale_batch_kernel(const v2w_split_desc* __restrict__ descs, const int32_t* __restrict__ starts, int n) {
    __shared__ double red[16];
    __shared__ float redm[4];
    const int li = split_find_layer(starts, n, blockIdx.x);
    const v2w_split_desc d = descs[li];
    const int row = blockIdx.x - starts[li];
    const int inner = d.c_in * d.k;
    const float* src = d.v + (size_t)row * inner;
    double acc = 0.0;
    float mx = 0.f;
    if ((inner & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {       // 16-byte loads, two in flight (a row of conv_pre: 5 376 floats)
        const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
        const int n4 = inner >> 2;
        int i = threadIdx.x;
        for (; i + 256 < n4; i += 512) {
            const f32x4 x = s4[i], y = s4[i + 256];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc += (double)x[e] * x[e]; mx = fmaxf(mx, fabsf(x[e])); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc += (double)y[e] * y[e]; mx = fmaxf(mx, fabsf(y[e])); }
        }
        if (i < n4) {
            const f32x4 x = s4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc += (double)x[e] * x[e]; mx = fmaxf(mx, fabsf(x[e])); }
        }
    } else {
        for (int i = threadIdx.x; i < inner; i += 256) { const float x = src[i]; acc += (double)x * x; mx = fmaxf(mx, fabsf(x)); }
    }
    const double n2 = v2w_block_sum(acc, red);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float sc = d.g ? (float)((double)d.g[row] / sqrt(n2)) : 1.f;
        d.rowscale[row] = sc;
        mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3])) * fabsf(sc);
        if (d.mode != 1)            // (bf16 fragments carry no scale record)
            atomicMax(reinterpret_cast<unsigned int*>(d.sc) + 2, __float_as_uint(mx));   // order of non-negative floats = order of their bits
    }
}

// one block per (layer, 32-row block mb, 16-channel chunk ch): the 32 x 16 x K sub-block of v goes through LDS (coalesced
// rows), then the K units of that (mb, ch) are written contiguously as (hi, lo) fragments
__global__ void __launch_bounds__(256)
split_pack_batch_kernel(const v2w_split_desc* __restrict__ descs, const int32_t* __restrict__ starts, int n) {
    extern __shared__ float tile[];
    const int li = split_find_layer(starts, n, blockIdx.x);
    const v2w_split_desc d = descs[li];
    const int blk = blockIdx.x - starts[li];
    const int K = d.k, nch = d.c_in / V2W_SPLIT_CK;
    const int mb = blk / nch, ch = blk % nch;
    const bool bf = d.mode == 1;
    const float scale = bf ? 1.f : split_scale_from_bits(reinterpret_cast<const unsigned int*>(d.sc)[2]);
    if (blk == 0 && threadIdx.x == 0) { d.sc[0] = 1.f / scale; d.sc[1] = scale; }
    const int rlen = V2W_SPLIT_CK * K, rstride = rlen + 1;
    if ((reinterpret_cast<uintptr_t>(d.v) & 15) == 0) {
        // a row's 16 K floats are contiguous and 16-byte aligned (16 K * 4 bytes per (row, chunk)): float4 loads, a quarter of the loads and of the
        // index divisions (round 6: conv_pre's pack is on the forward's critical path)
        const int r4 = rlen >> 2;
        for (int q = threadIdx.x; q < 32 * r4; q += 256) {
            const int r = q / r4, x = (q - r * r4) * 4;
            const int row = mb * 32 + r;                         // rows past C_out (C_out = 16): MFMA padding
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (row < d.c_out) {
                const float rs = d.rowscale[row];
                w = *reinterpret_cast<const f32x4*>(d.v + ((size_t)row * d.c_in + ch * V2W_SPLIT_CK) * K + x);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = w[e] * rs * scale;        // (the scalar form's products, in its order)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r * rstride + x + e] = w[e];
        }
    } else
    for (int idx = threadIdx.x; idx < 32 * rlen; idx += 256) {
        const int r = idx / rlen, x = idx - r * rlen;
        const int row = mb * 32 + r;                             // rows past C_out (C_out = 16): MFMA padding
        tile[r * rstride + x] = row < d.c_out ? d.v[((size_t)row * d.c_in + ch * V2W_SPLIT_CK) * K + x] * d.rowscale[row] * scale : 0.f;
    }
    __syncthreads();
    h8* dst = reinterpret_cast<h8*>(d.wps) + (size_t)(mb * nch + ch) * K * 128;
    for (int o = threadIdx.x; o < K * 64; o += 256) {
        const int lane = o & 63, t = o >> 6;
        const int r = lane & 31, c0 = 8 * (lane >> 5);
        if (bf) {
            b8 hb;
#pragma unroll
            for (int j = 0; j < 8; ++j) hb[j] = (__bf16)tile[r * rstride + (c0 + j) * K + t];
            reinterpret_cast<b8*>(dst)[t * 128 + lane] = hb;
            continue;
        }
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tile[r * rstride + (c0 + j) * K + t];
            const _Float16 h = (_Float16)v;
            hi[j] = h;
            lo[j] = (_Float16)(v - (float)h);
        }
        dst[t * 128 + lane] = hi;
        dst[t * 128 + 64 + lane] = lo;
    }
}

}  // namespace

// C_out = 16: the 16 rows are zero-padded to one 32-row block (the fused C = 16 stage kernel)
extern "C" int v2w_split_packable(int c_in, int c_out) {
    return (c_in > 0 && c_out > 0 && c_in % V2W_SPLIT_CK == 0 && (c_out % 32 == 0 || c_out == 16)) ? 1 : 0;
}

extern "C" int v2w_split_supported(int c_in, int c_out, int u) {
    return (u == 1 && c_in % V2W_SPLIT_CK == 0 && c_out % 64 == 0) ? 1 : 0;
}

// wps: k*c_in*c_out*4 + 2048 bytes (two halves per weight + one unit of padding: stage copies move whole tap pairs); sc: 4 floats of device scratch/output: [0] = 1/scale (pass as `winv`),
// [1] = scale, [2] = max |w| bits (zeroed here).  Three tiny launches on `stream`.
extern "C" int v2w_pack_split(const float* wf, void* wps, float* sc, int k, int c_in, int c_out, void* stream) {
    if (!wf || !wps || !sc || k <= 0 || c_in <= 0 || c_out <= 0) return V2W_E_ARG;
    if (!v2w_split_packable(c_in, c_out)) return V2W_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    unsigned int* amax = reinterpret_cast<unsigned int*>(sc + 2);
    hipError_t e = v2w_sink(st) ? hipSuccess : hipMemsetAsync(amax, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return (int)e;
    const size_t n = (size_t)k * c_in * c_out;
    int g1 = (int)((n + 255) / 256); if (g1 > 256) g1 = 256;
    V2W_LAUNCH(split_absmax_kernel, dim3(g1), dim3(256), 0, st, wf, n, amax);
    const size_t total = (size_t)k * c_in * ((c_out + 31) / 32 * 32) / 8;   // one thread per 8 (padded) weights
    int g2 = (int)((total + 255) / 256); if (g2 > 2048) g2 = 2048;
    V2W_LAUNCH(pack_split_kernel, dim3(g2), dim3(256), 0, st, wf, reinterpret_cast<h8*>(wps), amax, sc, k, c_in, c_out);
    return v2w_launch_status();
}

extern "C" int v2w_pack_bf16(const float* wf, void* wps, float* sc, int k, int c_in, int c_out, void* stream) {
    if (!wf || !wps || !sc || k <= 0 || c_in <= 0 || c_out <= 0) return V2W_E_ARG;
    if (!v2w_split_packable(c_in, c_out)) return V2W_E_SHAPE;
    const size_t total = (size_t)k * c_in * ((c_out + 31) / 32 * 32) / 8;
    int g2 = (int)((total + 255) / 256); if (g2 > 2048) g2 = 2048;
    V2W_LAUNCH(pack_bf16_kernel, dim3(g2), dim3(256), 0, (hipStream_t)stream, wf, reinterpret_cast<b8*>(wps), sc, k, c_in, c_out);
    return v2w_launch_status();
}

int v2w_conv1d_bf16(const v2w_conv1d_args* a, int n, hipStream_t stream, int32_t* cfg);   // v2w_conv_bf16.hip

// Called by v2w_api.hip for algo == V2W_ALGO_SPLIT (bf = false) and V2W_ALGO_BF16 (bf = true).  n problems sharing B, C_in, C_out, L in one launch.
int v2w_conv1d_split(const v2w_conv1d_args* a, int n, hipStream_t stream, bool bf) {
    if (n < 1 || n > V2W_MAX_MULTI) return V2W_E_ARG;
    if (!v2w_split_supported(a->C_in, a->C_out, 1) && !(bf && a->C_out == 32 && a->C_in % 32 == 0)) return V2W_E_SHAPE;
    if (bf) {      // bf16 operands: the chunk-per-barrier kernel of v2w_conv_bf16.hip; shapes it does not take fall through to this file's
        const int rc = v2w_conv1d_bf16(a, n, stream, nullptr);
        if (rc != V2W_E_SHAPE || a->C_out % 64 != 0) return rc;             // (32 output channels: the bf16 kernel or nothing)
        for (int i = 0; i < n; ++i) if (a[i].io_bf16) return V2W_E_SHAPE;      // this file's kernels read and write fp32 only
    }
    TileArgs ps[V2W_MAX_MULTI];
    long tiles256 = 0;
    for (int i = 0; i < n; ++i) {
        const v2w_conv1d_args* q = a + i;
        if (q->B != a->B || q->C_in != a->C_in || q->C_out != a->C_out || q->L != a->L) return V2W_E_SHAPE;
        if (!q->wps || !q->winv) return V2W_E_ARG;
        TileArgs p{};
        p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.wps = q->wps; p.winv = q->winv; p.bias = q->bias;
        p.res = q->res; p.res_a = q->res_a; p.res_s = q->res_s; p.out = q->out;
        p.add0 = q->add0; p.add1 = q->add1;
        p.mask_src = q->mask_src; p.mask_a = q->mask_a; p.mask_s = q->mask_s; p.mask_slope = q->mask_slope;
        p.B = q->B; p.Cin = q->C_in; p.Cout = q->C_out; p.L = q->L; p.K = q->k; p.dil = q->dil;
        p.CinT = q->in_ct > 0 ? q->in_ct : q->C_in; p.CoutT = q->out_ct > 0 ? q->out_ct : q->C_out;   // one group of a grouped conv: channel slices
        p.pad = 0; p.hl = p.hr = q->dil * (q->k - 1) / 2;
        if (q->pad_left >= 0) { p.hl = q->pad_left; p.hr = q->dil * (q->k - 1) - q->pad_left; if (p.hr < 0) return V2W_E_ARG; }
        p.in_stride = q->in_stride > 0 ? q->in_stride : 1; p.in_phase = q->in_phase;
        p.slope = q->slope; p.accumulate = q->accumulate; p.out_div = q->out_div;
        p.out_slope = q->out_slope == 0.f ? 1.f : q->out_slope;
        ps[i] = p;
        tiles256 += (long)p.B * ((p.L + 255) / 256) * (p.Cout / 128);
    }
    int hmax = 0;
    for (int i = 0; i < n; ++i) { const int h = ((ps[i].hl + 3) & ~3) > ps[i].hr ? ((ps[i].hl + 3) & ~3) : ps[i].hr; if (h > hmax) hmax = h; }
    if (hmax > V2W_SPLIT_HMAX) {      // DiscriminatorP's last conv at periods 17 / 19 (dilation = period on the flattened axis): the wide-halo instantiations
        if (bf || hmax > 40) return V2W_E_SHAPE;
        if (a->C_out % 128 == 0 && 2 * tiles256 >= 384) return launch_split<2, 2, 2, 2, 40>(ps, n, stream, false);
        return launch_split<1, 2, 2, 2, 40>(ps, n, stream, false);
    }
    // two workgroups per CU are resident: prefer the largest tile that still gives every slot ~1 workgroup
    if (a->C_out % 128 == 0) {
        if (V2W_SPLIT_FORCE == 1) return launch_split<1, 2, 2, 2>(ps, n, stream, bf);
        if (V2W_SPLIT_FORCE == 2) return launch_split<2, 2, 2, 2>(ps, n, stream, bf);
        // (128 x 256 = <2, 4, 2, 2> halves the weight traffic but needs > 256 VGPRs at two workgroups per CU: spills, 2x slower)
        if (2 * tiles256 >= 384) return launch_split<2, 2, 2, 2>(ps, n, stream, bf);       // 128 x 128
        return launch_split<1, 2, 2, 2>(ps, n, stream, bf);                                // 64 x 128
    }
    return launch_split<2, 2, 1, 4>(ps, n, stream, bf);                                    // 64 x 256
}

// Batched form of (v2w_wn_fold_conv + v2w_pack_split) for n Conv1d layers: descs / starts live in DEVICE memory;
// starts[0..n] = prefix sums of c_out (rows), starts[n+1 .. 2n+1] = prefix sums of (c_out/32)*(c_in/16) (pack blocks).
// all_bf16 != 0: every descriptor has mode 1 - no scale record is kept, the launch that zeroes the records is left out (two launches)
extern "C" int v2w_split_pack_batch(const v2w_split_desc* descs_dev, const int32_t* starts_dev, int n, int nblk_rows, int nblk_pack,
                                    int k_max, int all_bf16, void* stream) {
    if (!descs_dev || !starts_dev || n <= 0 || nblk_rows <= 0 || nblk_pack <= 0 || k_max <= 0) return V2W_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int lds = 32 * (V2W_SPLIT_CK * k_max + 1) * (int)sizeof(float);
    if (lds > 64 * 1024) return V2W_E_SHAPE;
    if (!all_bf16) V2W_LAUNCH(split_zero_batch_kernel, dim3((n + 63) / 64), dim3(64), 0, st, descs_dev, n);
    V2W_LAUNCH(split_rowscale_batch_kernel, dim3(nblk_rows), dim3(256), 0, st, descs_dev, starts_dev, n);
    V2W_LAUNCH(split_pack_batch_kernel, dim3(nblk_pack), dim3(256), lds, st, descs_dev, starts_dev + n + 1, n);
    return v2w_launch_status();
}
