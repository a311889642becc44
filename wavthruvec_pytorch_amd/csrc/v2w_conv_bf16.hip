// bf16-operand implicit-GEMM Conv1d on the gfx950 bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the wide layers of
// BASELINE configs[2] ("bf16 compute / fp32 accumulate"), V2W_ALGO_BF16.
//
//   out[b,co,l] = bias[co] + sum_{ci,t} bf16(W[t][ci][co]) * bf16(act(in[b,ci,l + (t-(k-1)/2)*dil]))      (+ the fused residual / addends)
//
// Same tile structure as the exact-fp32 kernel (v2w_conv_mfma.hip) - one barrier per 32-channel chunk, every tap a row offset into one
// position-major LDS tile, weights streamed from L2 into a register ring, LDS-transposed float4 epilogue - re-dimensioned for a
// matrix pipe that is 16x faster:
//   B (signal)  : LDS tile Xs[positions][32 ch bf16 + pad] (80-byte rows): the 8 k-values a lane feeds to ONE MFMA are one
//                 conflict-free ds_read_b128.  Affine + leaky_relu + rounding to bf16 happen once, when a chunk is staged.
//   A (weights) : the bf16 fragments of v2w_pack_bf16 / v2w_split_pack_batch (per 32-row block, 16-channel k-step and tap: 64 lanes x
//                 16 bytes), read with one global_load_dwordx4 (scalar base + lane offset) per MFMA row block and k-step.  A wave
//                 covers 128 positions (NI = 4), so a fragment feeds four MFMAs: 32 B/clk/CU of L2 -> CU traffic at full MFMA rate.
//   unit        : one tap of one chunk = 2 k-steps x MI x NI MFMAs; the next tap's fragments are requested a unit ahead, each column
//                 block's operand register is refilled right after its last use.
// ConvTranspose1d(k, stride U, pad (k-U)/2) runs on the SAME kernel (EPI = 2) as a Conv1d over UP * C_out "virtual" output channels
// (row = co * UP + phase, UP = U rounded up to a power of two): output phase r of position q reads inputs q + c_r - m, i.e. offsets
// -1, 0, +1 for every upsampler of the generator, so the transposed conv IS a 3-tap conv whose virtual weights hold the polyphase
// taps (zeros where a phase has no tap at an offset: 27-33 % of the MFMAs, on a pipe this path does not saturate).  The epilogue
// interleaves the phases back into (B, C_out, U*L) through an LDS scratch laid out like the output, stores float4s along positions and
// emits the BatchNorm partial sums (modules.py:23) of its tile.
// The older split kernel (v2w_conv_split.hip, one workgroup barrier per (16-channel chunk, tap) stage) stays the f16x3 path; in bf16
// mode a stage of it was 128 cycles of MFMA issue in ~1300 cycles.
#include <type_traits>
#include "v2w_tile.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define V2W_BF_CK 32        // channels per chunk (two MFMA k-steps)
#define V2W_BF_ROWB 80      // bytes per staged position: 32 ch bf16 (64 B) + 16 B pad: conflict-free ds_read_b128
#define V2W_BF_UNIT 2048    // byte pitch of the packed fragments of one (32-row block, 16-channel k-step, tap); the first KiB is bf16

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;          // plain casts: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
    return __builtin_bit_cast(unsigned int, v);
}

// bf16 <-> fp32 on raw words: element 0 of a packed pair is the low half
__device__ __forceinline__ float bf_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ f32x4 bf4_to_f32(u32x2 w) { return f32x4{bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])}; }
__device__ __forceinline__ u32x2 f32_to_bf4(f32x4 v) { return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; }

// IN_BF / OUT_BF: activation STORAGE in bf16 (`in`; `out`, `res`, `add0`, `add1`): 8-byte loads / stores of 4 positions instead of 16.
// CKT: channels per chunk.  32 (two k-steps per tap); 64 for the C_in = 64 layers, which then are ONE chunk: no chunk loop, twice the
// bytes in flight while the tile is staged (a 64-channel layer as two chunks spent 85 % of its tile time outside the MFMA phases).
// VEC: float4-aligned input (L % 4 == 0, 16-byte aligned base) - the vector staging path.  A template parameter, not a run-time flag:
// with the element-wise fallback in the same kernel its (never taken) loads sit on a path that joins the hot one, and hipcc then waits
// vmcnt(0) at the join - once in front of every chunk's prefetch (for the fragment loads in flight) and once right after it (for the
// prefetch itself: a full memory latency per chunk, in the open).
template <int MI, int NI, int WM, int WN, int NPF, int EPI, bool IN_BF, bool OUT_BF, int CKT = V2W_BF_CK, bool VEC = true>
__global__ void __launch_bounds__(64 * WM * WN, MI * NI >= 8 ? 2 : 3)      // (128 x 256: two workgroups per CU = at most 256 registers; else three)
conv_bf16_kernel(const MultiArgs m) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, CK = CKT, ROWB = 2 * CK + 16, KS = CK / 16, NCQ = CK / 4;
    static_assert(CK == 32 || CK == 64, "chunk = 2 or 4 k-steps");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // 2 x [xrows][ROWB] signal tiles, then the float tables

    int pq = 0;
#pragma unroll
    for (int i = 1; i < V2W_MAX_MULTI; ++i) pq += (int)blockIdx.x >= m.start[i] ? 1 : 0;
    const TileArgs p = pinned_tile_args(m.p[pq]);
    const int mtiles = p.Cout / MT;
    const int id = blockIdx.x - m.start[pq];
    const int grp = id / (8 * mtiles), rem = id % (8 * mtiles);
    const int mt = rem >> 3;
    const int tile = grp * 8 + (rem & 7);
    if (tile >= p.ntiles) return;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * NT;
    const int m0 = mt * MT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, hk = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MI);
    const int wn0 = (wave % WN) * (32 * NI);
    const int L = p.L, K = p.K;
    const float slope = p.slope;
    const int nch = p.Cin / CK;
    const int pos0 = n0 - p.hla;
    const int bufsz = p.xrows * ROWB;                       // bytes per signal buffer
    float* const etab = reinterpret_cast<float*>(smem_b) + p.atab_off;     // bias, res_a, res_s, mask_a, mask_s [MT] each
    float* const atab = etab + 5 * MT;                                     // a[Cin] then s[Cin]

    acc_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    V2W_STAMP(0);

    // ---- staging: an item = 4 consecutive channels x 4 positions: four float4 loads, four 8-byte LDS stores (the 4 channels of one
    // position are 8 contiguous bytes of its row).  Consecutive lanes take the 8 channel quads of one position quad (a whole 64-byte
    // row per 8 lanes: conflict-free stores; 128-byte pieces of 8 x 4 rows on the global side), then the next position quad.
    const int nq = p.xrows >> 2;
    typedef typename std::conditional<IN_BF, u32x2, f32x4>::type pf_t;       // 4 positions of one channel as loaded
    pf_t pf[NPF][4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto item = [&](int s, int& cq, int& row, bool& in_img, bool& in_seq) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int idx = t + s * NTHREADS;
        cq = idx & (NCQ - 1);
        row = (idx / NCQ) * 4;
        in_img = (idx / NCQ) < nq;
        const int pos = pos0 + row;
        in_seq = in_img && pos >= 0 && pos < L;         // L % 4 == 0 and pos % 4 == 0: a float4 is inside or outside as a whole
    };
    // item s of a chunk always reads (channel quad cq, positions row .. row + 3) of the chunk's 32 channels: the address is a wave-
    // uniform base (batch item, chunk, channel i of the quad: scalar arithmetic) + a per-lane byte offset that never changes
    constexpr int ESI = IN_BF ? 2 : 4;
    unsigned poff[NPF];
#pragma unroll
    for (int s = 0; s < NPF; ++s) {
        int cq, row; bool in_img, in_seq;
        item(s, cq, row, in_img, in_seq);
        poff[s] = (unsigned)(4 * cq * L + (in_seq ? pos0 + row : 0)) * ESI;       // outside the sequence: position 0, zeroed in commit
        asm volatile("" : "+v"(poff[s]));
    }
    auto prefetch = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* base = reinterpret_cast<const unsigned char*>(p.in) + (size_t)(b * p.Cin + ci0 + i) * L * ESI;
#pragma unroll
            for (int s = 0; s < NPF; ++s) pf[s][i] = *gptr<const pf_t>(base + poff[s]);
        }
    };
    auto commit = [&](int ci0, unsigned char* Xs) {
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            int cq, row; bool in_img, in_seq;
            item(s, cq, row, in_img, in_seq);
            if (!in_img) continue;
            float av[4], sv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = atab[ci0 + 4 * cq + i];               // (the table is (1, 0) without an affine: no scalar branch in here -
                sv[i] = atab[p.Cin + ci0 + 4 * cq + i];       //  two inlined copies of this lambda with one ended in a backend error)
            }
            unsigned char* dst = Xs + row * ROWB + cq * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float xv;
                    if constexpr (IN_BF) xv = (e & 1) ? bf_hi(pf[s][i][e >> 1]) : bf_lo(pf[s][i][e >> 1]);
                    else xv = pf[s][i][e];
                    a[i] = v2w_lrelu(fmaf(av[i], xv, sv[i]), slope);
                }
                u32x2 v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
                if (!in_seq) v = u32x2{0u, 0u};             // padding stays exactly 0 (it pads the ACTIVATED signal)
                *reinterpret_cast<u32x2*>(dst + e * ROWB) = v;
            }
        }
    };
    auto stage_scalar = [&](int ci0, unsigned char* Xs) {   // any L / alignment: one element at a time
        for (int c = wave; c < CK; c += WM * WN) {
            const int ch = b * p.Cin + ci0 + c;
            const float av = p.in_a ? gptr<const float>(p.in_a)[ch] : 1.f, sv = p.in_s ? p.in_s[ch] : 0.f;
            for (int j = lane; j < p.xrows; j += 64) {
                const int l = pos0 + j;
                float v = 0.f;
                if (l >= 0 && l < L) {
                    const float xv = IN_BF ? bf_lo(gptr<const unsigned short>(p.in)[(size_t)ch * L + l]) : gptr<const float>(p.in)[(size_t)ch * L + l];
                    v = v2w_lrelu(fmaf(av, xv, sv), slope);
                }
                reinterpret_cast<__bf16*>(Xs + j * ROWB)[c] = (__bf16)v;
            }
        }
    };

    // ---- weights: fragment (row block, 16-channel k-step c16, tap t) sits at ((rb * nst + c16 * K + t) * V2W_BF_UNIT) + lane * 16
    const int nst = KS * nch * K;                            // fragments per row block
    const unsigned char* ap[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
        ap[i] = reinterpret_cast<const unsigned char*>(p.wps) + (size_t)((m0 + wm0) / 32 + i) * nst * V2W_BF_UNIT;
    const unsigned lane16 = (unsigned)lane * 16u;
    // ring of four fragments, each refilled right after its use with the fragment four k-steps on: three k-steps (3 x 32 MI NI cycles of
    // MFMA issue) ahead of an L2 round trip of 500+.  CK = 64: a slot per k-step of a tap.  CK = 32 (two k-steps per tap): the taps of the
    // whole tile are walked in PAIRS - even tap slots 0, 1, odd tap slots 2, 3 - across chunk boundaries (tap counts are odd).
    constexpr bool PAIRS = KS == 2 && MI >= 2 && IN_BF && EPI != 1;      // (the other instantiations have no registers for it)
    u32x4 ar[(KS == 4 || PAIRS) ? 4 : 2][MI];
    auto load_frag = [&](u32x4 (&a)[MI], int ch, int s, int t) {   // k-step s of (chunk ch, tap t); clamped past the end (harmless re-read)
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));                        // keeps the address scalar base + 32-bit lane offset (see v2w_conv_mfma.hip)
        const int chc = ch < nch ? ch : nch - 1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *gptr<const u32x4>(ap[i] + (size_t)((KS * chc + s) * K + t) * V2W_BF_UNIT + l16);
    };

    // ---- B operands: one 16-byte fragment per column block, refilled right after its last use in the running k-step
    u32x4 bb[NI];
    auto mfma = [&](acc_t c, u32x4 a, u32x4 bfrag) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, bfrag), c, 0, 0, 0);
    };
    // one k-step: MI x NI MFMAs; `nxt` = this lane's 16 bytes of the NEXT k-step (same rows + 32 bytes, or the next tap's rows)
    auto kstep = [&](const u32x4 (&a)[MI], const unsigned char* nxt) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = mfma(acc[i][j], a[i], bb[j]);
            bb[j] = *reinterpret_cast<const u32x4*>(nxt + j * 32 * ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- typed access to the output-side tensors (out, res, add0, add1): fp32, or bf16 storage (OUT_BF) with fp32 arithmetic
    auto ld4 = [&](const float* base, size_t off) {
        if constexpr (OUT_BF) return bf4_to_f32(*gptr<const u32x2>(reinterpret_cast<const unsigned short*>(base) + off));
        else return *gptr<const f32x4>(base + off);
    };
    auto st4 = [&](float* base, size_t off, f32x4 v) {
        if constexpr (OUT_BF) *gptr<u32x2>(reinterpret_cast<unsigned short*>(base) + off) = f32_to_bf4(v);
        else *gptr<f32x4>(base + off) = v;
    };
    auto ld1 = [&](const float* base, size_t off) {
        if constexpr (OUT_BF) return bf_lo(gptr<const unsigned short>(base)[off]);
        else return gptr<const float>(base)[off];
    };
    auto st1 = [&](float* base, size_t off, float v) {
        if constexpr (OUT_BF) gptr<__bf16>(base)[off] = (__bf16)v;
        else gptr<float>(base)[off] = v;
    };

    // ---- prologue: the loads that do not depend on LDS go out first (chunk 0 of the signal, the first fragments)
    if constexpr (VEC) prefetch(0);
    load_frag(ar[0], 0, 0, 0);
    if constexpr (KS == 4) { load_frag(ar[1], 0, 1, 0); load_frag(ar[2], 0, 2, 0); load_frag(ar[3], 0, 3, 0); }
    else if constexpr (PAIRS) {
        load_frag(ar[1], 0, 1, 0);
        const int c1 = K > 1 ? 0 : 1, t1 = K > 1 ? 1 : 0;   // tap 1 of the tile
        load_frag(ar[2], c1, 0, t1); load_frag(ar[3], c1, 1, t1);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int c = tid; c < MT; c += NTHREADS) {
        etab[c] = p.bias ? p.bias[EPI == 2 ? (m0 + c) / p.up_p : m0 + c] : 0.f;
        etab[MT + c] = p.res_a ? p.res_a[b * p.Cout + m0 + c] : 1.f;
        etab[2 * MT + c] = p.res_a ? p.res_s[b * p.Cout + m0 + c] : 0.f;
        etab[3 * MT + c] = p.mask_a ? p.mask_a[b * p.Cout + m0 + c] : 1.f;
        etab[4 * MT + c] = p.mask_a ? p.mask_s[b * p.Cout + m0 + c] : 0.f;
    }
    for (int c = tid; c < p.Cin; c += NTHREADS) {
        atab[c] = p.in_a ? gptr<const float>(p.in_a)[b * p.Cin + c] : 1.f;
        atab[p.Cin + c] = p.in_a ? p.in_s[b * p.Cin + c] : 0.f;
    }
    __syncthreads();
    if constexpr (VEC) commit(0, smem_b);
    else stage_scalar(0, smem_b);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    V2W_STAMP(1);

    const int lbase = (wn0 + lr + p.hla - p.hl) * ROWB + 16 * hk;     // this lane's 16 bytes in the row of (its column, tap 0), k-step 0
    const int step = p.dil * ROWB;
    if constexpr (!PAIRS) {
    for (int ch = 0; ch < nch; ++ch) {
        const unsigned char* Xs = smem_b + (ch & 1) * bufsz;
        unsigned char* Xn = smem_b + ((ch + 1) & 1) * bufsz;
        const bool more = ch + 1 < nch;
        if constexpr (VEC) { if (more) prefetch((ch + 1) * CK); }
        __builtin_amdgcn_sched_barrier(0);
        if (ch < 6) V2W_STAMP(2 + 4 * ch);
        const unsigned char* xt = Xs + lbase;
#pragma unroll
        for (int j = 0; j < NI; ++j) bb[j] = *reinterpret_cast<const u32x4*>(xt + j * 32 * ROWB);
        // a tap = k-step 0 (channels 0-15 of the chunk: bytes 0-31 of a row) from ring slot 0, then k-step 1 (bytes 32-63) from slot 1;
        // each slot's next fragment is requested while the other slot computes; the chunk's last request is tap 0 of the next chunk
        if constexpr (KS == 4) {
            for (int t = 0; t < K; ++t, xt += step) {
                const bool last = t + 1 >= K;
                const int chn = last ? ch + 1 : ch, tn = last ? 0 : t + 1;     // (one load either way: no branch, no join for vmcnt)
#pragma unroll
                for (int sq = 0; sq < 4; ++sq) {
                    kstep(ar[sq], sq < 3 ? xt + 32 * (sq + 1) : (last ? xt : xt + step));
                    load_frag(ar[sq], chn, sq, tn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else
        for (int t = 0; t < K; ++t, xt += step) {
#pragma unroll
            for (int sp = 0; sp < KS; sp += 2) {             // k-steps sp (ring slot 0) and sp + 1 (slot 1) of the tap
                load_frag(ar[1], ch, sp + 1, t);
                __builtin_amdgcn_sched_barrier(0);
                kstep(ar[0], xt + 32 * (sp + 1));
                if (sp + 2 < KS) load_frag(ar[0], ch, sp + 2, t);
                else if (t + 1 < K) load_frag(ar[0], ch, 0, t + 1);
                else load_frag(ar[0], ch + 1, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                kstep(ar[1], sp + 2 < KS ? xt + 32 * (sp + 2) : (t + 1 < K ? xt + step : xt));
            }
        }
        if (ch < 6) V2W_STAMP(3 + 4 * ch);
        if (more) {
            if constexpr (VEC) commit((ch + 1) * CK, Xn);
            else stage_scalar((ch + 1) * CK, Xn);
            if (ch < 6) V2W_STAMP(4 + 4 * ch);
            __syncthreads();
            if (ch < 6) V2W_STAMP(5 + 4 * ch);
        }
    }
    } else {
        // ---- CK = 32: the tile's nch * K taps in pairs
        int ch = 0, t = 0;                                   // the running tap
        int qc = 0, qt = 2;                                  // the tap two ahead of it (whose fragments the running tap requests)
        while (qt >= K) { qt -= K; ++qc; }
        if constexpr (VEC) { if (nch > 1) prefetch(CK); }
        __builtin_amdgcn_sched_barrier(0);
        V2W_STAMP(2);
        const unsigned char* xt = smem_b + lbase;
#pragma unroll
        for (int j = 0; j < NI; ++j) bb[j] = *reinterpret_cast<const u32x4*>(xt + j * 32 * ROWB);
        auto tap = [&](auto par_c) {
            constexpr int S0 = 2 * decltype(par_c)::value;
            const bool last = t + 1 >= K;
            kstep(ar[S0], xt + 32);
            load_frag(ar[S0], qc, 0, qt);
            __builtin_amdgcn_sched_barrier(0);
            kstep(ar[S0 + 1], last ? xt : xt + step);
            load_frag(ar[S0 + 1], qc, 1, qt);
            __builtin_amdgcn_sched_barrier(0);
            if (++qt >= K) { qt = 0; ++qc; }
            if (!last) { ++t; xt += step; return; }
            // ---- chunk boundary (touches neither the accumulators nor the ring)
            if (ch < 6) V2W_STAMP(3 + 4 * ch);
            const bool more = ch + 1 < nch;
            if (more) {
                if constexpr (VEC) commit((ch + 1) * CK, smem_b + ((ch + 1) & 1) * bufsz);
                else stage_scalar((ch + 1) * CK, smem_b + ((ch + 1) & 1) * bufsz);
                if (ch < 6) V2W_STAMP(4 + 4 * ch);
                __syncthreads();
                if (ch < 6) V2W_STAMP(5 + 4 * ch);
            }
            ++ch; t = 0;
            if (ch < nch) {
                if constexpr (VEC) { if (ch + 1 < nch) prefetch((ch + 1) * CK); }
                __builtin_amdgcn_sched_barrier(0);
                if (ch < 6) V2W_STAMP(2 + 4 * ch);
                xt = smem_b + (ch & 1) * bufsz + lbase;
#pragma unroll
                for (int j = 0; j < NI; ++j) bb[j] = *reinterpret_cast<const u32x4*>(xt + j * 32 * ROWB);
            }
        };
        const int TT = nch * K;
        int g = 0;
        for (; g + 1 < TT; g += 2) { tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); }
        if (g < TT) tap(std::integral_constant<int, 0>{});
    }
    V2W_STAMP(26);

    if constexpr (EPI == 2) {
        // ---- transposed-conv epilogue: 64 input positions of one 32-row block at a time -> scratch [32 / UP channels][U * 64 output
        // positions] (the output's own layout) -> + bias, float4 stores along positions, per-channel (sum, sumsq) of what was stored.
        const int U = p.up_u, UP = p.up_p;
        const int CoutR = p.Cout / UP;                     // real output channels
        const int nco = 32 / UP;                           // channels per 32-row block
        const int ORS = U * 64, C4 = U * 16;               // scratch row (floats), float4s per row
        const int Lout = L * U;
        float* const red = atab + 2 * p.Cin;                   // [WN][MT / UP][2] partial sums of this workgroup's waves
        __syncthreads();
        float* const scr = reinterpret_cast<float*>(smem_b) + wave * 2048;
        for (int c = lane; c < (MI * 32 / UP) * 2; c += 64) red[((wave % WN) * (MT / UP) + (wm0 / UP)) * 2 + c] = 0.f;

        // Fast form (whole tile inside the sequence, float4-aligned output, a stride the generator uses): compile-time U / UP, so
        //   - the accumulators go to the scratch as 8- / 16-byte LDS stores where the phases of a channel are adjacent registers,
        //   - the scratch IS the output block ([channel][U * 64 positions], contiguous): float4 number lane + 64 g of it is one
        //     ds_read_b128 at a constant offset, all 64 lanes busy for every stride,
        //   - the BatchNorm partial sums stay in registers (a float4 of running sums per g) across the passes of a row block and are
        //     reduced once per row block, through the scratch, in a fixed order (the older form: two 64-lane shuffle reductions per
        //     channel and pass - 2 x 16 per pass at stride 2 - and half-empty waves for U = 2).
        auto fast = [&](auto u_c, auto up_c) {
            constexpr int U = decltype(u_c)::value, UP = decltype(up_c)::value;
            constexpr int NCO = 32 / UP, RW = 16 * U, G = NCO * U / 4, ORS = U * 64, ES = OUT_BF ? 2 : 4;
            constexpr int PARTS = 64 / (2 * NCO), NPER = RW / PARTS;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int Lout = L * U;
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
                        const acc_t& a = acc[i][jh + jj];
                        float* const sc = scr + U * (jj * 32 + lr);
#pragma unroll
                        for (int eg = 0; eg < 4; ++eg) {
                            if constexpr (UP == 2) {                 // rows 8 eg + 4 hk + {0..3}: channels 4 eg + 2 hk + {0, 1}, phases {0, 1}
                                *reinterpret_cast<f32x2*>(sc + (4 * eg + 2 * hk) * ORS) = f32x2{a[4 * eg], a[4 * eg + 1]};
                                *reinterpret_cast<f32x2*>(sc + (4 * eg + 2 * hk + 1) * ORS) = f32x2{a[4 * eg + 2], a[4 * eg + 3]};
                            } else if constexpr (UP == 4) {          // channel 2 eg + hk, phases 0..3
                                *reinterpret_cast<f32x4*>(sc + (2 * eg + hk) * ORS) = f32x4{a[4 * eg], a[4 * eg + 1], a[4 * eg + 2], a[4 * eg + 3]};
                            } else if constexpr (U == 8) {           // channel eg, phases 4 hk .. 4 hk + 3
                                *reinterpret_cast<f32x4*>(sc + eg * ORS + 4 * hk) = f32x4{a[4 * eg], a[4 * eg + 1], a[4 * eg + 2], a[4 * eg + 3]};
                            } else {                                 // UP = 8, U = 5: channel eg, phases 4 hk + {0..3} < 5
                                sc[eg * ORS + 4 * hk] = a[4 * eg];
                                if (hk == 0) { sc[eg * ORS + 1] = a[4 * eg + 1]; sc[eg * ORS + 2] = a[4 * eg + 2]; sc[eg * ORS + 3] = a[4 * eg + 3]; }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const size_t ub = ((size_t)b * CoutR + co0) * Lout + (size_t)U * (n0 + wn0 + jh * 32);     // (uniform)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int idx = lane + 64 * g;
                        int cl, c4;
                        if constexpr (RW == 32) { cl = 2 * g + (lane >> 5); c4 = lane & 31; }
                        else if constexpr (RW == 64) { cl = g; c4 = lane; }
                        else if constexpr (RW == 128) { cl = g >> 1; c4 = lane + 64 * (g & 1); }
                        else { cl = (idx >= RW) + (idx >= 2 * RW) + (idx >= 3 * RW); c4 = idx - cl * RW; }
                        f32x4 v = *reinterpret_cast<const f32x4*>(scr + 4 * idx);
                        const float bias = etab[wm0 + i * 32 + cl * UP];
                        v += f32x4{bias, bias, bias, bias};
                        if (p.stats_part) { sa[g] += v; sq[g] += v * v; }
                        unsigned vo = (unsigned)(cl * Lout + 4 * c4) * ES;
                        unsigned char* ob = reinterpret_cast<unsigned char*>(p.out) + ub * ES;
                        if constexpr (OUT_BF) *gptr<u32x2>(ob + vo) = f32_to_bf4(v);
                        else *gptr<f32x4>(ob + vo) = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (p.stats_part) {
                    // per-lane sums -> scratch[g * 64 + lane] = (sum, sumsq): slot k of it belongs to channel k / RW; lane = (channel,
                    // part, stat) adds NPER slots in order, the parts of a channel meet in a fixed shuffle tree
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        *reinterpret_cast<f32x2*>(scr + 2 * (g * 64 + lane)) =
                            f32x2{(sa[g][0] + sa[g][1]) + (sa[g][2] + sa[g][3]), (sq[g][0] + sq[g][1]) + (sq[g][2] + sq[g][3])};
                    const int stat = lane & 1, part = (lane >> 1) % PARTS, c = lane / (2 * PARTS);
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < NPER; ++k)        // (each lane starts at its own slot: same-bank reads of the 32 lanes of a stat otherwise)
                        t += scr[2 * (c * RW + part * NPER + (k + (lane >> 1)) % NPER) + stat];
#pragma unroll
                    for (int off = 2; off < 2 * PARTS; off <<= 1) t += __shfl_xor(t, off, 64);
                    if (part == 0) red[((wave % WN) * (MT / UP) + (wm0 + i * 32) / UP + c) * 2 + stat] += t;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        bool done = false;
        if (p.evec && n0 + NT <= L && (long long)L * U * 16 * 4 < (1ll << 31)) {
            typedef std::integral_constant<int, 2> c2; typedef std::integral_constant<int, 4> c4t;
            typedef std::integral_constant<int, 5> c5; typedef std::integral_constant<int, 8> c8;
            done = true;
            if (U == 2 && UP == 2) fast(c2{}, c2{});
            else if (U == 4 && UP == 4) fast(c4t{}, c4t{});
            else if (U == 5 && UP == 8) fast(c5{}, c8{});
            else if (U == 8 && UP == 8) fast(c8{}, c8{});
            else done = false;
        }
        if (!done)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int jh = 0; jh < NI; jh += 2) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int R = F::row(e, hk), r = R % UP;
                        if (r < U) scr[(R / UP) * ORS + U * (jj * 32 + lr) + r] = acc[i][jh + jj][e];
                    }
                __builtin_amdgcn_sched_barrier(0);
                const int q0 = n0 + wn0 + jh * 32;         // first input position of this pass
#pragma unroll 1
                for (int cl = 0; cl < nco; ++cl) {
                    const int col = wm0 + i * 32 + cl * UP;             // virtual row of the channel's phase 0 inside the tile
                    const int co = (m0 + col) / UP;
                    const float bias = etab[col];
                    const size_t dsto = ((size_t)b * CoutR + co) * Lout + (size_t)U * q0;
                    float s1 = 0.f, s2 = 0.f;
                    for (int c4 = lane; c4 < C4; c4 += 64) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(scr + cl * ORS + 4 * c4);
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            v[x] += bias;
                            if (U * q0 + 4 * c4 + x < Lout) { s1 += v[x]; s2 = fmaf(v[x], v[x], s2); }
                        }
                        if (p.evec) {
                            if (U * q0 + 4 * c4 < Lout) st4(p.out, dsto + 4 * c4, v);
                        } else {
#pragma unroll
                            for (int x = 0; x < 4; ++x)
                                if (U * q0 + 4 * c4 + x < Lout) st1(p.out, dsto + 4 * c4 + x, v[x]);
                        }
                    }
                    if (p.stats_part) {
                        s1 = v2w_wave_sum(s1); s2 = v2w_wave_sum(s2);
                        if (lane == 0) {
                            float* rd = red + ((wave % WN) * (MT / UP) + col / UP) * 2;
                            rd[0] += s1; rd[1] += s2;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (p.stats_part) {
            __syncthreads();
            for (int c = tid; c < MT / UP; c += NTHREADS) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) { t1 += red[(w * (MT / UP) + c) * 2]; t2 += red[(w * (MT / UP) + c) * 2 + 1]; }
                gptr<float>(p.stats_part)[((size_t)tile * CoutR + m0 / UP + c) * 2 + 0] = t1;
                gptr<float>(p.stats_part)[((size_t)tile * CoutR + m0 / UP + c) * 2 + 1] = t2;
            }
        }
        V2W_STAMP(27);
        return;
    }

    // ---- epilogue (as in v2w_conv_mfma.hip): 64 columns of one 32-row block at a time through a wave-private LDS scratch, back as
    // float4s along positions: 16-byte residual / addend loads and stores in a rolled loop
    constexpr bool MASK = EPI == 1;
    constexpr int ERS = 64, C4 = ERS / 4, NIT = 32 * C4 / 64, GV = 4;
    __syncthreads();
    float* const scr = reinterpret_cast<float*>(smem_b) + wave * (32 * ERS);
    const float dinv = p.out_div != 0.f ? 1.f / p.out_div : 1.f;
    const bool simple = p.evec && !p.accumulate && !p.add0 && !p.add1 && !(MASK && p.mask_src);

    // Pipelined form: a pass's residual / addend loads are requested NB passes ahead (the first NB before any store), held as loaded
    // (bf16 pairs stay packed) and converted when used.  vmcnt counts loads and stores in issue order, so a pass that requests its
    // operands after the previous pass's stores waits for a load round trip AND those stores' acknowledgements: at 2 workgroups per
    // CU that was 40 k of the 129 k cycles of a 128 x 256 tile (C = 128, k = 7).
    typedef typename std::conditional<OUT_BF, u32x2, f32x4>::type raw_t;
    constexpr int NPASS = MI * (NI / 2);
    const bool extra = p.accumulate || p.add0 || p.add1;
    auto pipelined = [&](auto ns_c, auto nb_c) {
        constexpr int NS = decltype(ns_c)::value, NB = decltype(nb_c)::value < NPASS ? decltype(nb_c)::value : NPASS;
        const float* const sp[3] = {p.res, p.accumulate ? p.out : p.add0, p.add1};
        raw_t rr[NB][NS][NIT];
        // addresses: element g of a pass is row (lane >> 4) + 4 g, positions 4 (lane & 15) .. + 3 of the pass's 32 x 64 block, i.e. a
        // wave-uniform base (scalar registers, scalar arithmetic) + ONE per-lane byte offset that does not depend on g
        constexpr int ES = OUT_BF ? 2 : 4;
        const int lrow = lane >> 4, lc4 = lane & 15;
        auto pass_base = [&](int ps) {                           // element offset of (row 0, position 0) of pass ps: uniform
            const int i = ps / (NI / 2), jh = (ps % (NI / 2)) * 2;
            return ((size_t)b * p.Cout + m0 + wm0 + i * 32) * L + n0 + wn0 + jh * 32;
        };
        auto lane_off = [&](int ps) {                            // bytes; lanes past the end of the sequence re-read position 0 of the pass
            const int jh = (ps % (NI / 2)) * 2;
            const int qb = n0 + wn0 + jh * 32;
            unsigned o = (unsigned)(lrow * L + (qb + 4 * lc4 < L ? 4 * lc4 : 0)) * ES;
            asm volatile("" : "+v"(o));
            return o;
        };
        auto pass_live = [&](int ps) { return n0 + wn0 + (ps % (NI / 2)) * 64 < L; };     // (uniform) the pass has positions inside L
        auto request = [&](raw_t (&dst)[NS][NIT], int ps, int g) {        // element g of pass ps, every stream
            if (!pass_live(ps)) return;
            const size_t eb = pass_base(ps) + (size_t)(4 * g) * L;
            const unsigned vo = lane_off(ps);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!sp[s]) continue;                                     // (uniform)
                const unsigned char* base = reinterpret_cast<const unsigned char*>(sp[s]) + eb * ES;
                dst[s][g] = *gptr<const raw_t>(base + vo);
            }
        };
        auto widen = [&](raw_t r) { if constexpr (OUT_BF) return bf4_to_f32(r); else return r; };
#pragma unroll
        for (int ps = 0; ps < NB; ++ps)
#pragma unroll
            for (int g = 0; g < NIT; ++g) request(rr[ps], ps, g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int i = ps / (NI / 2), jh = (ps % (NI / 2)) * 2;
            const int cbase = wm0 + i * 32;
            const int qb = n0 + wn0 + jh * 32;
            if (!pass_live(ps)) continue;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) scr[F::row(e, hk) * ERS + jj * 32 + lr] = acc[i][jh + jj][e];
            __builtin_amdgcn_sched_barrier(0);
            const unsigned vo = lane_off(ps);
            const bool ok = qb + 4 * lc4 < L;
#pragma unroll
            for (int g = 0; g < NIT; ++g) {
                const int row = lrow + 4 * g;
                const int col = cbase + row;
                const float bias = etab[col], ra = etab[MT + col], rs = etab[2 * MT + col];
                f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * ERS + 4 * lc4);
                f32x4 r0 = zero4, r1 = zero4, r2 = zero4;
                if (sp[0]) r0 = widen(rr[ps % NB][0][g]);
                if constexpr (NS > 1) {
                    if (sp[1]) r1 = widen(rr[ps % NB][1][g]);
                    if (sp[2]) r2 = widen(rr[ps % NB][2][g]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ps + NB < NPASS) request(rr[ps % NB], ps + NB, g);     // the slot's next occupant, ahead of this element's store
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float t2 = v[x] + bias;
                    if (sp[0]) t2 += fmaf(ra, r0[x], rs);
                    if constexpr (NS > 1) {
                        if (sp[2]) t2 += r1[x] + r2[x];
                        else if (sp[1]) t2 += r1[x];
                    }
                    if (p.out_div != 0.f) t2 = v2w_div_by(t2, p.out_div, dinv);
                    v[x] = t2;
                }
                if (ok) {
                    unsigned char* ob = reinterpret_cast<unsigned char*>(p.out) + (pass_base(ps) + (size_t)(4 * g) * L) * ES;
                    if constexpr (OUT_BF) *gptr<u32x2>(ob + vo) = f32_to_bf4(v);
                    else *gptr<f32x4>(ob + vo) = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<int, 1> one_t;
    typedef std::integral_constant<int, 3> three_t;
    if (p.evec && !(MASK && p.mask_src) && (!extra || OUT_BF)) {
        if (!extra) pipelined(one_t{}, std::integral_constant<int, OUT_BF ? 4 : 2>{});
        else pipelined(three_t{}, std::integral_constant<int, 1>{});
        V2W_STAMP(27);
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int cbase = wm0 + i * 32;
#pragma unroll
        for (int jh = 0; jh < NI; jh += 2) {
            const size_t gbase = ((size_t)b * p.Cout + m0 + cbase) * L + n0 + wn0 + jh * 32;
            const int qb = n0 + wn0 + jh * 32;              // first position of this pass
            f32x4 rall[NIT];
            if (simple) {
#pragma unroll
                for (int g = 0; g < NIT; ++g) {
                    const int idx = lane + 64 * g;
                    const int row = idx / C4, c4 = idx - row * C4;
                    rall[g] = zero4;
                    if (p.res && qb + 4 * c4 < L) rall[g] = ld4(p.res, gbase + (size_t)row * L + 4 * c4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) scr[F::row(e, hk) * ERS + jj * 32 + lr] = acc[i][jh + jj][e];
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
                        float t2 = v[x] + bias;
                        if (p.res) t2 += fmaf(ra, rall[g][x], rs);
                        if (p.out_div != 0.f) t2 = v2w_div_by(t2, p.out_div, dinv);
                        v[x] = t2;
                    }
                    if (qb + 4 * c4 < L) st4(p.out, gbase + (size_t)row * L + 4 * c4, v);
                }
            } else if (p.evec) {
#pragma unroll 1
                for (int g0 = 0; g0 < NIT; g0 += GV) {
                    f32x4 rv[GV], ov[GV], o2[GV], mv[MASK ? GV : 1];
#pragma unroll
                    for (int g = 0; g < GV; ++g) {
                        const int idx = lane + 64 * (g0 + g);
                        const int row = idx / C4, c4 = idx - row * C4;
                        const bool ok = qb + 4 * c4 < L;
                        const size_t goff = gbase + (size_t)row * L + 4 * c4;
                        rv[g] = ov[g] = o2[g] = zero4;
                        if (ok) {
                            if (p.res) rv[g] = ld4(p.res, goff);
                            if (p.accumulate) ov[g] = ld4(p.out, goff);
                            else if (p.add0) ov[g] = ld4(p.add0, goff);
                            if (p.add1) o2[g] = ld4(p.add1, goff);
                        }
                        if constexpr (MASK) {
                            mv[g] = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (ok && p.mask_src) mv[g] = *gptr<const f32x4>(p.mask_src + goff);
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
                            float t2 = v[x];
                            if constexpr (MASK)
                                if (p.mask_src) t2 = fmaf(etab[3 * MT + col], mv[g][x], etab[4 * MT + col]) > 0.f ? t2 : t2 * p.mask_slope;
                            t2 += bias;
                            if (p.res) t2 += fmaf(ra, rv[g][x], rs);
                            if (p.add1) t2 += ov[g][x] + o2[g][x];
                            else if (p.accumulate || p.add0) t2 += ov[g][x];
                            if (p.out_div != 0.f) t2 = v2w_div_by(t2, p.out_div, dinv);
                            v[x] = t2;
                        }
                        if (qb + 4 * c4 < L) st4(p.out, gbase + (size_t)row * L + 4 * c4, v);
                    }
                }
            } else {                                  // ragged L / unaligned operands: one element at a time
#pragma unroll 1
                for (int idx = lane; idx < 32 * ERS; idx += 64) {
                    const int row = idx / ERS, c = idx - row * ERS;
                    if (qb + c >= L) continue;
                    const int col = cbase + row;
                    const size_t goff = gbase + (size_t)row * L + c;
                    float t2 = scr[idx];
                    if constexpr (MASK)
                        if (p.mask_src) t2 = fmaf(etab[3 * MT + col], gptr<const float>(p.mask_src)[goff], etab[4 * MT + col]) > 0.f ? t2 : t2 * p.mask_slope;
                    t2 += etab[col];
                    if (p.res) t2 += fmaf(etab[MT + col], ld1(p.res, goff), etab[2 * MT + col]);
                    if (p.add1) t2 += ld1(p.add0, goff) + ld1(p.add1, goff);
                    else if (p.accumulate) t2 += ld1(p.out, goff);
                    else if (p.add0) t2 += ld1(p.add0, goff);
                    if (p.out_div != 0.f) t2 = v2w_div_by(t2, p.out_div, dinv);
                    st1(p.out, goff, t2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    V2W_STAMP(27);
}

template <int MI, int NI, int WM, int WN, int CK = V2W_BF_CK, bool VEC = true>
int launch_bf16(const TileArgs* ps, int nprob, hipStream_t stream, int32_t* cfg = nullptr) {
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, NTHREADS = 64 * WM * WN, HMAX = 32, ROWB = 2 * CK + 16;
    constexpr int NPF = ((CK / 4) * ((NT + 2 * HMAX) / 4) + NTHREADS - 1) / NTHREADS;
    static_assert(NI % 2 == 0, "the epilogue walks column blocks in pairs");
    MultiArgs m{};
    size_t lds = 0;
    int grid = 0, epi = 0;
    for (int i = 0; i < nprob; ++i) {
        TileArgs p = ps[i];
        if (p.Cout % MT != 0 || p.Cin % CK != 0) return V2W_E_SHAPE;
        p.hla = (p.hl + 3) & ~3;
        if (p.hla > HMAX || p.hr > HMAX) return V2W_E_SHAPE;
        p.ntl = (p.L + NT - 1) / NT;
        p.ntiles = p.B * p.ntl;
        p.xrows = (p.hla + NT + p.hr + 3) & ~3;
        p.vec4 = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0);
        if (VEC && !p.vec4) return V2W_E_ARG;                  // (the dispatcher sends unaligned inputs to the VEC = false instantiation)
        const int nbuf = p.Cin / CK > 1 ? 2 : 1;
        int tab = nbuf * p.xrows * ROWB / 4;                        // float index of the tables, after the signal buffers ...
        if (tab < WM * WN * 32 * 64) tab = WM * WN * 32 * 64;              // ... and after the epilogue scratch that overlays them
        p.atab_off = tab;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        p.evec = p.L % 4 == 0 && al16(p.out) && al16(p.res) && al16(p.add0) && al16(p.add1) && al16(p.mask_src);
        const size_t l = ((size_t)tab + 5 * MT + 2 * p.Cin) * sizeof(float);
        if (l > lds) lds = l;
        if (p.mask_src) epi = 1;
        m.p[i] = p;
        m.start[i] = grid;
        grid += ((p.ntiles + 7) / 8) * 8 * (p.Cout / MT);
    }
    m.start[nprob] = grid;
    for (int i = nprob + 1; i <= V2W_MAX_MULTI; ++i) m.start[i] = 0x7fffffff;
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    const int io = m.p[0].io_bf16;
    for (int i = 1; i < nprob; ++i) if (m.p[i].io_bf16 != io) return V2W_E_ARG;      // one instantiation per launch
    if (io == 1 || (epi && io != 0)) return V2W_E_SHAPE;     // bf16 in with fp32 out does not occur on the path; the mask epilogue is fp32-only
    if (cfg) {      // configuration query: the template arguments of the instantiation this launch would run
        const int32_t c[10] = {MI, NI, WM, WN, NPF, epi, io == 3, io >= 2, CK, VEC};
        for (int i = 0; i < 10; ++i) cfg[i] = c[i];
        return 0;
    }
    auto kern = epi ? conv_bf16_kernel<MI, NI, WM, WN, NPF, 1, false, false, CK, VEC>
              : io == 0 ? conv_bf16_kernel<MI, NI, WM, WN, NPF, 0, false, false, CK, VEC>
              : io == 2 ? conv_bf16_kernel<MI, NI, WM, WN, NPF, 0, false, true, CK, VEC>
                        : conv_bf16_kernel<MI, NI, WM, WN, NPF, 0, true, true, CK, VEC>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(NTHREADS), lds, stream, m);
    return v2w_launch_status();
}

// Virtual 3-tap (in general hl + hr + 1 tap) conv weights of a transposed conv, as bf16 fragments in the layout of pack_bf16_kernel:
// Wv[tv][ci][co * UP + r] = wf[t0_r + m * U][ci][co] with m = c_r - (tv - hl), t0_r = (r + pad) % U, c_r = (r + pad) / U (0 where the
// phase has no such tap, and for the padding phases r >= U).
__global__ void __launch_bounds__(256)
pack_bf16_convt_kernel(const float* __restrict__ wf, b8* __restrict__ wps, int K, int Cin, int Cout, int U, int UP, int hl, int KV) {
    const int nch = Cin / 16, pad = (K - U) / 2;
    const size_t total = (size_t)(Cout * UP / 32) * nch * KV * 64;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int lane = o & 63;
        size_t rest = o >> 6;
        const int tv = rest % KV; rest /= KV;
        const int ch = rest % nch;
        const int mb = rest / nch;
        const int R = mb * 32 + (lane & 31), co = R / UP, r = R % UP;
        const int c0 = ch * 16 + 8 * (lane >> 5);
        const int rp = r + pad, t0 = rp % U, cr = rp / U, mm = cr - (tv - hl), t = t0 + mm * U;
        const bool ok = r < U && mm >= 0 && t < K;
        b8 hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) hi[j] = (__bf16)(ok ? wf[((size_t)t * Cin + c0 + j) * Cout + co] : 0.f);
        wps[(((size_t)(mb * nch + ch) * KV + tv) * 2) * 64 + lane] = hi;
    }
}

}  // namespace
int v2w_convt1d_bf16_res(const v2w_convt1d_args* a, int UP, int hl, int KV, hipStream_t stream, int* ntiles_out, int32_t* cfg);   // v2w_convt_bf16_res.hip
namespace {

struct ConvtGeom { int UP, hl, hr, KV; };
static ConvtGeom convt_geom(int k, int u) {
    ConvtGeom g{1, 0, 0, 0};
    while (g.UP < u) g.UP *= 2;
    const int pad = (k - u) / 2;
    for (int r = 0; r < u; ++r) {
        const int rp = r + pad, t0 = rp % u, c = rp / u, nt = (k - t0 + u - 1) / u;
        if (c > g.hr) g.hr = c;
        if (nt - 1 - c > g.hl) g.hl = nt - 1 - c;
    }
    g.KV = g.hl + g.hr + 1;
    return g;
}

// v2w_pack_bf16_convt's buffer carries a second region (rows co * u + phase, no padding phases) when the stride is no power of two
static bool convt_exact_region(int c_out, int u, int UP) { return UP != u && (c_out * u) % 32 == 0; }

template <int MI, int NI, int WM, int WN, bool VEC = true>
int launch_bf16_convt(TileArgs p, hipStream_t stream, int* ntiles_out, int32_t* cfg = nullptr) {
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, NTHREADS = 64 * WM * WN, HMAX = 32;
    constexpr int NPF = (8 * ((NT + 2 * HMAX) / 4) + NTHREADS - 1) / NTHREADS;
    if (cfg) {
        const int32_t c[10] = {MI, NI, WM, WN, NPF, 2, p.io_bf16 != 0, p.io_bf16 != 0, V2W_BF_CK, VEC};
        for (int i = 0; i < 10; ++i) cfg[i] = c[i];
    }
    if (p.Cout % MT != 0 || p.Cin % V2W_BF_CK != 0) return V2W_E_SHAPE;
    p.hla = (p.hl + 3) & ~3;
    if (p.hla > HMAX || p.hr > HMAX) return V2W_E_SHAPE;
    p.ntl = (p.L + NT - 1) / NT;
    p.ntiles = p.B * p.ntl;
    p.xrows = (p.hla + NT + p.hr + 3) & ~3;
    p.vec4 = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0);
    p.evec = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
    const int nbuf = p.Cin / V2W_BF_CK > 1 ? 2 : 1;
    int tab = nbuf * p.xrows * V2W_BF_ROWB / 4;
    if (tab < WM * WN * 2048) tab = WM * WN * 2048;
    p.atab_off = tab;
    const size_t lds = ((size_t)tab + 5 * MT + 2 * p.Cin + WN * (MT / p.up_p) * 2) * sizeof(float);
    // every shape check stands in front of the query's answer: a configuration / tile-count query that says yes must mean the launch
    // will not decline (Generator._bf16_storage_kernels_exist decides on it before the forward starts)
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    if (p.io_bf16 != 0 && p.io_bf16 != 3) return V2W_E_SHAPE;
    if (VEC && !p.vec4) return V2W_E_ARG;                     // (the dispatcher sends unaligned inputs to the VEC = false instantiation)
    if (ntiles_out) { *ntiles_out = p.ntiles; return 0; }
    MultiArgs m{};
    m.p[0] = p;
    m.start[0] = 0;
    const int grid = ((p.ntiles + 7) / 8) * 8 * (p.Cout / MT);
    m.start[1] = grid;
    for (int i = 2; i <= V2W_MAX_MULTI; ++i) m.start[i] = 0x7fffffff;
    auto kern = p.io_bf16 ? conv_bf16_kernel<MI, NI, WM, WN, NPF, 2, true, true, V2W_BF_CK, VEC> : conv_bf16_kernel<MI, NI, WM, WN, NPF, 2, false, false, V2W_BF_CK, VEC>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(NTHREADS), lds, stream, m);
    return v2w_launch_status();
}

static int convt_bf16_dispatch(const v2w_convt1d_args* a, hipStream_t stream, int* ntiles_out, int32_t* cfg = nullptr) {
    if (a->B <= 0 || a->C_in <= 0 || a->C_out <= 0 || a->L <= 0 || a->k <= 0 || a->u <= 1) return V2W_E_ARG;
    if (a->k < a->u || ((a->k - a->u) & 1) || a->u > 8) return V2W_E_SHAPE;
    if (a->C_in % V2W_BF_CK != 0) return V2W_E_SHAPE;
    const ConvtGeom g = convt_geom(a->k, a->u);
    TileArgs p{};
    p.in = a->in; p.wps = a->wp; p.bias = a->bias; p.out = a->out; p.stats_part = a->stats_part;
    p.B = a->B; p.Cin = a->C_in; p.Cout = a->C_out * g.UP; p.L = a->L; p.K = g.KV; p.dil = 1;
    p.hl = g.hl; p.hr = g.hr; p.slope = a->slope; p.up_u = a->u; p.up_p = g.UP; p.io_bf16 = a->io_bf16;
    if (a->io_bf16 == 3 && (g.UP == a->u || a->u == 5)) {   // bf16 tensors: the resident-tile kernel (v2w_convt_bf16_res.hip; stride 5: whole tiles)
        const int rc = v2w_convt1d_bf16_res(a, g.UP, g.hl, g.KV, stream, ntiles_out, cfg);
        if (rc != V2W_E_SHAPE) return rc;
    }
    const int rows = p.Cout;
    // the stats tiling (rows of stats_part) depends on the tile width only: every configuration here is 256 or 512 positions wide, the
    // element-wise-staging fallback (unaligned input or a length that is not a multiple of 4) uses the widths of its aligned twin
    const bool vec = (a->L % 4 == 0) && ((reinterpret_cast<uintptr_t>(a->in) & 15) == 0);
    if (!vec) {     // (queries too: the answer comes from the instantiation that will launch, with every one of its checks)
        if (rows % 64 == 0 && !(rows % 128 == 0 && (long)a->B * ((a->L + 255) / 256) * (rows / 128) >= 512)) return launch_bf16_convt<1, 4, 2, 2, false>(p, stream, ntiles_out, cfg);
        if (rows % 128 == 0) return launch_bf16_convt<2, 4, 2, 2, false>(p, stream, ntiles_out, cfg);
        if (rows % 32 == 0) return launch_bf16_convt<1, 4, 1, 4, false>(p, stream, ntiles_out, cfg);
        return V2W_E_SHAPE;
    }
    if (rows % 128 == 0 && (long)a->B * ((a->L + 255) / 256) * (rows / 128) >= 512) return launch_bf16_convt<2, 4, 2, 2>(p, stream, ntiles_out, cfg);
    if (rows % 64 == 0) return launch_bf16_convt<1, 4, 2, 2>(p, stream, ntiles_out, cfg);
    if (rows % 32 == 0) return launch_bf16_convt<1, 4, 1, 4>(p, stream, ntiles_out, cfg);
    return V2W_E_SHAPE;
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_bf16)
#endif

// Called by v2w_conv1d_split for V2W_ALGO_BF16.  V2W_E_SHAPE: the caller falls back to the split kernel's bf16 form.
int v2w_conv1d_bf16(const v2w_conv1d_args* a, int n, hipStream_t stream, int32_t* cfg) {
    if (n < 1 || n > V2W_MAX_MULTI) return V2W_E_ARG;
    const bool c32 = a->C_out == 32;            // 32 output channels (the narrow stage of a generator trained in the bf16 arithmetic: input gradients on fp32 tensors)
    if (a->C_in % V2W_BF_CK != 0 || (a->C_out % 64 != 0 && !c32) || a->k < 1) return V2W_E_SHAPE;
    TileArgs ps[V2W_MAX_MULTI];
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        const v2w_conv1d_args* q = a + i;
        if (q->B != a->B || q->C_in != a->C_in || q->C_out != a->C_out || q->L != a->L) return V2W_E_SHAPE;
        if (!q->wps && !cfg) return V2W_E_ARG;
        if (q->in_stride > 1 || q->in_ct > 0 || q->out_ct > 0) return V2W_E_SHAPE;
        TileArgs p{};
        p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.wps = q->wps; p.winv = q->winv; p.bias = q->bias;
        p.res = q->res; p.res_a = q->res_a; p.res_s = q->res_s; p.out = q->out;
        p.add0 = q->add0; p.add1 = q->add1;
        p.mask_src = q->mask_src; p.mask_a = q->mask_a; p.mask_s = q->mask_s; p.mask_slope = q->mask_slope;
        p.B = q->B; p.Cin = q->C_in; p.Cout = q->C_out; p.L = q->L; p.K = q->k; p.dil = q->dil;
        p.hl = p.hr = q->dil * (q->k - 1) / 2;
        if (q->pad_left >= 0) { p.hl = q->pad_left; p.hr = q->dil * (q->k - 1) - q->pad_left; if (p.hr < 0) return V2W_E_ARG; }
        p.slope = q->slope; p.accumulate = q->accumulate; p.out_div = q->out_div;
        p.io_bf16 = q->io_bf16;
        ps[i] = p;
        tiles += (long)p.B * ((p.L + 255) / 256) * (p.Cout / 64);
    }
    if (c32) {                           // one 32-row block x 256 positions per workgroup, aligned fp32 tensors only
        for (int i = 0; i < n; ++i)
            if (a[i].L % 4 != 0 || (reinterpret_cast<uintptr_t>(a[i].in) & 15) != 0 || a[i].io_bf16 != 0) return V2W_E_SHAPE;
        return launch_bf16<1, 2, 1, 4>(ps, n, stream, cfg);
    }
    for (int i = 0; i < n; ++i)          // unaligned input or L % 4 != 0: element-wise staging, one small-tile instantiation serves every shape
        if (a[i].L % 4 != 0 || (reinterpret_cast<uintptr_t>(a[i].in) & 15) != 0) return launch_bf16<1, 2, 2, 2, V2W_BF_CK, false>(ps, n, stream, cfg);
    if (a->C_out % 128 == 0 && tiles >= 2 * 512) return launch_bf16<2, 4, 2, 2>(ps, n, stream, cfg);     // 128 x 256
    if (tiles >= 256 && a->C_in == 64 && a->io_bf16 == 3) return launch_bf16<1, 4, 2, 2, 64>(ps, n, stream, cfg);   // 64 x 256, bf16 tensors: the 64 input channels as ONE chunk
    // a deep layer (conv_pre: 768 / 1024 input channels x 7 taps) on a grid of one 64 x 256 tile per CU or less walks 24+ chunks behind a
    // barrier each: 64 x 128 tiles (twice the workgroups) with 64-channel chunks (half the barriers) - 135-143 -> 88-98 us at B = 32 x T = 256
    if (tiles >= 128 && tiles < 512 && n == 1 && a->C_in >= 512 && a->C_in % 64 == 0 && a->io_bf16 == 2) {
        // (round 6) from one 64 x 256 tile per CU up: 128 x 128 tiles on EIGHT waves (4 x 2, a 32 x 64 block each) - at B = 32 x T = 256 exactly
        // one workgroup per CU with two waves per SIMD to cover the chunk loop's latencies: 80 -> 69 us (the same tile on four waves: 75); below
        // that (BASELINE configs[4] at B = 16: 64 workgroups of 128 x 128) the small tile stays (76 against 88 us)
        if (tiles >= 256 && a->C_out % 128 == 0) return launch_bf16<1, 2, 4, 2, 64>(ps, n, stream, cfg);
        return launch_bf16<1, 2, 2, 2, 64>(ps, n, stream, cfg);
    }
    if (tiles >= 256) return launch_bf16<1, 4, 2, 2>(ps, n, stream, cfg);                                 // 64 x 256
    return launch_bf16<1, 2, 2, 2>(ps, n, stream, cfg);                                                   // 64 x 128: latency sizes
}

// ---- bf16 transposed conv (V2W_ALGO_BF16 counterpart of v2w_convt1d_fwd); a->wp = the fragments of v2w_pack_bf16_convt
extern "C" int v2w_pack_bf16_convt(const float* wf, void* wps, int k, int c_in, int c_out, int u, void* stream) {
    if (!wf || !wps || k <= 0 || c_in <= 0 || c_out <= 0 || u <= 1) return V2W_E_ARG;
    if (k < u || ((k - u) & 1) || u > 8 || c_in % 16 != 0) return V2W_E_SHAPE;
    const ConvtGeom g = convt_geom(k, u);
    if ((c_out * g.UP) % 32 != 0) return V2W_E_SHAPE;
    const size_t total = (size_t)(c_out * g.UP / 32) * (c_in / 16) * g.KV * 64;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    V2W_LAUNCH(pack_bf16_convt_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, wf, reinterpret_cast<b8*>(wps), k, c_in, c_out, u,
                       g.UP, g.hl, g.KV);
    if (convt_exact_region(c_out, u, g.UP)) {
        // a stride that is no power of two (5): the same taps again over the EXACT phases, rows co * u + phase, behind the padded rows - the
        // resident kernel of v2w_convt_bf16_res.hip then spends no MFMA on the dead phases
        const size_t total2 = (size_t)(c_out * u / 32) * (c_in / 16) * g.KV * 64;
        int grid2 = (int)((total2 + 255) / 256); if (grid2 > 4096) grid2 = 4096;
        V2W_LAUNCH(pack_bf16_convt_kernel, dim3(grid2), dim3(256), 0, (hipStream_t)stream, wf,
                           reinterpret_cast<b8*>(reinterpret_cast<unsigned char*>(wps) + total / 64 * V2W_BF_UNIT), k, c_in, c_out, u, u, g.hl, g.KV);
    }
    return v2w_launch_status();
}
// bytes of the fragment buffer of v2w_pack_bf16_convt (0: shape not supported)
extern "C" long long v2w_pack_bf16_convt_bytes(int k, int c_in, int c_out, int u) {
    if (k <= 0 || c_in <= 0 || c_out <= 0 || u <= 1 || k < u || ((k - u) & 1) || u > 8 || c_in % 32 != 0) return 0;
    const ConvtGeom g = convt_geom(k, u);
    if ((c_out * g.UP) % 32 != 0) return 0;
    return (long long)(c_out * g.UP / 32 + (convt_exact_region(c_out, u, g.UP) ? c_out * u / 32 : 0)) * (c_in / 16) * g.KV * V2W_BF_UNIT;
}
extern "C" int v2w_convt1d_bf16_fwd(const v2w_convt1d_args* a, void* stream) {
    if (!a || !a->in || !a->wp || !a->out) return V2W_E_ARG;
    return convt_bf16_dispatch(a, (hipStream_t)stream, nullptr);
}
// rows of `stats_part` ([rows][C_out][2]) that v2w_convt1d_bf16_fwd fills for this problem; < 0: error / unsupported shape
// cfg[10] = MI, NI, WM, WN, NPF, EPI, IN_BF, OUT_BF, CK, VEC of the conv_bf16_kernel instantiation the call would launch (host-only
// queries, no pointer is dereferenced beyond the alignment test of the vector staging path)
extern "C" int v2w_convt1d_bf16_config(const v2w_convt1d_args* a, int32_t* cfg) {
    if (!a || !cfg) return V2W_E_ARG;
    int n = 0;
    return convt_bf16_dispatch(a, nullptr, &n, cfg);
}
extern "C" int v2w_conv1d_bf16_config(const v2w_conv1d_args* a, int n, int32_t* cfg) {
    if (!a || !cfg) return V2W_E_ARG;
    return v2w_conv1d_bf16(a, n, nullptr, cfg);
}
extern "C" int v2w_convt1d_bf16_tiles(const v2w_convt1d_args* a) {
    if (!a) return V2W_E_ARG;
    int n = 0;
    const int rc = convt_bf16_dispatch(a, nullptr, &n);
    return rc == 0 ? n : rc;
}
