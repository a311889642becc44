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
// The older split kernel (v2w_conv_split.hip, one workgroup barrier per (16-channel chunk, tap) stage) stays the f16x3 path; in bf16
// mode a stage of it was 128 cycles of MFMA issue in ~1300 cycles.
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

template <int MI, int NI, int WM, int WN, int NPF, int EPI>
__global__ void __launch_bounds__(64 * WM * WN, MI * NI >= 8 ? 2 : 1)      // (128 x 256: two workgroups per CU = at most 256 registers)
conv_bf16_kernel(const MultiArgs m) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTHREADS = 64 * WM * WN;
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, CK = V2W_BF_CK, ROWB = V2W_BF_ROWB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // 2 x [xrows][ROWB] signal tiles, then the float tables

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

    // ---- staging: an item = 4 consecutive channels x 4 positions: four float4 loads, four 8-byte LDS stores (the 4 channels of one
    // position are 8 contiguous bytes of its row).  Consecutive lanes take the 8 channel quads of one position quad (a whole 64-byte
    // row per 8 lanes: conflict-free stores; 128-byte pieces of 8 x 4 rows on the global side), then the next position quad.
    const int nq = p.xrows >> 2;
    f32x4 pf[NPF][4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto item = [&](int s, int& cq, int& row, bool& in_img, bool& in_seq) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int idx = t + s * NTHREADS;
        cq = idx & 7;
        row = (idx >> 3) * 4;
        in_img = (idx >> 3) < nq;
        const int pos = pos0 + row;
        in_seq = in_img && pos >= 0 && pos < L;         // L % 4 == 0 and pos % 4 == 0: a float4 is inside or outside as a whole
    };
    auto prefetch = [&](int ci0) {
        const float* src = p.in + (size_t)(b * p.Cin + ci0) * L + pos0;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            int cq, row; bool in_img, in_seq;
            item(s, cq, row, in_img, in_seq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                pf[s][i] = zero4;
                if (in_seq) pf[s][i] = *reinterpret_cast<const f32x4*>(src + (size_t)(4 * cq + i) * L + row);
            }
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
                av[i] = p.in_a ? atab[ci0 + 4 * cq + i] : 1.f;
                sv[i] = p.in_a ? atab[p.Cin + ci0 + 4 * cq + i] : 0.f;
            }
            unsigned char* dst = Xs + row * ROWB + cq * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x2 v = {0u, 0u};                     // padding stays exactly 0 (it pads the ACTIVATED signal)
                if (in_seq) {
                    float a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = v2w_lrelu(fmaf(av[i], pf[s][i][e], sv[i]), slope);
                    v[0] = pack_bf16x2(a[0], a[1]); v[1] = pack_bf16x2(a[2], a[3]);
                }
                *reinterpret_cast<u32x2*>(dst + e * ROWB) = v;
            }
        }
    };
    auto stage_scalar = [&](int ci0, unsigned char* Xs) {   // any L / alignment: one element at a time
        for (int c = wave; c < CK; c += WM * WN) {
            const int ch = b * p.Cin + ci0 + c;
            const float* src = p.in + (size_t)ch * L;
            const float av = p.in_a ? p.in_a[ch] : 1.f, sv = p.in_s ? p.in_s[ch] : 0.f;
            for (int j = lane; j < p.xrows; j += 64) {
                const int l = pos0 + j;
                float v = 0.f;
                if (l >= 0 && l < L) v = v2w_lrelu(fmaf(av, src[l], sv), slope);
                reinterpret_cast<__bf16*>(Xs + j * ROWB)[c] = (__bf16)v;
            }
        }
    };

    // ---- weights: fragment (row block, 16-channel k-step c16, tap t) sits at ((rb * nst + c16 * K + t) * V2W_BF_UNIT) + lane * 16
    const int nst = 2 * nch * K;                             // fragments per row block
    const unsigned char* ap[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
        ap[i] = reinterpret_cast<const unsigned char*>(p.wps) + (size_t)((m0 + wm0) / 32 + i) * nst * V2W_BF_UNIT;
    const unsigned lane16 = (unsigned)lane * 16u;
    u32x4 ar[2][MI];                                         // [k-step of the tap][row block]: one computing, one in flight
    auto load_frag = [&](u32x4 (&a)[MI], int ch, int s, int t) {   // k-step s of (chunk ch, tap t); clamped past the end (harmless re-read)
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));                        // keeps the address scalar base + 32-bit lane offset (see v2w_conv_mfma.hip)
        const int chc = ch < nch ? ch : nch - 1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const u32x4*>(ap[i] + (size_t)((2 * chc + s) * K + t) * V2W_BF_UNIT + l16);
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

    // ---- prologue
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
    if (p.vec4) { prefetch(0); commit(0, smem_b); }
    else stage_scalar(0, smem_b);
    load_frag(ar[0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();

    const int lbase = (wn0 + lr + p.hla - p.hl) * ROWB + 16 * hk;     // this lane's 16 bytes in the row of (its column, tap 0), k-step 0
    const int step = p.dil * ROWB;
    for (int ch = 0; ch < nch; ++ch) {
        const unsigned char* Xs = smem_b + (ch & 1) * bufsz;
        unsigned char* Xn = smem_b + ((ch + 1) & 1) * bufsz;
        const bool more = ch + 1 < nch;
        if (more && p.vec4) prefetch((ch + 1) * CK);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* xt = Xs + lbase;
#pragma unroll
        for (int j = 0; j < NI; ++j) bb[j] = *reinterpret_cast<const u32x4*>(xt + j * 32 * ROWB);
        // a tap = k-step 0 (channels 0-15 of the chunk: bytes 0-31 of a row) from ring slot 0, then k-step 1 (bytes 32-63) from slot 1;
        // each slot's next fragment is requested while the other slot computes; the chunk's last request is tap 0 of the next chunk
        for (int t = 0; t < K; ++t, xt += step) {
            load_frag(ar[1], ch, 1, t);
            __builtin_amdgcn_sched_barrier(0);
            kstep(ar[0], xt + 32);
            if (t + 1 < K) load_frag(ar[0], ch, 0, t + 1); else load_frag(ar[0], ch + 1, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            kstep(ar[1], t + 1 < K ? xt + step : xt);
        }
        if (more) {
            if (p.vec4) commit((ch + 1) * CK, Xn);
            else stage_scalar((ch + 1) * CK, Xn);
            __syncthreads();
        }
    }

    // ---- epilogue (as in v2w_conv_mfma.hip): 64 columns of one 32-row block at a time through a wave-private LDS scratch, back as
    // float4s along positions: 16-byte residual / addend loads and stores in a rolled loop
    constexpr bool MASK = EPI == 1;
    constexpr int ERS = 64, C4 = ERS / 4, NIT = 32 * C4 / 64, GV = 4;
    __syncthreads();
    float* const scr = reinterpret_cast<float*>(smem_b) + wave * (32 * ERS);
    const float dinv = p.out_div != 0.f ? 1.f / p.out_div : 1.f;
    const bool simple = p.evec && !p.accumulate && !p.add0 && !p.add1 && !(MASK && p.mask_src);
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
                    if (p.res && qb + 4 * c4 < L) rall[g] = *reinterpret_cast<const f32x4*>(p.res + gbase + (size_t)row * L + 4 * c4);
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
                    if (qb + 4 * c4 < L) *reinterpret_cast<f32x4*>(p.out + gbase + (size_t)row * L + 4 * c4) = v;
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
                            if (p.res) rv[g] = *reinterpret_cast<const f32x4*>(p.res + goff);
                            if (p.accumulate) ov[g] = *reinterpret_cast<const f32x4*>(p.out + goff);
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
                        if (qb + 4 * c4 < L) *reinterpret_cast<f32x4*>(p.out + gbase + (size_t)row * L + 4 * c4) = v;
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
                        if (p.mask_src) t2 = fmaf(etab[3 * MT + col], p.mask_src[goff], etab[4 * MT + col]) > 0.f ? t2 : t2 * p.mask_slope;
                    t2 += etab[col];
                    if (p.res) t2 += fmaf(etab[MT + col], p.res[goff], etab[2 * MT + col]);
                    if (p.add1) t2 += p.add0[goff] + p.add1[goff];
                    else if (p.accumulate) t2 += p.out[goff];
                    else if (p.add0) t2 += p.add0[goff];
                    if (p.out_div != 0.f) t2 = v2w_div_by(t2, p.out_div, dinv);
                    p.out[goff] = t2;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MI, int NI, int WM, int WN>
int launch_bf16(const TileArgs* ps, int nprob, hipStream_t stream) {
    constexpr int MT = 32 * MI * WM, NT = 32 * NI * WN, NTHREADS = 64 * WM * WN, HMAX = 32;
    constexpr int NPF = (8 * ((NT + 2 * HMAX) / 4) + NTHREADS - 1) / NTHREADS;
    static_assert(NI % 2 == 0, "the epilogue walks column blocks in pairs");
    MultiArgs m{};
    size_t lds = 0;
    int grid = 0, epi = 0;
    for (int i = 0; i < nprob; ++i) {
        TileArgs p = ps[i];
        if (p.Cout % MT != 0 || p.Cin % V2W_BF_CK != 0) return V2W_E_SHAPE;
        p.hla = (p.hl + 3) & ~3;
        if (p.hla > HMAX || p.hr > HMAX) return V2W_E_SHAPE;
        p.ntl = (p.L + NT - 1) / NT;
        p.ntiles = p.B * p.ntl;
        p.xrows = (p.hla + NT + p.hr + 3) & ~3;
        p.vec4 = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.in) & 15) == 0);
        const int nbuf = p.Cin / V2W_BF_CK > 1 ? 2 : 1;
        int tab = nbuf * p.xrows * V2W_BF_ROWB / 4;                        // float index of the tables, after the signal buffers ...
        if (tab < WM * WN * 32 * 64) tab = WM * WN * 32 * 64;              // ... and after the epilogue scratch that overlays them
        p.atab_off = tab;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        p.evec = p.L % 4 == 0 && al16(p.out) && al16(p.res) && al16(p.add0) && al16(p.add1) && al16(p.mask_src);
        const size_t l = ((size_t)tab + 5 * MT + (p.in_a ? 2 * p.Cin : 0)) * sizeof(float);
        if (l > lds) lds = l;
        if (p.mask_src) epi = 1;
        m.p[i] = p;
        m.start[i] = grid;
        grid += ((p.ntiles + 7) / 8) * 8 * (p.Cout / MT);
    }
    m.start[nprob] = grid;
    for (int i = nprob + 1; i <= V2W_MAX_MULTI; ++i) m.start[i] = 0x7fffffff;
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    auto kern = epi ? conv_bf16_kernel<MI, NI, WM, WN, NPF, 1> : conv_bf16_kernel<MI, NI, WM, WN, NPF, 0>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, stream, m);
    return v2w_launch_status();
}

}  // namespace

// Called by v2w_conv1d_split for V2W_ALGO_BF16.  V2W_E_SHAPE: the caller falls back to the split kernel's bf16 form.
int v2w_conv1d_bf16(const v2w_conv1d_args* a, int n, hipStream_t stream) {
    if (n < 1 || n > V2W_MAX_MULTI) return V2W_E_ARG;
    if (a->C_in % V2W_BF_CK != 0 || a->C_out % 64 != 0 || a->k < 1) return V2W_E_SHAPE;
    TileArgs ps[V2W_MAX_MULTI];
    long tiles = 0;
    for (int i = 0; i < n; ++i) {
        const v2w_conv1d_args* q = a + i;
        if (q->B != a->B || q->C_in != a->C_in || q->C_out != a->C_out || q->L != a->L) return V2W_E_SHAPE;
        if (!q->wps) return V2W_E_ARG;
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
        ps[i] = p;
        tiles += (long)p.B * ((p.L + 255) / 256) * (p.Cout / 64);
    }
    if (a->C_out % 128 == 0 && tiles >= 2 * 512) return launch_bf16<2, 4, 2, 2>(ps, n, stream);     // 128 x 256
    if (tiles >= 256) return launch_bf16<1, 4, 2, 2>(ps, n, stream);                                 // 64 x 256
    return launch_bf16<1, 2, 2, 2>(ps, n, stream);                                                   // 64 x 128: latency sizes
}
