// The residual convolutions of a WIDE ResBlock2 stage (C = 64 / 128) on bf16 tensors, one launch per conv position of the block
// instead of one (merged) launch per branch group - BASELINE configs[2] ("bf16 compute / fp32 accumulate", bf16 activation storage):
//
//   mode 0  "first convs":   t1_j = x + conv_{k_j,d1_j}(lrelu(x)) + b1_j   for every branch j, x = a*xr + s (the folded CondBN affine)
//                            (models.py:135-141 with ResBlock2.forward's first loop iteration, models.py:66-69)
//   mode 1  "second convs":  out = ( sum_j [ t1_j + conv_{k_j,d2_j}(lrelu(t1_j)) + b2_j ] ) / nk            (second iteration + the mean)
//
// What differs from conv_bf16_kernel (v2w_conv_bf16.hip), and why - its tile spends 111 k cycles on 14 k cycles of MFMA issue:
//   * the signal tile is RESIDENT: all C input channels x (N + halo) positions are staged once (C = 128: 70 KB in one burst of loads -
//     what it takes to cover a loaded HBM latency of ~4 us), then the MFMA loop runs over every (channel chunk, tap) with NO barrier,
//     no staging and nothing but weight-fragment loads in the wave's in-order vmcnt queue (the chunk loop's signal prefetch shared
//     that queue: every chunk's MFMA phase absorbed one full memory latency).  Two workgroups per CU: one streams (prologue /
//     epilogue) while the other computes.
//   * rows are 64 bytes (32 channels, no padding: two tiles of C = 128 fit 160 KB) with the 16-byte slots of a row XOR-swizzled by
//     (row >> 2) & 3: the ds_read_b128 of an MFMA operand is conflict-free for every tap offset (lanes {i, 12+i, 20+i, 24+i} of a
//     read group land on rows whose (row >> 2) differ by 3, 5, 6 = four different slots).
//   * mode 0 stages x ONCE for the three branches (was: three problems, three stagings, three HBM reads of x);
//     mode 1 keeps ONE accumulator across the three branches (the MFMA adds conv2_j straight onto it), so o_0, o_1 are never
//     written and re-read.
//   * the residual is taken from the LDS tile itself, in the accumulator's own layout (8 bytes = the 4 channels of an accumulator
//     register quad at this lane's position; lrelu is undone exactly for v >= 0 and to 2^-9 for v < 0 - the tensors are bf16 anyway),
//     so the epilogue has no loads at all; the accumulators are transposed IN REGISTERS (4 x 4 across lane quads, DPP) to 4
//     consecutive positions per lane and leave as 8-byte stores, 64 contiguous bytes per channel row: no LDS scratch.
// Weights: the fragments of v2w_pack_bf16 / v2w_split_pack_batch, straight from L2 through a four-slot register ring (as the chunk
// kernel's PAIRS form: the tile's taps in pairs, fragments three k-steps ahead).
#include <type_traits>
#include "v2w_tile.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define V2W_RS_MAXB 3
#define V2W_RS_UNIT 2048     // byte pitch of the packed fragments of one (32-row block, 16-channel k-step, tap)
#define V2W_RS_HMAX 32

struct ResArgs {
    const unsigned short* in[V2W_RS_MAXB];      // bf16 (B, C, L); mode 0: in[0] only
    const float* in_a; const float* in_s;        // mode 0: per-(b, c) affine of the shared input (NULL: none)
    const unsigned char* wps[V2W_RS_MAXB];
    const float* bias[V2W_RS_MAXB];
    unsigned short* out[V2W_RS_MAXB];            // bf16 (B, C, L); mode 1: out[0] only
    int K[V2W_RS_MAXB], dil[V2W_RS_MAXB];
    int nbr, B, C, L;
    int hla, xrows, ntl, ntiles;
    float slope, inv_slope, out_div;
};

__device__ __forceinline__ unsigned int rs_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float rs_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float rs_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ int rs_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T* rs_uni(T* v) { pin_s(v); return v; }

// quad_perm [1,0,3,2] / [2,3,0,1]: the value of lane ^ 1 / lane ^ 2 inside every group of four lanes
__device__ __forceinline__ float rs_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float rs_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
// 4 x 4 transpose across a lane quad: afterwards register r of quad lane q holds what register q of quad lane r held
__device__ __forceinline__ void rs_quad_transpose(float (&x)[4], bool b0, bool b1) {
    const float r0 = rs_xor1(b0 ? x[0] : x[1]);
    const float r1 = rs_xor1(b0 ? x[2] : x[3]);
    x[0] = b0 ? r0 : x[0]; x[1] = b0 ? x[1] : r0;
    x[2] = b0 ? r1 : x[2]; x[3] = b0 ? x[3] : r1;
    const float r2 = rs_xor2(b1 ? x[0] : x[2]);
    const float r3 = rs_xor2(b1 ? x[1] : x[3]);
    x[0] = b1 ? r2 : x[0]; x[2] = b1 ? x[2] : r2;
    x[1] = b1 ? r3 : x[1]; x[3] = b1 ? x[3] : r3;
}

template <int MI, int NI, int WM, int WN, int MODE>
__global__ void __launch_bounds__(64 * WM * WN, 2)
conv_bf16_res_kernel(const ResArgs a) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTH = 64 * WM * WN, MT = 32 * MI * WM, NT = 32 * NI * WN;
    constexpr int NPF = (8 * ((NT + 2 * V2W_RS_HMAX) / 4) + NTH - 1) / NTH;      // staging items of one 32-channel plane per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];

    const int C = rs_uni(a.C), L = rs_uni(a.L), xrows = rs_uni(a.xrows), hla = rs_uni(a.hla), nbr = rs_uni(a.nbr);
    const int nch = C >> 5, psz = xrows * 64;
    const float slope = a.slope, inv_slope = a.inv_slope;
    float* const btab = reinterpret_cast<float*>(smem_r + nch * psz);       // bias[nbr][MT] (mode 1: their sum in row 0)
    float* const atab = btab + V2W_RS_MAXB * MT;                            // a[C], s[C] of this batch item (mode 0)

    const int mtiles = C / MT;
    const int id = blockIdx.x;
    const int grp = id / (8 * mtiles), rem = id % (8 * mtiles);
    const int mt = rem >> 3;
    const int tile = grp * 8 + (rem & 7);
    if (tile >= a.ntiles) return;
    const int b = tile / a.ntl;
    const int n0 = (tile % a.ntl) * NT;
    const int m0 = mt * MT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = rs_uni(tid >> 6);
    const int lr = lane & 31, hk = lane >> 5;
    const int wm0 = (wave / WN) * (32 * MI);
    const int wn0 = (wave % WN) * (32 * NI);
    const int pos0 = n0 - hla;

    // ---- staging: an item = 4 consecutive channels x 4 positions (four 8-byte loads of 4 positions, four 8-byte LDS stores of the 4
    // channels of one position); consecutive lanes take the 8 channel quads of one position quad
    const int nq = xrows >> 2;
    unsigned poff[NPF];
    bool in_img[NPF], in_seq[NPF];
    int srow[NPF], scq[NPF];
#pragma unroll
    for (int s = 0; s < NPF; ++s) {
        const int idx = tid + s * NTH;
        scq[s] = idx & 7;
        const int pq = idx >> 3;
        srow[s] = pq * 4;
        in_img[s] = pq < nq;
        const int pos = pos0 + pq * 4;
        in_seq[s] = in_img[s] && pos >= 0 && pos < L;           // L % 4 == 0 and pos % 4 == 0: a quad is inside or outside as a whole
        poff[s] = (unsigned)(4 * scq[s] * L + (in_seq[s] ? pos : 0)) * 2u;
        asm volatile("" : "+v"(poff[s]));
    }
    auto prefetch = [&](const unsigned short* in, int ch, u32x2 (&pf)[NPF][4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned char* base = reinterpret_cast<const unsigned char*>(in) + (size_t)(b * C + 32 * ch + i) * L * 2;
#pragma unroll
            for (int s = 0; s < NPF; ++s) pf[s][i] = *gptr<const u32x2>(base + poff[s]);
        }
    };
    auto prefetch1 = [&](const unsigned short* in, int ch, int s, u32x2 (&pf)[4]) {       // item s of plane ch alone
#pragma unroll
        for (int i = 0; i < 4; ++i)
            pf[i] = *gptr<const u32x2>(reinterpret_cast<const unsigned char*>(in) + (size_t)(b * C + 32 * ch + i) * L * 2 + poff[s]);
    };
    auto commit = [&](int ch, const u32x2 (&pf)[NPF][4], int only = -1) {
        unsigned char* const plane = smem_r + ch * psz;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            if (only >= 0 && s != only) continue;
            if (!in_img[s]) continue;
            const int cq = scq[s], row = srow[s];
            float av[4], sv[4];
            if constexpr (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { av[i] = atab[32 * ch + 4 * cq + i]; sv[i] = atab[C + 32 * ch + 4 * cq + i]; }
            }
            // slot (cq >> 1) of the row, swizzled by (row >> 2) & 3 - the same for the item's four rows
            unsigned char* dst = plane + row * 64 + ((((cq >> 1) ^ ((row >> 2) & 3)) << 4) | ((cq & 1) << 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float xv = (e & 1) ? rs_hi(pf[s][i][e >> 1]) : rs_lo(pf[s][i][e >> 1]);
                    if constexpr (MODE == 0) xv = fmaf(av[i], xv, sv[i]);
                    v[i] = v2w_lrelu(xv, slope);
                }
                u32x2 w = {rs_pack2(v[0], v[1]), rs_pack2(v[2], v[3])};
                if (!in_seq[s]) w = u32x2{0u, 0u};            // the padding of the ACTIVATED signal is exactly 0
                *reinterpret_cast<u32x2*>(dst + e * 64) = w;
            }
        }
    };
    // mode 1 stages under live accumulators: ONE plane of registers, item s of plane ch + 1 requested as soon as item s of plane ch is
    // committed (two buffers of a plane each spilled 17 registers)
    auto stage_rot = [&](const unsigned short* in) {
        u32x2 pf[NPF][4];
        prefetch(in, 0, pf);
        __builtin_amdgcn_sched_barrier(0);
        for (int ch = 0; ch + 1 < nch; ++ch) {
#pragma unroll
            for (int s = 0; s < NPF; ++s) {
                commit(ch, pf, s);
                prefetch1(in, ch + 1, s, pf[s]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        commit(nch - 1, pf);
    };
    auto stage = [&](const unsigned short* in) {              // every plane of the tile; two planes of loads in flight
        if constexpr (MODE == 1) { stage_rot(in); return; }
        u32x2 pf0[NPF][4], pf1[NPF][4];
        prefetch(in, 0, pf0);
        prefetch(in, 1, pf1);
        __builtin_amdgcn_sched_barrier(0);
        int ch = 0;
        for (; ch + 2 < nch; ch += 2) {
            commit(ch, pf0);
            prefetch(in, ch + 2, pf0);
            __builtin_amdgcn_sched_barrier(0);
            commit(ch + 1, pf1);
            prefetch(in, ch + 3, pf1);
            __builtin_amdgcn_sched_barrier(0);
        }
        commit(ch, pf0);
        commit(ch + 1, pf1);
    };

    // ---- tables (bias rows of this M-tile; the affine of the shared input)
    for (int i = tid; i < V2W_RS_MAXB * MT; i += NTH) {
        const int j = i / MT, c = i - j * MT;
        float v = (j < nbr && a.bias[j]) ? a.bias[j][m0 + c] : 0.f;
        if constexpr (MODE == 1) {                            // one accumulator for all branches: row 0 holds the sum of the biases
            if (j == 0) for (int jj = 1; jj < nbr; ++jj) v += a.bias[jj] ? a.bias[jj][m0 + c] : 0.f;
        }
        btab[i] = v;
    }
    if constexpr (MODE == 0) {
        for (int c = tid; c < C; c += NTH) {
            atab[c] = a.in_a ? a.in_a[b * C + c] : 1.f;
            atab[C + c] = a.in_a ? a.in_s[b * C + c] : 0.f;
        }
        __syncthreads();
    }

    acc_t acc[MI][NI];
    const unsigned lane16 = (unsigned)lane * 16u;
    u32x4 bb[1][NI];
    auto mfma = [&](acc_t c, u32x4 av, u32x4 bv) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, av), __builtin_bit_cast(b8, bv), c, 0, 0, 0);
    };
    // one k-step: MI x NI MFMAs; `nxt` = this lane's 16 bytes, in column block 0 (the blocks are 2 KiB apart), of the k-step that operand
    // set `bs` serves next
    auto kstep = [&](auto bs_c, const u32x4 (&av)[MI], unsigned nxt) {
        constexpr int bs = decltype(bs_c)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = mfma(acc[i][j], av[i], bb[bs][j]);
            bb[bs][j] = *reinterpret_cast<const u32x4*>(smem_r + nxt + j * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // this lane's 16 bytes of k-step 0 in the row of (its column of block 0, tap offset `row`): slot hk, swizzled by the row
    auto baddr = [&](int ch, int row) {                 // (a byte offset into the LDS image)
        return (unsigned)(ch * psz + row * 64 + ((hk ^ ((row >> 2) & 3)) << 4));
    };
    // accumulator registers 4g .. 4g+3 of block (i, j) <-> channels 8g + 4hk + {0..3} of plane (m0 + wm0) / 32 + i at this lane's position:
    // 8 contiguous bytes of the tile (the ACTIVATED value; lrelu undone)
    auto residual4 = [&](int i, int j, int g, float (&r)[4]) {
        int row = hla + wn0 + lr;                        // (+ 32 j: the swizzle term (row >> 2) & 3 does not change)
        asm volatile("" : "+v"(row));                    // recomputed at every use: hoisted out of the branch loop these addresses spill
        const unsigned q = (unsigned)(((m0 + wm0) / 32 + i) * psz + (row + 32 * j) * 64 + ((g ^ ((row >> 2) & 3)) << 4) + 8 * hk);
        const u32x2 w = *reinterpret_cast<const u32x2*>(smem_r + q);
        const float v[4] = {rs_lo(w[0]), rs_hi(w[0]), rs_lo(w[1]), rs_hi(w[1])};
#pragma unroll
        for (int x = 0; x < 4; ++x) r[x] = v[x] > 0.f ? v[x] : v[x] * inv_slope;
    };
    auto init_acc = [&](int brow) {                        // accumulators start at the bias
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(btab + brow * MT + wm0 + 32 * i + 8 * g + 4 * hk);
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j][4 * g + x] = bv[x];
            }
    };
    auto add_residual = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float r[4];
                    residual4(i, j, g, r);
#pragma unroll
                    for (int x = 0; x < 4; ++x) acc[i][j][4 * g + x] += r[x];
                    if (g == 3) __builtin_amdgcn_sched_barrier(0);      // (block by block: all 32 reads at once cost 64 registers)
                }
    };

    // ---- the MFMA loop of one conv over the resident tile: nch * K taps, walked in pairs (ring slots 0 / 1 even taps, 2 / 3 odd taps)
    auto conv = [&](const unsigned char* wps, int K, int dil) {
        const int hl = dil * (K - 1) / 2;
        const int nst = 2 * nch * K;
        const unsigned char* ap[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) ap[i] = wps + (size_t)((m0 + wm0) / 32 + i) * nst * V2W_RS_UNIT;
        u32x4 ar[4][MI];
        auto load_frag = [&](u32x4 (&av)[MI], int ch, int s, int t) {     // k-step s of (chunk ch, tap t); clamped past the end
            unsigned l16 = lane16;
            asm volatile("" : "+v"(l16));
            const int chc = ch < nch ? ch : nch - 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                av[i] = *gptr<const u32x4>(ap[i] + (size_t)((2 * chc + s) * K + t) * V2W_RS_UNIT + l16);
        };
        load_frag(ar[0], 0, 0, 0);
        load_frag(ar[1], 0, 1, 0);
        {
            const int c1 = K > 1 ? 0 : 1, t1 = K > 1 ? 1 : 0;     // tap 1 of the tile
            load_frag(ar[2], c1, 0, t1);
            load_frag(ar[3], c1, 1, t1);
        }
        __builtin_amdgcn_sched_barrier(0);
        int ch = 0, t = 0;                                   // the running tap
        int qc = 0, qt = 2;                                  // the tap two ahead of it (whose fragments the running tap requests)
        while (qt >= K) { qt -= K; ++qc; }
        const int r0 = hla - hl + wn0 + lr;                  // LDS row of (column lr of block 0, tap 0)
        unsigned xt = baddr(0, r0);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            bb[0][j] = *reinterpret_cast<const u32x4*>(smem_r + xt + j * 2048);
        }
        typedef std::integral_constant<int, 0> set0;
        typedef std::integral_constant<int, 0> set1;
        auto tap = [&](auto par_c) {
            constexpr int S0 = 2 * decltype(par_c)::value;
            // the next tap: same chunk one dilation step on, or tap 0 of the next chunk (past the end: this tap again - unused)
            int nch_ = ch, nt_ = t + 1;
            if (nt_ >= K) { nt_ = 0; ++nch_; }
            if (nch_ >= nch) { nch_ = ch; nt_ = t; }
            const unsigned xn = baddr(nch_, r0 + nt_ * dil);
            kstep(set0{}, ar[S0], xt ^ 32u);                 // k-step 1 of the tap: slot ^ 2
            load_frag(ar[S0], qc, 0, qt);
            __builtin_amdgcn_sched_barrier(0);
            kstep(set1{}, ar[S0 + 1], xn);
            load_frag(ar[S0 + 1], qc, 1, qt);
            __builtin_amdgcn_sched_barrier(0);
            if (++qt >= K) { qt = 0; ++qc; }
            ch = nch_; t = nt_; xt = xn;
        };
        const int TT = nch * K;
        int g = 0;
        for (; g + 1 < TT; g += 2) { tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); }
        if (g < TT) tap(std::integral_constant<int, 0>{});
    };

    // ---- epilogue: accumulators (+ residual) -> 4 x 4 register transposes -> bf16 x 4 positions, 8-byte stores
    const bool q0 = lane & 1, q1 = lane & 2;
    const int qm = lr >> 2, qp = lr & 3;
    auto store_tile = [&](unsigned short* out, bool with_res, float div) {
        const float dinv = div != 0.f ? 1.f / div : 1.f;
        unsigned vo = (unsigned)((4 * hk + qp) * L + 4 * qm) * 2u;             // lane part of the address: channel 4hk + qp, positions 4 qm ..
        asm volatile("" : "+v"(vo));
        unsigned char* const obase = reinterpret_cast<unsigned char*>(out) + (size_t)b * C * L * 2;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int q = n0 + wn0 + 32 * j;                             // (uniform) first position of the block
                const bool ok = q + 4 * qm < L;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = acc[i][j][4 * g + x];
                    if (with_res) {
                        float r[4];
                        residual4(i, j, g, r);
#pragma unroll
                        for (int x = 0; x < 4; ++x) v[x] += r[x];
                    }
                    if (div != 0.f) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], div, dinv);
                    }
                    rs_quad_transpose(v, q0, q1);
                    const u32x2 w = {rs_pack2(v[0], v[1]), rs_pack2(v[2], v[3])};
                    // uniform base of the batch item (64-bit, scalar) + a 32-bit offset: uniform part + the lane's (C * L * 2 < 2^31: checked by the launcher)
                    const unsigned uo = (unsigned)((m0 + wm0 + 32 * i + 8 * g) * L + q) * 2u;
                    if (ok) *gptr<u32x2>(obase + (uo + vo)) = w;
                }
                __builtin_amdgcn_sched_barrier(0);          // (block by block: hoisting all 32 residual reads to the top spills accumulators)
            }
    };

    V2W_STAMP(0);
    if constexpr (MODE == 0) {
        stage(a.in[0]);
        V2W_STAMP(1);
        __syncthreads();
        V2W_STAMP(2);
        for (int j = 0; j < nbr; ++j) {
            init_acc(j);
            V2W_STAMP(3 + 3 * j);
            conv(rs_uni(a.wps[j]), rs_uni(a.K[j]), rs_uni(a.dil[j]));
            V2W_STAMP(4 + 3 * j);
            store_tile(rs_uni(a.out[j]), true, 0.f);
            V2W_STAMP(5 + 3 * j);
        }
    } else {
        for (int j = 0; j < nbr; ++j) {
            if (j > 0) __syncthreads();                      // every wave is done with the previous branch's tile
            V2W_STAMP(1 + 5 * j);
            stage(rs_uni(a.in[j]));
            V2W_STAMP(2 + 5 * j);
            __syncthreads();
            V2W_STAMP(3 + 5 * j);
            if (j == 0) init_acc(0);
            add_residual();
            V2W_STAMP(4 + 5 * j);
            conv(rs_uni(a.wps[j]), rs_uni(a.K[j]), rs_uni(a.dil[j]));
            V2W_STAMP(5 + 5 * j);
        }
        store_tile(rs_uni(a.out[0]), false, a.out_div);
    }
    V2W_STAMP(20);
}

template <int MI, int NI, int WM, int WN>
int launch_res(const v2w_branch_convs_args* q, hipStream_t stream) {
    constexpr int NTH = 64 * WM * WN, MT = 32 * MI * WM, NT = 32 * NI * WN;
    ResArgs p{};
    int hmax = 0;
    for (int j = 0; j < q->nbr; ++j) {
        p.in[j] = static_cast<const unsigned short*>(q->in[j]);
        p.wps[j] = static_cast<const unsigned char*>(q->wps[j]);
        p.bias[j] = q->bias[j];
        p.out[j] = static_cast<unsigned short*>(q->out[j]);
        p.K[j] = q->k[j]; p.dil[j] = q->dil[j];
        const int h = q->dil[j] * (q->k[j] - 1) / 2;
        if (h > hmax) hmax = h;
    }
    if (hmax > V2W_RS_HMAX) return V2W_E_SHAPE;
    p.in_a = q->in_a; p.in_s = q->in_s;
    p.nbr = q->nbr; p.B = q->B; p.C = q->C; p.L = q->L;
    p.hla = (hmax + 3) & ~3;
    p.xrows = (p.hla + NT + hmax + 3) & ~3;
    p.ntl = (q->L + NT - 1) / NT;
    p.ntiles = q->B * p.ntl;
    p.slope = q->slope; p.inv_slope = 1.f / q->slope; p.out_div = q->out_div;
    if (q->C % MT != 0) return V2W_E_SHAPE;
    const size_t lds = (size_t)(q->C / 32) * p.xrows * 64 + (size_t)(V2W_RS_MAXB * MT + 2 * q->C) * sizeof(float);
    if (2 * lds > 160 * 1024) return V2W_E_SHAPE;            // two workgroups per CU, or the structure does not pay
    const int grid = ((p.ntiles + 7) / 8) * 8 * (q->C / MT);
    auto kern = q->mode == 0 ? conv_bf16_res_kernel<MI, NI, WM, WN, 0> : conv_bf16_res_kernel<MI, NI, WM, WN, 1>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(grid), dim3(NTH), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_res)
#endif

#define V2W_RS_C64_CFG 0

extern "C" int v2w_branch_convs_bf16_fwd(const v2w_branch_convs_args* a, void* stream) {
    if (!a) return V2W_E_ARG;
    if (a->nbr < 1 || a->nbr > V2W_RS_MAXB || (a->mode != 0 && a->mode != 1)) return V2W_E_ARG;
    if (a->B <= 0 || a->C <= 0 || a->L <= 0 || !(a->slope > 0.f)) return V2W_E_ARG;
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    for (int j = 0; j < a->nbr; ++j) {
        if (!a->wps[j] || a->k[j] < 1 || a->dil[j] < 1) return V2W_E_ARG;
        if ((a->k[j] & 1) == 0) return V2W_E_SHAPE;
        if (a->mode == 0 ? (!a->out[j] || !al16(a->out[j])) : (!a->in[j] || !al16(a->in[j]))) return a->mode == 0 && !a->out[j] ? V2W_E_ARG : V2W_E_SHAPE;
    }
    if (!a->in[0] || !a->out[0]) return V2W_E_ARG;
    if (!al16(a->in[0]) || !al16(a->out[0])) return V2W_E_SHAPE;
    if (a->L % 4 != 0 || a->C % 64 != 0) return V2W_E_SHAPE;            // vector staging only; an even number of 32-channel planes
    if ((long long)a->C * a->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;  // 32-bit lane offsets inside one batch item
    hipStream_t st = (hipStream_t)stream;
    if (a->C % 128 == 0) return launch_res<2, 4, 2, 2>(a, st);          // 128 x 256
    return launch_res<1, 4, 2, 2>(a, st);                               // 64 x 256
}
