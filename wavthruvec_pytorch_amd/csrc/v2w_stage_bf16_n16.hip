// The 16-channel ResBlock2 stage of the generator on bf16 tensors (reference: vec2wav/models.py:135-141 with h.resblock_kernel_sizes
// (3, 7, 11) and dilations (1, 3) - the reference's only configuration; anything else runs on the kernels of v2w_stage_bf16*.hip):
//   out = ( sum_j [ t1_j + conv_{k_j, 3}(lrelu(t1_j)) + b2_j ] ) / 3,   t1_j = x + conv_{k_j, 1}(lrelu(x)) + b1_j,   x = a * in + s.
//
// Why its own kernel.  At BASELINE configs[2] this stage moves 671 MB (84 us of HBM time) and its 0.45 TFLOP would take 0.2 ms even on
// half-empty 32-row MFMAs, yet stage_bf16_kernel<16> took 790 us and the resident-tile template (v2w_stage_bf16_wide.hip, CH = 16) 912 us.
// Measured on the latter: a tile's six conv loops took 3.3 x their MFMA issue time - with 16 channels a tap is ONE k-step of four MFMAs,
// and its 2 KiB weight fragment arrives from L2 (~1 k cycles) on a two-tap ring: the loop waits for weights, tile after tile, for the same
// 42 fragments.  Here:
//   * v_mfma_f32_16x16x32_bf16: M = the 16 output channels (no padding rows), N = 16 positions, K = 32 = TWO taps x 16 input channels;
//   * the weights of all six convs - 24 tap pairs x one 16-byte operand per lane - live in 96 REGISTERS of every wave for the whole
//     (persistent) kernel: the conv loops issue nothing but ds_read_b128 + MFMA, every offset an immediate;
//   * 32-byte LDS rows (one position, 16 channels): lane (j, kg) reads the 16 bytes of half kg & 1 of row (column j + tap (2 p + (kg >> 1))):
//     the 16-lane groups of a ds_read_b128 fall on 16 different 16-byte slots of the 256-byte bank row at every offset - no swizzle;
//   * the accumulator of a 16 x 16 block holds, per lane, 4 consecutive CHANNELS of one position = 8 contiguous bytes of a tile row: the
//     residual read and the t1 write of the epilogue are single 8-byte LDS accesses.
// One operand read per MFMA is exactly the LDS's 256 B / clk / CU: the conv loops are LDS-bound (~0.13 ms at configs[2]), not weight-bound.
#include <type_traits>
#include <utility>
#include "v2w_tile.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct N16Args {
    const unsigned short* in; const float* in_a; const float* in_s;
    const unsigned char* w1[3]; const float* bias1[3];
    const unsigned char* w2[3]; const float* bias2[3];
    unsigned short* out;
    int B, L, nto, ntl, ntiles;
    float slope, inv_slope, out_div;
    // fused tail (POST instantiation; models.py:143-145): y = tanh(conv_post(leaky_relu(out, post_slope))) with 7 taps, written as fp32 (B, 1, L);
    // `out` is not written.  A tile then advances by nadv = nto - 2 hout positions and starts hout = 4 positions early (3 feed the taps, 4
    // keep the output quads 16-byte aligned).
    const float* post_w; const float* post_b; float* post_out;
    float post_slope;
    int nadv, hout;
};

__device__ __forceinline__ unsigned int n16_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float n16_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float n16_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

template <int... I, class F> __device__ __forceinline__ void n16_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// Workgroup barrier for LDS hand-overs only: __syncthreads() also fences global memory - a wave that has just issued its tile's output
// stores would wait vmcnt(0) (the stores' acknowledgement, thousands of cycles under load) before it may even arrive at the barrier.
__device__ __forceinline__ void n16_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int N16_H1 = 5, N16_H2 = 15;           // halos of the widest branch: 11 taps at dilation 1 / 3
constexpr int N16_NB = 8;                        // 16-column blocks per wave (128 columns)

// WN waves per workgroup: a window of W = 128 WN columns (positions n0 - 15 .. n0 - 15 + W), of which the middle nto = (W - 30) & ~3 are
// valid outputs; x tile rows = positions n0 - 20 .. (W + 12 rows), t1 tile rows = window columns.
template <int WN, bool POST = false>
__global__ void __launch_bounds__(64 * WN, 2)
n16_stage_kernel(const N16Args a) {
    constexpr int NTH = 64 * WN, W = 128 * WN, XR = W + 12, RB = 32, NB = N16_NB;
    constexpr int SRS = W + 12;                                                 // scratch row stride (floats): 4 SRS = 16 mod 32 banks
    constexpr unsigned XB = 0, TB = XR * RB;                                    // LDS byte offsets of the x and t1 tiles
    constexpr unsigned RT = (XR + W + 16) * RB;                                 // ... and of the r tile (x itself, rows as the x tile)
    constexpr int NIT = 4 * (XR / 4), NPF = (NIT + NTH - 1) / NTH;              // staging items (4 channels x 4 positions), per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_n[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kg = lane >> 4;
    const int L = __builtin_amdgcn_readfirstlane(a.L), nto = __builtin_amdgcn_readfirstlane(a.nto);
    const float slope = a.slope;

    // ---- every weight of the stage into registers: pair p of a conv = taps 2p, 2p + 1 (the second one zero past the last tap).  The packed
    // fragment of tap t (v2w_pack_bf16, 2 KiB) holds, for lane' = row + 32 h, the 8 input channels 8h .. 8h + 7 of output channel `row`.
    u32x4 wa[2][12];
    {
        const unsigned lo16 = (unsigned)(j + 32 * (kg & 1)) * 16u;
        auto load_set = [&](int s, int jb, int K, int p0) {
            const unsigned char* w = s ? a.w2[jb] : a.w1[jb];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (2 * p >= K) break;
                const int t = 2 * p + (kg >> 1);
                u32x4 v = {0u, 0u, 0u, 0u};
                if (t < K) v = *reinterpret_cast<const u32x4*>(w + (size_t)t * 2048 + lo16);
                wa[s][p0 + p] = v;
            }
        };
#pragma unroll
        for (int s = 0; s < 2; ++s) { load_set(s, 0, 3, 0); load_set(s, 1, 7, 2); load_set(s, 2, 11, 6); }
    }
    // biases of this lane's 4 channels (4 kg .. 4 kg + 3)
    float b1[3][4], b2s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s2 = 0.f;
#pragma unroll
        for (int jb = 0; jb < 3; ++jb) {
            b1[jb][r] = a.bias1[jb] ? a.bias1[jb][4 * kg + r] : 0.f;
            s2 += a.bias2[jb] ? a.bias2[jb][4 * kg + r] : 0.f;
        }
        b2s[r] = s2;
    }

    auto mfma = [&](f32x4 c, u32x4 av, u32x4 bv) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, av), __builtin_bit_cast(b8, bv), c, 0, 0, 0);
    };
    const int col0 = 128 * wave + j;                                            // this lane's column in block 0 of its wave
    // one conv over a resident tile: K taps at dilation DIL, pairs P0 .. of weight set S; r0 = tile row of (column col0, tap 0)
    // The loop is ONE sequence of (pair, block) steps with a ring of RING operands in flight, written in inline assembly: left to itself hipcc
    // (at the register limit) issued one ds_read_b128, waited lgkmcnt(0), issued its MFMA - an LDS round trip per 16-cycle MFMA - and it
    // sinks plain C++ loads back to their uses.  `asm volatile` statements keep their order; the s_waitcnt names the operand it guards so
    // that the MFMA cannot move above it.  LDS reads return in order: before step n at most min(RING - 1, N - 1 - n) younger reads may be
    // outstanding.  (No scalar loads inside: sched_barrier on both sides.)
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_n);
    auto conv = [&](auto k_c, auto d_c, auto s_c, auto p_c, f32x4 (&acc)[NB], unsigned base, int r0) {
        constexpr int K = decltype(k_c)::value, DIL = decltype(d_c)::value, S = decltype(s_c)::value, P0 = decltype(p_c)::value;
        constexpr int NP = (K + 1) / 2, N = NP * NB, RING = 8;
        unsigned ab = lds0 + base + (unsigned)(r0 * RB + (kg & 1) * 16);
        asm volatile("" : "+v"(ab));
        const unsigned ab2 = ab + (unsigned)((kg >> 1) * DIL * RB);             // lanes of the pair's second tap: one dilation step on
        u32x4 ring[RING];
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (the zero tap past the end reads the last real tap's rows: finite values under zero weights)
        n16_for(std::make_integer_sequence<int, RING>{}, [&ring, &ab, &ab2](auto n_c) {
            constexpr int n = decltype(n_c)::value, p = n / NB, cb = n % NB;
            if constexpr (2 * p + 1 >= K) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n]) : "v"(ab), "n"((2 * p * DIL + 16 * cb) * RB));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n]) : "v"(ab2), "n"((2 * p * DIL + 16 * cb) * RB));
        });
        n16_for(std::make_integer_sequence<int, N>{}, [&ring, &ab, &ab2, &acc, &wa, &mfma](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            constexpr int left = (N - 1 - n) < (RING - 1) ? (N - 1 - n) : (RING - 1);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[n % RING]) : "n"(left));
            acc[n % NB] = mfma(acc[n % NB], wa[S][P0 + n / NB], ring[n % RING]);
            if constexpr (n + RING < N) {
                constexpr int m = n + RING, p = m / NB, cb = m % NB;
                if constexpr (2 * p + 1 >= K) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(ab), "n"((2 * p * DIL + 16 * cb) * RB));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(ab2), "n"((2 * p * DIL + 16 * cb) * RB));
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- staging, in two halves: issue = the global loads of a tile's x (an item = 4 channels x 4 positions: four 8-byte loads) and of its
    // affine, commit = activation, bf16 rows into LDS.  The loads of tile i + 1 are issued before the stores of tile i and land under them.
    u32x2 pf[NPF][4];
    float av[4], sv[4];
    const int cq = tid & 3;                                                     // (NTH % 4 == 0: a thread keeps its channel quad)
    auto issue_x = [&](int tile) {
        const int b = tile / a.ntl, n0 = (tile - b * a.ntl) * a.nadv - a.hout;
        const int pos0 = n0 - N16_H1 - N16_H2;
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * 16 * L * 2;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH, pq = idx >> 2;
            const int pos = pos0 + 4 * pq;
            const bool ok = idx < NIT && pos >= 0 && pos < L;
            unsigned vo = (unsigned)(4 * cq * L + (ok ? pos : 0)) * 2u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[s][i] = *gptr<const u32x2>(inb + (size_t)i * L * 2 + vo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = a.in_a ? a.in_a[b * 16 + 4 * cq + i] : 1.f;
            sv[i] = a.in_a ? a.in_s[b * 16 + 4 * cq + i] : 0.f;
        }
    };
    // rows of the x tile: lrelu(x) (the conv operand); rows of the r tile: x itself (the residual), both bf16, 0 outside the sequence
    auto commit_x = [&](int pos0) {
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH, pq = idx >> 2;
            if (idx >= NIT) continue;
            const int pos = pos0 + 4 * pq;
            const bool ok = pos >= 0 && pos < L;                                // L % 4 == 0: a position quad is inside or outside as a whole
            unsigned char* dst = smem_n + XB + (4 * pq) * RB + cq * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (e & 1) ? n16_hi(pf[s][i][e >> 1]) : n16_lo(pf[s][i][e >> 1]);
                    y[i] = fmaf(av[i], xv, sv[i]);
                    v[i] = fmaxf(y[i], y[i] * slope);
                }
                u32x2 w = {n16_pack2(v[0], v[1]), n16_pack2(v[2], v[3])};
                u32x2 r = {n16_pack2(y[0], y[1]), n16_pack2(y[2], y[3])};
                if (!ok) { w = u32x2{0u, 0u}; r = w; }                          // the padding of the ACTIVATED signal is exactly 0
                *reinterpret_cast<u32x2*>(dst + e * RB) = w;
                *reinterpret_cast<u32x2*>(dst + e * RB + RT) = r;
            }
        }
    };

    // the tail's weights, transposed to [channel][8] behind the tiles (never overlaid): a channel's 7 taps are two broadcast float4 reads
    float* const wl = reinterpret_cast<float*>(smem_n + (2 * XR + W + 16) * RB);       // (behind the x, t1 and r tiles)
    if constexpr (POST) {
        if (tid < 128) wl[tid] = (tid & 7) < 7 ? a.post_w[(tid & 7) * 16 + (tid >> 3)] : 0.f;
    }
    issue_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int b = tile / a.ntl, n0 = (tile - b * a.ntl) * a.nadv - a.hout;
        // a tile whose window and halo lie inside the sequence needs no per-position checks in the epilogues
        const bool edge = n0 - N16_H2 < 0 || n0 - N16_H2 + W > L;
        n16_lds_barrier();                                                        // the previous tile's stores have read the scratch
        V2W_STAMP(0);
        commit_x(n0 - N16_H1 - N16_H2);                                         // (position of x row 0: a multiple of 4)
        V2W_STAMP(1);
        n16_lds_barrier();
        V2W_STAMP(2);

        f32x4 oacc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) oacc[cb] = f32x4{b2s[0], b2s[1], b2s[2], b2s[3]};

        auto branch = [&](auto jb_c, auto k_c, auto p_c) {
            constexpr int JB = decltype(jb_c)::value, K = decltype(k_c)::value;
            constexpr int h1 = (K - 1) / 2, h2 = 3 * (K - 1) / 2;
            f32x4 acc1[NB];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) acc1[cb] = f32x4{b1[JB][0], b1[JB][1], b1[JB][2], b1[JB][3]};
            // conv1_j: window column col <-> x row col + 5
            conv(k_c, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, p_c, acc1, XB, col0 + N16_H1 - h1);
            V2W_STAMP(3 + 5 * JB);
            n16_lds_barrier();                                                    // conv2 of the previous branch has read the t1 tile
            V2W_STAMP(4 + 5 * JB);
            // t1 = acc + x (the r tile); the running output takes t1 in fp32, the t1 tile lrelu(t1) as bf16.  Registers 0 .. 3 of block cb
            // <-> channels 4 kg .. 4 kg + 3 at column 16 cb + j of this wave
            u32x2 rw[NB];                                                       // the residual rows of all blocks first: one LDS round trip, not eight
            int colv = col0;
            asm volatile("" : "+v"(colv));
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) rw[cb] = *reinterpret_cast<const u32x2*>(smem_n + RT + (colv + 16 * cb + N16_H1) * RB + 8 * kg);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int col = colv + 16 * cb;
                const u32x2 w = rw[cb];
                const f32x4 xr = {n16_lo(w[0]), n16_hi(w[0]), n16_lo(w[1]), n16_hi(w[1])};
                f32x4 t1v = acc1[cb] + xr;                                      // (vector forms: v_pk_add_f32 / v_pk_mul_f32)
                if (edge) {                                                     // conv2 zero-pads t1 outside the sequence
                    const int pos = n0 - N16_H2 + col;
                    if (pos < 0 || pos >= L) t1v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                oacc[cb] += t1v;
                const f32x4 ts = t1v * slope;
#pragma unroll
                for (int r = 0; r < 4; ++r) t1v[r] = fmaxf(t1v[r], ts[r]);
                *reinterpret_cast<u32x2*>(smem_n + TB + col * RB + 8 * kg) = u32x2{n16_pack2(t1v[0], t1v[1]), n16_pack2(t1v[2], t1v[3])};
            }
            V2W_STAMP(5 + 5 * JB);
            n16_lds_barrier();
            V2W_STAMP(6 + 5 * JB);
            // conv2_j on the same window (taps that reach past the t1 tile read the x tile / the slack behind it: columns that are never stored)
            conv(k_c, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, p_c, oacc, TB, col0 - h2);
            V2W_STAMP(7 + 5 * JB);
        };
        branch(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        branch(std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 2>{});
        branch(std::integral_constant<int, 2>{}, std::integral_constant<int, 11>{}, std::integral_constant<int, 6>{});

        // ---- the nto valid columns (window columns 15 .. 15 + nto) through an fp32 scratch [16][SRS] over the dead tiles: scratch column
        // = window column + 1 (output quads 16-byte aligned), then 8-byte bf16 stores along positions
        n16_lds_barrier();
        V2W_STAMP(18);
        {
            float* const scr = reinterpret_cast<float*>(smem_n);
            if constexpr (POST) {
                // the tail's operand z = leaky_relu(out / nk, post_slope), exactly 0 outside the sequence (conv_post zero-pads): fp32, never rounded
                const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f, ps = a.post_slope;
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    int col = col0;
                    asm volatile("" : "+v"(col));
                    col += 16 * cb;
                    const int pos = n0 - N16_H2 + col;
                    const bool in_seq = pos >= 0 && pos < L;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = oacc[cb][r];
                        if (a.out_div != 0.f) v = v2w_div_by(v, a.out_div, dinv);
                        scr[(4 * kg + r) * SRS + col + 1] = in_seq ? fmaxf(v, v * ps) : 0.f;
                    }
                }
            } else {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                int col = col0;
                asm volatile("" : "+v"(col));
                col += 16 * cb;
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(4 * kg + r) * SRS + col + 1] = oacc[cb][r];
            }
            }
            n16_lds_barrier();
            V2W_STAMP(19);
            // the next tile's x: in flight under this tile's stores (unconditional - past the end the last tile again, never committed: under
            // a condition the old values would stay live through the whole iteration as the other input of the join)
            issue_x(min(tile + (int)gridDim.x, a.ntiles - 1));
            if constexpr (POST) {
                // y[p] = tanh(b + sum_c sum_t w[t][c] z[c][p + t - 3]) for the nadv positions p = n0 + hout + m: window column of z[c][p + t - 3] is
                // 15 + hout + m + t - 3, scratch column one more = 17 + m + t (hout = 4).  A thread takes 4 consecutive outputs (m = 4 q ..) of 8
                // of the 16 channels: per channel the three aligned float4s from scratch column 16 + 4 q hold the 10 values it needs; the two
                // channel halves (threads tid and tid + NTH / 2) meet through LDS.
                constexpr int HALF = NTH / 2;
                const int q = tid % HALF, hf = tid / HALF;
                float* const part = scr + 16 * SRS;                       // [HALF] float4 behind the scratch rows
                const bool act = 4 * q < a.nadv;
                f32x4 y = {0.f, 0.f, 0.f, 0.f};
                if (act) {
#pragma unroll 1
                    for (int cc = 0; cc < 8; ++cc) {       // (rolled: unrolled, hipcc hoists the 24 float4 reads - 96 registers beside the 96 of weights)
                        const int c = 8 * hf + cc;
                        const float* zr = scr + c * SRS + 16 + 4 * q;
                        float z[12];
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const f32x4 zz = *reinterpret_cast<const f32x4*>(zr + 4 * u);
                            z[4 * u] = zz[0]; z[4 * u + 1] = zz[1]; z[4 * u + 2] = zz[2]; z[4 * u + 3] = zz[3];
                        }
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + 8 * c), w1 = *reinterpret_cast<const f32x4*>(wl + 8 * c + 4);
                        const float wv[7] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2]};
#pragma unroll
                        for (int t = 0; t < 7; ++t) {
#pragma unroll
                            for (int x = 0; x < 4; ++x) y[x] = fmaf(wv[t], z[1 + x + t], y[x]);
                        }
                    }
                    if (hf == 1) *reinterpret_cast<f32x4*>(part + 4 * q) = y;
                }
                n16_lds_barrier();
                if (act && hf == 0) {
                    const int p0 = n0 + a.hout + 4 * q;
                    if (p0 < L) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(part + 4 * q);
                        const float pb = a.post_b ? a.post_b[0] : 0.f;
#pragma unroll
                        for (int x = 0; x < 4; ++x) y[x] = tanhf((y[x] + o[x]) + pb);
                        *gptr<f32x4>(a.post_out + (size_t)b * L + p0) = y;
                    }
                }
            } else {
            const int nq = nto >> 2;
            const unsigned magic = (unsigned)(((1ull << 32) + nq - 1) / nq);
            const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f;
            unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + (size_t)b * 16 * L * 2;
            for (int idx = tid; idx < 16 * nq; idx += NTH) {
                const int row = (int)__umulhi((unsigned)idx, magic), q = idx - row * nq;
                const int pos = n0 + 4 * q;
                if (pos >= L) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * SRS + 16 + 4 * q);
                if (a.out_div != 0.f) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], a.out_div, dinv);
                }
                *gptr<u32x2>(obase + (unsigned)(row * L + pos) * 2u) = u32x2{n16_pack2(v[0], v[1]), n16_pack2(v[2], v[3])};
            }
            }     // (!POST)
        }
        V2W_STAMP(20);
    }
}

template <int WN>
int launch_n16(const v2w_stage_split_args* q, hipStream_t stream) {
    constexpr int NTH = 64 * WN, W = 128 * WN, XR = W + 12;
    N16Args p{};
    p.in = reinterpret_cast<const unsigned short*>(q->in); p.in_a = q->in_a; p.in_s = q->in_s;
    p.out = reinterpret_cast<unsigned short*>(q->out);
    for (int j = 0; j < 3; ++j) {
        p.w1[j] = static_cast<const unsigned char*>(q->wps1[j]); p.bias1[j] = q->bias1[j];
        p.w2[j] = static_cast<const unsigned char*>(q->wps2[j]); p.bias2[j] = q->bias2[j];
    }
    p.B = q->B; p.L = q->L; p.slope = q->slope; p.inv_slope = 1.f / q->slope; p.out_div = q->out_div;
    p.nto = (W - 2 * N16_H2) & ~3;
    p.nadv = p.nto; p.hout = 0;
    const bool post = q->post_out != nullptr;
    if (post) {                     // the generator's tail behind the stage: 7 taps, 16 -> 1 channels, fp32 output
        p.post_w = q->post_w; p.post_b = q->post_b; p.post_out = q->post_out; p.post_slope = q->post_slope;
        p.hout = 4; p.nadv = (p.nto - 2 * p.hout) & ~3;
    }
    p.ntl = (q->L + p.nadv - 1) / p.nadv;
    if ((long long)q->B * p.ntl > 0x7fffffffll) return V2W_E_SHAPE;
    p.ntiles = q->B * p.ntl;
    // x, t1 (+ 16 rows of slack behind it: conv2's taps past its end) and r tiles; the store scratch [16][W + 12] floats overlays them
    const size_t tiles = (size_t)(XR + W + 16 + XR) * 32, scratch = (size_t)16 * (W + 12) * sizeof(float) + (post ? (size_t)(NTH / 2) * 16 : 0);
    if (scratch > tiles) return V2W_E_SHAPE;                                    // (the scratch overlays the tiles)
    const size_t lds = tiles + (post ? 512 : 0);
    if (v2w_dry(stream)) return 0;
    const int ncu = v2w_num_cus();
    // persistent: the registers hold the stage's weights, so a workgroup walks tiles; 8 waves per CU (2 per SIMD: ~230 registers each)
    const int slots = ncu * (8 / WN);
    if (post) V2W_LAUNCH((n16_stage_kernel<WN, true>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(NTH), lds, stream, p);
    else V2W_LAUNCH((n16_stage_kernel<WN, false>), dim3(p.ntiles < slots ? p.ntiles : slots), dim3(NTH), lds, stream, p);
    return v2w_launch_status();
}


// ---- ResBlock1 on 16 channels (models.py:37-44 with h.resblock == '1'; v2w_stage_split_args::rb1): ONE (dilated conv, conv) pair of one branch,
//   out = ( x + conv_{K,1}(lrelu(u)) + b2 [ + add0 + add1 ] ) [ / out_div ],   u = conv_{K,D1}(lrelu(x)) + b1,   x = a * in + s,
// with the machinery of the stage kernel above: v_mfma_f32_16x16x32_bf16 on two taps x 16 channels per k-step, the pair's 2 x ceil(K / 2)
// weight operands in registers for the whole persistent kernel, conv loops of nothing but ds_read_b128 + MFMA.  (The resident-tile template's
// run-time form served these pairs at 0.06 of the bf16 MFMA peak: 275 us per three-branch launch at B = 32 x 81 920 positions against the ~100 us
// its 0.5 GB take.)  One launch per branch - K and D1 are template parameters; x tile rows = positions n0 - H2 - H1 - XOFF .., u tile rows = window
// columns (position n0 - H2 + column), valid outputs = columns H2 .. H2 + nto.
struct N16PairArgs {
    const unsigned short* in; const float* in_a; const float* in_s;
    const unsigned char* w1; const float* bias1;
    const unsigned char* w2; const float* bias2;
    unsigned short* out;
    const unsigned short* add0; const unsigned short* add1;
    int B, L, nto, ntl, ntiles;
    float slope, out_div;
};

template <int WN, int K, int D1>
__global__ void __launch_bounds__(64 * WN, 2)
n16_pair_kernel(const N16PairArgs a) {
    constexpr int H1 = D1 * (K - 1) / 2, H2 = (K - 1) / 2, XOFF = (4 - (H1 + H2) % 4) % 4;
    constexpr int NTH = 64 * WN, W = 128 * WN, XR = (W + 2 * H1 + XOFF + 3) & ~3, RB = 32, NB = N16_NB;
    constexpr int SRS = W + 12, SOFF = (4 - H2 % 4) % 4;                        // scratch row stride (floats); scratch column = window column + SOFF
    constexpr unsigned XB = 0, TB = XR * RB, RT = (XR + W + 16) * RB;           // x tile, u tile (+ 16 rows of slack), r tile (x itself)
    constexpr int NIT = 4 * (XR / 4), NPF = (NIT + NTH - 1) / NTH;
    constexpr int NP = (K + 1) / 2;
    static_assert(H2 <= 16 && 16 * SRS * 4 <= (2 * XR + W + 16) * RB, "conv2 may reach 16 rows past the u tile; the store scratch overlays the (dead) tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_n[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kg = lane >> 4;
    const int L = __builtin_amdgcn_readfirstlane(a.L), nto = __builtin_amdgcn_readfirstlane(a.nto);
    const float slope = a.slope;

    u32x4 wa[2][NP];
    {
        const unsigned lo16 = (unsigned)(j + 32 * (kg & 1)) * 16u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char* w = s ? a.w2 : a.w1;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int t = 2 * p + (kg >> 1);
                u32x4 v = {0u, 0u, 0u, 0u};
                if (t < K) v = *reinterpret_cast<const u32x4*>(w + (size_t)t * 2048 + lo16);
                wa[s][p] = v;
            }
        }
    }
    float b1[4], b2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b1[r] = a.bias1 ? a.bias1[4 * kg + r] : 0.f;
        b2[r] = a.bias2 ? a.bias2[4 * kg + r] : 0.f;
    }
    auto mfma = [&](f32x4 c, u32x4 av, u32x4 bv) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, av), __builtin_bit_cast(b8, bv), c, 0, 0, 0);
    };
    const int col0 = 128 * wave + j;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_n);
    // (the conv loop of n16_stage_kernel: a ring of 8 operands in flight, every offset an immediate)
    auto conv = [&](auto d_c, auto s_c, f32x4 (&acc)[NB], unsigned base, int r0) {
        constexpr int DIL = decltype(d_c)::value, S = decltype(s_c)::value;
        constexpr int N = NP * NB, RING = 8;
        unsigned ab = lds0 + base + (unsigned)(r0 * RB + (kg & 1) * 16);
        asm volatile("" : "+v"(ab));
        const unsigned ab2 = ab + (unsigned)((kg >> 1) * DIL * RB);
        u32x4 ring[RING];
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        n16_for(std::make_integer_sequence<int, RING>{}, [&ring, &ab, &ab2](auto n_c) {
            constexpr int n = decltype(n_c)::value, p = n / NB, cb = n % NB;
            if constexpr (2 * p + 1 >= K) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n]) : "v"(ab), "n"((2 * p * DIL + 16 * cb) * RB));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n]) : "v"(ab2), "n"((2 * p * DIL + 16 * cb) * RB));
        });
        n16_for(std::make_integer_sequence<int, N>{}, [&ring, &ab, &ab2, &acc, &wa, &mfma](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            constexpr int left = (N - 1 - n) < (RING - 1) ? (N - 1 - n) : (RING - 1);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[n % RING]) : "n"(left));
            acc[n % NB] = mfma(acc[n % NB], wa[S][n / NB], ring[n % RING]);
            if constexpr (n + RING < N) {
                constexpr int m = n + RING, p = m / NB, cb = m % NB;
                if constexpr (2 * p + 1 >= K) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(ab), "n"((2 * p * DIL + 16 * cb) * RB));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(ab2), "n"((2 * p * DIL + 16 * cb) * RB));
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    u32x2 pf[NPF][4];
    float av[4], sv[4];
    const int cq = tid & 3;
    auto issue_x = [&](int tile) {
        const int b = tile / a.ntl, n0 = (tile - b * a.ntl) * nto;
        const int pos0 = n0 - H1 - H2 - XOFF;
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * 16 * L * 2;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH, pq = idx >> 2;
            const int pos = pos0 + 4 * pq;
            const bool ok = idx < NIT && pos >= 0 && pos < L;
            unsigned vo = (unsigned)(4 * cq * L + (ok ? pos : 0)) * 2u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[s][i] = *gptr<const u32x2>(inb + (size_t)i * L * 2 + vo);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = a.in_a ? a.in_a[b * 16 + 4 * cq + i] : 1.f;
            sv[i] = a.in_a ? a.in_s[b * 16 + 4 * cq + i] : 0.f;
        }
    };
    auto commit_x = [&](int pos0) {
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = tid + s * NTH, pq = idx >> 2;
            if (idx >= NIT) continue;
            const int pos = pos0 + 4 * pq;
            const bool ok = pos >= 0 && pos < L;
            unsigned char* dst = smem_n + XB + (4 * pq) * RB + cq * 8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (e & 1) ? n16_hi(pf[s][i][e >> 1]) : n16_lo(pf[s][i][e >> 1]);
                    y[i] = fmaf(av[i], xv, sv[i]);
                    v[i] = fmaxf(y[i], y[i] * slope);
                }
                u32x2 w = {n16_pack2(v[0], v[1]), n16_pack2(v[2], v[3])};
                u32x2 r = {n16_pack2(y[0], y[1]), n16_pack2(y[2], y[3])};
                if (!ok) { w = u32x2{0u, 0u}; r = w; }
                *reinterpret_cast<u32x2*>(dst + e * RB) = w;
                *reinterpret_cast<u32x2*>(dst + e * RB + RT) = r;
            }
        }
    };

    issue_x(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int b = tile / a.ntl, n0 = (tile - b * a.ntl) * nto;
        const bool edge = n0 - H2 < 0 || n0 - H2 + W > L;
        n16_lds_barrier();                                                        // the previous tile's stores have read the scratch
        commit_x(n0 - H1 - H2 - XOFF);
        n16_lds_barrier();

        f32x4 acc[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{b1[0], b1[1], b1[2], b1[3]};
        // conv_a: window column col <-> x row col + H1 + XOFF; tap t reads row col + XOFF + t * D1
        conv(std::integral_constant<int, D1>{}, std::integral_constant<int, 0>{}, acc, XB, col0 + XOFF);
        // u (0 outside the sequence: conv_b zero-pads lrelu(u)) -> the u tile as bf16 lrelu(u); no residual on the intermediate
        {
            int colv = col0;
            asm volatile("" : "+v"(colv));
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int col = colv + 16 * cb;
                f32x4 u = acc[cb];
                if (edge) {
                    const int pos = n0 - H2 + col;
                    if (pos < 0 || pos >= L) u = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const f32x4 us = u * slope;
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = fmaxf(u[r], us[r]);
                *reinterpret_cast<u32x2*>(smem_n + TB + col * RB + 8 * kg) = u32x2{n16_pack2(u[0], u[1]), n16_pack2(u[2], u[3])};
            }
        }
        n16_lds_barrier();
        // conv_b (dilation 1) on the same window + the pair's residual x (the r tile) + b2
        {
            u32x2 rw[NB];
            int colv = col0;
            asm volatile("" : "+v"(colv));
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) rw[cb] = *reinterpret_cast<const u32x2*>(smem_n + RT + (colv + 16 * cb + H1 + XOFF) * RB + 8 * kg);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb)
                acc[cb] = f32x4{b2[0], b2[1], b2[2], b2[3]} + f32x4{n16_lo(rw[cb][0]), n16_hi(rw[cb][0]), n16_lo(rw[cb][1]), n16_hi(rw[cb][1])};
        }
        conv(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, acc, TB, col0 - H2);

        // ---- the nto valid columns (window columns H2 .. H2 + nto) through the fp32 scratch [16][SRS] over the dead tiles, then 8-byte bf16 stores
        n16_lds_barrier();
        {
            float* const scr = reinterpret_cast<float*>(smem_n);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                int col = col0;
                asm volatile("" : "+v"(col));
                col += 16 * cb;
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(4 * kg + r) * SRS + col + SOFF] = acc[cb][r];
            }
            n16_lds_barrier();
            issue_x(min(tile + (int)gridDim.x, a.ntiles - 1));                    // the next tile's x: in flight under this tile's stores
            const int nq = nto >> 2;
            const unsigned magic = (unsigned)(((1ull << 32) + nq - 1) / nq);
            const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f;
            const size_t boff = (size_t)b * 16 * L * 2;
            unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + boff;
            for (int idx = tid; idx < 16 * nq; idx += NTH) {
                const int row = (int)__umulhi((unsigned)idx, magic), q = idx - row * nq;
                const int pos = n0 + 4 * q;
                if (pos >= L) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * SRS + H2 + SOFF + 4 * q);
                const unsigned eo = (unsigned)(row * L + pos) * 2u;
                if (a.add0) {                                                     // ((add0 + add1) + value): the reference's order over the branches
                    const u32x2 p0 = *gptr<const u32x2>(reinterpret_cast<const unsigned char*>(a.add0) + boff + eo);
                    f32x4 s = {n16_lo(p0[0]), n16_hi(p0[0]), n16_lo(p0[1]), n16_hi(p0[1])};
                    if (a.add1) {
                        const u32x2 p1 = *gptr<const u32x2>(reinterpret_cast<const unsigned char*>(a.add1) + boff + eo);
                        s += f32x4{n16_lo(p1[0]), n16_hi(p1[0]), n16_lo(p1[1]), n16_hi(p1[1])};
                    }
                    v = s + v;
                }
                if (a.out_div != 0.f) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], a.out_div, dinv);
                }
                *gptr<u32x2>(obase + eo) = u32x2{n16_pack2(v[0], v[1]), n16_pack2(v[2], v[3])};
            }
        }
    }
}

template <int WN, int K, int D1>
int launch_n16_pair(const v2w_stage_split_args* q, int pr, bool last, hipStream_t stream) {
    constexpr int H1 = D1 * (K - 1) / 2, H2 = (K - 1) / 2, XOFF = (4 - (H1 + H2) % 4) % 4;
    constexpr int NTH = 64 * WN, W = 128 * WN, XR = (W + 2 * H1 + XOFF + 3) & ~3;
    N16PairArgs p{};
    p.in = reinterpret_cast<const unsigned short*>(q->in_b[pr] ? q->in_b[pr] : q->in);
    p.in_a = q->in_a; p.in_s = q->in_s;
    p.w1 = static_cast<const unsigned char*>(q->wps1[pr]); p.bias1 = q->bias1[pr];
    p.w2 = static_cast<const unsigned char*>(q->wps2[pr]); p.bias2 = q->bias2[pr];
    p.out = reinterpret_cast<unsigned short*>(q->out_b[pr]);
    if (last) { p.add0 = reinterpret_cast<const unsigned short*>(q->add0); p.add1 = reinterpret_cast<const unsigned short*>(q->add1); p.out_div = q->out_div; }
    p.B = q->B; p.L = q->L; p.slope = q->slope;
    p.nto = (W - 2 * H2) & ~3;
    p.ntl = (q->L + p.nto - 1) / p.nto;
    if ((long long)q->B * p.ntl > 0x7fffffffll) return V2W_E_SHAPE;
    p.ntiles = q->B * p.ntl;
    const size_t lds = (size_t)(XR + W + 16 + XR) * 32;
    if (lds > 80 * 1024) return V2W_E_SHAPE;                                     // two workgroups per CU
    if (v2w_dry(stream)) return 0;
    const int slots = v2w_num_cus() * (8 / WN);
    auto kern = n16_pair_kernel<WN, K, D1>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(p.ntiles < slots ? p.ntiles : slots), dim3(NTH), lds, stream, p);
    return v2w_launch_status();
}

template <int WN>
int launch_n16_pair_kd(const v2w_stage_split_args* q, int pr, bool last, hipStream_t stream) {
    const int key = q->k[pr] * 10 + q->dil1[pr];
    switch (key) {
        case 31:  return launch_n16_pair<WN, 3, 1>(q, pr, last, stream);
        case 33:  return launch_n16_pair<WN, 3, 3>(q, pr, last, stream);
        case 35:  return launch_n16_pair<WN, 3, 5>(q, pr, last, stream);
        case 71:  return launch_n16_pair<WN, 7, 1>(q, pr, last, stream);
        case 73:  return launch_n16_pair<WN, 7, 3>(q, pr, last, stream);
        case 75:  return launch_n16_pair<WN, 7, 5>(q, pr, last, stream);
        case 111: return launch_n16_pair<WN, 11, 1>(q, pr, last, stream);
        case 113: return launch_n16_pair<WN, 11, 3>(q, pr, last, stream);
        case 115: return launch_n16_pair<WN, 11, 5>(q, pr, last, stream);
        default:  return V2W_E_SHAPE;
    }
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_n16)
#endif

#define V2W_N16_WN 4

int v2w_resblock2_stage_bf16_n16s(const v2w_stage_split_args* a, hipStream_t stream);     // v2w_stage_bf16_n16s.hip: the streaming form, stage + tail

// Called by v2w_resblock2_stage_bf16 (v2w_stage_bf16.hip) for C = 16 on bf16 tensors.  V2W_E_SHAPE: not the reference's block set / not
// aligned - the caller runs its own kernels.
int v2w_resblock2_stage_bf16_n16(const v2w_stage_split_args* a, hipStream_t stream) {
    if (a->C != 16 || a->io_bf16 != 3 || !a->bf16 || a->nk != 3 || a->up_out) return V2W_E_SHAPE;
    if (a->post_out && (a->post_k != 7 || !a->post_w || (reinterpret_cast<uintptr_t>(a->post_out) & 15))) return V2W_E_SHAPE;     // the fused tail: 7 taps
    for (int j = 0; j < 3; ++j)
        if (a->k[j] != 3 + 4 * j || a->dil1[j] != 1 || a->dil2[j] != 3 || !a->wps1[j] || !a->wps2[j]) return V2W_E_SHAPE;
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    if (a->L % 4 != 0 || !al16(a->in) || !al16(a->out) || (!a->out && !a->post_out)) return V2W_E_SHAPE;
    if ((long long)16 * a->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;            // 32-bit offsets inside one batch item
    if (!(a->slope > 0.f && a->slope < 1.f)) return V2W_E_SHAPE;                // lrelu as max(v, slope v), undone as min(a, a / slope)
    if (a->post_out) {                               // stage + tail: the streaming kernel (one wave per workgroup, no barriers)
        const int rc = v2w_resblock2_stage_bf16_n16s(a, stream);
        if (rc != V2W_E_SHAPE) return rc;
    }
    return launch_n16<V2W_N16_WN>(a, stream);
}

// Called by v2w_resblock2_stage_bf16 (v2w_stage_bf16.hip) for the ResBlock1 pair mode (rb1) at C = 16: one launch per problem (branch).
// V2W_E_SHAPE - nothing launched - unless EVERY problem has a kernel (k in {3, 7, 11}, first dilation in {1, 3, 5}, second 1, aligned bf16 tensors).
int v2w_resblock1_pairs_bf16_n16(const v2w_stage_split_args* a, hipStream_t stream) {
    if (!a->rb1 || a->C != 16 || a->io_bf16 != 3 || !a->bf16 || a->nk < 1 || a->nk > 4) return V2W_E_SHAPE;
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    if (a->L % 4 != 0 || (long long)16 * a->L * 2 >= (1ll << 31) || !(a->slope > 0.f && a->slope < 1.f)) return V2W_E_SHAPE;
    if (!al16(a->add0) || !al16(a->add1) || (a->add1 && !a->add0)) return V2W_E_SHAPE;
    for (int p = 0; p < a->nk; ++p) {
        const void* src = a->in_b[p] ? a->in_b[p] : static_cast<const void*>(a->in);
        if (!al16(src) || !al16(a->out_b[p]) || a->dil2[p] != 1) return V2W_E_SHAPE;
        if (!v2w_dry(stream) && (!src || !a->out_b[p] || !a->wps1[p] || !a->wps2[p])) return V2W_E_ARG;
        const int rc = launch_n16_pair_kd<V2W_N16_WN>(a, p, false, V2W_DRY_STREAM);          // every problem first: all or nothing
        if (rc != 0) return rc;
    }
    if (v2w_dry(stream)) return 0;
    for (int p = 0; p < a->nk; ++p) {
        const int rc = launch_n16_pair_kd<V2W_N16_WN>(a, p, p == a->nk - 1, stream);
        if (rc != 0) return rc;
    }
    return 0;
}
