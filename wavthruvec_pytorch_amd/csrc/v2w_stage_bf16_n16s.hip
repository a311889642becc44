// The 16-channel ResBlock2 stage of the generator AND its tail on bf16 tensors as a STREAMING kernel (reference: vec2wav/models.py:135-141
// with h.resblock_kernel_sizes (3, 7, 11), dilations (1, 3), then models.py:143-145):
//   out = ( sum_j [ t1_j + conv_{k_j, 3}(lrelu(t1_j)) + b2_j ] ) / 3,   t1_j = x + conv_{k_j, 1}(lrelu(x)) + b1_j,   x = a * in + s,
//   y   = tanh(conv_post(lrelu(out, post_slope)) + post_b).
//
// Why a third form of this stage.  n16_stage_kernel (v2w_stage_bf16_n16.hip: weights in registers, a 512-position tile per 4-wave workgroup)
// measured 0.16 of the bf16 MFMA peak: 6.8 vector instructions per MFMA, 37 % of its LDS cycles in bank conflicts, thirteen workgroup barriers
// per tile, and one operand read per MFMA - exactly the LDS's 256 B / clk / CU.  This kernel changes the work, not the layout:
//   * ONE WAVE IS ONE WORKGROUP and walks along the sequence, 16 positions (one MFMA column block) per step, through wave-private LDS rings
//     (activated x, raw x, the three t1_j, z): no barrier anywhere, no halo recomputed between neighbours except at the ends of a run
//     (a run = R positions of one batch row, R chosen by the launcher), no accumulator arrays - a step owns 5 accumulators;
//   * the first convs of the three branches (k = 3, 7, 11 at dilation 1) read the SAME window of x: one operand read (pair of taps) feeds the
//     k = 11, the k = 7 and the k = 3 weights - 6 reads for 14 MFMAs instead of 12 for 12;
//   * the residual x of t1_j rides in the MFMA: the 12th tap slot of the k = 11 pair grid is free, its lanes read the raw-x ring against an
//     identity block (exact: 1.0 * bf16 accumulates without rounding); the biases are the accumulators' initial values; the running sum of
//     the t1_j is the initial value of the second convs' accumulator;
//   * the tail runs on the matrix pipe too: conv_post (16 -> 1, 7 taps) as 4 MFMAs per block on the z ring with the weight's hi and lo bf16
//     halves in output rows 0 and 1 (fp32 weights to 16 mantissa bits), / 3 folded into them (lrelu is positively homogeneous);
//   * LDS tiles are two planes of 16-byte rows (channels 0-7 / 8-15): operand reads are conflict-free at every tap offset, the 8-byte
//     epilogue writes 2-way instead of 4-way.
// Per step and wave: 22 ds_read_b128, 30 v_mfma_f32_16x16x32_bf16, ~90 vector instructions (3 per MFMA), 8 ds_write_b64.
// Step s of a run runs conv1 of block s, conv2 of block s - 2 and the tail of block s - 4: the three chains of a step are independent, all
// their operand reads form one sequence with 8 reads in flight (inline assembly: hipcc serialises read -> wait -> MFMA at this register
// count), the epilogues follow.  Ring slots: block m of a 4-block ring lives at slot m & 3 (+ 1); slot 0 is mirrored behind slot 3 and slot 3
// in front of slot 0, so that a window of three consecutive blocks is linear in memory at every step and every offset is an immediate.
#include <type_traits>
#include <utility>
#include "v2w_tile.h"

namespace {

typedef __bf16 sb8 __attribute__((ext_vector_type(8)));
typedef unsigned int su32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));

struct N16SArgs {
    const unsigned short* in; const float* in_a; const float* in_s;
    const unsigned char* w1[3]; const float* bias1[3];
    const unsigned char* w2[3]; const float* bias2[3];
    const float* post_w; const float* post_b; float* post_out;
    int B, L, R, rpr, nruns;                     // run length (multiple of 64), runs per batch row, B * rpr
    float slope, out_div, post_slope;
};

constexpr int S_BLK = 256;                                               // one 16-row block of one plane
constexpr int S_XP = 10 * S_BLK, S_RP = 8 * S_BLK, S_TP = 6 * S_BLK;     // plane strides: x ring (8 + 2 mirrors), raw-x ring (8), t1 / z rings (4 + 2)
constexpr int S_XOFF = 0, S_ROFF = S_XOFF + 2 * S_XP, S_TOFF = S_ROFF + 2 * S_RP, S_ZOFF = S_TOFF + 6 * S_TP, S_LDS = S_ZOFF + 2 * S_TP;
static_assert(S_LDS == 21504, "seven waves per CU");

__device__ __forceinline__ unsigned int s_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float s_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float s_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// max(v, w) as ONE instruction: fmaxf(v, w) costs two here - hipcc first canonicalises an operand that comes out of an MFMA with v_max_f32(v, v)
__device__ __forceinline__ float s_max(float v, float w) { float t; asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(w)); return t; }
// scalar-form multiplies and adds beside MFMAs: left to -O3 these are SLP-packed into v_pk_mul_f32 / v_pk_add_f32, which issue at well under half the
// rate of two plain instructions next to matrix work (MI355X guide, 'price of one filler beside MFMAs')
__device__ __forceinline__ float s_mul(float v, float w) { float t; asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(w)); return t; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float t; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c)); return t; }
__device__ __forceinline__ float s_add(float v, float w) { float t; asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(w)); return t; }

template <int... I, class F> __device__ __forceinline__ void s_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// ---- the operand reads and MFMAs of one step, as tables (I = step number mod 4: every ring slot of the step is a compile-time constant)
struct SRd { int base, imm; };              // base register: 0 conv1 pairs (x ring), 1 conv1 pair 5 (x ring | raw-x ring), 2 conv2 pairs (slot 1: + 3 rows),
                                            // 3 last pair of a conv2 / of the tail (both slots read the real tap's rows), 4 tail pairs (slot 1: + 1 row)
struct SMm { int acc, w, rd, first; };      // accumulator 0..2 conv1 (k = 3, 7, 11), 3 conv2, 4 tail; weight operand; the read it consumes; first of its chain
constexpr int S_NR = 22, S_NM = 30;
// weight operands: 0..5 k = 11 first conv (pair 5 = tap 10 | identity), 6..9 k = 7, 10..11 k = 3, 12 (0 | identity), 13.. second convs (2 + 4 + 6), 25..28 tail
constexpr int S_W11 = 0, S_W7 = 6, S_W3 = 10, S_WRES = 12, S_W2 = 13, S_WP = 25, S_NW = 29;
template <int I> struct StepProg {
    SRd rd[S_NR]; SMm mm[S_NM];
    constexpr StepProg() : rd{}, mm{} {
        int nr = 0, nm = 0;
        // conv1 of block s: x ring slot (4 (g & 1) + I) - the 4 (g & 1) part is in the base register
        for (int q = 0; q < 6; ++q) {
            if (q < 5) rd[nr] = SRd{0, S_XOFF + (I + 1) * S_BLK + (2 * q - 5) * 16};
            else rd[nr] = SRd{1, I * S_BLK};
            mm[nm++] = SMm{2, S_W11 + q, nr, q == 0};
            if (q >= 1 && q <= 4) mm[nm++] = SMm{1, S_W7 + q - 1, nr, q == 1};
            if (q == 5) mm[nm++] = SMm{1, S_WRES, nr, 0};
            if (q == 2 || q == 3) mm[nm++] = SMm{0, S_W3 + q - 2, nr, q == 2};
            if (q == 5) mm[nm++] = SMm{0, S_WRES, nr, 0};
            ++nr;
        }
        // conv2 of block s - 2: t1 ring slot (I + 2) & 3
        int w = S_W2;
        for (int b = 0; b < 3; ++b) {
            const int k = 3 + 4 * b, h = (k - 1) / 2, np = (k + 1) / 2;
            for (int p = 0; p < np; ++p) {
                rd[nr] = SRd{p == np - 1 ? 3 : 2, S_TOFF + b * 2 * S_TP + (((I + 2) & 3) + 1) * S_BLK + (6 * p - 3 * h) * 16};
                mm[nm++] = SMm{3, w++, nr, b == 0 && p == 0};
                ++nr;
            }
        }
        // tail of block s - 4: z ring slot I
        for (int p = 0; p < 4; ++p) {
            rd[nr] = SRd{p == 3 ? 3 : 4, S_ZOFF + (I + 1) * S_BLK + (2 * p - 3) * 16};
            mm[nm++] = SMm{4, S_WP + p, nr, p == 0};
            ++nr;
        }
    }
    constexpr int last_use(int r) const { int l = 0; for (int m = 0; m < S_NM; ++m) if (mm[m].rd == r) l = m; return l; }
    constexpr int first_use(int r) const { for (int m = 0; m < S_NM; ++m) if (mm[m].rd == r) return m; return 0; }
};
template <int I> inline constexpr StepProg<I> kStepProg{};

__global__ void __launch_bounds__(64, 2)
n16s_stage_kernel(const N16SArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    const int lane = threadIdx.x;
    const int j = lane & 15, kg = lane >> 4, h = kg & 1, sl = kg >> 1;
    const int L = __builtin_amdgcn_readfirstlane(a.L), R = __builtin_amdgcn_readfirstlane(a.R), rpr = __builtin_amdgcn_readfirstlane(a.rpr);
    const float slope = a.slope, pslope = a.post_slope;

    // ---- every weight of the stage and of the tail into registers (116 of them).  Fragment of tap t (v2w_pack_bf16, 2 KiB): lane' = row + 32 h'
    // holds the input channels 8 h' .. 8 h' + 7 of output channel `row`; this lane is output channel j, k-group kg = (tap slot sl, channel half h).
    su32x4 W[S_NW];
    {
        const unsigned lo16 = (unsigned)(j + 32 * h) * 16u;
        auto frag = [&](const unsigned char* w, int t, int K) {
            su32x4 v = {0u, 0u, 0u, 0u};
            if (t < K) v = *reinterpret_cast<const su32x4*>(w + (size_t)t * 2048 + lo16);
            return v;
        };
        su32x4 ident;                                                           // element e of this lane: input channel 8 h + e against output channel j
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) ident[wd] = (8 * h + 2 * wd == j ? 0x3f80u : 0u) | (8 * h + 2 * wd + 1 == j ? 0x3f800000u : 0u);
        const su32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int p = 0; p < 6; ++p) W[S_W11 + p] = frag(a.w1[2], 2 * p + sl, 11);
        if (sl) W[S_W11 + 5] = ident;
#pragma unroll
        for (int p = 0; p < 4; ++p) W[S_W7 + p] = frag(a.w1[1], 2 * p + sl, 7);
#pragma unroll
        for (int p = 0; p < 2; ++p) W[S_W3 + p] = frag(a.w1[0], 2 * p + sl, 3);
        W[S_WRES] = sl ? ident : zero4;
#pragma unroll
        for (int p = 0; p < 2; ++p) W[S_W2 + p] = frag(a.w2[0], 2 * p + sl, 3);
#pragma unroll
        for (int p = 0; p < 4; ++p) W[S_W2 + 2 + p] = frag(a.w2[1], 2 * p + sl, 7);
#pragma unroll
        for (int p = 0; p < 6; ++p) W[S_W2 + 6 + p] = frag(a.w2[2], 2 * p + sl, 11);
        // the tail: row 0 = bf16(w / out_div), row 1 = bf16 of what the rounding left
        const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            su32x4 v = zero4;
            const int t = 2 * p + sl;
            if (t < 7 && j < 2) {
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    float f0 = a.post_w[t * 16 + 8 * h + 2 * wd] * dinv, f1 = a.post_w[t * 16 + 8 * h + 2 * wd + 1] * dinv;
                    if (j == 1) { f0 -= (float)(__bf16)f0; f1 -= (float)(__bf16)f1; }
                    v[wd] = s_pack2(f0, f1);
                }
            }
            W[S_WP + p] = v;
        }
    }
    // biases of this lane's 4 channels (4 kg .. 4 kg + 3): the first convs' accumulators start at b1_j, the running output at the sum of the b2_j
    f32x4 b1v[3], b2s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b1v[b][r] = a.bias1[b] ? a.bias1[b][4 * kg + r] : 0.f;
            b2s[r] += a.bias2[b] ? a.bias2[b][4 * kg + r] : 0.f;
        }
    }
    const float pb = a.post_b ? a.post_b[0] : 0.f;

    // ---- LDS addresses.  Operand reads: lane (j, kg) takes the 16 bytes of plane h, row (block row + j + tap offset); the second tap slot of a
    // pair (sl = 1) is one dilation step on.  Immediate offsets carry ring, slot and tap; negative tap offsets reach into the slot in front.
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_s);
    const unsigned rb_t = lds0 + (unsigned)(h * S_TP + j * 16);                 // t1 / z rings, no slot offset (base 3)
    unsigned rbase[5];
    rbase[0] = lds0 + (unsigned)(h * S_XP + j * 16 + sl * 16);
    rbase[1] = sl ? lds0 + (unsigned)(S_ROFF + h * S_RP + j * 16) : lds0 + (unsigned)(S_XOFF + h * S_XP + j * 16 + S_BLK + 5 * 16);
    rbase[2] = rb_t + (unsigned)(sl * 48);
    rbase[3] = rb_t;
    rbase[4] = rb_t + (unsigned)(sl * 16);
    asm volatile("" : "+v"(rbase[2]), "+v"(rbase[3]), "+v"(rbase[4]));            // (opaque: one register each, never re-derived per read)
    // epilogue writes: this lane's 4 channels (4 kg ..) of position j = 8 bytes at plane sl, row j, half h
    unsigned char* const wbase = smem_s + sl * S_TP + j * 16 + h * 8;
    // staging: thread (channel quad cq, position quad c) of a 64-position burst; plane cq >> 1, half cq & 1
    const int cq = lane & 3, cpos = lane >> 2, cblk = cpos >> 2;
    unsigned char* const stx = smem_s + S_XOFF + (cq >> 1) * S_XP + (4 * (cpos & 3)) * 16 + (cq & 1) * 8 + S_BLK;
    unsigned char* const str = smem_s + S_ROFF + (cq >> 1) * S_RP + (4 * (cpos & 3)) * 16 + (cq & 1) * 8;

    auto mfma = [](f32x4 c, su32x4 av, su32x4 bv) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sb8, av), __builtin_bit_cast(sb8, bv), c, 0, 0, 0);
    };

    su32x2 pf[4];
    float av[4], sv[4];
    f32x4 ts[2] = {b2s, b2s};                                                   // running sums of the t1_j of the blocks s - 1 and s - 2 (+ the b2_j)

    for (int run = blockIdx.x; run < a.nruns; run += gridDim.x) {
        const int b = run / rpr, p0 = (run - b * rpr) * R;
        const int nblk = (min(R, L - p0) + 15) >> 4, ngrp = ((nblk + 3) >> 2) + 1;
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * 16 * L * 2;
        float* const yb = a.post_out + (size_t)b * L;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = a.in_a ? a.in_a[b * 16 + 4 * cq + i] : 1.f;
            sv[i] = a.in_a ? a.in_s[b * 16 + 4 * cq + i] : 0.f;
        }
        // burst k = the x blocks 4 k + 1 .. 4 k + 4 of the run (positions p0 + 64 k + 16 ..): four 8-byte loads per thread (4 channels x 4 positions)
        auto issue_x = [&](int k) {
            const int pos = p0 + 64 * k + 16 + 4 * cpos;
            const bool ok = pos >= 0 && pos < L;
            unsigned vo = (unsigned)(4 * cq * L + (ok ? pos : 0)) * 2u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = *gptr<const su32x2>(inb + (size_t)i * L * 2 + vo);
        };
        // rows of the x ring: lrelu(x) (the conv operand); rows of the raw-x ring: x itself (the residual); bf16, exactly 0 outside the sequence
        auto commit_x = [&](int k) {
            const int pos = p0 + 64 * k + 16 + 4 * cpos;
            const bool ok = pos >= 0 && pos < L;                                // L % 4 == 0: a position quad is inside or outside as a whole
            const int slot = (4 * (k & 1) + 1 + cblk) & 7;
            unsigned char* const dx = stx + slot * S_BLK;
            unsigned char* const dr = str + slot * S_BLK;
            // mirrors (odd bursts only): slot 7 also in front of slot 0, slot 0 also behind slot 7
            const int mir = slot == 7 ? -8 * S_BLK : (slot == 0 ? 8 * S_BLK : 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (e & 1) ? s_hi(pf[i][e >> 1]) : s_lo(pf[i][e >> 1]);
                    y[i] = s_fma(av[i], xv, sv[i]);
                    v[i] = s_max(y[i], s_mul(y[i], slope));
                }
                su32x2 w = {s_pack2(v[0], v[1]), s_pack2(v[2], v[3])};
                su32x2 r = {s_pack2(y[0], y[1]), s_pack2(y[2], y[3])};
                if (!ok) { w = su32x2{0u, 0u}; r = w; }
                *reinterpret_cast<su32x2*>(dx + e * 16) = w;
                *reinterpret_cast<su32x2*>(dr + e * 16) = r;
                if (mir != 0) *reinterpret_cast<su32x2*>(dx + e * 16 + mir) = w;
            }
        };

        // ---- one step: conv1 of block s (x ring slot 4 (g & 1) + I), conv2 of block s - 2, the tail of block s - 4
        auto step = [&](auto i_c, int s, unsigned bx0, unsigned bx1) __attribute__((always_inline)) {
            constexpr int I = decltype(i_c)::value;
            constexpr int RING = 8;
            const unsigned bt2 = rbase[2], bt3 = rbase[3], bt4 = rbase[4];
#ifdef V2W_TIMELINE
            const bool tl = s == 9 && run == (int)blockIdx.x;                  // one steady-state step of the wave's first run
#define S_STAMP(k) do { if (tl) V2W_STAMP(k); } while (0)
#else
#define S_STAMP(k) ((void)0)
#endif
            S_STAMP(0);
            f32x4 acc[5];
            const f32x4 init[5] = {b1v[0], b1v[1], b1v[2], ts[I & 1], f32x4{0.f, 0.f, 0.f, 0.f}};
            su32x4 ring[RING];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto rd = [&ring, &bx0, &bx1, &bt2, &bt3, &bt4](auto n_c) __attribute__((always_inline)) {
                constexpr int n = decltype(n_c)::value;
                constexpr int bs = kStepProg<I>.rd[n].base, imm = kStepProg<I>.rd[n].imm;
                static_assert(imm >= 0 && imm < 65536, "ds_read offset field");
                if constexpr (bs == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(bx0), "n"(imm));
                else if constexpr (bs == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(bx1), "n"(imm));
                else if constexpr (bs == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(bt2), "n"(imm));
                else if constexpr (bs == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(bt3), "n"(imm));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(bt4), "n"(imm));
            };
            S_STAMP(1);
            s_for(std::make_integer_sequence<int, RING>{}, rd);
            s_for(std::make_integer_sequence<int, S_NM>{}, [&ring, &acc, &init, &W, &rd, &mfma](auto m_c) __attribute__((always_inline)) {
                constexpr int m = decltype(m_c)::value;
                constexpr SMm q = kStepProg<I>.mm[m];
                if constexpr (kStepProg<I>.first_use(q.rd) == m) {
                    // LDS reads return in order: before the first use of read r at most min(RING - 1, NR - 1 - r) younger ones are outstanding
                    constexpr int left = (S_NR - 1 - q.rd) < (RING - 1) ? (S_NR - 1 - q.rd) : (RING - 1);
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[q.rd % RING]) : "n"(left));
                }
                if constexpr (q.first) acc[q.acc] = mfma(init[q.acc], W[q.w], ring[q.rd % RING]);
                else acc[q.acc] = mfma(acc[q.acc], W[q.w], ring[q.rd % RING]);
                if constexpr (kStepProg<I>.last_use(q.rd) == m && q.rd + RING < S_NR) rd(std::integral_constant<int, q.rd + RING>{});
            });
            __builtin_amdgcn_sched_barrier(0);
            S_STAMP(2);

            // ---- t1 epilogue of block s: acc[jb] = t1_jb (bias, residual and conv) at position p0 + 16 s + j, channels 4 kg ..; 0 outside the
            // sequence (conv2 zero-pads lrelu(t1)); the running output takes t1 in fp32, the ring lrelu(t1) as bf16
            const int pos1 = p0 + 16 * s + j;
            const bool edge1 = p0 + 16 * s < 0 || p0 + 16 * s + 16 > L;
            f32x4 tsum = b2s;
#pragma unroll
            for (int jb = 0; jb < 3; ++jb) {
                f32x4 t1v = acc[jb];
                if (edge1) {                                                    // (a uniform branch: the empty asm keeps hipcc from turning it into 4 selects per branch)
                    asm volatile("" ::: "memory");
                    if (pos1 < 0 || pos1 >= L) t1v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { tsum[r] = s_add(tsum[r], t1v[r]); t1v[r] = s_max(t1v[r], s_mul(t1v[r], slope)); }
                const su32x2 w = {s_pack2(t1v[0], t1v[1]), s_pack2(t1v[2], t1v[3])};
                unsigned char* const d = wbase + S_TOFF + jb * 2 * S_TP;
                *reinterpret_cast<su32x2*>(d + (I + 1) * S_BLK) = w;
                if constexpr (I == 0) *reinterpret_cast<su32x2*>(d + 5 * S_BLK) = w;
                if constexpr (I == 3) *reinterpret_cast<su32x2*>(d) = w;
            }
            ts[I & 1] = tsum;
            S_STAMP(3);
            // ---- z epilogue of block s - 2: z = lrelu(sum, post_slope) (the division by 3 sits in the tail's weights), 0 outside the sequence
            {
                const int pos2 = pos1 - 32;
                const bool edge2 = p0 + 16 * s - 32 < 0 || p0 + 16 * s - 16 > L;
                f32x4 z = acc[3];
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = s_max(z[r], s_mul(z[r], pslope));
                if (edge2) {
                    asm volatile("" ::: "memory");
                    if (pos2 < 0 || pos2 >= L) z = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const su32x2 w = {s_pack2(z[0], z[1]), s_pack2(z[2], z[3])};
                constexpr int ZS = (I + 2) & 3;
                unsigned char* const d = wbase + S_ZOFF;
                *reinterpret_cast<su32x2*>(d + (ZS + 1) * S_BLK) = w;
                if constexpr (ZS == 0) *reinterpret_cast<su32x2*>(d + 5 * S_BLK) = w;
                if constexpr (ZS == 3) *reinterpret_cast<su32x2*>(d) = w;
            }
            S_STAMP(4);
            // ---- the tail of block s - 4: rows 0 / 1 of the accumulator (lanes 0 .. 15) hold the hi / lo weight halves' sums for position j
            {
                const int pos4 = pos1 - 64;
                const float v = (acc[4][0] + acc[4][1]) + pb;
                // tanh(v) = 1 - 2 / (1 + e^(2 v)): e^(2 v) = inf -> 1, = 0 -> -1
                const float t = __builtin_amdgcn_exp2f(v * 2.8853900817779268f);
                const float y = fmaf(-2.f, __builtin_amdgcn_rcpf(t + 1.f), 1.f);
                if (kg == 0 && s >= 4 && s - 4 < nblk && pos4 < L) *gptr<float>(reinterpret_cast<unsigned char*>(yb) + (unsigned)pos4 * 4u) = y;
            }
            S_STAMP(5);
            if (s == 10 && run == (int)blockIdx.x) V2W_STAMP(6);              // (the start of the next step)
        };

        // ---- the run: burst -1 (blocks -3 .. 0), steps -2 and -1 (conv1 of the two blocks in front of the run; their conv2 / tail halves work on
        // stale rings and are never stored), then groups of four steps, each behind the commit of its burst
        issue_x(-1);
        commit_x(-1);
        issue_x(0);
        {
            unsigned bx0 = rbase[0] + 4u * S_BLK, bx1 = rbase[1] + 4u * S_BLK;
            asm volatile("" : "+v"(bx0), "+v"(bx1));
            step(std::integral_constant<int, 2>{}, -2, bx0, bx1);
            step(std::integral_constant<int, 3>{}, -1, bx0, bx1);
        }
        for (int g = 0; g < ngrp; ++g) {
            commit_x(g);
            issue_x(g + 1);
            unsigned bx0 = rbase[0] + ((g & 1) ? 4u * S_BLK : 0u), bx1 = rbase[1] + ((g & 1) ? 4u * S_BLK : 0u);
            asm volatile("" : "+v"(bx0), "+v"(bx1));
            step(std::integral_constant<int, 0>{}, 4 * g, bx0, bx1);
            step(std::integral_constant<int, 1>{}, 4 * g + 1, bx0, bx1);
            step(std::integral_constant<int, 2>{}, 4 * g + 2, bx0, bx1);
            step(std::integral_constant<int, 3>{}, 4 * g + 3, bx0, bx1);
        }
    }
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_n16s)
#endif

// Called by v2w_resblock2_stage_bf16_n16 (v2w_stage_bf16_n16.hip) for the 16-channel stage WITH the generator's 7-tap tail on bf16 tensors.
// V2W_E_SHAPE: not served (the caller runs n16_stage_kernel).  Host-only when `stream` is the dry-run sentinel.
int v2w_resblock2_stage_bf16_n16s(const v2w_stage_split_args* q, hipStream_t stream) {
    if (!q->post_out || q->post_k != 7 || !q->post_w || q->up_out) return V2W_E_SHAPE;
    if (!(q->post_slope > 0.f && q->post_slope < 1.f) || !(q->slope > 0.f && q->slope < 1.f)) return V2W_E_SHAPE;     // lrelu as max(v, slope v)
    if (q->L % 4 != 0 || (long long)16 * q->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;
    N16SArgs p{};
    p.in = reinterpret_cast<const unsigned short*>(q->in); p.in_a = q->in_a; p.in_s = q->in_s;
    for (int j = 0; j < 3; ++j) {
        p.w1[j] = static_cast<const unsigned char*>(q->wps1[j]); p.bias1[j] = q->bias1[j];
        p.w2[j] = static_cast<const unsigned char*>(q->wps2[j]); p.bias2[j] = q->bias2[j];
    }
    p.post_w = q->post_w; p.post_b = q->post_b; p.post_out = q->post_out;
    p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div; p.post_slope = q->post_slope;
    const int ncu = v2w_num_cus();
    const int nwaves = ncu * 7;
    // run length: the multiple of 64 positions that minimises (runs per wave) x (steps per run); a run costs its blocks + 6 steps of lead-in / drain
    long long best = -1; int bestR = 64;
    for (int R = 64; R <= 4096; R += 64) {
        const long long rpr = (q->L + R - 1) / R, runs = rpr * q->B;
        const long long cost = ((runs + nwaves - 1) / nwaves) * (R / 16 + 8);
        if (best < 0 || cost < best) { best = cost; bestR = R; }
        if (R >= q->L) break;
    }
    p.R = bestR; p.rpr = (q->L + bestR - 1) / bestR;
    if ((long long)q->B * p.rpr > 0x7fffffffll) return V2W_E_SHAPE;
    p.nruns = q->B * p.rpr;
    if (v2w_dry(stream)) return 0;
    V2W_LAUNCH(n16s_stage_kernel, dim3(p.nruns < nwaves ? p.nruns : nwaves), dim3(64), S_LDS, stream, p);
    return v2w_launch_status();
}
