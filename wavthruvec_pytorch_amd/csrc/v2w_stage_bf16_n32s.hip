// The 32-channel ResBlock2 stage of the generator AND the next stage's upsampler on bf16 tensors as a STREAMING kernel of four-wave teams
// (reference: vec2wav/models.py:135-141 with h.resblock_kernel_sizes (3, 7, 11), dilations (1, 3), then models.py:128-129 of the next loop
// iteration: leaky_relu -> ConvTranspose1d(32 -> 16, k = 4, stride 2) + bias, and the BatchNorm partial sums of its output, modules.py:23):
//   out = ( sum_j [ t1_j + conv_{k_j, 3}(lrelu(t1_j)) + b2_j ] ) / 3,   t1_j = x + conv_{k_j, 1}(lrelu(x)) + b1_j,   x = a * in + s,
//   up  = convT(lrelu(out, up_slope)) + up_bias.
//
// The streaming form of v2w_stage_bf16_n16s.hip (one wave walks along the sequence, 16 positions per step, through LDS rings; no halo, no
// accumulator arrays) needs every weight of the stage in registers.  At 32 channels that is 84 KB of bf16 fragments + 6 KB for the upsampler -
// more than one wave's register file - so FOUR waves form a team and each holds a quarter, by ROLE:
//   wave 0   conv1 of the k = 11 branch (both 16-channel halves of the output)        + the upsampler's channels 0..7
//   wave 1   conv1 of the k = 7 and k = 3 branches (they read the SAME seven windows)
//   wave 2   conv2 of the k = 11 branch, the sum of the branches, z = lrelu(out / 3)   + half of the staging of x
//   wave 3   conv2 of the k = 7 and k = 3 branches + the upsampler's channels 8..15     + half of the staging of x
// v_mfma_f32_16x16x32_bf16 with K = the 32 input channels of ONE tap: an operand read (16 positions x 32 channels, 1 KB) feeds both output
// halves, and for wave 1 both branches.  The waves of a team run in lock step, one s_barrier per step; the pipeline is skewed so that nobody
// waits for data of the same step: step s runs conv1 of block s, conv2 of block s - 2, the branch sum of block s - 3 (wave 3 hands its partial
// sums to wave 2 through a 2 KB exchange buffer) and the upsampler of block s - 5.  t1_j goes to TWO rings: lrelu(t1_j) (conv2's operand) and
// t1_j itself (bf16), which wave 2 / 3 add back through an identity block on the matrix pipe - as conv1 adds x from the raw-x ring.  (The
// reference under autocast rounds every one of these tensors to bf16 as well.)  LDS tiles: four planes of 16-byte rows (8 channels each).
// Per step and team: 50 ds_read_b128, 102 MFMAs; 58 KB of LDS, two teams per CU.
#include <type_traits>
#include <utility>
#include "v2w_tile.h"

namespace {

typedef __bf16 tb8 __attribute__((ext_vector_type(8)));
typedef unsigned int tu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int tu32x4 __attribute__((ext_vector_type(4)));

struct N32SArgs {
    const unsigned short* in; const float* in_a; const float* in_s;
    const unsigned char* w1[3]; const float* bias1[3];
    const unsigned char* w2[3]; const float* bias2[3];
    const unsigned char* up_w; const float* up_bias; unsigned short* up_out; float* up_stats;
    int B, L, R, rpr, nruns;                     // run length (multiple of 64), runs per batch row, B * rpr (= rows of up_stats)
    float slope, out_div, up_slope;
};

constexpr int T_BLK = 256;                                                          // one 16-row block of one plane
constexpr int T_XP = 10 * T_BLK, T_RP = 8 * T_BLK, T_AP = 6 * T_BLK, T_QP = 4 * T_BLK;   // plane strides: x (8 + 2 mirrors), raw x (8), lrelu(t1) / z (4 + 2), t1 (4)
constexpr int T_XOFF = 0, T_ROFF = T_XOFF + 4 * T_XP, T_AOFF = T_ROFF + 4 * T_RP, T_QOFF = T_AOFF + 12 * T_AP, T_ZOFF = T_QOFF + 12 * T_QP,
              T_EOFF = T_ZOFF + 4 * T_AP, T_LDS = T_EOFF + 2 * 2048;
static_assert(T_LDS == 59392, "two teams per CU");

__device__ __forceinline__ unsigned int t_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float t_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float t_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// one-instruction forms: fmaxf canonicalises an MFMA result first (v_max x, x), and -O3 SLP-packs adjacent f32 multiplies into v_pk_mul_f32,
// which issues slower than two plain multiplies beside matrix work
__device__ __forceinline__ float t_max(float v, float w) { float t; asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(w)); return t; }
__device__ __forceinline__ float t_mul(float v, float w) { float t; asm("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(w)); return t; }
__device__ __forceinline__ float t_fma(float a, float b, float c) { float t; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c)); return t; }
// the team's barrier: LDS traffic only (__syncthreads() would also wait for the acknowledgement of the output stores)
__device__ __forceinline__ void t_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int... I, class F> __device__ __forceinline__ void t_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// ---- the operand reads and MFMAs of one step of one role, as tables (I = step number mod 4: ring slots are compile-time constants)
struct TRd { int base, imm; };              // base register: 0 x ring, 1 raw-x ring, 2 lrelu(t1) / z rings, 3 t1 rings
struct TMm { int acc, w, rd, first; };
constexpr int T_MAXR = 16, T_MAXM = 28;
#ifndef V2W_N32S_RING
#define V2W_N32S_RING 12
#endif
constexpr int T_RING = V2W_N32S_RING;          // operand reads in flight per wave
template <int ROLE, int I> struct RoleProg {
    int nr, nm;
    TRd rd[T_MAXR]; TMm mm[T_MAXM];
    constexpr RoleProg() : nr(0), nm(0), rd{}, mm{} {
        if (ROLE == 0) {                                   // conv1, k = 11 (weights 0..21 = tap * 2 + half, 22 / 23 identity, 24..26 upsampler half 0)
            for (int t = 0; t < 11; ++t) {
                rd[nr] = TRd{0, T_XOFF + (I + 1) * T_BLK + (t - 5) * 16};
                mm[nm++] = TMm{0, 2 * t, nr, t == 0}; mm[nm++] = TMm{1, 2 * t + 1, nr, t == 0};
                ++nr;
            }
            rd[nr] = TRd{1, I * T_BLK};
            mm[nm++] = TMm{0, 22, nr, 0}; mm[nm++] = TMm{1, 23, nr, 0};
            ++nr;
            up(2, 24);
        } else if (ROLE == 1) {                            // conv1, k = 7 (0..13) and k = 3 (14..19) on the same seven windows; identity 20 / 21
            for (int t = 0; t < 7; ++t) {
                rd[nr] = TRd{0, T_XOFF + (I + 1) * T_BLK + (t - 3) * 16};
                mm[nm++] = TMm{0, 2 * t, nr, t == 0}; mm[nm++] = TMm{1, 2 * t + 1, nr, t == 0};
                if (t >= 2 && t <= 4) { mm[nm++] = TMm{2, 14 + 2 * (t - 2), nr, t == 2}; mm[nm++] = TMm{3, 15 + 2 * (t - 2), nr, t == 2}; }
                ++nr;
            }
            rd[nr] = TRd{1, I * T_BLK};
            mm[nm++] = TMm{0, 20, nr, 0}; mm[nm++] = TMm{1, 21, nr, 0}; mm[nm++] = TMm{2, 20, nr, 0}; mm[nm++] = TMm{3, 21, nr, 0};
            ++nr;
        } else if (ROLE == 2) {                            // conv2, k = 11, of block s - 2: lrelu(t1) ring slot (I + 2) & 3; + t1 itself through the identity
            const int cs = (I + 2) & 3;
            for (int t = 0; t < 11; ++t) {
                rd[nr] = TRd{2, T_AOFF + 2 * 4 * T_AP + (cs + 1) * T_BLK + 3 * (t - 5) * 16};
                mm[nm++] = TMm{0, 2 * t, nr, t == 0}; mm[nm++] = TMm{1, 2 * t + 1, nr, t == 0};
                ++nr;
            }
            rd[nr] = TRd{3, T_QOFF + 2 * 4 * T_QP + cs * T_BLK};
            mm[nm++] = TMm{0, 22, nr, 0}; mm[nm++] = TMm{1, 23, nr, 0};
            ++nr;
        } else {                                           // conv2, k = 7 (0..13) and k = 3 (14..19), ONE pair of accumulators; identity 20 / 21; upsampler half 1: 22..24
            const int cs = (I + 2) & 3;
            for (int t = 0; t < 7; ++t) {
                rd[nr] = TRd{2, T_AOFF + 1 * 4 * T_AP + (cs + 1) * T_BLK + 3 * (t - 3) * 16};
                mm[nm++] = TMm{0, 2 * t, nr, t == 0}; mm[nm++] = TMm{1, 2 * t + 1, nr, t == 0};
                ++nr;
            }
            for (int t = 0; t < 3; ++t) {
                rd[nr] = TRd{2, T_AOFF + (cs + 1) * T_BLK + 3 * (t - 1) * 16};
                mm[nm++] = TMm{0, 14 + 2 * t, nr, 0}; mm[nm++] = TMm{1, 15 + 2 * t, nr, 0};
                ++nr;
            }
            for (int jb = 1; jb >= 0; --jb) {
                rd[nr] = TRd{3, T_QOFF + jb * 4 * T_QP + cs * T_BLK};
                mm[nm++] = TMm{0, 20, nr, 0}; mm[nm++] = TMm{1, 21, nr, 0};
                ++nr;
            }
            up(2, 22);
        }
    }
    constexpr void up(int acc, int w0) {                   // the upsampler's 3 virtual taps on the z blocks around block s - 5 (z ring slot (I + 3) & 3)
        for (int tv = 0; tv < 3; ++tv) {
            rd[nr] = TRd{2, T_ZOFF + (((I + 3) & 3) + 1) * T_BLK + (tv - 1) * 16};
            mm[nm++] = TMm{acc, w0 + tv, nr, tv == 0};
            ++nr;
        }
    }
    constexpr int last_use(int r) const { int l = 0; for (int m = 0; m < nm; ++m) if (mm[m].rd == r) l = m; return l; }
    constexpr int first_use(int r) const { for (int m = 0; m < nm; ++m) if (mm[m].rd == r) return m; return 0; }
};
template <int ROLE, int I> inline constexpr RoleProg<ROLE, I> kRoleProg{};

// One role of a team.  Every role runs the same sequence of barriers: one after the lead-in staging of a run, one after each step.
template <int ROLE>
__device__ __forceinline__ void n32s_role(const N32SArgs& a, unsigned char* const smem_t) {
    constexpr int NW = ROLE == 0 ? 27 : ROLE == 1 ? 22 : ROLE == 2 ? 24 : 25;
    constexpr int NACC = ROLE == 0 ? 3 : ROLE == 1 ? 4 : ROLE == 2 ? 2 : 3;
    constexpr bool UP = ROLE == 0 || ROLE == 3;                                 // this role runs a half of the upsampler: virtual rows 16 UH ..
    constexpr int UH = ROLE == 3 ? 1 : 0;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, kg = lane >> 4;
    const int L = __builtin_amdgcn_readfirstlane(a.L), R = __builtin_amdgcn_readfirstlane(a.R), rpr = __builtin_amdgcn_readfirstlane(a.rpr);
    const float slope = a.slope;

    // ---- this role's weights.  Fragment unit (16-channel chunk ch, tap t) of a (k, 32, 32) layer (v2w_pack_bf16): 2 KiB, lane' = row + 32 h'
    // holds the input channels 16 ch + 8 h' .. + 7 of output channel `row`.  This lane: output channel 16 mh + j, input channels 8 kg .. 8 kg + 7.
    tu32x4 W[NW];
    {
        auto frag = [&](const unsigned char* w, int K, int t, int mh) {
            return *reinterpret_cast<const tu32x4*>(w + (size_t)((kg >> 1) * K + t) * 2048 + (unsigned)(16 * mh + j + 32 * (kg & 1)) * 16u);
        };
        tu32x4 ident[2];                                                        // element e of this lane: input channel 8 kg + e against output channel 16 mh + j
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int wd = 0; wd < 4; ++wd)
                ident[mh][wd] = (8 * kg + 2 * wd == 16 * mh + j ? 0x3f80u : 0u) | (8 * kg + 2 * wd + 1 == 16 * mh + j ? 0x3f800000u : 0u);
        if constexpr (ROLE == 0 || ROLE == 2) {
            const unsigned char* w = ROLE == 0 ? a.w1[2] : a.w2[2];
#pragma unroll
            for (int t = 0; t < 11; ++t) { W[2 * t] = frag(w, 11, t, 0); W[2 * t + 1] = frag(w, 11, t, 1); }
            W[22] = ident[0]; W[23] = ident[1];
        } else {
            const unsigned char* w7 = ROLE == 1 ? a.w1[1] : a.w2[1];
            const unsigned char* w3 = ROLE == 1 ? a.w1[0] : a.w2[0];
#pragma unroll
            for (int t = 0; t < 7; ++t) { W[2 * t] = frag(w7, 7, t, 0); W[2 * t + 1] = frag(w7, 7, t, 1); }
#pragma unroll
            for (int t = 0; t < 3; ++t) { W[14 + 2 * t] = frag(w3, 3, t, 0); W[15 + 2 * t] = frag(w3, 3, t, 1); }
            W[20] = ident[0]; W[21] = ident[1];
        }
        if constexpr (UP) {                                                     // the upsampler as a 3-tap conv over the 32 virtual rows co * 2 + phase (v2w_pack_bf16_convt)
#pragma unroll
            for (int tv = 0; tv < 3; ++tv) W[(ROLE == 0 ? 24 : 22) + tv] = frag(a.up_w, 3, tv, UH);
        }
    }
    // initial values of the accumulators: the biases of this lane's channels 16 mh + 4 kg .. + 3
    f32x4 init[NACC];
    {
        auto bias4 = [&](const float* b, int mh) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = b[16 * mh + 4 * kg + r];
            }
            return v;
        };
        if constexpr (ROLE == 0) { init[0] = bias4(a.bias1[2], 0); init[1] = bias4(a.bias1[2], 1); }
        if constexpr (ROLE == 1) { init[0] = bias4(a.bias1[1], 0); init[1] = bias4(a.bias1[1], 1); init[2] = bias4(a.bias1[0], 0); init[3] = bias4(a.bias1[0], 1); }
        if constexpr (ROLE == 2) { init[0] = bias4(a.bias2[2], 0); init[1] = bias4(a.bias2[2], 1); }
        if constexpr (ROLE == 3) { init[0] = bias4(a.bias2[1], 0) + bias4(a.bias2[0], 0); init[1] = bias4(a.bias2[1], 1) + bias4(a.bias2[0], 1); }
        if constexpr (UP) {                 // virtual rows 16 UH + 4 kg + r = channel 8 UH + 2 kg + (r >> 1), phase r & 1
            const int c0 = 8 * UH + 2 * kg;
            const float u0 = a.up_bias ? a.up_bias[c0] : 0.f, u1 = a.up_bias ? a.up_bias[c0 + 1] : 0.f;
            init[NACC - 1] = f32x4{u0, u0, u1, u1};
        }
    }

    // ---- LDS addresses.  Operand reads: lane (j, kg) takes the 16 bytes of plane kg, row (block row + j + tap offset); immediates carry ring, slot, tap.
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_t);
    unsigned rbase[4];
    rbase[0] = lds0 + (unsigned)(kg * T_XP + j * 16);
    rbase[1] = lds0 + (unsigned)(T_ROFF + kg * T_RP + j * 16);
    rbase[2] = lds0 + (unsigned)(kg * T_AP + j * 16);
    rbase[3] = lds0 + (unsigned)(kg * T_QP + j * 16);
    asm volatile("" : "+v"(rbase[2]), "+v"(rbase[3]));
    // epilogue writes: channels 16 mh + 4 kg .. + 3 of position j = 8 bytes at plane 2 mh + (kg >> 1), row j, half kg & 1
    unsigned char* const wa = smem_t + (kg >> 1) * T_AP + j * 16 + (kg & 1) * 8;      // (+ 2 T_AP for mh = 1)
    unsigned char* const wq = smem_t + (kg >> 1) * T_QP + j * 16 + (kg & 1) * 8;      // (+ 2 T_QP)
    // staging (roles 2 and 3): thread (channel quad cq of 8, position quad cpos of 16) of a 64-position burst; plane cq >> 1, half cq & 1
    const int st_t = (ROLE & 1) * 64 + lane;
    const int cq = st_t & 7, cpos = st_t >> 3, cblk = cpos >> 2;
    unsigned char* const stx = smem_t + T_XOFF + (cq >> 1) * T_XP + (4 * (cpos & 3)) * 16 + (cq & 1) * 8 + T_BLK;
    unsigned char* const str = smem_t + T_ROFF + (cq >> 1) * T_RP + (4 * (cpos & 3)) * 16 + (cq & 1) * 8;

    auto mfma = [](f32x4 c, tu32x4 av, tu32x4 bv) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tb8, av), __builtin_bit_cast(tb8, bv), c, 0, 0, 0);
    };

    tu32x2 pf[4];
    float av[4], sv[4];
    f32x4 osum[2][2];                                                           // role 2: its conv2 sums of the blocks s - 2 (this step) and s - 3 (the step before)
    osum[0][0] = osum[0][1] = osum[1][0] = osum[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Lout = 2 * L;

    for (int run = blockIdx.x; run < a.nruns; run += gridDim.x) {
        const int b = run / rpr, p0 = (run - b * rpr) * R;
        const int nblk = (min(R, L - p0) + 15) >> 4, ngrp = (nblk + 8) >> 2;       // steps -2 .. 4 ngrp - 1 >= nblk + 4 (the upsampler of the last block)
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * 32 * L * 2;
        unsigned char* const ob = reinterpret_cast<unsigned char*>(a.up_out) + (size_t)b * 16 * Lout * 2;
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};                           // roles 0 / 3: partial sums of the upsampler's output, channels 8 UH + 2 kg + {0, 1}
        if constexpr (ROLE >= 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = a.in_a ? a.in_a[b * 32 + 4 * cq + i] : 1.f;
                sv[i] = a.in_a ? a.in_s[b * 32 + 4 * cq + i] : 0.f;
            }
        }
        // burst k = the x blocks 4 k + 1 .. 4 k + 4 of the run: four 8-byte loads per thread of the two staging waves (4 channels x 4 positions)
        auto issue_x = [&](int k) {
            const int pos = p0 + 64 * k + 16 + 4 * cpos;
            const bool ok = pos >= 0 && pos < L;
            unsigned vo = (unsigned)(4 * cq * L + (ok ? pos : 0)) * 2u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[i] = *gptr<const tu32x2>(inb + (size_t)i * L * 2 + vo);
        };
        // (in two halves - position rows e = 0, 1 then 2, 3 of every quad - in two consecutive steps: the staging waves' extra work per barrier
        // interval is halved)
        auto commit_x = [&](int k, int e0, int e1) {
            const int pos = p0 + 64 * k + 16 + 4 * cpos;
            const bool ok = pos >= 0 && pos < L;                                // L % 4 == 0: a position quad is inside or outside as a whole
            const int slot = (4 * (k & 1) + 1 + cblk) & 7;
            unsigned char* const dx = stx + slot * T_BLK;
            unsigned char* const dr = str + slot * T_BLK;
            const int mir = slot == 7 ? -8 * T_BLK : (slot == 0 ? 8 * T_BLK : 0);      // slot 7 also in front of slot 0, slot 0 also behind slot 7
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e < e0 || e >= e1) continue;
                float y[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (e & 1) ? t_hi(pf[i][e >> 1]) : t_lo(pf[i][e >> 1]);
                    y[i] = t_fma(av[i], xv, sv[i]);
                    v[i] = t_max(y[i], t_mul(y[i], slope));
                }
                tu32x2 w = {t_pack2(v[0], v[1]), t_pack2(v[2], v[3])};
                tu32x2 r = {t_pack2(y[0], y[1]), t_pack2(y[2], y[3])};
                if (!ok) { w = tu32x2{0u, 0u}; r = w; }
                *reinterpret_cast<tu32x2*>(dx + e * 16) = w;
                *reinterpret_cast<tu32x2*>(dr + e * 16) = r;
                if (mir != 0) *reinterpret_cast<tu32x2*>(dx + e * 16 + mir) = w;
            }
        };

        auto step = [&](auto i_c, int s, unsigned bx0, unsigned bx1) __attribute__((always_inline)) {
            using IC = decltype(i_c);                                           // (a type: nested generic lambdas name it without a capture)
            constexpr int I = IC::value;
            constexpr int RING = T_RING;
            using Prog = std::integral_constant<const RoleProg<ROLE, IC::value>*, &kRoleProg<ROLE, IC::value>>;
            constexpr int NR = Prog::value->nr, NM = Prog::value->nm;
            const unsigned bt2 = rbase[2], bt3 = rbase[3];
            f32x4 acc[NACC];
            tu32x4 ring[RING];
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto rd = [&ring, &bx0, &bx1, &bt2, &bt3](auto n_c) __attribute__((always_inline)) {
                constexpr int n = decltype(n_c)::value;
                constexpr int bs = Prog::value->rd[n].base, imm = Prog::value->rd[n].imm;
                static_assert(imm >= 0 && imm < 65536, "ds_read offset field");
                if constexpr (bs == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % T_RING]) : "v"(bx0), "n"(imm));
                else if constexpr (bs == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % T_RING]) : "v"(bx1), "n"(imm));
                else if constexpr (bs == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % T_RING]) : "v"(bt2), "n"(imm));
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % T_RING]) : "v"(bt3), "n"(imm));
            };
            t_for(std::make_integer_sequence<int, (NR < RING ? NR : RING)>{}, rd);
            t_for(std::make_integer_sequence<int, NM>{}, [&ring, &acc, &init, &W, &rd, &mfma](auto m_c) __attribute__((always_inline)) {
                constexpr int m = decltype(m_c)::value;
                constexpr TMm q = Prog::value->mm[m];
                constexpr int NR = Prog::value->nr, RING = T_RING;
                if constexpr (Prog::value->first_use(q.rd) == m) {
                    constexpr int left = (NR - 1 - q.rd) < (RING - 1) ? (NR - 1 - q.rd) : (RING - 1);      // LDS reads return in order
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[q.rd % RING]) : "n"(left));
                }
                if constexpr (q.first) acc[q.acc] = mfma(init[q.acc], W[q.w], ring[q.rd % RING]);
                else acc[q.acc] = mfma(acc[q.acc], W[q.w], ring[q.rd % RING]);
                if constexpr (Prog::value->last_use(q.rd) == m && q.rd + RING < NR) rd(std::integral_constant<int, q.rd + RING>{});
            });
            __builtin_amdgcn_sched_barrier(0);

            const int pos1 = p0 + 16 * s + j;                                  // this lane's position in block s
            if constexpr (ROLE < 2) {
                // ---- t1 epilogue of block s: acc = t1_jb (bias, residual and conv), 0 outside the sequence (conv2 zero-pads lrelu(t1)); t1 itself
                // (bf16) into its ring for the branch sum, lrelu(t1) into the operand ring of conv2
                const bool edge1 = p0 + 16 * s < 0 || p0 + 16 * s + 16 > L;
                constexpr int NB = ROLE == 0 ? 1 : 2;
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int jb = ROLE == 0 ? 2 : 1 - q;                       // role 1: accumulators 0 / 1 = k 7 (branch 1), 2 / 3 = k 3 (branch 0)
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        f32x4 t1v = acc[2 * q + mh];
                        if (edge1) {
                            asm volatile("" ::: "memory");
                            if (pos1 < 0 || pos1 >= L) t1v = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        const tu32x2 raw = {t_pack2(t1v[0], t1v[1]), t_pack2(t1v[2], t1v[3])};
#pragma unroll
                        for (int r = 0; r < 4; ++r) t1v[r] = t_max(t1v[r], t_mul(t1v[r], slope));
                        const tu32x2 w = {t_pack2(t1v[0], t1v[1]), t_pack2(t1v[2], t1v[3])};
                        *reinterpret_cast<tu32x2*>(wq + T_QOFF + jb * 4 * T_QP + mh * 2 * T_QP + I * T_BLK) = raw;
                        unsigned char* const d = wa + T_AOFF + jb * 4 * T_AP + mh * 2 * T_AP;
                        *reinterpret_cast<tu32x2*>(d + (I + 1) * T_BLK) = w;
                        if constexpr (I == 0) *reinterpret_cast<tu32x2*>(d + 5 * T_BLK) = w;
                        if constexpr (I == 3) *reinterpret_cast<tu32x2*>(d) = w;
                    }
                }
            } else if constexpr (ROLE == 3) {
                // ---- this wave's part of the branch sum of block s - 2 to wave 2: [lane][half][4] floats
                unsigned char* const e = smem_t + T_EOFF + (I & 1) * 2048 + lane * 32;
                *reinterpret_cast<f32x4*>(e) = acc[0];
                *reinterpret_cast<f32x4*>(e + 16) = acc[1];
            } else if constexpr (ROLE == 2) {
                // ---- role 2: keep the conv2 sums of block s - 2; the branch sum of block s - 3 = last step's sums + wave 3's part of last step;
                // z = lrelu(out / nk, up_slope), 0 outside the sequence (the transposed conv sees the L positions of the sequence only)
                osum[I & 1][0] = acc[0]; osum[I & 1][1] = acc[1];
                const unsigned char* const e = smem_t + T_EOFF + ((I + 1) & 1) * 2048 + lane * 32;
                const int pos3 = pos1 - 48;
                const bool edge3 = p0 + 16 * s - 48 < 0 || p0 + 16 * s - 32 > L;
                const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f, dinv_us = dinv * a.up_slope;
                constexpr int ZS = (I + 1) & 3;
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
                    f32x4 z = osum[(I + 1) & 1][mh] + *reinterpret_cast<const f32x4*>(e + 16 * mh);
                    // lrelu(sum / nk) = max(sum * (1 / nk), sum * (slope / nk)): two multiplies and a max per element (the quotient by
                    // multiplication is within an ulp of the division - below the bf16 rounding of z by 2^16)
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = t_max(t_mul(z[r], dinv), t_mul(z[r], dinv_us));
                    if (edge3) {
                        asm volatile("" ::: "memory");
                        if (pos3 < 0 || pos3 >= L) z = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    const tu32x2 w = {t_pack2(z[0], z[1]), t_pack2(z[2], z[3])};
                    unsigned char* const d = wa + T_ZOFF + mh * 2 * T_AP;
                    *reinterpret_cast<tu32x2*>(d + (ZS + 1) * T_BLK) = w;
                    if constexpr (ZS == 0) *reinterpret_cast<tu32x2*>(d + 5 * T_BLK) = w;
                    if constexpr (ZS == 3) *reinterpret_cast<tu32x2*>(d) = w;
                }
            }
            if constexpr (UP) {
                // ---- the upsampler's output of block s - 5: registers (0, 1) = channel c0, outputs 2 q + {0, 1}; (2, 3) = channel c0 + 1
                const f32x4 u = acc[NACC - 1];
                const int q = pos1 - 80;
                const bool valid = s >= 5 && s - 5 < nblk && q < L;
                const int c0 = 8 * UH + 2 * kg;
                // 8-byte stores: lanes j and j ^ 1 hold the output pairs (2 q, 2 q + 1) of two neighbouring input positions for channels c0 and
                // c0 + 1 - the even lane takes both pairs of channel c0, the odd lane both of c0 + 1 (one DPP swap): per store instruction a
                // channel row receives 128 contiguous bytes instead of 64.  (q and q ^ 1 are valid together: blocks and L are multiples of 4.)
                const unsigned pa = t_pack2(u[0], u[1]), pb = t_pack2(u[2], u[3]);
                const bool odd = (j & 1) != 0;
                const unsigned got = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(odd ? pa : pb), 0xB1, 0xF, 0xF, false);     // quad_perm [1, 0, 3, 2]
                if (valid) {
                    const tu32x2 w2 = odd ? tu32x2{got, pb} : tu32x2{pa, got};
                    *gptr<tu32x2>(ob + (unsigned)((c0 + (odd ? 1 : 0)) * Lout + 2 * (q & ~1)) * 2u) = w2;
                }
                // (a select, not a product: the lead-in steps of a run work on stale rings)
                const float v0 = valid ? u[0] : 0.f, v1 = valid ? u[1] : 0.f, v2 = valid ? u[2] : 0.f, v3 = valid ? u[3] : 0.f;
                s1[0] += v0 + v1; s2[0] = fmaf(v0, v0, fmaf(v1, v1, s2[0]));
                s1[1] += v2 + v3; s2[1] = fmaf(v2, v2, fmaf(v3, v3, s2[1]));
            }
        };

        // ---- the run.  Staging: burst g (blocks 4 g + 1 .. 4 g + 4) is committed in steps 4 g - 2 and 4 g - 1 (half each), one barrier before conv1 first reads it.
        if constexpr (ROLE >= 2) { issue_x(-1); commit_x(-1, 0, 4); issue_x(0); }
        t_barrier();
        {
            unsigned bx0 = rbase[0] + 4u * T_BLK, bx1 = rbase[1] + 4u * T_BLK;
            asm volatile("" : "+v"(bx0), "+v"(bx1));
            step(std::integral_constant<int, 2>{}, -2, bx0, bx1);
            if constexpr (ROLE >= 2) commit_x(0, 0, 2);
            t_barrier();
            step(std::integral_constant<int, 3>{}, -1, bx0, bx1);
            if constexpr (ROLE >= 2) { commit_x(0, 2, 4); issue_x(1); }
            t_barrier();
        }
        for (int g = 0; g < ngrp; ++g) {
            unsigned bx0 = rbase[0] + ((g & 1) ? 4u * T_BLK : 0u), bx1 = rbase[1] + ((g & 1) ? 4u * T_BLK : 0u);
            asm volatile("" : "+v"(bx0), "+v"(bx1));
            step(std::integral_constant<int, 0>{}, 4 * g, bx0, bx1);
            t_barrier();
            step(std::integral_constant<int, 1>{}, 4 * g + 1, bx0, bx1);
            t_barrier();
            step(std::integral_constant<int, 2>{}, 4 * g + 2, bx0, bx1);
            if constexpr (ROLE >= 2) commit_x(g + 1, 0, 2);
            t_barrier();
            step(std::integral_constant<int, 3>{}, 4 * g + 3, bx0, bx1);
            if constexpr (ROLE >= 2) { commit_x(g + 1, 2, 4); issue_x(g + 2); }
            t_barrier();
        }
        // ---- BatchNorm partial sums of this run (one row of up_stats): the 16 lanes of a channel pair in a fixed (butterfly) order
        if constexpr (UP) {
            if (a.up_stats) {
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) { s1[c] += __shfl_xor(s1[c], off, 64); s2[c] += __shfl_xor(s2[c], off, 64); }
                }
                if (j == 0) {
                    float* const d = a.up_stats + ((size_t)run * 16 + 8 * UH + 2 * kg) * 2;
                    *gptr<f32x4>(d) = f32x4{s1[0], s2[0], s1[1], s2[1]};
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2)
n32s_stage_kernel(const N32SArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) n32s_role<0>(a, smem_t);
    else if (wave == 1) n32s_role<1>(a, smem_t);
    else if (wave == 2) n32s_role<2>(a, smem_t);
    else n32s_role<3>(a, smem_t);
}

}  // namespace

// Called by v2w_resblock2_stage_bf16 (v2w_stage_bf16.hip) for the 32-channel stage WITH the next stage's stride-2 upsampler on bf16 tensors.
// V2W_E_SHAPE: not served (the caller runs the resident-tile kernel).  up_tiles_out: rows of up_stats_part this call fills (= runs).
int v2w_resblock2_stage_bf16_n32s(const v2w_stage_split_args* q, hipStream_t stream, int* up_tiles_out) {
    if (q->C != 32 || q->io_bf16 != 3 || !q->bf16 || q->nk != 3 || q->rb1 || q->post_out) return V2W_E_SHAPE;
    if (!q->up_out || q->up_u != 2 || q->up_k != 4 || !q->up_wps) return V2W_E_SHAPE;
    for (int j = 0; j < 3; ++j)
        if (q->k[j] != 3 + 4 * j || q->dil1[j] != 1 || q->dil2[j] != 3 || !q->wps1[j] || !q->wps2[j]) return V2W_E_SHAPE;
    if (!(q->up_slope > 0.f && q->up_slope < 1.f) || !(q->slope > 0.f && q->slope < 1.f)) return V2W_E_SHAPE;     // lrelu as max(v, slope v)
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    if (q->L % 4 != 0 || !al16(q->in) || !al16(q->up_out) || !al16(q->up_stats_part) || (long long)32 * q->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;
    N32SArgs p{};
    p.in = reinterpret_cast<const unsigned short*>(q->in); p.in_a = q->in_a; p.in_s = q->in_s;
    for (int j = 0; j < 3; ++j) {
        p.w1[j] = static_cast<const unsigned char*>(q->wps1[j]); p.bias1[j] = q->bias1[j];
        p.w2[j] = static_cast<const unsigned char*>(q->wps2[j]); p.bias2[j] = q->bias2[j];
    }
    p.up_w = static_cast<const unsigned char*>(q->up_wps); p.up_bias = q->up_bias;
    p.up_out = reinterpret_cast<unsigned short*>(q->up_out); p.up_stats = q->up_stats_part;
    p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div; p.up_slope = q->up_slope;
    const int nteams = v2w_num_cus() * 2;
    // run length: the multiple of 64 positions that minimises (runs per team) x (steps per run); a run costs its blocks + 7 steps of lead-in / drain
    long long best = -1; int bestR = 64;
    for (int R = 64; R <= 4096; R += 64) {
        const long long rpr = (q->L + R - 1) / R, runs = rpr * q->B;
        const long long cost = ((runs + nteams - 1) / nteams) * (R / 16 + 9);
        if (best < 0 || cost < best) { best = cost; bestR = R; }
        if (R >= q->L) break;
    }
    p.R = bestR; p.rpr = (q->L + bestR - 1) / bestR;
    if ((long long)q->B * p.rpr > 0x7fffffffll) return V2W_E_SHAPE;
    p.nruns = q->B * p.rpr;
    if (up_tiles_out) *up_tiles_out = p.nruns;
    if (v2w_dry(stream)) return 0;
    V2W_LAUNCH(n32s_stage_kernel, dim3(p.nruns < nteams ? p.nruns : nteams), dim3(256), T_LDS, stream, p);
    return v2w_launch_status();
}
