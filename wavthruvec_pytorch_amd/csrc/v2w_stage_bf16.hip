// Whole residual section of a narrow ResBlock2 stage (C = 32 / 16) in ONE kernel on the bf16 matrix pipe (bf16 operands, fp32
// accumulate: BASELINE configs[2]); models.py:135-141 with ResBlock2.forward inlined:
//   out = ( sum_j [ t1_j + conv_{k_j,d2_j}(lrelu(t1_j)) + b2_j ] ) / nk ,   t1_j = x + conv_{k_j,d1_j}(lrelu(x)) + b1_j
//
// The structure of the exact-fp32 stage kernel (v2w_resblock_fused.hip) with everything that a 16x faster matrix pipe changes:
//   * the MFMAs of a whole tile are ~5 000 cycles of issue, an L2 round trip is ~600: the weight fragments of a conv (<= 11 taps x
//     2 k-steps x 1 KiB) are requested ALL AT ONCE when its phase starts and live in registers through it - the tap loop holds only
//     ds_read_b128s and MFMAs;
//   * LDS tiles are position-major bf16 rows of the ACTIVATED operands lrelu(x), lrelu(t1) (80 / 48 bytes per position: one
//     conflict-free ds_read_b128 = the 8 k-values a lane feeds to one v_mfma_f32_32x32x16_bf16); conv1 and conv2 run on the same window
//     of 256 positions so the residuals (x, t1: fp32) stay in the lane that produced them;
//   * C = 16 uses the 32-row MFMA with the packed fragments' rows 16-31 zero (the pipe is idle most of the time anyway).
// Weights: the bf16 fragments of v2w_pack_bf16 / v2w_split_pack_batch, [16-channel k-step][tap][2 KiB, first KiB used].
#include <type_traits>
#include "v2w_common.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define V2W_SB_MAXB 4
#define V2W_SB_KMAX 11
#define V2W_SB_UNIT 2048

struct StageBfArgs {
    const float* in; const float* in_a; const float* in_s;
    const unsigned char* w1[V2W_SB_MAXB]; const float* bias1[V2W_SB_MAXB];
    const unsigned char* w2[V2W_SB_MAXB]; const float* bias2[V2W_SB_MAXB];
    int K[V2W_SB_MAXB], d1[V2W_SB_MAXB], d2[V2W_SB_MAXB];
    float* out;
    int nk, B, L;
    int h1max, h2max;
    int xoff, xrows, nto, ntl;
    int vec4;
    float slope, out_div;
};

__device__ __forceinline__ unsigned int sb_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}

__device__ __forceinline__ float sb_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float sb_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// IO_BF: `in` and `out` are bf16 tensors (activation storage of BASELINE configs[2]); arithmetic stays fp32
template <int C, bool IO_BF>
__global__ void __launch_bounds__(256, C == 32 ? 2 : 3)
stage_bf16_kernel(const StageBfArgs p) {
    typedef Frag<32> F;
    typedef F::acc_t acc_t;
    constexpr int NTHREADS = 256, NI = 2, WN = 4, W = 32 * NI * WN;
    constexpr int ROWB = C == 32 ? 80 : 48;      // bytes per position: C bf16 + 16 B pad
    constexpr int KS = C / 16;                   // k-steps per tap
    constexpr int NR = C == 32 ? 16 : 8;         // accumulator registers that hold real output channels (rows < C)
    constexpr int NCQ = C / 4;                   // channel quads
    constexpr int KMAX = V2W_SB_KMAX;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tile = blockIdx.x;
    const int b = tile / p.ntl;
    const int n0 = (tile % p.ntl) * p.nto;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, hk = lane >> 5;
    const int wn0 = wave * (32 * NI);
    const int L = p.L;
    const float slope = p.slope;
    unsigned char* const Xa = smem_b;                                  // [xrows][ROWB]  lrelu(x) as bf16, exactly 0 outside the sequence
    unsigned char* const Ta = smem_b + p.xrows * ROWB;                 // [W][ROWB]      lrelu(t1_j); conv2's taps reach h2max rows past either end
    float* const etab = reinterpret_cast<float*>(Ta + (W + p.h2max) * ROWB);   // bias1[nk][C], bias2[nk][C], then in_a[C], in_s[C] of this batch item
    float* const atab = etab + 2 * V2W_SB_MAXB * C;
    unsigned char* const Wl = reinterpret_cast<unsigned char*>(atab + 2 * C);   // [K * KS][1 KiB]: the running conv's weight fragments, in the order it consumes them
    const int pos0 = n0 - p.h2max - p.h1max - p.xoff;                  // position of X row 0 (multiple of 4)
    const int xc0 = p.xoff + p.h1max;                                  // X row of window column 0 (position n0 - h2max)

    V2W_STAMP(0);
    auto fill_tables = [&]() {                 // (called AFTER the signal loads are out: a load -> LDS store round trip of its own)
        for (int i = tid; i < p.nk * C; i += NTHREADS) {
            const int j = i / C, c = i - j * C;
            etab[i] = p.bias1[j] ? p.bias1[j][c] : 0.f;
            etab[V2W_SB_MAXB * C + i] = p.bias2[j] ? p.bias2[j][c] : 0.f;
        }
        for (int c = tid; c < C; c += NTHREADS) {
            atab[c] = p.in_a ? p.in_a[b * C + c] : 1.f;
            atab[C + c] = p.in_s ? p.in_s[b * C + c] : 0.f;
        }
    };

    // ---- stage lrelu(a*x + s) as bf16: a thread takes 4 channels x 4 positions (8 bytes per position), the channel quads of one
    // position quad on consecutive lanes (whole rows per 8 / 4 lanes: conflict-free stores)
    if (p.vec4) {
        constexpr int NPF = (NCQ * ((W + 2 * 32 + 8) / 4) + NTHREADS - 1) / NTHREADS;
        const int xr4 = p.xrows >> 2;
        typedef typename std::conditional<IO_BF, u32x2, f32x4>::type ld_t;
        ld_t g[NPF][4];
        // (NTHREADS is a multiple of NCQ: a thread keeps its channel quad for every item) affine of its 4 channels, loaded once
        const int cq = tid % NCQ;
        float av[4], sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = p.in_a ? p.in_a[b * C + 4 * cq + i] : 1.f;
            sv[i] = p.in_s ? p.in_s[b * C + 4 * cq + i] : 0.f;
        }
        constexpr int ESI = IO_BF ? 2 : 4;
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(p.in) + (size_t)(b * C) * L * ESI;     // (uniform)
#pragma unroll
        for (int s = 0; s < NPF; ++s) {                         // unconditional loads: outside the sequence position 0, zeroed below
            const int pq = (tid + s * NTHREADS) / NCQ;
            const int pos = pos0 + pq * 4;
            const bool ok = pq < xr4 && pos >= 0 && pos < L;
            unsigned vo = (unsigned)(4 * cq * L + (ok ? pos : 0)) * ESI;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int i = 0; i < 4; ++i) g[s][i] = *reinterpret_cast<const ld_t*>(inb + (size_t)i * L * ESI + vo);
        }
        __builtin_amdgcn_sched_barrier(0);
        fill_tables();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int pq = (tid + s * NTHREADS) / NCQ;
            if (pq >= xr4) continue;
            const bool ok = pos0 + pq * 4 >= 0 && pos0 + pq * 4 < L;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float xv;
                    if constexpr (IO_BF) xv = (e & 1) ? sb_hi(g[s][i][e >> 1]) : sb_lo(g[s][i][e >> 1]);
                    else xv = g[s][i][e];
                    a[i] = v2w_lrelu(fmaf(av[i], xv, sv[i]), slope);
                }
                u32x2 v = {sb_pack2(a[0], a[1]), sb_pack2(a[2], a[3])};
                if (!ok) v = u32x2{0u, 0u};
                *reinterpret_cast<u32x2*>(Xa + (pq * 4 + e) * ROWB + cq * 8) = v;
                if constexpr (IO_BF) {
                    // the stored (bf16) input itself, position-major, for the window's 256 columns: parked in the T1 tile (unused until
                    // conv1 of branch 0 is done), where the residual x of every lane's outputs is read back with 8-byte LDS loads
                    const int col = pq * 4 + e - xc0;
                    if (col >= 0 && col < W) {
                        unsigned w0, w1;
                        if (e & 1) {
                            w0 = __builtin_amdgcn_perm(g[s][1][e >> 1], g[s][0][e >> 1], 0x07060302u);
                            w1 = __builtin_amdgcn_perm(g[s][3][e >> 1], g[s][2][e >> 1], 0x07060302u);
                        } else {
                            w0 = __builtin_amdgcn_perm(g[s][1][e >> 1], g[s][0][e >> 1], 0x05040100u);
                            w1 = __builtin_amdgcn_perm(g[s][3][e >> 1], g[s][2][e >> 1], 0x05040100u);
                        }
                        *reinterpret_cast<u32x2*>(Ta + col * ROWB + cq * 8) = u32x2{w0, w1};
                    }
                }
            }
        }
    } else {
        fill_tables();
        for (int i = tid; i < C * p.xrows; i += NTHREADS) {
            const int c = i / p.xrows, r = i - c * p.xrows, pos = pos0 + r;
            float v = 0.f;
            if (pos >= 0 && pos < L) {
                const int ch = b * C + c;
                const float xv = IO_BF ? sb_lo(reinterpret_cast<const unsigned short*>(p.in)[(size_t)ch * L + pos]) : p.in[(size_t)ch * L + pos];
                v = v2w_lrelu(fmaf(p.in_a ? p.in_a[ch] : 1.f, xv, p.in_s ? p.in_s[ch] : 0.f), slope);
            }
            reinterpret_cast<__bf16*>(Xa + r * ROWB)[c] = (__bf16)v;
        }
    }

    V2W_STAMP(1);
    // ---- the residual of conv1 in every branch: raw x = a*in + s at this lane's outputs (accumulator row e <-> channel F::row(e, hk),
    // its two columns).  bf16 storage: read back from the raw tile after the first barrier (below); otherwise re-read from global
    // memory, one element per load (L2-resident: this workgroup has just fetched the lines).
    const bool raw_tile = IO_BF && p.vec4;
    float xres[NI][NR];
    if (!raw_tile) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int pos = n0 - p.h2max + wn0 + j * 32 + lr;
            const bool in_seq = pos >= 0 && pos < L;
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int ch = b * C + F::row(e, hk);
                float v = 0.f;
                if (in_seq) {
                    const float xv = IO_BF ? sb_lo(reinterpret_cast<const unsigned short*>(p.in)[(size_t)ch * L + pos]) : p.in[(size_t)ch * L + pos];
                    v = fmaf(p.in_a ? p.in_a[ch] : 1.f, xv, p.in_s ? p.in_s[ch] : 0.f);
                }
                xres[j][e] = v;
            }
        }
    }

    // ---- weights.  Every wave of the workgroup - and every workgroup of the launch - needs the same fragments: fetched per wave into
    // registers (336 KB per tile against 28 KB of signal; 7 TB/s of L2 traffic over the launch) each conv phase began with an L2 round
    // trip of ~5 k cycles.  Now each wave fetches a QUARTER of the next conv's fragments into registers while the current conv runs,
    // the workgroup assembles them in LDS between two convs (in consumption order: tap-major) and the MFMA loop reads its A operand
    // with ds_read_b128 at immediate offsets, two k-steps ahead like the B operand.
    acc_t acc[NI];
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr int NWL = (KMAX * KS + 3) / 4;                 // fragments a wave fetches per conv
    u32x4 wl[NWL];
    auto wload = [&](const unsigned char* wbase, int K) {    // unconditional loads (past the end: the last fragment again)
        unsigned l16 = lane16;
        asm volatile("" : "+v"(l16));
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int v = wave + 4 * i;                      // consumption-order index: tap v / KS, k-step v % KS
            const int vc = v < K * KS ? v : K * KS - 1;
            wl[i] = *reinterpret_cast<const u32x4*>(wbase + (size_t)((vc % KS) * K + vc / KS) * V2W_SB_UNIT + l16);
        }
    };
    auto wstore = [&](int K) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int v = wave + 4 * i;
            if (v < K * KS) *reinterpret_cast<u32x4*>(Wl + v * 1024 + lane16) = wl[i];
        }
    };
    // `x0`: this lane's 16 bytes in the row of (its column, tap 0), k-step 0.  The tap loop is unrolled to KMAX with uniform guards.
    auto conv = [&](int K, const unsigned char* x0, int step, const float* bias) {
        // accumulators start at the bias (rows >= C of the 32-row MFMA are padding: they start, and stay, at 0)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float bv = e < NR ? bias[F::row(e, hk)] : 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[j][e] = bv;
        }
        // k-step q = t * KS + s reads rows x0 + t * step at byte 32 s and fragment q; bb[q & 1] / wa[q & 1] hold them, refilled with
        // k-step q + 2 after its MFMAs (an LDS read issued one k-step = NI MFMAs = 64 cycles ahead is not back when it is needed)
        auto rows = [&](int q) {
            const int t = q / KS, sq = q % KS;
            const int tc = t < K ? t : K - 1;                 // past the end: a valid (unused) address
            return x0 + tc * step + 32 * sq;
        };
        const unsigned char* const wq = Wl + lane16;
        u32x4 bb[2][NI], wa[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wa[q] = *reinterpret_cast<const u32x4*>(wq + q * 1024);
#pragma unroll
            for (int j = 0; j < NI; ++j) bb[q][j] = *reinterpret_cast<const u32x4*>(rows(q) + j * 32 * ROWB);
        }
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t < K) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int q = t * KS + s;
                    const unsigned char* nxt = rows(q + 2);
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, wa[q & 1]), __builtin_bit_cast(b8, bb[q & 1][j]), acc[j], 0, 0, 0);
                        bb[q & 1][j] = *reinterpret_cast<const u32x4*>(nxt + j * 32 * ROWB);
                    }
                    wa[q & 1] = *reinterpret_cast<const u32x4*>(wq + (q + 2 < KMAX * KS ? q + 2 : q) * 1024);   // (stale past K * KS: unused)
                }
            }
        }
    };
    wload(p.w1[0], p.K[0]);

    float t1r[NI][NR], oacc[NI][NR];
    const unsigned char* const xl = Xa + (wn0 + lr) * ROWB + 16 * hk;
    const unsigned char* const tl = Ta + (wn0 + lr) * ROWB + 16 * hk;
    V2W_STAMP(2);
    wstore(p.K[0]);
    __syncthreads();
    V2W_STAMP(3);
    if (raw_tile) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int pos = n0 - p.h2max + wn0 + j * 32 + lr;
            const bool in_seq = pos >= 0 && pos < L;
            const unsigned char* row = Ta + (wn0 + j * 32 + lr) * ROWB;
#pragma unroll
            for (int g4 = 0; g4 < NR / 4; ++g4) {   // accumulator registers 4g .. 4g + 3 = channels 8g + 4hk + {0..3}: 8 contiguous bytes
                const u32x2 raw = *reinterpret_cast<const u32x2*>(row + 2 * (8 * g4 + 4 * hk));
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(atab + 8 * g4 + 4 * hk);
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(atab + C + 8 * g4 + 4 * hk);
                xres[j][4 * g4 + 0] = in_seq ? fmaf(a4[0], sb_lo(raw[0]), s4[0]) : 0.f;
                xres[j][4 * g4 + 1] = in_seq ? fmaf(a4[1], sb_hi(raw[0]), s4[1]) : 0.f;
                xres[j][4 * g4 + 2] = in_seq ? fmaf(a4[2], sb_lo(raw[1]), s4[2]) : 0.f;
                xres[j][4 * g4 + 3] = in_seq ? fmaf(a4[3], sb_hi(raw[1]), s4[3]) : 0.f;
            }
        }
    }
    for (int jb = 0; jb < p.nk; ++jb) {
        const int K = p.K[jb], d1 = p.d1[jb], d2 = p.d2[jb];
        const int h1 = d1 * (K - 1) / 2, h2 = d2 * (K - 1) / 2;
        const bool more = jb + 1 < p.nk;

        // ---- conv1_j on the window: column col <-> position n0 - h2max + col
        wload(p.w2[jb], K);                                  // conv2_j's fragments travel while conv1_j computes
        conv(K, xl + (xc0 - h1) * ROWB, d1 * ROWB, etab + jb * C);
        V2W_STAMP(4 + 5 * jb);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int pos = n0 - p.h2max + wn0 + j * 32 + lr;
            const bool in_seq = pos >= 0 && pos < L;            // conv2 zero-pads t1 outside the sequence
#pragma unroll
            for (int e = 0; e < NR; ++e) t1r[j][e] = in_seq ? acc[j][e] + xres[j][e] : 0.f;
        }
        V2W_STAMP(5 + 5 * jb);
        __syncthreads();          // conv2 of the previous branch has finished reading T1 (branch 0: every wave has read its residuals)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned char* row = Ta + (wn0 + j * 32 + lr) * ROWB;
#pragma unroll
            for (int g4 = 0; g4 < NR / 4; ++g4) {   // accumulator registers 4g .. 4g + 3 = channels 8g + 4hk + {0..3}: 8 contiguous bytes
                float a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = v2w_lrelu(t1r[j][4 * g4 + r], slope);
                *reinterpret_cast<u32x2*>(row + 2 * (8 * g4 + 4 * hk)) = u32x2{sb_pack2(a[0], a[1]), sb_pack2(a[2], a[3])};
            }
        }
        wstore(K);                                           // (every wave is past conv1_j: the barrier above)
        V2W_STAMP(6 + 5 * jb);
        __syncthreads();
        V2W_STAMP(7 + 5 * jb);

        // ---- conv2_j on the same window ; r_j = (acc + b2) + t1_j ; branch sum in the reference's order
        if (more) wload(p.w1[jb + 1], p.K[jb + 1]);
        conv(K, tl - h2 * ROWB, d2 * ROWB, etab + V2W_SB_MAXB * C + jb * C);
        V2W_STAMP(8 + 5 * jb);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const float r = acc[j][e] + t1r[j][e];
                oacc[j][e] = jb == 0 ? r : oacc[j][e] + r;
            }
        if (more) {                                          // the next branch's conv1 fragments: every wave is done with this conv's
            __syncthreads();
            wstore(p.K[jb + 1]);
            __syncthreads();
        }
    }

    // ---- store through an aligned fp32 LDS scratch [C][W + 4] (both tiles are dead), float4s along positions
    V2W_STAMP(20);
    __syncthreads();
    {
        constexpr int SRS = W + 4;
        float* const scr = reinterpret_cast<float*>(smem_b);
        const int soff = (p.h2max + 3) & ~3;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int sc = wn0 + j * 32 + lr - p.h2max + soff;
#pragma unroll
            for (int e = 0; e < NR; ++e) scr[F::row(e, hk) * SRS + sc] = oacc[j][e];
        }
        __syncthreads();
        V2W_STAMP(21);
        const float dinv = p.out_div != 0.f ? 1.f / p.out_div : 1.f;
        const int nq = p.nto >> 2;
        const unsigned magic = (unsigned)(((1ull << 32) + nq - 1) / nq);
        for (int idx = tid; idx < C * nq; idx += NTHREADS) {
            const int row = (int)__umulhi((unsigned)idx, magic), q = idx - row * nq;
            const int pos = n0 + 4 * q;
            if (pos >= L) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * SRS + soff + 4 * q);
            if (p.out_div != 0.f) {
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], p.out_div, dinv);
            }
            const size_t doff = ((size_t)b * C + row) * L + pos;
            if constexpr (IO_BF) {
                unsigned short* dst = reinterpret_cast<unsigned short*>(p.out) + doff;
                if (p.vec4) {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{sb_pack2(v[0], v[1]), sb_pack2(v[2], v[3])};
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (pos + x < L) reinterpret_cast<__bf16*>(dst)[x] = (__bf16)v[x];
                }
            } else {
                float* dst = p.out + doff;
                if (p.vec4) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (pos + x < L) dst[x] = v[x];
                }
            }
        }
    }
    V2W_STAMP(22);
}

template <int C>
int launch_stage_bf16(const v2w_stage_split_args* q, hipStream_t stream) {
    constexpr int W = 256, ROWB = C == 32 ? 80 : 48;
    StageBfArgs p{};
    p.in = q->in; p.in_a = q->in_a; p.in_s = q->in_s; p.out = q->out;
    p.nk = q->nk; p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div;
    for (int j = 0; j < q->nk; ++j) {
        p.w1[j] = static_cast<const unsigned char*>(q->wps1[j]); p.bias1[j] = q->bias1[j];
        p.w2[j] = static_cast<const unsigned char*>(q->wps2[j]); p.bias2[j] = q->bias2[j];
        p.K[j] = q->k[j]; p.d1[j] = q->dil1[j]; p.d2[j] = q->dil2[j];
        if (q->k[j] > V2W_SB_KMAX) return V2W_E_SHAPE;
        const int h1 = q->dil1[j] * (q->k[j] - 1) / 2, h2 = q->dil2[j] * (q->k[j] - 1) / 2;
        if (h1 > p.h1max) p.h1max = h1;
        if (h2 > p.h2max) p.h2max = h2;
    }
    p.nto = (W - 2 * p.h2max) & ~3;
    if (p.nto < W / 2 || p.h1max > 32) return V2W_E_SHAPE;
    const int hsum = p.h1max + p.h2max;
    p.xoff = ((hsum + 3) & ~3) - hsum;
    p.xrows = (p.xoff + W + 2 * p.h1max + 3) & ~3;
    if (p.xrows < p.h2max) return V2W_E_SHAPE;
    p.ntl = (q->L + p.nto - 1) / p.nto;
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    p.vec4 = (q->L % 4 == 0) && al16(q->in) && al16(q->out);
    int kmax = 0;
    for (int j = 0; j < q->nk; ++j) if (q->k[j] > kmax) kmax = q->k[j];
    size_t lds = (size_t)(p.xrows + W + p.h2max) * ROWB + (2 * V2W_SB_MAXB + 2) * C * sizeof(float) + (size_t)kmax * (C / 16) * 1024;
    const size_t scr = (size_t)C * (W + 4) * sizeof(float);           // the store scratch overlays the tiles
    if (lds < scr) lds = scr;
    if (lds > 160 * 1024) return V2W_E_SHAPE;
    if (q->io_bf16 != 0 && q->io_bf16 != 3) return V2W_E_ARG;
    if (v2w_dry(stream)) return 0;
    auto kern = q->io_bf16 ? stage_bf16_kernel<C, true> : stage_bf16_kernel<C, false>;
    if (lds > 64 * 1024) {
        hipError_t e = v2w_max_lds(reinterpret_cast<const void*>(kern), (int)lds, stream);
        if (e != hipSuccess) return (int)e;
    }
    V2W_LAUNCH(kern, dim3(q->B * p.ntl), dim3(256), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_stage_bf16)
#endif

int v2w_resblock2_stage_bf16_wide(const v2w_stage_split_args* a, hipStream_t stream, int* up_tiles_out = nullptr);   // v2w_stage_bf16_wide.hip
int v2w_resblock2_stage_bf16_n16(const v2w_stage_split_args* a, hipStream_t stream);    // v2w_stage_bf16_n16.hip
int v2w_resblock1_pairs_bf16_n16(const v2w_stage_split_args* a, hipStream_t stream);    // v2w_stage_bf16_n16.hip
int v2w_resblock2_stage_bf16_n32s(const v2w_stage_split_args* a, hipStream_t stream, int* up_tiles_out);   // v2w_stage_bf16_n32s.hip

// Called by v2w_resblock2_stage_split_fwd when a->bf16 is set.  V2W_E_SHAPE: the caller falls back to the split stage kernel.
int v2w_resblock2_stage_bf16(const v2w_stage_split_args* a, hipStream_t stream, int* up_tiles_out) {
    if (a->up_out) {        // the stage + the next upsampler in one kernel: the resident-tile template only (C = 32 .. 256 on bf16 tensors)
        if (a->io_bf16 != 3 || a->C < 32) return V2W_E_SHAPE;
        if (a->C == 32) {      // 32 channels + the stride-2 upsampler: the streaming kernel of four-wave teams
            const int rc = v2w_resblock2_stage_bf16_n32s(a, stream, up_tiles_out);
            if (rc != V2W_E_SHAPE) return rc;
        }
        return v2w_resblock2_stage_bf16_wide(a, stream, up_tiles_out);
    }
    if (a->rb1) {           // ResBlock1 pair mode: 16 channels on the weights-in-registers kernel, else the resident-tile template's run-time form
        if (a->io_bf16 != 3) return V2W_E_SHAPE;
        if (a->C == 16) {
            const int rc = v2w_resblock1_pairs_bf16_n16(a, stream);
            if (rc != V2W_E_SHAPE) return rc;
        }
        return v2w_resblock2_stage_bf16_wide(a, stream);
    }
    if (a->C >= 64) return v2w_resblock2_stage_bf16_wide(a, stream);
    if (a->C == 16 && a->io_bf16 == 3) {      // the reference's block set on aligned bf16 tensors: weights in registers (+ the 7-tap tail)
        const int rc = v2w_resblock2_stage_bf16_n16(a, stream);
        if (rc != V2W_E_SHAPE) return rc;
    }
    // C = 32: the resident-tile family (two-wave workgroups, 729 us against 838 at configs[2]).  C = 16: only for the fused tail - as a plain
    // stage its one-k-step-per-tap form of that kernel measured 912 us against the 791 us of stage_bf16_kernel<16> below
    const bool c16_wide = a->post_out != nullptr;
    if ((a->C == 32 || (a->C == 16 && c16_wide)) && a->io_bf16 == 3) {
        const int rc = v2w_resblock2_stage_bf16_wide(a, stream);
        if (rc != V2W_E_SHAPE) return rc;
    }
    if (a->post_out) return V2W_E_SHAPE;
    if (a->C == 32) return launch_stage_bf16<32>(a, stream);
    if (a->C == 16) return launch_stage_bf16<16>(a, stream);
    return V2W_E_SHAPE;
}
