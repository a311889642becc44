// EXPERIMENT, not part of the library (round 3; moved out of csrc/ in round 4: nothing in the default build reached it).  To try it again: copy it
// back to wavthruvec_pytorch_amd/csrc/, add it to build.SOURCES and call v2w_resblock2_stage_bf16_n32 from v2w_resblock2_stage_bf16 for C == 32.
// The 32-channel ResBlock2 stage of the generator on bf16 tensors (reference: vec2wav/models.py:135-141 with the reference's block set,
// kernel sizes (3, 7, 11) x dilations (1, 3); anything else runs on v2w_stage_bf16_wide.hip):
//   out = ( sum_j [ t1_j + conv_{k_j, 3}(lrelu(t1_j)) + b2_j ] ) / 3,   t1_j = x + conv_{k_j, 1}(lrelu(x)) + b1_j,   x = a * in + s.
//
// WAVE SPECIALISATION.  On the resident-tile template (two-wave workgroups, four per CU) a 224-column tile of this stage took 74 k cycles of
// which 20 k were the six conv loops: the rest - staging, three t1 epilogues, the store, eight barriers - are vector-ALU / LDS phases during
// which the matrix pipe idles, and every wave of a workgroup is in them at the same time.  Here one 8-wave workgroup per CU runs TWO ROLES,
// one wave of each per SIMD:
//   * conv1 waves (role 0): stage x, run conv1_j and its epilogue (t1_j = acc + x; the t1 tile takes lrelu(t1_j) as bf16; the sum of the
//     three t1_j stays in fp32 registers and is handed over through an fp32 LDS array at the end of the tile);
//   * conv2 waves (role 1): run conv2_j on the t1 tiles one slot later, add the handed-over sum, store the output.
// The two roles meet at four barriers per tile (slots below); within a slot the vector-ALU work of one role runs beside the MFMAs of the other
// on the same SIMD.  As in v2w_stage_bf16_n16.hip the weights live in REGISTERS for the whole (persistent) kernel - a role needs one conv
// set only: 21 taps x one 16-byte operand per lane = 84 registers (v_mfma_f32_16x16x32_bf16: M = 16 of the 32 output channels, K = the 32
// input channels of one tap) - so the conv loops issue ds_read_b128 + MFMA and nothing else.
//
// Slots of iteration i (conv1 waves work on the workgroup's i-th tile, conv2 waves on its (i-1)-th; T[a], T[b]: the two t1 tiles):
//   slot 0   conv1: commit x(i) (loads issued in slot 3 of i-1)   | conv2: conv2_2(i-1) on T[a]
//   slot 1   conv1: conv1_0, epilogue -> T[a]                     | conv2: oacc += handed-over sum(i-1), back in place (fp32 array SF)
//   slot 2   conv1: conv1_1, epilogue -> T[b]                     | conv2: store out(i-1) from SF; conv2_0(i) on T[a]
//   slot 3   conv1: conv1_2, epilogue -> T[a]; sum -> SF; loads of x(i+1)   | conv2: conv2_1(i) on T[b]
#include <type_traits>
#include <utility>
#include "v2w_tile.h"

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct N32Args {
    const unsigned short* in; const float* in_a; const float* in_s;
    const unsigned char* w1[3]; const float* bias1[3];
    const unsigned char* w2[3]; const float* bias2[3];
    unsigned short* out;
    int B, L, nto, ntl, ntiles;
    float slope, out_div;
};

__device__ __forceinline__ unsigned int n32_pack2(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float n32_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float n32_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// Workgroup barrier for LDS hand-overs only: __syncthreads() also fences global memory - a wave that has just issued its tile's output
// stores would wait vmcnt(0) (the stores' acknowledgement, thousands of cycles under load) before it may even arrive at the barrier.
__device__ __forceinline__ void n32_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int... I, class F> __device__ __forceinline__ void n32_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

constexpr int N32_H1 = 5, N32_H2 = 15;           // halos of the widest branch: 11 taps at dilation 1 / 3
constexpr int N32_W = 384;                       // window columns per tile (positions n0 - 15 ..): 352 valid outputs
constexpr int N32_NB = 12;                       // 16-column blocks per wave (192 columns: half the window)
constexpr int N32_XR = N32_W + 12;               // rows of the x / r tiles (positions n0 - 20 ..)
constexpr int N32_SRS = N32_W + 12;              // row stride of the fp32 hand-over array (4 SRS = 16 mod 32 banks)

__global__ void __launch_bounds__(512, 2)
n32_stage_kernel(const N32Args a) {
    constexpr int W = N32_W, XR = N32_XR, RB = 64, NB = N32_NB, SRS = N32_SRS;
    // LDS: x tile | r tile | T[a] | T[b] | 16 rows of slack (conv2's taps past the end of T[b]) | SF
    constexpr unsigned XB = 0, RT = XR * RB, TA = 2 * XR * RB, TBb = TA + W * RB, SFB = TBb + W * RB + 16 * RB;
    constexpr int NIT = 8 * (XR / 4), NPF = (NIT + 255) / 256;                  // staging items (4 channels x 4 positions) per conv1 thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
    float* const SF = reinterpret_cast<float*>(smem_q + SFB);                   // [32][SRS]: column = window column + 1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;                                                 // 0: conv1 waves, 1: conv2 waves (waves w and w + 4 share a SIMD)
    const int rb = wave & 1, chalf = (wave >> 1) & 1;                           // output channels 16 rb .., window columns 192 chalf ..
    const int tr = tid & 255;                                                   // thread index inside the role
    const int j = lane & 15, kg = lane >> 4;
    const int L = __builtin_amdgcn_readfirstlane(a.L), nto = __builtin_amdgcn_readfirstlane(a.nto);
    const float slope = a.slope;

    // ---- this role's weights into registers: tap t of branch jb (v2w_pack_bf16: k-step c16 of tap t at ((c16 K + t) 2 KiB), lane' =
    // row + 32 h holds input channels 16 c16 + 8 h .. + 7 of output channel `row`); this lane: output channel 16 rb + j, input channels 8 kg ..
    u32x4 wa[21];
    {
        const unsigned lo16 = (unsigned)(16 * rb + j + 32 * (kg & 1)) * 16u;
        auto load_set = [&](int jb, int K, int p0) {
            const unsigned char* w = role ? a.w2[jb] : a.w1[jb];
#pragma unroll
            for (int t = 0; t < 11; ++t) {
                if (t >= K) break;
                wa[p0 + t] = *reinterpret_cast<const u32x4*>(w + (size_t)((kg >> 1) * K + t) * 2048 + lo16);
            }
        };
        load_set(0, 3, 0); load_set(1, 7, 3); load_set(2, 11, 10);
    }
    // biases of this lane's 4 channels (16 rb + 4 kg ..): conv1 waves b1_j, conv2 waves the sum of the b2_j
    float bb[3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s2 = 0.f;
#pragma unroll
        for (int jb = 0; jb < 3; ++jb) {
            const float* bp = role ? a.bias2[jb] : a.bias1[jb];
            const float v = bp ? bp[16 * rb + 4 * kg + r] : 0.f;
            bb[jb][r] = v;
            s2 += v;
        }
        if (role) bb[0][r] = s2;
    }

    auto mfma = [&](f32x4 c, u32x4 av, u32x4 bv) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, av), __builtin_bit_cast(b8, bv), c, 0, 0, 0);
    };
    const int col0 = 192 * chalf + j;                                           // this lane's column in block 0 of its wave
    // 64-byte rows (32 channels), 16-byte slots XOR-swizzled by (row >> 1) & 3: with lane (j, kg) on row r0 + j, slot kg, the 16-lane groups
    // of a ds_read_b128 fall on 16 different slots of the 256-byte bank row at every r0 (the (row >> 2) & 3 of the 32-row MFMA layouts is
    // two-way conflicted here); the block offset 16 cb rows leaves the swizzle alone, so a tap costs one address and NB immediates
    auto rowaddr = [&](unsigned base, int row, int slot) { return base + (unsigned)(row * RB + ((slot ^ ((row >> 1) & 3)) << 4)); };
    // The loop is ONE sequence of K * NB (tap, block) steps with a ring of RING operands in flight, written in inline assembly: left to itself
    // hipcc (at the register limit) issued one ds_read_b128, waited lgkmcnt(0), issued its MFMA - an LDS round trip (~60 cycles) per 16-cycle
    // MFMA - and it sinks plain C++ loads back to their uses.  `asm volatile` statements keep their order; the s_waitcnt names the operand
    // it guards so that the MFMA cannot move above it.  LDS reads return in order: before step n at most min(RING - 1, N - 1 - n) younger
    // reads may be outstanding.  (No scalar loads inside: sched_barrier on both sides.)
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem_q);
    auto conv = [&](auto k_c, auto d_c, auto p_c, f32x4 (&acc)[NB], unsigned base, int r0) {
        constexpr int K = decltype(k_c)::value, DIL = decltype(d_c)::value, P0 = decltype(p_c)::value;
        constexpr int N = K * NB, RING = 8;
        int rr = r0;
        asm volatile("" : "+v"(rr));
        unsigned ab[K];
#pragma unroll
        for (int t = 0; t < K; ++t) ab[t] = lds0 + rowaddr(base, rr + t * DIL, kg);
        u32x4 ring[RING];
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        n32_for(std::make_integer_sequence<int, RING>{}, [&ring, &ab](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n]) : "v"(ab[n / NB]), "n"((n % NB) * (16 * RB)));
        });
        n32_for(std::make_integer_sequence<int, N>{}, [&ring, &ab, &acc, &wa, &mfma](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            constexpr int left = (N - 1 - n) < (RING - 1) ? (N - 1 - n) : (RING - 1);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[n % RING]) : "n"(left));
            acc[n % NB] = mfma(acc[n % NB], wa[P0 + n / NB], ring[n % RING]);
            if constexpr (n + RING < N)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[n % RING]) : "v"(ab[(n + RING) / NB]), "n"(((n + RING) % NB) * (16 * RB)));
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- staging by the conv1 waves, in two halves (issue: global loads; commit: activation, bf16 rows of the x and r tiles)
    u32x2 pf[NPF][4];
    float av[4], sv[4];
    const int cq = tr & 7;                                                      // (256 % 8 == 0: a thread keeps its channel quad)
    auto tile_of = [&](int i, int& b, int& n0) {
        const int tile = blockIdx.x + i * gridDim.x;
        b = tile / a.ntl; n0 = (tile - b * a.ntl) * nto;
    };
    auto issue_x = [&](int i) {
        int b, n0;
        tile_of(i, b, n0);
        const int pos0 = n0 - N32_H1 - N32_H2;
        const unsigned char* const inb = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * 32 * L * 2;
        int trv = tr;                                                           // (opaque per call: hoisted out of the tile loop the per-item
        asm volatile("" : "+v"(trv));                                           // offsets are spilled, and their reload waits for vmcnt(0))
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = trv + s * 256, pq = idx >> 3;
            const int pos = pos0 + 4 * pq;
            const bool ok = idx < NIT && pos >= 0 && pos < L;
            unsigned vo = (unsigned)(4 * cq * L + (ok ? pos : 0)) * 2u;
            asm volatile("" : "+v"(vo));
#pragma unroll
            for (int q = 0; q < 4; ++q) pf[s][q] = *gptr<const u32x2>(inb + (size_t)q * L * 2 + vo);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = a.in_a ? a.in_a[b * 32 + 4 * cq + q] : 1.f;
            sv[q] = a.in_a ? a.in_s[b * 32 + 4 * cq + q] : 0.f;
        }
    };
    auto commit_x = [&](int pos0) {
        int trv = tr;
        asm volatile("" : "+v"(trv));
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int idx = trv + s * 256, pq = idx >> 3;
            if (idx >= NIT) continue;
            const int pos = pos0 + 4 * pq;
            const bool ok = pos >= 0 && pos < L;                                // L % 4 == 0: a position quad is inside or outside as a whole
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned char* dst = smem_q + rowaddr(XB, 4 * pq + e, cq >> 1) + ((cq & 1) << 3);
                float y[4], v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = (e & 1) ? n32_hi(pf[s][q][e >> 1]) : n32_lo(pf[s][q][e >> 1]);
                    y[q] = fmaf(av[q], xv, sv[q]);
                    v[q] = fmaxf(y[q], y[q] * slope);
                }
                u32x2 w = {n32_pack2(v[0], v[1]), n32_pack2(v[2], v[3])};
                u32x2 r = {n32_pack2(y[0], y[1]), n32_pack2(y[2], y[3])};
                if (!ok) { w = u32x2{0u, 0u}; r = w; }                          // the padding of the ACTIVATED signal is exactly 0
                *reinterpret_cast<u32x2*>(dst) = w;
                *reinterpret_cast<u32x2*>(dst + RT) = r;
            }
        }
    };

    const int count = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // tiles this workgroup walks (>= 1)
    typedef std::integral_constant<int, 1> D1;
    typedef std::integral_constant<int, 3> D3;
    typedef std::integral_constant<int, 3> K3;
    typedef std::integral_constant<int, 7> K7;
    typedef std::integral_constant<int, 11> K11;
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 3> P3;
    typedef std::integral_constant<int, 10> P10;

    // The two roles run their own loops (count + 1 iterations, four barriers each: s_barrier counts the workgroup's waves wherever they are).
    if (role == 0) {
        // =================== conv1 waves ===================
        f32x4 acc[NB], tsum[NB];
        issue_x(0);
        for (int it = 0; it <= count; ++it) {
            const bool a1 = it < count;                                         // (the last iteration only drains the conv2 waves)
            int b1_, n1;
            tile_of(a1 ? it : it - 1, b1_, n1);
            const bool edge = n1 - N32_H2 < 0 || n1 - N32_H2 + W > L;           // some window column lies outside the sequence
            // the t1 epilogue into tile TO: t1 = acc + x (r tile); the sum of the t1_j in fp32 registers, lrelu(t1) as bf16 into the tile
            auto epilogue = [&](unsigned TO) {
                int colv = col0;
                asm volatile("" : "+v"(colv));
#pragma unroll
                for (int h = 0; h < 2; ++h) {                                   // (two batches of residual rows: one LDS round trip each)
                    u32x2 rw[NB / 2];
#pragma unroll
                    for (int c = 0; c < NB / 2; ++c)
                        rw[c] = *reinterpret_cast<const u32x2*>(smem_q + rowaddr(RT, colv + 16 * (c + h * (NB / 2)) + N32_H1, 2 * rb + (kg >> 1)) + (kg & 1) * 8);
#pragma unroll
                    for (int c = 0; c < NB / 2; ++c) {
                        const int cb = c + h * (NB / 2);
                        const int col = colv + 16 * cb;
                        const u32x2 w = rw[c];
                        const f32x4 xr = {n32_lo(w[0]), n32_hi(w[0]), n32_lo(w[1]), n32_hi(w[1])};
                        f32x4 t1v = acc[cb] + xr;
                        if (edge) {                                             // conv2 zero-pads t1 outside the sequence
                            const int pos = n1 - N32_H2 + col;
                            if (pos < 0 || pos >= L) t1v = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        tsum[cb] += t1v;
                        const f32x4 ts = t1v * slope;
#pragma unroll
                        for (int r = 0; r < 4; ++r) t1v[r] = fmaxf(t1v[r], ts[r]);
                        *reinterpret_cast<u32x2*>(smem_q + rowaddr(TO, col, 2 * rb + (kg >> 1)) + (kg & 1) * 8) =
                            u32x2{n32_pack2(t1v[0], t1v[1]), n32_pack2(t1v[2], t1v[3])};
                    }
                }
            };
            auto init_acc = [&](const float (&bv)[4]) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{bv[0], bv[1], bv[2], bv[3]};
            };
            // ---- slot 0: the previous tile's sum of t1_j -> SF (the conv2 waves stored the tile before it in their slot 3), then x
            if (a1) V2W_STAMP(0);
            if (it >= 1) {
                int colv = col0;
                asm volatile("" : "+v"(colv));
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) SF[(16 * rb + 4 * kg + r) * SRS + colv + 16 * cb + 1] = tsum[cb][r];
            }
            if (a1) commit_x(n1 - N32_H1 - N32_H2);
            if (a1) V2W_STAMP(1);
            n32_lds_barrier();
            if (a1) V2W_STAMP(2);
            // ---- slot 1
            if (a1) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) tsum[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
                init_acc(bb[0]);
                conv(K3{}, D1{}, P0{}, acc, XB, col0 + N32_H1 - 1);
                if (a1) V2W_STAMP(3);
                epilogue(TA);
            }
            if (a1) V2W_STAMP(4);
            n32_lds_barrier();
            if (a1) V2W_STAMP(5);
            // ---- slot 2
            if (a1) {
                init_acc(bb[1]);
                conv(K7{}, D1{}, P3{}, acc, XB, col0 + N32_H1 - 3);
                if (a1) V2W_STAMP(6);
                epilogue(TBb);
            }
            if (a1) V2W_STAMP(7);
            n32_lds_barrier();
            if (a1) V2W_STAMP(8);
            // ---- slot 3
            if (a1) {
                init_acc(bb[2]);
                conv(K11{}, D1{}, P10{}, acc, XB, col0 + N32_H1 - 5);
                if (a1) V2W_STAMP(9);
            }
            if (a1) {
                epilogue(TA);
                if (a1) V2W_STAMP(10);
            }
            // the next tile's x, in flight under the barrier and slot 0's hand-over of the sum (unconditional - past the end the last tile again,
            // never committed: under a condition the old values would stay live through the whole iteration as the other input of the join)
            issue_x(min(it + 1, count - 1));
            n32_lds_barrier();
            if (a1) V2W_STAMP(13);
        }
    } else {
        // =================== conv2 waves ===================
        f32x4 acc[NB];
        for (int it = 0; it <= count; ++it) {
            const bool a1 = it < count, a2 = it >= 1;                           // tile `it` exists / tile `it - 1` is to be finished
            int b2_, n2;
            tile_of(a2 ? it - 1 : it, b2_, n2);
            // ---- slot 0: conv2_2 of the previous tile on T[a]
            if (a2) conv(K11{}, D3{}, P10{}, acc, TA, col0 - 15);
            n32_lds_barrier();
            // ---- slot 1: the running output takes the sum of the t1_j (fp32, written by the conv1 wave that owns the same channels and
            // columns), back in place
            if (a2) {
                int colv = col0;
                asm volatile("" : "+v"(colv));
#pragma unroll
                for (int cb = 0; cb < NB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = SF + (16 * rb + 4 * kg + r) * SRS + colv + 16 * cb + 1;
                        *p += acc[cb][r];
                    }
            }
            n32_lds_barrier();
            // ---- slot 2: conv2_0 of this tile on T[a]
            if (a1) {
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) acc[cb] = f32x4{bb[0][0], bb[0][1], bb[0][2], bb[0][3]};       // the sum of the b2_j
                conv(K3{}, D3{}, P0{}, acc, TA, col0 - 3);
            }
            n32_lds_barrier();
            // ---- slot 3: conv2_1 on T[b]; then the previous tile leaves: its nto valid columns (window columns 15 .. 15 + nto = SF columns
            // 16 ..) as 8-byte bf16 stores along positions (SF is rewritten by the conv1 waves in the next slot 0)
            if (a1) conv(K7{}, D3{}, P3{}, acc, TBb, col0 - 9);
            if (a2) {
                const int nq = nto >> 2;
                const unsigned magic = (unsigned)(((1ull << 32) + nq - 1) / nq);
                const float dinv = a.out_div != 0.f ? 1.f / a.out_div : 1.f;
                unsigned char* const obase = reinterpret_cast<unsigned char*>(a.out) + (size_t)b2_ * 32 * L * 2;
                for (int idx = tr; idx < 32 * nq; idx += 256) {
                    const int row = (int)__umulhi((unsigned)idx, magic), q = idx - row * nq;
                    const int pos = n2 + 4 * q;
                    if (pos >= L) continue;
                    f32x4 v = *reinterpret_cast<const f32x4*>(SF + row * SRS + 16 + 4 * q);
                    if (a.out_div != 0.f) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) v[x] = v2w_div_by(v[x], a.out_div, dinv);
                    }
                    *gptr<u32x2>(obase + (unsigned)(row * L + pos) * 2u) = u32x2{n32_pack2(v[0], v[1]), n32_pack2(v[2], v[3])};
                }
            }

            n32_lds_barrier();
        }
    }
}

int launch_n32(const v2w_stage_split_args* q, hipStream_t stream) {
    N32Args p{};
    p.in = reinterpret_cast<const unsigned short*>(q->in); p.in_a = q->in_a; p.in_s = q->in_s;
    p.out = reinterpret_cast<unsigned short*>(q->out);
    for (int j = 0; j < 3; ++j) {
        p.w1[j] = static_cast<const unsigned char*>(q->wps1[j]); p.bias1[j] = q->bias1[j];
        p.w2[j] = static_cast<const unsigned char*>(q->wps2[j]); p.bias2[j] = q->bias2[j];
    }
    p.B = q->B; p.L = q->L; p.slope = q->slope; p.out_div = q->out_div;
    p.nto = (N32_W - 2 * N32_H2) & ~3;
    p.ntl = (q->L + p.nto - 1) / p.nto;
    if ((long long)q->B * p.ntl > 0x7fffffffll) return V2W_E_SHAPE;
    p.ntiles = q->B * p.ntl;
    const size_t lds = (size_t)(2 * N32_XR + 2 * N32_W + 16) * 64 + (size_t)32 * N32_SRS * sizeof(float);
    const int ncu = v2w_num_cus();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(n32_stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    // persistent, one 8-wave workgroup per CU (151 KB of LDS; two waves per SIMD: one of each role)
    hipLaunchKernelGGL(n32_stage_kernel, dim3(p.ntiles < ncu ? p.ntiles : ncu), dim3(512), lds, stream, p);
    return v2w_launch_status();
}

}  // namespace

#ifdef V2W_TIMELINE
V2W_TL_SETTER(v2w_timeline_set_n32)
#endif

// Called by v2w_resblock2_stage_bf16 (v2w_stage_bf16.hip) for C = 32 on bf16 tensors.  V2W_E_SHAPE: not the reference's block set / not
// aligned - the caller runs the resident-tile template.
int v2w_resblock2_stage_bf16_n32(const v2w_stage_split_args* a, hipStream_t stream) {
    if (a->C != 32 || a->io_bf16 != 3 || !a->bf16 || a->nk != 3 || a->post_out) return V2W_E_SHAPE;
    for (int j = 0; j < 3; ++j)
        if (a->k[j] != 3 + 4 * j || a->dil1[j] != 1 || a->dil2[j] != 3 || !a->wps1[j] || !a->wps2[j]) return V2W_E_SHAPE;
    auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    if (a->L % 4 != 0 || !al16(a->in) || !al16(a->out) || !a->out) return V2W_E_SHAPE;
    if ((long long)32 * a->L * 2 >= (1ll << 31)) return V2W_E_SHAPE;            // 32-bit offsets inside one batch item
    if (!(a->slope > 0.f && a->slope < 1.f)) return V2W_E_SHAPE;
    return launch_n32(a, stream);
}
