// The generator's tail on bf16 activations: y = tanh(conv_post(leaky_relu(x, slope)))  (reference: vec2wav/models.py:143-145), x (B, C, L) bf16,
// C = 16 or 8 channels -> ONE output channel, k <= 9 taps, fp32 audio out.  Called by v2w_conv_post_tanh_bf16in (v2w_direct.hip) before its
// vector-ALU kernel, which at BASELINE configs[2] (B = 64, L = 163 840) spent 190 us on 112 FMAs + 36 activations per output.
//
// Here the taps run on the matrix pipe as a TOEPLITZ product, with no LDS tile and no transposition of x:
//   a tile = 256 consecutive outputs = 16 groups j of 16 outputs i;  out[16 j + i] = sum_c sum_s A_c[i][s] * X_c[s][j]
//   X_c[s][j] = x[c][tile + 16 j + s - 8]   (s = 0 .. 31: the B operand of v_mfma_f32_16x16x32_bf16 - lane (j, kg) holds s = 8 kg .. 8 kg + 7,
//                                            i.e. 16 CONTIGUOUS, 16-byte ALIGNED bytes of the row of channel c: one global_load_dwordx4)
//   A_c[i][s] = w[c][s - i + pad - 8]        (0 outside the k taps: a banded 16 x 32 matrix per channel, built once per workgroup in LDS;
//                                            the band covers s = 4 .. 27 for k <= 9)
// The activation is not applied element by element either: lrelu(x) = slope x + (1 - slope) relu(x), and relu of packed bf16 is ONE
// v_pk_max_i16 against 0 per register (a negative bf16 is a negative int16), so a channel costs 4 vector-ALU instructions per tile and
// three MFMAs: (slope w) . x  +  hi((1 - slope) w) . relu(x)  +  lo((1 - slope) w) . relu(x).  The weights of the dominant term are split
// into two bf16 (16 mantissa bits together), so the product stays at the precision of the bf16 activations; the slope-scaled term is 1 % of
// the sum and takes one bf16.
#include "v2w_common.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

struct PostArgs {
    const unsigned short* in; const float* wf; const float* bias; float* out;
    int B, L, k, jobs_per_item, njobs;
    float slope;
};

#define V2W_PM_NT 2
#define V2W_PM_DEPTH 3                            // channels of x in flight ahead of the running one
#define V2W_PM_WGS 3                              // resident workgroups per CU the grid is cut for
constexpr int PM_NT = V2W_PM_NT;                  // tiles of 256 outputs per wave and job
constexpr int PM_JOB = 4 * PM_NT * 256;           // outputs per workgroup job (4 waves)

__device__ __forceinline__ unsigned int pm_bf16_pair(float lo, float hi) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ float pm_round_bf16(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ unsigned int pm_relu2(unsigned int w) {
    return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), s16x2{0, 0}));
}
// tanh(v) = 1 - 2 / (1 + e^(2 v)): v_exp_f32 + v_rcp_f32, both 1 ulp; saturates to +-1 through inf / 0
__device__ __forceinline__ float pm_tanh(float v) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * 2.885390081777927f));
}

template <int C>
__global__ void __launch_bounds__(256, V2W_PM_WGS)
conv_post_tanh_mfma_kernel(const PostArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 aimg[];       // [C][3][64 lanes]: the three A operands of every channel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L, k = a.k, pad = (k - 1) / 2;
    const float slope = a.slope;
    for (int idx = tid; idx < C * 64; idx += 256) {
        const int c = idx >> 6, l = idx & 63;
        const int i = l & 15, kg = l >> 4;
        float w1[8], wh[8], wl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = 8 * kg + e - i + pad - 8;
            const float w = (t >= 0 && t < k) ? a.wf[t * C + c] : 0.f;
            w1[e] = slope * w;
            const float w2 = (1.f - slope) * w;
            wh[e] = pm_round_bf16(w2);
            wl[e] = w2 - wh[e];
        }
        aimg[(c * 3 + 0) * 64 + l] = u32x4{pm_bf16_pair(w1[0], w1[1]), pm_bf16_pair(w1[2], w1[3]), pm_bf16_pair(w1[4], w1[5]), pm_bf16_pair(w1[6], w1[7])};
        aimg[(c * 3 + 1) * 64 + l] = u32x4{pm_bf16_pair(wh[0], wh[1]), pm_bf16_pair(wh[2], wh[3]), pm_bf16_pair(wh[4], wh[5]), pm_bf16_pair(wh[6], wh[7])};
        aimg[(c * 3 + 2) * 64 + l] = u32x4{pm_bf16_pair(wl[0], wl[1]), pm_bf16_pair(wl[2], wl[3]), pm_bf16_pair(wl[4], wl[5]), pm_bf16_pair(wl[6], wl[7])};
    }
    __syncthreads();
    const float bv = a.bias ? a.bias[0] : 0.f;
    const int j = lane & 15, kg = lane >> 4;
    const int loff = 16 * j + 8 * kg - 8;                               // this lane's first element inside a tile's 16-byte-per-lane image

    for (int job = blockIdx.x; job < a.njobs; job += gridDim.x) {
        const int b = job / a.jobs_per_item;
        const int base = (job - b * a.jobs_per_item) * PM_JOB + wave * (PM_NT * 256);
        if (base >= L) continue;                                        // (no barrier below: waves past the end of the row just skip)
        const unsigned short* const rows = a.in + (size_t)b * C * L;
        // every position any lane of this wave reads lies inside the row: no per-load checks
        const bool interior = base >= 8 && base + PM_NT * 256 + 32 <= L;
        auto load = [&](int c, u32x4 (&xv)[PM_NT]) {
#pragma unroll
            for (int t = 0; t < PM_NT; ++t) {
                const int pos = base + 256 * t + loff;                 // a multiple of 8, like L: the 8 elements are inside or outside together
                const bool ok = interior || (pos >= 0 && pos < L);
                u32x4 v = {0u, 0u, 0u, 0u};                             // conv_post zero-pads the ACTIVATED signal, and lrelu(0) = 0
                if (ok) v = *reinterpret_cast<const u32x4*>(rows + (size_t)c * L + pos);
                xv[t] = v;
            }
        };
        f32x4 acc[PM_NT];
#pragma unroll
        for (int t = 0; t < PM_NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NB = V2W_PM_DEPTH + 1;
        u32x4 xv[NB][PM_NT];
#pragma unroll
        for (int c = 0; c < V2W_PM_DEPTH && c < C; ++c) load(c, xv[c]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (c + V2W_PM_DEPTH < C) load(c + V2W_PM_DEPTH, xv[(c + V2W_PM_DEPTH) % NB]);      // later channels' rows fly under this channel's MFMAs
            const b8 a1 = __builtin_bit_cast(b8, aimg[(c * 3 + 0) * 64 + lane]);
            const b8 ah = __builtin_bit_cast(b8, aimg[(c * 3 + 1) * 64 + lane]);
            const b8 al = __builtin_bit_cast(b8, aimg[(c * 3 + 2) * 64 + lane]);
#pragma unroll
            for (int t = 0; t < PM_NT; ++t) {
                const u32x4 x = xv[c % NB][t];
                const u32x4 r = {pm_relu2(x[0]), pm_relu2(x[1]), pm_relu2(x[2]), pm_relu2(x[3])};
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, __builtin_bit_cast(b8, x), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(b8, r), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(b8, r), acc[t], 0, 0, 0);
            }
        }
        // accumulator register r of lane (j, kg) = output 16 j + 4 kg + r of the tile: one aligned float4 per lane, 1 KiB contiguous per wave
        float* const orow = a.out + (size_t)b * L;
#pragma unroll
        for (int t = 0; t < PM_NT; ++t) {
            const int pos = base + 256 * t + 16 * j + 4 * kg;
            if (pos >= L) continue;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = pm_tanh(acc[t][r] + bv);
            *reinterpret_cast<f32x4*>(orow + pos) = y;
        }
    }
}

template <int C>
int launch_post(const PostArgs& p, hipStream_t stream) {
    const int ncu = v2w_num_cus();
    // two resident workgroups per CU, each builds the operand image once and walks its jobs
    const int grid = p.njobs < V2W_PM_WGS * ncu ? p.njobs : V2W_PM_WGS * ncu;
    V2W_LAUNCH(conv_post_tanh_mfma_kernel<C>, dim3(grid), dim3(256), (size_t)C * 3 * 64 * sizeof(u32x4), stream, p);
    return v2w_launch_status();
}

}  // namespace

// V2W_E_SHAPE: not this kernel's case (the caller runs its vector-ALU kernels).
int v2w_conv_post_tanh_bf16_mfma(const unsigned short* in, const float* wf, const float* bias, float* out,
                                 int B, int C_in, int L, int k, float slope, hipStream_t stream) {
    if ((C_in != 16 && C_in != 8) || k < 1 || k > 9 || !(k & 1) || L % 8 != 0 || L < 8) return V2W_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return V2W_E_SHAPE;
    if (!(slope >= 0.f && slope <= 1.f)) return V2W_E_SHAPE;             // lrelu as slope x + (1 - slope) relu(x)
    PostArgs p{};
    p.in = in; p.wf = wf; p.bias = bias; p.out = out; p.B = B; p.L = L; p.k = k; p.slope = slope;
    p.jobs_per_item = (L + PM_JOB - 1) / PM_JOB;
    if ((long long)B * p.jobs_per_item > 0x7fffffffll) return V2W_E_SHAPE;
    p.njobs = B * p.jobs_per_item;
    return C_in == 16 ? launch_post<16>(p, stream) : launch_post<8>(p, stream);
}
