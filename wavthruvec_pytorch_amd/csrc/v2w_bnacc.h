// BatchNorm statistics WITHOUT launches of their own (ABI v29; modules.py:14,23 - F.batch_norm in training mode - between two kernels of
// the bf16 pipeline).  Until round 4 a producing kernel wrote one row of per-tile (sum, sumsq) per channel, a reduce launch added the rows
// and a finalize launch turned the totals into the folded affine (a, s) of every (sample, channel): two latency-bound launches and three
// queue gaps (~6 us each) between every stage kernel and the next - 75-100 us of a 1.5 ms forward at B = 32 x T = 256.
//
// Here the PRODUCER adds its per-tile sums straight into an accumulator of 4 x int64 per channel with device-scope integer atomics, and
// the CONSUMER - the next stage kernel - turns the totals into (a, s) of its own sample in its prologue, while its tile's loads fly.
//   * Integer addition is associative: the totals do not depend on the order the tiles arrive in - bit-reproducible like the fixed-order
//     reductions they replace (float atomics would not be).
//   * Fixed point in TWO words per sum: p = hi * 2^8 + lo * 2^-40 with hi = trunc(p / 2^8) and lo = rint((p - hi 2^8) 2^40), |lo| < 2^48.
//     Range |sum of hi| < 2^63, i.e. per-tile sums up to 2^71 / tiles; resolution 2^-40 per tile (an fp32 partial carries 24 bits).
//   * The caller zeroes the accumulator before the producer runs (one memset per forward for all stages) and passes the element count.
// Arithmetic of the fold = v2w_bn_finalize (v2w_cbn.hip): fp64 mean / biased variance, rstd = (float)(1 / sqrt(var + eps)),
// a = gamma * rstd, s = fma(-a, (float)mean, beta); running statistics with the unbiased variance, by ONE workgroup of the consumer.
#pragma once
#include "v2w_common.h"

namespace {

// what a consumer needs to fold the statistics of its input itself (all NULL / 0: the (a, s) tables are given)
struct BnFoldArgs {
    const long long* acc;      // [C][4]: sum hi, sum lo, sumsq hi, sumsq lo (the producer's atomics)
    const float* gb;           // (B, 2 C): gamma | beta of every sample (v2w_cond_gamma_beta)
    float* running_mean; float* running_var; long long* nbt;      // updated by the workgroup that is told to (NULL: not at all)
    double count;              // elements per channel = B * L
    float eps, momentum;
};

__device__ __forceinline__ void bnacc_add(long long* acc, int c, float s1, float s2) {
    const float v[2] = {s1, s2};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double d = (double)v[i];
        const long long hi = (long long)(d * 0x1p-8);                       // truncation: |d - hi 2^8| < 2^8
        const long long lo = __double2ll_rn((d - (double)hi * 0x1p8) * 0x1p40);
        __hip_atomic_fetch_add(acc + 4 * c + 2 * i, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(acc + 4 * c + 2 * i + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (mean, rstd) of channel c from the accumulator; `update` (uniform per workgroup): this thread also writes the running statistics
__device__ __forceinline__ void bnacc_mean_rstd(const BnFoldArgs& f, int c, bool update, float& mean_f, float& rstd) {
    const long long* q = f.acc + 4 * c;
    const long long h1 = q[0], l1 = q[1], h2 = q[2], l2 = q[3];            // (written by the previous kernel's atomics: plain loads)
    const double s1 = (double)h1 * 0x1p8 + (double)l1 * 0x1p-40, s2 = (double)h2 * 0x1p8 + (double)l2 * 0x1p-40;
    const double mean = s1 / f.count;
    double var = s2 / f.count - mean * mean;                                // biased (normalisation) variance
    if (var < 0.0) var = 0.0;
    mean_f = (float)mean;
    rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    if (update && f.running_mean) {
        // F.batch_norm: running = (1 - m) running + m stat, running_var with the UNBIASED batch variance
        const double unb = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
        const double m = (double)f.momentum;
        f.running_mean[c] = (float)((1.0 - m) * (double)f.running_mean[c] + m * mean);
        f.running_var[c] = (float)((1.0 - m) * (double)f.running_var[c] + m * unb);
        if (c == 0 && f.nbt) *f.nbt += 1;
    }
}

}  // namespace
