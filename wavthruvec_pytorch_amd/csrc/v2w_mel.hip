// mel_spectrogram of the generated audio on the device (reference: vec2wav/dataset.py:53-77,
// called on the generator output in train.py:172-174, 266-269, 282-284):
//   reflect pad (n_fft - hop)/2  ->  STFT (hann window, center=False, onesided)  ->  sqrt(re^2 + im^2 + 1e-9)
//   ->  mel filterbank  ->  log(clamp(., 1e-5))
// The STFT is a dense contraction: with the padded signal de-interleaved by hop phase, Xp[b][p][f] = ypad[b][f*hop + p], frame f
// of the windowed DFT is  S[c][f] = sum_{j < n_fft/hop} sum_{p < hop} Wd[c][j*hop + p] * Xp[b][p][f + j]  - a Conv1d with hop
// input channels, n_fft/hop taps (pad_left = 0) and 2*(n_fft/2+1) output channels (cos rows, then -sin rows, window folded in):
// it runs on the f32 MFMA tile kernel through v2w_conv1d_fwd.  This file holds the two memory-bound ends.
#include "v2w_common.h"

namespace {

// Xp (B, hop, FP): Xp[b][p][f] = ypad[b][f*hop + p],  ypad = reflect-padded y (pad samples per side), 0 past the padded signal
__global__ void __launch_bounds__(256)
mel_phase_kernel(const float* __restrict__ y, float* __restrict__ xp, int L, int hop, int pad, int FP) {
    const int b = blockIdx.y;
    const int Lp = L + 2 * pad;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < hop * FP; idx += gridDim.x * 256) {
        const int p = idx / FP, f = idx - p * FP;
        const int i = f * hop + p;
        float v = 0.f;
        if (i < Lp) {
            int s = i - pad;
            s = s < 0 ? -s : (s >= L ? 2 * (L - 1) - s : s);     // torch 'reflect': no edge repeat
            v = y[(size_t)b * L + s];
        }
        xp[((size_t)b * hop + p) * FP + f] = v;
    }
}

// spec (B, Cs, FP): rows [0, nb) = Re, rows [nb, 2nb) = Im of the nb = n_fft/2+1 bins.  One block = 16 frames of one batch item:
// magnitudes into LDS, then the 16 x n_mels outputs (each a dot over the bins with the dense filterbank row).
#define V2W_MEL_FT 16
__global__ void __launch_bounds__(256)
mel_finish_kernel(const float* __restrict__ spec, const float* __restrict__ basis, float* __restrict__ out,
                  int Cs, int FP, int F, int nb, int n_mels) {
    extern __shared__ float mag[];                     // [nb][V2W_MEL_FT]
    const int b = blockIdx.y, f0 = blockIdx.x * V2W_MEL_FT;
    const float* sb = spec + (size_t)b * Cs * FP;
    for (int idx = threadIdx.x; idx < nb * V2W_MEL_FT; idx += 256) {
        const int c = idx / V2W_MEL_FT, ff = idx - c * V2W_MEL_FT;
        const int f = f0 + ff;
        float m = 0.f;
        if (f < F) {
            const float re = sb[(size_t)c * FP + f], im = sb[(size_t)(nb + c) * FP + f];
            m = sqrtf(re * re + im * im + 1e-9f);
        }
        mag[idx] = m;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_mels * V2W_MEL_FT; idx += 256) {
        const int m = idx / V2W_MEL_FT, ff = idx - m * V2W_MEL_FT;
        const int f = f0 + ff;
        if (f >= F) continue;
        const float* br = basis + (size_t)m * nb;
        float acc = 0.f;
        for (int c = 0; c < nb; ++c) acc = fmaf(br[c], mag[c * V2W_MEL_FT + ff], acc);
        out[((size_t)b * n_mels + m) * F + f] = logf(fmaxf(acc, 1e-5f));
    }
}

// Backward of mel_finish_kernel for the same 16-frame block: with acc = basis @ mag,
//   d_acc = g / acc where acc >= 1e-5 (the clamp passes the gradient on its closed side), d_mag = basis^T @ d_acc,
//   d_re = d_mag * re / mag, d_im = d_mag * im / mag      (mag = sqrt(re^2 + im^2 + 1e-9) > 0)
// basisT is the (nb, n_mels) transpose so phase 3 reads rows.  dspec must be zero-filled: only rows < 2*nb, frames < F are written.
__global__ void __launch_bounds__(256)
mel_finish_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ basis, const float* __restrict__ basisT,
                      const float* __restrict__ gout, float* __restrict__ dspec, int Cs, int FP, int F, int nb, int n_mels) {
    extern __shared__ float mag[];                     // [nb][FT] then d_acc [n_mels][FT]
    float* dacc = mag + (size_t)nb * V2W_MEL_FT;
    const int b = blockIdx.y, f0 = blockIdx.x * V2W_MEL_FT;
    const float* sb = spec + (size_t)b * Cs * FP;
    float* db = dspec + (size_t)b * Cs * FP;
    for (int idx = threadIdx.x; idx < nb * V2W_MEL_FT; idx += 256) {
        const int c = idx / V2W_MEL_FT, f = f0 + idx % V2W_MEL_FT;
        float m = 1.f;
        if (f < F) {
            const float re = sb[(size_t)c * FP + f], im = sb[(size_t)(nb + c) * FP + f];
            m = sqrtf(re * re + im * im + 1e-9f);
        }
        mag[idx] = m;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_mels * V2W_MEL_FT; idx += 256) {
        const int m = idx / V2W_MEL_FT, ff = idx - m * V2W_MEL_FT;
        const int f = f0 + ff;
        float d = 0.f;
        if (f < F) {
            const float* br = basis + (size_t)m * nb;
            float acc = 0.f;
            for (int c = 0; c < nb; ++c) acc = fmaf(br[c], mag[c * V2W_MEL_FT + ff], acc);
            if (acc >= 1e-5f) d = gout[((size_t)b * n_mels + m) * F + f] / acc;
        }
        dacc[idx] = d;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nb * V2W_MEL_FT; idx += 256) {
        const int c = idx / V2W_MEL_FT, ff = idx - c * V2W_MEL_FT;
        const int f = f0 + ff;
        if (f >= F) continue;
        const float* bt = basisT + (size_t)c * n_mels;
        float dm = 0.f;
        for (int m = 0; m < n_mels; ++m) dm = fmaf(bt[m], dacc[m * V2W_MEL_FT + ff], dm);
        const float s = dm / mag[idx];
        db[(size_t)c * FP + f] = s * sb[(size_t)c * FP + f];
        db[(size_t)(nb + c) * FP + f] = s * sb[(size_t)(nb + c) * FP + f];
    }
}

// Backward of mel_phase_kernel: dy[b][s] = sum of dxp over the padded positions i that read y[s]: i = s + pad always, the
// left reflection i = pad - s for 1 <= s <= pad, the right reflection i = 2(L-1) - s + pad for L-1-pad <= s <= L-2.
__global__ void __launch_bounds__(256)
mel_phase_bwd_kernel(const float* __restrict__ dxp, float* __restrict__ dy, int L, int hop, int pad, int FP) {
    const int b = blockIdx.y;
    const float* xb = dxp + (size_t)b * hop * FP;
    auto at = [&](int i) { return i / hop < FP ? xb[(size_t)(i % hop) * FP + i / hop] : 0.f; };    // samples past the last frame: unused
    for (int s = blockIdx.x * 256 + threadIdx.x; s < L; s += gridDim.x * 256) {
        float v = at(s + pad);
        if (s >= 1 && s <= pad) v += at(pad - s);
        if (s >= L - 1 - pad && s <= L - 2) v += at(2 * (L - 1) - s + pad);
        dy[(size_t)b * L + s] = v;
    }
}

}  // namespace

// y (B, L) -> xp (B, hop, FP); FP >= F + n_fft/hop - 1 frames columns (F = number of STFT frames), pad = (n_fft - hop)/2.
extern "C" int v2w_mel_phases(const float* y, float* xp, int B, int L, int hop, int pad, int FP, void* stream) {
    if (!y || !xp || B <= 0 || L <= 1 || hop <= 0 || pad < 0 || pad >= L || FP <= 0) return V2W_E_ARG;
    int gx = (hop * FP + 255) / 256; if (gx > 1024) gx = 1024;
    V2W_LAUNCH(mel_phase_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, y, xp, L, hop, pad, FP);
    return v2w_launch_status();
}

// spec (B, Cs, FP) from the DFT conv (Cs >= 2*nb rows), basis (n_mels, nb) -> out (B, n_mels, F) = log(clamp(basis @ |spec|, 1e-5))
extern "C" int v2w_mel_finish(const float* spec, const float* basis, float* out, int B, int Cs, int FP, int F, int nb, int n_mels,
                              void* stream) {
    if (!spec || !basis || !out || B <= 0 || F <= 0 || FP < F || nb <= 0 || Cs < 2 * nb || n_mels <= 0) return V2W_E_ARG;
    const size_t lds = (size_t)nb * V2W_MEL_FT * sizeof(float);
    if (lds > 64 * 1024) return V2W_E_SHAPE;
    V2W_LAUNCH(mel_finish_kernel, dim3((F + V2W_MEL_FT - 1) / V2W_MEL_FT, B), dim3(256), lds, (hipStream_t)stream,
                       spec, basis, out, Cs, FP, F, nb, n_mels);
    return v2w_launch_status();
}


// Backward of v2w_mel_finish: gout (B, n_mels, F) -> dspec (B, Cs, FP), which the caller zero-fills (pad rows / frames stay 0).
// basisT (nb, n_mels) is the transposed filterbank.
extern "C" int v2w_mel_finish_bwd(const float* spec, const float* basis, const float* basisT, const float* gout, float* dspec,
                                  int B, int Cs, int FP, int F, int nb, int n_mels, void* stream) {
    if (!spec || !basis || !basisT || !gout || !dspec || B <= 0 || F <= 0 || FP < F || nb <= 0 || Cs < 2 * nb || n_mels <= 0)
        return V2W_E_ARG;
    const size_t lds = (size_t)(nb + n_mels) * V2W_MEL_FT * sizeof(float);
    if (lds > 64 * 1024) return V2W_E_SHAPE;
    V2W_LAUNCH(mel_finish_bwd_kernel, dim3((F + V2W_MEL_FT - 1) / V2W_MEL_FT, B), dim3(256), lds, (hipStream_t)stream,
                       spec, basis, basisT, gout, dspec, Cs, FP, F, nb, n_mels);
    return v2w_launch_status();
}

// Backward of v2w_mel_phases: dxp (B, hop, FP) -> dy (B, L); the reflected samples fold back onto their sources.
extern "C" int v2w_mel_phases_bwd(const float* dxp, float* dy, int B, int L, int hop, int pad, int FP, void* stream) {
    if (!dxp || !dy || B <= 0 || L <= 1 || hop <= 0 || pad < 0 || pad >= L || FP <= 0) return V2W_E_ARG;
    int gx = (L + 255) / 256; if (gx > 1024) gx = 1024;
    V2W_LAUNCH(mel_phase_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, dxp, dy, L, hop, pad, FP);
    return v2w_launch_status();
}
