// extern "C" dispatch of the conv entry points declared in include/vec2wav_hip.h.
#include "v2w_common.h"

int v2w_conv1d_mfma(const v2w_conv1d_args* a, int n, hipStream_t stream, int* cfg_out, long long* ws_query = nullptr);
int v2w_convt1d_mfma(const v2w_convt1d_args* a, hipStream_t stream, int* cfg_out, long long* ws_query = nullptr);
int v2w_conv1d_split(const v2w_conv1d_args* a, int n, hipStream_t stream, bool bf16);
int v2w_conv1d_direct(const v2w_conv1d_args* a, hipStream_t stream);
int v2w_convt1d_direct(const v2w_convt1d_args* a, hipStream_t stream);

extern "C" int v2w_abi_version(void) { return V2W_ABI_VERSION; }
extern "C" const char* v2w_build_arch(void) { return "gfx950"; }

static int check_conv1d(const v2w_conv1d_args* a) {
    if (!a || !a->in || !a->out) return V2W_E_ARG;
    if ((a->algo == V2W_ALGO_SPLIT || a->algo == V2W_ALGO_BF16) ? (!a->wps || !a->winv) : (!a->wf && !a->wp)) return V2W_E_ARG;
    if (a->B <= 0 || a->C_in <= 0 || a->C_out <= 0 || a->L <= 0 || a->k <= 0 || a->dil <= 0) return V2W_E_ARG;
    if ((a->k & 1) == 0 && a->pad_left < 0) return V2W_E_SHAPE;    // symmetric padding d*(k-1)/2 keeps the length only for odd k
    if (a->pad_left > a->dil * (a->k - 1) || a->in_stride < 0 || a->in_phase < 0 || (a->in_stride > 0 && a->in_phase >= a->in_stride))
        return V2W_E_ARG;
    if ((a->in_a == nullptr) != (a->in_s == nullptr)) return V2W_E_ARG;
    if ((a->res_a == nullptr) != (a->res_s == nullptr)) return V2W_E_ARG;
    if (a->res_a && !a->res) return V2W_E_ARG;
    if (a->add1 && !a->add0) return V2W_E_ARG;
    if (a->add0 && a->accumulate) return V2W_E_ARG;               // either the running sum in `out` or explicit addends
    if ((a->mask_a == nullptr) != (a->mask_s == nullptr)) return V2W_E_ARG;
    if (a->mask_a && !a->mask_src) return V2W_E_ARG;
    if (a->in_ct < 0 || a->out_ct < 0 || (a->in_ct > 0 && a->in_ct < a->C_in) || (a->out_ct > 0 && a->out_ct < a->C_out)) return V2W_E_ARG;
    if (((a->in_ct > 0 && a->in_ct != a->C_in) && a->in_a) || ((a->out_ct > 0 && a->out_ct != a->C_out) && (a->res_a || a->mask_a)))
        return V2W_E_ARG;                                          // the affine tables are indexed by the slice's own channel count
    if (a->out_slope < 0.f) return V2W_E_ARG;
    if (a->rowsum_part && (a->algo != V2W_ALGO_MFMA || !a->mask_src)) return V2W_E_ARG;       // row sums: masked f32 MFMA launches only
    if (a->io_bf16 != 0 && (a->algo != V2W_ALGO_BF16 || a->io_bf16 < 0 || a->io_bf16 > 3)) return V2W_E_ARG;   // bf16 storage: bf16 kernels only
    const bool ext_ct = (a->in_ct > 0 && a->in_ct != a->C_in) || (a->out_ct > 0 && a->out_ct != a->C_out);
    const bool ext_slope = a->out_slope != 0.f && a->out_slope != 1.f;
    if ((ext_ct || ext_slope) && a->algo == V2W_ALGO_BF16) return V2W_E_SHAPE;      // channel slices, out_slope: the f32 and split-f16 kernels
    return 0;
}

extern "C" int v2w_conv1d_fwd(const v2w_conv1d_args* a, void* stream) {
    if (const int rc = check_conv1d(a)) return rc;
    hipStream_t st = (hipStream_t)stream;
    switch (a->algo) {
        case V2W_ALGO_DIRECT: return a->wf ? v2w_conv1d_direct(a, st) : V2W_E_ARG;
        case V2W_ALGO_MFMA: return v2w_conv1d_mfma(a, 1, st, nullptr);
        case V2W_ALGO_SPLIT: return v2w_conv1d_split(a, 1, st, false);
        case V2W_ALGO_BF16: return v2w_conv1d_split(a, 1, st, true);
        case V2W_ALGO_AUTO: {
            const int rc = v2w_conv1d_mfma(a, 1, st, nullptr);
            return rc == V2W_E_SHAPE ? (a->wf ? v2w_conv1d_direct(a, st) : V2W_E_ARG) : rc;
        }
        default: return V2W_E_ALGO;
    }
}

extern "C" int v2w_convt1d_fwd(const v2w_convt1d_args* a, void* stream) {
    if (!a || !a->in || (!a->wf && !a->wp) || !a->out || a->io_bf16 != 0) return V2W_E_ARG;
    if (a->B <= 0 || a->C_in <= 0 || a->C_out <= 0 || a->L <= 0 || a->k <= 0 || a->u <= 0) return V2W_E_ARG;
    if (a->k < a->u || ((a->k - a->u) & 1)) return V2W_E_SHAPE;    // L_out = u*L needs k-u even
    hipStream_t st = (hipStream_t)stream;
    switch (a->algo) {
        case V2W_ALGO_DIRECT: return (a->wf && !a->stats_part) ? v2w_convt1d_direct(a, st) : V2W_E_ARG;
        case V2W_ALGO_MFMA: return v2w_convt1d_mfma(a, st, nullptr);
        case V2W_ALGO_AUTO: {
            const int rc = v2w_convt1d_mfma(a, st, nullptr);
            return rc == V2W_E_SHAPE ? ((a->wf && !a->stats_part) ? v2w_convt1d_direct(a, st) : V2W_E_ARG) : rc;
        }
        default: return V2W_E_ALGO;
    }
}

// Which conv_tile_kernel<MF,U,MI,NI,WM,WN,CK,NPF,RING> instantiation V2W_ALGO_AUTO/MFMA picks for this problem
// (pointers in `a` are not dereferenced).  Returns 0 and fills cfg[9], or V2W_E_SHAPE when the direct kernel is used.
// cfg[10]: MF,U,MI,NI,WM,WN,CK,NPF,RING and cfg[9] = number of position tiles
extern "C" int v2w_conv1d_tile_config(const v2w_conv1d_args* a, int32_t* cfg) {
    if (!a || !cfg) return V2W_E_ARG;
    return v2w_conv1d_mfma(a, 1, nullptr, cfg);
}
extern "C" int v2w_convt1d_tile_config(const v2w_convt1d_args* a, int32_t* cfg) {
    if (!a || !cfg) return V2W_E_ARG;
    return v2w_convt1d_mfma(a, nullptr, cfg);
}

// Bytes of caller scratch (v2w_conv1d_args::splitk_ws) the f32 MFMA launch of these problems would split into; 0: unsplit or not an MFMA launch
extern "C" long long v2w_conv1d_splitk_ws_bytes(const v2w_conv1d_args* a, int n) {
    if (!a || n < 1 || n > 4 || (a->algo != V2W_ALGO_AUTO && a->algo != V2W_ALGO_MFMA)) return 0;
    long long bytes = 0;
    return v2w_conv1d_mfma(a, n, nullptr, nullptr, &bytes) == 0 ? bytes : 0;
}
extern "C" long long v2w_convt1d_splitk_ws_bytes(const v2w_convt1d_args* a) {
    if (!a || a->u <= 0 || a->k < a->u || (a->algo != V2W_ALGO_AUTO && a->algo != V2W_ALGO_MFMA)) return 0;
    long long bytes = 0;
    return v2w_convt1d_mfma(a, nullptr, nullptr, &bytes) == 0 ? bytes : 0;
}

// n (<= 4) fused convs that share B, C_in, C_out and L in ONE launch: the residual branches of a generator stage read the
// same input / write independent outputs, so their tiles are mixed (heaviest first) instead of paying one launch tail each.
// MFMA path only; V2W_E_SHAPE tells the caller to issue them one by one.
extern "C" int v2w_conv1d_fwd_multi(const v2w_conv1d_args* a, int n, void* stream) {
    if (!a || n < 1 || n > 4) return V2W_E_ARG;
    for (int i = 0; i < n; ++i) {
        if (const int rc = check_conv1d(a + i)) return rc;
        if (a[i].algo == V2W_ALGO_DIRECT) return V2W_E_SHAPE;
        if (a[i].algo != a[0].algo && (a[i].algo >= V2W_ALGO_SPLIT || a[0].algo >= V2W_ALGO_SPLIT)) return V2W_E_ARG;   // one kernel per launch
    }
    if (a[0].algo == V2W_ALGO_SPLIT || a[0].algo == V2W_ALGO_BF16)
        return v2w_conv1d_split(a, n, (hipStream_t)stream, a[0].algo == V2W_ALGO_BF16);
    return v2w_conv1d_mfma(a, n, (hipStream_t)stream, nullptr);
}
