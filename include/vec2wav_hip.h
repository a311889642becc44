/*
 * vec2wav_hip.h - C ABI of libvec2wav_hip.so: the Vec2Wav generator forward path as
 * hand-written HIP kernels for gfx950 (MI355X / CDNA4).
 *
 * The reference (p1an-lin-jung/WavThruVec_pytorch) has no FFI/operator layer: its boundary is
 * the Python nn.Module surface (vec2wav/models.py:77-156, vec2wav/modules.py:5-30) and everything
 * underneath is stock torch ops.  Each entry point below therefore cites the reference statement
 * (file:line under /root/reference) whose arithmetic it replaces; the Python mirror of the
 * reference surface (wavthruvec_pytorch_amd/models.py) is the only caller.  INTEGRATION.md shows
 * the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no torch types; every pointer is a DEVICE pointer unless said
 *     otherwise; activations are fp32, channels-first, contiguous (B, C, L);
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); launches are asynchronous,
 *     nothing allocates, nothing synchronises, no ownership is transferred;
 *   - return value: 0 = enqueued; < 0 = bad argument (V2W_E_*); > 0 = a hipError_t from the launch;
 *   - no global mutable state: safe to call concurrently on distinct streams.
 *
 * Folded weight layout ("wf"): fp32 [k][C_in][C_out] (C_out fastest) for both Conv1d and
 * ConvTranspose1d, produced by v2w_wn_fold_* from the reference's weight_g/weight_v parameters.
 */
#ifndef VEC2WAV_HIP_H
#define VEC2WAV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2W_ABI_VERSION 34

#define V2W_E_ARG      (-1)  /* null pointer / non-positive size */
#define V2W_E_SHAPE    (-2)  /* shape not supported by the requested algorithm */
#define V2W_E_ALGO     (-3)  /* unknown algorithm id */

/* algorithm selector of the conv entry points */
#define V2W_ALGO_AUTO   0    /* MFMA tile kernel when the shape allows, else the direct kernel */
#define V2W_ALGO_DIRECT 1    /* one-thread-per-output scalar FMA kernel: any shape; cross-check */
#define V2W_ALGO_MFMA   2    /* f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) implicit GEMM; V2W_E_SHAPE if unsupported */
#define V2W_ALGO_BF16   4    /* bf16 operands (rne), ONE v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate: BASELINE configs[2];
                              * same kernel, tiles and buffers as V2W_ALGO_SPLIT with fragments from v2w_pack_bf16 */
#define V2W_ALGO_SPLIT  3    /* split-f16 MFMA: x = hi + lo halves, x_hi*w_hi + x_hi*w_lo + x_lo*w_hi accumulated in fp32 (~22-bit
                              * products; v_mfma_f32_32x32x16_f16); needs wps / winv from v2w_pack_split; Conv1d with
                              * C_in % 16 == 0, C_out % 64 == 0 and an odd k >= 3, else V2W_E_SHAPE */

int         v2w_abi_version(void);
const char* v2w_build_arch(void);       /* "gfx950" */

/* ---- Which kernels would this call launch?  (ABI v33; host-only.)  Every launching entry point - the ones whose last argument is the stream -
 * accepts a NAME SINK in place of the stream: the call runs every shape check and every kernel selection it would make, launches nothing, sets
 * no attribute, dereferences no tensor pointer, and appends the demangled name of each kernel it would have launched, one per line and in launch
 * order, to the caller's buffer (exactly the string `rocprofv3 --kernel-trace` prints for that kernel: "void (anonymous namespace)::
 * wide_stage_bf16_kernel<1, 4, 1, 2, 2, 32, false, true, 2, false>((anonymous namespace)::WideArgs)").  The return value is the one the real call
 * would give up to the launch (0, or V2W_E_*).  bench.py labels its per-kernel rooflines and looks up the counter profiles with these names
 * instead of a table of template arguments kept by hand.  The sink is the caller's: { V2W_NAME_SINK_MAGIC, buf, cap, 0 }; `len` counts
 * the bytes written (names that do not fit are dropped, `len` stays below cap, the buffer is always NUL-terminated). */
typedef struct { uint64_t magic; char* buf; int32_t cap; int32_t len; } v2w_name_sink;
#define V2W_NAME_SINK_MAGIC 0x5632574e414d4531ull
#define V2W_NAME_SINK_STREAM(sink) ((void*)((uintptr_t)(sink) | (uintptr_t)1))   /* pass as `stream`: real stream handles are aligned pointers */

/* ---- K0: weight-norm fold (torch.nn.utils.weight_norm pre-forward hook, dim=0; triggered by
 * models.py:18-33,58-61,83,90-92,100).  w = g * v / ||v||, norm over all dims but 0.
 * conv : v (C_out, C_in, k), g (C_out)  -> wf [k][C_in][C_out]      (norm per C_out)
 * convt: v (C_in, C_out, k), g (C_in)   -> wf [k][C_in][C_out]      (norm per C_in)
 * g == NULL means "weight norm already removed" (models.py:149-156): v is the plain weight, relayout only.
 * scratch: >= rows floats (rows = C_out for conv, C_in for convt). */
int v2w_wn_fold_conv (const float* v, const float* g, float* wf, float* scratch,
                      int c_out, int c_in, int k, void* stream);
int v2w_wn_fold_convt(const float* v, const float* g, float* wf, float* scratch,
                      int c_in, int c_out, int k, void* stream);

/* dgrad weights: out[t][C_out][C_in] = wf[k-1-t][C_in][C_out]; the conv that back-propagates through Conv1d(w, dilation d)
 * is a Conv1d of the output gradient with these weights and the same dilation (then v2w_pack_mfma(out, k, C_out, C_in)). */
int v2w_wf_transpose_flip(const float* wf, float* out, int k, int c_in, int c_out, void* stream);
/* general form: out[m][C_out][C_in] = wf[t_start + m*t_step][C_in][C_out], m < n.  Phase r of ConvTranspose1d(k, u, pad):
 * t_start = (r+pad)%u, t_step = u, n = taps of the phase; run as Conv1d over dy's phase r with pad_left = (r+pad)/u. */
int v2w_wf_gather_transpose(const float* wf, float* out, int k, int c_in, int c_out, int t_start, int t_step, int n, void* stream);

/* MFMA operand packing: wf [k][C_in][C_out] -> wp, the same k*C_in*C_out weights as a stream of 1 KiB MFMA A-fragments
 * (64 lanes x float4 = four consecutive MFMA k-steps) in exactly the order the tile kernel of that layer consumes them:
 * [row block mb][C_in chunk][tap, phase-major for a transposed conv][fragment]; the kernel reads it strictly
 * sequentially.  u = 1 for Conv1d, the stride for ConvTranspose1d.  Returns V2W_E_SHAPE when the layer has no MFMA
 * tile configuration (C_in % 16 != 0, C_out neither 16 nor a multiple of 32, unsupported stride): such layers run on
 * the direct kernel with wp = NULL. */
int v2w_pack_mfma(const float* wf, float* wp, int k, int c_in, int c_out, int u, void* stream);
/* the stream of the layer's input-gradient conv (== v2w_pack_mfma of v2w_wf_transpose_flip(wf)) straight from wf [k][c_out][c_in]:
 * c_in / c_out are the GRADIENT conv's channel counts (backward of models.py:65-70: dx = conv(dy; W^T, taps reversed)) */
int v2w_pack_mfma_dgrad(const float* wf, float* wp, int k, int c_in, int c_out, void* stream);
/* n matrices back to back -> n packed streams back to back in one launch (the groups of a grouped conv) */
int v2w_pack_mfma_batch(const float* wf, float* wp, int k, int c_in, int c_out, int u, int n, void* stream);

/* Batched form of fold + pack for every MFMA layer of a generator: two launches instead of three per layer.
 *   v2w_fold_plan       (host only) fills mf/ck of each descriptor and starts[2*(n+1)] (block prefix sums of the scale
 *                       and of the pack kernel); returns the dynamic-LDS byte count (> 0) the pack kernel needs, or
 *                       V2W_E_SHAPE when a layer has no MFMA tile configuration (fold such layers one by one).
 *   v2w_fold_pack_batch descs_dev / starts_dev are DEVICE copies of the planned arrays; nblk_scale = starts[n],
 *                       nblk_pack = starts[2n+1].
 * v: weight_v (conv (C_out,C_in,k); transposed (C_in,C_out,k)); g: weight_g or NULL; scale: >= rows floats scratch. */
typedef struct {
    const float* v; const float* g; float* wp; float* scale;
    int32_t c_in, c_out, k, u, transposed;
    int32_t mf, ck;     /* filled by v2w_fold_plan */
    int32_t _pad;
    float* wf;          /* optional (ABI v28): the folded weight in the plain layout [k][C_in][C_out] as well (what v2w_wn_fold_conv / _convt
                         * write) - a forward that will be back-propagated needs it for the gradient kernels; NULL: not written */
    float* wpd;         /* optional (ABI v28; Conv1d layers with C_in == C_out whose tile configuration has MF == CK only, else must be NULL):
                         * the fragment stream of the layer's INPUT-GRADIENT conv (tap-reversed transpose, what v2w_pack_mfma_dgrad builds
                         * from wf) in the same pass */
} v2w_fold_desc;
int v2w_fold_plan(v2w_fold_desc* descs, int n, int32_t* starts);
int v2w_fold_pack_batch(const v2w_fold_desc* descs_dev, const int32_t* starts_dev, int n,
                        int nblk_scale, int nblk_pack, int lds_bytes, void* stream);

/* ---- K1/K5/K6/K7: fused [per-(b,c) affine] -> leaky_relu -> dilated Conv1d -> +bias [-> +residual]
 * [-> += out] [-> / out_div].  Replaces F.leaky_relu + Conv1d (+ `xt + x`, `xs += ...`, `xs / num_kernels`)
 * of models.py:37-44 (ResBlock1), 65-70 (ResBlock2), 123 (conv_pre), 135-141 (mean over kernels).
 *   in   (B, C_in, L)     in_a/in_s   (B, C_in)  or NULL : x = in_a*in + in_s before the activation
 *                                                          (the CondBN affine of modules.py:25-28 folded into the load)
 *   res  (B, C_out, L) or NULL; res_a/res_s (B, C_out) or NULL : residual = res_a*res + res_s
 *   out  (B, C_out, L);   padding = dil*(k-1)/2 (utils.py:35-36), k odd
 *   slope: leaky_relu negative slope applied to the conv operand (1.0f = none)
 *   accumulate != 0: out = out_old + value ; out_div != 0: out = value / out_div (applied last) */
typedef struct {
    const float* in;   const float* in_a;  const float* in_s;
    const float* wf;   /* [k][C_in][C_out]: read by the direct kernel; may be NULL when algo == V2W_ALGO_MFMA */
    const float* wp;   /* v2w_pack_mfma() form: read by the MFMA kernel; NULL -> direct kernel under V2W_ALGO_AUTO */
    const float* bias;
    const float* res;  const float* res_a; const float* res_s;
    const float* add0; const float* add1;   /* optional extra addends (B, C_out, L): out = (add0 [+ add1]) + value, then / out_div;
                                             * the explicit form of `xs += ...` (models.py:137-141) when the branches ran
                                             * concurrently into separate buffers; exclusive with `accumulate` */
    const float* mask_src; const float* mask_a; const float* mask_s;   /* optional (backward use): the conv result is
                                             * multiplied by lrelu'(mask_a*mask_src + mask_s) = 1 or mask_slope BEFORE bias /
                                             * residual / addends; mask_src (B, C_out, L), mask_a/_s (B, C_out) or NULL */
    float*       out;
    int32_t B, C_in, C_out, L, k, dil;
    float   slope;
    int32_t accumulate;
    float   out_div;
    int32_t algo;
    float   mask_slope;
    int32_t in_stride, in_phase;  /* 0/1, 0: plain.  > 1: the conv reads the de-interleaved phase in[.., in_stride*l + in_phase] of a
                                   * (B, C_in, in_stride*L) tensor (dgrad of a transposed conv, one launch per phase) */
    int32_t pad_left;             /* -1: symmetric padding dil*(k-1)/2.  >= 0: taps sit at offsets -pad_left + t*dil (k may be even) */
    const void*  wps;             /* V2W_ALGO_SPLIT: v2w_pack_split() fragments of this layer; else ignored */
    const float* winv;            /* V2W_ALGO_SPLIT: device pointer to 1/scale of the layer (sc[0] of v2w_pack_split) */
    int32_t in_ct, out_ct;        /* 0: plain.  > 0: `in` / `out` (and res, add, mask_src) point at the first channel of a C_in / C_out
                                   * channel slice of tensors with in_ct / out_ct channels per batch item: one group of a grouped
                                   * Conv1d (models.py:222-227), one launch per group (or 4 per launch through _fwd_multi).
                                   * Not combined with the per-(b, channel) affines; f32 MFMA and direct kernels only */
    float   out_slope;            /* 0 or 1: none.  Else leaky_relu(value, out_slope) on what is stored (applied last): the
                                   * discriminators keep the ACTIVATED feature maps (models.py:184-186, 236-238) */
    int32_t io_bf16;              /* V2W_ALGO_BF16 only (else must be 0): activation STORAGE in bf16 - BASELINE configs[2] priced at 2 bytes
                                   * per activation.  bit 0: `in` is bf16; bit 1: `out`, `res`, `add0`, `add1` are bf16.
                                   * Accumulation, bias, affines and the residual arithmetic stay fp32; one rounding at the store. */
    float*  rowsum_part;          /* optional, masked (mask_src != NULL) f32 MFMA launches with float4-aligned operands only: the launch also
                                   * writes, per position tile and output channel, (sum of the values it stored, 0) to
                                   * rowsum_part[tile][C_out][2] - tiles = v2w_conv1d_tile_config()[9] - for v2w_bn_reduce_partials to add up:
                                   * the bias gradient of the layer whose output gradient this launch produces (backward of models.py:65-70)
                                   * without another pass over the tensor.  V2W_E_ARG when the launch cannot provide it. */
    float*  splitk_ws;
    int64_t splitk_ws_bytes;      /* optional CALLER-OWNED scratch for the f32 MFMA path (ABI v28; the library allocates nothing and keeps no
                                   * state between calls): a launch that cannot fill the chip (<= 128 workgroups: inference at B = 1) is split
                                   * over slices of C_in, every slice stores plain partial sums to its own slab of this buffer and a second
                                   * kernel adds the slabs in slice order and applies bias / residual / addends / division (deterministic).
                                   * v2w_conv1d_splitk_ws_bytes() says how many bytes this launch would use (0: it runs unsplit); NULL, or fewer
                                   * bytes than that: the launch runs unsplit - same values up to the summation order over C_in.  Launches that
                                   * share a buffer must be ordered on ONE stream (a module keeps one buffer per stream it launches on).
                                   * _fwd_multi reads the fields of a[0]. */
} v2w_conv1d_args;
int v2w_conv1d_fwd(const v2w_conv1d_args* a, void* stream);   /* `a` is a HOST pointer, read before return */
/* Bytes of `splitk_ws` the f32 MFMA launch of a[0..n) would use (sizes only are read; 0: no split / another kernel serves it). */
long long v2w_conv1d_splitk_ws_bytes(const v2w_conv1d_args* a, int n);
/* a[0..n) (n <= 4) convs that share B, C_in, C_out, L in ONE launch (MFMA path; V2W_E_SHAPE -> issue them one by one):
 * the residual branches of one generator stage, heaviest first. */
int v2w_conv1d_fwd_multi(const v2w_conv1d_args* a, int n, void* stream);

/* ---- split-f16 weights (V2W_ALGO_SPLIT).  wf [k][C_in][C_out] (folded fp32) -> wps: k*C_in*C_out*4 + 2048 bytes (the tail is padding) holding, per 32-row
 * block, chunk of 32 input channels and tap, the (hi, lo) half-precision MFMA A fragments of scale*w in consumption order;
 * scale = the power of two that puts max|w| of the layer into [8192, 16384) (both halves stay in the normal f16 range).
 * sc: 4 floats of device memory: [0] = 1/scale (the `winv` of v2w_conv1d_args), [1] = scale, [2] = scratch.
 * Activations are split on the fly by the conv kernel and must satisfy |x| <= 65504 (they are clamped there). */
int v2w_split_supported(int c_in, int c_out, int u);   /* 1 when V2W_ALGO_SPLIT serves this layer shape */
int v2w_split_packable(int c_in, int c_out);           /* 1 when v2w_pack_split / _bf16 / _batch accept the shape (C_in % 16; C_out % 32, or C_out == 16 zero-padded to 32 rows) */
int v2w_pack_split(const float* wf, void* wps, float* sc, int k, int c_in, int c_out, void* stream);
int v2w_pack_bf16(const float* wf, void* wps, float* sc, int k, int c_in, int c_out, void* stream);   /* same buffers, V2W_ALGO_BF16 */
/* Batched weight-norm fold + split pack of n Conv1d layers straight from the parameters (weight_v (C_out, C_in, k), weight_g or
 * NULL): three launches for all of them.  descs / starts are DEVICE arrays; starts[0..n] = prefix sums of c_out,
 * starts[n+1..2n+1] = prefix sums of ceil(c_out/32)*(c_in/16); nblk_* = their totals; k_max = largest kernel size. */
typedef struct {
    const float* v; const float* g;   /* weight_v, weight_g (NULL: plain weight) */
    void* wps; float* sc;             /* outputs: as v2w_pack_split */
    float* rowscale;                  /* c_out floats of workspace */
    int32_t c_in, c_out, k;
    int32_t mode;                     /* 0: split f16 (hi, lo) fragments; 1: bf16 fragments for V2W_ALGO_BF16 */
} v2w_split_desc;
/* all_bf16 != 0: every descriptor has mode 1 (no scale record is kept: two launches instead of three) */
int v2w_split_pack_batch(const v2w_split_desc* descs_dev, const int32_t* starts_dev, int n, int nblk_rows, int nblk_pack,
                         int k_max, int all_bf16, void* stream);

/* ---- K6 fused pair for the narrow stages (C == 32 or 16; MFMA path): two chained convs of one residual block in ONE kernel,
 * the intermediate stays in LDS (these layers are HBM-bound as separate launches).
 *   res_mode 0 (ResBlock2, models.py:65-70): t1 = x + conv_{k,dil1}(lrelu(x)) + b1 ; out = t1 + conv_{k,dil2}(lrelu(t1)) + b2
 *   res_mode 1 (ResBlock1 pair, models.py:37-44): t1 = conv_{k,dil1}(lrelu(x)) + b1 ; out = x + conv_{k,dil2}(lrelu(t1)) + b2
 *   x = in_a*in + in_s when the affine is given; then out = ((add0 [+ add1]) + out) [/ out_div] as in v2w_conv1d_args.
 *   wp1 / wp2: v2w_pack_mfma(k, C, C, 1) streams.  a[0..n) (n <= 4) share B, C, L and go out as one launch.
 * Returns V2W_E_SHAPE when C or the receptive field does not fit: run the two convs with v2w_conv1d_fwd instead. */
typedef struct {
    const float* in; const float* in_a; const float* in_s;
    const float* wp1; const float* bias1; const float* wp2; const float* bias2;
    const float* add0; const float* add1;
    float* out;
    int32_t B, C, L, k, dil1, dil2;
    int32_t res_mode;
    float slope, out_div;
} v2w_pair_args;
int v2w_resblock_pair_fwd(const v2w_pair_args* a, int n, void* stream);

/* ---- K6+K7 fused for a narrow ResBlock2 stage (C == 32 or 16): the whole residual section of models.py:135-141,
 *   out = ( sum_j [ t1_j + conv_{k_j,dil2_j}(lrelu(t1_j)) + b2_j ] ) / out_div,  t1_j = x + conv_{k_j,dil1_j}(lrelu(x)) + b1_j,
 * x = in_a*in + in_s, in ONE kernel: x is read once, every t1_j stays in LDS, the branch sum stays in registers and is
 * added in the reference's order.  nk <= 4 branches.  V2W_E_SHAPE -> use the per-branch entry points. */
typedef struct {
    const float* in; const float* in_a; const float* in_s;
    const float* wp1[4]; const float* bias1[4]; const float* wp2[4]; const float* bias2[4];
    int32_t k[4], dil1[4], dil2[4];
    float* out;
    int32_t nk, B, C, L;
    float slope, out_div;
    /* optional fused tail of the generator (ABI v29; v2w_resblock2_stage_fwd with C == 16 only; models.py:143-145): when post_out != NULL the
     * kernel does not write `out` (it may be NULL) but  post_out (B, 1, L) fp32 = tanh(conv_post(leaky_relu(stage output, post_slope)))  with
     * post_w = the folded conv_post weight [post_k][C][1] (v2w_wn_fold_conv), post_b its bias (1 value or NULL), post_k odd <= 9.  The
     * stage's output - 168 MB at BASELINE configs[1] - is neither written nor read back, one launch less. */
    const float* post_w; const float* post_b; float* post_out;
    int32_t post_k;
    float post_slope;
    /* optional INPUT-GRADIENT form (ABI v31; v2w_resblock2_stage_fwd only; the backward of models.py:135-141 for a narrow stage, train.py:214):
     * when bwd_mask2 != NULL the kernel computes, with dr = in_a * in + in_s (the caller passes in = dL/d(out), in_a = 1 / nk, in_s = 0),
     *     bwd_mid[j] (B, C, L) = dt1_j = dr + lrelu'(bwd_mask1[j]) * conv(dr; wp1[j])                  bwd_mask1[j] = the forward's t1_j
     *     out (B, C, L)        = sum_j dt1_j + lrelu'(bwd_mask2_a * bwd_mask2 + bwd_mask2_s) * sum_j conv(dt1_j; wp2[j])      bwd_mask2 = the forward's xr
     * where wp1[j] / wp2[j] are the fragment streams of the TRANSPOSED, tap-reversed weights of conv2_j / conv1_j (v2w_pack_mfma_dgrad),
     * dil1[j] / dil2[j] their dilations (conv2_j's first), lrelu'(v) = v > 0 ? 1 : bwd_slope.  slope must be 1, the biases NULL, out_div 0. */
    const float* bwd_mask1[4]; float* bwd_mid[4];
    const float* bwd_mask2; const float* bwd_mask2_a; const float* bwd_mask2_s;
    float bwd_slope;
    /* bwd_rowsum[j] (optional): v2w_resblock2_stage_bwd_rows(a) rows of [C][2] floats - per (tile, wave) the channel sums of dt1_j over the
     * positions the tile owns (second value 0): the bias gradient of conv1_j after v2w_bn_reduce_partials adds the rows up. */
    float* bwd_rowsum[4];
} v2w_stage_args;
int v2w_resblock2_stage_fwd(const v2w_stage_args* a, void* stream);
int v2w_resblock2_stage_bwd_rows(const v2w_stage_args* a);
/* The same section for an 8-channel stage (the sixth stage of a x640 generator, upsample_rates (5,4,4,2,2,2); ABI v27), fp32 on the
 * vector ALU: C == 8, odd kernel sizes, halos <= 32, nk <= 4.  HERE wp1[j] / wp2[j] are the FOLDED weights [k][C][C] of
 * v2w_wn_fold_conv (there is no fragment stream for 8 channels).  V2W_E_SHAPE otherwise. */
int v2w_resblock2_stage_small_fwd(const v2w_stage_args* a, void* stream);

/* Split-operand counterpart (V2W_ALGO_SPLIT / V2W_ALGO_BF16 arithmetic) for C == 32 or 16 - and, with bf16 != 0 on bf16 tensors
 * (io_bf16 == 3), for the wide stages C == 64 / 128 / 256 as well (csrc/v2w_stage_bf16_wide.hip: x and t1_j resident in LDS, one kernel): wps / sc from v2w_pack_split or
 * v2w_pack_bf16 (or the batch) of the (k, C, C) layers; bf16 != 0 selects the bf16 single-MFMA form.  V2W_E_SHAPE otherwise.
 * Fragment streams: the fp32-tensor kernels (io_bf16 == 0: csrc/v2w_stage_split.hip) walk ONE weight stream - there the 2*nk streams must
 * lie BACK TO BACK in execution order (wps1[0], wps2[0], wps1[1], ...; (C/16)*k units of 2 KiB each, slices of one buffer), V2W_E_ARG if
 * they do not.  The bf16-tensor kernels (io_bf16 == 3: v2w_stage_bf16_wide.hip, v2w_stage_bf16_n16.hip, v2w_stage_bf16.hip) take every
 * stream through its own pointer: any placement. */
typedef struct {
    const float* in; const float* in_a; const float* in_s;
    const void*  wps1[4]; const float* sc1[4]; const float* bias1[4];
    const void*  wps2[4]; const float* sc2[4]; const float* bias2[4];
    int32_t k[4], dil1[4], dil2[4];
    float* out;
    int32_t nk, B, C, L;
    float slope, out_div;
    int32_t bf16;
    int32_t io_bf16;   /* bf16 != 0 only: bit 0 `in` is bf16, bit 1 `out` is bf16 (see v2w_conv1d_args) */
    /* optional fused tail of the generator (bf16 != 0, io_bf16 == 3, C == 16 only; models.py:143-145): when post_out != NULL the kernel does
     * not write `out` (it may be NULL) but  post_out (B, 1, L) fp32 = tanh(conv_post(leaky_relu(stage output, post_slope)))  with
     * post_w = the folded conv_post weight [post_k][C][1] (v2w_wn_fold_conv), post_b its bias (1 value or NULL), post_k odd <= 9 */
    const float* post_w; const float* post_b; float* post_out;
    int32_t post_k;
    float post_slope;
    /* optional fused UPSAMPLER of the next stage (ABI v28; bf16 != 0, io_bf16 == 3, the reference's block set k = (3, 7, 11), dilations (1, 3);
     * models.py:128-129): when up_out != NULL the kernel does not write `out` (it may be NULL) but
     *   up_out (B, C / 2, up_u L) bf16 = ConvTranspose1d(leaky_relu(stage output, up_slope); up_k = 2 up_u taps, stride up_u, padding up_u / 2) + up_bias
     * and, when up_stats_part != NULL, the BatchNorm partial sums of those fp32 values as v2w_convt1d_bf16_fwd writes them
     * ([v2w_resblock2_stage_up_tiles()][C / 2][2], for v2w_bn_reduce_partials).  up_wps: the fragments of v2w_pack_bf16_convt(k = up_k, C, C / 2,
     * up_u).  Served: (C, up_u) = (256, 4), (128, 4), (64, 2), (32, 2); V2W_E_SHAPE otherwise (run the stage and v2w_convt1d_bf16_fwd). */
    const void* up_wps; const float* up_bias; void* up_out; float* up_stats_part;
    int32_t up_k, up_u;
    float up_slope;
    /* ResBlock1 pair mode (ABI v28; rb1 != 0, bf16 != 0, io_bf16 == 3; models.py:37-44 on bf16 tensors): the nk entries are independent PROBLEMS -
     * branch p of a stage at one of its three (dilated conv, conv) pairs - run in one launch:
     *   out_b[p] = ( x_p + conv_{k[p], dil2[p]}(lrelu(conv_{k[p], dil1[p]}(lrelu x_p) + bias1[p])) + bias2[p] ) [ + add0 + add1 ] / out_div ,
     *   x_p = in_a * in_b[p] + in_s (in_b[p] == NULL: `in`; in_a / in_s as above, NULL for the second and third pair).
     * add0 / add1 (bf16 (B, C, L) or NULL): added, add0 + add1 first, by the LAST problem of the launch before the division - the final pair of
     * the last branch takes the other branches' results: ((r0 + r1) + r2) / nk.  `out`, post_*, up_* are not used.  C = 16 .. 256. */
    int32_t rb1;
    const void* in_b[4]; void* out_b[4];
    const void* add0; const void* add1;
} v2w_stage_split_args;
int v2w_resblock2_stage_split_fwd(const v2w_stage_split_args* a, void* stream);
/* Shape query (ABI v28; host-only, nothing is launched or dereferenced): 0 when the call above would run this stage as one kernel, else the
 * code it would return.  Read: B, C, L, nk, k, dil1, dil2, bf16, io_bf16, slope, post_* and the ALIGNMENT of `in` / `out` when they are set
 * (NULL tensor and weight pointers count as aligned). */
int v2w_resblock2_stage_split_config(const v2w_stage_split_args* a);
/* Rows of up_stats_part a call with the fused upsampler (up_u != 0) fills; <= 0: that call would not run fused.  Host-only, as above. */
int v2w_resblock2_stage_up_tiles(const v2w_stage_split_args* a);

/* ---- the residual convolutions of a WIDE ResBlock2 stage (C % 64 == 0) on bf16 tensors (BASELINE configs[2]: bf16 compute / fp32
 * accumulate, bf16 activation storage), one launch per conv position of the block instead of one per branch group
 * (csrc/v2w_conv_bf16_res.hip; models.py:135-141 with ResBlock2.forward, models.py:65-70, inlined):
 *   mode 0  t1_j = x + conv_{k_j,dil_j}(lrelu(x)) + bias_j  for j < nbr, x = in_a * in[0] + in_s (per (b, c); NULL: x = in[0]):
 *           in[0] is read ONCE for all branches, out[j] receives t1_j;
 *   mode 1  out[0] = ( sum_j [ in[j] + conv_{k_j,dil_j}(lrelu(in[j])) + bias_j ] ) / out_div   (out_div == 0: no division):
 *           one accumulator across the branches, nothing but out[0] is written.
 * in / out are bf16 (B, C, L) tensors, 16-byte aligned, L % 4 == 0; wps[j] = the fragments of v2w_pack_bf16 / v2w_split_pack_batch of
 * the (k_j, C, C) layer; k odd, dil * (k - 1) / 2 <= 32.  The residual is rebuilt from the staged (activated, bf16) operand: exact for
 * values >= 0, to 2^-9 relative below.  V2W_E_SHAPE: shape not served (the caller issues the per-layer launches instead). */
typedef struct {
    const void* in[3];
    const float* in_a; const float* in_s;
    const void* wps[3]; const float* bias[3];
    void* out[3];
    int32_t k[3], dil[3];
    int32_t nbr, mode, B, C, L;
    float slope, out_div;
    int32_t _pad;
} v2w_branch_convs_args;
int v2w_branch_convs_bf16_fwd(const v2w_branch_convs_args* a, void* stream);

/* ---- K2: fused leaky_relu -> ConvTranspose1d(k, stride u, padding (k-u)/2) -> +bias
 * (models.py:128-129).  in (B, C_in, L) -> out (B, C_out, L*u); requires (k-u) even and >= 0. */
typedef struct {
    const float* in; const float* wf; const float* wp; const float* bias; float* out;
    float* stats_part;  /* optional (MFMA path only): [ntiles][C_out][2] per-tile (sum, sumsq) of the output, the fused form of
                         * v2w_bn_stats; ntiles = cfg[9] of v2w_convt1d_tile_config; reduce with v2w_bn_reduce_partials */
    int32_t B, C_in, C_out, L, k, u;
    float   slope;
    int32_t algo;
    int32_t io_bf16;    /* v2w_convt1d_bf16_fwd only (else 0): bit 0 `in` is bf16, bit 1 `out` is bf16 (see v2w_conv1d_args) */
    int32_t _pad;
    float*  splitk_ws;  /* v2w_convt1d_fwd only: caller-owned split-over-C_in scratch, as in v2w_conv1d_args (NULL: unsplit) */
    int64_t splitk_ws_bytes;
} v2w_convt1d_args;
int v2w_convt1d_fwd(const v2w_convt1d_args* a, void* stream);
long long v2w_convt1d_splitk_ws_bytes(const v2w_convt1d_args* a);   /* as v2w_conv1d_splitk_ws_bytes */

/* bf16 counterpart (V2W_ALGO_BF16 arithmetic: bf16 operands, fp32 accumulate - BASELINE configs[2]) of v2w_convt1d_fwd for the
 * transposed convs of models.py:128-129.  The transposed conv runs as a Conv1d over UP * C_out virtual output channels on the bf16
 * matrix pipe (csrc/v2w_conv_bf16.hip); `a->wp` must point at the fragments of v2w_pack_bf16_convt (a->wf is ignored), `a->stats_part`
 * (optional) receives [v2w_convt1d_bf16_tiles(a)][C_out][2] partial (sum, sumsq) for v2w_bn_reduce_partials.
 * C_in % 32 == 0, u in 2..8, (k - u) even; V2W_E_SHAPE otherwise. */
int       v2w_pack_bf16_convt(const float* wf, void* wps, int k, int c_in, int c_out, int u, void* stream);
long long v2w_pack_bf16_convt_bytes(int k, int c_in, int c_out, int u);   /* size of `wps`; 0 = unsupported shape */
int       v2w_convt1d_bf16_fwd(const v2w_convt1d_args* a, void* stream);
int       v2w_convt1d_bf16_tiles(const v2w_convt1d_args* a);               /* rows of stats_part (host-only query) */

/* Introspection for profiling: which conv_tile_kernel<MF,U,MI,NI,WM,WN,CK,NPF,RING> instantiation the MFMA path uses for
 * this problem (only the sizes of `a` are read).  0 and cfg[10] filled (cfg[9] = number of position tiles), or V2W_E_SHAPE
 * when the direct kernel would run. */
int v2w_conv1d_tile_config(const v2w_conv1d_args* a, int32_t* cfg);
int v2w_convt1d_tile_config(const v2w_convt1d_args* a, int32_t* cfg);
/* The same for the V2W_ALGO_BF16 kernels: cfg[10] = MI, NI, WM, WN, NPF, EPI, IN_BF, OUT_BF, CK, VEC of the conv_bf16_kernel
 * instantiation a launch of these n (<= 4) problems / this transposed conv would run (sizes, io_bf16 and the alignment of `in` are
 * read; nothing is dereferenced).  V2W_E_SHAPE when the bf16 kernels do not take the shape. */
int v2w_conv1d_bf16_config(const v2w_conv1d_args* a, int n, int32_t* cfg);
int v2w_convt1d_bf16_config(const v2w_convt1d_args* a, int32_t* cfg);

/* ---- K3: per-stage conditioning  z = fcs[i](cat(spk, noise))  (models.py:120,131), legacy
 * spectral_norm power iteration on cbns[i].layer (modules.py:16,24) and [gamma|beta] = (W/sigma) z + b.
 * One call serves every stage (they depend on spk/noise only).  Arrays of n_stages DEVICE pointers are
 * passed by value inside the struct (HOST struct).
 *   spk (B, spk_dim), noise (B, noise_dim); fc_w[i] (128, spk_dim+noise_dim), fc_b[i] (128)
 *   sn_w[i] (2C_i, 128) = weight_orig, sn_b[i] (2C_i), sn_u[i] (2C_i), sn_v[i] (128)
 *   training != 0: one power iteration, u and v are UPDATED IN PLACE (as the reference's hook does)
 *   gb[i] (B, 2C_i): gamma = gb[:, :C], beta = gb[:, C:]   (chunk(2,1), modules.py:24)
 *   z_ws: workspace of n_stages*B*128 floats; sigma_ws: n_stages floats
 *   fc_w[i] == fc_b[i] == NULL: no fcs layer, z = cat(spk, noise) itself (needs spk_dim+noise_dim == 128;
 *   noise may be NULL with noise_dim == 0) - ConditionalBatchNorm1d.forward(inputs, noise) used on its own. */
#define V2W_MAX_STAGES 8
typedef struct {
    const float* spk; const float* noise;
    const float* fc_w[V2W_MAX_STAGES]; const float* fc_b[V2W_MAX_STAGES];
    const float* sn_w[V2W_MAX_STAGES]; const float* sn_b[V2W_MAX_STAGES];
    float* sn_u[V2W_MAX_STAGES]; float* sn_v[V2W_MAX_STAGES];
    float* gb[V2W_MAX_STAGES];
    int32_t C[V2W_MAX_STAGES];
    float* z_ws; float* sigma_ws;
    int32_t n_stages, B, spk_dim, noise_dim, training;
} v2w_cond_args;
int v2w_cond_gamma_beta(const v2w_cond_args* a, void* stream);
/* The spectral-norm half alone (ABI v28): sigma_ws[i] = u_i . (W_i v_i) of every stage (training != 0: after the power iteration, u / v
 * updated in place).  Read: sn_w, sn_u, sn_v, C, n_stages, training, sigma_ws.  In eval mode sigma is a function of the parameters only. */
int v2w_cond_sigma(const v2w_cond_args* a, void* stream);
/* Eval mode (ABI v28), one launch for every stage: from (spk, noise) straight to the folded per-sample affine of the Conditional BatchNorm
 * with its RUNNING statistics (modules.py:20-30 in eval; models.py:131-133):  z = fcs[i](cat(spk, noise)),
 * [gamma | beta] = (W_i z) / sigma_ws[i] + b_i,  a_out[i][b][c] = gamma / sqrt(running_var[i][c] + eps[i]),
 * s_out[i][b][c] = beta - a_out * running_mean[i][c].  c.sigma_ws must hold v2w_cond_sigma's result for the current parameters;
 * c.gb / c.z_ws / c.sn_u / c.sn_v are not used.  Replaces v2w_cond_gamma_beta + n_stages x v2w_bn_finalize(training = 0). */
typedef struct {
    v2w_cond_args c;
    const float* running_mean[V2W_MAX_STAGES]; const float* running_var[V2W_MAX_STAGES];
    float* a_out[V2W_MAX_STAGES]; float* s_out[V2W_MAX_STAGES];
    float eps[V2W_MAX_STAGES];
} v2w_cond_eval_args;
int v2w_cond_affine_eval(const v2w_cond_eval_args* e, void* stream);

/* ---- K4: BatchNorm1d(affine=False) statistics and finalisation (modules.py:14,23).
 * v2w_bn_stats   : x (B,C,L) -> stats[2C] doubles = [sum_c | sumsq_c] over (B,L); deterministic two-level
 *                  reduction; partial_ws >= 2*C*V2W_BN_SPLITS doubles.
 *                  In data-parallel runs the host all-reduces `stats` (RCCL, sum) between the two calls.
 * v2w_bn_finalize: training: mean/biased var from stats (count = stats[2C]), running_mean/var
 *                  momentum update with the unbiased var, num_batches_tracked += 1 (int64);
 *                  eval: running stats are used, nothing is written.
 *                  Then the folded per-sample affine  a[b,c] = gamma*rstd, s[b,c] = beta - gamma*mean*rstd
 *                  (so that CondBN(x) = a*x + s, modules.py:23-28) from gb (B, 2C). */
#define V2W_BN_SPLITS 64
/* stats: 2C+1 doubles = [sum_c | sumsq_c | count]; count = B*L is written by the kernel so that one
 * all-reduce(sum) of the whole array yields the global sums AND the global element count. */
int v2w_bn_stats(const float* x, double* stats, double* partial_ws, int B, int C, int L, void* stream);
/* Fixed-order fp64 reduction of the per-tile partials written by v2w_convt1d_fwd(stats_part) -> stats[2C+1]. */
int v2w_bn_reduce_partials(const float* part, int ntiles, int C, double count, double* stats, void* stream);
int v2w_bn_finalize(const double* stats, const float* gb,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* a_out, float* s_out, int B, int C, int training,
                    float momentum, float eps, void* stream);
/* v2w_bn_reduce_partials + v2w_bn_finalize(training = 1) as ONE launch (ABI v34; one block per channel: its rows added in the same fixed
 * order, `stats` [2C+1] written as v2w_bn_reduce_partials writes it, then the running statistics and (a, s) of that channel): bit-identical
 * to the two calls.  A data-parallel run, which all-reduces `stats` between them, keeps the two calls. */
int v2w_bn_reduce_finalize(const float* part, int ntiles, double count, const float* gb,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           double* stats, float* a_out, float* s_out, int B, int C, float momentum, float eps, void* stream);
/* Two-level form for layers with thousands of partial rows (ABI v28; the fused stage kernels write one row per 224 positions):
 * v2w_bn_reduce_slices adds slice s of the rows of `part` ([ntiles][C][2] floats) into slices[s][2 C] fp64 ([sum | sumsq], nslices <= 1024
 * blocks reading whole rows), v2w_bn_finalize_slices is v2w_bn_finalize(training = 1) on those slices, added in slice order, with the element
 * count given.  Deterministic; same values as the one-level form up to the order of the fp64 additions. */
int v2w_bn_reduce_slices(const float* part, int ntiles, int C, double* slices, int nslices, void* stream);
int v2w_bn_finalize_slices(const double* slices, int nslices, double count, const float* gb,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* a_out, float* s_out, int B, int C, float momentum, float eps, void* stream);

/* ---- K5 (standalone form): out = a[b,c]*x + s[b,c] over (B,C,L).  Only ConditionalBatchNorm1d.forward used on
 * its own (modules.py:20-30) materialises the normalised tensor; Generator.forward folds the affine into its
 * consumers' loads instead. */
int v2w_affine_apply(const float* x, const float* a, const float* s, float* out, int B, int C, int L, void* stream);

/* ---- backward building blocks (the forward entry points above run dgrad: see v2w_wf_transpose_flip).
 * v2w_wgrad: dwf [k][C_in][C_out] = sum_{b,q} lrelu(x_a*x + x_s)[b,ci,q + off_t] * dy[b,co,u*q + r_t]
 *   Conv1d(k, dil): u = 1 ; ConvTranspose1d(k, stride u, pad (k-u)/2): `dil` ignored, dy is (B, C_out, u*Lq).
 *   x (B, C_in, Lq); slab_ws: v2w_wgrad_slabs(...) * k*C_in*C_out floats (per-split partials, summed in fixed order).
 *   V2W_E_SHAPE unless C_in and C_out are multiples of 16. */
int v2w_wgrad_slabs(int B, int c_in, int c_out, int Lq);
int v2w_wgrad(const float* x, const float* x_a, const float* x_s, const float* dy, float* dwf, float* slab_ws,
              int B, int c_in, int c_out, int Lq, int k, int dil, int u, float slope, void* stream);

/* v2w_wgrad_bf16 (ABI v30): the same Conv1d weight gradient from bf16 OPERANDS, fp32 accumulation - what the reference's autocast backward
 * computes (vec2wav/train.py:167,214: `loss_gen_all.backward()` of a forward run under torch.autocast).  act(x) = lrelu(x_a*x + x_s) is
 * evaluated in fp32 and rounded once.  io_bf16 = 0: x and dy are fp32 tensors (rounded while staged); 3: both are bf16 tensors.
 *   slab_ws: v2w_wgrad_bf16_slabs(...) * k*C_in*C_out floats.  V2W_E_SHAPE (and 0 slabs) unless C_in == C_out in {16, 32} or a multiple of 64,
 *   k odd <= 11, Lq % 8 == 0 and the halo (k-1)/2*dil fits the staged tile: the caller runs v2w_wgrad then. */
int v2w_wgrad_bf16_slabs(int B, int c_in, int c_out, int Lq, int k);
int v2w_wgrad_bf16(const void* x, const float* x_a, const float* x_s, const void* dy, float* dwf, float* slab_ws,
                   int B, int c_in, int c_out, int Lq, int k, int dil, float slope, int io_bf16, void* stream);

/* Conditional BatchNorm backward (modules.py:20-30).  Given dx = dL/d(gamma*xhat+beta) and xr (the BN input):
 *   v2w_cbn_bwd_sums : dgb (B, 2C) = [dgamma | dbeta] and csum[2C] fp64 = [sum_b gamma*dbeta | sum_b gamma*dgamma]
 *                      (a data-parallel run all-reduces csum between the two calls); s12_ws: 2*B*C floats.
 *   v2w_cbn_bwd_apply: dxr = A*dx + Bc*xr + Cc  (train: full BatchNorm backward with batch statistics `stats` = the
 *                      forward's [sum|sumsq|count]; eval: dxr = gamma*rstd*dx); tab_ws: B*C + 2C floats. */
int v2w_cbn_bwd_sums(const float* dx, const float* xr, const float* gb, const double* stats,
                     const float* running_mean, const float* running_var, float* s12_ws, float* dgb, double* csum,
                     int B, int C, int L, int training, float eps, void* stream);
int v2w_cbn_bwd_apply(const float* dx, const float* xr, const float* gb, const double* stats, const double* csum,
                      const float* running_mean, const float* running_var, float* tab_ws, float* dxr,
                      int B, int C, int L, int training, float eps, void* stream);
/* tanh + conv_post backward (models.py:143-145): dp = dy*(1-y^2) (dp_ws: B*L floats; its sum is d conv_post.bias),
 * dx = lrelu'(x) * conv_post^T(dp), dwf [k][C_in][1]; part_ws: C_in*k*512 doubles (ABI v32: the one-pass kernel for C_in = 16, k = 7 writes one
 * row of partial weight gradients per persistent workgroup; 64 rows before). */
int v2w_tail_bwd(const float* dy, const float* y, const float* x, const float* wf, float* dp_ws, double* part_ws,
                 float* dx, float* dwf, int B, int C_in, int L, int k, float slope, void* stream);
/* weight-norm backward: (dwf [k][C_in][C_out], v, g) -> dv (v's layout), dg; g == NULL: dv = dw relayouted, dg untouched. */
int v2w_wn_bwd(const float* dwf, const float* v, const float* g, float* dv, float* dg,
               int c_in, int c_out, int k, int transposed, void* stream);
/* conditioning backward of one stage: dgb (B,2C) -> d weight_orig (2C,128), d layer.bias (2C), d fcs.weight (128,D), d fcs.bias;
 * z (B,128) and sigma are the forward's; dz_ws: B*128 + 1 floats. */
int v2w_cond_bwd(const float* dgb, const float* z, const float* sn_w, const float* sn_u, const float* sn_v, const float* sigma,
                 const float* spk, const float* noise, float* d_sn_w, float* d_sn_b, float* d_fc_w, float* d_fc_b,
                 float* dz_ws, int B, int C, int spk_dim, int noise_dim, void* stream);

/* ---- K8: leaky_relu(slope) -> Conv1d(C_in -> 1, k, pad (k-1)/2) -> +bias -> tanh  (models.py:143-145).
 * in (B, C_in, L) -> out (B, 1, L); wf [k][C_in][1]. */
/* the same with a bf16 input tensor (bf16 activation storage of BASELINE configs[2]); output stays fp32 */
int v2w_conv_post_tanh_bf16in(const void* x_bf16, const float* wf, const float* bias, float* out,
                              int B, int c_in, int L, int k, float slope, void* stream);
int v2w_conv_post_tanh(const float* in, const float* wf, const float* bias, float* out,
                       int B, int C_in, int L, int k, float slope, void* stream);

/* ---- mel_spectrogram of the generated audio (vec2wav/dataset.py:53-77, train.py:172-174,266-269).
 * The STFT itself is a Conv1d over the hop-phase de-interleaved signal and runs through v2w_conv1d_fwd (hop input channels,
 * n_fft/hop taps, pad_left = 0, windowed DFT rows as weights: see wavthruvec_pytorch_amd/mel.py); these are its two ends:
 *   v2w_mel_phases: y (B, L) -> xp (B, hop, FP), xp[b][p][f] = reflect_pad(y, pad)[f*hop + p] (0 past the padded signal)
 *   v2w_mel_finish: spec (B, Cs, FP) (rows [0,nb) = Re, [nb,2nb) = Im), basis (n_mels, nb) -> out (B, n_mels, F) =
 *                   log(clamp(basis @ sqrt(Re^2 + Im^2 + 1e-9), 1e-5))                                    (dataset.py:31-41,72-75) */
int v2w_mel_phases(const float* y, float* xp, int B, int L, int hop, int pad, int FP, void* stream);
int v2w_mel_finish(const float* spec, const float* basis, float* out, int B, int Cs, int FP, int F, int nb, int n_mels, void* stream);
/* Their backward (the training loss F.l1_loss(y_mel, mel_spectrogram(y_g_hat)) back-propagates through it, train.py:172-174,204):
 *   v2w_mel_finish_bwd: gout (B, n_mels, F), the forward's spec, basis and basisT (nb, n_mels) -> dspec (B, Cs, FP); the caller
 *                       zero-fills dspec (pad rows and frames >= F are not written).  d log(clamp(x, 1e-5)) = 1/x for x >= 1e-5.
 *   v2w_mel_phases_bwd: dxp (B, hop, FP) -> dy (B, L): every sample sums the padded positions that read it (reflections fold back).
 * Between them the DFT conv's input gradient is v2w_conv1d_fwd with the tap-flipped transposed weights and pad_left = k - 1. */
int v2w_mel_finish_bwd(const float* spec, const float* basis, const float* basisT, const float* gout, float* dspec,
                       int B, int Cs, int FP, int F, int nb, int n_mels, void* stream);
int v2w_mel_phases_bwd(const float* dxp, float* dy, int B, int L, int hop, int pad, int FP, void* stream);

/* ---- MPD / MSD discriminator forwards (models.py:158-275): the memory-bound ends; the convolutions run
 * through v2w_conv1d_fwd (in_ct / out_ct for groups, out_slope for the activated feature maps, dil = period for the (k, 1) Conv2d).
 *   v2w_phase_split: x (B, C, L, inner) -> out (B, s*C, ceil(L/s), inner), out[b][((c/Cg)*s + r)*Cg + c%Cg][u][w] = x[b][c][s*u + r][w]
 *                    (0 past L): a stride-s conv becomes a stride-1 conv over the stacked phases (Cg = channels per group).
 *   v2w_unfold1:     x (B, T) read as (H, inner) rows, reflect-padded on the right to H*inner (models.py:176-181) ->
 *                    out (B, rows, U*inner), out[b][j][u*inner + w] = xpad[(s*u + j - pad)*inner + w], U = (H + 2 pad - k)/s + 1,
 *                    rows >= k (the extra rows are 0): the C_in = 1 layers as 1-tap convs over `rows` channels.
 *   v2w_avgpool4:    AvgPool1d(4, 2, padding=2) (models.py:255-258): x (B, L) -> out (B, L/2 + 1). */
int v2w_phase_split(const float* x, float* out, int B, int C, int Cg, int L, int inner, int s, int ipitch, int opitch, void* stream);
int v2w_unfold1(const float* x, float* out, int B, int T, int H, int inner, int s, int k, int pad, int rows, int opitch, void* stream);
/* Row pitches (floats between consecutive channel rows; 0 = dense): the f32 MFMA kernel's float4 staging needs rows that are
 * multiples of 4 floats, so the discriminators keep (B, C, pitch) buffers with pitch = roundup4(length), run the convs at
 * L = pitch and return the feature maps as [:, :, :length] views.  The tail positions come out of a conv as ordinary
 * positions; v2w_zero_tail restores the zero padding before a stride-1 conv reads that buffer directly. */
int v2w_zero_tail(float* x, long long rows, int pitch, int valid, void* stream);
/* x (B, C, L, inner) -> out (B, k*C, U, inner), out[b][j*C + c][u][w] = x[b][c][s*u + j - pad][w] (0 outside), U = (L + 2 pad - k)/s + 1:
 * a short strided conv (DiscriminatorP: k = 5, stride 3) as one 1-tap conv over k*C channels with the reference's exact MAC count. */
int v2w_unfold_taps(const float* x, float* out, int B, int C, int L, int inner, int s, int k, int pad, int ipitch, int opitch, void* stream);
int v2w_avgpool4(const float* x, float* out, int B, int L, void* stream);

/* ---- discriminator backward (train.py:188-215 differentiates through MPD / MSD in both optimisation steps).  Input gradients of
 * the convs run through v2w_conv1d_fwd with v2w_wf_transpose_flip weights; weight gradients through v2w_wgrad_slice:
 *   v2w_wgrad_slice: v2w_wgrad for a Conv1d whose taps sit at offsets (t - tap0)*dil (tap0 = -1: symmetric) on channel slices:
 *                    x / dy point at the first channel of a c_in / c_out slice of tensors with x_ct / dy_ct channels per batch
 *                    item (0: dense).  slab_ws as for v2w_wgrad.
 *   v2w_disc_dz:     dz = (g + d) * lrelu'(f) over `rows` rows of `pitch` floats ([valid, pitch) = 0): f the ACTIVATED feature map,
 *                    g the gradient that arrived on the returned map (dense rows x valid, or NULL), d the next conv's input
 *                    gradient (pitched, or NULL); slope = 1: no activation (conv_post).
 *   v2w_phase_merge: inverse of v2w_phase_split (gradient of the stacked phases back to (B, C, L, inner)).
 *   v2w_fold1:       backward of v2w_unfold1: dxu (B, rows, ipitch) -> dx (B, T), the reflected tail folded back.
 *   v2w_avgpool4_bwd: backward of v2w_avgpool4: dout (B, L/2 + 1) -> dx (B, L).
 *   v2w_cout1_wgrad: weight gradient of a C_out = 1 conv: dwf [k][C] = sum_{b,l} x[b][c][l + (t - tap0)*dil] * dz[b][l]. */
int v2w_wgrad_slice(const float* x, const float* dy, float* dwf, float* slab_ws, int B, int c_in, int c_out, int Lq,
                    int k, int dil, int tap0, int x_ct, int dy_ct, void* stream);
/* every group of a grouped Conv1d in one launch per tap group (grid.z = group): x (B, G*c_in, Lq), dy (B, G*c_out, Lq),
 * dwf [G][k][c_in][c_out]; slab_ws: G * v2w_wgrad_group_slabs(B, c_in, c_out, Lq, G) * k*c_in*c_out floats (the position splits
 * that fill the GPU are shared out over the G groups of the launch) */
int v2w_wgrad_group_slabs(int B, int c_in, int c_out, int Lq, int ngroups);
int v2w_wgrad_groups(const float* x, const float* dy, float* dwf, float* slab_ws, int B, int c_in, int c_out, int Lq,
                     int k, int dil, int tap0, int ngroups, void* stream);
/* rowsum (optional, one float per row): the sum of each dz row, produced by the same pass; v2w_rowsum_reduce adds them over the
 * batch items (fp64, fixed order) into the bias gradient db (C). */
int v2w_disc_dz(const float* f, const float* g, const float* d, float* dz, float* rowsum, long long rows, int pitch, int valid, float slope,
                void* stream);
int v2w_rowsum_reduce(const float* rowsum, float* db, int B, int C, void* stream);
/* v2w_disc_dz with d in the phase-stacked form dxs (B, s*C, dpitch) of the strided layer above (v2w_phase_merge folded in) */
int v2w_disc_dz_merge(const float* f, const float* g, const float* dxs, float* dz, float* rowsum, int B, int C, int Cg, int L, int inner,
                      int s, int dpitch, int pitch, float slope, void* stream);
int v2w_phase_merge(const float* dxs, float* out, int B, int C, int Cg, int L, int inner, int s, int ipitch, int opitch, void* stream);
int v2w_fold1(const float* dxu, float* dx, int B, int T, int H, int inner, int s, int k, int pad, int rows, int ipitch, void* stream);
int v2w_avgpool4_bwd(const float* dout, float* dx, int B, int L, void* stream);
int v2w_cout1_wgrad(const float* x, const float* dz, float* dwf, int B, int C, int L, int k, int dil, int tap0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VEC2WAV_HIP_H */
