// Arguments of the tile kernels (f32 MFMA: v2w_conv_mfma.hip, split-f16 MFMA: v2w_conv_split.hip).  Host code fills
// TileArgs from the C-ABI structs; several problems of identical tile configuration travel in one MultiArgs.
#pragma once
#include "v2w_common.h"

namespace {

struct TileArgs {
    const float* in; const float* in_a; const float* in_s;
    const float* wp; const float* bias;
    const void* wps; const float* winv;      // split-f16 kernel: packed (hi, lo) half fragments and 1/scale of the layer
    const float* res; const float* res_a; const float* res_s;
    const float* add0; const float* add1;   // conv only, optional extra addends: out = ((add0 [+ add1]) + value)
    const float* mask_src; const float* mask_a; const float* mask_s;   // conv only: acc *= lrelu'(mask_a*mask_src + mask_s)
    float mask_slope;
    int in_stride, in_phase;   // the conv reads in[.., in_stride*pos + in_phase] (de-interleaved phase of a longer sequence)
    float* out;
    float* stats_part;   // convT only, optional: [ntiles][Cout][2] per-tile (sum, sumsq) of the output for BatchNorm
    int B, Cin, Cout, L, K, dil;
    int CinT, CoutT;   // channel counts of the tensors `in` / `out` (res, add, mask) live in: > Cin / Cout for one group of a grouped conv
    float out_slope;   // MASK instantiations only: leaky_relu on the stored value (1 = none)
    int pad;      // convT only: (K-U)/2
    int hl, hr;   // halo (input positions) left / right of the tile
    int hla;      // hl rounded up to a multiple of 4: LDS column 0 <-> position n0 - hla (16-B aligned rows)
    int xw;       // split kernel: unused (0)
    int xcols;    // split kernel: positions actually staged (multiple of 4)
    int xrows;    // f32 kernel: position rows of the LDS signal tile (multiple of 4): hla + NT + hr rounded up
    int vec4;     // 1: L % 4 == 0 and 16-B aligned base -> float4 staging
    int evec;     // split kernel: 1 = float4 epilogue (L % 4 == 0, every epilogue operand 16-B aligned)
    int atab_off; // LDS offset (floats) of the affine table: after the 1 or 2 signal buffers
    int ntl;      // position tiles per batch item
    int ntiles;   // B * ntl
    float slope;
    int accumulate;
    float out_div;
    int io_bf16;      // bf16 kernels: bit 0 `in` is bf16, bit 1 `out` / `res` / `add0` / `add1` are bf16 (pointers are typed float* regardless)
    int up_u, up_p;   // bf16 transposed conv run as a 3-tap conv over up_p * C_out virtual rows (row = co * up_p + phase): stride, padded phase count
    int* cfg_out; // host-only: when set, launch_tile reports its template configuration instead of launching
};

#define V2W_MAX_MULTI 4
// Up to V2W_MAX_MULTI problems of identical tile configuration in one launch (the residual branches of a stage):
// blocks [start[q], start[q+1]) belong to problem q; heaviest problem first so the tail of the launch is made of light tiles.
struct MultiArgs {
    TileArgs p[V2W_MAX_MULTI];
    int start[V2W_MAX_MULTI + 1];
};

}  // namespace
