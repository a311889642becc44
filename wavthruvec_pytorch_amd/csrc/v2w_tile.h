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
    int ksplit;       // f32 tile kernel: > 1 = the launch is the partial pass of a split over C_in chunks: slice s of the grid accumulates chunks
                      // [s, s + 1) * (C_in / CK / ksplit) and stores its plain sums to out + s * B * CoutT * L * U (see launch_tile)
    int up_u, up_p;   // bf16 transposed conv run as a 3-tap conv over up_p * C_out virtual rows (row = co * up_p + phase): stride, padded phase count
    int* cfg_out; // host-only: when set, launch_tile reports its template configuration instead of launching
    float* splitk_ws; long long splitk_ws_bytes;   // host-only (f32 tile kernel): the caller's split-over-C_in scratch (v2w_conv1d_args::splitk_ws)
    long long* ws_query;                           // host-only: when set, launch_tile reports the bytes of splitk_ws it would use instead of launching
};

// The tile kernels pick their problem with a per-workgroup index into MultiArgs::p, so every `p.field` is a scalar load from the
// kernel-argument segment - which the register allocator treats as free to repeat: short of SGPRs it re-issues the load at each use,
// inside the staging and epilogue loops, each time followed by an s_waitcnt lgkmcnt(0) that also drains the wave's LDS queue
// (conv_bf16_kernel: 341 s_loads, 12 of them serialised in front of the 12 loads of a chunk prefetch).  pinned_tile_args() returns a
// copy whose hot fields went through v_readfirstlane and therefore live in registers (SGPRs, or VGPR lanes when those run out).
// A pointer rebuilt from its two halves is a GENERIC pointer to the compiler (flat_load: counts against vmcnt AND lgkmcnt); every
// access through a pinned pointer goes through gptr<T>(), which names the global address space again.
// (v_readfirstlane of the loaded value: its result is a scalar register the compiler neither re-materialises nor - unlike the
// result of an opaque asm - treats as possibly divergent)
__device__ __forceinline__ void pin_s(int& v) { v = __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void pin_s(float& v) { v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
template <typename T> __device__ __forceinline__ void pin_s(T*& v) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    v = reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
template <typename T> using global_ptr = T __attribute__((address_space(1)))*;
template <typename T, typename U> __device__ __forceinline__ global_ptr<T> gptr(U* q) { return (global_ptr<T>)q; }
__device__ __forceinline__ TileArgs pinned_tile_args(const TileArgs& g) {
    TileArgs p = g;
    pin_s(p.in); pin_s(p.in_a); pin_s(p.wp); pin_s(p.wps); pin_s(p.res); pin_s(p.add0); pin_s(p.add1); pin_s(p.mask_src); pin_s(p.out);
    pin_s(p.stats_part);
    pin_s(p.Cin); pin_s(p.Cout); pin_s(p.L); pin_s(p.K); pin_s(p.dil); pin_s(p.CinT); pin_s(p.CoutT); pin_s(p.in_stride); pin_s(p.in_phase);
    pin_s(p.hl); pin_s(p.hla); pin_s(p.xrows); pin_s(p.xcols); pin_s(p.vec4); pin_s(p.evec); pin_s(p.atab_off); pin_s(p.ntl); pin_s(p.ntiles);
    pin_s(p.accumulate); pin_s(p.up_u); pin_s(p.up_p); pin_s(p.pad);
    pin_s(p.slope); pin_s(p.out_div); pin_s(p.mask_slope); pin_s(p.out_slope);
    return p;
}

#define V2W_MAX_MULTI 4
// Up to V2W_MAX_MULTI problems of identical tile configuration in one launch (the residual branches of a stage):
// blocks [start[q], start[q+1]) belong to problem q; heaviest problem first so the tail of the launch is made of light tiles.
struct MultiArgs {
    TileArgs p[V2W_MAX_MULTI];
    int start[V2W_MAX_MULTI + 1];
};

}  // namespace
