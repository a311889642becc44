// Shared device helpers for the Vec2Wav gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cxxabi.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/vec2wav_hip.h"

#define V2W_WAVE 64  // CDNA wavefront width (hard-coded: warpSize folds to 64 on gfx950)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static inline int v2w_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Host-only "would this launch be taken?" queries run the launcher itself up to the launch with this sentinel for a stream: every shape
// check the real call makes is made, nothing is launched, no attribute is set, no pointer is dereferenced.
#define V2W_DRY_STREAM (reinterpret_cast<hipStream_t>(static_cast<intptr_t>(-1)))
static inline bool v2w_dry(hipStream_t s) { return s == V2W_DRY_STREAM; }

// Name sink (include/vec2wav_hip.h, ABI v33): a "stream" with bit 0 set that is not the dry-run sentinel points at the caller's v2w_name_sink.
// A call made with it goes as far as a real call - every launch site is V2W_LAUNCH, every attribute change V2W_MAX_LDS - and appends the
// demangled name of each kernel it reaches instead of launching it.  The runtime maps the host stub to the kernel's symbol
// (hipKernelNameRefByPtr: a table lookup, no device needed), so no launch site spells a name.
static inline v2w_name_sink* v2w_sink(hipStream_t s) {
    const uintptr_t u = reinterpret_cast<uintptr_t>(s);
    if (!(u & 1) || s == V2W_DRY_STREAM) return nullptr;
    v2w_name_sink* k = reinterpret_cast<v2w_name_sink*>(u & ~static_cast<uintptr_t>(1));
    return k->magic == V2W_NAME_SINK_MAGIC ? k : nullptr;
}
static inline void v2w_sink_add(v2w_name_sink* k, const void* host_fn) {
    const char* m = hipKernelNameRefByPtr(host_fn, nullptr);
    int st = 0;
    char* d = m ? abi::__cxa_demangle(m, nullptr, nullptr, &st) : nullptr;
    const char* nm = d ? d : (m ? m : "?");
    const int n = (int)strlen(nm);
    if (k->buf && k->cap > 0 && k->len + n + 2 <= k->cap) {
        memcpy(k->buf + k->len, nm, n);
        k->buf[k->len + n] = '\n';
        k->len += n + 1;
        k->buf[k->len] = 0;
    }
    free(d);
}
#define V2W_LAUNCH(kern, grid, block, lds, stream, ...)                                                          \
    do {                                                                                                         \
        if (v2w_name_sink* v2w_ns_ = v2w_sink(stream)) v2w_sink_add(v2w_ns_, reinterpret_cast<const void*>(kern)); \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                    \
    } while (0)
// hipFuncAttributeMaxDynamicSharedMemorySize for a launch that needs more than 64 KB of LDS (set at every such launch: the library keeps no
// per-device state to remember it in); nothing to set for a name sink
static inline hipError_t v2w_max_lds(const void* host_fn, int lds, hipStream_t s) {
    return v2w_sink(s) ? hipSuccess : hipFuncSetAttribute(host_fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// Compute units of the CURRENT device (the device the caller's stream belongs to: every entry point runs with it made current).
// Asked of the runtime at every launch of a persistent kernel - no process-wide cache: a value remembered from the first caller's
// device would size the grids of every other device in the process.
static inline int v2w_num_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}

__device__ __forceinline__ float v2w_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// v / d with r = 1.f / d precomputed (one true division per thread instead of one per element): q = v*r is within one ulp of the
// quotient and r is the correctly rounded reciprocal, so one Newton step on the exact fma residual yields the correctly rounded
// quotient (Markstein) - the same bits as the division - in three VALU instructions instead of ~10.  (Denormal quotients excepted;
// the path's values are O(1).)
__device__ __forceinline__ float v2w_div_by(float v, float d, float r) {
    const float q = v * r;
    return fmaf(fmaf(-q, d, v), r, q);
}

// Sum over the 64 lanes of a wave (all lanes receive the total).
__device__ __forceinline__ double v2w_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float v2w_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum for blocks of up to 1024 threads; `red` is >= 16 elements of LDS scratch.
template <typename T>
__device__ __forceinline__ T v2w_block_sum(T v, T* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = v2w_wave_sum(v);
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    T t = 0;
    for (int i = 0; i < nw; ++i) t += red[i];  // fixed order: deterministic
    return t;
}

// ---- f32 MFMA fragments shared by the tile kernels.  A/B operands are ONE float per lane; for the 32x32x2 shape lane l
// holds A[row l&31][k l>>5] / B[k l>>5][col l&31], for 16x16x4 A[l&15][l>>4] / B[l>>4][l&15].
template <int MF> struct Frag;
template <> struct Frag<32> {
    typedef f32x16 acc_t;
    static constexpr int NREG = 16, KSTEP = 2;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    __device__ static __forceinline__ int row(int reg, int hk) { return (reg & 3) + 8 * (reg >> 2) + 4 * hk; }
};
template <> struct Frag<16> {
    typedef f32x4 acc_t;
    static constexpr int NREG = 4, KSTEP = 4;
    __device__ static __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D map: col = lane&15, row = (lane>>4)*4 + reg
    __device__ static __forceinline__ int row(int reg, int hk) { return hk * 4 + reg; }
};


// ---- in-kernel timeline, DIAGNOSTIC builds only (-DV2W_TIMELINE, built and read by tools/stage_timeline.py / tools/tile_timeline.py):
// lane 0 of every wave stores s_memtime stamps into a side buffer that no kernel reads.  In the product build V2W_STAMP compiles to
// nothing and the setters do not exist.  Each instrumented source file owns its buffer pointer (no relocatable device code needed)
// and exports its setter with V2W_TL_SETTER(name).
#ifdef V2W_TIMELINE
#define V2W_TL_SLOTS 32
static __device__ unsigned long long* v2w_tl_buf;      // [blocks][4 waves][V2W_TL_SLOTS]
static __device__ int v2w_tl_blocks;
#define V2W_TL_SETTER(name)                                                                          \
    extern "C" int name(void* buf, int nblocks) {                                                    \
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(v2w_tl_buf), &buf, sizeof(buf));                  \
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(v2w_tl_blocks), &nblocks, sizeof(nblocks)); \
        return (int)e;                                                                               \
    }
__device__ __forceinline__ void v2w_tl_stamp(int k) {
    if ((threadIdx.x & 63) == 0 && (int)blockIdx.x < v2w_tl_blocks && (threadIdx.x >> 6) < 4 && k < V2W_TL_SLOTS - 3) {
        unsigned long long* dst = v2w_tl_buf + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * V2W_TL_SLOTS;
        dst[k] = __builtin_amdgcn_s_memtime();
        if (k == 0) {
            dst[V2W_TL_SLOTS - 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
            dst[V2W_TL_SLOTS - 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
            dst[V2W_TL_SLOTS - 3] = __builtin_amdgcn_s_memrealtime();                 // 100 MHz: comparable across CUs
        }
    }
}
#define V2W_STAMP(k) v2w_tl_stamp(k)
#else
#define V2W_STAMP(k) ((void)0)
#endif
