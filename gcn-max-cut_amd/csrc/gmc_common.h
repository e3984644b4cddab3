// Shared device helpers for the gfx950 kernels of libgcnmaxcut_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gcnmaxcut.h"

#define GMC_WAVE 64

#define GMC_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Optional per-kernel timing (bench.py): hipEvents recorded around each launch of the fused
// step on the caller's stream.  Off by default (no events, capture-safe).
void gmc_probe_mark(int tag, bool begin, hipStream_t st);
struct GmcProbeScope {
    int tag; hipStream_t st;
    GmcProbeScope(int t, hipStream_t s) : tag(t), st(s) { gmc_probe_mark(tag, true, st); }
    ~GmcProbeScope() { gmc_probe_mark(tag, false, st); }
};

// internal form of gmc_spmm_f32 with a probe tag (variant 1 = shared-source W1 gather)
int gmc_spmm_launch(const int32_t *rowptr, const int32_t *col, const float *vals, const float *scale,
                    const float *X, int64_t ldx, const float *bias, int relu, float *Y, int64_t ldy,
                    int32_t n_rows, int32_t F, int32_t group_rows, const float *W2, float *Z0,
                    int tag, hipStream_t st);

static inline bool gmc_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

namespace gmc {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (GMC_WAVE - 1)); }

// wave-uniform value -> SGPR so that dependent loads become scalar loads
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Butterfly sum over the 64 lanes; every lane gets the total.  Fixed order => deterministic.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, GMC_WAVE);
    return v;
}

// max(x, 0) as ONE v_max_f32 (the C++ forms compile to a canonicalising v_max x,x plus the max)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
// max(x, lo) as one v_max_f32: lo = 0 is relu, lo = -inf the identity (no branch on a runtime flag)
__device__ __forceinline__ float max1(float x, float lo) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo));
    return r;
}
// Packed fp32 (v_pk_add/mul/fma_f32: two floats per lane in ONE 4-cycle VALU slot).  The LDS-tiled
// kernels are instruction-issue bound, so their per-row arithmetic is written on 2-vectors explicitly
// rather than left to the SLP vectoriser (which scalarised gather #2 of the fused forward).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f splat2(float s) { return (v2f)(s); }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v4f f4v(const float4 &a) { return (v4f){a.x, a.y, a.z, a.w}; }
__device__ __forceinline__ float4 v4f_f4(const v4f &a) { return make_float4(a.x, a.y, a.z, a.w); }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_add(float4 &a, const float4 &b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}
__device__ __forceinline__ void f4_fma(float4 &a, float s, const float4 &b) {
    a.x = fmaf(s, b.x, a.x); a.y = fmaf(s, b.y, a.y); a.z = fmaf(s, b.z, a.z); a.w = fmaf(s, b.w, a.w);
}

// W1 slab copy (gmc_model.W1_slab): element (row, col) of the [N,F] table lives at ((col/16)*N + row)*16 + col%16,
// i.e. [ceil(F/16)][N][16] - the 64 B a tile row of a 16-column slice needs are contiguous with the next row's.
__host__ __device__ __forceinline__ long slab16_index(long row, int col, int N) {
    return ((long)(col >> 4) * N + row) * 16 + (col & 15);
}

}  // namespace gmc
