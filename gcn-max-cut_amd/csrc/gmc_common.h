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
// does the batch carry overflow lists the LDS-tiled kernels have to walk?  (ovf_ptr without a single block - the host
// says so through ovf_max_blocks, the pointers are device memory - is a batch without lists: the OVF kernels read
// block 0 of the batch unconditionally for a graph that has none.)
static inline bool gmc_has_overflow(const gmc_batch *b) { return b->ovf_ptr != nullptr && b->ovf_ids != nullptr && b->ovf_max_blocks > 0; }

namespace gmc {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (GMC_WAVE - 1)); }

// wave-uniform value -> SGPR so that dependent loads become scalar loads
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// v[lane] + v[lane ^ O] on the VALU: `__shfl_xor` is a ds_bpermute_b32 - a trip through the LDS pipe the gathers of the
// tiled kernels live on, ~100 cycles of latency each in the latency-bound ones.  gfx950: v_permlane32_swap /
// v_permlane16_swap exchange half waves / neighbouring rows of 16, DPP covers the distances inside a row (row_ror:8 IS
// xor 8; xor 4 = row_shl:4 for lanes 0-3 / 8-11 of a row, row_shr:4 for the others, chosen by the bank mask).  The add
// is commutative, so every lane gets bitwise what `v + __shfl_xor(v, O)` gave it.
template <int O>
__device__ __forceinline__ float xor_add(float v) {
    static_assert(O == 32 || O == 16 || O == 8 || O == 4 || O == 2 || O == 1, "lane distance");
    const unsigned u = __float_as_uint(v);
    if constexpr (O == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (O == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (O == 4) {
        unsigned t = __builtin_amdgcn_update_dpp(u, u, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
        t = __builtin_amdgcn_update_dpp(t, u, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
        return v + __uint_as_float(t);
    } else {
        constexpr int ctrl = O == 8 ? 0x128 /* row_ror:8 */ : O == 2 ? 0x4e /* quad_perm:[2,3,0,1] */ : 0xb1 /* [1,0,3,2] */;
        return v + __uint_as_float(__builtin_amdgcn_update_dpp(0u, u, ctrl, 0xf, 0xf, false));
    }
}
// the butterfly steps of distance HI, HI/2, ..., LO (powers of two): sum over the lanes that differ in those bits
template <int HI, int LO>
__device__ __forceinline__ float xor_tree(float v) {
    if constexpr (HI >= LO && HI >= 1) {
        v = xor_add<HI>(v);
        if constexpr (HI / 2 >= LO && HI / 2 >= 1) v = xor_tree<HI / 2, LO>(v);
    }
    return v;
}

// Butterfly sum over the 64 lanes; every lane gets the total.  Fixed order => deterministic.
__device__ __forceinline__ float wave_sum(float v) { return xor_tree<32, 1>(v); }

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
