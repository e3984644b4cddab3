// Slab copy of conv1.weight for the fused forward (gmc_model.W1_slab).
//
// The fused layer-1 forward streams the W1 columns of one 16-wide slice into LDS per (graph, slice) tile.  From
// the DGL layout [N,F] a wave's LDS-DMA instruction (16 rows x 64 B) is 16 pieces at a 4F-byte stride - half of
// every 128-byte line, twice the requests; from the slab layout [ceil(F/16)][N][16] it is one contiguous KiB.
// The copy is 2 MB for the reference model (1000 x 500): built here once, then kept current by the kernels that
// update W1 (finish.hip, adam.hip).
#include "gmc_common.h"

namespace {
__global__ __launch_bounds__(256) void w1_slab_kernel(const float *W1, float *slab, int N, int F, long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 #i of the slab
    if (i >= total4) return;
    const long e = i * 4;
    const int c = (int)(e & 15), r = (int)((e >> 4) % N), s = (int)((e >> 4) / N);
    const int col = s * 16 + c;
    float4 v = gmc::f4_zero();                                    // pad columns (F % 4 == 0: all four or none)
    if (col < F) v = *reinterpret_cast<const float4 *>(W1 + (long)r * F + col);
    reinterpret_cast<float4 *>(slab)[i] = v;
}
}  // namespace

extern "C" size_t gmc_w1_slab_floats(int32_t N, int32_t F) {
    if (N <= 0 || F <= 0) return 0;
    return (size_t)((F + 15) / 16) * (size_t)N * 16;
}

extern "C" int gmc_w1_slab_f32(const float *W1, int32_t N, int32_t F, float *slab, gmc_stream_t stream) {
    if (!W1 || !slab) return GMC_ERR_NULL;
    if (N <= 0 || F <= 0 || F % 4) return GMC_ERR_SHAPE;
    if (!gmc_aligned16(W1) || !gmc_aligned16(slab)) return GMC_ERR_ALIGN;
    const long total4 = (long)gmc_w1_slab_floats(N, F) / 4;
    hipLaunchKernelGGL(w1_slab_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), W1, slab, N, F, total4);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

namespace {
__global__ __launch_bounds__(64) void publish_kernel(const float *src, int n, float *dst) {
    for (int i = threadIdx.x; i < n; i += 64) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

// n floats of device memory -> pinned host memory (its device-side address), one system-scope store each: what a
// data-parallel rank does with the all-reduced loss of a step (the slot after the gradient), so that the host
// has it while Adam is still running.  One tiny launch instead of a copy-engine transfer.
extern "C" int gmc_publish_f32(const float *src, int32_t n, float *pinned_dst, gmc_stream_t stream) {
    if (!src || !pinned_dst) return GMC_ERR_NULL;
    if (n < 0) return GMC_ERR_SHAPE;
    if (n == 0) return GMC_OK;
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), src, (int)n, pinned_dst);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
