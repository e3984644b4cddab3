// F.dropout(h, p, training=True) of GCNSoftmax.forward (TrainingNeural.py:82) for p > 0.
//
// Every reference configuration trains with dropout = 0.0, so this path is built for correctness, not
// speed: with model->dropout_p > 0 the training entry points run the one-kernel-per-operation sequence with
// three extra launches:  H <- H o mask / (1 - p) in place (this file), Z0 = dinv o (H @ W2) (this file; the
// fused-W2 SpMM epilogue would have seen the undropped H), and W2 / (1 - p) for the backward's
// relu'(H) o dinv o (GY2 W2^T) term.  Because H is stored AFTER the mask, dropped units are exact zeros
// there and the existing backward kernels see relu'(H_dropped) = relu'(H) o mask without knowing about the
// mask; only the 1/(1-p) factor has to be carried (through the scaled copy of W2).
//
// The mask is a counter-based hash of (seed, row, column): independent of the memory layout (row-major or
// slab), of the launch shape and of the batch a graph sits in, reproducible from the seed.  It is NOT
// torch's Philox stream: bit parity with F.dropout's random numbers is unobtainable ("parity unpinned" for
// the mask; the arithmetic around it is property-tested, tests/test_gpu_parity.py).
#include "gmc_common.h"

namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// uniform in [0,1) with 24 bits for element (row, col)
__device__ __forceinline__ float uniform_of(unsigned long long seed, int row, int col) {
    const unsigned long long h = mix64(seed + 0x9E3779B97F4A7C15ULL * ((unsigned long long)(unsigned)row * 4096ULL + (unsigned)col + 1ULL));
    return (float)(unsigned)(h >> 40) * (1.0f / 16777216.0f);
}

struct DropArgs {
    float *H;
    long R;
    int F;       // real columns
    int fs;      // slab width (0: row-major with leading dimension ld)
    long ld;
    float p, keep_inv;
    unsigned long long seed;
};

__global__ __launch_bounds__(256) void dropout_kernel(DropArgs a) {
    const long cols = a.fs ? (long)((a.F + a.fs - 1) / a.fs) * a.fs : a.ld;
    const long total = a.R * cols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long r;
        int c;
        if (a.fs) {  // [slice][R][fs]
            const long per = a.R * a.fs;
            const long s = e / per, rem = e - s * per;
            r = rem / a.fs;
            c = (int)(s * a.fs + rem % a.fs);
        } else {
            r = e / a.ld;
            c = (int)(e - r * a.ld);
        }
        if (c >= a.F) continue;
        const float h = a.H[e];
        a.H[e] = uniform_of(a.seed, (int)r, c) >= a.p ? h * a.keep_inv : 0.0f;
    }
}

struct Hw2Args {
    const float *H, *dinv, *W2;
    float *Z0;
    long R;
    int F, fs;
    long ld;
};

// Z0[r,:] = dinv[r] * (H[r,:] @ W2): one wave per row, fixed-order butterfly sum
__global__ __launch_bounds__(256) void hw2_rows_kernel(Hw2Args a) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int lane = gmc::lane_id();
    float z0 = 0.f, z1 = 0.f, z2 = 0.f;
    for (int c = lane; c < a.F; c += GMC_WAVE) {
        const float h = a.fs ? a.H[(long)(c / a.fs) * a.R * a.fs + r * a.fs + c % a.fs] : a.H[r * a.ld + c];
        z0 = fmaf(h, a.W2[3 * c], z0); z1 = fmaf(h, a.W2[3 * c + 1], z1); z2 = fmaf(h, a.W2[3 * c + 2], z2);
    }
    z0 = gmc::wave_sum(z0); z1 = gmc::wave_sum(z1); z2 = gmc::wave_sum(z2);
    if (lane == 0) {
        const float d = a.dinv[r];
        a.Z0[3 * r] = z0 * d; a.Z0[3 * r + 1] = z1 * d; a.Z0[3 * r + 2] = z2 * d;
    }
}

__global__ void scale_copy_kernel(const float *src, float *dst, int n, float s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * s;
}

}  // namespace

int gmc_dropout_launch(float *H, long R, int F, int fs, long ld, float p, unsigned long long seed, hipStream_t st) {
    if (R == 0) return GMC_OK;
    DropArgs a{H, R, F, fs, ld, p, 1.0f / (1.0f - p), seed};
    const long cols = fs ? (long)((F + fs - 1) / fs) * fs : ld;
    const long blocks = (R * cols + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st, a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

int gmc_hw2_rows_launch(const float *H, const float *dinv, const float *W2, float *Z0, long R, int F, int fs, long ld,
                        hipStream_t st) {
    if (R == 0) return GMC_OK;
    Hw2Args a{H, dinv, W2, Z0, R, F, fs, ld};
    hipLaunchKernelGGL(hw2_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

int gmc_scale_copy_launch(const float *src, float *dst, int n, float s, hipStream_t st) {
    hipLaunchKernelGGL(scale_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, n, s);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
