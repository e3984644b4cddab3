// Backward through conv2's feature transform and the relu, plus the column reductions.
//
// Replaces (autograd of TrainingNeural.py:81-83, run by loss.backward() :385):
//   dW2  = (H o dinv)^T @ GY2                     [F,3]
//   Gpre = relu'(H) o dinv o (GY2 @ W2^T)         [R,F]   (gradient w.r.t. conv1's output)
//   db1  = colsum(Gpre)                           [F]
//   Gs   = dinv o Gpre                            [R,F]   (operand of conv1's backward SpMM)
// One streaming pass over H: each thread owns 4 columns (float4, 2 KiB-coalesced rows) and
// walks a tile of rows keeping its dW2/db1 partials in registers; tile partials go to a
// small scratch and are folded in fixed order by colsum_reduce (no float atomics).
#include "gmc_common.h"

namespace {

constexpr int kTileRows = 64;   // rows per workgroup
constexpr int kColThreads = 128;  // x float4 = 512 columns per grid.y slice
constexpr int kRowLanes = 2;

struct HiddenArgs {
    const float *H;
    long ldh;
    const float *GY2;
    const float *W2;
    const float *dinv;
    float *Gs;
    long ldg;
    float *part;  // [tiles][F][4] = (dW2[f,0..2], db1[f])
    int R;
    int F;
};

__global__ __launch_bounds__(kColThreads * kRowLanes) void hidden_bwd_kernel(HiddenArgs a) {
    __shared__ float4 red[4][kColThreads];
    const int ct = threadIdx.x & (kColThreads - 1);
    const int rl = gmc::uniform((int)(threadIdx.x / kColThreads));
    const int c4 = blockIdx.y * kColThreads + ct;  // float4 column index
    const int F4 = a.F >> 2;
    const bool on = c4 < F4;
    const int cl = on ? c4 : F4 - 1;
    float w[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) w[j][k] = a.W2[((long)cl * 4 + j) * 3 + k];

    float dw[4][3] = {};
    float db[4] = {};
    const int rbeg = blockIdx.x * kTileRows;
    const int rend = min(rbeg + kTileRows, a.R);
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += kRowLanes) {
        const float4 h = reinterpret_cast<const float4 *>(a.H + (long)r * a.ldh)[cl];
        const float d = a.dinv[r];
        const float g0 = a.GY2[(long)r * 4], g1 = a.GY2[(long)r * 4 + 1], g2 = a.GY2[(long)r * 4 + 2];
        const float hv[4] = {h.x, h.y, h.z, h.w};
        float gs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gh = g0 * w[j][0] + g1 * w[j][1] + g2 * w[j][2];
            const float gpre = hv[j] > 0.f ? gh * d : 0.f;
            gs[j] = gpre * d;
            db[j] += gpre;
            const float hd = hv[j] * d;
            dw[j][0] = fmaf(hd, g0, dw[j][0]);
            dw[j][1] = fmaf(hd, g1, dw[j][1]);
            dw[j][2] = fmaf(hd, g2, dw[j][2]);
        }
        if (on) reinterpret_cast<float4 *>(a.Gs + (long)r * a.ldg)[c4] = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    // fold the row lanes (fixed order), then one 64 B store per thread
    if (rl == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[j][ct] = make_float4(dw[j][0], dw[j][1], dw[j][2], db[j]);
    }
    __syncthreads();
    if (rl == 0 && on) {
        float4 *out = reinterpret_cast<float4 *>(a.part) + ((long)blockIdx.x * a.F + (long)c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 o = red[j][ct];
            out[j] = make_float4(dw[j][0] + o.x, dw[j][1] + o.y, dw[j][2] + o.z, db[j] + o.w);
        }
    }
}

// Same pass over the slab layout [slice][R][FS] the LDS-tiled kernels use: a block owns
// (row tile, slice); thread (row lane, q) streams FS-wide rows contiguously (a 256-thread pass
// covers 256/Q full slab rows = 4 KiB contiguous), keeps the partials of its 4 columns in
// registers, then the lanes/waves sharing a q are folded in a fixed order.
constexpr int kSlabTileRows = 256;

struct HiddenSlabArgs {
    const float *H;
    const float *GY2;
    const float *W2;
    const float *dinv;
    float *Gs;
    float *part;  // [tiles][F][4]
    int R;
    int F;
};

template <int FS>
__global__ __launch_bounds__(256) void hidden_bwd_slab_kernel(HiddenSlabArgs a) {
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = 256 / Q;
    __shared__ float red[4][Q][16];
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int s = blockIdx.y;
    const int f0 = s * FS + 4 * q;
    const bool on = f0 < a.F;
    float w[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) w[j][k] = on ? a.W2[(long)(f0 + j) * 3 + k] : 0.f;
    float acc[16] = {};  // [j][0..2] dW2, [j][3] db1
    const long slab = (long)s * a.R * FS;
    const int rbeg = blockIdx.x * kSlabTileRows;
    const int rend = min(rbeg + kSlabTileRows, a.R);
#pragma unroll 4
    for (int r = rbeg + rl; r < rend; r += kRowsPerPass) {
        const long off = slab + (long)r * FS + 4 * q;
        const float4 h = *reinterpret_cast<const float4 *>(a.H + off);
        const float d = a.dinv[r];
        const float g0 = a.GY2[(long)r * 4], g1 = a.GY2[(long)r * 4 + 1], g2 = a.GY2[(long)r * 4 + 2];
        const float hv[4] = {h.x, h.y, h.z, h.w};
        float gs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gh = g0 * w[j][0] + g1 * w[j][1] + g2 * w[j][2];
            const float gpre = (on && hv[j] > 0.f) ? gh * d : 0.f;
            gs[j] = gpre * d;
            const float hd = on ? hv[j] * d : 0.f;
            acc[4 * j + 0] = fmaf(hd, g0, acc[4 * j + 0]);
            acc[4 * j + 1] = fmaf(hd, g1, acc[4 * j + 1]);
            acc[4 * j + 2] = fmaf(hd, g2, acc[4 * j + 2]);
            acc[4 * j + 3] += gpre;
        }
        *reinterpret_cast<float4 *>(a.Gs + off) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    // fold lanes with equal q inside the wave (xor over the lane bits above log2 Q), then waves
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = gmc::xor_tree<32, Q>(acc[i]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < Q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][lane][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < Q * 4 && s * FS + 4 * (threadIdx.x / 4) < a.F) {  // thread = (q, j)
        const int qq = threadIdx.x / 4, j = threadIdx.x % 4;
        float4 o;
        float *op = reinterpret_cast<float *>(&o);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            op[c] = ((red[0][qq][4 * j + c] + red[1][qq][4 * j + c]) + red[2][qq][4 * j + c]) + red[3][qq][4 * j + c];
        reinterpret_cast<float4 *>(a.part)[(long)blockIdx.x * a.F + s * FS + 4 * qq + j] = o;
    }
}

// dW2[f,k], db1[f] = sum over tiles (ascending) of part[tile][f][:]; db2 = sum over graphs.
struct ReduceArgs {
    const float *part;
    int tiles;
    int F;
    float *dW2;
    float *db1;
    const float *db2part;
    int B;
    float *db2;
};

constexpr int kRedCols = 16;
constexpr int kRedLanes = 64;

__global__ __launch_bounds__(kRedCols * kRedLanes) void colsum_reduce_kernel(ReduceArgs a) {
    __shared__ float4 red[kRedLanes][kRedCols];
    const int cl = threadIdx.x & (kRedCols - 1);
    const int tl = threadIdx.x / kRedCols;
    if ((int)blockIdx.x == (a.F + kRedCols - 1) / kRedCols) {  // extra block: db2
        if (threadIdx.x < 3) {
            float s = 0.f;
            for (int g = 0; g < a.B; ++g) s += a.db2part[g * 3 + threadIdx.x];
            a.db2[threadIdx.x] = s;
        }
        return;
    }
    const int f = blockIdx.x * kRedCols + cl;
    float4 s = gmc::f4_zero();
    if (f < a.F) {
        const float4 *p = reinterpret_cast<const float4 *>(a.part);
        for (int t = tl; t < a.tiles; t += kRedLanes) gmc::f4_add(s, p[(long)t * a.F + f]);
    }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && f < a.F) {
        float4 t = red[0][cl];
        for (int i = 1; i < kRedLanes; ++i) gmc::f4_add(t, red[i][cl]);
        a.dW2[(long)f * 3] = t.x; a.dW2[(long)f * 3 + 1] = t.y; a.dW2[(long)f * 3 + 2] = t.z;
        a.db1[f] = t.w;
    }
}

}  // namespace

int gmc_hidden_tiles(int R) { return (R + kTileRows - 1) / kTileRows; }

// part must hold gmc_hidden_tiles(R)*F*4 floats.
int gmc_hidden_bwd_launch(const float *H, long ldh, const float *GY2, const float *W2,
                          const float *dinv, float *Gs, long ldg, float *part, int R, int F,
                          hipStream_t st) {
    if (F % 4 || ldh % 4 || ldg % 4) return GMC_ERR_ALIGN;
    if (R == 0) return GMC_OK;
    HiddenArgs a{H, ldh, GY2, W2, dinv, Gs, ldg, part, R, F};
    dim3 grid(gmc_hidden_tiles(R), (F / 4 + kColThreads - 1) / kColThreads);
    GmcProbeScope probe(GMC_K_HIDDEN_BWD, st);
    hipLaunchKernelGGL(hidden_bwd_kernel, grid, dim3(kColThreads * kRowLanes), 0, st, a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

int gmc_hidden_slab_tiles(int R) { return (R + kSlabTileRows - 1) / kSlabTileRows; }

// slab-layout form: H, Gs are [slice][R][fs]; part must hold gmc_hidden_slab_tiles(R)*F*4 floats
int gmc_hidden_bwd_slab_launch(const float *H, const float *GY2, const float *W2, const float *dinv,
                               float *Gs, float *part, int R, int F, int fs, hipStream_t st) {
    if (F % 4) return GMC_ERR_ALIGN;
    if (R == 0) return GMC_OK;
    HiddenSlabArgs a{H, GY2, W2, dinv, Gs, part, R, F};
    dim3 grid(gmc_hidden_slab_tiles(R), (F + fs - 1) / fs);
    GmcProbeScope probe(GMC_K_HIDDEN_BWD, st);
    switch (fs) {
        case 64: hipLaunchKernelGGL(hidden_bwd_slab_kernel<64>, grid, dim3(256), 0, st, a); break;
        case 32: hipLaunchKernelGGL(hidden_bwd_slab_kernel<32>, grid, dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL(hidden_bwd_slab_kernel<16>, grid, dim3(256), 0, st, a); break;
        default: return GMC_ERR_UNSUPPORTED;
    }
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

int gmc_colsum_reduce_launch(const float *part, int tiles, int F, float *dW2, float *db1,
                             const float *db2part, int B, float *db2, hipStream_t st) {
    ReduceArgs a{part, tiles, F, dW2, db1, db2part, B, db2};
    const int grid = (F + kRedCols - 1) / kRedCols + 1;
    GmcProbeScope probe(GMC_K_COLSUM, st);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(grid), dim3(kRedCols * kRedLanes), 0, st, a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
