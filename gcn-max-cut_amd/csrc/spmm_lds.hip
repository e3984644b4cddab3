// LDS-staged block-diagonal SpMM for gfx950 (the fast path for graphs that fit a CU's LDS).
//
//   Y[g][r, c0:c0+FS] = act( scale[r] * sum_{e in row r} vals[e] * X_g[lcol[e], c0:c0+FS] + bias )
//
// One workgroup owns one graph and a group of its column slices.  It loads the graph's CSR
// once into LDS as 16-bit indices, then per slice: (1) stages the slice of ALL of the graph's
// source rows into LDS with coalesced, fully independent loads - every HBM byte of X is read
// exactly once, the d-fold neighbour re-reads never leave the CU; (2) every output row
// gathers its neighbours from LDS (ds_read_b128) and sums them in CSR order, bitwise
// identical to the row-per-wave kernel in spmm.hip; bias / relu / row scale are fused into
// the store.  Optionally the layer-2 feature transform (Y o scale) @ W2 (K = 3) is
// accumulated per row across the workgroup's slices (LDS, no atomics) and written as one
// partial per slice group; the head kernel folds the partials in fixed order.
//
// Replaces the same reference lines as spmm.hip: DGL's update_all(copy_u,sum) inside
// GraphConv (TrainingNeural.py:80,83), the X@W1 feature transform as a row gather of W1
// (:373: X is the padded adjacency) and (H*outdeg^-1/2)@W2 (:83).
//
// A second kernel, dw1_lds_kernel, is the transposed use: dW1[v, slice] = sum over the graphs
// of a chunk of A_val,g @ U_g - the accumulators stay in registers across graphs.
//
// HBM-bound by construction: bytes moved = algorithmic bytes (+ 2 B/edge of indices once per
// slice group).  All slices of a graph run on one XCD so the indices stay in its L2.
#include "gmc_common.h"
#include <stdlib.h>

namespace {

struct TileArgs {
    gmc_batch b;
    const float *X;     // source rows: batch rows (shared_src = 0) or one shared table (W1)
    long ldx;
    int shared_src;
    int use_vals;
    const float *scale;
    const float *bias;
    int relu;
    float *Y;
    long ldy;
    int F;
    int slices;          // ceil(F / FS)
    int groups;          // slice groups per graph (workgroups per graph)
    const float *W2;     // optional fused (Y o scale) @ W2
    float *Zpart;        // [groups][R][3]
    int dbg;             // diagnostic ablation mask (GMC_LDS_DBG): 1 no stage, 2 no gather, 4 no store
};

// block -> (graph, group): all groups of a graph on one XCD (blocks are dealt round-robin
// over the 8 XCDs), consecutive in that XCD's dispatch order.
__device__ __forceinline__ void tile_of(int b, int B, int S, int &g, int &s) {
    const int full = (B / 8) * 8 * S;
    if (b < full) {
        const int xcd = b & 7, j = b >> 3;
        g = (j / S) * 8 + xcd;
        s = j % S;
    } else {
        const int t = b - full;
        const int rem = B - (B / 8) * 8;  // < 8 graphs left: interleave them
        g = (B / 8) * 8 + t % rem;
        s = t / rem;
    }
}

struct LdsLayout {
    float *tile;           // [n_max][FS]
    unsigned short *rp;    // [n_max + 1]  row offsets relative to the graph's first edge
    unsigned short *lc;    // [nnz_max]    local neighbour ids
};

__device__ __forceinline__ LdsLayout carve_lds(float *base, int n_max, int FS) {
    LdsLayout l;
    l.tile = base;
    l.rp = reinterpret_cast<unsigned short *>(base + (size_t)n_max * FS);
    l.lc = l.rp + ((n_max + 2) & ~1);
    return l;
}

size_t lds_bytes(int n_max, int nnz_max, int FS) {
    return (size_t)n_max * FS * 4 + (size_t)((n_max + 2) & ~1) * 2 + (size_t)((nnz_max + 1) & ~1) * 2;
}

// graph CSR -> LDS (16-bit); caller syncs
template <int THREADS>
__device__ __forceinline__ void load_indices(const gmc_batch &b, int r0, int n, const LdsLayout &L) {
    const int e0 = b.rowptr[r0];
    for (int i = threadIdx.x; i <= n; i += THREADS) L.rp[i] = (unsigned short)(b.rowptr[r0 + i] - e0);
    const int nnz = b.rowptr[r0 + n] - e0;
    for (int i = threadIdx.x; i < nnz; i += THREADS) L.lc[i] = (unsigned short)b.lcol[e0 + i];
}

// stage X[src_row0 + row, c0 + 4q .. +3] for all rows into tile[row][4q..]; idx-th float4 of
// the tile = (row idx / Q, lane idx % Q); all loads of a thread are independent.
template <int FS, int THREADS>
__device__ __forceinline__ void stage_tile(const float *src_q, long ldx, int n, bool col_on, float *tile) {
    constexpr int Q = FS / 4;
    constexpr int kBatch = 8;
    const int total = n * Q;
    for (int base = threadIdx.x; base < total; base += THREADS * kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int idx = base + k * THREADS;
            v[k] = gmc::f4_zero();
            if (idx < total && col_on) v[k] = *reinterpret_cast<const float4 *>(src_q + (long)(idx / Q) * ldx);
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int idx = base + k * THREADS;
            if (idx < total) reinterpret_cast<float4 *>(tile)[idx] = v[k];
        }
    }
}

// sum over the row's neighbours, CSR order, from the LDS tile
template <int FS, bool HAS_VAL>
__device__ __forceinline__ float4 gather_row(const LdsLayout &L, const float *gvals, int beg, int end, int q) {
    constexpr int Q = FS / 4;
    float4 acc = gmc::f4_zero();
    for (int e = beg; e < end; e += 8) {
        int c[8];
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ee = min(e + u, end - 1);
            c[u] = L.lc[ee];
            if (HAS_VAL) w[u] = gvals[ee];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (e + u < end) {
                const float4 x = reinterpret_cast<const float4 *>(L.tile)[c[u] * Q + q];
                if (HAS_VAL) gmc::f4_fma(acc, w[u], x);
                else gmc::f4_add(acc, x);
            }
        }
    }
    return acc;
}

// ACC = rows per thread (ACC * rows-per-pass >= n_max): the per-row scale factors and, with
// EPI, the fused-W2 partial sums of a thread's rows live in registers across the slices.
template <int FS, int THREADS, bool HAS_VAL, int ACC, bool EPI>
__global__ __launch_bounds__(THREADS, 4) void spmm_lds_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = THREADS / Q;
    int g, grp;
    tile_of((int)blockIdx.x, a.b.B, a.groups, g, grp);
    const int r0 = a.b.goff[g];
    const int n = a.b.goff[g + 1] - r0;
    const LdsLayout L = carve_lds(lds, a.b.n_max, FS);
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const int e0 = a.b.rowptr[r0];
    const float *gvals = HAS_VAL ? a.b.vals + e0 : nullptr;
    float zr[EPI ? ACC : 1][3] = {};
    float sc[ACC];
#pragma unroll
    for (int k = 0; k < ACC; ++k) {
        const int l = lrow + k * kRowsPerPass;
        sc[k] = (a.scale && l < n) ? a.scale[r0 + l] : 1.0f;
    }

    load_indices<THREADS>(a.b, r0, n, L);

    const int per = (a.slices + a.groups - 1) / a.groups;
    const int s_beg = grp * per, s_end = min(a.slices, s_beg + per);
    for (int s = s_beg; s < s_end; ++s) {
        const int c0 = s * FS;
        const bool col_on = c0 + 4 * q < a.F;  // ragged last slice
        __syncthreads();                        // previous slice's readers are done with the tile
        if (!(a.dbg & 1))
            stage_tile<FS, THREADS>(a.X + (a.shared_src ? 0L : (long)r0 * a.ldx) + c0 + 4 * q, a.ldx, n, col_on, L.tile);
        float4 bias = gmc::f4_zero();
        if (a.bias && col_on) bias = *reinterpret_cast<const float4 *>(a.bias + c0 + 4 * q);
        float w2[EPI ? 12 : 1];
        if (EPI) {
#pragma unroll
            for (int j = 0; j < 12; ++j) w2[j] = col_on ? a.W2[(long)(c0 + 4 * q) * 3 + j] : 0.f;
        }
        __syncthreads();

#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < n) {
                float4 acc = gmc::f4_zero();
                if (!(a.dbg & 2)) acc = gather_row<FS, HAS_VAL>(L, gvals, L.rp[l], L.rp[l + 1], q);
                float4 y;
                y.x = fmaf(acc.x, sc[k], bias.x); y.y = fmaf(acc.y, sc[k], bias.y);
                y.z = fmaf(acc.z, sc[k], bias.z); y.w = fmaf(acc.w, sc[k], bias.w);
                if (a.relu) {
                    y.x = y.x > 0.f ? y.x : 0.f; y.y = y.y > 0.f ? y.y : 0.f;
                    y.z = y.z > 0.f ? y.z : 0.f; y.w = y.w > 0.f ? y.w : 0.f;
                }
                if (col_on && !(a.dbg & 4)) *reinterpret_cast<float4 *>(a.Y + (long)(r0 + l) * a.ldy + c0 + 4 * q) = y;
                if (EPI) {  // masked lanes carry w2 = 0
                    zr[k][0] += y.x * w2[0] + y.y * w2[3] + y.z * w2[6] + y.w * w2[9];
                    zr[k][1] += y.x * w2[1] + y.y * w2[4] + y.z * w2[7] + y.w * w2[10];
                    zr[k][2] += y.x * w2[2] + y.y * w2[5] + y.z * w2[8] + y.w * w2[11];
                }
            }
            if (EPI) __builtin_amdgcn_sched_barrier(0);  // one row at a time: keeps VGPRs under 128
        }
    }
    if (EPI) {  // fold the row's Q lanes (fixed xor tree), one partial per slice group
        float *zp = a.Zpart + ((long)grp * a.b.R + r0) * 3;
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            float z0 = zr[k][0], z1 = zr[k][1], z2 = zr[k][2];
#pragma unroll
            for (int o = Q / 2; o > 0; o >>= 1) {
                z0 += __shfl_xor(z0, o, GMC_WAVE); z1 += __shfl_xor(z1, o, GMC_WAVE); z2 += __shfl_xor(z2, o, GMC_WAVE);
            }
            const int l = lrow + k * kRowsPerPass;
            if (q == 0 && l < n) {
                zp[3 * l] = z0 * sc[k]; zp[3 * l + 1] = z1 * sc[k]; zp[3 * l + 2] = z2 * sc[k];
            }
        }
    }
}

// ---- dW1: per (slice, graph chunk) accumulate A_val,g @ U_g over the chunk's graphs ------
struct Dw1TileArgs {
    gmc_batch b;
    const float *U;
    long ldu;
    float *out;   // [chunks][n_max][F]
    int F;
    int slices;
    int chunks;
    int graphs_per_chunk;
};

template <int FS, int THREADS, int ACC, bool HAS_VAL>
__global__ __launch_bounds__(THREADS, 4) void dw1_lds_kernel(Dw1TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = THREADS / Q;
    // chunk-major in XCD order: blocks of one chunk (all its slices) share an XCD
    int chunk, s;
    tile_of((int)blockIdx.x, a.chunks, a.slices, chunk, s);
    const LdsLayout L = carve_lds(lds, a.b.n_max, FS);
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const int c0 = s * FS;
    const bool col_on = c0 + 4 * q < a.F;
    float4 acc[ACC];
#pragma unroll
    for (int k = 0; k < ACC; ++k) acc[k] = gmc::f4_zero();

    const int g0 = chunk * a.graphs_per_chunk, g1 = min(a.b.B, g0 + a.graphs_per_chunk);
    for (int g = g0; g < g1; ++g) {
        const int r0 = a.b.goff[g];
        const int n = a.b.goff[g + 1] - r0;
        const int e0 = a.b.rowptr[r0];
        __syncthreads();
        load_indices<THREADS>(a.b, r0, n, L);
        stage_tile<FS, THREADS>(a.U + (long)r0 * a.ldu + c0 + 4 * q, a.ldu, n, col_on, L.tile);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < n) {
                const float4 t = gather_row<FS, HAS_VAL>(L, HAS_VAL ? a.b.vals + e0 : nullptr, L.rp[l], L.rp[l + 1], q);
                gmc::f4_add(acc[k], t);
            }
        }
    }
    if (col_on) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < a.b.n_max) *reinterpret_cast<float4 *>(a.out + ((long)chunk * a.b.n_max + l) * a.F + c0 + 4 * q) = acc[k];
        }
    }
}

struct Shape { int fs, threads; };

// Slice width / workgroup size for graphs of up to n_max nodes and nnz_max edges.
// Preferred: the widest slice whose tile + indices fit 80 KiB, so two 512-thread workgroups
// share a CU and one stages while the other gathers; else one 1024-thread workgroup per CU.
Shape pick_shape(int n_max, int nnz_max) {
    if (n_max > 65535 || nnz_max > 65535) return {0, 0};
    for (int fs = 64; fs >= 16; fs >>= 1)
        if (lds_bytes(n_max, nnz_max, fs) <= 80 * 1024) return {fs, 512};
    for (int fs = 32; fs >= 8; fs >>= 1)
        if (lds_bytes(n_max, nnz_max, fs) <= 160 * 1024) return {fs, 1024};
    return {0, 0};
}

template <int FS, int THREADS>
int launch_spmm(const TileArgs &a, size_t lds, hipStream_t st) {
    const int grid = a.b.B * a.groups;
    constexpr int rows_per_pass = THREADS / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    if (acc > 16) return GMC_ERR_UNSUPPORTED;
#define GMC_GO(HV, AC, EP)                                                                                 \
    do {                                                                                                   \
        auto k = spmm_lds_kernel<FS, THREADS, HV, AC, EP>;                                                 \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k),                  \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, st, a);                                      \
    } while (0)
#define GMC_GO_ACC(HV, EP)                                                                                 \
    do {                                                                                                   \
        if (acc <= 4) GMC_GO(HV, 4, EP); else if (acc <= 8) GMC_GO(HV, 8, EP); else GMC_GO(HV, 16, EP);    \
    } while (0)
    if (a.use_vals) { if (a.Zpart) GMC_GO_ACC(true, true); else GMC_GO_ACC(true, false); }
    else { if (a.Zpart) GMC_GO_ACC(false, true); else GMC_GO_ACC(false, false); }
#undef GMC_GO_ACC
#undef GMC_GO
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

template <int FS, int THREADS>
int launch_dw1(const Dw1TileArgs &a, size_t lds, hipStream_t st) {
    constexpr int rows_per_pass = THREADS / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    if (acc > 16) return GMC_ERR_UNSUPPORTED;
    const int grid = a.slices * a.chunks;
#define GMC_GO(HV, AC)                                                                                     \
    do {                                                                                                   \
        auto k = dw1_lds_kernel<FS, THREADS, AC, HV>;                                                      \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k),                  \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, st, a);                                      \
    } while (0)
#define GMC_GO_ACC(HV)                                                                                     \
    do {                                                                                                   \
        if (acc <= 4) GMC_GO(HV, 4); else if (acc <= 8) GMC_GO(HV, 8); else GMC_GO(HV, 16);                \
    } while (0)
    if (a.b.vals) GMC_GO_ACC(true); else GMC_GO_ACC(false);
#undef GMC_GO_ACC
#undef GMC_GO
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

}  // namespace

// b->nnz_max carries nnz_max (largest per-graph edge count).
bool gmc_lds_fits(const gmc_batch *b) { return pick_shape(b->n_max, b->nnz_max).fs > 0; }

// slice groups (workgroups) per graph == number of Zpart partials of the fused W2 epilogue
int gmc_lds_groups(const gmc_batch *b, int F) {
    const Shape sh = pick_shape(b->n_max, b->nnz_max);
    if (!sh.fs) return 0;
    const int slices = (F + sh.fs - 1) / sh.fs;
    // two slices per workgroup amortise the index load; fixed (independent of the batch) so a
    // graph's result is bitwise the same whatever batch it is part of
    return (slices + 1) / 2;
}

// Y = act(scale * A_g @ X + bias) for every graph of the batch, LDS-staged; optional fused
// Zpart[group][r][:] = scale[r] * (Y[r, group's columns] @ W2[group's rows]).
int gmc_spmm_lds_launch(const gmc_batch *b, const float *X, long ldx, int shared_src, int use_vals,
                        const float *scale, const float *bias, int relu, float *Y, long ldy, int F,
                        const float *W2, float *Zpart, int tag, hipStream_t st) {
    if (!b || !X || !Y) return GMC_ERR_NULL;
    if (F % 4 || ldx % 4 || ldy % 4 || !gmc_aligned16(X) || !gmc_aligned16(Y) || (bias && !gmc_aligned16(bias)))
        return GMC_ERR_ALIGN;
    const Shape sh = pick_shape(b->n_max, b->nnz_max);
    if (sh.fs == 0) return GMC_ERR_UNSUPPORTED;
    if (b->B == 0) return GMC_OK;
    TileArgs a{*b, X, ldx, shared_src, use_vals && b->vals != nullptr, scale, bias, relu, Y, ldy, F,
               (F + sh.fs - 1) / sh.fs, gmc_lds_groups(b, F), W2, Zpart, 0};
    static const int dbg = getenv("GMC_LDS_DBG") ? atoi(getenv("GMC_LDS_DBG")) : 0;
    a.dbg = dbg;
    const size_t lds = lds_bytes(b->n_max, b->nnz_max, sh.fs);
    GmcProbeScope probe(tag, st);
    if (sh.threads == 512) {
        switch (sh.fs) {
            case 64: return launch_spmm<64, 512>(a, lds, st);
            case 32: return launch_spmm<32, 512>(a, lds, st);
            default: return launch_spmm<16, 512>(a, lds, st);
        }
    }
    switch (sh.fs) {
        case 32: return launch_spmm<32, 1024>(a, lds, st);
        case 16: return launch_spmm<16, 1024>(a, lds, st);
        default: return launch_spmm<8, 1024>(a, lds, st);
    }
}

// dW1 partials: out[chunk][v][:] = sum_{g in chunk} sum_e vals[e] * U[g][lcol[e], :], v < n_max
int gmc_dw1_lds_launch(const gmc_batch *b, const float *U, long ldu, float *out, int F, int chunks,
                       int graphs_per_chunk, hipStream_t st) {
    const Shape sh = pick_shape(b->n_max, b->nnz_max);
    if (sh.fs == 0) return GMC_ERR_UNSUPPORTED;
    Dw1TileArgs a{*b, U, ldu, out, F, (F + sh.fs - 1) / sh.fs, chunks, graphs_per_chunk};
    const size_t lds = lds_bytes(b->n_max, b->nnz_max, sh.fs);
    GmcProbeScope probe(GMC_K_DW1, st);
    if (sh.threads == 512) {
        switch (sh.fs) {
            case 64: return launch_dw1<64, 512>(a, lds, st);
            case 32: return launch_dw1<32, 512>(a, lds, st);
            default: return launch_dw1<16, 512>(a, lds, st);
        }
    }
    switch (sh.fs) {
        case 32: return launch_dw1<32, 1024>(a, lds, st);
        case 16: return launch_dw1<16, 1024>(a, lds, st);
        default: return launch_dw1<8, 1024>(a, lds, st);
    }
}
