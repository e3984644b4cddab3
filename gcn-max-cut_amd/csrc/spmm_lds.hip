// LDS-staged block-diagonal SpMM for gfx950 (the fast path for graphs that fit a CU's LDS).
//
//   Y[g][r, c0:c0+FS] = act( scale[r] * sum_{e in row r} vals[e] * X_g[nbr(e), c0:c0+FS] + bias )
//
// One workgroup owns one graph and a group of its column slices.  It copies the graph's
// neighbour table (ELL, 16-bit local ids, 8 or 16 slots per row, padded with the id of an
// all-zero row) into LDS once, then per slice: (1) stages the slice of ALL of the graph's
// source rows into LDS with coalesced, mutually independent loads - every HBM byte of X is
// read exactly once, the d-fold neighbour re-reads never leave the CU; (2) every output row
// reads its 8 neighbour ids with ONE ds_read_b128, gathers the 8 rows with ds_read_b128 and
// sums them in CSR order (padding adds +0.0f), bitwise identical to the row-per-wave kernel
// in spmm.hip; bias / relu / row scale are fused into the store.  The tile is double-buffered:
// the loads of slice s+1 are issued into registers BEFORE the gather of slice s and are written
// to the other LDS buffer after it (software pipeline, one barrier per slice: HBM reads
// overlap the LDS gather and the stores).  One 1024-thread workgroup per CU.  Optionally the layer-2
// feature transform (Y o scale) @ W2 (K = 3) is accumulated per row in registers across the
// workgroup's slices and written as one partial per slice group; the head kernel folds the
// partials in fixed order.
//
// Replaces the same reference lines as spmm.hip: DGL's update_all(copy_u,sum) inside
// GraphConv (TrainingNeural.py:80,83), the X@W1 feature transform as a row gather of W1
// (:373: X is the padded adjacency) and (H*outdeg^-1/2)@W2 (:83).
//
// dw1_lds_kernel is the transposed use: dW1[v, slice] = sum over the graphs of a chunk of
// A_val,g @ U_g - the accumulators stay in registers across graphs.
//
// HBM-bound by construction: bytes moved = algorithmic bytes (+ 2 B per neighbour slot once
// per slice group).  All slices of a graph run on one XCD so its table stays in that L2.
#include "gmc_common.h"
#include <stdlib.h>

// Diagnostic build only (-DGMC_STAMP, `make stamp`): wave 0 of every workgroup accumulates the
// shader-clock cycles it spends in each phase of the fused kernels' tile loop into g_stamps
// (never read by any kernel); gmc_debug_read_stamps copies them out.  The production library
// contains none of this.
#ifdef GMC_STAMP
__device__ unsigned long long g_stamps[4096 * 16];
#define STAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(); unsigned long long st_acc[12] = {}
#define STAMP(i)                                                     \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
        st_acc[i] += st_now - st_last;                               \
        st_last = st_now;                                            \
        __builtin_amdgcn_sched_barrier(0);                           \
    } while (0)
#define STAMP_FLUSH                                                                    \
    do {                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                     \
            for (int i = 0; i < 12; ++i) g_stamps[blockIdx.x * 16 + i] = st_acc[i];    \
    } while (0)
extern "C" int gmc_debug_read_stamps(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

// Diagnostic builds only (`make variant NAME=ablN DEFS=-DGMC_ABLATE=N`, scratch/run_ablate.sh): remove
// ONE component of the fused kernels' tile loop (results are then wrong by construction) to read off
// what that component costs in place.  The production library compiles with GMC_ABLATE == 0.
//   1 tile DMA after the first   2 global stores of the loop   3 fused W2 / column-partial math
//   4 LDS reads of gather #2     5 LDS reads of gather #1      6 workgroup barriers of the loop
//   7 bwd1: the H -> Gs transform
#ifndef GMC_ABLATE
#define GMC_ABLATE 0
#endif
#define ABL(n) (GMC_ABLATE == (n))

namespace {

// threads per workgroup of every kernel in this file; GMC_LDS_THREADS=512 builds the tuning variant
// (two co-resident workgroups per CU when their LDS fits) - never the shipped library
#ifndef GMC_LDS_THREADS
#define GMC_LDS_THREADS 1024
#endif
constexpr int kThreads = GMC_LDS_THREADS;
// 4 waves per SIMD either way (1 x 1024 or 2 x 512 threads per CU): 128 VGPRs per lane
#define GMC_LDS_BOUNDS __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))

struct TileArgs {
    gmc_batch b;
    const float *X;     // source rows: batch rows (shared_src = 0) or one shared table (W1)
    long x_rs, x_ss;    // element (row, col) of X lives at row*x_rs + (col/FS)*x_ss + col%FS:
                        // row-major = (ld, FS); slab layout [slice][row][FS] = (FS, R*FS)
    int shared_src;
    int use_vals;
    const float *scale;
    const float *bias;
    int relu;
    float *Y;
    long y_rs, y_ss;
    int F;
    int slices;          // ceil(F / FS)
    int groups;          // slice groups per graph (workgroups per graph)
    const float *W2;     // optional fused (Y o scale) @ W2
    float *Zpart;        // [groups][R][3]
    int items_per_wg;    // fwd1: (graph, group) items per persistent workgroup
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


// LDS: two tiles [(n_max + 4)][FS] floats (rows n..n+3 = zeros: the padding targets, one per
// bank quarter, see ell_arrange.hip), then the neighbour table [n_max][W] of 16-bit ids.
constexpr int kPadRows = 4;
__host__ __device__ inline size_t tile_floats(int n_max, int FS) { return (size_t)(n_max + kPadRows) * FS; }
constexpr int kMaxSlicesPerWg = 8;
// third LDS region (after the two tiles and the table): the larger of
//   - the column constants (bias, W2 rows): 16 B per column - every slice of up to 1024 columns for
//     the persistent fused forward, kMaxSlicesPerWg slices for the SpMM (8 * FS * 16 B <= that);
//   - the per-row constants (GY2[r,:], dinv[r]) of the graph in flight in bwd1: 16 B per row
// (bwd1's cross-wave fold area of 256 * FS B re-uses a tile buffer after its graph loop).
size_t lds_consts(int n_max, int FS) {
    (void)FS;
    const size_t cols = (size_t)16 * 1024, rows = (size_t)16 * (n_max + kPadRows);
    return cols > rows ? cols : rows;
}
size_t lds_bytes(int n_max, int W, int FS) {
    return 2 * tile_floats(n_max, FS) * 4 + (size_t)n_max * W * 2 + lds_consts(n_max, FS);
}

// Asynchronous tile load (LDS-DMA, global_load_lds_dwordx4): thread t fetches float4
// #(t + k*kThreads) of the tile = (row lrow + k*rows_per_pass, lane q) - exactly the (row, lane)
// pairs it later produces.  No VGPRs are involved; a wave's 64 x 16 B land contiguously at a
// wave-uniform LDS base, which is precisely the [row][FS] order of the tile.
//
// Issued through inline asm so hipcc does not see an LDS write: with the builtin it drains
// vmcnt(0) before the next ds_read (it cannot prove the gather reads the OTHER buffer), which
// serialises the DMA against the gather it is meant to overlap.  The price: completion is
// ours to wait for - dma_wait() before the barrier that precedes the first read of the tile.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's vector-memory operations are outstanding (they retire in issue
// order): used to wait for a tile's DMA while the N stores issued after it stay in flight
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits
// for every global store of the wave to be acknowledged - what the tile loops must not do.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the barriers inside the fused kernels' tile loops (removable in the GMC_ABLATE == 6 diagnostic build)
__device__ __forceinline__ void loop_barrier() {
    if (ABL(6)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else lds_barrier();
}
__device__ __forceinline__ void loop_syncthreads() {
    if (ABL(6)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
}

template <int FS, int ACC>
__device__ __forceinline__ void dma_tile(const float *src_q, long rs, int n, bool col_on, int lrow, float *tile) {
    constexpr int kRowsPerPass = kThreads / (FS / 4);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)tile;
    const unsigned wave_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 16u * (threadIdx.x & ~63u)));
#pragma unroll
    for (int k = 0; k < ACC; ++k) {
        const int l = lrow + k * kRowsPerPass;
        if (l < n && col_on) glds16(src_q + (long)l * rs, wave_dst + 16u * (unsigned)(k * kThreads));
    }
}


// Read the eight tile rows named by eight packed u16 ids (lane's 16 B of each row).  The byte
// address id * row_bytes + (tile + 16 q) is one v_mad_u32_u16 per row (op_sel picks the id's half
// of the dword) instead of the unpack + shift-add pair the compiler emits: the gathers spend about
// as many SIMD cycles on address arithmetic and adds as LDS cycles on the reads.
__device__ __forceinline__ void read8(const float *tile, int q, unsigned row_bytes, const uint4 ids, float4 (&x)[8]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    using lds_f4 = __attribute__((address_space(3))) const v4f;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)tile + 16u * (unsigned)q;
    const unsigned pk[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned lo, hi;
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(lo) : "v"(pk[j]), "s"(row_bytes), "v"(base));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(pk[j]), "s"(row_bytes), "v"(base));
        const v4f a = *(lds_f4 *)(size_t)lo, b = *(lds_f4 *)(size_t)hi;
        x[2 * j] = make_float4(a.x, a.y, a.z, a.w);
        x[2 * j + 1] = make_float4(b.x, b.y, b.z, b.w);
    }
}

// sum over the row's W neighbour slots (CSR order, padding -> zero row) from the LDS tile
template <int FS, int W, bool HAS_VAL>
__device__ __forceinline__ float4 gather_row(const float *tile, const unsigned short *nb, const float *wrow,
                                             int l, int q) {
    float4 acc = gmc::f4_zero();
#pragma unroll
    for (int blk = 0; blk < W / 8; ++blk) {
        const uint4 ids = *reinterpret_cast<const uint4 *>(nb + (long)l * W + blk * 8);
        float4 x[8];
        read8(tile, q, FS * 4, ids, x);
        if (HAS_VAL) {  // weights come from HBM/L2: this (rare) variant waits on them per row
            const float4 w0 = *reinterpret_cast<const float4 *>(wrow + blk * 8);
            const float4 w1 = *reinterpret_cast<const float4 *>(wrow + blk * 8 + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) gmc::f4_fma(acc, w[u], x[u]);
        } else if (blk == 0) {  // last read first: one wait per block (see gather_ids8)
            acc = x[7];
#pragma unroll
            for (int u = 6; u >= 0; --u) gmc::f4_add(acc, x[u]);
        } else {
#pragma unroll
            for (int u = 7; u >= 0; --u) gmc::f4_add(acc, x[u]);
        }
    }
    return acc;
}

// gather_row with the row's ids already fetched (W == 8: one uint4).  Callers issue the id read of
// their NEXT row before calling, so that it returns (LDS answers in order) under the same wait as
// this row's eight reads and no row read ever sits behind an id read of its own.
template <int FS, bool HAS_VAL>
__device__ __forceinline__ float4 gather_ids8(const float *tile, const uint4 ids, const float *wrow, int q) {
    float4 x[8];
    read8(tile, q, FS * 4, ids, x);
    float4 acc;
    if (HAS_VAL) {
        const float4 w0 = *reinterpret_cast<const float4 *>(wrow);
        const float4 w1 = *reinterpret_cast<const float4 *>(wrow + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        acc = make_float4(w[0] * x[0].x, w[0] * x[0].y, w[0] * x[0].z, w[0] * x[0].w);
#pragma unroll
        for (int u = 1; u < 8; ++u) gmc::f4_fma(acc, w[u], x[u]);
    } else {
        // Summed from the last read back: the first add then waits for all eight reads at once
        // (LDS answers in order) and the row costs one s_waitcnt instead of eight.  These kernels
        // are instruction-issue bound (rocprofv3: some instruction active 84 % of SIMD time), so
        // every instruction saved per row counts.  Starts from x[7], not 0 + x[7]: the compiler
        // may not drop an add of +0.0.
        acc = x[7];
#pragma unroll
        for (int u = 6; u >= 0; --u) gmc::f4_add(acc, x[u]);
    }
    return acc;
}

// gather_ids8 for unit weights with the sum written on 2-vectors: 14 v_pk_add_f32 per row wherever it
// is inlined (left to the SLP vectoriser, gather #2 of the fused forward came out as 28 v_add_f32).
template <int FS>
__device__ __forceinline__ gmc::v4f gather_ids8_pk(const float *tile, const uint4 ids, int q) {
    float4 x[8];
    read8(tile, q, FS * 4, ids, x);
    gmc::v2f lo = {x[7].x, x[7].y}, hi = {x[7].z, x[7].w};  // from the last read back: one wait per row
#pragma unroll
    for (int u = 6; u >= 0; --u) {
        lo += (gmc::v2f){x[u].x, x[u].y};
        hi += (gmc::v2f){x[u].z, x[u].w};
    }
    return (gmc::v4f){lo.x, lo.y, hi.x, hi.y};
}

// ACC = rows per thread (ACC * rows-per-pass >= n_max).
//
// Slice loop, software-pipelined so that no wait ever covers a freshly issued memory op
// (vmcnt retires in order and __syncthreads() drains it):
//   stores of slice s-1 (held in registers)  ->  loads of slice s+1 (into registers)  ->
//   LDS gather of slice s  ->  wait (loads s+1 done; stores s-1 long done)  ->
//   write slice s+1 to the other LDS buffer  ->  barrier.
// SHARED: the source is one table shared by every graph (W1 for the layer-1 feature transform)
template <int FS, int W, int ACC, bool EPI, bool HAS_VAL, bool SHARED>
__global__ GMC_LDS_BOUNDS void spmm_lds_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    int g, grp;
    tile_of((int)blockIdx.x, a.b.B, a.groups, g, grp);
    const int r0 = a.b.goff[g];
    const int n = a.b.goff[g + 1] - r0;
    const int per = (a.slices + a.groups - 1) / a.groups;
    const int s_beg = grp * per, s_end = min(a.slices, s_beg + per);
    if (s_beg >= s_end) return;

    const int TF = (int)tile_floats(a.b.n_max, FS);  // floats per tile buffer (offsets keep LDS addressing)
    unsigned short *nb = reinterpret_cast<unsigned short *>(lds + 2 * TF);
    float *cbias = reinterpret_cast<float *>(nb + (size_t)a.b.n_max * W);  // [per*FS] bias of my columns
    float *cw2 = cbias + per * FS;                                          // [per*FS][3] W2 rows of my columns
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
    const float *src0 = a.X + (SHARED ? 0L : (long)r0 * a.x_rs) + 4 * q;

    dma_tile<FS, ACC>(src0 + s_beg * a.x_ss - max(0, s_beg * FS + 4 * q - (a.F - 4)), a.x_rs, n, true, lrow, lds);

    // prologue: every global read is issued before the first use, so the workgroup pays one memory
    // latency (not one per table) before its first gather
    float zr[EPI ? ACC : 1][3] = {};
    float sc[ACC];
#pragma unroll
    for (int k = 0; k < ACC; ++k) {
        const int l = lrow + k * kRowsPerPass;
        sc[k] = a.scale ? a.scale[r0 + min(l, n - 1)] : 1.0f;  // rows past n redo row n-1 with ITS scale
    }
    constexpr int NT = (ACC * kRowsPerPass * (W / 8) + kThreads - 1) / kThreads;  // table uint4 per thread
    uint4 pt[NT];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.b.ell + (long)r0 * W);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int i = threadIdx.x + k * kThreads;
            pt[k] = i < n * (W / 8) ? src[i] : make_uint4(0, 0, 0, 0);
        }
    }
    const int ci = threadIdx.x, cc = s_beg * FS + ci;  // column constants of my slices (per * FS <= kThreads)
    const bool c_on = ci < per * FS;
    const float cb = (c_on && a.bias && cc < a.F) ? a.bias[cc] : 0.f;
    float cw[3] = {0.f, 0.f, 0.f};
    if (EPI && c_on && cc < a.F) {
#pragma unroll
        for (int j = 0; j < 3; ++j) cw[j] = a.W2[(long)cc * 3 + j];
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int i = threadIdx.x + k * kThreads;
        if (i < n * (W / 8)) reinterpret_cast<uint4 *>(nb)[i] = pt[k];
    }
    if (c_on) {
        cbias[ci] = cb;
        if (EPI) {
#pragma unroll
            for (int j = 0; j < 3; ++j) cw2[3 * ci + j] = cw[j];
        }
    }
    if (threadIdx.x < kPadRows * FS) {  // the zero rows padding entries point at
        lds[(long)n * FS + threadIdx.x] = 0.f;
        lds[TF + (long)n * FS + threadIdx.x] = 0.f;
    }
    dma_wait();
    __syncthreads();

    // gather + epilogue + store of slice s from tile buffer `cur` (and the fused-W2 partials).
    // Every wave issues exactly ACC store instructions per slice - rows past n repeat row n-1 (same
    // value to the same address) and a wave always holds lanes of in-range columns - which is what
    // lets the loop wait for the younger-by-ACC DMA with a counted vmcnt.
    auto compute = [&](int s, int cur) {
        const float *tile = lds + cur * TF;
        const int cl = (s - s_beg) * FS + 4 * q;
        const float4 bias = *reinterpret_cast<const float4 *>(cbias + cl);
        const bool col_on = s * FS + 4 * q < a.F;
        const float lo = a.relu ? 0.f : -__builtin_inff();
        float *ydst = a.Y + (long)s * a.y_ss + 4 * q;
        float4 wa, wb, wc;
        if (EPI) {  // W2 rows of my 4 columns from LDS (zero for pad columns)
            wa = *reinterpret_cast<const float4 *>(cw2 + 3 * cl);
            wb = *reinterpret_cast<const float4 *>(cw2 + 3 * cl + 4);
            wc = *reinterpret_cast<const float4 *>(cw2 + 3 * cl + 8);
        }
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = min(lrow + k * kRowsPerPass, n - 1);
            const float4 acc = gather_row<FS, W, HAS_VAL>(tile, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q);
            // pad columns: the tile holds finite values there (see prefetch), scale 0 and bias 0 make
            // them exact zeros without a per-row select; relu / identity is one v_max either way
            const float sk = (EPI && !col_on) ? 0.f : sc[k];
            float4 y;
            y.x = gmc::max1(fmaf(acc.x, sk, bias.x), lo); y.y = gmc::max1(fmaf(acc.y, sk, bias.y), lo);
            y.z = gmc::max1(fmaf(acc.z, sk, bias.z), lo); y.w = gmc::max1(fmaf(acc.w, sk, bias.w), lo);
            if (col_on) *reinterpret_cast<float4 *>(ydst + (long)(r0 + l) * a.y_rs) = y;
            if (EPI) {
                zr[k][0] += y.x * wa.x + y.y * wa.w + y.z * wb.z + y.w * wc.y;
                zr[k][1] += y.x * wa.y + y.y * wb.x + y.z * wb.w + y.w * wc.z;
                zr[k][2] += y.x * wa.z + y.y * wb.y + y.z * wc.x + y.w * wc.w;
            }
        }
    };
    // tile of slice s; pad columns (>= F, last slice) load the slice's last valid column group again,
    // so the tile never holds stale or uninitialised (possibly non-finite) values
    auto prefetch = [&](int s, int into) {
        const int back = max(0, s * FS + 4 * q - (a.F - 4));  // columns past the last valid group
        dma_tile<FS, ACC>(src0 + s * a.x_ss - back, a.x_rs, n, true, lrow, lds + into * TF);
    };

    // Steady state: the DMA of slice s+1 is issued first, the LDS gather of slice s runs while it is
    // in flight and stores each row as it is finished; the wait then covers the DMA only (vector
    // memory operations retire in issue order, the ACC stores are younger) and the barrier orders
    // LDS traffic only, so no wave ever waits for a store to be acknowledged.
    for (int s = s_beg; s < s_end; ++s) {
        const int cur = (s - s_beg) & 1;
        const bool more = s + 1 < s_end;
        if (more) prefetch(s + 1, cur ^ 1);
        compute(s, cur);
        if (more) {
            vm_wait<ACC>();
            lds_barrier();  // next tile landed; everyone is done reading this one
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
    long u_rs, u_ss;  // layout of U (see TileArgs)
    float *out;   // [chunks][n_max][F]
    int F;
    int slices;
    int chunks;
    int graphs_per_chunk;
};

template <int FS, int W, int ACC, bool HAS_VAL>
__global__ GMC_LDS_BOUNDS void dw1_lds_kernel(Dw1TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    constexpr int NT = (ACC * kRowsPerPass * (W / 8) + kThreads - 1) / kThreads;  // table uint4 per thread
    int chunk, s;  // chunk-major in XCD order: the slices of one chunk share an XCD
    tile_of((int)blockIdx.x, a.chunks, a.slices, chunk, s);
    const int TF = (int)tile_floats(a.b.n_max, FS);
    unsigned short *nb = reinterpret_cast<unsigned short *>(lds + 2 * TF);
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const int c0 = s * FS;
    const bool col_on = c0 + 4 * q < a.F;
    float4 acc[ACC];
    uint4 pt[NT];
#pragma unroll
    for (int k = 0; k < ACC; ++k) acc[k] = gmc::f4_zero();

    const int g0 = chunk * a.graphs_per_chunk, g1 = min(a.b.B, g0 + a.graphs_per_chunk);
    if (g0 >= g1) return;
    auto fetch = [&](int g, int into) {  // tile slice -> LDS buffer `into` (DMA); neighbour table -> registers
        const int r0 = a.b.goff[g];
        const int n = a.b.goff[g + 1] - r0;
        dma_tile<FS, ACC>(a.U + (long)r0 * a.u_rs + (long)s * a.u_ss + 4 * q, a.u_rs, n, col_on, lrow, lds + into * TF);
        const uint4 *src = reinterpret_cast<const uint4 *>(a.b.ell + (long)r0 * W);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int i = threadIdx.x + k * kThreads;
            pt[k] = i < n * (W / 8) ? src[i] : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit_table = [&](int n) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int i = threadIdx.x + k * kThreads;
            if (i < n * (W / 8)) reinterpret_cast<uint4 *>(nb)[i] = pt[k];
        }
    };
    fetch(g0, 0);
    dma_wait();
    {
        const int n = a.b.goff[g0 + 1] - a.b.goff[g0];
        commit_table(n);
        if (threadIdx.x < kPadRows * FS) lds[n * FS + threadIdx.x] = 0.f;
    }
    __syncthreads();
    for (int g = g0; g < g1; ++g) {
        const int cur = (g - g0) & 1;
        const int r0 = a.b.goff[g];
        const int n = a.b.goff[g + 1] - r0;
        if (g + 1 < g1) fetch(g + 1, cur ^ 1);
        const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < n) gmc::f4_add(acc[k], gather_row<FS, W, HAS_VAL>(lds + cur * TF, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q));
        }
        if (g + 1 < g1) {
            const int n1 = a.b.goff[g + 2] - a.b.goff[g + 1];
            if (threadIdx.x < kPadRows * FS) lds[(cur ^ 1) * TF + n1 * FS + threadIdx.x] = 0.f;
            __syncthreads();  // everyone is done with graph g's table
            commit_table(n1);
        }
        dma_wait();
        __syncthreads();
    }
    if (col_on) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < a.b.n_max) *reinterpret_cast<float4 *>(a.out + ((long)chunk * a.b.n_max + l) * a.F + c0 + 4 * q) = acc[k];
        }
    }
}

// ---- fused layer-1 forward: W1 row gather + aggregation (+ fused H@W2) in one kernel ---------
//
// Workgroup = (graph, slice group).  Per slice: the W1 slice (rows 0..n-1 of the shared
// [N,F] table, L2-resident) arrives in buffer A by LDS-DMA; gather #1 builds the T0 tile =
// dinv o (A_val @ W1[:n]) in buffer B (the X@W1 of TrainingNeural.py:80 with X = padded
// adjacency, :373); the next slice's W1 tile is then DMA'd into A while gather #2 produces
// H = relu(dinv o (A @ T0) + b1) (:80-81) from B, with the layer-2 feature transform
// (H o dinv) @ W2 (:83) accumulated in registers.  T0 never exists in HBM: the forward of layer 1
// writes H once and reads only W1 (from L2) and the neighbour table.
template <int FS, int W, int ACC, bool HAS_VAL>
__global__ GMC_LDS_BOUNDS void fwd1_lds_kernel(TileArgs a) {
    STAMP_DECL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    constexpr int NT = (ACC * kRowsPerPass * (W / 8) + kThreads - 1) / kThreads;  // table uint4 per thread
    // Persistent workgroup: items (graph, slice group) are numbered graph-major and every workgroup
    // owns a contiguous range of them (one workgroup per CU when the batch is large), so it walks
    // through consecutive slices of one graph, then of the next.  The graph's table / scales are set up
    // once per graph instead of once per item and the tile pipeline does not drain between the groups
    // of a graph; what a group leaves behind is only its Zpart partial (same values as one
    // workgroup per item: a graph's result does not depend on the batch it is part of).
    const int total = a.b.B * a.groups;
    const int it0 = (int)blockIdx.x * a.items_per_wg, it1 = min(total, it0 + a.items_per_wg);
    if (it0 >= it1) return;
    const int per = (a.slices + a.groups - 1) / a.groups;

    const int TF = (int)tile_floats(a.b.n_max, FS);
    float *bufA = lds, *bufB = lds + TF;
    unsigned short *nb = reinterpret_cast<unsigned short *>(lds + 2 * TF);
    // column constants of every slice, once per workgroup: [slices * FS] x (W2[c,0..2], b1[c])
    float4 *cst = reinterpret_cast<float4 *>(nb + (size_t)a.b.n_max * W);
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;

    {   // (slices * FS <= 1024 = kThreads: one column per thread; pad columns hold zeros)
        const int cc = threadIdx.x;
        const bool c_on = cc < a.slices * FS;
        float4 c = gmc::f4_zero();
        if (c_on && cc < a.F) {
            if (a.W2) { c.x = a.W2[(long)cc * 3]; c.y = a.W2[(long)cc * 3 + 1]; c.z = a.W2[(long)cc * 3 + 2]; }
            if (a.bias) c.w = a.bias[cc];
        }
        if (c_on) cst[cc] = c;
    }

    for (int it = it0; it < it1;) {
        const int g = it / a.groups;
        const int it_end = min(it1, (g + 1) * a.groups);           // my items of graph g
        const int s_lo = (it - g * a.groups) * per, s_hi = min(a.slices, (it_end - g * a.groups) * per);
        const int r0 = a.b.goff[g];
        const int n = a.b.goff[g + 1] - r0;
        const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
        // W1 tile of slice s, row-major [N][ldx].  Pad columns (>= F, last slice only) load the last
        // valid column group again: finite values, so that the masked scale below makes exact zeros
        auto dma = [&](int s) {
            dma_tile<FS, ACC>(a.X + min(s * FS + 4 * q, a.F - 4), a.x_rs, n, true, lrow, bufA);
        };
        if (it != it0) lds_barrier();  // every wave is done with the previous graph's table and tiles
        // graph prologue: every global read is issued before the first use (one memory latency)
        dma(s_lo);
        float sc[ACC];
#pragma unroll
        for (int k = 0; k < ACC; ++k) sc[k] = a.scale[r0 + min(lrow + k * kRowsPerPass, n - 1)];
        uint4 pt[NT];
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.b.ell + (long)r0 * W);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = threadIdx.x + k * kThreads;
                pt[k] = i < n * (W / 8) ? src[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = threadIdx.x + k * kThreads;
                if (i < n * (W / 8)) reinterpret_cast<uint4 *>(nb)[i] = pt[k];
            }
        }
        if (threadIdx.x < kPadRows * FS) {  // the zero rows padding entries point at
            bufA[(long)n * FS + threadIdx.x] = 0.f;
            bufB[(long)n * FS + threadIdx.x] = 0.f;
        }
        gmc::v2f z01[ACC];  // (Z[r,0], Z[r,1]) partial of my 4 columns, per row
        float z2[ACC];      //  Z[r,2]
#pragma unroll
        for (int k = 0; k < ACC; ++k) { z01[k] = gmc::splat2(0.f); z2[k] = 0.f; }
        // a slice group's Zpart partial.  Called one gather later than the group ends (after the next
        // slice's barrier 2): by then the group's H stores have long retired, so whatever vector-memory
        // wait the compiler attaches to this rarely-run block (spill reloads) costs nothing
        auto flush = [&](int grp) {
            if (a.Zpart) {
                float *zp = a.Zpart + ((long)grp * a.b.R + r0) * 3;
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    float z0 = z01[k].x, z1 = z01[k].y, zz = z2[k];
#pragma unroll
                    for (int o = Q / 2; o > 0; o >>= 1) {
                        z0 += __shfl_xor(z0, o, GMC_WAVE); z1 += __shfl_xor(z1, o, GMC_WAVE); zz += __shfl_xor(zz, o, GMC_WAVE);
                    }
                    int l = lrow + k * kRowsPerPass;
                    asm volatile("" : "+v"(l));  // keeps the store addresses out of the slice loop's live set
                    if (q == 0 && l < n) {
                        zp[3 * l] = z0 * sc[k]; zp[3 * l + 1] = z1 * sc[k]; zp[3 * l + 2] = zz * sc[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < ACC; ++k) { z01[k] = gmc::splat2(0.f); z2[k] = 0.f; }
        };
        dma_wait();  // table / first tile
        STAMP(11);  // prologue
        for (int s = s_lo; s < s_hi; ++s) {
            STAMP(0);  // loop overhead / previous tail
            // the W1 tile of slice s has landed once at most the ACC stores issued after its DMA are left
            if (s > s_lo && !ABL(1) && !ABL(2)) vm_wait<ACC>();
            STAMP(1);  // DMA wait
            loop_barrier();  // ... for every wave; readers of the previous T0 tile are done
            STAMP(2);  // barrier 1
            // gather #1: T0 tile
            if constexpr (W == 8) {
                uint4 ids = reinterpret_cast<const uint4 *>(nb)[min(lrow, n - 1)];
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    const int l = lrow + k * kRowsPerPass;
                    const uint4 cur = ids;
                    if (k + 1 < ACC) ids = reinterpret_cast<const uint4 *>(nb)[min(l + kRowsPerPass, n - 1)];
                    // rows past n redo row n-1 (same value to the same address): no exec-mask juggling
                    const int lc = min(l, n - 1);
                    float4 t = ABL(5) ? make_float4(sc[k], sc[k], sc[k], sc[k])
                                      : gather_ids8<FS, HAS_VAL>(bufA, cur, HAS_VAL ? wbase + (long)lc * W : nullptr, q);
                    t.x *= sc[k]; t.y *= sc[k]; t.z *= sc[k]; t.w *= sc[k];
                    reinterpret_cast<float4 *>(bufB)[lc * Q + q] = t;
                }
            } else {
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    const int l = lrow + k * kRowsPerPass;
                    if (l < n) {
                        float4 t = gather_row<FS, W, HAS_VAL>(bufA, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q);
                        t.x *= sc[k]; t.y *= sc[k]; t.z *= sc[k]; t.w *= sc[k];
                        reinterpret_cast<float4 *>(bufB)[l * Q + q] = t;
                    }
                }
            }
            STAMP(3);  // gather 1
            loop_barrier();
            STAMP(4);  // barrier 2
            if (s > s_lo && s % per == 0) flush(s / per - 1);  // the group that ended with slice s-1
            if (s + 1 < s_hi && !ABL(1)) dma(s + 1);  // buffer A is free: the next W1 tile streams in during gather #2
            STAMP(5);  // DMA issue
            // gather #2: H rows + fused W2; every thread issues exactly ACC stores (rows past n repeat
            // row n-1: same value to the same address) so that the vm_wait above counts exactly
            // my 4 columns' constants (indexed by absolute column): W2 rows as (w0,w1) pairs + w2, bias pairs
            const float4 c0 = cst[s * FS + 4 * q], c1 = cst[s * FS + 4 * q + 1], c2 = cst[s * FS + 4 * q + 2],
                         c3 = cst[s * FS + 4 * q + 3];
            const gmc::v2f w01[4] = {{c0.x, c0.y}, {c1.x, c1.y}, {c2.x, c2.y}, {c3.x, c3.y}};
            const float w2c[4] = {c0.z, c1.z, c2.z, c3.z};
            const gmc::v2f blo = {c0.w, c1.w}, bhi = {c2.w, c3.w};
            const bool col_pad = s * FS + 4 * q >= a.F;  // slab pad columns: stored as zeros
            float *ydst = a.Y + (long)s * a.y_ss + 4 * q;
            // pad columns: finite tile values (see dma) * scale 0 + bias 0 = exact zeros, no per-row select
            float scm[ACC];
#pragma unroll
            for (int k = 0; k < ACC; ++k) scm[k] = col_pad ? 0.f : sc[k];
            auto emit = [&](int k, const gmc::v4f acc) {
                const int l = min(lrow + k * kRowsPerPass, n - 1);
                const gmc::v2f s2 = gmc::splat2(scm[k]);
                const gmc::v2f ylo = gmc::pk_fma((gmc::v2f){acc.x, acc.y}, s2, blo);
                const gmc::v2f yhi = gmc::pk_fma((gmc::v2f){acc.z, acc.w}, s2, bhi);
                float4 y;
                y.x = gmc::relu1(ylo.x); y.y = gmc::relu1(ylo.y); y.z = gmc::relu1(yhi.x); y.w = gmc::relu1(yhi.y);  // F.relu, :81
                if (!ABL(2)) *reinterpret_cast<float4 *>(ydst + (long)(r0 + l) * a.y_rs) = y;
                if (ABL(3)) { z2[k] += y.x + y.y + y.z + y.w; return; }
                // (H o dinv) @ W2 for my columns (:83; dinv applied at the flush): 4 packed + 4 scalar FMAs
                z01[k] = gmc::pk_fma(gmc::splat2(y.x), w01[0], z01[k]); z2[k] = fmaf(y.x, w2c[0], z2[k]);
                z01[k] = gmc::pk_fma(gmc::splat2(y.y), w01[1], z01[k]); z2[k] = fmaf(y.y, w2c[1], z2[k]);
                z01[k] = gmc::pk_fma(gmc::splat2(y.z), w01[2], z01[k]); z2[k] = fmaf(y.z, w2c[2], z2[k]);
                z01[k] = gmc::pk_fma(gmc::splat2(y.w), w01[3], z01[k]); z2[k] = fmaf(y.w, w2c[3], z2[k]);
            };
            uint4 ids2 = reinterpret_cast<const uint4 *>(nb)[min(lrow, n - 1)];
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int l = min(lrow + k * kRowsPerPass, n - 1);
                if constexpr (W == 8) {
                    const uint4 cur = ids2;
                    if (k + 1 < ACC) ids2 = reinterpret_cast<const uint4 *>(nb)[min(l + kRowsPerPass, n - 1)];
                    emit(k, ABL(4) ? (gmc::v4f)(__uint_as_float(cur.x)) : gather_ids8_pk<FS>(bufB, cur, q));
                } else {
                    emit(k, gmc::f4v(gather_row<FS, W, false>(bufB, nb, nullptr, l, q)));
                }
            }
            STAMP(6);  // gather 2
        }
        flush((s_hi - 1) / per);  // last group of this graph segment
        it = it_end;
    }
    STAMP(7);  // epilogue
    STAMP_FLUSH;
}

// ---- fused layer-1 backward: hidden backward + aggregation + dW1 in one pass over H --------
//
// Workgroup = (column slice, chunk of graphs).  Per graph: the H tile arrives by LDS-DMA; every
// thread turns its own elements into Gs = dinv^2 o relu'(H) o (GY2 @ W2^T) IN PLACE (and adds
// their dW2 / db1 terms to register partials); gather #1 builds the U tile = dinv o (A @ Gs) in
// the second LDS buffer; the next graph's H tile is then DMA'd into the first buffer while
// gather #2 accumulates dW1 += A_val @ U in registers.  Gs and U never exist in HBM: the whole
// backward of layer 1 (autograd of TrainingNeural.py:80-83, run by loss.backward() :385)
// reads H once.  Outputs: dW1 partial [chunk][n_max][F] and column partials [chunk][F][4]
// = (dW2[f,0..2], db1[f]), folded in chunk order by fold_chunks / colsum_reduce.
struct Bwd1Args {
    gmc_batch b;
    const float *H;       // slab layout [slice][R][FS]
    const float *GY2;     // [R][4] = (GY2[r,0..2], dinv[r])
    const float *W2;      // [F][3]
    float *dw1part;       // [chunks][n_max][F]
    float *colpart;       // [chunks][F][4]
    int F;
    int slices;
    int chunks;
    int graphs_per_chunk;
};

template <int FS, int W, int ACC, bool HAS_VAL>
__global__ GMC_LDS_BOUNDS void bwd1_lds_kernel(Bwd1Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    constexpr int NT = (ACC * kRowsPerPass * (W / 8) + kThreads - 1) / kThreads;
    constexpr bool kRegIds = W == 8;
    int chunk, s;
    tile_of((int)blockIdx.x, a.chunks, a.slices, chunk, s);
    const int TF = (int)tile_floats(a.b.n_max, FS);
    float *bufA = lds, *bufB = lds + TF;
    unsigned short *nb = reinterpret_cast<unsigned short *>(lds + 2 * TF);
    float *gyl = reinterpret_cast<float *>(nb + (size_t)a.b.n_max * W);  // [n][4] = (GY2[r,:], dinv[r]) of the graph
    float *red = bufB;  // [16 waves][Q][16] cross-wave fold area, used after the graph loop
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const int f0 = s * FS + 4 * q;
    const bool col_on = f0 < a.F;
    const long slab = (long)s * a.b.R * FS;
    // my 4 columns as two pairs p = (f0+2p, f0+2p+1): W2 rows, dW2 / db1 partials - all on 2-vectors
    // (v_pk_fma_f32), the transform is VALU work on every element of H
    gmc::v2f w2p[2][3];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            w2p[p][k] = col_on ? (gmc::v2f){a.W2[(long)(f0 + 2 * p) * 3 + k], a.W2[(long)(f0 + 2 * p + 1) * 3 + k]}
                               : gmc::splat2(0.f);
    gmc::v4f acc[ACC];
    gmc::v2f cdw2[3][2], cdb1[2];  // dW2[pair, k] and db1[pair] partials of my rows
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        cdb1[p] = gmc::splat2(0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) cdw2[k][p] = gmc::splat2(0.f);
    }
    uint4 pt[NT];
#pragma unroll
    for (int k = 0; k < ACC; ++k) acc[k] = (gmc::v4f)(0.f);

    const int g0 = chunk * a.graphs_per_chunk, g1 = min(a.b.B, g0 + a.graphs_per_chunk);
    if (g0 >= g1) return;
    auto fetch = [&](int r0, int n) {  // H tile -> bufA (DMA); neighbour table -> registers
        dma_tile<FS, ACC>(a.H + slab + (long)r0 * FS + 4 * q, FS, n, true, lrow, bufA);
        {   // the graph's row constants, 16 B per row, by the same DMA path (lane i -> row i)
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)gyl;
            for (int i0 = 0; i0 < n; i0 += kThreads) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 16u * (unsigned)(i0 + (threadIdx.x & ~63u))));
                if (i0 + (int)threadIdx.x < n) glds16(a.GY2 + (long)(r0 + i0 + threadIdx.x) * 4, dst);
            }
        }
        if constexpr (!kRegIds) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.b.ell + (long)r0 * W);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = threadIdx.x + k * kThreads;
                pt[k] = i < n * (W / 8) ? src[i] : make_uint4(0, 0, 0, 0);
            }
        }
    };
    // W == 8: a thread's rows' neighbour ids live in registers for the whole graph (both gathers):
    // no table in LDS, no table commit and one barrier less per graph
    uint4 idr[kRegIds ? ACC : 1];
    // rows past n get eight pad ids (the zero row n): their gathers return +0, so the tile loop needs
    // no exec masks - such a thread adds 0 to its dW1 accumulator and writes 0 into the zero row
    auto load_ids = [&](int r0, int n) {
        const unsigned pad = (unsigned)n * 0x10001u;
#pragma unroll
        for (int k = 0; k < (kRegIds ? ACC : 1); ++k) {
            const int l = lrow + k * kRowsPerPass;
            const uint4 v = *reinterpret_cast<const uint4 *>(a.b.ell + (long)(r0 + min(l, n - 1)) * W);
            idr[k] = l < n ? v : make_uint4(pad, pad, pad, pad);
        }
    };
    auto commit_table = [&](int n) {  // (and the zero rows the padding entries point at)
        if constexpr (!kRegIds) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = threadIdx.x + k * kThreads;
                if (i < n * (W / 8)) reinterpret_cast<uint4 *>(nb)[i] = pt[k];
            }
        }
        if (threadIdx.x < kPadRows * FS) {
            bufA[n * FS + threadIdx.x] = 0.f;
            bufB[n * FS + threadIdx.x] = 0.f;
        }
    };
    // graph offsets are scalar loads: each is requested one graph ahead of its first use
    int r0 = a.b.goff[g0], n = a.b.goff[g0 + 1] - r0;
    fetch(r0, n);
    if constexpr (kRegIds) load_ids(r0, n);
    commit_table(n);
    dma_wait();
    __syncthreads();
    STAMP_DECL;
    for (int g = g0; g < g1; ++g) {
        const int r0n = r0 + n;                                        // == goff[g + 1]
        const int nn = g + 1 < g1 ? a.b.goff[g + 2] - r0n : n;         // next graph's size (used after gather 1)
        float dv[ACC];
        STAMP(0);
        // (1) H -> Gs in place + column partials.  The row constants (GY2[r,:], dinv[r]) came with the
        // tile by DMA: an LDS read per row instead of a global-memory latency per row.  Pad columns
        // need no masks: their W2 rows are 0 here and the H slab holds exact zeros there.  Rows past n
        // work on the zero row n (h = 0 -> Gs = 0, partials += 0): no exec masks either.
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            const float4 rck = reinterpret_cast<const float4 *>(gyl)[min(l, n - 1)];
            const float d = rck.w;
            dv[k] = d;
            if (ABL(7)) continue;
            float4 *cell = reinterpret_cast<float4 *>(bufA) + min(l, n) * Q + q;
            const float4 h = *cell;
            const gmc::v2f g0 = gmc::splat2(rck.x * d), g1 = gmc::splat2(rck.y * d), g2 = gmc::splat2(rck.z * d);
            const gmc::v2f hp[2] = {{h.x, h.y}, {h.z, h.w}};
            gmc::v2f gs[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                gmc::v2f ghd = g0 * w2p[p][0];
                ghd = gmc::pk_fma(g1, w2p[p][1], ghd);
                ghd = gmc::pk_fma(g2, w2p[p][2], ghd);
                const gmc::v2f gpre = {hp[p].x > 0.f ? ghd.x : 0.f, hp[p].y > 0.f ? ghd.y : 0.f};  // relu' o dinv o (GY2 W2^T)
                gs[p] = gpre * gmc::splat2(d);
                if (ABL(3)) continue;
                cdw2[0][p] = gmc::pk_fma(hp[p], g0, cdw2[0][p]);  // dW2 = (H o dinv)^T GY2
                cdw2[1][p] = gmc::pk_fma(hp[p], g1, cdw2[1][p]);
                cdw2[2][p] = gmc::pk_fma(hp[p], g2, cdw2[2][p]);
                cdb1[p] += gpre;                                  // db1
            }
            *cell = make_float4(gs[0].x, gs[0].y, gs[1].x, gs[1].y);
            // pin the partials here: their only user is the end of the graph loop, and left alone the
            // optimiser sinks these FMAs past the gathers - keeping every row's h and g alive (600 B of
            // scratch per lane)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                asm volatile("" : "+v"(cdw2[0][p]), "+v"(cdw2[1][p]), "+v"(cdw2[2][p]), "+v"(cdb1[p]));
        }
        STAMP(1);  // transform
        loop_syncthreads();
        STAMP(2);  // barrier A
        // (2) U tile = dinv o (A @ Gs); rows past n redo row n-1 (same value to the same address)
        if constexpr (kRegIds) {
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int l = min(lrow + k * kRowsPerPass, n);  // rows past n: 0 into the zero row
                float4 u = ABL(5) ? make_float4(dv[k], dv[k], dv[k], dv[k]) : gather_ids8<FS, false>(bufA, idr[k], nullptr, q);
                u.x *= dv[k]; u.y *= dv[k]; u.z *= dv[k]; u.w *= dv[k];
                reinterpret_cast<float4 *>(bufB)[l * Q + q] = u;
            }
        } else {
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int l = lrow + k * kRowsPerPass;
                if (l < n) {
                    float4 u = gather_row<FS, W, false>(bufA, nb, nullptr, l, q);
                    u.x *= dv[k]; u.y *= dv[k]; u.z *= dv[k]; u.w *= dv[k];
                    reinterpret_cast<float4 *>(bufB)[l * Q + q] = u;
                }
            }
        }
        STAMP(3);  // gather 1
        loop_syncthreads();
        STAMP(4);  // barrier B
        // (3) next graph's H tile streams into bufA while (4) gathers from bufB
        if (g + 1 < g1 && !ABL(1)) fetch(r0n, nn);
        STAMP(5);  // fetch issue
        const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
        if constexpr (kRegIds) {
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int l = min(lrow + k * kRowsPerPass, n - 1);  // (weights of a real row; the ids are pads past n)
                if constexpr (HAS_VAL) acc[k] += gmc::f4v(gather_ids8<FS, true>(bufB, idr[k], wbase + (long)l * W, q));
                else acc[k] += ABL(4) ? (gmc::v4f)(__uint_as_float(idr[k].x)) : gather_ids8_pk<FS>(bufB, idr[k], q);
                // the sum is needed HERE (its only user is the store after the graph loop: left alone the
                // optimiser sinks the adds and keeps four rows of reads, 128 VGPRs, alive)
                asm volatile("" : "+v"(acc[k]));
            }
        } else {
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int l = lrow + k * kRowsPerPass;
                if (l < n) acc[k] += gmc::f4v(gather_row<FS, W, HAS_VAL>(bufB, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q));
            }
        }
        STAMP(6);  // gather 2
        dma_wait();
        STAMP(7);  // DMA wait
        loop_syncthreads();  // everyone is done with graph g's table and U tile; DMA has landed
        STAMP(8);  // barrier C
        if (g + 1 < g1) {
            // next graph's ids: requested now, first needed after the transform and barrier A
            if constexpr (kRegIds) load_ids(r0n, nn);
            commit_table(nn);
        }
        STAMP(9);  // commit table
        if constexpr (!kRegIds) __syncthreads();  // (register ids: the pad rows are ordered by barrier A)
        STAMP(10); // barrier D
        r0 = r0n; n = nn;
    }
    STAMP_FLUSH;
    if (col_on) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < a.b.n_max) *reinterpret_cast<float4 *>(a.dw1part + ((long)chunk * a.b.n_max + l) * a.F + f0) = gmc::v4f_f4(acc[k]);
        }
    }
    // column partials: fold the lanes sharing q inside each wave, then the 16 waves (fixed order)
    float colp[16];  // [j][0..2] dW2, [j][3] db1 for my 4 columns j
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { colp[4 * (2 * p) + k] = cdw2[k][p].x; colp[4 * (2 * p + 1) + k] = cdw2[k][p].y; }
        colp[4 * (2 * p) + 3] = cdb1[p].x; colp[4 * (2 * p + 1) + 3] = cdb1[p].y;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int o = 32; o >= Q; o >>= 1) colp[i] += __shfl_xor(colp[i], o, GMC_WAVE);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < Q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * Q + lane) * 16 + i] = colp[i];
    }
    __syncthreads();
    if (threadIdx.x < Q * 4) {  // thread = (q, j): column f = s*FS + 4q + j
        const int qq = threadIdx.x / 4, j = threadIdx.x % 4;
        if (s * FS + 4 * qq < a.F) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int wv = 0; wv < kThreads / 64; ++wv)
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] += red[(wv * Q + qq) * 16 + 4 * j + c];
            reinterpret_cast<float4 *>(a.colpart)[(long)chunk * a.F + s * FS + 4 * qq + j] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Slice width for graphs of up to n_max nodes with W neighbour slots: the widest slice whose
// two tile buffers + table fit the CU's 160 KiB of LDS; 0 = does not fit (row kernels).
int pick_fs(int n_max, int W) {
    if (n_max >= 65535 || (W != 8 && W != 16)) return 0;
    static const int cap = getenv("GMC_LDS_MAX_FS") ? atoi(getenv("GMC_LDS_MAX_FS")) : 64;  // tuning runs only
    for (int fs = cap >= 16 ? cap : 64; fs >= 16; fs >>= 1)
        if (lds_bytes(n_max, W, fs) <= 160 * 1024) return fs;
    return 0;
}

template <typename K, typename A>
int launch(K k, int grid, size_t lds, hipStream_t st, const A &args) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, st, args);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

template <int FS, int W>
int launch_spmm(const TileArgs &a, size_t lds, hipStream_t st) {
    constexpr int rows_per_pass = kThreads / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    const int grid = a.b.B * a.groups;
    const bool epi = a.Zpart != nullptr;
#define GMC_PICK(AC)                                                                                          \
    do {                                                                                                      \
        if (a.shared_src)  /* W1 gather: weights apply, never fused with W2 */                               \
            return a.use_vals ? launch(spmm_lds_kernel<FS, W, AC, false, true, true>, grid, lds, st, a)       \
                              : launch(spmm_lds_kernel<FS, W, AC, false, false, true>, grid, lds, st, a);     \
        if (a.use_vals) return epi ? launch(spmm_lds_kernel<FS, W, AC, true, true, false>, grid, lds, st, a)  \
                                   : launch(spmm_lds_kernel<FS, W, AC, false, true, false>, grid, lds, st, a);\
        return epi ? launch(spmm_lds_kernel<FS, W, AC, true, false, false>, grid, lds, st, a)                 \
                   : launch(spmm_lds_kernel<FS, W, AC, false, false, false>, grid, lds, st, a);               \
    } while (0)
    if (acc <= 4) GMC_PICK(4);
    if (acc <= 8) GMC_PICK(8);
#undef GMC_PICK
    return GMC_ERR_UNSUPPORTED;
}

template <int FS, int W>
int launch_dw1(const Dw1TileArgs &a, size_t lds, hipStream_t st) {
    constexpr int rows_per_pass = kThreads / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    const int grid = a.slices * a.chunks;
    const bool hv = a.b.ell_vals != nullptr;
    if (acc <= 4) return hv ? launch(dw1_lds_kernel<FS, W, 4, true>, grid, lds, st, a)
                            : launch(dw1_lds_kernel<FS, W, 4, false>, grid, lds, st, a);
    if (acc <= 8) return hv ? launch(dw1_lds_kernel<FS, W, 8, true>, grid, lds, st, a)
                            : launch(dw1_lds_kernel<FS, W, 8, false>, grid, lds, st, a);
    return GMC_ERR_UNSUPPORTED;
}

}  // namespace

// slice width the LDS kernels (and the slab layout) use for this batch; 0 = row kernels
int gmc_lds_slice_width(const gmc_batch *b) { return b->ell ? pick_fs(b->n_max, b->ell_width) : 0; }

bool gmc_lds_fits(const gmc_batch *b) {
    if (!b->ell) return false;
    const int fs = pick_fs(b->n_max, b->ell_width);
    if (!fs) return false;
    const int rows_per_pass = kThreads / (fs / 4);
    return (b->n_max + rows_per_pass - 1) / rows_per_pass <= 8;
}

bool gmc_bwd1_fits(const gmc_batch *b) { return gmc_lds_fits(b); }

int device_cus(bool allow_override = true);

// column slices of the LDS-tiled kernels for F columns (0: graphs do not fit)
int gmc_lds_slices(const gmc_batch *b, int F) {
    const int fs = pick_fs(b->n_max, b->ell_width);
    return fs ? (F + fs - 1) / fs : 0;
}

// Slice groups per graph == Zpart partials of the fused W2 epilogue (one workgroup, or one item of a
// persistent workgroup, per group).  4 slices per group when that yields at least half a workgroup
// per CU, else 2, else 1: a single n=1000 graph gets 32 groups instead of 8, a batch of 20 n=500
// graphs 160 instead of 80.  The choice depends on the batch size only through these three classes,
// so results are bitwise reproducible run to run and independent of WHICH graphs share a batch;
// between classes the W2 partials are folded in a different association (last-ulp differences).
// GMC_LDS_SLICES_PER_WG overrides (tuning runs only).
int gmc_lds_groups(const gmc_batch *b, int F) {
    const int fs = pick_fs(b->n_max, b->ell_width);
    if (!fs) return 0;
    const int slices = (F + fs - 1) / fs;
    static const int per_env = getenv("GMC_LDS_SLICES_PER_WG") ? atoi(getenv("GMC_LDS_SLICES_PER_WG")) : 0;
    int per = 4;
    if (per_env > 0) {
        per = per_env;
    } else {
        const long want = device_cus(false) / 2;  // the hardware's CUs (the test override only re-splits items)
        while (per > 1 && (long)b->B * ((slices + per - 1) / per) < want) per >>= 1;
    }
    if (per > kMaxSlicesPerWg) per = kMaxSlicesPerWg;
    return (slices + per - 1) / per;
}

// Y = act(scale * A_g @ X + bias) for every graph of the batch, LDS-staged; optional fused
// Zpart[group][r][:] = scale[r] * (Y[r, group's columns] @ W2[group's rows]).
// x_slab / y_slab: the operand uses the slab layout [slice][R][FS] (ld ignored) instead of
// row-major with leading dimension ld.
int gmc_spmm_lds_launch(const gmc_batch *b, const float *X, long ldx, int x_slab, int shared_src, int use_vals,
                        const float *scale, const float *bias, int relu, float *Y, long ldy, int y_slab, int F,
                        const float *W2, float *Zpart, int tag, hipStream_t st) {
    if (!b || !X || !Y) return GMC_ERR_NULL;
    if (F % 4 || ldx % 4 || ldy % 4 || !gmc_aligned16(X) || !gmc_aligned16(Y))
        return GMC_ERR_ALIGN;
    if (!gmc_lds_fits(b)) return GMC_ERR_UNSUPPORTED;
    if (b->B == 0) return GMC_OK;
    const int fs = pick_fs(b->n_max, b->ell_width);
    const long slab_ss = (long)b->R * fs;
    TileArgs a{*b, X, x_slab ? fs : ldx, x_slab ? slab_ss : fs, shared_src, use_vals && b->ell_vals != nullptr,
               scale, bias, relu, Y, y_slab ? fs : ldy, y_slab ? slab_ss : fs, F,
               (F + fs - 1) / fs, gmc_lds_groups(b, F), W2, Zpart, 0};
    const size_t lds = lds_bytes(b->n_max, b->ell_width, fs);
    GmcProbeScope probe(tag, st);
    if (b->ell_width == 8) {
        switch (fs) {
            case 64: return launch_spmm<64, 8>(a, lds, st);
            case 32: return launch_spmm<32, 8>(a, lds, st);
            default: return launch_spmm<16, 8>(a, lds, st);
        }
    }
    switch (fs) {
        case 64: return launch_spmm<64, 16>(a, lds, st);
        case 32: return launch_spmm<32, 16>(a, lds, st);
        default: return launch_spmm<16, 16>(a, lds, st);
    }
}

template <int FS, int W>
int launch_bwd1(const Bwd1Args &a, size_t lds, hipStream_t st) {
    constexpr int rows_per_pass = kThreads / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    const int grid = a.slices * a.chunks;
    const bool hv = a.b.ell_vals != nullptr;
    if (acc <= 4) return hv ? launch(bwd1_lds_kernel<FS, W, 4, true>, grid, lds, st, a)
                            : launch(bwd1_lds_kernel<FS, W, 4, false>, grid, lds, st, a);
    if (acc <= 8) return hv ? launch(bwd1_lds_kernel<FS, W, 8, true>, grid, lds, st, a)
                            : launch(bwd1_lds_kernel<FS, W, 8, false>, grid, lds, st, a);
    return GMC_ERR_UNSUPPORTED;
}

template <int FS, int W>
int launch_fwd1(const TileArgs &a, size_t lds, int grid, hipStream_t st) {
    constexpr int rows_per_pass = kThreads / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    if (acc <= 4) return a.use_vals ? launch(fwd1_lds_kernel<FS, W, 4, true>, grid, lds, st, a)
                                    : launch(fwd1_lds_kernel<FS, W, 4, false>, grid, lds, st, a);
    if (acc <= 8) return a.use_vals ? launch(fwd1_lds_kernel<FS, W, 8, true>, grid, lds, st, a)
                                    : launch(fwd1_lds_kernel<FS, W, 8, false>, grid, lds, st, a);
    return GMC_ERR_UNSUPPORTED;
}

// compute units of the current device (persistent kernels launch one workgroup per CU)
int device_cus(bool allow_override) {
    if (const char *e = allow_override ? getenv("GMC_DEVICE_CUS") : nullptr) {  // tests: force long item ranges
        const int v = atoi(e);
        if (v > 0) return v;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    return cus;
}

// fused layer-1 forward: H (slab layout) = relu(dinv o (A @ (dinv o (A_val @ W1[:n]))) + b1) and
// Zpart[group][r][:] = dinv[r] * (H[r, group's columns] @ W2[group's rows])
int gmc_fwd1_lds_launch(const gmc_batch *b, const float *W1, const float *b1, const float *W2, float *H,
                        float *Zpart, int F, hipStream_t st) {
    if (!gmc_lds_fits(b)) return GMC_ERR_UNSUPPORTED;
    if (b->B == 0) return GMC_OK;
    const int fs = pick_fs(b->n_max, b->ell_width);
    const int slices = (F + fs - 1) / fs, groups = gmc_lds_groups(b, F);
    // the column constants of every slice sit in LDS: 16 B per (padded) column
    if ((size_t)slices * fs > (size_t)kThreads || (size_t)16 * slices * fs > lds_consts(b->n_max, fs)) return GMC_ERR_UNSUPPORTED;
    // contiguous ranges of (graph, group) items, one persistent workgroup per CU when there are enough
    const int total = b->B * groups, cus = device_cus();
    const int ipw = (total + cus - 1) / cus, grid = (total + ipw - 1) / ipw;
    TileArgs a{*b, W1, (long)F, (long)fs, 1, b->ell_vals != nullptr, b->dinv, b1, 1, H, (long)fs, (long)b->R * fs,
               F, slices, groups, W2, Zpart, ipw};
    const size_t lds = lds_bytes(b->n_max, b->ell_width, fs);
    GmcProbeScope probe(GMC_K_FWD1_FUSED, st);
    if (b->ell_width == 8) {
        switch (fs) {
            case 64: return launch_fwd1<64, 8>(a, lds, grid, st);
            case 32: return launch_fwd1<32, 8>(a, lds, grid, st);
            default: return launch_fwd1<16, 8>(a, lds, grid, st);
        }
    }
    switch (fs) {
        case 64: return launch_fwd1<64, 16>(a, lds, grid, st);
        case 32: return launch_fwd1<32, 16>(a, lds, grid, st);
        default: return launch_fwd1<16, 16>(a, lds, grid, st);
    }
}

// fused layer-1 backward over the slab-layout H: dW1 partials [chunks][n_max][F] and column
// partials [chunks][F][4] (dW2, db1)
int gmc_bwd1_lds_launch(const gmc_batch *b, const float *H, const float *GY2, const float *W2,
                        float *dw1part, float *colpart, int F, int chunks, int graphs_per_chunk,
                        hipStream_t st) {
    if (!gmc_lds_fits(b)) return GMC_ERR_UNSUPPORTED;
    const int fs = pick_fs(b->n_max, b->ell_width);
    Bwd1Args a{*b, H, GY2, W2, dw1part, colpart, F, (F + fs - 1) / fs, chunks, graphs_per_chunk};
    const size_t lds = lds_bytes(b->n_max, b->ell_width, fs);
    GmcProbeScope probe(GMC_K_BWD1_FUSED, st);
    if (b->ell_width == 8) {
        switch (fs) {
            case 64: return launch_bwd1<64, 8>(a, lds, st);
            case 32: return launch_bwd1<32, 8>(a, lds, st);
            default: return launch_bwd1<16, 8>(a, lds, st);
        }
    }
    switch (fs) {
        case 64: return launch_bwd1<64, 16>(a, lds, st);
        case 32: return launch_bwd1<32, 16>(a, lds, st);
        default: return launch_bwd1<16, 16>(a, lds, st);
    }
}

// dW1 partials: out[chunk][v][:] = sum_{g in chunk} sum_e vals[e] * U[g][nbr(e), :], v < n_max
int gmc_dw1_lds_launch(const gmc_batch *b, const float *U, long ldu, int u_slab, float *out, int F, int chunks,
                       int graphs_per_chunk, hipStream_t st) {
    if (!gmc_lds_fits(b)) return GMC_ERR_UNSUPPORTED;
    const int fs = pick_fs(b->n_max, b->ell_width);
    Dw1TileArgs a{*b, U, u_slab ? fs : ldu, u_slab ? (long)b->R * fs : fs, out, F, (F + fs - 1) / fs, chunks,
                  graphs_per_chunk};
    const size_t lds = lds_bytes(b->n_max, b->ell_width, fs);
    GmcProbeScope probe(GMC_K_DW1, st);
    if (b->ell_width == 8) {
        switch (fs) {
            case 64: return launch_dw1<64, 8>(a, lds, st);
            case 32: return launch_dw1<32, 8>(a, lds, st);
            default: return launch_dw1<16, 8>(a, lds, st);
        }
    }
    switch (fs) {
        case 64: return launch_dw1<64, 16>(a, lds, st);
        case 32: return launch_dw1<32, 16>(a, lds, st);
        default: return launch_dw1<16, 16>(a, lds, st);
    }
}
