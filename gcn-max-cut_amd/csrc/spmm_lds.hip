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
#include "lds_tile.h"

namespace {

// ACC = rows per thread (ACC * rows-per-pass >= n_max).
//
// Slice loop, software-pipelined so that no wait ever covers a freshly issued memory op
// (vmcnt retires in order and __syncthreads() drains it):
//   stores of slice s-1 (held in registers)  ->  loads of slice s+1 (into registers)  ->
//   LDS gather of slice s  ->  wait (loads s+1 done; stores s-1 long done)  ->
//   write slice s+1 to the other LDS buffer  ->  barrier.
// SHARED: the source is one table shared by every graph (W1 for the layer-1 feature transform)
template <int FS, int W, int ACC, bool EPI, bool HAS_VAL, bool SHARED, int NS = W>
__global__ GMC_LDS_BOUNDS void spmm_lds_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    // the counted vmcnt wait of the slice loop needs every WAVE to issue exactly ACC store instructions per
    // slice: a wave holds 64 / Q whole rows, i.e. every column group of the slice, and group 0 is always in
    // range - so no wave can be left without an active lane on the last (partial) slice
    static_assert(Q <= 64 && 64 % Q == 0, "a wave must hold whole rows of the tile");
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

    dma_tile<FS, ACC, !SHARED>(src0 + s_beg * a.x_ss - max(0, s_beg * FS + 4 * q - (a.F - 4)), a.x_rs, n, true, lrow, lds);

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
            const float4 acc = gather_row<FS, W, HAS_VAL, NS>(tile, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q);
            // pad columns: the tile holds finite values there (see prefetch), scale 0 and bias 0 make
            // them exact zeros without a per-row select; relu / identity is one v_max either way
            const float sk = (EPI && !col_on) ? 0.f : sc[k];
            float4 y;
            y.x = gmc::max1(fmaf(acc.x, sk, bias.x), lo); y.y = gmc::max1(fmaf(acc.y, sk, bias.y), lo);
            y.z = gmc::max1(fmaf(acc.z, sk, bias.z), lo); y.w = gmc::max1(fmaf(acc.w, sk, bias.w), lo);
            if (col_on) store_nt(ydst + (long)(r0 + l) * a.y_rs, y);
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
        dma_tile<FS, ACC, !SHARED>(src0 + s * a.x_ss - back, a.x_rs, n, true, lrow, lds + into * TF);   // batch rows: read once (nt)
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
            z0 = gmc::xor_tree<Q / 2, 1>(z0); z1 = gmc::xor_tree<Q / 2, 1>(z1); z2 = gmc::xor_tree<Q / 2, 1>(z2);
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

template <int FS, int W, int ACC, bool HAS_VAL, int NS = W>
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
        dma_tile<FS, ACC, true>(a.U + (long)r0 * a.u_rs + (long)s * a.u_ss + 4 * q, a.u_rs, n, col_on, lrow, lds + into * TF);   // U: read once
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
            if (l < n) gmc::f4_add(acc[k], gather_row<FS, W, HAS_VAL, NS>(lds + cur * TF, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q));
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
    // live slots (no row of the batch has more neighbours): the unit-weight aggregation skips the others
    const int ns = ns_class(W, a.b.ell_slots, !a.use_vals && !a.shared_src);
#define GMC_NS(NSK)                                                                                                    \
    do {                                                                                                               \
        if (acc <= 4) return epi ? launch(spmm_lds_kernel<FS, W, 4, true, false, false, NSK>, grid, lds, st, a)        \
                                 : launch(spmm_lds_kernel<FS, W, 4, false, false, false, NSK>, grid, lds, st, a);      \
        if (acc <= 8) return epi ? launch(spmm_lds_kernel<FS, W, 8, true, false, false, NSK>, grid, lds, st, a)        \
                                 : launch(spmm_lds_kernel<FS, W, 8, false, false, false, NSK>, grid, lds, st, a);      \
        return GMC_ERR_UNSUPPORTED;                                                                                    \
    } while (0)
    if constexpr (W == 8) {
        if (ns == 7) GMC_NS(7);
    } else {
        if (ns == 10) GMC_NS(10);
        if (ns == 12) GMC_NS(12);
        if (ns == 14) GMC_NS(14);
    }
#undef GMC_NS
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
    const int ns = ns_class(W, a.b.ell_slots, !hv);
#define GMC_NS(NSK)                                                                        \
    do {                                                                                   \
        if (acc <= 4) return launch(dw1_lds_kernel<FS, W, 4, false, NSK>, grid, lds, st, a); \
        if (acc <= 8) return launch(dw1_lds_kernel<FS, W, 8, false, NSK>, grid, lds, st, a); \
        return GMC_ERR_UNSUPPORTED;                                                        \
    } while (0)
    if constexpr (W == 8) {
        if (ns == 7) GMC_NS(7);
    } else {
        if (ns == 10) GMC_NS(10);
        if (ns == 12) GMC_NS(12);
        if (ns == 14) GMC_NS(14);
    }
#undef GMC_NS
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
    if ((b->n_max + rows_per_pass - 1) / rows_per_pass > 8) return false;
    // overflow lists: their per-row descriptors have to fit behind the fused kernels' own LDS (n <= ~1004 with 8-slot
    // tables at 16-column tiles; larger graphs with hub rows take the row kernels)
    // (and no edge weights: the overflow blocks' weights would have to be read from global memory inside a gather)
    return !gmc_has_overflow(b) || (!b->ell_vals && ovf_fits(b->n_max, b->ell_width, fs, b->ovf_max_blocks));
}

bool gmc_bwd1_fits(const gmc_batch *b) { return gmc_lds_fits(b); }


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
#ifdef GMC_TUNING   // tuning builds only: the shipped library reads no environment
    static const int per_env = getenv("GMC_LDS_SLICES_PER_WG") ? atoi(getenv("GMC_LDS_SLICES_PER_WG")) : 0;
#else
    constexpr int per_env = 0;
#endif
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
    if (!gmc_lds_fits(b) || gmc_has_overflow(b)) return GMC_ERR_UNSUPPORTED;   // (overflow lists: the fused kernels walk them)
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

// compute units of the current device (persistent kernels launch one workgroup per CU)
static int g_cus_override = 0;
// test hook (not in gcnmaxcut.h): pretend the device has `cus` compute units when the persistent forward deals its
// items to workgroups, so that a small batch exercises long item ranges; 0 = the hardware's count
extern "C" int gmc_debug_set_device_cus(int cus) { const int prev = g_cus_override; g_cus_override = cus > 0 ? cus : 0; return prev; }
int device_cus(bool allow_override) {
    if (allow_override && g_cus_override > 0) return g_cus_override;
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

// dW1 partials: out[chunk][v][:] = sum_{g in chunk} sum_e vals[e] * U[g][nbr(e), :], v < n_max
int gmc_dw1_lds_launch(const gmc_batch *b, const float *U, long ldu, int u_slab, float *out, int F, int chunks,
                       int graphs_per_chunk, hipStream_t st) {
    if (!gmc_lds_fits(b) || gmc_has_overflow(b)) return GMC_ERR_UNSUPPORTED;
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
