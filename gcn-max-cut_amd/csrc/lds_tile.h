// Shared pieces of the LDS-tiled kernels (spmm_lds.hip, fwd1_lds.hip, bwd1_lds.hip): workgroup shape,
// LDS budget, the LDS-DMA tile loader, the neighbour gathers and the launch helper.  Everything lives in an
// anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "gmc_common.h"
#include <stdlib.h>


// Diagnostic build only (-DGMC_STAMP, `make stamp`): EVERY wave of the first 256 workgroups accumulates the
// shader-clock cycles it spends in each phase of the fused kernels' tile loop into g_stamps[(block, wave)][12]
// (never read by any kernel); gmc_debug_read_stamps copies them out.  (Wave 0 alone gives a skewed picture: the
// LDS serves the oldest wave first, so it reaches every barrier early and waits the longest.)  The production library
// contains none of this.
#ifdef GMC_STAMP
static __device__ unsigned long long g_stamps[4096 * 16];  // one copy per translation unit (no device linking)
#ifdef GMC_MARKS_ONLY  // wall-clock marks without the per-phase cycle counters (which cost a few % themselves)
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#else
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
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && threadIdx.x < 1024)         \
            for (int i = 0; i < 12; ++i) g_stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + i] = st_acc[i]; \
    } while (0)
#endif
// wall-clock marks (s_memrealtime: one 100 MHz counter for the whole chip) in slots 12..15 of the workgroup:
// kernel entry, tile loop start, tile loop end, kernel exit - where a launch's fixed time goes
#define MARK(i)                                                                                  \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_stamps[blockIdx.x * 256 + 12 + (i)] = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#else
#define MARK(i)
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

// OVF flavours of the 8-slot kernels: 1 = read a row's slots four at a time (gather_ids8_halves: fewer registers in
// flight), 0 = the plain kernels' gathers (eight reads in flight per row).  The backward needs the former to stay
// free of scratch; the forward does not any more (its hub rows are a register bit now, not LDS descriptors).
#ifndef GMC_OVF_HALVES_FWD
#define GMC_OVF_HALVES_FWD 0
#endif
#ifndef GMC_OVF_HALVES_BWD
#define GMC_OVF_HALVES_BWD 1
#endif

namespace {

// threads per workgroup of every kernel in this file; GMC_LDS_THREADS=512 builds the tuning variant
// (two co-resident workgroups per CU when their LDS fits) - never the shipped library
#ifndef GMC_LDS_THREADS
#define GMC_LDS_THREADS 1024
#endif
constexpr int kThreads = GMC_LDS_THREADS;
// waves per SIMD the register allocation is sized for: 4 (128 VGPRs per lane) for one 1024-thread workgroup
// per CU; GMC_LDS_WAVES_PER_EU overrides (tuning builds: 512 threads with 2 = 256 VGPRs, or 4 for two
// co-resident workgroups)
#ifndef GMC_LDS_WAVES_PER_EU
#define GMC_LDS_WAVES_PER_EU 4
#endif
#define GMC_LDS_BOUNDS __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(GMC_LDS_WAVES_PER_EU, GMC_LDS_WAVES_PER_EU)))

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
    int x_slab16;        // fwd1: X is the 16-column slab copy of the shared table (gmc::slab16_index) with
    int x_rows;          //       x_rows rows; 0 = row-major X (x_rs)
    int ovf_cap;         // fwd1, OVF kernels: overflow blocks that fit into the LDS behind the kernel's regions
    int own_lds;         //       ... which end at this byte offset
};

// A graph offset (or any wave-uniform int of a read-only table) through the SCALAR cache.  Inside the tile loops the
// compiler will not use s_load for such a read: the LDS-DMA's inline asm clobbers memory, so it falls back to a VECTOR
// load - whose wait is a vmcnt wait, i.e. a wait for whatever tile or table traffic is in flight (round 2's "loop top:
// 9-12 % of the backward's cycles").  The tables read this way are never written by a kernel.
__device__ __forceinline__ int sload(const int32_t *p, int index) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p + index) : "memory");
    return v;
}

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
// third LDS region (after the two tiles and the table):
//   8-slot tables - the larger of
//   - the column constants (bias, W2 rows): 16 B per column - every slice of up to 1024 columns for
//     the persistent fused forward, kMaxSlicesPerWg slices for the SpMM (8 * FS * 16 B <= that);
//   - the per-row constants (GY2[r,:], dinv[r]) of the graph in flight in bwd1_reg: 16 B per row (its second
//     copy of them sits in the table region: that kernel keeps the neighbour ids in registers);
//   16-slot tables (the table is twice as large: 32 B per row) - only the column constants of the slices in
//   flight: kMaxSlicesPerWg slices for the SpMM; two slices (double-buffered, loaded slice by slice) for the
//   fused forward; the fused backward stages its row constants through the U tile's buffer, which is idle
//   while they are needed.  That is what lets n = 1000 graphs of degree 9..16 stay on the LDS path:
//   2 x 64,256 + 32,000 + 2,048 = 162,560 B of the 163,840.
// (bwd1's cross-wave fold area of 256 * FS B re-uses a tile buffer after its graph loop).
__host__ __device__ inline size_t lds_consts(int n_max, int FS, int W) {
    if (W == 16) return (size_t)16 * kMaxSlicesPerWg * FS;
    const size_t cols = (size_t)16 * 1024, rows = (size_t)16 * (n_max + kPadRows);
    return cols > rows ? cols : rows;
}
size_t lds_bytes(int n_max, int W, int FS) {
    return 2 * tile_floats(n_max, FS) * 4 + (size_t)n_max * W * 2 + lds_consts(n_max, FS, W);
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
//
// NT = the "nt" (non-temporal) cache policy: for bytes this launch reads exactly once (the H / X rows of the batch),
// so that they do not push the re-read data (W1, the small per-row tables, the partials the next kernel folds) out
// of L2 / MALL.  Measured on the 160-graph step: fused backward 117 -> 110 us, and the two small kernels after it
// 1.5 us faster each.  Never for a tile every CU re-reads (the W1 slices of the forward).
template <bool NT = false>
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// 16-byte store of a result row that this launch does not read again and that is far larger than the caches by the
// time anyone does (H: 320 MB per step): non-temporal, for the same reason
__device__ __forceinline__ void store_nt(float *dst, const float4 v) {
    __builtin_nontemporal_store((gmc::v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<gmc::v4f *>(dst));
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

// FRESH: derive the wave's LDS destination from the thread id anew on every call (kernels under register pressure:
// hoisted out of the slice loop the value is spilled, and its reload in front of the first piece is a vmcnt(0) wait)
template <int FS, int ACC, bool NT = false, bool FRESH = false>
__device__ __forceinline__ void dma_tile(const float *src_q, long rs, int n, bool col_on, int lrow, float *tile) {
    constexpr int kRowsPerPass = kThreads / (FS / 4);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)tile;
    unsigned tid = threadIdx.x;
    if (FRESH) asm volatile("" : "+v"(tid));
    const unsigned wave_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 16u * (tid & ~63u)));
#pragma unroll
    for (int k = 0; k < ACC; ++k) {
        const int l = lrow + k * kRowsPerPass;
        if (l < n && col_on) glds16<NT>(src_q + (long)l * rs, wave_dst + 16u * (unsigned)(k * kThreads));
    }
}


// Read the tile rows named by the first NS of eight packed u16 ids (lane's 16 B of each row).  The byte
// address id * row_bytes + (tile + 16 q) is one v_mad_u32_u16 per row (op_sel picks the id's half
// of the dword) instead of the unpack + shift-add pair the compiler emits: the gathers spend about
// as many SIMD cycles on address arithmetic and adds as LDS cycles on the reads.
//
// NS = neighbour slots in use (1..8 of this block of eight): for a batch whose largest degree is <= 7 - d = 7
// regular graphs, the headline workload - gmc_ell_arrange_host keeps slot 7 of every row for padding
// (gmc_batch.ell_slots == 7) and the gathers neither read that row of zeros nor add it: 1/8 fewer LDS reads and
// packed adds in all four gathers of a training step.  16-slot tables likewise: d = 12 reads 8 + 4 slots.
template <int NS = 8>
__device__ __forceinline__ void read8(const float *tile, int q, unsigned row_bytes, const uint4 ids, float4 (&x)[8]) {
    static_assert(NS >= 1 && NS <= 8, "slots of one block of eight");
    typedef float v4f __attribute__((ext_vector_type(4)));
    using lds_f4 = __attribute__((address_space(3))) const v4f;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)tile + 16u * (unsigned)q;
    const unsigned pk[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned lo, hi;
        if (2 * j >= NS) break;
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(lo) : "v"(pk[j]), "s"(row_bytes), "v"(base));
        const v4f a = *(lds_f4 *)(size_t)lo;
        x[2 * j] = make_float4(a.x, a.y, a.z, a.w);
        if (2 * j + 1 < NS) {
            asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(pk[j]), "s"(row_bytes), "v"(base));
            const v4f b = *(lds_f4 *)(size_t)hi;
            x[2 * j + 1] = make_float4(b.x, b.y, b.z, b.w);
        }
    }
}

// four tile rows named by two packed id pairs (the 16-slot flavours read a row's slots four at a time: with eight
// rows of reads in flight next to everything else those kernels hold, the compiler spilled their staging registers
// right behind the loads - a vmcnt wait in the shadow of the tile DMA - and reloaded values inside the gathers)
template <int NS4>
__device__ __forceinline__ void read4(const float *tile, int q, unsigned row_bytes, unsigned pk0, unsigned pk1, float4 (&x)[4]) {
    static_assert(NS4 >= 1 && NS4 <= 4, "slots of one half block");
    typedef float v4f __attribute__((ext_vector_type(4)));
    using lds_f4 = __attribute__((address_space(3))) const v4f;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)tile + 16u * (unsigned)q;
    const unsigned pk[2] = {pk0, pk1};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned lo, hi;
        if (2 * j >= NS4) break;
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(lo) : "v"(pk[j]), "s"(row_bytes), "v"(base));
        const v4f a = *(lds_f4 *)(size_t)lo;
        x[2 * j] = make_float4(a.x, a.y, a.z, a.w);
        if (2 * j + 1 < NS4) {
            asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(pk[j]), "s"(row_bytes), "v"(base));
            const v4f b = *(lds_f4 *)(size_t)hi;
            x[2 * j + 1] = make_float4(b.x, b.y, b.z, b.w);
        }
    }
}

// sum over the row's first NS of W neighbour slots (slot order, padding -> zero row) from the LDS tile
template <int FS, int W, bool HAS_VAL, int NS = W>
__device__ __forceinline__ float4 gather_row(const float *tile, const unsigned short *nb, const float *wrow,
                                             int l, int q) {
    static_assert((W == 8 && (NS == 7 || NS == 8)) || (W == 16 && NS > 8 && NS <= 16), "live slots of the table");
    float4 acc = gmc::f4_zero();
    if constexpr (W == 16) {   // four slots at a time
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            constexpr int kDummy = 0; (void)kDummy;
            const int live = NS - 4 * h >= 4 ? 4 : (NS - 4 * h > 0 ? NS - 4 * h : 0);
            if (live <= 0) break;
            const uint2 ids = *reinterpret_cast<const uint2 *>(nb + (long)l * W + h * 4);
            float4 x[4];
            if (live == 4) read4<4>(tile, q, FS * 4, ids.x, ids.y, x);
            else if (live == 3) read4<3>(tile, q, FS * 4, ids.x, ids.y, x);
            else if (live == 2) read4<2>(tile, q, FS * 4, ids.x, ids.y, x);
            else read4<1>(tile, q, FS * 4, ids.x, ids.y, x);
            if (HAS_VAL) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wrow + h * 4);
                const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < live) gmc::f4_fma(acc, w[u], x[u]);
            } else if (h == 0) {
                acc = x[3];   // (NS > 8: the first half block is always full) last read first: one wait per half block
#pragma unroll
                for (int u = 2; u >= 0; --u) gmc::f4_add(acc, x[u]);
            } else {
#pragma unroll
                for (int u = 3; u >= 0; --u)
                    if (u < live) gmc::f4_add(acc, x[u]);
            }
            // keep the half blocks apart: left alone the scheduler batches all sixteen reads again
            asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
        }
        return acc;
    }
#pragma unroll
    for (int blk = 0; blk < W / 8; ++blk) {
        constexpr int kFirst = NS < 8 ? NS : 8;
        const uint4 ids = *reinterpret_cast<const uint4 *>(nb + (long)l * W + blk * 8);
        float4 x[8];
        if (blk == 0) read8<kFirst>(tile, q, FS * 4, ids, x);
        else read8<(NS > 8 ? NS - 8 : 1)>(tile, q, FS * 4, ids, x);
        const int live = blk == 0 ? kFirst : NS - 8;
        if (HAS_VAL) {  // weights come from HBM/L2: this (rare) variant waits on them per row
            const float4 w0 = *reinterpret_cast<const float4 *>(wrow + blk * 8);
            const float4 w1 = *reinterpret_cast<const float4 *>(wrow + blk * 8 + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < live) gmc::f4_fma(acc, w[u], x[u]);
        } else if (blk == 0) {  // last read first: one wait per block (see gather_ids8)
            acc = x[kFirst - 1];
#pragma unroll
            for (int u = kFirst - 2; u >= 0; --u) gmc::f4_add(acc, x[u]);
        } else {
#pragma unroll
            for (int u = 7; u >= 0; --u)
                if (u < live) gmc::f4_add(acc, x[u]);
        }
    }
    return acc;
}

// Overflow lists of the graph in flight (gmc_batch.ovf_*: the neighbours beyond the table's W slots of its rows,
// eight ids per block, padded with the zero row).  Only the OVF instantiations of the fused kernels carry this.
// Everything a gather needs lives in the LDS the launch has left over behind the kernel's own regions (a few KB):
//   desc[n_max]  one 16-bit word per row: (blocks << 12) | first block relative to the graph's first; 0 = none (97-100 %
//                of the rows);
//   blocks[cap]  the graph's blocks of eight ids (gmc_lds_fits admits a batch only if every graph's blocks fit).
// A thread keeps its rows' descriptors (or one bit per row) in registers; the lists themselves are reached through
// LDS-address-space pointers only.  No global load, no spill reload and no flat load may sit on any path of a gather:
// each is a vector-memory operation the compiler fences with vmcnt(0), i.e. a wait for the tile DMA in flight behind
// the gather - in the OVF kernels' first versions ONE hub row in ONE graph of 160 made the step 2.3x slower (219 /
// 267 us against 88 / 112), a G(n, p = 0.01) batch 4.3x (scratch/ovf_probe.py, profiles/r03_ablation.json).
struct OvfLds {
    unsigned desc;    // LDS byte address of desc[]
    unsigned blocks;  // LDS byte address of blocks[]
    int cap;          // blocks that fit
};
constexpr int kOvfMaxBlocksPerRow = 15, kOvfMaxBlocksPerGraph = 4095;   // what a 16-bit descriptor can say
// the LDS an OVF launch asks for (all of it) and how the spare behind the kernel's `own_bytes` is split
constexpr size_t kOvfLdsBytes = 160 * 1024;
__host__ __device__ inline size_t ovf_desc_bytes(int n_max) { return ((size_t)n_max * 2 + 15) & ~(size_t)15; }
inline int ovf_cap_blocks(size_t own_bytes, int n_max) {
    const size_t used = own_bytes + ovf_desc_bytes(n_max);
    return used < kOvfLdsBytes ? (int)((kOvfLdsBytes - used) / 16) : 0;
}
// LDS the fused forward (kernel 0) / backward (kernel 1) themselves use for graphs of n_max nodes: what the OVF
// flavours have to stay behind.  16-slot tables: only the constants these two kernels really keep (the forward two
// slices' worth, the backward none) - n = 1000 then leaves 2.8 / 3.3 KB: the 2 KB of descriptors and 51 / 83 blocks.
inline size_t ovf_own_bytes(int kernel, int n_max, int W, int FS) {
    if (W != 16) return lds_bytes(n_max, W, FS);
    return 2 * tile_floats(n_max, FS) * 4 + (size_t)n_max * 32 + (kernel == 0 ? (size_t)2 * FS * 16 : 0);
}
// do the descriptors and EVERY overflow block of the batch's fullest graph fit behind both kernels' own LDS?  (The
// gathers must not contain a single global load, not even on a rare path: the compiler closes a branch that holds
// one with s_waitcnt vmcnt(0) at the join, which every row then executes - a wait for the tile DMA in flight.)
inline bool ovf_fits(int n_max, int W, int FS, int max_blocks) {
    if (max_blocks > kOvfMaxBlocksPerGraph) return false;   // (small graphs leave LDS for more than a descriptor can address)
    for (int kernel = 0; kernel < 2; ++kernel) {
        const size_t own = ovf_own_bytes(kernel, n_max, W, FS);
        if (own + ovf_desc_bytes(n_max) > kOvfLdsBytes || ovf_cap_blocks(own, n_max) < max_blocks) return false;
    }
    return true;
}
__device__ __forceinline__ OvfLds ovf_lds(const float *lds_base, int own_bytes, int n_max, int cap) {
    // (wave-uniform values: kept in scalar registers)
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)((unsigned)(size_t)(__attribute__((address_space(3))) const float *)lds_base + (unsigned)own_bytes));
    return OvfLds{base, base + (unsigned)ovf_desc_bytes(n_max), cap};
}
// Set up desc[] and blocks[] for the graph at rows [r0, r0 + n) in two steps, so that a caller can put its own global
// reads of the graph between them and pay ONE memory round trip for all of it (+ one for the blocks, whose place
// depends on the first): ovf_request issues the loads (a thread's row(s) - a graph has at most 2 * kThreads rows - and
// the graph's first / last block offset), ovf_commit writes the descriptors and copies the blocks.  Callers separate
// the commit from the gathers that read the previous graph's LDS copy with a barrier on either side.  `tid` =
// threadIdx.x, possibly laundered through an empty asm by the caller (keeps the address arithmetic inside its loop:
// hoisted out as loop invariants, these values were spilled and every reload came with its own s_waitcnt vmcnt(0)).
// Registers: ROWS = rows of a graph per thread (1 for the ACC = 4 flavours: n <= kThreads); one block per thread in
// registers - graphs with more than kThreads blocks (the 16-bit descriptors allow 4095) copy the rest in a loop of
// dependent round trips inside the commit.  Between step 2 and step 3 a thread holds its descriptors (16 bits each)
// and its block: what a kernel carries across a gather is 4 registers (ROWS = 1), then 5.
template <int ROWS>
struct OvfReq {
    int p0[ROWS], p1[ROWS], ob, oe;   // step 1
    uint4 blk;                        // step 2
};
// step 1: the block offsets of my row(s) and of the graph's first / last row
template <int ROWS>
__device__ __forceinline__ void ovf_request(const gmc_batch &b, int r0, int n, int tid, OvfReq<ROWS> &q) {
    q.ob = b.ovf_ptr[r0];
    q.oe = b.ovf_ptr[r0 + n];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int i = min(tid + j * kThreads, n - 1);
        q.p0[j] = b.ovf_ptr[r0 + i];
        q.p1[j] = b.ovf_ptr[r0 + i + 1];
    }
}
// step 2 (needs step 1's answer): my block of the graph's first kThreads.  Unconditional (a predicated load gets its
// own wait at the join): indices past the graph's last block re-read that block, a graph without blocks reads block
// 0 of the batch (an OVF launch has at least one).  The row offsets turn into descriptors here (p0[j] <- descriptor).
template <int ROWS>
__device__ __forceinline__ void ovf_blocks(const gmc_batch &b, int tid, const OvfLds &o, OvfReq<ROWS> &q) {
    const int ob = __builtin_amdgcn_readfirstlane(q.ob);
    const int nblk = min(__builtin_amdgcn_readfirstlane(q.oe) - ob, o.cap);
    const uint4 *src = reinterpret_cast<const uint4 *>(b.ovf_ids) + (nblk > 0 ? ob : 0);
    q.blk = src[min(tid, max(nblk - 1, 0))];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) q.p0[j] = q.p1[j] > q.p0[j] ? ((q.p1[j] - q.p0[j]) << 12) | (q.p0[j] - ob) : 0;
    q.ob = ob; q.oe = nblk;   // (wave-uniform from here on)
}
// step 3: descriptors and blocks into LDS
template <int ROWS>
__device__ __forceinline__ void ovf_commit(const gmc_batch &b, int n, int tid, const OvfLds &o, const OvfReq<ROWS> &q) {
    using lds_u16 = __attribute__((address_space(3))) unsigned short;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    using lds_u4 = __attribute__((address_space(3))) u4;
    const int ob = __builtin_amdgcn_readfirstlane(q.ob), nblk = __builtin_amdgcn_readfirstlane(q.oe);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int i = tid + j * kThreads;
        if (i < n) *(lds_u16 *)(size_t)(o.desc + 2u * (unsigned)i) = (unsigned short)q.p0[j];
    }
    if (tid < nblk) *(lds_u4 *)(size_t)(o.blocks + 16u * (unsigned)tid) = (u4){q.blk.x, q.blk.y, q.blk.z, q.blk.w};
    if (nblk > kThreads) {   // (wave-uniform, rare: a graph with more than 1024 overflow blocks)
        const uint4 *src = reinterpret_cast<const uint4 *>(b.ovf_ids) + ob;
        for (int i = tid + kThreads; i < nblk; i += kThreads) {
            const uint4 v = src[i];
            *(lds_u4 *)(size_t)(o.blocks + 16u * (unsigned)i) = (u4){v.x, v.y, v.z, v.w};
        }
    }
}
template <int ROWS>
__device__ __forceinline__ void ovf_setup(const gmc_batch &b, int r0, int n, const OvfLds &o) {
    OvfReq<ROWS> q;
    ovf_request(b, r0, n, (int)threadIdx.x, q);
    ovf_blocks(b, (int)threadIdx.x, o, q);
    ovf_commit(b, n, (int)threadIdx.x, o, q);
}
// a row's descriptor word (0: no overflow blocks)
__device__ __forceinline__ unsigned ovf_desc(const OvfLds &o, int l) {
    using lds_u16 = __attribute__((address_space(3))) const unsigned short;
    return *(lds_u16 *)(size_t)(o.desc + 2u * (unsigned)l);
}
// ONE hub row's sum computed by the whole wave: lane (j, q) reads the tile row of the row's j-th, (j + 64/Q)-th, ...
// neighbour for its four columns, a butterfly over j folds them (fixed order: bitwise reproducible); every lane returns
// the sum for its q.  The neighbour list = the row's WT table slots (WT = 0: none - the caller has that part already)
// followed by its overflow blocks (descriptor d, wave-uniform); entries past the end read `zero_row`, a tile row of
// zeros, as the padding ids do.  The ids of round r+1 are requested ahead of the tile reads of round r: under a
// workgroup's gathers every dependent LDS hop costs ~500 cycles of queueing, so a row costs (1 + rounds) hops.
// (Round 3, first version: the row's own four lanes walked the list, two tile reads per step of a dependent chain -
// 2,400 cycles per gather for a degree-40 row with everybody else at the barrier: the workgroup that met the ONE
// hub row of a 160-graph batch took 150 us, the others 110.)
template <int FS, int WT>
__device__ __forceinline__ float4 gather_hub_row_wave(const float *tile, const unsigned short *nb, const OvfLds &o, int l, unsigned d, int q,
                                                      int zero_row) {
    constexpr int Q = FS / 4, J = 64 / Q;
    typedef float v4f __attribute__((ext_vector_type(4)));
    using lds_f4 = __attribute__((address_space(3))) const v4f;
    using lds_u16 = __attribute__((address_space(3))) const unsigned short;
    const unsigned j = (threadIdx.x & 63u) / Q;
    const unsigned ne = (unsigned)WT + 8u * (d >> 12);      // list entries (the last block's padding ids = zero row)
    const unsigned ids = o.blocks + 16u * (d & 0xfffu);
    const unsigned tab = WT > 0 ? (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned short *)nb + 2u * (unsigned)(l * WT) : 0u;
    const unsigned tile_q = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)tile + 16u * (unsigned)q;
    auto id_of = [&](unsigned e) {
        const unsigned ec = min(e, ne - 1u);
        const unsigned addr = ec < (unsigned)WT ? tab + 2u * ec : ids + 2u * (ec - (unsigned)WT);
        return (unsigned)*(lds_u16 *)(size_t)addr;
    };
    v4f acc = (v4f)(0.f);
    unsigned id = id_of(j);
#pragma unroll 1
    for (unsigned e0 = 0; e0 < ne; e0 += J) {               // (wave-uniform trip count)
        const unsigned cur = e0 + j < ne ? id : (unsigned)zero_row;
        if (e0 + J < ne) id = id_of(e0 + J + j);            // next round's ids travel with this round's tile reads
        acc += *(lds_f4 *)(size_t)(tile_q + cur * (unsigned)(FS * 4));
    }
    return make_float4(gmc::xor_tree<32, Q>(acc.x), gmc::xor_tree<32, Q>(acc.y), gmc::xor_tree<32, Q>(acc.z), gmc::xor_tree<32, Q>(acc.w));
}
// For every DISTINCT hub row among this wave's rows of one pass (d_mine: my row's descriptor, 0 = no overflow blocks -
// read by the caller for all its rows at once; l_mine: my row): f(l, t) with the row l (wave-uniform) and its overflow
// sum t (for my q; WT > 0: the row's table slots included) - callers pick it up where l == l_mine and do everything
// else (read-modify-writes, stores) OUTSIDE, for all their hub rows in parallel: the loop is a serial chain per hub row, and with 2-3 %
// hub rows (G(n,p) tails) the fullest wave of a workgroup walks it five times while fifteen others wait at the barrier.
// Runs on scalar branches: a wave without hub rows in the pass falls through.
template <int FS, int WT, typename F>
__device__ __forceinline__ void for_hub_rows(unsigned d_mine, int l_mine, int q, const float *tile, const unsigned short *nb, const OvfLds &o,
                                             int zero_row, F &&f) {
    unsigned long long m = __builtin_amdgcn_ballot_w64(d_mine != 0);
    while (m) {
        const int src = __builtin_ctzll(m);
        const int l = __builtin_amdgcn_readlane(l_mine, src);
        const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)d_mine, src);
        m &= ~__builtin_amdgcn_ballot_w64(l_mine == l);     // every lane of the row (and its clamped duplicates) at once
        f(l, gather_hub_row_wave<FS, WT>(tile, nb, o, l, d, q, zero_row));
    }
}

// the kernels' NS specialisation for a table of W slots of which `slots` can hold a neighbour (gmc_batch.ell_slots)
inline int ns_class(int W, int slots, bool unit_weights) {
    if (!unit_weights || slots <= 0 || slots >= W) return W;
    if (W == 8) return slots <= 7 ? 7 : 8;
    return slots <= 10 ? 10 : slots <= 12 ? 12 : slots <= 14 ? 14 : 16;
}

// gather_row with the row's ids already fetched (W == 8: one uint4).  Callers issue the id read of
// their NEXT row before calling, so that it returns (LDS answers in order) under the same wait as
// this row's reads and no row read ever sits behind an id read of its own.
template <int FS, bool HAS_VAL, int NS = 8>
__device__ __forceinline__ float4 gather_ids8(const float *tile, const uint4 ids, const float *wrow, int q) {
    float4 x[8];
    read8<NS>(tile, q, FS * 4, ids, x);
    float4 acc;
    if (HAS_VAL) {
        const float4 w0 = *reinterpret_cast<const float4 *>(wrow);
        const float4 w1 = *reinterpret_cast<const float4 *>(wrow + 4);
        const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        acc = make_float4(w[0] * x[0].x, w[0] * x[0].y, w[0] * x[0].z, w[0] * x[0].w);
#pragma unroll
        for (int u = 1; u < NS; ++u) gmc::f4_fma(acc, w[u], x[u]);
    } else {
        // Summed from the last read back: the first add then waits for all the reads at once
        // (LDS answers in order) and the row costs one s_waitcnt instead of eight.  Starts from
        // x[NS-1], not 0 + x[NS-1]: the compiler may not drop an add of +0.0.
        acc = x[NS - 1];
#pragma unroll
        for (int u = NS - 2; u >= 0; --u) gmc::f4_add(acc, x[u]);
    }
    return acc;
}

// gather_ids8 for unit weights with the sum written on 2-vectors: 14 v_pk_add_f32 per row wherever it
// is inlined (left to the SLP vectoriser, gather #2 of the fused forward came out as 28 v_add_f32).
template <int FS, int NS = 8>
__device__ __forceinline__ gmc::v4f gather_ids8_pk(const float *tile, const uint4 ids, int q) {
    float4 x[8];
    read8<NS>(tile, q, FS * 4, ids, x);
    gmc::v2f lo = {x[NS - 1].x, x[NS - 1].y}, hi = {x[NS - 1].z, x[NS - 1].w};  // from the last read back: one wait per row
#pragma unroll
    for (int u = NS - 2; u >= 0; --u) {
        lo += (gmc::v2f){x[u].x, x[u].y};
        hi += (gmc::v2f){x[u].z, x[u].w};
    }
    return (gmc::v4f){lo.x, lo.y, hi.x, hi.y};
}

// gather_ids8 for unit weights with the row's eight slots read four at a time (two half blocks kept apart): what the
// OVF flavours of the 8-slot kernels use - they carry a few more live values than the plain kernels, and with eight
// rows of reads in flight that was enough to push values into scratch inside the gathers.
template <int FS>
__device__ __forceinline__ float4 gather_ids8_halves(const float *tile, const uint4 ids, int q) {
    float4 x[4];
    read4<4>(tile, q, FS * 4, ids.x, ids.y, x);
    float4 acc = x[3];
#pragma unroll
    for (int u = 2; u >= 0; --u) gmc::f4_add(acc, x[u]);
    asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
    read4<4>(tile, q, FS * 4, ids.z, ids.w, x);
#pragma unroll
    for (int u = 3; u >= 0; --u) gmc::f4_add(acc, x[u]);
    return acc;
}

// Slice width for graphs of up to n_max nodes with W neighbour slots: the widest slice whose
// two tile buffers + table fit the CU's 160 KiB of LDS; 0 = does not fit (row kernels).
int pick_fs(int n_max, int W) {
    if (n_max >= 65535 || (W != 8 && W != 16)) return 0;
#ifdef GMC_TUNING   // tuning builds only (`make variant DEFS=-DGMC_TUNING`): the shipped library reads no environment
    static const int cap = getenv("GMC_LDS_MAX_FS") ? atoi(getenv("GMC_LDS_MAX_FS")) : 64;
#else
    constexpr int cap = 64;
#endif
    for (int fs = cap >= 16 ? cap : 64; fs >= 16; fs >>= 1)
        if (lds_bytes(n_max, W, fs) <= 160 * 1024) return fs;
    return 0;
}

template <typename K, typename A>
int launch(K k, int grid, size_t lds, hipStream_t st, const A &args) {
    if (lds > 64 * 1024) {
        // the large-LDS opt-in is per kernel, not per launch: remember which kernels have it (a launch is
        // otherwise two runtime calls; single-threaded callers per process, as everywhere in this library)
        static const void *configured[256];
        static int n_configured = 0;
        const void *fn = reinterpret_cast<const void *>(k);
        bool seen = false;
        for (int i = 0; i < n_configured; ++i) seen |= configured[i] == fn;
        if (!seen) {
            (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (n_configured < 256) configured[n_configured++] = fn;
        }
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, st, args);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

}  // namespace

// host-side queries shared by the three translation units (defined in spmm_lds.hip)
int gmc_lds_slice_width(const gmc_batch *b);
bool gmc_lds_fits(const gmc_batch *b);
int gmc_lds_slices(const gmc_batch *b, int F);
int gmc_lds_groups(const gmc_batch *b, int F);
int device_cus(bool allow_override = true);
