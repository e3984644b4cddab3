// Fused layer-1 forward of the GCN (TrainingNeural.py:80-83) for graphs that fit a CU's LDS.
// Shared tile machinery: lds_tile.h.
#include "lds_tile.h"

#ifdef GMC_STAMP
extern "C" int gmc_debug_read_stamps_fwd(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace {

// ---- fused layer-1 forward: W1 row gather + aggregation (+ fused H@W2) in one kernel ---------
//
// Workgroup = (graph, slice group).  Per slice: the W1 slice (rows 0..n-1 of the shared
// [N,F] table, L2-resident) arrives in buffer A by LDS-DMA; gather #1 builds the T0 tile =
// dinv o (A_val @ W1[:n]) in buffer B (the X@W1 of TrainingNeural.py:80 with X = padded
// adjacency, :373); the next slice's W1 tile is then DMA'd into A while gather #2 produces
// H = relu(dinv o (A @ T0) + b1) (:80-81) from B, with the layer-2 feature transform
// (H o dinv) @ W2 (:83) accumulated in registers.  T0 never exists in HBM: the forward of layer 1
// writes H once and reads only W1 (from L2) and the neighbour table.
template <int FS, int W, int ACC, bool HAS_VAL, int NS, bool OVF>
__global__ GMC_LDS_BOUNDS void fwd1_lds_kernel(TileArgs a) {
    STAMP_DECL;
    MARK(0);
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
    // column constants (W2[c,0..2], b1[c]).  8-slot tables: those of every slice, once per workgroup
    // [slices * FS].  16-slot tables (the table takes the room): two slices' worth [2][FS], slice s+1 loaded
    // by the first FS threads while slice s is gathered (CSL)
    constexpr bool CSL = W == 16;
    float4 *cst = reinterpret_cast<float4 *>(nb + (size_t)a.b.n_max * W);
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    constexpr bool ovf = OVF;   // some row of the batch has more than W neighbours: overflow lists (gmc_batch.ovf_*)
    auto col_consts = [&](int cc) {   // (W2[cc,0..2], b1[cc]); zeros for pad columns
        float4 c = gmc::f4_zero();
        if (cc < a.F) {
            if (a.W2) { c.x = a.W2[(long)cc * 3]; c.y = a.W2[(long)cc * 3 + 1]; c.z = a.W2[(long)cc * 3 + 2]; }
            if (a.bias) c.w = a.bias[cc];
        }
        return c;
    };

    if (!CSL) {   // (slices * FS <= 1024 = kThreads: one column per thread; pad columns hold zeros)
        const int cc = threadIdx.x;
        const bool c_on = cc < a.slices * FS;
        float4 c = gmc::f4_zero();
        if (c_on && cc < a.F) {
            if (a.W2) { c.x = a.W2[(long)cc * 3]; c.y = a.W2[(long)cc * 3 + 1]; c.z = a.W2[(long)cc * 3 + 2]; }
            if (a.bias) c.w = a.bias[cc];
        }
        if (c_on) cst[cc] = c;
    }

    MARK(1);
    for (int it = it0; it < it1;) {
        const int g = it / a.groups;
        const int it_end = min(it1, (g + 1) * a.groups);           // my items of graph g
        const int s_lo = (it - g * a.groups) * per, s_hi = min(a.slices, (it_end - g * a.groups) * per);
        const int r0 = a.b.goff[g];
        const int n = a.b.goff[g + 1] - r0;
        const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
        // W1 tile of slice s, row-major [N][ldx].  Pad columns (>= F, last slice only) load the last
        // valid column group again: finite values, so that the masked scale below makes exact zeros
        // Source: the slab copy of W1 when the caller keeps one (a wave's 16 rows x 64 B are then ONE contiguous
        // KiB instead of 16 pieces at a 4F-byte stride: -8 % kernel time), else the row-major table
        auto dma = [&](int s) {
            const int col = s * FS + 4 * q;
            const int cs = min(col, ((a.F + 15) & ~15) - 4);
            const long off = a.x_slab16 ? gmc::slab16_index(0, cs, a.x_rows) : (long)min(col, a.F - 4);
            int lr = lrow;
            // OVF kernels: recompute the pieces' addresses here.  Hoisted out of the slice loop they are spilled, and the
            // reload in front of each piece comes with s_waitcnt vmcnt(0) - a wait for every H store of the previous
            // gather before the next tile's DMA may even start (that, not the hub rows, cost the OVF forward 2x)
            if constexpr (ovf) asm volatile("" : "+v"(lr));
            dma_tile<FS, ACC, false, ovf>(a.X + off, a.x_rs, n, true, lr, bufA);
        };
        if (it != it0) lds_barrier();  // every wave is done with the previous graph's table and tiles
        // graph prologue: every global read is issued before the first use (one memory latency)
        dma(s_lo);
        // overflow lists of this graph: per-row descriptors and the blocks into the spare LDS (published by barrier 1
        // of the first slice; the previous graph's readers are behind the barrier above).  Requested here, committed
        // below behind the prologue's other reads: one round trip for all of them, one more for the blocks.
        // (OVF: the prologue's per-thread addresses are computed from a laundered thread id - as loop invariants of the
        // graph loop they were spilled, and each of the eight reloads per graph came with its own s_waitcnt vmcnt(0).)
        int tid = threadIdx.x;
        if constexpr (ovf) asm volatile("" : "+v"(tid));
        OvfLds ol{};
        constexpr int kOvfRows = (ACC * kRowsPerPass + kThreads - 1) / kThreads;
        OvfReq<kOvfRows> orq;
        if constexpr (ovf) {
            ol = ovf_lds(lds, a.own_lds, a.b.n_max, a.ovf_cap);
            ovf_request(a.b, r0, n, tid, orq);
        }
        float4 cn = gmc::f4_zero();   // CSL: constants of my column of the NEXT slice (threads < FS), in flight
        if (CSL && threadIdx.x < FS) cn = col_consts(s_lo * FS + (int)threadIdx.x);
        float sc[ACC];
#pragma unroll
        for (int k = 0; k < ACC; ++k) sc[k] = a.scale[r0 + min(lrow + k * kRowsPerPass, n - 1)];
        // OVF: the overflow descriptors of my rows (16 bits each: blocks << 12 | first block; 0 = none), from the block
        // offsets themselves, once per graph and kept in registers - the slice loop never reads a descriptor from LDS
        // in front of a branch (under a workgroup's gathers every dependent LDS hop is ~500 cycles of queueing), and a
        // wave none of whose rows is a hub row skips the fix-up loops on a scalar branch
        unsigned hub[(ACC + 1) / 2] = {};   // two 16-bit descriptors per register
        auto desc_of = [&](int k) { return (hub[k / 2] >> (16 * (k & 1))) & 0xffffu; };
        int hp0[ovf ? ACC : 1], hp1[ovf ? ACC : 1];
        if constexpr (ovf) {
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int l = min(lrow + k * kRowsPerPass, n - 1);
                hp0[k] = a.b.ovf_ptr[r0 + l]; hp1[k] = a.b.ovf_ptr[r0 + l + 1];
            }
        }
        uint4 pt[NT];
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.b.ell + (long)r0 * W);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = tid + k * kThreads;
                pt[k] = i < n * (W / 8) ? src[i] : make_uint4(0, 0, 0, 0);
            }
            if constexpr (ovf) {
                ovf_blocks(a.b, tid, ol, orq);
                ovf_commit(a.b, n, tid, ol, orq);
                const int ob = orq.ob;   // (wave-uniform after ovf_blocks)
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    const unsigned dsc = hp1[k] > hp0[k] ? (unsigned)(((hp1[k] - hp0[k]) << 12) | (hp0[k] - ob)) : 0u;
                    hub[k / 2] |= dsc << (16 * (k & 1));
                }
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int i = tid + k * kThreads;
                if (i < n * (W / 8)) reinterpret_cast<uint4 *>(nb)[i] = pt[k];
            }
        }
        if (tid < kPadRows * FS) {  // the zero rows padding entries point at
            bufA[(long)n * FS + tid] = 0.f;
            bufB[(long)n * FS + tid] = 0.f;
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
                    z0 = gmc::xor_tree<Q / 2, 1>(z0); z1 = gmc::xor_tree<Q / 2, 1>(z1); zz = gmc::xor_tree<Q / 2, 1>(zz);
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
        if (CSL && threadIdx.x < FS) cst[(s_lo & 1) * FS + threadIdx.x] = cn;   // (published by barrier 1 of the first slice)
        STAMP(11);  // prologue
        for (int s = s_lo; s < s_hi; ++s) {
            STAMP(0);  // loop overhead / previous tail
            // the W1 tile of slice s has landed once at most the ACC stores issued after its DMA are left
            if (s > s_lo && !ABL(1) && !ABL(2)) {
                vm_wait<ACC>();
                // (CSL: the constants' loads are older than the stores as well; gather #2 of slice s-2, the last
                // reader of this half of the buffer, ended before barrier 1 of slice s-1)
                if (CSL && threadIdx.x < FS) cst[(s & 1) * FS + threadIdx.x] = cn;
            }
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
                    float4 t;
                    if constexpr (GMC_OVF_HALVES_FWD && ovf && !HAS_VAL) t = gather_ids8_halves<FS>(bufA, cur, q);   // (see lds_tile.h)
                    else t = ABL(5) ? make_float4(sc[k], sc[k], sc[k], sc[k])
                                    : gather_ids8<FS, HAS_VAL, NS>(bufA, cur, HAS_VAL ? wbase + (long)lc * W : nullptr, q);
                    t.x *= sc[k]; t.y *= sc[k]; t.z *= sc[k]; t.w *= sc[k];
                    // (OVF: a clamped duplicate must not write - its table-only sum could land on row n-1 AFTER the
                    // owner's fix-up below has added that row's overflow blocks)
                    if (!ovf || l < n) reinterpret_cast<float4 *>(bufB)[lc * Q + q] = t;
                }
            } else {
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    const int l = lrow + k * kRowsPerPass;
                    if (l < n) {
                        float4 t = gather_row<FS, W, HAS_VAL, NS>(bufA, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q);
                        t.x *= sc[k]; t.y *= sc[k]; t.z *= sc[k]; t.w *= sc[k];
                        reinterpret_cast<float4 *>(bufB)[l * Q + q] = t;
                    }
                }
            }
            bool hub_any = false;   // wave-uniform
            if constexpr (ovf) {
                unsigned any = 0;
#pragma unroll
                for (int i = 0; i < (ACC + 1) / 2; ++i) any |= hub[i];
                hub_any = __builtin_amdgcn_ballot_w64(any != 0) != 0;
            }
            if (hub_any) {   // hub rows: add their overflow blocks to the T0 row its lanes have just written
                unsigned dk[ACC];   // my rows' descriptors (rows past n: the thread that OWNS row n-1 adds - a read-modify-write is not idempotent)
#pragma unroll
                for (int k = 0; k < ACC; ++k) dk[k] = lrow + k * kRowsPerPass < n ? desc_of(k) : 0u;
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    const int lm = lrow + k * kRowsPerPass;
                    float4 mine = gmc::f4_zero();
                    for_hub_rows<FS, 0>(dk[k], lm, q, bufA, nullptr, ol, n, [&](int l, const float4 t) { if (lm == l) mine = t; });
                    if (dk[k] != 0) {
                        float4 *cell = reinterpret_cast<float4 *>(bufB) + lm * Q + q;
                        const float4 c = *cell;
                        *cell = make_float4(fmaf(mine.x, sc[k], c.x), fmaf(mine.y, sc[k], c.y), fmaf(mine.z, sc[k], c.z), fmaf(mine.w, sc[k], c.w));
                    }
                }
            }
            STAMP(3);  // gather 1
            loop_barrier();
            STAMP(4);  // barrier 2
            if (s > s_lo && s % per == 0) flush(s / per - 1);  // the group that ended with slice s-1
            if (s + 1 < s_hi && !ABL(1)) {
                dma(s + 1);  // buffer A is free: the next W1 tile streams in during gather #2
                if (CSL && threadIdx.x < FS) cn = col_consts((s + 1) * FS + (int)threadIdx.x);
            }
            STAMP(5);  // DMA issue
            // gather #2: H rows + fused W2; every thread issues exactly ACC stores (rows past n repeat
            // row n-1: same value to the same address) so that the vm_wait above counts exactly
            // my 4 columns' constants (indexed by absolute column): W2 rows as (w0,w1) pairs + w2, bias pairs
            const int cb = (CSL ? (s & 1) : s) * FS + 4 * q;
            const float4 c0 = cst[cb], c1 = cst[cb + 1], c2 = cst[cb + 2], c3 = cst[cb + 3];
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
                int l = min(lrow + k * kRowsPerPass, n - 1);
                // OVF: the store address is computed here - as an invariant of the slice loop the row's 64-bit offset was
                // spilled, and its reload in front of the store came with s_waitcnt vmcnt(0): a wait for the W1 tile just
                // requested, in the middle of the gather that is supposed to hide it
                if constexpr (ovf) asm volatile("" : "+v"(l));
                const gmc::v2f s2 = gmc::splat2(scm[k]);
                const gmc::v2f ylo = gmc::pk_fma((gmc::v2f){acc.x, acc.y}, s2, blo);
                const gmc::v2f yhi = gmc::pk_fma((gmc::v2f){acc.z, acc.w}, s2, bhi);
                float4 y;
                y.x = gmc::relu1(ylo.x); y.y = gmc::relu1(ylo.y); y.z = gmc::relu1(yhi.x); y.w = gmc::relu1(yhi.y);  // F.relu, :81
                if (!ABL(2)) store_nt(ydst + (long)(r0 + l) * a.y_rs, y);
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
                // hub rows (OVF kernels) are emitted by the fix-up loop below, with their
                // overflow blocks: H is relu of the WHOLE sum.  Every thread still stores exactly ACC rows per slice.
                bool later = false;
                if constexpr (ovf) later = desc_of(k) != 0;
                if constexpr (W == 8) {
                    const uint4 cur = ids2;
                    if (k + 1 < ACC) ids2 = reinterpret_cast<const uint4 *>(nb)[min(l + kRowsPerPass, n - 1)];
                    gmc::v4f h;
                    if constexpr (GMC_OVF_HALVES_FWD && ovf) h = gmc::f4v(gather_ids8_halves<FS>(bufB, cur, q));
                    else h = ABL(4) ? (gmc::v4f)(__uint_as_float(cur.x)) : gather_ids8_pk<FS, NS>(bufB, cur, q);
                    if (!later) emit(k, h);
                } else {
                    const gmc::v4f h = gmc::f4v(gather_row<FS, W, false, NS>(bufB, nb, nullptr, l, q));
                    if (!later) emit(k, h);
                }
            }
            if (hub_any) {   // hub rows: table slots and overflow blocks in one list, gathered by the whole wave
#pragma unroll
                for (int k = 0; k < ACC; ++k) {
                    const int lm = min(lrow + k * kRowsPerPass, n - 1);
                    gmc::v4f h = (gmc::v4f)(0.f);
                    for_hub_rows<FS, W>(desc_of(k), lm, q, bufB, nb, ol, n, [&](int l, const float4 t) { if (lm == l) h = gmc::f4v(t); });
                    if (desc_of(k) != 0) emit(k, h);
                }
            }
            STAMP(6);  // gather 2
        }
        flush((s_hi - 1) / per);  // last group of this graph segment
        it = it_end;
    }
    STAMP(7);  // epilogue
    STAMP_FLUSH;
    MARK(3);
}

template <int FS, int W>
int launch_fwd1(const TileArgs &a, size_t lds, int grid, hipStream_t st) {
    constexpr int rows_per_pass = kThreads / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    if (acc > 8) return GMC_ERR_UNSUPPORTED;
    // live slots (no row of the batch has more neighbours): unit-weight kernels skip the others
    const int ns = ns_class(W, a.b.ell_slots, !a.use_vals);
#define GMC_FWD1(AC, HV, NSK, OV) launch(fwd1_lds_kernel<FS, W, AC, HV, NSK, OV>, grid, lds, st, a)
#define GMC_FWD1_ACC(HV, NSK, OV) (acc <= 4 ? GMC_FWD1(4, HV, NSK, OV) : GMC_FWD1(8, HV, NSK, OV))
    if (gmc_has_overflow(&a.b)) {   // hub rows: every slot live (weights + overflow: row kernels, see gmc_lds_fits)
        if (a.use_vals) return GMC_ERR_UNSUPPORTED;
        return GMC_FWD1_ACC(false, W, true);
    }
    if (a.use_vals) return GMC_FWD1_ACC(true, W, false);
    if constexpr (W == 8) return ns == 7 ? GMC_FWD1_ACC(false, 7, false) : GMC_FWD1_ACC(false, 8, false);
    else return ns == 10 ? GMC_FWD1_ACC(false, 10, false) : ns == 12 ? GMC_FWD1_ACC(false, 12, false)
              : ns == 14 ? GMC_FWD1_ACC(false, 14, false) : GMC_FWD1_ACC(false, 16, false);
#undef GMC_FWD1_ACC
#undef GMC_FWD1
}

}  // namespace

// fused layer-1 forward: H (slab layout) = relu(dinv o (A @ (dinv o (A_val @ W1[:n]))) + b1) and
// Zpart[group][r][:] = dinv[r] * (H[r, group's columns] @ W2[group's rows])
// W1_slab (optional): the [ceil(F/16)][N][16] copy of W1 (gmc_w1_slab_f32); N = rows of W1
int gmc_fwd1_lds_launch(const gmc_batch *b, const float *W1, const float *b1, const float *W2, float *H,
                        float *Zpart, int F, hipStream_t st, const float *W1_slab, int N) {
    if (!gmc_lds_fits(b)) return GMC_ERR_UNSUPPORTED;
    if (b->B == 0) return GMC_OK;
    const int fs = pick_fs(b->n_max, b->ell_width);
    const int slices = (F + fs - 1) / fs, groups = gmc_lds_groups(b, F);
    // 8-slot tables: the column constants of every slice sit in LDS, 16 B per (padded) column (16-slot tables load
    // them slice by slice)
    if (b->ell_width == 8 && ((size_t)slices * fs > (size_t)kThreads || (size_t)16 * slices * fs > lds_consts(b->n_max, fs, 8)))
        return GMC_ERR_UNSUPPORTED;
    // contiguous ranges of (graph, group) items, one persistent workgroup per CU when there are enough
    const int total = b->B * groups, cus = device_cus();
    const int ipw = (total + cus - 1) / cus, grid = (total + ipw - 1) / ipw;
    TileArgs a{*b, W1, (long)F, (long)fs, 1, b->ell_vals != nullptr, b->dinv, b1, 1, H, (long)fs, (long)b->R * fs,
               F, slices, groups, W2, Zpart, ipw, 0, 0, 0, 0};
    if (W1_slab) { a.X = W1_slab; a.x_rs = 16; a.x_slab16 = 1; a.x_rows = N; }
    size_t lds = lds_bytes(b->n_max, b->ell_width, fs);
    if (gmc_has_overflow(b)) {   // hub rows: all of the CU's LDS, the spare holds the graph's first overflow blocks
        const size_t own = ovf_own_bytes(0, b->n_max, b->ell_width, fs);
        if (own + ovf_desc_bytes(b->n_max) > kOvfLdsBytes) return GMC_ERR_UNSUPPORTED;   // (gmc_lds_fits says so beforehand)
        a.own_lds = (int)own;
        a.ovf_cap = ovf_cap_blocks(own, b->n_max);
        lds = kOvfLdsBytes;
    }
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
