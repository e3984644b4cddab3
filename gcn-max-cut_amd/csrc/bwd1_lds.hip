// Fused layer-1 backward of the GCN (autograd of TrainingNeural.py:80-83, run by loss.backward() :385)
// for graphs that fit a CU's LDS.  Shared tile machinery: lds_tile.h.
#include "lds_tile.h"
#include "head_body.h"

#ifdef GMC_STAMP
extern "C" int gmc_debug_read_stamps_bwd(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
// (the head's marks when it runs inside this translation unit's one-graph backward)
extern "C" int gmc_debug_read_stamps_bwdhead(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hstamps), sizeof(unsigned long long) * n);
}
#endif

namespace {

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
    int ovf_cap;          // OVF kernels: overflow blocks that fit into the LDS behind the kernel's regions ...
    int own_lds;          // ... which end at this byte offset
    // HEAD kernels (one-graph launches of the reference schedule): the head's arguments - every workgroup computes the
    // graph's head itself, its (GY2, dinv) rows land in the row-constant LDS region and GY2 above is not read
    const float *hZ0;     // [zparts][R][3] slice-group partials of the forward
    int hzparts;
    float hC;
    const float *hb2;
    float *hP;
    int *hS;
    float *hloss;
    float *hdb2part;
    int *htick;
};

// per-thread state shared by the two kernel flavours: my 4 columns as two pairs p = (f0+2p, f0+2p+1) -
// W2 rows and the dW2 / db1 partials live on 2-vectors (v_pk_fma_f32): the transform is VALU work on
// every element of H
struct ColState {
    gmc::v2f w2p[2][3];
    gmc::v2f cdw2[3][2], cdb1[2];  // dW2[pair, k] and db1[pair] partials of my rows
};

__device__ __forceinline__ void col_state_init(ColState &c, const float *W2, int f0, bool col_on) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            c.w2p[p][k] = col_on ? (gmc::v2f){W2[(long)(f0 + 2 * p) * 3 + k], W2[(long)(f0 + 2 * p + 1) * 3 + k]}
                                 : gmc::splat2(0.f);
            c.cdw2[k][p] = gmc::splat2(0.f);
        }
        c.cdb1[p] = gmc::splat2(0.f);
    }
}

// (1) of the graph loop for ONE of my rows: H -> Gs = dinv^2 o relu'(H) o (GY2 @ W2^T) in place + the
// row's dW2 / db1 terms.  rck = (GY2[r,0..2], dinv[r]).  Pad columns need no masks (their W2 rows are 0
// here and the H slab holds exact zeros there); rows past n work on the zero row (h = 0 -> Gs = 0,
// partials += 0): no exec masks either.
__device__ __forceinline__ void transform_row(ColState &c, float4 *cell, const float4 rck, const float4 *hreg = nullptr) {
    const float d = rck.w;
    const float4 h = hreg ? *hreg : *cell;   // (hreg: the cell's H values are already in registers)
    const gmc::v2f g0 = gmc::splat2(rck.x * d), g1 = gmc::splat2(rck.y * d), g2 = gmc::splat2(rck.z * d);
    const gmc::v2f hp[2] = {{h.x, h.y}, {h.z, h.w}};
    gmc::v2f gs[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        gmc::v2f ghd = g0 * c.w2p[p][0];
        ghd = gmc::pk_fma(g1, c.w2p[p][1], ghd);
        ghd = gmc::pk_fma(g2, c.w2p[p][2], ghd);
        const gmc::v2f gpre = {hp[p].x > 0.f ? ghd.x : 0.f, hp[p].y > 0.f ? ghd.y : 0.f};  // relu' o dinv o (GY2 W2^T)
        gs[p] = gpre * gmc::splat2(d);
        if (ABL(3)) continue;
        c.cdw2[0][p] = gmc::pk_fma(hp[p], g0, c.cdw2[0][p]);  // dW2 = (H o dinv)^T GY2
        c.cdw2[1][p] = gmc::pk_fma(hp[p], g1, c.cdw2[1][p]);
        c.cdw2[2][p] = gmc::pk_fma(hp[p], g2, c.cdw2[2][p]);
        c.cdb1[p] += gpre;                                    // db1
    }
    *cell = make_float4(gs[0].x, gs[0].y, gs[1].x, gs[1].y);
    // pin the partials here: their only user is the end of the graph loop, and left alone the optimiser
    // sinks these FMAs past the gathers - keeping every row's h and g alive (600 B of scratch per lane)
#pragma unroll
    for (int p = 0; p < 2; ++p)
        asm volatile("" : "+v"(c.cdw2[0][p]), "+v"(c.cdw2[1][p]), "+v"(c.cdw2[2][p]), "+v"(c.cdb1[p]));
}

// after the graph loop: fold the column partials over the lanes sharing q inside each wave, then over the
// waves through `red` (fixed order), one [F][4] = (dW2[f,0..2], db1[f]) row set per chunk.  Callers
// make sure no wave still reads the buffer `red` aliases.
template <int Q>
__device__ __forceinline__ void col_epilogue(const ColState &c, float *red, float *colpart, int chunk, int s, int FS, int F) {
    float colp[16];  // [j][0..2] dW2, [j][3] db1 for my 4 columns j
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { colp[4 * (2 * p) + k] = c.cdw2[k][p].x; colp[4 * (2 * p + 1) + k] = c.cdw2[k][p].y; }
        colp[4 * (2 * p) + 3] = c.cdb1[p].x; colp[4 * (2 * p + 1) + 3] = c.cdb1[p].y;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) colp[i] = gmc::xor_tree<32, Q>(colp[i]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < Q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * Q + lane) * 16 + i] = colp[i];
    }
    lds_barrier();   // (orders the LDS writes above; the launch's global stores need not be acknowledged here)
    if (threadIdx.x < Q * 4) {  // thread = (q, j): column f = s*FS + 4q + j
        const int qq = threadIdx.x / 4, j = threadIdx.x % 4;
        if (s * FS + 4 * qq < F) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            for (int wv = 0; wv < kThreads / 64; ++wv)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) o[cc] += red[(wv * Q + qq) * 16 + 4 * j + cc];
            reinterpret_cast<float4 *>(colpart)[(long)chunk * F + s * FS + 4 * qq + j] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// the graph's row constants (GY2[r,:], dinv[r]), 16 B per row, by LDS-DMA (lane i -> row i)
__device__ __forceinline__ void dma_row_consts(const float *GY2, int r0, int n, float *gyl) {
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)gyl;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // recompute the per-thread address here: hoisted out of the graph loop it is spilled
    for (int i0 = 0; i0 < n; i0 += kThreads) {
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 16u * (unsigned)(i0 + (tid & ~63))));
        if (i0 + tid < n) glds16(GY2 + (long)(r0 + i0 + tid) * 4, dst);
    }
}

// ---- 8-slot tables (degree <= 8: every reference configuration) -------------------------------------
// A thread's rows' neighbour ids live in registers for the whole graph (both gathers): no table in LDS.
// Two barriers per graph; the transform of graph g+1 follows gather #2 of graph g in the same barrier interval:
//     -> A ->  gather #1(g) (bufA -> bufB)  -> B ->  DMA(g+1) -> bufA || gather #2(g), then transform(g+1)
// (the 16-slot flavour below needs more: its table lives in the LDS the second constants buffer uses)
// * a thread transforms exactly the cells its own DMA instructions fetched, so the H tile of graph g+1
//   needs no rendezvous between "landed" (the issuing wave's vmcnt) and the transform.  The loop can also
//   interleave the transform of piece k (counted vmcnt: pieces retire in issue order) with the rows of gather #2
//   (tuning builds, GMC_BWD1_LEAD) - measured, no gain, see the note at kLead.
// * the row constants are double-buffered (the second buffer is the LDS the 16-slot flavour keeps its
//   table in) and fetched a graph ahead, between A and B, so barrier B publishes them before transform(g+1);
//   (Tried and dropped, measured same-box: pulling the H tile of graph g+2 into L2 with 4-byte LDS-DMA
//   touches while the tile of g+1 streams in, so that the DMA runs at L2 latency - the kernel got 11 %
//   SLOWER.)
// HEAD: a one-graph launch that also computes the graph's head (head_body.h), in every workgroup: no head launch, no
// launch boundary in front of this one, and the H loads of the prologue travel while the head's fold waits for its
// partials.  Workgroup 0 stores the head's outputs (P, S, loss, db2 partial, step counter).
template <int FS, int ACC, bool HAS_VAL, int NS, bool OVF, bool HEAD = false>
__global__ GMC_LDS_BOUNDS void bwd1_reg_kernel(Bwd1Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W = 8;
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    int chunk, s;
    MARK(0);
    tile_of((int)blockIdx.x, a.chunks, a.slices, chunk, s);
    const int TF = (int)tile_floats(a.b.n_max, FS);
    float *bufA = lds, *bufB = lds + TF;
    float *gy0 = lds + 2 * TF + (size_t)a.b.n_max * W / 2;   // the constants region ...
    float *gy1 = lds + 2 * TF;                                // ... and the (unused) table region: 16 B per row each
    float *red = bufB;  // cross-wave fold area, used after the graph loop
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const int f0 = s * FS + 4 * q;
    const bool col_on = f0 < a.F;
    const float *Hs = a.H + (long)s * a.b.R * FS;  // my slice's slab
    ColState cs;
    col_state_init(cs, a.W2, f0, col_on);
    gmc::v4f acc[ACC];
#pragma unroll
    for (int k = 0; k < ACC; ++k) acc[k] = (gmc::v4f)(0.f);

    const int g0 = chunk * a.graphs_per_chunk, g1 = min(a.b.B, g0 + a.graphs_per_chunk);
    if (g0 >= g1) return;
    constexpr bool ovf = OVF;   // some row of the batch has more than 8 neighbours: overflow lists (gmc_batch.ovf_*)
    // first row of my wave in pass 0 (wave-uniform): pass k of the tile DMA is issued by this wave iff that row + k *
    // rows-per-pass is a row of the graph - what the counted waits below have to know
    const int wrow0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u) / Q);

    // KREG of my ACC cells of the next graph's H tile do not come by LDS-DMA after barrier B but by ordinary loads
    // into registers issued a phase EARLIER (behind barrier A, while bufA is still being gathered from): their
    // shadow is gather #1 + gather #2, and the DMA that is left moves (ACC - KREG) / ACC of the tile in gather #2's
    // shadow.  Every wave issues exactly KREG such loads per graph (rows past n re-read row n-1): the wait for the
    // row constants' DMA in front of barrier B counts on it.
    // Measured (round 3, same box, R = 160,000, us per launch): KREG 0: 112.4-114.7 | 1: 113.8-114.0 | 2: 111.7-112.6 |
    // 3: 110.9-111.6.  Worth 1-2 %: the tile's arrival is not what the loop waits for most.
#ifndef GMC_BWD1_KREG
#define GMC_BWD1_KREG 3
#endif
    constexpr int KREG = (!HAS_VAL && !OVF) ? (GMC_BWD1_KREG < ACC ? GMC_BWD1_KREG : ACC) : 0;
    float4 hreg[KREG > 0 ? KREG : 1];
    // RECYCLE (default with KREG > 0): NO LDS-DMA for the tile at all - the cells beyond the first KREG are loaded into
    // the staging registers of cells already transformed: cell j >= KREG into slot j % KREG, behind transform(j - KREG),
    // which rides between the rows of gather #2 (the data of the early cells IS there by then, unlike in the
    // interleaving experiment above).  Every wait for H data is then the compiler's own counted vmcnt on loads it
    // knows; the transform takes its cell from registers (no ds_read), the tile never lands raw in LDS.
#ifndef GMC_BWD1_RECYCLE
#define GMC_BWD1_RECYCLE 1
#endif
    constexpr bool kRecycle = GMC_BWD1_RECYCLE && KREG > 0;
    auto load_cell = [&](int slot, int k, int r0n, int nn) {
        const int l = min(lrow + k * kRowsPerPass, nn - 1);
        const gmc::v4f v = __builtin_nontemporal_load(reinterpret_cast<const gmc::v4f *>(Hs + (long)(r0n + l) * FS + 4 * q));
        hreg[slot] = make_float4(v.x, v.y, v.z, v.w);
    };
    auto load_hreg = [&](int r0n, int nn) {
#pragma unroll
        for (int k = 0; k < KREG; ++k) {
            const int l = min(lrow + k * kRowsPerPass, nn - 1);
            const gmc::v4f v = __builtin_nontemporal_load(reinterpret_cast<const gmc::v4f *>(Hs + (long)(r0n + l) * FS + 4 * q));
            hreg[k] = make_float4(v.x, v.y, v.z, v.w);
        }
    };
    uint4 idr[ACC];
    // overflow lists (OVF): per-row descriptors and the graph's first blocks in the spare LDS (single copy: the OVF
    // flavour pays a third barrier per graph to rewrite it between gather #2 of one graph and gather #1 of the next)
    OvfLds ol{};
    constexpr int kOvfRows = (ACC * kRowsPerPass + kThreads - 1) / kThreads;
    OvfReq<kOvfRows> orq;
    if constexpr (ovf) ol = ovf_lds(lds, a.own_lds, a.b.n_max, a.ovf_cap);
    // rows past n get eight pad ids (the zero row n): their gathers return +0, so the tile loop needs
    // no exec masks - such a thread adds 0 to its dW1 accumulator and writes 0 into the zero row
    // two steps: the loads (nothing uses their data - a use right behind a load is a vmcnt wait, and one placed in the
    // shadow of the tile DMA waits for the tile), then, once they have landed, the pad ids for the rows past n
    auto load_ids = [&](int r0, int n) {
#pragma unroll
        for (int k = 0; k < ACC; ++k)
            idr[k] = *reinterpret_cast<const uint4 *>(a.b.ell + (long)(r0 + min(lrow + k * kRowsPerPass, n - 1)) * W);
    };
    auto pad_ids = [&](int n) {
        const unsigned pad = (unsigned)n * 0x10001u;
#pragma unroll
        for (int k = 0; k < ACC; ++k)
            if (lrow + k * kRowsPerPass >= n) idr[k] = make_uint4(pad, pad, pad, pad);
    };
    auto zero_pads = [&](float *buf, int n) {  // the zero rows n..n+3 the padding entries point at
        if (threadIdx.x < kPadRows * FS) buf[n * FS + threadIdx.x] = 0.f;
    };
    auto fetch_tile = [&](int r0, int n, int k0) {  // H tile of the graph at rows [r0, r0+n) -> bufA, my passes k0..ACC-1
        constexpr int kRows = kThreads / (FS / 4);
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)bufA;
        const unsigned wave_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + 16u * (threadIdx.x & ~63u)));
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRows;
            if (k >= k0 && l < n) glds16<true>(Hs + (long)(r0 + l) * FS + 4 * q, wave_dst + 16u * (unsigned)(k * kThreads));   // H: read once
        }
    };
    // H -> Gs for my cell of pass k of a graph of n rows (row constants from `gyl`).  Rows past n: constants 0 make
    // Gs = 0 and add 0 to the partials whatever (finite) bytes the cell holds - no exec mask, and no dependence on
    // when another wave zeroed the padding rows.
    auto row_consts = [&](int k, int n, const float *gyl) {
        return reinterpret_cast<const float4 *>(gyl)[min(lrow + k * kRowsPerPass, n - 1)];
    };
    auto transform_with = [&](int k, int n, float4 rck) {
        if (ABL(7)) return;
        const int l = lrow + k * kRowsPerPass;
        if (l >= n) rck = gmc::f4_zero();
        const bool from_reg = kRecycle || k < KREG;
        transform_row(cs, reinterpret_cast<float4 *>(bufA) + min(l, n) * Q + q, rck, from_reg ? &hreg[KREG > 0 ? k % KREG : 0] : nullptr);
    };
    auto transform = [&](int k, int n, const float *gyl) { transform_with(k, n, row_consts(k, n, gyl)); };
    // wait until piece k of the tile DMA this wave issued `pieces` pieces of has landed (they retire in issue order;
    // younger loads of the gather only make the wait longer than needed, never shorter)
    auto wait_piece = [&](int k, int pieces) {
        const int younger = pieces - 1 - k;   // wave-uniform
#if defined(GMC_BWD1_FULLWAIT) || !defined(GMC_BWD1_LEAD)   // default: the whole tile before the first transform
        // (the ACC id loads of graph g+1 issued behind gather #2 are younger than the tile and stay in flight)
        if (k == 0) vm_wait<ACC>();
        return;
#endif
        if (younger <= 0) dma_wait();
        else if (younger == 1) vm_wait<1>();
        else if (younger == 2) vm_wait<2>();
        else if (younger == 3) vm_wait<3>();
        else if (younger == 4) vm_wait<4>();
        else if (younger == 5) vm_wait<5>();
        else if (younger == 6) vm_wait<6>();
        else vm_wait<7>();
    };

    // graph offsets are scalar loads: each is requested one graph ahead of its first use
    int r0 = a.b.goff[g0], n = a.b.goff[g0 + 1] - r0;
    int nn = g0 + 1 < g1 ? a.b.goff[g0 + 2] - (r0 + n) : 0;   // size of graph g+1 (0: none)
    if constexpr (!HEAD) dma_row_consts(a.GY2, r0, n, gy0);
    if (!kRecycle) fetch_tile(r0, n, KREG);
    load_hreg(r0, n);
    float4 hlast = gmc::f4_zero();
    if constexpr (kRecycle && ACC == KREG + 1) {
        const int l = min(lrow + (ACC - 1) * kRowsPerPass, n - 1);
        const gmc::v4f v = __builtin_nontemporal_load(reinterpret_cast<const gmc::v4f *>(Hs + (long)(r0 + l) * FS + 4 * q));
        hlast = make_float4(v.x, v.y, v.z, v.w);
    }
    load_ids(r0, n);
    if constexpr (ovf) ovf_setup<kOvfRows>(a.b, r0, n, ol);
    if constexpr (HEAD) {
        // the graph's head, with the H cells / ids above in flight; its LDS arrays lie in the U tile's buffer (free until
        // gather #1), its (GY2, dinv) rows go straight into the row-constant region the transforms below read
        static_assert(kThreads == kHeadThreads && !HAS_VAL && !OVF, "one-graph unit-weight launches only");
        const HeadArgs ha{a.b, a.hZ0, a.hzparts, a.hb2, a.hC, a.hP, a.hS, a.hloss, nullptr, a.hdb2part, a.htick};
        head_body<8>(ha, g0, bufB, blockIdx.x == 0, reinterpret_cast<float4 *>(gy0));
    }
    zero_pads(bufA, n);
    dma_wait();
    __syncthreads();
    STAMP_DECL;
    MARK(1);
    if constexpr (kRecycle && ACC == KREG + 1) {
        // first graph: nothing to hide behind - its last cell was requested with the others into a register of its own
        // (register pressure is low here), so the launch pays ONE memory round trip in front of its first transforms
        // instead of two (a one-graph launch of the reference schedule is mostly prologue and epilogue)
#pragma unroll
        for (int k = 0; k < KREG; ++k) transform(k, n, gy0);
        hreg[0] = hlast;
        transform(ACC - 1, n, gy0);
    } else {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            transform(k, n, gy0);
            if (kRecycle && k + KREG < ACC) load_cell(k % KREG, k + KREG, r0, n);
        }
    }
    for (int g = g0; g < g1; ++g) {
        const int cur = (g - g0) & 1;
        const float *gyl = cur ? gy1 : gy0;
        float *gyn = cur ? gy0 : gy1;
        const int r0n = r0 + n;                                              // == goff[g + 1]
        // (size of graph g+2, 0: none - read at the END of the iteration, through the scalar cache: see sload)
        STAMP(0);
        // The neighbour ids of graph g were requested behind gather #2 of graph g-1 and have landed under the wait
        // for the tile (below).  Tell the compiler HERE - an empty asm that reads them makes it place its own vmcnt for
        // those loads at this point, where nothing else is in flight; left to the first use in gather #1 its wait
        // (a vmcnt(0) at the join of the branch below) also covered the row-constant DMA issued just before the
        // barrier, i.e. every graph paid that DMA's latency in front of barrier A.
#pragma unroll
        for (int k = 0; k < ACC; ++k) asm volatile("" : "+v"(idr[k].x), "+v"(idr[k].y), "+v"(idr[k].z), "+v"(idr[k].w));
        pad_ids(n);
        // row constants of graph g+1 into the other buffer (its last reader, transform(g-1), is two barriers back):
        // published by barrier B, read by transform(g+1) behind it
        if (nn > 0 && !ABL(1)) dma_row_consts(a.GY2, r0n, nn, gyn);
        STAMP(1);
        loop_barrier();
        STAMP(2);  // barrier A: every Gs cell of graph g is written; every wave is done with gather #2 of graph g-1
        // (the transforms of graph g are done with hreg.  Unconditional - the last graph re-reads rows of its own -
        // so that every path issues the same number of loads: a branch here makes the compiler wait for vmcnt(0)
        // at the join, i.e. for these very loads, before the first gather read)
        if (KREG > 0) load_hreg(nn > 0 ? r0n : r0, nn > 0 ? nn : n);
        // OVF: my rows' overflow descriptors, read ONCE per graph - barrier A has published them - and ahead of gather
        // #1: the two fix-up loops find a wave's hub rows by ballot (no LDS round trip in front of a branch), and a
        // wave without hub rows skips them on a scalar branch
        unsigned hd[ovf ? ACC : 1] = {};   // my rows' descriptors (0: no overflow blocks / row past n)
        bool hub_any = false;              // wave-uniform
        if constexpr (ovf) {
#pragma unroll
            for (int k = 0; k < ACC; ++k) hd[k] = ovf_desc(ol, min(lrow + k * kRowsPerPass, n - 1));
            unsigned any = 0;
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                if (lrow + k * kRowsPerPass >= n) hd[k] = 0;
                any |= hd[k];
            }
            hub_any = __builtin_amdgcn_ballot_w64(any != 0) != 0;
        }
        // (2) U tile = dinv o (A @ Gs)
        zero_pads(bufB, n);
        float dv[ACC];
#pragma unroll
        for (int k = 0; k < ACC; ++k) dv[k] = gyl[4 * min(lrow + k * kRowsPerPass, n - 1) + 3];
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = min(lrow + k * kRowsPerPass, n);  // rows past n: 0 into the zero row
            float4 u;
            if constexpr (GMC_OVF_HALVES_BWD && ovf) u = gather_ids8_halves<FS>(bufA, idr[k], q);   // (see lds_tile.h)
            else u = ABL(5) ? make_float4(dv[k], dv[k], dv[k], dv[k]) : gather_ids8<FS, false, NS>(bufA, idr[k], nullptr, q);
            u.x *= dv[k]; u.y *= dv[k]; u.z *= dv[k]; u.w *= dv[k];
            reinterpret_cast<float4 *>(bufB)[l * Q + q] = u;
        }
        if (hub_any) {   // hub rows: their overflow blocks, added to the U row its lanes have just written
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int lm = lrow + k * kRowsPerPass;
                float4 mine = gmc::f4_zero();
                for_hub_rows<FS, 0>(hd[k], lm, q, bufA, nullptr, ol, n, [&](int l, const float4 t) { if (lm == l) mine = t; });
                if (hd[k] != 0) {   // (the read-modify-writes of all my wave's hub rows at once, outside the serial loop)
                    float4 *cell = reinterpret_cast<float4 *>(bufB) + lm * Q + q;
                    const float4 c = *cell;
                    *cell = make_float4(fmaf(mine.x, dv[k], c.x), fmaf(mine.y, dv[k], c.y), fmaf(mine.z, dv[k], c.z), fmaf(mine.w, dv[k], c.w));
                }
            }
        }
        STAMP(3);  // gather 1
        // my share of graph g+1's row constants (the KREG register loads behind it may stay in flight)
        if (KREG > 0) vm_wait<KREG>();
        else dma_wait();
        loop_barrier();
        STAMP(4);  // barrier B: U tile complete, bufA free, next row constants visible
        // (3) next graph's H tile streams into bufA while (4) gathers from bufB and (1') turns the landed pieces into Gs
        int pieces = 0;   // tile DMA instructions this wave issues for graph g+1 (wave-uniform)
        if (kRecycle) {
            if (nn > 0) zero_pads(bufA, nn);
        } else if (nn > 0 && !ABL(1)) {
            fetch_tile(r0n, nn, KREG);
            zero_pads(bufA, nn);
#pragma unroll
            for (int k = KREG; k < ACC; ++k) pieces += wrow0 + k * kRowsPerPass < nn ? 1 : 0;
            // OVF: the next graph's block offsets travel with its tile; its blocks are requested behind the transforms
            // (the first point where the offsets are known to have landed) and fly through the fix-up loop and the
            // barrier in front of the commit: no memory round trip of the overflow lists is left in the open
            if constexpr (ovf) ovf_request(a.b, r0n, nn, (int)threadIdx.x, orq);
        }
        STAMP(5);  // fetch issue
        const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
        // order inside a wave: the first half of gather #2 goes ahead of the first transform (the DMA needs about
        // that long), then they alternate; across waves nothing is in step any more - one wave's packed-FMA
        // transform runs beside the others' LDS reads
        // Measured (round 3, same box, us per launch at R = 160,000): transform after the whole gather and a full wait
        // 112.2-113.1 | lead ACC with counted waits 113.1-114.4 | lead 3: 113.0-115.2 | lead 2: 114.4-115.5 | lead 1:
        // 117.0.  The interleaving buys nothing: a piece lands ~2.6 us after its issue whatever the wave does
        // meanwhile (every CU streams its tile at the same moment), so the transforms still start when the tile is
        // there and run with all waves in the same VALU phase.  The default is the plain order.
#ifndef GMC_BWD1_LEAD
        constexpr int kLead = ACC;
#else
        constexpr int kLead = GMC_BWD1_LEAD < ACC ? GMC_BWD1_LEAD : ACC;   // tuning builds
#endif
        // GMC_BWD1_RCK_AHEAD (tuning): the row constants of transform k+1 are read from LDS together with the reads of
        // gather row k instead of in a round trip of their own in front of the transform
#ifndef GMC_BWD1_RCK_AHEAD
#define GMC_BWD1_RCK_AHEAD 0
#endif
        constexpr bool kRckAhead = GMC_BWD1_RCK_AHEAD && kRecycle;
        float4 rck_cur = gmc::f4_zero();
        if (kRckAhead && nn > 0) rck_cur = row_consts(0, nn, gyn);
#pragma unroll
        for (int k = 0; k < ACC + kLead; ++k) {
            if (k < ACC) {
                float4 rck_next = gmc::f4_zero();
                if (kRecycle && nn > 0) {   // cell k is in registers: turn it into Gs AHEAD of row k of the gather, then
                    if (kRckAhead) {        // send for the cell that recycles its registers (longest possible shadow)
                        transform_with(k, nn, rck_cur);
                        if (k + 1 < ACC) rck_next = row_consts(k + 1, nn, gyn);
                    } else {
                        transform(k, nn, gyn);
                    }
                    if (k + KREG < ACC) load_cell(k % KREG, k + KREG, r0n, nn);
                }
                const int l = min(lrow + k * kRowsPerPass, n - 1);  // (weights of a real row; the ids are pads past n)
                if constexpr (HAS_VAL) acc[k] += gmc::f4v(gather_ids8<FS, true, NS>(bufB, idr[k], wbase + (long)l * W, q));
                else if constexpr (GMC_OVF_HALVES_BWD && ovf) acc[k] += gmc::f4v(gather_ids8_halves<FS>(bufB, idr[k], q));
                else acc[k] += ABL(4) ? (gmc::v4f)(__uint_as_float(idr[k].x)) : gather_ids8_pk<FS, NS>(bufB, idr[k], q);
                // the sum is needed HERE (its only user is the store after the graph loop: left alone the
                // optimiser sinks the adds and keeps four rows of reads, 128 VGPRs, alive)
                asm volatile("" : "+v"(acc[k]));
                if (kRckAhead) {
                    asm volatile("" : "+v"(rck_next.x), "+v"(rck_next.y), "+v"(rck_next.z), "+v"(rck_next.w));
                    rck_cur = rck_next;
                }
                // plain order (kLead == ACC): gather #2 is done with the ids - request graph g+1's now, so that they
                // travel under the wait for the tile and the transforms instead of in front of barrier A
                if (k == ACC - 1 && kLead == ACC && nn > 0) load_ids(r0n, nn);
            }
            if (!kRecycle && k >= kLead && nn > 0) {
                const int j = k - kLead;
                wait_piece(j, pieces);
                __builtin_amdgcn_sched_barrier(0);   // the transform's LDS read must not move above the wait
                transform(j, nn, gyn);
            }
        }
        if constexpr (ovf) {
            if (nn > 0 && !ABL(1)) ovf_blocks(a.b, (int)threadIdx.x, ol, orq);
        }
        if (hub_any) {   // hub rows: their overflow blocks into the dW1 accumulators
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int lm = lrow + k * kRowsPerPass;
                for_hub_rows<FS, 0>(hd[k], lm, q, bufB, nullptr, ol, n, [&](int l, const float4 t) { if (lm == l) acc[k] += gmc::f4v(t); });
            }
        }
        STAMP(6);  // gather 2 + transform of the next graph
        // (default order: the wait in front of the first transform covered the tile; the id loads of graph g+1 - and an
        // OVF kernel's block loads - stay in flight through the barrier below)
        if (!kRecycle && kLead != ACC) dma_wait();
        STAMP(7);
        if (kLead != ACC && nn > 0) load_ids(r0n, nn);   // (interleaved tuning builds: the ids are in use until here)
        if constexpr (ovf) {
            if (nn > 0) {   // every wave is done with this graph's descriptors / blocks before they are rewritten
                loop_barrier();
                ovf_commit(a.b, nn, (int)threadIdx.x, ol, orq);   // (published by barrier A; no memory access of its own)
            }
        }
        STAMP(9);
        const int n2 = g + 2 < g1 ? sload(a.b.goff, g + 3) - (r0n + nn) : 0;
        r0 = r0n; n = nn; nn = n2;
    }
    STAMP_FLUSH;
    MARK(2);
    if (col_on) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            // (default cache policy: non-temporal stores here cost the launch 2.5 us and the fold behind it 0.4 - same-box A/B)
            if (l < a.b.n_max) *reinterpret_cast<float4 *>(a.dw1part + ((long)chunk * a.b.n_max + l) * a.F + f0) = gmc::v4f_f4(acc[k]);
        }
    }
    // gather #2 of the last graph is done everywhere: bufB becomes the fold area.  An LDS-only barrier: the dW1 stores
    // above need not be acknowledged first (no difference at 20 graphs per workgroup, where this runs once per 100 us;
    // a one-graph launch of the reference schedule is mostly prologue and epilogue)
    lds_barrier();
    col_epilogue<Q>(cs, red, a.colpart, chunk, s, FS, a.F);
    MARK(3);
}

// ---- 16-slot tables (degree 9..16): neighbour table in LDS ------------------------------------------
// LDS holds the two tiles and the 32-byte-per-row table, nothing else (n = 1000 fits: 2 x 64,256 + 32,000 B).
// The row constants (GY2[r,:], dinv[r]) the transform needs are staged through the U tile's buffer, which is
// idle between the end of gather #2 of graph g-1 and the start of gather #1 of graph g: a thread loads one
// row's 16 B of graph g+1 into registers behind gather #2 and commits it with the table.
//     transform(g) [own cells; row constants from bufB] -> S -> gather #1 (bufA -> bufB) -> S ->
//     DMA(g+1) -> bufA, table / row constants(g+1) -> registers || gather #2 -> S -> commit -> S
template <int FS, int W, int ACC, bool HAS_VAL, int NS, bool OVF>
__global__ GMC_LDS_BOUNDS void bwd1_lds_kernel(Bwd1Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = FS / 4;
    constexpr int kRowsPerPass = kThreads / Q;
    constexpr int NT = (ACC * kRowsPerPass * (W / 8) + kThreads - 1) / kThreads;
    constexpr int RC = (ACC * kRowsPerPass + kThreads - 1) / kThreads;   // row-constant rows per thread
    int chunk, s;
    tile_of((int)blockIdx.x, a.chunks, a.slices, chunk, s);
    const int TF = (int)tile_floats(a.b.n_max, FS);
    float *bufA = lds, *bufB = lds + TF;
    unsigned short *nb = reinterpret_cast<unsigned short *>(lds + 2 * TF);
    float4 *gyl = reinterpret_cast<float4 *>(bufB);  // [n] x (GY2[r,:], dinv[r]): 16 n B <= the tile's 4 FS n B, below its zero rows
    float *red = bufB;  // cross-wave fold area, used after the graph loop
    const int q = threadIdx.x % Q, lrow = threadIdx.x / Q;
    const int f0 = s * FS + 4 * q;
    const bool col_on = f0 < a.F;
    const long slab = (long)s * a.b.R * FS;
    constexpr bool ovf = OVF;   // some row of the batch has more than W neighbours: overflow lists (gmc_batch.ovf_*)
    ColState cs;
    col_state_init(cs, a.W2, f0, col_on);
    gmc::v4f acc[ACC];
    uint4 pt[NT];
    float4 rc[RC];
    // overflow lists (OVF): per-row descriptors and the graph's first blocks (spare LDS) are set up with the table
    OvfLds ol{};
    OvfReq<RC> orq;
    if constexpr (ovf) ol = ovf_lds(lds, a.own_lds, a.b.n_max, a.ovf_cap);
#pragma unroll
    for (int k = 0; k < ACC; ++k) acc[k] = (gmc::v4f)(0.f);

    const int g0 = chunk * a.graphs_per_chunk, g1 = min(a.b.B, g0 + a.graphs_per_chunk);
    if (g0 >= g1) return;
    auto fetch = [&](int r0, int n) {  // H tile -> bufA (DMA); neighbour table and row constants -> registers
        dma_tile<FS, ACC, true>(a.H + slab + (long)r0 * FS + 4 * q, FS, n, true, lrow, bufA);   // H: read once
        const uint4 *src = reinterpret_cast<const uint4 *>(a.b.ell + (long)r0 * W);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int i = threadIdx.x + k * kThreads;
            pt[k] = i < n * (W / 8) ? src[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const int i = threadIdx.x + k * kThreads;
            rc[k] = reinterpret_cast<const float4 *>(a.GY2)[r0 + min(i, n - 1)];
        }
        if constexpr (ovf) ovf_request(a.b, r0, n, (int)threadIdx.x, orq);   // the graph's block offsets travel with the table
    };
    // (OVF: the blocks themselves are requested by the caller once the offsets have landed - ovf_blocks - and written here)
    auto commit = [&](int r0, int n) {  // table, row constants (and the zero rows the padding entries point at)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int i = threadIdx.x + k * kThreads;
            if (i < n * (W / 8)) reinterpret_cast<uint4 *>(nb)[i] = pt[k];
        }
#pragma unroll
        for (int k = 0; k < RC; ++k) {
            const int i = threadIdx.x + k * kThreads;
            if (i < n) gyl[i] = rc[k];
        }
        if (threadIdx.x < kPadRows * FS) {
            bufA[n * FS + threadIdx.x] = 0.f;
            bufB[n * FS + threadIdx.x] = 0.f;
        }
        if constexpr (ovf) ovf_commit(a.b, n, (int)threadIdx.x, ol, orq);   // (between two barriers: nobody reads the previous graph's any more)
    };
    // graph offsets are scalar loads: each is requested one graph ahead of its first use
    int r0 = a.b.goff[g0], n = a.b.goff[g0 + 1] - r0;
    fetch(r0, n);
    if constexpr (ovf) ovf_blocks(a.b, (int)threadIdx.x, ol, orq);
    commit(r0, n);
    dma_wait();
    __syncthreads();
    for (int g = g0; g < g1; ++g) {
        const int r0n = r0 + n;                                        // == goff[g + 1]
        const int nn = g + 1 < g1 ? sload(a.b.goff, g + 2) - r0n : n;  // next graph's size (used after gather 1; scalar cache: see sload)
        float dv[ACC];
        // OVF: my rows' overflow descriptors, read once per graph: the fix-up loops find a wave's hub rows by ballot,
        // and a wave without hub rows skips them on a scalar branch
        // (this kernel has no registers to keep them in: one bit per row lives across the graph, a fix-up loop reads
        // the descriptor of its pass's row again - one LDS read, all hub rows of the wave at once)
        unsigned hub = 0;
        bool hub_any = false;              // wave-uniform
        if constexpr (ovf) {
            unsigned d[ACC];
#pragma unroll
            for (int k = 0; k < ACC; ++k) d[k] = ovf_desc(ol, min(lrow + k * kRowsPerPass, n - 1));
#pragma unroll
            for (int k = 0; k < ACC; ++k) hub |= (d[k] != 0 && lrow + k * kRowsPerPass < n ? 1u : 0u) << k;
            hub_any = __builtin_amdgcn_ballot_w64(hub != 0) != 0;
        }
        // (1) H -> Gs in place + column partials (row constants: staged in bufB with the table)
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            const float4 rck = gyl[min(l, n - 1)];
            dv[k] = rck.w;
            transform_row(cs, reinterpret_cast<float4 *>(bufA) + min(l, n) * Q + q, rck);
        }
        __syncthreads();   // every Gs cell is written; nobody reads the row constants any more: bufB is the U tile now
        // (2) U tile = dinv o (A @ Gs)
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < n) {
                float4 u = gather_row<FS, W, false, NS>(bufA, nb, nullptr, l, q);
                u.x *= dv[k]; u.y *= dv[k]; u.z *= dv[k]; u.w *= dv[k];
                reinterpret_cast<float4 *>(bufB)[l * Q + q] = u;
            }
        }
        if (hub_any) {   // hub rows: their overflow blocks, added to the U row its lanes have just written
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int lm = lrow + k * kRowsPerPass;
                const unsigned hdk = (hub >> k & 1u) ? ovf_desc(ol, lm) : 0u;
                float4 mine = gmc::f4_zero();
                for_hub_rows<FS, 0>(hdk, lm, q, bufA, nullptr, ol, n, [&](int l, const float4 t) { if (lm == l) mine = t; });
                if (hdk != 0) {   // (the read-modify-writes of all my wave's hub rows at once, outside the serial loop)
                    float4 *cell = reinterpret_cast<float4 *>(bufB) + lm * Q + q;
                    const float4 c = *cell;
                    *cell = make_float4(fmaf(mine.x, dv[k], c.x), fmaf(mine.y, dv[k], c.y), fmaf(mine.z, dv[k], c.z), fmaf(mine.w, dv[k], c.w));
                }
            }
        }
        __syncthreads();
        // (3) next graph's H tile streams into bufA while (4) gathers from bufB
        if (g + 1 < g1) fetch(r0n, nn);
        const float *wbase = HAS_VAL ? a.b.ell_vals + (long)r0 * W : nullptr;
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < n) {
                acc[k] += gmc::f4v(gather_row<FS, W, HAS_VAL, NS>(bufB, nb, HAS_VAL ? wbase + (long)l * W : nullptr, l, q));
                asm volatile("" : "+v"(acc[k]));
            }
        }
        if (hub_any) {   // hub rows: their overflow blocks into the dW1 accumulators
#pragma unroll
            for (int k = 0; k < ACC; ++k) {
                const int lm = lrow + k * kRowsPerPass;
                const unsigned hdk = (hub >> k & 1u) ? ovf_desc(ol, lm) : 0u;
                for_hub_rows<FS, 0>(hdk, lm, q, bufB, nullptr, ol, n, [&](int l, const float4 t) { if (lm == l) acc[k] += gmc::f4v(t); });
            }
        }
        dma_wait();
        if constexpr (ovf) {
            if (g + 1 < g1) ovf_blocks(a.b, (int)threadIdx.x, ol, orq);   // (the offsets have landed with the tile; the blocks fly through the barrier)
        }
        __syncthreads();  // everyone is done with graph g's table and U tile; DMA has landed
        if (g + 1 < g1) commit(r0n, nn);
        __syncthreads();
        r0 = r0n; n = nn;
    }
    if (col_on) {
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            const int l = lrow + k * kRowsPerPass;
            if (l < a.b.n_max) *reinterpret_cast<float4 *>(a.dw1part + ((long)chunk * a.b.n_max + l) * a.F + f0) = gmc::v4f_f4(acc[k]);
        }
    }
    col_epilogue<Q>(cs, red, a.colpart, chunk, s, FS, a.F);
}

template <int FS, int W>
int launch_bwd1(const Bwd1Args &a, size_t lds, hipStream_t st) {
    constexpr int rows_per_pass = kThreads / (FS / 4);
    const int acc = (a.b.n_max + rows_per_pass - 1) / rows_per_pass;
    if (acc > 8) return GMC_ERR_UNSUPPORTED;
    const int grid = a.slices * a.chunks;
    const bool hv = a.b.ell_vals != nullptr;
    const int ns = ns_class(W, a.b.ell_slots, !hv);   // live slots: no row of the batch has more neighbours
    const bool ov = gmc_has_overflow(&a.b);   // hub rows: every slot live, overflow lists walked
    if constexpr (W == 8) {
        if (a.hZ0) {   // one graph, unit weights, no lists, n <= 1024 (gmc_bwd1_takes_head): the head rides in this launch
            return ns == 7 ? launch(bwd1_reg_kernel<FS, 4, false, 7, false, true>, grid, lds, st, a)
                           : launch(bwd1_reg_kernel<FS, 4, false, 8, false, true>, grid, lds, st, a);
        }
#define GMC_BWD1(HV, NSK, OV) (acc <= 4 ? launch(bwd1_reg_kernel<FS, 4, HV, NSK, OV>, grid, lds, st, a) \
                                        : launch(bwd1_reg_kernel<FS, 8, HV, NSK, OV>, grid, lds, st, a))
        if (ov) return hv ? GMC_ERR_UNSUPPORTED : GMC_BWD1(false, 8, true);   // (weights + overflow: row kernels, see gmc_lds_fits)
        return hv ? GMC_BWD1(true, 8, false) : ns == 7 ? GMC_BWD1(false, 7, false) : GMC_BWD1(false, 8, false);
#undef GMC_BWD1
    } else {
#define GMC_BWD1(HV, NSK, OV) (acc <= 4 ? launch(bwd1_lds_kernel<FS, W, 4, HV, NSK, OV>, grid, lds, st, a) \
                                        : launch(bwd1_lds_kernel<FS, W, 8, HV, NSK, OV>, grid, lds, st, a))
        if (ov) return hv ? GMC_ERR_UNSUPPORTED : GMC_BWD1(false, 16, true);
        return hv ? GMC_BWD1(true, 16, false) : ns == 10 ? GMC_BWD1(false, 10, false) : ns == 12 ? GMC_BWD1(false, 12, false)
                  : ns == 14 ? GMC_BWD1(false, 14, false) : GMC_BWD1(false, 16, false);
#undef GMC_BWD1
    }
}

}  // namespace

// fused layer-1 backward over the slab-layout H: dW1 partials [chunks][n_max][F] and column
// partials [chunks][F][4] (dW2, db1)
// Can the fused backward of this batch compute the head as well (gmc_bwd1_head)?  One graph (every launch of the
// reference's one-step-per-graph schedule), 8-slot table, unit weights, no overflow lists, at most 1024 rows (one row
// per thread in the head, four passes in the backward), and the head's LDS arrays fit the U tile's buffer.
bool gmc_bwd1_takes_head(const gmc_batch *b) {
    if (!b || b->B != 1 || b->ell_width != 8 || b->ell_vals || gmc_has_overflow(b) || !gmc_lds_fits(b)) return false;
    const int fs = pick_fs(b->n_max, b->ell_width);
    const int rows_per_pass = kThreads / (fs / 4);
    if (b->n_max > 4 * rows_per_pass || b->n_max > kHeadThreads) return false;
    return (size_t)(7 * (b->n_max + 4) + 64) <= tile_floats(b->n_max, fs);
}

// head != nullptr (only if gmc_bwd1_takes_head): the launch computes the graph's head too and GY2 is not read
struct gmc_bwd1_head {
    const float *Z0; int zparts; const float *b2; float C; float *P; int *S; float *loss; float *db2part; int *tick;
};
int gmc_bwd1_lds_launch(const gmc_batch *b, const float *H, const float *GY2, const float *W2,
                        float *dw1part, float *colpart, int F, int chunks, int graphs_per_chunk,
                        hipStream_t st, const gmc_bwd1_head *head) {
    if (!gmc_lds_fits(b)) return GMC_ERR_UNSUPPORTED;
    if (head && (!gmc_bwd1_takes_head(b) || chunks != 1 || !head->Z0 || !head->b2 || !head->P || !head->db2part)) return GMC_ERR_UNSUPPORTED;
    const int fs = pick_fs(b->n_max, b->ell_width);
    Bwd1Args a{*b, H, GY2, W2, dw1part, colpart, F, (F + fs - 1) / fs, chunks, graphs_per_chunk, 0, 0,
               head ? head->Z0 : nullptr, head ? head->zparts : 0, head ? head->C : 0.f, head ? head->b2 : nullptr,
               head ? head->P : nullptr, head ? head->S : nullptr, head ? head->loss : nullptr,
               head ? head->db2part : nullptr, head ? head->tick : nullptr};
    size_t lds = lds_bytes(b->n_max, b->ell_width, fs);
    if (gmc_has_overflow(b)) {   // hub rows: all of the CU's LDS, the spare holds the graphs' first overflow blocks
        const size_t own = ovf_own_bytes(1, b->n_max, b->ell_width, fs);
        if (own + ovf_desc_bytes(b->n_max) > kOvfLdsBytes) return GMC_ERR_UNSUPPORTED;   // (gmc_lds_fits says so beforehand)
        a.own_lds = (int)own;
        a.ovf_cap = ovf_cap_blocks(own, b->n_max);
        lds = kOvfLdsBytes;
    }
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
