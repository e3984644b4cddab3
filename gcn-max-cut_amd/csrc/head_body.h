// The per-graph "head" as a device function (see head.hip for what it computes): head_kernel runs it once per graph;
// the one-graph backward of the reference schedule (bwd1_lds.hip) runs it in every workgroup of its launch, so that
// a graph-step is three launches instead of four and the rows (GY2, dinv) never leave the CU.
#pragma once
#include "gmc_common.h"

// Diagnostic build only (-DGMC_STAMP, `make stamp`): wall-clock marks (s_memrealtime: one 100 MHz counter for the whole
// chip) of block 0's oldest and youngest wave at the phase boundaries; scratch/seq_stamps.py puts them on one time line
// with the marks of the other kernels of a one-graph step.  The production library contains none of this.
#ifdef GMC_STAMP
static __device__ unsigned long long g_hstamps[2 * 16];   // (one copy per translation unit; head.hip's is the one read)
#define HMARK(i)                                                                                         \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == kHeadThreads - 64))                   \
            g_hstamps[(threadIdx.x ? 16 : 0) + (i)] = __builtin_amdgcn_s_memrealtime();                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define HMARK(i)
#endif

namespace {

constexpr int kHeadThreads = 1024;

struct HeadArgs {
    gmc_batch b;
    const float *Z0;
    int zparts;
    const float *b2;
    float C;
    float *P;
    int *S;
    float *loss;
    float *GY2;
    float *db2part;
    int *tick;  // optional device step counter, advanced once per launch (fused train step)
};

// Deterministic block sum of up to 4 values per thread; result valid in thread 0.
__device__ __forceinline__ void block_sum4(float (&v)[4], float *red /* [16*4] */) {
    const int lane = gmc::lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = gmc::wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) red[wave * 4 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // every wave's partial is requested before the first add (64 reads in flight instead of 64 dependent
        // scalar reads: ~2 us of wave 0's - and so of the launch's - critical path); same ascending order of the adds
        float part[kHeadThreads / 64][4];   // (scalar reads: `red` is only 4-byte aligned for odd graph sizes)
#pragma unroll
        for (int w = 0; w < kHeadThreads / 64; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) part[w][k] = red[w * 4 + k];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int w = 0; w < kHeadThreads / 64; ++w) { s0 += part[w][0]; s1 += part[w][1]; s2 += part[w][2]; s3 += part[w][3]; }
        v[0] = s0; v[1] = s1; v[2] = s2; v[3] = s3;
    }
}

// neighbours of local row l: through the 16-byte ELL row (one load, padding ids >= n hit zeroed
// slots) when the batch carries it, else the CSR row
// W: slots of the ELL table as a compile-time constant (8 or 16; 0 = CSR walk).  With a run-time width the unpacked
// ids lived in scratch (48 B per lane): three neighbour loops per row, each going through vector memory.
template <int W, typename F>
__device__ __forceinline__ void for_neighbours(const gmc_batch &b, int r0, int l, int n, bool cached, const uint4 c0,
                                               const uint4 c1, F &&f) {
    if constexpr (W > 0) {  // cached: the row's ids are already in registers (c0, c1) - no global read
        const uint4 *row = reinterpret_cast<const uint4 *>(b.ell + (long)(r0 + l) * W);
#pragma unroll
        for (int blk = 0; blk < W / 8; ++blk) {
            const uint4 ids = cached ? (blk == 0 ? c0 : c1) : row[blk];
            const unsigned pk[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned id = (u & 1) ? pk[u >> 1] >> 16 : pk[u >> 1] & 0xffffu;
                f((int)id, b.ell_vals ? b.ell_vals[(long)(r0 + l) * W + blk * 8 + u] : 1.0f);
            }
        }
        if (b.ovf_ptr) {  // rows of hub degree: their overflow blocks (padding ids hit the zeroed slots as well)
            for (int blk = b.ovf_ptr[r0 + l]; blk < b.ovf_ptr[r0 + l + 1]; ++blk) {
                const uint4 ids = reinterpret_cast<const uint4 *>(b.ovf_ids)[blk];
                const unsigned pk[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned id = (u & 1) ? pk[u >> 1] >> 16 : pk[u >> 1] & 0xffffu;
                    f((int)id, b.ovf_vals ? b.ovf_vals[8l * blk + u] : 1.0f);
                }
            }
        }
    } else {
        for (int e = b.rowptr[r0 + l]; e < b.rowptr[r0 + l + 1]; ++e) f(b.lcol[e], b.vals ? b.vals[e] : 1.0f);
    }
    (void)n;
}

// The head of graph g on the calling workgroup (kHeadThreads threads, all of them; `lds`: 7 * (n_max + 4) + 64 floats).
//   outputs  store P / S / loss / db2part and advance the step counter (one caller per graph does)
//   gy_lds   != nullptr: the rows (GY2[r,:], dinv[r]) go to this LDS array [n] instead of a.GY2 - a caller that consumes
//            them itself (the one-graph backward: bwd1_lds.hip)
template <int W>
__device__ __forceinline__ void head_body(const HeadArgs &a, const int g, float *lds, const bool outputs, float4 *gy_lds) {
    constexpr bool ELL = W > 0;
    HMARK(0);
    const int r0 = a.b.goff[g];
    const int n = a.b.goff[g + 1] - r0;
    const int NP = a.b.n_max + 4;             // + 4 padding slots (ELL padding ids n..n+3): zeros / class 3
    float *sA = lds;                          // [NP*3]  Z0, later dinv*GZ
    float *sP = lds + 3 * NP;                 // [NP*3]  softmax output
    int *sS = reinterpret_cast<int *>(lds + 6 * NP);  // [NP] argmax class
    float *red = lds + 7 * NP;                // [64]
    const bool train = a.GY2 != nullptr || gy_lds != nullptr;
    // nobody reads the counter during this kernel.  A no-return atomic: `*tick += 1` is a load the wave has to wait for
    // before its store - a memory round trip at the top of block 0's critical path (a one-graph launch IS block 0)
    if (a.tick && outputs && g == 0 && threadIdx.x == 0) (void)__hip_atomic_fetch_add(a.tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // The kernel is one workgroup per graph and latency-bound: request everything this thread will
    // need from global memory now (its first row's neighbour ids, dinv, the bias), so that the three
    // phases below pay one memory round trip instead of one each.
    const int l0 = threadIdx.x;
    uint4 cid0 = make_uint4(0, 0, 0, 0), cid1 = cid0;
    float cd = 0.f;
    if (l0 < n) {
        cd = a.b.dinv[r0 + l0];
        if (ELL) {
            const uint4 *row = reinterpret_cast<const uint4 *>(a.b.ell + (long)(r0 + l0) * W);
            cid0 = row[0];
            if (W > 8) cid1 = row[1];
        }
    }
    const float bias0 = a.b2[0], bias1 = a.b2[1], bias2 = a.b2[2];
    // fold the slice-group partials, ascending.  A thread owns up to kFoldElems elements (i, i + T, i + 2T);
    // every load of a round - kFoldRound partials of ALL its elements - is requested before the first add:
    // one memory round trip per round instead of one per (element, 8 partials).  A single n = 1000 graph has
    // 32 partials (one workgroup per slice): 2 round trips of 16 (round 2: 4 of 8; round 1: 12); the 160-graph
    // batch (8 partials) 1.  Same summation order as before: bitwise the same Z.
    constexpr int kFoldElems = 3, kFoldRound = 16;
    for (int i0 = threadIdx.x; i0 < 3 * n; i0 += kFoldElems * (int)blockDim.x) {
        float z[kFoldElems] = {};
        int ie[kFoldElems];
#pragma unroll
        for (int e = 0; e < kFoldElems; ++e) ie[e] = min(i0 + e * (int)blockDim.x, 3 * n - 1);
        for (int p0 = 0; p0 < a.zparts; p0 += kFoldRound) {
            float t[kFoldElems][kFoldRound];
#pragma unroll
            for (int u = 0; u < kFoldRound; ++u) {
                // a partial's base is wave-uniform (scalar registers), the element index a per-lane 32-bit offset
                const float *zu = a.Z0 + ((long)min(p0 + u, a.zparts - 1) * a.b.R + r0) * 3;
#pragma unroll
                for (int e = 0; e < kFoldElems; ++e) t[e][u] = zu[ie[e]];
            }
#pragma unroll
            for (int e = 0; e < kFoldElems; ++e)
#pragma unroll
                for (int u = 0; u < kFoldRound; ++u) z[e] += p0 + u < a.zparts ? t[e][u] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < kFoldElems; ++e) {
            const int i = i0 + e * (int)blockDim.x;
            if (i < 3 * n) sA[i] = z[e];
        }
    }
    if (threadIdx.x < 12) sA[3 * n + threadIdx.x] = 0.f;
    if (threadIdx.x < 4) sS[n + threadIdx.x] = 3;  // a class no node has
    HMARK(1);
    __syncthreads();
    HMARK(2);

    // phase 1: aggregate, bias, softmax, override, argmax
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const int r = r0 + l;
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
        for_neighbours<W>(a.b, r0, l, n, l == l0, cid0, cid1, [&](int c, float) { z0 += sA[3 * c]; z1 += sA[3 * c + 1]; z2 += sA[3 * c + 2]; });
        const float d = l == l0 ? cd : a.b.dinv[r];
        z0 = fmaf(z0, d, bias0); z1 = fmaf(z1, d, bias1); z2 = fmaf(z2, d, bias2);
        const float m = fmaxf(z0, fmaxf(z1, z2));
        const float e0 = expf(z0 - m), e1 = expf(z1 - m), e2 = expf(z2 - m);
        const float inv = 1.0f / (e0 + e1 + e2);
        const float p0 = e0 * inv, p1 = e1 * inv, p2 = e2 * inv;
        if (outputs) { a.P[(long)r * 3] = p0; a.P[(long)r * 3 + 1] = p1; a.P[(long)r * 3 + 2] = p2; }
        sP[3 * l] = p0; sP[3 * l + 1] = p1; sP[3 * l + 2] = p2;
        int s;
        if (l < 3) {
            s = l;  // (e_l + p) - p: 1 at l, exactly 0 elsewhere -> argmax is l
        } else {
            s = 0;  // torch.argmax: first maximum wins
            float best = p0;
            if (p1 > best) { best = p1; s = 1; }
            if (p2 > best) { s = 2; }
        }
        sS[l] = s;
        if (a.S && outputs) a.S[r] = s;
    }
    HMARK(3);
    __syncthreads();
    HMARK(4);

    // phase 2: cut value (+ GP, softmax backward, dinv*GZ when training)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // cut2, db2[0..2]
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const int r = r0 + l;
        const int me = sS[l];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, cut = 0.f;
        for_neighbours<W>(a.b, r0, l, n, l == l0, cid0, cid1, [&](int c, float w) {
            const int sc = sS[c];  // padding slots carry class 3: no contribution
            g0 += sc == 0 ? w : 0.f; g1 += sc == 1 ? w : 0.f; g2 += sc == 2 ? w : 0.f;
            cut += (sc != me && sc != 3) ? w : 0.f;
        });
        acc[0] += cut;
        if (train) {
            g0 *= a.C; g1 *= a.C; g2 *= a.C;
            const float p0 = sP[3 * l], p1 = sP[3 * l + 1], p2 = sP[3 * l + 2];
            const float dot = g0 * p0 + g1 * p1 + g2 * p2;
            const float z0 = p0 * (g0 - dot), z1 = p1 * (g1 - dot), z2 = p2 * (g2 - dot);
            acc[1] += z0; acc[2] += z1; acc[3] += z2;
            const float d = l == l0 ? cd : a.b.dinv[r];
            sA[3 * l] = z0 * d; sA[3 * l + 1] = z1 * d; sA[3 * l + 2] = z2 * d;
        }
    }
    HMARK(5);
    block_sum4(acc, red);  // contains a __syncthreads(): sA writes are visible after it
    HMARK(6);
    if (threadIdx.x == 0 && outputs) {
        // one system-scope store: `loss` may be pinned host memory the caller watches (the value is final here,
        // long before the launch - let alone a graph of launches - ends)
        if (a.loss) __hip_atomic_store(a.loss + g, -a.C * (acc[0] * 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (train && a.db2part) {
            a.db2part[g * 3] = acc[1]; a.db2part[g * 3 + 1] = acc[2]; a.db2part[g * 3 + 2] = acc[3];
        }
    }
    if (!train) return;
    HMARK(7);

    // phase 3: GY2 = A @ (dinv o GZ)   (A symmetric: A^T == A)
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const int r = r0 + l;
        float y0 = 0.f, y1 = 0.f, y2 = 0.f;
        for_neighbours<W>(a.b, r0, l, n, l == l0, cid0, cid1, [&](int c, float) { y0 += sA[3 * c]; y1 += sA[3 * c + 1]; y2 += sA[3 * c + 2]; });
        const float4 row = make_float4(y0, y1, y2, l == l0 ? cd : a.b.dinv[r]);
        if (gy_lds) gy_lds[l] = row;
        else *reinterpret_cast<float4 *>(a.GY2 + (long)r * 4) = row;
    }
    HMARK(8);
#ifdef GMC_STAMP
    __builtin_amdgcn_s_waitcnt(0);  // every store of this wave acknowledged
    HMARK(9);
#endif
}

}  // namespace
