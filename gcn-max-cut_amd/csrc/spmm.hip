// CSR SpMM for gfx950: Y[r,:] = act(scale[r] * sum_e vals[e] * X[col[e],:] + bias).
//
// Replaces DGL's update_all(copy_u, sum) inside GraphConv (reference call sites
// python/Training/TrainingNeural.py:80,83) and, with col = local node id and X = W1,
// the dense [n,1000]x[1000,F] feature GEMM of layer 1 (the features are the padded
// adjacency itself, :373).  HBM-bound: algorithmic bytes = 2*R*F*4 + nnz*4 + (R+1)*4 + R*4.
//
// Mapping (row-partitioned, one wave64 per row):
//   * a wave owns a row; lane l owns columns 4l..4l+3 (+256 per extra pass), so every
//     neighbour row is fetched as 1 KiB-coalesced global_load_dwordx4 bursts;
//   * rowptr/col are wave-uniform -> scalar loads (s_load), gathers use an SGPR base;
//   * up to 8 neighbour rows are in flight per lane before the first add; the adds run
//     in CSR order, so the sum is bitwise reproducible (no atomics anywhere);
//   * bias / relu / row scale are fused into the store; optionally the layer-2 feature
//     transform (Y*scale)@W2 (K=3) is fused too (wave-level reduction), so H is never
//     re-read for it;
//   * workgroups are dealt round-robin over the 8 XCDs; `group_wgs` consecutive
//     workgroup-chunks (= one graph of the block-diagonal batch) are given to ONE XCD
//     so the 7x neighbour re-reads of a graph hit that XCD's 4 MiB L2.
#include "gmc_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int kWavesPerWg = 4;
// defaults (tuned on MI355X, see profiles/): rows per wave, neighbour rows in flight, nt stores
constexpr int kRowsPerWave = 2;  // 8 rows per workgroup: 125 workgroups per n=1000 graph
constexpr int kUnroll = 8;

struct SpmmArgs {
    const int *rowptr;
    const int *col;
    const float *vals;
    const float *scale;
    const float *X;
    long ldx;
    const float *bias;
    int relu;
    float *Y;
    long ldy;
    int n_rows;
    int F;
    int group_wgs;
    const float *W2;
    float *Z0;
};

__device__ __forceinline__ int xcd_remap(int b, int nwg, int G) {
    if (G <= 0) return b;
    const int per_round = 8 * G;
    const int full = (nwg / per_round) * per_round;
    if (b >= full) return b;
    const int xcd = b & 7, j = b >> 3;
    return ((j / G) * 8 + xcd) * G + (j % G);
}

// One neighbour batch of a row: all lane->SGPR broadcasts first, then up to kUnroll
// row gathers in flight, then the adds in CSR order.
template <int NP, int kUnroll, bool HAS_VAL>
__device__ __forceinline__ void gather_rows(const SpmmArgs &a, int myc, float myv, int j, int cnt,
                                            const int (&cc)[NP], float4 (&acc)[NP]) {
    float4 x[kUnroll][NP];
    float w[kUnroll];
    const float *src[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {  // lanes >= cnt hold index 0 (a valid row)
        const long c = __builtin_amdgcn_readlane(myc, j + u);
        if (HAS_VAL) w[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                __builtin_bit_cast(int, myv), j + u));
        src[u] = a.X + c * a.ldx;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        if (j + u < cnt) {  // wave-uniform
#pragma unroll
            for (int p = 0; p < NP; ++p) x[u][p] = reinterpret_cast<const float4 *>(src[u])[cc[p]];
        }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        if (j + u < cnt) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if (HAS_VAL) gmc::f4_fma(acc[p], w[u], x[u][p]);
                else gmc::f4_add(acc[p], x[u][p]);
            }
        }
    }
}

// NP = number of 256-column passes a lane covers (F <= 256*NP); RPW = rows per wave.
// VARIANT only names the instantiation (0 = aggregation over the batch, 1 = gather from the
// shared W1) so profilers report the two uses of the kernel separately.
template <int NP, int RPW, int kUnroll, bool HAS_VAL, bool EPI, bool NT, int VARIANT>
__global__ __launch_bounds__(256) void spmm_rows_v4(SpmmArgs a) {
    const int lane = gmc::lane_id();
    const int wave = gmc::uniform((int)(threadIdx.x >> 6));
    const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x, a.group_wgs);
    const int F4 = a.F >> 2;
    const int r0 = gmc::uniform((wg * kWavesPerWg + wave) * RPW);
    if (r0 >= a.n_rows) return;
    const int nr = min(RPW, a.n_rows - r0);

    // Index prefetch for all RPW rows of this wave before any gather: rowptr (one load),
    // then the first <=64 neighbour ids (and weights) of every row, all in flight at once.
    const int rp = a.rowptr[r0 + min(lane, nr)];
    float myscale = 1.f;
    if (a.scale) myscale = a.scale[r0 + min(lane, nr - 1)];
    int myc[RPW];
    float myv[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int beg = __builtin_amdgcn_readlane(rp, i), end = __builtin_amdgcn_readlane(rp, i + 1);
        const bool have = i < nr && lane < end - beg;
        myc[i] = have ? a.col[beg + lane] : 0;
        myv[i] = 1.f;
        if (HAS_VAL) myv[i] = have ? a.vals[beg + lane] : 0.f;
    }

    // Lanes past the row end re-read the last valid float4 (same cache line, no extra
    // traffic) and are masked only at the store: keeps EXEC full through the gather.
    bool on[NP];
    int cc[NP];
    float4 bias[NP];
    float w2[EPI ? NP : 1][12];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = lane + 64 * p;
        on[p] = c < F4;
        cc[p] = on[p] ? c : F4 - 1;
        bias[p] = a.bias ? reinterpret_cast<const float4 *>(a.bias)[cc[p]] : gmc::f4_zero();
        if (EPI) {
#pragma unroll
            for (int j = 0; j < 12; ++j) w2[p][j] = on[p] ? a.W2[(long)c * 12 + j] : 0.f;
        }
    }

#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        if (i >= nr) break;
        const int r = r0 + i;
        const int beg = __builtin_amdgcn_readlane(rp, i), end = __builtin_amdgcn_readlane(rp, i + 1);
        float4 acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[p] = gmc::f4_zero();

        const int cnt0 = min(64, end - beg);
#pragma unroll 1
        for (int j = 0; j < cnt0; j += kUnroll) gather_rows<NP, kUnroll, HAS_VAL>(a, myc[i], myv[i], j, cnt0, cc, acc);
#pragma unroll 1
        for (int e0 = beg + 64; e0 < end; e0 += 64) {  // rows with more than 64 neighbours
            const int cnt = min(64, end - e0);
            const int c2 = lane < cnt ? a.col[e0 + lane] : 0;
            float v2 = 1.f;
            if (HAS_VAL) v2 = lane < cnt ? a.vals[e0 + lane] : 0.f;
#pragma unroll 1
            for (int j = 0; j < cnt; j += kUnroll) gather_rows<NP, kUnroll, HAS_VAL>(a, c2, v2, j, cnt, cc, acc);
        }

        const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myscale), i));
        float z0 = 0.f, z1 = 0.f, z2 = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float4 y;
            y.x = fmaf(acc[p].x, s, bias[p].x);
            y.y = fmaf(acc[p].y, s, bias[p].y);
            y.z = fmaf(acc[p].z, s, bias[p].z);
            y.w = fmaf(acc[p].w, s, bias[p].w);
            if (a.relu) {
                y.x = y.x > 0.f ? y.x : 0.f; y.y = y.y > 0.f ? y.y : 0.f;
                y.z = y.z > 0.f ? y.z : 0.f; y.w = y.w > 0.f ? y.w : 0.f;
            }
            if (on[p]) {
                float4 *dst = reinterpret_cast<float4 *>(a.Y + (long)r * a.ldy) + lane + 64 * p;
                if (NT) {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f t = {y.x, y.y, y.z, y.w};
                    __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(dst));
                } else {
                    *dst = y;
                }
            }
            if (EPI) {  // masked lanes carry w2 = 0
                z0 += y.x * w2[p][0] + y.y * w2[p][3] + y.z * w2[p][6] + y.w * w2[p][9];
                z1 += y.x * w2[p][1] + y.y * w2[p][4] + y.z * w2[p][7] + y.w * w2[p][10];
                z2 += y.x * w2[p][2] + y.y * w2[p][5] + y.z * w2[p][8] + y.w * w2[p][11];
            }
        }
        if (EPI) {
            z0 = gmc::wave_sum(z0); z1 = gmc::wave_sum(z1); z2 = gmc::wave_sum(z2);
            if (lane == 0) {
                float *z = a.Z0 + (long)r * 3;
                z[0] = z0 * s; z[1] = z1 * s; z[2] = z2 * s;
            }
        }
    }
}

// Any F / any alignment: lane owns single columns; used for odd hidden sizes only.
template <bool HAS_VAL>
__global__ __launch_bounds__(256) void spmm_rows_scalar(SpmmArgs a) {
    const int lane = gmc::lane_id();
    const int r = gmc::uniform((int)(blockIdx.x * kWavesPerWg + (threadIdx.x >> 6)));
    if (r >= a.n_rows) return;
    const int beg = a.rowptr[r], end = a.rowptr[r + 1];
    const float s = a.scale ? a.scale[r] : 1.0f;
    for (int c = lane; c < a.F; c += 64) {
        float acc = 0.f;
        for (int e = beg; e < end; ++e) {
            const float x = a.X[(long)a.col[e] * a.ldx + c];
            if (HAS_VAL) acc = fmaf(a.vals[e], x, acc);
            else acc += x;
        }
        float y = fmaf(acc, s, a.bias ? a.bias[c] : 0.f);
        if (a.relu) y = y > 0.f ? y : 0.f;
        a.Y[(long)r * a.ldy + c] = y;
    }
}

template <int NP, int RPW, int UNR, bool NT, int VARIANT>
int launch_cfg(SpmmArgs a, int group_rows, hipStream_t st) {
    constexpr int rows_per_wg = RPW * kWavesPerWg;
    const int grid = (a.n_rows + rows_per_wg - 1) / rows_per_wg;
    a.group_wgs = group_rows > 0 ? (group_rows + rows_per_wg - 1) / rows_per_wg : 0;
    const bool hv = a.vals != nullptr, epi = a.Z0 != nullptr;
    if (hv) {
        if (epi) hipLaunchKernelGGL((spmm_rows_v4<NP, RPW, UNR, true, true, NT, VARIANT>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((spmm_rows_v4<NP, RPW, UNR, true, false, NT, VARIANT>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (epi) hipLaunchKernelGGL((spmm_rows_v4<NP, RPW, UNR, false, true, NT, VARIANT>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((spmm_rows_v4<NP, RPW, UNR, false, false, NT, VARIANT>), dim3(grid), dim3(256), 0, st, a);
    }
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

// GMC_SPMM_TUNE="rpw,unroll,nt" selects a tuning variant (F in (256,512], aggregation use);
// read once.  Only for the sweeps recorded under profiles/: production uses the default.
struct Tune { int rpw = 0, unr = 0, nt = 0; };
const Tune &tune() {
    static Tune t = [] {
        Tune v;
#ifdef GMC_TUNING   // tuning builds only (`make variant DEFS=-DGMC_TUNING`): the shipped library reads no environment
        if (const char *e = getenv("GMC_SPMM_TUNE")) sscanf(e, "%d,%d,%d", &v.rpw, &v.unr, &v.nt);
#endif
        return v;
    }();
    return t;
}

template <int VARIANT>
int launch_np(const SpmmArgs &a, int group_rows, hipStream_t st) {
    if (a.F <= 256) return launch_cfg<1, kRowsPerWave, kUnroll, false, VARIANT>(a, group_rows, st);
    if (a.F <= 512) {
        if (VARIANT == 0 && tune().rpw) {
            const Tune &t = tune();
#define GMC_TUNE_CASE(R, U, N) \
    if (t.rpw == R && t.unr == U && t.nt == N) return launch_cfg<2, R, U, N != 0, 0>(a, group_rows, st);
            GMC_TUNE_CASE(1, 4, 0) GMC_TUNE_CASE(1, 8, 0) GMC_TUNE_CASE(2, 4, 0) GMC_TUNE_CASE(2, 8, 0)
            GMC_TUNE_CASE(4, 4, 0) GMC_TUNE_CASE(4, 8, 0) GMC_TUNE_CASE(1, 4, 1) GMC_TUNE_CASE(1, 8, 1)
            GMC_TUNE_CASE(2, 4, 1) GMC_TUNE_CASE(2, 8, 1) GMC_TUNE_CASE(4, 4, 1) GMC_TUNE_CASE(4, 8, 1)
#undef GMC_TUNE_CASE
        }
        return launch_cfg<2, kRowsPerWave, kUnroll, false, VARIANT>(a, group_rows, st);
    }
    return launch_cfg<4, kRowsPerWave, kUnroll, false, VARIANT>(a, group_rows, st);
}

}  // namespace

int gmc_spmm_launch(const int32_t *rowptr, const int32_t *col, const float *vals, const float *scale,
                    const float *X, int64_t ldx, const float *bias, int relu, float *Y, int64_t ldy,
                    int32_t n_rows, int32_t F, int32_t group_rows, const float *W2, float *Z0,
                    int tag, hipStream_t st) {
    if (!rowptr || !col || !X || !Y) return GMC_ERR_NULL;
    if (n_rows < 0 || F <= 0 || ldx < F || ldy < F) return GMC_ERR_SHAPE;
    if ((W2 == nullptr) != (Z0 == nullptr)) return GMC_ERR_NULL;
    if (n_rows == 0) return GMC_OK;
    SpmmArgs a{rowptr, col, vals, scale, X, (long)ldx, bias, relu, Y, (long)ldy,
               n_rows, F, 0, W2, Z0};
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && gmc_aligned16(X) &&
                     gmc_aligned16(Y) && (!bias || gmc_aligned16(bias)) && F <= 1024;
    GmcProbeScope probe(tag, st);
    if (!vec) {
        if (Z0) return GMC_ERR_UNSUPPORTED;  // callers use gmc_dense_hw2_f32 instead
        const int grid = (n_rows + kWavesPerWg - 1) / kWavesPerWg;
        if (vals) hipLaunchKernelGGL((spmm_rows_scalar<true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((spmm_rows_scalar<false>), dim3(grid), dim3(256), 0, st, a);
        GMC_LAUNCH_CHECK();
        return GMC_OK;
    }
    return tag == GMC_K_GATHER_W1 ? launch_np<1>(a, group_rows, st) : launch_np<0>(a, group_rows, st);
}

extern "C" int gmc_spmm_f32(const int32_t *rowptr, const int32_t *col, const float *vals,
                            const float *scale, const float *X, int64_t ldx, const float *bias,
                            int relu, float *Y, int64_t ldy, int32_t n_rows, int32_t F,
                            int32_t group_rows, const float *W2, float *Z0, gmc_stream_t stream) {
    return gmc_spmm_launch(rowptr, col, vals, scale, X, ldx, bias, relu, Y, ldy, n_rows, F,
                           group_rows, W2, Z0, GMC_K_SPMM_USER, static_cast<hipStream_t>(stream));
}
