// Decode + random-sampling post-processing of the GCN's node probabilities on the GPU.
//
// Replaces python/Testing/TestingNeuralNetwork.py:18-98 (the direct caller of the hot path in
// BASELINE configs[4], 99.7 % of the reference's inference wall time):
//   assign_partitions            :18-46  per non-terminal node one uniform draw compared with the
//                                        running sum of the row, last class as fallback; nodes 0,1,2
//                                        fixed to 0,1,2.  Arithmetic of the reference's PINNED
//                                        environment (envList.txt:105, NumPy 1.x): `cumulative_prob = 0`
//                                        `+= np.float32` promotes to float64, so the running sum of the
//                                        float32 probabilities is taken in DOUBLE and compared in double
//                                        with the (double) draw.  (Under NumPy >= 2 the same source keeps
//                                        the sum in float32 and compares in float32; the two differ only
//                                        for a draw within ~6e-8 of a boundary - tests/test_gpu_parity.py
//                                        has the directed cases.)
//   calculate_cut_value          :48-64  sum of weights of edges whose endpoints differ
//   post_processing_optimization :66-98  `iterations` samples, keep the strictly best (first wins)
// The uniforms are generated on the HOST with numpy's global RNG in the reference's draw order
// (graph -> iteration -> node), so the sampled assignments are the reference's, bit for bit.
// One workgroup per (iteration, graph): the sampled assignment lives in LDS, the cut is an
// edge-parallel count over the CSR (each undirected edge seen twice -> / 2).
#include "gmc_common.h"

namespace {

struct DecodeArgs {
    gmc_batch b;
    const float *P;          // [R,3]
    const double *uniforms;  // per graph: [iters][n_g - 3]
    const long long *uoff;   // [B+1] offsets into `uniforms`
    int iters;
    signed char *assign_all; // [iters][R]
    float *cut_all;          // [B][iters]
};

__global__ __launch_bounds__(256) void decode_sample_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sa[];
    __shared__ float red[4];
    const int it = blockIdx.x, g = blockIdx.y;
    const int r0 = a.b.goff[g];
    const int n = a.b.goff[g + 1] - r0;
    const double *u = a.uniforms + a.uoff[g] + (long long)it * (n - 3);
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        int c = l;
        if (l >= 3) {
            const float *p = a.P + (long)(r0 + l) * 3;
            const double r = u[l - 3];
            const double c0 = (double)p[0], c1 = c0 + (double)p[1];   // running sum in double (NumPy 1.x)
            c = r < c0 ? 0 : (r < c1 ? 1 : 2);                        // class 2: r < c2, or the fallback
        }
        sa[l] = (unsigned char)c;
        a.assign_all[(long)it * a.b.R + r0 + l] = (signed char)c;
    }
    __syncthreads();
    float cut = 0.f;
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const int r = r0 + l;
        const int me = sa[l];
        for (int e = a.b.rowptr[r]; e < a.b.rowptr[r + 1]; ++e) {
            const float w = a.b.vals ? a.b.vals[e] : 1.0f;
            cut += sa[a.b.lcol[e]] != me ? w : 0.f;
        }
    }
    cut = gmc::wave_sum(cut);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cut;
    __syncthreads();
    if (threadIdx.x == 0) a.cut_all[(long)g * a.iters + it] = (((red[0] + red[1]) + red[2]) + red[3]) * 0.5f;
}

struct PickArgs {
    gmc_batch b;
    int iters;
    const signed char *assign_all;
    const float *cut_all;
    int *best_assign;  // [R]
    float *best_cut;   // [B]
    int *best_iter;    // [B]
};

__global__ __launch_bounds__(256) void decode_pick_kernel(PickArgs a) {
    __shared__ int sbest;
    const int g = blockIdx.x;
    if (threadIdx.x == 0) {  // strict '>' keeps the first best (TestingNeuralNetwork.py:94)
        int bi = 0;
        float bc = a.cut_all[(long)g * a.iters];
        for (int i = 1; i < a.iters; ++i) {
            const float c = a.cut_all[(long)g * a.iters + i];
            if (c > bc) { bc = c; bi = i; }
        }
        sbest = bi;
        a.best_cut[g] = bc;
        a.best_iter[g] = bi;
    }
    __syncthreads();
    const int r0 = a.b.goff[g], n = a.b.goff[g + 1] - r0;
    for (int l = threadIdx.x; l < n; l += blockDim.x)
        a.best_assign[r0 + l] = a.assign_all[(long)sbest * a.b.R + r0 + l];
}

}  // namespace

extern "C" int gmc_decode_sample_f32(const gmc_batch *batch, const float *P, const double *uniforms,
                                     const int64_t *uoff, int32_t iters, int8_t *assign_all, float *cut_all,
                                     int32_t *best_assign, float *best_cut, int32_t *best_iter,
                                     gmc_stream_t stream) {
    if (!batch || !P || !uniforms || !uoff || !assign_all || !cut_all || !best_assign || !best_cut || !best_iter)
        return GMC_ERR_NULL;
    if (batch->abi != GMC_VERSION) return GMC_ERR_ABI;
    if (!batch->goff || !batch->rowptr || !batch->lcol) return GMC_ERR_NULL;
    if (iters < 1 || batch->B < 0) return GMC_ERR_SHAPE;
    if (batch->B > 0 && (batch->n_max < 3 || batch->n_max > 65535)) return GMC_ERR_GRAPH_SIZE;
    if (batch->B == 0) return GMC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DecodeArgs a{*batch, P, uniforms, reinterpret_cast<const long long *>(uoff), iters,
                 reinterpret_cast<signed char *>(assign_all), cut_all};
    {
        GmcProbeScope probe(GMC_K_DECODE, st);
        hipLaunchKernelGGL(decode_sample_kernel, dim3(iters, batch->B), dim3(256), (size_t)batch->n_max, st, a);
        GMC_LAUNCH_CHECK();
    }
    PickArgs p{*batch, iters, reinterpret_cast<const signed char *>(assign_all), cut_all, best_assign, best_cut, best_iter};
    hipLaunchKernelGGL(decode_pick_kernel, dim3(batch->B), dim3(256), 0, st, p);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
