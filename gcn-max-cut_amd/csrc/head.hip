// Per-graph "head" of the GCN max-cut step: everything that is [n,3]-shaped.
//
// One workgroup owns one graph of the block-diagonal batch and keeps its [n,3] tiles in
// LDS, so the K=3 aggregations (forward Z = A Z0, backward GY2 = A (dinv*GZ)) never touch
// HBM between phases.  Reference lines replaced:
//   TrainingNeural.py:83-84   conv2 aggregate + bias, softmax
//   TrainingNeural.py:87-94   override_fixed_nodes (rows 0,1,2 <- e0,e1,e2)
//   TrainingNeural.py:96-106  per-row argmax one-hot (Python loop over n rows there)
//   TrainingNeural.py:154-176,:291-309  loss = -C * cut(S)   (dense [n,1000] there)
//   loss.backward() (:385) down to the input of conv2's aggregation:
//      GP = C * A_val @ onehot(S);  GZ = P o (GP - rowsum(GP o P));  db2 = colsum(GZ);
//      GY2 = A @ (dinv o GZ)
// Sums run in CSR / fixed tree order: bitwise reproducible.
#include "head_body.h"

#ifdef GMC_STAMP
extern "C" int gmc_debug_read_stamps_head(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hstamps), sizeof(unsigned long long) * n);
}
#endif

namespace {

template <int W>
__global__ __launch_bounds__(kHeadThreads) void head_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    head_body<W>(a, (int)blockIdx.x, lds, true, nullptr);
}

// GY2 for a caller-supplied dLoss/dP (autograd path): same phases 2b/3 as above.
struct HeadBwdArgs {
    gmc_batch b;
    const float *P;
    const float *GP;
    float *GY2;
    float *db2part;
};

__global__ __launch_bounds__(kHeadThreads) void head_bwd_kernel(HeadBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int g = blockIdx.x;
    const int r0 = a.b.goff[g];
    const int n = a.b.goff[g + 1] - r0;
    float *sA = lds;
    float *red = lds + 3 * a.b.n_max;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const long r = r0 + l;
        const float g0 = a.GP[r * 3], g1 = a.GP[r * 3 + 1], g2 = a.GP[r * 3 + 2];
        const float p0 = a.P[r * 3], p1 = a.P[r * 3 + 1], p2 = a.P[r * 3 + 2];
        const float dot = g0 * p0 + g1 * p1 + g2 * p2;
        const float z0 = p0 * (g0 - dot), z1 = p1 * (g1 - dot), z2 = p2 * (g2 - dot);
        acc[1] += z0; acc[2] += z1; acc[3] += z2;
        const float d = a.b.dinv[r];
        sA[3 * l] = z0 * d; sA[3 * l + 1] = z1 * d; sA[3 * l + 2] = z2 * d;
    }
    block_sum4(acc, red);
    if (threadIdx.x == 0) {
        a.db2part[g * 3] = acc[1]; a.db2part[g * 3 + 1] = acc[2]; a.db2part[g * 3 + 2] = acc[3];
    }
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        const int r = r0 + l;
        const int beg = a.b.rowptr[r], end = a.b.rowptr[r + 1];
        float y0 = 0.f, y1 = 0.f, y2 = 0.f;
        for (int e = beg; e < end; ++e) {
            const int c = a.b.lcol[e];
            y0 += sA[3 * c]; y1 += sA[3 * c + 1]; y2 += sA[3 * c + 2];
        }
        *reinterpret_cast<float4 *>(a.GY2 + (long)r * 4) = make_float4(y0, y1, y2, a.b.dinv[r]);
    }
}

int check_batch(const gmc_batch *b) {
    if (!b) return GMC_ERR_NULL;
    if (b->abi != GMC_VERSION) return GMC_ERR_ABI;
    if (!b->goff || !b->rowptr || !b->gcol || !b->lcol || !b->dinv) return GMC_ERR_NULL;
    if (b->B < 0 || b->R < 0 || b->nnz < 0) return GMC_ERR_SHAPE;
    if (b->B > 0 && (b->n_max < 3 || b->n_max > GMC_MAX_GRAPH_NODES)) return GMC_ERR_GRAPH_SIZE;
    return GMC_OK;
}

}  // namespace

// Internal launcher.  tick != nullptr: the launch also advances the device-side Adam step counter
// (gmc_train_step_f32; saves a one-thread launch per step).
int gmc_head_launch(const gmc_batch *batch, const float *Z0, int32_t z_parts, const float *b2, float C, float *P,
                    int32_t *S, float *loss, float *GY2, float *db2part, int *tick, hipStream_t stream);

extern "C" int gmc_head_f32(const gmc_batch *batch, const float *Z0, int32_t z_parts, const float *b2,
                            float C, float *P, int32_t *S, float *loss, float *GY2, float *db2part,
                            gmc_stream_t stream) {
    return gmc_head_launch(batch, Z0, z_parts, b2, C, P, S, loss, GY2, db2part, nullptr,
                           static_cast<hipStream_t>(stream));
}

int gmc_head_launch(const gmc_batch *batch, const float *Z0, int32_t z_parts, const float *b2, float C, float *P,
                    int32_t *S, float *loss, float *GY2, float *db2part, int *tick, hipStream_t stream) {
    int rc = check_batch(batch);
    if (rc) return rc;
    if (!Z0 || !b2 || !P) return GMC_ERR_NULL;
    if (z_parts < 1) return GMC_ERR_SHAPE;
    if (GY2 && !db2part) return GMC_ERR_NULL;
    if (batch->B == 0) return GMC_OK;
    HeadArgs a{*batch, Z0, z_parts, b2, C, P, S, loss, GY2, db2part, tick};
    const size_t lds = sizeof(float) * (7 * ((size_t)batch->n_max + 4) + 64);
    const int w = (batch->ell != nullptr && (batch->ell_width == 8 || batch->ell_width == 16)) ? batch->ell_width : 0;
    const void *fn = w == 8 ? reinterpret_cast<const void *>(head_kernel<8>)
                   : w == 16 ? reinterpret_cast<const void *>(head_kernel<16>) : reinterpret_cast<const void *>(head_kernel<0>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    GmcProbeScope probe(GMC_K_HEAD, stream);
    if (w == 8) hipLaunchKernelGGL(head_kernel<8>, dim3(batch->B), dim3(kHeadThreads), lds, stream, a);
    else if (w == 16) hipLaunchKernelGGL(head_kernel<16>, dim3(batch->B), dim3(kHeadThreads), lds, stream, a);
    else hipLaunchKernelGGL(head_kernel<0>, dim3(batch->B), dim3(kHeadThreads), lds, stream, a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

// internal (api.hip)
int gmc_head_bwd_launch(const gmc_batch *batch, const float *P, const float *GP, float *GY2,
                        float *db2part, hipStream_t st) {
    int rc = check_batch(batch);
    if (rc) return rc;
    if (batch->B == 0) return GMC_OK;
    HeadBwdArgs a{*batch, P, GP, GY2, db2part};
    const size_t lds = sizeof(float) * (3 * (size_t)batch->n_max + 64);
    GmcProbeScope probe(GMC_K_HEAD, st);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(batch->B), dim3(kHeadThreads), lds, st, a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
