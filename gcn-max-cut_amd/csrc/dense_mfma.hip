// Z0[r,:] = dinv[r] * (H[r,:] @ W2), W2 = conv2.weight [F,3]: the only dense contraction
// on the path ((H*outdeg^-1/2)@W2 inside GraphConv layer 2, TrainingNeural.py:83), on the
// matrix cores with v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain).
//
// One wave owns 16 rows.  Per 16 columns of H a lane loads ONE float4 (row l&15, columns
// 16t + 4*(l>>4) .. +3), which feeds 4 MFMA steps: step s uses k-slot q = l>>4 ->
// column 16t+4q+s for A and the matching W2 row for B (the k order inside a product is
// free as long as A and B agree).  B columns 3..15 are zero; lanes with (l&15) < 3 hold
// the 3 useful output columns for rows 4*(l>>4)+j.
//
// The fused training path does this contraction in the SpMM epilogue instead (the row is
// already in registers there, so H is never re-read); this kernel serves callers that
// hold H only (gmc_dense_hw2_f32) and odd hidden sizes.
#include "gmc_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct DenseArgs {
    const float *H;
    long ldh;
    const float *dinv;
    const float *W2;
    float *Z0;
    int R;
    int F;
};

__global__ __launch_bounds__(256) void dense_hw2_mfma_kernel(DenseArgs a) {
    const int lane = gmc::lane_id();
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row0 = wave * 16;
    if (row0 >= a.R) return;
    const int i = lane & 15, q = lane >> 4;
    const int r = min(row0 + i, a.R - 1);
    const float *hrow = a.H + (long)r * a.ldh;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int Fv = a.F & ~15;
    for (int t = 0; t < Fv; t += 16) {
        const float4 h = *reinterpret_cast<const float4 *>(hrow + t + 4 * q);
        const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float b = i < 3 ? a.W2[(long)(t + 4 * q + s) * 3 + i] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[s], b, acc, 0, 0, 0);
        }
    }
    for (int t = Fv; t < a.F; t += 4) {  // tail: plain k = t + q
        const int k = t + q;
        const float av = k < a.F ? hrow[k] : 0.f;
        const float b = (i < 3 && k < a.F) ? a.W2[(long)k * 3 + i] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc, 0, 0, 0);
    }
    if (i < 3) {  // D: col = lane&15, row = 4*(lane>>4) + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = row0 + 4 * q + j;
            if (rr < a.R) a.Z0[(long)rr * 3 + i] = acc[j] * a.dinv[rr];
        }
    }
}

}  // namespace

extern "C" int gmc_dense_hw2_f32(const float *H, int64_t ldh, const float *dinv, const float *W2,
                                 float *Z0, int32_t n_rows, int32_t F, gmc_stream_t stream) {
    if (!H || !dinv || !W2 || !Z0) return GMC_ERR_NULL;
    if (n_rows < 0 || F <= 0 || ldh < F) return GMC_ERR_SHAPE;
    if (ldh % 4 || !gmc_aligned16(H)) return GMC_ERR_ALIGN;
    if (n_rows == 0) return GMC_OK;
    DenseArgs a{H, (long)ldh, dinv, W2, Z0, n_rows, F};
    const int waves = (n_rows + 15) / 16;
    GmcProbeScope probe(GMC_K_DENSE_MFMA, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(dense_hw2_mfma_kernel, dim3((waves + 3) / 4), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
