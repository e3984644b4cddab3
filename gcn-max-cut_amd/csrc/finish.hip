// "Finish" of the fused training step: fold the partial gradients and (optionally) apply Adam,
// one launch over the flat parameter buffer [W1 | b1 | W2 | b2].
//
//   dW1[v,f]  = sum_chunk dw1part[chunk][v][f]          (v < n_max, else 0)
//   dW2[f,k]  = sum_chunk colpart[chunk][f][k],  db1[f] = sum_chunk colpart[chunk][f][3]
//   db2[k]    = sum_graph db2part[g][k]          (lane-strided partial sums + fixed tree)
// all in a fixed order (bitwise reproducible), then - when the optimizer pointers are given -
// torch.optim.Adam.step (TrainingNeural.py:386) on the same element while it is in registers.
// Replaces three tiny latency-bound launches (colsum, fold, adam) of the fused path by one.
#include "gmc_common.h"
#include <math.h>

namespace {

struct FinishArgs {
    const float *dw1part;  // [chunks][n_max][F]
    const float *colpart;  // [chunks][F][4]
    const float *db2part;  // [B][3]
    int chunks, n_max, N, F, B;
    float *grad;           // flat
    // Adam (param == nullptr: gradients only)
    float *param, *m, *v;
    double lr, beta1, beta2;
    float eps;
    const int *step_counter;  // device; already advanced by this step's head launch: the update uses it as is
    const float *loss;        // [B] per-graph losses, or nullptr: their sum goes to grad[count] (GMC_MODEL_GRAD_TAIL)
    float *w1_slab;           // optional slab copy of W1 (gmc_model.W1_slab): receives the updated W1 as well
};

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float w1, float b2, float w2,
                                          float step_size, float bc2_sqrt, float eps) {
    m = m + (g - m) * w1;
    v = fmaf(w2 * g, g, v * b2);
    p = p - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
}

__global__ __launch_bounds__(256) void finish_kernel(FinishArgs a) {
    __shared__ float sh[2];
    const bool adam = a.param != nullptr;
    if (adam && threadIdx.x == 0) {
        const double t = (double)(*a.step_counter);
        sh[0] = (float)(a.lr / (1.0 - pow(a.beta1, t)));
        sh[1] = (float)sqrt(1.0 - pow(a.beta2, t));
    }
    __syncthreads();
    const float w1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, w2 = (float)(1.0 - a.beta2);
    const float step_size = adam ? sh[0] : 0.f, bc2_sqrt = adam ? sh[1] : 1.f;
    const long nW1 = (long)a.N * a.F, live = (long)a.n_max * a.F;
    const long n4 = nW1 >> 2;  // F % 4 == 0
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        // every read of the element is requested before the first use: the launch is short and
        // latency-bound, so it pays one memory round trip, not one per batch of partials
        float4 p, m, v;
        if (adam) {
            p = reinterpret_cast<const float4 *>(a.param)[i];
            m = reinterpret_cast<const float4 *>(a.m)[i];
            v = reinterpret_cast<const float4 *>(a.v)[i];
        }
        float4 g = gmc::f4_zero();
        if (i * 4 < live) {  // chunk partials, summed in ascending chunk order
            const float4 *src = reinterpret_cast<const float4 *>(a.dw1part) + i;
            const long cs = live >> 2;
            for (int c0 = 0; c0 < a.chunks; c0 += 16) {
                float4 t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = c0 + u < a.chunks ? src[(long)(c0 + u) * cs] : gmc::f4_zero();
#pragma unroll
                for (int u = 0; u < 16; ++u) gmc::f4_add(g, t[u]);
            }
        }
        reinterpret_cast<float4 *>(a.grad)[i] = g;
        if (adam) {
            adam_elem(p.x, g.x, m.x, v.x, w1, b2, w2, step_size, bc2_sqrt, a.eps);
            adam_elem(p.y, g.y, m.y, v.y, w1, b2, w2, step_size, bc2_sqrt, a.eps);
            adam_elem(p.z, g.z, m.z, v.z, w1, b2, w2, step_size, bc2_sqrt, a.eps);
            adam_elem(p.w, g.w, m.w, v.w, w1, b2, w2, step_size, bc2_sqrt, a.eps);
            reinterpret_cast<float4 *>(a.param)[i] = p;
            reinterpret_cast<float4 *>(a.m)[i] = m;
            reinterpret_cast<float4 *>(a.v)[i] = v;
            if (a.w1_slab) {  // F % 4 == 0: the four columns stay inside one 16-column slab row
                const long e = i * 4, r = e / a.F;
                *reinterpret_cast<float4 *>(a.w1_slab + gmc::slab16_index(r, (int)(e - r * a.F), a.N)) = p;
            }
        }
    }
    // tail of the flat buffer: b1 [F], W2 [F,3] (column partials of the chunks, ascending order)
    const long tail = (long)a.F + (long)a.F * 3;
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < tail; j += stride) {
        long e = (long)j * 4 + 3;  // db1[f]
        if (j >= a.F) { const long w = j - a.F; e = (w / 3) * 4 + w % 3; }  // dW2[f,k]
        float g = 0.f;
        for (int c0 = 0; c0 < a.chunks; c0 += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = c0 + u < a.chunks ? a.colpart[(long)(c0 + u) * a.F * 4 + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) g += t[u];
        }
        const long idx = nW1 + j;
        a.grad[idx] = g;
        if (adam) adam_elem(a.param[idx], g, a.m[idx], a.v[idx], w1, b2, w2, step_size, bc2_sqrt, a.eps);
    }
    // b2 [3]: one wave folds the per-graph partials (lane-strided sums, then a fixed xor tree)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {
        float g[3] = {0.f, 0.f, 0.f};
        for (int b = threadIdx.x; b < a.B; b += 64) {
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] += a.db2part[b * 3 + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) g[k] += __shfl_xor(g[k], o, GMC_WAVE);
        if (threadIdx.x < 3) {
            const float gk = threadIdx.x == 0 ? g[0] : threadIdx.x == 1 ? g[1] : g[2];
            const long idx = nW1 + tail + threadIdx.x;
            a.grad[idx] = gk;
            if (adam) adam_elem(a.param[idx], gk, a.m[idx], a.v[idx], w1, b2, w2, step_size, bc2_sqrt, a.eps);
        }
        if (a.loss) {  // the batch's loss sum rides in the slot after the gradient (same fixed order)
            float ls = 0.f;
            for (int b = threadIdx.x; b < a.B; b += 64) ls += a.loss[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, GMC_WAVE);
            if (threadIdx.x == 0) a.grad[nW1 + tail + 3] = ls;
        }
    }
}

__global__ __launch_bounds__(64) void loss_tail_kernel(const float *loss, int B, float *slot) {
    float ls = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) ls += loss[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, GMC_WAVE);
    if (threadIdx.x == 0) *slot = ls;
}

}  // namespace

// sum of the per-graph losses into one slot (same order as the fused fold uses)
int gmc_loss_tail_launch(const float *loss, int B, float *slot, hipStream_t st) {
    hipLaunchKernelGGL(loss_tail_kernel, dim3(1), dim3(64), 0, st, loss, B, slot);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}

// param == nullptr -> gradients only.  *step_counter (device) must already hold this step's number.
int gmc_finish_launch(const float *dw1part, const float *colpart, const float *db2part, int chunks, int n_max,
                      int N, int F, int B, float *grad, float *param, float *m, float *v, double lr, double beta1,
                      double beta2, double eps, int *step_counter, const float *loss_for_tail, hipStream_t st,
                      float *w1_slab) {
    FinishArgs a{dw1part, colpart, db2part, chunks, n_max, N, F, B, grad, param, m, v, lr, beta1, beta2, (float)eps,
                 step_counter, loss_for_tail, param ? w1_slab : nullptr};
    const long n4 = (long)N * F / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    {
        GmcProbeScope probe(GMC_K_FINISH, st);
        hipLaunchKernelGGL(finish_kernel, dim3((int)blocks), dim3(256), 0, st, a);
        GMC_LAUNCH_CHECK();
    }
    return GMC_OK;
}
