// Fused Adam step over one flat fp32 buffer (torch.optim.Adam defaults as constructed at
// TrainingNeural.py:337 and stepped at :386): exp_avg lerp, exp_avg_sq addcmul, bias
// corrections, addcdiv - one streaming pass, 28 B/parameter (read p,g,m,v; write p,m,v).
#include "gmc_common.h"
#include <math.h>

namespace {

struct AdamArgs {
    float *p;
    const float *g;
    float *m;
    float *v;
    long n;
    float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamArgs &a) {
    m = m + (g - m) * a.one_minus_b1;                 // exp_avg.lerp_(grad, 1-beta1)
    v = fmaf(a.one_minus_b2 * g, g, v * a.b2);        // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);                // param.addcdiv_(m, denom, -step_size)
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const long n4 = a.n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4 *>(a.p)[i];
        const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
        float4 m = reinterpret_cast<float4 *>(a.m)[i];
        float4 v = reinterpret_cast<float4 *>(a.v)[i];
        adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a);
        adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4 *>(a.p)[i] = p;
        reinterpret_cast<float4 *>(a.m)[i] = m;
        reinterpret_cast<float4 *>(a.v)[i] = v;
    }
    if (blockIdx.x == 0) {  // tail (count % 4)
        const long i = (n4 << 2) + threadIdx.x;
        if (i < a.n) adam1(a.p[i], a.g[i], a.m[i], a.v[i], a);
    }
}

// Graph-replayable form: the step number lives in device memory (a captured launch cannot take
// a new scalar per replay).  Every block derives the bias corrections from *step_counter + 1 in
// double, exactly as the host does for gmc_adam_f32; adam_tick_kernel then advances the counter.
struct AdamDevArgs {
    float *p;
    const float *g;
    float *m;
    float *v;
    long n;
    double lr, beta1, beta2;
    float eps;
    const int *step_counter;
    float *w1_slab;  // optional: slab copy of the first N*F parameters (conv1.weight), refreshed with the update
    int N, F;
    int step_bias;   // 1: *step_counter = steps done so far (this launch applies step + 1); 0: it already holds this step's number
};

__global__ __launch_bounds__(256) void adam_devstep_kernel(AdamDevArgs d) {
    __shared__ float sh[2];
    if (threadIdx.x == 0) {
        const double t = (double)(*d.step_counter + d.step_bias);
        const double bc1 = 1.0 - pow(d.beta1, t), bc2 = 1.0 - pow(d.beta2, t);
        sh[0] = (float)(d.lr / bc1);
        sh[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    const AdamArgs a{d.p, d.g, d.m, d.v, d.n, (float)(1.0 - d.beta1), (float)d.beta2, (float)(1.0 - d.beta2),
                     sh[0], sh[1], d.eps};
    const long n4 = a.n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4 *>(a.p)[i];
        const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
        float4 m = reinterpret_cast<float4 *>(a.m)[i];
        float4 v = reinterpret_cast<float4 *>(a.v)[i];
        adam1(p.x, g.x, m.x, v.x, a); adam1(p.y, g.y, m.y, v.y, a);
        adam1(p.z, g.z, m.z, v.z, a); adam1(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4 *>(a.p)[i] = p;
        reinterpret_cast<float4 *>(a.m)[i] = m;
        reinterpret_cast<float4 *>(a.v)[i] = v;
        if (d.w1_slab && i * 4 < (long)d.N * d.F) {  // (F % 4 == 0: a float4 never leaves its slab row)
            const long e = i * 4, r = e / d.F;
            *reinterpret_cast<float4 *>(d.w1_slab + gmc::slab16_index(r, (int)(e - r * d.F), d.N)) = p;
        }
    }
    if (blockIdx.x == 0) {
        const long i = (n4 << 2) + threadIdx.x;
        if (i < a.n) adam1(a.p[i], a.g[i], a.m[i], a.v[i], a);
    }
}

__global__ void adam_tick_kernel(int *step_counter) { *step_counter += 1; }

// one wave in front of a data-parallel rank's Adam launch: hands the step's (all-reduced) loss to the host - one
// system-scope store per value into pinned memory - and advances the step counter, so that the Adam launch behind it
// uses the counter as it stands and no one-thread launch has to trail it
__global__ __launch_bounds__(64) void publish_tick_kernel(const float *src, int n, float *dst, int *step_counter) {
    for (int i = threadIdx.x; i < n; i += 64) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) *step_counter += 1;
}

}  // namespace

namespace {
int adam_devstep(float *param, const float *grad, float *m, float *v, int64_t count, double lr, double beta1,
                 double beta2, double eps, int32_t *step_counter, gmc_stream_t stream, float *w1_slab, int N, int F,
                 const float *publish_src = nullptr, int publish_n = 0, float *publish_dst = nullptr) {
    if (!param || !grad || !m || !v || !step_counter) return GMC_ERR_NULL;
    if (count < 0) return GMC_ERR_SHAPE;
    if (!gmc_aligned16(param) || !gmc_aligned16(grad) || !gmc_aligned16(m) || !gmc_aligned16(v))
        return GMC_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ticked = publish_dst != nullptr;   // publish + tick first, then Adam with the counter as it stands
    if (ticked) {
        hipLaunchKernelGGL(publish_tick_kernel, dim3(1), dim3(64), 0, st, publish_src, publish_n, publish_dst, step_counter);
        GMC_LAUNCH_CHECK();
    }
    if (count > 0) {
        AdamDevArgs d{param, grad, m, v, (long)count, lr, beta1, beta2, (float)eps, step_counter, w1_slab, N, F, ticked ? 0 : 1};
        long blocks = ((count >> 2) + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > 2048) blocks = 2048;
        GmcProbeScope probe(GMC_K_ADAM, st);
        hipLaunchKernelGGL(adam_devstep_kernel, dim3((int)blocks), dim3(256), 0, st, d);
        GMC_LAUNCH_CHECK();
    }
    if (!ticked) {
        hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_counter);
        GMC_LAUNCH_CHECK();
    }
    return GMC_OK;
}
}  // namespace

extern "C" int gmc_adam_devstep_f32(float *param, const float *grad, float *m, float *v, int64_t count,
                                    double lr, double beta1, double beta2, double eps, int32_t *step_counter,
                                    gmc_stream_t stream) {
    return adam_devstep(param, grad, m, v, count, lr, beta1, beta2, eps, step_counter, stream, nullptr, 0, 0);
}

// gmc_adam_devstep_f32 over the flat [W1 | b1 | W2 | b2] buffer of an N x F x 3 model that also refreshes the
// slab copy of W1 (gmc_model.W1_slab; NULL = none) with the updated values
extern "C" int gmc_adam_devstep_model_f32(float *param, const float *grad, float *m, float *v, int32_t N, int32_t F,
                                          float *w1_slab, double lr, double beta1, double beta2, double eps,
                                          int32_t *step_counter, gmc_stream_t stream) {
    if (N <= 0 || F <= 0 || F % 4) return GMC_ERR_SHAPE;
    if (w1_slab && !gmc_aligned16(w1_slab)) return GMC_ERR_ALIGN;
    const int64_t count = (int64_t)N * F + F + (int64_t)F * 3 + 3;
    return adam_devstep(param, grad, m, v, count, lr, beta1, beta2, eps, step_counter, stream, w1_slab, N, F);
}

// the data-parallel rank's tail of a step in two launches: (1) one wave stores `publish_n` floats from `publish_src`
// (the all-reduced loss: the slot after the gradient) into pinned host memory and advances the step counter;
// (2) gmc_adam_devstep_model_f32's sweep with the counter as it then stands
extern "C" int gmc_publish_adam_devstep_model_f32(const float *publish_src, int32_t publish_n, float *pinned_dst, float *param,
                                                  const float *grad, float *m, float *v, int32_t N, int32_t F, float *w1_slab,
                                                  double lr, double beta1, double beta2, double eps, int32_t *step_counter,
                                                  gmc_stream_t stream) {
    if (!publish_src || !pinned_dst) return GMC_ERR_NULL;
    if (N <= 0 || F <= 0 || F % 4 || publish_n < 1) return GMC_ERR_SHAPE;
    if (w1_slab && !gmc_aligned16(w1_slab)) return GMC_ERR_ALIGN;
    const int64_t count = (int64_t)N * F + F + (int64_t)F * 3 + 3;
    return adam_devstep(param, grad, m, v, count, lr, beta1, beta2, eps, step_counter, stream, w1_slab, N, F, publish_src,
                        publish_n, pinned_dst);
}

extern "C" int gmc_adam_f32(float *param, const float *grad, float *m, float *v, int64_t count,
                            double lr, double beta1, double beta2, double eps, int32_t step,
                            gmc_stream_t stream) {
    if (!param || !grad || !m || !v) return GMC_ERR_NULL;
    if (count < 0 || step < 1) return GMC_ERR_SHAPE;
    if (!gmc_aligned16(param) || !gmc_aligned16(grad) || !gmc_aligned16(m) || !gmc_aligned16(v))
        return GMC_ERR_ALIGN;
    if (count == 0) return GMC_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{param, grad, m, v, (long)count, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
               (float)(lr / bc1), (float)sqrt(bc2), (float)eps};
    long blocks = ((count >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    GmcProbeScope probe(GMC_K_ADAM, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    GMC_LAUNCH_CHECK();
    return GMC_OK;
}
