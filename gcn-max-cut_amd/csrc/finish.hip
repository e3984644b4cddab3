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

// Diagnostic build only (-DGMC_STAMP): wall-clock entry / exit marks of every block (see head.hip)
#ifdef GMC_STAMP
static __device__ unsigned long long g_fstamps[2048 * 2];
extern "C" int gmc_debug_read_stamps_fin(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fstamps), sizeof(unsigned long long) * n);
}
#define FMARK(i) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) g_fstamps[blockIdx.x * 2 + (i)] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FMARK(i)
#endif

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

// R: chunk partials requested per round trip (a power of two >= the number of chunks, at most 16): every load of a
// round is unconditional (indices past the last chunk re-read the last one and are dropped by a select), so the
// loop body has no branches for the compiler to sink the adds into - which it did, with a wait behind the first load.
template <int R>
__global__ __launch_bounds__(256) void finish_kernel(FinishArgs a) {
    FMARK(0);
    const bool adam = a.param != nullptr;
    // Bias corrections of this step (torch.optim.Adam: step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t), in double).
    // The launch is 7-11 us long and latency-bound, and two double-precision pow() calls are ~1 us of it: the step
    // number comes by a SCALAR load requested first of all (its own counter: no vmcnt wait), every lane computes the
    // corrections for itself (no LDS hand-over, no barrier), and the element loop requests its first element's
    // p / m / v / partials BEFORE that arithmetic, which then runs in the shadow of the memory round trip.
    int tstep = 0;
    if (adam) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(tstep) : "s"(a.step_counter) : "memory");
    const float w1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, w2 = (float)(1.0 - a.beta2);
    const long nW1 = (long)a.N * a.F, live = (long)a.n_max * a.F;
    const long n4 = nW1 >> 2;  // F % 4 == 0
    const long stride = (long)gridDim.x * blockDim.x;
    const long cs = live >> 2;
    const long i_first = blockIdx.x * (long)blockDim.x + threadIdx.x;
    float4 p, m, v, t[R];
    // reads of one element: p / m / v and the first R chunk partials (rows past n_max: no graph reaches them, gradient 0)
    auto request = [&](long i) {
        if (adam) {
            p = reinterpret_cast<const float4 *>(a.param)[i];
            m = reinterpret_cast<const float4 *>(a.m)[i];
            v = reinterpret_cast<const float4 *>(a.v)[i];
        }
        const float4 *src = reinterpret_cast<const float4 *>(a.dw1part) + (i * 4 < live ? i : 0);
#pragma unroll
        for (int u = 0; u < R; ++u) t[u] = src[(long)min(u, a.chunks - 1) * cs];
    };
    request(min(i_first, n4 - 1));   // (threads beyond the last element re-read it and drop it)
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (adam) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tstep) : : "memory");   // (nothing reads tstep before this)
        const double ts = (double)tstep;
        step_size = (float)(a.lr / (1.0 - pow(a.beta1, ts)));
        bc2_sqrt = (float)sqrt(1.0 - pow(a.beta2, ts));
    }
    for (long i = i_first; i < n4; i += stride) {
        if (i != i_first) request(i);
        const bool has_parts = i * 4 < live;
        float4 g = gmc::f4_zero();
        // chunk partials, summed in ascending chunk order
#pragma unroll
        for (int u = 0; u < R; ++u) gmc::f4_add(g, has_parts && u < a.chunks ? t[u] : gmc::f4_zero());
        if (has_parts) {
            const float4 *src = reinterpret_cast<const float4 *>(a.dw1part) + i;
            for (int c0 = R; c0 < a.chunks; c0 += R) {   // (more than 16 chunks)
#pragma unroll
                for (int u = 0; u < R; ++u) t[u] = src[(long)min(c0 + u, a.chunks - 1) * cs];
#pragma unroll
                for (int u = 0; u < R; ++u) gmc::f4_add(g, c0 + u < a.chunks ? t[u] : gmc::f4_zero());
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
        for (int k = 0; k < 3; ++k) g[k] = gmc::wave_sum(g[k]);
        if (threadIdx.x < 3) {
            const float gk = threadIdx.x == 0 ? g[0] : threadIdx.x == 1 ? g[1] : g[2];
            const long idx = nW1 + tail + threadIdx.x;
            a.grad[idx] = gk;
            if (adam) adam_elem(a.param[idx], gk, a.m[idx], a.v[idx], w1, b2, w2, step_size, bc2_sqrt, a.eps);
        }
        if (a.loss) {  // the batch's loss sum rides in the slot after the gradient (same fixed order)
            float ls = 0.f;
            for (int b = threadIdx.x; b < a.B; b += 64) ls += a.loss[b];
            ls = gmc::wave_sum(ls);
            if (threadIdx.x == 0) a.grad[nW1 + tail + 3] = ls;
        }
    }
#ifdef GMC_STAMP
    __builtin_amdgcn_s_waitcnt(0);
    FMARK(1);
#endif
}

__global__ __launch_bounds__(64) void loss_tail_kernel(const float *loss, int B, float *slot) {
    float ls = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) ls += loss[b];
    ls = gmc::wave_sum(ls);
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
        if (chunks <= 1) hipLaunchKernelGGL(finish_kernel<1>, dim3((int)blocks), dim3(256), 0, st, a);
        else if (chunks <= 2) hipLaunchKernelGGL(finish_kernel<2>, dim3((int)blocks), dim3(256), 0, st, a);
        else if (chunks <= 4) hipLaunchKernelGGL(finish_kernel<4>, dim3((int)blocks), dim3(256), 0, st, a);
        else if (chunks <= 8) hipLaunchKernelGGL(finish_kernel<8>, dim3((int)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(finish_kernel<16>, dim3((int)blocks), dim3(256), 0, st, a);
        GMC_LAUNCH_CHECK();
    }
    return GMC_OK;
}
