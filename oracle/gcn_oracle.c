/* TEST INFRASTRUCTURE ONLY - plain-C CSR oracle for the GCN max-cut hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this
 * library; the shipped HIP path never links, loads or calls it.
 *
 * Scalar fp32 restatement (one thread, CSR order, no blocking) of the arithmetic in
 *   python/Training/TrainingNeural.py:79-85   (two GraphConv layers + softmax)
 *   python/Training/TrainingNeural.py:87-106  (terminal override, argmax one-hot)
 *   python/Training/TrainingNeural.py:154-176,291-309 (cut loss, only C is used)
 *   loss.backward() of the above (:385) and torch.optim.Adam.step (:337,:386)
 * with dgl==2.0.0 GraphConv(norm='both') restated from its published algorithm
 * (DGL itself is absent from /root/reference and this image).
 *
 * Because the layer-1 features are the padded adjacency itself (:373), X@W1 is done
 * here as a sparse row combination of W1 (zeros contribute exactly +0.0f), which is
 * what lets this oracle reach the full n=1000 sizes in milliseconds.  The dense
 * reference-structured variant lives in oracle/ref_dense.py; tests check the two
 * against each other.
 *
 * PARITY STATUS: parity unpinned at the DGL boundary (see oracle/ref_dense.py
 * header for what is pinned against the reference's own importable modules).
 *
 * Graph arguments: CSR of the symmetric adjacency, local node ids, `val` = edge
 * weight (NULL = all ones).  Aggregation uses structure only (DGL passes no edge
 * weight); `val` enters through the features X and the loss.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXK 8

static float deg_inv_sqrt(const int *rowptr, int r) {
    int d = rowptr[r + 1] - rowptr[r];
    if (d < 1) d = 1; /* clamp(min=1) */
    return 1.0f / sqrtf((float)d);
}

/* Y[r,:] = act(scale[r] * sum_e val_e * X[col_e,:] + bias) ; generic SpMM used by tests */
int orc_spmm(int n_rows, const int *rowptr, const int *col, const float *val,
             const float *scale, const float *X, long ldx, const float *bias, int relu,
             float *Y, long ldy, int F) {
    for (int r = 0; r < n_rows; ++r) {
        float *y = Y + (long)r * ldy;
        for (int f = 0; f < F; ++f) y[f] = 0.0f;
        for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float *x = X + (long)col[e] * ldx;
            if (val) { float w = val[e]; for (int f = 0; f < F; ++f) y[f] += w * x[f]; }
            else     { for (int f = 0; f < F; ++f) y[f] += x[f]; }
        }
        float s = scale ? scale[r] : 1.0f;
        for (int f = 0; f < F; ++f) {
            float t = y[f] * s + (bias ? bias[f] : 0.0f);
            y[f] = (relu && !(t > 0.0f)) ? 0.0f : t;
        }
    }
    return 0;
}

/* TrainingNeural.py:79-85.  T0,H: [n,F]; Z0,P: [n,K].  Returns -1 on zero-degree node. */
int orc_forward(int n, const int *rowptr, const int *col, const float *val, int F, int K,
                const float *W1, const float *b1, const float *W2, const float *b2,
                float *T0, float *H, float *Z0, float *P) {
    if (K > ORC_MAXK) return -2;
    for (int r = 0; r < n; ++r) if (rowptr[r + 1] == rowptr[r]) return -1;
    /* conv1, in>out branch: (X * outdeg^-1/2) @ W1 first */
    for (int u = 0; u < n; ++u) {
        float du = deg_inv_sqrt(rowptr, u);
        float *t = T0 + (long)u * F;
        for (int f = 0; f < F; ++f) t[f] = 0.0f;
        for (int e = rowptr[u]; e < rowptr[u + 1]; ++e) {
            float x = (val ? val[e] : 1.0f) * du;
            const float *w = W1 + (long)col[e] * F;
            for (int f = 0; f < F; ++f) t[f] += x * w[f];
        }
    }
    /* aggregate over in-edges, * indeg^-1/2, + b1, relu */
    for (int v = 0; v < n; ++v) {
        float dv = deg_inv_sqrt(rowptr, v);
        float *h = H + (long)v * F;
        for (int f = 0; f < F; ++f) h[f] = 0.0f;
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            const float *t = T0 + (long)col[e] * F;
            for (int f = 0; f < F; ++f) h[f] += t[f];
        }
        for (int f = 0; f < F; ++f) {
            float x = h[f] * dv + b1[f];
            h[f] = x > 0.0f ? x : 0.0f;
        }
    }
    /* conv2: (H * outdeg^-1/2) @ W2, aggregate, * indeg^-1/2, + b2, softmax */
    for (int u = 0; u < n; ++u) {
        float du = deg_inv_sqrt(rowptr, u);
        const float *h = H + (long)u * F;
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int f = 0; f < F; ++f) acc += (h[f] * du) * W2[(long)f * K + k];
            Z0[(long)u * K + k] = acc;
        }
    }
    for (int v = 0; v < n; ++v) {
        float dv = deg_inv_sqrt(rowptr, v);
        float z[ORC_MAXK];
        for (int k = 0; k < K; ++k) z[k] = 0.0f;
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e)
            for (int k = 0; k < K; ++k) z[k] += Z0[(long)col[e] * K + k];
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) { z[k] = z[k] * dv + b2[k]; if (z[k] > m) m = z[k]; }
        float s = 0.0f;
        for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - m); s += z[k]; }
        for (int k = 0; k < K; ++k) P[(long)v * K + k] = z[k] / s;
    }
    return 0;
}

/* :87-106 + :154-176 + :291-309.  S[r] = argmax of the overridden row (first max wins);
 * loss = -C * cut(S);  GP = dLoss/dP = C * A_val @ onehot(S) (straight-through). */
int orc_loss_grad(int n, const int *rowptr, const int *col, const float *val, int K,
                  const float *P, float C, int *S, float *loss, float *GP) {
    for (int r = 0; r < n; ++r) {
        if (r < 3 && r < K) { S[r] = r; continue; }
        int a = 0;
        for (int k = 1; k < K; ++k) if (P[(long)r * K + k] > P[(long)r * K + a]) a = k;
        S[r] = a;
    }
    double cut2 = 0.0;
    for (int r = 0; r < n; ++r) {
        if (GP) for (int k = 0; k < K; ++k) GP[(long)r * K + k] = 0.0f;
        for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            float w = val ? val[e] : 1.0f;
            if (S[col[e]] != S[r]) cut2 += w;
            if (GP) GP[(long)r * K + S[col[e]]] += C * w;
        }
    }
    *loss = (float)(-(double)C * (cut2 / 2.0));
    return 0;
}

/* autograd of the forward given GP.  Gradients are ACCUMULATED (+=) so a batch is a
 * loop over graphs; scratch is allocated here (oracle: clarity over speed). */
int orc_backward(int n, const int *rowptr, const int *col, const float *val, int F, int K,
                 const float *W2, const float *H, const float *P, const float *GP,
                 float *dW1, float *db1, float *dW2, float *db2) {
    float *gagg = (float *)malloc(sizeof(float) * (size_t)n * K);
    float *gy2 = (float *)malloc(sizeof(float) * (size_t)n * K);
    float *gs = (float *)malloc(sizeof(float) * (size_t)n * F);
    float *gy1 = (float *)malloc(sizeof(float) * (size_t)n * F);
    if (!gagg || !gy2 || !gs || !gy1) return -3;
    for (int r = 0; r < n; ++r) { /* softmax backward, db2, scale by indeg^-1/2 */
        float dot = 0.0f, dr = deg_inv_sqrt(rowptr, r);
        for (int k = 0; k < K; ++k) dot += GP[(long)r * K + k] * P[(long)r * K + k];
        for (int k = 0; k < K; ++k) {
            float gz = P[(long)r * K + k] * (GP[(long)r * K + k] - dot);
            db2[k] += gz;
            gagg[(long)r * K + k] = gz * dr;
        }
    }
    for (int r = 0; r < n; ++r) /* A^T = A */
        for (int k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) acc += gagg[(long)col[e] * K + k];
            gy2[(long)r * K + k] = acc;
        }
    for (int r = 0; r < n; ++r) {
        float dr = deg_inv_sqrt(rowptr, r);
        const float *h = H + (long)r * F;
        for (int f = 0; f < F; ++f) {
            float g = 0.0f;
            for (int k = 0; k < K; ++k) {
                dW2[(long)f * K + k] += (h[f] * dr) * gy2[(long)r * K + k];
                g += gy2[(long)r * K + k] * W2[(long)f * K + k];
            }
            g = (h[f] > 0.0f) ? g * dr : 0.0f; /* d(H*outdeg^-1/2), relu' */
            db1[f] += g;
            gs[(long)r * F + f] = g * dr; /* * indeg^-1/2 of conv1 */
        }
    }
    for (int r = 0; r < n; ++r) {
        float *y = gy1 + (long)r * F;
        for (int f = 0; f < F; ++f) y[f] = 0.0f;
        for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float *x = gs + (long)col[e] * F;
            for (int f = 0; f < F; ++f) y[f] += x[f];
        }
    }
    /* dW1 = (X * outdeg^-1/2)^T @ gy1 ; X[u,v] = val(u,v) */
    for (int v = 0; v < n; ++v) {
        float *d = dW1 + (long)v * F;
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            int u = col[e];
            float x = (val ? val[e] : 1.0f) * deg_inv_sqrt(rowptr, u);
            const float *y = gy1 + (long)u * F;
            for (int f = 0; f < F; ++f) d[f] += x * y[f];
        }
    }
    free(gagg); free(gy2); free(gs); free(gy1);
    return 0;
}

/* torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay, no amsgrad),
 * single-tensor formulation: lerp, addcmul, sqrt/bc2_sqrt + eps, addcdiv. */
int orc_adam(float *p, const float *g, float *m, float *v, long count, double lr,
             double beta1, double beta2, double eps_d, int step) {
    double bc1 = 1.0 - pow(beta1, (double)step);
    double bc2 = 1.0 - pow(beta2, (double)step);
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2), eps = (float)eps_d;
    for (long i = 0; i < count; ++i) {
        m[i] = m[i] + (g[i] - m[i]) * w1;
        v[i] = v[i] * b2 + w2 * g[i] * g[i];
        float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
    return 0;
}

/* One optimizer step over a batch of B graphs (B == 1 is the reference's schedule,
 * :371-386).  Parameters are one flat buffer [W1 | b1 | W2 | b2]; N = in_feats. */
int orc_train_step(int B, const int *n_of, const int *const *rowptrs, const int *const *cols,
                   const float *const *vals, int N, int F, int K, float *params, float *grads,
                   float *m, float *v, double lr, float C, int step, float *losses) {
    long oW1 = 0, ob1 = (long)N * F, oW2 = ob1 + F, ob2 = oW2 + (long)F * K, P_ = ob2 + K;
    memset(grads, 0, sizeof(float) * (size_t)P_);
    for (int b = 0; b < B; ++b) {
        int n = n_of[b];
        float *T0 = (float *)malloc(sizeof(float) * (size_t)n * F);
        float *H = (float *)malloc(sizeof(float) * (size_t)n * F);
        float *Z0 = (float *)malloc(sizeof(float) * (size_t)n * K);
        float *Pm = (float *)malloc(sizeof(float) * (size_t)n * K);
        float *GP = (float *)malloc(sizeof(float) * (size_t)n * K);
        int *S = (int *)malloc(sizeof(int) * (size_t)n);
        if (!T0 || !H || !Z0 || !Pm || !GP || !S) return -3;
        int rc = orc_forward(n, rowptrs[b], cols[b], vals ? vals[b] : 0, F, K, params + oW1,
                             params + ob1, params + oW2, params + ob2, T0, H, Z0, Pm);
        if (rc) return rc;
        orc_loss_grad(n, rowptrs[b], cols[b], vals ? vals[b] : 0, K, Pm, C, S, losses + b, GP);
        rc = orc_backward(n, rowptrs[b], cols[b], vals ? vals[b] : 0, F, K, params + oW2, H, Pm,
                          GP, grads + oW1, grads + ob1, grads + oW2, grads + ob2);
        free(T0); free(H); free(Z0); free(Pm); free(GP); free(S);
        if (rc) return rc;
    }
    return orc_adam(params, grads, m, v, P_, (double)lr, 0.9, 0.999, 1e-8, step);
}
