// C-ABI orchestration: the fused forward / training entry points of gcnmaxcut.h.
// Everything is enqueued on the caller's stream; nothing allocates or synchronises.
#include "gmc_common.h"
#include <stdlib.h>

// launchers defined in the kernel files
int gmc_head_bwd_launch(const gmc_batch *, const float *, const float *, float *, float *, hipStream_t);
int gmc_hidden_tiles(int R);
int gmc_hidden_slab_tiles(int R);
int gmc_hidden_bwd_slab_launch(const float *, const float *, const float *, const float *, float *, float *, int,
                               int, int, hipStream_t);
int gmc_lds_slice_width(const gmc_batch *b);
int gmc_dw1_chunks(int B, bool lds, int slices);
int gmc_lds_slices(const gmc_batch *b, int F);
int gmc_fold_chunks_launch(const float *, float *, int, int, int, int, hipStream_t);
bool gmc_bwd1_fits(const gmc_batch *b);
int gmc_head_launch(const gmc_batch *, const float *, int32_t, const float *, float, float *, int32_t *, float *,
                    float *, float *, int *, hipStream_t);
int gmc_finish_launch(const float *, const float *, const float *, int, int, int, int, int, float *, float *, float *,
                      float *, double, double, double, double, int *, const float *, hipStream_t, float *);
int gmc_loss_tail_launch(const float *, int, float *, hipStream_t);
int gmc_fwd1_lds_launch(const gmc_batch *, const float *, const float *, const float *, float *, float *, int,
                        hipStream_t, const float *, int);
struct gmc_bwd1_head {   // (bwd1_lds.hip) the head's arguments for a backward launch that computes it as well
    const float *Z0; int zparts; const float *b2; float C; float *P; int *S; float *loss; float *db2part; int *tick;
};
bool gmc_bwd1_takes_head(const gmc_batch *b);
int gmc_bwd1_lds_launch(const gmc_batch *, const float *, const float *, const float *, float *, float *, int, int,
                        int, hipStream_t, const gmc_bwd1_head *);
int gmc_hidden_bwd_launch(const float *, long, const float *, const float *, const float *, float *,
                          long, float *, int, int, hipStream_t);
int gmc_colsum_reduce_launch(const float *, int, int, float *, float *, const float *, int, float *,
                             hipStream_t);
size_t gmc_dw1_scratch_floats(const gmc_batch *b, int N, int F, bool lds);
int gmc_dw1_launch(const gmc_batch *, const float *, long, float *, float *, int, int, bool, hipStream_t);
bool gmc_lds_fits(const gmc_batch *b);
int gmc_lds_groups(const gmc_batch *b, int F);
int gmc_spmm_lds_launch(const gmc_batch *, const float *, long, int, int, int, const float *, const float *, int,
                        float *, long, int, int, const float *, float *, int, hipStream_t);
int gmc_dropout_launch(float *, long, int, int, long, float, unsigned long long, hipStream_t);
int gmc_hw2_rows_launch(const float *, const float *, const float *, float *, long, int, int, long, hipStream_t);
int gmc_scale_copy_launch(const float *, float *, int, float, hipStream_t);

namespace {

struct Workspace {
    long ld;         // row kernels: leading dimension of the [R,F] buffers (F rounded up to 32 floats)
    int fs;          // LDS kernels: slice width of the slab layout [slice][R][fs] (0 = row-major)
    int zparts;      // partials in Z0 (LDS path: one per slice group)
    float *T0;       // [R,ld]  (X o dinv)@W1, later Gs = dinv o Gpre
    float *H;        // [R,ld]  relu(conv1), later U = dinv o (A @ Gs)
    float *Z0;       // [zparts,R,3]
    float *GY2;      // [R,4] = (GY2[r,0..2], dinv[r])
    float *part;     // [tiles,F,4]
    float *db2part;  // [B,3]
    float *dw1part;  // [chunks,N,F]
    float *W2s;      // [F,3] = W2 / (1-p), dropout only (last, so that forward-only carving shares the prefix)
    size_t bytes;
};

bool dropout_on(const gmc_model *m) { return m->dropout_p > 0.f; }
bool fuse_enabled();
bool use_lds(const gmc_batch *b, const gmc_model *m);
unsigned long long dropout_seed(const gmc_model *m) { return ((unsigned long long)m->dropout_seed_hi << 32) | m->dropout_seed_lo; }

// Fused layer kernels (default) vs the one-kernel-per-op sequence: gmc_set_fuse(0) selects the latter
// (bench.py times the stand-alone SpMM kernel that way).  The shipped library reads no environment variables;
// tuning builds (`make variant DEFS=-DGMC_TUNING`) also honour GMC_FUSE=0 and GMC_SPMM_ALGO=rows.
int g_fuse = -1;
bool fuse_enabled() {
    if (g_fuse < 0) {
        g_fuse = 1;
#ifdef GMC_TUNING
        const char *e = getenv("GMC_FUSE");
        if (e && e[0] == '0') g_fuse = 0;
#endif
    }
    return g_fuse != 0;
}

// LDS-staged tiles whenever the largest graph fits a CU's LDS.  A batch with overflow lists (rows of more than
// ell_width neighbours) is served by the FUSED LDS kernels only: with the one-kernel-per-operation sequence
// selected, or with dropout (which runs that sequence), it takes the row kernels.
bool use_lds(const gmc_batch *b, const gmc_model *m) {
#ifdef GMC_TUNING
    static const int forced = [] {
        const char *e = getenv("GMC_SPMM_ALGO");
        return !e ? 0 : (e[0] == 'r' ? 1 : 2);
    }();
    if (forced == 1) return false;
#endif
    if (gmc_has_overflow(b) && (!fuse_enabled() || (m && m->dropout_p > 0.f))) return false;
    return gmc_lds_fits(b);
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

Workspace carve(const gmc_batch *b, const gmc_model *m, int training, void *base) {
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t floats) {
        float *p = base ? reinterpret_cast<float *>(static_cast<char *>(base) + off) : nullptr;
        off += align_up(floats * sizeof(float));
        return p;
    };
    const size_t R = (size_t)b->R, F = (size_t)m->F;
    w.ld = (long)((F + 31) / 32 * 32);
    w.fs = use_lds(b, m) ? gmc_lds_slice_width(b) : 0;
    w.zparts = (w.fs && !dropout_on(m)) ? gmc_lds_groups(b, m->F) : 1;   // (dropout: Z0 comes from its own kernel)
    const size_t cols = w.fs ? (F + w.fs - 1) / w.fs * w.fs : (size_t)w.ld;
    w.T0 = take(R * cols);
    w.H = take(R * cols);
    w.Z0 = take((size_t)w.zparts * R * 3);
    if (training) {
        w.GY2 = take(R * 4);  // (GY2[r,0..2], dinv[r]) per row: one aligned 16 B load downstream
        size_t tiles = w.fs ? gmc_hidden_slab_tiles(b->R) : gmc_hidden_tiles(b->R);
        if (w.fs && (size_t)gmc_dw1_chunks(b->B, true, gmc_lds_slices(b, F)) > tiles) tiles = gmc_dw1_chunks(b->B, true, gmc_lds_slices(b, F));
        w.part = take(tiles * F * 4);
        w.db2part = take((size_t)b->B * 3);
        w.dw1part = take(gmc_dw1_scratch_floats(b, m->N, m->F, w.fs != 0));
        if (dropout_on(m)) w.W2s = take(F * 3);
    }
    w.bytes = off;
    return w;
}

int check(const gmc_batch *b, const gmc_model *m) {
    if (!b || !m) return GMC_ERR_NULL;
    if (b->abi != GMC_VERSION || m->abi != GMC_VERSION) return GMC_ERR_ABI;   // built against another header
    if (!b->goff || !b->rowptr || !b->gcol || !b->lcol || !b->dinv) return GMC_ERR_NULL;
    if (!m->W1 || !m->b1 || !m->W2 || !m->b2) return GMC_ERR_NULL;
    if (m->K != 3) return GMC_ERR_CLASSES;
    if (b->B < 0 || b->R < 0 || b->nnz < 0 || m->N <= 0 || m->F <= 0) return GMC_ERR_SHAPE;
    if (m->F % 4 || m->F > 1024) return GMC_ERR_UNSUPPORTED;  // float4 rows, <= 4 passes per lane
    if (!(m->dropout_p >= 0.f && m->dropout_p < 1.f)) return GMC_ERR_SHAPE;
    if (m->W1_slab && !gmc_aligned16(m->W1_slab)) return GMC_ERR_ALIGN;
    if (b->B > 0 && (b->n_max < 3 || b->n_max > GMC_MAX_GRAPH_NODES)) return GMC_ERR_GRAPH_SIZE;
    if (b->n_max > m->N) return GMC_ERR_SHAPE;  // more nodes than rows of conv1.weight
    return GMC_OK;
}

int group_rows(const gmc_batch *b) { return b->uniform_n > 0 ? b->uniform_n : b->n_max; }


// Y = act(dinv o (A @ X) + bias) over the batch with either implementation; Z0 (optional) gets
// the fused layer-2 feature transform (zparts partials on the LDS path).
int aggregate(const gmc_batch *b, const Workspace &w, const float *X, float *Y, int F, const float *bias, int relu,
              const float *W2, float *Z0, int tag, hipStream_t st) {
    if (w.fs)
        return gmc_spmm_lds_launch(b, X, w.ld, 1, 0, 0, b->dinv, bias, relu, Y, w.ld, 1, F, W2, Z0, tag, st);
    return gmc_spmm_launch(b->rowptr, b->gcol, nullptr, b->dinv, X, w.ld, bias, relu, Y, w.ld, b->R, F,
                           group_rows(b), W2, Z0, tag, st);
}

int forward_body(const gmc_batch *b, const gmc_model *m, const Workspace &w, hipStream_t st) {
    const int F = m->F;
    if (w.fs && fuse_enabled() && !dropout_on(m))  // T0 lives only in LDS
        return gmc_fwd1_lds_launch(b, m->W1, m->b1, m->W2, w.H, w.Z0, F, st, m->W1_slab, m->N);
    // layer 1 feature transform as a row gather of W1:  T0 = dinv o (A_val @ W1[:n])
    int rc = w.fs ? gmc_spmm_lds_launch(b, m->W1, F, 0, 1, 1, b->dinv, nullptr, 0, w.T0, w.ld, 1, F, nullptr,
                                        nullptr, GMC_K_GATHER_W1, st)
                  : gmc_spmm_launch(b->rowptr, b->lcol, b->vals, b->dinv, m->W1, F, nullptr, 0, w.T0, w.ld,
                                    b->R, F, group_rows(b), nullptr, nullptr, GMC_K_GATHER_W1, st);
    if (rc) return rc;
    if (dropout_on(m)) {  // relu -> dropout -> layer-2 feature transform of the DROPPED activations (:81-83)
        rc = aggregate(b, w, w.T0, w.H, F, m->b1, 1, nullptr, nullptr, GMC_K_AGG_FWD, st);
        if (rc) return rc;
        rc = gmc_dropout_launch(w.H, b->R, F, w.fs, w.ld, m->dropout_p, dropout_seed(m), st);
        if (rc) return rc;
        return gmc_hw2_rows_launch(w.H, b->dinv, m->W2, w.Z0, b->R, F, w.fs, w.ld, st);
    }
    // layer 1 aggregation + bias + relu with the layer 2 feature transform fused in
    return aggregate(b, w, w.T0, w.H, F, m->b1, 1, m->W2, w.Z0, GMC_K_AGG_FWD, st);
}

struct AdamFuse {  // optional Adam fused into the gradient fold (single GPU)
    float *param = nullptr, *m = nullptr, *v = nullptr;
    double lr = 0, beta1 = 0, beta2 = 0, eps = 0;
    int *step_counter = nullptr;
    float *w1_slab = nullptr;  // slab copy of W1 to refresh with the update (gmc_model.W1_slab)
};

// loss_tail: per-graph losses whose sum goes to the slot after the gradient (GMC_MODEL_GRAD_TAIL), or nullptr
int backward_body(const gmc_batch *b, const gmc_model *m, const Workspace &w, float *grad,
                  hipStream_t st, const AdamFuse *af = nullptr, const float *loss_tail = nullptr,
                  const gmc_bwd1_head *head = nullptr) {
    const long F = m->F;
    float *dW1 = grad, *db1 = grad + (long)m->N * F, *dW2 = db1 + F, *db2 = dW2 + F * 3;
    float *Gs = w.T0, *U = w.H;
    const float *W2b = m->W2;
    if (dropout_on(m)) {  // relu'(H) o mask / (1-p): the mask is the zeros of the stored H, the factor rides on W2
        if (af || !w.W2s) return GMC_ERR_UNSUPPORTED;
        int rc = gmc_scale_copy_launch(m->W2, w.W2s, (int)F * 3, 1.0f / (1.0f - m->dropout_p), st);
        if (rc) return rc;
        W2b = w.W2s;
    }
    if (w.fs && fuse_enabled() && gmc_bwd1_fits(b) && !dropout_on(m)) {  // one pass over H: Gs and U live only in LDS
        const int chunks = gmc_dw1_chunks(b->B, true, gmc_lds_slices(b, m->F)), per = (b->B + chunks - 1) / chunks;
        int rc = gmc_bwd1_lds_launch(b, w.H, w.GY2, m->W2, w.dw1part, w.part, m->F, chunks, per, st, head);
        if (rc) return rc;
        return gmc_finish_launch(w.dw1part, w.part, w.db2part, chunks, b->n_max, m->N, m->F, b->B, grad,
                                 af ? af->param : nullptr, af ? af->m : nullptr, af ? af->v : nullptr,
                                 af ? af->lr : 0, af ? af->beta1 : 0, af ? af->beta2 : 0, af ? af->eps : 0,
                                 af ? af->step_counter : nullptr, loss_tail, st, af ? af->w1_slab : nullptr);
    }
    if (af) return GMC_ERR_UNSUPPORTED;  // the fused Adam rides on the fused backward
    int rc = w.fs ? gmc_hidden_bwd_slab_launch(w.H, w.GY2, W2b, b->dinv, Gs, w.part, b->R, m->F, w.fs, st)
                  : gmc_hidden_bwd_launch(w.H, w.ld, w.GY2, W2b, b->dinv, Gs, w.ld, w.part, b->R, m->F, st);
    if (rc) return rc;
    rc = gmc_colsum_reduce_launch(w.part, w.fs ? gmc_hidden_slab_tiles(b->R) : gmc_hidden_tiles(b->R), m->F,
                                  dW2, db1, w.db2part, b->B, db2, st);
    if (rc) return rc;
    // conv1 backward aggregation:  U = dinv o (A @ Gs)
    rc = aggregate(b, w, Gs, U, m->F, nullptr, 0, nullptr, nullptr, GMC_K_AGG_BWD, st);
    if (rc) return rc;
    rc = gmc_dw1_launch(b, U, w.ld, dW1, w.dw1part, m->N, m->F, w.fs != 0, st);
    if (rc || !loss_tail) return rc;
    return gmc_loss_tail_launch(loss_tail, b->B, db2 + 3, st);
}

}  // namespace

// ---- timing probe -------------------------------------------------------------------
#include <vector>
namespace {
struct ProbeRec { int tag; hipEvent_t a, b; };
struct ProbeState {
    bool on = false;
    std::vector<ProbeRec> pool;
    size_t used = 0;
} g_probe;
}  // namespace

void gmc_probe_mark(int tag, bool begin, hipStream_t st) {
    if (!g_probe.on) return;
    if (begin) {
        if (g_probe.used >= g_probe.pool.size()) return;  // capacity exhausted: stop recording
        ProbeRec &r = g_probe.pool[g_probe.used];
        r.tag = tag;
        (void)hipEventRecord(r.a, st);
    } else {
        if (g_probe.used >= g_probe.pool.size()) return;
        ProbeRec &r = g_probe.pool[g_probe.used];
        if (r.tag != tag) return;
        (void)hipEventRecord(r.b, st);
        ++g_probe.used;
    }
}

extern "C" int gmc_probe_begin(int32_t capacity) {
    if (capacity < 0) return GMC_ERR_SHAPE;
    while ((int)g_probe.pool.size() < capacity) {
        ProbeRec r{-1, nullptr, nullptr};
        hipError_t e = hipEventCreate(&r.a);
        if (e == hipSuccess) e = hipEventCreate(&r.b);
        if (e != hipSuccess) return (int)e;
        g_probe.pool.push_back(r);
    }
    g_probe.used = 0;
    g_probe.on = capacity > 0;
    return GMC_OK;
}

extern "C" int gmc_probe_end(int32_t *tags, float *ms, int32_t max) {
    g_probe.on = false;
    const int n = (int)g_probe.used;
    if (n > 0) {
        hipError_t e = hipEventSynchronize(g_probe.pool[n - 1].b);
        if (e != hipSuccess) return -(int)e - 1000;
    }
    for (int i = 0; i < n && i < max; ++i) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_probe.pool[i].a, g_probe.pool[i].b);
        if (tags) tags[i] = g_probe.pool[i].tag;
        if (ms) ms[i] = t;
    }
    return n;
}

extern "C" int gmc_set_fuse(int on) {
    const int prev = fuse_enabled() ? 1 : 0;
    g_fuse = on ? 1 : 0;
    return prev;
}

extern "C" int gmc_version(void) { return GMC_VERSION; }

extern "C" const char *gmc_error_string(int code) {
    switch (code) {
        case GMC_OK: return "ok";
        case GMC_ERR_NULL: return "required pointer is NULL";
        case GMC_ERR_SHAPE: return "negative or inconsistent sizes";
        case GMC_ERR_CLASSES: return "number_classes must be 3 (override_fixed_nodes is 3-wide)";
        case GMC_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case GMC_ERR_WORKSPACE: return "workspace too small";
        case GMC_ERR_GRAPH_SIZE: return "graph has fewer than 3 or more than GMC_MAX_GRAPH_NODES nodes";
        case GMC_ERR_UNSUPPORTED: return "unsupported shape (the leading dimension F must be a multiple of 4 and <= 1024)";
        case GMC_ERR_ABI: return "gmc_batch.abi / gmc_model.abi differs from the library's GMC_VERSION: rebuild the caller against this include/gcnmaxcut.h";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown gmc error";
    }
}

// device-side address of pinned (hipHostMalloc / hipHostRegister) host memory, for callers that let the kernels
// write a result - the per-graph losses - straight into host memory they poll
extern "C" int gmc_host_device_pointer(void *pinned_host, void **device_ptr) {
    if (!pinned_host || !device_ptr) return GMC_ERR_NULL;
    return (int)hipHostGetDevicePointer(device_ptr, pinned_host, 0);
}

extern "C" size_t gmc_workspace_bytes(const gmc_batch *batch, const gmc_model *model, int training) {
    if (!batch || !model) return 0;
    return carve(batch, model, training, nullptr).bytes;
}

extern "C" int gmc_forward(const gmc_batch *batch, const gmc_model *model, float C, void *workspace,
                           size_t workspace_bytes, float *P, int32_t *S, float *loss,
                           gmc_stream_t stream) {
    int rc = check(batch, model);
    if (rc) return rc;
    if (!P || !workspace) return GMC_ERR_NULL;
    Workspace w = carve(batch, model, 0, workspace);
    if (w.bytes > workspace_bytes) return GMC_ERR_WORKSPACE;
    if (batch->R == 0) return GMC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = forward_body(batch, model, w, st);
    if (rc) return rc;
    return gmc_head_f32(batch, w.Z0, w.zparts, model->b2, C, P, S, loss, nullptr, nullptr, stream);
}

extern "C" int gmc_train_fwd_bwd(const gmc_batch *batch, const gmc_model *model, float C,
                                 void *workspace, size_t workspace_bytes, float *P, int32_t *S,
                                 float *loss, float *grad, gmc_stream_t stream) {
    int rc = check(batch, model);
    if (rc) return rc;
    if (!P || !workspace || !grad) return GMC_ERR_NULL;
    if (!gmc_aligned16(grad)) return GMC_ERR_ALIGN;
    Workspace w = carve(batch, model, 1, workspace);
    if (w.bytes > workspace_bytes) return GMC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tail = (model->flags & GMC_MODEL_GRAD_TAIL) != 0;
    if (tail && !loss) return GMC_ERR_NULL;
    if (batch->R == 0) {
        const size_t n = (size_t)model->N * model->F + model->F + (size_t)model->F * 3 + 3 + (tail ? 1 : 0);
        return (int)hipMemsetAsync(grad, 0, n * sizeof(float), st);
    }
    rc = forward_body(batch, model, w, st);
    if (rc) return rc;
    rc = gmc_head_f32(batch, w.Z0, w.zparts, model->b2, C, P, S, loss, w.GY2, w.db2part, stream);
    if (rc) return rc;
    return backward_body(batch, model, w, grad, st, nullptr, tail ? loss : nullptr);
}

extern "C" int gmc_adam_devstep_model_f32(float *, const float *, float *, float *, int32_t, int32_t, float *, double,
                                          double, double, double, int32_t *, gmc_stream_t);

extern "C" int gmc_train_step_f32(const gmc_batch *batch, int32_t N, int32_t F, float *param, float C,
                                  void *workspace, size_t workspace_bytes, float *P, int32_t *S, float *loss,
                                  float *grad, float *mom, float *var, double lr, double beta1, double beta2,
                                  double eps, int32_t *step_counter, float *w1_slab, gmc_stream_t stream) {
    if (!param || !grad || !mom || !var || !step_counter) return GMC_ERR_NULL;
    if (!gmc_aligned16(param) || !gmc_aligned16(mom) || !gmc_aligned16(var)) return GMC_ERR_ALIGN;
    const long nW1 = (long)N * F;
    gmc_model model{GMC_VERSION, N, F, 3, 0, param, param + nW1, param + nW1 + F, param + nW1 + F + (long)F * 3, 0.f, 0u, 0u, w1_slab};
    int rc = check(batch, &model);
    if (rc) return rc;
    if (!P || !workspace) return GMC_ERR_NULL;
    if (!gmc_aligned16(grad)) return GMC_ERR_ALIGN;
    Workspace w = carve(batch, &model, 1, workspace);
    if (w.bytes > workspace_bytes) return GMC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fused = batch->R > 0 && w.fs && fuse_enabled() && gmc_bwd1_fits(batch);
    if (!fused) {  // general shapes: gradient, then the stand-alone Adam
        rc = gmc_train_fwd_bwd(batch, &model, C, workspace, workspace_bytes, P, S, loss, grad, stream);
        if (rc) return rc;
        return gmc_adam_devstep_model_f32(param, grad, mom, var, N, F, w1_slab, lr, beta1, beta2, eps, step_counter, stream);
    }
    rc = forward_body(batch, &model, w, st);
    if (rc) return rc;
    // One graph per step (the reference's own schedule): the backward launch computes the head as well - every one of
    // its workgroups for itself, the rows (GY2, dinv) never leave the CU - three launches per graph-step instead of four
    const bool head_in_bwd = gmc_bwd1_takes_head(batch) && gmc_dw1_chunks(batch->B, true, gmc_lds_slices(batch, F)) == 1;
    // the head (launch) advances the step counter for the fused Adam of the finish kernel
    if (!head_in_bwd) {
        rc = gmc_head_launch(batch, w.Z0, w.zparts, model.b2, C, P, S, loss, w.GY2, w.db2part, step_counter, st);
        if (rc) return rc;
    }
    AdamFuse af;
    af.param = param; af.m = mom; af.v = var; af.lr = lr; af.beta1 = beta1; af.beta2 = beta2; af.eps = eps;
    af.step_counter = step_counter;
    af.w1_slab = w1_slab;
    const gmc_bwd1_head hd{w.Z0, w.zparts, model.b2, C, P, S, loss, w.db2part, step_counter};
    return backward_body(batch, &model, w, grad, st, &af, nullptr, head_in_bwd ? &hd : nullptr);
}

extern "C" int gmc_backward_from_gp(const gmc_batch *batch, const gmc_model *model, void *workspace,
                                    size_t workspace_bytes, const float *P, const float *GP,
                                    float *grad, gmc_stream_t stream) {
    int rc = check(batch, model);
    if (rc) return rc;
    if (!P || !GP || !workspace || !grad) return GMC_ERR_NULL;
    if (!gmc_aligned16(grad)) return GMC_ERR_ALIGN;
    Workspace w = carve(batch, model, 1, workspace);
    if (w.bytes > workspace_bytes) return GMC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (batch->R == 0) {
        const size_t n = (size_t)model->N * model->F + model->F + (size_t)model->F * 3 + 3;
        return (int)hipMemsetAsync(grad, 0, n * sizeof(float), st);
    }
    rc = gmc_head_bwd_launch(batch, P, GP, w.GY2, w.db2part, st);
    if (rc) return rc;
    return backward_body(batch, model, w, grad, st);
}
