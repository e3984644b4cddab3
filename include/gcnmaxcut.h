/* gcnmaxcut.h - C ABI of libgcnmaxcut_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ONE hot path of MJavaadAkhtar/GCN-max-cut: the 2-layer
 * GraphConv forward/backward, the 3-class cut loss and the Adam update behind
 * python/Training/TrainingNeural.py.  The reference is pure Python on top of
 * dgl.nn.pytorch.GraphConv + torch; it has no FFI of its own, so these are the entry
 * points a ctypes binding inside TrainingNeural.py would call (INTEGRATION.md shows
 * the stub).  Every comment below names the reference lines an entry point replaces.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE (HBM) pointers
 *    unless a comment says host; the library borrows them for the duration of the call.
 *  - every function returns int: 0 ok, <0 argument/shape error (gmc_error_string),
 *    >0 a hipError_t.  Nothing throws, nothing calls exit.
 *  - all work is enqueued on the caller's stream (`gmc_stream_t` == hipStream_t);
 *    no hidden synchronisation, no internal threads, no allocation, so every call can
 *    be captured into a hipGraph.
 *  - fp32 row-major, int32 indices.  Results are bitwise reproducible run to run
 *    (fixed summation orders, no float atomics).
 */
#ifndef GCNMAXCUT_H
#define GCNMAXCUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.2.0.  Incompatible with 0.1.x: gmc_batch and gmc_model start with an `abi` word and gmc_batch carries the
 * overflow lists, so callers compiled against a 0.1.x header must be recompiled ("ABI versions").
 * Every entry point that takes one of the two structs checks its `abi` word and returns GMC_ERR_ABI on a
 * mismatch instead of reading a differently laid out struct. */
#define GMC_VERSION 200

typedef void *gmc_stream_t; /* hipStream_t */

enum {
    GMC_OK = 0,
    GMC_ERR_NULL = -1,        /* required pointer is NULL */
    GMC_ERR_SHAPE = -2,       /* negative / inconsistent sizes */
    GMC_ERR_CLASSES = -3,     /* number_classes != 3: override_fixed_nodes is 3-wide (TrainingNeural.py:91-93) */
    GMC_ERR_ALIGN = -4,       /* pointer / leading dimension not 16-byte aligned */
    GMC_ERR_WORKSPACE = -5,   /* workspace too small */
    GMC_ERR_GRAPH_SIZE = -6,  /* a graph has < 3 or > GMC_MAX_GRAPH_NODES nodes */
    GMC_ERR_UNSUPPORTED = -7,
    GMC_ERR_ABI = -8          /* gmc_batch.abi / gmc_model.abi != GMC_VERSION: caller built against another header */
};

#define GMC_MAX_GRAPH_NODES 4096 /* per-graph head kernel keeps [n,3] tiles in LDS */

/* A block-diagonal batch of B graphs resident in HBM.  This is the CSR form of what
 * graphExtender.process_graphs_from_folder emits per graph (graphExtender.py:102-114:
 * the DGL graph == CSR structure, the padded adjacency == `vals` on that structure).
 * Built once per dataset by the host (gcn-max-cut_amd/graph.py). */
typedef struct gmc_batch {
    int32_t abi;           /* GMC_VERSION of the header the caller was compiled against (checked) */
    int32_t B;             /* graphs in the batch */
    int32_t R;             /* total nodes = sum n_g */
    int32_t nnz;           /* total directed edges = sum 2|E_g| */
    int32_t n_max;         /* max n_g */
    int32_t uniform_n;     /* n if every graph has n nodes, else 0 (XCD grouping hint) */
    int32_t nnz_max;       /* max directed edges of one graph (sizes the LDS index cache) */
    const int32_t *goff;   /* [B+1] first row of each graph */
    const int32_t *rowptr; /* [R+1] */
    const int32_t *gcol;   /* [nnz] neighbour as batch row id  (aggregation operand) */
    const int32_t *lcol;   /* [nnz] neighbour as local node id (row of W1 / column of X) */
    const float *vals;     /* [nnz] edge weight = X[u,v], or NULL when all ones */
    const float *dinv;     /* [R] clamp(degree,1)^-1/2  (in == out degree: undirected) */
    /* Optional ELL copy of the same structure for the LDS-tiled kernels (NULL: row kernels
     * only): W = ell_width slots per row (8 or 16), the row's FIRST min(degree, W) neighbours (CSR order) as
     * local ids in the slot order gmc_ell_arrange_host chose, padded with n_g..n_g+3 (all-zero tile rows),
     * weights 0.  Neighbours beyond the W-th of a row live in the overflow lists below. */
    const uint16_t *ell;   /* [R][W] */
    const float *ell_vals; /* [R][W] or NULL when all ones */
    int32_t ell_width;
    /* leading slots of a row that can hold a neighbour: 0 or ell_width = all of them; s < ell_width = slots
     * s..ell_width-1 of EVERY row are padding (gmc_ell_arrange_host lays a batch out that way when no row has
     * more than s neighbours, gmc_ell_slots_for tells s) - the LDS-tiled kernels then neither read nor add
     * those slots: 7 of 8 for d = 7 regular graphs (the headline workload), 12 of 16 for d = 12. */
    int32_t ell_slots;
    /* Overflow lists (NULL: no row of the batch has more than ell_width neighbours): the neighbours of row r
     * beyond its first ell_width, CSR order, in blocks of 8 local ids - blocks ovf_ptr[r] .. ovf_ptr[r+1]-1,
     * i.e. ids ovf_ids[8*ovf_ptr[r] ..), the last block of a row padded with n_g (weight 0).  A row with a
     * hub's degree then costs its own extra blocks (gathered by the wave that owns the row) instead of moving
     * the whole batch to the row kernels.  Summation order of a row: fixed (ELL slots and block entries dealt
     * over the lanes of a wave, then a butterfly), bitwise reproducible.  ovf_ptr with ovf_max_blocks == 0 (or
     * ovf_ids == NULL) is a batch without lists. */
    const int32_t *ovf_ptr;   /* [R+1] in blocks, or NULL */
    const uint16_t *ovf_ids;  /* [8 * ovf_ptr[R]] */
    const float *ovf_vals;    /* [8 * ovf_ptr[R]] or NULL when all ones */
    /* largest number of overflow blocks one graph of the batch has (host-known: the pointers above are device memory).
     * The fused LDS kernels keep a graph's blocks and one 16-bit descriptor per row in the few KB of LDS their tiles
     * leave over - at most 15 blocks per row, and as many per graph as fit (51 at n = 1000 with a 16-slot table, more
     * for smaller graphs); a batch beyond that, or one with edge weights AND overflow lists, runs on the row kernels. */
    int32_t ovf_max_blocks;
} gmc_batch;

/* GCNSoftmax parameters in DGL GraphConv layout (TrainingNeural.py:72-77):
 * conv1.weight [N,F], conv1.bias [F], conv2.weight [F,K], conv2.bias [K]. */
typedef struct gmc_model {
    int32_t abi;   /* GMC_VERSION of the header the caller was compiled against (checked) */
    int32_t N, F, K;
    int32_t flags; /* GMC_MODEL_* bits, 0 by default */
    const float *W1, *b1, *W2, *b2;
    /* F.dropout(h, p, training) between the layers (TrainingNeural.py:82; TrainingConfig.dropout :43).
     * 0 = identity: eval mode, and every reference configuration.  With p in (0,1) the entry points that
     * take a gmc_model apply H <- H o mask / (1-p) after relu, mask from a counter-based hash of
     * (dropout_seed, row, column) - reproducible from the seed, not torch's random stream - and run the
     * one-kernel-per-operation sequence.  gmc_backward_from_gp must be given the same p (and the workspace)
     * as the gmc_forward that produced P.  gmc_train_step_f32 has no dropout (it builds its own model). */
    float dropout_p;
    uint32_t dropout_seed_lo, dropout_seed_hi;
    /* Optional (NULL = not used): a copy of conv1.weight in the slab layout [ceil(F/16)][N][16] (element (r, c) at
     * ((c/16)*N + r)*16 + c%16, pad columns zero; gmc_w1_slab_floats floats).  The fused layer-1 forward streams the
     * 16 W1 columns of a slice into LDS per (graph, slice): from this copy a wave's LDS-DMA instruction reads one
     * contiguous KiB, from the [N,F] layout sixteen 64-byte pieces (8 % of the kernel's time).  The CALLER keeps
     * it equal to W1: gmc_w1_slab_f32 builds it; gmc_train_step_f32 and gmc_adam_devstep_model_f32 write the
     * updated W1 to both.  Results do not depend on it (same values, same order of operations). */
    const float *W1_slab;
} gmc_model;
/* gmc_train_fwd_bwd: grad has ONE more float after the N*F + F + F*3 + 3 gradient entries and
 * receives the sum of the batch's per-graph losses there (loss must be non-NULL).  Lets a
 * data-parallel caller carry the step's loss (TrainingNeural.py:387-388) through the gradient
 * all-reduce instead of a second collective. */
#define GMC_MODEL_GRAD_TAIL 1

int gmc_version(void);
const char *gmc_error_string(int code);

/* HOST helper (all pointers are host pointers): fills the ELL neighbour table of a batch from
 * its CSR.  Inside each group of four rows that share an LDS cycle of the tiled kernels the
 * neighbours are ordered over the W slots so that a slot's four fetches fall into different LDS
 * bank quarters where possible; padding entries are n_g .. n_g+3 (four all-zero tile rows).
 * The slot order is the summation order of the LDS-tiled kernels (fixed per batch).  When no row of the
 * batch is longer than s = gmc_ell_slots_for(...) < W the neighbours are arranged over slots 0..s-1 and slots
 * s..W-1 of every row are padding: set gmc_batch.ell_slots = s for such a batch.  Of a row longer than W
 * the first W neighbours (CSR order) are placed; the caller puts the rest into the overflow lists. */
int gmc_ell_arrange_host(int32_t B, const int32_t *goff, const int32_t *rowptr, const int32_t *lcol,
                         const float *vals, int32_t W, uint16_t *ell, float *ell_vals);
/* the slots gmc_ell_arrange_host uses for a batch with R rows (host pointer): the largest row length,
 * at least 7 (W = 8) / 9 (W = 16), at most W */
int gmc_ell_slots_for(int32_t R, const int32_t *rowptr, int32_t W);

/* Kernel tags reported by the timing probe (one per launch of the fused step). */
enum {
    GMC_K_GATHER_W1 = 0,  /* (X o dinv) @ W1 as a row gather of W1          TrainingNeural.py:80 */
    GMC_K_AGG_FWD = 1,    /* layer-1 aggregation + b1 + relu (+ fused H@W2)  :80-83 */
    GMC_K_HEAD = 2,       /* per-graph [n,3] head: softmax, decode, loss, GY2 :83-106,:154-176 */
    GMC_K_HIDDEN_BWD = 3, /* dW2/db1 partials + Gs                           backward of :81-83 */
    GMC_K_COLSUM = 4,     /* fold of the partials, db2 */
    GMC_K_AGG_BWD = 5,    /* conv1 backward aggregation                      backward of :80 */
    GMC_K_DW1 = 6,        /* dW1 gather-reduce over graphs */
    GMC_K_DW1_FOLD = 7,   /* fold of the dW1 chunk partials */
    GMC_K_ADAM = 8,       /* fused Adam                                      :386 */
    GMC_K_SPMM_USER = 9,  /* gmc_spmm_f32 called directly */
    GMC_K_DENSE_MFMA = 10,
    GMC_K_BWD1_FUSED = 11, /* hidden backward + conv1 backward aggregation + dW1, one pass over H (a one-graph
                            * gmc_train_step_f32 computes the head in this launch as well: no GMC_K_HEAD record) */
    GMC_K_FWD1_FUSED = 12, /* W1 gather + layer-1 aggregation (+ fused H@W2), one kernel */
    GMC_K_DECODE = 13,     /* post-processing sampler + cut count */
    GMC_K_FINISH = 14,     /* fold of the gradient partials (+ fused Adam) over the flat buffer */
    GMC_K_COUNT = 15
};

/* Timing probe for bench.py: between gmc_probe_begin and gmc_probe_end every kernel launch
 * of this library is bracketed by hipEventRecord on the stream it is launched on.
 * gmc_probe_end synchronises on the last event and returns the number of launches seen,
 * writing up to `max` (tag, milliseconds) pairs (host pointers). */
/* Kernel-sequence option: 1 (default) = fused layer kernels (T0 / Gs / U never leave LDS);
 * 0 = one kernel per operation (stand-alone SpMM, hidden backward, dW1).  Same results to
 * rounding.  Returns the previous setting.  Workspace sizes do not depend on it. */
int gmc_set_fuse(int on);

int gmc_probe_begin(int32_t capacity);
int gmc_probe_end(int32_t *tags, float *ms, int32_t max);

/* ---- building blocks (each is also used by the fused entry points below) ---------- */

/* Y[r,:] = act( scale[r] * sum_{e in row r} vals[e] * X[col[e],:] + bias ),  r < n_rows.
 * One CSR SpMM serves: the X@W1 row-gather (col=lcol, X=W1), DGL's update_all(copy_u,sum)
 * of GraphConv layer 1 (col=gcol, + bias + relu; TrainingNeural.py:80-81) and both
 * F-wide aggregations of its autograd backward (:385).  vals/scale/bias may be NULL.
 * group_rows > 0 asks for XCD-grouped scheduling: consecutive runs of that many rows
 * (one graph) are processed by workgroups of one XCD so neighbour rows are L2 hits.
 * If W2/Z0 are non-NULL (K must be 3) the layer-2 feature transform is fused into the
 * epilogue: Z0[r,:] = scale[r] * (Y[r,:] @ W2)   ((H*outdeg^-1/2)@W2, :83). */
int gmc_spmm_f32(const int32_t *rowptr, const int32_t *col, const float *vals,
                 const float *scale, const float *X, int64_t ldx, const float *bias, int relu,
                 float *Y, int64_t ldy, int32_t n_rows, int32_t F, int32_t group_rows,
                 const float *W2, float *Z0, gmc_stream_t stream);

/* Z0[r,:] = dinv[r] * (H[r,:] @ W2), K == 3, on the matrix cores (v_mfma_f32_16x16x4_f32).
 * Stand-alone form of the only dense contraction on the path (TrainingNeural.py:83). */
int gmc_dense_hw2_f32(const float *H, int64_t ldh, const float *dinv, const float *W2,
                      float *Z0, int32_t n_rows, int32_t F, gmc_stream_t stream);

/* Per-graph head: Z = dinv * (A @ Z0) + b2, P = softmax(Z) (TrainingNeural.py:83-84);
 * rows 0,1,2 forced to e0,e1,e2 and S = row-argmax, first max wins (:87-106);
 * loss[g] = -C * cut(S) (:154-176,:291-309).  When GY2 != NULL also the start of
 * loss.backward(): GP = C*A_val@onehot(S), softmax backward, db2part[g,:] = colsum(GZ),
 * GY2 = A @ (dinv*GZ).  P [R,3], S [R], loss [B], db2part [B,3], GY2 [R,4] = (GY2[r,0..2], dinv[r])
 * (16-byte rows so the backward kernels fetch a row's constants with one aligned load).
 * Z0 is [z_parts][R][3]: partial products of the LDS-tiled layer-1 kernel (one per column
 * slice group), folded here in ascending order; z_parts = 1 for a plain [R,3] Z0. */
int gmc_head_f32(const gmc_batch *batch, const float *Z0, int32_t z_parts, const float *b2, float C,
                 float *P, int32_t *S, float *loss, float *GY2, float *db2part, gmc_stream_t stream);

/* torch.optim.Adam.step (TrainingNeural.py:337,:386) over one flat buffer, fused:
 * m,v update + bias correction + parameter update in a single sweep. step >= 1.  The
 * hyper-parameters are doubles because torch derives 1-beta, lr/bias_correction1 and
 * sqrt(bias_correction2) in Python floats (doubles) before rounding them to fp32. */
int gmc_adam_f32(float *param, const float *grad, float *m, float *v, int64_t count, double lr,
                 double beta1, double beta2, double eps, int32_t step, gmc_stream_t stream);

/* Same update with the step number kept in device memory (*step_counter = steps done so far;
 * the call uses step_counter+1 and then increments it on the stream).  All arguments are
 * replay-invariant, so a whole epoch of the reference's one-step-per-graph schedule
 * (TrainingNeural.py:371-386) can be captured once into a hipGraph and replayed. */
int gmc_adam_devstep_f32(float *param, const float *grad, float *m, float *v, int64_t count, double lr,
                         double beta1, double beta2, double eps, int32_t *step_counter,
                         gmc_stream_t stream);

/* gmc_adam_devstep_f32 over the flat [W1 | b1 | W2 | b2] buffer of an N x F x 3 model; w1_slab (NULL = none)
 * receives the updated conv1.weight in the layout of gmc_model.W1_slab. */
int gmc_adam_devstep_model_f32(float *param, const float *grad, float *m, float *v, int32_t N, int32_t F,
                               float *w1_slab, double lr, double beta1, double beta2, double eps,
                               int32_t *step_counter, gmc_stream_t stream);

/* floats of, and the launch that builds, the slab copy of conv1.weight [N,F] described at gmc_model.W1_slab
 * (no counterpart in the reference: a layout the LDS-DMA of the fused forward reads in whole KiB pieces) */
size_t gmc_w1_slab_floats(int32_t N, int32_t F);
int gmc_w1_slab_f32(const float *W1, int32_t N, int32_t F, float *slab, gmc_stream_t stream);

/* ---- fused entry points ------------------------------------------------------------ */

/* `loss` of the entry points below may be device memory or PINNED HOST memory mapped into the device: each
 * graph's value is written with one system-scope store as soon as it is final (by the loss kernel, before the
 * backward kernels of the same call run), so a host thread watching that memory has the step's loss while the
 * rest of the step is still executing - the reference reads loss.item() every step (TrainingNeural.py:387-388).
 * gmc_host_device_pointer: the device-side address of such memory (hipHostGetDevicePointer; > 0 = hipError_t). */
int gmc_host_device_pointer(void *pinned_host, void **device_ptr);
/* n device floats -> pinned host memory (device-side address), one system-scope store each: how a data-parallel
 * rank hands the all-reduced loss of a step (the slot after the gradient, GMC_MODEL_GRAD_TAIL) to its host
 * thread before the optimizer kernels of that step run. */
int gmc_publish_f32(const float *src, int32_t n, float *pinned_dst, gmc_stream_t stream);

/* gmc_publish_f32 + gmc_adam_devstep_model_f32 as TWO launches instead of three: one wave stores the loss values and
 * advances *step_counter, the Adam sweep behind it uses the counter as it then stands (no trailing one-thread launch).
 * The tail of a data-parallel rank's step after the gradient all-reduce (TrainingNeural.py:386-388 on N GPUs). */
int gmc_publish_adam_devstep_model_f32(const float *publish_src, int32_t publish_n, float *pinned_dst, float *param,
                                       const float *grad, float *m, float *v, int32_t N, int32_t F, float *w1_slab,
                                       double lr, double beta1, double beta2, double eps, int32_t *step_counter,
                                       gmc_stream_t stream);

/* bytes of scratch gmc_forward / gmc_train_fwd_bwd need for this batch and model */
size_t gmc_workspace_bytes(const gmc_batch *batch, const gmc_model *model, int training);

/* GCNSoftmax.forward for every graph of the batch (TrainingNeural.py:79-85):
 * P[R,3] = softmax(conv2(relu(conv1(A_pad)))).  If S/loss are non-NULL also the decode
 * and loss of evaluate_model (:555-561). */
int gmc_forward(const gmc_batch *batch, const gmc_model *model, float C, void *workspace,
                size_t workspace_bytes, float *P, int32_t *S, float *loss, gmc_stream_t stream);

/* forward + loss + loss.backward() of train_single_epoch's loop body (:373-385) for the
 * SUM of the batch's per-graph losses.  grad is the flat [W1 | b1 | W2 | b2] buffer
 * (N*F + F + F*3 + 3 floats) and is overwritten (rows of dW1 no graph reaches are 0). */
int gmc_train_fwd_bwd(const gmc_batch *batch, const gmc_model *model, float C, void *workspace,
                      size_t workspace_bytes, float *P, int32_t *S, float *loss, float *grad,
                      gmc_stream_t stream);

/* One whole optimizer step of train_single_epoch's loop body (:373-386) for the batch:
 * gmc_train_fwd_bwd followed by Adam, with the gradient fold and the Adam update fused into
 * one sweep over the flat buffers (param/grad/m/v, each N*F + F + F*3 + 3 floats, 16-byte
 * aligned).  The step number lives in device memory as for gmc_adam_devstep_f32, so the call is
 * replay-invariant (hipGraph).  Single-GPU form: with data parallelism the all-reduce has to
 * sit between the gradient and Adam (gmc_train_fwd_bwd + all-reduce + gmc_adam_*).
 * w1_slab (NULL = none): the slab copy of conv1.weight (gmc_model.W1_slab), equal to it on entry; the forward
 * reads it and the Adam sweep writes the updated weights to it as well, so it stays equal. */
int gmc_train_step_f32(const gmc_batch *batch, int32_t N, int32_t F, float *param, float C, void *workspace,
                       size_t workspace_bytes, float *P, int32_t *S, float *loss, float *grad, float *m,
                       float *v, double lr, double beta1, double beta2, double eps, int32_t *step_counter,
                       float *w1_slab, gmc_stream_t stream);

/* Backward for a caller-supplied dLoss/dP (autograd.Function path: callers that build
 * their own loss from GCNSoftmax.forward's output, e.g. the reference's
 * override_fixed_nodes/apply_max_to_one_hot/compute_loss chain).  Requires the
 * workspace of the gmc_forward/gmc_train_fwd_bwd call that produced P. */
int gmc_backward_from_gp(const gmc_batch *batch, const gmc_model *model, void *workspace,
                         size_t workspace_bytes, const float *P, const float *GP, float *grad,
                         gmc_stream_t stream);

/* ---- decode / post-processing (the caller of the path in BASELINE configs[4]) -------- */

/* Random-sampling post-processing of Testing/TestingNeuralNetwork.py:18-98 for every graph of
 * the batch: `iters` samples of the node probabilities P [R,3] (nodes 0,1,2 of each graph fixed
 * to classes 0,1,2; node l >= 3 takes the first class whose running float32 sum exceeds its
 * uniform draw, last class as fallback), cut value of each sample, strictly-best sample kept.
 * uniforms: device doubles, for graph g `iters` x (n_g - 3) values starting at uoff[g] (device
 * int64 [B+1]) in the reference's draw order (iteration-major, node-minor).
 * Outputs (device): assign_all [iters][R] int8, cut_all [B][iters], best_assign [R] int32,
 * best_cut [B], best_iter [B]. */
int gmc_decode_sample_f32(const gmc_batch *batch, const float *P, const double *uniforms,
                          const int64_t *uoff, int32_t iters, int8_t *assign_all, float *cut_all,
                          int32_t *best_assign, float *best_cut, int32_t *best_iter,
                          gmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GCNMAXCUT_H */
