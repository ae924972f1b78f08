/*
 * yelprec_engine.h — C ABI of the MI355X (gfx950) engine for the BPR / NGCF / CDAE
 * hot path of twndus/YelpRecommendation.
 *
 * The reference has no FFI: its "operator interface" for this path is the set of
 * ATen ops its Python issues.  Each entry point below replaces one
 * such op group and cites the reference call site it stands in for.  The Python
 * classes in yelprecommendation_amd/ (same names and signatures as the reference's
 * models/ and trainers/ modules) are the only callers; INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the
 *     caller (PyTorch allocates tables, gradients, optimizer state, workspaces);
 *     the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); launches are
 *     asynchronous on it; functions are stateless and re-entrant;
 *   - tables are row-major contiguous float32 [rows, D]; indices are int64 (what the
 *     reference's DataLoader yields: data/datasets/mf_dataset.py:26-31);
 *   - supported embedding widths D: 16, 32, 64, 128 (YR_ERR_UNSUPPORTED otherwise);
 *   - return value: 0 on success, a positive hipError_t if a launch failed, or a
 *     negative YR_ERR_* for a rejected argument.  Nothing throws.
 *   - `err_flag` (int32, device, may be NULL): kernels that consume indices OR a
 *     bit into it when they meet an out-of-range index (YR_FLAG_*), skip that
 *     element and carry on, instead of faulting the GPU.
 */
#ifndef YELPREC_ENGINE_H
#define YELPREC_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YR_ENGINE_VERSION 28

#define YR_ERR_UNSUPPORTED (-1) /* embedding width / option not compiled in   */
#define YR_ERR_BADARG      (-2) /* null pointer, negative size, misalignment  */

#define YR_FLAG_BAD_USER 1
#define YR_FLAG_BAD_ITEM 2

#define YR_PULL_USER_PHASE 1 /* yr_bpr_mf_pull_apply: owner pass over the user rows          */
#define YR_PULL_ITEM_PHASE 2 /* yr_bpr_mf_pull_apply: item pass over [item_row_begin, end) */

#define YR_OPT_ADAM  0 /* torch.optim.Adam  : grad += wd * p               */
#define YR_OPT_ADAMW 1 /* torch.optim.AdamW : p *= 1 - lr * wd (decoupled) */

/* Number of float32 partial sums a loss-producing kernel writes (one per workgroup
 * slot, unused slots written as 0).  Callers size `loss_partials` with this.      */
#define YR_LOSS_PARTIALS 2048
/* A count that many workgroups add to (yr_cdae_decode_loss) is kept as YR_COUNT_SLOTS partial counts, one per
 * 128-byte line of a YR_COUNT_WORDS-word int32 buffer (word s * YR_COUNT_WORDS / YR_COUNT_SLOTS); its readers
 * (yr_gemm_f32_ex alpha_count, yr_cdae_hidden_bwd) add the slots up. */
#define YR_COUNT_SLOTS 64
#define YR_COUNT_WORDS 2048

/* Load check: returns YR_ENGINE_VERSION. */
int yr_engine_version(void);

/* The gfx arch string the code objects were built for ("gfx950"). */
const char *yr_engine_arch(void);

/* ---------------------------------------------------------------------------
 * MatrixFactorization.forward                      (reference models/mf.py:20-23)
 *   out[b] = sum_d U[user[b], d] * I[item[b], d]
 * replaces: 2 x nn.Embedding gather + torch.mul + torch.sum(dim=1).
 * ------------------------------------------------------------------------- */
int yr_mf_score(const float *U, const float *I,
                const int64_t *user, const int64_t *item,
                int64_t B, int D, int64_t num_users, int64_t num_items,
                float *out, int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * autograd of MatrixFactorization.forward  (triggered at trainers/mf_trainer.py:111)
 *   gradU[user[b]] += gout[b] * I[item[b]];  gradI[item[b]] += gout[b] * U[user[b]]
 * replaces: mul/sum backward + 2 x embedding_dense_backward (index_add_ into the
 * dense [rows, D] gradient).  gradU / gradI are ACCUMULATED into (caller zero-fills,
 * as optimizer.zero_grad() / autograd do in the reference).
 * ------------------------------------------------------------------------- */
int yr_mf_score_backward(const float *U, const float *I,
                         const int64_t *user, const int64_t *item, const float *gout,
                         int64_t B, int D, int64_t num_users, int64_t num_items,
                         float *gradU, float *gradI, int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * One fused BPR forward + backward over a batch of (user, pos, neg) triplets
 *   (reference trainers/mf_trainer.py:106-111 = models/mf.py:20-23 twice,
 *    loss.py:25-27, loss.backward()).
 *   x_b   = U[u_b] . (I[p_b] - I[n_b])
 *   loss  = mean_b softplus(-x_b)                      -> loss_partials (unscaled sums)
 *   g_b   = -sigmoid(-x_b) * inv_batch
 *   gradU[u_b] += g_b (I[p_b] - I[n_b]);  gradI[p_b] += g_b U[u_b];  gradI[n_b] -= g_b U[u_b]
 * `inv_batch` is 1/B for one GPU and 1/B_global when the batch is sharded by user
 * across ranks (keeps loss.py:27's mean over the global batch).
 * `gradU`/`gradI` may both be NULL: forward + loss only (MFTrainer.validate,
 * mf_trainer.py:118-132).  loss_partials: YR_LOSS_PARTIALS floats, fully overwritten.
 * ------------------------------------------------------------------------- */
int yr_bpr_mf_fwd_bwd(const float *U, const float *I,
                      const int64_t *user, const int64_t *pos, const int64_t *neg,
                      int64_t B, int D, int64_t num_users, int64_t num_items,
                      float inv_batch, float *gradU, float *gradI,
                      float *loss_partials, int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * The whole step for SMALL batches in two launches (the pull form below wins from a few
 * thousand triplets upwards): yr_bpr_mf_fwd_bwd's scatter, marking the rows it touches in
 * `touched` (uint8[num_users + num_items], all zero on entry and again on exit; may be NULL),
 * then yr_adam_dense_dual: dense Adam/AdamW over BOTH tables in one launch that reads / clears a
 * gradient row only where it was marked (every row still gets its update: dense semantics of
 * trainers/base_trainer.py:34-38) and reduces the loss partials into loss_out / loss_accum
 * (mf_trainer.py:114 without the host sync).  gradU / gradI: dense, zero on entry and on exit.
 * yr_adam_dense_dual on its own: p0/p1 any two tensors with n0/n1 elements (multiples of 4),
 * touched0/touched1 per row of `row_width` elements or NULL (gradient always read and cleared);
 * with marks, row_width / 4 must divide 64 (the lanes of a row share one wave: 4, 8, 16, 32, 64, 128,
 * 256 — anything else is YR_ERR_BADARG); loss_partials NULL = no loss reduction.
 * ------------------------------------------------------------------------- */
int yr_bpr_mf_scatter_step(float *U, float *I, float *gradU, float *gradI,
                           float *mU, float *vU, float *mI, float *vI, uint8_t *touched,
                           const int64_t *user, const int64_t *pos, const int64_t *neg,
                           int64_t B, int D, int64_t num_users, int64_t num_items, float inv_batch,
                           double lr, double step_size, double bc2_sqrt,
                           double beta1, double beta2, double eps, double weight_decay, int mode,
                           float *loss_partials, float *loss_out, double *loss_accum,
                           int32_t *err_flag, void *stream);
int yr_adam_dense_dual(float *p0, float *g0, float *m0, float *v0, int64_t n0,
                       float *p1, float *g1, float *m1, float *v1, int64_t n1,
                       int row_width, uint8_t *touched0, uint8_t *touched1,
                       double lr, double step_size, double bc2_sqrt,
                       double beta1, double beta2, double eps, double weight_decay, int mode,
                       const float *loss_partials, float loss_scale, float *loss_out, double *loss_accum,
                       void *stream);

/* ---------------------------------------------------------------------------
 * One whole BPR-MF optimisation step, pull-based (no float atomics, no global integer atomics,
 * no gradient buffers):
 *   reference trainers/mf_trainer.py:106-112 = 2 x forward + BPRLoss + loss.backward()
 *   + torch.optim.Adam/AdamW.step() (trainers/base_trainer.py:34-38), dense semantics
 *   (every row of both tables is updated, rows without contributions with grad = 0), and the
 *   `train_loss += loss.item()` of mf_trainer.py:114 without the host sync.
 * THREE launches on `stream` (csrc/bpr_pull.hip): a tile partition of the batch (every tile of
 * 1024..8192 triplets is counting-sorted in LDS by destination bucket of 1024/D rows, once by
 * user and once by item, into the tile's own region of the record arrays), an owner pass over the
 * user buckets (the owner workgroup of a bucket reads its segment of every tile, sorts it by row in
 * LDS and walks the sorted stream: scores, loss, user gradient in registers, Adam -> U_new; one
 * coefficient per triplet for the item side) and an owner pass over the item buckets (item gradient from the OLD user
 * rows, Adam in place; its first workgroup also reduces the loss partials in fixed order).
 * Limits: rows / (1024/D) <= 16383 buckets per table, num_users < 2^26 (YR_ERR_UNSUPPORTED beyond).
 *
 *   U_old  [num_users, D]  read;  U_new [num_users, D] written (must not alias U_old:
 *          the caller ping-pongs the two buffers between steps);
 *   I      [num_items, D]  updated in place;  mU,vU,mI,vI: Adam state, updated in place;
 *   gradI_out: NULL for one GPU.  If non-NULL the item pass writes the dense item gradient
 *          [num_items, D] there INSTEAD of applying Adam (mI/vI/I untouched): the caller
 *          all-reduces it across ranks and applies yr_adam_dense (user-sharded multi-GPU);
 *   inv_batch: 1 / (global batch size); loss_partials as for yr_bpr_mf_fwd_bwd;
 *   loss_out (float[1]) / loss_accum (double[1]), either may be NULL: the step's mean loss
 *          (sum of partials x inv_batch) is stored / added there by the item pass;
 *   lr..weight_decay, step_size, bc2_sqrt, mode: as for yr_adam_dense;
 *   deterministic: non-zero = bitwise reproducible results (the reference is, under a seed: SURVEY 8c):
 *          the contributions of a row are summed in triplet order (the owner pass re-ranks every row by
 *          triplet id after its LDS sort), chunks are cut at tile boundaries, and a tile segment larger
 *          than a chunk (a few rows taking most of a batch; tables of a few hundred rows) is taken in windows
 *          of triplet ids; costs a few percent;
 *   workspace: >= yr_bpr_mf_pull_workspace_bytes(B, num_users, num_items, D) bytes, 16-byte
 *          aligned, contents irrelevant on entry (a size computed for max_batch serves every
 *          B <= max_batch).
 * ------------------------------------------------------------------------- */
int64_t yr_bpr_mf_pull_workspace_bytes(int64_t max_batch, int64_t num_users, int64_t num_items, int D);
int yr_bpr_mf_pull_step(const float *U_old, float *U_new, float *I,
                        float *mU, float *vU, float *mI, float *vI, float *gradI_out,
                        const int64_t *user, const int64_t *pos, const int64_t *neg,
                        int64_t B, int D, int64_t num_users, int64_t num_items, float inv_batch,
                        double lr, double step_size, double bc2_sqrt,
                        double beta1, double beta2, double eps, double weight_decay, int mode,
                        int deterministic, void *workspace, int64_t workspace_bytes,
                        float *loss_partials, float *loss_out, double *loss_accum,
                        int32_t *err_flag, void *stream);

/* The same step in two phases, for callers that overlap work:
 *   yr_bpr_mf_pull_index  partitions the batch (launch 1) into `workspace`; it depends only on the
 *                         triplets, so it may be enqueued for batch k+1 while the all-reduce of
 *                         batch k is in flight (use a second workspace);
 *   yr_bpr_mf_pull_apply  runs the owner passes on a partition built for the same B, D,
 *                         num_users, num_items in the same workspace.  `phases` selects
 *                         YR_PULL_USER_PHASE and/or YR_PULL_ITEM_PHASE; the item phase covers item
 *                         rows [item_row_begin, item_row_end) only, so the item gradient can be
 *                         produced (and all-reduced) in chunks; both bounds must be multiples of
 *                         the bucket size 1024/D (item_row_end may also equal num_items).  The user
 *                         phase must have run for the batch before any item phase.  The call that
 *                         is given loss_out / loss_accum reduces the loss partials of the user
 *                         phase (in its item launch, or a one-workgroup launch if it has none):
 *                         pass them to exactly one call per batch, not before the user phase.
 * yr_bpr_mf_pull_step == index, then apply with both phases over all item rows.             */
int yr_bpr_mf_pull_index(const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B, int D,
                         int64_t num_users, int64_t num_items,
                         void *workspace, int64_t workspace_bytes, int32_t *err_flag, void *stream);
int yr_bpr_mf_pull_apply(const float *U_old, float *U_new, float *I,
                         float *mU, float *vU, float *mI, float *vI, float *gradI_out,
                         int64_t B, int D, int64_t num_users, int64_t num_items, float inv_batch,
                         double lr, double step_size, double bc2_sqrt,
                         double beta1, double beta2, double eps, double weight_decay, int mode,
                         int deterministic, void *workspace, int64_t workspace_bytes,
                         float *loss_partials, float *loss_out, double *loss_accum,
                         int phases, int64_t item_row_begin, int64_t item_row_end,
                         void *stream);
/* yr_bpr_mf_pull_apply_ordered: the same with a start order for the item pass.  Its workgroups start in slot
 * order and there are more item buckets (yr_bpr_mf_pull_item_buckets = ceil(num_items / (1024 / D))) than
 * resident workgroups, so the last ones start when the first finish: item_bucket_order[slot] = bucket, a
 * permutation of all item buckets with the heaviest (most occurrences per step) first, keeps the late starters
 * short.  Popularity is a property of the data set, so one order (e.g. from the train set's item degrees) serves
 * every step.  Results do not depend on it.  Ignored (as NULL) when the call covers only part of the item rows. */
int yr_bpr_mf_pull_item_buckets(int64_t num_items, int D);
int yr_bpr_mf_pull_apply_ordered(const float *U_old, float *U_new, float *I,
                         float *mU, float *vU, float *mI, float *vI, float *gradI_out,
                         int64_t B, int D, int64_t num_users, int64_t num_items, float inv_batch,
                         double lr, double step_size, double bc2_sqrt,
                         double beta1, double beta2, double eps, double weight_decay, int mode,
                         int deterministic, void *workspace, int64_t workspace_bytes,
                         float *loss_partials, float *loss_out, double *loss_accum,
                         int phases, int64_t item_row_begin, int64_t item_row_end,
                         const int32_t *item_bucket_order, void *stream);

/* ---------------------------------------------------------------------------
 * NGCF message passing            (reference models/ngcf.py:60-72, embedding_propagation:
 *   E' = leaky_relu( W1((L + I) E) + W2(E * (L E)) ), slope 0.01, Linear(x) = x @ W^T)
 *
 * yr_spmm_csr: Y = L X (accumulate = 0) or Y += L X (accumulate != 0), L in CSR with int32
 *   rowptr[n+1] / col[nnz] and float32 val[nnz], X and Y [n, D] (distinct buffers).
 *   replaces torch.sparse.mm(L, E) (models/ngcf.py:64,67; COO in the reference) — one SpMM
 *   serves both terms since (L + I)E = LE + E (no eye(N, N) temporary).  `heavy_rows` (may be
 *   NULL with n_heavy = 0) must list EXACTLY the rows with more than `heavy_threshold`
 *   non-zeros; a whole workgroup sums such a row, one wave every other row.
 * yr_ngcf_dense_fwd: Eout = leaky_relu((Z + E) W1^T + (E * Z) W2^T), W1/W2 [D, D] ([out, in]).
 * yr_ngcf_dense_bwd_data: dP = dEout * leaky_relu'(Eout); dZ = dP W1 + (dP W2) * E;
 *   dE += dP W1 + (dP W2) * Z.   W1T / W2T are the TRANSPOSED weights ([in, out] row-major).
 * yr_ngcf_dense_bwd_weight: dW1 += dP^T (Z + E); dW2 += dP^T (E * Z)   (float atomics on the
 *   2 D^2 outputs; caller zero-fills).
 * The backward SpMM (dE += L^T dZ) is yr_spmm_csr with accumulate = 1: L is symmetric.
 * yr_ngcf_score_fwd: the tail of NGCF.bpr_forward / forward (models/ngcf.py:44-58, :26-41): the
 *   layer outputs E_0..E_K ([num_users + num_items, D] each; users first) are concatenated along
 *   the feature axis and scored by a dot product = the sum over layers of per-layer dot products:
 *   out_pos[b] = sum_k <E_k[user[b]], E_k[num_users + pos[b]]>, out_neg likewise with neg.
 *   `layers` is a HOST array of n_layers (<= YR_NGCF_MAX_LAYERS) device pointers.  neg / out_neg
 *   may both be NULL (forward(user, item)).  Out-of-range ids: flag raised, score 0.
 * yr_ngcf_score_bwd: its autograd — dlayers[k] (host array of device pointers, caller zero-fills
 *   or accumulates) += gradient of every layer buffer, float atomics (like index_add_).
 * ------------------------------------------------------------------------- */
#define YR_NGCF_MAX_LAYERS 8
int yr_ngcf_score_fwd(const float *const *layers, int n_layers, const int64_t *user, const int64_t *pos,
                      const int64_t *neg, int64_t B, int D, int64_t num_users, int64_t num_items,
                      float *out_pos, float *out_neg, int32_t *err_flag, void *stream);
int yr_ngcf_score_bwd(const float *const *layers, float *const *dlayers, int n_layers, const int64_t *user,
                      const int64_t *pos, const int64_t *neg, const float *gpos, const float *gneg,
                      int64_t B, int D, int64_t num_users, int64_t num_items, int32_t *err_flag, void *stream);
int yr_spmm_csr(const int32_t *rowptr, const int32_t *col, const float *val,
                const float *X, float *Y, int64_t n, int D, int accumulate,
                const int32_t *heavy_rows, int64_t n_heavy, int heavy_threshold, void *stream);
/* The same product with the D floats of a row cut into slices of 16: the workgroups that share an XCD
 * (equal blockIdx % 8) gather ONE slice of 32 floats, whose half table (rows x 128 B) mostly stays in that
 * XCD's L2 (measured at Yelp2018 size: L2 hits 52 -> 76 %, fabric traffic 425 -> 219 MB per launch).
 * row_order (int32[n], may be NULL = 0..n-1): the order rows are visited in (e.g. by falling degree).  */
int yr_spmm_csr_sliced(const int32_t *rowptr, const int32_t *col, const float *val,
                       const float *X, float *Y, int64_t n, int D, int accumulate,
                       const int32_t *row_order, void *stream);
int yr_ngcf_dense_fwd(const float *E, const float *Z, const float *W1, const float *W2,
                      int64_t n, int D, float *Eout, void *stream);
int yr_ngcf_dense_bwd_data(const float *dEout, const float *Eout, const float *E, const float *Z,
                           const float *W1T, const float *W2T, int64_t n, int D,
                           float *dZ, float *dE, void *stream);
int yr_ngcf_dense_bwd_weight(const float *dEout, const float *Eout, const float *E, const float *Z,
                             int64_t n, int D, float *dW1, float *dW2, void *stream);

/* ---------------------------------------------------------------------------
 * NGCF, batch-aware propagation (ABI v28).  The reference propagates the WHOLE graph for every batch
 * (trainers/ngcf_trainer.py:102-117 -> models/ngcf.py:30-45) although bpr_forward reads layer K at the batch's
 * rows only (models/ngcf.py:37-39: last[u], last[U + p], last[U + n]).  With S_K = those rows and
 * S_{k-1} = S_k + neighbours(S_k), layer k is needed on S_k alone; these entry points compute exactly that and
 * give, on the rows they compute, bit-identical results to the full-graph entry points above.
 *
 * A row set is `flags` (int32[n], 1 = member), `rows` (int32[n]: the members in rows[0 .. *count), in no particular
 *   order) and `count` (int32 on the DEVICE: the host never waits for it).
 * yr_ngcf_frontier_mark: the set {user[b], num_users + pos[b], num_users + neg[b] : b < B} (neg may be NULL);
 *   out-of-range ids are skipped (yr_ngcf_score_fwd reports them).  clear != 0: flags and *count are zeroed first;
 *   clear = 0: the caller has zeroed them (or extends an existing set).
 * yr_ngcf_frontier_expand: the set (flags, rows, count) += the rows rows_in[0 .. *count_in) and all their
 *   neighbours in the CSR; max_rows_in (<= n) = the host's upper bound of *count_in (sizes the grid).
 * yr_spmm_csr_subset: yr_spmm_csr restricted to the rows with row_active[row] != 0 (int32 flags; other rows of Y
 *   untouched; NULL = all rows).  row_list / row_count / max_rows (optional, with row_active): the same set as a
 *   list — the rows with at most heavy_threshold non-zeros are then taken from the list (a set of a few rows
 *   costs a few waves instead of one early-exiting wave per graph row).
 * yr_ngcf_dense_{fwd,bwd_data,bwd_weight}_rows: the dense part of a layer over the rows rows[0 .. *count) instead
 *   of 0 .. n-1; max_rows (<= n) is the host's upper bound of *count and only sizes the grid (the workgroups
 *   stride over the list, so any count up to n is covered).  zero_rows (fwd, may be NULL): an [n, D] buffer whose
 *   listed rows are cleared by the same launch (the layer's gradient buffer: the backward pass touches no others).
 * ------------------------------------------------------------------------- */
int yr_ngcf_frontier_mark(const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B,
                          int64_t num_users, int64_t num_items,
                          int32_t *flags, int32_t *rows, int32_t *count, int clear, void *stream);
int yr_ngcf_frontier_expand(const int32_t *rowptr, const int32_t *col, int64_t n,
                            const int32_t *rows_in, const int32_t *count_in, int64_t max_rows_in,
                            int32_t *flags, int32_t *rows, int32_t *count, int clear, void *stream);
int yr_spmm_csr_subset(const int32_t *rowptr, const int32_t *col, const float *val,
                       const float *X, float *Y, int64_t n, int D, int accumulate,
                       const int32_t *heavy_rows, int64_t n_heavy, int heavy_threshold,
                       const int32_t *row_active,
                       const int32_t *row_list, const int32_t *row_count, int64_t max_rows, void *stream);
/* yr_spmm_csr_push_rows: Y[j] += sum over the listed rows r of L[r, j] * X[r] — for a symmetric L the product
 * Y += L X restricted to the columns rows[0 .. *count), as a scatter from those rows (float atomics); the cost is
 * the list's non-zeros, not the graph's.  The backward product of a layer whose dZ lives on a few rows. */
int yr_spmm_csr_push_rows(const int32_t *rowptr, const int32_t *col, const float *val,
                          const float *X, float *Y, int64_t n, int D,
                          const int32_t *rows, const int32_t *count, int64_t max_rows, void *stream);
int yr_ngcf_dense_fwd_rows(const float *E, const float *Z, const float *W1, const float *W2,
                           int64_t n, int D, float *Eout,
                           const int32_t *rows, const int32_t *count, int64_t max_rows, float *zero_rows,
                           void *stream);
int yr_ngcf_dense_bwd_data_rows(const float *dEout, const float *Eout, const float *E, const float *Z,
                                const float *W1T, const float *W2T, int64_t n, int D, float *dZ, float *dE,
                                const int32_t *rows, const int32_t *count, int64_t max_rows, void *stream);
int yr_ngcf_dense_bwd_weight_rows(const float *dEout, const float *Eout, const float *E, const float *Z,
                                  int64_t n, int D, float *dW1, float *dW2,
                                  const int32_t *rows, const int32_t *count, int64_t max_rows, void *stream);

/* ---------------------------------------------------------------------------
 * yr_ngcf_bpr_step: ONE NGCF training step, every launch issued from C (ABI v28) — replaces, per batch,
 *   pos, neg = model.bpr_forward(u, p, n, L); optimizer.zero_grad(); loss = BPRLoss(pos, neg); loss.backward();
 *   optimizer.step(); train_loss += loss.item()              (reference trainers/ngcf_trainer.py:104-116,
 *   models/ngcf.py:30-72, loss.py:25-27, trainers/base_trainer.py:34-38 with Adam / AdamW)
 * with the kernels declared above (the autograd route issues the same ones from Python, ~0.55 ms of host time per
 * step; this entry point costs the host ~0.1 ms).  Batch-aware: layer k + 1 runs on the rows the batch's scores
 * need when their estimated number (3B, x (1 + nnz / n) per hop down) is at most subset_fraction * n — a plan made
 * from host-known numbers only; 0 = always the whole graph.
 *   graph: CSR of the symmetric propagation matrix + heavy-row list as for yr_spmm_csr; users are nodes
 *   0 .. num_users-1, items the rest.
 *   params / exp_avg / exp_avg_sq: HOST arrays of 1 + 2K device pointers in the order
 *   [embedding.weight [n, D], W1.0 .. W1.K-1, W2.0 .. W2.K-1 ([D, D] each, [out, in])] — updated in place.
 *   step_size / bc2_sqrt as for yr_adam_dense.  loss_out[0] = the batch's mean loss, loss_accum[0] += it
 *   (either may be NULL).  workspace: yr_ngcf_step_workspace_bytes(n, D, K, B) bytes, 256-byte aligned, contents
 *   irrelevant on entry; a workspace sized for batch B serves every smaller batch.
 * ------------------------------------------------------------------------- */
int64_t yr_ngcf_step_workspace_bytes(int64_t n, int D, int K, int64_t B);
int yr_ngcf_bpr_step(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                     const int32_t *heavy_rows, int64_t n_heavy, int heavy_threshold, int64_t num_users,
                     float *const *params, float *const *exp_avg, float *const *exp_avg_sq, int K, int D,
                     const int64_t *user, const int64_t *pos, const int64_t *neg, int64_t B,
                     double lr, double step_size, double bc2_sqrt, double beta1, double beta2, double eps,
                     double weight_decay, int mode, double subset_fraction,
                     void *workspace, int64_t workspace_bytes, float *loss_out, double *loss_accum,
                     int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * CDAE                        (reference models/cdae.py:46-52, loss.py:12-16 and their autograd)
 *
 * yr_gemm_f32: C[M,N] (+)= op(A)[M,K] . op(B)[K,N] on v_mfma_f32_32x32x2_f32 (exact f32), row-major,
 *   transA/transB select A(m,k) = A[k*lda+m] / B(k,n) = B[n*ldb+k].  Plain store epilogue:
 *   C = act(acc + bias[n]) (bias may be NULL; act 0 = identity, 1 = sigmoid).  With split_k > 1 or
 *   accumulate != 0 the partial products are added atomically into C (caller pre-fills C; bias/act
 *   must then be NULL/0).  Replaces nn.Linear forward (x @ W^T + b: transB = 1), its input gradient
 *   (dy @ W) and its weight gradient (dy^T @ x: transA = 1).
 * yr_cdae_hidden_init: zpre[b,:] = bias + V[user[b],:]   (b_h + user_nodes(user_id), cdae.py:49).
 * yr_dropout: out = rnd >= p ? x / (1 - p) : 0, rnd uniform [0,1) supplied by the caller (nn.Dropout).
 * yr_dropout_seeded: the same with the uniforms drawn in the kernel (Philox4x32-10 keyed by `seed`,
 *   counter = index of the 4-element group): nothing but x is read.  x / out 16-byte aligned.
 * yr_sigmoid / yr_sigmoid_bwd: x = sigmoid(x) in place;  g = dy * y (1 - y)  (g may alias dy).
 * yr_colsum: out[c] (+)= sum_r X[r,c]                    (bias gradients).
 * yr_row_scatter_add: dV[user[b],:] += G[b,:]            (embedding_dense_backward of user_nodes).
 * yr_nsbce_fwd: NSBCELoss — positions with target + negative_mask != 0 (all positions when
 *   negative_mask is NULL: plain nn.BCELoss); stats[0] = mean BCE (log terms clamped at -100),
 *   stats[1] = number of selected positions.  workspace: 2 * YR_LOSS_PARTIALS floats.
 * yr_nsbce_bwd: dpred = gout[0] * (p - t) / max((1-p) p, 1e-12) / stats[1] on selected positions, else 0.
 * ------------------------------------------------------------------------- */
int yr_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K,
                const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                const float *bias, int act, int accumulate, int split_k, void *stream);
/* yr_gemm_f32_ex: yr_gemm_f32 with two extras (16-byte aligned operands, lda / ldb multiples of 4 only):
 *   alpha_count  device int32[YR_COUNT_WORDS] (a spread count, see YR_COUNT_SLOTS): the product is scaled by
 *                1 / count (by 0 when the count is 0) — the 1 / (number of loss positions) of a mean loss that
 *                only the device knows;
 *   rowsum       rowsum[m] = alpha * sum_k op(A)(m,k), from the operand values the MFMAs read anyway (the
 *                bias gradient beside dW = G^T z; split_k must be 1).
 * The three entry points below are the fused pieces of one CDAE training step (cdae_trainer.py:36-54:
 * forward, NS-BCE / BCE, backward), driven by yelprecommendation_amd/cdae_step.py:
 * yr_cdae_decode_loss: y = act(z W_o^T + b_o) on the matrix cores with the loss in the epilogue:
 *   G[b,i] = (y - t) / max((1 - y) y, 1e-12) * act'(y) on the selected positions (target + negative_mask != 0;
 *   every position when negative_mask is NULL), else 0 — the gradient w.r.t. the decoder's pre-activation
 *   WITHOUT its 1 / count factor; partial_loss[yr_cdae_decode_loss_partials(B, I)] = per-workgroup sums of the
 *   clamped BCE terms; count (int32[YR_COUNT_WORDS], all zero on entry) += number of selected positions,
 *   spread over its slots.  pred (may be NULL)
 *   receives y.  z / Wo 16-byte aligned, H a multiple of 4; ldg >= I = leading dimension of G and pred (a
 *   multiple of 4 lets the gradient products that read G take the tiled kernel).
 * yr_cdae_hidden_bwd: dz <- dz * act'(z) in place (scale_dz != 0: dz / count first), dbh = column sums, dV[user[b],:] += dz[b,:] with
 *   touched_users[user[b]] = 1 (may be NULL); with n_partials > 0 also the step's loss: stats[0] = (sum of
 *   partial_loss in fixed order) / count, stats[1] = count, *loss_accum += stats[0] (may be NULL). */
int yr_gemm_f32_ex(int transA, int transB, int64_t M, int64_t N, int64_t K,
                   const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                   const float *bias, int act, int accumulate, int split_k,
                   const int32_t *alpha_count, float *rowsum, void *stream);
int64_t yr_cdae_decode_loss_partials(int64_t B, int64_t I);
int yr_cdae_decode_loss(const float *z, const float *Wo, const float *bo, const float *target,
                        const float *negative_mask, int64_t B, int64_t I, int H, int act, float *G,
                        int64_t ldg, float *pred, float *partial_loss, int32_t *count, void *stream);
int yr_cdae_hidden_bwd(float *dz, const float *z, int act, const int64_t *user, int64_t B, int H,
                       int64_t num_users, float *dV, uint8_t *touched_users, float *dbh,
                       const float *partial_loss, int64_t n_partials, const int32_t *count,
                       float *stats, double *loss_accum, int scale_dz, void *stream);
/* The decoder of a TRAINING step on the loss positions only (NSBCELoss reads the prediction where
 * target + negative_mask != 0 and nowhere else — loss.py:14-16 — so its gradient w.r.t. every other position
 * is exactly zero and the three dense decoder products reduce to a pass over the position lists):
 * yr_cdae_compact_pair: yr_cdae_compact_rows (lists of dropout_p(x), the encoder's input) and, from the same
 *   pass over x and negative_mask, the loss positions of every row as (column, target) lists in the same
 *   32-sub-list layout (loss_cols / loss_targets / loss_count sized like cols / vals / count).
 * yr_cdae_sampled_decode: per position (b, i): y = act(z[b] . W_o[i] + b_o[i]), its BCE term (clamped at -100)
 *   into partial_loss[b * yr_cdae_sampled_decode_splits(B) + s] (fixed order), g = (y - t) / max((1 - y) y,
 *   1e-12) * act'(y), then dz[b,:] += g W_o[i,:] (dz zero on entry), dW_o[i,:] += g z[b,:], db_o[i] += g
 *   (float atomics; both buffers zero on entry), count (spread,
 *   YR_COUNT_WORDS, zero on entry) += positions.  Nothing carries the 1 / count of the mean: the consumers apply
 *   it (yr_cdae_hidden_bwd scale_dz = 1, yr_adam_dense_flat scaled[k] = 1).  H a multiple of 4, <= 256. */
int yr_cdae_compact_pair(const float *x, const float *negative_mask, int64_t B, int64_t I, uint64_t seed,
                         double p, int32_t *cols, float *vals, int32_t *count, int32_t *loss_cols,
                         float *loss_targets, int32_t *loss_count, void *stream);
/* yr_cdae_train_lists: a training batch as lists straight from the per-user item CSR (ptr / idx: int64, ids
 *   ascending and distinct inside a user), no dense [B, I] row or mask (cdae_dataset.py:36-59 builds both per
 *   user on the host): cols / vals / count = the user's items with nn.Dropout(p) applied (the mask
 *   yr_cdae_compact_rows gives the dense row: Philox word of flat position b * I + column, seed drop_seed);
 *   loss_* = its NS-BCE positions (column, target): the positives (1) and exactly neg_times * positives distinct
 *   non-positive items (0), every subset equally likely (np.random.choice(replace=False), cdae_dataset.py:27:
 *   the first distinct non-positive values of a uniform Philox sequence keyed by neg_seed and the row).  All six
 *   buffers sized and laid out as for yr_cdae_compact_pair.  ptr2 / idx2 (may be NULL): a second CSR whose items
 *   are positives of the LOSS list as well (validation: target = train + held-out items, cdae_trainer.py:67) and
 *   count towards neg_times, but do not enter the encoder list.  err_flag: YR_FLAG_BAD_USER, YR_FLAG_BAD_ITEM (bad or
 *   repeated item id, or more negatives wanted than non-positives exist).  I <= 245,760 (163,840 with ptr2). */
int yr_cdae_train_lists(const int64_t *ptr, const int64_t *idx, const int64_t *ptr2, const int64_t *idx2,
                        const int64_t *users, int64_t B,
                        int64_t num_users, int64_t I, int neg_times, uint64_t neg_seed,
                        uint64_t drop_seed, double p, int32_t *cols, float *vals, int32_t *count,
                        int32_t *loss_cols, float *loss_targets, int32_t *loss_count,
                        int32_t *err_flag, void *stream);
/* yr_cdae_train_lists_batched: the lists of SEVERAL batches in one launch — row r is row r % batch_rows of batch
 *   r / batch_rows and takes neg_seeds / drop_seeds [r / batch_rows] (device arrays): exactly the lists one
 *   yr_cdae_train_lists call per batch gives, B = all rows.  yr_cdae_loss_finalize_batched: after ONE
 *   yr_cdae_sampled_decode over those rows (dz = dWo = dbo = NULL), *loss_accum += sum over the batches of
 *   (sum of the batch's loss partials / its position count) — CDAETrainer.validate's `valid_loss += loss`
 *   (trainers/cdae_trainer.py:56-88) for all those batches; splits = yr_cdae_sampled_decode_splits(B) of that
 *   launch, means: float[number of batches] scratch, arrive: int32[1], zero on entry and on exit. */
int yr_cdae_train_lists_batched(const int64_t *ptr, const int64_t *idx, const int64_t *ptr2, const int64_t *idx2,
                                const int64_t *users, int64_t B, int64_t num_users, int64_t I, int neg_times,
                                const uint64_t *neg_seeds, const uint64_t *drop_seeds, int64_t batch_rows, double p,
                                int32_t *cols, float *vals, int32_t *count, int32_t *loss_cols, float *loss_targets,
                                int32_t *loss_count, int32_t *err_flag, void *stream);
int yr_cdae_loss_finalize_batched(const float *partial_loss, int splits, const int32_t *loss_count, int64_t rows,
                                  int64_t batch_rows, float *means, int32_t *arrive, double *loss_accum,
                                  void *stream);
int yr_cdae_sampled_decode_splits(int64_t B);
/* yr_cdae_sampled_decode with dz = dWo = dbo = NULL computes the loss partials and the count only (validation);
 * yr_cdae_loss_finalize then gives stats[0] = sum(partials) / count (fixed order), stats[1] = count,
 * *loss_accum += stats[0] (may be NULL). */
int yr_cdae_loss_finalize(const float *partial_loss, int64_t n_partials, const int32_t *count, float *stats,
                          double *loss_accum, void *stream);
int yr_cdae_sampled_decode(const int32_t *loss_cols, const float *loss_targets, const int32_t *loss_count,
                           const float *z, const float *Wo, const float *bo, int64_t B, int64_t I, int H,
                           int act, float *dz, float *dWo, float *dbo, float *partial_loss,
                           int32_t *count, void *stream);
int yr_cdae_hidden_init(float *zpre, const float *bias, const float *V, const int64_t *user,
                        int64_t B, int H, int64_t num_users, int32_t *err_flag, void *stream);
int yr_dropout(const float *x, const float *rnd, double p, int64_t n, float *out, void *stream);
int yr_sigmoid(float *x, int64_t n, void *stream);
int yr_dropout_seeded(const float *x, uint64_t seed, double p, int64_t n, float *out, void *stream);
int yr_sigmoid_bwd(const float *dy, const float *y, float *g, int64_t n, void *stream);
int yr_colsum(const float *X, int64_t rows, int64_t cols, float *out, int accumulate, void *stream);
int yr_row_scatter_add(const float *G, const int64_t *user, int64_t B, int H, int64_t num_users,
                       float *dV, void *stream);
int yr_nsbce_fwd(const float *pred, const float *target, const float *negative_mask, int64_t n,
                 float *workspace, float *stats, void *stream);
int yr_nsbce_bwd(const float *pred, const float *target, const float *negative_mask,
                 const float *stats, const float *gout, int64_t n, float *dpred, void *stream);

/* loss_out[0] = scale * sum(loss_partials);  if loss_accum: loss_accum[0] += same.
 * (`train_loss += loss.item()` of mf_trainer.py:114 without the per-step host sync;
 *  loss_accum is float64 like the Python float it replaces.)                        */
int yr_loss_finalize(const float *loss_partials, float scale,
                     float *loss_out, double *loss_accum, void *stream);

/* ---------------------------------------------------------------------------
 * Full-catalogue scores for evaluation      (reference trainers/mf_trainer.py:138-140:
 *   for each eval user  pred = model([user] * num_items, arange(num_items)) )
 *   scores[r, j] = U[users[r]] . I[j]   for r < nrows, j < num_items
 * as one float32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products and
 * fma-chained accumulation, so rankings match the CPU reference except at float near-ties).
 * scores: [nrows, row_stride] floats, row_stride >= num_items.  Feed to yr_topk_masked.
 * ------------------------------------------------------------------------- */
int yr_mf_scores_gemm(const float *U, const float *I, const int64_t *users, int64_t nrows, int D,
                      int64_t num_users, int64_t num_items, float *scores, int64_t row_stride,
                      int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * Fused evaluation: scores + mask + top-k without the score matrix
 *   (reference trainers/mf_trainer.py:134-144 + :163-178 for all eval users at once)
 *   out[r, 0..k) = the k best items of user users[r] by U[users[r]] . I[j], after forcing the
 *   scores of the items in mask_idx[mask_ptr[r] .. mask_ptr[r+1]) to mask_value; score
 *   descending, item id ascending among equal scores.  k <= 32 (the lists live in registers: 4, 10, 16 or 32
 *   entries); k > 16 up to D = 64 only (YR_ERR_UNSUPPORTED beyond: yr_mf_scores_gemm + yr_topk_masked).
 * The mask lists must be sorted ASCENDING inside each row (the kernel walks them with a cursor as
 * it sweeps the catalogue).  mask_ptr may be NULL.
 * mode: how the matrix cores compute the f32 scores —
 *   YR_EVAL_F32     v_mfma_f32_32x32x2_f32 on the f32 tables (as yr_mf_scores_gemm);
 *   YR_EVAL_BF16X3  every operand as the sum of three bfloat16 terms (x = x1 + x2 + x3 to 2^-27 |x|), every product
 *                   as its six partial products of weight >= 2^-18, accumulated in f32 by v_mfma_f32_32x32x16_bf16:
 *                   what is dropped is below 2^-25 |x y|, less than the f32 rounding of the product, so the scores
 *                   differ from YR_EVAL_F32 by what two f32 summation orders differ by, at 3/8 of the matrix-core
 *                   cycles.  The item planes (6 D bytes per item) are rebuilt from I by every call.
 * workspace: yr_mf_eval_topk_workspace_bytes(nrows, num_items, D, k, mode) bytes of device memory, used in this
 *   order: the item planes of YR_EVAL_BF16X3 (yr_mf_eval_topk_planes_bytes(num_items, D): required in that mode,
 *   YR_ERR_BADARG without); 4 bytes per row for the thresholds of hint lists; the group maxima of the prescan; the
 *   per-slice partial lists when the catalogue is cut into slices to fill the chip.  Everything after the planes is
 *   optional: without room for a part, the hint is ignored / the prescan skipped / the kernel runs unsliced — the
 *   same result, slower.
 * ------------------------------------------------------------------------- */
#define YR_EVAL_F32 0
#define YR_EVAL_BF16X3 1
/* or-ed into mode.  By default catalogues of 16,384 items and more with k > 4 get a PRESCAN launch before the sweep:
 * the scores of a sample of the catalogue (an eighth, 4,096 items at most) per user, reduced to 32 group maxima per
 * user and workgroup, whose k-th largest is a lower bound of the user's k-th best score; the sweep's lists start
 * from it instead of from -inf (three to four times fewer candidates to insert).  The result is the same with and
 * without (the bound comes from the same scores).  Needs 128 bytes x slices per row of workspace (after the
 * planes); skipped without room. */
#define YR_EVAL_NO_PRESCAN 2
#define YR_EVAL_FORCE_PRESCAN 4   /* prescan whatever the catalogue size (tests) */
/* The sweep has two forms with the same scores and the same lists.  FOUR WAVES: workgroups of 128 users, three per
 * CU, every wave alternating between the matrix instructions and the rest of a tile.  TWO ROLES (YR_EVAL_BF16X3;
 * D = 64 with k <= 16, D = 128 with k <= 10): eight-wave workgroups of 256 users, one per CU; the two waves of a SIMD
 * alternate between the matrix instructions of a tile and everything else, a workgroup barrier between the
 * intervals.  The library picks, from 2,048 rows: two roles at D = 128 (sweep 1.90 -> 1.55 ms at Yelp2018 size) and
 * at D = 64 for 16-entry lists with thresholds from hint lists (1.27 -> 1.14 ms); four waves otherwise (D = 64, k <= 10:
 * equal with hints, 10 % faster without).  These flags, or-ed into mode, force one form where both exist and are
 * ignored elsewhere (tests, comparisons; not both). */
#define YR_EVAL_TWO_ROLES 8
#define YR_EVAL_FOUR_WAVES 16
/* hint (may be NULL): int64 [nrows, k], any k item ids per row — typically `out` of the previous evaluation of the
 * same rows (the same buffer may be passed as hint and out).  k DIFFERENT items whose scores are all >= b prove that
 * the row's k-th best score is >= b, so the lists start from the smallest hint score (lowered by more than f32
 * rounding can account for; masked hints count with mask_value) instead of from -inf or from the prescan's bound,
 * and the prescan launch is skipped.  Rows whose hint holds an id outside [0, num_items) — e.g. the -1 padding of a
 * short list — or a repeated id get no bound.  The result never depends on the hint; a model that moved little since
 * the hint was computed keeps most of its top-k, and the sweep then inserts an order of magnitude fewer candidates.
 * Needs 4 bytes per row of workspace (after the planes; always part of yr_mf_eval_topk_workspace_bytes). */
int64_t yr_mf_eval_topk_planes_bytes(int64_t num_items, int D);
int64_t yr_mf_eval_topk_workspace_bytes(int64_t nrows, int64_t num_items, int D, int k, int mode);
int yr_mf_eval_topk(const float *U, const float *I, const int64_t *users, int64_t nrows, int D,
                    int64_t num_users, int64_t num_items,
                    const int64_t *mask_ptr, const int64_t *mask_idx, float mask_value,
                    int k, int64_t *out, void *workspace, int64_t workspace_bytes, int mode,
                    const int64_t *hint, int32_t *err_flag, void *stream);
/* yr_mf_eval_topk_bias: the same with scores U[users[r]] . I[j] + item_bias[j] (item_bias NULL: yr_mf_eval_topk).
 *   The evaluation of CDAE (trainers/cdae_trainer.py:90-144) for ALL users at once: U = the hidden rows z
 *   [users, H], I = output_layer.weight [items, H], item_bias = output_layer.bias — sigmoid is monotone, so the
 *   top-k of the pre-activations is the top-k of pred, and the reference's multiply-mask (seen items -> 0, below
 *   every sigmoid output) is mask_value = -3.40282e+38 here. */
int yr_mf_eval_topk_bias(const float *U, const float *I, const float *item_bias, const int64_t *users,
                         int64_t nrows, int D, int64_t num_users, int64_t num_items,
                         const int64_t *mask_ptr, const int64_t *mask_idx, float mask_value,
                         int k, int64_t *out, void *workspace, int64_t workspace_bytes, int mode,
                         const int64_t *hint, int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * Masked row-wise top-k      (reference trainers/mf_trainer.py:163-178,
 *                             trainers/ngcf_trainer.py:167-182, trainers/cdae_trainer.py:123-144)
 *   for each row r of scores[nrows, ncols] (row pitch row_stride floats):
 *     s[c] = mask_value for c in mask_idx[mask_ptr[r] .. mask_ptr[r+1])   (CSR over rows)
 *     out[r, 0..k) = the k best column ids, score descending, id ascending on ties
 * mask_value = -3.40282e+38 reproduces `pred[mask_items] = -3.40282e+38`; 0 reproduces
 * CDAE's `pred * logical_not(input_mask)`.  mask_ptr may be NULL (no mask).  k <= 64.
 * Rows shorter than k are padded with -1.  `scores` is not modified.  ncols <= 2^19 - 2048 (the row's
 * mask bitmap lives in LDS; YR_ERR_UNSUPPORTED beyond).
 * mask_rows (may be NULL): row r's mask list is CSR row mask_rows[r] instead of r.
 * replaces: numpy fancy-index store + argpartition + take_along_axis + argsort per user.
 * ------------------------------------------------------------------------- */
int yr_topk_masked(const float *scores, int64_t nrows, int64_t ncols, int64_t row_stride,
                   const int64_t *mask_ptr, const int64_t *mask_idx, const int64_t *mask_rows,
                   float mask_value, int k, int64_t *out, void *stream);

/* ---------------------------------------------------------------------------
 * Ranking metrics on the device      (reference metric.py:7-109, with its quirks: recall / MAP /
 *   NDCG skip users with an empty `actual` and shrink the denominator; AP truncates `actual` too
 *   and divides by len(actual); DCG scans positions 1..min(len(actual), k) only)
 *   topk [n, k] int64 (e.g. from yr_mf_eval_topk), pos_ptr [n+1] / pos_idx: the held-out items of
 *   each row in their ORIGINAL order (AP depends on it).
 *   out (10 float64): [0..3] = precision@k, recall@k, MAP@k, NDCG@k, [4] = users with non-empty
 *   `actual`, [5..8] = the four un-normalised sums and [9] = n (a user-sharded evaluation sums
 *   [4..9] over ranks and divides once).  workspace: yr_rank_metrics_workspace_bytes(n) bytes.
 *   pos_rows (may be NULL): row r's list is CSR row pos_rows[r] instead of r — a batch of users can
 *   point into one per-user CSR that lives on the device (no per-batch CSR is built).
 * ------------------------------------------------------------------------- */
int64_t yr_rank_metrics_workspace_bytes(int64_t n);
int yr_rank_metrics(const int64_t *topk, int64_t n, int k, const int64_t *pos_ptr, const int64_t *pos_idx,
                    const int64_t *pos_rows, double *workspace, double *out, void *stream);

/* ---------------------------------------------------------------------------
 * BPRLoss.forward / backward on score vectors        (reference loss.py:25-27)
 *   loss = mean_b( -logsigmoid(pos[b] - neg[b]) )  -> loss_partials (unscaled sums;
 *   finish with yr_loss_finalize(scale = 1/B));
 *   gpos[b] = -sigmoid(-(pos[b]-neg[b])) * gout[0] * inv_batch,  gneg[b] = -gpos[b].
 * For callers that keep the reference's three-call shape (two model() calls, then
 * the loss); the fused yr_bpr_mf_fwd_bwd above never materialises pos/neg.
 * ------------------------------------------------------------------------- */
int yr_bpr_loss_fwd(const float *pos, const float *neg, int64_t B,
                    float *loss_partials, void *stream);
int yr_bpr_loss_bwd(const float *pos, const float *neg, const float *gout,
                    float inv_batch, int64_t B, float *gpos, float *gneg, void *stream);

/* ---------------------------------------------------------------------------
 * Dense Adam / AdamW step over n float32 elements
 *   (reference trainers/base_trainer.py:34-38 -> torch.optim.Adam/AdamW.step,
 *    torch defaults; the CPU single-tensor formula, see oracle/adam.py).
 * Hyper-parameters are doubles (torch keeps them as Python floats); derived scalars
 * (1 - beta1, 1 - beta2, 1 - lr*wd) are formed in double and rounded to float32 once,
 * as torch does when a Python scalar meets a float32 tensor.  The caller supplies
 *   step_size = lr / (1 - beta1^t),  bc2_sqrt = sqrt(1 - beta2^t).
 * zero_grad != 0 also clears g (the next step's optimizer.zero_grad()).
 * Any n >= 0; all pointers must be 16-byte aligned (YR_ERR_BADARG otherwise).
 * ------------------------------------------------------------------------- */
int yr_adam_dense(float *p, float *g, float *m, float *v, int64_t n,
                  double lr, double step_size, double bc2_sqrt,
                  double beta1, double beta2, double eps, double weight_decay,
                  int mode, int zero_grad, void *stream);

/* Dense SGD (no momentum): p -= lr * (g + wd * p)   (base_trainer.py:39-40). */
/* yr_adam_dense_multi: the same update for up to YR_ADAM_MULTI_MAX tensors in one launch (the small
 * weight matrices and biases beside an embedding table; all at the same step count).  p / g / m / v
 * / n are HOST arrays of `count` device pointers / element counts; no alignment requirement. */
#define YR_ADAM_MULTI_MAX 16
int yr_adam_dense_multi(float *const *p, float *const *g, float *const *m, float *const *v,
                        const int64_t *n, int count, double lr, double step_size, double bc2_sqrt,
                        double beta1, double beta2, double eps, double weight_decay, int mode,
                        int zero_grad, void *stream);
/* yr_adam_dense_flat: up to YR_ADAM_MULTI_MAX tensors of ANY size in one launch at 16 bytes per lane
 * (every buffer 16-byte aligned).  touched[k] (HOST array of device pointers,
 * entries may be NULL): one byte per row of row_width[k] floats (row_width / 4 a power of two <= 64) — the
 * gradient of a row is read, cleared and unmarked only where the mark is set (every row is still updated, with
 * grad = 0 elsewhere).  clear[k] = 1: the gradient is cleared after it is read; 2: only where it is non-zero.  scaled[k] != 0 (with
 * grad_count, a spread count, see YR_COUNT_SLOTS): the gradient is multiplied by 1 / count first. */
int yr_adam_dense_flat(float *const *p, float *const *g, float *const *m, float *const *v,
                       const int64_t *n, uint8_t *const *touched, const int *row_width, const int *clear,
                       const int *scaled, const int32_t *grad_count, int count, double lr, double step_size, double bc2_sqrt, double beta1,
                       double beta2, double eps, double weight_decay, int mode, void *stream);
int yr_sgd_dense(float *p, float *g, int64_t n, double lr, double weight_decay,
                 int zero_grad, void *stream);

/* ---------------------------------------------------------------------------
 * CDAE batches on the device      (reference data/datasets/cdae_dataset.py:20-59, which builds the
 *   dense rows and the negative mask of every user on the host)
 * yr_csr_rows_to_dense: out[b, :] (float32 [B, num_items]) = 0/1 row of user users[b] from a per-user
 *   item CSR (ptr [num_users+1] / idx, int64); accumulate != 0 ORs onto the existing rows.
 * yr_negative_mask: out[b, :] = exactly neg_times * (number of positives[b, :] > 0) distinct items with
 *   positives[b, i] <= 0, every subset equally likely (np.random.choice(.., replace=False),
 *   cdae_dataset.py:20-34); i.i.d. 64-bit Philox keys per (seed, row, item), the smallest win.  A row
 *   that asks for more negatives than it has non-positives raises YR_FLAG_BAD_ITEM in err_flag (the
 *   reference raises ValueError) and takes them all.
 * ------------------------------------------------------------------------- */
int yr_csr_rows_to_dense(const int64_t *ptr, const int64_t *idx, const int64_t *users, int64_t B,
                         int64_t num_users, int64_t num_items, int accumulate, float *out,
                         int32_t *err_flag, void *stream);
int yr_negative_mask(const float *positives, int64_t B, int64_t num_items, int neg_times, uint64_t seed,
                     float *out, int32_t *err_flag, void *stream);

/* ---------------------------------------------------------------------------
 * Sparse-input form of the CDAE encoder   (reference models/cdae.py:43-49; SURVEY 2.1: the input rows are
 *   ~0.1 % dense): z = act_h(W_h . dropout_p(x) + b_h + V[u]) and dW_h = dz^T . dropout_p(x) over the
 *   non-zeros only, instead of two GEMMs over K = num_items.
 * yr_cdae_compact_rows: the non-zeros of dropout_p(x[B, I]) as (column, value) lists in ascending column
 *   order, 32 sub-lists per row (one per range of cpp = yr_cdae_sparse_part_columns(I) columns):
 *   cols / vals [B, 32, cpp], count [B, 32] — sized for the worst case, nothing can overflow.  p = 0: no
 *   dropout; p > 0: the mask yr_dropout_seeded(seed) applies, scale 1/(1-p) — no dense corrupted copy of
 *   x is written.
 * yr_cdae_sparse_encode: z[B, H] from the lists, W_h [H, I], b_h [H], V [num_users, H], user [B];
 *   act 0 identity / 1 sigmoid (bias, user-node add and activation fused).
 * yr_cdae_sparse_dwh: dWh [H, I] (all zero on entry) = dz^T . lists.  Two launches: the H-vector of every non-zero is added into
 *   the transposed scratch_T [I, H] (contiguous float atomics), then the touched columns are moved into dWh and
 *   cleared.  scratch_T: all zero on entry and on exit; claim int32[I]: any contents, never equal to a future
 *   epoch (zero-fill once; pass a different non-zero `epoch` per call); touched int32[I], n_touched int32[1]:
 *   scratch.
 * ------------------------------------------------------------------------- */
int64_t yr_cdae_sparse_part_columns(int64_t I);
int yr_cdae_compact_rows(const float *x, int64_t B, int64_t I, uint64_t seed, double p,
                         int32_t *cols, float *vals, int32_t *count, void *stream);
int yr_cdae_sparse_encode(const int32_t *cols, const float *vals, const int32_t *count,
                          const float *Wh, const float *bh, const float *V, const int64_t *user,
                          int64_t B, int64_t I, int H, int64_t num_users, int act, float *z,
                          int32_t *err_flag, void *stream);
int yr_cdae_sparse_dwh(const int32_t *cols, const float *vals, const int32_t *count,
                       const float *dz, int64_t B, int64_t I, int H, float *dWh,
                       float *scratch_T, int32_t *claim, int32_t epoch, int32_t *touched, int32_t *n_touched,
                       void *stream);
/* The same two pieces on a TRANSPOSED working copy of W_h ([I, H]; cdae_step.py keeps one during an epoch and
 * writes it back at the end): yr_cdae_sparse_encode_t reads row `col` of WhT (H contiguous floats per non-zero);
 * yr_cdae_sparse_dwh_t adds dz[b,:] * val into row `col` of dWhT (contiguous float atomics; zero on entry where
 * unmarked) and sets touched_items[col] = 1 — no scratch, one launch. */
int yr_cdae_sparse_encode_t(const int32_t *cols, const float *vals, const int32_t *count, const float *WhT,
                            const float *bh, const float *V, const int64_t *user, int64_t B, int64_t I,
                            int H, int64_t num_users, int act, float *z, int32_t *err_flag, void *stream);
int yr_cdae_sparse_dwh_t(const int32_t *cols, const float *vals, const int32_t *count, const float *dz,
                         int64_t B, int64_t I, int H, float *dWhT, uint8_t *touched_items, void *stream);
/* yr_cdae_hidden_bwd_dwh_t = yr_cdae_hidden_bwd + yr_cdae_sparse_dwh_t in one launch (a workgroup per batch row keeps
 * the row's dz in registers): db_h += dz (float atomics: db_h zero on entry), dV[user] += dz (user marked), dWhT rows
 * of the input items += dz * val (items marked), loss of the step from the decoder's partials (workgroup 0).
 * pos_count: the spread count of loss positions (always needed for the loss; divides dz when scale_dz != 0).  H <= 512. */
int yr_cdae_hidden_bwd_dwh_t(const int32_t *cols, const float *vals, const int32_t *count, const float *dz,
                             const float *z, int act, int scale_dz, const int32_t *pos_count,
                             const int64_t *user, int64_t B, int64_t I, int H, int64_t num_users, float *dV,
                             uint8_t *touched_users, float *dbh, float *dWhT, uint8_t *touched_items,
                             const float *partial_loss, int64_t n_partials, float *stats,
                             double *loss_accum, void *stream);

/* ---------------------------------------------------------------------------
 * Device-side BPR triplet stream   (reference train.py:76-77: DataLoader(MFDataset, shuffle=True);
 *   data/datasets/mf_dataset.py:18-32: __getitem__ + _negative_sampling)
 * Stream positions [first, first + count) of epoch `epoch`: position t reads row P(t) of
 * (row_user, row_item) — P a keyed pseudo-random permutation of [0, n_rows) when `shuffle`, the
 * identity otherwise — and draws neg uniformly from [0, num_items), redrawing while it is in the
 * user's avoid list (CSR avoid_ptr[num_users + 1] / avoid_idx, ascending and duplicate-free inside a
 * user; the reference avoids the train positives for train rows, train + valid positives for valid
 * rows: mf_data_pipeline.py:47-48).  A pure function of (seed, epoch, t): stateless, no host
 * synchronisation, any slice of an epoch on its own.  The generator is the engine's (Feistel + Philox),
 * not NumPy's: parity runs replay recorded streams.  user_out / pos_out / neg_out: int64[count].
 * num_items < 2^32.  A user whose avoid list covers the whole catalogue sets YR_FLAG_BAD_ITEM.
 * ------------------------------------------------------------------------- */
int yr_triplet_sample(const int64_t *row_user, const int64_t *row_item, int64_t n_rows,
                      const int64_t *avoid_ptr, const int64_t *avoid_idx,
                      int64_t num_users, int64_t num_items, uint64_t seed, uint64_t epoch, int shuffle,
                      int64_t first, int64_t count,
                      int64_t *user_out, int64_t *pos_out, int64_t *neg_out,
                      int32_t *err_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YELPREC_ENGINE_H */
