// One NGCF training step enqueued from C: the whole of
//     pos, neg = model.bpr_forward(u, p, n, L); optimizer.zero_grad(); loss = BPRLoss(pos, neg); loss.backward();
//     optimizer.step(); train_loss += loss.item()            (reference trainers/ngcf_trainer.py:104-116)
// as ONE call that launches every kernel of the step back to back (the launches are the ones of csrc/ngcf.hip,
// csrc/bpr_mf.hip and csrc/optim.hip — the autograd route of models/ngcf.py issues the same kernels from Python).
// Why: the step is ~35 launches; issued one by one through Python / ctypes / autograd they cost the host 0.55-0.6 ms
// per step, which is a floor under the step whatever the GPU does (measured: whole-graph step 0.75 ms of kernels,
// batch-aware step 0.45 ms of kernels but still 0.74 ms of wall time).  From C a launch costs 2-4 us.
//
// Batch-aware propagation (see yr_ngcf_frontier_* in the header): layer k + 1 is computed on the rows S[k] the batch's
// scores need (S[K-1] = the batch's rows, S[k-1] = S[k] + neighbours) when the ESTIMATED size of S[k] is below
// subset_fraction x n, else on the whole graph; the plan depends on host-known numbers only (B, n, nnz), never on
// a size read back from the device.
#include "common.h"

namespace yr {

// loss partials (sum of softplus(-(pos - neg)), like bpr_loss_fwd_kernel) and gpos / gneg (like bpr_loss_bwd_kernel
// with gout = 1) in one launch
__global__ __launch_bounds__(kBlock) void ngcf_step_loss_kernel(const float* __restrict__ pos,
                                                                const float* __restrict__ neg, int64_t B,
                                                                float inv_batch, float* __restrict__ gpos,
                                                                float* __restrict__ gneg,
                                                                float* __restrict__ partials,
                                                                float* __restrict__ loss_out,
                                                                double* __restrict__ loss_accum) {
  __shared__ float s_red[kWavesPerBlock];
  float acc = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += stride) {
    const float x = pos[b] - neg[b];
    acc += softplus_neg(x);
    const float g = -sigmoid_neg(x) * (1.0f * inv_batch);
    gpos[b] = g;
    gneg[b] = -g;
  }
  const float total = block_sum(acc, s_red);
  if (gridDim.x == 1) {
    // small batch, one workgroup: the loss is final here (no partials, no finalize launch)
    if (threadIdx.x == 0) {
      const float v = total * inv_batch;
      if (loss_out) loss_out[0] = v;
      if (loss_accum) loss_accum[0] += (double)v;
    }
    return;
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
  if (blockIdx.x == 0)
    for (int i = gridDim.x + threadIdx.x; i < YR_LOSS_PARTIALS; i += kBlock) partials[i] = 0.0f;
}

struct WeightPtrs { const float* p[2 * YR_NGCF_MAX_LAYERS]; };

// The step's first launch: clears [zero, zero + zero_vec4) (16-byte units: the row sets' flags and counts, dW) with
// the workgroups below `zero_blocks`, and writes WT[m] = W[m]^T for the 2K weight matrices ([D, D] each) with the
// rest — one workgroup per matrix and 32 x 32 tile.  (Four memsets and a transpose launch cost 5 us each.)
__global__ __launch_bounds__(kBlock) void ngcf_step_prep_kernel(float4* __restrict__ zero, int64_t zero_vec4,
                                                                int zero_blocks, WeightPtrs W, int D,
                                                                float* __restrict__ WT) {
  if ((int)blockIdx.x < zero_blocks) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < zero_vec4; i += (int64_t)zero_blocks * kBlock)
      zero[i] = z;
    return;
  }
  __shared__ float s_t[32][33];
  const int tiles = (D + 31) / 32;
  const int t = blockIdx.x - zero_blocks;
  const int m = t / (tiles * tiles), tr = ((t / tiles) % tiles) * 32, tc = (t % tiles) * 32;
  const float* src = W.p[m];
  float* dst = WT + (int64_t)m * D * D;
  const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
  for (int y = y0; y < 32; y += kBlock / 32)
    if (tr + y < D && tc + x < D) s_t[y][x] = src[(tr + y) * D + tc + x];
  __syncthreads();
  for (int y = y0; y < 32; y += kBlock / 32)
    if (tc + y < D && tr + x < D) dst[(tc + y) * D + tr + x] = s_t[x][y];
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct StepLayout {
  // zeroed by the step's first launch: [flags of the K row sets | their counts | dW] — one contiguous range
  int64_t layers, Z, dlayers, dZ, zero_begin, flags, counts, dW, zero_end, rows, scores, WT, partials, total;
  StepLayout(int64_t n, int D, int K, int64_t B) {
    const int64_t nd = align256(n * D * 4);
    int64_t at = 0;
    layers = at;   at += (int64_t)K * nd;
    Z = at;        at += (int64_t)K * nd;
    dlayers = at;  at += (int64_t)(K + 1) * nd;
    dZ = at;       at += nd;
    zero_begin = at;
    flags = at;    at += (int64_t)K * align256(n * 4);
    counts = at;   at += align256((int64_t)K * 4);
    dW = at;       at += align256((int64_t)2 * K * D * D * 4);
    zero_end = at;
    rows = at;     at += (int64_t)K * align256(n * 4);
    scores = at;   at += 4 * align256(B * 4);
    WT = at;       at += align256((int64_t)2 * K * D * D * 4);
    partials = at; at += align256(YR_LOSS_PARTIALS * 4);
    total = at;
  }
};

}  // namespace yr

using namespace yr;

extern "C" int64_t yr_ngcf_step_workspace_bytes(int64_t n, int D, int K, int64_t B) {
  if (n <= 0 || n > 0x7fffffff || B < 0 || K < 0 || K >= YR_NGCF_MAX_LAYERS) return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  return StepLayout(n, D, K, B).total;
}

#define YR_TRY(call)            \
  do {                          \
    const int rc_ = (call);     \
    if (rc_ != 0) return rc_;   \
  } while (0)

extern "C" int yr_ngcf_bpr_step(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n, int64_t nnz,
                                const int32_t* heavy_rows, int64_t n_heavy, int heavy_threshold,
                                int64_t num_users, float* const* params, float* const* exp_avg,
                                float* const* exp_avg_sq, int K, int D, const int64_t* user, const int64_t* pos,
                                const int64_t* neg, int64_t B, double lr, double step_size, double bc2_sqrt,
                                double beta1, double beta2, double eps, double weight_decay, int mode,
                                double subset_fraction, void* workspace, int64_t workspace_bytes, float* loss_out,
                                double* loss_accum, int32_t* err_flag, void* stream) {
  if (n <= 0 || n > 0x7fffffff || nnz < 0 || num_users <= 0 || num_users >= n || B < 0 || K < 0 ||
      K >= YR_NGCF_MAX_LAYERS)
    return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  if (!rowptr || !col || !val || !params || !exp_avg || !exp_avg_sq || !workspace) return YR_ERR_BADARG;
  for (int t = 0; t < 1 + 2 * K; ++t)
    if (!params[t] || !exp_avg[t] || !exp_avg_sq[t]) return YR_ERR_BADARG;
  if (B > 0 && (!user || !pos || !neg)) return YR_ERR_BADARG;
  const StepLayout lay(n, D, K, B);
  if (workspace_bytes < lay.total || ((uintptr_t)workspace & 255)) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int64_t nd = align256(n * D * 4);
  float* E0 = params[0];
  float* const* W1 = params + 1;
  float* const* W2 = params + 1 + K;
  auto layer = [&](int k) { return k == 0 ? E0 : (float*)(ws + lay.layers + (int64_t)(k - 1) * nd); };   // E_k
  auto Zk = [&](int k) { return (float*)(ws + lay.Z + (int64_t)k * nd); };              // Z of layer k + 1
  auto dlayer = [&](int k) { return (float*)(ws + lay.dlayers + (int64_t)k * nd); };
  float* dZ = (float*)(ws + lay.dZ);
  auto flags = [&](int k) { return (int32_t*)(ws + lay.flags + (int64_t)k * align256(n * 4)); };
  auto rows = [&](int k) { return (int32_t*)(ws + lay.rows + (int64_t)k * align256(n * 4)); };
  int32_t* counts = (int32_t*)(ws + lay.counts);
  float* s_pos = (float*)(ws + lay.scores);
  float* s_neg = (float*)(ws + lay.scores + align256(B * 4));
  float* g_pos = (float*)(ws + lay.scores + 2 * align256(B * 4));
  float* g_neg = (float*)(ws + lay.scores + 3 * align256(B * 4));
  float* dW = (float*)(ws + lay.dW);
  float* WT = (float*)(ws + lay.WT);
  float* partials = (float*)(ws + lay.partials);
  const int64_t dd = (int64_t)D * D;

  // ---- plan: which layers run on the batch's rows
  bool sub[YR_NGCF_MAX_LAYERS] = {};
  int64_t max_rows[YR_NGCF_MAX_LAYERS] = {};
  double est_rows[YR_NGCF_MAX_LAYERS] = {};
  const double hop = 1.0 + (double)nnz / (double)n;
  constexpr double kPushMaxPairs = 50000.0;      // scatter form of the backward product below this many (row, neighbour) pairs
  if (subset_fraction > 0.0) {
    int64_t est = 3 * B < n ? 3 * B : n;
    for (int k = K - 1; k >= 0; --k) {
      if ((double)est > subset_fraction * (double)n) break;
      sub[k] = true;
      est_rows[k] = (double)est;
      max_rows[k] = k == K - 1 ? est : n;
      const double next = (double)est * hop;
      est = next < (double)n ? (int64_t)next : n;
    }
  }
  // first launch: clear the row sets' flags / counts and dW, transpose the weights for the data-gradient products
  {
    WeightPtrs wp{};
    for (int t = 0; t < 2 * K; ++t) wp.p[t] = params[1 + t];
    const int64_t vec4 = (lay.zero_end - lay.zero_begin) / 16;
    const int zero_blocks = grid_for(vec4, kBlock * 4);
    const int tiles = (D + 31) / 32;
    hipLaunchKernelGGL(ngcf_step_prep_kernel, dim3(zero_blocks + 2 * K * tiles * tiles), dim3(kBlock), 0, s,
                       (float4*)(ws + lay.zero_begin), vec4, zero_blocks, wp, D, WT);
  }
  for (int k = K - 1; k >= 0 && sub[k]; --k) {
    if (k == K - 1)
      YR_TRY(yr_ngcf_frontier_mark(user, pos, neg, B, num_users, n - num_users, flags(k), rows(k), counts + k, 0, stream));
    else
      YR_TRY(yr_ngcf_frontier_expand(rowptr, col, n, rows(k + 1), counts + k + 1, max_rows[k + 1], flags(k), rows(k),
                                     counts + k, 0, stream));
  }

  // ---- forward: K propagation layers, layer-sum scores, loss
  for (int k = 0; k < K; ++k) {
    if (sub[k]) {
      YR_TRY(yr_spmm_csr_subset(rowptr, col, val, layer(k), Zk(k), n, D, 0, heavy_rows, n_heavy, heavy_threshold,
                                flags(k), rows(k), counts + k, max_rows[k], stream));
      // (the rows of this layer's gradient buffer that the backward pass will touch are cleared by the same launch)
      YR_TRY(yr_ngcf_dense_fwd_rows(layer(k), Zk(k), W1[k], W2[k], n, D, layer(k + 1), rows(k), counts + k,
                                    max_rows[k], dlayer(k + 1), stream));
    } else {
      YR_TRY(yr_spmm_csr(rowptr, col, val, layer(k), Zk(k), n, D, 0, heavy_rows, n_heavy, heavy_threshold, stream));
      YR_TRY(yr_ngcf_dense_fwd(layer(k), Zk(k), W1[k], W2[k], n, D, layer(k + 1), stream));
    }
  }
  const float* layer_ptrs[YR_NGCF_MAX_LAYERS];
  float* dlayer_ptrs[YR_NGCF_MAX_LAYERS];
  for (int k = 0; k <= K; ++k) {
    layer_ptrs[k] = layer(k);
    dlayer_ptrs[k] = dlayer(k);
  }
  const float inv_batch = B > 0 ? 1.0f / (float)B : 0.0f;
  if (B > 0) {
    YR_TRY(yr_ngcf_score_fwd(layer_ptrs, K + 1, user, pos, neg, B, D, num_users, n - num_users, s_pos, s_neg,
                             err_flag, stream));
    const int loss_grid = B <= 8192 ? 1 : grid_for(B, kBlock);       // one workgroup finishes the loss itself
    hipLaunchKernelGGL(ngcf_step_loss_kernel, dim3(loss_grid), dim3(kBlock), 0, s, s_pos, s_neg, B, inv_batch, g_pos,
                       g_neg, partials, loss_out, loss_accum);
    if (loss_grid > 1) YR_TRY(yr_loss_finalize(partials, inv_batch, loss_out, loss_accum, stream));
  } else {
    if (hipMemsetAsync(partials, 0, YR_LOSS_PARTIALS * 4, s) != hipSuccess) return (int)hipGetLastError();
    YR_TRY(yr_loss_finalize(partials, inv_batch, loss_out, loss_accum, stream));
  }

  // ---- backward (optimizer.zero_grad() = the two clears)
  // gradient buffers: whole for the layers propagated on the whole graph; a restricted layer's rows were cleared
  // by its forward launch
  int first_sub = K;
  for (int k = 0; k < K; ++k)
    if (sub[k]) { first_sub = k; break; }
  if (hipMemsetAsync(dlayer(0), 0, (size_t)((first_sub + 1) * nd), s) != hipSuccess) return (int)hipGetLastError();
  if (B > 0)
    YR_TRY(yr_ngcf_score_bwd(layer_ptrs, dlayer_ptrs, K + 1, user, pos, neg, g_pos, g_neg, B, D, num_users,
                             n - num_users, err_flag, stream));
  for (int k = K - 1; k >= 0; --k) {
    float* dW1 = dW + (int64_t)k * dd;
    float* dW2 = dW + (int64_t)(K + k) * dd;
    const float* W1T = WT + (int64_t)k * dd;
    const float* W2T = WT + (int64_t)(K + k) * dd;
    if (sub[k]) {
      // dZ is written on the layer's rows only.  Few rows: they scatter into their neighbours (cost = their
      // non-zeros).  Many: dZ is cleared first and the pull product adds zeros elsewhere.
      const bool push = est_rows[k] * hop <= kPushMaxPairs;
      if (!push && hipMemsetAsync(dZ, 0, (size_t)(n * D * 4), s) != hipSuccess) return (int)hipGetLastError();
      YR_TRY(yr_ngcf_dense_bwd_weight_rows(dlayer(k + 1), layer(k + 1), layer(k), Zk(k), n, D, dW1, dW2, rows(k),
                                           counts + k, max_rows[k], stream));
      YR_TRY(yr_ngcf_dense_bwd_data_rows(dlayer(k + 1), layer(k + 1), layer(k), Zk(k), W1T, W2T, n, D, dZ, dlayer(k),
                                         rows(k), counts + k, max_rows[k], stream));
      if (push)
        YR_TRY(yr_spmm_csr_push_rows(rowptr, col, val, dZ, dlayer(k), n, D, rows(k), counts + k, max_rows[k], stream));
      else   // every neighbour of this layer's rows lies in the set of the layer below: only those rows receive
      {
        const bool below = k > 0 && sub[k - 1];
        YR_TRY(yr_spmm_csr_subset(rowptr, col, val, dZ, dlayer(k), n, D, 1, heavy_rows, n_heavy, heavy_threshold,
                                  below ? flags(k - 1) : nullptr, below ? rows(k - 1) : nullptr,
                                  below ? counts + k - 1 : nullptr, below ? max_rows[k - 1] : 0, stream));
      }
    } else {
      YR_TRY(yr_ngcf_dense_bwd_weight(dlayer(k + 1), layer(k + 1), layer(k), Zk(k), n, D, dW1, dW2, stream));
      YR_TRY(yr_ngcf_dense_bwd_data(dlayer(k + 1), layer(k + 1), layer(k), Zk(k), W1T, W2T, n, D, dZ, dlayer(k), stream));
      YR_TRY(yr_spmm_csr(rowptr, col, val, dZ, dlayer(k), n, D, 1, heavy_rows, n_heavy, heavy_threshold, stream));
    }
  }

  // ---- optimizer.step(): dense Adam on the embedding table, one multi-tensor launch for the 2K weight matrices
  YR_TRY(yr_adam_dense(E0, dlayer(0), exp_avg[0], exp_avg_sq[0], n * D, lr, step_size, bc2_sqrt, beta1, beta2, eps,
                       weight_decay, mode, 0, stream));
  if (K > 0) {
    float* g[2 * YR_NGCF_MAX_LAYERS];
    int64_t cnt[2 * YR_NGCF_MAX_LAYERS];
    for (int t = 0; t < 2 * K; ++t) {
      g[t] = dW + (int64_t)t * dd;
      cnt[t] = dd;
    }
    YR_TRY(yr_adam_dense_multi(params + 1, g, exp_avg + 1, exp_avg_sq + 1, cnt, 2 * K, lr, step_size, bc2_sqrt, beta1,
                               beta2, eps, weight_decay, mode, 0, stream));
  }
  return launch_status();
}
