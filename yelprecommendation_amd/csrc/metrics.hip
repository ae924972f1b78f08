// Ranking metrics on the device: the per-user terms of precision / recall / MAP /
// NDCG @k straight from the top-k lists and the held-out item lists (CSR), so that the Python loops
// of the reference's metric.py (:20-24, :40-45, :63-77, :97-109) leave the epoch's critical path
// (at Yelp2018 size they take ~100x longer than the fused scoring + top-k kernel).
// The host functions in yelprecommendation_amd/metric.py stay the checked definition; this kernel
// reproduces their quirks:
//   precision  |set(actual) & set(pred[:k])| / k                       over ALL users
//   recall     same hits / |set(actual)|                               users with empty actual skipped
//   AP         sum over hit positions i of |set(actual[:i]) & set(pred[:i])| / i, divided by
//              len(actual) — the ACTUAL list is truncated to i as well
//   NDCG       DCG over positions 1..min(len(actual), k) only; ideal DCG = same positions, all hits
// One thread per user (lists are tens of items); float64 sums, fixed-order reduction (wave butterfly
// per workgroup, then one wave over the workgroup partials).
#include "common.h"

namespace yr {

__device__ __forceinline__ bool contains(const int64_t* a, int n, int64_t x) {
  for (int j = 0; j < n; ++j)
    if (a[j] == x) return true;
  return false;
}

// partial[b * 5 + {0..4}] = sums over the block of precision, recall, AP, NDCG terms and the number
// of users with a non-empty actual list
constexpr int kMetricBlock = kWave;   // one wave per workgroup: ~500 workgroups at Yelp2018 size, every CU busy

__global__ __launch_bounds__(kMetricBlock) void rank_metrics_kernel(const int64_t* __restrict__ topk, int64_t n, int k,
                                                              const int64_t* __restrict__ pos_ptr,
                                                              const int64_t* __restrict__ pos_idx,
                                                              const int64_t* __restrict__ pos_rows,
                                                              double* __restrict__ partial) {
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int64_t u = (int64_t)blockIdx.x * kMetricBlock + threadIdx.x;
  if (u < n) {
    const int64_t* pred = topk + u * k;
    const int64_t pr = pos_rows ? pos_rows[u] : u;               // which CSR row holds this user's list
    const int64_t* act = pos_idx + pos_ptr[pr];
    const int len = (int)(pos_ptr[pr + 1] - pos_ptr[pr]);
    // hits = |set(actual) & set(pred[:k])| : distinct predicted items that occur in actual
    int hits = 0;
    for (int i = 0; i < k; ++i) {
      const int64_t x = pred[i];
      bool dup = false;
      for (int j = 0; j < i; ++j) dup |= pred[j] == x;
      if (!dup && contains(act, len, x)) ++hits;
    }
    v[0] = (double)hits / (double)k;
    if (len > 0) {
      int uniq = 0;                                   // |set(actual)|
      for (int j = 0; j < len; ++j) {
        bool dup = false;
        for (int q = 0; q < j; ++q) dup |= act[q] == act[j];
        if (!dup) ++uniq;
      }
      v[1] = (double)hits / (double)uniq;
      double ap = 0.0, dcg = 0.0, idcg = 0.0;
      const int span = len < k ? len : k;
      for (int i = 1; i <= k; ++i) {
        if (!contains(act, len, pred[i - 1])) continue;
        // |set(actual[:i]) & set(pred[:i])|
        const int na = len < i ? len : i;
        int c = 0;
        for (int a = 0; a < na; ++a) {
          bool dup = false;
          for (int q = 0; q < a; ++q) dup |= act[q] == act[a];
          if (!dup && contains(pred, i, act[a])) ++c;
        }
        ap += (double)c / (double)i;
        if (i <= span) dcg += 1.0 / log2((double)(i + 1));
      }
      for (int i = 1; i <= span; ++i) idcg += 1.0 / log2((double)(i + 1));
      v[2] = ap / (double)len;
      v[3] = dcg / idcg;
      v[4] = 1.0;
    }
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    double x = v[q];
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, kWave);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 5 + q] = x;
  }
}

// out[0..3] = precision, recall, MAP, NDCG;  out[4] = users with a non-empty actual list;
// out[5..8] = the four un-normalised sums, out[9] = n (what a user-sharded evaluation all-reduces)
__global__ __launch_bounds__(kWave) void rank_metrics_finalize_kernel(const double* __restrict__ partial, int nblocks,
                                                                      int64_t n, double* __restrict__ out) {
  // lane l sums blocks l, l + 64, ... in order, then a fixed butterfly: deterministic
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += kWave)
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] += partial[(int64_t)b * 5 + q];
#pragma unroll
  for (int q = 0; q < 5; ++q)
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) s[q] += __shfl_xor(s[q], m, kWave);
  if (threadIdx.x != 0) return;
  out[0] = s[0] / (double)n;
  out[1] = s[1] / s[4];
  out[2] = s[2] / s[4];
  out[3] = s[3] / s[4];
  out[4] = s[4];
  out[5] = s[0]; out[6] = s[1]; out[7] = s[2]; out[8] = s[3];
  out[9] = (double)n;
}

}  // namespace yr

using namespace yr;

extern "C" int64_t yr_rank_metrics_workspace_bytes(int64_t n) {
  if (n < 0) return YR_ERR_BADARG;
  return (int64_t)((n + kMetricBlock - 1) / kMetricBlock + 1) * 5 * (int64_t)sizeof(double);
}

extern "C" int yr_rank_metrics(const int64_t* topk, int64_t n, int k, const int64_t* pos_ptr, const int64_t* pos_idx,
                               const int64_t* pos_rows, double* workspace, double* out, void* stream) {
  if (n <= 0 || k <= 0) return YR_ERR_BADARG;
  if (!topk || !pos_ptr || !workspace || !out) return YR_ERR_BADARG;
  const int nblocks = (int)((n + kMetricBlock - 1) / kMetricBlock);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rank_metrics_kernel, dim3(nblocks), dim3(kMetricBlock), 0, s, topk, n, k, pos_ptr, pos_idx,
                     pos_rows, workspace);
  hipLaunchKernelGGL(rank_metrics_finalize_kernel, dim3(1), dim3(kWave), 0, s, workspace, nblocks, n, out);
  return launch_status();
}
