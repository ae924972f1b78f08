// Row-wise masked top-k for gfx950 (wave = 64): one 256-thread workgroup per row.
//
// Replaces, per eval user, the host sequence of the reference
//   pred[mask_items] = -3.40282e+38; np.argpartition(pred, -k)[-k:]; np.argsort(-topk)
// (reference trainers/mf_trainer.py:163-178, trainers/ngcf_trainer.py:167-182) and,
// with mask_value = 0, the CDAE form  pred * logical_not(input_mask)  followed by the
// same argpartition/argsort (trainers/cdae_trainer.py:123-144).
//
// Scheme: (1) the row's mask list (CSR) is turned into an LDS bitmap; (2) every thread
// streams its strided share of the row (coalesced) keeping a sorted top-K list in
// registers — after a short warm-up almost every element fails the `> current k-th`
// test, so the pass is bandwidth-bound; (3) K rounds of workgroup arg-max over the 256
// list heads emit the result in rank order.  Order: score descending, item id
// ascending among equal scores (the reference's order among exact ties is unspecified).
#include "common.h"

namespace yr {

struct Cand {
  float s;
  int32_t i;
};

__device__ __forceinline__ bool better(float s, int32_t i, float s2, int32_t i2) {
  return s > s2 || (s == s2 && i < i2);
}

// kTkThreads threads per row: 1024 when there are few rows (a 256-row CDAE batch with 256 threads ran
// one wave per SIMD, every element a fully exposed memory latency), 256 when the rows alone fill the chip
template <int K, int kTkThreads>
__global__ __launch_bounds__(kTkThreads) void topk_masked_kernel(const float* __restrict__ scores, int64_t ncols,
                                                             int64_t row_stride,
                                                             const int64_t* __restrict__ mask_ptr,
                                                             const int64_t* __restrict__ mask_idx,
                                                             const int64_t* __restrict__ mask_rows, float mask_value,
                                                             int k, int64_t* __restrict__ out) {
  constexpr int kTkWaves = kTkThreads / kWave;
  extern __shared__ uint32_t s_bits[];                 // ceil(ncols / 32) words
  __shared__ float s_ws[kTkWaves];
  __shared__ int32_t s_wi[kTkWaves];
  __shared__ int32_t s_win_i;

  const int64_t row = blockIdx.x;
  const float* __restrict__ x = scores + row * row_stride;
  const int nwords = (int)((ncols + 31) / 32);
  for (int w = threadIdx.x; w < nwords; w += kTkThreads) s_bits[w] = 0u;
  __syncthreads();
  if (mask_ptr) {
    const int64_t mr = mask_rows ? mask_rows[row] : row;          // which CSR row holds this row's mask list
    const int64_t lo = mask_ptr[mr], hi = mask_ptr[mr + 1];
    for (int64_t q = lo + threadIdx.x; q < hi; q += kTkThreads) {
      const int64_t c = mask_idx[q];
      if ((uint64_t)c < (uint64_t)ncols) atomicOr(&s_bits[c >> 5], 1u << (c & 31));
    }
  }
  __syncthreads();

  // per-thread sorted list (best first); slots beyond the data stay at (-inf, INT_MAX)
  float ls[K];
  int32_t li[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    ls[j] = -INFINITY;
    li[j] = 0x7fffffff;
  }
  // the row is streamed in rounds of kLoads coalesced loads per thread, all issued before the first
  // comparison (one load per iteration left every thread waiting a full memory latency per element)
  constexpr int kLoads = 8;
  for (int64_t c0 = threadIdx.x; c0 < ncols; c0 += (int64_t)kTkThreads * kLoads) {
    float v[kLoads];
#pragma unroll
    for (int q = 0; q < kLoads; ++q) {
      const int64_t c = c0 + (int64_t)q * kTkThreads;
      v[q] = c < ncols ? x[c] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kLoads; ++q) {
      const int64_t c = c0 + (int64_t)q * kTkThreads;
      if (c >= ncols) break;
      float s = v[q];
      if ((s_bits[c >> 5] >> (c & 31)) & 1u) s = mask_value;
      const int32_t ci = (int32_t)c;
      if (better(s, ci, ls[K - 1], li[K - 1])) {
        // insert: bubble the new element up from the tail (fully unrolled, registers only)
        ls[K - 1] = s;
        li[K - 1] = ci;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
          if (better(ls[j], li[j], ls[j - 1], li[j - 1])) {
            const float ts = ls[j]; ls[j] = ls[j - 1]; ls[j - 1] = ts;
            const int32_t ti = li[j]; li[j] = li[j - 1]; li[j - 1] = ti;
          }
        }
      }
    }
  }

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  for (int r = 0; r < k; ++r) {
    // workgroup arg-max over the list heads
    float bs = ls[0];
    int32_t bi = li[0];
#pragma unroll
    for (int m = kWave / 2; m >= 1; m >>= 1) {
      const float os = __shfl_xor(bs, m, kWave);
      const int32_t oi = __shfl_xor(bi, m, kWave);
      if (better(os, oi, bs, bi)) { bs = os; bi = oi; }
    }
    if (lane == 0) { s_ws[wave] = bs; s_wi[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ws = s_ws[0];
      int32_t wi = s_wi[0];
#pragma unroll
      for (int w = 1; w < kTkWaves; ++w)
        if (better(s_ws[w], s_wi[w], ws, wi)) { ws = s_ws[w]; wi = s_wi[w]; }
      s_win_i = wi;
      out[row * k + r] = wi == 0x7fffffff ? -1 : (int64_t)wi;   // fewer than k columns
    }
    __syncthreads();
    if (li[0] == s_win_i && s_win_i != 0x7fffffff) {
      // pop the head (item ids are unique within a row, so exactly one thread matches)
#pragma unroll
      for (int j = 0; j < K - 1; ++j) { ls[j] = ls[j + 1]; li[j] = li[j + 1]; }
      ls[K - 1] = -INFINITY;
      li[K - 1] = 0x7fffffff;
    }
    __syncthreads();
  }
}

}  // namespace yr

using namespace yr;

extern "C" int yr_topk_masked(const float* scores, int64_t nrows, int64_t ncols, int64_t row_stride,
                              const int64_t* mask_ptr, const int64_t* mask_idx, const int64_t* mask_rows,
                              float mask_value, int k, int64_t* out, void* stream) {
  if (nrows < 0 || ncols <= 0 || k <= 0 || k > 64 || row_stride < ncols) return YR_ERR_BADARG;
  // the mask bitmap of a row lives in dynamic LDS (ncols / 8 bytes): 64 KiB without raising the launch
  // attribute, less the static 132 bytes -> at most 2^19 - 2^11 columns (no silent HIP launch error beyond)
  if (ncols > ((int64_t)1 << 19) - 2048) return YR_ERR_UNSUPPORTED;
  if (nrows == 0) return 0;
  if (!scores || !out || (mask_ptr && !mask_idx)) return YR_ERR_BADARG;
  const size_t lds = (size_t)((ncols + 31) / 32) * sizeof(uint32_t);
  hipStream_t s = (hipStream_t)stream;
  // per-thread list size = next of {16, 32, 64} holding k
#define YR_TOPK_LAUNCH(KK, TT)                                                                               \
  hipLaunchKernelGGL((topk_masked_kernel<KK, TT>), dim3((unsigned)nrows), dim3(TT), lds, s, scores, ncols,    \
                     row_stride, mask_ptr, mask_idx, mask_rows, mask_value, k, out)
  const bool few_rows = nrows < 2048;
  if (k <= 16) { if (few_rows) YR_TOPK_LAUNCH(16, 1024); else YR_TOPK_LAUNCH(16, 256); }
  else if (k <= 32) { if (few_rows) YR_TOPK_LAUNCH(32, 1024); else YR_TOPK_LAUNCH(32, 256); }
  else { if (few_rows) YR_TOPK_LAUNCH(64, 512); else YR_TOPK_LAUNCH(64, 256); }
#undef YR_TOPK_LAUNCH
  return launch_status();
}
