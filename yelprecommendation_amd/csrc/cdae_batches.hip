// Device-side construction of CDAE batches (the data side of reference
// data/datasets/cdae_dataset.py:20-59, which builds dense rows and negative masks per user on the host).
//
//   yr_csr_rows_to_dense   out[b, :] = 0/1 row of user users[b] from a per-user item CSR (optionally
//                          OR-ed onto what is already there: train | valid for the test-time input)
//   yr_negative_mask       per row: exactly neg_times * positives distinct NON-positive items, every
//                          subset equally likely — np.random.choice(non-positives, k, replace=False)
//                          of cdae_dataset.py:20-34.  Each non-positive item gets an i.i.d. 64-bit
//                          Philox key (regenerated from (seed, row, item) whenever needed, never
//                          stored); the k smallest keys win.  The k-th smallest key: one LDS
//                          histogram of the top 11 key bits, then the few keys of the bin that
//                          holds it are ranked in LDS (11 more bits per pass only for rows so long
//                          that a bin overflows the list).
#include "common.h"

namespace yr {

__device__ __forceinline__ uint4 cb_philox(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {                        // Philox4x32-7 (the shortest variant that passes BigCrush)
    const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

__device__ __forceinline__ uint64_t cb_key(uint64_t seed, int64_t row, int64_t item) {
  const uint4 r = cb_philox(make_uint4((uint32_t)item, (uint32_t)(item >> 32), (uint32_t)row, (uint32_t)(row >> 32)),
                            make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  return ((uint64_t)r.x << 32) | r.y;
}

// one workgroup per output row
__global__ __launch_bounds__(kBlock) void csr_rows_to_dense_kernel(const int64_t* __restrict__ ptr,
                                                                   const int64_t* __restrict__ idx,
                                                                   const int64_t* __restrict__ users, int64_t num_users,
                                                                   int64_t num_items, int accumulate,
                                                                   float* __restrict__ out, int32_t* __restrict__ err_flag) {
  float* o = out + (int64_t)blockIdx.x * num_items;
  if (!accumulate) {
    for (int64_t c = threadIdx.x; c < num_items; c += kBlock) o[c] = 0.0f;
    __syncthreads();                                   // same workgroup wrote the zeros: ordered before the ones
  }
  const int64_t u = users[blockIdx.x];
  if ((uint64_t)u >= (uint64_t)num_users) {
    if (threadIdx.x == 0 && err_flag) atomicOr(err_flag, YR_FLAG_BAD_USER);
    return;
  }
  for (int64_t j = ptr[u] + threadIdx.x; j < ptr[u + 1]; j += kBlock) {
    const int64_t it = idx[j];
    if ((uint64_t)it < (uint64_t)num_items) o[it] = 1.0f;
    else if (err_flag) atomicOr(err_flag, YR_FLAG_BAD_ITEM);
  }
}

constexpr int kNmDigitBits = 11;
constexpr int kNmBins = 1 << kNmDigitBits;

constexpr int kNmListCap = 512;                       // keys of the selected top-digit bin kept in LDS

// one workgroup of kNmThreads per row: 16 waves on the CU that owns the row (with 256 threads the
// kernel ran one wave per SIMD and every pass was a chain of exposed memory latencies)
constexpr int kNmThreads = 1024;
constexpr int kNmWaves = kNmThreads / kWave;

__global__ __launch_bounds__(kNmThreads) void negative_mask_kernel(const float* __restrict__ pos, int64_t num_items,
                                                               int neg_times, uint64_t seed, float* __restrict__ out,
                                                               int32_t* __restrict__ err_flag) {
  __shared__ int s_hist[kNmBins];
  __shared__ int s_red[kNmWaves];
  __shared__ unsigned long long s_keys[kNmListCap];
  __shared__ int s_sel_digit, s_sel_before, s_sel_count, s_nkeys;
  __shared__ unsigned long long s_thr;
  const int64_t row = blockIdx.x;
  const float* p = pos + row * num_items;
  float* o = out + row * num_items;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;

  // pass 1: positives of the row, and the histogram of the top 11 key bits of the non-positives
  for (int b = threadIdx.x; b < kNmBins; b += kNmThreads) s_hist[b] = 0;
  if (threadIdx.x == 0) s_nkeys = 0;
  __syncthreads();
  int cnt = 0;
  for (int64_t c = threadIdx.x; c < num_items; c += kNmThreads) {
    if (p[c] > 0.0f) { ++cnt; continue; }
    atomicAdd(&s_hist[(int)(cb_key(seed, row, c) >> (64 - kNmDigitBits))], 1);
  }
#pragma unroll
  for (int m = kWave / 2; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, kWave);
  if (lane == 0) s_red[wave] = cnt;
  __syncthreads();
  int positives = 0;
#pragma unroll
  for (int w = 0; w < kNmWaves; ++w) positives += s_red[w];
  int64_t need = (int64_t)neg_times * positives;
  const int64_t room = num_items - positives;
  if (need > room) {                                   // np.random.choice(replace=False) raises; flagged for the host
    if (threadIdx.x == 0 && err_flag) atomicOr(err_flag, YR_FLAG_BAD_ITEM);
    need = room;
  }
  if (need <= 0) {
    for (int64_t c = threadIdx.x; c < num_items; c += kNmThreads) o[c] = 0.0f;
    return;
  }

  // the digit (bin) that holds the need-th smallest key: wave 0 scans the histogram
  auto find_digit = [&](int64_t want) {
    if (wave == 0) {
      constexpr int PER = kNmBins / kWave;
      int local = 0;
      for (int b = 0; b < PER; ++b) local += s_hist[lane * PER + b];
      int incl = local;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const int v = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += v;
      }
      const int before_lane = incl - local;
      if (before_lane < want && want <= incl) {        // exactly one lane
        int run = before_lane;
        for (int b = 0; b < PER; ++b) {
          const int hcount = s_hist[lane * PER + b];
          if (run + hcount >= want) { s_sel_digit = lane * PER + b; s_sel_before = run; s_sel_count = hcount; break; }
          run += hcount;
        }
      }
    }
    __syncthreads();
  };
  find_digit(need);
  uint64_t prefix = (uint64_t)s_sel_digit;             // the key bits fixed so far (top bits_done bits)
  int bits_done = kNmDigitBits;
  need -= s_sel_before;
  const int in_bin = s_sel_count;
  bool by_prefix = in_bin == need;                     // every key of the bin is taken: the prefix decides
  uint64_t thr = 0;
  __syncthreads();
  if (!by_prefix && in_bin <= kNmListCap) {
    // pass 2 (the usual case: a bin holds ~I / 2048 keys): gather the bin's keys, rank them in LDS
    for (int64_t c = threadIdx.x; c < num_items; c += kNmThreads) {
      if (p[c] > 0.0f) continue;
      const uint64_t k = cb_key(seed, row, c);
      if ((k >> (64 - kNmDigitBits)) == prefix) s_keys[atomicAdd(&s_nkeys, 1)] = k;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < in_bin; t += kNmThreads) {
      const unsigned long long mine = s_keys[t];
      int rank = 0;
      for (int j = 0; j < in_bin; ++j) rank += s_keys[j] < mine;
      if (rank == (int)need - 1) s_thr = mine;         // keys are distinct (64 random bits)
    }
    __syncthreads();
    thr = s_thr;
  } else if (!by_prefix) {
    // crowded bin (rows of millions of items): keep refining the prefix, 11 bits per pass
    while (bits_done < 64) {
      const int width = 64 - bits_done < kNmDigitBits ? 64 - bits_done : kNmDigitBits;
      const int shift = 64 - bits_done - width;
      for (int b = threadIdx.x; b < kNmBins; b += kNmThreads) s_hist[b] = 0;
      __syncthreads();
      for (int64_t c = threadIdx.x; c < num_items; c += kNmThreads) {
        if (p[c] > 0.0f) continue;
        const uint64_t k = cb_key(seed, row, c);
        if ((k >> (64 - bits_done)) == prefix) atomicAdd(&s_hist[(int)((k >> shift) & ((1u << width) - 1))], 1);
      }
      __syncthreads();
      find_digit(need);
      prefix = (prefix << width) | (uint64_t)s_sel_digit;
      bits_done += width;
      need -= s_sel_before;
      const bool exact = s_sel_count == need;
      __syncthreads();
      if (exact) break;
    }
    by_prefix = true;
  }
  // final pass: a non-positive item is selected when its key is <= the threshold key, or (prefix form)
  // when its top bits_done bits are <= the prefix
  for (int64_t c = threadIdx.x; c < num_items; c += kNmThreads) {
    float v = 0.0f;
    if (!(p[c] > 0.0f)) {
      const uint64_t k = cb_key(seed, row, c);
      const bool sel = by_prefix ? (bits_done >= 64 ? k : (k >> (64 - bits_done))) <= prefix : k <= thr;
      v = sel ? 1.0f : 0.0f;
    }
    o[c] = v;
  }
}

}  // namespace yr

using namespace yr;

extern "C" int yr_csr_rows_to_dense(const int64_t* ptr, const int64_t* idx, const int64_t* users, int64_t B,
                                    int64_t num_users, int64_t num_items, int accumulate, float* out,
                                    int32_t* err_flag, void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || B > 0x7fffffff) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!ptr || !users || !out) return YR_ERR_BADARG;
  hipLaunchKernelGGL(csr_rows_to_dense_kernel, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, ptr, idx, users,
                     num_users, num_items, accumulate, out, err_flag);
  return launch_status();
}

extern "C" int yr_negative_mask(const float* positives, int64_t B, int64_t num_items, int neg_times, uint64_t seed,
                                float* out, int32_t* err_flag, void* stream) {
  if (B < 0 || num_items <= 0 || neg_times < 0 || B > 0x7fffffff) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!positives || !out || positives == out) return YR_ERR_BADARG;
  hipLaunchKernelGGL(negative_mask_kernel, dim3((unsigned)B), dim3(kNmThreads), 0, (hipStream_t)stream, positives, num_items,
                     neg_times, seed, out, err_flag);
  return launch_status();
}
