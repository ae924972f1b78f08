// Device-side BPR triplet stream for gfx950 — the producer side of the reference's
//   DataLoader(MFDataset, shuffle=True)      (train.py:76-77)
//   MFDataset.__getitem__ / _negative_sampling (data/datasets/mf_dataset.py:18-32)
// i.e. one random permutation of the rows per epoch and, for every row, a negative item drawn
// uniformly from [0, num_items) and redrawn while it is one of that user's positives.  The
// reference does this per row on the host (pandas .iloc + a Python `in list` test, ~65 us per row);
// here position t of epoch e is a pure function of (seed, e, t): no state, no host round trip, any
// slice [first, first + count) of an epoch can be produced on its own (a batch, a rank's share).
//
//   row(t)  = P(t), P a keyed bijection of [0, n_rows): a 4-round Feistel network over the smallest
//             even-width bit domain that holds n_rows, cycle-walked back into range
//   neg(t)  = first draw d = 0, 1, ... of  floor(u32 * num_items / 2^32)  (Lemire's unbiased
//             multiply-shift; u32 from Philox4x32-7 keyed by (seed, epoch), counter (t, d)) that is
//             not in the user's sorted avoid-list (binary search)
// Integer work: the CPU statement oracle/triplet_sampler.py gives the same words bit for bit.
#include "common.h"

namespace yr {

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {       // murmur3 finaliser
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// keyed bijection of [0, 2^(2 half_bits)): balanced Feistel, 4 rounds
__device__ __forceinline__ uint64_t feistel(uint64_t x, int half_bits, uint32_t k0, uint32_t k1) {
  const uint32_t mask = half_bits >= 32 ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
  uint32_t L = (uint32_t)(x >> half_bits) & mask, R = (uint32_t)x & mask;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t f = fmix32(R ^ (r & 1 ? k1 : k0) ^ (0x9E3779B9u * (uint32_t)(r + 1))) & mask;
    const uint32_t t = L ^ f;
    L = R;
    R = t;
  }
  return ((uint64_t)L << half_bits) | R;
}

__device__ __forceinline__ uint4 philox4x32_7(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

constexpr int kMaxDraws = 4096;   // after this many rejected draws the first free item >= the last draw is taken

__global__ __launch_bounds__(kBlock) void triplet_sample_kernel(
    const int64_t* __restrict__ row_user, const int64_t* __restrict__ row_item, int64_t n_rows,
    const int64_t* __restrict__ avoid_ptr, const int64_t* __restrict__ avoid_idx, int64_t num_users,
    int64_t num_items, uint64_t seed, uint64_t epoch, int shuffle, int half_bits, int64_t first, int64_t count,
    int64_t* __restrict__ user_out, int64_t* __restrict__ pos_out, int64_t* __restrict__ neg_out,
    int32_t* __restrict__ err_flag) {
  const uint32_t k0 = fmix32((uint32_t)seed ^ 0x243F6A88u) ^ (uint32_t)epoch;
  const uint32_t k1 = fmix32((uint32_t)(seed >> 32) ^ 0x85A308D3u) ^ (uint32_t)(epoch >> 32) ^ fmix32((uint32_t)epoch);
  const uint2 key = make_uint2((uint32_t)seed ^ (uint32_t)(epoch * 0x9E3779B97F4A7C15ull >> 32),
                               (uint32_t)(seed >> 32) ^ (uint32_t)epoch);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
    const int64_t t = first + i;
    uint64_t r = (uint64_t)t;
    if (shuffle) {
      do { r = feistel(r, half_bits, k0, k1); } while (r >= (uint64_t)n_rows);   // cycle walking: < 4 rounds expected
    }
    const int64_t u = row_user[r], p = row_item[r];
    int64_t neg = 0;
    if ((uint64_t)u >= (uint64_t)num_users || (uint64_t)p >= (uint64_t)num_items) {
      if (err_flag) atomicOr(err_flag, (uint64_t)u >= (uint64_t)num_users ? YR_FLAG_BAD_USER : YR_FLAG_BAD_ITEM);
      user_out[i] = u; pos_out[i] = p; neg_out[i] = 0;
      continue;
    }
    const int64_t lo0 = avoid_ptr[u], hi0 = avoid_ptr[u + 1];
    for (int d = 0;; ++d) {
      // unbiased integer in [0, num_items): Lemire's multiply-shift with rejection of the short first interval
      const uint4 w = philox4x32_7(make_uint4((uint32_t)t, (uint32_t)((uint64_t)t >> 32), (uint32_t)d, 0u), key);
      const uint32_t n32 = (uint32_t)num_items;
      uint64_t m = (uint64_t)w.x * n32;
      if ((uint32_t)m < n32) {
        const uint32_t thr = (0u - n32) % n32;
        const uint32_t alt[3] = {w.y, w.z, w.w};
        int a = 0;
        while ((uint32_t)m < thr && a < 3) m = (uint64_t)alt[a++] * n32;
      }
      neg = (int64_t)(m >> 32);
      // member of the user's avoid list (ascending)?
      int64_t lo = lo0, hi = hi0;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (avoid_idx[mid] < neg) lo = mid + 1; else hi = mid;
      }
      const bool taken = lo < hi0 && avoid_idx[lo] == neg;
      if (!taken) break;
      if (d + 1 >= kMaxDraws) {
        // a user whose avoid list covers (almost) the whole catalogue: walk to the next free item
        // (the reference would loop forever on a full list; here: flag and emit item 0)
        int64_t c = neg, steps = 0;
        while (steps < num_items) {
          c = c + 1 == num_items ? 0 : c + 1;
          ++steps;
          int64_t l2 = lo0, h2 = hi0;
          while (l2 < h2) {
            const int64_t mid = (l2 + h2) >> 1;
            if (avoid_idx[mid] < c) l2 = mid + 1; else h2 = mid;
          }
          if (!(l2 < hi0 && avoid_idx[l2] == c)) break;
        }
        if (steps >= num_items) {
          if (err_flag) atomicOr(err_flag, YR_FLAG_BAD_ITEM);
          c = 0;
        }
        neg = c;
        break;
      }
    }
    user_out[i] = u;
    pos_out[i] = p;
    neg_out[i] = neg;
  }
}

}  // namespace yr

using namespace yr;

extern "C" int yr_triplet_sample(const int64_t* row_user, const int64_t* row_item, int64_t n_rows,
                                 const int64_t* avoid_ptr, const int64_t* avoid_idx, int64_t num_users,
                                 int64_t num_items, uint64_t seed, uint64_t epoch, int shuffle, int64_t first,
                                 int64_t count, int64_t* user_out, int64_t* pos_out, int64_t* neg_out,
                                 int32_t* err_flag, void* stream) {
  if (n_rows < 0 || count < 0 || first < 0 || first + count > n_rows || num_users <= 0 || num_items <= 0)
    return YR_ERR_BADARG;
  if (num_items > 0xFFFFFFFFll) return YR_ERR_UNSUPPORTED;
  if (count == 0) return 0;
  if (!row_user || !row_item || !avoid_ptr || !avoid_idx || !user_out || !pos_out || !neg_out) return YR_ERR_BADARG;
  // smallest even-width domain 2^(2 h) >= n_rows (at least 2 bits so that both halves exist)
  int half_bits = 1;
  while (half_bits < 32 && ((uint64_t)1 << (2 * half_bits)) < (uint64_t)n_rows) ++half_bits;
  hipLaunchKernelGGL(triplet_sample_kernel, dim3(grid_for(count, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     row_user, row_item, n_rows, avoid_ptr, avoid_idx, num_users, num_items, seed, epoch, shuffle,
                     half_bits, first, count, user_out, pos_out, neg_out, err_flag);
  return launch_status();
}
