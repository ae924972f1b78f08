// CDAE training batches as LISTS, straight from the per-user item CSR (gfx950).
//
// The reference builds, per user and per fetch, a dense 0/1 input row and a dense negative mask over the whole
// catalogue (data/datasets/cdae_dataset.py:20-59: np.random.choice(non-positives, neg_times * positives,
// replace=False)); the dense route here does the same on the device (cdae_batches.hip) and the training step
// then compacts both back into lists (cdae_sparse.hip).  At Yelp2018 size that round trip through [B, I] floats
// is 40 % of an epoch (negative mask 66 us + dense rows 9 us + fills + compaction 17 us per 256-user batch against
// a 150 us step).  yr_cdae_train_lists goes from the CSR to the two lists the step consumes:
//
//   encoder list   the user's train items with nn.Dropout(p) applied: (column, 1 / (1 - p)) for the kept ones —
//                  the mask yr_dropout_seeded / yr_cdae_compact_rows would give the dense row (same Philox word per
//                  flat position b * I + column), so the dense and the list route see the same corrupted input;
//   loss list      the NS-BCE positions of the row, (column, target): its positives (1) and exactly
//                  neg_times * positives distinct non-positive items (0), every subset equally likely.  With a
//                  second CSR (validation: the held-out items, cdae_trainer.py:67 target = input + valid mask)
//                  its items are positives of the loss list too — not of the encoder list.
//
// Negatives: the items are drawn one after the other, uniformly from the catalogue, a draw that hits a positive
// or an item already taken is discarded — i.e. the first `need` distinct non-positive values of an i.i.d.
// uniform sequence, which is a uniform subset (the law of np.random.choice(replace=False)).  The sequence is
// Philox(seed; row, draw index); a round of the loop examines as many draws as are still needed (at most one
// per thread), so it can never overshoot and the set does not depend on which thread wins a duplicate.  Taken
// items are bits of a catalogue bitmap in LDS (4.7 KB at 38,048 items).  When more than half of the
// non-positives are wanted the EXCLUDED ones are drawn instead.
// Both lists come out in ascending column order in the 32-sub-list layout of cdae_sparse.hip.
#include "common.h"

namespace yr {

constexpr int kListParts = 32;                  // == cdae_sparse.hip kParts
// one workgroup of 16 waves per row: the row's work (a bitmap scan of the catalogue, 64 columns per wave step) is a
// chain of short steps, and with 4 waves it ran at one wave per SIMD (41 us per 256 rows; the rows' CUs idle otherwise)
constexpr int kListThreads = 1024;
constexpr int kListWaves = kListThreads / kWave;

__device__ __forceinline__ uint4 cl_philox(uint4 ctr, uint2 key, int rounds) {
  for (int r = 0; r < rounds; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

__device__ __forceinline__ float cl_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(kListThreads) void cdae_train_lists_kernel(
    const int64_t* __restrict__ ptr, const int64_t* __restrict__ idx, const int64_t* __restrict__ ptr2,
    const int64_t* __restrict__ idx2, const int64_t* __restrict__ users, int64_t num_users, int64_t I, int neg_times, uint64_t neg_seed, uint64_t drop_seed, float p, float scale,
    const uint64_t* __restrict__ neg_seeds, const uint64_t* __restrict__ drop_seeds, int64_t batch_rows,
    int64_t cpp, int words, int32_t* __restrict__ cols, float* __restrict__ vals, int32_t* __restrict__ count,
    int32_t* __restrict__ lcols, float* __restrict__ lvals, int32_t* __restrict__ lcount,
    int32_t* __restrict__ err_flag) {
  extern __shared__ uint32_t s_bits[];            // [words] input items, [words] drawn items, [words] all positives
  __shared__ int s_got, s_bad;
  uint32_t* s_in = s_bits;                        // the encoder's input: the items of (ptr, idx)
  uint32_t* s_drawn = s_bits + words;
  uint32_t* s_pos = ptr2 ? s_bits + 2 * words : s_bits;   // loss positives: input items + those of (ptr2, idx2)
  const int nmaps = ptr2 ? 3 : 2;
  const int64_t r = blockIdx.x;
  // Several batches in one launch (yr_cdae_train_lists_batched): row r is row rl = r % batch_rows of batch
  // r / batch_rows and takes that batch's seeds — the lists are then exactly those of one launch per batch.
  const int64_t rl = neg_seeds ? r % batch_rows : r;
  if (neg_seeds) {
    neg_seed = neg_seeds[r / batch_rows];
    drop_seed = drop_seeds[r / batch_rows];
  }
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // the row's CSR ranges are requested first and arrive while the bitmaps are cleared (they were two dependent
  // round trips in front of everything else: with two workgroups per CU nothing hides them)
  const int64_t u = users[r];
  const bool user_ok = (uint64_t)u < (uint64_t)num_users;
  int64_t lo = 0, hi = 0, lo2 = 0, hi2 = 0;
  if (user_ok) {
    lo = ptr[u]; hi = ptr[u + 1];
    if (ptr2) { lo2 = ptr2[u]; hi2 = ptr2[u + 1]; }
  }
  for (int w = tid; w < nmaps * words; w += kListThreads) s_bits[w] = 0u;
  if (tid == 0) { s_got = 0; s_bad = 0; }
  __syncthreads();
  if (!user_ok && tid == 0 && err_flag) atomicOr(err_flag, YR_FLAG_BAD_USER);
  for (int64_t j = lo + tid; j < hi; j += kListThreads) {
    const int64_t it = idx[j];
    if ((uint64_t)it < (uint64_t)I) {
      const uint32_t bit = 1u << (it & 31);
      if (atomicOr(&s_in[it >> 5], bit) & bit) s_bad = 1;       // a repeated item: the CSR is not a set
      if (ptr2) atomicOr(&s_pos[it >> 5], bit);
    } else {
      s_bad = 1;
    }
  }
  if (ptr2) {
    for (int64_t j = lo2 + tid; j < hi2; j += kListThreads) {
      const int64_t it = idx2[j];
      if ((uint64_t)it < (uint64_t)I) atomicOr(&s_pos[it >> 5], 1u << (it & 31));
      else s_bad = 1;
    }
  }
  __syncthreads();
  if (s_bad && tid == 0 && err_flag) atomicOr(err_flag, YR_FLAG_BAD_ITEM);
  // positives = bits set (repeats and out-of-range ids were flagged and do not count)
  int npos = 0;
  for (int w = tid; w < words; w += kListThreads) npos += __popc(s_pos[w]);
  __shared__ int s_red[kListWaves];
#pragma unroll
  for (int d = kWave / 2; d >= 1; d >>= 1) npos += __shfl_xor(npos, d, kWave);
  if (lane == 0) s_red[wave] = npos;
  __syncthreads();
  npos = 0;
#pragma unroll
  for (int w = 0; w < kListWaves; ++w) npos += s_red[w];

  const int64_t room = I - npos;
  int64_t need = (int64_t)neg_times * npos;
  if (need > room) {                              // np.random.choice(replace=False) raises; flagged for the host
    if (tid == 0 && err_flag) atomicOr(err_flag, YR_FLAG_BAD_ITEM);
    need = room;
  }
  const bool invert = need > room / 2;            // then the items NOT taken are drawn
  const int target = (int)(invert ? room - need : need);
  const uint2 nkey = make_uint2((uint32_t)neg_seed, (uint32_t)(neg_seed >> 32));
  const uint32_t span = (uint32_t)I;
  const uint32_t lemire_min = (0u - span) % span;  // draws whose low product word is below this are biased: skipped
  uint64_t next = 0;                              // index of the next unexamined draw of the row's sequence
  for (;;) {
    const int got = s_got;
    __syncthreads();                              // everyone has read s_got before anyone adds to it
    if (got >= target) break;
    const int k = min(target - got, kListThreads);
    if (tid < k) {
      const uint64_t d = next + tid;
      const uint4 w = cl_philox(make_uint4((uint32_t)d, (uint32_t)(d >> 32), (uint32_t)rl, (uint32_t)(rl >> 32)), nkey, 7);
      const uint64_t m = (uint64_t)w.x * span;
      if ((uint32_t)m >= lemire_min) {
        const uint32_t it = (uint32_t)(m >> 32);
        const uint32_t bit = 1u << (it & 31);
        if (!(s_pos[it >> 5] & bit) && !(atomicOr(&s_drawn[it >> 5], bit) & bit)) atomicAdd(&s_got, 1);
      }
    }
    next += k;
    __syncthreads();
  }

  // emission: wave w writes parts w, w + 16.  A lane takes one 32-column WORD of the bitmaps (64 words = 2,048 columns
  // per wave step: a part of the Yelp2018 catalogue is one step), its entries' positions are a wave prefix sum of the
  // lanes' popcounts, and it writes them bit by bit in ascending order.  (64 COLUMNS per step — 3 LDS reads, a
  // ballot and a branch for mostly empty steps — was 80 % of the kernel: 149 -> 32 us per 4,096 rows without it.)
  const uint2 dkey = make_uint2((uint32_t)drop_seed, (uint32_t)(drop_seed >> 32));
  for (int part = wave; part < kListParts; part += kListWaves) {
    const int64_t c_lo = (int64_t)part * cpp, c_hi = min(I, c_lo + cpp);
    const int64_t at0 = (r * kListParts + part) * cpp;
    int base = 0, lbase = 0;
    if (c_lo < c_hi) {
      const int w_first = (int)(c_lo >> 5), w_last = (int)((c_hi - 1) >> 5);
      for (int wb = w_first; wb <= w_last; wb += kWave) {          // wave-uniform trip count
        const int w = wb + lane;
        uint32_t in_bits = 0, pos_bits = 0, neg_bits = 0;
        if (w <= w_last) {
          const int64_t col0 = (int64_t)w << 5;
          uint32_t mask = 0xffffffffu;
          if (col0 < c_lo) mask &= 0xffffffffu << (int)(c_lo - col0);
          if (col0 + 32 > c_hi) mask &= 0xffffffffu >> (int)(col0 + 32 - c_hi);
          pos_bits = s_pos[w] & mask;
          in_bits = s_in[w] & mask;
          const uint32_t drawn = s_drawn[w];
          neg_bits = (invert ? ~drawn : drawn) & ~pos_bits & mask;
        }
        uint32_t enc_bits = in_bits;
        if (p > 0.0f) {                                             // nn.Dropout: the kept ones of the input items
          enc_bits = 0;
          for (uint32_t b = in_bits; b; b &= b - 1) {
            const int bit = __ffs((int)b) - 1;
            const int64_t e = rl * I + (((int64_t)w << 5) + bit);   // flat position of the dense batch: its Philox group and word
            const uint4 wd = cl_philox(make_uint4((uint32_t)(e >> 2), (uint32_t)((e >> 2) >> 32), 0u, 0u), dkey, 10);
            const uint32_t word = (e & 3) == 0 ? wd.x : (e & 3) == 1 ? wd.y : (e & 3) == 2 ? wd.z : wd.w;
            if (cl_u01(word) >= p) enc_bits |= 1u << bit;
          }
        }
        const uint32_t loss_bits = pos_bits | neg_bits;
        const int ne = __popc(enc_bits), nl = __popc(loss_bits);
        int ie = ne, il = nl;                                       // inclusive scans over the lanes
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const int te = __shfl_up(ie, d, kWave), tl = __shfl_up(il, d, kWave);
          if (lane >= d) { ie += te; il += tl; }
        }
        int at = base + ie - ne;
        const float kept = p > 0.0f ? scale : 1.0f;
        for (uint32_t b = enc_bits; b; b &= b - 1) {
          cols[at0 + at] = (int32_t)((w << 5) + __ffs((int)b) - 1);
          vals[at0 + at] = kept;
          ++at;
        }
        at = lbase + il - nl;
        for (uint32_t b = loss_bits; b; b &= b - 1) {
          const int bit = __ffs((int)b) - 1;
          lcols[at0 + at] = (int32_t)((w << 5) + bit);
          lvals[at0 + at] = ((pos_bits >> bit) & 1u) ? 1.0f : 0.0f;
          ++at;
        }
        base += __shfl(ie, kWave - 1, kWave);
        lbase += __shfl(il, kWave - 1, kWave);
      }
    }
    if (lane == 0) { count[r * kListParts + part] = base; lcount[r * kListParts + part] = lbase; }
  }
}

}  // namespace yr

using namespace yr;

static int train_lists_impl(const int64_t* ptr, const int64_t* idx, const int64_t* ptr2, const int64_t* idx2,
                            const int64_t* users, int64_t B, int64_t num_users, int64_t I, int neg_times,
                            uint64_t neg_seed, uint64_t drop_seed, const uint64_t* neg_seeds, const uint64_t* drop_seeds,
                            int64_t batch_rows, double p, int32_t* cols, float* vals, int32_t* count,
                            int32_t* loss_cols, float* loss_targets, int32_t* loss_count, int32_t* err_flag,
                            void* stream) {
  if (B < 0 || num_users <= 0 || I <= 0 || neg_times < 0 || p < 0.0 || p >= 1.0 || B > 0x7fffffff) return YR_ERR_BADARG;
  if (I > 0x7fffffff) return YR_ERR_UNSUPPORTED;
  if (B == 0) return 0;
  if (!ptr || !idx || !users || (ptr2 && !idx2) || !cols || !vals || !count || !loss_cols || !loss_targets || !loss_count)
    return YR_ERR_BADARG;
  const int words = (int)((I + 31) / 32);
  const size_t lds = (size_t)(ptr2 ? 3 : 2) * words * sizeof(uint32_t);
  if (lds > 60 * 1024) return YR_ERR_UNSUPPORTED;        // catalogues beyond ~245 k items: the dense route
  const int64_t cpp = ((I + kListParts - 1) / kListParts + 3) / 4 * 4;   // == yr_cdae_sparse_part_columns(I)
  hipLaunchKernelGGL(cdae_train_lists_kernel, dim3((unsigned)B), dim3(kListThreads), lds, (hipStream_t)stream, ptr, idx, ptr2,
                     idx2, users, num_users, I, neg_times, neg_seed, drop_seed, (float)p, (float)(1.0 / (1.0 - p)), neg_seeds, drop_seeds,
                     batch_rows, cpp, words,
                     cols, vals, count, loss_cols, loss_targets, loss_count, err_flag);
  return launch_status();
}

extern "C" int yr_cdae_train_lists(const int64_t* ptr, const int64_t* idx, const int64_t* ptr2, const int64_t* idx2,
                                   const int64_t* users, int64_t B,
                                   int64_t num_users, int64_t I, int neg_times, uint64_t neg_seed, uint64_t drop_seed,
                                   double p, int32_t* cols, float* vals, int32_t* count, int32_t* loss_cols,
                                   float* loss_targets, int32_t* loss_count, int32_t* err_flag, void* stream) {
  return train_lists_impl(ptr, idx, ptr2, idx2, users, B, num_users, I, neg_times, neg_seed, drop_seed, nullptr, nullptr,
                          0, p, cols, vals, count, loss_cols, loss_targets, loss_count, err_flag, stream);
}

extern "C" int yr_cdae_train_lists_batched(const int64_t* ptr, const int64_t* idx, const int64_t* ptr2,
                                           const int64_t* idx2, const int64_t* users, int64_t B, int64_t num_users,
                                           int64_t I, int neg_times, const uint64_t* neg_seeds,
                                           const uint64_t* drop_seeds, int64_t batch_rows, double p, int32_t* cols,
                                           float* vals, int32_t* count, int32_t* loss_cols, float* loss_targets,
                                           int32_t* loss_count, int32_t* err_flag, void* stream) {
  if (!neg_seeds || !drop_seeds || batch_rows <= 0) return YR_ERR_BADARG;
  return train_lists_impl(ptr, idx, ptr2, idx2, users, B, num_users, I, neg_times, 0, 0, neg_seeds, drop_seeds,
                          batch_rows, p, cols, vals, count, loss_cols, loss_targets, loss_count, err_flag, stream);
}
