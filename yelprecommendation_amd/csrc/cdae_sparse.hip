// Sparse-input form of the CDAE encoder for gfx950 (the input rows are 0/1 interaction rows,
// ~0.1 % dense, and dropout thins them further) — reference models/cdae.py:43-49:
//     z = act_h( W_h . dropout_p(x) + b_h + V[u] )          and, in backward,  dW_h = dz^T . dropout_p(x)
// The dense form runs both as f32 MFMA GEMMs over K = I = 38,048 (2.5 GFLOP and ~60 MB each for ~5 k
// non-zeros per batch).  Here:
//   yr_cdae_compact_rows   one wave per (row, 1/32 of the columns): the non-zeros of dropout_p(x[r, :]) as
//                          (column, value) lists in ascending column order — the dropout mask is the one yr_dropout_seeded would
//                          apply (same Philox words per flat position), so no dense corrupted copy of x is
//                          ever written;
//   yr_cdae_sparse_encode  z[r, h] = act(b_h[h] + V[u_r, h] + sum_j val_j W_h[h, col_j])   (bias + user-node
//                          add and the activation fused: no hidden_init / sigmoid launches);
//   yr_cdae_sparse_dwh     dW_h[h, col_j] += dz[r, h] val_j                                 (float atomics into
//                          the zeroed gradient: a few thousand columns are touched per batch).
#include "common.h"

namespace yr {

__device__ __forceinline__ uint4 cs_philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

__device__ __forceinline__ float cs_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// Row r of x[B, I] is cut into kParts column ranges of `cpp` columns (a multiple of 4); part q's non-zeros go to
// cols/vals[(r * kParts + q) * cpp ..] in ascending column order, count[r * kParts + q].  One WAVE per (row,
// part): 1 KB coalesced loads, wave prefix with shuffles, no barriers (a workgroup per row with block-wide
// prefixes took 60 us at I = 38,048: 38 dependent load -> scan -> store rounds per row).
// p > 0: nn.Dropout(p) in training mode with the mask of yr_dropout_seeded(seed) — uniform word k of
// Philox(counter = flat index / 4) for flat index 4 * (flat / 4) + k.
constexpr int kParts = 32;

// PAIR: the same pass also lists the LOSS positions of the row, target + negative_mask != 0, as (column, target)
// (see the sampled decoder below).  The next 1 KB of the row(s) is fetched while the current one is scanned.
template <bool PAIR>
__global__ __launch_bounds__(kBlock) void cdae_compact_rows_kernel(
    const float* __restrict__ x, const float* __restrict__ negmask, int64_t I, uint64_t seed, float p, float scale,
    int64_t cpp, int32_t* __restrict__ cols, float* __restrict__ vals, int32_t* __restrict__ count,
    int32_t* __restrict__ lcols, float* __restrict__ lvals, int32_t* __restrict__ lcount) {
  const int64_t r = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int part = blockIdx.y * kWavesPerBlock + wave;
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const float* row = x + r * I;
  const float* nrow = PAIR ? negmask + r * I : nullptr;
  const int64_t c_lo = (int64_t)part * cpp, c_hi = min(I, c_lo + cpp);
  const int64_t at0 = (r * kParts + part) * cpp;
  const bool vec = (I & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | (PAIR ? reinterpret_cast<uintptr_t>(negmask) : 0)) & 15) == 0;
  auto fetch = [&](int64_t c, float (&t)[4], float (&m)[4]) {
    if (vec && c + 3 < c_hi) {
      const float4 a = *reinterpret_cast<const float4*>(row + c);
      t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w;
      if (PAIR) {
        const float4 b = *reinterpret_cast<const float4*>(nrow + c);
        m[0] = b.x; m[1] = b.y; m[2] = b.z; m[3] = b.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t[k] = c + k < c_hi ? row[c + k] : 0.0f;
        if (PAIR) m[k] = c + k < c_hi ? nrow[c + k] : 0.0f;
      }
    }
  };
  float tn[4], mn[4] = {0.f, 0.f, 0.f, 0.f};
  fetch(c_lo + lane * 4, tn, mn);
  int base = 0, lbase = 0;
  for (int64_t c0 = c_lo; c0 < c_hi; c0 += kWave * 4) {
    const int64_t c = c0 + lane * 4;
    float t[4], m[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { t[k] = tn[k]; m[k] = mn[k]; v[k] = tn[k]; }
    if (c0 + kWave * 4 < c_hi) fetch(c + kWave * 4, tn, mn);
    if (p > 0.0f && (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f || v[3] != 0.f)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (v[k] == 0.f) continue;
        const int64_t e = r * I + c + k;            // flat position: its Philox group and word
        const uint4 w = cs_philox4x32_10(make_uint4((uint32_t)(e >> 2), (uint32_t)((e >> 2) >> 32), 0u, 0u), key);
        const uint32_t word = (e & 3) == 0 ? w.x : (e & 3) == 1 ? w.y : (e & 3) == 2 ? w.z : w.w;
        v[k] = cs_u01(word) >= p ? v[k] * scale : 0.0f;
      }
    }
    bool sel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sel[k] = PAIR && t[k] + m[k] != 0.0f;
    const int mine = (v[0] != 0.f) + (v[1] != 0.f) + (v[2] != 0.f) + (v[3] != 0.f);
    const int lmine = (int)sel[0] + (int)sel[1] + (int)sel[2] + (int)sel[3];
    int inc = mine | (lmine << 16);                    // PAIR: both prefix sums in one scan (each < 2^15 per part)
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int tt = __shfl_up(inc, d, kWave);
      if (lane >= d) inc += tt;
    }
    int at = base + (PAIR ? inc & 0xffff : inc) - mine;
    int lat = lbase + (inc >> 16) - lmine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (v[k] != 0.f) { cols[at0 + at] = (int32_t)(c + k); vals[at0 + at] = v[k]; ++at; }
      if (PAIR && sel[k]) { lcols[at0 + lat] = (int32_t)(c + k); lvals[at0 + lat] = t[k]; ++lat; }
    }
    const int tot = __shfl(inc, kWave - 1, kWave);
    base += PAIR ? tot & 0xffff : tot;
    lbase += tot >> 16;
  }
  if (lane == 0) {
    count[r * kParts + part] = base;
    if (PAIR) lcount[r * kParts + part] = lbase;
  }
}

// The (column, value) list of one row, gathered from its kParts sub-lists into LDS in column order, kListCap
// entries at a time.
constexpr int kListCap = 2048;

__device__ __forceinline__ int gather_row_list(const int32_t* __restrict__ cols, const float* __restrict__ vals,
                                               const int32_t* __restrict__ count, int64_t cpp, int64_t r, int skip,
                                               int* s_pre, int32_t* s_col, float* s_val) {
  // s_pre[q] = number of entries before part q; returns the number of entries staged (<= kListCap),
  // starting at overall entry `skip`
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int q = 0; q < kParts; ++q) { s_pre[q] = acc; acc += count[r * kParts + q]; }
    s_pre[kParts] = acc;
  }
  __syncthreads();
  const int total = s_pre[kParts];
  const int n = min(kListCap, total - skip);
  for (int i = threadIdx.x; i < n; i += kBlock) {
    const int e = skip + i;
    int q = 0;
#pragma unroll
    for (int s = kParts / 2; s >= 1; s >>= 1)
      if (s_pre[q + s] <= e) q += s;
    const int64_t at = (r * kParts + q) * cpp + (e - s_pre[q]);
    s_col[i] = cols[at];
    s_val[i] = vals[at];
  }
  __syncthreads();
  return n;
}

// one workgroup per row, one thread per hidden unit (looped when H > 256).  WT: W_h is given TRANSPOSED, [I, H]
// (the working copy cdae_step.py keeps during an epoch): the 'column' of an entry is then H contiguous floats —
// one 512-byte line per entry and workgroup instead of one line per entry and THREAD (44 MB of fetches for ~3,000
// entries per batch in the [H, I] layout).
template <bool WT>
__global__ __launch_bounds__(kBlock) void cdae_sparse_encode_kernel(
    const int32_t* __restrict__ cols, const float* __restrict__ vals, const int32_t* __restrict__ count, int64_t cpp,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ V,
    const int64_t* __restrict__ user, int64_t I, int H, int64_t num_users, int act, float* __restrict__ z,
    int32_t* __restrict__ err_flag) {
  __shared__ int s_pre[kParts + 1];
  __shared__ int32_t s_col[kListCap];
  __shared__ float s_val[kListCap];
  const int64_t r = blockIdx.x;
  const int64_t u = user[r];
  const bool ok = (uint64_t)u < (uint64_t)num_users;
  if (!ok && threadIdx.x == 0 && err_flag) atomicOr(err_flag, YR_FLAG_BAD_USER);
  for (int h0 = 0; h0 < H; h0 += kBlock) {           // H <= 256: one round
    const int h = h0 + threadIdx.x;
    const int hc = min(h, H - 1);
    const float* wrow = WT ? Wh + hc : Wh + (int64_t)hc * I;      // element of column c: wrow[c * pitch]
    const int64_t pitch = WT ? H : 1;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int skip = 0;; skip += kListCap) {
      const int n = gather_row_list(cols, vals, count, cpp, r, skip, s_pre, s_col, s_val);
      int j = 0;
      for (; j + 4 <= n; j += 4) {
        acc0 = fmaf(s_val[j], wrow[s_col[j] * pitch], acc0);
        acc1 = fmaf(s_val[j + 1], wrow[s_col[j + 1] * pitch], acc1);
        acc2 = fmaf(s_val[j + 2], wrow[s_col[j + 2] * pitch], acc2);
        acc3 = fmaf(s_val[j + 3], wrow[s_col[j + 3] * pitch], acc3);
      }
      for (; j < n; ++j) acc0 = fmaf(s_val[j], wrow[s_col[j] * pitch], acc0);
      const bool more = skip + n < s_pre[kParts];
      __syncthreads();                               // the staged list is overwritten by the next round
      if (!more) break;
    }
    if (h < H) {
      float pre = (acc0 + acc1) + (acc2 + acc3) + bh[h];
      if (ok) pre += V[u * H + h];
      z[r * H + h] = act == 1 ? 1.0f / (1.0f + expf(-pre)) : pre;
    }
  }
}

// dW_h in two launches.  Adding dz[r, h] * val straight into dWh[h, col] (row pitch I) makes every lane of a
// wave hit a different cache line (35 us of float atomics for ~5 k non-zeros at H = 128).  Instead:
//   1  add the H-vector of every non-zero into a TRANSPOSED scratch T[col, 0..H) — 4 H contiguous bytes per
//      non-zero, the shape the memory-side float atomics run at full rate for; the first workgroup to reach a
//      column (atomic exchange on claim[col], stamped with the launch's epoch) puts the column on a list;
//   2  one wave per listed column: dWh[h, col] = T[col, h] (plain strided stores; dWh is zero on entry), T[col, :] = 0.
// T stays all-zero between steps; claim never needs clearing (the epoch changes every step).
__global__ __launch_bounds__(kBlock) void cdae_sparse_dwh_accumulate_kernel(
    const int32_t* __restrict__ cols, const float* __restrict__ vals, const int32_t* __restrict__ count, int64_t cpp,
    const float* __restrict__ dz, int H, float* __restrict__ T, int32_t* __restrict__ claim, int32_t epoch,
    int32_t* __restrict__ touched, int32_t* __restrict__ n_touched) {
  __shared__ int s_pre[kParts + 1];
  __shared__ int32_t s_col[kListCap];
  __shared__ float s_val[kListCap];
  const int64_t r = blockIdx.x;
  for (int skip = 0;; skip += kListCap) {
    const int n = gather_row_list(cols, vals, count, cpp, r, skip, s_pre, s_col, s_val);
    for (int j = threadIdx.x; j < n; j += kBlock)                 // claim the columns of this round
      if (atomicExch(&claim[s_col[j]], epoch) != epoch) touched[atomicAdd(n_touched, 1)] = s_col[j];
    for (int h = threadIdx.x; h < H; h += kBlock) {
      const float g = dz[r * H + h];
      for (int j = 0; j < n; ++j) atomicAdd(T + (int64_t)s_col[j] * H + h, g * s_val[j]);
    }
    const bool more = skip + n < s_pre[kParts];
    __syncthreads();
    if (!more) break;
  }
}

__global__ __launch_bounds__(kBlock) void cdae_sparse_dwh_scatter_kernel(const int32_t* __restrict__ touched,
                                                                         const int32_t* __restrict__ n_touched,
                                                                         float* __restrict__ T, int64_t I, int H,
                                                                         float* __restrict__ dWh) {
  const int n = n_touched[0];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  for (int i = blockIdx.x * kWavesPerBlock + wave; i < n; i += gridDim.x * kWavesPerBlock) {   // a wave per column
    const int64_t col = touched[i];
    for (int h = lane; h < H; h += kWave) {
      float* t = T + col * H + h;
      dWh[(int64_t)h * I + col] = *t;                 // dWh is zero on entry: a store, not a strided read-modify-write
      *t = 0.0f;
    }
  }
}

// dW_h when the gradient is kept transposed ([I, H], beside the transposed working copy of W_h): the H-vector of
// every non-zero goes straight into row `col` with contiguous float atomics and the row is marked for the Adam
// launch (which reads and clears marked rows only) — no scratch, no claim words, no second launch.
__global__ __launch_bounds__(kBlock) void cdae_sparse_dwh_t_kernel(
    const int32_t* __restrict__ cols, const float* __restrict__ vals, const int32_t* __restrict__ count, int64_t cpp,
    const float* __restrict__ dz, int H, float* __restrict__ dWhT, uint8_t* __restrict__ touched_items) {
  __shared__ int s_pre[kParts + 1];
  __shared__ int32_t s_col[kListCap];
  __shared__ float s_val[kListCap];
  const int64_t r = blockIdx.x;
  for (int skip = 0;; skip += kListCap) {
    const int n = gather_row_list(cols, vals, count, cpp, r, skip, s_pre, s_col, s_val);
    for (int j = threadIdx.x; j < n; j += kBlock) touched_items[s_col[j]] = 1;
    for (int h = threadIdx.x; h < H; h += kBlock) {
      const float g = dz[r * H + h];
      for (int j = 0; j < n; ++j) atomicAdd(dWhT + (int64_t)s_col[j] * H + h, g * s_val[j]);
    }
    const bool more = skip + n < s_pre[kParts];
    __syncthreads();
    if (!more) break;
  }
}

// The hidden layer's backward and dW_h in ONE launch (transposed working copy): workgroup r owns batch row r —
//   dz[r, :] <- dz[r, :] (/ count) * act'(z[r, :])        (kept in registers; written back only if dz_out != NULL)
//   db_h     += dz[r, :]                                  (float atomics; db_h zero on entry)
//   dV[u_r]  += dz[r, :], user u_r marked
//   dW_h^T[col_j, :] += dz[r, :] * val_j  for the row's input entries, items marked
// and workgroup 0 also turns the decoder's loss partials into the step's loss (fixed order).  Replaces
// yr_cdae_hidden_bwd + yr_cdae_sparse_dwh_t: one launch and one round trip through dz less (10 + 5 -> 6 us).
constexpr int kHbRows = 1;      // batch rows per workgroup (4, with their db_h contributions summed before the atomic,
                                // was slower: 12.6 against 10.3 us — the rows' list passes then run one after the other)
__global__ __launch_bounds__(kBlock) void cdae_hidden_bwd_dwh_t_kernel(
    const int32_t* __restrict__ cols, const float* __restrict__ vals, const int32_t* __restrict__ count, int64_t cpp,
    const float* __restrict__ dz, const float* __restrict__ z, int act, int scale_dz,
    const int32_t* __restrict__ pos_count, const int64_t* __restrict__ user, int64_t B, int64_t num_users, int H,
    float* __restrict__ dV, uint8_t* __restrict__ touched_users, float* __restrict__ dbh, float* __restrict__ dWhT,
    uint8_t* __restrict__ touched_items, const float* __restrict__ partial_loss, int64_t n_partials,
    float* __restrict__ stats, double* __restrict__ loss_accum) {
  __shared__ int s_pre[kParts + 1];
  __shared__ int32_t s_col[kListCap];
  __shared__ float s_val[kListCap];
  __shared__ float s_red[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  const int32_t npos = spread_count(pos_count, lane);             // every lane of every wave takes part
  const float alpha = scale_dz ? (npos > 0 ? 1.0f / (float)npos : 0.0f) : 1.0f;
  // H <= kBlock in practice; a thread keeps the gradient of its hidden units (up to 2 per thread: H <= 512)
  const int h0 = threadIdx.x, h1 = threadIdx.x + kBlock;
  const int64_t r_lo = (int64_t)blockIdx.x * kHbRows;
  float g0[kHbRows], g1[kHbRows];
  int64_t us[kHbRows];
  float b0 = 0.0f, b1 = 0.0f;
#pragma unroll
  for (int q = 0; q < kHbRows; ++q) {                             // all loads of the workgroup's rows first
    const int64_t r = min(r_lo + q, B - 1);
    const bool live = r_lo + q < B;
    us[q] = user[r];
    g0[q] = (live && h0 < H) ? dz[r * H + h0] : 0.0f;
    g1[q] = (live && h1 < H) ? dz[r * H + h1] : 0.0f;
    const float y0 = (live && h0 < H) ? z[r * H + h0] : 0.0f;
    const float y1 = (live && h1 < H) ? z[r * H + h1] : 0.0f;
    g0[q] *= alpha;
    g1[q] *= alpha;
    if (act == 1) { g0[q] *= y0 * (1.0f - y0); g1[q] *= y1 * (1.0f - y1); }
    b0 += g0[q];
    b1 += g1[q];
  }
  if (h0 < H) atomicAdd(dbh + h0, b0);
  if (h1 < H) atomicAdd(dbh + h1, b1);
#pragma unroll 1
  for (int q = 0; q < kHbRows; ++q) {
    const int64_t r = r_lo + q;
    if (r >= B) break;                                            // workgroup-uniform
    const int64_t u = us[q];
    const bool ok = (uint64_t)u < (uint64_t)num_users;
    if (ok && threadIdx.x == 0 && touched_users) touched_users[u] = 1;
    if (ok && h0 < H) atomicAdd(dV + u * H + h0, g0[q]);
    if (ok && h1 < H) atomicAdd(dV + u * H + h1, g1[q]);
    for (int skip = 0;; skip += kListCap) {
      const int n = gather_row_list(cols, vals, count, cpp, r, skip, s_pre, s_col, s_val);
      for (int j = threadIdx.x; j < n; j += kBlock) touched_items[s_col[j]] = 1;
      if (h0 < H)
        for (int j = 0; j < n; ++j) atomicAdd(dWhT + (int64_t)s_col[j] * H + h0, g0[q] * s_val[j]);
      if (h1 < H)
        for (int j = 0; j < n; ++j) atomicAdd(dWhT + (int64_t)s_col[j] * H + h1, g1[q] * s_val[j]);
      const bool more = skip + n < s_pre[kParts];
      __syncthreads();
      if (!more) break;
    }
  }
  if (blockIdx.x == 0 && n_partials > 0) {
    float sacc = 0.0f;
    for (int64_t k = threadIdx.x; k < n_partials; k += kBlock) sacc += partial_loss[k];
    const float tot = block_sum(sacc, s_red);
    if (threadIdx.x == 0) {
      const float mean = npos > 0 ? tot / (float)npos : 0.0f;
      stats[0] = mean;
      stats[1] = (float)npos;
      if (loss_accum) loss_accum[0] += (double)mean;
    }
  }
}

// ---- the decoder of a TRAINING step on the loss positions only ----
// NSBCELoss (loss.py:12-16) reads the prediction at the positions where target + negative_mask != 0 and nowhere
// else: ~(1 + neg_times) x the positives of a row, a fraction of a percent of the catalogue.  The gradient w.r.t.
// the prediction is exactly zero elsewhere, so the three dense decoder products of a training step
// (z W_o^T, G^T z, G W_o: 2.5 GFLOP each at full size) reduce to one pass over the position lists.
//
// yr_cdae_compact_pair: ONE pass over x and the negative mask makes both lists of a row — the non-zeros of
//   dropout_p(x) (the encoder's input, as yr_cdae_compact_rows) and the loss positions (column, target):
//   cdae_compact_rows_kernel<true> above.
// yr_cdae_sampled_decode: one workgroup per (row, 1 / S of its position list); a HALF-wave per position: lane l
//   holds floats [4 l, 4 l + 4) of the W_o row (one 512-byte gather per position at H = 128), the dot with z
//   is a 32-lane DPP sum; then, with the same W_o row still in registers,
//     y = act(z . W_o[i] + b_o[i]);  BCE term -> the workgroup's loss partial (fixed order);
//     g = (y - t) / max((1 - y) y, 1e-12) * act'(y)            (without 1 / count: the consumers scale)
//     dz[b, :]   += g W_o[i, :]        registers, combined over the half-waves in fixed order
//     dW_o[i, :] += g z[b, :]          float atomics, 512 contiguous bytes per position
//     db_o[i]    += g
//   count (spread, see YR_COUNT_SLOTS) += positions.  dW_o / db_o must be zero where no earlier position of
//   this step wrote; dz zero on entry when S > 1.
constexpr int kHalf = 32;
constexpr int kHalves = kBlock / kHalf;

// LOSS_ONLY (validation: NULL gradient buffers): lane l holds floats [NK l, NK l + NK) of a row (one 16-byte load per
// lane and position at H = 128), and the BCE term of a position is computed ONCE, by the lane whose number is the
// position's place in its group of 32 — the exponential and the two logarithms were 60 % of the instructions when
// every lane of the half-wave computed them (82 -> 45 us per 4,096 rows).
template <int NK, bool LOSS_ONLY>   // H <= 32 * NK: lane l of a half-wave holds floats l, l + 32, ... of a row, so that every
                        // load / atomic instruction of a half-wave covers 128 contiguous bytes
__global__ __launch_bounds__(kBlock) void cdae_sampled_decode_kernel(
    const int32_t* __restrict__ lcols, const float* __restrict__ lvals, const int32_t* __restrict__ lcount,
    int64_t cpp, const float* __restrict__ z, const float* __restrict__ Wo, const float* __restrict__ bo, int H,
    int act, int splits, float* __restrict__ dz, float* __restrict__ dWo, float* __restrict__ dbo,
    float* __restrict__ partial_loss, int32_t* __restrict__ count) {
  __shared__ int s_pre[kParts + 1];
  __shared__ int32_t s_col[kListCap];
  __shared__ float s_val[kListCap];
  __shared__ float s_dz[kHalves][kHalf * NK];
  __shared__ float s_loss[kHalves];
  __shared__ int s_done[kHalves];
  const int64_t r = blockIdx.x;
  const int split = blockIdx.y;
  const int lane = threadIdx.x & (kHalf - 1), half = threadIdx.x / kHalf;
  float zr[NK], acc[NK];
  bool in[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    in[k] = lane + kHalf * k < H;
    zr[k] = in[k] ? z[r * H + lane + kHalf * k] : 0.0f;
    acc[k] = 0.0f;
  }
  float loss = 0.0f;
  int done = 0;
  const bool grads = !LOSS_ONLY && dWo != nullptr;   // NULL gradient buffers: the loss only (validation)
  if constexpr (LOSS_ONLY) {
    static_assert(NK == 4 || NK == 8, "16-byte loads");
#pragma unroll
    for (int k = 0; k < NK; ++k) {                  // this form's layout of z
      in[k] = NK * lane + k < H;
      zr[k] = in[k] ? z[r * H + NK * lane + k] : 0.0f;
    }
  }
  for (int skip = 0;; skip += kListCap) {
    const int n = gather_row_list(lcols, lvals, lcount, cpp, r, skip, s_pre, s_col, s_val);
    if constexpr (LOSS_ONLY) {
      float pre_mine = 0.0f, t_mine = 0.0f;
      bool have = false;
      int slot = 0;
      auto settle = [&]() {                         // every lane: the BCE term of the position it was handed
        if (have) {
          float y = pre_mine;
          if (act == 1) y = 1.0f / (1.0f + expf(-y));
          loss -= t_mine * fmaxf(logf(y), -100.0f) + (1.0f - t_mine) * fmaxf(logf(1.0f - y), -100.0f);
        }
        have = false;
        slot = 0;
      };
      for (int j0 = split + half * splits; j0 < n; j0 += 2 * kHalves * splits) {
        const int j1 = j0 + kHalves * splits;
        const bool two = j1 < n;
        const int col0 = s_col[j0], col1 = s_col[two ? j1 : j0];
        float4 w0[NK / 4], w1[NK / 4];
#pragma unroll
        for (int k = 0; k < NK / 4; ++k) {          // (rows are 16-byte aligned: H % 4 == 0 checked by the entry point)
          const int q = min(NK * lane + 4 * k, H - 4);
          w0[k] = *reinterpret_cast<const float4*>(Wo + (int64_t)col0 * H + q);
          w1[k] = *reinterpret_cast<const float4*>(Wo + (int64_t)col1 * H + q);
        }
        const float b0 = bo ? bo[col0] : 0.0f, b1 = bo ? bo[col1] : 0.0f;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          if (which == 1 && !two) break;
          float d = 0.0f;
#pragma unroll
          for (int k = 0; k < NK / 4; ++k) {
            const float4 w = which ? w1[k] : w0[k];
            d += (in[4 * k] ? w.x * zr[4 * k] : 0.0f) + (in[4 * k + 1] ? w.y * zr[4 * k + 1] : 0.0f) +
                 (in[4 * k + 2] ? w.z * zr[4 * k + 2] : 0.0f) + (in[4 * k + 3] ? w.w * zr[4 * k + 3] : 0.0f);
          }
          d = group_sum_dpp<kHalf>(d);
          if (lane == slot) {
            pre_mine = d + (which ? b1 : b0);
            t_mine = s_val[which ? j1 : j0];
            have = true;
          }
          ++done;
          if (++slot == kHalf) settle();
        }
      }
      settle();
    } else

    // this workgroup's share of the staged entries: j = split, split + splits, ...; half-wave `half` takes every
    // kHalves-th of those, two at a time (the second W_o row is in flight while the first is used)
    for (int j0 = split + half * splits; j0 < n; j0 += 2 * kHalves * splits) {
      const int j1 = j0 + kHalves * splits;
      const bool two = j1 < n;
      const int col0 = s_col[j0], col1 = s_col[two ? j1 : j0];
      float w0[NK], w1[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int q = min(lane + kHalf * k, H - 1);
        w0[k] = Wo[(int64_t)col0 * H + q];
        w1[k] = Wo[(int64_t)col1 * H + q];
      }
      const float b0 = bo ? bo[col0] : 0.0f, b1 = bo ? bo[col1] : 0.0f;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        if (which == 1 && !two) break;
        const int col = which ? col1 : col0;
        const float t = s_val[which ? j1 : j0];
        float d = 0.0f;
#pragma unroll
        for (int k = 0; k < NK; ++k) d += in[k] ? (which ? w1[k] : w0[k]) * zr[k] : 0.0f;
        d = group_sum_dpp<kHalf>(d);
        float y = d + (which ? b1 : b0);
        if (act == 1) y = 1.0f / (1.0f + expf(-y));
        loss -= t * fmaxf(logf(y), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - y), -100.0f);
        float g = (y - t) / fmaxf((1.0f - y) * y, 1e-12f);
        if (act == 1) g *= y * (1.0f - y);
        ++done;
        if (grads) {
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            acc[k] += g * (which ? w1[k] : w0[k]);
            if (in[k]) atomicAdd(dWo + (int64_t)col * H + lane + kHalf * k, g * zr[k]);
          }
          if (lane == 0) atomicAdd(dbo + col, g);
        }
      }
    }
    const bool more = skip + n < s_pre[kParts];
    __syncthreads();
    if (!more) break;
  }
  // combine the half-waves in fixed order
#pragma unroll
  for (int k = 0; k < NK; ++k) s_dz[half][lane + kHalf * k] = acc[k];
  if constexpr (LOSS_ONLY) loss = group_sum_dpp<kHalf>(loss);     // the lanes' own positions, fixed order
  if (lane == 0) { s_loss[half] = loss; s_done[half] = done; }
  __syncthreads();
  for (int h = threadIdx.x; grads && h < H; h += kBlock) {
    float tsum = 0.0f;
#pragma unroll
    for (int k = 0; k < kHalves; ++k) tsum += s_dz[k][h];
    if (splits > 1) atomicAdd(dz + r * H + h, tsum);
    else dz[r * H + h] = tsum;
  }
  if (threadIdx.x == 0) {
    float tl = 0.0f;
    int tot = 0;
#pragma unroll
    for (int k = 0; k < kHalves; ++k) { tl += s_loss[k]; tot += s_done[k]; }
    partial_loss[r * splits + split] = tl;
    spread_count_add(count, blockIdx.y * gridDim.x + blockIdx.x, tot);
  }
}

// stats[0] = (sum of the partials, fixed order) / count, stats[1] = count, *loss_accum += stats[0]
__global__ __launch_bounds__(kBlock) void cdae_loss_finalize_kernel(const float* __restrict__ partial_loss,
                                                                    int64_t n_partials, const int32_t* __restrict__ count,
                                                                    float* __restrict__ stats, double* __restrict__ loss_accum) {
  __shared__ float s_red[kWavesPerBlock];
  float s = 0.0f;
  for (int64_t k = threadIdx.x; k < n_partials; k += kBlock) s += partial_loss[k];
  const int32_t c = spread_count(count, threadIdx.x & (kWave - 1));
  const float tot = block_sum(s, s_red);
  if (threadIdx.x == 0) {
    const float mean = c > 0 ? tot / (float)c : 0.0f;
    stats[0] = mean;
    stats[1] = (float)c;
    if (loss_accum) loss_accum[0] += (double)mean;
  }
}

// The validation loss of SEVERAL batches scored by one yr_cdae_sampled_decode launch: workgroup q sums the loss
// partials and the position counts (loss_count, kParts per row) of the rows of batch q -> mean_q; the last
// workgroup to arrive adds the means in batch order (fixed order) to *loss_accum.  `arrive` is zero on entry and is
// left zero.
__global__ __launch_bounds__(kBlock) void cdae_loss_finalize_batched_kernel(
    const float* __restrict__ partial_loss, int splits, const int32_t* __restrict__ loss_count, int64_t rows,
    int64_t batch_rows, float* __restrict__ means, int32_t* __restrict__ arrive, double* __restrict__ loss_accum) {
  __shared__ float s_red[kWavesPerBlock];
  __shared__ int s_cnt[kWavesPerBlock];
  __shared__ int s_last;
  const int64_t r0 = (int64_t)blockIdx.x * batch_rows, r1 = min(rows, r0 + batch_rows);
  float s = 0.0f;
  for (int64_t k = r0 * splits + threadIdx.x; k < r1 * splits; k += kBlock) s += partial_loss[k];
  int c = 0;
  for (int64_t k = r0 * kParts + threadIdx.x; k < r1 * kParts; k += kBlock) c += loss_count[k];
#pragma unroll
  for (int d = kWave / 2; d >= 1; d >>= 1) c += __shfl_xor(c, d, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x / kWave] = c;
  const float tot = block_sum(s, s_red);
  if (threadIdx.x == 0) {
    int cnt = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) cnt += s_cnt[w];
    means[blockIdx.x] = cnt > 0 ? tot / (float)cnt : 0.0f;
    __threadfence();
    s_last = atomicAdd(arrive, 1) == (int)gridDim.x - 1;
    if (s_last) {
      __threadfence();
      double acc = 0.0;
      for (unsigned q = 0; q < gridDim.x; ++q) acc += (double)__builtin_nontemporal_load(means + q);
      if (loss_accum) loss_accum[0] += acc;
      *arrive = 0;
    }
  }
}

}  // namespace yr

using namespace yr;

extern "C" int yr_cdae_loss_finalize_batched(const float* partial_loss, int splits, const int32_t* loss_count,
                                             int64_t rows, int64_t batch_rows, float* means, int32_t* arrive,
                                             double* loss_accum, void* stream) {
  if (rows < 0 || batch_rows <= 0 || splits <= 0) return YR_ERR_BADARG;
  if (rows == 0) return 0;
  if (!partial_loss || !loss_count || !means || !arrive) return YR_ERR_BADARG;
  const int64_t batches = (rows + batch_rows - 1) / batch_rows;
  if (batches > 65535) return YR_ERR_BADARG;
  hipLaunchKernelGGL(cdae_loss_finalize_batched_kernel, dim3((unsigned)batches), dim3(kBlock), 0, (hipStream_t)stream,
                     partial_loss, splits, loss_count, rows, batch_rows, means, arrive, loss_accum);
  return launch_status();
}

extern "C" int64_t yr_cdae_sparse_part_columns(int64_t I) {
  // columns per part: I / 32 rounded up to a multiple of 4
  if (I <= 0) return YR_ERR_BADARG;
  return ((I + kParts - 1) / kParts + 3) / 4 * 4;
}

extern "C" int yr_cdae_compact_rows(const float* x, int64_t B, int64_t I, uint64_t seed, double p, int32_t* cols,
                                    float* vals, int32_t* count, void* stream) {
  if (B < 0 || I <= 0 || p < 0.0 || p >= 1.0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!x || !cols || !vals || !count) return YR_ERR_BADARG;
  const int64_t cpp = yr_cdae_sparse_part_columns(I);
  hipLaunchKernelGGL((cdae_compact_rows_kernel<false>), dim3((unsigned)B, kParts / kWavesPerBlock), dim3(kBlock), 0,
                     (hipStream_t)stream, x, (const float*)nullptr, I, seed, (float)p, (float)(1.0 / (1.0 - p)), cpp,
                     cols, vals, count, (int32_t*)nullptr, (float*)nullptr, (int32_t*)nullptr);
  return launch_status();
}

extern "C" int yr_cdae_sparse_encode(const int32_t* cols, const float* vals, const int32_t* count, const float* Wh,
                                     const float* bh, const float* V, const int64_t* user, int64_t B, int64_t I,
                                     int H, int64_t num_users, int act, float* z, int32_t* err_flag, void* stream) {
  if (B < 0 || I <= 0 || H <= 0 || num_users <= 0 || (act != 0 && act != 1)) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!cols || !vals || !count || !Wh || !bh || !V || !user || !z) return YR_ERR_BADARG;
  hipLaunchKernelGGL((cdae_sparse_encode_kernel<false>), dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, cols,
                     vals, count, yr_cdae_sparse_part_columns(I), Wh, bh, V, user, I, H, num_users, act, z, err_flag);
  return launch_status();
}

extern "C" int yr_cdae_sparse_encode_t(const int32_t* cols, const float* vals, const int32_t* count, const float* WhT,
                                       const float* bh, const float* V, const int64_t* user, int64_t B, int64_t I,
                                       int H, int64_t num_users, int act, float* z, int32_t* err_flag, void* stream) {
  if (B < 0 || I <= 0 || H <= 0 || num_users <= 0 || (act != 0 && act != 1)) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!cols || !vals || !count || !WhT || !bh || !V || !user || !z) return YR_ERR_BADARG;
  hipLaunchKernelGGL((cdae_sparse_encode_kernel<true>), dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, cols,
                     vals, count, yr_cdae_sparse_part_columns(I), WhT, bh, V, user, I, H, num_users, act, z, err_flag);
  return launch_status();
}

extern "C" int yr_cdae_sparse_dwh_t(const int32_t* cols, const float* vals, const int32_t* count, const float* dz,
                                    int64_t B, int64_t I, int H, float* dWhT, uint8_t* touched_items, void* stream) {
  if (B < 0 || I <= 0 || H <= 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!cols || !vals || !count || !dz || !dWhT || !touched_items) return YR_ERR_BADARG;
  hipLaunchKernelGGL(cdae_sparse_dwh_t_kernel, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, cols, vals,
                     count, yr_cdae_sparse_part_columns(I), dz, H, dWhT, touched_items);
  return launch_status();
}

extern "C" int yr_cdae_sparse_dwh(const int32_t* cols, const float* vals, const int32_t* count, const float* dz,
                                  int64_t B, int64_t I, int H, float* dWh, float* scratch_T, int32_t* claim,
                                  int32_t epoch, int32_t* touched, int32_t* n_touched, void* stream) {
  if (B < 0 || I <= 0 || H <= 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!cols || !vals || !count || !dz || !dWh || !scratch_T || !claim || !touched || !n_touched) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(n_touched, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  const int64_t cpp = yr_cdae_sparse_part_columns(I);
  hipLaunchKernelGGL(cdae_sparse_dwh_accumulate_kernel, dim3((unsigned)B), dim3(kBlock), 0, s, cols, vals, count, cpp,
                     dz, H, scratch_T, claim, epoch, touched, n_touched);
  hipLaunchKernelGGL(cdae_sparse_dwh_scatter_kernel, dim3(2048), dim3(kBlock), 0, s, touched, n_touched, scratch_T, I,
                     H, dWh);
  return launch_status();
}

extern "C" int yr_cdae_compact_pair(const float* x, const float* negative_mask, int64_t B, int64_t I, uint64_t seed,
                                    double p, int32_t* cols, float* vals, int32_t* count, int32_t* loss_cols,
                                    float* loss_targets, int32_t* loss_count, void* stream) {
  if (B < 0 || I <= 0 || p < 0.0 || p >= 1.0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!x || !negative_mask || !cols || !vals || !count || !loss_cols || !loss_targets || !loss_count)
    return YR_ERR_BADARG;
  const int64_t cpp = yr_cdae_sparse_part_columns(I);
  if (cpp >= 32768) return YR_ERR_UNSUPPORTED;           // the two prefix sums share one 32-bit scan
  hipLaunchKernelGGL((cdae_compact_rows_kernel<true>), dim3((unsigned)B, kParts / kWavesPerBlock), dim3(kBlock), 0,
                     (hipStream_t)stream, x, negative_mask, I, seed, (float)p, (float)(1.0 / (1.0 - p)), cpp, cols,
                     vals, count, loss_cols, loss_targets, loss_count);
  return launch_status();
}

// workgroups per row: enough of them to fill the chip at small batches (at 256 rows 1 ... 8 splits measured equal:
// the kernel then runs at the rate of its float atomics)
extern "C" int yr_cdae_sampled_decode_splits(int64_t B) {
  if (B <= 0) return 1;
  const int64_t s = 512 / B;
  return (int)(s < 1 ? 1 : s > 8 ? 8 : s);
}

extern "C" int yr_cdae_sampled_decode(const int32_t* loss_cols, const float* loss_targets, const int32_t* loss_count,
                                      const float* z, const float* Wo, const float* bo, int64_t B, int64_t I, int H,
                                      int act, float* dz, float* dWo, float* dbo, float* partial_loss,
                                      int32_t* count, void* stream) {
  if (B < 0 || I <= 0 || H <= 0 || (act != 0 && act != 1)) return YR_ERR_BADARG;
  if (H > 256) return YR_ERR_UNSUPPORTED;
  if (B == 0) return 0;
  if (!loss_cols || !loss_targets || !loss_count || !z || !Wo || !partial_loss || !count) return YR_ERR_BADARG;
  if ((dz || dWo || dbo) && !(dz && dWo && dbo)) return YR_ERR_BADARG;      // all three gradients or none
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(Wo)) & 15) return YR_ERR_BADARG;
  const int splits = yr_cdae_sampled_decode_splits(B);
  const int64_t cpp = yr_cdae_sparse_part_columns(I);
  const dim3 grid((unsigned)B, (unsigned)splits);
  hipStream_t s = (hipStream_t)stream;
  const bool loss_only = !dWo && H % 4 == 0;
#define YR_SD(NK, LO)                                                                                                 \
  hipLaunchKernelGGL((cdae_sampled_decode_kernel<NK, LO>), grid, dim3(kBlock), 0, s, loss_cols, loss_targets, loss_count, \
                     cpp, z, Wo, bo, H, act, splits, dz, dWo, dbo, partial_loss, count)
  if (H <= 128) { if (loss_only) YR_SD(4, true); else YR_SD(4, false); }
  else { if (loss_only) YR_SD(8, true); else YR_SD(8, false); }
#undef YR_SD
  return launch_status();
}

extern "C" int yr_cdae_loss_finalize(const float* partial_loss, int64_t n_partials, const int32_t* count, float* stats,
                                     double* loss_accum, void* stream) {
  if (n_partials < 0 || !partial_loss || !count || !stats) return YR_ERR_BADARG;
  hipLaunchKernelGGL(cdae_loss_finalize_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, partial_loss, n_partials,
                     count, stats, loss_accum);
  return launch_status();
}

extern "C" int yr_cdae_hidden_bwd_dwh_t(const int32_t* cols, const float* vals, const int32_t* count, const float* dz,
                                        const float* z, int act, int scale_dz, const int32_t* pos_count,
                                        const int64_t* user, int64_t B, int64_t I, int H, int64_t num_users, float* dV,
                                        uint8_t* touched_users, float* dbh, float* dWhT, uint8_t* touched_items,
                                        const float* partial_loss, int64_t n_partials, float* stats,
                                        double* loss_accum, void* stream) {
  if (B < 0 || I <= 0 || H <= 0 || H > 2 * kBlock || num_users <= 0 || n_partials < 0 || (act != 0 && act != 1))
    return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!cols || !vals || !count || !dz || !z || !pos_count || !user || !dV || !dbh || !dWhT || !touched_items)
    return YR_ERR_BADARG;
  if (n_partials > 0 && (!partial_loss || !stats)) return YR_ERR_BADARG;
  hipLaunchKernelGGL(cdae_hidden_bwd_dwh_t_kernel, dim3((unsigned)((B + kHbRows - 1) / kHbRows)), dim3(kBlock), 0,
                     (hipStream_t)stream, cols, vals, count, yr_cdae_sparse_part_columns(I), dz, z, act,
                     scale_dz ? 1 : 0, pos_count, user, B, num_users, H,
                     dV, touched_users, dbh, dWhT, touched_items, partial_loss, n_partials, stats, loss_accum);
  return launch_status();
}
