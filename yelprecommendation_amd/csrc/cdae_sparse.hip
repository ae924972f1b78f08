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

__global__ __launch_bounds__(kBlock) void cdae_compact_rows_kernel(const float* __restrict__ x, int64_t I,
                                                                   uint64_t seed, float p, float scale, int64_t cpp,
                                                                   int32_t* __restrict__ cols, float* __restrict__ vals,
                                                                   int32_t* __restrict__ count) {
  const int64_t r = blockIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int part = blockIdx.y * kWavesPerBlock + wave;
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const float* row = x + r * I;
  const int64_t c_lo = (int64_t)part * cpp, c_hi = min(I, c_lo + cpp);
  int32_t* oc = cols + (r * kParts + part) * cpp;
  float* ov = vals + (r * kParts + part) * cpp;
  int base = 0;
  for (int64_t c0 = c_lo; c0 < c_hi; c0 += kWave * 4) {
    const int64_t c = c0 + lane * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c + k < c_hi) v[k] = row[c + k];
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
    const int mine = (v[0] != 0.f) + (v[1] != 0.f) + (v[2] != 0.f) + (v[3] != 0.f);
    int inc = mine;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int t = __shfl_up(inc, d, kWave);
      if (lane >= d) inc += t;
    }
    int at = base + inc - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v[k] != 0.f) { oc[at] = (int32_t)(c + k); ov[at] = v[k]; ++at; }
    base += __shfl(inc, kWave - 1, kWave);
  }
  if (lane == 0) count[r * kParts + part] = base;
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

// one workgroup per row, one thread per hidden unit (looped when H > 256)
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
    const float* wrow = Wh + (int64_t)min(h, H - 1) * I;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int skip = 0;; skip += kListCap) {
      const int n = gather_row_list(cols, vals, count, cpp, r, skip, s_pre, s_col, s_val);
      int j = 0;
      for (; j + 4 <= n; j += 4) {
        acc0 = fmaf(s_val[j], wrow[s_col[j]], acc0);
        acc1 = fmaf(s_val[j + 1], wrow[s_col[j + 1]], acc1);
        acc2 = fmaf(s_val[j + 2], wrow[s_col[j + 2]], acc2);
        acc3 = fmaf(s_val[j + 3], wrow[s_col[j + 3]], acc3);
      }
      for (; j < n; ++j) acc0 = fmaf(s_val[j], wrow[s_col[j]], acc0);
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

}  // namespace yr

using namespace yr;

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
  hipLaunchKernelGGL(cdae_compact_rows_kernel, dim3((unsigned)B, kParts / kWavesPerBlock), dim3(kBlock), 0,
                     (hipStream_t)stream, x, I, seed, (float)p, (float)(1.0 / (1.0 - p)),
                     yr_cdae_sparse_part_columns(I), cols, vals, count);
  return launch_status();
}

extern "C" int yr_cdae_sparse_encode(const int32_t* cols, const float* vals, const int32_t* count, const float* Wh,
                                     const float* bh, const float* V, const int64_t* user, int64_t B, int64_t I,
                                     int H, int64_t num_users, int act, float* z, int32_t* err_flag, void* stream) {
  if (B < 0 || I <= 0 || H <= 0 || num_users <= 0 || (act != 0 && act != 1)) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!cols || !vals || !count || !Wh || !bh || !V || !user || !z) return YR_ERR_BADARG;
  hipLaunchKernelGGL(cdae_sparse_encode_kernel, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, cols, vals,
                     count, yr_cdae_sparse_part_columns(I), Wh, bh, V, user, I, H, num_users, act, z, err_flag);
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
