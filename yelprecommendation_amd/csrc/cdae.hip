// CDAE kernels for gfx950 (MI355X): a float32 MFMA GEMM for the encoder / decoder / their
// gradients, and the element-wise and masked-BCE pieces around it.
//
// Replaces, for reference models/cdae.py:46-52 and loss.py:12-16 (+ their autograd):
//   nn.Dropout, nn.Linear(I -> H) (+ bias, + nn.Embedding user row), sigmoid, nn.Linear(H -> I)
//   (+ bias), sigmoid, nonzero(target + negative_mask) + binary_cross_entropy.
// The two Linear layers and their three gradient products are GEMMs on the matrix cores
// (v_mfma_f32_32x32x2_f32, exact f32).  Everything else is byte-bound element-wise work.
#include "common.h"

namespace yr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// --------------------------------------------------------------------------- f32 MFMA GEMM
// C[M, N] (+)= op(A)[M, K] . op(B)[K, N]     row-major, leading dimensions lda / ldb / ldc
//   transA = 0: A(m, k) = A[m*lda + k]      transA = 1: A(m, k) = A[k*lda + m]
//   transB = 0: B(k, n) = B[k*ldb + n]      transB = 1: B(k, n) = B[n*ldb + k]
// Workgroup = 4 waves = a 64 x 64 tile of C (each wave 32 x 32), K walked in steps of 32 staged in
// LDS as sA[m][k], sB[n][k] with pitch 33 (conflict-free for both the fill and the MFMA operand
// reads: lane (i, h) reads row i, k = 16 h + s).  blockIdx.z splits K; with more than one split
// (or accumulate) the epilogue adds atomically, otherwise it stores act(acc + bias[n]).
constexpr int kGemmTile = 64;
constexpr int kGemmK = 32;
constexpr int kGemmPitch = kGemmK + 1;

__global__ __launch_bounds__(kBlock) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, int64_t lda,
                                                          int64_t ldb, int64_t ldc, int transA, int transB,
                                                          const float* __restrict__ bias, int act, int atomic,
                                                          int k_per_split) {
  __shared__ float sA[kGemmTile * kGemmPitch];
  __shared__ float sB[kGemmTile * kGemmPitch];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * kGemmTile, n0 = blockIdx.x * kGemmTile;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k0 = kbeg; k0 < kend; k0 += kGemmK) {
    // fill: 64 x 32 elements of each operand, 8 per thread, coalesced along the contiguous axis
#pragma unroll
    for (int q = 0; q < (kGemmTile * kGemmK) / kBlock; ++q) {
      const int e = threadIdx.x + q * kBlock;
      int m, k;
      if (transA) { k = e / kGemmTile; m = e % kGemmTile; } else { m = e / kGemmK; k = e % kGemmK; }
      float v = 0.0f;
      if (m0 + m < M && k0 + k < kend)
        v = transA ? A[(int64_t)(k0 + k) * lda + (m0 + m)] : A[(int64_t)(m0 + m) * lda + (k0 + k)];
      sA[m * kGemmPitch + k] = v;
      int n, kb;
      if (transB) { n = e / kGemmK; kb = e % kGemmK; } else { kb = e / kGemmTile; n = e % kGemmTile; }
      float w = 0.0f;
      if (n0 + n < N && k0 + kb < kend)
        w = transB ? B[(int64_t)(n0 + n) * ldb + (k0 + kb)] : B[(int64_t)(k0 + kb) * ldb + (n0 + n)];
      sB[n * kGemmPitch + kb] = w;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kGemmK / 2; ++s) {
      const float a = sA[(wm + i) * kGemmPitch + h * (kGemmK / 2) + s];
      const float b = sB[(wn + i) * kGemmPitch + h * (kGemmK / 2) + s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn + i;
  if (col < N) {
    const float bv = (bias && !atomic) ? bias[col] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row < M) {
        float* dst = C + (int64_t)row * ldc + col;
        if (atomic) {
          atomicAdd(dst, acc[reg]);
        } else {
          float v = acc[reg] + bv;
          if (act == 1) v = 1.0f / (1.0f + expf(-v));
          *dst = v;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------- tiled f32 MFMA GEMM
// The fast path of yr_gemm_f32 (operands 16-byte aligned, leading dimensions multiples of 4):
// workgroup = 4 waves = a BM x BN tile of C (BM, BN in {64, 128}; each wave a quarter, 1-4 MFMA
// tiles of 32 x 32), K walked in steps of 16.  Both operand tiles live in LDS as [k][row] (row
// contiguous, pitch rows + 4), so an MFMA operand read is one conflict-free ds_read_b32 per lane
// (lane (i, h): k = 2 s + h, row = i) shared by up to two MFMAs.  The next K-step is fetched from
// global memory with 16-byte loads into registers while the MFMAs of the current one run, then
// written to the other LDS buffer: one barrier per K-step.
//   K-contiguous operand (A with transA = 0, B with transB = 1): lane -> (row = lane % 16 + 16 c,
//     4 k's): 16 rows x 64 B per wave instruction, transposed into [k][row] by 4 ds_write_b32
//     (banks 16 (lane / 16) + lane % 16: conflict-free).
//   row-contiguous operand (transA = 1, transB = 0): lane -> (4 rows, one k): 512 B contiguous
//     per 32 lanes, one aligned ds_write_b128.
constexpr int kTK = 16;

template <int R, bool KCONTIG>
struct TileLoader {
  static constexpr int P = R + 4;
  static constexpr int NV = R / 64;                    // float4 per thread per K-step

  // fetch this thread's pieces of the tile rows [r0, r0 + R) x k [k0, k0 + 16) (zero outside
  // rows < rmax, k < kend)
  __device__ static __forceinline__ void fetch(const float* __restrict__ X, int64_t ld, int r0, int rmax, int k0,
                                               int kend, float4 (&v)[NV]) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KCONTIG) {
        const int r = r0 + (lane & 15) + 16 * (wave + 4 * j);
        const int k = k0 + 4 * (lane >> 4);
        if (r < rmax) {
          const float* src = X + (int64_t)r * ld + k;
          if (k + 3 < kend) {
            x = *reinterpret_cast<const float4*>(src);
          } else {
            if (k + 0 < kend) x.x = src[0];
            if (k + 1 < kend) x.y = src[1];
            if (k + 2 < kend) x.z = src[2];
          }
        }
      } else {
        const int idx = threadIdx.x + j * kBlock;
        const int r = r0 + 4 * (idx % (R / 4));
        const int k = k0 + idx / (R / 4);
        if (k < kend) {
          const float* src = X + (int64_t)k * ld + r;
          if (r + 3 < rmax) {
            x = *reinterpret_cast<const float4*>(src);
          } else {
            if (r + 0 < rmax) x.x = src[0];
            if (r + 1 < rmax) x.y = src[1];
            if (r + 2 < rmax) x.z = src[2];
          }
        }
      }
      v[j] = x;
    }
  }

  __device__ static __forceinline__ void stash(float* __restrict__ s, const float4 (&v)[NV]) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (KCONTIG) {
        const int r = (lane & 15) + 16 * (wave + 4 * j);
        const int k = 4 * (lane >> 4);
        s[(k + 0) * P + r] = v[j].x;
        s[(k + 1) * P + r] = v[j].y;
        s[(k + 2) * P + r] = v[j].z;
        s[(k + 3) * P + r] = v[j].w;
      } else {
        const int idx = threadIdx.x + j * kBlock;
        *reinterpret_cast<float4*>(s + (idx / (R / 4)) * P + 4 * (idx % (R / 4))) = v[j];
      }
    }
  }
};

// Optional extras of the tiled kernel (all NULL for a plain GEMM):
//   count      C gets alpha * (op(A) op(B)), alpha = 1 / *count (0 when the count is 0) — the 1 / (number of
//              loss positions) factor of a mean loss, known only on the device;
//   rowsum     rowsum[m] = alpha * sum_k op(A)(m, k), taken from the operand values the MFMAs read anyway (the
//              bias gradient next to a weight gradient dW = G^T z: no separate column-sum pass over G);
//   EPI == 1   (decoder of the CDAE training step) y = act(acc + bias), and instead of y the kernel stores the
//              gradient of the NS-BCE loss w.r.t. the pre-activation, without its 1 / count factor:
//                  selected (target + negmask != 0):  (y - t) / max((1 - y) y, 1e-12) * act'(y),  else 0
//              plus one loss partial per workgroup (fixed summation order) and the number of selected positions
//              (integer atomic: exact, order-free).  `pred` (may be NULL) receives y.
struct GemmExtra {
  const int32_t* count;
  float* rowsum;
  const float* target;
  const float* negmask;
  int64_t ldt;
  float* pred;
  float* partial_loss;
  int32_t* count_out;
};

template <int BM, int BN, bool TA, bool TB, int EPI = 0>
__global__ __launch_bounds__(kBlock) void gemm_f32_tiled_kernel(const float* __restrict__ A,
                                                                const float* __restrict__ B, float* __restrict__ C,
                                                                int M, int N, int K, int64_t lda, int64_t ldb,
                                                                int64_t ldc, const float* __restrict__ bias, int act,
                                                                int atomic, int k_per_split, GemmExtra ex) {
  using LA = TileLoader<BM, !TA>;
  using LB = TileLoader<BN, TB>;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  __shared__ float sA[2][kTK * LA::P];
  __shared__ float sB[2][kTK * LB::P];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const bool want_rowsum = EPI == 0 && ex.rowsum != nullptr;
  f32x16 acc[TM][TN];
  float rs[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    rs[a] = 0.0f;
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  }

  float4 va[LA::NV], vb[LB::NV];
  LA::fetch(A, lda, m0, M, kbeg, kend, va);
  LB::fetch(B, ldb, n0, N, kbeg, kend, vb);
  // EPI == 1: this lane's target / mask values are fetched now, under the product (they do not depend on it;
  // loaded one by one in the epilogue, between stores the compiler must assume alias them, they cost a
  // dependent memory round trip each: 149 us instead of ~50 for the decoder at full size)
  constexpr int NE = EPI == 1 ? TM * TN * 16 : 1;
  float tv[NE], mv[NE];
  if (EPI == 1) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn + b * 32 + i;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = m0 + wm + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const int e = (b * TM + a) * 16 + reg;
          const bool in = col < N && row < M;
          const int64_t at = in ? (int64_t)row * ex.ldt + col : 0;   // clamped address: a plain load, no branch
          const float t = ex.target[at];
          const float m = ex.negmask ? ex.negmask[at] : 1.0f;
          tv[e] = in ? t : 0.0f;
          mv[e] = in ? m : 0.0f;
        }
      }
    }
  }
  LA::stash(sA[0], va);
  LB::stash(sB[0], vb);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kTK) {
    const bool more = k0 + kTK < kend;
    if (more) {
      LA::fetch(A, lda, m0, M, k0 + kTK, kend, va);
      LB::fetch(B, ldb, n0, N, k0 + kTK, kend, vb);
    }
    const float* a_s = sA[buf];
    const float* b_s = sB[buf];
#pragma unroll
    for (int s = 0; s < kTK / 2; ++s) {
      const int k = 2 * s + h;
      float av[TM], bv[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) av[a] = a_s[k * LA::P + wm + a * 32 + i];
#pragma unroll
      for (int b = 0; b < TN; ++b) bv[b] = b_s[k * LB::P + wn + b * 32 + i];
      if (want_rowsum) {
#pragma unroll
        for (int a = 0; a < TM; ++a) rs[a] += av[a];
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (more) {
      LA::stash(sA[buf ^ 1], va);
      LB::stash(sB[buf ^ 1], vb);
    }
    __syncthreads();
    buf ^= 1;
  }

  float alpha = 1.0f;
  if (ex.count) {
    const int32_t c = spread_count(ex.count, lane);
    alpha = c > 0 ? 1.0f / (float)c : 0.0f;
  }
  if (want_rowsum && blockIdx.y == 0 && (wave & 1) == 0) {       // k even (h = 0) + k odd (h = 1), in this order
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const float other = __shfl_xor(rs[a], 32, kWave);
      const int row = m0 + wm + a * 32 + i;
      if (h == 0 && row < M) ex.rowsum[row] = alpha * (rs[a] + other);
    }
  }

  float loss = 0.0f;
  int cnt = 0;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = n0 + wn + b * 32 + i;
    const bool col_ok = col < N;
    const float bvl = (bias && !atomic && col_ok) ? bias[col] : 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wm + a * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (col_ok && row < M) {
          float* dst = C + (int64_t)row * ldc + col;
          if (EPI == 1) {
            float y = acc[a][b][reg] + bvl;
            if (act == 1) y = 1.0f / (1.0f + expf(-y));
            const float t = tv[(b * TM + a) * 16 + reg];
            const float m = mv[(b * TM + a) * 16 + reg];
            float g = 0.0f;
            if (t + m != 0.0f) {
              loss -= t * fmaxf(logf(y), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - y), -100.0f);
              ++cnt;
              g = (y - t) / fmaxf((1.0f - y) * y, 1e-12f);
              if (act == 1) g *= y * (1.0f - y);
            }
            *dst = g;
            if (ex.pred) ex.pred[(int64_t)row * ldc + col] = y;
          } else if (atomic) {
            atomicAdd(dst, alpha * acc[a][b][reg]);
          } else {
            float v = alpha * acc[a][b][reg] + bvl;
            if (act == 1) v = 1.0f / (1.0f + expf(-v));
            *dst = v;
          }
        }
      }
    }
  }
  if (EPI == 1) {
    float* s_red = &sA[0][0];                          // the operand tiles are dead: every wave passed the last barrier
    const float tl = block_sum(loss, s_red);
    if (threadIdx.x == 0) ex.partial_loss[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = tl;
    int c = cnt;
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1) c += __shfl_xor(c, d, kWave);
    __shared__ int s_cnt[kWavesPerBlock];
    if (lane == 0) s_cnt[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tot = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
      const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
      spread_count_add(ex.count_out, wg, tot);
    }
  }
}

// --------------------------------------------------------------------------- element-wise pieces
// zpre[b, :] = bias[:] + V[user[b], :]      (b_h + user_nodes(user_id), models/cdae.py:49)
__global__ __launch_bounds__(kBlock) void cdae_hidden_init_kernel(float* __restrict__ zpre,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ V,
                                                                  const int64_t* __restrict__ user, int64_t B, int H,
                                                                  int64_t num_users, int32_t* __restrict__ err_flag) {
  const int64_t total = B * H;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
    const int64_t b = e / H;
    const int c = (int)(e % H);
    const int64_t u = user[b];
    float v = bias[c];
    if ((uint64_t)u < (uint64_t)num_users) v += V[u * H + c];
    else if (err_flag) atomicOr(err_flag, YR_FLAG_BAD_USER);
    zpre[e] = v;
  }
}

// out = rnd >= p ? x / (1 - p) : 0     (nn.Dropout(p) in training mode, models/cdae.py:43-44)
__global__ __launch_bounds__(kBlock) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ rnd,
                                                         float p, float scale, int64_t n, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride)
    out[e] = rnd[e] >= p ? x[e] * scale : 0.0f;
}

// nn.Dropout with the uniforms drawn in the kernel: Philox4x32-10 keyed by `seed`, counter = index
// of the float4 group, so the mask depends on (seed, position) only — no 4-byte-per-element random
// tensor is written and read back.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
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

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(kBlock) void dropout_seeded_kernel(const float* __restrict__ x, uint64_t seed, float p,
                                                                float scale, int64_t n, float* __restrict__ out) {
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const int64_t n4 = (n + 3) / 4;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n4; q += stride) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)q, (uint32_t)(q >> 32), 0u, 0u), key);
    const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
    const int64_t e = 4 * q;
    if (e + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + e);
      *reinterpret_cast<float4*>(out + e) = make_float4(u[0] >= p ? v.x * scale : 0.0f, u[1] >= p ? v.y * scale : 0.0f,
                                                        u[2] >= p ? v.z * scale : 0.0f, u[3] >= p ? v.w * scale : 0.0f);
    } else {
      for (int k = 0; e + k < n; ++k) out[e + k] = u[k] >= p ? x[e + k] * scale : 0.0f;
    }
  }
}

__global__ __launch_bounds__(kBlock) void sigmoid_kernel(float* __restrict__ x, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride)
    x[e] = 1.0f / (1.0f + expf(-x[e]));
}

// g = dy * y (1 - y)     (sigmoid backward through the OUTPUT y; g may be dy itself)
__global__ __launch_bounds__(kBlock) void sigmoid_bwd_kernel(const float* dy, const float* __restrict__ y,
                                                             float* g, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) g[e] = dy[e] * (y[e] * (1.0f - y[e]));
}

// out[c] (+)= sum_r X[r, c]        (bias gradients)
// Workgroup = 64 columns; wave w sums rows w, w + 4, ... (256 contiguous bytes per load, 8 loads
// in flight), the four partial sums are combined in wave order: deterministic.
__global__ __launch_bounds__(kBlock) void colsum_kernel(const float* __restrict__ X, int64_t rows, int64_t cols,
                                                        float* __restrict__ out, int accumulate) {
  __shared__ float s_part[kWavesPerBlock][kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t c = (int64_t)blockIdx.x * kWave + lane;
  float acc = 0.0f;
  if (c < cols) {
    constexpr int U = 8;
    int64_t r = wave;
    for (; r + (U - 1) * kWavesPerBlock < rows; r += U * kWavesPerBlock) {
      float v[U];
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = X[(r + q * kWavesPerBlock) * cols + c];
#pragma unroll
      for (int q = 0; q < U; ++q) acc += v[q];
    }
    for (; r < rows; r += kWavesPerBlock) acc += X[r * cols + c];
  }
  s_part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < cols) {
    const float t = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// dV[user[b], :] += G[b, :]        (embedding_dense_backward of user_nodes)
__global__ __launch_bounds__(kBlock) void row_scatter_add_kernel(const float* __restrict__ G,
                                                                 const int64_t* __restrict__ user, int64_t B, int H,
                                                                 int64_t num_users, float* __restrict__ dV) {
  const int64_t total = B * H;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
    const int64_t u = user[e / H];
    if ((uint64_t)u < (uint64_t)num_users) atomicAdd(dV + u * H + (e % H), G[e]);
  }
}

// NS-BCE (loss.py:12-16): positions with target + negative_mask != 0 are selected;
// stats[0] += sum of -(t max(log p, -100) + (1 - t) max(log(1 - p), -100)), stats[1] += count.
__global__ __launch_bounds__(kBlock) void nsbce_fwd_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ target,
                                                           const float* __restrict__ negmask, int64_t n,
                                                           float* __restrict__ partial_loss,
                                                           float* __restrict__ partial_cnt) {
  __shared__ float s_red[kWavesPerBlock];
  float loss = 0.0f, cnt = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
    const float t = target[e];
    const float m = negmask ? negmask[e] : 1.0f;
    if (t + m != 0.0f) {
      const float p = pred[e];
      loss -= t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - p), -100.0f);
      cnt += 1.0f;
    }
  }
  const float tl = block_sum(loss, s_red);
  __syncthreads();
  const float tc = block_sum(cnt, s_red);
  if (threadIdx.x == 0) {
    partial_loss[blockIdx.x] = tl;
    partial_cnt[blockIdx.x] = tc;
  }
}

// stats[0] = mean loss, stats[1] = count     (fixed-order sums of the partials)
__global__ __launch_bounds__(kBlock) void nsbce_finalize_kernel(const float* __restrict__ partial_loss,
                                                                const float* __restrict__ partial_cnt, int nparts,
                                                                float* __restrict__ stats) {
  __shared__ float s_red[kWavesPerBlock];
  float l = 0.0f, c = 0.0f;
  for (int i = threadIdx.x; i < nparts; i += kBlock) { l += partial_loss[i]; c += partial_cnt[i]; }
  const float tl = block_sum(l, s_red);
  __syncthreads();
  const float tc = block_sum(c, s_red);
  if (threadIdx.x == 0) {
    stats[0] = tc > 0.0f ? tl / tc : 0.0f;
    stats[1] = tc;
  }
}

// dpred = gout * (p - t) / max((1 - p) p, 1e-12) / count on the selected positions, 0 elsewhere
__global__ __launch_bounds__(kBlock) void nsbce_bwd_kernel(const float* __restrict__ pred,
                                                           const float* __restrict__ target,
                                                           const float* __restrict__ negmask,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gout, int64_t n,
                                                           float* __restrict__ dpred) {
  const float scale = stats[1] > 0.0f ? gout[0] / stats[1] : 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
    const float t = target[e];
    const float m = negmask ? negmask[e] : 1.0f;
    float g = 0.0f;
    if (t + m != 0.0f) {
      const float p = pred[e];
      g = (p - t) / fmaxf((1.0f - p) * p, 1e-12f) * scale;
    }
    dpred[e] = g;
  }
}

// One launch for what follows the input-gradient product of the decoder in the CDAE training step:
//   dz <- dz * act'(z) (in place; dz = G W_o arrives already scaled by 1 / count),
//   dbh[c] = sum_b dz[b, c]  (wave w of the 16 takes rows w, w + 16, ...; the 16 partial sums are combined in
//   wave order: deterministic),  dV[user[b], :] += dz[b, :] with the user's row marked in `touched`,
// and, in one extra workgroup, the loss of the step: stats[0] = (sum of the decoder's per-workgroup
// partials, fixed order) / count, stats[1] = count, *loss_accum += stats[0].
constexpr int kHiddenBwdBlock = 1024;
constexpr int kHiddenBwdWaves = kHiddenBwdBlock / kWave;

__global__ __launch_bounds__(kHiddenBwdBlock) void cdae_hidden_bwd_kernel(
    float* __restrict__ dz, const float* __restrict__ z, int act, const int64_t* __restrict__ user, int64_t B, int H,
    int64_t num_users, float* __restrict__ dV, uint8_t* __restrict__ touched, float* __restrict__ dbh,
    const float* __restrict__ partial_loss, int64_t n_partials, const int32_t* __restrict__ count,
    float* __restrict__ stats, double* __restrict__ loss_accum, unsigned col_blocks, int scale_dz) {
  __shared__ float s_part[kHiddenBwdWaves][kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (blockIdx.x >= col_blocks) {                    // the loss workgroup
    float s = 0.0f;
    for (int64_t k = threadIdx.x; k < n_partials; k += kHiddenBwdBlock) s += partial_loss[k];
    s = wave_sum(s);
    if (lane == 0) s_part[wave][0] = s;
    const int32_t c = spread_count(count, lane);
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.0f;
      for (int w = 0; w < kHiddenBwdWaves; ++w) tot += s_part[w][0];
      const float mean = c > 0 ? tot / (float)c : 0.0f;
      stats[0] = mean;
      stats[1] = (float)c;
      if (loss_accum) loss_accum[0] += (double)mean;
    }
    return;
  }
  const int c = blockIdx.x * kWave + lane;
  float acc = 0.0f;
  float alpha = 1.0f;
  if (scale_dz) {                                      // dz arrives without the 1 / count of the mean loss
    const int32_t n = spread_count(count, lane);
    alpha = n > 0 ? 1.0f / (float)n : 0.0f;
  }
  if (c < H) {
    constexpr int U = 8;                               // rows in flight per wave: loads first, then stores / atomics
    for (int64_t r0 = wave; r0 < B; r0 += U * kHiddenBwdWaves) {
      float g[U], y[U];
      int64_t u[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t r = min(r0 + (int64_t)q * kHiddenBwdWaves, B - 1);
        g[q] = dz[r * H + c];
        y[q] = z[r * H + c];
        u[q] = user[r];
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t r = r0 + (int64_t)q * kHiddenBwdWaves;
        if (r >= B) break;
        g[q] *= alpha;
        if (act == 1) g[q] *= y[q] * (1.0f - y[q]);
        dz[r * H + c] = g[q];
        acc += g[q];
        if ((uint64_t)u[q] < (uint64_t)num_users) {
          atomicAdd(dV + u[q] * H + c, g[q]);
          if (touched && c == 0) touched[u[q]] = 1;
        }
      }
    }
  }
  s_part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < H) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < kHiddenBwdWaves; ++w) t += s_part[w][lane];
    dbh[c] = t;
  }
}

inline int ew_grid(int64_t n) { return grid_for(n, kBlock); }

}  // namespace yr

using namespace yr;

extern "C" int yr_gemm_f32_ex(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                              const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                              int accumulate, int split_k, const int32_t* alpha_count, float* rowsum, void* stream) {
  if (M < 0 || N < 0 || K < 0 || M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) return YR_ERR_BADARG;
  if (M == 0 || N == 0) return 0;
  if (!A || !B || !C) return YR_ERR_BADARG;
  if (act != 0 && act != 1) return YR_ERR_UNSUPPORTED;
  if (split_k < 1) split_k = 1;
  int kps = (int)((K + split_k - 1) / split_k);
  kps = ((kps + kGemmK - 1) / kGemmK) * kGemmK;                 // whole K-steps per split
  if (kps == 0) kps = kGemmK;
  const int splits = (int)((K + kps - 1) / kps) > 0 ? (int)((K + kps - 1) / kps) : 1;
  const int atomic = (splits > 1 || accumulate) ? 1 : 0;
  if (atomic && (bias || act)) return YR_ERR_BADARG;            // fused epilogue only on a plain store
  if (rowsum && splits > 1) return YR_ERR_BADARG;               // the row sums come from one pass over all of K
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 &&
                       lda % 4 == 0 && ldb % 4 == 0;
  GemmExtra ex{};
  ex.count = alpha_count;
  ex.rowsum = rowsum;
  if (aligned) {
    // largest tile that still gives every CU about eight workgroups; shrink along the dimension
    // with more tiles first (the other operand keeps its reuse)
    int bm = 128, bn = 128;
    auto blocks = [&](int m, int n) { return ((M + m - 1) / m) * ((N + n - 1) / n) * (int64_t)splits; };
    const int64_t minb = 2048;                 // measured: ~8 workgroups per CU hide the fill latency best
    if (blocks(bm, bn) < minb) {
      if ((M + 127) / 128 >= (N + 127) / 128) bm = 64; else bn = 64;
      if (blocks(bm, bn) < minb) bm = bn = 64;
    }
    const dim3 grid((unsigned)((M + bm - 1) / bm), (unsigned)((N + bn - 1) / bn), (unsigned)splits);
    if (grid.y > 65535 || grid.z > 65535) return YR_ERR_BADARG;
#define YR_GEMM_LAUNCH(BM, BN, TA, TB)                                                                          \
  hipLaunchKernelGGL((gemm_f32_tiled_kernel<BM, BN, TA, TB>), grid, dim3(kBlock), 0, st, A, B, C, (int)M, (int)N, \
                     (int)K, lda, ldb, ldc, bias, act, atomic, kps, ex)
#define YR_GEMM_TRANS(BM, BN)                                        \
  do {                                                               \
    if (transA) {                                                    \
      if (transB) YR_GEMM_LAUNCH(BM, BN, true, true); else YR_GEMM_LAUNCH(BM, BN, true, false);   \
    } else {                                                         \
      if (transB) YR_GEMM_LAUNCH(BM, BN, false, true); else YR_GEMM_LAUNCH(BM, BN, false, false); \
    }                                                                \
  } while (0)
    if (bm == 128 && bn == 128) YR_GEMM_TRANS(128, 128);
    else if (bm == 64 && bn == 128) YR_GEMM_TRANS(64, 128);
    else if (bm == 128 && bn == 64) YR_GEMM_TRANS(128, 64);
    else YR_GEMM_TRANS(64, 64);
#undef YR_GEMM_TRANS
#undef YR_GEMM_LAUNCH
    return launch_status();
  }
  if (alpha_count || rowsum) return YR_ERR_UNSUPPORTED;          // the extras live in the tiled kernel only
  const dim3 grid((unsigned)((N + kGemmTile - 1) / kGemmTile), (unsigned)((M + kGemmTile - 1) / kGemmTile),
                  (unsigned)splits);
  if (grid.y > 65535 || grid.z > 65535) return YR_ERR_BADARG;
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(kBlock), 0, st, A, B, C, (int)M, (int)N, (int)K,
                     lda, ldb, ldc, transA ? 1 : 0, transB ? 1 : 0, bias, act, atomic, kps);
  return launch_status();
}

extern "C" int yr_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                           const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                           int accumulate, int split_k, void* stream) {
  return yr_gemm_f32_ex(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, accumulate, split_k, nullptr,
                        nullptr, stream);
}

// ---- fused pieces of the CDAE training step (yelprecommendation_amd/cdae_step.py) ----
constexpr int kDecodeTile = 64;

extern "C" int64_t yr_cdae_decode_loss_partials(int64_t B, int64_t I) {
  if (B < 0 || I < 0) return YR_ERR_BADARG;
  return ((B + kDecodeTile - 1) / kDecodeTile) * ((I + kDecodeTile - 1) / kDecodeTile);
}

extern "C" int yr_cdae_decode_loss(const float* z, const float* Wo, const float* bo, const float* target,
                                   const float* negative_mask, int64_t B, int64_t I, int H, int act, float* G,
                                   int64_t ldg, float* pred, float* partial_loss, int32_t* count, void* stream) {
  if (B < 0 || I < 0 || H <= 0 || B > 0x7fffffff || I > 0x7fffffff) return YR_ERR_BADARG;
  if (act != 0 && act != 1) return YR_ERR_UNSUPPORTED;
  if (B == 0 || I == 0) return 0;
  if (!z || !Wo || !target || !G || !partial_loss || !count || ldg < I) return YR_ERR_BADARG;
  if (((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(Wo)) & 15) || H % 4) return YR_ERR_BADARG;
  const dim3 grid((unsigned)((B + kDecodeTile - 1) / kDecodeTile), (unsigned)((I + kDecodeTile - 1) / kDecodeTile), 1);
  if (grid.y > 65535) return YR_ERR_BADARG;
  GemmExtra ex{};
  ex.target = target;
  ex.negmask = negative_mask;
  ex.ldt = I;
  ex.pred = pred;
  ex.partial_loss = partial_loss;
  ex.count_out = count;
  hipLaunchKernelGGL((gemm_f32_tiled_kernel<kDecodeTile, kDecodeTile, false, true, 1>), grid, dim3(kBlock), 0,
                     (hipStream_t)stream, z, Wo, G, (int)B, (int)I, H, (int64_t)H, (int64_t)H, ldg, bo, act, 0, H, ex);
  return launch_status();
}

extern "C" int yr_cdae_hidden_bwd(float* dz, const float* z, int act, const int64_t* user, int64_t B, int H,
                                  int64_t num_users, float* dV, uint8_t* touched_users, float* dbh,
                                  const float* partial_loss, int64_t n_partials, const int32_t* count, float* stats,
                                  double* loss_accum, int scale_dz, void* stream) {
  if (B < 0 || H <= 0 || num_users <= 0 || n_partials < 0 || (act != 0 && act != 1)) return YR_ERR_BADARG;
  if (!dz || !z || !user || !dV || !dbh) return YR_ERR_BADARG;
  if (n_partials > 0 && (!partial_loss || !count || !stats)) return YR_ERR_BADARG;
  if (scale_dz && !count) return YR_ERR_BADARG;
  const unsigned col_blocks = (unsigned)((H + kWave - 1) / kWave);
  hipLaunchKernelGGL(cdae_hidden_bwd_kernel, dim3(col_blocks + (n_partials > 0 ? 1u : 0u)), dim3(kHiddenBwdBlock), 0,
                     (hipStream_t)stream, dz, z, act, user, B, H, num_users, dV, touched_users, dbh, partial_loss,
                     n_partials, count, stats, loss_accum, col_blocks, scale_dz ? 1 : 0);
  return launch_status();
}

extern "C" int yr_cdae_hidden_init(float* zpre, const float* bias, const float* V, const int64_t* user, int64_t B,
                                   int H, int64_t num_users, int32_t* err_flag, void* stream) {
  if (B < 0 || H <= 0 || num_users <= 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!zpre || !bias || !V || !user) return YR_ERR_BADARG;
  hipLaunchKernelGGL(cdae_hidden_init_kernel, dim3(ew_grid(B * H)), dim3(kBlock), 0, (hipStream_t)stream, zpre, bias,
                     V, user, B, H, num_users, err_flag);
  return launch_status();
}

extern "C" int yr_dropout(const float* x, const float* rnd, double p, int64_t n, float* out, void* stream) {
  if (n < 0 || p < 0.0 || p >= 1.0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!x || !rnd || !out) return YR_ERR_BADARG;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, x, rnd, (float)p,
                     (float)(1.0 / (1.0 - p)), n, out);
  return launch_status();
}

extern "C" int yr_sigmoid(float* x, int64_t n, void* stream) {
  if (n < 0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!x) return YR_ERR_BADARG;
  hipLaunchKernelGGL(sigmoid_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, x, n);
  return launch_status();
}

extern "C" int yr_sigmoid_bwd(const float* dy, const float* y, float* g, int64_t n, void* stream) {
  if (n < 0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!dy || !g || !y) return YR_ERR_BADARG;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, dy, y, g, n);
  return launch_status();
}

extern "C" int yr_dropout_seeded(const float* x, uint64_t seed, double p, int64_t n, float* out, void* stream) {
  if (n < 0 || p < 0.0 || p >= 1.0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!x || !out || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15u)) return YR_ERR_BADARG;
  hipLaunchKernelGGL(dropout_seeded_kernel, dim3(ew_grid((n + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, x, seed,
                     (float)p, (float)(1.0 / (1.0 - p)), n, out);
  return launch_status();
}

extern "C" int yr_colsum(const float* X, int64_t rows, int64_t cols, float* out, int accumulate, void* stream) {
  if (rows < 0 || cols < 0) return YR_ERR_BADARG;
  if (cols == 0) return 0;
  if (!out || (rows > 0 && !X)) return YR_ERR_BADARG;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((cols + kWave - 1) / kWave)), dim3(kBlock), 0,
                     (hipStream_t)stream, X, rows, cols, out, accumulate);
  return launch_status();
}

extern "C" int yr_row_scatter_add(const float* G, const int64_t* user, int64_t B, int H, int64_t num_users, float* dV,
                                  void* stream) {
  if (B < 0 || H <= 0 || num_users <= 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!G || !user || !dV) return YR_ERR_BADARG;
  hipLaunchKernelGGL(row_scatter_add_kernel, dim3(ew_grid(B * H)), dim3(kBlock), 0, (hipStream_t)stream, G, user, B,
                     H, num_users, dV);
  return launch_status();
}

extern "C" int yr_nsbce_fwd(const float* pred, const float* target, const float* negative_mask, int64_t n,
                            float* workspace /* 2 * YR_LOSS_PARTIALS floats */, float* stats /* [2] */,
                            void* stream) {
  if (n < 0) return YR_ERR_BADARG;
  if (!workspace || !stats || (n > 0 && (!pred || !target))) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int grid = n > 0 ? ew_grid(n) : 1;
  hipLaunchKernelGGL(nsbce_fwd_kernel, dim3(grid), dim3(kBlock), 0, s, pred, target, negative_mask, n, workspace,
                     workspace + YR_LOSS_PARTIALS);
  hipLaunchKernelGGL(nsbce_finalize_kernel, dim3(1), dim3(kBlock), 0, s, workspace, workspace + YR_LOSS_PARTIALS,
                     grid, stats);
  return launch_status();
}

extern "C" int yr_nsbce_bwd(const float* pred, const float* target, const float* negative_mask, const float* stats,
                            const float* gout, int64_t n, float* dpred, void* stream) {
  if (n < 0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!pred || !target || !stats || !gout || !dpred) return YR_ERR_BADARG;
  hipLaunchKernelGGL(nsbce_bwd_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, (hipStream_t)stream, pred, target,
                     negative_mask, stats, gout, n, dpred);
  return launch_status();
}
