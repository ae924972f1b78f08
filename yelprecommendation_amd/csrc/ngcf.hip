// NGCF message passing for gfx950 (MI355X): normalised-Laplacian SpMM + the layer's dense part.
//
// Replaces reference models/ngcf.py:60-72 (embedding_propagation)
//     E' = leaky_relu( W1((L + I) E) + W2(E * (L E)) )
// and its autograd.  The reference builds eye(N, N).to_sparse() per layer per batch (O(N^2)) and
// runs two torch.sparse.mm on COO; here (L + I)E = LE + E, ONE CSR SpMM per layer serves both
// terms, and the dense part is a float32 MFMA GEMM with the element-wise pieces fused in.
//
//   yr_spmm_csr               Z = L X  (or Z += L X)   HBM/cache bound gather, pull form: every
//                             output row is owned by one wave (16-byte loads, 64/(D/4) neighbours
//                             per pass, 4 passes in flight), very long rows by a whole workgroup
//   yr_ngcf_dense_fwd         E' = lrelu([Z+E | E*Z] . [W1 | W2]^T)            v_mfma_f32_32x32x2_f32
//   yr_ngcf_dense_bwd_data    dP = dE' * lrelu'(E');  [dA | dH] = dP . [W1 | W2];
//                             dZ = dA + dH*E;  dE += dA + dH*Z                 (MFMA + fused epilogue)
//   yr_ngcf_dense_bwd_weight  dW1 += dP^T (Z+E);  dW2 += dP^T (E*Z)            (MFMA over row chunks
//                             staged in LDS, float atomics on the 2*D*D outputs)
// The SpMM of the backward pass (dE += L^T dZ) is the same kernel: L is symmetric by construction
// (D^-1/2 A D^-1/2 with A = [[0,R],[R^T,0]], reference data/datasets/ngcf_data_pipeline.py:23-42).
#include "common.h"

namespace yr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float4 ngcf_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// --------------------------------------------------------------------------- SpMM
constexpr int kSpmmUnroll = 4;
constexpr int kSpmmHeavyBlocks = 256;

template <int D>
__device__ __forceinline__ void spmm_accumulate(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                const float* __restrict__ X, int lo, int hi, int first, int step,
                                                int l, float4& acc) {
  for (int base = lo; base < hi; base += step * kSpmmUnroll) {
    float4 r[kSpmmUnroll];
    float w[kSpmmUnroll];
#pragma unroll
    for (int q = 0; q < kSpmmUnroll; ++q) {
      const int idx = base + first + q * step;
      const bool ok = idx < hi;
      const int c = ok ? col[idx] : 0;
      w[q] = ok ? val[idx] : 0.0f;
      r[q] = ngcf_ld4(X + (int64_t)c * D + 4 * l);
    }
#pragma unroll
    for (int q = 0; q < kSpmmUnroll; ++q) {
      acc.x = fmaf(w[q], r[q].x, acc.x); acc.y = fmaf(w[q], r[q].y, acc.y);
      acc.z = fmaf(w[q], r[q].z, acc.z); acc.w = fmaf(w[q], r[q].w, acc.w);
    }
  }
}

template <int D, bool ACCUM>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ val,
                                                          const float* __restrict__ X, float* __restrict__ Y,
                                                          int n, const int32_t* __restrict__ heavy, int n_heavy,
                                                          int heavy_t) {
  constexpr int LPR = D / 4, GPW = kWave / LPR;
  __shared__ float4 s_acc[kWavesPerBlock][LPR];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / LPR, l = lane % LPR;
  if ((int)blockIdx.x < kSpmmHeavyBlocks) {
    for (int h = blockIdx.x; h < n_heavy; h += kSpmmHeavyBlocks) {
      const int row = heavy[h];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      spmm_accumulate<D>(col, val, X, rowptr[row], rowptr[row + 1], wave * GPW + grp, kWavesPerBlock * GPW, l, acc);
#pragma unroll
      for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
      }
      if (grp == 0) s_acc[wave][l] = acc;
      __syncthreads();
      if (wave == 0 && grp == 0) {
        float4 t = s_acc[0][l];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) {
          const float4 o = s_acc[w][l];
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        float* dst = Y + (int64_t)row * D + 4 * l;
        if (ACCUM) {
          const float4 old = ngcf_ld4(dst);
          t.x += old.x; t.y += old.y; t.z += old.z; t.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = t;
      }
      __syncthreads();
    }
  } else {
    const int nwaves = (gridDim.x - kSpmmHeavyBlocks) * kWavesPerBlock;
    for (int row = (blockIdx.x - kSpmmHeavyBlocks) * kWavesPerBlock + wave; row < n; row += nwaves) {
      const int lo = rowptr[row], hi = rowptr[row + 1];
      if (hi - lo > heavy_t) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      spmm_accumulate<D>(col, val, X, lo, hi, grp, GPW, l, acc);
#pragma unroll
      for (int m = LPR; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
      }
      if (grp == 0) {
        float* dst = Y + (int64_t)row * D + 4 * l;
        if (ACCUM) {
          const float4 old = ngcf_ld4(dst);
          acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = acc;
      }
    }
  }
}

// --------------------------------------------------------------------------- dense part, MFMA
// Operand convention of the f32 MFMA used throughout (see csrc/eval_gemm.hip): for a 32-row
// operand, lane (i = lane & 31, h = lane >> 5) holds dims [h*K/2, (h+1)*K/2) of row i — the two
// k-slots of each v_mfma_f32_32x32x2_f32 step are mapped to the two halves of the K range.
// Output: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
constexpr int kDenseRows = 128;   // rows per workgroup (32 per wave)
constexpr float kSlope = 0.01f;   // nn.functional.leaky_relu default (models/ngcf.py:72)

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}

template <int D>
__global__ __launch_bounds__(kBlock) void ngcf_dense_fwd_kernel(const float* __restrict__ E,
                                                                const float* __restrict__ Z,
                                                                const float* __restrict__ W1,
                                                                const float* __restrict__ W2, int n,
                                                                float* __restrict__ Eout) {
  constexpr int HALF = D / 2;
  constexpr int CT = (D + 31) / 32;          // 32-column output tiles
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * kDenseRows + wave * 32 + i;
  float aA[HALF], aH[HALF];
#pragma unroll
  for (int q = 0; q < HALF / 4; ++q) {
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f), z = e;
    if (row < n) {
      e = ngcf_ld4(E + (int64_t)row * D + h * HALF + 4 * q);
      z = ngcf_ld4(Z + (int64_t)row * D + h * HALF + 4 * q);
    }
    aA[4 * q + 0] = z.x + e.x; aA[4 * q + 1] = z.y + e.y; aA[4 * q + 2] = z.z + e.z; aA[4 * q + 3] = z.w + e.w;
    aH[4 * q + 0] = e.x * z.x; aH[4 * q + 1] = e.y * z.y; aH[4 * q + 2] = e.z * z.z; aH[4 * q + 3] = e.w * z.w;
  }
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int j = t * 32 + i;               // output column = row j of W1 / W2 ([out, in])
    f32x16 acc = zero16();
    float b[HALF];
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 w = j < D ? ngcf_ld4(W1 + j * D + h * HALF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[4 * q + 0] = w.x; b[4 * q + 1] = w.y; b[4 * q + 2] = w.z; b[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aA[s], b[s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 w = j < D ? ngcf_ld4(W2 + j * D + h * HALF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[4 * q + 0] = w.x; b[4 * q + 1] = w.y; b[4 * q + 2] = w.z; b[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aH[s], b[s], acc, 0, 0, 0);
    if (j < D) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = blockIdx.x * kDenseRows + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (r < n) {
          const float p = acc[reg];
          Eout[(int64_t)r * D + j] = p > 0.0f ? p : kSlope * p;
        }
      }
    }
  }
}

// dP = dEout * lrelu'(Eout);  [dA | dH] = dP . [W1 | W2]  (W1T/W2T = transposed weights, [in, out]);
// dZ = dA + dH * E;  dE += dA + dH * Z
template <int D>
__global__ __launch_bounds__(kBlock) void ngcf_dense_bwd_data_kernel(
    const float* __restrict__ dEout, const float* __restrict__ Eout, const float* __restrict__ E,
    const float* __restrict__ Z, const float* __restrict__ W1T, const float* __restrict__ W2T, int n,
    float* __restrict__ dZ, float* __restrict__ dE) {
  constexpr int HALF = D / 2;
  constexpr int CT = (D + 31) / 32;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int row = blockIdx.x * kDenseRows + wave * 32 + i;
  float a[HALF];
#pragma unroll
  for (int q = 0; q < HALF / 4; ++q) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), o = g;
    if (row < n) {
      g = ngcf_ld4(dEout + (int64_t)row * D + h * HALF + 4 * q);
      o = ngcf_ld4(Eout + (int64_t)row * D + h * HALF + 4 * q);
    }
    a[4 * q + 0] = o.x > 0.0f ? g.x : kSlope * g.x; a[4 * q + 1] = o.y > 0.0f ? g.y : kSlope * g.y;
    a[4 * q + 2] = o.z > 0.0f ? g.z : kSlope * g.z; a[4 * q + 3] = o.w > 0.0f ? g.w : kSlope * g.w;
  }
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = t * 32 + i;               // output column c = row c of W^T
    f32x16 accA = zero16(), accH = zero16();
    float b[HALF];
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 w = c < D ? ngcf_ld4(W1T + c * D + h * HALF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[4 * q + 0] = w.x; b[4 * q + 1] = w.y; b[4 * q + 2] = w.z; b[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], accA, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 w = c < D ? ngcf_ld4(W2T + c * D + h * HALF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[4 * q + 0] = w.x; b[4 * q + 1] = w.y; b[4 * q + 2] = w.z; b[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) accH = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], accH, 0, 0, 0);
    if (c < D) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = blockIdx.x * kDenseRows + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (r < n) {
          const int64_t o = (int64_t)r * D + c;
          const float e = E[o], z = Z[o];
          dZ[o] = accA[reg] + accH[reg] * e;
          dE[o] += accA[reg] + accH[reg] * z;
        }
      }
    }
  }
}

// dW1[j, c] += sum_r dP[r, j] (Z+E)[r, c];  dW2[j, c] += sum_r dP[r, j] (E*Z)[r, c]
// One workgroup per chunk of WChunk<D>::ROWS rows: dP, A = Z+E and H = E*Z of the chunk are staged in LDS
// (pitch D+1: the column reads below are conflict-free) and every wave computes whole 32x32
// output tiles over K = ROWS with one ds_read_b32 per operand per MFMA.
template <int D>
struct WChunk {
  static constexpr int ROWS = 4096 / D;      // rows per workgroup: 3 staged tiles stay under 64 KiB of LDS
};
template <int D>
__global__ __launch_bounds__(kBlock) void ngcf_dense_bwd_weight_kernel(
    const float* __restrict__ dEout, const float* __restrict__ Eout, const float* __restrict__ E,
    const float* __restrict__ Z, int n, float* __restrict__ dW1, float* __restrict__ dW2) {
  constexpr int PITCH = D + 1;
  constexpr int kWRows = WChunk<D>::ROWS;
  constexpr int RT = (D + 31) / 32;          // tiles along j (rows of dW) and along c (per matrix)
  constexpr int NT = RT * RT * 2;            // output tiles: RT x RT for dW1, same for dW2
  __shared__ float s_dp[kWRows * PITCH];
  __shared__ float s_a[kWRows * PITCH];
  __shared__ float s_h[kWRows * PITCH];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * kWRows;
  for (int q = threadIdx.x; q < kWRows * D; q += kBlock) {
    const int r = q / D, c = q % D;
    float dp = 0.f, av = 0.f, hv = 0.f;
    if (row0 + r < n) {
      const int64_t o = (int64_t)(row0 + r) * D + c;
      const float g = dEout[o], eo = Eout[o], e = E[o], z = Z[o];
      dp = eo > 0.0f ? g : kSlope * g;
      av = z + e;
      hv = e * z;
    }
    s_dp[r * PITCH + c] = dp;
    s_a[r * PITCH + c] = av;
    s_h[r * PITCH + c] = hv;
  }
  __syncthreads();
  for (int tile = wave; tile < NT; tile += kWavesPerBlock) {
    const int which = tile / (RT * RT);      // 0: dW1 (A), 1: dW2 (H)
    const int tj = (tile % (RT * RT)) / RT, tc = tile % RT;
    const float* s_b = which ? s_h : s_a;
    const int j = tj * 32 + i, c = tc * 32 + i;
    f32x16 acc = zero16();
#pragma unroll 8
    for (int s = 0; s < kWRows / 2; ++s) {
      const int r = h * (kWRows / 2) + s;
      const float av = j < D ? s_dp[r * PITCH + j] : 0.0f;
      const float bv = c < D ? s_b[r * PITCH + c] : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    float* out = which ? dW2 : dW1;
    if (c < D) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int jj = tj * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (jj < D) atomicAdd(out + jj * D + c, acc[reg]);
      }
    }
  }
}

}  // namespace yr

using namespace yr;

#define YR_NGCF_DISPATCH(D, ...)                               \
  switch (D) {                                                 \
    case 16: { constexpr int kD = 16; __VA_ARGS__; } break;    \
    case 32: { constexpr int kD = 32; __VA_ARGS__; } break;    \
    case 64: { constexpr int kD = 64; __VA_ARGS__; } break;    \
    case 128: { constexpr int kD = 128; __VA_ARGS__; } break;  \
    default: return YR_ERR_UNSUPPORTED;                        \
  }

extern "C" int yr_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y,
                           int64_t n, int D, int accumulate, const int32_t* heavy_rows, int64_t n_heavy,
                           int heavy_threshold, void* stream) {
  if (n < 0 || n > 0x7fffffff || n_heavy < 0 || n_heavy > n) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!rowptr || !X || !Y || X == Y) return YR_ERR_BADARG;
  if (n_heavy > 0 && !heavy_rows) return YR_ERR_BADARG;
  if (heavy_threshold <= 0 || n_heavy == 0) heavy_threshold = n_heavy > 0 ? 256 : 0x7fffffff;
  int light = (int)((n + kWavesPerBlock - 1) / kWavesPerBlock);
  if (light > 4096) light = 4096;
  const int grid = kSpmmHeavyBlocks + light;
  hipStream_t s = (hipStream_t)stream;
  if (accumulate) {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_csr_kernel<kD, true>), dim3(grid), dim3(kBlock), 0, s, rowptr, col,
                                           val, X, Y, (int)n, heavy_rows, (int)n_heavy, heavy_threshold));
  } else {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_csr_kernel<kD, false>), dim3(grid), dim3(kBlock), 0, s, rowptr, col,
                                           val, X, Y, (int)n, heavy_rows, (int)n_heavy, heavy_threshold));
  }
  return launch_status();
}

extern "C" int yr_ngcf_dense_fwd(const float* E, const float* Z, const float* W1, const float* W2, int64_t n, int D,
                                 float* Eout, void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!E || !Z || !W1 || !W2 || !Eout) return YR_ERR_BADARG;
  const int grid = (int)((n + kDenseRows - 1) / kDenseRows);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_fwd_kernel<kD>), dim3(grid), dim3(kBlock), 0,
                                         (hipStream_t)stream, E, Z, W1, W2, (int)n, Eout));
  return launch_status();
}

extern "C" int yr_ngcf_dense_bwd_data(const float* dEout, const float* Eout, const float* E, const float* Z,
                                      const float* W1T, const float* W2T, int64_t n, int D, float* dZ, float* dE,
                                      void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!dEout || !Eout || !E || !Z || !W1T || !W2T || !dZ || !dE) return YR_ERR_BADARG;
  const int grid = (int)((n + kDenseRows - 1) / kDenseRows);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_bwd_data_kernel<kD>), dim3(grid), dim3(kBlock), 0,
                                         (hipStream_t)stream, dEout, Eout, E, Z, W1T, W2T, (int)n, dZ, dE));
  return launch_status();
}

extern "C" int yr_ngcf_dense_bwd_weight(const float* dEout, const float* Eout, const float* E, const float* Z,
                                        int64_t n, int D, float* dW1, float* dW2, void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!dEout || !Eout || !E || !Z || !dW1 || !dW2) return YR_ERR_BADARG;
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_bwd_weight_kernel<kD>),
                                         dim3((unsigned)((n + WChunk<kD>::ROWS - 1) / WChunk<kD>::ROWS)),
                                         dim3(kBlock), 0, (hipStream_t)stream, dEout, Eout, E, Z, (int)n, dW1,
                                         dW2));
  return launch_status();
}
