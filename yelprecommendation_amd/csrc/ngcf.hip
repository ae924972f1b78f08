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
//                             per pass, 8 passes in flight, next round's indices prefetched), very long rows by a whole workgroup
//   yr_ngcf_dense_fwd         E' = lrelu([Z+E | E*Z] . [W1 | W2]^T)            v_mfma_f32_32x32x2_f32, weights as
//                             the A operand: a lane ends up with 4 consecutive output columns of ONE node per
//                             accumulator quad, so the epilogues move 16 bytes per instruction (bwd_data 40 -> 34 us)
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
constexpr int kSpmmUnroll = 8;
constexpr int kSpmmHeavyBlocks = 256;

// One lane group's share of a CSR row: neighbours lo+first, lo+first+step, ...  The column/value
// pairs of the NEXT round are fetched before this round's rows are gathered (two dependent
// latencies per round otherwise), and slots past the row end issue no row load at all.
template <int D>
__device__ __forceinline__ void spmm_accumulate(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                const float* __restrict__ Xl /* X + this lane's 4 floats */, int lo,
                                                int hi, int first, int step, float4& acc) {
  int c[kSpmmUnroll];
  float w[kSpmmUnroll];
#pragma unroll
  for (int q = 0; q < kSpmmUnroll; ++q) {
    const int idx = lo + first + q * step;
    const bool ok = idx < hi;
    c[q] = ok ? col[idx] : -1;
    w[q] = ok ? val[idx] : 0.0f;
  }
  for (int base = lo; base < hi; base += step * kSpmmUnroll) {
    float4 r[kSpmmUnroll];
    float wc[kSpmmUnroll];
#pragma unroll
    for (int q = 0; q < kSpmmUnroll; ++q) {
      wc[q] = w[q];
      r[q] = c[q] >= 0 ? ngcf_ld4(Xl + (int64_t)c[q] * D) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int nbase = base + step * kSpmmUnroll;
#pragma unroll
    for (int q = 0; q < kSpmmUnroll; ++q) {
      const int idx = nbase + first + q * step;
      const bool ok = idx < hi;
      c[q] = ok ? col[idx] : -1;
      w[q] = ok ? val[idx] : 0.0f;
      }
#pragma unroll
    for (int q = 0; q < kSpmmUnroll; ++q) {
      acc.x = fmaf(wc[q], r[q].x, acc.x); acc.y = fmaf(wc[q], r[q].y, acc.y);
      acc.z = fmaf(wc[q], r[q].z, acc.z); acc.w = fmaf(wc[q], r[q].w, acc.w);
    }
  }
}

// SUB (row-subset form, yr_spmm_csr_subset): only the rows whose `row_active` flag is set are computed (the others
// are left untouched).  A computed row goes through exactly the instructions of the full form: bit-identical.
template <int D, bool ACCUM, bool SUB = false>
__global__ __launch_bounds__(kBlock) void spmm_csr_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ val,
                                                          const float* __restrict__ X, float* __restrict__ Y,
                                                          int n, const int32_t* __restrict__ heavy, int n_heavy,
                                                          int heavy_t, const int32_t* __restrict__ row_active = nullptr,
                                                          const int32_t* __restrict__ row_list = nullptr,
                                                          const int32_t* __restrict__ row_count = nullptr) {
  constexpr int LPR = D / 4, GPW = kWave / LPR;
  __shared__ float4 s_acc[kWavesPerBlock][LPR];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / LPR, l = lane % LPR;
  if ((int)blockIdx.x < kSpmmHeavyBlocks) {
    for (int h = blockIdx.x; h < n_heavy; h += kSpmmHeavyBlocks) {
      const int row = heavy[h];
      if (SUB && row_active && !row_active[row]) continue;          // workgroup-uniform
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      spmm_accumulate<D>(col, val, X + 4 * l, rowptr[row], rowptr[row + 1], wave * GPW + grp, kWavesPerBlock * GPW, acc);
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
    const int lb = blockIdx.x - kSpmmHeavyBlocks;
    const int nwaves = (gridDim.x - kSpmmHeavyBlocks) * kWavesPerBlock;
    // row_list (with row_active): the light rows come from the set's LIST, so a set of a few rows costs a few waves
    // instead of one early-exiting wave per graph row (20 us for 96 rows at Yelp2018 size)
    const int limit = (SUB && row_list) ? *row_count : n;
    for (int it = lb * kWavesPerBlock + wave; it < limit; it += nwaves) {
      int row = it;
      if (SUB && row_list) row = row_list[it];
      else if (SUB && row_active && !row_active[row]) continue;     // wave-uniform: one flag, then the next row
      const int lo = rowptr[row], hi = rowptr[row + 1];
      if (hi - lo > heavy_t) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      spmm_accumulate<D>(col, val, X + 4 * l, lo, hi, grp, GPW, acc);
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

// Feature-sliced form: the row-per-wave kernel above gathers 256-byte rows out of an 8-10 MB half table
// that no XCD's 4 MiB L2 holds (measured: 52 % L2 hits, 7x the algorithmic bytes at the fabric).  Here the
// D floats of a row are cut into slices of 16 (64 bytes); the workgroups of one XCD (equal
// blockIdx.x % 8 — a placement label: a different placement costs L2 hits, never correctness) work on
// ONE slice, so the slice of the gathered half table (38 k rows x 64 B = 2.4 MB) stays in that XCD's L2
// after first touch.  A wave owns a row of its slice: 16 lane groups of 4 lanes = 16 neighbours per pass,
// kSlicedUnroll passes in flight.  Rows are visited in `row_order` (each half by falling degree: the long
// rows start first), every XCD runs through the user half and then the item half.
#ifndef YR_SLICE_WIDTH
#define YR_SLICE_WIDTH 32
#endif
constexpr int kSliceWidth = YR_SLICE_WIDTH;        // floats per slice (32: one 128-byte line per gathered row slice;
                                                   // 16 was measured slower: twice the cache-line requests per byte)

template <int D, bool ACCUM>
__global__ __launch_bounds__(kBlock) void spmm_csr_sliced_kernel(const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ X, float* __restrict__ Y,
                                                                 int n, const int32_t* __restrict__ row_order) {
  constexpr int SW = D < kSliceWidth ? D : kSliceWidth;
  constexpr int NS = D / SW;                      // slices: 1, 2, 4
  constexpr int XPS = 8 / NS;                     // workgroup classes per slice
  constexpr int LG = SW / 4;                      // lanes per group (one slice of a row)
  constexpr int GPW = kWave / LG;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / LG, l = lane % LG;
  const int cls = blockIdx.x & 7, slice = cls % NS, part = cls / NS;
  const int cb = blockIdx.x >> 3;                 // workgroup number inside its class
  const float* Xl = X + slice * SW + 4 * l;
  const int out_off = slice * SW + 4 * l;
  {
    const int gw = cb * kWavesPerBlock + wave;
    const int nw = (gridDim.x >> 3) * kWavesPerBlock;
    for (int i = gw * XPS + part; i < n; i += nw * XPS) {
      const int row = row_order ? row_order[i] : i;
      const int lo = rowptr[row], hi = rowptr[row + 1];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      spmm_accumulate<D>(col, val, Xl, lo, hi, grp, GPW, acc);
#pragma unroll
      for (int m = LG; m < kWave; m <<= 1) {
        acc.x += __shfl_xor(acc.x, m, kWave); acc.y += __shfl_xor(acc.y, m, kWave);
        acc.z += __shfl_xor(acc.z, m, kWave); acc.w += __shfl_xor(acc.w, m, kWave);
      }
      if (grp == 0) {
        float* dst = Y + (int64_t)row * D + out_off;
        if (ACCUM) {
          const float4 old = ngcf_ld4(dst);
          acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = acc;
      }
    }
  }
}

// --------------------------------------------------------------------------- layer-sum scoring
// bpr_forward's tail (reference models/ngcf.py:44-58): the K+1 layer outputs are concatenated and
// the score is the dot product of the user's and the item's concatenated rows, i.e. the SUM over
// layers of per-layer dot products.  One launch scores positives and negatives over all layers
// (one lane group per triplet, 16 bytes per lane, every layer's three rows in flight at once);
// one launch scatter-adds the gradient of every layer buffer (float atomics, like index_add_).
struct LayerPtrs { const float* p[YR_NGCF_MAX_LAYERS]; };
struct LayerGradPtrs { float* p[YR_NGCF_MAX_LAYERS]; };

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

template <int D, bool NEG>
__global__ __launch_bounds__(kBlock) void ngcf_score_kernel(LayerPtrs L, int n_layers,
                                                            const int64_t* __restrict__ user,
                                                            const int64_t* __restrict__ pos,
                                                            const int64_t* __restrict__ neg, int64_t B,
                                                            int64_t num_users, int64_t num_items,
                                                            float* __restrict__ out_pos, float* __restrict__ out_neg,
                                                            int32_t* __restrict__ err_flag) {
  constexpr int LPR = D / 4, GPW = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / LPR, l = lane % LPR;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * GPW;
  int bad = 0;
  for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * GPW; base < B; base += stride) {
    const int64_t b = base + grp;
    int64_t u = 0, p = 0, n = 0;
    bool ok = b < B;
    if (ok) {
      u = user[b]; p = pos[b]; n = NEG ? neg[b] : 0;
      if (u < 0 || u >= num_users) { bad |= YR_FLAG_BAD_USER; ok = false; }
      if (p < 0 || p >= num_items || n < 0 || n >= num_items) { bad |= YR_FLAG_BAD_ITEM; ok = false; }
    }
    float sp = 0.0f, sn = 0.0f;
    if (ok) {
      const int64_t ou = u * D + 4 * l, op = (num_users + p) * D + 4 * l, on = (num_users + n) * D + 4 * l;
#pragma unroll 4
      for (int k = 0; k < n_layers; ++k) {
        const float* E = L.p[k];
        const float4 ru = ngcf_ld4(E + ou), rp = ngcf_ld4(E + op);
        sp += dot4(ru, rp);
        if (NEG) sn += dot4(ru, ngcf_ld4(E + on));
      }
    }
    sp = group_sum<LPR>(sp);
    if (NEG) sn = group_sum<LPR>(sn);
    if (l == 0 && b < B) {
      out_pos[b] = sp;
      if (NEG) out_neg[b] = sn;
    }
  }
  if (bad && err_flag) atomicOr(err_flag, bad);
}

// Lane l of a group owns dims l, l + LPR, l + 2 LPR, l + 3 LPR here (not 4 consecutive ones): every
// atomic instruction then covers LPR consecutive floats per row, which the memory pipeline
// coalesces into whole 64-byte requests (the 16-byte-strided form ran at a third of the rate).
template <int D, bool NEG>
__global__ __launch_bounds__(kBlock) void ngcf_score_bwd_kernel(LayerPtrs L, LayerGradPtrs G, int n_layers,
                                                                const int64_t* __restrict__ user,
                                                                const int64_t* __restrict__ pos,
                                                                const int64_t* __restrict__ neg,
                                                                const float* __restrict__ gpos,
                                                                const float* __restrict__ gneg, int64_t B,
                                                                int64_t num_users, int64_t num_items,
                                                                int32_t* __restrict__ err_flag) {
  constexpr int LPR = D / 4, GPW = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / LPR, l = lane % LPR;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * GPW;
  int bad = 0;
  for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * GPW; base < B; base += stride) {
    const int64_t b = base + grp;
    if (b >= B) continue;
    const int64_t u = user[b], p = pos[b], n = NEG ? neg[b] : 0;
    if (u < 0 || u >= num_users) { bad |= YR_FLAG_BAD_USER; continue; }
    if (p < 0 || p >= num_items || n < 0 || n >= num_items) { bad |= YR_FLAG_BAD_ITEM; continue; }
    const float gp = gpos[b], gn = NEG ? gneg[b] : 0.0f;
    const int64_t ou = u * D + l, op = (num_users + p) * D + l, on = (num_users + n) * D + l;
#pragma unroll 2
    for (int k = 0; k < n_layers; ++k) {
      const float* E = L.p[k];
      float* dE = G.p[k];
      float ru[4], rp[4], rn[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ru[j] = E[ou + j * LPR];
        rp[j] = E[op + j * LPR];
        rn[j] = NEG ? E[on + j * LPR] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        atomicAdd(dE + ou + j * LPR, NEG ? fmaf(gn, rn[j], gp * rp[j]) : gp * rp[j]);
        atomicAdd(dE + op + j * LPR, gp * ru[j]);
        if (NEG) atomicAdd(dE + on + j * LPR, gn * ru[j]);
      }
    }
  }
  if (bad && err_flag) atomicOr(err_flag, bad);
}

// --------------------------------------------------------------------------- dense part, MFMA
// Operand convention of the f32 MFMA used throughout (see csrc/eval_gemm.hip): for a 32-row
// operand, lane (i = lane & 31, h = lane >> 5) holds dims [h*K/2, (h+1)*K/2) of row i — the two
// k-slots of each v_mfma_f32_32x32x2_f32 step are mapped to the two halves of the K range.
// Output: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
constexpr float kSlope = 0.01f;   // nn.functional.leaky_relu default (models/ngcf.py:72)

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  return z;
}

// The 32-row operand tiles are fetched with coalesced 16-byte loads (whole rows per lane group) and
// handed to the MFMA register layout through a wave-private LDS tile (pitch D+4: conflict-free
// ds_read_b128); one wave per workgroup, so staging and MFMA phases of different workgroups on a
// SIMD overlap.  (Loading the register layout straight from global memory touches every 128-byte
// line with 8 separate instructions and thrashes the 32 KiB L1.)
template <int D>
struct DenseTile {
  static constexpr int PITCH = D + 4;
  static constexpr int FLOATS = 32 * PITCH;
  static constexpr int LPR = D / 4;          // lanes per staged row
  static constexpr int RPI = kWave / LPR;    // rows per load instruction
};

// SUB (row-list form, yr_ngcf_dense_*_rows): the tiles are made of the rows `rows[0 .. *count)` (node numbers, any
// order; the count lives on the device, so the host never waits for it) instead of rows 0 .. n-1, and the
// workgroups stride over the tiles.  Every output row depends on its own input row only and takes the same
// instruction sequence in both forms: bit-identical results on the listed rows.
// Full form: one tile per workgroup.  Row-list form: the workgroups stride over the tiles; the weight pointers go
// through an empty asm per tile so that the compiler does not hoist the (tile-invariant) 2 D^2 weight loads out of
// the loop into registers — that took ngcf_dense_fwd_kernel<64> from 112 to 252 VGPRs (one wave per SIMD).
#define YR_DENSE_TILES(tile, WA, WB)                                                        \
  if (!SUB) {                                                                               \
    if ((int64_t)blockIdx.x * 32 < cnt) tile((int64_t)blockIdx.x * 32, WA, WB);             \
  } else {                                                                                  \
    _Pragma("nounroll") for (int64_t row0 = (int64_t)blockIdx.x * 32; row0 < cnt;           \
                             row0 += (int64_t)gridDim.x * 32) {                             \
      const float* wa = WA;                                                                 \
      const float* wb = WB;                                                                 \
      asm volatile("" : "+s"(wa), "+s"(wb));                                                \
      tile(row0, wa, wb);                                                                   \
      __syncthreads(); /* the staged tile is free for the next one */                       \
    }                                                                                       \
  }

template <int D, bool SUB>
__device__ __forceinline__ int64_t dense_row(const int32_t* __restrict__ rows, int64_t idx, int cnt) {
  return idx < cnt ? (SUB ? (int64_t)rows[idx] : idx) : -1;
}

template <int D, bool SUB = false>
__global__ __launch_bounds__(kWave) void ngcf_dense_fwd_kernel(const float* __restrict__ E,
                                                               const float* __restrict__ Z,
                                                               const float* __restrict__ W1,
                                                               const float* __restrict__ W2, int n,
                                                               float* __restrict__ Eout,
                                                               const int32_t* __restrict__ rows = nullptr,
                                                               const int32_t* __restrict__ count = nullptr,
                                                               float* __restrict__ zero_rows = nullptr) {
  using T = DenseTile<D>;
  constexpr int HALF = D / 2;
  constexpr int CT = (D + 31) / 32;          // 32-column output tiles
  __shared__ float s_a[T::FLOATS];
  __shared__ float s_h[T::FLOATS];
  const int lane = threadIdx.x;
  const int i = lane & 31, h = lane >> 5;
  const int cnt = SUB ? *count : n;
  auto tile = [&](const int64_t row0, const float* __restrict__ W1, const float* __restrict__ W2) {
  const int64_t out_row = dense_row<D, SUB>(rows, row0 + i, cnt);
  {
    const int c4 = (lane % T::LPR) * 4;
#pragma unroll
    for (int rr = 0; rr < 32 / T::RPI; ++rr) {
      const int r = rr * T::RPI + lane / T::LPR;
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f), z = e;
      const int64_t pr = dense_row<D, SUB>(rows, row0 + r, cnt);
      if (pr >= 0) {
        e = ngcf_ld4(E + pr * D + c4);
        z = ngcf_ld4(Z + pr * D + c4);
      }
      *reinterpret_cast<float4*>(s_a + r * T::PITCH + c4) = make_float4(z.x + e.x, z.y + e.y, z.z + e.z, z.w + e.w);
      *reinterpret_cast<float4*>(s_h + r * T::PITCH + c4) = make_float4(e.x * z.x, e.y * z.y, e.z * z.z, e.w * z.w);
    }
  }
  __syncthreads();
  float aA[HALF], aH[HALF];
#pragma unroll
  for (int q = 0; q < HALF / 4; ++q) {
    const float4 a = *reinterpret_cast<const float4*>(s_a + i * T::PITCH + h * HALF + 4 * q);
    const float4 m = *reinterpret_cast<const float4*>(s_h + i * T::PITCH + h * HALF + 4 * q);
    aA[4 * q + 0] = a.x; aA[4 * q + 1] = a.y; aA[4 * q + 2] = a.z; aA[4 * q + 3] = a.w;
    aH[4 * q + 0] = m.x; aH[4 * q + 1] = m.y; aH[4 * q + 2] = m.z; aH[4 * q + 3] = m.w;
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
    for (int s = 0; s < HALF; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], aA[s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 w = j < D ? ngcf_ld4(W2 + j * D + h * HALF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[4 * q + 0] = w.x; b[4 * q + 1] = w.y; b[4 * q + 2] = w.z; b[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], aH[s], acc, 0, 0, 0);
    // the WEIGHTS are the A operand: lane (i, h) holds, for node row0 + i, the output columns
    // 32 t + 8 g + 4 h + {0..3} in registers 4 g .. 4 g + 3 — four consecutive floats, one 16-byte store each
    if (out_row >= 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = t * 32 + 8 * g + 4 * h;
        if (c < D) {
          float4 o;
          o.x = acc[4 * g + 0] > 0.0f ? acc[4 * g + 0] : kSlope * acc[4 * g + 0];
          o.y = acc[4 * g + 1] > 0.0f ? acc[4 * g + 1] : kSlope * acc[4 * g + 1];
          o.z = acc[4 * g + 2] > 0.0f ? acc[4 * g + 2] : kSlope * acc[4 * g + 2];
          o.w = acc[4 * g + 3] > 0.0f ? acc[4 * g + 3] : kSlope * acc[4 * g + 3];
          *reinterpret_cast<float4*>(Eout + out_row * D + c) = o;
          // row-list form: the same rows of the layer's GRADIENT buffer are cleared here (the backward pass touches
          // no others), instead of a memset of the whole buffer
          if (SUB && zero_rows) *reinterpret_cast<float4*>(zero_rows + out_row * D + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
  };
  YR_DENSE_TILES(tile, W1, W2)
}

// dP = dEout * lrelu'(Eout);  [dA | dH] = dP . [W1 | W2]  (W1T/W2T = transposed weights, [in, out]);
// dZ = dA + dH * E;  dE += dA + dH * Z
template <int D, bool SUB = false>
__global__ __launch_bounds__(kWave) void ngcf_dense_bwd_data_kernel(
    const float* __restrict__ dEout, const float* __restrict__ Eout, const float* __restrict__ E,
    const float* __restrict__ Z, const float* __restrict__ W1T, const float* __restrict__ W2T, int n,
    float* __restrict__ dZ, float* __restrict__ dE, const int32_t* __restrict__ rows = nullptr,
    const int32_t* __restrict__ count = nullptr) {
  using T = DenseTile<D>;
  constexpr int HALF = D / 2;
  constexpr int CT = (D + 31) / 32;
  __shared__ float s_p[T::FLOATS];
  const int lane = threadIdx.x;
  const int i = lane & 31, h = lane >> 5;
  const int cnt = SUB ? *count : n;
  auto tile = [&](const int64_t row0, const float* __restrict__ W1T, const float* __restrict__ W2T) {
  const int64_t out_row = dense_row<D, SUB>(rows, row0 + i, cnt);
  {
    const int c4 = (lane % T::LPR) * 4;
#pragma unroll
    for (int rr = 0; rr < 32 / T::RPI; ++rr) {
      const int r = rr * T::RPI + lane / T::LPR;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), o = g;
      const int64_t pr = dense_row<D, SUB>(rows, row0 + r, cnt);
      if (pr >= 0) {
        g = ngcf_ld4(dEout + pr * D + c4);
        o = ngcf_ld4(Eout + pr * D + c4);
      }
      *reinterpret_cast<float4*>(s_p + r * T::PITCH + c4) =
          make_float4(o.x > 0.0f ? g.x : kSlope * g.x, o.y > 0.0f ? g.y : kSlope * g.y,
                      o.z > 0.0f ? g.z : kSlope * g.z, o.w > 0.0f ? g.w : kSlope * g.w);
    }
  }
  __syncthreads();
  float a[HALF];
#pragma unroll
  for (int q = 0; q < HALF / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(s_p + i * T::PITCH + h * HALF + 4 * q);
    a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
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
    for (int s = 0; s < HALF; ++s) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], accA, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 w = c < D ? ngcf_ld4(W2T + c * D + h * HALF + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      b[4 * q + 0] = w.x; b[4 * q + 1] = w.y; b[4 * q + 2] = w.z; b[4 * q + 3] = w.w;
    }
#pragma unroll
    for (int s = 0; s < HALF; ++s) accH = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], accH, 0, 0, 0);
    // weights as the A operand (see the forward kernel): lane (i, h) holds columns 32 t + 8 g + 4 h + {0..3} of
    // node row0 + i in registers 4 g .. 4 g + 3 — 16-byte loads and stores instead of 4-byte ones
    if (out_row >= 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = t * 32 + 8 * g + 4 * h;
        if (cc < D) {
          const int64_t o = out_row * D + cc;
          const float4 e = ngcf_ld4(E + o), z = ngcf_ld4(Z + o);
          float4 de = *reinterpret_cast<const float4*>(dE + o);
          float4 dz;
          dz.x = accA[4 * g + 0] + accH[4 * g + 0] * e.x;
          dz.y = accA[4 * g + 1] + accH[4 * g + 1] * e.y;
          dz.z = accA[4 * g + 2] + accH[4 * g + 2] * e.z;
          dz.w = accA[4 * g + 3] + accH[4 * g + 3] * e.w;
          de.x += accA[4 * g + 0] + accH[4 * g + 0] * z.x;
          de.y += accA[4 * g + 1] + accH[4 * g + 1] * z.y;
          de.z += accA[4 * g + 2] + accH[4 * g + 2] * z.z;
          de.w += accA[4 * g + 3] + accH[4 * g + 3] * z.w;
          *reinterpret_cast<float4*>(dZ + o) = dz;
          *reinterpret_cast<float4*>(dE + o) = de;
        }
      }
    }
  }
  };
  YR_DENSE_TILES(tile, W1T, W2T)
}

// dW1[j, c] += sum_r dP[r, j] (Z+E)[r, c];  dW2[j, c] += sum_r dP[r, j] (E*Z)[r, c]
// One workgroup per chunk of WChunk<D>::ROWS rows: dP, A = Z+E and H = E*Z of the chunk are staged in LDS
// (pitch D+1: the column reads below are conflict-free) and every wave computes whole 32x32
// output tiles over K = ROWS with one ds_read_b32 per operand per MFMA.
template <int D>
struct WChunk {
  static constexpr int ROWS = 4096 / D;      // rows per workgroup: 3 staged tiles stay under 64 KiB of LDS
};
template <int D, bool SUB = false>
__global__ __launch_bounds__(kBlock) void ngcf_dense_bwd_weight_kernel(
    const float* __restrict__ dEout, const float* __restrict__ Eout, const float* __restrict__ E,
    const float* __restrict__ Z, int n_all, float* __restrict__ dW1, float* __restrict__ dW2,
    const int32_t* __restrict__ rows = nullptr, const int32_t* __restrict__ count = nullptr) {
  const int n = SUB ? *count : n_all;         // rows to sum over: the list's length in the row-list form
  // pitch D: an MFMA operand read is 32 consecutive floats of one staged row per half-wave, which is
  // conflict-free at any pitch, and a multiple of 4 keeps the staging stores 16 bytes wide
  constexpr int PITCH = D;
  constexpr int kWRows = WChunk<D>::ROWS;
  constexpr int RT = (D + 31) / 32;          // tiles along j (rows of dW) and along c (per matrix)
  constexpr int NT = RT * RT * 2;            // output tiles: RT x RT for dW1, same for dW2
  constexpr int NV = kWRows * D / 4 / kBlock;   // float4 per thread per operand and chunk (= 4)
  if (SUB && (int64_t)blockIdx.x * kWRows >= n) return;   // row-list form: the grid is sized for an upper bound
  __shared__ __attribute__((aligned(16))) float s_dp[kWRows * PITCH];
  __shared__ __attribute__((aligned(16))) float s_a[kWRows * PITCH];
  __shared__ __attribute__((aligned(16))) float s_h[kWRows * PITCH];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  constexpr int TPW = (NT + kWavesPerBlock - 1) / kWavesPerBlock;   // output tiles per wave
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = zero16();
  // the next chunk's rows travel in registers while the MFMAs of the current chunk run
  float4 pg[NV], po[NV], pe[NV], pz[NV];
  auto fetch = [&](int64_t row0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = threadIdx.x + v * kBlock;
      const int64_t r = row0 + q / (D / 4);
      const int c4 = (q % (D / 4)) * 4;
      pg[v] = po[v] = pe[v] = pz[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < n) {
        const int64_t o = (SUB ? (int64_t)rows[r] : r) * D + c4;
        pg[v] = ngcf_ld4(dEout + o); po[v] = ngcf_ld4(Eout + o); pe[v] = ngcf_ld4(E + o); pz[v] = ngcf_ld4(Z + o);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = threadIdx.x + v * kBlock;
      const int at = (q / (D / 4)) * PITCH + (q % (D / 4)) * 4;
      const float4 g = pg[v], o = po[v], e = pe[v], z = pz[v];
      *reinterpret_cast<float4*>(s_dp + at) = make_float4(o.x > 0.0f ? g.x : kSlope * g.x, o.y > 0.0f ? g.y : kSlope * g.y,
                                                          o.z > 0.0f ? g.z : kSlope * g.z, o.w > 0.0f ? g.w : kSlope * g.w);
      *reinterpret_cast<float4*>(s_a + at) = make_float4(z.x + e.x, z.y + e.y, z.z + e.z, z.w + e.w);
      *reinterpret_cast<float4*>(s_h + at) = make_float4(e.x * z.x, e.y * z.y, e.z * z.z, e.w * z.w);
    }
  };
  // persistent over row chunks: the 2 D^2 partial sums stay in registers until the end, so the
  // float atomics on dW scale with the grid, not with n
  const int64_t stride = (int64_t)gridDim.x * kWRows;
  int64_t row0 = (int64_t)blockIdx.x * kWRows;
  if (row0 < n) fetch(row0);
  for (; row0 < n; row0 += stride) {
    __syncthreads();                                 // the previous chunk's MFMAs are done with the LDS tiles
    stash();
    __syncthreads();
    if (row0 + stride < n) fetch(row0 + stride);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + t * kWavesPerBlock;
      if (tile < NT) {
        const int which = tile / (RT * RT);      // 0: dW1 (A), 1: dW2 (H)
        const int tj = (tile % (RT * RT)) / RT, tc = tile % RT;
        const float* s_b = which ? s_h : s_a;
        const int j = tj * 32 + i, c = tc * 32 + i;
#pragma unroll 8
        for (int s = 0; s < kWRows / 2; ++s) {
          const int r = h * (kWRows / 2) + s;
          const float av = j < D ? s_dp[r * PITCH + j] : 0.0f;
          const float bv = c < D ? s_b[r * PITCH + c] : 0.0f;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave + t * kWavesPerBlock;
    if (tile < NT) {
      const int which = tile / (RT * RT);
      const int tj = (tile % (RT * RT)) / RT, tc = tile % RT;
      const int c = tc * 32 + i;
      float* out = which ? dW2 : dW1;
      if (c < D) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int jj = tj * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (jj < D) atomicAdd(out + jj * D + c, acc[t][reg]);
        }
      }
    }
  }
}

// Push form of Y += L X for a FEW source rows (the backward product of a layer whose dZ lives on the batch's rows
// only): a wave per listed row r scatters val(r, j) * X[r] into Y[j] for every neighbour j with float atomics — L is
// symmetric, so this is the pull product restricted to the columns in the list, at a cost proportional to the
// list's non-zeros instead of the graph's.  Lane l of a group owns dims l, l + LPR, ... (every atomic instruction
// covers LPR consecutive floats per row, see ngcf_score_bwd_kernel).
constexpr int kPushParts = 16;     // workgroups per listed row: a popular item's row (thousands of non-zeros) is
                                   // cut into this many slices, or one wave would walk it alone (measured: 128 us
                                   // for 96 rows with a wave per row)
template <int D>
__global__ __launch_bounds__(kBlock) void spmm_push_rows_kernel(const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col,
                                                                const float* __restrict__ val,
                                                                const float* __restrict__ X, float* __restrict__ Y,
                                                                const int32_t* __restrict__ rows,
                                                                const int32_t* __restrict__ count) {
  constexpr int LPR = D / 4, GPW = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / LPR, l = lane % LPR;
  const int64_t work = (int64_t)(*count) * kPushParts;
  for (int64_t w = blockIdx.x; w < work; w += gridDim.x) {
    const int row = rows[w / kPushParts], part = (int)(w % kPushParts);
    const int lo = rowptr[row], len = rowptr[row + 1] - lo;
    const int a = lo + (int)((int64_t)len * part / kPushParts), b = lo + (int)((int64_t)len * (part + 1) / kPushParts);
    if (a == b) continue;
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = X[(int64_t)row * D + l + j * LPR];
    for (int idx = a + wave * GPW + grp; idx < b; idx += kWavesPerBlock * GPW) {
      const float wv = val[idx];
      float* dst = Y + (int64_t)col[idx] * D + l;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dst + j * LPR, wv * x[j]);
    }
  }
}

// --------------------------------------------------------------------------- frontier of a batch
// The scores of a batch read layer K at the rows R_K = {u} + {U + p} + {U + n} only (models/ngcf.py:37-39), layer
// K-1 is then needed at R_K and its neighbours, and so on: S_K = R_K, S_{k-1} = S_k + N(S_k).  Flags are one byte
// per node; the row list of a flag set is compacted with one counter atomic per wave (any order: every row is
// computed on its own), its length stays on the device.
// Wave-aggregated append of the lanes that flipped their node's flag 0 -> 1: one counter atomic per wave.
__device__ __forceinline__ void frontier_append(bool won, int node, int32_t* __restrict__ rows,
                                                int32_t* __restrict__ count) {
  const unsigned long long m = __ballot(won);
  if (m == 0) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int leader = __ffsll((long long)m) - 1;
  int at = 0;
  if (lane == leader) at = atomicAdd(count, __popcll(m));
  at = __shfl(at, leader, kWave);
  if (won) rows[at + __popcll(m & ((1ull << lane) - 1ull))] = node;
}

__global__ __launch_bounds__(kBlock) void ngcf_frontier_mark_kernel(const int64_t* __restrict__ user,
                                                                    const int64_t* __restrict__ pos,
                                                                    const int64_t* __restrict__ neg, int64_t B,
                                                                    int64_t num_users, int64_t num_items,
                                                                    int32_t* __restrict__ flags,
                                                                    int32_t* __restrict__ rows,
                                                                    int32_t* __restrict__ count) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t b0 = (int64_t)blockIdx.x * kBlock; b0 < B; b0 += stride) {       // wave-uniform trip count
    const int64_t b = b0 + threadIdx.x;
    const bool in = b < B;
    const int64_t u = in ? user[b] : -1, p = in ? pos[b] : -1, q = (in && neg) ? neg[b] : -1;
    // out-of-range ids are skipped here: the score kernel raises the flag
    const bool ou = u >= 0 && u < num_users, op = p >= 0 && p < num_items, oq = q >= 0 && q < num_items;
    const int nu_ = (int)u, np_ = (int)(num_users + p), nq_ = (int)(num_users + q);
    frontier_append(ou && atomicExch(flags + (ou ? nu_ : 0), 1) == 0, nu_, rows, count);
    frontier_append(op && atomicExch(flags + (op ? np_ : 0), 1) == 0, np_, rows, count);
    frontier_append(oq && atomicExch(flags + (oq ? nq_ : 0), 1) == 0, nq_, rows, count);
  }
}

// out = in + neighbours(in).  A listed row is cut into kExpandParts slices, one workgroup each (a popular item's row has
// thousands of neighbours: with a wave per row the 96 rows of a 32-triplet batch took 8-10 us).
constexpr int kExpandParts = 8;
__global__ __launch_bounds__(kBlock) void ngcf_frontier_expand_kernel(const int32_t* __restrict__ rowptr,
                                                                      const int32_t* __restrict__ col,
                                                                      const int32_t* __restrict__ rows_in,
                                                                      const int32_t* __restrict__ count_in,
                                                                      int32_t* __restrict__ flags,
                                                                      int32_t* __restrict__ rows,
                                                                      int32_t* __restrict__ count) {
  const int64_t work = (int64_t)(*count_in) * kExpandParts;
  for (int64_t w = blockIdx.x; w < work; w += gridDim.x) {
    const int row = rows_in[w / kExpandParts], part = (int)(w % kExpandParts);
    if (part == 0) frontier_append(threadIdx.x == 0 && atomicExch(flags + row, 1) == 0, row, rows, count);
    const int lo = rowptr[row], len = rowptr[row + 1] - lo;
    const int a = lo + (int)((int64_t)len * part / kExpandParts), b = lo + (int)((int64_t)len * (part + 1) / kExpandParts);
    for (int base = a; base < b; base += kBlock) {                              // workgroup-uniform trip count
      const int idx = base + threadIdx.x;
      const int j = idx < b ? col[idx] : -1;
      frontier_append(j >= 0 && atomicExch(flags + (j >= 0 ? j : 0), 1) == 0, j, rows, count);
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
  if (light > 65536) light = 65536;                    // one row per wave; the row loop covers larger graphs
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

extern "C" int yr_spmm_csr_subset(const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                                  float* Y, int64_t n, int D, int accumulate, const int32_t* heavy_rows,
                                  int64_t n_heavy, int heavy_threshold, const int32_t* row_active,
                                  const int32_t* row_list, const int32_t* row_count, int64_t max_rows,
                                  void* stream) {
  if (n < 0 || n > 0x7fffffff || n_heavy < 0 || n_heavy > n) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!rowptr || !X || !Y || X == Y) return YR_ERR_BADARG;
  if (n_heavy > 0 && !heavy_rows) return YR_ERR_BADARG;
  if (row_list && (!row_active || !row_count || max_rows < 0 || max_rows > n)) return YR_ERR_BADARG;
  if (heavy_threshold <= 0 || n_heavy == 0) heavy_threshold = n_heavy > 0 ? 256 : 0x7fffffff;
  int light = (int)(((row_list ? max_rows : n) + kWavesPerBlock - 1) / kWavesPerBlock);
  if (light > 65536) light = 65536;
  if (light < 1) light = 1;
  const int grid = kSpmmHeavyBlocks + light;
  hipStream_t s = (hipStream_t)stream;
  if (accumulate) {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_csr_kernel<kD, true, true>), dim3(grid), dim3(kBlock), 0, s, rowptr,
                                           col, val, X, Y, (int)n, heavy_rows, (int)n_heavy, heavy_threshold,
                                           row_active, row_list, row_count));
  } else {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_csr_kernel<kD, false, true>), dim3(grid), dim3(kBlock), 0, s, rowptr,
                                           col, val, X, Y, (int)n, heavy_rows, (int)n_heavy, heavy_threshold,
                                           row_active, row_list, row_count));
  }
  return launch_status();
}

extern "C" int yr_spmm_csr_push_rows(const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                                     float* Y, int64_t n, int D, const int32_t* rows, const int32_t* count,
                                     int64_t max_rows, void* stream) {
  if (n < 0 || n > 0x7fffffff || max_rows < 0 || max_rows > n) return YR_ERR_BADARG;
  if (n == 0 || max_rows == 0) return 0;
  if (!rowptr || !col || !val || !X || !Y || X == Y || !rows || !count) return YR_ERR_BADARG;
  int64_t g = max_rows * kPushParts;
  const int grid = (int)(g > 32768 ? 32768 : g);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_push_rows_kernel<kD>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                                         rowptr, col, val, X, Y, rows, count));
  return launch_status();
}

extern "C" int yr_ngcf_frontier_mark(const int64_t* user, const int64_t* pos, const int64_t* neg, int64_t B,
                                     int64_t num_users, int64_t num_items, int32_t* flags, int32_t* rows,
                                     int32_t* count, int clear, void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || num_users + num_items > 0x7fffffff || !flags || !rows || !count)
    return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (clear) {
    if (hipMemsetAsync(flags, 0, (size_t)(num_users + num_items) * 4, s) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(count, 0, 4, s) != hipSuccess) return (int)hipGetLastError();
  }
  if (B == 0) return 0;
  if (!user || !pos) return YR_ERR_BADARG;
  hipLaunchKernelGGL(ngcf_frontier_mark_kernel, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, s, user, pos, neg, B,
                     num_users, num_items, flags, rows, count);
  return launch_status();
}

extern "C" int yr_ngcf_frontier_expand(const int32_t* rowptr, const int32_t* col, int64_t n, const int32_t* rows_in,
                                       const int32_t* count_in, int64_t max_rows_in, int32_t* flags, int32_t* rows,
                                       int32_t* count, int clear, void* stream) {
  if (n < 0 || n > 0x7fffffff || max_rows_in < 0 || max_rows_in > n) return YR_ERR_BADARG;
  if (!flags || !rows || !count) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (clear) {
    if (n && hipMemsetAsync(flags, 0, (size_t)n * 4, s) != hipSuccess) return (int)hipGetLastError();
    if (hipMemsetAsync(count, 0, 4, s) != hipSuccess) return (int)hipGetLastError();
  }
  if (n == 0 || max_rows_in == 0) return 0;
  if (!rowptr || !col || !rows_in || !count_in || rows_in == rows) return YR_ERR_BADARG;
  const int64_t g = max_rows_in * kExpandParts;
  const int grid = (int)(g > 16384 ? 16384 : g);
  hipLaunchKernelGGL(ngcf_frontier_expand_kernel, dim3(grid), dim3(kBlock), 0, s, rowptr, col, rows_in, count_in, flags,
                     rows, count);
  return launch_status();
}

extern "C" int yr_spmm_csr_sliced(const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                                  float* Y, int64_t n, int D, int accumulate, const int32_t* row_order,
                                  void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!rowptr || !X || !Y || X == Y) return YR_ERR_BADARG;
  // eight workgroup classes (one per XCD); one (row, slice) per wave while that stays below 2^19
  // workgroups, the row loop covers larger graphs
  const int sw = D < kSliceWidth ? D : kSliceWidth;
  const int xps = 8 / (D / sw);
  const int64_t per_class = (n + xps * kWavesPerBlock - 1) / (xps * kWavesPerBlock);
  const int grid = 8 * (int)(per_class > 65536 ? 65536 : per_class);
  hipStream_t s = (hipStream_t)stream;
  if (accumulate) {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_csr_sliced_kernel<kD, true>), dim3(grid), dim3(kBlock), 0, s, rowptr,
                                           col, val, X, Y, (int)n, row_order));
  } else {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((spmm_csr_sliced_kernel<kD, false>), dim3(grid), dim3(kBlock), 0, s, rowptr,
                                           col, val, X, Y, (int)n, row_order));
  }
  return launch_status();
}

static int score_grid(int64_t B, int D) {
  const int64_t per_block = (int64_t)kWavesPerBlock * (kWave / (D / 4));
  int64_t g = (B + per_block - 1) / per_block;
  return (int)(g > 8192 ? 8192 : g);
}

extern "C" int yr_ngcf_score_fwd(const float* const* layers, int n_layers, const int64_t* user, const int64_t* pos,
                                 const int64_t* neg, int64_t B, int D, int64_t num_users, int64_t num_items,
                                 float* out_pos, float* out_neg, int32_t* err_flag, void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || n_layers <= 0 || n_layers > YR_NGCF_MAX_LAYERS) return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  if (B == 0) return 0;
  if (!layers || !user || !pos || !out_pos || (neg != nullptr) != (out_neg != nullptr)) return YR_ERR_BADARG;
  LayerPtrs L{};
  for (int k = 0; k < n_layers; ++k) {
    if (!layers[k]) return YR_ERR_BADARG;
    L.p[k] = layers[k];
  }
  const int grid = score_grid(B, D);
  hipStream_t s = (hipStream_t)stream;
  if (neg) {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_score_kernel<kD, true>), dim3(grid), dim3(kBlock), 0, s, L, n_layers,
                                           user, pos, neg, B, num_users, num_items, out_pos, out_neg, err_flag));
  } else {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_score_kernel<kD, false>), dim3(grid), dim3(kBlock), 0, s, L, n_layers,
                                           user, pos, neg, B, num_users, num_items, out_pos, out_neg, err_flag));
  }
  return launch_status();
}

extern "C" int yr_ngcf_score_bwd(const float* const* layers, float* const* dlayers, int n_layers, const int64_t* user,
                                 const int64_t* pos, const int64_t* neg, const float* gpos, const float* gneg,
                                 int64_t B, int D, int64_t num_users, int64_t num_items, int32_t* err_flag,
                                 void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || n_layers <= 0 || n_layers > YR_NGCF_MAX_LAYERS) return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  if (B == 0) return 0;
  if (!layers || !dlayers || !user || !pos || !gpos || (neg != nullptr) != (gneg != nullptr)) return YR_ERR_BADARG;
  LayerPtrs L{};
  LayerGradPtrs G{};
  for (int k = 0; k < n_layers; ++k) {
    if (!layers[k] || !dlayers[k]) return YR_ERR_BADARG;
    L.p[k] = layers[k];
    G.p[k] = dlayers[k];
  }
  const int grid = score_grid(B, D);
  hipStream_t s = (hipStream_t)stream;
  if (neg) {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_score_bwd_kernel<kD, true>), dim3(grid), dim3(kBlock), 0, s, L, G,
                                           n_layers, user, pos, neg, gpos, gneg, B, num_users, num_items, err_flag));
  } else {
    YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_score_bwd_kernel<kD, false>), dim3(grid), dim3(kBlock), 0, s, L, G,
                                           n_layers, user, pos, neg, gpos, gneg, B, num_users, num_items, err_flag));
  }
  return launch_status();
}

extern "C" int yr_ngcf_dense_fwd(const float* E, const float* Z, const float* W1, const float* W2, int64_t n, int D,
                                 float* Eout, void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!E || !Z || !W1 || !W2 || !Eout) return YR_ERR_BADARG;
  const int grid = (int)((n + 31) / 32);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_fwd_kernel<kD>), dim3(grid), dim3(kWave), 0,
                                         (hipStream_t)stream, E, Z, W1, W2, (int)n, Eout));
  return launch_status();
}

extern "C" int yr_ngcf_dense_bwd_data(const float* dEout, const float* Eout, const float* E, const float* Z,
                                      const float* W1T, const float* W2T, int64_t n, int D, float* dZ, float* dE,
                                      void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!dEout || !Eout || !E || !Z || !W1T || !W2T || !dZ || !dE) return YR_ERR_BADARG;
  const int grid = (int)((n + 31) / 32);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_bwd_data_kernel<kD>), dim3(grid), dim3(kWave), 0,
                                         (hipStream_t)stream, dEout, Eout, E, Z, W1T, W2T, (int)n, dZ, dE));
  return launch_status();
}

extern "C" int yr_ngcf_dense_bwd_weight(const float* dEout, const float* Eout, const float* E, const float* Z,
                                        int64_t n, int D, float* dW1, float* dW2, void* stream) {
  if (n < 0 || n > 0x7fffffff) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!dEout || !Eout || !E || !Z || !dW1 || !dW2) return YR_ERR_BADARG;
  YR_NGCF_DISPATCH(D, {
    const int64_t chunks = (n + WChunk<kD>::ROWS - 1) / WChunk<kD>::ROWS;
    const int64_t per_block = (chunks + 511) / 512;            // <= 512 workgroups, equal shares
    hipLaunchKernelGGL((ngcf_dense_bwd_weight_kernel<kD>), dim3((unsigned)((chunks + per_block - 1) / per_block)),
                       dim3(kBlock), 0, (hipStream_t)stream, dEout, Eout, E, Z, (int)n, dW1, dW2);
  });
  return launch_status();
}

// Row-list forms: the same kernels over the rows `rows[0 .. *count)`; `max_rows` (<= n) is the host's upper bound of
// *count and only sizes the grid — the workgroups stride, so any count up to n is covered.
static int rows_args_ok(int64_t n, const int32_t* rows, const int32_t* count, int64_t max_rows) {
  return n >= 0 && n <= 0x7fffffff && rows && count && max_rows >= 0 && max_rows <= n;
}

extern "C" int yr_ngcf_dense_fwd_rows(const float* E, const float* Z, const float* W1, const float* W2, int64_t n,
                                      int D, float* Eout, const int32_t* rows, const int32_t* count,
                                      int64_t max_rows, float* zero_rows, void* stream) {
  if (!rows_args_ok(n, rows, count, max_rows)) return YR_ERR_BADARG;
  if (n == 0 || max_rows == 0) return 0;
  if (!E || !Z || !W1 || !W2 || !Eout) return YR_ERR_BADARG;
  const int grid = (int)((max_rows + 31) / 32);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_fwd_kernel<kD, true>), dim3(grid), dim3(kWave), 0,
                                         (hipStream_t)stream, E, Z, W1, W2, (int)n, Eout, rows, count, zero_rows));
  return launch_status();
}

extern "C" int yr_ngcf_dense_bwd_data_rows(const float* dEout, const float* Eout, const float* E, const float* Z,
                                           const float* W1T, const float* W2T, int64_t n, int D, float* dZ,
                                           float* dE, const int32_t* rows, const int32_t* count, int64_t max_rows,
                                           void* stream) {
  if (!rows_args_ok(n, rows, count, max_rows)) return YR_ERR_BADARG;
  if (n == 0 || max_rows == 0) return 0;
  if (!dEout || !Eout || !E || !Z || !W1T || !W2T || !dZ || !dE) return YR_ERR_BADARG;
  const int grid = (int)((max_rows + 31) / 32);
  YR_NGCF_DISPATCH(D, hipLaunchKernelGGL((ngcf_dense_bwd_data_kernel<kD, true>), dim3(grid), dim3(kWave), 0,
                                         (hipStream_t)stream, dEout, Eout, E, Z, W1T, W2T, (int)n, dZ, dE, rows, count));
  return launch_status();
}

extern "C" int yr_ngcf_dense_bwd_weight_rows(const float* dEout, const float* Eout, const float* E, const float* Z,
                                             int64_t n, int D, float* dW1, float* dW2, const int32_t* rows,
                                             const int32_t* count, int64_t max_rows, void* stream) {
  if (!rows_args_ok(n, rows, count, max_rows)) return YR_ERR_BADARG;
  if (n == 0 || max_rows == 0) return 0;
  if (!dEout || !Eout || !E || !Z || !dW1 || !dW2) return YR_ERR_BADARG;
  YR_NGCF_DISPATCH(D, {
    const int64_t chunks = (max_rows + WChunk<kD>::ROWS - 1) / WChunk<kD>::ROWS;
    const int64_t per_block = (chunks + 511) / 512;
    hipLaunchKernelGGL((ngcf_dense_bwd_weight_kernel<kD, true>), dim3((unsigned)((chunks + per_block - 1) / per_block)),
                       dim3(kBlock), 0, (hipStream_t)stream, dEout, Eout, E, Z, (int)n, dW1, dW2, rows, count);
  });
  return launch_status();
}
