// Dense optimizer steps for gfx950: one streaming pass, 16 bytes per lane.
//
// Replaces torch.optim.Adam / AdamW / SGD .step() as built by the reference's
// trainers/base_trainer.py:34-43 (dense gradients: nn.Embedding(sparse=False), so
// every row moves every step — a lazy/sparse update would NOT be equivalent).
// Formula = torch/optim/adam.py::_single_tensor_adam (see oracle/adam.py).
// Pure HBM-bound elementwise work: read p,g,m,v, write p,m,v (+ g = 0).
#include "common.h"

namespace yr {

struct AdamScalars {
  float decay_mul, neg_step, bc2_sqrt, one_m_b1, beta2, one_m_b2, eps, wd;
};

template <bool DECOUPLED>
__device__ __forceinline__ void adam_element(float& p, float grad, float& m, float& v, const AdamScalars& c) {
  if (c.wd != 0.0f) {
    if (DECOUPLED) p *= c.decay_mul;
    else grad = grad + c.wd * p;
  }
  m = m + c.one_m_b1 * (grad - m);               // lerp_
  v = v * c.beta2 + (c.one_m_b2 * grad) * grad;   // mul_, addcmul_
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p + (c.neg_step * m) / denom;               // addcdiv_
}

// main body: 16 bytes per lane; elements [4*n4, n) are the scalar tail of block 0
template <bool DECOUPLED, bool ZERO_GRAD>
__global__ __launch_bounds__(kBlock) void adam_dense_kernel(float* __restrict__ p, float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            int64_t n4, int64_t n, AdamScalars c) {
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 P = p4[i], G = g4[i], M = m4[i], V = v4[i];
    adam_element<DECOUPLED>(P.x, G.x, M.x, V.x, c);
    adam_element<DECOUPLED>(P.y, G.y, M.y, V.y, c);
    adam_element<DECOUPLED>(P.z, G.z, M.z, V.z, c);
    adam_element<DECOUPLED>(P.w, G.w, M.w, V.w, c);
    p4[i] = P;
    m4[i] = M;
    v4[i] = V;
    if (ZERO_GRAD) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0) {
    const int64_t i = 4 * n4 + threadIdx.x;
    if (i < n) {
      adam_element<DECOUPLED>(p[i], g[i], m[i], v[i], c);
      if (ZERO_GRAD) g[i] = 0.0f;
    }
  }
}

// Several small tensors in one launch (blockIdx.y = tensor): the weight matrices and biases next
// to an embedding table cost a launch each otherwise, ~5 us apiece for a few KB of work.
struct AdamMulti {
  float* p[YR_ADAM_MULTI_MAX];
  float* g[YR_ADAM_MULTI_MAX];
  float* m[YR_ADAM_MULTI_MAX];
  float* v[YR_ADAM_MULTI_MAX];
  int64_t n[YR_ADAM_MULTI_MAX];
};

template <bool DECOUPLED, bool ZERO_GRAD>
__global__ __launch_bounds__(kBlock) void adam_dense_multi_kernel(AdamMulti t, AdamScalars c) {
  const int k = blockIdx.y;
  float* p = t.p[k]; float* g = t.g[k]; float* m = t.m[k]; float* v = t.v[k];
  const int64_t n = t.n[k];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float P = p[i], G = g[i], M = m[i], V = v[i];
    adam_element<DECOUPLED>(P, G, M, V, c);
    p[i] = P; m[i] = M; v[i] = V;
    if (ZERO_GRAD) g[i] = 0.0f;
  }
}

// Two tensors in one launch (the user and the item table), 16 bytes per lane, plus the loss reduction
// of the step: replaces two adam_dense launches and loss_finalize.  With `touched` (one byte per row
// of `row4` float4, set by the scatter kernel) the gradient of a row is read and cleared only where
// the batch touched it — every row still gets its Adam update (dense semantics, grad = 0).
struct DualAdam {
  float4 *p0, *g0, *m0, *v0, *p1, *g1, *m1, *v1;
  uint8_t *touched0, *touched1;
  int64_t n4_0, n4_1;
  int row4;
};

template <bool DECOUPLED>
__global__ __launch_bounds__(kBlock) void adam_dual_kernel(DualAdam t, AdamScalars c,
                                                           const float* __restrict__ partials, float loss_scale,
                                                           float* __restrict__ loss_out,
                                                           double* __restrict__ loss_accum) {
  const int64_t total = t.n4_0 + t.n4_1;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
    const bool second = i >= t.n4_0;
    const int64_t j = second ? i - t.n4_0 : i;
    float4* pp = second ? t.p1 : t.p0;
    float4* gp = second ? t.g1 : t.g0;
    float4* mp = second ? t.m1 : t.m0;
    float4* vp = second ? t.v1 : t.v0;
    uint8_t* tp = second ? t.touched1 : t.touched0;
    float4 P = pp[j], M = mp[j], V = vp[j];
    float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t row = tp ? j / t.row4 : 0;
    const bool has = tp ? tp[row] != 0 : true;
    if (has) {
      G = gp[j];
      gp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tp && j % t.row4 == 0) tp[row] = 0;     // after every lane of the row (same wave) has read the mark
    }
    adam_element<DECOUPLED>(P.x, G.x, M.x, V.x, c);
    adam_element<DECOUPLED>(P.y, G.y, M.y, V.y, c);
    adam_element<DECOUPLED>(P.z, G.z, M.z, V.z, c);
    adam_element<DECOUPLED>(P.w, G.w, M.w, V.w, c);
    pp[j] = P;
    mp[j] = M;
    vp[j] = V;
  }
  if (partials && blockIdx.x == 0) {              // fixed-order loss reduction (partials come from the previous launch)
    __shared__ float s_red[kWavesPerBlock];
    float s = 0.0f;
    for (int i = threadIdx.x; i < YR_LOSS_PARTIALS; i += kBlock) s += partials[i];
    const float tot = block_sum(s, s_red);
    if (threadIdx.x == 0) {
      const float v = tot * loss_scale;
      if (loss_out) loss_out[0] = v;
      if (loss_accum) loss_accum[0] += (double)v;
    }
  }
}

// Up to YR_ADAM_MULTI_MAX tensors of any size in ONE launch, 16 bytes per lane (every buffer 16-byte aligned;
// the 1-3 elements after the last whole float4 of a tensor are done by one lane).  The work is cut into chunks of kBlock float4 that never straddle two tensors,
// so the tensor a workgroup iteration works on is wave-uniform (its pointers stay in scalar registers).
// Per tensor:
//   touched[k] != NULL  one byte per row of row4[k] float4: the gradient of a row is read — and cleared, with
//                       its mark — only where the step touched it (dense Adam semantics: every row is updated,
//                       with grad = 0 where nothing arrived); the gradient buffer stays all-zero between steps;
//   clear[k]            1: the gradient is cleared after it is read (a buffer the next step accumulates into);
//                       2: only where it is non-zero (a mostly-zero buffer: no marks to load first, and the
//                       lines that hold nothing are not written).
struct AdamFlat {
  float4* p[YR_ADAM_MULTI_MAX];
  float4* g[YR_ADAM_MULTI_MAX];
  float4* m[YR_ADAM_MULTI_MAX];
  float4* v[YR_ADAM_MULTI_MAX];
  uint8_t* touched[YR_ADAM_MULTI_MAX];
  int64_t n4[YR_ADAM_MULTI_MAX];
  int64_t chunk_end[YR_ADAM_MULTI_MAX];  // running end of tensor k in chunks
  int row4[YR_ADAM_MULTI_MAX];           // log2 of the float4 per marked row
  int clear[YR_ADAM_MULTI_MAX];
  int tail[YR_ADAM_MULTI_MAX];           // n % 4
  int scaled[YR_ADAM_MULTI_MAX];         // gradient is multiplied by 1 / grad_count before use
  const int32_t* grad_count;             // spread count (YR_COUNT_SLOTS) or NULL
  int count;
};

template <bool DECOUPLED>
__global__ __launch_bounds__(kBlock) void adam_flat_kernel(AdamFlat t, AdamScalars c) {
  const int64_t chunks = t.chunk_end[t.count - 1];
  float inv_count = 1.0f;
  if (t.grad_count) {
    const int32_t n = spread_count(t.grad_count, threadIdx.x & (kWave - 1));
    inv_count = n > 0 ? 1.0f / (float)n : 0.0f;
  }
  for (int64_t ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
    int k = 0;
    for (int q = 0; q + 1 < t.count; ++q)
      if (ch >= t.chunk_end[q]) k = q + 1;
    const int64_t j = (ch - (k > 0 ? t.chunk_end[k - 1] : 0)) * kBlock + threadIdx.x;
    if (j >= t.n4[k]) {
      if (j == t.n4[k] && t.tail[k]) {                // the 1-3 elements after the last whole float4
        float* p = reinterpret_cast<float*>(t.p[k] + j);
        float* g = reinterpret_cast<float*>(t.g[k] + j);
        float* m = reinterpret_cast<float*>(t.m[k] + j);
        float* v = reinterpret_cast<float*>(t.v[k] + j);
        const float gs = t.scaled[k] ? inv_count : 1.0f;
        for (int e = 0; e < t.tail[k]; ++e) {
          adam_element<DECOUPLED>(p[e], g[e] * gs, m[e], v[e], c);
          if (t.clear[k]) g[e] = 0.0f;
        }
      }
      continue;
    }
    float4* gp = t.g[k];
    uint8_t* tp = t.touched[k];
    const int shift = t.row4[k];                      // log2(float4 per row)
    float4 P = t.p[k][j], M = t.m[k][j], V = t.v[k][j];
    float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t row = j >> shift;
    const bool has = tp ? tp[row] != 0 : true;
    if (has) {
      G = gp[j];
      const int clr = t.clear[k];
      if (tp || clr == 1 || (clr == 2 && (G.x != 0.f || G.y != 0.f || G.z != 0.f || G.w != 0.f)))
        gp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tp && (j & ((1 << shift) - 1)) == 0) tp[row] = 0;   // after every lane of the row (same wave) has read the mark
    }
    if (t.scaled[k]) { G.x *= inv_count; G.y *= inv_count; G.z *= inv_count; G.w *= inv_count; }
    adam_element<DECOUPLED>(P.x, G.x, M.x, V.x, c);
    adam_element<DECOUPLED>(P.y, G.y, M.y, V.y, c);
    adam_element<DECOUPLED>(P.z, G.z, M.z, V.z, c);
    adam_element<DECOUPLED>(P.w, G.w, M.w, V.w, c);
    t.p[k][j] = P;
    t.m[k][j] = M;
    t.v[k][j] = V;
  }
}

template <bool ZERO_GRAD>
__global__ __launch_bounds__(kBlock) void sgd_dense_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           int64_t n4, int64_t n, float lr, float wd) {
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 P = p4[i], G = g4[i];
    float* pp = reinterpret_cast<float*>(&P);
    float* gg = reinterpret_cast<float*>(&G);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float grad = gg[k];
      if (wd != 0.0f) grad = grad + wd * pp[k];
      pp[k] = pp[k] + (-lr) * grad;
    }
    p4[i] = P;
    if (ZERO_GRAD) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0) {
    const int64_t i = 4 * n4 + threadIdx.x;
    if (i < n) {
      float grad = g[i];
      if (wd != 0.0f) grad = grad + wd * p[i];
      p[i] = p[i] + (-lr) * grad;
      if (ZERO_GRAD) g[i] = 0.0f;
    }
  }
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace yr

using namespace yr;

extern "C" int yr_adam_dense(float* p, float* g, float* m, float* v, int64_t n, double lr, double step_size,
                             double bc2_sqrt, double beta1, double beta2, double eps, double weight_decay,
                             int mode, int zero_grad, void* stream) {
  if (n < 0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!p || !g || !m || !v) return YR_ERR_BADARG;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  // vector body only when all four buffers are 16-byte aligned; otherwise all-scalar tail is
  // impossible (tail <= 255 elements), so misaligned buffers are rejected
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return YR_ERR_BADARG;
  const int64_t n4 = n / 4;
  const int grid = grid_for(n4 > 0 ? n4 : 1, kBlock);
  hipStream_t s = (hipStream_t)stream;
  AdamScalars c;
  c.decay_mul = (float)(1.0 - lr * weight_decay);
  c.neg_step = (float)(-step_size);
  c.bc2_sqrt = (float)bc2_sqrt;
  c.one_m_b1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.one_m_b2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  c.wd = (float)weight_decay;
#define YR_LAUNCH_ADAM(DEC, ZG) \
  hipLaunchKernelGGL((adam_dense_kernel<DEC, ZG>), dim3(grid), dim3(kBlock), 0, s, p, g, m, v, n4, n, c)
  if (mode == YR_OPT_ADAMW) {
    if (zero_grad) YR_LAUNCH_ADAM(true, true); else YR_LAUNCH_ADAM(true, false);
  } else {
    if (zero_grad) YR_LAUNCH_ADAM(false, true); else YR_LAUNCH_ADAM(false, false);
  }
#undef YR_LAUNCH_ADAM
  return launch_status();
}

extern "C" int yr_adam_dense_dual(float* p0, float* g0, float* m0, float* v0, int64_t n0, float* p1, float* g1,
                                  float* m1, float* v1, int64_t n1, int row_width, uint8_t* touched0,
                                  uint8_t* touched1, double lr, double step_size, double bc2_sqrt, double beta1,
                                  double beta2, double eps, double weight_decay, int mode,
                                  const float* loss_partials, float loss_scale, float* loss_out, double* loss_accum,
                                  void* stream) {
  if (n0 < 0 || n1 < 0 || (n0 & 3) || (n1 & 3)) return YR_ERR_BADARG;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  if ((n0 > 0 && (!p0 || !g0 || !m0 || !v0)) || (n1 > 0 && (!p1 || !g1 || !m1 || !v1))) return YR_ERR_BADARG;
  if (!aligned16(p0) || !aligned16(g0) || !aligned16(m0) || !aligned16(v0) || !aligned16(p1) || !aligned16(g1) ||
      !aligned16(m1) || !aligned16(v1))
    return YR_ERR_BADARG;
  // a row's mark is read by all row_width / 4 lanes of the row and cleared by the first of them: the lanes of a
  // row must sit in ONE wave (row_width / 4 divides 64), as yr_adam_dense_flat requires
  if ((touched0 || touched1) && (row_width <= 0 || (row_width & 3) || n0 % row_width || n1 % row_width ||
                                 row_width / 4 > kWave || kWave % (row_width / 4)))
    return YR_ERR_BADARG;
  DualAdam t;
  t.p0 = (float4*)p0; t.g0 = (float4*)g0; t.m0 = (float4*)m0; t.v0 = (float4*)v0;
  t.p1 = (float4*)p1; t.g1 = (float4*)g1; t.m1 = (float4*)m1; t.v1 = (float4*)v1;
  t.touched0 = touched0; t.touched1 = touched1;
  t.n4_0 = n0 / 4; t.n4_1 = n1 / 4;
  t.row4 = row_width > 0 ? row_width / 4 : 1;
  AdamScalars c;
  c.decay_mul = (float)(1.0 - lr * weight_decay);
  c.neg_step = (float)(-step_size);
  c.bc2_sqrt = (float)bc2_sqrt;
  c.one_m_b1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.one_m_b2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  c.wd = (float)weight_decay;
  const int64_t total = t.n4_0 + t.n4_1;
  const int grid = grid_for(total > 0 ? total : 1, kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (mode == YR_OPT_ADAMW)
    hipLaunchKernelGGL((adam_dual_kernel<true>), dim3(grid), dim3(kBlock), 0, s, t, c, loss_partials, loss_scale,
                       loss_out, loss_accum);
  else
    hipLaunchKernelGGL((adam_dual_kernel<false>), dim3(grid), dim3(kBlock), 0, s, t, c, loss_partials, loss_scale,
                       loss_out, loss_accum);
  return launch_status();
}

extern "C" int yr_adam_dense_multi(float* const* p, float* const* g, float* const* m, float* const* v,
                                   const int64_t* n, int count, double lr, double step_size, double bc2_sqrt,
                                   double beta1, double beta2, double eps, double weight_decay, int mode,
                                   int zero_grad, void* stream) {
  if (count < 0 || count > YR_ADAM_MULTI_MAX) return YR_ERR_BADARG;
  if (count == 0) return 0;
  if (!p || !g || !m || !v || !n) return YR_ERR_BADARG;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  AdamMulti t{};
  int64_t longest = 0;
  for (int k = 0; k < count; ++k) {
    if (n[k] < 0 || (n[k] > 0 && (!p[k] || !g[k] || !m[k] || !v[k]))) return YR_ERR_BADARG;
    t.p[k] = p[k]; t.g[k] = g[k]; t.m[k] = m[k]; t.v[k] = v[k]; t.n[k] = n[k];
    if (n[k] > longest) longest = n[k];
  }
  if (longest == 0) return 0;
  AdamScalars c;
  c.decay_mul = (float)(1.0 - lr * weight_decay);
  c.neg_step = (float)(-step_size);
  c.bc2_sqrt = (float)bc2_sqrt;
  c.one_m_b1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.one_m_b2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  c.wd = (float)weight_decay;
  int gx = grid_for(longest, kBlock);
  if (gx > 1024) gx = 1024;
  const dim3 grid((unsigned)gx, (unsigned)count);
  hipStream_t s = (hipStream_t)stream;
#define YR_LAUNCH_ADAM_MULTI(DEC, ZG) \
  hipLaunchKernelGGL((adam_dense_multi_kernel<DEC, ZG>), grid, dim3(kBlock), 0, s, t, c)
  if (mode == YR_OPT_ADAMW) {
    if (zero_grad) YR_LAUNCH_ADAM_MULTI(true, true); else YR_LAUNCH_ADAM_MULTI(true, false);
  } else {
    if (zero_grad) YR_LAUNCH_ADAM_MULTI(false, true); else YR_LAUNCH_ADAM_MULTI(false, false);
  }
#undef YR_LAUNCH_ADAM_MULTI
  return launch_status();
}

extern "C" int yr_adam_dense_flat(float* const* p, float* const* g, float* const* m, float* const* v,
                                  const int64_t* n, uint8_t* const* touched, const int* row_width, const int* clear,
                                  const int* scaled, const int32_t* grad_count, int count, double lr, double step_size, double bc2_sqrt, double beta1, double beta2,
                                  double eps, double weight_decay, int mode, void* stream) {
  if (count < 0 || count > YR_ADAM_MULTI_MAX) return YR_ERR_BADARG;
  if (count == 0) return 0;
  if (!p || !g || !m || !v || !n) return YR_ERR_BADARG;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  AdamFlat t{};
  int64_t chunks = 0;
  int used = 0;
  for (int k = 0; k < count; ++k) {
    if (n[k] < 0) return YR_ERR_BADARG;
    if (n[k] == 0) continue;
    if (!p[k] || !g[k] || !m[k] || !v[k]) return YR_ERR_BADARG;
    if (!aligned16(p[k]) || !aligned16(g[k]) || !aligned16(m[k]) || !aligned16(v[k])) return YR_ERR_BADARG;
    uint8_t* tp = touched ? touched[k] : nullptr;
    const int rw = row_width ? row_width[k] : 0;
    // the lanes of a marked row must sit in one wave (they all read the mark before one of them clears it)
    if (tp && (rw <= 0 || (rw & 3) || n[k] % rw || rw / 4 > kWave || kWave % (rw / 4))) return YR_ERR_BADARG;
    t.p[used] = (float4*)p[k]; t.g[used] = (float4*)g[k]; t.m[used] = (float4*)m[k]; t.v[used] = (float4*)v[k];
    t.touched[used] = tp;
    int lg = 0;
    while (tp && (1 << lg) < rw / 4) ++lg;
    t.row4[used] = lg;
    t.clear[used] = clear ? clear[k] : 0;
    t.scaled[used] = (scaled && grad_count) ? scaled[k] : 0;
    t.n4[used] = n[k] / 4;
    t.tail[used] = (int)(n[k] & 3);
    chunks += (n[k] / 4 + (t.tail[used] ? 1 : 0) + kBlock - 1) / kBlock;   // one more lane for the tail
    t.chunk_end[used] = chunks;
    ++used;
  }
  if (used == 0) return 0;
  t.count = used;
  t.grad_count = grad_count;
  AdamScalars c;
  c.decay_mul = (float)(1.0 - lr * weight_decay);
  c.neg_step = (float)(-step_size);
  c.bc2_sqrt = (float)bc2_sqrt;
  c.one_m_b1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.one_m_b2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  c.wd = (float)weight_decay;
  const int grid = (int)(chunks < kMaxGrid ? chunks : kMaxGrid);
  hipStream_t s = (hipStream_t)stream;
  if (mode == YR_OPT_ADAMW) hipLaunchKernelGGL((adam_flat_kernel<true>), dim3(grid), dim3(kBlock), 0, s, t, c);
  else hipLaunchKernelGGL((adam_flat_kernel<false>), dim3(grid), dim3(kBlock), 0, s, t, c);
  return launch_status();
}

extern "C" int yr_sgd_dense(float* p, float* g, int64_t n, double lr, double weight_decay, int zero_grad,
                            void* stream) {
  if (n < 0) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!p || !g || !aligned16(p) || !aligned16(g)) return YR_ERR_BADARG;
  const int64_t n4 = n / 4;
  const int grid = grid_for(n4 > 0 ? n4 : 1, kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (zero_grad)
    hipLaunchKernelGGL((sgd_dense_kernel<true>), dim3(grid), dim3(kBlock), 0, s, p, g, n4, n, (float)lr,
                       (float)weight_decay);
  else
    hipLaunchKernelGGL((sgd_dense_kernel<false>), dim3(grid), dim3(kBlock), 0, s, p, g, n4, n, (float)lr,
                       (float)weight_decay);
  return launch_status();
}
