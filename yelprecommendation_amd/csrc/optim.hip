// Dense optimizer steps for gfx950: one streaming pass, 16 bytes per lane.
//
// Replaces torch.optim.Adam / AdamW / SGD .step() as built by the reference's
// trainers/base_trainer.py:34-43 (dense gradients: nn.Embedding(sparse=False), so
// every row moves every step — a lazy/sparse update would NOT be equivalent).
// Formula = torch/optim/adam.py::_single_tensor_adam (see oracle/adam.py).
// Pure HBM-bound elementwise work: read p,g,m,v, write p,m,v (+ g = 0).
#include "common.h"

namespace yr {

template <bool DECOUPLED, bool ZERO_GRAD>
__global__ __launch_bounds__(kBlock) void adam_dense_kernel(float4* __restrict__ p, float4* __restrict__ g,
                                                            float4* __restrict__ m, float4* __restrict__ v,
                                                            int64_t n4, float decay_mul, float neg_step,
                                                            float bc2_sqrt, float one_m_b1, float beta2,
                                                            float one_m_b2, float eps, float wd) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 P = p[i], G = g[i], M = m[i], V = v[i];
    float* pp = reinterpret_cast<float*>(&P);
    float* gg = reinterpret_cast<float*>(&G);
    float* mm = reinterpret_cast<float*>(&M);
    float* vv = reinterpret_cast<float*>(&V);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float grad = gg[k];
      if (wd != 0.0f) {
        if (DECOUPLED) pp[k] *= decay_mul;
        else grad = grad + wd * pp[k];
      }
      mm[k] = mm[k] + one_m_b1 * (grad - mm[k]);            // lerp_
      vv[k] = vv[k] * beta2 + (one_m_b2 * grad) * grad;      // mul_, addcmul_
      const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
      pp[k] = pp[k] + (neg_step * mm[k]) / denom;            // addcdiv_
    }
    p[i] = P;
    m[i] = M;
    v[i] = V;
    if (ZERO_GRAD) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <bool ZERO_GRAD>
__global__ __launch_bounds__(kBlock) void sgd_dense_kernel(float4* __restrict__ p, float4* __restrict__ g,
                                                           int64_t n4, float lr, float wd) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 P = p[i], G = g[i];
    float* pp = reinterpret_cast<float*>(&P);
    float* gg = reinterpret_cast<float*>(&G);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float grad = gg[k];
      if (wd != 0.0f) grad = grad + wd * pp[k];
      pp[k] = pp[k] + (-lr) * grad;
    }
    p[i] = P;
    if (ZERO_GRAD) g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace yr

using namespace yr;

extern "C" int yr_adam_dense(float* p, float* g, float* m, float* v, int64_t n, double lr, double step_size,
                             double bc2_sqrt, double beta1, double beta2, double eps, double weight_decay,
                             int mode, int zero_grad, void* stream) {
  if (n < 0 || (n & 3)) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!p || !g || !m || !v) return YR_ERR_BADARG;
  if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return YR_ERR_BADARG;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  const int64_t n4 = n / 4;
  const int grid = grid_for(n4, kBlock);
  hipStream_t s = (hipStream_t)stream;
#define YR_LAUNCH_ADAM(DEC, ZG)                                                                              \
  hipLaunchKernelGGL((adam_dense_kernel<DEC, ZG>), dim3(grid), dim3(kBlock), 0, s, (float4*)p, (float4*)g,   \
                     (float4*)m, (float4*)v, n4, (float)(1.0 - lr * weight_decay), (float)(-step_size),          \
                     (float)bc2_sqrt, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,       \
                     (float)weight_decay)
  if (mode == YR_OPT_ADAMW) {
    if (zero_grad) YR_LAUNCH_ADAM(true, true); else YR_LAUNCH_ADAM(true, false);
  } else {
    if (zero_grad) YR_LAUNCH_ADAM(false, true); else YR_LAUNCH_ADAM(false, false);
  }
#undef YR_LAUNCH_ADAM
  return launch_status();
}

extern "C" int yr_sgd_dense(float* p, float* g, int64_t n, double lr, double weight_decay, int zero_grad,
                            void* stream) {
  if (n < 0 || (n & 3)) return YR_ERR_BADARG;
  if (n == 0) return 0;
  if (!p || !g || !aligned16(p) || !aligned16(g)) return YR_ERR_BADARG;
  const int64_t n4 = n / 4;
  const int grid = grid_for(n4, kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (zero_grad)
    hipLaunchKernelGGL((sgd_dense_kernel<true>), dim3(grid), dim3(kBlock), 0, s, (float4*)p, (float4*)g, n4,
                       (float)lr, (float)weight_decay);
  else
    hipLaunchKernelGGL((sgd_dense_kernel<false>), dim3(grid), dim3(kBlock), 0, s, (float4*)p, (float4*)g, n4,
                       (float)lr, (float)weight_decay);
  return launch_status();
}
