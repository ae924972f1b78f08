// Shared device helpers for the gfx950 engine (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yelprec_engine.h"

namespace yr {

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves, one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxGrid = YR_LOSS_PARTIALS;   // 256 CUs x 8 workgroups

// Row geometry for an embedding width D: a row is spread over LPR lanes holding
// EPL consecutive floats each, so one wave-instruction touches 64/LPR rows with
// 64*EPL*4 contiguous bytes per row group (the shape the memory-side float
// atomics run at full rate for: 256 contiguous bytes or two 128-B segments).
template <int D>
struct RowGeom {
  static_assert(D == 16 || D == 32 || D == 64 || D == 128, "unsupported width");
  static constexpr int LPR = D < 64 ? D : 64;   // lanes per row
  static constexpr int EPL = D / LPR;           // elements per lane (1 or 2)
  static constexpr int RPW = kWave / LPR;       // rows per wave pass
};

// sum over the LPR lanes of a row group; every lane of the group gets the total
template <int LPR>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, kWave);
  return x;
}

__device__ __forceinline__ float wave_sum(float x) { return group_sum<kWave>(x); }

// x + (x moved across lanes by a DPP control); all lanes active
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false);
  return x + __int_as_float(moved);
}

// Sum over aligned groups of LPR lanes using DPP (no LDS crossbar traffic): quad swaps, then the
// mirror forms inside a 16-lane DPP row; 32-lane groups add one cross-row shuffle.
template <int LPR>
__device__ __forceinline__ float group_sum_dpp(float x) {
  static_assert(LPR == 4 || LPR == 8 || LPR == 16 || LPR == 32, "group size");
  x = dpp_add<0xB1>(x);                      // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);                      // quad_perm [2,3,0,1]
  if (LPR >= 8) x = dpp_add<0x141>(x);       // row_half_mirror
  if (LPR >= 16) x = dpp_add<0x140>(x);      // row_mirror
  if (LPR >= 32) x += __shfl_xor(x, 16, kWave);
  return x;
}

// Fast transcendental forms (v_exp_f32 / v_log_f32 / v_rcp_f32, about 1 ulp each) for the
// per-triplet sigmoid / softplus: the library expf/log1pf/IEEE divide cost ~10x the instructions.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// (softplus(-x), sigmoid(-x)) from one exponential
__device__ __forceinline__ void bpr_terms(float x, float& softplus_negx, float& sigmoid_negx) {
  const float z = fast_exp(-fabsf(x));
  const float r = fast_rcp(1.0f + z);
  sigmoid_negx = x < 0.0f ? r : z * r;
  softplus_negx = fmaxf(-x, 0.0f) + fast_log(1.0f + z);
}

// softplus(-x) = -logsigmoid(x) = max(-x, 0) + log1p(exp(-|x|))
__device__ __forceinline__ float softplus_neg(float x) {
  return fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
}

// sigmoid(-x), evaluated the way ATen's log_sigmoid_backward does
__device__ __forceinline__ float sigmoid_neg(float x) {
  const float z = expf(-fabsf(x));
  return x < 0.0f ? 1.0f / (1.0f + z) : z / (1.0f + z);
}

// block-wide sum of one float per thread -> valid in thread 0
__device__ __forceinline__ float block_sum(float x, float* smem /* >= kWavesPerBlock */) {
  x = wave_sum(x);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) smem[wave] = x;
  __syncthreads();
  float t = 0.0f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) t += smem[w];
  }
  return t;
}

// A count that thousands of workgroups add to is kept as YR_COUNT_SLOTS partial counts, one per 128-byte line
// (one counter would serialise ~10 k same-address atomics: measured +90 us on a 40 us kernel); every reader
// adds the slots up with one 64-lane gather and a wave sum.
constexpr int kCountStride = YR_COUNT_WORDS / YR_COUNT_SLOTS;
static_assert(YR_COUNT_SLOTS == kWave, "one slot per lane");

__device__ __forceinline__ int32_t spread_count(const int32_t* __restrict__ count, int lane_in_wave) {
  int32_t c = count[lane_in_wave * kCountStride];
#pragma unroll
  for (int d = kWave / 2; d >= 1; d >>= 1) c += __shfl_xor(c, d, kWave);
  return c;
}

__device__ __forceinline__ void spread_count_add(int32_t* __restrict__ count, unsigned workgroup, int value) {
  if (value > 0) atomicAdd(count + (workgroup % YR_COUNT_SLOTS) * kCountStride, value);
}

inline int grid_for(int64_t work_items, int per_block) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace yr
