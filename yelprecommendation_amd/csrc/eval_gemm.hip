// Full-catalogue scoring for evaluation on the matrix cores of gfx950 (MI355X).
//
// Replaces the reference's per-user Python loop
//   for user in eval_users: pred = model([user] * num_items, arange(num_items))
// (reference trainers/mf_trainer.py:138-140 -> models/mf.py:20-23, one GEMV per user) with
// ONE float32 GEMM  S[r, j] = U[users[r]] . I[j]  on v_mfma_f32_32x32x2_f32: exact f32
// (each product rounded once, fma-chained), so rankings match the CPU reference except at
// float near-ties.  The score rows feed csrc/topk.hip (mask + top-k).
//
// Tiling: a 256-thread workgroup (4 waves) computes a 128-user x 128-item block of S.  Each
// wave owns 32 users: their rows live in registers as the A operand for the whole block (lane
// (i, h) holds dims [h*D/2, (h+1)*D/2) of user i — the MFMA's two k-slots per step are mapped to
// the two halves of the row, which keeps every global load a contiguous 16-byte piece).  The
// 128 item rows are staged once in LDS (row pitch D+4 floats: conflict-free ds_read_b128) and
// shared by the 4 waves as the B operand.  K = D fits in registers, so there is no k-loop: D/2
// MFMAs per 32x32 tile, 4 tiles per wave.
#include "common.h"

namespace yr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kEvalUsers = 128;   // per workgroup
constexpr int kEvalItems = 128;

template <int D>
__global__ __launch_bounds__(kBlock) void mf_scores_mfma_kernel(const float* __restrict__ U,
                                                                const float* __restrict__ I,
                                                                const int64_t* __restrict__ users, int64_t nrows,
                                                                int64_t num_users, int64_t num_items,
                                                                float* __restrict__ S, int64_t row_stride,
                                                                int32_t* __restrict__ err_flag) {
  constexpr int HALF = D / 2;            // dims per lane
  constexpr int PITCH = D + 4;           // LDS row pitch in floats
  __shared__ __attribute__((aligned(16))) float s_items[kEvalItems * PITCH];

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.y * kEvalUsers;
  const int64_t item0 = (int64_t)blockIdx.x * kEvalItems;

  // stage the item rows (zeros beyond the catalogue)
  for (int q = threadIdx.x; q < kEvalItems * (D / 4); q += kBlock) {
    const int r = q / (D / 4), c = q % (D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item0 + r < num_items) v = *reinterpret_cast<const float4*>(I + (item0 + r) * D + 4 * c);
    *reinterpret_cast<float4*>(s_items + r * PITCH + 4 * c) = v;
  }

  // A operand: this lane's half of its user's row
  float a[HALF];
  const int64_t r = row0 + wave * 32 + i;
  int64_t uid = r < nrows ? users[r] : 0;
  bool ok = r < nrows;
  if (ok && (uint64_t)uid >= (uint64_t)num_users) {
    if (err_flag) atomicOr(err_flag, YR_FLAG_BAD_USER);
    ok = false;
    uid = 0;
  }
#pragma unroll
  for (int q = 0; q < HALF / 4; ++q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4*>(U + uid * D + h * HALF + 4 * q);
    a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
  }
  __syncthreads();

#pragma unroll
  for (int t = 0; t < kEvalItems / 32; ++t) {
    float b[HALF];
    const float* src = s_items + (t * 32 + i) * PITCH + h * HALF;
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < HALF; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    // C layout: col (item) = lane & 31, row (user) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int64_t col = item0 + t * 32 + i;
    if (col < num_items) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int64_t row = row0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (row < nrows) S[row * row_stride + col] = acc[reg];
      }
    }
  }
}

}  // namespace yr

using namespace yr;

extern "C" int yr_mf_scores_gemm(const float* U, const float* I, const int64_t* users, int64_t nrows, int D,
                                 int64_t num_users, int64_t num_items, float* scores, int64_t row_stride,
                                 int32_t* err_flag, void* stream) {
  if (nrows < 0 || num_users <= 0 || num_items <= 0 || row_stride < num_items) return YR_ERR_BADARG;
  if (nrows == 0) return 0;
  if (!U || !I || !users || !scores) return YR_ERR_BADARG;
  const dim3 grid((unsigned)((num_items + kEvalItems - 1) / kEvalItems),
                  (unsigned)((nrows + kEvalUsers - 1) / kEvalUsers));
  if (grid.y > 65535) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 16: hipLaunchKernelGGL((mf_scores_mfma_kernel<16>), grid, dim3(kBlock), 0, s, U, I, users, nrows, num_users, num_items, scores, row_stride, err_flag); break;
    case 32: hipLaunchKernelGGL((mf_scores_mfma_kernel<32>), grid, dim3(kBlock), 0, s, U, I, users, nrows, num_users, num_items, scores, row_stride, err_flag); break;
    case 64: hipLaunchKernelGGL((mf_scores_mfma_kernel<64>), grid, dim3(kBlock), 0, s, U, I, users, nrows, num_users, num_items, scores, row_stride, err_flag); break;
    case 128: hipLaunchKernelGGL((mf_scores_mfma_kernel<128>), grid, dim3(kBlock), 0, s, U, I, users, nrows, num_users, num_items, scores, row_stride, err_flag); break;
    default: return YR_ERR_UNSUPPORTED;
  }
  return launch_status();
}
