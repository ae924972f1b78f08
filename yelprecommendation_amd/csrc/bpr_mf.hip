// BPR matrix-factorisation kernels for gfx950 (MI355X, CDNA4; wave = 64).
//
// Replaces the ATen op chain the reference issues per batch
// (reference trainers/mf_trainer.py:104-111 -> models/mf.py:20-23, loss.py:25-27,
// autograd's embedding_dense_backward) with one gather + score + loss +
// scatter-add kernel.  HBM/Infinity-Cache bound byte work: no MFMA here.
//
// Layout: tables are row-major float32 [rows, D].  One row lives on LPR = min(D,64)
// lanes (element l + 64*j on lane l), so every load, store and float atomic of a row
// is one or two wave-instructions over 256 (128 for D=32, 64 for D=16) contiguous
// bytes — the access shape the memory-side float atomics sustain their full rate on.
// The (u, i, j) triplets of a tile are staged through LDS once (coalesced 8-byte
// loads, range-checked, narrowed to int32) and then broadcast-read by the waves.
#include "common.h"

namespace yr {

constexpr int kTile = kBlock;   // triplets staged per workgroup iteration
constexpr int kUnroll = 4;      // row groups in flight per wave

template <int D>
struct Row {
  float e[RowGeom<D>::EPL];
};

template <int D>
__device__ __forceinline__ Row<D> load_row(const float* __restrict__ T, int32_t r, int l) {
  Row<D> x;
  const float* p = T + (int64_t)r * D + l;
#pragma unroll
  for (int j = 0; j < RowGeom<D>::EPL; ++j) x.e[j] = p[j * kWave];
  return x;
}

// Stage one tile of triplets into LDS as int32, -1 for tail / out-of-range entries.
template <int NIDX, int TILE = kTile>
__device__ __forceinline__ int stage_indices(int32_t (*s_idx)[TILE], const int64_t* const* idx,
                                             const int64_t* limit, const int* flag_bit,
                                             int64_t tile, int64_t B) {
  if (TILE < kBlock && (int)threadIdx.x >= TILE) return 0;
  const int64_t b = tile * TILE + threadIdx.x;
  int32_t v[NIDX];
  int bad = 0;
#pragma unroll
  for (int k = 0; k < NIDX; ++k) v[k] = -1;
  if (b < B) {
    int64_t raw[NIDX];
#pragma unroll
    for (int k = 0; k < NIDX; ++k) {
      raw[k] = idx[k][b];
      if ((uint64_t)raw[k] >= (uint64_t)limit[k]) bad |= flag_bit[k];
    }
    if (!bad) {
#pragma unroll
      for (int k = 0; k < NIDX; ++k) v[k] = (int32_t)raw[k];
    }
  }
#pragma unroll
  for (int k = 0; k < NIDX; ++k) s_idx[k][threadIdx.x] = v[k];
  return bad;
}

// ---------------------------------------------------------------------------
// fused forward + loss (+ backward scatter-add when BWD)
// ---------------------------------------------------------------------------
// TILE triplets are staged per workgroup iteration, TILE/4 per wave: 256 for throughput, 64 for
// small batches (4x the workgroups, a quarter of the serial passes per wave: latency, not bandwidth,
// is what a 32..16k-triplet step waits for).
template <int D, bool BWD, int TILE>
__global__ __launch_bounds__(kBlock) void bpr_fwd_bwd_kernel(
    const float* __restrict__ U, const float* __restrict__ I,
    const int64_t* __restrict__ user, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    int64_t B, int64_t num_users, int64_t num_items, float inv_batch,
    float* __restrict__ gradU, float* __restrict__ gradI,
    float* __restrict__ loss_partials, int32_t* __restrict__ err_flag, uint8_t* __restrict__ touched) {
  using G = RowGeom<D>;
  constexpr int PER_WAVE = TILE / kWavesPerBlock;
  __shared__ int32_t s_idx[3][TILE];
  __shared__ float s_red[kWavesPerBlock];

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int sub = lane / G::LPR, l = lane % G::LPR;
  // slots of the partial-sum array no workgroup owns
  if (blockIdx.x == 0)
    for (int i = gridDim.x + threadIdx.x; i < YR_LOSS_PARTIALS; i += kBlock) loss_partials[i] = 0.0f;
  const int64_t* idx[3] = {user, pos, neg};
  const int64_t limit[3] = {num_users, num_items, num_items};
  const int flag_bit[3] = {YR_FLAG_BAD_USER, YR_FLAG_BAD_ITEM, YR_FLAG_BAD_ITEM};

  float loss_acc = 0.0f;
  int bad = 0;
  const int64_t ntiles = (B + TILE - 1) / TILE;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    bad |= stage_indices<3, TILE>(s_idx, idx, limit, flag_bit, tile, B);
    __syncthreads();

    // this wave owns triplets [wave*PER_WAVE, (wave+1)*PER_WAVE) of the tile
    for (int k = 0; k < PER_WAVE; k += G::RPW * kUnroll) {
      int32_t uu[kUnroll], pp[kUnroll], nn[kUnroll];
      Row<D> ru[kUnroll], rp[kUnroll], rn[kUnroll];
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        const int t = wave * PER_WAVE + k + q * G::RPW + sub;
        const bool in = k + q * G::RPW + sub < PER_WAVE;      // D = 16 with the small tile: 16 slots, 16 triplets
        uu[q] = in ? s_idx[0][t] : -1;
        pp[q] = in ? s_idx[1][t] : -1;
        nn[q] = in ? s_idx[2][t] : -1;
      }
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        // invalid entries read row 0 (harmless) and are masked out below
        ru[q] = load_row<D>(U, max(uu[q], 0), l);
        rp[q] = load_row<D>(I, max(pp[q], 0), l);
        rn[q] = load_row<D>(I, max(nn[q], 0), l);
      }
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        const bool valid = uu[q] >= 0;
        float diff[G::EPL], part = 0.0f;
#pragma unroll
        for (int j = 0; j < G::EPL; ++j) {
          diff[j] = rp[q].e[j] - rn[q].e[j];
          part = fmaf(ru[q].e[j], diff[j], part);
        }
        const float x = group_sum<G::LPR>(part);   // s+ - s- = u . (i+ - i-)
        if (valid && l == 0) loss_acc += softplus_neg(x);
        if (BWD) {
          const float g = -sigmoid_neg(x) * inv_batch;
          if (valid) {
            float* du = gradU + (int64_t)uu[q] * D + l;
            float* dp = gradI + (int64_t)pp[q] * D + l;
            float* dn = gradI + (int64_t)nn[q] * D + l;
#pragma unroll
            for (int j = 0; j < G::EPL; ++j) {
              const float gu = g * ru[q].e[j];
              atomicAdd(du + j * kWave, g * diff[j]);
              atomicAdd(dp + j * kWave, gu);
              atomicAdd(dn + j * kWave, -gu);
            }
            if (touched && l == 0) {              // rows whose gradient is not zero: [users | items]
              touched[uu[q]] = 1;
              touched[num_users + pp[q]] = 1;
              touched[num_users + nn[q]] = 1;
            }
          }
        }
      }
    }
    __syncthreads();
  }

  const float total = block_sum(loss_acc, s_red);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = total;
  if (bad && err_flag) atomicOr(err_flag, bad);
}

// zero the loss-partial slots no workgroup owns
__global__ void clear_tail_kernel(float* p, int from, int to) {
  const int i = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < to) p[i] = 0.0f;
}

// ---------------------------------------------------------------------------
// MatrixFactorization.forward and its backward
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void mf_score_kernel(
    const float* __restrict__ U, const float* __restrict__ I,
    const int64_t* __restrict__ user, const int64_t* __restrict__ item,
    int64_t B, int64_t num_users, int64_t num_items,
    float* __restrict__ out, int32_t* __restrict__ err_flag) {
  using G = RowGeom<D>;
  __shared__ int32_t s_idx[2][kTile];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int sub = lane / G::LPR, l = lane % G::LPR;
  const int64_t* idx[2] = {user, item};
  const int64_t limit[2] = {num_users, num_items};
  const int flag_bit[2] = {YR_FLAG_BAD_USER, YR_FLAG_BAD_ITEM};
  int bad = 0;
  const int64_t ntiles = (B + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    bad |= stage_indices<2>(s_idx, idx, limit, flag_bit, tile, B);
    __syncthreads();
    for (int k = 0; k < kWave; k += G::RPW * kUnroll) {
      int32_t uu[kUnroll], ii[kUnroll];
      Row<D> ru[kUnroll], ri[kUnroll];
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        const int t = wave * kWave + k + q * G::RPW + sub;
        uu[q] = s_idx[0][t];
        ii[q] = s_idx[1][t];
      }
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        ru[q] = load_row<D>(U, max(uu[q], 0), l);
        ri[q] = load_row<D>(I, max(ii[q], 0), l);
      }
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        float part = 0.0f;
#pragma unroll
        for (int j = 0; j < G::EPL; ++j) part = fmaf(ru[q].e[j], ri[q].e[j], part);
        const float s = group_sum<G::LPR>(part);
        const int64_t b = tile * kTile + wave * kWave + k + q * G::RPW + sub;
        if (l == 0 && b < B) out[b] = uu[q] >= 0 ? s : 0.0f;
      }
    }
    __syncthreads();
  }
  if (bad && err_flag) atomicOr(err_flag, bad);
}

template <int D>
__global__ __launch_bounds__(kBlock) void mf_score_backward_kernel(
    const float* __restrict__ U, const float* __restrict__ I,
    const int64_t* __restrict__ user, const int64_t* __restrict__ item,
    const float* __restrict__ gout, int64_t B, int64_t num_users, int64_t num_items,
    float* __restrict__ gradU, float* __restrict__ gradI, int32_t* __restrict__ err_flag) {
  using G = RowGeom<D>;
  __shared__ int32_t s_idx[2][kTile];
  __shared__ float s_g[kTile];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int sub = lane / G::LPR, l = lane % G::LPR;
  const int64_t* idx[2] = {user, item};
  const int64_t limit[2] = {num_users, num_items};
  const int flag_bit[2] = {YR_FLAG_BAD_USER, YR_FLAG_BAD_ITEM};
  int bad = 0;
  const int64_t ntiles = (B + kTile - 1) / kTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    bad |= stage_indices<2>(s_idx, idx, limit, flag_bit, tile, B);
    {
      const int64_t b = tile * kTile + threadIdx.x;
      s_g[threadIdx.x] = b < B ? gout[b] : 0.0f;
    }
    __syncthreads();
    for (int k = 0; k < kWave; k += G::RPW * kUnroll) {
#pragma unroll
      for (int q = 0; q < kUnroll; ++q) {
        const int t = wave * kWave + k + q * G::RPW + sub;
        const int32_t uu = s_idx[0][t], ii = s_idx[1][t];
        const float g = s_g[t];
        if (uu >= 0) {
          const Row<D> ru = load_row<D>(U, uu, l);
          const Row<D> ri = load_row<D>(I, ii, l);
          float* du = gradU + (int64_t)uu * D + l;
          float* di = gradI + (int64_t)ii * D + l;
#pragma unroll
          for (int j = 0; j < G::EPL; ++j) {
            atomicAdd(du + j * kWave, g * ri.e[j]);
            atomicAdd(di + j * kWave, g * ru.e[j]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (bad && err_flag) atomicOr(err_flag, bad);
}

// loss_out = scale * sum(partials)  (fixed order => bit-reproducible loss)
__global__ __launch_bounds__(kBlock) void loss_finalize_kernel(const float* __restrict__ partials, float scale,
                                                               float* __restrict__ loss_out,
                                                               double* __restrict__ loss_accum) {
  __shared__ float s_red[kWavesPerBlock];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < YR_LOSS_PARTIALS; i += kBlock) acc += partials[i];
  const float total = block_sum(acc, s_red);
  if (threadIdx.x == 0) {
    const float v = total * scale;
    if (loss_out) loss_out[0] = v;
    if (loss_accum) loss_accum[0] += (double)v;
  }
}

}  // namespace yr

using namespace yr;

#define YR_DISPATCH_D(D, ...)                                 \
  switch (D) {                                                \
    case 16: { constexpr int kD = 16; __VA_ARGS__; } break;   \
    case 32: { constexpr int kD = 32; __VA_ARGS__; } break;   \
    case 64: { constexpr int kD = 64; __VA_ARGS__; } break;   \
    case 128: { constexpr int kD = 128; __VA_ARGS__; } break; \
    default: return YR_ERR_UNSUPPORTED;                       \
  }

extern "C" int yr_mf_score(const float* U, const float* I, const int64_t* user, const int64_t* item,
                           int64_t B, int D, int64_t num_users, int64_t num_items, float* out,
                           int32_t* err_flag, void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!U || !I || !user || !item || !out) return YR_ERR_BADARG;
  const int grid = grid_for(B, kTile);
  YR_DISPATCH_D(D, hipLaunchKernelGGL((mf_score_kernel<kD>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                                       U, I, user, item, B, num_users, num_items, out, err_flag));
  return launch_status();
}

extern "C" int yr_mf_score_backward(const float* U, const float* I, const int64_t* user, const int64_t* item,
                                    const float* gout, int64_t B, int D, int64_t num_users, int64_t num_items,
                                    float* gradU, float* gradI, int32_t* err_flag, void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!U || !I || !user || !item || !gout || !gradU || !gradI) return YR_ERR_BADARG;
  const int grid = grid_for(B, kTile);
  YR_DISPATCH_D(D, hipLaunchKernelGGL((mf_score_backward_kernel<kD>), dim3(grid), dim3(kBlock), 0,
                                       (hipStream_t)stream, U, I, user, item, gout, B, num_users, num_items,
                                       gradU, gradI, err_flag));
  return launch_status();
}

static int fwd_bwd_launch(const float* U, const float* I, const int64_t* user, const int64_t* pos, const int64_t* neg,
                          int64_t B, int D, int64_t num_users, int64_t num_items, float inv_batch, float* gradU,
                          float* gradI, float* loss_partials, int32_t* err_flag, uint8_t* touched, void* stream) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || !loss_partials) return YR_ERR_BADARG;
  if ((gradU == nullptr) != (gradI == nullptr)) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) {
    hipLaunchKernelGGL(clear_tail_kernel, dim3(YR_LOSS_PARTIALS / kBlock), dim3(kBlock), 0, s, loss_partials, 0,
                       YR_LOSS_PARTIALS);
    return launch_status();
  }
  if (!U || !I || !user || !pos || !neg) return YR_ERR_BADARG;
  // triplets staged per workgroup: 256 for throughput; 64 up to 131,072 triplets so that every CU gets
  // workgroups (measured: 32k 77 -> 51 us, 64k 80 -> 73 us); 16 up to 16,384 — a wave then makes ONE pass
  // of four triplets instead of four passes in a row (a 32-triplet step waited 10 us for those round trips)
  const int tile = B <= 16384 ? 16 : B <= 131072 ? 64 : kTile;
  const int grid = grid_for(B, tile);
#define YR_LAUNCH_FB(BWD, TILE)                                                                                \
  YR_DISPATCH_D(D, hipLaunchKernelGGL((bpr_fwd_bwd_kernel<kD, BWD, TILE>), dim3(grid), dim3(kBlock), 0, s, U, I, \
                                      user, pos, neg, B, num_users, num_items, inv_batch, gradU, gradI,          \
                                      loss_partials, err_flag, touched))
  if (gradU) {
    if (tile == 16) { YR_LAUNCH_FB(true, 16); } else if (tile == 64) { YR_LAUNCH_FB(true, 64); } else { YR_LAUNCH_FB(true, kTile); }
  } else {
    if (tile == 16) { YR_LAUNCH_FB(false, 16); } else if (tile == 64) { YR_LAUNCH_FB(false, 64); } else { YR_LAUNCH_FB(false, kTile); }
  }
#undef YR_LAUNCH_FB
  return launch_status();
}

extern "C" int yr_bpr_mf_fwd_bwd(const float* U, const float* I, const int64_t* user, const int64_t* pos,
                                 const int64_t* neg, int64_t B, int D, int64_t num_users, int64_t num_items,
                                 float inv_batch, float* gradU, float* gradI, float* loss_partials,
                                 int32_t* err_flag, void* stream) {
  return fwd_bwd_launch(U, I, user, pos, neg, B, D, num_users, num_items, inv_batch, gradU, gradI,
                        loss_partials, err_flag, nullptr, stream);
}

// The whole step for small batches in two launches: scatter (above, marking the rows it touches) and
// one dense Adam pass over BOTH tables that reads / clears a gradient row only where it was marked
// and reduces the loss partials (csrc/optim.hip).
extern "C" int yr_adam_dense_dual(float* p0, float* g0, float* m0, float* v0, int64_t n0, float* p1, float* g1,
                                  float* m1, float* v1, int64_t n1, int row_width, uint8_t* touched0,
                                  uint8_t* touched1, double lr, double step_size, double bc2_sqrt, double beta1,
                                  double beta2, double eps, double weight_decay, int mode,
                                  const float* loss_partials, float loss_scale, float* loss_out, double* loss_accum,
                                  void* stream);

extern "C" int yr_bpr_mf_scatter_step(float* U, float* I, float* gradU, float* gradI, float* mU, float* vU,
                                      float* mI, float* vI, uint8_t* touched, const int64_t* user,
                                      const int64_t* pos, const int64_t* neg, int64_t B, int D, int64_t num_users,
                                      int64_t num_items, float inv_batch, double lr, double step_size,
                                      double bc2_sqrt, double beta1, double beta2, double eps, double weight_decay,
                                      int mode, float* loss_partials, float* loss_out, double* loss_accum,
                                      int32_t* err_flag, void* stream) {
  if (!gradU || !gradI || !mU || !vU || !mI || !vI || !U || !I) return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  const int rc = fwd_bwd_launch(U, I, user, pos, neg, B, D, num_users, num_items, inv_batch, gradU, gradI,
                                loss_partials, err_flag, touched, stream);
  if (rc) return rc;
  return yr_adam_dense_dual(U, gradU, mU, vU, num_users * D, I, gradI, mI, vI, num_items * D, D, touched,
                            touched ? touched + num_users : nullptr, lr, step_size, bc2_sqrt, beta1, beta2, eps,
                            weight_decay, mode, loss_partials, inv_batch, loss_out, loss_accum, stream);
}

extern "C" int yr_loss_finalize(const float* loss_partials, float scale, float* loss_out, double* loss_accum,
                                void* stream) {
  if (!loss_partials) return YR_ERR_BADARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, loss_partials, scale,
                     loss_out, loss_accum);
  return launch_status();
}

// ---------------------------------------------------------------------------
// BPRLoss on its own (reference loss.py:25-27) for callers that keep the
// reference's three-call shape  model(u,p), model(u,n), loss(pos, neg).
// ---------------------------------------------------------------------------
namespace yr {

__global__ __launch_bounds__(kBlock) void bpr_loss_fwd_kernel(const float* __restrict__ pos,
                                                              const float* __restrict__ neg, int64_t B,
                                                              float* __restrict__ partials) {
  __shared__ float s_red[kWavesPerBlock];
  float acc = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += stride)
    acc += softplus_neg(pos[b] - neg[b]);
  const float total = block_sum(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
  // the slots no workgroup owns are cleared by workgroup 0 (no second launch for the tail)
  if (blockIdx.x == 0)
    for (int i = gridDim.x + threadIdx.x; i < YR_LOSS_PARTIALS; i += kBlock) partials[i] = 0.0f;
}

// gpos[b] = -sigmoid(-(pos-neg)) * gout[0] * inv_batch ; gneg[b] = -gpos[b]
__global__ __launch_bounds__(kBlock) void bpr_loss_bwd_kernel(const float* __restrict__ pos,
                                                              const float* __restrict__ neg,
                                                              const float* __restrict__ gout, float inv_batch,
                                                              int64_t B, float* __restrict__ gpos,
                                                              float* __restrict__ gneg) {
  const float go = gout[0] * inv_batch;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += stride) {
    const float g = -sigmoid_neg(pos[b] - neg[b]) * go;
    gpos[b] = g;
    gneg[b] = -g;
  }
}

}  // namespace yr

extern "C" int yr_bpr_loss_fwd(const float* pos, const float* neg, int64_t B, float* loss_partials, void* stream) {
  if (B < 0 || !loss_partials) return YR_ERR_BADARG;
  if (B > 0 && (!pos || !neg)) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int grid = B > 0 ? grid_for(B, kBlock) : 0;
  if (grid) {
    hipLaunchKernelGGL(bpr_loss_fwd_kernel, dim3(grid), dim3(kBlock), 0, s, pos, neg, B, loss_partials);
  } else {
    hipLaunchKernelGGL(clear_tail_kernel, dim3(YR_LOSS_PARTIALS / kBlock), dim3(kBlock), 0, s, loss_partials, 0,
                       YR_LOSS_PARTIALS);
  }
  return launch_status();
}

extern "C" int yr_bpr_loss_bwd(const float* pos, const float* neg, const float* gout, float inv_batch, int64_t B,
                               float* gpos, float* gneg, void* stream) {
  if (B < 0) return YR_ERR_BADARG;
  if (B == 0) return 0;
  if (!pos || !neg || !gout || !gpos || !gneg) return YR_ERR_BADARG;
  hipLaunchKernelGGL(bpr_loss_bwd_kernel, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pos,
                     neg, gout, inv_batch, B, gpos, gneg);
  return launch_status();
}
