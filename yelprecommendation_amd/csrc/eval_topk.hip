// Fused evaluation for gfx950 (MI355X): full-catalogue scores on the f32 matrix cores with the
// train-item mask and the top-k selection in the GEMM epilogue — the U x I score matrix (4.8 GB at
// Yelp2018 size) is never written.
//
// Replaces, for ALL eval users at once, the reference's per-user loop
//   pred = model([user] * num_items, arange(num_items))            (trainers/mf_trainer.py:138-140)
//   pred[mask_items] = -3.40282e+38; argpartition; argsort         (trainers/mf_trainer.py:163-178)
//
// Workgroup = 2 waves = 64 eval users (32 per wave, their rows held in registers as the MFMA A
// operand for the whole kernel).  The catalogue is walked in chunks of 128 items staged in LDS
// (pitch D+4) and shared by both waves; per 32x32 tile D/2 x v_mfma_f32_32x32x2_f32 (exact f32).
// Epilogue per tile: each lane holds one item's score for 16 users.  A score is a candidate when it
// beats the user's current k-th best (threshold kept in LDS, refreshed per tile); candidates are
// rare after the first few chunks (about k ln(I/k) per user in all), and each is inserted by its
// own lane into the user's sorted top-k list in LDS.  A user's list is only ever touched by the
// wave that owns the user, and within one accumulator register the 32 lanes of a half-wave all
// belong to the SAME user, so insertions are serialised per half-wave with ballot/ffs and the two
// halves proceed in parallel.
// Masks: each user's mask list (CSR, item ids ASCENDING) is walked by a cursor as the chunks advance;
// the (rare) masked items of the current 32-item tile become a 32-bit row mask in LDS, consulted
// only for tiles that contain one.
// Order: score descending, item id ascending among equal scores (as csrc/topk.hip).
#include "common.h"

namespace yr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kEtUsersPerWave = 32;
constexpr int kEtWaves = 2;
constexpr int kEtThreads = kEtWaves * kWave;
constexpr int kEtUsers = kEtWaves * kEtUsersPerWave;   // 64 per workgroup
constexpr int kEtStageVec = 16;                        // float4 registers per thread for the next chunk
constexpr int kEtMaxK = 16;

template <int D>
struct EtChunk {
  // items per LDS stage: chosen so that a chunk is exactly kEtStageVec float4 per thread
  static constexpr int ITEMS = kEtStageVec * kEtThreads / (D / 4);     // 128 at D = 64
};

struct TopEntry {
  float s;
  int32_t i;
};

__device__ __forceinline__ bool et_better(float s, int32_t i, float s2, int32_t i2) {
  return s > s2 || (s == s2 && i < i2);
}

// Insert this register's candidates into the top-k lists.  The 32 lanes of a half-wave hold scores
// of 32 items for ONE user (list L, threshold *tau); the two halves work on their two users in
// parallel.  Per round each half takes its lowest candidate lane, broadcasts (score, item), all
// lanes read the list in parallel (lane e = entry e), the insertion position is a popcount over
// "entry stays ahead of the candidate", and the shifted list is written back in parallel.
__device__ __forceinline__ void et_insert_candidates(volatile TopEntry* L, volatile float* tau, float s, int item,
                                                  bool cand, int k, int lane) {
  const int e = lane & 31;
  const unsigned long long half_mask = (lane >> 5) ? 0xffffffff00000000ull : 0x00000000ffffffffull;
  unsigned long long pending = __ballot(cand);
  while (pending) {
    const unsigned long long mine = pending & half_mask;
    const int first = mine ? __ffsll((long long)mine) - 1 : -1;        // uniform inside the half
    const int src = first >= 0 ? first : lane;
    const float cs = __shfl(s, src, kWave);
    const int ci = __shfl(item, src, kWave);
    // entry e of the list (lanes e >= k see a sentinel that never stays ahead)
    const float ls = e < k ? L[e].s : -INFINITY;
    const int li = e < k ? L[e].i : 0x7fffffff;
    const bool ahead = e < k && et_better(ls, li, cs, ci);
    const int pos = __popcll(__ballot(ahead) & half_mask);             // entries that stay in front
    const float prev_s = __shfl_up(ls, 1, kWave);                      // entry e-1 (same half for e >= 1)
    const int prev_i = __shfl_up(li, 1, kWave);
    if (first >= 0 && pos < k && e < k && e >= pos) {
      const float ns = e == pos ? cs : prev_s;
      const int ni = e == pos ? ci : prev_i;
      L[e].s = ns;
      L[e].i = ni;
      if (e == k - 1) *tau = ns;                                       // the new k-th best
    }
    if (lane == first) cand = false;
    if (cand && s < *tau) cand = false;                                // re-check against the raised threshold
    pending = __ballot(cand);
  }
}

template <int D>
__global__ __launch_bounds__(kEtThreads) void mf_eval_topk_kernel(
    const float* __restrict__ U, const float* __restrict__ I, const int64_t* __restrict__ users, int64_t nrows,
    int64_t num_users, int num_items, const int64_t* __restrict__ mask_ptr, const int64_t* __restrict__ mask_idx,
    float mask_value, int k, int64_t* __restrict__ out, int32_t* __restrict__ err_flag) {
  constexpr int HALF = D / 2;
  constexpr int PITCH = D + 4;
  constexpr int kEtChunk = EtChunk<D>::ITEMS;
  __shared__ __attribute__((aligned(16))) float s_items[kEtChunk * PITCH];
  // lists and thresholds are read by lanes other than the one that wrote them (same wave, in
  // program order): volatile keeps the compiler from caching them in registers
  __shared__ volatile TopEntry s_list[kEtUsers][kEtMaxK];
  __shared__ volatile float s_tau[kEtUsers];        // current k-th best score per user
  __shared__ int64_t s_cur[kEtUsers];               // cursor into the user's mask list
  __shared__ int64_t s_end[kEtUsers];
  __shared__ uint32_t s_mbits[kEtUsers];            // masked items of the current 32-item tile

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * kEtUsers;
  const int ubase = wave * kEtUsersPerWave;          // this wave's users inside the workgroup

  // per-user state
  for (int q = threadIdx.x; q < kEtUsers; q += kEtThreads) {
    const int64_t r = row0 + q;
    s_tau[q] = r < nrows ? -INFINITY : INFINITY;     // rows beyond the input never become candidates
    s_cur[q] = (mask_ptr && r < nrows) ? mask_ptr[r] : 0;
    s_end[q] = (mask_ptr && r < nrows) ? mask_ptr[r + 1] : 0;
    for (int e = 0; e < kEtMaxK; ++e) { s_list[q][e].s = -INFINITY; s_list[q][e].i = 0x7fffffff; }
  }

  // A operand: this lane's half of its user's row
  float a[HALF];
  {
    const int64_t r = row0 + ubase + i;
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
  }

  __syncthreads();
  // lane i < 32 owns user (ubase + i)'s mask cursor; its next masked item id stays in a register so
  // that the list is only touched when a masked item actually falls into the current tile
  int64_t m_cur = 0, m_end = 0;
  int next_masked = 0x7fffffff;
  if (mask_ptr && h == 0) {
    m_cur = s_cur[ubase + i];
    m_end = s_end[ubase + i];
    if (m_cur < m_end) next_masked = (int)mask_idx[m_cur];
  }

  // register staging of the item chunks: the loads of chunk c+1 are issued before the tiles of
  // chunk c are computed and land in LDS after the next barrier (global latency hidden under MFMA)
  float4 stage[kEtStageVec];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int v = 0; v < kEtStageVec; ++v) {
      const int q = threadIdx.x + v * kEtThreads;
      const int r = q / (D / 4), c = q % (D / 4);
      stage[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + r < num_items) stage[v] = *reinterpret_cast<const float4*>(I + (int64_t)(c0 + r) * D + 4 * c);
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < num_items; c0 += kEtChunk) {
    __syncthreads();                                 // previous chunk fully consumed
#pragma unroll
    for (int v = 0; v < kEtStageVec; ++v) {
      const int q = threadIdx.x + v * kEtThreads;
      *reinterpret_cast<float4*>(s_items + (q / (D / 4)) * PITCH + 4 * (q % (D / 4))) = stage[v];
    }
    __syncthreads();
    if (c0 + kEtChunk < num_items) fetch(c0 + kEtChunk);

#pragma unroll 1
    for (int t = 0; t < kEtChunk / 32; ++t) {
      const int item0 = c0 + t * 32;
      if (item0 >= num_items) break;                 // wave-uniform
      // ---- scores of 32 users x 32 items
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      {
        const float* src = s_items + (t * 32 + i) * PITCH + h * HALF;
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(src + 4 * q);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 0], b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], b.w, acc, 0, 0, 0);
        }
      }
      // ---- masked items of this tile: lane i (< 32) advances user (ubase + i)'s cursor
      bool any_masked = false;
      if (mask_ptr) {
        uint32_t bits = 0;
        while (next_masked < item0 + 32) {           // only lanes h == 0 can have next_masked < INT_MAX
          if (next_masked >= item0) bits |= 1u << (next_masked - item0);
          ++m_cur;
          next_masked = m_cur < m_end ? (int)mask_idx[m_cur] : 0x7fffffff;
        }
        any_masked = __ballot(bits != 0) != 0ull;    // wave-uniform
        if (any_masked && h == 0) s_mbits[ubase + i] = bits;   // read back by the whole wave below
      }
      const int item = item0 + i;
      const bool item_ok = item < num_items;
      // ---- candidates: one bit per accumulator register, tested against register copies of the
      // thresholds (refreshed from LDS only after this wave inserted something)
      if (any_masked) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int ul = ubase + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if ((s_mbits[ul] >> i) & 1u) acc[reg] = mask_value;
        }
      }
      uint32_t cmask = 0;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        // a stale (lower) threshold is safe: it only lets extra candidates through to the exact check
        const float tau = const_cast<const float*>(s_tau)[ubase + (reg & 3) + 8 * (reg >> 2) + 4 * h];
        if (acc[reg] >= tau) cmask |= 1u << reg;
      }
      if (!item_ok) cmask = 0;
      if (__ballot(cmask != 0) == 0ull) continue;    // no candidate in this tile
      // registers that hold a candidate in some lane (wave-uniform 16-bit mask), then ONE copy of the
      // insertion code looped over them (the accumulator is picked with a select chain: no dynamic
      // register indexing, no 16-fold code expansion)
      uint32_t regs = 0;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        if (__ballot((cmask >> reg) & 1u) != 0ull) regs |= 1u << reg;
      while (regs) {
        const int r = __builtin_ctz(regs);           // wave-uniform
        regs &= regs - 1;
        float sc = acc[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) sc = r == q ? acc[q] : sc;
        const int ul = ubase + (r & 3) + 8 * (r >> 2) + 4 * h;
        et_insert_candidates(s_list[ul], &s_tau[ul], sc, item, ((cmask >> r) & 1u) != 0, k, lane);
      }
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < kEtUsers * k; q += kEtThreads) {
    const int ul = q / k, e = q % k;
    const int64_t r = row0 + ul;
    if (r < nrows) {
      const int32_t it = s_list[ul][e].i;
      out[r * k + e] = it == 0x7fffffff ? -1 : (int64_t)it;
    }
  }
}

}  // namespace yr

using namespace yr;


extern "C" int yr_mf_eval_topk(const float* U, const float* I, const int64_t* users, int64_t nrows, int D,
                               int64_t num_users, int64_t num_items, const int64_t* mask_ptr,
                               const int64_t* mask_idx, float mask_value, int k, int64_t* out, int32_t* err_flag,
                               void* stream) {
  if (nrows < 0 || num_users <= 0 || num_items <= 0 || num_items > 0x7ffffff0 || k <= 0 || k > kEtMaxK)
    return YR_ERR_BADARG;
  if (nrows == 0) return 0;
  if (!U || !I || !users || !out || (mask_ptr && !mask_idx)) return YR_ERR_BADARG;
  const unsigned grid = (unsigned)((nrows + kEtUsers - 1) / kEtUsers);
  hipStream_t s = (hipStream_t)stream;
#define YR_ET_CASE(DD)                                                                                          \
  case DD:                                                                                                      \
    hipLaunchKernelGGL((mf_eval_topk_kernel<DD>), dim3(grid), dim3(kEtThreads), 0, s, U, I, users, nrows,       \
                       num_users, (int)num_items, mask_ptr, mask_idx, mask_value, k, out, err_flag);            \
    break
  switch (D) {
    YR_ET_CASE(16);
    YR_ET_CASE(32);
    YR_ET_CASE(64);
    YR_ET_CASE(128);
    default: return YR_ERR_UNSUPPORTED;
  }
#undef YR_ET_CASE
  return launch_status();
}
