// Fused evaluation for gfx950 (MI355X): full-catalogue f32 scores on the matrix cores with the
// train-item mask and the top-k selection in the GEMM epilogue — the U x I score matrix (4.8 GB at
// Yelp2018 size) is never written.
//
// Replaces, for ALL eval users at once, the reference's per-user loop
//   pred = model([user] * num_items, arange(num_items))            (trainers/mf_trainer.py:138-140)
//   pred[mask_items] = -3.40282e+38; argpartition; argsort         (trainers/mf_trainer.py:163-178)
//
// Workgroup = 4 waves = 128 eval users (32 per wave) x one SLICE of the catalogue (blockIdx.y).
// The users are the B operand of the matrix instruction — v_mfma_f32_32x32x2_f32 on the f32 rows, or
// v_mfma_f32_32x32x16_bf16 on three-term bf16 splits of both operands (SPLIT, below: the same f32 scores
// at 3/8 of the matrix-core cycles) —, held in registers for the whole kernel; the items are the A operand,
// staged through LDS in stages shared by the four waves (row pitch an odd number of 16-byte units, next
// stage prefetched into registers under the MFMAs).  In the 32x32 accumulator
// lane (i, h) then holds the scores of ITS OWN user i for 16 items of the tile (the other half-wave
// holds the other 16), so selection needs no cross-lane traffic:
//   * every lane keeps a private sorted top-KK list (score, item) in registers;
//   * a score that reaches the lane's threshold (its KK-th best as of the last flush — stale, hence
//     a superset) is appended to the lane's private LDS buffer: one compare and one predicated
//     ds_write per accumulator register;
//   * when some lane's buffer is half full, the whole wave flushes: slot j of all 64 buffers is
//     inserted into the 64 private lists in lockstep (a branch-free bubble pass), so the insertion
//     cost is shared by every lane that has a slot-j candidate instead of being paid per candidate;
//   * at the end the two half-waves exchange their lists with shuffles and merge them.
// Masks: each lane walks its user's mask list (CSR, item ids ASCENDING) with a private cursor and
// turns the masked items of the current tile into a 32-bit word.
// Slicing the catalogue (S slices -> S x as many workgroups, three resident per CU) is what fills the
// chip at Yelp2018 size (one wave per 32 users alone is < 1 wave per SIMD); each slice keeps its own
// top-k per user and a second small kernel merges the S sorted partial lists.  A PRESCAN launch (below)
// gives all lists of a user a common starting threshold.
// Order: score descending, item id ascending among equal scores (as csrc/topk.hip).
// Built with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile): accumulators in VGPRs.
#include <algorithm>

#include "common.h"

namespace yr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kEtUsersPerWave = 32;
constexpr int kEtWaves = 4;
constexpr int kEtThreads = kEtWaves * kWave;
constexpr int kEtUsers = kEtWaves * kEtUsersPerWave;   // 128 per workgroup
constexpr int kEtChunkItems = 64;                      // items per LDS stage (two 32-item tiles)
constexpr int kEtMaxK = 32;
// Candidate buffers of 8 slots (flush when some lane holds more than 4, checked after every 4 accumulator
// registers) and three catalogue slices at Yelp2018 size leave room for three workgroups per CU (f32 form:
// 2.08 -> 1.94 ms; 16 slots / check every 8 / two per CU before; 6 to 13 slots make no difference in the split form).
// YR_ET_SPLIT_CHUNK: items per LDS stage of the split form at D = 64 (32: 25.6 KB of stages, three workgroups
// per CU, 1.11 ms; 64: two per CU, 1.70 ms).
#ifndef YR_ET_FLUSH_AT
#define YR_ET_FLUSH_AT 4
#endif
#ifndef YR_ET_CHECK_EVERY
#define YR_ET_CHECK_EVERY 4
#endif
#ifndef YR_ET_TARGET_WGS
#define YR_ET_TARGET_WGS 768
#endif
#ifndef YR_ET_EARLY_OUT
#define YR_ET_EARLY_OUT 1
#endif
#ifndef YR_ET_SPLIT_CHUNK
#define YR_ET_SPLIT_CHUNK 32
#endif
// wave priority experiments (YR_ET_PRIO): 0 none; 1 matrix phase high; 2 epilogue high; 3 a fixed level per workgroup
#ifndef YR_ET_PRIO
#define YR_ET_PRIO 0
#endif
constexpr int kEtFlushAt = YR_ET_FLUSH_AT;             // flush when some lane holds more than this
constexpr int kEtCheckEvery = YR_ET_CHECK_EVERY;       // ... checked after this many accumulator registers

#ifdef YR_ET_STAMPS
// -DYR_ET_STAMPS: shader-clock cycles every wave spends per phase (scratch/eval_phases.sh), summed over the waves
__device__ unsigned long long g_et_phase[8];
#define ET_CLK() clock64()
#else
#define ET_CLK() 0ll
#endif

struct TopEntry {
  float s;
  int32_t i;
};

__device__ __forceinline__ bool et_better(float s, int32_t i, float s2, int32_t i2) {
  return s > s2 || (s == s2 && i < i2);
}

// insert (cs, ci) into the lane's sorted list where `live`; the displaced entries bubble down
template <int KK>
__device__ __forceinline__ void et_bubble(float (&Ls)[KK], int32_t (&Li)[KK], float cs, int32_t ci, bool live) {
#pragma unroll
  for (int e = 0; e < KK; ++e) {
    const bool sw = live && et_better(cs, ci, Ls[e], Li[e]);
    const float ts = Ls[e];
    const int32_t ti = Li[e];
    Ls[e] = sw ? cs : ts;
    Li[e] = sw ? ci : ti;
    cs = sw ? ts : cs;
    ci = sw ? ti : ci;
  }
}

// x = x1 + x2 + x3 with three bfloat16 terms (round to nearest each time; the residuals x - x1 and x - x1 - x2 are
// exact in f32): |x - x1 - x2 - x3| <= 2^-27 |x|.  Returns the three 16-bit patterns.
__device__ __forceinline__ void et_split3(float x, uint32_t& b1, uint32_t& b2, uint32_t& b3) {
  const __bf16 h1 = (__bf16)x;
  const float r1 = x - (float)h1;
  const __bf16 h2 = (__bf16)r1;
  const float r2 = r1 - (float)h2;
  const __bf16 h3 = (__bf16)r2;
  b1 = __builtin_bit_cast(unsigned short, h1);
  b2 = __builtin_bit_cast(unsigned short, h2);
  b3 = __builtin_bit_cast(unsigned short, h3);
}

// eight consecutive floats -> their three bf16 planes, eight 16-bit values (one uint4) each
__device__ __forceinline__ void et_split3x8(const float4 lo, const float4 hi, uint4& p1, uint4& p2, uint4& p3) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t w1[4], w2[4], w3[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    uint32_t a1, a2, a3, c1, c2, c3;
    et_split3(x[2 * m], a1, a2, a3);
    et_split3(x[2 * m + 1], c1, c2, c3);
    w1[m] = a1 | (c1 << 16);
    w2[m] = a2 | (c2 << 16);
    w3[m] = a3 | (c3 << 16);
  }
  p1 = make_uint4(w1[0], w1[1], w1[2], w1[3]);
  p2 = make_uint4(w2[0], w2[1], w2[2], w2[3]);
  p3 = make_uint4(w3[0], w3[1], w3[2], w3[3]);
}

// planes[(row * 3 + p) * D + d] = term p of X[row * D + d]: the item table as the split kernel stages it
// (6 D bytes per item, contiguous).  One thread per eight floats.
__global__ __launch_bounds__(kBlock) void et_split_rows_kernel(const float* __restrict__ X, int64_t groups, int D8,
                                                               uint4* __restrict__ planes) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= groups) return;
  const float4 lo = *reinterpret_cast<const float4*>(X + g * 8);
  const float4 hi = *reinterpret_cast<const float4*>(X + g * 8 + 4);
  uint4 p1, p2, p3;
  et_split3x8(lo, hi, p1, p2, p3);
  const int64_t row = g / D8;
  const int c = (int)(g % D8);
  planes[(row * 3 + 0) * D8 + c] = p1;
  planes[(row * 3 + 1) * D8 + c] = p2;
  planes[(row * 3 + 2) * D8 + c] = p3;
}

// A threshold from HINT lists (hint[r, 0..k): any k item ids per row — the top-k of an earlier evaluation of a
// model that has moved a little since): k DIFFERENT items whose scores are all >= b prove that the row's k-th best
// score is >= b, wherever the ids came from.  One lane group per row computes the k dot products in plain f32
// together with A = sum_d |u_d i_d| (+ |bias|), lowers each by 1.6e-5 A — more than this kernel's and the sweep's
// rounding of the same score can differ by (64 terms: <= 3.8e-6 A each) —, gives masked hints the mask value they
// have in the sweep, and writes the smallest: row_tau[r].  Out-of-range or repeated ids, or a bad user id: no
// bound (-inf).  A stale or arbitrary hint costs candidates, never correctness.
// Sixteen lanes per row, D / 16 floats of the rows each.  Lane l owns hint l: its binary search in the row's masked
// ids and its repeated-id test run once (not once per lane), the k dot products are shared work (47 us with every
// lane doing everything, hint after hint; 56 us with the k searches of a lane interleaved).
template <int D>
__global__ __launch_bounds__(kBlock) void et_hint_bound_kernel(
    const float* __restrict__ U, const float* __restrict__ I, const float* __restrict__ item_bias,
    const int64_t* __restrict__ users, int64_t nrows, int64_t num_users, int64_t num_items,
    const int64_t* __restrict__ mask_ptr, const int64_t* __restrict__ mask_idx, float mask_value,
    const int64_t* __restrict__ hint, int k, float* __restrict__ row_tau) {
  constexpr int G = 16, C = D / G, HPL = kEtMaxK / G;  // lane l owns hints l, l + 16, ...
  const int l = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
  const int64_t r = row < nrows ? row : nrows - 1;   // every lane group runs everything (shuffles below)
  const int64_t uid = users[r];
  const bool user_ok = (uint64_t)uid < (uint64_t)num_users;
  float uu[C];
#pragma unroll
  for (int c = 0; c < C; ++c) uu[c] = user_ok ? U[uid * D + C * l + c] : 0.0f;
  const int64_t m_lo = mask_ptr ? mask_ptr[r] : 0;
  const int len = mask_ptr ? (int)(mask_ptr[r + 1] - m_lo) : 0;
  const int64_t* mrow = mask_idx + m_lo;
  int my[HPL];
  bool masked[HPL];
  bool fine = true;                                  // every owned hint in range and not a repetition of an earlier one
#pragma unroll
  for (int t = 0; t < HPL; ++t) {
    const int slot = l + G * t;
    const int64_t h = slot < k ? hint[r * k + slot] : -1;
    my[t] = (uint64_t)h < (uint64_t)num_items ? (int)h : -1;          // num_items < 2^31
    fine = fine && (slot >= k || my[t] >= 0);
    int lo = 0, hi = my[t] >= 0 ? len : 0;           // my hint among the row's (ascending) masked ids?
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (mrow[mid] < my[t]) lo = mid + 1;
      else hi = mid;
    }
    masked[t] = my[t] >= 0 && lo < len && mrow[lo] == my[t];
  }
  float mys[HPL], mya[HPL];
#pragma unroll
  for (int t = 0; t < HPL; ++t) mys[t] = mya[t] = 0.0f;
  for (int j = 0; j < k; ++j) {
    const int idj = j < G ? __shfl(my[0], j, G) : __shfl(my[HPL - 1], j - G, G);
#pragma unroll
    for (int t = 0; t < HPL; ++t) fine = fine && !(j < l + G * t && l + G * t < k && idj == my[t]);
    float s = 0.0f, a = 0.0f;
    if (idj >= 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float x = uu[c] * I[(int64_t)idj * D + C * l + c];
        s += x;
        a += fabsf(x);
      }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      s += __shfl_xor(s, off, G);
      a += __shfl_xor(a, off, G);
    }
#pragma unroll
    for (int t = 0; t < HPL; ++t)
      if (l + G * t == j) { mys[t] = s; mya[t] = a; }
  }
  float eff = INFINITY;
#pragma unroll
  for (int t = 0; t < HPL; ++t) {
    const float b = item_bias && my[t] >= 0 ? item_bias[my[t]] : 0.0f;
    if (l + G * t < k) {
      const float e = masked[t] ? mask_value : (mys[t] + b) - 1.6e-5f * (mya[t] + fabsf(b));
      fine = fine && e == e;                         // a NaN score proves nothing (fminf would drop it silently)
      eff = fminf(eff, e);
    }
  }
  int all_fine = fine;                               // (taken after the NaN test above)
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    eff = fminf(eff, __shfl_xor(eff, off, G));
    all_fine &= __shfl_xor(all_fine, off, G);
  }
  if (row < nrows && l == 0) row_tau[row] = user_ok && k > 0 && all_fine ? eff : -INFINITY;
}

// items per LDS stage: 64 where three workgroups per CU fit with it (f32; SPLIT at D <= 32), else 32 — except SPLIT
// at D = 64 with 16- or 32-entry lists, whose registers allow two workgroups per CU either way
__host__ __device__ constexpr int et_chunk_items(int D, int KK, bool split) {
  return !split ? kEtChunkItems : (D <= 32 ? 64 : (D == 64 && KK > 10 ? 64 : YR_ET_SPLIT_CHUNK));
}

// BIAS: score = <user, item> + item_bias[item] — the decoder of CDAE, z . W_o[i] + b_o[i] (models/cdae.py:52), whose
// sigmoid is monotone, so the top-k of the pre-activations is the top-k of the predictions.  The bias enters as
// the initial value of the MFMA accumulator (staged through LDS with the item chunk): no extra instruction per score.
//
// SPLIT: the same f32 scores from bf16 matrix instructions.  Every operand is the sum of three bf16 terms
// (et_split3), every product the six partial products whose weight is 2^-18 or more — x1 y1, x1 y2, x2 y1, x2 y2,
// x1 y3, x3 y1; what is dropped is below 2^-25 |x y|, under the f32 rounding of the product itself — accumulated in
// f32 by v_mfma_f32_32x32x16_bf16: 6 x 4 instructions of 32 cycles per 32-item tile at D = 64 against 32 of 64 cycles
// (v_mfma_f32_32x32x2_f32).  The accumulator layout is the same, so is everything after it.  The items arrive
// pre-split (et_split_rows_kernel, `I` is then the plane table), the users are split once into registers.
//
// PRESCAN (a launch of its own before the sweep, `parts` workgroups per 128 users): the same scores for a strided
// sample of the stages only (every chunk_stride-th, dealt round-robin to the parts), and per lane the running maximum
// of each of its 16 accumulator registers — 16 disjoint groups of items, 32 per user and part, no candidates, no
// lists.  Masked items count with the mask value, as in the sweep.  The sweep's prologue reads the 32 x parts group
// maxima of its user: their k-th largest is reached by k different items, so it is a lower bound of the user's
// k-th best score over the whole catalogue, and every list of the user (both half-waves, every slice) starts with
// a threshold just below it instead of -inf.  What the lists then never see could not have ended in the top k.
// (Hint lists — et_hint_bound_kernel, row_tau — give a bound of the same kind from k rescored items instead.)
// ---- pieces both forms of the sweep share (forceinline: the lists and planes stay in the callers' registers) ----

// the user id of this lane's row (false: a row beyond the input or a bad id -> flagged, an all-zero operand)
__device__ __forceinline__ bool et_row_user(const int64_t* __restrict__ users, int64_t row, int64_t nrows,
                                            int64_t num_users, int32_t* __restrict__ err_flag, int64_t& uid) {
  bool ok = row < nrows;
  uid = ok ? users[row] : 0;
  if (ok && (uint64_t)uid >= (uint64_t)num_users) {
    if (err_flag) atomicOr(err_flag, YR_FLAG_BAD_USER);
    ok = false;
  }
  return ok;
}

// B operand of the split form: dims 16 kb + 8 h + j of every 16-deep block kb of the user's row, three bf16 planes
template <int D>
__device__ __forceinline__ void et_user_planes(const float* __restrict__ U, int64_t uid, bool ok, int h,
                                               uint4 (&us)[3][D / 16]) {
#pragma unroll
  for (int kb = 0; kb < D / 16; ++kb) {
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (ok) {
      lo = *reinterpret_cast<const float4*>(U + uid * D + 16 * kb + 8 * h);
      hi = *reinterpret_cast<const float4*>(U + uid * D + 16 * kb + 8 * h + 4);
    }
    et_split3x8(lo, hi, us[0][kb], us[1][kb], us[2][kb]);
  }
}

// The private list's start and the floor of its threshold.  Only the top k leave the kernel: the first KK - k places
// are held by phantom entries (+inf, no item), so the list's last score — the threshold — is the lane's k-th best,
// not its KK-th (+inf everywhere: nothing enters).  The floor comes from hint lists (row_tau) or from the prescan's
// group maxima (gmax: the k-th largest of them), strictly below the bound: scores equal to it must pass the strict test.
template <int KK>
__device__ __forceinline__ float et_start_lists(float (&Ls)[KK], int32_t (&Li)[KK], int k, bool ok, int64_t row,
                                                const float* __restrict__ row_tau, const float* __restrict__ gmax,
                                                int parts) {
#pragma unroll
  for (int e = 0; e < KK; ++e) { Ls[e] = (!ok || e < KK - k) ? INFINITY : -INFINITY; Li[e] = 0x7fffffff; }
  float tau0 = -INFINITY;
  if (row_tau && ok) {
    const float b = row_tau[row];
    tau0 = b < INFINITY ? b - fmaxf(fabsf(b) * 1.0e-6f, 1.0e-30f) : 3.0e38f;
  } else if (gmax && ok) {
    float T[KK];                                     // the largest group maxima of this lane's user, descending
#pragma unroll
    for (int e = 0; e < KK; ++e) T[e] = e < KK - k ? INFINITY : -INFINITY;   // the k-th largest ends up last (see Ls)
    const float4* g4 = reinterpret_cast<const float4*>(gmax + row * parts * 32);
    for (int q = 0; q < parts * 8; ++q) {
      const float4 g = g4[q];
      const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = gv[j];
#pragma unroll
        for (int e = 0; e < KK; ++e) {
          const bool sw = v > T[e];
          const float t = T[e];
          T[e] = sw ? v : t;
          v = sw ? t : v;
        }
      }
    }
    const float b = T[KK - 1];                       // -FLT_MAX and -inf give -inf
    tau0 = b < INFINITY ? b - fmaxf(fabsf(b) * 1.0e-6f, 1.0e-30f) : 3.0e38f;
  }
  return tau0;
}

// the two half-waves' lists of the same user (lane i <-> lane i + 32) merged, then the top k written: to the slice's
// partial lists, or to `out` (-1 for an empty place)
template <int KK>
__device__ __forceinline__ void et_finish_lists(float (&Ls)[KK], int32_t (&Li)[KK], int k, int h, int64_t row,
                                                int64_t nrows, TopEntry* __restrict__ partial,
                                                int64_t* __restrict__ out) {
  float Os[KK];
  int32_t Oi[KK];
#pragma unroll
  for (int e = 0; e < KK; ++e) {
    Os[e] = __shfl_xor(Ls[e], 32, kWave);
    Oi[e] = __shfl_xor(Li[e], 32, kWave);
  }
#pragma unroll
  for (int e = 0; e < KK; ++e) et_bubble<KK>(Ls, Li, Os[e], Oi[e], Oi[e] != 0x7fffffff);
  if (h == 0 && row < nrows) {
#pragma unroll
    for (int e = 0; e < KK; ++e) {
      const int o = e - (KK - k);                    // behind the phantom entries
      if (o >= 0) {
        if (partial) {
          TopEntry t;
          t.s = Ls[e];
          t.i = Li[e];
          partial[(row * gridDim.y + blockIdx.y) * k + o] = t;
        } else {
          out[row * k + o] = Li[e] == 0x7fffffff ? -1 : (int64_t)Li[e];
        }
      }
    }
  }
}

template <int D, int KK, bool BIAS, bool SPLIT, bool PRESCAN>
__global__ __launch_bounds__(kEtThreads) void mf_eval_topk_kernel(
    const float* __restrict__ U, const void* __restrict__ I_any, const float* __restrict__ item_bias,
    const int64_t* __restrict__ users, int64_t nrows,
    int64_t num_users, int num_items, const int64_t* __restrict__ mask_ptr, const int64_t* __restrict__ mask_idx,
    float mask_value, int k, int64_t* __restrict__ out, TopEntry* __restrict__ partial, int items_per_slice,
    float* __restrict__ gmax, int parts, int chunk_stride, const float* __restrict__ row_tau,
    int32_t* __restrict__ err_flag) {
  constexpr int HALF = D / 2;
  constexpr int KB = D / 16;                                      // SPLIT: 16-deep matrix instructions per plane pair
  constexpr int CH = et_chunk_items(D, KK, SPLIT);                // items per LDS stage
  // bytes per staged item: D + 4 floats, or three bf16 planes + 16 (an odd number of 16-byte units either way:
  // the 32 rows of a ds_read_b128 fall into different banks)
  constexpr int ROWB = SPLIT ? 6 * D + 16 : 4 * (D + 4);
  constexpr int ROW16 = SPLIT ? 3 * D / 8 : D / 4;                // 16-byte units of payload per item
  constexpr int NV = (CH * ROW16 + kEtThreads - 1) / kEtThreads;  // 16-byte units per thread per chunk
  __shared__ __attribute__((aligned(16))) unsigned char s_items[2][CH * ROWB];   // double-buffered
  // the split form with 32-item stages has LDS to spare under three workgroups per CU: 12 slots, the fill checked
  // twice per tile instead of four times (1.10 -> 1.05 ms cold, 0.85 -> 0.83 ms with hints)
  constexpr int CE = (SPLIT && CH == 32) ? 2 * kEtCheckEvery : kEtCheckEvery;
  constexpr int BUFCAP = kEtFlushAt + CE;
  __shared__ TopEntry s_buf[BUFCAP][kEtThreads];                  // slot-major: conflict-free per slot
  __shared__ __attribute__((aligned(16))) float s_bias[2][BIAS ? CH : 4];
  const uint4* __restrict__ I16 = static_cast<const uint4*>(I_any);   // f32 rows or plane rows, 16 bytes at a time

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int i = lane & 31, h = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * kEtUsers + wave * kEtUsersPerWave + i;   // this lane's user row
  // this workgroup's slice of the catalogue, stage after stage; PRESCAN: the stages blockIdx.y, blockIdx.y + parts,
  // ... of the sample
  const int64_t step = PRESCAN ? (int64_t)parts * chunk_stride * CH : CH;
  const int item_lo = PRESCAN ? blockIdx.y * chunk_stride * CH : blockIdx.y * items_per_slice;
  const int item_hi = PRESCAN ? num_items : min(num_items, item_lo + items_per_slice);

  // B operand: this lane's half of its user's row (zeros for rows beyond the input / bad ids) — f32: dims
  // [h D/2, (h+1) D/2); SPLIT: dims 16 kb + 8 h + j of every 16-deep block kb, three planes
  float ub[SPLIT ? 1 : HALF];
  uint4 us[3][KB];                                   // (unused, and removed by the compiler, in the f32 form)
  int64_t uid;
  const bool ok = et_row_user(users, row, nrows, num_users, err_flag, uid);
  if constexpr (SPLIT) {
    et_user_planes<D>(U, uid, ok, h, us);
  } else {
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(U + uid * D + h * HALF + 4 * q);
      ub[4 * q + 0] = v.x; ub[4 * q + 1] = v.y; ub[4 * q + 2] = v.z; ub[4 * q + 3] = v.w;
    }
  }

  // private list, threshold, buffer fill and mask cursor
  float Ls[KK];
  int32_t Li[KK];
  const float tau0 = et_start_lists<KK>(Ls, Li, k, ok, row, PRESCAN ? nullptr : row_tau, PRESCAN ? nullptr : gmax, parts);
  float tau = fmaxf(Ls[KK - 1], tau0);
  float gm[PRESCAN ? 16 : 1];                        // PRESCAN: running maxima of the 16 accumulator registers
#pragma unroll
  for (int e = 0; e < (PRESCAN ? 16 : 1); ++e) gm[e] = -INFINITY;
  int cnt = 0;
  const bool lazy_mask = mask_value <= -3.0e38f;     // uniform
  // the next TWO masked item ids stay in registers: the load that refills the second one is issued
  // a tile (or more) before its value is needed, so the sweep never waits on it
  int64_t m_cur = 0, m_end = 0;
  int next_masked = 0x7fffffff, after_next = 0x7fffffff;
  if (mask_ptr && row < nrows) {
    m_cur = mask_ptr[row];
    m_end = mask_ptr[row + 1];
    while (m_cur < m_end && mask_idx[m_cur] < item_lo) ++m_cur;   // masks below this slice
    if (m_cur < m_end) next_masked = (int)mask_idx[m_cur];
    if (m_cur + 1 < m_end) after_next = (int)mask_idx[m_cur + 1];
  }

  auto flush = [&]() {
    for (int j = 0; __ballot(j < cnt) != 0ull; ++j) {              // wave-uniform trip count
      const bool live = j < cnt;
      const float cs = s_buf[j][threadIdx.x].s;
      const int32_t ci = s_buf[j][threadIdx.x].i;
      et_bubble<KK>(Ls, Li, cs, ci, live);
    }
    cnt = 0;
    tau = fmaxf(Ls[KK - 1], tau0);
  };

  // register staging of the item chunks: the loads of chunk c+1 are issued before the tiles of
  // chunk c are computed and land in LDS after the next barrier (global latency hidden under MFMA)
  uint4 stage[NV];
  float stage_b = 0.0f;
  auto fetch = [&](int c0) {
    if (BIAS && threadIdx.x < CH)
      stage_b = c0 + (int)threadIdx.x < item_hi ? item_bias[c0 + threadIdx.x] : 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = threadIdx.x + v * kEtThreads;
      const int r = q / ROW16, c = q % ROW16;
      stage[v] = make_uint4(0u, 0u, 0u, 0u);
      if ((CH * ROW16 % kEtThreads == 0 || q < CH * ROW16) && c0 + r < item_hi)
        stage[v] = I16[(int64_t)(c0 + r) * ROW16 + c];
    }
  };
  auto stash = [&](unsigned char* dst, int buf) {
    if (BIAS && threadIdx.x < CH) s_bias[buf][threadIdx.x] = stage_b;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = threadIdx.x + v * kEtThreads;
      if (CH * ROW16 % kEtThreads == 0 || q < CH * ROW16)
        *reinterpret_cast<uint4*>(dst + (q / ROW16) * ROWB + 16 * (q % ROW16)) = stage[v];
    }
  };
#if YR_ET_PRIO == 3
  switch ((blockIdx.x + 5 * blockIdx.y) % 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
#endif
  [[maybe_unused]] long long ph_mfma = 0, ph_mask = 0, ph_epi = 0, ph_flush = 0, ph_sync = 0, ph_total = ET_CLK();
  fetch(item_lo);
  stash(s_items[0], 0);
  __syncthreads();
  int cur = 0;
  for (int c0 = item_lo; c0 < item_hi; c0 += (int)step) {
    const bool more = c0 + step < item_hi;
#ifndef YR_ET_EXP_NOSTAGE
    if (more) fetch(c0 + (int)step);                 // lands in the other buffer at the end of this chunk
#endif
    const unsigned char* chunk = s_items[cur];

#pragma unroll 1
    for (int t = 0; t < CH / 32; ++t) {
      const int item0 = c0 + t * 32;
      if (item0 >= item_hi) break;                   // wave-uniform
      // ---- scores: acc[reg] = <item item0 + row(reg, h), user of this lane>
      const long long t_a = ET_CLK();
#if YR_ET_PRIO == 1
      __builtin_amdgcn_s_setprio(3);
#elif YR_ET_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (BIAS) {                                    // accumulator register 4 g + j holds item 8 g + 4 h + j of the tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[cur][t * 32 + 8 * g + 4 * h]);
          acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
        }
      }
      if constexpr (SPLIT) {
        // item row i of the tile, dims 16 kb + 8 h + j: one ds_read_b128 per plane and block; the small terms first
        const unsigned char* src = chunk + (t * 32 + i) * ROWB + 16 * h;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(src + 32 * kb);
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(src + 2 * D + 32 * kb);
          const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(src + 4 * D + 32 * kb);
          const bf16x8 u1 = __builtin_bit_cast(bf16x8, us[0][kb]);
          const bf16x8 u2 = __builtin_bit_cast(bf16x8, us[1][kb]);
          const bf16x8 u3 = __builtin_bit_cast(bf16x8, us[2][kb]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, u1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, u3, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, u2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, u1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, u2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, u1, acc, 0, 0, 0);
        }
      } else {
        const float* src = reinterpret_cast<const float*>(chunk) + (t * 32 + i) * (D + 4) + h * HALF;
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
          const float4 b = *reinterpret_cast<const float4*>(src + 4 * q);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, ub[4 * q + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, ub[4 * q + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, ub[4 * q + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, ub[4 * q + 3], acc, 0, 0, 0);
        }
      }
#ifdef YR_ET_STAMPS
      asm volatile("" ::"v"(acc[0]));                 // the scores have arrived
#endif
#if YR_ET_PRIO == 1
      __builtin_amdgcn_s_setprio(0);
#elif YR_ET_PRIO == 2
      __builtin_amdgcn_s_setprio(3);
#endif
      const long long t_b = ET_CLK();
      ph_mfma += t_b - t_a;
      // ---- masked items of this tile (bit r = item item0 + r), shifted to this half's rows
      uint32_t bits = 0;
#ifdef YR_ET_EXP_NOMASK      // timing experiment only (wrong results): the mask lists are ignored
      next_masked = 0x7fffffff;
#endif
      if constexpr (PRESCAN) {
        // the sample skips most of the catalogue: jump over the entries below this tile eight at a time (the walk
        // below is one dependent load per entry)
        if (next_masked < item0) {
          const int64_t before = m_cur;
          while (m_cur + 8 < m_end && (int)mask_idx[m_cur + 8] < item0) m_cur += 8;
          if (m_cur != before) {
            next_masked = (int)mask_idx[m_cur];
            after_next = m_cur + 1 < m_end ? (int)mask_idx[m_cur + 1] : 0x7fffffff;
          }
        }
      }
      while (next_masked < item0 + 32) {
        if (next_masked >= item0) bits |= 1u << (next_masked - item0);
        ++m_cur;
        next_masked = after_next;
        after_next = m_cur + 1 < m_end ? (int)mask_idx[m_cur + 1] : 0x7fffffff;
      }
      // With the reference's mask value (-FLT_MAX, below every real score) the mask is applied
      // lazily, inside the candidate branch only: a masked item whose real score does not beat the
      // threshold could not enter with -FLT_MAX either (the threshold is -inf, and then every
      // score is a candidate, or already >= -FLT_MAX).  Any other mask value rewrites the scores first.
      const long long t_c = ET_CLK();
      ph_mask += t_c - t_b;
      const uint32_t mine = bits >> (4 * h);
      if ((PRESCAN || !lazy_mask) && __ballot(bits != 0) != 0ull) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if ((mine >> ((reg & 3) + 8 * (reg >> 2))) & 1u) acc[reg] = mask_value;
      }
      if (item0 + 32 > item_hi) {                    // wave-uniform: last, partial tile of the slice
        const uint32_t beyond = (~0u << (item_hi - item0)) >> (4 * h);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if ((beyond >> ((reg & 3) + 8 * (reg >> 2))) & 1u) acc[reg] = -INFINITY;   // not an item
      }
      if constexpr (PRESCAN) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) gm[reg] = fmaxf(gm[reg], acc[reg]);
      }
      // ---- candidates -> private buffer.  Strict comparison is exact: a lane meets its items in
      // ascending id order, so a later score EQUAL to the threshold loses the tie anyway; it also
      // keeps the -inf of the rows beyond the slice out.
      // With thresholds from hint lists most tiles hold no candidate in the whole wave: the lane's largest score
      // against its threshold first (8 v_max3 + one compare instead of 16 compare-and-branch blocks).  With the
      // looser bounds of the prescan, or none, nearly every tile holds one, and the test would only add to it.
      bool scan = !PRESCAN;
#if YR_ET_EARLY_OUT
      if (!PRESCAN && row_tau) {                     // wave-uniform
        float mx = acc[0];
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) mx = fmaxf(mx, acc[reg]);
        scan = __ballot(mx > tau) != 0ull;
      }
#endif
      if (scan)
#pragma unroll
      for (int half = 0; half < 16 / CE; ++half) {
#pragma unroll
        for (int q = 0; q < CE; ++q) {
          const int reg = half * CE + q;
          float sc = acc[reg];
#ifdef YR_ET_EXP_NOSCAN      // timing experiment only (wrong results): nothing ever becomes a candidate
          if (sc == 12345.678f) {
#else
          if (sc > tau) {
#endif
            if (lazy_mask && ((mine >> ((reg & 3) + 8 * (reg >> 2))) & 1u)) sc = mask_value;
            if (sc > tau) {
              TopEntry c;
              c.s = sc;
              c.i = item0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
              s_buf[cnt][threadIdx.x] = c;
              ++cnt;
            }
          }
        }
        if (__ballot(cnt > kEtFlushAt) != 0ull) {
          const long long t_f = ET_CLK();
          flush();
          ph_flush += ET_CLK() - t_f;
        }
      }
      ph_epi += ET_CLK() - t_c;
    }
    const long long t_s = ET_CLK();
    // one barrier per chunk: everyone is done reading s_items[cur ^ 1] since the previous barrier,
    // so the next chunk can be written there while slower waves still read s_items[cur]
#ifdef YR_ET_EXP_NOSTAGE     // timing experiment only (wrong results): no global loads, no LDS writes, no barrier
#elif defined(YR_ET_EXP_NOBARRIER)   // timing experiment only (wrong results): the waves of a workgroup never wait for each other
    if (more) stash(s_items[cur ^ 1], cur ^ 1);
#else
    if (more) stash(s_items[cur ^ 1], cur ^ 1);
    __syncthreads();
#endif
    ph_sync += ET_CLK() - t_s;
    cur ^= 1;
  }
  if constexpr (PRESCAN) {
    if (row < nrows) {
      float4* g4 = reinterpret_cast<float4*>(gmax + (row * parts + blockIdx.y) * 32 + 16 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) g4[q] = make_float4(gm[4 * q], gm[4 * q + 1], gm[4 * q + 2], gm[4 * q + 3]);
    }
    return;
  }
  flush();
#ifdef YR_ET_STAMPS
  if (lane == 0) {
    atomicAdd(&g_et_phase[0], (unsigned long long)(ET_CLK() - ph_total));
    atomicAdd(&g_et_phase[1], (unsigned long long)ph_mfma);
    atomicAdd(&g_et_phase[2], (unsigned long long)ph_epi);
    atomicAdd(&g_et_phase[3], (unsigned long long)ph_flush);
    atomicAdd(&g_et_phase[4], (unsigned long long)ph_sync);
    atomicAdd(&g_et_phase[5], 1ull);
    atomicAdd(&g_et_phase[6], (unsigned long long)ph_mask);
  }
#endif

  et_finish_lists<KK>(Ls, Li, k, h, row, nrows, partial, out);
}

// ---- the sweep as TWO ROLES per SIMD ("ping-pong") ------------------------------------------------------------------
// A 512-thread workgroup = 256 eval users; waves w and w + 4 share a SIMD.  Waves 0-3 run ONE PHASE AHEAD of waves 4-7:
// while one half issues the 6 D/16 matrix instructions of a tile (operands already in registers: nothing but MFMAs
// between two barriers), the other half does everything else for ITS previous tile — mask walk, threshold test,
// candidates, its half of the global -> LDS staging two tiles ahead, and the LDS -> register reads of its next tile —
// so a SIMD's matrix pipe always has one wave feeding it and its vector issue always belongs to the other
// (`MI355X_MICROARCH.md`, "Two waves per SIMD").  One workgroup per CU, 256 registers per lane to spend.
//   interval n (between barriers n and n + 1):  waves 0-3: scores(t) in 2t, rest(t) in 2t + 1;  waves 4-7: one later
//   tile t lives in stage t & 1: written in intervals 2t - 3 (rows 0-15, waves 0-3) and 2t - 2 (rows 16-31, waves 4-7),
//   read in 2t - 1 and 2t; the next tile of that stage (t + 2) is written from 2t + 1 on.
// Same instruction order per tile as mf_eval_topk_kernel<.., SPLIT>: identical scores, identical lists.
#ifdef YR_PP_TRACE
// -DYR_PP_TRACE: shader-clock stamps of waves 0 and 4 of workgroup (1, 0) around every interval of tiles 100..163
__device__ long long g_pp_trace[2][64][8];
#define PP_STAMP(slot)                                                                               \
  if (blockIdx.x == 1 && blockIdx.y == 0 && (wave & 3) == 0 && lane == 0 && t >= 100 && t < 164)     \
    g_pp_trace[role][t - 100][slot] = clock64()
#else
#define PP_STAMP(slot)
#endif
constexpr int kPpWaves = 8;
constexpr int kPpThreads = kPpWaves * kWave;             // 512
constexpr int kPpUsers = kPpWaves * kEtUsersPerWave;     // 256 per workgroup
#ifndef YR_PP_PRIO
#define YR_PP_PRIO 0          // experiments: 1 = the scores interval at s_setprio 3, 2 = the rest interval
#endif
#ifndef YR_PP_TILES_D64
#define YR_PP_TILES_D64 2     // tiles per interval at D = 64 with lists up to 10 entries (two: the matrix interval as
                              // long as at D = 128; with 16-entry lists the second accumulator spills)
#endif
#ifndef YR_PP_BUFCAP
#define YR_PP_BUFCAP 6
#endif

template <int D, int KK, bool BIAS, int TILES>
__global__ __launch_bounds__(kPpThreads) void mf_eval_topk_pp_kernel(
    const float* __restrict__ U, const void* __restrict__ I_any, const float* __restrict__ item_bias,
    const int64_t* __restrict__ users, int64_t nrows, int64_t num_users, int num_items,
    const int64_t* __restrict__ mask_ptr, const int64_t* __restrict__ mask_idx, float mask_value, int k,
    int64_t* __restrict__ out, TopEntry* __restrict__ partial, int items_per_slice, const float* __restrict__ gmax,
    int parts, const float* __restrict__ row_tau, int32_t* __restrict__ err_flag) {
  constexpr int KB = D / 16;
  constexpr int ROWB = 6 * D + 16;                      // three bf16 planes + 16: rows of a ds_read_b128 in different banks
  constexpr int ROW16 = 3 * D / 8;                      // 16-byte units of payload per item
  // An interval covers TILES 32-item tiles (a "step": 32 TILES items, one accumulator per tile): two at D = 64, so that
  // the matrix interval (48 instructions either way) is long enough to cover the partner's.
  constexpr int MT = 32 * TILES;                        // items per step
  constexpr int NB = KB * TILES;                        // 16-deep operand blocks per step
  constexpr int HALF16 = MT / 2 * ROW16;                // 16-byte units of the MT / 2 rows a role stages
  constexpr int NV = (HALF16 + 255) / 256;
  constexpr int BUFCAP = YR_PP_BUFCAP;                  // candidate slots per lane; flushed when some lane holds more than 4
  static_assert(BUFCAP > kEtFlushAt, "candidate buffer");
  // Up to four blocks: all of a step's operands are read into registers in the rest interval.  Eight blocks (D = 128;
  // D = 64 with two tiles): two in the rest interval, the others two at a time UNDER the matrix instructions of the
  // two before; the step is then still being read in the interval after, so it lives in a ring of three stages.
  constexpr int PRE = NB <= 4 ? NB : 2;
  constexpr int NBUF = PRE < NB ? 3 : 2;
  __shared__ __attribute__((aligned(16))) unsigned char s_items[NBUF][MT * ROWB];
  __shared__ TopEntry s_buf[BUFCAP][kPpThreads];
  __shared__ __attribute__((aligned(16))) float s_bias[NBUF][BIAS ? MT : 4];
  const uint4* __restrict__ I16 = static_cast<const uint4*>(I_any);

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int role = wave >> 2;                           // wave-uniform
  const int tr = threadIdx.x & 255;                     // thread within the role
  const int i = lane & 31, h = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * kPpUsers + wave * kEtUsersPerWave + i;
  const int item_lo = blockIdx.y * items_per_slice;
  const int item_hi = min(num_items, item_lo + items_per_slice);
  const int T = (item_hi - item_lo + MT - 1) / MT;      // steps of this slice (uniform over the workgroup)

  uint4 us[3][KB];                                      // B operand: the user's row, three planes
  int64_t uid;
  const bool ok = et_row_user(users, row, nrows, num_users, err_flag, uid);
  et_user_planes<D>(U, uid, ok, h, us);
  float Ls[KK];
  int32_t Li[KK];
  const float tau0 = et_start_lists<KK>(Ls, Li, k, ok, row, row_tau, gmax, parts);
  float tau = fmaxf(Ls[KK - 1], tau0);
  int cnt = 0;
  const bool lazy_mask = mask_value <= -3.0e38f;
  int64_t m_cur = 0, m_end = 0;
  int next_masked = 0x7fffffff, after_next = 0x7fffffff;
  if (mask_ptr && row < nrows) {
    m_cur = mask_ptr[row];
    m_end = mask_ptr[row + 1];
    while (m_cur < m_end && mask_idx[m_cur] < item_lo) ++m_cur;
    if (m_cur < m_end) next_masked = (int)mask_idx[m_cur];
    if (m_cur + 1 < m_end) after_next = (int)mask_idx[m_cur + 1];
  }

  auto flush = [&]() {
    for (int j = 0; __ballot(j < cnt) != 0ull; ++j) {
      const bool live = j < cnt;
      const float cs = s_buf[j][threadIdx.x].s;
      const int32_t ci = s_buf[j][threadIdx.x].i;
      et_bubble<KK>(Ls, Li, cs, ci, live);
    }
    cnt = 0;
    tau = fmaxf(Ls[KK - 1], tau0);
  };

  // this role's 16 rows of a tile: global -> registers (one rest phase ahead), registers -> LDS.  No branches: rows
  // beyond the slice read its last row (rest() turns their scores into -inf), and at D = 64, where a role's 384
  // 16-byte units are one and a half per thread, threads 128-255 repeat the second unit of threads 0-127 (same
  // bytes to the same place from another wave).
  uint4 stage[NV];
  float stage_b = 0.0f;
  static_assert((HALF16 % 256 == 0 || HALF16 % 256 == 128) && NV <= 3, "units per thread");
  auto unit = [&](int v) {                              // (loop invariant: the compiler keeps it in registers)
    const int q = tr + v * 256;
    return (HALF16 % 256 != 0 && q >= HALF16) ? q - 128 : q;
  };
  auto fetch = [&](int tile) {
    const int c0 = item_lo + MT * tile + MT / 2 * role;
    if (BIAS) stage_b = item_bias[min(c0 + (tr & (MT / 2 - 1)), item_hi - 1)];
    auto load = [&](int v) {
      const int q = unit(v);
      return I16[(int64_t)min(c0 + q / ROW16, item_hi - 1) * ROW16 + q % ROW16];
    };
    stage[0] = load(0);                                 // (constant indices: a loop here leaves `stage` in scratch)
    if constexpr (NV > 1) stage[1] = load(1);
    if constexpr (NV > 2) stage[2] = load(2);
  };
  auto stash = [&](int tile) {
    unsigned char* dst = s_items[tile % NBUF] + MT / 2 * role * ROWB;
    if (BIAS && tr < MT / 2) s_bias[tile % NBUF][MT / 2 * role + tr] = stage_b;
    auto store = [&](int v, const uint4 x) {
      const int q = unit(v);
      *reinterpret_cast<uint4*>(dst + (q / ROW16) * ROWB + 16 * (q % ROW16)) = x;
    };
    store(0, stage[0]);
    if constexpr (NV > 1) store(1, stage[1]);
    if constexpr (NV > 2) store(2, stage[2]);
  };

  // A operand of the next step: block b = tile b / KB of the step, dims 16 (b % KB) + 8 h + j of item row i, three
  // planes; blocks [0, PRE) here, in the rest interval; and the accumulators' start
  constexpr int SETB = PRE == NB ? (NB + 1) / 2 : 2;    // blocks per register set (two sets)
  bf16x8 A[2][3][SETB];
  f32x16 acc[TILES];
  auto blocks = [&](const unsigned char* stage_base, int set, int b0) {
#pragma unroll
    for (int j = 0; j < SETB; ++j) {
      const int b = b0 + j;
      if (b < NB) {
        const unsigned char* src = stage_base + (b / KB * 32 + i) * ROWB + 16 * h + 32 * (b % KB);
        A[set][0][j] = *reinterpret_cast<const bf16x8*>(src);
        A[set][1][j] = *reinterpret_cast<const bf16x8*>(src + 2 * D);
        A[set][2][j] = *reinterpret_cast<const bf16x8*>(src + 4 * D);
      }
    }
  };
  constexpr bool STREAM = PRE < NB;                     // operands streamed under the matrix instructions (ring of 3)
  auto first_blocks = [&](int tile) {                   // blocks [0, PRE) of step `tile` into set 0 (and 1)
    const unsigned char* base = s_items[tile % NBUF];
    blocks(base, 0, 0);
    if constexpr (!STREAM) blocks(base, 1, SETB);
  };
  // the accumulators' start: the decoder bias; without bias the first product of a tile takes the constant 0 instead
  auto start_acc = [&](int tile) {
    if constexpr (BIAS) {                              // accumulator register 4 g + q holds item 8 g + 4 h + q of its tile
#pragma unroll
      for (int j = 0; j < TILES; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(&s_bias[tile % NBUF][32 * j + 8 * g + 4 * h]);
          acc[j][4 * g + 0] = b4.x; acc[j][4 * g + 1] = b4.y; acc[j][4 * g + 2] = b4.z; acc[j][4 * g + 3] = b4.w;
        }
      }
    }
  };
  auto products = [&](int set, int b0) {               // the small terms first, as in the four-wave kernel
#pragma unroll
    for (int j = 0; j < SETB; ++j) {
      const int b = b0 + j;
      if (b < NB) {
        const bf16x8 u1 = __builtin_bit_cast(bf16x8, us[0][b % KB]);
        const bf16x8 u2 = __builtin_bit_cast(bf16x8, us[1][b % KB]);
        const bf16x8 u3 = __builtin_bit_cast(bf16x8, us[2][b % KB]);
        f32x16& c = acc[b / KB];
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[set][2][j], u1, (!BIAS && b % KB == 0) ? zero : c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[set][0][j], u3, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[set][1][j], u2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[set][1][j], u1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[set][0][j], u2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[set][0][j], u1, c, 0, 0, 0);
      }
    }
  };
  auto scores = [&](int tile) {
    if constexpr (PRE == NB) {
      products(0, 0);
      products(1, SETB);
    } else {
      const unsigned char* base = s_items[tile % NBUF];
#pragma unroll
      for (int g = 0; g < NB / SETB; ++g) {
        if (g + 1 < NB / SETB) blocks(base, (g + 1) & 1, SETB * (g + 1));   // the next two blocks under these products
        __builtin_amdgcn_sched_barrier(0);
        products(g & 1, SETB * g);
        // in the issue slots the dependent matrix instructions leave free: the staging (branch-free: steps beyond
        // the slice re-read its last row into stages nobody reads) and, at the end, the first blocks of the next step
        if (g == 0) { stash(tile + 2); fetch(tile + 3); }
        if (g == NB / SETB - 1) first_blocks(tile + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // mask, threshold test, candidates of step `tile` (the scores are in acc[0 .. TILES))
  auto rest = [&](int tile) {
    const int step0 = item_lo + MT * tile;
    uint64_t bits = 0;
#ifdef YR_PP_EXP_NOWALK
    next_masked = 0x7fffffff;
#endif
    while (next_masked < step0 + MT) {
      if (next_masked >= step0) bits |= 1ull << (next_masked - step0);
      ++m_cur;
      next_masked = after_next;
      after_next = m_cur + 1 < m_end ? (int)mask_idx[m_cur + 1] : 0x7fffffff;
    }
    if (!lazy_mask && __ballot(bits != 0) != 0ull) {
#pragma unroll
      for (int j = 0; j < TILES; ++j) {
        const uint32_t mine = (uint32_t)(bits >> (32 * j)) >> (4 * h);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if ((mine >> ((reg & 3) + 8 * (reg >> 2))) & 1u) acc[j][reg] = mask_value;
      }
    }
    if (step0 + MT > item_hi) {                        // wave-uniform: last, partial step of the slice
#pragma unroll
      for (int j = 0; j < TILES; ++j) {
        const int valid = min(32, max(0, item_hi - (step0 + 32 * j)));
        const uint32_t beyond = (valid >= 32 ? 0u : ~0u << valid) >> (4 * h);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if ((beyond >> ((reg & 3) + 8 * (reg >> 2))) & 1u) acc[j][reg] = -INFINITY;
      }
    }
    float mx = acc[0][0];
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) mx = fmaxf(mx, acc[j][reg]);
    }
#ifdef YR_PP_EXP_NOSCAN
    asm volatile("" ::"v"(mx));
    return;
#endif
    if (__ballot(mx > tau) == 0ull) return;            // no lane of the wave has a candidate in this step
    // The four waves of a role wait for each other every interval, so this must be short in the common case: which of
    // the scores pass (a bit each), then one candidate per lane and round, lowest register first (ascending item
    // ids, as the strict comparison wants) — one round with thresholds from hint lists, and ONE place that flushes.
    // (Wave-uniform register masks + a uniform register index instead: the same with hints, 20-35 % slower without.)
#pragma unroll
    for (int j = 0; j < TILES; ++j) {
      const uint32_t mine = (uint32_t)(bits >> (32 * j)) >> (4 * h);
      uint32_t pass = 0;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) pass |= acc[j][reg] > tau ? 1u << reg : 0u;
      while (__ballot(pass != 0) != 0ull) {
        if (pass != 0) {
          const int reg = __ffs((int)pass) - 1;
          pass &= pass - 1;
          float sc = acc[j][0];
#pragma unroll
          for (int q = 1; q < 16; ++q) sc = reg == q ? acc[j][q] : sc;
          const int pos = (reg & 3) + 8 * (reg >> 2);
          if (lazy_mask && ((mine >> pos) & 1u)) sc = mask_value;
          if (sc > tau) {                              // (tau may have risen in a flush since `pass` was taken)
            TopEntry c;
            c.s = sc;
            c.i = step0 + 32 * j + pos + 4 * h;
            s_buf[cnt][threadIdx.x] = c;
            ++cnt;
          }
        }
        if (__ballot(cnt > kEtFlushAt) != 0ull) flush();
      }
    }
  };

  // tiles 0 and 1 into LDS, tile 2 on its way; the first operands
  if (T > 0) { fetch(0); stash(0); }
  if (T > 1) { fetch(1); stash(1); }
  if (T > 2) fetch(2);
  __syncthreads();
  if (T > 0) { first_blocks(0); start_acc(0); }
  if (role == 1) __syncthreads();                      // waves 4-7 run one interval behind
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    PP_STAMP(0);
#if YR_PP_PRIO == 1
    __builtin_amdgcn_s_setprio(3);
#elif YR_PP_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
#ifndef YR_PP_EXP_NOMFMA     // (YR_PP_EXP_*: timing experiments only, wrong results)
    scores(t);
#endif
#if YR_PP_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#elif YR_PP_PRIO == 2
    __builtin_amdgcn_s_setprio(3);
#endif
    PP_STAMP(1);
    // the matrix instructions are register-only: without this the scheduler moves 23 of the 24 BELOW the barrier,
    // into the interval that belongs to the partner's
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(2);
#ifndef YR_PP_EXP_NOSTAGE
    if (!STREAM && t + 2 < T) stash(t + 2);
#endif
    PP_STAMP(5);
#ifndef YR_PP_EXP_NOREST
    rest(t);
#else
    asm volatile("" ::"v"(acc[0][0]), "v"(acc[TILES - 1][15]));   // (the scores stay computed)
#endif
    PP_STAMP(6);
#ifndef YR_PP_EXP_NOOPERANDS
    if (t + 1 < T) {
      if (!STREAM) first_blocks(t + 1);
      start_acc(t + 1);
    }
#endif
    PP_STAMP(7);
    // the global loads LAST: the mask walk waits with vmcnt(0) for its own (older) load and would wait for these too
#ifndef YR_PP_EXP_NOSTAGE
    if (!STREAM && t + 3 < T) fetch(t + 3);
#endif
    PP_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(4);
  }
  if (role == 0) __syncthreads();
  flush();
  et_finish_lists<KK>(Ls, Li, k, h, row, nrows, partial, out);
}

// out[r, :] = the k best of the S sorted partial lists of row r (score descending, item ascending
// among equal scores; empty slots carry item 0x7fffffff and lose every comparison).  One thread per
// row: S cursors, k rounds.
// up to 24 slices (a user shard of an 8-GPU run has 31 row blocks: 24 x 31 workgroups; 8 slices: 3,958 users with
// hints 0.28 -> 0.24 ms, 7,917 users 0.33 -> 0.29 ms)
#ifndef YR_ET_MAX_SLICES
#define YR_ET_MAX_SLICES 24
#endif
constexpr int kEtMaxSlices = YR_ET_MAX_SLICES;

__global__ __launch_bounds__(kBlock) void mf_eval_merge_kernel(const TopEntry* __restrict__ partial, int64_t nrows,
                                                               int S, int k, int64_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nrows) return;
  const TopEntry* P = partial + r * S * k;
  int cur[kEtMaxSlices];
#pragma unroll
  for (int s = 0; s < kEtMaxSlices; ++s) cur[s] = 0;
  for (int e = 0; e < k; ++e) {
    int best = -1;
    float bs = 0.0f;
    int32_t bi = 0x7fffffff;
    for (int s = 0; s < S; ++s) {
      if (cur[s] >= k) continue;
      const TopEntry t = P[s * k + cur[s]];
      if (t.i == 0x7fffffff) continue;
      if (best < 0 || et_better(t.s, t.i, bs, bi)) { best = s; bs = t.s; bi = t.i; }
    }
    if (best >= 0) {
#pragma unroll
      for (int s = 0; s < kEtMaxSlices; ++s) cur[s] += (s == best);
    }
    out[r * k + e] = best >= 0 ? (int64_t)bi : -1;
  }
}

}  // namespace yr

using namespace yr;

// slices of the catalogue per 128-user workgroup row: about three workgroups per CU in one round
static int et_slices(int64_t nrows, int64_t num_items) {
  const int64_t rows = (nrows + kEtUsers - 1) / kEtUsers;
  int64_t S = YR_ET_TARGET_WGS / rows;
  const int64_t by_items = num_items / 2048;          // keep slices long enough to amortise their top-k
  if (S > by_items) S = by_items;
  if (S > kEtMaxSlices) S = kEtMaxSlices;
  return (int)(S < 1 ? 1 : S);
}

// The two-role sweep (mf_eval_topk_pp_kernel): 256-user workgroups, ONE per CU, so at most 256 of them; split form,
// D = 64 with lists up to 16 entries, D = 128 up to 10.  The library's rule, from 2,048 rows: D = 128 always (sweep
// 1.90 -> 1.55 ms hinted, whole call cold 2.34 -> 1.78 at Yelp2018 size); D = 64 with 16-entry lists and thresholds
// from hint lists (1.27 -> 1.14 ms); at D = 64 with lists up to 10 entries the two forms take the same time with
// hints (0.80 vs 0.82 ms) and the four-wave form is faster without (1.12 vs 1.23: more candidates, and every wave of
// a role waits for the slowest), so it stays.  YR_EVAL_TWO_ROLES / YR_EVAL_FOUR_WAVES force a form.
static bool et_pp_wanted(int64_t nrows, int D, int k, int mode, bool hinted) {
  if (!(mode & YR_EVAL_BF16X3) || (mode & YR_EVAL_FOUR_WAVES)) return false;
  const bool exists = (D == 64 && k <= 16) || (D == 128 && k <= 10);
  if (mode & YR_EVAL_TWO_ROLES) return exists;
  return exists && nrows >= 2048 && (D == 128 || (hinted && k > 10));
}
static int et_pp_slices(int64_t nrows, int64_t num_items) {
  const int64_t rows = (nrows + kPpUsers - 1) / kPpUsers;
  int64_t S = 256 / rows;
  const int64_t by_items = num_items / 2048;
  if (S > by_items) S = by_items;
  if (S > kEtMaxSlices) S = kEtMaxSlices;
  return (int)(S < 1 ? 1 : S);
}

// Workspace, in this order: the item planes (YR_EVAL_BF16X3), the row thresholds of hint lists, the group maxima
// of the prescan, the partial lists
static int64_t et_plane_bytes(int64_t num_items, int D) { return (num_items * 6 * D + 255) / 256 * 256; }
static int64_t et_gmax_bytes(int64_t nrows, int parts) { return (nrows * parts * 32 * 4 + 255) / 256 * 256; }
static int64_t et_tau_bytes(int64_t nrows) { return (nrows * 4 + 255) / 256 * 256; }

#ifndef YR_ET_SAMPLE_ITEMS
#define YR_ET_SAMPLE_ITEMS 4096
#endif
// The prescan scores an eighth of the catalogue per user, 4,096 items at most (a prescan stage costs about what a
// stage of the sweep costs: Yelp2018 size, sweep alone -> with 1,024 / 2,048 / 4,096 sampled items: k = 10
// 1.20 -> 1.19 / 1.16 / 1.11 ms, k = 16 2.63 -> 2.44 / 2.30 / 2.13 ms, k = 4 0.88 -> 0.92 ms whatever the sample).
// Worth its launch from 16,384 items and lists of more than four entries.
constexpr int kEtSampleItems = YR_ET_SAMPLE_ITEMS;
constexpr int64_t kEtPrescanMinItems = 16384;
static bool et_prescan_wanted(int64_t num_items, int k, int mode) {
  if (mode & YR_EVAL_NO_PRESCAN) return false;
  return (mode & YR_EVAL_FORCE_PRESCAN) || (num_items >= kEtPrescanMinItems && k > 4);
}
static bool et_mode_ok(int mode) {
  return (mode & ~(YR_EVAL_BF16X3 | YR_EVAL_NO_PRESCAN | YR_EVAL_FORCE_PRESCAN | YR_EVAL_TWO_ROLES | YR_EVAL_FOUR_WAVES)) == 0 &&
         (mode & (YR_EVAL_TWO_ROLES | YR_EVAL_FOUR_WAVES)) != (YR_EVAL_TWO_ROLES | YR_EVAL_FOUR_WAVES) &&
         (mode & (YR_EVAL_NO_PRESCAN | YR_EVAL_FORCE_PRESCAN)) != (YR_EVAL_NO_PRESCAN | YR_EVAL_FORCE_PRESCAN);
}
static bool et_dim_ok(int D) { return D == 16 || D == 32 || D == 64 || D == 128; }

extern "C" int64_t yr_mf_eval_topk_planes_bytes(int64_t num_items, int D) {
  if (num_items <= 0 || !et_dim_ok(D)) return YR_ERR_BADARG;
  return et_plane_bytes(num_items, D);
}

extern "C" int64_t yr_mf_eval_topk_workspace_bytes(int64_t nrows, int64_t num_items, int D, int k, int mode) {
  if (nrows < 0 || num_items <= 0 || k <= 0 || k > kEtMaxK || !et_mode_ok(mode) || !et_dim_ok(D)) return YR_ERR_BADARG;
  const int S = et_slices(nrows, num_items);
  return ((mode & YR_EVAL_BF16X3) ? et_plane_bytes(num_items, D) : 0) + et_tau_bytes(nrows) +
         (et_prescan_wanted(num_items, k, mode) ? et_gmax_bytes(nrows, S) : 0) +
         (S > 1 ? nrows * S * k * (int64_t)sizeof(TopEntry) : 0);
}

namespace {
struct EtArgs {
  const float* U;
  const void* items;
  const float* item_bias;
  const int64_t* users;
  int64_t nrows, num_users;
  int num_items;
  const int64_t *mask_ptr, *mask_idx;
  float mask_value;
  int k;
  int64_t* out;
  TopEntry* partial;
  int per;
  float* gmax;          // NULL: no prescan
  int parts;
  const float* row_tau; // NULL: no hint lists
  int32_t* err_flag;
  unsigned row_blocks, slices;
  bool two_roles;       // the sweep as mf_eval_topk_pp_kernel (row blocks of 256)
};

template <int DD, int KK, bool BB, bool SS, bool PP>
void et_launch_one(const EtArgs& a, dim3 grid, int chunk_stride, hipStream_t s) {
  hipLaunchKernelGGL((mf_eval_topk_kernel<DD, KK, BB, SS, PP>), grid, dim3(kEtThreads), 0, s, a.U, a.items, a.item_bias,
                     a.users, a.nrows, a.num_users, a.num_items, a.mask_ptr, a.mask_idx, a.mask_value, a.k, a.out,
                     a.partial, a.per, a.gmax, a.parts, chunk_stride, a.row_tau, a.err_flag);
}

template <int DD, bool BB, bool SS>
void et_launch(const EtArgs& a, hipStream_t s) {
  if (a.gmax) {
    // every stride-th stage of the catalogue, dealt to `parts` workgroups
    constexpr int CHP = et_chunk_items(DD, 4, SS);
    const int chunks = (a.num_items + CHP - 1) / CHP;
    const int sample = std::min(kEtSampleItems, std::max(CHP, a.num_items / 8));
    const int stride = std::max(1, chunks / std::max(1, sample / CHP));
    et_launch_one<DD, 4, BB, SS, true>(a, dim3(a.row_blocks, (unsigned)a.parts), stride, s);
  }
  if constexpr (SS && (DD == 64 || DD == 128)) {
    if (a.two_roles) {
      const dim3 grid2((unsigned)((a.nrows + kPpUsers - 1) / kPpUsers), a.slices);
#define YR_ET_PP(KK)                                                                                                  \
  hipLaunchKernelGGL((mf_eval_topk_pp_kernel<DD, KK, BB, (DD == 64 && KK <= 10) ? YR_PP_TILES_D64 : 1>), grid2, dim3(kPpThreads), 0, s, a.U, a.items, a.item_bias, \
                     a.users, a.nrows, a.num_users, a.num_items, a.mask_ptr, a.mask_idx, a.mask_value, a.k, a.out,     \
                     a.partial, a.per, a.gmax, a.parts, a.row_tau, a.err_flag)
      if (a.k <= 4) YR_ET_PP(4);
      else if (a.k <= 10) YR_ET_PP(10);
      else if constexpr (DD == 64) YR_ET_PP(16);       // (D = 128 with 16-entry lists spills: not offered, see et_pp_wanted)
#undef YR_ET_PP
      return;
    }
  }
  const dim3 grid(a.row_blocks, a.slices);
  if (a.k <= 4) et_launch_one<DD, 4, BB, SS, false>(a, grid, 0, s);
  else if (a.k <= 10) et_launch_one<DD, 10, BB, SS, false>(a, grid, 0, s);
  else if (a.k <= 16) et_launch_one<DD, 16, BB, SS, false>(a, grid, 0, s);
  else if constexpr (DD <= 64) et_launch_one<DD, 32, BB, SS, false>(a, grid, 0, s);   // (D = 128: refused by the caller)
}

template <int DD>
void et_launch_d(const EtArgs& a, bool split, hipStream_t s) {
  if (a.item_bias) {
    if (split) et_launch<DD, true, true>(a, s);
    else et_launch<DD, true, false>(a, s);
  } else {
    if (split) et_launch<DD, false, true>(a, s);
    else et_launch<DD, false, false>(a, s);
  }
}
}  // namespace

extern "C" int yr_mf_eval_topk_bias(const float* U, const float* I, const float* item_bias, const int64_t* users,
                                    int64_t nrows, int D, int64_t num_users, int64_t num_items,
                                    const int64_t* mask_ptr, const int64_t* mask_idx, float mask_value, int k,
                                    int64_t* out, void* workspace, int64_t workspace_bytes, int mode,
                                    const int64_t* hint, int32_t* err_flag, void* stream) {
  if (nrows < 0 || num_users <= 0 || num_items <= 0 || num_items > 0x7ffffff0 || k <= 0 || k > kEtMaxK ||
      !et_mode_ok(mode) || workspace_bytes < 0)
    return YR_ERR_BADARG;
  if (!et_dim_ok(D) || (k > 16 && D > 64)) return YR_ERR_UNSUPPORTED;        // 32-entry lists: registers up to D = 64
  if (nrows == 0) return 0;
  if (!U || !I || !users || !out || (mask_ptr && !mask_idx)) return YR_ERR_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const bool split = (mode & YR_EVAL_BF16X3) != 0;
  char* ws = static_cast<char*>(workspace);
  if (!ws) workspace_bytes = 0;
  EtArgs a{};
  a.U = U;
  a.items = I;
  if (split) {
    const int64_t pb = et_plane_bytes(num_items, D);
    if (workspace_bytes < pb) return YR_ERR_BADARG;                            // the planes are not optional
    const int64_t groups = num_items * (D / 8);
    hipLaunchKernelGGL(et_split_rows_kernel, dim3((unsigned)((groups + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, I,
                       groups, D / 8, reinterpret_cast<uint4*>(ws));
    a.items = ws;
    ws += pb;
    workspace_bytes -= pb;
  }
  int S = et_slices(nrows, num_items);
  if (hint && workspace_bytes >= et_tau_bytes(nrows)) {                       // no room: the hint is ignored
    float* tau = reinterpret_cast<float*>(ws);
    const dim3 hgrid((unsigned)((nrows + kBlock / 16 - 1) / (kBlock / 16)));
#define YR_ET_HINT(DD)                                                                                              \
  hipLaunchKernelGGL((et_hint_bound_kernel<DD>), hgrid, dim3(kBlock), 0, s, U, I, item_bias, users, nrows, num_users, \
                     num_items, mask_ptr, mask_idx, mask_value, hint, k, tau)
    switch (D) {
      case 16: YR_ET_HINT(16); break;
      case 32: YR_ET_HINT(32); break;
      case 64: YR_ET_HINT(64); break;
      default: YR_ET_HINT(128); break;
    }
#undef YR_ET_HINT
    a.row_tau = tau;
    ws += et_tau_bytes(nrows);
    workspace_bytes -= et_tau_bytes(nrows);
  }
  if (!a.row_tau && et_prescan_wanted(num_items, k, mode) && workspace_bytes >= et_gmax_bytes(nrows, S)) {   // no room: no prescan
    a.gmax = reinterpret_cast<float*>(ws);
    a.parts = S;
    ws += et_gmax_bytes(nrows, S);
    workspace_bytes -= et_gmax_bytes(nrows, S);
  }
  const bool two_roles = et_pp_wanted(nrows, D, k, mode, a.row_tau != nullptr);
  if (two_roles) S = std::min(S, et_pp_slices(nrows, num_items));                          // (the prescan keeps its parts)
  if (S > 1 && workspace_bytes < nrows * S * k * (int64_t)sizeof(TopEntry)) S = 1;          // no room: one slice
  int per = (int)((num_items + S - 1) / S);
  per = (per + 31) / 32 * 32;                          // whole 32-item tiles
  S = (int)((num_items + per - 1) / per);
  a.item_bias = item_bias;
  a.users = users;
  a.nrows = nrows;
  a.num_users = num_users;
  a.num_items = (int)num_items;
  a.mask_ptr = mask_ptr;
  a.mask_idx = mask_idx;
  a.mask_value = mask_value;
  a.k = k;
  a.out = out;
  a.partial = S > 1 ? reinterpret_cast<TopEntry*>(ws) : nullptr;
  a.per = per;
  a.err_flag = err_flag;
  a.row_blocks = (unsigned)((nrows + kEtUsers - 1) / kEtUsers);
  a.slices = (unsigned)S;
  a.two_roles = two_roles;
  switch (D) {
    case 16: et_launch_d<16>(a, split, s); break;
    case 32: et_launch_d<32>(a, split, s); break;
    case 64: et_launch_d<64>(a, split, s); break;
    default: et_launch_d<128>(a, split, s); break;
  }
  if (S > 1)
    hipLaunchKernelGGL(mf_eval_merge_kernel, dim3((unsigned)((nrows + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       a.partial, nrows, S, k, out);
  return launch_status();
}

extern "C" int yr_mf_eval_topk(const float* U, const float* I, const int64_t* users, int64_t nrows, int D,
                               int64_t num_users, int64_t num_items, const int64_t* mask_ptr,
                               const int64_t* mask_idx, float mask_value, int k, int64_t* out, void* workspace,
                               int64_t workspace_bytes, int mode, const int64_t* hint, int32_t* err_flag,
                               void* stream) {
  return yr_mf_eval_topk_bias(U, I, nullptr, users, nrows, D, num_users, num_items, mask_ptr, mask_idx, mask_value, k,
                              out, workspace, workspace_bytes, mode, hint, err_flag, stream);
}

#ifdef YR_PP_TRACE
extern "C" int yr_debug_pp_trace(long long* host_out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_trace), sizeof(long long) * 2 * 64 * 8);
}
#endif

#ifdef YR_ET_STAMPS
extern "C" int yr_debug_eval_phases(unsigned long long* host_out, int reset) {
  (void)hipDeviceSynchronize();
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_et_phase), sizeof(unsigned long long) * 8);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_et_phase), z, sizeof(z));
  }
  return (int)e;
}
#endif
