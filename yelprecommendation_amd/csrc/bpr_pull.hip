// Pull-based BPR-MF training step for gfx950 (MI355X): no float atomics, no gradient buffers —
// the whole reference step
//   2 x forward, BPRLoss, loss.backward() (embedding_dense_backward), Adam.step()
//   (reference trainers/mf_trainer.py:106-112, trainers/base_trainer.py:34-36)
// as: two-level counting sort of the batch by user and by item -> one fused pass over the USER
// rows -> one fused pass over the ITEM rows.
//
// Why: the push form (csrc/bpr_mf.hip) is capped by the memory-side float-atomic rate
// (~1.3 TB/s of added bytes, 3 rows per triplet).  Here every destination row is owned by one
// wave (or one workgroup when it has very many contributions): the wave PULLS the partner rows
// it needs with plain 16-byte loads, sums them in registers, applies Adam to its own row and
// writes it once.  Dense-Adam semantics are kept: every row of both tables is visited every
// step, rows without contributions get grad = 0 (their m/v still decay).
//
// Index build.  Device-scope integer atomics cost one fabric request each (~20 G/s measured on
// MI355X: a flat counting sort with 3 global atomics per triplet took 0.5 ms per 1 M triplets),
// and LDS atomics retire about one lane per clock per CU, so both are rationed:
//   level 1  rows are cut into buckets of 64 consecutive rows.  A workgroup histograms its tile
//            of 8192 triplets per bucket in LDS (integer LDS atomics), reserves a range in each
//            bucket with ONE global atomic per (workgroup, bucket), and scatters
//              rec1[slot] = {pos, neg | local_user_row << 24, triplet id b}          by user bucket
//              occ1[slot] = {user | local_item_row << 24, b | sign}  (one per pos/neg occurrence)
//                                                                                    by item bucket
//   level 2  one workgroup per bucket counting-sorts the bucket's records by local row in LDS and
//            writes rec2 / occ2 in row order plus the row offsets
//            offU / offI; rows with more than `heavy` contributions go on a heavy list (a whole
//            workgroup sums such a row, one wave every other row).
//   user pass    row u: x_b = U[u].(I[p_b]-I[n_b]); loss += softplus(-x_b);
//                g_b = -sigmoid(-x_b)/B; acc += g_b (I[p_b]-I[n_b]);
//                coeff[b] = g_b (4 B per triplet, stays in L2);  U_new[u] = Adam(U[u], acc)
//   permute      g_item[j] = (+/-) coeff[b_j] for every occurrence j in item order (plain gather)
//   item pass    row i: acc = sum_j g_item[j] * U[user_j];  I[i] = Adam(I[i], acc)
// The user table is double-buffered (U -> U_new) because the item pass needs the OLD user rows;
// the item table is updated in place (a row is read only by its own wave).
//
// Layout: a row of D floats sits on LPR = D/4 lanes as float4 (16-byte loads/stores); a wave
// works on 64/LPR contributions at once; the per-contribution dot product is a reduction over
// LPR lanes only.
#include "common.h"

namespace yr {

constexpr int kBucketRows = 64;                 // rows per bucket (power of two)
constexpr int kBucketShift = 6;
constexpr int kPartThreads = 1024;              // partition workgroup
#ifndef YR_PART_PER_THREAD
#define YR_PART_PER_THREAD 8
#endif
constexpr int kPartPerThread = YR_PART_PER_THREAD;
constexpr int kPartTile = kPartThreads * kPartPerThread;   // triplets per partition workgroup
constexpr int kLocalShift = 24;                 // ids < 2^24 share a word with the local row
constexpr int kIdMask = (1 << kLocalShift) - 1;

struct AdamC {
  float decay_mul, neg_step, bc2_sqrt, one_m_b1, beta2, one_m_b2, eps, wd;
  int decoupled;
};

__device__ __forceinline__ void adam1(float& p, float grad, float& m, float& v, const AdamC& c) {
  if (c.wd != 0.0f) {
    if (c.decoupled) p *= c.decay_mul;
    else grad = grad + c.wd * p;
  }
  m = m + c.one_m_b1 * (grad - m);
  v = v * c.beta2 + (c.one_m_b2 * grad) * grad;
  const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
  p = p + (c.neg_step * m) / denom;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// --------------------------------------------------------------------------- partition
// Each workgroup owns one tile of kPartTile triplets, kept in registers as int32.
struct Tile {
  int32_t u[kPartPerThread], p[kPartPerThread], n[kPartPerThread];   // u < 0: skip
};

__device__ __forceinline__ int load_tile(Tile& t, const int64_t* __restrict__ user,
                                         const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                         int64_t B, int64_t nU, int64_t nI) {
  int bad = 0;
  const int64_t base = (int64_t)blockIdx.x * kPartTile + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPartPerThread; ++k) {
    const int64_t b = base + (int64_t)k * kPartThreads;
    t.u[k] = -1; t.p[k] = 0; t.n[k] = 0;
    if (b < B) {
      const int64_t u = user[b], p = pos[b], n = neg[b];
      int f = 0;
      if ((uint64_t)u >= (uint64_t)nU) f |= YR_FLAG_BAD_USER;
      if ((uint64_t)p >= (uint64_t)nI || (uint64_t)n >= (uint64_t)nI) f |= YR_FLAG_BAD_ITEM;
      if (f) bad |= f;
      else { t.u[k] = (int32_t)u; t.p[k] = (int32_t)p; t.n[k] = (int32_t)n; }
    }
  }
  return bad;
}

// LDS histogram of the tile per bucket: s_cnt[0..nbU) users, s_cnt[nbU..nbU+nbI) items.
// The partition kernels run TWO workgroups per tile (blockIdx.y = side): side 0 handles the user-side
// buckets and records, side 1 the item-side ones — a tile of 8192 triplets per workgroup alone gives
// only B / 8192 workgroups (128 at B = 2^20: half the CUs idle).
__device__ __forceinline__ void tile_histogram(const Tile& t, int32_t* s_cnt, int nbU, int nb_all, int side) {
  const int lo = side ? nbU : 0, hi = side ? nb_all : nbU;
  for (int i = lo + threadIdx.x; i < hi; i += kPartThreads) s_cnt[i] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPartPerThread; ++k) {
    if (t.u[k] >= 0) {
      if (side == 0) {
        atomicAdd(&s_cnt[t.u[k] >> kBucketShift], 1);
      } else {
        atomicAdd(&s_cnt[nbU + (t.p[k] >> kBucketShift)], 1);
        atomicAdd(&s_cnt[nbU + (t.n[k] >> kBucketShift)], 1);
      }
    }
  }
  __syncthreads();
}

// pass 1: bucket totals.  cnt_all = [user buckets | item buckets], zeroed by the caller.
__global__ __launch_bounds__(kPartThreads) void part_count_kernel(const int64_t* __restrict__ user,
                                                                  const int64_t* __restrict__ pos,
                                                                  const int64_t* __restrict__ neg, int64_t B,
                                                                  int64_t nU, int64_t nI, int nbU, int nb_all,
                                                                  int32_t* __restrict__ cnt_all,
                                                                  int32_t* __restrict__ cnt_tile,
                                                                  int32_t* __restrict__ err_flag) {
  extern __shared__ int32_t s_cnt[];
  Tile t;
  const int side = blockIdx.y;
  const int bad = load_tile(t, user, pos, neg, B, nU, nI);
  tile_histogram(t, s_cnt, nbU, nb_all, side);
  int32_t* mine = cnt_tile + (int64_t)blockIdx.x * nb_all;     // this tile's counts, reused by the scatter pass
  const int lo = side ? nbU : 0, hi = side ? nb_all : nbU;
  for (int i = lo + threadIdx.x; i < hi; i += kPartThreads) {
    const int c = s_cnt[i];
    mine[i] = c;
    if (c) atomicAdd(&cnt_all[i], c);
  }
  if (side == 0 && bad && err_flag) atomicOr(err_flag, bad);
}

// exclusive scans of the user-bucket and item-bucket totals (block 0 / block 1):
// base[i] = start of bucket i in its record array, base[nb] = total; cur[i] = base[i].
__global__ __launch_bounds__(kPartThreads) void part_scan_kernel(const int32_t* __restrict__ cnt_all,
                                                                 int nbU, int nbI, int32_t* __restrict__ baseU,
                                                                 int32_t* __restrict__ baseI,
                                                                 int32_t* __restrict__ cur_all) {
  __shared__ int s_wave[kPartThreads / kWave];
  __shared__ int s_carry;
  const bool items = blockIdx.x == 1;
  const int32_t* cnt = items ? cnt_all + nbU : cnt_all;
  int32_t* base = items ? baseI : baseU;
  int32_t* cur = items ? cur_all + nbU : cur_all;
  const int n = items ? nbI : nbU;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int start = 0; start < n; start += kPartThreads) {
    const int i = start + threadIdx.x;
    const int c = i < n ? cnt[i] : 0;
    int inc = c;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int o = __shfl_up(inc, d, kWave);
      if (lane >= d) inc += o;
    }
    if (lane == kWave - 1) s_wave[wave] = inc;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (i < n) {
      base[i] = before + inc - c;
      cur[i] = before + inc - c;
    }
    __syncthreads();
    if (threadIdx.x == kPartThreads - 1) s_carry = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) base[n] = s_carry;
}

// pass 2: reserve a range per (workgroup, bucket), then scatter the records.
__global__ __launch_bounds__(kPartThreads) void part_scatter_kernel(const int64_t* __restrict__ user,
                                                                    const int64_t* __restrict__ pos,
                                                                    const int64_t* __restrict__ neg, int64_t B,
                                                                    int64_t nU, int64_t nI, int nbU, int nb_all,
                                                                    int32_t* __restrict__ cur_all,
                                                                    const int32_t* __restrict__ cnt_tile,
                                                                    int4* __restrict__ user_rec,
                                                                    int2* __restrict__ occ_rec) {
  extern __shared__ int32_t s_mem[];
  int32_t* s_cnt = s_mem;             // [nb_all] running ranks
  int32_t* s_start = s_mem + nb_all;  // [nb_all] reserved start per bucket
  Tile t;
  load_tile(t, user, pos, neg, B, nU, nI);
  const int32_t* mine = cnt_tile + (int64_t)blockIdx.x * nb_all;   // counted by part_count_kernel
  const int side = blockIdx.y;                                     // 0: user-side records, 1: item-side
  const int lo = side ? nbU : 0, hi = side ? nb_all : nbU;
  for (int i = lo + threadIdx.x; i < hi; i += kPartThreads) {
    const int c = mine[i];
    s_start[i] = c ? atomicAdd(&cur_all[i], c) : 0;
    s_cnt[i] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPartPerThread; ++k) {
    if (t.u[k] >= 0) {
      const int b = (int)(blockIdx.x * kPartTile + threadIdx.x + k * kPartThreads);   // triplet id
      if (side == 0) {
        const int bu = t.u[k] >> kBucketShift;
        const int su = s_start[bu] + atomicAdd(&s_cnt[bu], 1);
        user_rec[su] = make_int4(t.p[k], t.n[k] | ((t.u[k] & (kBucketRows - 1)) << kLocalShift), b, 0);
      } else {
        const int bp = nbU + (t.p[k] >> kBucketShift), bn = nbU + (t.n[k] >> kBucketShift);
        const int sp = s_start[bp] + atomicAdd(&s_cnt[bp], 1);
        const int sn = s_start[bn] + atomicAdd(&s_cnt[bn], 1);
        occ_rec[sp] = make_int2(t.u[k] | ((t.p[k] & (kBucketRows - 1)) << kLocalShift), b);
        occ_rec[sn] = make_int2(t.u[k] | ((t.n[k] & (kBucketRows - 1)) << kLocalShift), b | (int)0x80000000);
      }
    }
  }
}

// --------------------------------------------------------------------------- level-2 sort
// One workgroup per bucket: counting sort of the bucket's records by local row.
#ifndef YR_SORT_THREADS
#define YR_SORT_THREADS 1024
#endif
constexpr int kSortThreads = YR_SORT_THREADS;

struct SortSide {
  const int32_t* base;     // [buckets + 1]
  int32_t* off;            // [rows + 1]
  int32_t* heavy;
  int32_t* nheavy;
  int buckets, rows;
};

// One workgroup per bucket: blocks [0, su.buckets) sort user buckets (rec1 -> rec2), the others item
// buckets (occ1 -> occ2), both by local row with LDS rank atomics.
__global__ __launch_bounds__(kSortThreads) void bucket_sort_kernel(SortSide su, SortSide si,
                                                                   const int4* __restrict__ rec1,
                                                                   const int2* __restrict__ occ1,
                                                                   int4* __restrict__ rec2, int2* __restrict__ occ2,
                                                                   int heavy_t) {
  __shared__ int s_cnt[kBucketRows];
  __shared__ int s_start[kBucketRows];
  const bool user = (int)blockIdx.x < su.buckets;
  const SortSide& sd = user ? su : si;
  const int bucket = user ? blockIdx.x : blockIdx.x - su.buckets;
  const int lo = sd.base[bucket], hi = sd.base[bucket + 1];
  const int row0 = bucket * kBucketRows;
  if (threadIdx.x < kBucketRows) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int i = lo + threadIdx.x; i < hi; i += kSortThreads) {
    const uint32_t w = user ? (uint32_t)rec1[i].y : (uint32_t)occ1[i].x;
    atomicAdd(&s_cnt[w >> kLocalShift], 1);
  }
  __syncthreads();
  if (threadIdx.x < kWave) {                     // wave 0: exclusive scan of the 64 row counts
    const int c = s_cnt[threadIdx.x];
    int inc = c;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int o = __shfl_up(inc, d, kWave);
      if ((int)threadIdx.x >= d) inc += o;
    }
    const int start = lo + inc - c;
    s_start[threadIdx.x] = start;
    const int row = row0 + threadIdx.x;
    if (row < sd.rows) {
      sd.off[row] = start;
      if (c > heavy_t) sd.heavy[atomicAdd(sd.nheavy, 1)] = row;
    }
    if (row == sd.rows - 1) sd.off[sd.rows] = start + c;
  }
  __syncthreads();
  for (int i = lo + threadIdx.x; i < hi; i += kSortThreads) {
    if (user) {
      int4 r = rec1[i];
      const int local = (uint32_t)r.y >> kLocalShift;
      r.y &= kIdMask;
      rec2[atomicAdd(&s_start[local], 1)] = r;
    } else {
      int2 o = occ1[i];
      const int local = (uint32_t)o.x >> kLocalShift;
      o.x &= kIdMask;
      occ2[atomicAdd(&s_start[local], 1)] = o;
    }
  }
}

// --------------------------------------------------------------------------- row passes
struct RowPassArgs {
  const float* own_old;      // table whose rows are updated (read)
  float* own_new;            // where the updated rows go (== own_old for the item pass)
  const float* other;        // table whose rows are gathered
  float* m;
  float* v;
  float* grad_out;           // item pass, FUSE_ADAM = false: dense gradient rows instead of Adam
  const int32_t* off;        // [rows + 1] contribution range of each row
  const int4* rec;           // user pass: {pos, neg, triplet id, -} in user order
  const int2* occ;           // item pass: {user, triplet id | sign bit (neg occurrence)} in item order
  float* coeff;              // user pass: [B] g_b by triplet id (written); item pass: [2B] signed g in item order
  const int32_t* heavy;      // heavy-row list and its length
  const int32_t* nheavy;
  float* loss_partials;      // user pass
  int rows;                  // rows [row_begin, rows) ... the pass covers [row_begin, row_end)
  int row_begin, row_end;
  int heavy_t;
  float inv_batch;
  AdamC adam;
};

template <int D>
struct PullGeom {
  static_assert(D == 16 || D == 32 || D == 64 || D == 128, "unsupported width");
  static constexpr int LPR = D / 4;            // lanes per row (float4 each)
  static constexpr int GPW = kWave / LPR;      // contributions per wave pass
};

// contributions in flight per lane group and pass
constexpr int kUserUnroll = 2;   // VALU-heavier, two rows per contribution
constexpr int kItemUnroll = 4;   // pure latency: rows of TWO passes are kept in flight

// User pass: accumulate the contributions [lo, hi) of one user row, visiting indices
// base + first + k*step.  Adds into this lane group's partial gradient (float4 at column 4*l) and
// loss.  Software pipelined: the index words of the NEXT pass are loaded before the rows of this
// one are used, so the dependent chain index -> row never leaves the wave without loads in flight.
template <int D>
__device__ __forceinline__ void pull_accumulate_user(const RowPassArgs& a, int lo, int hi, int first, int step,
                                                     const float4& own, int l, float4& acc, float& loss) {
  using G = PullGeom<D>;
  constexpr int N = kUserUnroll;
  int4 cur[N], nxt[N];
  bool cur_valid[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int idx = lo + first + q * step;
    cur_valid[q] = idx < hi;
    cur[q] = cur_valid[q] ? a.rec[idx] : make_int4(0, 0, 0, 0);
  }
  // uniform trip count across the wave / workgroup: the start is common, `first` only offsets idx
  for (int base = lo; base < hi; base += step * N) {
    float4 r0[N], r1[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      r0[q] = ld4(a.other + (int64_t)cur[q].x * D + 4 * l);
      r1[q] = ld4(a.other + (int64_t)cur[q].y * D + 4 * l);
    }
    bool nxt_valid[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int idx = base + step * N + first + q * step;
      nxt_valid[q] = idx < hi;
      nxt[q] = nxt_valid[q] ? a.rec[idx] : make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      float4 d;
      d.x = r0[q].x - r1[q].x; d.y = r0[q].y - r1[q].y; d.z = r0[q].z - r1[q].z; d.w = r0[q].w - r1[q].w;
      float part = own.x * d.x;
      part = fmaf(own.y, d.y, part);
      part = fmaf(own.z, d.z, part);
      part = fmaf(own.w, d.w, part);
      const float x = group_sum_dpp<G::LPR>(part);
      float sp, sg;
      bpr_terms(x, sp, sg);
      const float gg = cur_valid[q] ? -sg * a.inv_batch : 0.0f;
      acc.x = fmaf(gg, d.x, acc.x); acc.y = fmaf(gg, d.y, acc.y);
      acc.z = fmaf(gg, d.z, acc.z); acc.w = fmaf(gg, d.w, acc.w);
      if (cur_valid[q] && l == 0) {
        a.coeff[cur[q].z] = gg;                 // one hand-off word per triplet; the item side applies the sign
        loss += sp;
      }
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      cur[q] = nxt[q];
      cur_valid[q] = nxt_valid[q];
    }
  }
}

// Item pass: acc += sum over [lo, hi) of (+/-) g_b * U[user], {user, b | sign} read contiguously from
// `occ`, g_b from the per-triplet coefficient array.  Two passes of rows are in flight: the rows
// (and coefficients) of pass k+1 are requested before those of pass k are consumed, and the index
// words of pass k+2 before that.
template <int D>
__device__ __forceinline__ void pull_accumulate_item(const RowPassArgs& a, int lo, int hi, int first, int step,
                                                     int l, float4& acc) {
  constexpr int N = kItemUnroll;
  int cur[N], nxt[N];                             // user ids; padding = user 0 with g = 0
  float4 rcur[N];
  float gcur[N], gnx[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int idx = lo + first + q * step;
    cur[q] = idx < hi ? a.occ[idx].x : 0;
    gcur[q] = idx < hi ? a.coeff[idx] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int idx = lo + step * N + first + q * step;
    nxt[q] = idx < hi ? a.occ[idx].x : 0;
    gnx[q] = idx < hi ? a.coeff[idx] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < N; ++q) rcur[q] = ld4(a.other + (int64_t)cur[q] * D + 4 * l);
  for (int base = lo; base < hi; base += step * N) {
    float4 rnxt[N];
    int nn[N];
    float gn[N];
#pragma unroll
    for (int q = 0; q < N; ++q) rnxt[q] = ld4(a.other + (int64_t)nxt[q] * D + 4 * l);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int idx = base + 2 * step * N + first + q * step;
      nn[q] = idx < hi ? a.occ[idx].x : 0;
      gn[q] = idx < hi ? a.coeff[idx] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const float g = gcur[q];
      acc.x = fmaf(g, rcur[q].x, acc.x); acc.y = fmaf(g, rcur[q].y, acc.y);
      acc.z = fmaf(g, rcur[q].z, acc.z); acc.w = fmaf(g, rcur[q].w, acc.w);
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      cur[q] = nxt[q];
      rcur[q] = rnxt[q];
      gcur[q] = gnx[q];
      nxt[q] = nn[q];
      gnx[q] = gn[q];
    }
  }
}

template <int D, bool USER>
__device__ __forceinline__ void pull_accumulate(const RowPassArgs& a, int lo, int hi, int first, int step,
                                                const float4& own, int l, float4& acc, float& loss) {
  if (USER) pull_accumulate_user<D>(a, lo, hi, first, step, own, l, acc, loss);
  else pull_accumulate_item<D>(a, lo, hi, first, step, l, acc);
}

// sum a float4 over the lane groups of a wave (lanes with equal l)
template <int LPR>
__device__ __forceinline__ void cross_group_sum(float4& a) {
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    a.x += __shfl_xor(a.x, m, kWave);
    a.y += __shfl_xor(a.y, m, kWave);
    a.z += __shfl_xor(a.z, m, kWave);
    a.w += __shfl_xor(a.w, m, kWave);
  }
}

template <int D, bool FUSE_ADAM>
__device__ __forceinline__ void finish_row(const RowPassArgs& a, int row, float4 own, const float4& grad, int l) {
  const int64_t o = (int64_t)row * D + 4 * l;
  if (FUSE_ADAM) {
    float4 M = ld4(a.m + o), V = ld4(a.v + o);
    adam1(own.x, grad.x, M.x, V.x, a.adam);
    adam1(own.y, grad.y, M.y, V.y, a.adam);
    adam1(own.z, grad.z, M.z, V.z, a.adam);
    adam1(own.w, grad.w, M.w, V.w, a.adam);
    st4(a.own_new + o, own);
    st4(a.m + o, M);
    st4(a.v + o, V);
  } else {
    st4(a.grad_out + o, grad);
  }
}

// One launch per table.  Workgroups [0, kHeavyBlocks) walk the heavy-row list, a whole workgroup
// (4 waves) per row with an LDS reduction; the others give one wave to each remaining row.  The
// heavy workgroups have the lowest ids, so they start first and their long rows overlap the rest.
constexpr int kHeavyBlocks = 512;
template <int D, bool USER, bool FUSE_ADAM>
__global__ __launch_bounds__(kBlock) void pull_rows_kernel(RowPassArgs a) {
  using G = PullGeom<D>;
  __shared__ float4 s_acc[kWavesPerBlock][G::LPR];
  __shared__ float s_red[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int grp = lane / G::LPR, l = lane % G::LPR;
  float loss = 0.0f;
  if ((int)blockIdx.x < kHeavyBlocks) {
    const int nh = a.nheavy[0];
    for (int h = blockIdx.x; h < nh; h += kHeavyBlocks) {
      const int row = a.heavy[h];
      if (row < a.row_begin || row >= a.row_end) continue;      // workgroup-uniform
      const int lo = a.off[row], hi = a.off[row + 1];
      const float4 own = ld4(a.own_old + (int64_t)row * D + 4 * l);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      pull_accumulate<D, USER>(a, lo, hi, wave * G::GPW + grp, kWavesPerBlock * G::GPW, own, l, acc, loss);
      cross_group_sum<G::LPR>(acc);
      if (grp == 0) s_acc[wave][l] = acc;
      __syncthreads();
      if (wave == 0 && grp == 0) {
        float4 t = s_acc[0][l];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) {
          const float4 o = s_acc[w][l];
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        finish_row<D, FUSE_ADAM>(a, row, own, t, l);
      }
      __syncthreads();
    }
  } else {
    const int nwaves = (gridDim.x - kHeavyBlocks) * kWavesPerBlock;
    for (int row = a.row_begin + (blockIdx.x - kHeavyBlocks) * kWavesPerBlock + wave; row < a.row_end;
         row += nwaves) {
      const int lo = a.off[row], hi = a.off[row + 1];
      if (hi - lo > a.heavy_t) continue;
      const float4 own = ld4(a.own_old + (int64_t)row * D + 4 * l);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      pull_accumulate<D, USER>(a, lo, hi, grp, G::GPW, own, l, acc, loss);
      cross_group_sum<G::LPR>(acc);
      if (grp == 0) finish_row<D, FUSE_ADAM>(a, row, own, acc, l);
    }
  }
  if (USER) {
    const float total = block_sum(loss, s_red);
    if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = total;
  }
}

// g_item[j] = (+/-) coeff[b_j] for every occurrence j in item order: a plain massively parallel
// gather (coalesced index reads and stores) so that the item pass streams {user, g} contiguously
// instead of chasing the per-triplet coefficient from inside its latency-bound row loop.
__global__ __launch_bounds__(kBlock) void pull_permute_coeff_kernel(const int2* __restrict__ occ,
                                                                    const float* __restrict__ coeff,
                                                                    const int32_t* __restrict__ n_occ,
                                                                    float* __restrict__ g_item) {
  // only the occurrences of VALID triplets exist (out-of-range triplets were skipped by the
  // partition): their number is the end of the last item bucket, not 2 * B
  const int64_t n = n_occ[0];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
    const int b = occ[j].y;
    const float g = coeff[b & 0x7fffffff];
    g_item[j] = b < 0 ? -g : g;
  }
}

__global__ void pull_clear_partials_kernel(float* p, int from, int to) {
  const int i = from + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < to) p[i] = 0.0f;
}

// workspace carve-up (all 16-byte aligned)
struct PullWorkspace {
  int32_t *cnt_all, *nheavy, *cur_all, *baseU, *baseI, *offU, *offI, *heavyU, *heavyI, *cnt_tile;
  int2* occ1;
  float *coeff, *g_item;
  int4 *rec1, *rec2;
  int2* occ2;
  int nbU, nbI;
  size_t bytes;
};

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

inline PullWorkspace carve(void* base, int64_t B, int64_t nU, int64_t nI) {
  PullWorkspace w;
  w.nbU = (int)((nU + kBucketRows - 1) / kBucketRows);
  w.nbI = (int)((nI + kBucketRows - 1) / kBucketRows);
  const size_t nb_all = (size_t)w.nbU + (size_t)w.nbI;
  char* p = static_cast<char*>(base);
  size_t o = 0;
  // cnt_all and nheavy are cleared by one memset per step
  w.cnt_all = (int32_t*)(p + o); o += align16(nb_all * 4);
  w.nheavy = (int32_t*)(p + o); o += 16;
  w.cur_all = (int32_t*)(p + o); o += align16(nb_all * 4);
  w.baseU = (int32_t*)(p + o); o += align16((size_t)(w.nbU + 1) * 4);
  w.baseI = (int32_t*)(p + o); o += align16((size_t)(w.nbI + 1) * 4);
  w.offU = (int32_t*)(p + o); o += align16((size_t)(nU + 1) * 4);
  w.offI = (int32_t*)(p + o); o += align16((size_t)(nI + 1) * 4);
  w.heavyU = (int32_t*)(p + o); o += align16((size_t)nU * 4);
  w.heavyI = (int32_t*)(p + o); o += align16((size_t)nI * 4);
  w.rec1 = (int4*)(p + o); o += align16((size_t)B * sizeof(int4));
  w.rec2 = (int4*)(p + o); o += align16((size_t)B * sizeof(int4));
  w.occ1 = (int2*)(p + o); o += align16((size_t)B * 2 * sizeof(int2));
  w.occ2 = (int2*)(p + o); o += align16((size_t)B * 2 * sizeof(int2));
  w.coeff = (float*)(p + o); o += align16((size_t)B * 4);
  w.g_item = (float*)(p + o); o += align16((size_t)B * 2 * 4);
  const size_t ptiles = (size_t)((B + kPartTile - 1) / kPartTile);
  w.cnt_tile = (int32_t*)(p + o); o += align16(ptiles * nb_all * 4);
  w.bytes = o;
  return w;
}

}  // namespace yr

using namespace yr;

extern "C" int64_t yr_bpr_mf_pull_workspace_bytes(int64_t max_batch, int64_t num_users, int64_t num_items) {
  if (max_batch < 0 || num_users <= 0 || num_items <= 0) return YR_ERR_BADARG;
  return (int64_t)carve(nullptr, max_batch, num_users, num_items).bytes;
}

// phase 1: index build (independent of the tables: may run ahead, e.g. under the previous step's
// all-reduce) -> workspace
static int pull_index_impl(const int64_t* user, const int64_t* pos, const int64_t* neg, int64_t B, int64_t nU,
                           int64_t nI, int heavy_t, void* workspace, int32_t* err_flag, hipStream_t s) {
  PullWorkspace w = carve(workspace, B, nU, nI);
  const int nb_all = w.nbU + w.nbI;
  // level-1 partition of the batch into user buckets and item buckets
  hipError_t e = hipMemsetAsync(w.cnt_all, 0, (size_t)((char*)w.nheavy - (char*)w.cnt_all) + 16, s);
  if (e != hipSuccess) return (int)e;
  const int ptiles = (int)((B + kPartTile - 1) / kPartTile);
  if (ptiles > 0)
    hipLaunchKernelGGL(part_count_kernel, dim3(ptiles, 2), dim3(kPartThreads), (size_t)nb_all * 4, s, user, pos, neg, B,
                       nU, nI, w.nbU, nb_all, w.cnt_all, w.cnt_tile, err_flag);
  hipLaunchKernelGGL(part_scan_kernel, dim3(2), dim3(kPartThreads), 0, s, w.cnt_all, w.nbU, w.nbI, w.baseU, w.baseI,
                     w.cur_all);
  if (ptiles > 0)
    hipLaunchKernelGGL(part_scatter_kernel, dim3(ptiles, 2), dim3(kPartThreads), (size_t)nb_all * 8, s, user, pos, neg,
                       B, nU, nI, w.nbU, nb_all, w.cur_all, w.cnt_tile, w.rec1, w.occ1);
  // level-2 sort inside every bucket -> row offsets, records in row order, heavy lists
  SortSide su, si;
  su.base = w.baseU; su.off = w.offU; su.heavy = w.heavyU; su.nheavy = w.nheavy; su.buckets = w.nbU; su.rows = (int)nU;
  si.base = w.baseI; si.off = w.offI; si.heavy = w.heavyI; si.nheavy = w.nheavy + 1; si.buckets = w.nbI; si.rows = (int)nI;
  hipLaunchKernelGGL(bucket_sort_kernel, dim3(nb_all), dim3(kSortThreads), 0, s, su, si, w.rec1, w.occ1, w.rec2,
                     w.occ2, heavy_t);
  return launch_status();
}

// phase 2: the two fused row passes over an index built by phase 1 for the same batch
template <int D>
static int pull_apply_impl(const float* U_old, float* U_new, float* I, float* mU, float* vU, float* mI, float* vI,
                           float* gradI_out, int64_t B, int64_t nU, int64_t nI, float inv_batch, const AdamC& adam,
                           int heavy_t, void* workspace, float* loss_partials, int phases, int64_t item_begin,
                           int64_t item_end, hipStream_t s) {
  PullWorkspace w = carve(workspace, B, nU, nI);
  if (phases & YR_PULL_USER_PHASE) {
  // user pass (reads U_old + I, writes U_new, the per-triplet coefficients, loss partials)
  RowPassArgs ua;
  ua.own_old = U_old; ua.own_new = U_new; ua.other = I; ua.m = mU; ua.v = vU; ua.grad_out = nullptr;
  ua.off = w.offU; ua.rec = w.rec2; ua.occ = nullptr; ua.coeff = w.coeff;
  ua.heavy = w.heavyU; ua.nheavy = w.nheavy; ua.loss_partials = loss_partials;
  ua.rows = (int)nU; ua.row_begin = 0; ua.row_end = (int)nU;
  ua.heavy_t = heavy_t; ua.inv_batch = inv_batch; ua.adam = adam;
  const int light_cap = YR_LOSS_PARTIALS - kHeavyBlocks;        // one loss-partial slot per workgroup
  int gu = (int)((nU + kWavesPerBlock - 1) / kWavesPerBlock);
  if (gu > light_cap) gu = light_cap;
  gu += kHeavyBlocks;
  hipLaunchKernelGGL((pull_rows_kernel<D, true, true>), dim3(gu), dim3(kBlock), 0, s, ua);
  if (gu < YR_LOSS_PARTIALS)
    hipLaunchKernelGGL(pull_clear_partials_kernel, dim3((YR_LOSS_PARTIALS - gu + kBlock - 1) / kBlock), dim3(kBlock),
                       0, s, loss_partials, gu, YR_LOSS_PARTIALS);
  // coefficients into item order, then the item pass (reads U_old + occ2 + g_item, updates I in
  // place or writes gradI_out)
  if (B > 0)
    hipLaunchKernelGGL(pull_permute_coeff_kernel, dim3(grid_for(2 * B, kBlock)), dim3(kBlock), 0, s, w.occ2, w.coeff,
                       w.baseI + w.nbI, w.g_item);
  }
  if (!(phases & YR_PULL_ITEM_PHASE) || item_end <= item_begin) return launch_status();
  RowPassArgs ia;
  ia.own_old = I; ia.own_new = I; ia.other = U_old; ia.m = mI; ia.v = vI; ia.grad_out = gradI_out;
  ia.off = w.offI; ia.rec = nullptr; ia.occ = w.occ2; ia.coeff = w.g_item;
  ia.heavy = w.heavyI; ia.nheavy = w.nheavy + 1; ia.loss_partials = nullptr;
  ia.rows = (int)nI; ia.row_begin = (int)item_begin; ia.row_end = (int)item_end;
  ia.heavy_t = heavy_t; ia.inv_batch = inv_batch; ia.adam = adam;
  int gi = (int)((item_end - item_begin + kWavesPerBlock - 1) / kWavesPerBlock);
  if (gi > kMaxGrid) gi = kMaxGrid;
  gi += kHeavyBlocks;
  if (gradI_out)
    hipLaunchKernelGGL((pull_rows_kernel<D, false, false>), dim3(gi), dim3(kBlock), 0, s, ia);
  else
    hipLaunchKernelGGL((pull_rows_kernel<D, false, true>), dim3(gi), dim3(kBlock), 0, s, ia);
  return launch_status();
}

static int pull_check_common(int64_t B, int64_t num_users, int64_t num_items, const void* workspace,
                             int64_t workspace_bytes) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || B > 0x3fffffff) return YR_ERR_BADARG;
  // ids share a 32-bit word with the 8-bit local row number inside a bucket
  if (num_users > kIdMask || num_items > kIdMask) return YR_ERR_UNSUPPORTED;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return YR_ERR_BADARG;
  if ((int64_t)carve(nullptr, B, num_users, num_items).bytes > workspace_bytes) return YR_ERR_BADARG;
  return 0;
}

extern "C" int yr_bpr_mf_pull_index(const int64_t* user, const int64_t* pos, const int64_t* neg, int64_t B,
                                    int64_t num_users, int64_t num_items, int heavy_threshold, void* workspace,
                                    int64_t workspace_bytes, int32_t* err_flag, void* stream) {
  const int rc = pull_check_common(B, num_users, num_items, workspace, workspace_bytes);
  if (rc) return rc;
  if (B > 0 && (!user || !pos || !neg)) return YR_ERR_BADARG;
  if (heavy_threshold <= 0) heavy_threshold = 256;
  return pull_index_impl(user, pos, neg, B, num_users, num_items, heavy_threshold, workspace, err_flag,
                         (hipStream_t)stream);
}

static int make_adam(AdamC& c, double lr, double step_size, double bc2_sqrt, double beta1, double beta2, double eps,
                     double weight_decay, int mode) {
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  c.decay_mul = (float)(1.0 - lr * weight_decay);
  c.neg_step = (float)(-step_size);
  c.bc2_sqrt = (float)bc2_sqrt;
  c.one_m_b1 = (float)(1.0 - beta1);
  c.beta2 = (float)beta2;
  c.one_m_b2 = (float)(1.0 - beta2);
  c.eps = (float)eps;
  c.wd = (float)weight_decay;
  c.decoupled = mode == YR_OPT_ADAMW;
  return 0;
}

extern "C" int yr_bpr_mf_pull_apply(const float* U_old, float* U_new, float* I, float* mU, float* vU, float* mI,
                                    float* vI, float* gradI_out, int64_t B, int D, int64_t num_users,
                                    int64_t num_items, float inv_batch, double lr, double step_size, double bc2_sqrt,
                                    double beta1, double beta2, double eps, double weight_decay, int mode,
                                    int heavy_threshold, void* workspace, int64_t workspace_bytes,
                                    float* loss_partials, int phases, int64_t item_row_begin,
                                    int64_t item_row_end, void* stream) {
  int rc = pull_check_common(B, num_users, num_items, workspace, workspace_bytes);
  if (rc) return rc;
  if (!(phases & (YR_PULL_USER_PHASE | YR_PULL_ITEM_PHASE))) return YR_ERR_BADARG;
  if (item_row_begin < 0 || item_row_end > num_items || item_row_begin > item_row_end) return YR_ERR_BADARG;
  if (!U_old || !U_new || U_old == U_new || !I || !mU || !vU || !loss_partials) return YR_ERR_BADARG;
  if (!gradI_out && (!mI || !vI)) return YR_ERR_BADARG;
  if (heavy_threshold <= 0) heavy_threshold = 256;
  AdamC c;
  rc = make_adam(c, lr, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, mode);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
#define YR_APPLY_CASE(DD)                                                                                       \
  case DD:                                                                                                      \
    return pull_apply_impl<DD>(U_old, U_new, I, mU, vU, mI, vI, gradI_out, B, num_users, num_items, inv_batch,  \
                               c, heavy_threshold, workspace, loss_partials, phases, item_row_begin,            \
                               item_row_end, s)
  switch (D) {
    YR_APPLY_CASE(16);
    YR_APPLY_CASE(32);
    YR_APPLY_CASE(64);
    YR_APPLY_CASE(128);
    default: return YR_ERR_UNSUPPORTED;
  }
#undef YR_APPLY_CASE
}

extern "C" int yr_bpr_mf_pull_step(const float* U_old, float* U_new, float* I, float* mU, float* vU, float* mI,
                                   float* vI, float* gradI_out, const int64_t* user, const int64_t* pos,
                                   const int64_t* neg, int64_t B, int D, int64_t num_users, int64_t num_items,
                                   float inv_batch, double lr, double step_size, double bc2_sqrt, double beta1,
                                   double beta2, double eps, double weight_decay, int mode, int heavy_threshold,
                                   void* workspace, int64_t workspace_bytes, float* loss_partials,
                                   int32_t* err_flag, void* stream) {
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  if (!U_old || !U_new || U_old == U_new || !I || !mU || !vU || !loss_partials) return YR_ERR_BADARG;
  if (!gradI_out && (!mI || !vI)) return YR_ERR_BADARG;
  int rc = yr_bpr_mf_pull_index(user, pos, neg, B, num_users, num_items, heavy_threshold, workspace, workspace_bytes,
                                err_flag, stream);
  if (rc) return rc;
  return yr_bpr_mf_pull_apply(U_old, U_new, I, mU, vU, mI, vI, gradI_out, B, D, num_users, num_items, inv_batch, lr,
                              step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, mode, heavy_threshold, workspace,
                              workspace_bytes, loss_partials, YR_PULL_USER_PHASE | YR_PULL_ITEM_PHASE, 0, num_items,
                              stream);
}
