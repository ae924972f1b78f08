// Pull-based BPR-MF training step for gfx950 (MI355X): no float atomics, no global integer
// atomics, no gradient buffers — the whole reference step
//   2 x forward, BPRLoss, loss.backward() (embedding_dense_backward), Adam.step()
//   (reference trainers/mf_trainer.py:106-112, trainers/base_trainer.py:34-36)
// in THREE launches: tile partition -> owner pass over the USER rows -> owner pass over the ITEM rows.
//
// Why pull: the push form (csrc/bpr_mf.hip) is capped by the memory-side float-atomic rate
// (~1.3 TB/s of added bytes, 3 rows per triplet).  Here every destination row is owned by one
// lane group of one workgroup, which PULLS the partner rows it needs with plain 16-byte loads, sums
// them in registers, applies Adam to its own row and writes it once.  Dense-Adam semantics are
// kept: every row of both tables is visited every step, rows without contributions get grad = 0
// (their m/v still decay).
//
// Rows are cut into buckets of R = 1024/D consecutive rows (4 KB of table: one 256-thread owner
// workgroup, one row per lane group of D/4 lanes).
//   1  tile_partition_kernel   a workgroup takes one TILE of the batch (1024..8192 triplets) and one
//        side (user / item), counting-sorts the tile by destination bucket in LDS (integer LDS
//        atomics give the rank inside (tile, bucket)) and writes the records into the tile's OWN
//        region of the record array, bucket by bucket, plus the tile's bucket offsets
//          rec[tile]  = {pos, neg, triplet id b, local user row}            (16 B, by user bucket)
//          occ[tile]  = {user | local item row << 26, b | sign}              ( 8 B, by item bucket,
//                                                       one per pos / neg occurrence)
//        A tile's region is written by one workgroup only: no global atomics, no count/scan passes,
//        no partial lines shared between workgroups.
//   2  owner_pass_kernel<USER>  the owner of bucket k reads segment k of every tile (offsets ->
//        LDS prefix, flattened index -> (tile, position) by binary search), ranks the records by
//        local row with LDS integer atomics and so gets ONE stream sorted by row in LDS.  Every wave
//        walks a contiguous quarter of the stream, GPW records per step (one per lane group), several
//        steps in flight: x_b = U[u].(I[p_b]-I[n_b]); loss += softplus(-x_b); g_b = -sigmoid(-x_b)/B;
//        cur += g_b (I[p_b]-I[n_b]); coeff[b] = g_b (4 B per triplet for the item side).  When a
//        lane group's row changes, the groups closing that row combine their sums with shuffles and
//        add them to the wave's own slab of gradient rows in LDS (plain read-modify-write: no
//        atomics anywhere).  Rows of any length are thereby spread over all waves; more than kCap
//        records per bucket: further chunks into the same slabs.  Finally each lane group adds the
//        four slabs of its row in fixed order and applies Adam: U_new[u] = Adam(U[u], acc).
//   3  owner_pass_kernel<ITEM>  the same over the item buckets: {user, (+/-) coeff[b]} per
//        occurrence, acc = sum g * U_old[user], I[i] = Adam(I[i], acc) (or the dense gradient rows
//        for the multi-GPU all-reduce); workgroup 0 also reduces the loss partials of launch 2
//        (fixed order) into the step loss and the running epoch loss.
// The user table is double-buffered (U -> U_new) because the item pass needs the OLD user rows;
// the item table is updated in place (a row is read only by its own lane group).
//
// Layout: a row of D floats sits on LPR = D/4 lanes as float4 (16-byte loads/stores); a wave
// works on GPW = 64/LPR contributions at once; the per-contribution dot product is a reduction
// over LPR lanes only.  At the end a wave holds GPW consecutive rows, one per lane group: own row,
// m, v and the Adam update stream 1 KB per wave instruction.
#include "common.h"

namespace yr {

constexpr int kPartThreads = 1024;              // partition workgroup
#ifndef YR_ITEM_CAP
#define YR_ITEM_CAP 1024
#endif
constexpr int kCap = YR_ITEM_CAP;                      // records an owner workgroup sorts per chunk (item side)
constexpr int kUserCap = 768;                   // user side (12 B of LDS per record instead of 8)
constexpr int kTileGroup = 256;                 // tiles whose segment descriptors an owner holds at once
constexpr int kOccShift = 26;                   // occ.x = user | local item row << 26
constexpr int kOccMask = (1 << kOccShift) - 1;
constexpr int kMaxBuckets = 16383;              // (buckets + 1) * 4 B of dynamic LDS <= 64 KB
constexpr int kMaxOwnerGrid = 4096;
#ifndef YR_HEAVY_ROW
#define YR_HEAVY_ROW 96
#endif
constexpr int kHeavyRow = YR_HEAVY_ROW;          // records per row and chunk from which all four waves share the row
// Oversize item buckets (a few rows that take several per cent of a batch: one workgroup would walk them chunk after
// chunk, 1.3 ms at 2^20 triplets): the bucket's tiles are shared out between `parts` workgroups — the owner and
// parts - 1 helpers —, each leaves the partial sums of its tile range in a scratch slot, the LAST to arrive adds
// the slots in part order (a fixed order: the deterministic mode survives) and applies Adam.
#ifndef YR_SPLIT_MIN
#define YR_SPLIT_MIN 2048
#endif
constexpr int kSplitMin = YR_SPLIT_MIN;          // records of a bucket from which it is split
#ifndef YR_SPLIT_AVG_MIN
#define YR_SPLIT_AVG_MIN 2.5
#endif
#ifndef YR_SPLIT_AVG_TARGET
#define YR_SPLIT_AVG_TARGET 1.25
#endif
#ifndef YR_SPLIT_TARGET
#define YR_SPLIT_TARGET 1024
#endif
constexpr int kSplitTarget = YR_SPLIT_TARGET;    // records per part (one chunk of the item pass)
constexpr int kMaxParts = 64;
constexpr int kMaxTasks = 512;                   // helper workgroups at the front of the item pass's grid
constexpr int kMaxSlots = 1024;                  // scratch slots of 1024 floats (one bucket's rows)
constexpr int kBuildLanes = 4;                   // lanes that share one item bucket in the sizing workgroups of the USER pass
#ifndef YR_OWNER_WAVES
#define YR_OWNER_WAVES 8              // waves per SIMD the owner pass is compiled for (8 workgroups per CU: <= 64 VGPRs)
#endif

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
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// row gathers: uniform base + 32-bit element offset (a table holds at most 16383 buckets x 1024
// floats), so the address stays one VGPR next to an SGPR pair
__device__ __forceinline__ float4 ld4o(const float* base, uint32_t off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (uint32_t)(off * 4u));
}

template <int D>
struct PullGeom {
  static_assert(D == 16 || D == 32 || D == 64 || D == 128, "unsupported width");
  static constexpr int LPR = D / 4;              // lanes per row (float4 each)
  static constexpr int GPW = kWave / LPR;        // lane groups (= contributions per pass) per wave
  static constexpr int R = kWavesPerBlock * GPW; // rows per bucket = 1024 / D (a wave owns GPW rows)
};
// A table with few rows and many contributions per row (a rank's user shard of an 8-GPU run: 3,959 rows,
// 133 triplets per row and step) would leave most CUs without a bucket: then a wave owns ONE row and a
// bucket is 4 rows (kNarrowRows), four times the workgroups.
#ifndef YR_MIN_TILES
#define YR_MIN_TILES 64
#endif
constexpr int kMinTiles = YR_MIN_TILES;
constexpr int kNarrowRows = kWavesPerBlock;
constexpr int kNarrowShift = 2;
#ifndef YR_NARROW_BELOW
#define YR_NARROW_BELOW 768
#endif
constexpr int kNarrowBelow = YR_NARROW_BELOW;                // ... when there would be fewer 1024/D-row buckets than this

inline int bucket_shift(int D) { return D == 16 ? 6 : D == 32 ? 5 : D == 64 ? 4 : 3; }

// --------------------------------------------------------------------------- 1: tile partition
// grid (tiles, 2): blockIdx.y = 0 sorts the tile's triplets by user bucket, 1 its 2 x TILE item
// occurrences by item bucket.  off[tile][0..nb] = start of every bucket inside the tile's region.
template <int PT>
__global__ __launch_bounds__(kPartThreads) void tile_partition_kernel(
    const int64_t* __restrict__ user, const int64_t* __restrict__ pos, const int64_t* __restrict__ neg, int64_t B,
    int64_t nU, int64_t nI, int shiftU, int shiftI, int nbU, int nbI, int32_t* __restrict__ offU,
    int32_t* __restrict__ offI,
    int4* __restrict__ rec, int2* __restrict__ occ, int32_t* __restrict__ err_flag, int stage_off,
    int32_t* __restrict__ split_counters) {
  extern __shared__ int32_t s_cnt[];            // [nb + 1] bucket counters, then (16-byte aligned) the staged tile
  __shared__ int s_wave[kPartThreads / kWave];
  int4* s_stage = reinterpret_cast<int4*>(s_cnt + stage_off);   // TILE records of 16 B, or 2 TILE of 8 B
  constexpr int TILE = kPartThreads * PT;
  const int side = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  // the split block of this batch starts clean: no tasks, no slots, every item bucket whole (the user pass's sizing
  // workgroups overwrite it; an item pass run without them must not meet the previous batch's parts)
  if (side == 0 && tile == 0) {
    if (tid < 2) split_counters[tid] = 0;
    for (int i = tid; i < nbI; i += kPartThreads) split_counters[4 + i] = 1;
  }
  const int nb = side ? nbI : nbU;
  const int shift = side ? shiftI : shiftU;
  const int rmask = (1 << shift) - 1;

  int32_t u[PT], p[PT], n[PT];
  int bad = 0;
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int64_t b = (int64_t)tile * TILE + (int64_t)k * kPartThreads + tid;
    u[k] = -1; p[k] = 0; n[k] = 0;
    if (b < B) {
      const int64_t uu = user[b], pp = pos[b], nn = neg[b];
      int f = 0;
      if ((uint64_t)uu >= (uint64_t)nU) f |= YR_FLAG_BAD_USER;
      if ((uint64_t)pp >= (uint64_t)nI || (uint64_t)nn >= (uint64_t)nI) f |= YR_FLAG_BAD_ITEM;
      if (f) bad |= f;
      else { u[k] = (int32_t)uu; p[k] = (int32_t)pp; n[k] = (int32_t)nn; }
    }
  }
  for (int i = tid; i <= nb; i += kPartThreads) s_cnt[i] = 0;
  __syncthreads();
  int ra[PT], rb[PT];                           // rank inside (tile, bucket)
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    ra[k] = 0; rb[k] = 0;
    if (u[k] >= 0) {
      if (side == 0) {
        ra[k] = atomicAdd(&s_cnt[u[k] >> shift], 1);
      } else {
        ra[k] = atomicAdd(&s_cnt[p[k] >> shift], 1);
        rb[k] = atomicAdd(&s_cnt[n[k] >> shift], 1);
      }
    }
  }
  __syncthreads();
  // exclusive scan of s_cnt[0..nb) in place, s_cnt[nb] = total: each thread owns `per` consecutive entries
  {
    const int per = (nb + kPartThreads - 1) / kPartThreads;
    const int i0 = tid * per;
    int sum = 0;
    for (int i = i0; i < i0 + per && i < nb; ++i) sum += s_cnt[i];
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    int inc = sum;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int o = __shfl_up(inc, d, kWave);
      if (lane >= d) inc += o;
    }
    if (lane == kWave - 1) s_wave[wave] = inc;
    __syncthreads();
    int before = inc - sum;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    for (int i = i0; i < i0 + per && i < nb; ++i) {
      const int c = s_cnt[i];
      s_cnt[i] = before;
      before += c;
    }
    if (tid == kPartThreads - 1) s_cnt[nb] = before;
  }
  __syncthreads();
  int32_t* off = (side ? offI : offU) + (int64_t)tile * (nb + 1);
  for (int i = tid; i <= nb; i += kPartThreads) off[i] = s_cnt[i];
  // The records are first placed bucket by bucket in LDS and then written out with 16-byte coalesced
  // stores: scattering them straight to global memory (8/16 bytes per lane to 64 different lines per wave
  // instruction) was bound by the request rate of the L2, not by bytes.
  const int total = s_cnt[nb];
  if (side == 0) {
#pragma unroll
    for (int k = 0; k < PT; ++k)
      if (u[k] >= 0) {
        const int b = tile * TILE + k * kPartThreads + tid;
        s_stage[s_cnt[u[k] >> shift] + ra[k]] = make_int4(p[k], n[k], b, u[k] & rmask);
      }
    if (bad && err_flag) atomicOr(err_flag, bad);
    __syncthreads();
    int4* out = rec + (int64_t)tile * TILE;
    for (int i = tid; i < total; i += kPartThreads) out[i] = s_stage[i];
  } else {
    int2* stage2 = reinterpret_cast<int2*>(s_stage);
#pragma unroll
    for (int k = 0; k < PT; ++k)
      if (u[k] >= 0) {
        const int b = tile * TILE + k * kPartThreads + tid;
        stage2[s_cnt[p[k] >> shift] + ra[k]] = make_int2(u[k] | ((p[k] & rmask) << kOccShift), b);
        stage2[s_cnt[n[k] >> shift] + rb[k]] = make_int2(u[k] | ((n[k] & rmask) << kOccShift), b | (int)0x80000000);
      }
    __syncthreads();
    int4* out = reinterpret_cast<int4*>(occ + (int64_t)tile * TILE * 2);   // two occurrences per 16 bytes
    const int n16 = (total + 1) >> 1;
    for (int i = tid; i < n16; i += kPartThreads) out[i] = s_stage[i];
  }
}

#ifdef YR_STAMPS
__device__ long long g_stamps[8192 * 8];
#define YR_STAMP(i) do { if (threadIdx.x == 0 && first_bucket) g_stamps[(blockIdx.x + (USER ? 0 : 4096)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define YR_STAMP(i)
#endif

// --------------------------------------------------------------------------- 2, 3: owner passes
struct OwnerArgs {
  const float* own_old;      // table whose rows are updated (read)
  float* own_new;            // where the updated rows go (== own_old for the item pass)
  const float* other;        // table whose rows are gathered
  float* m;
  float* v;
  float* grad_out;           // item pass, FUSE_ADAM = false: dense gradient rows instead of Adam
  const int32_t* off;        // [T][nb + 1] bucket offsets inside every tile region
  const void* recs;          // int4 rec (user side) / int2 occ (item side), tile-major
  float* coeff;              // [B] g_b by triplet id: written by the user pass, read by the item pass
  float* loss_partials;      // user pass: one slot per workgroup; item pass workgroup 0 reduces them
  float* loss_out;           // item pass: step loss (may be null)
  double* loss_accum;        // item pass: running epoch loss (may be null)
  int finalize;              // item pass: workgroup 0 reduces the loss partials
  int nb, T, tile_stride;    // buckets on this side, tiles, records per tile region
  int rows;                  // rows of the own table
  int bucket_begin, bucket_end;
  const int32_t* order;      // optional: workgroup slot -> bucket (a permutation of [0, nb): heaviest buckets first)
  int heavy_t;               // rows with more records in a chunk are walked by all four waves
  float inv_batch;
  AdamC adam;
  // oversize item buckets (see kSplitMin): `split` = the bookkeeping block of the workspace — {tasks, slots}
  // counters (16 B), then parts[nbI], slot[nbI], arrive[nbI] (each padded to 16 B), then the task list.
  // Item pass: the first helper_blocks workgroups are helpers (task j); parts[k] > 1 marks a split bucket whose
  // parts leave their sums in scratch slot slot[k] + part.  User pass: its last build_blocks workgroups fill the
  // block for the item pass that follows, from the item-side offsets b_off [T][b_nb + 1].
  // (One pointer instead of seven: the kernel arguments live in scalar registers, and the item pass has none to spare.)
  char* split;
  float* scratch;
  const int32_t* b_off;
  int helper_blocks, build_blocks, b_nb;
  int b_split_min, b_split_target;   // records from which a bucket is split / per part (host: scaled with the batch)
};

// the pieces of the `split` block
__host__ __device__ __forceinline__ size_t split_per(int nbI) { return ((size_t)nbI * 4 + 15) & ~(size_t)15; }
__device__ __forceinline__ int32_t* split_counters(char* b) { return (int32_t*)b; }
__device__ __forceinline__ int32_t* split_parts(char* b) { return (int32_t*)(b + 16); }
__device__ __forceinline__ int32_t* split_slot(char* b, int nbI) { return (int32_t*)(b + 16 + split_per(nbI)); }
__device__ __forceinline__ int32_t* split_arrive(char* b, int nbI) { return (int32_t*)(b + 16 + 2 * split_per(nbI)); }
__device__ __forceinline__ int4* split_tasks(char* b, int nbI) { return (int4*)(b + 16 + 3 * split_per(nbI)); }

// Sizing of the item buckets (first workgroups of the user pass): kBuildLanes lanes per bucket add up its records
// over the tiles (lane j takes tiles j, j + kBuildLanes, ...), the group's first lane decides the parts; split
// buckets get scratch slots and helper tasks.
__device__ __forceinline__ void build_splits(const OwnerArgs& a, int builder) {
  const int k = (builder * kBlock + (int)threadIdx.x) / kBuildLanes, j = threadIdx.x % kBuildLanes;
  int total = 0;
  if (k < a.b_nb) {
    int t = j;
    const int64_t pitch = a.b_nb + 1;
    for (; t + 3 * kBuildLanes < a.T; t += 4 * kBuildLanes) {           // four tiles in flight per lane
      const int32_t* o0 = a.b_off + (int64_t)t * pitch + k;
      const int32_t* o1 = o0 + kBuildLanes * pitch;
      const int32_t* o2 = o1 + kBuildLanes * pitch;
      const int32_t* o3 = o2 + kBuildLanes * pitch;
      const int a0 = o0[0], b0 = o0[1], a1 = o1[0], b1 = o1[1], a2 = o2[0], b2 = o2[1], a3 = o3[0], b3 = o3[1];
      total += ((b0 - a0) + (b1 - a1)) + ((b2 - a2) + (b3 - a3));
    }
    for (; t < a.T; t += kBuildLanes) {
      const int32_t* orow = a.b_off + (int64_t)t * (a.b_nb + 1) + k;
      total += orow[1] - orow[0];
    }
  }
#pragma unroll
  for (int m = 1; m < kBuildLanes; m <<= 1) total += __shfl_xor(total, m, kWave);
  if (k >= a.b_nb || j != 0) return;
  int parts = 1;
  if (total >= a.b_split_min) {
    parts = (total + a.b_split_target - 1) / a.b_split_target;
    parts = min(parts, min(kMaxParts, a.T));
  }
  int slot = 0;
  if (parts > 1) {
    slot = atomicAdd(split_counters(a.split) + 1, parts);
    const int first = atomicAdd(split_counters(a.split), parts - 1);
    // no room left (more than kMaxSlots / kMaxTasks parts in one batch): the bucket stays whole and the task
    // entries it took are marked void
    const bool fits = slot + parts <= kMaxSlots && first + parts - 1 <= kMaxTasks;
    for (int q = 1; q < parts; ++q)
      if (first + q - 1 < kMaxTasks) split_tasks(a.split, a.b_nb)[first + q - 1] = make_int4(fits ? k : -1, q, 0, 0);
    if (!fits) parts = 1;
  }
  split_parts(a.split)[k] = parts;
  split_slot(a.split, a.b_nb)[k] = slot;
  split_arrive(a.split, a.b_nb)[k] = 0;
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

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int x, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int t = __shfl_up(x, d, kWave);
    if (lane >= d) x += t;
  }
  return x;
}

// Per lane group: the sum of the row this group finishes (a wave owns GPW rows, one per lane group at the end).
// A wave's stream is sorted by row, so a lane group sums into `cur` while its row stays the same; when the row
// changes, the groups closing the same row combine with shuffles and the owning group adds the result.  (One
// register accumulator per row with predicated adds — no cross-lane traffic until the bucket is finished — cost 12
// more VGPRs and with them the eighth workgroup per CU: measured slower and removed.)
template <int LPR, int GPW>
struct RowSums {
  float4 total;                                   // running sum of this group's own row
  __device__ __forceinline__ void clear() { total = zero4(); }
  // `cur_row` = row index inside the wave (0..GPW-1)
  __device__ __forceinline__ void book(bool closing, int cur_row, const float4& cur, int grp) {
#pragma unroll 1
    for (int j = 0; j < GPW; ++j) {
      const bool sel = closing && cur_row == j;
      if (!__ballot(sel)) continue;               // wave-uniform
      float4 v = sel ? cur : zero4();
      cross_group_sum<LPR>(v);
      if (grp == j) { total.x += v.x; total.y += v.y; total.z += v.z; total.w += v.w; }
    }
  }
  __device__ __forceinline__ float4 finish(int) { return total; }
};

#ifndef YR_USER_UNROLL
#define YR_USER_UNROLL 1
#endif
#ifndef YR_ITEM_UNROLL
#define YR_ITEM_UNROLL 2
#endif
constexpr int kUserUnroll = YR_USER_UNROLL;   // steps in flight per lane group; two gathered rows per contribution
constexpr int kItemUnroll = YR_ITEM_UNROLL;
constexpr int kTagShift = 10;    // s_idx entry = load-order index (< kCap) | local row << 10

// One pass of a wave over the stream positions lo + first + k * stride < hi (sorted by row); books
// row sums into `sums` (rows relative to `row_base`) or, with HEAVY, returns the sum of ONE row in `cur`.
template <int D, bool USER, bool HEAVY>
__device__ __forceinline__ void walk_stream(const OwnerArgs& a, const unsigned short* s_idx, const int* s_x,
                                            const int* s_y, const int* s_z, const float4* s_own, int lo, int hi,
                                            int first, int stride, int row_base, int grp, int l,
                                            RowSums<PullGeom<D>::LPR, PullGeom<D>::GPW>& sums, float4& cur,
                                            float& loss) {
  using G = PullGeom<D>;
  constexpr int LPR = G::LPR;
  constexpr int N = USER ? kUserUnroll : kItemUnroll;
  int cur_row = -1;
  // the trip count is the same for every lane of the wave (the start is common, `first` only offsets
  // the position): the loop body holds cross-lane operations
  for (int base = lo; base < hi; base += stride * N) {
    bool valid[N];
    int tag[N], ia[N], ib[N], ic[N];
    float4 r0[N], r1[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const int pos = base + first + q * stride;
      valid[q] = pos < hi;
      const int e = s_idx[min(pos, hi - 1)];                    // past the end: the last record, weight 0
      tag[q] = e >> kTagShift;
      const int c = e & ((1 << kTagShift) - 1);
      ia[q] = s_x[c];
      ib[q] = s_y[c];
      ic[q] = USER ? s_z[c] : 0;
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      r0[q] = ld4o(a.other, (uint32_t)(ia[q] * D + 4 * l));
      if (USER) r1[q] = ld4o(a.other, (uint32_t)(ib[q] * D + 4 * l));
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      if (!HEAVY) {
        const bool closing = valid[q] && cur_row >= 0 && tag[q] != cur_row;
        if (__ballot(closing)) {
          sums.book(closing, cur_row - row_base, cur, grp);
          if (closing) cur = zero4();
        }
        if (valid[q]) cur_row = tag[q];
      }
      if (USER) {
        const float4 ownj = s_own[tag[q] * LPR + l];
        float4 d;
        d.x = r0[q].x - r1[q].x; d.y = r0[q].y - r1[q].y; d.z = r0[q].z - r1[q].z; d.w = r0[q].w - r1[q].w;
        float part = ownj.x * d.x;
        part = fmaf(ownj.y, d.y, part);
        part = fmaf(ownj.z, d.z, part);
        part = fmaf(ownj.w, d.w, part);
        const float x = group_sum_dpp<LPR>(part);
        float sp, sg;
        bpr_terms(x, sp, sg);
        const float gg = valid[q] ? -sg * a.inv_batch : 0.0f;
        cur.x = fmaf(gg, d.x, cur.x); cur.y = fmaf(gg, d.y, cur.y);
        cur.z = fmaf(gg, d.z, cur.z); cur.w = fmaf(gg, d.w, cur.w);
        if (valid[q] && l == 0) {
          a.coeff[ic[q]] = gg;                    // one hand-off word per triplet; the item side applies the sign
          loss += sp;
        }
      } else {
        const float gg = valid[q] ? __int_as_float(ib[q]) : 0.0f;
        cur.x = fmaf(gg, r0[q].x, cur.x); cur.y = fmaf(gg, r0[q].y, cur.y);
        cur.z = fmaf(gg, r0[q].z, cur.z); cur.w = fmaf(gg, r0[q].w, cur.w);
      }
    }
  }
  if (!HEAVY) sums.book(cur_row >= 0, cur_row - row_base, cur, grp);
}

template <int D, bool USER, bool FUSE_ADAM, bool DET, int RPWX = 0>
__global__ __launch_bounds__(kBlock, YR_OWNER_WAVES) void owner_pass_kernel(OwnerArgs a) {
  using G = PullGeom<D>;
  constexpr int LPR = G::LPR, GPW = G::GPW;
  constexpr int RPW = RPWX ? RPWX : GPW;         // rows a wave owns (RPWX = 0: one per lane group)
  static_assert(RPW <= GPW, "a wave finishes at most one row per lane group");
  constexpr int R = kWavesPerBlock * RPW;        // rows per bucket
  constexpr int CAP = USER ? kUserCap : kCap;    // records per chunk
  constexpr int PT = CAP / kBlock;
  __shared__ int s_pre[kTileGroup + 1];          // flattened start of every tile's segment
  __shared__ int s_base[kTileGroup];             // where the segment sits in the record array
  __shared__ int s_cnt[kWave];                   // records per local row in this chunk
  __shared__ int s_start[kWave];                 // where a row's records start in the sorted stream
  __shared__ int s_light[kWave + 1];             // the same for light rows only (heavy rows: empty range)
  __shared__ int s_x[CAP];                       // user: pos     item: user            (load order)
  __shared__ int s_y[CAP];                       // user: neg     item: signed coefficient (float bits)
  __shared__ int s_z[(USER || DET) ? CAP : 1];   // triplet id (item pass: | sign bit, kept for the deterministic order only)
  __shared__ unsigned short s_idx[CAP];          // row order -> load order | local row << 10
  __shared__ float4 s_own[USER ? R * LPR : 1];   // user pass: the bucket's own rows (they enter the scores)
  __shared__ float4 s_heavy[kWavesPerBlock][LPR];
  __shared__ float s_red[kWavesPerBlock];
  __shared__ int s_scan[kWavesPerBlock];
  __shared__ int s_n;                            // DET: records kept by the current window of an oversize segment
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int grp = lane / LPR, l = lane % LPR;
  const bool finisher = grp < RPW;                // lane groups that finish a row (all of them unless RPW < GPW)
  const int row_l = wave * RPW + (finisher ? grp : 0);
  float loss = 0.0f;
  __shared__ int s_last;
  if (USER && (int)blockIdx.x < a.build_blocks) {
    // the user pass's first workgroups size the item buckets for the item pass that follows (no launch of its own;
    // first, so that they are done long before the owners)
    build_splits(a, blockIdx.x);
    return;
  }
  // item pass: the first hb workgroups are helpers of split buckets; user pass: the first hb are the sizing ones
  const int hb = USER ? a.build_blocks : a.helper_blocks;
  const bool helper = !USER && (int)blockIdx.x < hb;
  const int owners = (int)gridDim.x - hb;
#ifdef YR_STAMPS
  bool first_bucket = true;
#endif
  YR_STAMP(0);

  for (int ks = helper ? (int)blockIdx.x : a.bucket_begin + (int)blockIdx.x - hb;
       helper ? ks == (int)blockIdx.x : ks < a.bucket_end; ks += owners) {
    // workgroups start in slot order: the caller may put heavy buckets first (wave-uniform: kept in a scalar register)
    int k, part = 0, parts = 1;
    if (helper) {
      if (ks >= min(split_counters(a.split)[0], kMaxTasks)) break;
      const int4 task = split_tasks(a.split, a.nb)[ks];
      k = __builtin_amdgcn_readfirstlane(task.x);
      part = __builtin_amdgcn_readfirstlane(task.y);
      if (k < a.bucket_begin || k >= a.bucket_end) break;       // void task, or a bucket of another item chunk
    } else {
      k = __builtin_amdgcn_readfirstlane(a.order ? a.order[ks] : ks);
    }
    if (!USER && a.split) parts = __builtin_amdgcn_readfirstlane(split_parts(a.split)[k]);
    // this workgroup's share of the bucket: the tiles [t_begin, t_end)
    // (an integer division runs on the vector unit: the results are moved back to scalar registers, or they would
    // cost the item pass two VGPRs it does not have — 12 B of scratch per lane, caught by scripts/kernel_resources.py)
    const int t_begin = __builtin_amdgcn_readfirstlane(a.T * part / parts);
    const int t_end = __builtin_amdgcn_readfirstlane(a.T * (part + 1) / parts);
    const int row_f = k * R + row_l;
    const bool valid_f = finisher && row_f < a.rows;
    const uint32_t o_f = (uint32_t)(row_f * D + 4 * l);
    if (USER && finisher) s_own[row_l * LPR + l] = valid_f ? ld4o(a.own_old, o_f) : zero4();
    YR_STAMP(1);
    RowSums<LPR, GPW> sums;
    sums.clear();

    for (int tg0 = t_begin; tg0 < t_end; tg0 += kTileGroup) {
      const int nt = min(kTileGroup, t_end - tg0);
      // segment descriptors of bucket k in tiles [tg0, tg0 + nt), exclusive scan of their lengths
      int len = 0;
      if (tid < nt) {
        const int32_t* orow = a.off + (int64_t)(tg0 + tid) * (a.nb + 1) + k;
        const int o0 = orow[0], o1 = orow[1];
        len = o1 - o0;
        s_base[tid] = (tg0 + tid) * a.tile_stride + o0;
      }
      int inc = 0, before = 0;
      if (nt > kWave) {
        inc = wave_inclusive_scan(len, lane);
        if (lane == kWave - 1) s_scan[wave] = inc;
        __syncthreads();
        for (int w = 0; w < wave; ++w) before += s_scan[w];
      } else if (wave == 0) {                    // up to 64 tiles: one wave scans
        inc = wave_inclusive_scan(len, lane);
      }
      if (nt > kWave || wave == 0) {
        s_pre[tid] = before + inc - len;
        if (lane == kWave - 1) s_pre[tid + 1] = before + inc;   // entry after a wave's last (the total after the last wave)
      }
      __syncthreads();
      const int total = s_pre[nt];
      YR_STAMP(2);

      for (int c0 = 0, cend = 0, sub = 0; c0 < total; c0 = cend) {
        cend = min(total, c0 + CAP);
        // DET: which records share a chunk must not depend on the (arbitrary) order inside a tile's segment.  Chunks
        // are cut at tile boundaries; a segment larger than a chunk is taken in WINDOWS of triplet ids (a window of
        // CAP ids holds at most CAP user-side records, one of CAP / 2 ids at most CAP item occurrences): every window
        // scans the whole segment and keeps what falls into it — a fixed set, ranked by triplet id below.
        int wt = 0;
        bool windowed = false;
        if (DET) {
          int tlo = 0, thi = nt;
          while (thi - tlo > 1) {                    // the tile whose segment starts at c0 (chunks start at boundaries)
            const int mid = (tlo + thi) >> 1;
            if (s_pre[mid] <= c0) tlo = mid; else thi = mid;
          }
          wt = tlo;
          windowed = s_pre[wt + 1] - s_pre[wt] > CAP;
          if (!windowed && cend < total) {
            tlo = 0, thi = nt;
            while (thi - tlo > 1) {
              const int mid = (tlo + thi) >> 1;
              if (s_pre[mid] <= cend) tlo = mid; else thi = mid;
            }
            if (s_pre[tlo] > c0) cend = s_pre[tlo];
          }
        }
        if (tid < kWave) s_cnt[tid] = 0;
        if (DET && tid == 0) s_n = 0;
        __syncthreads();
        // load this chunk's records (flattened index -> tile by binary search) into LDS in load
        // order and rank them by local row; key = local row | rank << 8
        int key[PT];
        int n_rec = cend - c0;
        if (DET && windowed) {
          constexpr int W = USER ? CAP : CAP / 2;
          const int tile_ids = USER ? a.tile_stride : a.tile_stride >> 1;
          const int seg = s_pre[wt + 1] - s_pre[wt];
          const uint32_t rbase = (uint32_t)s_base[wt];
          const uint32_t id_lo = (uint32_t)(tg0 + wt) * (uint32_t)tile_ids + (uint32_t)sub * (uint32_t)W;
          for (int sidx = tid; sidx < seg; sidx += kBlock) {
            if (USER) {
              const int4 r = static_cast<const int4*>(a.recs)[rbase + sidx];
              if ((uint32_t)r.z - id_lo < (uint32_t)W) {
                const int slot = atomicAdd(&s_n, 1);
                s_x[slot] = r.x; s_y[slot] = r.y; s_z[slot] = r.z;
                s_idx[slot] = (unsigned short)r.w;     // the local row, until the keys below are taken
              }
            } else {
              const int2 oc = static_cast<const int2*>(a.recs)[rbase + sidx];
              const uint32_t b = (uint32_t)(oc.y & 0x7fffffff);
              if (b - id_lo < (uint32_t)W) {
                const float g = a.coeff[b];
                const int slot = atomicAdd(&s_n, 1);
                s_x[slot] = oc.x & kOccMask;
                s_y[slot] = __float_as_int(oc.y < 0 ? -g : g);
                s_z[slot] = oc.y;
                s_idx[slot] = (unsigned short)((uint32_t)oc.x >> kOccShift);
              }
            }
          }
          __syncthreads();
          n_rec = s_n;
#pragma unroll
          for (int q = 0; q < PT; ++q) {
            const int c = tid + q * kBlock;
            key[q] = -1;
            if (c < n_rec) {
              const int local = s_idx[c];
              key[q] = local | (atomicAdd(&s_cnt[local], 1) << 8);
            }
          }
          ++sub;                                       // the next window of this segment, or the next tile boundary
          if (sub * W >= tile_ids) { sub = 0; cend = s_pre[wt + 1]; }
          else cend = c0;
          if (n_rec == 0) {                            // workgroup-uniform: nothing in this window
            __syncthreads();
            continue;
          }
        } else {
        uint32_t addr[PT];
#pragma unroll
        for (int q = 0; q < PT; ++q) {
          const int j = c0 + tid + q * kBlock;
          key[q] = j < cend ? 0 : -1;
          int tlo = 0, thi = nt;
          while (thi - tlo > 1) {
            const int mid = (tlo + thi) >> 1;
            if (s_pre[mid] <= j) tlo = mid; else thi = mid;
          }
          addr[q] = j < cend ? (uint32_t)(s_base[tlo] + (j - s_pre[tlo])) : 0u;
        }
        if (USER) {
          int4 r[PT];                             // lanes past the end re-read record 0 of the array
#pragma unroll
          for (int q = 0; q < PT; ++q)
            r[q] = *reinterpret_cast<const int4*>(static_cast<const char*>(a.recs) + (uint32_t)(addr[q] * 16u));
#pragma unroll
          for (int q = 0; q < PT; ++q)
            if (key[q] >= 0) {
              const int c = tid + q * kBlock;
              s_x[c] = r[q].x; s_y[c] = r[q].y; s_z[c] = r[q].z;
              key[q] = r[q].w | (atomicAdd(&s_cnt[r[q].w], 1) << 8);
            }
        } else {
          int2 oc[PT];
          float g[PT];
#pragma unroll
          for (int q = 0; q < PT; ++q)
            oc[q] = *reinterpret_cast<const int2*>(static_cast<const char*>(a.recs) + (uint32_t)(addr[q] * 8u));
#pragma unroll
          for (int q = 0; q < PT; ++q) {
            const uint32_t b = key[q] >= 0 ? (uint32_t)(oc[q].y & 0x7fffffff) : 0u;
            g[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.coeff) + (uint32_t)(b * 4u));
          }
#pragma unroll
          for (int q = 0; q < PT; ++q)
            if (key[q] >= 0) {
              const int c = tid + q * kBlock;
              const int local = (int)((uint32_t)oc[q].x >> kOccShift);
              s_x[c] = oc[q].x & kOccMask;
              s_y[c] = __float_as_int(oc[q].y < 0 ? -g[q] : g[q]);
              if (DET) s_z[c] = oc[q].y;
              key[q] = local | (atomicAdd(&s_cnt[local], 1) << 8);
            }
        }
        }
        __syncthreads();
        // sorted stream = light rows in row order, then the heavy rows (more than heavy_t records)
        if (wave == 0) {
          const int c = s_cnt[lane];
          const bool heavy = c > a.heavy_t;
          const int cl = heavy ? 0 : c, ch = heavy ? c : 0;
          const int il = wave_inclusive_scan(cl, lane), ih = wave_inclusive_scan(ch, lane);
          const int total_light = __shfl(il, kWave - 1, kWave);
          s_light[lane] = il - cl;
          if (lane == kWave - 1) s_light[kWave] = il;
          s_start[lane] = heavy ? total_light + ih - ch : il - cl;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PT; ++q)
          if (key[q] >= 0)
            s_idx[s_start[key[q] & 255] + (key[q] >> 8)] =
                (unsigned short)((tid + q * kBlock) | ((key[q] & 255) << kTagShift));
        __syncthreads();
        if (DET) {
          // the ranks above came from LDS atomics in arrival order: re-rank every row by triplet id
          // (unique inside a row), so that the order of every floating-point sum below is fixed
          int at[PT];
          unsigned short ent[PT];
#pragma unroll
          for (int q = 0; q < PT; ++q) {
            const int pos = tid + q * kBlock;
            at[q] = -1;
            if (pos < n_rec) {
              ent[q] = s_idx[pos];
              const int r = ent[q] >> kTagShift;
              const uint32_t mine = (uint32_t)s_z[ent[q] & ((1 << kTagShift) - 1)];
              const int lo = s_start[r], cnt = s_cnt[r];
              int before = 0;
              for (int j = lo; j < lo + cnt; ++j)
                before += (uint32_t)s_z[s_idx[j] & ((1 << kTagShift) - 1)] < mine ? 1 : 0;
              at[q] = lo + before;
            }
          }
          __syncthreads();
#pragma unroll
          for (int q = 0; q < PT; ++q)
            if (at[q] >= 0) s_idx[at[q]] = ent[q];
          __syncthreads();
        }
        // light rows: wave w walks the records of ITS rows [w GPW, (w+1) GPW), one per lane group and step
        {
          float4 cur = zero4();
          walk_stream<D, USER, false>(a, s_idx, s_x, s_y, s_z, s_own, s_light[wave * RPW], s_light[wave * RPW + RPW], grp,
                                      GPW, wave * RPW, grp, l, sums, cur, loss);
        }
        // heavy rows of the chunk: all waves on one row, partial sums combined in wave order
        if (n_rec - s_light[kWave] > 0) {   // workgroup-uniform
#pragma unroll 1
          for (int r = 0; r < R; ++r) {
            const int cnt = s_cnt[r];
            if (cnt <= a.heavy_t) continue;       // workgroup-uniform
            const int lo = s_start[r];
            float4 t = zero4();
            walk_stream<D, USER, true>(a, s_idx, s_x, s_y, s_z, s_own, lo, lo + cnt, wave * GPW + grp,
                                       kWavesPerBlock * GPW, 0, grp, l, sums, t, loss);
            cross_group_sum<LPR>(t);
            if (grp == 0) s_heavy[wave][l] = t;
            __syncthreads();
            if (finisher && row_l == r) {
#pragma unroll
              for (int w = 0; w < kWavesPerBlock; ++w) {
                const float4 h = s_heavy[w][l];
                sums.total.x += h.x; sums.total.y += h.y; sums.total.z += h.z; sums.total.w += h.w;
              }
            }
            __syncthreads();
          }
        }
        __syncthreads();                         // LDS records are reused by the next chunk
      }
    }

    YR_STAMP(3);
    float4 acc = sums.finish(grp);
    if (!USER && parts > 1) {
      // a split bucket: leave this part's sums in its scratch slot; the last part to arrive adds the slots in part
      // order and goes on to the update, the others are done with the bucket
      // (the base goes through an empty asm: otherwise scratch + this lane's offset is hoisted out of the bucket
      // loop into a VGPR pair the item pass does not have — it was spilled to scratch memory)
      float* sbase = a.scratch;
      asm volatile("" : "+s"(sbase));
      float* slots = sbase + (int64_t)split_slot(a.split, a.nb)[k] * (R * D);
      if (finisher) st4(slots + (int64_t)part * (R * D) + row_l * D + 4 * l, acc);
      __threadfence();
      __syncthreads();
      if (tid == 0) s_last = atomicAdd(split_arrive(a.split, a.nb) + k, 1) == parts - 1;
      __syncthreads();
      const bool last = s_last != 0;
      __syncthreads();                           // s_last may be rewritten by the next bucket
      if (!last) continue;
      __threadfence();
      acc = zero4();
      if (finisher) {
        for (int q = 0; q < parts; ++q) {
          const float4 t = ld4(slots + (int64_t)q * (R * D) + row_l * D + 4 * l);
          acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
      }
    }
    if (valid_f) {
      if (FUSE_ADAM) {
        float4 own = USER ? s_own[row_l * LPR + l] : ld4o(a.own_old, o_f);
        float4 M = ld4o(a.m, o_f), V = ld4o(a.v, o_f);
        adam1(own.x, acc.x, M.x, V.x, a.adam);
        adam1(own.y, acc.y, M.y, V.y, a.adam);
        adam1(own.z, acc.z, M.z, V.z, a.adam);
        adam1(own.w, acc.w, M.w, V.w, a.adam);
        st4(a.own_new + o_f, own);
        st4(a.m + o_f, M);
        st4(a.v + o_f, V);
      } else {
        st4(a.grad_out + o_f, acc);
      }
    }
    YR_STAMP(4);
    if (USER) __syncthreads();                   // s_own is rewritten for the next bucket
    YR_STAMP(5);
#ifdef YR_STAMPS
    first_bucket = false;
#endif
  }

  if (USER) {
    const float total = block_sum(loss, s_red);
    if (tid == 0) a.loss_partials[(int)blockIdx.x - hb] = total;
    if ((int)blockIdx.x == hb)                    // slots no workgroup owns
      for (int i = owners + tid; i < YR_LOSS_PARTIALS; i += kBlock) a.loss_partials[i] = 0.0f;
  } else if (a.finalize && (int)blockIdx.x == hb) {
    // the user pass (previous launch) left one partial per workgroup: fixed-order sum -> step loss
    float s = 0.0f;
    for (int i = tid; i < YR_LOSS_PARTIALS; i += kBlock) s += a.loss_partials[i];
    const float total = block_sum(s, s_red);
    if (tid == 0) {
      const float v = total * a.inv_batch;
      if (a.loss_out) a.loss_out[0] = v;
      if (a.loss_accum) a.loss_accum[0] += (double)v;
    }
  }
}

__global__ __launch_bounds__(kBlock) void pull_loss_finalize_kernel(const float* __restrict__ partials, float scale,
                                                                    float* __restrict__ loss_out,
                                                                    double* __restrict__ loss_accum) {
  __shared__ float s_red[kWavesPerBlock];
  float s = 0.0f;
  for (int i = threadIdx.x; i < YR_LOSS_PARTIALS; i += kBlock) s += partials[i];
  const float total = block_sum(s, s_red);
  if (threadIdx.x == 0) {
    const float v = total * scale;
    if (loss_out) loss_out[0] = v;
    if (loss_accum) loss_accum[0] += (double)v;
  }
}

// --------------------------------------------------------------------------- host side
// How a batch of B triplets is cut into tiles, and the workspace carve-up (all 16-byte aligned).
inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
struct PullPlan {
  int pt, tile, T, shiftU, shiftI, narrow_users, nbU, nbI;
  size_t o_offU, o_offI, o_rec, o_occ, o_coeff, o_split, o_scratch, bytes;
};
// split bookkeeping inside the workspace: {tasks, slots} counters (16 B), parts / slot / arrivals per item bucket,
// the task list; then the scratch slots
inline size_t split_bytes(int nbI) { return 16 + (size_t)3 * align16((size_t)nbI * 4) + (size_t)kMaxTasks * sizeof(int4); }

// tiles of 4096 triplets, halved while there would be fewer than kMinTiles of them (the partition launch
// and the owners' segment runs both want tiles that are neither too few nor too small)
inline int tile_pt(int64_t B) {
  int pt = 4;                                    // a staged tile of 4096 triplets is 64 KB of LDS
  while (pt > 1 && (B + (int64_t)kPartThreads * pt - 1) / ((int64_t)kPartThreads * pt) < kMinTiles) pt >>= 1;
  return pt;
}

inline PullPlan make_plan(int64_t B, int64_t nU, int64_t nI, int D, bool upper_bound) {
  PullPlan p;
  p.shiftI = bucket_shift(D);
  const int R = 1 << p.shiftI;
  // user buckets of 4 rows (a wave per row) when buckets of 1024/D rows would leave most of the 2,048
  // resident owner workgroups without work (user shards of a multi-GPU run, small tables)
  p.narrow_users = (nU + R - 1) / R < kNarrowBelow && R > kNarrowRows;
  p.shiftU = p.narrow_users ? kNarrowShift : p.shiftI;
  p.nbU = (int)((nU + (1 << p.shiftU) - 1) >> p.shiftU);
  p.nbI = (int)((nI + (1 << p.shiftI) - 1) >> p.shiftI);
  p.pt = tile_pt(B);
  p.tile = kPartThreads * p.pt;
  p.T = (int)((B + p.tile - 1) / p.tile);
  // upper bound over every batch size <= B (workspace sizing): T < 2 kMinTiles whenever pt < 4
  const int64_t T = upper_bound ? ((B + 4095) / 4096 > 2 * kMinTiles ? (B + 4095) / 4096 : 2 * kMinTiles) : p.T;
  const int64_t slots = upper_bound ? B + 4096 : (int64_t)p.T * p.tile;
  size_t o = 0;
  p.o_offU = o; o += align16((size_t)T * (p.nbU + 1) * 4);
  p.o_offI = o; o += align16((size_t)T * (p.nbI + 1) * 4);
  p.o_rec = o; o += align16((size_t)slots * sizeof(int4));
  p.o_occ = o; o += align16((size_t)slots * 2 * sizeof(int2));
  p.o_coeff = o; o += align16((size_t)(B > 0 ? B : 1) * 4);
  p.o_split = o; o += align16(split_bytes(p.nbI));
  p.o_scratch = o; o += (size_t)kMaxSlots * 1024 * 4;
  p.bytes = o;
  return p;
}

}  // namespace yr

using namespace yr;

static int pull_check_common(int64_t B, int D, int64_t num_users, int64_t num_items, const void* workspace,
                             int64_t workspace_bytes) {
  if (B < 0 || num_users <= 0 || num_items <= 0 || B > 0x3fffffff) return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  // the partition's LDS histogram holds one counter per bucket of 1024/D rows; user ids share a
  // word with the local item row
  const PullPlan pl = make_plan(0, num_users, num_items, D, false);
  if (pl.nbU > kMaxBuckets || pl.nbI > kMaxBuckets) return YR_ERR_UNSUPPORTED;
  if (num_users > kOccMask) return YR_ERR_UNSUPPORTED;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) return YR_ERR_BADARG;
  if ((int64_t)make_plan(B, num_users, num_items, D, false).bytes > workspace_bytes) return YR_ERR_BADARG;
  return 0;
}

#ifdef YR_STAMPS
extern "C" int yr_debug_read_stamps(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n);
}
#endif

extern "C" int64_t yr_bpr_mf_pull_workspace_bytes(int64_t max_batch, int64_t num_users, int64_t num_items, int D) {
  if (max_batch < 0 || num_users <= 0 || num_items <= 0) return YR_ERR_BADARG;
  if (D != 16 && D != 32 && D != 64 && D != 128) return YR_ERR_UNSUPPORTED;
  return (int64_t)make_plan(max_batch, num_users, num_items, D, true).bytes;
}

// phase 1: tile partition (independent of the tables: may run ahead, e.g. under the previous
// step's all-reduce) -> workspace
extern "C" int yr_bpr_mf_pull_index(const int64_t* user, const int64_t* pos, const int64_t* neg, int64_t B, int D,
                                    int64_t num_users, int64_t num_items, void* workspace, int64_t workspace_bytes,
                                    int32_t* err_flag, void* stream) {
  const int rc = pull_check_common(B, D, num_users, num_items, workspace, workspace_bytes);
  if (rc) return rc;
  if (B == 0) return 0;
  if (!user || !pos || !neg) return YR_ERR_BADARG;
  const PullPlan p = make_plan(B, num_users, num_items, D, false);
  char* w = static_cast<char*>(workspace);
  int32_t* offU = (int32_t*)(w + p.o_offU);
  int32_t* offI = (int32_t*)(w + p.o_offI);
  int4* rec = (int4*)(w + p.o_rec);
  int2* occ = (int2*)(w + p.o_occ);
  // dynamic LDS: the bucket counters, then the staged tile (16 B per triplet on either side)
  const int stage_off = (((p.nbU > p.nbI ? p.nbU : p.nbI) + 1) + 3) & ~3;           // in ints, 16-byte aligned
  const size_t lds = (size_t)stage_off * 4 + (size_t)p.tile * 16;
  const dim3 grid(p.T, 2);
  hipStream_t s = (hipStream_t)stream;
#define YR_PART_CASE(PT)                                                                                         \
  case PT: {                                                                                                     \
    static bool raised = false;                   /* more than the default 64 KB of dynamic LDS: once per process */ \
    if (!raised && lds > 65536) {                                                                                \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_partition_kernel<PT>),        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);    \
      if (e != hipSuccess) return (int)e;                                                                        \
      raised = true;                                                                                             \
    }                                                                                                            \
    hipLaunchKernelGGL((tile_partition_kernel<PT>), grid, dim3(kPartThreads), lds, s, user, pos, neg, B, num_users, \
                       num_items, p.shiftU, p.shiftI, p.nbU, p.nbI, offU, offI, rec, occ, err_flag, stage_off,   \
                       (int32_t*)(w + p.o_split));                                                              \
  } break
  switch (p.pt) {
    YR_PART_CASE(1);
    YR_PART_CASE(2);
    YR_PART_CASE(4);
    default: return YR_ERR_BADARG;
  }
#undef YR_PART_CASE
  return launch_status();
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

// phase 2: the two owner passes over a partition built by phase 1 for the same batch
template <int D>
static int pull_apply_impl(const float* U_old, float* U_new, float* I, float* mU, float* vU, float* mI, float* vI,
                           float* gradI_out, int64_t B, int64_t nU, int64_t nI, float inv_batch, const AdamC& adam,
                           int deterministic, void* workspace, float* loss_partials, float* loss_out,
                           double* loss_accum, int phases, int64_t item_begin, int64_t item_end,
                           const int32_t* item_order, hipStream_t s) {
  const PullPlan p = make_plan(B, nU, nI, D, false);
  char* w = static_cast<char*>(workspace);
  const bool want_loss = loss_out || loss_accum;
  if (phases & YR_PULL_USER_PHASE) {
    OwnerArgs ua{};
    ua.own_old = U_old; ua.own_new = U_new; ua.other = I; ua.m = mU; ua.v = vU; ua.grad_out = nullptr;
    ua.off = (const int32_t*)(w + p.o_offU); ua.recs = w + p.o_rec; ua.coeff = (float*)(w + p.o_coeff);
    ua.loss_partials = loss_partials;
    ua.nb = p.nbU; ua.T = p.T; ua.tile_stride = p.tile; ua.rows = (int)nU;
    ua.bucket_begin = 0; ua.bucket_end = p.nbU;
    ua.order = nullptr;
    ua.heavy_t = kHeavyRow; ua.inv_batch = inv_batch; ua.adam = adam;
    // the last kSplitBuilders workgroups size the item buckets (oversize ones are shared out in the item pass)
    ua.build_blocks = (p.nbI * kBuildLanes + kBlock - 1) / kBlock;
    ua.b_off = (const int32_t*)(w + p.o_offI); ua.b_nb = p.nbI;
    {
      // oversize = well above the average bucket of THIS batch (2B occurrences over nbI buckets), and never below
      // the floors: at one epoch per step the average bucket already holds 770 records
      const double avg = 2.0 * (double)B / (double)p.nbI;
      const int by_avg_min = (int)(YR_SPLIT_AVG_MIN * avg), by_avg_target = (int)(YR_SPLIT_AVG_TARGET * avg);
      ua.b_split_min = by_avg_min > kSplitMin ? by_avg_min : kSplitMin;
      ua.b_split_target = by_avg_target > kSplitTarget ? by_avg_target : kSplitTarget;
    }
    ua.split = w + p.o_split;
    const int gu = (p.nbU < YR_LOSS_PARTIALS ? p.nbU : YR_LOSS_PARTIALS) + ua.build_blocks;   // one loss-partial slot per owner
    if (p.narrow_users && deterministic)
      hipLaunchKernelGGL((owner_pass_kernel<D, true, true, true, 1>), dim3(gu), dim3(kBlock), 0, s, ua);
    else if (p.narrow_users)
      hipLaunchKernelGGL((owner_pass_kernel<D, true, true, false, 1>), dim3(gu), dim3(kBlock), 0, s, ua);
    else if (deterministic)
      hipLaunchKernelGGL((owner_pass_kernel<D, true, true, true>), dim3(gu), dim3(kBlock), 0, s, ua);
    else
      hipLaunchKernelGGL((owner_pass_kernel<D, true, true, false>), dim3(gu), dim3(kBlock), 0, s, ua);
  }
  const bool items = (phases & YR_PULL_ITEM_PHASE) && item_end > item_begin;
  if (items) {
    OwnerArgs ia{};
    ia.own_old = I; ia.own_new = I; ia.other = U_old; ia.m = mI; ia.v = vI; ia.grad_out = gradI_out;
    ia.off = (const int32_t*)(w + p.o_offI); ia.recs = w + p.o_occ; ia.coeff = (float*)(w + p.o_coeff);
    ia.loss_partials = loss_partials; ia.loss_out = loss_out; ia.loss_accum = loss_accum;
    ia.finalize = want_loss ? 1 : 0;
    ia.nb = p.nbI; ia.T = p.T; ia.tile_stride = 2 * p.tile; ia.rows = (int)nI;
    ia.bucket_begin = (int)(item_begin >> p.shiftI);
    ia.bucket_end = (int)((item_end + (1 << p.shiftI) - 1) >> p.shiftI);
    // a start order is a permutation of ALL item buckets: used when the call covers all of them
    ia.order = (item_order && ia.bucket_begin == 0 && ia.bucket_end == p.nbI) ? item_order : nullptr;
    ia.heavy_t = kHeavyRow; ia.inv_batch = inv_batch; ia.adam = adam;
    ia.helper_blocks = kMaxTasks;
    ia.split = w + p.o_split;
    ia.scratch = (float*)(w + p.o_scratch);
    int gi = ia.bucket_end - ia.bucket_begin;
    if (gi > kMaxOwnerGrid) gi = kMaxOwnerGrid;
    gi += kMaxTasks;                              // helpers first: they start before the owners
    if (gradI_out && deterministic)
      hipLaunchKernelGGL((owner_pass_kernel<D, false, false, true>), dim3(gi), dim3(kBlock), 0, s, ia);
    else if (gradI_out)
      hipLaunchKernelGGL((owner_pass_kernel<D, false, false, false>), dim3(gi), dim3(kBlock), 0, s, ia);
    else if (deterministic)
      hipLaunchKernelGGL((owner_pass_kernel<D, false, true, true>), dim3(gi), dim3(kBlock), 0, s, ia);
    else
      hipLaunchKernelGGL((owner_pass_kernel<D, false, true, false>), dim3(gi), dim3(kBlock), 0, s, ia);
  } else if (want_loss) {
    hipLaunchKernelGGL(pull_loss_finalize_kernel, dim3(1), dim3(kBlock), 0, s, loss_partials, inv_batch, loss_out,
                       loss_accum);
  }
  return launch_status();
}

extern "C" int yr_bpr_mf_pull_item_buckets(int64_t num_items, int D) {
  if (num_items <= 0 || (D != 16 && D != 32 && D != 64 && D != 128)) return YR_ERR_BADARG;
  const int sh = bucket_shift(D);
  return (int)((num_items + (1 << sh) - 1) >> sh);
}

extern "C" int yr_bpr_mf_pull_apply_ordered(const float* U_old, float* U_new, float* I, float* mU, float* vU,
                                            float* mI, float* vI, float* gradI_out, int64_t B, int D,
                                            int64_t num_users, int64_t num_items, float inv_batch, double lr,
                                            double step_size, double bc2_sqrt, double beta1, double beta2, double eps,
                                            double weight_decay, int mode, int deterministic, void* workspace,
                                            int64_t workspace_bytes, float* loss_partials, float* loss_out,
                                            double* loss_accum, int phases, int64_t item_row_begin,
                                            int64_t item_row_end, const int32_t* item_bucket_order, void* stream) {
  int rc = pull_check_common(B, D, num_users, num_items, workspace, workspace_bytes);
  if (rc) return rc;
  if (!(phases & (YR_PULL_USER_PHASE | YR_PULL_ITEM_PHASE))) return YR_ERR_BADARG;
  if (item_row_begin < 0 || item_row_end > num_items || item_row_begin > item_row_end) return YR_ERR_BADARG;
  // item chunks are whole buckets of 1024/D rows (the last one may end with the table)
  const int R = 1 << bucket_shift(D);
  if (item_row_begin % R != 0 || (item_row_end % R != 0 && item_row_end != num_items)) return YR_ERR_BADARG;
  if (!U_old || !U_new || U_old == U_new || !I || !mU || !vU || !loss_partials) return YR_ERR_BADARG;
  if (!gradI_out && (!mI || !vI)) return YR_ERR_BADARG;
  AdamC c;
  rc = make_adam(c, lr, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, mode);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
#define YR_APPLY_CASE(DD)                                                                                       \
  case DD:                                                                                                      \
    return pull_apply_impl<DD>(U_old, U_new, I, mU, vU, mI, vI, gradI_out, B, num_users, num_items, inv_batch,  \
                               c, deterministic ? 1 : 0, workspace, loss_partials, loss_out, loss_accum, phases,      \
                               item_row_begin, item_row_end, item_bucket_order, s)
  switch (D) {
    YR_APPLY_CASE(16);
    YR_APPLY_CASE(32);
    YR_APPLY_CASE(64);
    YR_APPLY_CASE(128);
    default: return YR_ERR_UNSUPPORTED;
  }
#undef YR_APPLY_CASE
}

extern "C" int yr_bpr_mf_pull_apply(const float* U_old, float* U_new, float* I, float* mU, float* vU, float* mI,
                                    float* vI, float* gradI_out, int64_t B, int D, int64_t num_users,
                                    int64_t num_items, float inv_batch, double lr, double step_size, double bc2_sqrt,
                                    double beta1, double beta2, double eps, double weight_decay, int mode,
                                    int deterministic, void* workspace, int64_t workspace_bytes,
                                    float* loss_partials, float* loss_out, double* loss_accum, int phases,
                                    int64_t item_row_begin, int64_t item_row_end, void* stream) {
  return yr_bpr_mf_pull_apply_ordered(U_old, U_new, I, mU, vU, mI, vI, gradI_out, B, D, num_users, num_items,
                                      inv_batch, lr, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, mode,
                                      deterministic, workspace, workspace_bytes, loss_partials, loss_out, loss_accum,
                                      phases, item_row_begin, item_row_end, nullptr, stream);
}

extern "C" int yr_bpr_mf_pull_step(const float* U_old, float* U_new, float* I, float* mU, float* vU, float* mI,
                                   float* vI, float* gradI_out, const int64_t* user, const int64_t* pos,
                                   const int64_t* neg, int64_t B, int D, int64_t num_users, int64_t num_items,
                                   float inv_batch, double lr, double step_size, double bc2_sqrt, double beta1,
                                   double beta2, double eps, double weight_decay, int mode, int deterministic,
                                   void* workspace, int64_t workspace_bytes, float* loss_partials, float* loss_out,
                                   double* loss_accum, int32_t* err_flag, void* stream) {
  if (mode != YR_OPT_ADAM && mode != YR_OPT_ADAMW) return YR_ERR_UNSUPPORTED;
  if (!U_old || !U_new || U_old == U_new || !I || !mU || !vU || !loss_partials) return YR_ERR_BADARG;
  if (!gradI_out && (!mI || !vI)) return YR_ERR_BADARG;
  int rc = yr_bpr_mf_pull_index(user, pos, neg, B, D, num_users, num_items, workspace, workspace_bytes, err_flag,
                                stream);
  if (rc) return rc;
  return yr_bpr_mf_pull_apply(U_old, U_new, I, mU, vU, mI, vI, gradI_out, B, D, num_users, num_items, inv_batch, lr,
                              step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, mode, deterministic, workspace,
                              workspace_bytes, loss_partials, loss_out, loss_accum,
                              YR_PULL_USER_PHASE | YR_PULL_ITEM_PHASE, 0, num_items, stream);
}
