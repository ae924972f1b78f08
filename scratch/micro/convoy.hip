// Three independent waves per SIMD, each looping [24 matrix instructions ; a block of plain VALU]: how busy is the matrix pipe?
// Under processor sharing of the pipe the waves fall into a convoy (all in the matrix block together, then all in the VALU
// block with the pipe idle); a wave that could KEEP the pipe for its whole block (first come, first served) would not.
//   chain kinds: 1 = one dependent accumulator chain; 2 = two alternating accumulators (a ready instruction at every slot)
//   prio: 0 none; 1 = s_setprio 3 inside the matrix block; 2 = a fixed, different priority per wave of a SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
#define VALU8(x)                                                                                  \
  asm volatile("v_max_f32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
               "v_max_f32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y))

template <int CHAINS, int PRIO>
__global__ __launch_bounds__(1024) void k(float* out, int T, int valu_n, int stagger) {
  const int wave = threadIdx.x / 64;
  f32x16 acc = {0}, acc2 = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j - 3); }
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  float y = 0.999f + 1e-9f * blockIdx.x;
  if (PRIO == 2) {
    switch (wave >> 2) { case 0: __builtin_amdgcn_s_setprio(3); break; case 1: __builtin_amdgcn_s_setprio(2); break; case 2: __builtin_amdgcn_s_setprio(1); break; default: break; }
  }
  if (stagger) { for (int r = 0; r < (wave >> 2) * stagger; ++r) VALU8(x); }
  for (int t = 0; t < T; ++t) {
    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      if (CHAINS == 2 && (q & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    { float m = acc[0] + acc2[0]; x[0] += m; }           // the VALU block starts from the scores
#pragma unroll 1
    for (int r = 0; r < valu_n; ++r) { VALU8(x); VALU8(x); }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j] + acc2[j];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 12345.0f) out[threadIdx.x] = s;
}

template <int CHAINS, int PRIO>
static float run(int waves_per_simd, int T, int valu_n, int stagger) {
  float* out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k<CHAINS, PRIO><<<256, 256 * waves_per_simd>>>(out, T, valu_n, stagger);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  (void)hipFree(out);
  return best * 1e3f;
}

int main() {
  const int T = 1000;
  const float one = run<1, 0>(1, T, 0, 0);             // one wave per SIMD, matrix block only: the pipe's own time
  printf("matrix block alone, one wave per SIMD: %.1f ns per block\n", one * 1e3 / T);
  for (int valu_n : {6, 10, 16}) {
    const float v = run<1, 0>(1, T, valu_n, 0) - one;
    printf("VALU block of %d instructions: %.0f ns alone\n", valu_n * 16, v * 1e3 / T);
    for (int w : {2, 3, 4}) {
      const float ideal = one * w;
      printf("  %d waves per SIMD: pipe busy  one chain %.0f %%  +prio in block %.0f %%  +fixed prios %.0f %%  +stagger %.0f %% | two chains %.0f %%  +prio in block %.0f %%  +fixed prios %.0f %%\n", w,
             100 * ideal / run<1, 0>(w, T, valu_n, 0), 100 * ideal / run<1, 1>(w, T, valu_n, 0), 100 * ideal / run<1, 2>(w, T, valu_n, 0),
             100 * ideal / run<1, 0>(w, T, valu_n, valu_n * 2 / w + 1),
             100 * ideal / run<2, 0>(w, T, valu_n, 0), 100 * ideal / run<2, 1>(w, T, valu_n, 0), 100 * ideal / run<2, 2>(w, T, valu_n, 0));
    }
  }
  return 0;
}
