// The two-role ("ping-pong") phase structure in isolation: 512-thread workgroups, one per CU; waves 4-7 run one interval behind
// waves 0-3; every interval one half issues 24 dependent v_mfma_f32_32x32x16_bf16 and the other a block of plain VALU
// (v_max_f32 / v_add_u32 on 8 independent registers), a workgroup barrier between intervals.  Per interval: max or sum?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
#define VALU8(x)                                                                                  \
  asm volatile("v_max_f32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
               "v_max_f32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y))

// mode bit 0: matrix block; bit 1: valu block; bit 2: the valu block reads the accumulator first (as the epilogue does)
__global__ __launch_bounds__(512) void pp(float* out, int T, int mode, int valu_n) {
  const int wave = threadIdx.x / 64, role = wave >> 2;
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j - 3); }
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  float y = 0.999f + 1e-9f * blockIdx.x;
  __syncthreads();
  if (role == 1) __syncthreads();
  for (int t = 0; t < T; ++t) {
    if (mode & 1) {
#pragma unroll
      for (int q = 0; q < 24; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (mode & 2) {
      if (mode & 4) { float m = acc[0]; for (int j = 1; j < 16; ++j) m = fmaxf(m, acc[j]); x[0] += m; }
#pragma unroll 1
      for (int r = 0; r < valu_n; ++r) { VALU8(x); VALU8(x); }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (role == 0) __syncthreads();
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 12345.0f) out[threadIdx.x] = s;
}

static float run(int T, int mode, int valu_n) {
  float* out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    pp<<<256, 512>>>(out, T, mode, valu_n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  (void)hipFree(out);
  return best * 1e3f;
}

int main() {
  const int T = 2000;
  for (int valu_n : {4, 8, 12, 16}) {
    const float m = run(T, 1, valu_n), v = run(T, 2, valu_n), both = run(T, 3, valu_n), both_dep = run(T, 7, valu_n);
    printf("valu block %3d instructions: matrix only %.0f us, valu only %.0f us, both %.0f us, both with the valu block reading the accumulator %.0f us"
           "   (per interval: %.0f / %.0f / %.0f / %.0f ns)\n", valu_n * 16, m, v, both, both_dep,
           m * 1e3 / (2 * T), v * 1e3 / (2 * T), both * 1e3 / (2 * T), both_dep * 1e3 / (2 * T));
  }
  return 0;
}
