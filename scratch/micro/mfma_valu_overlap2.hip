// Follow-up of mfma_valu_overlap.hip: which instruction mixes overlap with the matrix pipe on gfx950?
//   matrix kinds: M1 one dependent accumulator chain; M2 two alternating accumulators
//   valu kinds:   V0 v_pk_fma_f32 (dependent chains); V1 v_max_f32 / v_add_u32 on 8 independent registers (asm)
//   placement:    separate waves of a SIMD (roles), or interleaved in ONE wave (mfma; n valu; mfma; n valu ...)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

#define VALU8(x)                                                                                  \
  asm volatile("v_max_f32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
               "v_max_f32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y))

template <int MK, int VK, int INTER>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int mode, int valu_n, int split_roles) {
  const int wave = threadIdx.x / 64, nw = blockDim.x / 64;
  f32x16 acc = {0}, acc2 = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j - 3); }
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  float y = 0.999f + 1e-9f * blockIdx.x;
  const bool mrole = split_roles ? wave < nw / 2 : true, vrole = split_roles ? wave >= nw / 2 : true;
  for (int it = 0; it < iters; ++it) {
    if constexpr (INTER > 0) {                       // one wave: 24 x (mfma ; INTER x 8 valu)
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        if (mode & 1) {
          if (MK == 2 && (q & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (mode & 2) {
#pragma unroll
          for (int r = 0; r < INTER; ++r) VALU8(x);
        }
      }
    } else {
      if ((mode & 1) && mrole) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
          if (MK == 2 && (q & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
      }
      if ((mode & 2) && vrole) {
        if constexpr (VK == 0) {
#pragma unroll 1
          for (int r = 0; r < valu_n; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], y, 1.0f);
          }
        } else {
#pragma unroll 1
          for (int r = 0; r < valu_n; ++r) { VALU8(x); VALU8(x); }
        }
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j] + acc2[j];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 12345.0f) out[threadIdx.x] = s;
}

template <int MK, int VK, int INTER>
static float run(int threads, int iters, int mode, int valu_n, int split) {
  float* out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MK, VK, INTER><<<256, threads>>>(out, iters, mode, valu_n, split);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MK, VK, INTER><<<256, threads>>>(out, iters, mode, valu_n, split);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms * 1e3f;
}

template <int MK, int VK>
static void roles(const char* name, int valu_n) {
  for (int threads : {512, 1024, 1536 > 1024 ? 1024 : 1536})
    printf("  %s, roles on separate waves, %d waves per SIMD: matrix only %.0f us, valu only %.0f us, both %.0f us\n", name,
           threads / 256, run<MK, VK, 0>(threads, 2000, 1, valu_n, 1), run<MK, VK, 0>(threads, 2000, 2, valu_n, 1),
           run<MK, VK, 0>(threads, 2000, 3, valu_n, 1));
}
template <int MK, int INTER>
static void inter(const char* name) {
  for (int threads : {256, 512, 768})
    printf("  %s, ONE wave interleaves 24 x (mfma ; %d valu), %d waves per SIMD: matrix only %.0f us, valu only %.0f us, both %.0f us\n",
           name, 8 * INTER, threads / 256, run<MK, 1, INTER>(threads, 2000, 1, 0, 0), run<MK, 1, INTER>(threads, 2000, 2, 0, 0),
           run<MK, 1, INTER>(threads, 2000, 3, 0, 0));
}

int main() {
  roles<1, 0>("M1 chain + v_pk_fma_f32", 24);
  roles<1, 1>("M1 chain + v_max/v_add (asm, independent)", 12);
  roles<2, 1>("M2 two accumulators + v_max/v_add", 12);
  inter<1, 1>("M1");
  inter<2, 1>("M2");
  inter<1, 2>("M1");
  return 0;
}
