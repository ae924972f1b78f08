// Do matrix-core work and VALU work of DIFFERENT waves of a SIMD overlap on gfx950?  (standalone: hipcc
// --offload-arch=gfx950 -O3 scratch/micro/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov)
// One 512-thread workgroup per CU: wave w runs on SIMD w % 4.  Waves 0-3 run a chain of dependent
// v_mfma_f32_32x32x16_bf16, waves 4-7 a chain of v_fma_f32; each role can be switched off.  Also: both in ONE wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

__device__ __forceinline__ f32x16 mfma_block(f32x16 acc, bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int q = 0; q < 24; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void valu_block(float (&x)[8], float y, int n) {
#pragma unroll 1
  for (int r = 0; r < n; ++r) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], y, 1.0f);
  }
}

// mode bit 0: matrix waves work; bit 1: VALU waves work; bit 2: every wave does both, one after the other per
// iteration (the shape of the evaluation kernel's tile loop); waves_per_simd = blockDim / 256
__global__ __launch_bounds__(1024) void overlap_kernel(float* out, int iters, int mode, int valu_n, int split_roles) {
  const int wave = threadIdx.x / 64;
  const int nw = blockDim.x / 64;
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j - 3); }
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  const float y = 0.999f + 1e-9f * blockIdx.x;
  const bool matrix_role = split_roles ? wave < nw / 2 : true;
  const bool valu_role = split_roles ? wave >= nw / 2 : true;
  for (int it = 0; it < iters; ++it) {
    if ((mode & 1) && matrix_role) acc = mfma_block(acc, a, b);
    if ((mode & 2) && valu_role) valu_block(x, y, valu_n);
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j];
  for (int j = 0; j < 8; ++j) s += x[j];
  if (s == 12345.0f) out[threadIdx.x] = s;
}

static float run(int threads, int iters, int mode, int valu_n, int split) {
  float* out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  overlap_kernel<<<256, threads>>>(out, iters, mode, valu_n, split);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  overlap_kernel<<<256, threads>>>(out, iters, mode, valu_n, split);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms * 1e3f;
}

int main() {
  const int iters = 2000;
  // valu_n chosen so that the VALU block of a wave is about as long as its matrix block (24 x 32 = 768 cycles):
  // 8 fma x 4 cycles x 24 = 768
  for (int valu_n : {24, 48, 96}) {
    printf("valu block = %d x 4 v_pk_fma_f32 + loop; matrix block = 24 mfma 32x32x16 bf16 (768 cycles)\n", valu_n);
    printf("  roles on separate waves, 2 waves per SIMD:  matrix only %.0f us, valu only %.0f us, both %.0f us\n",
           run(512, iters, 1, valu_n, 1), run(512, iters, 2, valu_n, 1), run(512, iters, 3, valu_n, 1));
    printf("  roles on separate waves, 4 waves per SIMD:  matrix only %.0f us, valu only %.0f us, both %.0f us\n",
           run(1024, iters, 1, valu_n, 1), run(1024, iters, 2, valu_n, 1), run(1024, iters, 3, valu_n, 1));
    for (int threads : {256, 512, 768, 1024})
      printf("  every wave does matrix block then valu block, %d waves per SIMD: matrix only %.0f us, valu only %.0f us, both %.0f us\n",
             threads / 256, run(threads, iters, 1, valu_n, 0), run(threads, iters, 2, valu_n, 0), run(threads, iters, 3, valu_n, 0));
  }
  return 0;
}
