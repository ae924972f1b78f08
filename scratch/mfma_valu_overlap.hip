// Do VALU instructions of one wave issue while another wave of the same SIMD runs an MFMA chain?
// Workgroup = 8 waves (two per SIMD).  Waves 0-3: chain of v_mfma_f32_32x32x2_f32; waves 4-7: dependent VALU chain.
// hipcc -O3 --offload-arch=gfx950 scratch/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int mode, float a, float b) {
  const int wave = threadIdx.x / 64;
  const bool do_mfma = wave < 4 ? (mode & 1) : false;
  const bool do_valu = wave >= 4 ? (mode & 2) : false;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = (float)(threadIdx.x + r);
  float x = (float)threadIdx.x, y = 1.0001f;
  const long long t0 = clock64();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) x = fmaf(x, y, a);      // 64 dependent VALU instructions per iteration
    }
  }
  const long long t1 = clock64();
  float s = x;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  const int grid = 256, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * grid * 512);
  hipMalloc(&cyc, sizeof(long long) * grid * 8);
  long long h[8];
  for (int mode = 1; mode <= 3; ++mode) {
    k<<<grid, 512>>>(out, cyc, iters, mode, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (%s): mfma wave %.1f cycles per MFMA, valu wave %.2f cycles per VALU instruction\n", mode,
           mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both",
           (double)h[0] / (iters * 32.0), (double)h[4] / (iters * 64.0));
  }
  return 0;
}
