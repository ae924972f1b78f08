// What the f32 matrix cores of this device sustain: v_mfma_f32_32x32x2_f32 chains, no memory traffic.
// hipcc -O3 --offload-arch=gfx950 scratch/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_chain(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = (float)(threadIdx.x + c + r);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
void run(int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu, iters = 4000;
  float* out;
  hipMalloc(&out, sizeof(float) * grid * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_chain<CHAINS><<<grid, 256>>>(out, 100, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_chain<CHAINS><<<grid, 256>>>(out, iters, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * 8 * CHAINS * 4096.0;
  printf("chains %d, workgroups per CU %d: %.1f TFLOP/s (%.2f ms)\n", CHAINS, wgs_per_cu, flop / ms / 1e9, ms);
  hipFree(out);
}

int main() {
  run<1>(1); run<1>(2); run<2>(1); run<2>(2); run<4>(1); run<4>(2);
  return 0;
}
