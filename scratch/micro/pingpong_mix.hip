// pingpong_phases.hip with other instruction kinds in the non-matrix interval: which of them stop overlapping with the
// partner wave's 24 v_mfma_f32_32x32x16_bf16?   kinds: 1 plain VALU, 2 + 12 ds_read_b128 / 2 ds_write_b128,
// 4 + 2 global_load_dwordx4 (consumed one interval later), 8 + ballot-and-branch blocks, 16 + s_memtime stamps
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
#define VALU8(x)                                                                                  \
  asm volatile("v_max_f32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
               "v_max_f32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" \
               : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y))

__global__ __launch_bounds__(512) void pp(float* out, const uint4* __restrict__ src, int T, int matrix, int kinds, int valu_n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][32 * 400];
  const int wave = threadIdx.x / 64, role = wave >> 2, lane = threadIdx.x & 63;
  f32x16 acc = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j - 3); }
  bf16x8 B[12];
  for (int q = 0; q < 12; ++q) for (int j = 0; j < 8; ++j) B[q][j] = (__bf16)(float)(threadIdx.x + j + q);
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
  float y = 0.999f + 1e-9f * blockIdx.x;
  uint4 st0 = make_uint4(0, 0, 0, 0), st1 = st0;
  uint4 op[12];
  for (int j = 0; j < 12; ++j) op[j] = st0;
  long long stamp = 0;
  for (int q = threadIdx.x; q < 2 * 32 * 400 / 16; q += 512) reinterpret_cast<uint4*>(&lds[0][0])[q] = st0;
  __syncthreads();
  if (role == 1) __syncthreads();
  for (int t = 0; t < T; ++t) {
    if (matrix == 1) {
#pragma unroll
      for (int q = 0; q < 24; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else if (matrix == 2) {                          // distinct operand registers, the evaluation kernel's order
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, op[kb]), a2 = __builtin_bit_cast(bf16x8, op[4 + kb]), a3 = __builtin_bit_cast(bf16x8, op[8 + kb]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, B[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B[8 + kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B[4 + kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B[4 + kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B[kb], acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (kinds & 4) {                                   // the rows loaded one interval ago -> LDS
      *reinterpret_cast<uint4*>(&lds[t & 1][(threadIdx.x & 255) * 16 + 6400 * role]) = st0;
      *reinterpret_cast<uint4*>(&lds[t & 1][(threadIdx.x & 255) * 16 + 4096 + 6400 * role]) = st1;
    }
    if (kinds & 8) {
#pragma unroll 1
      for (int r = 0; r < 8; ++r) {
        if (__ballot(x[r & 7] > 1.0e30f) != 0ull) { x[0] += 1.0f; }
        asm volatile("" : "+v"(x[0]));
      }
    }
    if (kinds & 16) stamp += clock64();
    if (kinds & 32) { float m = acc[0]; for (int j = 1; j < 16; ++j) m = fmaxf(m, acc[j]); if (__ballot(m > 1.0e30f) != 0ull) x[1] += 1.0f; }
    if (kinds & 1) {
#pragma unroll 1
      for (int r = 0; r < valu_n; ++r) { VALU8(x); VALU8(x); }
    }
    if (kinds & 16) stamp += clock64();
    if (kinds & 2) {
      const unsigned char* p = &lds[(t + 1) & 1][(lane & 31) * 400 + 16 * (lane >> 5)];
#pragma unroll
      for (int j = 0; j < 12; ++j) op[j] = *reinterpret_cast<const uint4*>(p + 32 * j);
      a = __builtin_bit_cast(bf16x8, op[0]);           // (the matrix block's operands come from here: zeros)
      if (kinds & 32) { for (int j = 0; j < 16; ++j) acc[j] = 0.0f; }
    }
    if (kinds & 4) {
      const uint4* g = src + ((size_t)(t & 1023) * 1024 + (threadIdx.x & 255) + 256 * role + 64 * (blockIdx.x & 7));
      st0 = g[0];
      st1 = g[512];
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (role == 0) __syncthreads();
  float s = (float)stamp;
  for (int j = 0; j < 16; ++j) s += acc[j];
  for (int j = 0; j < 8; ++j) s += x[j];
  for (int j = 0; j < 12; ++j) s += op[j].x;
  s += st0.x + st1.y;
  if (s == 12345.0f) out[threadIdx.x] = s;
}

static float run(const uint4* src, int T, int matrix, int kinds, int valu_n) {
  float* out; (void)hipMalloc(&out, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    pp<<<248, 512>>>(out, src, T, matrix, kinds, valu_n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  (void)hipFree(out);
  return best * 1e3f;
}

int main() {
  const int T = 1200;
  uint4* src; (void)hipMalloc(&src, (size_t)1100 * 1024 * 16); (void)hipMemset(src, 0, (size_t)1100 * 1024 * 16);
  const char* names[] = {"plain VALU", "+ LDS reads", "+ global loads -> LDS", "+ LDS reads + global loads", "+ ballot branches", "+ all", "+ all + 2 clock stamps",
                         "+ all + accumulator read and reset"};
  const int kinds[] = {1, 1 | 2, 1 | 4, 1 | 2 | 4, 1 | 8, 1 | 2 | 4 | 8, 1 | 2 | 4 | 8 | 16, 1 | 2 | 4 | 8 | 32};
  for (int mk = 1; mk <= 2; ++mk) {
    printf("matrix block: %s\n", mk == 1 ? "one operand register pair" : "12 + 12 operand registers (A from the LDS reads), the evaluation kernel's order");
    for (int v = 0; v < 8; ++v) {
      const float e = run(src, T, 0, kinds[v], 6), both = run(src, T, mk, kinds[v], 6), m = run(src, T, mk, 0, 6);
      printf("  %-34s per interval: matrix only %.0f ns, other only %.0f ns, both %.0f ns\n", names[v], m * 1e3 / (2 * T), e * 1e3 / (2 * T), both * 1e3 / (2 * T));
    }
  }
  return 0;
}
