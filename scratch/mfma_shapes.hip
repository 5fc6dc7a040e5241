// Micro-experiment: sustained f32-input MFMA rate of 32x32x2 vs 16x16x4 on random operands (does one shape hold a higher clock?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k32(const float* in, float* out, int iters) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(const float* in, float* out, int iters) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 2048 * 256 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000, blocks = 512;            // 2 workgroups per CU = 2 waves per SIMD (the score kernel's occupancy)
  for (int rep = 0; rep < 8; ++rep) {
    for (int which = 0; which < 2; ++which) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per wave per iter: 32x32x2: 4 MFMAs x 4096 flop; 16x16x4: 16 x 2048 flop  -> both 16384*... per wave
      double flop = (double)blocks * 4 * iters * (which == 0 ? 4 * 4096.0 : 16 * 2048.0);
      printf("%s: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "32x32x2" : "16x16x4", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
