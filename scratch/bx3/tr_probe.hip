// Probe of the two gfx950 idioms the bf16x3 scorer relies on, with exact small-integer data:
//  (1) ds_read_b64_tr_b16 (builtin ds_read_tr16_b64): what each lane receives from a [row][col] bf16 image
//  (2) a 32x32 f32 accumulator X[c][r] (mfma_f32_32x32x16_bf16 C layout) re-used as the B operand of a second MFMA that
//      sums over c, with the A operand K^T[d][c] fetched by transposing reads from the row-major image K[c][d].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LSB = 136;   // image row stride in bf16 (128 + 8)

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__global__ void probe(const float* K /*[32][128]*/, const float* A1 /*[32 c][16 k]*/, const float* B1 /*[16 k][32 r]*/,
                      float* X_out /*[32 c][32 r]*/, float* G_out /*[128 d][32 r]*/, float* tr_out /*[64][4]*/) {
  __shared__ __attribute__((aligned(16))) __bf16 img[32 * LSB];
  const int lane = threadIdx.x, ln = lane & 31, h = lane >> 5;
  for (int i = lane; i < 32 * 128; i += 64) img[(i / 128) * LSB + (i % 128)] = (__bf16)K[i];
  __syncthreads();
  // (1) raw transposing read of the block rows 0..3, cols 16*(lane>>4)... : group g16 = lane>>4, in-group index i16
  {
    const int g16 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    auto ptr = (__attribute__((address_space(3))) s16x4*)(img + q * LSB + 16 * g16 + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int e = 0; e < 4; ++e) { __bf16 b; short s = v[e]; __builtin_memcpy(&b, &s, 2); tr_out[lane * 4 + e] = (float)b; }
  }
  // (2) X = A1 * B1 (one MFMA), then G^T[d][r] = sum_c K[c][d] * X[c][r]
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A1[ln * 16 + 8 * h + j]; b[j] = (__bf16)B1[(8 * h + j) * 32 + ln]; }
  f32x16 X = {};
  X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, X, 0, 0, 0);
  for (int reg = 0; reg < 16; ++reg) X_out[acc_row(reg, h) * 32 + ln] = X[reg];
  for (int dblk = 0; dblk < 4; ++dblk) {
    f32x16 G = {};
    for (int s = 0; s < 2; ++s) {
      bf16x8 bx;                                    // B operand: registers 8s..8s+7 of X
      for (int j = 0; j < 8; ++j) bx[j] = (__bf16)X[8 * s + j];
      // A operand: K^T[d = 32 dblk + ln][c = 16 s + 8 (j>>2) + 4 h + (j&3)]  — two transposing reads (j = 0..3, 4..7)
      const int g16 = (lane >> 4) & 1;              // which half of the 32 d's this 16-lane group covers
      const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
      bf16x8 ak;
      for (int half = 0; half < 2; ++half) {
        const int r0 = 16 * s + 8 * half + 4 * h;
        auto ptr = (__attribute__((address_space(3))) s16x4*)(img + (r0 + q) * LSB + 32 * dblk + 16 * g16 + 4 * p);
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
        for (int e = 0; e < 4; ++e) { __bf16 t; short sv = v[e]; __builtin_memcpy(&t, &sv, 2); ak[4 * half + e] = t; }
      }
      G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, bx, G, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) G_out[(32 * dblk + acc_row(reg, h)) * 32 + ln] = G[reg];
  }
}

int main() {
  std::vector<float> K(32 * 128), A1(32 * 16), B1(16 * 32);
  for (int c = 0; c < 32; ++c) for (int d = 0; d < 128; ++d) K[c * 128 + d] = (float)((c * 7 + d * 3) % 13 - 6);
  for (int c = 0; c < 32; ++c) for (int k = 0; k < 16; ++k) A1[c * 16 + k] = (float)((c + 2 * k) % 5 - 2);
  for (int k = 0; k < 16; ++k) for (int r = 0; r < 32; ++r) B1[k * 32 + r] = (float)((3 * k + r) % 7 - 3);
  float *dK, *dA, *dB, *dX, *dG, *dT;
  hipMalloc(&dK, K.size() * 4); hipMalloc(&dA, A1.size() * 4); hipMalloc(&dB, B1.size() * 4);
  hipMalloc(&dX, 32 * 32 * 4); hipMalloc(&dG, 128 * 32 * 4); hipMalloc(&dT, 64 * 4 * 4);
  hipMemcpy(dK, K.data(), K.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dA, A1.data(), A1.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B1.data(), B1.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dK, dA, dB, dX, dG, dT);
  std::vector<float> X(32 * 32), G(128 * 32), T(64 * 4);
  hipMemcpy(X.data(), dX, X.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(G.data(), dG, G.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(T.data(), dT, T.size() * 4, hipMemcpyDeviceToHost);
  // (1) expectation from the guide: lane i of the 16-lane group receives column i of the 4 rows (row q in element q)
  int bad_tr = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 4; ++e) {
      const int g16 = lane >> 4, i16 = lane & 15;
      const float want = K[e * 128 + 16 * g16 + i16];
      if (T[lane * 4 + e] != want) { if (bad_tr < 8) printf("tr lane %d e %d: got %g want %g\n", lane, e, T[lane * 4 + e], want); ++bad_tr; }
    }
  int bad_x = 0, bad_g = 0;
  std::vector<float> Xr(32 * 32, 0.f);
  for (int c = 0; c < 32; ++c) for (int r = 0; r < 32; ++r) { float s = 0; for (int k = 0; k < 16; ++k) s += A1[c * 16 + k] * B1[k * 32 + r]; Xr[c * 32 + r] = s; if (X[c * 32 + r] != s) ++bad_x; }
  for (int d = 0; d < 128; ++d) for (int r = 0; r < 32; ++r) { float s = 0; for (int c = 0; c < 32; ++c) s += K[c * 128 + d] * Xr[c * 32 + r]; if (G[d * 32 + r] != s) { if (bad_g < 8) printf("G d %d r %d: got %g want %g\n", d, r, G[d * 32 + r], s); ++bad_g; } }
  printf("tr mismatches %d, X mismatches %d, G mismatches %d\n", bad_tr, bad_x, bad_g);
  return (bad_tr || bad_x || bad_g) ? 1 : 0;
}
