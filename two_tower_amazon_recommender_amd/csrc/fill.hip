// Counter-based synthetic inputs, bit-identical to oracle/synth.py.
// HBM-write-bound streaming kernels; 16 B per lane stores.
#include "common.h"

namespace {

__device__ __forceinline__ float u24_to_f32(uint64_t x, float lo, float scale) {
  float u = __fmul_rn((float)(uint32_t)(x >> 40), 5.9604644775390625e-08f);  // * 2^-24, exact
  return __fadd_rn(__fmul_rn(u, scale), lo);                                 // two roundings, no fma
}

__global__ __launch_bounds__(256) void fill_uniform_kernel(float* __restrict__ dst, int64_t n, uint64_t key,
                                                           int64_t start, float lo, float scale) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    const uint64_t base = key + (uint64_t)(start + 4 * v);
    tt::f32x4 o;
    o[0] = u24_to_f32(tt::splitmix(base + 0), lo, scale);
    o[1] = u24_to_f32(tt::splitmix(base + 1), lo, scale);
    o[2] = u24_to_f32(tt::splitmix(base + 2), lo, scale);
    o[3] = u24_to_f32(tt::splitmix(base + 3), lo, scale);
    *reinterpret_cast<tt::f32x4*>(dst + 4 * v) = o;
  }
  // tail (n % 4)
  const int64_t t = (nvec << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = u24_to_f32(tt::splitmix(key + (uint64_t)(start + t)), lo, scale);
}

__global__ __launch_bounds__(256) void fill_rows_kernel(float* __restrict__ dst, int64_t n_rows, int dim4, int64_t row_start,
                                                        int64_t row_stride, uint64_t key, float lo, float scale) {
  const int64_t nvec = n_rows * dim4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    const int64_t lr = v / dim4;
    const int c = (int)(v - lr * dim4);
    const uint64_t base = key + (uint64_t)((row_start + lr * row_stride) * (4 * (int64_t)dim4) + 4 * c);
    tt::f32x4 o;
    o[0] = u24_to_f32(tt::splitmix(base + 0), lo, scale);
    o[1] = u24_to_f32(tt::splitmix(base + 1), lo, scale);
    o[2] = u24_to_f32(tt::splitmix(base + 2), lo, scale);
    o[3] = u24_to_f32(tt::splitmix(base + 3), lo, scale);
    *reinterpret_cast<tt::f32x4*>(dst + 4 * v) = o;
  }
}

__global__ __launch_bounds__(256) void fill_ids_kernel(int64_t* __restrict__ dst, int64_t n, uint64_t key,
                                                       int64_t start, int64_t num_rows, int variant) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t x = tt::splitmix(key + (uint64_t)(start + i));
    int64_t id;
    if (variant == TT_IDS_UNIFORM) {
      id = (int64_t)(((x >> 32) * (uint64_t)num_rows) >> 32);
    } else {
      const double u = __dmul_rn((double)(x >> 11), 1.1102230246251565e-16);  // * 2^-53, exact
      const double u2 = __dmul_rn(u, u);
      const double u4 = __dmul_rn(u2, u2);
      id = (int64_t)floor(__dmul_rn((double)num_rows, u4));
      if (id > num_rows - 1) id = num_rows - 1;
    }
    dst[i] = id;
  }
}

uint64_t stream_key(uint64_t seed, uint64_t tensor_id) {
  return tt::splitmix_host(tt::splitmix_host(seed) ^ (tensor_id * 0xD6E8FEB86659FD93ull));
}

int grid_for(int64_t work_items) {
  int64_t blocks = (work_items + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  return (int)blocks;
}

}  // namespace

extern "C" int tt_fill_uniform_f32(float* dst, int64_t n, uint64_t seed, uint64_t tensor_id, int64_t start,
                                   float lo, float scale, tt_stream_t stream) {
  TT_REQUIRE(dst != nullptr && n >= 0 && start >= 0, "tt_fill_uniform_f32: bad dst/n/start");
  TT_REQUIRE(tt::aligned16(dst), "tt_fill_uniform_f32: dst must be 16-byte aligned");
  if (n == 0) return TT_OK;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, tt::as_stream(stream), dst, n,
                     stream_key(seed, tensor_id), start, lo, scale);
  return tt::check_launch("tt_fill_uniform_f32");
}

extern "C" int tt_fill_uniform_rows_f32(float* dst, int64_t n_rows, int32_t dim, int64_t row_start, int64_t row_stride,
                                        uint64_t seed, uint64_t tensor_id, float lo, float scale, tt_stream_t stream) {
  TT_REQUIRE(dst != nullptr && n_rows >= 0 && row_start >= 0 && row_stride > 0, "tt_fill_uniform_rows_f32: bad arguments");
  TT_REQUIRE(dim > 0 && dim % 4 == 0 && tt::aligned16(dst), "tt_fill_uniform_rows_f32: dim %% 4 == 0 and 16-byte alignment required");
  if (n_rows == 0) return TT_OK;
  hipLaunchKernelGGL(fill_rows_kernel, dim3(grid_for(n_rows * (dim / 4))), dim3(256), 0, tt::as_stream(stream), dst, n_rows,
                     dim / 4, row_start, row_stride, stream_key(seed, tensor_id), lo, scale);
  return tt::check_launch("tt_fill_uniform_rows_f32");
}

extern "C" int tt_fill_ids_i64(int64_t* dst, int64_t n, uint64_t seed, uint64_t tensor_id, int64_t start,
                               int64_t num_rows, int32_t variant, tt_stream_t stream) {
  TT_REQUIRE(dst != nullptr && n >= 0 && start >= 0 && num_rows > 0, "tt_fill_ids_i64: bad dst/n/start/num_rows");
  TT_REQUIRE(num_rows <= (int64_t)1 << 32, "tt_fill_ids_i64: num_rows must be <= 2^32");
  TT_REQUIRE(variant == TT_IDS_UNIFORM || variant == TT_IDS_POWERLAW, "tt_fill_ids_i64: unknown variant %d", variant);
  if (n == 0) return TT_OK;
  hipLaunchKernelGGL(fill_ids_kernel, dim3(grid_for(n)), dim3(256), 0, tt::as_stream(stream), dst, n,
                     stream_key(seed, tensor_id), start, num_rows, variant);
  return tt::check_launch("tt_fill_ids_i64");
}
