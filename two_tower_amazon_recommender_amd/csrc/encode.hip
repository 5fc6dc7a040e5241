// Id encoding on the GPU: the reference maps every user_id / parent_asin string to the
// rank of the string among the sorted distinct strings —
//   scripts/data_processing/prepare_training_data.py:113-123,209-210  (sorted(unique) + enumerate + Series.map)
//   src/data/preprocessor.py:478-491                                    (LabelEncoder().fit_transform)
// — int64 codes.  Python str order = Unicode code-point order = UTF-8 byte order, so the GPU sorts the zero-padded
// UTF-8 bytes: LSD radix over 8-byte big-endian chunks (stable rocPRIM radix sort of (chunk key, permutation) per
// chunk, last chunk first), then head flags (row != previous row), an inclusive scan, and a scatter of rank-1 to the
// original positions.  Integer/byte work, HBM-bound: every pass streams n*(8+4) bytes a few times; the string matrix
// itself is gathered once per chunk.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

__global__ __launch_bounds__(256) void iota_kernel(int32_t* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

// key[i] = big-endian u64 of bytes [8*chunk, 8*chunk+8) of row perm[i]
__global__ __launch_bounds__(256) void chunk_key_kernel(const uint8_t* __restrict__ rows, int64_t n, int width, int chunk,
                                                        const int32_t* __restrict__ perm, uint64_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = *reinterpret_cast<const uint64_t*>(rows + (int64_t)perm[i] * width + 8 * chunk);   // width % 8 == 0
  key[i] = __builtin_bswap64(v);
}

__global__ __launch_bounds__(256) void head_flag_kernel(const uint8_t* __restrict__ rows, int64_t n, int width,
                                                        const int32_t* __restrict__ perm, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int f = 1;
  if (i > 0) {
    const uint64_t* a = reinterpret_cast<const uint64_t*>(rows + (int64_t)perm[i] * width);
    const uint64_t* b = reinterpret_cast<const uint64_t*>(rows + (int64_t)perm[i - 1] * width);
    f = 0;
    for (int w = 0; w < width / 8; ++w) f |= (a[w] != b[w]);
  }
  flag[i] = f;
}

__global__ __launch_bounds__(256) void code_scatter_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ rank,
                                                           int64_t n, int64_t* __restrict__ codes, int32_t* __restrict__ n_unique) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  codes[perm[i]] = (int64_t)rank[i] - 1;
  if (i == n - 1 && n_unique != nullptr) *n_unique = rank[i];
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct EncWs {
  int64_t off_key_a, off_key_b, off_perm_a, off_perm_b, off_temp, temp_bytes, total;
};

EncWs enc_ws(int64_t n) {
  EncWs w{};
  size_t t_sort = 0, t_scan = 0;
  uint64_t* k = nullptr;
  int32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t_sort, k, k, v, v, (size_t)n, 0u, 64u, (hipStream_t) nullptr, false);
  (void)rocprim::inclusive_scan(nullptr, t_scan, v, v, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t) nullptr, false);
  w.temp_bytes = (int64_t)(t_sort > t_scan ? t_sort : t_scan);
  int64_t o = 0;
  w.off_key_a = o;  o = align_up(o + n * 8, 256);
  w.off_key_b = o;  o = align_up(o + n * 8, 256);
  w.off_perm_a = o; o = align_up(o + n * 4, 256);
  w.off_perm_b = o; o = align_up(o + n * 4, 256);
  w.off_temp = o;   o = align_up(o + w.temp_bytes, 256);
  w.total = o + 256;
  return w;
}

}  // namespace

extern "C" int64_t tt_encode_ids_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return enc_ws(n).total;
}

extern "C" int tt_encode_ids_u8(const uint8_t* rows, int64_t n, int32_t width, void* workspace, int64_t workspace_bytes,
                                int64_t* codes, int32_t* n_unique, tt_stream_t stream_) {
  TT_REQUIRE(n >= 0 && n <= 0x7fffffff, "tt_encode_ids_u8: n must fit in int32");
  TT_REQUIRE(width > 0 && width % 8 == 0, "tt_encode_ids_u8: width must be a positive multiple of 8 (zero-pad the strings)");
  if (n == 0) return TT_OK;
  TT_REQUIRE(rows && workspace && codes, "tt_encode_ids_u8: null pointer");
  TT_REQUIRE((reinterpret_cast<uintptr_t>(rows) & 7u) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
             "tt_encode_ids_u8: rows must be 8-byte aligned, workspace 256-byte aligned");
  const EncWs w = enc_ws(n);
  if (workspace_bytes < w.total)
    return tt::fail(TT_ERR_WORKSPACE, "tt_encode_ids_u8: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)w.total);
  hipStream_t stream = tt::as_stream(stream_);
  char* ws = static_cast<char*>(workspace);
  uint64_t* key_a = reinterpret_cast<uint64_t*>(ws + w.off_key_a);
  uint64_t* key_b = reinterpret_cast<uint64_t*>(ws + w.off_key_b);
  int32_t* perm_a = reinterpret_cast<int32_t*>(ws + w.off_perm_a);
  int32_t* perm_b = reinterpret_cast<int32_t*>(ws + w.off_perm_b);
  void* temp = ws + w.off_temp;
  size_t temp_bytes = (size_t)w.temp_bytes;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  tt::ProfScope prof("encode_ids", stream);
  hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, stream, perm_a, n);
  for (int chunk = width / 8 - 1; chunk >= 0; --chunk) {          // LSD: least significant chunk first, stable
    hipLaunchKernelGGL(chunk_key_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, width, chunk, perm_a, key_a);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, key_a, key_b, perm_a, perm_b, (size_t)n, 0u, 64u, stream, false);
    if (e != hipSuccess) return tt::fail(TT_ERR_LAUNCH, "tt_encode_ids_u8: radix sort: %s", hipGetErrorString(e));
    int32_t* t = perm_a; perm_a = perm_b; perm_b = t;
  }
  int32_t* flag = reinterpret_cast<int32_t*>(key_a);               // key buffers are free now
  int32_t* rank = reinterpret_cast<int32_t*>(key_b);
  hipLaunchKernelGGL(head_flag_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, width, perm_a, flag);
  hipError_t e = rocprim::inclusive_scan(temp, temp_bytes, flag, rank, (size_t)n, rocprim::plus<int32_t>(), stream, false);
  if (e != hipSuccess) return tt::fail(TT_ERR_LAUNCH, "tt_encode_ids_u8: scan: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(code_scatter_kernel, dim3(blocks), dim3(256), 0, stream, perm_a, rank, n, codes, n_unique);
  return tt::check_launch("tt_encode_ids_u8");
}
