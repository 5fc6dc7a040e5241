// Id encoding on the GPU: the reference maps every user_id / parent_asin string to the
// rank of the string among the sorted distinct strings —
//   scripts/data_processing/prepare_training_data.py:113-123,209-210  (sorted(unique) + enumerate + Series.map)
//   src/data/preprocessor.py:478-491                                    (LabelEncoder().fit_transform)
// — int64 codes.  Python str order = Unicode code-point order = UTF-8 byte order, so the GPU sorts the zero-padded UTF-8
// rows byte-wise.  All of it is hand-written (r03; r01-r02 used rocPRIM's device radix sort and scan):
//
//   sort    LSD radix over the BYTES of the rows, last byte first: a permutation is sorted, the string matrix is only read
//           (one byte of row perm[i] per element and pass).  Per pass three launches:
//             digit_hist_kernel     1024 elements per workgroup, 256-bin histogram in LDS -> table[digit][workgroup]
//             exclusive scan        of the digit-major table (the scan below) = where each (digit, workgroup) run starts
//             digit_scatter_kernel  the same 1024 elements again; stable rank inside the workgroup by the wave-ballot
//                                   match-any ranking of csrc/part_sort.h (rounds in element order, leaders bump LDS
//                                   counters, digit-major / wave-minor offsets) -> perm_out[start + rank] = perm_in[i]
//   flags   head_flag_kernel: row perm[i] != row perm[i-1]
//   scan    two-level exclusive scan: 2048-element workgroup scans + one workgroup over the workgroup sums + add-back
//   scatter codes[perm[i]] = (flags before i) + flag[i] - 1
// Integer / byte work, HBM-bound, one-shot (not on the per-step path): width passes of ~12 n bytes each plus n scattered
// byte reads.
#include "common.h"
#include "part_sort.h"
#include <cstring>

namespace {

constexpr int EB = 1024;          // elements per workgroup of the radix passes (4 waves x 4 rounds x 64 lanes)
constexpr int SB = 2048;          // elements per workgroup of the scan (1024 threads x 2)

__global__ __launch_bounds__(256) void iota_kernel(int32_t* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

// element e of workgroup b: index b*EB + (w*4 + r)*64 + lane  (wave w, round r): consecutive in memory per (wave, round)
__device__ __forceinline__ int64_t elem_index(int64_t blk, int w, int r, int lane) { return blk * EB + (w * 4 + r) * 64 + lane; }

__global__ __launch_bounds__(256) void digit_hist_kernel(const uint8_t* __restrict__ rows, int width, int byte,
                                                         const int32_t* __restrict__ perm, int64_t n, int32_t* __restrict__ table,
                                                         int64_t nblk) {
  __shared__ int hist[256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  hist[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = elem_index(blockIdx.x, w, r, lane);
    if (i < n) atomicAdd(&hist[rows[(int64_t)perm[i] * width + byte]], 1);
  }
  __syncthreads();
  table[(int64_t)tid * nblk + blockIdx.x] = hist[tid];       // digit-major: the scan turns it into run starts
}

__global__ __launch_bounds__(256) void digit_scatter_kernel(const uint8_t* __restrict__ rows, int width, int byte,
                                                            const int32_t* __restrict__ perm_in, int32_t* __restrict__ perm_out,
                                                            int64_t n, const int32_t* __restrict__ start, int64_t nblk) {
  __shared__ uint32_t cnt[4][256];     // per wave: elements of each digit seen so far (then: the wave's offset inside the workgroup)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int j = lane; j < 256; j += 64) cnt[w][j] = 0u;
  uint32_t dg[4], rk[4], lead[4], old[4];
  int32_t src[4];
  bool live[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = elem_index(blockIdx.x, w, r, lane);
    live[r] = i < n;
    src[r] = live[r] ? perm_in[i] : 0;
    dg[r] = live[r] ? rows[(int64_t)src[r] * width + byte] : 0x100u;          // a slot past n matches no live digit
    const uint64_t peers = tt::match_any<9>(dg[r]);
    rk[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
    lead[r] = (uint32_t)__ffsll((unsigned long long)peers) - 1u;
    old[r] = (uint32_t)__popcll(peers);
  }
  // the leaders' counter bumps, rounds in element order (one wave's LDS operations execute in order): old = same-digit
  // elements of this wave in earlier rounds
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (live[r] && rk[r] == 0u) old[r] = atomicAdd(&cnt[w][dg[r]], old[r]);
#pragma unroll
  for (int r = 0; r < 4; ++r) old[r] = (uint32_t)__shfl((int)old[r], (int)lead[r]) + rk[r];
  __syncthreads();
  {                                    // digit tid: exclusive scan over the 4 waves
    uint32_t run = 0u;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const uint32_t c = cnt[ww][tid];
      cnt[ww][tid] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (live[r]) perm_out[(int64_t)start[(int64_t)dg[r] * nblk + blockIdx.x] + cnt[w][dg[r]] + old[r]] = src[r];
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

// ---- exclusive scan of int32 (sums < 2^31): workgroup scans, one workgroup over their sums, add-back ----
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t m,
                                                           int32_t* __restrict__ sums) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * SB + 2 * tid;
  const int a = i0 < m ? in[i0] : 0, b = i0 + 1 < m ? in[i0 + 1] : 0;
  int incl = a + b;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
  for (int ww = 0; ww < w; ++ww) base += wsum[ww];
  const int excl = base + incl - (a + b);
  if (i0 < m) out[i0] = excl;
  if (i0 + 1 < m) out[i0 + 1] = excl + a;
  if (tid == 1023) sums[blockIdx.x] = base + incl;
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(int32_t* __restrict__ sums, int64_t nb) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < nb; c0 += 1024) {
    const int64_t i = c0 + tid;
    const int v = i < nb ? sums[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = carry_s;
    for (int ww = 0; ww < w; ++ww) base += wsum[ww];
    if (i < nb) sums[i] = base + incl - v;
    __syncthreads();                               // everyone has read carry_s and wsum
    if (tid == 1023) carry_s = base + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void scan_add_kernel(int32_t* __restrict__ out, int64_t m, const int32_t* __restrict__ sums) {
  const int64_t i0 = (int64_t)blockIdx.x * SB + 2 * threadIdx.x;
  const int add = sums[blockIdx.x];
  if (i0 < m) out[i0] += add;
  if (i0 + 1 < m) out[i0 + 1] += add;
}

int exclusive_scan(const int32_t* in, int32_t* out, int64_t m, int32_t* sums, hipStream_t stream) {
  const int64_t nb = (m + SB - 1) / SB;
  hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)nb), dim3(1024), 0, stream, in, out, m, sums);
  if (nb > 1) {
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, stream, sums, nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(1024), 0, stream, out, m, sums);
  }
  return tt::check_launch("exclusive_scan");
}

__global__ __launch_bounds__(256) void code_scatter_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ excl,
                                                           const int32_t* __restrict__ flag, int64_t n, int64_t* __restrict__ codes,
                                                           int32_t* __restrict__ n_unique) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t rank = excl[i] + flag[i];           // inclusive count of distinct rows up to i
  codes[perm[i]] = (int64_t)rank - 1;
  if (i == n - 1 && n_unique != nullptr) *n_unique = rank;
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct EncWs {
  int64_t nblk, off_perm_a, off_perm_b, off_table, off_start, off_sums, off_flag, off_excl, total;
};

EncWs enc_ws(int64_t n) {
  EncWs w{};
  w.nblk = (n + EB - 1) / EB;
  const int64_t tab = 256 * w.nblk;                                   // histogram table entries
  const int64_t scan_m = tab > n ? tab : n;
  int64_t o = 0;
  w.off_perm_a = o; o = align_up(o + n * 4, 256);
  w.off_perm_b = o; o = align_up(o + n * 4, 256);
  w.off_table = o;  o = align_up(o + tab * 4, 256);
  w.off_start = o;  o = align_up(o + tab * 4, 256);
  w.off_sums = o;   o = align_up(o + ((scan_m + SB - 1) / SB) * 4, 256);
  w.off_flag = o;   o = align_up(o + n * 4, 256);
  w.off_excl = o;   o = align_up(o + n * 4, 256);
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
  int32_t* perm_a = reinterpret_cast<int32_t*>(ws + w.off_perm_a);
  int32_t* perm_b = reinterpret_cast<int32_t*>(ws + w.off_perm_b);
  int32_t* table = reinterpret_cast<int32_t*>(ws + w.off_table);
  int32_t* start = reinterpret_cast<int32_t*>(ws + w.off_start);
  int32_t* sums = reinterpret_cast<int32_t*>(ws + w.off_sums);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + w.off_flag);
  int32_t* excl = reinterpret_cast<int32_t*>(ws + w.off_excl);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  int rc;
  tt::ProfScope prof("encode_ids", stream);
  hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, stream, perm_a, n);
  for (int byte = width - 1; byte >= 0; --byte) {                  // LSD: least significant byte first, every pass stable
    hipLaunchKernelGGL(digit_hist_kernel, dim3((unsigned)w.nblk), dim3(256), 0, stream, rows, width, byte, perm_a, n, table, w.nblk);
    if ((rc = exclusive_scan(table, start, 256 * w.nblk, sums, stream)) != TT_OK) return rc;
    hipLaunchKernelGGL(digit_scatter_kernel, dim3((unsigned)w.nblk), dim3(256), 0, stream, rows, width, byte, perm_a, perm_b, n, start,
                       w.nblk);
    int32_t* t = perm_a; perm_a = perm_b; perm_b = t;
  }
  if ((rc = tt::check_launch("tt_encode_ids_u8(sort)")) != TT_OK) return rc;
  hipLaunchKernelGGL(head_flag_kernel, dim3(blocks), dim3(256), 0, stream, rows, n, width, perm_a, flag);
  if ((rc = exclusive_scan(flag, excl, n, sums, stream)) != TT_OK) return rc;
  hipLaunchKernelGGL(code_scatter_kernel, dim3(blocks), dim3(256), 0, stream, perm_a, excl, flag, n, codes, n_unique);
  return tt::check_launch("tt_encode_ids_u8");
}
