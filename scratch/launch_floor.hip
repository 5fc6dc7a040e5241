// r04 probe: what does the optimizer launch cost when everything but the unavoidable memory traffic is taken
// away?  "Floor" kernel: every workgroup gets its ~n/G (row id, batch position) pairs PRE-BUCKETED in HBM (no sort, no scan, no
// duplicates), reads gradient row + table row, writes the SGD-updated row; beside them the dense blocks sum 32 gradient slabs into
// the tower parameters (18 MB) - the same 25 + 18 MB as optimizer_ids_kernel at cfg3.  Also: empty kernels of the same launch shapes
// (dispatch + end-of-kernel cost alone).  Durations from the dispatch's own timestamps (hipExtLaunchKernelGGL event pair).
//   hipcc -O3 --offload-arch=gfx950 -o scratch/variants/launch_floor scratch/launch_floor.hip && scratch/variants/launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#include <functional>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Floor {
  const uint2* pairs[2];      // [groups][cap] (row, position)
  const int* counts[2];       // [groups]
  f32x4* table[2];
  const f32x4* grads[2];
  int groups, cap;
  // dense: 32 slabs of n4 float4
  const f32x4* slabs; f32x4* param; int n4, dense_blocks;
  float lr;
};

template <int THREADS, int RA, bool NT>
__global__ __launch_bounds__(THREADS) void floor_kernel(Floor a) {
  extern __shared__ uint32_t smem[];
  const int b = (int)blockIdx.x - a.dense_blocks;
  if (b < 0) {
    // dense blocks: 16 slabs + param in flight, as dense_update_body
    const int d = blockIdx.x;
    for (int i = d * THREADS + threadIdx.x; i < a.n4; i += a.dense_blocks * THREADS) {
      f32x4 s[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) s[k] = a.slabs[(size_t)k * a.n4 + i];
      f32x4 w = a.param[i];
      f32x4 g = s[0];
#pragma unroll
      for (int k = 1; k < 32; ++k) g += s[k];
      a.param[i] = w - a.lr * g;
    }
    return;
  }
  const int t = b / a.groups, g = b - t * a.groups;
  const int n = a.counts[t][g];
  const uint2* pr = a.pairs[t] + (size_t)g * a.cap;
  const int grp = threadIdx.x >> 5, l = threadIdx.x & 31, ngroups = THREADS / 32;
  f32x4* table = a.table[t];
  const f32x4* grads = a.grads[t];
  for (int i0 = 0; i0 < n; i0 += ngroups * RA) {
    uint2 p[RA];
    f32x4 gr[RA], w[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) { const int i = i0 + grp + r * ngroups; p[r] = pr[i < n ? i : 0]; }
#pragma unroll
    for (int r = 0; r < RA; ++r) { gr[r] = grads[(size_t)p[r].y * 32 + l]; w[r] = table[(size_t)p[r].x * 32 + l]; }
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int i = i0 + grp + r * ngroups;
      if (i < n) {
        const f32x4 v = w[r] - a.lr * gr[r];
        if (NT) __builtin_nontemporal_store(v, table + (size_t)p[r].x * 32 + l); else table[(size_t)p[r].x * 32 + l] = v;
      }
    }
  }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void empty_kernel(int* p) { if (p != nullptr && threadIdx.x == 4096) *p = 1; }

static float time_launch(hipStream_t s, std::function<void(hipEvent_t, hipEvent_t)> launch, int iters = 30) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> v;
  static char* flush = nullptr;
  static const bool do_flush = std::getenv("FLUSH") != nullptr;
  if (do_flush && flush == nullptr) CK(hipMalloc(&flush, (size_t)1 << 30));
  for (int i = 0; i < iters + 5; ++i) {
    if (do_flush) CK(hipMemsetAsync(flush, i, (size_t)1 << 30, s));      // 1 GB through L2 / MALL: the rows come from HBM again
    launch(e0, e1);
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 5) v.push_back(ms * 1e3f);
  }
  std::sort(v.begin(), v.end());
  float mean = 0; for (float x : v) mean += x; mean /= v.size();
  printf("  mean %.2f us  median %.2f  min %.2f  max %.2f\n", mean, v[v.size() / 2], v.front(), v.back());
  return mean;
}

#include <functional>

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int B = 8192, D4 = 32;
  const int64_t rows[2] = {5000000, 10000000};
  Floor a{};
  f32x4* tab[2]; f32x4* gr[2];
  std::mt19937_64 rng(7);
  for (int t = 0; t < 2; ++t) {
    CK(hipMalloc(&tab[t], (size_t)rows[t] * 512)); CK(hipMemsetAsync(tab[t], 0, (size_t)rows[t] * 512, s));
    CK(hipMalloc(&gr[t], (size_t)B * 512)); CK(hipMemsetAsync(gr[t], 0, (size_t)B * 512, s));
    a.table[t] = tab[t]; a.grads[t] = gr[t];
  }
  const int n4 = (2 * (128 * 256 + 256 * 128 + 256 + 128)) / 4;
  f32x4 *slabs, *param;
  CK(hipMalloc(&slabs, (size_t)32 * n4 * 16)); CK(hipMemsetAsync(slabs, 0, (size_t)32 * n4 * 16, s));
  CK(hipMalloc(&param, (size_t)n4 * 16)); CK(hipMemsetAsync(param, 0, (size_t)n4 * 16, s));
  a.slabs = slabs; a.param = param; a.n4 = n4; a.lr = 0.001f;
  auto setup = [&](int groups, int dense_blocks) {
    a.groups = groups; a.cap = 256; a.dense_blocks = dense_blocks;
    for (int t = 0; t < 2; ++t) {
      std::vector<uint2> pairs((size_t)groups * 256);
      std::vector<int> counts(groups, 0);
      // uniform ids -> the row range that holds them (sorted by range, as the forward lookup's lists are)
      const uint32_t width = (uint32_t)((rows[t] + groups - 1) / groups);
      for (int p = 0; p < B; ++p) {
        const uint32_t id = (uint32_t)(rng() % (uint64_t)rows[t]);
        const int g = id / width;
        if (counts[g] < 256) pairs[(size_t)g * 256 + counts[g]++] = make_uint2(id, (uint32_t)p);
      }
      uint2* dp; int* dc;
      CK(hipMalloc(&dp, pairs.size() * 8)); CK(hipMalloc(&dc, groups * 4));
      CK(hipMemcpy(dp, pairs.data(), pairs.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dc, counts.data(), groups * 4, hipMemcpyHostToDevice));
      a.pairs[t] = dp; a.counts[t] = dc;
    }
  };
  CK(hipStreamSynchronize(s));

  printf("empty kernel, grid 256 x 1024 threads, 67 KB LDS:\n");
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 68608));
  time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL(empty_kernel<1024>, dim3(256), dim3(1024), 68608, s, e0, e1, 0, (int*)nullptr); });
  printf("empty kernel, grid 256 x 1024 threads, no LDS:\n");
  time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL(empty_kernel<1024>, dim3(256), dim3(1024), 0, s, e0, e1, 0, (int*)nullptr); });
  printf("empty kernel, grid 256 x 256 threads:\n");
  time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL(empty_kernel<256>, dim3(256), dim3(256), 0, s, e0, e1, 0, (int*)nullptr); });
  printf("empty kernel, grid 1 x 64 threads:\n");
  time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL(empty_kernel<256>, dim3(1), dim3(64), 0, s, e0, e1, 0, (int*)nullptr); });

  auto run = [&](const char* name, auto kern, int threads, int groups, int dense_blocks, int lds) {
    setup(groups, dense_blocks);
    printf("%s: %d + 2 x %d workgroups of %d threads, LDS %d:\n", name, dense_blocks, groups, threads, lds);
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL(kern, dim3(dense_blocks + 2 * groups), dim3(threads), lds, s, e0, e1, 0, a); });
  };
  run("floor 1024 thr, 4 rows ahead, nt stores", floor_kernel<1024, 4, true>, 1024, 110, 36, 68608);
  run("floor 1024 thr, 4 rows ahead, nt stores, no LDS", floor_kernel<1024, 4, true>, 1024, 110, 36, 0);
  run("floor 1024 thr, 4 rows ahead, plain stores", floor_kernel<1024, 4, false>, 1024, 110, 36, 0);
  run("floor 512 thr, 4 rows ahead, nt, 2 WG/CU", floor_kernel<512, 4, true>, 512, 220, 72, 0);
  run("floor 256 thr, 4 rows ahead, nt, 4 WG/CU", floor_kernel<256, 4, true>, 256, 440, 144, 0);
  run("floor 256 thr, 2 rows ahead, nt, 8 WG/CU", floor_kernel<256, 2, true>, 256, 880, 144, 0);
  run("floor 1024 thr, no dense blocks", floor_kernel<1024, 4, true>, 1024, 128, 0, 0);
  run("floor 256 thr, no dense blocks, 4 WG/CU", floor_kernel<256, 4, true>, 256, 512, 0, 0);
  a.n4 = n4;
  {
    // dense only
    setup(1, 36);
    Floor d = a; d.groups = 0;
    printf("dense blocks only (36 x 1024):\n");
    time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL((floor_kernel<1024, 4, true>), dim3(36), dim3(1024), 0, s, e0, e1, 0, d); });
    printf("dense blocks only (144 x 256):\n");
    d.dense_blocks = 144;
    time_launch(s, [&](hipEvent_t e0, hipEvent_t e1) { hipExtLaunchKernelGGL((floor_kernel<256, 4, true>), dim3(144), dim3(256), 0, s, e0, e1, 0, d); });
  }
  return 0;
}
