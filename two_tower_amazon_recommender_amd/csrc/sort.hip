// K2 plan — stable sort of (id, position) by id for the fused sparse optimizer.
//
// One launch, one 512/1024-thread workgroup per TABLE (user, item and the hashed-category table of a step sort side by
// side on three CUs): an LSD radix sort that lives entirely in LDS.  Keys are the ids clamped to 32 bits — every id
// outside [0, num_rows) becomes the sentinel 2^bits - 1 >= num_rows, so an out-of-range id can never alias (and cut
// the run of) a valid one; the apply kernel skips the sentinel like any other out-of-range id.  ceil(bits / 9) passes
// of 8- or 9-bit digits (24 bits = 10 M rows: 3 passes; 27 bits = 100 M rows: 3 passes).  Per pass and wave:
//   rank   the 64 lanes of a round find their equal-digit peers with DBITS ballots (match-any), the first peer bumps
//          the wave's digit counter in LDS — rounds in element order, so equal digits keep their order (stable);
//   scan   digit-major, wave-minor exclusive scan of the 16 x 2^DBITS counters;
//   move   keys and positions (u16) scatter inside LDS.
// Integer-only, VALU-bound by the ballots; 8192 ids: 23-26 us on one CU, on a side stream beside the forward pass (rocPRIM's
// device radix sort needs 6 launches / ~65 us for the same job).  Above 16384 ids per table (32-bit keys + 16-bit positions
// + counters no longer fit 160 KiB of LDS) the list is sorted in 16384-id chunks (one workgroup each, same launch) and a
// second launch merges them by rank (merge_rank_kernel); only beyond 16 chunks (262144 ids) rocPRIM takes over.
#include "common.h"
#include "part_sort.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

constexpr int kMaxLdsSortIds = 16384;   // ids one workgroup sorts in LDS
constexpr int kMaxTables = 4;           // tables per tt_sparse_plan_batched call
constexpr int kMaxEntries = 16;         // (table, chunk) entries per launch of the LDS sort
constexpr int kMaxChunks = 16;          // longer lists (> 262144 ids) fall back to rocPRIM

int id_bits(int64_t num_rows) {
  int bits = 1;
  while (bits < 63 && ((int64_t)1 << bits) < num_rows) ++bits;
  return bits;
}

struct SortTable {
  const int64_t* ids;
  int64_t* sorted_ids;
  int32_t* order;
  int64_t num_rows;
  int32_t n;
  int32_t npass;
  uint32_t sentinel;
};
struct SortBatch {
  SortTable t[kMaxEntries];
};

// ---- lists longer than one workgroup's LDS: chunks of 16384 ids are sorted by the kernel below (one workgroup each,
// all in one launch; chunk-local positions), then every element finds its place in the whole by binary search in the
// OTHER chunks: rank = own index + #(keys <= mine) in earlier chunks + #(keys < mine) in later ones — earlier chunks
// hold the smaller positions, so this is exactly the stable order.  n threads, <= 14 dependent L2 loads per other chunk.
struct MergeArgs {
  const int64_t* keys;    // [n]  chunk-wise sorted (clamped) keys
  const int32_t* lpos;    // [n]  chunk-local positions
  int64_t* sorted_ids;    // [n]  out
  int32_t* order;         // [n]  out
  int32_t n, nchunks;
};

__global__ __launch_bounds__(256) void merge_rank_kernel(MergeArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.n) return;
  const int c = e / kMaxLdsSortIds;
  const int64_t key = a.keys[e];
  int rank = e - c * kMaxLdsSortIds;
  for (int o = 0; o < a.nchunks; ++o) {
    if (o == c) continue;
    const int64_t* run = a.keys + (int64_t)o * kMaxLdsSortIds;
    int lo = 0, hi = a.n - o * kMaxLdsSortIds;
    if (hi > kMaxLdsSortIds) hi = kMaxLdsSortIds;
    while (lo < hi) {                     // o < c: upper bound (equal keys of earlier chunks go first); o > c: lower bound
      const int mid = (lo + hi) >> 1;
      const int64_t v = run[mid];
      if (o < c ? v <= key : v < key) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  a.sorted_ids[rank] = key;
  a.order[rank] = c * kMaxLdsSortIds + a.lpos[e];
}

// element i of the workgroup's sequence sits in (wave, round, lane) = (i / (64*ITEMS), (i / 64) % ITEMS, i % 64)
template <int ITEMS, int DBITS>
__global__ __launch_bounds__(1024) void lds_sort_kernel(SortBatch batch) {
  constexpr int RADIX = 1 << DBITS;
  constexpr int MAXW = 16;
  const SortTable t = batch.t[blockIdx.x];
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int T = blockDim.x, W = T >> 6;
  const int cap = T * ITEMS;
  uint32_t* keys = smem;                                   // [cap]
  uint32_t* cnt = keys + cap;                              // [W][RADIX]
  uint32_t* wtot = cnt + W * RADIX;                        // [16]
  uint16_t* poss = reinterpret_cast<uint16_t*>(wtot + 16); // [cap]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = t.n;
  uint32_t* mycnt = cnt + w * RADIX;
  uint32_t key[ITEMS], pos[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = (w * ITEMS + r) * 64 + lane;
    pos[r] = (uint32_t)i;
    const int64_t id = t.ids[i < n ? i : 0];               // unconditional: all ITEMS loads in flight together
    // slots past n get the sentinel too and sort behind everything (stable: their pos is >= n)
    key[r] = (i < n && id >= 0 && id < t.num_rows) ? (uint32_t)id : t.sentinel;
  }

  for (int p = 0; p < t.npass; ++p) {
    const int shift = p * DBITS;
    for (int j = lane; j < RADIX; j += 64) mycnt[j] = 0u;
    // ---- rank inside the wave.  All ballots first (VALU/SALU only), then the leaders' counter bumps as LDS atomics
    // issued back to back: the LDS executes one wave's instructions in order, so round r sees the bumps of rounds < r
    // without a round trip per round; the old value reaches the other peers through one ds_bpermute each. ----
    // SLIM (ITEMS > 8, the 16384-id chunks): one element at a time from digit to slot - the LDS atomic's round trip is paid per
    // element instead of once per pass and the digit is recomputed where it is needed - so that only key / position / slot are
    // live across the pass: with the three-loop form this instantiation spilled 56 B per lane inside the 128-VGPR budget of a
    // 1024-thread workgroup (r03 did the same to the hot-range sort of csrc/part_sort.h).  Same ranks, same result.
    constexpr bool SLIM = ITEMS > 8;
    auto digit = [&](int r) -> uint32_t { return (key[r] >> shift) & (RADIX - 1); };
    uint32_t dg[SLIM ? 1 : ITEMS], rk[SLIM ? 1 : ITEMS], lead[SLIM ? 1 : ITEMS], old[ITEMS];
    if constexpr (SLIM) {
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const uint32_t d = digit(r);
        const uint64_t peers = tt::match_any<DBITS>(d);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        uint32_t o = (uint32_t)__popcll(peers);                // the leader's increment
        if (rank == 0u) o = atomicAdd(&mycnt[d], o);
        old[r] = (uint32_t)__shfl((int)o, (int)((uint32_t)__ffsll((unsigned long long)peers) - 1u)) + rank;   // rank inside this wave
      }
    } else {
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        dg[r] = digit(r);
        const uint64_t peers = tt::match_any<DBITS>(dg[r]);
        rk[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        lead[r] = (uint32_t)__ffsll((unsigned long long)peers) - 1u;
        old[r] = (uint32_t)__popcll(peers);                  // the leader's increment
      }
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        if (rk[r] == 0u) old[r] = atomicAdd(&mycnt[dg[r]], old[r]);
      }
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) old[r] = (uint32_t)__shfl((int)old[r], (int)lead[r]) + rk[r];   // rank inside this wave
    }
    auto bucket = [&](int r) -> uint32_t { if constexpr (SLIM) return mycnt[digit(r)]; else return mycnt[dg[r]]; };
    __syncthreads();
    // ---- exclusive scan, digit-major / wave-minor; the base of the digit is folded into the per-wave offsets ----
    uint32_t v[MAXW], total = 0u;
    if (tid < RADIX) {
#pragma unroll
      for (int ww = 0; ww < MAXW; ++ww) v[ww] = ww < W ? cnt[ww * RADIX + tid] : 0u;
#pragma unroll
      for (int ww = 0; ww < MAXW; ++ww) {
        const uint32_t x = v[ww];
        v[ww] = total;
        total += x;
      }
    }
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    if (tid < RADIX) {
      uint32_t base = incl - total;
      for (int ww = 0; ww < w; ++ww) base += wtot[ww];
#pragma unroll
      for (int ww = 0; ww < MAXW; ++ww)
        if (ww < W) cnt[ww * RADIX + tid] = base + v[ww];
    }
    __syncthreads();
    // ---- move ----
    if (p + 1 < t.npass) {
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const uint32_t dst = bucket(r) + old[r];
        keys[dst] = key[r];
        poss[dst] = (uint16_t)pos[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const int i = (w * ITEMS + r) * 64 + lane;
        key[r] = keys[i];
        pos[r] = poss[i];
      }
    } else {
      // last pass: straight to global (8-byte / 4-byte scattered stores; n <= 16384 of them)
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const uint32_t dst = bucket(r) + old[r];
        if (dst < (uint32_t)n) {
          t.sorted_ids[dst] = (int64_t)key[r];
          t.order[dst] = (int32_t)pos[r];
        }
      }
    }
  }
}

// ---- partitioned sort (csrc/part_sort.h): G workgroups per table, each sorting the ids of its own row range ----
struct PartBatch {
  tt::PartTable t[kMaxTables];
  int32_t cap;   // LDS capacity in ids: the longest list of the launch, rounded up to 1024
};

template <int DBITS, int JMAX>     // JMAX: ids per thread of the scan (8: lists <= 8192, 16: <= 16384)
__global__ __launch_bounds__(1024) void part_sort_kernel(PartBatch batch) {
  const tt::PartTable t = batch.t[blockIdx.y];
  if ((int)blockIdx.x >= t.groups) return;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  uint32_t offset, base_key;
  (void)tt::part_sort_body<DBITS, JMAX, true>(t, (int)blockIdx.x, batch.cap, smem, offset, base_key);
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct ClampKey {
  int64_t rows, sentinel;
  __host__ __device__ int64_t operator()(int64_t id) const { return (id >= 0 && id < rows) ? id : sentinel; }
};
using ClampIt = rocprim::transform_iterator<const int64_t*, ClampKey, int64_t>;

size_t rocprim_temp_bytes(int64_t n_ids) {
  size_t bytes = 0;
  int64_t* kout = nullptr;
  int32_t* vout = nullptr;
  ClampIt kin(nullptr, ClampKey{1, 1});
  (void)rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, rocprim::counting_iterator<int32_t>(0), vout, (size_t)n_ids, 0u, 64u,
                                  (hipStream_t) nullptr, false);
  return bytes;
}

template <int ITEMS, int DBITS>
int launch_lds_sort(const SortBatch& b, int n_tables, int threads, hipStream_t stream) {
  const int W = threads / 64, cap = threads * ITEMS, radix = 1 << DBITS;
  const int lds = (cap + W * radix + 16) * 4 + cap * 2;
  auto kern = lds_sort_kernel<ITEMS, DBITS>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return tt::fail(TT_ERR_LAUNCH, "tt_sparse_plan: hipFuncSetAttribute(LDS %d) failed", lds);
  tt::launch("sparse_plan", kern, dim3((unsigned)n_tables), dim3((unsigned)threads), lds, stream, b);
  return tt::check_launch("tt_sparse_plan");
}

int plan_rocprim(const tt_sparse_plan_args& a, hipStream_t stream) {
  const size_t temp = rocprim_temp_bytes(a.n_ids);
  if ((int64_t)temp > a.workspace_bytes)
    return tt::fail(TT_ERR_WORKSPACE, "tt_sparse_plan: workspace %lld < %lld bytes", (long long)a.workspace_bytes, (long long)temp);
  TT_REQUIRE(a.workspace != nullptr, "tt_sparse_plan: null workspace");
  const int bits = id_bits(a.num_rows + 1);
  ClampIt kin(a.ids, ClampKey{a.num_rows, ((int64_t)1 << bits) - 1});
  size_t tb = temp;
  tt::ProfScope prof("sparse_plan", stream);
  hipError_t e = rocprim::radix_sort_pairs(a.workspace, tb, kin, a.sorted_ids, rocprim::counting_iterator<int32_t>(0), a.order,
                                           (size_t)a.n_ids, 0u, (unsigned)bits, stream, false);
  if (e != hipSuccess) return tt::fail(TT_ERR_LAUNCH, "tt_sparse_plan: rocprim radix sort: %s", hipGetErrorString(e));
  return TT_OK;
}

}  // namespace

#ifdef TT_SORT_STAMPS
extern "C" int tt_debug_sort_stamps(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tt::g_sort_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 2;
}
#endif

extern "C" int32_t tt_sparse_plan_max_lds_ids(void) { return kMaxLdsSortIds; }

extern "C" int64_t tt_sparse_plan_workspace_bytes(int64_t n_ids) {
  if (n_ids <= kMaxLdsSortIds) return 256;         // the LDS sort needs no global scratch
  if (n_ids <= (int64_t)kMaxChunks * kMaxLdsSortIds) return align_up(n_ids * 8, 256) + align_up(n_ids * 4, 256);   // chunk-sorted keys + positions
  return align_up((int64_t)rocprim_temp_bytes(n_ids), 256) + 256;
}

namespace {
// TT_SORT_GROUPS: 0 = the single-workgroup sort for every list (A/B), k > 0 = force k partitions, unset = n / 128
// (8192 ids -> 64 workgroups per table).  Measured, 2 tables x 8192 ids: 21-25 us single-workgroup -> 9.5-10 us (of which
// ~9 us is the host call rate of the microbench), Zipf ids 12-15 us; in the cfg3 step (plan on the main stream) 16 / 32 /
// 64 / 128 groups: 0.6787 / 0.6780 / 0.6766 / 0.6773 ms.
int env_sort_groups() {
  static const int v = [] {
    const char* e = std::getenv("TT_SORT_GROUPS");
    return e ? std::atoi(e) : -1;
  }();
  return v;
}

int part_groups(int n, int64_t num_rows) {
  const int forced = env_sort_groups();
  int64_t g = forced > 0 ? forced : (n + 127) / 128;
  const int64_t cap = forced > 0 ? 256 : 128;
  if (g > cap) g = cap;
  if (g > num_rows / 2) g = num_rows / 2;          // a group is at least 2 ids wide (the multiply-high division needs width >= 2)
  return g < 1 ? 1 : (int)g;
}

int launch_part(PartBatch& b, int nb, int max_n, int max_groups, int max_lbits, hipStream_t stream) {
  // digits (hot ranges only): 8 bits when the passes needed are the same as with 9 (fewer ballots, smaller counter table)
  const int npass9 = (max_lbits + 8) / 9, npass8 = (max_lbits + 7) / 8;
  const bool nine = npass9 < npass8;
  b.cap = (max_n + 1023) / 1024 * 1024;
  const int radix = nine ? 512 : 256;
  const int lds = tt::part_sort_lds_bytes(b.cap, radix);
  const bool small = max_n <= 8 * 1024;
  auto kern = small ? (nine ? part_sort_kernel<9, 8> : part_sort_kernel<8, 8>) : (nine ? part_sort_kernel<9, 16> : part_sort_kernel<8, 16>);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return tt::fail(TT_ERR_LAUNCH, "tt_sparse_plan: hipFuncSetAttribute(LDS %d) failed", lds);
  tt::launch("sparse_plan", kern, dim3((unsigned)max_groups, (unsigned)nb), dim3(1024), lds, stream, b);
  return tt::check_launch("tt_sparse_plan(partitioned)");
}

int launch_entries(SortBatch& b, int nb, int max_n, int max_bits, hipStream_t stream) {
  // digits: 8 bits when the passes needed are the same as with 9 (fewer ballots, smaller counter table)
  const int npass9 = (max_bits + 8) / 9, npass8 = (max_bits + 7) / 8;
  const bool nine = npass9 < npass8;
  for (int i = 0; i < nb; ++i) {
    const int bits = id_bits(b.t[i].num_rows + 1);
    b.t[i].npass = nine ? (bits + 8) / 9 : (bits + 7) / 8;
  }
  const int items = max_n > 8192 ? 16 : 8;
  const int threads = max_n <= 512 * items ? 512 : 1024;
  if (items == 8) return nine ? launch_lds_sort<8, 9>(b, nb, threads, stream) : launch_lds_sort<8, 8>(b, nb, threads, stream);
  return nine ? launch_lds_sort<16, 9>(b, nb, threads, stream) : launch_lds_sort<16, 8>(b, nb, threads, stream);
}
}  // namespace

extern "C" int tt_sparse_plan_batched(const tt_sparse_plan_args* tables, int32_t n_tables, tt_stream_t stream_) {
  TT_REQUIRE(tables != nullptr && n_tables >= 1 && n_tables <= kMaxTables, "tt_sparse_plan_batched: 1..%d tables", kMaxTables);
  hipStream_t stream = tt::as_stream(stream_);
  SortBatch b{};
  PartBatch pb{};
  MergeArgs merges[kMaxTables];
  int nb = 0, max_n = 0, max_bits = 0, n_merge = 0, rc;
  int npb = 0, p_max_n = 0, p_max_groups = 0, p_max_lbits = 0;
  for (int i = 0; i < n_tables; ++i) {
    const tt_sparse_plan_args& a = tables[i];
    TT_REQUIRE(a.n_ids >= 0 && a.num_rows > 0, "tt_sparse_plan: bad n_ids/num_rows");
    TT_REQUIRE(a.n_ids <= 0x7fffffff, "tt_sparse_plan: n_ids must fit in int32");
    if (a.n_ids == 0) continue;
    TT_REQUIRE(a.ids && a.sorted_ids && a.order, "tt_sparse_plan: null pointer");
    const int bits = id_bits(a.num_rows + 1);
    const int chunks = (int)((a.n_ids + kMaxLdsSortIds - 1) / kMaxLdsSortIds);
    if (chunks > kMaxChunks || bits > 31) {   // very long id lists: rocPRIM, one table at a time
      if ((rc = plan_rocprim(a, stream)) != TT_OK) return rc;
      continue;
    }
    if (chunks == 1 && env_sort_groups() != 0) {   // one list that fits a workgroup's LDS: key-range partitions, one launch
      tt::PartTable& t = pb.t[npb++];
      t.ids = a.ids; t.sorted_ids = a.sorted_ids; t.order = a.order; t.num_rows = a.num_rows; t.n = (int32_t)a.n_ids;
      t.groups = part_groups(t.n, a.num_rows);
      t.width = (uint32_t)((a.num_rows + t.groups - 1) / t.groups);
      if (t.width < 2u) t.width = 2u;
      t.magic = (uint32_t)((((uint64_t)1 << 32) / t.width) + 1u);
      t.sentinel = (uint32_t)(((uint64_t)1 << bits) - 1);
      const int lbits = id_bits((int64_t)t.width);
      if (t.n > p_max_n) p_max_n = t.n;
      if (t.groups > p_max_groups) p_max_groups = t.groups;
      if (lbits > p_max_lbits) p_max_lbits = lbits;
      continue;
    }
    int64_t* keys_out = a.sorted_ids;
    int32_t* pos_out = a.order;
    if (chunks > 1) {                         // chunk-wise sorted (key, local position) go to the workspace, then the merge
      const int64_t need = tt_sparse_plan_workspace_bytes(a.n_ids);
      if (a.workspace == nullptr || a.workspace_bytes < need)
        return tt::fail(TT_ERR_WORKSPACE, "tt_sparse_plan: workspace %lld < %lld bytes", (long long)a.workspace_bytes, (long long)need);
      TT_REQUIRE((reinterpret_cast<uintptr_t>(a.workspace) & 255u) == 0, "tt_sparse_plan: workspace must be 256-byte aligned");
      keys_out = static_cast<int64_t*>(a.workspace);
      pos_out = reinterpret_cast<int32_t*>(static_cast<char*>(a.workspace) + align_up(a.n_ids * 8, 256));
      merges[n_merge++] = MergeArgs{keys_out, pos_out, a.sorted_ids, a.order, (int32_t)a.n_ids, chunks};
    }
    for (int c = 0; c < chunks; ++c) {
      if (nb == kMaxEntries) {                // (only with several multi-chunk tables in one call)
        if ((rc = launch_entries(b, nb, max_n, max_bits, stream)) != TT_OK) return rc;
        nb = 0; max_n = 0; max_bits = 0;
      }
      const int64_t off = (int64_t)c * kMaxLdsSortIds;
      SortTable& t = b.t[nb++];
      t.ids = a.ids + off; t.sorted_ids = keys_out + off; t.order = pos_out + off; t.num_rows = a.num_rows;
      t.n = (int32_t)(a.n_ids - off < kMaxLdsSortIds ? a.n_ids - off : kMaxLdsSortIds);
      t.sentinel = (uint32_t)(((uint64_t)1 << bits) - 1);
      if (t.n > max_n) max_n = t.n;
      if (bits > max_bits) max_bits = bits;
    }
  }
  if (npb > 0 && (rc = launch_part(pb, npb, p_max_n, p_max_groups, p_max_lbits, stream)) != TT_OK) return rc;
  if (nb > 0 && (rc = launch_entries(b, nb, max_n, max_bits, stream)) != TT_OK) return rc;
  for (int i = 0; i < n_merge; ++i) {
    tt::ProfScope prof("sparse_plan", stream);
    hipLaunchKernelGGL(merge_rank_kernel, dim3((unsigned)((merges[i].n + 255) / 256)), dim3(256), 0, stream, merges[i]);
    if ((rc = tt::check_launch("tt_sparse_plan(merge)")) != TT_OK) return rc;
  }
  return TT_OK;
}

extern "C" int tt_sparse_plan(const int64_t* ids, int64_t n_ids, int64_t num_rows, void* workspace, int64_t workspace_bytes,
                              int64_t* sorted_ids, int32_t* order, tt_stream_t stream) {
  const tt_sparse_plan_args a{ids, n_ids, num_rows, workspace, workspace_bytes, sorted_ids, order};
  return tt_sparse_plan_batched(&a, 1, stream);
}
