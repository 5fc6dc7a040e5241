// Requester-side routing of the row-sharded embedding exchange.
//
// route_kernel   : stable partition of a batch of ids by owner rank (owner = id % world, local row = id / world
//                  + the table's offset inside the owner's combined shard) into fixed-capacity per-owner send
//                  buffers (padding id -1) + the flat slot of every position; one workgroup per table.
//                  16 waves, 512 consecutive positions per wave read as 8 coalesced loads; per 8192-position round:
//                  `world` x 8 ballots give every position its rank inside the wave, a 16 x world table in LDS the
//                  offsets across waves (ascending position order inside every owner bucket: deterministic).
//                  Integer/byte work, latency-bound (a few microseconds).
// scatter_rows   : dst[idx[p], :] = src[p, :] (idx < 0 skipped) — per-position gradient rows into the send buffer;
//                  HBM-bound, dim/4 lanes per row like the gather.
#include "common.h"

namespace {

constexpr int kMaxWorld = 16;

struct RouteTables {
  tt_route_table t[TT_ROUTE_MAX_TABLES];
};

// One workgroup per table (blockIdx.x = table t); bucket (owner o, table t) of the send buffer is
// send_ids[(o*n_tables + t)*cap ...]: one all-to-all moves every table's ids (and later rows) at once.
// A wave owns kIpt * 64 CONSECUTIVE positions per round and reads them 64 at a time (load j = positions 64 j .. 64 j + 63 of
// its span: one coalesced 512-byte request), so "ascending position" = (wave, j, lane) order: per owner, the ballot of a load
// gives every lane its rank among the wave's earlier positions, the running popcounts the load's base, and a 16 x world table
// in LDS the offsets across waves; a batch of 8192 ids is one round (two barriers).
// (Through r02 a THREAD owned 8 consecutive positions: every load and store instruction then touched 64 different cache
// lines - 24 such stores and 8 loads per wave and round, ~10 us of line requests on the one CU that runs a table; r03.)
constexpr int kIpt = 8;

// WORLD is a template parameter: id % WORLD and id / WORLD on int64 are multiply-shift sequences for a constant
// divisor, but a ~100-instruction software division for a runtime one (that alone made the kernel 15 us).
template <int WORLD>
__global__ __launch_bounds__(1024) void route_kernel(RouteTables tabs, int n_tables, int64_t n, int cap,
                                                     int64_t* __restrict__ send_ids, int32_t* __restrict__ flags) {
  constexpr int world = WORLD;
  __shared__ int wave_cnt[16][kMaxWorld];
  __shared__ int wave_off[16][kMaxWorld];
  __shared__ int running[kMaxWorld];
  const int t = blockIdx.x;
  const int64_t* __restrict__ ids = tabs.t[t].ids;
  int64_t* __restrict__ pos_flat = tabs.t[t].pos_flat;
  const int64_t num_rows = tabs.t[t].num_rows, local_offset = tabs.t[t].local_offset;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int o = 0; o < world; ++o)
    for (int i = tid; i < cap; i += 1024) send_ids[((int64_t)o * n_tables + t) * cap + i] = -1;
  if (tid < kMaxWorld) running[tid] = 0;
  __syncthreads();
  bool oob = false, over = false;
  const int64_t chunk = 1024 * kIpt;
  const int64_t rounds = (n + chunk - 1) / chunk;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t base = r * chunk + (int64_t)wave * (64 * kIpt) + lane;      // this lane's position of load 0
    int64_t id[kIpt];
    int owner[kIpt], loc[kIpt];
    // all eight ids requested before any is looked at, unconditionally from clamped positions
#pragma unroll
    for (int j = 0; j < kIpt; ++j) {
      const int64_t p = base + 64 * j;
      id[j] = ids[p < n ? p : n - 1];
    }
#pragma unroll
    for (int j = 0; j < kIpt; ++j) {
      const int64_t p = base + 64 * j;
      const bool valid = p < n;
      const bool bad = valid && (id[j] < 0 || id[j] >= num_rows);
      oob = oob || bad;
      owner[j] = (valid && !bad) ? (int)((uint64_t)id[j] % (uint64_t)world) : -1;
      loc[j] = 0;
    }
    for (int o = 0; o < world; ++o) {
      int c = 0;                                                               // (wave-uniform) positions of owner o so far
#pragma unroll
      for (int j = 0; j < kIpt; ++j) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(owner[j] == o);
        if (owner[j] == o) loc[j] = c + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        c += (int)__popcll(m);
      }
      if (lane == 0) wave_cnt[wave][o] = c;
    }
    __syncthreads();
    if (tid < world) {
      int run = running[tid];
      for (int w = 0; w < 16; ++w) {
        wave_off[w][tid] = run;
        run += wave_cnt[w][tid];
      }
      running[tid] = run;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIpt; ++j) {
      const int64_t p = base + 64 * j;
      if (p < n) {
        const int o = owner[j];
        int64_t flat = -1;
        if (o >= 0) {
          const int slot = wave_off[wave][o] + loc[j];
          if (slot < cap) {
            flat = ((int64_t)o * n_tables + t) * cap + slot;
            send_ids[flat] = (int64_t)((uint64_t)id[j] / (uint64_t)world) + local_offset;
          } else {
            over = true;
          }
        }
        pos_flat[p] = flat;
      }
    }
    __syncthreads();                        // wave_cnt / wave_off are rewritten by the next round
  }
  if (flags != nullptr) {
    if (oob) atomicOr(&flags[0], 1);
    if (over) atomicOr(&flags[1], 1);
  }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const tt::f32x4* __restrict__ src, const int64_t* __restrict__ idx,
                                                           int64_t n, int dim4, int lpr_log2, tt::f32x4* __restrict__ dst,
                                                           int64_t dst_rows) {
  const int lpr = 1 << lpr_log2;
  const int groups = 256 >> lpr_log2;
  const int64_t p = (int64_t)blockIdx.x * groups + (threadIdx.x >> lpr_log2);
  const int l = threadIdx.x & (lpr - 1);
  if (p >= n) return;
  const int64_t d = idx[p];
  if (d < 0 || d >= dst_rows) return;
  for (int c = l; c < dim4; c += lpr) dst[d * dim4 + c] = src[p * dim4 + c];
}

}  // namespace

extern "C" int tt_route_tables_by_owner_i64(const tt_route_table* tables, int32_t n_tables, int64_t n_ids, int32_t world,
                                            int32_t cap, int64_t* send_ids, int32_t* flags, tt_stream_t stream_) {
  TT_REQUIRE(tables != nullptr && n_tables >= 1 && n_tables <= TT_ROUTE_MAX_TABLES,
             "tt_route_tables_by_owner_i64: 1 <= n_tables <= %d", TT_ROUTE_MAX_TABLES);
  TT_REQUIRE(n_ids >= 0 && world >= 1 && world <= kMaxWorld && cap >= 1,
             "tt_route_tables_by_owner_i64: need n_ids >= 0, 1 <= world <= %d, cap >= 1", kMaxWorld);
  TT_REQUIRE(send_ids != nullptr, "tt_route_tables_by_owner_i64: null pointer");
  RouteTables tabs = {};
  for (int t = 0; t < n_tables; ++t) {
    TT_REQUIRE(tables[t].num_rows > 0 && tables[t].local_offset >= 0, "tt_route_tables_by_owner_i64: table %d: num_rows > 0, local_offset >= 0", t);
    TT_REQUIRE(n_ids == 0 || (tables[t].ids != nullptr && tables[t].pos_flat != nullptr), "tt_route_tables_by_owner_i64: table %d: null pointer", t);
    tabs.t[t] = tables[t];
  }
  hipStream_t stream = tt::as_stream(stream_);
  switch (world) {
#define TT_ROUTE_CASE(W)                                                                                                   \
  case W:                                                                                                                  \
    tt::launch("route", route_kernel<W>, dim3(n_tables), dim3(1024), 0, stream, tabs, n_tables, n_ids, cap, send_ids, flags); \
    break;
    TT_ROUTE_CASE(1) TT_ROUTE_CASE(2) TT_ROUTE_CASE(3) TT_ROUTE_CASE(4) TT_ROUTE_CASE(5) TT_ROUTE_CASE(6) TT_ROUTE_CASE(7)
    TT_ROUTE_CASE(8) TT_ROUTE_CASE(9) TT_ROUTE_CASE(10) TT_ROUTE_CASE(11) TT_ROUTE_CASE(12) TT_ROUTE_CASE(13)
    TT_ROUTE_CASE(14) TT_ROUTE_CASE(15) TT_ROUTE_CASE(16)
#undef TT_ROUTE_CASE
  }
  return tt::check_launch("tt_route_tables_by_owner_i64");
}

extern "C" int tt_route_by_owner_i64(const int64_t* ids, int64_t n_ids, int32_t world, int64_t num_rows, int32_t cap,
                                     int64_t* send_ids, int64_t* pos_flat, int32_t* flags, tt_stream_t stream_) {
  TT_REQUIRE(num_rows > 0, "tt_route_by_owner_i64: num_rows > 0");
  const tt_route_table tab{ids, num_rows, 0, pos_flat};
  return tt_route_tables_by_owner_i64(&tab, 1, n_ids, world, cap, send_ids, flags, stream_);
}

extern "C" int tt_scatter_rows_f32(const float* src, const int64_t* idx, int64_t n, int32_t dim, float* dst, int64_t dst_rows,
                                   tt_stream_t stream_) {
  TT_REQUIRE(n >= 0 && dim > 0 && dim % 4 == 0 && dst_rows >= 0, "tt_scatter_rows_f32: bad n/dim/dst_rows");
  if (n == 0) return TT_OK;
  TT_REQUIRE(src && idx && dst, "tt_scatter_rows_f32: null pointer");
  TT_REQUIRE(tt::aligned16(src) && tt::aligned16(dst), "tt_scatter_rows_f32: src/dst must be 16-byte aligned");
  const int dim4 = dim / 4;
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < dim4 && lpr_log2 < 6) ++lpr_log2;
  const int groups = 256 >> lpr_log2;
  const int64_t blocks = (n + groups - 1) / groups;
  TT_REQUIRE(blocks <= 0x7fffffff, "tt_scatter_rows_f32: n too large");
  hipStream_t stream = tt::as_stream(stream_);
  tt::launch("scatter_rows", scatter_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const tt::f32x4*>(src),
                     idx, n, dim4, lpr_log2, reinterpret_cast<tt::f32x4*>(dst), dst_rows);
  return tt::check_launch("tt_scatter_rows_f32");
}
