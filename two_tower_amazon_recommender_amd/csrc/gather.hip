// K1 — embedding row gather.  HBM-bound: random 4*dim-byte row reads,
// streaming 16 B/lane stores.  A group of dim/4 lanes (32 lanes = half a wave at dim 128) moves one
// row with one global_load_dwordx4 per lane; every lane keeps UNROLL independent rows in flight so a
// wave has UNROLL*2 random rows outstanding (guide: >=4 rows in flight per wave, >=16 waves per CU).
// Algorithmic bytes per row: 4*dim read + 4*dim written + 8 (the id).
#include "common.h"

namespace {

struct GatherArgs {
  const float* table[2];
  const int64_t* ids[2];
  float* out[2];
  int64_t rows[2];
};

// ACC: out[p,:] += table[ids[p],:] (a further feature summed into a tower input: the hashed category of cfg5)
template <int UNROLL, bool ACC>
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a, int dim4 /* dim/4 */, int lpr_log2,
                                                     int64_t n_ids, int32_t* __restrict__ oob_flag) {
  const int t = blockIdx.y;
  const tt::f32x4* __restrict__ table = reinterpret_cast<const tt::f32x4*>(a.table[t]);
  const int64_t* __restrict__ ids = a.ids[t];
  tt::f32x4* __restrict__ out = reinterpret_cast<tt::f32x4*>(a.out[t]);
  const int64_t rows = a.rows[t];

  const int lpr = 1 << lpr_log2;                 // lanes per row
  const int groups = 256 >> lpr_log2;            // row groups per block
  const int g = threadIdx.x >> lpr_log2;
  const int l = threadIdx.x & (lpr - 1);
  const int64_t row0 = ((int64_t)blockIdx.x * UNROLL) * groups + g;

  for (int c = l; c < dim4; c += lpr) {          // one trip unless dim > 256
    tt::f32x4 v[UNROLL];
    int64_t b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      b[u] = row0 + (int64_t)u * groups;
      v[u] = tt::f32x4{0.f, 0.f, 0.f, 0.f};
      if (b[u] < n_ids) {
        const int64_t id = ids[b[u]];
        if (id >= 0 && id < rows) {
          v[u] = table[id * dim4 + c];
        } else if (oob_flag != nullptr && c == 0 && id != -1) {   // -1 = padding slot (sharded exchange): zero row, no flag
          atomicOr(oob_flag, 1);
        }
      }
    }
    if constexpr (ACC) {
      tt::f32x4 o[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (b[u] < n_ids) o[u] = out[b[u] * dim4 + c];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (b[u] < n_ids) out[b[u] * dim4 + c] = o[u] + v[u];      // one f32 add per element (oracle: a + b)
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (b[u] < n_ids) out[b[u] * dim4 + c] = v[u];
    }
  }
}

// FNV-1a (64-bit) of the bytes of each zero-padded row up to its first NUL, reduced modulo n_buckets.
__global__ __launch_bounds__(256) void hash_bucket_kernel(const uint8_t* __restrict__ rows, int64_t n, int width,
                                                          uint64_t n_buckets, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = rows + i * width;
  uint64_t h = 0xCBF29CE484222325ull;
  for (int k = 0; k < width; ++k) {
    const uint8_t c = r[k];
    if (c == 0) break;
    h = (h ^ c) * 0x100000001B3ull;
  }
  out[i] = (int64_t)(h % n_buckets);
}

int launch(const GatherArgs& a, int n_tables, int32_t dim, int64_t n_ids, int32_t* oob_flag, hipStream_t stream,
           const char* what, bool accumulate = false) {
  if (n_ids == 0) return TT_OK;
  const int dim4 = dim / 4;
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < dim4 && lpr_log2 < 6) ++lpr_log2;
  const int groups = 256 >> lpr_log2;
  constexpr int UNROLL = 4;
  const int64_t rows_per_block = (int64_t)groups * UNROLL;
  const int64_t blocks = (n_ids + rows_per_block - 1) / rows_per_block;
  TT_REQUIRE(blocks <= 0x7fffffff, "%s: n_ids too large", what);
  if (accumulate)
    tt::launch("gather", (gather_kernel<UNROLL, true>), dim3((unsigned)blocks, n_tables), dim3(256), 0, stream, a, dim4, lpr_log2,
                       n_ids, oob_flag);
  else
    tt::launch("gather", (gather_kernel<UNROLL, false>), dim3((unsigned)blocks, n_tables), dim3(256), 0, stream, a, dim4, lpr_log2,
                       n_ids, oob_flag);
  return tt::check_launch(what);
}

}  // namespace

extern "C" int tt_embedding_gather_f32(const float* table, int64_t num_rows, int32_t dim, const int64_t* ids,
                                       int64_t n_ids, float* out, int32_t* oob_flag, tt_stream_t stream) {
  TT_REQUIRE(n_ids >= 0 && num_rows > 0, "tt_embedding_gather_f32: bad n_ids/num_rows");
  TT_REQUIRE(dim > 0 && dim % 4 == 0, "tt_embedding_gather_f32: dim must be a positive multiple of 4 (got %d)", dim);
  TT_REQUIRE(n_ids == 0 || (table && ids && out), "tt_embedding_gather_f32: null pointer");
  TT_REQUIRE(tt::aligned16(table) && tt::aligned16(out), "tt_embedding_gather_f32: table/out must be 16-byte aligned");
  GatherArgs a{};
  a.table[0] = table; a.ids[0] = ids; a.out[0] = out; a.rows[0] = num_rows;
  return launch(a, 1, dim, n_ids, oob_flag, tt::as_stream(stream), "tt_embedding_gather_f32");
}

extern "C" int tt_embedding_gather2_f32(const float* table_a, int64_t rows_a, const int64_t* ids_a, float* out_a,
                                        const float* table_b, int64_t rows_b, const int64_t* ids_b, float* out_b,
                                        int32_t dim, int64_t n_ids, int32_t* oob_flag, tt_stream_t stream) {
  TT_REQUIRE(n_ids >= 0 && rows_a > 0 && rows_b > 0, "tt_embedding_gather2_f32: bad n_ids/rows");
  TT_REQUIRE(dim > 0 && dim % 4 == 0, "tt_embedding_gather2_f32: dim must be a positive multiple of 4 (got %d)", dim);
  TT_REQUIRE(n_ids == 0 || (table_a && ids_a && out_a && table_b && ids_b && out_b),
             "tt_embedding_gather2_f32: null pointer");
  TT_REQUIRE(tt::aligned16(table_a) && tt::aligned16(out_a) && tt::aligned16(table_b) && tt::aligned16(out_b),
             "tt_embedding_gather2_f32: tables/outs must be 16-byte aligned");
  GatherArgs a{};
  a.table[0] = table_a; a.ids[0] = ids_a; a.out[0] = out_a; a.rows[0] = rows_a;
  a.table[1] = table_b; a.ids[1] = ids_b; a.out[1] = out_b; a.rows[1] = rows_b;
  return launch(a, 2, dim, n_ids, oob_flag, tt::as_stream(stream), "tt_embedding_gather2_f32");
}

extern "C" int tt_embedding_gather_add_f32(const float* table, int64_t num_rows, int32_t dim, const int64_t* ids,
                                           int64_t n_ids, float* out, int32_t* oob_flag, tt_stream_t stream) {
  TT_REQUIRE(n_ids >= 0 && num_rows > 0, "tt_embedding_gather_add_f32: bad n_ids/num_rows");
  TT_REQUIRE(dim > 0 && dim % 4 == 0, "tt_embedding_gather_add_f32: dim must be a positive multiple of 4 (got %d)", dim);
  TT_REQUIRE(n_ids == 0 || (table && ids && out), "tt_embedding_gather_add_f32: null pointer");
  TT_REQUIRE(tt::aligned16(table) && tt::aligned16(out), "tt_embedding_gather_add_f32: table/out must be 16-byte aligned");
  GatherArgs a{};
  a.table[0] = table; a.ids[0] = ids; a.out[0] = out; a.rows[0] = num_rows;
  return launch(a, 1, dim, n_ids, oob_flag, tt::as_stream(stream), "tt_embedding_gather_add_f32", true);
}

extern "C" int tt_hash_bucket_u8(const uint8_t* rows, int64_t n, int32_t width, int64_t n_buckets, int64_t* out,
                                 tt_stream_t stream_) {
  TT_REQUIRE(n >= 0 && width > 0 && n_buckets > 0, "tt_hash_bucket_u8: need n >= 0, width > 0, n_buckets > 0");
  if (n == 0) return TT_OK;
  TT_REQUIRE(rows && out, "tt_hash_bucket_u8: null pointer");
  const int64_t blocks = (n + 255) / 256;
  TT_REQUIRE(blocks <= 0x7fffffff, "tt_hash_bucket_u8: n too large");
  hipStream_t stream = tt::as_stream(stream_);
  tt::launch("hash_bucket", hash_bucket_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, n, width, (uint64_t)n_buckets, out);
  return tt::check_launch("tt_hash_bucket_u8");
}
