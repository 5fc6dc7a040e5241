// tt_train_step_f32 - the whole train step behind ONE C entry (include/twotower_hip.h, ABI v8).  Host code only: it calls the
// library's own entry points in the order a caller would, so the arithmetic and the launches are theirs, bit for bit.
#include "common.h"
#include <atomic>
#include <cstdlib>

extern "C" int tt_train_step_f32(const tt_train_step* s, tt_stream_t stream) {
  TT_REQUIRE(s != nullptr, "tt_train_step_f32: null step");
  TT_REQUIRE(s->n_layers >= 1 && s->n_layers <= TT_MAX_TOWER_LAYERS, "tt_train_step_f32: 1..%d layers per tower", TT_MAX_TOWER_LAYERS);
  TT_REQUIRE(s->batch > 0, "tt_train_step_f32: batch must be positive");
  TT_REQUIRE(s->dropout_rate >= 0.f && s->dropout_rate < 1.f, "tt_train_step_f32: dropout_rate must be in [0, 1)");
  TT_REQUIRE(s->scorer_precision == 0 || s->scorer_precision == 1, "tt_train_step_f32: scorer_precision must be 0 (f32) or 1 (bf16x3)");
  TT_REQUIRE(s->n_tables >= 2 && s->n_tables <= 3, "tt_train_step_f32: 2 or 3 embedding tables");
  TT_REQUIRE(s->n_segs >= 1 && s->n_segs <= TT_MAX_DENSE_SEGS, "tt_train_step_f32: 1..%d dense segments", TT_MAX_DENSE_SEGS);
  const int L = s->n_layers;
  int rc;
  // two-layer towers: both layers of both towers in one launch (csrc/tower.hip); TT_FUSED_TOWER=0 keeps the two launches (A/B)
  static const bool fused_tower = std::getenv("TT_FUSED_TOWER") == nullptr || std::atoi(std::getenv("TT_FUSED_TOWER")) != 0;
  // (r04: towers of MORE than two layers - the reference's own [512, 256, 128], configs/data_config.yaml:56-57 - run their last
  // two layers, ReLU hidden + linear output: the fused kernel's pattern, as the fused launch and the layers below one by one)
  const bool fwd2 = fused_tower && L >= 2 && tt_tower_fwd2_supported(s->batch, s->dims[L - 2], s->dims[L - 1], s->dims[L]);
  const int Lf = fwd2 ? L - 2 : L;                 // layers that run as their own launch
  // r04, measured and left OFF (TT_DENSE_RIDERS=1 switches it on): the dense update of layer l's parameters riding in the backward
  // launch of layer l - 1 (tt_dense_bwd_batched_update_f32: its gradient slabs are complete and nothing below reads its weights),
  // so that the optimizer launch - whose HBM burst is the embedding rows' - keeps only the segments of layer 0.  Same arithmetic.
  // cfg3, same box, alternating (profiles/r04_optimizer_ab.txt): optimizer launch 12.0 -> 11.5 us, but the two backward launches
  // 90.9 -> 93.0 us together - the step 0.5624 -> 0.5636 ms.
  static const bool want_riders = std::getenv("TT_DENSE_RIDERS") != nullptr && std::atoi(std::getenv("TT_DENSE_RIDERS")) != 0;
  tt_dense_seg rest[TT_MAX_DENSE_SEGS];
  int layer_of[TT_MAX_DENSE_SEGS];
  for (int i = 0; i < s->n_segs; ++i) {
    layer_of[i] = 0;
    for (int l = 1; l < L && want_riders; ++l)
      for (int t = 0; t < 2; ++t)
        if (s->segs[i].grad_slabs != nullptr && (s->segs[i].grad_slabs == s->bwd[l][t].dw_slabs || s->segs[i].grad_slabs == s->bwd[l][t].db_slabs))
          layer_of[i] = l;
  }
  int n_rest = 0;
  for (int i = 0; i < s->n_segs; ++i)
    if (layer_of[i] == 0) rest[n_rest++] = s->segs[i];
  TT_REQUIRE(n_rest >= 1, "tt_train_step_f32: no dense segment is left for the optimizer launch");
  // r04: the forward lookup hands the optimizer launch its row-range id lists (tt_id_buckets): with a bucket workspace, a fused
  // tower forward (the lookup that can fill them) and a shape that takes lists, both launches get the same descriptors - cut by
  // tt_optimizer_ids_geometry, one generation per step.  TT_ID_BUCKETS=0 keeps the optimizer's own id scan (A/B).
  static const bool want_lists = std::getenv("TT_ID_BUCKETS") == nullptr || std::atoi(std::getenv("TT_ID_BUCKETS")) != 0;
  static std::atomic<uint32_t> generation{0};
  tt_dense_fwd_args f0[2] = {s->fwd[0][0], s->fwd[0][1]};
  tt_sparse_table_ids tabs[3] = {s->tables[0], s->tables[1], s->tables[2]};
  // (the lookup that fills them is the fused two-layer forward's when the towers have two layers, else layer 0's own launch)
  if (want_lists && (Lf > 0 || L == 2) && s->id_bucket_ws != nullptr && f0[0].lookup.ids != nullptr && f0[1].lookup.ids != nullptr &&
      f0[0].lookup.ids == tabs[0].ids && f0[1].lookup.ids == tabs[1].ids && f0[0].lookup.table_rows == tabs[0].rows &&
      f0[1].lookup.table_rows == tabs[1].rows) {
    const int64_t per = tt_id_buckets_workspace_bytes();
    TT_REQUIRE((reinterpret_cast<uintptr_t>(s->id_bucket_ws) & 255u) == 0 && s->id_bucket_ws_bytes >= 2 * per,
               "tt_train_step_f32: id_bucket_ws must be 256-byte aligned and hold 2 * tt_id_buckets_workspace_bytes() = %lld bytes",
               (long long)(2 * per));
    int64_t rows[3] = {tabs[0].rows, tabs[1].rows, s->n_tables > 2 ? tabs[2].rows : 1};
    int32_t groups[3] = {0, 0, 0}, cap = 0;
    uint32_t width[3] = {0, 0, 0};
    rc = tt_optimizer_ids_geometry(rows, s->n_tables, s->dims[0], s->batch, rest, n_rest, groups, width, &cap);      // (the segments the optimizer launch will get)
    if (rc != TT_OK) return rc;
    if (cap > 0 && groups[0] <= 256 && groups[1] <= 256) {
      const uint32_t gen = generation.fetch_add(1u) + 1u;
      for (int t = 0; t < 2; ++t) {
        char* base = static_cast<char*>(s->id_bucket_ws) + t * per;
        tt_id_buckets bk{reinterpret_cast<uint32_t*>(base), reinterpret_cast<uint64_t*>(base + (size_t)tt::kBucketGroupsMax * tt::kBucketCountStride * 4), groups[t], width[t], cap, gen};
        f0[t].lookup.buckets = bk;
        tabs[t].buckets = bk;
      }
    }
  }
  for (int l = 0; l < Lf; ++l) {
    const bool hidden = l < L - 1;
    const bool drop = hidden && s->dropout_rate > 0.f;
    rc = tt_dense_fwd_batched_f32(l == 0 ? f0 : s->fwd[l], 2, s->batch, s->dims[l], s->dims[l + 1], hidden ? 1 : 0, drop ? s->dropout_rate : 0.f,
                                  s->dropout_seed, drop ? s->dropout_row0 * (uint64_t)s->dims[l + 1] : 0ull, stream);
    if (rc != TT_OK) return rc;
  }
  if (fwd2) {
    const bool drop = s->dropout_rate > 0.f;
    rc = tt_tower_fwd2_batched_f32(L == 2 ? f0 : s->fwd[L - 2], s->fwd[L - 1], 2, s->batch, s->dims[L - 2], s->dims[L - 1], s->dims[L],
                                   drop ? s->dropout_rate : 0.f, s->dropout_seed, drop ? s->dropout_row0 * (uint64_t)s->dims[L - 1] : 0ull, stream);
    if (rc != TT_OK) return rc;
  }
  const float* q = s->fwd[L - 1][0].y;
  const float* c = s->fwd[L - 1][1].y;
  float* dq = const_cast<float*>(s->bwd[L - 1][0].dz);
  float* dc = const_cast<float*>(s->bwd[L - 1][1].dz);
  if (s->scorer_precision == 0)
    rc = tt_retrieval_fwd_bwd_f32(q, c, s->batch, s->batch, s->dims[L], 0, s->inv_temperature, s->sample_weight, s->cand_prob,
                                  s->cand_ids, nullptr, 1.0f, s->retrieval_ws, s->retrieval_ws_bytes, s->lse, s->per_row, s->loss, dq, dc, stream);
  else
    rc = tt_retrieval_fwd_bwd_bf16x3_f32(q, c, s->batch, s->batch, s->dims[L], 0, s->inv_temperature, s->sample_weight, s->cand_prob,
                                         s->cand_ids, nullptr, 1.0f, s->retrieval_ws, s->retrieval_ws_bytes, s->lse, s->per_row, s->loss, dq, dc, stream);
  if (rc != TT_OK) return rc;
  // the same f32 arithmetic as the forward launcher's keep scale: 1.0f / (1.0f - rate)
  const float dx_scale = s->dropout_rate > 0.f ? 1.0f / (1.0f - s->dropout_rate) : 1.0f;
  for (int l = L - 1; l >= 0; --l) {
    tt_dense_seg ride[TT_MAX_DENSE_SEGS];
    int n_ride = 0;
    for (int i = 0; i < s->n_segs; ++i)
      if (layer_of[i] == l + 1) ride[n_ride++] = s->segs[i];
    if (n_ride > 0)
      rc = tt_dense_bwd_batched_update_f32(s->bwd[l], 2, l > 0 ? dx_scale : 1.0f, s->batch, s->dims[l], s->dims[l + 1], ride, n_ride, s->opt,
                                           s->lr, s->eps, stream);
    else
      rc = tt_dense_bwd_batched_f32(s->bwd[l], 2, l > 0 ? dx_scale : 1.0f, s->batch, s->dims[l], s->dims[l + 1], stream);
    if (rc != TT_OK) return rc;
  }
  return tt_optimizer_step_ids_f32(s->opt, tabs, s->n_tables, s->dims[0], s->batch, rest, n_rest, s->lr, s->eps, stream);
}
