// Dense (tower) parameter update: sums the split-K gradient slabs in slab order, adds the L2 term
// (Keras kernel_regularizer=l2: d/dw l2*sum(w^2) = 2*l2*w; configs/data_config.yaml:59) and applies
// SGD or Keras-2.15 Adagrad in place — every Dense kernel and bias of both towers in ONE launch
// (blockIdx.y = segment).  Elementwise, HBM/L2-bound; ~0.5 MB of parameters in the reference config.
#include "common.h"
#include "dense_update_body.h"

namespace {

using tt::SegTable;

template <int OPT>
__global__ __launch_bounds__(256) void dense_update_kernel(SegTable tbl, int apply, float lr, float eps) {
  tt::dense_update_body<OPT, 256, 16>(tbl.seg[blockIdx.y], blockIdx.x, gridDim.x, apply, lr, eps);
}

}  // namespace

extern "C" int tt_dense_update_f32(const tt_dense_seg* segs, int32_t n_segs, int32_t opt, int32_t apply, float lr, float eps,
                                   tt_stream_t stream) {
  TT_REQUIRE(segs != nullptr && n_segs > 0 && n_segs <= TT_MAX_DENSE_SEGS, "tt_dense_update_f32: need 1..%d segments",
             TT_MAX_DENSE_SEGS);
  TT_REQUIRE(opt == TT_OPT_SGD || opt == TT_OPT_ADAGRAD, "tt_dense_update_f32: unknown optimizer %d", opt);
  SegTable tbl{};
  int64_t max_count = 0;
  for (int i = 0; i < n_segs; ++i) {
    const tt_dense_seg& s = segs[i];
    TT_REQUIRE(s.count > 0 && s.n_slabs >= 1 && s.grad_slabs != nullptr, "tt_dense_update_f32: segment %d: bad count/slabs", i);
    TT_REQUIRE(!apply || s.param != nullptr, "tt_dense_update_f32: segment %d: null param", i);
    TT_REQUIRE(!apply || opt == TT_OPT_SGD || s.accum != nullptr, "tt_dense_update_f32: segment %d: Adagrad needs accum", i);
    TT_REQUIRE(apply || s.grad_out != nullptr, "tt_dense_update_f32: segment %d: apply == 0 needs grad_out", i);
    tbl.seg[i] = s;
    if (s.count > max_count) max_count = s.count;
  }
  int64_t bx = (max_count + 255) / 256;
  if (bx > 512) bx = 512;
  if (opt == TT_OPT_SGD)
    tt::launch("dense_update", dense_update_kernel<TT_OPT_SGD>, dim3((unsigned)bx, (unsigned)n_segs), dim3(256), 0,
                       tt::as_stream(stream), tbl, apply, lr, eps);
  else
    tt::launch("dense_update", dense_update_kernel<TT_OPT_ADAGRAD>, dim3((unsigned)bx, (unsigned)n_segs), dim3(256), 0,
                       tt::as_stream(stream), tbl, apply, lr, eps);
  return tt::check_launch("tt_dense_update_f32");
}
