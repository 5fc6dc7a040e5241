// Dense (tower) parameter update: sums the split-K gradient slabs in slab order, adds the L2 term
// (Keras kernel_regularizer=l2: d/dw l2*sum(w^2) = 2*l2*w; configs/data_config.yaml:59) and applies
// SGD or Keras-2.15 Adagrad in place — every Dense kernel and bias of both towers in ONE launch
// (blockIdx.y = segment).  Elementwise, HBM/L2-bound; ~0.5 MB of parameters in the reference config.
#include "common.h"

namespace {

struct SegTable {
  tt_dense_seg seg[TT_MAX_DENSE_SEGS];
};

__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int OPT>
__device__ __forceinline__ float apply_one(const tt_dense_seg& s, int64_t i, float g, float lr, float eps) {
  float w = s.param[i];
  g = __fadd_rn(g, __fmul_rn(2.0f * s.l2, w));
  if constexpr (OPT == TT_OPT_SGD) {
    w = __fsub_rn(w, __fmul_rn(lr, g));
  } else {
    const float acc = __fadd_rn(s.accum[i], __fmul_rn(g, g));
    s.accum[i] = acc;
    w = __fsub_rn(w, __fdiv_rn(__fmul_rn(lr, g), sqrtf(__fadd_rn(acc, eps))));
  }
  return w;
}

// One thread owns 4 consecutive elements (float4 loads when the segment allows it) and keeps 8 slabs' loads in
// flight; the additions stay in slab order (the oracle's order).
template <int OPT>
__global__ __launch_bounds__(256) void dense_update_kernel(SegTable tbl, int apply, float lr, float eps) {
  const tt_dense_seg s = tbl.seg[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * 256;
  const bool vec = (s.count % 4 == 0) && (s.slab_stride % 4 == 0) && al16(s.grad_slabs) &&
                   (s.grad_out == nullptr || al16(s.grad_out)) && (!apply || al16(s.param)) &&
                   (!apply || OPT == TT_OPT_SGD || al16(s.accum));
  if (vec) {
    const int64_t n4 = s.count / 4, st4 = s.slab_stride / 4;
    const tt::f32x4* __restrict__ gs = reinterpret_cast<const tt::f32x4*>(s.grad_slabs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      tt::f32x4 g = gs[i];
      for (int k0 = 1; k0 < s.n_slabs; k0 += 8) {
        tt::f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (k0 + u < s.n_slabs) ? gs[(int64_t)(k0 + u) * st4 + i] : tt::f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < s.n_slabs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = __fadd_rn(g[e], v[u][e]);
          }
      }
      if (s.grad_out != nullptr) reinterpret_cast<tt::f32x4*>(s.grad_out)[i] = g;
      if (!apply) continue;
      tt::f32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = apply_one<OPT>(s, 4 * i + e, g[e], lr, eps);
      reinterpret_cast<tt::f32x4*>(s.param)[i] = w;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < s.count; i += stride) {
    float g = s.grad_slabs[i];
    for (int k = 1; k < s.n_slabs; ++k) g = __fadd_rn(g, s.grad_slabs[(int64_t)k * s.slab_stride + i]);
    if (s.grad_out != nullptr) s.grad_out[i] = g;
    if (!apply) continue;
    s.param[i] = apply_one<OPT>(s, i, g, lr, eps);
  }
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
  tt::ProfScope prof("dense_update", tt::as_stream(stream));
  if (opt == TT_OPT_SGD)
    hipLaunchKernelGGL(dense_update_kernel<TT_OPT_SGD>, dim3((unsigned)bx, (unsigned)n_segs), dim3(256), 0,
                       tt::as_stream(stream), tbl, apply, lr, eps);
  else
    hipLaunchKernelGGL(dense_update_kernel<TT_OPT_ADAGRAD>, dim3((unsigned)bx, (unsigned)n_segs), dim3(256), 0,
                       tt::as_stream(stream), tbl, apply, lr, eps);
  return tt::check_launch("tt_dense_update_f32");
}
