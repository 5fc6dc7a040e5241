// Device body of the dense (tower) parameter update, shared by dense_update.hip (its own launch: the sharded trainer's
// reduce / apply passes) and sparse.hip (the single-launch optimizer step of the plain trainer).
#pragma once
#include "common.h"

namespace tt {

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
// (TPB = threads per block of the launch; every element is summed and updated on its own, so TPB changes nothing in the results)
template <int OPT, int TPB = 256>
__device__ __forceinline__ void dense_update_body(const tt_dense_seg& s, const int bx, const int nbx, int apply, float lr, float eps) {
  const int64_t stride = (int64_t)nbx * TPB;
  const bool vec = (s.count % 4 == 0) && (s.slab_stride % 4 == 0) && al16(s.grad_slabs) &&
                   (s.grad_out == nullptr || al16(s.grad_out)) && (!apply || al16(s.param)) &&
                   (!apply || OPT == TT_OPT_SGD || al16(s.accum));
  if (vec) {
    const int64_t n4 = s.count / 4, st4 = s.slab_stride / 4;
    const f32x4* __restrict__ gs = reinterpret_cast<const f32x4*>(s.grad_slabs);
    for (int64_t i = (int64_t)bx * TPB + threadIdx.x; i < n4; i += stride) {
      f32x4 g = gs[i];
      for (int k0 = 1; k0 < s.n_slabs; k0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (k0 + u < s.n_slabs) ? gs[(int64_t)(k0 + u) * st4 + i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < s.n_slabs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = __fadd_rn(g[e], v[u][e]);
          }
      }
      if (s.grad_out != nullptr) reinterpret_cast<f32x4*>(s.grad_out)[i] = g;
      if (!apply) continue;
      f32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = apply_one<OPT>(s, 4 * i + e, g[e], lr, eps);
      reinterpret_cast<f32x4*>(s.param)[i] = w;
    }
    return;
  }
  for (int64_t i = (int64_t)bx * TPB + threadIdx.x; i < s.count; i += stride) {
    float g = s.grad_slabs[i];
    for (int k = 1; k < s.n_slabs; ++k) g = __fadd_rn(g, s.grad_slabs[(int64_t)k * s.slab_stride + i]);
    if (s.grad_out != nullptr) s.grad_out[i] = g;
    if (!apply) continue;
    s.param[i] = apply_one<OPT>(s, i, g, lr, eps);
  }
}


}  // namespace tt
