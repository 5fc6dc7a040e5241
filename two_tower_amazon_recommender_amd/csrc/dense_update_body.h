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

// the same update on values already in registers (the vector path loads parameter and accumulator as float4, with the slabs)
template <int OPT>
__device__ __forceinline__ float apply_val(float w, float& acc, float g, float l2, float lr, float eps) {
  g = __fadd_rn(g, __fmul_rn(2.0f * l2, w));
  if constexpr (OPT == TT_OPT_SGD) {
    return __fsub_rn(w, __fmul_rn(lr, g));
  } else {
    acc = __fadd_rn(acc, __fmul_rn(g, g));
    return __fsub_rn(w, __fdiv_rn(__fmul_rn(lr, g), sqrtf(__fadd_rn(acc, eps))));
  }
}

// One thread owns 4 consecutive elements (float4 loads when the segment allows it) and keeps U (8 or 16) slabs' loads in flight -
// together with the element's parameter (and accumulator) float4, requested FIRST; the additions stay in slab order (the
// oracle's order).  All loads are unconditional from clamped slab indices: with U = 16, 32 slabs are two memory round trips.  (Through
// r02: 8 in flight behind a first load of slab 0 and in front of four scalar parameter loads - six dependent round trips,
// 5 us for the step's 18 MB on the 36 CUs the fused optimizer launch leaves for it; r03 stamps.)
// (TPB = threads per block of the launch; every element is summed and updated on its own, so TPB changes nothing in the results)
// (U = slab loads in flight per thread: 16 where the kernel has the registers - 64 VGPRs of loads -, 8 inside kernels bound to
// 8 waves per SIMD)
template <int OPT, int TPB = 256, int U = 8>
__device__ __forceinline__ void dense_update_body(const tt_dense_seg& s, const int bx, const int nbx, int apply, float lr, float eps) {
  const int64_t stride = (int64_t)nbx * TPB;
  const bool vec = (s.count % 4 == 0) && (s.slab_stride % 4 == 0) && al16(s.grad_slabs) &&
                   (s.grad_out == nullptr || al16(s.grad_out)) && (!apply || al16(s.param)) &&
                   (!apply || OPT == TT_OPT_SGD || al16(s.accum));
  if (vec) {
    const int64_t n4 = s.count / 4, st4 = s.slab_stride / 4;
    const f32x4* __restrict__ gs = reinterpret_cast<const f32x4*>(s.grad_slabs);
    const int ns = s.n_slabs;
    const float l2 = s.l2;
    f32x4* param4 = reinterpret_cast<f32x4*>(s.param);
    f32x4* accum4 = reinterpret_cast<f32x4*>(s.accum);
    f32x4* gout4 = reinterpret_cast<f32x4*>(s.grad_out);
    for (int64_t i = (int64_t)bx * TPB + threadIdx.x; i < n4; i += stride) {
      f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f}, ac = w;
      if (apply) {                                        // (uniform)
        w = param4[i];
        if constexpr (OPT != TT_OPT_SGD) ac = accum4[i];
      }
      f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < ns; k0 += U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = k0 + u < ns ? k0 + u : ns - 1;    // past the last slab: re-read it (an L1 hit), not added
          v[u] = gs[(int64_t)k * st4 + i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (k0 + u < ns) {
            if (k0 + u == 0) g = v[u];                    // the sum STARTS at slab 0 (0 + x would turn -0 into +0)
            else
#pragma unroll
              for (int e = 0; e < 4; ++e) g[e] = __fadd_rn(g[e], v[u][e]);
          }
      }
      if (gout4 != nullptr) gout4[i] = g;
      if (!apply) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = ac[e];
        w[e] = apply_val<OPT>(w[e], acc, g[e], l2, lr, eps);
        ac[e] = acc;
      }
      if constexpr (OPT != TT_OPT_SGD) accum4[i] = ac;
      param4[i] = w;
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
