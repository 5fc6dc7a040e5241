// Negative control for tests/isa_audit/audit_barriers.py: a workgroup barrier inside a loop whose trip count is a PER-LANE value
// (what the r02 note described: the exit test is a vector compare).  The audit must flag it.  Never launched.
#include <hip/hip_runtime.h>
namespace {
template <int A, int B, bool C, bool D, int E, int F>
__global__ void score_kernel(const int* n, float* out) {
  __shared__ float buf[256];
  const int trips = n[threadIdx.x];                 // per lane
  float acc = 0.f;
  for (int t = 0; t < trips; ++t) {
    buf[threadIdx.x] = acc + t;
    __syncthreads();
    acc += buf[(threadIdx.x + 1) & 255];
    __syncthreads();
  }
  out[threadIdx.x] = acc;
}
}  // namespace
void launch(const int* n, float* out) { hipLaunchKernelGGL((score_kernel<0, 0, false, false, 4, 0>), dim3(1), dim3(256), 0, 0, n, out); }
