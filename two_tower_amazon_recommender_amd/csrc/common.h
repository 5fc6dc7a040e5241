// Shared host/device helpers for the gfx950 kernels behind include/twotower_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/twotower_hip.h"

namespace tt {

// thread-local last-error message (tt_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(tt_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TT_OK;
}

// built-in kernel timing (tt_profile_enable / tt_profile_read); see capi_common.hip
extern bool g_prof_on;
void prof_record(const char* tag, hipStream_t stream, bool end);
// A scope of SEVERAL launches: a hipEventRecord pair around them (each record is a barrier packet: 4-7 us of stream time)
struct ProfScope {
  const char* tag;
  hipStream_t stream;
  ProfScope(const char* t, hipStream_t s) : tag(t), stream(s) { if (g_prof_on) prof_record(tag, stream, false); }
  ~ProfScope() { if (g_prof_on) prof_record(tag, stream, true); }
};
// ONE kernel under a tag: launched through hipExtLaunchKernelGGL with a start / stop event pair, which the runtime fills with
// the dispatch's own begin / end timestamps (the ones rocprofv3 --kernel-trace reports) - no barrier packet, nothing added to
// the stream.  prof_kernel_events hands out the pair of the next sampled launch of `tag` (false: not enabled / not sampled /
// capacity reached).  TT_PROF_BRACKETS=1 (environment, read once) restores the hipEventRecord brackets of r01-r03 for an A/B.
bool prof_kernel_events(const char* tag, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop, bool* bracket);
void prof_kernel_end(const char* tag, hipStream_t stream);
template <typename K, typename... Args>
inline void launch(const char* tag, K kern, dim3 grid, dim3 block, unsigned lds, hipStream_t stream, Args... args) {
  if (g_prof_on) {
    hipEvent_t e0, e1;
    bool bracket = false;
    if (prof_kernel_events(tag, stream, &e0, &e1, &bracket)) {
      hipExtLaunchKernelGGL(kern, grid, block, lds, stream, e0, e1, 0u, args...);
      return;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, args...);
    if (bracket) prof_kernel_end(tag, stream);
    return;
  }
  hipLaunchKernelGGL(kern, grid, block, lds, stream, args...);
}

// row-range id lists (tt_id_buckets): uint32 words between two ranges' counters - one counter per 256-byte line, so that the
// forward pass's atomics on different ranges never serialise on a shared line
constexpr int kBucketCountStride = 64;
constexpr int kBucketGroupsMax = 256;      // counters / lists per table in the workspace

int gemm_nt(const float* a, const float* b, float* c, int64_t m, int64_t n, int64_t k, hipStream_t stream);   // gemm.hip

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define TT_REQUIRE(cond, ...) do { if (!(cond)) return ::tt::fail(TT_ERR_INVALID_ARG, __VA_ARGS__); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// C/D row of accumulator register `reg` for lane half h (32x32 MFMA): guide §3.
__device__ __forceinline__ constexpr int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// dst's lane `lane` (a compile-time constant) = the wave-uniform value v (v_writelane_b32; this hipcc has no builtin for it).
// s_nop 4: v usually is (half of) a lane mask a VALU compare has JUST written to an SGPR pair, and an SGPR written by the VALU
// is not yet readable as a v_writelane source on the next cycles - without the wait states the lane got the PREVIOUS
// compare's mask (r03: test_relu_sign_bits_are_the_activation_mask_bit_for_bit caught it); the compiler's hazard recogniser
// does not look inside inline asm.
__device__ __forceinline__ void writelane(uint32_t& dst, uint32_t v, int lane) {
  asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(v), "n"(lane));
}

__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t splitmix_host(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace tt
