// Shared helpers for librecattend.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "recattend.h"

namespace ra {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Returns 0 or the hipError_t of the launch that was just issued.
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
  }
  return 0;
}

inline int fail(int code, const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

constexpr int kWave = 64;  // CDNA wavefront

// Wave priority of the decode loop's LATENCY-bound kernels (controller, patch-sized convs, extract, paste, score).  In the
// decode pipeline they share every SIMD with another batch's MFMA-bound controller-CNN waves, and the arbiter deals issue
// slots oldest-first: a tail wave's few hundred dependent instructions each queue behind the other waves' 32-cycle MFMAs.
// s_setprio would let the tail wave issue as soon as the pipe is free.  MEASURED in round 5 and NOT the lever: with
// RA_TAIL_PRIO=3 on all of them the pipelined rate falls 52.3k -> 50.7k, on any subset (RA_TAIL_PRIO_MASK) it is equal or
// lower within the run-to-run spread (profiles/r05_decode_schedule_probes.txt).  Kept as a measuring aid, default 0 = off.
int tail_prio(int kind = 4);  // ra_core.hip; kind: 1 = controller, 2 = patch-sized conv, 4 = extract / paste / score (RA_TAIL_PRIO_MASK)
__device__ __forceinline__ void raise_prio(int p) {
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace ra
