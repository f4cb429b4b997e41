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

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace ra
