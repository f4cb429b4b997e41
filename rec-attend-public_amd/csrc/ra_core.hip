// Library-wide bits of librecattend.so: version and the thread-local last-error string.
#include <cstdarg>
#include <cstdio>

#include "ra_common.h"

namespace ra {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ra

extern "C" int ra_version(void) { return 100; }
extern "C" const char *ra_last_error_string(void) { return ra::g_err; }

// Test aid: leave NaN in the LDS of every CU, so that a kernel which consumes shared memory it
// never wrote shows up in the parity tests instead of depending on what ran before it.
namespace ra {
namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(float *sink) {
  extern __shared__ float sh[];
  const unsigned nan_bits = 0x7fc00000u;
  for (int e = threadIdx.x; e < 160 * 256; e += 256) sh[e] = __builtin_bit_cast(float, nan_bits);
  __syncthreads();
  if (sink && sh[(threadIdx.x * 37) % (160 * 256)] == 0.0f) sink[0] = 1.0f;  // keep the stores alive
}
}  // namespace
}  // namespace ra

extern "C" int ra_debug_poison_lds(void *stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra::poison_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ra::poison_lds_kernel, dim3(1024), dim3(256), 160 * 1024, ra::as_stream(stream),
                     static_cast<float *>(nullptr));
  return ra::launch_status("ra_debug_poison_lds");
}
