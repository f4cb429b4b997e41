// Library-wide bits of librecattend.so: version and the thread-local last-error string.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

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

namespace ra {
int tail_prio(int kind) {  // read on every call (an int parse): tests and A/B probes flip RA_TAIL_PRIO between launches / captures
  const char *m = getenv("RA_TAIL_PRIO_MASK");
  if (m && !(atoi(m) & kind)) return 0;
  const char *e = getenv("RA_TAIL_PRIO");
  const int v = e ? atoi(e) : 0;  // measured (profiles/r05_decode_schedule_probes.txt): no level, on no subset of the kernels, helps
  return v < 0 ? 0 : (v > 3 ? 3 : v);
}
}  // namespace ra

// ---- dynamic tile tickets (ra_common.h): the calling thread's bound scratch and its cursor ----
namespace ra {
namespace {
thread_local unsigned *g_tk_base = nullptr;
thread_local int g_tk_slots = 0, g_tk_next = 0;
__global__ __launch_bounds__(64) void xcc_census_kernel(int *seen) {
  if (threadIdx.x == 0) atomicOr(seen, 1 << xcc_id());
}
// The pools are keyed by HW_REG_XCC_ID 0..7: a device (or partition mode) whose workgroups do not land on exactly those eight
// XCDs would leave pools undrawn, so tickets are only handed out where a census launch has seen all eight and nothing else.
}  // namespace
int xcc_census_ok() {
  static int ok = -1;
  if (ok < 0) {
    int *seen = nullptr, h = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(nullptr, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return -1;  // decide outside a capture
    if (hipMalloc(&seen, sizeof(int)) != hipSuccess) return ok = 0;
    (void)hipMemset(seen, 0, sizeof(int));
    hipLaunchKernelGGL(xcc_census_kernel, dim3(1024), dim3(64), 0, nullptr, seen);
    const bool good = hipMemcpy(&h, seen, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(seen);
    ok = (good && h == 0xff) ? 1 : 0;
  }
  return ok;
}
namespace {
int tickets_supported() {
  const char *e = getenv("RA_TILE_TICKETS");
  if (e && atoi(e) == 0) return 0;
  return xcc_census_ok() == 1 ? 1 : 0;
}
}  // namespace
unsigned *take_ticket_slots(int n, int grid_x) {
  // fewer than 8 workgroups per XCD: the static walk (a pool must never be left without a workgroup to draw from it)
  if (!g_tk_base || n <= 0 || grid_x < 64 || g_tk_next + n > g_tk_slots) return nullptr;
  unsigned *p = g_tk_base + (size_t)g_tk_next * kTicketSlotWords;
  g_tk_next += n;
  return p;
}
}  // namespace ra

extern "C" int ra_tile_tickets_bind(void *scratch, int slots) {
  if (!scratch || slots <= 0) {
    ra::g_tk_base = nullptr;
    ra::g_tk_slots = ra::g_tk_next = 0;
    return 0;
  }
  if (reinterpret_cast<uintptr_t>(scratch) & 127) return ra::fail(RA_E_INVALID, "ra_tile_tickets_bind: the scratch must be 128-byte aligned");
  if (!ra::tickets_supported()) {  // not an error: the launches keep their static tile walk
    ra::g_tk_base = nullptr;
    ra::g_tk_slots = ra::g_tk_next = 0;
    return 0;
  }
  ra::g_tk_base = static_cast<unsigned *>(scratch);
  ra::g_tk_slots = slots;
  ra::g_tk_next = 0;
  return 1;
}
extern "C" int ra_tile_tickets_slot_bytes(void) { return ra::kTicketSlotWords * (int)sizeof(unsigned); }

extern "C" int ra_version(void) { return RA_ABI_VERSION; }
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

// Test aid: hold part of ONE XCD's CUs for a while (see recattend.h).
namespace ra {
namespace {
__global__ __launch_bounds__(64) void park_xcd_kernel(int xcd, unsigned long long ticks, int *resident) {
  extern __shared__ float sh[];
  if (xcc_id() != xcd) return;
  if (threadIdx.x == 0) {
    sh[0] = 1.0f;  // (the dynamic LDS is what keeps a second workgroup off this CU)
    if (resident) atomicAdd(resident, 1);
    const unsigned long long t0 = wall_clock64();  // 100 MHz, constant
    for (unsigned n = 0; n < 400000000u; ++n) {    // bounded whatever the clock does
      if (wall_clock64() - t0 >= ticks) break;
      __builtin_amdgcn_s_sleep(64);
    }
  }
}
}  // namespace
}  // namespace ra

extern "C" int ra_debug_park_xcd(int xcd, int n_wg, int lds_bytes, int millis, int *resident, void *stream) {
  if (xcd < 0 || xcd > 7 || n_wg <= 0 || n_wg > 64 || lds_bytes < 0 || lds_bytes > 160 * 1024 || millis < 0)
    return ra::fail(RA_E_INVALID, "ra_debug_park_xcd: bad argument");
  if (millis > 20000) millis = 20000;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra::park_xcd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ra::park_xcd_kernel, dim3(8 * n_wg), dim3(64), (size_t)lds_bytes, ra::as_stream(stream), xcd,
                     (unsigned long long)millis * 100000ull, resident);
  return ra::launch_status("ra_debug_park_xcd");
}

// ---------------------------------------------------------------------------------------------
// Small utilities of the decode loop that must not cost a framework kernel inside the HIP graph.
namespace ra {
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_kernel(float *p, size_t n4, size_t n, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  const f32x4 vv = f32x4{v, v, v, v};
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += stride)
    reinterpret_cast<f32x4 *>(p)[e] = vv;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[4 * n4 + threadIdx.x] = v;
}
// f_greedy_match with matched == 0 (modellib.py:365-379): match = (score == max) / count.
__global__ __launch_bounds__(64) void greedy_match_kernel(const float *score, int T, float *match) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float mx = -__builtin_inff();
  for (int t = lane; t < T; t += 64) mx = fmaxf(mx, score[(size_t)b * T + t]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float cnt = 0.f;
  for (int t = lane; t < T; t += 64) cnt += score[(size_t)b * T + t] == mx ? 1.f : 0.f;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  for (int t = lane; t < T; t += 64) match[(size_t)b * T + t] = (score[(size_t)b * T + t] == mx ? 1.f : 0.f) / cnt;
}
// out[i] = idx[i] >= 0 ? src[idx[i]] : 0
__global__ __launch_bounds__(256) void gather_kernel(const float *src, const int *idx, size_t n, float *out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int j = idx[i];
    out[i] = j >= 0 ? src[j] : 0.f;
  }
}
}  // namespace
}  // namespace ra

extern "C" int ra_gather_f32(const float *src, const int *idx, size_t n, float *out, void *stream) {
  if (!src || !idx || !out) return ra::fail(RA_E_INVALID, "ra_gather_f32: null pointer");
  if (n == 0) return 0;
  size_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ra::gather_kernel, dim3((unsigned)grid), dim3(256), 0, ra::as_stream(stream), src, idx, n, out);
  return ra::launch_status("ra_gather_f32");
}

extern "C" int ra_fill_f32(float *p, size_t n, float value, void *stream) {
  if (!p || (reinterpret_cast<uintptr_t>(p) & 15)) return ra::fail(RA_E_INVALID, "ra_fill_f32: null / unaligned pointer");
  if (n == 0) return 0;
  const size_t n4 = n / 4;
  size_t grid = (n4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (grid == 0) grid = 1;
  hipLaunchKernelGGL(ra::fill_kernel, dim3((unsigned)grid), dim3(256), 0, ra::as_stream(stream), p, n4, n, value);
  return ra::launch_status("ra_fill_f32");
}

extern "C" int ra_greedy_match_f32(const float *score, int B, int T, float *match, void *stream) {
  if (!score || !match || B <= 0 || T <= 0) return ra::fail(RA_E_INVALID, "ra_greedy_match_f32: bad argument");
  hipLaunchKernelGGL(ra::greedy_match_kernel, dim3(B), dim3(64), 0, ra::as_stream(stream), score, T, match);
  return ra::launch_status("ra_greedy_match_f32");
}
