// Latency / issue-rate numbers a latency-bound kernel is made of, measured the way such a kernel meets
// them: a fresh launch (cold L2 after the kernel boundary), 256-1024 workgroups, thread 0 of each
// workgroup stamping the 100 MHz wall clock between steps.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the stamp is tied to the value it must come after (the compiler is free to move a plain clock read)
#define STAMP(k, dep) asm volatile("s_nop 0\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T[k]) : "v"(dep) : "memory")

__global__ __launch_bounds__(256) void probe(const float *__restrict__ rec, const float *__restrict__ big, float *__restrict__ out,
                                             long long *__restrict__ st, int stride) {
  const int t = threadIdx.x, wg = blockIdx.x;
  __shared__ float lds[4096];
  long long T[10];
  STAMP(0, t);
  const float r0 = rec[wg & 7];  // scalar load (uniform address)
  float acc = r0;
  if (acc > 1e30f) acc = 0;
  STAMP(1, acc);
  // one vector load per thread, cold
  const float *p = big + (size_t)wg * stride + t * 4;
  f32x4 v = *reinterpret_cast<const f32x4 *>(p);
  acc += v.x + v.y + v.z + v.w;
  STAMP(2, acc);
  // 16 independent 16-byte loads per thread (64 KB per workgroup)
  f32x4 xs[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) xs[k] = *reinterpret_cast<const f32x4 *>(p + 4096 + k * 1024);
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += xs[k].x + xs[k].w;
  STAMP(3, acc);
  // 256 dependent FMAs
#pragma unroll
  for (int k = 0; k < 256; ++k) acc = fmaf(acc, 1.0001f, 0.5f);
  STAMP(4, acc);
  // 64 dependent v_exp
#pragma unroll
  for (int k = 0; k < 64; ++k) acc = __expf(acc * 1e-3f);
  STAMP(5, acc);
  // 16 dependent LDS round trips
  lds[t] = acc;
  __syncthreads();
  int idx = (t * 7) & 255;
#pragma unroll
  for (int k = 0; k < 16; ++k) idx = ((int)lds[idx] + idx + k) & 255;
  acc += idx;
  STAMP(6, acc);
  // 8 barriers
#pragma unroll
  for (int k = 0; k < 8; ++k) __syncthreads();
  STAMP(7, acc);
  out[(size_t)wg * 256 + t] = acc;
  if (t == 0)
    for (int k = 0; k < 8; ++k) st[(size_t)wg * 8 + k] = T[k];
}

int main(int argc, char **argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256, stride = 1 << 16;
  float *rec, *big, *out;
  long long *st;
  CK(hipMalloc(&rec, 64));
  CK(hipMalloc(&big, (size_t)nwg * stride * 4 + (1 << 22)));
  CK(hipMalloc(&out, (size_t)nwg * 256 * 4));
  CK(hipMalloc(&st, (size_t)nwg * 64));
  CK(hipMemset(rec, 0, 64));
  CK(hipMemset(big, 0, (size_t)nwg * stride * 4 + (1 << 22)));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, rec, big, out, st, stride);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> h((size_t)nwg * 8);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  const char *names[] = {"start", "scalar load (1)", "vector load, cold (1 x 16 B / thread)", "16 x 16 B loads / thread (64 KB / wg)",
                         "256 dependent FMAs", "64 dependent v_exp (+mul)", "store+barrier+16 dependent LDS reads", "8 barriers"};
  long long t0 = h[0];
  for (int w = 0; w < nwg; ++w) t0 = std::min(t0, h[(size_t)w * 8]);
  printf("%d workgroups of 256 threads\n", nwg);
  for (int k = 0; k < 8; ++k) {
    std::vector<double> v;
    for (int w = 0; w < nwg; ++w) v.push_back(k == 0 ? (h[(size_t)w * 8] - t0) * 0.01 : (h[(size_t)w * 8 + k] - h[(size_t)w * 8 + k - 1]) * 0.01);
    std::sort(v.begin(), v.end());
    printf("  %-42s min %.2f  median %.2f  max %.2f us\n", names[k], v.front(), v[v.size() / 2], v.back());
  }
  return 0;
}
