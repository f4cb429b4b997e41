// Issue rate of the float32 MFMA forms on one SIMD: N back-to-back instructions on 4 independent accumulators, one wave per
// SIMD, s_memtime around the loop.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int FORM>
__global__ void k(float *out, long long *cyc, int n) {
  f32x4 acc[4] = {};
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (FORM == 0) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[FORM] = t1 - t0;
}
int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  const int n = 4096;
  for (int waves = 1; waves <= 2; ++waves) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, n);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%d wave(s) per SIMD: 4x4x1_16B %.1f shader cycles per MFMA per wave, 16x16x4 %.1f (s_memrealtime-free: __builtin_readcyclecounter)\n",
           waves, (double)h[0] / (4.0 * n), (double)h[1] / (4.0 * n));
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form = 0; form < 2; ++form) {
    hipEventRecord(e0);
    if (form == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, out, cyc, n * 8);
    else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, out, cyc, n * 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (form == 0 ? 512.0 : 2048.0) * 4.0 * n * 8 * 1024 * 4;
    printf("%s: %.3f ms, %.1f TFLOP/s chip\n", form == 0 ? "4x4x1_16B" : "16x16x4  ", ms, flops / ms * 1e-9);
  }
  return 0;
}
