// Sustained-clock ceiling of v_mfma_f32_16x16x4_f32 on this chip (tuning aid, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs, int iters) {
  float *out;
  hipMalloc(&out, wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NACC><<<wgs, 256>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<wgs, 256>>>(out, iters, 1.f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)wgs * 4 * iters * 8 * NACC * 2048.0;
  printf("NACC=%d wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs, iters, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, 2000);
  run<8>(256, 1000);
  run<8>(512, 1000);
  run<8>(256, 100);   // ~30 us kernel, like a conv layer
  run<8>(256, 50);
  run<8>(1024, 2000);
  return 0;
}
