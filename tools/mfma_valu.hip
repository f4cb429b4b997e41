// Do MFMA and VALU work overlap on a gfx950 SIMD?  (tuning aid, not product)
//   mode 0: MFMA only   mode 1: VALU only   mode 2: both, interleaved in every wave
//   mode 3: both, even workgroups MFMA-only / odd workgroups VALU-only (2 WGs per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 * i + threadIdx.x;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && !(blockIdx.x & 1));
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (do_m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], b, a);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NV>
void run(int wgs, int iters) {
  float *out;
  hipMalloc(&out, wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE, NV><<<wgs, 256>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NV><<<wgs, 256>>>(out, iters, 1.f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d NV=%d wgs=%d: %.3f ms  (per iter per wave: %d MFMA = %d pipe cycles, %d VALU)\n", MODE, NV,
         wgs, ms, 32, 32 * 32, 8 * NV);
  hipFree(out);
}
int main() {
  const int it = 2000;
  for (int wgs : {256, 512, 1024}) {
    run<0, 16>(wgs, it);
    run<1, 16>(wgs, it);
    run<2, 16>(wgs, it);
    run<3, 16>(wgs, it);
    run<1, 64>(wgs, it);
    run<2, 64>(wgs, it);
    run<3, 64>(wgs, it);
  }
  return 0;
}
