// Rate and lane layout of v_mfma_f32_4x4x1_16B_f32 on gfx950 (tuning aid, not product): the
// multi-block form has N = 4 output columns per block, so layers with 8 output channels would
// not pad half an MFMA tile with zeros the way the 16x16x4 form does.
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
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void layout(float *out) {  // one wave: A = lane id, B = 1 -> D tells which A lanes feed each D entry
  const int lane = threadIdx.x;
  f32x4 acc = f32x4{0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.0f, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
  acc = f32x4{0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)lane, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = acc[r];
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
  double flops = (double)wgs * 4 * iters * 8 * NACC * 512.0;  // 16 blocks x 4x4x1 x 2
  printf("4x4x1 NACC=%d wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs, iters, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(256, 4000);
  run<8>(256, 4000);
  run<8>(1024, 4000);
  float *d, h[512];
  hipMalloc(&d, 512 * 4);
  layout<<<1, 64>>>(d);
  hipMemcpy(h, d, 512 * 4, hipMemcpyDeviceToHost);
  printf("D[lane][r] with A = lane, B = 1 (which A lane feeds row):\n");
  for (int l = 0; l < 64; l += 1) printf("%s%g,%g,%g,%g", l % 8 ? "  " : "\n", h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  printf("\nD[lane][r] with A = 1, B = lane (which B lane feeds column):\n");
  for (int l = 0; l < 64; l += 1) printf("%s%g,%g,%g,%g", l % 8 ? "  " : "\n", h[256 + l * 4], h[256 + l * 4 + 1], h[256 + l * 4 + 2], h[256 + l * 4 + 3]);
  printf("\n");
  return 0;
}
