// f32 MFMA fed by wide LDS reads, like the conv main loop (tuning aid, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PIX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, int iters, float b0) {
  __shared__ __attribute__((aligned(16))) float tile[18 * 34 * 20];
  for (int i = threadIdx.x; i < 18 * 34 * 20; i += 64 * WAVES) tile[i] = (float)(i & 7);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, ksub = lane >> 4, q = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
  const int base = (((wave & 3) * 4 + dy) * 34 + 2 * q + dx) * PIX + ksub * 4;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 av[8];
#pragma unroll
      for (int g = 0; g < 8; ++g)
        av[g] = *reinterpret_cast<const f32x4 *>(&tile[base + ((2 * (g / 4) + tap / 3) * 34 + 8 * (g % 4) + tap % 3) * PIX]);
#pragma unroll
      for (int cg = 0; cg < 4; ++cg)
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][cg], b, acc[g], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}
template <int PIX, int WAVES>
void run(int wgs, int iters) {
  float *out;
  hipMalloc(&out, wgs * 64 * WAVES * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<PIX, WAVES><<<wgs, 64 * WAVES>>>(out, 2, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<PIX, WAVES><<<wgs, 64 * WAVES>>>(out, iters, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)wgs * WAVES * iters * 9 * 32 * 2048.0;
  printf("PIX=%d waves/WG=%d wgs=%d: %.3f ms  %.1f TFLOP/s\n", PIX, WAVES, wgs, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<16, 4>(256, 200);
  run<20, 4>(256, 200);
  run<16, 8>(256, 100);
  run<16, 4>(512, 100);
  return 0;
}
