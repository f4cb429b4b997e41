// Round 5 probe: could the controller CNN's float32 contractions run on the bf16 matrix pipe at float32 accuracy?
// (tuning aid, not product:  hipcc -O3 --offload-arch=gfx950 tools/mfma_split_probe.hip -o tools/bin/mfma_split_probe)
//
// A float32 value splits exactly into three bf16 pieces  a = a_h + a_m + a_l  (8 + 8 + 8 mantissa bits), and a product
// a * b = sum of nine piece products, of which six carry everything above 2^-24 of it:  hh, hm, mh, hl, lh, mm.  Products of
// two bf16 numbers are exact in float32, so six v_mfma_f32_16x16x32_bf16 (K = 32 each) replace the eight
// v_mfma_f32_16x16x4_f32 (K = 4 each) of one K = 32 block — at float32 accuracy, IF the dropped terms and the float32
// accumulation behave.  Three questions, answered by measurement:
//   (1) rate: cycles of a SIMD per K = 32 block and 16 x 16 tile, f32 form (8 MFMAs) against split forms (3, 4, 6 MFMAs);
//   (2) overlap: DESIGN.md §4 found that f32 MFMAs and FP32 VALU instructions of a SIMD do NOT overlap (same lanes).  Do bf16
//       MFMAs overlap with VALU work (the staging / transform / epilogue instructions every conv kernel here is made of)?
//   (3) accuracy: K = 576 dot products (the controller CNN's widest, 64 channels x 9 taps) of N(0,1) operands against
//       float64: float32 MFMA chain, 2-piece split with 3 and 4 products, 3-piece split with 6 products.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short bf16_rne(float v) {  // round to nearest even, as v_cvt_pk_bf16_f32
  unsigned u = __builtin_bit_cast(unsigned, v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

// ---------------------------------------------------------------------------------------------- (1) + (2) rate / overlap
// MODE 0: f32 MFMAs only (8 per block)   1: bf16 MFMAs only (NB per block)   2: VALU only (NV FMAs per block)
//      3: f32 MFMAs + VALU               4: bf16 MFMAs + VALU   — interleaved in every wave
template <int MODE, int NB, int NV>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float a0) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-3f, b = a0 * 0.5f;
  s16x8 pa, pb;
  for (int i = 0; i < 8; ++i) {
    pa[i] = (short)bf16_rne(a + i);
    pb[i] = (short)bf16_rne(b - i);
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 * i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {  // four K = 32 blocks per iteration, each on its own accumulator tile
      if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[blk], 0, 0, 0);
      }
      if (MODE == 1 || MODE == 4) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
          acc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa), __builtin_bit_cast(bf16x8, pb), acc[blk], 0, 0, 0);
      }
      if (MODE >= 2) {
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

template <int MODE, int NB, int NV>
double run_rate(int wgs, int iters, const char *what) {
  float *out;
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  rate_kernel<MODE, NB, NV><<<wgs, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<MODE, NB, NV><<<wgs, 256>>>(out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // blocks per SIMD: wgs * 4 waves / 1024 SIMDs (256 CUs x 4), 4 blocks per iteration and wave
  const double blocks_per_simd = (double)wgs * 4 / 1024.0 * 4.0 * iters;
  const double ns_per_block = ms * 1e6 / blocks_per_simd;
  printf("  %-44s wgs=%4d  %.3f ms   %.1f ns of a SIMD per K=32 block and 16x16 tile\n", what, wgs, ms, ns_per_block);
  hipFree(out);
  return ns_per_block;
}

// ---------------------------------------------------------------------------------------------- (3) accuracy
// One wave computes D[16,16] = A[16,K] B[K,16] four ways.  A, B float32 in global memory (row-major A [16][K], B [K][16]).
// Lane (m = lane & 15, kb = lane >> 4): f32 form k-step s uses A[m][4 s + kb], B[4 s + kb][n = m];  bf16 form block t uses
// A[m][32 t + 8 kb + j], B[32 t + 8 kb + j][n], j = 0..7.  D: lane holds rows 4 kb + r, column m.
template <int WHICH>  // 0: f32 chain  1: 2 pieces, hh + hl + lh  2: 2 pieces, + ll  3: 3 pieces, six products
__global__ __launch_bounds__(64) void acc_kernel(const float *A, const float *B, int K, float *D) {
  const int lane = threadIdx.x, m = lane & 15, kb = lane >> 4;
  f32x4 acc = f32x4{0, 0, 0, 0};
  if (WHICH == 0) {
    for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * K + 4 * s + kb], B[(4 * s + kb) * 16 + m], acc, 0, 0, 0);
  } else {
    for (int t = 0; t < K / 32; ++t) {
      s16x8 a[3], b[3];
      for (int j = 0; j < 8; ++j) {
        float av = A[m * K + 32 * t + 8 * kb + j], bv = B[(32 * t + 8 * kb + j) * 16 + m];
        for (int p = 0; p < 3; ++p) {
          const unsigned short ah = bf16_rne(av), bh = bf16_rne(bv);
          a[p][j] = (short)ah;
          b[p][j] = (short)bh;
          av -= bf16_f(ah);  // exact: the difference of a float and its bf16 rounding is a float
          bv -= bf16_f(bh);
        }
      }
      auto mm = [&](int p, int q) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[p]), __builtin_bit_cast(bf16x8, b[q]), acc, 0, 0, 0);
      };
      // smallest terms first: they are added to the running sum of the previous blocks anyway, but within a block the
      // order costs nothing
      if (WHICH == 3) {
        mm(1, 1);
        mm(0, 2);
        mm(2, 0);
      }
      if (WHICH == 2) mm(1, 1);
      mm(0, 1);
      mm(1, 0);
      mm(0, 0);
    }
  }
  for (int r = 0; r < 4; ++r) D[(4 * kb + r) * 16 + m] = acc[r];
}

int main() {
  const int it = 4000;
  printf("(1) rate, MFMAs only (one wave per SIMD at 256 workgroups, two at 512, four at 1024):\n");
  for (int wgs : {256, 512, 1024}) {
    run_rate<0, 0, 0>(wgs, it, "f32   8 x v_mfma_f32_16x16x4_f32");
    run_rate<1, 3, 0>(wgs, it, "bf16  3 x v_mfma_f32_16x16x32_bf16 (2 pieces)");
    run_rate<1, 4, 0>(wgs, it, "bf16  4 x v_mfma_f32_16x16x32_bf16 (2 pieces + ll)");
    run_rate<1, 6, 0>(wgs, it, "bf16  6 x v_mfma_f32_16x16x32_bf16 (3 pieces)");
  }
  printf("(2) overlap with FP32 VALU work (NV dependent-free FMAs per K=32 block in the same wave):\n");
  for (int wgs : {256, 1024}) {
    run_rate<2, 0, 32>(wgs, it, "VALU only, 32 FMAs per block");
    run_rate<3, 0, 32>(wgs, it, "f32 MFMAs (8) + 32 FMAs");
    run_rate<4, 6, 32>(wgs, it, "bf16 MFMAs (6) + 32 FMAs");
    run_rate<2, 0, 64>(wgs, it, "VALU only, 64 FMAs per block");
    run_rate<3, 0, 64>(wgs, it, "f32 MFMAs (8) + 64 FMAs");
    run_rate<4, 6, 64>(wgs, it, "bf16 MFMAs (6) + 64 FMAs");
    run_rate<4, 3, 64>(wgs, it, "bf16 MFMAs (3) + 64 FMAs");
  }
  printf("(3) accuracy, K = 576 dot products of N(0,1) operands, 16 x 16 outputs x 64 trials, against float64:\n");
  const int K = 576, trials = 64;
  std::vector<float> hA(16 * K), hB(K * 16), hD(256);
  float *dA, *dB, *dD;
  hipMalloc(&dA, hA.size() * 4);
  hipMalloc(&dB, hB.size() * 4);
  hipMalloc(&dD, 256 * 4);
  const char *names[4] = {"float32 MFMA chain (what ships)", "2 pieces, 3 products (hh + hl + lh)", "2 pieces, 4 products (+ ll)",
                          "3 pieces, 6 products (hh hm mh hl lh mm)"};
  double worst[4] = {0, 0, 0, 0}, rms[4] = {0, 0, 0, 0};
  srand(12345);
  auto gauss = []() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
  };
  for (int tr = 0; tr < trials; ++tr) {
    for (auto &x : hA) x = gauss();
    for (auto &x : hB) x = gauss();
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256, 0.0), scale(256, 0.0);
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n)
        for (int k = 0; k < K; ++k) {
          ref[m * 16 + n] += (double)hA[m * K + k] * hB[k * 16 + n];
          scale[m * 16 + n] += fabs((double)hA[m * K + k] * hB[k * 16 + n]);
        }
    for (int w = 0; w < 4; ++w) {
      if (w == 0) acc_kernel<0><<<1, 64>>>(dA, dB, K, dD);
      if (w == 1) acc_kernel<1><<<1, 64>>>(dA, dB, K, dD);
      if (w == 2) acc_kernel<2><<<1, 64>>>(dA, dB, K, dD);
      if (w == 3) acc_kernel<3><<<1, 64>>>(dA, dB, K, dD);
      hipMemcpy(hD.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
      for (int e = 0; e < 256; ++e) {
        const double err = fabs(hD[e] - ref[e]) / scale[e];  // relative to sum |a b|: the natural scale of a dot product's error
        worst[w] = err > worst[w] ? err : worst[w];
        rms[w] += err * err;
      }
    }
  }
  for (int w = 0; w < 4; ++w) printf("  %-44s max |err| / sum|ab| = %.2e   rms = %.2e\n", names[w], worst[w], sqrt(rms[w] / (256.0 * trials)));
  return 0;
}
