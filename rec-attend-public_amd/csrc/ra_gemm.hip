// The controller's parameter gradients of a training step: C += A^T B over the K rows of two row-major matrices —
// A [K, M] holds a layer's inputs and B [K, N] the gradients of its pre-activations for every (timestep, image, glimpse
// iteration) of the step (K = T * B * iters, a few hundred rows), C is the weight gradient [M, N] in the flat gradient
// bucket, and the bias gradient (the column sums of B) rides along as one more output row (A's virtual column of ones).
// These are short-K, wide-output products: rocBLAS picks 256 x 256 macro-tiles for them (8 workgroups, 126-151 us for the
// LSTM's [320, 1024]); here a workgroup owns a 16 x 32 tile of C, its four waves split K and meet in LDS in a fixed order.
// The output may be SEGMENTED: rows [0, row_split) and [row_split, M) and column blocks of col_block go to separate
// tensors (the LSTM's eight weight and four bias parameters) through a device table of pointers — no scatter passes.
#include "ra_common.h"

namespace ra {
namespace gemm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Out {
  float *C;                   // plain output [M, N], row stride ldc (seg == nullptr)
  int ldc;
  float *bias;                // plain bias [N] or nullptr
  float *const *seg;          // segmented: seg[r * ncb + j], r = 0 top rows, 1 bottom rows, 2 bias; entries may be null
  int row_split, col_block, ncb;
};

__global__ __launch_bounds__(256) void gemm_tn_acc_kernel(const float *A, int lda, const float *B, int ldb, int K, int M, int N,
                                                          int k_period, Out o) {
  __shared__ f32x4 red[3][64][2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 32;
  const int mi = m0 + (lane & 15), kk = lane >> 4;
  const int nj0 = n0 + (lane & 15), nj1 = nj0 + 16;
  // this wave's quarter of K, in 4-row steps
  const int steps = (K + 3) / 4, per = (steps + 3) / 4, s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  auto load = [&](int s, float &a, float &b0, float &b1) {
    const int k = 4 * s + kk;
    const bool ok = k < K && !(k_period && (k % k_period) == k_period - 1);
    a = ok ? (mi < M ? A[(size_t)k * lda + mi] : (mi == M ? 1.f : 0.f)) : 0.f;
    b0 = (ok && nj0 < N) ? B[(size_t)k * ldb + nj0] : 0.f;
    b1 = (ok && nj1 < N) ? B[(size_t)k * ldb + nj1] : 0.f;
  };
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    float a[8], b0[8], b1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) load(s + u, a[u], b0[u], b1[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b1[u], acc1, 0, 0, 0);
    }
  }
  for (; s < s1; ++s) {
    float a, b0, b1;
    load(s, a, b0, b1);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
  }
  if (wave) red[wave - 1][lane][0] = acc0, red[wave - 1][lane][1] = acc1;
  __syncthreads();
  if (wave) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) acc0 += red[w][lane][0], acc1 += red[w][lane][1];
  // accumulator r of lane l: row m0 + 4 * (l / 16) + r, column n0 (+16) + l % 16
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = j ? nj1 : nj0;
    if (n >= N) continue;
    const f32x4 v = j ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * kk + r;
      if (m > M || (m == M && !o.seg && !o.bias)) continue;
      float *p;
      if (o.seg) {
        const int cb = n / o.col_block, nc = n - cb * o.col_block;
        if (m == M) {
          p = o.seg[2 * o.ncb + cb];
          if (p) p += nc;
        } else {
          const int rs = m >= o.row_split;
          p = o.seg[rs * o.ncb + cb];
          if (p) p += (size_t)(m - rs * o.row_split) * o.col_block + nc;
        }
      } else {
        p = m == M ? o.bias + n : o.C + (size_t)m * o.ldc + n;
      }
      if (p) *p += v[r];
    }
  }
}

}  // namespace gemm
}  // namespace ra

// C [M, N] += A^T B (+ bias [N] += column sums of B) over rows k in [0, K) of A [K, M] / B [K, N], skipping the rows with
// k % k_period == k_period - 1 when k_period > 0 (the last glimpse iteration of every image has no successor).
// seg == NULL: C with row stride ldc, bias or NULL.  seg != NULL (a DEVICE table of 3 * ceil(N / col_block) pointers:
// [top rows | bottom rows | bias] x column block; NULL entries are skipped): rows [0, row_split) / [row_split, M) and column
// blocks of col_block land in separate dense tensors [rows, col_block]; row_split % 16 == 0 and col_block % 16 == 0.
extern "C" int ra_gemm_tn_acc_f32(const float *A, int lda, const float *B, int ldb, int K, int M, int N, int k_period, float *C, int ldc,
                                  float *bias, const void *const *seg, int row_split, int col_block, void *stream) {
  using namespace ra;
  if (!A || !B || (!C && !seg) || K <= 0 || M <= 0 || N <= 0 || lda < M || ldb < N || k_period < 0)
    return fail(RA_E_INVALID, "ra_gemm_tn_acc_f32: bad argument");
  gemm::Out o{};
  if (seg) {
    if (row_split < 0 || row_split > M || (row_split & 15) || col_block <= 0 || (col_block & 15))
      return fail(RA_E_SHAPE, "ra_gemm_tn_acc_f32: segmented output needs row_split %% 16 == 0 and col_block %% 16 == 0 (%d, %d)", row_split,
                  col_block);
    o.seg = reinterpret_cast<float *const *>(const_cast<void *const *>(seg));
    o.row_split = row_split, o.col_block = col_block, o.ncb = ceil_div(N, col_block);
  } else {
    if (ldc < N) return fail(RA_E_INVALID, "ra_gemm_tn_acc_f32: ldc < N");
    o.C = C, o.ldc = ldc, o.bias = bias;
  }
  const int rows = M + ((seg || bias) ? 1 : 0);
  hipLaunchKernelGGL(gemm::gemm_tn_acc_kernel, dim3(ceil_div(N, 32), ceil_div(rows, 16)), dim3(256), 0, as_stream(stream), A, lda, B, ldb, K,
                     M, N, k_period, o);
  return launch_status("ra_gemm_tn_acc_f32");
}
