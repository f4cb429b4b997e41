// Training-step kernels (full_model.py:1039-1057 and the backward passes they need).
//
// ra_adam_step_f32 — the reference's optimizer on ONE flat float32 bucket:
//   gvs = optimizer.compute_gradients(total_loss); grad = clip_by_value(grad, -1, 1);
//   tf.train.AdamOptimizer(learn_rate, epsilon=1e-7).apply_gradients          full_model.py:1048-1056
// with the weight-decay term wd * l2_loss(w) of nnlib.weight_variable (nnlib.py:59-61), which is
// part of total_loss and therefore of the clipped gradient, and the data-parallel mean
// (grad_scale = 1 / world after the RCCL sum) folded in — one pass over five arrays, HBM-bound.
// TF's Adam: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
//            p -= lr_t * m / (sqrt(v) + eps)        (epsilon outside the bias correction).
#include <type_traits>

#include "ra_common.h"

namespace ra {
namespace train {
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_kernel(float *p, const float *g, float *m, float *v, const float *wd,
                                                   size_t n, float lr_t, float b1, float b2, float eps, float clip,
                                                   float gscale, const int *status, int n_solver, int n_other) {
  // the guarded form: a NEGATIVE solver status of this step (a matching that hit one of the reference's LOG(FATAL) caps; 1 =
  // the outer cap, where the reference logs and carries on with the partial matching, hungarian.cc:363-377) or any NON-ZERO
  // other word (a controller workgroup that timed out, another rank's failure flag) and the update is NOT applied —
  // parameters and moments stay as they are.  The words are uniform across the grid (scalar loads), written by launches
  // earlier on the stream.
  for (int k = 0; k < n_solver; ++k)
    if (status[k] < 0) return;
  for (int k = 0; k < n_other; ++k)
    if (status[n_solver + k] != 0) return;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float pi = p[i];
    float gi = g[i] * gscale + (wd ? wd[i] * pi : 0.f);
    gi = fminf(fmaxf(gi, -clip), clip);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
  }
}
// =================================================================================================
// Fast forms of the BatchNorm elementwise / reduction passes for C % 4 == 0 with C / 4 a power of two
// (every layer of the three CNNs except 1- or 3-channel ends): a thread owns FOUR channels of one
// pooling window — float4 loads, 32-bit indices, the window's (up to 4) pixels read once instead of
// once per pixel, the per-channel constants hoisted (the grid stride is a multiple of C / 4, so a
// thread's channel group never changes).  The generic kernels above moved 0.5 TB/s.
// Summation order is fixed (no atomics): bit-reproducible.
struct BnConst {
  f32x4 mu, g, be, rstd;
};
// Storage of a channel quad: float32 (16 bytes) or, in the bf16 mode's tensors between the conv layers' passes
// (model_opt['compute_dtype'] = 'bf16'), bf16 (8 bytes; loads are exact, stores round to nearest even as v_cvt_pk_bf16_f32).
typedef unsigned u32x2q __attribute__((ext_vector_type(2)));
template <bool BF>
struct Q4 {
  typedef f32x4 T;
  static __device__ inline f32x4 ld(const T *p, size_t i) { return p[i]; }
  static __device__ inline void st(T *p, size_t i, const f32x4 v) { p[i] = v; }
};
template <>
struct Q4<true> {
  typedef u32x2q T;
  static __device__ inline f32x4 ld(const T *p, size_t i) {
    const u32x2q q = p[i];
    return f32x4{__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xffff0000u),
                 __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xffff0000u)};
  }
  static __device__ inline void st(T *p, size_t i, const f32x4 v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    p[i] = u32x2q{__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, bf16x2)),
                  __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, bf16x2))};
  }
};
__device__ inline BnConst bn_const(const float *mean, const float *var, const float *gamma, const float *beta, float eps,
                                   int c0) {
  BnConst k;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float rstd = var ? rsqrtf(var[c0 + i] + eps) : 1.f;
    k.rstd[i] = rstd;
    k.g[i] = (gamma ? gamma[c0 + i] : 1.f) * rstd;
    k.mu[i] = mean ? mean[c0 + i] : 0.f;
    k.be[i] = beta ? beta[c0 + i] : 0.f;
  }
  return k;
}
// sum over the threads of a workgroup that share (tid % C4); valid in threads tid < C4.  C4 = 1 << lg <= 64.
__device__ inline float sum_by_group(float v, int C4, float *red) {
  for (int off = 32; off >= C4; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane < C4) red[wave * 64 + lane] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x < C4) t = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
  return t;
}

// per-channel sum (mean == nullptr) or sum of squared deviations over u [n4 = npix * C4] float4s
__global__ __launch_bounds__(256) void chan_sum_v4_kernel(const f32x4 *u, int n4, int C4, const float *mean, float *part) {
  __shared__ float red[256];
  const int tid = threadIdx.x, cg = tid & (C4 - 1);
  f32x4 mu = f32x4{0.f, 0.f, 0.f, 0.f};
  if (mean)
    for (int i = 0; i < 4; ++i) mu[i] = mean[4 * cg + i];
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int e = blockIdx.x * 256 + tid; e < n4; e += gridDim.x * 256) {
    const f32x4 v = u[e] - mu;
    s += mean ? v * v : v;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = sum_by_group(s[i], C4, red);
    if (tid < C4) part[(size_t)blockIdx.x * 4 * C4 + 4 * tid + i] = t;
  }
}

// the window of one thread: POOL x POOL pixels x 4 channels
template <int POOL, bool UB = false>
__device__ inline void load_window(const typename Q4<UB>::T *u, int b, int yo, int xo, int H, int W, int C4, int cg,
                                   f32x4 (&w)[POOL * POOL]) {
#pragma unroll
  for (int k = 0; k < POOL * POOL; ++k)
    w[k] = Q4<UB>::ld(u, ((b * H + yo * POOL + (k / POOL)) * W + xo * POOL + (k % POOL)) * C4 + cg);
}

template <int POOL>
constexpr int kRowsPerIter = POOL == 1 ? 4 : 1;  // output rows a thread of the float4 BatchNorm kernels handles per loop iteration

template <int POOL, bool UB = false, bool YB = false>
__global__ __launch_bounds__(256) void bn_act_pool_v4_kernel(const typename Q4<UB>::T *u, const float *mean, const float *var,
                                                             const float *gamma, const float *beta, float eps, int relu,
                                                             int B, int H, int W, int C4, int lg, typename Q4<YB>::T *y) {
  const int Ho = H / POOL, Wo = W / POOL;
  const int er = blockIdx.x * 256 + threadIdx.x;
  if (er >= Wo * C4) return;
  const int xo = er >> lg, cg = er & (C4 - 1);
  const BnConst k = bn_const(mean, var, gamma, beta, eps, 4 * cg);
  const float lo = relu ? 0.f : -__builtin_inff();
  // kRowsPerIter<POOL> rows per iteration, every load issued before the first use: a thread of the unpooled form moved 16
  // bytes per round trip (3.5 TB/s on the full-resolution layers; the pooled form's four loads per thread ran at 5.7)
  constexpr int RU = kRowsPerIter<POOL>;
  const int rows = B * Ho;
  for (int row0 = blockIdx.y * RU; row0 < rows; row0 += gridDim.y * RU) {
    f32x4 w[RU][POOL * POOL];
#pragma unroll
    for (int r = 0; r < RU; ++r)
      if (row0 + r < rows) {
        const int b = (row0 + r) / Ho, yo = (row0 + r) - b * Ho;
        load_window<POOL, UB>(u, b, yo, xo, H, W, C4, cg, w[r]);
      }
#pragma unroll
    for (int r = 0; r < RU; ++r)
      if (row0 + r < rows) {
        f32x4 best;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float m = -__builtin_inff();
#pragma unroll
          for (int q = 0; q < POOL * POOL; ++q) m = fmaxf(m, fmaxf((w[r][q][i] - k.mu[i]) * k.g[i] + k.be[i], lo));
          best[i] = m;
        }
        Q4<YB>::st(y, ((row0 + r) * Wo + xo) * C4 + cg, best);
      }
  }
}

// dv of every pixel of the window (dy routed to the FIRST maximum, masked by the ReLU) and xhat
template <int POOL>
__device__ inline void window_grad(const f32x4 (&w)[POOL * POOL], const f32x4 dyv, const BnConst &k, float lo, int relu,
                                   f32x4 (&dv)[POOL * POOL], f32x4 (&xh)[POOL * POOL]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float best = -__builtin_inff();
    int arg = 0;
    float v[POOL * POOL];
#pragma unroll
    for (int q = 0; q < POOL * POOL; ++q) {
      xh[q][i] = (w[q][i] - k.mu[i]) * k.rstd[i];
      v[q] = (w[q][i] - k.mu[i]) * k.g[i] + k.be[i];
      const float a = fmaxf(v[q], lo);
      if (a > best) {
        best = a;
        arg = q;
      }
    }
#pragma unroll
    for (int q = 0; q < POOL * POOL; ++q) dv[q][i] = (q == arg && !(relu && v[q] <= 0.f)) ? dyv[i] : 0.f;
  }
}

template <int POOL, bool UB = false, bool DB = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_v4_kernel(const typename Q4<UB>::T *u, const typename Q4<DB>::T *dy, const float *mean,
                                                               const float *var, const float *gamma, const float *beta,
                                                               float eps, int relu, int B, int H, int W, int C4, int lg,
                                                               float *part, const float *const *tabs = nullptr, int G = 1) {
  __shared__ float red[256];
  const int Ho = H / POOL, Wo = W / POOL;
  if (tabs) {  // group blockIdx.z of G calls of the layer stacked along the batch: its own statistics and parameters
    const int g = blockIdx.z;
    mean = tabs[g], var = tabs[G + g], gamma = tabs[2 * G + g], beta = tabs[3 * G + g];
    u += (size_t)g * B * H * W * C4;
    dy += (size_t)g * B * Ho * Wo * C4;
    part += (size_t)g * gridDim.x * gridDim.y * 2 * 4 * C4;
  }
  const int er = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x;
  const bool live = er < Wo * C4;
  const int xo = er >> lg, cg = er & (C4 - 1);
  f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (live) {
    const BnConst k = bn_const(mean, var, gamma, beta, eps, 4 * cg);
    const float lo = relu ? 0.f : -__builtin_inff();
    for (int row = blockIdx.y; row < B * Ho; row += gridDim.y) {
      const int b = row / Ho, yo = row - b * Ho;
      f32x4 w[POOL * POOL], dv[POOL * POOL], xh[POOL * POOL];
      load_window<POOL, UB>(u, b, yo, xo, H, W, C4, cg, w);
      window_grad<POOL>(w, Q4<DB>::ld(dy, (row * Wo + xo) * C4 + cg), k, lo, relu, dv, xh);
#pragma unroll
      for (int q = 0; q < POOL * POOL; ++q) {
        s0 += dv[q];
        s1 += dv[q] * xh[q];
      }
    }
  }
  const int blk = blockIdx.y * gridDim.x + blockIdx.x, C = 4 * C4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t0 = sum_by_group(s0[i], C4, red), t1 = sum_by_group(s1[i], C4, red);
    if (tid < C4) {
      part[((size_t)blk * 2) * C + 4 * tid + i] = t0;
      part[((size_t)blk * 2 + 1) * C + 4 * tid + i] = t1;
    }
  }
}

template <int POOL, bool UB = false, bool DB = false>
__global__ __launch_bounds__(256) void bn_bwd_dx_v4_kernel(const typename Q4<UB>::T *u, const typename Q4<DB>::T *dy, const float *mean,
                                                           const float *var, const float *gamma, const float *beta,
                                                           const float *dbeta, const float *dgamma, float eps, int relu,
                                                           int B, int H, int W, int C4, int lg, typename Q4<UB>::T *du, float inv_n,
                                                           const float *const *tabs = nullptr, int G = 1) {
  const int Ho = H / POOL, Wo = W / POOL;
  if (tabs) {
    const int g = blockIdx.z;
    mean = tabs[g], var = tabs[G + g], gamma = tabs[2 * G + g], beta = tabs[3 * G + g];
    u += (size_t)g * B * H * W * C4;
    dy += (size_t)g * B * Ho * Wo * C4;
    du += (size_t)g * B * H * W * C4;
    dbeta += (size_t)g * 4 * C4;
    dgamma += (size_t)g * 4 * C4;
  }
  const int er = blockIdx.x * 256 + threadIdx.x;
  if (er >= Wo * C4) return;
  const int xo = er >> lg, cg = er & (C4 - 1);
  const BnConst k = bn_const(mean, var, gamma, beta, eps, 4 * cg);
  const float lo = relu ? 0.f : -__builtin_inff();
  f32x4 db, dg;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    db[i] = dbeta[4 * cg + i] * inv_n;
    dg[i] = dgamma[4 * cg + i] * inv_n;
  }
  constexpr int RU = kRowsPerIter<POOL>;
  const int rows = B * Ho;
  for (int row0 = blockIdx.y * RU; row0 < rows; row0 += gridDim.y * RU) {
    f32x4 w[RU][POOL * POOL], dyv[RU];
#pragma unroll
    for (int r = 0; r < RU; ++r)
      if (row0 + r < rows) {
        const int b = (row0 + r) / Ho, yo = (row0 + r) - b * Ho;
        load_window<POOL, UB>(u, b, yo, xo, H, W, C4, cg, w[r]);
        dyv[r] = Q4<DB>::ld(dy, ((row0 + r) * Wo + xo) * C4 + cg);
      }
#pragma unroll
    for (int r = 0; r < RU; ++r)
      if (row0 + r < rows) {
        const int b = (row0 + r) / Ho, yo = (row0 + r) - b * Ho;
        f32x4 dv[POOL * POOL], xh[POOL * POOL];
        window_grad<POOL>(w[r], dyv[r], k, lo, relu, dv, xh);
#pragma unroll
        for (int q = 0; q < POOL * POOL; ++q) {
          const f32x4 rr = var ? k.g * (dv[q] - db - xh[q] * dg) : dv[q];
          Q4<UB>::st(du, ((b * H + yo * POOL + (q / POOL)) * W + xo * POOL + (q % POOL)) * C4 + cg, rr);
        }
      }
  }
}

// 1 << lg == C / 4 if the fast forms apply to this shape, else -1
inline int v4_log2(int C, size_t elems) {
  if (C % 4 || elems >= (1ull << 31)) return -1;
  const int C4 = C / 4;
  for (int lg = 0; lg <= 6; ++lg)
    if ((1 << lg) == C4) return lg;
  return -1;
}

}  // namespace train
}  // namespace ra

using namespace ra;

extern "C" int ra_adam_step_f32(float *params, const float *grads, float *m, float *v, const float *wd_coef,
                                size_t n, float lr_t, float beta1, float beta2, float eps, float clip,
                                float grad_scale, void *stream) {
  if (!params || !grads || !m || !v) return fail(RA_E_INVALID, "ra_adam_step_f32: null pointer");
  if (n == 0) return 0;
  size_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(train::adam_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), params, grads, m, v,
                     wd_coef, n, lr_t, beta1, beta2, eps, clip, grad_scale, (const int *)nullptr, 0, 0);
  return launch_status("ra_adam_step_f32");
}

extern "C" int ra_adam_step_guarded_f32(float *params, const float *grads, float *m, float *v, const float *wd_coef,
                                        size_t n, float lr_t, float beta1, float beta2, float eps, float clip,
                                        float grad_scale, const int *status, int n_solver, int n_other, void *stream) {
  if (!params || !grads || !m || !v) return fail(RA_E_INVALID, "ra_adam_step_guarded_f32: null pointer");
  if (n_solver < 0 || n_other < 0 || (n_solver + n_other > 0 && !status))
    return fail(RA_E_INVALID, "ra_adam_step_guarded_f32: status words");
  if (n == 0) return 0;
  size_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(train::adam_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), params, grads, m, v,
                     wd_coef, n, lr_t, beta1, beta2, eps, clip, grad_scale, status, n_solver, n_other);
  return launch_status("ra_adam_step_guarded_f32");
}

// =================================================================================================
// Train-mode layer pieces.  A layer of nnlib.cnn / nnlib.dcnn in training is
//   u = conv(x, w) + b                       ra_conv3x3_f32 (scale 1, shift b, no ReLU, no pool)
//   mean, var = moments(u over B,H,W)        ra_bn_moments_f32            nnlib.py:98 (biased variance)
//   y = pool(relu(gamma (u - mean) rsqrt(var + 1e-3) + beta))   ra_bn_act_pool_f32   nnlib.py:111-119,250-253
// and backward (batch statistics are part of the graph, nnlib.py:98-112):
//   dbeta, dgamma, du                        ra_bn_act_pool_bwd_f32
//   dx = conv(du, w^T flipped)               ra_conv3x3_f32 on ra_conv_pack_weights_dev(.., TRANSPOSED)
//   dw, db                                   ra_conv3x3_wgrad_f32
// All reductions are two-stage with a fixed summation order (no atomics): bit-reproducible.
namespace ra {
namespace train {

constexpr int kRedBlocks = 512;

// ---- per-channel moments: pass 1 sum, pass 2 sum of squared deviations (tf.nn.moments) ----
__global__ __launch_bounds__(256) void chan_sum_kernel(const float *u, size_t npix, int C, const float *mean,
                                                       float *part) {
  // thread = (pixel lane, channel): channel = tid % C when C divides 256; generic otherwise
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int lanes = 256 / C;  // pixel lanes per block (C <= 256, power-of-two-friendly but generic)
  const int c = tid % C, pl = tid / C;
  float s = 0.f;
  if (pl < lanes) {
    const float mu = mean ? mean[c] : 0.f;
    for (size_t p = (size_t)blockIdx.x * lanes + pl; p < npix; p += (size_t)gridDim.x * lanes) {
      const float v = u[p * C + c] - mu;
      s += mean ? v * v : v;
    }
  }
  red[tid] = pl < lanes ? s : 0.f;
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int k = 0; k < lanes; ++k) t += red[k * C + tid];
    part[(size_t)blockIdx.x * C + tid] = t;
  }
}
// One workgroup per channel: 256 threads stride over the partial blocks, then a fixed-shape tree
// (deterministic); a single thread per channel walking 512 strided partials took ~60 us.
__device__ inline float block_sum256(float v, float *red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void chan_final_kernel(const float *part, int nblocks, int C, float inv_n, float *out) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float t = 0.f;
  for (int k = threadIdx.x; k < nblocks; k += 256) t += part[(size_t)k * C + c];
  t = block_sum256(t, red);
  if (threadIdx.x == 0) out[c] = t * inv_n;
}

// tf.nn.moments from the records the conv epilogue left (ra_conv3x3_moments_f32: {n, S1, S2, pivot} per channel and
// record, sums of (u - pivot) and (u - pivot)^2): one workgroup per channel and ONE pass over the records.  Every record is
// re-based in float64 onto a common reference P0 (the first record's pivot — an actual value of the channel):
//   sum (u - P0) = S1 + n d,   sum (u - P0)^2 = S2 + d (2 S1 + n d),   d = pivot - P0
// and mean = P0 + A / N, var = Q / N - (A / N)^2.  The subtraction cancels only (mean - P0)^2 against the spread — a few
// sigma^2 at most, 53 bits under it — not the E[x^2] - E[x]^2 of raw float32 sums.
template <int NT>  // threads per channel: 64 (one wave, no barrier) up to 512 records, else 256
__global__ __launch_bounds__(NT) void moments_from_partials_kernel(const float *part, int nparts, int C, int CP, float *mean,
                                                                   float *var) {
  __shared__ double red[3][NT / 64];
  const int c = blockIdx.x, tid = threadIdx.x;
  const f32x4 r0 = *reinterpret_cast<const f32x4 *>(part + (size_t)c * 4);
  const double P0 = r0[0] > 0.f ? (double)r0[3] : 0.0;
  double n = 0.0, sa = 0.0, sq = 0.0;
  for (int k = tid; k < nparts; k += NT) {
    const f32x4 r = *reinterpret_cast<const f32x4 *>(part + ((size_t)k * CP + c) * 4);
    if (r[0] > 0.f) {
      const double d = (double)r[3] - P0, nd = (double)r[0] * d;
      n += (double)r[0];
      sa += (double)r[1] + nd;
      sq += (double)r[2] + d * (2.0 * (double)r[1] + nd);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    sa += __shfl_xor(sa, o, 64);
    sq += __shfl_xor(sq, o, 64);
  }
  if constexpr (NT > 64) {
    if ((tid & 63) == 0) red[0][tid >> 6] = n, red[1][tid >> 6] = sa, red[2][tid >> 6] = sq;
    __syncthreads();
    n = sa = sq = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) n += red[0][w], sa += red[1][w], sq += red[2][w];
  }
  if (tid == 0) {
    const double N = n > 0.0 ? n : 1.0, a = sa / N, v = sq / N - a * a;
    mean[c] = (float)(P0 + a);
    var[c] = (float)(v > 0.0 ? v : 0.0);
  }
}

// ---- y = pool(relu(gamma * (u - mean) * rstd + beta)) ----
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const float *u, const float *mean, const float *var,
                                                          const float *gamma, const float *beta, float eps, int relu,
                                                          int pool, int B, int H, int W, int C, float *y) {
  const int Ho = H / pool, Wo = W / pool;
  const size_t total = (size_t)B * Ho * Wo * C;
  const float lo = relu ? 0.f : -__builtin_inff();
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t r = e / C;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    // (u - mean) first: the folded form u * g + (beta - mean * g) cancels badly in channels whose
    // mean is large against their spread, and rstd amplifies that error layer after layer
    const float g = (gamma ? gamma[c] : 1.f) * (var ? rsqrtf(var[c] + eps) : 1.f);
    const float mu = mean ? mean[c] : 0.f, be = beta ? beta[c] : 0.f;
    float best = -__builtin_inff();
    for (int dy = 0; dy < pool; ++dy)
      for (int dx = 0; dx < pool; ++dx) {
        const float v = (u[(((size_t)b * H + yo * pool + dy) * W + xo * pool + dx) * C + c] - mu) * g + be;
        best = fmaxf(best, fmaxf(v, lo));
      }
    y[e] = best;
  }
}

// ---- backward, stage 1: per-channel sums of dv and dv * xhat (dv = dy routed through pool + ReLU) ----
__device__ inline void bwd_point(const float *u, const float *dy, float g, float be, float mu, float rstd, float lo,
                                 int relu, int pool, int b, int yy, int xx, int H, int W, int C, int c, float &dv,
                                 float &xhat) {
  // gradient reaching pre-activation v at conv pixel (yy, xx): the pooled window's FIRST maximum gets dy
  const float uv = u[(((size_t)b * H + yy) * W + xx) * C + c];
  xhat = (uv - mu) * rstd;
  const float v = (uv - mu) * g + be;
  if (pool == 1) {
    dv = (relu && v <= 0.f) ? 0.f : dy[(((size_t)b * H + yy) * W + xx) * C + c];
    return;
  }
  const int yo = yy >> 1, xo = xx >> 1, Ho = H >> 1, Wo = W >> 1;
  float best = -__builtin_inff();
  int arg = 0;
  for (int k = 0; k < 4; ++k) {
    const float w = (u[(((size_t)b * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1)) * C + c] - mu) * g + be;
    const float a = fmaxf(w, lo);
    if (a > best) {
      best = a;
      arg = k;
    }
  }
  const bool mine = arg == (((yy & 1) << 1) | (xx & 1));
  dv = (mine && !(relu && v <= 0.f)) ? dy[(((size_t)b * Ho + yo) * Wo + xo) * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *u, const float *dy, const float *mean,
                                                            const float *var, const float *gamma, const float *beta,
                                                            float eps, int relu, int pool, int B, int H, int W, int C,
                                                            float *part) {
  __shared__ float r0[256], r1[256];
  const int tid = threadIdx.x, lanes = 256 / C, c = tid % C, pl = tid / C;
  float s0 = 0.f, s1 = 0.f;
  if (pl < lanes) {
    const float rstd = var ? rsqrtf(var[c] + eps) : 1.f, mu = mean ? mean[c] : 0.f;
    const float g = (gamma ? gamma[c] : 1.f) * rstd, sh = beta ? beta[c] : 0.f;
    const float lo = relu ? 0.f : -__builtin_inff();
    const size_t npix = (size_t)B * H * W;
    for (size_t p = (size_t)blockIdx.x * lanes + pl; p < npix; p += (size_t)gridDim.x * lanes) {
      const int xx = (int)(p % W);
      const size_t r = p / W;
      const int yy = (int)(r % H), b = (int)(r / H);
      float dv, xhat;
      bwd_point(u, dy, g, sh, mu, rstd, lo, relu, pool, b, yy, xx, H, W, C, c, dv, xhat);
      s0 += dv;
      s1 += dv * xhat;
    }
  }
  r0[tid] = pl < lanes ? s0 : 0.f;
  r1[tid] = pl < lanes ? s1 : 0.f;
  __syncthreads();
  if (tid < C) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < lanes; ++k) {
      t0 += r0[k * C + tid];
      t1 += r1[k * C + tid];
    }
    part[((size_t)blockIdx.x * 2) * C + tid] = t0;
    part[((size_t)blockIdx.x * 2 + 1) * C + tid] = t1;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float *part, int nblocks, int C, float *dbeta, float *dgamma,
                                                           float *acc_beta = nullptr, float *acc_gamma = nullptr,
                                                           float *const *tabs = nullptr, int G = 1) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  if (tabs) {
    const int g = blockIdx.y;
    part += (size_t)g * nblocks * 2 * C;
    dbeta += (size_t)g * C;
    dgamma += (size_t)g * C;
    acc_gamma = tabs[4 * G + g], acc_beta = tabs[5 * G + g];
  }
  float t0 = 0.f, t1 = 0.f;
  for (int k = threadIdx.x; k < nblocks; k += 256) {
    t0 += part[((size_t)k * 2) * C + c];
    t1 += part[((size_t)k * 2 + 1) * C + c];
  }
  t0 = block_sum256(t0, red);
  t1 = block_sum256(t1, red);
  if (threadIdx.x == 0) {
    dbeta[c] = t0;
    dgamma[c] = t1;
    if (acc_beta) acc_beta[c] += t0;    // straight into the gradient bucket (one writer per element)
    if (acc_gamma) acc_gamma[c] += t1;
  }
}
// ---- stage 2: du = gamma * rstd * (dv - dbeta / n - xhat * dgamma / n)   (batch-norm: var given)
//               du = dv                                                    (no BN)
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float *u, const float *dy, const float *mean,
                                                        const float *var, const float *gamma, const float *beta,
                                                        const float *dbeta, const float *dgamma, float eps, int relu,
                                                        int pool, int B, int H, int W, int C, float *du, float inv_n) {
  const size_t total = (size_t)B * H * W * C;
  const float lo = relu ? 0.f : -__builtin_inff();
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t r = e / C;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H), b = (int)(r / H);
    const float rstd = var ? rsqrtf(var[c] + eps) : 1.f, mu = mean ? mean[c] : 0.f;
    const float g = (gamma ? gamma[c] : 1.f) * rstd, sh = beta ? beta[c] : 0.f;
    float dv, xhat;
    bwd_point(u, dy, g, sh, mu, rstd, lo, relu, pool, b, yy, xx, H, W, C, c, dv, xhat);
    du[e] = var ? g * (dv - dbeta[c] * inv_n - xhat * dgamma[c] * inv_n) : dv;
  }
}

// ---- small tensors (the one-channel output layer of the deconvolution net: B x 48 x 48 values per timestep): the
// whole BatchNorm backward of a call — both sums, dbeta / dgamma and du — in ONE workgroup per call; G calls (a layer's
// timesteps) = G workgroups of one launch.  1024 % C == 0 keeps a thread on one channel; fixed-shape LDS tree.
constexpr int kSmallThreads = 1024;
constexpr size_t kSmallElems = 65536;  // per group
inline bool small_ok(int C, size_t elems, int which = 1) {
  static int on = -1;  // RA_BN_SMALL=0: the multi-launch forms; 2: only the moments, 3: only the backward (debugging aids)
  if (on < 0) {
    const char *e = getenv("RA_BN_SMALL");
    on = e ? atoi(e) : 1;
  }
  return (on == 1 || on == which) && C >= 1 && C <= 64 && (C & (C - 1)) == 0 && elems <= kSmallElems;
}
__global__ __launch_bounds__(kSmallThreads) void bn_bwd_small_kernel(const float *u, const float *dy, const float *mean, const float *var,
                                                                    const float *gamma, const float *beta, float eps, int relu, int pool,
                                                                    int B, int H, int W, int C, float *dbeta, float *dgamma,
                                                                    float *acc_beta, float *acc_gamma, float *du, float inv_n,
                                                                    const float *const *tabs, int G) {
  __shared__ float r0[kSmallThreads], r1[kSmallThreads];
  const int g_ = blockIdx.x, tid = threadIdx.x, c = tid % C;
  const size_t total = (size_t)B * H * W * C;
  if (tabs) {
    mean = tabs[g_], var = tabs[G + g_], gamma = tabs[2 * G + g_], beta = tabs[3 * G + g_];
    acc_gamma = const_cast<float *>(tabs[4 * G + g_]), acc_beta = const_cast<float *>(tabs[5 * G + g_]);
    u += (size_t)g_ * total;
    dy += (size_t)g_ * (total / (pool * pool));
    du += (size_t)g_ * total;
    dbeta += (size_t)g_ * C, dgamma += (size_t)g_ * C;
  }
  const float rstd = var ? rsqrtf(var[c] + eps) : 1.f, mu = mean ? mean[c] : 0.f;
  const float g = (gamma ? gamma[c] : 1.f) * rstd, sh = beta ? beta[c] : 0.f;
  const float lo = relu ? 0.f : -__builtin_inff();
  float s0 = 0.f, s1 = 0.f;
  for (size_t e = tid; e < total; e += kSmallThreads) {
    size_t r = e / C;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H), b = (int)(r / H);
    float dv, xhat;
    bwd_point(u, dy, g, sh, mu, rstd, lo, relu, pool, b, yy, xx, H, W, C, c, dv, xhat);
    s0 += dv;
    s1 += dv * xhat;
  }
  r0[tid] = s0, r1[tid] = s1;
  __syncthreads();
  for (int o = kSmallThreads / 2; o >= C; o >>= 1) {
    if (tid < o) r0[tid] += r0[tid + o], r1[tid] += r1[tid + o];
    __syncthreads();
  }
  const float db = r0[c], dg = r1[c];
  if (tid < C) {
    dbeta[c] = db, dgamma[c] = dg;
    if (acc_beta) acc_beta[c] += db;
    if (acc_gamma) acc_gamma[c] += dg;
  }
  for (size_t e = tid; e < total; e += kSmallThreads) {
    size_t r = e / C;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H), b = (int)(r / H);
    float dv, xhat;
    bwd_point(u, dy, g, sh, mu, rstd, lo, relu, pool, b, yy, xx, H, W, C, c, dv, xhat);
    du[e] = var ? g * (dv - db * inv_n - xhat * dg * inv_n) : dv;
  }
}
// tf.nn.moments of a small tensor in one launch: mean, then the mean of the squared deviations about it
__global__ __launch_bounds__(kSmallThreads) void moments_small_kernel(const float *u, size_t total, int C, float inv_n, float *mean,
                                                                     float *var) {
  __shared__ float red[kSmallThreads];
  const int tid = threadIdx.x, c = tid % C;
  auto chan_sum = [&](float v) {
    red[tid] = v;
    __syncthreads();
    for (int o = kSmallThreads / 2; o >= C; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const float r = red[c];
    __syncthreads();
    return r;
  };
  float s = 0.f;
  for (size_t e = tid; e < total; e += kSmallThreads) s += u[e];
  const float mu = chan_sum(s) * inv_n;
  s = 0.f;
  for (size_t e = tid; e < total; e += kSmallThreads) {
    const float d = u[e] - mu;
    s += d * d;
  }
  const float v = chan_sum(s) * inv_n;
  if (tid < C) mean[c] = mu, var[c] = v;
}

// ---- device-side weight repack (the host form is ra_conv_pack_weights) ----
__global__ void pack_weights_kernel(const float *w, int Cin_w, int Cout, int Cin, const int *chan_map, int tr, int CK,
                                    int cp, float *out) {
  const int total = 9 * Cin * cp;
  const int NCG = CK / 4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int co = e % cp;
    int r = e / cp;
    const int ksub = r % 4;
    r /= 4;
    const int cg = r % NCG;
    r /= NCG;
    const int tap = r % 9, chunk = r / 9;
    const int c = chunk * CK + cg * 4 + ksub;
    const int src_c = chan_map ? chan_map[c] : c;
    const int ky = tap / 3, kx = tap % 3;
    float v = 0.f;
    if (co < Cout && src_c >= 0) {
      if (!tr)
        v = w[(((size_t)ky * 3 + kx) * Cin_w + src_c) * Cout + co];
      else
        v = w[(((size_t)(2 - ky) * 3 + (2 - kx)) * Cout + co) * Cin_w + src_c];
    }
    out[e] = v;
  }
}

// ---- y[b,i,j,:] = x[b,2i+1,2j+1,:]: the adjoint of the zero-stuffing of a stride-2 transposed conv ----
__global__ __launch_bounds__(256) void subsample_odd_kernel(const float *x, int B, int H, int W, int C, float *y) {
  const size_t total = (size_t)B * H * W * C;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t r = e / C;
    const int j = (int)(r % W);
    r /= W;
    const int i = (int)(r % H), b = (int)(r / H);
    y[e] = x[(((size_t)b * 2 * H + 2 * i + 1) * 2 * W + 2 * j + 1) * C + c];
  }
}

// ---- out[b,n,p] = sum_t w[b,n,t] * y[b,t,p] + bias[b,n]: the adjoint of the pairwise soft IoU ----
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void weighted_sum_multi_kernel(const float *w, const float *bias, const float *y, int N,
                                                                 int T, int HW, float *out, size_t o_img, size_t o_row) {
  const int b = blockIdx.z, n = blockIdx.y;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= HW) return;
  const float *yb = y + (size_t)b * T * HW + e;
  const float b0 = bias ? bias[(size_t)b * N + n] : 0.f;
  f32x4 acc = f32x4{b0, b0, b0, b0};
  for (int t = 0; t < T; ++t) {
    const float wt = w[((size_t)b * N + n) * T + t];
    if (wt != 0.f) acc += wt * *reinterpret_cast<const f32x4 *>(yb + (size_t)t * HW);  // uniform per (b, n)
  }
  *reinterpret_cast<f32x4 *>(out + (size_t)b * o_img + (size_t)n * o_row + e) = acc;
}

// ---- the canvas of the next timestep (full_model.py:826-848), written straight into the next packed controller-CNN
// input: per pixel   g = sum_t match[b,t] y_gt[b,t,p];  g -= g * noise[b,p];  y_c = knob[b] g + (1 - knob[b]) y[b,p];
// canvas' = max(y_c, canvas)   (no knob: y_c = y), every product and sum rounded on its own like the element-wise
// chain it replaces.  nxt[b,p,:] = prev[b,p,:] with channel `cc` = canvas'.  C in {4, 8, ...}: one float4 group of the
// pixel holds the canvas; the other groups are copied. ----
__global__ __launch_bounds__(256) void canvas_step_kernel(const f32x4 *prev, int C4, int cc, int HW, const float *y,
                                                          const float *match, const float *y_gt, int T, const float *noise,
                                                          const float *knob, int knob_stride, f32x4 *nxt) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const size_t px = (size_t)b * HW + p;
  float yc = y[px];
  if (match) {
    float g = 0.f;
    const float *yb = y_gt + (size_t)b * T * HW + p;
    for (int t = 0; t < T; ++t) {
      const float wt = match[(size_t)b * T + t];
      if (wt != 0.f) g = __fadd_rn(g, __fmul_rn(wt, yb[(size_t)t * HW]));  // uniform per image
    }
    if (noise) g = __fsub_rn(g, __fmul_rn(g, noise[px]));
    const float k = knob[(size_t)b * knob_stride];
    yc = __fadd_rn(__fmul_rn(k, g), __fmul_rn(__fsub_rn(1.0f, k), yc));
  }
  const int cg = cc >> 2, cl = cc & 3;
  for (int q = 0; q < C4; ++q) {
    f32x4 v = prev[px * C4 + q];
    if (q == cg) v[cl] = fmaxf(yc, v[cl]);
    nxt[px * C4 + q] = v;
  }
}

}  // namespace train
}  // namespace ra

extern "C" int ra_canvas_step_f32(const float *inp_prev, int C, int canvas_chan, int B, int HW, const float *y, const float *match,
                                  const float *y_gt, int T, const float *noise, const float *knob, int knob_stride,
                                  float *inp_next, void *stream) {
  if (!inp_prev || !y || !inp_next || B <= 0 || HW <= 0 || C <= 0 || C % 4 || canvas_chan < 0 || canvas_chan >= C ||
      (match && (!y_gt || !knob || T <= 0 || knob_stride < 1)))
    return fail(RA_E_INVALID, "ra_canvas_step_f32: bad argument");
  hipLaunchKernelGGL(train::canvas_step_kernel, dim3(ceil_div(HW, 256), B), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const train::f32x4 *>(inp_prev), C / 4, canvas_chan, HW, y, match, y_gt, T, noise, knob,
                     knob_stride, reinterpret_cast<train::f32x4 *>(inp_next));
  return launch_status("ra_canvas_step_f32");
}

extern "C" size_t ra_bn_workspace_floats(int C) { return (size_t)ra::train::kRedBlocks * 2 * (C > 0 ? C : 1); }

extern "C" int ra_bn_moments_f32(const float *u, size_t npix, int C, float *ws, size_t ws_floats, float *mean,
                                 float *var, void *stream) {
  if (!u || !ws || !mean || !var || npix == 0 || C <= 0) return fail(RA_E_INVALID, "ra_bn_moments_f32: bad argument");
  if (C > 256) return fail(RA_E_SHAPE, "ra_bn_moments_f32: C %d > 256", C);
  if (ws_floats < ra_bn_workspace_floats(C)) return fail(RA_E_WORKSPACE, "ra_bn_moments_f32: workspace too small");
  hipStream_t st = as_stream(stream);
  const float inv_n = 1.f / (float)npix;
  if (train::small_ok(C, npix * C, 2)) {
    hipLaunchKernelGGL(train::moments_small_kernel, dim3(1), dim3(train::kSmallThreads), 0, st, u, npix * C, C, inv_n, mean, var);
    return launch_status("ra_bn_moments_f32");
  }
  if (const int lg = train::v4_log2(C, npix * C); lg >= 0) {
    const int C4 = C / 4, n4 = (int)(npix * C4);
    int nb4 = ceil_div(n4, 256);
    if (nb4 > train::kRedBlocks) nb4 = train::kRedBlocks;
    const train::f32x4 *u4 = reinterpret_cast<const train::f32x4 *>(u);
    hipLaunchKernelGGL(train::chan_sum_v4_kernel, dim3(nb4), dim3(256), 0, st, u4, n4, C4, static_cast<const float *>(nullptr), ws);
    hipLaunchKernelGGL(train::chan_final_kernel, dim3(C), dim3(256), 0, st, ws, nb4, C, inv_n, mean);
    hipLaunchKernelGGL(train::chan_sum_v4_kernel, dim3(nb4), dim3(256), 0, st, u4, n4, C4, mean, ws);
    hipLaunchKernelGGL(train::chan_final_kernel, dim3(C), dim3(256), 0, st, ws, nb4, C, inv_n, var);
    return launch_status("ra_bn_moments_f32");
  }
  const int lanes = 256 / C;
  int nb = (int)((npix + lanes - 1) / lanes);
  if (nb > train::kRedBlocks) nb = train::kRedBlocks;
  hipLaunchKernelGGL(train::chan_sum_kernel, dim3(nb), dim3(256), 0, st, u, npix, C, static_cast<const float *>(nullptr), ws);
  hipLaunchKernelGGL(train::chan_final_kernel, dim3(C), dim3(256), 0, st, ws, nb, C, inv_n, mean);
  hipLaunchKernelGGL(train::chan_sum_kernel, dim3(nb), dim3(256), 0, st, u, npix, C, mean, ws);
  hipLaunchKernelGGL(train::chan_final_kernel, dim3(C), dim3(256), 0, st, ws, nb, C, inv_n, var);
  return launch_status("ra_bn_moments_f32");
}

extern "C" int ra_bn_moments_from_partials_f32(const float *part, int nparts, int C, float *mean, float *var, void *stream) {
  const int cp = ra_conv_cout_padded(C);
  if (!part || !mean || !var || nparts <= 0 || C <= 0 || !cp) return fail(RA_E_INVALID, "ra_bn_moments_from_partials_f32: bad argument");
  if (nparts <= 512)
    hipLaunchKernelGGL(train::moments_from_partials_kernel<64>, dim3(C), dim3(64), 0, as_stream(stream), part, nparts, C, cp, mean, var);
  else  // 1024 threads measured slower than 256 at 4096 records (6.4 against ~5.1 us): the launch, not the loop
    hipLaunchKernelGGL(train::moments_from_partials_kernel<256>, dim3(C), dim3(256), 0, as_stream(stream), part, nparts, C, cp, mean, var);
  return launch_status("ra_bn_moments_from_partials_f32");
}

namespace {
template <int POOL, bool UB, bool YB>
void launch_bn_act_pool_v4(dim3 g4, hipStream_t st, const void *u, const float *mean, const float *var, const float *gamma,
                           const float *beta, float eps, int relu, int B, int H, int W, int C4, int lg, void *y) {
  hipLaunchKernelGGL((train::bn_act_pool_v4_kernel<POOL, UB, YB>), g4, dim3(256), 0, st,
                     reinterpret_cast<const typename train::Q4<UB>::T *>(u), mean, var, gamma, beta, eps, relu, B, H, W, C4, lg,
                     reinterpret_cast<typename train::Q4<YB>::T *>(y));
}
// flags: bit 0 = u is stored as bf16, bit 1 = y is written as bf16 (only combinations the bf16 mode produces: 0, 1, 3)
int bn_act_pool_impl(const void *u, const float *mean, const float *var, const float *gamma, const float *beta, float eps, int relu,
                     int pool, int B, int H, int W, int C, void *y, int flags, void *stream) {
  if (!u || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(RA_E_INVALID, "ra_bn_act_pool_f32: bad argument");
  if ((pool != 1 && pool != 2) || (pool == 2 && ((H | W) & 1))) return fail(RA_E_SHAPE, "ra_bn_act_pool_f32: pool");
  if (const int lg = train::v4_log2(C, (size_t)B * H * W * C); lg >= 0) {
    const int C4 = C / 4, rows = B * (H / pool);
    const int ry = ceil_div(rows, pool == 1 ? train::kRowsPerIter<1> : train::kRowsPerIter<2>);
    const dim3 g4(ceil_div((W / pool) * C4, 256), ry < 16384 ? ry : 16384);
    hipStream_t st = as_stream(stream);
#define RA_BNF(P, UB, YB) launch_bn_act_pool_v4<P, UB, YB>(g4, st, u, mean, var, gamma, beta, eps, relu, B, H, W, C4, lg, y)
    if (flags == 0) { if (pool == 2) RA_BNF(2, false, false); else RA_BNF(1, false, false); }
    else if (flags == 1) { if (pool == 2) RA_BNF(2, true, false); else RA_BNF(1, true, false); }
    else if (flags == 3) { if (pool == 2) RA_BNF(2, true, true); else RA_BNF(1, true, true); }
    else return fail(RA_E_INVALID, "ra_bn_act_pool_bf16_f32: flags %d", flags);
#undef RA_BNF
    return launch_status("ra_bn_act_pool_f32");
  }
  if (flags) return fail(RA_E_SHAPE, "ra_bn_act_pool_bf16_f32: bf16 storage needs C %% 4 == 0 with C / 4 a power of two (C %d)", C);
  return -1000;  // the generic float32 kernel (the caller below)
}
}  // namespace

extern "C" int ra_bn_act_pool_bf16_f32(const void *u, const float *mean, const float *var, const float *gamma, const float *beta,
                                       float eps, int relu, int pool, int B, int H, int W, int C, void *y, int flags, void *stream) {
  const int rc = bn_act_pool_impl(u, mean, var, gamma, beta, eps, relu, pool, B, H, W, C, y, flags, stream);
  if (rc != -1000) return rc;
  return ra_bn_act_pool_f32(static_cast<const float *>(u), mean, var, gamma, beta, eps, relu, pool, B, H, W, C, static_cast<float *>(y), stream);
}

extern "C" int ra_bn_act_pool_f32(const float *u, const float *mean, const float *var, const float *gamma,
                                  const float *beta, float eps, int relu, int pool, int B, int H, int W, int C, float *y,
                                  void *stream) {
  {
    const int rc = bn_act_pool_impl(u, mean, var, gamma, beta, eps, relu, pool, B, H, W, C, y, 0, stream);
    if (rc != -1000) return rc;
  }
  const size_t total = (size_t)B * (H / pool) * (W / pool) * C;
  size_t grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(train::bn_act_pool_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), u, mean, var, gamma,
                     beta, eps, relu, pool, B, H, W, C, y);
  return launch_status("ra_bn_act_pool_f32");
}

namespace {
// stages: 1 = the two reductions (dbeta, dgamma over THIS call's pixels), 2 = du from dbeta / dgamma and the count
// they were summed over (n_total; 0 = this call's B*H*W).  Data-parallel training with whole-batch statistics
// runs stage 1, all-reduces the 2C sums, then stage 2 with the global count (ra_train.ConvBNActPool).
// the v4 kernels of one (or G stacked) BatchNorm backward call(s), for one storage format (UB: u and du bf16, DB: dy bf16)
template <int POOL, bool UB, bool DB>
void launch_bn_bwd_v4(int stages, dim3 gr, dim3 gd, hipStream_t st, const void *u, const void *dy, const float *mean, const float *var,
                      const float *gamma, const float *beta, float eps, int relu, int B, int H, int W, int C, int C4, int lg, float *ws,
                      int nblk, float *dbeta, float *dgamma, float *acc_beta, float *acc_gamma, void *du, float inv_n,
                      const float *const *ct = nullptr, float *const *mt = nullptr, int G = 1) {
  typedef typename train::Q4<UB>::T TU;
  typedef typename train::Q4<DB>::T TD;
  const TU *u4 = reinterpret_cast<const TU *>(u);
  const TD *dy4 = reinterpret_cast<const TD *>(dy);
  if (stages & 1) {
    hipLaunchKernelGGL((train::bn_bwd_reduce_v4_kernel<POOL, UB, DB>), gr, dim3(256), 0, st, u4, dy4, mean, var, gamma, beta, eps, relu, B,
                       H, W, C4, lg, ws, ct, G);
    if (ct)
      hipLaunchKernelGGL(train::bn_bwd_final_kernel, dim3(C, G), dim3(256), 0, st, ws, nblk, C, dbeta, dgamma, (float *)nullptr,
                         (float *)nullptr, mt, G);
    else
      hipLaunchKernelGGL(train::bn_bwd_final_kernel, dim3(C), dim3(256), 0, st, ws, nblk, C, dbeta, dgamma, acc_beta, acc_gamma);
  }
  if (stages & 2)
    hipLaunchKernelGGL((train::bn_bwd_dx_v4_kernel<POOL, UB, DB>), gd, dim3(256), 0, st, u4, dy4, mean, var, gamma, beta, dbeta, dgamma,
                       eps, relu, B, H, W, C4, lg, reinterpret_cast<TU *>(du), inv_n, ct, G);
}
template <typename... A>
int dispatch_bn_bwd_v4(int pool, int flags, A... a) {
  if (flags == 0) { if (pool == 2) launch_bn_bwd_v4<2, false, false>(a...); else launch_bn_bwd_v4<1, false, false>(a...); }
  else if (flags == 1) { if (pool == 2) launch_bn_bwd_v4<2, true, false>(a...); else launch_bn_bwd_v4<1, true, false>(a...); }
  else if (flags == 3) { if (pool == 2) launch_bn_bwd_v4<2, true, true>(a...); else launch_bn_bwd_v4<1, true, true>(a...); }
  else return fail(RA_E_INVALID, "ra_bn_act_pool_bwd: storage flags %d", flags);
  return 0;
}

int bn_bwd_impl(const float *u, const float *dy, const float *mean, const float *var, const float *gamma,
                const float *beta, float eps, int relu, int pool, int B, int H, int W, int C, float *ws,
                size_t ws_floats, float *dgamma, float *dbeta, float *du, float *acc_gamma, float *acc_beta,
                void *stream, int stages = 3, double n_total = 0.0, int flags = 0) {
  if (!u || !dy || !dgamma || !dbeta || ((stages & 1) && !ws) || ((stages & 2) && !du) || B <= 0 || H <= 0 || W <= 0 || C <= 0)
    return fail(RA_E_INVALID, "ra_bn_act_pool_bwd_f32: bad argument");
  if (C > 256) return fail(RA_E_SHAPE, "ra_bn_act_pool_bwd_f32: C %d > 256", C);
  if ((pool != 1 && pool != 2) || (pool == 2 && ((H | W) & 1))) return fail(RA_E_SHAPE, "ra_bn_act_pool_bwd_f32: pool");
  if ((stages & 1) && ws_floats < ra_bn_workspace_floats(C)) return fail(RA_E_WORKSPACE, "ra_bn_act_pool_bwd_f32: workspace too small");
  hipStream_t st = as_stream(stream);
  const size_t npix = (size_t)B * H * W;
  const float inv_n = (float)(1.0 / (n_total > 0.0 ? n_total : (double)npix));
  if (const int lg = train::v4_log2(C, npix * C); lg >= 0 && ceil_div((W / pool) * (C / 4), 256) <= train::kRedBlocks) {
    const int C4 = C / 4, rows = B * (H / pool), gx = ceil_div((W / pool) * C4, 256);
    int gy = train::kRedBlocks / gx;
    if (gy > rows) gy = rows;
    const int ry = ceil_div(rows, pool == 1 ? train::kRowsPerIter<1> : train::kRowsPerIter<2>);
    const dim3 gr(gx, gy), gd(gx, ry < 16384 ? ry : 16384);
    if (const int rc = dispatch_bn_bwd_v4(pool, flags, stages, gr, gd, st, (const void *)u, (const void *)dy, mean, var, gamma, beta, eps, relu,
                                          B, H, W, C, C4, lg, ws, gx * gy, dbeta, dgamma, acc_beta, acc_gamma, (void *)du, inv_n,
                                          (const float *const *)nullptr, (float *const *)nullptr, 1))
      return rc;
    return launch_status("ra_bn_act_pool_bwd_f32");
  }
  if (flags) return fail(RA_E_SHAPE, "ra_bn_act_pool_bwd: bf16 storage needs C %% 4 == 0 with C / 4 a power of two (C %d)", C);
  if (stages == 3 && train::small_ok(C, npix * C, 3)) {
    hipLaunchKernelGGL(train::bn_bwd_small_kernel, dim3(1), dim3(train::kSmallThreads), 0, st, u, dy, mean, var, gamma, beta, eps, relu,
                       pool, B, H, W, C, dbeta, dgamma, acc_beta, acc_gamma, du, inv_n, (const float *const *)nullptr, 1);
    return launch_status("ra_bn_act_pool_bwd_f32");
  }
  const int lanes = 256 / C;
  int nb = (int)((npix + lanes - 1) / lanes);
  if (nb > train::kRedBlocks) nb = train::kRedBlocks;
  if (stages & 1) {
    hipLaunchKernelGGL(train::bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, u, dy, mean, var, gamma, beta, eps, relu, pool,
                       B, H, W, C, ws);
    hipLaunchKernelGGL(train::bn_bwd_final_kernel, dim3(C), dim3(256), 0, st, ws, nb, C, dbeta, dgamma, acc_beta, acc_gamma);
  }
  if (stages & 2) {
    size_t grid = (npix * C + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(train::bn_bwd_dx_kernel, dim3((unsigned)grid), dim3(256), 0, st, u, dy, mean, var, gamma, beta, dbeta,
                       dgamma, eps, relu, pool, B, H, W, C, du, inv_n);
  }
  return launch_status("ra_bn_act_pool_bwd_f32");
}
}  // namespace

// G calls of one BatchNorm layer (its G timesteps) stacked along the batch — u [G*B,H,W,C], dy [G*B,H/pool,W/pool,C] — in
// one reduce / final / dx triple: every group has its own statistics and parameters, read through a device table of
// 6 G pointers {mean, var, gamma, beta, grad-bucket gamma, grad-bucket beta}[G]; dgamma / dbeta [G,C].
static int bn_bwd_grouped_impl(const void *u, const void *dy, const void *const *tabs, int G, float eps, int relu, int pool, int B,
                               int H, int W, int C, float *ws, size_t ws_floats, float *dgamma, float *dbeta, void *du, int flags,
                               void *stream) {
  if (!u || !dy || !tabs || !ws || !dgamma || !dbeta || !du || G <= 0 || B <= 0 || H <= 0 || W <= 0 || C <= 0)
    return fail(RA_E_INVALID, "ra_bn_act_pool_bwd_grouped_f32: bad argument");
  if ((pool != 1 && pool != 2) || (pool == 2 && ((H | W) & 1))) return fail(RA_E_SHAPE, "ra_bn_act_pool_bwd_grouped_f32: pool");
  if (ws_floats < (size_t)G * ra_bn_workspace_floats(C)) return fail(RA_E_WORKSPACE, "ra_bn_act_pool_bwd_grouped_f32: workspace too small");
  const size_t npix = (size_t)B * H * W;
  const int lg = train::v4_log2(C, npix * C);
  if (lg < 0 && flags == 0 && G <= 65535 && train::small_ok(C, npix * C, 3)) {  // one workgroup per group (the one-channel output layer)
    hipLaunchKernelGGL(train::bn_bwd_small_kernel, dim3(G), dim3(train::kSmallThreads), 0, as_stream(stream), static_cast<const float *>(u),
                       static_cast<const float *>(dy), (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                       (const float *)nullptr, eps, relu, pool, B, H, W, C, dbeta, dgamma, (float *)nullptr, (float *)nullptr,
                       static_cast<float *>(du), (float)(1.0 / (double)npix), reinterpret_cast<const float *const *>(tabs), G);
    return launch_status("ra_bn_act_pool_bwd_grouped_f32");
  }
  if (lg < 0 || ceil_div((W / pool) * (C / 4), 256) > train::kRedBlocks || G > 65535)
    return fail(RA_E_SHAPE, "ra_bn_act_pool_bwd_grouped_f32: C %d (needs C %% 4 == 0, C / 4 a power of two <= 64)", C);
  hipStream_t st = as_stream(stream);
  const float inv_n = (float)(1.0 / (double)npix);
  const int C4 = C / 4, rows = B * (H / pool), gx = ceil_div((W / pool) * C4, 256);
  int gy = train::kRedBlocks / gx;
  if (gy > rows) gy = rows;
  const float *const *ct = reinterpret_cast<const float *const *>(tabs);
  float *const *mt = reinterpret_cast<float *const *>(const_cast<void *const *>(tabs));
  const int ry = ceil_div(rows, pool == 1 ? train::kRowsPerIter<1> : train::kRowsPerIter<2>);
  const dim3 gr(gx, gy, G), gd(gx, ry < 16384 ? ry : 16384, G);
  const float *nul = nullptr;
  if (const int rc = dispatch_bn_bwd_v4(pool, flags, 3, gr, gd, st, u, dy, nul, nul, nul, nul, eps, relu, B, H, W, C, C4, lg, ws, gx * gy,
                                        dbeta, dgamma, (float *)nullptr, (float *)nullptr, du, inv_n, ct, mt, G))
    return rc;
  return launch_status("ra_bn_act_pool_bwd_grouped_f32");
}

extern "C" int ra_bn_act_pool_bwd_grouped_f32(const float *u, const float *dy, const void *const *tabs, int G, float eps, int relu,
                                              int pool, int B, int H, int W, int C, float *ws, size_t ws_floats, float *dgamma,
                                              float *dbeta, float *du, void *stream) {
  return bn_bwd_grouped_impl(u, dy, tabs, G, eps, relu, pool, B, H, W, C, ws, ws_floats, dgamma, dbeta, du, 0, stream);
}

// The bf16 mode's forms (model_opt['compute_dtype'] = 'bf16'): flags bit 0 = u is read and du written as bf16, bit 1 = dy is
// read as bf16 (0, 1 or 3); float32 statistics, sums and parameter gradients.
extern "C" int ra_bn_act_pool_bwd_grouped_bf16_f32(const void *u, const void *dy, const void *const *tabs, int G, float eps, int relu,
                                                   int pool, int B, int H, int W, int C, float *ws, size_t ws_floats, float *dgamma,
                                                   float *dbeta, void *du, int flags, void *stream) {
  return bn_bwd_grouped_impl(u, dy, tabs, G, eps, relu, pool, B, H, W, C, ws, ws_floats, dgamma, dbeta, du, flags, stream);
}

extern "C" int ra_bn_act_pool_bwd_acc_bf16_f32(const void *u, const void *dy, const float *mean, const float *var, const float *gamma,
                                               const float *beta, float eps, int relu, int pool, int B, int H, int W, int C, float *ws,
                                               size_t ws_floats, float *dgamma, float *dbeta, void *du, float *acc_gamma,
                                               float *acc_beta, int flags, void *stream) {
  return bn_bwd_impl(static_cast<const float *>(u), static_cast<const float *>(dy), mean, var, gamma, beta, eps, relu, pool, B, H, W, C, ws,
                     ws_floats, dgamma, dbeta, static_cast<float *>(du), acc_gamma, acc_beta, stream, 3, 0.0, flags);
}

extern "C" int ra_bn_act_pool_bwd_f32(const float *u, const float *dy, const float *mean, const float *var,
                                      const float *gamma, const float *beta, float eps, int relu, int pool, int B,
                                      int H, int W, int C, float *ws, size_t ws_floats, float *dgamma, float *dbeta,
                                      float *du, void *stream) {
  return bn_bwd_impl(u, dy, mean, var, gamma, beta, eps, relu, pool, B, H, W, C, ws, ws_floats, dgamma, dbeta, du, nullptr,
                     nullptr, stream);
}

extern "C" int ra_bn_act_pool_bwd_acc_f32(const float *u, const float *dy, const float *mean, const float *var,
                                          const float *gamma, const float *beta, float eps, int relu, int pool, int B,
                                          int H, int W, int C, float *ws, size_t ws_floats, float *dgamma, float *dbeta,
                                          float *du, float *acc_gamma, float *acc_beta, void *stream) {
  return bn_bwd_impl(u, dy, mean, var, gamma, beta, eps, relu, pool, B, H, W, C, ws, ws_floats, dgamma, dbeta, du, acc_gamma,
                     acc_beta, stream);
}

extern "C" int ra_bn_act_pool_bwd_reduce_f32(const float *u, const float *dy, const float *mean, const float *var,
                                             const float *gamma, const float *beta, float eps, int relu, int pool, int B,
                                             int H, int W, int C, float *ws, size_t ws_floats, float *dgamma, float *dbeta,
                                             float *acc_gamma, float *acc_beta, void *stream) {
  return bn_bwd_impl(u, dy, mean, var, gamma, beta, eps, relu, pool, B, H, W, C, ws, ws_floats, dgamma, dbeta, nullptr,
                     acc_gamma, acc_beta, stream, 1);
}

extern "C" int ra_bn_act_pool_bwd_dx_f32(const float *u, const float *dy, const float *mean, const float *var,
                                         const float *gamma, const float *beta, const float *dgamma_sum,
                                         const float *dbeta_sum, double n_total, float eps, int relu, int pool, int B, int H,
                                         int W, int C, float *du, void *stream) {
  return bn_bwd_impl(u, dy, mean, var, gamma, beta, eps, relu, pool, B, H, W, C, nullptr, 0, const_cast<float *>(dgamma_sum),
                     const_cast<float *>(dbeta_sum), du, nullptr, nullptr, stream, 2, n_total);
}

extern "C" int ra_conv_pack_weights_dev(const float *w, int Cin_w, int Cout, int Cin, const int *chan_map, int flags,
                                        float *out, void *stream) {
  const int cp = ra_conv_cout_padded(Cout);
  if (!w || !out || Cin_w <= 0 || Cin <= 0) return fail(RA_E_INVALID, "ra_conv_pack_weights_dev: bad argument");
  if (Cin % 4 || !cp) return fail(RA_E_SHAPE, "ra_conv_pack_weights_dev: Cin %d %% 4 or Cout %d", Cin, Cout);
  if (!chan_map && Cin_w != Cin) return fail(RA_E_SHAPE, "ra_conv_pack_weights_dev: Cin_w != Cin without map");
  const int CK = (Cin % 16 == 0) ? 16 : (Cin % 8 == 0) ? 8 : 4;
  const int total = 9 * Cin * cp;
  hipLaunchKernelGGL(train::pack_weights_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w, Cin_w, Cout,
                     Cin, chan_map, (flags & RA_CONV_TRANSPOSED) ? 1 : 0, CK, cp, out);
  return launch_status("ra_conv_pack_weights_dev");
}

extern "C" int ra_subsample_odd_f32(const float *x, int B, int H, int W, int C, float *y, void *stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(RA_E_INVALID, "ra_subsample_odd_f32: bad argument");
  size_t grid = ((size_t)B * H * W * C + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(train::subsample_odd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, B, H, W, C, y);
  return launch_status("ra_subsample_odd_f32");
}

extern "C" int ra_weighted_sum_multi_strided_f32(const float *w, const float *bias, const float *y, int B, int N, int T, int HW,
                                                 float *out, size_t o_img, size_t o_row, void *stream);
extern "C" int ra_weighted_sum_multi_f32(const float *w, const float *bias, const float *y, int B, int N, int T, int HW,
                                         float *out, void *stream) {
  return ra_weighted_sum_multi_strided_f32(w, bias, y, B, N, T, HW, out, (size_t)N * HW, (size_t)HW, stream);
}
// out[b][n] at b * o_img + n * o_row (floats): the gradient of timestep-major masks is written timestep-major
extern "C" int ra_weighted_sum_multi_strided_f32(const float *w, const float *bias, const float *y, int B, int N, int T, int HW,
                                                 float *out, size_t o_img, size_t o_row, void *stream) {
  if ((o_img | o_row) & 3) return fail(RA_E_SHAPE, "ra_weighted_sum_multi_strided_f32: strides must be multiples of 4 floats");
  if (!w || !y || !out || B <= 0 || N <= 0 || T <= 0 || HW <= 0)
    return fail(RA_E_INVALID, "ra_weighted_sum_multi_f32: bad argument");
  if (HW % 4 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15))
    return fail(RA_E_SHAPE, "ra_weighted_sum_multi_f32: H*W %% 4 and 16-byte aligned tensors required");
  hipLaunchKernelGGL(train::weighted_sum_multi_kernel, dim3(ceil_div(HW, 1024), N, B), dim3(256), 0, as_stream(stream), w,
                     bias, y, N, T, HW, out, o_img, o_row);
  return launch_status("ra_weighted_sum_multi_f32");
}

// =================================================================================================
// conv3x3 backward-weight on f32 MFMA.  For the SAME conv u = conv(X, Wf) the kernel returns
//   dWf[ky][kx][ci][co] = sum_{b,y,x} X[b, y+ky-1, x+kx-1, ci] * dU[b, y, x, co]      (X zero-padded;
//   with `upsample` X is the zero-stuffed image of a stride-2 transposed conv) and  db[co] = sum dU.
// GEMM view per tap: D[ci, co] = sum_pixels A[ci, pixel] * B[pixel, co]  — pixels are the K
// dimension of v_mfma_f32_16x16x4_f32 (A = 16 input channels x 4 pixels, B = 4 pixels x 16 couts).
// A workgroup owns a 16-channel slice of Cin (blockIdx.y) and walks 8 x 32 pixel tiles
// persistently; its 4 waves split the tile's rows (K split), every wave keeps all
// 10 (9 taps + bias) x CoutP/16 accumulator tiles for its rows in registers across tiles; at the
// end the waves are summed through LDS and the workgroup writes ONE partial; a second kernel adds
// the partials in a fixed order (deterministic, no atomics).
namespace ra {
namespace train {

constexpr int WTH = 8, WTW = 32, WLW = WTW + 2, WLH = WTH + 2;

// PACK = 0: the M rows of an MFMA are the 16 input channels of the slice, one accumulator tile per tap
// (9 + bias).  PACK = Cin (4 or 8): the M rows are (tap, channel) pairs, 9 * Cin of them in
// ceil(9 * Cin / 16) tiles — 3 instead of 9 k-step MFMAs per pixel quad for Cin = 4 (where 12 of the 16
// channel rows were zero), 5 for Cin = 8; a lane reads its row's pixel through a per-tile LDS offset.
// BF16 (compute_dtype = 'bf16'): the staged float32 pixels are rounded to bf16 as they leave LDS and four K steps
// (32 consecutive pixels of a row) go through ONE v_mfma_f32_16x16x32_bf16 (the gfx950 form); accumulation stays float32.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
__device__ inline bf16x4 pack_bf16(float v0, float v1, float v2, float v3) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v2, v3}, bf16x2));
  return __builtin_bit_cast(bf16x4, u32x2{lo, hi});
}

// PRE (Cout % 4 == 0, NT <= 2): the NEXT tile's global loads are issued into registers before the MFMA loop of the
// current one and written to LDS after it (one staging buffer, two barriers per tile as before): the loads' latency
// and the HBM stream hide behind the matrix work instead of in front of it.
// four consecutive elements at element offset `off` of a tensor stored as float32 or (bf: the bf16 mode's storage) bf16
__device__ inline f32x4 ld4_fmt(const float *base, size_t off, bool bf) {
  if (bf) {
    const u32x2q q = *reinterpret_cast<const u32x2q *>(reinterpret_cast<const char *>(base) + off * 2);
    return f32x4{__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xffff0000u),
                 __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xffff0000u)};
  }
  return *reinterpret_cast<const f32x4 *>(base + off);
}

__device__ inline float ld1_fmt(const float *base, size_t off, bool bf) {
  if (bf) return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(base)[off] << 16);
  return base[off];
}
template <int NT, int PACK = 0, bool BF16 = false, bool PRE = false>  // NT: output channels per workgroup / 16; blockIdx.z selects a 16*NT-wide slice of Cout
__global__ __launch_bounds__(256) void wgrad_kernel(const float *x, const float *du, int B, int Hs, int Ws, int Cin,
                                                    int ups, int H, int W, int Cout, int tiles_x, int tiles_y,
                                                    int ntiles, float *part, const float *const *xtab,
                                                    const float *const *dutab, int Bseg, int fmt) {
  // fmt (BF16 kernels only): bit 0 = x is stored as bf16, bit 1 = du is stored as bf16
  const bool xbf = BF16 && (fmt & 1), ubf = BF16 && (fmt & 2);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tx = lds;                      // [WLH][WLW][16]   input slice + halo, channel-contiguous
  float *tu = lds + WLH * WLW * 16;     // [WTH][WTW][16*NT] output gradient tile
  constexpr int CP = 16 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, ksub = lane >> 4;
  const int co0 = blockIdx.z * 16 * NT;       // first output channel of this workgroup's slice
  const int c0 = blockIdx.y * 16;             // first input channel of this workgroup's slice
  const int cn = Cin - c0 < 16 ? Cin - c0 : 16;  // real channels in the slice
  constexpr int MT = PACK ? (9 * PACK + 15) / 16 : 9;  // accumulator tiles of the filter taps; tile MT = the bias
  f32x4 acc[MT + 1][NT];
#pragma unroll
  for (int t = 0; t <= MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // PACK: LDS offset (floats, relative to the lane's pixel) of row m of tile t: its tap's pixel shift + its
  // channel; padding rows read channel 15 of the pixel, which is staged as zero (cn <= 8)
  int aoff[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if constexpr (PACK != 0) {
      const int R = 16 * t + m, tap = R / PACK, ci = R - tap * PACK;
      aoff[t] = tap < 9 ? ((tap / 3) * WLW + tap % 3) * 16 + ci : 15;
    } else {
      aoff[t] = ((t / 3) * WLW + t % 3) * 16 + m;
    }
  }
  // the channel groups beyond the slice's real channels are zero for the whole launch: written once, not per tile
  const int ng = (cn + 3) >> 2;
  if (ng < 4) {
    for (int e = tid; e < WLH * WLW * 4; e += 256) *reinterpret_cast<f32x4 *>(tx + e * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool vec_du = (Cout & 3) == 0;  // float4 loads of the output gradient
  const int per = tiles_x * tiles_y;
  constexpr int NRX = PRE ? (WLH * WLW * 4 + 255) / 256 : 1, NRU = PRE ? CP / 4 : 1;
  f32x4 rx[NRX], ru[NRU];
  auto prefetch = [&](int tile) {  // every load unconditional (clamped address, value selected afterwards)
    int b = tile / per;
    const int tr = tile - b * per;
    const float *xb = x, *ub = du;
    if (xtab) {
      const int seg = b / Bseg;
      xb = xtab[seg];
      ub = dutab[seg];
      b -= seg * Bseg;
    }
    const int ty0 = (tr / tiles_x) * WTH, tx0 = (tr % tiles_x) * WTW;
    auto load_x = [&](auto bf) {  // the storage format is uniform: ONE test around the whole unrolled loop
#pragma unroll
      for (int i = 0; i < NRX; ++i) {
        const int e = tid + 256 * i;
        const int pix = e / ng, c4 = e - pix * ng;
        const int r = pix / WLW, c = pix - r * WLW;
        const int Y = ty0 + r - 1, X = tx0 + c - 1;
        bool ok = (e < WLH * WLW * ng) & (Y >= 0) & (Y < H) & (X >= 0) & (X < W);
        int ys = Y, xs = X;
        if (ups) {
          ok = ok & (Y & 1) & (X & 1);
          ys = (Y - 1) >> 1;
          xs = (X - 1) >> 1;
        }
        const size_t off = ok ? (((size_t)b * Hs + ys) * Ws + xs) * Cin + c0 + 4 * c4 : 0;
        const f32x4 v = ld4_fmt(xb, off, decltype(bf)::value);
        rx[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto load_u = [&](auto bf) {
#pragma unroll
      for (int i = 0; i < NRU; ++i) {
        const int e = tid + 256 * i;
        const int c4 = e % (CP / 4), pix = e / (CP / 4);
        const int r = pix / WTW, c = pix - r * WTW;
        const int Y = ty0 + r, X = tx0 + c;
        const bool ok = (Y < H) & (X < W) & (co0 + 4 * c4 < Cout);
        const size_t off = ok ? (((size_t)b * H + Y) * W + X) * Cout + co0 + 4 * c4 : 0;
        const f32x4 v = ld4_fmt(ub, off, decltype(bf)::value);
        ru[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    if (xbf) load_x(std::true_type{}); else load_x(std::false_type{});
    if (ubf) load_u(std::true_type{}); else load_u(std::false_type{});
  };
  auto commit = [&]() {  // the prefetched tile -> LDS
#pragma unroll
    for (int i = 0; i < NRX; ++i) {
      const int e = tid + 256 * i;
      const int pix = e / ng, c4 = e - pix * ng;
      if (e < WLH * WLW * ng) *reinterpret_cast<f32x4 *>(tx + pix * 16 + 4 * c4) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < NRU; ++i) {
      const int e = tid + 256 * i;
      const int c4 = e % (CP / 4), pix = e / (CP / 4);
      *reinterpret_cast<f32x4 *>(tu + pix * CP + 4 * c4) = ru[i];
    }
  };
  if constexpr (PRE) {
    if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile / per;
    const int tr = tile - b * per;
    if (xtab) {  // the images of several calls of the layer (one per timestep): segment tables, Bseg images each
      const int seg = b / Bseg;
      x = xtab[seg];
      du = dutab[seg];
      b -= seg * Bseg;
    }
    const int ty0 = (tr / tiles_x) * WTH, tx0 = (tr % tiles_x) * WTW;
    __syncthreads();  // the previous tile's MFMA reads are complete
    if constexpr (PRE) {
      commit();
      __syncthreads();
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);  // in flight across the MFMA loop below
    } else {
    for (int e = tid; e < WLH * WLW * ng; e += 256) {  // input slice: one float4 (4 channels) per item
      const int pix = e / ng, c4 = e - pix * ng;
      const int r = pix / WLW, c = pix - r * WLW;
      const int Y = ty0 + r - 1, X = tx0 + c - 1;
      bool ok = (Y >= 0) & (Y < H) & (X >= 0) & (X < W);
      int ys = Y, xs = X;
      if (ups) {
        ok = ok & (Y & 1) & (X & 1);
        ys = (Y - 1) >> 1;
        xs = (X - 1) >> 1;
      }
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) v = ld4_fmt(x, (((size_t)b * Hs + ys) * Ws + xs) * Cin + c0 + 4 * c4, xbf);
      *reinterpret_cast<f32x4 *>(tx + pix * 16 + 4 * c4) = v;
    }
    if (vec_du) {
      for (int e = tid; e < WTH * WTW * (CP / 4); e += 256) {  // output-gradient tile, zero beyond Cout / the image
        const int c4 = e % (CP / 4), pix = e / (CP / 4);
        const int r = pix / WTW, c = pix - r * WTW;
        const int Y = ty0 + r, X = tx0 + c;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (Y < H && X < W && co0 + 4 * c4 < Cout)
          v = ld4_fmt(du, (((size_t)b * H + Y) * W + X) * Cout + co0 + 4 * c4, ubf);
        *reinterpret_cast<f32x4 *>(tu + pix * CP + 4 * c4) = v;
      }
    } else {
      for (int e = tid; e < WTH * WTW * CP; e += 256) {
        const int co = e % CP, pix = e / CP;
        const int r = pix / WTW, c = pix - r * WTW;
        const int Y = ty0 + r, X = tx0 + c;
        float v = 0.f;
        if (Y < H && X < W && co0 + co < Cout) v = ld1_fmt(du, (((size_t)b * H + Y) * W + X) * Cout + co0 + co, ubf);
        tu[e] = v;
      }
    }
    __syncthreads();
    }
    // this wave's rows: 2 of the 8; K steps of 4 consecutive pixels of a row
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
      if constexpr (BF16) {
        // v_mfma_f32_16x16x32_bf16 (gfx950): a lane's 8 k-values = its pixels col, col + 4, ..., col + 28 of the row (two of
        // the K = 16 form's quads; slot j of the A lane meets slot j of the B lane, so any shared assignment contracts right)
        static_assert(WTW % 32 == 0, "32 pixels of a row per MFMA");
        typedef short bf16x8s __attribute__((ext_vector_type(8)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        auto cat8 = [](bf16x4 lo, bf16x4 hi) { return __builtin_bit_cast(bf16x8, (bf16x8s)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)); };
#pragma unroll
        for (int s0 = 0; s0 < WTW / 32; ++s0) {
          const int col = 32 * s0 + ksub;
          bf16x8 bv[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float *q = tu + (row * WTW + col) * CP + 16 * n + m;
            bv[n] = cat8(pack_bf16(q[0], q[4 * CP], q[8 * CP], q[12 * CP]), pack_bf16(q[16 * CP], q[20 * CP], q[24 * CP], q[28 * CP]));
          }
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const float *q = tx + (row * WLW + col) * 16 + aoff[t];
            const bf16x8 av = cat8(pack_bf16(q[0], q[64], q[128], q[192]), pack_bf16(q[256], q[320], q[384], q[448]));
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[n], acc[t][n], 0, 0, 0);
          }
          const bf16x4 one4 = bf16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80};
          const bf16x8 one = cat8(one4, one4);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[MT][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(one, bv[n], acc[MT][n], 0, 0, 0);
        }
        continue;
      }
#pragma unroll 2
      for (int s = 0; s < WTW / 4; ++s) {
        const int col = 4 * s + ksub;  // this lane's pixel within the K step
        float bv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = tu[(row * WTW + col) * CP + 16 * n + m];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const float av = tx[(row * WLW + col) * 16 + aoff[t]];
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[n], acc[t][n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[MT][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, bv[n], acc[MT][n], 0, 0, 0);
      }
    }
  }
  // sum the 4 waves (K split) through LDS in wave order, then one partial per workgroup:
  // part[(blockIdx.y * gridDim.x + blockIdx.x)][10][16][CP]; D layout: rows 4*(lane>>4)+r, column lane&15
  __syncthreads();
  float *red = lds;  // 10 * 16 * CP floats <= the staging area: [tap (9 = bias)][channel of the slice][cout]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t <= MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int slot = t * 16 + 4 * ksub + r;  // D row 4 * ksub + r of tile t
            if constexpr (PACK != 0) {
              if (t < MT) {
                const int R = 16 * t + 4 * ksub + r, tap = R / PACK;
                slot = tap < 9 ? tap * 16 + (R - tap * PACK) : -1;
              } else {
                slot = 9 * 16 + 4 * ksub + r;
              }
            }
            if (slot >= 0) {
              float *d = red + slot * CP + 16 * n + m;
              *d = (w == 0 ? 0.f : *d) + acc[t][n][r];
            }
          }
    }
    __syncthreads();
  }
  float *dst = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (10 * 16 * CP);
  for (int e = tid; e < 10 * 16 * CP; e += 256) dst[e] = red[e];
}

// ---- the 8-output-channel layers (Cin = 4 or 8: the two full-resolution layers of the controller CNN, the last layers
// of the attention nets) on v_mfma_f32_4x4x1_16B_f32.  A 16x16x4 tile is 16 output channels wide, so with 8 of them half
// of every MFMA is padding (and 72 + 1 rows fill 6 tiles of 16): 38 % useful.  The 16-block form multiplies sixteen
// independent 4x4x1 outer products per instruction at the same flop rate (tools/mfma_4x4.hip: 123-133 TF/s): block b =
// lane / 4 takes pixel b of a run of 16, lane 4b + j supplies row 4 RB + j of the (tap, channel) rows as the A operand
// and output channel 4 CB + j as the B operand, and accumulator (RB, CB) collects the 4x4 block of dW for that pixel
// residue — (9 Cin + 1) / 4 x 2 blocks with no padding but the bias block's three empty rows.  The sixteen pixel
// residues are added up once at the end (xor-shuffles over the block index).  LDS records are 9 (Cin + 1) floats per
// pixel so that the sixteen pixels of a read fall into different banks.  Same persistent walk, partial layout and
// finishing kernels as wgrad_kernel. ----
template <int CIN>
__global__ __launch_bounds__(256, 2) void wgrad_small_kernel(const float *x, const float *du, int B, int Hs, int Ws, int H, int W,
                                                          int tiles_x, int tiles_y, int ntiles, float *part,
                                                          const float *const *xtab, const float *const *dutab, int Bseg,
                                                          int CP) {
  constexpr int SX = CIN + 1, SU = 9, NRB = (9 * CIN + 1 + 3) / 4;  // the last row block = the bias row + 3 empty rows
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tx = lds;                    // [WLH][WLW][SX]
  float *tu = lds + WLH * WLW * SX;   // [WTH][WTW][SU]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = lane >> 2, j = lane & 3;
  f32x4 acc[NRB][2];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) acc[rb][0] = acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  int aoff[NRB - 1];
#pragma unroll
  for (int rb = 0; rb < NRB - 1; ++rb) {
    const int R = 4 * rb + j, tap = R / CIN, ci = R - tap * CIN;
    aoff[rb] = ((tap / 3) * WLW + tap % 3) * SX + ci;
  }
  const float abias = j == 0 ? 1.0f : 0.0f;
  const int per = tiles_x * tiles_y;
  // the next tile's global loads travel in registers across the MFMA phase of the current one (as wgrad_kernel<PRE>)
  constexpr int NRX = (WLH * WLW * (CIN / 4) + 255) / 256, NRU = WTH * WTW * 2 / 256;
  f32x4 rx[NRX], ru[NRU];
  auto prefetch = [&](int tile) {
    int b = tile / per;
    const int tr = tile - b * per;
    const float *xb = x, *ub = du;
    if (xtab) {
      const int seg = b / Bseg;
      xb = xtab[seg];
      ub = dutab[seg];
      b -= seg * Bseg;
    }
    const int ty0 = (tr / tiles_x) * WTH, tx0 = (tr % tiles_x) * WTW;
#pragma unroll
    for (int i = 0; i < NRX; ++i) {
      const int e = tid + 256 * i;
      const int pix = e / (CIN / 4), c4 = e - pix * (CIN / 4);
      const int r = pix / WLW, c = pix - r * WLW;
      const int Y = ty0 + r - 1, X = tx0 + c - 1;
      const bool ok = (e < WLH * WLW * (CIN / 4)) & (Y >= 0) & (Y < H) & (X >= 0) & (X < W);
      const size_t off = ok ? (((size_t)b * Hs + Y) * Ws + X) * CIN + 4 * c4 : 0;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + off);
      rx[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NRU; ++i) {
      const int e = tid + 256 * i;
      const int pix = e >> 1, c4 = e & 1;
      const int r = pix / WTW, c = pix - r * WTW;
      const int Y = ty0 + r, X = tx0 + c;
      const bool ok = (Y < H) & (X < W);
      const size_t off = ok ? (((size_t)b * H + Y) * W + X) * 8 + 4 * c4 : 0;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(ub + off);
      ru[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's MFMA reads are complete
#pragma unroll
    for (int i = 0; i < NRX; ++i) {
      const int e = tid + 256 * i;
      const int pix = e / (CIN / 4), c4 = e - pix * (CIN / 4);
      if (e < WLH * WLW * (CIN / 4)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tx[pix * SX + 4 * c4 + k] = rx[i][k];
      }
    }
#pragma unroll
    for (int i = 0; i < NRU; ++i) {
      const int e = tid + 256 * i;
      const int pix = e >> 1, c4 = e & 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) tu[pix * SU + 4 * c4 + k] = ru[i][k];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {  // this wave's rows 2 wave, 2 wave + 1; two runs of 16 pixels per row
      const int row = wave * 2 + (g >> 1), px = 16 * (g & 1) + blk;
      const float *ax = tx + (row * WLW + px) * SX;
      const float *bu = tu + (row * WTW + px) * SU;
      const float b0 = bu[j], b1 = bu[4 + j];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const float a = rb < NRB - 1 ? ax[aoff[rb < NRB - 1 ? rb : 0]] : abias;
        acc[rb][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b0, acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b1, acc[rb][1], 0, 0, 0);
      }
    }
  }
  // the sixteen pixel residues (blocks) of an accumulator are added up by xor-shuffles over the block index; lanes 0..3
  // then hold D[4 rb + r][4 cb + lane] and put it into the wave's own copy of the partial record
  __syncthreads();
  float *red = lds;  // [wave][tap (9 = bias)][channel of the slice (16)][CP], the record as wgrad_kernel writes it
  for (int e = tid; e < 4 * 10 * 16 * CP; e += 256) red[e] = 0.f;
  __syncthreads();
  float *mine = red + wave * (10 * 16 * CP);
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[rb][cb][r];
        for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        const int R = 4 * rb + r;
        int slot = -1;
        if (R < 9 * CIN) {
          const int tap = R / CIN;
          slot = tap * 16 + (R - tap * CIN);
        } else if (R == 9 * CIN) {
          slot = 9 * 16;
        }
        if (slot >= 0 && lane < 4) mine[slot * CP + 4 * cb + lane] = v;
      }
  __syncthreads();
  float *dst = part + (size_t)blockIdx.x * (10 * 16 * CP);
  for (int e = tid; e < 10 * 16 * CP; e += 256)
    dst[e] = (red[e] + red[e + 10 * 16 * CP]) + (red[e + 2 * 10 * 16 * CP] + red[e + 3 * 10 * 16 * CP]);
}

// The filter gradient of the 8-output-channel full-resolution layers, third form.  wgrad_small_kernel (the 16-block MFMA)
// stages its tiles through registers with ~400 vector instructions per wave and tile, holds 38 accumulator tiles per lane
// (two waves per SIMD) and runs at 2.1-2.5 TB/s.  Measured on a first rewrite (transposed channel planes in LDS, one
// ds_read_b128 per four MFMAs): the launch is the SUM of its staging (190 us with 3/4 of the MFMAs compiled out) and its
// MFMAs (183 us) — vector instructions and MFMAs of a SIMD do not overlap (tools/mfma_valu.hip), and every workgroup of a
// CU is in the same phase.  So this form takes the vector instructions out of the staging: both tiles go HBM -> LDS
// directly (buffer_load_dwordx4 ... lds, pixel-major as they lie in memory, double-buffered, ONE barrier per tile), and
// the operands are read from there with scalar ds_reads: v_mfma_f32_16x16x4_f32 with M rows = (tap, ci) pairs + the bias
// row (A = 1), N = the 8 output channels (columns 8..15 repeat them and are dropped), k-slot (i, kq) = pixel 4 kq + i of a
// run of 16.  5 (Cin = 8) or 3 (Cin = 4) accumulator tiles.  Partial record per workgroup as wgrad_kernel writes it
// ([tap (9 = bias)][16][CP]): the same final reduction.
template <int CIN>
__global__ __launch_bounds__(256) void wgrad8_kernel(const float *x, const float *du, int B, int Hs, int Ws, int H, int W, int tiles_x,
                                                     int tiles_y, int ntiles, float *part, const float *const *xtab,
                                                     const float *const *dutab, int Bseg, int CP) {
  constexpr int NR = 9 * CIN + 1, NT = (NR + 15) / 16;  // rows: (tap, ci) pairs + bias; M tiles
  constexpr int IPX = CIN / 4;                           // 16-byte items per x pixel
  constexpr int NIX = WLH * WLW * IPX, NIU = WTH * WTW * 2;
  constexpr int XB = (NIX + 63) / 64 * 1024, UB = NIU * 16;  // bytes of one x / dU tile in LDS (whole 64-lane pieces)
  constexpr int NITX = (NIX + 255) / 256, NITU = NIU / 256;
  constexpr int kOOB = 0x7fffffff;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];  // [2][XB + UB], then reused for the reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kq = lane >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int aoff[NT];  // float offset of row R = 16 t + n inside the x tile, for pixel 4 kq of a run starting at tile column 0
  bool abias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int R = 16 * t + n, Rc = R < 9 * CIN ? R : 0, tap = Rc / CIN, ci = Rc - tap * CIN;
    aoff[t] = ((tap / 3) * WLW + tap % 3 + 4 * kq) * CIN + ci;
    abias[t] = R == 9 * CIN;
  }
  const int boff = 4 * kq * 8 + (n & 7);
  const int per = tiles_x * tiles_y;
  const size_t seg_imgs = xtab ? (size_t)Bseg : (size_t)B;
  const int bytes_x = (int)(seg_imgs * Hs * Ws * CIN * 4), bytes_u = (int)(seg_imgs * H * W * 8 * 4);
  auto load_tile = [&](int tile, int buf) {
    int b = tile / per;
    const int tr = tile - b * per;
    const float *xb = x, *ub = du;
    if (xtab) {
      const int seg = b / Bseg;
      xb = xtab[seg];
      ub = dutab[seg];
      b -= seg * Bseg;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xb), 0, bytes_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ub), 0, bytes_u, 0x00020000);
    const int ty0 = (tr / tiles_x) * WTH, tx0 = (tr % tiles_x) * WTW;
    unsigned char *dst = ldsb + buf * (XB + UB);
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      if (256 * it + 64 * wave >= NIX) continue;  // wave-uniform
      const int e = tid + 256 * it;
      const int pix = e / IPX, c4 = e - pix * IPX;
      const int r = pix / WLW, c = pix - r * WLW;
      const int Y = ty0 + r - 1, X = tx0 + c - 1;
      const bool ok = (e < NIX) & ((unsigned)Y < (unsigned)H) & ((unsigned)X < (unsigned)W);
      const int off = ok ? ((b * Hs + Y) * Ws + X) * CIN * 4 + 16 * c4 : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void *)(dst + (256 * it + 64 * wave) * 16), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NITU; ++it) {
      const int e = tid + 256 * it;
      const int pix = e >> 1, c4 = e & 1;
      const int r = pix / WTW, c = pix - r * WTW;
      const int Y = ty0 + r, X = tx0 + c;
      const bool ok = (Y < H) & (X < W);
      const int off = ok ? ((b * H + Y) * W + X) * 32 + 16 * c4 : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void *)(dst + XB + (256 * it + 64 * wave) * 16), 16, off, 0, 0, 0);
    }
  };
  int buf = 0;
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x, 0);
  __syncthreads();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool has_next = tile + (int)gridDim.x < ntiles;
    if (has_next) load_tile(tile + gridDim.x, buf ^ 1);  // in flight across the MFMAs; the barrier below waits for it
    const float *xs = reinterpret_cast<const float *>(ldsb + buf * (XB + UB));
    const float *us = reinterpret_cast<const float *>(ldsb + buf * (XB + UB) + XB);
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // this wave's rows 2 wave, 2 wave + 1; two runs of 16 pixels per row
      const int row = wave * 2 + (g >> 1), c0 = 16 * (g & 1);
      float bv[4], av[NT][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bv[i] = us[(row * WTW + c0 + i) * 8 + boff];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          av[t][i] = xs[(row * WLW + c0 + i) * CIN + aoff[t]];
          if (t == NT - 1 && abias[t]) av[t][i] = 1.f;  // the bias row lives in the last tile only: one select per value there
        }
#ifdef RA_W8_SKIP  // measuring aid: one MFMA per tile and group instead of four (the LDS reads stay)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32((av[t][0] + av[t][1]) + (av[t][2] + av[t][3]), (bv[0] + bv[1]) + (bv[2] + bv[3]), acc[t], 0, 0, 0);
#else
#pragma unroll
      for (int i = 0; i < 4; ++i)  // consecutive MFMAs on different accumulators: no wait for a dependent result
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][i], bv[i], acc[t], 0, 0, 0);
#endif
    }
    __syncthreads();  // the next tile has landed (vmcnt(0)) and every wave is done reading this one
    buf ^= 1;
  }
  // D lane (n, q): rows 16 t + 4 q + r, column n (= output channel for n < 8).  The four waves saw different pixels: summed
  // through LDS in a fixed order.
  float *red = reinterpret_cast<float *>(ldsb);  // [wave][tap (9 = bias)][16][CP]
  for (int e = tid; e < 4 * 10 * 16 * CP; e += 256) red[e] = 0.f;
  __syncthreads();
  float *mine = red + wave * (10 * 16 * CP);
  if (n < 8) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int R = 16 * t + 4 * kq + r;
        int slot = -1;
        if (R < 9 * CIN) {
          const int tap = R / CIN;
          slot = tap * 16 + (R - tap * CIN);
        } else if (R == 9 * CIN) {
          slot = 9 * 16;
        }
        if (slot >= 0) mine[slot * CP + n] = acc[t][r];
      }
  }
  __syncthreads();
  float *dstp = part + (size_t)blockIdx.x * (10 * 16 * CP);
  for (int e = tid; e < 10 * 16 * CP; e += 256)
    dstp[e] = (red[e] + red[e + 10 * 16 * CP]) + (red[e + 2 * 10 * 16 * CP] + red[e + 3 * 10 * 16 * CP]);
}

// ... and in the bf16 mode's stacked step (round 5: x of the 8 -> 8 layer and every dU are STORED as bf16, the first layer's x is
// the float32 packed image): wgrad8b_kernel.  The bf16-operand wgrad_kernel these launches took stages float32 tiles through
// registers and ran at 242 us (8 -> 8, 43 images) / 397 us (4 -> 8, 64 images) — the 4 -> 8 one slower than the float32 mode's
// wgrad8_kernel.  Same scheme as wgrad8_kernel — both tiles HBM -> LDS directly (16 bytes per pixel each: 8 bf16 channels, or
// the 4 float32 channels of the image), double-buffered, one barrier per tile, M rows = (tap, ci) pairs + the bias row — on
// v_mfma_f32_16x16x32_bf16 with K = the 32 pixels of one tile row: a lane's operand is 8 consecutive pixels of ONE channel,
// gathered from the pixel-major tile by 8 ds_read_b32 and 4 v_perm_b32 (bf16 tiles: the low or high half of each word) or 4
// v_cvt_pk_bf16_f32 (the float32 image: rounded to nearest even, as every bf16 operand of this mode).  5 (3) MFMAs per 32
// pixels instead of 40 (24) float32 ones, half the tile bytes.  Partial records as wgrad_kernel writes them.
typedef short s16x8t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
template <int CIN>  // 8: x stored as bf16 (16 bytes per pixel); 4: x float32 (the packed image, 16 bytes per pixel)
__global__ __launch_bounds__(256) void wgrad8b_kernel(const void *x, const void *du, int B, int Hs, int Ws, int H, int W, int tiles_x,
                                                      int tiles_y, int ntiles, float *part, const void *const *xtab,
                                                      const void *const *dutab, int Bseg, int CP) {
  constexpr int NR = 9 * CIN + 1, NT = (NR + 15) / 16;
  constexpr int NIX = WLH * WLW, NIU = WTH * WTW;            // 16-byte items: one per pixel in both tiles
  constexpr int XB = (NIX + 63) / 64 * 1024, UB = NIU * 16;  // bytes of one x / dU tile in LDS (whole 64-lane pieces)
  constexpr int NITX = (NIX + 255) / 256, NITU = NIU / 256;
  constexpr int kOOB = 0x7fffffff;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];  // [2][XB + UB], then reused for the reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kb = lane >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // row R = 16 t + n of tile t: (tap, ci) -> byte offset of (pixel 8 kb of the row, channel ci) inside the x tile, and whether
  // the bf16 element is the high half of its word
  int aoff[NT];
  unsigned asel[NT];
  bool abias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int R = 16 * t + n, Rc = R < 9 * CIN ? R : 0, tap = Rc / CIN, ci = Rc - tap * CIN;
    const int pix = (tap / 3) * WLW + tap % 3 + 8 * kb;
    aoff[t] = CIN == 8 ? pix * 16 + (ci >> 1) * 4 : pix * 16 + ci * 4;
    asel[t] = (ci & 1) ? 0x07060302u : 0x05040100u;  // v_perm_b32(hi word, lo word): the two high / the two low halves
    abias[t] = R == 9 * CIN;
  }
  const int co = n & 7;
  const int boff = 8 * kb * 16 + (co >> 1) * 4;
  const unsigned bsel = (co & 1) ? 0x07060302u : 0x05040100u;
  const int per = tiles_x * tiles_y;
  const size_t seg_imgs = xtab ? (size_t)Bseg : (size_t)B;
  const int bytes_x = (int)(seg_imgs * Hs * Ws * 16), bytes_u = (int)(seg_imgs * H * W * 16);
  auto load_tile = [&](int tile, int buf) {
    int b = tile / per;
    const int tr = tile - b * per;
    const void *xb = x, *ub = du;
    if (xtab) {
      const int seg = b / Bseg;
      xb = xtab[seg];
      ub = dutab[seg];
      b -= seg * Bseg;
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(xb), 0, bytes_x, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ub), 0, bytes_u, 0x00020000);
    const int ty0 = (tr / tiles_x) * WTH, tx0 = (tr % tiles_x) * WTW;
    unsigned char *dst = ldsb + buf * (XB + UB);
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      if (256 * it + 64 * wave >= NIX) continue;  // wave-uniform
      const int e = tid + 256 * it;
      const int r = e / WLW, c = e - r * WLW;
      const int Y = ty0 + r - 1, X = tx0 + c - 1;
      const bool ok = (e < NIX) & ((unsigned)Y < (unsigned)H) & ((unsigned)X < (unsigned)W);
      const int off = ok ? ((b * Hs + Y) * Ws + X) * 16 : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void *)(dst + (256 * it + 64 * wave) * 16), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NITU; ++it) {
      const int e = tid + 256 * it;
      const int r = e / WTW, c = e - r * WTW;
      const int Y = ty0 + r, X = tx0 + c;
      const bool ok = (Y < H) & (X < W);
      const int off = ok ? ((b * H + Y) * W + X) * 16 : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void *)(dst + XB + (256 * it + 64 * wave) * 16), 16, off, 0, 0, 0);
    }
  };
  int buf = 0;
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x, 0);
  __syncthreads();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool has_next = tile + (int)gridDim.x < ntiles;
    if (has_next) load_tile(tile + gridDim.x, buf ^ 1);  // in flight across the MFMAs; the barrier below waits for it
    const unsigned char *xs = ldsb + buf * (XB + UB);
    const unsigned char *us = xs + XB;
#pragma unroll
    for (int g = 0; g < 2; ++g) {  // this wave's rows 2 wave, 2 wave + 1: K = the row's 32 pixels
      const int row = wave * 2 + g;
      unsigned bw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bw[j] = *reinterpret_cast<const unsigned *>(us + (row * WTW + j) * 16 + boff);
      u32x4t bv;
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = __builtin_amdgcn_perm(bw[2 * j + 1], bw[2 * j], bsel);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        u32x4t av;
        if constexpr (CIN == 8) {
          unsigned aw[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) aw[j] = *reinterpret_cast<const unsigned *>(xs + (row * WLW + j) * 16 + aoff[t]);
#pragma unroll
          for (int j = 0; j < 4; ++j) av[j] = __builtin_amdgcn_perm(aw[2 * j + 1], aw[2 * j], asel[t]);
        } else {
          float af[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) af[j] = *reinterpret_cast<const float *>(xs + (row * WLW + j) * 16 + aoff[t]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            typedef float f32x2c __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
            av[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2c{af[2 * j], af[2 * j + 1]}, bf16x2c));
          }
        }
        if (t == NT - 1 && abias[t]) av = u32x4t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};  // the bias row: A = 1
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8t, av), __builtin_bit_cast(bf16x8t, bv), acc[t], 0, 0, 0);
      }
    }
    __syncthreads();  // the next tile has landed (vmcnt(0)) and every wave is done reading this one
    buf ^= 1;
  }
  float *red = reinterpret_cast<float *>(ldsb);  // [wave][tap (9 = bias)][16][CP]
  for (int e = tid; e < 4 * 10 * 16 * CP; e += 256) red[e] = 0.f;
  __syncthreads();
  float *mine = red + wave * (10 * 16 * CP);
  if (n < 8) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int R = 16 * t + 4 * kb + r;
        int slot = -1;
        if (R < 9 * CIN) {
          const int tap = R / CIN;
          slot = tap * 16 + (R - tap * CIN);
        } else if (R == 9 * CIN) {
          slot = 9 * 16;
        }
        if (slot >= 0) mine[slot * CP + n] = acc[t][r];
      }
  }
  __syncthreads();
  float *dstp = part + (size_t)blockIdx.x * (10 * 16 * CP);
  for (int e = tid; e < 10 * 16 * CP; e += 256)
    dstp[e] = (red[e] + red[e + 10 * 16 * CP]) + (red[e + 2 * 10 * 16 * CP] + red[e + 3 * 10 * 16 * CP]);
}

// dW[tap][ci][co] (= TF [3,3,Cin,Cout]) and db[co] from the partials, fixed order.
__global__ __launch_bounds__(256) void wgrad_final_kernel(const float *part, int nwg, int nchunks, int CP, int Cin, int Cout,
                                                          float *dw, float *db) {  // CP = couts per slice
  // 4 output elements per workgroup, one wave each: the 64 lanes stride over the partials, then a
  // fixed butterfly (deterministic)
  const int total = 9 * Cin * Cout + Cout;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= total) return;
  int tap, ci, co;
  if (e < 9 * Cin * Cout) {
    co = e % Cout;
    ci = (e / Cout) % Cin;
    tap = e / (Cout * Cin);
  } else {
    tap = 9;
    ci = 0;
    co = e - 9 * Cin * Cout;
  }
  const int chunk = ci / 16, cl = ci % 16, slice = co / CP, cs = co % CP;
  float s = 0.f;
  for (int k = lane; k < nwg; k += 64)
    s += part[((((size_t)slice * nchunks + chunk) * nwg + k) * 10 + tap) * 16 * CP + cl * CP + cs];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    if (tap < 9) dw[e] = s;
    else if (db) db[co] = s;
  }
}

// ---- LSTM cell pointwise part (nnlib.py:641-646): pre [B][4*hid] = the four gate pre-activations
// (i, f, o, u), c_prev [B][hid]  ->  c = f c_prev + i u,  h = o tanh(c).  act keeps the gate values
// for the backward; one thread per (image, unit).
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float *pre, const float *c_prev, int n, int hid, float *h,
                                                        float *c, float *act) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int b = idx / hid, j = idx - b * hid;
  const float *p = pre + (size_t)b * 4 * hid + j;
  const float gi = 1.f / (1.f + expf(-p[0])), gf = 1.f / (1.f + expf(-p[hid])), go = 1.f / (1.f + expf(-p[2 * hid])),
              gu = tanhf(p[3 * hid]);
  const float cn = gf * c_prev[idx] + gi * gu;
  float *a = act + (size_t)b * 4 * hid + j;
  a[0] = gi;
  a[hid] = gf;
  a[2 * hid] = go;
  a[3 * hid] = gu;
  c[idx] = cn;
  h[idx] = go * tanhf(cn);
}
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float *act, const float *c_prev, const float *c,
                                                            const float *dh, const float *dc, int n, int hid, float *dpre,
                                                            float *dc_prev) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int b = idx / hid, j = idx - b * hid;
  const float *a = act + (size_t)b * 4 * hid + j;
  const float gi = a[0], gf = a[hid], go = a[2 * hid], gu = a[3 * hid];
  const float tc = tanhf(c[idx]);
  const float gh = dh ? dh[idx] : 0.f;
  const float dcn = (dc ? dc[idx] : 0.f) + gh * go * (1.f - tc * tc);
  float *d = dpre + (size_t)b * 4 * hid + j;
  d[0] = dcn * gu * gi * (1.f - gi);
  d[hid] = dcn * c_prev[idx] * gf * (1.f - gf);
  d[2 * hid] = gh * tc * go * (1.f - go);
  d[3 * hid] = dcn * gi * (1.f - gu * gu);
  dc_prev[idx] = dcn * gf;
}

// ---- Gaussian filter bank (modellib.py:581-612): F[b][l][j] = N(l; mu_j, var), mu_j = ctr + (size + 1) / NF * (j - (NF - 1) / 2),
// var = exp(lg_var); its adjoint reduces over the whole [L, NF] bank of an image: one workgroup per image.
// sc / ss / sv: elements between consecutive images in ctr / size / lg_var (1: a dense [B] vector; the training graph
// hands in one column of a [B,2] tensor without copying it out)
__global__ __launch_bounds__(256) void gauss_filter_kernel(const float *ctr, const float *size, const float *lg_var, int sc,
                                                           int ss, int sv, int L, int NF, float *out) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= L * NF) return;
  const int l = e / NF, j = e - l * NF;
  const float var = expf(lg_var[(size_t)b * sv]);
  const float mu = ctr[(size_t)b * sc] + (size[(size_t)b * ss] + 1.0f) / (float)NF * ((float)j - 0.5f * (float)(NF - 1));
  const float dd = (float)l - mu;
  out[(size_t)b * L * NF + e] = expf(-0.5f * dd * dd / var) / (sqrtf(var) * 2.5066282746310002f);
}
__global__ __launch_bounds__(256) void gauss_filter_bwd_kernel(const float *ctr, const float *size, const float *lg_var,
                                                               int sc, int ss, int sv, const float *g, int L, int NF,
                                                               float *dctr, float *dsize, float *dlgv, int sd) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  const float var = expf(lg_var[(size_t)b * sv]), c0 = ctr[(size_t)b * sc], step = (size[(size_t)b * ss] + 1.0f) / (float)NF;
  const float norm = 1.0f / (sqrtf(var) * 2.5066282746310002f);
  float a_ctr = 0.f, a_size = 0.f, a_var = 0.f;
  for (int e = threadIdx.x; e < L * NF; e += 256) {
    const int l = e / NF, j = e - l * NF;
    const float off = (float)j - 0.5f * (float)(NF - 1);
    const float dd = (float)l - (c0 + step * off);
    const float f = expf(-0.5f * dd * dd / var) * norm;
    const float gf = g[(size_t)b * L * NF + e] * f;
    const float dmu = gf * dd / var;
    a_ctr += dmu;
    a_size += dmu * off / (float)NF;
    a_var += gf * (0.5f * dd * dd / (var * var) - 0.5f / var);
  }
  a_ctr = block_sum256(a_ctr, red);
  a_size = block_sum256(a_size, red);
  a_var = block_sum256(a_var, red);
  if (threadIdx.x == 0) {
    dctr[(size_t)b * sd] = a_ctr;
    dsize[(size_t)b * sd] = a_size;
    dlgv[(size_t)b * sd] = a_var * var;
  }
}

// The same reduction, ADDED to the filter's gradient in the reference's own layout (the gradient
// bucket): [3,3,cin_w,Cout], or [3,3,Cout,cin_w] with the taps flipped for a transposed (dcnn) layer;
// chan_map sends a packed kernel channel to its filter row (-1: padding).  One writer per element.
// COALESCED reads: a lane owns one element of the record (64 consecutive floats per wave load), the four waves of a
// workgroup split the partial records and meet in LDS in a fixed order.  The wave-per-element form (wgrad_final_kernel) gives
// every lane its own record — 64 separate 4-byte loads per instruction: 96 / 107 us for the 64-channel layers' 2 048 records.
// grid (ceil(160 CP / 64), nchunks, slices).
__global__ __launch_bounds__(256) void wgrad_final_acc_rows_kernel(const float *part, int nwg, int nchunks, int CP, int Cin, int Cout,
                                                                   const int *chan_map, int cin_w, int transposed, float *gw,
                                                                   float *gb) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rec = 160 * CP;
  const int r = blockIdx.x * 64 + lane, chunk = blockIdx.y, slice = blockIdx.z;
  float s = 0.f;
  if (r < rec) {
    const float *p = part + ((size_t)slice * nchunks + chunk) * nwg * rec + r;
    int k = wave;
    for (; k + 28 < nwg; k += 32) {  // eight loads in flight
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + 4 * u) * rec];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nwg; k += 4) s += p[(size_t)k * rec];
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || r >= rec) return;
  s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  const int tap = r / (16 * CP), cl = (r / CP) & 15, cs = r % CP;
  const int ci = chunk * 16 + cl, co = slice * CP + cs;
  if (co >= Cout) return;
  if (tap == 9) {
    if (gb && chunk == 0 && cl == 0) gb[co] += s;
    return;
  }
  if (ci >= Cin) return;
  const int j = chan_map ? chan_map[ci] : (ci < cin_w ? ci : -1);
  if (j < 0) return;
  const int ky = tap / 3, kx = tap - 3 * ky;
  const size_t idx = transposed ? ((size_t)((2 - ky) * 3 + (2 - kx)) * Cout + co) * cin_w + j
                                : ((size_t)(ky * 3 + kx) * cin_w + j) * Cout + co;
  gw[idx] += s;
}

}  // namespace train
}  // namespace ra

namespace {
inline int wgrad_grid_x(int ntiles) {
  // persistent workgroups: 4 per CU (38 KB of LDS each) hide the un-prefetched tile staging; 256 left 4 waves per CU
  static int cap = 0;
  if (!cap) {
    const char *e = getenv("RA_WGRAD_WGS");
    cap = e ? atoi(e) : 1024;
    if (cap < 1) cap = 1;
  }
  return ntiles < cap ? ntiles : cap;
}
}  // namespace

extern "C" size_t ra_conv3x3_wgrad_workspace_floats(int Cin, int Cout, int B, int H, int W) {
  const int cp = ra_conv_cout_padded(Cout);
  if (!cp || Cin <= 0 || B <= 0) return 0;
  const int ntiles = ceil_div(W, ra::train::WTW) * ceil_div(H, ra::train::WTH) * B;
  const int per = cp < 64 ? cp : 64;
  return (size_t)(cp / per) * ceil_div(Cin, 16) * wgrad_grid_x(ntiles) * 10 * 16 * per;
}

namespace {
// acc == false: dw / db are written in the kernel's own [3,3,Cin,Cout] / [Cout] layout; acc == true: the
// sums are added to gw / gb in the reference layout (chan_map, cin_w, transposed as in wgrad_final_acc_rows_kernel)
int wgrad_impl(const float *x, int Cin, int B, int Hs, int Ws, int upsample, const float *du, int Cout, float *ws,
               size_t ws_floats, float *dw, float *db, bool acc, const int *chan_map, int cin_w, int transposed,
               void *stream, bool bf16 = false, const float *const *xtab = nullptr, const float *const *dutab = nullptr,
               int Bseg = 0, int fmt = 0) {
  if (fmt && !bf16) return fail(RA_E_INVALID, "ra_conv3x3_wgrad: bf16 storage needs the bf16-operand kernels");
  if (xtab) x = du = reinterpret_cast<const float *>(xtab);  // (not read: the tables are)
  if (!x || !du || !ws || !dw || B <= 0 || Hs <= 0 || Ws <= 0 || Cin <= 0 || Cout <= 0)
    return fail(RA_E_INVALID, "ra_conv3x3_wgrad_f32: bad argument");
  const int cp = ra_conv_cout_padded(Cout);
  if (Cin % 4 || !cp) return fail(RA_E_SHAPE, "ra_conv3x3_wgrad_f32: Cin %d %% 4 or Cout %d", Cin, Cout);
  const int per = cp < 64 ? cp : 64, slices = cp / per;  // output channels per workgroup
  const int ups = upsample ? 1 : 0, H = Hs * (1 + ups), W = Ws * (1 + ups);
  if (ws_floats < ra_conv3x3_wgrad_workspace_floats(Cin, Cout, B, H, W))
    return fail(RA_E_WORKSPACE, "ra_conv3x3_wgrad_f32: workspace too small");
  using namespace ra::train;
  const int tiles_x = ceil_div(W, WTW), tiles_y = ceil_div(H, WTH), ntiles = tiles_x * tiles_y * B;
  const int gx = wgrad_grid_x(ntiles), chunks = ceil_div(Cin, 16);
  const size_t lds_stage = (size_t)(WLH * WLW * 16 + WTH * WTW * per) * sizeof(float);
  const size_t lds_red = (size_t)10 * 16 * per * sizeof(float);
  const size_t lds = lds_stage > lds_red ? lds_stage : lds_red;
  hipStream_t st = as_stream(stream);
#define RA_WGRAD_T(NT, PACK, BF, PRE)                                                                             \
  {                                                                                                               \
    static bool attr = false;                                                                                     \
    if (!attr) {                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel<NT, PACK, BF, PRE>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);                         \
      attr = true;                                                                                                \
    }                                                                                                             \
    hipLaunchKernelGGL((wgrad_kernel<NT, PACK, BF, PRE>), dim3(gx, chunks, slices), dim3(256), lds, st, x, du, B, Hs, Ws, Cin, \
                       ups, H, W, Cout, tiles_x, tiles_y, ntiles, ws, xtab, dutab, Bseg, fmt);                   \
  }
#define RA_WGRAD_P(NT, PACK, BF)                                                                                  \
  {                                                                                                               \
    if (NT <= 2 && pre_ok) RA_WGRAD_T(NT, PACK, BF, (NT <= 2)) else RA_WGRAD_T(NT, PACK, BF, false)               \
  }
#define RA_WGRAD(NT, PACK)                                                                                        \
  {                                                                                                               \
    if (bf16) RA_WGRAD_P(NT, PACK, true) else RA_WGRAD_P(NT, PACK, false)                                         \
  }
  static int pre_env = -1;  // RA_WGRAD_PRE=0: tuning aid, no register prefetch of the next tile
  if (pre_env < 0) {
    const char *e = getenv("RA_WGRAD_PRE");
    pre_env = e ? atoi(e) : 1;
  }
  const bool pre_ok = pre_env && (Cout & 3) == 0;
  static int pack_ok = -1;  // RA_WGRAD_PACK=0: tuning aid, channel rows for every Cin
  if (pack_ok < 0) {
    const char *e = getenv("RA_WGRAD_PACK");
    pack_ok = e ? atoi(e) : 1;
  }
  const int pack = (pack_ok && (Cin == 4 || Cin == 8)) ? Cin : 0;
  static int small_ok = -1;  // RA_WGRAD_SMALL=0: tuning aid, the 16x16x4 form for the 8-output-channel layers too
  if (small_ok < 0) {
    const char *e = getenv("RA_WGRAD_SMALL");
    small_ok = e ? atoi(e) : 1;
  }
  const bool small = small_ok && !bf16 && !ups && Cout == 8 && (Cin == 4 || Cin == 8);
  static int t8_ok = -1;  // RA_WGRAD8=0: tuning aid, the 16-block form (wgrad_small_kernel) instead of the transposed-tile one
  if (t8_ok < 0) {
    const char *e = getenv("RA_WGRAD8");
    t8_ok = e ? atoi(e) : 1;
  }
  const size_t seg_bytes = (size_t)(xtab ? Bseg : B) * H * W * 8 * 4;  // the larger of the two tensors of a segment
  // bf16 mode, stacked step: dU stored as bf16 and x either stored as bf16 (8 channels) or the float32 packed image (4 channels)
  const bool small_b = small_ok && t8_ok && bf16 && !ups && Cout == 8 && (fmt & 2) &&
                       ((Cin == 8 && (fmt & 1)) || (Cin == 4 && !(fmt & 1))) && seg_bytes < (1ull << 31) && Hs == H && Ws == W;
  if (small_b) {
    const size_t lds_s = 2 * (size_t)(((WLH * WLW + 63) / 64) * 1024 + WTH * WTW * 16);
    const size_t lds_8 = lds_s > 4 * lds_red ? lds_s : 4 * lds_red;
    const void *const *xt = reinterpret_cast<const void *const *>(xtab), *const *ut = reinterpret_cast<const void *const *>(dutab);
    if (Cin == 4)
      hipLaunchKernelGGL(wgrad8b_kernel<4>, dim3(gx), dim3(256), lds_8, st, (const void *)x, (const void *)du, B, Hs, Ws, H, W, tiles_x, tiles_y,
                         ntiles, ws, xt, ut, Bseg, per);
    else
      hipLaunchKernelGGL(wgrad8b_kernel<8>, dim3(gx), dim3(256), lds_8, st, (const void *)x, (const void *)du, B, Hs, Ws, H, W, tiles_x, tiles_y,
                         ntiles, ws, xt, ut, Bseg, per);
  } else if (small && t8_ok && seg_bytes < (1ull << 31) && Hs == H && Ws == W) {
    const size_t lds_s = 2 * (size_t)(((WLH * WLW * (Cin / 4) + 63) / 64) * 1024 + WTH * WTW * 32);
    const size_t lds_8 = lds_s > 4 * lds_red ? lds_s : 4 * lds_red;
    if (Cin == 4)
      hipLaunchKernelGGL(wgrad8_kernel<4>, dim3(gx), dim3(256), lds_8, st, x, du, B, Hs, Ws, H, W, tiles_x, tiles_y, ntiles, ws, xtab,
                         dutab, Bseg, per);
    else
      hipLaunchKernelGGL(wgrad8_kernel<8>, dim3(gx), dim3(256), lds_8, st, x, du, B, Hs, Ws, H, W, tiles_x, tiles_y, ntiles, ws, xtab,
                         dutab, Bseg, per);
  } else if (small) {
    const size_t lds_s = (size_t)(WLH * WLW * (Cin + 1) + WTH * WTW * 9) * sizeof(float);
    const size_t lds_small = lds_s > 4 * lds_red ? lds_s : 4 * lds_red;  // the four waves' partial records at the end
    if (Cin == 4)
      hipLaunchKernelGGL(wgrad_small_kernel<4>, dim3(gx), dim3(256), lds_small, st, x, du, B, Hs, Ws, H, W, tiles_x, tiles_y,
                         ntiles, ws, xtab, dutab, Bseg, per);
    else
      hipLaunchKernelGGL(wgrad_small_kernel<8>, dim3(gx), dim3(256), lds_small, st, x, du, B, Hs, Ws, H, W, tiles_x, tiles_y,
                         ntiles, ws, xtab, dutab, Bseg, per);
  } else
  switch (per / 16) {
    case 1:
      if (pack == 4) RA_WGRAD(1, 4) else if (pack == 8) RA_WGRAD(1, 8) else RA_WGRAD(1, 0)
      break;
    case 2:
      if (pack == 4) RA_WGRAD(2, 4) else if (pack == 8) RA_WGRAD(2, 8) else RA_WGRAD(2, 0)
      break;
    default: RA_WGRAD(4, 0) break;
  }
#undef RA_WGRAD
#undef RA_WGRAD_P
#undef RA_WGRAD_T
  const int total = 9 * Cin * Cout + Cout;
  if (acc)
    hipLaunchKernelGGL(wgrad_final_acc_rows_kernel, dim3(ceil_div(160 * per, 64), chunks, slices), dim3(256), 0, st, ws, gx, chunks, per, Cin,
                       Cout, chan_map, cin_w, transposed, dw, db);
  else
    hipLaunchKernelGGL(wgrad_final_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, st, ws, gx, chunks, per, Cin, Cout, dw, db);
  return launch_status("ra_conv3x3_wgrad_f32");
}
}  // namespace

extern "C" int ra_conv3x3_wgrad_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample, const float *du,
                                    int Cout, float *ws, size_t ws_floats, float *dw, float *db, void *stream) {
  return wgrad_impl(x, Cin, B, Hs, Ws, upsample, du, Cout, ws, ws_floats, dw, db, false, nullptr, Cin, 0, stream);
}

extern "C" int ra_conv3x3_wgrad_acc_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample, const float *du,
                                        int Cout, float *ws, size_t ws_floats, const int *chan_map, int cin_w,
                                        int transposed, float *gw, float *gb, void *stream) {
  if (cin_w <= 0 || (!chan_map && cin_w > Cin)) return fail(RA_E_INVALID, "ra_conv3x3_wgrad_acc_f32: cin_w %d", cin_w);
  return wgrad_impl(x, Cin, B, Hs, Ws, upsample, du, Cout, ws, ws_floats, gw, gb, true, chan_map, cin_w, transposed ? 1 : 0,
                    stream);
}

extern "C" int ra_conv3x3_wgrad_bf16ops_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample, const float *du,
                                            int Cout, float *ws, size_t ws_floats, float *dw, float *db, void *stream) {
  return wgrad_impl(x, Cin, B, Hs, Ws, upsample, du, Cout, ws, ws_floats, dw, db, false, nullptr, Cin, 0, stream, true);
}

extern "C" int ra_conv3x3_wgrad_acc_bf16ops_f32(const float *x, int Cin, int B, int Hs, int Ws, int upsample, const float *du,
                                                int Cout, float *ws, size_t ws_floats, const int *chan_map, int cin_w,
                                                int transposed, float *gw, float *gb, void *stream) {
  if (cin_w <= 0 || (!chan_map && cin_w > Cin)) return fail(RA_E_INVALID, "ra_conv3x3_wgrad_acc_bf16ops_f32: cin_w %d", cin_w);
  return wgrad_impl(x, Cin, B, Hs, Ws, upsample, du, Cout, ws, ws_floats, gw, gb, true, chan_map, cin_w, transposed ? 1 : 0,
                    stream, true);
}

// ... and with the tensors stored as bf16 (the bf16 mode's layers between themselves): fmt bit 0 = x, bit 1 = du
extern "C" int ra_conv3x3_wgrad_acc_bf16_f32(const void *x, int Cin, int B, int Hs, int Ws, int upsample, const void *du, int Cout,
                                             float *ws, size_t ws_floats, const int *chan_map, int cin_w, int transposed, float *gw,
                                             float *gb, int fmt, void *stream) {
  if (cin_w <= 0 || (!chan_map && cin_w > Cin)) return fail(RA_E_INVALID, "ra_conv3x3_wgrad_acc_bf16_f32: cin_w %d", cin_w);
  return wgrad_impl(static_cast<const float *>(x), Cin, B, Hs, Ws, upsample, static_cast<const float *>(du), Cout, ws, ws_floats, gw, gb,
                    true, chan_map, cin_w, transposed ? 1 : 0, stream, true, nullptr, nullptr, 0, fmt & 3);
}

namespace ra {
namespace train {
struct PtrTable {
  const float *p[64];
};
__global__ __launch_bounds__(64) void ptr_table_kernel(const PtrTable t, int n, const float **out) {
#pragma unroll
  for (int i = 0; i < 64; ++i)
    if ((int)threadIdx.x == i && i < n) out[i] = t.p[i];
}
}  // namespace train
}  // namespace ra

// Up to 64 device pointers (a HOST array) -> a device table, as a kernel launch (capturable in a HIP graph, where a
// host-to-device copy of pageable memory is not): the segment tables of ra_conv3x3_wgrad_multi_acc_f32.
extern "C" int ra_ptr_table(const void *const *host_ptrs, int n, void **dev_table, void *stream) {
  if (!host_ptrs || !dev_table || n <= 0 || n > 64) return fail(RA_E_INVALID, "ra_ptr_table: 1..64 pointers");
  ra::train::PtrTable t{};
  for (int i = 0; i < n; ++i) t.p[i] = static_cast<const float *>(host_ptrs[i]);
  hipLaunchKernelGGL(ra::train::ptr_table_kernel, dim3(1), dim3(64), 0, as_stream(stream), t, n,
                     (const float **)dev_table);
  return launch_status("ra_ptr_table");
}

// The filter gradient of a layer over the images of SEVERAL calls (its T timesteps in a training step) in one pass:
// xtab / dutab are device tables of nseg pointers to the calls' x [Bseg,Hs,Ws,Cin] and du [Bseg,H,W,Cout].
// bf16_operands: bit 0 = bf16 operands on the bf16 MFMA; with it, bit 1 = the x tensors are stored as bf16, bit 2 = the du tensors.
extern "C" int ra_conv3x3_wgrad_multi_acc_f32(const void *const *xtab, const void *const *dutab, int nseg, int Cin, int Bseg,
                                              int Hs, int Ws, int upsample, int Cout, float *ws, size_t ws_floats,
                                              const int *chan_map, int cin_w, int transposed, float *gw, float *gb,
                                              int bf16_operands, void *stream) {
  if (!xtab || !dutab || nseg <= 0 || Bseg <= 0) return fail(RA_E_INVALID, "ra_conv3x3_wgrad_multi_acc_f32: bad argument");
  if (cin_w <= 0 || (!chan_map && cin_w > Cin)) return fail(RA_E_INVALID, "ra_conv3x3_wgrad_multi_acc_f32: cin_w %d", cin_w);
  return wgrad_impl(nullptr, Cin, nseg * Bseg, Hs, Ws, upsample, nullptr, Cout, ws, ws_floats, gw, gb, true, chan_map, cin_w,
                    transposed ? 1 : 0, stream, (bf16_operands & 1) != 0, reinterpret_cast<const float *const *>(xtab),
                    reinterpret_cast<const float *const *>(dutab), Bseg, (bf16_operands >> 1) & 3);
}

extern "C" int ra_lstm_cell_f32(const float *pre, const float *c_prev, int B, int hid, float *h, float *c, float *act,
                                void *stream) {
  if (!pre || !c_prev || !h || !c || !act || B <= 0 || hid <= 0) return fail(RA_E_INVALID, "ra_lstm_cell_f32: bad argument");
  const int n = B * hid;
  hipLaunchKernelGGL(train::lstm_cell_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), pre, c_prev, n, hid, h, c,
                     act);
  return launch_status("ra_lstm_cell_f32");
}

extern "C" int ra_lstm_cell_bwd_f32(const float *act, const float *c_prev, const float *c, const float *dh, const float *dc,
                                    int B, int hid, float *dpre, float *dc_prev, void *stream) {
  if (!act || !c_prev || !c || !dpre || !dc_prev || B <= 0 || hid <= 0)
    return fail(RA_E_INVALID, "ra_lstm_cell_bwd_f32: bad argument");
  const int n = B * hid;
  hipLaunchKernelGGL(train::lstm_cell_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), act, c_prev, c, dh,
                     dc, n, hid, dpre, dc_prev);
  return launch_status("ra_lstm_cell_bwd_f32");
}

extern "C" int ra_gauss_filter_strided_f32(const float *ctr, const float *size, const float *lg_var, int stride_ctr,
                                           int stride_size, int stride_lg_var, int B, int L, int NF, float *out, void *stream) {
  if (!ctr || !size || !lg_var || !out || B <= 0 || L <= 0 || NF <= 0 || stride_ctr < 1 || stride_size < 1 || stride_lg_var < 0)
    return fail(RA_E_INVALID, "ra_gauss_filter_f32: bad argument");
  hipLaunchKernelGGL(train::gauss_filter_kernel, dim3(ceil_div(L * NF, 256), B), dim3(256), 0, as_stream(stream), ctr, size,
                     lg_var, stride_ctr, stride_size, stride_lg_var, L, NF, out);
  return launch_status("ra_gauss_filter_f32");
}
extern "C" int ra_gauss_filter_f32(const float *ctr, const float *size, const float *lg_var, int B, int L, int NF, float *out,
                                   void *stream) {
  return ra_gauss_filter_strided_f32(ctr, size, lg_var, 1, 1, 1, B, L, NF, out, stream);
}

extern "C" int ra_gauss_filter_strided_bwd_f32(const float *ctr, const float *size, const float *lg_var, int stride_ctr,
                                               int stride_size, int stride_lg_var, const float *g, int B, int L, int NF,
                                               float *dctr, float *dsize, float *dlg_var, int stride_grad, void *stream) {
  if (!ctr || !size || !lg_var || !g || !dctr || !dsize || !dlg_var || B <= 0 || L <= 0 || NF <= 0 || stride_ctr < 1 ||
      stride_size < 1 || stride_lg_var < 0 || stride_grad < 1)
    return fail(RA_E_INVALID, "ra_gauss_filter_bwd_f32: bad argument");
  hipLaunchKernelGGL(train::gauss_filter_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), ctr, size, lg_var, stride_ctr,
                     stride_size, stride_lg_var, g, L, NF, dctr, dsize, dlg_var, stride_grad);
  return launch_status("ra_gauss_filter_bwd_f32");
}
extern "C" int ra_gauss_filter_bwd_f32(const float *ctr, const float *size, const float *lg_var, const float *g, int B, int L,
                                       int NF, float *dctr, float *dsize, float *dlg_var, void *stream) {
  return ra_gauss_filter_strided_bwd_f32(ctr, size, lg_var, 1, 1, 1, g, B, L, NF, dctr, dsize, dlg_var, 1, stream);
}

// ---- the attention head of the training graph: controller output -> attention parameters (full_model.py:702-722,
// modellib.py:752-764,812-825), and the ground-truth knob on them (full_model.py:744-773).  Scalar math on nine numbers
// per image; as separate torch ops it was ~45 launches per timestep forward + backward. ----
namespace ra {
namespace train {
constexpr int kHeadStride = 16;  // floats per image in the head's output record
// record: [0,1] cn  [2,3] ls  [4,5] ctr  [6,7] size  [8,9] lg_var  [10] attn_gamma  [11] box_gamma  [12] y_lg_gamma
__device__ inline float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch.nn.functional.softplus
__global__ __launch_bounds__(64) void attn_head_kernel(const float *co, int sco, int B, float H, float W, float Fh, float Fw,
                                                       int flags, float *out, float *arec) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float *c = co + (size_t)b * sco;
  float *o = out + (size_t)b * kHeadStride;
  const bool squash = flags & 1, fixed_var = flags & 2, dynamic_var = flags & 4, fixed_gamma = flags & 8;
  const float dim[2] = {H, W}, fs[2] = {Fh, Fw};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float cn = squash ? tanhf(c[k]) : c[k];
    const float ls = squash ? -softplus_t(c[2 + k]) : c[2 + k];
    const float ctr = (cn + 1.0f) * dim[k] / 2.0f, size = expf(ls) * dim[k];
    float lv = fixed_var ? 0.f : logf(size) - logf(fs[k]);
    if (dynamic_var) lv = c[4 + k];
    o[k] = cn;
    o[2 + k] = ls;
    o[4 + k] = ctr;
    o[6 + k] = size;
    o[8 + k] = lv;
  }
  o[10] = fixed_gamma ? 1.f : expf(c[6]);
  o[11] = expf(c[7]);
  o[12] = fixed_gamma ? 2.f : c[8];
  if (arec) {  // the same window as the resample kernels' attention record (ctr, size, lg_var, the three gammas, zeros)
    float *r = arec + (size_t)b * RA_ATTN_STRIDE;
#pragma unroll
    for (int k = 0; k < 2; ++k) r[k] = o[4 + k], r[2 + k] = o[6 + k], r[4 + k] = o[8 + k];
    r[6] = o[10], r[7] = o[11], r[8] = o[12];
#pragma unroll
    for (int k = 9; k < RA_ATTN_STRIDE; ++k) r[k] = 0.f;
  }
}
// g_*: gradients of the record's fields, each nullable (no gradient = zero), dense [B,2] / [B]
__global__ __launch_bounds__(64) void attn_head_bwd_kernel(const float *co, int sco, const float *out, const float *g_cn,
                                                           const float *g_ls, const float *g_ctr, const float *g_size,
                                                           const float *g_lv, const float *g_ag, const float *g_bg,
                                                           const float *g_ylg, int B, float H, float W, int flags, float *dco) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float *c = co + (size_t)b * sco, *o = out + (size_t)b * kHeadStride;
  float *d = dco + (size_t)b * 9;
  const bool squash = flags & 1, fixed_var = flags & 2, dynamic_var = flags & 4, fixed_gamma = flags & 8;
  const float dim[2] = {H, W};
  auto at2 = [&](const float *g, int k) { return g ? g[2 * b + k] : 0.f; };
  auto at1 = [&](const float *g) { return g ? g[b] : 0.f; };
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float cn = o[k], size = o[6 + k];
    const float d_cn = at2(g_cn, k) + at2(g_ctr, k) * dim[k] / 2.0f;
    // size = exp(ls) dim, lg_var = ls + log(dim / F): both follow ls
    const float d_ls = at2(g_ls, k) + at2(g_size, k) * size + ((fixed_var || dynamic_var) ? 0.f : at2(g_lv, k));
    d[k] = squash ? d_cn * (1.0f - cn * cn) : d_cn;
    const float x = c[2 + k];
    d[2 + k] = squash ? d_ls * (x > 20.f ? -1.0f : -1.0f / (1.0f + expf(-x))) : d_ls;
    d[4 + k] = dynamic_var ? at2(g_lv, k) : 0.f;
  }
  d[6] = fixed_gamma ? 0.f : at1(g_ag) * o[10];
  d[7] = at1(g_bg) * o[11];
  d[8] = fixed_gamma ? 0.f : at1(g_ylg);
}
// p2 = kb m + (1 - kb) p for the window centre and size, m = sum_t match[b][t] gt[b][t][:] (the matched noisy GT box)
__global__ __launch_bounds__(64) void knob_mix_kernel(const float *ctr, const float *size, const float *match, const float *ctr_gt,
                                                      const float *size_gt, const float *kb, int skb, int sp, int B, int T,
                                                      float *ctr2, float *size2, const float *arec, float *arec2) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  float mc[2] = {0.f, 0.f}, ms[2] = {0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const float w = match[(size_t)b * T + t];
    mc[0] += w * ctr_gt[((size_t)b * T + t) * 2];
    mc[1] += w * ctr_gt[((size_t)b * T + t) * 2 + 1];
    ms[0] += w * size_gt[((size_t)b * T + t) * 2];
    ms[1] += w * size_gt[((size_t)b * T + t) * 2 + 1];
  }
  const float k = kb[(size_t)b * skb];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ctr2[2 * b + i] = k * mc[i] + (1.0f - k) * ctr[(size_t)b * sp + i];
    size2[2 * b + i] = k * ms[i] + (1.0f - k) * size[(size_t)b * sp + i];
  }
  if (arec2) {  // the attention record with the mixed window (variance and gammas as predicted)
    const float *r = arec + (size_t)b * RA_ATTN_STRIDE;
    float *r2 = arec2 + (size_t)b * RA_ATTN_STRIDE;
#pragma unroll
    for (int i = 0; i < 2; ++i) r2[i] = ctr2[2 * b + i], r2[2 + i] = size2[2 * b + i];
#pragma unroll
    for (int i = 4; i < RA_ATTN_STRIDE; ++i) r2[i] = r[i];
  }
}
__global__ __launch_bounds__(64) void knob_mix_bwd_kernel(const float *g_ctr2, const float *g_size2, const float *kb, int skb, int B,
                                                          float *d_ctr, float *d_size) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float k = 1.0f - kb[(size_t)b * skb];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    d_ctr[2 * b + i] = g_ctr2 ? k * g_ctr2[2 * b + i] : 0.f;
    d_size[2 * b + i] = g_size2 ? k * g_size2[2 * b + i] : 0.f;
  }
}
}  // namespace train
}  // namespace ra

extern "C" int ra_attn_head_rec_f32(const float *ctrl_out, int stride, int B, int H, int W, int Fh, int Fw, int flags, float *out,
                                    float *attn_rec, void *stream) {
  if (!ctrl_out || !out || B <= 0 || stride < 9) return fail(RA_E_INVALID, "ra_attn_head_f32: bad argument");
  hipLaunchKernelGGL(train::attn_head_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), ctrl_out, stride, B, (float)H,
                     (float)W, (float)Fh, (float)Fw, flags, out, attn_rec);
  return launch_status("ra_attn_head_f32");
}
extern "C" int ra_attn_head_f32(const float *ctrl_out, int stride, int B, int H, int W, int Fh, int Fw, int flags, float *out,
                                void *stream) {
  return ra_attn_head_rec_f32(ctrl_out, stride, B, H, W, Fh, Fw, flags, out, nullptr, stream);
}
extern "C" int ra_attn_head_bwd_f32(const float *ctrl_out, int stride, const float *out, const float *g_cn, const float *g_ls,
                                    const float *g_ctr, const float *g_size, const float *g_lg_var, const float *g_attn_gamma,
                                    const float *g_box_gamma, const float *g_y_lg_gamma, int B, int H, int W, int flags,
                                    float *d_ctrl_out, void *stream) {
  if (!ctrl_out || !out || !d_ctrl_out || B <= 0 || stride < 9) return fail(RA_E_INVALID, "ra_attn_head_bwd_f32: bad argument");
  hipLaunchKernelGGL(train::attn_head_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), ctrl_out, stride, out, g_cn,
                     g_ls, g_ctr, g_size, g_lg_var, g_attn_gamma, g_box_gamma, g_y_lg_gamma, B, (float)H, (float)W, flags,
                     d_ctrl_out);
  return launch_status("ra_attn_head_bwd_f32");
}
extern "C" int ra_knob_mix_rec_f32(const float *ctr, const float *size, const float *match, const float *ctr_gt, const float *size_gt,
                                   const float *knob, int knob_stride, int row_stride, int B, int T, float *ctr2, float *size2,
                                   const float *attn_rec, float *attn_rec2, void *stream) {
  if (!ctr || !size || !match || !ctr_gt || !size_gt || !knob || !ctr2 || !size2 || B <= 0 || T <= 0 || row_stride < 2 ||
      (attn_rec2 && !attn_rec))
    return fail(RA_E_INVALID, "ra_knob_mix_f32: bad argument");
  hipLaunchKernelGGL(train::knob_mix_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), ctr, size, match, ctr_gt, size_gt,
                     knob, knob_stride, row_stride, B, T, ctr2, size2, attn_rec, attn_rec2);
  return launch_status("ra_knob_mix_f32");
}
extern "C" int ra_knob_mix_f32(const float *ctr, const float *size, const float *match, const float *ctr_gt, const float *size_gt,
                               const float *knob, int knob_stride, int row_stride, int B, int T, float *ctr2, float *size2,
                               void *stream) {
  return ra_knob_mix_rec_f32(ctr, size, match, ctr_gt, size_gt, knob, knob_stride, row_stride, B, T, ctr2, size2, nullptr, nullptr, stream);
}
extern "C" int ra_knob_mix_bwd_f32(const float *g_ctr2, const float *g_size2, const float *knob, int knob_stride, int B,
                                   float *d_ctr, float *d_size, void *stream) {
  if (!knob || !d_ctr || !d_size || B <= 0) return fail(RA_E_INVALID, "ra_knob_mix_bwd_f32: bad argument");
  hipLaunchKernelGGL(train::knob_mix_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), g_ctr2, g_size2, knob,
                     knob_stride, B, d_ctr, d_size);
  return launch_status("ra_knob_mix_bwd_f32");
}
