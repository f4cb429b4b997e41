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
#include "ra_common.h"

namespace ra {
namespace train {
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_kernel(float *p, const float *g, float *m, float *v, const float *wd,
                                                   size_t n, float lr_t, float b1, float b2, float eps, float clip,
                                                   float gscale) {
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
                     wd_coef, n, lr_t, beta1, beta2, eps, clip, grad_scale);
  return launch_status("ra_adam_step_f32");
}
