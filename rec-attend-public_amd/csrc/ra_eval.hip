// K11..K12 — evaluation post-processing and metrics on the device:
//   K11 postprocess: utils/postprocess.py apply_confidence (:15-29) -> apply_one_label (:32-52) ->
//       apply_threshold (:5-12) [-> mask_foreground (:139-147)] fused into one pass over
//       y_out [B,T,H,W]; remove_tiny (:109-136) zeroes whole instance planes afterwards.  The cv2
//       steps (upsample + bilateral filter :75-106, morph :55-72) are not built.
//   K12 eval_metrics: the per-image statistics of analysis.py:314-760 from the pairwise
//       intersections and instance sizes of the binary masks (one ra_pair_stats_f32 pass):
//       pairwise IoU, symmetric best DICE, weighted / unweighted coverage, false positives /
//       negatives, object precision / recall flags, counting.
#include <cmath>

#include "ra_common.h"

namespace ra {
namespace eval {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxT = 32;

// One thread per 4 pixels: weight by the confidence, keep only the arg-max instance (first
// maximum, like numpy.argmax), threshold, optional foreground mask; also the union plane.
__global__ __launch_bounds__(256) void postprocess_kernel(const float *y, const float *s, int T, int HW,
                                                           float thresh, const float *fg, float *y_bin,
                                                           float *s_hard, float *uni) {
  const int b = blockIdx.y;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (blockIdx.x == 0 && threadIdx.x < T && s_hard)
    s_hard[(size_t)b * T + threadIdx.x] = s[(size_t)b * T + threadIdx.x] > 0.5f ? 1.f : 0.f;  // :28
  if (e >= HW) return;
  const float *yb = y + (size_t)b * T * HW + e;
  f32x4 best = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  int arg[4] = {0, 0, 0, 0};
  for (int t = 0; t < T; ++t) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(yb + (size_t)t * HW) * s[(size_t)b * T + t];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (v[k] > best[k]) {  // strict: the first maximum wins
        best[k] = v[k];
        arg[k] = t;
      }
  }
  f32x4 keep;
#pragma unroll
  for (int k = 0; k < 4; ++k) keep[k] = best[k] > thresh ? 1.f : 0.f;
  if (fg) keep = keep * *reinterpret_cast<const f32x4 *>(fg + (size_t)b * HW + e);
  float *ob = y_bin + (size_t)b * T * HW + e;
  for (int t = 0; t < T; ++t) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = arg[k] == t ? keep[k] : 0.f;
    *reinterpret_cast<f32x4 *>(ob + (size_t)t * HW) = o;
  }
  if (uni) *reinterpret_cast<f32x4 *>(uni + (size_t)b * HW + e) = keep;
}

// uni[b,px] = max_t y[b,t,px]  (analysis.py:547,570: y.max(axis=0))
__global__ __launch_bounds__(256) void union_kernel(const float *y, int T, int HW, float *uni) {
  const int b = blockIdx.y;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= HW) return;
  const float *yb = y + (size_t)b * T * HW + e;
  f32x4 m = *reinterpret_cast<const f32x4 *>(yb);
  for (int t = 1; t < T; ++t) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(yb + (size_t)t * HW);
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
  }
  *reinterpret_cast<f32x4 *>(uni + (size_t)b * HW + e) = m;
}

// morph_single (postprocess.py:63-72): cv2.dilate(plane, ones(5, 5)) = the maximum over the 5 x 5 window around every pixel,
// border pixels outside the image ignored (cv2's default border value for a dilation behaves as -inf).  One thread per
// pixel; the window's 25 loads hit L1 / L2 (the planes are small next to the caches).
__global__ __launch_bounds__(256) void dilate_kernel(const float *y, int H, int W, int R, float *out) {
  const int n = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * W) return;
  const int r = e / W, c = e - r * W;
  const float *p = y + (size_t)n * H * W;
  float m = -__builtin_inff();
  for (int dy = -R; dy <= R; ++dy) {
    const int yy = r + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -R; dx <= R; ++dx) {
      const int xx = c + dx;
      if (xx >= 0 && xx < W) m = fmaxf(m, p[yy * W + xx]);
    }
  }
  out[(size_t)n * H * W + e] = m;
}

// upsample_single (postprocess.py:93-106): cv2.resize(a, (W, H), INTER_LINEAR) then cv2.bilateralFilter(b, 5, 10, 10).
//   resize: pixel centres aligned — source coordinate (d + 0.5) * (src / dst) - 0.5, clamped to the image, two-tap linear
//   weights in float32 (cv2's float path; its 8-bit path uses fixed-point coefficients);
//   bilateral: d = 5 -> radius 2, the CIRCULAR neighbourhood dy^2 + dx^2 <= 4 (13 pixels), weight
//   exp(-(dy^2 + dx^2) / (2 * 10^2)) * exp(-(v - v0)^2 / (2 * 10^2)), borders reflected without the edge pixel
//   (BORDER_REFLECT_101).  cv2 evaluates the colour weight from an interpolated table; this is the formula the table
//   approximates (the two cannot be compared here: cv2 is not part of this stack).
__global__ __launch_bounds__(256) void resize_linear_kernel(const float *y, int Hs, int Ws, int H, int W, float *out) {
  const int n = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * W) return;
  const int r = e / W, c = e - r * W;
  const float sy = (float)Hs / (float)H, sx = (float)Ws / (float)W;
  float fy = ((float)r + 0.5f) * sy - 0.5f, fx = ((float)c + 0.5f) * sx - 0.5f;
  int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
  fy -= (float)y0;
  fx -= (float)x0;
  if (y0 < 0) y0 = 0, fy = 0.f;
  if (y0 >= Hs - 1) y0 = Hs - 1, fy = 0.f;
  if (x0 < 0) x0 = 0, fx = 0.f;
  if (x0 >= Ws - 1) x0 = Ws - 1, fx = 0.f;
  const int y1 = y0 + 1 < Hs ? y0 + 1 : y0, x1 = x0 + 1 < Ws ? x0 + 1 : x0;
  const float *p = y + (size_t)n * Hs * Ws;
  const float top = p[y0 * Ws + x0] * (1.f - fx) + p[y0 * Ws + x1] * fx;
  const float bot = p[y1 * Ws + x0] * (1.f - fx) + p[y1 * Ws + x1] * fx;
  out[(size_t)n * H * W + e] = top * (1.f - fy) + bot * fy;
}
__device__ inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}
__global__ __launch_bounds__(256) void bilateral5_kernel(const float *y, int H, int W, float sigma_color, float sigma_space, float *out) {
  const int n = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= H * W) return;
  const int r = e / W, c = e - r * W;
  const float *p = y + (size_t)n * H * W;
  const float v0 = p[e], gc = -0.5f / (sigma_color * sigma_color), gs = -0.5f / (sigma_space * sigma_space);
  float num = 0.f, den = 0.f;
  for (int dy = -2; dy <= 2; ++dy)
    for (int dx = -2; dx <= 2; ++dx) {
      if (dy * dy + dx * dx > 4) continue;
      const float v = p[reflect101(r + dy, H) * W + reflect101(c + dx, W)];
      const float w = expf((float)(dy * dy + dx * dx) * gs + (v - v0) * (v - v0) * gc);
      num += w * v;
      den += w;
    }
  out[(size_t)n * H * W + e] = num / den;
}

// remove_tiny_single (postprocess.py:126-136): planes of at most `threshold` pixels vanish
__global__ __launch_bounds__(256) void remove_tiny_kernel(float *y_bin, const float *sizes, float *conf, int HW,
                                                           float threshold) {
  const int inst = blockIdx.y;
  if (sizes[inst] > threshold) return;
  if (blockIdx.x == 0 && threadIdx.x == 0 && conf) conf[inst] = 0.f;
  float *p = y_bin + (size_t)inst * HW;
  for (int e = (blockIdx.x * 256 + threadIdx.x) * 4; e < HW; e += gridDim.x * 256 * 4)
    *reinterpret_cast<f32x4 *>(p + e) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// One workgroup per image; everything is T x T sized.  Per-image scalars -> stats[b][RA_EVAL_*],
// per-instance vectors -> inst[b][RA_EVALI_*][T].
__global__ __launch_bounds__(256) void eval_metrics_kernel(const float *inter, const float *sa, const float *sb,
                                                            const float *s_gt, const float *fg_inter,
                                                            const float *fg_a, const float *fg_b,
                                                            const float *a_in_fgb, const float *b_in_fga, int T,
                                                            float *iou_out, float *stats, float *inst) {
  __shared__ float iou[kMaxT][kMaxT], dice[kMaxT][kMaxT];
  __shared__ float red[8][kMaxT];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *I = inter + (size_t)b * T * T, *A = sa + (size_t)b * T, *Bs = sb + (size_t)b * T;
  const float *sg = s_gt + (size_t)b * T;
  for (int e = tid; e < T * T; e += 256) {
    const int i = e / T, j = e - i * T;
    const float in = I[e], uni = A[i] + Bs[j] - in, card = A[i] + Bs[j];
    const float v = in / (uni + (uni == 0.f ? 1.f : 0.f));  // analysis.py:314-326
    iou[i][j] = v;
    dice[i][j] = 2.f * in / (card + (card == 0.f ? 1.f : 0.f));  // :352-367
    if (iou_out) iou_out[(size_t)b * T * T + e] = v;
  }
  __syncthreads();
  if (tid < T) {
    const int k = tid;
    float bd_a = 0.f, bd_b = 0.f, cov = 0.f, rs = 0.f, cs = 0.f, mx_r = 0.f;
    for (int j = 0; j < T; ++j) {
      bd_a = fmaxf(bd_a, dice[k][j]);  // best DICE of output k over the GT (:370-386)
      bd_b = fmaxf(bd_b, dice[j][k]);  // best DICE of GT k over the outputs
      cov = fmaxf(cov, iou[j][k]);     // coverage of GT k (:463-464, max over axis 0)
      mx_r = fmaxf(mx_r, iou[k][j]);
      rs += iou[k][j];
      cs += iou[j][k];
    }
    red[0][k] = bd_a;
    red[1][k] = bd_b;
    red[2][k] = cov;
    red[3][k] = rs;
    red[4][k] = cs;
    red[5][k] = mx_r;
  }
  __syncthreads();
  if (tid == 0) {
    float count_gt = 0.f, tot_gt = 0.f, count_out = 0.f;
    for (int t = 0; t < T; ++t) {
      count_gt += sg[t];
      tot_gt += Bs[t];
      count_out += A[t] > 0.f ? 1.f : 0.f;  // f_count_out :766-770
    }
    const float num_obj = fmaxf(count_gt, 1.f);  // _f_num_obj :773-787
    const int no = (int)num_obj < T ? (int)num_obj : T;
    float bda = 0.f, bdb = 0.f, wt = 0.f, unwt = 0.f, fp = 0.f, fn = 0.f;
    for (int t = 0; t < no; ++t) {
      bda += red[0][t];
      bdb += red[1][t];
      wt += red[2][t] * (Bs[t] / (tot_gt + (tot_gt == 0.f ? 1.f : 0.f)));  // :467-478 weighted
      unwt += red[2][t] * (1.f / num_obj);
    }
    for (int t = 0; t < T; ++t) {
      fp += (A[t] > 0.f ? 1.f : 0.f) * (red[3][t] == 0.f ? 1.f : 0.f);  // :579-592
      fn += sg[t] * (red[4][t] == 0.f ? 1.f : 0.f);                      // :595-605
    }
    float *o = stats + (size_t)b * RA_EVAL_COUNT;
    o[RA_EVAL_SBD] = fminf(bda / (float)no, bdb / (float)no);  // :434-460
    o[RA_EVAL_WT_COV] = wt;
    o[RA_EVAL_UNWT_COV] = unwt;
    o[RA_EVAL_FP] = fp;
    o[RA_EVAL_FN] = fn;
    const float d = count_out - count_gt;
    o[RA_EVAL_COUNT_ACC] = d == 0.f ? 1.f : 0.f;  // :711-726
    o[RA_EVAL_COUNT_MSE] = d * d;                 // :693-708
    o[RA_EVAL_DIC] = d;                           // :729-744
    o[RA_EVAL_DIC_ABS] = fabsf(d);                // :747-763
    if (fg_inter) {  // foreground IoU / DICE of the unions (:533-576)
      const float in = fg_inter[b], fa = fg_a[b], fb = fg_b[b], uni = fa + fb - in, card = fa + fb;
      o[RA_EVAL_FG_IOU] = in / (uni + (uni == 0.f ? 1.f : 0.f));
      o[RA_EVAL_FG_DICE] = 2.f * in / (card + (card == 0.f ? 1.f : 0.f));
    } else {
      o[RA_EVAL_FG_IOU] = o[RA_EVAL_FG_DICE] = 0.f;
    }
    o[RA_EVAL_NUM_OBJ] = num_obj;
    o[RA_EVAL_COUNT_OUT] = count_out;
  }
  if (inst && tid < T) {
    float *o = inst + (size_t)b * RA_EVALI_COUNT * T;
    const int k = tid;
    float cgt = 0.f;
    for (int t = 0; t < T; ++t) cgt += sg[t];
    const float has_out = A[k] > 0.f ? 1.f : 0.f, is_gt = (float)k < cgt ? 1.f : 0.f;
    o[RA_EVALI_OBJ_PR * T + k] = red[5][k] >= 0.5f ? 1.f : 0.f;  // :653-671 (valid where has_out)
    o[RA_EVALI_OBJ_RE * T + k] = red[2][k] >= 0.5f ? 1.f : 0.f;  // :674-690 (valid where is_gt)
    o[RA_EVALI_HAS_OUT * T + k] = has_out;
    o[RA_EVALI_IS_GT * T + k] = is_gt;
    // pixel precision / recall against the union of the other side (:608-650, _f_pr :337-349)
    const float pa = a_in_fgb ? a_in_fgb[(size_t)b * T + k] : 0.f, pb = b_in_fga ? b_in_fga[(size_t)b * T + k] : 0.f;
    o[RA_EVALI_PIX_PR * T + k] = pa / (A[k] + (A[k] == 0.f ? 1.f : 0.f));
    o[RA_EVALI_PIX_RE * T + k] = pb / (Bs[k] + (Bs[k] == 0.f ? 1.f : 0.f));
  }
}

}  // namespace eval
}  // namespace ra

using namespace ra;

extern "C" int ra_postprocess_f32(const float *y_out, const float *s_out, int B, int T, int H, int W, float thresh,
                                  const float *fg, float *y_bin, float *s_hard, float *union_out, void *stream) {
  if (!y_out || !s_out || !y_bin || B <= 0 || T <= 0 || H <= 0 || W <= 0)
    return fail(RA_E_INVALID, "ra_postprocess_f32: bad argument");
  const int HW = H * W;
  if (HW % 4 || T > 256 || ((reinterpret_cast<uintptr_t>(y_out) | reinterpret_cast<uintptr_t>(y_bin)) & 15) ||
      (fg && (reinterpret_cast<uintptr_t>(fg) & 15)) || (union_out && (reinterpret_cast<uintptr_t>(union_out) & 15)))
    return fail(RA_E_SHAPE, "ra_postprocess_f32: H*W %% 4, T <= 256 and 16-byte aligned tensors required");
  hipLaunchKernelGGL(eval::postprocess_kernel, dim3(ceil_div(HW, 1024), B), dim3(256), 0, as_stream(stream), y_out,
                     s_out, T, HW, thresh, fg, y_bin, s_hard, union_out);
  return launch_status("ra_postprocess_f32");
}

extern "C" int ra_union_f32(const float *y, int B, int T, int HW, float *union_out, void *stream) {
  if (!y || !union_out || B <= 0 || T <= 0 || HW <= 0) return fail(RA_E_INVALID, "ra_union_f32: bad argument");
  if (HW % 4 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(union_out)) & 15))
    return fail(RA_E_SHAPE, "ra_union_f32: H*W %% 4 and 16-byte aligned tensors required");
  hipLaunchKernelGGL(eval::union_kernel, dim3(ceil_div(HW, 1024), B), dim3(256), 0, as_stream(stream), y, T, HW,
                     union_out);
  return launch_status("ra_union_f32");
}

extern "C" int ra_dilate_f32(const float *y, int N, int H, int W, int radius, float *out, void *stream) {
  if (!y || !out || y == out || N <= 0 || H <= 0 || W <= 0 || radius < 0 || radius > 16) return fail(RA_E_INVALID, "ra_dilate_f32: bad argument");
  hipLaunchKernelGGL(eval::dilate_kernel, dim3(ceil_div(H * W, 256), N), dim3(256), 0, as_stream(stream), y, H, W, radius, out);
  return launch_status("ra_dilate_f32");
}

extern "C" int ra_resize_linear_f32(const float *y, int N, int Hs, int Ws, int H, int W, float *out, void *stream) {
  if (!y || !out || N <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return fail(RA_E_INVALID, "ra_resize_linear_f32: bad argument");
  hipLaunchKernelGGL(eval::resize_linear_kernel, dim3(ceil_div(H * W, 256), N), dim3(256), 0, as_stream(stream), y, Hs, Ws, H, W, out);
  return launch_status("ra_resize_linear_f32");
}

extern "C" int ra_bilateral5_f32(const float *y, int N, int H, int W, float sigma_color, float sigma_space, float *out, void *stream) {
  if (!y || !out || y == out || N <= 0 || H <= 0 || W <= 0 || !(sigma_color > 0.f) || !(sigma_space > 0.f))
    return fail(RA_E_INVALID, "ra_bilateral5_f32: bad argument");
  hipLaunchKernelGGL(eval::bilateral5_kernel, dim3(ceil_div(H * W, 256), N), dim3(256), 0, as_stream(stream), y, H, W, sigma_color,
                     sigma_space, out);
  return launch_status("ra_bilateral5_f32");
}

extern "C" int ra_remove_tiny_f32(float *y_bin, const float *sizes, float *conf, int B, int T, int HW,
                                  float threshold, void *stream) {
  if (!y_bin || !sizes || B <= 0 || T <= 0 || HW <= 0) return fail(RA_E_INVALID, "ra_remove_tiny_f32: bad argument");
  if (HW % 4 || (reinterpret_cast<uintptr_t>(y_bin) & 15))
    return fail(RA_E_SHAPE, "ra_remove_tiny_f32: H*W %% 4 and a 16-byte aligned tensor required");
  hipLaunchKernelGGL(eval::remove_tiny_kernel, dim3(ceil_div(HW, 256 * 4 * 8), B * T), dim3(256), 0,
                     as_stream(stream), y_bin, sizes, conf, HW, threshold);
  return launch_status("ra_remove_tiny_f32");
}

extern "C" int ra_eval_metrics_f32(const float *inter, const float *sum_a, const float *sum_b, const float *s_gt,
                                   const float *fg_inter, const float *fg_a, const float *fg_b,
                                   const float *a_in_fgb, const float *b_in_fga, int B, int T, float *iou_pairwise,
                                   float *stats, float *inst, void *stream) {
  if (!inter || !sum_a || !sum_b || !s_gt || !stats || B <= 0 || T <= 0)
    return fail(RA_E_INVALID, "ra_eval_metrics_f32: bad argument");
  if (T > eval::kMaxT) return fail(RA_E_SHAPE, "ra_eval_metrics_f32: T=%d (max %d)", T, eval::kMaxT);
  if ((fg_inter != nullptr) != (fg_a != nullptr) || (fg_inter != nullptr) != (fg_b != nullptr))
    return fail(RA_E_INVALID, "ra_eval_metrics_f32: fg_inter, fg_a, fg_b go together");
  hipLaunchKernelGGL(eval::eval_metrics_kernel, dim3(B), dim3(256), 0, as_stream(stream), inter, sum_a, sum_b, s_gt,
                     fg_inter, fg_a, fg_b, a_in_fgb, b_in_fga, T, iou_pairwise, stats, inst);
  return launch_status("ra_eval_metrics_f32");
}

// ---- K13: in-graph augmentation, image_ops.random_transformation (image_ops.py:9-113) ----
// Zero-pad by `padding`, crop H x W at (off_y, off_x) (one offset per batch, :52), reverse along
// H / W (:85-91), transpose H <-> W (:93-97): one gather.  x is [N, H, W, C] (images, d, c) or,
// with C == 1, any stack of planes ([B*T, H, W] instance masks, :57-59).
namespace ra {
namespace eval {
__global__ __launch_bounds__(256) void random_transform_kernel(const float *x, int H, int W, int C, int padding,
                                                                int off_y, int off_x, int flip_v, int flip_h,
                                                                int transpose, float *out) {
  const int n = blockIdx.y;
  const size_t plane = (size_t)H * W * C;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < plane; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int j = (int)((e / C) % W), i = (int)(e / ((size_t)C * W));  // output pixel (i, j)
    // undo transpose, then the flips, then the crop offset
    int ri = transpose ? j : i, rj = transpose ? i : j;
    if (flip_v) ri = H - 1 - ri;
    if (flip_h) rj = W - 1 - rj;
    const int sy = ri + off_y - padding, sx = rj + off_x - padding;
    float v = 0.f;
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[(size_t)n * plane + ((size_t)sy * W + sx) * C + c];
    out[(size_t)n * plane + e] = v;
  }
}
}  // namespace eval
}  // namespace ra

extern "C" int ra_random_transform_f32(const float *x, int N, int H, int W, int C, int padding, int off_y, int off_x,
                                       int flip_v, int flip_h, int transpose, float *out, void *stream) {
  if (!x || !out || x == out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || padding < 0)
    return fail(RA_E_INVALID, "ra_random_transform_f32: bad argument");
  if (off_y < 0 || off_x < 0 || off_y > 2 * padding || off_x > 2 * padding)
    return fail(RA_E_INVALID, "ra_random_transform_f32: offset (%d, %d) outside [0, 2*padding]", off_y, off_x);
  if (transpose && H != W) return fail(RA_E_SHAPE, "ra_random_transform_f32: transpose needs H == W");
  const size_t plane = (size_t)H * W * C;
  const int gx = (int)((plane + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(eval::random_transform_kernel, dim3(gx < 1 ? 1 : gx, N), dim3(256), 0, as_stream(stream), x, H, W,
                     C, padding, off_y, off_x, flip_v, flip_h, transpose, out);
  return launch_status("ra_random_transform_f32");
}

// ---- colour jitter of the in-graph augmentation (image_ops.py:99-103: random_hue 0.1, random_saturation 0.9..1.1,
// tf.image.random_brightness 0.1, tf.image.random_contrast 0.9..1.1; one draw of each per batch) ----
// TensorFlow's kernels, restated from their published algorithm (tensorflow/core/kernels/colorspace_op.h,
// adjust_contrast_op.cc; python/ops/image_ops.py adjust_hue / adjust_saturation / adjust_brightness): RGB -> HSV, hue =
// (hue + delta + 1) mod 1, saturation = clip(saturation * factor, 0, 1), HSV -> RGB, + brightness delta (float images are
// not clipped), then (x - mean) * contrast + mean with the mean of each image and channel over H x W.
namespace ra {
namespace eval {
__device__ inline void hue_sat_pixel(float &r, float &g, float &b, float dh, float sf) {
  const float v = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), range = v - mn;
  float s = v > 0.f ? range / v : 0.f;
  const float norm = 1.0f / (6.0f * range);
  float h = (r == v) ? norm * (g - b) : (g == v) ? norm * (b - r) + 2.0f / 6.0f : norm * (r - g) + 4.0f / 6.0f;
  h = range > 0.f ? h : 0.f;
  h = h < 0.f ? h + 1.0f : h;
  h = fmodf(h + (dh + 1.0f), 1.0f);
  s = fminf(fmaxf(s * sf, 0.f), 1.f);
  const float d6 = h * 6.0f, one_s = 1.0f - s;
  const float dr = fminf(fmaxf(fabsf(d6 - 3.0f) - 1.0f, 0.f), 1.f);
  const float dg = fminf(fmaxf(2.0f - fabsf(d6 - 2.0f), 0.f), 1.f);
  const float db = fminf(fmaxf(2.0f - fabsf(d6 - 4.0f), 0.f), 1.f);
  r = (one_s + s * dr) * v;
  g = (one_s + s * dg) * v;
  b = (one_s + s * db) * v;
}

// pass 1: hue / saturation / brightness per pixel, and this block's per-channel sums -> part[image][block][3]
__global__ __launch_bounds__(256) void colour_pass1_kernel(const float *x, int HW, float dh, float sf, float db, float *out, float *part) {
  const int img = blockIdx.y;
  const float *xi = x + (size_t)img * HW * 3;
  float *oi = out + (size_t)img * HW * 3;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float r = xi[3 * p], g = xi[3 * p + 1], b = xi[3 * p + 2];
    hue_sat_pixel(r, g, b, dh, sf);
    r += db;
    g += db;
    b += db;
    oi[3 * p] = r;
    oi[3 * p + 1] = g;
    oi[3 * p + 2] = b;
    acc[0] += r;
    acc[1] += g;
    acc[2] += b;
  }
  __shared__ float sm[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = acc[c];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) part[((size_t)img * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// pass 2: contrast about the image's channel means (the blocks' partial sums added in block order: deterministic)
__global__ __launch_bounds__(256) void colour_pass2_kernel(float *x, int HW, int nblk, const float *part, float cf) {
  const int img = blockIdx.y;
  __shared__ float mean[3];
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += part[((size_t)img * nblk + k) * 3 + threadIdx.x];
    mean[threadIdx.x] = s / (float)HW;
  }
  __syncthreads();
  float *xi = x + (size_t)img * HW * 3;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 3 * HW; e += gridDim.x * 256) {
    const float m = mean[e % 3];
    xi[e] = (xi[e] - m) * cf + m;
  }
}
}  // namespace eval
}  // namespace ra

extern "C" size_t ra_colour_jitter_workspace_floats(int B) { return B > 0 ? (size_t)B * 64 * 3 : 0; }

extern "C" int ra_colour_jitter_f32(const float *x, int B, int HW, float hue_delta, float saturation_factor,
                                    float brightness_delta, float contrast_factor, float *ws, size_t ws_floats, float *out,
                                    void *stream) {
  if (!x || !out || !ws || B <= 0 || HW <= 0) return fail(RA_E_INVALID, "ra_colour_jitter_f32: bad argument");
  if (ws_floats < ra_colour_jitter_workspace_floats(B)) return fail(RA_E_WORKSPACE, "ra_colour_jitter_f32: workspace");
  const int nblk = 64;
  hipLaunchKernelGGL(eval::colour_pass1_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), x, HW, hue_delta, saturation_factor,
                     brightness_delta, out, ws);
  hipLaunchKernelGGL(eval::colour_pass2_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), out, HW, nblk, ws, contrast_factor);
  return launch_status("ra_colour_jitter_f32");
}

// out[b,p] = sum_t w[b,t] * y[b,t,p] — the ground-truth instance picked by box_model's greedy match
// (box_model.py:487-499: reduce_sum(grd_match * y_gt, 1)).
namespace ra {
namespace eval {
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float *w, const float *y, int T, int HW,
                                                            float *out) {
  const int b = blockIdx.y;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= HW) return;
  const float *yb = y + (size_t)b * T * HW + e;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const float wt = w[(size_t)b * T + t];
    if (wt != 0.f) acc += wt * *reinterpret_cast<const f32x4 *>(yb + (size_t)t * HW);  // uniform per image
  }
  *reinterpret_cast<f32x4 *>(out + (size_t)b * HW + e) = acc;
}
}  // namespace eval
}  // namespace ra

extern "C" int ra_weighted_sum_f32(const float *w, const float *y, int B, int T, int HW, float *out, void *stream) {
  if (!w || !y || !out || B <= 0 || T <= 0 || HW <= 0) return fail(RA_E_INVALID, "ra_weighted_sum_f32: bad argument");
  if (HW % 4 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15))
    return fail(RA_E_SHAPE, "ra_weighted_sum_f32: H*W %% 4 and 16-byte aligned tensors required");
  hipLaunchKernelGGL(eval::weighted_sum_kernel, dim3(ceil_div(HW, 1024), B), dim3(256), 0, as_stream(stream), w, y, T,
                     HW, out);
  return launch_status("ra_weighted_sum_f32");
}
