// K8..K10 — the loss / statistics head of the training graph, forward only
// (full_model.py:913-1097; modellib.py:28-37,71-155,265-339,382-415,482-511,663-701).
//
//   K8  pair_stats:  one streaming pass over a [B,N,H,W] and b [B,M,H,W] produces the pairwise
//       soft IoU, the pairwise IoU and DICE of (a > 0.5), and the per-instance sums — the
//       reference makes three passes (f_iou soft :981, f_iou hard :1065, f_dice :1073).  The
//       [N,HW] x [HW,M] contraction runs on v_mfma_f32_16x16x4_f32 with the pixel axis as K;
//       at 4 FLOP/B it is HBM-bound (32 MiB per image at cfg2), the MFMA only keeps the VALU
//       out of the way.  Partials per pixel chunk are reduced in a fixed order by a finishing
//       kernel, so results do not depend on scheduling.
//   K9  gt_box:      get_gt_box — bounding box of every GT instance by min/max reductions, then
//       the filled, padded rectangle mask.
//   K10 segm_match pre/post-processing around the device Hungarian solver, and the scalar
//       statistics (coverage, matched IoU, DICE, confidence loss, counting) in one workgroup.
#include <cmath>
#include <thread>
#include <vector>

#include "ra_common.h"

namespace ra {
namespace loss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxT = 32;        // instances per image supported (two 16-row MFMA blocks)
constexpr int kChunkPx = 4096;   // pixels per workgroup of the streaming pass
constexpr int kPartFloats = 2 * kMaxT * kMaxT + 3 * kMaxT;  // per-(image, chunk) partial record

// partial record layout: [soft inter NxM (32x32)][hard inter][sum a (32)][sum a>0.5][sum b]
template <int NI, int NJ>
__global__ __launch_bounds__(256) void pair_stats_kernel(const float *a, const float *b, int N, int M,
                                                          int HW, int nchunks, float *part, size_t a_img, size_t a_row) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int chunk = blockIdx.x, img = blockIdx.y;
  const float *ab = a + (size_t)img * a_img, *bb = b + (size_t)img * M * HW;  // a[img][row] at img * a_img + row * a_row
  f32x4 accs[NI][NJ], acch[NI][NJ];
  float sa[NI], sah[NI], sb[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    sa[i] = sah[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) accs[i][j] = acch[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) sb[j] = 0.f;
  // a wave streams 64 pixels per step: lane (row r, quad q) loads float4s at 16u + 4q, so a row is
  // read in 256-byte runs; MFMA step e pairs element e of the a- and b-float4 of the same lane
  // (A[m=r][k=q] = a[r][px], B[k=q][n=r] = b[r][px]: any pixel order works if both use it)
  const int px_begin = chunk * kChunkPx + wave * (kChunkPx / 4), px_end = px_begin + kChunkPx / 4;
  f32x4 av[NI][4], bv[NJ][4], an[NI][4], bn[NJ][4];
  auto fetch = [&](int p0, f32x4 (&xa)[NI][4], f32x4 (&xb)[NJ][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int px = p0 + 16 * u + 4 * q;
      const bool ok = (px < HW) & (p0 < px_end);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        xa[i][u] = (ok && 16 * i + r < N) ? *reinterpret_cast<const f32x4 *>(ab + (size_t)(16 * i + r) * a_row + px)
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        xb[j][u] = (ok && 16 * j + r < M) ? *reinterpret_cast<const f32x4 *>(bb + (size_t)(16 * j + r) * HW + px)
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch(px_begin, an, bn);
  for (int p0 = px_begin; p0 < px_end; p0 += 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i][u] = an[i][u];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j][u] = bn[j][u];
    }
    fetch(p0 + 64, an, bn);  // the next step's loads fly while this step is computed
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 ah[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ah[i][e] = av[i][u][e] > 0.5f ? 1.f : 0.f;  // full_model.py:1064
        sa[i] += (av[i][u][0] + av[i][u][1]) + (av[i][u][2] + av[i][u][3]);
        sah[i] += (ah[i][0] + ah[i][1]) + (ah[i][2] + ah[i][3]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) sb[j] += (bv[j][u][0] + bv[j][u][1]) + (bv[j][u][2] + bv[j][u][3]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            accs[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][u][e], bv[j][u][e], accs[i][j], 0, 0, 0);
            acch[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i][e], bv[j][u][e], acch[i][j], 0, 0, 0);
          }
    }
  }
  // reduce the 4 waves through LDS in a fixed order, then one record per (image, chunk)
  __shared__ float red[4][kPartFloats];
  float *mine = red[wave];
  for (int e = lane; e < kPartFloats; e += 64) mine[e] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // D[m = 4*(lane>>4) + k][n = lane & 15]
        const int m = 16 * i + 4 * q + k, n = 16 * j + r;
        mine[m * kMaxT + n] = accs[i][j][k];
        mine[kMaxT * kMaxT + m * kMaxT + n] = acch[i][j][k];
      }
  // row sums: lanes (r, q = 0..3) hold partials of the same row
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float s = sa[i], h = sah[i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    h += __shfl_xor(h, 16);
    h += __shfl_xor(h, 32);
    if (q == 0) {
      mine[2 * kMaxT * kMaxT + 16 * i + r] = s;
      mine[2 * kMaxT * kMaxT + kMaxT + 16 * i + r] = h;
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float s = sb[j];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (q == 0) mine[2 * kMaxT * kMaxT + 2 * kMaxT + 16 * j + r] = s;
  }
  __syncthreads();
  float *dst = part + ((size_t)img * nchunks + chunk) * kPartFloats;
  for (int e = tid; e < kPartFloats; e += 256) dst[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// sum the chunk records of one image in a fixed order: block (x = 256-element slice, y = image)
__global__ __launch_bounds__(256) void pair_stats_reduce_kernel(const float *part, int nchunks, float *tot) {
  const int img = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kPartFloats) return;
  const float *src = part + (size_t)img * nchunks * kPartFloats + e;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 8 <= nchunks; c += 8)  // 8 independent loads in flight
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += src[(size_t)(c + k) * kPartFloats];
  for (; c < nchunks; ++c) s[0] += src[(size_t)c * kPartFloats];
  tot[(size_t)img * kPartFloats + e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// one workgroup per image: the ratios
__global__ __launch_bounds__(256) void pair_stats_finish_kernel(const float *totals, int N, int M, int HW,
                                                                 float *iou_soft, float *iou_hard,
                                                                 float *dice_hard, float *sum_a, float *sum_b,
                                                                 float *inter_soft, float *sum_a_hard) {
  const int img = blockIdx.x, tid = threadIdx.x;
  const float *tot = totals + (size_t)img * kPartFloats;
  const float eps_hw = 1e-5f * (float)HW;  // modellib.py:119-122: the eps is summed per pixel
  const float *sa = tot + 2 * kMaxT * kMaxT, *sah = sa + kMaxT, *sb = sah + kMaxT;
  for (int e = tid; e < N * M; e += 256) {
    const int i = e / M, j = e - i * M;
    const float is = tot[i * kMaxT + j], ih = tot[kMaxT * kMaxT + i * kMaxT + j];
    const size_t o = ((size_t)img * N + i) * M + j;
    if (iou_soft) iou_soft[o] = is / (sa[i] + sb[j] - is + eps_hw);
    if (iou_hard) iou_hard[o] = ih / (sah[i] + sb[j] - ih + eps_hw);
    if (dice_hard) dice_hard[o] = 2.f * ih / ((sah[i] + eps_hw) + (sb[j] + eps_hw));  // modellib.py:92-98
    if (inter_soft) inter_soft[o] = is;
  }
  if (sum_a_hard)
    for (int i = tid; i < N; i += 256) sum_a_hard[(size_t)img * N + i] = sah[i];
  if (sum_a)
    for (int i = tid; i < N; i += 256) sum_a[(size_t)img * N + i] = sa[i];
  if (sum_b)
    for (int j = tid; j < M; j += 256) sum_b[(size_t)img * M + j] = sb[j];
}

// ---- K9: get_gt_box (modellib.py:663-701) ----
// params record per (b, t): [0..1] top_left (y, x), [2..3] bot_right after the empty-instance
// fix-up, [4..7] the padded rectangle the mask is filled with (tl_y, tl_x, br_y, br_x)
constexpr int kGtSplit = 8;  // workgroups per instance in the min/max/sum reduction

// partial record per (instance, slice): min_y, min_x, max_y, max_x, sum
__global__ __launch_bounds__(256) void gt_box_reduce_kernel(const float *y_gt, int H, int W, float *partial) {
  __shared__ float red[5][256];
  const int inst = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
  const float *y = y_gt + (size_t)inst * H * W;
  const float big = (float)(H * W);
  const int per = ((H * W / 4 + kGtSplit - 1) / kGtSplit) * 4;  // pixels per slice, multiple of 4
  const int e0 = slice * per, e1 = (e0 + per < H * W) ? e0 + per : H * W;
  float mny = 3.0e38f, mnx = 3.0e38f, mxy = -3.0e38f, mxx = -3.0e38f, sum = 0.f;
  for (int e = e0 + tid * 4; e < e1; e += 256 * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(y + e);  // H*W % 4 == 0 checked by the host
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = (e + k) / W, xx = (e + k) - yy * W;
      const float g = v[k];
      mny = fminf(mny, (float)yy + (1.f - g) * big);
      mnx = fminf(mnx, (float)xx + (1.f - g) * big);
      mxy = fmaxf(mxy, (float)yy * g);
      mxx = fmaxf(mxx, (float)xx * g);
      sum += g;
    }
  }
  red[0][tid] = mny;
  red[1][tid] = mnx;
  red[2][tid] = mxy;
  red[3][tid] = mxx;
  red[4][tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] = fminf(red[0][tid], red[0][tid + s]);
      red[1][tid] = fminf(red[1][tid], red[1][tid + s]);
      red[2][tid] = fmaxf(red[2][tid], red[2][tid + s]);
      red[3][tid] = fmaxf(red[3][tid], red[3][tid + s]);
      red[4][tid] += red[4][tid + s];
    }
    __syncthreads();
  }
  if (tid < 5) partial[((size_t)inst * kGtSplit + slice) * 5 + tid] = red[tid][0];
}

// one thread per instance: combine the slices in order, pad, write the params record
__global__ void gt_box_finish_kernel(const float *partial, int ninst, float pad_ratio, float min_pad,
                                     float *params) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= ninst) return;
  const float *q = partial + (size_t)inst * kGtSplit * 5;
  float tl[2] = {q[0], q[1]}, br[2] = {q[2], q[3]}, sum = q[4];
  for (int s = 1; s < kGtSplit; ++s) {
    tl[0] = fminf(tl[0], q[s * 5 + 0]);
    tl[1] = fminf(tl[1], q[s * 5 + 1]);
    br[0] = fmaxf(br[0], q[s * 5 + 2]);
    br[1] = fmaxf(br[1], q[s * 5 + 3]);
    sum += q[s * 5 + 4];
  }
  const float nz = sum > 0.f ? 1.f : 0.f;
  float *p = params + (size_t)inst * 8;
  for (int k = 0; k < 2; ++k) {
    const float size = br[k] - tl[k];
    const float pad = fmaxf(pad_ratio * size, min_pad);
    tl[k] -= pad;
    br[k] += pad;
    p[4 + k] = tl[k];
    p[6 + k] = br[k];
    p[k] = tl[k] * nz;
    p[2 + k] = nz * br[k] + (1.f - nz) * (2.f * min_pad);
  }
}

// ---- the step's noisy ground-truth attention and knob masks (modellib.get_gt_attn with per-instance padding / centre
// shift, full_model.py:567-577, and the Bernoulli knobs of :596-625) from the min / max / sum partials ra_gt_box_f32 left
// in its workspace: one thread per instance, the float32 operations in the order the element-wise form takes them
// (no contraction).  Replaces a second reduction over y_gt, a pairwise-statistics pass (for "instance not empty") and
// ~28 element-wise launches on B T numbers.
__global__ void knob_setup_kernel(const float *partial, int ninst, int T, const float *pad, const float *shift, const float *u_box,
                                  const float *u_segm, const float *sched, float min_pad, int timescale, float *ctr, float *size,
                                  float *kbox, float *ksegm) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= ninst) return;
  const float *q = partial + (size_t)inst * kGtSplit * 5;
  float tl[2] = {q[0], q[1]}, br[2] = {q[2], q[3]}, sum = q[4];
  for (int s = 1; s < kGtSplit; ++s) {
    tl[0] = fminf(tl[0], q[s * 5 + 0]);
    tl[1] = fminf(tl[1], q[s * 5 + 1]);
    br[0] = fmaxf(br[0], q[s * 5 + 2]);
    br[1] = fmaxf(br[1], q[s * 5 + 3]);
    sum += q[s * 5 + 4];
  }
  const float nz = sum > 0.f ? 1.f : 0.f;
  const float far = __fmul_rn(1.f - nz, 2.f * min_pad);
  for (int k = 0; k < 2; ++k) {
    const float sz = br[k] - tl[k];
    const float padv = fmaxf(__fmul_rn(pad[inst], sz), min_pad);
    const float sh = __fmul_rn(shift[inst * 2 + k], sz);
    const float tln = __fmul_rn(__fsub_rn(__fadd_rn(tl[k], sh), padv), nz);
    const float brn = __fadd_rn(__fmul_rn(nz, __fadd_rn(__fadd_rn(br[k], sh), padv)), far);
    ctr[inst * 2 + k] = __fadd_rn(tln, brn) / 2.f;
    size[inst * 2 + k] = __fsub_rn(brn, tln);
  }
  const float scale = timescale ? 1.f + logf(1.f + (float)(inst % T) * 3.f) : 1.f;
  kbox[inst] = u_box[inst] <= fminf(__fmul_rn(sched[0], scale), 1.f) ? 1.f : 0.f;
  ksegm[inst] = u_segm[inst] <= fminf(__fmul_rn(sched[1], scale), 1.f) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void gt_box_fill_kernel(const float *params, int H, int W, float *box) {
  const int inst = blockIdx.y;
  const float *p = params + (size_t)inst * 8;
  const float ty = p[4], tx = p[5], by = p[6], bx = p[7];
  float *o = box + (size_t)inst * H * W;
  for (int e = (blockIdx.x * 256 + threadIdx.x) * 4; e < H * W; e += gridDim.x * 256 * 4) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = (e + k) / W, xx = (e + k) - yy * W;
      v[k] = ((float)yy >= ty && (float)xx >= tx && (float)yy <= by && (float)xx <= bx) ? 1.f : 0.f;
    }
    *reinterpret_cast<f32x4 *>(o + e) = v;
  }
}

// ---- soft IoU of ONE map per image against the T filled rectangles of get_gt_box (the training graph's per-timestep
// f_iou(attn_box, attn_box_gt, pairwise) row, full_model.py:744-758): inter_t = sum of the map inside rectangle t,
// sum_b = the rectangle's pixel count — one read of the map instead of the map + T rectangle planes.  64 workgroups
// per image leave partial sums, a second launch adds them in a fixed order. ----
constexpr int kRectBlocks = 64;  // workgroups per image in the first stage
template <int TM>  // the unrolled rectangle count: the smallest of 8 / 16 / 32 that holds T (the loop body is predicated, not skipped)
__global__ __launch_bounds__(256) void box_iou_rects_kernel(const float *box, const float *params, int T, int H, int W,
                                                            float *part, int nblk) {
  __shared__ float red[4][kMaxT + 1];
  __shared__ float rect[kMaxT][4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < T * 4) rect[tid >> 2][tid & 3] = params[((size_t)b * T + (tid >> 2)) * 8 + 4 + (tid & 3)];
  __syncthreads();
  const float *a = box + (size_t)b * H * W;
  float acc[TM + 1];
#pragma unroll
  for (int t = 0; t <= TM; ++t) acc[t] = 0.f;
  for (int e = (blockIdx.x * 256 + tid) * 4; e < H * W; e += nblk * 1024) {  // W % 4 == 0: four pixels of one row
    const f32x4 v = *reinterpret_cast<const f32x4 *>(a + e);
    const int yy = e / W, xx = e - yy * W;
    acc[TM] += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if (t < T) {
        const float ty = rect[t][0], tx = rect[t][1], by = rect[t][2], bx = rect[t][3];
        if ((float)yy >= ty && (float)yy <= by) {
          float sx = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) sx += ((float)(xx + k) >= tx && (float)(xx + k) <= bx) ? v[k] : 0.f;
          acc[t] += sx;
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t <= TM; ++t) {
    float s = acc[t];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wave][t == TM ? kMaxT : t] = s;
  }
  __syncthreads();
  if (tid <= kMaxT && (tid < TM || tid == kMaxT))
    part[((size_t)b * kRectBlocks + blockIdx.x) * (kMaxT + 1) + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(64) void box_iou_rects_finish_kernel(const float *part, const float *params, int T, int H, int W,
                                                                  float *iou, int nblk) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= T) return;
  float is = 0.f, sa = 0.f;
  for (int k = 0; k < nblk; ++k) {  // fixed order
    is += part[((size_t)b * kRectBlocks + k) * (kMaxT + 1) + t];
    sa += part[((size_t)b * kRectBlocks + k) * (kMaxT + 1) + kMaxT];
  }
  // pixels the fill marks: integers y in [ty, by] and x in [tx, bx] inside the image (gt_box_fill_kernel)
  const float *p = params + ((size_t)b * T + t) * 8 + 4;
  const int y0 = (int)fmaxf(ceilf(p[0]), 0.f), y1 = (int)fminf(floorf(p[2]), (float)(H - 1));
  const int x0 = (int)fmaxf(ceilf(p[1]), 0.f), x1 = (int)fminf(floorf(p[3]), (float)(W - 1));
  const float sb = (y1 >= y0 && x1 >= x0) ? (float)(y1 - y0 + 1) * (float)(x1 - x0 + 1) : 0.f;
  iou[(size_t)b * T + t] = is / (sa + sb - is + 1e-5f * (float)(H * W));
}

// ---- K10: f_segm_match pre/post (modellib.py:395-413) ----
__global__ void match_pre_kernel(const float *iou, const float *s_gt, int N, int total, float *w) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int j = e % N, i = (e / N) % N, b = e / (N * N);
  const float m = iou[e] * s_gt[b * N + j] * s_gt[b * N + i];
  // tf.round(x * 1e6) / 1e6 taken as floor(x + 0.5), then + eps
  w[e] = floorf(m * 1e6f + 0.5f) / 1e6f + 1e-5f;
}

__global__ void match_post_kernel(float *match, const float *s_gt, int N, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int j = e % N, i = (e / N) % N, b = e / (N * N);
  match[e] = match[e] * s_gt[b * N + j] * s_gt[b * N + i];
}

// Scalar statistics, stage 1: one workgroup per image, thread t owns output row t and ground-truth
// column t of the T x T matrices; the 12 per-image terms go to partial[b][12].
constexpr int kNTerms = 12;
__global__ __launch_bounds__(64) void loss_stats_image_kernel(
    const float *iou_soft, const float *iou_hard, const float *dice, const float *match_real,
    const float *iou_box, const float *match_box, const float *s_out, const float *s_gt,
    const float *sum_gt, int T, int fixed_order, float *partial) {
  __shared__ float red[16][kMaxT];
  __shared__ float cmin_s[kMaxT], cmax_s[kMaxT];
  const int b = blockIdx.x, t = threadIdx.x;
  const float *sg = s_gt + (size_t)b * T, *so = s_out + (size_t)b * T;
  const float *is = iou_soft + (size_t)b * T * T, *ih = iou_hard + (size_t)b * T * T;
  const float *dc = dice + (size_t)b * T * T, *mr = match_real + (size_t)b * T * T;
  const float *ib = iou_box + (size_t)b * T * T, *mb = match_box + (size_t)b * T * T;
  if (t == 0) {  // cumulative min from the start / max to the end (modellib.py:40-68)
    float run = 3.0e38f;
    for (int i = 0; i < T; ++i) {
      run = fminf(run, so[i]);
      cmin_s[i] = run;
    }
    run = -3.0e38f;
    for (int i = T - 1; i >= 0; --i) {
      run = fmaxf(run, so[i]);
      cmax_s[i] = run;
    }
  }
  __syncthreads();
  float v[16];
  for (int k = 0; k < 16; ++k) v[k] = 0.f;
  if (t < T) {
    const int i = t;
    float cnt = 0.f, cnt_box = 0.f, m_is = 0.f, m_ib = 0.f, m_ih = 0.f, m_dc = 0.f, msum = 0.f;
    float mxs = -3.0e38f, mxh = -3.0e38f;
    for (int j = 0; j < T; ++j) {
      // identity match (modellib.py:28-37) when fixed_order, else the Hungarian matches
      const float ident = (i == j) ? sg[i] * sg[j] : 0.f;
      const float m = fixed_order ? ident : mr[i * T + j];
      const float mbx = fixed_order ? ident : mb[i * T + j];
      cnt += m;
      cnt_box += mbx;
      msum += m;
      m_is += is[i * T + j] * m;
      m_ib += ib[i * T + j] * mbx;
      m_ih += ih[i * T + j] * mr[i * T + j];  // hard statistics always use the real match (:1062)
      m_dc += dc[i * T + j] * mr[i * T + j];
      mxs = fmaxf(mxs, is[j * T + i]);  // coverage of ground-truth instance i: max over outputs j
      mxh = fmaxf(mxh, ih[j * T + i]);
    }
    v[0] = cnt;
    v[1] = cnt_box;
    v[2] = m_is;
    v[3] = m_ib;
    v[4] = m_ih;
    v[5] = m_dc;
    v[6] = mxs;
    v[7] = mxh;
    v[8] = sum_gt[(size_t)b * T + i];
    // confidence loss with cumulative min / max (modellib.py:316-339,430-437)
    v[9] = -msum * logf(cmin_s[i] + 1e-5f) - (1.f - msum) * logf(1.f - cmax_s[i] + 1e-5f);
    v[10] = so[i] > 0.5f ? 1.f : 0.f;
    v[11] = sg[i];
  }
  if (t < kMaxT)
    for (int k = 0; k < 12; ++k) red[k][t] = v[k];
  __syncthreads();
  if (t == 0) {  // combine the T rows in order (T <= 32)
    float cnt = 0, cnt_box = 0, m_is = 0, m_ib = 0, m_ih = 0, m_dc = 0, tot_gt = 0, conf = 0, cout = 0, cgt = 0;
    for (int i = 0; i < T; ++i) {
      cnt += red[0][i];
      cnt_box += red[1][i];
      m_is += red[2][i];
      m_ib += red[3][i];
      m_ih += red[4][i];
      m_dc += red[5][i];
      tot_gt += red[8][i];
      conf += red[9][i];
      cout += red[10][i];
      cgt += red[11][i];
    }
    cnt = fmaxf(1.f, cnt);
    cnt_box = fmaxf(1.f, cnt_box);
    float cs = 0, ch = 0, ws = 0, wh = 0;  // coverage (modellib.py:265-313)
    for (int j = 0; j < T; ++j) {
      const float g = red[8][j];
      const float wt = g / (tot_gt + (g == 0.f ? 1.f : 0.f));
      cs += red[6][j];
      ch += red[7][j];
      ws += red[6][j] * wt;
      wh += red[7][j] * wt;
    }
    float *o = partial + (size_t)b * kNTerms;
    o[0] = m_is / cnt;      // matched soft IoU
    o[1] = m_ib / cnt_box;  // matched box IoU
    o[2] = ws;              // weighted coverage, soft
    o[3] = cs / cnt;        // unweighted coverage, soft
    o[4] = wh;
    o[5] = ch / cnt;
    o[6] = m_ih / cnt;
    o[7] = m_dc / cnt;
    o[8] = conf;
    o[9] = (cout == cgt) ? 1.f : 0.f;
    o[10] = cout - cgt;
    o[11] = fabsf(cout - cgt);
  }
}

// stage 2: mean over the batch (fixed order) and the losses.  out: RA_STAT_* in recattend.h.
__global__ __launch_bounds__(64) void loss_stats_final_kernel(const float *partial, int B, int T, int segm_fn,
                                                               float mix, float *out) {
  __shared__ float tot[kNTerms];
  const int t = threadIdx.x;
  if (t < kNTerms) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += partial[(size_t)b * kNTerms + t];
    tot[t] = s / (float)B;
  }
  __syncthreads();
  if (t == 0) {
    const float iou_soft_s = tot[0], iou_box_s = tot[1], wt_soft = tot[2], conf = tot[8] / (float)T;
    const float box_loss = -iou_box_s;  // box_loss_fn 'iou' (full_model.py:965-966)
    const float segm_loss = segm_fn == 1 ? -wt_soft : -iou_soft_s;  // :1013-1016
    out[RA_STAT_LOSS] = box_loss + segm_loss + mix * conf;
    out[RA_STAT_BOX_LOSS] = box_loss;
    out[RA_STAT_SEGM_LOSS] = segm_loss;
    out[RA_STAT_CONF_LOSS] = conf;
    out[RA_STAT_IOU_SOFT] = iou_soft_s;
    out[RA_STAT_IOU_SOFT_BOX] = iou_box_s;
    out[RA_STAT_WT_COV_SOFT] = wt_soft;
    out[RA_STAT_UNWT_COV_SOFT] = tot[3];
    out[RA_STAT_IOU_HARD] = tot[6];
    out[RA_STAT_WT_COV_HARD] = tot[4];
    out[RA_STAT_UNWT_COV_HARD] = tot[5];
    out[RA_STAT_DICE] = tot[7];
    out[RA_STAT_COUNT_ACC] = tot[9];
    out[RA_STAT_DIC] = tot[10];
    out[RA_STAT_DIC_ABS] = tot[11];
  }
}

inline int nchunks_for(int HW) { return ceil_div(HW, kChunkPx); }


// ---- the scalar head of the training loss (full_model.py:913-1035, box_loss_fn = segm_loss_fn = 'iou') in one launch each
// way: matched soft IoU of masks and boxes, the confidence loss on the cumulative min / max of the scores, their mix — and,
// backward, the coefficient tensors of the two pairwise-IoU adjoints (PairIoU.backward's c1 / c0) and d s_out.  One wave
// per image, fixed summation order.  iou / inter [B,T,T], sum_a / sum_b [B,T], match [B,T,T] (pred, gt), s_out [B,T].
struct LossHeadArgs {
  const float *iou_s, *iou_b, *m_s, *m_b, *s_out;
  const float *inter_s, *sa_s, *sb_s, *inter_b, *sa_b, *sb_b;  // backward only
  const float *g;                                              // backward: upstream gradient of `loss` (device scalar) or null = 1
  int B, T, HW;
  float mix;
  float *pieces;    // forward: [6 + 3 B]: loss, box_loss, segm_loss, conf, iou_soft, iou_box, then per image (iou_s, iou_b, conf)
  float *c1_s, *c0_s, *c1_b, *c0_b, *ds;  // backward
};
__device__ inline float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// the image's count-normalised matched IoU: sum(iou * m) / max(sum m, 1); cnt returned through *cnt
__device__ inline float matched_mean(const float *iou, const float *m, int n, int lane, float *cnt) {
  float sm = 0.f, si = 0.f;
  for (int e = lane; e < n; e += 64) {
    sm += m[e];
    si += iou[e] * m[e];
  }
  sm = wave_sum(sm), si = wave_sum(si);
  const float c = fmaxf(sm, 1.0f);
  *cnt = c;
  return si / c;
}
// s_min[t] = min_{t' <= t} s[t'] with torch.cummin's index (the LAST position of the minimum so far); s_max[t] = the max of
// s[t..T) as flip(cummax(flip(s))) computes it (index: the FIRST position of the maximum among t..T-1 scanning from the end,
// i.e. the smallest t' >= t holding it)
__device__ inline void cum_ext(const float *s, int T, int t, float &smin, int &imin, float &smax, int &imax) {
  smin = s[0], imin = 0;
  for (int k = 1; k <= t; ++k)
    if (s[k] <= smin) smin = s[k], imin = k;
  smax = s[T - 1], imax = T - 1;
  for (int k = T - 2; k >= t; --k)
    if (s[k] >= smax) smax = s[k], imax = k;
}
template <bool BWD>
__global__ __launch_bounds__(256) void loss_head_kernel(LossHeadArgs a) {
  __shared__ float per[3][256];        // forward: per-image (iou_s, iou_b, conf)
  __shared__ float coef[4][2][64];     // backward, per wave: the two per-timestep coefficients of d s_out
  __shared__ int idx[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = a.T, TT = T * T;
  const float invB = 1.0f / (float)a.B, g = (BWD && a.g) ? a.g[0] : 1.0f;
  for (int b = wave; b < a.B; b += 4) {
    const float *ms_ = a.m_s + (size_t)b * TT, *mb_ = a.m_b + (size_t)b * TT, *s = a.s_out + (size_t)b * T;
    float cnt_s, cnt_b;
    const float is = matched_mean(a.iou_s + (size_t)b * TT, ms_, TT, lane, &cnt_s);
    const float ib = matched_mean(a.iou_b + (size_t)b * TT, mb_, TT, lane, &cnt_b);
    float smin = 0.f, smax = 0.f, msum = 0.f;
    int imin = 0, imax = 0;
    if (lane < T) {
      cum_ext(s, T, lane, smin, imin, smax, imax);
      for (int m = 0; m < T; ++m) msum += ms_[lane * T + m];
    }
    if constexpr (!BWD) {
      float cf = lane < T ? (-msum * logf(smin + 1e-5f) - (1.0f - msum) * logf(1.0f - smax + 1e-5f)) : 0.f;
      cf = wave_sum(cf);
      if (lane == 0) per[0][b] = is, per[1][b] = ib, per[2][b] = cf;
    } else {
      // d loss / d iou = -m / cnt / B (both heads); PairIoU's adjoint: U = sa + sb - I + 1e-5 HW, c1 = G (U + I) / U^2,
      // c0[n] = -sum_m G I / U^2
      for (int h = 0; h < 2; ++h) {
        const float *mm = h ? mb_ : ms_, *I = (h ? a.inter_b : a.inter_s) + (size_t)b * TT;
        const float *sa = (h ? a.sa_b : a.sa_s) + (size_t)b * T, *sb = (h ? a.sb_b : a.sb_s) + (size_t)b * T;
        float *c1 = (h ? a.c1_b : a.c1_s) + (size_t)b * TT, *c0 = (h ? a.c0_b : a.c0_s) + (size_t)b * T;
        const float sc = -g * invB / (h ? cnt_b : cnt_s);
        for (int e = lane; e < TT; e += 64) {
          const int n = e / T, m = e - n * T;
          const float U = sa[n] + sb[m] - I[e] + 1e-5f * (float)a.HW;
          c1[e] = sc * mm[e] * (U + I[e]) / (U * U);
        }
        if (lane < T) {
          float acc = 0.f;
          for (int m = 0; m < T; ++m) {
            const int e = lane * T + m;
            const float U = sa[lane] + sb[m] - I[e] + 1e-5f * (float)a.HW;
            acc += sc * mm[e] * I[e] / (U * U);
          }
          c0[lane] = -acc;
        }
      }
      // d conf / d s: each timestep's two terms go to the positions the cumulative min / max took their values from
      const float k = g * a.mix * invB / (float)T;
      if (lane < T) {
        coef[wave][0][lane] = -k * msum / (smin + 1e-5f);
        coef[wave][1][lane] = k * (1.0f - msum) / (1.0f - smax + 1e-5f);
        idx[wave][0][lane] = imin, idx[wave][1][lane] = imax;
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      if (lane < T) {
        float d = 0.f;
        for (int t = 0; t < T; ++t) {
          if (idx[wave][0][t] == lane) d += coef[wave][0][t];
          if (idx[wave][1][t] == lane) d += coef[wave][1][t];
        }
        a.ds[(size_t)b * T + lane] = d;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if constexpr (!BWD) {
    __syncthreads();
    if (tid == 0) {
      float is = 0.f, ib = 0.f, cf = 0.f;
      for (int b = 0; b < a.B; ++b) is += per[0][b], ib += per[1][b], cf += per[2][b];
      is *= invB, ib *= invB;
      cf = cf * invB / (float)T;
      a.pieces[0] = -ib - is + a.mix * cf;
      a.pieces[1] = -ib, a.pieces[2] = -is, a.pieces[3] = cf, a.pieces[4] = is, a.pieces[5] = ib;
    }
  }
}
}  // namespace loss
}  // namespace ra

using namespace ra;

extern "C" size_t ra_pair_stats_workspace_floats(int B, int HW) {
  if (B <= 0 || HW <= 0) return 0;
  return (size_t)B * (loss::nchunks_for(HW) + 1) * loss::kPartFloats;  // chunk records + totals
}

static int pair_stats_impl(const float *a, size_t a_img, size_t a_row, const float *b, int B, int N, int M, int HW, float *ws,
                           size_t ws_floats, float *iou_soft, float *iou_hard, float *dice_hard,
                           float *sum_a, float *sum_b, float *inter, float *sum_a_hard, void *stream) {
  if (!a || !b || !ws || B <= 0 || N <= 0 || M <= 0 || HW <= 0)
    return fail(RA_E_INVALID, "ra_pair_stats_f32: bad argument");
  if (N > loss::kMaxT || M > loss::kMaxT || HW % 4)
    return fail(RA_E_SHAPE, "ra_pair_stats_f32: N=%d M=%d (max %d), HW=%d (multiple of 4)", N, M, loss::kMaxT, HW);
  if (ws_floats < ra_pair_stats_workspace_floats(B, HW))
    return fail(RA_E_WORKSPACE, "ra_pair_stats_f32: workspace of %zu floats, need %zu", ws_floats,
                ra_pair_stats_workspace_floats(B, HW));
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15)
    return fail(RA_E_INVALID, "ra_pair_stats_f32: inputs must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const int nch = loss::nchunks_for(HW);
  const dim3 grid(nch, B);
  const int ni = N > 16 ? 2 : 1, nj = M > 16 ? 2 : 1;
  if (ni == 1 && nj == 1)
    hipLaunchKernelGGL((loss::pair_stats_kernel<1, 1>), grid, dim3(256), 0, st, a, b, N, M, HW, nch, ws, a_img, a_row);
  else if (ni == 2 && nj == 2)
    hipLaunchKernelGGL((loss::pair_stats_kernel<2, 2>), grid, dim3(256), 0, st, a, b, N, M, HW, nch, ws, a_img, a_row);
  else if (ni == 1)
    hipLaunchKernelGGL((loss::pair_stats_kernel<1, 2>), grid, dim3(256), 0, st, a, b, N, M, HW, nch, ws, a_img, a_row);
  else
    hipLaunchKernelGGL((loss::pair_stats_kernel<2, 1>), grid, dim3(256), 0, st, a, b, N, M, HW, nch, ws, a_img, a_row);
  int rc = launch_status("ra_pair_stats_f32");
  if (rc) return rc;
  float *tot = ws + (size_t)B * nch * loss::kPartFloats;
  hipLaunchKernelGGL(loss::pair_stats_reduce_kernel, dim3(ceil_div(loss::kPartFloats, 256), B), dim3(256), 0, st, ws,
                     nch, tot);
  hipLaunchKernelGGL(loss::pair_stats_finish_kernel, dim3(B), dim3(256), 0, st, tot, N, M, HW, iou_soft, iou_hard,
                     dice_hard, sum_a, sum_b, inter, sum_a_hard);
  return launch_status("ra_pair_stats_f32");
}

extern "C" int ra_pair_stats_f32(const float *a, const float *b, int B, int N, int M, int HW, float *ws,
                                 size_t ws_floats, float *iou_soft, float *iou_hard, float *dice_hard,
                                 float *sum_a, float *sum_b, float *inter, float *sum_a_hard, void *stream) {
  return pair_stats_impl(a, (size_t)N * HW, (size_t)HW, b, B, N, M, HW, ws, ws_floats, iou_soft, iou_hard, dice_hard, sum_a, sum_b, inter,
                         sum_a_hard, stream);
}

// the same with a's planes addressed through strides (floats): a[img][row] starts at img * a_img + row * a_row — the
// training step's masks of all timesteps lie timestep-major ([T,B,H,W]: a_img = H*W, a_row = B*H*W), no transposed copy
extern "C" int ra_pair_stats_strided_f32(const float *a, size_t a_img, size_t a_row, const float *b, int B, int N, int M, int HW, float *ws,
                                         size_t ws_floats, float *iou_soft, float *iou_hard, float *dice_hard, float *sum_a, float *sum_b,
                                         float *inter, float *sum_a_hard, void *stream) {
  if ((a_img | a_row) & 3) return fail(RA_E_SHAPE, "ra_pair_stats_strided_f32: strides must be multiples of 4 floats");
  return pair_stats_impl(a, a_img, a_row, b, B, N, M, HW, ws, ws_floats, iou_soft, iou_hard, dice_hard, sum_a, sum_b, inter, sum_a_hard,
                         stream);
}

extern "C" size_t ra_gt_box_workspace_floats(int B, int T) {
  if (B <= 0 || T <= 0) return 0;
  return (size_t)B * T * loss::kGtSplit * 5;
}

extern "C" int ra_gt_box_f32(const float *y_gt, int B, int T, int H, int W, float padding_ratio,
                             float min_padding, float *ws, size_t ws_floats, float *params, float *box,
                             void *stream) {
  if (!y_gt || !params || !ws || B <= 0 || T <= 0 || H <= 0 || W <= 0)
    return fail(RA_E_INVALID, "ra_gt_box_f32: bad argument");
  if (ws_floats < ra_gt_box_workspace_floats(B, T))
    return fail(RA_E_WORKSPACE, "ra_gt_box_f32: workspace of %zu floats, need %zu", ws_floats,
                ra_gt_box_workspace_floats(B, T));
  if ((H * W) % 4 || (reinterpret_cast<uintptr_t>(y_gt) & 15) || (box && (reinterpret_cast<uintptr_t>(box) & 15)))
    return fail(RA_E_SHAPE, "ra_gt_box_f32: H*W must be a multiple of 4 and the tensors 16-byte aligned");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(loss::gt_box_reduce_kernel, dim3(loss::kGtSplit, B * T), dim3(256), 0, st, y_gt, H, W, ws);
  hipLaunchKernelGGL(loss::gt_box_finish_kernel, dim3(ceil_div(B * T, 64)), dim3(64), 0, st, ws, B * T,
                     padding_ratio, min_padding, params);
  int rc = launch_status("ra_gt_box_f32");
  if (rc || !box) return rc;
  const int gx = ceil_div(H * W, 256 * 4 * 4);
  hipLaunchKernelGGL(loss::gt_box_fill_kernel, dim3(gx, B * T), dim3(256), 0, st, params, H, W, box);
  return launch_status("ra_gt_box_f32");
}

extern "C" int ra_knob_setup_f32(const float *gt_box_ws, int B, int T, const float *pad, const float *shift, const float *u_box,
                                 const float *u_segm, const float *sched, float min_padding, int timescale, float *ctr, float *size,
                                 float *knob_box, float *knob_segm, void *stream) {
  if (!gt_box_ws || !pad || !shift || !u_box || !u_segm || !sched || !ctr || !size || !knob_box || !knob_segm || B <= 0 || T <= 0)
    return fail(RA_E_INVALID, "ra_knob_setup_f32: bad argument");
  hipLaunchKernelGGL(loss::knob_setup_kernel, dim3(ceil_div(B * T, 64)), dim3(64), 0, as_stream(stream), gt_box_ws, B * T, T, pad, shift,
                     u_box, u_segm, sched, min_padding, timescale, ctr, size, knob_box, knob_segm);
  return launch_status("ra_knob_setup_f32");
}

extern "C" size_t ra_box_iou_rects_workspace_floats(int B) {
  return B > 0 ? (size_t)B * loss::kRectBlocks * (loss::kMaxT + 1) : 0;
}

extern "C" int ra_box_iou_rects_f32(const float *box, const float *params, int B, int T, int H, int W, float *ws,
                                    size_t ws_floats, float *iou, void *stream) {
  if (!box || !params || !iou || !ws || B <= 0 || T <= 0 || H <= 0 || W <= 0)
    return fail(RA_E_INVALID, "ra_box_iou_rects_f32: bad argument");
  if (T > loss::kMaxT || W % 4 || (reinterpret_cast<uintptr_t>(box) & 15))
    return fail(RA_E_SHAPE, "ra_box_iou_rects_f32: T <= %d, W %% 4 == 0 and a 16-byte aligned map required", loss::kMaxT);
  if (ws_floats < ra_box_iou_rects_workspace_floats(B)) return fail(RA_E_WORKSPACE, "ra_box_iou_rects_f32: workspace too small");
  hipStream_t st = as_stream(stream);
  // workgroups per image: about four 16-byte loads per thread — a workgroup's fixed cost is its T + 1 cross-wave reductions, and 64
  // workgroups on a 128 x 448 map (box_model at KITTI size: under one load per thread) made the launch 38 us for 7 MB (round 6)
  int nblk = ceil_div(H * W / 4, 256 * 4);
  nblk = nblk < 1 ? 1 : (nblk > loss::kRectBlocks ? loss::kRectBlocks : nblk);
  if (T <= 8)
    hipLaunchKernelGGL(loss::box_iou_rects_kernel<8>, dim3(nblk, B), dim3(256), 0, st, box, params, T, H, W, ws, nblk);
  else if (T <= 16)
    hipLaunchKernelGGL(loss::box_iou_rects_kernel<16>, dim3(nblk, B), dim3(256), 0, st, box, params, T, H, W, ws, nblk);
  else
    hipLaunchKernelGGL(loss::box_iou_rects_kernel<32>, dim3(nblk, B), dim3(256), 0, st, box, params, T, H, W, ws, nblk);
  hipLaunchKernelGGL(loss::box_iou_rects_finish_kernel, dim3(B), dim3(64), 0, st, ws, params, T, H, W, iou, nblk);
  return launch_status("ra_box_iou_rects_f32");
}

extern "C" size_t ra_segm_match_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const size_t w = (size_t)B * N * N * sizeof(float);                         // quantised weights
  const size_t cov = (size_t)B * 2 * N * sizeof(float);                       // covers (discarded)
  return ((w + cov + 255) / 256) * 256 + ra_hungarian_dev_workspace_bytes(B, N, N);
}

extern "C" int ra_segm_match_f32(const float *iou, const float *s_gt, int B, int N, void *ws, size_t ws_bytes,
                                 float *match, int *status, void *stream) {
  if (!iou || !s_gt || !ws || !match || !status || B <= 0 || N <= 0)
    return fail(RA_E_INVALID, "ra_segm_match_f32: bad argument");
  if (ws_bytes < ra_segm_match_workspace_bytes(B, N))
    return fail(RA_E_WORKSPACE, "ra_segm_match_f32: workspace of %zu bytes, need %zu", ws_bytes,
                ra_segm_match_workspace_bytes(B, N));
  hipStream_t st = as_stream(stream);
  float *w = static_cast<float *>(ws);
  float *cx = w + (size_t)B * N * N, *cy = cx + (size_t)B * N;
  const size_t head = (((size_t)B * N * N + (size_t)B * 2 * N) * sizeof(float) + 255) / 256 * 256;
  char *hws = static_cast<char *>(ws) + head;
  const int total = B * N * N;
  hipLaunchKernelGGL(loss::match_pre_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, iou, s_gt, N, total, w);
  int rc = launch_status("ra_segm_match_f32");
  if (rc) return rc;
  rc = ra_hungarian_f32_dev(w, B, N, N, match, cx, cy, status, hws, ws_bytes - head, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(loss::match_post_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, match, s_gt, N, total);
  return launch_status("ra_segm_match_f32");
}

// ---- f_segm_match with the solver on HOST cores, as a host node of the stream / captured graph (round 6) --------------------
// The device solver runs one wave per problem and a launch lasts as long as its slowest problem: 1.7 ms for the 16 x 16 problems
// of a cfg4 training step, serial on the step's critical path (every gradient waits for the matching), where one host core
// needs ~0.5 ms per problem (hungarian.cc's control flow is serial and branchy: a CPU core runs it 3 x faster than a wave).
// Here the B problems go to host threads in parallel: precondition kernel -> D2H copy into the caller's PINNED block -> host
// function (hipLaunchHostFunc: a host node when the stream is capturing) -> H2D copy -> re-mask kernel.  Results are those of
// ra_hungarian_f32 — the same source as the device solver (RA_HUNG solve()), bit-identical by construction and by test.
// The pinned block: [ctl: 64 bytes | w: B N N floats | match: B N N floats | status: B ints]; it carries the host function's
// arguments, so it must stay allocated (and untouched by the caller) for as long as a graph that captured the call lives.
namespace ra {
namespace loss {
struct HostMatchCtl {
  int B, N, threads, pad;
  float *w, *match;
  int *status;
};
static_assert(sizeof(HostMatchCtl) <= 64, "control block");
static void host_match_fn(void *p) {
  const HostMatchCtl c = *static_cast<const HostMatchCtl *>(p);
  const int nthr = c.threads < 1 ? 1 : (c.threads > c.B ? c.B : c.threads);
  auto work = [&c, nthr](int t0) {
    std::vector<float> cx(c.N), cy(c.N);
    for (int b = t0; b < c.B; b += nthr) {
      const size_t o = (size_t)b * c.N * c.N;
      c.status[b] = ra_hungarian_f32(c.w + o, 1, c.N, c.N, c.match + o, cx.data(), cy.data());
    }
  };
  if (nthr == 1) {
    work(0);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(nthr - 1);
  for (int t = 1; t < nthr; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &t : th) t.join();
}
}  // namespace loss
}  // namespace ra

extern "C" size_t ra_segm_match_host_block_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return 64 + (size_t)B * N * N * 8 + (size_t)B * 4;
}

extern "C" int ra_segm_match_host_f32(const float *iou, const float *s_gt, int B, int N, float *w_dev, void *pinned, size_t pinned_bytes,
                                      int threads, float *match, int *status, void *stream) {
  if (!iou || !s_gt || !w_dev || !pinned || !match || !status || B <= 0 || N <= 0)
    return fail(RA_E_INVALID, "ra_segm_match_host_f32: bad argument");
  if (pinned_bytes < ra_segm_match_host_block_bytes(B, N) || (reinterpret_cast<uintptr_t>(pinned) & 15))
    return fail(RA_E_WORKSPACE, "ra_segm_match_host_f32: pinned block of %zu bytes (16-byte aligned), need %zu", pinned_bytes,
                ra_segm_match_host_block_bytes(B, N));
  hipStream_t st = as_stream(stream);
  const int total = B * N * N;
  char *blk = static_cast<char *>(pinned);
  loss::HostMatchCtl *ctl = reinterpret_cast<loss::HostMatchCtl *>(blk);
  float *hw = reinterpret_cast<float *>(blk + 64), *hm = hw + total;
  int *hs = reinterpret_cast<int *>(hm + total);
  // (written at enqueue time: a graph that captured an earlier call on this block replays with the same values)
  ctl->B = B;
  ctl->N = N;
  ctl->threads = threads;
  ctl->w = hw;
  ctl->match = hm;
  ctl->status = hs;
  hipLaunchKernelGGL(loss::match_pre_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, iou, s_gt, N, total, w_dev);
  int rc = launch_status("ra_segm_match_host_f32");
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync(hw, w_dev, (size_t)total * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipLaunchHostFunc(st, loss::host_match_fn, ctl);
  if (e == hipSuccess) e = hipMemcpyAsync(match, hm, (size_t)total * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(status, hs, (size_t)B * 4, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return fail((int)e, "ra_segm_match_host_f32: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(loss::match_post_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, match, s_gt, N, total);
  return launch_status("ra_segm_match_host_f32");
}

extern "C" size_t ra_loss_stats_workspace_floats(int B) { return B > 0 ? (size_t)B * loss::kNTerms : 0; }

extern "C" int ra_loss_stats_f32(const float *iou_soft, const float *iou_hard, const float *dice,
                                 const float *match_real, const float *iou_box, const float *match_box,
                                 const float *s_out, const float *s_gt, const float *sum_gt, int B, int T,
                                 int fixed_order, int segm_loss_fn, float loss_mix_ratio, float *ws,
                                 size_t ws_floats, float *out, void *stream) {
  if (!iou_soft || !iou_hard || !dice || !match_real || !iou_box || !match_box || !s_out || !s_gt || !sum_gt ||
      !out || !ws || B <= 0 || T <= 0)
    return fail(RA_E_INVALID, "ra_loss_stats_f32: bad argument");
  if (ws_floats < ra_loss_stats_workspace_floats(B))
    return fail(RA_E_WORKSPACE, "ra_loss_stats_f32: workspace of %zu floats, need %zu", ws_floats,
                ra_loss_stats_workspace_floats(B));
  if (T > loss::kMaxT) return fail(RA_E_SHAPE, "ra_loss_stats_f32: T=%d (max %d)", T, loss::kMaxT);
  if (segm_loss_fn != 0 && segm_loss_fn != 1) return fail(RA_E_INVALID, "ra_loss_stats_f32: segm_loss_fn");
  hipLaunchKernelGGL(loss::loss_stats_image_kernel, dim3(B), dim3(64), 0, as_stream(stream), iou_soft, iou_hard,
                     dice, match_real, iou_box, match_box, s_out, s_gt, sum_gt, T, fixed_order, ws);
  hipLaunchKernelGGL(loss::loss_stats_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), ws, B, T, segm_loss_fn,
                     loss_mix_ratio, out);
  return launch_status("ra_loss_stats_f32");
}

/* see include/recattend.h */
extern "C" int ra_loss_head_f32(const float *iou_s, const float *iou_b, const float *m_s, const float *m_b, const float *s_out, int B, int T,
                                float mix, float *pieces, void *stream) {
  if (!iou_s || !iou_b || !m_s || !m_b || !s_out || !pieces || B <= 0 || T <= 0) return fail(RA_E_INVALID, "ra_loss_head_f32: bad argument");
  if (B > 256 || T > 64) return fail(RA_E_SHAPE, "ra_loss_head_f32: B %d (max 256), T %d (max 64)", B, T);
  loss::LossHeadArgs a{};
  a.iou_s = iou_s, a.iou_b = iou_b, a.m_s = m_s, a.m_b = m_b, a.s_out = s_out, a.B = B, a.T = T, a.mix = mix, a.pieces = pieces;
  hipLaunchKernelGGL(loss::loss_head_kernel<false>, dim3(1), dim3(256), 0, as_stream(stream), a);
  return launch_status("ra_loss_head_f32");
}
extern "C" int ra_loss_head_bwd_f32(const float *g, const float *m_s, const float *m_b, const float *s_out, const float *inter_s,
                                    const float *sum_a_s, const float *sum_b_s, const float *inter_b, const float *sum_a_b,
                                    const float *sum_b_b, int B, int T, int HW, float mix, float *c1_s, float *c0_s, float *c1_b,
                                    float *c0_b, float *d_s_out, void *stream) {
  if (!m_s || !m_b || !s_out || !inter_s || !sum_a_s || !sum_b_s || !inter_b || !sum_a_b || !sum_b_b || !c1_s || !c0_s || !c1_b || !c0_b ||
      !d_s_out || B <= 0 || T <= 0 || HW <= 0)
    return fail(RA_E_INVALID, "ra_loss_head_bwd_f32: bad argument");
  if (B > 256 || T > 64) return fail(RA_E_SHAPE, "ra_loss_head_bwd_f32: B %d (max 256), T %d (max 64)", B, T);
  loss::LossHeadArgs a{};
  // (the matched means only need the matches here: the IoU pointers may alias them)
  a.iou_s = m_s, a.iou_b = m_b, a.m_s = m_s, a.m_b = m_b, a.s_out = s_out, a.inter_s = inter_s, a.sa_s = sum_a_s, a.sb_s = sum_b_s;
  a.inter_b = inter_b, a.sa_b = sum_a_b, a.sb_b = sum_b_b, a.g = g, a.B = B, a.T = T, a.HW = HW, a.mix = mix;
  a.c1_s = c1_s, a.c0_s = c0_s, a.c1_b = c1_b, a.c0_b = c0_b, a.ds = d_s_out;
  hipLaunchKernelGGL(loss::loss_head_kernel<true>, dim3(1), dim3(256), 0, as_stream(stream), a);
  return launch_status("ra_loss_head_bwd_f32");
}
