// K3 / K5 (direct form) — Gaussian attention read and write driven straight from the attention
// record: the separable filter weights are computed on the fly (one v_exp per tap-pixel pair)
// instead of being read back from dense [L,48] tables, so one timestep needs TWO launches
// (extract, paste) instead of four (filters, extract, paste_u, paste), and the canvas can live in
// its own [B,H,W] plane: the paste then moves exactly its algorithmic bytes (read canvas, write
// canvas, write y_out: 12 B per pixel) instead of read-modify-writing 16-byte packed pixel records.
//   modellib.get_gaussian_filter modellib.py:581-612, extract_patch :615-641,
//   full_model.py:778-789 (read), :810-818,:843-845 (write + canvas), :738-741 (attention box).
// Same banding rule as ra_attn.hip: taps whose weight is below exp(-30) of the peak are skipped.
#include <cstdlib>

#include "ra_common.h"

namespace ra {
namespace attnd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kBandLog = 30.0f;
constexpr float kInvSqrt2Pi = 0.3989422804014327f;

struct Axis {  // one axis of one example's filter bank
  float ctr, step, inv_step, half, inv2var, norm, R;
  int L, F;
  __device__ inline float mu(int j) const { return ctr + step * ((float)j - half); }
  __device__ inline float w(float l, int j) const {  // modellib.py:610-611
    const float d = l - mu(j);
    return norm * __expf(-d * d * inv2var);
  }
  // pixel band [lo, hi) of tap j
  __device__ inline void band(int j, int &lo, int &hi) const {
    const float m = mu(j);
    float a = ceilf(m - R), c = floorf(m + R) + 1.0f;
    a = fminf(fmaxf(a, 0.0f), (float)L);
    c = fminf(fmaxf(c, 0.0f), (float)L);
    if (!(a == a) || !(c == c)) {
      a = 0.0f;
      c = (float)L;
    }
    lo = (int)a;
    hi = (int)c > lo ? (int)c : lo;
  }
  // tap range [jlo, jhi) whose band contains pixel l (widened by one tap each side: extra terms
  // are harmless, missing ones are not)
  __device__ inline void taps(int l, int &jlo, int &jhi) const {
    float a = ((float)l - R - ctr) * inv_step + half, c = ((float)l + R - ctr) * inv_step + half;
    if (!(a == a) || !(c == c) || !(step > 0.0f)) {
      jlo = 0;
      jhi = F;
      return;
    }
    a = fminf(fmaxf(floorf(a) - 1.0f, 0.0f), (float)F);
    c = fminf(fmaxf(ceilf(c) + 2.0f, 0.0f), (float)F);
    jlo = (int)a;
    jhi = (int)c;
  }
};

__device__ inline Axis make_axis(const float *rec, int axis, int L, int F) {
  Axis A;
  const float var = __expf(rec[4 + axis]);
  A.ctr = rec[0 + axis];
  A.step = (rec[2 + axis] + 1.0f) / (float)F;       // modellib.py:599
  A.inv_step = 1.0f / A.step;
  A.half = ((float)F - 1.0f) / 2.0f;
  A.inv2var = 0.5f / var;
  A.norm = kInvSqrt2Pi / sqrtf(var);                // 1/sqrt(var)/sqrt(2 pi)
  A.R = sqrtf(2.0f * kBandLog * var);
  A.L = L;
  A.F = F;
  return A;
}

constexpr int kColPass = 256;  // image columns per pass of the extract kernel (one per thread)

// patch[b,j,i,4cg..] = gamma * sum_l sum_w fy(l,j) X[b,l,w,4cg..] fx(w,i)
// Workgroup = (TJ consecutive taps j, channel group cg, image b); thread = image column.
// Stage 1: every thread streams its column over the union of the TJ taps' row bands (adjacent taps
// share most of their rows: one load feeds TJ accumulators) with 16 independent 16-byte loads in
// flight — the kernel is bounded by its dependent load chain, not by bytes.  Stage 2: the TJ x Fw
// outputs contract the column sums with fx through LDS.
template <int TJ>
__global__ __launch_bounds__(256) void extract_direct_kernel(const float *img, int Ci, int chan0,
                                                              const float *canvas, int canvas_chan,
                                                              const float *attn, int H, int W, int Fh,
                                                              int Fw, int Cp, int use_gamma,
                                                              float *patch) {
  __shared__ f32x4 tl[TJ][kColPass];
  __shared__ float fyw[TJ][256];
  const int t = threadIdx.x, j0 = blockIdx.x * TJ, cg = blockIdx.y, b = blockIdx.z;
  const float *rec = attn + (size_t)b * RA_ATTN_STRIDE;
  const Axis Ay = make_axis(rec, 0, H, Fh), Ax = make_axis(rec, 1, W, Fw);
  int l0, l1, w0, w1, tmp;
  Ay.band(j0, l0, tmp);
  Ay.band(j0 + TJ - 1, tmp, l1);  // tap centres are monotone in j: the union is one interval
  if (l1 < l0) l1 = l0;
  Ax.band(0, w0, tmp);
  Ax.band(Fw - 1, tmp, w1);
  const float *imb = img + (size_t)b * H * W * Ci + chan0 + 4 * cg;
  const bool use_canvas = canvas != nullptr && (canvas_chan >= chan0 + 4 * cg) && (canvas_chan < chan0 + 4 * cg + 4);
  const int cslot = canvas_chan - (chan0 + 4 * cg);
  const float *cvb = canvas ? canvas + (size_t)b * H * W : nullptr;

  // stage-2 ownership: output (tap tj2, column i) for t < TJ * Fw
  const int tj2 = t / Fw, oi = t - tj2 * Fw;
  const bool owner = t < TJ * Fw;
  int bi_lo = 0, bi_hi = 0;
  if (owner) Ax.band(oi, bi_lo, bi_hi);
  f32x4 P = f32x4{0, 0, 0, 0};

  for (int wp = w0; wp < w1; wp += kColPass) {
    const int wend = (wp + kColPass < w1) ? wp + kColPass : w1;
    const int w = wp + t;
    const bool col_ok = w < wend;
    f32x4 acc[TJ];
#pragma unroll
    for (int k = 0; k < TJ; ++k) acc[k] = f32x4{0, 0, 0, 0};
    for (int lr = l0; lr < l1; lr += 256) {  // rows in chunks of 256 (weights staged in LDS)
      const int nrow = (l1 - lr) < 256 ? (l1 - lr) : 256;
      __syncthreads();  // previous chunk's weights fully consumed
      if (t < nrow) {
#pragma unroll
        for (int k = 0; k < TJ; ++k) {
          int a, c;
          Ay.band(j0 + k, a, c);
          const int l = lr + t;
          fyw[k][t] = (l >= a && l < c) ? Ay.w((float)l, j0 + k) : 0.0f;  // outside tap k's own band
        }
      }
      __syncthreads();
      constexpr int U = 16;
      for (int r0 = 0; r0 < nrow; r0 += U) {
        f32x4 xv[U];
        float cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool ok = col_ok & (r0 + u < nrow);
          const int rr = ok ? r0 + u : 0, ww = ok ? w : wp;
          xv[u] = *reinterpret_cast<const f32x4 *>(imb + ((size_t)(lr + rr) * W + ww) * Ci);
          cv[u] = use_canvas ? cvb[(size_t)(lr + rr) * W + ww] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (col_ok & (r0 + u < nrow)) {
            f32x4 x = xv[u];
            if (use_canvas) {
              x.x = cslot == 0 ? cv[u] : x.x;
              x.y = cslot == 1 ? cv[u] : x.y;
              x.z = cslot == 2 ? cv[u] : x.z;
              x.w = cslot == 3 ? cv[u] : x.w;
            }
#pragma unroll
            for (int k = 0; k < TJ; ++k) acc[k] += fyw[k][r0 + u] * x;
          }
        }
      }
    }
    __syncthreads();  // previous pass's stage 2 done with tl
#pragma unroll
    for (int k = 0; k < TJ; ++k) tl[k][t] = acc[k];
    __syncthreads();
    // stage 2: P[tj2, oi] += sum_{w in pass and band(oi)} tl[tj2][w] * fx(w, oi)
    if (owner) {
      const int a = bi_lo > wp ? bi_lo : wp, c = bi_hi < wend ? bi_hi : wend;
      for (int ww = a; ww < c; ++ww) P += Ax.w((float)ww, oi) * tl[tj2][ww - wp];
    }
  }
  if (owner && j0 + tj2 < Fh) {
    const float gamma = use_gamma ? rec[6] : 1.0f;
    *reinterpret_cast<f32x4 *>(patch + (((size_t)b * Fh + j0 + tj2) * Fw + oi) * Cp + 4 * cg) = gamma * P;
  }
}

__device__ inline float sigmoidf(float z) { return 1.0f / (1.0f + __expf(-z)); }

// y[b,l,w] = sigmoid(e^g * sum_j sum_i fy(l,j) P[j,i] fx(w,i) + beta) [* (1 - canvas)];
// canvas = max(canvas, y).  One workgroup per kPasteRows image rows.  MODE 0: paste, 1: attention box.
struct PasteGeo {
  int rows, threads;
};
inline PasteGeo paste_geo() {  // RA_PASTE_GEO=<rows>,<threads>: tuning aid
  static PasteGeo g = {0, 0};
  if (!g.rows) {
    g = PasteGeo{4, 256};
    if (const char *e = getenv("RA_PASTE_GEO")) sscanf(e, "%d,%d", &g.rows, &g.threads);
  }
  return g;
}
// Optional rider on the paste launch: the score MLP of the same timestep (full_model.py:794,821-822:
// s = sigmoid([h | h_core] . w + b)) runs in one extra workgroup per image instead of its own launch.
struct ScoreArgs {
  const float *h, *core, *w, *bias;  // h [B,K0], core [B,K1], w [K0+K1], bias [1]
  float *s_out;                       // element b at s_out[b * stride]
  int K0, K1;
  size_t stride;
};

template <int MODE>
__global__ __launch_bounds__(256) void paste_direct_kernel(const float *patch, int Cp, int pc,
                                                            const float *attn, int H, int W, int Fh,
                                                            int Fw, float beta, int disable_overwrite,
                                                            float *canvas, float *img, int Ci,
                                                            int canvas_chan, float *y_out,
                                                            size_t y_stride_b, int flags, int kPasteRows,
                                                            ScoreArgs sc) {
  extern __shared__ float V[];  // [kPasteRows][Fw]:  V[r][i] = sum_j fy(l0 + r, j) P[j,i]
  const int l0 = blockIdx.x * kPasteRows, b = blockIdx.y, t = threadIdx.x;
  if (l0 >= H) {  // the rider workgroup (grid.x is one larger when a score is requested)
    float s = 0.0f;
    const int K = sc.K0 + sc.K1;
    for (int k = t; k < K; k += blockDim.x) {
      const float xv = (k < sc.K0) ? sc.h[(size_t)b * sc.K0 + k] : sc.core[(size_t)b * sc.K1 + (k - sc.K0)];
      s += xv * sc.w[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) V[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      float tot = sc.bias ? sc.bias[0] : 0.0f;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot += V[wv];
      sc.s_out[(size_t)b * sc.stride] = sigmoidf(tot);
    }
    return;
  }
  const float *rec = attn + (size_t)b * RA_ATTN_STRIDE;
  const Axis Ay = make_axis(rec, 0, H, Fh), Ax = make_axis(rec, 1, W, Fw);
  const int nrow = (H - l0) < kPasteRows ? (H - l0) : kPasteRows;
  // Pixels no tap reaches have y = sigmoid(beta) and leave a canvas that is already >= sigmoid(beta)
  // unchanged: with both promises from the caller (flags) only the window is touched.
  const bool skip_dead = (flags & RA_PASTE_Y_PREFILLED) && (MODE != 0 || !canvas || (flags & RA_PASTE_CANVAS_FLOORED)) &&
                         (MODE != 0 || canvas || !img);
  int jall_lo, jall_hi, jtmp;
  Ay.taps(l0, jall_lo, jtmp);
  Ay.taps(l0 + nrow - 1, jtmp, jall_hi);  // tap ranges are monotone in l: union over the rows
  if (skip_dead && jall_lo >= jall_hi) return;
  int wbeg = 0, wend = W;
  if (skip_dead) {
    int w0, w1, tmp;
    Ax.band(0, w0, tmp);
    Ax.band(Fw - 1, tmp, w1);
    wbeg = w0 & ~3;
    wend = (w1 + 3) & ~3;
    wend = wend < W ? wend : W;
  }
  const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(y_out) & 15) == 0) && ((y_stride_b & 3) == 0) &&
                   (!canvas || (reinterpret_cast<uintptr_t>(canvas) & 15) == 0);
  const int ngrp = (wend - wbeg + 3) >> 2;  // float4 column groups of the window
  // the canvas values of this thread's first item are fetched before the V phase, so that the two
  // global round trips of the kernel (patch, canvas) overlap
  f32x4 cv_pre = f32x4{0, 0, 0, 0};
  if (MODE == 0 && canvas && vec && t < nrow * ngrp) {
    const int r = t / ngrp, w4 = wbeg + 4 * (t - r * ngrp);
    cv_pre = *reinterpret_cast<const f32x4 *>(canvas + ((size_t)b * H + l0 + r) * W + w4);
  }
  for (int e = t; e < nrow * Fw; e += blockDim.x) {
    const int r = e / Fw, i = e - r * Fw, l = l0 + r;
    int jlo, jhi;
    Ay.taps(l, jlo, jhi);
    float s = 0.0f;
    if (MODE == 0) {
      // batches of 8 independent loads: a dependent load-FMA chain would cost one L2 round trip
      // per tap, and the taps of a row are the whole critical path of this kernel
      const float *pb = patch + ((size_t)b * Fh * Fw + i) * Cp + pc;
      for (int j0 = jlo; j0 < jhi; j0 += 8) {
        float pvv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = (j0 + u < jhi) ? j0 + u : jhi - 1;
          pvv[u] = pb[(size_t)j * Fw * Cp];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + u < jhi) s += Ay.w((float)l, j0 + u) * pvv[u];
      }
    } else {
      for (int j = jlo; j < jhi; ++j) s += Ay.w((float)l, j);  // P == 1 (const_ones)
    }
    V[e] = s;
  }
  __syncthreads();
  const float gain = (MODE == 0) ? __expf(rec[8]) : rec[7];
  const float y_dead = sigmoidf(beta);
  for (int e = t; e < nrow * ngrp; e += blockDim.x) {
    const int r = e / ngrp, w4 = wbeg + 4 * (e - r * ngrp), l = l0 + r;
    int jlo, jhi;
    Ay.taps(l, jlo, jhi);
    const bool row_live = jlo < jhi;
    if (skip_dead && !row_live) continue;
    float *yrow = y_out + (size_t)b * y_stride_b + (size_t)l * W;
    float *crow = canvas ? canvas + ((size_t)b * H + l) * W : nullptr;
    float *prow = (!canvas && img && canvas_chan >= 0) ? img + ((size_t)b * H + l) * W * Ci + canvas_chan : nullptr;
    const float *Vr = V + r * Fw;
    f32x4 cv = f32x4{0, 0, 0, 0};
    if (MODE == 0) {
      if (crow && vec) cv = (e == t) ? cv_pre : *reinterpret_cast<const f32x4 *>(crow + w4);
      else
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (w4 + k < W) cv[k] = crow ? crow[w4 + k] : (prow ? prow[(size_t)(w4 + k) * Ci] : 0.0f);
    }
    f32x4 y;
    {
      // one tap loop for the 4 pixels (tap ranges are monotone in w: take the union; the extra
      // terms are below e^-30 of the peak) so that the four exponentials per tap are independent
      f32x4 sacc = f32x4{0, 0, 0, 0};
      int ilo = 0, ihi = 0;
      if (row_live) {
        int t0, t1;
        Ax.taps(w4, ilo, t0);
        Ax.taps(w4 + 3, t1, ihi);
        for (int i = ilo; i < ihi; ++i) {
          const float vi = Vr[i];
#pragma unroll
          for (int k = 0; k < 4; ++k) sacc[k] += vi * Ax.w((float)(w4 + k), i);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = (row_live && ilo < ihi) ? sigmoidf(gain * sacc[k] + beta) : y_dead;
        if (MODE == 0 && disable_overwrite) v *= (1.0f - cv[k]);
        y[k] = v;
        cv[k] = fmaxf(cv[k], v);
      }
    }
    if (vec) {
      *reinterpret_cast<f32x4 *>(yrow + w4) = y;
      if (MODE == 0 && crow) *reinterpret_cast<f32x4 *>(crow + w4) = cv;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (w4 + k < W) {
          yrow[w4 + k] = y[k];
          if (MODE == 0 && crow) crow[w4 + k] = cv[k];
        }
    }
    if (MODE == 0 && prow) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (w4 + k < W) prow[(size_t)(w4 + k) * Ci] = cv[k];
    }
  }
}

}  // namespace attnd
}  // namespace ra

using namespace ra;

extern "C" int ra_extract_direct_f32(const float *img, int Ci, int chan0, const float *canvas,
                                     int canvas_chan, const float *attn_rec, int B, int H, int W, int Fh,
                                     int Fw, int Cp, int use_gamma, float *patch, void *stream) {
  if (!img || !attn_rec || !patch || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0)
    return fail(RA_E_INVALID, "ra_extract_direct_f32: bad argument");
  if (Ci % 4 || Cp % 4 || chan0 % 4 || chan0 + Cp > Ci || Cp <= 0 || Fw > 256)
    return fail(RA_E_SHAPE, "ra_extract_direct_f32: Ci=%d chan0=%d Cp=%d Fw=%d", Ci, chan0, Cp, Fw);
  if (Fh % 4 == 0 && 4 * Fw <= 256)
    hipLaunchKernelGGL(attnd::extract_direct_kernel<4>, dim3(Fh / 4, Cp / 4, B), dim3(256), 0, as_stream(stream),
                       img, Ci, chan0, canvas, canvas_chan, attn_rec, H, W, Fh, Fw, Cp, use_gamma, patch);
  else
    hipLaunchKernelGGL(attnd::extract_direct_kernel<1>, dim3(Fh, Cp / 4, B), dim3(256), 0, as_stream(stream),
                       img, Ci, chan0, canvas, canvas_chan, attn_rec, H, W, Fh, Fw, Cp, use_gamma, patch);
  return launch_status("ra_extract_direct_f32");
}

extern "C" int ra_paste_direct_f32(const float *patch, int Cp, int pc, const float *attn_rec, int B, int H,
                                   int W, int Fh, int Fw, float beta, int disable_overwrite,
                                   float *canvas, float *img, int Ci, int canvas_chan, float *y_out,
                                   size_t y_stride_b, int flags, void *stream) {
  if (!patch || !attn_rec || !y_out || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0 || Cp <= 0 ||
      pc < 0 || pc >= Cp)
    return fail(RA_E_INVALID, "ra_paste_direct_f32: bad argument");
  const attnd::PasteGeo pg = attnd::paste_geo();
  hipLaunchKernelGGL(attnd::paste_direct_kernel<0>, dim3(ceil_div(H, pg.rows), B), dim3(pg.threads),
                     pg.rows * Fw * sizeof(float),
                     as_stream(stream), patch, Cp, pc, attn_rec, H, W, Fh, Fw, beta, disable_overwrite, canvas,
                     img, Ci, canvas_chan, y_out, y_stride_b, disable_overwrite ? (flags & ~RA_PASTE_Y_PREFILLED) : flags,
                     pg.rows, attnd::ScoreArgs{});
  return launch_status("ra_paste_direct_f32");
}

extern "C" int ra_attn_box_direct_f32(const float *attn_rec, int B, int H, int W, int Fh, int Fw, float beta,
                                      float *box_out, size_t stride_b, void *stream) {
  if (!attn_rec || !box_out || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0)
    return fail(RA_E_INVALID, "ra_attn_box_direct_f32: bad argument");
  const attnd::PasteGeo pg = attnd::paste_geo();
  hipLaunchKernelGGL(attnd::paste_direct_kernel<1>, dim3(ceil_div(H, pg.rows), B), dim3(pg.threads),
                     pg.rows * Fw * sizeof(float),
                     as_stream(stream), nullptr, 1, 0, attn_rec, H, W, Fh, Fw, beta, 0, nullptr, nullptr, 0, -1,
                     box_out, stride_b, 0, pg.rows, attnd::ScoreArgs{});
  return launch_status("ra_attn_box_direct_f32");
}

extern "C" int ra_paste_score_direct_f32(const float *patch, int Cp, int pc, const float *attn_rec, int B, int H,
                                         int W, int Fh, int Fw, float beta, int disable_overwrite, float *canvas,
                                         float *y_out, size_t y_stride_b, int flags, const float *h, int K0,
                                         const float *core, int K1, const float *w, const float *bias,
                                         float *s_out, size_t s_stride_b, void *stream) {
  if (!patch || !attn_rec || !y_out || !canvas || !h || !w || !s_out || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 ||
      Fw <= 0 || Cp <= 0 || pc < 0 || pc >= Cp || K0 <= 0 || K1 < 0 || (K1 > 0 && !core))
    return fail(RA_E_INVALID, "ra_paste_score_direct_f32: bad argument");
  const attnd::PasteGeo pg = attnd::paste_geo();
  attnd::ScoreArgs sc{h, core, w, bias, s_out, K0, K1, s_stride_b};
  const size_t lds = (size_t)(pg.rows * Fw > 8 ? pg.rows * Fw : 8) * sizeof(float);
  hipLaunchKernelGGL(attnd::paste_direct_kernel<0>, dim3(ceil_div(H, pg.rows) + 1, B), dim3(pg.threads), lds,
                     as_stream(stream), patch, Cp, pc, attn_rec, H, W, Fh, Fw, beta, disable_overwrite, canvas,
                     static_cast<float *>(nullptr), 0, -1, y_out, y_stride_b,
                     disable_overwrite ? (flags & ~RA_PASTE_Y_PREFILLED) : flags, pg.rows, sc);
  return launch_status("ra_paste_score_direct_f32");
}
