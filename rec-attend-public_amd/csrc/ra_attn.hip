// K3 / K5 — Gaussian (DRAW-style) attention read and write for gfx950.
//   modellib.get_gaussian_filter  modellib.py:581-612   -> attn_filters / gaussian_filter
//   modellib.extract_patch        modellib.py:615-641   -> extract_patch (banded), extract_dense
//   full_model.py:810-818,843-845 (paste, sigmoid, overwrite mask, canvas max) -> paste_u + paste
//   full_model.py:738-741 / box_model.py:479-482 (attention box)               -> attn_box
// These are HBM/L2-bound streaming kernels: no MFMA.  The filters are banded (sigma ~ 1-3 px,
// 48 taps): the dense tables are still materialised (they ARE the reference operator and cost
// 2 x L x 48 floats per example) but the contractions only walk the rows/cols where a tap's
// weight is >= exp(-30) of its peak.
#include "ra_common.h"

namespace ra {
namespace attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kBandLog = 30.0f;  // weights below exp(-30) * peak are outside the band

// Reference expression order, modellib.py:598-611.
__device__ inline float tap_mu(float ctr, float size, int F, int j) {
  return ctr + (size + 1.0f) / (float)F * ((float)j - ((float)F - 1.0f) / 2.0f);
}
__device__ inline float gauss(float l, float mu, float var) {
  const float d = l - mu;
  return (1.0f / sqrtf(var) / sqrtf(2.0f * 3.14159265358979323846f)) * expf(-0.5f * d * d / var);
}

__global__ void gaussian_filter_kernel(const float *center, const float *size, const float *lg_var,
                                       int B, int L, int F, float *out) {
  const size_t n = (size_t)B * L * F;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const int f = e % F;
    const int l = (e / F) % L;
    const int b = e / ((size_t)F * L);
    out[e] = gauss((float)l, tap_mu(center[b], size[b], F, f), expf(lg_var[b]));
  }
}

// band layout per example (ints): [ylo F][yhi F][xlo F][xhi F][y_jlo H][y_jhi H][x_jlo W][x_jhi W]
__host__ __device__ inline size_t band_ints(int H, int W, int Fh, int Fw) {
  return 2 * (size_t)(Fh + Fw) + 2 * (size_t)(H + W);
}

constexpr int kFiltRows = 64;  // pixels of one axis per workgroup

// One workgroup = kFiltRows pixels of one axis of one example: their dense filter rows, the
// tap range covering each of them, and (chunk 0 only) the per-tap pixel band.
__global__ __launch_bounds__(256) void attn_filters_kernel(const float *attn, int H, int W, int Fh,
                                                            int Fw, float *fy, float *fx, int *band) {
  extern __shared__ int s_band[];  // [F lo][F hi]
  const int b = blockIdx.z, axis = blockIdx.y;
  const int L = axis ? W : H, F = axis ? Fw : Fh;
  const int l0 = blockIdx.x * kFiltRows;
  if (l0 >= L) return;
  const float *rec = attn + (size_t)b * RA_ATTN_STRIDE;
  const float ctr = rec[0 + axis], size = rec[2 + axis], var = expf(rec[4 + axis]);
  int *bd = band + (size_t)b * band_ints(H, W, Fh, Fw);
  int *lo = axis ? bd + 2 * Fh : bd;
  int *hi = lo + F;
  int *jlo = bd + 2 * (Fh + Fw) + (axis ? 2 * H : 0);
  int *jhi = jlo + L;
  float *tab = axis ? fx + (size_t)b * W * Fw : fy + (size_t)b * H * Fh;
  int *s_lo = s_band, *s_hi = s_band + F;
  const float R = sqrtf(2.0f * kBandLog * var);
  for (int j = threadIdx.x; j < F; j += blockDim.x) {
    const float mu = tap_mu(ctr, size, F, j);
    float a = ceilf(mu - R), c = floorf(mu + R) + 1.0f;
    a = fminf(fmaxf(a, 0.0f), (float)L);
    c = fminf(fmaxf(c, 0.0f), (float)L);
    if (!(a == a) || !(c == c)) {  // NaN parameters: keep the whole axis
      a = 0.0f;
      c = (float)L;
    }
    const int ia = (int)a, ic = (int)c;
    s_lo[j] = ia;
    s_hi[j] = ic > ia ? ic : ia;
    if (blockIdx.x == 0) {
      lo[j] = s_lo[j];
      hi[j] = s_hi[j];
    }
  }
  const int nl = (L - l0) < kFiltRows ? (L - l0) : kFiltRows;
  for (int e = threadIdx.x; e < nl * F; e += blockDim.x) {
    const int j = e % F, l = l0 + e / F;
    tab[(size_t)l * F + j] = gauss((float)l, tap_mu(ctr, size, F, j), var);
  }
  __syncthreads();
  // taps covering pixel l: lo/hi are non-decreasing in j, so the set is one contiguous range
  for (int l = l0 + threadIdx.x; l < l0 + nl; l += blockDim.x) {
    int a = 0;
    while (a < F && s_hi[a] <= l) ++a;
    int c = a;
    while (c < F && s_lo[c] <= l) ++c;
    jlo[l] = a;
    jhi[l] = c;
  }
}

// ---- read: patch[b,j,i,c] = gamma * sum_l sum_w fy[l,j] img[l,w,c] fx[w,i] -------------------
constexpr int kJB = 4;     // filter rows (j) per workgroup
constexpr int kWC = 256;   // image columns per pass (= threads)

__global__ __launch_bounds__(256) void extract_patch_kernel(const float *img, int Ci, int chan0,
                                                             const float *attn, const float *fy,
                                                             const float *fx, const int *band, int H,
                                                             int W, int Fh, int Fw, int Cp,
                                                             int use_gamma, float *patch) {
  __shared__ f32x4 tl[kJB][kWC];
  const int t = threadIdx.x;
  const int j0 = blockIdx.x * kJB;
  const int cg = blockIdx.y;
  const int b = blockIdx.z;
  const int nj = (Fh - j0) < kJB ? (Fh - j0) : kJB;
  const int *bd = band + (size_t)b * band_ints(H, W, Fh, Fw);
  const int *ylo = bd, *yhi = bd + Fh, *xlo = bd + 2 * Fh, *xhi = bd + 2 * Fh + Fw;
  const int l0 = ylo[j0], l1 = yhi[j0 + nj - 1];
  const int w0 = xlo[0], w1 = xhi[Fw - 1];
  const float *fyb = fy + (size_t)b * H * Fh;
  const float *fxb = fx + (size_t)b * W * Fw;
  const float *imb = img + (size_t)b * H * W * Ci + chan0 + 4 * cg;

  // each thread owns up to 2 of the nj*Fw outputs
  const int nout = nj * Fw;
  f32x4 P[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};

  for (int wc = w0; wc < w1; wc += kWC) {
    const int w = wc + t;
    f32x4 T[kJB];
#pragma unroll
    for (int jj = 0; jj < kJB; ++jj) T[jj] = f32x4{0, 0, 0, 0};
    if (w < w1) {
      constexpr int U = 8;  // rows in flight per thread: the loop is latency-, not ALU-bound
      for (int l = l0; l < l1; l += U) {
        f32x4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ll = (l + u < l1) ? l + u : l1 - 1;
          xv[u] = *reinterpret_cast<const f32x4 *>(imb + ((size_t)ll * W + w) * Ci);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (l + u < l1) {
            const float *wrow = fyb + (size_t)(l + u) * Fh + j0;
#pragma unroll
            for (int jj = 0; jj < kJB; ++jj) {
              const float wt = (jj < nj) ? wrow[jj] : 0.0f;
              T[jj] += wt * xv[u];
            }
          }
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < kJB; ++jj) tl[jj][t] = T[jj];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int o = t + k * 256;
      if (o < nout) {
        const int jj = o / Fw, i = o % Fw;
        int a = xlo[i] > wc ? xlo[i] : wc;
        int c = xhi[i] < wc + kWC ? xhi[i] : wc + kWC;
        c = c < w1 ? c : w1;
        f32x4 s = P[k];
        for (int ww = a; ww < c; ++ww) s += fxb[(size_t)ww * Fw + i] * tl[jj][ww - wc];
        P[k] = s;
      }
    }
    __syncthreads();
  }
  const float gamma = use_gamma ? attn[(size_t)b * RA_ATTN_STRIDE + 6] : 1.0f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int o = t + k * 256;
    if (o < nout) {
      const int jj = o / Fw, i = o % Fw;
      *reinterpret_cast<f32x4 *>(patch + (((size_t)b * Fh + j0 + jj) * Fw + i) * Cp + 4 * cg) =
          gamma * P[k];
    }
  }
}

// ---- write, stage 1: U[b,j,w] = sum_i P[b,j,i] fx[b,w,i] ---------------------------------------
__global__ __launch_bounds__(256) void paste_u_kernel(const float *patch, int Cp, int pc,
                                                       const float *fx, const int *band, int H, int W,
                                                       int Fh, int Fw, float *u) {
  extern __shared__ float sp[];  // [kJB][Fw]
  const int t = threadIdx.x;
  const int w = blockIdx.x * 256 + t;
  const int j0 = blockIdx.y * kJB;
  const int b = blockIdx.z;
  const int nj = (Fh - j0) < kJB ? (Fh - j0) : kJB;
  for (int e = t; e < kJB * Fw; e += 256) {
    const int jj = e / Fw, i = e % Fw;
    sp[e] = (jj < nj) ? patch[(((size_t)b * Fh + j0 + jj) * Fw + i) * Cp + pc] : 0.0f;
  }
  __syncthreads();
  if (w >= W) return;
  const int *bd = band + (size_t)b * band_ints(H, W, Fh, Fw);
  const int *x_jlo = bd + 2 * (Fh + Fw) + 2 * H, *x_jhi = x_jlo + W;
  const int a = x_jlo[w], c = x_jhi[w];
  const float *fxw = fx + ((size_t)b * W + w) * Fw;
  float acc[kJB];
#pragma unroll
  for (int jj = 0; jj < kJB; ++jj) acc[jj] = 0.0f;
  for (int i = a; i < c; ++i) {
    const float f = fxw[i];
#pragma unroll
    for (int jj = 0; jj < kJB; ++jj) acc[jj] += sp[jj * Fw + i] * f;
  }
#pragma unroll
  for (int jj = 0; jj < kJB; ++jj)
    if (jj < nj) u[((size_t)b * Fh + j0 + jj) * W + w] = acc[jj];
}

__device__ inline float sigmoidf(float z) { return 1.0f / (1.0f + expf(-z)); }

// ---- write, stage 2: y = sigmoid(gamma_y * sum_j fy[l,j] U[j,w] + beta) [* (1-canvas)] ---------
__global__ __launch_bounds__(128) void paste_kernel(const float *u, const float *attn, const float *fy,
                                                     const int *band, int H, int W, int Fh, int Fw,
                                                     float beta, int disable_overwrite, float *img,
                                                     int Ci, int canvas_chan, float *y_out,
                                                     size_t y_stride_b) {
  const int l = blockIdx.x;
  const int b = blockIdx.y;
  const int *bd = band + (size_t)b * band_ints(H, W, Fh, Fw);
  const int *y_jlo = bd + 2 * (Fh + Fw), *y_jhi = y_jlo + H;
  const int a = y_jlo[l], c = y_jhi[l];
  const float gy = expf(attn[(size_t)b * RA_ATTN_STRIDE + 8]);
  const float *fyl = fy + ((size_t)b * H + l) * Fh;
  const float *ub = u + (size_t)b * Fh * W;
  float *yrow = y_out + (size_t)b * y_stride_b + (size_t)l * W;
  float *crow = (canvas_chan >= 0) ? img + ((size_t)b * H + l) * W * Ci + canvas_chan : nullptr;
  const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(yrow) & 15) == 0);
  if (vec && crow && Ci == 4) {
    // 16-byte pixel records: read-modify-write the whole record, 4 pixels per thread
    float *prow = img + ((size_t)b * H + l) * W * 4;
    for (int w4 = threadIdx.x * 4; w4 < W; w4 += blockDim.x * 4) {
      f32x4 px[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) px[k] = *reinterpret_cast<const f32x4 *>(prow + (size_t)(w4 + k) * 4);
      f32x4 s = f32x4{0, 0, 0, 0};
      for (int j = a; j < c; ++j)
        s += fyl[j] * *reinterpret_cast<const f32x4 *>(ub + (size_t)j * W + w4);
      f32x4 y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = sigmoidf(gy * s[k] + beta);
        // select, not a runtime vector index (that would spill the record to scratch)
        const float cv = canvas_chan == 0 ? px[k].x : canvas_chan == 1 ? px[k].y
                       : canvas_chan == 2 ? px[k].z : px[k].w;
        if (disable_overwrite) v *= (1.0f - cv);
        const float nv = fmaxf(cv, v);
        px[k].x = canvas_chan == 0 ? nv : px[k].x;
        px[k].y = canvas_chan == 1 ? nv : px[k].y;
        px[k].z = canvas_chan == 2 ? nv : px[k].z;
        px[k].w = canvas_chan == 3 ? nv : px[k].w;
        y[k] = v;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4 *>(prow + (size_t)(w4 + k) * 4) = px[k];
      *reinterpret_cast<f32x4 *>(yrow + w4) = y;
    }
  } else if (vec) {
    for (int w4 = threadIdx.x * 4; w4 < W; w4 += blockDim.x * 4) {
      f32x4 s = f32x4{0, 0, 0, 0};
      for (int j = a; j < c; ++j)
        s += fyl[j] * *reinterpret_cast<const f32x4 *>(ub + (size_t)j * W + w4);
      f32x4 y;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = sigmoidf(gy * s[k] + beta);
        if (crow) {
          const float cv = crow[(size_t)(w4 + k) * Ci];
          if (disable_overwrite) v *= (1.0f - cv);
          crow[(size_t)(w4 + k) * Ci] = fmaxf(cv, v);
        }
        y[k] = v;
      }
      *reinterpret_cast<f32x4 *>(yrow + w4) = y;
    }
  } else {
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
      float s = 0.0f;
      for (int j = a; j < c; ++j) s += fyl[j] * ub[(size_t)j * W + w];
      float v = sigmoidf(gy * s + beta);
      if (crow) {
        const float cv = crow[(size_t)w * Ci];
        if (disable_overwrite) v *= (1.0f - cv);
        crow[(size_t)w * Ci] = fmaxf(cv, v);
      }
      yrow[w] = v;
    }
  }
}

// ---- attention box: sigmoid(box_gamma * rowsum(fy)[l] * rowsum(fx)[w] + beta) -------------------
__global__ __launch_bounds__(128) void attn_box_kernel(const float *attn, const float *fy,
                                                        const float *fx, const int *band, int H, int W,
                                                        int Fh, int Fw, float beta, float *out,
                                                        size_t stride_b) {
  const int l = blockIdx.x;
  const int b = blockIdx.y;
  const int *bd = band + (size_t)b * band_ints(H, W, Fh, Fw);
  const int *y_jlo = bd + 2 * (Fh + Fw), *y_jhi = y_jlo + H;
  const int *x_jlo = y_jhi + H, *x_jhi = x_jlo + W;
  const float g = attn[(size_t)b * RA_ATTN_STRIDE + 7];
  const float *fyl = fy + ((size_t)b * H + l) * Fh;
  float sy = 0.0f;
  for (int j = y_jlo[l]; j < y_jhi[l]; ++j) sy += fyl[j];
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    const float *fxw = fx + ((size_t)b * W + w) * Fw;
    float sx = 0.0f;
    for (int i = x_jlo[w]; i < x_jhi[w]; ++i) sx += fxw[i];
    out[(size_t)b * stride_b + (size_t)l * W + w] = sigmoidf(g * sy * sx + beta);
  }
}

// ---- generic dense extract_patch with caller filters (operator surface) ------------------------
constexpr int kDenseChunk = 4096;
__global__ __launch_bounds__(256) void extract_dense_kernel(const float *x, const float *f_y,
                                                             const float *f_x, int H, int W, int D,
                                                             int FH, int FW, float *out) {
  __shared__ float tbuf[kDenseChunk];
  const int fh = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const float *xb = x + (size_t)b * H * W * D;
  const float *fyb = f_y + (size_t)b * H * FH;
  const float *fxb = f_x + (size_t)b * W * FW;
  const int WD = W * D;
  const int nout = FW * D;  // outputs (fw, d) of this fh; loop if > 256
  for (int o0 = 0; o0 < nout; o0 += 256) {
    const int o = o0 + t;
    const int fw = o / D, d = o % D;
    float acc = 0.0f;
    for (int c0 = 0; c0 < WD; c0 += kDenseChunk) {
      const int cn = (WD - c0) < kDenseChunk ? (WD - c0) : kDenseChunk;
      __syncthreads();
      for (int e = t; e < cn; e += 256) {
        float s = 0.0f;
        for (int h = 0; h < H; ++h) s += fyb[(size_t)h * FH + fh] * xb[(size_t)h * WD + c0 + e];
        tbuf[e] = s;
      }
      __syncthreads();
      if (o < nout) {
        // elements (w, d) with index w*D + d inside [c0, c0+cn)
        int wlo = (c0 - d + D - 1) / D;
        if (wlo < 0) wlo = 0;
        for (int w = wlo; w < W; ++w) {
          const int idx = w * D + d - c0;
          if (idx >= cn) break;
          acc += tbuf[idx] * fxb[(size_t)w * FW + fw];
        }
      }
    }
    if (o < nout) out[(((size_t)b * FH + fh) * FW + fw) * D + d] = acc;
  }
}

// ---- elementwise helpers --------------------------------------------------------------------------
__global__ void pack_input_kernel(const float *x, int D, const float *d_in, int Dd, const float *y_in,
                                  int Dy, size_t npix, int Cp, float *packed, float *plane) {
  for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix;
       p += (size_t)gridDim.x * blockDim.x) {
    if (plane) plane[p] = 0.0f;  // the canvas plane of the decode loop starts at zero too (full_model.py:239)
    float *o = packed + p * Cp;
    int c = 0;
    for (int k = 0; k < D; ++k) o[c++] = x[p * D + k];
    o[c++] = 0.0f;  // canvas (full_model.py:239)
    for (int k = 0; k < Dd; ++k) o[c++] = d_in[p * Dd + k];
    for (int k = 0; k < Dy; ++k) o[c++] = y_in[p * Dy + k];
    for (; c < Cp; ++c) o[c] = 0.0f;
  }
}

__global__ void canvas_max_kernel(float *img, int Ci, int canvas_chan, const float *ysel,
                                  const float *noise, size_t npix) {
  for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix;
       p += (size_t)gridDim.x * blockDim.x) {
    float v = ysel[p];
    if (noise) v = v - v * noise[p];
    float *c = img + p * Ci + canvas_chan;
    *c = fmaxf(v, *c);
  }
}

__global__ void affine_act_kernel(const float *x, const float *scale, const float *shift, size_t n,
                                  int C, int relu, float *y) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const int c = e % C;
    float v = x[e] * scale[c] + shift[c];
    if (relu) v = fmaxf(v, 0.0f);
    y[e] = v;
  }
}

__global__ void max_pool_kernel(const float *x, int B, int H, int W, int C, int r, int Ho, int Wo,
                                float *y) {
  const size_t n = (size_t)B * Ho * Wo * C;
  const int pt = (Ho * r - H) / 2, pl = (Wo * r - W) / 2;  // TF 'SAME': extra padding at the end
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const int c = e % C;
    const int ox = (e / C) % Wo;
    const int oy = (e / ((size_t)C * Wo)) % Ho;
    const int b = e / ((size_t)C * Wo * Ho);
    float m = -INFINITY;
    for (int dy = 0; dy < r; ++dy)
      for (int dx = 0; dx < r; ++dx) {
        const int yy = oy * r - pt + dy, xx = ox * r - pl + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
          m = fmaxf(m, x[(((size_t)b * H + yy) * W + xx) * C + c]);
      }
    y[e] = m;
  }
}

inline int grid_for(size_t n, int block) {
  size_t g = (n + block - 1) / block;
  return (int)(g < 2048 ? (g ? g : 1) : 2048);
}

}  // namespace attn
}  // namespace ra

using namespace ra;

extern "C" int ra_gaussian_filter_f32(const float *center, const float *size, const float *lg_var,
                                      int B, int L, int F, float *out, void *stream) {
  if (!center || !size || !lg_var || !out || B <= 0 || L <= 0 || F <= 0)
    return fail(RA_E_INVALID, "ra_gaussian_filter_f32: bad argument");
  hipLaunchKernelGGL(attn::gaussian_filter_kernel, dim3(attn::grid_for((size_t)B * L * F, 256)),
                     dim3(256), 0, as_stream(stream), center, size, lg_var, B, L, F, out);
  return launch_status("ra_gaussian_filter_f32");
}

extern "C" size_t ra_attn_band_ints(int H, int W, int Fh, int Fw) {
  return attn::band_ints(H, W, Fh, Fw);
}

extern "C" int ra_attn_filters_f32(const float *attn_rec, int B, int H, int W, int Fh, int Fw,
                                   float *fy, float *fx, int *band, void *stream) {
  if (!attn_rec || !fy || !fx || !band || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0)
    return fail(RA_E_INVALID, "ra_attn_filters_f32: bad argument");
  const int Fm = Fh > Fw ? Fh : Fw, Lm = H > W ? H : W;
  hipLaunchKernelGGL(attn::attn_filters_kernel, dim3(ceil_div(Lm, attn::kFiltRows), 2, B), dim3(256),
                     2 * Fm * sizeof(int), as_stream(stream), attn_rec, H, W, Fh, Fw, fy, fx, band);
  return launch_status("ra_attn_filters_f32");
}

extern "C" int ra_extract_patch_f32(const float *img, int Ci, int chan0, const float *attn_rec,
                                    const float *fy, const float *fx, const int *band, int B, int H,
                                    int W, int Fh, int Fw, int Cp, int use_gamma, float *patch,
                                    void *stream) {
  if (!img || !attn_rec || !fy || !fx || !band || !patch || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 ||
      Fw <= 0)
    return fail(RA_E_INVALID, "ra_extract_patch_f32: bad argument");
  if (Ci % 4 || Cp % 4 || chan0 % 4 || chan0 + Cp > Ci || Cp <= 0)
    return fail(RA_E_SHAPE, "ra_extract_patch_f32: Ci=%d chan0=%d Cp=%d", Ci, chan0, Cp);
  if (attn::kJB * Fw > 512) return fail(RA_E_SHAPE, "ra_extract_patch_f32: Fw %d > 64", Fw);
  dim3 grid(ceil_div(Fh, attn::kJB), Cp / 4, B);
  hipLaunchKernelGGL(attn::extract_patch_kernel, grid, dim3(256), 0, as_stream(stream), img, Ci,
                     chan0, attn_rec, fy, fx, band, H, W, Fh, Fw, Cp, use_gamma, patch);
  return launch_status("ra_extract_patch_f32");
}

extern "C" int ra_paste_canvas_f32(const float *patch, int Cp, int pc, const float *attn_rec,
                                   const float *fy, const float *fx, const int *band, int B, int H,
                                   int W, int Fh, int Fw, float beta, int disable_overwrite,
                                   float *img, int Ci, int canvas_chan, float *y_out,
                                   size_t y_stride_b, float *u_ws, void *stream) {
  if (!patch || !attn_rec || !fy || !fx || !band || !y_out || !u_ws || B <= 0 || H <= 0 || W <= 0 ||
      Fh <= 0 || Fw <= 0 || Cp <= 0 || pc < 0 || pc >= Cp)
    return fail(RA_E_INVALID, "ra_paste_canvas_f32: bad argument");
  if (canvas_chan >= 0 && (!img || canvas_chan >= Ci))
    return fail(RA_E_INVALID, "ra_paste_canvas_f32: canvas channel %d of %d", canvas_chan, Ci);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(attn::paste_u_kernel, dim3(ceil_div(W, 256), ceil_div(Fh, attn::kJB), B),
                     dim3(256), attn::kJB * Fw * sizeof(float), st, patch, Cp, pc, fx, band, H, W, Fh,
                     Fw, u_ws);
  int rc = launch_status("ra_paste_canvas_f32(u)");
  if (rc) return rc;
  hipLaunchKernelGGL(attn::paste_kernel, dim3(H, B), dim3(128), 0, st, u_ws, attn_rec, fy, band, H, W,
                     Fh, Fw, beta, disable_overwrite, img, Ci, canvas_chan, y_out, y_stride_b);
  return launch_status("ra_paste_canvas_f32");
}

extern "C" int ra_attn_box_f32(const float *attn_rec, const float *fy, const float *fx,
                               const int *band, int B, int H, int W, int Fh, int Fw, float beta,
                               float *box_out, size_t stride_b, void *stream) {
  if (!attn_rec || !fy || !fx || !band || !box_out || B <= 0 || H <= 0 || W <= 0)
    return fail(RA_E_INVALID, "ra_attn_box_f32: bad argument");
  hipLaunchKernelGGL(attn::attn_box_kernel, dim3(H, B), dim3(128), 0, as_stream(stream), attn_rec, fy,
                     fx, band, H, W, Fh, Fw, beta, box_out, stride_b);
  return launch_status("ra_attn_box_f32");
}

extern "C" int ra_extract_patch_dense_f32(const float *x, const float *f_y, const float *f_x, int B,
                                          int H, int W, int D, int FH, int FW, float *out,
                                          void *stream) {
  if (!x || !f_y || !f_x || !out || B <= 0 || H <= 0 || W <= 0 || D <= 0 || FH <= 0 || FW <= 0)
    return fail(RA_E_INVALID, "ra_extract_patch_dense_f32: bad argument");
  hipLaunchKernelGGL(attn::extract_dense_kernel, dim3(FH, B), dim3(256), 0, as_stream(stream), x, f_y,
                     f_x, H, W, D, FH, FW, out);
  return launch_status("ra_extract_patch_dense_f32");
}

extern "C" int ra_pack_input_f32(const float *x, int D, const float *d_in, int Dd, const float *y_in,
                                 int Dy, int B, int H, int W, int Cp, float *packed, void *stream) {
  return ra_pack_input_plane_f32(x, D, d_in, Dd, y_in, Dy, B, H, W, Cp, packed, nullptr, stream);
}

extern "C" int ra_pack_input_plane_f32(const float *x, int D, const float *d_in, int Dd, const float *y_in,
                                       int Dy, int B, int H, int W, int Cp, float *packed, float *canvas_plane,
                                       void *stream) {
  if (!x || !packed || B <= 0 || H <= 0 || W <= 0 || D <= 0 || (Dd > 0 && !d_in) || (Dy > 0 && !y_in))
    return fail(RA_E_INVALID, "ra_pack_input_f32: bad argument");
  if (Cp % 4 || D + 1 + Dd + Dy > Cp) return fail(RA_E_SHAPE, "ra_pack_input_f32: Cp %d", Cp);
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(attn::pack_input_kernel, dim3(attn::grid_for(npix, 256)), dim3(256), 0,
                     as_stream(stream), x, D, d_in, Dd, y_in, Dy, npix, Cp, packed, canvas_plane);
  return launch_status("ra_pack_input_f32");
}

extern "C" int ra_canvas_max_f32(float *img, int Ci, int canvas_chan, const float *ysel,
                                 const float *noise, int B, int H, int W, void *stream) {
  if (!img || !ysel || B <= 0 || H <= 0 || W <= 0 || canvas_chan < 0 || canvas_chan >= Ci)
    return fail(RA_E_INVALID, "ra_canvas_max_f32: bad argument");
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(attn::canvas_max_kernel, dim3(attn::grid_for(npix, 256)), dim3(256), 0,
                     as_stream(stream), img, Ci, canvas_chan, ysel, noise, npix);
  return launch_status("ra_canvas_max_f32");
}

extern "C" int ra_affine_act_f32(const float *x, const float *scale, const float *shift, size_t npix,
                                 int C, int relu, float *y, void *stream) {
  if (!x || !scale || !shift || !y || C <= 0) return fail(RA_E_INVALID, "ra_affine_act_f32: bad argument");
  if (npix == 0) return 0;
  const size_t n = npix * C;
  hipLaunchKernelGGL(attn::affine_act_kernel, dim3(attn::grid_for(n, 256)), dim3(256), 0,
                     as_stream(stream), x, scale, shift, n, C, relu, y);
  return launch_status("ra_affine_act_f32");
}

extern "C" int ra_max_pool_f32(const float *x, int B, int H, int W, int C, int ratio, float *y,
                               void *stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ratio <= 0)
    return fail(RA_E_INVALID, "ra_max_pool_f32: bad argument");
  const int Ho = ceil_div(H, ratio), Wo = ceil_div(W, ratio);
  hipLaunchKernelGGL(attn::max_pool_kernel, dim3(attn::grid_for((size_t)B * Ho * Wo * C, 256)),
                     dim3(256), 0, as_stream(stream), x, B, H, W, C, ratio, Ho, Wo, y);
  return launch_status("ra_max_pool_f32");
}

