// Training-side adjoints of the attention resample (modellib.extract_patch, modellib.py:615-641, and its
// uses full_model.py:738-741 box, :778-789 read, :810-818 write) WITHOUT materialising the [L,F] Gaussian
// banks or their gradients: the reference differentiates  fy^T X fx  through two dense [L,F] matrices per
// window; here one banded kernel per use turns the upstream gradient straight into the gradients of the six
// window parameters (centre, size, log-variance per axis), the gamma, and — for the paste — the patch.
//
// One form covers the three uses.  With E(X) = fy^T X fx  (a Fh x Fw "extract" of an image-sized X):
//   read  (x_patch = gamma E(inp)):           X = inp (not differentiated: the canvas gradient is stopped),
//                                             Q = d x_patch;   d gamma = sum E.Q;  d params = gamma * D(X, Q)
//   write (y = sigmoid(g S - 5), S = fy P fx^T):  X = dS = g * dy y (1 - y),  Q = P;
//                                             dP = E(X);  d lg_gamma = sum E.Q;  d params = D(X, Q)
//   box   (the same with P == 1, g = box_gamma): d box_gamma = sum E.Q / g
// where D(X, Q) contracts  d fy[l,j] = sum_{w,i,c} X[l,w,c] fx[w,i] Q[j,i,c]  and
// d fx[w,i] = sum_{l,j,c} fy[l,j] X[l,w,c] Q[j,i,c]  with the filters' derivatives
//   d w / d mu = w (l - mu) / var,  d mu_j / d ctr = 1,  d mu_j / d size = (j - (F-1)/2) / F,
//   d w / d lg_var = w (-1/2 + (l - mu)^2 / (2 var)).
// Workgroup = (tap j, channel group, image), the geometry of extract_rows_kernel: its tap's row band x the
// window's columns, one pass.  Everything is restricted to the band where the weight exceeds e^-30 of the peak.
#include "ra_attn_axis.h"
#include "ra_common.h"

namespace ra {
namespace attnt {

using attnd::Axis;
using attnd::f32x4;
using attnd::make_axis;
using attnd::readlane_f;

constexpr int kKR = 4;

template <int NC>
struct VT;
template <>
struct VT<4> {
  typedef f32x4 type;
  static __device__ inline float dot(const f32x4 &a, const f32x4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
  static __device__ inline f32x4 zero() { return f32x4{0, 0, 0, 0}; }
  static __device__ inline f32x4 splat(float v) { return f32x4{v, v, v, v}; }
};
template <>
struct VT<1> {
  typedef float type;
  static __device__ inline float dot(float a, float b) { return a * b; }
  static __device__ inline float zero() { return 0.0f; }
  static __device__ inline float splat(float v) { return v; }
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct BwdArgs {
  const float *X;   // NC = 4: [B,H,W,Cx], channels chan0 + 4 cg ..  (PASTE: unused)
  int Cx, chan0;
  const float *dY, *Yv;  // PASTE: X = gain * dY * Yv * (1 - Yv), planes [B,H,W]
  int gain_mode;         // PASTE: 0: gain = exp(rec[8]) (mask paste, y_lg_gamma), 1: gain = rec[7] (box_gamma)
  const float *rec;      // [B, RA_ATTN_STRIDE]
  const float *Q;        // [B,Fh,Fw,Cq] (channels 4 cg .. / channel 0) or nullptr = ones
  int Cq;
  float *E;              // optional [B,Fh,Fw,Ce]: fy^T X fx
  int Ce;
  float *part;           // [B][Fh][ncg][8]
  int H, W, Fh, Fw, ncg;
};

template <int NC, bool PASTE>
__global__ __launch_bounds__(256) void resample_bwd_kernel(const BwdArgs a, int n_items, int chunk) {
  typedef typename VT<NC>::type V;
  constexpr int KR = kKR;
  __shared__ V red[4][256];
  __shared__ V Qs[256];
  __shared__ float sred[4][8];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int slot = blockIdx.x >> 3;
  const int item = (blockIdx.x & 7) * chunk + slot;  // XCD-contiguous: the taps of an image share an L2
  if (slot >= chunk || item >= n_items) return;
  const int H = a.H, W = a.W, Fh = a.Fh, Fw = a.Fw, ncg = a.ncg;
  const int b = item / (Fh * ncg), rem = item - b * Fh * ncg;
  const int j = rem / ncg, cg = rem - j * ncg;
  const float *rec = a.rec + (size_t)b * RA_ATTN_STRIDE;
  const Axis Ay = make_axis(rec, 0, H, Fh), Ax = make_axis(rec, 1, W, Fw);
  int l0, l1, w0, w1, tmp;
  Ay.band(j, l0, l1);
  Ax.band(0, w0, tmp);
  Ax.band(Fw - 1, tmp, w1);
  const float gain = PASTE ? (a.gain_mode ? rec[7] : __expf(rec[8])) : 1.0f;
  const float ivy = 2.0f * Ay.inv2var, ivx = 2.0f * Ax.inv2var;  // 1 / var
  const float muj = Ay.mu(j);
  // this tap's row of Q
  for (int i = t; i < Fw; i += 256) {
    V q = VT<NC>::splat(1.0f);
    if (a.Q) {
      const float *qp = a.Q + (((size_t)b * Fh + j) * Fw + i) * a.Cq + (NC == 4 ? 4 * cg : 0);
      if constexpr (NC == 4) q = *reinterpret_cast<const f32x4 *>(qp);
      else q = qp[0];
    }
    Qs[i] = q;
  }
  __syncthreads();
  constexpr int kOOB = 0x7fffffff;
  const size_t img_px = (size_t)H * W;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(PASTE ? a.dY + (size_t)b * img_px : a.X + (size_t)b * img_px * a.Cx + a.chan0 + 4 * cg), 0,
      PASTE ? (int)(img_px * 4) : (int)(img_px * a.Cx * 4 - (size_t)(a.chan0 + 4 * cg) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(PASTE ? a.Yv + (size_t)b * img_px : a.rec), 0, PASTE ? (int)(img_px * 4) : 0, 0x00020000);

  // stage-2 role: output column oi of this tap, its band's columns dealt to `parts` neighbouring lanes
  const int parts = (Fw * 4 <= 256) ? 4 : (Fw * 2 <= 256) ? 2 : 1;
  const int oi = t / parts, part = t - oi * parts;
  const bool owner = oi < Fw;
  int bi_lo = 0, bi_hi = 0;
  if (owner) Ax.band(oi, bi_lo, bi_hi);
  const float mui = Ax.mu(oi);
  V P = VT<NC>::zero();
  float py0 = 0.f, py1 = 0.f, py2 = 0.f, px0 = 0.f, px1 = 0.f, px2 = 0.f;
  const float jrel = ((float)j - Ay.half) / (float)Fh, irel = ((float)oi - Ax.half) / (float)Fw;

  for (int cp = w0; cp < w1; cp += 256) {
    // G[w] = sum_i fx(w, i) Q[j, i] for this thread's four columns
    V G[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int w = cp + 64 * p + lane;
      V g = VT<NC>::zero();
      if (w < w1) {
        int ilo, ihi;
        Ax.taps(w, ilo, ihi);
        for (int i = ilo; i < ihi; ++i) g += Ax.w((float)w, i) * Qs[i];
      }
      G[p] = g;
    }
    V acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = VT<NC>::zero();
    for (int rb = l0; rb < l1; rb += 4 * KR) {
      V xv[KR][4];
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int row = rb + 4 * k + wv;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int col = cp + 64 * p + lane;
          const int pix = (row < l1 && col < w1) ? row * W + col : -1;
          if constexpr (PASTE) {
            const float dy = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, pix >= 0 ? pix * 4 : kOOB, 0, 0));
            const float yv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, pix >= 0 ? pix * 4 : kOOB, 0, 0));
            xv[k][p] = gain * dy * yv * (1.0f - yv);
          } else {
            xv[k][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, pix >= 0 ? pix * a.Cx * 4 : kOOB, 0, 0));
          }
        }
      }
      // the row filter of this wave's KR rows: lane k evaluates row rb + 4k + wv
      const int rowl = rb + 4 * lane + wv;
      const float fyv = (lane < KR && rowl < l1) ? Ay.w((float)rowl, j) : 0.0f;
      float smine = 0.0f;  // lane k: sum_w X[row_k, w] . G[w]
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const float wy = readlane_f(fyv, k);
        float rd = 0.0f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          rd += VT<NC>::dot(xv[k][p], G[p]);
          acc[p] += wy * xv[k][p];
        }
        rd = wave_sum(rd);
        smine = (lane == k) ? rd : smine;
      }
      // d fy[l, j] = smine  ->  the three y-axis parameters
      const float d = (float)rowl - muj, tt = smine * fyv;
      py0 += tt * d * ivy;
      py2 += tt * (-0.5f + 0.5f * d * d * ivy);
    }
    __syncthreads();  // the previous column group's stage 2 is done with `red`
#pragma unroll
    for (int p = 0; p < 4; ++p) red[wv][64 * p + lane] = acc[p];
    __syncthreads();
    red[0][t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);  // T1[w] = sum_l fy[l, j] X[l, w]
    __syncthreads();
    if (owner) {
      const int cend = (cp + 256 < w1) ? cp + 256 : w1;
      const int lo = bi_lo > cp ? bi_lo : cp, hi = bi_hi < cend ? bi_hi : cend;
      const V q = Qs[oi];
      for (int ww = lo + part; ww < hi; ww += parts) {
        const float f = Ax.w((float)ww, oi);
        const V T1 = red[0][ww - cp];
        P += f * T1;
        const float d = (float)ww - mui, tt = VT<NC>::dot(T1, q) * f;  // d fx[w, i] * fx[w, i]
        px0 += tt * d * ivx;
        px2 += tt * (-0.5f + 0.5f * d * d * ivx);
      }
    }
  }
  py1 = py0 * jrel;
  px1 = px0 * irel;
  // E[j, oi] and sum E . Q
  float eq = 0.0f;
  {
    V v = P;
    if constexpr (NC == 4) {
      if (parts >= 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], 1);
      }
      if (parts >= 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += __shfl_xor(v[c], 2);
      }
    } else {
      if (parts >= 2) v += __shfl_xor(v, 1);
      if (parts >= 4) v += __shfl_xor(v, 2);
    }
    if (owner && part == 0) {
      eq = VT<NC>::dot(v, Qs[oi]);
      if (a.E) {
        float *ep = a.E + (((size_t)b * Fh + j) * Fw + oi) * a.Ce + (NC == 4 ? 4 * cg : 0);
        if constexpr (NC == 4) *reinterpret_cast<f32x4 *>(ep) = v;
        else ep[0] = v;
      }
    }
  }
  // the seven sums of this workgroup, in a fixed order
  float vals[7] = {py0, py1, py2, px0, px1, px2, eq};
#pragma unroll
  for (int k = 0; k < 7; ++k) vals[k] = wave_sum(vals[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) sred[wv][k] = vals[k];
  }
  __syncthreads();
  if (t < 8) {
    const float s = t < 7 ? (sred[0][t] + sred[1][t]) + (sred[2][t] + sred[3][t]) : 0.0f;
    a.part[(((size_t)b * Fh + j) * ncg + cg) * 8 + t] = s;
  }
}

// out[b] = (d ctr_y, d ctr_x, d size_y, d size_x, d lg_var_y, d lg_var_x, d gamma-like, 0): the workgroups' partial
// sums added in a fixed order; the six filter parameters scaled by scale[b] (gamma of the read), the seventh divided
// by div[b] (box_gamma of the box)
__global__ __launch_bounds__(64) void resample_bwd_final_kernel(const float *part, int n_per_image, const float *scale,
                                                                int scale_stride, const float *div, int div_stride,
                                                                float *out) {
  const int b = blockIdx.x, t = threadIdx.x;
  __shared__ float s[64];
  const int k = t & 7, g = t >> 3;  // 8 groups of 8 lanes: group g sums records g, g+8, ...
  float v = 0.0f;
  for (int r = g; r < n_per_image; r += 8) v += part[((size_t)b * n_per_image + r) * 8 + k];
  s[t] = v;
  __syncthreads();
  if (t < 8) {
    float tot = 0.0f;
    for (int gg = 0; gg < 8; ++gg) tot += s[gg * 8 + t];
    const float sc = scale ? scale[(size_t)b * scale_stride] : 1.0f;
    // partial order (py0 ctr_y, py1 size_y, py2 lgv_y, px0 ctr_x, px1 size_x, px2 lgv_x, eq) -> output order
    const int dst = (t == 0) ? 0 : (t == 1) ? 2 : (t == 2) ? 4 : (t == 3) ? 1 : (t == 4) ? 3 : (t == 5) ? 5 : t;
    float val = tot;
    if (t < 6) val *= sc;
    else if (t == 6 && div) val /= div[(size_t)b * div_stride];
    out[(size_t)b * 8 + dst] = (t == 7) ? 0.0f : val;
  }
}

}  // namespace attnt
}  // namespace ra

using namespace ra;

extern "C" size_t ra_resample_bwd_workspace_floats(int B, int Fh, int C) {
  if (B <= 0 || Fh <= 0 || C <= 0) return 0;
  return (size_t)B * Fh * ((C + 3) / 4) * 8;
}

extern "C" int ra_resample_bwd_f32(int mode, const float *X, int Cx, int chan0, int C, const float *dY, const float *Y,
                                   const float *attn_rec, const float *Q, int Cq, float *E, int Ce, int B, int H, int W,
                                   int Fh, int Fw, const float *scale, int scale_stride, const float *div, int div_stride,
                                   float *ws, size_t ws_floats, float *out, void *stream) {
  if (!attn_rec || !ws || !out || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0 || Fw > 256)
    return fail(RA_E_INVALID, "ra_resample_bwd_f32: bad argument");
  if (mode < 0 || mode > 2) return fail(RA_E_INVALID, "ra_resample_bwd_f32: mode %d", mode);
  attnt::BwdArgs a{};
  a.rec = attn_rec;
  a.Q = Q;
  a.Cq = Cq;
  a.E = E;
  a.Ce = Ce;
  a.part = ws;
  a.H = H;
  a.W = W;
  a.Fh = Fh;
  a.Fw = Fw;
  hipStream_t st = as_stream(stream);
  if (mode == RA_RESAMPLE_READ) {
    if (!X || C <= 0 || (C & 3) || (Cx & 3) || (chan0 & 3) || chan0 + C > Cx || (Q && (Cq & 3)) || (E && (Ce & 3)))
      return fail(RA_E_SHAPE, "ra_resample_bwd_f32: read: channel counts must be multiples of 4");
    if ((size_t)H * W * Cx * 4 >= 0x7fffffffu) return fail(RA_E_SHAPE, "ra_resample_bwd_f32: one image exceeds 2 GiB");
    a.X = X;
    a.Cx = Cx;
    a.chan0 = chan0;
    a.ncg = C / 4;
  } else {
    if (!dY || !Y) return fail(RA_E_INVALID, "ra_resample_bwd_f32: write / box need dY and Y");
    if ((size_t)H * W * 4 >= 0x7fffffffu) return fail(RA_E_SHAPE, "ra_resample_bwd_f32: one image exceeds 2 GiB");
    a.dY = dY;
    a.Yv = Y;
    a.gain_mode = mode == RA_RESAMPLE_BOX ? 1 : 0;
    a.ncg = 1;
  }
  if (ws_floats < (size_t)B * Fh * a.ncg * 8) return fail(RA_E_WORKSPACE, "ra_resample_bwd_f32: workspace too small");
  const int n_items = Fh * a.ncg * B, chunk = ceil_div(n_items, 8);
  if (mode == RA_RESAMPLE_READ)
    hipLaunchKernelGGL((attnt::resample_bwd_kernel<4, false>), dim3(8 * chunk), dim3(256), 0, st, a, n_items, chunk);
  else
    hipLaunchKernelGGL((attnt::resample_bwd_kernel<1, true>), dim3(8 * chunk), dim3(256), 0, st, a, n_items, chunk);
  hipLaunchKernelGGL(attnt::resample_bwd_final_kernel, dim3(B), dim3(64), 0, st, ws, Fh * a.ncg, scale, scale_stride, div,
                     div_stride, out);
  return launch_status("ra_resample_bwd_f32");
}
