// The literal (dense) operators of the reference's attention and a few element-wise helpers:
//   modellib.get_gaussian_filter  modellib.py:581-612   -> ra_gaussian_filter_f32 (the [L,F] bank itself)
//   modellib.extract_patch        modellib.py:615-641   -> ra_extract_patch_dense_f32 (F_y^T X F_x for GIVEN banks)
//   input packing, canvas max (box_model), affine + activation, max pool.
// The decode loop and the training step never materialise the banks: their banded kernels (ra_attn_direct.hip,
// ra_attn_train.hip) evaluate the weights on the fly from the attention record.
#include "ra_common.h"

namespace ra {
namespace attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// One thread per (pixel, channel quad) of the packed image: 16-byte stores, consecutive threads -> consecutive addresses (round 6;
// the one-thread-per-pixel form wrote Cp scalars 4 Cp bytes apart per lane and ran at 0.9 TB/s: 413 us per forward at cfg5's 16
// images x 256 x 512 x 24 channels).  The quad's four channels come from x | 0 (canvas slot, full_model.py:239) | d_in | y_in | 0-pad.
__global__ void pack_input_kernel(const float *x, int D, const float *d_in, int Dd, const float *y_in,
                                  int Dy, size_t npix, int Cp, float *packed, float *plane) {
  const int C4 = Cp >> 2;
  const size_t n = npix * C4;
  typedef float f32x4p __attribute__((ext_vector_type(4)));
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const size_t p = e / C4;
    const int q = (int)(e - p * C4);
    if (plane && q == 0) plane[p] = 0.0f;  // the canvas plane of the decode loop starts at zero too (full_model.py:239)
    f32x4p v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * q + j;
      float val = 0.0f;
      if (c < D) val = x[p * D + c];
      else if (c > D && c <= D + Dd) val = d_in[p * Dd + (c - D - 1)];
      else if (c > D + Dd && c <= D + Dd + Dy) val = y_in[p * Dy + (c - D - 1 - Dd)];
      v[j] = val;
    }
    reinterpret_cast<f32x4p *>(packed)[e] = v;
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
  if (Cp % 4 || D + 1 + Dd + Dy > Cp || (reinterpret_cast<uintptr_t>(packed) & 15))
    return fail(RA_E_SHAPE, "ra_pack_input_f32: Cp %d (a multiple of 4 holding D + 1 + Dd + Dy channels), packed 16-byte aligned", Cp);
  const size_t npix = (size_t)B * H * W;
  hipLaunchKernelGGL(attn::pack_input_kernel, dim3(attn::grid_for(npix * (size_t)(Cp / 4), 256)), dim3(256), 0,
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

