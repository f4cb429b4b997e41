// The controller of the TRAINING graph (full_model.py:668-689: soft-attention glimpse read-out -> dense LSTM ->
// glimpse MLP with a softmax over the feature map, num_ctrl_rnn_iter times, then the controller MLP) as one
// forward and one backward launch per timestep instead of ~35 + ~70 library GEMMs and element-wise launches.
// One workgroup (16 waves) per image, the geometry of ra_ctrl.hip's inference kernel: the feature map lives in
// LDS, the 1.8 MB of weights stream from L2 (shared by all images).  The forward saves what the backward needs
// per glimpse iteration; the backward runs the recurrence in reverse (BPTT), accumulates d feat in LDS, and
// writes the pre-activation gradients of every dense layer.  Parameter gradients are NOT formed here: they are
// sums over images, iterations and timesteps of (layer input)^T (pre-activation gradient), i.e. ONE GEMM per weight
// matrix per optimisation step over the saved rows (ra_train.ControllerFn: four addmm_ per step instead of 960
// launches).  Any depth of the two MLPs up to kMaxL layers each (full_model.py:350-352,382-384: the glimpse MLP is
// n_g layers [hid] * n_g + [G], ReLU on all but the last, whose softmax is the next glimpse map; the controller MLP is
// n_c layers [hid] + [mlp] * (n_c - 1) + [nout], ReLU on all but the last); every run script uses 2 and 1.
#include "ra_common.h"

namespace ra {
namespace ctrlt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

constexpr int kMaxL = 4;  // layers per MLP
struct Dims {
  int G, Cf, hid, iters, nout;  // nout = 9 controller outputs
  int n_g, n_c, mlp;            // glimpse-MLP layers, controller-MLP layers, controller-MLP hidden width
};

__device__ inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// out[n] (+ bias[n]) = sum_k xs[k] W[k][n], n < N (N % 4 == 0, N <= 1024): thread = (output quad, k part); xs / out / red in LDS
__device__ void gemv_cols(const float *xs, int K, const float *__restrict__ W, int N, const float *__restrict__ bias, float *out,
                          float *red) {
  const int t = threadIdx.x, quads = N >> 2;
  int parts = kThreads / quads;
  if (parts > 16) parts = 16;
  if (t < quads * parts) {
    const int qd = t % quads, part = t / quads;
    f32x4 acc = f32x4{0, 0, 0, 0};
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(W) + qd;
#pragma unroll 8
    for (int k = part; k < K; k += parts) acc += xs[k] * wp[(size_t)k * quads];
    *reinterpret_cast<f32x4 *>(red + (size_t)part * N + 4 * qd) = acc;
  }
  __syncthreads();
  for (int n = t; n < N; n += kThreads) {
    float s = bias ? bias[n] : 0.0f;
    for (int p = 0; p < parts; ++p) s += red[p * N + n];
    out[n] = s;
  }
  __syncthreads();
}

// out[k] (+)= sum_n W[k][n] v[n], k < K: a wave per row (rows are contiguous: 16-byte loads), v in LDS
__device__ void gemv_rows(const float *v, int N, const float *__restrict__ W, int K, float *out, bool accumulate) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int k = wv; k < K; k += kWaves) {
    const f32x4 *wr = reinterpret_cast<const f32x4 *>(W + (size_t)k * N);
    float s = 0.0f;
    for (int q = lane; q < (N >> 2); q += 64) {
      const f32x4 w = wr[q];
      const f32x4 x = *reinterpret_cast<const f32x4 *>(v + 4 * q);
      s += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
    }
    s = wave_sum(s);
    if (lane == 0) out[k] = accumulate ? out[k] + s : s;
  }
  __syncthreads();
}

__host__ __device__ inline int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
__host__ __device__ inline int va_floats(const Dims &d) { return max3(4 * d.hid, 2 * d.hid + d.G, 2 * d.mlp); }

__device__ float block_reduce(float v, bool is_max, float *red) {
  const int t = threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, other) : v + other;
  }
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < kWaves; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  __syncthreads();
  return r;
}

// Saved per (image, iteration), floats:  xh [Cf + hid] | act [4 hid] (i, f, o, u after their nonlinearities) |
// c [hid] | z [(n_g - 1) hid] (the glimpse MLP's hidden layers after their ReLUs) | gm [G] (the map this iteration READ with).
// Per image (save_c): the controller MLP's hidden layers after their ReLUs [(n_c - 1) mlp].
__host__ __device__ inline int save_floats(const Dims &d) { return (d.Cf + d.hid) + 4 * d.hid + d.hid + (d.n_g - 1) * d.hid + d.G; }

struct FwdArgs {
  Dims d;
  const float *feat;                       // [B, G, Cf]
  const float *Wg, *bg;                    // [Cf + hid, 4 hid] gate order i f o u, [4 hid]
  const float *gW[kMaxL], *gb[kMaxL];      // glimpse MLP: [hid, hid] ... [hid, G]
  const float *cW[kMaxL], *cb[kMaxL];      // controller MLP: [hid, mlp] [mlp, mlp] ... [., nout]
  float *h_last, *co;                      // [B, hid], [B, nout]
  float *save;                             // [B, iters, save_floats]
  float *save_c;                           // [B, (n_c - 1) mlp] (null for n_c == 1)
};

__global__ __launch_bounds__(kThreads) void ctrl_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims d = a.d;
  const int t = threadIdx.x, b = blockIdx.x, G = d.G, Cf = d.Cf, hid = d.hid;
  float *red = smem;                         // 16 * 4 hid
  float *xh = red + 16 * 4 * hid;            // [Cf + hid]
  float *cst = xh + Cf + hid;                // [hid]
  float *va = cst + hid;                     // [va_floats]: the gates; the MLPs' ping-pong buffers and logits
  float *gm = va + va_floats(d);             // [G]
  float *fl = gm + G;                        // [G * Cf]
  const float *fsrc = a.feat + (size_t)b * G * Cf;
  for (int e = t * 4; e < G * Cf; e += kThreads * 4) *reinterpret_cast<f32x4 *>(fl + e) = *reinterpret_cast<const f32x4 *>(fsrc + e);
  for (int e = t; e < hid; e += kThreads) {
    xh[Cf + e] = 0.0f;
    cst[e] = 0.0f;
  }
  for (int g = t; g < G; g += kThreads) gm[g] = 1.0f / (float)G;
  __syncthreads();
  const int SF = save_floats(d);
  for (int it = 0; it < d.iters; ++it) {
    float *sv = a.save + ((size_t)b * d.iters + it) * SF;
    float *sv_act = sv + (Cf + hid), *sv_c = sv_act + 4 * hid, *sv_z = sv_c + hid, *sv_gm = sv_z + (d.n_g - 1) * hid;
    for (int g = t; g < G; g += kThreads) sv_gm[g] = gm[g];
    {  // glimpse[c] = sum_g feat[g, c] map[g]
      const int parts = kThreads / Cf, c = t % Cf, part = t / Cf;
      if (part < parts) {
        float s = 0.0f;
        for (int g = part; g < G; g += parts) s += fl[(size_t)g * Cf + c] * gm[g];
        red[part * Cf + c] = s;
      }
      __syncthreads();
      if (t < Cf) {
        float s = 0.0f;
        for (int p = 0; p < parts; ++p) s += red[p * Cf + t];
        xh[t] = s;
      }
      __syncthreads();
    }
    for (int e = t; e < Cf + hid; e += kThreads) sv[e] = xh[e];
    gemv_cols(xh, Cf + hid, a.Wg, 4 * hid, a.bg, va, red);
    if (t < hid) {
      const float gi = sigm(va[t]), gf = sigm(va[hid + t]), go = sigm(va[2 * hid + t]), u = tanhf(va[3 * hid + t]);
      const float c = gf * cst[t] + gi * u;
      cst[t] = c;
      xh[Cf + t] = go * tanhf(c);
      sv_act[t] = gi;
      sv_act[hid + t] = gf;
      sv_act[2 * hid + t] = go;
      sv_act[3 * hid + t] = u;
      sv_c[t] = c;
    }
    __syncthreads();
    if (it < d.iters - 1) {
      // glimpse MLP: hidden layers ping-pong between va[0, hid) and va[hid, 2 hid); the logits go to va + 2 hid
      const float *vin = xh + Cf;
      for (int l = 0; l + 1 < d.n_g; ++l) {
        float *vo = va + (l & 1) * hid;
        gemv_cols(vin, hid, a.gW[l], hid, a.gb[l], vo, red);
        for (int n = t; n < hid; n += kThreads) {
          const float z = fmaxf(vo[n], 0.0f);
          vo[n] = z;
          sv_z[l * hid + n] = z;
        }
        __syncthreads();
        vin = vo;
      }
      float *lg = va + 2 * hid;
      gemv_cols(vin, hid, a.gW[d.n_g - 1], G, a.gb[d.n_g - 1], lg, red);
      float mx = -3.0e38f;
      for (int n = t; n < G; n += kThreads) mx = fmaxf(mx, lg[n]);
      mx = block_reduce(mx, true, red);
      float sum = 0.0f;
      for (int n = t; n < G; n += kThreads) {
        const float e = expf(lg[n] - mx);
        lg[n] = e;
        sum += e;
      }
      sum = block_reduce(sum, false, red);
      for (int n = t; n < G; n += kThreads) gm[n] = lg[n] / sum;
      __syncthreads();
    } else {
      for (int n = t; n < (d.n_g - 1) * hid; n += kThreads) sv_z[n] = 0.0f;
    }
  }
  // controller MLP: hidden layers (ReLU) through gemv_cols, then co = v Wc + bc
  if (t < hid) a.h_last[(size_t)b * hid + t] = xh[Cf + t];
  const float *vin = xh + Cf;
  int K = hid;
  for (int l = 0; l + 1 < d.n_c; ++l) {
    float *vo = va + (l & 1) * d.mlp;
    gemv_cols(vin, K, a.cW[l], d.mlp, a.cb[l], vo, red);
    for (int n = t; n < d.mlp; n += kThreads) {
      const float z = fmaxf(vo[n], 0.0f);
      vo[n] = z;
      a.save_c[((size_t)b * (d.n_c - 1) + l) * d.mlp + n] = z;
    }
    __syncthreads();
    vin = vo;
    K = d.mlp;
  }
  {  // a wave per output, pairwise reduction (a 256-term serial sum costs half a digit the window position amplifies)
    const int lane = t & 63, wv = t >> 6;
    const float *Wl = a.cW[d.n_c - 1], *bl = a.cb[d.n_c - 1];
    for (int n = wv; n < d.nout; n += kWaves) {
      float s = 0.0f;
      for (int k = lane; k < K; k += 64) s += vin[k] * Wl[(size_t)k * d.nout + n];
      s = wave_sum(s);
      if (lane == 0) a.co[(size_t)b * d.nout + n] = s + bl[n];
    }
  }
}

struct BwdArgs {
  Dims d;
  const float *feat, *Wg;
  const float *gW[kMaxL], *cW[kMaxL];
  const float *save, *save_c;         // from the forward
  const float *dh_last, *dco;         // [B, hid] (may be null), [B, nout] (may be null)
  float *dfeat;                       // [B, G, Cf]
  float *dpre, *dlog;                 // [B, iters, 4 hid], [B, iters, G]: pre-activation gradients
  float *dzg;                         // [n_g - 1][rows][B, iters, hid]: the glimpse MLP's hidden layers (stride dzg_stride floats per layer)
  float *dzc;                         // [n_c - 1][.][B, mlp]: the controller MLP's hidden layers (stride dzc_stride)
  size_t dzg_stride, dzc_stride;
};

__global__ __launch_bounds__(kThreads) void ctrl_bwd_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims d = a.d;
  const int t = threadIdx.x, b = blockIdx.x, G = d.G, Cf = d.Cf, hid = d.hid;
  float *red = smem;                  // kWaves
  float *dh = red + 64;               // [hid]
  float *dc = dh + hid;               // [hid]
  float *dp = dc + hid;               // [4 hid]  d pre (gates)
  float *dxh = dp + 4 * hid;          // [Cf + hid]
  float *dgm = dxh + Cf + hid;        // [G]   d map of the iteration just processed (read by the previous one's softmax)
  const int VT = max3(hid, G, d.mlp);
  float *vt = dgm + G;                // [max(hid, G, mlp)] scratch
  float *sc2 = vt + VT;               // [max(hid, mlp)] scratch
  float *fl = sc2 + (hid > d.mlp ? hid : d.mlp);  // [G * Cf] feature map
  float *dfl = fl + G * Cf;           // [G * Cf] d feat
  const float *fsrc = a.feat + (size_t)b * G * Cf;
  for (int e = t * 4; e < G * Cf; e += kThreads * 4) {
    *reinterpret_cast<f32x4 *>(fl + e) = *reinterpret_cast<const f32x4 *>(fsrc + e);
    *reinterpret_cast<f32x4 *>(dfl + e) = f32x4{0, 0, 0, 0};
  }
  for (int e = t; e < hid; e += kThreads) {
    dh[e] = a.dh_last ? a.dh_last[(size_t)b * hid + e] : 0.0f;
    dc[e] = 0.0f;
  }
  __syncthreads();
  if (a.dco) {  // controller MLP backward: co = v Wc + bc behind (n_c - 1) ReLU layers
    const int Kl = d.n_c == 1 ? hid : d.mlp;
    const float *Wl = a.cW[d.n_c - 1];
    float *dv = d.n_c == 1 ? dxh : vt;  // d (input of the last layer); dxh is free until the loop below
    for (int k = t; k < Kl; k += kThreads) {
      float s = 0.0f;
      for (int n = 0; n < d.nout; ++n) s += Wl[(size_t)k * d.nout + n] * a.dco[(size_t)b * d.nout + n];
      dv[k] = s;
    }
    __syncthreads();
    for (int l = d.n_c - 2; l >= 0; --l) {  // vt holds d (output of hidden layer l)
      const float *zc = a.save_c + ((size_t)b * (d.n_c - 1) + l) * d.mlp;
      float *o = a.dzc + (size_t)l * a.dzc_stride + (size_t)b * d.mlp;
      for (int n = t; n < d.mlp; n += kThreads) {
        const float v = zc[n] > 0.0f ? vt[n] : 0.0f;
        vt[n] = v;
        o[n] = v;
      }
      __syncthreads();
      if (l > 0) {
        gemv_rows(vt, d.mlp, a.cW[l], d.mlp, sc2, false);  // d (output of hidden layer l - 1)
        for (int n = t; n < d.mlp; n += kThreads) vt[n] = sc2[n];
        __syncthreads();
      } else {
        gemv_rows(vt, d.mlp, a.cW[0], hid, dxh, false);
      }
    }
    for (int k = t; k < hid; k += kThreads) dh[k] += dxh[k];
    __syncthreads();
  }
  const int SF = save_floats(d);
  for (int it = d.iters - 1; it >= 0; --it) {
    const float *sv = a.save + ((size_t)b * d.iters + it) * SF;
    const float *sv_act = sv + (Cf + hid), *sv_c = sv_act + 4 * hid, *sv_z = sv_c + hid, *sv_gm = sv_z + (d.n_g - 1) * hid;
    float *o_dpre = a.dpre + ((size_t)b * d.iters + it) * 4 * hid;
    float *o_dlog = a.dlog + ((size_t)b * d.iters + it) * G;
    const size_t zrow = ((size_t)b * d.iters + it) * hid;  // this (image, iteration)'s row in every dzg layer
    if (it < d.iters - 1) {
      // the glimpse MLP behind this iteration's LSTM produced the map iteration it + 1 read with (saved there);
      // dgm holds that map's gradient.  softmax: dlog = gm (dgm - sum gm dgm)
      const float *gm_next = a.save + ((size_t)b * d.iters + it + 1) * SF + (Cf + hid) + 4 * hid + hid + (d.n_g - 1) * hid;
      float dot = 0.0f;
      for (int n = t; n < G; n += kThreads) dot += gm_next[n] * dgm[n];
      dot = block_reduce(dot, false, red);
      for (int n = t; n < G; n += kThreads) {
        const float v = gm_next[n] * (dgm[n] - dot);
        vt[n] = v;
        o_dlog[n] = v;
      }
      __syncthreads();
      // back through the glimpse MLP: d (input of the last layer), then each hidden layer's ReLU and weights
      if (d.n_g == 1) {
        gemv_rows(vt, G, a.gW[0], hid, dh, true);  // one layer: logits = h W + b
      } else {
        gemv_rows(vt, G, a.gW[d.n_g - 1], hid, dxh, false);  // d z_{n_g-2} (dxh as scratch)
        for (int l = d.n_g - 2; l >= 0; --l) {
          float *o_dz = a.dzg + (size_t)l * a.dzg_stride + zrow;
          for (int k = t; k < hid; k += kThreads) {
            const float v = sv_z[l * hid + k] > 0.0f ? dxh[k] : 0.0f;
            vt[k] = v;
            o_dz[k] = v;
          }
          __syncthreads();
          if (l > 0)
            gemv_rows(vt, hid, a.gW[l], hid, dxh, false);  // d z_{l-1}
          else
            gemv_rows(vt, hid, a.gW[0], hid, dh, true);    // d h += W0 d z0pre
        }
      }
    } else {
      for (int n = t; n < G; n += kThreads) o_dlog[n] = 0.0f;
      for (int l = 0; l + 1 < d.n_g; ++l)
        for (int k = t; k < hid; k += kThreads) a.dzg[(size_t)l * a.dzg_stride + zrow + k] = 0.0f;
    }
    // LSTM cell backward (nnlib.py:641-646)
    if (t < hid) {
      const float gi = sv_act[t], gf = sv_act[hid + t], go = sv_act[2 * hid + t], u = sv_act[3 * hid + t];
      const float c = sv_c[t], cp = it > 0 ? (sv_c - SF)[t] : 0.0f;
      const float tc = tanhf(c);
      const float dho = dh[t];
      const float dct = dc[t] + dho * go * (1.0f - tc * tc);
      const float p0 = dct * u * gi * (1.0f - gi), p1 = dct * cp * gf * (1.0f - gf), p2 = dho * tc * go * (1.0f - go),
                  p3 = dct * gi * (1.0f - u * u);
      dp[t] = p0;
      dp[hid + t] = p1;
      dp[2 * hid + t] = p2;
      dp[3 * hid + t] = p3;
      o_dpre[t] = p0;
      o_dpre[hid + t] = p1;
      o_dpre[2 * hid + t] = p2;
      o_dpre[3 * hid + t] = p3;
      dc[t] = dct * gf;
    }
    __syncthreads();
    gemv_rows(dp, 4 * hid, a.Wg, Cf + hid, dxh, false);  // [d glimpse | d h_prev]
    for (int k = t; k < hid; k += kThreads) dh[k] = dxh[Cf + k];
    // d feat[g, c] += map[g] d glimpse[c];  d map[g] = sum_c feat[g, c] d glimpse[c]
    for (int e = t; e < G * Cf; e += kThreads) {
      const int g = e / Cf, c = e - g * Cf;
      dfl[e] += sv_gm[g] * dxh[c];
    }
    if (it > 0) {
      const int lane = t & 63, wv = t >> 6;
      for (int g = wv; g < G; g += kWaves) {
        float s = 0.0f;
        for (int c = lane; c < Cf; c += 64) s += fl[(size_t)g * Cf + c] * dxh[c];
        s = wave_sum(s);
        if (lane == 0) dgm[g] = s;
      }
    }
    __syncthreads();
  }
  float *dst = a.dfeat + (size_t)b * G * Cf;
  for (int e = t * 4; e < G * Cf; e += kThreads * 4) *reinterpret_cast<f32x4 *>(dst + e) = *reinterpret_cast<const f32x4 *>(dfl + e);
}

}  // namespace ctrlt
}  // namespace ra

using namespace ra;

namespace {
using ra::ctrlt::Dims;
size_t fwd_lds_floats(const Dims &d) {
  return (size_t)16 * 4 * d.hid + (d.Cf + d.hid) + d.hid + ra::ctrlt::va_floats(d) + d.G + (size_t)d.G * d.Cf;
}
size_t bwd_lds_floats(const Dims &d) {
  return (size_t)64 + 2 * d.hid + 4 * d.hid + (d.Cf + d.hid) + d.G + ra::ctrlt::max3(d.hid, d.G, d.mlp) + (d.hid > d.mlp ? d.hid : d.mlp) +
         (size_t)2 * d.G * d.Cf;
}
bool ctrl_train_dims_ok(const Dims &d) {
  return d.G > 0 && d.Cf > 0 && d.hid > 0 && d.iters > 0 && d.nout > 0 && d.nout <= 64 && (d.G % 4) == 0 && (d.Cf % 4) == 0 &&
         (d.hid % 4) == 0 && 4 * d.hid <= 1024 && d.G <= 1024 && d.G <= 4 * d.hid && d.Cf <= 1024 && d.n_g >= 1 &&
         d.n_g <= ra::ctrlt::kMaxL && d.n_c >= 1 && d.n_c <= ra::ctrlt::kMaxL &&
         (d.n_c == 1 || (d.mlp > 0 && (d.mlp % 4) == 0 && d.mlp <= 4 * d.hid)) && fwd_lds_floats(d) * 4 <= 160 * 1024 &&
         bwd_lds_floats(d) * 4 <= 160 * 1024;
}
Dims dims_of(int G, int Cf, int hid, int iters, int nout, int n_g, int n_c, int mlp) {
  return Dims{G, Cf, hid, iters, nout, n_g, n_c, n_c > 1 ? mlp : 4};
}
}  // namespace

extern "C" int ra_ctrl_train_supported_n(int G, int Cf, int hid, int iters, int nout, int n_glimpse, int n_ctrl, int mlp_dim) {
  return ctrl_train_dims_ok(dims_of(G, Cf, hid, iters, nout, n_glimpse, n_ctrl, mlp_dim)) ? 1 : 0;
}
extern "C" int ra_ctrl_train_supported(int G, int Cf, int hid, int iters, int nout) {
  return ra_ctrl_train_supported_n(G, Cf, hid, iters, nout, 2, 1, 0);
}

extern "C" size_t ra_ctrl_train_save_floats_n(int G, int Cf, int hid, int iters, int n_glimpse) {
  return (size_t)iters * ctrlt::save_floats(dims_of(G, Cf, hid, iters, 9, n_glimpse, 1, 0));
}
extern "C" size_t ra_ctrl_train_save_floats(int G, int Cf, int hid, int iters) { return ra_ctrl_train_save_floats_n(G, Cf, hid, iters, 2); }

// Any depth: gW / gb (n_glimpse host pointers to device tensors), cW / cb (n_ctrl); save_c [B, (n_ctrl - 1) mlp_dim] (NULL
// for one controller-MLP layer).
extern "C" int ra_ctrl_train_fwd_n_f32(int B, int G, int Cf, int hid, int iters, int nout, int n_glimpse, int n_ctrl, int mlp_dim,
                                       const float *feat, const float *Wg, const float *bg, const float *const *gW,
                                       const float *const *gb, const float *const *cW, const float *const *cb, float *h_last,
                                       float *co, float *save, float *save_c, void *stream) {
  const Dims d = dims_of(G, Cf, hid, iters, nout, n_glimpse, n_ctrl, mlp_dim);
  if (!feat || !Wg || !bg || !gW || !gb || !cW || !cb || !h_last || !co || !save || B <= 0 || (n_ctrl > 1 && !save_c))
    return fail(RA_E_INVALID, "ra_ctrl_train_fwd_f32: bad argument");
  if (!ctrl_train_dims_ok(d)) return fail(RA_E_SHAPE, "ra_ctrl_train_fwd_f32: unsupported dimensions");
  ctrlt::FwdArgs a{};
  a.d = d, a.feat = feat, a.Wg = Wg, a.bg = bg, a.h_last = h_last, a.co = co, a.save = save, a.save_c = save_c;
  for (int l = 0; l < n_glimpse; ++l) {
    if (!gW[l] || !gb[l]) return fail(RA_E_INVALID, "ra_ctrl_train_fwd_f32: null glimpse-MLP layer %d", l);
    a.gW[l] = gW[l], a.gb[l] = gb[l];
  }
  for (int l = 0; l < n_ctrl; ++l) {
    if (!cW[l] || !cb[l]) return fail(RA_E_INVALID, "ra_ctrl_train_fwd_f32: null controller-MLP layer %d", l);
    a.cW[l] = cW[l], a.cb[l] = cb[l];
  }
  const size_t lds = fwd_lds_floats(d) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctrlt::ctrl_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ctrlt::ctrl_fwd_kernel, dim3(B), dim3(ctrlt::kThreads), lds, as_stream(stream), a);
  return launch_status("ra_ctrl_train_fwd_f32");
}

extern "C" int ra_ctrl_train_fwd_f32(int B, int G, int Cf, int hid, int iters, int nout, const float *feat, const float *Wg,
                                     const float *bg, const float *W0, const float *b0, const float *W1, const float *b1,
                                     const float *Wc, const float *bc, float *h_last, float *co, float *save, void *stream) {
  const float *gW[2] = {W0, W1}, *gb[2] = {b0, b1}, *cW[1] = {Wc}, *cb[1] = {bc};
  return ra_ctrl_train_fwd_n_f32(B, G, Cf, hid, iters, nout, 2, 1, 0, feat, Wg, bg, gW, gb, cW, cb, h_last, co, save, nullptr, stream);
}

// dzg: [n_glimpse - 1] layers, dzg_stride floats apart, each [B, iters, hid]; dzc: [n_ctrl - 1] layers, dzc_stride apart, each
// [B, mlp_dim] (NULL where the depth has no hidden layer).
extern "C" int ra_ctrl_train_bwd_n_f32(int B, int G, int Cf, int hid, int iters, int nout, int n_glimpse, int n_ctrl, int mlp_dim,
                                       const float *feat, const float *Wg, const float *const *gW, const float *const *cW,
                                       const float *save, const float *save_c, const float *dh_last, const float *dco, float *dfeat,
                                       float *dpre, float *dlog, float *dzg, size_t dzg_stride, float *dzc, size_t dzc_stride,
                                       void *stream) {
  const Dims d = dims_of(G, Cf, hid, iters, nout, n_glimpse, n_ctrl, mlp_dim);
  if (!feat || !Wg || !gW || !cW || !save || !dfeat || !dpre || !dlog || B <= 0 || (n_glimpse > 1 && !dzg) ||
      (n_ctrl > 1 && (!dzc || !save_c)))
    return fail(RA_E_INVALID, "ra_ctrl_train_bwd_f32: bad argument");
  if (!ctrl_train_dims_ok(d)) return fail(RA_E_SHAPE, "ra_ctrl_train_bwd_f32: unsupported dimensions");
  ctrlt::BwdArgs a{};
  a.d = d, a.feat = feat, a.Wg = Wg, a.save = save, a.save_c = save_c, a.dh_last = dh_last, a.dco = dco, a.dfeat = dfeat;
  a.dpre = dpre, a.dlog = dlog, a.dzg = dzg, a.dzc = dzc, a.dzg_stride = dzg_stride, a.dzc_stride = dzc_stride;
  for (int l = 0; l < n_glimpse; ++l) {
    if (!gW[l]) return fail(RA_E_INVALID, "ra_ctrl_train_bwd_f32: null glimpse-MLP layer %d", l);
    a.gW[l] = gW[l];
  }
  for (int l = 0; l < n_ctrl; ++l) {
    if (!cW[l]) return fail(RA_E_INVALID, "ra_ctrl_train_bwd_f32: null controller-MLP layer %d", l);
    a.cW[l] = cW[l];
  }
  const size_t lds = bwd_lds_floats(d) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctrlt::ctrl_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ctrlt::ctrl_bwd_kernel, dim3(B), dim3(ctrlt::kThreads), lds, as_stream(stream), a);
  return launch_status("ra_ctrl_train_bwd_f32");
}

extern "C" int ra_ctrl_train_bwd_f32(int B, int G, int Cf, int hid, int iters, int nout, const float *feat, const float *Wg,
                                     const float *W0, const float *W1, const float *Wc, const float *save,
                                     const float *dh_last, const float *dco, float *dfeat, float *dpre, float *dz1, float *dlog,
                                     void *stream) {
  const float *gW[2] = {W0, W1}, *cW[1] = {Wc};
  return ra_ctrl_train_bwd_n_f32(B, G, Cf, hid, iters, nout, 2, 1, 0, feat, Wg, gW, cW, save, nullptr, dh_last, dco, dfeat, dpre, dlog, dz1, 0,
                                 nullptr, 0, stream);
}
