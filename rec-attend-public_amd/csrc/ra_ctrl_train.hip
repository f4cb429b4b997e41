// The controller of the TRAINING graph (full_model.py:668-689: soft-attention glimpse read-out -> dense LSTM ->
// glimpse MLP with a softmax over the feature map, num_ctrl_rnn_iter times, then the controller MLP) as one
// forward and one backward launch per timestep instead of ~35 + ~70 library GEMMs and element-wise launches.
// One workgroup (16 waves) per image, the geometry of ra_ctrl.hip's inference kernel: the feature map lives in
// LDS, the 1.8 MB of weights stream from L2 (shared by all images).  The forward saves what the backward needs
// per glimpse iteration; the backward runs the recurrence in reverse (BPTT), accumulates d feat in LDS, and
// writes the pre-activation gradients of every dense layer.  Parameter gradients are NOT formed here: they are
// sums over images, iterations and timesteps of (layer input)^T (pre-activation gradient), i.e. ONE GEMM per weight
// matrix per optimisation step over the saved rows (ra_train.ControllerFn: four addmm_ per step instead of 960
// launches).  Architecture: num_glimpse_mlp_layers = 2, num_ctrl_mlp_layers = 1 (every run script); other depths
// keep the library path.
#include "ra_common.h"

namespace ra {
namespace ctrlt {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

struct Dims {
  int G, Cf, hid, iters, nout;  // nout = 9 controller outputs
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
// c [hid] | z1 [hid] (glimpse-MLP hidden layer after ReLU) | gm [G] (the map this iteration READ with)
__host__ __device__ inline int save_floats(const Dims &d) { return (d.Cf + d.hid) + 4 * d.hid + d.hid + d.hid + d.G; }

struct FwdArgs {
  Dims d;
  const float *feat;                       // [B, G, Cf]
  const float *Wg, *bg;                    // [Cf + hid, 4 hid] gate order i f o u, [4 hid]
  const float *W0, *b0, *W1, *b1;          // [hid, hid], [hid], [hid, G], [G]
  const float *Wc, *bc;                    // [hid, nout], [nout]
  float *h_last, *co;                      // [B, hid], [B, nout]
  float *save;                             // [B, iters, save_floats]
};

__global__ __launch_bounds__(kThreads) void ctrl_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims d = a.d;
  const int t = threadIdx.x, b = blockIdx.x, G = d.G, Cf = d.Cf, hid = d.hid;
  float *red = smem;                         // 16 * 4 hid
  float *xh = red + 16 * 4 * hid;            // [Cf + hid]
  float *cst = xh + Cf + hid;                // [hid]
  float *va = cst + hid;                     // [4 hid]
  float *gm = va + 4 * hid;                  // [G]
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
    float *sv_act = sv + (Cf + hid), *sv_c = sv_act + 4 * hid, *sv_z1 = sv_c + hid, *sv_gm = sv_z1 + hid;
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
      gemv_cols(xh + Cf, hid, a.W0, hid, a.b0, va, red);
      for (int n = t; n < hid; n += kThreads) {
        const float z = fmaxf(va[n], 0.0f);
        va[n] = z;
        sv_z1[n] = z;
      }
      __syncthreads();
      gemv_cols(va, hid, a.W1, G, a.b1, va + hid, red);
      float *lg = va + hid;
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
      for (int n = t; n < hid; n += kThreads) sv_z1[n] = 0.0f;
    }
  }
  // controller MLP (one layer): co = h Wc + bc
  if (t < hid) a.h_last[(size_t)b * hid + t] = xh[Cf + t];
  {  // a wave per output, pairwise reduction (a 256-term serial sum costs half a digit the window position amplifies)
    const int lane = t & 63, wv = t >> 6;
    for (int n = wv; n < d.nout; n += kWaves) {
      float s = 0.0f;
      for (int k = lane; k < hid; k += 64) s += xh[Cf + k] * a.Wc[(size_t)k * d.nout + n];
      s = wave_sum(s);
      if (lane == 0) a.co[(size_t)b * d.nout + n] = s + a.bc[n];
    }
  }
}

struct BwdArgs {
  Dims d;
  const float *feat, *Wg, *W0, *W1, *Wc;
  const float *save;                  // from the forward
  const float *dh_last, *dco;         // [B, hid] (may be null), [B, nout] (may be null)
  float *dfeat;                       // [B, G, Cf]
  float *dpre, *dz1, *dlog;           // [B, iters, 4 hid], [B, iters, hid], [B, iters, G]: pre-activation gradients
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
  float *vt = dgm + G;                // [max(hid, G)] scratch
  float *fl = vt + (hid > G ? hid : G);  // [G * Cf] feature map
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
  if (a.dco && t < hid) {  // co = h Wc + bc
    float s = 0.0f;
    for (int n = 0; n < d.nout; ++n) s += a.Wc[(size_t)t * d.nout + n] * a.dco[(size_t)b * d.nout + n];
    dh[t] += s;
  }
  __syncthreads();
  const int SF = save_floats(d);
  for (int it = d.iters - 1; it >= 0; --it) {
    const float *sv = a.save + ((size_t)b * d.iters + it) * SF;
    const float *sv_act = sv + (Cf + hid), *sv_c = sv_act + 4 * hid, *sv_z1 = sv_c + hid, *sv_gm = sv_z1 + hid;
    float *o_dpre = a.dpre + ((size_t)b * d.iters + it) * 4 * hid;
    float *o_dz1 = a.dz1 + ((size_t)b * d.iters + it) * hid, *o_dlog = a.dlog + ((size_t)b * d.iters + it) * G;
    if (it < d.iters - 1) {
      // the glimpse MLP behind this iteration's LSTM produced the map iteration it + 1 read with (saved there);
      // dgm holds that map's gradient.  softmax: dlog = gm (dgm - sum gm dgm)
      const float *gm_next = a.save + ((size_t)b * d.iters + it + 1) * SF + (Cf + hid) + 4 * hid + hid + hid;
      float dot = 0.0f;
      for (int n = t; n < G; n += kThreads) dot += gm_next[n] * dgm[n];
      dot = block_reduce(dot, false, red);
      for (int n = t; n < G; n += kThreads) {
        const float v = gm_next[n] * (dgm[n] - dot);
        vt[n] = v;
        o_dlog[n] = v;
      }
      __syncthreads();
      gemv_rows(vt, G, a.W1, hid, dxh, false);  // d z1 [hid] (dxh as scratch)
      for (int k = t; k < hid; k += kThreads) {
        const float v = sv_z1[k] > 0.0f ? dxh[k] : 0.0f;
        vt[k] = v;
        o_dz1[k] = v;
      }
      __syncthreads();
      gemv_rows(vt, hid, a.W0, hid, dh, true);  // d h += W0 d z1pre
    } else {
      for (int n = t; n < G; n += kThreads) o_dlog[n] = 0.0f;
      for (int k = t; k < hid; k += kThreads) o_dz1[k] = 0.0f;
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
size_t fwd_lds_floats(int G, int Cf, int hid) { return (size_t)16 * 4 * hid + (Cf + hid) + hid + 4 * hid + G + (size_t)G * Cf; }
size_t bwd_lds_floats(int G, int Cf, int hid) {
  return (size_t)64 + 2 * hid + 4 * hid + (Cf + hid) + G + (hid > G ? hid : G) + (size_t)2 * G * Cf;
}
bool ctrl_train_dims_ok(int G, int Cf, int hid, int iters, int nout) {
  return G > 0 && Cf > 0 && hid > 0 && iters > 0 && nout > 0 && nout <= 64 && (G % 4) == 0 && (Cf % 4) == 0 && (hid % 4) == 0 &&
         4 * hid <= 1024 && G <= 1024 && Cf <= 1024 && fwd_lds_floats(G, Cf, hid) * 4 <= 160 * 1024 &&
         bwd_lds_floats(G, Cf, hid) * 4 <= 160 * 1024;
}
}  // namespace

extern "C" int ra_ctrl_train_supported(int G, int Cf, int hid, int iters, int nout) {
  return ctrl_train_dims_ok(G, Cf, hid, iters, nout) ? 1 : 0;
}

extern "C" size_t ra_ctrl_train_save_floats(int G, int Cf, int hid, int iters) {
  ctrlt::Dims d{G, Cf, hid, iters, 9};
  return (size_t)iters * ctrlt::save_floats(d);
}

extern "C" int ra_ctrl_train_fwd_f32(int B, int G, int Cf, int hid, int iters, int nout, const float *feat, const float *Wg,
                                     const float *bg, const float *W0, const float *b0, const float *W1, const float *b1,
                                     const float *Wc, const float *bc, float *h_last, float *co, float *save, void *stream) {
  if (!feat || !Wg || !bg || !W0 || !b0 || !W1 || !b1 || !Wc || !bc || !h_last || !co || !save || B <= 0)
    return fail(RA_E_INVALID, "ra_ctrl_train_fwd_f32: bad argument");
  if (!ctrl_train_dims_ok(G, Cf, hid, iters, nout)) return fail(RA_E_SHAPE, "ra_ctrl_train_fwd_f32: unsupported dimensions");
  ctrlt::FwdArgs a{{G, Cf, hid, iters, nout}, feat, Wg, bg, W0, b0, W1, b1, Wc, bc, h_last, co, save};
  const size_t lds = fwd_lds_floats(G, Cf, hid) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctrlt::ctrl_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ctrlt::ctrl_fwd_kernel, dim3(B), dim3(ctrlt::kThreads), lds, as_stream(stream), a);
  return launch_status("ra_ctrl_train_fwd_f32");
}

extern "C" int ra_ctrl_train_bwd_f32(int B, int G, int Cf, int hid, int iters, int nout, const float *feat, const float *Wg,
                                     const float *W0, const float *W1, const float *Wc, const float *save,
                                     const float *dh_last, const float *dco, float *dfeat, float *dpre, float *dz1, float *dlog,
                                     void *stream) {
  if (!feat || !Wg || !W0 || !W1 || !Wc || !save || !dfeat || !dpre || !dz1 || !dlog || B <= 0)
    return fail(RA_E_INVALID, "ra_ctrl_train_bwd_f32: bad argument");
  if (!ctrl_train_dims_ok(G, Cf, hid, iters, nout)) return fail(RA_E_SHAPE, "ra_ctrl_train_bwd_f32: unsupported dimensions");
  ctrlt::BwdArgs a{{G, Cf, hid, iters, nout}, feat, Wg, W0, W1, Wc, save, dh_last, dco, dfeat, dpre, dz1, dlog};
  const size_t lds = bwd_lds_floats(G, Cf, hid) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctrlt::ctrl_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ctrlt::ctrl_bwd_kernel, dim3(B), dim3(ctrlt::kThreads), lds, as_stream(stream), a);
  return launch_status("ra_ctrl_train_bwd_f32");
}
