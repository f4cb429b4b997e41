// K2 — the controller: soft-attention glimpse read-out + dense LSTM + glimpse MLP, iterated
// num_ctrl_rnn_iter times, then the controller MLP and the decode of its 9 outputs into
// attention parameters.  full_model.py:668-722 / box_model.py:416-468; nnlib.py:476-493
// (run_mlp), :637-649 (LSTM unroll, state = [c | h], zero state every output step).
// K6 — small dense layers (score MLP, full_model.py:821-822).
//
// One workgroup (16 waves) per example: the whole recurrence stays on one CU with the feature
// map in LDS; the ~1.8 MiB of weights are streamed from L2 with 16-byte loads each iteration
// (they are shared by all examples, so they stay L2/MALL resident).  This is a latency-bound
// GEMV chain, not a roofline kernel.
#include "ra_common.h"

namespace ra {
namespace ctrl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024;
constexpr int kRed = 4096;         // floats of reduction scratch
constexpr int kMaxFeatLds = 24576; // floats (96 KiB) of feature map kept in LDS

struct Layout {
  size_t lstm_w, lstm_b;
  size_t gmlp_w[8], gmlp_b[8];
  size_t cmlp_w[8], cmlp_b[8];
  int gmlp_in[8], gmlp_out[8], gmlp_outp[8];
  int cmlp_in[8], cmlp_out[8], cmlp_outp[8];
  size_t total;
};

__host__ __device__ inline Layout layout(const ra_ctrl_desc &d) {
  Layout L;
  size_t off = 0;
  L.lstm_w = off;
  off += (size_t)(d.Cf + d.hid) * 4 * d.hid;
  L.lstm_b = off;
  off += (size_t)4 * d.hid;
  for (int l = 0; l < d.n_gmlp; ++l) {
    L.gmlp_in[l] = d.hid;
    L.gmlp_out[l] = (l == d.n_gmlp - 1) ? d.G : d.hid;
    L.gmlp_outp[l] = round_up(L.gmlp_out[l], 4);
    L.gmlp_w[l] = off;
    off += (size_t)L.gmlp_in[l] * L.gmlp_outp[l];
    L.gmlp_b[l] = off;
    off += L.gmlp_outp[l];
  }
  for (int l = 0; l < d.n_cmlp; ++l) {
    L.cmlp_in[l] = (l == 0) ? d.hid : d.mlp_dim;
    L.cmlp_out[l] = (l == d.n_cmlp - 1) ? 9 : d.mlp_dim;
    L.cmlp_outp[l] = round_up(L.cmlp_out[l], 4);
    L.cmlp_w[l] = off;
    off += (size_t)L.cmlp_in[l] * L.cmlp_outp[l];
    L.cmlp_b[l] = off;
    off += L.cmlp_outp[l];
  }
  L.total = off;
  return L;
}

// out[n] = sum_k xs[k] * Wt[k][n] for n < N (N % 4 == 0); xs, out, red in LDS.  All kThreads call.
__device__ void gemv(const float *xs, int K, const float *__restrict__ Wt, int N, float *out,
                     float *red) {
  const int t = threadIdx.x;
  const int quads = N / 4;
  int parts = kThreads / quads;
  if (parts > kRed / N) parts = kRed / N;
  if (parts > 32) parts = 32;
  if (parts < 1) parts = 1;
  // quads may exceed the thread count only if N > 4096, which the host rejects
  if (t < quads * parts) {
    const int qd = t % quads, part = t / quads;
    f32x4 acc = f32x4{0, 0, 0, 0};
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(Wt) + qd;
#pragma unroll 8
    for (int k = part; k < K; k += parts) acc += xs[k] * wp[(size_t)k * quads];
    *reinterpret_cast<f32x4 *>(red + (size_t)part * N + 4 * qd) = acc;
  }
  __syncthreads();
  for (int n = t; n < N; n += kThreads) {
    float s = 0.0f;
    for (int p = 0; p < parts; ++p) s += red[p * N + n];
    out[n] = s;
  }
  __syncthreads();
}

__device__ inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }

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
  for (int w = 1; w < kThreads / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kThreads) void controller_kernel(const ra_ctrl_desc d,
                                                               const float *feat,
                                                               const float *__restrict__ wp,
                                                               float *h_last, float *ctrl_out,
                                                               float *gmaps, float *attn,
                                                               int feat_in_lds, int prio) {
  raise_prio(prio);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Layout L = layout(d);
  const int t = threadIdx.x, b = blockIdx.x;
  const int G = d.G, Cf = d.Cf, hid = d.hid;
  const int Gp = round_up(G, 4);
  int vmax = 4 * hid;
  if (Gp > vmax) vmax = Gp;
  if (d.mlp_dim > vmax) vmax = d.mlp_dim;
  // LDS carve (all offsets multiples of 4 floats)
  float *red = smem;                    // kRed
  float *xh = red + kRed;               // [Cf + hid]  LSTM input = [glimpse ; h]
  float *cst = xh + round_up(Cf + hid, 4);  // [hid] cell state
  float *va = cst + hid;                // [vmax] scratch vector A
  float *vb = va + vmax;                // [vmax] scratch vector B
  float *gm = vb + vmax;                // [Gp] glimpse map
  float *fl = gm + Gp;                  // [G*Cf] feature map (optional)
  const float *fsrc = feat + (size_t)b * G * Cf;
  if (feat_in_lds) {
    for (int e = t * 4; e < G * Cf; e += kThreads * 4)
      *reinterpret_cast<f32x4 *>(fl + e) = *reinterpret_cast<const f32x4 *>(fsrc + e);
    fsrc = fl;
  }
  for (int e = t; e < hid; e += kThreads) {
    xh[Cf + e] = 0.0f;  // h = 0
    cst[e] = 0.0f;      // c = 0   (full_model.py:674)
  }
  for (int g = t; g < Gp; g += kThreads) gm[g] = (g < G) ? 1.0f / (float)G : 0.0f;  // :676-677
  __syncthreads();

  for (int it = 0; it < d.iters; ++it) {
    if (gmaps)
      for (int g = t; g < G; g += kThreads) gmaps[((size_t)b * d.iters + it) * G + g] = gm[g];
    // glimpse[c] = sum_g feat[g,c] * map[g]   (full_model.py:680)
    {
      const int parts = kThreads / Cf;  // Cf <= 1024
      const int c = t % Cf, part = t / Cf;
      if (part < parts) {
        float s = 0.0f;
        for (int g = part; g < G; g += parts) s += fsrc[(size_t)g * Cf + c] * gm[g];
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
    // LSTM (nnlib.py:641-646): columns gate-major i, f, o, u
    gemv(xh, Cf + hid, wp + L.lstm_w, 4 * hid, va, red);
    if (t < hid) {
      const float *bb = wp + L.lstm_b;
      const float gi = sigm(va[t] + bb[t]);
      const float gf = sigm(va[hid + t] + bb[hid + t]);
      const float go = sigm(va[2 * hid + t] + bb[2 * hid + t]);
      const float u = tanhf(va[3 * hid + t] + bb[3 * hid + t]);
      const float c = gf * cst[t] + gi * u;
      cst[t] = c;
      xh[Cf + t] = go * tanhf(c);
    }
    __syncthreads();
    // glimpse MLP (full_model.py:350-352,686-688); its output is unused after the last iteration
    if (it < d.iters - 1) {
      const float *in = xh + Cf;
      float *o1 = va, *o2 = vb;
      for (int l = 0; l < d.n_gmlp; ++l) {
        const int N = L.gmlp_outp[l], No = L.gmlp_out[l];
        gemv(in, L.gmlp_in[l], wp + L.gmlp_w[l], N, o1, red);
        const float *bb = wp + L.gmlp_b[l];
        if (l < d.n_gmlp - 1) {
          for (int n = t; n < N; n += kThreads) o1[n] = fmaxf(o1[n] + bb[n], 0.0f);
          __syncthreads();
          in = o1;
          float *tmp = o1;
          o1 = o2;
          o2 = tmp;
        } else {  // softmax over G
          float mx = -3.0e38f;
          for (int n = t; n < No; n += kThreads) {
            o1[n] += bb[n];
            mx = fmaxf(mx, o1[n]);
          }
          mx = block_reduce(mx, true, red);
          float sum = 0.0f;
          for (int n = t; n < No; n += kThreads) {
            const float e = expf(o1[n] - mx);
            o1[n] = e;
            sum += e;
          }
          sum = block_reduce(sum, false, red);
          for (int n = t; n < Gp; n += kThreads) gm[n] = (n < No) ? o1[n] / sum : 0.0f;
          __syncthreads();
        }
      }
    }
  }
  // controller MLP (full_model.py:382-384,689)
  {
    const float *in = xh + Cf;
    float *o1 = va, *o2 = vb;
    for (int l = 0; l < d.n_cmlp; ++l) {
      const int N = L.cmlp_outp[l];
      gemv(in, L.cmlp_in[l], wp + L.cmlp_w[l], N, o1, red);
      const float *bb = wp + L.cmlp_b[l];
      const bool last = (l == d.n_cmlp - 1);
      for (int n = t; n < N; n += kThreads) {
        const float v = o1[n] + bb[n];
        o1[n] = last ? v : fmaxf(v, 0.0f);
      }
      __syncthreads();
      in = o1;
      float *tmp = o1;
      o1 = o2;
      o2 = tmp;
    }
    const float *co = in;  // 9 outputs
    if (t < hid && h_last) h_last[(size_t)b * hid + t] = xh[Cf + t];
    if (t < 9 && ctrl_out) ctrl_out[(size_t)b * 9 + t] = co[t];
    if (t == 0 && attn) {
      float *r = attn + (size_t)b * RA_ATTN_STRIDE;
      float cn[2] = {co[0], co[1]}, ls[2] = {co[2], co[3]};
      if (d.squash) {  // full_model.py:695-697
        cn[0] = tanhf(cn[0]);
        cn[1] = tanhf(cn[1]);
        ls[0] = -log1pf(expf(ls[0]));
        ls[1] = -log1pf(expf(ls[1]));
      }
      const float dim[2] = {(float)d.H, (float)d.W}, fs[2] = {(float)d.Fh, (float)d.Fw};
      for (int k = 0; k < 2; ++k) {
        const float ctr = (cn[k] + 1.0f) * (dim[k] / 2.0f);  // modellib.py:761-763
        const float size = expf(ls[k]) * dim[k];             // modellib.py:821-823
        float lv = d.fixed_var ? 0.0f : logf(size) - logf(fs[k]);  // modellib.py:791-792
        if (d.dynamic_var) lv = co[4 + k];
        r[0 + k] = ctr;
        r[2 + k] = size;
        r[4 + k] = lv;
        r[9 + k] = cn[k];
        r[11 + k] = ls[k];
      }
      r[6] = d.fixed_gamma ? 1.0f : expf(co[6]);  // full_model.py:711-719
      r[7] = expf(co[7]);
      r[8] = d.fixed_gamma ? 2.0f : co[8];
      r[13] = r[14] = r[15] = 0.0f;
    }
  }
}

// out[b, n] = act(sum_k [x0|x1][b,k] W[k,n] + bias[n]); one workgroup per example.
__global__ __launch_bounds__(256) void dense_kernel(const float *x0, int K0, const float *x1, int K1,
                                                     const float *W, const float *bias, int N, int act,
                                                     float *out, size_t out_stride_b, int prio) {
  raise_prio(prio);
  __shared__ float red[4];
  extern __shared__ float vals[];  // N
  const int t = threadIdx.x, b = blockIdx.x;
  const int K = K0 + K1;
  for (int n = 0; n < N; ++n) {
    float s = 0.0f;
    for (int k = t; k < K; k += 256) {
      const float xv = (k < K0) ? x0[(size_t)b * K0 + k] : x1[(size_t)b * K1 + (k - K0)];
      s += xv * W[(size_t)k * N + n];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) vals[n] = red[0] + red[1] + red[2] + red[3] + (bias ? bias[n] : 0.0f);
  }
  __syncthreads();
  if (t == 0) {
    float mx = -3.0e38f, sum = 0.0f;
    if (act == 3) {
      for (int n = 0; n < N; ++n) mx = fmaxf(mx, vals[n]);
      for (int n = 0; n < N; ++n) sum += expf(vals[n] - mx);
    }
    for (int n = 0; n < N; ++n) {
      float v = vals[n];
      if (act == 1) v = fmaxf(v, 0.0f);
      else if (act == 2) v = sigm(v);
      else if (act == 3) v = expf(v - mx) / sum;
      else if (act == 4) v = tanhf(v);
      out[(size_t)b * out_stride_b + n] = v;
    }
  }
}

inline int check_desc(const ra_ctrl_desc *d) {
  if (!d) return RA_E_INVALID;
  if (d->G <= 0 || d->Cf <= 0 || d->hid <= 0 || d->iters <= 0 || d->n_gmlp < 1 || d->n_cmlp < 1 ||
      d->n_gmlp > 8 || d->n_cmlp > 8)
    return RA_E_INVALID;
  if (d->Cf % 4 || d->hid % 4 || (d->n_cmlp > 1 && d->mlp_dim % 4)) return RA_E_SHAPE;
  if (d->Cf > kThreads || 4 * d->hid > kRed || round_up(d->G, 4) > kRed || d->mlp_dim > kRed)
    return RA_E_SHAPE;
  return 0;
}

}  // namespace ctrl
}  // namespace ra

using namespace ra;

extern "C" size_t ra_ctrl_packed_floats(const ra_ctrl_desc *d) {
  if (ctrl::check_desc(d)) return 0;
  return ctrl::layout(*d).total;
}

extern "C" int ra_ctrl_pack_weights(const ra_ctrl_desc *d, const float *const *lstm_w,
                                    const float *const *gmlp_w, const float *const *cmlp_w,
                                    float *out) {
  int rc = ctrl::check_desc(d);
  if (rc || !lstm_w || !gmlp_w || !cmlp_w || !out)
    return fail(rc ? rc : RA_E_INVALID, "ra_ctrl_pack_weights: bad descriptor/argument");
  const ctrl::Layout L = ctrl::layout(*d);
  const int Cf = d->Cf, hid = d->hid, N = 4 * hid;
  // reference order i, f, u, o (nnlib.py:532-609) -> packed gate order i, f, o, u
  const int gate_of_ref[4] = {0, 1, 3, 2};
  for (int r = 0; r < 4; ++r) {
    const float *wx = lstm_w[3 * r], *wh = lstm_w[3 * r + 1], *bb = lstm_w[3 * r + 2];
    if (!wx || !wh || !bb) return fail(RA_E_INVALID, "ra_ctrl_pack_weights: null lstm weight");
    const int g = gate_of_ref[r];
    for (int k = 0; k < Cf; ++k)
      for (int j = 0; j < hid; ++j) out[L.lstm_w + (size_t)k * N + g * hid + j] = wx[(size_t)k * hid + j];
    for (int k = 0; k < hid; ++k)
      for (int j = 0; j < hid; ++j)
        out[L.lstm_w + (size_t)(Cf + k) * N + g * hid + j] = wh[(size_t)k * hid + j];
    for (int j = 0; j < hid; ++j) out[L.lstm_b + g * hid + j] = bb[j];
  }
  auto pack_mlp = [&](const float *const *w, int nl, const int *in, const int *outn, const int *outp,
                      const size_t *ow, const size_t *ob) {
    for (int l = 0; l < nl; ++l) {
      const float *ww = w[2 * l], *bb = w[2 * l + 1];
      if (!ww || !bb) return false;
      for (int k = 0; k < in[l]; ++k)
        for (int n = 0; n < outp[l]; ++n)
          out[ow[l] + (size_t)k * outp[l] + n] = (n < outn[l]) ? ww[(size_t)k * outn[l] + n] : 0.0f;
      for (int n = 0; n < outp[l]; ++n) out[ob[l] + n] = (n < outn[l]) ? bb[n] : 0.0f;
    }
    return true;
  };
  if (!pack_mlp(gmlp_w, d->n_gmlp, L.gmlp_in, L.gmlp_out, L.gmlp_outp, L.gmlp_w, L.gmlp_b) ||
      !pack_mlp(cmlp_w, d->n_cmlp, L.cmlp_in, L.cmlp_out, L.cmlp_outp, L.cmlp_w, L.cmlp_b))
    return fail(RA_E_INVALID, "ra_ctrl_pack_weights: null mlp weight");
  return 0;
}

extern "C" int ra_controller_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked, int B,
                                 float *h_last, float *ctrl_out, float *glimpse_maps, float *attn,
                                 void *stream) {
  int rc = ctrl::check_desc(d);
  if (rc) return fail(rc, "ra_controller_f32: unsupported descriptor");
  if (!feat || !wpacked || B <= 0) return fail(RA_E_INVALID, "ra_controller_f32: bad argument");
  const int Gp = round_up(d->G, 4);
  int vmax = 4 * d->hid;
  if (Gp > vmax) vmax = Gp;
  if (d->mlp_dim > vmax) vmax = d->mlp_dim;
  size_t fl = (size_t)ctrl::kRed + round_up(d->Cf + d->hid, 4) + d->hid + 2 * (size_t)vmax + Gp;
  const size_t featf = (size_t)d->G * d->Cf;
  const int in_lds = featf <= (size_t)ctrl::kMaxFeatLds;
  if (in_lds) fl += featf;
  const size_t bytes = fl * sizeof(float);
  if (bytes > 160 * 1024) return fail(RA_E_SHAPE, "ra_controller_f32: %zu B of LDS", bytes);
  static bool attr_set = false;  // idempotent; benign if raced
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ctrl::controller_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(ctrl::controller_kernel, dim3(B), dim3(ctrl::kThreads), bytes, as_stream(stream),
                     *d, feat, wpacked, h_last, ctrl_out, glimpse_maps, attn, in_lds, tail_prio(1));
  return launch_status("ra_controller_f32");
}

extern "C" int ra_dense_f32(const float *x0, int K0, const float *x1, int K1, const float *W,
                            const float *b, int B, int N, int act, float *out, size_t out_stride_b,
                            void *stream) {
  if (!x0 || !W || !out || B <= 0 || N <= 0 || K0 <= 0 || K1 < 0 || (K1 > 0 && !x1) || N > 4096)
    return fail(RA_E_INVALID, "ra_dense_f32: bad argument");
  hipLaunchKernelGGL(ctrl::dense_kernel, dim3(B), dim3(256), N * sizeof(float), as_stream(stream), x0,
                     K0, x1, K1, W, b, N, act, out, out_stride_b, tail_prio());
  return launch_status("ra_dense_f32");
}
