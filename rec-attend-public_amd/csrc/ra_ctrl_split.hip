// K2 (split form) — the controller recurrence of full_model.py:668-722 spread over kP = 16
// workgroups per image, with every weight slice STATIONARY in LDS for the whole launch.
//
// Why: the single-workgroup form (ra_ctrl.hip) re-streams 7.25 MiB of weights per image-timestep
// through one CU's L2 port (~110-130 us).  Here workgroup p of an image owns hidden units
// [p*us, (p+1)*us) of the LSTM (all four gates), the same column slice of every glimpse-MLP hidden
// layer and a slice of the G logits: 112 KiB of weights, loaded into LDS once.  Per iteration only
// the small activation vectors cross workgroups: h (256), the MLP hidden vector (256) and the
// logits (G), all-gathered through 8-byte {tag, value} granules in global memory (one relaxed
// agent-scope atomic store per value, relaxed polls, no fences: the data IS the flag).  Tags are a
// per-image generation number kept in device memory, so a HIP-graph replay needs no memset.
// The feature map lives in registers (each workgroup computes the soft-attention glimpse
// redundantly), the softmax is computed redundantly from the gathered logits.
// Results are identical in structure to the single-workgroup kernel (same products, different
// summation grouping; float32 round-off class).
#include <cstdlib>

#include "ra_common.h"

namespace ra {
namespace ctrl2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kP = 16;         // workgroups per image
constexpr int kThreads = 256;  // 4 waves
constexpr int kMaxFeatRegs = 96;
constexpr unsigned kSpinLimit = 4000000u;

struct Layout {            // per-slice packed weights (floats), identical for every slice
  int us, gs, K, NL;       // units per slice, logits per slice, Cf + hid, 4 * us
  size_t lstm_w, lstm_b;   // [K][NL], [NL]      column = gate * us + u, gates i, f, o, u
  size_t gh_w[8], gh_b[8]; // hidden glimpse-MLP layers: [hid][us], [us]
  size_t gl_w, gl_b;       // last glimpse-MLP layer:    [hid][gs], [gs]
  size_t slice;            // floats per slice
  size_t cmlp;             // offset of the (unsliced) controller MLP after kP slices
  size_t cm_w[8], cm_b[8];
  int cm_in[8], cm_out[8];
  size_t total;
  int n_hidden;
};

__host__ __device__ inline Layout layout(const ra_ctrl_desc &d) {
  Layout L;
  L.us = d.hid / kP;
  L.gs = ceil_div(d.G, kP);
  L.K = d.Cf + d.hid;
  L.NL = 4 * L.us;
  L.n_hidden = d.n_gmlp - 1;
  size_t off = 0;
  L.lstm_w = off;
  off += (size_t)L.K * L.NL;
  L.lstm_b = off;
  off += L.NL;
  for (int l = 0; l < L.n_hidden; ++l) {
    L.gh_w[l] = off;
    off += (size_t)d.hid * L.us;
    L.gh_b[l] = off;
    off += L.us;
  }
  L.gl_w = off;
  off += (size_t)d.hid * L.gs;
  L.gl_b = off;
  off += L.gs;
  L.slice = round_up((int)off, 4);
  off = L.slice * kP;
  L.cmlp = off;
  for (int l = 0; l < d.n_cmlp; ++l) {
    L.cm_in[l] = (l == 0) ? d.hid : d.mlp_dim;
    L.cm_out[l] = (l == d.n_cmlp - 1) ? 9 : d.mlp_dim;
    L.cm_w[l] = off;
    off += (size_t)L.cm_in[l] * L.cm_out[l];
    L.cm_b[l] = off;
    off += L.cm_out[l];
  }
  L.total = off;
  return L;
}

// granules per image: per iteration h (hid) [+ hidden layers (hid each) + logits (kP*gs)]
__host__ __device__ inline size_t granules_per_image(const ra_ctrl_desc &d) {
  const Layout L = layout(d);
  return (size_t)d.iters * (d.hid + (size_t)L.n_hidden * d.hid + (size_t)kP * L.gs);
}
__host__ __device__ inline size_t ws_words_per_image(const ra_ctrl_desc &d) {
  return 2 + 2 * granules_per_image(d);  // 32-bit words: [generation, pad] + 8-byte granules
}

__host__ inline int supported(const ra_ctrl_desc &d) {
  if (d.hid % kP || d.hid > kThreads || d.Cf <= 0 || kThreads % d.Cf || d.n_gmlp < 1 ||
      d.n_gmlp > 8 || d.n_cmlp < 1 || d.n_cmlp > 8 || d.G <= 0 || d.G > 4096 || d.iters <= 0 ||
      d.mlp_dim > kThreads)
    return 0;
  if (kThreads % (d.hid / kP) || kThreads % (4 * (d.hid / kP))) return 0;
  if ((size_t)d.G * d.Cf > (size_t)kMaxFeatRegs * kThreads) return 0;
  const Layout L = layout(d);
  if (4 * L.us > kThreads) return 0;
  const size_t lds = (L.slice + 4 * (size_t)kThreads + round_up(L.K, 4) + d.hid +
                      round_up(kP * L.gs, 4) + 64) * sizeof(float);
  return lds <= 160 * 1024;
}

__device__ inline void publish(u64 *g, unsigned tag, float v) {
  __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// every thread gathers values t, t+256, ... of an n-value exchange into dst (LDS)
__device__ inline void gather(const u64 *g, int n, unsigned tag, float *dst, int *err) {
  for (int i = threadIdx.x; i < n; i += kThreads) {
    unsigned spins = 0;
    u64 x;
    while (true) {
      x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(x >> 32) == tag) break;
      if (++spins > kSpinLimit) {
        *err = 1;  // a peer workgroup never arrived: report instead of hanging
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    dst[i] = __uint_as_float((unsigned)x);
  }
  __syncthreads();
}

// XCD-local exchange (the XL form of the per-image kernel): when the 16 workgroups of an image sit on ONE XCD, that XCD's L2 is
// their coherence point — a granule is a plain 8-byte store (the vector L1 writes through) and a poll is a load that only
// bypasses the CU's L1 (sc0), 0.3-0.5 us a round instead of the ~2.5 us of an agent-scope exchange through the memory fabric
// (tools/xcd_barrier_probe.hip; the 13 gathers of a timestep were 37 of the launch's 63 us).
typedef unsigned u32x2g __attribute__((ext_vector_type(2)));
__device__ inline void publish_l2(__amdgpu_buffer_rsrc_t r, int idx, unsigned tag, float v) {
  __builtin_amdgcn_raw_buffer_store_b64(u32x2g{__float_as_uint(v), tag}, r, idx * 8, 0, 0);
}
__device__ inline void gather_l2(__amdgpu_buffer_rsrc_t r, int first, int n, unsigned tag, float *dst, int *err) {
  for (int i = threadIdx.x; i < n; i += kThreads) {
    unsigned spins = 0;
    u32x2g x;
    while (true) {
      x = __builtin_amdgcn_raw_buffer_load_b64(r, (first + i) * 8, 0, (int)0x80000001u);  // sc0 (past the L1), volatile
      if (x.y == tag) break;
      if (++spins > kSpinLimit) {
        *err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    dst[i] = __uint_as_float(x.x);
  }
  __syncthreads();
}

__device__ inline float sigm(float z) { return 1.0f / (1.0f + expf(-z)); }

__device__ float block_reduce(float v, bool is_max, float *red4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(v, o);
    v = is_max ? fmaxf(v, other) : v + other;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  const float a = red4[0], b = red4[1], c = red4[2], d = red4[3];
  return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

// dot products of one weight block held in LDS: out[col] = sum_k x[k] * W[k][col], ncol columns
// (ncol <= 64 divides 256): threads = (col, k-part); result in red[part * ncol + col].
__device__ inline void slice_gemv(const float *W, int ncol, const float *x, int k0, int K, float *red) {
  const int t = threadIdx.x;
  const int parts = kThreads / ncol;
  const int col = t % ncol, part = t / ncol;
  float acc0 = 0.0f, acc1 = 0.0f;
  int k = k0 + part;
#pragma unroll 4
  for (; k + parts < K; k += 2 * parts) {
    acc0 += x[k] * W[(size_t)k * ncol + col];
    acc1 += x[k + parts] * W[(size_t)(k + parts) * ncol + col];
  }
  if (k < K) acc0 += x[k] * W[(size_t)k * ncol + col];
  red[part * ncol + col] = acc0 + acc1;
  __syncthreads();
}

// same for a block with an arbitrary column count n (<= 256), weights in LDS or global memory:
// columns are padded to a power of two; result in red[part * ncol + col], returns parts.
__device__ inline int any_gemv(const float *Wm, int n, const float *x, int K, float *red, int *ncol_out) {
  const int t = threadIdx.x;
  int ncol = 1;
  while (ncol < n) ncol <<= 1;
  const int parts = kThreads / ncol;
  const int col = t % ncol, part = t / ncol;
  float acc = 0.0f;
  if (col < n)
    for (int k = part; k < K; k += parts) acc += x[k] * Wm[(size_t)k * n + col];
  red[part * ncol + col] = acc;
  __syncthreads();
  *ncol_out = ncol;
  return parts;
}

// XL = false: grid (16, B), workgroup (p, b) = blockIdx; the image's workgroups are dealt over all eight XCDs and exchange
// through agent-scope atomics.  XL = true: a 1-D grid of 128 * ceil(B / 8) workgroups (the dispatcher deals them to the XCDs
// round robin: 16 * ceil(B / 8) each); a workgroup on XCD x draws a role from x's ticket counter (L2-local atomic; the
// counter only ever counts up, roles are tickets modulo the XCD's share) and becomes slice p of image b = x + 8 j — every
// image's team shares one L2.  Teams of images >= B leave at once.
template <int FR, bool XL = false>  // FR: feature registers per thread = ceil(G*Cf / 256)
__global__ __launch_bounds__(kThreads) void controller_split_kernel(
    const ra_ctrl_desc d, const float *feat, const float *__restrict__ wp, float *h_last,
    float *ctrl_out, float *gmaps, float *attn, unsigned *ws, int *status, int prio, int B, unsigned *tickets) {
  raise_prio(prio);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Layout L = layout(d);
  const int t = threadIdx.x;
  int p = blockIdx.x, b = blockIdx.y;
  if constexpr (XL) {
    unsigned *role_sh = reinterpret_cast<unsigned *>(smem);  // (dynamic LDS: the launch's 160 KB limit leaves no static word)
    const int x = xcc_id(), share = kP * ((B + 7) >> 3);
    if (t == 0) *role_sh = ticket_draw(tickets + x * kTicketPoolStride) % (unsigned)share;
    __syncthreads();
    const int role = (int)*role_sh;
    __syncthreads();  // read by all before the weight slice overwrites it
    p = role % kP;
    b = x + 8 * (role / kP);
    if (b >= B) return;  // (the whole team)
  }
  const int G = d.G, Cf = d.Cf, hid = d.hid, us = L.us, gs = L.gs, K = L.K;
  const int Gx = kP * gs;  // logits exchanged (>= G; tail slices hold padding)
  // LDS carve
  float *W = smem;                       // the slice
  float *red = W + L.slice;              // 4 * 256
  float *xh = red + 4 * kThreads;        // [K]  = [glimpse ; h]
  float *va = xh + round_up(K, 4);       // [hid] hidden MLP vector
  float *gm = va + hid;                  // [Gx] logits -> glimpse map
  float *red4 = gm + round_up(Gx, 4);    // small reduction scratch
  unsigned *wsb = ws + (size_t)b * ws_words_per_image(d);
  const unsigned tag = wsb[0] + 1u;
  u64 *gran = reinterpret_cast<u64 *>(wsb + 2);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, (int)(granules_per_image(d) * 8), 0x00020000);
  int err = 0;
  auto pub = [&](size_t idx, float v) {
    if constexpr (XL) publish_l2(grs, (int)idx, tag, v);
    else publish(gran + idx, tag, v);
  };
  auto gat = [&](size_t first, int n, float *dst) {
    if constexpr (XL) gather_l2(grs, (int)first, n, tag, dst, &err);
    else gather(gran + first, n, tag, dst, &err);
  };

  {  // weight slice -> LDS (stays for the whole launch)
    // batches of 16 independent 16-byte loads per thread: the fill is latency-, not
    // bandwidth-bound, so keep many loads in flight before the first LDS store
    const f32x4 *src = reinterpret_cast<const f32x4 *>(wp + (size_t)p * L.slice);
    f32x4 *dst = reinterpret_cast<f32x4 *>(W);
    const int n4 = (int)(L.slice / 4);
    constexpr int U = 16;
    for (int e0 = t; e0 < n4; e0 += kThreads * U) {
      f32x4 tmp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        tmp[u] = src[e < n4 ? e : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        if (e < n4) dst[e] = tmp[u];
      }
    }
  }
  // feature map -> registers: thread (lane = channel group, wave = position phase)
  // element index e = t + 256 * i  ->  (g, c) = (e / Cf, e % Cf); Cf divides 256 or 64 | Cf...
  float fr[FR];
  const float *fsrc = feat + (size_t)b * G * Cf;
#pragma unroll
  for (int i = 0; i < FR; ++i) {
    const int e = t + kThreads * i;
    fr[i] = (e < G * Cf) ? fsrc[e] : 0.0f;
  }
  for (int e = t; e < hid; e += kThreads) xh[Cf + e] = 0.0f;
  for (int g = t; g < Gx; g += kThreads) gm[g] = (g < G) ? 1.0f / (float)G : 0.0f;
  float cst = 0.0f;  // cell state of unit (p*us + t), threads t < us
  __syncthreads();

  size_t goff = 0;
  for (int it = 0; it < d.iters; ++it) {
    if (gmaps && p == 0)
      for (int g = t; g < G; g += kThreads) gmaps[((size_t)b * d.iters + it) * G + g] = gm[g];
    // ---- glimpse[c] = sum_g feat[g,c] * map[g]  (every workgroup, from registers) ----
    {
      // e = t + 256*i ; since Cf divides 256, channel c = t % Cf is fixed per thread and
      // g = t / Cf + (256 / Cf) * i
      const int g0 = t / Cf, gstep = kThreads / Cf;
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < FR; ++i) {
        const int g = g0 + gstep * i;
        s += fr[i] * ((g < G) ? gm[g] : 0.0f);
      }
      red[t] = s;
      __syncthreads();
      if (t < Cf) {
        float a = 0.0f;
        for (int q = 0; q < gstep; ++q) a += red[q * Cf + t];
        xh[t] = a;
      }
      __syncthreads();
    }
    // ---- LSTM slice (nnlib.py:641-646) ----
    slice_gemv(W + L.lstm_w, L.NL, xh, 0, (it == 0) ? Cf : K, red);  // h == 0 at it == 0
    if (t < us) {
      const int parts = kThreads / L.NL;
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float a = W[L.lstm_b + g * us + t];
        for (int q = 0; q < parts; ++q) a += red[q * L.NL + g * us + t];
        pre[g] = a;
      }
      const float gi = sigm(pre[0]), gf = sigm(pre[1]), go = sigm(pre[2]), u = tanhf(pre[3]);
      cst = gf * cst + gi * u;
      pub(goff + p * us + t, go * tanhf(cst));
    }
    gat(goff, hid, xh + Cf);
    goff += hid;
    if (it == d.iters - 1) break;
    // ---- glimpse MLP hidden layers (relu) ----
    const float *in = xh + Cf;
    for (int l = 0; l < L.n_hidden; ++l) {
      slice_gemv(W + L.gh_w[l], us, in, 0, hid, red);
      if (t < us) {
        const int parts = kThreads / us;
        float a = W[L.gh_b[l] + t];
        for (int q = 0; q < parts; ++q) a += red[q * us + t];
        pub(goff + p * us + t, fmaxf(a, 0.0f));
      }
      gat(goff, hid, va);
      goff += hid;
      in = va;
    }
    // ---- logits slice, gathered, softmax over G (redundantly) ----
    {
      int ncol;
      const int parts = any_gemv(W + L.gl_w, gs, in, hid, red, &ncol);
      if (t < gs) {
        float a = W[L.gl_b + t];
        for (int q = 0; q < parts; ++q) a += red[q * ncol + t];
        pub(goff + p * gs + t, a);
      }
      gat(goff, Gx, gm);
      goff += Gx;
      float mx = -3.0e38f;
      for (int g = t; g < G; g += kThreads) mx = fmaxf(mx, gm[g]);
      mx = block_reduce(mx, true, red4);
      float sum = 0.0f;
      for (int g = t; g < G; g += kThreads) sum += expf(gm[g] - mx);
      sum = block_reduce(sum, false, red4);
      __syncthreads();
      for (int g = t; g < Gx; g += kThreads) gm[g] = (g < G) ? expf(gm[g] - mx) / sum : 0.0f;
      __syncthreads();
    }
  }

  if (p == 0) {
    // ---- controller MLP + attention decode (workgroup 0 of the image; weights from L2) ----
    const float *in = xh + Cf;
    float *o1 = va, *o2 = gm;
    for (int l = 0; l < d.n_cmlp; ++l) {
      const int N = L.cm_out[l], Kin = L.cm_in[l];
      const float *Wc = wp + L.cm_w[l], *bc = wp + L.cm_b[l];
      const bool last = (l == d.n_cmlp - 1);
      int ncol;
      const int parts = any_gemv(Wc, N, in, Kin, red, &ncol);
      if (t < N) {
        float a = bc[t];
        for (int q = 0; q < parts; ++q) a += red[q * ncol + t];
        o1[t] = last ? a : fmaxf(a, 0.0f);
      }
      __syncthreads();
      in = o1;
      float *tmp = o1;
      o1 = o2;
      o2 = tmp;
    }
    const float *co = in;
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
        const float ctr = (cn[k] + 1.0f) * (dim[k] / 2.0f);
        const float size = expf(ls[k]) * dim[k];
        float lv = d.fixed_var ? 0.0f : logf(size) - logf(fs[k]);
        if (d.dynamic_var) lv = co[4 + k];
        r[0 + k] = ctr;
        r[2 + k] = size;
        r[4 + k] = lv;
        r[9 + k] = cn[k];
        r[11 + k] = ls[k];
      }
      r[6] = d.fixed_gamma ? 1.0f : expf(co[6]);
      r[7] = expf(co[7]);
      r[8] = d.fixed_gamma ? 2.0f : co[8];
      r[13] = r[14] = r[15] = 0.0f;
    }
    // new generation for the next launch: every peer of this image has read ws[0] long ago (this
    // workgroup could not have finished its last gather otherwise)
    if (t == 0)
      __hip_atomic_store(wsb, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (err && status) atomicMax(status, 1);
}

template <int FR>
int launch(const ra_ctrl_desc &d, const float *feat, const float *wp, int B, float *h_last,
           float *ctrl_out, float *gmaps, float *attn, unsigned *ws, size_t ws_bytes, int *status, size_t lds,
           hipStream_t st) {
  auto kern = controller_split_kernel<FR, false>;
  auto kern_xl = controller_split_kernel<FR, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern_xl),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // the XCD-local form wherever the device's workgroups report the XCC ids 0..7 (ra_core.hip's census); RA_CTRL_XCD=0: the old one
  static int xl = -1, xl_all = 0;
  if (xl < 0) {
    const char *e = getenv("RA_CTRL_XCD");
    xl_all = (e && atoi(e) == 2) ? 1 : 0;
    if (e && atoi(e) == 0) xl = 0;
    else if (const int c = xcc_census_ok(); c >= 0) xl = c;  // (-1: asked inside a stream capture — decide at the next launch)
  }
  // 8 pools of kTicketPoolStride words at the END of the caller's workspace (not behind THIS launch's images: a workspace sized for
  // more images than it is launched with keeps its granules and its role tickets apart)
  unsigned *tickets = ws + ws_bytes / 4 - 8 * kTicketPoolStride;
  // up to 8 images: one team per XCD, 16 of its 32 CUs — as much headroom as the grid (16, B) form has on the whole chip.  9-14
  // images would put two teams on some XCDs and need ALL their CUs: those launches keep the grid (16, B) form (RA_CTRL_XCD=2: both)
  if (xl == 1 && (B <= 8 || xl_all))
    hipLaunchKernelGGL(kern_xl, dim3(8 * kP * ((B + 7) / 8)), dim3(kThreads), lds, st, d, feat, wp, h_last, ctrl_out, gmaps, attn, ws,
                       status, tail_prio(1), B, tickets);
  else
    hipLaunchKernelGGL(kern, dim3(kP, B), dim3(kThreads), lds, st, d, feat, wp, h_last, ctrl_out,
                       gmaps, attn, ws, status, tail_prio(1), B, tickets);
  return launch_status("ra_controller_split_f32");
}

// -------------------------------------------------------------------------------------------------
// K2b — the same recurrence with the 16 weight slices shared by a GROUP of up to NI images: 16 workgroups
// per group instead of 16 per image.  A launch of 8 images is 32 workgroups, so four decode pipelines'
// controllers (or cfg3's 16-image batches: two groups of 8) are resident together many times over — the
// co-residency that the per-image form cannot guarantee beside its own kind — and every
// weight is read from LDS once for all images of the group.  Per glimpse iteration the workgroups
// exchange the glimpses (each image's soft-attention read-out is computed by ONE workgroup, from the
// feature map it keeps in registers), h, the MLP hidden vector and the logits: 4 all-gathers of
// {tag, value} granules instead of 3.
// Images per group, NI: 4 for launches of up to 8 images, 8 above (ra_ctrl_batch_group_images()), so a launch is at
// most 32 workgroups up to 16 images.  At cfg2 a launch takes 88.9 us with groups of 4 and 112.6 us with groups of 8;
// once the encoder launches had been made to share the CUs, the shorter tail showed in the pipelined rate (51.8k vs
// 51.2k instance-timesteps/s, 8 batches on 4 streams).  KITTI's 16-image batches keep groups of 8: with six parts in
// flight, four groups each would not fit the co-residency margin.
template <int NI>
__host__ __device__ inline size_t granules_per_group(const ra_ctrl_desc &d) {
  const Layout L = layout(d);
  return (size_t)d.iters * NI * (d.Cf + d.hid + (size_t)L.n_hidden * d.hid + (size_t)kP * L.gs);
}
template <int NI>
__host__ __device__ inline size_t ws_words_per_group(const ra_ctrl_desc &d) { return 2 + 2 * granules_per_group<NI>(d); }

template <int NI>
__host__ inline size_t batch_lds_bytes(const ra_ctrl_desc &d) {
  const Layout L = layout(d);
  return (L.slice + (size_t)kThreads * NI + (size_t)NI * round_up(L.K, 4) + (size_t)NI * d.hid +
          (size_t)NI * round_up(kP * L.gs, 4) + 64) * sizeof(float);
}

template <int NI>
__host__ inline int batch_supported(const ra_ctrl_desc &d) {
  if (!supported(d)) return 0;
  const Layout L = layout(d);
  int gsp = 1;
  while (gsp < L.gs) gsp <<= 1;
  if (gsp > 64 || L.us * NI > kThreads || d.Cf > kThreads) return 0;
  return batch_lds_bytes<NI>(d) <= 160 * 1024;
}

// n values per image, images i < nimg: granule (i * n + j) -> dst[i * stride + j].  A thread owns up to 8
// granules per pass and polls them TOGETHER (all loads in flight, then the tag checks): polled one after
// the other, eight L2 round trips per gather made the kernel slower than the one-workgroup form.
__device__ inline void gather_multi(const u64 *g, int n, int nimg, unsigned tag, float *dst, int stride, int *err) {
  const int total = n * nimg;
  for (int base = 0; base < total; base += kThreads * 8) {
    unsigned pending = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (base + threadIdx.x + kThreads * j < total) pending |= 1u << j;
    unsigned spins = 0;
    while (pending) {
      u64 x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (pending & (1u << j))
          x[j] = __hip_atomic_load(g + base + threadIdx.x + kThreads * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((pending & (1u << j)) && (unsigned)(x[j] >> 32) == tag) {
          const int e = base + threadIdx.x + kThreads * j;
          const int i = e / n, c = e - i * n;
          dst[i * stride + c] = __uint_as_float((unsigned)x[j]);
          pending &= ~(1u << j);
        }
      if (pending) {
        if (++spins > kSpinLimit) {
          *err = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

// ... the same through the XCD's L2 (the XL form of K2b, round 6): granules are plain 8-byte stores of the producer, polled
// with loads that bypass this CU's L1 — as gather_l2 above, eight in flight per thread.
__device__ inline void gather_multi_l2(__amdgpu_buffer_rsrc_t r, size_t first, int n, int nimg, unsigned tag, float *dst, int stride, int *err) {
  const int total = n * nimg;
  for (int base = 0; base < total; base += kThreads * 8) {
    unsigned pending = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (base + threadIdx.x + kThreads * j < total) pending |= 1u << j;
    unsigned spins = 0;
    while (pending) {
      u32x2g x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (pending & (1u << j))
          x[j] = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(first + base + threadIdx.x + kThreads * j) * 8, 0, (int)0x80000001u);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((pending & (1u << j)) && x[j].y == tag) {
          const int e = base + threadIdx.x + kThreads * j;
          const int i = e / n, c = e - i * n;
          dst[i * stride + c] = __uint_as_float(x[j].x);
          pending &= ~(1u << j);
        }
      if (pending) {
        if (++spins > kSpinLimit) {
          *err = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

// out[i][col] = sum_k x[i][k] * W[k][col] for the images of the group; ncol a power of two <= 64, n real columns;
// threads = (col, k-part), a part walks quads of consecutive k (one 16-byte LDS read of x per image and quad);
// xs and K multiples of 4 (K: the tail is zero-padded by the caller's layout or handled below);
// result in red[(part * NI + i) * ncol + col], returns parts.
template <int NI>
__device__ inline int gemv_multi(const float *W, int n, int ncol, const float *x, int xs, int K, float *red) {
  const int t = threadIdx.x, parts = kThreads / ncol;
  const int col = t % ncol, part = t / ncol;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.0f;
  if (col < n) {
    const int K4 = K & ~3;
    for (int k = 4 * part; k < K4; k += 4 * parts) {
      const float w0 = W[(size_t)k * n + col], w1 = W[(size_t)(k + 1) * n + col], w2 = W[(size_t)(k + 2) * n + col],
                  w3 = W[(size_t)(k + 3) * n + col];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + i * xs + k);
        acc[i] += xv.x * w0 + xv.y * w1 + xv.z * w2 + xv.w * w3;
      }
    }
    if (part == 0)
      for (int k = K4; k < K; ++k) {
        const float w = W[(size_t)k * n + col];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] += x[i * xs + k] * w;
      }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) red[(part * NI + i) * ncol + col] = acc[i];
  __syncthreads();
  return parts;
}

// XL = true (round 6): the XCD-local form, as controller_split_kernel's — a 1-D grid of 128 workgroups, 16 per XCD; a workgroup
// on XCD x draws its slice p from x's role tickets and serves group (x - xcd_off) mod 8, so a group's 16 workgroups share one
// L2 and exchange through it (plain stores, L1-bypassing polls) instead of agent-scope atomics; XCDs without a group leave at
// once.  At most 8 groups per launch; concurrent launches are given different xcd_off by the caller (one team per XCD).
template <int FR, int NI, bool XL = false>
__global__ __launch_bounds__(kThreads) void controller_batch_kernel(const ra_ctrl_desc d, const float *feat,
                                                                    const float *__restrict__ wp, int B, float *h_last,
                                                                    float *ctrl_out, float *gmaps, float *attn,
                                                                    unsigned *ws, int *status, int prio, unsigned *tickets, int xcd_off) {
  raise_prio(prio);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Layout L = layout(d);
  const int t = threadIdx.x;
  int p = blockIdx.x, grp = blockIdx.y;
  if constexpr (XL) {
    unsigned *role_sh = reinterpret_cast<unsigned *>(smem);
    const int x = xcc_id();
    if (t == 0) *role_sh = ticket_draw(tickets + x * kTicketPoolStride) % (unsigned)kP;
    __syncthreads();
    p = (int)*role_sh;
    __syncthreads();  // read by all before the weight slice overwrites it
    grp = (x - xcd_off) & 7;
    if (grp * NI >= B) return;  // (the whole team)
  }
  const int G = d.G, Cf = d.Cf, hid = d.hid, us = L.us, gs = L.gs, K = L.K;
  const int Gx = kP * gs, Kp = round_up(K, 4), Gxp = round_up(Gx, 4);
  const int b0 = grp * NI, nimg = (B - b0 < NI) ? B - b0 : NI;
  float *W = smem;                          // the slice
  float *red = W + L.slice;                 // [256 * NI]
  float *xh = red + kThreads * NI;         // [NI][Kp]  = [glimpse ; h] per image
  float *va = xh + NI * Kp;                // [NI][hid] hidden MLP vector
  float *gm = va + NI * hid;               // [NI][Gxp] logits -> glimpse map
  unsigned *wsg = ws + (size_t)grp * ws_words_per_group<NI>(d);
  const unsigned tag = wsg[0] + 1u;
  u64 *gran = reinterpret_cast<u64 *>(wsg + 2);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(gran, 0, (int)(granules_per_group<NI>(d) * 8), 0x00020000);
  int err = 0;
  auto publish = [&](u64 *g, unsigned tg, float v) {  // (shadows the agent-scope helper: g is an address inside `gran`)
    if constexpr (XL) publish_l2(grs, (int)(g - gran), tg, v);
    else ctrl2::publish(g, tg, v);
  };
  auto gather_multi = [&](const u64 *g, int n, int nimg_, unsigned tg, float *dst, int stride, int *e) {
    if constexpr (XL) gather_multi_l2(grs, (size_t)(g - gran), n, nimg_, tg, dst, stride, e);
    else ctrl2::gather_multi(g, n, nimg_, tg, dst, stride, e);
  };

  {  // weight slice -> LDS (stays for the whole launch)
    const f32x4 *src = reinterpret_cast<const f32x4 *>(wp + (size_t)p * L.slice);
    f32x4 *dst = reinterpret_cast<f32x4 *>(W);
    const int n4 = (int)(L.slice / 4);
    constexpr int U = 16;
    for (int e0 = t; e0 < n4; e0 += kThreads * U) {
      f32x4 tmp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        tmp[u] = src[e < n4 ? e : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * kThreads;
        if (e < n4) dst[e] = tmp[u];
      }
    }
  }
  // this workgroup reads out the glimpse of image `mine` (the first nimg workgroups publish theirs)
  const int mine = p % nimg;
  float fr[FR];
  const float *fsrc = feat + (size_t)(b0 + mine) * G * Cf;
#pragma unroll
  for (int i = 0; i < FR; ++i) {
    const int e = t + kThreads * i;
    fr[i] = (e < G * Cf) ? fsrc[e] : 0.0f;
  }
  for (int e = t; e < NI * Kp; e += kThreads) xh[e] = 0.0f;  // h = 0 (and defined glimpse slots)
  for (int e = t; e < NI * Gxp; e += kThreads) gm[e] = ((e % Gxp) < G) ? 1.0f / (float)G : 0.0f;
  float cst = 0.0f;  // cell state of (image t / us, unit p * us + t % us), threads t < us * NI
  const int ci = t / us, cu = t % us;
  __syncthreads();

  size_t goff = 0;
  for (int it = 0; it < d.iters; ++it) {
    if (gmaps && p == 0)
      for (int e = t; e < nimg * G; e += kThreads) {
        const int i = e / G, g = e - i * G;
        gmaps[((size_t)(b0 + i) * d.iters + it) * G + g] = gm[i * Gxp + g];
      }
    {  // ---- glimpse of image `mine` from registers ----
      const int g0 = t / Cf, gstep = kThreads / Cf;
      const float *gmm = gm + mine * Gxp;
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < FR; ++i) {
        const int g = g0 + gstep * i;
        s += fr[i] * ((g < G) ? gmm[g] : 0.0f);
      }
      red[t] = s;
      __syncthreads();
      if (t < Cf && p < nimg) {
        float a = 0.0f;
        for (int q = 0; q < gstep; ++q) a += red[q * Cf + t];
        publish(gran + goff + (size_t)mine * Cf + t, tag, a);
      }
      __syncthreads();
    }
    gather_multi(gran + goff, Cf, nimg, tag, xh, Kp, &err);
    goff += (size_t)NI * Cf;
    // ---- LSTM slice, all images ----
    {
      const int parts = gemv_multi<NI>(W + L.lstm_w, L.NL, L.NL, xh, Kp, (it == 0) ? Cf : K, red);
      if (t < us * NI && ci < nimg) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float a = W[L.lstm_b + g * us + cu];
          for (int q = 0; q < parts; ++q) a += red[(q * NI + ci) * L.NL + g * us + cu];
          pre[g] = a;
        }
        const float gi = sigm(pre[0]), gf = sigm(pre[1]), go = sigm(pre[2]), u = tanhf(pre[3]);
        cst = gf * cst + gi * u;
        publish(gran + goff + (size_t)ci * hid + p * us + cu, tag, go * tanhf(cst));
      }
    }
    gather_multi(gran + goff, hid, nimg, tag, xh + Cf, Kp, &err);
    goff += (size_t)NI * hid;
    if (it == d.iters - 1) break;
    // ---- glimpse MLP hidden layers (relu) ----
    const float *in = xh + Cf;
    int ins = Kp;
    for (int l = 0; l < L.n_hidden; ++l) {
      const int parts = gemv_multi<NI>(W + L.gh_w[l], us, us, in, ins, hid, red);
      if (t < us * NI && ci < nimg) {
        float a = W[L.gh_b[l] + cu];
        for (int q = 0; q < parts; ++q) a += red[(q * NI + ci) * us + cu];
        publish(gran + goff + (size_t)ci * hid + p * us + cu, tag, fmaxf(a, 0.0f));
      }
      gather_multi(gran + goff, hid, nimg, tag, va, hid, &err);
      goff += (size_t)NI * hid;
      in = va;
      ins = hid;
    }
    {  // ---- logits slice, gathered; softmax over G per image, two images per wave ----
      int ncol = 1;
      while (ncol < gs) ncol <<= 1;
      const int parts = gemv_multi<NI>(W + L.gl_w, gs, ncol, in, ins, hid, red);
      const int li = t / ncol, lu = t % ncol;  // (image, logit) for the first ncol * NI threads
      if (t < ncol * NI && li < nimg && lu < gs) {
        float a = W[L.gl_b + lu];
        for (int q = 0; q < parts; ++q) a += red[(q * NI + li) * ncol + lu];
        publish(gran + goff + (size_t)li * Gx + p * gs + lu, tag, a);
      }
      gather_multi(gran + goff, Gx, nimg, tag, gm, Gxp, &err);
      goff += (size_t)NI * Gx;
      const int wave = t >> 6, lane = t & 63;
      for (int i = wave; i < nimg; i += kThreads / 64) {
        float *gi = gm + i * Gxp;
        float mx = -3.0e38f;
        for (int g = lane; g < G; g += 64) mx = fmaxf(mx, gi[g]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.0f;
        for (int g = lane; g < G; g += 64) sum += expf(gi[g] - mx);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        for (int g = lane; g < Gx; g += 64) gi[g] = (g < G) ? expf(gi[g] - mx) / sum : 0.0f;
      }
      __syncthreads();
    }
  }

  if (p < nimg) {
    // ---- controller MLP + attention decode of image p (weights from L2) ----
    const int b = b0 + p;
    const float *in = xh + p * Kp + Cf;
    float *o1 = va + p * hid, *o2 = red;  // va row p is this workgroup's own; red is free now
    for (int l = 0; l < d.n_cmlp; ++l) {
      const int N = L.cm_out[l], Kin = L.cm_in[l];
      const float *Wc = wp + L.cm_w[l], *bc = wp + L.cm_b[l];
      const bool last = (l == d.n_cmlp - 1);
      int ncol;
      float *scratch = red + 2 * kThreads;  // one partial per thread; red is [256 * NI], o2 uses its first hid <= 256 floats
      const int parts = any_gemv(Wc, N, in, Kin, scratch, &ncol);
      if (t < N) {
        float a = bc[t];
        for (int q = 0; q < parts; ++q) a += scratch[q * ncol + t];
        o1[t] = last ? a : fmaxf(a, 0.0f);
      }
      __syncthreads();
      in = o1;
      float *tmp = o1;
      o1 = o2;
      o2 = tmp;
    }
    const float *co = in;
    if (t < hid && h_last) h_last[(size_t)b * hid + t] = xh[p * Kp + Cf + t];
    if (t < 9 && ctrl_out) ctrl_out[(size_t)b * 9 + t] = co[t];
    if (t == 0 && attn) {
      float *r = attn + (size_t)b * RA_ATTN_STRIDE;
      float cn[2] = {co[0], co[1]}, ls[2] = {co[2], co[3]};
      if (d.squash) {
        cn[0] = tanhf(cn[0]);
        cn[1] = tanhf(cn[1]);
        ls[0] = -log1pf(expf(ls[0]));
        ls[1] = -log1pf(expf(ls[1]));
      }
      const float dim[2] = {(float)d.H, (float)d.W}, fs[2] = {(float)d.Fh, (float)d.Fw};
      for (int k = 0; k < 2; ++k) {
        const float ctr = (cn[k] + 1.0f) * (dim[k] / 2.0f);
        const float size = expf(ls[k]) * dim[k];
        float lv = d.fixed_var ? 0.0f : logf(size) - logf(fs[k]);
        if (d.dynamic_var) lv = co[4 + k];
        r[0 + k] = ctr;
        r[2 + k] = size;
        r[4 + k] = lv;
        r[9 + k] = cn[k];
        r[11 + k] = ls[k];
      }
      r[6] = d.fixed_gamma ? 1.0f : expf(co[6]);
      r[7] = expf(co[7]);
      r[8] = d.fixed_gamma ? 2.0f : co[8];
      r[13] = r[14] = r[15] = 0.0f;
    }
  }
  // new generation for the next launch: workgroup 0 could not have finished its last gather unless every peer
  // had published with this tag, i.e. had read wsg[0]
  if (p == 0 && t == 0) __hip_atomic_store(wsg, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (err && status) atomicMax(status, 1);
}

template <int FR, int NI>
int launch_batch(const ra_ctrl_desc &d, const float *feat, const float *wp, int B, float *h_last, float *ctrl_out,
                 float *gmaps, float *attn, unsigned *ws, size_t ws_bytes, int *status, size_t lds, int xcd_off, hipStream_t st) {
  auto kern = controller_batch_kernel<FR, NI, false>;
  auto kern_xl = controller_batch_kernel<FR, NI, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern_xl), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // the XCD-local form: only when the caller names an XCD offset (xcd_off >= 0: it vouches that concurrent launches use others),
  // the launch has at most 8 groups, and the device's workgroups report the XCC ids 0..7 (RA_CTRL_XCD=0: never)
  static int xl = -1;
  if (xl < 0) {
    const char *e = getenv("RA_CTRL_XCD");
    if (e && atoi(e) == 0) xl = 0;
    else if (const int c = xcc_census_ok(); c >= 0) xl = c;
  }
  unsigned *tickets = ws + ws_bytes / 4 - 8 * kTicketPoolStride;
  if (xl == 1 && xcd_off >= 0 && ceil_div(B, NI) <= 8)
    hipLaunchKernelGGL(kern_xl, dim3(8 * kP), dim3(kThreads), lds, st, d, feat, wp, B, h_last, ctrl_out, gmaps, attn, ws, status,
                       tail_prio(1), tickets, xcd_off & 7);
  else
    hipLaunchKernelGGL(kern, dim3(kP, ceil_div(B, NI)), dim3(kThreads), lds, st, d, feat, wp, B, h_last, ctrl_out, gmaps,
                       attn, ws, status, tail_prio(1), tickets, 0);
  return launch_status("ra_controller_batch_f32");
}

// the group size a launch of B images uses (groups of 8 only where their larger LDS footprint fits)
inline int group_images(const ra_ctrl_desc &d, int B) { return (B > 8 && batch_supported<8>(d)) ? 8 : 4; }

}  // namespace ctrl2
}  // namespace ra

using namespace ra;

extern "C" int ra_ctrl_split_supported(const ra_ctrl_desc *d) { return d ? ctrl2::supported(*d) : 0; }

extern "C" size_t ra_ctrl_split_packed_floats(const ra_ctrl_desc *d) {
  if (!d || !ctrl2::supported(*d)) return 0;
  return ctrl2::layout(*d).total;
}

extern "C" size_t ra_ctrl_split_workspace_bytes(const ra_ctrl_desc *d, int B) {
  if (!d || B <= 0 || !ctrl2::supported(*d)) return 0;
  return ((size_t)B * ctrl2::ws_words_per_image(*d) + 8 * kTicketPoolStride) * 4;  // + the XCD-local form's role tickets
}

extern "C" int ra_ctrl_split_pack_weights(const ra_ctrl_desc *d, const float *const *lstm_w,
                                          const float *const *gmlp_w, const float *const *cmlp_w,
                                          float *out) {
  if (!d || !lstm_w || !gmlp_w || !cmlp_w || !out || !ctrl2::supported(*d))
    return fail(RA_E_SHAPE, "ra_ctrl_split_pack_weights: unsupported descriptor / null argument");
  const ctrl2::Layout L = ctrl2::layout(*d);
  const int Cf = d->Cf, hid = d->hid, us = L.us, gs = L.gs;
  for (size_t i = 0; i < L.total; ++i) out[i] = 0.0f;
  const int gate_of_ref[4] = {0, 1, 3, 2};  // reference order i, f, u, o -> packed i, f, o, u
  for (int p = 0; p < ctrl2::kP; ++p) {
    float *S = out + (size_t)p * L.slice;
    for (int r = 0; r < 4; ++r) {
      const float *wx = lstm_w[3 * r], *wh = lstm_w[3 * r + 1], *bb = lstm_w[3 * r + 2];
      const int g = gate_of_ref[r];
      for (int u = 0; u < us; ++u) {
        const int j = p * us + u, col = g * us + u;
        for (int k = 0; k < Cf; ++k) S[L.lstm_w + (size_t)k * L.NL + col] = wx[(size_t)k * hid + j];
        for (int k = 0; k < hid; ++k) S[L.lstm_w + (size_t)(Cf + k) * L.NL + col] = wh[(size_t)k * hid + j];
        S[L.lstm_b + col] = bb[j];
      }
    }
    for (int l = 0; l < L.n_hidden; ++l) {
      const float *ww = gmlp_w[2 * l], *bb = gmlp_w[2 * l + 1];
      for (int u = 0; u < us; ++u) {
        for (int k = 0; k < hid; ++k) S[L.gh_w[l] + (size_t)k * us + u] = ww[(size_t)k * hid + p * us + u];
        S[L.gh_b[l] + u] = bb[p * us + u];
      }
    }
    const float *ww = gmlp_w[2 * L.n_hidden], *bb = gmlp_w[2 * L.n_hidden + 1];
    for (int u = 0; u < gs; ++u) {
      const int g = p * gs + u;
      if (g >= d->G) continue;
      for (int k = 0; k < hid; ++k) S[L.gl_w + (size_t)k * gs + u] = ww[(size_t)k * d->G + g];
      S[L.gl_b + u] = bb[g];
    }
  }
  for (int l = 0; l < d->n_cmlp; ++l) {
    const float *ww = cmlp_w[2 * l], *bb = cmlp_w[2 * l + 1];
    for (size_t i = 0; i < (size_t)L.cm_in[l] * L.cm_out[l]; ++i) out[L.cm_w[l] + i] = ww[i];
    for (int i = 0; i < L.cm_out[l]; ++i) out[L.cm_b[l] + i] = bb[i];
  }
  return 0;
}

extern "C" int ra_controller_split_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked,
                                       int B, float *h_last, float *ctrl_out, float *glimpse_maps,
                                       float *attn, void *ws, size_t ws_bytes, int *status_dev,
                                       void *stream) {
  if (!d || !feat || !wpacked || !ws || B <= 0) return fail(RA_E_INVALID, "ra_controller_split_f32: bad argument");
  if (!ctrl2::supported(*d)) return fail(RA_E_SHAPE, "ra_controller_split_f32: unsupported descriptor");
  if (B * ctrl2::kP > 224) return fail(RA_E_SHAPE, "ra_controller_split_f32: B=%d exceeds co-residency (14)", B);
  if (ws_bytes < ra_ctrl_split_workspace_bytes(d, B)) return fail(RA_E_WORKSPACE, "ra_controller_split_f32: workspace");
  const ctrl2::Layout L = ctrl2::layout(*d);
  const size_t lds = (L.slice + 4 * (size_t)ctrl2::kThreads + round_up(L.K, 4) + d->hid +
                      round_up(ctrl2::kP * L.gs, 4) + 64) * sizeof(float);
  const int fr = ceil_div(d->G * d->Cf, ctrl2::kThreads);
  hipStream_t st = as_stream(stream);
  unsigned *w = reinterpret_cast<unsigned *>(ws);
#define RA_C2(FR) return ctrl2::launch<FR>(*d, feat, wpacked, B, h_last, ctrl_out, glimpse_maps, attn, w, ws_bytes, status_dev, lds, st)
  if (fr <= 4) RA_C2(4);
  if (fr <= 16) RA_C2(16);
  if (fr <= 32) RA_C2(32);
  if (fr <= 64) RA_C2(64);
  RA_C2(96);
#undef RA_C2
}

// ---- K2b: one group of 16 workgroups per 4 or 8 images (weights packed as for ra_controller_split_f32) ----
extern "C" int ra_ctrl_batch_supported(const ra_ctrl_desc *d) { return d ? ctrl2::batch_supported<4>(*d) : 0; }

extern "C" int ra_ctrl_batch_group_images(const ra_ctrl_desc *d, int B) {
  if (!d || B <= 0 || !ctrl2::batch_supported<4>(*d)) return 0;
  return ctrl2::group_images(*d, B);
}

extern "C" size_t ra_ctrl_batch_workspace_bytes(const ra_ctrl_desc *d, int B) {
  if (!d || B <= 0 || !ctrl2::batch_supported<4>(*d)) return 0;
  const int g = ctrl2::group_images(*d, B);
  // (+ the role tickets of the XCD-local form, at the end: 8 pools of one 128-byte line)
  return ((size_t)ceil_div(B, g) * (g == 8 ? ctrl2::ws_words_per_group<8>(*d) : ctrl2::ws_words_per_group<4>(*d)) + 8 * kTicketPoolStride) * 4;
}

extern "C" int ra_controller_batch_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked, int B,
                                       float *h_last, float *ctrl_out, float *glimpse_maps, float *attn, void *ws,
                                       size_t ws_bytes, int *status_dev, void *stream) {
  return ra_controller_batch_xcd_f32(d, feat, wpacked, B, h_last, ctrl_out, glimpse_maps, attn, ws, ws_bytes, status_dev, -1, stream);
}

extern "C" int ra_controller_batch_xcd_f32(const ra_ctrl_desc *d, const float *feat, const float *wpacked, int B,
                                           float *h_last, float *ctrl_out, float *glimpse_maps, float *attn, void *ws,
                                           size_t ws_bytes, int *status_dev, int xcd_offset, void *stream) {
  if (!d || !feat || !wpacked || !ws || B <= 0) return fail(RA_E_INVALID, "ra_controller_batch_f32: bad argument");
  if (!ctrl2::batch_supported<4>(*d)) return fail(RA_E_SHAPE, "ra_controller_batch_f32: unsupported descriptor");
  const int g = ctrl2::group_images(*d, B);
  if (ceil_div(B, g) * ctrl2::kP > 224)
    return fail(RA_E_SHAPE, "ra_controller_batch_f32: B=%d exceeds co-residency (%d)", B, 14 * g);
  if (ws_bytes < ra_ctrl_batch_workspace_bytes(d, B)) return fail(RA_E_WORKSPACE, "ra_controller_batch_f32: workspace");
  const size_t lds = g == 8 ? ctrl2::batch_lds_bytes<8>(*d) : ctrl2::batch_lds_bytes<4>(*d);
  const int fr = ceil_div(d->G * d->Cf, ctrl2::kThreads);
  hipStream_t st = as_stream(stream);
  unsigned *w = reinterpret_cast<unsigned *>(ws);
#define RA_C3(FR)                                                                                                        \
  return g == 8 ? ctrl2::launch_batch<FR, 8>(*d, feat, wpacked, B, h_last, ctrl_out, glimpse_maps, attn, w, ws_bytes, status_dev, lds, xcd_offset, st) \
                : ctrl2::launch_batch<FR, 4>(*d, feat, wpacked, B, h_last, ctrl_out, glimpse_maps, attn, w, ws_bytes, status_dev, lds, xcd_offset, st)
  if (fr <= 4) RA_C3(4);
  if (fr <= 16) RA_C3(16);
  if (fr <= 32) RA_C3(32);
  if (fr <= 64) RA_C3(64);
  RA_C3(96);
#undef RA_C3
}
