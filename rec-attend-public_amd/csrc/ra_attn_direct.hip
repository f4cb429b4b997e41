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

#include "ra_attn_axis.h"
#include "ra_common.h"

namespace ra {
namespace attnd {

// tools/attn_probe.hip builds this file with -DRA_PROBE: thread 0 of every workgroup stamps the 100 MHz
// wall clock at a few points of the kernel (where the time of a latency-bound kernel goes)
#ifdef RA_PROBE
__device__ long long *ra_probe_buf;
#define RA_PROBE_AT(k)                                                                              \
  do {                                                                                              \
    if (threadIdx.x == 0)                                                                           \
      ra_probe_buf[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define RA_PROBE_AT(k)
#endif

// ---- extract, row-parallel form ---------------------------------------------------------------------
// Workgroup = (one tap j, channel group cg, image b).  The tap's row band (2R+1 rows) x the window's
// columns is read in rounds of 16 rows: lane = image column (a wave reads 1 KB runs), wave w takes rows
// w, w+4, ... and every thread keeps KR x 4 independent 16-byte buffer loads in flight (rows / columns
// outside the tile are not issued), so the dependent chain is  record -> load round(s) -> LDS reduce ->
// store  and a launch is Fh x Cp/4 x B workgroups (384 at cfg2's B = 8; round 2's whole-band form: 96).
// What tools/attn_probe.hip (wall-clock stamps inside the kernel) says bounds it at cfg2: the bytes the CUs
// pull from L2 — a tap spacing of (size+1)/F against a band of 2R+1 rows means every image row is read by
// ~8 taps' workgroups, 50 MB per launch at ~75 GB/s per CU — and the 2.5 us of a dependent launch; 2-D
// tiles (TJ taps x NI outputs per workgroup) cut the bytes to 20 MB but pay it back in multiply-adds per
// loaded row and a longer reduction, and measured no faster.
// The row filter's weights of a round are evaluated once per wave, one per lane, and handed to the
// multiply-adds through v_readlane (a v_exp costs 16 cycles of a SIMD that holds one or two waves).
// Work items are dealt to the 8 XCDs in contiguous chunks (workgroup id % 8 = XCD): the taps of one
// image share almost all their rows, and those re-reads then hit ONE L2.
template <int KR>
__global__ __launch_bounds__(256) void extract_rows_kernel(const float *__restrict__ img, int Ci, int chan0,
                                                            const float *__restrict__ canvas, int canvas_chan,
                                                            const float *__restrict__ attn, int H, int W, int Fh, int Fw,
                                                            int Cp, int use_gamma, float *__restrict__ patch,
                                                            int n_items, int chunk, int prio) {
  raise_prio(prio);
  __shared__ f32x4 red[4][256];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int slot = blockIdx.x >> 3;
  const int item = (blockIdx.x & 7) * chunk + slot;
  if (slot >= chunk || item >= n_items) return;
  RA_PROBE_AT(0);
  const int ncg = Cp >> 2;
  const int b = item / (Fh * ncg), rem = item - b * Fh * ncg;
  const int j = rem / ncg, cg = rem - j * ncg;
  const float *rec = attn + (size_t)b * RA_ATTN_STRIDE;
  const Axis Ay = make_axis(rec, 0, H, Fh), Ax = make_axis(rec, 1, W, Fw);
  int l0, l1, w0, w1, tmp;
  Ay.band(j, l0, l1);
  Ax.band(0, w0, tmp);
  Ax.band(Fw - 1, tmp, w1);
  if (w1 > -1000000) RA_PROBE_AT(1);  // the record has arrived
  const int ch0 = chan0 + 4 * cg;
  const bool use_canvas = canvas != nullptr && (canvas_chan >= ch0) && (canvas_chan < ch0 + 4);
  const int cslot = canvas_chan - ch0;
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(img + (size_t)b * H * W * Ci + ch0), 0, (int)((size_t)H * W * Ci * 4 - (size_t)ch0 * 4),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(use_canvas ? canvas + (size_t)b * H * W : img), 0, use_canvas ? H * W * 4 : 0, 0x00020000);
  constexpr int kOOB = 0x7fffffff;

  // stage-2 role: output column oi, its band's columns dealt to `parts` neighbouring lanes
  const int parts = (Fw * 4 <= 256) ? 4 : (Fw * 2 <= 256) ? 2 : 1;
  const int oi = t / parts, part = t - oi * parts;
  const bool owner = oi < Fw;
  int bi_lo = 0, bi_hi = 0;
  if (owner) Ax.band(oi, bi_lo, bi_hi);
  f32x4 P = f32x4{0, 0, 0, 0};
  constexpr int kPre = 8;
  float fpre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) fpre[u] = 0.0f;

  for (int cp = w0; cp < w1; cp += 256) {
    f32x4 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = f32x4{0, 0, 0, 0};
    for (int rb = l0; rb < l1; rb += 4 * KR) {
      f32x4 xv[KR][4];
      float cv[KR][4];
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int row = rb + 4 * k + wv;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int col = cp + 64 * p + lane;
          const int pix = (row < l1 && col < w1) ? row * W + col : -1;
          xv[k][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, pix >= 0 ? pix * Ci * 4 : kOOB, 0, 0));
          cv[k][p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, pix >= 0 ? pix * 4 : kOOB, 0, 0));
        }
      }
      // this thread's first 8 column-filter weights (terms bi_lo + part + u * parts) depend on the record only:
      // evaluated here, under the loads' latency, instead of behind their wait
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const int ww = bi_lo + part + u * parts;
        fpre[u] = (owner && ww < bi_hi) ? Ax.w((float)ww, oi) : 0.0f;
      }
      // the row filter of this wave's KR rows: lane k evaluates row rb + 4k + wv
      const int rowl = rb + 4 * lane + wv;
      const float fyv = (lane < KR && rowl < l1) ? Ay.w((float)rowl, j) : 0.0f;
      // straight-line on purpose: a branch per row makes the compiler sink every load to its use (one
      // round trip per row); rows beyond the band were not loaded (0) and carry zero weights
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const float wy = readlane_f(fyv, k);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          f32x4 x = xv[k][p];
          if (use_canvas) {
            x.x = cslot == 0 ? cv[k][p] : x.x;
            x.y = cslot == 1 ? cv[k][p] : x.y;
            x.z = cslot == 2 ? cv[k][p] : x.z;
            x.w = cslot == 3 ? cv[k][p] : x.w;
          }
          acc[p] += wy * x;
        }
      }
    }
    __syncthreads();  // the previous column group's stage 2 is done with `red`
#pragma unroll
    for (int p = 0; p < 4; ++p) red[wv][64 * p + lane] = acc[p];
    __syncthreads();
    if (cp == w0) RA_PROBE_AT(2);  // all rows loaded and reduced
    red[0][t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);  // column t of this group, all four row slots
    __syncthreads();
    if (owner) {
      const int cend = (cp + 256 < w1) ? cp + 256 : w1;
      const int a = bi_lo > cp ? bi_lo : cp, c = bi_hi < cend ? bi_hi : cend;
      if (bi_lo >= cp && bi_hi <= cend && bi_hi - bi_lo <= kPre * parts) {  // the usual case: the pre-computed weights
        f32x4 sv[kPre];
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          const int ww = bi_lo + part + u * parts;
          const int wi = (ww < bi_hi ? ww : bi_hi - 1) - cp;
          sv[u] = red[0][wi > 0 ? wi : 0];
        }
#pragma unroll
        for (int u = 0; u < kPre; ++u) P += fpre[u] * sv[u];
      } else {
        for (int wb = a; wb < c; wb += 4 * parts) {  // batches of 4 terms: independent exps and LDS reads
          float f[4];
          f32x4 sv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ww = wb + u * parts + part;
            const int wc = ww < c ? ww : c - 1;
            f[u] = (ww < c) ? Ax.w((float)wc, oi) : 0.0f;
            sv[u] = red[0][wc - cp];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) P += f[u] * sv[u];
        }
      }
    }
  }
  if (parts >= 2) {
#pragma unroll
    for (int c = 0; c < 4; ++c) P[c] += __shfl_xor(P[c], 1);
  }
  if (parts >= 4) {
#pragma unroll
    for (int c = 0; c < 4; ++c) P[c] += __shfl_xor(P[c], 2);
  }
  if (owner && part == 0) {
    const float gamma = use_gamma ? rec[6] : 1.0f;
    *reinterpret_cast<f32x4 *>(patch + (((size_t)b * Fh + j) * Fw + oi) * Cp + 4 * cg) = gamma * P;
  }
  RA_PROBE_AT(3);
}

// ---- extract + the first attention-CNN layer in ONE launch (round 5) --------------------------------------
// The decode loop's tail is a chain of dependent launches (extract -> 6 conv -> 7 transposed conv -> paste), each
// ~2.5 us of launch floor.  The first conv layer (nnlib.cnn layer 0 of the attention CNN: 3x3 SAME on the 48 x 48 patch,
// BN + ReLU, no pooling; full_model.py:792-795, nnlib.py:229-253) needs, for output row j, the patch rows j-1, j, j+1 —
// and the three taps' row bands overlap almost completely (band 2R+1 = 31 rows against a tap spacing of 3.7 rows at
// cfg2: the union is 38 rows).  So workgroup (tap j, image b) reduces its rows ONCE with three row-filter weights per
// row, contracts the three row sums with the column filter, keeps the three patch rows in LDS and evaluates the
// conv's output row j from them: no second launch and no cross-workgroup hand-over (the neighbours' rows are
// recomputed, not waited for).  Writes x_patch row j (the model's output; dcnn skip source) and h_acnn[0] row j.
// Shapes: one packed channel group (Cp = 4: the CVPPP input), Fw <= 64, Cout <= 16.
template <int KR>
__global__ __launch_bounds__(256) void extract_conv0_kernel(const float *__restrict__ img, int Ci, int chan0,
                                                             const float *__restrict__ canvas, int canvas_chan,
                                                             const float *__restrict__ attn, int H, int W, int Fh, int Fw,
                                                             int use_gamma, float *__restrict__ patch,
                                                             const float *__restrict__ w0, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int Cout, int relu,
                                                             float *__restrict__ y0, int n_items, int chunk) {
  __shared__ f32x4 red[3][4][256];
  __shared__ f32x4 p3[3][66];       // patch rows j-1, j, j+1 with one zero column each side (SAME padding)
  __shared__ float w0s[9 * 4 * 16];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int slot = blockIdx.x >> 3;
  const int item = (blockIdx.x & 7) * chunk + slot;
  if (slot >= chunk || item >= n_items) return;
  const int b = item / Fh, j = item - b * Fh;
  const float *rec = attn + (size_t)b * RA_ATTN_STRIDE;
  const Axis Ay = make_axis(rec, 0, H, Fh), Ax = make_axis(rec, 1, W, Fw);
  int l0d[3], l1d[3], l0u = H, l1u = 0, w0c, w1c, tmp;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int jd = j - 1 + d;
    l0d[d] = l1d[d] = 0;
    if (jd >= 0 && jd < Fh) {
      Ay.band(jd, l0d[d], l1d[d]);
      l0u = l0d[d] < l0u ? l0d[d] : l0u;
      l1u = l1d[d] > l1u ? l1d[d] : l1u;
    }
  }
  Ax.band(0, w0c, tmp);
  Ax.band(Fw - 1, tmp, w1c);
  for (int e = t; e < 36 * Cout; e += 256) w0s[e] = w0[e];
  if (t < 6) p3[t >> 1][(t & 1) ? Fw + 1 : 0] = f32x4{0, 0, 0, 0};
  const bool use_canvas = canvas != nullptr && (canvas_chan >= chan0) && (canvas_chan < chan0 + 4);
  const int cslot = canvas_chan - chan0;
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(img + (size_t)b * H * W * Ci + chan0), 0, (int)((size_t)H * W * Ci * 4 - (size_t)chan0 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(use_canvas ? canvas + (size_t)b * H * W : img), 0, use_canvas ? H * W * 4 : 0, 0x00020000);
  constexpr int kOOB = 0x7fffffff;

  const int oi = t >> 2, part = t & 3;  // stage 2: output column oi, its band's columns dealt to 4 neighbouring lanes
  const bool owner = oi < Fw;
  int bi_lo = 0, bi_hi = 0;
  if (owner) Ax.band(oi, bi_lo, bi_hi);
  f32x4 P[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) P[d] = f32x4{0, 0, 0, 0};
  constexpr int kPre = 8;
  float fpre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) fpre[u] = 0.0f;

  for (int cp = w0c; cp < w1c; cp += 256) {
    f32x4 acc[3][4];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[d][p] = f32x4{0, 0, 0, 0};
    for (int rb = l0u; rb < l1u; rb += 4 * KR) {
      f32x4 xv[KR][4];
      float cv[KR][4];
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int row = rb + 4 * k + wv;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int col = cp + 64 * p + lane;
          const int pix = (row < l1u && col < w1c) ? row * W + col : -1;
          xv[k][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, pix >= 0 ? pix * Ci * 4 : kOOB, 0, 0));
          cv[k][p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, pix >= 0 ? pix * 4 : kOOB, 0, 0));
        }
      }
#pragma unroll
      for (int u = 0; u < kPre; ++u) {  // under the loads' latency: this thread's first 8 column-filter weights
        const int ww = bi_lo + part + u * 4;
        fpre[u] = (owner && ww < bi_hi) ? Ax.w((float)ww, oi) : 0.0f;
      }
      // the three taps' row-filter weights of this wave's KR rows: lane k evaluates row rb + 4k + wv (zero outside a tap's band)
      const int rowl = rb + 4 * lane + wv;
      float fyv[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) fyv[d] = (lane < KR && rowl >= l0d[d] && rowl < l1d[d]) ? Ay.w((float)rowl, j - 1 + d) : 0.0f;
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const float wy0 = readlane_f(fyv[0], k), wy1 = readlane_f(fyv[1], k), wy2 = readlane_f(fyv[2], k);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          f32x4 x = xv[k][p];
          if (use_canvas) {
            x.x = cslot == 0 ? cv[k][p] : x.x;
            x.y = cslot == 1 ? cv[k][p] : x.y;
            x.z = cslot == 2 ? cv[k][p] : x.z;
            x.w = cslot == 3 ? cv[k][p] : x.w;
          }
          acc[0][p] += wy0 * x;
          acc[1][p] += wy1 * x;
          acc[2][p] += wy2 * x;
        }
      }
    }
    __syncthreads();  // the previous column group's stage 2 is done with `red`
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int p = 0; p < 4; ++p) red[d][wv][64 * p + lane] = acc[d][p];
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; ++d) red[d][0][t] = (red[d][0][t] + red[d][1][t]) + (red[d][2][t] + red[d][3][t]);
    __syncthreads();
    if (owner) {
      const int cend = (cp + 256 < w1c) ? cp + 256 : w1c;
      const int a = bi_lo > cp ? bi_lo : cp, c = bi_hi < cend ? bi_hi : cend;
      if (bi_lo >= cp && bi_hi <= cend && bi_hi - bi_lo <= kPre * 4) {  // the usual case: the pre-computed weights
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
          const int ww = bi_lo + part + u * 4;
          const int wi = (ww < bi_hi ? ww : bi_hi - 1) - cp;
          const int wj = wi > 0 ? wi : 0;
#pragma unroll
          for (int d = 0; d < 3; ++d) P[d] += fpre[u] * red[d][0][wj];
        }
      } else {
        for (int wb = a; wb < c; wb += 16) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ww = wb + u * 4 + part;
            const int wc = ww < c ? ww : c - 1;
            const float f = (ww < c) ? Ax.w((float)wc, oi) : 0.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) P[d] += f * red[d][0][wc - cp];
          }
        }
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      P[d][c] += __shfl_xor(P[d][c], 1);
      P[d][c] += __shfl_xor(P[d][c], 2);
    }
  if (owner && part == 0) {
    const float gamma = use_gamma ? rec[6] : 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) p3[d][oi + 1] = gamma * P[d];
    *reinterpret_cast<f32x4 *>(patch + (((size_t)b * Fh + j) * Fw + oi) * 4) = gamma * P[1];
  }
  __syncthreads();
  // the conv layer's output row j: u[i, co] = sum_{d, kx, c} w0[d][kx][c][co] * P[j - 1 + d][i + kx - 1][c], BN (folded) + ReLU
  const float lo = relu ? 0.f : -__builtin_inff();
  for (int e = t; e < Fw * Cout; e += 256) {
    const int i = e / Cout, co = e - i * Cout;
    float u = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f32x4 pv = p3[d][i + kx];
        const float *wr = w0s + ((d * 3 + kx) * 4) * Cout + co;
        u += wr[0] * pv.x;
        u += wr[Cout] * pv.y;
        u += wr[2 * Cout] * pv.z;
        u += wr[3 * Cout] * pv.w;
      }
    y0[(((size_t)b * Fh + j) * Fw + i) * Cout + co] = fmaxf(u * scale[co] + shift[co], lo);
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


// ---- paste, one-round-trip form (the product's shapes) -------------------------------------------------
// Same operator as paste_direct_kernel for the case the decode loop runs — canvas in its own plane,
// one-channel patch, W % 4 == 0, RB * W <= 4096 — re-ordered so that every global load a workgroup can
// need is issued in ONE round, right after the attention record (scalar loads) has said whether the
// workgroup holds window rows at all: the whole patch plane (Fh x Fw floats, staged to LDS) and — before
// the window's columns are worked out — the canvas of the workgroup's own RB rows (buffer loads: rows
// beyond the image return 0).  Chain:
//   launch -> record -> one load round -> V (LDS) -> tap loop -> stores.
// The rest is instruction issue (one or two waves per SIMD: ~5 cycles per dependent instruction, 16 per
// v_exp; tools/lat_probe.hip), so: in the V phase a wave owns one image row and lane k evaluates the row
// filter's tap k once (v_readlane hands it to the multiply-adds); in the tap loop a thread owns ONE image
// column and all RB rows — the column filter fx(w, i) is evaluated once per tap and column instead of once
// per tap and pixel — and V is stored [i][RB] so that one LDS read feeds RB multiply-adds; stores are
// buffer stores with a scalar row offset (no 64-bit address arithmetic per pixel).
constexpr int kPsQuads = 3;  // float4 groups of the patch plane a thread stages (3 x 256 x 4 = 3072 >= 48 x 48)

__device__ inline float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

template <int MODE, int RB>
__global__ __launch_bounds__(256) void paste_win_kernel(const float *__restrict__ patch,
                                                         const float *__restrict__ attn, int H, int W, int Fh, int Fw,
                                                         float beta, int disable_overwrite, float *__restrict__ canvas,
                                                         float *__restrict__ y_out, size_t y_stride_b, int flags,
                                                         ScoreArgs sc, int prio) {
  static_assert(RB == 4 || RB == 8, "rows per workgroup");
  raise_prio(prio);
  constexpr int nthr = 256, CVQ = RB;  // blind canvas float4 groups per thread (RB * W <= 4096 floats)
  extern __shared__ f32x4 smem4[];    // Cs [CVQ * 256] float4 | V [Fw][RB] | Ps [Fh * Fw]
  const int l0 = blockIdx.x * RB, b = blockIdx.y, t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  float *Cs = reinterpret_cast<float *>(smem4);
  float *V = reinterpret_cast<float *>(smem4 + CVQ * nthr);
  float *Ps = V + RB * Fw;
  if (l0 >= H) {  // the rider workgroup (grid.x is one larger when a score is requested)
    float s = 0.0f;
    const int K = sc.K0 + sc.K1;
    for (int k = t; k < K; k += nthr) {
      const float xv = (k < sc.K0) ? sc.h[(size_t)b * sc.K0 + k] : sc.core[(size_t)b * sc.K1 + (k - sc.K0)];
      s += xv * sc.w[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) V[wv] = s;
    __syncthreads();
    if (t == 0) {
      float tot = sc.bias ? sc.bias[0] : 0.0f;
      for (int k = 0; k < (nthr >> 6); ++k) tot += V[k];
      sc.s_out[(size_t)b * sc.stride] = sigmoidf(tot);
    }
    return;
  }
  RA_PROBE_AT(0);
  const int nrow = (H - l0) < RB ? (H - l0) : RB;
  // (1) the record: which of this workgroup's rows hold window pixels (60 % of the workgroups of a cfg2
  // launch hold none, and their share of the blind loads below would only queue in front of the others')
  const float *rec = attn + (size_t)b * RA_ATTN_STRIDE;
  const Axis Ay = make_axis(rec, 0, H, Fh), Ax = make_axis(rec, 1, W, Fw);
  const float gain = (MODE == 0) ? __expf(rec[8]) : rec[7];
  const bool has_cv = (MODE == 0);
  const bool floored = !has_cv || (flags & RA_PASTE_CANVAS_FLOORED);
  const bool y_pre = (flags & RA_PASTE_Y_PREFILLED) != 0;
  const bool dead_needs_cv = has_cv && (disable_overwrite || !floored);
  const bool dead_noop = y_pre && !dead_needs_cv;  // nothing to do outside the window
  int jlo[RB], jhi[RB], live_mask = 0;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    int a, c;
    Ay.taps(l0 + r, a, c);
    jlo[r] = __builtin_amdgcn_readfirstlane(a);
    jhi[r] = __builtin_amdgcn_readfirstlane(r < nrow ? c : a);
    live_mask |= (jlo[r] < jhi[r]) ? (1 << r) : 0;
  }
  if (live_mask == 0 && dead_noop) return;
  // (2) every load the rest can need, in one round: the patch plane and this workgroup's canvas rows
  f32x4 pr[kPsQuads], cvp[CVQ];
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc(y_out + (size_t)b * y_stride_b + (size_t)l0 * W, 0, nrow * W * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
      MODE == 0 ? canvas + ((size_t)b * H + l0) * W : y_out, 0, MODE == 0 ? nrow * W * 4 : 0, 0x00020000);
  if (MODE == 0) {
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(patch + (size_t)b * Fh * Fw), 0, live_mask ? Fh * Fw * 4 : 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < kPsQuads; ++q)
      pr[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, (t + q * nthr) * 16, 0, 0));
#pragma unroll
    for (int q = 0; q < CVQ; ++q)
      cvp[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, (t + q * nthr) * 16, 0, 0));
  }
  int wbeg, wend;
  {
    int w0, w1, tmp;
    Ax.band(0, w0, tmp);
    Ax.band(Fw - 1, tmp, w1);
    wbeg = w0 & ~3;
    wend = (w1 + 3) & ~3;
    wend = wend < W ? wend : W;
    if (wend < wbeg) wend = wbeg;
    wbeg = __builtin_amdgcn_readfirstlane(wbeg);
    wend = __builtin_amdgcn_readfirstlane(wend);
  }
  const float y_dead = fast_sigmoid(beta);
  if (wend > -1000000) RA_PROBE_AT(1);  // the record has arrived

  // (3) pixels outside the window: nothing depends on the patch
  if (!dead_noop) {
    const int ngf = W >> 2;  // float4 column groups of a full row
#pragma unroll
    for (int q = 0; q < CVQ; ++q) {
      const int e = t + q * nthr;
      const int r = e / ngf, w4 = 4 * (e - r * ngf);
      const bool live = ((live_mask >> r) & 1) && w4 >= wbeg && w4 < wend;
      if (e < nrow * ngf && !live) {
        f32x4 cv = (MODE == 0 && dead_needs_cv) ? cvp[q] : f32x4{0, 0, 0, 0}, y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = y_dead;
          if (MODE == 0 && disable_overwrite) v *= (1.0f - cv[k]);
          y[k] = v;
          cv[k] = fmaxf(cv[k], v);
        }
        if (!y_pre) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y), rsY, e * 16, 0, 0);
        if (has_cv && !floored)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, cv), rsV, e * 16, 0, 0);
      }
    }
  }
  RA_PROBE_AT(2);  // dead pixels issued
  if (live_mask == 0 || wbeg >= wend) return;

  // (4) V[i][r] = gain * sum_j fy(l0 + r, j) P[j, i]; the blind canvas groups change hands through LDS
  if (MODE == 0) {
#pragma unroll
    for (int q = 0; q < CVQ; ++q) smem4[t + q * nthr] = cvp[q];
#pragma unroll
    for (int q = 0; q < kPsQuads; ++q)
      if ((t + q * nthr) * 4 < Fh * Fw) reinterpret_cast<f32x4 *>(Ps)[t + q * nthr] = pr[q];
    __syncthreads();
  }
#pragma unroll
  for (int r0 = 0; r0 < RB; r0 += 4) {  // a wave owns a row: its tap range and weights are wave-uniform
    int ja = jlo[r0], jb = jhi[r0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (wv == k) {
        ja = jlo[r0 + k];
        jb = jhi[r0 + k];
      }
    const int r = r0 + wv, l = l0 + r, i = lane;  // Fw <= 64: lane = output column
    float s = 0.0f;
    for (int jc = ja; jc < jb; jc += 64) {
      const int nj = (jb - jc) < 64 ? (jb - jc) : 64;
      // tap jc + lane: one v_exp per wave, evaluated by ALL lanes (v_readlane below picks any of them)
      const float wl = (lane < nj) ? Ay.w((float)l, jc + lane) : 0.0f;
      if (i < Fw) {
        if (MODE == 0) {
          const float *pp = Ps + jc * Fw + i;
          int k = 0;
          for (; k + 4 <= nj; k += 4) {
            const float p0 = pp[k * Fw], p1 = pp[(k + 1) * Fw], p2 = pp[(k + 2) * Fw], p3 = pp[(k + 3) * Fw];
            s += readlane_f(wl, k) * p0;
            s += readlane_f(wl, k + 1) * p1;
            s += readlane_f(wl, k + 2) * p2;
            s += readlane_f(wl, k + 3) * p3;
          }
          for (; k < nj; ++k) s += readlane_f(wl, k) * pp[k * Fw];
        } else {
          for (int k = 0; k < nj; ++k) s += readlane_f(wl, k);  // P == 1 (const_ones)
        }
      }
    }
    if (i < Fw) V[i * RB + r] = gain * s;
  }
  __syncthreads();
  RA_PROBE_AT(3);  // V done

  // (5) the window: a thread owns one column and all RB rows
  const float dis = (MODE == 0 && disable_overwrite) ? 1.0f : 0.0f;
  for (int w = wbeg + t; w < wend; w += nthr) {
    int ilo, ihi;
    Ax.taps(w, ilo, ihi);
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
    for (int i = ilo; i < ihi; i += 4) {
      float fx[4];
      f32x4 v4[4][RB / 4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int iu = (i + u < ihi) ? i + u : ihi - 1;
        const float f = Ax.w((float)w, iu);
        fx[u] = (i + u < ihi) ? f : 0.0f;
#pragma unroll
        for (int h = 0; h < RB / 4; ++h) v4[u][h] = reinterpret_cast<const f32x4 *>(V)[iu * (RB / 4) + h];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] += v4[u][r / 4][r % 4] * fx[u];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (!((live_mask >> r) & 1)) continue;  // a dead row of this block (or beyond H): workgroup-uniform
      const float cv = (MODE == 0) ? Cs[r * W + w] : 0.0f;
      float v = fast_sigmoid(acc[r] + beta);
      v -= dis * v * cv;  // y *= (1 - canvas) when overwriting is disabled
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, w * 4, r * W * 4, 0);
      if (MODE == 0)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(cv, v)), rsV, w * 4, r * W * 4, 0);
    }
  }
  RA_PROBE_AT(4);
}

}  // namespace attnd
}  // namespace ra

using namespace ra;

namespace {
// paste_win_kernel where its shape conditions hold (the decode loop's launches), the general kernel otherwise
template <int MODE>
void launch_paste(int extra_wg, const float *patch, int Cp, int pc, const float *attn_rec, int B, int H, int W, int Fh,
                  int Fw, float beta, int disable_overwrite, float *canvas, float *img, int Ci, int canvas_chan,
                  float *y_out, size_t y_stride_b, int flags, attnd::ScoreArgs sc, void *stream) {
  const attnd::PasteGeo pg = attnd::paste_geo();
  const int rb = pg.rows == 8 ? 8 : 4;
  const bool a16 = ((reinterpret_cast<uintptr_t>(y_out) | reinterpret_cast<uintptr_t>(canvas) |
                     reinterpret_cast<uintptr_t>(patch)) & 15) == 0;
  const bool win_ok = W % 4 == 0 && Fw <= 64 && rb * W <= 4096 && (y_stride_b & 3) == 0 && a16 &&
                      (size_t)H * W * 4 < 0x7fffffffu && Fh * Fw % 4 == 0 && Fh * Fw <= 4 * attnd::kPsQuads * 256 &&
                      (MODE == 1 || (canvas && !img && Cp == 1 && pc == 0));
  if (!win_ok) {
    const size_t lds = (size_t)(pg.rows * Fw > 8 ? pg.rows * Fw : 8) * sizeof(float);
    hipLaunchKernelGGL(attnd::paste_direct_kernel<MODE>, dim3(ceil_div(H, pg.rows) + extra_wg, B), dim3(pg.threads), lds,
                       as_stream(stream), patch, Cp, pc, attn_rec, H, W, Fh, Fw, beta, disable_overwrite, canvas, img, Ci,
                       canvas_chan, y_out, y_stride_b, flags, pg.rows, sc);
    return;
  }
  const size_t lds = (size_t)rb * 256 * 16 + (size_t)(rb * Fw + Fh * Fw + 16) * sizeof(float);
  if (rb == 8)
    hipLaunchKernelGGL((attnd::paste_win_kernel<MODE, 8>), dim3(ceil_div(H, 8) + extra_wg, B), dim3(256), lds,
                       as_stream(stream), patch, attn_rec, H, W, Fh, Fw, beta, disable_overwrite, canvas, y_out, y_stride_b,
                       flags, sc, tail_prio());
  else
    hipLaunchKernelGGL((attnd::paste_win_kernel<MODE, 4>), dim3(ceil_div(H, 4) + extra_wg, B), dim3(256), lds,
                       as_stream(stream), patch, attn_rec, H, W, Fh, Fw, beta, disable_overwrite, canvas, y_out, y_stride_b,
                       flags, sc, tail_prio());
}
}  // namespace

extern "C" int ra_extract_direct_f32(const float *img, int Ci, int chan0, const float *canvas,
                                     int canvas_chan, const float *attn_rec, int B, int H, int W, int Fh,
                                     int Fw, int Cp, int use_gamma, float *patch, void *stream) {
  if (!img || !attn_rec || !patch || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0)
    return fail(RA_E_INVALID, "ra_extract_direct_f32: bad argument");
  if (Ci % 4 || Cp % 4 || chan0 % 4 || chan0 + Cp > Ci || Cp <= 0 || Fw > 256)
    return fail(RA_E_SHAPE, "ra_extract_direct_f32: Ci=%d chan0=%d Cp=%d Fw=%d", Ci, chan0, Cp, Fw);
  if ((size_t)H * W * Ci * 4 >= 0x7fffffffu) return fail(RA_E_SHAPE, "ra_extract_direct_f32: one image exceeds 2 GiB");
  const int n_items = Fh * (Cp / 4) * B, chunk = ceil_div(n_items, 8);
  hipLaunchKernelGGL((attnd::extract_rows_kernel<4>), dim3(8 * chunk), dim3(256), 0, as_stream(stream), img, Ci, chan0,
                     canvas, canvas_chan, attn_rec, H, W, Fh, Fw, Cp, use_gamma, patch, n_items, chunk, tail_prio());
  return launch_status("ra_extract_direct_f32");
}

extern "C" int ra_extract_conv0_supported(int Cp, int Fh, int Fw, int Cout, int pool) {
  return Cp == 4 && Fw >= 1 && Fw <= 64 && Fh >= 1 && Cout >= 1 && Cout <= 16 && pool == 1;
}

extern "C" int ra_extract_conv0_f32(const float *img, int Ci, int chan0, const float *canvas, int canvas_chan,
                                    const float *attn_rec, int B, int H, int W, int Fh, int Fw, int use_gamma, float *patch,
                                    const float *w0, const float *scale, const float *shift, int Cout, int relu, float *y0,
                                    void *stream) {
  if (!img || !attn_rec || !patch || !w0 || !scale || !shift || !y0 || B <= 0 || H <= 0 || W <= 0)
    return fail(RA_E_INVALID, "ra_extract_conv0_f32: bad argument");
  if (!ra_extract_conv0_supported(4, Fh, Fw, Cout, 1) || Ci % 4 || chan0 % 4 || chan0 + 4 > Ci)
    return fail(RA_E_SHAPE, "ra_extract_conv0_f32: Ci=%d chan0=%d Fh=%d Fw=%d Cout=%d", Ci, chan0, Fh, Fw, Cout);
  if ((size_t)H * W * Ci * 4 >= 0x7fffffffu) return fail(RA_E_SHAPE, "ra_extract_conv0_f32: one image exceeds 2 GiB");
  const int n_items = Fh * B, chunk = ceil_div(n_items, 8);
  static int kr = 0;  // RA_EXC0_KR=4: tuning aid (rows per wave and load round; 5 covers cfg2's 38-row union in two rounds)
  if (!kr) {
    const char *e = getenv("RA_EXC0_KR");
    kr = (e && atoi(e) == 4) ? 4 : 5;
  }
  if (kr == 4)
    hipLaunchKernelGGL((attnd::extract_conv0_kernel<4>), dim3(8 * chunk), dim3(256), 0, as_stream(stream), img, Ci, chan0, canvas,
                       canvas_chan, attn_rec, H, W, Fh, Fw, use_gamma, patch, w0, scale, shift, Cout, relu, y0, n_items, chunk);
  else
    hipLaunchKernelGGL((attnd::extract_conv0_kernel<5>), dim3(8 * chunk), dim3(256), 0, as_stream(stream), img, Ci, chan0, canvas,
                       canvas_chan, attn_rec, H, W, Fh, Fw, use_gamma, patch, w0, scale, shift, Cout, relu, y0, n_items, chunk);
  return launch_status("ra_extract_conv0_f32");
}

extern "C" int ra_paste_direct_f32(const float *patch, int Cp, int pc, const float *attn_rec, int B, int H,
                                   int W, int Fh, int Fw, float beta, int disable_overwrite,
                                   float *canvas, float *img, int Ci, int canvas_chan, float *y_out,
                                   size_t y_stride_b, int flags, void *stream) {
  if (!patch || !attn_rec || !y_out || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0 || Cp <= 0 ||
      pc < 0 || pc >= Cp)
    return fail(RA_E_INVALID, "ra_paste_direct_f32: bad argument");
  launch_paste<0>(0, patch, Cp, pc, attn_rec, B, H, W, Fh, Fw, beta, disable_overwrite, canvas, img, Ci, canvas_chan, y_out,
                  y_stride_b, disable_overwrite ? (flags & ~RA_PASTE_Y_PREFILLED) : flags, attnd::ScoreArgs{}, stream);
  return launch_status("ra_paste_direct_f32");
}

extern "C" int ra_attn_box_direct_f32(const float *attn_rec, int B, int H, int W, int Fh, int Fw, float beta,
                                      float *box_out, size_t stride_b, void *stream) {
  if (!attn_rec || !box_out || B <= 0 || H <= 0 || W <= 0 || Fh <= 0 || Fw <= 0)
    return fail(RA_E_INVALID, "ra_attn_box_direct_f32: bad argument");
  launch_paste<1>(0, nullptr, 1, 0, attn_rec, B, H, W, Fh, Fw, beta, 0, nullptr, nullptr, 0, -1, box_out, stride_b, 0,
                  attnd::ScoreArgs{}, stream);
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
  attnd::ScoreArgs sc{h, core, w, bias, s_out, K0, K1, s_stride_b};
  launch_paste<0>(1, patch, Cp, pc, attn_rec, B, H, W, Fh, Fw, beta, disable_overwrite, canvas, nullptr, 0, -1, y_out,
                  y_stride_b, disable_overwrite ? (flags & ~RA_PASTE_Y_PREFILLED) : flags, sc, stream);
  return launch_status("ra_paste_score_direct_f32");
}
