// K1 — 3x3 SAME convolution as an f32-MFMA implicit GEMM for gfx950, with the bias + BatchNorm
// (eval) + ReLU + 2x2 max-pool epilogue fused.  One kernel serves
//   * nnlib.cnn layers            (nnlib.py:229-253)      ctrl_cnn / attn_cnn
//   * nnlib.dcnn layers           (nnlib.py:362-400)      attn_dcnn: a SAME conv2d_transpose is a
//     SAME conv of the (stride 2: zero-stuffed) input with the spatially flipped, in/out-swapped
//     filter; the concat(prev, skip) is two source pointers.
//
// GEMM view: D[pixel, cout] = sum_k A[pixel, k] * B[k, cout],  k = (ky, kx, ci).
// v_mfma_f32_16x16x4_f32 (exact f32, 32 cycles/SIMD): A operand = 16 pixels x 4 k, one
// ds_read_b32 per lane; B operand = 4 k x 16 couts held in registers for a whole Cin chunk;
// D = 4 VGPRs/lane = rows 4*(lane>>4)+r of column lane&15.  The 16 pixel rows of a tile are
// mapped m = 4*q + r with r = (dy,dx) of a 2x2 window and q = pooled x position, so the four
// accumulator registers of a lane ARE one max-pool window of one output channel: the pool is
// three v_max in registers and the store is 64 B contiguous per 16 lanes.
//
// Workgroup = 4 waves.  WN waves split the cout groups, 4/WN waves split pixel rows.  Each wave
// owns PM = GX*GY pixel groups (2 rows x 8 cols each) x NC cout groups of 16.
// LDS holds one Cin chunk (CK channels) of the input tile + halo as [CK/4][rows][cols][4].
#include "ra_common.h"

namespace ra {
namespace conv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const float *src0;
  const float *src1;
  const float *wp;
  const float *scale;
  const float *shift;
  float *y;
  int C0, C1;      // channels of src0 / src1
  int Hs, Ws;      // source spatial size
  int H, W;        // conv (pre-pool) spatial size = Hs*(1+ups)
  int ups;         // zero-stuffed stride-2 transposed-conv input
  int Cout, CoutP;
  int relu, pool;
  int Ho, Wo;
};

template <int CK, int NC, int WN, int GX, int GY>
struct Geo {
  static constexpr int WM = 4 / WN;            // waves along pixel rows
  static constexpr int PM = GX * GY;           // pixel groups per wave
  static constexpr int TW = 8 * GX;            // tile cols
  static constexpr int WR = 2 * GY;            // rows per wave
  static constexpr int TH = WR * WM;           // tile rows
  static constexpr int LW = TW + 2;            // LDS cols (halo)
  static constexpr int LH = TH + 2;            // LDS rows
  static constexpr int NCG = CK / 4;           // channel groups per chunk
  static constexpr int PLANE0 = LH * LW * 4;   // floats per channel-group plane
  // pad the plane stride to == 8 (mod 32) dwords so the 16-B staging writes of the NCG planes
  // land on different bank quads
  static constexpr int PLANE = PLANE0 + ((8 - (PLANE0 % 32)) + 32) % 32;
  static constexpr int KS = 9 * NCG;           // MFMA k-steps per chunk
  static constexpr int LDS_FLOATS = NCG * PLANE;
};

template <int CK, int NC, int WN, int GX, int GY>
__global__ __launch_bounds__(256) void conv3x3_mfma(const Args a) {
  using G = Geo<CK, NC, WN, GX, GY>;
  __shared__ __attribute__((aligned(16))) float tile[G::LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wm = wave / WN;
  const int b = blockIdx.z;
  const int ty0 = blockIdx.y * G::TH;
  const int tx0 = blockIdx.x * G::TW;
  const int Cin = a.C0 + a.C1;
  const int nchunks = Cin / CK;

  // A-operand lane geometry: m = lane & 15 -> (q, dy, dx); ksub = lane >> 4.
  const int m = lane & 15, ksub = lane >> 4;
  const int q = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
  const int a_base = ((wm * G::WR + dy) * G::LW + 2 * q + dx) * 4 + ksub;  // floats

  f32x4 acc[G::PM][NC];
#pragma unroll
  for (int g = 0; g < G::PM; ++g)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int co_lane = lane & 15;
  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- B operand for this chunk -> registers (L2-resident packed weights) ----
    float breg[G::KS][NC];
    {
      const float *wrow = a.wp + ((size_t)ch * G::KS * 4 + ksub) * a.CoutP + 16 * (wn * NC) + co_lane;
#pragma unroll
      for (int s = 0; s < G::KS; ++s)
#pragma unroll
        for (int n = 0; n < NC; ++n) breg[s][n] = wrow[(size_t)s * 4 * a.CoutP + 16 * n];
    }
    // ---- stage the input tile (+halo) of this chunk into LDS ----
    if (ch > 0) __syncthreads();
    for (int e = tid; e < G::NCG * G::LH * G::LW; e += 256) {
      const int cg = e % G::NCG;
      const int c = (e / G::NCG) % G::LW;
      const int r = e / (G::NCG * G::LW);
      const int Y = ty0 + r - 1, X = tx0 + c - 1;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
      int ys = Y, xs = X;
      if (a.ups) {
        ok = ok & (Y & 1) & (X & 1);
        ys = Y >> 1;
        xs = X >> 1;
      }
      if (ok) {
        const int chan = ch * CK + cg * 4;
        const float *p = (chan < a.C0)
                             ? a.src0 + (((size_t)b * a.Hs + ys) * a.Ws + xs) * a.C0 + chan
                             : a.src1 + (((size_t)b * a.Hs + ys) * a.Ws + xs) * a.C1 + (chan - a.C0);
        v = *reinterpret_cast<const f32x4 *>(p);
      }
      *reinterpret_cast<f32x4 *>(&tile[cg * G::PLANE + (r * G::LW + c) * 4]) = v;
    }
    __syncthreads();
    // ---- MFMA main loop: 9 taps x NCG channel groups ----
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int cg = 0; cg < G::NCG; ++cg) {
        const int s = tap * G::NCG + cg;
        float av[G::PM];
#pragma unroll
        for (int g = 0; g < G::PM; ++g) {
          const int gx = g % GX, gy = g / GX;
          av[g] = tile[a_base + cg * G::PLANE + ((2 * gy + ky) * G::LW + 8 * gx + kx) * 4];
        }
#pragma unroll
        for (int g = 0; g < G::PM; ++g)
#pragma unroll
          for (int n = 0; n < NC; ++n)
            acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g], breg[s][n], acc[g][n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: scale/shift (bias + BN), ReLU, 2x2 max-pool, store ----
  const int qo = lane >> 4;  // D rows 4*qo + r  ->  pooled x position qo, window element r
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int co = 16 * (wn * NC + n) + co_lane;
    const float sc = a.scale[co], sh = a.shift[co];
    const bool co_ok = co < a.Cout;
#pragma unroll
    for (int g = 0; g < G::PM; ++g) {
      const int gx = g % GX, gy = g / GX;
      const int row0 = ty0 + wm * G::WR + 2 * gy;
      const int col0 = tx0 + 8 * gx + 2 * qo;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[g][n][r] * sc + sh;
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
      }
      if (a.pool == 2) {
        const float o = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        const int pr = row0 >> 1, pc = col0 >> 1;
        if (co_ok && pr < a.Ho && pc < a.Wo)
          a.y[(((size_t)b * a.Ho + pr) * a.Wo + pc) * a.Cout + co] = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pr = row0 + (r >> 1), pc = col0 + (r & 1);
          if (co_ok && pr < a.Ho && pc < a.Wo)
            a.y[(((size_t)b * a.Ho + pr) * a.Wo + pc) * a.Cout + co] = v[r];
        }
      }
    }
  }
}

template <int CK, int NC, int WN, int GX, int GY>
int launch(const Args &a, int B, hipStream_t st) {
  using G = Geo<CK, NC, WN, GX, GY>;
  dim3 grid(ceil_div(a.W, G::TW), ceil_div(a.H, G::TH), B);
  hipLaunchKernelGGL((conv3x3_mfma<CK, NC, WN, GX, GY>), grid, dim3(256), 0, st, a);
  return launch_status("ra_conv3x3_f32");
}

// Tile geometry choice: the biggest tile that still yields >= ~2 workgroups per CU, narrow
// (16-col) tiles when the image width would leave a 32-col tile more than half empty.
template <int CK, int NC, int WN>
int dispatch_geo(const Args &a, int B, hipStream_t st) {
  constexpr int WM = 4 / WN;
  auto wgs = [&](int gx, int gy) {
    return (long)ceil_div(a.W, 8 * gx) * ceil_div(a.H, 2 * gy * WM) * B;
  };
  const bool narrow = (a.W % 32 != 0) && (a.W % 32 <= 16);
  const long want = 512;
  if (!narrow) {
    if (wgs(4, 2) >= want) return launch<CK, NC, WN, 4, 2>(a, B, st);
    if (wgs(4, 1) >= want) return launch<CK, NC, WN, 4, 1>(a, B, st);
  }
  if (wgs(2, 2) >= want) return launch<CK, NC, WN, 2, 2>(a, B, st);
  return launch<CK, NC, WN, 2, 1>(a, B, st);
}

template <int CK>
int dispatch_cout(const Args &a, int B, hipStream_t st) {
  switch (a.CoutP) {
    case 16: return dispatch_geo<CK, 1, 1>(a, B, st);
    case 32: return dispatch_geo<CK, 2, 1>(a, B, st);
    case 64: return dispatch_geo<CK, 2, 2>(a, B, st);
    case 128: return dispatch_geo<CK, 2, 4>(a, B, st);
    default: return fail(RA_E_SHAPE, "ra_conv3x3_f32: CoutP %d unsupported", a.CoutP);
  }
}

inline int chunk_of(int Cin) { return (Cin % 16 == 0) ? 16 : (Cin % 8 == 0) ? 8 : 4; }

}  // namespace conv
}  // namespace ra

extern "C" int ra_conv_cout_padded(int Cout) {
  if (Cout <= 0) return 0;
  if (Cout <= 16) return 16;
  if (Cout <= 32) return 32;
  if (Cout <= 64) return 64;
  if (Cout <= 128) return 128;
  return 0;
}

extern "C" size_t ra_conv_packed_floats(int Cin, int Cout) {
  const int cp = ra_conv_cout_padded(Cout);
  if (Cin <= 0 || Cin % 4 || !cp) return 0;
  return (size_t)9 * Cin * cp;
}

// Packed order: [chunk][tap = ky*3+kx][cg][ksub][CoutP]; channel = chunk*CK + cg*4 + ksub.
extern "C" int ra_conv_pack_weights(const float *w, int Cin_w, int Cout, int Cin, const int *chan_map,
                                    int flags, float *out) {
  const int cp = ra_conv_cout_padded(Cout);
  if (!w || !out || Cin_w <= 0 || Cin <= 0) return ra::fail(RA_E_INVALID, "ra_conv_pack_weights: bad argument");
  if (Cin % 4 || !cp) return ra::fail(RA_E_SHAPE, "ra_conv_pack_weights: Cin %d %% 4 or Cout %d", Cin, Cout);
  if (!chan_map && Cin_w != Cin) return ra::fail(RA_E_SHAPE, "ra_conv_pack_weights: Cin_w != Cin without map");
  const int CK = ra::conv::chunk_of(Cin), NCG = CK / 4;
  const bool tr = flags & RA_CONV_TRANSPOSED;
  for (int c = 0; c < Cin; ++c) {
    const int src_c = chan_map ? chan_map[c] : c;
    if (src_c >= Cin_w) return ra::fail(RA_E_SHAPE, "ra_conv_pack_weights: chan_map[%d] = %d", c, src_c);
    const int chunk = c / CK, cg = (c % CK) / 4, ksub = c % 4;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int tap = ky * 3 + kx;
        float *dst = out + ((((size_t)chunk * 9 + tap) * NCG + cg) * 4 + ksub) * cp;
        for (int co = 0; co < cp; ++co) {
          float v = 0.f;
          if (co < Cout && src_c >= 0) {
            if (!tr)
              v = w[(((size_t)ky * 3 + kx) * Cin_w + src_c) * Cout + co];
            else  // conv2d_transpose filter [3,3,Cout,Cin_w]: flip taps, swap in/out
              v = w[(((size_t)(2 - ky) * 3 + (2 - kx)) * Cout + co) * Cin_w + src_c];
          }
          dst[co] = v;
        }
      }
  }
  return 0;
}

extern "C" int ra_conv_fold_bn(const float *bias, const float *beta, const float *gamma,
                               const float *mean, const float *var, int Cout, float eps, float *scale,
                               float *shift) {
  const int cp = ra_conv_cout_padded(Cout);
  if (!scale || !shift || !cp) return ra::fail(RA_E_INVALID, "ra_conv_fold_bn: bad argument");
  for (int c = 0; c < cp; ++c) {
    float sc = 1.f, sh = 0.f;
    if (c < Cout) {
      const float bv = bias ? bias[c] : 0.f;
      if (gamma) {
        // tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma; y = x*inv + (beta - mean*inv)
        sc = gamma[c] / sqrtf(var[c] + eps);
        sh = (bv - mean[c]) * sc + beta[c];
      } else {
        sh = bv;
      }
    }
    scale[c] = sc;
    shift[c] = sh;
  }
  return 0;
}

extern "C" int ra_conv3x3_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs,
                              int Ws, int upsample, const float *wpacked, const float *scale,
                              const float *shift, int Cout, int relu, int pool, float *y,
                              void *stream) {
  if (!src0 || !wpacked || !scale || !shift || !y || B <= 0 || Hs <= 0 || Ws <= 0 || C0 <= 0 ||
      C1 < 0 || (C1 > 0 && !src1))
    return ra::fail(RA_E_INVALID, "ra_conv3x3_f32: bad argument");
  if (C0 % 4 || C1 % 4) return ra::fail(RA_E_SHAPE, "ra_conv3x3_f32: C0=%d C1=%d must be %% 4", C0, C1);
  ra::conv::Args a;
  a.src0 = src0;
  a.src1 = src1;
  a.wp = wpacked;
  a.scale = scale;
  a.shift = shift;
  a.y = y;
  a.C0 = C0;
  a.C1 = C1;
  a.Hs = Hs;
  a.Ws = Ws;
  a.ups = upsample ? 1 : 0;
  a.H = Hs * (1 + a.ups);
  a.W = Ws * (1 + a.ups);
  a.Cout = Cout;
  a.CoutP = ra_conv_cout_padded(Cout);
  a.relu = relu;
  a.pool = pool;
  if (!a.CoutP) return ra::fail(RA_E_SHAPE, "ra_conv3x3_f32: Cout %d", Cout);
  if (pool != 1 && pool != 2) return ra::fail(RA_E_SHAPE, "ra_conv3x3_f32: pool %d", pool);
  if (pool == 2 && ((a.H | a.W) & 1)) return ra::fail(RA_E_SHAPE, "ra_conv3x3_f32: odd size with pool 2");
  a.Ho = a.H / pool;
  a.Wo = a.W / pool;
  const int Cin = C0 + C1;
  const int CK = ra::conv::chunk_of(Cin);
  // a chunk may not straddle the src0/src1 boundary at finer than 4 channels (always true) but
  // the chunk index arithmetic needs C0 % 4 == 0 only: chunks are resolved per channel group.
  hipStream_t st = ra::as_stream(stream);
  switch (CK) {
    case 16: return ra::conv::dispatch_cout<16>(a, B, st);
    case 8: return ra::conv::dispatch_cout<8>(a, B, st);
    default: return ra::conv::dispatch_cout<4>(a, B, st);
  }
}
