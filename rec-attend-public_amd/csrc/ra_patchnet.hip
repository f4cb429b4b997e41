// K4 — the whole patch-sized network of one timestep in ONE launch:
//   attention CNN   nnlib.cnn  run_cnn  (nnlib.py:214-255)   conv3x3 + b + BN(tt) + ReLU + max-pool
//   attention DCNN  nnlib.dcnn run_dcnn (nnlib.py:339-402)   conv2d_transpose SAME (stride 1|2) + b + BN + ReLU
//   score MLP       full_model.py:794,821-822                 s = sigmoid([h | h_core] . w + b)
// i.e. full_model.py:792-807,821-822 between extract_patch and the paste.
//
// Why one launch: at 48x48 these 13 layers are 23.5 MFLOP per image — microseconds of MFMA work —
// but as 13 dependent launches each costs a kernel start (weights, tile, LDS, MFMA, store: 4.7-6.5 us)
// plus a stream boundary (~1.5 us): ~85 us per timestep of pure latency (profiles/r01).  Here a
// group of kNW workgroups per image walks a PHASE list:
//   * consecutive layers at one resolution are chained inside a workgroup — the intermediate of a
//     tile (+ halo, recomputed) never leaves LDS;
//   * between phases the activations go through L2 with write-through (sc1) stores, a per-image
//     arrival counter (agent-scope atomics) and sc1 loads — the producer/consumer form of
//     cdna_hip_programming.md Guideline 16 (no dependence on dispatch order or XCD placement);
//   * >= 24-pixel resolutions are split 4 x 4 tiles over the group; smaller ones are computed from
//     a full copy of the (tiny) input, the MFMA (pixel tile x cout tile) tasks split over the
//     group's 64 waves.
// Same arithmetic as ra_conv.hip: v_mfma_f32_16x16x4_f32 implicit GEMM, k = (ky, kx, ci), packed
// weights of ra_conv_pack_weights, bias + BN folded to scale/shift per timestep; a stride-2
// transposed conv is the SAME conv of the zero-stuffed input with the flipped filter.
#include <cstdlib>
#include <cstring>

#include "ra_common.h"

namespace ra {
namespace pnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kNW = 16;        // workgroups per image: 4 x 4 tiles
constexpr int kMaxLayers = 16;
constexpr int kMaxChain = 3;   // layers chained through LDS in one phase
constexpr int kSC1 = 16;       // buffer-instruction aux bit: sc1 (device-coherent, write-through)
constexpr int kOOB = 0x7fffffff;
constexpr int kSpinLimit = 1 << 21;

struct Layer {
  const float *wp, *scale, *shift;  // scale / shift already point at this timestep's row
  float *out;                       // global output (phase-final layers), else nullptr
  int NCG;                          // input channels / 4 (1, 2, 4, 8)
  int Cout, CoutP;
  int ups, pool, relu;
  int H, W;                         // conv resolution (after zero-stuffing, before pooling)
  int out_bytes;
};

struct Phase {
  const float *src;     // global input of the first layer [B, Hs, Ws, Cs]
  int src_bytes;
  int Hs, Ws, Cs;
  int first, n;         // layers [first, first + n)
  int share;            // 1: one tile per workgroup; kNW: whole image, tasks split over the group
  int TH, TW, tiles_x;  // tile size in conv pixels
};

struct Args {
  Layer L[kMaxLayers];
  Phase P[kMaxLayers];
  int nphases, B, lds_half;
  int *cnt, *done, *status;
  // score rider (nullable): s = sigmoid([h | core] . w + bias)
  const float *h, *core, *sw, *sbias;
  float *s_out;
  int K0, K1, core_bytes;
  long s_stride_b;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// Global -> LDS staging of one input region (conv coordinates [ry0, ry0+rh) x [rx0, rx0+rw)),
// zero outside the image and, for a stride-2 transposed conv, on the stuffed positions
// U[2i+1, 2j+1] = x[i, j].  LDS record of a pixel: [ksub 0..3][cg] (channel = 4*cg + ksub), so one
// wide ds_read fetches a lane's A operands of all k-steps of a tap.
template <int NCG, bool SC1>
__device__ __forceinline__ void stage(const Phase &P, int ups, int RH, int RW, int b, int ry0, int rx0, int rh, int rw,
                             float *lds) {
  const __amdgpu_buffer_rsrc_t rs = rsrc(P.src, P.src_bytes);
  const int npix = rh * rw;
  for (int e0 = 0; e0 < npix; e0 += 256) {
    const int e = e0 + threadIdx.x;
    const int y = e / rw, x = e - y * rw;
    const int Y = ry0 + y, X = rx0 + x;
    bool ok = (e < npix) & (Y >= 0) & (Y < RH) & (X >= 0) & (X < RW);
    int ys = Y, xs = X;
    if (ups) {
      ok = ok & (Y & 1) & (X & 1);
      ys = (Y - 1) >> 1;
      xs = (X - 1) >> 1;
    }
    const int off = ok ? (((b * P.Hs + ys) * P.Ws + xs) * P.Cs) * 4 : kOOB;
    f32x4 v[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
      v[cg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16 * cg, SC1 ? kSC1 : 0));
    if (e < npix) {
      float *rec = lds + e * (4 * NCG);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) rec[ks * NCG + cg] = v[cg][ks];
    }
  }
}

template <int N>
struct vec_of {
  typedef float type __attribute__((ext_vector_type(N)));
};

// One conv layer over an LDS-resident input region.  src: (h+2) x (w+2) pixel records (the output
// region + 1-pixel halo), output region = conv coordinates [y0, y0+h) x [x0, x0+w).
// Tasks = (16-pixel M tile, 16-cout N tile); wave `gw` of `gstride` takes tasks gw, gw+gstride, ...
// (gstride is a multiple of the N-tile count, so a wave keeps one B operand in registers).
//   dst != nullptr: result -> LDS records of the next layer (CnN channels), zero outside the image
//   else          : result -> global L.out (max-pooled if L.pool == 2), write-through stores
template <int NCG>
struct BOp {  // a wave's B operand (9 taps x NCG k-steps of one 16-cout tile) + its epilogue constants
  float w[9 * NCG];
  float sc, sh;
  int ng;
};

// Issued BEFORE the wait for the previous phase: the weights do not depend on it, so their L2 / HBM
// round trip hides behind the inter-workgroup hand-off instead of following it.
template <int NCG>
__device__ __forceinline__ void load_b(const Layer &L, int ng, BOp<NCG> &B) {
  constexpr int CK = NCG >= 4 ? 16 : 4 * NCG, NCGc = CK / 4;
  const int lane = threadIdx.x & 63;
  const int ksub = lane >> 4, n = lane & 15;
  const float *wrow = L.wp + (size_t)ksub * L.CoutP + 16 * ng + n;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
      B.w[tap * NCG + cg] = wrow[(size_t)((((cg / NCGc) * 9 + tap) * NCGc + (cg % NCGc)) * 4) * L.CoutP];
  B.sc = L.scale[16 * ng + n];
  B.sh = L.shift[16 * ng + n];
  B.ng = ng;
}

template <int NCG, bool SC1>
__device__ __forceinline__ void conv_layer(const Layer &L, const float *src, int y0, int x0, int h, int w, float *dst,
                                  int CnN, int b, int gw, int gstride, BOp<NCG> &B) {
  typedef typename vec_of<NCG>::type avec;
  constexpr int Cin = 4 * NCG;
  constexpr int AUX = SC1 ? kSC1 : 0;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, ksub = lane >> 4, n = lane & 15, qo = lane >> 4;
  const int sw = w + 2;
  const int NNT = L.CoutP >> 4;
  const bool pl = L.pool == 2;
  const int wq = pl ? (w >> 1) : w;
  const int nunits = pl ? (h >> 1) * wq : h * w;  // pooled pixels / pixels
  const int upt = pl ? 4 : 16;                    // units per M tile
  const int ntasks = ((nunits + upt - 1) / upt) * NNT;
  const float lo = L.relu ? 0.f : -__builtin_inff();
  const __amdgpu_buffer_rsrc_t ry = rsrc(L.out, L.out ? L.out_bytes : 0);
  for (int t = gw; t < ntasks; t += gstride) {
    const int ng = t % NNT, mt = t / NNT;
    if (ng != B.ng) load_b<NCG>(L, ng, B);
    const float sc = B.sc, sh = B.sh;
    int u = pl ? mt * 4 + (m >> 2) : mt * 16 + m;
    u = u < nunits ? u : nunits - 1;
    const int uy = u / wq, ux = u - uy * wq;
    const int ay = pl ? 2 * uy + ((m >> 1) & 1) : uy, ax = pl ? 2 * ux + (m & 1) : ux;
    const float *ap = src + (ay * sw + ax) * Cin + ksub * NCG;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const avec av = *reinterpret_cast<const avec *>(ap + ((tap / 3) * sw + (tap % 3)) * Cin);
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cg], B.w[tap * NCG + cg], acc, 0, 0, 0);
    }
    const int co = 16 * ng + n;
    if (pl) {
      float v = fmaxf(fmaxf(acc[0] * sc + sh, acc[1] * sc + sh), fmaxf(acc[2] * sc + sh, acc[3] * sc + sh));
      v = fmaxf(v, lo);
      const int uo = mt * 4 + qo;
      if (uo < nunits && co < L.Cout) {
        const int oy = uo / wq, ox = uo - oy * wq;
        const int gy = (y0 >> 1) + oy, gx = (x0 >> 1) + ox;
        const int off = (((b * (L.H >> 1) + gy) * (L.W >> 1) + gx) * L.Cout + co) * 4;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, off, 0, AUX);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uo = mt * 16 + 4 * qo + r;
        if (uo < nunits) {
          const int oy = uo / w, ox = uo - oy * w;
          const int gy = y0 + oy, gx = x0 + ox;
          const bool inside = (gy >= 0) & (gy < L.H) & (gx >= 0) & (gx < L.W);
          float v = fmaxf(acc[r] * sc + sh, lo);
          if (dst) {  // the next layer's SAME padding is zero, and so are its padded channels
            v = (inside & (co < L.Cout)) ? v : 0.f;
            if (co < CnN) dst[(oy * w + ox) * CnN + (co & 3) * (CnN >> 2) + (co >> 2)] = v;
          } else if (inside & (co < L.Cout)) {
            const int off = (((b * L.H + gy) * L.W + gx) * L.Cout + co) * 4;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, off, 0, AUX);
          }
        }
      }
    }
  }
}

// One phase for a workgroup whose first layer has 4*NCG input channels: prefetch that layer's B
// operand, wait for the producers (fused launch only), stage, then run the chained layers.
// The phase / layer records are passed BY VALUE (scalar registers): taking the address of the
// kernel-argument struct would make the compiler copy all of it to scratch memory.
template <int NCG, bool SC1>
__device__ __forceinline__ void run_phase(const Phase P, const Layer L0, const Layer L1, const Layer L2, int lds_half,
                                          int *status, int wait_for, int b, int wg, int wave, int *cnt,
                                          float *lds) {
  const int gw = P.share == 1 ? wave : wg * 4 + wave, gstride = 4 * P.share;
  BOp<NCG> B0;
  load_b<NCG>(L0, gw % (L0.CoutP >> 4), B0);
  if (wait_for > 0) {  // every workgroup of this image has published the previous phase
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_for) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) {
          if (status) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
  const int tile = P.share == 1 ? wg : 0;
  const int ty0 = (tile / P.tiles_x) * P.TH, tx0 = (tile % P.tiles_x) * P.TW;
  const int n = P.n;
  // staged input: the tile grown by one pixel per chained layer
  stage<NCG, SC1>(P, L0.ups, L0.H, L0.W, b, ty0 - n, tx0 - n, P.TH + 2 * n, P.TW + 2 * n, lds);
  __syncthreads();
  {
    const int g = n - 1;
    conv_layer<NCG, SC1>(L0, lds, ty0 - g, tx0 - g, P.TH + 2 * g, P.TW + 2 * g, n > 1 ? lds + lds_half : nullptr,
                         n > 1 ? 4 * L1.NCG : 0, b, gw, gstride, B0);
  }
  if (n > 1) {  // second layer: LDS -> LDS (-> third) or -> global
    __syncthreads();
    const int g = n - 2;
    float *dst = n > 2 ? lds : nullptr;
    const int CnN = n > 2 ? 4 * L2.NCG : 0;
    const float *src = lds + lds_half;
    const int y0 = ty0 - g, x0 = tx0 - g, h = P.TH + 2 * g, w = P.TW + 2 * g;
    if (L1.NCG == 2) {
      BOp<2> B;
      B.ng = -1;
      conv_layer<2, SC1>(L1, src, y0, x0, h, w, dst, CnN, b, gw, gstride, B);
    } else if (L1.NCG == 4) {
      BOp<4> B;
      B.ng = -1;
      conv_layer<4, SC1>(L1, src, y0, x0, h, w, dst, CnN, b, gw, gstride, B);
    } else {
      BOp<8> B;
      B.ng = -1;
      conv_layer<8, SC1>(L1, src, y0, x0, h, w, dst, CnN, b, gw, gstride, B);
    }
  }
  if (n > 2) {  // third layer -> global; its input has 8 channels on every supported net (checked on the host)
    __syncthreads();
    BOp<2> B;
    B.ng = -1;
    conv_layer<2, SC1>(L2, lds, ty0, tx0, P.TH, P.TW, nullptr, 0, b, gw, gstride, B);
  }
}

// FUSED = true : phases [p_begin, p_end) in ONE launch, exchanged through L2 (sc1 + counters)
// FUSED = false: the launch runs exactly one phase; the stream boundary is the synchronisation
template <bool FUSED>
__global__ __launch_bounds__(256) void patchnet_kernel(const Args a, int p_begin, int p_end) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x / kNW, wg = blockIdx.x - b * kNW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int *cnt = a.cnt + b;
  for (int p = p_begin; p < p_end; ++p) {
    const Phase P = a.P[p];
    const int l1 = P.n > 1 ? P.first + 1 : P.first, l2 = P.n > 2 ? P.first + 2 : P.first;
    const Layer L0 = a.L[P.first], L1 = a.L[l1], L2 = a.L[l2];
    const int wait_for = (FUSED && p > p_begin) ? kNW * (p - p_begin) : 0;
    switch (L0.NCG) {
      case 1: run_phase<1, FUSED>(P, L0, L1, L2, a.lds_half, a.status, wait_for, b, wg, wave, cnt, lds); break;
      case 2: run_phase<2, FUSED>(P, L0, L1, L2, a.lds_half, a.status, wait_for, b, wg, wave, cnt, lds); break;
      case 4: run_phase<4, FUSED>(P, L0, L1, L2, a.lds_half, a.status, wait_for, b, wg, wave, cnt, lds); break;
      default: run_phase<8, FUSED>(P, L0, L1, L2, a.lds_half, a.status, wait_for, b, wg, wave, cnt, lds); break;
    }
    if (FUSED && p + 1 < p_end) {
      // publish: every storing wave drains its write-through stores, then ONE arrival per workgroup
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (p_end != a.nphases) return;
  // score rider: one workgroup per image, after the core layer's phase (which it has waited for)
  if (a.s_out && wg == kNW - 1) {
    __shared__ float red[4];
    const __amdgpu_buffer_rsrc_t rc = rsrc(a.core, a.core_bytes);
    float s = 0.f;
    for (int k = threadIdx.x; k < a.K0; k += 256) s += a.h[(size_t)b * a.K0 + k] * a.sw[k];
    for (int k = threadIdx.x; k < a.K1; k += 256) {
      const float cv = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rc, (b * a.K1 + k) * 4, 0, FUSED ? kSC1 : 0));
      s += cv * a.sw[a.K0 + k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float z = red[0] + red[1] + red[2] + red[3] + a.sbias[0];
      a.s_out[(size_t)b * a.s_stride_b] = 1.f / (1.f + __expf(-z));
    }
  }
  // the last workgroup of the image to finish re-arms the counter for the next launch / replay
  if (FUSED && threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(a.done + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == kNW - 1) {
      __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.done + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------ host
struct Plan {
  int nphases = 0;
  int first[kMaxLayers], n[kMaxLayers], share[kMaxLayers], TH[kMaxLayers], TW[kMaxLayers], tiles_x[kMaxLayers];
  int convH[kMaxLayers], convW[kMaxLayers];  // per LAYER
  int inH[kMaxLayers], inW[kMaxLayers];      // per LAYER: source resolution
  size_t inter_off[kMaxLayers];              // per LAYER: float offset of its global output in ws, or ~0
  size_t inter_floats = 0;
  int lds_half = 0;
};

inline int r4(int c) { return (c + 3) & ~3; }

// Phase list: chain consecutive layers at one resolution (only the first may upsample, only the
// last may pool) while the resolution allows 4 x 4 tiles; small resolutions are single layers.
int make_plan(const ra_pnet_layer *ls, int nl, int Hp, int Wp, int C0, int B, Plan &pl) {
  if (!ls || nl < 1 || nl > kMaxLayers || Hp < 4 || Wp < 4) return RA_E_INVALID;
  int h = Hp, w = Wp, cin = C0;
  for (int i = 0; i < nl; ++i) {
    const ra_pnet_layer &l = ls[i];
    if (l.Cin != r4(cin) || (l.Cin != 4 && l.Cin != 8 && l.Cin != 16 && l.Cin != 32)) return RA_E_SHAPE;
    if (l.Cout < 1 || l.Cout > 32 || (l.pool != 1 && l.pool != 2)) return RA_E_SHAPE;
    if (i + 1 < nl && (l.Cout & 3)) return RA_E_SHAPE;
    pl.inH[i] = h;
    pl.inW[i] = w;
    h *= l.upsample ? 2 : 1;
    w *= l.upsample ? 2 : 1;
    pl.convH[i] = h;
    pl.convW[i] = w;
    if (l.pool == 2 && ((h | w) & 1)) return RA_E_SHAPE;
    h /= l.pool;
    w /= l.pool;
    cin = l.Cout;
  }
  int i = 0, lds_half = 0;
  pl.inter_floats = 0;
  while (i < nl) {
    const int p = pl.nphases++;
    const int ch = pl.convH[i], cw = pl.convW[i];
    const bool tiled = ch >= 24 && cw >= 24 && (ch % 8) == 0 && (cw % 8) == 0;
    int n = 1;
    if (tiled)
      while (n < kMaxChain && i + n < nl && ls[i + n - 1].pool == 1 && !ls[i + n].upsample) ++n;
    pl.first[p] = i;
    pl.n[p] = n;
    pl.share[p] = tiled ? 1 : kNW;
    pl.TH[p] = tiled ? ch / 4 : ch;
    pl.TW[p] = tiled ? cw / 4 : cw;
    pl.tiles_x[p] = tiled ? 4 : 1;
    if (n > 2 && ls[i + 2].Cin != 8) n = 2;   // the kernel's third chained layer is the 8-channel form
    if (n > 1 && ls[i + 1].Cin == 4) n = 1;   // and its second one takes 8 / 16 / 32 channels
    pl.n[p] = n;
    for (int k = 0; k < n; ++k) {  // LDS regions: staged input, then every chained intermediate
      const int g = n - k;         // growth of layer k's INPUT region
      const int fl = (pl.TH[p] + 2 * g) * (pl.TW[p] + 2 * g) * ls[i + k].Cin;
      if (fl > lds_half) lds_half = fl;
      pl.inter_off[i + k] = ~(size_t)0;
    }
    const int last = i + n - 1;
    if (last < nl - 1) {
      pl.inter_off[last] = pl.inter_floats;
      const size_t fl = (size_t)B * (pl.convH[last] / ls[last].pool) * (pl.convW[last] / ls[last].pool) * ls[last].Cout;
      pl.inter_floats += (fl + 3) & ~(size_t)3;
    }
    i += n;
  }
  pl.lds_half = (lds_half + 3) & ~3;
  return 0;
}

constexpr size_t kCtrBytes = 1024;  // cnt[B] | done[B] (B <= 64), then the intermediates

inline int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace pnet
}  // namespace ra

using namespace ra;

extern "C" size_t ra_patchnet_workspace_bytes(const ra_pnet_layer *layers, int n_layers, int B, int Hp, int Wp) {
  pnet::Plan pl;
  if (!layers || n_layers < 1 || B < 1 || B > 64) return 0;
  if (pnet::make_plan(layers, n_layers, Hp, Wp, layers[0].Cin, B, pl) != 0) return 0;
  return pnet::kCtrBytes + pl.inter_floats * sizeof(float);
}

extern "C" int ra_patchnet_supported(const ra_pnet_layer *layers, int n_layers, int B, int Hp, int Wp) {
  pnet::Plan pl;
  if (!layers || n_layers < 1 || B < 1 || B > 64) return 0;
  if (pnet::make_plan(layers, n_layers, Hp, Wp, layers[0].Cin, B, pl) != 0) return 0;
  if ((size_t)pl.lds_half * 2 * sizeof(float) > 64 * 1024) return 0;
  // every workgroup of the launch must be co-resident (they wait for each other): one per CU is
  // always admitted at <= 64 KiB of LDS, two per CU with half of that
  const int per_cu = (size_t)pl.lds_half * 2 * sizeof(float) <= 32 * 1024 ? 2 : 1;
  return B * pnet::kNW <= per_cu * pnet::num_cus();
}

extern "C" int ra_patchnet_f32(const ra_pnet_layer *layers, int n_layers, int core_layer, const float *x, int B,
                               int Hp, int Wp, int tt, float *y, const float *h, int K0, const float *sw,
                               const float *sbias, float *s_out, size_t s_stride_b, void *ws, size_t ws_bytes,
                               int *status_dev, void *stream) {
  if (!layers || !x || !y || !ws || B < 1 || B > 64 || tt < 0)
    return fail(RA_E_INVALID, "ra_patchnet_f32: bad argument");
  pnet::Plan pl;
  const int rc = pnet::make_plan(layers, n_layers, Hp, Wp, layers[0].Cin, B, pl);
  if (rc) return fail(rc, "ra_patchnet_f32: unsupported layer list");
  if (!ra_patchnet_supported(layers, n_layers, B, Hp, Wp))
    return fail(RA_E_SHAPE, "ra_patchnet_f32: tile does not fit LDS or %d images exceed the co-resident grid", B);
  if (ws_bytes < pnet::kCtrBytes + pl.inter_floats * sizeof(float))
    return fail(RA_E_WORKSPACE, "ra_patchnet_f32: workspace too small");
  if (s_out && (!h || !sw || !sbias || core_layer < 0 || core_layer >= n_layers - 1 ||
                pl.inter_off[core_layer] == ~(size_t)0))
    return fail(RA_E_INVALID, "ra_patchnet_f32: score rider needs h, w, bias and a phase-final core layer");
  pnet::Args a;
  memset(&a, 0, sizeof(a));
  float *inter = reinterpret_cast<float *>(static_cast<char *>(ws) + pnet::kCtrBytes);
  for (int i = 0; i < n_layers; ++i) {
    const ra_pnet_layer &l = layers[i];
    pnet::Layer &L = a.L[i];
    const int cp = ra_conv_cout_padded(l.Cout);
    L.wp = l.wpacked;
    L.scale = l.scale + (size_t)tt * cp;
    L.shift = l.shift + (size_t)tt * cp;
    L.NCG = l.Cin / 4;
    L.Cout = l.Cout;
    L.CoutP = cp;
    L.ups = l.upsample ? 1 : 0;
    L.pool = l.pool;
    L.relu = l.relu;
    L.H = pl.convH[i];
    L.W = pl.convW[i];
    const size_t ob = (size_t)B * (L.H / l.pool) * (L.W / l.pool) * l.Cout * sizeof(float);
    if (ob >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_patchnet_f32: activation exceeds 2 GiB");
    L.out_bytes = (int)ob;
    L.out = (i == n_layers - 1) ? y : (pl.inter_off[i] != ~(size_t)0 ? inter + pl.inter_off[i] : nullptr);
  }
  for (int p = 0; p < pl.nphases; ++p) {
    pnet::Phase &P = a.P[p];
    const int f = pl.first[p];
    P.first = f;
    P.n = pl.n[p];
    P.share = pl.share[p];
    P.TH = pl.TH[p];
    P.TW = pl.TW[p];
    P.tiles_x = pl.tiles_x[p];
    P.src = f == 0 ? x : a.L[f - 1].out;
    P.Hs = pl.inH[f];
    P.Ws = pl.inW[f];
    P.Cs = layers[f].Cin;
    P.src_bytes = (int)((size_t)B * P.Hs * P.Ws * P.Cs * sizeof(float));
    if (f > 0 && (layers[f - 1].Cout & 3)) return fail(RA_E_SHAPE, "ra_patchnet_f32: inner Cout %% 4");
  }
  a.nphases = pl.nphases;
  a.B = B;
  a.lds_half = pl.lds_half;
  a.cnt = static_cast<int *>(ws);
  a.done = a.cnt + 64;
  a.status = status_dev;
  if (s_out) {
    a.h = h;
    a.sw = sw;
    a.sbias = sbias;
    a.s_out = s_out;
    a.s_stride_b = (long)s_stride_b;
    a.K0 = K0;
    a.core = a.L[core_layer].out;
    a.K1 = a.L[core_layer].out_bytes / (int)sizeof(float) / B;
    a.core_bytes = a.L[core_layer].out_bytes;
  }
  const size_t lds = (size_t)pl.lds_half * 2 * sizeof(float);
  static bool attr = false;
  static int mode = -1;  // RA_PNET_MODE: 0 = one launch per phase (default), 1 = one launch, L2 hand-offs
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pnet::patchnet_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pnet::patchnet_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const char *e = getenv("RA_PNET_MODE");
    mode = e ? atoi(e) : 0;
    attr = true;
  }
  if (mode == 1) {
    hipLaunchKernelGGL(pnet::patchnet_kernel<true>, dim3(B * pnet::kNW), dim3(256), lds, as_stream(stream), a, 0,
                       pl.nphases);
    return launch_status("ra_patchnet_f32");
  }
  for (int p = 0; p < pl.nphases; ++p) {
    hipLaunchKernelGGL(pnet::patchnet_kernel<false>, dim3(B * pnet::kNW), dim3(256), lds, as_stream(stream), a, p,
                       p + 1);
    const int rc2 = launch_status("ra_patchnet_f32");
    if (rc2) return rc2;
  }
  return 0;
}
