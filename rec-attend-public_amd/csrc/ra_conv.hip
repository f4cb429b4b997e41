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
#include <cstdlib>

#include "ra_common.h"

// This file is compiled three times (Makefile): RA_K1_PART 0 = the decode loop's float32 kernels + every host entry
// point, 1 = the bf16-operand variants, 2 = float32 with batch moments in the epilogue (the training forward).  The
// three sets of template instantiations build in parallel; part 0's entry forwards to the other parts' dispatchers.
#ifndef RA_K1_PART
#define RA_K1_PART 0
#endif

namespace ra {
namespace conv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  int bytes0, bytes1, bytes_y;  // tensor sizes for the buffer descriptors (each < 2 GiB)
  const float *plane;           // optional [B,Hs,Ws] plane that REPLACES input channel plane_chan
  int plane_chan, bytes_p;      // (the canvas, kept outside the packed image)
  int bf16;                     // 1: bf16 operands, float32 accumulation
  int in_bf16, out_bf16;        // bf16 kernels only: src0 / src1, respectively y, are STORED as bf16 (2 bytes per element)
  float *mom_part;              // MOM kernels: per-(workgroup, wave row) channel sums of the pre-activation output
  int *nparts_out;              // host: the number of partial records the launch writes (grid * WM)
  int prio;                     // wave priority (s_setprio) of a patch-sized launch: the decode loop's latency-bound tail (ra_common.h)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kOOB = 0x7fffffff;  // a buffer offset past every tensor: loads return 0, stores drop
template <int CTRL>
__device__ inline float quad_swap(float v) {  // DPP quad permute: VALU rate, no LDS crossbar
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

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
  // LDS record of one pixel: [ksub 0..3][cg 0..NCG-1] (channel = 4*cg + ksub), so ONE wide
  // ds_read (b128 for CK=16, b64 for CK=8) fetches a lane's A operands of all NCG k-steps of a
  // tap.
  static constexpr int PIX = CK;  // unpadded: modelled b128/b64 conflicts are lowest at 16 / 8
  static constexpr int KS = 9 * NCG;           // MFMA k-steps per chunk
  static constexpr int LDS_FLOATS = LH * LW * PIX;
};

// Persistent, software-pipelined form: a workgroup walks tiles t = blockIdx.x, +gridDim.x, ...;
// the work items are (tile, Cin chunk).  While the MFMAs of item i run out of LDS buffer i&1, the
// global loads of item i+1 are already in flight into registers; they are written to the other LDS
// buffer after the MFMA loop, followed by the only barrier of the item.  Single-chunk layers keep
// their B operand in registers for the whole launch.
// SWAP = false: pixels are the MFMA A operand -> a lane's 4 accumulators are the 2x2 pool window of
//   one channel (pool layers: in-register max, one 4-byte store per lane).
// SWAP = true : weights are the A operand -> a lane's 4 accumulators are 4 consecutive channels of
//   one pixel (no-pool layers: one 16-byte store per lane instead of four 4-byte stores).
// BF16 = true (the training step's mixed-precision mode, model_opt['compute_dtype'] = 'bf16'): the same kernel with
// bf16 OPERANDS and float32 accumulation — the staged float32 pixels are rounded to bf16 (v_cvt_pk_bf16_f32, RNE)
// as they leave LDS, the chunk's weights once per chunk, and four k-steps of v_mfma_f32_16x16x4_f32 become ONE
// v_mfma_f32_16x16x32_bf16 (gfx950's K = 32 form: two quads of 4 k-values per lane; a quad = the lane's ksub in 4 k-steps:
// 4 channel groups of a tap for CK = 16,
// 2 taps x 2 groups for CK = 8, 4 taps for CK = 4; steps beyond the 9 taps carry zero weights).  Tensors in HBM and
// LDS stay float32; with 1/8 of the matrix-pipe time the kernel is bound by staging, LDS and HBM instead.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
// gfx950's K = 32 form, v_mfma_f32_16x16x32_bf16: a lane holds 8 bf16 per operand.  Two of the K = 16 quads side by side:
// slot j of A-lane (m, kb) meets slot j of B-lane (n, kb), so ANY assignment of k-values to slots that both operands share
// is a valid contraction — here slots 0..3 = the first quad's k-steps, 4..7 = the second's.
typedef short bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ inline f32x4 mfma32(bf16x4 a0, bf16x4 a1, bf16x4 b0, bf16x4 b1, f32x4 c) {
  const bf16x8s A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), B = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), c, 0, 0, 0);
}
// four bf16 values stored in 8 bytes -> float32 (exact)
__device__ inline f32x4 bf16x4_to_f32(u32x2s p) {
  return f32x4{__builtin_bit_cast(float, p.x << 16), __builtin_bit_cast(float, p.x & 0xffff0000u),
               __builtin_bit_cast(float, p.y << 16), __builtin_bit_cast(float, p.y & 0xffff0000u)};
}
__device__ inline bf16x4 pack_bf16(float v0, float v1, float v2, float v3) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v2, v3}, bf16x2));
  return __builtin_bit_cast(bf16x4, u32x2{lo, hi});
}

// MOM = true (training, SWAP layers without pooling): the epilogue also accumulates, per lane, the count, sum and sum
// of squares of every output value u = acc * scale + shift (BEFORE the ReLU) about a per-wave pivot — the wave's first
// output of that channel — and the wave leaves one record {n, S1, S2, pivot} per channel in a.mom_part
// [(workgroup * WM + wave row)][CoutP][4].  tf.nn.moments of the layer (nnlib.py:98) then costs one small finishing
// launch (ra_bn_moments_from_partials_f32: the records re-based onto one reference in float64) instead of two more passes over u.
// UPS = true (round 6; SWAP layers on the (2, 2) geometry): the SUB-PIXEL form of a stride-2 transposed conv.  The staged tile is
// still the zero-stuffed image U[2i+1, 2j+1] = x[i, j], but a wave's four pixel groups are no longer four 8 x 2 blocks of its
// 16 x 4 region: group g = (py, px) is the region's 16 pixels of ONE parity class (rows 2 dy + py, columns 2 (2 q + dx) + px).
// Tile origins are even, so for such a pixel only the taps with ky = py and kx = px (mod 2) meet stuffed data — 4, 2, 2 and 1
// taps for the four classes, 9 per 64 pixels instead of 36: a quarter of the MFMAs and of the A-operand reads, same results
// (the skipped products are exact zeros).
template <int CK, int NC, int WN, int GX, int GY, bool SWAP, bool BF16 = false, bool MOM = false, bool UPS = false>
__global__ __launch_bounds__(256, (GX * GY * NC > 8) ? 1 : 2) void conv3x3_mfma(const Args a, int tiles_x, int tiles_y, int ntiles) {
  static_assert(!MOM || SWAP, "batch moments ride on the channel-vector epilogue");
  static_assert(!UPS || (SWAP && GX == 2 && GY == 2 && !BF16 && !MOM), "sub-pixel form: float32 SWAP layers on the (2, 2) geometry");
  raise_prio(a.prio);
  using G = Geo<CK, NC, WN, GX, GY>;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // 2 * G::LDS_FLOATS
  constexpr int NPIX = G::LH * G::LW;        // pixel records of one staged chunk
  constexpr int NST = (NPIX + 255) / 256;    // pixels per thread
  typedef float avec __attribute__((ext_vector_type(G::NCG)));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wm = wave / WN;
  const int Cin = a.C0 + a.C1;
  const int nchunks = Cin / CK;

  // A-operand lane geometry: m = lane & 15 -> (q, dy, dx); ksub = lane >> 4.
  const int m = lane & 15, ksub = lane >> 4;
  const int q = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
  const int a_base = UPS ? ((wm * G::WR + 2 * dy) * G::LW + 2 * (2 * q + dx)) * G::PIX + ksub * G::NCG
                         : ((wm * G::WR + dy) * G::LW + 2 * q + dx) * G::PIX + ksub * G::NCG;  // floats
  const int co_lane = lane & 15;
  const int qo = lane >> 4;  // D rows 4*qo + r -> pooled x position qo, window element r

  // multi-chunk layers with one cout group per wave prefetch the NEXT chunk's B operand into a
  // second register set while the MFMAs of the current chunk run (36 more VGPRs); otherwise the
  // weight reload between chunks is an exposed L2 round trip on every item
  constexpr bool BDB = (NC == 1) && (G::PM <= 4) && (CK == 16);
  f32x4 acc[G::PM][NC];
  float breg[G::KS][NC];
  float bnext[BDB ? G::KS : 1][NC];
  constexpr int NMF = (G::KS + 3) / 4;  // bf16 MFMAs per chunk and (pixel group, cout group)
  bf16x4 bpk[BF16 ? NMF : 1][NC];
  // The MFMA is issued as D = W^T-slice x pixels (weights are the A operand), so a lane holds
  // pixel (lane & 15) and, in its 4 accumulator registers, output channels 4*(lane>>4)..+3:
  // the epilogue stores one float4 per lane.  Per-lane epilogue constants, loaded once:
  f32x4 sc4[NC], sh4[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    if constexpr (SWAP) {
      sc4[n] = *reinterpret_cast<const f32x4 *>(a.scale + 16 * (wn * NC + n) + 4 * ksub);
      sh4[n] = *reinterpret_cast<const f32x4 *>(a.shift + 16 * (wn * NC + n) + 4 * ksub);
    } else {
      const float sc = a.scale[16 * (wn * NC + n) + co_lane], sh = a.shift[16 * (wn * NC + n) + co_lane];
      sc4[n] = f32x4{sc, sc, sc, sc};
      sh4[n] = f32x4{sh, sh, sh, sh};
    }
  }
  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.src0, a.bytes0);
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.src1 ? a.src1 : a.src0, a.src1 ? a.bytes1 : 0);
  const __amdgpu_buffer_rsrc_t rsy = make_rsrc(a.y, a.bytes_y);
  const __amdgpu_buffer_rsrc_t rsp = make_rsrc(a.plane ? a.plane : a.src0, a.plane ? a.bytes_p : 0);
  f32x4 st[NST][G::NCG];
  u32x2s straw[BF16 ? NST : 1][BF16 ? G::NCG : 1];  // bf16 input: the loaded bits, untouched until store_item (any use at the load
                                                    // makes the compiler wait for each load before it issues the next)
  f32x4 ms1[MOM ? NC : 1], ms2[MOM ? NC : 1], mpv[MOM ? NC : 1];  // moments about the pivot, per lane
  float mcnt = 0.f;
  bool mhave = false;  // the pivots are set by the first tile
  if constexpr (MOM) {
#pragma unroll
    for (int n = 0; n < NC; ++n) ms1[n] = ms2[n] = mpv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto zero_acc = [&]() {
#pragma unroll
    for (int g = 0; g < G::PM; ++g)
#pragma unroll
      for (int n = 0; n < NC; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto load_b = [&](int ch) {  // B operand of one chunk -> registers (L2-resident packed weights)
    const float *wrow = a.wp + ((size_t)ch * G::KS * 4 + ksub) * a.CoutP + 16 * (wn * NC) + co_lane;
#pragma unroll
    for (int s = 0; s < G::KS; ++s)
#pragma unroll
      for (int n = 0; n < NC; ++n) breg[s][n] = wrow[(size_t)s * 4 * a.CoutP + 16 * n];
  };
  auto load_b_next = [&](int ch) {
    if constexpr (BDB) {
      const float *wrow = a.wp + ((size_t)ch * G::KS * 4 + ksub) * a.CoutP + 16 * (wn * NC) + co_lane;
#pragma unroll
      for (int s = 0; s < G::KS; ++s)
#pragma unroll
        for (int n = 0; n < NC; ++n) bnext[s][n] = wrow[(size_t)s * 4 * a.CoutP + 16 * n];
    }
  };
  auto swap_b = [&]() {
    if constexpr (BDB) {
#pragma unroll
      for (int s = 0; s < G::KS; ++s)
#pragma unroll
        for (int n = 0; n < NC; ++n) breg[s][n] = bnext[s][n];
    }
  };
  auto pack_b = [&]() {  // BF16: the chunk's weights as bf16 quads (k-steps 4f .. 4f+3 of this lane's ksub)
    if constexpr (BF16) {
#pragma unroll
      for (int f = 0; f < NMF; ++f)
#pragma unroll
        for (int n = 0; n < NC; ++n) {
          float w4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) w4[j] = (4 * f + j < G::KS) ? breg[(4 * f + j < G::KS) ? 4 * f + j : 0][n] : 0.0f;
          bpk[f][n] = pack_bf16(w4[0], w4[1], w4[2], w4[3]);
        }
    }
  };
  auto tile_origin = [&](int T, int &b, int &ty0, int &tx0) {  // wave-uniform (scalar ALU)
    const int per = tiles_x * tiles_y;
    b = T / per;
    const int r = T - b * per;
    ty0 = (r / tiles_x) * G::TH;
    tx0 = (r % tiles_x) * G::TW;
  };
  // Tile-independent part of every staged pixel's source offset, computed ONCE: the per-item
  // address arithmetic is then one add + four compares per pixel (32-bit; 64-bit multiplies per
  // load cost as much VALU time as the MFMAs they feed).  Tile origins are even, so for the
  // zero-stuffed (stride-2 transposed) input  (ty0 + r - 1) >> 1 == ty0/2 + ((r - 1) >> 1).
  int rel_r[NST], rel_c[NST], off0[NST], off1[NST], offp[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = tid + 256 * i;
    rel_r[i] = e / G::LW - 1;
    rel_c[i] = e % G::LW - 1;
    const int ys = a.ups ? (rel_r[i] >> 1) : rel_r[i];
    const int xs = a.ups ? (rel_c[i] >> 1) : rel_c[i];
    off0[i] = (ys * a.Ws + xs) * a.C0;
    off1[i] = (ys * a.Ws + xs) * a.C1;
    offp[i] = ys * a.Ws + xs;
    if (e >= NPIX) rel_r[i] = -(1 << 28);  // never in range
    if (a.ups && !((rel_r[i] & 1) && (rel_c[i] & 1))) rel_r[i] = -(1 << 28);  // stuffed zero
  }
  // single source, no zero-stuffing, no canvas plane: the common case, kept free of per-element
  // flag tests (they are wave-uniform, but inside the unrolled loops each becomes a branch)
  const bool simple = !a.ups && a.C1 == 0 && a.plane == nullptr;
  int org_b = 0, org_ty0 = 0, org_tx0 = 0;     // origin of the tile load_item staged last (the NEXT tile inside the loop)
  int cur_b = 0, cur_ty0 = 0, cur_tx0 = 0;     // origin of the tile being computed: two scalar divisions per tile, not four
  auto load_item = [&](int T, int ch) {  // global -> registers (input tile + halo of one chunk)
    int b, ty0, tx0;
    tile_origin(T, b, ty0, tx0);
    org_b = b, org_ty0 = ty0, org_tx0 = tx0;
    const int ylo = -ty0, yhi = a.H - ty0, xlo = -tx0, xhi = a.W - tx0;
    if (simple) {
      const int base = ((b * a.Hs + ty0) * a.Ws + tx0) * a.C0 + ch * CK;  // scalar
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const bool ok = (rel_r[i] >= ylo) & (rel_r[i] < yhi) & (rel_c[i] >= xlo) & (rel_c[i] < xhi);
        // out-of-image pixels read past the descriptor's range: the buffer unit returns zeros,
        // so the SAME padding costs no branch
        if (BF16 && a.in_bf16) {  // uniform: the tensor is stored as bf16 (8 bytes per channel quad)
          const int off = ok ? (base + off0[i]) * 2 : kOOB;
#pragma unroll
          for (int cg = 0; cg < G::NCG; ++cg) straw[i][cg] = __builtin_bit_cast(u32x2s, __builtin_amdgcn_raw_buffer_load_b64(rs0, off, 8 * cg, 0));
          continue;
        }
        const int off = ok ? (base + off0[i]) * 4 : kOOB;
#pragma unroll
        for (int cg = 0; cg < G::NCG; ++cg)
          st[i][cg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 16 * cg, 0));
      }
      return;
    }
    const int sy0 = a.ups ? (ty0 >> 1) : ty0, sx0 = a.ups ? (tx0 >> 1) : tx0;
    const int pbase = (b * a.Hs + sy0) * a.Ws + sx0;  // scalar: source pixel index of the origin
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const bool ok = (rel_r[i] >= ylo) & (rel_r[i] < yhi) & (rel_c[i] >= xlo) & (rel_c[i] < xhi);
#pragma unroll
      for (int cg = 0; cg < G::NCG; ++cg) {
        const int chan = ch * CK + cg * 4;  // uniform
        if (chan < a.C0) {
          if (BF16 && a.in_bf16) {
            const int off = ok ? (pbase * a.C0 + off0[i] + chan) * 2 : kOOB;
            straw[i][cg] = __builtin_bit_cast(u32x2s, __builtin_amdgcn_raw_buffer_load_b64(rs0, off, 0, 0));
            continue;
          }
          const int off = ok ? (pbase * a.C0 + off0[i] + chan) * 4 : kOOB;
          st[i][cg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
          if (a.plane && chan == (a.plane_chan & ~3)) {  // uniform: the canvas lives in its own plane
            const float pv = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rsp, ok ? (pbase + offp[i]) * 4 : kOOB, 0, 0));
            const int slot = a.plane_chan & 3;
            st[i][cg].x = slot == 0 ? pv : st[i][cg].x;
            st[i][cg].y = slot == 1 ? pv : st[i][cg].y;
            st[i][cg].z = slot == 2 ? pv : st[i][cg].z;
            st[i][cg].w = slot == 3 ? pv : st[i][cg].w;
          }
        } else if (BF16 && a.in_bf16) {
          const int off = ok ? (pbase * a.C1 + off1[i] + chan - a.C0) * 2 : kOOB;
          straw[i][cg] = __builtin_bit_cast(u32x2s, __builtin_amdgcn_raw_buffer_load_b64(rs1, off, 0, 0));
        } else {
          const int off = ok ? (pbase * a.C1 + off1[i] + chan - a.C0) * 4 : kOOB;
          st[i][cg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
        }
      }
    }
  };
  auto store_item = [&](int buf) {  // registers -> LDS buffer, transposed to [ksub][cg]
    float *tb = tile + buf * G::LDS_FLOATS;
    if (BF16 && a.in_bf16) {  // (uniform) the staged values still are the loaded bf16 bits: widened only here, after the MFMA
                              // loop they flew across — widening at the load put the wait for them in front of that loop
#pragma unroll
      for (int i = 0; i < NST; ++i)
#pragma unroll
        for (int cg = 0; cg < G::NCG; ++cg) st[i][cg] = bf16x4_to_f32(straw[i][cg]);
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int e = tid + 256 * i;
      if (e < NPIX) {
        float *rec = tb + e * G::PIX;
        if constexpr (G::NCG == 1) {
          *reinterpret_cast<f32x4 *>(rec) = st[i][0];
        } else if constexpr (G::NCG == 2) {
          *reinterpret_cast<f32x4 *>(rec) = f32x4{st[i][0].x, st[i][1].x, st[i][0].y, st[i][1].y};
          *reinterpret_cast<f32x4 *>(rec + 4) = f32x4{st[i][0].z, st[i][1].z, st[i][0].w, st[i][1].w};
        } else {
          *reinterpret_cast<f32x4 *>(rec) = f32x4{st[i][0].x, st[i][1].x, st[i][2].x, st[i][3].x};
          *reinterpret_cast<f32x4 *>(rec + 4) = f32x4{st[i][0].y, st[i][1].y, st[i][2].y, st[i][3].y};
          *reinterpret_cast<f32x4 *>(rec + 8) = f32x4{st[i][0].z, st[i][1].z, st[i][2].z, st[i][3].z};
          *reinterpret_cast<f32x4 *>(rec + 12) = f32x4{st[i][0].w, st[i][1].w, st[i][2].w, st[i][3].w};
        }
      }
    }
  };
  auto compute = [&](int buf) {  // MFMA main loop: 9 taps, one wide A read per (tap, group)
    const float *tb = tile + buf * G::LDS_FLOATS;
    if constexpr (BF16) {
      constexpr int TPF = 4 / G::NCG;  // taps per bf16 quad (4 k-steps of this lane's ksub)
      auto quad = [&](int f, int g) {  // the A operand's quad f of pixel group g (beyond the chunk's k-steps: zeros)
        if (f >= NMF) return bf16x4{0, 0, 0, 0};
        const int gx = g % GX, gy = g / GX;
        float v4[4];
#pragma unroll
        for (int tp = 0; tp < TPF; ++tp) {
          const int tap = (f * TPF + tp < 9) ? f * TPF + tp : 8;  // beyond the 9 taps: any staged pixel (zero weights)
          const int ky = tap / 3, kx = tap % 3;
          const avec av = *reinterpret_cast<const avec *>(&tb[a_base + ((2 * gy + ky) * G::LW + 8 * gx + kx) * G::PIX]);
#pragma unroll
          for (int cg = 0; cg < G::NCG; ++cg) v4[tp * G::NCG + cg] = av[cg];
        }
        return pack_bf16(v4[0], v4[1], v4[2], v4[3]);
      };
      // two quads per v_mfma_f32_16x16x32_bf16 (the CDNA4 form; CDNA3's K = 16 form took one)
#pragma unroll
      for (int f = 0; f < NMF; f += 2) {
        bf16x4 a0[G::PM], a1[G::PM];
#pragma unroll
        for (int g = 0; g < G::PM; ++g) {
          a0[g] = quad(f, g);
          a1[g] = quad(f + 1, g);
        }
#pragma unroll
        for (int g = 0; g < G::PM; ++g)
#pragma unroll
          for (int n = 0; n < NC; ++n) {
            const bf16x4 b1 = f + 1 < NMF ? bpk[f + 1 < NMF ? f + 1 : 0][n] : bf16x4{0, 0, 0, 0};
            if constexpr (SWAP)
              acc[g][n] = mfma32(bpk[f][n], b1, a0[g], a1[g], acc[g][n]);
            else
              acc[g][n] = mfma32(a0[g], a1[g], bpk[f][n], b1, acc[g][n]);
          }
      }
      return;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      avec av[G::PM];
#pragma unroll
      for (int g = 0; g < G::PM; ++g) {
        const int gx = g % GX, gy = g / GX;
        if constexpr (UPS) {  // g = parity class (py, px) = (gy, gx): only its own taps touch stuffed data
          if (((ky ^ gy) & 1) | ((kx ^ gx) & 1)) continue;
          av[g] = *reinterpret_cast<const avec *>(&tb[a_base + ((gy + ky) * G::LW + gx + kx) * G::PIX]);
        } else {
          av[g] = *reinterpret_cast<const avec *>(
              &tb[a_base + ((2 * gy + ky) * G::LW + 8 * gx + kx) * G::PIX]);
        }
      }
#pragma unroll
      for (int cg = 0; cg < G::NCG; ++cg) {
        const int s = tap * G::NCG + cg;
#pragma unroll
        for (int g = 0; g < G::PM; ++g) {
          if constexpr (UPS) {
            if (((ky ^ (g / GX)) & 1) | ((kx ^ (g % GX)) & 1)) continue;
          }
          const float aval = av[g][cg];
#pragma unroll
          for (int n = 0; n < NC; ++n)
            if constexpr (SWAP)
              acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[s][n], aval, acc[g][n], 0, 0, 0);
            else
              acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, breg[s][n], acc[g][n], 0, 0, 0);
        }
      }
    }
  };
  auto store1 = [&](float v, int elem_off_or_oob_bytes4) {  // one output value; the offset comes in float32 bytes (or kOOB)
    if (BF16 && a.out_bf16) {
      const bf16x4 pk4 = pack_bf16(v, 0.f, 0.f, 0.f);
      __builtin_amdgcn_raw_buffer_store_b16(pk4.x, rsy, elem_off_or_oob_bytes4 == kOOB ? kOOB : elem_off_or_oob_bytes4 >> 1, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsy, elem_off_or_oob_bytes4, 0, 0);
    }
  };
  const int opool = a.pool;  // 1 or 2
  const float lo = a.relu ? 0.f : -__builtin_inff();  // ReLU as one v_max, no flag test per value
  auto epilogue = [&](int) {  // scale/shift (bias + BN), ReLU, 2x2 max-pool, store
    const int b = cur_b, ty0 = cur_ty0, tx0 = cur_tx0;
    if constexpr (!SWAP) {  // lane = channel co_lane; registers r = window element (dy,dx) of pixel group qo
      const int wrow0 = ty0 + wm * G::WR, lcol0 = tx0 + 2 * qo;
      if (opool == 2) {
#pragma unroll
        for (int n = 0; n < NC; ++n) {
          const int co = 16 * (wn * NC + n) + co_lane;
          const bool co_ok = co < a.Cout;
          const int obase = ((b * a.Ho + (wrow0 >> 1)) * a.Wo + (lcol0 >> 1)) * a.Cout + co;
#pragma unroll
          for (int g = 0; g < G::PM; ++g) {
            const int gx = g % GX, gy = g / GX;
            const int row0 = wrow0 + 2 * gy, col0 = lcol0 + 8 * gx;
            // max commutes with the (monotone or not) affine only after it: apply BN first
            const f32x4 v = acc[g][n] * sc4[n] + sh4[n];
            const float o = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), lo);
            const bool ok = co_ok & ((row0 >> 1) < a.Ho) & ((col0 >> 1) < a.Wo);
            const int boff = ok ? (obase + (gy * a.Wo + 4 * gx) * a.Cout) * 4 : kOOB;
            store1(o, boff);
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < NC; ++n) {
          const int co = 16 * (wn * NC + n) + co_lane;
          const bool co_ok = co < a.Cout;
          const int obase = ((b * a.Ho + wrow0) * a.Wo + lcol0) * a.Cout + co;
#pragma unroll
          for (int g = 0; g < G::PM; ++g) {
            const int gx = g % GX, gy = g / GX;
            const int row0 = wrow0 + 2 * gy, col0 = lcol0 + 8 * gx;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[g][n][r] * sc4[n].x + sh4[n].x, lo);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = co_ok & (row0 + (r >> 1) < a.Ho) & (col0 + (r & 1) < a.Wo);
              const int boff = ok ? (obase + ((2 * gy + (r >> 1)) * a.Wo + 8 * gx + (r & 1)) * a.Cout) * 4 : kOOB;
              store1(v[r], boff);
            }
          }
        }
      }
      return;
    }
    // SWAP layers never pool and always have Cout % 4 == 0 (the dispatch's condition for this form): one 16- or 8-byte
    // store per lane and pixel group, no window maximum, no division by the pool size (a run-time divisor cost two integer
    // divisions per pixel group here: a third of the epilogue's instructions)
    const int lrow = ty0 + wm * G::WR + (UPS ? 2 * dy : dy);  // conv row / col of this lane's pixel in group 0
    const int lcol = tx0 + (UPS ? 2 * (2 * q + dx) : 2 * q + dx);
    constexpr int GDY = UPS ? 1 : 2, GDX = UPS ? 1 : 8;  // a group's row / column offset from group 0 (UPS: its parity class)
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int co0 = 16 * (wn * NC + n) + 4 * ksub;
      const int obase = ((b * a.Ho + lrow) * a.Wo + lcol) * a.Cout + co0;
      const bool co_ok = co0 < a.Cout;
#pragma unroll
      for (int g = 0; g < G::PM; ++g) {
        const int gx = g % GX, gy = g / GX;
        f32x4 v = acc[g][n] * sc4[n] + sh4[n];
        const bool okp = (lrow + GDY * gy < a.H) & (lcol + GDX * gx < a.W);
        if constexpr (MOM) {
          if (g == 0 && !mhave) {  // pivot: this wave's first output of the channel (pixel lane 0 of the 16)
#pragma unroll
            for (int r = 0; r < 4; ++r) mpv[n][r] = __shfl(v[r], lane & 48, 64);
          }
          // whole-vector form (packed float32 adds / FMAs): the pixel's weight 1 / 0 multiplies d instead of selecting it
          // per element — 8 packed operations per pixel group where the selects took 26 scalar ones
          const float wgt = okp ? 1.f : 0.f;
          const f32x4 d = (v - mpv[n]) * f32x4{wgt, wgt, wgt, wgt};
          ms1[n] += d;
          ms2[n] += d * d;
          if (n == 0) mcnt += wgt;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo);
        const int off = obase + (GDY * gy * a.Wo + GDX * gx) * a.Cout;
        if (BF16 && a.out_bf16) {  // four channels = 8 bytes (RNE, as the operand rounding)
          const int boff = (okp & co_ok) ? off * 2 : kOOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, pack_bf16(v[0], v[1], v[2], v[3])), rsy, boff, 0, 0);
        } else {
          const int boff = (okp & co_ok) ? off * 4 : kOOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, boff, 0, 0);
        }
      }
    }
    mhave = true;
  };

  int T = blockIdx.x, ch = 0, buf = 0;
  if (T >= ntiles) return;
  load_b(0);
  load_item(T, 0);  // (ahead of pack_b: the bf16 form's packing waits for the weights — the first tile's loads fly meanwhile)
  pack_b();
  cur_b = org_b, cur_ty0 = org_ty0, cur_tx0 = org_tx0;
  store_item(0);
  zero_acc();
  __syncthreads();
  while (true) {
    int nT = T, nch = ch + 1;
    if (nch == nchunks) {
      nch = 0;
      nT = T + gridDim.x;
    }
    const bool has_next = nT < ntiles;
    if (has_next) load_item(nT, nch);  // in flight across the MFMA loop
    if (BDB && has_next && nchunks > 1) load_b_next(nch);
    compute(buf);
    if (ch == nchunks - 1) {
      epilogue(T);
      zero_acc();
    }
    if (!has_next) break;
    if (nchunks > 1) {
      if constexpr (BDB) swap_b();
      else load_b(nch);
      pack_b();
    }
    store_item(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    T = nT;
    ch = nch;
    cur_b = org_b, cur_ty0 = org_ty0, cur_tx0 = org_tx0;
  }
  if constexpr (MOM) {  // the wave's record: sums over its 16 pixel lanes, one float4 {n, S1, S2, pivot} per channel
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      mcnt += __shfl_xor(mcnt, o, 64);
#pragma unroll
      for (int n = 0; n < NC; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ms1[n][r] += __shfl_xor(ms1[n][r], o, 64);
          ms2[n][r] += __shfl_xor(ms2[n][r], o, 64);
        }
    }
    if ((lane & 15) == 0) {
      f32x4 *rec = reinterpret_cast<f32x4 *>(a.mom_part) + (size_t)(blockIdx.x * G::WM + wm) * a.CoutP;
#pragma unroll
      for (int n = 0; n < NC; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) rec[16 * (wn * NC + n) + 4 * ksub + r] = f32x4{mcnt, ms1[n][r], ms2[n][r], mpv[n][r]};
    }
  }
}

template <int CK, int NC, int WN, int GX, int GY, bool SWAP, bool BF16 = false, bool MOM = false, bool UPS = false>
int launch_s(const Args &a, int B, hipStream_t st) {
  using G = Geo<CK, NC, WN, GX, GY>;
  auto kern = conv3x3_mfma<CK, NC, WN, GX, GY, SWAP, BF16, MOM, UPS>;
  constexpr size_t lds = 2 * G::LDS_FLOATS * sizeof(float);
  static int wgs_per_cu = 0;  // idempotent lazy init
  if (!wgs_per_cu) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, lds) != hipSuccess || n < 1) n = 1;
    wgs_per_cu = n > 4 ? 4 : n;
  }
  const int tiles_x = ceil_div(a.W, G::TW), tiles_y = ceil_div(a.H, G::TH);
  const int ntiles = tiles_x * tiles_y * B;
  const int cap = wgs_per_cu * num_cus();
  const int grid = ntiles < cap ? ntiles : cap;
  if (a.nparts_out) *a.nparts_out = grid * G::WM;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, tiles_x, tiles_y, ntiles);
  return launch_status("ra_conv3x3_f32");
}

namespace {  // the dispatch chain differs between the parts of this file (RA_K1_PART): internal linkage, one per part
inline bool ups_subpixel() {  // RA_CONV_UPS_SUBPIXEL=0: the zero-stuffed form of rounds 1-5 (A/B aid)
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("RA_CONV_UPS_SUBPIXEL");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  return on == 1;
}
template <int CK, int NC, int WN, int GX, int GY>
int launch(const Args &a, int B, hipStream_t st) {
  // channel-vector stores pay off when there is no pooling and the channel count allows float4
  const bool swap = a.pool == 1 && (a.Cout & 3) == 0;
  if (a.mom_part && !swap)
    return ra::fail(RA_E_SHAPE, "ra_conv3x3_moments_f32: needs pool 1 and Cout %% 4 == 0 (Cout %d, pool %d)", a.Cout, a.pool);
#if RA_K1_PART == 1  // bf16 operands
  if (swap) {
    if (a.mom_part) return launch_s<CK, NC, WN, GX, GY, true, true, true>(a, B, st);
    return launch_s<CK, NC, WN, GX, GY, true, true>(a, B, st);
  }
  return launch_s<CK, NC, WN, GX, GY, false, true>(a, B, st);
#elif RA_K1_PART == 2  // float32 + batch moments
  return launch_s<CK, NC, WN, GX, GY, true, false, true>(a, B, st);
#else
  if (swap) {
    if constexpr (GX == 2 && GY == 2) {
      if (a.ups && ups_subpixel()) return launch_s<CK, NC, WN, GX, GY, true, false, false, true>(a, B, st);
    }
    return launch_s<CK, NC, WN, GX, GY, true>(a, B, st);
  }
  return launch_s<CK, NC, WN, GX, GY, false>(a, B, st);
#endif
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
  static int force = -1;  // RA_CONV_GEO=<gx><gy> (e.g. 41) forces a geometry: tuning aid only
  if (force < 0) {
    const char *e = getenv("RA_CONV_GEO");
    force = e ? atoi(e) : 0;
  }
#if RA_K1_PART == 0
  // a stride-2 transposed conv without pooling runs its sub-pixel form (UPS), which lives on the (2, 2) geometry
  // ... where that geometry still yields enough workgroups: with few (a lone batch of 8 CVPPP patches: 16-72) the zero-stuffed
  // form on one-group tiles has 4 x the workgroups at the same chain length per wave and wins (RA_CONV_UPS_MIN_WGS, profiles/r06_k1s_sweep.txt)
  static int ups_min = -1;
  if (ups_min < 0) {
    const char *e = getenv("RA_CONV_UPS_MIN_WGS");
    ups_min = e ? atoi(e) : 96;
  }
  if (!force && a.ups && a.pool == 1 && (a.Cout & 3) == 0 && !a.mom_part && ups_subpixel() && wgs(2, 2) >= ups_min)
    return launch<CK, NC, WN, 2, 2>(a, B, st);
#endif
  if (force == 42) return launch<CK, NC, WN, 4, 2>(a, B, st);
  if (force == 41) return launch<CK, NC, WN, 4, 1>(a, B, st);
  if (force == 22) return launch<CK, NC, WN, 2, 2>(a, B, st);
  if (force == 21) return launch<CK, NC, WN, 2, 1>(a, B, st);
  if (!narrow) {
    if (wgs(4, 2) >= want) return launch<CK, NC, WN, 4, 2>(a, B, st);
    if (wgs(4, 1) >= want) return launch<CK, NC, WN, 4, 1>(a, B, st);
  }
  if (wgs(2, 2) >= want) return launch<CK, NC, WN, 2, 2>(a, B, st);
#if RA_K1_PART == 0
  // round 6: ONE pixel group per wave (8-column tiles) where even the smallest two-group geometry leaves most of the chip
  // without a workgroup — the patch-sized layers with many channels (KITTI's attention DCNN: 128 -> 64 at 12 x 12 is 288
  // dependent k-steps per pixel group; two groups per wave and 96 workgroups made it 16.5 us whatever the batch).  RA_CONV_TINY_WGS:
  // the workgroup count of the (2, 1) geometry below which the (1, 1) form is taken (0 = never).
  static int tiny = -1;
  if (tiny < 0) {
    const char *e = getenv("RA_CONV_TINY_WGS");
    tiny = e ? atoi(e) : 200;
  }
  if (force == 11 || (!force && wgs(2, 1) < tiny)) return launch<CK, NC, WN, 1, 1>(a, B, st);
#endif
  return launch<CK, NC, WN, 2, 1>(a, B, st);
}

// Cout groups per wave (NC) x waves along cout (WN).  Small problems (few tiles) split the cout
// groups over more waves so that each wave's serial MFMA chain is shorter.
template <int CK>
int dispatch_cout(const Args &a, int B, hipStream_t st) {
  const long tiles_big = (long)ceil_div(a.W, 16) * ceil_div(a.H, 8) * B;  // 8x16 tiles, WN = 1
  const bool small = tiles_big < 256;
  switch (a.CoutP) {
    case 16: return dispatch_geo<CK, 1, 1>(a, B, st);
    case 32: return small ? dispatch_geo<CK, 1, 2>(a, B, st) : dispatch_geo<CK, 2, 1>(a, B, st);
    case 64: return small ? dispatch_geo<CK, 1, 4>(a, B, st) : dispatch_geo<CK, 2, 2>(a, B, st);
    case 128: return dispatch_geo<CK, 2, 4>(a, B, st);
    default: return fail(RA_E_SHAPE, "ra_conv3x3_f32: CoutP %d unsupported", a.CoutP);
  }
}

}  // namespace
inline int chunk_of(int Cin) { return (Cin % 16 == 0) ? 16 : (Cin % 8 == 0) ? 8 : 4; }

// this part's dispatcher (the chain of template choices above); parts 1 and 2 export theirs to part 0's entry
#if RA_K1_PART == 1
#define RA_K1_DISPATCH k1_dispatch_bf16
#elif RA_K1_PART == 2
#define RA_K1_DISPATCH k1_dispatch_moments
#else
#define RA_K1_DISPATCH k1_dispatch_plain
int k1_dispatch_bf16(const Args &a, int B, hipStream_t st);
int k1_dispatch_moments(const Args &a, int B, hipStream_t st);
#endif
int RA_K1_DISPATCH(const Args &a, int B, hipStream_t st) {
  switch (chunk_of(a.C0 + a.C1)) {
    case 16: return dispatch_cout<16>(a, B, st);
    case 8: return dispatch_cout<8>(a, B, st);
    default: return dispatch_cout<4>(a, B, st);
  }
}

}  // namespace conv
}  // namespace ra

#if RA_K1_PART == 0

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

namespace ra {
namespace conv8 {
bool takes(int Cin, int Cout, int in_bf16, int B, int H, int W);
bool takes_f32(int Cin, int Cout, int B, int H, int W);
bool takes16_f32(int Cin, int Cout, int B, int H, int W);
int run16_f32(const void *x, int Cin, int B, int H, int W, const float *wp, const float *scale, const float *shift, int relu, void *y,
              float *part, int *nparts, int cus, hipStream_t st);
int run(const void *x, int Cin, int in_bf16, int B, int H, int W, const float *wp, const float *scale, const float *shift, int relu,
        void *y, int out_bf16, float *part, int *nparts, int cus, hipStream_t st, int Cout);
}  // namespace conv8
}  // namespace ra

static int conv3x3_entry(const float *src0, int C0, const float *src1, int C1, int B, int Hs, int Ws, int upsample,
                         const float *wpacked, const float *scale, const float *shift, int Cout, int relu, int pool,
                         const float *plane, int plane_chan, float *y, void *stream, int bf16, float *mom_part = nullptr,
                         int *nparts = nullptr, int store_flags = 0) {
  if (!src0 || !wpacked || !scale || !shift || !y || B <= 0 || Hs <= 0 || Ws <= 0 || C0 <= 0 ||
      C1 < 0 || (C1 > 0 && !src1))
    return ra::fail(RA_E_INVALID, "ra_conv3x3_f32: bad argument");
  if (C0 % 4 || C1 % 4) return ra::fail(RA_E_SHAPE, "ra_conv3x3_f32: C0=%d C1=%d must be %% 4", C0, C1);
  ra::conv::Args a;
  // patch-sized launches (the attention CNN / decoder on 48 x 48 and below: <= 32 K conv pixels) are links of the decode
  // loop's serial tail, a few us each, and run beside other batches' controller CNNs in the pipeline
  a.prio = ((size_t)B * Hs * Ws * (upsample ? 4 : 1) <= 32768) ? ra::tail_prio(2) : 0;
  a.bf16 = bf16;
  a.in_bf16 = (bf16 && (store_flags & 1)) ? 1 : 0;
  a.out_bf16 = (bf16 && (store_flags & 2)) ? 1 : 0;
  if (store_flags && (!bf16 || plane)) return ra::fail(RA_E_INVALID, "ra_conv3x3_bf16_f32: bf16 storage needs the bf16-operand kernels, no canvas plane");
  a.mom_part = mom_part;
  a.nparts_out = nparts;
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
  {
    const size_t n0 = (size_t)B * Hs * Ws * C0 * (a.in_bf16 ? 2 : 4), n1 = (size_t)B * Hs * Ws * C1 * (a.in_bf16 ? 2 : 4),
                 ny = (size_t)B * a.Ho * a.Wo * Cout * (a.out_bf16 ? 2 : 4);
    if (n0 >= (1ull << 31) || n1 >= (1ull << 31) || ny >= (1ull << 31))
      return ra::fail(RA_E_SHAPE, "ra_conv3x3_f32: a tensor exceeds 2 GiB (32-bit buffer offsets)");
    a.bytes0 = (int)n0;
    a.bytes1 = (int)n1;
    a.bytes_y = (int)ny;
    a.plane = plane;
    a.plane_chan = plane_chan;
    a.bytes_p = (int)((size_t)B * Hs * Ws * 4);
    if (plane && (plane_chan < 0 || plane_chan >= C0))
      return ra::fail(RA_E_INVALID, "ra_conv3x3_f32: plane channel %d of %d", plane_chan, C0);
  }
  // a chunk may not straddle the src0/src1 boundary at finer than 4 channels (always true) but
  // the chunk index arithmetic needs C0 % 4 == 0 only: chunks are resolved per channel group.
  hipStream_t st = ra::as_stream(stream);
  // bf16 mode, eight output channels at full resolution: the bf16-LDS kernel (ra_conv8.hip)
  if (a.bf16 && !C1 && !a.ups && pool == 1 && !plane && ra::conv8::takes(C0, Cout, a.in_bf16, B, a.H, a.W))
    return ra::conv8::run(src0, C0, a.in_bf16, B, a.H, a.W, wpacked, scale, shift, relu, y, a.out_bf16, mom_part, nparts,
                          ra::conv::num_cus(), st, Cout);
  if (a.bf16) return ra::conv::k1_dispatch_bf16(a, B, st);
  // float32, eight output channels at full resolution (training: forward with moments, data gradients): the 16-block MFMA form
  if (!C1 && !a.ups && pool == 1 && !plane && ra::conv8::takes_f32(C0, Cout, B, a.H, a.W))
    return ra::conv8::run(src0, C0, -1, B, a.H, a.W, wpacked, scale, shift, relu, y, 0, mom_part, nparts, ra::conv::num_cus(), st, 8);
  if (!C1 && !a.ups && pool == 1 && !plane && ra::conv8::takes16_f32(C0, Cout, B, a.H, a.W))  // 16 channels at half resolution
    return ra::conv8::run16_f32(src0, C0, B, a.H, a.W, wpacked, scale, shift, relu, y, mom_part, nparts, ra::conv::num_cus(), st);
  if (a.mom_part) return ra::conv::k1_dispatch_moments(a, B, st);
  return ra::conv::k1_dispatch_plain(a, B, st);
}

extern "C" int ra_conv3x3_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs,
                              int Ws, int upsample, const float *wpacked, const float *scale,
                              const float *shift, int Cout, int relu, int pool, const float *plane,
                              int plane_chan, float *y, void *stream) {
  return conv3x3_entry(src0, C0, src1, C1, B, Hs, Ws, upsample, wpacked, scale, shift, Cout, relu, pool, plane,
                       plane_chan, y, stream, 0);
}

extern "C" size_t ra_conv3x3_moments_part_floats(int Cout) {
  const int cp = ra_conv_cout_padded(Cout);
  return (size_t)4 * ra::conv::num_cus() * 4 * cp * 4;  // <= 4 workgroups per CU x 4 wave rows x CoutP records of 4 floats
}

extern "C" int ra_conv3x3_moments_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs, int Ws,
                                      int upsample, const float *wpacked, const float *scale, const float *shift, int Cout,
                                      int relu, int bf16_operands, float *y, float *part, size_t part_floats, int *nparts,
                                      void *stream) {
  if (!part || !nparts || part_floats < ra_conv3x3_moments_part_floats(Cout))
    return ra::fail(RA_E_WORKSPACE, "ra_conv3x3_moments_f32: partial buffer");
  return conv3x3_entry(src0, C0, src1, C1, B, Hs, Ws, upsample, wpacked, scale, shift, Cout, relu, 1, nullptr, -1, y, stream,
                       bf16_operands ? 1 : 0, part, nparts);
}

extern "C" int ra_conv3x3_bf16ops_f32(const float *src0, int C0, const float *src1, int C1, int B, int Hs,
                                      int Ws, int upsample, const float *wpacked, const float *scale,
                                      const float *shift, int Cout, int relu, int pool, const float *plane,
                                      int plane_chan, float *y, void *stream) {
  return conv3x3_entry(src0, C0, src1, C1, B, Hs, Ws, upsample, wpacked, scale, shift, Cout, relu, pool, plane,
                       plane_chan, y, stream, 1);
}

// The bf16 mode's layers between themselves (model_opt['compute_dtype'] = 'bf16', DESIGN.md "Mixed precision"): bf16
// operands as ra_conv3x3_bf16ops_f32 / ra_conv3x3_moments_f32(bf16_operands = 1), and the tensors STORED as bf16 —
// store_flags bit 0: src0 / src1 hold bf16 values (2 bytes each), bit 1: y is written as bf16 (round to nearest even; the
// batch moments of `part` are taken from the float32 accumulators, before the rounding).  part = NULL: no moments (pool as
// given); part != NULL: as ra_conv3x3_moments_f32 (pool 1, Cout % 4 == 0).
extern "C" int ra_conv3x3_bf16_f32(const void *src0, int C0, const void *src1, int C1, int B, int Hs, int Ws, int upsample,
                                   const float *wpacked, const float *scale, const float *shift, int Cout, int relu, int pool,
                                   void *y, float *part, size_t part_floats, int *nparts, int store_flags, void *stream) {
  if (part && (!nparts || part_floats < ra_conv3x3_moments_part_floats(Cout)))
    return ra::fail(RA_E_WORKSPACE, "ra_conv3x3_bf16_f32: partial buffer");
  return conv3x3_entry(static_cast<const float *>(src0), C0, static_cast<const float *>(src1), C1, B, Hs, Ws, upsample, wpacked, scale,
                       shift, Cout, relu, part ? 1 : pool, nullptr, -1, static_cast<float *>(y), stream, 1, part, part ? nparts : nullptr,
                       store_flags);
}
#endif  // RA_K1_PART == 0
