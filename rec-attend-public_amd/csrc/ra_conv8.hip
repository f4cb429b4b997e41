// The training step's bf16 mode, 3x3 convolutions with EIGHT output channels at full resolution (the controller CNN's first
// two layers and the data gradients that come back through them: cnn_filter_size[0:2] of full_model.py:240-262 at 512x512,
// 8 to 128 images per launch).  K1 (ra_conv.hip) is instruction-bound there: it stages float32 through LDS, rounds every
// operand to bf16 as it leaves LDS and spends ~1400 instructions per wave-tile around 32 MFMAs.  This kernel keeps the tile
// in LDS AS bf16, pixel-major ([row][col][Cin] — the tensor's own layout), so that a lane's whole B operand of one
// v_mfma_f32_16x16x32_bf16 is ONE ds_read_b128 with no conversion:
//   A (16 x 32) = the weights: row = output channel (8 real, 8 zero), a lane's 8 k-slots = 8 (tap, ci) pairs, held in
//       registers for the whole launch;
//   B (32 x 16) = 16 consecutive pixels of a row: lane (px, kb) reads the 16 bytes at pixel (px + tap offset): Cin = 8 ->
//       the 8 channels of tap 4m + kb; Cin = 16 -> half a pixel of tap 2m + kb/2; Cin = 4 -> two taps (two ds_read_b64);
//   D lane (px, q) = channels 4q..4q+3 of pixel px; q >= 2 is padding, and the epilogue fills those lanes with the wave's
//       second row (v_permlane32_swap): every lane stores 8 bytes (bf16) or 16 bytes (float32) per pair of groups.
// 9 taps x Cin = 36 / 72 / 144 k-values -> 2 / 3 / 5 MFMAs per 16 pixels (slots past tap 8 carry zero weights and read a
// valid pixel).  Same contraction as K1's bf16 kernels (operands rounded to bf16 RNE, float32 accumulation), a different
// order of the float32 sums.  The batch-moment records are K1's ({n, S1, S2, pivot} per wave and channel).
#include <type_traits>

#include "ra_common.h"

namespace ra {
namespace conv8 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int kOOB = 0x7fffffff;
constexpr int TW = 64, TH = 8, LW = TW + 2, LH = TH + 2, NPIX = LW * LH;  // a tile and its halo

struct Args {
  const void *x;
  const float *wp, *scale, *shift;
  void *y;
  float *part;
  int B, H, W, relu, out_bf16, bytes_x, bytes_y, ntx, nty, ntiles;
};

__device__ inline unsigned pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// lanes 0..31 keep `a`, lanes 32..63 receive lanes 0..31 of `b` (v_permlane32_swap).  By-value floats: __builtin_bit_cast
// applied to a vector ELEMENT lvalue read element 0 for every index (hipcc 7.2)
__device__ inline float lower_from(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, r[0]);
}

template <int CIN>
struct Map {
  static constexpr int NMF = CIN == 4 ? 2 : CIN == 8 ? 3 : 5;
  // slot j (0..7) of k-block kb (0..3) in MFMA m: the tap and the input channel it stands for
  __device__ static int tap(int m, int kb, int j) { return CIN == 4 ? 8 * m + 2 * kb + (j >> 2) : CIN == 8 ? 4 * m + kb : 2 * m + (kb >> 1); }
  __device__ static int ci(int kb, int j) { return CIN == 4 ? (j & 3) : CIN == 8 ? j : 8 * (kb & 1) + j; }
};

template <int CIN, bool INB, bool MOM, int COUT = 8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CIN == 16 || (CIN == 8 && !INB)) ? 3 : 4, 4))) void conv8_kernel(const Args a) {
  constexpr int PB = CIN * 2;                    // LDS bytes per pixel
  constexpr int ESZ = INB ? 2 : 4;
  constexpr int IPP = CIN * ESZ / 16;            // 16-byte HBM items per pixel
  constexpr int NITEM = NPIX * IPP;
  constexpr int NIT = (NITEM + 255) / 256;
  constexpr int NMF = Map<CIN>::NMF;
  static_assert(IPP >= 1 && IPP <= 2, "conv8: 16 or 32 bytes per input pixel");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][NPIX * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, kb = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, a.bytes_x, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes_y, 0x00020000);

  // the weights: K1's packed order [tap][ci][CoutP = 16] (one chunk: Cin <= 16)
  bf16x8 wreg[NMF];
#pragma unroll
  for (int m = 0; m < NMF; ++m) {
    f32x8 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = Map<CIN>::tap(m, kb, j), ci = Map<CIN>::ci(kb, j);
      w[j] = (px < COUT && tap < 9) ? a.wp[(tap * CIN + ci) * 16 + px] : 0.f;
    }
    wreg[m] = __builtin_convertvector(w, bf16x8);
  }
  // byte offsets of this lane's B operands inside the tile, relative to its pixel in the wave's first row
  int toff[NMF], toffb[CIN == 4 ? NMF : 1];
#pragma unroll
  for (int m = 0; m < NMF; ++m) {
    const int tap = Map<CIN>::tap(m, kb, 0);
    const int t = tap < 9 ? tap : 0;
    toff[m] = ((t / 3) * LW + t % 3) * PB + (CIN == 16 ? (kb & 1) * 16 : 0);
    if constexpr (CIN == 4) {
      const int t2 = tap + 1 < 9 ? tap + 1 : 0;
      toffb[m] = ((t2 / 3) * LW + t2 % 3) * PB;
    }
  }
  const int lanebase = (2 * wave * LW + px) * PB;
  // COUT = 8: after the epilogue's lane swap a lane owns channels 4 (kb & 1) ..; COUT = 16: channels 4 kb .. (no padding rows)
  const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(a.scale + 4 * (COUT == 16 ? kb : (kb & 1)));
  const f32x4 sh4 = *reinterpret_cast<const f32x4 *>(a.shift + 4 * (COUT == 16 ? kb : (kb & 1)));
  const float lo = a.relu ? 0.f : -__builtin_inff();

  u32x4 raw[NIT];
  int org_b = 0, org_ty = 0, org_tx = 0;
  auto load_tile = [&](int T) {
    const int tx = T % a.ntx, r = T / a.ntx;
    const int ty = r % a.nty, b = r / a.nty;
    org_b = b, org_ty = ty * TH, org_tx = tx * TW;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      const int pix = i / IPP, sub = i % IPP;
      const int row = pix / LW, col = pix - row * LW;
      const int gy = org_ty - 1 + row, gx = org_tx - 1 + col;
      const bool ok = (i < NITEM) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
      const int off = ok ? (((b * a.H + gy) * a.W + gx) * CIN * ESZ + sub * 16) : kOOB;
      raw[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 256 * it;
      if (i < NITEM) {
        const int pix = i / IPP, sub = i % IPP;
        if constexpr (INB) {
          *reinterpret_cast<u32x4 *>(&lds[buf][pix * PB + sub * 16]) = raw[it];
        } else {  // four float32 channels -> four bf16 (RNE, as K1's operand rounding)
          const f32x4 v = __builtin_bit_cast(f32x4, raw[it]);
          *reinterpret_cast<u32x2 *>(&lds[buf][pix * PB + sub * 8]) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
        }
      }
    }
  };

  f32x4 ms1 = {0.f, 0.f, 0.f, 0.f}, ms2 = ms1, mpv = ms1;
  float mcnt = 0.f;
  bool mhave = false;

  int T = blockIdx.x, buf = 0;
  if (T >= a.ntiles) return;
  load_tile(T);
  store_tile(0);
  int cur_b = org_b, cur_ty = org_ty, cur_tx = org_tx;
  __syncthreads();
  while (true) {
    const int nT = T + gridDim.x;
    const bool has_next = nT < a.ntiles;
    if (has_next) load_tile(nT);  // in flight across the MFMAs and the epilogue
    const unsigned char *base = &lds[buf][lanebase];
    f32x4 acc[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int goff = ((g >> 2) * LW + 16 * (g & 3)) * PB;
#pragma unroll
      for (int m = 0; m < NMF; ++m) {
        bf16x8 bv;
        if constexpr (CIN == 4) {
          const u32x2 p0 = *reinterpret_cast<const u32x2 *>(base + goff + toff[m]);
          const u32x2 p1 = *reinterpret_cast<const u32x2 *>(base + goff + toffb[m]);
          bv = __builtin_bit_cast(bf16x8, u32x4{p0.x, p0.y, p1.x, p1.y});
        } else {
          bv = *reinterpret_cast<const bf16x8 *>(base + goff + toff[m]);
        }
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[m], bv, acc[g], 0, 0, 0);
      }
      // keep the scheduler from hoisting every group's LDS reads above the first MFMA (40 x 4 registers at Cin = 16)
      if (NMF * (g + 1) % 10 == 0 || (NMF < 5 && (g & 3) == 3)) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (COUT == 16) {
      // sixteen output channels: every D row is a channel — lane (px, kb) owns channels 4 kb .. + 3 of pixel px in each of
      // the wave's 8 groups, one 8-byte (bf16) or 16-byte store per group
      const bool interior = (cur_ty + TH <= a.H) & (cur_tx + TW <= a.W);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int row = cur_ty + 2 * wave + (g >> 2), col = cur_tx + 16 * (g & 3) + px;
        f32x4 v = acc[g] * sc4 + sh4;
        const bool okp = interior || ((row < a.H) & (col < a.W));
        if constexpr (MOM) {
          if (g == 0 && !mhave) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mpv[r] = __shfl(v[r], lane & 48, 64);
          }
          const float wgt = okp ? 1.f : 0.f;
          const f32x4 d = (v - mpv) * f32x4{wgt, wgt, wgt, wgt};
          ms1 += d;
          ms2 += d * d;
          mcnt += wgt;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo);
        const int off = ((cur_b * a.H + row) * a.W + col) * 16 + 4 * kb;
        if (a.out_bf16)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])}, rsy, okp ? off * 2 : kOOB, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, okp ? off * 4 : kOOB, 0, 0);
      }
      mhave = true;
    } else
    // epilogue.  Half of a D tile is padding (output channels 8..15, lanes kb >= 2): v_permlane32_swap moves the wave's
    // SECOND row's channels into those lanes, so that one pass with all 64 lanes active finishes two groups: lane
    // (px, kb) owns channels 4 (kb & 1) .. + 3 of pixel (row + (kb >> 1), col + px).
    {
      const int hi = kb >> 1;
      const int row0 = cur_ty + 2 * wave + hi, col0 = cur_tx + px;
      const int obase = ((cur_b * a.H + row0) * a.W + col0) * 8 + 4 * (kb & 1);
      auto finish = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
        for (int gc = 0; gc < 4; ++gc) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] = lower_from(acc[gc][r], acc[gc + 4][r]);
          v = v * sc4 + sh4;
          const bool okp = INTERIOR || ((row0 < a.H) & (col0 + 16 * gc < a.W));
          if constexpr (MOM) {
            if (gc == 0 && !mhave) {  // pivot: the wave's first output of the channel (row 0, pixel 0: lanes 0 and 16)
#pragma unroll
              for (int r = 0; r < 4; ++r) mpv[r] = __shfl(v[r], lane & 16, 64);
            }
            if constexpr (INTERIOR) {
              const f32x4 d = v - mpv;
              ms1 += d;
              ms2 += d * d;
              mcnt += 1.f;
            } else {
              const float wgt = okp ? 1.f : 0.f;
              const f32x4 d = (v - mpv) * f32x4{wgt, wgt, wgt, wgt};
              ms1 += d;
              ms2 += d * d;
              mcnt += wgt;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo);
          const int off = obase + 16 * gc * 8;
          if (a.out_bf16)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])}, rsy, okp ? off * 2 : kOOB, 0, 0);
          else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, okp ? off * 4 : kOOB, 0, 0);
        }
      };
      if ((cur_ty + TH <= a.H) & (cur_tx + TW <= a.W)) finish(std::true_type{});
      else finish(std::false_type{});
      mhave = true;
    }
    if (!has_next) break;
    store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
    T = nT;
    cur_b = org_b, cur_ty = org_ty, cur_tx = org_tx;
  }
  if constexpr (MOM && COUT == 16) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {  // over the 16 pixels; lane bits 4-5 = the channel quad
      mcnt += __shfl_xor(mcnt, o, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ms1[r] += __shfl_xor(ms1[r], o, 64);
        ms2[r] += __shfl_xor(ms2[r], o, 64);
      }
    }
    if (px == 0) {
      f32x4 *rec = reinterpret_cast<f32x4 *>(a.part) + (size_t)(blockIdx.x * 4 + wave) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) rec[4 * kb + r] = f32x4{mcnt, ms1[r], ms2[r], mpv[r]};
    }
    return;
  }
  if constexpr (MOM) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {  // over the 16 pixels and the two rows (lane bits 0-3 and 5); bit 4 = the channel quad
      if (o == 16) continue;
      mcnt += __shfl_xor(mcnt, o, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ms1[r] += __shfl_xor(ms1[r], o, 64);
        ms2[r] += __shfl_xor(ms2[r], o, 64);
      }
    }
    if (px == 0 && kb < 2) {
      f32x4 *rec = reinterpret_cast<f32x4 *>(a.part) + (size_t)(blockIdx.x * 4 + wave) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) rec[4 * kb + r] = f32x4{mcnt, ms1[r], ms2[r], mpv[r]};
    }
  }
}

// ---- float32 mode: the same tile, exact float32 on v_mfma_f32_4x4x1_16B_f32 ----
// K1's 16x16x4 tile pads 8 output channels to 16 (half of every MFMA is zero rows; measured 50 % of the float32 matrix
// rate at that).  The 16-block form has no padding to give away: a block is 4 output channels x 4 pixels x ONE k-value,
// 16 blocks = 2 channel quads x 8 pixel quads = 32 pixels x 8 channels per instruction (8 cycles, the same 64 FLOP/clk).
//   A: lane (blk, j) = the weight of channel 4 (blk & 1) + j at this (tap, ci) — 9 Cin registers for the whole launch;
//   B: lane (blk, j) = pixel 4 (blk >> 1) + j: one ds_read_b128 = 4 input channels = 4 k-steps; the wave computes TWO output
//      rows from each read (input row i is tap row i of output row 0 and tap row i - 1 of row 1): 12 x Cin/4 reads per
//      18 Cin MFMAs, so LDS stays under the matrix pipe;
//   D: lane = 4 consecutive channels of its pixel: one 16-byte store, every lane active.
// Measured (tools/mfma_rate_probe.hip): the 16-block form issues every 14 cycles from one wave and every ~11.4 per SIMD
// with four, not the 8 of its 2 passes: 110 TFLOP/s chip.  Where the 39 us at 8 -> 8, 512 x 512 x 8 go: built with
// -DRA_C8_SKIP (one MFMA per LDS read kept, 1/8 of them) the same launch takes 25 us = 5.4 TB/s, the HBM roofline of the
// shape; the 4 600 MFMAs per SIMD are 22-27 us at 11.4-14 cycles; three waves per SIMD (72 weight registers per lane)
// overlap the two only partly.  Also measured and not kept: the same tile on the vector unit (a lane = a pixel column, 4
// v_pk_fma_f32 per (tap, ci, pixel) with the filter pair in scalar registers — no padding, no MFMA issue gap on paper).
// Left to hipcc the filter's s_loads either hoist out of the tile loop and spill (576 values) or serialise against
// lgkmcnt(0) every 16 packed FMAs: 43.5 us; issued one step ahead from inline asm (wait lgkmcnt(0), issue the next 32
// values into the other register set, 32 FMAs): 39.0 us — the same as this kernel, and equal to it on cache-resident
// sizes too (51 against 48 TFLOP/s at 2 images), so the simpler matrix form stays.  A third inner loop — v_mfma_f32_16x16x4
// with the 16 rows = 2 output rows x 8 channels over the four input rows they see (24 k-steps per 16 pixels x 2 rows, 9/12
// useful, one ds_read_b128 per 4 MFMAs, 24 filter registers instead of 72) — also took 39.5 us, and 42 us with three LDS
// tiles and the loads two tiles ahead (counted vmcnt, raw s_barrier).  Three loops with 22-31 us of arithmetic each land on
// the same 39: the launch behaves like the SUM of most of its memory pipeline (25 us, measured with the arithmetic
// compiled out) and its arithmetic, not their maximum, whatever the prefetch distance; not understood, not pursued.
template <int CIN, bool MOM>
__global__ __launch_bounds__(256) void conv8f_kernel(const Args a) {
  constexpr int PB = CIN * 4;    // LDS bytes per pixel
  constexpr int IPP = CIN / 4;   // 16-byte items per pixel
  constexpr int NITEM = NPIX * IPP;
  constexpr int NIT = (NITEM + 255) / 256;
  // the tile goes from HBM straight into LDS (buffer_load_dwordx4 ... lds: wave-uniform base + 16 bytes per lane, which IS
  // the tile's layout in item order; lanes past the image or the tile read the descriptor's out-of-range zero): no staging
  // registers (24 at Cin = 8, where the 72 weight registers already decide the occupancy) and no ds_write pass
  constexpr int LDSB = (NITEM + 63) / 64 * 1024;  // whole 64-lane pieces
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][LDSB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 3, coq = (lane >> 2) & 1, pix = 4 * (lane >> 3) + j;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, a.bytes_x, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes_y, 0x00020000);
  float w[9][CIN];  // K1's packed order [tap][ci][CoutP = 16]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) w[t][ci] = a.wp[(t * CIN + ci) * 16 + 4 * coq + j];
  const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(a.scale + 4 * coq);
  const f32x4 sh4 = *reinterpret_cast<const f32x4 *>(a.shift + 4 * coq);
  const float lo = a.relu ? 0.f : -__builtin_inff();
  const int lanebase = (2 * wave * LW + pix) * PB;

  int org_b = 0, org_ty = 0, org_tx = 0;
  auto load_tile = [&](int T, int buf) {
    const int tx = T % a.ntx, r = T / a.ntx;
    const int ty = r % a.nty, b = r / a.nty;
    org_b = b, org_ty = ty * TH, org_tx = tx * TW;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (256 * it + 64 * wave >= NITEM) continue;  // wave-uniform
      const int i = tid + 256 * it;
      const int p = i / IPP, sub = i % IPP;
      const int row = p / LW, col = p - row * LW;
      const int gy = org_ty - 1 + row, gx = org_tx - 1 + col;
      const bool ok = (i < NITEM) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
      const int off = ok ? (((b * a.H + gy) * a.W + gx) * CIN * 4 + sub * 16) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void *)(&lds[buf][(256 * it + 64 * wave) * 16]), 16,
                                               off, 0, 0, 0);
    }
  };

  f32x4 ms1 = {0.f, 0.f, 0.f, 0.f}, ms2 = ms1, mpv = ms1;
  float mcnt = 0.f;
  bool mhave = false;

  int T = blockIdx.x, buf = 0;
  if (T >= a.ntiles) return;
  load_tile(T, 0);
  int cur_b = org_b, cur_ty = org_ty, cur_tx = org_tx;
  __syncthreads();
  while (true) {
    const int nT = T + gridDim.x;
    const bool has_next = nT < a.ntiles;
    if (has_next) load_tile(nT, buf ^ 1);  // in flight across the MFMAs; the barrier below waits for it
    const unsigned char *base = &lds[buf][lanebase];
    const bool interior = (cur_ty + TH <= a.H) & (cur_tx + TW <= a.W);
    f32x4 accs[2][2];  // [pixel group][output row]
#pragma unroll
    for (int gx = 0; gx < 2; ++gx) {
      // two accumulators per output row (even / odd input channel): the first and last input rows feed ONE output row, and
      // a chain of dependent 2-pass MFMAs would wait on its own result
      f32x4 acc2[2][2];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc2[o >> 1][o & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
      // 12 steps (input row, tap column), the next step's pixels read while this step's MFMAs issue; the fence after every
      // step keeps the scheduler from hoisting all 12 x Cin/4 LDS reads above the first MFMA (+56 registers)
      f32x4 xv[2][IPP];
      auto fetch = [&](int st, f32x4 *dst) {
        const int ir = st / 3, kx = st % 3;
#pragma unroll
        for (int h = 0; h < IPP; ++h) dst[h] = *reinterpret_cast<const f32x4 *>(base + (ir * LW + 32 * gx + kx) * PB + 16 * h);
      };
      fetch(0, xv[0]);
#pragma unroll
      for (int st = 0; st < 12; ++st) {
        const int ir = st / 3, kx = st % 3;
        if (st + 1 < 12) fetch(st + 1, xv[(st + 1) & 1]);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
          for (int orow = 0; orow < 2; ++orow) {
            const int ky = ir - orow;
            if (ky < 0 || ky > 2) continue;
#ifdef RA_C8_SKIP  // measuring aid: the tile loop without its MFMAs (one per LDS read keeps the reads alive)
            if (ci % 4 == 0 && orow == (ir == 3))
#endif
            acc2[orow][ci & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[ky * 3 + kx][ci], xv[st & 1][ci / 4][ci % 4], acc2[orow][ci & 1], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      accs[gx][0] = acc2[0][0] + acc2[0][1];
      accs[gx][1] = acc2[1][0] + acc2[1][1];
    }
    // the barrier BEFORE the stores: its vmcnt(0) (the next tile's HBM -> LDS loads) would otherwise also wait for this
    // tile's stores to be acknowledged — gfx9 counts loads and stores in one counter — once per tile, with nothing to hide it
    if (has_next) __syncthreads();
#pragma unroll
    for (int gx = 0; gx < 2; ++gx) {
      const f32x4 *acc = accs[gx];
      const int col = cur_tx + 32 * gx + pix;
#pragma unroll
      for (int orow = 0; orow < 2; ++orow) {
        const int row = cur_ty + 2 * wave + orow;
        f32x4 v = acc[orow] * sc4 + sh4;
        const bool okp = interior || ((row < a.H) & (col < a.W));
        if constexpr (MOM) {
          if (gx == 0 && orow == 0 && !mhave) {  // pivot: the wave's first output of the channel (pixel 0: lanes 0 and 4)
#pragma unroll
            for (int r = 0; r < 4; ++r) mpv[r] = __shfl(v[r], lane & 4, 64);
          }
          const float wgt = okp ? 1.f : 0.f;
          const f32x4 d = (v - mpv) * f32x4{wgt, wgt, wgt, wgt};
          ms1 += d;
          ms2 += d * d;
          mcnt += wgt;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo);
        const int off = ((cur_b * a.H + row) * a.W + col) * 8 + 4 * coq;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, okp ? off * 4 : kOOB, 0, 0);
      }
      mhave = true;
    }
    if (!has_next) break;
    buf ^= 1;
    T = nT;
    cur_b = org_b, cur_ty = org_ty, cur_tx = org_tx;
  }
  if constexpr (MOM) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {  // over the 32 pixels (lane bits 0-1 and 3-5); bit 2 = the channel quad
      if (o == 4) continue;
      mcnt += __shfl_xor(mcnt, o, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ms1[r] += __shfl_xor(ms1[r], o, 64);
        ms2[r] += __shfl_xor(ms2[r], o, 64);
      }
    }
    if ((lane & ~4) == 0) {
      f32x4 *rec = reinterpret_cast<f32x4 *>(a.part) + (size_t)(blockIdx.x * 4 + wave) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) rec[4 * coq + r] = f32x4{mcnt, ms1[r], ms2[r], mpv[r]};
    }
  }
}

// ---- float32 mode, SIXTEEN output channels (the controller CNN's half-resolution layers: 8 -> 16 and 16 -> 16 at 256 x 256) ----
// K1 runs these at half of the float32 matrix rate (33.8 us for 16 -> 16 at 256 x 256 x 8: 2.4 GFLOP).  Same tile scheme as
// the 8-channel kernels (HBM -> LDS directly, double-buffered, persistent), v_mfma_f32_16x16x4_f32 with A = the filter
// (16 channels x 4 k-values, held in registers: 4 * ceil(9 Cin / 16) per lane) and B = 16 consecutive pixels.  The k-slots
// of a group of four MFMAs are laid out so that ONE ds_read_b128 per lane feeds all four: slot (m, kq) = channel
// 4 (kq % QP) + m of tap TPG g + kq / QP (QP = Cin / 4 quads per pixel, TPG = 16 / Cin taps per group; Cin = 8 pads its
// ninth tap's group with zero weights: 20 MFMAs per 16 pixels for 18).  D lane (px, q) = channels 4q .. 4q + 3 of pixel px.
constexpr int TW2 = 32, LW2 = TW2 + 2, NPIX2 = LW2 * LH;
template <int CIN, bool MOM>
__global__ __launch_bounds__(256) void conv16_kernel(const Args a) {
  constexpr int PB = CIN * 4;
  constexpr int IPP = CIN / 4;
  constexpr int NITEM = NPIX2 * IPP;
  constexpr int NIT = (NITEM + 255) / 256;
  constexpr int LDSB = (NITEM + 63) / 64 * 1024;
  constexpr int QP = CIN / 4, TPG = 4 / QP, NG = (9 + TPG - 1) / TPG;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][LDSB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, px = lane & 15, q = lane >> 4;
  const int csel = q % QP, tsel = q / QP;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.x), 0, a.bytes_x, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes_y, 0x00020000);
  float wreg[NG][4];  // A operand: row = output channel (lane & 15), k-slot (m, q)
  int toff[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int tap = TPG * g + tsel, t = tap < 9 ? tap : 0;
    toff[g] = ((t / 3) * LW2 + t % 3) * PB + 16 * csel;
#pragma unroll
    for (int m = 0; m < 4; ++m) wreg[g][m] = tap < 9 ? a.wp[(tap * CIN + 4 * csel + m) * 16 + px] : 0.f;
  }
  const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(a.scale + 4 * q);
  const f32x4 sh4 = *reinterpret_cast<const f32x4 *>(a.shift + 4 * q);
  const float lo = a.relu ? 0.f : -__builtin_inff();
  const int lanebase = (2 * wave * LW2 + px) * PB;

  int org_b = 0, org_ty = 0, org_tx = 0;
  auto load_tile = [&](int T, int buf) {
    const int tx = T % a.ntx, r = T / a.ntx;
    const int ty = r % a.nty, b = r / a.nty;
    org_b = b, org_ty = ty * TH, org_tx = tx * TW2;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (256 * it + 64 * wave >= NITEM) continue;  // wave-uniform
      const int i = tid + 256 * it;
      const int p = i / IPP, sub = i % IPP;
      const int row = p / LW2, col = p - row * LW2;
      const int gy = org_ty - 1 + row, gx = org_tx - 1 + col;
      const bool ok = (i < NITEM) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gx < (unsigned)a.W);
      const int off = ok ? (((b * a.H + gy) * a.W + gx) * CIN * 4 + sub * 16) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void *)(&lds[buf][(256 * it + 64 * wave) * 16]), 16,
                                               off, 0, 0, 0);
    }
  };

  f32x4 ms1 = {0.f, 0.f, 0.f, 0.f}, ms2 = ms1, mpv = ms1;
  float mcnt = 0.f;
  bool mhave = false;

  int T = blockIdx.x, buf = 0;
  if (T >= a.ntiles) return;
  load_tile(T, 0);
  int cur_b = org_b, cur_ty = org_ty, cur_tx = org_tx;
  __syncthreads();
  while (true) {
    const int nT = T + gridDim.x;
    const bool has_next = nT < a.ntiles;
    if (has_next) load_tile(nT, buf ^ 1);  // in flight across the MFMAs; the barrier below waits for it
    const unsigned char *base = &lds[buf][lanebase];
    f32x4 acc[4];  // [row of the wave's two][16-pixel group of the row's two]
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a 16x16x4 result is ready after 40 cycles, the same accumulator's next instruction may issue after 32: the four
    // accumulators take turns
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      f32x4 xv[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) xv[o] = *reinterpret_cast<const f32x4 *>(base + toff[g] + ((o >> 1) * LW2 + 16 * (o & 1)) * PB);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g][m], xv[o][m], acc[o], 0, 0, 0);
    }
    if (has_next) __syncthreads();  // ahead of the stores: its vmcnt(0) is for the next tile's loads
    {
      const bool interior = (cur_ty + TH <= a.H) & (cur_tx + TW2 <= a.W);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int row = cur_ty + 2 * wave + (o >> 1), col = cur_tx + 16 * (o & 1) + px;
        f32x4 v = acc[o] * sc4 + sh4;
        const bool okp = interior || ((row < a.H) & (col < a.W));
        if constexpr (MOM) {
          if (o == 0 && !mhave) {  // pivot: the wave's first output of the channel (pixel 0 of the lane's quarter)
#pragma unroll
            for (int r = 0; r < 4; ++r) mpv[r] = __shfl(v[r], lane & 48, 64);
          }
          const float wgt = okp ? 1.f : 0.f;
          const f32x4 d = (v - mpv) * f32x4{wgt, wgt, wgt, wgt};
          ms1 += d;
          ms2 += d * d;
          mcnt += wgt;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo);
        const int off = ((cur_b * a.H + row) * a.W + col) * 16 + 4 * q;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsy, okp ? off * 4 : kOOB, 0, 0);
      }
      mhave = true;
    }
    if (!has_next) break;
    buf ^= 1;
    T = nT;
    cur_b = org_b, cur_ty = org_ty, cur_tx = org_tx;
  }
  if constexpr (MOM) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {  // over the 16 pixels; lane bits 4-5 = the channel quad
      mcnt += __shfl_xor(mcnt, o, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ms1[r] += __shfl_xor(ms1[r], o, 64);
        ms2[r] += __shfl_xor(ms2[r], o, 64);
      }
    }
    if (px == 0) {
      f32x4 *rec = reinterpret_cast<f32x4 *>(a.part) + (size_t)(blockIdx.x * 4 + wave) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) rec[4 * q + r] = f32x4{mcnt, ms1[r], ms2[r], mpv[r]};
    }
  }
}

template <int CIN, bool INB, int COUT = 8>
static int launch(const Args &a, int grid, hipStream_t st) {
  if (a.part)
    hipLaunchKernelGGL((conv8_kernel<CIN, INB, true, COUT>), dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv8_kernel<CIN, INB, false, COUT>), dim3(grid), dim3(256), 0, st, a);
  return launch_status("ra_conv3x3_bf16_f32 (8- / 16-channel form)");
}

bool enabled() {
  static int on = -1;
  if (on < 0) {
    const char *e = getenv("RA_CONV8");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on != 0;
}

// 1 when the shape is this kernel's (the caller has already checked: bf16 operands, one source, no transposed stride, no
// pooling, no canvas plane)
bool takes(int Cin, int Cout, int in_bf16, int B, int H, int W) {
  if (!enabled() || (Cout != 8 && Cout != 16)) return false;
  if (!((Cin == 4 && !in_bf16 && Cout == 8) || Cin == 8 || (Cin == 16 && in_bf16))) return false;
  return (size_t)B * H * W >= (size_t)64 * 64 * 8;  // K1's tiles fill the chip better on small launches
}

bool takes_f32(int Cin, int Cout, int B, int H, int W) {
  if (!enabled() || Cout != 8 || (Cin != 4 && Cin != 8)) return false;
  return (size_t)B * H * W >= (size_t)64 * 64 * 8;
}

bool takes16_f32(int Cin, int Cout, int B, int H, int W) {
  if (!enabled() || Cout != 16 || (Cin != 8 && Cin != 16)) return false;
  return (size_t)B * H * W >= (size_t)64 * 64 * 8;
}

int run16_f32(const void *x, int Cin, int B, int H, int W, const float *wp, const float *scale, const float *shift, int relu, void *y,
              float *part, int *nparts, int cus, hipStream_t st) {
  Args a;
  a.x = x, a.wp = wp, a.scale = scale, a.shift = shift, a.y = y, a.part = part;
  a.B = B, a.H = H, a.W = W, a.relu = relu, a.out_bf16 = 0;
  a.bytes_x = (int)((size_t)B * H * W * Cin * 4);
  a.bytes_y = (int)((size_t)B * H * W * 16 * 4);
  a.ntx = ceil_div(W, TW2), a.nty = ceil_div(H, TH);
  const long long nt = (long long)B * a.ntx * a.nty;
  if (nt >= (1ll << 31)) return fail(RA_E_SHAPE, "ra_conv3x3_f32: tile count");
  a.ntiles = (int)nt;
  void (*kern)(const Args) = Cin == 8 ? (part ? conv16_kernel<8, true> : conv16_kernel<8, false>)
                                      : (part ? conv16_kernel<16, true> : conv16_kernel<16, false>);
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  const int cap = (per_cu < 4 ? per_cu : 4) * cus;  // the moment records' buffer holds 4 workgroups per CU x 4 waves
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  if (nparts) *nparts = grid * 4;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
  return launch_status("ra_conv3x3_f32 (16-channel form)");
}

int run(const void *x, int Cin, int in_bf16, int B, int H, int W, const float *wp, const float *scale, const float *shift, int relu,
        void *y, int out_bf16, float *part, int *nparts, int cus, hipStream_t st, int Cout) {
  Args a;
  a.x = x, a.wp = wp, a.scale = scale, a.shift = shift, a.y = y, a.part = part;
  a.B = B, a.H = H, a.W = W, a.relu = relu, a.out_bf16 = out_bf16;
  a.bytes_x = (int)((size_t)B * H * W * Cin * (in_bf16 > 0 ? 2 : 4));  // in_bf16 < 0: the float32 kernels
  a.bytes_y = (int)((size_t)B * H * W * Cout * (out_bf16 ? 2 : 4));
  a.ntx = ceil_div(W, TW), a.nty = ceil_div(H, TH);
  const long long nt = (long long)B * a.ntx * a.nty;
  if (nt >= (1ll << 31)) return fail(RA_E_SHAPE, "ra_conv3x3_bf16_f32: tile count");
  a.ntiles = (int)nt;
  // the moment records' buffer holds 4 workgroups per CU x 4 waves; the float32 kernel at Cin = 8 keeps 3 resident (LDS)
  const int cap = (in_bf16 < 0 && Cin == 8 ? 3 : 4) * cus;
  const int grid = a.ntiles < cap ? a.ntiles : cap;
  if (nparts) *nparts = grid * 4;
  if (in_bf16 < 0) {  // float32 mode
    void (*kern)(const Args) = Cin == 4 ? (a.part ? conv8f_kernel<4, true> : conv8f_kernel<4, false>)
                                        : (a.part ? conv8f_kernel<8, true> : conv8f_kernel<8, false>);
    int per_cu = 0;  // registers decide (72 weights per lane at Cin = 8): ask the runtime, the moment buffer's 4 at most
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    const int gridf = a.ntiles < (per_cu < 4 ? per_cu : 4) * cus ? a.ntiles : (per_cu < 4 ? per_cu : 4) * cus;
    if (nparts) *nparts = gridf * 4;
    hipLaunchKernelGGL(kern, dim3(gridf), dim3(256), 0, st, a);
    return launch_status("ra_conv3x3_f32 (8-channel form)");
  }
  if (Cout == 16) {
    if (Cin == 16) return launch<16, true, 16>(a, grid, st);
    return in_bf16 ? launch<8, true, 16>(a, grid, st) : launch<8, false, 16>(a, grid, st);
  }
  if (Cin == 4) return launch<4, false>(a, grid, st);
  if (Cin == 16) return launch<16, true>(a, grid, st);
  return in_bf16 ? launch<8, true>(a, grid, st) : launch<8, false>(a, grid, st);
}

}  // namespace conv8
}  // namespace ra
