// K1 (pair form) — TWO consecutive 3x3 SAME conv layers fused in one launch:
//   A: conv3x3 (+bias+BN+ReLU, no pool; optionally the zero-stuffed stride-2 transposed conv)
//   B: conv3x3 (+bias+BN+ReLU, max-pool 1|2)
// The A output of a tile (+1-pixel halo, recomputed) never leaves LDS, which removes the
// HBM write+read of the intermediate activation — the dominant cost of the first controller-CNN
// layers (L0 writes 8 MiB per 512x512 image that L1 reads straight back) — and halves the number
// of launches of the patch-sized attention CNN / DCNN.  Same f32 MFMA implicit-GEMM machinery as
// ra_conv.hip (v_mfma_f32_16x16x4_f32, 2x2-window row mapping, [pixel][ksub][cg] LDS records
// read with one wide ds_read per tap).  nnlib.py:229-253 (cnn) / :362-400 (dcnn), two layers.
#include <cstdlib>

#include "ra_common.h"

namespace ra {
namespace cpair {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

struct PArgs {
  const float *src;
  float *y;
  const float *wpA, *scA, *shA, *wpB, *scB, *shB;
  int C0, Hs, Ws, H, W, ups;
  int CoutAP, CoutB, CoutBP, poolB, Ho, Wo, reluA, reluB;
  const float *plane;  // optional [B,Hs,Ws] plane replacing input channel plane_chan (the canvas)
  int plane_chan;
  int bytes0, bytes_p;  // tensor sizes for the buffer descriptors (each < 2 GiB)
  const float *cache;   // conv_pair8 CACHED form: layer A's timestep-invariant partial sums
  int cache_rows, cache_gx, bytes_c;
  int bytes_y;  // conv_pair8: size of y for its buffer descriptor (< 2 GiB whenever the input is)
  int xcd_map;  // conv_pair8: 1 = each XCD (workgroup id mod 8) walks its own contiguous eighth of the tiles
  // conv_pair8, un-cached form only: a constant fill of another buffer rides on the launch (the decode loop's
  // once-per-forward prefill of y_out, 134 MB at cfg2: this kernel is MFMA-bound and leaves HBM idle, so the
  // stores, a few per thread and tile, cost nothing on the timeline — as a launch of its own they cost 28 us)
  float *rider_dst;
  int rider_quads;  // float4 groups to write (rider_dst 16-byte aligned, < 2 GiB)
  float rider_val;
  unsigned *tickets;  // conv_pair8: this launch's slot of tile-ticket pools (ra_common.h), nullptr = the static tile walk
};

template <int CINA, int CMID, int NCB, int GX, int GYB>
struct PGeo {
  static constexpr int NCA = (CMID + 15) / 16;
  static constexpr int TWB = 8 * GX, THB = 8 * GYB, PMB = GX * GYB;
  static constexpr int GXA = GX + 1, GRA = THB / 2 + 1, NGA = GXA * GRA;
  static constexpr int AW = 8 * GXA, AH = 2 * GRA;     // A-out tile (B tile + halo, padded)
  static constexpr int LWA = AW + 2, LHA = AH + 2;     // input tile of A
  static constexpr int NCGA = CINA / 4, CKA = CINA < 16 ? CINA : 16, NCHA = CINA / CKA, NCGAC = CKA / 4;
  static constexpr int NCGB = CMID / 4, CKB = CMID < 16 ? CMID : 16, NCHB = CMID / CKB, NCGBC = CKB / 4;
  static constexpr int KSA = 9 * NCGAC, KSB = 9 * NCGBC;
  static constexpr int IN_FLOATS = LHA * LWA * CINA;
  static constexpr int MID_FLOATS = AH * AW * CMID;
  static constexpr int RA = 4;                          // A groups per wave per round
  static constexpr int GPW = (NGA + 3) / 4;             // A groups per wave
  static constexpr int ROUNDS = (GPW + RA - 1) / RA;
};

template <int N>
struct vec_of {
  typedef float type __attribute__((ext_vector_type(N)));
};

template <int CINA, int CMID, int NCB, int GX, int GYB>
__global__ __launch_bounds__(256) void conv_pair_mfma(const PArgs a, int tiles_x, int tiles_y) {
  using G = PGeo<CINA, CMID, NCB, GX, GYB>;
  constexpr int NCA = G::NCA;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tin = lds;                  // [LHA][LWA][CINA]  records [ksub][cg]
  float *tmid = lds + G::IN_FLOATS;  // [AH][AW][CMID]    records [ksub][cg]
  typedef typename vec_of<G::NCGAC>::type avecA;
  typedef typename vec_of<G::NCGBC>::type avecB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: group bookkeeping on the SALU
  const int m = lane & 15, ksub = lane >> 4;
  const int q = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
  const int co_lane = lane & 15, qo = lane >> 4;
  const int per = tiles_x * tiles_y;
  const int b = blockIdx.x / per;
  const int trem = blockIdx.x - b * per;
  const int ty0 = (trem / tiles_x) * G::THB, tx0 = (trem % tiles_x) * G::TWB;

  // ---------------- phase 0: stage A's input tile (+2-pixel halo) ----------------
  {
    const int sy0 = a.ups ? (ty0 >> 1) : ty0, sx0 = a.ups ? (tx0 >> 1) : tx0;
    const float *base = a.src + ((size_t)(b * a.Hs + sy0) * a.Ws + sx0) * a.C0;
    for (int e = tid; e < G::LHA * G::LWA; e += 256) {
      const int rr = e / G::LWA - 2, cc = e % G::LWA - 2;  // relative to the B-tile origin
      const int Y = ty0 + rr, X = tx0 + cc;
      bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
      int ys = rr, xs = cc;
      if (a.ups) {
        ok = ok & (rr & 1) & (cc & 1);  // origins are even: parity of Y == parity of rr
        ys = rr >> 1;
        xs = cc >> 1;
      }
      f32x4 v[G::NCGA];
#pragma unroll
      for (int cg = 0; cg < G::NCGA; ++cg) {
        v[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) v[cg] = *reinterpret_cast<const f32x4 *>(base + (ys * a.Ws + xs) * a.C0 + 4 * cg);
      }
      if (a.plane && ok) {
        const float pv = a.plane[(size_t)(b * a.Hs + sy0 + ys) * a.Ws + sx0 + xs];
        const int pg = a.plane_chan >> 2, slot = a.plane_chan & 3;
#pragma unroll
        for (int cg = 0; cg < G::NCGA; ++cg) {  // selects, not runtime register indexing
          v[cg].x = (cg == pg && slot == 0) ? pv : v[cg].x;
          v[cg].y = (cg == pg && slot == 1) ? pv : v[cg].y;
          v[cg].z = (cg == pg && slot == 2) ? pv : v[cg].z;
          v[cg].w = (cg == pg && slot == 3) ? pv : v[cg].w;
        }
      }
      float *rec = tin + e * CINA;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (G::NCGA == 1) {
          rec[ks] = v[0][ks];
        } else if constexpr (G::NCGA == 2) {
          rec[ks * 2] = v[0][ks];
          rec[ks * 2 + 1] = v[1][ks];
        } else {
#pragma unroll
          for (int c4 = 0; c4 < G::NCGA / 4; ++c4)
            *reinterpret_cast<f32x4 *>(rec + ks * G::NCGA + 4 * c4) =
                f32x4{v[4 * c4][ks], v[4 * c4 + 1][ks], v[4 * c4 + 2][ks], v[4 * c4 + 3][ks]};
        }
      }
    }
  }
  __syncthreads();

  // ---------------- phase A: conv A over the B tile + halo, result -> LDS ----------------
  {
    float scA[NCA], shA[NCA];
#pragma unroll
    for (int n = 0; n < NCA; ++n) {
      scA[n] = a.scA[16 * n + co_lane];
      shA[n] = a.shA[16 * n + co_lane];
    }
    const int lane_in = (dy * G::LWA + 2 * q + dx) * CINA + ksub * G::NCGA;
    float bregA[G::KSA][NCA];
    auto load_bA = [&](int ch) {
      const float *wrow = a.wpA + ((size_t)ch * G::KSA * 4 + ksub) * a.CoutAP + co_lane;
#pragma unroll
      for (int s = 0; s < G::KSA; ++s)
#pragma unroll
        for (int n = 0; n < NCA; ++n) bregA[s][n] = wrow[(size_t)s * 4 * a.CoutAP + 16 * n];
    };
    if (G::NCHA == 1) load_bA(0);
    for (int rd = 0; rd < G::ROUNDS; ++rd) {
      f32x4 acc[G::RA][NCA];
      int gbase[G::RA], gr[G::RA], gc[G::RA];
#pragma unroll
      for (int j = 0; j < G::RA; ++j) {
        int gi = wave + 4 * (rd * G::RA + j);
        if (gi >= G::NGA) gi = G::NGA - 1;  // duplicate work, masked at the store
        gr[j] = gi / G::GXA;
        gc[j] = gi % G::GXA;
        gbase[j] = (2 * gr[j] * G::LWA + 8 * gc[j]) * CINA + lane_in;
#pragma unroll
        for (int n = 0; n < NCA; ++n) acc[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (int ch = 0; ch < G::NCHA; ++ch) {
        if (G::NCHA > 1) load_bA(ch);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          avecA av[G::RA];
#pragma unroll
          for (int j = 0; j < G::RA; ++j)
            av[j] = *reinterpret_cast<const avecA *>(
                &tin[gbase[j] + (ky * G::LWA + kx) * CINA + ch * G::NCGAC]);
#pragma unroll
          for (int cg = 0; cg < G::NCGAC; ++cg)
#pragma unroll
            for (int j = 0; j < G::RA; ++j)
#pragma unroll
              for (int n = 0; n < NCA; ++n)
                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][cg], bregA[tap * G::NCGAC + cg][n],
                                                                 acc[j][n], 0, 0, 0);
        }
      }
      // A epilogue -> tmid (zero outside the image: it is B's SAME padding).  Tiles whose whole
      // A region lies inside the image (uniform test) skip the per-value bounds arithmetic.
      const bool interior = (ty0 >= 1) & (ty0 - 1 + G::AH <= a.H) & (tx0 >= 1) & (tx0 + G::TWB + 1 <= a.W);
      const float loA = a.reluA ? 0.f : -__builtin_inff();
#pragma unroll
      for (int j = 0; j < G::RA; ++j) {
        const bool live = (wave + 4 * (rd * G::RA + j)) < G::NGA;
#pragma unroll
        for (int n = 0; n < NCA; ++n) {
          const int co = 16 * n + co_lane;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[j][n][r] * scA[n] + shA[n], loA);
          if (!interior) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int Y = ty0 - 1 + 2 * gr[j] + (r >> 1), X = tx0 - 1 + 8 * gc[j] + 2 * qo + (r & 1);
              o[r] = ((Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W)) ? o[r] : 0.f;
            }
          }
          if (live && co < CMID) {
            float *dst = tmid + ((2 * gr[j]) * G::AW + 8 * gc[j] + 2 * qo) * CMID + (co & 3) * G::NCGB + (co >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((r >> 1) * G::AW + (r & 1)) * CMID] = o[r];
          }
        }
      }
    }
  }
  __syncthreads();

  // ---------------- phase B: conv B out of tmid, epilogue -> global ----------------
  {
    f32x4 acc[G::PMB][NCB];
#pragma unroll
    for (int g = 0; g < G::PMB; ++g)
#pragma unroll
      for (int n = 0; n < NCB; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_base = ((wave * 2 * GYB + dy) * G::AW + 2 * q + dx) * CMID + ksub * G::NCGB;
    float bregB[G::KSB][NCB];
    for (int ch = 0; ch < G::NCHB; ++ch) {
      {
        const float *wrow = a.wpB + ((size_t)ch * G::KSB * 4 + ksub) * a.CoutBP + co_lane;
#pragma unroll
        for (int s = 0; s < G::KSB; ++s)
#pragma unroll
          for (int n = 0; n < NCB; ++n) bregB[s][n] = wrow[(size_t)s * 4 * a.CoutBP + 16 * n];
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        avecB av[G::PMB];
#pragma unroll
        for (int g = 0; g < G::PMB; ++g) {
          const int gx = g % GX, gy = g / GX;
          av[g] = *reinterpret_cast<const avecB *>(
              &tmid[a_base + ((2 * gy + ky) * G::AW + 8 * gx + kx) * CMID + ch * G::NCGBC]);
        }
#pragma unroll
        for (int cg = 0; cg < G::NCGBC; ++cg)
#pragma unroll
          for (int g = 0; g < G::PMB; ++g)
#pragma unroll
            for (int n = 0; n < NCB; ++n)
              acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][cg], bregB[tap * G::NCGBC + cg][n],
                                                               acc[g][n], 0, 0, 0);
      }
    }
    const int opool = a.poolB;
    const float loB = a.reluB ? 0.f : -__builtin_inff();
    const int wrow0 = ty0 + wave * 2 * GYB, lcol0 = tx0 + 2 * qo;
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
      const int co = 16 * n + co_lane;
      const float sc = a.scB[co], sh = a.shB[co];
      const bool co_ok = co < a.CoutB;
      const int obase = ((b * a.Ho + wrow0 / opool) * a.Wo + lcol0 / opool) * a.CoutB + co;
#pragma unroll
      for (int g = 0; g < G::PMB; ++g) {
        const int gx = g % GX, gy = g / GX;
        const int row0 = wrow0 + 2 * gy, col0 = lcol0 + 8 * gx;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[g][n][r] * sc + sh, loB);
        if (opool == 2) {
          const float o = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
          if (co_ok && (row0 >> 1) < a.Ho && (col0 >> 1) < a.Wo)
            a.y[obase + (gy * a.Wo + 4 * gx) * a.CoutB] = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co_ok && row0 + (r >> 1) < a.Ho && col0 + (r & 1) < a.Wo)
              a.y[obase + ((2 * gy + (r >> 1)) * a.Wo + 8 * gx + (r & 1)) * a.CoutB] = v[r];
        }
      }
    }
  }
}

template <int CINA, int CMID, int NCB, int GX, int GYB>
__global__ __launch_bounds__(256) void conv_pair_persist_mfma(const PArgs a, int tiles_x, int tiles_y, int ntiles) {
  using G = PGeo<CINA, CMID, NCB, GX, GYB>;
  constexpr int NCA = G::NCA;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tin = lds;                  // [LHA][LWA][CINA]  records [ksub][cg]
  float *tmid = lds + G::IN_FLOATS;  // [AH][AW][CMID]    records [ksub][cg]
  typedef typename vec_of<G::NCGAC>::type avecA;
  typedef typename vec_of<G::NCGBC>::type avecB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: group bookkeeping on the SALU
  const int m = lane & 15, ksub = lane >> 4;
  const int q = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
  const int co_lane = lane & 15, qo = lane >> 4;
  const int per = tiles_x * tiles_y;
  static_assert(G::NCHA == 1 && G::NCHB == 1, "persistent pair: single-chunk layers");
  // both layers' B operands and epilogue constants: once per workgroup
  float scA[NCA], shA[NCA];
#pragma unroll
  for (int n = 0; n < NCA; ++n) {
    scA[n] = a.scA[16 * n + co_lane];
    shA[n] = a.shA[16 * n + co_lane];
  }
  float bregA[G::KSA][NCA], bregB[G::KSB][NCB];
  {
    const float *wrow = a.wpA + (size_t)ksub * a.CoutAP + co_lane;
#pragma unroll
    for (int s = 0; s < G::KSA; ++s)
#pragma unroll
      for (int n = 0; n < NCA; ++n) bregA[s][n] = wrow[(size_t)s * 4 * a.CoutAP + 16 * n];
    const float *wrowb = a.wpB + (size_t)ksub * a.CoutBP + co_lane;
#pragma unroll
    for (int s = 0; s < G::KSB; ++s)
#pragma unroll
      for (int n = 0; n < NCB; ++n) bregB[s][n] = wrowb[(size_t)s * 4 * a.CoutBP + 16 * n];
  }
  // the input window (tile + 2-pixel halo) of a tile is fetched into registers one tile ahead
  constexpr int NPIXA = G::LHA * G::LWA, NIT = (NPIXA + 255) / 256;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.src), 0, a.bytes0, 0x00020000);
  int e_r[NIT], e_c[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + 256 * i;
    e_r[i] = e / G::LWA - 2;
    e_c[i] = e % G::LWA - 2;
    if (e >= NPIXA) e_r[i] = -(1 << 20);
  }
  f32x4 pre[NIT][G::NCGA];
  auto fetch = [&](int T) {
    const int fb = T / per, fr = T - fb * per;
    const int fy0 = (fr / tiles_x) * G::THB, fx0 = (fr % tiles_x) * G::TWB;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int Y = fy0 + e_r[i], X = fx0 + e_c[i];
      const bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
      const int off = ok ? (((fb * a.Hs + Y) * a.Ws + X) * a.C0) * 4 : 0x7fffffff;
#pragma unroll
      for (int cg = 0; cg < G::NCGA; ++cg)
        pre[i][cg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16 * cg, 0));
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
  const int b = tile / per;
  const int trem = tile - b * per;
  const int ty0 = (trem / tiles_x) * G::THB, tx0 = (trem % tiles_x) * G::TWB;

  // ---------------- phase 0: prefetched registers -> LDS (previous tile's phase B is complete
  // for every wave once all have arrived at the barrier below) ----------------
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + 256 * i;
    if (e < NPIXA) {
      float *rec = tin + e * CINA;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (G::NCGA == 1) {
          rec[ks] = pre[i][0][ks];
        } else if constexpr (G::NCGA == 2) {
          rec[ks * 2] = pre[i][0][ks];
          rec[ks * 2 + 1] = pre[i][1][ks];
        } else {
#pragma unroll
          for (int c4 = 0; c4 < G::NCGA / 4; ++c4)
            *reinterpret_cast<f32x4 *>(rec + ks * G::NCGA + 4 * c4) =
                f32x4{pre[i][4 * c4][ks], pre[i][4 * c4 + 1][ks], pre[i][4 * c4 + 2][ks], pre[i][4 * c4 + 3][ks]};
        }
      }
    }
  }
  __syncthreads();
  {
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch(next);  // in flight across both MFMA phases
  }

  // ---------------- phase A: conv A over the B tile + halo, result -> LDS ----------------
  {
    const int lane_in = (dy * G::LWA + 2 * q + dx) * CINA + ksub * G::NCGA;
    for (int rd = 0; rd < G::ROUNDS; ++rd) {
      f32x4 acc[G::RA][NCA];
      int gbase[G::RA], gr[G::RA], gc[G::RA];
#pragma unroll
      for (int j = 0; j < G::RA; ++j) {
        int gi = wave + 4 * (rd * G::RA + j);
        if (gi >= G::NGA) gi = G::NGA - 1;  // duplicate work, masked at the store
        gr[j] = gi / G::GXA;
        gc[j] = gi % G::GXA;
        gbase[j] = (2 * gr[j] * G::LWA + 8 * gc[j]) * CINA + lane_in;
#pragma unroll
        for (int n = 0; n < NCA; ++n) acc[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (int ch = 0; ch < G::NCHA; ++ch) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          avecA av[G::RA];
#pragma unroll
          for (int j = 0; j < G::RA; ++j)
            av[j] = *reinterpret_cast<const avecA *>(
                &tin[gbase[j] + (ky * G::LWA + kx) * CINA + ch * G::NCGAC]);
#pragma unroll
          for (int cg = 0; cg < G::NCGAC; ++cg)
#pragma unroll
            for (int j = 0; j < G::RA; ++j)
#pragma unroll
              for (int n = 0; n < NCA; ++n)
                acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][cg], bregA[tap * G::NCGAC + cg][n],
                                                                 acc[j][n], 0, 0, 0);
        }
      }
      // A epilogue -> tmid (zero outside the image: it is B's SAME padding).  Tiles whose whole
      // A region lies inside the image (uniform test) skip the per-value bounds arithmetic.
      const bool interior = (ty0 >= 1) & (ty0 - 1 + G::AH <= a.H) & (tx0 >= 1) & (tx0 + G::TWB + 1 <= a.W);
      const float loA = a.reluA ? 0.f : -__builtin_inff();
#pragma unroll
      for (int j = 0; j < G::RA; ++j) {
        const bool live = (wave + 4 * (rd * G::RA + j)) < G::NGA;
#pragma unroll
        for (int n = 0; n < NCA; ++n) {
          const int co = 16 * n + co_lane;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[j][n][r] * scA[n] + shA[n], loA);
          if (!interior) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int Y = ty0 - 1 + 2 * gr[j] + (r >> 1), X = tx0 - 1 + 8 * gc[j] + 2 * qo + (r & 1);
              o[r] = ((Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W)) ? o[r] : 0.f;
            }
          }
          if (live && co < CMID) {
            float *dst = tmid + ((2 * gr[j]) * G::AW + 8 * gc[j] + 2 * qo) * CMID + (co & 3) * G::NCGB + (co >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((r >> 1) * G::AW + (r & 1)) * CMID] = o[r];
          }
        }
      }
    }
  }
  __syncthreads();

  // ---------------- phase B: conv B out of tmid, epilogue -> global ----------------
  {
    f32x4 acc[G::PMB][NCB];
#pragma unroll
    for (int g = 0; g < G::PMB; ++g)
#pragma unroll
      for (int n = 0; n < NCB; ++n) acc[g][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_base = ((wave * 2 * GYB + dy) * G::AW + 2 * q + dx) * CMID + ksub * G::NCGB;
    for (int ch = 0; ch < G::NCHB; ++ch) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        avecB av[G::PMB];
#pragma unroll
        for (int g = 0; g < G::PMB; ++g) {
          const int gx = g % GX, gy = g / GX;
          av[g] = *reinterpret_cast<const avecB *>(
              &tmid[a_base + ((2 * gy + ky) * G::AW + 8 * gx + kx) * CMID + ch * G::NCGBC]);
        }
#pragma unroll
        for (int cg = 0; cg < G::NCGBC; ++cg)
#pragma unroll
          for (int g = 0; g < G::PMB; ++g)
#pragma unroll
            for (int n = 0; n < NCB; ++n)
              acc[g][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][cg], bregB[tap * G::NCGBC + cg][n],
                                                               acc[g][n], 0, 0, 0);
      }
    }
    const int opool = a.poolB;
    const float loB = a.reluB ? 0.f : -__builtin_inff();
    const int wrow0 = ty0 + wave * 2 * GYB, lcol0 = tx0 + 2 * qo;
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
      const int co = 16 * n + co_lane;
      const float sc = a.scB[co], sh = a.shB[co];
      const bool co_ok = co < a.CoutB;
      const int obase = ((b * a.Ho + wrow0 / opool) * a.Wo + lcol0 / opool) * a.CoutB + co;
#pragma unroll
      for (int g = 0; g < G::PMB; ++g) {
        const int gx = g % GX, gy = g / GX;
        const int row0 = wrow0 + 2 * gy, col0 = lcol0 + 8 * gx;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[g][n][r] * sc + sh, loB);
        if (opool == 2) {
          const float o = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
          if (co_ok && (row0 >> 1) < a.Ho && (col0 >> 1) < a.Wo)
            a.y[obase + (gy * a.Wo + 4 * gx) * a.CoutB] = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co_ok && row0 + (r >> 1) < a.Ho && col0 + (r & 1) < a.Wo)
              a.y[obase + ((2 * gy + (r >> 1)) * a.Wo + 8 * gx + (r & 1)) * a.CoutB] = v[r];
        }
      }
    }
  }
  }  // persistent tile loop
}

// ---------------------------------------------------------------------------------------------
// N-packed form for the full-resolution 8-channel pairs (controller CNN L0+L1: Cin -> 8 -> <=8,
// pool 2).  A 16x16x4 MFMA has 16 output columns; with 8 output channels half of them would
// multiply zeros.  Here the 16 columns are 2 horizontally adjacent pixels x 8 channels,
//   n = p*8 + co,  D[m][n] = sum_{ky, kx' in 0..3, ci} in[y+ky-1][x_even-1+kx'][ci] * W'[ky][kx'][ci][n],
//   W'[ky][kx'][ci][p*8+co] = W[ky][kx'-p][ci][co]  (0 outside 0 <= kx'-p <= 2),
// so one MFMA row is a pixel PAIR: 12 k-steps per 32 pixels instead of 2 x 9.  W' is built in
// registers from the standard packed filter (predicated loads), the interface does not change.
// Tile = 16 x 32 conv pixels (8 x 16 pooled).  Row mappings:
//   phase A (no pool): m -> (row m>>2, pair m&3), groups of 4 rows x 8 cols, 5 x 5 groups cover
//                      the 18 x 36 (even-aligned) region layer B needs; result -> LDS tile `tmid`
//   phase B (pool 2):  m -> (pair m>>1, row m&1), groups of 2 rows x 16 cols; the pool window of a
//                      pair is registers (2j, 2j+1) of lanes n and n^8 -> v_max + one DPP row_ror:8.
// Both LDS tiles use odd row strides (43 / 41 pixels) so the strided operand reads are 2-way
// instead of 4-way bank-conflicted (MI355X_MICROARCH.md, LDS: bank = dword address mod 32 / 64).
template <int CINA>
struct NGeo {
  static constexpr int TH = 16, TW = 32;
  static constexpr int NCGA = CINA / 4;
  static constexpr int AGX = 5, AGY = 5, NGA = AGX * AGY, GPW = (NGA + 3) / 4;
  static constexpr int AW = 41, AHS = TH + 2;  // tmid: row stride (pixels), rows stored
  static constexpr int LW = 43, LH = 4 * AGY + 2;  // tin: row stride, rows addressable
  // tin rows / cols actually loaded (20 x 38): every value that shares an MFMA row with a needed
  // output must be finite even where its weight is zero (pair 17 = columns 34|35 reads tin
  // columns 34..37; 0 * NaN would poison column 34)
  static constexpr int LHL = TH + 4, LWL = TW + 6;
  static constexpr int IN_FLOATS = LH * LW * CINA;
  static constexpr int MID_FLOATS = AHS * AW * 8;
};

// Round 6, measured and NOT kept (-DRA_PAIR8_CACHE_AHEAD=1 builds it): the NEXT tile's cached sums requested at the start of phase B,
// a whole phase ahead.  Their 28 registers stay live across phase B: at three workgroups per CU (168 VGPRs) the compiler spills 16
// (L0+L1 32.7 -> 38.6 us at cfg2, B = 8), at two per CU without spills 37.3 us (512 workgroups) / 44.4 (768): the occupancy is worth
// more than the 14 % wait it removes (profiles/r06_encoder_levers.txt).
#ifndef RA_PAIR8_CACHE_AHEAD
#define RA_PAIR8_CACHE_AHEAD 0
#endif
#ifndef RA_PAIR8_OCC
#define RA_PAIR8_OCC 3  // workgroups per CU: 3 x 38.7 KB LDS, <= 168 VGPRs (4 spills)
#endif
// tools/pair8_probe.hip builds this file with -DRA_PROBE8: every workgroup accumulates the shader-clock time its
// wave 0 spends between a few points of the tile loop and leaves the sums in ra_probe8_buf[workgroup][8]
// (-DRA_P8_NOBAR: the tile loop's barriers dropped — timing only, results wrong).
#ifdef RA_PROBE8
__device__ long long *ra_probe8_buf;
#define RA_P8_DECL long long p8_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, p8_t = (long long)__builtin_readcyclecounter(), p8_t0 = (long long)wall_clock64()
#define RA_P8_AT(k)                                                 \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const long long n_ = (long long)__builtin_readcyclecounter();   \
    p8_acc[k] += n_ - p8_t;                                         \
    p8_t = n_;                                                      \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
#define RA_P8_END                                                                   \
  do {                                                                              \
    if (threadIdx.x == 0 && ra_probe8_buf) {                                        \
      p8_acc[7] = (long long)wall_clock64() - p8_t0;                                \
      for (int k_ = 0; k_ < 8; ++k_) ra_probe8_buf[(size_t)blockIdx.x * 8 + k_] = p8_acc[k_]; \
    }                                                                               \
  } while (0)
#else
#define RA_P8_DECL
#define RA_P8_AT(k)
#define RA_P8_END
#endif
#ifdef RA_P8_NOBAR
#define RA_P8_SYNC() __builtin_amdgcn_s_waitcnt(0xc07f)  /* lgkmcnt(0) only */
#else
#define RA_P8_SYNC() __syncthreads()
#endif
// CACHED form (CINA == 4 only): of layer A's input only the canvas channel changes between
// timesteps (full_model.py:640-661,843-848), so the contribution of the image channels,
// S[pixel][co] = sum_{tap, ci != canvas} x * W (no bias, no BN), is computed ONCE per forward by
// first_cache_kernel into the accumulator layout of phase A; per timestep layer A is then
//   acc = S * scale(tt) + shift(tt)  (+)  3 MFMAs over the 3 x 4 canvas window
// instead of 12 MFMAs over the 4-channel window, and only the 4-byte canvas plane is staged.
// SPLIT form (round 5, CACHED only): layer B — 24 of the pair's 27 MFMAs per pixel group — runs on the BF16 matrix pipe at
// float32 accuracy.  A float32 value is exactly the sum of three bf16 pieces (8 + 8 + 8 mantissa bits), products of bf16
// numbers are exact in float32, and of the nine piece products of a * b six carry everything above 2^-24 of it (hh, hm, mh,
// hl, lh, mm): six v_mfma_f32_16x16x32_bf16 per K = 32 block do what eight v_mfma_f32_16x16x4_f32 do, in 41 ns of a SIMD
// instead of 108 (tools/mfma_split_probe.hip, profiles/r05_mfma_split_probe.txt: K = 576 dot products come out at 1.9e-7 of
// sum |a b| against 2.3e-7 for the float32 chain).  Phase A writes its output as three bf16 tiles [pixel][8 channels]
// (hi / mid / lo); a K = 32 block of phase B is one row ky of the 3 x 4 tap window x 8 channels, so a lane's whole A operand
// of a block and piece is ONE ds_read_b128 (lane kb = window column), and the filter is 3 x 3 x 4 registers per lane.
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
__device__ inline unsigned pk_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32: two floats -> two bf16 (RNE), `lo` in the low half
  typedef float f32x2c __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2c{lo, hi}, bf16x2c));
}
// (a, b) -> three packed bf16 pairs H, M, L with a = a_H + a_M + a_L exactly (every difference below is exact in float32)
__device__ inline void split3_pair(float a, float b, unsigned &H, unsigned &M, unsigned &L) {
  H = pk_bf16(a, b);
  float ra = a - __builtin_bit_cast(float, H << 16), rb = b - __builtin_bit_cast(float, H & 0xffff0000u);
  M = pk_bf16(ra, rb);
  ra -= __builtin_bit_cast(float, M << 16);
  rb -= __builtin_bit_cast(float, M & 0xffff0000u);
  L = pk_bf16(ra, rb);
}
template <int CINA, bool CACHED, bool SPLIT = false>
__global__ __launch_bounds__(256, CACHED ? (SPLIT ? (RA_PAIR8_CACHE_AHEAD ? 2 : 3) : 4) : RA_PAIR8_OCC) void conv_pair8_mfma(const PArgs a, int tiles_x, int tiles_y, int ntiles) {
  using G = NGeo<CINA>;
  constexpr int NCGA = G::NCGA;
  static_assert(!CACHED || CINA == 4, "cached form: 4 input channels");
  static_assert(!SPLIT || CACHED, "the split-precision layer B exists for the cached (steady-state) form");
  constexpr int PLANE_B = G::AHS * G::AW * 16;                 // bytes of one bf16 tile [AHS][AW][8]
  constexpr int MIDF = SPLIT ? 3 * PLANE_B / 4 : G::MID_FLOATS;  // floats of the intermediate tile(s)
  constexpr int RECA = CACHED ? 1 : CINA;  // floats per staged input pixel
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int IN_FLOATS = (G::LH * G::LW * RECA + 3) & ~3;
  float *tin = lds;              // [LH][LW] records [ksub][cg]  (channel = 4*cg + ksub); CACHED: the canvas only
  float *tmid = lds + IN_FLOATS;  // [AHS][AW] records [ksub][cg], 8 channels
  typedef typename vec_of<NCGA>::type avecA;
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  const int tid = threadIdx.x, lane = tid & 63;
  // dynamic tile tickets (a.tickets, ra_common.h): the workgroup draws its tiles from its XCD's pool instead of walking them
  __shared__ unsigned tk_sh[2];
  TicketWalk tk;
  const bool dyn = a.tickets != nullptr;
  if (dyn) tk.issue(a.tickets, ntiles);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, ksub = lane >> 4;  // A-operand side
  const int n = lane & 15, qo = lane >> 4;    // D side: column n = (p, co), rows 4*qo + r
  const int p = n >> 3, co = n & 7;
  const int per = tiles_x * tiles_y;

  // BN scale is folded into the weights and the shift into the accumulator's initial value, so
  // the epilogue of a value is one v_max (ReLU); FP32 MFMA shares the FP32 VALU lanes on gfx950
  // (tools/mfma_valu.hip: their times add), so every VALU instruction here costs MFMA time.
  const float scA = a.scA[co], scB = a.scB[co], shB = a.shB[co];
  const float loA = a.reluA ? 0.f : -__builtin_inff(), loB = a.reluB ? 0.f : -__builtin_inff();
  // Phase A runs its MFMAs with the operands SWAPPED (filter = A operand, pixels = B operand: the same lane contents, the other
  // argument order), so its accumulators are D^T: lane (pixel mA = lane & 15, channel block g4 = lane >> 4) holds the FOUR
  // channels 4 g4 .. 4 g4 + 3 of column n = (p, co), i.e. channels coA0 .. coA0 + 3 of ONE pixel (row qA of the group, column
  // 2 rA + pA).  With pixels as rows a lane held one channel of four pixels and wrote the bf16 tiles of layer B with twelve
  // 2-byte LDS stores per group; now it is three 8-byte stores (phase A was 45 % of a workgroup's life, issue-bound on them).
  const int qA = m >> 2, rA = m & 3, pA = ksub >> 1, coA0 = 4 * (ksub & 1);
  f32x4 scA4, shA4;
#pragma unroll
  for (int j = 0; j < 4; ++j) scA4[j] = a.scA[coA0 + j], shA4[j] = a.shA[coA0 + j];
  // W' of both layers, once per workgroup: one dword per (tap', cg) per lane, zero where the
  // tap misses pixel p
  // FILL (un-cached kernel with a cache pointer): the first timestep of a forward.  Its canvas is all
  // zero, so layer A's raw sums ARE the image part: they are written to the cache on the way
  // (weights left unscaled, scale / shift applied afterwards) and no separate cache kernel runs.
  const bool fill = !CACHED && a.cache != nullptr;
  float bA[CACHED ? 1 : 12][NCGA], bB[12][2];
  float bAc[3];  // CACHED: k = the 4 window columns of row ky of the canvas channel alone
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int kx = ksub - p;
    const bool ok = (kx >= 0) & (kx <= 2);
    const int tap = ok ? ky * 3 + kx : 0;
    const float w = a.wpA[((tap * NCGA + (a.plane_chan >> 2)) * 4 + (a.plane_chan & 3)) * a.CoutAP + co];
    bAc[ky] = (CACHED && ok) ? w * scA : 0.f;
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kxp = 0; kxp < 4; ++kxp) {
      const int kx = kxp - p;
      const bool ok = (kx >= 0) & (kx <= 2);
      const int tap = ok ? ky * 3 + kx : 0;
      if constexpr (!CACHED) {
#pragma unroll
        for (int cg = 0; cg < NCGA; ++cg) {
          const float w = a.wpA[((tap * NCGA + cg) * 4 + ksub) * a.CoutAP + co];
          bA[ky * 4 + kxp][cg] = ok ? w * (fill ? 1.f : scA) : 0.f;
        }
      }
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const float w = a.wpB[((tap * 2 + cg) * 4 + ksub) * a.CoutBP + co];
        bB[ky * 4 + kxp][cg] = ok ? w * scB : 0.f;
      }
    }

  // SPLIT: this lane's B operands — block ky, k-slot j = input channel j of window column kb = ksub, column n = (p, co) —
  // as three bf16 pieces (the BN scale folded in before the split)
  s16x8 wB[SPLIT ? 3 : 1][SPLIT ? 3 : 1];
  if constexpr (SPLIT) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int kx = ksub - p;
      const bool ok = (kx >= 0) & (kx <= 2);
      const int tap = ok ? ky * 3 + kx : 0;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float w0 = a.wpB[((tap * 2 + (j >> 2)) * 4 + (j & 3)) * a.CoutBP + co];
        const float w1 = a.wpB[((tap * 2 + ((j + 1) >> 2)) * 4 + ((j + 1) & 3)) * a.CoutBP + co];
        unsigned H, M, L;
        split3_pair(ok ? w0 * scB : 0.f, ok ? w1 * scB : 0.f, H, M, L);
        wB[ky][0][j] = (short)(H & 0xffffu), wB[ky][0][j + 1] = (short)(H >> 16);
        wB[ky][1][j] = (short)(M & 0xffffu), wB[ky][1][j + 1] = (short)(M >> 16);
        wB[ky][2][j] = (short)(L & 0xffffu), wB[ky][2][j + 1] = (short)(L >> 16);
      }
    }
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.src), 0, a.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.plane ? a.plane : a.src), 0, a.plane ? a.bytes_p : 0, 0x00020000);
  constexpr int NE = G::LHL * G::LWL, NIT = (NE + 255) / 256;
  // this thread's staged pixels (tile-independent): position in the loaded window and LDS record
  int e_rr[NIT], e_cc[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + 256 * i;
    e_rr[i] = e / G::LWL;
    e_cc[i] = e - e_rr[i] * G::LWL;
    if (e >= NE) e_rr[i] = -(1 << 20);  // never inside the image
  }
  // byte offset of those pixels inside the window, (rr * W + cc) * 4 (x CINA for the packed input):
  // tile-invariant, so an interior window costs ONE add per load; elements past the window carry 2^31,
  // which keeps any sum with a tile base outside the buffer (reads as 0)
  unsigned e_offp[NIT], e_offs[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const bool in = tid + 256 * i < NE;
    const unsigned o = (unsigned)(e_rr[i] * a.W + e_cc[i]) * 4u;
    e_offp[i] = in ? o : 0x80000000u;
    e_offs[i] = in ? o * CINA : 0x80000000u;
  }
  // tile index -> (image, tile row, tile column) without divisions in the loop: the stride gridDim.x
  // is decomposed once and added with carries (all scalar)
  struct TC {
    int b, ty, tx;
  };
  auto split = [&](int t) {
    TC c;
    c.b = t / per;
    const int r = t - c.b * per;
    c.ty = r / tiles_x;
    c.tx = r - c.ty * tiles_x;
    return c;
  };
  // tile walk: workgroup g takes tiles g, g + gridDim.x, ...; or, XCD-contiguous (a.xcd_map, gridDim.x % 8 == 0):
  // workgroups are dealt to the 8 XCDs round robin, so XCD x = g % 8 walks the tiles [x * chunk, (x + 1) * chunk) with its
  // gridDim.x / 8 workgroups — neighbouring tiles (shared halo rows, the cache's halo) then meet in ONE L2
  const int nwx = a.xcd_map ? (int)gridDim.x >> 3 : (int)gridDim.x;
  const int chunk = (ntiles + 7) >> 3;
  const int t_first = a.xcd_map ? ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_end = a.xcd_map ? (((int)blockIdx.x & 7) * chunk + chunk < ntiles ? ((int)blockIdx.x & 7) * chunk + chunk : ntiles) : ntiles;
  const TC stride = split(nwx);
  auto advance = [&](TC c) {
    c.tx += stride.tx;
    if (c.tx >= tiles_x) {
      c.tx -= tiles_x;
      ++c.ty;
    }
    c.ty += stride.ty;
    if (c.ty >= tiles_y) {
      c.ty -= tiles_y;
      ++c.b;
    }
    c.b += stride.b;
    return c;
  };
  f32x4 v[NIT][NCGA];
  float pv[NIT];
  // global loads of one tile's input window (tile + halo) into registers; zeros outside the image
  auto fetch = [&](const TC &c) {
    const int fb = c.b, fy0 = c.ty * G::TH - 2, fx0 = c.tx * G::TW - 3;
    const bool inside = (fy0 >= 0) & (fy0 + G::LHL <= a.H) & (fx0 >= 0) & (fx0 + G::LWL <= a.W);
    if (inside) {  // uniform
      const unsigned base = (unsigned)((fb * a.H + fy0) * a.W + fx0) * 4u;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        if constexpr (!CACHED) {
#pragma unroll
          for (int cg = 0; cg < NCGA; ++cg)
            v[i][cg] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base * CINA + e_offs[i] + 16u * cg), 0, 0));
        }
        pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)(base + e_offp[i]), 0, 0));
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int Y = fy0 + e_rr[i], X = fx0 + e_cc[i];
      const bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
      const int pix = (fb * a.H + Y) * a.W + X;
      if constexpr (!CACHED) {
#pragma unroll
        for (int cg = 0; cg < NCGA; ++cg)
          v[i][cg] = __builtin_bit_cast(
              f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (pix * CINA + 4 * cg) * 4 : 0x7fffffff, 0, 0));
      }
      pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, ok ? pix * 4 : 0x7fffffff, 0, 0));
    }
  };

  const int lane_in = CACHED ? (m >> 2) * G::LW + 2 * (m & 3) + ksub
                             : ((m >> 2) * G::LW + 2 * (m & 3)) * CINA + ksub * NCGA;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(a.cache ? a.cache : a.src), 0, a.cache ? a.bytes_c : 0, 0x00020000);
  const int lane_b = ((m & 1) * G::AW + 2 * (m >> 1) + 1) * 8 + ksub * 2;
  const int pg = a.plane_chan >> 2, slot = a.plane_chan & 3;

  // tile-invariant pieces of the cache / output addresses: per group slot (scalar) and per lane
  int slot_c[G::GPW];
#pragma unroll
  for (int s = 0; s < G::GPW; ++s) {
    int gi = wave + 4 * s;
    if (gi >= G::NGA) gi = G::NGA - 1;
    const int gr = gi / G::AGX, gc = gi - gr * G::AGX;
    slot_c[s] = (4 * gr * a.cache_gx + gc) * 256;
  }
  const int lane_c = qA * a.cache_gx * 256 + rA * 64 + ksub * 16;  // cell (row, column group): [pair r][n] floats, this lane's n = 4 g4 ..
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes_y, 0x00020000);
  const unsigned lane_y = co < a.CoutB ? (unsigned)(((2 * qo + p) * a.CoutB + co) * 4) : 0x80000000u;
  int g_y[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) g_y[g] = ((g >> 1) * a.Wo + 8 * (g & 1)) * a.CoutB * 4;

  int tile = t_first;
  if (dyn) {
    tk.begin(tk_sh);
    tile = tk.cur;
  }
  bool have = dyn ? tile >= 0 : tile < t_end;
  TC cur = split(have ? tile : 0), nxt = cur;
  if (have) fetch(cur);
  RA_P8_DECL;
  // the padded groups read LDS this kernel never writes; whatever an earlier kernel left there
  // must not be NaN/Inf (their results are discarded, but keep the arithmetic clean)
  for (int e = tid; e < (IN_FLOATS + MIDF) / 4; e += 256)
    reinterpret_cast<f32x4 *>(lds)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // the rider: this workgroup's share of the constant fill, dealt over its tiles
  const bool rider = !CACHED && a.rider_dst != nullptr;
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rider ? a.rider_dst : a.y, 0,
                                                                       rider ? a.rider_quads * 16 : 0, 0x00020000);
  const int r_chunk = rider ? (a.rider_quads + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  int r_idx = (int)blockIdx.x * r_chunk + tid;
  const int r_end = ((int)blockIdx.x + 1) * r_chunk < a.rider_quads ? ((int)blockIdx.x + 1) * r_chunk : a.rider_quads;
  const int my_tiles = t_end > t_first ? (t_end - t_first + nwx - 1) / nwx : 1;
  const int r_per_tile = (r_chunk + 256 * my_tiles - 1) / (256 * my_tiles);
  const u32x4r r_bits = __builtin_bit_cast(u32x4r, f32x4{a.rider_val, a.rider_val, a.rider_val, a.rider_val});
  bool have_n = false;
  f32x4 cpre[CACHED ? G::GPW : 1];
  auto load_cache = [&](const TC &tc) {
    const int tile_c0 = ((tc.b * a.cache_rows + tc.ty * G::TH) * a.cache_gx + ((tc.tx * G::TW) >> 3)) * 256;
#pragma unroll
    for (int s = 0; s < (CACHED ? G::GPW : 1); ++s)
      cpre[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, tile_c0 + slot_c[s] + lane_c, 0, 0));
  };
#if RA_PAIR8_CACHE_AHEAD
  if constexpr (CACHED) {
    if (have) load_cache(cur);
  }
#endif
  for (; have; tile = dyn ? (tk.step(), tk.cur) : tile + nwx, cur = nxt, have = have_n) {
    const int b = cur.b, ty0 = cur.ty * G::TH, tx0 = cur.tx * G::TW;
    // CACHED: this tile's cached sums of layer A — 32 bytes per pixel, the launch's largest read — were requested a whole
    // phase B ahead (round 6; RA_PAIR8_CACHE_AHEAD=0 at build time: at the top of the tile, one staging pass and a barrier ahead
    // of their use, as in round 5 — the probe's "waiting for the cached sums 14 %")
#if !RA_PAIR8_CACHE_AHEAD
    if constexpr (CACHED) load_cache(cur);
#endif

    // ---------------- stage layer A's input window (prefetched registers -> LDS) ----------------
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if constexpr (CACHED) {
        if (e_rr[i] >= 0) tin[e_rr[i] * G::LW + e_cc[i]] = pv[i];
        continue;
      }
      if (a.plane) {
#pragma unroll
        for (int cg = 0; cg < NCGA; ++cg) {  // selects, not runtime register indexing
          v[i][cg].x = (cg == pg && slot == 0) ? pv[i] : v[i][cg].x;
          v[i][cg].y = (cg == pg && slot == 1) ? pv[i] : v[i][cg].y;
          v[i][cg].z = (cg == pg && slot == 2) ? pv[i] : v[i][cg].z;
          v[i][cg].w = (cg == pg && slot == 3) ? pv[i] : v[i][cg].w;
        }
      }
      if (e_rr[i] >= 0) {
        float *rec = tin + (e_rr[i] * G::LW + e_cc[i]) * CINA;
        if constexpr (NCGA == 1) {
          *reinterpret_cast<f32x4 *>(rec) = v[i][0];
        } else {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int cg = 0; cg < NCGA; ++cg) rec[ks * NCGA + cg] = v[i][cg][ks];
        }
      }
    }
    if (dyn) tk.publish(tk_sh);
    RA_P8_SYNC();
    RA_P8_AT(0);  // staged + barrier
    if (dyn) {
      tk.read_next(tk_sh);
      tk.request();  // the draw for the tile after next: older than the prefetch loads below, in flight across this tile
      have_n = tk.nxt >= 0;
      nxt = split(have_n ? tk.nxt : 0);
    } else {
      have_n = tile + nwx < t_end;
      nxt = advance(cur);
    }
    if (have_n) fetch(nxt);  // the next tile's loads fly while this one is computed
    if constexpr (!CACHED) {
      if (rider)
        for (int u = 0; u < r_per_tile; ++u, r_idx += 256)
          __builtin_amdgcn_raw_buffer_store_b128(r_bits, rr, r_idx < r_end ? r_idx * 16 : 0x7fffffff, 0, 0);
    }

    // ---------------- phase A: layer A on the 18 x 36 region -> tmid ----------------
    {
      const bool interior = (ty0 >= 1) & (ty0 + G::TH + 1 <= a.H) & (tx0 >= 2) & (tx0 + G::TW + 1 <= a.W);
      const int tile_c = ((b * a.cache_rows + ty0) * a.cache_gx + (tx0 >> 3)) * 256;  // (the FILL form's cache stores)
      f32x4 acc[G::GPW];
      int gin[G::GPW];
#pragma unroll
      for (int s = 0; s < G::GPW; ++s) {
        int gi = wave + 4 * s;
        if (gi >= G::NGA) gi = G::NGA - 1;  // duplicate work, masked at the store
        const int gr = gi / G::AGX, gc = gi - gr * G::AGX;
        gin[s] = (4 * gr * G::LW + 8 * gc) * RECA + lane_in;
        if constexpr (CACHED) {
          // this lane's 4 partial sums (columns n = 4 g4 .. + 3 of pixel mA of the group) are one float4 of the cache:
          // [image][row ty0-1+4gr+qA (+1)][column group tx0/8+gc][pair rA][n]
          // (byte offset = tile part + slot part + lane part; only the first changes per tile)
          acc[s] = cpre[s] * scA4 + shA4;
        } else {
          acc[s] = fill ? f32x4{0.f, 0.f, 0.f, 0.f} : shA4;
        }
      }
      RA_P8_AT(1);  // layer A's cached sums have arrived (accumulators initialised)
      if constexpr (CACHED) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          float av[G::GPW];
#pragma unroll
          for (int s = 0; s < G::GPW; ++s) av[s] = tin[gin[s] + ky * G::LW];
#pragma unroll
          for (int s = 0; s < G::GPW; ++s)
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(bAc[ky], av[s], acc[s], 0, 0, 0);  // D^T: rows = (p, co), columns = pixels
        }
      } else {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kxp = 0; kxp < 4; ++kxp) {
          avecA av[G::GPW];
#pragma unroll
          for (int s = 0; s < G::GPW; ++s)
            av[s] = *reinterpret_cast<const avecA *>(&tin[gin[s] + (ky * G::LW + kxp) * CINA]);
#pragma unroll
          for (int cg = 0; cg < NCGA; ++cg)
#pragma unroll
            for (int s = 0; s < G::GPW; ++s)
              acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(bA[ky * 4 + kxp][cg], av[s][cg], acc[s], 0, 0, 0);
        }
      }
      if (fill) {  // uniform: raw sums -> cache, then the folded scale / shift
#pragma unroll
        for (int s = 0; s < G::GPW; ++s) {
          int gi = wave + 4 * s;
          if (gi >= G::NGA) gi = G::NGA - 1;
          const int gr = gi / G::AGX, gc = gi - gr * G::AGX;
          // a tile stores only cells whose four pixel pairs it computed from a complete window: its
          // rows 0..17 (16 / 17 are computed identically by the tile below) and column groups 0..3;
          // group 4 (pairs 2, 3 reach past the staged window) belongs to the tile on the right, except
          // in the last tile column, where those pairs lie outside the image
          const bool mine = (4 * gr + qA < G::AHS) & ((gc < 4) | (tx0 + G::TW >= a.W));
          const int off = mine ? tile_c + slot_c[s] + lane_c : 0x7fffffff;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, acc[s]), rc, off, 0, 0);
          acc[s] = acc[s] * scA4 + shA4;
        }
      }
#pragma unroll
      for (int s = 0; s < G::GPW; ++s) {
        const int gi = wave + 4 * s;
        const int gr = gi / G::AGX, gc = gi - gr * G::AGX;
        const bool live = (gi < G::NGA) & (4 * gr + qA < G::AHS);
        float o[4];  // channels coA0 .. coA0 + 3 of the pixel (row 4 gr + qA, column 8 gc + 2 rA + pA) of the region
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(acc[s][j], loA);
        if (!interior) {  // outside the image the intermediate is layer B's SAME padding: zero
          const int Y = ty0 - 1 + 4 * gr + qA, X = tx0 - 2 + 8 * gc + 2 * rA + pA;
          const bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ok ? o[j] : 0.f;
        }
        const int pix = (4 * gr + qA) * G::AW + 8 * gc + 2 * rA + pA;
        if constexpr (SPLIT) {
          // three bf16 tiles [pixel][channel]: the lane's four channels are 8 contiguous bytes of the pixel's record in each
          unsigned H01, M01, L01, H23, M23, L23;
          split3_pair(o[0], o[1], H01, M01, L01);
          split3_pair(o[2], o[3], H23, M23, L23);
          if (live) {
            typedef unsigned u32x2t __attribute__((ext_vector_type(2)));
            unsigned char *d0 = reinterpret_cast<unsigned char *>(tmid) + pix * 16 + coA0 * 2;
            *reinterpret_cast<u32x2t *>(d0) = u32x2t{H01, H23};
            *reinterpret_cast<u32x2t *>(d0 + PLANE_B) = u32x2t{M01, M23};
            *reinterpret_cast<u32x2t *>(d0 + 2 * PLANE_B) = u32x2t{L01, L23};
          }
        } else if (live) {
          // float32 tile, records [ksub][cg] (channel c at 2 (c & 3) + (c >> 2)): this lane's channels sit two floats apart
          float *dst = tmid + pix * 8 + (ksub & 1);
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[2 * j] = o[j];
        }
      }
    }
    RA_P8_AT(2);  // phase A computed and written to the LDS tile
    RA_P8_SYNC();
    RA_P8_AT(3);  // barrier

    // ---------------- phase B: layer B out of tmid, BN + ReLU + 2x2 max-pool -> global ----------------
#if RA_PAIR8_CACHE_AHEAD
    if constexpr (CACHED) {
      if (have_n) load_cache(nxt);  // phase A has consumed cpre; the next tile's sums fly across phase B and the next staging
    }
#endif
    {
      f32x4 acc[4];
      int gmid[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int gy = 2 * wave + (g >> 1), gx = g & 1;
        gmid[g] = (2 * gy * G::AW + 16 * gx) * 8 + lane_b;
        acc[g] = f32x4{shB, shB, shB, shB};
      }
      if constexpr (SPLIT) {
        // lane (m, kb): pixel (row (m & 1) + ky, column 2 (m >> 1) + 1 + kb) of the group, its 8 channels = one 16-byte read
        const unsigned char *tb = reinterpret_cast<const unsigned char *>(tmid);
        const int lane_px = ((m & 1) * G::AW + 2 * (m >> 1) + 1 + ksub) * 16;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int gh = 0; gh < 2; ++gh) {  // two pixel groups at a time: 24 operand registers in flight instead of 48
            s16x8 av[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int g = 2 * gh + u;
              const int gy = 2 * wave + (g >> 1), gx = g & 1;
              const int off = ((2 * gy + ky) * G::AW + 16 * gx) * 16 + lane_px;
#pragma unroll
              for (int pc = 0; pc < 3; ++pc) av[u][pc] = *reinterpret_cast<const s16x8 *>(tb + off + pc * PLANE_B);
            }
            // six piece products per block, smallest first; consecutive MFMAs alternate between the two accumulators
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
              for (int u = 0; u < 2; ++u)
                acc[2 * gh + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8v, av[u][PA[t]]),
                                                                          __builtin_bit_cast(bf16x8v, wB[ky][PB[t]]), acc[2 * gh + u], 0, 0, 0);
          }
      } else {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kxp = 0; kxp < 4; ++kxp) {
          f32x2 av[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
            av[g] = *reinterpret_cast<const f32x2 *>(&tmid[gmid[g] + (ky * G::AW + kxp) * 8]);
#pragma unroll
          for (int cg = 0; cg < 2; ++cg)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][cg], bB[ky * 4 + kxp][cg], acc[g], 0, 0, 0);
        }
      }
      const int prow0 = (ty0 >> 1) + wave * 2, pcol0 = (tx0 >> 1) + 2 * qo + p;
      const bool whole = ((ty0 >> 1) + G::TH / 2 <= a.Ho) & ((tx0 >> 1) + G::TW / 2 <= a.Wo);  // uniform
      const unsigned tile_y = (unsigned)(((b * a.Ho + prow0) * a.Wo + (tx0 >> 1)) * a.CoutB * 4) + lane_y;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // ReLU commutes with max: pool first.  registers (2j, 2j+1) = rows (0, 1) of pair
        // 2*qo + j; lane n^8 holds the pair's other pixel
        const float t0 = fmaxf(fmaxf(acc[g][0], acc[g][1]), loB), t1 = fmaxf(fmaxf(acc[g][2], acc[g][3]), loB);
        const float u0 = fmaxf(t0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t0), 0x128, 0xf, 0xf, true)));
        const float u1 = fmaxf(t1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, t1), 0x128, 0xf, 0xf, true)));
        const float ov = p ? u1 : u0;  // lane (p, co) stores pooled pixel 2*qo + p
        const int prow = prow0 + (g >> 1), pcol = pcol0 + 8 * (g & 1);
        unsigned off = tile_y + (unsigned)g_y[g];
        if (!whole) off = ((prow < a.Ho) & (pcol < a.Wo)) ? off : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov), ry, (int)off, 0, 0);
      }
    }
    RA_P8_AT(4);  // phase B, pooled and stored
  }
  RA_P8_END;
  if constexpr (!CACHED) {
    if (rider)  // what rounding left of this workgroup's share (and all of it for a workgroup without tiles)
      for (; r_idx < r_end; r_idx += 256) __builtin_amdgcn_raw_buffer_store_b128(r_bits, rr, r_idx * 16, 0, 0);
  }
}

template <int CINA, bool CACHED = false, bool SPLIT = false>
int launch8(const PArgs &a_in, int B, hipStream_t st) {
  using G = NGeo<CINA>;
  PArgs a = a_in;
  a.bytes_y = (int)((size_t)B * a.Ho * a.Wo * a.CoutB * sizeof(float));
  auto kern = conv_pair8_mfma<CINA, CACHED, SPLIT>;
  constexpr size_t lds = (size_t)(((G::LH * G::LW * (CACHED ? 1 : CINA) + 3) & ~3) + (SPLIT ? 3 * G::AHS * G::AW * 4 : G::MID_FLOATS)) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int tiles_x = ceil_div(a.W, G::TW), tiles_y = ceil_div(a.H, G::TH);
  const int ntiles = tiles_x * tiles_y * B;
  static int wgs = -1;  // RA_PAIR8_WGS: tuning aid, persistent workgroups (default 3 per CU)
  if (wgs < 0) {
    const char *e = getenv("RA_PAIR8_WGS");
    // the cached form (122 VGPRs) could run 4 workgroups per CU and is 0.3 us faster alone that way, but 3 leave
    // room for the kernels of the other decode graphs: 50.4k vs 49.7k instance-timesteps/s with four batches in flight
    wgs = e ? atoi(e) : 768;
  }
  static int xcd = -1;  // RA_PAIR8_XCD=0: tuning aid, the interleaved tile walk (51.7k vs 52.1k instance-timesteps/s pipelined)
  if (xcd < 0) {
    const char *e = getenv("RA_PAIR8_XCD");
    xcd = e ? atoi(e) : 1;
  }
  PArgs a2 = a;
  const int grid = ntiles < wgs ? ntiles : wgs;
  a2.xcd_map = (xcd && grid % 8 == 0 && grid >= 8) ? 1 : 0;
  a2.tickets = (CACHED && ntiles >= kTicketMinTilesPerWg * grid) ? take_ticket_slots(1, grid) : nullptr;  // the steady-state form; bound scratch only
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a2, tiles_x, tiles_y, ntiles);
  return launch_status("ra_conv_pair_f32");
}

// Timestep-invariant partial sums of the N-packed pair's layer A (see conv_pair8_mfma, CACHED):
//   S[b][Y+1][gx][r][n = p*8 + co] = sum_{ky,kx} sum_{ci != plane_chan} x[b][Y+ky-1][X+kx-1][ci] * W[ky][kx][ci][co]
// with X = 8*gx - 2 + 2*r + p (SAME zero padding; 0 for pixels outside the image), i.e. exactly
// the float4 a lane of phase A initialises its accumulator with.  The cache must be zero-filled when
// allocated: entries outside the image are never written.
__global__ __launch_bounds__(256) void first_cache_kernel(const float *x, const float *wpA, int CoutAP, int plane_chan,
                                                          int B, int H, int W, int rows, int ngx, float *cache) {
  // a workgroup = 4 image rows x 64 columns starting at X = 64*bx - 2, i.e. 8 complete column
  // groups of the cache.  One thread per pixel computes all 8 output channels (9 float4 loads,
  // 27 x 8 FMAs with wave-uniform weights), the block is transposed through LDS and written as
  // coalesced float4 [n][r] records.
  __shared__ float sm[4][64][9];  // +1: conflict-free transposed reads
  const int xl = threadIdx.x & 63, yl = threadIdx.x >> 6;
  const int X = blockIdx.x * 64 - 2 + xl, Y = blockIdx.y * 4 + yl, b = blockIdx.z;
  float acc[8];
#pragma unroll
  for (int co = 0; co < 8; ++co) acc[co] = 0.f;
  if (X >= 0 && X < W && Y < H) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = Y + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = X + kx - 1;
        const bool ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4 *>(x + ((size_t)(b * H + yy) * W + xx) * 4);
        const float *wt = wpA + (size_t)((ky * 3 + kx) * 4) * CoutAP;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const float xv = ci == plane_chan ? 0.f : v[ci];
#pragma unroll
          for (int co = 0; co < 8; ++co) acc[co] = fmaf(xv, wt[ci * CoutAP + co], acc[co]);
        }
      }
    }
  }
#pragma unroll
  for (int co = 0; co < 8; ++co) sm[yl][xl][co] = acc[co];
  __syncthreads();
  for (int e = threadIdx.x; e < 4 * 8 * 16; e += 256) {
    const int n4 = e & 3, r = (e >> 2) & 3, g = (e >> 4) & 7, row = e >> 7;  // columns n = 4 n4 .. 4 n4 + 3 of pair r
    const int p = n4 >> 1, co0 = 4 * (n4 & 1);
    const int Yo = blockIdx.y * 4 + row, gx = blockIdx.x * 8 + g;
    if (Yo < H && gx < ngx) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = sm[row][8 * g + 2 * r + p][co0 + j];
      *reinterpret_cast<f32x4 *>(cache + (((size_t)b * rows + (Yo + 1)) * ngx + gx) * 64 + r * 16 + 4 * n4) = o;
    }
  }
}

inline void cache_dims(int H, int W, int &rows, int &ngx) {
  rows = ceil_div(H, NGeo<4>::TH) * NGeo<4>::TH + 4;
  ngx = ceil_div(W, NGeo<4>::TW) * (NGeo<4>::TW / 8) + 1;
}

template <int CINA, int CMID, int NCB, int GX, int GYB>
int launch(const PArgs &a, int B, hipStream_t st) {
  using G = PGeo<CINA, CMID, NCB, GX, GYB>;
  auto kern = conv_pair_mfma<CINA, CMID, NCB, GX, GYB>;
  constexpr size_t lds = (size_t)(G::IN_FLOATS + G::MID_FLOATS) * sizeof(float);
  static_assert(lds <= 160 * 1024, "pair tile does not fit LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int tiles_x = ceil_div(a.W, G::TWB), tiles_y = ceil_div(a.H, G::THB);
  const int ntiles = tiles_x * tiles_y * B;
  if constexpr (G::NCHA == 1 && G::NCHB == 1) {
    // plain single-chunk pairs (controller CNN L2+L3 at full size): persistent workgroups, weights
    // loaded once, the next tile's input prefetched into registers behind the MFMA phases
    static int pers = -1, cap = 0;
    if (pers < 0) {
      const char *e = getenv("RA_PAIR_PERSIST");
      pers = e ? atoi(e) : 1;
      auto kp = conv_pair_persist_mfma<CINA, CMID, NCB, GX, GYB>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kp, 256, lds) != hipSuccess || nb < 1) nb = 1;
      hipDeviceProp_t prop;
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      cap = nb * cus;
    }
    if (pers && !a.ups && !a.plane && ntiles > cap && a.bytes0 > 0) {
      hipLaunchKernelGGL((conv_pair_persist_mfma<CINA, CMID, NCB, GX, GYB>), dim3(cap), dim3(256), lds, st, a, tiles_x,
                         tiles_y, ntiles);
      return launch_status("ra_conv_pair_f32");
    }
  }
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(256), lds, st, a, tiles_x, tiles_y);
  return launch_status("ra_conv_pair_f32");
}

template <int CINA, int CMID, int NCB>
int dispatch_geo(const PArgs &a, int B, hipStream_t st) {
  using Big = PGeo<CINA, CMID, NCB, 4, 2>;
  constexpr bool big_fits = (size_t)(Big::IN_FLOATS + Big::MID_FLOATS) * 4 <= 80 * 1024;
  const bool narrow = (a.W % 32 != 0) && (a.W % 32 <= 16);
  auto wgs = [&](int gx, int gyb) { return (long)ceil_div(a.W, 8 * gx) * ceil_div(a.H, 8 * gyb) * B; };
  static int force = -1;  // RA_PAIR_GEO=<gx><gyb>: tuning aid
  if (force < 0) {
    const char *e = getenv("RA_PAIR_GEO");
    force = e ? atoi(e) : 0;
  }
  // single-chunk plain pairs run the persistent kernel, which measures fastest with the 8-row tile
  // (cfg2 L2+L3: 41.8 us at 32x8 against 47.4 at 32x16 and 45.3 one-shot; profiles/r02)
  constexpr bool single_chunk = Big::NCHA == 1 && Big::NCHB == 1;
  if (!force && single_chunk && !a.ups && !a.plane && !narrow && wgs(4, 1) >= 2048)
    return launch<CINA, CMID, NCB, 4, 1>(a, B, st);
  if constexpr (big_fits) {
    if (force == 42 || (!force && !narrow && wgs(4, 2) >= 512)) return launch<CINA, CMID, NCB, 4, 2>(a, B, st);
  }
  if (force == 41 || (!force && !narrow)) return launch<CINA, CMID, NCB, 4, 1>(a, B, st);
  if (force == 22 || (!force && wgs(2, 2) >= 512)) return launch<CINA, CMID, NCB, 2, 2>(a, B, st);
  return launch<CINA, CMID, NCB, 2, 1>(a, B, st);
}

template <int CINA, int CMID>
int dispatch_b(const PArgs &a, int B, hipStream_t st) {
  if (a.CoutBP == 16) return dispatch_geo<CINA, CMID, 1>(a, B, st);
  if (a.CoutBP == 32) return dispatch_geo<CINA, CMID, 2>(a, B, st);
  return fail(RA_E_SHAPE, "ra_conv_pair_f32: CoutB %d unsupported", a.CoutB);
}

template <int CINA>
int dispatch_mid(const PArgs &a, int cmid, int B, hipStream_t st) {
  switch (cmid) {
    case 8: return dispatch_b<CINA, 8>(a, B, st);
    case 16: return dispatch_b<CINA, 16>(a, B, st);
    case 32: return dispatch_b<CINA, 32>(a, B, st);
    default: return fail(RA_E_SHAPE, "ra_conv_pair_f32: CoutA %d unsupported", cmid);
  }
}

}  // namespace cpair
}  // namespace ra

using namespace ra;

extern "C" int ra_conv_pair_supported(int Cin, int CoutA, int CoutB) {
  const bool cin_ok = Cin == 4 || Cin == 8 || Cin == 16 || Cin == 32;
  const bool mid_ok = CoutA == 8 || CoutA == 16 || CoutA == 32;
  return cin_ok && mid_ok && CoutB >= 1 && CoutB <= 32;
}

extern "C" int ra_conv_pair_f32(const float *src, int Cin, int B, int Hs, int Ws, int upsampleA,
                                const float *wpA, const float *scaleA, const float *shiftA, int CoutA,
                                int reluA, const float *wpB, const float *scaleB, const float *shiftB,
                                int CoutB, int reluB, int poolB, const float *plane, int plane_chan,
                                float *y, void *stream) {
  if (!src || !wpA || !scaleA || !shiftA || !wpB || !scaleB || !shiftB || !y || B <= 0 || Hs <= 0 ||
      Ws <= 0)
    return fail(RA_E_INVALID, "ra_conv_pair_f32: bad argument");
  if (!ra_conv_pair_supported(Cin, CoutA, CoutB))
    return fail(RA_E_SHAPE, "ra_conv_pair_f32: Cin=%d CoutA=%d CoutB=%d", Cin, CoutA, CoutB);
  if (poolB != 1 && poolB != 2) return fail(RA_E_SHAPE, "ra_conv_pair_f32: pool %d", poolB);
  cpair::PArgs a{};
  a.src = src;
  a.y = y;
  a.wpA = wpA;
  a.scA = scaleA;
  a.shA = shiftA;
  a.wpB = wpB;
  a.scB = scaleB;
  a.shB = shiftB;
  a.C0 = Cin;
  a.Hs = Hs;
  a.Ws = Ws;
  a.ups = upsampleA ? 1 : 0;
  a.H = Hs * (1 + a.ups);
  a.W = Ws * (1 + a.ups);
  if (poolB == 2 && ((a.H | a.W) & 1)) return fail(RA_E_SHAPE, "ra_conv_pair_f32: odd size with pool 2");
  a.CoutAP = ra_conv_cout_padded(CoutA);
  a.CoutB = CoutB;
  a.CoutBP = ra_conv_cout_padded(CoutB);
  a.poolB = poolB;
  a.Ho = a.H / poolB;
  a.Wo = a.W / poolB;
  a.reluA = reluA;
  a.reluB = reluB;
  a.plane = plane;
  a.plane_chan = plane_chan;
  if (plane && (plane_chan < 0 || plane_chan >= Cin)) return fail(RA_E_INVALID, "ra_conv_pair_f32: plane channel");
  hipStream_t st = as_stream(stream);
  const size_t bytes0 = (size_t)B * Hs * Ws * Cin * 4;
  a.bytes0 = (int)bytes0;
  a.bytes_p = (int)((size_t)B * Hs * Ws * 4);
  a.cache = nullptr;
  a.cache_rows = a.cache_gx = a.bytes_c = 0;
  static int no8 = -1;  // RA_PAIR_NO8=1: tuning aid, disables the N-packed kernel
  if (no8 < 0) no8 = getenv("RA_PAIR_NO8") ? 1 : 0;
  if (!no8 && CoutA == 8 && CoutB <= 8 && poolB == 2 && !a.ups && bytes0 < (1u << 31) && a.W > 16) {
    if (Cin == 4) return cpair::launch8<4>(a, B, st);
    if (Cin == 8) return cpair::launch8<8>(a, B, st);
  }
  switch (Cin) {
    case 4: return cpair::dispatch_mid<4>(a, CoutA, B, st);
    case 8: return cpair::dispatch_mid<8>(a, CoutA, B, st);
    case 16: return cpair::dispatch_mid<16>(a, CoutA, B, st);
    default: return cpair::dispatch_mid<32>(a, CoutA, B, st);
  }
}

// ---------------------------------------------------------------------------------------------
// CACHED form of the N-packed first pair (controller CNN L0 + L1 on the 4-channel packed image):
// ra_conv_first_cache_f32 once per forward, ra_conv_pair_cached_f32 per timestep.
extern "C" int ra_conv_first_cache_supported(int Cin, int CoutA, int CoutB, int poolB, int H, int W) {
  return Cin == 4 && CoutA == 8 && CoutB >= 1 && CoutB <= 8 && poolB == 2 && W > 16 && !((H | W) & 1);
}

extern "C" size_t ra_conv_first_cache_floats(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  int rows, ngx;
  cpair::cache_dims(H, W, rows, ngx);
  return (size_t)B * rows * ngx * 64;
}

extern "C" int ra_conv_first_cache_f32(const float *src, int B, int H, int W, const float *wpA, int CoutA,
                                       int plane_chan, float *cache, void *stream) {
  if (!src || !wpA || !cache || B <= 0 || H <= 0 || W <= 0 || plane_chan < 0 || plane_chan > 3)
    return fail(RA_E_INVALID, "ra_conv_first_cache_f32: bad argument");
  if (CoutA != 8) return fail(RA_E_SHAPE, "ra_conv_first_cache_f32: CoutA %d", CoutA);
  int rows, ngx;
  cpair::cache_dims(H, W, rows, ngx);
  const size_t total = (size_t)B * rows * ngx * 16;
  if (total * 16 >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_conv_first_cache_f32: cache exceeds 2 GiB");
  hipLaunchKernelGGL(cpair::first_cache_kernel, dim3(ceil_div(W + 2, 64), ceil_div(H, 4), B), dim3(256), 0,
                     as_stream(stream), src, wpA, ra_conv_cout_padded(CoutA), plane_chan, B, H, W, rows, ngx, cache);
  return launch_status("ra_conv_first_cache_f32");
}

extern "C" int ra_conv_pair_fill_cache_f32(const float *src, const float *plane, int plane_chan, int B, int H, int W,
                                           const float *wpA, const float *scaleA, const float *shiftA, int reluA,
                                           const float *wpB, const float *scaleB, const float *shiftB, int CoutB,
                                           int reluB, float *cache, float *y, void *stream) {
  return ra_conv_pair_fill_cache_rider_f32(src, plane, plane_chan, B, H, W, wpA, scaleA, shiftA, reluA, wpB, scaleB, shiftB,
                                           CoutB, reluB, cache, y, nullptr, 0, 0.0f, stream);
}

extern "C" int ra_conv_pair_fill_cache_rider_f32(const float *src, const float *plane, int plane_chan, int B, int H, int W,
                                                 const float *wpA, const float *scaleA, const float *shiftA, int reluA,
                                                 const float *wpB, const float *scaleB, const float *shiftB, int CoutB,
                                                 int reluB, float *cache, float *y, float *fill_dst, size_t fill_floats,
                                                 float fill_value, void *stream) {
  if (!src || !plane || !cache || !wpA || !scaleA || !shiftA || !wpB || !scaleB || !shiftB || !y || B <= 0)
    return fail(RA_E_INVALID, "ra_conv_pair_fill_cache_f32: bad argument");
  if (fill_dst && ((reinterpret_cast<uintptr_t>(fill_dst) & 15) || (fill_floats & 3) || fill_floats * 4 >= (1ull << 31)))
    return fail(RA_E_SHAPE, "ra_conv_pair_fill_cache_rider_f32: the fill must be 16-byte aligned, a multiple of 4 floats, < 2 GiB");
  if (!ra_conv_first_cache_supported(4, 8, CoutB, 2, H, W) || plane_chan < 0 || plane_chan > 3)
    return fail(RA_E_SHAPE, "ra_conv_pair_fill_cache_f32: unsupported shape");
  cpair::PArgs a{};
  a.src = src;
  a.y = y;
  a.wpA = wpA;
  a.scA = scaleA;
  a.shA = shiftA;
  a.wpB = wpB;
  a.scB = scaleB;
  a.shB = shiftB;
  a.C0 = 4;
  a.Hs = a.H = H;
  a.Ws = a.W = W;
  a.ups = 0;
  a.CoutAP = ra_conv_cout_padded(8);
  a.CoutB = CoutB;
  a.CoutBP = ra_conv_cout_padded(CoutB);
  a.poolB = 2;
  a.Ho = H / 2;
  a.Wo = W / 2;
  a.reluA = reluA;
  a.reluB = reluB;
  a.plane = plane;
  a.plane_chan = plane_chan;
  const size_t b0 = (size_t)B * H * W * 4 * sizeof(float);
  if (b0 >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_conv_pair_fill_cache_f32: input exceeds 2 GiB");
  a.bytes0 = (int)b0;
  a.bytes_p = (int)((size_t)B * H * W * 4);
  a.cache = cache;
  cpair::cache_dims(H, W, a.cache_rows, a.cache_gx);
  const size_t cb = (size_t)B * a.cache_rows * a.cache_gx * 64 * sizeof(float);
  if (cb >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_conv_pair_fill_cache_f32: cache exceeds 2 GiB");
  a.bytes_c = (int)cb;
  a.rider_dst = fill_floats ? fill_dst : nullptr;
  a.rider_quads = (int)(fill_floats / 4);
  a.rider_val = fill_value;
  return cpair::launch8<4, false>(a, B, as_stream(stream));
}

extern "C" int ra_conv_pair_cached_f32(const float *cache, const float *plane, int plane_chan, int B, int H, int W,
                                       const float *wpA, const float *scaleA, const float *shiftA, int reluA,
                                       const float *wpB, const float *scaleB, const float *shiftB, int CoutB,
                                       int reluB, float *y, void *stream) {
  if (!cache || !plane || !wpA || !scaleA || !shiftA || !wpB || !scaleB || !shiftB || !y || B <= 0)
    return fail(RA_E_INVALID, "ra_conv_pair_cached_f32: bad argument");
  if (!ra_conv_first_cache_supported(4, 8, CoutB, 2, H, W) || plane_chan < 0 || plane_chan > 3)
    return fail(RA_E_SHAPE, "ra_conv_pair_cached_f32: unsupported shape");
  cpair::PArgs a{};
  a.src = plane;  // unused by the cached form (only the canvas plane is staged)
  a.y = y;
  a.wpA = wpA;
  a.scA = scaleA;
  a.shA = shiftA;
  a.wpB = wpB;
  a.scB = scaleB;
  a.shB = shiftB;
  a.C0 = 4;
  a.Hs = a.H = H;
  a.Ws = a.W = W;
  a.ups = 0;
  a.CoutAP = ra_conv_cout_padded(8);
  a.CoutB = CoutB;
  a.CoutBP = ra_conv_cout_padded(CoutB);
  a.poolB = 2;
  a.Ho = H / 2;
  a.Wo = W / 2;
  a.reluA = reluA;
  a.reluB = reluB;
  a.plane = plane;
  a.plane_chan = plane_chan;
  a.bytes0 = 0;
  a.bytes_p = (int)((size_t)B * H * W * 4);
  a.cache = cache;
  cpair::cache_dims(H, W, a.cache_rows, a.cache_gx);
  const size_t cb = (size_t)B * a.cache_rows * a.cache_gx * 64 * sizeof(float);
  if (cb >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_conv_pair_cached_f32: cache exceeds 2 GiB");
  a.bytes_c = (int)cb;
  static int split = -1;  // RA_PAIR8_SPLIT=0: layer B on the float32 MFMA (rounds 2-4) instead of the split-precision bf16 form
  if (split < 0) {
    const char *e = getenv("RA_PAIR8_SPLIT");
    split = e ? atoi(e) : 1;
  }
  if (split) return cpair::launch8<4, true, true>(a, B, as_stream(stream));
  return cpair::launch8<4, true>(a, B, as_stream(stream));
}
