// K1w — 3x3 convolution as Winograd F(2x2, 3x3) on the f32 MFMA, for the mid-resolution layers of the
// controller CNN (nnlib.cnn: conv3x3 SAME + BN + ReLU + max-pool, nnlib.py:229-253).
//
// Why: the decode pipeline is bound by the controller CNN, the CNN by the f32 matrix core (157 TF/s;
// FP32 VALU time adds to it).  F(2x2, 3x3) does 16 multiplies per 2x2 outputs and channel pair where the
// direct form does 36: 2.25x fewer MFMAs, paid with adds on the VALU (input transform 8 per MFMA
// group, output transform spread over all lanes).  Exact-arithmetic identity; in float32 the result
// differs from the direct kernel by summation order and the 0.5 factors of G (~1e-6 relative).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        per 2x2 output tile and output channel
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
//
// Work split (the round-1 cut kept all 16 transformed filters per wave and ran one wave per SIMD with
// 140 KB of LDS): a workgroup owns a 16x16 output tile = 64 Winograd tiles = 4 MFMA row blocks, and
// 32 output channels; WAVE p owns ROW p of the 4x4 transform domain:
//   * its B operands U[p][q][ci][co] (q = 0..3) stay in registers for the whole launch (64 VGPRs at Cin = 32);
//   * per row block and k-step it reads 2 of the 4 patch rows from LDS (8 ds_read_b32), forms the 4
//     transformed values V[p][q] (8 adds) and issues 8 MFMAs (4 q x 2 cout blocks);
//   * it applies the q half of A^T in registers (T[p][j], 2 values from 4) and hands T to LDS;
//   * after a barrier all 256 threads finish the p half of A^T for (tile, cout) pairs, apply the folded
//     BatchNorm scale / shift, ReLU and the 2x2 max-pool (one Winograd tile IS one pooling window)
//     and store channel-contiguous.
// LDS: (TSY + 2) x 18 staged pixels x (Cin + 2) floats (the +2 makes the 16 tiles of a row block hit 16
// different banks) + 18 KB exchange = 62 KB (TSY = 16) / 42 KB (TSY = 8) at Cin = 32.
// Where the time goes at cfg2's L5 (8 x 128 x 128, 32 -> 32, 16.6 us against 23.4 direct): the MFMA floor
// of this form is 6.8 us per CU (2048 MFMAs over 4 SIMDs), the k-loop measures 11 us (tools/wino_variants.py:
// ablation builds), the rest is per-workgroup fixed cost — filter load 1.5, first rows, exchange,
// stores — that the two workgroups of a CU pay in lockstep because every tile starts at once.
#include "ra_common.h"

#ifndef RA_PAIRW_SPLIT_OCC
#define RA_PAIRW_SPLIT_OCC 3  // workgroups per CU of the fused L2+L3 pair's SPLIT form
#endif
namespace ra {
namespace wino {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TS = 16;       // output tile width (and height of the tall form)
constexpr int WS = TS + 2;   // staged window width
constexpr int TEX = 36;      // exchange stride per tile (floats): 16 * ksub banks apart

struct WArgs {
  const float *x, *wp, *scale, *shift;
  float *y;
  int B, H, W, Cout, relu;
  int bytes_x;
  int xcd_map;
  unsigned *tickets;  // this launch's slots of tile-ticket pools, one per channel slice (ra_common.h); nullptr = the static walk
};

// XCD-contiguous tile walk (see conv_pair8_mfma): workgroups are dealt to the 8 XCDs round robin, so with
// gridDim.x % 8 == 0 XCD x = blockIdx.x % 8 walks the tiles [x * chunk, (x + 1) * chunk) with its gridDim.x / 8
// workgroups and neighbouring tiles' halos meet in one L2.  first / end / step of this workgroup's walk.
struct TileWalk {
  int first, end, step;
};
__device__ inline TileWalk tile_walk(int ntiles, int xcd_map) {
  TileWalk w;
  if (!xcd_map) {
    w.first = blockIdx.x;
    w.end = ntiles;
    w.step = gridDim.x;
    return w;
  }
  const int chunk = (ntiles + 7) >> 3, x = blockIdx.x & 7;
  w.first = x * chunk + ((int)blockIdx.x >> 3);
  w.end = (x * chunk + chunk < ntiles) ? x * chunk + chunk : ntiles;
  w.step = (int)gridDim.x >> 3;
  return w;
}

// TSY: output tile height, 16 (4 row blocks) or 8 (2 row blocks: more, smaller workgroups for the layers whose
// 16x16 tiles would not give every CU two workgroups)
// NB: blocks of 16 output channels per workgroup (2, or 1 for layers with 16 output channels)
template <int CIN, int POOL, int TSY, int NB>
__global__ __launch_bounds__(256, 2) void conv_wino_mfma(const WArgs a, int tiles_x, int tiles_y, int ntiles) {
  constexpr int KK = CIN / 4, S = CIN + 2, C4 = CIN / 4, NMB = TSY / 4, WSY = TSY + 2, CO = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tin = lds;                  // [WSY][WS][S]
  float *tex = lds + WSY * WS * S;   // [4 p][2 j][16 tiles][TEX]
  const int tid = threadIdx.x, lane = tid & 63;
  // dynamic tile tickets (a.tickets): tiles are drawn from this XCD's pool of the channel slice instead of walked
  __shared__ unsigned tk_sh[2];
  TicketWalk tk;
  const bool dyn = a.tickets != nullptr;
  if (dyn) tk.issue(a.tickets + blockIdx.y * kTicketSlotWords, ntiles);
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, ksub = lane >> 4;
  const int slice = blockIdx.y, NBT = a.Cout / 16;

  // this wave's transformed filters, q = 0..3, all k-steps, 2 blocks of 16 output channels
  float bw[4][KK][NB];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        bw[q][kk][nb] = a.wp[((size_t)((p * 4 + q) * KK + kk) * NBT + NB * slice + nb) * 64 + lane];

  // rows of the 4x4 patch that row p of B^T combines: r_j = d[ra][j] + sg * d[rb][j]
  const int ra = (p == 0) ? 0 : (p == 2) ? 2 : 1;
  const int rb = (p == 0) ? 2 : (p == 1) ? 2 : (p == 2) ? 1 : 3;
  const float sg = (p == 1) ? 1.f : -1.f;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, a.bytes_x, 0x00020000);
  const int per = tiles_x * tiles_y;
  // epilogue constants of this thread's (tile, cout) pairs of a row block: cout = tid % CO
  const int eco = tid % CO;
  const float sc = a.scale[CO * slice + eco], sh = a.shift[CO * slice + eco];
  const float lo = a.relu ? 0.f : -__builtin_inff();
  const int Ho = a.H / POOL, Wo = a.W / POOL;

  // Staging is cut by row blocks: block k needs window rows 4k .. 4k + 5.  Rows 0..5 are staged up front,
  // the 4 new rows of block k + 1 are loaded into registers before block k's MFMAs and written to LDS after
  // them (nobody reads those rows yet; the exchange barrier publishes them), so only the first 6 rows of a
  // tile are exposed.
  constexpr int N0 = (6 * WS * C4 + 255) / 256, N1 = (4 * WS * C4 + 255) / 256;
  auto load_rows = [&](int b, int oy, int ox, int row0, int nrows, int i) -> f32x4 {  // item i of this thread
    const int e = tid + 256 * i;
    const int c4 = e % C4, pix = e / C4;
    const int r = row0 + pix / WS, c = pix % WS;
    const int Y = oy + r, X = ox + c;
    const bool ok = (e < nrows * WS * C4) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
    const int off = ok ? (((b * a.H + Y) * a.W + X) * CIN + 4 * c4) * 4 : 0x7fffffff;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
  };
  auto store_rows = [&](int row0, int nrows, int i, f32x4 v) {
    const int e = tid + 256 * i;
    if (e < nrows * WS * C4) {
      const int c4 = e % C4, pix = e / C4;
      float *d = tin + (row0 * WS + pix) * S + 4 * c4;
      *reinterpret_cast<f32x2 *>(d) = f32x2{v.x, v.y};
      *reinterpret_cast<f32x2 *>(d + 2) = f32x2{v.z, v.w};
    }
  };

  const TileWalk tw = tile_walk(ntiles, a.xcd_map);
  if (dyn) tk.begin(tk_sh);
  const int t_end = dyn ? ntiles : tw.end;  // drawn tiles come from the pool of the XCD the workgroup is ON (not blockIdx % 8's chunk)
  for (int tile = dyn ? (tk.cur >= 0 ? tk.cur : t_end) : tw.first; tile < t_end;
       tile = dyn ? (tk.step(), tk.cur >= 0 ? tk.cur : t_end) : tile + tw.step) {
    const int b = tile / per, trem = tile - b * per;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy = ty * TSY - 1, ox = tx * TS - 1;
    __syncthreads();  // the previous tile's reads of tin / tex are complete
    {
      f32x4 v0[N0];
#pragma unroll
      for (int i = 0; i < N0; ++i) v0[i] = load_rows(b, oy, ox, 0, 6, i);
#pragma unroll
      for (int i = 0; i < N0; ++i) store_rows(0, 6, i, v0[i]);
    }
    if (dyn) tk.publish(tk_sh);
    __syncthreads();
    if (dyn) {
      tk.read_next(tk_sh);
      tk.request();
    }

#pragma unroll 1
    for (int mblk = 0; mblk < NMB; ++mblk) {
      f32x4 vn[N1];
      if (mblk + 1 < NMB) {
#pragma unroll
        for (int i = 0; i < N1; ++i) vn[i] = load_rows(b, oy, ox, 4 * mblk + 6, 4, i);
      }
      // lane m's Winograd tile of this row block and its patch origin in the staged window
      const int tyi = 2 * mblk + (m >> 3), txi = m & 7;
      const float *pa = tin + ((2 * tyi + ra) * WS + 2 * txi) * S + ksub;
      const float *pb = tin + ((2 * tyi + rb) * WS + 2 * txi) * S + ksub;
      f32x4 acc[4][NB];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[q][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the patch values of k-step kk + 1 are read while the MFMAs of k-step kk issue
      float da[2][4], db[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        da[0][j] = pa[j * S];
        db[0][j] = pb[j * S];
      }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        if (kk + 1 < KK) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            da[(kk + 1) & 1][j] = pa[j * S + 4 * (kk + 1)];
            db[(kk + 1) & 1][j] = pb[j * S + 4 * (kk + 1)];
          }
        }
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = da[kk & 1][j] + sg * db[kk & 1][j];
        const float v[4] = {r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[q][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[q][kk][nb], v[q], acc[q][nb], 0, 0, 0);  // D^T: rows = couts, columns = tiles
      }
      if (mblk + 1 < NMB) {
#pragma unroll
        for (int i = 0; i < N1; ++i) store_rows(4 * mblk + 6, 4, i, vn[i]);
      }
      // q half of A^T: T[p][0] = M0 + M1 + M2, T[p][1] = M1 - M2 - M3.  The MFMAs ran with the operands swapped (filter = A
      // operand), so a lane holds couts 4 ksub .. 4 ksub + 3 of tile m: one 16-byte store per half instead of four 4-byte ones
      if (mblk) __syncthreads();  // the previous row block's exchange has been consumed
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 t0 = acc[0][nb] + acc[1][nb] + acc[2][nb];
        const f32x4 t1 = acc[1][nb] - acc[2][nb] - acc[3][nb];
        *reinterpret_cast<f32x4 *>(&tex[((p * 2 + 0) * 16 + m) * TEX + 16 * nb + 4 * ksub]) = t0;
        *reinterpret_cast<f32x4 *>(&tex[((p * 2 + 1) * 16 + m) * TEX + 16 * nb + 4 * ksub]) = t1;
      }
      __syncthreads();
      // p half of A^T + BN + ReLU + pool for the 16 x CO (tile, cout) pairs of the row block, NB per thread
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int tl = tid / CO + (256 / CO) * k;  // tile 0..15 of the row block
        float T[4][2];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
          for (int j = 0; j < 2; ++j) T[pp][j] = tex[((pp * 2 + j) * 16 + tl) * TEX + eco];
        float yv[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          yv[0][j] = fmaxf((T[0][j] + T[1][j] + T[2][j]) * sc + sh, lo);
          yv[1][j] = fmaxf((T[1][j] - T[2][j] - T[3][j]) * sc + sh, lo);
        }
        const int oty = ty * (TSY / 2) + 2 * mblk + (tl >> 3), otx = tx * 8 + (tl & 7);  // Winograd tile coordinates in the image
        if constexpr (POOL == 2) {
          const float best = fmaxf(fmaxf(yv[0][0], yv[0][1]), fmaxf(yv[1][0], yv[1][1]));
          a.y[((size_t)(b * Ho + oty) * Wo + otx) * a.Cout + CO * slice + eco] = best;
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              a.y[((size_t)(b * Ho + 2 * oty + i) * Wo + 2 * otx + j) * a.Cout + CO * slice + eco] = yv[i][j];
        }
      }
    }
  }
}

inline int cu_count();

template <int CIN, int POOL, int TSY, int NB>
int launch(const WArgs &a, hipStream_t st) {
  auto kern = conv_wino_mfma<CIN, POOL, TSY, NB>;
  constexpr size_t lds = (size_t)((TSY + 2) * WS * (CIN + 2) + 8 * 16 * TEX) * sizeof(float);
  static bool attr = false;
  static int cap = 0;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    cap = nb * cu_count();
    attr = true;
  }
  const int tiles_x = a.W / TS, tiles_y = a.H / TSY, ntiles = tiles_x * tiles_y * a.B, slices = a.Cout / (16 * NB);
  int gx = cap / slices;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  WArgs a2 = a;  // XCD-contiguous tile walk when the grid's rows are whole rounds of the 8 XCDs
  a2.xcd_map = (gx % 8 == 0) ? 1 : 0;
  a2.tickets = ntiles >= kTicketMinTilesPerWg * gx ? take_ticket_slots(slices, gx) : nullptr;  // several tiles per workgroup and a bound scratch: drawn tiles
  hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(256), lds, st, a2, tiles_x, tiles_y, ntiles);
  return launch_status("ra_conv_wino_f32");
}

inline int cu_count() {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  }
  return cus;
}

// 16-row tiles only when they give every CU more than two workgroups (RA_WINO_TSY=8|16 forces one form):
// measured at cfg2, 8-row tiles 9.8 vs 11.5 us (L6), 12.7 vs 13.3 (L4), 16.5 vs 16.6 (L5)
template <int CIN, int POOL, int NB>
int launch_any(const WArgs &a, hipStream_t st) {
  static int force = -1;
  if (force < 0) {
    const char *e = getenv("RA_WINO_TSY");
    force = e ? atoi(e) : 0;
  }
  const int tall = (a.W / TS) * (a.H / 16) * a.B * (a.Cout / (16 * NB));
  const bool small = force ? force == 8 : tall <= 2 * cu_count();
  return small ? launch<CIN, POOL, 8, NB>(a, st) : launch<CIN, POOL, 16, NB>(a, st);
}

// -------------------------------------------------------------------------------------------------
// K1pw — a fused layer pair whose second layer is Winograd: the controller CNN's L2+L3 (8 -> 16 -> 16, pool 2).
// Layer A (direct 3x3 on the MFMA, as in conv_pair_persist_mfma) is computed on the output tile + 1-pixel
// halo straight into the LDS window layer B's Winograd reads — [window pixel][16 + 2 floats], zero outside
// the image (layer B's SAME padding) — so the intermediate never leaves the CU and layer B (2/3 of the
// pair's FLOPs) costs 2.25x fewer MFMAs.  The A region is walked as a LINEAR list of its 18 x (TSY + 2)
// pixels in groups of 16 (1.31x / 1.41x the tile's pixels at TSY = 16 / 8; the 8-pixel-group walk of the
// direct pair kernel pads 32 x 8 tiles to 1.56x).  Persistent; the next tile's input window is fetched
// into registers behind both phases; the Winograd exchange buffer aliases the input tile, which is dead
// by then.
struct PWArgs {
  const float *x, *wpA, *scA, *shA, *wpB, *scB, *shB;
  float *y;
  int B, H, W, CoutAP, reluA, reluB;
  int bytes_x;
  int xcd_map;
  unsigned *tickets;  // this launch's slot of tile-ticket pools (ra_common.h); nullptr = the static walk
};

// SPLIT (round 5): layer A — the direct 8 -> 16 conv, 54 of the kernel's 86 MFMAs per wave and tile — on the BF16 matrix pipe at
// float32 accuracy, as in conv_pair8_mfma's SPLIT form (ra_conv_pair.hip): the staged input window is kept as three bf16
// tiles [pixel][8 channels] (the exact three-piece split of every float32 value, made once per staged element), a K = 32
// block is four taps x 8 channels (three blocks for the nine taps), a lane's A operand of a block and piece is one
// ds_read_b128, and six piece products per block replace eight float32 MFMAs.
typedef short s16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
__device__ inline unsigned pk_bf16w(float lo, float hi) {
  typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2c));
}
__device__ inline void split3_pairw(float a, float b, unsigned &H, unsigned &M, unsigned &L) {  // a = a_H + a_M + a_L exactly
  H = pk_bf16w(a, b);
  float ra = a - __builtin_bit_cast(float, H << 16), rb = b - __builtin_bit_cast(float, H & 0xffff0000u);
  M = pk_bf16w(ra, rb);
  ra -= __builtin_bit_cast(float, M << 16);
  rb -= __builtin_bit_cast(float, M & 0xffff0000u);
  L = pk_bf16w(ra, rb);
}
// tools/pairw_probe.hip builds this file with -DRA_PROBEW: wave 0 of every workgroup accumulates the shader-clock time between a
// few points of conv_pair_wino_mfma's tile loop and leaves the sums in ra_probew_buf[workgroup][8] (as RA_PROBE8 in ra_conv_pair.hip)
#ifdef RA_PROBEW
__device__ long long *ra_probew_buf;
#define RA_PW_DECL long long pw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pw_t = (long long)__builtin_readcyclecounter(), pw_t0 = (long long)wall_clock64()
#define RA_PW_AT(k)                                                \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const long long n_ = (long long)__builtin_readcyclecounter();  \
    pw_acc[k] += n_ - pw_t;                                        \
    pw_t = n_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#define RA_PW_END                                                                    \
  do {                                                                               \
    if (threadIdx.x == 0 && ra_probew_buf) {                                         \
      pw_acc[7] = (long long)wall_clock64() - pw_t0;                                 \
      for (int k_ = 0; k_ < 8; ++k_) ra_probew_buf[(size_t)blockIdx.x * 8 + k_] = pw_acc[k_]; \
    }                                                                                \
  } while (0)
#else
#define RA_PW_DECL
#define RA_PW_AT(k)
#define RA_PW_END
#endif
template <int TSY, bool SPLIT = false>
__global__ __launch_bounds__(256, TSY == 8 ? (SPLIT ? RA_PAIRW_SPLIT_OCC : 4) : 2) void conv_pair_wino_mfma(const PWArgs a, int tiles_x, int tiles_y, int ntiles) {
  constexpr int CINA = 8, CMID = 16, KK = CMID / 4, S = CMID + 2, NMB = TSY / 4;
  constexpr int AWY = TSY + 2, IWY = TSY + 4, IWX = TS + 4;   // layer-A output window / input window (rows; 18 / 20 wide)
  constexpr int NPA = AWY * WS, NGA = (NPA + 15) / 16, GPW = (NGA + 3) / 4;
  constexpr int NPI = IWY * IWX, NIT = (NPI * 2 + 255) / 256;  // input items: (pixel, half of its 8 channels)
  constexpr int TEXP = 20;                   // exchange stride per tile for 16 output channels: 16 * ksub banks apart
  constexpr int PLB = NPI * 16;              // SPLIT: bytes of one bf16 input tile [pixel][8]
  constexpr int INF = SPLIT ? 3 * PLB / 4 : NPI * CINA;  // floats of the staged input
  constexpr int R0 = INF > 8 * 16 * TEXP ? INF : 8 * 16 * TEXP;
  // dynamic tile tickets (a.tickets): tiles are drawn from this XCD's pool instead of walked (static: tile += tw.step)
  __shared__ unsigned tk_sh[2];
  TicketWalk tk;
  const bool dyn = a.tickets != nullptr;
  if (dyn) tk.issue(a.tickets, ntiles);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *tinp = lds;                         // [IWY][IWX][8]   records [ksub][cg]  (channel = 4 * cg + ksub)
  float *tex = lds;                          // [4 p][2 j][16 tiles][TEXP]: phase B only, when tinp is dead
  float *tin = lds + R0;                     // [AWY * 18][S]   layer-A output window (+ one group of slack:
                                             // the padding rows of the last group land there)
  const int tid = threadIdx.x, lane = tid & 63;
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, ksub = lane >> 4;
  const int per = tiles_x * tiles_y;

  // layer A: direct-form B operands (9 taps x 2 channel groups) and epilogue constants of column m
  float bA[SPLIT ? 1 : 9][2];
  s16x8w wA[SPLIT ? 3 : 1][SPLIT ? 3 : 1];  // SPLIT: block blk, k-slot j = input channel j of tap 4 blk + ksub, column m, three pieces
  if constexpr (SPLIT) {
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
      const int tap = 4 * blk + ksub;
      const bool ok = tap < 9;
      const int tp = ok ? tap : 0;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float w0 = a.wpA[(size_t)((tp * 2 + (j >> 2)) * 4 + (j & 3)) * a.CoutAP + m];
        const float w1 = a.wpA[(size_t)((tp * 2 + ((j + 1) >> 2)) * 4 + ((j + 1) & 3)) * a.CoutAP + m];
        unsigned H, M, L;
        split3_pairw(ok ? w0 : 0.f, ok ? w1 : 0.f, H, M, L);
        wA[blk][0][j] = (short)(H & 0xffffu), wA[blk][0][j + 1] = (short)(H >> 16);
        wA[blk][1][j] = (short)(M & 0xffffu), wA[blk][1][j + 1] = (short)(M >> 16);
        wA[blk][2][j] = (short)(L & 0xffffu), wA[blk][2][j + 1] = (short)(L >> 16);
      }
    }
  } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) bA[tap][cg] = a.wpA[(size_t)((tap * 2 + cg) * 4 + ksub) * a.CoutAP + m];
  }
  // Layer A's MFMAs run with the operands swapped (filter = A operand, window pixels = B operand: the same lane contents, the
  // other argument order), so a lane's accumulator holds channels 4 ksub .. 4 ksub + 3 of ONE window pixel (16 g + m) instead of
  // one channel of four pixels: two 8-byte LDS stores and one bounds check per group instead of four 4-byte stores and four checks
  f32x4 scA4, shA4;
#pragma unroll
  for (int j = 0; j < 4; ++j) scA4[j] = a.scA[4 * ksub + j], shA4[j] = a.shA[4 * ksub + j];
  const float loA = a.reluA ? 0.f : -__builtin_inff();
  // layer B: this wave's row p of the transformed filters (one block of 16 output channels)
  float bw[4][KK];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) bw[q][kk] = a.wpB[((size_t)((p * 4 + q) * KK + kk)) * 64 + lane];
  const int ra = (p == 0) ? 0 : (p == 2) ? 2 : 1;
  const int rb = (p == 0) ? 2 : (p == 1) ? 2 : (p == 2) ? 1 : 3;
  const float sg = (p == 1) ? 1.f : -1.f;
  const int eco = tid & 15;
  const float scB = a.scB[eco], shB = a.shB[eco];
  const float loB = a.reluB ? 0.f : -__builtin_inff();
  const int Ho = a.H / 2, Wo = a.W / 2;

  // phase A bookkeeping: LDS offset of this lane's pixel of each of the wave's groups (tile-invariant)
  int ain[GPW];
#pragma unroll
  for (int s = 0; s < GPW; ++s) {
    int li = 16 * (p + 4 * s) + m;
    if (li >= NPA) li = NPA - 1;  // padding rows of the last group repeat a real pixel; their results go to the slack
    const int r = li / WS, c = li - r * WS;
    ain[s] = SPLIT ? (r * IWX + c) * 16 : (r * IWX + c) * CINA + 2 * ksub;  // SPLIT: byte offset of the pixel's 16-byte record
  }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, a.bytes_x, 0x00020000);
  f32x4 pre[NIT];
  auto fetch = [&](int T) {
    const int fb = T / per, fr = T - fb * per;
    const int fy0 = (fr / tiles_x) * TSY - 2, fx0 = (fr % tiles_x) * TS - 2;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = tid + 256 * i, cg = e & 1, pix = e >> 1;
      const int r = pix / IWX, c = pix - r * IWX;
      const int Y = fy0 + r, X = fx0 + c;
      const bool ok = (e < NPI * 2) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
      const int off = ok ? (((fb * a.H + Y) * a.W + X) * CINA + 4 * cg) * 4 : 0x7fffffff;
      pre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    }
  };

  const TileWalk tw = tile_walk(ntiles, a.xcd_map);
  int tile = tw.first, tnext = 0;
  const int t_end = dyn ? ntiles : tw.end;  // drawn tiles come from the pool of the XCD the workgroup is ON (not blockIdx % 8's chunk)
  if (dyn) {
    tk.begin(tk_sh);
    tile = tk.cur >= 0 ? tk.cur : t_end;
  }
  if (tile < t_end) fetch(tile);
  RA_PW_DECL;
  for (; tile < t_end; tile = tnext) {
    const int b = tile / per, trem = tile - b * per;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    __syncthreads();  // the previous tile's phase B (exchange reads) is complete
    RA_PW_AT(0);  // top barrier
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = tid + 256 * i, cg = e & 1, pix = e >> 1;
      if constexpr (SPLIT) {
        unsigned H0, M0, L0, H1, M1, L1;  // channels 4 cg .. 4 cg + 3 of the pixel: 8 bytes of its record in each of the three tiles
        split3_pairw(pre[i].x, pre[i].y, H0, M0, L0);
        split3_pairw(pre[i].z, pre[i].w, H1, M1, L1);
        if (e < NPI * 2) {
          unsigned char *rec = reinterpret_cast<unsigned char *>(tinp) + pix * 16 + cg * 8;
          *reinterpret_cast<u32x2w *>(rec) = u32x2w{H0, H1};
          *reinterpret_cast<u32x2w *>(rec + PLB) = u32x2w{M0, M1};
          *reinterpret_cast<u32x2w *>(rec + 2 * PLB) = u32x2w{L0, L1};
        }
        continue;
      }
      if (e < NPI * 2) {
        float *rec = tinp + pix * CINA + cg;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) rec[2 * ks] = pre[i][ks];
      }
    }
    if (dyn) tk.publish(tk_sh);
    __syncthreads();
    RA_PW_AT(1);  // staged (split into three bf16 tiles) + barrier
    if (dyn) {
      tk.read_next(tk_sh);
      tk.request();  // older than the prefetch loads below: consumed with them at the next tile's staging
      tk.step();
      tnext = tk.cur >= 0 ? tk.cur : t_end;
    } else {
      tnext = tile + tw.step;
    }
    if (tnext < t_end) fetch(tnext);

    // ---------------- phase A: layer A on the window, BN + ReLU, -> tin ----------------
    {
      const int oyA = ty * TSY - 1, oxA = tx * TS - 1;  // image coordinates of window pixel (0, 0)
      const bool interior = (oyA >= 0) & (oyA + AWY <= a.H) & (oxA >= 0) & (oxA + WS <= a.W);
      f32x4 acc[GPW];
#pragma unroll
      for (int s = 0; s < GPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT) {
        const unsigned char *tb = reinterpret_cast<const unsigned char *>(tinp);
        const int t0 = ksub, t1 = 4 + ksub, t2 = 8;  // this lane's tap of the three blocks (block 2: tap 8; its other k-slots carry zero weights)
        const int toff[3] = {((t0 / 3) * IWX + t0 % 3) * 16, ((t1 / 3) * IWX + t1 % 3) * 16, ((t2 / 3) * IWX + t2 % 3) * 16};
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};  // six piece products, smallest first
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
          for (int s = 0; s < GPW; ++s) {
            s16x8w av[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) av[pc] = *reinterpret_cast<const s16x8w *>(tb + ain[s] + toff[blk] + pc * PLB);
#pragma unroll
            for (int t = 0; t < 6; ++t)
              acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, wA[blk][PB[t]]), __builtin_bit_cast(bf16x8w, av[PA[t]]),
                                                               acc[s], 0, 0, 0);
          }
      } else {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        f32x2 av[GPW];
#pragma unroll
        for (int s = 0; s < GPW; ++s)
          av[s] = *reinterpret_cast<const f32x2 *>(&tinp[ain[s] + ((tap / 3) * IWX + tap % 3) * CINA]);
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
          for (int s = 0; s < GPW; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(bA[tap][cg], av[s][cg], acc[s], 0, 0, 0);
      }
      }
      RA_PW_AT(2);  // layer A's MFMAs
#pragma unroll
      for (int s = 0; s < GPW; ++s) {
        const int g = p + 4 * s;
        if (g < NGA) {  // wave-uniform
          const int li = 16 * g + m;  // D^T column m of the group = window pixel li; rows 4 ksub + j = its channels
          f32x4 o = acc[s] * scA4 + shA4;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], loA);
          if (!interior) {
            const int wr = li / WS, wc = li - wr * WS;
            const int Y = oyA + wr, X = oxA + wc;
            const bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ok ? o[j] : 0.f;
          }
          float *d = tin + li * S + 4 * ksub;  // li >= NPA: the slack behind the window; S = 18: 8-byte aligned
          *reinterpret_cast<f32x2 *>(d) = f32x2{o[0], o[1]};
          *reinterpret_cast<f32x2 *>(d + 2) = f32x2{o[2], o[3]};
        }
      }
    }
    RA_PW_AT(3);  // layer A's epilogue -> tin
    __syncthreads();
    RA_PW_AT(4);  // barrier

    // ---------------- phase B: Winograd F(2x2, 3x3) out of tin (see conv_wino_mfma) ----------------
#pragma unroll 1
    for (int mblk = 0; mblk < NMB; ++mblk) {
      const int tyi = 2 * mblk + (m >> 3), txi = m & 7;
      const float *pa = tin + ((2 * tyi + ra) * WS + 2 * txi) * S + ksub;
      const float *pb = tin + ((2 * tyi + rb) * WS + 2 * txi) * S + ksub;
      f32x4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = pa[j * S + 4 * kk] + sg * pb[j * S + 4 * kk];
        const float v[4] = {r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[q][kk], v[q], acc[q], 0, 0, 0);  // D^T (see conv_wino_mfma)
      }
      RA_PW_AT(5);  // phase B: input transform + 16 MFMAs of the row block
      if (mblk) __syncthreads();  // the previous row block's exchange has been consumed
      {
        const f32x4 t0 = acc[0] + acc[1] + acc[2];
        const f32x4 t1 = acc[1] - acc[2] - acc[3];
        *reinterpret_cast<f32x4 *>(&tex[((p * 2 + 0) * 16 + m) * TEXP + 4 * ksub]) = t0;  // couts 4 ksub .. + 3 of tile m
        *reinterpret_cast<f32x4 *>(&tex[((p * 2 + 1) * 16 + m) * TEXP + 4 * ksub]) = t1;
      }
      __syncthreads();
      {
        const int tl = tid >> 4;  // tile 0..15 of the row block, cout = tid % 16
        float T[4][2];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
          for (int j = 0; j < 2; ++j) T[pp][j] = tex[((pp * 2 + j) * 16 + tl) * TEXP + eco];
        float best = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          best = fmaxf(best, fmaxf((T[0][j] + T[1][j] + T[2][j]) * scB + shB, loB));
          best = fmaxf(best, fmaxf((T[1][j] - T[2][j] - T[3][j]) * scB + shB, loB));
        }
        const int oty = ty * (TSY / 2) + 2 * mblk + (tl >> 3), otx = tx * 8 + (tl & 7);
        a.y[((size_t)(b * Ho + oty) * Wo + otx) * 16 + eco] = best;
      }
      RA_PW_AT(6);  // phase B: exchange, output transform, BN + ReLU + pool, store
    }
  }
  RA_PW_END;
}

template <int TSY, bool SPLIT = false>
int launch_pair(const PWArgs &a, hipStream_t st) {
  auto kern = conv_pair_wino_mfma<TSY, SPLIT>;
  constexpr int inf = SPLIT ? 3 * (TSY + 4) * (TS + 4) * 4 : (TSY + 4) * (TS + 4) * 8;
  constexpr int r0 = inf > 8 * 16 * 20 ? inf : 8 * 16 * 20;
  constexpr size_t lds = (size_t)(r0 + ((TSY + 2) * WS + 16) * 18) * sizeof(float);
  static bool attr = false;
  static int cap = 0;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    cap = nb * cu_count();
    const char *e = getenv("RA_PAIRW_WGS");
    if (e && atoi(e) > 0) cap = atoi(e);
    attr = true;
  }
  const int tiles_x = a.W / TS, tiles_y = a.H / TSY, ntiles = tiles_x * tiles_y * a.B;
  static int xcd = -1;  // RA_PAIRW_XCD=0: tuning aid, the interleaved tile walk
  if (xcd < 0) {
    const char *e = getenv("RA_PAIRW_XCD");
    xcd = e ? atoi(e) : 1;
  }
  const int grid = ntiles < cap ? ntiles : cap;
  PWArgs a2 = a;
  a2.xcd_map = (xcd && grid % 8 == 0 && grid >= 8) ? 1 : 0;
  // (this pair from 6 tiles per workgroup: at 4 — cfg2's batch of 8 alone — drawing costs it 2 us of 33, at 8 — the 16 images of a
  // pipeline slot — 1.3 of 66 against 9-14 us saved next to another slot's tail)
  a2.tickets = ntiles >= 2 * kTicketMinTilesPerWg * grid ? take_ticket_slots(1, grid) : nullptr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a2, tiles_x, tiles_y, ntiles);
  return launch_status("ra_conv_pair_wino_f32");
}

}  // namespace wino
}  // namespace ra

using namespace ra;

extern "C" int ra_conv_wino_supported(int Cin, int Cout, int pool, int H, int W) {
  return (Cin == 16 || Cin == 32) && Cout > 0 && Cout % 16 == 0 && (pool == 1 || pool == 2) && H > 0 && W > 0 &&
         H % wino::TS == 0 && W % wino::TS == 0;
}

extern "C" size_t ra_conv_wino_packed_floats(int Cin, int Cout) {
  if (!(Cin == 16 || Cin == 32) || Cout <= 0 || Cout % 16) return 0;
  return (size_t)16 * Cin * Cout;
}

// w: the reference's [3,3,Cin,Cout] filter (host).  out[(((p*4+q)*KK + kk)*NBT + nb)*64 + lane] =
// (G g G^T)[p][q] of input channel 4*kk + lane/16 and output channel 16*nb + lane%16: the B operand of
// one MFMA, one coalesced 256-byte load per wave.
extern "C" int ra_conv_wino_pack_weights(const float *w, int Cin, int Cout, float *out) {
  if (!w || !out || !ra_conv_wino_packed_floats(Cin, Cout)) return fail(RA_E_SHAPE, "ra_conv_wino_pack_weights: Cin %d Cout %d", Cin, Cout);
  static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const int KK = Cin / 4, NBT = Cout / 16;
  for (int p = 0; p < 4; ++p)
    for (int q = 0; q < 4; ++q)
      for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co) {
          double u = 0.0;
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) u += (double)G[p][ky] * (double)G[q][kx] * (double)w[((size_t)(ky * 3 + kx) * Cin + ci) * Cout + co];
          const int kk = ci / 4, ks = ci % 4, nb = co / 16, n = co % 16;
          out[((size_t)((p * 4 + q) * KK + kk) * NBT + nb) * 64 + ks * 16 + n] = (float)u;
        }
  return 0;
}

extern "C" int ra_conv_wino_f32(const float *x, int B, int H, int W, int Cin, const float *wpacked, const float *scale,
                                const float *shift, int Cout, int relu, int pool, float *y, void *stream) {
  if (!x || !wpacked || !scale || !shift || !y || B <= 0) return fail(RA_E_INVALID, "ra_conv_wino_f32: bad argument");
  if (!ra_conv_wino_supported(Cin, Cout, pool, H, W))
    return fail(RA_E_SHAPE, "ra_conv_wino_f32: Cin=%d Cout=%d pool=%d %dx%d", Cin, Cout, pool, H, W);
  const size_t bytes = (size_t)B * H * W * Cin * sizeof(float);
  if (bytes >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_conv_wino_f32: input exceeds 2 GiB");
  wino::WArgs a{};
  a.x = x;
  a.wp = wpacked;
  a.scale = scale;
  a.shift = shift;
  a.y = y;
  a.B = B;
  a.H = H;
  a.W = W;
  a.Cout = Cout;
  a.relu = relu;
  a.bytes_x = (int)bytes;
  a.xcd_map = 0;
  hipStream_t st = as_stream(stream);
  if (Cout % 32) {  // 16 output channels per workgroup
    if (Cin == 16) return pool == 2 ? wino::launch_any<16, 2, 1>(a, st) : wino::launch_any<16, 1, 1>(a, st);
    return pool == 2 ? wino::launch_any<32, 2, 1>(a, st) : wino::launch_any<32, 1, 1>(a, st);
  }
  if (Cin == 16) return pool == 2 ? wino::launch_any<16, 2, 2>(a, st) : wino::launch_any<16, 1, 2>(a, st);
  return pool == 2 ? wino::launch_any<32, 2, 2>(a, st) : wino::launch_any<32, 1, 2>(a, st);
}

extern "C" int ra_conv_pair_wino_supported(int Cin, int CoutA, int CoutB, int poolB, int H, int W) {
  return Cin == 8 && CoutA == 16 && CoutB == 16 && poolB == 2 && H > 0 && W > 0 && H % wino::TS == 0 && W % wino::TS == 0;
}

extern "C" int ra_conv_pair_wino_f32(const float *x, int B, int H, int W, const float *wpA, const float *scaleA,
                                     const float *shiftA, int reluA, const float *wpB_wino, const float *scaleB,
                                     const float *shiftB, int reluB, float *y, void *stream) {
  if (!x || !wpA || !scaleA || !shiftA || !wpB_wino || !scaleB || !shiftB || !y || B <= 0)
    return fail(RA_E_INVALID, "ra_conv_pair_wino_f32: bad argument");
  if (!ra_conv_pair_wino_supported(8, 16, 16, 2, H, W)) return fail(RA_E_SHAPE, "ra_conv_pair_wino_f32: %dx%d", H, W);
  const size_t bytes = (size_t)B * H * W * 8 * sizeof(float);
  if (bytes >= (1ull << 31)) return fail(RA_E_SHAPE, "ra_conv_pair_wino_f32: input exceeds 2 GiB");
  wino::PWArgs a{};
  a.x = x;
  a.wpA = wpA;
  a.scA = scaleA;
  a.shA = shiftA;
  a.wpB = wpB_wino;
  a.scB = scaleB;
  a.shB = shiftB;
  a.y = y;
  a.B = B;
  a.H = H;
  a.W = W;
  a.CoutAP = ra_conv_cout_padded(16);
  a.reluA = reluA;
  a.reluB = reluB;
  a.bytes_x = (int)bytes;
  a.xcd_map = 0;
  // phase B through an LDS exchange (128 VGPRs, 4 workgroups per CU).  Two other forms were built and measured in
  // round 2 — row-block waves with the output transform in registers (244 VGPRs: 37.6 vs 39.0 us alone, but 48.9k vs
  // 49.8k instance-timesteps/s with four batches in flight) and role-split 8-wave workgroups (the same) — and removed
  // (git history: 58353f0, d1e62e4).
  static int tsy = 0;  // RA_PAIRW_TSY=16: tuning aid
  if (!tsy) {
    const char *e = getenv("RA_PAIRW_TSY");
    tsy = (e && atoi(e) == 16) ? 16 : 8;  // 8-row tiles: 128 VGPRs and 24 KB of LDS, 4 workgroups per CU (39.1 vs 42.5 us)
  }
  static int split = -1;  // RA_PAIRW_SPLIT=0: layer A on the float32 MFMA (rounds 2-4)
  if (split < 0) {
    const char *e = getenv("RA_PAIRW_SPLIT");
    split = e ? atoi(e) : 1;
  }
  if (split && tsy == 8) return wino::launch_pair<8, true>(a, as_stream(stream));
  return tsy == 8 ? wino::launch_pair<8>(a, as_stream(stream)) : wino::launch_pair<16>(a, as_stream(stream));
}
