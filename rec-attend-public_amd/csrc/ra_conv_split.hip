// K1s — direct 3x3 convolution on the BF16 matrix pipe at float32 accuracy (round 5), for the mid-resolution layers of the
// controller CNN (nnlib.cnn: conv3x3 SAME + b -> BatchNorm(eval, folded) -> ReLU -> max-pool, nnlib.py:229-253).
//
// Why: gfx950's bf16 MFMA does 16x the float32 MFMA's FLOPs per cycle, and a float32 value is EXACTLY the sum of three bf16
// pieces (8 + 8 + 8 mantissa bits).  Products of bf16 numbers are exact in float32, and of the nine piece products of a * b
// six carry everything above 2^-24 of it (hh, hm, mh, hl, lh, mm), so six v_mfma_f32_16x16x32_bf16 per K = 32 block compute
// what eight v_mfma_f32_16x16x4_f32 compute, in 41 ns of a SIMD instead of 108 (tools/mfma_split_probe.hip,
// profiles/r05_mfma_split_probe.txt; K = 576 dot products: 1.9e-7 of sum |a b| against 2.3e-7 for the float32 chain).  At that
// rate the DIRECT form's nine taps cost less matrix time than Winograd F(2x2,3x3)'s four products on the float32 pipe
// (9 / 2.67 = 3.4 against 4), without the transform adds, the LDS exchange and the two barriers per row block of K1w.
//
// Layout: a workgroup owns a 16 x 16 output tile (pre-pool) and 16 NB output channels (NB = 2; grid.y = cout slices).
//   * the input window (18 x 18 pixels x Cin) is split ONCE per element on its way into LDS: three bf16 tiles
//     [pixel][Cin], pixel stride 2 Cin + 16 bytes, row stride 20 pixels — with 4 x 4-pixel MFMA row groups the 16 lanes of a
//     ds_read_b128 then hit 16 different bank quads;
//   * the filter's three pieces, pre-split on the host in B-operand order, stay in LDS for the launch; a wave reads the
//     3 NB operand vectors of a K-block once and uses them for its four pixel groups;
//   * a K = 32 block is one tap x 32 channels (Cin = 16: two taps x 16); a lane's A operand of a block and piece is the 8
//     consecutive channels of its pixel: ONE ds_read_b128; 3 reads feed 6 NB MFMAs;
//   * MFMA rows m = 4 q + r = element r (dy, dx) of pooling window q of a 4 x 4-pixel group, so a lane's four accumulators
//     are one 2x2 pooling window: pool = three v_max (as K1's non-SWAP form);
//   * persistent workgroups; the next tile's window is fetched into registers behind the MFMA loop.
// Results differ from K1 / K1w by summation order and the dropped 2^-24 terms only (tests/test_kernels_gpu.py).
#include <cstring>

#include "ra_common.h"

namespace ra {
namespace csplit {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 16;                     // output tile width (conv resolution); its height TH is 16, or 8 at Cin = 64 (Geo)
constexpr int WSX = TW + 2;                // staged window width
constexpr int WSP = 20;                    // LDS row stride in pixels (bank spread, see above)
constexpr int tile_rows(int cin) { return (cin >= 64 || cin == 24) ? 8 : 16; }  // Cin = 64: a 10-row window (3 x 28.8 KB) beside 54 KB of filter; 24: two workgroups per CU (below)
// bytes per staged pixel record: 2 Cin + 16, an ODD number of 16-byte units (48, 80, 144), so that the 16 lanes of a ds_read_b128
// hit 16 different bank quads; Cin = 24 would give 64 (four-way conflicts) and takes 80
constexpr int rec_stride(int cin) { return cin == 24 ? 80 : 2 * cin + 16; }
// K = 32 blocks: the 9 Cin / 8 channel octets of a pixel's 3 x 3 neighbourhood in (tap, octet) order, four to a block — octet
// o = 4 blk + kb is channels 8 (o % (Cin / 8)) .. + 8 of tap o / (Cin / 8); the last block's spare octets carry zero weights.
// (Cin = 32: one tap per block; 64: half a tap; 16: two taps; 24, round 6: one and a third.)
constexpr int k_blocks(int cin) { return (9 * (cin / 8) + 3) / 4; }

struct SArgs {
  const float *x;
  const unsigned short *wp;
  const float *scale, *shift;
  float *y;
  int B, H, W, Cout, relu;
  int bytes_x, bytes_y;
  unsigned *tickets;  // this launch's slots of tile-ticket pools, one per channel slice (ra_common.h); nullptr = the static walk
  const float *plane;  // PL: a [B,H,W] plane standing in for channel plane_chan of x (the canvas)
  int plane_chan, bytes_p;
};

template <int CIN, int NB>  // NB: blocks of 16 output channels per workgroup (2; 1 where Cout is an odd multiple of 16)
struct Geo {
  static constexpr int TH = tile_rows(CIN), WSY = TH + 2;      // output tile height, staged window height
  static constexpr int GPW = TH / 4;                           // 4 x 4-pixel groups per wave (TH / 4 rows of 4 groups over 4 waves)
  static constexpr int NBLK = k_blocks(CIN);
  static constexpr int RS = rec_stride(CIN);                   // bytes per staged pixel record
  static constexpr int PLANE = WSY * WSP * RS;                 // bytes of one bf16 tile
  static constexpr int WBYTES = NBLK * 3 * NB * 1024;          // one cout slice of the packed filter: [blk][piece][nb][kb][n][8] bf16
  static constexpr int LDS = 3 * PLANE + WBYTES;
  // Workgroups per CU (round 6).  A layer with few input channels has a SHORT k-loop (Cin = 16, 16 couts: 120 MFMAs per wave and
  // tile, 0.8 us) — too short to cover the next window's trip from HBM, which is issued behind the staging barrier: with one
  // workgroup per CU the KITTI architectures' first layer (13 -> 16 channels at full resolution) ran at 4.5 us per tile, slower
  // than the float32-MFMA K1.  Where two workgroups' LDS fits a CU (Cin = 16 / 24 with 16 couts: 67 / 70 KB each) the grid is
  // two per CU and one's window fetch, split and barriers hide under the other's k-loop.
  static constexpr int OCC = 2 * (LDS + 64) <= 160 * 1024 ? 2 : 1;
};

__device__ inline unsigned pk_bf16(float lo, float hi) {
  typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2c));
}
// (a, b) -> packed bf16 pairs H, M, L with a = a_H + a_M + a_L exactly (each difference is exact in float32)
__device__ inline void split3_pair(float a, float b, unsigned &H, unsigned &M, unsigned &L) {
  H = pk_bf16(a, b);
  float ra = a - __builtin_bit_cast(float, H << 16), rb = b - __builtin_bit_cast(float, H & 0xffff0000u);
  M = pk_bf16(ra, rb);
  ra -= __builtin_bit_cast(float, M << 16);
  rb -= __builtin_bit_cast(float, M & 0xffff0000u);
  L = pk_bf16(ra, rb);
}

// tools/split_probe.hip builds this file with -DRA_PROBES: wave 0 of every workgroup accumulates the shader-clock time between
// points of the tile loop (as RA_PROBE8 in ra_conv_pair.hip) and leaves the sums in ra_probes_buf[workgroup][8]
#ifdef RA_PROBES
__device__ long long *ra_probes_buf;
#define RA_PS_DECL long long ps_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps_t = (long long)__builtin_readcyclecounter(), ps_t0 = (long long)wall_clock64()
#define RA_PS_AT(k)                                                \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const long long n_ = (long long)__builtin_readcyclecounter();  \
    ps_acc[k] += n_ - ps_t;                                        \
    ps_t = n_;                                                     \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#define RA_PS_END                                                                    \
  do {                                                                               \
    if (threadIdx.x == 0 && ra_probes_buf) {                                         \
      ps_acc[7] = (long long)wall_clock64() - ps_t0;                                 \
      for (int k_ = 0; k_ < 8; ++k_) ra_probes_buf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k_] = ps_acc[k_]; \
    }                                                                                \
  } while (0)
#else
#define RA_PS_DECL
#define RA_PS_AT(k)
#define RA_PS_END
#endif
template <int CIN, int POOL, int NB, bool PL = false>  // PL: channel a.plane_chan of x comes from the plane a.plane (the first layer's canvas)
__global__ __launch_bounds__(256, (Geo<CIN, NB>::OCC)) void conv_split_kernel(const SArgs a, int tiles_x, int tiles_y, int ntiles) {
  using G = Geo<CIN, NB>;
  constexpr int RS = G::RS, PLANE = G::PLANE, NBLK = G::NBLK, C4 = CIN / 4, TH = G::TH, WSY = G::WSY, GPW = G::GPW;
  constexpr int NITEMS = WSY * WSX * C4, NIT = (NITEMS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *tin = lds;               // three bf16 tiles [WSY][WSP][RS]
  unsigned char *wl = lds + 3 * PLANE;    // the slice's filter pieces
  const int tid = threadIdx.x, lane = tid & 63;
  RA_PS_DECL;
  // dynamic tile tickets (a.tickets): tiles are drawn from this XCD's pool of the channel slice instead of walked
  __shared__ unsigned tk_sh[2];
  TicketWalk tk;
  const bool dyn = a.tickets != nullptr;
  if (dyn) tk.issue(a.tickets + blockIdx.y * kTicketSlotWords, ntiles);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kb = lane >> 4;  // A side: row m, k-group kb;  B / D side: column n = m, D rows 4 kb + r
  const int slice = blockIdx.y;

  // A-operand byte offsets: the lane's pixel inside a 4 x 4 group (row m = 4 q + r: window q = (qy, qx), element r = (dy, dx))
  // and, per K-block, its tap and channel octet
  const int qy = m >> 3, qx = (m >> 2) & 1, dy = (m >> 1) & 1, dx = m & 1;
  const int lane_pix = ((2 * qy + dy) * WSP + 2 * qx + dx) * RS;
  int toff[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    constexpr int OPT = CIN / 8;
    const int o = 4 * b + kb;
    int tap = o / OPT;
    const int ch = 8 * (o % OPT);
    tap = tap < 9 ? tap : 8;  // a spare octet of the last block carries zero weights: any staged pixel will do
    toff[b] = ((tap / 3) * WSP + tap % 3) * RS + 2 * ch;
  }
  const int woff = (kb * 16 + m) * 16;
  float sc[NB], sh[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    sc[nb] = a.scale[16 * (NB * slice + nb) + m];
    sh[nb] = a.shift[16 * (NB * slice + nb) + m];
  }
  const float lo = a.relu ? 0.f : -__builtin_inff();
  const int Ho = a.H / POOL, Wo = a.W / POOL;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, a.bytes_x, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.bytes_y, 0x00020000);
  const int per = tiles_x * tiles_y;

  // staging items: (window pixel, channel quad); position inside the window is tile-invariant
  int it_off[NIT], it_r[NIT], it_c[NIT], it_lds[NIT], it_pl[PL ? NIT : 1];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + 256 * i;
    const int c4 = e % C4, pix = e / C4;
    it_r[i] = pix / WSX;
    it_c[i] = pix - it_r[i] * WSX;
    it_off[i] = ((it_r[i] * a.W + it_c[i]) * CIN + 4 * c4) * 4;
    it_lds[i] = (it_r[i] * WSP + it_c[i]) * RS + 8 * c4;
    if (e >= NITEMS) it_r[i] = -(1 << 20);  // never inside the image, never stored
    if constexpr (PL) it_pl[i] = (c4 == (a.plane_chan >> 2)) ? (it_r[i] * a.W + it_c[i]) * 4 : -1;
  }
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(PL ? a.plane : a.x), 0, PL ? a.bytes_p : 0, 0x00020000);
  const int pl_lane = a.plane_chan & 3;
  // Two windows in flight (round 6): the static walk requests the window of the tile AFTER next while a tile is computed, so a
  // request has a whole tile period to come back (one tile ahead left a short k-loop — Cin = 16 / 24: under a microsecond —
  // waiting for HBM at every tile).  Drawn tiles (a.tickets) keep one window in flight: the ticket walk knows one tile ahead.
  f32x4 preA[NIT], preB[NIT];
  float plA[PL ? NIT : 1], plB[PL ? NIT : 1];  // PL: the plane's values travel beside the window and are merged at the split
  auto fetch = [&](int T, f32x4(&pre)[NIT], float(&ppl)[PL ? NIT : 1]) {
    const int fb = T / per, fr = T - fb * per;
    const int fy0 = (fr / tiles_x) * TH - 1, fx0 = (fr % tiles_x) * TW - 1;
    const int base = ((fb * a.H + fy0) * a.W + fx0) * CIN * 4;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int Y = fy0 + it_r[i], X = fx0 + it_c[i];
      const bool ok = (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
      pre[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? base + it_off[i] : 0x7fffffff, 0, 0));
    }
    if constexpr (PL) {  // the quad that holds the plane's channel: that component comes from the plane (zero outside the image)
      const int pbase = ((fb * a.H + fy0) * a.W + fx0) * 4;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int Y = fy0 + it_r[i], X = fx0 + it_c[i];
        const bool ok = (it_pl[i] >= 0) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
        ppl[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, ok ? pbase + it_pl[i] : 0x7fffffff, 0, 0));
      }
    }
  };

  // the filter: a straight 16-byte copy, once per workgroup — ALL of a thread's loads (14 at Cin = 32) issued before its first
  // LDS store, and the first window's loads behind them, before those stores (as a plain load / store loop the copy was a chain
  // of dependent L2 round trips at the head of every launch: one workgroup per CU, nothing else to hide it)
  constexpr int NWQ = G::WBYTES / 16, NWI = (NWQ + 255) / 256;
  u32x4 wtmp[NWI];
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(a.wp) + (size_t)slice * G::WBYTES);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int e = tid + 256 * i;
      wtmp[i] = src[e < NWQ ? e : NWQ - 1];
    }
  }
  int tile = blockIdx.x, tnext = 0;
  const int gstep = (int)gridDim.x;
  if (dyn) {
    tk.begin(tk_sh);
    tile = tk.cur >= 0 ? tk.cur : ntiles;
  }
  if (tile < ntiles) fetch(tile, preA, plA);
  if (!dyn && tile + gstep < ntiles) fetch(tile + gstep, preB, plB);
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int e = tid + 256 * i;
    if (e < NWQ) reinterpret_cast<u32x4 *>(wl)[e] = wtmp[i];
  }
  RA_PS_AT(0);  // prologue: constants, filter copy, first window(s) requested

  // a window's registers -> three bf16 tiles in LDS, between the tile loop's two barriers
  auto stage = [&](f32x4(&pre)[NIT], float(&ppl)[PL ? NIT : 1]) {
    __syncthreads();  // the previous tile's operand reads are complete (and, first time round, the filter copy is issued)
    RA_PS_AT(1);  // top barrier
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      f32x4 v = pre[i];
      if constexpr (PL) {
        if (it_pl[i] >= 0) {
          v.x = pl_lane == 0 ? ppl[i] : v.x;
          v.y = pl_lane == 1 ? ppl[i] : v.y;
          v.z = pl_lane == 2 ? ppl[i] : v.z;
          v.w = pl_lane == 3 ? ppl[i] : v.w;
        }
      }
      unsigned H0, M0, L0, H1, M1, L1;
      split3_pair(v.x, v.y, H0, M0, L0);
      split3_pair(v.z, v.w, H1, M1, L1);
      if (it_r[i] >= 0) {
        unsigned char *rec = tin + it_lds[i];
        *reinterpret_cast<u32x2 *>(rec) = u32x2{H0, H1};
        *reinterpret_cast<u32x2 *>(rec + PLANE) = u32x2{M0, M1};
        *reinterpret_cast<u32x2 *>(rec + 2 * PLANE) = u32x2{L0, L1};
      }
    }
    RA_PS_AT(2);  // window arrived, split into three bf16 tiles, stored
    if (dyn) tk.publish(tk_sh);
    __syncthreads();
    RA_PS_AT(3);  // staging barrier
  };

  // the k-loop and the epilogue of one staged tile
  auto compute = [&](int T) {
    const int b = T / per, trem = T - b * per;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    f32x4 acc[GPW][NB];
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[g][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's 4 x 4-pixel groups: a whole row of four (TH = 16: row = wave) or half a row (TH = 8: row = wave / 2, groups
    // 2 (wave & 1) and the next)
    const int grow = GPW == 4 ? wave : wave >> 1, gx0 = GPW == 4 ? 0 : 2 * (wave & 1);
    const int wave_pix = (4 * grow * WSP + 4 * gx0) * RS + lane_pix;
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};  // six piece products per block, smallest first
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      s16x8 wv[3][NB];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wv[pc][nb] = *reinterpret_cast<const s16x8 *>(wl + ((blk * 3 + pc) * NB + nb) * 1024 + woff);
#pragma unroll
      for (int g = 0; g < GPW; ++g) {
        s16x8 av[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) av[pc] = *reinterpret_cast<const s16x8 *>(tin + pc * PLANE + wave_pix + 4 * g * RS + toff[blk]);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[g][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[PA[t]]), __builtin_bit_cast(bf16x8, wv[PB[t]][nb]),
                                                                 acc[g][nb], 0, 0, 0);
      }
    }
    RA_PS_AT(4);  // next window requested + the k-loop (9 blocks x 4 groups x 6 products x NB)
    // epilogue: lane (column n = m, D rows 4 kb + r) holds the four elements r = (dy, dx) of pooling window kb = (wy, wx) of group g
    const int wy = kb >> 1, wx = kb & 1;
#pragma unroll
    for (int g = 0; g < GPW; ++g)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int co = 16 * (NB * slice + nb) + m;
        const f32x4 v = acc[g][nb] * sc[nb] + sh[nb];
        if constexpr (POOL == 2) {
          const float o = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), lo);
          const int oy = ty * (TH / 2) + 2 * grow + wy, ox = tx * (TW / 2) + 2 * (gx0 + g) + wx;
          if ((ox < Wo) & (oy < Ho))  // (ragged last tiles: a width that is not a multiple of 16, a height that is not one of TH)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ry, (((b * Ho + oy) * Wo + ox) * a.Cout + co) * 4, 0, 0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int oy = ty * TH + 4 * grow + 2 * wy + (r >> 1), ox = tx * TW + 4 * (gx0 + g) + 2 * wx + (r & 1);
            if ((ox < Wo) & (oy < Ho))
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(v[r], lo)), ry,
                                                    (((b * Ho + oy) * Wo + ox) * a.Cout + co) * 4, 0, 0);
          }
        }
      }
    RA_PS_AT(5);  // epilogue: scale / shift, pool, ReLU, stores issued
  };

  if (dyn) {
    for (; tile < ntiles; tile = tnext) {
      stage(preA, plA);
      tk.read_next(tk_sh);
      tk.request();  // older than the prefetch loads below: consumed with them at the next tile's staging
      tk.step();
      tnext = tk.cur >= 0 ? tk.cur : ntiles;
      if (tnext < ntiles) fetch(tnext, preA, plA);  // in flight across the MFMA loop
      compute(tile);
    }
  } else {
    for (; tile < ntiles; tile += 2 * gstep) {
      stage(preA, plA);
      if (tile + 2 * gstep < ntiles) fetch(tile + 2 * gstep, preA, plA);  // in flight across two tiles
      compute(tile);
      if (tile + gstep >= ntiles) break;
      stage(preB, plB);
      if (tile + 3 * gstep < ntiles) fetch(tile + 3 * gstep, preB, plB);
      compute(tile + gstep);
    }
  }
  RA_PS_END;
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

template <int CIN, int POOL, int NB, bool PL = false>
int launch(const SArgs &a, hipStream_t st) {
  auto kern = conv_split_kernel<CIN, POOL, NB, PL>;
  constexpr int lds = Geo<CIN, NB>::LDS;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + Geo<CIN, NB>::TH - 1) / Geo<CIN, NB>::TH, ntiles = tiles_x * tiles_y * a.B, slices = a.Cout / (16 * NB);
  int gx = Geo<CIN, NB>::OCC * cu_count() / slices;  // one workgroup per CU (141 KB of LDS at Cin = 32), two where their LDS fits (Geo::OCC)
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  SArgs a2 = a;
  a2.tickets = ntiles >= kTicketMinTilesPerWg * gx ? take_ticket_slots(slices, gx) : nullptr;  // several tiles per workgroup and a bound scratch: drawn tiles
  hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(256), lds, st, a2, tiles_x, tiles_y, ntiles);
  return launch_status("ra_conv_split_f32");
}

inline unsigned short bf16_rne_host(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float bf16_to_float_host(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace csplit
}  // namespace ra

using namespace ra;

extern "C" int ra_conv_split_supported(int Cin, int Cout, int pool, int H, int W) {
  return (Cin == 16 || Cin == 24 || Cin == 32 || Cin == 64) && Cout > 0 && Cout % 16 == 0 && (pool == 1 || pool == 2) && H > 0 && W > 0 &&
         H % 4 == 0 && W % 4 == 0;  // (whole 4 x 4-pixel groups: the last tile of a row / column may be ragged — round 6)
}

extern "C" size_t ra_conv_split_packed_halfs(int Cin, int Cout) {
  if (!(Cin == 16 || Cin == 24 || Cin == 32 || Cin == 64) || Cout <= 0 || Cout % 16) return 0;
  return (size_t)(Cout / 16) * csplit::k_blocks(Cin) * 3 * 512;
}

namespace ra {
namespace csplit {
// one packed 16-bit word: (slice, block, piece, nb, kb, n, j) -> (tap, ci, co) -> the piece of the filter value there.
// transposed: w is a conv2d_transpose filter [3,3,Cout,Cin] (equivalently: the data gradient of a cnn layer whose filter is
// [3,3,Cin_w = Cout,Cout_w = Cin]): taps flipped, in / out swapped, as ra_conv_pack_weights(RA_CONV_TRANSPOSED).
__host__ __device__ inline void slot_of(size_t e, int Cin, int Cout, int nblk, int nbw, int &pc, int &tap, int &ci, int &co) {
  const int j = (int)(e & 7), nn = (int)((e >> 3) & 15), kb = (int)((e >> 7) & 3);
  size_t r = e >> 9;  // ((slice * nblk + b) * 3 + pc) * nbw + nb
  const int nb = (int)(r % nbw);
  r /= nbw;
  pc = (int)(r % 3);
  r /= 3;
  const int b = (int)(r % nblk), s = (int)(r / nblk);
  const int o = 4 * b + kb, opt = Cin / 8;  // (k_blocks: octet o of the pixel's neighbourhood)
  tap = o / opt;
  ci = 8 * (o % opt) + j;
  co = 16 * (nbw * s + nb) + nn;
}
__host__ __device__ inline float filter_at(const float *w, int Cin, int Cout, int transposed, int tap, int ci, int co) {
  if (tap >= 9) return 0.f;
  return transposed ? w[((size_t)(8 - tap) * Cout + co) * Cin + ci] : w[((size_t)tap * Cin + ci) * Cout + co];
}
__global__ __launch_bounds__(256) void pack_split_kernel(const float *w, int Cin, int Cout, int transposed, int nblk, int nbw, size_t n,
                                                         unsigned short *out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  int pc, tap, ci, co;
  slot_of(e, Cin, Cout, nblk, nbw, pc, tap, ci, co);
  float v = filter_at(w, Cin, Cout, transposed, tap, ci, co);
  unsigned short h = 0;
  for (int k = 0; k <= pc; ++k) {  // the pc-th piece: round, subtract, round ... (every difference exact)
    unsigned u = __builtin_bit_cast(unsigned, v);
    u += 0x7fffu + ((u >> 16) & 1u);
    h = (unsigned short)(u >> 16);
    v -= __builtin_bit_cast(float, (unsigned)h << 16);
  }
  out[e] = h;
}
}  // namespace csplit
}  // namespace ra

// The same packing on device pointers (the training step's filters change every step).
extern "C" int ra_conv_split_pack_weights_dev(const float *w, int Cin, int Cout, int transposed, unsigned short *out, void *stream) {
  const size_t n = ra_conv_split_packed_halfs(Cin, Cout);
  if (!w || !out || !n) return fail(RA_E_SHAPE, "ra_conv_split_pack_weights_dev: Cin=%d Cout=%d", Cin, Cout);
  const int nblk = csplit::k_blocks(Cin), nbw = (Cout % 32 == 0 && Cin < 64) ? 2 : 1;  // Cin = 64: 16 couts per workgroup
  hipLaunchKernelGGL(csplit::pack_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, Cin, Cout,
                     transposed ? 1 : 0, nblk, nbw, n, out);
  return launch_status("ra_conv_split_pack_weights_dev");
}

// w: the reference's [3,3,Cin,Cout] filter (host) -> out[slice][blk][piece][nb][kb][n][8] bf16: the exact three-piece split of
// W[tap][ci][cout] with (tap, ci) = K-slot (blk, kb, j) and cout = 16 (NB slice + nb) + n; slots beyond the nine taps are zero.
extern "C" int ra_conv_split_pack_weights(const float *w, int Cin, int Cout, unsigned short *out) {
  const size_t n = ra_conv_split_packed_halfs(Cin, Cout);
  if (!w || !out || !n) return fail(RA_E_SHAPE, "ra_conv_split_pack_weights: Cin=%d Cout=%d", Cin, Cout);
  const int nblk = csplit::k_blocks(Cin), nbw = (Cout % 32 == 0 && Cin < 64) ? 2 : 1;  // Cin = 64: 16 couts per workgroup
  for (size_t e = 0; e < n; ++e) {
    int pc, tap, ci, co;
    csplit::slot_of(e, Cin, Cout, nblk, nbw, pc, tap, ci, co);
    float v = csplit::filter_at(w, Cin, Cout, 0, tap, ci, co);
    unsigned short h = 0;
    for (int k = 0; k <= pc; ++k) {
      h = csplit::bf16_rne_host(v);
      v -= csplit::bf16_to_float_host(h);
    }
    out[e] = h;
  }
  return 0;
}

namespace ra {
namespace csplit {
int run(const float *x, int B, int H, int W, int Cin, const float *plane, int plane_chan, const unsigned short *wpacked, const float *scale,
        const float *shift, int Cout, int relu, int pool, float *y, void *stream, const char *who) {
  if (!x || !wpacked || !scale || !shift || !y || B <= 0) return fail(RA_E_INVALID, "%s: bad argument", who);
  if (!ra_conv_split_supported(Cin, Cout, pool, H, W))
    return fail(RA_E_SHAPE, "%s: Cin=%d Cout=%d pool=%d %dx%d", who, Cin, Cout, pool, H, W);
  if (plane && (plane_chan < 0 || plane_chan >= Cin)) return fail(RA_E_INVALID, "%s: plane_chan %d outside [0, %d)", who, plane_chan, Cin);
  const size_t bx = (size_t)B * H * W * Cin * 4, by = (size_t)B * (H / pool) * (W / pool) * Cout * 4;
  if (bx >= (1ull << 31) || by >= (1ull << 31)) return fail(RA_E_SHAPE, "%s: a tensor exceeds 2 GiB", who);
  SArgs a{};
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
  a.bytes_x = (int)bx;
  a.bytes_y = (int)by;
  a.plane = plane;
  a.plane_chan = plane ? plane_chan : 0;
  a.bytes_p = (int)((size_t)B * H * W * 4);
  hipStream_t st = as_stream(stream);
  const bool two = Cout % 32 == 0 && Cin < 64;  // 32 output channels per workgroup (Cin = 64: 16, its filter slice is 54 KB)
#define RA_CS(CIN_)                                                                                             \
  if (Cin == CIN_) {                                                                                            \
    if (plane) {                                                                                                \
      if (two) return pool == 2 ? launch<CIN_, 2, 2, true>(a, st) : launch<CIN_, 1, 2, true>(a, st);            \
      return pool == 2 ? launch<CIN_, 2, 1, true>(a, st) : launch<CIN_, 1, 1, true>(a, st);                     \
    }                                                                                                           \
    if (two) return pool == 2 ? launch<CIN_, 2, 2>(a, st) : launch<CIN_, 1, 2>(a, st);                          \
    return pool == 2 ? launch<CIN_, 2, 1>(a, st) : launch<CIN_, 1, 1>(a, st);                                   \
  }
  RA_CS(16)
  RA_CS(24)
  RA_CS(32)
#undef RA_CS
  if (plane) return fail(RA_E_SHAPE, "%s: the plane form is built for Cin 16 / 24 / 32", who);
  return pool == 2 ? launch<64, 2, 1>(a, st) : launch<64, 1, 1>(a, st);
}
}  // namespace csplit
}  // namespace ra

extern "C" int ra_conv_split_f32(const float *x, int B, int H, int W, int Cin, const unsigned short *wpacked, const float *scale,
                                 const float *shift, int Cout, int relu, int pool, float *y, void *stream) {
  return csplit::run(x, B, H, W, Cin, nullptr, 0, wpacked, scale, shift, Cout, relu, pool, y, stream, "ra_conv_split_f32");
}

extern "C" int ra_conv_split_plane_f32(const float *x, int B, int H, int W, int Cin, const float *plane, int plane_chan,
                                       const unsigned short *wpacked, const float *scale, const float *shift, int Cout, int relu, int pool,
                                       float *y, void *stream) {
  if (!plane) return fail(RA_E_INVALID, "ra_conv_split_plane_f32: null plane");
  return csplit::run(x, B, H, W, Cin, plane, plane_chan, wpacked, scale, shift, Cout, relu, pool, y, stream, "ra_conv_split_plane_f32");
}
