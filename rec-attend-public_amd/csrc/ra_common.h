// Shared helpers for librecattend.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "recattend.h"

namespace ra {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Returns 0 or the hipError_t of the launch that was just issued.
inline int launch_status(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
  }
  return 0;
}

inline int fail(int code, const char *fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

constexpr int kWave = 64;  // CDNA wavefront

// Wave priority of the decode loop's LATENCY-bound kernels (controller, patch-sized convs, extract, paste, score).  In the
// decode pipeline they share every SIMD with another batch's MFMA-bound controller-CNN waves, and the arbiter deals issue
// slots oldest-first: a tail wave's few hundred dependent instructions each queue behind the other waves' 32-cycle MFMAs.
// s_setprio would let the tail wave issue as soon as the pipe is free.  MEASURED in round 5 and NOT the lever: with
// RA_TAIL_PRIO=3 on all of them the pipelined rate falls 52.3k -> 50.7k, on any subset (RA_TAIL_PRIO_MASK) it is equal or
// lower within the run-to-run spread (profiles/r05_decode_schedule_probes.txt).  Kept as a measuring aid, default 0 = off.
int tail_prio(int kind = 4);  // ra_core.hip; kind: 1 = controller, 2 = patch-sized conv, 4 = extract / paste / score (RA_TAIL_PRIO_MASK)
__device__ __forceinline__ void raise_prio(int p) {
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
}

// ---- Dynamic tile tickets for the persistent conv kernels (round 5) ------------------------------------------------------
// A persistent grid that walks its tiles statically (tile = blockIdx, + gridDim, ...) is only as fast as its LAST workgroup
// to start: in the decode pipeline another batch's controller holds 32 CUs for ~110 us, the workgroups that were meant for
// those CUs start when the first ones finish, and the launch takes up to twice as long (tools/contention_probe.py: the
// controller CNN x1.38 with controllers as company).  With tickets a workgroup DRAWS its next tile, so a late or missing
// workgroup costs its share of the chip and nothing more.
//   * one pool of tiles per XCD — tiles [x * chunk, (x + 1) * chunk) belong to the XCD with HW_REG_XCC_ID = x — drawn with an
//     atomic that executes in that XCD's L2 (buffer atomic, glc, no sc1: 0.6 us per draw; one agent-scope counter serialises
//     at ~13 ns per draw and would cost 8192 tiles 120 us — tools/ticket_probe.hip).  Only workgroups ON XCD x touch pool x,
//     so the XCD-local L2 is the coherence point; the pools are zeroed once per forward by an ordinary launch (kernel
//     boundaries make that visible), every ticketed launch of the forward owns a fresh slot of pools.
//   * two tickets are kept in flight: the draw for the tile after next is issued when a tile starts and handed to the other
//     waves (one LDS word, the barriers the tile loop has anyway) when it ends, so the next tile's loads can still be prefetched.
// The host side (ra_core.hip): ra_tile_tickets_bind(scratch, slots) makes a zeroed scratch current for the calling thread;
// every ticketed launch takes the next slot(s), ra_tile_tickets_bind(nullptr, 0) ends it.  Unbound = the static walk.
constexpr int kTicketPoolStride = 32;                      // unsigned per pool: one 128-byte line each
constexpr int kTicketSlotWords = 8 * kTicketPoolStride;    // one slot = the 8 XCD pools of one (launch, channel slice)
// Tiles are drawn only where a workgroup has at least this many: with two, the two draws in flight at kernel entry ARE the
// static walk and only their latency remains (cfg2, 16 images: L4 22.6 -> 28.8 us alone, L6 16.0 -> 17.9), while from three or
// four upwards the launch is as fast or faster alone (L0+L1 79.5 -> 72.1 us back to back: the pools also even out the XCDs)
// and loses 3-8 % instead of 13-50 % to another slot's controller (tools/contention_by_layer.py).
constexpr int kTicketMinTilesPerWg = 3;
unsigned *take_ticket_slots(int n, int grid_x);  // ra_core.hip: n consecutive slots of the bound scratch, or nullptr
int xcc_census_ok();  // ra_core.hip: 1 = this device's workgroups report exactly the XCC ids 0..7, 0 = not, -1 = cannot tell now (stream capture)

__device__ __forceinline__ int xcc_id() {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  return (int)(id & 7u);
}
__device__ __forceinline__ unsigned ticket_draw(unsigned *pool) {  // old value; performed in this XCD's L2
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(pool, 0, 4, 0x00020000);
  return (unsigned)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r, 0, 0, 1);
}
// The walk of one workgroup over its XCD's pool.  Uniform state; `pend` is meaningful in thread 0 only.  Per tile (a tile loop
// with a "staging" barrier — the tile's prefetched registers have been written to LDS — and at least one more barrier):
//   publish(sh)    BEFORE the staging barrier: the draw made one tile ago becomes visible to the other waves.  The tile's
//                  prefetched loads were issued after that draw and have just been consumed, so the wait is free (placed at
//                  the end of a tile instead, the wait for the draw became s_waitcnt vmcnt(0) behind the tile's output stores:
//                  +2 us per launch at cfg2's L2+L3 and L5)
//   read_next(sh)  AFTER the staging barrier: nxt = the tile to prefetch now and to run next (-1: none)
//   request()      right after it, BEFORE the prefetch loads are issued: the draw for the tile after next
//   step()         cur <- nxt
struct TicketWalk {
  unsigned *pool;
  int base, lim;  // the pool's tiles are base + [0, lim)
  int cur, nxt;   // tile indices; -1 = none
  unsigned pend;
  __device__ __forceinline__ int tile_of(unsigned t) const { return t < (unsigned)lim ? base + (int)t : -1; }
  // issue(): FIRST thing in the kernel — the two draws of a workgroup's first tiles go out and fly across its prologue (filter
  // loads, LDS clearing: 0.6 us each, otherwise exposed at every launch); slot: the launch's slot (of this channel slice).
  // begin(): where the first tile is needed; sh: two LDS words.  Contains one barrier.
  unsigned t0, t1;
  __device__ __forceinline__ void issue(unsigned *slot, int ntiles) {
    const int x = xcc_id(), chunk = (ntiles + 7) >> 3;
    pool = slot + x * kTicketPoolStride;
    base = x * chunk;
    lim = base + chunk <= ntiles ? chunk : (ntiles > base ? ntiles - base : 0);
    t0 = t1 = 0;
    if (threadIdx.x == 0) {
      t0 = ticket_draw(pool);
      t1 = ticket_draw(pool);
    }
  }
  __device__ __forceinline__ void begin(volatile unsigned *sh) {
    if (threadIdx.x == 0) sh[1] = t0;
    __syncthreads();
    cur = tile_of(__builtin_amdgcn_readfirstlane(sh[1]));
    nxt = -1;
    pend = t1;  // the first tile's publish hands the second draw over
  }
  __device__ __forceinline__ void request() {
    if (threadIdx.x == 0) pend = ticket_draw(pool);
  }
  __device__ __forceinline__ void read_next(volatile unsigned *sh) { nxt = tile_of(__builtin_amdgcn_readfirstlane(sh[0])); }
  __device__ __forceinline__ void publish(volatile unsigned *sh) {
    if (threadIdx.x == 0) sh[0] = pend;
  }
  __device__ __forceinline__ void step() { cur = nxt; }
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace ra
