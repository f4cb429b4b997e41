// Hungarian matching for librecattend.so — host (CPU) and device (gfx950) entry points built
// from ONE solver source so both are the same arithmetic.
//
// Behavioural contract = the reference TF op `Hungarian` (hungarian.cc:26-30,36-85): max-weight
// bipartite matching by primal-dual vertex-cover updates where every matching phase is a
// max-flow recomputed from scratch with the reference's non-textbook BFS (nodes marked when
// popped, parents overwritten by later pushers; hungarian.cc:107-177).  float32 state,
// double-typed comparisons against 1e-6 and 1.0 (hungarian.cc:18,292,302,318,428), ordered
// sets (std::set<int> -> bitmaps scanned in ascending order).  Only add / sub / min /
// compare touch the data, so host and device results are bit-identical.
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ra_common.h"

namespace ra {
namespace hung {

constexpr int kMaxIter = 1000;  // MAX_NUM_ITERATION, hungarian.cc:20
#define RA_HUNG_EPS 1e-6        // EPSILON (a double), hungarian.cc:18

// The device keeps a window of the BFS queue in LDS; typing it as an LDS pointer keeps the compiler
// from folding "ring or global queue" into one generic (flat) access.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) int ring_int;
#else
typedef int ring_int;
#endif

// Flow network + scratch for one [nx, ny] problem, carved out of a caller buffer.
struct Scratch {
  int n;  // nx + ny + 2 nodes: s = 0, X = 1..nx, Y = nx+1..nx+ny, t = n-1 (hungarian.cc:197-202)
  float *cap, *flow, *res, *eq;
  int *queue, *parent;
  unsigned char *mark, *inS, *inT, *inN;
  ring_int *ring;  // device: LDS window over the tail of the BFS queue (ring_n entries, a power of two)
  int ring_n;  // 0 = none, every queue access goes to `queue`
  int chunked;  // device, n <= 64: the chunk-parallel search (augment_wave64c); needs 64 more LDS entries behind the ring
};

// The scratch has a HOT part (flow network, equality graph, marks: a few n^2 floats, touched by
// every step of the serial solver) and the BFS QUEUE ((kMaxIter + 2) * n ints, touched
// sequentially).  On the device the hot part lives in LDS when it fits — a dependent access costs
// ~60 cycles there against most of a microsecond in global memory — and the queue stays in the
// caller's workspace.
__host__ __device__ inline size_t hot_bytes(int nx, int ny) {
  size_t n = (size_t)nx + ny + 2;
  size_t f = 3 * n * n + (size_t)nx * ny;
  size_t b = n + nx + 2 * (size_t)ny;
  return ((f + n) * 4 + b + 15) / 16 * 16;
}
__host__ __device__ inline size_t queue_bytes(int nx, int ny) {
  size_t n = (size_t)nx + ny + 2;
  return ((size_t)(kMaxIter + 2) * n * 4 + 15) / 16 * 16;
}
__host__ __device__ inline size_t scratch_bytes(int nx, int ny) { return hot_bytes(nx, ny) + queue_bytes(nx, ny); }

__host__ __device__ inline Scratch carve(void *hot, void *queue, int nx, int ny) {
  Scratch s;
  s.n = nx + ny + 2;
  size_t nn = (size_t)s.n * s.n;
  float *f = reinterpret_cast<float *>(hot);
  s.cap = f;
  s.flow = f + nn;
  s.res = f + 2 * nn;
  s.eq = f + 3 * nn;
  s.parent = reinterpret_cast<int *>(s.eq + (size_t)nx * ny);
  unsigned char *bp = reinterpret_cast<unsigned char *>(s.parent + s.n);
  s.mark = bp;
  s.inS = bp + s.n;
  s.inT = s.inS + nx;
  s.inN = s.inT + ny;
  s.queue = reinterpret_cast<int *>(queue);
  s.ring = nullptr;
  s.ring_n = 0;
  s.chunked = 0;
  return s;
}

// One augmenting-path search + push.  1 = augmented, 0 = no path, <0 = reference LOG(FATAL).
__host__ __device__ inline int augment(Scratch &g, float cap_max) {
  const int n = g.n, src = 0, dst = n - 1;
  int qh = 0, qt = 0;
  g.queue[qt++] = src;
  for (int v = 0; v < n; ++v) {
    g.mark[v] = 0;
    g.parent[v] = -1;
  }
  bool reached = false;
  for (int it = 0; qt > qh && it <= kMaxIter; ++it) {
    if (it == kMaxIter) return RA_E_HUNG_BFS;
    const int v = g.queue[qh++];
    g.mark[v] = 1;  // marked on pop, not on push
    if (v == dst) {
      reached = true;
      break;
    }
    const float *row = g.res + (size_t)v * n;
    for (int u = 0; u < n; ++u)
      if (!g.mark[u] && row[u] > 0) {
        g.queue[qt++] = u;
        g.parent[u] = v;  // later pushers overwrite
      }
  }
  if (!reached) return 0;

  float bottleneck = cap_max;  // capacity.maxCoeff(), hungarian.cc:144 (constant within a max-flow)
  int v = dst;
  for (int it = 0; g.parent[v] != -1 && it <= kMaxIter; ++it) {
    if (it == kMaxIter) return RA_E_HUNG_PATH;
    const float r = g.res[(size_t)g.parent[v] * n + v];
    bottleneck = (bottleneck < r) ? bottleneck : r;  // MIN macro
    v = g.parent[v];
  }
  v = dst;
  for (int it = 0; g.parent[v] != -1 && it <= kMaxIter; ++it) {
    if (it == kMaxIter) return RA_E_HUNG_PATH;
    const int p = g.parent[v];
    if (g.cap[(size_t)p * n + v] > 0)
      g.flow[(size_t)p * n + v] += bottleneck;
    else
      g.flow[(size_t)v * n + p] -= bottleneck;
    g.res[(size_t)p * n + v] -= bottleneck;
    g.res[(size_t)v * n + p] += bottleneck;
    v = p;
  }
  return 1;
}

// Max bipartite matching of the 0/1 graph g.eq via max-flow from scratch (hungarian.cc:179-217).
__host__ __device__ inline int rematch(Scratch &g, int nx, int ny, float *M) {
  const int n = g.n, dst = n - 1;
  for (size_t k = 0; k < (size_t)n * n; ++k) g.cap[k] = 0.0f;
  for (int x = 0; x < nx; ++x) {
    g.cap[1 + x] = 1.0f;  // s -> x
    for (int y = 0; y < ny; ++y) g.cap[(size_t)(1 + x) * n + (1 + nx + y)] = g.eq[x * ny + y];
  }
  for (int y = 0; y < ny; ++y) g.cap[(size_t)(1 + nx + y) * n + dst] = 1.0f;  // y -> t
  for (size_t k = 0; k < (size_t)n * n; ++k) {
    g.flow[k] = 0.0f;
    g.res[k] = g.cap[k];
  }
  float cap_max = g.cap[0];
  for (int k = 1; k < n * n; ++k) cap_max = (g.cap[k] > cap_max) ? g.cap[k] : cap_max;
  for (int it = 0;; ++it) {
    const int r = augment(g, cap_max);
    if (r < 0) return r;
    if (r == 0 || it > kMaxIter) break;
    if (it == kMaxIter) return RA_E_HUNG_FLOW;
  }
  for (int x = 0; x < nx; ++x)
    for (int y = 0; y < ny; ++y) M[x * ny + y] = g.flow[(size_t)(1 + x) * n + (1 + nx + y)];
  return 0;
}

// hungarian.cc:219-248: every vertex of the smaller side is matched.
__host__ __device__ inline bool saturating(const float *M, int nx, int ny) {
  const bool by_col = nx >= ny;
  const int outer = by_col ? ny : nx, inner = by_col ? nx : ny;
  for (int a = 0; a < outer; ++a) {
    float sum = 0;
    for (int b = 0; b < inner; ++b) sum += by_col ? M[b * ny + a] : M[a * ny + b];
    if (sum == 0) return false;
  }
  return true;
}

// hungarian.cc:335-488.  0 solved, 1 outer cap (partial result kept), <0 fatal.
__host__ __device__ inline int solve(const float *w, int nx, int ny, float *M, float *cx,
                                     float *cy, void *hot_buf, void *queue_buf) {
  Scratch g = carve(hot_buf, queue_buf, nx, ny);
  for (int x = 0; x < nx; ++x) {
    float top = w[x * ny];
    for (int y = 1; y < ny; ++y) top = (w[x * ny + y] > top) ? w[x * ny + y] : top;
    cx[x] = top;
    g.inS[x] = 0;
  }
  for (int y = 0; y < ny; ++y) {
    cy[y] = 0.0f;
    g.inT[y] = 0;
  }
  for (int k = 0; k < nx * ny; ++k) M[k] = 0.0f;
  int cntT = 0;
  bool need_match = true;

  for (int it = 0; it <= kMaxIter; ++it) {
    if (it == kMaxIter) return 1;
    // equality graph (hungarian.cc:309-325): float expression, double comparison
    for (int x = 0; x < nx; ++x)
      for (int y = 0; y < ny; ++y) {
        const float slack = cx[x] + cy[y] - w[x * ny + y];
        const float mag = (slack > 0) ? slack : -slack;
        g.eq[x * ny + y] = (mag <= RA_HUNG_EPS && (cx[x] > 0 || cy[y] > 0)) ? 1.0f : 0.0f;
      }
    if (need_match) {
      const int r = rematch(g, nx, ny, M);
      if (r < 0) return r;
      if (saturating(M, nx, ny)) return 0;
      for (int x = 0; x < nx; ++x) {  // first exposed x seeds S (hungarian.cc:394-403)
        bool exposed = true;
        for (int y = 0; y < ny && exposed; ++y) exposed = !(M[x * ny + y] == 1.0);
        if (exposed) {
          for (int a = 0; a < nx; ++a) g.inS[a] = 0;
          for (int b = 0; b < ny; ++b) g.inT[b] = 0;
          g.inS[x] = 1;
          cntT = 0;
          break;
        }
      }
    }
    int cntN = 0;  // N(S) in the equality graph
    for (int y = 0; y < ny; ++y) g.inN[y] = 0;
    for (int x = 0; x < nx; ++x)
      if (g.inS[x])
        for (int y = 0; y < ny; ++y)
          if (g.eq[x * ny + y] > 0 && !g.inN[y]) {
            g.inN[y] = 1;
            ++cntN;
          }
    bool same = cntN == cntT;
    for (int y = 0; y < ny && same; ++y) same = !(g.inN[y] && !g.inT[y]);

    if (same) {  // cover update (hungarian.cc:415-443)
      float a = FLT_MAX;
      for (int x = 0; x < nx; ++x)
        if (g.inS[x])
          for (int y = 0; y < ny; ++y)
            if (!g.inT[y]) {
              const float slack = cx[x] + cy[y] - w[x * ny + y];
              a = (a < slack) ? a : slack;
            }
      if (a < RA_HUNG_EPS) {
        need_match = true;
        continue;
      }
      for (int x = 0; x < nx; ++x)
        if (g.inS[x]) cx[x] -= a;
      for (int y = 0; y < ny; ++y)
        if (g.inT[y]) cy[y] += a;
    } else {  // grow the alternating tree (hungarian.cc:444-483)
      for (int j = 0; cntN > cntT && j <= kMaxIter; ++j) {
        if (j == kMaxIter) return RA_E_HUNG_EQUALIZE;
        int y = 0;
        while (!(g.inN[y] && !g.inT[y])) ++y;  // smallest y in N(S) \ T
        int z = -1;
        for (int x = 0; x < nx; ++x)
          if (M[x * ny + y] == 1.0) {
            z = x;
            break;
          }
        if (z < 0) {
          need_match = true;
          break;
        }
        need_match = false;
        g.inS[z] = 1;
        for (int v = 0; v < ny; ++v)
          if (g.eq[z * ny + v] > 0.0 && !g.inN[v]) {
            g.inN[v] = 1;
            ++cntN;
          }
        g.inT[y] = 1;
        ++cntT;
      }
    }
  }
  return 1;
}

// ---- wave-cooperative form of the same algorithm (device only) ----
// The 64 lanes of one wave execute solve_wave() together: every data-parallel loop of the serial
// solver (equality graph, network reset, neighbourhood marks, slack minimum, cover update, ...) is
// strided over the lanes, every order-dependent step keeps the serial order (BFS pushes go in
// ascending node order through a ballot, the path walk and the tree growth run on uniform
// scalars).  min / max / or-reductions are exact and order-independent, so the result is the
// serial solver's, bit for bit (tests/test_hungarian.py checks it on the reference's vectors and
// hundreds of random problems).  All lanes hold identical copies of the scalar state.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void wsync() { __syncthreads(); }  // one wave per workgroup: orders LDS/global traffic
__device__ __forceinline__ float wave_min(float v) {
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(v, o);
    v = (t < v) ? t : v;
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(v, o);
    v = (t > v) ? t : v;
  }
  return v;
}

// The BFS queue can grow to (kMaxIter + 2) * n entries (nodes are re-pushed until popped), which
// only global memory holds — but a pop that waits on a global load, behind a fence on the pushes
// before it, is most of a microsecond, and a solve makes tens of thousands of them.  So the queue
// lives in an LDS ring while its LIVE window [qh, qt) fits (it always does at the path's sizes);
// the first push that would overrun the ring moves the window to the global queue and the search
// continues there.  Same pops in the same order either way.
__device__ __forceinline__ int augment_wave(Scratch &g, float cap_max, int lane) {
  const int n = g.n, src = 0, dst = n - 1;
  const int R = g.ring_n;
  bool spilled = (R == 0);
  for (int v = lane; v < n; v += 64) {
    g.mark[v] = 0;
    g.parent[v] = -1;
  }
  if (lane == 0) {
    if (spilled)
      g.queue[0] = src;
    else
      g.ring[0] = src;
  }
  if (spilled) __threadfence_block();
  wsync();
  int qh = 0, qt = 1;
  bool reached = false;
  for (int it = 0; qt > qh && it <= kMaxIter; ++it) {
    if (it == kMaxIter) return RA_E_HUNG_BFS;
    int v;
    if (spilled)
      v = g.queue[qh];
    else
      v = g.ring[qh & (R - 1)];
    ++qh;
    if (lane == 0) g.mark[v] = 1;  // marked on pop, not on push
    wsync();
    if (v == dst) {
      reached = true;
      break;
    }
    const float *row = g.res + (size_t)v * n;
    for (int base = 0; base < n; base += 64) {
      const int u = base + lane;
      const bool push = u < n && !g.mark[u] && row[u] > 0;
      const unsigned long long m = __ballot(push);
      const int cnt = __popcll(m);
      if (!spilled && qt + cnt - qh > R) {
        for (int i = qh + lane; i < qt; i += 64) g.queue[i] = g.ring[i & (R - 1)];
        spilled = true;
      }
      if (push) {
        const int at = qt + __popcll(m & ((1ull << lane) - 1ull));  // ascending u, like the serial scan
        if (spilled)
          g.queue[at] = u;
        else
          g.ring[at & (R - 1)] = u;
        g.parent[u] = v;  // later pushers overwrite
      }
      qt += cnt;
    }
    if (spilled) __threadfence_block();
    wsync();
  }
  if (!reached) return 0;
  int rc = 1;
  if (lane == 0) {  // the augmenting path is at most n long: walk it on one lane
    float bottleneck = cap_max;  // capacity.maxCoeff(), hungarian.cc:144
    int v = dst;
    for (int it = 0; g.parent[v] != -1 && it <= kMaxIter; ++it) {
      if (it == kMaxIter) {
        rc = RA_E_HUNG_PATH;
        break;
      }
      const float r = g.res[(size_t)g.parent[v] * n + v];
      bottleneck = (bottleneck < r) ? bottleneck : r;
      v = g.parent[v];
    }
    v = dst;
    for (int it = 0; rc == 1 && g.parent[v] != -1 && it <= kMaxIter; ++it) {
      if (it == kMaxIter) {
        rc = RA_E_HUNG_PATH;
        break;
      }
      const int p = g.parent[v];
      if (g.cap[(size_t)p * n + v] > 0)
        g.flow[(size_t)p * n + v] += bottleneck;
      else
        g.flow[(size_t)v * n + p] -= bottleneck;
      g.res[(size_t)p * n + v] -= bottleneck;
      g.res[(size_t)v * n + p] += bottleneck;
      v = p;
    }
  }
  rc = __shfl(rc, 0);
  wsync();
  return rc;
}

__device__ __forceinline__ int rematch_wave(Scratch &g, int nx, int ny, float *M, int lane) {
  const int n = g.n, dst = n - 1, nn = n * n;
  for (int k = lane; k < nn; k += 64) g.cap[k] = 0.0f;
  wsync();
  for (int k = lane; k < nx * ny; k += 64) {
    const int x = k / ny, y = k - x * ny;
    g.cap[(size_t)(1 + x) * n + (1 + nx + y)] = g.eq[k];
  }
  for (int x = lane; x < nx; x += 64) g.cap[1 + x] = 1.0f;                                // s -> x
  for (int y = lane; y < ny; y += 64) g.cap[(size_t)(1 + nx + y) * n + dst] = 1.0f;      // y -> t
  wsync();
  float cap_max = -FLT_MAX;
  for (int k = lane; k < nn; k += 64) {
    const float c = g.cap[k];
    g.flow[k] = 0.0f;
    g.res[k] = c;
    cap_max = (c > cap_max) ? c : cap_max;
  }
  cap_max = wave_max(cap_max);
  wsync();
  for (int it = 0;; ++it) {
    const int r = augment_wave(g, cap_max, lane);
    if (r < 0) return r;
    if (r == 0 || it > kMaxIter) break;
    if (it == kMaxIter) return RA_E_HUNG_FLOW;
  }
  for (int k = lane; k < nx * ny; k += 64) {
    const int x = k / ny, y = k - x * ny;
    M[k] = g.flow[(size_t)(1 + x) * n + (1 + nx + y)];
  }
  wsync();
  return 0;
}

// ---- n <= 64: the search state lives in registers ----
// With at most 64 nodes the BFS needs no memory round trip per pop: lane v keeps row v of the
// residual graph as a 64-bit adjacency mask (bit u = res[v][u] > 0) and parent[v]; the marks are
// one uniform 64-bit mask; the queue is read 64 entries at a time into a register window and
// popped with v_readlane.  A pop is then a handful of scalar instructions plus one LDS store per
// pushed node, instead of two LDS round trips and two barriers.  The float network (cap / flow /
// res) stays in LDS and is updated along the path exactly as before; the masks are derived from it.
// One wave's LDS operations execute in issue order, so only the compiler needs a fence.
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
// Inclusive scans over the 64 lanes on the DPP data path (row_shr 1, 2, 4, 8 inside the rows of 16, then the row_bcast15 /
// row_bcast31 steps that carry a row's total into the following rows): 12 VALU instructions where six __shfl_up steps were
// six dependent ds_bpermute round trips.  Lanes without a source take 0, the identity of both operations.
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned dpp0(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xf, false);
}
__device__ __forceinline__ unsigned scan_or64(unsigned x) {
  x |= dpp0<0x111, 0xf>(x);
  x |= dpp0<0x112, 0xf>(x);
  x |= dpp0<0x114, 0xf>(x);
  x |= dpp0<0x118, 0xf>(x);
  x |= dpp0<0x142, 0xa>(x);  // row_bcast15 -> rows 1, 3
  x |= dpp0<0x143, 0xc>(x);  // row_bcast31 -> rows 2, 3
  return x;
}
__device__ __forceinline__ int scan_add64(int v) {
  unsigned x = (unsigned)v;
  x += dpp0<0x111, 0xf>(x);
  x += dpp0<0x112, 0xf>(x);
  x += dpp0<0x114, 0xf>(x);
  x += dpp0<0x118, 0xf>(x);
  x += dpp0<0x142, 0xa>(x);
  x += dpp0<0x143, 0xc>(x);
  return (int)x;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long x, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// The augmenting path's walk for the register-resident searches below, on uniform scalars: every lane reads the same
// LDS words, lane 0 writes, lanes p and v update their adjacency masks.  parent: lane u holds parent[u].
__device__ __forceinline__ int push_path64(Scratch &g, float cap_max, int lane, unsigned long long &adj, int parent) {
  const int n = g.n, dst = n - 1;
  // the path walk, on uniform scalars: every lane reads the same LDS words, lane 0 writes
  float bottleneck = cap_max;  // capacity.maxCoeff(), hungarian.cc:144
  int v = dst;
  for (int it = 0;; ++it) {
    const int p = __builtin_amdgcn_readlane(parent, v);
    if (p == -1) break;
    if (it == kMaxIter) return RA_E_HUNG_PATH;
    const float r = g.res[(size_t)p * n + v];
    bottleneck = (bottleneck < r) ? bottleneck : r;
    v = p;
  }
  v = dst;
  for (int it = 0;; ++it) {
    const int p = __builtin_amdgcn_readlane(parent, v);
    if (p == -1) break;
    if (it == kMaxIter) return RA_E_HUNG_PATH;
    const size_t pv = (size_t)p * n + v, vp = (size_t)v * n + p;
    const bool fwd = g.cap[pv] > 0;
    const float f = fwd ? g.flow[pv] + bottleneck : g.flow[vp] - bottleneck;
    const float rpv = g.res[pv] - bottleneck, rvp = g.res[vp] + bottleneck;
    lds_order();
    if (lane == 0) {
      g.flow[fwd ? pv : vp] = f;
      g.res[pv] = rpv;
      g.res[vp] = rvp;
    }
    lds_order();
    if (lane == p) adj = (adj & ~(1ull << v)) | ((rpv > 0) ? (1ull << v) : 0ull);
    if (lane == v) adj = (adj & ~(1ull << p)) | ((rvp > 0) ? (1ull << p) : 0ull);
    v = p;
  }
  return 1;
}


__device__ __forceinline__ int augment_wave64(Scratch &g, float cap_max, int lane, unsigned long long &adj) {
  const int n = g.n, src = 0, dst = n - 1;
  const int R = g.ring_n;  // >= 64 on this path
  bool spilled = false;
  unsigned long long marks = 0;
  int parent = -1;
  if (lane == 0) g.ring[0] = src;
  int win = src, wbase = 0, wvalid = 1;  // queue[wbase + lane] for wbase + lane < wvalid
  int qh = 0, qt = 1;
  bool reached = false;
  for (int it = 0; qt > qh && it <= kMaxIter; ++it) {
    if (it == kMaxIter) return RA_E_HUNG_BFS;
    if (qh >= wvalid) {  // refill the window from the queue
      if (spilled) __threadfence_block();
      lds_order();
      wbase = qh;
      wvalid = (qt < qh + 64) ? qt : qh + 64;
      const int at = qh + lane;
      win = 0;
      if (at < wvalid) {
        if (spilled)
          win = g.queue[at];
        else
          win = g.ring[at & (R - 1)];
      }
    }
    const int v = __builtin_amdgcn_readlane(win, qh - wbase);
    ++qh;
    marks |= 1ull << v;  // marked on pop, not on push
    if (v == dst) {
      reached = true;
      break;
    }
    const unsigned long long m = readlane64(adj, v) & ~marks;
    if (m == 0ull) continue;
    const int cnt = __popcll(m);
    if (!spilled && qt + cnt - qh > R) {  // the live window would overrun the ring: go global
      lds_order();
      for (int i = qh + lane; i < qt; i += 64) g.queue[i] = g.ring[i & (R - 1)];
      spilled = true;
    }
    if ((m >> lane) & 1ull) {
      const int at = qt + __popcll(m & ((1ull << lane) - 1ull));  // ascending u, like the serial scan
      if (spilled)
        g.queue[at] = lane;
      else
        g.ring[at & (R - 1)] = lane;
      parent = v;  // later pushers overwrite
    }
    qt += cnt;
  }
  if (!reached) return 0;
  return push_path64(g, cap_max, lane, adj, parent);
}

// Chunk-parallel form of the same search.  The queue entries [qh, qt) are known before any of them is popped, and
// a pop only ever ADDS marks (pushes do not mark), so up to 64 pops are processed at once, lane k taking queue entry
// qh + k: the marks pop k sees are the marks so far OR-ed with the bits of the nodes popped at lanes <= k (an inclusive
// prefix-OR over the lanes), its pushes go to the queue behind the pushes of lanes < k (an exclusive prefix sum of the
// push counts), parent[u] is the node popped at the LAST lane that pushed u (an LDS max), and everything from the
// first pop of the sink onwards is discarded.  Same pops, same pushes in the same order, same parents as one pop
// at a time — a search of ~100 pops becomes ~5 rounds.  A round whose pushes would overrun the LDS ring restarts
// the search in augment_wave64 (which can move the queue to global memory); nothing but the search state was touched.
__device__ __forceinline__ int augment_wave64c(Scratch &g, float cap_max, int lane, unsigned long long &adj) {
  const int n = g.n, dst = n - 1;
  const int R = g.ring_n;
  unsigned long long marks = 0;
  int parent = -1;
  if (lane == 0) g.ring[0] = 0;  // the source
  int qh = 0, qt = 1, it = 0;
  bool reached = false;
  while (qt > qh) {
    if (it >= kMaxIter) return RA_E_HUNG_BFS;
    int len = qt - qh;
    len = len > 64 ? 64 : len;
    len = len > kMaxIter - it ? kMaxIter - it : len;
    lds_order();
    const bool valid = lane < len;
    const int v = valid ? (int)g.ring[(qh + lane) & (R - 1)] : 0;
    const unsigned long long sink = __ballot(valid && v == dst);
    const int kstop = sink ? (int)__builtin_ctzll(sink) : 64;  // lane of the first pop of the sink
    const int npop = kstop < len ? kstop + 1 : len;
    unsigned long long pm = lane < npop ? 1ull << v : 0ull;  // -> marks added by the pops at lanes <= this one
    pm = ((unsigned long long)scan_or64((unsigned)(pm >> 32)) << 32) | scan_or64((unsigned)pm);
    const unsigned long long row = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(adj >> 32), v) << 32) |
                                   (unsigned)__shfl((int)(unsigned)adj, v);
    const unsigned long long m = (valid && lane < kstop) ? (row & ~(marks | pm)) : 0ull;
    const int cnt = __popcll(m);
    int off = scan_add64(cnt);
    const int total = __builtin_amdgcn_readlane(off, 63);
    off -= cnt;
    if (qt + total - (qh + npop) > R) return augment_wave64(g, cap_max, lane, adj);
    // one uniform pass over the nodes anybody pushes this round (ascending u, like the serial scan): a pusher appends u
    // behind its earlier pushes; the LAST pushing lane's node becomes u's parent (later pushers overwrite) — a ballot and
    // two scalar reads per node instead of an LDS max per (lane, node) and a shuffle
    unsigned long long U = ((unsigned long long)__builtin_amdgcn_readlane((int)scan_or64((unsigned)(m >> 32)), 63) << 32) |
                           (unsigned)__builtin_amdgcn_readlane((int)scan_or64((unsigned)m), 63);
    int pos = qt + off;
    while (U) {
      const int u = (int)__builtin_ctzll(U);
      U &= U - 1;
      const bool has = (m >> u) & 1ull;
      const unsigned long long bm = __ballot(has);
      if (has) {
        g.ring[pos & (R - 1)] = u;
        ++pos;
      }
      const int pv = __builtin_amdgcn_readlane(v, 63 - (int)__builtin_clzll(bm));
      if (lane == u) parent = pv;
    }
    marks |= readlane64(pm, npop - 1);
    qh += npop;
    it += npop;
    qt += total;
    if (kstop < len) {
      reached = true;
      break;
    }
  }
  if (!reached) return 0;
  return push_path64(g, cap_max, lane, adj, parent);
}

__device__ __forceinline__ int rematch_wave64(Scratch &g, int nx, int ny, float *M, int lane) {
  const int n = g.n, dst = n - 1, nn = n * n;
  for (int k = lane; k < nn; k += 64) {
    g.cap[k] = 0.0f;
    g.flow[k] = 0.0f;
  }
  lds_order();
  for (int k = lane; k < nx * ny; k += 64) {
    const int x = k / ny, y = k - x * ny;
    g.cap[(size_t)(1 + x) * n + (1 + nx + y)] = g.eq[k];
  }
  for (int x = lane; x < nx; x += 64) g.cap[1 + x] = 1.0f;                                // s -> x
  for (int y = lane; y < ny; y += 64) g.cap[(size_t)(1 + nx + y) * n + dst] = 1.0f;      // y -> t
  lds_order();
  float cap_max = -FLT_MAX;
  for (int k = lane; k < nn; k += 64) {
    const float c = g.cap[k];
    g.res[k] = c;
    cap_max = (c > cap_max) ? c : cap_max;
  }
  cap_max = wave_max(cap_max);
  unsigned long long adj = 0;  // row `lane` of the residual graph
  if (lane < n)
    for (int u = 0; u < n; ++u) adj |= (g.cap[(size_t)lane * n + u] > 0) ? (1ull << u) : 0ull;
  for (int it = 0;; ++it) {
    const int r = g.chunked ? augment_wave64c(g, cap_max, lane, adj) : augment_wave64(g, cap_max, lane, adj);
    if (r < 0) return r;
    if (r == 0 || it > kMaxIter) break;
    if (it == kMaxIter) return RA_E_HUNG_FLOW;
  }
  lds_order();
  for (int k = lane; k < nx * ny; k += 64) {
    const int x = k / ny, y = k - x * ny;
    M[k] = g.flow[(size_t)(1 + x) * n + (1 + nx + y)];
  }
  wsync();
  return 0;
}

__device__ __forceinline__ int solve_wave(const float *w, int nx, int ny, float *M, float *cx, float *cy,
                                 void *hot_buf, void *queue_buf, ring_int *ring, int ring_n, int chunked, int lane) {
  Scratch g = carve(hot_buf, queue_buf, nx, ny);
  g.ring = ring;
  g.ring_n = ring_n;
  g.chunked = chunked;
  for (int x = lane; x < nx; x += 64) {
    float top = w[x * ny];
    for (int y = 1; y < ny; ++y) top = (w[x * ny + y] > top) ? w[x * ny + y] : top;
    cx[x] = top;
    g.inS[x] = 0;
  }
  for (int y = lane; y < ny; y += 64) {
    cy[y] = 0.0f;
    g.inT[y] = 0;
  }
  for (int k = lane; k < nx * ny; k += 64) M[k] = 0.0f;
  wsync();
  int cntT = 0;
  bool need_match = true;

  for (int it = 0; it <= kMaxIter; ++it) {
    if (it == kMaxIter) return 1;
    for (int k = lane; k < nx * ny; k += 64) {  // equality graph (hungarian.cc:309-325)
      const int x = k / ny, y = k - x * ny;
      const float slack = cx[x] + cy[y] - w[k];
      const float mag = (slack > 0) ? slack : -slack;
      g.eq[k] = (mag <= RA_HUNG_EPS && (cx[x] > 0 || cy[y] > 0)) ? 1.0f : 0.0f;
    }
    wsync();
    if (need_match) {
      const int r = (g.n <= 64 && g.ring_n >= 64) ? rematch_wave64(g, nx, ny, M, lane)
                                                  : rematch_wave(g, nx, ny, M, lane);
      if (r < 0) return r;
      {  // saturating(): every vertex of the smaller side is matched (hungarian.cc:219-248)
        const bool by_col = nx >= ny;
        const int outer = by_col ? ny : nx, inner = by_col ? nx : ny;
        bool unsat = false;
        for (int a = lane; a < outer; a += 64) {
          float sum = 0;
          for (int b = 0; b < inner; ++b) sum += by_col ? M[b * ny + a] : M[a * ny + b];
          unsat = unsat || (sum == 0);
        }
        if (__ballot(unsat) == 0ull) return 0;
      }
      int first = -1;  // first exposed x seeds S (hungarian.cc:394-403)
      for (int base = 0; base < nx && first < 0; base += 64) {
        const int x = base + lane;
        bool exposed = x < nx;
        if (exposed)
          for (int y = 0; y < ny && exposed; ++y) exposed = !(M[x * ny + y] == 1.0);
        const unsigned long long m = __ballot(exposed);
        if (m) first = base + (int)__builtin_ctzll(m);
      }
      if (first >= 0) {
        for (int a = lane; a < nx; a += 64) g.inS[a] = (a == first) ? 1 : 0;
        for (int b = lane; b < ny; b += 64) g.inT[b] = 0;
        cntT = 0;
      }
      wsync();
    }
    int cntN = 0;  // N(S) in the equality graph
    bool diff = false;
    for (int base = 0; base < ny; base += 64) {
      const int y = base + lane;
      bool inN = false;
      if (y < ny)
        for (int x = 0; x < nx; ++x) inN = inN || (g.inS[x] && g.eq[x * ny + y] > 0);
      if (y < ny) g.inN[y] = inN ? 1 : 0;
      cntN += __popcll(__ballot(inN));
      diff = diff || (inN && !g.inT[y < ny ? y : 0]);
    }
    const bool same = (cntN == cntT) && (__ballot(diff) == 0ull);
    wsync();

    if (same) {  // cover update (hungarian.cc:415-443)
      float a = FLT_MAX;
      for (int k = lane; k < nx * ny; k += 64) {
        const int x = k / ny, y = k - x * ny;
        if (g.inS[x] && !g.inT[y]) {
          const float slack = cx[x] + cy[y] - w[k];
          a = (a < slack) ? a : slack;
        }
      }
      a = wave_min(a);
      if (a < RA_HUNG_EPS) {
        need_match = true;
        continue;
      }
      for (int x = lane; x < nx; x += 64)
        if (g.inS[x]) cx[x] -= a;
      for (int y = lane; y < ny; y += 64)
        if (g.inT[y]) cy[y] += a;
      wsync();
    } else {  // grow the alternating tree (hungarian.cc:444-483)
      for (int j = 0; cntN > cntT && j <= kMaxIter; ++j) {
        if (j == kMaxIter) return RA_E_HUNG_EQUALIZE;
        int y = -1;  // smallest y in N(S) \ T
        for (int base = 0; base < ny && y < 0; base += 64) {
          const int c = base + lane;
          const unsigned long long m = __ballot(c < ny && g.inN[c] && !g.inT[c]);
          if (m) y = base + (int)__builtin_ctzll(m);
        }
        int z = -1;  // its match
        for (int base = 0; base < nx && z < 0 && y >= 0; base += 64) {
          const int x = base + lane;
          const unsigned long long m = __ballot(x < nx && M[x * ny + y] == 1.0);
          if (m) z = base + (int)__builtin_ctzll(m);
        }
        if (z < 0) {
          need_match = true;
          break;
        }
        need_match = false;
        for (int base = 0; base < ny; base += 64) {
          const int v = base + lane;
          const bool add = v < ny && g.eq[z * ny + v] > 0.0 && !g.inN[v];
          if (add) g.inN[v] = 1;
          cntN += __popcll(__ballot(add));
        }
        if (lane == 0) {
          g.inS[z] = 1;
          g.inT[y] = 1;
        }
        ++cntT;
        wsync();
      }
    }
  }
  return 1;
}
#endif  // __HIP_DEVICE_COMPILE__

// One workgroup (one wave) per example running solve_wave().  use_lds: the hot scratch and private
// copies of w / M / cx / cy live in LDS (all 64 lanes stage them in and out).  ring_n: entries of
// the BFS-queue ring that follows them in LDS (a power of two, or 0).
__global__ __launch_bounds__(64) void hungarian_kernel(const float *w, int nx, int ny, float *M,
                                                        float *cx, float *cy, int *status,
                                                        char *ws, size_t ws_per_ex, int use_lds,
                                                        int ring_n, int ring_off, int chunked) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.x, t = threadIdx.x;
  const float *wb = w + (size_t)b * nx * ny;
  float *Mb = M + (size_t)b * nx * ny, *cxb = cx + (size_t)b * nx, *cyb = cy + (size_t)b * ny;
  [[maybe_unused]] char *wsb = ws + (size_t)b * ws_per_ex;  // device pass only
  if (!use_lds) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int rc = solve_wave(wb, nx, ny, Mb, cxb, cyb, wsb, wsb + hot_bytes(nx, ny),
                              (ring_int *)(lds + ring_off), ring_n, chunked, t);
    if (t == 0 && status) status[b] = rc;
#endif
    return;
  }
  float *lw = reinterpret_cast<float *>(lds + hot_bytes(nx, ny));
  float *lM = lw + nx * ny, *lcx = lM + nx * ny, *lcy = lcx + nx;
  for (int e = t; e < nx * ny; e += 64) lw[e] = wb[e];
  __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const int rc = solve_wave(lw, nx, ny, lM, lcx, lcy, lds, wsb + hot_bytes(nx, ny),
                              (ring_int *)(lds + ring_off), ring_n, chunked, t);
    if (t == 0 && status) status[b] = rc;
  }
#endif
  __syncthreads();
  for (int e = t; e < nx * ny; e += 64) Mb[e] = lM[e];
  for (int e = t; e < nx; e += 64) cxb[e] = lcx[e];
  for (int e = t; e < ny; e += 64) cyb[e] = lcy[e];
}

inline int merge(int worst, int rc) {
  if (rc < 0) return (worst >= 0 || rc < worst) ? rc : worst;
  return (worst >= 0 && rc > worst) ? rc : worst;
}

}  // namespace hung
}  // namespace ra

extern "C" int ra_hungarian_f32(const float *weights, int B, int N, int M, float *matching,
                                float *cover_x, float *cover_y) {
  if (!weights || !matching || !cover_x || !cover_y || B < 0 || N <= 0 || M <= 0)
    return ra::fail(RA_E_INVALID, "ra_hungarian_f32: bad argument");
  std::vector<char> scratch(ra::hung::scratch_bytes(N, M));
  int worst = 0;
  for (int b = 0; b < B; ++b) {
    const int rc = ra::hung::solve(weights + (size_t)b * N * M, N, M, matching + (size_t)b * N * M,
                                   cover_x + (size_t)b * N, cover_y + (size_t)b * M, scratch.data(),
                                   scratch.data() + ra::hung::hot_bytes(N, M));
    worst = ra::hung::merge(worst, rc);
  }
  if (worst) ra::set_error("ra_hungarian_f32: status %d", worst);
  return worst;
}

extern "C" size_t ra_hungarian_dev_workspace_bytes(int B, int N, int M) {
  if (B <= 0 || N <= 0 || M <= 0) return 0;
  return (size_t)B * ra::hung::scratch_bytes(N, M);
}

extern "C" int ra_hungarian_f32_dev(const float *weights, int B, int N, int M, float *matching,
                                    float *cover_x, float *cover_y, int *status_dev, void *ws,
                                    size_t ws_bytes, void *stream) {
  if (!weights || !matching || !cover_x || !cover_y || !ws || B < 0 || N <= 0 || M <= 0)
    return ra::fail(RA_E_INVALID, "ra_hungarian_f32_dev: bad argument");
  if (B == 0) return 0;
  const size_t per = ra::hung::scratch_bytes(N, M);
  if (ws_bytes < per * (size_t)B)
    return ra::fail(RA_E_WORKSPACE, "ra_hungarian_f32_dev: workspace %zu < %zu", ws_bytes,
                    per * (size_t)B);
  const size_t lds = ra::hung::hot_bytes(N, M) + ((size_t)2 * N * M + N + M) * sizeof(float);
  const int use_lds = lds <= 150 * 1024;
  // BFS-queue ring after the staged data: the largest power of two that fits, at most 8192 entries
  // (RA_HUNG_RING=0 keeps the queue in global memory)
  const size_t ring_off = use_lds ? (lds + 15) / 16 * 16 : 0;
  const char *ring_env = getenv("RA_HUNG_RING");  // read per call: the tests force the spill path
  const int ring_cap = ring_env ? atoi(ring_env) : 8192;
  int ring_n = 0;
  for (int r = 64; r <= ring_cap && ring_off + (size_t)(r + 64) * 4 <= 150 * 1024; r *= 2) ring_n = r;  // + 64: augment_wave64c's table
  const char *bfs_env = getenv("RA_HUNG_BFS");  // 0: one pop at a time (tests run both forms)
  const int chunked = (bfs_env ? atoi(bfs_env) : 1) && ring_n >= 64;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ra::hung::hungarian_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(ra::hung::hungarian_kernel, dim3(B), dim3(64), ring_off + (size_t)(ring_n + 64) * 4,
                     ra::as_stream(stream), weights, N, M, matching, cover_x, cover_y, status_dev,
                     reinterpret_cast<char *>(ws), per, use_lds, ring_n, (int)ring_off, chunked);
  return ra::launch_status("ra_hungarian_f32_dev");
}
