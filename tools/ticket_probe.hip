// What does a dynamic tile ticket cost?  A persistent grid of G workgroups draws N tickets (and nothing else) from
//   mode 0: ONE counter with agent-scope atomics (sc1: performed past the XCD's L2, coherent across the 8 XCDs)
//   mode 1: one counter per XCD (keyed by HW_REG_XCC_ID) with L2-local atomics (glc, no sc1) — each XCD draws from its own pool
// keeping two tickets in flight per workgroup (the form the conv kernels use: the ticket of the tile after next is requested
// while a tile is computed).  `work` = ns of s_sleep per tile, to see whether the draw hides behind a tile's compute.
// Prints the launch time (HIP events over 20 launches in a graph-free loop minus nothing: compare modes), the tickets handed
// out (must be N: every tile exactly once) and the round-trip latency of a draw by the 100 MHz clock.
//   hipcc -O3 --offload-arch=gfx950 tools/ticket_probe.hip -o tools/bin/ticket_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

__device__ inline unsigned l2_atomic_inc(unsigned *p) {  // old value; executes in this XCD's L2
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 4, 0x00020000);
  return (unsigned)__builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r, 0, 0, 1);
}
__device__ inline unsigned dev_atomic_inc(unsigned *p) {
  return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void draw(unsigned *cnt, int ntiles, int work_ticks, unsigned *taken, unsigned *dup, long long *lat) {
  __shared__ unsigned sh;
  const int tid = threadIdx.x;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const int x = (int)(id & 7);
  const int chunk = (ntiles + 7) >> 3;
  const int base = MODE == 1 ? x * chunk : 0;
  const int lim = MODE == 1 ? (base + chunk < ntiles ? chunk : ntiles - base) : ntiles;
  unsigned *c = MODE == 1 ? cnt + 16 * x : cnt;
  unsigned t0 = 0, t1 = 0;
  long long w0 = 0;
  if (tid == 0) {
    w0 = wall_clock64();
    t0 = MODE == 1 ? l2_atomic_inc(c) : dev_atomic_inc(c);
    t1 = MODE == 1 ? l2_atomic_inc(c) : dev_atomic_inc(c);
    sh = t0;
    lat[blockIdx.x] = wall_clock64() - w0;
  }
  __syncthreads();
  unsigned cur = sh;
  __syncthreads();
  if (tid == 0) sh = t1;
  __syncthreads();
  unsigned nxt = sh;
  int mine = 0;
  if (MODE == 2) {  // the static walk: tile = blockIdx.x, + gridDim.x, ...
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      if (tid == 0) __hip_atomic_fetch_add(&dup[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long s = wall_clock64();
      while (wall_clock64() - s < work_ticks) __builtin_amdgcn_s_sleep(2);
      ++mine;
      __syncthreads();
      __syncthreads();
    }
    if (tid == 0 && mine) atomicAdd(taken, (unsigned)mine);
    return;
  }
  while ((int)cur < lim) {
    unsigned p = 0;
    if (tid == 0) p = MODE == 1 ? l2_atomic_inc(c) : dev_atomic_inc(c);
    // the "tile"
    if (tid == 0) __hip_atomic_fetch_add(&dup[base + cur], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // no return used: fire and forget
    const long long s = wall_clock64();
    while (wall_clock64() - s < work_ticks) __builtin_amdgcn_s_sleep(2);
    ++mine;
    __syncthreads();
    if (tid == 0) sh = p;
    __syncthreads();
    cur = nxt;
    nxt = sh;
  }
  if (tid == 0 && mine) atomicAdd(taken, (unsigned)mine);
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 768, N = argc > 2 ? atoi(argv[2]) : 8192;
  unsigned *cnt, *taken, *dup;
  long long *lat;
  CK(hipMalloc(&cnt, 16 * 8 * 4));
  CK(hipMalloc(&taken, 8));
  CK(hipMalloc(&dup, (N + 8) * 4));
  CK(hipMalloc(&lat, G * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int work_us : {0, 2, 6}) {
    for (int mode = 0; mode < 3; ++mode) {
      float tot = 0;
      unsigned ht[2] = {0, 0};
      const int reps = 20;
      for (int r = 0; r < reps; ++r) {
        CK(hipMemset(cnt, 0, 16 * 8 * 4));
        CK(hipMemset(taken, 0, 8));
        CK(hipMemset(dup, 0, (N + 8) * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0)
          hipLaunchKernelGGL(draw<0>, dim3(G), dim3(256), 0, 0, cnt, N, work_us * 100, taken, dup, lat);
        else if (mode == 2)
          hipLaunchKernelGGL(draw<2>, dim3(G), dim3(256), 0, 0, cnt, N, work_us * 100, taken, dup, lat);
        else
          hipLaunchKernelGGL(draw<1>, dim3(G), dim3(256), 0, 0, cnt, N, work_us * 100, taken, dup, lat);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
        CK(hipMemcpy(ht, taken, 8, hipMemcpyDeviceToHost));
      }
      std::vector<unsigned> hd(N);
      CK(hipMemcpy(hd.data(), dup, N * 4, hipMemcpyDeviceToHost));
      ht[1] = 0;
      for (unsigned v : hd) ht[1] += v != 1;
      std::vector<long long> hl(G);
      CK(hipMemcpy(hl.data(), lat, G * 8, hipMemcpyDeviceToHost));
      double ml = 0;
      for (long long v : hl) ml += v * 0.01 / G;
      const double ideal = (double)N / G * work_us;
      printf("work %d us/tile  mode %d (%s): %.1f us per launch (tiles / workgroups x work = %.1f), tiles done %u of %d, tiles not done exactly once %u, first two draws %.2f us\n",
             work_us, mode, mode == 2 ? "static walk" : mode ? "per-XCD pools, L2-local atomics" : "one counter, agent-scope atomics", 1e3 * tot / reps, ideal, ht[0], N, ht[1], ml);
    }
  }
  return 0;
}
