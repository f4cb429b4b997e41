// Can the workgroups of ONE XCD synchronise through their shared L2 cheaply?  Launches 8 * NW workgroups; those with
// blockIdx.x % 8 == X take part (the dispatcher deals workgroups to the 8 XCDs round robin), report their XCC_ID, publish a
// record with plain stores, meet at a barrier built from L2 atomics WITHOUT the cross-XCD (sc1) bit, and read every
// participant's record back with sc0 loads (past the CU's L1, served by the XCD's L2).  Prints: the XCC ids seen, whether
// every record arrived intact, and the barrier's cost by the 100 MHz wall clock.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_barrier_probe.hip -o tools/bin/xcd_barrier_probe
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

__device__ inline int l2_atomic_add(int *p, int v) {  // returns the old value; executes in this XCD's L2 (glc, no sc1)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 4, 0x00020000);
  return __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(v, r, 0, 0, 1);
}
__device__ inline int l2_load(const int *p, int idx, int bytes) {  // sc0: misses the CU's L1
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(p), 0, bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4, 0, (int)0x80000001u);  // bit 31: volatile (not hoisted out of the spin loop)
}

__global__ void probe(int X, int NW, int rounds, int *counter, int *rec, int *xcc, long long *ticks, int *bad) {
  if ((int)blockIdx.x % 8 != X) return;
  const int w = blockIdx.x / 8, tid = threadIdx.x;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (tid == 0) xcc[w] = (int)(id & 0xf);
  long long t0 = 0, acc = 0;
  for (int r = 0; r < rounds; ++r) {
    // publish
    rec[(r & 1) * NW * 64 + w * 64 + (tid & 63)] = r * 1000 + w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      t0 = wall_clock64();
      l2_atomic_add(counter, 1);
      int spins = 0;
      while (l2_load(counter, 0, 4) < (r + 1) * NW && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
      if (spins >= (1 << 22)) atomicAdd(bad, 1 << 20);
      acc += wall_clock64() - t0;
    }
    __syncthreads();
    // read everybody's record
    int wrong = 0;
    for (int k = tid; k < NW * 64; k += blockDim.x)
      if (l2_load(rec, (r & 1) * NW * 64 + k, 2 * NW * 64 * 4) != r * 1000 + k / 64) ++wrong;
    if (wrong) atomicAdd(bad, wrong);
  }
  if (tid == 0) ticks[w] = acc;
}

int main(int argc, char **argv) {
  const int NW = argc > 1 ? atoi(argv[1]) : 48, rounds = 20;
  int *counter, *rec, *xcc, *bad;
  long long *ticks;
  CK(hipMalloc(&counter, 64));
  CK(hipMalloc(&rec, 2 * NW * 64 * 4));
  CK(hipMalloc(&xcc, NW * 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&ticks, NW * 8));
  for (int X = 0; X < 8; X += 3) {
    CK(hipMemset(counter, 0, 64));
    CK(hipMemset(bad, 0, 4));
    CK(hipMemset(rec, 0xff, 2 * NW * 64 * 4));
    hipLaunchKernelGGL(probe, dim3(8 * NW), dim3(256), 0, 0, X, NW, rounds, counter, rec, xcc, ticks, bad);
    CK(hipDeviceSynchronize());
    std::vector<int> hx(NW);
    std::vector<long long> ht(NW);
    int hb;
    CK(hipMemcpy(hx.data(), xcc, NW * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), ticks, NW * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    int seen = 0;
    for (int v : hx) seen |= 1 << v;
    double mean = 0;
    for (long long t : ht) mean += t * 0.01 / rounds;
    printf("X=%d NW=%d: XCC ids seen (bitmask) 0x%x, wrong / timed-out records %d, barrier wait %.2f us mean per round\n", X, NW, seen, hb,
           mean / NW);
  }
  return 0;
}
