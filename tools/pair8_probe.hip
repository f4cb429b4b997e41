// Where the time of the first controller-CNN launch (conv_pair8_mfma<4, CACHED>: L0 + L1 at full resolution) goes:
// builds csrc/ra_conv_pair.hip with -DRA_PROBE8 (wave 0 of every workgroup accumulates the shader-clock time between
// a few points of its tile loop) and prints, per phase, the share of a workgroup's time, plus the HIP-event duration
// of the launch.  Phases: 0 = next canvas window staged to LDS + barrier, 1 = layer A's cached sums arrived, 2 = phase A
// (3 MFMAs per group, ReLU, LDS tile), 3 = barrier, 4 = phase B (24 MFMAs per group, pool, store).
// Without -DRA_PROBE8 it only times the launch (HIP graph of 8 copies, and eager).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DRA_PROBE8 [-DRA_P8_NOBAR]] -Iinclude -Irec-attend-public_amd/csrc \
//         tools/pair8_probe.hip -o tools/bin/pair8_probe
#include "../rec-attend-public_amd/csrc/ra_conv_pair.hip"

#include <algorithm>
#include <vector>

namespace ra {
void set_error(const char *, ...) {}
unsigned *take_ticket_slots(int, int) { return nullptr; }  // the static tile walk (ra_common.h)
}  // namespace ra
extern "C" int ra_conv_cout_padded(int Cout) { return (Cout + 15) / 16 * 16; }

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e_ = (x);                                               \
    if (e_ != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                         \
    }                                                                  \
  } while (0)

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = 512, W = 512;
  float *img, *canvas, *cache, *y, *wA, *wB, *sc, *sh;
  CK(hipMalloc(&img, (size_t)B * H * W * 16));
  CK(hipMalloc(&canvas, (size_t)B * H * W * 4));
  CK(hipMalloc(&y, (size_t)B * (H / 2) * (W / 2) * 8 * 4));
  const size_t cf = ra_conv_first_cache_floats(B, H, W);
  CK(hipMalloc(&cache, cf * 4));
  CK(hipMemset(cache, 0, cf * 4));
  CK(hipMalloc(&wA, 9 * 4 * 16 * 4));
  CK(hipMalloc(&wB, 9 * 8 * 16 * 4));
  CK(hipMalloc(&sc, 64));
  CK(hipMalloc(&sh, 64));
  std::vector<float> hi((size_t)B * H * W * 4), hc((size_t)B * H * W), hw(9 * 8 * 16), one(16, 1.f), zero(16, 0.1f);
  for (auto &v : hi) v = (float)rand() / RAND_MAX;
  for (auto &v : hc) v = (float)rand() / RAND_MAX;
  for (auto &v : hw) v = 0.2f * ((float)rand() / RAND_MAX - 0.5f);
  CK(hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(canvas, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wA, hw.data(), 9 * 4 * 16 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wB, hw.data(), 9 * 8 * 16 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, one.data(), 64, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, zero.data(), 64, hipMemcpyHostToDevice));
  long long *probe;
  const int nwg = getenv("RA_PAIR8_WGS") ? atoi(getenv("RA_PAIR8_WGS")) : 768;
  CK(hipMalloc(&probe, (size_t)4096 * 8 * 8));
  CK(hipMemset(probe, 0, (size_t)4096 * 8 * 8));
#ifdef RA_PROBE8
  long long *nul = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::cpair::ra_probe8_buf), &nul, sizeof(nul)));
#endif
  int rc = ra_conv_pair_fill_cache_f32(img, canvas, 3, B, H, W, wA, sc, sh, 1, wB, sc, sh, 8, 1, cache, y, nullptr);
  CK(hipDeviceSynchronize());
  if (rc) {
    printf("fill_cache rc=%d\n", rc);
    return 1;
  }
  auto launch = [&] { ra_conv_pair_cached_f32(cache, canvas, 3, B, H, W, wA, sc, sh, 1, wB, sc, sh, 8, 1, y, nullptr); };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // the same launch, 8 copies in a HIP graph (what bench.py's roofline times)
  float gus = 0.f;
  {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto launch_s = [&] { ra_conv_pair_cached_f32(cache, canvas, 3, B, H, W, wA, sc, sh, 1, wB, sc, sh, 8, 1, y, st); };
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 8; ++i) launch_s();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&gus, e0, e1));
    gus = 1e3f * gus / 160;
  }
  printf("B=%d conv_pair8_mfma<4,true>: %.2f us/launch in a HIP graph (8 copies x 20 replays), %.2f us eager back to back"
#ifdef RA_PROBE8
         " (probe code compiled in)"
#endif
         "\n", B, gus, 1e3 * ms / 20);
#ifndef RA_PROBE8
  return 0;
#else
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::cpair::ra_probe8_buf), &probe, sizeof(probe)));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<long long> h((size_t)nwg * 8);
  CK(hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost));
  const char *names[5] = {"stage+barrier", "cache arrived", "phase A", "barrier", "phase B+store"};
  double tot = 0, wall = 0;
  double sum[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < nwg; ++w) {
    for (int k = 0; k < 5; ++k) sum[k] += h[(size_t)w * 8 + k];
    wall += h[(size_t)w * 8 + 7] * 0.01;
  }
  for (int k = 0; k < 5; ++k) tot += sum[k];
  printf("mean workgroup life %.2f us (100 MHz clock), shader-clock ticks in the tile loop per workgroup %.0f\n", wall / nwg, tot / nwg);
  for (int k = 0; k < 5; ++k) printf("  %-14s %5.1f %%   (%.2f us of the workgroup's life)\n", names[k], 100.0 * sum[k] / tot, sum[k] / tot * wall / nwg);
  return 0;
#endif
}
