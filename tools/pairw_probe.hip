// Where the time of the second controller-CNN launch (conv_pair_wino_mfma<8, SPLIT>: L2 direct on the bf16 pipe + L3 Winograd, at
// 256 x 256) goes: builds csrc/ra_conv_wino.hip with -DRA_PROBEW (wave 0 of every workgroup accumulates the shader-clock time
// between points of its tile loop) and prints the share of a workgroup's life per phase next to the HIP-graph launch time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRA_PROBEW -Iinclude -Irec-attend-public_amd/csrc tools/pairw_probe.hip -o tools/bin/pairw_probe
#include "../rec-attend-public_amd/csrc/ra_conv_wino.hip"

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
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = 256, W = 256;
  float *x, *y, *wA, *wB, *sc, *sh;
  CK(hipMalloc(&x, (size_t)B * H * W * 8 * 4));
  CK(hipMalloc(&y, (size_t)B * (H / 2) * (W / 2) * 16 * 4));
  CK(hipMalloc(&wA, 9 * 8 * 16 * 4));
  CK(hipMalloc(&wB, 16 * 4 * 4 * 64 * 4 * 4));
  CK(hipMalloc(&sc, 64));
  CK(hipMalloc(&sh, 64));
  std::vector<float> hx((size_t)B * H * W * 8), hw(16 * 4 * 4 * 64 * 4), one(16, 1.f), zero(16, 0.1f);
  for (auto &v : hx) v = (float)(rand() % 1000) * 1e-3f;
  for (auto &v : hw) v = 0.2f * ((float)(rand() % 1000) * 1e-3f - 0.5f);
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wA, hw.data(), 9 * 8 * 16 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wB, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, one.data(), 64, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, zero.data(), 64, hipMemcpyHostToDevice));
  long long *probe;
  CK(hipMalloc(&probe, (size_t)4096 * 8 * 8));
  CK(hipMemset(probe, 0, (size_t)4096 * 8 * 8));
#ifdef RA_PROBEW
  long long *nul = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::wino::ra_probew_buf), &nul, sizeof(nul)));
#endif
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto launch = [&] {
    const int rc = ra_conv_pair_wino_f32(x, B, H, W, wA, sc, sh, 1, wB, sc, sh, 1, y, st);
    if (rc) {
      printf("rc=%d\n", rc);
      exit(1);
    }
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 8; ++i) launch();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float gus;
  CK(hipEventElapsedTime(&gus, e0, e1));
  printf("B=%d conv_pair_wino_mfma<8, SPLIT>: %.2f us/launch in a HIP graph (8 copies x 20 replays)\n", B, 1e3f * gus / 160);
#ifdef RA_PROBEW
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::wino::ra_probew_buf), &probe, sizeof(probe)));
  launch();
  CK(hipStreamSynchronize(st));
  const int nwg = 1024;
  std::vector<long long> h((size_t)nwg * 8);
  CK(hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost));
  const char *names[7] = {"top barrier", "stage (3 bf16 tiles) + barrier", "layer A MFMAs", "layer A epilogue -> tin", "barrier",
                          "layer B transform + MFMAs", "layer B exchange + output"};
  double tot = 0, wall = 0, sum[7] = {0, 0, 0, 0, 0, 0, 0};
  int live = 0;
  for (int w = 0; w < nwg; ++w) {
    if (!h[(size_t)w * 8 + 7]) continue;
    ++live;
    for (int k = 0; k < 7; ++k) sum[k] += h[(size_t)w * 8 + k];
    wall += h[(size_t)w * 8 + 7] * 0.01;
  }
  for (int k = 0; k < 7; ++k) tot += sum[k];
  printf("%d workgroups, mean life in the tile loop %.2f us (100 MHz clock)\n", live, wall / live);
  for (int k = 0; k < 7; ++k) printf("  %-32s %5.1f %%   (%.2f us)\n", names[k], 100.0 * sum[k] / tot, sum[k] / tot * wall / live);
#endif
  return 0;
}
