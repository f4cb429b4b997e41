// Where the time of K1s (conv_split_kernel<32, 2, NB>: cfg2's L5 32 -> 32 at 128 x 128 and L6 32 -> 64 at 64 x 64, pool 2) goes:
// builds csrc/ra_conv_split.hip with -DRA_PROBES (wave 0 of every workgroup accumulates the shader-clock time between points of
// its tile loop) and prints the share of a workgroup's life per phase next to the HIP-graph launch time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRA_PROBES -Iinclude -Irec-attend-public_amd/csrc tools/split_probe.hip -o tools/bin/split_probe
// usage: split_probe B [H = 128] [Cout = 32]
#include "../rec-attend-public_amd/csrc/ra_conv_split.hip"

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
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 128, W = H, Cin = 32, Cout = argc > 3 ? atoi(argv[3]) : 32;
  float *x, *y, *sc, *sh;
  unsigned short *wp;
  CK(hipMalloc(&x, (size_t)B * H * W * Cin * 4));
  CK(hipMalloc(&y, (size_t)B * (H / 2) * (W / 2) * Cout * 4));
  CK(hipMalloc(&sc, 256));
  CK(hipMalloc(&sh, 256));
  std::vector<float> hx((size_t)B * H * W * Cin), hw(9 * Cin * Cout), one(64, 1.f), zero(64, 0.1f);
  for (auto &v : hx) v = (float)(rand() % 1000) * 1e-3f;
  for (auto &v : hw) v = 0.1f * ((float)(rand() % 1000) * 1e-3f - 0.5f);
  std::vector<unsigned short> hp(ra_conv_split_packed_halfs(Cin, Cout));
  if (ra_conv_split_pack_weights(hw.data(), Cin, Cout, hp.data())) return 1;
  CK(hipMalloc(&wp, hp.size() * 2));
  CK(hipMemcpy(wp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, one.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, zero.data(), 256, hipMemcpyHostToDevice));
  long long *probe;
  CK(hipMalloc(&probe, (size_t)4096 * 8 * 8));
  CK(hipMemset(probe, 0, (size_t)4096 * 8 * 8));
#ifdef RA_PROBES
  long long *nul = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::csplit::ra_probes_buf), &nul, sizeof(nul)));
#endif
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto launch = [&] {
    const int rc = ra_conv_split_f32(x, B, H, W, Cin, wp, sc, sh, Cout, 1, 2, y, st);
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
  printf("B=%d %dx%d 32 -> %d conv_split_kernel: %.2f us/launch in a HIP graph (8 copies x 20 replays)\n", B, H, W, Cout, 1e3f * gus / 160);
#ifdef RA_PROBES
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::csplit::ra_probes_buf), &probe, sizeof(probe)));
  launch();
  CK(hipStreamSynchronize(st));
  const int nwg = 256;
  std::vector<long long> h((size_t)nwg * 8);
  CK(hipMemcpy(h.data(), probe, h.size() * 8, hipMemcpyDeviceToHost));
  const char *names[6] = {"prologue (constants, filter copy issued)", "top barrier", "window arrived + split + LDS stores", "staging barrier",
                          "next window requested + k-loop", "epilogue"};
  double tot = 0, wall = 0, sum[6] = {0, 0, 0, 0, 0, 0};
  int live = 0;
  for (int w = 0; w < nwg; ++w) {
    if (!h[(size_t)w * 8 + 7]) continue;
    ++live;
    for (int k = 0; k < 6; ++k) sum[k] += h[(size_t)w * 8 + k];
    wall += h[(size_t)w * 8 + 7] * 0.01;
  }
  for (int k = 0; k < 6; ++k) tot += sum[k];
  printf("%d workgroups, mean life %.2f us (100 MHz clock)\n", live, wall / live);
  for (int k = 0; k < 6; ++k) printf("  %-42s %5.1f %%   (%.2f us)\n", names[k], 100.0 * sum[k] / tot, sum[k] / tot * wall / live);
#endif
  return 0;
}
