// Where the time of the attention-resample kernels (K3 extract, K5 paste) goes: builds csrc/ra_attn_direct.hip with -DRA_PROBE
// (thread 0 of every workgroup stamps the 100 MHz wall clock at a few points) and prints, per kernel
// form, the phase times relative to the first workgroup's start, plus the HIP-event duration.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRA_PROBE -Iinclude -Irec-attend-public_amd/csrc tools/attn_probe.hip -o tools/bin/attn_probe
#include "../rec-attend-public_amd/csrc/ra_attn_direct.hip"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

namespace ra {
void set_error(const char *, ...) {}
int tail_prio(int) { return 0; }
}  // namespace ra

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

static long long *g_probe;
static size_t g_probe_n = 1 << 20;

template <typename F>
static void run(const char *name, int nwg, int nslots, F launch) {
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
  CK(hipMemset(g_probe, 0, g_probe_n * 8));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<long long> h((size_t)nwg * 8);
  CK(hipMemcpy(h.data(), g_probe, h.size() * 8, hipMemcpyDeviceToHost));
  long long t0 = -1;
  for (int w = 0; w < nwg; ++w)
    if (h[(size_t)w * 8] && (t0 < 0 || h[(size_t)w * 8] < t0)) t0 = h[(size_t)w * 8];
  printf("%-34s %6.2f us/launch (back-to-back eager) |", name, 1e3 * ms / 20);
  for (int s = 0; s < nslots; ++s) {
    std::vector<double> v;
    for (int w = 0; w < nwg; ++w)
      if (h[(size_t)w * 8 + s]) v.push_back((h[(size_t)w * 8 + s] - t0) * 0.01);
    if (v.empty()) {
      printf(" s%d: -", s);
      continue;
    }
    std::sort(v.begin(), v.end());
    printf(" s%d[n=%zu]: %.2f/%.2f/%.2f", s, v.size(), v.front(), v[v.size() / 2], v.back());
  }
  printf("  (us since first start: min/median/max)\n");
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = 512, W = 512, Fh = 48, Fw = 48, Ci = 4, Cp = 4;
  const float lgvar = argc > 2 ? atof(argv[2]) : 1.32f, size = argc > 3 ? atof(argv[3]) : 179.2f;
  float *img, *canvas, *attn, *patch, *ypatch, *yout;
  CK(hipMalloc(&img, (size_t)B * H * W * Ci * 4));
  CK(hipMalloc(&canvas, (size_t)B * H * W * 4));
  CK(hipMalloc(&attn, (size_t)B * RA_ATTN_STRIDE * 4));
  CK(hipMalloc(&patch, (size_t)B * Fh * Fw * Cp * 4));
  CK(hipMalloc(&ypatch, (size_t)B * Fh * Fw * 4));
  CK(hipMalloc(&yout, (size_t)B * 2 * H * W * 4));
  CK(hipMalloc(&g_probe, g_probe_n * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ra::attnd::ra_probe_buf), &g_probe, sizeof(g_probe)));
  std::vector<float> hi((size_t)B * H * W * Ci), hc((size_t)B * H * W, 0.2f), ha((size_t)B * RA_ATTN_STRIDE, 0.f),
      hp((size_t)B * Fh * Fw);
  for (auto &v : hi) v = (float)rand() / RAND_MAX;
  for (auto &v : hp) v = (float)rand() / RAND_MAX - 0.5f;
  for (int b = 0; b < B; ++b) {
    float *r = &ha[(size_t)b * RA_ATTN_STRIDE];
    r[0] = 255.f + b;
    r[1] = 256.f - b;
    r[2] = r[3] = size;
    r[4] = r[5] = lgvar;
    r[6] = 1.f;
    r[7] = 1.f;
    r[8] = 2.f;
  }
  CK(hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(canvas, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(attn, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ypatch, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  printf("B=%d lg_var=%.2f size=%.1f\n", B, lgvar, size);
  using namespace ra::attnd;
  {
    const int n_items = Fh * (Cp / 4) * B, chunk = (n_items + 7) / 8;
    run("extract_rows<4>", 8 * chunk, 4, [&] {
      hipLaunchKernelGGL((extract_rows_kernel<4>), dim3(8 * chunk), dim3(256), 0, 0, img, Ci, 0, canvas, 3, attn, H, W, Fh,
                         Fw, Cp, 1, patch, n_items, chunk, 0);
    });
  }
  for (int flags : {3, 2, 0}) {
    char nm[64];
    snprintf(nm, sizeof nm, "paste_direct<0> flags=%d (general form)", flags);
    run(nm, H / 4 * B, 1, [&] {
      hipLaunchKernelGGL(paste_direct_kernel<0>, dim3(H / 4, B), dim3(256), 4 * Fw * 4, 0, ypatch, 1, 0, attn, H, W, Fh, Fw,
                         -5.0f, 0, canvas, (float *)nullptr, 0, -1, yout, (size_t)2 * H * W, flags, 4, ScoreArgs{});
    });
    snprintf(nm, sizeof nm, "paste_win<0,4> flags=%d", flags);
    size_t lds = (size_t)4 * 256 * 16 + (size_t)(4 * Fw + Fh * Fw + 16) * 4;
    run(nm, H / 4 * B, 5, [&] {
      hipLaunchKernelGGL((paste_win_kernel<0, 4>), dim3(H / 4, B), dim3(256), lds, 0, ypatch, attn, H, W, Fh, Fw, -5.0f, 0,
                         canvas, yout, (size_t)2 * H * W, flags, ScoreArgs{}, 0);
    });
    snprintf(nm, sizeof nm, "paste_win<0,8> flags=%d", flags);
    lds = (size_t)8 * 256 * 16 + (size_t)(8 * Fw + Fh * Fw + 16) * 4;
    run(nm, H / 8 * B, 5, [&] {
      hipLaunchKernelGGL((paste_win_kernel<0, 8>), dim3(H / 8, B), dim3(256), lds, 0, ypatch, attn, H, W, Fh, Fw, -5.0f, 0,
                         canvas, yout, (size_t)2 * H * W, flags, ScoreArgs{}, 0);
    });
  }
  return 0;
}
