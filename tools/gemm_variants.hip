// A/B of the library's LDS-DMA GEMM template instantiations on one shape, interleaved in ONE process (diagnostic).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Isugar_amd/csrc tools/gemm_variants.hip -o tools/gemm_variants
// Run:   tools/gemm_variants [R N K]
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "../sugar_amd/csrc/conv1x1.hip"

namespace sugar {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
int device_cu_count(int* n) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SUGAR_EHIP;
  return hipDeviceGetAttribute(n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess ? SUGAR_OK : SUGAR_EHIP;
}
int64_t tune_get(int) { return 0; }   // (library tunables: all at their "off" value here)
}  // namespace sugar

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace sugar;

static void* g_flush = nullptr;   // cold mode: a 1 GB memset between launches evicts L2 and the Infinity Cache

template <typename F>
float timeit(F f, int iters) {
  if (g_flush) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0;
    for (int i = 0; i < iters; ++i) {
      CK(hipMemsetAsync(g_flush, i, 1u << 30, 0));
      CK(hipEventRecord(e0));
      f();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      tot += ms;
    }
    return tot / iters;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 77056, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  std::vector<uint16_t> hx((size_t)R * K), hw((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return host_f32_to_bf16(((s >> 8) & 0xffff) / 32768.0f - 1.0f); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  bf16 *x, *w, *y, *yref;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2));
  CK(hipMalloc(&y, (size_t)R * N * 2)); CK(hipMalloc(&yref, (size_t)R * N * 2));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  Gemm1x1Args a;
  a.x = x; a.w = w; a.y = yref; a.y2 = nullptr; a.scale = nullptr; a.shift = nullptr;
  a.ldx = K; a.ldw = K; a.ldy = N; a.ldy2 = 0; a.y2_from = 0; a.R = R; a.cout = N; a.K = K; a.flags = EPI_RELU;
  const double flop = 2.0 * R * N * K;
  a.n_co_tiles = N / 256;
  launch_variant<256, 256, 2>(a, 0);
  CK(hipDeviceSynchronize());
  a.y = y;
  std::vector<uint16_t> h0((size_t)R * N), h1((size_t)R * N);
  CK(hipMemcpy(h0.data(), yref, h0.size() * 2, hipMemcpyDeviceToHost));
  auto check = [&](const char* name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1.data(), y, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    printf("  %-18s mismatches vs <256,256,2>: %zu\n", name, bad);
    CK(hipMemset(y, 0, h1.size() * 2));
  };
  a.n_co_tiles = N / 256; launch_split(a, 0); check("split");
  a.n_co_tiles = N / 128; launch_variant<128, 128, 2>(a, 0); check("<128,128,2>");
  a.n_co_tiles = N / 128; launch_variant<128, 256, 2>(a, 0); check("<128,256,2>");
  a.n_co_tiles = N / 256; launch_variant<256, 128, 2>(a, 0); check("<256,128,2>");
  a.n_co_tiles = N / 256; launch_variant<256, 256, 2, true>(a, 0); check("<256,256,2> pp");
  a.n_co_tiles = N / 256; launch_split<true>(a, 0); check("split pp");
  a.n_co_tiles = N / 256; launch_variant<256, 256, 2, 2>(a, 0); check("<256,256,2> pp2");
  a.n_co_tiles = N / 256; launch_split<2>(a, 0); check("split pp2");
  a.n_co_tiles = N / 128; launch_variant<128, 128, 2, 2>(a, 0); check("<128,128,2> pp2");
  a.n_co_tiles = N / 128; launch_variant<128, 128, 2, true>(a, 0); check("<128,128,2> pp");
#ifdef GEMM_STAMPS
  {
    long long* stamps;
    CK(hipMalloc(&stamps, 2 * 64 * 8));
    CK(hipMemset(stamps, 0, 2 * 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sugar::g_gemm_stamps), &stamps, sizeof(stamps)));
    a.n_co_tiles = N / 256;
    launch_variant<256, 256, 2, GEMM_STAMPS>(a, 0);
    CK(hipDeviceSynchronize());
    long long h[128];
    CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    long long* null_ptr = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sugar::g_gemm_stamps), &null_ptr, sizeof(null_ptr)));
    // 4 stamps (A read opens, B fragments in, C MFMA opens, D MFMAs issued) per half k-tile, k-tiles 4..6
    for (int w = 0; w < 2; ++w) {
      printf("stamps wave %d (ticks since first; A B C D per half k-tile):\n", w * 4);
      for (int i = 0; i < 24; ++i) printf("%s%6lld", i % 4 == 0 ? "\n   " : " ", h[w * 64 + i] - h[0]);
      printf("\n");
    }
  }
#endif
  for (int round = 0; round < 4; ++round) {
    if (round == 2) CK(hipMalloc(&g_flush, 1u << 30));
    printf("round %d  R=%d N=%d K=%d %s\n", round, R, N, K, g_flush ? "COLD (caches flushed before every launch)" : "warm");
    float t;
    a.n_co_tiles = N / 256;
    t = timeit([&] { launch_variant<256, 256, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<256,256,2>", t, flop / t / 1e9);
    t = timeit([&] { launch_split(a, 0); }, 10);                printf("  %-18s %.4f ms  %.0f TF\n", "split", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<256, 128, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<256,128,2>", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<256, 256, 2, true>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<256,256,2> pp", t, flop / t / 1e9);
    t = timeit([&] { launch_split<true>(a, 0); }, 10);                printf("  %-18s %.4f ms  %.0f TF\n", "split pp", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<256, 256, 2, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<256,256,2> pp2", t, flop / t / 1e9);
    t = timeit([&] { launch_split<2>(a, 0); }, 10);                printf("  %-18s %.4f ms  %.0f TF\n", "split pp2", t, flop / t / 1e9);
    a.n_co_tiles = N / 128;
    t = timeit([&] { launch_variant<128, 128, 2, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<128,128,2> pp2", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 128, 2, true>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<128,128,2> pp", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 128, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<128,128,2> 2WG/CU", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 256, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<128,256,2>", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 128, 3>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<128,128,3>", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 64, 3>(a, 0); }, 10);  printf("  %-18s %.4f ms  %.0f TF\n", "<128,64,3> 2WG/CU", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 64, 2>(a, 0); }, 10);  printf("  %-18s %.4f ms  %.0f TF\n", "<128,64,2> 3WG/CU", t, flop / t / 1e9);
    t = timeit([&] { launch_variant<128, 128, 2>(a, 0); }, 10); printf("  %-18s %.4f ms  %.0f TF\n", "<128,128,2> again", t, flop / t / 1e9);
  }
  return 0;
}
