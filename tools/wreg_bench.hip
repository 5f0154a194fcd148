// Weights-in-registers GEMM (gemm_wreg_kernel) against the 128 x 128 tile kernel on the 512-channel conv shape, in one
// process; with -DWREG_KNOCK=bits the kernel leaves parts out (see conv1x1.hip) to show what its time is made of.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Isugar_amd/csrc -Iinclude [-DWREG_KNOCK=n] tools/wreg_bench.hip -o tools/wreg_bench
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
int64_t tune_get(int) { return 0; }
}  // namespace sugar

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using namespace sugar;

template <typename F>
float timeit(F f, int iters, void* flush) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float tot = 0;
  f();
  for (int i = 0; i < iters; ++i) {
    if (flush) CK(hipMemsetAsync(flush, i, 1u << 30, 0));
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

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 77056, N = 512, K = 512;
  std::vector<uint16_t> hx((size_t)R * K), hw((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return host_f32_to_bf16(((s >> 8) & 0xffff) / 32768.0f - 1.0f); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  bf16 *x, *w, *y, *yref;
  void* flush;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2));
  CK(hipMalloc(&y, (size_t)R * N * 2)); CK(hipMalloc(&yref, (size_t)R * N * 2)); CK(hipMalloc(&flush, 1u << 30));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  // the weights once more in gemm_wreg_kernel<8, 2, 2>'s fragment order (pack_conv, conv_gemm.hip)
  std::vector<uint16_t> hf(hw.size());
  {
    size_t d = 0;
    for (int slab = 0; slab < N / 256; ++slab)
      for (int wv = 0; wv < 8; ++wv)
        for (int c = 0; c < 16; ++c)
          for (int m = 0; m < 2; ++m)
            for (int lane = 0; lane < 64; ++lane) {
              const int co = slab * 256 + wv * 32 + m * 16 + (lane & 15), kk = c * 32 + (lane >> 4) * 8;
              for (int q = 0; q < 8; ++q) hf[d++] = hw[(size_t)co * 512 + kk + q];
            }
  }
  bf16* wfrag;
  CK(hipMalloc(&wfrag, hf.size() * 2));
  CK(hipMemcpy(wfrag, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  Gemm1x1Args a;
  a.x = x; a.w = w; a.y = yref; a.y2 = nullptr; a.scale = nullptr; a.shift = nullptr;
  a.ldx = K; a.ldw = K; a.ldy = N; a.ldy2 = 0; a.y2_from = 0; a.R = R; a.cout = N; a.K = K; a.flags = EPI_RELU;
  a.n_co_tiles = N / 128;
  launch_variant<128, 128, 2>(a, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> h0((size_t)R * N), h1((size_t)R * N);
  CK(hipMemcpy(h0.data(), yref, h0.size() * 2, hipMemcpyDeviceToHost));
  a.y = y;
  struct V { const char* name; int (*fn)(const Gemm1x1Args&, hipStream_t); bool frag; };
  const V vs[] = {{"base", launch_wreg<8, 2, 2, 0, 0>, false}, {"base+wfrag", launch_wreg<8, 2, 2, 0, 0>, true},
                  {"lag2+wfrag", launch_wreg<8, 2, 2, 2, 0>, true}, {"defer+wfrag", launch_wreg<8, 2, 2, 0, 1>, true},
                  {"lag2+defer+wfrag", launch_wreg<8, 2, 2, 2, 1>, true}};
  for (const V& v : vs) {
    a.wfrag = v.frag ? wfrag : nullptr;
    CK(hipMemset(y, 0, (size_t)R * N * 2));
    v.fn(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1.data(), y, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    printf("WREG_KNOCK=%d  R=%d  %-18s vs <128,128,2>: %zu of %zu outputs differ\n", WREG_KNOCK, R, v.name, bad, h0.size());
  }
  const double flop = 2.0 * R * N * K;
  for (int round = 0; round < 3; ++round)
    for (int cold = 0; cold < 2; ++cold) {
      a.wfrag = nullptr;
      const float t0 = timeit([&] { launch_variant<128, 128, 2>(a, 0); }, 10, cold ? flush : nullptr);
      printf("  %s  <128,128,2> %.4f ms %.0f TF |", cold ? "cold" : "warm", t0, flop / t0 / 1e9);
      for (const V& v : vs) {
        a.wfrag = v.frag ? wfrag : nullptr;
        const float t = timeit([&] { v.fn(a, 0); }, 10, cold ? flush : nullptr);
        printf(" %s %.4f |", v.name, t);
      }
      printf("\n");
    }
  return 0;
}
