// Does the catconv GEMM's operand over-fetch cost time?  (FETCH_SIZE x 2 + WRITE_SIZE = 0.80 GB
// per launch against 0.47 GB algorithmic; DESIGN argues the excess is Infinity-Cache hits that the kernel, bound by its
// MFMA / LDS phases, never waits for -- asserted, not shown.)  The library's catconv launch (gemm1x1_persist_kernel<2>,
// quadrant schedule) on the catconv shape, same instruction stream, two activation layouts:
//   real      ldx = K: every 256-frame row tile is its own 786 KB of x
//   aliased   ldx = 0: every frame reads row 0 -- the activation operand is 3 KB, always in L1 / L2: NO x traffic at all
// If the two take the same time, the memory side of x is not what the kernel waits for.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Isugar_amd/csrc -Iinclude tools/catconv_traffic.hip -o tools/catconv_traffic
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
int64_t tune_get(int) { return 0; }   // library defaults: persistent grid, quadrant schedule in two phases
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
  const int R = argc > 1 ? atoi(argv[1]) : 77056, N = 1536, K = 1536;
  std::vector<uint16_t> hx((size_t)R * K), hw((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return host_f32_to_bf16(((s >> 8) & 0xffff) / 32768.0f - 1.0f); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  bf16 *x, *w, *y;
  void* flush;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&y, (size_t)R * N * 2));
  CK(hipMalloc(&flush, 1u << 30));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  Gemm1x1Args a;
  a.x = x; a.w = w; a.y = y; a.y2 = nullptr; a.scale = nullptr; a.shift = nullptr;
  a.ldx = K; a.ldw = K; a.ldy = N; a.ldy2 = 0; a.y2_from = 0; a.R = R; a.cout = N; a.K = K; a.flags = EPI_RELU;
  a.n_co_tiles = N / 256;
  Gemm1x1Args al = a;
  al.ldx = 0;                                     // every frame = row 0
  const double flop = 2.0 * R * N * K;
  printf("catconv shape R=%d K=%d N=%d (library launch: launch_split<1>)\n", R, K, N);
  for (int round = 0; round < 3; ++round)
    for (int cold = 0; cold < 2; ++cold) {
      const float t0 = timeit([&] { launch_split<1>(a, 0); }, 10, cold ? flush : nullptr);
      const float t1 = timeit([&] { launch_split<1>(al, 0); }, 10, cold ? flush : nullptr);
      printf("  %s  real layout %.4f ms (%.0f TF) | activation rows aliased to row 0 (no x traffic) %.4f ms (%.0f TF)\n",
             cold ? "after a 1 GB memset" : "back to back       ", t0, flop / t0 / 1e9, t1, flop / t1 / 1e9);
    }
  return 0;
}
