// The 128 x 128 tile GEMM on the narrow 1x1 convs of sampled subnets (diagnostic): time against the HBM roofline.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Isugar_amd/csrc tools/narrow_gemm.hip -o tools/narrow_gemm
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

int main() {
  const int R = 77056;
  const int shapes[][2] = {{128, 128}, {256, 256}, {256, 384}, {384, 384}, {384, 512}, {512, 384}, {512, 512}};   // K, N
  bf16 *x, *w, *y;
  void* flush;
  CK(hipMalloc(&x, (size_t)R * 512 * 2)); CK(hipMalloc(&w, 512 * 512 * 2)); CK(hipMalloc(&y, (size_t)R * 512 * 2));
  CK(hipMalloc(&flush, 1u << 30));
  CK(hipMemset(x, 0x3c, (size_t)R * 512 * 2)); CK(hipMemset(w, 0x3c, 512 * 512 * 2));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& sh : shapes) {
    const int K = sh[0], N = sh[1];
    Gemm1x1Args a;
    a.x = x; a.w = w; a.y = y; a.y2 = nullptr; a.scale = nullptr; a.shift = nullptr;
    a.ldx = K; a.ldw = K; a.ldy = N; a.ldy2 = 0; a.y2_from = 0; a.R = R; a.cout = N; a.K = K; a.flags = EPI_RELU;
    a.n_co_tiles = (N + 127) / 128;
    const double mb = (double)R * (K + N) * 2 / 1e6;
    for (int cold = 0; cold < 2; ++cold) {
      float tot = 0;
      const int iters = 10;
      launch_variant<128, 128, 2>(a, 0);
      for (int i = 0; i < iters; ++i) {
        if (cold) CK(hipMemsetAsync(flush, i, 1u << 30, 0));
        CK(hipEventRecord(e0));
        launch_variant<128, 128, 2>(a, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
      }
      const double us = tot / iters * 1e3;
      printf("K %3d -> N %3d  %s  %6.1f us  %5.1f MB  %5.2f TB/s  %.2f of the 8 TB/s roofline\n", K, N, cold ? "cold" : "warm",
             us, mb, mb / us, mb / us / 8.0);
    }
  }
  return 0;
}
