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

template <int KC, int CM, int NW>
int run(int R) {
  const int N = 16 * CM * NW * (KC == 8 ? 2 : 1), K = 64 * KC;
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
  // the weights once more in the kernel's fragment order (pack_conv, conv_gemm.hip)
  std::vector<uint16_t> hf(hw.size());
  {
    size_t d = 0;
    const int slab_ch = 16 * CM * NW;
    for (int slab = 0; slab < N / slab_ch; ++slab)
      for (int wv = 0; wv < NW; ++wv)
        for (int c = 0; c < 2 * KC; ++c)
          for (int m = 0; m < CM; ++m)
            for (int lane = 0; lane < 64; ++lane) {
              const int co = slab * slab_ch + wv * 16 * CM + (CM == 2 ? wreg_row(m, lane & 15) : m * 16 + (lane & 15)), kk = c * 32 + (lane >> 4) * 8;
              for (int q = 0; q < 8; ++q) hf[d++] = hw[(size_t)co * K + kk + q];
            }
  }
  bf16* wfrag;
  CK(hipMalloc(&wfrag, hf.size() * 2));
  CK(hipMemcpy(wfrag, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  Gemm1x1Args a;
  a.x = x; a.w = w; a.y = yref; a.y2 = nullptr; a.scale = nullptr; a.shift = nullptr;
  a.ldx = K; a.ldw = K; a.ldy = N; a.ldy2 = 0; a.y2_from = 0; a.R = R; a.cout = N; a.K = K; a.flags = EPI_RELU;
  auto tiles = [&](const Gemm1x1Args& aa, hipStream_t st) {
    Gemm1x1Args b = aa;
    if (KC == 8) { b.n_co_tiles = N / 128; return launch_variant<128, 128, 2>(b, st); }
    b.n_co_tiles = 1;
    return launch_variant<128, 64, 3>(b, st);
  };
  tiles(a, 0);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> h0((size_t)R * N), h1((size_t)R * N);
  CK(hipMemcpy(h0.data(), yref, h0.size() * 2, hipMemcpyDeviceToHost));
  a.y = y;
  a.wfrag = wfrag;
  struct V { const char* name; int (*fn)(const Gemm1x1Args&, hipStream_t); };
  const V vs[] = {{"SK2", launch_wreg<KC, CM, 2, NW>}, {"SK4", launch_wreg<KC, CM, 4, NW>}, {"SK8", launch_wreg<KC, CM, 8, NW>}};
  for (const V& v : vs) {
    CK(hipMemset(y, 0, (size_t)R * N * 2));
    v.fn(a, 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1.data(), y, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    printf("K=%d N=%d R=%d  %-6s vs tile kernel: %zu of %zu outputs differ\n", K, N, R, v.name, bad, h0.size());
  }
  const double flop = 2.0 * R * N * K, bytes = 2.0 * R * (K + N);
  for (int round = 0; round < 3; ++round)
    for (int cold = 0; cold < 2; ++cold) {
      const float t0 = timeit([&] { tiles(a, 0); }, 10, cold ? flush : nullptr);
      printf("  %s  tiles %.4f ms %.0f TF %.2f TB/s |", cold ? "cold" : "warm", t0, flop / t0 / 1e9, bytes / t0 / 1e9);
      for (const V& v : vs) {
        const float t = timeit([&] { v.fn(a, 0); }, 10, cold ? flush : nullptr);
        printf(" %s %.4f (%.2f TB/s) |", v.name, t, bytes / t / 1e9);
      }
      printf("\n");
    }
  return 0;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 77056, K = argc > 2 ? atoi(argv[2]) : 512;
  if (K == 512) return run<8, 2, 8>(R);   // (<8, 4, 4>, one wave per SIMD: profiles/r02_wreg_one_wave_per_simd.log)
  return run<24, 2, 4>(R);
}
