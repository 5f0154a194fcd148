// Stand-alone timing + in-kernel stamp harness for the fused Res2 chain (diagnostic; not part of the library).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -DR2_STAMPS -Isugar_amd/csrc tools/res2_bench.hip -o tools/res2_bench
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "../sugar_amd/csrc/res2.hip"

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

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 301, wp = 64, C = 512;
  const int ktaps = 3, dil = argc > 3 ? atoi(argv[3]) : 2, k_pad = 192;
  const int64_t R = (int64_t)B * T;
  std::vector<uint16_t> hx((size_t)R * C), hw((size_t)64 * k_pad);
  uint32_t s = 777;
  auto rnd = [&](float sc) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * sc; return sugar::host_f32_to_bf16(f); };
  for (auto& v : hx) v = rnd(1.0f);
  for (auto& v : hw) v = rnd(0.1f);
  std::vector<int32_t> ro(B + 1);
  for (int b = 0; b <= B; ++b) ro[b] = b * T;
  std::vector<float> hs(64, 1.0f), hb(64, 0.0f);
  void *x, *y, *w; float *sc, *sh; int32_t* dro; long long* stamps;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&y, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2));
  CK(hipMalloc(&sc, 256)); CK(hipMalloc(&sh, 256)); CK(hipMalloc(&dro, (B + 1) * 4)); CK(hipMalloc(&stamps, 4 * 64 * 8));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, hs.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, hb.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(dro, ro.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(stamps, 0, 4 * 64 * 8));
  sugar::ConvLayer steps[7];
  for (int j = 0; j < 7; ++j) {
    steps[j].w = w; steps[j].scale = sc; steps[j].shift = sh; steps[j].cout = wp; steps[j].n_pad = 64; steps[j].cin_p = wp;
    steps[j].ktaps = ktaps; steps[j].dil = dil; steps[j].k_pad = k_pad; steps[j].flags = 3;
  }
  long long* null_ptr = nullptr;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(sugar::g_r2_stamps), &null_ptr, sizeof(null_ptr)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i)
      if (sugar::launch_res2_chain(steps, x, y, C, wp, dro, B, T, 0)) return 1;
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("res2 chain B=%d T=%d dil=%d: %.2f us / launch\n", B, T, dil, ms * 1000 / 20);
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(sugar::g_r2_stamps), &stamps, sizeof(stamps)));
  if (sugar::launch_res2_chain(steps, x, y, C, wp, dro, B, T, 0)) return 1;
  CK(hipDeviceSynchronize());
  std::vector<long long> st(4 * 64);
  CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
  const char* who[4] = {"wg0 wave0", "wg0 wave7", "wg200 wave0", "wg200 wave7"};
  const char* lab[6] = {"barrierB", "issue loads", "mfma loop", "epilogue", "barrierA", "write_u+store_w"};
  for (int k = 0; k < 4; ++k) {
    const long long* t = st.data() + k * 64;
    printf("%s: start->first barrier %lld cyc; total %lld cyc\n", who[k], t[1] - t[0], t[1 + 42] - t[0]);
    for (int j = 0; j < 7; ++j) {
      printf("  step %d:", j);
      long long prev = t[1 + 6 * j - (j ? 0 : 0)];
      prev = j ? t[1 + 6 * j] : t[1];
      for (int p = 0; p < 6; ++p) {
        const long long cur = t[2 + 6 * j + p];
        printf(" %s %lld |", lab[p], cur - prev);
        prev = cur;
      }
      printf("\n");
    }
  }
  return 0;
}
