// Reproducer: the library's front-end (sugar_fbank_fwd: wave-level FFT, VALU + LDS only) on stream A while a synthetic
// co-runner occupies the same CUs from stream B; every launch's log-mel output is compared with the output of a launch
// that ran alone.  Co-runners: none | mfma (pure MFMA, no memory at all) | valu (pure FMA) | lds (LDS traffic only) |
// mem (global streaming).  With a library built WITH packed-FP32 instructions
// (SUGAR_ALLOW_PACKED_FP32=1 python -m sugar_amd.build --force) the mfma co-runner makes ~10-100 of 600 launches
// differ (single frames, one FFT element wrong); without them (the default build) none do.  DESIGN.md section 5b.
// Build: hipcc -O3 --offload-arch=gfx950 -Iinclude tools/fbank_corun.hip -Lsugar_amd -lsugar_amd -Wl,-rpath,'$ORIGIN/../sugar_amd' -o tools/fbank_corun
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "sugar_amd.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void burn_mfma(int iters, float* sink) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x4 acc[8];
  for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
  float s = 0.f;
  for (int m = 0; m < 8; ++m) s += acc[m][0];
  if (s == 1234.5f) *sink = s;
}
__global__ __launch_bounds__(256) void burn_valu(int iters, float* sink) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], 1.0001f, 0.5f);
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 1234.5f) *sink = s;
}
__global__ __launch_bounds__(256) void burn_lds(int iters, float* sink) {
  extern __shared__ float dyn[];
  float s = 0.f;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < 8192; i += 256) dyn[i] = s + i;
    __syncthreads();
    for (int i = threadIdx.x; i < 8192; i += 256) s += dyn[(i * 33) & 8191];
    __syncthreads();
  }
  if (s == 1234.5f) *sink = s;
}
__global__ __launch_bounds__(256) void burn_mem(const float4* src, float4* dst, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "mfma";
  const int reps = argc > 2 ? atoi(argv[2]) : 400;
  const int B = 4, L = 36000, T = 1 + L / 160;
  std::vector<float> hw((size_t)B * L);
  unsigned s = 1;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  std::vector<int64_t> hoff(B); std::vector<int32_t> hlen(B, L), hrow(B + 1);
  for (int b = 0; b < B; ++b) { hoff[b] = (int64_t)b * L; hrow[b + 1] = hrow[b] + T; }
  const int64_t R = hrow[B];
  float *wav, *logmel, *sink; int64_t* off; int32_t *len, *row; void* feats;
  CK(hipMalloc(&wav, hw.size() * 4)); CK(hipMalloc(&off, B * 8)); CK(hipMalloc(&len, B * 4)); CK(hipMalloc(&row, (B + 1) * 4));
  CK(hipMalloc(&logmel, sugar_fbank_scratch_bytes(B, R))); CK(hipMalloc(&feats, R * 80 * 2)); CK(hipMalloc(&sink, 4));
  CK(hipMemcpy(wav, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(off, hoff.data(), B * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(len, hlen.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(row, hrow.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  float4 *src = nullptr, *dst = nullptr; const size_t nmem = 1u << 22;
  CK(hipMalloc(&src, nmem * 16)); CK(hipMalloc(&dst, nmem * 16)); CK(hipMemset(src, 0, nmem * 16));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  std::vector<float> ref((size_t)R * 80), got((size_t)R * 80);
  if (sugar_fbank_fwd(wav, off, len, row, B, R, T, logmel, feats, SUGAR_BF16, a) != 0) { printf("fbank: %s\n", sugar_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), logmel, ref.size() * 4, hipMemcpyDeviceToHost));
  int bad_launches = 0;
  size_t bad_values = 0;
  for (int r = 0; r < reps; ++r) {
    for (int k = 0; k < 3; ++k) {
      if (!strcmp(mode, "mfma")) hipLaunchKernelGGL(burn_mfma, dim3(512), dim3(256), 0, b, 300, sink);
      else if (!strcmp(mode, "valu")) hipLaunchKernelGGL(burn_valu, dim3(512), dim3(256), 0, b, 2000, sink);
      else if (!strcmp(mode, "lds")) hipLaunchKernelGGL(burn_lds, dim3(512), dim3(256), 32768, b, 8, sink);
      else if (!strcmp(mode, "mem")) hipLaunchKernelGGL(burn_mem, dim3(512), dim3(256), 0, b, src, dst, nmem);
    }
    if (sugar_fbank_fwd(wav, off, len, row, B, R, T, logmel, feats, SUGAR_BF16, a) != 0) { printf("fbank: %s\n", sugar_last_error()); return 1; }
    CK(hipStreamSynchronize(a));
    CK(hipMemcpy(got.data(), logmel, got.size() * 4, hipMemcpyDeviceToHost));
    size_t nb = 0;
    for (size_t i = 0; i < got.size(); ++i) nb += memcmp(&got[i], &ref[i], 4) != 0;
    bad_launches += nb != 0;
    bad_values += nb;
  }
  CK(hipDeviceSynchronize());
  printf("co-runner %-5s: %d fbank launches, %d differ from the launch that ran alone (%zu log-mel values)\n", mode, reps,
         bad_launches, bad_values);
  return bad_launches != 0;
}
