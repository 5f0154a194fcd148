// What does the SHAPE of an LDS-DMA request stream cost when the data comes from HBM?
// Every LDS-DMA GEMM of this library stages [rows x 64 k] sub-tiles: one global_load_lds_dwordx4 wave-instruction
// fetches 8 rows x 128 bytes at the row pitch (1 KiB for K = 512, 3 KiB for K = 1536), and the next k-tile comes back to
// the same rows 128 bytes further on, a microsecond later.  Cold (HBM-resident) inputs stream at only ~2.4 TB/s through
// those kernels (tools/wreg_bench, knock-out builds: the DMA alone takes 33 us for 79 MB).  This probe streams a buffer
// far larger than the Infinity Cache through the same ring (96 KiB in flight per CU, no compute, nothing stored) with
//   pattern 0: sub-tiles of 64 rows x 128 B, walking k inside a 64-row tile (what the kernels do)
//   pattern 1: one instruction = one whole contiguous KiB (a full K = 512 row); a 64-row tile is one 64 KiB run
//   pattern 2: as 0 but 16 rows x 512 B per 8 KiB sub-tile (4 x 128 B chunks per row per instruction)
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/dma_pattern.hip -o tools/dma_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int SUB = 8192;          // bytes per sub-tile (8 wave-instructions of 1 KiB)
constexpr int NSLOT = 13, D = 12;  // ring slots, sub-tiles in flight

template <int PATTERN>
__global__ __launch_bounds__(512, 2) void stream_kernel(const unsigned char* __restrict__ x, long pitch, int n_tiles, int P,
                                                        unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = blockIdx.x;
  const int t0 = (int)((long)p * n_tiles / P), t1 = (int)((long)(p + 1) * n_tiles / P);
  const int G = (t1 - t0) * 8;     // sub-tiles: 8 per 64-row tile (64 rows x 1 KiB)
  auto issue = [&](int g) {
    const int tile = t0 + g / 8, s = g % 8;
    const unsigned char* src;
    if (PATTERN == 0) {            // rows 8 wv .. 8 wv + 7 of the tile, bytes [128 s, 128 s + 128)
      const int row = wv * 8 + (lane >> 3);
      src = x + ((long)tile * 64 + row) * pitch + s * 128 + (lane & 7) * 16;
    } else if (PATTERN == 1) {     // row 8 s + wv of the tile, all 1024 bytes
      src = x + ((long)tile * 64 + s * 8 + wv) * pitch + lane * 16;
    } else {                       // rows 16 (s & 3) + 2 wv + {0, 1}, bytes [512 (s >> 2), + 512)
      const int row = (s & 3) * 16 + wv * 2 + (lane >> 5);
      src = x + ((long)tile * 64 + row) * pitch + (s >> 2) * 512 + (lane & 31) * 16;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + (g % NSLOT) * SUB + wv * 1024), 16, 0, 0);
  };
  for (int g = 0; g < D && g < G; ++g) issue(g);
  unsigned acc = 0;
  for (int g = 0; g < G; ++g) {
    if (g + D < G) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    acc += *(const unsigned*)(smem + (g % NSLOT) * SUB + threadIdx.x * 16);   // touch the stage
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (g + D < G) issue(g + D);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int PATTERN>
float run(const unsigned char* x, long pitch, int n_tiles, unsigned* sink, void* flush) {
  const int P = 256;
  const int smem = NSLOT * SUB;
  CK(hipFuncSetAttribute((const void*)stream_kernel<PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    if (flush) CK(hipMemsetAsync(flush, it, 1u << 30, 0));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream_kernel<PATTERN>, dim3(P), dim3(512), smem, 0, x, pitch, n_tiles, P, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 77056;       // 64-row tiles of 1 KiB rows
  const long pitch = argc > 2 ? atol(argv[2]) : 1024;       // row pitch in bytes (3072: the K = 1536 tensors)
  const int n_tiles = (int)(rows / 64);
  unsigned char* x;
  unsigned* sink;
  void* flush;
  CK(hipMalloc(&x, rows * pitch + 4096));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&flush, 1u << 30));
  CK(hipMemset(x, 1, rows * pitch + 4096));
  const double bytes = (double)n_tiles * 64 * 1024;
  for (int cold = 0; cold < 2; ++cold) {
    const float a = run<0>(x, pitch, n_tiles, sink, cold ? flush : nullptr);
    const float b = run<1>(x, pitch, n_tiles, sink, cold ? flush : nullptr);
    const float c = run<2>(x, pitch, n_tiles, sink, cold ? flush : nullptr);
    printf("rows %ld pitch %ld %s: 64x128B sub-tiles %.4f ms (%.2f TB/s) | whole-KiB rows %.4f ms (%.2f TB/s) | 16x512B %.4f ms (%.2f TB/s)\n",
           rows, pitch, cold ? "cold" : "warm", a, bytes / a / 1e9, b, bytes / b / 1e9, c, bytes / c / 1e9);
  }
  return 0;
}
