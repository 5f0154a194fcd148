// Diagnostic: does a workgroup's LDS stay intact while workgroups of ANOTHER kernel (another stream) share its CU?
// checker: 40 000 B of static LDS (the size of fbank_fft_kernel's), writes a pattern, barrier, verifies, repeats.
// writer:  short kernel with 48 KB (or argv) of dynamic LDS that fills ONLY its own allocation with 0xDEAD....
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_corun.hip -o tools/lds_corun
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NW = 10000;
__global__ __launch_bounds__(256, 4) void checker(int iters, unsigned* err, unsigned* log) {
  __shared__ unsigned buf[NW];
  const int tid = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const unsigned tag = (blockIdx.x * 977u + it) * 2654435761u;
    for (int i = tid; i < NW; i += 256) buf[i] = tag ^ (unsigned)i;
    __syncthreads();
    for (int r = 0; r < 4; ++r)
      for (int i = (tid * 7 + r * 64) % 256; i < NW; i += 256) {
        const unsigned got = buf[i], want = tag ^ (unsigned)i;
        if (got != want) {
          const unsigned k = atomicAdd(err, 1u);
          if (k < 64) { log[4 * k] = blockIdx.x; log[4 * k + 1] = i; log[4 * k + 2] = got; log[4 * k + 3] = want; }
        }
      }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void writer(int words, int reps, unsigned* sink) {
  extern __shared__ unsigned dyn[];
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i < words; i += 256) dyn[i] = 0xDEAD0000u | (unsigned)(i & 0xffff);
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += 256) acc += dyn[(i * 33) % words];
    __syncthreads();
  }
  if (acc == 12345u) *sink = acc;
}

int main(int argc, char** argv) {
  const int wbytes = argc > 1 ? atoi(argv[1]) : 49152;
  const int wgrid = argc > 2 ? atoi(argv[2]) : 64;
  const int launches = argc > 3 ? atoi(argv[3]) : 2000;
  unsigned *err, *log, *sink;
  CK(hipMalloc(&err, 4)); CK(hipMalloc(&log, 64 * 16)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(err, 0, 4));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)writer, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int round = 0; round < 2; ++round) {
    const bool corun = round == 1;
    CK(hipMemset(err, 0, 4));
    hipLaunchKernelGGL(checker, dim3(512), dim3(256), 0, a, 3000, err, log);
    if (corun)
      for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(writer, dim3(wgrid), dim3(256), wbytes, b, wbytes / 4, 4, sink);
    CK(hipDeviceSynchronize());
    unsigned h = 0, hl[256];
    CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hl, log, sizeof(hl), hipMemcpyDeviceToHost));
    printf("%s: writer LDS %d B x %d WGs x %d launches: checker errors %u\n", corun ? "co-run" : "alone ", wbytes, wgrid, launches, h);
    for (unsigned k = 0; k < h && k < 8; ++k) printf("   block %u word %u got %08x want %08x\n", hl[4 * k], hl[4 * k + 1], hl[4 * k + 2], hl[4 * k + 3]);
  }
  return 0;
}
