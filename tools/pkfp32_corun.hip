// Minimal reproducer, independent of the library: a kernel that executes ONE packed-FP32 opcode in a loop (inline asm:
// v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, or the scalar v_fma_f32 as control) on stream A, while a kernel that
// issues only v_mfma_f32_16x16x32_bf16 runs on stream B.  The victim's output is a pure function of the thread index;
// every launch is compared with a launch that ran alone.  See DESIGN.md section 5b.
// Build: hipcc -O3 --offload-arch=gfx950 tools/pkfp32_corun.hip -o tools/pkfp32_corun
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OP>
__global__ __launch_bounds__(256) void victim(int iters, f32x2* out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f32x2 acc = {1.0f + (t & 1023) * 1e-3f, 2.0f - (t & 511) * 1e-3f};
  const f32x2 a = {0.99991f, 1.00007f}, b = {1e-3f, -2e-3f};
  for (int i = 0; i < iters && OP < 4; ++i) {
    if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    if (OP == 1) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "v"(a)); }
    if (OP == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(b)); }
    if (OP == 3) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.x) : "v"(a.x), "v"(b.x));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc.y) : "v"(a.y), "v"(b.y));
    }
  }
  // OP >= 4: the complex rotation x *= (c + i s), every 8th iteration followed by an exchange with the neighbouring lane
  // through LDS inside the wave, in the instruction forms the compiler emits for the FFT butterflies of fbank.hip
  // (op_sel half swaps, neg modifiers).  Bits of (OP - 4): 1 = rotation in packed ops, 2 = LDS exchange,
  // 4 = arithmetic after the exchange in packed ops, 8 = s_waitcnt between the LDS write and the read,
  // 16 = 16 idle cycles between the s_waitcnt for the LDS read and the packed op, 32 = packed op without modifiers,
  // 64 = the LDS-loaded operand is copied by v_mov_b32 before the packed op reads it, 128 / 256 = only the op_sel /
  // only the neg modifiers.
  if (OP >= 4) {
    constexpr int F = OP - 4;
    __shared__ f32x2 ex[256];
    const f32x2 c = {0.999876632f, 0.999876632f}, s = {0.015707317f, 0.015707317f};
    f32x2 tt, u, x2;
    for (int i = 0; i < iters; ++i) {
      if (F & 1) {
        asm volatile(
            "v_pk_mul_f32 %1, %0, %5 op_sel:[1,0] op_sel_hi:[0,1]\n"
            "v_pk_fma_f32 %2, %0, %4, %1 neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
            "v_pk_fma_f32 %3, %0, %4, %1\n"
            : "+v"(acc), "=&v"(tt), "=&v"(u), "=&v"(x2) : "v"(c), "v"(s));
        acc.x = u.x; acc.y = x2.y;
      } else {
        float re, im;
        asm volatile(
            "v_mul_f32 %0, %3, %5\n"
            "v_mul_f32 %1, %2, %5\n"
            "v_fma_f32 %0, %2, %4, -%0\n"
            "v_fma_f32 %1, %3, %4, %1\n"
            : "=&v"(re), "=&v"(im) : "v"(acc.x), "v"(acc.y), "v"(c.x), "v"(s.x));
        acc.x = re; acc.y = im;
      }
      if ((F & 2) && (i & 7) == 7) {
        ex[threadIdx.x] = acc;
        if (F & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        f32x2 o = ex[threadIdx.x ^ 1];
        __builtin_amdgcn_wave_barrier();
        if (F & 16) asm volatile("s_nop 7\ns_nop 7" ::: "memory");
        if (F & 64) asm volatile("v_mov_b32 %0, %0\nv_mov_b32 %1, %1\ns_nop 1" : "+v"(o.x), "+v"(o.y));
        if ((F & 4) && (F & 32)) {
          asm volatile("v_pk_add_f32 %0, %0, %1\n"
                       "v_pk_mul_f32 %0, %0, %2" : "+v"(acc) : "v"(o), "v"((f32x2){0.5f, 0.5f}));
        } else if ((F & 4) && (F & 128)) {
          asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n"
                       "v_pk_mul_f32 %0, %0, %2" : "+v"(acc) : "v"(o), "v"((f32x2){0.5f, 0.5f}));
        } else if ((F & 4) && (F & 256)) {
          asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n"
                       "v_pk_mul_f32 %0, %0, %2" : "+v"(acc) : "v"(o), "v"((f32x2){0.7071f, 0.7071f}));
        } else if (F & 4) {
          asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                       "v_pk_mul_f32 %0, %0, %2" : "+v"(acc) : "v"(o), "v"((f32x2){0.7071f, 0.7071f}));
        } else {
          float re, im;
          asm volatile(
              "v_sub_f32 %0, %2, %5\n"
              "v_sub_f32 %1, %3, %4\n"
              "v_mul_f32 %0, %0, %6\n"
              "v_mul_f32 %1, %1, %6\n"
              : "=&v"(re), "=&v"(im) : "v"(acc.x), "v"(acc.y), "v"(o.x), "v"(o.y), "v"(0.7071f));
          acc.x = re; acc.y = im;
        }
      }
    }
  }
  out[t] = acc;
}

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

template <int OP>
void run(const char* name, int reps, hipStream_t sa, hipStream_t sb, f32x2* out, float* sink) {
  const int n = 512 * 256;
  std::vector<f32x2> ref(n), got(n);
  int shown = 0;
  hipLaunchKernelGGL(victim<OP>, dim3(512), dim3(256), 0, sa, 4000, out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), out, n * sizeof(f32x2), hipMemcpyDeviceToHost));
  for (int corun = 0; corun < 2; ++corun) {
    int bad_launches = 0;
    size_t bad_vals = 0;
    for (int r = 0; r < reps; ++r) {
      if (corun)
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(burn_mfma, dim3(512), dim3(256), 0, sb, 300, sink);
      hipLaunchKernelGGL(victim<OP>, dim3(512), dim3(256), 0, sa, 4000, out);
      CK(hipStreamSynchronize(sa));
      CK(hipMemcpy(got.data(), out, n * sizeof(f32x2), hipMemcpyDeviceToHost));
      size_t nb = 0;
      for (int i = 0; i < n; ++i) {
        const bool d = memcmp(&got[i], &ref[i], sizeof(f32x2)) != 0;
        if (d && shown < 6) {
          ++shown;
          printf("    thread %6d (lane %2d): alone (%.9g, %.9g)  co-run (%.9g, %.9g)\n", i, i & 63, ref[i].x, ref[i].y,
                 got[i].x, got[i].y);
        }
        nb += d;
      }
      bad_launches += nb != 0;
      bad_vals += nb;
    }
    CK(hipDeviceSynchronize());
    printf("%-16s %-22s: %d of %d launches differ from the launch that ran alone (%zu of %d values per launch on average)\n",
           name, corun ? "with the MFMA burner" : "alone", bad_launches, reps, bad_launches ? bad_vals / bad_launches : 0, n);
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  f32x2* out; float* sink;
  CK(hipMalloc(&out, 512 * 256 * sizeof(f32x2))); CK(hipMalloc(&sink, 4));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  const bool all = argc > 2;
  if (all) {
    run<0>("v_pk_fma_f32", reps, a, b, out, sink);
    run<1>("v_pk_mul_f32", reps, a, b, out, sink);
    run<2>("v_pk_add_f32", reps, a, b, out, sink);
    run<3>("v_fma_f32 x2", reps, a, b, out, sink);
    run<4 + 1>("pk rot", reps, a, b, out, sink);
    run<4 + 0>("f32 rot", reps, a, b, out, sink);
    run<4 + 1 + 2>("pk rot+LDS+f32", reps, a, b, out, sink);
    run<4 + 1 + 2 + 4>("pk rot+LDS+pk", reps, a, b, out, sink);
    run<4 + 1 + 2 + 4 + 8>("pk rot+LDSw+pk", reps, a, b, out, sink);
  }
  run<4 + 2>("f32 rot+LDS+f32", reps, a, b, out, sink);
  run<4 + 2 + 4>("f32 rot+LDS+pk", reps, a, b, out, sink);
  run<4 + 2 + 4 + 16>("  + 16 nops", reps, a, b, out, sink);
  run<4 + 2 + 4 + 32>("  no modifiers", reps, a, b, out, sink);
  run<4 + 2 + 4 + 64>("  via v_mov", reps, a, b, out, sink);
  run<4 + 2 + 4 + 128>("  op_sel only", reps, a, b, out, sink);
  run<4 + 2 + 4 + 256>("  neg only", reps, a, b, out, sink);
  return 0;
}
