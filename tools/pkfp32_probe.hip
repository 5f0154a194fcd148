// Single-instruction probe for the observation in DESIGN.md section 5b: one packed-FP32 instruction per iteration, its
// result checked in the kernel against the same sums made by scalar v_add_f32 / v_mul_f32 / v_fma_f32, with and without
// a kernel that only issues v_mfma_f32_16x16x32_bf16 on another stream.  Mismatches are counted per 16-lane quarter of
// the wave and classified (does the result equal the sum WITHOUT the op_sel half swap?).
// Build: hipcc -O3 --offload-arch=gfx950 tools/pkfp32_probe.hip -o tools/pkfp32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Rec { int thread, iter; float x0, x1, y0, y1, r0, r1, e0, e1; };
struct Stat { unsigned quarter[4], unswapped, other, nrec, pad, iterb[4]; Rec rec[8]; };

__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int V>
__global__ __launch_bounds__(256) void probe(int iters, Stat* st) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f32x2 x = {1.0f + (t & 1023) * 1e-3f, 2.0f - (t & 511) * 1e-3f};
  f32x2 y = {0.25f + (t & 255) * 3e-3f, -0.5f + (t & 127) * 7e-3f};
  const f32x2 z = {0.75f, -1.25f};
  for (int i = 0; i < iters; ++i) {
    x.x = sadd(x.x, 1e-3f); x.y = sadd(x.y, -2e-3f); y.x = sadd(y.x, 3e-3f); y.y = sadd(y.y, 5e-4f);
    f32x2 r, e, u;   // e: expected, u: what the instruction would give without its op_sel swap
    if (V == 0) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, y.y); e.y = sadd(x.y, y.x); u.x = sadd(x.x, y.x); u.y = sadd(x.y, y.y); }
    if (V == 1) { asm volatile("s_nop 7\ns_nop 7\nv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\ns_nop 7\ns_nop 7" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, y.y); e.y = sadd(x.y, y.x); u.x = sadd(x.x, y.x); u.y = sadd(x.y, y.y); }
    if (V == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = smul(x.y, y.x); e.y = smul(x.x, y.y); u.x = smul(x.x, y.x); u.y = smul(x.y, y.y); }
    if (V == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.x, y.y, z.x); e.y = sfma(x.y, y.x, z.y); u.x = sfma(x.x, y.x, z.x); u.y = sfma(x.y, y.y, z.y); }
    if (V == 4) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, y.x); e.y = sadd(x.y, y.y); u = e; }
    if (V == 5) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, -y.x); e.y = sadd(x.y, -y.y); u = e; }
    if (V == 6) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));   // broadcast y.lo
                  e.x = sadd(x.x, y.x); e.y = sadd(x.y, y.x); u.x = e.x; u.y = sadd(x.y, y.y); }
    if (V == 7) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(y));   // broadcast y.hi
                  e.x = sadd(x.x, y.y); e.y = sadd(x.y, y.y); u.x = sadd(x.x, y.x); u.y = e.y; }
    if (V == 8) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = smul(x.x, y.y); e.y = smul(x.y, y.x); u.x = smul(x.x, y.x); u.y = smul(x.y, y.y); }
    if (V == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.x, y.x, z.y); e.y = sfma(x.y, y.y, z.x); u.x = sfma(x.x, y.x, z.x); u.y = sfma(x.y, y.y, z.y); }
    if (V == 10) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.y, y.x); e.y = sadd(x.x, y.y); u.x = sadd(x.x, y.x); u.y = sadd(x.y, y.y); }
    // round 2: the forms the front-end's packed butterflies use (src0 half swap, with neg on src0 or src1; FMA with src0 swap)
    if (V == 11) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.y, y.x); e.y = sadd(-x.x, y.y); u.x = sadd(x.x, y.x); u.y = sadd(-x.y, y.y); }
    if (V == 12) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(-x.y, y.x); e.y = sadd(x.x, y.y); u.x = sadd(-x.x, y.x); u.y = sadd(x.y, y.y); }
    if (V == 13) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.y, y.x, z.x); e.y = sfma(x.x, y.y, z.y); u.x = sfma(x.x, y.x, z.x); u.y = sfma(x.y, y.y, z.y); }
    if (V == 14) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, y.x); e.y = sadd(x.y, -y.y); u = e; }
    if (V == 15) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.x, y.x, z.x); e.y = sfma(x.y, y.y, z.y); u = e; }
    if (V == 16) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
                  e.x = smul(x.x, y.x); e.y = smul(x.y, y.y); u = e; }
    // forms hipcc itself picks for scalar code when the feature is on (low-half broadcasts: every op_sel bit 0)
    if (V == 17) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = smul(x.x, y.x); e.y = smul(x.y, y.x); u.x = e.x; u.y = smul(x.y, y.y); }
    if (V == 18) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.x, y.x, z.x); e.y = sfma(x.y, y.x, z.y); u.x = e.x; u.y = sfma(x.y, y.y, z.y); }
    if (V == 19) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.x, y.x, z.x); e.y = sfma(x.x, y.y, z.y); u.x = e.x; u.y = sfma(x.y, y.y, z.y); }
    if (V == 20) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
                  e.x = sfma(x.x, y.x, z.x); e.y = sfma(x.y, y.y, z.x); u.x = e.x; u.y = sfma(x.y, y.y, z.y); }
    if (V == 21) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, y.x); e.y = sadd(x.x, y.y); u.x = e.x; u.y = sadd(x.y, y.y); }
    if (V == 23) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));   // broadcast x.hi
                  e.x = sfma(x.y, y.x, z.x); e.y = sfma(x.y, y.y, z.y); u.x = sfma(x.x, y.x, z.x); u.y = e.y; }
    if (V == 24) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = sadd(x.x, -y.x); e.y = sadd(x.x, -y.y); u.x = e.x; u.y = sadd(x.y, -y.y); }
    if (V == 22) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
                  e.x = smul(x.x, y.x); e.y = smul(x.x, y.y); u.x = e.x; u.y = smul(x.y, y.y); }
    if (r.x != e.x || r.y != e.y) {
      atomicAdd(&st->quarter[(threadIdx.x & 63) >> 4], 1u);
      if (r.x == u.x && r.y == u.y) atomicAdd(&st->unswapped, 1u); else atomicAdd(&st->other, 1u);
      atomicAdd(&st->iterb[i == 0 ? 0 : i < 10 ? 1 : i < 100 ? 2 : 3], 1u);
      const unsigned k = atomicAdd(&st->nrec, 1u);
      if (k < 8) st->rec[k] = Rec{t, i, x.x, x.y, y.x, y.y, r.x, r.y, e.x, e.y};
    }
  }
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

__global__ __launch_bounds__(256) void burn_valu(int iters, float* sink) {
  float a[8];
  for (int m = 0; m < 8; ++m) a[m] = 0.001f * (threadIdx.x + m);
  for (int it = 0; it < iters * 8; ++it)
#pragma unroll
    for (int m = 0; m < 8; ++m) a[m] = sfma(a[m], 0.9999f, 1e-4f);
  float s = 0.f;
  for (int m = 0; m < 8; ++m) s += a[m];
  if (s == 1234.5f) *sink = s;
}

static int g_burner = 0;   // 0: MFMA, 1: scalar-FP32 VALU

template <int V>
void run(const char* name, int reps, hipStream_t sa, hipStream_t sb, Stat* st, float* sink) {
  for (int corun = 0; corun < 2; ++corun) {
    CK(hipMemset(st, 0, sizeof(Stat)));
    for (int r = 0; r < reps; ++r) {
      if (corun)
        for (int k = 0; k < 3; ++k) {
          if (g_burner == 0) hipLaunchKernelGGL(burn_mfma, dim3(512), dim3(256), 0, sb, 300, sink);
          else hipLaunchKernelGGL(burn_valu, dim3(512), dim3(256), 0, sb, 300, sink);
        }
      hipLaunchKernelGGL(probe<V>, dim3(512), dim3(256), 0, sa, 2000, st);
      CK(hipStreamSynchronize(sa));
    }
    CK(hipDeviceSynchronize());
    Stat h;
    CK(hipMemcpy(&h, st, sizeof(Stat), hipMemcpyDeviceToHost));
    const double total = (double)reps * 512 * 256 * 2000;
    printf("%-44s %-20s: wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63: %u / %u / %u / %u of %.3g;  equal to the op without its half swap: %u, other: %u\n",
           name, corun ? (g_burner ? "with the VALU burner" : "with the MFMA burner") : "alone", h.quarter[0], h.quarter[1], h.quarter[2], h.quarter[3], total,
           h.unswapped, h.other);
    if (h.nrec) printf("      by iteration of the wave: first %u, 2nd-10th %u, 11th-100th %u, later %u\n", h.iterb[0], h.iterb[1], h.iterb[2], h.iterb[3]);
    for (unsigned k = 0; k < h.nrec && k < 4; ++k) {
      const Rec& q = h.rec[k];
      printf("      thread %d lane %d iter %d: x (%.9g, %.9g) y (%.9g, %.9g) -> (%.9g, %.9g), expected (%.9g, %.9g)\n", q.thread,
             q.thread & 63, q.iter, q.x0, q.x1, q.y0, q.y1, q.r0, q.r1, q.e0, q.e1);
    }
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  Stat* st; float* sink;
  CK(hipMalloc(&st, sizeof(Stat))); CK(hipMalloc(&sink, 4));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  if (argc > 2 && !strcmp(argv[2], "round2")) {   // only the forms added in round 2 (+ one known-bad form as the positive control)
    run<11>("v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]", reps, a, b, st, sink);
    run<12>("v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]", reps, a, b, st, sink);
    run<13>("v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1]", reps, a, b, st, sink);
    run<14>("v_pk_add_f32 neg_hi:[0,1]", reps, a, b, st, sink);
    run<15>("v_pk_fma_f32 (no modifiers)", reps, a, b, st, sink);
    run<16>("v_pk_mul_f32 (no modifiers)", reps, a, b, st, sink);
    run<17>("v_pk_mul_f32 op_sel_hi:[1,0]", reps, a, b, st, sink);
    run<18>("v_pk_fma_f32 op_sel_hi:[1,0,1]", reps, a, b, st, sink);
    run<19>("v_pk_fma_f32 op_sel_hi:[0,1,1]", reps, a, b, st, sink);
    run<20>("v_pk_fma_f32 op_sel_hi:[1,1,0]", reps, a, b, st, sink);
    run<21>("v_pk_add_f32 op_sel_hi:[0,1]", reps, a, b, st, sink);
    run<22>("v_pk_mul_f32 op_sel_hi:[0,1]", reps, a, b, st, sink);
    run<23>("v_pk_fma_f32 op_sel:[1,0,0]", reps, a, b, st, sink);
    run<24>("v_pk_add_f32 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]", reps, a, b, st, sink);
    run<0>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (control)", reps, a, b, st, sink);
    return 0;
  }
  run<4>("v_pk_add_f32 (no modifiers)", reps, a, b, st, sink);
  run<5>("v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]", reps, a, b, st, sink);
  run<0>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", reps, a, b, st, sink);
  run<1>("  the same between 16-cycle s_nops", reps, a, b, st, sink);
  run<6>("v_pk_add_f32 op_sel:[0,0] op_sel_hi:[1,0]", reps, a, b, st, sink);
  run<7>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]", reps, a, b, st, sink);
  run<2>("v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", reps, a, b, st, sink);
  run<3>("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", reps, a, b, st, sink);
  run<10>("v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]", reps, a, b, st, sink);
  run<8>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", reps, a, b, st, sink);
  run<9>("v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]", reps, a, b, st, sink);
  g_burner = 1;
  run<0>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", reps, a, b, st, sink);
  return 0;
}
