// Stand-alone A/B bench for the wide 1x1-conv GEMM (catconv shape), variants interleaved in ONE process.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/gemm_bench.hip -o tools/gemm_bench   Run: tools/gemm_bench
// Variant bits: 1 no DMA in the loop, 2 no MFMA, 4 no epilogue stores, 8 setprio around MFMA clusters,
//               16 raw s_barrier (no __syncthreads), 32 explicit fragment double-buffer, 64 32x32x16 MFMA
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Args {
  const bf16* x; const bf16* w; bf16* y;
  int ldx, ldw, ldy, R, cout, K, n_co_tiles;
};

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int STAGE_BYTES = (TM + TN) * 128;
constexpr int EPI_STRIDE = 272;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, seq = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
}

template <int V>
__global__ __launch_bounds__(512, 2) void gemm_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 2, wn = wv & 3;
  const int vid = xcd_remap2(blockIdx.x, gridDim.x);
  const int co_base = (vid % a.n_co_tiles) * TM;
  const int r_base = (vid / a.n_co_tiles) * TN;
  const int srow = lane >> 3;
  const bf16* wsrc[4];
  const bf16* xsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wv * 32 + i * 8 + srow;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    wsrc[i] = a.w + (size_t)(co_base + row) * a.ldw + chunk * 8;
    const int xr = min(r_base + row, a.R - 1);
    xsrc[i] = a.x + (size_t)xr * a.ldx + chunk * 8;
  }
  auto stage = [&](int buf, int kt) {
    unsigned char* base = smem + buf * STAGE_BYTES + wv * 32 * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * TK),
                                       (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + kt * TK),
                                       (__attribute__((address_space(3))) void*)(base + TM * 128 + i * 1024), 16, 0, 0);
  };
  auto barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (V & 16) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } else { __syncthreads(); }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = a.K / TK;

  if constexpr (V & 64) {
    // 32x32x16 MFMA: wave tile 128 x 64 = 4 x 2 tiles, lane (r32 = lane&31, h = lane>>5), chunk = 2*g + h, g = 0..3
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
    const int r32 = lane & 31, h = lane >> 5;
    stage(0, 0);
    barrier();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (!(V & 1) && kt + 1 < nk) stage(cur ^ 1, kt + 1);
      const unsigned char* wb = smem + ((V & 1) ? 0 : cur) * STAGE_BYTES;
      const unsigned char* xb = wb + TM * 128;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x8 af[4], bf[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *(const bf16x8*)(wb + swz(wm * 128 + m * 32 + r32, 2 * g + h));
#pragma unroll
        for (int n = 0; n < 2; ++n) bf[n] = *(const bf16x8*)(xb + swz(wn * 64 + n * 32 + r32, 2 * g + h));
        if constexpr (V & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
        if constexpr (V & 8) __builtin_amdgcn_s_setprio(0);
      }
      barrier();
    }
    // simple direct epilogue (8-byte stores), only to keep the accumulators live
    if (!(V & 4)) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = r_base + wn * 64 + n * 32 + r32, co = co_base + wm * 128 + m * 32 + 8 * q + 4 * h;
            const bf16x4 o = {(bf16)acc[m][n][4 * q], (bf16)acc[m][n][4 * q + 1], (bf16)acc[m][n][4 * q + 2], (bf16)acc[m][n][4 * q + 3]};
            if (r < a.R) *(bf16x4*)(a.y + (size_t)r * a.ldy + co) = o;
          }
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) asm volatile("" ::"v"(acc[m][n]));
    }
    return;
  } else {
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    barrier();
    auto load_frags = [&](const unsigned char* wb, const unsigned char* xb, int s, bf16x8 (&af)[8], bf16x8 (&bf)[4]) {
#pragma unroll
      for (int m = 0; m < 8; ++m) af[m] = *(const bf16x8*)(wb + swz(wm * 128 + m * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int n = 0; n < 4; ++n) bf[n] = *(const bf16x8*)(xb + swz(wn * 64 + n * 16 + fr, 4 * s + fq));
    };
    auto mma = [&](const bf16x8 (&af)[8], const bf16x8 (&bf)[4]) {
      if constexpr (V & 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(af[m]));
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(bf[n]));
      } else {
        if constexpr (V & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
        if constexpr (V & 8) __builtin_amdgcn_s_setprio(0);
      }
    };
    if constexpr (V & 768) {
      // 256: X operand through registers (global_load_dwordx4 -> ds_write_b128), W by LDS-DMA; 512: both through registers
      const bf16* xlin[4];
      const bf16* wlin[4];
      int dst[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wv * 32 + i * 8 + srow, c = lane & 7;
        xlin[i] = a.x + (size_t)min(r_base + row, a.R - 1) * a.ldx + c * 8;
        wlin[i] = a.w + (size_t)(co_base + row) * a.ldw + c * 8;
        dst[i] = swz(row, c);
      }
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        u32x4 xr[4], wr[4];
        if (more) {
          if (!(V & 512)) {
            unsigned char* base = smem + (cur ^ 1) * STAGE_BYTES + wv * 32 * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + (kt + 1) * TK),
                                               (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) xr[i] = *(const u32x4*)(xlin[i] + (kt + 1) * TK);
          if (V & 512) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wr[i] = *(const u32x4*)(wlin[i] + (kt + 1) * TK);
          }
        }
        const unsigned char* wb = smem + cur * STAGE_BYTES;
        const unsigned char* xb = wb + TM * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 af[8], bf[4];
          load_frags(wb, xb, s, af, bf);
          mma(af, bf);
        }
        if (more) {
          unsigned char* nb = smem + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
          for (int i = 0; i < 4; ++i) *(u32x4*)(nb + TM * 128 + dst[i]) = xr[i];
          if (V & 512) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(nb + dst[i]) = wr[i];
          }
        }
        barrier();
      }
    } else
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (!(V & 1) && !((V & 128) && wm == 1) && kt + 1 < nk) stage(cur ^ 1, kt + 1);
      const unsigned char* wb = smem + ((V & 1) ? 0 : cur) * STAGE_BYTES;
      const unsigned char* xb = wb + TM * 128;
      if constexpr (V & 128) {
        // staggered DMA issue: waves 0-3 issue before their MFMAs, waves 4-7 between their two MFMA halves
        {
          bf16x8 af[8], bf[4];
          load_frags(wb, xb, 0, af, bf);
          mma(af, bf);
        }
        if (wm == 1 && kt + 1 < nk) stage(cur ^ 1, kt + 1);
        {
          bf16x8 af[8], bf[4];
          load_frags(wb, xb, 1, af, bf);
          mma(af, bf);
        }
      } else if constexpr (V & 32) {
        bf16x8 af0[8], bf0[4], af1[8], bf1[4];
        load_frags(wb, xb, 0, af0, bf0);
        load_frags(wb, xb, 1, af1, bf1);
        mma(af0, bf0);
        mma(af1, bf1);
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 af[8], bf[4];
          load_frags(wb, xb, s, af, bf);
          mma(af, bf);
        }
      }
      barrier();
    }
    unsigned char* ep = smem + wv * (32 * EPI_STRIDE);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int col = m * 16 + 4 * fq;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
          f32x4 v = acc[m][half * 2 + nn];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)(ep + (nn * 16 + fr) * EPI_STRIDE + col * 2) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = i * 4 + fq;
        const int r = r_base + wn * 64 + half * 32 + f;
        const int co = co_base + wm * 128 + fr * 8;
        const u32x4 v = *(const u32x4*)(ep + f * EPI_STRIDE + fr * 16);
        if constexpr (V & 4) { asm volatile("" ::"v"(v)); } else {
          if (r < a.R && co < a.cout) *(u32x4*)(a.y + (size_t)r * a.ldy + co) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- 5-slot, 32-deep software pipeline: DMA 4 sub-tiles ahead, fragments one sub-tile ahead of the MFMAs -----
constexpr int SK = 32;
constexpr int SUB_BYTES = (TM + TN) * 64;   // 32 KB
constexpr int NSLOT = 5;
__device__ __forceinline__ int swz5(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }

// V bits: 4 no stores, 8 setprio, 1 interleave hint (sched_group_barrier)
template <int V>
__global__ __launch_bounds__(512, 2) void gemm5_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 2, wn = wv & 3;
  const int vid = xcd_remap2(blockIdx.x, gridDim.x);
  const int co_base = (vid % a.n_co_tiles) * TM;
  const int r_base = (vid / a.n_co_tiles) * TN;
  const bf16* wsrc[2];
  const bf16* xsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wv * 32 + i * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((0 - (row >> 2)) & 3);
    wsrc[i] = a.w + (size_t)(co_base + row) * a.ldw + chunk * 8;
    const int xr = min(r_base + row, a.R - 1);
    xsrc[i] = a.x + (size_t)xr * a.ldx + chunk * 8;
  }
  auto stage = [&](int slot, int t) {
    unsigned char* base = smem + slot * SUB_BYTES + wv * 32 * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + t * SK),
                                       (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + t * SK),
                                       (__attribute__((address_space(3))) void*)(base + TM * 64 + i * 1024), 16, 0, 0);
  };
  const int fr = lane & 15, fq = lane >> 4;
  const int nt = a.K / SK;
  int aoff[8], boff[4];
#pragma unroll
  for (int m = 0; m < 8; ++m) aoff[m] = swz5(wm * 128 + m * 16 + fr, fq);
#pragma unroll
  for (int n = 0; n < 4; ++n) boff[n] = TM * 64 + swz5(wn * 64 + n * 16 + fr, fq);

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 af[2][8], bf[2][4];
  auto load_frags = [&](int slot, bf16x8 (&fa)[8], bf16x8 (&fb)[4]) {
    const unsigned char* sb = smem + slot * SUB_BYTES;
#pragma unroll
    for (int m = 0; m < 8; ++m) fa[m] = *(const bf16x8*)(sb + aoff[m]);
#pragma unroll
    for (int n = 0; n < 4; ++n) fb[n] = *(const bf16x8*)(sb + boff[n]);
  };
  auto mma = [&](const bf16x8 (&fa)[8], const bf16x8 (&fb)[4]) {
    if constexpr (V & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
    if constexpr (V & 8) __builtin_amdgcn_s_setprio(0);
  };

  // prologue: sub-tiles 0..3 in flight (nt >= 4 is guaranteed by the launcher: K >= 128)
#pragma unroll
  for (int t = 0; t < 4; ++t) stage(t, t);
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  load_frags(0, af[0], bf[0]);

  int slot_rd = 1, slot_wr = 4;   // slot of sub-tile t+1, slot of sub-tile t+4
  for (int t = 0; t < nt; t += 2) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int tt = t + p;
      // own DMA of sub-tile tt+1 landed (tt+2, tt+3 may stay in flight: 8 instructions)
      if (tt + 3 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (!(V & 1) && tt + 4 < nt) stage(slot_wr, tt + 4);
      load_frags(slot_rd, af[p ^ 1], bf[p ^ 1]);   // past the end: stale slot, never consumed
      mma(af[p], bf[p]);
      slot_rd = slot_rd == NSLOT - 1 ? 0 : slot_rd + 1;
      slot_wr = slot_wr == NSLOT - 1 ? 0 : slot_wr + 1;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  unsigned char* ep = smem + wv * (32 * EPI_STRIDE);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int col = m * 16 + 4 * fq;
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        f32x4 v = acc[m][half * 2 + nn];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *(bf16x4*)(ep + (nn * 16 + fr) * EPI_STRIDE + col * 2) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = i * 4 + fq;
      const int r = r_base + wn * 64 + half * 32 + f;
      const int co = co_base + wm * 128 + fr * 8;
      const u32x4 v = *(const u32x4*)(ep + f * EPI_STRIDE + fr * 16);
      if constexpr (V & 4) { asm volatile("" ::"v"(v)); } else {
        if (r < a.R && co < a.cout) *(u32x4*)(a.y + (size_t)r * a.ldy + co) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
float run(const Args& a, int grid, int iters) {
  const int smem = 2 * STAGE_BYTES;
  CK(hipFuncSetAttribute((const void*)gemm_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_kernel<V>, dim3(grid), dim3(512), smem, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

template <int V>
float run5(const Args& a, int grid, int iters) {
  const int smem = NSLOT * SUB_BYTES;
  CK(hipFuncSetAttribute((const void*)gemm5_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm5_kernel<V>, dim3(grid), dim3(512), smem, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 77056, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 1536;
  std::vector<uint16_t> hx((size_t)R * K), hw((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; float f = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd();
  bf16 *x, *w, *y;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&y, (size_t)R * N * 2));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  Args a{x, w, y, K, K, N, R, N, K, (N + TM - 1) / TM};
  const int grid = a.n_co_tiles * ((R + TN - 1) / TN);
  const double flop = 2.0 * R * N * K;
  const char* names[] = {"base", "noDMA", "noMFMA", "noStore", "setprio", "rawbar", "fragdbuf", "rawbar+setprio", "rawbar+fragdbuf+setprio",
                         "mfma32", "mfma32+rawbar+setprio", "noDMA+noStore", "noMFMA+noStore"};
  {
    // correctness of the pipelined kernel against the base kernel (same MFMA order: bit-identical)
    std::vector<uint16_t> y0((size_t)R * N), y1((size_t)R * N);
    run<0>(a, grid, 1);
    CK(hipMemcpy(y0.data(), y, y0.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemset(y, 0, y0.size() * 2));
    run5<0>(a, grid, 1);
    CK(hipMemcpy(y1.data(), y, y1.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < y0.size(); ++i) bad += y0[i] != y1[i];
    printf("gemm5 vs base: %zu mismatches of %zu\n", bad, y0.size());
    CK(hipMemset(y, 0, y0.size() * 2));
    run<256>(a, grid, 1);
    CK(hipMemcpy(y1.data(), y, y1.size() * 2, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < y0.size(); ++i) bad += y0[i] != y1[i];
    printf("Xreg vs base: %zu mismatches\n", bad);
    CK(hipMemset(y, 0, y0.size() * 2));
    run<512>(a, grid, 1);
    CK(hipMemcpy(y1.data(), y, y1.size() * 2, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < y0.size(); ++i) bad += y0[i] != y1[i];
    printf("XWreg vs base: %zu mismatches\n", bad);
  }
  for (int round = 0; round < 3; ++round) {
    printf("  Xreg+Wdma    %.4f ms\n", run<256>(a, grid, 10));
    printf("  XWreg        %.4f ms\n", run<512>(a, grid, 10));
    printf("  stagger      %.4f ms\n", run<128>(a, grid, 10));
    printf("  stagger+raw  %.4f ms\n", run<144>(a, grid, 10));
    printf("  base         %.4f ms\n", run<0>(a, grid, 10));
    printf("  gemm5        %.4f ms\n", run5<0>(a, grid, 10));
    printf("  gemm5+prio   %.4f ms\n", run5<8>(a, grid, 10));
    printf("  gemm5 noSt   %.4f ms\n", run5<4>(a, grid, 10));
    printf("  gemm5 noDMA  %.4f ms\n", run5<1>(a, grid, 10));
    printf("  gemm5 noDMA noSt %.4f ms\n", run5<5>(a, grid, 10));
    float t[13];
    t[0] = run<0>(a, grid, 10);
    t[1] = run<1>(a, grid, 10);
    t[2] = run<2>(a, grid, 10);
    t[3] = run<4>(a, grid, 10);
    t[4] = run<8>(a, grid, 10);
    t[5] = run<16>(a, grid, 10);
    t[6] = run<32>(a, grid, 10);
    t[7] = run<24>(a, grid, 10);
    t[8] = run<56>(a, grid, 10);
    t[9] = run<64>(a, grid, 10);
    t[10] = run<88>(a, grid, 10);
    t[11] = run<5>(a, grid, 10);
    t[12] = run<6>(a, grid, 10);
    printf("round %d  R=%d N=%d K=%d grid=%d\n", round, R, N, K, grid);
    for (int i = 0; i < 13; ++i) printf("  %-26s %.4f ms  %.0f TF-equiv\n", names[i], t[i], flop / (t[i] * 1e-3) / 1e12);
  }
  return 0;
}
