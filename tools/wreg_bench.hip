// Weights-in-registers GEMM (gemm_wreg_kernel) against the 128 x 128 tile kernel on the 512-channel conv shape, in one
// process; with -DWREG_KNOCK=bits the kernel leaves parts out (see conv1x1.hip) to show what its time is made of.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Isugar_amd/csrc -Iinclude [-DWREG_KNOCK=n] tools/wreg_bench.hip -o tools/wreg_bench
// -DWLDS [-DWL_EXP=bits] adds gemm_wlds_kernel (below): weights resident in LDS, activations loaded straight into MFMA
// fragment registers -- the layout VERDICT round 2 item 4 asked to be measured; WL_EXP knock-outs: 2 = coalesced (wrong)
// load addresses, 4 = no MFMAs.
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

// ---- round 3 experiment, measured and NOT in the library (-DWLDS builds it; profiles/r03_wlds_knockout.log) ----------
#ifdef WLDS
namespace sugar {
namespace {
// ---------------------------------------------------------------------------------------------------------------------
// gemm_wlds_kernel: K = 512 convs with the WEIGHTS resident in LDS and the ACTIVATIONS loaded straight into MFMA
// fragment registers (round 3; the layout round 2's knock-out table asked for).
//
// Channels-last rows ARE the B-operand layout of v_mfma_f32_16x16x32_bf16: lane (fr, fq)'s 8 k-values of frame fr are 16
// contiguous bytes of that frame's row.  So a wave loads the [32 frames x 512 k] operand of its tile with plain 16-byte
// global loads (the path that streams HBM at 6 TB/s; LDS-DMA reaches 2.4-2.7 cold) into 128 VGPRs, and the
// workgroup keeps a 128-channel weight slab (128 KiB, in fragment order: every ds_read_b128 of a wave is one contiguous
// KiB, conflict-free) in LDS for the whole launch.  What that buys over gemm_wreg_kernel:
//   * LDS traffic per MFMA is halved: one A fragment per NT = 2 MFMAs (weights-in-registers: every wave re-read the whole
//     activation tile, one B fragment per CM = 2 MFMAs, LDS time = MFMA pipe time);
//   * no ring, no barrier, no counted DMA waits in the loop: the 8 waves run on their own tiles and drift apart, so the
//     two waves of a SIMD overlap each other's loads, epilogues and MFMAs by themselves;
//   * the next tile's fragments are requested k-step by k-step into the registers the current tile has just consumed
//     (rolling prefetch: ~15 k-steps of MFMAs = ~2 us of lookahead per load, no second register set).
// Same fragments into the same MFMAs in the same k order as gemm_tile / gemm_wreg_kernel: bit-identical outputs.
// The weight slab is a.wfrag (gemm_wreg_kernel's pre-packed order, 256-channel slabs of 8 waves x 32 channels): slab128
// s = waves 4 (s & 1) .. + 3 of slab256 s >> 1; fragment (c, m8 = 2 wvl + m) goes to LDS block c * 8 + m8.
constexpr int WL_KC = 16, WL_M = 8, WL_SLAB = 128, WL_WBYTES = WL_KC * WL_M * 1024;

// vmcnt bookkeeping: the rolling prefetch keeps 16 NT loads in flight across the tile boundary, and the compiler's
// waitcnt pass gives up counting (vmcnt(0) before the first MFMA of every tile) as soon as stores or scratch accesses
// are pending beside them (gfx9 counts both in vmcnt and the pass treats mixed kinds as out of order).  So the output
// stores are inline asm the pass does not see: its counted waits for the loads stay, and the stores it does not know of
// can only make such a wait stricter (more requests outstanding than it assumes), never too loose -- loads return in
// order among themselves.  A store reads its data registers when it issues (no expcnt hand-shake on gfx9).
__device__ __forceinline__ void wl_store16(uint32_t voff, const u32x4& v, const void* sbase) {
#if defined(WL_EXP) && WL_EXP == 1
  asm volatile("; nostore %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
#endif
}

// D: k-steps of activation fragments in flight per frame tile (a ring over the 16 k-steps: 4 D NT VGPRs).  D = 16 (a whole
// tile, 128 VGPRs at NT = 2) leaves 2 waves per SIMD no room beside 64 accumulators and the weight fragments and spills.
template <int NT, int D>
__global__ __launch_bounds__(512, 1) void gemm_wlds_kernel(Gemm1x1Args a, int n_units, int n_slabs, int P) {
  static_assert(D == 4 || D == 8 || D == 16, "ring depth in k-steps: slot = k-step % D must hold across tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int UF = 16 * NT;                       // frames per wave tile
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // workgroups b, b + 8, b + 16, ... (same XCD) take the channel slabs of one frame range: x comes from HBM once
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int slab = j % n_slabs, p = xcd + 8 * (j / n_slabs);
  const int u0 = (int)((int64_t)p * n_units / P), u1 = (int)((int64_t)(p + 1) * n_units / P);
  if (u0 >= u1) return;

  // ---- weight slab -> LDS (LDS-DMA, 16 KiB per wave), folded-BN vectors behind it ------------------------------
#pragma unroll
  for (int i = 0; i < WL_KC * WL_M / 8; ++i) {
    const int blk = wv * (WL_KC * WL_M / 8) + i, c = blk >> 3, wvl = (blk >> 1) & 3, m = blk & 1;
    const bf16* src = a.wfrag + ((size_t)(((slab * 4 + wvl) * WL_KC + c) * 2 + m) * 64 + lane) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + blk * 1024), 16, 0, 0);
  }
  float* ss = (float*)(smem + WL_WBYTES);           // [scale 128 | shift 128]
  const bool relu = a.flags & EPI_RELU, affine = a.flags & EPI_AFFINE;
  if (tid < 2 * WL_SLAB) {
    const int ch = slab * WL_SLAB + (tid & (WL_SLAB - 1));
    ss[tid] = affine ? (tid < WL_SLAB ? a.scale[ch] : a.shift[ch]) : (tid < WL_SLAB ? 1.f : 0.f);
  }

  // ---- this wave's tiles: units u0 + wv, u0 + wv + 8, ... ; fragment (c, n) of unit u = 16 bytes of row
  //      UF u + 16 n + fr at k = 32 c + 8 fq: a wave-uniform base per unit (SGPRs) + a 32-bit per-lane offset.
  //      Rows past R are clamped to row R - 1 (valid memory) and masked at the store.
  const uint32_t row_bytes = (uint32_t)a.ldx * 2;
  auto unit_base = [&](int u) { return (const unsigned char*)a.x + (size_t)u * UF * row_bytes; };
  auto lane_off = [&](int u, int n) {
#if defined(WL_EXP) && (WL_EXP & 2)   // knock-out: every load instruction reads one contiguous KiB (wrong operands, same bytes)
    return (uint32_t)(min(16 * n + (lane >> 2), a.R - 1 - u * UF)) * row_bytes + 16u * (lane & 3);
#endif
    return (uint32_t)(min(16 * n + fr, a.R - 1 - u * UF)) * row_bytes + 16u * fq;
  };
  u32x4 xf[D][NT];                                   // slot c % D holds k-step c
  int u = u0 + wv;
  uint32_t coff[NT];                                 // this lane's row offsets in the current unit
  {
    const int uu = min(u, u1 - 1);
    const unsigned char* ub = unit_base(uu);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      coff[n] = lane_off(uu, n);
#pragma unroll
      for (int c = 0; c < D; ++c) xf[c][n] = *(const u32x4*)(ub + coff[n] + 64 * c);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // once: the weight slab (and the first tile)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const unsigned char* wl = smem + lane * 16;
  // output: wave-uniform base per unit and frame tile + this lane's 32-bit offset (frame fr, channels 8 fq .. of the slab)
  const uint32_t yrow = (uint32_t)a.ldy * 2, y2row = (uint32_t)a.ldy2 * 2;
  const uint32_t yoff = fr * yrow + (uint32_t)(slab * WL_SLAB + 8 * fq) * 2;
  const uint32_t y2off = fr * y2row + (uint32_t)(slab * WL_SLAB + 8 * fq) * 2;
  for (; u < u1; u += 8) {
    const int un = u + 8 < u1 ? u + 8 : u;                  // next tile of this wave (none: reload this one, unused)
    const unsigned char* cb = unit_base(u);
    const unsigned char* nb = unit_base(un);
    uint32_t noff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) noff[n] = lane_off(un, n);
    f32x4 acc[WL_M][NT];
#pragma unroll
    for (int m = 0; m < WL_M; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // k loop in halves of 4 weight fragments: the next half's ds_reads are in flight during this half's 4 NT MFMAs
    bf16x8 af[2][WL_M / 2];
#pragma unroll
    for (int m = 0; m < WL_M / 2; ++m) af[0][m] = *(const bf16x8*)(wl + m * 1024);
#pragma unroll
    for (int c = 0; c < WL_KC; ++c) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int nx = 2 * c + h + 1;                       // next half: k-step nx >> 1, weight tiles 4 (nx & 1) ..
        if (nx < 2 * WL_KC) {
#pragma unroll
          for (int m = 0; m < WL_M / 2; ++m)
            af[nx & 1][m] = *(const bf16x8*)(wl + ((nx >> 1) * WL_M + (nx & 1) * (WL_M / 2) + m) * 1024);
        }
#pragma unroll
        for (int m = 0; m < WL_M / 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#if defined(WL_EXP) && (WL_EXP & 4)   // knock-out: no MFMAs
            asm volatile("" : "+v"(acc[h * (WL_M / 2) + m][n]) : "v"(af[h][m]), "v"(xf[c % D][n]));
#else
            acc[h * (WL_M / 2) + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                af[h][m], __builtin_bit_cast(bf16x8, xf[c % D][n]), acc[h * (WL_M / 2) + m][n], 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      // k-step c is consumed: its registers take k-step c + D -- of this tile, or of the next one
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (c + D < WL_KC) xf[c % D][n] = *(const u32x4*)(cb + coff[n] + 64 * (c + D));
        else xf[c % D][n] = *(const u32x4*)(nb + noff[n] + 64 * (c + D - WL_KC));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) coff[n] = noff[n];
    // epilogue: lane (fr, fq) holds, over the tile pair (2 pr, 2 pr + 1), channels 32 pr + 8 fq .. + 7 of frame 16 n + fr
    // (wreg_row): ReLU -> folded BN -> bf16, ONE 16-byte store per frame and pair
    const int rows_left = a.R - u * UF;                     // frames of this unit inside the tensor
#pragma unroll
    for (int pr = 0; pr < WL_M / 2; ++pr) {
      const int cl = 32 * pr + 8 * fq, co = slab * WL_SLAB + cl;
      const f32x4 sc0 = *(const f32x4*)(ss + cl), sc1 = *(const f32x4*)(ss + cl + 4);
      const f32x4 sh0 = *(const f32x4*)(ss + WL_SLAB + cl), sh1 = *(const f32x4*)(ss + WL_SLAB + cl + 4);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 v0 = acc[2 * pr][n], v1 = acc[2 * pr + 1][n];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] = relu_f32(v0[e]); v1[e] = relu_f32(v1[e]); }
        }
        v0 = v0 * sc0 + sh0;
        v1 = v1 * sc1 + sh1;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (bf16)v0[e]; o[4 + e] = (bf16)v1[e]; }
        if (16 * n + fr < rows_left) {
          const u32x4 v = __builtin_bit_cast(u32x4, o);
          wl_store16(yoff + 64u * pr, v, (const unsigned char*)a.y + ((size_t)u * UF + 16 * n) * yrow);
          if (a.y2 && co >= a.y2_from)
            wl_store16(y2off + 64u * pr, v, (const unsigned char*)a.y2 + ((size_t)u * UF + 16 * n) * y2row);
        }
      }
    }
  }
}

template <int NT = 2, int D = 8>
int launch_wlds(const Gemm1x1Args& a, hipStream_t s) {
  constexpr int smem = WL_WBYTES + 2 * WL_SLAB * 4;
  static OptInMask opted{0};
  if (int rc = lds_opt_in((const void*)gemm_wlds_kernel<NT, D>, smem, &opted)) return rc;
  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu)) return rc;
  const int n_slabs = (int)ceil_div(a.cout, WL_SLAB), n_units = (int)ceil_div(a.R, 16 * NT);
  int P = n_cu / n_slabs / 8 * 8;
  if (P < 8) P = 8;
  hipLaunchKernelGGL((gemm_wlds_kernel<NT, D>), dim3(n_slabs * P), dim3(512), smem, s, a, n_units, n_slabs, P);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace
}  // namespace sugar
#endif

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
#ifdef WLDS
  const V vs[] = {{"SK4", launch_wreg<KC, CM, 4, NW>}, {"WL2D8", launch_wlds<2, 8>}, {"WL2D4", launch_wlds<2, 4>}, {"WL1D16", launch_wlds<1, 16>}};
#else
  const V vs[] = {{"SK2", launch_wreg<KC, CM, 2, NW>}, {"SK4", launch_wreg<KC, CM, 4, NW>}, {"SK8", launch_wreg<KC, CM, 8, NW>}};
#endif
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
