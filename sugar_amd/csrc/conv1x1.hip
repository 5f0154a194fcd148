// 1x1 conv as a 256 x 256 x 64 MFMA GEMM with direct global->LDS staging (gfx950, bf16 operands / f32 accumulate).
//
// The 512- and 1536-channel 1x1 convs of EfficientTDNN (first_conv / last_conv of every SE-Res2 block,
// sugar/models/dynamictdnn.py:392-394,402-404, and catconv :633,663) carry 82 % of the MACs.  They have no taps and
// no utterance boundaries, so the activation operand needs no per-row masks and can be staged by LDS-DMA
// (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
//
//   tile      256 out channels (M, MFMA A operand = weight panel) x 256 frames (N, B operand), BK = 64
//   waves     8 = 2 (channels) x 4 (frames); each owns 128 x 64 outputs = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16
//   LDS       2 stages x (32 KB weights + 32 KB activations); rows of 8 x 16-byte chunks, chunk XOR (row>>1)&7 on the
//             per-lane SOURCE address (the DMA writes LDS lane-linearly) and the same XOR on the fragment reads
//   pipeline  issue the DMA of K-tile t+1, compute K-tile t (24 ds_read_b128 + 64 MFMA per wave), vmcnt(0) + barrier
//   epilogue  ReLU -> folded-BN affine in registers, transpose through LDS, 16-byte-per-lane row-contiguous stores
//
// 256-wide tiles make the kernel MFMA-bound rather than L2-bound (128 FLOP per staged byte); co-tiles that share
// an activation row tile are placed on one XCD (xcd_remap) so the re-read hits that XCD's L2.
#include "common.h"

namespace sugar {

namespace {

struct Gemm1x1Args {
  const bf16* x;
  const bf16* w;
  bf16* y;
  bf16* y2;
  const float* scale;
  const float* shift;
  int ldx, ldw, ldy, ldy2, y2_from;
  int R, cout, K, flags;
  int n_co_tiles;
};

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int STAGE_BYTES = (TM + TN) * 128;   // 64 KB
constexpr int EPI_STRIDE = 272;                // bytes per staged output row (256 + 16 pad)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, seq = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
}

__global__ __launch_bounds__(512, 2) void gemm1x1_kernel(Gemm1x1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 2, wn = wv & 3;
  const int vid = xcd_remap2(blockIdx.x, gridDim.x);
  const int co_base = (vid % a.n_co_tiles) * TM;
  const int r_base = (vid / a.n_co_tiles) * TN;

  // ---- DMA addressing: each wave fills 32 rows of both operand tiles, 8 rows (1 KiB) per instruction -----------
  const int srow = lane >> 3;                       // row within the 8-row slab
  const bf16* wsrc[4];
  const bf16* xsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wv * 32 + i * 8 + srow;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);          // source chunk that belongs at LDS position lane&7
    wsrc[i] = a.w + (size_t)(co_base + row) * a.ldw + chunk * 8;
    const int xr = min(r_base + row, a.R - 1);                // rows past R: any valid row, masked at the store
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

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = a.K / TK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const unsigned char* wb = smem + cur * STAGE_BYTES;
    const unsigned char* xb = wb + TM * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[8], bf[4];
#pragma unroll
      for (int m = 0; m < 8; ++m) af[m] = *(const bf16x8*)(wb + swz(wm * 128 + m * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int n = 0; n < 4; ++n) bf[n] = *(const bf16x8*)(xb + swz(wn * 64 + n * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------------------------
  // accumulator tile (m, n): lane (fr, fq) holds frame wn*64 + n*16 + fr, channels wm*128 + m*16 + 4 fq + {0..3}
  const bool relu = a.flags & EPI_RELU, affine = a.flags & EPI_AFFINE;
  unsigned char* ep = smem + wv * (32 * EPI_STRIDE);          // this wave's [32 frames][128 channels] staging
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int col = m * 16 + 4 * fq;                        // channel within the wave's 128
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (affine) {
        sc = *(const f32x4*)(a.scale + co_base + wm * 128 + col);
        sh = *(const f32x4*)(a.shift + co_base + wm * 128 + col);
      }
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        f32x4 v = acc[m][half * 2 + nn];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        v = v * sc + sh;
        const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *(bf16x4*)(ep + (nn * 16 + fr) * EPI_STRIDE + col * 2) = o;
      }
    }
    // the staging region is private to the wave: no workgroup barrier, only the LDS counter
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = i * 4 + fq;                               // frame within the half (0..31)
      const int r = r_base + wn * 64 + half * 32 + f;
      const int co = co_base + wm * 128 + fr * 8;
      const u32x4 v = *(const u32x4*)(ep + f * EPI_STRIDE + fr * 16);
      if (r < a.R && co < a.cout) {
        *(u32x4*)(a.y + (size_t)r * a.ldy + co) = v;
        if (a.y2 && co >= a.y2_from) *(u32x4*)(a.y2 + (size_t)r * a.ldy2 + co) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// Eligibility: bf16, single tap, no addend, K and the padded weight rows tile-aligned, >128 output channels.
bool gemm1x1_eligible(const ConvLayer& L, const ConvIO& io, int dtype) {
  return dtype == SUGAR_BF16 && L.ktaps == 1 && !io.x2 && !(L.flags & EPI_TANH) && L.cin_p % TK == 0 &&
         L.k_pad == L.cin_p && L.n_pad % TM == 0 && L.cout > 128 && L.cout % 8 == 0 && io.ldy % 8 == 0 &&
         ((uintptr_t)io.y & 15) == 0 && (!io.y2 || (io.ldy2 % 8 == 0 && ((uintptr_t)io.y2 & 15) == 0 && io.y2_from % 8 == 0));
}

int launch_gemm1x1(const ConvLayer& L, const ConvIO& io, int64_t R, hipStream_t s) {
  Gemm1x1Args a;
  a.x = (const bf16*)io.x; a.w = (const bf16*)L.w; a.y = (bf16*)io.y; a.y2 = (bf16*)io.y2;
  a.scale = L.scale; a.shift = L.shift;
  a.ldx = io.ldx; a.ldw = L.k_pad; a.ldy = io.ldy; a.ldy2 = io.ldy2; a.y2_from = io.y2_from;
  a.R = (int)R; a.cout = L.cout; a.K = L.cin_p; a.flags = L.flags;
  a.n_co_tiles = (int)ceil_div(L.cout, TM);
  const int grid = a.n_co_tiles * (int)ceil_div(R, TN);
  const int smem = 2 * STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    SUGAR_HIP(hipFuncSetAttribute((const void*)gemm1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm1x1_kernel, dim3(grid), dim3(512), smem, s, a);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
