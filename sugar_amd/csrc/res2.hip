// Res2 chain in one kernel (gfx950, bf16 operands / f32 accumulate).
//
// DynamicRes2Conv1dReLUBN.forward (sugar/models/dynamictdnn.py:290-303): the first_conv output is split into 8 groups
// of `w` channels; 7 SEQUENTIAL dilated convs   sp_j = BN(ReLU(conv_j(x_j + sp_{j-1})))   (sp_{-1} = 0), the 8th group
// passes through.  As 7 launches each step is a tiny GEMM (M = w <= 64) and the chain is launch/latency-bound (the
// reference's V100 table shows ~3.2 ms per block whatever the width, tdnn8m2g_latency_v100_80_300.yml:7155).
//
// Here one workgroup (8 waves) owns one utterance for the whole chain:
//   * u = x_j + sp_{j-1} lives in LDS as [frame + halo][w] bf16 (zero halo rows = the conv's "same" padding at the
//     utterance's true ends), chunk-swizzled so the shifted fragment reads of every tap are bank-conflict free;
//   * each wave owns up to 5 tiles of 16 frames; D[channel][frame] = sum over (tap, ci) W[co][tap*w+ci] u[frame+off][ci]
//     with v_mfma_f32_16x16x32_bf16, weights of step j staged in LDS (rows padded by one chunk: conflict-free);
//   * sp_j never round-trips through memory: the epilogue (ReLU -> folded BN) stores it to the output slice AND
//     adds the prefetched x_{j+1} to form the next u in registers.
// Utterances longer than 640 frames (or f32) take the generic per-step path in conv_gemm.hip.
#include <stdlib.h>

#include "common.h"

namespace sugar {
namespace {

constexpr int R2_TPW = 5;            // frame tiles (of 16) per wave
constexpr int R2_WAVES = 8;
constexpr int R2_MAX_T = 16 * R2_TPW * R2_WAVES;   // 640 frames

struct Res2Args {
  const bf16* x;     // first_conv output [R][ld]
  bf16* y;           // Res2 output [R][ld] (groups 0..6 written here)
  const bf16* w[SUGAR_RES2_SCALE - 1];        // packed [64][k_pad]
  const float* scale[SUGAR_RES2_SCALE - 1];
  const float* shift[SUGAR_RES2_SCALE - 1];
  const int32_t* row_off;
  int ld, wp, ktaps, dil, k_pad;
};

template <int MT>
__global__ __launch_bounds__(512, 2) void res2_chain_kernel(Res2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int s0 = a.row_off[b], T = a.row_off[b + 1] - s0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int f = lane & 15, q = lane >> 4;
  const int wp = a.wp, ctr = (a.ktaps - 1) >> 1, halo = ctr * a.dil;
  const int NT = (T + 15) >> 4;
  const int urow = wp * 2;                              // bytes per u row
  const int nch = wp >> 3;                              // 16-byte chunks per u row
  const int swz_mask = (nch & (nch - 1)) == 0 ? (nch - 1 < 7 ? nch - 1 : 7) : 0;
  const int u_rows = NT * 16 + 2 * halo;
  const int u_bytes = (u_rows * urow + 255) & ~255;
  const int wstride = a.k_pad * 2 + 16;                 // weight row stride in LDS (one chunk of padding)
  unsigned char* u = smem;
  unsigned char* wl = smem + u_bytes;
  const int nks = a.k_pad >> 5;                         // k-steps of 32

  // zero the whole u tile once: halo rows and frames >= T stay zero for the whole chain
  for (int i = tid; i < u_bytes / 16; i += 512) ((u32x4*)u)[i] = (u32x4){0u, 0u, 0u, 0u};

  int ntw = 0;                                          // tiles owned by this wave: t = wv + 8 n
#pragma unroll
  for (int n = 0; n < R2_TPW; ++n) ntw += (wv + R2_WAVES * n) < NT;

  auto load_x = [&](int j, bf16x4 (&xin)[R2_TPW][MT]) {
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n) {
      const int r = (wv + R2_WAVES * n) * 16 + f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ci = m * 16 + 4 * q;
        bf16x4 v = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        if (n < ntw && r < T && ci < wp) v = *(const bf16x4*)(a.x + (size_t)(s0 + r) * a.ld + j * wp + ci);
        xin[n][m] = v;
      }
    }
  };
  auto stage_w = [&](int j) {
    const int chunks_per_row = a.k_pad >> 3;
    for (int i = tid; i < 64 * chunks_per_row; i += 512) {
      const int row = i / chunks_per_row, c = i - row * chunks_per_row;
      *(u32x4*)(wl + row * wstride + c * 16) = *(const u32x4*)(a.w[j] + (size_t)row * a.k_pad + c * 8);
    }
  };
  auto write_u = [&](const bf16x4 (&un)[R2_TPW][MT]) {
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n) {
      const int r = (wv + R2_WAVES * n) * 16 + f;
      const int row = r + halo;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ci = m * 16 + 4 * q;
        if (n < ntw && r < T && ci < wp)
          *(bf16x4*)(u + row * urow + (((ci >> 3) ^ (row & swz_mask)) << 4) + (ci & 7) * 2) = un[n][m];
      }
    }
  };

  bf16x4 xin[R2_TPW][MT], un[R2_TPW][MT];
  load_x(0, un);                                        // u_0 = x_0
  __syncthreads();                                      // zero fill complete before the first u write
  for (int j = 0; j < SUGAR_RES2_SCALE - 1; ++j) {
    write_u(un);
    stage_w(j);
    __syncthreads();
    if (j + 1 < SUGAR_RES2_SCALE - 1) load_x(j + 1, xin);   // in flight during the MFMA loop

    f32x4 acc[R2_TPW][MT];
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ks = 0; ks < nks; ++ks) {
      const int c = 4 * ks + q;                         // this lane's K chunk
      int e = c * 8, tap = e / wp, ci = e - tap * wp;
      if (tap >= a.ktaps) { tap = ctr; ci = 0; }        // K padding: weights are zero, read any valid row
      const int roff = halo + (tap - ctr) * a.dil;
      bf16x8 af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *(const bf16x8*)(wl + (m * 16 + f) * wstride + c * 16);
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n) {
        if (n < ntw) {
          const int row = (wv + R2_WAVES * n) * 16 + f + roff;
          const bf16x8 bf = *(const bf16x8*)(u + row * urow + (((ci >> 3) ^ (row & swz_mask)) << 4));
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf, acc[n][m], 0, 0, 0);
        }
      }
    }

    // epilogue: lane (f, q) holds frame t*16 + f, channels m*16 + 4q + {0..3}
    const bool last = j + 1 == SUGAR_RES2_SCALE - 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = m * 16 + 4 * q;
      const f32x4 sc = *(const f32x4*)(a.scale[j] + co), sh = *(const f32x4*)(a.shift[j] + co);
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n) {
        if (n < ntw) {
          const int r = (wv + R2_WAVES * n) * 16 + f;
          f32x4 v = acc[n][m];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
          v = v * sc + sh;
          const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          if (r < T && co < wp) *(bf16x4*)(a.y + (size_t)(s0 + r) * a.ld + j * wp + co) = o;
          if (!last) {
            // next conv input: x_{j+1} + sp_j (the reference adds in f32, dynamictdnn.py:298)
            const bf16x4 xv = xin[n][m];
            un[n][m] = (bf16x4){(bf16)((float)xv[0] + v[0]), (bf16)((float)xv[1] + v[1]), (bf16)((float)xv[2] + v[2]),
                                (bf16)((float)xv[3] + v[3])};
          }
        }
      }
    }
    __syncthreads();                                    // every wave is done reading u and the staged weights
  }
}

}  // namespace

bool res2_chain_eligible(int dtype, int t_max, int wp, int k_pad, int n_pad) {
  if (dtype != SUGAR_BF16 || t_max > R2_MAX_T || wp > 64 || wp % 8 != 0 || n_pad != 64) return false;
  if (getenv("SUGAR_GENERIC_RES2")) return false;   // A/B switch for the parity tests: force the per-step path
  const int u_bytes = ((t_max + 15) / 16 * 16 + 2 * 10) * wp * 2 + 256;
  return u_bytes + 64 * (k_pad * 2 + 16) <= 160 * 1024;
}

int launch_res2_chain(const ConvLayer* steps, const void* d_x, void* d_y, int ld, int wp, const int32_t* d_row_off,
                      int B, int t_max, hipStream_t s) {
  Res2Args a;
  a.x = (const bf16*)d_x; a.y = (bf16*)d_y; a.row_off = d_row_off;
  a.ld = ld; a.wp = wp; a.ktaps = steps[0].ktaps; a.dil = steps[0].dil; a.k_pad = steps[0].k_pad;
  for (int j = 0; j < SUGAR_RES2_SCALE - 1; ++j) {
    SUGAR_CHECK_ARG(steps[j].k_pad == a.k_pad && steps[j].ktaps == a.ktaps && steps[j].scale && steps[j].n_pad == 64,
                    "res2: inconsistent step %d", j);
    a.w[j] = (const bf16*)steps[j].w; a.scale[j] = steps[j].scale; a.shift[j] = steps[j].shift;
  }
  const int halo = ((a.ktaps - 1) / 2) * a.dil;
  const int u_rows = (t_max + 15) / 16 * 16 + 2 * halo;
  const int smem = ((u_rows * wp * 2 + 255) & ~255) + 64 * (a.k_pad * 2 + 16);
  const int mt = (wp + 15) / 16;
  const void* fn = mt == 1 ? (const void*)res2_chain_kernel<1> : mt == 2 ? (const void*)res2_chain_kernel<2>
                 : mt == 3 ? (const void*)res2_chain_kernel<3> : (const void*)res2_chain_kernel<4>;
  static bool attr_set[5] = {false, false, false, false, false};
  if (!attr_set[mt]) {
    SUGAR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[mt] = true;
  }
  switch (mt) {
    case 1: hipLaunchKernelGGL(res2_chain_kernel<1>, dim3(B), dim3(512), smem, s, a); break;
    case 2: hipLaunchKernelGGL(res2_chain_kernel<2>, dim3(B), dim3(512), smem, s, a); break;
    case 3: hipLaunchKernelGGL(res2_chain_kernel<3>, dim3(B), dim3(512), smem, s, a); break;
    default: hipLaunchKernelGGL(res2_chain_kernel<4>, dim3(B), dim3(512), smem, s, a); break;
  }
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
