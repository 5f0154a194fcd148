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
//     with v_mfma_f32_16x16x32_bf16; the weights of step j+1 are prefetched into registers during step j and
//     staged in LDS with the same XOR swizzle;
//   * sp_j never round-trips through memory: the epilogue (ReLU -> folded BN) stores it to the output slice AND
//     adds the prefetched x_{j+1} to form the next u in registers.
// Utterances longer than 640 frames (or f32) take the generic per-step path in conv_gemm.hip.
#include <stdlib.h>

#include "common.h"

namespace sugar {
namespace {

constexpr int R2_TPW_MAX = 5;        // frame tiles (of 16) per wave: 3 (T <= 384) or 5 (T <= 640)
constexpr int R2_WAVES = 8;
constexpr int R2_MAX_T = 16 * R2_TPW_MAX * R2_WAVES;   // 640 frames

struct Res2Args {
  const bf16* x;     // first_conv output [R][ld]
  bf16* y;           // Res2 output [R][ld] (groups 0..6 written here)
  const bf16* w[SUGAR_RES2_SCALE - 1];        // packed [64][k_pad]
  const float* scale[SUGAR_RES2_SCALE - 1];
  const float* shift[SUGAR_RES2_SCALE - 1];
  const int32_t* row_off;
  int ld, wp, ktaps, dil, k_pad;
};

constexpr int R2_WREG = 5;           // 16-byte weight chunks each thread stages per step (k_pad <= 320)

template <int MT, int R2_TPW>
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
  const int wstride = a.k_pad * 2;                      // weight rows: groups of 8 chunks, chunk ^= row & 7 inside a group
  const int wchunks = a.k_pad >> 3;                     // chunks per weight row (multiple of 8)
  unsigned char* u = smem;
  unsigned char* wl = smem + u_bytes;
  const int nks = a.k_pad >> 5;                         // k-steps of 32

  // zero the whole u tile once: halo rows and frames >= T stay zero for the whole chain
  for (int i = tid; i < u_bytes / 16; i += 512) ((u32x4*)u)[i] = (u32x4){0u, 0u, 0u, 0u};

  int ntw = 0;                                          // tiles owned by this wave: t = wv + 8 n
#pragma unroll
  for (int n = 0; n < R2_TPW; ++n) ntw += (wv + R2_WAVES * n) < NT;

  // per-thread weight staging slots: chunk idx = tid + 512 i of the [64][wchunks] panel
  int wsrc[R2_WREG], wdst[R2_WREG];
#pragma unroll
  for (int i = 0; i < R2_WREG; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / wchunks, c = idx - row * wchunks;
    wsrc[i] = idx < 64 * wchunks ? row * a.k_pad + c * 8 : -1;
    wdst[i] = row * wstride + (((c & ~7) | ((c & 7) ^ (row & 7))) << 4);
  }
  u32x4 wreg[R2_WREG];
  auto load_w = [&](int j) {
#pragma unroll
    for (int i = 0; i < R2_WREG; ++i) wreg[i] = *(const u32x4*)(a.w[j] + max(wsrc[i], 0));   // unconditional, see load_x
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < R2_WREG; ++i)
      if (wsrc[i] >= 0) *(u32x4*)(wl + wdst[i]) = wreg[i];
  };
  // xu holds x_j on load and is turned in place into u_{j+1} = x_{j+1} + sp_j by the epilogue
  bf16x4 xu[R2_TPW][MT];
  auto load_x = [&](int j) {
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n) {
      const int r = (wv + R2_WAVES * n) * 16 + f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ci = m * 16 + 4 * q;
        // unconditional load from a clamped (always valid) address, then select: a branch around each load would
        // make hipcc wait vmcnt(0) per element and serialise 12 memory round trips per step
        const bf16x4 v = *(const bf16x4*)(a.x + (size_t)(s0 + min(r, T - 1)) * a.ld + j * wp + min(ci, wp - 4));
        const bool ok = n < ntw && r < T && ci < wp;
        xu[n][m] = ok ? v : (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
      }
    }
  };
  auto write_u = [&]() {
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n) {
      const int r = (wv + R2_WAVES * n) * 16 + f;
      const int row = r + halo;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ci = m * 16 + 4 * q;
        if (n < ntw && r < T && ci < wp)
          *(bf16x4*)(u + row * urow + (((ci >> 3) ^ (row & swz_mask)) << 4) + (ci & 7) * 2) = xu[n][m];
      }
    }
  };

  load_w(0);
  load_x(0);                                            // u_0 = x_0
  __syncthreads();                                      // zero fill complete before the first u write
  for (int j = 0; j < SUGAR_RES2_SCALE - 1; ++j) {
    const bool last = j + 1 == SUGAR_RES2_SCALE - 1;
    write_u();
    store_w();
    // LDS-only barriers: waves exchange data through LDS alone, so only the LDS counter is drained; a full
    // __syncthreads() would also wait (vmcnt(0)) for this step's output stores and the prefetches in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!last) {                                        // both in flight during the MFMA loop
      load_w(j + 1);
      load_x(j + 1);
    }
    f32x4 sc[MT], sh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      sc[m] = *(const f32x4*)(a.scale[j] + m * 16 + 4 * q);
      sh[m] = *(const f32x4*)(a.shift[j] + m * 16 + 4 * q);
    }

    f32x4 acc[R2_TPW][MT];
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int tap = (q * 8) / wp, ci = q * 8 - tap * wp;      // this lane's K chunk c = 4 ks + q, walked incrementally
    for (int ks = 0; ks < nks; ++ks) {
      const int c = 4 * ks + q;
      const bool kpad = tap >= a.ktaps;                 // K padding: weights are zero, read any valid row
      const int roff = halo + (kpad ? 0 : (tap - ctr) * a.dil);
      const int cch = kpad ? 0 : (ci >> 3);
      bf16x8 af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        af[m] = *(const bf16x8*)(wl + (m * 16 + f) * wstride + (((c & ~7) | ((c & 7) ^ (f & 7))) << 4));
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n) {
        if (n < ntw) {
          const int row = (wv + R2_WAVES * n) * 16 + f + roff;
          const bf16x8 bf = *(const bf16x8*)(u + row * urow + ((cch ^ (row & swz_mask)) << 4));
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf, acc[n][m], 0, 0, 0);
        }
      }
      ci += 32;
      while (ci >= wp) { ci -= wp; ++tap; }
    }

    // epilogue: lane (f, q) holds frame t*16 + f, channels m*16 + 4q + {0..3}
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = m * 16 + 4 * q;
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n) {
        if (n < ntw) {
          const int r = (wv + R2_WAVES * n) * 16 + f;
          f32x4 v = acc[n][m];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
          v = v * sc[m] + sh[m];
          const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          if (r < T && co < wp) *(bf16x4*)(a.y + (size_t)(s0 + r) * a.ld + j * wp + co) = o;
          if (!last) {
            // next conv input: x_{j+1} + sp_j (the reference adds in f32, dynamictdnn.py:298)
            const bf16x4 xv = xu[n][m];
            xu[n][m] = (bf16x4){(bf16)((float)xv[0] + v[0]), (bf16)((float)xv[1] + v[1]), (bf16)((float)xv[2] + v[2]),
                                (bf16)((float)xv[3] + v[3])};
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves done reading u and the weights
  }
}

}  // namespace

bool res2_chain_eligible(int dtype, int t_max, int wp, int k_pad, int n_pad) {
  if (dtype != SUGAR_BF16 || t_max > R2_MAX_T || wp > 64 || wp % 8 != 0 || n_pad != 64) return false;
  if (getenv("SUGAR_GENERIC_RES2")) return false;   // A/B switch for the parity tests: force the per-step path
  const int u_bytes = ((t_max + 15) / 16 * 16 + 2 * 10) * wp * 2 + 256;
  return k_pad <= 64 * R2_WREG && u_bytes + 64 * k_pad * 2 <= 160 * 1024;
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
  const int smem = ((u_rows * wp * 2 + 255) & ~255) + 64 * a.k_pad * 2;
  const int mt = (wp + 15) / 16;
  const bool small = t_max <= 16 * 3 * R2_WAVES;        // 3 frame tiles per wave cover 384 frames
#define R2_CASE(M, TP)                                                                                         \
  {                                                                                                            \
    static bool attr_set = false;                                                                              \
    if (!attr_set) {                                                                                           \
      SUGAR_HIP(hipFuncSetAttribute((const void*)res2_chain_kernel<M, TP>,                                     \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                  \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL((res2_chain_kernel<M, TP>), dim3(B), dim3(512), smem, s, a);                            \
  }
  switch (mt * 2 + (small ? 0 : 1)) {
    case 2: R2_CASE(1, 3) break;
    case 3: R2_CASE(1, 5) break;
    case 4: R2_CASE(2, 3) break;
    case 5: R2_CASE(2, 5) break;
    case 6: R2_CASE(3, 3) break;
    case 7: R2_CASE(3, 5) break;
    case 8: R2_CASE(4, 3) break;
    default: R2_CASE(4, 5) break;
  }
#undef R2_CASE
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
