// Attentive statistics pooling (gfx950): DynamicAttentiveStatsPool.forward (sugar/models/dynamictdnn.py:505-515)
// after linear1 + tanh, as ONE kernel:
//     logits = W2 . a          (K = 128 bottleneck, MFMA)
//     alpha  = softmax over the utterance's frames, per channel
//     mean   = sum_t alpha h,   std = sqrt(clamp(sum_t alpha h^2 - mean^2, 1e-9))
// The accumulator is laid out [frame][channel] (A operand = a rows, B operand = W2 rows), so every lane owns one
// channel and 16 frames of each 32-frame tile: softmax / weighted sums are an in-lane online-softmax recurrence,
// logits and alpha never exist in memory, and h -- the largest activation of the network -- is read exactly once.
//
// One workgroup = 8 waves = (utterance, 256 channels).  Per 32-frame tile the a tile [32 x 128] and the h tile
// [32 x 256] are staged through LDS with 16-byte coalesced loads (register prefetch of tile t+1 during tile t, two LDS
// buffers, one barrier per tile); chunk index XOR (row & 15) keeps the ds_read_b128 fragment reads conflict-free.
//
// Utterances longer than POOL_SEG = 1024 frames are cut into segments of 1024 frames, one workgroup each (a 60 s
// utterance is a serial loop of 188 tiles otherwise, and a batch of 16 of them fills a third of the chip): a segment
// leaves its online-softmax state (max, sum p, sum p h, sum p h^2) in a table and pool_merge_kernel folds the
// segments of an utterance in order.  The segment length is fixed, so an utterance's statistics do not depend on
// what else is in the batch; utterances of one segment are finished by the main kernel as before.
#include <type_traits>

#include "common.h"

namespace sugar {
namespace {

template <typename T>
struct PoolMma;
template <>
struct PoolMma<bf16> {
  typedef bf16x8 Frag;
  static constexpr int E = 8;
  static constexpr int NG = 8;   // k-groups: 128 / 16
  __device__ __forceinline__ static void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct PoolMma<float> {
  typedef f32x4 Frag;
  static constexpr int E = 4;
  static constexpr int NG = 16;  // 128 / 8
  __device__ __forceinline__ static void mma(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
  }
};

constexpr int PT = 32;     // frames per tile
constexpr int NT = 512;    // threads per workgroup
constexpr int PC = NT / 2;    // channels per workgroup (32 per wave)
constexpr int BOTT = 128;
constexpr int POOL_SEG = 1024;   // frames per time segment (multiple of PT)

__device__ __forceinline__ int pool_slot(const int32_t* row_off, int b) { return row_off[b] / POOL_SEG + b; }

// launch index -> work index, bijective on [0, nwg): the workgroups of XCD x (launch indices x, x + 8, ...) get a
// contiguous run of work indices
__device__ __forceinline__ int pool_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, seq = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
}

template <typename T>
__global__ __launch_bounds__(NT, sizeof(T) == 2 ? 4 : 2) void pool_stats_kernel(const T* __restrict__ a, int lda, const T* __restrict__ h,
                                                         int ldh, const T* __restrict__ w2, int ccat_p,
                                                         const int32_t* __restrict__ row_off,
                                                         float* __restrict__ pooled, float* __restrict__ part) {
  constexpr int E = PoolMma<T>::E;
  constexpr int NG = PoolMma<T>::NG;
  constexpr int A_CH = BOTT / E;                 // 16-byte chunks per a row   (16 bf16 / 32 f32)
  constexpr int H_CH = PC / E;                   // chunks per h row slab      (32 / 64)
  constexpr int A_BYTES = PT * BOTT * (int)sizeof(T);
  constexpr int H_BYTES = PT * PC * (int)sizeof(T);
  constexpr int A_PER = PT * A_CH / NT;         // chunks each thread stages  (1 / 2)
  constexpr int H_PER = PT * H_CH / NT;         //                            (2 / 4)
  typedef typename PoolMma<T>::Frag Frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x (a tile | h tile)

  // The channel slabs of an utterance all read its a rows.  Workgroups go to the 8 XCDs round-robin in launch order, so
  // with (slab, utterance) = (blockIdx.x, blockIdx.y) the slabs of one utterance sit on different XCDs and a comes from
  // HBM once per slab (PMC: 362 MB fetched for 257 MB algorithmic).  Renumber the launch plane so that consecutive
  // workgroups of ONE XCD take the slabs of one utterance: a is fetched once and re-read from that XCD's L2.
  const int n_slab = gridDim.x, plane = gridDim.x * gridDim.y;
  const int vid = pool_xcd_remap(blockIdx.x + n_slab * blockIdx.y, plane);
  const int b = vid / n_slab, slab = vid - b * n_slab;
  const int s_utt = row_off[b], n_utt = row_off[b + 1] - s_utt;
  const int seg0 = blockIdx.z * POOL_SEG;
  if (seg0 >= n_utt) return;
  const int s = s_utt + seg0, n = min(POOL_SEG, n_utt - seg0);   // this workgroup's frames
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int c_base = slab * PC;
  const int co = c_base + wv * 32 + fr;          // this lane's channel (W2 rows are padded to a multiple of 256)
  const bool co_ok = co < ccat_p;

  Frag wf[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wf[g] = *(const Frag*)(w2 + (size_t)co * BOTT + (2 * g + fh) * E);
  // the weights are in registers from here on: without this hipcc carries their loads' vmcnt bookkeeping into the tile
  // loop and makes the MFMAs wait for the tile prefetch issued at the top of the same iteration
#pragma unroll
  for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(wf[g]));

  // two tiles of global loads in flight per thread (register sets q = 0 / 1; 96 KiB per CU on their way instead of
  // 48).  Measured with the two changes above in one job, 256 x 3 s: 0.0672 -> 0.0650 ms.  Knock-outs of the same
  // build: no MFMAs 0.0485, no softmax arithmetic 0.0492, no LDS staging 0.0519, no loads 0.0577,
  // neither arithmetic nor loads 0.0368 -- the parts add up, the 8 lock-stepped waves of a workgroup overlap little
  u32x4 areg[2][A_PER], hreg[2][H_PER];
  auto load_tile = [&](auto qt, int t0) {
    constexpr int q = decltype(qt)::value;
    // past the last tile (uniform): every row index collapses to the segment's first row -- the same three load
    // instructions are issued (L2 hits on 0.5 KiB), so the vmcnt arithmetic of store_tile below is the same on every
    // path; with a branch around the loads hipcc waits for ALL outstanding loads there, the newest tile's included
    const int rmax = t0 < n ? n - 1 : 0;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT, row = idx / A_CH, ch = idx % A_CH;
      // unconditional load from a clamped row; the zero-select happens in store_tile, a tile or two later, so nothing
      // waits on the load here and it stays in flight under the MFMAs and exps of the tiles before
      areg[q][i] = *(const u32x4*)(a + (size_t)(s + min(t0 + row, rmax)) * lda + ch * E);
    }
#pragma unroll
    for (int i = 0; i < H_PER; ++i) {
      const int idx = tid + i * NT, row = idx / H_CH, ch = idx % H_CH;
      const int cc = min(c_base + ch * E, ccat_p - E);
      hreg[q][i] = *(const u32x4*)(h + (size_t)(s + min(t0 + row, rmax)) * ldh + cc);
    }
  };
  auto store_tile = [&](auto qt, int buf, int t0) {
    constexpr int q = decltype(qt)::value;
    unsigned char* ab = smem + buf * (A_BYTES + H_BYTES);
    unsigned char* hb = ab + A_BYTES;
    const bool full = t0 + PT <= n && c_base + PC <= ccat_p;      // (uniform) no row or channel of the tile to zero
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int idx = tid + i * NT, row = idx / A_CH, ch = idx % A_CH;
      *(u32x4*)(ab + row * (A_CH * 16) + ((ch ^ (row & 15)) << 4)) = full || t0 + row < n ? areg[q][i] : (u32x4){0u, 0u, 0u, 0u};
    }
    if (full) {
#pragma unroll
      for (int i = 0; i < H_PER; ++i) *(u32x4*)(hb + (tid + i * NT) * 16) = hreg[q][i];
    } else {
#pragma unroll
      for (int i = 0; i < H_PER; ++i) {
        const int idx = tid + i * NT, row = idx / H_CH, ch = idx % H_CH;
        const bool ok = t0 + row < n && c_base + ch * E < ccat_p;
        *(u32x4*)(hb + idx * 16) = ok ? hreg[q][i] : (u32x4){0u, 0u, 0u, 0u};        // plain [row][channel] image
      }
    }
  };

  float m = -1e30f, l = 0.f, s1 = 0.f, s2 = 0.f;
  const int ntiles = (n + PT - 1) / PT;
  typedef std::integral_constant<int, 0> Q0;
  typedef std::integral_constant<int, 1> Q1;
  load_tile(Q0{}, 0);
  load_tile(Q1{}, PT);
  store_tile(Q0{}, 0, 0);
  __syncthreads();
  // tile `it` is computed from LDS buffer it & 1; its register set (it & 1) is free and takes tile it + 2, while the
  // other set holds tile it + 1, requested one iteration ago and staged into the other buffer at the end of this one
  auto step = [&](auto qt, int it) {
    constexpr int q = decltype(qt)::value;
    const int t0 = it * PT;
    load_tile(std::integral_constant<int, q>{}, t0 + 2 * PT);
    const unsigned char* ab = smem + (it & 1) * (A_BYTES + H_BYTES);
    const T* hs = (const T*)(ab + A_BYTES);
    const T* hl = hs + (4 * fh) * PC + wv * 32 + fr;    // frame e of this lane: hl[((e & 3) + 8 * (e >> 2)) * PC]
    // logits tile [32 frames][32 channels]
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    constexpr int FB = 4;                           // a-fragments requested ahead of the MFMAs that consume them
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += FB) {
      Frag af[FB];
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        const int ch = 2 * (g0 + j) + fh;
        af[j] = *(const Frag*)(ab + fr * (A_CH * 16) + ((ch ^ (fr & 15)) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);            // (hipcc otherwise sinks every read to right before its MFMA)
#pragma unroll
      for (int j = 0; j < FB; ++j) PoolMma<T>::mma(acc, af[j], wf[g0 + j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // online softmax over this tile's 16 frames of the lane's channel, in the exp2 domain (v_exp_f32 is 2^x): m is the
    // running maximum of logit * log2(e).  Only the utterance's last tile can hold frames past n: the row masks are
    // compiled into that tile's copy of the body alone (the loop was VALU-issue-bound: ~300 VALU per tile and wave,
    // a third of them row tests, selects and LDS address arithmetic).
    auto body = [&](auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      constexpr float LOG2E = 1.4426950408889634f;
      float tmax = -1e30f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (LAST) acc[e] = t0 + (e & 3) + 8 * (e >> 2) + 4 * fh < n ? acc[e] : -1e30f;
        tmax = fmaxf(tmax, acc[e]);
      }
      tmax *= LOG2E;
      if (tmax > m) {
        const float f = __builtin_amdgcn_exp2f(m - tmax);
        l *= f; s1 *= f; s2 *= f;
        m = tmax;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float hv = to_f32(hl[((e & 3) + 8 * (e >> 2)) * PC]);      // zero-filled past n / ccat_p
        float p = __builtin_amdgcn_exp2f(fmaf(acc[e], LOG2E, -m));
        if (LAST) p = acc[e] > -1e29f ? p : 0.f;
        l += p;
        const float ph = p * hv;
        s1 += ph;
        s2 = fmaf(ph, hv, s2);
      }
    };
    if (t0 + PT <= n) body(std::false_type{});
    else body(std::true_type{});
    if (it + 1 < ntiles) store_tile(std::integral_constant<int, q ^ 1>{}, (it + 1) & 1, t0 + PT);
    __syncthreads();
  };
  for (int it = 0; it < ntiles; it += 2) {
    step(Q0{}, it);
    if (it + 1 < ntiles) step(Q1{}, it + 1);
  }
  // merge the two half-waves (same channel, disjoint frames)
  const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32), s12 = __shfl_xor(s1, 32), s22 = __shfl_xor(s2, 32);
  const float M = fmaxf(m, m2);                    // (maxima in the exp2 domain: logit * log2(e))
  const float f1 = __builtin_amdgcn_exp2f(m - M), f2 = __builtin_amdgcn_exp2f(m2 - M);
  const float lt = l * f1 + l2 * f2, s1t = s1 * f1 + s12 * f2, s2t = s2 * f1 + s22 * f2;
  if (n_utt > POOL_SEG) {                          // one of several segments: leave the state to pool_merge_kernel
    if (fh == 0 && co_ok) {
      float* p = part + (size_t)(pool_slot(row_off, b) + blockIdx.z) * 4 * ccat_p + co;
      p[0] = M; p[ccat_p] = lt; p[2 * ccat_p] = s1t; p[3 * ccat_p] = s2t;
    }
    return;
  }
  if (fh == 0 && co_ok) {
    const float mean = s1t / lt;
    const float var = s2t / lt - mean * mean;
    pooled[(size_t)b * 2 * ccat_p + co] = mean;
    pooled[(size_t)b * 2 * ccat_p + ccat_p + co] = sqrtf(fmaxf(var, 1e-9f));
  }
}

__global__ __launch_bounds__(256) void pool_merge_kernel(const float* __restrict__ part, int ccat_p,
                                                         const int32_t* __restrict__ row_off, float* __restrict__ pooled) {
  const int b = blockIdx.y, co = blockIdx.x * 256 + threadIdx.x;
  const int n = row_off[b + 1] - row_off[b];
  if (n <= POOL_SEG || co >= ccat_p) return;
  const int nseg = (n + POOL_SEG - 1) / POOL_SEG;
  const float* p = part + (size_t)pool_slot(row_off, b) * 4 * ccat_p + co;
  float m = p[0], l = p[ccat_p], s1 = p[2 * ccat_p], s2 = p[3 * ccat_p];
  for (int k = 1; k < nseg; ++k) {
    const float* q = p + (size_t)k * 4 * ccat_p;
    const float m2 = q[0], M = fmaxf(m, m2);       // (exp2 domain, as pool_stats_kernel leaves them)
    const float f1 = __builtin_amdgcn_exp2f(m - M), f2 = __builtin_amdgcn_exp2f(m2 - M);
    l = l * f1 + q[ccat_p] * f2;
    s1 = s1 * f1 + q[2 * ccat_p] * f2;
    s2 = s2 * f1 + q[3 * ccat_p] * f2;
    m = M;
  }
  const float mean = s1 / l;
  const float var = s2 / l - mean * mean;
  pooled[(size_t)b * 2 * ccat_p + co] = mean;
  pooled[(size_t)b * 2 * ccat_p + ccat_p + co] = sqrtf(fmaxf(var, 1e-9f));
}

template <typename T>
int launch_t(const PoolLayer& P, const void* d_a, int lda, const void* d_h, int ldh, const int32_t* d_row_off, int B,
             int t_max, float* d_pooled, float* d_part, hipStream_t s) {
  const int smem = 2 * PT * (BOTT + PC) * (int)sizeof(T);
  static OptInMask opted{0};
  if (int rc = lds_opt_in((const void*)pool_stats_kernel<T>, smem, &opted)) return rc;
  const dim3 grid((unsigned)ceil_div(P.ccat_p, PC), (unsigned)B, (unsigned)ceil_div(t_max, POOL_SEG));
  hipLaunchKernelGGL(pool_stats_kernel<T>, grid, dim3(NT), smem, s, (const T*)d_a, lda, (const T*)d_h, ldh,
                     (const T*)P.w2, P.ccat_p, d_row_off, d_pooled, d_part);
  SUGAR_LAUNCH_CHECK();
  if (t_max > POOL_SEG) {
    hipLaunchKernelGGL(pool_merge_kernel, dim3((unsigned)ceil_div(P.ccat_p, 256), (unsigned)B), dim3(256), 0, s, d_part,
                       P.ccat_p, d_row_off, d_pooled);
    SUGAR_LAUNCH_CHECK();
  }
  return SUGAR_OK;
}

}  // namespace

int64_t pool_part_floats(const PoolLayer& P, int B, int64_t R) { return (R / POOL_SEG + B + 1) * 4 * (int64_t)P.ccat_p; }

int launch_pool_stats(const PoolLayer& P, const void* d_a, int lda, const void* d_h, int ldh,
                      const int32_t* d_row_off, int B, int t_max, float* d_pooled, float* d_part, int dtype,
                      hipStream_t s) {
  SUGAR_CHECK_ARG(t_max > 0 && (t_max <= POOL_SEG || d_part), "pool: segment table missing (t_max %d)", t_max);
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(P.bott == BOTT, "pool: bottleneck %d unsupported (reference fixes 128, dynamictdnn.py:494)", P.bott);
  SUGAR_CHECK_ARG(lda % E == 0 && ldh % E == 0 && P.ccat_p % E == 0 && ((uintptr_t)d_a & 15) == 0 &&
                      ((uintptr_t)d_h & 15) == 0,
                  "pool: unaligned activations");
  if (dtype == SUGAR_F32) return launch_t<float>(P, d_a, lda, d_h, ldh, d_row_off, B, t_max, d_pooled, d_part, s);
  return launch_t<bf16>(P, d_a, lda, d_h, ldh, d_row_off, B, t_max, d_pooled, d_part, s);
}

}  // namespace sugar
