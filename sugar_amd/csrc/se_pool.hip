// Squeeze-excite, attentive statistics pooling and the folded embedding tail (gfx950).
//
//  se_scale_kernel  : DynamicSELayer.forward (sugar/models/dynamictdnn.py:94-97): mean over an utterance's true
//                     frames -> Linear -> ReLU -> Linear -> sigmoid, one workgroup per utterance.
//  se_apply_kernel  : the channel scale (:98-99) fused with the dense residual bookkeeping of
//                     DynamicTDNN.forward (:657-660):  out = y * s + S ;  S = S + out.
//  pool_stats_kernel: DynamicAttentiveStatsPool.forward (:505-515) after linear1+tanh: the linear2 GEMM
//                     (K = 128) runs on MFMA with the accumulator laid out [frame][channel], so each lane owns one
//                     channel and 16 frames of a 32-frame tile: softmax over time, sum(alpha h) and sum(alpha h^2)
//                     are an in-lane online-softmax recurrence; logits and alpha are never written to memory and
//                     h (the largest activation of the network) is read exactly once.
//  tail_kernel      : bnpool -> linear -> bnembed (:664-665) folded into one affine map at pack time.
//  l2norm_kernel    : F.normalize(p=2, dim=1) (sugar/vectors.py:143-144).
#include "common.h"

namespace sugar {

namespace {

template <typename T>
struct Chunk;
template <>
struct Chunk<float> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void load(const float* p, float* v) {
    const f32x4 t = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = t[i];
  }
  __device__ __forceinline__ static void store(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *(f32x4*)p = t;
  }
};
template <>
struct Chunk<bf16> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void load(const bf16* p, float* v) {
    const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ __forceinline__ static void store(bf16* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *(bf16x8*)p = t;
  }
};

// ------------------------------------------------------------------------------------------ SE scale
template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(const T* __restrict__ y, int ldy,
                                                       const int32_t* __restrict__ row_off, SELayer se,
                                                       float* __restrict__ scale) {
  constexpr int E = Chunk<T>::N;
  extern __shared__ float sm[];          // partial [G][c_p] (<= 256*E floats) | mean [c_p] | hidden [hid]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int s = row_off[b], n = row_off[b + 1] - s;
  const int nch = se.c_p / E;            // <= 256 (checked on the host)
  const int G = 256 / nch;
  const int ch = tid % nch, g = tid / nch;
  float* partial = sm;
  float* mean = sm + 256 * E;
  float* hidden = mean + se.c_p;
  float acc[E];
#pragma unroll
  for (int i = 0; i < E; ++i) acc[i] = 0.f;
  if (g < G) {
    const T* __restrict__ p = y + (size_t)s * ldy + ch * E;
    for (int r = g; r < n; r += G) {
      float v[E];
      Chunk<T>::load(p + (size_t)r * ldy, v);
#pragma unroll
      for (int i = 0; i < E; ++i) acc[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < E; ++i) partial[g * se.c_p + ch * E + i] = acc[i];
  }
  __syncthreads();
  for (int c = tid; c < se.c_p; c += 256) {
    float t = 0.f;
    for (int k = 0; k < G; ++k) t += partial[k * se.c_p + c];
    mean[c] = t / (float)n;
  }
  __syncthreads();
  for (int j = tid; j < se.hid; j += 256) {
    float t = se.b1[j];
    for (int i = 0; i < se.c; ++i) t = fmaf(se.w1t[(size_t)i * se.hid + j], mean[i], t);
    hidden[j] = fmaxf(t, 0.f);
  }
  __syncthreads();
  for (int c = tid; c < se.c_p; c += 256) {
    float t = se.b2[c];
    for (int j = 0; j < se.hid; ++j) t = fmaf(se.w2t[(size_t)j * se.c_p + c], hidden[j], t);
    scale[(size_t)b * se.c_p + c] = 1.0f / (1.0f + expf(-t));
  }
}

// ------------------------------------------------------------------------------------------ SE apply
constexpr int APPLY_ROWS = 32;

template <typename T>
__global__ __launch_bounds__(256) void se_apply_kernel(const T* __restrict__ y, int ldy,
                                                       const float* __restrict__ scale, int c_p, T* __restrict__ S,
                                                       int ldS, T* __restrict__ out, int ldo, int write_S,
                                                       const int32_t* __restrict__ row_off) {
  constexpr int E = Chunk<T>::N;
  const int b = blockIdx.y;
  const int s = row_off[b], n = row_off[b + 1] - s;
  const int r0 = blockIdx.x * APPLY_ROWS;
  if (r0 >= n) return;
  const int rows = min(APPLY_ROWS, n - r0);
  const int nch = c_p / E;
  const float* __restrict__ sc = scale + (size_t)b * c_p;
  for (int item = threadIdx.x; item < rows * nch; item += 256) {
    const int r = item / nch, ch = item - r * nch;
    const size_t row = (size_t)(s + r0 + r);
    float vy[E], vs[E], vo[E];
    Chunk<T>::load(y + row * ldy + ch * E, vy);
    Chunk<T>::load(S + row * ldS + ch * E, vs);
#pragma unroll
    for (int i = 0; i < E; ++i) vo[i] = fmaf(vy[i], sc[ch * E + i], vs[i]);
    // the reference adds in f32; the running sum re-reads the *stored* (rounded) block output, as the next
    // block does, so out and S stay consistent in the compute dtype
    Chunk<T>::store(out + row * ldo + ch * E, vo);
    if (write_S) {
#pragma unroll
      for (int i = 0; i < E; ++i) vs[i] += to_f32(from_f32<T>(vo[i]));
      Chunk<T>::store(S + row * ldS + ch * E, vs);
    }
  }
}

// ------------------------------------------------------------------------------------------ attentive stats
template <typename T>
struct PoolMma;
template <>
struct PoolMma<bf16> {
  typedef bf16x8 Frag;
  static constexpr int NG = 8;   // 128 / 16
  __device__ __forceinline__ static void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct PoolMma<float> {
  typedef f32x4 Frag;
  static constexpr int NG = 16;  // 128 / 8
  __device__ __forceinline__ static void mma(f32x16& acc, const Frag& a, const Frag& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void pool_stats_kernel(const T* __restrict__ a, int lda, const T* __restrict__ h,
                                                         int ldh, const T* __restrict__ w2, int ccat_p,
                                                         const int32_t* __restrict__ row_off,
                                                         float* __restrict__ pooled) {
  constexpr int E = Chunk<T>::N;
  constexpr int NG = PoolMma<T>::NG;
  constexpr int BOTT = 128;
  typedef typename PoolMma<T>::Frag Frag;
  const int b = blockIdx.y;
  const int s = row_off[b], n = row_off[b + 1] - s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int co = blockIdx.x * 128 + wv * 32 + fr;          // this lane's channel (w2 rows are padded to 128)

  Frag wf[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wf[g] = *(const Frag*)(w2 + (size_t)co * BOTT + (2 * g + fh) * E);

  const bool co_ok = co < ccat_p;
  float m = -1e30f, l = 0.f, s1 = 0.f, s2 = 0.f;
  for (int t0 = 0; t0 < n; t0 += 32) {
    // logits tile [32 frames][32 channels]: A = tanh(linear1) rows, B = linear2 weight rows
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int ar = t0 + fr;
    const bool a_ok = ar < n;
    const T* __restrict__ ap = a + (size_t)(s + (a_ok ? ar : 0)) * lda + fh * E;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      Frag af;
      if (a_ok) af = *(const Frag*)(ap + 2 * g * E);
      else {
#pragma unroll
        for (int i = 0; i < E; ++i) af[i] = (T)0.f;
      }
      PoolMma<T>::mma(acc, af, wf[g]);
    }
    float hv[16];
    float tmax = -1e30f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = t0 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      const bool ok = r < n && co_ok;
      hv[e] = ok ? to_f32(h[(size_t)(s + r) * ldh + co]) : 0.f;
      acc[e] = ok ? acc[e] : -1e30f;
      tmax = fmaxf(tmax, acc[e]);
    }
    if (tmax > m) {
      const float f = __expf(m - tmax);
      l *= f; s1 *= f; s2 *= f;
      m = tmax;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = acc[e] > -1e29f ? __expf(acc[e] - m) : 0.f;
      l += p;
      const float ph = p * hv[e];
      s1 += ph;
      s2 = fmaf(ph, hv[e], s2);
    }
  }
  // merge the two half-waves (same channel, disjoint frames)
  const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32), s12 = __shfl_xor(s1, 32), s22 = __shfl_xor(s2, 32);
  const float M = fmaxf(m, m2);
  const float f1 = __expf(m - M), f2 = __expf(m2 - M);
  l = l * f1 + l2 * f2;
  s1 = s1 * f1 + s12 * f2;
  s2 = s2 * f1 + s22 * f2;
  if (fh == 0 && co_ok) {
    const float mean = s1 / l;
    const float var = s2 / l - mean * mean;
    pooled[(size_t)b * 2 * ccat_p + co] = mean;
    pooled[(size_t)b * 2 * ccat_p + ccat_p + co] = sqrtf(fmaxf(var, 1e-9f));
  }
}

// ------------------------------------------------------------------------------------------ tail
constexpr int TAIL_U = 4;

__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ wt, const float* __restrict__ bias,
                                                   int in_p, int embed, const float* __restrict__ pooled, int B,
                                                   float* __restrict__ emb) {
  __shared__ float red[4][TAIL_U][64];
  const int o = blockIdx.y * 64 + (threadIdx.x & 63), kl = threadIdx.x >> 6;
  const int b0 = blockIdx.x * TAIL_U;
  const bool o_ok = o < embed;
  float acc[TAIL_U];
#pragma unroll
  for (int u = 0; u < TAIL_U; ++u) acc[u] = 0.f;
  const float* pu[TAIL_U];
#pragma unroll
  for (int u = 0; u < TAIL_U; ++u) pu[u] = pooled + (size_t)min(b0 + u, B - 1) * in_p;
  for (int i = kl; i < in_p; i += 4) {
    const float w = o_ok ? wt[(size_t)i * embed + o] : 0.f;
#pragma unroll
    for (int u = 0; u < TAIL_U; ++u) acc[u] = fmaf(w, pu[u][i], acc[u]);
  }
#pragma unroll
  for (int u = 0; u < TAIL_U; ++u) red[kl][u][threadIdx.x & 63] = acc[u];
  __syncthreads();
  if (kl == 0 && o_ok) {
#pragma unroll
    for (int u = 0; u < TAIL_U; ++u) {
      if (b0 + u < B) {
        const int x = threadIdx.x & 63;
        emb[(size_t)(b0 + u) * embed + o] = red[0][u][x] + red[1][u][x] + red[2][u][x] + red[3][u][x] + bias[o];
      }
    }
  }
}

__global__ __launch_bounds__(64) void l2norm_kernel(float* __restrict__ x, int dim) {
  float* row = x + (size_t)blockIdx.x * dim;
  float ss = 0.f;
  for (int i = threadIdx.x; i < dim; i += 64) ss = fmaf(row[i], row[i], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize eps
  for (int i = threadIdx.x; i < dim; i += 64) row[i] *= inv;
}

}  // namespace

int launch_se_scale(const SELayer& se, const void* d_y, int ldy, const int32_t* d_row_off, int B, float* d_scale,
                    int dtype, hipStream_t s) {
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(se.c_p % E == 0 && se.c_p / E <= 256 && se.c_p / E > 0, "se: channels %d unsupported (max %d)",
                  se.c_p, 256 * E);
  SUGAR_CHECK_ARG(se.hid > 0 && ldy % E == 0, "se: bad hidden %d / ld %d", se.hid, ldy);
  const size_t smem = (256 * E + se.c_p + se.hid) * sizeof(float);
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(se_scale_kernel<float>, dim3(B), dim3(256), smem, s, (const float*)d_y, ldy, d_row_off, se,
                       d_scale);
  else
    hipLaunchKernelGGL(se_scale_kernel<bf16>, dim3(B), dim3(256), smem, s, (const bf16*)d_y, ldy, d_row_off, se,
                       d_scale);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

int launch_se_apply(const void* d_y, int ldy, const float* d_scale, int c_p, void* d_S, int ldS, void* d_out,
                    int ldo, int write_S, const int32_t* d_row_off, int B, int t_max, int dtype, hipStream_t s) {
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(c_p % E == 0 && ldy % E == 0 && ldS % E == 0 && ldo % E == 0, "se_apply: unaligned channels");
  const dim3 grid((unsigned)ceil_div(t_max, APPLY_ROWS), (unsigned)B);
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(se_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)d_y, ldy, d_scale, c_p,
                       (float*)d_S, ldS, (float*)d_out, ldo, write_S, d_row_off);
  else
    hipLaunchKernelGGL(se_apply_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)d_y, ldy, d_scale, c_p,
                       (bf16*)d_S, ldS, (bf16*)d_out, ldo, write_S, d_row_off);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

int launch_pool_stats(const PoolLayer& P, const void* d_a, int lda, const void* d_h, int ldh,
                      const int32_t* d_row_off, int B, float* d_pooled, int dtype, hipStream_t s) {
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(P.bott == 128, "pool: bottleneck %d unsupported (reference fixes 128, dynamictdnn.py:494)", P.bott);
  SUGAR_CHECK_ARG(lda % E == 0 && ((uintptr_t)d_a & 15) == 0, "pool: unaligned bottleneck activations");
  const dim3 grid((unsigned)ceil_div(P.ccat_p, 128), (unsigned)B);
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(pool_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)d_a, lda, (const float*)d_h,
                       ldh, (const float*)P.w2, P.ccat_p, d_row_off, d_pooled);
  else
    hipLaunchKernelGGL(pool_stats_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)d_a, lda, (const bf16*)d_h, ldh,
                       (const bf16*)P.w2, P.ccat_p, d_row_off, d_pooled);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

int launch_tail(const TailLayer& t, const float* d_pooled, int B, float* d_emb, hipStream_t s) {
  const dim3 grid((unsigned)ceil_div(B, TAIL_U), (unsigned)ceil_div(t.embed, 64));
  hipLaunchKernelGGL(tail_kernel, grid, dim3(256), 0, s, t.wt, t.bias, t.in_p, t.embed, d_pooled, B, d_emb);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

int launch_l2norm(float* d_x, int64_t n, int dim, hipStream_t s) {
  SUGAR_CHECK_ARG(d_x && n > 0 && n < (1ll << 31) && dim > 0, "l2norm: bad arguments");
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)n), dim3(64), 0, s, d_x, dim);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
