// Squeeze-excite, attentive statistics pooling and the folded embedding tail (gfx950).
//
//  se_scale_kernel  : DynamicSELayer.forward (sugar/models/dynamictdnn.py:94-97): mean over an utterance's true
//                     frames -> Linear -> ReLU -> Linear -> sigmoid, one workgroup per utterance.
//  se_apply_kernel  : the channel scale (:98-99) fused with the dense residual bookkeeping of
//                     DynamicTDNN.forward (:657-660):  out = y * s + S ;  S = S + out.
//  (attentive statistics pooling lives in pool.hip)
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
// se_mean_kernel: grid (channel slabs of 8 chunks, utterances, time segments); 256 threads = 8 chunks x 32 row lanes.
// Utterances longer than SE_SEG frames are summed in segments of SE_SEG frames (fixed: the result does not depend on
// the rest of the batch); the segment sums go to a table that se_mlp_kernel adds up in order.
constexpr int SE_SEG = 2048;

__device__ __forceinline__ int se_slot(const int32_t* row_off, int b) { return row_off[b] / SE_SEG + b; }

template <typename T>
__global__ __launch_bounds__(256) void se_mean_kernel(const T* __restrict__ y, int ldy, int c_p,
                                                      const int32_t* __restrict__ row_off,
                                                      float* __restrict__ mean, float* __restrict__ part) {
  constexpr int E = Chunk<T>::N;
  __shared__ float partial[32][8 * E + 1];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int s_utt = row_off[b], n_utt = row_off[b + 1] - s_utt;
  const int seg0 = blockIdx.z * SE_SEG;
  if (seg0 >= n_utt) return;
  const int s = s_utt + seg0, n = min(SE_SEG, n_utt - seg0);
  const int tx = tid & 7, ty = tid >> 3;
  const int c0 = (blockIdx.x * 8 + tx) * E;
  float acc[E];
#pragma unroll
  for (int i = 0; i < E; ++i) acc[i] = 0.f;
  if (c0 < c_p) {
    const T* __restrict__ p = y + (size_t)s * ldy + c0;
#pragma unroll 4
    for (int r = ty; r < n; r += 32) {
      float v[E];
      Chunk<T>::load(p + (size_t)r * ldy, v);
#pragma unroll
      for (int i = 0; i < E; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) partial[ty][tx * E + i] = acc[i];
  __syncthreads();
  if (tid < 8 * E) {
    const int c = blockIdx.x * 8 * E + tid;
    if (c < c_p) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) t += partial[k][tid];
      if (n_utt > SE_SEG) part[(size_t)(se_slot(row_off, b) + blockIdx.z) * c_p + c] = t;
      else mean[(size_t)b * c_p + c] = t / (float)n;
    }
  }
}

// se_mlp_kernel: one 1024-thread workgroup per utterance; both tiny GEMVs split over k so no thread walks more
// than ~64 weights, weights stored transposed so every load is coalesced.
__global__ __launch_bounds__(1024) void se_mlp_kernel(const float* __restrict__ mean, const float* __restrict__ segsum,
                                                      const int32_t* __restrict__ row_off, SELayer se,
                                                      float* __restrict__ scale) {
  extern __shared__ float sm[];          // mean [c_p] | hidden [hid] | partial [1024]
  float* m = sm;
  float* hidden = sm + se.c_p;
  float* part = hidden + se.hid;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n_utt = row_off[b + 1] - row_off[b];
  if (n_utt > SE_SEG) {                              // long utterance: add up its segment sums
    const int nseg = (n_utt + SE_SEG - 1) / SE_SEG;
    const float* p = segsum + (size_t)se_slot(row_off, b) * se.c_p;
    for (int c = tid; c < se.c_p; c += 1024) {
      float t = 0.f;
      for (int k = 0; k < nseg; ++k) t += p[(size_t)k * se.c_p + c];
      m[c] = t / (float)n_utt;
    }
  } else {
    for (int c = tid; c < se.c_p; c += 1024) m[c] = mean[(size_t)b * se.c_p + c];
  }
  __syncthreads();
  {  // hidden = relu(W1 mean + b1): thread (j, kp), kp-th slice of the channel range
    const int KP = 1024 / se.hid > 0 ? 1024 / se.hid : 1;
    const int j = tid % se.hid, kp = tid / se.hid;
    float t = 0.f;
    if (kp < KP) {
      const int per = (se.c + KP - 1) / KP;
      const int i0 = kp * per, i1 = min(se.c, i0 + per);
#pragma unroll 8
      for (int i = i0; i < i1; ++i) t = fmaf(se.w1t[(size_t)i * se.hid + j], m[i], t);
    }
    part[tid] = t;
    __syncthreads();
    if (tid < se.hid) {
      float t2 = se.b1[tid];
      for (int k = 0; k < KP; ++k) t2 += part[k * se.hid + tid];
      hidden[tid] = fmaxf(t2, 0.f);
    }
    __syncthreads();
  }
  // scale = sigmoid(W2 hidden + b2): channel blocks of 256, 4 k-slices each
  for (int cb = 0; cb < se.c_p; cb += 256) {
    const int c = cb + (tid & 255), kp = tid >> 8;
    float t = 0.f;
    if (c < se.c_p) {
      const int per = (se.hid + 3) / 4;
      const int j0 = kp * per, j1 = min(se.hid, j0 + per);
#pragma unroll 8
      for (int j = j0; j < j1; ++j) t = fmaf(se.w2t[(size_t)j * se.c_p + c], hidden[j], t);
    }
    part[tid] = t;
    __syncthreads();
    if (tid < 256 && c < se.c_p) {
      const float z = se.b2[c] + part[tid] + part[256 + tid] + part[512 + tid] + part[768 + tid];
      scale[(size_t)b * se.c_p + c] = 1.0f / (1.0f + expf(-z));
    }
    __syncthreads();
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

// ------------------------------------------------------------------------------------------ tail
constexpr int TAIL_U = 16;    // utterances per workgroup
constexpr int TAIL_K = 128;   // pooled features per k-split (24 splits at 3072 features: 1152 workgroups, short dependent-load chains)

// partial[ks][b][o] = sum_{i in k-split ks} wt[i][o] * pooled[b][i]
__global__ __launch_bounds__(256) void tail_partial_kernel(const float* __restrict__ wt, int in_p, int embed,
                                                           const float* __restrict__ pooled, int B,
                                                           float* __restrict__ partial) {
  __shared__ float ps[TAIL_U][TAIL_K];
  __shared__ float red[4][TAIL_U][64];
  const int tid = threadIdx.x, ol = tid & 63, kl = tid >> 6;
  const int o = blockIdx.y * 64 + ol;
  const int b0 = blockIdx.x * TAIL_U;
  const int k0 = blockIdx.z * TAIL_K;
  const int kn = min(TAIL_K, in_p - k0);
  for (int idx = tid; idx < TAIL_U * TAIL_K; idx += 256) {
    const int u = idx / TAIL_K, i = idx - u * TAIL_K;
    ps[u][i] = (b0 + u < B && i < kn) ? pooled[(size_t)(b0 + u) * in_p + k0 + i] : 0.f;
  }
  __syncthreads();
  float acc[TAIL_U];
#pragma unroll
  for (int u = 0; u < TAIL_U; ++u) acc[u] = 0.f;
  const bool o_ok = o < embed;
#pragma unroll 8
  for (int i = kl; i < kn; i += 4) {
    const float w = o_ok ? wt[(size_t)(k0 + i) * embed + o] : 0.f;
#pragma unroll
    for (int u = 0; u < TAIL_U; ++u) acc[u] = fmaf(w, ps[u][i], acc[u]);
  }
#pragma unroll
  for (int u = 0; u < TAIL_U; ++u) red[kl][u][ol] = acc[u];
  __syncthreads();
  for (int idx = tid; idx < TAIL_U * 64; idx += 256) {
    const int u = idx >> 6, x = idx & 63;
    const int oo = blockIdx.y * 64 + x;
    if (b0 + u < B && oo < embed)
      partial[((size_t)blockIdx.z * B + b0 + u) * embed + oo] = red[0][u][x] + red[1][u][x] + red[2][u][x] + red[3][u][x];
  }
}

// emb[b][o] = bias[o] + sum_ks partial[ks][b][o]; optional F.normalize(p=2, dim=1, eps=1e-12)
__global__ __launch_bounds__(256) void tail_finalize_kernel(const float* __restrict__ partial, int ks, int B,
                                                            int embed, const float* __restrict__ bias,
                                                            int normalize, float* __restrict__ emb) {
  __shared__ float wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float ss = 0.f;
  for (int o = tid; o < embed; o += 256) {
    float t = bias[o];
#pragma unroll 8
    for (int k = 0; k < ks; ++k) t += partial[((size_t)k * B + b) * embed + o];   // fixed order: deterministic
    emb[(size_t)b * embed + o] = t;
    ss = fmaf(t, t, ss);
  }
  if (!normalize) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  const float inv = 1.0f / fmaxf(sqrtf(wsum[0] + wsum[1] + wsum[2] + wsum[3]), 1e-12f);
  for (int o = tid; o < embed; o += 256) emb[(size_t)b * embed + o] *= inv;
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

int64_t se_part_floats(int c_p, int B, int64_t R) { return (R / SE_SEG + B + 1) * (int64_t)c_p; }

int launch_se_scale(const SELayer& se, const void* d_y, int ldy, const int32_t* d_row_off, int B, int t_max,
                    float* d_mean, float* d_part, float* d_scale, int dtype, hipStream_t s) {
  SUGAR_CHECK_ARG(t_max > 0 && (t_max <= SE_SEG || d_part), "se: segment table missing (t_max %d)", t_max);
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(se.c_p % E == 0 && se.c_p > 0, "se: channels %d unsupported", se.c_p);
  SUGAR_CHECK_ARG(se.hid > 0 && se.hid <= 1024 && ldy % E == 0, "se: bad hidden %d / ld %d", se.hid, ldy);
  const dim3 grid((unsigned)ceil_div(se.c_p / E, 8), (unsigned)B, (unsigned)ceil_div(t_max, SE_SEG));
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(se_mean_kernel<float>, grid, dim3(256), 0, s, (const float*)d_y, ldy, se.c_p, d_row_off, d_mean,
                       d_part);
  else
    hipLaunchKernelGGL(se_mean_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)d_y, ldy, se.c_p, d_row_off, d_mean,
                       d_part);
  SUGAR_LAUNCH_CHECK();
  const size_t smem = (se.c_p + se.hid + 1024) * sizeof(float);
  hipLaunchKernelGGL(se_mlp_kernel, dim3(B), dim3(1024), smem, s, d_mean, d_part, d_row_off, se, d_scale);
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

int tail_splits(const TailLayer& t) { return (int)ceil_div(t.in_p, TAIL_K); }

int launch_tail(const TailLayer& t, const float* d_pooled, int B, float* d_partial, float* d_emb, int normalize,
                hipStream_t s) {
  const int ks = tail_splits(t);
  const dim3 grid((unsigned)ceil_div(B, TAIL_U), (unsigned)ceil_div(t.embed, 64), (unsigned)ks);
  hipLaunchKernelGGL(tail_partial_kernel, grid, dim3(256), 0, s, t.wt, t.in_p, t.embed, d_pooled, B, d_partial);
  SUGAR_LAUNCH_CHECK();
  hipLaunchKernelGGL(tail_finalize_kernel, dim3(B), dim3(256), 0, s, d_partial, ks, B, t.embed, t.bias, normalize,
                     d_emb);
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
