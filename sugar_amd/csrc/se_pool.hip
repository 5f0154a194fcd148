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
  __device__ __forceinline__ static void store_nt(float* p, const float* v) { store(p, v); }
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
  __device__ __forceinline__ static void store_nt(bf16* p, const float* v) {   // non-temporal: does not displace cached lines
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, t), (u32x4*)p);
  }
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
// the rest of the batch).  The workgroup does not stop at the channel sums: W1 (mean) is linear in the mean, so it
// goes on to its slab's share of the reduce layer,  hpart[segment][slab][j] = sum_{c in slab} W1[j][c] sum_t y[t][c] / T,
// 32 KiB of weights per workgroup, spread over 2 048 workgroups.  se_expand_kernel adds the shares up in a fixed
// order.  (Round 1 ran the whole MLP as one workgroup per utterance: 512 KiB of weights per workgroup at the ~50 GB/s
// a single CU pulls from L2 with dependent loads = 14.6 us per launch, as long as the mean pass itself.)
constexpr int SE_SEG = 2048;
#ifndef SE_WPRE
#define SE_WPRE 1
#endif

__device__ __forceinline__ int se_slot(const int32_t* row_off, int b) { return row_off[b] / SE_SEG + b; }

// BNIN (BN recalibration): y is the conv's output BEFORE its train-mode BatchNorm, whose (scale, shift) of this batch
// are bsc / bsh; the mean is linear, so mean(BN(y)) = scale mean(y) + shift and the BatchNorm needs no pass over y of
// its own (se_apply_kernel<.., BNIN> applies it to the rows it reads anyway).
template <typename T, bool BNIN = false>
__global__ __launch_bounds__(256) void se_mean_kernel(const T* __restrict__ y, int ldy, const int32_t* __restrict__ row_off,
                                                      SELayer se, int nslab, float* __restrict__ hpart,
                                                      const float* __restrict__ bsc = nullptr,
                                                      const float* __restrict__ bsh = nullptr) {
  constexpr int E = Chunk<T>::N, SC = 8 * E;          // channels per slab
  __shared__ float partial[32][SC + 1];
  __shared__ float msum[SC];
  __shared__ float hp2[2][256];
  const int b = blockIdx.y, tid = threadIdx.x, c_p = se.c_p;
  const int s_utt = row_off[b], n_utt = row_off[b + 1] - s_utt;
  const int seg0 = blockIdx.z * SE_SEG;
  if (seg0 >= n_utt) return;
  const int s = s_utt + seg0, n = min(SE_SEG, n_utt - seg0);
  const int tx = tid & 7, ty = tid >> 3;
  const int c0 = (blockIdx.x * 8 + tx) * E;
  // The reduce-layer weights of this slab (32 per thread) do not depend on y: with hid <= 128 and a full slab they are
  // requested first, so that their L2 round trip runs under the pass over y instead of behind it (every workgroup of
  // the launch reaches that point at the same time: 64 MB of L2 reads in one burst)
  const int hid = se.hid, cbase = blockIdx.x * SC;
  const int wj = min(tid & 127, hid - 1), whalf = tid >> 7;
  const bool wpre = SE_WPRE && hid <= 128 && cbase + SC <= c_p;
  float wq[SC / 2];
  if (wpre) {
    const float* __restrict__ wrow = se.w1t + (size_t)(cbase + whalf * (SC / 2)) * hid + wj;
#pragma unroll
    for (int k = 0; k < SC / 2; ++k) wq[k] = wrow[(size_t)k * hid];
  }
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
  if (tid < SC) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += partial[k][tid];
    float share = t / (float)n_utt;                    // this segment's share of the utterance mean
    if (BNIN) {
      const int c = min(cbase + tid, c_p - 1);
      share = cbase + tid < c_p ? fmaf(bsc[c], share, bsh[c] * ((float)n / (float)n_utt)) : 0.f;
    }
    msum[tid] = share;
  }
  __syncthreads();
  // this slab's share of reduce.weight x mean: thread (j, half) walks SC / 2 channels; weight rows of padded channels are 0
  float* out = hpart + ((size_t)(se_slot(row_off, b) + blockIdx.z) * nslab + blockIdx.x) * hid;
  if (wpre) {                                          // same sums in the same order as the loop below
    const int j = tid & 127, cc0 = whalf * (SC / 2);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SC / 2; ++k) t = fmaf(wq[k], msum[cc0 + k], t);
    hp2[whalf][j] = j < hid ? t : 0.f;
    __syncthreads();
    if (tid < 128 && j < hid) out[j] = hp2[0][tid] + hp2[1][tid];
    return;
  }
  for (int j0 = 0; j0 < hid; j0 += 128) {
    const int j = j0 + (tid & 127), half = tid >> 7;
    float t = 0.f;
    if (j < hid) {
      // no per-element condition inside the loops: hipcc would branch around every load and wait for each one; the
      // full slab (every slab but a ragged last one) is unrolled completely so that all its loads are in flight at once
      const int cc0 = half * (SC / 2), lim = min(SC / 2, c_p - cbase - cc0);
      const float* __restrict__ wrow = se.w1t + (size_t)(cbase + cc0) * hid + j;
      if (lim == SC / 2) {
        float w[SC / 2];
#pragma unroll
        for (int k = 0; k < SC / 2; ++k) w[k] = wrow[(size_t)k * hid];
#pragma unroll
        for (int k = 0; k < SC / 2; ++k) t = fmaf(w[k], msum[cc0 + k], t);
      } else {
        for (int k = 0; k < lim; ++k) t = fmaf(wrow[(size_t)k * hid], msum[cc0 + k], t);
      }
    }
    hp2[half][tid & 127] = t;
    __syncthreads();
    if (tid < 128 && j < hid) out[j] = hp2[0][tid] + hp2[1][tid];
    __syncthreads();
  }
}

// se_expand_kernel: grid (groups of SE_U utterances, blocks of 64 channels), 1024 threads = 64 channels x 16 k-slices.
// hidden = relu(b1 + the shares of every segment and slab, in order); scale = sigmoid(W2 hidden + b2) with W2 stored
// transposed (coalesced); every weight load is used for SE_U utterances.  The arithmetic is nothing; what the kernel
// waits for is L2 latency, so every thread issues all its loads before the first use (8 shares, <= 16 weights).
constexpr int SE_U = 8, SE_CB = 64, SE_KP = 1024 / SE_CB;

__global__ __launch_bounds__(1024) void se_expand_kernel(const float* __restrict__ hpart, const int32_t* __restrict__ row_off,
                                                         int B, SELayer se, int nslab, float* __restrict__ scale) {
  extern __shared__ float sm[];          // hidden [SE_U][hid] | partial [SE_U][1024]
  float* hidden = sm;
  float* part = sm + SE_U * se.hid;
  const int b0 = blockIdx.x * SE_U, tid = threadIdx.x, hid = se.hid;
  const int nu = min(SE_U, B - b0);
  // this thread's first 16 expand weights do not depend on the hidden vector: requested here, their L2 round trip runs
  // under the one of the shares below instead of behind the barrier
  const int c = blockIdx.y * SE_CB + (tid & (SE_CB - 1)), kp = tid / SE_CB;
  const int cl = min(c, se.c_p - 1);
  const int per = (hid + SE_KP - 1) / SE_KP;
  const int j0 = kp * per, j1 = min(hid, j0 + per);
  float w0[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w0[k] = se.w2t[(size_t)min(j0 + k, hid - 1) * se.c_p + cl];
  for (int idx = tid; idx < SE_U * hid; idx += 1024) {
    const int u = idx / hid, j = idx - u * hid;
    float t = 0.f;
    if (u < nu) {
      const int b = b0 + u;
      const int n_utt = row_off[b + 1] - row_off[b];
      const int nparts = ((n_utt + SE_SEG - 1) / SE_SEG) * nslab;
      const float* p = hpart + (size_t)se_slot(row_off, b) * nslab * hid + j;
      t = se.b1[j];
      for (int k0 = 0; k0 < nparts; k0 += 8) {                          // fixed order: segment-major, then slab
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)min(k0 + k, nparts - 1) * hid];   // clamped: always a valid share
#pragma unroll
        for (int k = 0; k < 8; ++k) t += (k0 + k < nparts) ? v[k] : 0.f;
      }
      t = fmaxf(t, 0.f);
    }
    hidden[idx] = t;
  }
  __syncthreads();
  float t[SE_U];
#pragma unroll
  for (int u = 0; u < SE_U; ++u) t[u] = 0.f;
  for (int jb = j0; jb < j1; jb += 16) {
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = jb == j0 ? w0[k] : se.w2t[(size_t)min(jb + k, hid - 1) * se.c_p + cl];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float wk = (jb + k < j1) ? w[k] : 0.f;
      const int jj = min(jb + k, hid - 1);
#pragma unroll
      for (int u = 0; u < SE_U; ++u) t[u] = fmaf(wk, hidden[u * hid + jj], t[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < SE_U; ++u) part[u * 1024 + tid] = t[u];
  __syncthreads();
  if (tid < SE_CB && c < se.c_p) {
    const float bias = se.b2[c];
    for (int u = 0; u < nu; ++u) {
      const float* pu = part + u * 1024 + tid;
      float z = bias;
#pragma unroll
      for (int k = 0; k < SE_KP; ++k) z += pu[k * SE_CB];
      scale[(size_t)(b0 + u) * se.c_p + c] = 1.0f / (1.0f + expf(-z));
    }
  }
}

// ------------------------------------------------------------------------------------------ SE apply
constexpr int APPLY_ROWS = 4;     // 256 chunks = one per thread (bf16): 32 rows (8 dependent round trips per thread) ran at 5.9 TB/s, 4 rows at 6.8
// The block output `out` (a column slice of the catconv input) is not read again before catconv, the running sum S is
// read by the next block's first_conv right away: `out` goes out with non-temporal stores, so that it does not push S
// (and the y / mid tensors of the next block) out of the Infinity Cache.  Same job, 256 x 3 s: first_conv 0.165 ->
// 0.156, last_conv 0.152 -> 0.146, se_apply 0.157 -> 0.162 ms per step.  (Non-temporal loads of y: slower all round.)
// The last block's output is what catconv reads next: ordinary stores.
#ifndef SE_NT
#define SE_NT 1
#endif

template <typename T, bool NT_OUT = false, bool BNIN = false>
__global__ __launch_bounds__(256) void se_apply_kernel(const T* __restrict__ y, int ldy,
                                                       const float* __restrict__ scale, int c_p, T* __restrict__ S,
                                                       int ldS, T* __restrict__ out, int ldo, int write_S,
                                                       const int32_t* __restrict__ row_off,
                                                       const float* __restrict__ bsc = nullptr,
                                                       const float* __restrict__ bsh = nullptr) {
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
    if (BNIN) {                                        // y arrives before its train-mode BatchNorm (see se_mean_kernel)
#pragma unroll
      for (int i = 0; i < E; ++i) vy[i] = fmaf(vy[i], bsc[ch * E + i], bsh[ch * E + i]);
    }
#pragma unroll
    for (int i = 0; i < E; ++i) vo[i] = fmaf(vy[i], sc[ch * E + i], vs[i]);
    // the reference adds in f32; the running sum re-reads the *stored* (rounded) block output, as the next
    // block does, so out and S stay consistent in the compute dtype
    if (NT_OUT) Chunk<T>::store_nt(out + row * ldo + ch * E, vo);   // (a run-time choice compiles to ONE plain store)
    else Chunk<T>::store(out + row * ldo + ch * E, vo);
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
  // pooled tile as [feature][utterance]: the 16 utterances of one feature are four 16-byte broadcast reads (as
  // [utterance][feature] the loop issued 16 scalar LDS reads per weight and was LDS-issue-bound: 20 us)
  __shared__ __attribute__((aligned(16))) float ps[TAIL_K][TAIL_U];
  __shared__ float red[4][TAIL_U][64];
  const int tid = threadIdx.x, ol = tid & 63, kl = tid >> 6;
  const int o = blockIdx.y * 64 + ol;
  const int b0 = blockIdx.x * TAIL_U;
  const int k0 = blockIdx.z * TAIL_K;
  const int kn = min(TAIL_K, in_p - k0);
  for (int idx = tid; idx < TAIL_U * TAIL_K; idx += 256) {
    const int u = idx / TAIL_K, i = idx - u * TAIL_K;
    ps[i][u] = (b0 + u < B && i < kn) ? pooled[(size_t)(b0 + u) * in_p + k0 + i] : 0.f;
  }
  __syncthreads();
  float acc[TAIL_U];
#pragma unroll
  for (int u = 0; u < TAIL_U; ++u) acc[u] = 0.f;
  const bool o_ok = o < embed;
  const float* wcol = wt + (size_t)k0 * embed + (o_ok ? o : 0);
  // all of a thread's weights (<= TAIL_K / 4 = 32) are requested before the first is used: with 8 at a time a wave paid
  // four L2 round trips in a row and lived 9 us; rows past kn are clamped to a valid one and masked
  constexpr int NW = TAIL_K / 4;
  float w[NW];
#pragma unroll
  for (int n = 0; n < NW; ++n) w[n] = wcol[(size_t)min(kl + 4 * n, kn - 1) * embed];
#pragma unroll
  for (int n = 0; n < NW; ++n) {
    const int i = kl + 4 * n;
    const float wn = o_ok && i < kn ? w[n] : 0.f;
#pragma unroll
    for (int u4 = 0; u4 < TAIL_U / 4; ++u4) {
      const f32x4 pv = *(const f32x4*)&ps[min(i, TAIL_K - 1)][4 * u4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * u4 + e] = fmaf(wn, pv[e], acc[4 * u4 + e]);
    }
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

static int se_slabs(const SELayer& se, int dtype) { return (int)ceil_div(se.c_p, dtype == SUGAR_F32 ? 32 : 64); }

int64_t se_part_floats(const SELayer& se, int dtype, int B, int64_t R) {
  return (R / SE_SEG + B + 1) * (int64_t)se_slabs(se, dtype) * se.hid;
}

// scale[b][c] = sigmoid(W2 relu(W1 mean_t(y[b]) + b1) + b2)
// Measured and dropped (round 2): the whole SE step of a block (mean, MLP, apply) as ONE kernel with the utterance's y
// tile [301][512] held in the registers of a 1024-thread workgroup (y read once, two launches fewer): 86 us per block
// against 81 us for the three kernels -- with one workgroup per CU and equal-length utterances every CU idles through the
// MLP's L2 round trips at the same time.
int launch_se_scale(const SELayer& se, const void* d_y, int ldy, const int32_t* d_row_off, int B, int t_max,
                           float* d_part, float* d_scale, int dtype, hipStream_t s, const float* d_bn_scale,
                           const float* d_bn_shift) {
  SUGAR_CHECK_ARG(t_max > 0 && d_part, "se: partial table missing");
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(se.c_p % E == 0 && se.c_p > 0, "se: channels %d unsupported", se.c_p);
  SUGAR_CHECK_ARG(se.hid > 0 && se.hid <= 1024 && ldy % E == 0, "se: bad hidden %d / ld %d", se.hid, ldy);
  const int nslab = se_slabs(se, dtype);
  const dim3 grid((unsigned)nslab, (unsigned)B, (unsigned)ceil_div(t_max, SE_SEG));
  if (d_bn_scale) {                                    // recalibration: the BatchNorm of y folded in
    if (dtype == SUGAR_F32)
      hipLaunchKernelGGL((se_mean_kernel<float, true>), grid, dim3(256), 0, s, (const float*)d_y, ldy, d_row_off, se, nslab,
                         d_part, d_bn_scale, d_bn_shift);
    else
      hipLaunchKernelGGL((se_mean_kernel<bf16, true>), grid, dim3(256), 0, s, (const bf16*)d_y, ldy, d_row_off, se, nslab,
                         d_part, d_bn_scale, d_bn_shift);
  } else if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(se_mean_kernel<float>, grid, dim3(256), 0, s, (const float*)d_y, ldy, d_row_off, se, nslab, d_part);
  else
    hipLaunchKernelGGL(se_mean_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)d_y, ldy, d_row_off, se, nslab, d_part);
  SUGAR_LAUNCH_CHECK();
  const size_t smem = (size_t)SE_U * (se.hid + 1024) * sizeof(float);
  hipLaunchKernelGGL(se_expand_kernel, dim3((unsigned)ceil_div(B, SE_U), (unsigned)ceil_div(se.c_p, SE_CB)), dim3(1024), smem, s,
                     d_part, d_row_off, B, se, nslab, d_scale);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

// out = y * scale[b] + S ;  S += out (when write_S)
int launch_se_apply(const void* d_y, int ldy, const float* d_scale, int c_p, void* d_S, int ldS, void* d_out,
                           int ldo, int write_S, const int32_t* d_row_off, int B, int t_max, int dtype, hipStream_t s,
                           const float* d_bn_scale, const float* d_bn_shift) {
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(c_p % E == 0 && ldy % E == 0 && ldS % E == 0 && ldo % E == 0, "se_apply: unaligned channels");
  const dim3 grid((unsigned)ceil_div(t_max, APPLY_ROWS), (unsigned)B);
  if (d_bn_scale) {                                    // recalibration: the BatchNorm of y folded in
    if (dtype == SUGAR_F32)
      hipLaunchKernelGGL((se_apply_kernel<float, false, true>), grid, dim3(256), 0, s, (const float*)d_y, ldy, d_scale, c_p,
                         (float*)d_S, ldS, (float*)d_out, ldo, write_S, d_row_off, d_bn_scale, d_bn_shift);
    else
      hipLaunchKernelGGL((se_apply_kernel<bf16, false, true>), grid, dim3(256), 0, s, (const bf16*)d_y, ldy, d_scale, c_p,
                         (bf16*)d_S, ldS, (bf16*)d_out, ldo, write_S, d_row_off, d_bn_scale, d_bn_shift);
  } else if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(se_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)d_y, ldy, d_scale, c_p,
                       (float*)d_S, ldS, (float*)d_out, ldo, write_S, d_row_off);
  else if (SE_NT && write_S)
    hipLaunchKernelGGL((se_apply_kernel<bf16, true>), grid, dim3(256), 0, s, (const bf16*)d_y, ldy, d_scale, c_p,
                       (bf16*)d_S, ldS, (bf16*)d_out, ldo, write_S, d_row_off);
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
