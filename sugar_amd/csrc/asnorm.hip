// Adaptive score normalisation back-end (gfx950): SURVEY section 8f rank 2.
//
//   sugar_cohort_scores  score_cohorts (sugar/scores.py:77-111): for every utterance, the cosine similarity of each of
//                        its segment vectors with every cohort vector (F.cosine_similarity, eps 1e-8: each vector is
//                        divided by max(||v||, eps)), averaged over the segments -> [n_utts][n_cohort].
//                        mean_i <v_i/|v_i|, c/|c|> = <mean_i v_i/|v_i|, c/|c|>: one pass builds the unit-mean matrix
//                        U [n_utts][dim] and the unit cohort matrix C^ [n_cohort][dim], then an f32 tiled GEMM U C^T
//                        (11 GFLOP for 4 874 x 6 000 x 192: not worth reduced precision).
//   sugar_topk_stats     the per-row statistics of asnorm (:370-379) / snorm (:305-308): the nTop largest (nTop < 0:
//                        smallest) scores of a row are selected EXACTLY with a 4-pass radix select on the
//                        order-preserving integer image of the floats (ties at the threshold are counted, so the result
//                        equals torch.topk's multiset), then mean and sqrt(sum((v - mean)^2) / divisor) in double.
//   sugar_snorm_trials   0.5 * ((s - mean_e) / max(std_e, eps) + (s - mean_t) / max(std_t, eps))   (:388-399)
#include "common.h"

namespace sugar {
namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per output row: rows [0, n_utts) = mean of an utterance's unit segment vectors, rows [n_utts, +n_cohort) =
// unit cohort vectors
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* __restrict__ vec, int dim,
                                                        const int32_t* __restrict__ seg_off, int n_utts,
                                                        const float* __restrict__ cohort, int n_cohort,
                                                        float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n_utts + n_cohort) return;
  float* __restrict__ o = out + (size_t)row * dim;
  if (row >= n_utts) {
    const float* __restrict__ x = cohort + (size_t)(row - n_utts) * dim;
    float n = 0.f;
    for (int d = lane; d < dim; d += 64) n = fmaf(x[d], x[d], n);
    n = fmaxf(sqrtf(wave_sum_f(n)), 1e-8f);
    for (int d = lane; d < dim; d += 64) o[d] = x[d] / n;
    return;
  }
  const int s0 = seg_off[row], s1 = seg_off[row + 1];
  for (int d = lane; d < dim; d += 64) o[d] = 0.f;
  for (int i = s0; i < s1; ++i) {
    const float* __restrict__ x = vec + (size_t)i * dim;
    float n = 0.f;
    for (int d = lane; d < dim; d += 64) n = fmaf(x[d], x[d], n);
    n = fmaxf(sqrtf(wave_sum_f(n)), 1e-8f);
    for (int d = lane; d < dim; d += 64) o[d] += x[d] / n;
  }
  const float inv = 1.0f / (float)(s1 - s0);
  for (int d = lane; d < dim; d += 64) o[d] *= inv;
}

// out[u][m] = sum_d U[u][d] C[m][d]; 64 x 64 tile per workgroup, 4 x 4 outputs per thread, K in slabs of 16
__global__ __launch_bounds__(256) void cohort_gemm_kernel(const float* __restrict__ U, int n_utts,
                                                          const float* __restrict__ Cn, int n_cohort, int dim,
                                                          float* __restrict__ out) {
  __shared__ float As[16][68], Bs[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int u0 = blockIdx.y * 64, m0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int lr = tid >> 2, lk = (tid & 3) * 4;        // staging: row 0..63, k offset 0,4,8,12
  for (int k0 = 0; k0 < dim; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + lk + e;
      As[lk + e][lr] = (u0 + lr < n_utts && k < dim) ? U[(size_t)(u0 + lr) * dim + k] : 0.f;
      Bs[lk + e][lr] = (m0 + lr < n_cohort && k < dim) ? Cn[(size_t)(m0 + lr) * dim + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = u0 + ty * 4 + i;
    if (u >= n_utts) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + tx * 4 + j;
      if (m < n_cohort) out[(size_t)u * n_cohort + m] = acc[i][j];
    }
  }
}

// order-preserving image of a float: larger float <-> larger unsigned key (desc = 0), or the reverse (desc = 1)
__device__ __forceinline__ unsigned fkey(float f, int smallest) {
  const unsigned u = __float_as_uint(f);
  const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return smallest ? ~k : k;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// one workgroup per row: exact top-k selection by 4 radix passes, then mean and sqrt(sum (v-mean)^2 / divisor)
__global__ __launch_bounds__(256) void topk_stats_kernel(const float* __restrict__ scores, int n_cols, int k,
                                                         int smallest, float divisor, float* __restrict__ mean,
                                                         float* __restrict__ sd) {
  __shared__ int hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_remaining;
  __shared__ double red[256];
  const float* __restrict__ row = scores + (size_t)blockIdx.x * n_cols;
  const int tid = threadIdx.x;
  if (tid == 0) { s_prefix = 0u; s_remaining = k; }
  for (int pass = 3; pass >= 0; --pass) {
    hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 3 ? 0u : (0xffffffffu << (8 * (pass + 1)));
    for (int i = tid; i < n_cols; i += 256) {
      const unsigned key = fkey(row[i], smallest);
      if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = s_remaining, b = 255;
      for (; b > 0; --b) {                         // walk down from the largest byte value
        if (hist[b] >= rem) break;
        rem -= hist[b];
      }
      s_remaining = rem;                           // still to take among the elements whose byte == b
      s_prefix = prefix | ((unsigned)b << (8 * pass));
    }
    __syncthreads();
  }
  const unsigned thr = s_prefix;                   // key of the k-th selected element
  const int ties = s_remaining;                    // how many elements equal to it are selected
  double s = 0.0;
  float tval = 0.f;
  for (int i = tid; i < n_cols; i += 256) {
    const float v = row[i];
    const unsigned key = fkey(v, smallest);
    if (key > thr) s += (double)v;
    else if (key == thr) tval = v;
  }
  // every thread that saw the threshold key holds the same float; broadcast one
  __shared__ float s_tval;
  if (tid == 0) s_tval = 0.f;
  __syncthreads();
  if (tval != 0.f) s_tval = tval;                  // identical values: benign race (0.0f ties contribute nothing)
  __syncthreads();
  const double tv = (double)s_tval;
  const double m = (block_sum(s, red) + tv * ties) / (double)k;
  double q = 0.0;
  for (int i = tid; i < n_cols; i += 256) {
    const float v = row[i];
    if (fkey(v, smallest) > thr) q += ((double)v - m) * ((double)v - m);
  }
  const double qq = block_sum(q, red) + (tv - m) * (tv - m) * ties;
  if (tid == 0) {
    mean[blockIdx.x] = (float)m;
    sd[blockIdx.x] = (float)sqrt(qq / (double)divisor);
  }
}

__global__ __launch_bounds__(256) void snorm_trials_kernel(const float* __restrict__ scores,
                                                           const int32_t* __restrict__ enroll,
                                                           const int32_t* __restrict__ test, int64_t n,
                                                           const float* __restrict__ mean, const float* __restrict__ sd,
                                                           float std_eps, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const float s = scores[t];
  const int e = enroll[t], u = test[t];
  const float z = (s - mean[e]) / fmaxf(sd[e], std_eps);
  const float tn = (s - mean[u]) / fmaxf(sd[u], std_eps);
  out[t] = 0.5f * (z + tn);
}

}  // namespace
}  // namespace sugar

using namespace sugar;

extern "C" int sugar_cohort_scores(const float* d_vec, int dim, const int32_t* d_seg_off, int n_utts,
                                   const float* d_cohort, int n_cohort, float* d_scratch, float* d_out, void* stream) {
  SUGAR_CHECK_ARG(d_vec && d_seg_off && d_cohort && d_scratch && d_out, "cohort_scores: null pointer");
  SUGAR_CHECK_ARG(dim > 0 && n_utts > 0 && n_cohort > 0, "cohort_scores: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)ceil_div((int64_t)n_utts + n_cohort, 4)), dim3(256), 0, s, d_vec,
                     dim, d_seg_off, n_utts, d_cohort, n_cohort, d_scratch);
  SUGAR_LAUNCH_CHECK();
  hipLaunchKernelGGL(cohort_gemm_kernel, dim3((unsigned)ceil_div(n_cohort, 64), (unsigned)ceil_div(n_utts, 64)),
                     dim3(256), 0, s, d_scratch, n_utts, d_scratch + (size_t)n_utts * dim, n_cohort, dim, d_out);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

extern "C" int sugar_topk_stats(const float* d_scores, int64_t n_rows, int n_cols, int ntop, float divisor,
                                float* d_mean, float* d_std, void* stream) {
  SUGAR_CHECK_ARG(d_scores && d_mean && d_std, "topk_stats: null pointer");
  SUGAR_CHECK_ARG(n_rows > 0 && n_rows < (1ll << 31) && n_cols > 0, "topk_stats: bad sizes");
  const int k = ntop < 0 ? -ntop : ntop;
  SUGAR_CHECK_ARG(k >= 1 && k <= n_cols, "topk_stats: nTop %d outside 1..%d (torch.topk: selected index k out of range)",
                  ntop, n_cols);
  SUGAR_CHECK_ARG(divisor > 0.f, "topk_stats: divisor must be positive");
  hipLaunchKernelGGL(topk_stats_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, d_scores, n_cols, k,
                     ntop < 0 ? 1 : 0, divisor, d_mean, d_std);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

extern "C" int sugar_snorm_trials(const float* d_scores, const int32_t* d_enroll, const int32_t* d_test,
                                  int64_t n_trials, const float* d_mean, const float* d_std, float std_eps,
                                  float* d_out, void* stream) {
  SUGAR_CHECK_ARG(d_scores && d_enroll && d_test && d_mean && d_std && d_out, "snorm_trials: null pointer");
  SUGAR_CHECK_ARG(n_trials >= 0, "snorm_trials: bad size");
  if (n_trials == 0) return SUGAR_OK;
  hipLaunchKernelGGL(snorm_trials_kernel, dim3((unsigned)ceil_div(n_trials, 256)), dim3(256), 0, (hipStream_t)stream,
                     d_scores, d_enroll, d_test, n_trials, d_mean, d_std, std_eps, d_out);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}
