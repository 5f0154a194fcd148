// Batched cosine trial scoring (gfx950).
//
// Replaces the python loop of score_trials (sugar/scores.py:188-206) over calculate_score(mode="cosine")
// (:146-150): for each trial, F.cosine_similarity broadcast over the [n, D, 1] x [1, D, m] segment vectors of the
// enrol and test utterances (each vector divided by max(||v||, 1e-8), then the dot product), then .mean().
// One wavefront per trial; the 192-d dot products are lane-strided and reduced with cross-lane shuffles.
#include "common.h"

namespace sugar {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void cosine_trials_kernel(const float* __restrict__ vec, int dim,
                                                            const int32_t* __restrict__ seg_off,
                                                            const int32_t* __restrict__ enroll,
                                                            const int32_t* __restrict__ test, int64_t n_trials,
                                                            float* __restrict__ scores) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_trials) return;                       // wave-uniform
  const int lane = threadIdx.x & 63;
  const int eu = enroll[t], tu = test[t];
  const int e0 = seg_off[eu], e1 = seg_off[eu + 1], t0 = seg_off[tu], t1 = seg_off[tu + 1];
  float total = 0.f;
  for (int i = e0; i < e1; ++i) {
    const float* __restrict__ x1 = vec + (size_t)i * dim;
    float n1 = 0.f;
    for (int d = lane; d < dim; d += 64) n1 = fmaf(x1[d], x1[d], n1);
    n1 = fmaxf(sqrtf(wave_sum(n1)), 1e-8f);
    for (int j = t0; j < t1; ++j) {
      const float* __restrict__ x2 = vec + (size_t)j * dim;
      float n2 = 0.f;
      for (int d = lane; d < dim; d += 64) n2 = fmaf(x2[d], x2[d], n2);
      n2 = fmaxf(sqrtf(wave_sum(n2)), 1e-8f);
      float dot = 0.f;
      for (int d = lane; d < dim; d += 64) dot = fmaf(x1[d] / n1, x2[d] / n2, dot);
      total += wave_sum(dot);
    }
  }
  if (lane == 0) scores[t] = total / (float)((e1 - e0) * (t1 - t0));
}

// calculate_score(mode="norm2") (sugar/scores.py:141-145), see include/sugar_amd.h: lanes run over the vector
// dimension, each sums the squared (e - t + eps) over the test segments, takes the root, and the roots are averaged
// over (enrol segment, dimension).
__global__ __launch_bounds__(256) void norm2_trials_kernel(const float* __restrict__ vec, int dim,
                                                           const int32_t* __restrict__ seg_off,
                                                           const int32_t* __restrict__ enroll,
                                                           const int32_t* __restrict__ test, int64_t n_trials,
                                                           float* __restrict__ scores) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_trials) return;                       // wave-uniform
  const int lane = threadIdx.x & 63;
  const int eu = enroll[t], tu = test[t];
  const int e0 = seg_off[eu], e1 = seg_off[eu + 1], t0 = seg_off[tu], t1 = seg_off[tu + 1];
  float total = 0.f;
  for (int i = e0; i < e1; ++i) {
    const float* __restrict__ x1 = vec + (size_t)i * dim;
    for (int d = lane; d < dim; d += 64) {
      const float e = x1[d];
      float ss = 0.f;
      for (int j = t0; j < t1; ++j) {
        const float diff = (e - vec[(size_t)j * dim + d]) + 1e-6f;
        ss = fmaf(diff, diff, ss);
      }
      total += sqrtf(ss);
    }
  }
  total = wave_sum(total);
  if (lane == 0) scores[t] = -total / ((float)(e1 - e0) * (float)dim);
}

}  // namespace
}  // namespace sugar

extern "C" int sugar_cosine_trials(const float* d_vec, int dim, const int32_t* d_seg_off, int n_utts,
                                   const int32_t* d_enroll, const int32_t* d_test, int64_t n_trials,
                                   float* d_scores, void* stream) {
  using namespace sugar;
  SUGAR_CHECK_ARG(d_vec && d_seg_off && d_enroll && d_test && d_scores, "cosine_trials: null pointer");
  SUGAR_CHECK_ARG(dim > 0 && n_utts > 0 && n_trials >= 0, "cosine_trials: bad sizes");
  if (n_trials == 0) return SUGAR_OK;
  SUGAR_CHECK_ARG(ceil_div(n_trials, 4) < (1ll << 31), "cosine_trials: too many trials for one launch");
  hipLaunchKernelGGL(cosine_trials_kernel, dim3((unsigned)ceil_div(n_trials, 4)), dim3(256), 0, (hipStream_t)stream,
                     d_vec, dim, d_seg_off, d_enroll, d_test, n_trials, d_scores);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

extern "C" int sugar_norm2_trials(const float* d_vec, int dim, const int32_t* d_seg_off, int n_utts,
                                  const int32_t* d_enroll, const int32_t* d_test, int64_t n_trials,
                                  float* d_scores, void* stream) {
  using namespace sugar;
  SUGAR_CHECK_ARG(d_vec && d_seg_off && d_enroll && d_test && d_scores, "norm2_trials: null pointer");
  SUGAR_CHECK_ARG(dim > 0 && n_utts > 0 && n_trials >= 0, "norm2_trials: bad sizes");
  if (n_trials == 0) return SUGAR_OK;
  SUGAR_CHECK_ARG(ceil_div(n_trials, 4) < (1ll << 31), "norm2_trials: too many trials for one launch");
  hipLaunchKernelGGL(norm2_trials_kernel, dim3((unsigned)ceil_div(n_trials, 4)), dim3(256), 0, (hipStream_t)stream,
                     d_vec, dim, d_seg_off, d_enroll, d_test, n_trials, d_scores);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}
