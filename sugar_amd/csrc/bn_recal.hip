// Train-mode BatchNorm for BN recalibration (gfx950).
//
// `batch_forward` (sugar/models/__init__.py:201-242) runs a sampled subnet in train() mode over a few thousand
// utterances so that every BatchNorm re-estimates its running statistics (DynamicBatchNorm1d,
// sugar/modules/dynamicarch.py:254-291 -> F.batch_norm(training=True)):
//     y = (z - mean_B) / sqrt(var_B + eps) * gamma + beta          var_B biased, over all frames of the batch
//     running_mean = (1 - m) running_mean + m mean_B
//     running_var  = (1 - m) running_var  + m var_B n / (n - 1)
// Here z is a packed channels-last activation [R][C] (row stride ld) already holding ReLU(conv(x)):
//   bn_stats_kernel    per (row block, 8-chunk channel slab): sum and sum of squares in double
//   bn_finalize_kernel per channel: fixed-order sum of the partials (deterministic), batch statistics, running
//                      update, and the (scale, shift) of this batch
//   bn_apply_kernel    z <- z * scale + shift in place, 16 bytes per lane
// bn_rows_kernel does all three for the tiny [B][C] f32 matrices of bnpool / bnembed (one thread per channel).
// bn_fold_kernel turns the running statistics back into the (scale, shift) the eval-mode conv epilogues apply.
#include <stdlib.h>

#include "common.h"

namespace sugar {
namespace {

constexpr float BN_EPS = 1e-5f;
constexpr int BN_BLOCKS = 128;      // row blocks of the partial sums, wide tensors
// Narrow tensors (the 64 .. 128-channel Res2 steps: one or two channel slabs) get four times the row blocks: with 128
// the launch was 128 workgroups walking 19 rows per thread in dependent round trips (10 us for 10 MB); the partial table
// of a launch stays within bn_workspace_doubles
constexpr int BN_BLOCKS_NARROW = 512, BN_NARROW_C = 128;

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ z, int ld, int64_t R, int C,
                                                       double* __restrict__ part) {
  constexpr int E = DT<T>::EPC;
  __shared__ double red[2][32][8 * E];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = (blockIdx.x * 8 + tx) * E;
  const int64_t per = (R + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = blockIdx.y * per, r1 = r0 + per < R ? r0 + per : R;
  double s[E], q[E];
#pragma unroll
  for (int i = 0; i < E; ++i) s[i] = q[i] = 0.0;
  if (c0 < C) {
#pragma unroll 4
    for (int64_t r = r0 + ty; r < r1; r += 32) {
      const u32x4 raw = *(const u32x4*)(z + r * ld + c0);
      const T* v = (const T*)&raw;
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const double x = (double)to_f32(v[i]);
        s[i] += x;
        q[i] += x * x;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) {
    red[0][ty][tx * E + i] = s[i];
    red[1][ty][tx * E + i] = q[i];
  }
  __syncthreads();
  if (threadIdx.x < 8 * E) {
    const int c = blockIdx.x * 8 * E + threadIdx.x;
    if (c < C) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        a += red[0][k][threadIdx.x];
        b += red[1][k][threadIdx.x];
      }
      part[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
      part[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
  }
}

// 256 threads = 32 channels x 8 partial lanes: every lane adds its partials (k = lane, lane + 8, ...: independent
// loads), the 8 lane sums of a channel are added in lane order, so the result does not depend on scheduling.  (One
// thread per channel walking all 128 partials was a 36 us chain of dependent loads, 28 times per step.)
// LP = partial lanes per channel: 8 (wide tensors, 32 channels per workgroup) or 32 (narrow ones with 512 row blocks: 8
// channels per workgroup, 16 partials per lane).
template <int LP>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ part, int nblk, int C, double n,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float momentum,
                                                          float* __restrict__ scale, float* __restrict__ shift) {
  constexpr int CW = 256 / LP;        // channels per workgroup
  __shared__ double red[2][LP][CW];
  const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
  const int c = blockIdx.x * CW + tx;
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 8
    for (int k = ty; k < nblk; k += LP) {
      a += part[((size_t)k * 2 + 0) * C + c];
      b += part[((size_t)k * 2 + 1) * C + c];
    }
  }
  red[0][ty][tx] = a;
  red[1][ty][tx] = b;
  __syncthreads();
  if (ty != 0 || c >= C) return;
  a = 0.0;
  b = 0.0;
#pragma unroll
  for (int k = 0; k < LP; ++k) {
    a += red[0][k][tx];
    b += red[1][k][tx];
  }
  const double mean = a / n;
  double var = b / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double sc = (double)gamma[c] / sqrt(var + (double)BN_EPS);
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - mean * sc);
  const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
  rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
  rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(T* __restrict__ z, int ld, int64_t R, int C,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift) {
  constexpr int E = DT<T>::EPC;
  const int nch = C / E;
  const int64_t items = R * nch;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int64_t r = it / nch;
    const int ch = (int)(it - r * nch);
    u32x4 raw = *(const u32x4*)(z + r * ld + ch * E);
    T* v = (T*)&raw;
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = from_f32<T>(fmaf(to_f32(v[i]), scale[ch * E + i], shift[ch * E + i]));
    *(u32x4*)(z + r * ld + ch * E) = raw;
  }
}

// x [B][ld] f32: train-mode BN in place + running update (bnpool, bnembed).  256 threads = 32 channels x 8 row lanes;
// lane sums are added in lane order (deterministic).
__global__ __launch_bounds__(256) void bn_rows_kernel(float* __restrict__ x, int ld, int B, int C,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ rmean, float* __restrict__ rvar,
                                                      float momentum) {
  __shared__ double red[2][8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  const bool ok = c < C;
  double a = 0.0, b = 0.0;
  if (ok) {
#pragma unroll 4
    for (int r = ty; r < B; r += 8) {
      const double v = x[(size_t)r * ld + c];
      a += v;
      b += v * v;
    }
  }
  red[0][ty][tx] = a;
  red[1][ty][tx] = b;
  __syncthreads();
  a = 0.0;
  b = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a += red[0][k][tx];
    b += red[1][k][tx];
  }
  if (!ok) return;
  const double n = B, mean = a / n;
  double var = b / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double sc = (double)gamma[c] / sqrt(var + (double)BN_EPS), sh = (double)beta[c] - mean * sc;
#pragma unroll 4
  for (int r = ty; r < B; r += 8) x[(size_t)r * ld + c] = (float)((double)x[(size_t)r * ld + c] * sc + sh);
  if (ty != 0) return;
  const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
  rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
  rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbiased);
}

__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                      int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double sc = (double)gamma[c] / sqrt((double)rvar[c] + (double)BN_EPS);
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - (double)rmean[c] * sc);
}

}  // namespace

int bn_state_create(BNState* st, const SugarBN* bn, int c, int n_packed, const int* map) {
  SUGAR_CHECK_ARG(bn && bn->weight && bn->bias && bn->running_mean && bn->running_var && c > 0 && n_packed >= c,
                  "bn_state: null tensor or bad size");
  st->n = n_packed;
  st->c = c;
  st->h_map = (int*)malloc((size_t)c * sizeof(int));
  float* host = (float*)calloc(4 * (size_t)n_packed, sizeof(float));
  if (!st->h_map || !host) { free(host); set_error("bn_state: host alloc"); return SUGAR_ENOMEM; }
  for (int i = 0; i < n_packed; ++i) host[3 * (size_t)n_packed + i] = 1.f;     // padded channels: var 1, the rest 0
  for (int o = 0; o < c; ++o) {
    const int p = map ? map[o] : o;
    st->h_map[o] = p;
    host[p] = bn->weight[o];
    host[(size_t)n_packed + p] = bn->bias[o];
    host[2 * (size_t)n_packed + p] = bn->running_mean[o];
    host[3 * (size_t)n_packed + p] = bn->running_var[o];
  }
  int rc = dev_alloc(&st->p, 4 * (size_t)n_packed * sizeof(float));
  if (!rc) rc = dev_upload(st->p, host, 4 * (size_t)n_packed * sizeof(float));
  free(host);
  return rc;
}

void bn_state_free(BNState* st) {
  dev_free(st->p);
  free(st->h_map);
  *st = BNState();
}

int64_t bn_workspace_doubles(int c_max, int64_t R) {
  const int64_t narrow = BN_BLOCKS_NARROW * 2 * BN_NARROW_C, wide = BN_BLOCKS * 2 * c_max;
  const int64_t fused = (int64_t)conv_stat_blocks(R) * 2 * c_max;      // partials written by the producing conv
  const int64_t m = wide > narrow ? wide : narrow;
  return m > fused ? m : fused;
}

int launch_bn_batch(void* d_z, int ld, int64_t R, int C, const BNState& st, float momentum, double* d_part,
                    float* d_scale, float* d_shift, int dtype, hipStream_t s, int part_blocks, bool apply) {
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(C % E == 0 && C <= st.n && ld % E == 0 && ((uintptr_t)d_z & 15) == 0 && R > 0,
                  "bn_batch: bad shape C %d ld %d", C, ld);
  const bool narrow = C <= BN_NARROW_C;
  const int cap = narrow ? BN_BLOCKS_NARROW : BN_BLOCKS;
  const int nblk = part_blocks > 0 ? part_blocks : (int)(R < 32 * (int64_t)cap ? ceil_div(R, 32) : cap);
  if (part_blocks <= 0) {
    const dim3 grid((unsigned)ceil_div(C / E, 8), (unsigned)nblk);
    if (dtype == SUGAR_F32)
      hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)d_z, ld, R, C, d_part);
    else
      hipLaunchKernelGGL(bn_stats_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)d_z, ld, R, C, d_part);
    SUGAR_LAUNCH_CHECK();
  }
  if (narrow)
    hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3((unsigned)ceil_div(C, 8)), dim3(256), 0, s, d_part, nblk, C, (double)R,
                       st.gamma(), st.beta(), st.mean(), st.var(), momentum, d_scale, d_shift);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3((unsigned)ceil_div(C, 32)), dim3(256), 0, s, d_part, nblk, C, (double)R,
                       st.gamma(), st.beta(), st.mean(), st.var(), momentum, d_scale, d_shift);
  SUGAR_LAUNCH_CHECK();
  if (!apply) return SUGAR_OK;
  const int64_t items = R * (C / E);
  const unsigned blocks = (unsigned)(ceil_div(items, 256) < 4096 ? ceil_div(items, 256) : 4096);
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)d_z, ld, R, C, d_scale, d_shift);
  else
    hipLaunchKernelGGL(bn_apply_kernel<bf16>, dim3(blocks), dim3(256), 0, s, (bf16*)d_z, ld, R, C, d_scale, d_shift);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

int launch_bn_rows(float* d_x, int ld, int B, const BNState& st, float momentum, hipStream_t s) {
  hipLaunchKernelGGL(bn_rows_kernel, dim3((unsigned)ceil_div(st.n, 32)), dim3(256), 0, s, d_x, ld, B, st.n, st.gamma(),
                     st.beta(), st.mean(), st.var(), momentum);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

int launch_bn_fold(const BNState& st, float* d_scale, float* d_shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)ceil_div(st.n, 256)), dim3(256), 0, s, st.gamma(), st.beta(),
                     st.mean(), st.var(), st.n, d_scale, d_shift);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
