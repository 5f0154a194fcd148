// Fused log-mel front-end (gfx950).
//
// Replaces LogMelFbanks.forward (sugar/transforms.py:60-77): PreEmphasis (:22-27) -> torchaudio MelSpectrogram
// (:50-52: torch.stft n_fft 512 / hop 160 / periodic Hamming(400) centred in the 512 frame / center=True reflect
// padding, |X|^2, HTK triangular mel filters 20..7600 Hz, norm=None) -> +1e-6 -> log -> InstanceNorm1d(80).
//
// Kernel 1 (fbank_dft_kernel): one workgroup = 64 frames of one utterance.
//   * the 10 480 samples the 64 frames touch are loaded once, reflect-padded at the utterance's true ends and
//     pre-emphasised on the fly, into LDS (skewed by 4 floats per 160 so the per-frame fragment reads of the
//     MFMA A operand are bank-conflict free);
//   * the windowed DFT is an exact-f32 MFMA GEMM  [64 frames x 400] . [400 x (256 cos | 256 sin)]  with
//     v_mfma_f32_32x32x2_f32; the Hamming window is folded into the basis, whose layout is pre-permuted so every
//     lane fetches its B fragments for 4 consecutive MFMAs with one 16-byte load (bins 0..255 only: the Nyquist
//     bin has zero mel weight because f_max = 7600 Hz < 8000 Hz);
//   * power = re^2 + im^2 goes to LDS, the 477-tap sparse mel projection and log run on the VALU, and the
//     [64 x 80] log-mel tile leaves through LDS as one contiguous, coalesced 20 KB store.
// Kernel 2 (instnorm_kernel): per utterance and mel bin, mean / biased variance over the true frames (two-pass),
//   (x - mean) / sqrt(var + 1e-5), written channels-last [R][80] in the compute dtype.
#include <math.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace sugar {

namespace {
constexpr int NFFT = 512, WIN = 400, HOP = SUGAR_HOP, NMEL = SUGAR_N_MELS;
constexpr int NBIN = 256;                 // bins 0..255
constexpr int FRAMES = 64;                // frames per workgroup
constexpr int SPAN = (FRAMES - 1) * HOP + WIN;          // 10480 samples
constexpr int SAMPLE_LDS = SPAN + 4 * (SPAN / HOP + 1);  // skewed
constexpr int PLD = NBIN + 1;             // power tile leading dim (odd: conflict-free column walks)
constexpr int KQ = WIN / 8;               // 50 k-groups of 8
constexpr int MEL_MAXW = 24;              // max non-zero taps of one mel filter (17 for this filterbank)
constexpr int SLD = NMEL + 1;

struct FbankTables {
  float* basis = nullptr;    // [KQ][2 (lane half)][2 (cos, sin)][NBIN][4]
  int* mel_lo = nullptr;     // [NMEL]
  int* mel_cnt = nullptr;    // [NMEL]
  float* mel_w = nullptr;    // [NMEL][MEL_MAXW]
};

std::mutex g_mu;
std::vector<FbankTables> g_tables;  // indexed by device

double hz_to_mel(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
double mel_to_hz(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

int get_tables(FbankTables* out) {
  int dev = 0;
  SUGAR_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  if ((int)g_tables.size() <= dev) g_tables.resize(dev + 1);
  FbankTables& t = g_tables[dev];
  if (!t.basis) {
    std::vector<float> basis((size_t)KQ * 2 * 2 * NBIN * 4);
    const double PI = 3.14159265358979323846;
    for (int q = 0; q < KQ; ++q)
      for (int h = 0; h < 2; ++h)
        for (int col = 0; col < NBIN; ++col)
          for (int i = 0; i < 4; ++i) {
            const int n = 8 * q + 4 * h + i;
            const double w = 0.54 - 0.46 * cos(2.0 * PI * n / WIN);   // torch.hamming_window(400), periodic
            const double ph = 2.0 * PI * (double)((col * n) % NFFT) / NFFT;
            basis[((((size_t)q * 2 + h) * 2 + 0) * NBIN + col) * 4 + i] = (float)(w * cos(ph));
            basis[((((size_t)q * 2 + h) * 2 + 1) * NBIN + col) * 4 + i] = (float)(w * sin(ph));
          }
    // torchaudio.functional.melscale_fbanks(257, 20, 7600, 80, 16000, norm=None, mel_scale="htk")
    std::vector<int> lo(NMEL), cnt(NMEL);
    std::vector<float> mw((size_t)NMEL * MEL_MAXW, 0.f);
    const double m_min = hz_to_mel(20.0), m_max = hz_to_mel(7600.0);
    std::vector<double> fpts(NMEL + 2);
    for (int i = 0; i < NMEL + 2; ++i) fpts[i] = mel_to_hz(m_min + (m_max - m_min) * i / (NMEL + 1));
    for (int m = 0; m < NMEL; ++m) {
      int first = -1, n = 0;
      for (int k = 0; k <= NFFT / 2; ++k) {
        const double f = 8000.0 * k / (NFFT / 2);
        const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
        const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
        const double v = fmax(0.0, fmin(down, up));
        if (v > 0.0) {
          if (first < 0) first = k;
          if (k >= NBIN || k - first >= MEL_MAXW) { set_error("fbank: mel filter %d does not fit the tables", m); return SUGAR_EINVAL; }
          mw[(size_t)m * MEL_MAXW + (k - first)] = (float)v;
          n = k - first + 1;
        }
      }
      lo[m] = first < 0 ? 0 : first;
      cnt[m] = n;
    }
    SUGAR_HIP(hipMalloc(&t.basis, basis.size() * sizeof(float)));
    SUGAR_HIP(hipMemcpy(t.basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMalloc(&t.mel_lo, NMEL * sizeof(int)));
    SUGAR_HIP(hipMemcpy(t.mel_lo, lo.data(), NMEL * sizeof(int), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMalloc(&t.mel_cnt, NMEL * sizeof(int)));
    SUGAR_HIP(hipMemcpy(t.mel_cnt, cnt.data(), NMEL * sizeof(int), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMalloc(&t.mel_w, mw.size() * sizeof(float)));
    SUGAR_HIP(hipMemcpy(t.mel_w, mw.data(), mw.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  *out = t;
  return SUGAR_OK;
}

__device__ __forceinline__ int sample_addr(int p) { return p + 4 * (p / HOP); }

__global__ __launch_bounds__(256, 2) void fbank_dft_kernel(const float* __restrict__ wav,
                                                           const int64_t* __restrict__ wav_off,
                                                           const int32_t* __restrict__ wav_len,
                                                           const int32_t* __restrict__ row_off, FbankTables tb,
                                                           float* __restrict__ logmel) {
  // one LDS array, three lives: samples -> power tile -> log-mel staging
  __shared__ __attribute__((aligned(16))) float lds[FRAMES * PLD];
  static_assert(FRAMES * PLD >= SAMPLE_LDS, "power tile must cover the sample buffer");
  static_assert(FRAMES * PLD >= FRAMES * SLD, "power tile must cover the staging tile");

  const int b = blockIdx.y;
  const int row0 = row_off[b];
  const int T = row_off[b + 1] - row0;
  const int t0 = blockIdx.x * FRAMES;
  if (t0 >= T) return;                               // uniform per workgroup
  const int L = wav_len[b];
  const float* __restrict__ x = wav + wav_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  // ---- stage pre-emphasised, reflect-padded samples --------------------------------------------
  const int base = t0 * HOP - WIN / 2;               // original index of buffer position 0 (= t0*160 - 256 + 56)
  for (int p = tid; p < SPAN; p += 256) {
    int idx = base + p;
    if (idx < 0) idx = -idx;                         // reflect (torch.stft center=True, pad_mode="reflect")
    if (idx >= L) idx = 2 * (L - 1) - idx;
    idx = min(max(idx, 0), L - 1);                   // frames past T of the last tile: keep loads in bounds
    const int prev = idx > 0 ? idx - 1 : 1;          // PreEmphasis reflect pad (1, 0): x[-1] := x[1]
    lds[sample_addr(p)] = x[idx] - 0.97f * x[min(prev, L - 1)];
  }
  __syncthreads();

  // ---- windowed DFT on the f32 matrix cores ----------------------------------------------------
  // wave wv owns bins [64 wv, 64 wv + 64): 2 bin tiles x (cos, sin) x 2 frame tiles = 8 accumulators
  f32x16 acc[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][k][e] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const float* __restrict__ bptr = tb.basis + ((size_t)fh * 2 * NBIN + wv * 64 + fr) * 4;
#pragma unroll 2
  for (int q = 0; q < KQ; ++q) {
    const int n0 = 8 * q + 4 * fh;
    f32x4 af[2], bf[2][2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) af[ft] = *(const f32x4*)&lds[sample_addr((ft * 32 + fr) * HOP + n0)];
    const float* bq = bptr + (size_t)q * (2 * 2 * NBIN * 4);
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int part = 0; part < 2; ++part) bf[bt][part] = *(const f32x4*)(bq + ((size_t)part * NBIN + bt * 32) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int bt = 0; bt < 2; ++bt)
#pragma unroll
          for (int part = 0; part < 2; ++part)
            acc[ft][bt][part] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ft][i], bf[bt][part][i], acc[ft][bt][part], 0, 0, 0);
  }
  __syncthreads();   // every wave is done with the sample buffer

  // ---- power spectrum -> LDS [frame][bin] -------------------------------------------------------
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = ft * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const float re = acc[ft][bt][0][e], im = acc[ft][bt][1][e];
        lds[f * PLD + wv * 64 + bt * 32 + fr] = re * re + im * im;
      }
  __syncthreads();

  // ---- sparse mel projection + log: thread = (frame tid & 63, 20 mel bins) ----------------------
  const int f = tid & 63, mg = tid >> 6;
  float out[NMEL / 4];
#pragma unroll 1
  for (int j = 0; j < NMEL / 4; ++j) {
    const int m = mg * (NMEL / 4) + j;
    const int lo = tb.mel_lo[m], cnt = tb.mel_cnt[m];
    const float* __restrict__ wts = tb.mel_w + m * MEL_MAXW;
    float s = 0.f;
    for (int i = 0; i < cnt; ++i) s = fmaf(lds[f * PLD + lo + i], wts[i], s);
    out[j] = logf(s + 1e-6f);
  }
  __syncthreads();   // power tile fully consumed
#pragma unroll
  for (int j = 0; j < NMEL / 4; ++j) lds[f * SLD + mg * (NMEL / 4) + j] = out[j];
  __syncthreads();
  const int nf = min(FRAMES, T - t0);
  float* __restrict__ dst = logmel + (size_t)(row0 + t0) * NMEL;
  for (int i = tid; i < nf * NMEL; i += 256) dst[i] = lds[(i / NMEL) * SLD + (i % NMEL)];
}

constexpr int NORM_LANES = 8;   // row lanes per mel bin

template <typename T>
__global__ __launch_bounds__(NMEL* NORM_LANES) void instnorm_kernel(const float* __restrict__ logmel,
                                                                     const int32_t* __restrict__ row_off,
                                                                     T* __restrict__ feats) {
  __shared__ float red[NORM_LANES][NMEL];
  const int b = blockIdx.x;
  const int s = row_off[b], n = row_off[b + 1] - s;
  const int m = threadIdx.x % NMEL, rl = threadIdx.x / NMEL;
  const float* __restrict__ src = logmel + (size_t)s * NMEL + m;
  float acc = 0.f;
  for (int r = rl; r < n; r += NORM_LANES) acc += src[(size_t)r * NMEL];
  red[rl][m] = acc;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_LANES; ++i) mean += red[i][m];
  mean /= (float)n;
  __syncthreads();
  acc = 0.f;
  for (int r = rl; r < n; r += NORM_LANES) {
    const float d = src[(size_t)r * NMEL] - mean;
    acc = fmaf(d, d, acc);
  }
  red[rl][m] = acc;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_LANES; ++i) var += red[i][m];
  var /= (float)n;                                     // biased, as InstanceNorm1d
  const float inv = 1.0f / sqrtf(var + 1e-5f);
  T* __restrict__ dst = feats + (size_t)s * NMEL + m;
  for (int r = rl; r < n; r += NORM_LANES) dst[(size_t)r * NMEL] = from_f32<T>((src[(size_t)r * NMEL] - mean) * inv);
}
}  // namespace

int fbank_init() {
  FbankTables t;
  return get_tables(&t);
}

int launch_fbank(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len, const int32_t* d_row_off,
                 int B, int t_max, float* d_logmel, void* d_feats, int dtype, hipStream_t s) {
  SUGAR_CHECK_ARG(d_wav && d_wav_off && d_wav_len && d_row_off && d_logmel && d_feats, "fbank: null pointer");
  SUGAR_CHECK_ARG(B > 0 && B <= 65535 && t_max > 0, "fbank: bad batch %d / t_max %d", B, t_max);
  SUGAR_CHECK_ARG(dtype == SUGAR_F32 || dtype == SUGAR_BF16, "fbank: bad dtype %d", dtype);
  FbankTables tb;
  int rc = get_tables(&tb);
  if (rc) return rc;
  hipLaunchKernelGGL(fbank_dft_kernel, dim3((unsigned)ceil_div(t_max, FRAMES), (unsigned)B), dim3(256), 0, s, d_wav,
                     d_wav_off, d_wav_len, d_row_off, tb, d_logmel);
  SUGAR_LAUNCH_CHECK();
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(instnorm_kernel<float>, dim3((unsigned)B), dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off,
                       (float*)d_feats);
  else
    hipLaunchKernelGGL(instnorm_kernel<bf16>, dim3((unsigned)B), dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off,
                       (bf16*)d_feats);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
