// Fused log-mel front-end (gfx950).
//
// Replaces LogMelFbanks.forward (sugar/transforms.py:60-77): PreEmphasis (:22-27) -> torchaudio MelSpectrogram
// (:50-52: torch.stft n_fft 512 / hop 160 / periodic Hamming(400) centred in the 512 frame / center=True reflect
// padding, |X|^2, HTK triangular mel filters 20..7600 Hz, norm=None) -> +1e-6 -> log -> InstanceNorm1d(80).
//
// Kernel 1 (fbank_fft_kernel): one workgroup = 32 frames of one utterance, HBM-bound by design (each sample is read
// once, each log-mel value written once: 288 320 B per 3 s utterance).
//   * the 5 360 samples the 32 frames touch are loaded once into LDS, reflect-padded at the utterance's true ends
//     and pre-emphasised on the fly;
//   * each wavefront transforms one frame at a time: the 512-point real FFT is a 256-point complex FFT of
//     z[m] = x[2m] + i x[2m+1] (the window only covers 400 samples; a frame's position inside the 512-point buffer
//     only changes the phase, not |X|^2), done as 4 radix-4 Stockham passes with one butterfly per lane and the
//     exchange between passes through 4 KB of wave-private LDS; window and twiddle factors live in registers;
//   * split-radix post-processing X[k] = E[k] + W512^k O[k], |X|^2 for bins 0..255 (the Nyquist bin has zero mel
//     weight: f_max = 7600 Hz < 8000 Hz), the 477-tap sparse mel projection and log run on the VALU;
//   * the [32 x 80] log-mel tile leaves through LDS as one contiguous, coalesced 10 KB store.
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
constexpr int FRAMES = 32;                // frames per workgroup
constexpr int SPAN = (FRAMES - 1) * HOP + WIN;   // 5360 samples
constexpr int MEL_MAXW = 20;              // table row length: max non-zero taps of one mel filter (16 for this filterbank)
constexpr int MEL_W0 = 10;                // filters 0..63 have at most 10 taps: lane j projects filter j
constexpr int MEL_W1 = 4;                 // filters 64..79 (<= 16 taps): lane j takes taps 4 (j & 3) .. + 3 of filter 64 + j / 4

struct FbankTables {
  float2* tw = nullptr;      // [512] exp(-2 pi i n / 512)
  float* win = nullptr;      // [512] periodic Hamming(400), zero for n >= 400
  int* mel_lo = nullptr;     // [NMEL]
  int* mel_cnt = nullptr;    // [NMEL]
  float* mel_w = nullptr;    // [NMEL][MEL_MAXW]
};

// The tables and the scratch of the allocation-free entry point live in file-scope __device__ arrays: every GPU the
// library runs on holds its own copy at its own address, found once per device with hipGetSymbolAddress (in host code
// the array NAMES are host-side stand-ins, never device addresses) and kept in g_dev.  Nothing here calls hipMalloc.
__device__ __align__(256) float2 d_tw[NFFT];
__device__ __align__(256) float d_win[NFFT];
__device__ __align__(256) int d_mel_lo[NMEL];
__device__ __align__(256) int d_mel_cnt[NMEL];
__device__ __align__(256) float d_mel_w[NMEL * MEL_MAXW];

// scratch of sugar_fbank_fwd_noalloc: the log-mel rows before normalisation + the chunk statistics of long utterances
// (what d_logmel brings in for sugar_fbank_fwd), for at most NOALLOC_FRAMES frames in NOALLOC_BATCH utterances: 5.44 MB
constexpr int NOALLOC_FRAMES = SUGAR_NOALLOC_MAX_FRAMES, NOALLOC_BATCH = SUGAR_NOALLOC_MAX_BATCH;
constexpr int64_t NOALLOC_FLOATS =
    (int64_t)NOALLOC_FRAMES * NMEL + (NOALLOC_FRAMES / (8 * 40) + NOALLOC_BATCH + 1) * (2 * NMEL);
static __device__ __align__(256) float scratch[NOALLOC_FLOATS];

struct DevState {
  bool ready = false;
  FbankTables tb;
  float* scratch = nullptr;     // device address of `scratch` on this GPU
};
constexpr int MAX_DEVICES = 64;
std::mutex g_mu;
DevState g_dev[MAX_DEVICES];

double hz_to_mel(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
double mel_to_hz(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

// Per-device set-up, with that device current: symbol addresses + table upload.  Runs once per device, from
// sugar_subnet_create (the library's init) or from the first front-end call on a device.
int get_dev(DevState* out) {
  int dev = 0;
  SUGAR_HIP(hipGetDevice(&dev));
  SUGAR_CHECK_ARG(dev >= 0 && dev < MAX_DEVICES, "fbank: device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_mu);
  DevState& t = g_dev[dev];
  if (!t.ready) {
    const double PI = 3.14159265358979323846;
    std::vector<float> tw(2 * NFFT), win(NFFT, 0.f);
    for (int n = 0; n < NFFT; ++n) {
      tw[2 * n] = (float)cos(2.0 * PI * n / NFFT);
      tw[2 * n + 1] = (float)(-sin(2.0 * PI * n / NFFT));
    }
    for (int n = 0; n < WIN; ++n) win[n] = (float)(0.54 - 0.46 * cos(2.0 * PI * n / WIN));  // torch.hamming_window(400)
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
          mw[(size_t)m * MEL_MAXW + (k - first)] = (float)v * 0.25f;   // the kernel's power spectrum is |2 X|^2
          n = k - first + 1;
        }
      }
      lo[m] = first < 0 ? 0 : first;
      cnt[m] = n;
      if (n > (m < 64 ? MEL_W0 : 4 * MEL_W1)) { set_error("fbank: mel filter %d has %d taps", m, n); return SUGAR_EINVAL; }
    }
    FbankTables tb;
    SUGAR_HIP(hipGetSymbolAddress((void**)&tb.tw, HIP_SYMBOL(d_tw)));
    SUGAR_HIP(hipGetSymbolAddress((void**)&tb.win, HIP_SYMBOL(d_win)));
    SUGAR_HIP(hipGetSymbolAddress((void**)&tb.mel_lo, HIP_SYMBOL(d_mel_lo)));
    SUGAR_HIP(hipGetSymbolAddress((void**)&tb.mel_cnt, HIP_SYMBOL(d_mel_cnt)));
    SUGAR_HIP(hipGetSymbolAddress((void**)&tb.mel_w, HIP_SYMBOL(d_mel_w)));
    float* sc = nullptr;
    SUGAR_HIP(hipGetSymbolAddress((void**)&sc, HIP_SYMBOL(scratch)));
    SUGAR_HIP(hipMemcpy(tb.tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMemcpy(tb.win, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMemcpy(tb.mel_lo, lo.data(), NMEL * sizeof(int), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMemcpy(tb.mel_cnt, cnt.data(), NMEL * sizeof(int), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMemcpy(tb.mel_w, mw.data(), mw.size() * sizeof(float), hipMemcpyHostToDevice));
    t.tb = tb;
    t.scratch = sc;
    t.ready = true;
  }
  *out = t;
  return SUGAR_OK;
}

// Padded indices into a wave's FFT exchange buffer (float2 entries, 32 banks of 4 bytes = 16 entries per LDS pass).
// Every exchange is one scattered write and one contiguous read of the same layout, so each may pick its own padding:
// PX (one pad entry per 16) makes the writes of passes 0 and 2 conflict-free; pass 1 writes entries 16 g + e + 4 u
// (g = j / 4, e = j % 4) and wants FOUR pad entries per 16 so that four consecutive g land 8 banks apart (with PX they
// overlap: up to 4 lanes per bank pair -- 41 % of the kernel's LDS cycles were bank conflicts).
__device__ __forceinline__ int PX(int i) { return i + (i >> 4); }
__device__ __forceinline__ int PX4(int i) { return i + ((i >> 4) << 2); }

// Packed FP32 by hand.  The library is compiled WITHOUT packed-FP32 instructions because one operand form gives wrong
// results next to MFMA waves (the low result taking the HIGH half of src1).  The complex arithmetic of
// this kernel is two-lane by nature and the kernel is VALU-issue-bound, so its butterflies use v_pk_* through inline
// asm in exactly the forms that tools/pkfp32_probe.hip ran 1.3e10 times each beside an MFMA burner with 0 wrong
// results: no modifiers; neg_* on src1; the half swap on SRC0 (op_sel:[1,0] op_sel_hi:[0,1]) alone or with neg_lo /
// neg_hi on src0; FMA with the src0 swap.  A complex number is an aligned VGPR pair (re, im).
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define SUGAR_PK2(name, text)                                              \
  __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b) {                \
    f32x2 r;                                                               \
    asm(text : "=v"(r) : "v"(a), "v"(b));                                  \
    return r;                                                              \
  }
SUGAR_PK2(pk_add, "v_pk_add_f32 %0, %1, %2")                                               // a + b
SUGAR_PK2(pk_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                     // a - b
SUGAR_PK2(pk_addc, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                 // a + conj b
SUGAR_PK2(pk_subc, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                 // a - conj b
SUGAR_PK2(pk_mul, "v_pk_mul_f32 %0, %1, %2")                                               // (a.re b.re, a.im b.im)
SUGAR_PK2(pk_mi_add, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]")  // b - i a
SUGAR_PK2(pk_pi_add, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]")  // b + i a
#undef SUGAR_PK2
// (a.im b.re + c.re, a.re b.im + c.im)
__device__ __forceinline__ f32x2 pk_fma_swap0(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// a twiddle w as the two pairs the packed complex product wants: (w.re, w.re) and (-w.im, w.im)
struct Twid {
  f32x2 rr, ii;
};
__device__ __forceinline__ Twid twid(float2 w) { return Twid{f32x2{w.x, w.x}, f32x2{-w.y, w.y}}; }
// a w = (a.re w.re - a.im w.im, a.im w.re + a.re w.im): two instructions
__device__ __forceinline__ f32x2 cmul(f32x2 a, const Twid& w) { return pk_fma_swap0(a, w.ii, pk_mul(a, w.rr)); }

// in-place DFT of size 4 (forward, e^{-2 pi i u t / 4}):  y0 = a+b+c+d, y1 = a - i b - c + i d, y2 = a-b+c-d, y3 = a + i b - c - i d
__device__ __forceinline__ void fft4(f32x2& a, f32x2& b, f32x2& c, f32x2& d) {
  const f32x2 s0 = pk_add(a, c), d0 = pk_sub(a, c), s1 = pk_add(b, d), d1 = pk_sub(b, d);
  a = pk_add(s0, s1);
  c = pk_sub(s0, s1);
  b = pk_mi_add(d1, d0);   // d0 - i d1
  d = pk_pi_add(d1, d0);   // d0 + i d1
}

__global__ __launch_bounds__(256, 4) void fbank_fft_kernel(const float* __restrict__ wav,
                                                        const int64_t* __restrict__ wav_off,
                                                        const int32_t* __restrict__ wav_len,
                                                        const int32_t* __restrict__ row_off, FbankTables tb,
                                                        float* __restrict__ logmel) {
  __shared__ __attribute__((aligned(16))) float samples[SPAN];
  // per wave ONE 2 KB exchange buffer: every pass reads its 4 inputs into registers before it writes, and a
  // wavefront's LDS instructions execute in program order, so passes run in place; |X|^2 reuses the same bytes
  // (entry i of the exchange buffer sits at i + i / 16: with plain indexing the scattered writes of passes 0-2 put up
  // to 16 lanes on one bank pair -- 34 % of the kernel's LDS cycles were bank conflicts; with one pad slot per 16 entries
  // every pass reads and writes 4 lanes per bank pair, the minimum for 64 lanes x 8 bytes)
  __shared__ __attribute__((aligned(16))) float2 fbuf[4][NBIN + NBIN / 4];

  const int b = blockIdx.y;
  const int row0 = row_off[b];
  const int T = row_off[b + 1] - row0;
  const int t0 = blockIdx.x * FRAMES;
  if (t0 >= T) return;                               // uniform per workgroup
  const int L = wav_len[b];
  const float* __restrict__ x = wav + wav_off[b];
  const int tid = threadIdx.x, j = tid & 63, wv = tid >> 6;
  const int nf = min(FRAMES, T - t0);

  // ---- stage pre-emphasised, reflect-padded samples + the mel weights ------------------------------
  const int base = t0 * HOP - WIN / 2;               // original index of buffer position 0 (= t0*160 - 256 + 56)
  const int span = (nf - 1) * HOP + WIN;
  // all 2 x 21 loads of a thread are issued before the first use: a rolled loop here paid one HBM round trip per
  // iteration and was 3/4 of the kernel's time
  constexpr int NSTG = (SPAN + 255) / 256;
  float xc[NSTG], xp[NSTG];
#pragma unroll
  for (int i = 0; i < NSTG; ++i) {
    const int p = min(tid + 256 * i, span - 1);
    int idx = base + p;
    if (idx < 0) idx = -idx;                         // reflect (torch.stft center=True, pad_mode="reflect")
    if (idx >= L) idx = 2 * (L - 1) - idx;
    idx = min(max(idx, 0), L - 1);
    const int prev = idx > 0 ? idx - 1 : 1;          // PreEmphasis reflect pad (1, 0): x[-1] := x[1]
    xc[i] = x[idx];
    xp[i] = x[min(prev, L - 1)];
  }
#pragma unroll
  for (int i = 0; i < NSTG; ++i) {
    const int p = tid + 256 * i;
    if (p < span) samples[p] = xc[i] - 0.97f * xp[i];
  }

  // ---- per-lane constants: window taps, twiddles of passes 1..3, W512^k of the real-FFT split -------
  f32x2 wn[4];                                       // window at samples 2m, 2m+1, m = j + 64 t
#pragma unroll
  for (int t = 0; t < 4; ++t) wn[t] = f32x2{tb.win[2 * (j + 64 * t)], tb.win[2 * (j + 64 * t) + 1]};
  Twid tw1[3], tw2[3], tw3[3], twp[4];
#pragma unroll
  for (int t = 1; t < 4; ++t) {
    tw1[t - 1] = twid(tb.tw[(t * (j & 3) * 32) & 511]);    // exp(-2 pi i t k / 16),  k = j % 4
    tw2[t - 1] = twid(tb.tw[(t * (j & 15) * 8) & 511]);    // exp(-2 pi i t k / 64),  k = j % 16
    tw3[t - 1] = twid(tb.tw[(t * j * 2) & 511]);           // exp(-2 pi i t k / 256), k = j
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {                      // -i W512^k, k = j + 64 t: the real-FFT split's rotation and twiddle in one
    const float2 w = tb.tw[j + 64 * t];
    twp[t] = twid(make_float2(w.y, -w.x));
  }
  // mel projection: 477 filter taps in all.  Lane j projects filter j (<= 10 taps); the 16 wide filters 64..79 are
  // split over quads of lanes, 4 taps each, and added up with two DPP steps.  (One lane per filter with 2 x 20 taps
  // was 40 LDS reads per lane and frame -- a third of the kernel's LDS bytes, and the kernel is bound by those.)
  const int mel0 = j, mel1 = 64 + (j >> 2);
  const int lo0 = tb.mel_lo[mel0];
  const int lo1 = tb.mel_lo[mel1] + MEL_W1 * (j & 3);
  // taps in registers (table rows are zero-padded to MEL_MAXW)
  float mw0[MEL_W0], mw1[MEL_W1];
#pragma unroll
  for (int i = 0; i < MEL_W0; ++i) mw0[i] = tb.mel_w[mel0 * MEL_MAXW + i];
#pragma unroll
  for (int i = 0; i < MEL_W1; ++i) mw1[i] = tb.mel_w[mel1 * MEL_MAXW + MEL_W1 * (j & 3) + i];
  __syncthreads();

  f32x2* A = (f32x2*)fbuf[wv];
  f32x2* Bf = (f32x2*)fbuf[wv];
  float* pw = (float*)fbuf[wv];
  for (int f = wv; f < nf; f += 4) {
    const f32x2* sp = (const f32x2*)(samples + f * HOP);     // 160 f floats: 8-byte aligned
    f32x2 v0, v1, v2, v3;
    // pass 0 (Ns = 1, no twiddles): z[m] = x[2m] w[2m] + i x[2m+1] w[2m+1], zero beyond the 400-sample window
    {
      const f32x2 s0 = sp[j], s1 = sp[j + 64], s2 = sp[j + 128];
      const f32x2 s3 = j < 8 ? sp[j + 192] : f32x2{0.f, 0.f};
      v0 = pk_mul(s0, wn[0]);
      v1 = pk_mul(s1, wn[1]);
      v2 = pk_mul(s2, wn[2]);
      v3 = pk_mul(s3, wn[3]);
      fft4(v0, v1, v2, v3);
      A[PX(4 * j)] = v0; A[PX(4 * j + 1)] = v1; A[PX(4 * j + 2)] = v2; A[PX(4 * j + 3)] = v3;
    }
    __builtin_amdgcn_wave_barrier();
    // pass 1 (Ns = 4)
    {
      v0 = A[PX(j)]; v1 = cmul(A[PX(j + 64)], tw1[0]); v2 = cmul(A[PX(j + 128)], tw1[1]); v3 = cmul(A[PX(j + 192)], tw1[2]);
      fft4(v0, v1, v2, v3);
      const int o = ((j >> 2) << 4) + (j & 3);
      Bf[PX4(o)] = v0; Bf[PX4(o + 4)] = v1; Bf[PX4(o + 8)] = v2; Bf[PX4(o + 12)] = v3;
    }
    __builtin_amdgcn_wave_barrier();
    // pass 2 (Ns = 16)
    {
      v0 = Bf[PX4(j)]; v1 = cmul(Bf[PX4(j + 64)], tw2[0]); v2 = cmul(Bf[PX4(j + 128)], tw2[1]); v3 = cmul(Bf[PX4(j + 192)], tw2[2]);
      fft4(v0, v1, v2, v3);
      const int o = ((j >> 4) << 6) + (j & 15);
      A[PX(o)] = v0; A[PX(o + 16)] = v1; A[PX(o + 32)] = v2; A[PX(o + 48)] = v3;
    }
    __builtin_amdgcn_wave_barrier();
    // pass 3 (Ns = 64): natural-order Z[0..255] lands in Bf
    {
      v0 = A[PX(j)]; v1 = cmul(A[PX(j + 64)], tw3[0]); v2 = cmul(A[PX(j + 128)], tw3[1]); v3 = cmul(A[PX(j + 192)], tw3[2]);
      fft4(v0, v1, v2, v3);
      Bf[PX(j)] = v0; Bf[PX(j + 64)] = v1; Bf[PX(j + 128)] = v2; Bf[PX(j + 192)] = v3;
    }
    __builtin_amdgcn_wave_barrier();
    // real-FFT split: E = (Z[k] + conj Z[256-k]) / 2, O = -i (Z[k] - conj Z[256-k]) / 2, X = E + W512^k O, i.e.
    // 2 X = (Z[k] + conj Zm) + (-i W512^k) (Z[k] - conj Zm); the power |2 X|^2 carries a factor 4 that the mel taps
    // take back (they are stored times 0.25: exact).
    // (Z[k], k = j + 64 t, is what this lane has just written: only the mirror bins come back from LDS)
    const f32x2 zks[4] = {v0, v1, v2, v3};
    f32x2 zms[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zms[t] = Bf[PX((NBIN - j - 64 * t) & (NBIN - 1))];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = j + 64 * t;
      const f32x2 x2 = pk_add(pk_addc(zks[t], zms[t]), cmul(pk_subc(zks[t], zms[t]), twp[t]));
      const f32x2 q = pk_mul(x2, x2);
      pw[k] = q.x + q.y;
    }
    __builtin_amdgcn_wave_barrier();
    // sparse mel projection + log
    {
      // taps past a filter's end have zero weight; lo + i may run a few floats past bin 255: still inside this wave's
      // buffer (finite leftovers of pass 3)
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MEL_W0; ++i) s = fmaf(pw[lo0 + i], mw0[i], s);
#pragma unroll
      for (int i = 0; i < MEL_W1; ++i) s2 = fmaf(pw[lo1 + i], mw1[i], s2);
      s2 += dpp_row_f32<0xB1>(s2);                       // + the neighbour's 4 taps (quad_perm [1,0,3,2])
      s2 += dpp_row_f32<0x4E>(s2);                       // + the other pair's 8 (quad_perm [2,3,0,1])
      // straight to the log-mel rows (256 + 64 contiguous bytes per frame): a staging tile in LDS cost 10 KiB per
      // workgroup and, with it, the fourth workgroup per CU
      float* __restrict__ dst = logmel + (size_t)(row0 + t0 + f) * NMEL;
      // log = log2 * ln 2 with the hardware log2 (arguments >= 1e-6: no denormals; logf's range handling is ~20 VALU)
      dst[mel0] = __builtin_amdgcn_logf(s + 1e-6f) * 0.6931471805599453f;
      if ((j & 3) == 0) dst[mel1] = __builtin_amdgcn_logf(s2 + 1e-6f) * 0.6931471805599453f;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

constexpr int NORM_LANES = 8;   // row lanes per mel bin
constexpr int NORM_CACHE = 40;  // rows per lane kept in registers

// Output rows: utterance b's frame t at row row_off[b] + t, or -- halo > 0, the layout the layer1 tap GEMM reads
// (conv1x1.hip, launch_pad_rows) -- at row_off[b] + 2 halo b + halo + t with the halo rows (and, behind the last
// utterance, `slack` more rows) zeroed here.
// orow (optional, orow_n entries): the layer1 GEMM's output-row map of that padded row space (conv1x1.hip,
// build_orow_kernel): frame t of this part (t_lo <= t < t_hi) -> row_off[b] + t, halo / slack rows -> -1.
template <typename T>
__device__ __forceinline__ T* padded_rows(T* feats, int s, int n, int b, int n_utts, int halo, int slack, bool first_part,
                                          bool last_part, int32_t* orow, int orow_n, int t_lo, int t_hi) {
  if (!halo) return feats + (size_t)s * NMEL;
  T* base = feats + ((size_t)s + (size_t)2 * halo * b + halo) * NMEL;
  if (orow) {
    // GEMM row q0 + t reads the K vector that STARTS at padded row q0 + t (taps t - halo .. t + halo of frame t)
    const int q0 = s + 2 * halo * b;
    if (last_part) {
      const int after = 2 * halo + (b == n_utts - 1 ? slack : 0);
      for (int i = threadIdx.x; i < after && q0 + n + i < orow_n; i += blockDim.x) orow[q0 + n + i] = -1;
    }
    for (int t = t_lo + threadIdx.x; t < t_hi; t += blockDim.x) orow[q0 + t] = s + t;
  }
  if (first_part)
    for (int i = threadIdx.x; i < halo * NMEL; i += blockDim.x) (base - halo * NMEL)[i] = from_f32<T>(0.f);
  if (last_part) {
    const int after = (halo + (b == n_utts - 1 ? slack : 0)) * NMEL;
    for (int i = threadIdx.x; i < after; i += blockDim.x) (base + (size_t)n * NMEL)[i] = from_f32<T>(0.f);
  }
  return base;
}

template <typename T>
__global__ __launch_bounds__(NMEL* NORM_LANES) void instnorm_kernel(const float* __restrict__ logmel,
                                                                     const int32_t* __restrict__ row_off,
                                                                     T* __restrict__ feats, int halo, int slack,
                                                                     int32_t* __restrict__ orow, int orow_n) {
  __shared__ float red[NORM_LANES][NMEL];
  const int b = blockIdx.x;
  const int s = row_off[b], n = row_off[b + 1] - s;
  const int m = threadIdx.x % NMEL, rl = threadIdx.x / NMEL;
  const float* __restrict__ src = logmel + (size_t)s * NMEL + m;
  // the first NORM_CACHE rows of every row lane stay in registers (all loads in flight at once; utterances up to
  // NORM_LANES * NORM_CACHE = 320 frames are read exactly once); longer ones re-read the remainder
  float vals[NORM_CACHE];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) {
    const int r = rl + NORM_LANES * i;
    vals[i] = src[(size_t)min(r, n - 1) * NMEL];
  }
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) acc += (rl + NORM_LANES * i < n) ? vals[i] : 0.f;
  for (int r = rl + NORM_LANES * NORM_CACHE; r < n; r += NORM_LANES) acc += src[(size_t)r * NMEL];
  red[rl][m] = acc;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_LANES; ++i) mean += red[i][m];
  mean /= (float)n;
  __syncthreads();
  acc = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) {
    const float d = (rl + NORM_LANES * i < n) ? vals[i] - mean : 0.f;
    acc = fmaf(d, d, acc);
  }
  for (int r = rl + NORM_LANES * NORM_CACHE; r < n; r += NORM_LANES) {
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
  T* __restrict__ dst = padded_rows(feats, s, n, b, (int)gridDim.x, halo, slack, true, true, orow, orow_n, 0, n) + m;
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) {
    const int r = rl + NORM_LANES * i;
    if (r < n) dst[(size_t)r * NMEL] = from_f32<T>((vals[i] - mean) * inv);
  }
  for (int r = rl + NORM_LANES * NORM_CACHE; r < n; r += NORM_LANES)
    dst[(size_t)r * NMEL] = from_f32<T>((src[(size_t)r * NMEL] - mean) * inv);
}

// ---- utterances longer than NORM_LANES * NORM_CACHE = 320 frames: chunks of 320 frames, one workgroup each ----------
// One workgroup per utterance is a serial loop over the frames (60 s of audio: 750 dependent trips per lane, 0.38 ms
// for 16 utterances).  Chunk statistics (mean_c, M2_c = sum (x - mean_c)^2, both from registers exactly as above) go
// to a small table; every chunk's apply workgroup merges the table of its utterance (Chan et al.:
// M2 = sum M2_c + n_c (mean_c - mean)^2) and normalises its rows.  An utterance of one chunk merges to exactly the
// numbers of instnorm_kernel, so an utterance's features do not depend on what else is in the batch.
constexpr int NORM_CHUNK = NORM_LANES * NORM_CACHE;
static_assert(NOALLOC_FLOATS == (int64_t)NOALLOC_FRAMES * NMEL + (NOALLOC_FRAMES / NORM_CHUNK + NOALLOC_BATCH + 1) * (2 * NMEL),
              "scratch of the allocation-free entry point = fbank_scratch_floats(NOALLOC_BATCH, NOALLOC_FRAMES)");

__device__ __forceinline__ int norm_slot(const int32_t* row_off, int b) { return row_off[b] / NORM_CHUNK + b; }

__global__ __launch_bounds__(NMEL* NORM_LANES) void instnorm_stats_kernel(const float* __restrict__ logmel,
                                                                           const int32_t* __restrict__ row_off,
                                                                           float* __restrict__ part) {
  __shared__ float red[NORM_LANES][NMEL];
  const int b = blockIdx.y, c = blockIdx.x;
  const int s = row_off[b], nu = row_off[b + 1] - s;
  const int r0 = c * NORM_CHUNK;
  if (r0 >= nu) return;
  const int n = min(NORM_CHUNK, nu - r0);
  const int m = threadIdx.x % NMEL, rl = threadIdx.x / NMEL;
  const float* __restrict__ src = logmel + (size_t)(s + r0) * NMEL + m;
  float vals[NORM_CACHE];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) vals[i] = src[(size_t)min(rl + NORM_LANES * i, n - 1) * NMEL];
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) acc += (rl + NORM_LANES * i < n) ? vals[i] : 0.f;
  red[rl][m] = acc;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_LANES; ++i) mean += red[i][m];
  mean /= (float)n;
  __syncthreads();
  acc = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) {
    const float d = (rl + NORM_LANES * i < n) ? vals[i] - mean : 0.f;
    acc = fmaf(d, d, acc);
  }
  red[rl][m] = acc;
  __syncthreads();
  if (rl == 0) {
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_LANES; ++i) m2 += red[i][m];
    float* p = part + (size_t)(norm_slot(row_off, b) + c) * (2 * NMEL);
    p[m] = mean;
    p[NMEL + m] = m2;
  }
}

template <typename T>
__global__ __launch_bounds__(NMEL* NORM_LANES) void instnorm_apply_kernel(const float* __restrict__ logmel,
                                                                           const int32_t* __restrict__ row_off,
                                                                           const float* __restrict__ part,
                                                                           T* __restrict__ feats, int halo, int slack,
                                                                           int32_t* __restrict__ orow, int orow_n) {
  const int b = blockIdx.y, c = blockIdx.x;
  const int s = row_off[b], nu = row_off[b + 1] - s;
  const int r0 = c * NORM_CHUNK;
  if (r0 >= nu) return;
  const int n = min(NORM_CHUNK, nu - r0);
  const int m = threadIdx.x % NMEL, rl = threadIdx.x / NMEL;
  const float* __restrict__ src = logmel + (size_t)(s + r0) * NMEL + m;
  float vals[NORM_CACHE];
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) vals[i] = src[(size_t)min(rl + NORM_LANES * i, n - 1) * NMEL];
  const float* __restrict__ p = part + (size_t)norm_slot(row_off, b) * (2 * NMEL);
  const int nc = (nu + NORM_CHUNK - 1) / NORM_CHUNK;
  float mean, var;
  if (nc == 1) {
    mean = p[m];
    var = p[NMEL + m] / (float)nu;
  } else {
    double sum = 0.0;
    for (int k = 0; k < nc; ++k) sum += (double)min(NORM_CHUNK, nu - k * NORM_CHUNK) * (double)p[k * 2 * NMEL + m];
    const double mu = sum / (double)nu;
    double m2 = 0.0;
    for (int k = 0; k < nc; ++k) {
      const double d = (double)p[k * 2 * NMEL + m] - mu;
      m2 += (double)p[k * 2 * NMEL + NMEL + m] + (double)min(NORM_CHUNK, nu - k * NORM_CHUNK) * d * d;
    }
    mean = (float)mu;
    var = (float)(m2 / (double)nu);                    // biased, as InstanceNorm1d
  }
  const float inv = 1.0f / sqrtf(var + 1e-5f);
  T* __restrict__ dst = padded_rows(feats, s, nu, b, (int)gridDim.y, halo, slack, c == 0, r0 + NORM_CHUNK >= nu, orow, orow_n,
                                    r0, min(nu, r0 + NORM_CHUNK)) +
                        (size_t)r0 * NMEL + m;
#pragma unroll
  for (int i = 0; i < NORM_CACHE; ++i) {
    const int r = rl + NORM_LANES * i;
    if (r < n) dst[(size_t)r * NMEL] = from_f32<T>((vals[i] - mean) * inv);
  }
}
}  // namespace

int fbank_init() {
  DevState t;
  return get_dev(&t);
}

int64_t fbank_scratch_floats(int B, int64_t R) { return R * NMEL + (R / NORM_CHUNK + B + 1) * (2 * NMEL); }

int launch_fbank(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len, const int32_t* d_row_off,
                 int B, int64_t R, int t_max, float* d_logmel, void* d_feats, int dtype, hipStream_t s, int halo, int slack,
                 int32_t* d_orow, int64_t orow_n) {
  SUGAR_CHECK_ARG(!d_orow || (halo > 0 && orow_n > 0 && orow_n < (1ll << 31)), "fbank: output-row map without a padded layout");
  SUGAR_CHECK_ARG(d_wav && d_wav_off && d_wav_len && d_row_off && d_logmel && d_feats, "fbank: null pointer");
  SUGAR_CHECK_ARG(B > 0 && B <= 65535 && t_max > 0 && R >= t_max, "fbank: bad batch %d / t_max %d / R %lld", B, t_max, (long long)R);
  SUGAR_CHECK_ARG(dtype == SUGAR_F32 || dtype == SUGAR_BF16, "fbank: bad dtype %d", dtype);
  DevState ds;
  int rc = get_dev(&ds);
  if (rc) return rc;
  const FbankTables tb = ds.tb;
  hipLaunchKernelGGL(fbank_fft_kernel, dim3((unsigned)ceil_div(t_max, FRAMES), (unsigned)B), dim3(256), 0, s, d_wav,
                     d_wav_off, d_wav_len, d_row_off, tb, d_logmel);
  SUGAR_LAUNCH_CHECK();
  if (t_max > NORM_CHUNK) {
    float* part = d_logmel + R * NMEL;                  // chunk statistics live behind the log-mel rows
    const dim3 grid((unsigned)ceil_div(t_max, NORM_CHUNK), (unsigned)B);
    hipLaunchKernelGGL(instnorm_stats_kernel, grid, dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off, part);
    SUGAR_LAUNCH_CHECK();
    if (dtype == SUGAR_F32)
      hipLaunchKernelGGL(instnorm_apply_kernel<float>, grid, dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off, part,
                         (float*)d_feats, halo, slack, d_orow, (int)orow_n);
    else
      hipLaunchKernelGGL(instnorm_apply_kernel<bf16>, grid, dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off, part,
                         (bf16*)d_feats, halo, slack, d_orow, (int)orow_n);
    SUGAR_LAUNCH_CHECK();
    return SUGAR_OK;
  }
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(instnorm_kernel<float>, dim3((unsigned)B), dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off,
                       (float*)d_feats, halo, slack, d_orow, (int)orow_n);
  else
    hipLaunchKernelGGL(instnorm_kernel<bf16>, dim3((unsigned)B), dim3(NMEL * NORM_LANES), 0, s, d_logmel, d_row_off,
                       (bf16*)d_feats, halo, slack, d_orow, (int)orow_n);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar

// Allocation-free variant of sugar_fbank_fwd (include/sugar_amd.h): same kernels, the log-mel scratch is this GPU's
// copy of the file-scope `scratch` array instead of a caller buffer.  After the one-off per-device set-up (get_dev: done
// by sugar_subnet_create, else by the first front-end call on the device) it calls no allocating routine.
extern "C" int sugar_fbank_fwd_noalloc(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len,
                                       const int32_t* d_row_off, int B, int64_t R, int t_max, void* d_feats, int dtype,
                                       void* stream) {
  using namespace sugar;
  SUGAR_CHECK_ARG(B > 0 && B <= NOALLOC_BATCH && R > 0 && R <= NOALLOC_FRAMES,
                  "fbank_fwd_noalloc: %d utterances / %lld frames exceed the static scratch (at most %d utterances, %d "
                  "frames per call); use sugar_fbank_fwd with a caller buffer",
                  B, (long long)R, NOALLOC_BATCH, NOALLOC_FRAMES);
  DevState ds;
  int rc = get_dev(&ds);
  if (rc) return rc;
  SUGAR_CHECK_ARG(fbank_scratch_floats(B, R) <= NOALLOC_FLOATS, "fbank_fwd_noalloc: scratch too small");
  return launch_fbank(d_wav, d_wav_off, d_wav_len, d_row_off, B, R, t_max, ds.scratch, d_feats, dtype, (hipStream_t)stream);
}

