// Shared helpers for libsugar_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/sugar_amd.h"

namespace sugar {

void set_error(const char* fmt, ...);

#define SUGAR_CHECK_ARG(cond, ...)      \
  do {                                  \
    if (!(cond)) {                      \
      ::sugar::set_error(__VA_ARGS__);  \
      return SUGAR_EINVAL;              \
    }                                   \
  } while (0)

#define SUGAR_HIP(call)                                                                     \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      ::sugar::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      return SUGAR_EHIP;                                                                    \
    }                                                                                       \
  } while (0)

#define SUGAR_LAUNCH_CHECK()                                                                \
  do {                                                                                      \
    hipError_t e__ = hipGetLastError();                                                     \
    if (e__ != hipSuccess) {                                                                \
      ::sugar::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
      return SUGAR_EHIP;                                                                    \
    }                                                                                       \
  } while (0)

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Opt a kernel in to more than 64 KiB of dynamic LDS.  Function attributes are per device, so the "done" state is a
// bit per device ordinal (one atomic word per call site); a race between two threads only repeats the call.
typedef std::atomic<unsigned long long> OptInMask;
static inline int lds_opt_in(const void* fn, int bytes, OptInMask* done_mask) {
  int dev = 0;
  SUGAR_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && ((done_mask->load(std::memory_order_acquire) >> dev) & 1ull)) return SUGAR_OK;
  SUGAR_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (dev >= 0 && dev < 64) done_mask->fetch_or(1ull << dev, std::memory_order_release);
  return SUGAR_OK;
}

// Compute units of the CURRENT device (cached per device ordinal).
int device_cu_count(int* n_cu);

// Tunables (subnet.hip): read ONCE from the environment (SUGAR_<NAME>), afterwards only through sugar_tune_set /
// sugar_tune_get (atomics: safe against concurrent launches).  They exist for parity tests and in-job A/B runs.
enum {
  TUNE_GEMM_WREG_MIN_R = 0,   // K = 512 convs use the weights-in-registers GEMM from this many frames up (19200; pooling.linear1 from 1.5 x that)
  TUNE_GEMM_UNIFORM,          // != 0: wide 1x1 convs on the uniform 256 x 256 grid (no split tail)
  TUNE_GEMM_NO_PINGPONG,      // != 0: both waves of a SIMD in step in the wide GEMM
  TUNE_GENERIC_RES2,          // != 0: Res2 chain as 7 generic launches
  TUNE_RES2_MAX_T,            // frames one Res2 workgroup holds (0 = the kernel's 384)
  TUNE_GENERIC_LAYER1,        // != 0: layer1 through the generic masked-load conv
  TUNE_GEMM_NO_PERSIST,       // != 0: the wide GEMM as one workgroup per tile (no persistent loop)
  TUNE_GEMM_NO_QUAD,          // persistent wide GEMM: 0 quadrant schedule in two phases per k-tile, 2 in four phases, 1 two-half ping-pong
  TUNE_RES2_NO_DB,            // != 0: Res2 chain kernel without the double-buffered u tile (two barriers per step)
  TUNE_RECAL_SEPARATE_STATS,  // != 0: BN recalibration computes every BatchNorm's statistics in a pass of its own (A/B, tests)
  TUNE_COUNT
};
int64_t tune_get(int key);

// Every C-ABI entry that takes a SugarSubnet* runs with the subnet's device current and restores the caller's.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define SUGAR_DEVICE_GUARD(dev)                                                                         \
  ::sugar::DeviceGuard guard__(dev);                                                                    \
  if (guard__.err != hipSuccess) {                                                                      \
    ::sugar::set_error("cannot make device %d current: %s", (dev), hipGetErrorString(guard__.err));      \
    return SUGAR_EHIP;                                                                                  \
  }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

template <typename T>
struct DT;
template <>
struct DT<float> {
  static constexpr int id = SUGAR_F32;
  static constexpr int EPC = 4;  // elements per 16-byte chunk
};
template <>
struct DT<bf16> {
  static constexpr int id = SUGAR_BF16;
  static constexpr int EPC = 8;
};

// v of another lane of the same 16-lane DPP row (CTRL: a DPP control, e.g. 0xB1 = quad_perm [1,0,3,2])
template <int CTRL>
__device__ __forceinline__ float dpp_row_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// tanh of the pooling.linear1 epilogue when the result is stored as bf16: 1 - 2 / (2^(2 x log2 e) + 1) with the hardware
// exp2 / rcp (absolute error ~1e-7, far below the bf16 rounding that follows; exact limits +-1 for large |x|).  tanhf is
// ~30 VALU instructions per element, and pooling.linear1's lone wave per SIMD spent a third of every tile in it.
__device__ __forceinline__ float tanh_bf16_epilogue(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// host-side f32 -> bf16 bits, round to nearest even (weights are finite)
static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// ---- device memory of packed weights (arena.hip) --------------------------------------------------
// Inside an ArenaScope dev_alloc carves 256-byte aligned pieces out of a few large chunks owned by the arena and
// dev_upload fills a host mirror that arena_flush sends with one asynchronous copy per chunk on a private stream;
// dev_free ignores arena memory.  Outside a scope they are hipMalloc / hipMemcpy / hipFree.
struct DevArena {
  struct Chunk {
    char* dev = nullptr;
    char* host = nullptr;     // mirror until arena_flush
    size_t cap = 0, used = 0;
  };
  std::vector<Chunk> chunks;
};
class ArenaScope {            // makes `a` the calling thread's arena for its lifetime
 public:
  explicit ArenaScope(DevArena* a);
  ~ArenaScope();
  ArenaScope(const ArenaScope&) = delete;
  ArenaScope& operator=(const ArenaScope&) = delete;

 private:
  DevArena* prev_;
};
int dev_alloc(void** p, size_t bytes);
template <typename T>
inline int dev_alloc(T** p, size_t bytes) { return dev_alloc((void**)p, bytes); }
int dev_upload(void* d_dst, const void* h_src, size_t bytes);
void dev_free(void* p);
int arena_flush(DevArena* a);      // copies + waits for the upload stream; the mirrors are freed
void arena_release(DevArena* a);   // frees the chunks

// ---- per-row utterance segment table ------------------------------------------------------------
// rowseg[r] = (first row, one-past-last row) of the utterance owning row r.
int launch_build_rowseg(const int32_t* d_row_off, int B, int t_max, int2* d_rowseg, hipStream_t s);

// ---- implicit-GEMM conv (conv_gemm.hip) ---------------------------------------------------------
// ReLU in ONE instruction: fmaxf(v, 0) compiles to a canonicalising v_max_f32 v, v, v in front of the max (the compiler
// cannot know that an MFMA result is no signalling NaN), i.e. two VALU instructions per accumulator in every epilogue
__device__ __forceinline__ float relu_f32(float v) {
  float r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}
enum { EPI_RELU = 1, EPI_AFFINE = 2, EPI_TANH = 4 };

struct ConvLayer {       // packed, device-resident
  void* w = nullptr;     // [n_pad][k_pad] dtype, K index = tap * cin_p + ci
  void* wfrag = nullptr; // K = 512 bf16 1x1 convs: the same weights in gemm_wreg_kernel's fragment-load order (conv1x1.hip)
  float* scale = nullptr;  // [n_pad] (nullptr when no affine)
  float* shift = nullptr;
  int cout = 0;          // valid output channels (padded internal layout)
  int n_pad = 0;         // rows of w
  int cin_p = 0;         // padded input channels (row length of the activation slice read)
  int ktaps = 1;
  int dil = 1;
  int k_pad = 0;
  int flags = 0;
};

struct ConvIO {
  const void* x = nullptr;   // [R][..] activation, column offset already applied
  int ldx = 0;
  const void* x2 = nullptr;  // optional addend, same shape
  int ldx2 = 0;
  void* y = nullptr;
  int ldy = 0;
  void* y2 = nullptr;        // optional duplicate store of columns >= y2_from (pointer has NO column offset)
  int ldy2 = 0;
  int y2_from = 0;
  // BN recalibration: when set, the conv also leaves the statistics partials of its output that launch_bn_batch would
  // otherwise compute in a pass of its own, [row blocks][sum | sum of squares][stat_c] doubles (stat_cap: doubles the
  // table holds), if its kernel can (the generic kernel and the weights-in-registers GEMM, stat_c == the conv's cout);
  // launch_conv reports the row blocks written through `stat_blocks` (0: none, run the statistics pass).
  double* stat_part = nullptr;
  int stat_c = 0;
  int64_t stat_cap = 0;
};

int launch_conv(const ConvLayer& L, const ConvIO& io, const int2* d_rowseg, int64_t R, int dtype, hipStream_t s,
                int* stat_blocks = nullptr);
// true when launch_conv sends this conv to the LDS-DMA 1x1 GEMMs of conv1x1.hip, which do not read d_rowseg
bool gemm1x1_eligible(const ConvLayer& L, const ConvIO& io, int dtype);
int conv_stat_blocks(int64_t R);

// gemm_wreg_kernel with two MFMA tiles per wave (CM = 2): row i of tile m is output channel 8 (i / 4) + 4 m + i % 4 of
// the wave's 32, so that lane (fr, fq), which holds rows 4 fq .. 4 fq + 3 of both tiles, owns channels 8 fq .. 8 fq + 7
// of its frames: one 16-byte store per frame straight from the accumulators, no transposition through LDS.
__host__ __device__ inline int wreg_row(int m, int i) { return 8 * (i >> 2) + 4 * m + (i & 3); }

// Tap conv (dilation 1, bf16) as an LDS-DMA GEMM over overlapping rows of a zero-padded copy of x (conv1x1.hip):
// d_xpad must hold gemm_taps_pad_rows(L, B, R) rows of L.cin_p elements.
bool gemm_taps_eligible(const ConvLayer& L, int ldx, int dtype);
int64_t gemm_taps_pad_rows(const ConvLayer& L, int B, int64_t R);
int gemm_taps_slack(const ConvLayer& L, int B, int64_t R);   // rows that close the padded buffer
int launch_pad_rows(const ConvLayer& L, const void* d_x, void* d_xpad, const int32_t* d_row_off, int B, int64_t R, int t_max,
                    hipStream_t s);
int64_t gemm_taps_orow_ints(const ConvLayer& L, int B, int64_t R);   // int32 entries of the output-row map d_orow
// orow_ready: d_orow already holds the map (the fused front-end wrote it); otherwise a small kernel builds it first
// d_stat / stat_c / stat_io: as ConvIO::stat_part / stat_c and, in: stat_cap, out: the row blocks written (0: none)
int launch_gemm_taps(const ConvLayer& L, const void* d_xpad, int32_t* d_orow, void* d_y, int ldy, const int32_t* d_row_off,
                     int B, int64_t R, int t_max, hipStream_t s, bool orow_ready = false, double* d_stat = nullptr,
                     int stat_c = 0, int64_t* stat_io = nullptr);

// Pack a reference-layout conv weight [cout][cin][k] (host f32) into ConvLayer storage with an internal channel
// permutation / padding: out_map[o] = packed output row of reference channel o, in_map[i] = packed input
// column of reference input channel i (nullptr = identity).  bn may be nullptr.
int pack_conv(ConvLayer* L, const float* h_w, const SugarBN* bn, int cout, int cin, int k, int dil, int flags,
              int cout_p, const int* out_map, int cin_p, const int* in_map, int dtype);
void free_conv(ConvLayer* L);

// ---- fused Res2 chain (res2.hip): all 7 sequential step convs of one block in one kernel ---------
bool res2_chain_eligible(int dtype, int t_max, int wp, int k_pad, int n_pad);
int launch_res2_chain(const ConvLayer* steps, const void* d_x, void* d_y, int ld, int wp, const int32_t* d_row_off,
                      int B, int t_max, hipStream_t s);

// ---- front-end (fbank.hip) ----------------------------------------------------------------------
// d_logmel: fbank_scratch_floats(B, R) floats = [R][80] log-mel rows + the chunk statistics of long utterances
int64_t fbank_scratch_floats(int B, int64_t R);
// halo > 0: d_feats is the zero-padded layout of launch_pad_rows (utterance b's frame t at row row_off[b] + 2 halo b +
// halo + t, halo zero rows either side of every utterance, `slack` more after the last one), written directly
int launch_fbank(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len, const int32_t* d_row_off,
                 int B, int64_t R, int t_max, float* d_logmel, void* d_feats, int dtype, hipStream_t s, int halo = 0,
                 int slack = 0, int32_t* d_orow = nullptr, int64_t orow_n = 0);   // d_orow: also the layer1 GEMM's output-row map

// ---- SE / pooling / tail (se_pool.hip) ----------------------------------------------------------
struct SELayer {            // device, f32, transposed for coalesced reads
  float* w1t = nullptr;     // [c0_p][hid]   (reduce.weight^T)
  float* b1 = nullptr;      // [hid]
  float* w2t = nullptr;     // [hid][c0_p]   (expand.weight^T)
  float* b2 = nullptr;      // [c0_p]
  int c = 0, c_p = 0, hid = 0;
};
// scale[b][c] = sigmoid(W2 relu(W1 mean_t(y[b]) + b1) + b2)
// d_part: se_part_floats floats, the per-(segment, channel slab) shares of the reduce layer (se_pool.hip)
int64_t se_part_floats(const SELayer& se, int dtype, int B, int64_t R);
// d_bn_scale / d_bn_shift (recalibration): y is read BEFORE its train-mode BatchNorm, whose affine of this batch they are
int launch_se_scale(const SELayer& se, const void* d_y, int ldy, const int32_t* d_row_off, int B, int t_max,
                    float* d_part, float* d_scale, int dtype, hipStream_t s, const float* d_bn_scale = nullptr,
                    const float* d_bn_shift = nullptr);
// out = y * scale[b] + S ;  S += out (when write_S)                                    dynamictdnn.py:98-100, :657-660
int launch_se_apply(const void* d_y, int ldy, const float* d_scale, int c_p, void* d_S, int ldS, void* d_out,
                    int ldo, int write_S, const int32_t* d_row_off, int B, int t_max, int dtype, hipStream_t s,
                    const float* d_bn_scale = nullptr, const float* d_bn_shift = nullptr);

struct PoolLayer {
  void* w2 = nullptr;       // [ccat_pad32][bott] dtype : pooling.linear2.weight rows padded to a multiple of 32
  int ccat = 0, ccat_p = 0, bott = 0;
};
// attentive statistics: pooled[b] = [mean(ccat_p) | std(ccat_p)] (f32)
// d_part: pool_part_floats(P, B, R) floats, the segment table of utterances longer than 1024 frames
int64_t pool_part_floats(const PoolLayer& P, int B, int64_t R);
int launch_pool_stats(const PoolLayer& P, const void* d_a, int lda, const void* d_h, int ldh,
                      const int32_t* d_row_off, int B, int t_max, float* d_pooled, float* d_part, int dtype,
                      hipStream_t s);

struct TailLayer {          // folded bnpool . linear . bnembed, f32
  float* wt = nullptr;      // [2*ccat_p][embed]
  float* bias = nullptr;    // [embed]
  int in_p = 0, embed = 0;
};
int tail_splits(const TailLayer& t);   // k-splits of the tail GEMV = slabs of the partial buffer
int launch_tail(const TailLayer& t, const float* d_pooled, int B, float* d_partial, float* d_emb, int normalize,
                hipStream_t s);
int launch_l2norm(float* d_x, int64_t n, int dim, hipStream_t s);

// ---- train-mode BatchNorm for BN recalibration (bn_recal.hip) ------------------------------------
struct BNState {            // device f32 [4][n]: gamma | beta | running_mean | running_var, packed channel order
  float* p = nullptr;
  int n = 0;                // packed channels (padded ones: gamma 0, beta 0, mean 0, var 1)
  int c = 0;                // reference channels
  int* h_map = nullptr;     // host: reference channel -> packed index
  float* gamma() const { return p; }
  float* beta() const { return p + n; }
  float* mean() const { return p + 2 * (size_t)n; }
  float* var() const { return p + 3 * (size_t)n; }
};
int bn_state_create(BNState* st, const SugarBN* bn, int c, int n_packed, const int* map);
void bn_state_free(BNState* st);
int64_t bn_workspace_doubles(int c_max, int64_t R);
// z [R][C] (row stride ld, compute dtype) <- train-mode BN of z, in place; running statistics of st updated.
// part_blocks > 0: d_part already holds that many row blocks of partials (written by the conv that produced z,
// ConvIO::stat_part) and the statistics pass over z is skipped.
int launch_bn_batch(void* d_z, int ld, int64_t R, int C, const BNState& st, float momentum, double* d_part,
                    float* d_scale, float* d_shift, int dtype, hipStream_t s, int part_blocks = 0, bool apply = true);
// (apply = false: statistics, running update and this batch's (scale, shift) only; the consumer applies the affine)
// x [B][ld] f32 <- train-mode BN over the B rows, in place (st.n channels)
int launch_bn_rows(float* d_x, int ld, int B, const BNState& st, float momentum, hipStream_t s);
// (scale, shift) of eval-mode BN from the running statistics
int launch_bn_fold(const BNState& st, float* d_scale, float* d_shift, hipStream_t s);

// dtype -> f32 copy of a strided [R][C] activation (debug taps)
int launch_to_f32(const void* d_x, int ldx, int64_t R, int C, float* d_out, int dtype, hipStream_t s);

}  // namespace sugar
