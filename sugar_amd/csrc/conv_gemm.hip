// Implicit-GEMM dilated conv1d on MFMA (gfx950), channels-last packed rows.
//
// Replaces F.conv1d + ReLU + F.batch_norm of DynamicConv1dReLUBN.forward (sugar/models/dynamictdnn.py:182-183),
// the Res2 step convs (:290-303), catconv + ReLU (:663) and pooling.linear1 + tanh (:509).
//
//   D[co][r] = sum_{tap, ci} W[co][tap*Cin + ci] * X[r + (tap - ctr)*dil][ci]      (zero outside r's utterance)
//   y[r][co] = epi(D[co][r]),  epi = ReLU -> per-channel affine (folded eval BN)   or ReLU   or tanh   or none
//
// MFMA operand roles: A = weight panel (M = out channels), B = activation rows (N = frames), so that each lane's
// accumulator registers run over 4 consecutive CHANNELS of one frame -> vector stores into channels-last rows.
// Both operands are K-contiguous in LDS as rows of 8 x 16-byte chunks (64 bf16 or 32 f32 of K per tile), chunk
// index XOR-swizzled by (row >> 1) & 7 so ds_write_b128 and the ds_read_b128 fragment reads are conflict-free.
// One k-group = one 16-byte chunk per operand per lane: bf16 -> one v_mfma_f32_32x32x16_bf16; f32 -> four
// v_mfma_f32_32x32x2_f32 (exact f32).  Global -> register -> LDS staging (per-row utterance masks are applied in
// registers), two LDS buffers, one barrier per K tile.
#include "common.h"

namespace sugar {

struct ConvArgs {
  const void* x;
  const void* x2;
  const void* w;
  void* y;
  void* y2;
  const float* scale;
  const float* shift;
  const int2* rowseg;
  int ldx, ldx2, ldy, ldy2, y2_from;
  int R, cout, cin_p, ktaps, dil, k_pad, flags;
  int n_co_tiles, n_row_tiles;
  double* stat;   // STATS: per-(row tile, channel) sum and sum of squares of the stored outputs (bn_recal.hip's partials)
  int stat_c;     //        channels of that table
};

template <typename T>
struct Mma;
template <>
struct Mma<bf16> {
  typedef bf16x8 Frag;
  __device__ __forceinline__ static void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  typedef f32x4 Frag;
  __device__ __forceinline__ static void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
  }
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T>
__device__ __forceinline__ u32x4 add_chunks(u32x4 a, u32x4 b);
template <>
__device__ __forceinline__ u32x4 add_chunks<float>(u32x4 a, u32x4 b) {
  f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
  return __builtin_bit_cast(u32x4, fa + fb);
}
template <>
__device__ __forceinline__ u32x4 add_chunks<bf16>(u32x4 a, u32x4 b) {
  bf16x8 ha = __builtin_bit_cast(bf16x8, a), hb = __builtin_bit_cast(bf16x8, b), r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (bf16)((float)ha[i] + (float)hb[i]);
  return __builtin_bit_cast(u32x4, r);
}

// Bijective XCD-aware remap (blocks b and b+8 share an XCD's L2): consecutive virtual ids stay on one XCD, so
// the co-tiles that re-read one activation row tile hit that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, seq = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
}

// Fixed-order reduce-scatter over the 32 lanes of one half wave (lanes that share `lane >> 5`): every lane brings 16
// values; afterwards lane l holds, for value index (l >> 1) & 15, the sum over the 32 lanes (both lanes of a pair hold
// the same number).  16 cross-lane moves instead of 16 x 5 for one butterfly per value.
template <int HALF>
__device__ __forceinline__ void reduce_scatter_step(float (&v)[16], int l) {
  constexpr int MASK = 2 * HALF;                     // lane bit that picks the half a lane keeps
  const bool up = l & MASK;
#pragma unroll
  for (int k = 0; k < HALF; ++k) {
    const float keep = up ? v[k + HALF] : v[k], send = up ? v[k] : v[k + HALF];
    v[k] = keep + __shfl_xor(send, MASK);
  }
}
__device__ __forceinline__ float reduce_scatter32(float (&v)[16], int l) {
  reduce_scatter_step<8>(v, l);                      // (explicit steps: as one loop over `half` the array indices were
  reduce_scatter_step<4>(v, l);                      //  not compile-time constants and became 900 selects)
  reduce_scatter_step<2>(v, l);
  reduce_scatter_step<1>(v, l);
  return v[0] + __shfl_xor(v[0], 1);
}

// STATS (BN recalibration, bn_recal.hip): the epilogue also leaves, per row tile and output channel, the sum and the sum
// of squares of the values it STORES (after the rounding to T) in a.stat -- the partial table bn_finalize_kernel adds
// up -- so that train-mode BatchNorm needs no pass of its own over the tensor for its statistics.
template <typename T, int BCO, int BR, bool STATS = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvArgs a) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int BKE = 8 * EPC;          // K elements per tile
  constexpr int WROWS = BCO / 32;       // 16-byte chunks each thread stages per tile, weights
  constexpr int XROWS = BR / 32;        //                                           , activations
  constexpr int MT = BCO / 64;          // 32x32 MFMA tiles per wave along channels
  constexpr int NT = BR / 64;           //                           along frames
  constexpr int TILE_BYTES = (BCO + BR) * 128;
  typedef typename Mma<T>::Frag Frag;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int wco = wv & 1, wr = wv >> 1;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int co_tile = vid % a.n_co_tiles, row_tile = vid / a.n_co_tiles;
  const int co_base = co_tile * BCO;
  const int r_base = row_tile * BR;

  const int lc = tid & 7;    // chunk column this thread stages
  const int lr = tid >> 3;   // first row this thread stages (then +32, +64, ...)

  const T* __restrict__ x = (const T*)a.x;
  const T* __restrict__ x2 = (const T*)a.x2;
  const T* __restrict__ w = (const T*)a.w;
  const int ctr = (a.ktaps - 1) >> 1;

  int2 seg[XROWS];
#pragma unroll
  for (int i = 0; i < XROWS; ++i) {
    const int r = r_base + lr + 32 * i;
    seg[i] = (r < a.R) ? a.rowseg[r] : make_int2(0, 0);
  }

  u32x4 wreg[WROWS], xreg[XROWS];

  auto load_tile = [&](int kt) {
    const int ke = (kt * 8 + lc) * EPC;
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const T* p = w + (size_t)(co_base + lr + 32 * i) * a.k_pad + ke;
      wreg[i] = *(const u32x4*)p;
    }
    const int tap = ke / a.cin_p;
    const int ci = ke - tap * a.cin_p;
    const int roff = (tap - ctr) * a.dil;
    const bool kvalid = tap < a.ktaps;
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
      const int src = r_base + lr + 32 * i + roff;
      // unconditional loads from a clamped (always valid) row, then select: a branch around each load makes hipcc
      // wait vmcnt(0) per element and serialises the tile's memory round trips
      const bool ok = kvalid && src >= seg[i].x && src < seg[i].y;
      const int srcc = max(min(src, seg[i].y - 1), seg[i].x);
      u32x4 v = *(const u32x4*)(x + (size_t)srcc * a.ldx + ci);
      if (x2) v = add_chunks<T>(v, *(const u32x4*)(x2 + (size_t)srcc * a.ldx2 + ci));
      xreg[i] = ok ? v : (u32x4){0u, 0u, 0u, 0u};
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* base = smem + buf * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < WROWS; ++i) *(u32x4*)(base + lds_off(lr + 32 * i, lc)) = wreg[i];
    unsigned char* xb = base + BCO * 128;
#pragma unroll
    for (int i = 0; i < XROWS; ++i) *(u32x4*)(xb + lds_off(lr + 32 * i, lc)) = xreg[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  const int nk = a.k_pad / BKE;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) load_tile(kt + 1);
    const unsigned char* wb = smem + (kt & 1) * TILE_BYTES;
    const unsigned char* xb = wb + BCO * 128;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      Frag af[MT], bf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *(const Frag*)(wb + lds_off(wco * (BCO / 2) + m * 32 + frow, 2 * g + fh));
#pragma unroll
      for (int n = 0; n < NT; ++n) bf[n] = *(const Frag*)(xb + lds_off(wr * (BR / 2) + n * 32 + frow, 2 * g + fh));
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) Mma<T>::mma(acc[m][n], af[m], bf[n]);
    }
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane (frow, fh) owns frame r, channels co0 + 8q + 4fh + {0..3}, q = 0..3 ----
  T* __restrict__ y = (T*)a.y;
  T* __restrict__ y2 = (T*)a.y2;
  const bool relu = a.flags & EPI_RELU, affine = a.flags & EPI_AFFINE, tanh_ = a.flags & EPI_TANH;
  if constexpr (STATS) {
    // Statistics first, straight from the accumulators and branch-free (inside the store loop below the compiler turned
    // every update into a select per guarding branch: 900 v_cndmask, 20.7 us against 13.1 for the 64-channel Res2 step).
    // Rows past R were loaded as zeros and stay 0 through ReLU, so they need no mask; channels past cout are never
    // written out.  Lanes of one half wave (fh) hold 32 frames of the same 16 channels per MFMA tile: reduce-scatter
    // over them; the two row halves of the workgroup (wr) are added through LDS (free since the barrier that closed
    // the K loop).
    float* red = (float*)smem;                       // [wr][sum | sum of squares][BCO]
    const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float ssum[16], ssq[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float t = 0.f, t2 = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float x = to_f32(from_f32<T>(fmaxf(acc[m][n][e], lo)));      // the value the loop below stores
          t += x;
          t2 = fmaf(x, x, t2);
        }
        ssum[e] = t;
        ssq[e] = t2;
      }
      const float s1 = reduce_scatter32(ssum, frow), s2 = reduce_scatter32(ssq, frow);
      const int e = (frow >> 1) & 15;
      const int ch = wco * (BCO / 2) + m * 32 + 8 * (e >> 2) + 4 * fh + (e & 3);
      if (!(frow & 1)) {
        red[(wr * 2 + 0) * BCO + ch] = s1;
        red[(wr * 2 + 1) * BCO + ch] = s2;
      }
    }
    __syncthreads();
    if (tid < BCO && co_base + tid < a.cout) {
      double* p = a.stat + (size_t)row_tile * 2 * a.stat_c + co_base + tid;
      p[0] = (double)red[0 * BCO + tid] + (double)red[2 * BCO + tid];
      p[a.stat_c] = (double)red[1 * BCO + tid] + (double)red[3 * BCO + tid];
    }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int r = r_base + wr * (BR / 2) + n * 32 + frow;
    if (r >= a.R) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co_base + wco * (BCO / 2) + m * 32 + 8 * q + 4 * fh;
        if (co >= a.cout) continue;
        f32x4 v = {acc[m][n][4 * q + 0], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu_f32(v[e]);
        }
        if (affine) {
          const f32x4 sc = *(const f32x4*)(a.scale + co), sh = *(const f32x4*)(a.shift + co);
          v = v * sc + sh;
        }
        if (tanh_) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = sizeof(T) == 2 ? tanh_bf16_epilogue(v[e]) : tanhf(v[e]);   // (bf16: as conv1x1.hip)
        }
        if constexpr (DT<T>::id == SUGAR_F32) {
          *(f32x4*)(y + (size_t)r * a.ldy + co) = v;
          if (y2 && co >= a.y2_from) *(f32x4*)(y2 + (size_t)r * a.ldy2 + co) = v;
        } else {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)(y + (size_t)r * a.ldy + co) = o;
          if (y2 && co >= a.y2_from) *(bf16x4*)(y2 + (size_t)r * a.ldy2 + co) = o;
        }
      }
    }
  }
}

constexpr int CONV_BR = 128;      // frames per tile of every instantiation below = rows per block of the STATS table

template <typename T, int BCO, int BR, bool STATS>
static int launch_conv_ts(const ConvArgs& a, hipStream_t s) {
  const size_t smem = 2 * (BCO + BR) * 128;
  static OptInMask opted{0};   // > 64 KiB never needed here, but keep the opt-in explicit, once per device
  if (int rc = lds_opt_in((const void*)conv_gemm_kernel<T, BCO, BR, STATS>, (int)smem, &opted)) return rc;
  const int grid = a.n_co_tiles * a.n_row_tiles;
  hipLaunchKernelGGL((conv_gemm_kernel<T, BCO, BR, STATS>), dim3(grid), dim3(256), smem, s, a);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

template <typename T, int BCO, int BR>
static int launch_conv_t(const ConvArgs& a0, hipStream_t s) {
  static_assert(BR == CONV_BR, "conv_stat_blocks assumes one tile height");
  ConvArgs a = a0;
  a.n_co_tiles = (int)ceil_div(a.cout, BCO);
  a.n_row_tiles = (int)ceil_div(a.R, BR);
  return a.stat ? launch_conv_ts<T, BCO, BR, true>(a, s) : launch_conv_ts<T, BCO, BR, false>(a, s);
}

int conv_stat_blocks(int64_t R) { return (int)ceil_div(R, CONV_BR); }

bool gemm1x1_eligible(const ConvLayer& L, const ConvIO& io, int dtype);
int launch_gemm1x1(const ConvLayer& L, const ConvIO& io, int64_t R, hipStream_t s, int64_t* stat_io);

int launch_conv(const ConvLayer& L, const ConvIO& io, const int2* d_rowseg, int64_t R, int dtype, hipStream_t s,
                int* stat_blocks) {
  int64_t stat_io = io.stat_cap;
  if (stat_blocks) *stat_blocks = 0;
  const int epc = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(R > 0 && R < (1ll << 31), "conv: bad row count %lld", (long long)R);
  SUGAR_CHECK_ARG(L.w && io.x && io.y && d_rowseg, "conv: null pointer");
  SUGAR_CHECK_ARG(((uintptr_t)io.x & 15) == 0 && io.ldx % epc == 0, "conv: activation pointer / ld must be 16-byte aligned");
  SUGAR_CHECK_ARG(!io.stat_part || (stat_blocks && !(L.flags & (EPI_TANH | EPI_AFFINE))),
                  "conv: statistics need a plain conv -> ReLU epilogue and a place for the block count");
  if (gemm1x1_eligible(L, io, dtype)) {                                     // wide 1x1 convs: LDS-DMA GEMMs
    const int rc = launch_gemm1x1(L, io, R, s, io.stat_part ? &stat_io : nullptr);
    if (stat_blocks && io.stat_part) *stat_blocks = (int)stat_io;
    return rc;
  }
  SUGAR_CHECK_ARG(L.cin_p % epc == 0 && io.ldx % epc == 0 && io.ldy % 4 == 0 && L.cout % 4 == 0,
                  "conv: channel counts must be multiples of the 16-byte chunk (cin_p %d ldx %d ldy %d cout %d)",
                  L.cin_p, io.ldx, io.ldy, L.cout);
  SUGAR_CHECK_ARG(((uintptr_t)io.x & 15) == 0 && ((uintptr_t)io.y & 7) == 0 && ((uintptr_t)io.x2 & 15) == 0,
                  "conv: activation pointers must be 16-byte aligned");
  SUGAR_CHECK_ARG(L.k_pad % (8 * epc) == 0 && L.k_pad >= L.ktaps * L.cin_p, "conv: bad k_pad %d", L.k_pad);
  SUGAR_CHECK_ARG(!(L.flags & EPI_AFFINE) || (L.scale && L.shift), "conv: affine epilogue without scale/shift");
  ConvArgs a;
  a.x = io.x; a.x2 = io.x2; a.w = L.w; a.y = io.y; a.y2 = io.y2;
  a.scale = L.scale; a.shift = L.shift; a.rowseg = d_rowseg;
  a.ldx = io.ldx; a.ldx2 = io.ldx2; a.ldy = io.ldy; a.ldy2 = io.ldy2; a.y2_from = io.y2_from;
  a.R = (int)R; a.cout = L.cout; a.cin_p = L.cin_p; a.ktaps = L.ktaps; a.dil = L.dil; a.k_pad = L.k_pad;
  a.flags = L.flags; a.n_co_tiles = 0; a.n_row_tiles = 0;
  // the statistics table of the generic kernel: one row block per 128-frame tile
  const bool stats = io.stat_part && io.stat_c == L.cout && (int64_t)conv_stat_blocks(R) * 2 * io.stat_c <= io.stat_cap;
  a.stat = stats ? io.stat_part : nullptr; a.stat_c = io.stat_c;
  if (stats) *stat_blocks = conv_stat_blocks(R);
  const bool small = L.n_pad <= 64;
  SUGAR_CHECK_ARG(L.n_pad % (small ? 64 : 128) == 0, "conv: n_pad %d not tile aligned", L.n_pad);
  if (dtype == SUGAR_F32) return small ? launch_conv_t<float, 64, 128>(a, s) : launch_conv_t<float, 128, 128>(a, s);
  return small ? launch_conv_t<bf16, 64, 128>(a, s) : launch_conv_t<bf16, 128, 128>(a, s);
}

// ---------------------------------------------------------------------------------- packing (host)
int pack_conv(ConvLayer* L, const float* h_w, const SugarBN* bn, int cout, int cin, int k, int dil, int flags,
              int cout_p, const int* out_map, int cin_p, const int* in_map, int dtype) {
  SUGAR_CHECK_ARG(h_w && cout > 0 && cin > 0 && (k == 1 || k == 3 || k == 5 || k == 7 || k == 9),
                  "pack_conv: bad shape cout %d cin %d k %d", cout, cin, k);
  const int epc = dtype == SUGAR_F32 ? 4 : 8;
  SUGAR_CHECK_ARG(cin_p % epc == 0 && cout_p % 4 == 0 && cin_p >= cin && cout_p >= cout, "pack_conv: bad padding");
  const int bke = 8 * epc;
  L->cout = cout_p;
  L->n_pad = (int)(cout_p <= 64 ? 64 : (cout_p <= 128 ? 128 : round_up(cout_p, 256)));
  L->cin_p = cin_p;
  L->ktaps = k;
  L->dil = dil;
  L->k_pad = (int)round_up((int64_t)k * cin_p, bke);
  L->flags = flags & ~EPI_AFFINE;
  const size_t n = (size_t)L->n_pad * L->k_pad;
  const size_t esz = dtype == SUGAR_F32 ? 4 : 2;
  void* host = calloc(n, esz);
  if (!host) { set_error("pack_conv: host alloc"); return SUGAR_ENOMEM; }
  for (int o = 0; o < cout; ++o) {
    const int po = out_map ? out_map[o] : o;
    for (int i = 0; i < cin; ++i) {
      const int pi = in_map ? in_map[i] : i;
      for (int j = 0; j < k; ++j) {
        const float v = h_w[((size_t)o * cin + i) * k + j];
        const size_t d = (size_t)po * L->k_pad + (size_t)j * cin_p + pi;
        if (dtype == SUGAR_F32) ((float*)host)[d] = v; else ((uint16_t*)host)[d] = host_f32_to_bf16(v);
      }
    }
  }
  int rc = dev_alloc(&L->w, n * esz);
  if (!rc) rc = dev_upload(L->w, host, n * esz);
  // gemm_wreg_kernel<KC, CM, SK, NW> (conv1x1.hip) wants the weights once more in its fragment-load order: wave wv of
  // channel slab `slab` loads fragment (c, m) as one contiguous KiB, lane fq * 16 + fr holding
  // W[slab * 16 CM NW + wv * 16 CM + wreg_row(m, fr)][c * 32 + fq * 8 .. + 7] (CM = 2; wreg_row interleaves the two MFMA
  // tiles of a wave so that a lane's 8 accumulators of a frame are 8 consecutive channels).  Shapes that kernel serves:
  //   K = 512, 256-channel slabs (CM 2, NW 8): first / last convs of the blocks;  K = 1536 -> 128 (CM 2, NW 4): pooling.linear1;
  //   K = 448 (80 mel x 5 taps, padded), 256-channel slabs (CM 2, NW 8): layer1 as a GEMM over overlapping rows
  int f_cm = 0, f_nw = 8;
  if (dtype == SUGAR_BF16 && k == 1 && L->k_pad == cin_p) {
    if (cin_p == 512 && L->n_pad % 256 == 0) f_cm = 2;
    else if (cin_p == 1536 && cout_p == 128 && L->n_pad == 128) f_cm = 2, f_nw = 4;
  } else if (dtype == SUGAR_BF16 && dil == 1 && L->k_pad == 448 && L->n_pad % 256 == 0) {
    f_cm = 2;
  }
  if (!rc && f_cm) {
    uint16_t* frag = (uint16_t*)malloc(n * 2);
    if (!frag) { free(host); set_error("pack_conv: host alloc"); return SUGAR_ENOMEM; }
    const uint16_t* src = (const uint16_t*)host;
    const int slab_ch = 16 * f_cm * f_nw, kc2 = L->k_pad / 32;
    size_t d = 0;
    for (int slab = 0; slab < L->n_pad / slab_ch; ++slab)
      for (int wv = 0; wv < f_nw; ++wv)
        for (int c = 0; c < kc2; ++c)
          for (int m = 0; m < f_cm; ++m)
            for (int lane = 0; lane < 64; ++lane) {
              const int co = slab * slab_ch + wv * 16 * f_cm + wreg_row(m, lane & 15), kk = c * 32 + (lane >> 4) * 8;
              for (int q = 0; q < 8; ++q) frag[d++] = src[(size_t)co * L->k_pad + kk + q];
            }
    rc = dev_alloc(&L->wfrag, n * 2);
    if (!rc) rc = dev_upload(L->wfrag, frag, n * 2);
    free(frag);
  }
  free(host);
  if (rc) return rc;
  if (bn) {
    float* sc = (float*)calloc(2 * (size_t)L->n_pad, sizeof(float));
    if (!sc) { set_error("pack_conv: host alloc"); return SUGAR_ENOMEM; }
    float* sh = sc + L->n_pad;
    for (int o = 0; o < cout; ++o) {
      const int po = out_map ? out_map[o] : o;
      const double s = (double)bn->weight[o] / sqrt((double)bn->running_var[o] + 1e-5);
      sc[po] = (float)s;
      sh[po] = (float)((double)bn->bias[o] - (double)bn->running_mean[o] * s);
    }
    rc = dev_alloc(&L->scale, 2 * (size_t)L->n_pad * sizeof(float));
    if (!rc) rc = dev_upload(L->scale, sc, 2 * (size_t)L->n_pad * sizeof(float));
    free(sc);
    if (rc) return rc;
    L->shift = L->scale + L->n_pad;
    L->flags |= EPI_AFFINE;
  }
  return SUGAR_OK;
}

void free_conv(ConvLayer* L) {
  dev_free(L->w);
  dev_free(L->wfrag);
  dev_free(L->scale);
  *L = ConvLayer();
}

// ---------------------------------------------------------------------------------- rowseg
__global__ void build_rowseg_kernel(const int32_t* __restrict__ row_off, int2* __restrict__ rowseg) {
  const int b = blockIdx.y;
  const int s = row_off[b], e = row_off[b + 1];
  const int r = s + blockIdx.x * blockDim.x + threadIdx.x;
  if (r < e) rowseg[r] = make_int2(s, e);
}

int launch_build_rowseg(const int32_t* d_row_off, int B, int t_max, int2* d_rowseg, hipStream_t s) {
  SUGAR_CHECK_ARG(d_row_off && d_rowseg && B > 0 && t_max > 0, "rowseg: bad arguments");
  hipLaunchKernelGGL(build_rowseg_kernel, dim3((unsigned)ceil_div(t_max, 256), (unsigned)B), dim3(256), 0, s,
                     d_row_off, d_rowseg);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

// ---------------------------------------------------------------------------------- debug copy
template <typename T>
__global__ void to_f32_kernel(const T* __restrict__ x, int ldx, int64_t R, int C, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * C) return;
  const int64_t r = i / C;
  const int c = (int)(i - r * C);
  out[i] = to_f32(x[r * ldx + c]);
}

int launch_to_f32(const void* d_x, int ldx, int64_t R, int C, float* d_out, int dtype, hipStream_t s) {
  const int64_t n = R * C;
  const unsigned grid = (unsigned)ceil_div(n, 256);
  if (dtype == SUGAR_F32)
    hipLaunchKernelGGL(to_f32_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)d_x, ldx, R, C, d_out);
  else
    hipLaunchKernelGGL(to_f32_kernel<bf16>, dim3(grid), dim3(256), 0, s, (const bf16*)d_x, ldx, R, C, d_out);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
