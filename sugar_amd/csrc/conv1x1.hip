// 1x1 conv as an MFMA GEMM with direct global->LDS staging (gfx950, bf16 operands / f32 accumulate).
//
// The 512- and 1536-channel 1x1 convs of EfficientTDNN (first_conv / last_conv of every SE-Res2 block,
// sugar/models/dynamictdnn.py:392-394,402-404, and catconv :633,663) carry 82 % of the MACs; pooling.linear1
// (sugar/models/dynamictdnn.py:533, 1536 -> 128 + tanh) reads the whole catconv output once.  They have no taps and
// no utterance boundaries, so the activation operand needs no per-row masks and can be staged by LDS-DMA
// (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
//
//   tile      TM out channels (M, MFMA A operand = weight panel) x TN frames (N, B operand), BK = 64
//             <256, 256, 2 stages> + split tail for catconv, <128, 128, 2> for layer1 and the narrower 1x1 convs,
//             <128, 64, 3 stages> for pooling.linear1; the 512 -> 512 convs of large batches run on gemm_wreg_kernel
//             (weights in registers, activations through one cross-tile ring; second half of this file)
//   waves     8 = 2 (channels) x 4 (frames); each owns TM/2 x TN/4 outputs in tiles of v_mfma_f32_16x16x32_bf16
//   LDS       NS stages x (TM + TN) rows of 128 bytes; rows of 8 x 16-byte chunks, chunk XOR (row>>1)&7 on the
//             per-lane SOURCE address (the DMA writes LDS lane-linearly) and the same XOR on the fragment reads
//   pipeline  ring of NS stages: wait for this wave's share of k-tile t (counted vmcnt), barrier, issue the DMA of
//             k-tile t+NS-1 into the buffer k-tile t-1 was read from, compute k-tile t
//   epilogue  ReLU -> folded-BN affine (-> tanh) in registers, transpose through LDS, 16-byte-per-lane row-contiguous stores
//
// 256-wide tiles make the wide kernel MFMA-bound rather than L2-bound (128 FLOP per staged byte); co-tiles that share
// an activation row tile are placed on one XCD (xcd_remap) so the re-read hits that XCD's L2.  The pooling.linear1
// shape is HBM-bound (237 MB of activations for 30 GFLOP): small tiles, two workgroups per CU and a deeper ring keep
// ~100 KB per CU in flight and let the hardware dispatcher balance the 1204 tiles.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace sugar {

namespace {

struct Gemm1x1Args {
  const bf16* x;
  const bf16* w;
  bf16* y;
  bf16* y2;
  const float* scale;
  const float* shift;
  int ldx, ldw, ldy, ldy2, y2_from;
  int R, cout, K, flags;
  int n_co_tiles;
  // tap convs as a GEMM over overlapping rows (layer1): the activation buffer holds every utterance with `xmap / 2`
  // zero rows before and after it, so the K vector of output row r (taps t-h..t+h, channels-last) is the contiguous
  // run starting at padded row r + xmap * b(r), b = utterance of r (found from row_off).  xmap = 0: plain rows.
  const int32_t* row_off = nullptr;
  int B = 0, xmap = 0;
  const bf16* wfrag = nullptr;   // gemm_wreg_kernel: the weights once more, in its fragment-load order (or nullptr)
  // gemm_wreg_kernel over the PADDED row space of a tap conv (layer1): GEMM row q reads its K vector at x + q * ldx and
  // stores to output row orow[q] (-1: a row between two utterances, nothing to store).  nullptr: output row = q.
  const int32_t* orow = nullptr;
  // gemm_wreg_kernel<..., STATS> (BN recalibration): [P frame ranges][sum | sum of squares][stat_c] doubles of the stored outputs
  double* stat = nullptr;
  int stat_c = 0;
};

constexpr int TK = 64;

// Diagnostic build only (tools/gemm_variants.hip with -DGEMM_STAMPS): shader-clock stamps of waves 0 and 4 of one
// workgroup around the phases of the ping-pong loop; the library build compiles none of this.
#ifdef GEMM_STAMPS
__device__ long long* g_gemm_stamps;
#define GEMM_STAMP()                                                                                    \
  do {                                                                                                  \
    if (stamp_on && kt >= 4 && kt < 7) { stamp_buf[stamp_n] = (long long)__builtin_amdgcn_s_memtime(); ++stamp_n; } \
  } while (0)
#else
#define GEMM_STAMP() do {} while (0)
#endif

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, seq = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
}

// One TM x TN output tile at (co_base, r_base), by the whole 512-thread workgroup.
template <int TM, int TN, int NS, int PP = 0>
__device__ __forceinline__ void gemm_tile(const Gemm1x1Args& a, const int co_base, const int r_base) {
  constexpr int STAGE_BYTES = (TM + TN) * 128;
  constexpr int MI = TM / 32;                    // 16-channel MFMA tiles per wave
  constexpr int NJ = TN / 64;                    // 16-frame MFMA tiles per wave
  constexpr int CW = TM / 2;                     // channels per wave
  constexpr int FW = TN / 4;                     // frames per wave
  constexpr int WI = TM / 64;                    // weight-tile DMA instructions per wave (8 rows each)
  constexpr int XI = TN / 64;                    // activation-tile DMA instructions per wave
  constexpr int EPI_STRIDE = CW * 2 + 16;        // bytes per staged output row
  constexpr int HN = NJ >= 2 ? 2 : 1;            // frame tiles per epilogue pass
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;

  // ---- DMA addressing: each wave fills its rows of both operand tiles, 8 rows (1 KiB) per instruction ----------
  const int srow = lane >> 3;                       // row within the 8-row slab
  const bf16* wsrc[WI];
  const bf16* xsrc[XI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = wv * (WI * 8) + i * 8 + srow;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);          // source chunk that belongs at LDS position lane&7
    wsrc[i] = a.w + (size_t)(co_base + row) * a.ldw + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = wv * (XI * 8) + i * 8 + srow;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int xr = min(r_base + row, a.R - 1);                      // rows past R: any valid row, masked at the store
    if (a.xmap) {                                             // utterance index: last b with row_off[b] <= xr
      int lo = 0, hi = a.B;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.row_off[mid] <= xr) lo = mid; else hi = mid;
      }
      xr += a.xmap * lo;
    }
    xsrc[i] = a.x + (size_t)xr * a.ldx + chunk * 8;
  }
  auto stage = [&](int buf, int kt) {
    unsigned char* wbase = smem + buf * STAGE_BYTES + wv * (WI * 8) * 128;
    unsigned char* xbase = smem + buf * STAGE_BYTES + TM * 128 + wv * (XI * 8) * 128;
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * TK),
                                       (__attribute__((address_space(3))) void*)(wbase + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < XI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + kt * TK),
                                       (__attribute__((address_space(3))) void*)(xbase + i * 1024), 16, 0, 0);
  };

  auto dma_one = [&](int buf, int kt, int i) {        // instruction i of stage(): 0..WI-1 weights, then activations
    unsigned char* wbase = smem + buf * STAGE_BYTES + wv * (WI * 8) * 128;
    unsigned char* xbase = smem + buf * STAGE_BYTES + TM * 128 + wv * (XI * 8) * 128;
    if (i < WI)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * TK),
                                       (__attribute__((address_space(3))) void*)(wbase + i * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i - WI] + kt * TK),
                                       (__attribute__((address_space(3))) void*)(xbase + (i - WI) * 1024), 16, 0, 0);
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int n = 0; n < NJ; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = a.K / TK;
  const bool relu = a.flags & EPI_RELU, affine = a.flags & EPI_AFFINE, tanh_ = a.flags & EPI_TANH;
  // folded-BN vectors of this lane's channels: with <= 4 channel tiles per wave they are fetched here, ahead of every
  // DMA (older operations never disturb the counted vmcnt waits below), instead of stalling the epilogue
  constexpr bool PRE = MI <= 4;
  f32x4 scp[PRE ? MI : 1], shp[PRE ? MI : 1];
  if (PRE) {
#pragma unroll
    for (int m = 0; m < MI; ++m) {
      scp[m] = (f32x4){1.f, 1.f, 1.f, 1.f};
      shp[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (affine) {
        scp[m] = *(const f32x4*)(a.scale + co_base + wm * CW + m * 16 + 4 * fq);
        shp[m] = *(const f32x4*)(a.shift + co_base + wm * CW + m * 16 + 4 * fq);
      }
    }
  }
  // NS-stage ring: k-tile t+D (D = NS-1) is issued right after the barrier that opens k-tile t, into the buffer
  // k-tile t-1 was read from
  constexpr int D = NS - 1, NDMA = WI + XI;
  if constexpr (PP) {
    // ---- ping-pong schedule (NS == 2) ---------------------------------------------------------------------------
    // The two waves of a SIMD (wm = 0 / 1: hardware places wave w on SIMD w % 4) alternate: while one reads its 12
    // fragments of a 32-deep half k-tile from LDS, the other issues the 32 MFMAs of the half it read one phase
    // earlier.  Every phase ends in a workgroup barrier; group 1 runs one phase behind group 0.  Phase numbering
    // p (group 0): 4 kt + 2 h = read (kt, h), + 1 = MFMA; group 1 does the same at p + 1.  Buffer (kt+1) % 2 was
    // last read in phase 4 kt - 1, so the DMA of k-tile kt + 1 is issued in each group's read phase of (kt, 0) and
    // drained (vmcnt) before the barrier that closes phase 4 kt + 3, after which group 0 reads it.
    static_assert(!PP || NS == 2, "ping-pong needs the two-stage ring");
#ifdef GEMM_STAMPS
    const bool stamp_on = g_gemm_stamps && lane == 0 && (wv == 0 || wv == 4) && blockIdx.x == 8;
    long long* stamp_buf = g_gemm_stamps + (wv ? 1 : 0) * 64;
    int stamp_n = 0;
#endif
    // PP == 1: the DMA of k-tile kt + 1 is issued at the top of each group's read phase of (kt, 0).
    // PP == 2: it is issued one instruction per MFMA row inside an MFMA phase, where the VMEM issue slots are idle
    //          (in-kernel stamps: 8 DMA instructions at the top of a read phase stretch it from ~370 to ~860 clocks,
    //          past the ~660 of the other group's MFMA phase): group 0 in its MFMA phase of (kt, 0) = phase 4 kt + 1,
    //          group 1 in its MFMA phase of (kt - 1, 1) = phase 4 kt, the first phase in which that buffer is free.
    static_assert(PP != 2 || WI + XI <= MI, "one DMA instruction per MFMA row");
    stage(0, 0);
    if (PP == 2 && wm && nk > 1) {
      stage(1, 1);                                                 // group 1's share of k-tile 1 has no earlier MFMA phase
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    if (wm) asm volatile("s_barrier" ::: "memory");               // group 1 idles through phase 0
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* wb = smem + (kt & 1) * STAGE_BYTES;
      const unsigned char* xb = wb + TM * 128;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[MI], bf[NJ];
        GEMM_STAMP();                                              // A: read phase opens
        if (PP == 1 && s == 0 && kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
#pragma unroll
        for (int m = 0; m < MI; ++m) af[m] = *(const bf16x8*)(wb + swz(wm * CW + m * 16 + fr, 4 * s + fq));
#pragma unroll
        for (int n = 0; n < NJ; ++n) bf[n] = *(const bf16x8*)(xb + swz(wn * FW + n * 16 + fr, 4 * s + fq));
        if (s == 1 && wm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GEMM_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GEMM_STAMP();                                              // B: fragments have arrived
#endif
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STAMP();                                              // C: MFMA phase opens
        const int tk = wm ? kt + 2 : kt + 1;                       // k-tile this wave stages during this MFMA phase
        const bool issue = PP == 2 && s == (wm ? 1 : 0) && tk < nk;
#pragma unroll
        for (int m = 0; m < MI; ++m) {
#pragma unroll
          for (int n = 0; n < NJ; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
          if (PP == 2) {
            if (m < NDMA && issue) dma_one(tk & 1, tk, m);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STAMP();                                              // D: MFMAs issued
        if (s == 1 && !wm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // group 1's last MFMA phase needs no closing barrier: nobody reads LDS after phase 4 nk - 1, so both groups
        // execute the same number of barriers and each goes to its epilogue straight away
        if (!(wm && s == 1 && kt + 1 == nk)) asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) stage(t, t);
  int cur = 0, nxt = D;
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's share of k-tile kt has landed (younger ones may stay in flight), then everyone's has
    if (D > 1 && kt + D - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + D < nk) stage(nxt, kt + D);
    const unsigned char* wb = smem + cur * STAGE_BYTES;
    const unsigned char* xb = wb + TM * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MI], bf[NJ];
#pragma unroll
      for (int m = 0; m < MI; ++m) af[m] = *(const bf16x8*)(wb + swz(wm * CW + m * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int n = 0; n < NJ; ++n) bf[n] = *(const bf16x8*)(xb + swz(wn * FW + n * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NJ; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage memory becomes the epilogue staging
  }

  // ---- epilogue --------------------------------------------------------------------------------------------
  // accumulator tile (m, n): lane (fr, fq) holds frame wn*FW + n*16 + fr, channels wm*CW + m*16 + 4 fq + {0..3}
  unsigned char* ep = smem + wv * (16 * HN * EPI_STRIDE);     // this wave's [16 HN frames][CW channels] staging
  constexpr int CPR = CW / 8, RPP = 64 / CPR;                 // 16-byte chunks per staged row, rows per read pass
#pragma unroll
  for (int pass = 0; pass < NJ / HN; ++pass) {
#pragma unroll
    for (int m = 0; m < MI; ++m) {
      const int col = m * 16 + 4 * fq;                        // channel within the wave's CW
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (PRE) {
        sc = scp[m];
        sh = shp[m];
      } else if (affine) {
        sc = *(const f32x4*)(a.scale + co_base + wm * CW + col);
        sh = *(const f32x4*)(a.shift + co_base + wm * CW + col);
      }
#pragma unroll
      for (int nn = 0; nn < HN; ++nn) {
        f32x4 v = acc[m][pass * HN + nn];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu_f32(v[e]);
        }
        v = v * sc + sh;
        if (tanh_) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tanh_bf16_epilogue(v[e]);
        }
        const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *(bf16x4*)(ep + (nn * 16 + fr) * EPI_STRIDE + col * 2) = o;
      }
    }
    // the staging region is private to the wave: no workgroup barrier, only the LDS counter
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16 * HN / RPP; ++i) {
      const int f = i * RPP + lane / CPR;                     // frame within the pass
      const int ch = lane % CPR;
      const int r = r_base + wn * FW + pass * (16 * HN) + f;
      const int co = co_base + wm * CW + ch * 8;
      const u32x4 v = *(const u32x4*)(ep + f * EPI_STRIDE + ch * 16);
      if (r < a.R && co < a.cout) {
        *(u32x4*)(a.y + (size_t)r * a.ldy + co) = v;
        if (a.y2 && co >= a.y2_from) *(u32x4*)(a.y2 + (size_t)r * a.ldy2 + co) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// Uniform grid: every workgroup one TM x TN tile.
template <int TM, int TN, int NS, int PP = 0>
__global__ __launch_bounds__(512, TM == 256 ? 2 : 4) void gemm1x1_kernel(Gemm1x1Args a) {
  const int vid = xcd_remap2(blockIdx.x, gridDim.x);
  gemm_tile<TM, TN, NS, PP>(a, (vid % a.n_co_tiles) * TM, (vid / a.n_co_tiles) * TN);
}

// Split-tail grid for the wide convs.  With one 256 x 256 tile per CU at a time the makespan is ceil(tiles / CUs)
// rounds, and the last round is often nearly empty (512-channel convs on 256 x 301 frames: 602 tiles = 2.35 rounds;
// catconv: 1806 = 7.05).  The first n_main tiles (whole rounds) run as 256 x 256; each tile of the remainder is cut
// along the frames into `split` (1, 2 or 4) sub-tiles with their own wave tiling, so the last round fills the chip
// with pieces that take 1/2 or 1/4 of a round.  Same outputs bit for bit: every output element still sums its K
// products in the same order.
template <int NS, int PP = 0>
__global__ __launch_bounds__(512, 2) void gemm1x1_split_kernel(Gemm1x1Args a, int n_main, int split) {
  const int bid = blockIdx.x;
  if (bid < n_main) {
    const int vid = xcd_remap2(bid, n_main);
    gemm_tile<256, 256, NS, PP>(a, (vid % a.n_co_tiles) * 256, (vid / a.n_co_tiles) * 256);
    return;
  }
  const int t = bid - n_main;
  const int tile = n_main + t / split, sub = t - (t / split) * split;
  const int co_base = (tile % a.n_co_tiles) * 256, r_tile = (tile / a.n_co_tiles) * 256;
  if (split == 4) gemm_tile<256, 64, NS, PP>(a, co_base, r_tile + sub * 64);
  else if (split == 2) gemm_tile<256, 128, NS, PP>(a, co_base, r_tile + sub * 128);
  else gemm_tile<256, 256, NS, PP>(a, co_base, r_tile);
}

// ---- persistent form of the split-tail grid (the wide convs: catconv, the 1024-channel convs) -----------------------
// One workgroup per CU walks its 256 x 256 tiles round by round (ping-pong schedule, same sums in the same order as
// gemm_tile<256, 256, 2, 1>: bit-identical) and then takes its sub-tile of the split tail.  What the loop buys over one
// workgroup per tile: (i) the first k-tile of the NEXT tile is DMA'd into stage 0 while the epilogue of the current one
// runs out of stage 1 + the 32 KiB the two stages leave free, so no tile but the first waits for a DMA round trip before
// its first MFMA; (ii) the epilogue's 16 stores per wave drain under the next tile's k-loop instead of holding the CU
// until the workgroup has exited (one workgroup fits a CU: the next one could not start before).
// vmcnt bookkeeping: vector memory operations of a wave return in issue order, and the prefetch is issued BEFORE the
// epilogue's stores, so "my share of the prefetch has landed" is vmcnt(<stores issued since>): 16 for a full tile
// without a duplicate store, otherwise the wave just waits for everything.
// SCHED 0 is the two-half ping-pong of gemm_tile (kept for A/B runs and the bit-identity tests).
// SCHED 1 / 2, the quadrant schedule (the shape of the 8-phase GEMM template of the CDNA programming guide): a k-tile is
// four phases of 16 MFMAs, one 64 x 32 quadrant of the wave's 128 x 64 outputs each, on operand sub-tiles that stay in
// registers (phase 1 reads b0 + a0, 2 reads b1, 3 reads a1, 4 reads nothing: a0 x b0, a0 x b1, a1 x b1, a1 x b0).
// SCHED 2, the default, leaves out the barriers between phases 1 / 2 and 3 / 4: two phases of 32 MFMAs, half the
// barriers (catconv 0.314 -> 0.298 ms).  All
// LDS reads of a k-tile are over after phase 3 (the B half after phase 2), so its stage can be re-filled from there on
// and the DMA runs up to 1.5 k-tiles ahead with TWO instructions per wave and phase instead of eight in one burst:
// activations of k-tile kt + 2 in phases 3 / 4, weights of kt + 1 in phases 1 / 2 (the weights are L2 hits, the shorter
// lead is theirs).  Every phase is [reads + DMA | barrier | 16 MFMAs | barrier]; group 1 (wm = 1) runs one barrier
// behind group 0, so one group's reads sit under the other's MFMAs.  Same sums in the same order: bit-identical.
template <int SCHED = 0>
__device__ __forceinline__ void gemm_tiles_persist(const Gemm1x1Args& a, const int xcd_first, const int stride, const int rounds) {
  constexpr int TM = 256, TN = 256, STAGE_BYTES = (TM + TN) * 128;
  constexpr int MI = TM / 32, NJ = TN / 64, CW = TM / 2, FW = TN / 4, WI = TM / 64, XI = TN / 64;
  constexpr int EPI_STRIDE = CW * 2 + 16, HN = 2;
  constexpr int CPR = CW / 8, RPP = 64 / CPR, NSTORE = (NJ / HN) * (16 * HN / RPP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;
  const int srow = lane >> 3, fr = lane & 15, fq = lane >> 4;
  const int nk = a.K / TK;
  const bool relu = a.flags & EPI_RELU, affine = a.flags & EPI_AFFINE, tanh_ = a.flags & EPI_TANH;

  // folded-BN vectors of ALL output channels in LDS behind the staging: a global load in the epilogue would be younger
  // than the prefetch in flight, and waiting for it (in-order returns) would expose the very round trip the prefetch hides
  float* bn = (float*)(smem + STAGE_BYTES + 8 * (16 * HN * EPI_STRIDE));
  if (affine) {
    for (int i = tid; i < a.cout; i += 512) {
      bn[i] = a.scale[i];
      bn[a.cout + i] = a.shift[i];
    }
  }
  // DMA sources: each wave its rows of both operands, 8 rows (1 KiB) per instruction.  Kept as 32-bit element offsets
  // from uniform bases (launch_split checks that they fit): 64-bit pointers per instruction cost 8 more VGPRs, the
  // kernel then spilled, and the reloads' vmcnt(0) waits sat between the DMA instructions of the prefetch.
  unsigned woff[WI], xoff[XI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = wv * (WI * 8) + i * 8 + srow;
    woff[i] = (unsigned)(row * a.ldw + ((lane & 7) ^ ((row >> 1) & 7)) * 8);   // source chunk that belongs at LDS position lane & 7
  }
  const bf16* wtile = a.w;                                     // weight rows of the tile being staged
  auto point_at = [&](int co_base, int r_base) {
    wtile = a.w + (size_t)co_base * a.ldw;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = wv * (XI * 8) + i * 8 + srow;
      const int xr = min(r_base + row, a.R - 1);               // rows past R: any valid row, masked at the store
      xoff[i] = (unsigned)xr * (unsigned)a.ldx + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
  };
  auto stage = [&](int buf, int kt) {
    unsigned char* wbase = smem + buf * STAGE_BYTES + wv * (WI * 8) * 128;
    unsigned char* xbase = smem + buf * STAGE_BYTES + TM * 128 + wv * (XI * 8) * 128;
    const bf16* wk = wtile + kt * TK;
    const bf16* xk = a.x + kt * TK;
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + woff[i]),
                                       (__attribute__((address_space(3))) void*)(wbase + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < XI; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xk + xoff[i]),
                                       (__attribute__((address_space(3))) void*)(xbase + i * 1024), 16, 0, 0);
  };
  auto dma_w = [&](int buf, int kt, int i0) {          // weight instructions i0, i0 + 1 of k-tile kt of the tile wtile points at
    unsigned char* wbase = smem + buf * STAGE_BYTES + wv * (WI * 8) * 128;
    const bf16* wk = wtile + kt * TK;
#pragma unroll
    for (int i = i0; i < i0 + 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + woff[i]),
                                       (__attribute__((address_space(3))) void*)(wbase + i * 1024), 16, 0, 0);
  };
  auto dma_x = [&](int buf, int kt, int i0, int n) {   // activation instructions i0 .. i0 + n - 1 (rows of xoff)
    unsigned char* xbase = smem + buf * STAGE_BYTES + TM * 128 + wv * (XI * 8) * 128;
    const bf16* xk = a.x + kt * TK;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (i >= i0 && i < i0 + n)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xk + xoff[i]),
                                         (__attribute__((address_space(3))) void*)(xbase + i * 1024), 16, 0, 0);
  };
  auto tile_at = [&](int r, int& co_base, int& r_base) {
    const int vid = xcd_first + r * stride;
    co_base = (vid % a.n_co_tiles) * TM;
    r_base = (vid / a.n_co_tiles) * TN;
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int n = 0; n < NJ; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int co_base, r_base;
  tile_at(0, co_base, r_base);
  point_at(co_base, r_base);
  stage(0, 0);
  bool counted = false;                                        // the wait below may leave NSTORE younger stores in flight
  for (int r = 0; r < rounds; ++r) {
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // k-tile 0 in stage 0; every wave out of its epilogue
    if (wm) asm volatile("s_barrier" ::: "memory");             // group 1 idles through phase 0
    const int co_prev = co_base, r_prev = r_base;                // this round's tile
    if constexpr (SCHED >= 1) {
      constexpr bool TWO = SCHED == 2;      // two phases of 32 MFMAs per k-tile instead of four of 16 (half the barriers)
      const bool more = r + 1 < rounds;
      int co_next = co_base, r_next = r_base;
      if (more) tile_at(r + 1, co_next, r_next);
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned char* wb = smem + cur * STAGE_BYTES;
        const unsigned char* xb = wb + TM * 128;
        bf16x8 a_[4][2], b0[2][2], b1[2][2];
        auto read_a = [&](int m0) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 2; ++k) a_[m][k] = *(const bf16x8*)(wb + swz(wm * CW + (m0 + m) * 16 + fr, 4 * k + fq));
        };
        auto read_b = [&](bf16x8 (&b)[2][2], int n0) {
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int k = 0; k < 2; ++k) b[n][k] = *(const bf16x8*)(xb + swz(wn * FW + (n0 + n) * 16 + fr, 4 * k + fq));
        };
        auto quad = [&](int m0, bf16x8 (&b)[2][2], int n0) {     // per accumulator: k = 0 then k = 1, as the half phases did
          __builtin_amdgcn_s_setprio(1);                       // (keeps hipcc from moving MFMAs across the barriers; 0.8 %)
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int n = 0; n < 2; ++n)
                acc[m0 + m][n0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[m][k], b[n][k], acc[m0 + m][n0 + n], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        };
        auto close_read = [&]() {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        };
        auto close_mfma = [&](bool last) {
          __builtin_amdgcn_sched_barrier(0);
          if (!last) asm volatile("s_barrier" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        };
        // ---- phase 1: weights of the next k-tile (the next TILE's first one behind the last k-tile) ----------------
        if (kt + 1 < nk) dma_w(cur ^ 1, kt + 1, 0);
        else if (more) { wtile = a.w + (size_t)co_next * a.ldw; dma_w(0, 0, 0); }
        if (kt == 0 && nk > 1) dma_x(1, 1, 0, XI);              // (its activations could not be staged under the epilogue)
        read_b(b0, 0);
        read_a(0);
        if (!TWO) {
          close_read();
          quad(0, b0, 0);
          close_mfma(false);
        }
        // ---- phase 2 -------------------------------------------------------------------------------------------------
        if (kt + 1 < nk) dma_w(cur ^ 1, kt + 1, 2);
        else if (more) dma_w(0, 0, 2);
        read_b(b1, 2);
        close_read();
        if (TWO) quad(0, b0, 0);
        quad(0, b1, 2);
        close_mfma(false);
        // ---- phase 3: activations two k-tiles ahead, into the half of THIS stage that nobody reads any more -------------
        const bool x_cur = kt + 2 < nk, x_next = kt + 2 == nk && more;
        if (x_next) {                                           // the stream moves on to the next tile's rows
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            const int row = wv * (XI * 8) + i * 8 + srow;
            const int xr = min(r_next + row, a.R - 1);
            xoff[i] = (unsigned)xr * (unsigned)a.ldx + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 8);
          }
        }
        if (x_cur) dma_x(cur, kt + 2, 0, 2);
        else if (x_next) dma_x(cur, 0, 0, 2);
        read_a(4);
        if (!TWO) {
          close_read();
          quad(4, b1, 2);
          close_mfma(false);
        }
        // ---- phase 4 -------------------------------------------------------------------------------------------------
        if (x_cur) dma_x(cur, kt + 2, 2, 2);
        else if (x_next) dma_x(cur, 0, 2, 2);
        // k-tile kt + 1 has landed (this wave's share) once only the activations issued in phases 3 / 4 are outstanding
        if (kt + 1 < nk) {
          if (x_cur || x_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XI) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        close_read();
        if (TWO) quad(4, b1, 2);
        quad(4, b0, 0);
        close_mfma(wm && kt + 1 == nk);
      }
      co_base = co_next;
      r_base = r_next;
    } else {
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* wb = smem + (kt & 1) * STAGE_BYTES;
      const unsigned char* xb = wb + TM * 128;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[MI], bf[NJ];
        if (s == 0 && kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
#pragma unroll
        for (int m = 0; m < MI; ++m) af[m] = *(const bf16x8*)(wb + swz(wm * CW + m * 16 + fr, 4 * s + fq));
#pragma unroll
        for (int n = 0; n < NJ; ++n) bf[n] = *(const bf16x8*)(xb + swz(wn * FW + n * 16 + fr, 4 * s + fq));
        if (s == 1 && wm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
          for (int n = 0; n < NJ; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s == 1 && !wm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(wm && s == 1 && kt + 1 == nk)) asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    // every wave is past the last LDS read of stage 0 (k-tile nk - 2; nk is even): the next tile's first k-tile goes there
    // (SCHED 1 has staged it during the last two k-tiles already and moved co_base / r_base on)
    const int co_cur = SCHED >= 1 ? co_prev : co_base, r_cur = SCHED >= 1 ? r_prev : r_base;
    if (SCHED == 0 && r + 1 < rounds) {
      tile_at(r + 1, co_base, r_base);
      point_at(co_base, r_base);
      stage(0, 0);
    }
    counted = r_cur + TN <= a.R && co_cur + TM <= a.cout && !a.y2;   // exactly NSTORE stores follow, none masked off
    // ---- epilogue out of stage 1 + the free tail of LDS (gemm_tile's, with the staging moved) -------------------
    unsigned char* ep = smem + STAGE_BYTES + wv * (16 * HN * EPI_STRIDE);
#pragma unroll
    for (int pass = 0; pass < NJ / HN; ++pass) {
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int col = m * 16 + 4 * fq;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (affine) {
          sc = *(const f32x4*)(bn + co_cur + wm * CW + col);
          sh = *(const f32x4*)(bn + a.cout + co_cur + wm * CW + col);
        }
#pragma unroll
        for (int nn = 0; nn < HN; ++nn) {
          f32x4 v = acc[m][pass * HN + nn];
          acc[m][pass * HN + nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu_f32(v[e]);
          }
          v = v * sc + sh;
          if (tanh_) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanh_bf16_epilogue(v[e]);
          }
          const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4*)(ep + (nn * 16 + fr) * EPI_STRIDE + col * 2) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 16 * HN / RPP; ++i) {
        const int f = i * RPP + lane / CPR;
        const int ch = lane % CPR;
        const int rr = r_cur + wn * FW + pass * (16 * HN) + f;
        const int co = co_cur + wm * CW + ch * 8;
        const u32x4 v = *(const u32x4*)(ep + f * EPI_STRIDE + ch * 16);
        if (rr < a.R && co < a.cout) {
          *(u32x4*)(a.y + (size_t)rr * a.ldy + co) = v;
          if (a.y2 && co >= a.y2_from) *(u32x4*)(a.y2 + (size_t)rr * a.ldy2 + co) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // staging free before a tail sub-tile reuses the stages
}

// grid = the CU count (a multiple of 8); n_main = rounds * grid tiles of 256 x 256, then `tail` tiles cut into `split`
template <int SCHED>
__global__ __launch_bounds__(512, 2) void gemm1x1_persist_kernel(Gemm1x1Args a, int n_main, int split, int tail) {
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int xcd = bid & 7, seq = bid >> 3, wpx = nwg >> 3, rounds = n_main / nwg;
  // XCD x owns the contiguous tile range [x rounds wpx, (x + 1) rounds wpx); in round r its workgroups take consecutive
  // tiles of it, so co-tiles that share an activation row tile run at the same time on one XCD (shared L2)
  gemm_tiles_persist<SCHED>(a, xcd * rounds * wpx + seq, wpx, rounds);
  if (bid >= tail * split) return;
  const int tile = n_main + bid / split, sub = bid - (bid / split) * split;
  const int co_base = (tile % a.n_co_tiles) * 256, r_tile = (tile / a.n_co_tiles) * 256;
  if (split == 4) gemm_tile<256, 64, 2, 1>(a, co_base, r_tile + sub * 64);
  else if (split == 2) gemm_tile<256, 128, 2, 1>(a, co_base, r_tile + sub * 128);
  else gemm_tile<256, 256, 2, 1>(a, co_base, r_tile);
}

template <int TM, int TN, int NS, int PP = 0>
int launch_variant(const Gemm1x1Args& a, hipStream_t s) {
  constexpr int smem = NS * (TM + TN) * 128;
  static OptInMask opted{0};
  if (int rc = lds_opt_in((const void*)gemm1x1_kernel<TM, TN, NS, PP>, smem, &opted)) return rc;
  const int grid = a.n_co_tiles * (int)ceil_div(a.R, TN);
  hipLaunchKernelGGL((gemm1x1_kernel<TM, TN, NS, PP>), dim3(grid), dim3(512), smem, s, a);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

template <int PP = 0>
int launch_split(const Gemm1x1Args& a, hipStream_t s) {
  constexpr int smem = 2 * (256 + 256) * 128;
  static OptInMask opted{0};
  if (int rc = lds_opt_in((const void*)gemm1x1_split_kernel<2, PP>, smem, &opted)) return rc;
  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu)) return rc;
  const int total = a.n_co_tiles * (int)ceil_div(a.R, 256);
  const int n_main = total / n_cu * n_cu, tail = total - n_main;
  const int split = tail == 0 ? 1 : tail * 4 <= n_cu ? 4 : tail * 2 <= n_cu ? 2 : 1;
  if (PP == 1 && n_main >= n_cu && n_cu % 8 == 0 && (a.K / TK) % 2 == 0 && !a.xmap && a.cout <= 3072 && a.cout % 4 == 0 &&
      (int64_t)a.R * a.ldx < (1ll << 31) && (int64_t)256 * a.ldw < (1ll << 31) &&
      !tune_get(TUNE_GEMM_NO_PERSIST)) {
    // one round or more: one persistent workgroup per CU (stage 1 + 8 x 8 704 bytes of epilogue staging behind it); with
    // a single round nothing is prefetched across tiles, but the quadrant schedule of the loop is still the faster one
    // (64 x 3 s: catconv 0.0870 -> 0.0845 ms)
    constexpr int psmem = (256 + 256) * 128 + 8 * (16 * 2 * (128 * 2 + 16)) + 2 * 3072 * 4;   // + folded-BN vectors
    const int64_t sched = a.K / TK >= 4 ? tune_get(TUNE_GEMM_NO_QUAD) : 1;
    if (sched == 0) {                                           // quadrant schedule, two 32-MFMA phases per k-tile
      static OptInMask q2opted{0};
      if (int rc = lds_opt_in((const void*)gemm1x1_persist_kernel<2>, psmem, &q2opted)) return rc;
      hipLaunchKernelGGL(gemm1x1_persist_kernel<2>, dim3(n_cu), dim3(512), psmem, s, a, n_main, split, tail);
    } else if (sched == 2) {                                    // quadrant schedule, four 16-MFMA phases
      static OptInMask qopted{0};
      if (int rc = lds_opt_in((const void*)gemm1x1_persist_kernel<1>, psmem, &qopted)) return rc;
      hipLaunchKernelGGL(gemm1x1_persist_kernel<1>, dim3(n_cu), dim3(512), psmem, s, a, n_main, split, tail);
    } else {                                                    // two-half ping-pong
      static OptInMask popted{0};
      if (int rc = lds_opt_in((const void*)gemm1x1_persist_kernel<0>, psmem, &popted)) return rc;
      hipLaunchKernelGGL(gemm1x1_persist_kernel<0>, dim3(n_cu), dim3(512), psmem, s, a, n_main, split, tail);
    }
    SUGAR_LAUNCH_CHECK();
    return SUGAR_OK;
  }
  hipLaunchKernelGGL((gemm1x1_split_kernel<2, PP>), dim3(n_main + tail * split), dim3(512), smem, s, a, n_main, split);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

// ---- weights-in-registers GEMM for the K = 512 convs (first_conv / last_conv of the SE-Res2 blocks) ---------------
// With 8 k-tiles per output tile the LDS-DMA tile kernels above are latency-bound: one stage in flight per workgroup,
// every k-step waits a full DMA round trip (65 us for 77056 x 512 x 512, MFMA-bound would be 16).  Here the WEIGHTS
// never go through LDS: a workgroup owns a slab of 256 output channels, each of its waves keeps the [16 CM channels x
// K] weight fragments of its channels in registers (CM K / 8 VGPRs) for the whole launch and walks a contiguous range
// of 64-frame tiles.  Only the activations stream through LDS, as one continuous ring of stages of SK [64 frames x
// 64 k] sub-tiles (8 KiB each) that runs 12 sub-tiles (96 KiB per CU) ahead ACROSS tile boundaries, so after the first tile no DMA latency
// and no tile prologue is exposed, and the DMA bytes per MFMA are a quarter of the 128 x 128 tile's.  Every wave reads
// all 64 frames of a stage (8 B-operand fragments) for its own channels.
//   CM = 4: 4 waves, one per SIMD, 64 channels each (256 weight VGPRs of the 512 a lone wave may use)
//   CM = 2: 8 waves, two per SIMD, 32 channels each
// Software pipeline over 32-deep halves: while the MFMAs of one half issue, the fragments of the next are on their way
// from LDS.  One workgroup barrier per stage (SK = 2, 128-deep stages: 0.061 vs 0.062 ms, the barrier is ~250 clocks);
// the barrier that opens stage g + 1 sits before the last half of stage g: every wave has then read all of stage g
// (slot free for the DMA of stage g + 1 + D) and has its own rows of stage g + 1 in LDS.
// Same sums in the same order as gemm_tile: bit-identical outputs.
// Diagnostic builds only (tools/wreg_bench.hip with -DWREG_KNOCK=bits): leave out the MFMAs (1), the LDS fragment reads
// (2), the activation DMA (4), the output stores (8), all but one weight-fragment address (16), the per-stage workgroup
// barrier (32) or the counted vmcnt wait (64) to see what the kernel's time is made of; wrong results.
#ifndef WREG_KNOCK
#define WREG_KNOCK 0
#endif
#ifndef WREG_SPREAD
#define WREG_SPREAD 1
#endif
constexpr int WR_TN = 64, WR_SUB = WR_TN * 128;   // a [64 frames x 64 k] sub-tile: 8 KiB

// Round 2, measured and dropped (tools/wreg_bench, R = 77 056, all bit-identical; profiles/r02_wreg_variants.log): the
// SIMD partners (waves NW/2 ..) computing two stages (half a tile) behind the early waves so that the two waves of a
// SIMD reach their epilogues apart: 0.058 vs 0.0505 ms warm, 0.066 vs 0.0628 cold (the ring runs two stages shorter);
// the late waves issuing their DMA share after the first MFMA block instead of right behind the barrier: 0.0502 /
// 0.0610 standalone, no difference in the pipeline (first_conv + last_conv 0.334 ms per step either way).  Kept from
// that round: the weight fragments pre-packed in load order (a.wfrag): 0.0546 -> 0.0505 warm, 0.0661 -> 0.0628 cold.
// K = 64 KC; 16 CM channels per wave; SK sub-tiles (64 k each) per stage and barrier; NW waves = a slab of 16 CM NW channels
//   <8, 2, 4, 8>   512 -> 512 convs of the SE-Res2 blocks: 256-channel slabs, 128 weight VGPRs; 256-deep stages (two
//                  barriers per tile; 0.0467-0.0495 ms warm against 0.0496-0.0529 for 128-deep, 0.060 against 0.0634 cold)
//   <7, 2, 7, 8>   layer1 (80 mel x 5 taps = K 400 -> 448) over the padded row space, outputs through the row map a.orow;
//                  one stage = one tile (one barrier per tile: 45 us against 55 with 64-deep stages)
//   <24, 2, 4, 4, STAGED>  pooling.linear1 (1536 -> 128, tanh): one 128-channel slab, one wave per SIMD with 384 weight VGPRs
//                  (8 waves x 16 channels need 256 registers each and spill).  The 128 x 64 tile kernel
//                  re-streamed the 384 KiB weight panel from L2 for every 64-frame tile (2/3 of its DMA bytes) and ran at
//                  3.1 TB/s of activations; here only the activations move.
// STAGED: the epilogue re-orders a wave's [64 frames x 32 channels] through LDS so that 4 consecutive lanes store one
// frame's 64 contiguous bytes.  Straight from the accumulators consecutive lanes are consecutive FRAMES (16-byte
// pieces at the row pitch, four times the write requests); with two waves per SIMD that costs nothing measurable and
// frees 40 KiB of LDS for deeper stages (K = 512: SK 2 -> 4, layer1: SK 1 -> 7), a lone wave per SIMD
// (pooling.linear1) stalls on it: 74.5 vs 69 us, so that instantiation keeps the staging.
// STATS (BN recalibration): every lane also keeps the running sum and sum of squares of the 8 channels it stores, over
// all frames of the workgroup's range (16 VGPRs, ~100 VALU per tile); at the end the 16 lanes that share the channels are
// added up with four DPP steps and the range's (sum, sum of squares) go to a.stat -- the partial table
// bn_finalize_kernel reads (bn_recal.hip) -- so the BatchNorm behind the conv needs no statistics pass of its own.
template <int KC, int CM, int SK, int NW = 16 / CM, bool STAGED = false, bool STATS = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void gemm_wreg_kernel(Gemm1x1Args a, int n_half, int n_slabs, int P) {
  static_assert(KC % SK == 0, "whole stages per tile");
  static_assert(!(STATS && STAGED), "statistics come from the direct epilogue");
  static_assert(NW == 4 || NW == 8, "one or two waves per SIMD");
  constexpr int DI = 8 / NW, SLAB = 16 * CM * NW;      // DMA instructions per wave and sub-tile; channels per workgroup
  constexpr int WR_NS = 12 / SK + 1, WR_D = WR_NS - 1, WR_STAGE = SK * WR_SUB, KS = KC / SK, H = 2 * SK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // workgroups b and b + 8 (same XCD) take the two channel slabs of one frame range: the second read of x hits L2
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int slab = j % n_slabs, p = xcd + 8 * (j / n_slabs);
  // The frames are dealt out in HALF tiles of 32: with whole 64-frame tiles 2 x 1 204 units on 256 workgroups (256 x 3 s)
  // is 9.4 per workgroup and the launch lasts as long as the workgroups that got 10.  A range of an odd number of half
  // tiles ends in a tile of which only the first 32 frames are its own: that tile skips the MFMAs of frames 32 .. 63
  // (HALF below) and its epilogue stops at the range's last frame; the neighbour starts there with a tile of its own.
  // (pooling.linear1, NW = 4: 256 ranges of 4.7 tiles gain nothing from halves and measured 1.5 % slower cold: whole tiles)
  constexpr bool HALVES = NW == 8;
  const int n_unit = HALVES ? n_half : (n_half + 1) >> 1, hs = HALVES ? 1 : 2;
  const int h0 = hs * (int)((int64_t)p * n_unit / P), h1 = hs * (int)((int64_t)(p + 1) * n_unit / P);
  if (h0 >= h1) {
    if constexpr (STATS) {                            // an empty range still owns a row of the table
      const int c = slab * SLAB + wv * (CM * 16) + (lane & 31);
      if (c < a.cout) a.stat[((size_t)p * 2 + (lane >> 5)) * a.stat_c + c] = 0.0;
    }
    return;
  }
  const int f0 = h0 * (WR_TN / 2), q_end = min(h1 * (WR_TN / 2), a.R);   // first frame of the range, one past its last
  const int NTL = (h1 - h0 + 1) >> 1;                 // tiles of this workgroup, the last one HALF if the count is odd
  const bool odd = (h1 - h0) & 1;
  const int G = NTL * KS;                             // stages of this workgroup
  const int co_w = slab * SLAB + wv * (CM * 16);      // first channel of this wave

  // output-row map of the tiles in flight (a.orow): 4 slots of 64 int32 behind the ring, filled by wave 0 with one
  // 256-byte LDS-DMA per tile, issued with the tile's first stage (so it has landed, and every wave has passed a
  // barrier behind wave 0's wait for it, long before the tile's epilogue)
  constexpr int EPI_STRIDE = CM * 32 + 16, EPW = STAGED ? WR_TN * EPI_STRIDE : 0;
  int32_t* omap = (int32_t*)(smem + WR_NS * WR_STAGE + NW * EPW);

  // ---- activation DMA: wave wv fills rows 8 DI wv .. + 8 DI - 1 of a stage ----------------------------------------
  const int drow = wv * (8 * DI) + (lane >> 3);
  // sub-tiles s0 .. s1 - 1 of stage g of this workgroup -> slot g % WR_NS
  auto issue_part = [&](int g, int s0, int s1) {
    if (WREG_KNOCK & 4) return;
    const int slot = g % WR_NS, tf = f0 + (g / KS) * WR_TN, ks = g % KS;    // tf: first frame of the stage's tile
    if (s0 == 0 && a.orow && ks == 0 && wv == 0)      // the counted waits below only get stricter by this extra request
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.orow + min(tf + lane, a.R - 1)),
                                       (__attribute__((address_space(3))) void*)(omap + ((g / KS) & 3) * WR_TN), 4, 0, 0);
#pragma unroll
    for (int sub = 0; sub < SK; ++sub)
#pragma unroll
      for (int i = 0; i < DI; ++i) {
        if (sub < s0 || sub >= s1) continue;
        const int row = drow + 8 * i;
        const int dchunk = (lane & 7) ^ ((row >> 1) & 7);  // source chunk that belongs at LDS position lane & 7
        const int xr = min(tf + row, a.R - 1);             // rows past R: any valid row, masked at the store
        const bf16* src = a.x + (size_t)xr * a.ldx + dchunk * 8 + (ks * SK + sub) * TK;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(smem + slot * WR_STAGE + sub * WR_SUB + (wv * DI + i) * 1024), 16, 0, 0);
      }
  };
  auto issue = [&](int g) { issue_part(g, 0, SK); };
#pragma unroll
  for (int g = 0; g < WR_D; ++g)
    if (g < G) issue(g);                           // the ring fills while the weights load

  // ---- weight fragments: wf[c][m] = W[co_w + 16 m + fr][32 c + 8 fq .. + 7], c = 0 .. 2 KC - 1 -----------------
  // a.wfrag (packed at subnet creation): the same fragments in the order this loop reads them, so every load
  // instruction is one contiguous KiB instead of sixteen 64-byte pieces
  bf16x8 wf[2 * KC][CM];
  if (a.wfrag) {
    const bf16x8* wp = (const bf16x8*)a.wfrag + ((size_t)(slab * NW + wv) * (2 * KC) * CM) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 2 * KC; ++c)
#pragma unroll
      for (int m = 0; m < CM; ++m) wf[c][m] = wp[(c * CM + m) * 64];
  } else {
#pragma unroll
    for (int c = 0; c < 2 * KC; ++c)
#pragma unroll
      for (int m = 0; m < CM; ++m)
        wf[c][m] = *(const bf16x8*)(a.w + (size_t)(co_w + wreg_row(m, fr)) * a.ldw + ((WREG_KNOCK & 16) ? 0 : c * 32) + fq * 8);
  }
  const bool relu = a.flags & EPI_RELU, affine = a.flags & EPI_AFFINE, tanh_ = a.flags & EPI_TANH;
  // folded-BN vectors of this lane's channels, fetched once: a global load inside the tile loop would be younger than
  // the DMA stages in flight, and waiting for it (in-order returns) would drain the ring at every tile end
  f32x4 scp[CM], shp[CM];
#pragma unroll
  for (int m = 0; m < CM; ++m) {
    scp[m] = (f32x4){1.f, 1.f, 1.f, 1.f};
    shp[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (affine) {                                   // channels wreg_row(m, 4 fq .. 4 fq + 3)
      scp[m] = *(const f32x4*)(a.scale + co_w + 8 * fq + 4 * m);
      shp[m] = *(const f32x4*)(a.shift + co_w + 8 * fq + 4 * m);
    }
  }

  f32x4 acc[CM][4];
#pragma unroll
  for (int m = 0; m < CM; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};


  auto frags = [&](bf16x8 (&f)[4], int slot, int hh) {   // half hh of the stage: sub-tile hh / 2, 32-deep half hh % 2
    const unsigned char* xb = smem + slot * WR_STAGE + (hh >> 1) * WR_SUB;
    const int h = hh & 1;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (WREG_KNOCK & 2) asm volatile("" : "=v"(f[n]));
      else f[n] = *(const bf16x8*)(xb + swz(n * 16 + fr, 4 * h + fq));
    }
  };
  auto mfmas = [&](auto half_tag, int c, bf16x8 (&f)[4]) {
    constexpr int NN = decltype(half_tag)::value ? 2 : 4;      // a HALF tile: frames 0 .. 31 only
#pragma unroll
    for (int m = 0; m < CM; ++m)
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        if (WREG_KNOCK & 1) asm volatile("" : "+v"(acc[m][n]) : "v"(wf[c][m]), "v"(f[n]));
        else acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c][m], f[n], acc[m][n], 0, 0, 0);
      }
  };
  // tile t: lane (fr, fq) of acc[m][n] holds frame 16 n + fr, rows 4 fq .. 4 fq + 3 of MFMA tile m = (CM = 2, wreg_row)
  // channels 8 fq .. 8 fq + 7 of the wave's 32 over m = 0, 1: ReLU -> folded BN -> bf16 and ONE 16-byte store per frame,
  // 64 contiguous bytes per frame and wave (what the transposition through LDS used to build, without the LDS trip)
  static_assert(CM == 2, "the direct epilogue pairs two MFMA tiles per wave");
  unsigned char* ep = smem + WR_NS * WR_STAGE + wv * EPW;   // STAGED: this wave's [64 frames][32 channels] staging
  float ssum[STATS ? 8 : 1], ssq[STATS ? 8 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ssum[i] = ssq[i] = 0.f;
  }
  auto epilogue = [&](int lt) {                         // lt: tile of this workgroup, frames f0 + 64 lt ..
    const int32_t* om = omap + (lt & 3) * WR_TN;
    const int tf = f0 + lt * WR_TN;
    const int co = co_w + 8 * fq;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bf16x8 o;
#pragma unroll
      for (int m = 0; m < CM; ++m) {
        f32x4 v = acc[m][n];
        acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu_f32(v[e]);
        }
        v = v * scp[m] + shp[m];
        if (tanh_) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tanh_bf16_epilogue(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * m + e] = (bf16)v[e];
      }
      if (STAGED) {
        *(bf16x8*)(ep + (n * 16 + fr) * EPI_STRIDE + 16 * fq) = o;
        continue;
      }
      const int f = n * 16 + fr, q = tf + f;
      const int r = a.orow ? om[f] : q;                      // orow is -1 beyond the last padded row as well
      const u32x4 v = __builtin_bit_cast(u32x4, o);
      if (q < q_end && r >= 0 && co < a.cout && !((WREG_KNOCK & 8) && v[0] != 0x12345u)) {
        *(u32x4*)(a.y + (size_t)r * a.ldy + co) = v;
        if (a.y2 && co >= a.y2_from) *(u32x4*)(a.y2 + (size_t)r * a.ldy2 + co) = v;
      }
      if constexpr (STATS) {                            // unconditional updates of what was stored (0 for the rows that were not)
        const bool live = q < q_end && r >= 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float x = live ? (float)o[i] : 0.f;
          ssum[i] += x;
          ssq[i] = fmaf(x, x, ssq[i]);
        }
      }
    }
    if (STAGED) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      constexpr int LPR = CM * 2, RPP = 64 / LPR;            // 16-byte lanes per staged row, rows per pass
#pragma unroll
      for (int i = 0; i < WR_TN / RPP; ++i) {
        const int f = i * RPP + lane / LPR, ch = lane % LPR;
        const int q = tf + f, cs = co_w + ch * 8;
        const int r = a.orow ? om[f] : q;
        const u32x4 v = *(const u32x4*)(ep + f * EPI_STRIDE + ch * 16);
        if (q < q_end && r >= 0 && cs < a.cout && !((WREG_KNOCK & 8) && v[0] != 0x12345u)) {
          *(u32x4*)(a.y + (size_t)r * a.ldy + cs) = v;
          if (a.y2 && cs >= a.y2_from) *(u32x4*)(a.y2 + (size_t)r * a.ldy2 + cs) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // once: weights and the first D stages
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // The DMA of a stage goes out one sub-tile at a time: the first right behind the barrier that frees its slot, the
  // others in front of every second half of the stage that follows (WREG_SPREAD; all eight waves issuing the whole
  // stage at once held every wave ~60 clocks per instruction before its next MFMAs)
  if (WR_D < G) issue_part(WR_D, 0, WREG_SPREAD ? 1 : SK);
  bf16x8 fb[2][4];
  frags(fb[0], 0, 0);
  int g = 0;
  auto tile_stages = [&](auto half_tag) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks, ++g) {
      const int cur = g % WR_NS;
#pragma unroll
      for (int h = 0; h + 1 < H; ++h) {
        if (WREG_SPREAD && (h & 1) && (h + 1) / 2 < SK && g + WR_D < G) issue_part(g + WR_D, (h + 1) / 2, (h + 1) / 2 + 1);
        frags(fb[(h + 1) & 1], cur, h + 1);
        mfmas(half_tag, H * ks + h, fb[h & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // own rows of stage g + 1 have landed once at most the D - 1 younger stages are outstanding (epilogue stores in
      // the count can only make the wait longer)
      if (!(WREG_KNOCK & 64)) {
        if (g + WR_D < G) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WR_D - 1) * DI * SK) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (WREG_KNOCK & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (g + 1 + WR_D < G) issue_part(g + 1 + WR_D, 0, WREG_SPREAD ? 1 : SK);   // into the slot stage g was read from
      if (g + 1 < G) frags(fb[0], (g + 1) % WR_NS, 0);
      mfmas(half_tag, H * ks + H - 1, fb[(H - 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int n_full = NTL - (odd ? 1 : 0);
  for (int lt = 0; lt < n_full; ++lt) {
    tile_stages(std::false_type{});
    epilogue(lt);
  }
  if constexpr (HALVES) {
    if (odd) {                                          // accumulators of frames 32 .. 63 stay zero; their rows are past q_end
      tile_stages(std::true_type{});
      epilogue(NTL - 1);
    }
  }
  if constexpr (STATS) {
    // the 16 lanes fr = 0 .. 15 of a group fq hold different frames of the same 8 channels: quad, quad pair, half row,
    // row -- a fixed order, every lane ends with the same totals
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float s1 = ssum[i], s2 = ssq[i];
      s1 += dpp_row_f32<0xB1>(s1);  s2 += dpp_row_f32<0xB1>(s2);     // quad_perm [1,0,3,2]
      s1 += dpp_row_f32<0x4E>(s1);  s2 += dpp_row_f32<0x4E>(s2);     // quad_perm [2,3,0,1]
      s1 += dpp_row_f32<0x141>(s1); s2 += dpp_row_f32<0x141>(s2);    // row_half_mirror
      s1 += dpp_row_f32<0x140>(s1); s2 += dpp_row_f32<0x140>(s2);    // row_mirror
      ssum[i] = s1;
      ssq[i] = s2;
    }
    const int co = co_w + 8 * fq;
    if (fr == 0 && co < a.cout) {
      double* pp = a.stat + (size_t)p * 2 * a.stat_c + co;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        pp[i] = (double)ssum[i];
        pp[a.stat_c + i] = (double)ssq[i];
      }
    }
  }
}

// stat_io (with a.stat set): in = doubles the table a.stat can hold; out = the row blocks written (the frame ranges P),
// or 0 when this launch could not leave the statistics behind
template <int KC, int CM, int SK, int NW = 16 / CM, bool STAGED = false>
int launch_wreg(const Gemm1x1Args& a, hipStream_t s, int64_t* stat_io = nullptr) {
  // the ring + (STAGED) the waves' output staging + the output-row map slots
  constexpr int smem = (12 / SK + 1) * SK * WR_SUB + (STAGED ? NW * WR_TN * (CM * 32 + 16) : 0) + 4 * WR_TN * 4;
  int n_cu = 0;
  if (int rc = device_cu_count(&n_cu)) return rc;
  const int n_slabs = (int)ceil_div(a.cout, 16 * CM * NW), n_half = (int)ceil_div(a.R, WR_TN / 2);
  int P = n_cu / n_slabs / 8 * 8;
  if (P < 8) P = 8;
  if constexpr (!STAGED) {
    if (a.stat && stat_io && a.stat_c == a.cout && (int64_t)P * 2 * a.stat_c <= *stat_io) {
      static OptInMask opted_s{0};
      if (int rc = lds_opt_in((const void*)gemm_wreg_kernel<KC, CM, SK, NW, false, true>, smem, &opted_s)) return rc;
      hipLaunchKernelGGL((gemm_wreg_kernel<KC, CM, SK, NW, false, true>), dim3(n_slabs * P), dim3(64 * NW), smem, s, a, n_half,
                         n_slabs, P);
      SUGAR_LAUNCH_CHECK();
      *stat_io = P;
      return SUGAR_OK;
    }
  }
  if (stat_io) *stat_io = 0;
  static OptInMask opted{0};
  if (int rc = lds_opt_in((const void*)gemm_wreg_kernel<KC, CM, SK, NW, STAGED>, smem, &opted)) return rc;
  hipLaunchKernelGGL((gemm_wreg_kernel<KC, CM, SK, NW, STAGED>), dim3(n_slabs * P), dim3(64 * NW), smem, s, a, n_half, n_slabs, P);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

// dst[(r + 2 h b + h)][C] = src[r][C] for the rows of utterance b (16 bytes per thread), zeros in the h rows before
// and after it; the last utterance's workgroups also zero the `slack` rows that close the buffer
__global__ __launch_bounds__(256) void pad_rows_kernel(const bf16* __restrict__ src, int C, const int32_t* __restrict__ row_off,
                                                       int halo, int slack, bf16* __restrict__ dst) {
  const int b = blockIdx.y;
  const int s0 = row_off[b], n = row_off[b + 1] - s0;
  const int nch = C >> 3;
  const int64_t first = (int64_t)s0 + (int64_t)2 * halo * b;          // padded row of the leading halo
  const int rows = n + 2 * halo + (b == (int)gridDim.y - 1 ? slack : 0);
  for (int it = blockIdx.x * 256 + threadIdx.x; it < rows * nch; it += gridDim.x * 256) {
    const int r = it / nch, ch = it - r * nch, t = r - halo;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (t >= 0 && t < n) v = *(const u32x4*)(src + ((int64_t)s0 + t) * C + ch * 8);
    *(u32x4*)(dst + (first + r) * C + ch * 8) = v;
  }
}

}  // namespace

// ---- tap conv (dilation 1) as a GEMM over overlapping rows of a zero-padded copy of the activations ---------------
// layer1 (80 mel bins, 5 taps: K = 400) is 4 % of the MACs but ran at a tenth of the GEMM rate through the generic
// masked-load kernel; its K vector is contiguous in a channels-last buffer, so the LDS-DMA GEMM can read it in place.
bool gemm_taps_eligible(const ConvLayer& L, int ldx, int dtype) {
  return dtype == SUGAR_BF16 && L.dil == 1 && L.ktaps > 1 && ldx == L.cin_p && L.cin_p % 8 == 0 && L.k_pad % TK == 0 &&
         L.n_pad % 128 == 0 && L.cout > 64 && L.cout % 8 == 0 && !(L.flags & EPI_TANH);
}

int64_t gemm_taps_pad_rows(const ConvLayer& L, int B, int64_t R) {
  const int halo = (L.ktaps - 1) / 2;
  // + slack: the last row's K vector is read up to k_pad elements (zero weights beyond ktaps * cin_p)
  return R + (int64_t)2 * halo * B + ceil_div(L.k_pad, L.cin_p) + 1;
}

int gemm_taps_slack(const ConvLayer& L, int B, int64_t R) {
  const int halo = (L.ktaps - 1) / 2;
  return (int)(gemm_taps_pad_rows(L, B, R) - R - (int64_t)2 * halo * B);
}

// d_xpad <- d_x re-laid with `halo` zero rows before and after every utterance (+ the closing slack rows).  The fused
// front-end writes that layout itself (launch_fbank with halo > 0) and skips this copy.
int launch_pad_rows(const ConvLayer& L, const void* d_x, void* d_xpad, const int32_t* d_row_off, int B, int64_t R, int t_max,
                    hipStream_t s) {
  const int halo = (L.ktaps - 1) / 2, C = L.cin_p;
  SUGAR_CHECK_ARG(((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_xpad & 15) == 0, "gemm_taps: unaligned buffers");
  const int slack = gemm_taps_slack(L, B, R);
  const int64_t items = (int64_t)(t_max + 2 * halo + slack) * (C >> 3);
  const unsigned gx = (unsigned)(ceil_div(items, 256) < 64 ? ceil_div(items, 256) : 64);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, (const bf16*)d_x, C, d_row_off, halo, slack,
                     (bf16*)d_xpad);
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

// orow[q] = output row of GEMM row q of the padded row space (launch_pad_rows layout: utterance b's frame t is the
// K vector starting at padded row row_off[b] + 2 halo b + t), or -1 for the 2 halo rows between utterances and for the
// rows up to the next multiple of 64 behind the last one.
__global__ __launch_bounds__(256) void build_orow_kernel(const int32_t* __restrict__ row_off, int halo, int n_total,
                                                         int32_t* __restrict__ orow) {
  const int b = blockIdx.y, B = gridDim.y;
  const int s0 = row_off[b], n = row_off[b + 1] - s0;
  const int q0 = s0 + 2 * halo * b;
  const int rows = b == B - 1 ? n_total - q0 : n + 2 * halo;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) orow[q0 + i] = i < n ? s0 + i : -1;
}

int64_t gemm_taps_orow_ints(const ConvLayer& L, int B, int64_t R) {
  return round_up(R + (int64_t)(L.ktaps - 1) * B, WR_TN);
}

int launch_gemm_taps(const ConvLayer& L, const void* d_xpad, int32_t* d_orow, void* d_y, int ldy, const int32_t* d_row_off,
                     int B, int64_t R, int t_max, hipStream_t s, bool orow_ready, double* d_stat, int stat_c, int64_t* stat_io) {
  const int halo = (L.ktaps - 1) / 2, C = L.cin_p;
  SUGAR_CHECK_ARG(((uintptr_t)d_xpad & 15) == 0 && ((uintptr_t)d_y & 15) == 0 && ldy % 8 == 0, "gemm_taps: unaligned buffers");
  Gemm1x1Args a;
  a.x = (const bf16*)d_xpad; a.w = (const bf16*)L.w; a.y = (bf16*)d_y; a.y2 = nullptr;
  a.scale = L.scale; a.shift = L.shift;
  a.ldx = C; a.ldw = L.k_pad; a.ldy = ldy; a.ldy2 = 0; a.y2_from = 0;
  a.R = (int)R; a.cout = L.cout; a.K = L.k_pad; a.flags = L.flags;
  if (L.wfrag && d_orow && L.k_pad == 448 && L.cout % 256 == 0 && R >= tune_get(TUNE_GEMM_WREG_MIN_R)) {
    // large inputs: weights in registers, one pass over the padded rows (the tile kernel below re-streams the weight
    // panel and the activation rows from L2 for each of its 128 x 128 tiles: 554 MB of DMA for 91 MB of tensors)
    const int64_t n_total = gemm_taps_orow_ints(L, B, R);
    if (!orow_ready) {
      const unsigned gx = (unsigned)(ceil_div(t_max + 2 * halo + WR_TN, 256) < 8 ? ceil_div(t_max + 2 * halo + WR_TN, 256) : 8);
      hipLaunchKernelGGL(build_orow_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, d_row_off, halo, (int)n_total, d_orow);
      SUGAR_LAUNCH_CHECK();
    }
    a.R = (int)(R + (int64_t)2 * halo * B);           // GEMM rows = padded rows
    a.orow = d_orow;
    a.wfrag = (const bf16*)L.wfrag;
    a.stat = d_stat; a.stat_c = stat_c;
    return launch_wreg<7, 2, 7, 8>(a, s, stat_io);
  }
  if (stat_io) *stat_io = 0;
  a.n_co_tiles = (int)ceil_div(L.cout, 128);
  a.row_off = d_row_off; a.B = B; a.xmap = 2 * halo;
  return launch_variant<128, 128, 2>(a, s);
}

// Eligibility: bf16, single tap, no addend, K tile-aligned, weight rows padded to the channel tile.
bool gemm1x1_eligible(const ConvLayer& L, const ConvIO& io, int dtype) {
  const bool wide = L.cout > 128 && L.n_pad % 256 == 0;       // 256-channel tiles
  const bool mid = L.cout > 64 && L.cout <= 128 && L.n_pad == 128;   // one 128-channel tile (pooling.linear1)
  return dtype == SUGAR_BF16 && L.ktaps == 1 && !io.x2 && (wide || mid) && L.cin_p % TK == 0 &&
         L.k_pad == L.cin_p && L.cout % 8 == 0 && io.ldy % 8 == 0 && io.ldx % 8 == 0 && ((uintptr_t)io.x & 15) == 0 &&
         ((uintptr_t)io.y & 15) == 0 && (!io.y2 || (io.ldy2 % 8 == 0 && ((uintptr_t)io.y2 & 15) == 0 && io.y2_from % 8 == 0));
}

int launch_gemm1x1(const ConvLayer& L, const ConvIO& io, int64_t R, hipStream_t s, int64_t* stat_io) {
  if (stat_io && !(io.stat_part && L.cout > 128 && L.cin_p <= 768)) *stat_io = 0;   // only the K = 512 weights-in-registers launch below
  Gemm1x1Args a;
  a.x = (const bf16*)io.x; a.w = (const bf16*)L.w; a.y = (bf16*)io.y; a.y2 = (bf16*)io.y2;
  a.scale = L.scale; a.shift = L.shift;
  a.ldx = io.ldx; a.ldw = L.k_pad; a.ldy = io.ldy; a.ldy2 = io.ldy2; a.y2_from = io.y2_from;
  a.R = (int)R; a.cout = L.cout; a.K = L.cin_p; a.flags = L.flags;
  if (L.cout > 128) {
    if (a.K <= 768) {
      // short K loops (the 512-channel convs: 8 k-tiles): tile prologue and epilogue are ~1/3 of a 256 x 256 tile's
      // time and one such workgroup fills the CU's LDS, so nothing overlaps them.  128 x 128 tiles at two
      // workgroups per CU do (tools/gemm_variants: 0.051 vs 0.057 ms on 77056 x 512 x 512); for K = 1536 the
      // 256 x 256 tile's lower L2 traffic and DMA issue count per MFMA win.
      // (in the pipeline, in one job: first_conv + last_conv 0.404 ms with these tiles, 0.429 with the 256 x 256
      // split grid, 0.425 with its ping-pong form)
      // K = 512, large R: weights-in-registers kernel (a quarter of the DMA bytes per MFMA, x read once from HBM).
      // Same job, Base 256 x 3 s (R = 77 056): one stream 178.4 k vs 177.0 k emb/s, two streams 199.8 k vs 191.0 k.
      // Its weight load is a fixed ~7 us per launch, so short inputs stay on the tile kernel (R = 12 040: 22 vs 15 us);
      // whole forwards crossed over at ~140 x 3 s in round 1 (profiles/r01_wreg_threshold.log); with the round-2 kernel
      // (pre-packed fragments, half-tile ranges, spread DMA issue) at ~64 x 3 s: 128 x 3 s 175.5 k -> 185.4 k emb/s,
      // 64 x 3 s equal (profiles/r02_wreg_threshold.log), so the default threshold is 19 200 frames.
      // The tunable gemm_wreg_min_r moves the threshold (0: always, for the parity tests; huge: never, for A/B runs).
      const int64_t wreg_min_r = tune_get(TUNE_GEMM_WREG_MIN_R);
      // (K = 256 .. 448 instantiations were measured on the NAS sweep, tools/sweep_bench.py: 31.5-32.7 vs 30.1-31.0 ms
      // per subnet, so the narrower convs stay on the tile kernel)
      // (the persistent 256 x 256 kernel on these shapes: 0.208 / 0.192 ms per step for first / last conv against
      // 0.170 / 0.158 here -- 8 k-tiles per tile leave its epilogue a third of the tile; on the quadrant schedule, whose
      // DMA stream runs across tiles, 0.180 / 0.169 against 0.167 / 0.155)
      if (L.wfrag && R >= wreg_min_r && a.K == 512 && L.cout % 256 == 0) {
        a.wfrag = (const bf16*)L.wfrag;
        a.stat = io.stat_part; a.stat_c = io.stat_c;
        return launch_wreg<8, 2, 4>(a, s, stat_io);
      }
      if (stat_io) *stat_io = 0;
      a.n_co_tiles = (int)ceil_div(L.cout, 128);
      return launch_variant<128, 128, 2>(a, s);
    }
    a.n_co_tiles = (int)ceil_div(L.cout, 256);
    const bool uniform = tune_get(TUNE_GEMM_UNIFORM) != 0;         // A/B switch: no split tail
    const bool no_pp = tune_get(TUNE_GEMM_NO_PINGPONG) != 0;       // A/B switch: both waves of a SIMD in step
    if (uniform) return launch_variant<256, 256, 2>(a, s);
    return no_pp ? launch_split<0>(a, s) : launch_split<1>(a, s);
  }
  // pooling.linear1 at full width (K = 1536 -> 128 + tanh) on large inputs: weights in registers, only the activations
  // stream (the tunable gemm_wreg_min_r switches it together with the 512-channel convs)
  // (its crossover lies higher than the convs': one wave per SIMD; 64 x 3 s 0.0202 vs 0.0252 ms on the tile kernel,
  // 128 x 3 s 0.0423 vs 0.0401)
  if (L.wfrag && a.K == 1536 && L.cout == 128 && R >= tune_get(TUNE_GEMM_WREG_MIN_R) / 2 * 3) {
    a.wfrag = (const bf16*)L.wfrag;
    // (walking the frame tiles from the last one down, so that the first reads hit what the catconv launch wrote last
    // into the Infinity Cache, measured no difference: 70.8 vs 70.9 us in the pipeline)
    return launch_wreg<24, 2, 4, 4, true>(a, s);
  }
  a.n_co_tiles = 1;
  return launch_variant<128, 64, 3>(a, s);
}

}  // namespace sugar
