// Res2 chain in one kernel (gfx950, bf16 operands / f32 accumulate).
//
// DynamicRes2Conv1dReLUBN.forward (sugar/models/dynamictdnn.py:290-303): the first_conv output is split into 8 groups
// of `w` channels; 7 SEQUENTIAL dilated convs   sp_j = BN(ReLU(conv_j(x_j + sp_{j-1})))   (sp_{-1} = 0), the 8th group
// passes through.  As 7 launches each step is a tiny GEMM (M = w <= 64) and the chain is launch/latency-bound (the
// reference's V100 table shows ~3.2 ms per block whatever the width, tdnn8m2g_latency_v100_80_300.yml:7155).
//
// Here one workgroup (8 waves) owns one utterance for the whole chain:
//   * u = x_j + sp_{j-1} lives in LDS as [frame + halo][w] bf16 (zero halo rows = the conv's "same" padding at the
//     utterance's true ends), chunk-swizzled so the shifted fragment reads of every tap are bank-conflict free;
//   * each wave owns up to 3 tiles of 16 frames; D[channel][frame] = sum over (tap, ci) W[co][tap*w+ci] u[frame+off][ci]
//     with v_mfma_f32_16x16x32_bf16; the weights of step j+1 are prefetched into registers during step j and
//     staged in LDS with the same XOR swizzle; the folded-BN vectors of all 7 steps sit in LDS;
//   * global traffic is row-contiguous, 16 bytes per lane: the MFMA epilogue (ReLU -> folded BN) drops sp_j of a
//     frame tile into LDS in MFMA layout, the wave reads it back as whole 16-byte chunks, stores them to the output
//     slice and adds the x_{j+1} chunks (prefetched two steps ahead in the same layout) to form u_{j+1}.
//     In-kernel stamps showed the earlier 8-byte-per-lane loads/stores in MFMA layout (16 half-used lines per
//     instruction, 37 vector-memory instructions per wave and step) spent 60 % of every step in memory issue;
//   * DB form (3-tap panels): u and the weight panel are double-buffered, the staging area of a wave is its own rows of
//     the next u buffer, u_{j+1} is left there in place: one workgroup barrier per step;
//   * every address that does not change from step to step is a per-lane offset computed once and pinned in a
//     register: recomputed per fragment read / per staged item, the address arithmetic (quarter-rate 32-bit
//     multiplies, swizzles) took more VALU issue time than the step's 72 MFMAs per wave.
// Utterances longer than the 384 frames one workgroup holds are cut into time segments, one workgroup each: a segment
// processes its frames plus H = 7 * halo context frames on every side that is not a true utterance end.  Treating the
// clipped side as zero padding corrupts sp_j within (j + 1) * halo frames of it, i.e. never inside the stored range,
// so the outputs equal the unsegmented chain bit for bit (the context frames are recomputed: +17 % work at halo 4).
// f32 takes the generic per-step path in conv_gemm.hip.
#include <stdlib.h>

#include "common.h"

namespace sugar {
namespace {

// x (the first_conv output) is dead once the chain has read it: non-temporal loads, so that it leaves the caches
// first and the chain's output, which last_conv reads next, stays.  Same job: Res2 0.130 -> 0.133, last_conv 0.146 ->
// 0.141 ms per step.
#ifndef R2_NT
#define R2_NT 1
#endif
constexpr int R2_TPW_MAX = 3;        // frame tiles (of 16) per wave
constexpr int R2_WAVES = 8;
constexpr int R2_MAX_T = 16 * R2_TPW_MAX * R2_WAVES;   // 384 frames per workgroup (segment incl. its context)
constexpr int R2_STEPS = SUGAR_RES2_SCALE - 1;

struct Res2Args {
  const bf16* x;     // first_conv output [R][ld]
  bf16* y;           // Res2 output [R][ld] (groups 0..6 written here)
  const bf16* w[R2_STEPS];        // packed [64][k_pad]
  const float* scale[R2_STEPS];
  const float* shift[R2_STEPS];
  const int32_t* row_off;
  int ld, wp, ktaps, dil, k_pad;
  int n_pad;         // rows of the weight panels = length of the scale / shift vectors (64 or 128)
  int max_t;         // frames one workgroup processes: R2_MAX_T (the parity tests lower it to force segments)
  int seg_inner;     // most stored frames per segment (multiple of 16): max_t - 2 * 7 * halo
};

// Diagnostic build only (tools/res2_bench.hip defines R2_STAMPS): shader-clock stamps of wave 0 / wave 7 of two
// workgroups go to a buffer of their own; the library build compiles none of this.
#ifdef R2_STAMPS
__device__ long long* g_r2_stamps;
#define R2_STAMP()                                                                                      \
  do {                                                                                                  \
    if (stamp_on) { stamp_buf[stamp_n] = (long long)__builtin_amdgcn_s_memtime(); ++stamp_n; }          \
  } while (0)
#else
#define R2_STAMP() do {} while (0)
#endif

__device__ __forceinline__ u32x4 add_bf16x8(u32x4 xa, u32x4 xb) {
  // per element: bf16(f32(a) + f32(b)) -- the reference adds x_{j+1} + sp_j in f32 (dynamictdnn.py:298)
  const bf16x8 a = __builtin_bit_cast(bf16x8, xa), b = __builtin_bit_cast(bf16x8, xb);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16)((float)a[e] + (float)b[e]);
  return __builtin_bit_cast(u32x4, r);
}

// Round 2, measured and dropped: more waves per utterance.  With B = 256 utterances on 256 CUs the kernel runs ONE
// workgroup of 8 waves per CU (234 VGPRs), 3 frame tiles per wave; 12 waves x 2 tiles (168 VGPRs, 13 spilled) gave
// bit-identical outputs but 0.186 ms per step against 0.131 (more fragment reads of the shared weight panel per MFMA,
// 12-wave barriers), and cutting 301-frame utterances into two time segments for two workgroups per CU 0.205 ms.
// WREG = 16-byte weight chunks each thread stages per step: 3 (k_pad <= 192) or 5 (k_pad <= 320)
// DB (the default where it fits 160 KiB: 64-channel groups of 3 taps): u and the weight panel are double-buffered and a
// wave's own rows of the NEXT u buffer serve as its sp staging area, so the epilogue of step j leaves u_{j+1} in place
// and a step needs ONE workgroup barrier (without DB: staging tiles of their own, u_{j+1} written between two barriers).
template <int MT, int R2_TPW, int WREG, bool DB>
__global__ __launch_bounds__(512, 2) void res2_chain_kernel(Res2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NI = (MT + 1) / 2;                      // 16-byte chunk items per lane and frame tile
  const int b = blockIdx.x;
  const int wp = a.wp, ctr = (a.ktaps - 1) >> 1, halo = ctr * a.dil;
  // segment blockIdx.y of utterance b: stored frames [t0, t1), processed frames [p0, p1) of the utterance
  const int u0 = a.row_off[b], Tu = a.row_off[b + 1] - u0;
  const int nseg = Tu <= a.max_t ? 1 : (Tu + a.seg_inner - 1) / a.seg_inner;
  if ((int)blockIdx.y >= nseg) return;
  const int inner = nseg == 1 ? Tu : (((Tu + nseg - 1) / nseg + 15) & ~15);
  const int t0 = blockIdx.y * inner, t1 = min(Tu, t0 + inner);
  if (t0 >= t1) return;
  const int ctx = R2_STEPS * halo;
  const int p0 = max(0, t0 - ctx), p1 = min(Tu, t1 + ctx);
  const int s0 = u0 + p0, T = p1 - p0, st_lo = t0 - p0, st_hi = t1 - p0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int f = lane & 15, q = lane >> 4;
#ifdef R2_STAMPS
  const bool stamp_on = g_r2_stamps && lane == 0 && (wv == 0 || wv == 7) && (b == 0 || b == 200);
  long long* stamp_buf = g_r2_stamps + ((b ? 2 : 0) + (wv ? 1 : 0)) * 64;
  int stamp_n = 0;
#endif
  R2_STAMP();
  const int NT = (T + 15) >> 4;
  const int urow = wp * 2;                              // bytes per u row
  const int nch = wp >> 3;                              // 16-byte chunks per u row
  const int swz_mask = (nch & (nch - 1)) == 0 ? (nch - 1 < 7 ? nch - 1 : 7) : 0;
  const int u_rows = NT * 16 + 2 * halo;
  const int u_bytes = (u_rows * urow + 255) & ~255;
  const int wstride = a.k_pad * 2;                      // weight rows: groups of 8 chunks, chunk ^= row & 7 inside a group
  const int wchunks = a.k_pad >> 3;                     // chunks per weight row (multiple of 8)
  constexpr int NB = DB ? 2 : 1;
  unsigned char* u = smem;                              // the u tile of the current step (DB: alternates)
  unsigned char* wl = smem + NB * u_bytes;              // its weight panel
  constexpr int OT = R2_TPW <= 3 ? R2_TPW : 1;          // sp staging tiles per wave
  unsigned char* ob = wl + 64 * wstride + wv * (OT * 16 * urow);        // (!DB) this wave's sp staging tiles [16][wp]
  float* ss = (float*)(wl + NB * 64 * wstride + (DB ? 0 : R2_WAVES * OT * 16 * urow));  // [7][scale 64 | shift 64]
  const int nks = a.k_pad >> 5;                         // k-steps of 32

  // zero the whole u tile once: halo rows and frames >= T stay zero for the whole chain
  for (int i = tid; i < NB * u_bytes / 16; i += 512) ((u32x4*)u)[i] = (u32x4){0u, 0u, 0u, 0u};
  for (int i = tid; i < R2_STEPS * 128; i += 512) {
    const int j = i >> 7, c = i & 63;
    ss[i] = (i & 64) ? a.shift[j][c] : a.scale[j][c];
  }

  int ntw = 0;                                          // tiles owned by this wave: t = wv + 8 n
#pragma unroll
  for (int n = 0; n < R2_TPW; ++n) ntw += (wv + R2_WAVES * n) < NT;

  // per-thread weight staging slots: chunk idx = tid + 512 i of the [64][wchunks] panel
  int wsrc[WREG], wdst[WREG];
#pragma unroll
  for (int i = 0; i < WREG; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / wchunks, c = idx - row * wchunks;
    wsrc[i] = idx < 64 * wchunks ? row * a.k_pad + c * 8 : -1;
    wdst[i] = row * wstride + (((c & ~7) | ((c & 7) ^ (row & 7))) << 4);
  }
  u32x4 wreg[WREG];
  auto load_w = [&](int j) {
#pragma unroll
    for (int i = 0; i < WREG; ++i) wreg[i] = *(const u32x4*)(a.w[j] + max(wsrc[i], 0));   // unconditional, see load_x
  };
  auto store_w = [&](unsigned char* wdst_base) {
#pragma unroll
    for (int i = 0; i < WREG; ++i)
      if (wsrc[i] >= 0) *(u32x4*)(wdst_base + wdst[i]) = wreg[i];
  };

  // row-contiguous chunk items of a 16-frame tile: item = lane + 64 i -> (frame it / nch, chunk it % nch)
  int irow[NI], ich[NI], ooff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int it = lane + 64 * i;
    const int rl = it / nch;
    irow[i] = rl < 16 ? rl : 16;                        // 16 = no such item
    ich[i] = it - rl * nch;
    ooff[i] = (rl & 15) * urow + ((ich[i] ^ (rl & swz_mask)) << 4);
  }
  const bf16* xb = a.x + (size_t)s0 * a.ld;
  bf16* yb = a.y + (size_t)s0 * a.ld;
  // xs[(j+1)&1] holds the x_{j+1} chunks during step j and is turned in place into u_{j+1} = x_{j+1} + sp_j by the
  // epilogue.  With <= 3 tiles per wave the registers allow loading TWO steps ahead (x_{j+2} during step j): the
  // stamps showed a one-step prefetch still exposed ~2 us of memory latency per step, with every CU bursting at once.
  constexpr bool DEEP = R2_TPW <= 3;
  u32x4 xs[2][R2_TPW][NI];
  // Byte offsets that do not change from step to step, computed once and pinned in registers (recomputed in all 7
  // unrolled epilogues they were a third of the epilogue's VALU time: 24 quarter-rate 32-bit multiplies per step):
  //   uoff: item (n, i) inside a u buffer;  xoff: its row and chunk in x and y (without the group's column; the item's
  //   own row where it has one, else a valid row: loads are unconditional, stores masked);
  //   soff / smx: the MFMA-layout rows and chunk slots stage_sp writes (DB: in the next u buffer)
  int uoff[R2_TPW][NI], xoff[R2_TPW][NI], soff[R2_TPW], smx[MT];
#pragma unroll
  for (int n = 0; n < R2_TPW; ++n) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (wv + R2_WAVES * n) * 16 + irow[i];
      const int row = min(r, 16 * NT - 1) + halo;       // (irow 16 = no such item: any row of the buffer, never used)
      uoff[n][i] = row * urow + ((ich[i] ^ (row & swz_mask)) << 4);
      xoff[n][i] = (min((wv + R2_WAVES * n) * 16 + min(irow[i], 15), T - 1) * a.ld + ich[i] * 8) * 2;   // a valid row for every lane
      asm volatile("" : "+v"(uoff[n][i]), "+v"(xoff[n][i]));
    }
    soff[n] = ((wv + R2_WAVES * n) * 16 + f + halo) * urow;
    asm volatile("" : "+v"(soff[n]));
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int co = m * 16 + 4 * q, srow = f + halo;     // (16-row tiles: row & swz_mask does not depend on the tile)
    smx[m] = (((co >> 3) ^ (srow & swz_mask)) << 4) + (co & 7) * 2;
    asm volatile("" : "+v"(smx[m]));
  }
  // fragment addresses of the K loop (see load_frags): boff[ks] = this lane's row (frame f + the tap's shift) and
  // swizzled chunk of k-step ks inside a u buffer for a tile at row 0, aoff0 / aoff1 = its weight row and swizzled
  // chunk in even / odd k-steps, tbase[n] = the wave-uniform row base of tile n
  constexpr int NKS = WREG == 3 ? 6 : 10;               // k_pad <= 192 / 320
  int boff[NKS], tbase[R2_TPW];
  {
    const int inv_wp_ = (65536 + wp - 1) / wp;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int c = 4 * ks + q;                         // this lane's K chunk
      const int tap = (c * 8 * inv_wp_) >> 16, ci = c * 8 - tap * wp;   // == c*8 / wp, % wp  (c*8 < 1024)
      const bool kpad = tap >= a.ktaps || ks >= nks;    // K padding: weights are zero, read any valid row
      const int row = f + halo + (kpad ? 0 : (tap - ctr) * a.dil);
      boff[ks] = row * urow + (((kpad ? 0 : (ci >> 3)) ^ (row & swz_mask)) << 4);
      asm volatile("" : "+v"(boff[ks]));
    }
  }
#pragma unroll
  for (int n = 0; n < R2_TPW; ++n)
    tbase[n] = __builtin_amdgcn_readfirstlane((n < ntw ? (wv + R2_WAVES * n) * 16 : 0) * urow);
  int aoff0 = f * wstride + ((q ^ (f & 7)) << 4), aoff1 = f * wstride + (((4 + q) ^ (f & 7)) << 4);
  asm volatile("" : "+v"(aoff0), "+v"(aoff1));
  auto item_ok = [&](int n, int i) { return n < ntw && irow[i] < 16 && (wv + R2_WAVES * n) * 16 + irow[i] < T; };
  auto load_x = [&](int j, int buf) {
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        // unconditional load from a clamped (always valid) address, then select: a branch around each load would
        // make hipcc wait vmcnt(0) per element and serialise the memory round trips
        const unsigned char* src = (const unsigned char*)(xb + j * wp) + xoff[n][i];   // items not item_ok are never used
        if (R2_NT) xs[buf][n][i] = __builtin_nontemporal_load((const u32x4*)src);
        else xs[buf][n][i] = *(const u32x4*)src;
      }
  };
  auto write_u = [&](int buf) {
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n)
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (item_ok(n, i)) *(u32x4*)(u + uoff[n][i]) = xs[buf][n][i];
  };

  load_w(0);
  load_x(0, 0);                                         // u_0 = x_0
  if (DEEP) load_x(1, 1);
  __syncthreads();                                      // zero fill complete before the first u write
  R2_STAMP();
  write_u(0);
  store_w(wl);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned char* const u_first = u;
  unsigned char* const wl_first = wl;
#pragma unroll
  for (int j = 0; j < R2_STEPS; ++j) {                  // fully unrolled: the xs buffer index is static
    const bool last = j + 1 == R2_STEPS;
    // DB: this step reads buffer j & 1 and builds u_{j+1} / stages the next panel in the other one
    unsigned char* const un = DB ? u_first + ((j + 1) & 1) * u_bytes : u_first;
    unsigned char* const wn = DB ? wl_first + ((j + 1) & 1) * (64 * wstride) : wl_first;
    if (DB) {
      u = u_first + (j & 1) * u_bytes;
      wl = wl_first + (j & 1) * (64 * wstride);
    }
    R2_STAMP();
    if (!last) load_w(j + 1);                           // in flight during the MFMA loop
    if (DEEP) {
      if (j + 2 < R2_STEPS) load_x(j + 2, j & 1);
    } else if (!last) {
      load_x(j + 1, (j + 1) & 1);
    }

    f32x4 acc[R2_TPW][MT];
#pragma unroll
    for (int n = 0; n < R2_TPW; ++n)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    R2_STAMP();

    // K loop, fragments of k-step ks+1 read while the MFMAs of ks run (nks is even: k_pad % 64 == 0).  Tiles past
    // ntw read a clamped (valid) row and their MFMAs are skipped.  Every address is a pinned per-lane offset (aoff /
    // boff above) + a wave-uniform term + an immediate: computed afresh per k-step (tap = chunk / wp, row * urow,
    // the swizzles) the loop spent more VALU time on 32-bit multiplies than the MFMAs take.
    auto load_frags = [&](int ks, bf16x8 (&af)[MT], bf16x8 (&bf)[R2_TPW]) {   // ks: compile-time after unrolling
      const unsigned char* ap = wl + ((ks & 1) ? aoff1 : aoff0) + (ks >> 1) * 128;
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *(const bf16x8*)(ap + m * 16 * wstride);
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n) bf[n] = *(const bf16x8*)(u + boff[ks] + tbase[n]);
    };
    auto mma = [&](const bf16x8 (&af)[MT], const bf16x8 (&bf)[R2_TPW]) {
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n)
        if (n < ntw) {
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[n][m], 0, 0, 0);
        }
    };
    {
      bf16x8 af0[MT], bf0[R2_TPW], af1[MT], bf1[R2_TPW];
      load_frags(0, af0, bf0);
#pragma unroll
      for (int ks = 0; ks < NKS; ks += 2) {
        if (ks < nks) {                                 // (wave-uniform)
          load_frags(ks + 1, af1, bf1);
          mma(af0, bf0);
          if (ks + 2 < NKS && ks + 2 < nks) load_frags(ks + 2 < NKS ? ks + 2 : 0, af0, bf0);
          mma(af1, bf1);
        }
      }
    }
    R2_STAMP();

    // epilogue, one frame tile at a time: lane (f, q) holds frame f, channels m*16 + 4q + {0..3} -> wave-private LDS
    // tile -> whole 16-byte chunks.  LDS executes one wave's instructions in order, so the tile is reused without
    // waiting; the wave barriers only pin the compiler's order.
    f32x4 sc[MT], sh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      sc[m] = *(const f32x4*)(ss + j * 128 + m * 16 + 4 * q);
      sh[m] = *(const f32x4*)(ss + j * 128 + 64 + m * 16 + 4 * q);
    }
    auto stage_sp = [&](int n) {
      // DB: row (frame + halo) of the next u buffer, in u's own swizzle; frames >= T keep their zeros
      unsigned char* ot = DB ? un + soff[n] : ob + (OT > 1 ? n : 0) * (16 * urow) + f * urow;
      const bool f_ok = !DB || (wv + R2_WAVES * n) * 16 + f < T;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int co = m * 16 + 4 * q;
        f32x4 v = acc[n][m];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = relu_f32(v[e]);
        v = v * sc[m] + sh[m];
        const bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        if (co < wp && f_ok)
          *(bf16x4*)(ot + (DB ? smx[m] : (((co >> 3) ^ (f & swz_mask)) << 4) + (co & 7) * 2)) = o;
      }
    };
    auto drain_sp = [&](int n) {
      const unsigned char* ot = ob + (OT > 1 ? n : 0) * (16 * urow);
      // all staged chunks of the tile are read first (valid addresses for every lane), so that one LDS round trip is
      // exposed per tile and not one per item behind its branch
      u32x4 spv[NI];
      unsigned char* up[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        up[i] = un + uoff[n][i];
        spv[i] = DB ? *(const u32x4*)up[i] : *(const u32x4*)(ot + ooff[i]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(spv[i]));
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = (wv + R2_WAVES * n) * 16 + irow[i];
        const u32x4 sp = spv[i];
        if (item_ok(n, i)) {
          if (r >= st_lo && r < st_hi) *(u32x4*)((unsigned char*)(yb + j * wp) + xoff[n][i]) = sp;
          if (!last) {
            xs[(j + 1) & 1][n][i] = add_bf16x8(xs[(j + 1) & 1][n][i], sp);
            if (DB) *(u32x4*)up[i] = xs[(j + 1) & 1][n][i];     // u_{j+1} = x_{j+1} + sp_j, in place
          }
        }
      }
    };
    if (OT > 1) {                                       // a staging tile per frame tile: one LDS round trip per step
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n)
        if (n < ntw) stage_sp(n);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n)
        if (n < ntw) drain_sp(n);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int n = 0; n < R2_TPW; ++n) {
        if (n < ntw) {
          stage_sp(n);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          drain_sp(n);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    R2_STAMP();
    // LDS-only barriers: waves exchange data through LDS alone, so only the LDS counter is drained; a full
    // __syncthreads() would also wait (vmcnt(0)) for this step's output stores and the prefetches in flight
    if (DB) {
      if (!last) store_w(wn);
      R2_STAMP();
      R2_STAMP();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves done reading u and the weights
      R2_STAMP();
      if (!last) {
        write_u((j + 1) & 1);
        store_w(wl);
      }
      R2_STAMP();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

int r2_smem_bytes(int t_max, int wp, int halo, int k_pad, bool db = false) {
  t_max = t_max < R2_MAX_T ? t_max : R2_MAX_T;
  const int u_rows = (t_max + 15) / 16 * 16 + 2 * halo;
  const int ot = 3;                                     // sp staging tiles per wave (kernel: OT)
  if (db) return 2 * ((u_rows * wp * 2 + 255) & ~255) + 2 * 64 * k_pad * 2 + R2_STEPS * 128 * 4;
  return ((u_rows * wp * 2 + 255) & ~255) + 64 * k_pad * 2 + R2_WAVES * ot * 16 * wp * 2 + R2_STEPS * 128 * 4;
}


// ---- wide groups (64 < w <= 128: the 1024-channel backbones) -------------------------------------------------------
// Same chain, same sums, another decomposition: a step's weight panel [w][k_pad] (up to 128 x 640 bf16 = 160 KiB) does not
// fit LDS next to the utterance's u tile (up to 100 KiB), so a step runs as NSUB sub-passes over 16 SUBM output channels
// each, whose weights (<= 48 KiB) are staged while the previous sub-pass computes.  The two MFMA tiles of a channel pair
// are interleaved as in the weights-in-registers GEMM (wreg_row): lane (f, q) then owns 8 consecutive channels of its
// frames, so sp_j goes to global memory and into u_{j+1} = x_{j+1} + sp_j as 16-byte chunks straight from the
// accumulators -- no staging tile (there is no LDS left for one).  x_{j+1} is prefetched in that same lane layout.
template <int SUBM, int WREG>
__global__ __launch_bounds__(512, 2) void res2_wide_kernel(Res2Args a) {
  constexpr int R2_WAVES = 8, TPW = 3, SUBC = 16 * SUBM, NP = SUBM / 2, NSUB_MAX = 128 / SUBC, NPG = 4;
  static_assert(SUBM == 2 || SUBM == 4, "whole channel pairs per sub-pass");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int wp = a.wp, ctr = (a.ktaps - 1) >> 1, halo = ctr * a.dil;
  const int u0 = a.row_off[b], Tu = a.row_off[b + 1] - u0;
  const int nseg = Tu <= a.max_t ? 1 : (Tu + a.seg_inner - 1) / a.seg_inner;
  if ((int)blockIdx.y >= nseg) return;
  const int inner = nseg == 1 ? Tu : (((Tu + nseg - 1) / nseg + 15) & ~15);
  const int t0 = blockIdx.y * inner, t1 = min(Tu, t0 + inner);
  if (t0 >= t1) return;
  const int ctx = R2_STEPS * halo;
  const int p0 = max(0, t0 - ctx), p1 = min(Tu, t1 + ctx);
  const int s0 = u0 + p0, T = p1 - p0, st_lo = t0 - p0, st_hi = t1 - p0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int f = lane & 15, q = lane >> 4;
  const int NT = (T + 15) >> 4;
  const int urow = wp * 2, nch = wp >> 3, nsub = wp / SUBC, npg = wp >> 5;
  const int swz_mask = (nch & (nch - 1)) == 0 ? 7 : 0;
  const int u_rows = NT * 16 + 2 * halo;
  const int u_bytes = (u_rows * urow + 255) & ~255;
  const int wstride = a.k_pad * 2, wchunks = a.k_pad >> 3, nks = a.k_pad >> 5;
  unsigned char* u = smem;
  unsigned char* wl = smem + u_bytes;
  float* ss = (float*)(wl + SUBC * wstride);            // [7][scale 128 | shift 128]

  for (int i = tid; i < u_bytes / 16; i += 512) ((u32x4*)u)[i] = (u32x4){0u, 0u, 0u, 0u};
  for (int i = tid; i < R2_STEPS * 256; i += 512) {
    const int j = i >> 8, c = i & 127;
    ss[i] = c >= a.n_pad ? 0.f : (i & 128) ? a.shift[j][c] : a.scale[j][c];
  }
  int ntw = 0;
#pragma unroll
  for (int n = 0; n < TPW; ++n) ntw += (wv + R2_WAVES * n) < NT;

  // weight staging: chunk idx = tid + 512 i of a sub-pass's [SUBC][wchunks] panel; LDS row 16 m + i holds channel
  // 32 (m / 2) + wreg_row(m % 2, i) of the sub-pass
  int wsrc[WREG], wdst[WREG];
#pragma unroll
  for (int i = 0; i < WREG; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / wchunks, c = idx - row * wchunks;
    const int ch = 32 * (row >> 5) + wreg_row((row >> 4) & 1, row & 15);
    wsrc[i] = idx < SUBC * wchunks ? ch * a.k_pad + c * 8 : -1;
    wdst[i] = row * wstride + (((c & ~7) | ((c & 7) ^ (row & 7))) << 4);
  }
  u32x4 wreg[WREG];
  auto load_w = [&](int j, int sub) {
    const bf16* wb = a.w[j] + (size_t)sub * SUBC * a.k_pad;
#pragma unroll
    for (int i = 0; i < WREG; ++i) wreg[i] = *(const u32x4*)(wb + max(wsrc[i], 0));
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WREG; ++i)
      if (wsrc[i] >= 0) *(u32x4*)(wl + wdst[i]) = wreg[i];
  };

  const bf16* xb = a.x + (size_t)s0 * a.ld;
  bf16* yb = a.y + (size_t)s0 * a.ld;
  // xs[n][pg]: channels 32 pg + 8 q .. + 7 of frame 16 (wv + 8 n) + f: x_{j+1} during step j, turned in place into u_{j+1}
  u32x4 xs[TPW][NPG];
  auto load_x = [&](int j) {
#pragma unroll
    for (int n = 0; n < TPW; ++n)
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) {                 // clamped (always valid) addresses; unused items are never stored
        const int r = min((wv + R2_WAVES * n) * 16 + f, T - 1);
        xs[n][pg] = *(const u32x4*)(xb + r * a.ld + j * wp + min(4 * pg + q, nch - 1) * 8);
      }
  };
  auto write_u = [&]() {
#pragma unroll
    for (int n = 0; n < TPW; ++n)
#pragma unroll
      for (int pg = 0; pg < NPG; ++pg) {
        const int r = (wv + R2_WAVES * n) * 16 + f;
        if (n < ntw && pg < npg && r < T) {
          const int row = r + halo;
          *(u32x4*)(u + row * urow + (((4 * pg + q) ^ (row & swz_mask)) << 4)) = xs[n][pg];
        }
      }
  };

  // per-lane offsets of the K loop's fragment reads (load_frags) and the wave-uniform row bases of the tiles
  int aoff = f * wstride + ((q ^ (f & 7)) << 4), fh = f + halo, fhu = (f + halo) * urow;
  asm volatile("" : "+v"(aoff), "+v"(fh), "+v"(fhu));
  int tbase[TPW];
#pragma unroll
  for (int n = 0; n < TPW; ++n) tbase[n] = __builtin_amdgcn_readfirstlane((n < ntw ? (wv + R2_WAVES * n) * 16 : 0) * urow);
  load_w(0, 0);
  load_x(0);                                            // u_0 = x_0
  __syncthreads();                                      // zero fill complete before the first u write
  write_u();
  store_w();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int j = 0; j < R2_STEPS; ++j) {
    const bool last = j + 1 == R2_STEPS;
    if (!last) load_x(j + 1);                           // in flight during the first sub-pass
#pragma unroll
    for (int sub = 0; sub < NSUB_MAX; ++sub) {
      if (sub < nsub) {
        const bool more_sub = sub + 1 < nsub;
        if (more_sub) load_w(j, sub + 1);
        else if (!last) load_w(j + 1, 0);
        f32x4 acc[TPW][SUBM];
#pragma unroll
        for (int n = 0; n < TPW; ++n)
#pragma unroll
          for (int m = 0; m < SUBM; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Fragment addresses without per-lane multiplies (they cost the K loop more issue time than its MFMAs): wp is a
        // multiple of 32 here, so the 4 chunks of a k-step lie in ONE tap -- tap = ks / npg and the chunk 4 (ks % npg) + q
        // are kept as wave-uniform counters, and a lane adds its pinned offsets (aoff, fhu, fh) to scalar terms.
        int ks_ld = 0, tap_s = 0, kin_s = 0;
        auto load_frags = [&](bf16x8 (&af)[SUBM], bf16x8 (&bf)[TPW]) {
          const bool kpad = tap_s >= a.ktaps || ks_ld >= nks;   // K padding: weights are zero, read any valid row
          const int shift = kpad ? 0 : (tap_s - ctr) * a.dil;   // the tap's row shift
          const int c4 = kpad ? 0 : 4 * kin_s;                  // this k-step's first chunk among the tap's channels
          const int kw = ks_ld < nks ? ks_ld : 0;               // one step past the end: any valid chunk, never consumed
          const unsigned char* ap = wl + (aoff ^ ((kw & 1) << 6)) + (kw >> 1) * 128;
#pragma unroll
          for (int m = 0; m < SUBM; ++m) af[m] = *(const bf16x8*)(ap + m * 16 * wstride);
          const int t = (fh + shift) & swz_mask;
          const unsigned char* bp = u + fhu + (((q ^ t) ^ c4) << 4) + shift * urow;
#pragma unroll
          for (int n = 0; n < TPW; ++n) bf[n] = *(const bf16x8*)(bp + tbase[n]);
          ++ks_ld;
          if (++kin_s == npg) { kin_s = 0; ++tap_s; }
        };
        auto mma = [&](const bf16x8 (&af)[SUBM], const bf16x8 (&bf)[TPW]) {
#pragma unroll
          for (int n = 0; n < TPW; ++n)
            if (n < ntw) {
#pragma unroll
              for (int m = 0; m < SUBM; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[n][m], 0, 0, 0);
            }
        };
        {
          bf16x8 af0[SUBM], bf0[TPW], af1[SUBM], bf1[TPW];
          load_frags(af0, bf0);
          for (int ks = 0; ks < nks; ks += 2) {
            load_frags(af1, bf1);
            mma(af0, bf0);
            load_frags(af0, bf0);
            mma(af1, bf1);
          }
        }
        // epilogue of the sub-pass: ReLU -> folded BN -> bf16, 8 consecutive channels per lane and frame
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          const int pg = sub * NP + pp;
          const int cb = 32 * pg + 8 * q;                 // first of this lane's 8 channels
          const f32x4 sc0 = *(const f32x4*)(ss + j * 256 + cb), sc1 = *(const f32x4*)(ss + j * 256 + cb + 4);
          const f32x4 sh0 = *(const f32x4*)(ss + j * 256 + 128 + cb), sh1 = *(const f32x4*)(ss + j * 256 + 128 + cb + 4);
#pragma unroll
          for (int n = 0; n < TPW; ++n) {
            if (n < ntw) {
              f32x4 v0 = acc[n][2 * pp], v1 = acc[n][2 * pp + 1];
#pragma unroll
              for (int e = 0; e < 4; ++e) { v0[e] = relu_f32(v0[e]); v1[e] = relu_f32(v1[e]); }
              v0 = v0 * sc0 + sh0;
              v1 = v1 * sc1 + sh1;
              const bf16x8 o = {(bf16)v0[0], (bf16)v0[1], (bf16)v0[2], (bf16)v0[3], (bf16)v1[0], (bf16)v1[1], (bf16)v1[2], (bf16)v1[3]};
              const u32x4 sp = __builtin_bit_cast(u32x4, o);
              const int r = (wv + R2_WAVES * n) * 16 + f;
              if (r < T && cb < wp) {
                if (r >= st_lo && r < st_hi) *(u32x4*)(yb + r * a.ld + j * wp + cb) = sp;
                if (!last) xs[n][pg] = add_bf16x8(xs[n][pg], sp);
              }
            }
          }
        }
        // LDS-only barriers (see res2_chain_kernel): every wave has read this sub-pass's weights (and, behind the last
        // sub-pass, u_j) before they are overwritten
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (more_sub || !last) store_w();
        if (!more_sub && !last) write_u();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
}

int r2_wide_smem_bytes(int t_max, int wp, int halo, int k_pad, int subm) {
  t_max = t_max < R2_MAX_T ? t_max : R2_MAX_T;
  const int u_rows = (t_max + 15) / 16 * 16 + 2 * halo;
  return ((u_rows * wp * 2 + 255) & ~255) + 16 * subm * k_pad * 2 + R2_STEPS * 256 * 4;
}

// m-tiles per sub-pass for this shape: 4 (64 channels) where LDS allows, else 2; 0 = the wide kernel cannot take it
int r2_wide_subm(int t_max, int wp, int halo, int k_pad) {
  if (wp <= 64 || wp > 128 || wp % 32 != 0 || k_pad > 640 || k_pad % 64 != 0) return 0;
  if (wp % 64 == 0 && r2_wide_smem_bytes(t_max, wp, halo, k_pad, 4) <= 160 * 1024) return 4;
  return r2_wide_smem_bytes(t_max, wp, halo, k_pad, 2) <= 160 * 1024 ? 2 : 0;
}

}  // namespace

bool res2_chain_eligible(int dtype, int t_max, int wp, int k_pad, int n_pad) {
  if (dtype != SUGAR_BF16 || t_max < 1 || wp % 8 != 0) return false;
  if (tune_get(TUNE_GENERIC_RES2)) return false;    // A/B switch for the parity tests: force the per-step path
  if (wp > 64)                                      // wide groups: halo unknown here, 16 = 5 taps at dilation 4 ... 9 taps at 2
    return n_pad == 128 && r2_wide_subm(t_max, wp, 16, k_pad) != 0;
  return n_pad == 64 && k_pad <= 64 * 5 && r2_smem_bytes(t_max, wp, 10, k_pad) <= 160 * 1024;
}

int launch_res2_chain(const ConvLayer* steps, const void* d_x, void* d_y, int ld, int wp, const int32_t* d_row_off,
                      int B, int t_max, hipStream_t s) {
  Res2Args a;
  a.x = (const bf16*)d_x; a.y = (bf16*)d_y; a.row_off = d_row_off;
  a.ld = ld; a.wp = wp; a.ktaps = steps[0].ktaps; a.dil = steps[0].dil; a.k_pad = steps[0].k_pad;
  const int n_pad = wp > 64 ? 128 : 64;
  a.n_pad = n_pad;
  for (int j = 0; j < R2_STEPS; ++j) {
    SUGAR_CHECK_ARG(steps[j].k_pad == a.k_pad && steps[j].ktaps == a.ktaps && steps[j].scale && steps[j].n_pad == n_pad,
                    "res2: inconsistent step %d", j);
    a.w[j] = (const bf16*)steps[j].w; a.scale[j] = steps[j].scale; a.shift[j] = steps[j].shift;
  }
  const int halo = ((a.ktaps - 1) / 2) * a.dil;
  a.max_t = R2_MAX_T;
  {                                                      // parity tests: segment utterances the kernel could hold whole
    const int v = (int)tune_get(TUNE_RES2_MAX_T);
    if (v >= 128 && v <= R2_MAX_T) a.max_t = v & ~15;
    else if (v == 0) {
      // Few utterances: one workgroup per utterance leaves most CUs idle (64 x 3 s: a quarter of the chip, and the
      // chain took as long as at 256 x 3 s).  Cut the utterances into time segments until the grid covers the chip;
      // every segment recomputes 7 halo frames of context per side, outputs are bit-identical (segment tests).
      int n_cu = 0;
      if (int rc = device_cu_count(&n_cu)) return rc;
      if (B < n_cu) {
        const int want = (int)ceil_div(n_cu, B);
        int mt = (int)round_up(round_up(ceil_div(t_max, want), 16) + 2 * R2_STEPS * halo, 16);
        mt = mt < 128 ? 128 : mt;
        if (mt < a.max_t && mt < t_max) a.max_t = mt;
      }
    }
  }
  a.seg_inner = (a.max_t - 2 * R2_STEPS * halo) & ~15;
  if (a.seg_inner < 16) {                                // override too small for this halo: ignore it
    a.max_t = R2_MAX_T;
    a.seg_inner = (a.max_t - 2 * R2_STEPS * halo) & ~15;
  }
  SUGAR_CHECK_ARG(a.seg_inner >= 16, "res2: halo %d leaves no room for time segments", halo);
  const int nseg_max = t_max <= a.max_t ? 1 : (int)ceil_div(t_max, a.seg_inner);
  SUGAR_CHECK_ARG(ld % 8 == 0 && ((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_y & 15) == 0 && (int64_t)(R2_MAX_T + 16) * ld < (1ll << 30),
                  "res2: activations must be 16-byte aligned with ld %% 8 == 0 (ld %d)", ld);
  if (wp > 64) {       // (this kernel on 64-channel groups: 0.149 vs 0.129 ms per step for res2_chain_kernel)
    const int subm = r2_wide_subm(t_max, wp, halo, a.k_pad);
    SUGAR_CHECK_ARG(subm != 0 && halo <= 16, "res2: %d-channel groups with k_pad %d / halo %d do not fit the wide kernel", wp, a.k_pad, halo);
    const int wsmem = r2_wide_smem_bytes(t_max, wp, halo, a.k_pad, subm);
    const int wchunks = 16 * subm * (a.k_pad >> 3);
#define R2_WIDE(SM, WR)                                                                                        \
  {                                                                                                            \
    static OptInMask opted{0};                                                                                 \
    if (int rc = lds_opt_in((const void*)res2_wide_kernel<SM, WR>, 160 * 1024, &opted)) return rc;             \
    hipLaunchKernelGGL((res2_wide_kernel<SM, WR>), dim3(B, nseg_max), dim3(512), wsmem, s, a);                 \
  }
    if (subm == 4) { if (wchunks <= 512 * 3) R2_WIDE(4, 3) else R2_WIDE(4, 6) }   // (64 x 384 / 8 = 3 072 chunks at most: LDS)
    else { if (wchunks <= 512 * 3) R2_WIDE(2, 3) else R2_WIDE(2, 5) }
#undef R2_WIDE
    SUGAR_LAUNCH_CHECK();
    return SUGAR_OK;
  }
  const int t_wg = t_max < a.max_t ? t_max : a.max_t;    // (a segment holds <= max_t frames)
  const int smem_db = r2_smem_bytes(t_wg, wp, halo, a.k_pad, true);
  const bool db = smem_db <= 160 * 1024 && a.k_pad <= 192 && !tune_get(TUNE_RES2_NO_DB);   // (5 taps: the DB form spills)
  const int smem = db ? smem_db : r2_smem_bytes(t_wg, wp, halo, a.k_pad);
  SUGAR_CHECK_ARG(smem <= 160 * 1024 && a.k_pad <= 320, "res2: tile does not fit LDS (%d bytes)", smem);
  const int mt = (wp + 15) / 16;
  const bool w3 = a.k_pad <= 192;
#define R2_CASE(M, TP, WR, D)                                                                                  \
  {                                                                                                            \
    static OptInMask opted{0};                                                                                 \
    if (int rc = lds_opt_in((const void*)res2_chain_kernel<M, TP, WR, D>, 160 * 1024, &opted)) return rc;      \
    hipLaunchKernelGGL((res2_chain_kernel<M, TP, WR, D>), dim3(B, nseg_max), dim3(512), smem, s, a);           \
  }
#define R2_CASE_W(M, TP) \
  if (w3 && db) R2_CASE(M, TP, 3, true) else if (w3) R2_CASE(M, TP, 3, false) else R2_CASE(M, TP, 5, false)
  switch (mt) {
    case 1: R2_CASE_W(1, 3) break;
    case 2: R2_CASE_W(2, 3) break;
    case 3: R2_CASE_W(3, 3) break;
    default: R2_CASE_W(4, 3) break;
  }
#undef R2_CASE_W
#undef R2_CASE
  SUGAR_LAUNCH_CHECK();
  return SUGAR_OK;
}

}  // namespace sugar
