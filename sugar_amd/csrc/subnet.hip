// Host side of libsugar_amd: packed subnets, workspace planning, the forward schedule and the C-ABI.
//
// The schedule is DynamicTDNN.forward (sugar/models/dynamictdnn.py:645-666) on packed channels-last rows:
//   layer1 -> [first_conv -> 7 Res2 steps -> last_conv -> SE scale -> (scale, +S, S+=)] x depth
//          -> catconv+ReLU -> pooling.linear1+tanh -> fused linear2/softmax/statistics -> folded tail.
// torch.cat (:662) never materialises: every block writes its output straight into its column slice of the
// catconv input; the Res2 split/cat (:291,302) are column offsets into two [R][mid] buffers.
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace sugar {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fbank_init();

// ---- optional hipEvent bracketing of the launches of one stage (bench.py's live per-kernel timing) ----
static const char* kStageNames[SUGAR_STAGE_COUNT] = {"fbank_dft+instnorm", "layer1", "first_conv", "res2_steps", "last_conv",
                                                     "se_scale", "se_apply", "catconv", "pool_linear1", "pool_stats",
                                                     "tail+l2norm", "rowseg"};
struct ProfRec { int stage; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static std::atomic<unsigned> g_prof_mask{0};
static unsigned g_prof_gen = 0;              // bumped by sugar_profile_enable: brackets opened before it are dropped
static std::vector<ProfRec> g_prof_recs;     // recorded pairs
static std::vector<hipEvent_t> g_prof_pool;  // reusable events

static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

struct Prof {
  hipStream_t s;
  ProfRec r{-1, nullptr, nullptr};
  unsigned gen = 0;
  Prof(int stage, hipStream_t s_) : s(s_) {
    if (!(g_prof_mask.load(std::memory_order_relaxed) >> stage & 1u)) return;
    {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      r.a = prof_event();
      r.b = prof_event();
      gen = g_prof_gen;
    }
    if (!r.a || !r.b) return;
    r.stage = stage;
    (void)hipEventRecord(r.a, s);
  }
  ~Prof() {
    if (r.stage < 0) return;
    (void)hipEventRecord(r.b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (gen == g_prof_gen) g_prof_recs.push_back(r);      // the record list was not cleared in between
    else { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  }
};

// ---- tunables: environment read once, then atomics ----
static const char* kTuneNames[TUNE_COUNT] = {"gemm_wreg_min_r", "gemm_uniform", "gemm_no_pingpong", "generic_res2",
                                             "res2_max_t", "generic_layer1", "gemm_no_persist", "gemm_no_quad",
                                             "res2_no_db", "recal_separate_stats"};
static const int64_t kTuneDefaults[TUNE_COUNT] = {19200, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static std::atomic<int64_t> g_tune[TUNE_COUNT];
static std::once_flag g_tune_once;

static void tune_init() {
  std::call_once(g_tune_once, [] {
    for (int k = 0; k < TUNE_COUNT; ++k) {
      char env[64] = "SUGAR_";
      size_t n = strlen(env);
      for (const char* c = kTuneNames[k]; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
      env[n] = 0;
      const char* v = getenv(env);
      // a switch set to the empty string counts as "on", as the getenv() != nullptr tests of round 1 did
      g_tune[k].store(v ? (*v ? atoll(v) : 1) : kTuneDefaults[k], std::memory_order_relaxed);
    }
  });
}

int64_t tune_get(int key) {
  tune_init();
  return g_tune[key].load(std::memory_order_relaxed);
}

int device_cu_count(int* n_cu) {
  static std::atomic<int> cache[64];
  int dev = 0;
  SUGAR_HIP(hipGetDevice(&dev));
  int v = dev >= 0 && dev < 64 ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (!v) {
    SUGAR_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) cache[dev].store(v, std::memory_order_relaxed);
  }
  *n_cu = v;
  return SUGAR_OK;
}

struct Block {
  ConvLayer first, res[SUGAR_RES2_SCALE - 1], last;
  SELayer se;
  BNState bn_first, bn_res[SUGAR_RES2_SCALE - 1], bn_last;   // raw BN parameters + running statistics (recalibration)
  int w = 0, wp = 0, midp = 0, kernel = 1;
};

}  // namespace sugar

struct SugarSubnet {
  int dtype = SUGAR_BF16, device = 0;
  int in_feats = 80, embed = 192, depth = 0, c0 = 0, c0p = 0, ccat = 0, ccatp = 0, bott = 128, midp_max = 0;
  sugar::ConvLayer layer1, catconv, pool1;
  sugar::Block blocks[SUGAR_MAX_DEPTH];
  sugar::PoolLayer pool;
  sugar::TailLayer tail;
  // BN recalibration: raw BN state of every BatchNorm, the unfolded linear layer, host copies for the tail refold
  sugar::BNState bn_layer1, bn_pool, bn_embed;
  sugar::TailLayer tail_raw;
  std::vector<float> h_linear_w, h_linear_b;
  sugar::DevArena arena;     // every packed tensor above lives in its chunks (arena.hip)
};

namespace sugar {

static int upload(float** dst, const std::vector<float>& v) {
  if (int rc = dev_alloc(dst, v.size() * sizeof(float))) return rc;
  return dev_upload(*dst, v.data(), v.size() * sizeof(float));
}

static bool bn_ok(const SugarBN& b) { return b.weight && b.bias && b.running_mean && b.running_var; }

static void destroy(SugarSubnet* n) {
  {
    ArenaScope scope(&n->arena);                       // the frees below skip what the arena owns
    free_conv(&n->layer1);
    free_conv(&n->catconv);
    free_conv(&n->pool1);
    for (int i = 0; i < SUGAR_MAX_DEPTH; ++i) {
      Block& b = n->blocks[i];
      free_conv(&b.first);
      free_conv(&b.last);
      for (auto& r : b.res) free_conv(&r);
      dev_free(b.se.w1t);
      dev_free(b.se.b1);
      dev_free(b.se.w2t);
      dev_free(b.se.b2);
      bn_state_free(&b.bn_first);
      bn_state_free(&b.bn_last);
      for (auto& r : b.bn_res) bn_state_free(&r);
    }
    dev_free(n->pool.w2);
    dev_free(n->tail.wt);
    dev_free(n->tail.bias);
    dev_free(n->tail_raw.wt);
    dev_free(n->tail_raw.bias);
    bn_state_free(&n->bn_layer1);
    bn_state_free(&n->bn_pool);
    bn_state_free(&n->bn_embed);
  }
  arena_release(&n->arena);
  delete n;
}

// tail: bnembed(linear(bnpool(p))) = W' p + b'  folded in double               dynamictdnn.py:664-665
// BN arrays in reference channel order (gamma, beta, running_mean, running_var); the first call allocates.
static int fold_tail(SugarSubnet* n, const float* const pool_bn[4], const float* const embed_bn[4]) {
  const int in = 2 * n->ccat, inp = 2 * n->ccatp, E_ = n->embed;
  std::vector<float> wt((size_t)inp * E_, 0.f), bias(E_);
  std::vector<double> sp(in), tp(in);
  for (int i = 0; i < in; ++i) {
    sp[i] = (double)pool_bn[0][i] / sqrt((double)pool_bn[3][i] + 1e-5);
    tp[i] = (double)pool_bn[1][i] - (double)pool_bn[2][i] * sp[i];
  }
  for (int o = 0; o < E_; ++o) {
    const double se = (double)embed_bn[0][o] / sqrt((double)embed_bn[3][o] + 1e-5);
    const double te = (double)embed_bn[1][o] - (double)embed_bn[2][o] * se;
    double acc = n->h_linear_b[o];
    for (int i = 0; i < in; ++i) {
      const double wv = n->h_linear_w[(size_t)o * in + i];
      acc += wv * tp[i];
      const int pi = i < n->ccat ? i : n->ccatp + (i - n->ccat);
      wt[(size_t)pi * E_ + o] = (float)(se * wv * sp[i]);
    }
    bias[o] = (float)(se * acc + te);
  }
  if (!n->tail.wt) {
    int rc;
    if ((rc = upload(&n->tail.wt, wt)) || (rc = upload(&n->tail.bias, bias))) return rc;
  } else {
    SUGAR_HIP(hipMemcpy(n->tail.wt, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
    SUGAR_HIP(hipMemcpy(n->tail.bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  n->tail.in_p = inp;
  n->tail.embed = E_;
  return SUGAR_OK;
}

static int create(const SugarSubnetDesc* d, int dtype, SugarSubnet* n) {
  const int E = dtype == SUGAR_F32 ? 4 : 8;
  n->dtype = dtype;
  n->in_feats = d->in_feats;
  n->embed = d->embed_dim;
  n->depth = d->depth;
  n->c0 = d->c0;
  n->c0p = (int)round_up(d->c0, 8);
  n->ccat = d->ccat;
  n->ccatp = (int)round_up(d->ccat, 8);
  n->bott = d->bottleneck;
  int rc;

  // layer1: (in_feats, C0, K0), dilation 1                                         dynamictdnn.py:628
  if ((rc = pack_conv(&n->layer1, d->layer1_w, &d->layer1_bn, d->c0, d->in_feats, d->k0, 1, EPI_RELU, n->c0p, nullptr,
                      d->in_feats, nullptr, dtype)) ||
      (rc = bn_state_create(&n->bn_layer1, &d->layer1_bn, d->c0, n->c0p, nullptr)))
    return rc;

  for (int i = 0; i < d->depth; ++i) {
    const SugarBlock& sb = d->blocks[i];
    Block& b = n->blocks[i];
    SUGAR_CHECK_ARG(sb.mid_channels > 0 && sb.mid_channels % SUGAR_RES2_SCALE == 0,
                    "block %d: mid_channels %d must be a positive multiple of 8", i, sb.mid_channels);
    SUGAR_CHECK_ARG(sb.first_w && sb.last_w && bn_ok(sb.first_bn) && bn_ok(sb.last_bn) && sb.se_reduce_w &&
                        sb.se_reduce_b && sb.se_expand_w && sb.se_expand_b,
                    "block %d: null tensor", i);
    b.kernel = sb.kernel;
    b.w = sb.mid_channels / SUGAR_RES2_SCALE;
    b.wp = (int)round_up(b.w, 8);
    b.midp = b.wp * SUGAR_RES2_SCALE;
    if (b.midp > n->midp_max) n->midp_max = b.midp;
    const int dil = i + 2;                                                         // dynamictdnn.py:629-632
    std::vector<int> midmap(sb.mid_channels);
    for (int g = 0; g < SUGAR_RES2_SCALE; ++g)
      for (int j = 0; j < b.w; ++j) midmap[g * b.w + j] = g * b.wp + j;
    if ((rc = pack_conv(&b.first, sb.first_w, &sb.first_bn, sb.mid_channels, d->c0, 1, 1, EPI_RELU, b.midp,
                        midmap.data(), n->c0p, nullptr, dtype)) ||
        (rc = bn_state_create(&b.bn_first, &sb.first_bn, sb.mid_channels, b.midp, midmap.data())))
      return rc;
    for (int j = 0; j < SUGAR_RES2_SCALE - 1; ++j) {
      SUGAR_CHECK_ARG(sb.res_w[j] && bn_ok(sb.res_bn[j]), "block %d res2 conv %d: null tensor", i, j);
      if ((rc = pack_conv(&b.res[j], sb.res_w[j], &sb.res_bn[j], b.w, b.w, sb.kernel, dil, EPI_RELU, b.wp, nullptr,
                          b.wp, nullptr, dtype)) ||
          (rc = bn_state_create(&b.bn_res[j], &sb.res_bn[j], b.w, b.wp, nullptr)))
        return rc;
    }
    if ((rc = pack_conv(&b.last, sb.last_w, &sb.last_bn, d->c0, sb.mid_channels, 1, 1, EPI_RELU, n->c0p, nullptr,
                        b.midp, midmap.data(), dtype)) ||
        (rc = bn_state_create(&b.bn_last, &sb.last_bn, d->c0, n->c0p, nullptr)))
      return rc;
    // SE: reduce [hid][c0], expand [c0][hid]  -> transposed, channel-padded             dynamictdnn.py:88-89
    const int hid = d->c0 / 4;
    SUGAR_CHECK_ARG(hid > 0, "block %d: c0 %d too small for SE reduction 4", i, d->c0);
    b.se.c = d->c0;
    b.se.c_p = n->c0p;
    b.se.hid = hid;
    std::vector<float> w1t((size_t)n->c0p * hid, 0.f), b1(hid), w2t((size_t)hid * n->c0p, 0.f), b2(n->c0p, 0.f);
    for (int j = 0; j < hid; ++j) {
      b1[j] = sb.se_reduce_b[j];
      for (int c = 0; c < d->c0; ++c) {
        w1t[(size_t)c * hid + j] = sb.se_reduce_w[(size_t)j * d->c0 + c];
        w2t[(size_t)j * n->c0p + c] = sb.se_expand_w[(size_t)c * hid + j];
      }
    }
    for (int c = 0; c < d->c0; ++c) b2[c] = sb.se_expand_b[c];
    if ((rc = upload(&b.se.w1t, w1t)) || (rc = upload(&b.se.b1, b1)) || (rc = upload(&b.se.w2t, w2t)) ||
        (rc = upload(&b.se.b2, b2)))
      return rc;
  }

  // catconv: input = cat of block outputs, each block padded to c0p                  dynamictdnn.py:633,662-663
  {
    std::vector<int> catmap((size_t)d->depth * d->c0);
    for (int blk = 0; blk < d->depth; ++blk)
      for (int j = 0; j < d->c0; ++j) catmap[(size_t)blk * d->c0 + j] = blk * n->c0p + j;
    if ((rc = pack_conv(&n->catconv, d->catconv_w, nullptr, d->ccat, d->depth * d->c0, 1, 1, EPI_RELU, n->ccatp,
                        nullptr, d->depth * n->c0p, catmap.data(), dtype)))
      return rc;
  }
  // pooling.linear1 (+tanh) as a conv; pooling.linear2 as the fused kernel's B operand  dynamictdnn.py:498-499
  if ((rc = pack_conv(&n->pool1, d->pool1_w, nullptr, d->bottleneck, d->ccat, 1, 1, EPI_TANH, d->bottleneck, nullptr,
                      n->ccatp, nullptr, dtype)))
    return rc;
  {
    const int rows = (int)round_up(n->ccatp, 256);   // pool.hip: 256 channels per workgroup
    const size_t cnt = (size_t)rows * d->bottleneck;
    const size_t esz = dtype == SUGAR_F32 ? 4 : 2;
    std::vector<unsigned char> host(cnt * esz, 0);
    for (int c = 0; c < d->ccat; ++c)
      for (int k = 0; k < d->bottleneck; ++k) {
        const float v = d->pool2_w[(size_t)c * d->bottleneck + k];
        if (dtype == SUGAR_F32) ((float*)host.data())[(size_t)c * d->bottleneck + k] = v;
        else ((uint16_t*)host.data())[(size_t)c * d->bottleneck + k] = host_f32_to_bf16(v);
      }
    if ((rc = dev_alloc(&n->pool.w2, cnt * esz)) || (rc = dev_upload(n->pool.w2, host.data(), cnt * esz))) return rc;
    n->pool.ccat = d->ccat;
    n->pool.ccat_p = n->ccatp;
    n->pool.bott = d->bottleneck;
  }
  // tail: folded affine for eval, raw linear + BN states for recalibration
  {
    const int in = 2 * d->ccat, inp = 2 * n->ccatp, E_ = d->embed_dim;
    n->h_linear_w.assign(d->linear_w, d->linear_w + (size_t)E_ * in);
    n->h_linear_b.assign(d->linear_b, d->linear_b + E_);
    const float* const pool_bn[4] = {d->bnpool.weight, d->bnpool.bias, d->bnpool.running_mean, d->bnpool.running_var};
    const float* const embed_bn[4] = {d->bnembed.weight, d->bnembed.bias, d->bnembed.running_mean, d->bnembed.running_var};
    if ((rc = fold_tail(n, pool_bn, embed_bn))) return rc;
    std::vector<int> poolmap(in);
    for (int i = 0; i < in; ++i) poolmap[i] = i < d->ccat ? i : n->ccatp + (i - d->ccat);
    if ((rc = bn_state_create(&n->bn_pool, &d->bnpool, in, inp, poolmap.data())) ||
        (rc = bn_state_create(&n->bn_embed, &d->bnembed, E_, E_, nullptr)))
      return rc;
    std::vector<float> wraw((size_t)inp * E_, 0.f);
    for (int o = 0; o < E_; ++o)
      for (int i = 0; i < in; ++i) wraw[(size_t)poolmap[i] * E_ + o] = d->linear_w[(size_t)o * in + i];
    if ((rc = upload(&n->tail_raw.wt, wraw)) || (rc = upload(&n->tail_raw.bias, n->h_linear_b))) return rc;
    n->tail_raw.in_p = inp;
    n->tail_raw.embed = E_;
  }
  (void)E;
  return fbank_init();
}

// ---------------------------------------------------------------------------------- workspace
struct Workspace {
  int2* rowseg;
  float* logmel;
  void *feats, *featp, *S, *cat, *mid1, *mid2, *ybuf, *hbuf, *abuf;
  int32_t* orow;     // layer1 tap GEMM: output-row map of the padded row space
  float *separt, *sescale, *pooled, *poolpart, *tailpart;
  double* bnpart;    // recalibration: partial sums, batch (scale, shift), pre-BN embeddings
  float *bnscale, *bnshift, *embpre;
  int64_t bytes;
};

static Workspace plan(const SugarSubnet* n, int B, int64_t R, void* base) {
  const int64_t esz = n->dtype == SUGAR_F32 ? 4 : 2;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    void* p = base ? (void*)((char*)base + off) : nullptr;
    off += round_up(bytes, 256);
    return p;
  };
  Workspace w;
  w.rowseg = (int2*)take(R * (int64_t)sizeof(int2));
  w.logmel = (float*)take(fbank_scratch_floats(B, R) * 4);
  w.feats = take(R * SUGAR_N_MELS * esz);
  w.featp = take(gemm_taps_pad_rows(n->layer1, B, R) * n->in_feats * esz);   // zero-padded copy for the layer1 GEMM
  w.orow = (int32_t*)take(gemm_taps_orow_ints(n->layer1, B, R) * 4);
  w.S = take(R * n->c0p * esz);
  w.cat = take(R * (int64_t)n->depth * n->c0p * esz);
  w.mid1 = take(R * n->midp_max * esz);
  w.mid2 = take(R * n->midp_max * esz);
  w.ybuf = take(R * n->c0p * esz);
  w.hbuf = take(R * n->ccatp * esz);
  w.abuf = take(R * n->bott * esz);
  w.separt = (float*)take(se_part_floats(n->blocks[0].se, n->dtype, B, R) * 4);   // every block: same c0 / hidden
  w.sescale = (float*)take((int64_t)B * n->c0p * 4);
  w.pooled = (float*)take((int64_t)B * 2 * n->ccatp * 4);
  w.poolpart = (float*)take(pool_part_floats(n->pool, B, R) * 4);
  w.tailpart = (float*)take((int64_t)tail_splits(n->tail) * B * n->embed * 4);
  const int cbn = n->c0p > n->midp_max ? n->c0p : n->midp_max;
  w.bnpart = (double*)take(bn_workspace_doubles(cbn, R) * 8);
  w.bnscale = (float*)take((int64_t)cbn * 4);
  w.bnshift = (float*)take((int64_t)cbn * 4);
  w.embpre = (float*)take((int64_t)B * n->embed * 4);
  w.bytes = off;
  return w;
}

// ---------------------------------------------------------------------------------- forward
// Does layer1 run as the tap GEMM over the zero-padded feature copy (w.featp)?
static bool layer1_as_tap_gemm(const SugarSubnet* n) {
  return !tune_get(TUNE_GENERIC_LAYER1) && gemm_taps_eligible(n->layer1, n->in_feats, n->dtype);
}

// feats == nullptr: the front-end has already written the padded copy w.featp (sugar_embed_fwd).
// l1_taps: layer1_as_tap_gemm(n), read ONCE per call by the entry point (a concurrent sugar_tune_set may not land
// between the front-end's choice of layout and this function's choice of kernel).
static int tdnn_forward(const SugarSubnet* n, const void* feats, const int32_t* row_off, int B, int64_t R, int t_max,
                        const Workspace& w, float* d_emb, int normalize, int tap, float* tap_out, hipStream_t s,
                        bool l1_taps) {
  const int dt = n->dtype;
  const int64_t esz = dt == SUGAR_F32 ? 4 : 2;
  int rc;
  auto col = [&](void* p, int64_t c) { return (void*)((char*)p + c * esz); };
  auto tapcopy = [&](int id, const void* p, int ld, int C) -> int {
    if (tap != id || !tap_out) return SUGAR_OK;
    return launch_to_f32(p, ld, R, C, tap_out, dt, s);
  };
  // The per-row (start, end) table is what the generic masked-load conv reads; the LDS-DMA GEMMs and the fused Res2
  // chain do not.  It is built at the first launch that needs it: the bf16 path of a full-width subnet never does.
  bool rowseg_built = false;
  auto conv = [&](const ConvLayer& L, const ConvIO& cio) -> int {
    if (!rowseg_built && !gemm1x1_eligible(L, cio, dt)) {
      Prof p(SUGAR_STAGE_ROWSEG, s);
      if (int rc2 = launch_build_rowseg(row_off, B, t_max, w.rowseg, s)) return rc2;
      rowseg_built = true;
    }
    return launch_conv(L, cio, w.rowseg, R, dt, s);
  };

  ConvIO io;
  io.x = feats; io.ldx = n->in_feats; io.y = w.S; io.ldy = n->c0p;
  {
    Prof p(SUGAR_STAGE_LAYER1, s);
    if (l1_taps) {
      if (feats && (rc = launch_pad_rows(n->layer1, feats, w.featp, row_off, B, R, t_max, s))) return rc;
      // feats == nullptr: the fused front-end has written the padded layout AND its output-row map
      rc = launch_gemm_taps(n->layer1, w.featp, w.orow, w.S, n->c0p, row_off, B, R, t_max, s, feats == nullptr);
    } else {
      SUGAR_CHECK_ARG(feats, "forward: features missing");
      rc = conv(n->layer1, io);
    }
    if (rc) return rc;
  }
  if ((rc = tapcopy(0, w.S, n->c0p, n->c0p))) return rc;

  const int ldcat = n->depth * n->c0p;
  for (int i = 0; i < n->depth; ++i) {
    const Block& b = n->blocks[i];
    // block input: layer1 output (still == S) for block 0, else the previous block's slice of the cat buffer
    const void* xin = i == 0 ? w.S : col(w.cat, (int64_t)(i - 1) * n->c0p);
    const int ldin = i == 0 ? n->c0p : ldcat;
    ConvIO f;
    f.x = xin; f.ldx = ldin; f.y = w.mid1; f.ldy = b.midp;
    f.y2 = w.mid2; f.ldy2 = b.midp; f.y2_from = (SUGAR_RES2_SCALE - 1) * b.wp;   // 8th split passes through (:300-301)
    { Prof p(SUGAR_STAGE_FIRST, s); if ((rc = conv(b.first, f))) return rc; }
    if ((rc = tapcopy(1 + 4 * i, w.mid1, b.midp, b.midp))) return rc;
    {
    Prof p(SUGAR_STAGE_RES2, s);
    if (res2_chain_eligible(dt, t_max, b.wp, b.res[0].k_pad, b.res[0].n_pad)) {   // whole chain in one kernel
      if ((rc = launch_res2_chain(b.res, w.mid1, w.mid2, b.midp, b.wp, row_off, B, t_max, s))) return rc;
    } else
    for (int j = 0; j < SUGAR_RES2_SCALE - 1; ++j) {                              // dynamictdnn.py:294-299
      ConvIO r;
      r.x = col(w.mid1, (int64_t)j * b.wp); r.ldx = b.midp;
      if (j > 0) { r.x2 = col(w.mid2, (int64_t)(j - 1) * b.wp); r.ldx2 = b.midp; }
      r.y = col(w.mid2, (int64_t)j * b.wp); r.ldy = b.midp;
      if ((rc = conv(b.res[j], r))) return rc;
    }
    }
    if ((rc = tapcopy(2 + 4 * i, w.mid2, b.midp, b.midp))) return rc;
    ConvIO l;
    l.x = w.mid2; l.ldx = b.midp; l.y = w.ybuf; l.ldy = n->c0p;
    { Prof p(SUGAR_STAGE_LAST, s); if ((rc = conv(b.last, l))) return rc; }
    if ((rc = tapcopy(3 + 4 * i, w.ybuf, n->c0p, n->c0p))) return rc;
    { Prof p(SUGAR_STAGE_SE_SCALE, s); if ((rc = launch_se_scale(b.se, w.ybuf, n->c0p, row_off, B, t_max, w.separt, w.sescale, dt, s))) return rc; }
    void* out = col(w.cat, (int64_t)i * n->c0p);
    {
      Prof p(SUGAR_STAGE_SE_APPLY, s);
      if ((rc = launch_se_apply(w.ybuf, n->c0p, w.sescale, n->c0p, w.S, n->c0p, out, ldcat, i + 1 < n->depth, row_off, B,
                                t_max, dt, s)))
        return rc;
    }
    if ((rc = tapcopy(4 + 4 * i, out, ldcat, n->c0p))) return rc;
  }
  ConvIO c;
  c.x = w.cat; c.ldx = ldcat; c.y = w.hbuf; c.ldy = n->ccatp;
  { Prof p(SUGAR_STAGE_CATCONV, s); if ((rc = conv(n->catconv, c))) return rc; }
  if ((rc = tapcopy(100, w.hbuf, n->ccatp, n->ccatp))) return rc;
  ConvIO p1;
  p1.x = w.hbuf; p1.ldx = n->ccatp; p1.y = w.abuf; p1.ldy = n->bott;
  { Prof p(SUGAR_STAGE_POOL1, s); if ((rc = conv(n->pool1, p1))) return rc; }
  if ((rc = tapcopy(102, w.abuf, n->bott, n->bott))) return rc;
  { Prof p(SUGAR_STAGE_POOL_STATS, s); if ((rc = launch_pool_stats(n->pool, w.abuf, n->bott, w.hbuf, n->ccatp, row_off, B, t_max, w.pooled, w.poolpart, dt, s))) return rc; }
  if (tap == 101 && tap_out)
    SUGAR_HIP(hipMemcpyAsync(tap_out, w.pooled, (size_t)B * 2 * n->ccatp * 4, hipMemcpyDeviceToDevice, s));
  Prof p(SUGAR_STAGE_TAIL, s);
  if ((rc = launch_tail(n->tail, w.pooled, B, w.tailpart, d_emb, normalize, s))) return rc;
  return SUGAR_OK;
}

// ---------------------------------------------------------------------------------- BN recalibration
// One train-mode forward (batch_forward, sugar/models/__init__.py:201-242): same schedule as tdnn_forward, but every
// conv -> ReLU -> BN runs as conv+ReLU (affine off), then BatchNorm with the statistics of THIS batch, in place, and
// the BN's running statistics move by `momentum`.  The Res2 chain runs step by step (each step's BN needs a
// reduction over all utterances), the tail unfolded (bnpool -> linear -> bnembed).  d_emb receives the train-mode
// embeddings (what the reference's forward returns in train()).
static ConvLayer raw_conv(const ConvLayer& L) {
  ConvLayer r = L;
  r.flags = L.flags & ~EPI_AFFINE;
  r.scale = r.shift = nullptr;
  return r;
}

static int tdnn_recalibrate(SugarSubnet* n, const void* feats, const int32_t* row_off, int B, int64_t R, int t_max,
                            const Workspace& w, float momentum, float* d_emb, hipStream_t s) {
  const int dt = n->dtype;
  const int64_t esz = dt == SUGAR_F32 ? 4 : 2;
  int rc;
  auto col = [&](void* p, int64_t c) { return (void*)((char*)p + c * esz); };
  // conv -> ReLU, then train-mode BN of its output in place.  The generic conv kernel and the weights-in-registers
  // GEMM leave the BatchNorm's statistics partials behind (ConvIO::stat_part) and the BN skips its own pass over the
  // tensor; the tile GEMMs (short batches, 1024-channel convs) do not, and their BN runs bn_stats_kernel as before.
  const bool fuse_stats = !tune_get(TUNE_RECAL_SEPARATE_STATS);
  const int cbn = n->c0p > n->midp_max ? n->c0p : n->midp_max;
  const int64_t stat_cap = bn_workspace_doubles(cbn, R);
  auto conv_bn = [&](const ConvLayer& L, ConvIO io, int C, const BNState& st, bool apply = true) {
    int blocks = 0;
    if (fuse_stats) { io.stat_part = w.bnpart; io.stat_c = C; io.stat_cap = stat_cap; }
    if (int rc2 = launch_conv(raw_conv(L), io, w.rowseg, R, dt, s, &blocks)) return rc2;
    return launch_bn_batch(io.y, io.ldy, R, C, st, momentum, w.bnpart, w.bnscale, w.bnshift, dt, s, blocks, apply);
  };
  if ((rc = launch_build_rowseg(row_off, B, t_max, w.rowseg, s))) return rc;

  {
    const ConvLayer l1 = raw_conv(n->layer1);
    if (layer1_as_tap_gemm(n)) {                      // the eval path's tap GEMM with the affine switched off
      int64_t stat_io = stat_cap;
      if ((rc = launch_pad_rows(l1, feats, w.featp, row_off, B, R, t_max, s)) ||
          (rc = launch_gemm_taps(l1, w.featp, w.orow, w.S, n->c0p, row_off, B, R, t_max, s, false,
                                 fuse_stats ? w.bnpart : nullptr, n->c0p, fuse_stats ? &stat_io : nullptr)) ||
          (rc = launch_bn_batch(w.S, n->c0p, R, n->c0p, n->bn_layer1, momentum, w.bnpart, w.bnscale, w.bnshift, dt, s,
                                fuse_stats ? (int)stat_io : 0)))
        return rc;
    } else {
      ConvIO io;
      io.x = feats; io.ldx = n->in_feats; io.y = w.S; io.ldy = n->c0p;
      if ((rc = conv_bn(n->layer1, io, n->c0p, n->bn_layer1))) return rc;
    }
  }

  const int ldcat = n->depth * n->c0p;
  for (int i = 0; i < n->depth; ++i) {
    const Block& b = n->blocks[i];
    const void* xin = i == 0 ? w.S : col(w.cat, (int64_t)(i - 1) * n->c0p);
    const int ldin = i == 0 ? n->c0p : ldcat;
    ConvIO f;
    f.x = xin; f.ldx = ldin; f.y = w.mid1; f.ldy = b.midp;
    if ((rc = conv_bn(b.first, f, b.midp, b.bn_first))) return rc;
    for (int j = 0; j < SUGAR_RES2_SCALE - 1; ++j) {                              // dynamictdnn.py:294-299
      ConvIO r;
      r.x = col(w.mid1, (int64_t)j * b.wp); r.ldx = b.midp;
      if (j > 0) { r.x2 = col(w.mid2, (int64_t)(j - 1) * b.wp); r.ldx2 = b.midp; }
      r.y = col(w.mid2, (int64_t)j * b.wp); r.ldy = b.midp;
      if ((rc = conv_bn(b.res[j], r, b.wp, b.bn_res[j]))) return rc;
    }
    // the 8th split passes through (:300-301): the normalised first_conv columns
    const int64_t c7 = (int64_t)(SUGAR_RES2_SCALE - 1) * b.wp;
    SUGAR_HIP(hipMemcpy2DAsync(col(w.mid2, c7), (size_t)b.midp * esz, col(w.mid1, c7), (size_t)b.midp * esz,
                               (size_t)b.wp * esz, (size_t)R, hipMemcpyDeviceToDevice, s));
    ConvIO l;
    l.x = w.mid2; l.ldx = b.midp; l.y = w.ybuf; l.ldy = n->c0p;
    // last_conv's BatchNorm is not applied in a pass of its own: its consumers are the SE mean (linear: the affine goes
    // on the mean) and se_apply (which reads every row anyway); w.bnscale / w.bnshift hold this batch's affine until
    // the next BatchNorm's finalize, i.e. past both
    const bool fold = fuse_stats;
    if ((rc = conv_bn(b.last, l, n->c0p, b.bn_last, !fold))) return rc;
    const float* bsc = fold ? w.bnscale : nullptr;
    const float* bsh = fold ? w.bnshift : nullptr;
    if ((rc = launch_se_scale(b.se, w.ybuf, n->c0p, row_off, B, t_max, w.separt, w.sescale, dt, s, bsc, bsh))) return rc;
    void* out = col(w.cat, (int64_t)i * n->c0p);
    if ((rc = launch_se_apply(w.ybuf, n->c0p, w.sescale, n->c0p, w.S, n->c0p, out, ldcat, i + 1 < n->depth, row_off, B,
                              t_max, dt, s, bsc, bsh)))
      return rc;
  }
  ConvIO c;
  c.x = w.cat; c.ldx = ldcat; c.y = w.hbuf; c.ldy = n->ccatp;
  if ((rc = launch_conv(n->catconv, c, w.rowseg, R, dt, s))) return rc;
  ConvIO p1;
  p1.x = w.hbuf; p1.ldx = n->ccatp; p1.y = w.abuf; p1.ldy = n->bott;
  if ((rc = launch_conv(n->pool1, p1, w.rowseg, R, dt, s))) return rc;
  if ((rc = launch_pool_stats(n->pool, w.abuf, n->bott, w.hbuf, n->ccatp, row_off, B, t_max, w.pooled, w.poolpart, dt, s))) return rc;
  // unfolded tail: bnpool (statistics over the B utterances) -> linear -> bnembed        dynamictdnn.py:664-665
  if ((rc = launch_bn_rows(w.pooled, 2 * n->ccatp, B, n->bn_pool, momentum, s))) return rc;
  if ((rc = launch_tail(n->tail_raw, w.pooled, B, w.tailpart, w.embpre, 0, s))) return rc;
  if ((rc = launch_bn_rows(w.embpre, n->embed, B, n->bn_embed, momentum, s))) return rc;
  if (d_emb) SUGAR_HIP(hipMemcpyAsync(d_emb, w.embpre, (size_t)B * n->embed * 4, hipMemcpyDeviceToDevice, s));
  return SUGAR_OK;
}

// the BN layers in the order of the C-ABI's `bn_index`: 0 layer1; 1 + 9 i + {0 first, 1..7 res2 step, 8 last};
// 1 + 9 depth: bnpool; 2 + 9 depth: bnembed
static BNState* bn_by_index(SugarSubnet* n, int idx, ConvLayer** conv) {
  *conv = nullptr;
  if (idx == 0) { *conv = &n->layer1; return &n->bn_layer1; }
  const int last = 1 + 9 * n->depth;
  if (idx == last) return &n->bn_pool;
  if (idx == last + 1) return &n->bn_embed;
  if (idx < 0 || idx > last + 1) return nullptr;
  Block& b = n->blocks[(idx - 1) / 9];
  const int k = (idx - 1) % 9;
  if (k == 0) { *conv = &b.first; return &b.bn_first; }
  if (k == 8) { *conv = &b.last; return &b.bn_last; }
  *conv = &b.res[k - 1];
  return &b.bn_res[k - 1];
}

static int bn_download(const BNState& st, std::vector<float> out[4]) {
  std::vector<float> packed(4 * (size_t)st.n);
  SUGAR_HIP(hipMemcpy(packed.data(), st.p, packed.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int k = 0; k < 4; ++k) {
    out[k].resize(st.c);
    for (int o = 0; o < st.c; ++o) out[k][o] = packed[(size_t)k * st.n + st.h_map[o]];
  }
  return SUGAR_OK;
}

static int check_fwd_args(const SugarSubnet* n, const void* in, const int32_t* row_off, int B, int64_t R, int t_max,
                          void* ws, int64_t ws_bytes, float* emb) {
  SUGAR_CHECK_ARG(n && in && row_off && ws && emb, "forward: null pointer");
  SUGAR_CHECK_ARG(B > 0 && B <= 65535, "forward: batch %d out of range (1..65535)", B);
  SUGAR_CHECK_ARG(R >= B && R < (1ll << 27), "forward: total frames %lld out of range", (long long)R);
  SUGAR_CHECK_ARG(t_max > 0 && t_max <= R, "forward: t_max %d inconsistent with R %lld", t_max, (long long)R);
  SUGAR_CHECK_ARG(((uintptr_t)ws & 255) == 0, "forward: workspace must be 256-byte aligned");
  const int64_t need = plan(n, B, R, nullptr).bytes;
  if (ws_bytes < need) {
    set_error("forward: workspace %lld bytes < required %lld", (long long)ws_bytes, (long long)need);
    return SUGAR_ENOMEM;
  }
  return SUGAR_OK;
}

}  // namespace sugar

using namespace sugar;

extern "C" {

const char* sugar_last_error(void) { return g_err; }
const char* sugar_version(void) { return "sugar_amd 0.1 (gfx950)"; }

int sugar_subnet_create(const SugarSubnetDesc* d, int dtype, int device, SugarSubnet** out) {
  SUGAR_CHECK_ARG(d && out, "subnet_create: null pointer");
  SUGAR_CHECK_ARG(dtype == SUGAR_F32 || dtype == SUGAR_BF16, "subnet_create: bad dtype %d", dtype);
  SUGAR_CHECK_ARG(d->depth >= 1 && d->depth <= SUGAR_MAX_DEPTH, "subnet_create: depth %d out of range", d->depth);
  SUGAR_CHECK_ARG(d->in_feats == SUGAR_N_MELS, "subnet_create: in_feats %d (only 80 supported)", d->in_feats);
  SUGAR_CHECK_ARG(d->c0 > 0 && d->ccat > 0 && d->embed_dim > 0 && d->bottleneck == 128,
                  "subnet_create: bad dims c0 %d ccat %d embed %d bottleneck %d", d->c0, d->ccat, d->embed_dim,
                  d->bottleneck);
  SUGAR_CHECK_ARG(d->layer1_w && bn_ok(d->layer1_bn) && d->catconv_w && d->pool1_w && d->pool2_w && bn_ok(d->bnpool) &&
                      d->linear_w && d->linear_b && bn_ok(d->bnembed),
                  "subnet_create: null tensor");
  SUGAR_DEVICE_GUARD(device);
  SugarSubnet* n = new SugarSubnet();
  n->device = device;
  int rc;
  {
    // all packed tensors into the subnet's chunks, sent in one go (SUGAR_NO_ARENA: one hipMalloc + hipMemcpy per tensor
    // as before, for A/B runs)
    ArenaScope scope(getenv("SUGAR_NO_ARENA") ? nullptr : &n->arena);
    rc = create(d, dtype, n);
  }
  if (!rc) rc = arena_flush(&n->arena);
  if (rc) {
    destroy(n);
    return rc;
  }
  *out = n;
  return SUGAR_OK;
}

int sugar_subnet_destroy(SugarSubnet* net) {
  if (!net) return SUGAR_OK;
  SUGAR_DEVICE_GUARD(net->device);
  destroy(net);
  return SUGAR_OK;
}

int64_t sugar_workspace_bytes(const SugarSubnet* net, int B, int64_t R) {
  if (!net || B <= 0 || R <= 0) {
    set_error("workspace_bytes: bad arguments");
    return SUGAR_EINVAL;
  }
  return plan(net, B, R, nullptr).bytes;
}

int64_t sugar_fbank_scratch_bytes(int B, int64_t R) {
  if (B <= 0 || R <= 0) {
    set_error("fbank_scratch_bytes: bad arguments");
    return SUGAR_EINVAL;
  }
  return fbank_scratch_floats(B, R) * 4;
}

int sugar_fbank_fwd(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len,
                    const int32_t* d_row_off, int B, int64_t R, int t_max, float* d_logmel, void* d_feats, int dtype,
                    void* stream) {
  return launch_fbank(d_wav, d_wav_off, d_wav_len, d_row_off, B, R, t_max, d_logmel, d_feats, dtype, (hipStream_t)stream);
}

int sugar_tdnn_fwd(const SugarSubnet* net, const void* d_feats, const int32_t* d_row_off, int B, int64_t R, int t_max,
                   void* d_workspace, int64_t workspace_bytes, float* d_emb, int normalize, void* stream) {
  int rc = check_fwd_args(net, d_feats, d_row_off, B, R, t_max, d_workspace, workspace_bytes, d_emb);
  if (rc) return rc;
  SUGAR_DEVICE_GUARD(net->device);
  const Workspace w = plan(net, B, R, d_workspace);
  return tdnn_forward(net, d_feats, d_row_off, B, R, t_max, w, d_emb, normalize, -1, nullptr, (hipStream_t)stream,
                      layer1_as_tap_gemm(net));
}

int sugar_embed_fwd(const SugarSubnet* net, const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len,
                    const int32_t* d_row_off, int B, int64_t R, int t_max, void* d_workspace, int64_t workspace_bytes,
                    float* d_emb, int normalize, void* stream) {
  int rc = check_fwd_args(net, d_wav, d_row_off, B, R, t_max, d_workspace, workspace_bytes, d_emb);
  if (rc) return rc;
  SUGAR_CHECK_ARG(d_wav_off && d_wav_len, "embed_fwd: null pointer");
  SUGAR_DEVICE_GUARD(net->device);
  const Workspace w = plan(net, B, R, d_workspace);
  hipStream_t s = (hipStream_t)stream;
  // the front-end writes its features straight into the zero-padded layout the layer1 tap GEMM reads (no copy kernel)
  const bool padded = layer1_as_tap_gemm(net);
  const int halo = padded ? (net->layer1.ktaps - 1) / 2 : 0;
  {
    Prof p(SUGAR_STAGE_FBANK, s);
    if ((rc = launch_fbank(d_wav, d_wav_off, d_wav_len, d_row_off, B, R, t_max, w.logmel, padded ? w.featp : w.feats,
                           net->dtype, s, halo, padded ? gemm_taps_slack(net->layer1, B, R) : 0, padded ? w.orow : nullptr,
                           padded ? gemm_taps_orow_ints(net->layer1, B, R) : 0)))
      return rc;
  }
  return tdnn_forward(net, padded ? nullptr : w.feats, d_row_off, B, R, t_max, w, d_emb, normalize, -1, nullptr, s, padded);
}

int sugar_tdnn_recalibrate(SugarSubnet* net, const void* d_feats, const int32_t* d_row_off, int B, int64_t R, int t_max,
                           void* d_workspace, int64_t workspace_bytes, float momentum, float* d_emb, void* stream) {
  float dummy;
  int rc = check_fwd_args(net, d_feats, d_row_off, B, R, t_max, d_workspace, workspace_bytes, d_emb ? d_emb : &dummy);
  if (rc) return rc;
  SUGAR_CHECK_ARG(momentum >= 0.f && momentum <= 1.f, "recalibrate: momentum %g outside [0, 1]", momentum);
  SUGAR_CHECK_ARG(B > 1, "recalibrate: train-mode BatchNorm needs more than one utterance per batch");
  SUGAR_DEVICE_GUARD(net->device);
  const Workspace w = plan(net, B, R, d_workspace);
  return tdnn_recalibrate(net, d_feats, d_row_off, B, R, t_max, w, momentum, d_emb, (hipStream_t)stream);
}

int sugar_embed_recalibrate(SugarSubnet* net, const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len,
                            const int32_t* d_row_off, int B, int64_t R, int t_max, void* d_workspace,
                            int64_t workspace_bytes, float momentum, float* d_emb, void* stream) {
  float dummy;
  int rc = check_fwd_args(net, d_wav, d_row_off, B, R, t_max, d_workspace, workspace_bytes, d_emb ? d_emb : &dummy);
  if (rc) return rc;
  SUGAR_CHECK_ARG(d_wav_off && d_wav_len, "embed_recalibrate: null pointer");
  SUGAR_CHECK_ARG(momentum >= 0.f && momentum <= 1.f, "recalibrate: momentum %g outside [0, 1]", momentum);
  SUGAR_CHECK_ARG(B > 1, "recalibrate: train-mode BatchNorm needs more than one utterance per batch");
  SUGAR_DEVICE_GUARD(net->device);
  const Workspace w = plan(net, B, R, d_workspace);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = launch_fbank(d_wav, d_wav_off, d_wav_len, d_row_off, B, R, t_max, w.logmel, w.feats, net->dtype, s))) return rc;
  return tdnn_recalibrate(net, w.feats, d_row_off, B, R, t_max, w, momentum, d_emb, s);
}

int sugar_subnet_bn_count(const SugarSubnet* net) { return net ? 3 + 9 * net->depth : SUGAR_EINVAL; }

int sugar_subnet_bn_read(SugarSubnet* net, int bn_index, float* h_mean, float* h_var, int n) {
  SUGAR_CHECK_ARG(net && h_mean && h_var, "bn_read: null pointer");
  ConvLayer* conv;
  BNState* st = bn_by_index(net, bn_index, &conv);
  SUGAR_CHECK_ARG(st && st->p, "bn_read: no BatchNorm %d", bn_index);
  SUGAR_CHECK_ARG(n == st->c, "bn_read: BatchNorm %d has %d channels, buffers hold %d", bn_index, st->c, n);
  SUGAR_DEVICE_GUARD(net->device);
  SUGAR_HIP(hipDeviceSynchronize());
  std::vector<float> v[4];
  int rc = bn_download(*st, v);
  if (rc) return rc;
  memcpy(h_mean, v[2].data(), (size_t)n * sizeof(float));
  memcpy(h_var, v[3].data(), (size_t)n * sizeof(float));
  return SUGAR_OK;
}

int sugar_subnet_bn_reset(SugarSubnet* net) {
  SUGAR_CHECK_ARG(net, "bn_reset: null pointer");
  SUGAR_DEVICE_GUARD(net->device);
  for (int i = 0; i < sugar_subnet_bn_count(net); ++i) {
    ConvLayer* conv;
    BNState* st = bn_by_index(net, i, &conv);
    std::vector<float> mv(2 * (size_t)st->n, 0.f);
    for (int k = 0; k < st->n; ++k) mv[(size_t)st->n + k] = 1.f;          // nn.BatchNorm1d.reset_running_stats
    SUGAR_HIP(hipMemcpy(st->mean(), mv.data(), mv.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return SUGAR_OK;
}

int sugar_subnet_bn_commit(SugarSubnet* net) {
  SUGAR_CHECK_ARG(net, "bn_commit: null pointer");
  SUGAR_DEVICE_GUARD(net->device);
  SUGAR_HIP(hipDeviceSynchronize());
  int rc;
  for (int i = 0; i < sugar_subnet_bn_count(net) - 2; ++i) {
    ConvLayer* conv;
    BNState* st = bn_by_index(net, i, &conv);
    if ((rc = launch_bn_fold(*st, conv->scale, conv->shift, nullptr))) return rc;
  }
  std::vector<float> p[4], e[4];
  if ((rc = bn_download(net->bn_pool, p)) || (rc = bn_download(net->bn_embed, e))) return rc;
  const float* const pool_bn[4] = {p[0].data(), p[1].data(), p[2].data(), p[3].data()};
  const float* const embed_bn[4] = {e[0].data(), e[1].data(), e[2].data(), e[3].data()};
  if ((rc = fold_tail(net, pool_bn, embed_bn))) return rc;
  SUGAR_HIP(hipDeviceSynchronize());
  return SUGAR_OK;
}

int sugar_tdnn_tap(const SugarSubnet* net, const void* d_feats, const int32_t* d_row_off, int B, int64_t R, int t_max,
                   void* d_workspace, int64_t workspace_bytes, int what, float* d_out, int64_t out_elems,
                   float* d_emb, void* stream) {
  int rc = check_fwd_args(net, d_feats, d_row_off, B, R, t_max, d_workspace, workspace_bytes, d_emb);
  if (rc) return rc;
  SUGAR_CHECK_ARG(d_out, "tap: null output");
  int C = -1;
  int64_t rows = R;
  if (what == 0) C = net->c0p;
  else if (what >= 1 && what <= 4 * net->depth) {
    const int i = (what - 1) / 4, k = (what - 1) % 4;
    C = (k == 0 || k == 1) ? net->blocks[i].midp : net->c0p;
  } else if (what == 100) C = net->ccatp;
  else if (what == 101) { C = 2 * net->ccatp; rows = B; }
  else if (what == 102) C = net->bott;
  SUGAR_CHECK_ARG(C > 0, "tap: unknown stage %d", what);
  SUGAR_CHECK_ARG(out_elems >= rows * C, "tap: output holds %lld elements, stage needs %lld", (long long)out_elems,
                  (long long)(rows * C));
  SUGAR_DEVICE_GUARD(net->device);
  const Workspace w = plan(net, B, R, d_workspace);
  rc = tdnn_forward(net, d_feats, d_row_off, B, R, t_max, w, d_emb, 0, what, d_out, (hipStream_t)stream,
                    layer1_as_tap_gemm(net));
  return rc ? rc : C;
}

int sugar_conv1d_fwd(const float* h_w, const SugarBN* h_bn, int cout, int cin, int k, int dilation, int flags,
                     const void* d_x, const void* d_x2, void* d_y, const int32_t* d_row_off, int B, int64_t R,
                     int t_max, int dtype, void* stream) {
  SUGAR_CHECK_ARG(dtype == SUGAR_F32 || dtype == SUGAR_BF16, "conv1d: bad dtype %d", dtype);
  SUGAR_CHECK_ARG(h_w && d_x && d_y && d_row_off && B > 0 && R > 0 && t_max > 0, "conv1d: bad arguments");
  SUGAR_CHECK_ARG(cin % 8 == 0 && cout % 8 == 0, "conv1d: this test entry needs channel counts that are multiples of 8");
  ConvLayer L;
  int rc = pack_conv(&L, h_w, (flags & EPI_AFFINE) ? h_bn : nullptr, cout, cin, k, dilation, flags, cout, nullptr, cin,
                     nullptr, dtype);
  if (rc) { free_conv(&L); return rc; }
  int2* rowseg = nullptr;
  void* xpad = nullptr;
  int32_t* orow = nullptr;
  hipStream_t s = (hipStream_t)stream;
  // same dispatch as the forward (tdnn_forward): dilation-1 tap convs that qualify run as the LDS-DMA GEMM over
  // overlapping rows (layer1), everything else through launch_conv
  const bool taps = !d_x2 && !tune_get(TUNE_GENERIC_LAYER1) && gemm_taps_eligible(L, cin, dtype);
  if (hipMalloc(&rowseg, R * sizeof(int2)) != hipSuccess ||
      (taps && (hipMalloc(&xpad, (size_t)gemm_taps_pad_rows(L, B, R) * cin * 2) != hipSuccess ||
                hipMalloc(&orow, (size_t)gemm_taps_orow_ints(L, B, R) * 4) != hipSuccess))) {
    if (rowseg) (void)hipFree(rowseg);
    if (xpad) (void)hipFree(xpad);
    free_conv(&L);
    set_error("conv1d: hipMalloc");
    return SUGAR_EHIP;
  }
  if (taps) {
    rc = launch_pad_rows(L, d_x, xpad, d_row_off, B, R, t_max, s);
    if (!rc) rc = launch_gemm_taps(L, xpad, orow, d_y, cout, d_row_off, B, R, t_max, s);
  } else {
    rc = launch_build_rowseg(d_row_off, B, t_max, rowseg, s);
    if (!rc) {
      ConvIO io;
      io.x = d_x; io.ldx = cin; io.x2 = d_x2; io.ldx2 = cin; io.y = d_y; io.ldy = cout;
      rc = launch_conv(L, io, rowseg, R, dtype, s);
    }
  }
  (void)hipStreamSynchronize(s);
  (void)hipFree(rowseg);
  if (xpad) (void)hipFree(xpad);
  if (orow) (void)hipFree(orow);
  free_conv(&L);
  return rc;
}

int sugar_profile_enable(unsigned mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof_recs.clear();
  ++g_prof_gen;
  g_prof_mask.store(mask, std::memory_order_relaxed);
  return SUGAR_OK;
}

int sugar_profile_read(int stage, int* count, double* total_ms) {
  SUGAR_CHECK_ARG(stage >= 0 && stage < SUGAR_STAGE_COUNT && count && total_ms, "profile_read: bad arguments");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  double t = 0.0;
  for (auto& r : g_prof_recs) {
    if (r.stage != stage) continue;
    SUGAR_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    SUGAR_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    t += ms;
    ++n;
  }
  *count = n;
  *total_ms = t;
  return SUGAR_OK;
}

const char* sugar_profile_stage_name(int stage) {
  return (stage >= 0 && stage < SUGAR_STAGE_COUNT) ? kStageNames[stage] : "";
}

int sugar_tune_set(const char* name, int64_t value) {
  SUGAR_CHECK_ARG(name, "tune_set: null name");
  tune_init();
  for (int k = 0; k < TUNE_COUNT; ++k)
    if (!strcmp(name, kTuneNames[k])) {
      g_tune[k].store(value, std::memory_order_relaxed);
      return SUGAR_OK;
    }
  set_error("tune_set: unknown tunable '%s'", name);
  return SUGAR_EINVAL;
}

int sugar_tune_get(const char* name, int64_t* value) {
  SUGAR_CHECK_ARG(name && value, "tune_get: null pointer");
  for (int k = 0; k < TUNE_COUNT; ++k)
    if (!strcmp(name, kTuneNames[k])) {
      *value = tune_get(k);
      return SUGAR_OK;
    }
  set_error("tune_get: unknown tunable '%s'", name);
  return SUGAR_EINVAL;
}

int sugar_l2norm(float* d_x, int64_t n, int dim, void* stream) { return launch_l2norm(d_x, n, dim, (hipStream_t)stream); }

}  // extern "C"
