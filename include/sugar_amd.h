/*
 * sugar_amd.h -- C-ABI of libsugar_amd.so: the MI355X (gfx950) embedding-extraction hot path of
 * EfficientTDNN (mechanicalsea/sugar), hand-written HIP.
 *
 * The reference has no FFI / plugin registry: its boundary for this path is the Python call surface
 * (SURVEY.md section 8b).  Every entry point below therefore cites the reference *function* it replaces
 * (file:line relative to the reference root).  INTEGRATION.md shows the ctypes stub a sugar maintainer
 * would add to bind them.
 *
 * Conventions
 *   - plain C types only; every pointer named d_* is a DEVICE pointer owned by the caller, h_* is a HOST pointer
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued, nothing syncs
 *   - device: entry points that take a SugarSubnet* make the subnet's device current for the call and restore the
 *     caller's device on return; `stream` must belong to that device.  Entry points WITHOUT a subnet handle
 *     (sugar_fbank_fwd, sugar_l2norm, the scoring calls, sugar_conv1d_fwd) run on the caller's CURRENT device, which
 *     must be the device of `stream` and of every d_* pointer.
 *   - threads: re-entrant per (device, stream); the only process-wide mutable state is the profiling record list
 *     (mutex), the tunables (atomics) and per-device read-only tables built once (mutex)
 *   - return 0 on success, negative SUGAR_E* on failure; sugar_last_error() gives the per-thread message
 *   - activations are CHANNELS-LAST and PACKED over utterances: row r of a [R, C] buffer is one frame, utterance
 *     b owns rows [row_off[b], row_off[b+1]);  R = row_off[B].  There is no padding between utterances, so
 *     variable-length batches need no masks: every conv zero-pads at each utterance's true ends and every
 *     reduction (instance norm, SE mean, attentive pooling) runs over each utterance's true frames.
 *   - dtype: SUGAR_F32 = exact-f32 MFMA path (parity >= 0.9999 cosine), SUGAR_BF16 = bf16 operands with f32
 *     accumulation (parity >= 0.999 cosine).  It selects the element type of feats / activations / weights.
 */
#ifndef SUGAR_AMD_H
#define SUGAR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUGAR_F32 0
#define SUGAR_BF16 1

#define SUGAR_OK 0
#define SUGAR_EINVAL (-1)   /* bad argument (shape, alignment, NULL)            */
#define SUGAR_EHIP (-2)     /* a HIP runtime call failed                        */
#define SUGAR_ENOMEM (-3)   /* workspace too small / allocation failed          */

#define SUGAR_MAX_DEPTH 8
#define SUGAR_RES2_SCALE 8  /* Res2 scale, sugar/models/dynamictdnn.py:384      */
#define SUGAR_N_MELS 80
#define SUGAR_HOP 160

/* Per-thread message of the last failing call. */
const char* sugar_last_error(void);
/* Library version string; also a cheap "does the .so load" probe. */
const char* sugar_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Subnet description: the tensors of `DynamicTDNN.clone(config).state_dict()`
 * (sugar/models/dynamictdnn.py:713-730; key names in SURVEY.md Appendix A), as host float32 pointers in the
 * reference's own layouts: conv weights [Cout][Cin][k], linear weights [out][in], BN as 4 vectors of length C.
 * Weight materialisation (channel slices / gathers and the k53/k31 kernel transforms,
 * sugar/modules/dynamicarch.py:411-432) happens in the host mirror before this call.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
  const float* weight;        /* gamma */
  const float* bias;          /* beta  */
  const float* running_mean;
  const float* running_var;
} SugarBN;

typedef struct {
  int mid_channels;           /* 8 * width: out channels of first_conv (dynamictdnn.py:410-411)            */
  int kernel;                 /* Res2 conv kernel size: 1, 3 or 5                                          */
  const float* first_w;       /* [mid][C0][1]    blocks.i.first_conv.conv.weight                           */
  SugarBN first_bn;           /* [mid]                                                                     */
  const float* res_w[SUGAR_RES2_SCALE - 1];   /* [w][w][kernel]  blocks.i.resconv.convs.j.conv.weight      */
  SugarBN res_bn[SUGAR_RES2_SCALE - 1];       /* [w]                                                       */
  const float* last_w;        /* [C0][mid][1]    blocks.i.last_conv.conv.weight                            */
  SugarBN last_bn;            /* [C0]                                                                      */
  const float* se_reduce_w;   /* [C0/4][C0]      blocks.i.se.reduce.weight                                 */
  const float* se_reduce_b;   /* [C0/4]                                                                    */
  const float* se_expand_w;   /* [C0][C0/4]      blocks.i.se.expand.weight                                 */
  const float* se_expand_b;   /* [C0]                                                                      */
} SugarBlock;

typedef struct {
  int in_feats;               /* 80                                                                        */
  int embed_dim;              /* 192                                                                       */
  int depth;                  /* number of SE-Res2 blocks (2..4 for tdnn8m2g)                              */
  int c0;                     /* layer1 / block in-out channels                                            */
  int k0;                     /* layer1 kernel size                                                        */
  int ccat;                   /* catconv out channels                                                      */
  int bottleneck;             /* attentive pooling bottleneck, 128 (dynamictdnn.py:494)                    */
  const float* layer1_w;      /* [C0][in_feats][k0]   layer1.conv.weight                                   */
  SugarBN layer1_bn;          /* [C0]                                                                      */
  SugarBlock blocks[SUGAR_MAX_DEPTH];
  const float* catconv_w;     /* [Ccat][depth*C0][1]  catconv.weight   (ReLU, no BN: dynamictdnn.py:663)   */
  const float* pool1_w;       /* [bottleneck][Ccat][1] pooling.linear1.weight                              */
  const float* pool2_w;       /* [Ccat][bottleneck][1] pooling.linear2.weight                              */
  SugarBN bnpool;             /* [2*Ccat]                                                                  */
  const float* linear_w;      /* [embed][2*Ccat]      linear.weight                                        */
  const float* linear_b;      /* [embed]                                                                   */
  SugarBN bnembed;            /* [embed]                                                                   */
} SugarSubnetDesc;

typedef struct SugarSubnet SugarSubnet;   /* opaque, library-owned packed weights on one device */

/* Replaces: DynamicTDNN.clone() materialising static nn.Conv1d/BatchNorm1d/Linear leaves
 * (dynamictdnn.py:713-730, dynamicarch.py:188-208, 320-339, 482-501) + `.to(device)`.
 * Packs weights into MFMA-friendly [Cout][tap*Cin] panels (dtype), folds every eval-mode BN into a per-channel
 * (scale, shift) epilogue applied AFTER the ReLU (conv -> ReLU -> BN, dynamictdnn.py:183), folds
 * bnpool . linear . bnembed (dynamictdnn.py:664-665) into one affine map, and uploads to `device`.
 * This call allocates device memory (a few large chunks that hold every packed tensor of the subnet) and sends the
 * weights with one asynchronous copy per chunk on a stream of the library's own, for which alone it waits: it does not
 * touch the null stream and does not wait for other work on the device, so a host thread may pack the next model while
 * the current one runs (sugar_amd/sweep.py).  sugar_subnet_destroy frees the chunks (hipFree waits for the device). */
int sugar_subnet_create(const SugarSubnetDesc* h_desc, int dtype, int device, SugarSubnet** out);
int sugar_subnet_destroy(SugarSubnet* net);

/* Bytes of device workspace sugar_embed_fwd / sugar_tdnn_fwd need for B utterances totalling R frames. */
int64_t sugar_workspace_bytes(const SugarSubnet* net, int B, int64_t R);

/* Replaces: LogMelFbanks.forward (sugar/transforms.py:60-77) = PreEmphasis (:22-27) + torchaudio MelSpectrogram
 * (:50-52; STFT n_fft 512, Hamming(400) periodic, hop 160, centre reflect pad, |.|^2, HTK mel 80 bins 20-7600 Hz)
 * + 1e-6, log (:70-72) + InstanceNorm1d(80) (:47,73), as one fused DFT kernel plus a per-utterance normaliser.
 *   d_wav      float32 samples of all utterances; utterance b = d_wav[wav_off[b] .. wav_off[b]+wav_len[b])
 *   d_row_off  int32[B+1] frame offsets, row_off[b+1]-row_off[b] must equal 1 + wav_len[b]/160
 *   d_logmel   float32 scratch of sugar_fbank_scratch_bytes(B, R): [R][80] log mel before normalisation, followed by
 *              the per-chunk statistics of utterances longer than 320 frames
 *   d_feats    dtype   [R][80] output, channels-last (the reference returns [B,1,80,T])
 *   R          total frames = row_off[B];  t_max  max frames of any utterance (grid sizing) */
int64_t sugar_fbank_scratch_bytes(int B, int64_t R);
int sugar_fbank_fwd(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len,
                    const int32_t* d_row_off, int B, int64_t R, int t_max,
                    float* d_logmel, void* d_feats, int dtype, void* stream);

/* Allocation-free variant of sugar_fbank_fwd.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory or carry a scratch buffer -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same kernels and same results as sugar_fbank_fwd bit for bit; the log-mel scratch that d_logmel
 * brings in there is a file-scope __device__ array of the library here (5.44 MB per GPU, held until the process ends),
 * so one call handles at most SUGAR_NOALLOC_MAX_BATCH utterances totalling SUGAR_NOALLOC_MAX_FRAMES frames (163 s of
 * audio) and larger calls return SUGAR_EINVAL.  The caller still passes inputs and the output.  After the one-off
 * per-device set-up (done by sugar_subnet_create on that device, otherwise by the first front-end call there) it calls
 * no hipMalloc / hipFree / allocating routine.  Every call on a GPU shares that GPU's array: calls that overlap on one
 * GPU (two streams, two threads) race; serialise them.  Nothing in the array has to start at zero. */
#define SUGAR_NOALLOC_MAX_FRAMES 16384
#define SUGAR_NOALLOC_MAX_BATCH 256
int sugar_fbank_fwd_noalloc(const float* d_wav, const int64_t* d_wav_off, const int32_t* d_wav_len,
                            const int32_t* d_row_off, int B, int64_t R, int t_max, void* d_feats, int dtype,
                            void* stream);

/* Replaces: DynamicTDNN.forward on a cloned subnet (dynamictdnn.py:645-666) [+ F.normalize(p=2) of
 * sugar/vectors.py:143-144 when `normalize` != 0].
 *   d_feats [R][80] dtype of the subnet;  d_emb float32 [B][embed_dim] */
int sugar_tdnn_fwd(const SugarSubnet* net, const void* d_feats, const int32_t* d_row_off, int B, int64_t R,
                   int t_max, void* d_workspace, int64_t workspace_bytes, float* d_emb, int normalize,
                   void* stream);

/* Replaces: SpeakerModel.forward with a transform (sugar/models/__init__.py:79-114): waveforms -> embeddings.
 * = sugar_fbank_fwd + sugar_tdnn_fwd on one stream, intermediate buffers in the workspace. */
int sugar_embed_fwd(const SugarSubnet* net, const float* d_wav, const int64_t* d_wav_off,
                    const int32_t* d_wav_len, const int32_t* d_row_off, int B, int64_t R, int t_max,
                    void* d_workspace, int64_t workspace_bytes, float* d_emb, int normalize, void* stream);

/* ---- BN recalibration ---------------------------------------------------------------
 * Replaces: one iteration of batch_forward (sugar/models/__init__.py:201-242): `model.train(); model.forward(x, None)`
 * -- every BatchNorm normalises with the statistics of THIS batch (biased variance over all frames / utterances of
 * the batch, DynamicBatchNorm1d.forward, sugar/modules/dynamicarch.py:254-291) and moves its running statistics by
 * `momentum` (nn.BatchNorm1d default 0.1; unbiased variance in the running update).  The running statistics live in
 * the handle; the eval-mode entry points keep using the statistics of the last sugar_subnet_bn_commit.
 * d_emb (may be NULL) receives the train-mode embeddings [B][embed].  B must be > 1 (as torch requires). */
int sugar_tdnn_recalibrate(SugarSubnet* net, const void* d_feats, const int32_t* d_row_off, int B, int64_t R,
                           int t_max, void* d_workspace, int64_t workspace_bytes, float momentum, float* d_emb,
                           void* stream);
/* = sugar_fbank_fwd + sugar_tdnn_recalibrate (SpeakerModel.forward in train(), models/__init__.py:96-105). */
int sugar_embed_recalibrate(SugarSubnet* net, const float* d_wav, const int64_t* d_wav_off,
                            const int32_t* d_wav_len, const int32_t* d_row_off, int B, int64_t R, int t_max,
                            void* d_workspace, int64_t workspace_bytes, float momentum, float* d_emb, void* stream);
/* BatchNorm layers of a packed subnet, in state-dict order: 0 layer1.bn; 1 + 9*i + {0 first_conv.bn,
 * 1..7 resconv.convs.(k-1).bn, 8 last_conv.bn} for block i; 1 + 9*depth bnpool; 2 + 9*depth bnembed. */
int sugar_subnet_bn_count(const SugarSubnet* net);
/* Replaces: reading bn.running_mean / bn.running_var after batch_forward (bn_state_dict, sugar/utils/utility.py:9-24):
 * copies the current running statistics of BatchNorm `bn_index` to host arrays of n = its channel count, in the
 * reference's channel order.  Synchronises the device. */
int sugar_subnet_bn_read(SugarSubnet* net, int bn_index, float* h_mean, float* h_var, int n);
/* Replaces: nn.BatchNorm1d.reset_running_stats on every BN of the subnet (mean 0, var 1). */
int sugar_subnet_bn_reset(SugarSubnet* net);
/* Re-derives the folded (scale, shift) vectors and the folded tail from the current running statistics, so that
 * the eval-mode entry points use them (the reference needs no such step: `model.eval()` reads the buffers).
 * Synchronises the device. */
int sugar_subnet_bn_commit(SugarSubnet* net);

/* Debug / parity tap: runs sugar_tdnn_fwd (un-normalised embeddings to d_emb) and additionally copies one
 * intermediate activation, converted to float32, to d_out as [R][C] in the library's INTERNAL channel layout
 * (channels padded to multiples of 8; Res2 groups padded per group -- see DESIGN.md "data layout").
 * `what`: 0 layer1; 1+4*i+{0 first_conv, 1 res2, 2 last_conv, 3 block output} for block i; 100 catconv;
 * 101 pooled statistics [B][2*Ccat_p]; 102 tanh(pooling.linear1) [R][128].
 * Returns C (>0) or a negative error.  No reference counterpart: the reference exposes its nn.Modules directly
 * (tests call sub.layer1(x), sub.blocks[i].first_conv(...) ...). */
int sugar_tdnn_tap(const SugarSubnet* net, const void* d_feats, const int32_t* d_row_off, int B, int64_t R,
                   int t_max, void* d_workspace, int64_t workspace_bytes, int what, float* d_out,
                   int64_t out_elems, float* d_emb, void* stream);

/* Replaces: one DynamicConv1dReLUBN / DynamicConv1d forward (dynamictdnn.py:182-183, dynamicarch.py:401-409)
 * on packed channels-last rows -- the implicit-GEMM MFMA kernel in isolation, for per-kernel parity tests.
 *   h_w [Cout][Cin][k] host float32 (packed + uploaded inside: NOT a fast path), optional h_bn (NULL = no BN),
 *   d_x [R][Cin] dtype (+ optional d_x2 added element-wise before the conv, NULL = none), d_y [R][Cout] dtype.
 *   flags: bit0 ReLU, bit1 BN affine after ReLU, bit2 tanh.  Cin and Cout must be multiples of 8.
 *   Allocates and synchronises internally. */
int sugar_conv1d_fwd(const float* h_w, const SugarBN* h_bn, int cout, int cin, int k, int dilation, int flags,
                     const void* d_x, const void* d_x2, void* d_y, const int32_t* d_row_off, int B, int64_t R,
                     int t_max, int dtype, void* stream);

/* HOST memory only.  Replaces: Utterance.collate_fn's torch.cat of the segment stacks (sugar/database.py:144-158) when the
 * waveforms live in host memory: h_dst (typically pinned) <- the n float32 segments h_src[i][0 .. h_len[i]) back to
 * back, copied by n_threads threads over equal byte ranges.  No GPU call; the caller then issues ONE host-to-device copy. */
int sugar_pack_segments(const float* const* h_src, const int64_t* h_len, int n, float* h_dst, int n_threads);

/* HOST memory and files only.  Replace, for evaluation lists of mono 16-bit PCM wav files: the per-item Python loader
 * loadWAV (sugar/waveforms.py:47-115: scipy wavfile.read -> np.pad(mode="wrap") -> whole utterance or np.linspace crops
 * -> float32) as run by the reference's DataLoader workers / joblib threads (sugar/vectors.py:117-121, 130-133),
 * followed by Utterance.collate_fn's torch.cat (sugar/database.py:144-158).
 *   sugar_wav_probe: walks the RIFF chunks of h_paths[0 .. n): h_n_samples[i] = samples of file i and h_data_off[i] = byte
 *     offset of its data when it is mono, 16-bit PCM at `sample_rate`; h_n_samples[i] = -1 for any other format (the
 *     caller keeps the Python loader for such a list).  SUGAR_EINVAL when a file cannot be read.
 *   sugar_wav_load: segment j = h_seg_len[j] samples of file h_seg_file[j] from sample h_seg_start[j] on, continued
 *     periodically past the file's end (what the wrap padding appends), converted to float32 (exact) and written to
 *     h_dst (typically pinned) one segment after the other, by n_threads threads over equal sample ranges using pread.
 * Which segments a file yields (padding floor, crop starts) stays on the Python side (sugar_amd/waveforms.py). */
int sugar_wav_probe(const char* const* h_paths, int n, int sample_rate, int64_t* h_n_samples, int64_t* h_data_off,
                    int n_threads);
int sugar_wav_load(const char* const* h_paths, const int64_t* h_n_samples, const int64_t* h_data_off, int n_files,
                   const int32_t* h_seg_file, const int64_t* h_seg_start, const int64_t* h_seg_len, int n_segs,
                   float* h_dst, int n_threads);

/* Replaces: F.normalize(vec, p=2, dim=1) (sugar/vectors.py:108-109,143-144), in place, float32 [n][dim]. */
int sugar_l2norm(float* d_x, int64_t n, int dim, void* stream);

/* Replaces: score_trials' python loop over calculate_score(mode="cosine")
 * (sugar/scores.py:146-150, 188-206): score[t] = mean over the n x m pairwise cosine similarities
 * (F.cosine_similarity, eps 1e-8) of the enrol utterance's n segment vectors and the test utterance's m.
 *   d_vec      float32 [n_rows][dim]      all segment vectors, utterance u owns rows [seg_off[u], seg_off[u+1])
 *   d_seg_off  int32[n_utts+1]
 *   d_enroll / d_test  int32[n_trials] utterance ids
 *   d_scores   float32[n_trials] */
int sugar_cosine_trials(const float* d_vec, int dim, const int32_t* d_seg_off, int n_utts,
                        const int32_t* d_enroll, const int32_t* d_test, int64_t n_trials,
                        float* d_scores, void* stream);

/* Replaces: the same loop over calculate_score(mode="norm2") -- the reference's DEFAULT mode (sugar/scores.py:114,
 * 141-145): F.pairwise_distance(enroll[n, D, 1], test[1, D, m]) broadcasts to [n, D, m] and takes the 2-norm of
 * (x1 - x2 + 1e-6) over the LAST axis (the m test segments), so
 *   score[t] = - mean over (i < n, d < dim) of sqrt( sum over j < m of (e[i][d] - t[j][d] + 1e-6)^2 ).
 * Arguments as sugar_cosine_trials. */
int sugar_norm2_trials(const float* d_vec, int dim, const int32_t* d_seg_off, int n_utts,
                       const int32_t* d_enroll, const int32_t* d_test, int64_t n_trials,
                       float* d_scores, void* stream);

/* ---- adaptive score normalisation back-end -----------------------------------------
 * Replaces: score_cohorts (sugar/scores.py:77-111): d_out[u][m] = mean over utterance u's segment vectors of
 * F.cosine_similarity(segment, cohort m) (eps 1e-8).  d_vec / d_seg_off as in sugar_cosine_trials,
 * d_cohort float32 [n_cohort][dim], d_scratch float32 [(n_utts + n_cohort) * dim], d_out float32 [n_utts][n_cohort]. */
int sugar_cohort_scores(const float* d_vec, int dim, const int32_t* d_seg_off, int n_utts, const float* d_cohort,
                        int n_cohort, float* d_scratch, float* d_out, void* stream);
/* Replaces: the "Cohort Statistics" loop of asnorm (sugar/scores.py:370-379; torch.topk(value, nTop) -> mean,
 * sqrt(sum((v - mean)^2 / |nTop|))) and of snorm (:305-308; all values, divisor = number of trials, as written
 * there).  Per row of d_scores [n_rows][n_cols]: the ntop largest (ntop < 0: the |ntop| smallest) values are
 * selected exactly; d_mean[r] = their mean, d_std[r] = sqrt(sum((v - mean)^2) / divisor).  1 <= |ntop| <= n_cols. */
int sugar_topk_stats(const float* d_scores, int64_t n_rows, int n_cols, int ntop, float divisor, float* d_mean,
                     float* d_std, void* stream);
/* Replaces: the "Normalization Statistics" loop + the final lines of asnorm / snorm (sugar/scores.py:388-399):
 * d_out[t] = 0.5 * ((s - mean[e]) / max(std[e], std_eps) + (s - mean[u]) / max(std[u], std_eps)),
 * e = d_enroll[t], u = d_test[t] row indices into d_mean / d_std. */
int sugar_snorm_trials(const float* d_scores, const int32_t* d_enroll, const int32_t* d_test, int64_t n_trials,
                       const float* d_mean, const float* d_std, float std_eps, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Live per-stage timing (bench.py's roofline figures).  No reference counterpart; the reference times whole
 * forwards with torch.cuda.synchronize (sugar/metrics.py:389-404).
 * sugar_profile_enable(mask): bit i set => every launch group of stage i inside sugar_embed_fwd / sugar_tdnn_fwd is
 * bracketed by a hipEvent pair recorded on the call's own stream; clears earlier records.  mask 0 = off (default).
 * sugar_profile_read(stage, &count, &total_ms): synchronises those events and returns how many brackets were
 * recorded and the sum of their elapsed times. */
#define SUGAR_STAGE_FBANK 0
#define SUGAR_STAGE_LAYER1 1
#define SUGAR_STAGE_FIRST 2
#define SUGAR_STAGE_RES2 3
#define SUGAR_STAGE_LAST 4
#define SUGAR_STAGE_SE_SCALE 5
#define SUGAR_STAGE_SE_APPLY 6
#define SUGAR_STAGE_CATCONV 7
#define SUGAR_STAGE_POOL1 8
#define SUGAR_STAGE_POOL_STATS 9
#define SUGAR_STAGE_TAIL 10
#define SUGAR_STAGE_ROWSEG 11
#define SUGAR_STAGE_COUNT 12
int sugar_profile_enable(unsigned mask);
int sugar_profile_read(int stage, int* count, double* total_ms);
const char* sugar_profile_stage_name(int stage);

/* Tunables for parity tests and in-job A/B runs.  No reference counterpart.  Each is read ONCE from the environment
 * variable SUGAR_<NAME IN CAPITALS> when the library first needs one; afterwards only these calls change them
 * (atomics: safe while other threads launch).  Names: "gemm_wreg_min_r" (default 19200: frames from which the 512-channel
 * 1x1 convs use the weights-in-registers GEMM), "gemm_uniform", "gemm_no_pingpong", "generic_res2", "res2_max_t",
 * "generic_layer1", "gemm_no_persist", "res2_no_db", "recal_separate_stats" (switches, default 0), "gemm_no_quad" (0 / 1 / 2: schedule of the
 * persistent GEMM). */
int sugar_tune_set(const char* name, int64_t value);
int sugar_tune_get(const char* name, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* SUGAR_AMD_H */
