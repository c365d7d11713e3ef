/*
 * nacf_hip.h -- C ABI of libnacf_hip.so, the MI355X (gfx950) implementation of
 * the NACF video-captioning hot path.
 *
 * The reference (yangbang18/Non-Autoregressive-Video-Captioning) is pure
 * Python/PyTorch: it has no FFI of its own, so there is no upstream binding to
 * mirror symbol-for-symbol.  Each entry point below replaces the aten work
 * issued by one reference call site (cited as file:line, relative to the
 * upstream checkout); INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add to bind them.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative NACF_E* code on failure; the text of the
 *     last failure on the calling thread is returned by nacf_last_error();
 *   - never allocate, never synchronise, never throw: every launch is
 *     asynchronous on `stream` (safe to capture in a hipGraph);
 *   - pointers are DEVICE pointers into caller-owned, fp32 / int64 buffers on
 *     the current device; `ld*` arguments are leading dimensions in ELEMENTS;
 *   - results are deterministic for fixed inputs (no floating-point atomics);
 *   - dropout masks are a pure function of (rng_state[0]=seed,
 *     rng_state[1]=step, salt, element index), so backward regenerates them
 *     instead of storing them, and graph replays see fresh masks once
 *     nacf_rng_advance() has bumped the device-side step.
 */
#ifndef NACF_HIP_H
#define NACF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* nacf_stream_t; /* == hipStream_t */

/* error codes */
#define NACF_OK 0
#define NACF_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define NACF_EWORKSPACE (-2) /* workspace too small */
#define NACF_ELAUNCH (-3)  /* hipGetLastError() after launch */
#define NACF_EUNSUPPORTED (-4)

/* token ids, config/Constants.py:1-6 */
#define NACF_PAD 0
#define NACF_UNK 1
#define NACF_BOS 2
#define NACF_EOS 3
#define NACF_MASK 4
#define NACF_VIS 5

/* activations, models/bert.py:9-19 (ACT2FN) + models/Encoder.py:20-22 */
#define NACF_ACT_NONE 0
#define NACF_ACT_RELU 1
#define NACF_ACT_GELU_NEW 2   /* tanh form, models/bert.py:12-13 */
#define NACF_ACT_TANH 3
#define NACF_ACT_SIGMOID 4
#define NACF_ACT_TANH_SIGMOID 5 /* cols < act_split: tanh, else sigmoid (packed HighWay w1|w2) */
#define NACF_ACT_GELU_ERF 6   /* models/bert.py:9-10 */

const char* nacf_last_error(void);
int nacf_version(void);
/* number of entry points this header declares (every `nacf_*` function below and above, the three bookkeeping ones
 * included).  nacf_abi_count() returns the value the LIBRARY was built with; the ctypes loader
 * (runtime/lib.py:load) refuses a library whose count differs from its own signature table, so a stale
 * libnacf_hip.so next to a newer Python package fails at import time, not at the first missing symbol. */
#define NACF_ABI_COUNT 92
int nacf_abi_count(void);

/* ---- batch construction (reference: dataloader.py) ----------------------------------------
 * The reference assembles every sample in Python on the host.  Here the feature shards sit in HBM (a whole split of
 * MSRVTT is 9.8 GB of fp32 [N, 60, 2048] rows) or pass through a pinned staging buffer, and two launches build a batch.
 *
 * nacf_sample_frames: out[b, i, :] = src[video[b], frame(b, i), :]           (video == NULL: b; src_len == NULL: T)
 *   mode 0 'equally_sampling': middle of segment i of n_frames equal segments of the clip (dataloader.py:24-37);
 *   mode 1 'segment_random'  : one uniform draw per segment (Philox, device {seed, step} + salt);
 *   mode 2 'all_random'      : n_frames distinct frames of the clip, ascending (selection sampling on the same stream);
 *   a clip shorter than n_frames is stretched: round-half-even(i * (S-1) / (n_frames-1))      (dataloader.py:20-21,305)
 *   frame_ids (optional) receives the chosen indices [B, n_frames]. */
int nacf_sample_frames(const float* src, const int32_t* video, const int32_t* src_len, int B, int T, int D,
                       int n_frames, int mode, uint32_t salt, const uint64_t* rng_state, float* out,
                       int32_t* frame_ids, nacf_stream_t stream);
/* nacf_gather_clips_h2d: dst[j, :] = src_host[rows[j], :] for j < n, one asynchronous copy of `clip_bytes` per clip on
 *   `stream` (pinned host memory -> device staging; replaces the reference's per-sample HDF5 read + DataLoader collate,
 *   dataloader.py:222-239, when most frames of a clip are needed).  `rows` is a HOST array; src_host must be pinned for
 *   the copies to overlap with kernels. */
int nacf_gather_clips_h2d(void* dst, const void* src_host, const int32_t* rows, int n, size_t clip_bytes,
                          nacf_stream_t stream);
/* nacf_gather_clips_zc: the same gather by kernel-issued PCIe reads: src_host is PINNED host memory (device-mapped, 16-byte
 * aligned), rows_dev a DEVICE array; `workgroups` of 256 threads keep 16 x 16 bytes per thread in flight (<= 0: 24).
 * clip_bytes % 16 == 0.  Replaces the per-clip DMAs of dataloader.py:132-144,263-315's batch assembly when whole clips are
 * needed and the shards live in pinned host memory. */
int nacf_gather_clips_zc(void* dst, const void* src_host, const int32_t* rows_dev, int n, size_t clip_bytes, int workgroups,
                         nacf_stream_t stream);
/* nacf_build_targets: the decoder inputs / labels of B captions, dataloader.py:317-425.
 *   caps[b, 0..cap_len[b]) = <bos> w1 .. wn <eos> (int32, row pitch ld_caps), pos_tags alike.
 *   narformer != 0: masked-LM pair (:346-380).  train: a uniformly random subset of the n word slots (size uniform
 *     in [max(int(n*beta_low),1), max(int(n*beta_high),1)) -- widened by one when empty) is <mask> in `tokens` and
 *     keeps its word in `labels`, all other labels are <pad>; eval: every word is <mask>, labels = the sentence.
 *   narformer == 0: tokens = labels = the caption padded / cut to max_len with a closing <eos> (:333-337).
 *   visual_word && train: tokens_1 = <vis> per slot, labels_1 = the word where tag_demanded[tag] && !word_is_be[word],
 *     else <mask> (AR form: wrapped in <bos> .. <eos>)  (:382-425).
 *   All outputs are int64 [B, max_len]. */
int nacf_build_targets(const int32_t* caps, int ld_caps, const int32_t* cap_len, const int32_t* pos_tags,
                       const uint8_t* tag_demanded, const uint8_t* word_is_be, int B, int max_len, int narformer,
                       int visual_word, int train, double beta_low, double beta_high, uint32_t salt,
                       const uint64_t* rng_state, int64_t* tokens, int64_t* labels, int64_t* tokens_1,
                       int64_t* labels_1, nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * RNG state: device uint64[2] = {seed, step}.
 * ---------------------------------------------------------------------- */
int nacf_rng_advance(uint64_t* rng_state, nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * Linear layers (nn.Linear: y = x W^T + b, W is [N, K] row-major).
 * Fused epilogue, applied in this order to z = x W^T + bias:
 *   preact <- z (optional store) ; a = act(z) ; a = dropout(a, p_drop1) ;
 *   a += residual ; a = dropout(a, p_drop2) ; a *= (row_tokens[m] != PAD) .
 * Replaces: models/Encoder.py:19-25,65 ; models/bert.py:146-148 (q/k/v),
 * :192-200 (BertSelfOutput), :227-230 (BertIntermediate), :240-247
 * (BertOutput, which applies dropout twice), :271-299 (x non_pad_mask);
 * models/Predictor.py:15-20 ; models/__init__.py:83 (tgt_word_prj).
 * ---------------------------------------------------------------------- */
typedef struct nacf_epilogue {
  const float* bias;        /* [N] or NULL */
  int32_t act;              /* NACF_ACT_* */
  int32_t act_split;        /* NACF_ACT_TANH_SIGMOID only */
  float* preact;            /* optional [M, ld_preact] store of z */
  int64_t ld_preact;
  float p_drop1;
  uint32_t salt1;
  const float* residual;    /* optional [M, ld_residual] */
  int64_t ld_residual;
  float p_drop2;
  uint32_t salt2;
  const int64_t* row_tokens; /* optional [M] */
  const uint64_t* rng_state; /* required when a p_drop > 0 */
} nacf_epilogue;

/* Live-row list ("varlen" without repacking).  Decoder activations are [sequence, position]
 * slots of which 40-70 % are <pad>; every GEMM entry point takes an optional device-side list
 * of the live slots (ascending indices + device count).  The GEMM then walks only those rows
 * (gathering operands / scattering results through the list; dW reduces over them) and tiles
 * beyond `count` do no arithmetic.  Buffers keep their dense layout, shapes stay static, the host
 * never reads the count (hipGraph-safe).  NULL = all rows.
 * Rows outside the list: nacf_rowset_build also lists them, after the live ones (rows[count .. n), any order),
 * and a call made with zero_dead != 0 has the otherwise idle workgroups of the dead row tiles write
 * zeros to those rows of the outputs (Y and preact of linear_fwd, dX of linear_bwd_data), so the
 * output is fully defined without a separate memset.  With zero_dead == 0 dead rows are left untouched. */
typedef struct nacf_rowset {
  const int32_t* rows;   /* [n_slots]: count live slot indices ascending, then the dead ones (any order) */
  const int32_t* count;  /* device int32[1] */
  int32_t zero_dead;     /* != 0: zero the dead rows of the outputs (requires rows from nacf_rowset_build) */
} nacf_rowset;
/* live = { i : (tokens == NULL || tokens[i] != PAD) && (flags == NULL || flags[i] != 0) }:
 * rows[0 .. count) = live ascending, rows[count .. n) = the rest (descending), count[0] = |live| */
int nacf_rowset_build(const int64_t* tokens, const uint8_t* flags, int64_t n, int32_t* rows,
                      int32_t* count, nacf_stream_t stream);

int nacf_linear_fwd(const float* X, int64_t ldx, const float* W, int64_t ldw,
                    float* Y, int64_t ldy, int M, int N, int K,
                    const nacf_epilogue* ep, const nacf_rowset* rows, nacf_stream_t stream);

/* dX[M,K] = beta*dX + dZ[M,N] W[N,K]   (autograd of nn.Linear wrt input).  When the reduce
 * dimension N is long and the output narrow (the vocabulary projection), N is split over
 * workgroups into fp32 slabs in `ws` (nacf_linear_bwd_data_workspace bytes; 256 when no split
 * is used, ws may then be NULL) and combined in a fixed order. */
size_t nacf_linear_bwd_data_workspace(int M, int N, int K);
int nacf_linear_bwd_data(const float* dZ, int64_t lddz, const float* W, int64_t ldw,
                         float* dX, int64_t lddx, int M, int N, int K, float beta,
                         void* ws, size_t ws_bytes, const nacf_rowset* rows, nacf_stream_t stream);

/* dW[N,K] = beta*dW + dZ^T X ; db[N] = beta*db + colsum(dZ) (db may be NULL).
 * The M reduction is split over workgroups; partial slabs live in `ws`
 * (nacf_linear_bwd_weight_workspace bytes) and are combined in a fixed order. */
size_t nacf_linear_bwd_weight_workspace(int M, int N, int K);
int nacf_linear_bwd_weight(const float* dZ, int64_t lddz, const float* X, int64_t ldx,
                           float* dW, int64_t lddw, float* db, int M, int N, int K,
                           float beta, void* ws, size_t ws_bytes, const nacf_rowset* rows,
                           nacf_stream_t stream);

/* Grouped weight gradients.  Between nacf_dw_group_begin(defer_gemm) and nacf_dw_group_flush(stream):
 *   defer_gemm = 0: nacf_linear_bwd_weight launches its GEMM at once but QUEUES the combine of its partial slabs (and of
 *     the bias-gradient partials); the flush runs every queued combine in one launch.
 *   defer_gemm = 1: the GEMM itself is queued too (calls that the 128x128 bf16 kernel serves: bf16 / bf16x3 mode,
 *     16-byte addressable operands, N, K >= 128; the others behave as with 0).  The flush launches all queued GEMMs as
 *     ONE grid per 16 problems (a device-side table of problems, longest reduce walks first) with reduce splits chosen
 *     for the group as a whole -- 1-4 instead of the 8-17 a lone launch needs to fill 256 CUs -- then the combines.
 * Sums are taken in a fixed order either way (slab index), so results are deterministic; with defer_gemm = 1 the split
 * count differs from the ungrouped call's, so they are equal to it only to rounding.
 * Contract for the caller while a group is open: each call gets its own `ws` region, untouched until the flush; with
 * defer_gemm = 1 dZ, X and the row set are untouched until the flush as well; one dW target is queued at most once (a
 * second call for the same dW returns NACF_EINVAL -- flush first); dW / db are not read before the flush.
 * nacf_dw_group_pending: queued GEMMs + combines, or -1 when no group is open.  nacf_dw_group_stats: grouped GEMM
 * launches and their workgroups at the last flush.  One group at a time per process. */
int nacf_dw_group_begin(int defer_gemm);
int nacf_dw_group_flush(nacf_stream_t stream);
/* launches the queued GEMMs only (returns how many problems; < 0 = error); the combines wait for nacf_dw_group_flush */
int nacf_dw_group_launch_gemms(nacf_stream_t stream);
int nacf_dw_group_pending(void);
int nacf_dw_group_stats(int* launches, int* workgroups);

/* Grouped forward / dX GEMMs of INDEPENDENT problems.  The reference runs the modalities of the visual encoder one after
 * the other (models/Encoder.py:47-59: `for i in range(self.num_feats)`); each stream's Linear(2048 -> 512) is 120 big
 * output tiles at 128 videos -- half an MI355X.  Between nacf_wide_group_begin() and nacf_wide_group_flush(stream),
 * nacf_linear_fwd / nacf_linear_bwd_data calls that the wide-wave-tile kernel serves (csrc/gemm_bf16_wide.hpp: bf16x3
 * mode, registered weight image, K % 64 == 0, K >= 128, no reduce split, enough tiles) are QUEUED and the flush launches
 * the queued problems of each kind as one grid; every other call launches at once, as without a group.  Each problem's
 * arithmetic is exactly that of its own launch (bit-identical results).  Contract while a group is open: the queued
 * calls' operands and outputs are untouched until the flush, and none of them reads what another one (or anything
 * launched in between) writes.  The flush returns the number of GEMMs it launched (>= 0) or a negative error code.
 * nacf_gemm_last_kernel() reads "gemm_wide_queued" after a call that was queued.  The queue belongs to the calling host
 * thread: begin, the queued calls and the flush come from one thread; calls made by other threads meanwhile launch at once. */
int nacf_wide_group_begin(void);
int nacf_wide_group_flush(nacf_stream_t stream);

/* Which GEMM kernel a call will launch (for profiling / roofline bookkeeping):
 * kind 0 = linear_fwd (also vocab_argmax), 1 = linear_bwd_data, 2 = linear_bwd_weight,
 * with the SAME (M, N, K) the entry point takes.  tile[0] = 128 or 64 (square
 * workgroup tile), splits[0] = number of reduce-dimension splits. */
/* kind: 0 fwd, 1 dX, 2 dW; | 0x100: called with a live-row list; | 0x200: fwd with a transcendental activation */
int nacf_gemm_config(int kind, int M, int N, int K, int* tile, int* splits);

/* ------------------------------------------------------------------------
 * GEMM arithmetic mode and pre-split weight images (round 2)
 * Every GEMM entry point above (and nacf_vocab_argmax / nacf_vocab_lse_fwd below) runs on one of:
 *   NACF_GEMM_F32     v_mfma_f32_16x16x4_f32: exact fp32 products, 157 TFLOP/s peak
 *   NACF_GEMM_BF16X3  v_mfma_f32_16x16x32_bf16 on an EXACT three-term bf16 split of every fp32 operand, six MFMAs
 *                     per product block (dropped terms <= 2^-24 relative): fp32-accurate results -- same parity
 *                     bars, bit-exact greedy tokens -- at up to 2.5 PFLOP/s / 6 = 417 TFLOP/s
 *   NACF_GEMM_BF16    the same MFMA on operands rounded to bf16 (fp32 accumulate, fp32 master weights): the
 *                     throughput mode of BASELINE.json configs[1]; logits within ~1e-2, tokens NOT bit-exact
 * The mode is process-wide; the environment variable NACF_GEMM_MODE (f32 | bf16 | bf16x3) overrides it per call.
 * Operands that are not 16-byte addressable fall back to the fp32 kernels.
 * ---------------------------------------------------------------------- */
#define NACF_GEMM_F32 0
#define NACF_GEMM_BF16 1
#define NACF_GEMM_BF16X3 3
#define NACF_GEMM_DEFAULT_MODE NACF_GEMM_BF16X3
int nacf_gemm_set_mode(int mode);
int nacf_gemm_get_mode(void);
/* template name of the GEMM kernel the calling thread launched last (matches rocprofv3's kernel names) */
const char* nacf_gemm_last_kernel(void);

/* Weight images.  A weight matrix is the P operand of its forward GEMM (y = x W^T) and, transposed, of its dX GEMM
 * (dX = dZ W), for every row tile of both: in the bf16 modes it is converted / split ONCE per optimiser step into
 * bf16 image planes (ns = 1 or 3 planes, `plane_elems` elements apart) and the GEMMs copy those.  The images are
 * k-TILE-MAJOR, zero-padded to whole 32-wide k-tiles (a layout private to this library: the 64 bytes a row
 * contributes to one k-tile sit next to its neighbours', so a wave's load is 1 KB of whole cache lines):
 *   img  [s][(k / 32) * N + n][k % 32] = term s of W[n][k]     ceil(K/32) * 32 * N elements per plane
 *   imgT [s][(n / 32) * K + k][n % 32] = term s of W[n][k]     ceil(N/32) * 32 * K elements per plane
 *   nacf_wimage_register   : the matrix W [N, K] (row pitch ldw) has these images (either may be NULL).  The forward
 *                            image also serves any run of whole rows of W (packed q|k|v used slice-wise); the
 *                            transposed one is found by exact address / shape;
 *   nacf_wimage_unregister : forget every image whose fp32 source starts inside [w_base, w_base + n_elems);
 *   nacf_wimage_refresh    : rebuild the images listed in a DEVICE table of descriptors from the current fp32 values
 *                            (one launch, n_tiles = sum of ceil(N/32)*ceil(K/32) workgroups; capturable).
 * Registration is host-side bookkeeping only; the caller owns all buffers and must refresh after changing weights
 * (models/seq2seq.py does so at every forward entry). */
typedef struct nacf_wimage_desc {
  const float* w;        /* [N, ld] fp32 source */
  uint16_t* img;         /* forward image planes or NULL */
  uint16_t* imgT;        /* transposed image planes or NULL */
  int64_t ld, plane, planeT;
  int32_t N, K;
  int32_t tile0;         /* index of this matrix' first 32x32 tile in the launch */
  int32_t tiles_k;       /* ceil(K / 32) */
} nacf_wimage_desc;
int nacf_wimage_register(const float* w, int N, int K, int64_t ldw, const uint16_t* img, int64_t plane_elems,
                         const uint16_t* imgT, int64_t planeT_elems, int ns);
int nacf_wimage_unregister(const float* w_base, int64_t n_elems);
int nacf_wimage_refresh(const nacf_wimage_desc* table, int n_desc, int n_tiles, int ns, nacf_stream_t stream);

/* Backward of the fused epilogue: from dY produce dZ (grad of the pre-bias
 * GEMM output) and, when ep->residual != NULL, dR (+= when accumulate_dR).
 * `ep` must be the struct used in forward (preact required when act != NONE).
 * dZ may alias dY.
 * rs (optional live-row list, the one the forward GEMM took): only the live rows are walked -- the dead rows of dZ are
 * left UNTOUCHED (the dX / dW GEMMs that read dZ take the same list), the dead rows of dR receive zeros (what the row mask
 * gives; nothing is read for them) unless accumulate_dR.  NULL = every row. */
int nacf_epilogue_bwd(const float* dY, int64_t lddy, float* dZ, int64_t lddz,
                      float* dR, int64_t lddr, int accumulate_dR,
                      int M, int N, const nacf_epilogue* ep, const nacf_rowset* rs, nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * Visual encoder tail + feature fusion  
 * ---------------------------------------------------------------------- */
/* HighWay gate mix + dropout, models/Encoder.py:19-25,65:
 *   out = dropout(G*H + (1-G)*T),  TG = [T | G] packed [rows, 2D]. */
int nacf_highway_mix_fwd(const float* H, const float* TG, float* out, int rows, int D,
                         float p_drop, uint32_t salt, const uint64_t* rng_state,
                         nacf_stream_t stream);
/* given dOut: dH_direct = d*G ; dP = [ d*(1-G)*(1-T^2) | d*(H-T)*G*(1-G) ]  (pre-activation grads) */
int nacf_highway_mix_bwd(const float* dOut, const float* H, const float* TG,
                         float* dH, float* dP, int rows, int D,
                         float p_drop, uint32_t salt, const uint64_t* rng_state,
                         nacf_stream_t stream);

/* BatchNorm1d over the flattened rows of one modality and temporal concat,
 * models/joint_representation.py:40-51.  x is [B, F, D]; the result is written
 * to out[b, f_off + f, :] of a [B, M_total, D] memory.
 * training != 0: batch statistics (biased var), running stats updated with
 * `momentum` (unbiased var), save_mean/save_invstd [D] kept for backward.
 * training == 0: running statistics.  ws: nacf_bn_workspace(rows, D) bytes. */
size_t nacf_bn_workspace(int rows, int D);
int nacf_bn_concat_fwd(const float* x, float* out, int B, int F, int D, int M_total, int f_off,
                       const float* weight, const float* bias,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked,
                       float* save_mean, float* save_invstd,
                       int training, float momentum, float eps,
                       void* ws, size_t ws_bytes, nacf_stream_t stream);
/* dOut is the gradient wrt the [B, M_total, D] memory; dx is [B, F, D]. */
int nacf_bn_concat_bwd(const float* dOut, const float* x, float* dx, int B, int F, int D,
                       int M_total, int f_off, const float* weight,
                       const float* save_mean, const float* save_invstd,
                       float* dweight, float* dbias, float beta,
                       void* ws, size_t ws_bytes, nacf_stream_t stream);

/* Every modality of one joint representation in the same launches (models/joint_representation.py:40-51 loops over the
 * modalities; each of the BatchNorm kernels is a latency-bound chain of row loads, so two modalities side by side cost
 * what one does).  n_mod <= 4; the array arguments are HOST arrays of n_mod entries (device pointers / ints), an array
 * that is absent for every modality may be NULL.  Same results as n_mod calls of nacf_bn_concat_fwd / _bwd.
 * Data-parallel form (see below): stats_global [2][n_mod][D] = (sum | squared deviations) of the GLOBAL batch with
 * n_total[i] rows behind modality i (HOST array) replaces the local statistics passes (= nacf_bn_concat_fwd_sync per
 * modality); sums_global [n_mod][2][D] = the all-reduced (sum dy | sum dy*xhat) replaces the local backward sums and
 * dweight / dbias are not touched (= nacf_bn_concat_bwd_sync).  NULL / NULL otherwise.
 * ws: n_mod * nacf_bn_workspace(rows, D) bytes. */
int nacf_bn_concat_fwd_multi(int n_mod, const float* const* x, float* out, int B, const int* F, int D, int M_total,
                             const int* f_off, const float* const* weight, const float* const* bias,
                             float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                             float* const* save_mean, float* const* save_invstd,
                             int training, float momentum, float eps,
                             const float* stats_global, const int64_t* n_total,
                             void* ws, size_t ws_bytes, nacf_stream_t stream);
int nacf_bn_concat_bwd_multi(int n_mod, const float* dOut, const float* const* x, float* const* dx, int B, const int* F, int D,
                             int M_total, const int* f_off, const float* const* weight,
                             const float* const* save_mean, const float* const* save_invstd,
                             float* const* dweight, float* const* dbias, float beta,
                             const float* sums_global, const int64_t* n_total,
                             void* ws, size_t ws_bytes, nacf_stream_t stream);

/* Data-parallel ("synchronised") BatchNorm: models/joint_representation.py:43-45 computes batch statistics over ALL
 * B*F rows of the batch; with the batch sharded over ranks the same two-pass statistics are formed over the global batch
 * by exchanging one [D] vector per pass.  The library knows nothing about processes: the caller
 * all-reduces (sum) the vectors between the calls.
 *   nacf_bn_sync_stat(x, .., sum_global = NULL)  -> out[d] = sum_r x[r, d]                      (all-reduce -> S)
 *   nacf_bn_sync_stat(x, .., sum_global = S)     -> out[d] = sum_r (x[r, d] - S[d] / n_total)^2  (all-reduce -> Q)
 *   nacf_bn_concat_fwd_sync(.., S, Q, n_total)   normalises with mean = S/n, var = Q/n (running_var: Q/(n-1))
 *   nacf_bn_sync_bwd_stat  -> sums2 = [sum dy | sum dy*xhat] over the LOCAL rows (all-reduce -> global), and
 *                             dbias / dweight (+)= the local sums (the gradient all-reduce adds the ranks up)
 *   nacf_bn_concat_bwd_sync(.., sums2_global, n_total)   dx = w*invstd*(dy - sum_dy/n - xhat*sum_dyx/n)
 * n_total = rows of ALL ranks.  ws: nacf_bn_workspace(rows, D) bytes. */
int nacf_bn_sync_stat(const float* x, int rows, int D, const float* sum_global, int64_t n_total, float* out, void* ws,
                      size_t ws_bytes, nacf_stream_t stream);
/* Forward statistics from ONE exchange instead of two: every rank computes (sum, squared deviations about its OWN mean)
 * with two nacf_bn_sync_stat calls (the second with its local sum and local row count), the ranks all-GATHER the
 * [2][n_mod][D] vectors, and this merges them exactly about the global mean (parallel-variance formula, ranks in fixed
 * order): out [2][n_mod][D] = (global sum | global squared deviations), the operands of nacf_bn_concat_fwd_sync.
 * gathered: [world][rank_stride] floats, rank r's [2][n_mod][D] statistics first; rows_per_rank: HOST array [n_mod] (every
 * rank must hold the same number of rows).  rank_stride >= 2 n_mod D + n_mod: the n_mod floats behind a rank's statistics are
 * the row counts IT holds (they travel in the same all-gather); a rank whose count differs from rows_per_rank makes the merged
 * statistics NaN (the step fails on every rank at once, no extra collective, no host read) and sets *ragged_flag (optional
 * device int32, sticky) for the host to read at its next sync point.  rank_stride == 2 n_mod D: no check. */
int nacf_bn_sync_merge(const float* gathered, int world, int n_mod, int D, const float* rows_per_rank, int64_t rank_stride, float* out,
                       int32_t* ragged_flag, nacf_stream_t stream);
/* The local halves of the data-parallel statistics for every modality at once:
 * nacf_bn_sync_local_multi      loc  [2][n_mod][D] = (sum | squared deviations about this rank's OWN mean) -- gathered by
 *                               the ranks and merged by nacf_bn_sync_merge (3 launches instead of 4 per modality);
 * nacf_bn_sync_bwd_local_multi  sums [n_mod][2][D] = (sum dy | sum dy*xhat) over this rank's rows, also accumulated (beta)
 *                               into the LOCAL dbias | dweight (= nacf_bn_sync_bwd_stat per modality, 2 launches).
 * ws: n_mod * nacf_bn_workspace(rows, D) bytes. */
int nacf_bn_sync_local_multi(int n_mod, const float* const* x, int B, const int* F, int D, float* loc,
                             void* ws, size_t ws_bytes, nacf_stream_t stream);
int nacf_bn_sync_bwd_local_multi(int n_mod, const float* dOut, const float* const* x, int B, const int* F, int D, int M_total,
                                 const int* f_off, const float* const* save_mean, const float* const* save_invstd,
                                 float* sums, float* const* dweight, float* const* dbias, float beta,
                                 void* ws, size_t ws_bytes, nacf_stream_t stream);
int nacf_bn_concat_fwd_sync(const float* x, float* out, int B, int F, int D, int M_total, int f_off, const float* weight,
                            const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                            float* save_mean, float* save_invstd, float momentum, float eps, const float* sum_global,
                            const float* sqdev_global, int64_t n_total, nacf_stream_t stream);
int nacf_bn_sync_bwd_stat(const float* dOut, const float* x, int B, int F, int D, int M_total, int f_off,
                          const float* save_mean, const float* save_invstd, float* sums2, float* dweight, float* dbias,
                          float beta, void* ws, size_t ws_bytes, nacf_stream_t stream);
int nacf_bn_concat_bwd_sync(const float* dOut, const float* x, float* dx, int B, int F, int D, int M_total, int f_off,
                            const float* weight, const float* save_mean, const float* save_invstd, const float* sums2_global,
                            int64_t n_total, nacf_stream_t stream);

/* mean over the time axis: out[b, :] = mean_t x[b, t, :]   (x is [B, T, D]);
 * models/Predictor.py:29, models/Decoder.py:137, models/Encoder.py:51 */
int nacf_mean_time_fwd(const float* x, float* out, int B, int T, int D, nacf_stream_t stream);
/* dx[b, t, :] (+)= (dOut[b, :] + dOut2[b, :]) / T ; dOut2 may be NULL (the mean has two consumers in the model -- length
 * head and decoder input enhancement: their gradients are added here, not by a kernel of their own) */
int nacf_mean_time_bwd(const float* dOut, const float* dOut2, float* dx, int B, int T, int D, int accumulate,
                       nacf_stream_t stream);

/* row-wise log_softmax for narrow rows (length head, N <= 1024),
 * models/Predictor.py:30; in == out allowed. */
int nacf_log_softmax_rows(const float* in, float* out, int rows, int N, nacf_stream_t stream);
/* d(in) = dOut - exp(out) * rowsum(dOut) */
int nacf_log_softmax_rows_bwd(const float* dOut, const float* out, float* dIn, int rows, int N,
                              nacf_stream_t stream);
/* nn.KLDivLoss() legacy 'mean': loss = sum(t*(log t - x))/numel, misc/crit.py:223.
 * loss_out: optional device float[1]; dX (optional) = -t * gscale[0]*scale / numel,
 * gscale: optional device float[1] (upstream gradient). */
int nacf_kldiv_mean(const float* x, const float* t, float* loss_out, float* dX,
                    const float* gscale, float scale, int rows, int N, nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * Decoder  
 * ---------------------------------------------------------------------- */
/* BertEmbeddings, models/bert.py:70-96:
 *   out[r,l,:] = dropout(LN(word[tok[r,l]] + pos[l] + cat[category[r/vdiv % vmod]] + add[r/vdiv % vmod]))
 * category / cat_emb / additional may be NULL.  `additional` is [n_video, D]
 * (the mean-pooled memory of enhance_input=2, models/Decoder.py:136-137).
 * xhat (normalised, pre-affine) and rstd [R*L] are saved when non-NULL. */
int nacf_embed_ln_fwd(const int64_t* tokens, const int64_t* category, const float* additional,
                      const float* word_emb, const float* pos_emb, const float* cat_emb,
                      const float* ln_w, const float* ln_b, float* out,
                      float* xhat, float* rstd,
                      int R, int L, int D, int vdiv, int vmod, float eps,
                      float p_drop, uint32_t salt, const uint64_t* rng_state,
                      nacf_stream_t stream);
/* Backward: dE (grad of the pre-LN sum, [R*L, D]) plus LN weight/bias grads. */
size_t nacf_embed_ln_bwd_workspace(int R, int L, int D);
int nacf_embed_ln_bwd(const float* dOut, const float* xhat, const float* rstd,
                      const float* ln_w, float* dE, float* dln_w, float* dln_b, float beta,
                      int R, int L, int D, float p_drop, uint32_t salt, const uint64_t* rng_state,
                      void* ws, size_t ws_bytes, nacf_stream_t stream);
/* Deterministic scatter of dE into the embedding tables (fixed summation order):
 *   dword[tok] += dE rows (PAD row left untouched: padding_idx, models/bert.py:55); the special
 *                 ids 1..5 (<unk> <bos> <eos> <mask> <vis>) label thousands of rows each and are
 *                 summed chunk-parallel, every other id by the workgroup of its first occurrence
 *   dpos[l]    += sum_r dE[r,l]
 *   dadd[v]     = sum over rows/positions of video v ; dcat[c] += sum of those over the videos of category c
 * ws: nacf_embed_scatter_bwd_workspace(R, L, D, n_video) bytes. */
size_t nacf_embed_scatter_bwd_workspace(int R, int L, int D, int n_video);
int nacf_embed_scatter_bwd(const float* dE, const int64_t* tokens, const int64_t* category,
                           float* dword, float* dpos, float* dcat, float* dadd,
                           int R, int L, int D, int V, int n_cat, int n_video, int vdiv, int vmod,
                           void* ws, size_t ws_bytes, nacf_stream_t stream);

/* Generic LayerNorm over the last dimension (with_layernorm: models/bert.py:189,196-199,237,244-245;
 * norm_type=ln: models/joint_representation.py:21,47):
 *   out[orow] = (row_tokens[r] == PAD ? 0 : dropout(LN(x[r]), p_drop)),
 *   orow = (r / seg_in) * seg_out + seg_off + r % seg_in   (seg_in = seg_out = rows, seg_off = 0: in place layout;
 *   otherwise a modality's [B, F, D] rows land in their slice of the [B, sum F, D] memory).
 * xhat [rows, D] / rstd [rows] are saved for backward when non-NULL. */
int nacf_layernorm_fwd(const float* x, const float* ln_w, const float* ln_b, float* out, float* xhat, float* rstd,
                       int rows, int D, float eps, int seg_in, int seg_out, int seg_off,
                       float p_drop, uint32_t salt, const uint64_t* rng_state, const int64_t* row_tokens,
                       nacf_stream_t stream);
size_t nacf_layernorm_bwd_workspace(int rows, int D);
int nacf_layernorm_bwd(const float* dOut, const float* xhat, const float* rstd, const float* ln_w, float* dX,
                       float* dln_w, float* dln_b, float beta, int rows, int D, int seg_in, int seg_out, int seg_off,
                       float p_drop, uint32_t salt, const uint64_t* rng_state, const int64_t* row_tokens,
                       void* ws, size_t ws_bytes, nacf_stream_t stream);

/* Multi-head attention core, models/bert.py:150-179:
 *   S = Q K^T / sqrt(dk) ; S[key masked] = -1e7 ; P = softmax(S) ; O = P V
 * Q: [R, Lq, *] row stride ldq (head h at column h*dk); K, V: [n_kv, Lk, *];
 * query row r reads kv row (r / kv_div) % kv_mod.
 * key_tokens: optional [n_kv... R, Lk] int64 (self-attention: the decoder
 * tokens; key masked when token == PAD, models/Decoder.py:13-22);
 * causal != 0 adds the strict upper-triangular mask (models/Decoder.py:24-39); causal = 1 + w with w > 0 also masks the
 * keys k <= q - w (--watch w, models/Decoder.py:27-29: a query sees itself and the w - 1 tokens before it).
 * probs: optional [H, R, Lq, Lk] output (models/bert.py:179). */
int nacf_attention_fwd(const float* Q, int64_t ldq, const float* K, int64_t ldk,
                       const float* V, int64_t ldv, float* O, int64_t ldo,
                       const int64_t* key_tokens, int causal, float* probs,
                       int R, int H, int Lq, int Lk, int dk, int kv_div, int kv_mod,
                       nacf_stream_t stream);
/* dQ [R,Lq,*], dK/dV [n_kv, Lk, *]; each kv row sums, in a fixed order, the
 * contributions of the query rows mapped to it. */
int nacf_attention_bwd(const float* Q, int64_t ldq, const float* K, int64_t ldk,
                       const float* V, int64_t ldv, const float* dO, int64_t lddo,
                       float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV, int64_t lddv,
                       const int64_t* key_tokens, int causal,
                       int R, int n_kv, int H, int Lq, int Lk, int dk, int kv_div, int kv_mod,
                       nacf_stream_t stream);

/* The same with attention_probs_dropout_prob (models/bert.py:135,169: P = dropout(softmax(S)) before P V; the returned
 * `probs` are the dropped ones, as upstream).  Mask of element ((r * H + h) * Lq + q) * Lk + k from the device Philox stream
 * {seed, step} = rng_state[0..1] and `salt`, keep-scale 1 / (1 - p); p_drop = 0 is nacf_attention_fwd / _bwd.  p_drop > 0
 * runs the LDS-tile kernels (Lq, Lk, dk must fit 160 KB of LDS). */
int nacf_attention_fwd_dropout(const float* Q, int64_t ldq, const float* K, int64_t ldk,
                               const float* V, int64_t ldv, float* O, int64_t ldo,
                               const int64_t* key_tokens, int causal, float* probs,
                               int R, int H, int Lq, int Lk, int dk, int kv_div, int kv_mod,
                               float p_drop, uint32_t salt, const uint64_t* rng_state, nacf_stream_t stream);
int nacf_attention_bwd_dropout(const float* Q, int64_t ldq, const float* K, int64_t ldk,
                               const float* V, int64_t ldv, const float* dO, int64_t lddo,
                               float* dQ, int64_t lddq, float* dK, int64_t lddk, float* dV, int64_t lddv,
                               const int64_t* key_tokens, int causal,
                               int R, int n_kv, int H, int Lq, int Lk, int dk, int kv_div, int kv_mod,
                               float p_drop, uint32_t salt, const uint64_t* rng_state, nacf_stream_t stream);

/* masked row mean of the last layer (the `embs` output, models/bert.py:301):
 * out[r,:] = sum_l y[r,l,:] / count(tokens[r,:] != PAD) */
int nacf_masked_mean_fwd(const float* y, const int64_t* tokens, float* out, int R, int L, int D,
                         nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * Vocabulary projection + loss (training)  
 * ---------------------------------------------------------------------- */
/* In place: logits[rows, ld] -> log-probs (torch.log_softmax, models/seq2seq.py:103).
 * Also emits, per row: lse, argmax (for Word Acc, misc/crit.py:86-98), and
 * the log-prob of labels[row] (for the NLL / perplexity, misc/crit.py:62-114);
 * labels may be NULL. */
/* skip_pad_rows != 0: rows whose label is PAD are left untouched (they were not projected
 * because the caller used a live-row list built from the labels). */
int nacf_vocab_logsoftmax_fwd(float* logits, int64_t ld, int rows, int V,
                              const int64_t* labels, float* lse, int64_t* argmax,
                              float* label_logp, int skip_pad_rows, nacf_stream_t stream);
/* Criterion tail, misc/crit.py:40-45,107-121: the weighted total and the running meters in ONE launch.
 *   total[0]          = sum_t coef[t] * slab[t * stride]                    (t < n_terms)
 *   meters[m_dst[j]] += m_scale[j] * slab[m_src[j]]                         (j < n_meters, in order)
 * `slab` holds the per-term result vectors side by side (e.g. the 5 outputs of nacf_nll_reduce per
 * pass, the KL scalar).  Backward: gslab[t * stride] = coef[t] * gtotal[0], every other entry 0. */
int nacf_loss_combine(const float* slab, int n_terms, int stride, const float* coef, float* total,
                      const int32_t* m_dst, const int32_t* m_src, const float* m_scale, int n_meters,
                      float* meters, nacf_stream_t stream);
int nacf_loss_combine_bwd(const float* gtotal, const float* coef, int n_terms, int stride, float* gslab,
                          nacf_stream_t stream);
/* The same tail with its producers folded in: ONE launch forward, ONE backward (misc/crit.py:62-114,214-239 behind the fused
 * vocabulary loss).  Forward: per pass i < n_pass the five scalars of nacf_nll_reduce over rows[i] rows -> slab[slot[i] * stride ..
 * + 4]; the legacy KLDivLoss mean of nacf_kldiv_mean over kl_total elements -> slab[kl_slot * stride] (kl_x == NULL: none); then
 * exactly nacf_loss_combine.  Bit-identical to nacf_nll_reduce(_multi) + nacf_kldiv_mean + nacf_loss_combine.  Backward: gslab as
 * nacf_loss_combine_bwd, and kl_dx[e] = -kl_t[e] * coef[kl_slot] * gtotal[0] / kl_total (nacf_kldiv_mean's gradient form).
 * Limits of the one-launch form (NACF_EINVAL beyond them; the separate entry points have none): n_terms * stride <= 128 slab floats,
 * n_meters <= 64. */
typedef struct nacf_crit_tail {
  int32_t n_pass;                 /* <= 4 */
  const float* label_logp[4];     /* device, rows[i] entries each */
  const int64_t* argmax[4];
  const int64_t* labels[4];
  int32_t rows[4];
  int32_t exclude[4];             /* != 0: <mask> labels do not count in the accuracy meter (misc/crit.py:88-90) */
  int32_t slot[4];                /* term index of pass i in the slab */
  const float* kl_x;              /* log-probabilities of the length head, or NULL */
  const float* kl_t;              /* target distribution */
  int32_t kl_total;               /* elements (the legacy 'mean' divides by all of them) */
  int32_t kl_slot;
} nacf_crit_tail;
int nacf_crit_tail_fwd(const nacf_crit_tail* tail, float* slab, int n_terms, int stride, const float* coef, float* total,
                       const int32_t* m_dst, const int32_t* m_src, const float* m_scale, int n_meters, float* meters,
                       nacf_stream_t stream);
int nacf_crit_tail_bwd(const nacf_crit_tail* tail, const float* gtotal, const float* coef, int n_terms, int stride, float* gslab,
                       float* kl_dx, nacf_stream_t stream);

/* Reduce per-row results to the scalars the criterion reports:
 * out[0] = -sum_{label!=PAD} logp[label]      (token-SUM NLL, misc/crit.py:82)
 * out[1] = #(argmax == label) over the accuracy set, out[2] = |accuracy set|
 *          (label != PAD, and != MASK when exclude_mask: misc/crit.py:88-90)
 * out[3] = sum of gathered logp over label != PAD, out[4] = count (perplexity) */
int nacf_nll_reduce(const float* label_logp, const int64_t* argmax, const int64_t* labels,
                    int rows, int exclude_mask, float* out5, nacf_stream_t stream);
/* dlogits = (exp(logp) - onehot(label)) * gscale[0]*scale for rows with label != PAD, else 0.
 * gscale: optional device float[1] (upstream gradient of the loss). In place on logp allowed. */
int nacf_xent_bwd(const float* logp, int64_t ld, float* dlogits, int64_t ldd, int rows, int V,
                  const int64_t* labels, const float* gscale, float scale, int skip_pad_rows,
                  nacf_stream_t stream);
/* Training vocabulary projection with the soft-max statistics fused into the GEMM epilogue (replaces nacf_linear_fwd +
 * nacf_vocab_logsoftmax_fwd for models/__init__.py:83 + F.log_softmax + nn.NLLLoss, misc/crit.py:62-114):
 *   logits[r, :] = hidden[r, :] W^T + bias     (stored raw, row pitch ldl % 4 == 0, live rows of `rs` only)
 *   lse[r] = log sum_n exp(logits[r, n]);  argmax[r];  label_logp[r] = logits[r, labels[r]] - lse[r]
 * `ws`: nacf_vocab_argmax_workspace(rows, V) bytes.  Backward: nacf_xent_bwd_lse on the raw logits. */
int nacf_vocab_lse_fwd(const float* hidden, int64_t ldh, const float* W, int64_t ldw, const float* bias, int rows,
                       int V, int K, float* logits, int64_t ldl, const int64_t* labels, float* lse, int64_t* argmax,
                       float* label_logp, void* ws, size_t ws_bytes, const nacf_rowset* rs, nacf_stream_t stream);
/* dlogits = (exp(logits - lse[row]) - onehot(label)) * gscale[0]*scale for rows with label != PAD, else 0. */
int nacf_xent_bwd_lse(const float* logits, int64_t ld, const float* lse, float* dlogits, int64_t ldd, int rows, int V,
                      const int64_t* labels, const float* gscale, float scale, int skip_pad_rows,
                      nacf_stream_t stream);
/* Several decoding passes back to back in one [n_pass * rows_per_pass, .] batch (the two NACF passes share one
 * projection launch): the same per pass, in ONE launch each.  exclude_mask / out5 / gscales: HOST arrays of n_pass (<= 4)
 * entries (out5[i], gscales[i]: device pointers; pass i covers rows [i * rows_per_pass, (i + 1) * rows_per_pass)). */
int nacf_nll_reduce_multi(const float* label_logp, const int64_t* argmax, const int64_t* labels, int rows_per_pass,
                          int n_pass, const int* exclude_mask, float* const* out5, nacf_stream_t stream);
int nacf_xent_bwd_lse_multi(const float* logits, int64_t ld, const float* lse, float* dlogits, int64_t ldd,
                            int rows_per_pass, int n_pass, int V, const int64_t* labels, const float* const* gscales,
                            float scale, int skip_pad_rows, nacf_stream_t stream);
/* Generic log_softmax backward for wide rows (when the caller consumes the
 * log-probs with its own criterion): dlogits = dlogp - exp(logp)*rowsum(dlogp) */
int nacf_vocab_logsoftmax_bwd(const float* dlogp, int64_t ldg, const float* logp, int64_t ld,
                              float* dlogits, int64_t ldd, int rows, int V, nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * NA decoding  
 * ---------------------------------------------------------------------- */
/* One fused "project + softmax + max" step, decoding/algorithms.py:7-15,149:
 * for every row of hidden [rows, K]: idx = argmax_v (h W^T), prob = max softmax.
 * The [rows, V] logits are never materialised.  Then the bookkeeping of
 * decoding/algorithms.py:154-155,140,262-265 is applied in the same pass:
 *   pad_tokens[row] == PAD      -> (PAD, 1.0)
 *   zero_mask_prob && idx==MASK -> prob = 0          (coarse-grained template pass)
 *   update_mask != NULL         -> tokens/probs only overwritten where update_mask != 0
 * `rows` (optional live-row list over the hidden rows): only those slots are projected and
 * written, e.g. the slots re-masked in this iteration.
 * ws: nacf_vocab_argmax_workspace(rows, V) bytes. */
size_t nacf_vocab_argmax_workspace(int rows, int V);
int nacf_vocab_argmax(const float* hidden, int64_t ldh, const float* W, int64_t ldw, const float* bias,
                      int rows, int V, int K,
                      const int64_t* pad_tokens, int zero_mask_prob, const uint8_t* update_mask,
                      int64_t* tokens, float* probs, void* ws, size_t ws_bytes,
                      const nacf_rowset* live_rows, nacf_stream_t stream);

/* predict_length_beam + canvas, decoding/na_generate.py:35-50,116-135:
 * beam[b,j] = clamp(top-k index of pred_length[b,:] (descending) + bias, 4, max_len-1);
 * beam_max[0] = max over all (device int32, read back once by the host). */
int nacf_length_beam(const float* pred_length, int B, int max_len, int lbs, int length_bias,
                     int32_t* beam, int32_t* beam_max, nacf_stream_t stream);
/* the gold-length beam of opt['load_generated_captions'], decoding/na_generate.py:25-26,118-122:
 * gold[b] = # non-PAD of tgt_tokens[b, 0..T); beam[b,j] = clamp(gold[b] - (lbs-1)/2 + j, 4, max_len-1), j < lbs;
 * beam_max[0] = max over all. */
int nacf_length_beam_gold(const int64_t* tgt_tokens, int B, int T, int max_len, int lbs,
                          int32_t* beam, int32_t* beam_max, nacf_stream_t stream);
/* tokens[b*lbs+j, l] = l < beam[b,j] ? MASK : PAD ; width Lp */
int nacf_canvas_init(const int32_t* beam, int rows, int Lp, int64_t* tokens, nacf_stream_t stream);

/* the canvas under opt['load_generated_captions'], decoding/na_generate.py:42-50:
 * tokens[b*lbs+j, l] = l < beam[b,j] ? (tgt_tokens[b,l] == PAD ? MASK : tgt_tokens[b,l]) : PAD ; tgt_tokens [rows/lbs, T]
 * (a slot l >= T inside a candidate's length starts as MASK) */
int nacf_canvas_init_gold(const int32_t* beam, const int64_t* tgt_tokens, int T, int rows, int lbs, int Lp,
                          int64_t* tokens, nacf_stream_t stream);

/* select_worst + re-mask, decoding/algorithms.py:206-215,255-260: per row,
 * n = max(1, num_mask_lut[seq_len(row)]) slots of lowest score = probs*teacher
 * are flagged in mask_out and set to MASK in tokens.  seq_len = # non-PAD of
 * pad_tokens (the initial canvas).  mode 1: mask = (tokens == MASK) instead
 * (the counter==1 step with coarse templates, algorithms.py:250-253);
 * mode 2: mask = tokens != MASK && pad_tokens != PAD  (visual_mask, :292). */
int nacf_select_mask(const float* probs, const float* teacher_probs, const int64_t* pad_tokens,
                     const int32_t* num_mask_lut, int mode, int rows, int Lp,
                     int64_t* tokens, uint8_t* mask_out, nacf_stream_t stream);

/* small index helpers of the decode loop (flat arrays of n elements):
 *   token_replace: tokens[i] == from -> to        (<mask> -> <vis>, algorithms.py:137-138)
 *   teacher_probs: out[i] = pad_tokens[i] == PAD ? 1 : exp(label_logp[i])   (algorithms.py:197-203)
 *   init_probs:    probs[i] = pad_tokens[i] == PAD ? 1 : 0                  (algorithms.py:294-295)
 *   apply_mask:    tokens[i] = mask[i] ? value : tokens[i] */
int nacf_token_replace(int64_t* tokens, int64_t n, int64_t from, int64_t to, nacf_stream_t stream);
int nacf_teacher_probs(const float* label_logp, const int64_t* pad_tokens, float* out, int64_t n,
                       nacf_stream_t stream);
int nacf_init_probs(const int64_t* pad_tokens, float* probs, int64_t n, nacf_stream_t stream);
int nacf_apply_mask(int64_t* tokens, const uint8_t* mask, int64_t value, int64_t n, nacf_stream_t stream);
/* Left2Right / EasyFirst bookkeeping (algorithms.py:299-324,371-393):
 * mask_rank:  rank[r,l] = index of slot l among the <mask> slots of row r, -1 elsewhere;
 *             counts[0] = max <mask> count of a row, counts[1] = total count (device int32[2]);
 * select_rank: mask = (cur <= rank < cur+q), tokens[mask] = MASK   (select_left);
 * easy_first_update: per row, the min(q, remaining) <mask> slots with the largest
 *             new_probs take (new_tokens, new_probs)   (select_most_confidence). */
int nacf_mask_rank(const int64_t* tokens, int rows, int Lp, int32_t* rank, int32_t* counts,
                   nacf_stream_t stream);
int nacf_select_rank(const int32_t* rank, int cur, int q, int rows, int Lp, int64_t* tokens,
                     uint8_t* mask_out, nacf_stream_t stream);
int nacf_easy_first_update(int64_t* tokens, float* probs, const int64_t* new_tokens,
                           const float* new_probs, int q, int rows, int Lp, nacf_stream_t stream);

/* score = sum_l log(probs*teacher) / len^alpha ; best = argmax_j ; out[b,:] = tokens[b*lbs+best,:]
 * decoding/na_generate.py:66-77.  cand_lprobs optional [rows, Lp] output. */
int nacf_best_candidate(const int64_t* tokens, const float* probs, const float* teacher_probs,
                        const int32_t* beam, float alpha, int B, int lbs, int Lp,
                        int64_t* out_tokens, int32_t* best_idx, float* cand_lprobs,
                        nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * AR beam search  (models/Beam.py:68-130, models/Translator.py:94-161)
 * ---------------------------------------------------------------------- */
/* One step (t = 1 .. max_len-1) of Beam.advance for B instances at once.
 * logp [B*n_bm, ldl]: log-probs of the next word for every hypothesis (row b*n_bm+k).
 * Device state owned by the caller: seqs int64 [B, n_bm, max_len] (<bos> at [b,0,0], PAD elsewhere
 * initially), scores [B, n_bm], finished list fin_scores/fin_len [B, want] + fin_tokens
 * [B, want, max_len] + fin_count [B], done int32 [B] (zero-initialised).
 * Semantics: t == 1 ranks beam 0 only; hypotheses whose last word is <eos> are masked with
 * -1e20; flat top-n_bm over beam x vocab (ties: lower flat index); hypotheses are re-ordered
 * by back-pointer and extended; every emitted <eos> (in beam order) is appended to the finished
 * list; an instance is done once `want` are finished, or at t == max_len-1 (then, if none
 * finished, all beams are appended).  n_active (optional device int32[1]) = #instances not done. */
int nacf_beam_step(const float* logp, int64_t ldl, int B, int n_bm, int V, int t, int max_len, int want,
                   int64_t* seqs, float* scores, float* fin_scores, int32_t* fin_len,
                   int64_t* fin_tokens, int32_t* fin_count, int32_t* done, int32_t* n_active,
                   nacf_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimiser  
 * ---------------------------------------------------------------------- */
/* clip_grad_value_(+-clip) (misc/run.py:260) then Adam with L2 weight decay
 * (misc/optim.py:61-62) over flat buffers.  g is first multiplied by
 * grad_scale (1/world_size after the RCCL all-reduce).  step_count: device
 * int64[1], incremented by this call; lr: device float[1]. */
int nacf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   int64_t n, const float* lr, int64_t* step_count,
                   float beta1, float beta2, float eps, float weight_decay,
                   float grad_clip, float grad_scale, nacf_stream_t stream);
/* torch.optim.RMSprop as misc/optim.py:52-60 constructs it (alpha 0.99, eps 1e-8, momentum 0, not centred, L2 weight decay in the
 * gradient) over the flat buffers, behind the same elementwise clip (misc/run.py:260) and gradient scale as nacf_adam_step:
 *   g = clip(grad * grad_scale) + wd * p;  sq = alpha * sq + (1 - alpha) g^2;  p -= lr[0] * g / (sqrt(sq) + eps)
 * zero_grad != 0 leaves grad zeroed (the step engine's next zero_grad, folded into the walk). */
int nacf_rmsprop_step(float* param, float* grad, float* square_avg, int64_t n, const float* lr, float alpha, float eps,
                      float weight_decay, float grad_clip, float grad_scale, int zero_grad, nacf_stream_t stream);
/* The same update over a PART of the flat buffers (data-parallel training updates the parameters whose gradient
 * bucket has been reduced while the other bucket is still in flight).  `bump` is a bit set: 1 = increment step_count first
 * (exactly one part of a step does), 2 = leave the gradient ZEROED behind (the next step's optimizer.zero_grad() of
 * misc/run.py:254 folded into this walk: the step engine then skips its fill of the gradient buffer). */
int nacf_adam_step_part(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                        int64_t n, const float* lr, int64_t* step_count,
                        float beta1, float beta2, float eps, float weight_decay,
                        float grad_clip, float grad_scale, int bump, nacf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NACF_HIP_H */
