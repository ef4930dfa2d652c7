/*
 * texocr.h -- C ABI of the MI355X-native engine for TeXOCR's OCRModel.generate() hot path.
 *
 * The reference (olibridge01/TeXOCR) is pure Python; it has no FFI/plugin interface, so the drop-in
 * boundary is its nn.Module call surface.  Each entry point below replaces one reference callable
 * (file:line relative to the reference tree); texocr_amd/model.py binds them through ctypes and presents
 * the reference's own class/method names.  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - every *_dev pointer is DEVICE memory owned by the caller; the engine owns its weight copy, KV caches
 *     and workspace (allocated in txo_engine_create, freed in txo_engine_destroy; no allocation in any
 *     encode/decode call -- the diagnostic modes TXO_STAMPS=<file> and txo_profile_enable(e, 1) are the exceptions: they
 *     create their stamp buffer / HIP events on first use);
 *   - all work is enqueued on the caller's hipStream_t (`stream`, may be NULL = default stream) and is
 *     asynchronous except where noted; one engine per (device, stream); a handle is not thread-safe;
 *   - return value: 0 = ok, <0 = error (TXO_E_*); txo_last_error() returns a thread-local message;
 *   - images: float32 NCHW; token ids: int64; encoder output / logits: float32.
 */
#ifndef TEXOCR_H
#define TEXOCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TXO_OK 0
#define TXO_E_INVALID (-1)   /* bad argument / unsupported shape (maps to ValueError)  */
#define TXO_E_STATE (-2)     /* call order / missing weights (maps to RuntimeError)    */
#define TXO_E_HIP (-3)       /* HIP runtime error (maps to RuntimeError)               */

#define TXO_EMBED_PATCH 0
#define TXO_EMBED_HYBRID 1

#define TXO_F32 0            /* parity mode: f32 storage, exact-f32 MFMA               */
#define TXO_BF16 1           /* perf mode: bf16 weights / K,V caches / GEMM operands, f32 accumulate */

typedef struct txo_engine txo_engine;

/* Mirrors the values create_model(config) reads (model/ocr_model.py:113-130, model/encoder.py:172-191,
 * model/decoder.py:148-173) plus capacity limits for the engine-owned buffers. */
typedef struct txo_config {
    int32_t canvas_h;      /* VisionTransformer img_size: max canvas height, pixels (encoder.py:95)       */
    int32_t canvas_w;      /* ... and width; the hybrid factory uses (160, 1008) (encoder.py:183)         */
    int32_t embed;         /* TXO_EMBED_PATCH: PatchEmbedding (encoder.py:11-28) | TXO_EMBED_HYBRID:
                              ResNetV2 [2,4,6] backbone + 1x1 proj, what create_encoder builds (:162-191) */
    int32_t in_channels;   /* image channels (PatchEmbedding in_channels; the hybrid embedder needs 1)    */
    int32_t embed_dim;     /* encoder == decoder width (no enc->dec projection, attention.py:89-91)       */
    int32_t enc_heads, enc_layers, dec_heads, dec_layers;
    int32_t enc_exp, dec_exp;   /* FFN expansion (MLP exp_factor, attention.py:46)                        */
    int32_t vocab;         /* config['vocab_size']                                                        */
    int32_t max_len;       /* config['max_length'] = decoder positional table length, decoder.py:28       */
    int32_t bos, eos, pad; /* config bos_token / eos_token / trg_pad_idx                                  */
    int32_t dtype;         /* TXO_F32 | TXO_BF16                                                          */
    int32_t max_batch;     /* capacity: images per call                                                   */
    int32_t max_tokens;    /* capacity: encoder tokens per image (<= 1 + canvas_h*canvas_w/256); 0 = that maximum */
} txo_config;

/* OCRModel.__init__ / create_model (ocr_model.py:16-32,113-130): allocate the engine. */
int txo_engine_create(const txo_config* cfg, txo_engine** out);
void txo_engine_destroy(txo_engine* e);

/* nn.Module.load_state_dict, one tensor at a time (key layout of OCRModel.state_dict()):
 * `key` is the reference state_dict key, `data` HOST float32, `shape`/`ndim` its shape.  Aliased shared
 * LayerNorm keys (layers.{s}.0.weight for every s, attention.py:200,221) may all be passed; they must
 * carry identical values.  txo_engine_finalize_weights checks completeness, refuses non-finite values (TXO_E_INVALID, the key is
 * named in txo_last_error) and uploads. */
int txo_engine_set_weight(txo_engine* e, const char* key, const float* data, const int64_t* shape, int32_t ndim);
int txo_engine_finalize_weights(txo_engine* e);

/* VisionEncoder.forward (encoder.py:128-152): img_dev [B,C,H,W] -> enc_out_dev [B, 1+(H/16)(W/16), D].
 * Non-finite input: weights are checked (txo_engine_finalize_weights refuses NaN / inf); pixels are not.  A non-finite pixel makes the
 * rows of ITS OWN image unspecified -- encoder rows, logits, and token ids that are unspecified but always inside the vocabulary (the
 * reference would return NaN logits and argmax's pick among them) -- and touches no other image of the batch: rows never interact. */
int txo_encode(txo_engine* e, const float* img_dev, int32_t B, int32_t C, int32_t H, int32_t W,
               float* enc_out_dev, void* stream);

/* Start a decode over `enc_dev` [B,N,D] (the `enc=` kwarg of decoder.generate / net, decoder.py:56,103):
 * projects the cross-attention K/V of every decoder layer once (attention.py:125-126), clears the
 * self-attention cache, position <- 0, current token <- bos. */
int txo_decode_begin(txo_engine* e, const float* enc_dev, int32_t B, int32_t N, void* stream);

/* One position of Transformer.forward in KV-cached form (decoder.py:41-67): feeds tok_in_dev[B] (NULL =
 * the engine's current token: bos, or the previous step's argmax) at position t (t must equal the number
 * of positions decoded since txo_decode_begin, or less to rewind), writes the position's logits
 * [B,vocab] to logits_out_dev (may be NULL) and their argmax to tok_out_dev[B] (may be NULL).  Token ids outside [0, vocab) are
 * forced into the table (the reference's nn.Embedding raises IndexError; the Python facade checks and raises too). */
int txo_decode_step(txo_engine* e, const int64_t* tok_in_dev, int32_t t, float* logits_out_dev,
                    int64_t* tok_out_dev, void* stream);

/* Transformer.forward over a whole prefix in ONE pass (reference model/decoder.py:41-67: token + position embedding, the
 * decoder stack with a causal self attention, final LayerNorm, logits for EVERY position) -- the multi-position form of
 * txo_decode_step.  tokens_dev: int64 [B][t] row-major (B = the batch of the session opened by txo_decode_begin), positions
 * 0..t-1, 1 <= t <= cfg.max_len.  logits_out_dev: float [B][t][vocab] or NULL.  Side effect: the self-attention K/V cache holds
 * rows 0..t-1 afterwards, so txo_decode_step(e, tok, t, ...) continues behind it.  This is what decoder.net(x, mask, enc=) maps to
 * (teacher-forced logits), and what the sliding window of AutoRegressiveDecoder.generate (decoder.py:99-100) costs per token. */
int txo_decode_prefill(txo_engine* e, const int64_t* tokens_dev, int32_t t, float* logits_out_dev, void* stream);

/* The `mask` argument of decoder.generate / decoder.net (model/decoder.py:95-101,112: a (B, T0) bool over the start tokens, padded
 * with True for every generated token; model/attention.py:130-155: energy filled with -FLT_MAX where query or key is masked).
 * mask_dev: uint8 [B][cols] on the device, 0 = padding; positions >= cols are not padding; NULL clears the mask.  Applies to the
 * txo_decode_step and txo_decode_prefill calls of the session opened by txo_decode_begin: a padded position is never attended by a
 * query that is not padding.  Rows of padded positions themselves are computed but unspecified (the reference softmaxes them
 * uniformly over all keys; nothing downstream reads them). */
int txo_decode_set_key_mask(txo_engine* e, const uint8_t* mask_dev, int32_t cols, void* stream);

/* OCRModel.generate (ocr_model.py:46-66) + AutoRegressiveDecoder.generate (decoder.py:77-122), greedy:
 * encode, then up to max_len steps; stops early only when EVERY row contains `eos` (pass eos < 0 for
 * eos_tok=None).  tokens_out_dev is [B, max_len] int64 (row stride max_len); *n_steps_out (HOST) receives
 * the number of valid columns -- the reference returns output[:, :n_steps].  logits_out_dev (may be
 * NULL) is [B, max_len, vocab].  max_len may exceed cfg.max_len: the reference then slides its window (decoder.py:99-100:
 * every further token sees the last cfg.max_len tokens at positions re-indexed from 0) and so does this call -- the first
 * cfg.max_len positions through the KV cache, each later token through one multi-position forward of its window
 * (txo_decode_prefill's pass).  Synchronises the stream before returning. */
int txo_generate(txo_engine* e, const float* img_dev, int32_t B, int32_t C, int32_t H, int32_t W,
                 int32_t max_len, int32_t eos, int64_t* tokens_out_dev, int32_t* n_steps_out,
                 float* logits_out_dev, void* stream);

/* Same loop over a caller-provided encoder output (decoder.generate(start_tokens=[bos], enc=enc)). */
int txo_generate_from_enc(txo_engine* e, const float* enc_dev, int32_t B, int32_t N, int32_t max_len,
                          int32_t eos, int64_t* tokens_out_dev, int32_t* n_steps_out, float* logits_out_dev,
                          void* stream);

/* Beam search over txo_generate's loop -- a BUILD EXTENSION: the reference has no beam search (BASELINE config 5 asks for
 * k = 5), so parity is anchored only at beams = 1 (== greedy).  Score = sum of log_softmax(logits) of the chosen tokens
 * (not length-normalised); a beam that has emitted eos is finished and continues with eos at no cost; the loop stops when
 * every beam of every image is finished.  Needs cfg.max_batch >= B * beams, beams <= 8.  tokens_out [B, max_len] receives
 * the best beam per image; scores_out [B, beams] (may be NULL) the final scores, best first; all_tokens_out
 * [B * beams, max_len] (may be NULL) every beam. */
int txo_generate_beam(txo_engine* e, const float* img_dev, int32_t B, int32_t C, int32_t H, int32_t W, int32_t beams,
                      int32_t max_len, int32_t eos, int64_t* tokens_out_dev, float* scores_out_dev,
                      int64_t* all_tokens_out_dev, int32_t* n_steps_out, void* stream);

/* Token selection for the following decode steps / generate calls.  mode 0 (default): greedy argmax.  mode 1:
 * the reference's sampler (decoder.py:104-108 + utils.topk, utils.py:85-91): keep the `topk` largest logits
 * (the reference uses int((1 - 0.9) * vocab) = 99 for vocab 1000), softmax(logits / temp), one multinomial draw,
 * from a counter-based RNG keyed by (`seed`; row of the batch, position): reproducible, independent of the decode path and of
 * how the engine splits the batch into row ranges; a different stream than torch.multinomial. */
int txo_set_sampling(txo_engine* e, int32_t mode, int32_t topk, float temp, uint64_t seed);

/* Where a decode stops -- the eos handling of txo_generate / txo_generate_from_enc (AutoRegressiveDecoder.generate, decoder.py:97-118).
 * TXO_STOP_GLOBAL (default) is the reference: rows keep producing tokens after their eos and the loop breaks only when EVERY row contains
 * eos (decoder.py:115-116).  TXO_STOP_ROW is a BUILD EXTENSION: a row that has produced eos is finished -- every later token
 * of that row in tokens_out is cfg.pad -- and the loop still ends at the position at which the last row produced its eos, so *n_steps_out
 * and every row's tokens up to and including its first eos are exactly TXO_STOP_GLOBAL's.  What changes is the cost: beyond 128 rows (one launch per
 * stage) the live rows of a row range are compacted to its front every few positions and the launches shrink with them.  Rows whose
 * BOS equals eos are finished from the start.  Not combined with logits_out (a finished row's logits are unspecified: no compaction then)
 * and not applied to txo_generate_beam (a finished beam already costs nothing there). */
#define TXO_STOP_GLOBAL 0
#define TXO_STOP_ROW 1
int txo_set_stop_mode(txo_engine* e, int32_t mode);

/* Timing hooks for bench.py: average duration (ms) of the decode-step cross-attention launches and of
 * the encoder launches recorded with HIP events on the stream the kernels run on, since profiling was last
 * enabled (txo_profile_enable(e, 1) also clears the previous samples); *count = number of launches averaged.  kind: 0 = cross-attention decode kernel,
 * 1 = encoder (whole txo_encode), 2 = whole decode step.
 * on = 1: everything above (adds marker commands around every encode and step -- use in a separate pass); on = 2: only every
 * fourth cross-attention dispatch carries events (bound to the dispatch, no extra commands; safe inside a timed region; samples are
 * kept for the first 16 generate() calls after enabling); on = 3: the persistent decode launch carries events (kind 3 =
 * its duration; the launch-per-stage path is not instrumented in this mode, and modes 1 / 2 make generate() take that path);
 * on = 0: off. */
int txo_profile_enable(txo_engine* e, int32_t on);
int txo_profile_read(txo_engine* e, int32_t kind, double* avg_ms, int64_t* count);

/* Introspection for tests and bench.py.  what = TXO_Q_LAST_PERSISTENT: 1 if the last txo_generate* ran its decode loop as ONE
 * persistent launch (texocr_amd/csrc/persist.h; greedy decode of the reference widths), 0 if it ran one launch per stage
 * (sampling, beam search, profiling modes, other widths, TXO_PERSIST=0, or after a fallback).  TXO_Q_PERSIST_FALLBACKS: how
 * many persistent launches gave up (placement check / bounded spin) and were redone with launches since the engine was created. */
#define TXO_Q_LAST_PERSISTENT 0
#define TXO_Q_PERSIST_FALLBACKS 1
#define TXO_Q_LAST_ROW_RANGES 2   /* row ranges (streams) the last generate decoded on: 1, or 2 beyond 128 images in bf16 (launch path) */
#define TXO_Q_LAST_LATENT 3       /* 1 if the last generate's cross attention ran in latent form (csrc/lat_attn.h: against the raw encoder rows) */
#define TXO_Q_RELOAD_KNOBS 4      /* not a question: re-read the TXO_* development knobs of generate() from the environment (the engine reads them once,
                                   * at creation; tests flip TXO_PERSIST / TXO_LANES on a live engine).  *out = 0 */
#define TXO_Q_LAST_COMPACTIONS 5   /* live-row compactions of the last txo_generate* (TXO_STOP_ROW on the launch path; 0 otherwise) */
int txo_engine_query(txo_engine* e, int32_t what, int64_t* out);

const char* txo_last_error(void);
const char* txo_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TEXOCR_H */
