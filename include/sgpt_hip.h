/*
 * sgpt_hip.h -- C ABI of the MI355X (gfx950) SGPT bi-encoder retrieval hot path.
 *
 * The reference (Muennighoff/sgpt) has NO native layer: the hot path is Python that
 * reaches implicit PyTorch/cuBLAS kernels through HuggingFace `AutoModel`, `torch.mm`
 * and `torch.topk`.  Each entry point below names the reference call site whose device
 * work it replaces (paths relative to /root/reference; HF: = huggingface transformers,
 * the un-vendored dependency that holds the transformer arithmetic).  The Python side
 * (sgpt_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer marked "device" is a HIP device pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); the library allocates only its own workspace and
 *     its packed copy of the weights, both tied to the ctx / model handle;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream) and performs no hidden device synchronisation, except
 *     sgpt_ctx_create / sgpt_model_load / *_destroy / *_free and workspace growth;
 *   - no C++ exception crosses the ABI: functions return SGPT_OK (0) or a negative
 *     sgpt_status; sgpt_last_error(ctx) returns a message for the last failure on ctx;
 *   - calls on one ctx must be issued from one thread onto ONE stream at a time: the activation / scorer workspaces are
 *     per-ctx scratch shared by every model and call on it (use one ctx per stream for concurrent streams);
 *   - a ctx is bound to one HIP device and is re-entrant per ctx, not thread-safe
 *     (the reference runs one single-threaded Python process per GPU,
 *     sentence_transformers/SentenceTransformer.py:275-283).
 */
#ifndef SGPT_HIP_H
#define SGPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGPT_ABI_VERSION 8

typedef int sgpt_status;
#define SGPT_OK 0
#define SGPT_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define SGPT_ERR_HIP (-2)         /* a HIP runtime call failed */
#define SGPT_ERR_MISSING (-3)     /* a required weight tensor was not supplied */
#define SGPT_ERR_OOM (-4)
#define SGPT_ERR_COMM (-6)        /* an RCCL call failed */
#define SGPT_ERR_RANGE (-5)       /* SGPT_F16: a matmul weight does not fit the f16 range, or an activation class is beyond every range shift */

typedef struct sgpt_ctx sgpt_ctx;
typedef struct sgpt_model sgpt_model;

enum { SGPT_F32 = 0, SGPT_BF16 = 1, SGPT_FP8W = 2, SGPT_F16 = 3, SGPT_FP8M = 4 };   /* element types; FP8W / FP8M: model compute_dtype only */
enum { SGPT_ARCH_GPTNEO = 0, SGPT_ARCH_GPTJ = 1, SGPT_ARCH_BLOOM = 2 };
enum { SGPT_POOL_WEIGHTEDMEAN = 0, SGPT_POOL_MEAN = 1, SGPT_POOL_LASTTOKEN = 2, SGPT_POOL_LEARNTMEAN = 3 };
enum { SGPT_COS = 0, SGPT_DOT = 1 };

/* Model hyper-parameters = the fields of HF GPTNeoConfig the forward pass reads
 * (HF:gpt_neo/configuration_gpt_neo.py; values of the SGPT checkpoints in SURVEY.md 8). */
typedef struct {
    int32_t arch;            /* SGPT_ARCH_GPTNEO (SGPT-125M/1.3B/2.7B) | SGPT_ARCH_GPTJ (SGPT-5.8B) | SGPT_ARCH_BLOOM (sgpt-bloom-7b1) */
    int32_t n_layers;
    int32_t d_model;         /* multiple of 128 */
    int32_t n_heads;         /* d_model / n_heads in {64, 128} */
    int32_t d_ffn;           /* multiple of 128 */
    int32_t vocab;
    int32_t max_pos;
    int32_t window;          /* GPT-Neo local-attention window (256) */
    float ln_eps;            /* 1e-5 */
    float attn_scale;        /* 1.0 for GPT-Neo (no 1/sqrt(dh), HF:gpt_neo:110); 1/sqrt(dh) for GPT-J (HF:gptj:148) and BLOOM (HF:bloom:186) */
    int32_t compute_dtype;   /* SGPT_F16 : IEEE half MFMA operands (weights, LN output, q/k/v, probabilities, context, GELU
                                           output), fp32 accumulate/residual/LN/softmax -- the same MFMA rate as bf16 with 3
                                           more mantissa bits: the mode that meets the 1e-3 cosine bar against the fp32
                                           reference.  Range-managed: every operand class of a block (LayerNorm
                                           outputs, q | k | v + attention context, GELU output) is stored under a power-of-two
                                           down-shift (0 by default) that the consuming GEMM undoes on its fp32 accumulators
                                           -- exact, the f16 analogue of the e4m3 scales of SGPT_FP8M.  sgpt_model_load
                                           derives the LayerNorm shifts from the parameters and refuses only weights outside
                                           the format (SGPT_ERR_RANGE); every kernel that rounds an activation to f16 records
                                           |v| >= 32768 in a per-model device word (sgpt_model_range_check), from which
                                           sgpt_model_range_adapt raises the shifts of exactly the classes that overflowed;
                                SGPT_BF16: bf16 MFMA operands, fp32 accumulate/residual/LN/softmax;
                                SGPT_F32 : exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), the parity gate;
                                SGPT_FP8W: the six matmul weights per block are STORED as OCP e4m3fn with one
                                           power-of-two fp32 scale per output channel (the
                                           reference loads sgpt-bloom-7b1 8-bit through bitsandbytes) and
                                           de-quantised -- exactly -- to bf16 per block; arithmetic as SGPT_BF16;
                                SGPT_FP8M: weights stored as SGPT_FP8W, and the four projections of every block COMPUTED in fp8:
                                           v_mfma_f32_16x16x128_f8f6f4 on e4m3 x e4m3 operands at twice the bf16 MFMA rate
                                           (BASELINE configs[4]).  The LayerNorms emit e4m3 codes with one power-of-two scale
                                           per row, the GELU output and the attention context are re-quantised under one
                                           calibrated power-of-two scale per block (sgpt_model_calibrate_begin / _end), scales
                                           are applied to the fp32 accumulators; q / k / V^T, softmax and P.V stay bf16 / fp32,
                                           the residual stream fp32.  Shapes that do not fit the 256x256x256 fp8 tile
                                           run the SGPT_FP8W arithmetic.  Cannot meet the 1e-3 bar (3 mantissa bits): the
                                           tests report max |dcos| and top-10 overlap against the fp32 oracle instead */
    const uint8_t* layer_is_local;  /* host, [n_layers]: 1 = sliding-window layer (HF:gpt_neo:66) */
    int32_t rotary_dim;      /* GPT-J: leading dims of every head that get rotary position embedding (64) */
    int32_t qk_split;        /* SGPT_F16 / SGPT_BF16 only.  1 = split-precision Q / K projection: the LayerNorm output a and the
                                Wq / Wk weights enter the projection as hi + lo pairs of 16-bit values (a_hi.W_hi + a_lo.W_hi +
                                a_hi.W_lo: one GEMM over K' = 3 d on the same MFMA), i.e. to ~2^-22 instead of 2^-11 each; q / k
                                are stored in 16 bits as before.  For GPT-Neo (attention without 1/sqrt(dh), HF:gpt_neo:110) the
                                path LayerNorm -> Wq / Wk -> q / k carries 80 % of the 16-bit deviation at d = 2048:
                                SGPT-1.3B shape goes from 8.2e-4 / 1.09e-3 (cosine / embedding) to well inside the 1e-3 bar, for
                                three times the FLOPs of one of the five projections.  0 (default) = off.  (= precision plan entry
                                SGPT_PC_LN1 = 1 in every block, see sgpt_model_set_precision) */
    int32_t split_weights;   /* SGPT_F16 / SGPT_BF16 only.  1 = also keep [W_hi | W_hi | W_lo] copies of all four matmul weights of
                                every block (3 x the 16-bit weight bytes on top of the plain copy) so that sgpt_model_set_precision
                                can move any operand class of any block to split precision after load.  0 (default) = off */
} sgpt_model_desc;

/* One named fp32 weight tensor under its HF state-dict name
 * (GPT-Neo: "wte.weight", "wpe.weight", "h.0.attn.attention.q_proj.weight", ..., "ln_f.bias";
 *  GPT-J:   "wte.weight", "h.0.attn.q_proj.weight", "h.0.mlp.fc_in.weight", ..., plus the rotary tables
 *           "rotary.sin" / "rotary.cos" fp32[max_pos, rotary_dim/2] = HF create_sinusoidal_positions,
 *           HF:gptj/modeling_gptj.py:47-50, computed by the host so they match the reference bit for bit;
 *  BLOOM:   "word_embeddings.weight", "word_embeddings_layernorm.*", "h.0.self_attention.query_key_value.{weight,bias}"
 *           (fused, head-interleaved [n_head, 3, head_dim] rows, HF:bloom:214 -- de-interleaved by the library),
 *           "h.0.self_attention.dense.*", "h.0.mlp.dense_h_to_4h.*", ..., plus "alibi.slopes" fp32[n_head]
 *           (HF build_alibi_tensor, HF:bloom:62-79)). */
typedef struct {
    const char* name;
    const float* ptr;        /* device, contiguous fp32 */
    int64_t numel;
} sgpt_tensor_view;

/* -- lifetime ------------------------------------------------------------------------- */
int sgpt_abi_version(void);
sgpt_status sgpt_ctx_create(int hip_device, sgpt_ctx** out);
void sgpt_ctx_destroy(sgpt_ctx* ctx);
const char* sgpt_last_error(const sgpt_ctx* ctx);

/* Replaces AutoModel.from_pretrained(...).to(device) (biencoder/beir/beir_dense_retriever.py:123;
 * sentence_transformers/models/Transformer.py:38).  Copies / packs the weights into
 * library-owned device memory (fused [3d,d] QKV; bf16 rounding (RNE) of the six matmul
 * weights per block when compute_dtype == SGPT_BF16); the caller may free its tensors after
 * return. */
sgpt_status sgpt_model_load(sgpt_ctx* ctx, const sgpt_model_desc* desc,
                            const sgpt_tensor_view* tensors, size_t n_tensors, sgpt_model** out);
void sgpt_model_free(sgpt_model* model);

/* SGPT_FP8M activation-scale calibration.  Between _begin and _end every sgpt_encode on the model runs the SGPT_FP8W
 * arithmetic and records max |gelu output| and max |attention context| per block; _end turns the maxima into per-block
 * power-of-two scales (smallest 2^k with margin * max / 2^k <= 448; margin >= 1, default 2) and returns them (host
 * float[2 * n_layers]: the GELU-output scales, then the context scales; may be NULL).  sgpt_model_set_act_scales installs
 * scales computed elsewhere (2 * n_layers powers of two).  A later batch that saturates the e4m3 range raises bit 1 of the
 * flag read by sgpt_range_check. */
sgpt_status sgpt_model_calibrate_begin(sgpt_model* model);
sgpt_status sgpt_model_calibrate_end(sgpt_model* model, float margin, float* scales_out);
sgpt_status sgpt_model_set_act_scales(sgpt_model* model, const float* scales, int32_t n_scales);

/* -- a2+a3+a4: forward + pool --------------------------------------------------------- */
/* Replaces, in ONE call and with no hidden-state D2H:
 *   AutoModel(**tokens, output_hidden_states=True)   beir_dense_retriever.py:204-205 / Transformer.py:72
 *     = HF GPTNeoModel.forward                       HF:gpt_neo/modeling_gpt_neo.py:398-505
 *   hidden_state = all_hidden_states[layeridx]       beir_dense_retriever.py:233
 *   weightedmean / mean / lasttoken pooling          beir_dense_retriever.py:238-282; Pooling.py:99-125
 *   optional F.normalize                             SentenceTransformer.py:248-249
 *
 * Token layout (varlen, no FLOPs on padding): sequence b owns rows
 * [seq_off[b], seq_off[b+1]) of the packed token axis; its seq_len[b] real tokens come
 * first.  seq_off[] are even (ABI <= v5 asked for multiples of 8; any such layout is still valid), seq_off[B] <= T_pad, T_pad is a multiple of 32
 * (ABI <= v7: 128; rows past seq_off[B] are filler: computed by the row-wise kernels, never attended to).  Layouts of at most
 * 4096 rows take the query- / mid-sized kernels (csrc/qgemm.hip: tiles of 32 .. 128 rows, LayerNorm inside the projections); same bits.
 *   ids      device int32[T_pad]  token ids (filler rows: any valid id)
 *   pos      device int32[T_pad]  absolute position id = pad_left[b] + t  (HF:gpt_neo:451 uses
 *                                 the PADDED index arange(S); filler rows: 0)
 *   pad_left device int32[B]      left-padding of the reference batch (0 for right-padded
 *                                 GPT-2 tokenizers); the pooling weight of token t is
 *                                 pad_left[b] + t + 1 (Pooling.py:104-112, padded index)
 *   n_layers_run  number of blocks to run (n_layers for layeridx=-1)
 *   apply_final_ln  1 = ln_f (hidden_states[-1]); 0 = raw residual (hidden_states[i<L])
 *   out      device fp32[B, d_model]
 *   hidden_out  optional device fp32[T_pad, d_model]: the selected hidden state per token
 */
sgpt_status sgpt_encode(sgpt_model* model, const int32_t* ids, const int32_t* pos,
                        const int32_t* seq_off, const int32_t* seq_len, const int32_t* pad_left,
                        int32_t B, int32_t T_pad, int32_t max_alloc_len,
                        int32_t pool_mode, int32_t n_layers_run, int32_t apply_final_ln,
                        int32_t normalize, float* out, float* hidden_out, void* stream);

/* All L+1 hidden states pooled in ONE forward: the `meanmean` / `lasttokenmean` methods
 * (beir_dense_retriever.py:243-257, 284-301; useb_dense_retriever.py:219-302) average the pooled vector of
 * every entry of `all_hidden_states`.  Entry i < L is the input of block i (raw residual stream), entry L is
 * ln_f of the last block's output (HF:gpt_neo:475-505).  Same token layout as sgpt_encode.
 *   out_layers device fp32[n_layers+1, B, d_model] (or NULL); normalize applies to every entry;
 *   out_mean   device fp32[B, d_model] (or NULL): the average of the L+1 entries = the method's embedding. */
sgpt_status sgpt_encode_layers(sgpt_model* model, const int32_t* ids, const int32_t* pos,
                               const int32_t* seq_off, const int32_t* seq_len, const int32_t* pad_left,
                               int32_t B, int32_t T_pad, int32_t max_alloc_len, int32_t pool_mode,
                               int32_t normalize, float* out_layers, float* out_mean, void* stream);

/* Trained position weights for SGPT_POOL_LEARNTMEAN: `position_weights` of
 * models/WeightedMeanPooling.py:21-39 (the reference reads them from 1_WeightedMeanPooling/pytorch_model.bin,
 * useb_dense_retriever.py:253-270).  Token t of a sequence is weighted by weights[pad_left + t]; the weight sum
 * is clamped at 1e-9.  The n fp32 values are copied (device pointer); n must cover the longest padded sequence. */
sgpt_status sgpt_model_set_pool_weights(sgpt_model* model, const float* weights, int32_t n);

/* Cross-encoder scoring: log P(target | prefix) under the LM head for selected token rows.
 * Replaces, for the rows that matter, `F.log_softmax(model(inps)[0], dim=-1)` + `torch.gather` + `argmax`
 * of crossencoder/beir/sgptce.py:233-255.  hidden = the post-ln_f hidden states of sgpt_encode (`hidden_out`,
 * fp32 [T_pad, d_model]); row_idx[i] = the token row whose next-token distribution is asked for, targets[i] the
 * token id to score.  The LM head is the embedding matrix (GPT-Neo, BLOOM: tied) or "lm_head.weight" /
 * "lm_head.bias" when those tensors were passed to sgpt_model_load (GPT-J).  Logits are computed in exact-fp32 MFMA.
 *   out_logprob device fp32[n]; out_greedy device int32[n] (argmax token, first maximum) or NULL. */
sgpt_status sgpt_lm_logprobs(sgpt_model* model, const float* hidden, const int32_t* row_idx, const int32_t* targets,
                             int32_t n, float* out_logprob, int32_t* out_greedy, void* stream);

/* Stand-alone pooling over caller-supplied hidden states (USEB layer sweeps, parity tests).
 * Replaces Pooling.forward (sentence_transformers/models/Pooling.py:99-125,129-164) and
 * CustomEmbedder.embed_batcher's pooling branch (beir_dense_retriever.py:238-282).
 *   hidden device [B,S,d] (fp32 or bf16), mask device int32[B,S] (0/1, any padding side);
 *   weights follow the padded index t+1; weight sum clamped at 1e-9 (Pooling.py:122). */
sgpt_status sgpt_pool(sgpt_ctx* ctx, const void* hidden, int32_t hidden_dtype, const int32_t* mask,
                      int32_t B, int32_t S, int32_t d, int32_t pool_mode, float* out, void* stream);
/* Same with per-position weights (SGPT_POOL_LEARNTMEAN): pos_weights device fp32[>= S]. */
sgpt_status sgpt_pool_learnt(sgpt_ctx* ctx, const void* hidden, int32_t hidden_dtype, const int32_t* mask,
                             int32_t B, int32_t S, int32_t d, const float* pos_weights, float* out, void* stream);

/* Row-wise x / max(||x||_2, 1e-12): torch.nn.functional.normalize(p=2, dim=1)
 * (sentence_transformers/util.py:41-42, 66-70).  out may alias in when out_dtype == SGPT_F32. */
sgpt_status sgpt_l2_normalize(sgpt_ctx* ctx, const float* in, int64_t n, int32_t d,
                              void* out, int32_t out_dtype, void* stream);

/* Row-wise scores of two [n, d] fp32 matrices (ABI v8): out[i] = dot(a[i], b[i]) -- util.pairwise_dot_score,
 * sentence_transformers/util.py:66-76 -- or, cosine != 0, the same sum over rows normalised as sgpt_l2_normalize does --
 * util.pairwise_cos_sim, util.py:79-91 (`pairwise_dot_score(normalize_embeddings(a), normalize_embeddings(b))`).
 * The reference's own cases: tests/test_util.py:69-76. */
sgpt_status sgpt_pairwise_scores(sgpt_ctx* ctx, const float* a, const float* b, int64_t n, int32_t d, int32_t cosine,
                                 float* out, void* stream);

/* fp32 -> bf16 / f16 (RNE) element-wise; used to keep a corpus shard in HBM in the scorer's 16-bit operand format
 * (out_dtype SGPT_BF16 | SGPT_F16; normalised embeddings are in [-1, 1], inside either range). */
sgpt_status sgpt_f32_to_16(sgpt_ctx* ctx, const float* in, int64_t numel, void* out, int32_t out_dtype, void* stream);
sgpt_status sgpt_f32_to_bf16(sgpt_ctx* ctx, const float* in, int64_t numel, void* out, void* stream);

/* Range guards.  The reference's fp32 CPU path cannot overflow, so the 16-bit / 8-bit paths must never do so silently.
 * sgpt_model_range_check: *flagged = the model's guard word since the last reset --
 *     bit 0: an sgpt_encode* call on this SGPT_F16 model rounded an activation of magnitude >= 32768 (or +-inf) to f16: the
 *            results of the affected calls are not trustworthy; call sgpt_model_range_adapt and re-run them;
 *     bit 1: an SGPT_FP8M GELU output saturated its e4m3 codes;  bit 2: an SGPT_FP8M attention context did (re-calibrate).
 *   Per model: a flag raised by one model is never attributed to another one sharing the ctx.  Synchronises `stream` (one
 *   4-byte read-back); the Python host calls it once per encode_ids().
 * sgpt_model_range_adapt (SGPT_F16): reads the magnitudes the flagged launches recorded per (block, operand class), raises
 *   the power-of-two shift of each such class so that the recorded maximum lands at <= 16384, clears bit 0 and returns the
 *   number of classes raised through *n_raised (0: nothing recorded).  Shifts never decrease; a class that would need more
 *   than 2^40 returns SGPT_ERR_RANGE.  Bumps sgpt_ctx_generation (captured graphs carry the factors as kernel arguments).
 * sgpt_model_get/set_range_shifts: the 4 * n_layers exponents (per block: LayerNorm-1 output, q | k | v, LayerNorm-2 output,
 *   GELU output) -- to pin the shifts found on one run for reproducible embeddings on the next (a shift changes results only
 *   through f16 subnormals: |v| * 2^-k < 6.1e-5).  A LayerNorm shift below the bound sgpt_model_load derived from this
 *   checkpoint's LayerNorm parameters is refused (the LayerNorm kernels carry no run-time tracker), and a pooled embedding
 *   that is not finite raises bit 0 whatever produced it.
 * sgpt_range_check: the same bit 0 for the ctx-level stand-alone ops (sgpt_linear with f16 output). */
sgpt_status sgpt_model_range_check(sgpt_model* model, int32_t* flagged, int32_t reset, void* stream);
sgpt_status sgpt_model_range_adapt(sgpt_model* model, int32_t* n_raised, void* stream);
sgpt_status sgpt_model_get_range_shifts(sgpt_model* model, int32_t* shifts, int32_t n);
sgpt_status sgpt_model_set_range_shifts(sgpt_model* model, const int32_t* shifts, int32_t n);
sgpt_status sgpt_range_check(sgpt_ctx* ctx, int32_t* flagged, int32_t reset, void* stream);

/* Precision plan (ABI v6): which operand classes of which blocks enter their MFMAs as split-precision pairs.
 * The reference runs fp32 everywhere (beir_dense_retriever.py:123,204-205; Transformer.py:38,72), so it has no checkpoint
 * on which 11-bit operands are not enough; this path does (real GPT-Neo checkpoints carry outlier channels: a row whose
 * energy sits in one or two entries puts their whole relative rounding error on the dot product).  A split class is stored
 * as hi = round16(v) and lo = round16(v - hi) -- a [hi | lo | hi] row of 3 K that the UNCHANGED 16-bit GEMM contracts
 * against [W_hi | W_hi | W_lo] (a_hi.W_hi + a_lo.W_hi + a_hi.W_lo: the product to ~2^-22 instead of 2^-11 per operand, at
 * a third of the 16-bit MFMA rate and ~5 x the exact-fp32 MFMA mode).
 *   plan: host int32[n_layers][SGPT_PREC_CLASSES], per block
 *     SGPT_PC_LN1  LayerNorm-1 output -> Q / K / V projection: 0 = plain, 1 = Q and K split (V from the hi block), 2 = Q, K, V,
 *                  3 = Q and K with the ACTIVATION alone split ([a_hi | a_lo] . [W_hi | W_hi]: K' = 2 d, half the extra FLOPs of 1)
 *     SGPT_PC_ATT  inside the attention: q, k, V^T and the probabilities as hi + lo pairs (logits and P.V as three MFMA
 *                  passes each; head_dim 64 / 128, GPT-Neo and BLOOM)
 *     SGPT_PC_CTX  attention context -> out-projection
 *     SGPT_PC_LN2  LayerNorm-2 output -> fc1 (GPT-J's parallel block reads ln_1's output: needs LN1 != 0 there)
 *     SGPT_PC_H    GELU output -> fc2
 *   Every class but LN1 = 1 / 3 (qk_split is enough) and ATT (activations only) needs sgpt_model_desc.split_weights.  All ones (LN1 = 2) = the "f16x3" mode: embeddings within
 *   ~1e-5 of the fp32 reference on any checkpoint the f16 RANGE guard accepts.  Changing the plan bumps sgpt_ctx_generation.
 * sgpt_model_precision_probe_begin / _end: between the two calls every sgpt_encode on the model also records, per block and
 *   class (LayerNorm-1 output, attention context, LayerNorm-2 output, GELU output: 4 * n_layers floats), the largest crest
 *   factor max|v| / rms(v) over the rows of the operand -- 4-10 for a well-conditioned row, ~sqrt(K / 2) when two entries carry
 *   the row.  The host turns them into a plan (sgpt_amd/model.py: classes above 12 / 20 are split, and with them the
 *   attention of the block; by default any such class moves the whole model to f16x3).  The recording launches are extra
 *   kernels: results of the probed calls are unchanged.
 * sgpt_split16: fp32 rows [n, d] -> 16-bit rows [n, 3 d] for a split-precision SCORER on the same kernels: layout 0 =
 *   [hi | lo | hi] (documents), layout 1 = [hi | hi | lo] (queries); sgpt_scores / sgpt_score_topk over d' = 3 d then give
 *   q_hi.c_hi + q_hi.c_lo + q_lo.c_hi.  (Normalised embeddings that two channels dominate lose up to 4e-4 of cosine to the
 *   16-bit corpus format alone.)
 * sgpt_linear_split: sgpt_linear (epi 0 | 1 | 4, 16-bit output) with the split store epilogue: hi at out, lo at out + lo_delta,
 *   a second hi at out + hi2_delta when != 0 (element offsets; ldo = leading dimension of out): kernel-level tests. */
#define SGPT_PREC_CLASSES 5
enum { SGPT_PC_LN1 = 0, SGPT_PC_ATT = 1, SGPT_PC_CTX = 2, SGPT_PC_LN2 = 3, SGPT_PC_H = 4 };
sgpt_status sgpt_model_set_precision(sgpt_model* model, const int32_t* plan, int32_t n);
sgpt_status sgpt_model_get_precision(sgpt_model* model, int32_t* plan, int32_t n);
/* sgpt_model_release_split_weights: give the [W_hi | W_hi | W_lo] copies of sgpt_model_desc.split_weights back (3 x the 16-bit
 *   weight bytes: +0.5 GB at SGPT-125M, +35 GB at GPT-J-6B shape) once the plan is known not to need them -- what the host does when
 *   the probe of precision='auto' settles on plain operands.  Refused while the installed plan reads the out-projection / fc1 /
 *   fc2 copies (context, LayerNorm-2, GELU classes); the Q / K / V copy stays when a LayerNorm-1 entry (precise_qk) reads it; afterwards
 *   sgpt_model_set_precision refuses plans that would, exactly as for a model loaded without split_weights.  bytes_freed may
 *   be NULL.  (The reference keeps one fp32 copy of the weights: `AutoModel.from_pretrained`, beir_dense_retriever.py:123.) */
sgpt_status sgpt_model_release_split_weights(sgpt_model* model, int64_t* bytes_freed);
sgpt_status sgpt_model_precision_probe_begin(sgpt_model* model);
sgpt_status sgpt_model_precision_probe_end(sgpt_model* model, float* crest_out /* host float[4 * n_layers] or NULL */);
 /* sgpt_row_crest: the probe's statistic for caller-owned 16-bit rows [n, d] (leading dimension ld): max over the rows of
 *   max|v| / rms(v), returned to the host (synchronises `stream`).  The search host uses it on the gathered query embeddings --
 *   identical on every rank -- to decide whether a 16-bit corpus is kept as split pairs. */
sgpt_status sgpt_row_crest(sgpt_ctx* ctx, const void* x, int32_t dtype, int64_t n, int32_t d, int64_t ld, float* crest_out,
                           void* stream);
sgpt_status sgpt_split16(sgpt_ctx* ctx, const float* in, int64_t n, int32_t d, int32_t layout, void* out, int32_t out_dtype,
                         void* stream);
sgpt_status sgpt_linear_split(sgpt_ctx* ctx, int32_t dtype, int32_t epi, const void* A, const void* W, const float* bias,
                              void* out, int64_t ldo, int64_t lo_delta, int64_t hi2_delta, int32_t M, int32_t N, int32_t K,
                              void* stream);

/* hipGraph support.  sgpt_ctx_generation changes whenever a library-owned buffer that launched kernels point into is
 * re-allocated (workspace growth, a larger learnt-pooling table): a graph captured at generation g must be re-captured
 * (or refused) once the value differs.  sgpt_ctx_reserve grows the encoder / scorer workspaces to at least the given
 * sizes up front so that no growth happens while graphs are alive. */
uint64_t sgpt_ctx_generation(const sgpt_ctx* ctx);
sgpt_status sgpt_ctx_reserve(sgpt_ctx* ctx, size_t encode_bytes, size_t score_bytes);

/* fp8 weight storage (SGPT_FP8W) building blocks, exported for parity tests and for callers that keep
 * their own quantised checkpoints.  w device fp32[rows, cols] (cols % 4 == 0) -> codes uint8[rows, cols]
 * (OCP e4m3fn, round-to-nearest-even) + scale fp32[rows], scale = the smallest power of two with
 * max|w_row| / scale <= 448.  De-quantisation code * scale is exact in bf16 and in fp32. */
sgpt_status sgpt_fp8_quantize_rows(sgpt_ctx* ctx, const float* w, int64_t rows, int64_t cols,
                                   uint8_t* codes, float* scale, void* stream);
sgpt_status sgpt_fp8_dequantize_rows(sgpt_ctx* ctx, const uint8_t* codes, const float* scale, int64_t rows,
                                     int64_t cols, void* out, int32_t out_dtype, void* stream);

/* -- a7: scoring ---------------------------------------------------------------------- */
/* Replaces cos_sim / dot_score = torch.mm(a, b.T) (sentence_transformers/util.py:24-63;
 * beir.util, imported at custommodels/exact_search.py:9): out[i][j] = <a_i, b_j>, NaN kept.
 * For cosine the caller normalises first (sgpt_l2_normalize), exactly as util.py:41-43.
 *   a device [na,d], b device [nb,d], both of `dtype` (fp32: exact fp32 MFMA; bf16 / f16: 16-bit MFMA,
 *   fp32 accumulate); d multiple of 4 (fp32) / 8 (16-bit); out device fp32[na, ldo], ldo >= nb, ldo%4==0. */
sgpt_status sgpt_scores(sgpt_ctx* ctx, const void* a, const void* b, int32_t dtype,
                        int64_t na, int64_t nb, int32_t d, float* out, int64_t ldo, void* stream);

/* -- a7+a8: fused chunked score + top-k ------------------------------------------------ */
/* Replaces the body of the corpus-chunk loop of DenseRetrievalExactSearch.search
 * (custommodels/exact_search.py:96-108): scores = q . corpus^T ; scores[isnan] = -1 ;
 * topk(min(k, n)) -- and, when run_val/run_idx already hold `n_run` entries per query from
 * earlier chunks, the heapq.nlargest merge of :121-132 (the set of the k best of old+new).
 *   q       device [nq,d] of `dtype`;  corpus device [N,d] of `dtype`
 *   idx_base  global index of corpus row 0 (rank offset when the corpus is sharded)
 *   run_val device fp32[nq,k], run_idx device int64[nq,k]: in = running best (first n_run
 *           columns valid), out = new running best, sorted by descending score
 *           (ties: ascending index); unused tail = (-inf, -1).
 *   returns through *n_out (host) the number of valid columns = min(k, n_run + N).
 * Long corpora (bf16 or f16, d % 64 == 0, d >= 128, k <= 1024, N >= two chunks): only the first chunk's scores are materialised;
 * later chunks double in length and their GEMM epilogue appends just the scores above the query's running k-th
 * best to a candidate list that a small merge kernel folds into the running top-k.  The result is identical to
 * the materialised loop; a candidate-list overflow raises a device flag on which a materialised recomputation of
 * the call is predicated (its kernels exit at once otherwise), so the call stays asynchronous on `stream`. */
sgpt_status sgpt_score_topk(sgpt_ctx* ctx, const void* q, const void* corpus, int32_t dtype,
                            int32_t nq, int64_t N, int32_t d, int32_t k, int64_t idx_base,
                            float* run_val, int64_t* run_idx, int32_t n_run, int32_t* n_out,
                            void* stream);

/* sgpt_score_topk_refined (ABI v7): the reference's fp32 scoring + top-k (`torch.mm(a, b.T)`, util.py:41-43,63; `torch.topk`,
 * exact_search.py:96-108) -- fp32 scores of the returned documents, the fp32 top-k set -- at the speed of the 16-bit scorer.
 *   q         device fp32 [nq, d]; corpus32 device fp32 [N, d]; corpus16 device 16-bit [N, d] = corpus32 rounded (sgpt_f32_to_16
 *             or sgpt_l2_normalize with a 16-bit output); dtype16 SGPT_F16 | SGPT_BF16
 *   stage 1   the filtered 16-bit scorer (sgpt_score_topk on 16-bit copies) takes k' = k + head-room candidates per query;
 *   stage 2   every candidate is re-scored in exact fp32 from corpus32, merged with the running list, top-k (ties: ascending index).
 *   margin    the caller's bound 2 eps on |s16 - s32| for its rows: 2.5e-3 for L2-normalised rows with SGPT_F16 copies (two
 *             roundings of relative 2^-11, Cauchy-Schwarz; subnormal flushes and both fp32 accumulations inside the slack); scale by
 *             max|q| max|d| for un-normalised rows.  When the (k')-th best 16-bit score of a query is not more than `margin` below
 *             its k-th best, outsiders could still reach the fp32 top-k: a device flag is raised and the call's exact-fp32
 *             materialise-and-select pass, every launch predicated on that flag, redoes the chunk -- exact either way, and
 *             asynchronous on `stream` unless fallback_flag_out (host int32*, may be NULL) is given: then the call synchronises and
 *             reports 1 if the exact pass ran, 0 if not, -1 if the call went straight to the exact pass (k' would not fit).
 *   run_val / run_idx / n_run / n_out as sgpt_score_topk (fp32 scores throughout). */
sgpt_status sgpt_score_topk_refined(sgpt_ctx* ctx, const float* q, const float* corpus32, const void* corpus16, int32_t dtype16,
                                    int32_t nq, int64_t N, int32_t d, int32_t k, int64_t idx_base, float margin,
                                    float* run_val, int64_t* run_idx, int32_t n_run, int32_t* n_out, int32_t* fallback_flag_out,
                                    void* stream);

/* k best of m candidate (score, index) pairs per query: the cross-rank / cross-chunk merge
 * (exact_search.py:121-132 heapq.nlargest).  Candidates with idx < 0 are ignored, and so is
 * the candidate whose idx == exclude_idx[q] (the `corpus_id != query_id` rule of :118;
 * exclude_idx may be NULL).
 *   val device fp32[nq,m], idx device int64[nq,m] -> out_val fp32[nq,k], out_idx int64[nq,k]
 *   (sorted descending; tail (-inf,-1)). */
sgpt_status sgpt_topk_merge(sgpt_ctx* ctx, const float* val, const int64_t* idx, int32_t nq,
                            int32_t m, int32_t k, const int64_t* exclude_idx,
                            float* out_val, int64_t* out_idx, void* stream);

/* -- a9 / 8e: multi-GPU exchange steps on RCCL (one process per GPU, xGMI) ---------------------------------------------- */
/* One communicator per ctx.  Rank 0 creates the 128-byte id (ncclGetUniqueId), the host carries it to the other ranks by
 * any means (sgpt_amd/dist.py: one torch.distributed broadcast, the only use of torch.distributed on this path), every rank
 * calls sgpt_comm_init (collective, blocking: ncclCommInitRank).  sgpt_ctx_destroy tears the communicator down.
 *
 * sgpt_allgather_rows replaces the reference's two-step util.mismatched_sizes_all_gather
 * (sentence_transformers/util.py:326-347: all-gather the sizes, pad to the maximum, all-gather the rows): rank r contributes
 * counts[r] rows of row_bytes bytes (counts: host int64[world], known to every rank because shards are a pure function of the
 * global length list -- SentenceTransformer.py:159-163, or the token-balanced cut of sgpt_amd/st.py); out device
 * [sum(counts)][row_bytes] receives the blocks in rank order.  Equal counts: ONE ncclAllGather straight into `out`; ragged:
 * padded to the largest block through the exchange workspace and compacted.  Asynchronous on `stream`.
 *
 * sgpt_exchange_topk is the last step of the corpus-sharded search: every rank holds the k best (score, global
 * index) pairs per query of ITS corpus shard (sgpt_score_topk with idx_base = the shard's first global index); the lists are
 * all-gathered (two ncclAllGathers in one group) and the k_out best of the world * k candidates per query are selected by
 * the library's merge kernel on the same stream -- the heapq.nlargest merge of exact_search.py:121-132 including the
 * `corpus_id != query_id` rule (:118) through exclude_idx (device int64[nq] or NULL).  Ties -> lowest index: every rank gets
 * identical results.  val device fp32[nq,k], idx device int64[nq,k] -> out_val fp32[nq,k_out], out_idx int64[nq,k_out].
 *
 * sgpt_fold_gathered_topk is the rank-local half of that step on its own (no communicator needed): gathered_val / gathered_idx
 * device [world][nq][k] in rank order, as an all-gather delivers them -> the same k_out best per query.  For callers that move
 * the lists with their own transport, and for testing the fold of a many-rank world on one GPU. */
#define SGPT_COMM_ID_BYTES 128
sgpt_status sgpt_comm_unique_id(uint8_t* id /* [SGPT_COMM_ID_BYTES], host */);
sgpt_status sgpt_comm_init(sgpt_ctx* ctx, const uint8_t* id, int32_t rank, int32_t world);
sgpt_status sgpt_comm_destroy(sgpt_ctx* ctx);
int32_t sgpt_comm_world(const sgpt_ctx* ctx);     /* 0 = no communicator */
int32_t sgpt_comm_rank(const sgpt_ctx* ctx);
sgpt_status sgpt_allgather_rows(sgpt_ctx* ctx, const void* local, const int64_t* counts, int64_t row_bytes, void* out,
                                void* stream);
/* The same result through the RAGGED branch whatever the counts are (pad to the largest block, gather into the exchange
 * workspace, compact in rank order): lets a world of one -- or equal shards -- exercise the code path unequal shards take. */
sgpt_status sgpt_allgather_rows_padded(sgpt_ctx* ctx, const void* local, const int64_t* counts, int64_t row_bytes, void* out,
                                       void* stream);
sgpt_status sgpt_exchange_topk(sgpt_ctx* ctx, const float* val, const int64_t* idx, int32_t nq, int32_t k, int32_t k_out,
                               const int64_t* exclude_idx, float* out_val, int64_t* out_idx, void* stream);
sgpt_status sgpt_fold_gathered_topk(sgpt_ctx* ctx, const float* gathered_val, const int64_t* gathered_idx, int32_t world,
                                    int32_t nq, int32_t k, int32_t k_out, const int64_t* exclude_idx, float* out_val,
                                    int64_t* out_idx, void* stream);

/* Plain top-k over a materialised score matrix: torch.topk(scores, k, dim=1)
 * (exact_search.py:102-108; util.semantic_search util.py:241).  NaN -> -1 first (:99). */
sgpt_status sgpt_topk(sgpt_ctx* ctx, const float* scores, int32_t nq, int64_t n, int64_t ld,
                      int32_t k, int64_t idx_base, float* out_val, int64_t* out_idx, void* stream);

/* -- the projection GEMM with its fused epilogues, as a stand-alone op ----------------------------------------------- */
/* out = epilogue(A . W^T): the nn.Linear calls of the transformer blocks (HF:gpt_neo:141-143,155,304-309) with the
 * element-wise tail fused -- exactly the kernels sgpt_encode launches, exposed for kernel-level parity tests and for
 * callers that assemble their own blocks.  A device [M,K], W device [N,K] (both row-major, `dtype` SGPT_F32 / SGPT_BF16 /
 * SGPT_F16), fp32 accumulation.
 *   epi 0: out[M,N] = acc (+ bias if given)              out_dtype = dtype (16-bit) or SGPT_F32
 *   epi 1: out[M,N] = gelu_new(acc + bias)               out_dtype = dtype           (HF NewGELUActivation)
 *   epi 2: out[M,N] = resid + acc + bias                 fp32; out may alias resid (the residual stream, in place)
 *   epi 4: out[N,M] = acc (+ bias[n])                    16-bit transposed store (V^T for the attention P.V operand), M % 128 == 0
 * bias device fp32[N]; resid device fp32[M,N].  M % 256 == 0, N % 256 == 0, K % 64 == 0 with at least half a wave of
 * 256x256 tiles take the LDS-DMA throughput kernel, everything else the 128x128 / 64x64 register-staged one; both
 * feed every output element the same MFMA sequence, so the result does not depend on which one ran. */
sgpt_status sgpt_linear(sgpt_ctx* ctx, int32_t dtype, int32_t epi, int32_t out_dtype, const void* A, const void* W,
                        const float* bias, const float* resid, void* out, int32_t M, int32_t N, int32_t K, void* stream);

/* The query- / mid-sized projection kernels of sgpt_encode, stand-alone (ABI v8; csrc/qgemm.hip: layouts of at most 4096 token rows --
 * `SentenceTransformer.encode("one query")`, SentenceTransformer.py:143-146; USEB's 21-sentence batches,
 * useb/useb/useb/evaluators/askubuntu.py:144-148), for kernel-level parity tests.  16-bit operands (dtype SGPT_BF16 | SGPT_F16),
 * fp32 accumulation, W device [N, K] row-major.
 *   A != NULL: A device [M, K] in `dtype`.
 *   x != NULL: the LayerNorm prologue -- A = nn.LayerNorm(K, eps)(x) rounded to `dtype` (x device fp32 [M, K]: the residual stream;
 *              HF:gpt_neo:317-319,385), computed by the projection itself; epi 7 and 1 only.
 *   epi 0: out[M,N] = acc (+ bias if given)         epi 1: out[M,N] = gelu_new(acc + bias)        (out in `dtype`)
 *   epi 2: out[M,N] = resid + acc + bias, fp32 (out may alias resid)
 *   epi 7: the fused Q | K | V projection: columns [0, n_split) -> out[M, n_split] row-major, columns [n_split, N) -> out_vt[N - n_split, M]
 *          (V^T, the attention kernel's P.V operand).
 * Same bits as sgpt_linear (+ the LayerNorm kernel) on the same operands.  SGPT_ERR_INVALID if the shape is not served. */
sgpt_status sgpt_linear_query(sgpt_ctx* ctx, int32_t dtype, int32_t epi, const void* A, const float* x, const float* ln_gamma,
                              const float* ln_beta, float ln_eps, const void* W, const float* bias, const float* resid, void* out,
                              void* out_vt, int32_t n_split, int32_t M, int32_t N, int32_t K, void* stream);

/* The fp8-MFMA building blocks of SGPT_FP8M, stand-alone (kernel-level tests, custom blocks).
 * sgpt_layernorm_fp8: nn.LayerNorm(x)[T,d] -> e4m3fn codes + one power-of-two scale per row (true value = code * scale).
 * sgpt_linear_fp8:    acc = (A8 . W8^T)[m][n] * a_scale[m] * a_scalar * w_scale[n]   (A8 [M,K], W8 [N,K] e4m3fn codes, fp32
 *                     accumulate; a_scale NULL = 1; M, N, K multiples of 256)
 *     epi 0: out 16-bit[M,N] = acc (+ bias if given)          (out_dtype SGPT_BF16 | SGPT_F16)
 *     epi 1: out u8[M,N] = e4m3( gelu_new(acc + bias) / out_scale ), saturating at +-448
 *     epi 2: out fp32[M,N] = resid + acc + bias (out may alias resid)
 *     epi 4: out 16-bit[N,M] = acc (+ bias[n])                (transposed store: V^T) */
sgpt_status sgpt_layernorm_fp8(sgpt_ctx* ctx, const float* x, const float* gamma, const float* beta, int32_t T, int32_t d,
                               float eps, uint8_t* codes, float* row_scale, void* stream);
sgpt_status sgpt_linear_fp8(sgpt_ctx* ctx, int32_t epi, int32_t out_dtype, const uint8_t* A, const float* a_scale,
                            float a_scalar, const uint8_t* W, const float* w_scale, const float* bias, const float* resid,
                            void* out, float out_scale, int32_t M, int32_t N, int32_t K, void* stream);

/* -- measurement ----------------------------------------------------------------------- */
/* bench.py's live roofline: when enabled, every GEMM launched by sgpt_encode /
 * sgpt_scores / sgpt_score_topk is bracketed by hipEvents on the launch stream;
 * sgpt_prof_read synchronises and returns launches, summed milliseconds and summed
 * algorithmic FLOPs (2*M*N*K of the un-padded problem) since the last reset. */
sgpt_status sgpt_prof_enable(sgpt_ctx* ctx, int32_t on);
sgpt_status sgpt_prof_read(sgpt_ctx* ctx, int64_t* launches, double* ms, double* flops, int32_t reset);

/* Per-ctx run-time policies (the library holds no process-global mutable state and reads no environment variable).
 * sgpt_ctx_set_low_latency: on = 1 lets GEMM launches that have fewer 64x64 tiles than workgroup slots (query-sized batches)
 *   split each tile's k range over two groups of waves that run concurrently and add their fp32 accumulators in a fixed
 *   order (launches with >= 6 k-steps of 128 elements per group).  It was worth 16 % on a 16-query SGPT-125M encode before
 *   the query-sized launches got 128-element k-steps (round 3), ~1 % since (0.82 -> 0.81 ms).  Deterministic, but the sum is no longer the k-ascending
 *   one every other kernel produces, so with the mode on an embedding depends (at 16-bit operand-rounding level, <= 5e-4 on
 *   normalised bf16 embeddings) on whether its batch was small enough to take this path.  Off (default), every batch size
 *   produces identical bits.  Returns the previous setting.
 * sgpt_ctx_set_tile_policy: 0 (default) = problems with less than half a wave of 256x256 tiles take the 128x128 / 64x64
 *   register-staged kernel; 1 = keep the 256x256 LDS-DMA kernel wherever the shape allows (kernel-level tests of
 *   single-tile shapes; identical bits either way); 2 = layouts small enough for the query- / mid-sized kernels (csrc/qgemm.hip)
 *   keep the small-tile register-staged kernels of the bulk path (same-box A/Bs, tests; identical bits).  Returns the previous policy. */
 /* sgpt_ctx_set_gemm_cu_cap: n > 0 = the persistent 256x256 projection kernel launches at most n workgroups (rounded down to
 *   a multiple of 8; 0 = one per CU, the default) -- for two contexts that run their calls half a block out of phase on two
 *   streams, so that one pipeline's LayerNorm / attention / embed kernels find free CUs while the other is in its k-loops
 *   (scripts/dual_stream_probe.py).  Results do not depend on it.  Returns the previous value. */
int32_t sgpt_ctx_set_gemm_cu_cap(sgpt_ctx* ctx, int32_t n);
int32_t sgpt_ctx_set_low_latency(sgpt_ctx* ctx, int32_t on);
int32_t sgpt_ctx_set_tile_policy(sgpt_ctx* ctx, int32_t policy);

/* Micro-benchmark of one GEMM launch configuration (library-owned pseudo-random operands, never
 * zeros): average milliseconds per launch over `iters` launches.  epi: 0 store, 1 bias+gelu,
 * 2 bias+residual, 3 score, 4 V^T store (see sgpt_amd/csrc/common.h GemmEpi). */
sgpt_status sgpt_bench_gemm(sgpt_ctx* ctx, int32_t dtype, int32_t epi, int32_t out_dtype,
                            int32_t M, int32_t N, int32_t K, int32_t iters, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* SGPT_HIP_H */
