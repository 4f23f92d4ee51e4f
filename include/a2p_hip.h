/*
 * a2p_hip.h -- C ABI of liba2p_hip.so, the MI355X (gfx950) implementation of
 * audio2photoreal's audio-to-motion diffusion sampling hot path.
 *
 * The reference is pure Python/PyTorch and has NO FFI for this path (SURVEY.md
 * §0, §8b "C-ABI to define (new; nothing to mirror)").  Each entry point below
 * therefore cites the reference *Python* interface it replaces
 * (paths relative to /root/reference); INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *  - tensors are dense row-major fp32 unless stated; int64 for timesteps
 *    (the reference passes torch.long, gaussian_diffusion.py:911);
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing
 *    synchronises the device except a2p_ctx_destroy and a2p_kernel_time_ms;
 *  - return value: 0 = ok, negative = A2P_ERR_* (no exceptions cross the ABI);
 *    a2p_last_error() returns a static thread-local message;
 *  - inputs are borrowed for the duration of the enqueued work, outputs are
 *    caller-allocated (ownership rules of the reference: SURVEY.md §8b
 *    "Ownership / errors / threading").
 */
#ifndef A2P_HIP_H
#define A2P_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A2P_OK 0
#define A2P_ERR_ARG (-1)      /* bad argument / shape (reference: Python assert, gaussian_diffusion.py:286,318-322) */
#define A2P_ERR_STATE (-2)    /* call order: weights not finalized / conditioning not prepared */
#define A2P_ERR_HIP (-3)      /* HIP runtime error */
#define A2P_ERR_NOWEIGHT (-4) /* unknown / missing / mis-sized parameter (reference: load_model asserts, utils/model_util.py:30-38) */
#define A2P_ERR_NONFINITE (-5) /* a2p_check_finite: a denoiser output held inf / nan (16-bit operand overflow, or non-finite inputs / weights) */

#define A2P_FACE 0
#define A2P_POSE 1

#define A2P_PREC_F32 0  /* fp32 operands, v_mfma_f32_16x16x4_f32: the parity mode (<=1e-3 vs CPU reference) */
#define A2P_PREC_BF16 1 /* bf16 operands, fp32 accumulate/statistics/residual: the throughput mode      */

/* cond_drop_prob selector of FiLMTransformer.forward (model/diffusion.py:338-344);
 * A2P_PASS_CFG = ClassifierFreeSampleModel.forward (model/cfg_sampler.py:30-33). */
#define A2P_PASS_COND 0
#define A2P_PASS_UNCOND 1
#define A2P_PASS_CFG 2

#define A2P_SAMPLER_DDIM 0 /* GaussianDiffusion.ddim_sample (gaussian_diffusion.py:667-718) */
#define A2P_SAMPLER_DDPM 1 /* GaussianDiffusion.p_sample    (gaussian_diffusion.py:434-477, noise restored) */

typedef struct a2p_ctx a2p_ctx;

/* Constructor contract of FiLMTransformer (model/diffusion.py:83-99) as produced by
 * utils/model_util.py:49-76, plus the capacity the workspace is sized for. */
typedef struct a2p_config {
  int32_t data_format;      /* A2P_FACE | A2P_POSE                      */
  int32_t nfeats;           /* 256 | 104                                 */
  int32_t latent_dim;       /* 512 | 256   (must be 256 or 512)          */
  int32_t ff_size;          /* 1024                                      */
  int32_t num_layers;       /* 8 | 6                                     */
  int32_t num_heads;        /* 8   (latent_dim/num_heads must be 32|64)  */
  int32_t cond_feature_dim; /* 2038 | 1024                               */
  int32_t max_frames;       /* args.max_seq_length = 600                 */
  int32_t emb_len;          /* 1998 (model/diffusion.py:136)             */
  int32_t keyframe_dim;     /* 104                                       */
  int32_t keyframe_step;    /* 30                                        */
  int32_t precision;        /* A2P_PREC_*                                */
  int32_t max_batch;        /* largest B (samples) of any later call     */
  int32_t reserved;
} a2p_config;

/* ---- lifetime --------------------------------------------------------------- */
int a2p_ctx_create(const a2p_config* cfg, a2p_ctx** out);
int a2p_ctx_destroy(a2p_ctx* ctx);
const char* a2p_last_error(void);
const char* a2p_version(void);

/* ---- weights: nn.Module.load_state_dict (utils/model_util.py:30-38) ----------
 * `name` is the reference state_dict key (SURVEY.md §8b "Weights"), `data` a device
 * fp32 pointer of `numel` elements (copied).  Unknown names that belong to the
 * out-of-scope conditioning producers (audio_model.*, lip_model.*, *.rotary.freqs,
 * transformer.*, tokenizer.*) are accepted and ignored (returns 1). */
int a2p_set_weight(a2p_ctx* ctx, const char* name, const float* data, int64_t numel, void* stream);
/* Checks every hot-path parameter was provided, builds the packed compute-dtype
 * copies, rotary tables and the batch-invariant unconditional K/V caches. */
int a2p_finalize_weights(a2p_ctx* ctx, void* stream);

/* ---- hoisted conditioning (t-independent part of FiLMTransformer.forward,
 *      model/diffusion.py:360-381 + the audio-token K/V of every decoder layer) ----
 * cond_embed : [B, n_tok, cond_feature_dim]  = what encode_audio/encode_lip return
 *              (model/diffusion.py:355-358); n_tok = 1998 for 600 frames, 798 for 240.
 * keyframes  : pose only, [B, n_key, keyframe_dim]   (y["keyframes"])
 * key_mask   : pose only, uint8 [B, n_key], 1 = known (y["mask"][..., ::step]); NULL = all known
 * frames     : T of the motion sequence that will be denoised. */
int a2p_prepare_cond(a2p_ctx* ctx, const float* cond_embed, int32_t batch, int32_t n_tok,
                     const float* keyframes, const uint8_t* key_mask, int32_t n_key,
                     int32_t frames, void* stream);

/* ---- one denoiser evaluation -------------------------------------------------
 * FiLMTransformer.forward / ClassifierFreeSampleModel.forward.
 * x [B, nfeats, 1, T] ; t_orig int64 [B] (already mapped to 0..999, respace.py:140-145);
 * scale fp32 [B] (y["scale"], A2P_PASS_CFG only) ; out [B, T, nfeats]. */
int a2p_denoise_forward(a2p_ctx* ctx, const float* x, const int64_t* t_orig, const float* scale,
                        int32_t pass, float* out, void* stream);

/* ---- fused sampler step: p_mean_variance + ddim_sample / p_sample -------------
 * tables: fp32 [A2P_NTAB, n_steps] rows in the order of a2p_table_id, built by the host
 * from the float64 schedule exactly as _extract_into_tensor does (.float());
 * t_idx int64 [B]: step index into the (respaced) chain; timestep_map int64 [n_steps].
 * noise may be NULL for DDIM with eta == 0.  Outputs: x_next, pred_xstart, both
 * [B, nfeats, 1, T] (x_next may alias x). */
enum a2p_table_id {
  A2P_TAB_POST_COEF1 = 0,
  A2P_TAB_POST_COEF2,
  A2P_TAB_POST_VAR,
  A2P_TAB_POST_LOGVAR,
  A2P_TAB_SQRT_RECIP_ACP,
  A2P_TAB_SQRT_RECIPM1_ACP,
  A2P_TAB_ACP,
  A2P_TAB_ACP_PREV,
  A2P_TAB_SQRT_ACP,
  A2P_TAB_SQRT_1M_ACP,
  A2P_TAB_ACP_NEXT,
  A2P_NTAB
};
int a2p_sample_step(a2p_ctx* ctx, int32_t sampler, const float* x, const int64_t* t_idx,
                    const int64_t* timestep_map, const float* tables, int32_t n_steps,
                    const float* scale, const float* noise, float eta, int32_t clip_denoised,
                    float* x_next, float* pred_xstart, void* stream);

/* ---- stand-alone sampler arithmetic (any model callable on the host side) ------
 * model_out [B, T, C] -> pred_xstart/mean [B, C, 1, T]: GaussianDiffusion.p_mean_variance
 * (gaussian_diffusion.py:305-316) + q_posterior_mean_variance (:235-257). */
int a2p_p_mean_variance(const float* model_out, const float* x, const int64_t* t_idx, const float* tables,
                        int32_t n_steps, int32_t batch, int32_t nfeats, int32_t frames, int32_t clip_denoised,
                        float* pred_xstart, float* mean, void* stream);
/* ddim_sample update (gaussian_diffusion.py:699-717) from pred_xstart. */
int a2p_ddim_update(const float* pred_xstart, const float* x, const int64_t* t_idx, const float* tables,
                    int32_t n_steps, const float* noise, float eta, int32_t batch, int64_t per_sample,
                    float* sample, void* stream);
/* p_sample update (gaussian_diffusion.py:470-476): mean + [t!=0] exp(.5 logvar) noise. */
int a2p_p_sample_update(const float* mean, const int64_t* t_idx, const float* tables, int32_t n_steps,
                        const float* noise, int32_t batch, int64_t per_sample, float* sample, void* stream);
/* _predict_eps_from_xstart (gaussian_diffusion.py:347-351). */
int a2p_eps_from_xstart(const float* x, const float* pred_xstart, const int64_t* t_idx, const float* tables,
                        int32_t n_steps, int32_t batch, int64_t per_sample, float* eps, void* stream);
/* plms_sample arithmetic (gaussian_diffusion.py:990-1041).  eps0 is the newest eps, eps1..3 older ones (NULL when the mode
 * does not read them).  PREDICT writes the Euler predictor x0 sqrt(abar_prev) + sqrt(1-abar_prev) eps0 that the
 * reference feeds to the model at t-1; AB1..AB4 / EULER write the step's "sample" (pred_xstart where t == 0). */
enum a2p_plms_mode { A2P_PLMS_PREDICT = 0, A2P_PLMS_AB1, A2P_PLMS_AB2, A2P_PLMS_AB3, A2P_PLMS_AB4, A2P_PLMS_EULER };
int a2p_plms_update(const float* x, const float* pred_xstart, const int64_t* t_idx, const float* tables, int32_t n_steps,
                    const float* eps0, const float* eps1, const float* eps2, const float* eps3, int32_t mode,
                    int32_t batch, int64_t per_sample, float* sample, void* stream);
/* ddim_reverse_sample update, eta = 0 (gaussian_diffusion.py:781-813). */
int a2p_ddim_reverse_update(const float* pred_xstart, const float* x, const int64_t* t_idx, const float* tables,
                            int32_t n_steps, int32_t batch, int64_t per_sample, float* sample, void* stream);
/* q_sample (gaussian_diffusion.py:215-233). */
int a2p_q_sample(const float* x_start, const int64_t* t_idx, const float* tables, int32_t n_steps,
                 const float* noise, int32_t batch, int64_t per_sample, float* out, void* stream);

/* ---- unit entry points (parity tests of single kernels / one decoder layer) -----
 * FiLMTransformerDecoderLayer.forward (model/modules/transformer_modules.py:178-217):
 * x [N, T, d] updated in place; memory [N, S, d]; t [N, d]; memory2 [N, S2, d] or NULL.
 * Uses the weights of decoder layer `layer`. */
int a2p_decoder_layer_forward(a2p_ctx* ctx, int32_t layer, float* x, const float* memory, const float* t,
                              const float* memory2, int32_t nseq, int32_t frames, int32_t mem_len,
                              int32_t mem2_len, void* stream);
/* C[M,N] = A[M,K] W[N,K]^T + bias on the MFMA GEMM kernel of the context's precision. */
int a2p_gemm(a2p_ctx* ctx, const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N,
             int32_t K, void* stream);
/* softmax(q k^T / sqrt(dh)) v per head; q [N, Tq, d], k/v [N, S, d], out [N, Tq, d]. */
int a2p_attention(a2p_ctx* ctx, const float* q, const float* k, const float* v, float* out, int32_t nseq,
                  int32_t tq, int32_t s, void* stream);

/* ---- measurement (bench.py roofline leg) ----------------------------------------
 * When enabled, every launch of the kernel class `kind` is bracketed by hipEvents on
 * the launch stream; a2p_kernel_time_ms synchronises and returns total ms and count. */
#define A2P_KERNEL_GEMM 0
#define A2P_KERNEL_ATTN_SELF 1
#define A2P_KERNEL_ATTN_CROSS 2
#define A2P_KERNEL_LNROPE 3
#define A2P_KERNEL_CHAIN 4 /* fused row-panel chain kernels (projections + FiLM + LayerNorm + FFN), bf16 mode */
/* finer classes of the launches above (a launch is timed when its class OR its sub-class is selected) */
#define A2P_KERNEL_CHAIN_PRE 5     /* norm1 -> rotary -> [Q|K], V^T (first layer only; later layers: tail of POST) */
#define A2P_KERNEL_CHAIN_MID 6     /* out_proj -> FiLM + residual -> LayerNorm -> rotary -> Q */
#define A2P_KERNEL_CHAIN_POST 7    /* out_proj -> FiLM -> norm3 -> FFN -> FiLM [-> next layer's PRE work | final_layer] */
#define A2P_KERNEL_CHAIN_MIDPOST 8 /* body model: MID2 | keyframe attention | POST in one launch */
#define A2P_KERNEL_POSE_TAIL 9     /* body model: final_layer + 6 dilated convs + final_conv (a sub-class of A2P_KERNEL_GEMM) */
int a2p_kernel_timing(a2p_ctx* ctx, int32_t kind, int32_t enable);
int a2p_kernel_time_ms(a2p_ctx* ctx, double* total_ms, int64_t* launches);

/* ---- sample-parallel runs: which kernel family a forward takes (fused row-panel chains for large forwards, small-tile GEMMs for
 * small ones) depends on its row count, and the families differ in operand rounding.  A rank that denoises a BLOCK of a larger
 * batch names the size of the whole batch here, so that every shard takes the family the unsharded run takes and the gathered
 * samples equal the single-process samples bit for bit (sample_parallel.py does this; 0 = no hint). */
int a2p_set_batch_hint(a2p_ctx* ctx, int32_t global_batch);

/* ---- non-finite detection ------------------------------------------------------
 * The 16-bit throughput modes stage Q|K|V, the FFN hidden activation and the split-operand rows as IEEE half (liba2p_hip_f16.so:
 * |x| <= 65504) or bfloat16; a checkpoint whose activations leave that range produces inf / nan, which the reference's fp32 path
 * would not.  Every denoiser evaluation (a2p_denoise_forward, a2p_sample_step) ORs "the model output held a non-finite value"
 * into a device flag of the context at no measurable cost (the fused step tail reads every output element anyway).
 * a2p_check_finite synchronises `stream`, reads and CLEARS the flag: returns 0 when every evaluation since the last check was
 * finite, A2P_ERR_NONFINITE otherwise (a2p_last_error names the remedy: precision "bf16" for range, "fp32" for parity mode).
 * The Python loops (GaussianDiffusion.*_sample_loop) call it once per sampling call and raise A2PError; the reference has no
 * counterpart (its torch CPU / CUDA fp32 path cannot overflow on these models). */
int a2p_check_finite(a2p_ctx* ctx, void* stream);

/* ---- validity envelope of the 16-bit modes ----------------------------------------
 * The largest row maximum of the scaled attention scores q.k / sqrt(head_dim) that any self- or cross-attention query of the
 * denoiser saw since the last call (synchronises `stream`, then resets; -inf when no attention ran).  The operand rounding of the
 * 16-bit modes becomes a logit error proportional to the logits' magnitude, and softmax turns logit errors into probability
 * errors one for one: measured on the CPU model of the rounding sites and on the GPU (profiles/r04_trained_like_budget.json,
 * tests/test_hip_round4.py), IEEE-half operands hold <= 1e-3 on the loop's return value up to a maximum of ~13-15 and reach
 * 2.4e-3 at ~29; bfloat16 is 8x worse throughout.  The Python model mirror warns (A2PPrecisionWarning) above 20 in the 16-bit
 * modes; precision="fp32" is the answer there.  No reference counterpart (its path is fp32). */
int a2p_attention_logit_max(a2p_ctx* ctx, float* max_logit_host, void* stream);
/* The same read-and-reset, plus the library's own verdict: *outside = 1 when the context computes on 16-bit operands AND the maximum
 * exceeds A2P_LOGIT_ENVELOPE_16BIT -- the caller should re-create the context with precision A2P_PREC_F32 (exact at any magnitude:
 * 1e-6..5e-6 on the same scenarios) and repeat the sampling call.  The Python mirror does exactly that, once and for good per model
 * (FiLMTransformer.check_finite: "escalation", sticky; GaussianDiffusion's loops check after the first step and at the end of a call
 * and re-run a call that ended outside).  fp32 contexts always report 0. */
#define A2P_LOGIT_ENVELOPE_16BIT 20.0f
int a2p_precision_verdict(a2p_ctx* ctx, float* max_logit_host, int32_t* outside, void* stream);

/* ---- run-time switches: the A2P_* environment variables that steer a forward (INTEGRATION.md "Environment switches") are read
 * when the context is created; a host that changes one afterwards calls this (the Python mirror does, model/diffusion.py). */
int a2p_reload_env(a2p_ctx* ctx);

/* ---- debugging aid: copies an internal buffer ("film", "ktail", "vtail", "tvec", "x", "qk", "vt", "ao", "mo") to host
 * memory after a device synchronise (race hunts, scratch/stress*.py); not part of the reference's interface. */
int a2p_debug_read(a2p_ctx* ctx, const char* name, void* host, int64_t bytes);

/* ---- guide transformer + residual-VQ decode --------------------------------------
 * GuideTransformer (model/guide.py:26-83) predicts the body model's keyframe tokens from the audio features;
 * TemporalVertexCodec.decode (model/vqvae.py:508-521) turns them into the `keyframes` the pose denoiser consumes
 * (sample/generate.py:51-71 `_replace_keyframes`).  fp32 throughout.  Parameter names are the reference's state_dict keys
 * (`audio_model.*` and `*.rotary.freqs` are accepted and ignored). */
typedef struct a2p_guide_ctx a2p_guide_ctx;
typedef struct a2p_guide_config {
  int32_t tokens;           /* codebook size; id `tokens` is the sequence-start token (model/guide.py:42-45) */
  int32_t dim, num_layers, num_heads, ff_size;
  int32_t cond_feature_dim; /* audio feature width (1024) */
  int32_t emb_len;          /* rows of null_cond_embed (1998) = upper bound of the audio tokens left after the conv stack */
  int32_t num_audio_layers; /* blocks of 6 dilated convs in `pre_audio` (model/guide.py:84-109) */
  int32_t max_batch;
  int32_t max_positions;    /* longest token prefix incl. the start token (self-attention cache depth) */
  int32_t reserved[2];
} a2p_guide_config;
int a2p_guide_create(const a2p_guide_config* cfg, a2p_guide_ctx** out);
int a2p_guide_destroy(a2p_guide_ctx* ctx);
int a2p_guide_set_weight(a2p_guide_ctx* ctx, const char* name, const float* dev_ptr, int64_t numel, void* stream);
int a2p_guide_finalize(a2p_guide_ctx* ctx, void* stream);
/* Token-independent part of GuideTransformer.forward (model/guide.py:150-169), hoisted out of the 80-step loop of
 * `generate`: pre_audio conv stack, cond projection, pooled FiLM vector, norm_cond, per-layer cross-attention K/V.
 * cond_embed [batch, n_tokens, cond_feature_dim] fp32 = what encode_audio returns (:111-119); cond_drop 0|1 selects the
 * null embeddings (:156-165). */
int a2p_guide_prepare(a2p_guide_ctx* ctx, const float* cond_embed, int32_t batch, int32_t n_tokens, int32_t cond_drop,
                      void* stream);
/* GuideTransformer.forward on the prepared condition: tokens int64 [batch, len] (causal) -> logits fp32 [batch, len, tokens]. */
int a2p_guide_forward(a2p_guide_ctx* ctx, const int64_t* tokens, int32_t batch, int32_t len, float* logits, void* stream);
/* GuideTransformer.generate (model/guide.py:175-222) as one persistent launch: n_steps tokens per sequence by nucleus
 * sampling (top_p); the categorical draw of step i, sequence b is the inverse CDF at uniforms[i * batch + b] in [0, 1).
 * tokens_out int64 [batch, n_steps]; sorted_probs_out (nullable) fp32 [n_steps, batch, tokens] = the renormalised sorted
 * nucleus probabilities the reference hands to Categorical (:212-214). */
int a2p_guide_generate(a2p_guide_ctx* ctx, int32_t batch, int32_t n_steps, float top_p, const float* uniforms,
                       int64_t* tokens_out, float* sorted_probs_out, void* stream);
/* test / diagnostics: copy a prepared buffer ("pre_audio", "ct", "mem", "memr", "hidden", "film", "kc", "vc") to the host */
int a2p_guide_debug_read(a2p_guide_ctx* ctx, const char* name, void* host, int64_t bytes);
/* TemporalVertexCodec.decode: q int64 [batch, T, depth] -> out fp32 [batch, T, vertices].  codebooks: `depth` device
 * pointers [categories, latent] (quantizer.layers.i._codebook.embed); conv_w / conv_b: the 5 Conv1d of decoder.dec
 * (indices 0,2,4,6: [latent, latent, 2], dilation 1,2,3,1; index 8: [vertices, latent, 1]).  The pointer arrays are host memory. */
int a2p_vq_decode(const int64_t* q, int32_t batch, int32_t T, int32_t depth, int32_t categories, int32_t latent,
                  int32_t vertices, const float* const* codebooks, const float* const* conv_w, const float* const* conv_b,
                  float* out, void* stream);

/* ---- audio front end ------------------------------------------------------------------
 * What FiLMTransformer.forward computes from y["audio"] before anything else, in every step and pass (model/diffusion.py:
 * 354-358): encode_audio (:285-293, both stereo channels through the vq-wav2vec conv feature extractor of model/utils.py:18-26,
 * after torchaudio Resample(48000, 16000)) and, for the face model, encode_lip (:295-313: Audio2LipRegressionTransformer :37-79 =
 * Wav2VecEncoder (model/modules/audio_encoder.py:24-46) + RegressionTransformer (model/modules/transformer_modules.py:560-627)
 * + Linear, over 120-frame chunks; nearest-exact interpolation to the token count; concatenation).  Here it is one call per
 * clip.  fp32.  Parameter names are the reference's state_dict keys; the conv feature extractors take
 * `audio_model.feature_extractor.conv_layers.{i}.0.weight` / `lip_model.audio_encoder.wav2vec_model.feature_extractor.
 * conv_layers.{i}.0.weight` ([512, Cin, k], bias-free conv + ReLU, (k, stride) = (10,5) (8,4) (4,2) (4,2) (4,2) (1,1) (1,1) (1,1));
 * other `audio_model.*` / `lip_model.*` tensors of a checkpoint are accepted and ignored (a2p_frontend_set_weight returns 1). */
typedef struct a2p_frontend_ctx a2p_frontend_ctx;
typedef struct a2p_frontend_config {
  int32_t conv_dim;          /* 512 */
  int32_t resample;          /* 0: x[::3]; 1: torchaudio Resample(48000, 16000) windowed sinc (hann, width 6, rolloff 0.99) */
  int32_t lip;               /* 1: the lip regressor is present (face model) */
  int32_t d_model, num_heads, ff_size, enc_layers, dec_layers; /* RegressionTransformer: 512, 4, 1024, 2, 4 */
  int32_t lip_out;           /* n_vertices * 3 = 1014 */
  int32_t lip_pad;           /* zeros prepended at 16 kHz by Wav2VecEncoder (320) */
  int32_t chunk_frames;      /* 120 (model/diffusion.py:303) */
  int32_t samples_per_frame; /* 1600 at 48 kHz */
  int32_t max_batch, max_frames;
  int32_t conv_16bit;        /* 1: the two conv feature extractors (99 % of the front end's FLOPs) run on 16-bit MFMA operands -- IEEE half in
                              * liba2p_hip_f16.so, bfloat16 in liba2p_hip.so -- with fp32 accumulation; 0: exact-fp32 MFMA (parity mode).
                              * The resampler, the first conv layer's arithmetic and the lip regressor stay fp32 either way. */
  /* fairseq's published blocks (round 4; fairseq 0.12 models/wav2vec/wav2vec.py ConvFeatureExtractionModel / ConvAggregator -- the
   * package is absent offline: PARITY UNPINNED, restated in oracle/frontend_oracle.py).  All zero = the stub geometry of rounds 2-3
   * (bias-free Conv1d + ReLU, 8 layers, identity aggregator).  `a_*`: audio_model.feature_extractor (vq-wav2vec.pt, model/utils.py:18-26),
   * `l_*`: lip_model.audio_encoder.wav2vec_model (wav2vec_large.pt, model/modules/audio_encoder.py:24-46). */
  int32_t a_group_norm, l_group_norm;     /* 1: Conv1d -> Fp32GroupNorm(1, C, affine) -> activation; parameters conv_layers.{i}.2.{weight,bias} */
  int32_t a_activation, l_activation;     /* 0: ReLU, 1: GELU (erf) */
  int32_t a_log_compression, l_log_compression; /* 1: log(|x| + 1) behind the last conv layer */
  int32_t a_skip, l_skip;                 /* 1: skip connections between equal-width layers: (x + residual[..., ::r][..., :T]) * sqrt(residual_scale) */
  float a_residual_scale, l_residual_scale;
  int32_t l_layers;                       /* conv layers of the lip encoder's feature extractor: 0 or 8 = (10,5)(8,4)(4,2)x3(1,1)x3; 7 drops the last (1,1) */
  int32_t agg_layers;                     /* 0: identity aggregator; n <= 12: ConvAggregator layers of kernel 2, 3, ..., n + 1 (stride 1, causal padding),
                                           * parameters feature_aggregator.conv_layers.{j}.1.{weight[,bias]}, .3.{weight,bias} (GroupNorm) */
  int32_t agg_skip;                       /* 1: x = (block(x) + x) * sqrt(agg_residual_scale) */
  float agg_residual_scale;
  int32_t agg_conv_bias;                  /* 1: the aggregator's convolutions carry a bias */
  int32_t agg_zero_pad;                   /* 1: ZeroPad1d(k - 1, 0) instead of ReplicationPad1d((k - 1, 0)) */
  int32_t agg_activation;                 /* as a_activation */
  int32_t reserved[2];
} a2p_frontend_config;
int a2p_frontend_create(const a2p_frontend_config* cfg, a2p_frontend_ctx** out);
int a2p_frontend_destroy(a2p_frontend_ctx* ctx);
int a2p_frontend_set_weight(a2p_frontend_ctx* ctx, const char* name, const float* dev_ptr, int64_t numel, void* stream);
int a2p_frontend_finalize(a2p_frontend_ctx* ctx, void* stream);
/* encode_audio: audio fp32 [batch, samples, 2] (48 kHz stereo) -> out fp32 [batch, n_tokens, 2 * conv_dim]; fails if the conv
 * geometry does not give exactly n_tokens. */
int a2p_frontend_encode_audio(a2p_frontend_ctx* ctx, const float* audio, int32_t batch, int64_t samples, float* out,
                              int32_t n_tokens, void* stream);
/* encode_lip: out [batch, n_tokens, cond_dim + lip_out] = cat(cond_in [batch, n_tokens, cond_dim], interpolate(lip(audio[..., 0]))). */
int a2p_frontend_encode_lip(a2p_frontend_ctx* ctx, const float* audio, int32_t batch, int64_t samples, const float* cond_in,
                            int32_t n_tokens, int32_t cond_dim, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* A2P_HIP_H */
