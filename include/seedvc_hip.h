/* seedvc_hip.h -- C ABI of libseedvc_hip.so: the MI355X (gfx950) hot path of seed-vc inference.
 *
 * Every entry point replaces one seam of the reference's Python call surface.  The
 * caller (PyTorch-ROCm, or any C host) owns all input/output buffers; pointers are DEVICE pointers to
 * contiguous fp32 unless stated otherwise; work is enqueued on the caller's stream (hipStream_t passed
 * as void*; NULL = default stream).  The sampler, estimator and length-regulator calls do not synchronise the host in
 * steady state: HOST arrays (lengths) are copied into handle-owned pinned staging slots before the call returns, so the
 * caller may reuse them at once.  A call synchronises only when it has to grow the handle's workspace (first call, or a
 * larger batch / sequence than any before), svc_ar_generate reads tokens back every `check_every` steps, and the *_create
 * functions finish packing before they return.  The library owns only packed
 * weights and per-model workspace.  Every function returns 0 on success and non-zero on failure with a
 * message available from svc_last_error() (the Python shim re-raises it as RuntimeError, where the
 * reference raises Python exceptions: diffusion_transformer.py:121, inference.py:137,313).
 *
 * Devices and threading: a handle belongs to the device that was current when it was created, and must be used with
 * that device current (per-device state such as the zero page and kernel attributes is created lazily, once per device,
 * under a mutex).  Calls on one handle must be serialised by the caller (the reference is single-threaded Python under
 * torch.inference_mode, flow_matching.py:30); different handles may run concurrently on different streams.
 */
#ifndef SEEDVC_HIP_H
#define SEEDVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVC_ABI_VERSION 1

/* One entry of an already-loaded state_dict (fp32, device memory).  Checkpoint ingestion itself stays
 * in the reference (modules/commons.py:412-479 build_model + load_checkpoint; bigvgan.py:413-492;
 * inference.py:118-119): the shim hands the loaded module's state_dict() to *_create. */
typedef struct svc_tensor_desc {
    const char* name;
    const float* data;
    int ndim;
    int64_t shape[4];
} svc_tensor_desc_t;

/* ---------------------------------------------------------------- DiT estimator + CFM sampler */
/* Hyper-parameters the reference reads from configs/presets/NAME.yml (model_params.DiT / .wavenet /
 * .style_encoder) or configs/v2/vc_wrapper.yaml (cfm.estimator). */
typedef struct svc_dit_config {
    int version;            /* 1: modules/diffusion_transformer.py DiT ; 2: modules/v2/dit_wrapper.py DiT */
    int hidden_dim, num_heads, depth, in_channels, content_dim, style_dim;
    int final_layer_type;   /* 0 = mlp, 1 = wavenet */
    int time_as_token, style_as_token, uvit_skip_connection, long_skip_connection, style_condition;
    int wn_hidden_dim, wn_num_layers, wn_kernel_size, wn_dilation_rate;
} svc_dit_config_t;

typedef struct svc_dit svc_dit_t;

/* Packs a DiT from `CFM.estimator.state_dict()` (keys listed in SURVEY.md 8b).  Replaces nothing in the
 * reference's loading code; it is what `model.cfm.estimator.setup_caches(...)` (inference.py:90) plus
 * the first forward would have prepared (RoPE table, skip lists). */
int svc_dit_create(const svc_dit_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights,
                   void* stream, svc_dit_t** out);
void svc_dit_destroy(svc_dit_t* m);

/* Utterances processed together inside one sampler call (micro-batch whose activations stay cache
 * resident); 0 restores the default. */
int svc_dit_set_microbatch(svc_dit_t* m, int utterances);

/* Transformer layers run on the fused row-panel kernel (one launch per layer besides attention, csrc/fused.hip) when a
 * launch covers at least `rows` token rows (streams x utterances x padded frames) and the width is supported (hidden_dim
 * 384 or 512); smaller launches use the tap-GEMM kernels.  rows < 0 restores the default (10240), 0 forces the fused
 * kernel, a huge value disables it.  The two paths agree to fp16-operand rounding (not bit for bit). */
int svc_dit_set_fused_min_rows(svc_dit_t* m, long rows);
int svc_dit_fused_available(svc_dit_t* m);
/* Optional (off by default: the sampler is GPU-bound on MI355X, replay measured 3 % slower than eager launches at B = 1):
 * the Euler loop of svc_cfm_sample (all steps of a micro-batch group: estimator + state update) is captured into a hipGraph
 * the second time a (batch, length, steps, guidance) combination is seen and replayed afterwards (the reference's
 * counterpart: `compile_cfm`, modules/v2/vc_wrapper.py:116-123).  Replays are bit-identical to eager runs.  on = 0 turns
 * capture off and drops the cached graphs; the environment variable SVC_DIT_GRAPH=0|1 overrides. */
int svc_dit_set_graphs(svc_dit_t* m, int on);

typedef struct svc_cfm_args {
    int B;                      /* utterances; each is an independent B=1 run of the reference sampler */
    int T;                      /* max frames (prompt + source) */
    int P;                      /* frames in `prompt` */
    const float* mu;            /* [B][T][content_dim]   (cat_condition) */
    const float* prompt;        /* [B][in_channels][P]   (mel2) */
    const float* style;         /* [B][style_dim]        (style2) */
    const float* z;             /* [B][in_channels][T]   noise: torch.randn in flow_matching.py:50 */
    const int64_t* x_lens;      /* HOST [B] valid frames per utterance (<= T); NULL = all T */
    const int64_t* prompt_lens; /* HOST [B] prompt frames per utterance (<= P); NULL = all P */
    int n_timesteps;
    float temperature;
    float cfg_rate[2];          /* v1: cfg_rate[0] = inference_cfg_rate ; v2: [intelligibility, similarity] */
    int random_voice;           /* v2 only (modules/v2/cfm.py:77-87) */
    float* out;                 /* [B][in_channels][T] */
} svc_cfm_args_t;

/* Replaces `model.cfm.inference(mu, x_lens, prompt, style, f0, n_timesteps, temperature,
 * inference_cfg_rate)` = BASECFM.inference + solve_euler (modules/flow_matching.py:30-112) and the v2
 * CFM.inference + solve_euler (modules/v2/cfm.py:16-132; cosine-warped t_span, 1/2/3-way CFG). */
int svc_cfm_sample(svc_dit_t* m, const svc_cfm_args_t* args, void* stream);

/* Replaces one estimator evaluation `estimator(x, prompt_x, x_lens, t, style, mu)` =
 * DiT.forward (modules/diffusion_transformer.py:486-537, modules/v2/dit_wrapper.py:114-152).
 * x, prompt_x, out: [N][in_channels][T]; style [N][style_dim]; mu [N][T][content_dim]; t scalar
 * (the sampler always passes one t for the whole batch). x_lens HOST [N] or NULL. */
int svc_dit_forward(svc_dit_t* m, int N, int T, const float* x, const float* prompt_x, const int64_t* x_lens,
                    float t, const float* style, const float* mu, float* out, void* stream);

/* ---------------------------------------------------------------- vocoders */
typedef struct svc_bigvgan_config {   /* modules/bigvgan/config.json */
    int num_mels, upsample_initial_channel, num_upsamples, num_kernels;
    int upsample_rates[8], upsample_kernel_sizes[8];
    int resblock_kernel_sizes[4], resblock_dilation_sizes[4][3];
    int use_tanh_at_final, use_bias_at_final, snake_logscale, snakebeta;
    int precision;                    /* 0 = fp32 MFMA (exact); 1 = plain fp16 operands (RMS 2.4e-4: outside the 1e-4 bound,
                                         reported only); 2 = fp16x3: hi/lo fp16 split of both operands, three MFMA products,
                                         fp32 accumulate (RMS <= 2e-6); 3 = fp16p8: as 2, but on the long stride-1 convs the
                                         two correction products run as ONE block-scaled fp8 MFMA (RMS 6e-6 at S = 430
                                         against the reference; bound 1e-4) -- the value the Python mirrors pass by default */
} svc_bigvgan_config_t;
typedef struct svc_bigvgan svc_bigvgan_t;
int svc_bigvgan_create(const svc_bigvgan_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights,
                       void* stream, svc_bigvgan_t** out);
void svc_bigvgan_destroy(svc_bigvgan_t* m);
/* Replaces `vocoder_fn(mel)` = BigVGAN.forward (modules/bigvgan/bigvgan.py:360-386).
 * mel [B][num_mels][S] -> out [B][1][S * prod(upsample_rates)]. */
int svc_bigvgan_forward(svc_bigvgan_t* m, const float* mel, int B, int S, float* out, void* stream);

typedef struct svc_hift_config {      /* configs/hifigan.yml */
    int in_channels, base_channels, nb_harmonics, sampling_rate;
    float nsf_alpha, nsf_sigma, nsf_voiced_threshold;
    int num_upsamples, upsample_rates[4], upsample_kernel_sizes[4];
    int istft_n_fft, istft_hop;
    int num_kernels, resblock_kernel_sizes[4], resblock_dilation_sizes[4][3];
    int source_resblock_kernel_sizes[4], source_resblock_dilation_sizes[4][3];
    float lrelu_slope, audio_limit;
    int f0_cond_channels;
    int precision;                    /* 0 fp32 / 1 fp16 / 2 fp16x3 / 3 fp16p8, as svc_bigvgan_config.precision */
} svc_hift_config_t;
typedef struct svc_hift svc_hift_t;
int svc_hift_create(const svc_hift_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights,
                    void* stream, svc_hift_t** out);
void svc_hift_destroy(svc_hift_t* m);
/* Replaces `vocoder_fn(mel)` = HiFTGenerator.forward (modules/hifigan/generator.py:400-436).
 * mel [B][80][S]; f0 [B][S] or NULL (NULL -> the model's f0_predictor, f0_predictor.py:51-55);
 * phase0 [B][nb_harmonics+1][1] = the U(-pi,pi) draw of SineGen (generator.py:208-210);
 * noise [B][nb_harmonics+1][S*up] = its randn_like draw (generator.py:222);  out [B][S*up].
 * f0_out (optional, [B][S]) receives the f0 actually used. */
int svc_hift_forward(svc_hift_t* m, const float* mel, const float* f0, const float* phase0, const float* noise,
                     int B, int S, float* out, float* f0_out, void* stream);
/* Utterances per internal pass of the vocoders (0 = default 32); results do not depend on it. */
int svc_bigvgan_set_microbatch(svc_bigvgan_t* m, int utterances);
int svc_hift_set_microbatch(svc_hift_t* m, int utterances);

/* ---------------------------------------------------------------- v2 AR model (decode step) */
typedef struct svc_ar_config {        /* configs/v2/vc_wrapper.yaml:39-53 (modules.v2.ar.NaiveModelArgs) */
    int dim, n_head, n_local_heads, head_dim, n_layer, intermediate_size, vocab_size, max_seq_len;
    float rope_base, norm_eps;
} svc_ar_config_t;
typedef struct svc_ar svc_ar_t;
/* Packs `NaiveWrapper.state_dict()` (keys model.layers.N.*, model.norm, model.output); allocates the fp32 KV cache
 * that `setup_caches(1, max_seq_len)` would (modules/v2/ar.py:160-179). */
int svc_ar_create(const svc_ar_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream, svc_ar_t** out);
void svc_ar_destroy(svc_ar_t* m);
int svc_ar_reset(svc_ar_t* m, void* stream);      /* zero the KV cache */
/* Replaces `model.forward_generate(x, input_pos, kv_pos)` (modules/v2/ar.py:239-267) for B = 1: x [S][dim] fp32,
 * input_pos / kv_pos HOST [S]; logits_out [vocab] = logits of the last token. */
int svc_ar_forward_generate(svc_ar_t* m, const float* x, int S, const int64_t* input_pos, const int64_t* kv_pos,
                            float* logits_out, void* stream);
/* The same one-token step replayed from a captured hipGraph (replaces `compiled_decode_fn`, the
 * torch.compile(mode="reduce-overhead") step of modules/v2/vc_wrapper.py:105-114).  set_pos != 0 (re)sets the device
 * positions; every call afterwards advances input_pos and kv_pos by one (ar.py:402-403). */
int svc_ar_decode_step(svc_ar_t* m, const float* x, int set_pos, int64_t input_pos, int64_t kv_pos, float* logits_out,
                       void* stream);
/* Replaces the token loop of `NaiveWrapper.generate(prompt_text, prompt_target, ...)` (modules/v2/ar.py:382-422), B = 1
 * (SURVEY.md 8f row 4).  x_prefill [S][dim] = [sep, prompt_text, sep, embed(prompt_target)] as the reference builds it
 * (ar.py:390-396), input_pos / kv_pos HOST [S]; exp_noise device [max_new][vocab] = the Exp(1) draws of
 * multinomial_sample_one_no_sync, row t for token t; EOS (vocab-1) is suppressed while fewer than
 * min_tokens_before_eos (reference: 10) tokens exist; the host inspects the tokens every `check_every` steps only.
 * tokens_out device [max_new]; *n_tokens = tokens generated before EOS (<= max_new; reference cap 4001).
 * Needs `model.embeddings.weight` in the state dict given to svc_ar_create. */
int svc_ar_generate(svc_ar_t* m, const float* x_prefill, int S, const int64_t* input_pos, const int64_t* kv_pos,
                    const float* exp_noise, int max_new, int min_tokens_before_eos, float temperature, float top_p,
                    float repetition_penalty, int check_every, int32_t* tokens_out, int32_t* n_tokens, void* stream);
/* Replaces `sample(logits, previous_tokens, suppress_tokens, temperature, top_p, repetition_penalty)`
 * (modules/v2/ar.py:712-763): repetition penalty, top-p, temperature softmax, argmax(probs / q) with q = exp_noise
 * (the Exp(1) draw of multinomial_sample_one_no_sync, supplied by the caller).  suppress_token < 0 = none. */
int svc_ar_sample(svc_ar_t* m, const float* logits, const int32_t* prev_tokens, int n_prev, int suppress_token,
                  float temperature, float top_p, float repetition_penalty, const float* exp_noise, int32_t* idx_out,
                  float* probs_out, void* stream);

/* ---------------------------------------------------------------- length regulator */
typedef struct svc_lr_config {        /* modules/length_regulator.py:29-88, modules/v2/length_regulator.py:28-72 */
    int channels, in_channels, out_channels;
    int is_discrete, codebook_size;   /* discrete: token ids through `embedding`; else `content_in_proj` */
    int n_convs;                      /* len(sampling_ratios): Conv1d(k3) + GroupNorm(1) + Mish blocks */
    int interpolate;                  /* len(sampling_ratios) > 0 */
    int has_final_conv;               /* v1: always; v2: only when out_channels != channels (else nn.Identity) */
    int f0_condition, n_f0_bins;
} svc_lr_config_t;
typedef struct svc_lr svc_lr_t;
/* Packs `InterpolateRegulator.state_dict()` (keys model.N.*, embedding.weight, content_in_proj.*, f0_embedding.weight,
 * f0_mask). */
int svc_lr_create(const svc_lr_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream, svc_lr_t** out);
void svc_lr_destroy(svc_lr_t* m);
/* Replaces `length_regulator(x, ylens=..., n_quantizers=..., f0=...)[0]` (modules/length_regulator.py:90-141; v2
 * modules/v2/length_regulator.py:74-105) for a batch of independent utterances.
 * x [B][tin_max][in_channels] fp32 (continuous) or tokens [B][tin_max] int64 (discrete); in_lens / ylens / f0_lens are
 * HOST arrays [B]; f0 [B][tf0_max] Hz or NULL (NULL with f0_condition -> `f0_mask`); out [B][tout_max][out_channels],
 * rows >= ylens[b] zero (the reference's `out * mask`). */
int svc_lr_forward(svc_lr_t* m, const float* x, const int64_t* tokens, const int32_t* in_lens, int B, int tin_max,
                   const int32_t* ylens, int tout_max, const float* f0, const int32_t* f0_lens, int tf0_max, float* out,
                   void* stream);

/* Replaces the reference's only native seam: anti_alias_activation_cuda.forward(inputs, up_ftr,
 * down_ftr, alpha, beta) (modules/bigvgan/alias_free_activation/cuda/anti_alias_activation.cpp:19-23,
 * anti_alias_activation_cuda.cu:43-246).  x, y: [B][C][L]; dtype 0 = fp32, 1 = fp16, 2 = bf16;
 * up12 / down12: 12 filter taps (fp32); log_alpha / log_beta: [C] log-scale SnakeBeta parameters
 * (exp is applied in the kernel, as in the reference kernel). Accumulates in fp32 for every dtype. */
int svc_anti_alias_act_fwd(const void* x, void* y, const float* up12, const float* down12,
                           const float* log_alpha, const float* log_beta, int B, int C, int L, int dtype,
                           void* stream);

/* ---------------------------------------------------------------- log-mel front-end (first half) */
typedef struct svc_mel svc_mel_t;
/* Replaces `mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False)`
 * (modules/audio.py:45-82).  window [win] = torch.hann_window(win_size), mel_basis [n_mels][n_fft/2+1] =
 * librosa.filters.mel(...) as the reference caches them (device fp32); win must equal n_fft (all presets). */
int svc_mel_create(int n_fft, int hop, int win, int n_mels, const float* window, const float* mel_basis, void* stream,
                   svc_mel_t** out);
void svc_mel_destroy(svc_mel_t* m);
int svc_mel_frames(const svc_mel_t* m, int L);          /* frames for L samples: 1 + (L - hop) / hop */
/* y [B][L] fp32 in [-1, 1] -> out [B][n_mels][frames] = log(clamp(mel @ sqrt(|STFT|^2 + 1e-9), 1e-5)). */
int svc_mel_forward(svc_mel_t* m, const float* y, int B, int L, float* out, void* stream);

/* ---------------------------------------------------------------- CAMPPlus style encoder + Kaldi fbank (second half) */
typedef struct svc_campplus_config {  /* modules/campplus/DTDNN.py:54-62 (CAMPPlus.__init__ defaults; the drivers use embedding_size 192) */
    int feat_dim, embedding_size, growth_rate, bn_size, init_channels, m_channels;
    int n_blocks, block_layers[4], block_kernel[4], block_dilation[4];
    int seg_len;                      /* CAMLayer.seg_pooling segment length (layers.py:119) */
} svc_campplus_config_t;
typedef struct svc_campplus svc_campplus_t;
/* Packs `CAMPPlus.state_dict()` (eval mode: BatchNorm running statistics are folded). */
int svc_campplus_create(const svc_campplus_config_t* cfg, const svc_tensor_desc_t* weights, int n_weights, void* stream,
                        svc_campplus_t** out);
void svc_campplus_destroy(svc_campplus_t* m);
/* Replaces `campplus_model(feat)` = CAMPPlus.forward(x) (modules/campplus/DTDNN.py:132-137; inference.py:430,
 * seed_vc_wrapper.py): feat [B][T][feat_dim] (mean-normalised fbank) -> out [B][embedding_size].  All clips of a batch
 * have T frames (the drivers call it with B = 1 and no x_lens). */
int svc_campplus_forward(svc_campplus_t* m, const float* feat, int B, int T, float* out, void* stream);
/* Replaces `torchaudio.compliance.kaldi.fbank(wave, num_mel_bins=feat_dim, dither=0, sample_frequency=16000)`
 * (inference.py:418-428): wave [n_samples] fp32 -> out [frames][feat_dim], frames = svc_kaldi_fbank_frames(n_samples). */
int svc_kaldi_fbank_frames(int n_samples);
int svc_kaldi_fbank(svc_campplus_t* m, const float* wave, int n_samples, float* out, void* stream);

/* Device-side counterpart of `crossfade(chunk1, chunk2, overlap)` (inference.py:343-350): the first n samples of
 * chunk2 become chunk2 * fade_in + chunk1_tail * fade_out in float64, stored as float32 (bit-identical to the numpy
 * arithmetic).  fade_in / fade_out: the caller's cos^2 windows (double, device). */
int svc_crossfade(float* chunk2, const float* chunk1_tail, const double* fade_in, const double* fade_out, int n,
                  void* stream);

/* ---------------------------------------------------------------- op-level entry points (parity tests) */
/* C[M][N] (fp32) = A[M][K] * W[N][K]^T + bias ; dtype 0: operands rounded to fp16, 1: fp32 MFMA. */
int svc_op_linear(const float* a, const float* w, const float* bias, float* c, int M, int N, int K, int dtype,
                  int act, void* stream);
/* Channels-last Conv1d through the tap-GEMM: x [B][L][Cin], w [Cout][Cin][k] (torch layout),
 * y [B][Lout][Cout]; pad_mode 0 zero / 1 reflect / 2 replicate; explicit left pad. */
int svc_op_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cin, int Cout,
                  int k, int dilation, int stride, int pad_left, int Lout, int pad_mode, int dtype, void* stream);
/* ConvTranspose1d (k = 2*stride, padding = stride/2): x [B][L][Cin], w [Cin][Cout][k], y [B][L*stride][Cout]. */
int svc_op_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int L, int Cin,
                            int Cout, int k, int stride, int dtype, void* stream);
/* softmax(q k^T / 8 + key-padding mask) v ; q,k,v,out [N][T][H][64] fp32 (computed in fp16 MFMA). */
int svc_op_attention(const float* q, const float* k, const float* v, float* out, int N, int T, int H,
                     const int64_t* kv_lens_host, void* stream);
/* y = rmsnorm(x) * gamma * (add_one + w) + b ; x [rows][D]. */
int svc_op_rmsnorm(const float* x, const float* gamma, const float* w, const float* b, int add_one, float* y,
                   int rows, int D, void* stream);

/* Tuning / measurement aid (tools/gemm_bench.py, tools/gemm_small_bench.py; no model uses it): times `iters` launches of the
 * tap-GEMM on synthetic operands of shape M x N x K; dtype 0 fp16 / 1 fp32, epi = epilogue kind, debug = tile-form override
 * bits; *out_ms = average launch time. */
int svc_op_gemm_bench(int M, int N, int K, int dtype, int epi, int iters, int debug, float* out_ms, void* stream);
/* Test aid: one fp16 tap-GEMM (epi: 0 store + bias + residual, 1 SwiGLU, 2 tanh-sigmoid, 3 QKV + RoPE + V^T) computed under
 * two tile-form overrides on the same pseudo-random operands; *n_diff = 32-bit output words that differ (every tile form keeps
 * the k order of each output element, so 0 is the contract). */
int svc_op_gemm_forms_diff(int M, int N, int K, int epi, int debug_a, int debug_b, long long* n_diff, void* stream);

/* Optional launch timing: HIP events around every tap-GEMM / attention launch on its stream.
 * svc_prof_collect fills out[cls*4 + {0..3}] = {launches, total ms, algorithmic flops, algorithmic bytes}
 * for the n_cls <= 4 classes 0 = fp16 tap-GEMM + resident-tile conv, 1 = fp32 tap-GEMM, 2 = attention, 3 = fused DiT
 * row-panel kernel, and clears the records. */
int svc_prof_enable(int on);
int svc_prof_collect(double* out, int n_cls);

const char* svc_last_error(void);
int svc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
