/* t2p.h -- C ABI of libt2p_hip.so: the text2protein reverse-diffusion sampling path on MI355X.
 *
 * The reference (szhan227/text2protein) has no FFI layer; its seam for this path is a set of
 * Python callables.  Each entry point below names the reference
 * interface it stands in for.  All pointers marked "device" are HIP device pointers owned by
 * the caller; nothing is retained beyond the call unless stated.  `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  Every function returns 0 on success or a
 * non-zero status; t2p_last_error() then returns a message for the calling thread.
 * There is no CPU fallback: every compute entry point launches HIP kernels.
 */
#ifndef T2P_H_
#define T2P_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2P_DTYPE_F32 0   /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32)      */
#define T2P_DTYPE_BF16 1  /* bf16 operands, fp32 accumulate                */
#define T2P_DTYPE_F16 2   /* fp16 operands, fp32 accumulate                */

#define T2P_SDE_VE 0
#define T2P_SDE_VP 1

/* Flat view of the YAML keys the score network reads
 * (reference score_sde_pytorch/models/ncsnpp.py:74-217, configs/test_config.yml). */
typedef struct t2p_model_config {
  int32_t num_channels;        /* data.num_channels                        */
  int32_t max_res_num;         /* data.max_res_num (map side L)            */
  int32_t nf;                  /* model.nf                                 */
  int32_t num_res_blocks;      /* model.num_res_blocks                     */
  int32_t n_ch_mult;           /* len(model.ch_mult), <= 8                 */
  int32_t ch_mult[8];
  int32_t n_attn_resolutions;  /* len(model.attn_resolutions), <= 8        */
  int32_t attn_resolutions[8];
  int32_t n_heads;             /* model.n_heads                            */
  int32_t context_dim;         /* model.context_dim                        */
  int32_t num_scales;          /* model.num_scales (N)                     */
  double sigma_min, sigma_max; /* model.sigma_{min,max}                    */
  int32_t skip_rescale;        /* model.skip_rescale                       */
  int32_t scale_by_sigma;      /* model.scale_by_sigma                     */
  int32_t compute_dtype;       /* T2P_DTYPE_*                              */
} t2p_model_config;

typedef struct t2p_engine t2p_engine;    /* the score network: UNetModel (ncsnpp.py:71-263) */
typedef struct t2p_sampler t2p_sampler;  /* the PC loop: get_pc_sampler (sampling.py:213-291) */

const char* t2p_last_error(void);
/* number of visible HIP devices, or -1 (no driver / no GPU).  Does not create a context. */
int t2p_device_count(void);

/* ---- score network ------------------------------------------------------------------------
 * t2p_engine_create      <- UNetModel.__init__ / get_model   (ncsnpp.py:74-217, score_sde_pytorch/utils.py:4-9)
 * t2p_engine_load_param  <- load_state_dict of one tensor    (score_sde_pytorch/utils.py:14);
 *                           `name` is the reference state-dict key, with or without "module."
 * t2p_engine_finalize    <- end of restore_checkpoint: fails if any tensor is missing
 * t2p_engine_set_context <- the `context=` argument          (sampling_6d.py:137,152): projects
 *                           the frozen text embedding through every to_k / to_v once
 * t2p_engine_score       <- model(x, labels, context)        (ncsnpp.py:220-263)                  */
int t2p_engine_create(const t2p_model_config* cfg, t2p_engine** out);
void t2p_engine_destroy(t2p_engine* e);
int t2p_engine_num_params(const t2p_engine* e);
/* name and shape (ndim <= 4) of expected tensor i, in reference parameters() order */
int t2p_engine_param_info(const t2p_engine* e, int i, const char** name, int64_t shape[4], int* ndim);
int t2p_engine_load_param(t2p_engine* e, const char* name, const float* host_data, const int64_t* shape, int ndim);
int t2p_engine_finalize(t2p_engine* e);
/* context: device fp32 [batch][tokens][context_dim] */
int t2p_engine_set_context(t2p_engine* e, const float* context, int batch, int tokens, void* stream);
/* x: device fp32 (batch, C, L, L); labels: device int32 [batch] time labels (index into the
 * descending sigma table); out: device fp32 (batch, C, L, L) = network output / sigma[label].
 * Range precondition of the 16-bit engines: with sigma_max <= 4096 the input convolution runs on f16 operand pairs and needs
 * |x| <= 65504 (a VE state stays within a few sigma_max); larger magnitudes saturate to that bound -- finite but wrong -- rather
 * than producing inf / NaN.  The f32 engine has no such bound. */
int t2p_engine_score(t2p_engine* e, const float* x, const int32_t* labels, float* out, int batch, void* stream);
/* as t2p_engine_score, with optional fractional time values for the sinusoidal embedding: the VP
 * branch of get_score_fn passes labels = t * (N - 1) as floats (models/utils.py:150-152); the
 * model embeds the float and indexes its sigma table with the truncated value (ncsnpp.py:223-224).
 * labels_f: device fp32 [batch] or NULL (= embed the integer labels). */
int t2p_engine_score_ex(t2p_engine* e, const float* x, const int32_t* labels, const float* labels_f, float* out,
                        int batch, void* stream);
/* bytes of device memory currently held (weights + cached activations) */
int64_t t2p_engine_device_bytes(const t2p_engine* e);
/* activation buffers the LAST score evaluation / set_context left checked out and the engine took back at its end: 0 after every
 * complete evaluation (a leak check for tests); > 0 only after a call that failed half-way */
int t2p_engine_pool_reclaimed(const t2p_engine* e);

/* ---- predictor-corrector sampler ------------------------------------------------------------
 * t2p_sampler_create        <- get_sampling_fn / get_pc_sampler (sampling.py:78-104, 213-243)
 * t2p_sampler_set_condition <- what sampling.py:259-277 reduces `condition` to:
 *                              conditional_mask (1 = evolves) and x_initial
 * t2p_sampler_step          <- one iteration of the loop body  (sampling.py:279-285)
 * t2p_sampler_run           <- pc_sampler(model, condition, context) (sampling.py:245-289)         */
typedef struct t2p_sampler_config {
  int32_t sde;               /* T2P_SDE_VE (every shipped config) or T2P_SDE_VP                  */
  int32_t N;                 /* sde.N = model.num_scales                                          */
  double sigma_min, sigma_max, beta_min, beta_max;
  double snr;                /* sampling.snr                                                      */
  int32_t n_steps_each;      /* sampling.n_steps_each                                             */
  int32_t probability_flow;  /* sampling.probability_flow                                         */
  int32_t denoise;           /* sampling.noise_removal                                            */
  double eps;                /* integrate to eps (1e-5 for VE, sampling_6d.py:79)                 */
  int32_t batch;             /* chains held by this process                                       */
  int32_t global_batch;      /* chains the Langevin batch-mean runs over (>= batch; > batch needs
                                t2p_sampler_set_norm_allreduce)                                   */
  uint64_t seed;             /* on-device Philox noise                                            */
} t2p_sampler_config;

/* g_table: optional host float[N], G_i of step i as the reference computes it in float32
 * (sde_lib.py:237-245); NULL = computed here in double.
 * label_table: optional host int32[N], the time label the score network receives at loop step i,
 * round((T - linspace(T, eps, N)[i]) (N - 1)) in the reference's float32 arithmetic (sampling.py:257,
 * models/utils.py:159-171); NULL = computed here in double from cfg->eps.  (It equals i only for tiny eps.) */
int t2p_sampler_create(t2p_engine* e, const t2p_sampler_config* cfg, const float* g_table, const int32_t* label_table,
                       t2p_sampler** out);
void t2p_sampler_destroy(t2p_sampler* s);
/* Seed of the on-device Philox noise of the following steps.  The reference draws fresh torch.randn noise on
 * every call of pc_sampler; the Python mirror derives one seed per (seed, call index) and sets it here. */
int t2p_sampler_set_seed(t2p_sampler* s, uint64_t seed);
/* VP SDE (VPSDE, sde_lib.py:106-157; score function models/utils.py:138-157; Langevin alpha sampling.py:184-186) in the fused
 * loop.  A sampler created with cfg.sde = T2P_SDE_VP takes g_table[i] = sqrt(beta_k) and label_table[i] = floor(t_i (N - 1)) at
 * t2p_sampler_create and, here, four more host tables of N floats for loop step i (t_i = linspace(T, eps, N)[i], k = the
 * reference's truncated timestep): label_f = t_i (N - 1) (the fractional label the network embeds), score_scale = -1 /
 * sqrt_1m_alphas_cumprod[label] (score = -model / std), x_coef = 2 - sqrt(alpha_k) (x - f with f = (sqrt(alpha) - 1) x),
 * corr_alpha = alpha_k. */
int t2p_sampler_set_vp_tables(t2p_sampler* s, const float* label_f, const float* score_scale, const float* x_coef, const float* corr_alpha);
/* Global-batch Langevin step size (the reference's DataParallel run takes the norm means over the whole batch,
 * sampling.py:193-195; cfg.global_batch > cfg.batch): every corrector step writes {sum_b ||grad_b||,
 * sum_b ||noise_b||} of this process's chains to `device_sums2` (caller-owned device float[2]) and calls
 * fn(device_sums2, stream, user), which must sum the two floats over all processes in stream order (one RCCL
 * all_reduce) and return 0.  NULL, NULL, NULL removes the hook. */
typedef int (*t2p_allreduce_fn)(float* device_sums2, void* stream, void* user);
int t2p_sampler_set_norm_allreduce(t2p_sampler* s, float* device_sums2, t2p_allreduce_fn fn, void* user);
/* mask: device uint8 (B,C,L,L), 1 where the chain evolves; x_initial: device fp32; both NULL = unconditional */
int t2p_sampler_set_condition(t2p_sampler* s, const uint8_t* mask, const float* x_initial);
/* reset the step counter to `step` (0 at the start of a run); a step at index >= N is refused */
int t2p_sampler_reset(t2p_sampler* s, int step, void* stream);
/* One PC step at the current step index, in place on x (device fp32 (B,C,L,L)); x_mean receives the
 * predictor's mean.  noise_corrector / noise_predictor: device standard-normal draws to use
 * (parity runs); NULL = generate on device. */
int t2p_sampler_step(t2p_sampler* s, float* x, float* x_mean, const float* noise_corrector,
                     const float* noise_predictor, void* stream);
/* The same step replayed from a captured hipGraph (on-device noise only).  The first call runs
 * eagerly (it sizes the activation pool), the second captures, later calls replay; x / x_mean /
 * the condition pointers must stay the same between calls (a change triggers a re-capture).
 * `stream` must be a real stream (capture on the NULL stream is not allowed). */
int t2p_sampler_step_graph(t2p_sampler* s, float* x, float* x_mean, void* stream);
/* Full run: x must hold the (already conditioned) prior sample on entry when `prior_given`, else it
 * is drawn on device (randn * sigma_max, then mask applied).  out receives x_mean (denoise) or x. */
int t2p_sampler_run(t2p_sampler* s, float* x, float* out, int prior_given, int n_steps, void* stream);
/* measurement: the number of device dispatches (graph nodes) one PC step (sampling.py:279-285) enqueues; the step is captured
 * on `stream` (not the default stream) and discarded, nothing executes; call after at least one eager step */
int t2p_sampler_count_dispatches(t2p_sampler* s, float* x, float* x_mean, void* stream, int* n_out);

/* ---- training step (first slice: fp32 arithmetic only, VE SDE) ------------------
 * t2p_train_create      <- get_model + get_optimizer + ExponentialMovingAverage(model.parameters(), decay)
 *                          (score_sde_pytorch/utils.py:4-9, losses.py:26-36, models/ema.py:13-30; train.py builds `state` from them)
 * t2p_train_load_param  <- load_state_dict of one tensor; the EMA shadow starts as a copy (ema.py:28-29)
 * t2p_train_loss        <- loss_fn(model, batch, condition)      (losses.py:105-134), optionally with loss.backward()
 * t2p_train_step        <- step_fn(state, batch, condition), train=True (losses.py:165-176): zero_grad, loss, backward,
 *                          optimize_fn (warm-up on state['step'], clip_grad_norm_, Adam: losses.py:41-49), step += 1, ema.update
 * t2p_train_eval_loss   <- step_fn with train=False (losses.py:177-183): the loss under the EMA weights, model in eval mode
 * The model runs in train mode (models/utils.py:116-118): Dropout_0 of every residual block is active when dropout > 0.       */
typedef struct t2p_train_config {
  double lr, beta1, eps, weight_decay;   /* optim.lr / beta1 / eps / weight_decay; beta2 = 0.999 as get_optimizer fixes it    */
  double warmup;                          /* optim.warmup: lr * min(step / warmup, 1) when > 0                                   */
  double grad_clip;                       /* optim.grad_clip: clip_grad_norm_(max_norm) when >= 0                                */
  double ema_rate;                        /* model.ema_rate                                                                      */
  double dropout;                         /* model.dropout (Dropout_0 of ResnetBlockBigGANpp, layers.py:293,318)                 */
  double t_eps;                           /* smallest time drawn, 1e-5 (get_sde_loss_fn's eps)                                   */
  int32_t cond_flags;                     /* model.condition: 1 length | 2 ss | 4 inpainting (losses.py:113-123)                */
  uint64_t seed;                          /* on-device draws of t, z and the dropout masks when the batch does not supply them   */
} t2p_train_config;

typedef struct t2p_train_batch {
  const float* coords_6d;      /* device fp32 (batch, C, L, L): batch["coords_6d"]                                               */
  const uint8_t* mask_pair;    /* device uint8 (batch, L, L):   batch["mask_pair"]                                               */
  const uint8_t* mask_inpaint; /* device uint8 (batch, L, L):   batch["mask_inpaint"], needed with the inpainting condition      */
  const float* context;        /* device fp32 (batch, tokens, context_dim): llm.model.embed_tokens(caption tokens)              */
  int32_t batch, tokens;
  const float* t;              /* device fp32 [batch] or NULL = t ~ U(t_eps, 1) drawn on the device (losses.py:106)              */
  const float* z;              /* device fp32 (batch, C, L, L) or NULL = z ~ N(0, 1) drawn on the device (losses.py:107)         */
} t2p_train_batch;

typedef struct t2p_trainer t2p_trainer;
int t2p_train_create(const t2p_model_config* model, const t2p_train_config* train, t2p_trainer** out);
void t2p_train_destroy(t2p_trainer* t);
int t2p_train_num_params(const t2p_trainer* t);
/* name and shape of tensor i in the reference's parameters() order (the order of the EMA shadow list and of the optimizer state) */
int t2p_train_param_info(const t2p_trainer* t, int i, const char** name, int64_t shape[4], int* ndim);
int t2p_train_load_param(t2p_trainer* t, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* which: 0 parameter, 1 gradient of the last backward pass (after t2p_train_step: as clip_grad_norm_ left it), 2 EMA shadow,
 * 3 Adam exp_avg, 4 Adam exp_avg_sq; host buffers of the tensor's size, reference layout */
int t2p_train_read(t2p_trainer* t, int which, const char* name, float* host_out);
int t2p_train_write(t2p_trainer* t, int which, const char* name, const float* host_in);
/* state['step'] (drives the warm-up), the optimizer's own update count (Adam bias correction) and ema.num_updates */
int t2p_train_set_step(t2p_trainer* t, int64_t step, int64_t adam_updates, int64_t ema_updates);
int t2p_train_get_step(const t2p_trainer* t, int64_t out3[3]);
/* parity runs: keep-masks of Dropout_0 for the next pass, one device uint8 [batch][H][W][C] (NHWC, the block's own resolution and
 * width) per residual block in forward order; n = 0 returns to on-device Philox masks */
int t2p_train_set_dropout_masks(t2p_trainer* t, const uint8_t* const* device_masks, int n);
/* loss_host: host float; score_out (optional): device fp32 (batch, C, L, L), the score the loss was computed from */
int t2p_train_loss(t2p_trainer* t, const t2p_train_batch* batch, int backward, float* loss_host, float* score_out, void* stream);
int t2p_train_step(t2p_trainer* t, const t2p_train_batch* batch, float* loss_host, void* stream);
/* Data-parallel training (the reference wraps the model in DataParallel, score_sde_pytorch/utils.py:8: one gradient over the whole
 * batch): every process runs t2p_train_loss(backward = 1) on its shard, the flat gradient buffer (device fp32 [n], parameters() order,
 * owned by the trainer) is averaged over the processes -- ONE RCCL all-reduce -- and t2p_train_apply runs optimize_fn + step += 1 +
 * ema.update on it.  t2p_train_step = t2p_train_loss(backward = 1) + t2p_train_apply. */
int t2p_train_grad_buffer(t2p_trainer* t, float** device_ptr, int64_t* n);
int t2p_train_apply(t2p_trainer* t, void* stream);
int t2p_train_eval_loss(t2p_trainer* t, const t2p_train_batch* batch, float* loss_host, void* stream);
int64_t t2p_train_device_bytes(const t2p_trainer* t);
/* the strided fp32 GEMM of the backward pass: C[z][m][n] = alpha sum_k A(z,m,k) B(z,k,n) + beta C, element strides as given
 * (one stride of each operand must be 1); conv = 1: B is the 3x3 window gather of the NHWC map X [batch][H][W][conv_C] (weight
 * gradient of a convolution: K = batch H W, N = 9 conv_C, sBk/sBn ignored, ldx = sBz0); ksplit 0 = chosen by the library (needs beta = 1 when > 1) */
int t2p_op_tgemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBk, int64_t sBn, float* C, int64_t ldc, int M, int N,
                 int K, int nz, int64_t sAz, int64_t sBz, int64_t sCz, float alpha, float beta, const float* bias_n, int ksplit, int conv,
                 int H, int W, int conv_C, void* stream);
/* backward halves of the operators (gradients accumulate into dx / dgamma / dbeta / du) */
int t2p_op_groupnorm_backward(const float* x, const float* dy, const float* gamma, const float* beta, int silu, int batch, int HW, int C,
                              int groups, float eps, float* dx, float* dgamma, float* dbeta, void* stream);
int t2p_op_layernorm_backward(const float* x, const float* dy, const float* gamma, int64_t rows, int C, float eps, float* dx, float* dgamma,
                              float* dbeta, void* stream);
int t2p_op_softmax_backward(const float* P, float* dP_inout, int64_t rows, int n, float scale, void* stream);
int t2p_op_geglu_backward(const float* u, const float* dy, float* du, int64_t rows, int inner, void* stream);

/* ---- individual operators (parity tests call these through the same ABI) -------------------- */
int t2p_op_gemm(int dtype, const void* A, int a_f32, const void* Bw, void* C, int c_f32, int M, int N, int K,
                int64_t lda, int64_t ldb, int64_t ldc, const float* bias_n, const float* residual, float alpha,
                void* stream);
/* the same with the residual stored in the compute dtype (f16 / bf16) instead of fp32: the form the engine uses for
 * the residual stream between blocks in f16 mode */
int t2p_op_gemm_r16(int dtype, const void* A, int a_f32, const void* Bw, void* C, int c_f32, int M, int N, int K, int64_t lda,
                    int64_t ldb, int64_t ldc, const float* bias_n, const void* residual16, float alpha, void* stream);
/* x: NHWC in the given layout; w: [Cout][3][3][Cin] compute dtype; out fp32 NHWC */
int t2p_op_conv3x3(int dtype, const void* x, int a_f32, const void* w, const float* bias, float* out, int batch,
                   int H, int W, int Cin, int Cout, int upsample, void* stream);
/* second convolution of a residual block with its 1x1 shortcut in the same K loop (layers.py:322-327, h + Conv_2(x)):
 * out = alpha * (conv3x3(a) + [x0 | x1] Wx^T + bias); a, x0, x1: NHWC compute dtype (x1 may be NULL with CX1 = 0: the
 * second source of a concatenated block input); w: [Cout][9 * C + CX0 + CX1], the 3x3 taps first; out fp32 or compute
 * dtype.  16-bit modes on the LDS-DMA kernels only (C, CX0, CX1 multiples of 64): anything else is refused */
int t2p_op_conv3x3_shortcut(int dtype, const void* a, const void* w, const float* bias, const void* x0, int CX0,
                            const void* x1, int CX1, float alpha, void* out, int c_f32, int batch, int H, int W, int C,
                            int Cout, void* stream);
/* a 3x3 convolution FOLLOWED by a GroupNorm (+SiLU) of its output, as inside ResnetBlockBigGANpp.forward (layers.py:304-321:
 * h = Conv_0(...) + Dense_0(temb); h = act(GroupNorm_1(h)), and the block output -> the next block's GroupNorm_0): low-resolution
 * convolutions run with the K loop split over workgroups, and the pass that sums the partial tiles applies the norm too.
 * a [batch][H][W][C] and w [Cout][9 C] in the 16-bit compute dtype; bias [Cout], bias_bn [batch][Cout] (time-embedding bias) and
 * residual [batch][H][W][Cout] (compute dtype) optional; out (fp32 or compute dtype, may be null) = alpha (conv + biases + residual);
 * normed (compute dtype) = act(GroupNorm(out)); col_stats optional (per-64-row column sums of out, H W % 64 == 0).
 * T2P_ERR_INVALID when the launch would not take that plan (no workspace: t2p_debug_set(10, MiB); K loop too short; H W > 256) */
int t2p_op_conv3x3_groupnorm(int dtype, const void* a, const void* w, const float* bias, const float* bias_bn, const void* residual,
                             float alpha, int upsample, int groups, const float* gamma, const float* beta, float eps, int silu,
                             void* out, int out_f32, void* normed, float* col_stats, int batch, int H, int W, int C, int Cout, void* stream);
/* a 3x3 convolution of an 8x8 or 4x4 map with everything ResnetBlockBigGANpp.forward (layers.py:303-327) does around it, in ONE
 * launch with no second pass: a workgroup owns whole samples x whole GroupNorm groups (64 rows x 16 channels), so the norm that
 * follows is applied on the spot.  a [batch][H][W][C], w [Cout][ldw] (K index = tap C + c, then CX0 + CX1 shortcut columns read
 * at the output pixel from x0 | x1), residual [batch][H][W][Cout]: all in the 16-bit compute dtype; bias [Cout], bias_bn
 * [batch][Cout] fp32.  out (optional, fp32 or 16-bit) = alpha (conv + shortcut + biases + residual); col_stats (optional, 8x8
 * maps) its per-64-row column sums; normed (optional, 16-bit) = act(GroupNorm(out)) with `groups` groups of 8 or 16 channels.
 * H = W in {4, 8}, batch H W % 64 == 0, C % 32 == 0, Cout % 16 == 0. */
int t2p_op_small_conv_groupnorm(int dtype, const void* a, const void* w, int64_t ldw, const void* x0, int CX0, const void* x1, int CX1,
                                const float* bias, const float* bias_bn, const void* residual, float alpha, void* out, int out_f32,
                                float* col_stats, void* normed, int groups, const float* gamma, const float* beta, float eps, int silu,
                                int batch, int H, int W, int C, int Cout, void* stream);
/* row-wise chains of a SpatialTransformer block as ONE launch over 32-row blocks (model/attention.py:250-256 GroupNorm + proj_in,
 * :208-215 LayerNorm + residual adds, :170-193 to_q / to_k / to_v / to_out):
 *   t = a W_in^T + b_in (+ residual);  out2 = LayerNorm(t) W_2^T        (t [batch n][C], out2 [batch n][n2], 16-bit)
 * with a = GroupNorm(x) (statistics from col_stats, the per-64-row column sums of x, [batch n / 64][C][2]) or a = x (col_stats
 * NULL).  Entry of the block: W_in = proj_in, W_2 = to_q | to_k | to_v stacked (n2 = 3 C).  After the self-attention: x = its
 * output, W_in = to_out, residual = t (may alias the output t), W_2 = the cross-attention's to_q (n2 = C).  After the
 * cross-attention: the same with W_2 = ff.net.0 (n2 = 8 C, rows interleaved (value_j, gate_j), bias b_2) and geglu = 1:
 * out2 [batch n][4 C] = value * gelu_erf(gate) (model/attention.py:37-64).  With w_3 [C][5 C] (ff.net.2 and proj_out as one
 * matrix over [out2 | t], :213-215, 259-263), b_3 and res3 (the block input) a third product follows in the same launch:
 * y [batch n][C] = [out2 | t] W_3^T + b_3 + res3, y_stats its per-64-row column sums (out2 then stays on chip: out2 may be NULL).
 * x, w_in [C][C], w_2 [n2][C], residual in the 16-bit compute dtype.  C = 256, or C = 512 without geglu / w_3; n % 32 == 0
 * (n % 64 == 0 with col_stats), batch n <= 8192; anything else is refused */
int t2p_op_st_entry(int dtype, const void* x, const float* col_stats, int groups, const float* gn_gamma, const float* gn_beta,
                    float gn_eps, const void* w_in, const float* b_in, const void* residual, const float* ln_gamma,
                    const float* ln_beta, float ln_eps, const void* w_2, int n2, const float* b_2, int geglu, void* t, void* out2,
                    const void* w_3, const float* b_3, const void* res3, void* y, float* y_stats, int batch, int n, int C,
                    void* stream);
/* the projections of an AttnBlockpp (layers.py:160-167) in ONE launch over 32-row blocks at C = 256: h = GroupNorm(x) (statistics from
 * col_stats as in t2p_op_st_entry, or x already normalised), qk [batch n][2 C] = h [W_0 | W_1]^T + b (NIN_0 | NIN_1), and
 * vt [batch][C][npad] = (h W_v^T)^T: the value projection written transposed (the engine passes NIN_2 . NIN_3 as W_v).
 * x, w_qk [2 C][C], w_v [C][C], qk, vt in the 16-bit compute dtype.  n % 32 == 0, batch n <= 8192, npad >= n, npad % 4 == 0.
 * With k_fm (npad == n): the k half goes to k_fm and vt is written in the fragment-major order of t2p_op_attention_wide_fm instead */
int t2p_op_attn_proj(int dtype, const void* x, const float* col_stats, int groups, const float* gn_gamma, const float* gn_beta,
                     float gn_eps, const void* w_qk, const float* b_qk, const void* w_v, void* qk, void* vt, void* k_fm,
                     int64_t npad, int batch, int n, int C, void* stream);
/* the network's input convolution (pre_conv, ncsnpp.py:230: 3x3, C = 5 or 8 input channels -> nf) straight from the NCHW fp32
 * sample, in fp32 arithmetic: x [batch][C][H][W] fp32; w_tcn [3*3][C][nf] fp32 (tap-major); out NHWC [batch][H][W][nf] in
 * out_dtype.  col_stats (optional; W % 64 == 0, nf | 256): [batch H W / 64][nf][2] fp32 = (sum, sum of squares) of the fp32
 * results per 64-pixel chunk and channel -- what the first GroupNorm needs, so that it does not re-read the tensor.
 * 16-bit out_dtype with W % 64 == 0, an even H and nf in {64, 128, 256}: the layer runs on the 16-bit matrix pipe with every fp32
 * operand split into two f16 terms (fp32-class accuracy, ~2^-22; needs |x| < 65504 -- the engine takes this form for
 * sigma_max <= 4096); the call then returns after the stream has drained (it prepares the split weights in a temporary) */
int t2p_op_input_conv(const float* x, const float* w_tcn, const float* bias, void* out, int out_dtype, int batch, int C,
                      int H, int W, int nf, float* col_stats, void* stream);
int t2p_op_groupnorm(const float* x0, const float* x1, int C0, int C1, int batch, int H, int W, int groups,
                     const float* gamma, const float* beta, float eps, int silu, int down, void* out, int dtype,
                     void* stream);
int t2p_op_layernorm(const float* x, const float* gamma, const float* beta, void* out, int dtype, int64_t rows,
                     int C, float eps, void* stream);
/* the same on rows stored in the compute dtype (f16 / bf16), output in that dtype: the form the engine uses inside the
 * transformer blocks in 16-bit modes (attention.py:203-205) */
int t2p_op_layernorm16(const void* x16, const float* gamma, const float* beta, void* out16, int dtype, int64_t rows, int C,
                       float eps, void* stream);
int t2p_op_softmax(const float* S, int64_t lds, void* P, int64_t ldp, int dtype, int64_t rows, int n, float scale,
                   void* stream);
int t2p_op_geglu(const float* u, void* out, int dtype, int64_t rows, int inner, void* stream);
/* q: [B][nq][ldq], k: [B][nk][ldk] (head h at column h*d), vt: [B][heads*d][ldvt]; out [B][nq][heads*d].
 * Operands in compute dtype; scores/softmax in fp32.  workspace sizes via t2p_op_attention_ws. */
int64_t t2p_op_attention_ws(int dtype, int batch, int heads, int nq, int nk);
int t2p_op_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt,
                     int64_t ldvt, void* out, int batch, int heads, int nq, int nk, int d, float scale,
                     void* workspace, void* stream);
/* AttnBlockpp.forward after its projections (layers.py:168-176: w = softmax(q k^T / sqrt C) over ALL h w pixels, ONE head of
 * width d = C; h = w v; (x + NIN_3(h)) / sqrt 2) in one launch: q, k [batch][n][ld*] and vt = v^T [batch][d][ldvt] in the 16-bit
 * compute dtype -- the engine folds NIN_3 into the value projection (rows of w sum to 1), so what remains of the block's tail is
 * out[batch][n][d] = alpha (w v + bias[d] + residual[batch][n][d]), stored fp32 (out_f32) or in the compute dtype; col_stats
 * (optional, n % 64 == 0): [batch n / 64][d][2] column sums / sums of squares of out's fp32 values per 64 queries (the next
 * GroupNorm's input).  d = 256 / 512 / 1024, n <= 1024 (<= 512 at d = 1024), n % 8 == 0. */
int t2p_op_attention_wide(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* out,
                          int out_f32, const float* bias, const void* residual, int residual_16bit, float alpha, float* col_stats,
                          int batch, int n, int d, float scale, void* stream);
/* the same with k and vt FRAGMENT-MAJOR ([batch][n d] each: element (r, c) -- r = key, c = channel for k; r = channel, c = key for
 * vt -- at ((r / 32) (cols / 32) + c / 32) 1024 + ((c / 8) % 2) 512 + ((c / 16) % 2) 256 + (r % 32) 8 + c % 8): the kernel's fragment
 * loads are then 1 KiB contiguous.  d = 512 with 512 < n <= 1024, or d = 256 with n <= 256; n % 32 == 0.  Bit-identical to
 * t2p_op_attention_wide. */
int t2p_op_attention_wide_fm(int dtype, const void* q, int64_t ldq, const void* k_fm, const void* vt_fm, void* out, int out_f32,
                             const float* bias, const void* residual, int residual_16bit, float alpha, float* col_stats, int batch,
                             int n, int d, float scale, void* stream);
/* C = A Bw^T (16-bit, [M][N] row-major; batch > 1: A [M][K] shared, Bw [batch][N][K], C [batch][M][N]) with the columns from
 * frag_col0 on written fragment-major into c_frag instead (layout above; per sample of rows_per_batch rows, or per batch entry):
 * what the engine asks of the q | k projection and of the transposed value projection of an AttnBlockpp (layers.py:163-167).
 * Refused unless the product runs on the 256 x 256 register-epilogue kernel with whole 32 x 32 fragments */
int t2p_op_gemm_frag_major(int dtype, const void* A, const void* Bw, void* C, void* c_frag, int M, int N, int K, int frag_col0,
                           int rows_per_batch, int batch, void* stream);
/* self-attention on the output of ONE stacked projection (CrossAttention.forward with context = x, model/attention.py:
 * 170-191): qkv [batch][n][ld] holds q | k | v in three column blocks of heads*d (head h at column h*d of its block);
 * out [batch][n][heads*d].  V is read row-major -- the fused kernel transposes it on the way out of LDS -- so no
 * separate V^T projection is needed.  16-bit dtypes, d in {32, 64, 128} only (anything else is refused) */
int t2p_op_attention_qkv(int dtype, const void* qkv, int64_t ld, void* out, int batch, int heads, int n, int d,
                         float scale, void* stream);
int t2p_op_langevin(const float* x, const float* grad, const float* noise, const uint8_t* mask,
                    const float* x_initial, float* x_out, float* x_mean_out, int batch, int64_t per_sample,
                    float snr, float alpha, float* sums_out /* device float[2], may be NULL */, void* stream);
/* split form: norms -> (optional all-reduce of sums over ranks by the caller) -> update.
 * workspace: device float[batch * 128]; sums: device float[2] = {sum_b ||grad_b||, sum_b ||noise_b||} */
int t2p_op_langevin_norms(const float* grad, const float* noise, int batch, int64_t per_sample, float* workspace,
                          float* sums, void* stream);
int t2p_op_langevin_update(const float* x, const float* grad, const float* noise, const uint8_t* mask,
                           const float* x_initial, float* x_out, float* x_mean_out, int64_t n, const float* sums,
                           float batch_total, float snr, float alpha, void* stream);
int t2p_op_predictor(const float* x, const float* score, const float* noise, const uint8_t* mask,
                     const float* x_initial, float* x_out, float* x_mean_out, int64_t n, float G,
                     int probability_flow, void* stream);
int t2p_op_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);
int t2p_op_convert(const float* in, void* out, int dtype, int64_t n, void* stream);
/* ---- either side of the sampler ----
 * 6D decode of finished samples, sampling_rosetta.py:69-96: x = (batch, channels, L, L) fp32 with the padding mask in
 * the last channel.  lengths[b] = sqrt(#(round(mask) == 1)) or -1 when that count is not a perfect square (the
 * reference raises ValueError); clipped / absval = (batch, 4, L*L) fp32: per channel (dist, omega, theta, phi) the
 * first lengths[b]^2 entries are clip(x[c][mask == 1], -1, 1) in row-major order (= .reshape(L, L)) and its inverse
 * scaling (dist+1)*10, omega*pi, theta*pi, (phi+1)*pi/2. */
int t2p_op_decode_6d(const float* x, int batch, int channels, int L, float* clipped, float* absval, int32_t* lengths, void* stream);
/* text context = embed_tokens(ids), sampling_6d.py:134-137: out[(b,t)][:] = table[ids[(b,t)]][:] as fp32; the table
 * ([vocab][dim], fp32 / bf16 / f16 by table_dtype) stays resident.  *bad_flag (device int, zeroed by the caller) is
 * set to 1 when an id falls outside [0, vocab). */
int t2p_op_embedding_gather(const void* table, int table_dtype, const int32_t* ids, float* out, int64_t n_tokens, int dim, int vocab,
                            int32_t* bad_flag, void* stream);
/* x = where(mask, x, x_initial) (sampling.py:283,285,287) */
int t2p_op_apply_mask(float* x, const uint8_t* mask, const float* x_initial, int64_t n, void* stream);

/* ---- measurement hooks (bench.py): time every MFMA GEMM launch with HIP events on its stream ----
 * out9 = {LDS-DMA conv3x3: ms, flops, launches; other GEMMs: ...; conv3x3 on the register-staged kernel: ...} since t2p_profile_begin */
int t2p_profile_begin(void);
/* Plan switches for tests and A/B measurements: they select between kernel geometries / fusions that all
 * produce correct results (key 0 LDS-DMA GEMM on/off, 2 tile geometry, 3 split-K, 4..25 individual fusions and
 * kernel forms -- the full list with one line each: tools/README.md; text2protein_amd/csrc/capi.cpp).  Key 1 is the
 * timing-only ablation mask of the LDS-DMA kernel: its bits 128 / 256 (DMA issue order) and 4096 (LDS-staged instead of
 * register epilogue), results unchanged, are always accepted, the bits that skip work and so produce
 * WRONG results exist only in a library built with -DT2P_ABLATION (python -m text2protein_amd.build --ablation)
 * and are refused (status 1) by the product build. */
int t2p_debug_set(int key, int value);
/* 1 when the library was built with -DT2P_ABLATION (never the shipped one) */
int t2p_built_with_ablation(void);
int t2p_profile_end(double* out9);
/* after t2p_profile_end: the 3x3-convolution kernel instantiation with the largest total time in that region --
 * out4 = {ms (main kernel only), flops, launches, algorithmic bytes (inputs, weights, residual and output once each)},
 * name = the kernel name as rocprofv3 reports it (no argument list), so that bench.py's live average launch duration
 * can be checked against profiles/<round>_kernel_stats.csv and the PMC traffic against the algorithmic bytes */
int t2p_profile_dominant(double* out4, char* name, int name_len);
/* after t2p_profile_end: the fused-attention launches of that region (CrossAttention.forward, model/attention.py:181-191,
 * self and text cross-attention): out3 = {ms, flops (4 nq nk d per head), launches} */
int t2p_profile_attention(double* out3);
/* after t2p_profile_end: the GEMM / convolution launches of that region by operand shape, as CSV text
 * "kind,M,N,K,taps,batch,launches,ms,flops" (kind as in out9; K "a+b" = a channels per tap + b shortcut columns;
 * taps negative = gathered from the half-resolution map).  T2P_ERR_INVALID when the buffer is too small */
int t2p_profile_shapes(char* buf, int len);
/* per-block timing of the score network (UNetModel.forward, ncsnpp.py:220-263): HIP events on the launch stream at every block
 * boundary between _begin and _end; _end synchronises the device and writes CSV lines
 * "prefix,kind,map side,in channels,out channels,ms" in launch order (pre = embedding + input convolution, head = out.*) */
/* development: from now on every score evaluation also writes the output of block number `block_index` (launch order of
 * t2p_profile_layers_*, the pre / head entries not counted; -1 = off) to `dst` as fp32 NHWC [batch][H][W][C] (capacity in floats);
 * shape4 (optional) receives {C, H, W, stored-in-16-bit flag} of the LAST tap taken */
int t2p_debug_tap(int block_index, float* dst, int64_t capacity, int64_t* shape4);
int t2p_profile_layers_begin(void);
int t2p_profile_layers_end(char* buf, int len);

#ifdef __cplusplus
}
#endif
#endif /* T2P_H_ */
