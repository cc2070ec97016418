// Launchers for the non-GEMM kernels of the score network and the SDE update.
#pragma once
#include "t2p_common.h"

namespace t2p {

// ---- GroupNorm (reference nn.GroupNorm, layers.py:282,292; attention.py:77; ncsnpp.py:214) ----
// x is NHWC fp32, optionally the channel-concat of two sources.  stats = [B][G][2] (mean, rstd).
struct GroupNormArgs {
  const float* x0 = nullptr; const float* x1 = nullptr;
  int C0 = 0, C1 = 0;
  int B = 0, HW = 0;       // input pixels per sample
  int G = 0;
  float eps = 1e-6f;
  int lowp_dtype = DT_F32;   // DT_F16 / DT_BF16: x0 (and x1) hold 16-bit values
  float* partial = nullptr;  // workspace [B][nchunk][G][2], nchunk = gn_num_chunks(HW)
  float* stats = nullptr;    // [B][G][2]
};
int gn_num_chunks(int HW);
int launch_gn_stats(const GroupNormArgs& a, hipStream_t s);

// statistics from per-chunk column sums written by the GEMM epilogue (GemmParams::col_stats):
// cs0 / cs1 = [B * HW / 64][C0 or C1][2] for the two channel-concatenated sources
int launch_gn_finalize_cols(const float* cs0, const float* cs1, int C0, int C1, int B, int HW, int G, float eps,
                            float* stats, hipStream_t s);

struct GroupNormApplyArgs {
  const float* x0 = nullptr; const float* x1 = nullptr;
  int C0 = 0, C1 = 0, B = 0, H = 0, W = 0, G = 0;
  const float* stats = nullptr;       // [B][G][2]
  const float* gamma = nullptr; const float* beta = nullptr;
  int silu = 0;
  int down = 0;                        // 2x2 mean of the activated map (layers.py:185-188)
  void* out = nullptr;                 // [B][H'*W'][C] in `dtype`
  int dtype = DT_F32;
  int x0_lowp = 0;                     // x0 is stored in `dtype` (16-bit) instead of fp32 (single source only)
  void* raw_out = nullptr;             // optional: un-normalised x (concat of both sources) in `dtype`,
                                       // same resolution as the input (not with `down`)
  float eps = 0.f;                     // launch_gn_small / launch_gn_apply_cols only
  const float* cs0 = nullptr; const float* cs1 = nullptr;   // launch_gn_apply_cols: column statistics of the two sources
};
int launch_gn_apply(const GroupNormApplyArgs& a, hipStream_t s);
// 16-bit maps of <= 4096 pixels whose statistics arrive as per-64-row column sums (GEMM epilogues): finalize and apply in ONE
// launch -- a block owns a 64-channel slab (whole groups) of one sample, folds the slab's column sums itself, then normalises
bool gn_apply_cols_eligible(const GroupNormApplyArgs& a);
int launch_gn_apply_cols(const GroupNormApplyArgs& a, hipStream_t s);
// statistics + normalisation (+SiLU, + raw copy) of a small map in ONE launch (no `stats` input, no `down`)
bool gn_small_eligible(const GroupNormApplyArgs& a);
int launch_gn_small(const GroupNormApplyArgs& a, hipStream_t s);

// ---- LayerNorm over the last axis (attention.py:203-205), eps 1e-5 ---------------------------
int launch_layernorm(const float* x, const float* gamma, const float* beta, void* out, int dtype,
                     long rows, int C, float eps, hipStream_t s, int x_lowp = 0);

// ---- row softmax: P[r][0:n] = softmax(scale * S[r][0:n]); P[r][n:ldp] = 0 ----------------------
int launch_softmax(const float* S, long lds, void* P, long ldp, int dtype, long rows, int n, float scale,
                   hipStream_t s);

// ---- GEGLU (attention.py:37-44): out[r][j] = u[r][j] * gelu_erf(u[r][inner + j]) ----------------
int launch_geglu(const float* u, void* out, int dtype, long rows, int inner, hipStream_t s, int interleaved = 0);

// ---- 2x2 mean pooling of an NHWC fp32 map (skip branch of a down block, layers.py:309-311) ------
int launch_pool2x2(const float* x, void* out, int dtype, int B, int H, int W, int C, hipStream_t s, int x_lowp = 0);

// ---- pre_conv: 3x3, C in {5, 8} NCHW fp32 -> nf NHWC (fp32 or a 16-bit out_dtype); w = [9][C][nf] fp32 (tap-major, output channel contiguous) --
int launch_pre_conv(const float* x, const float* w, const float* bias, void* out, int out_dtype, int B, int C, int H, int W, int nf,
                    hipStream_t s, float* cstats = nullptr);
bool pre_conv_fuses_col_stats(int W, int nf);
// the same layer on the 16-bit matrix pipe with every fp32 operand split into two f16 terms (fp32-class accuracy, plan switch 38):
// wsplit = the weights in that form (pre_conv_split_weight_bytes bytes, written by launch_pre_conv_split_weights from w [9][C][nf])
extern bool g_pre_conv_split;
size_t pre_conv_split_weight_bytes(int C, int nf);
bool pre_conv_split_ok(int out_dtype, int C, int H, int W, int nf);
int launch_pre_conv_split_weights(const float* w, void* wsplit, int C, int nf, hipStream_t s);
int launch_pre_conv_split(const float* x, const void* wsplit, const float* bias, void* out, int out_dtype, int B, int C, int H, int W, int nf,
                          hipStream_t s, float* cstats = nullptr);

// ---- NCHW fp32 (B,C,L,L) -> NHWC fp32 [B][L*L][Cpad], zero padded channels -----------------------
int launch_nchw_to_nhwc(const float* x, float* out, int B, int C, int HW, int Cpad, hipStream_t s);

// ---- time embedding ------------------------------------------------------------------------------
// emb[r][:] = [sin(t f_k) | cos(t f_k)] (layers.py:97-111); label of row r = labels ? labels[r]
// : *step_counter (device scalar) -- the sampler advances a device-side counter so that a
// captured graph of one PC step can be replayed.
// labels_f (optional): fractional time values for the embedding (VP path: labels = t * (N - 1),
// reference models/utils.py:150-152); the sigma lookup always uses the integer labels
// label_table (optional, with step_counter): the label is label_table[*step_counter] instead of *step_counter
// label_f_table (optional, with step_counter): fractional label of loop step i (VP SDE in the fused sampler)
int launch_timestep_embedding(const int* labels, const float* labels_f, const int* step_counter, float* emb, int rows,
                              int dim, hipStream_t s, const int* label_table = nullptr, int n_table = 0, const float* label_f_table = nullptr);
// out[r][n] = bias[n] + sum_k act(in[r][k]) * W[n][k]   (fp32; act = SiLU when silu != 0)
int launch_small_linear(const float* in, const float* W, const float* bias, float* out, int rows, int K, int N,
                        int silu, hipStream_t s);

// ---- reverse-diffusion predictor + Langevin corrector (sampling.py:157-199) ------------------------
// sums[0] = sum_b ||grad_b||_2, sums[1] = sum_b ||noise_b||_2  (per-sample norms, summed over b)
int launch_langevin_norms(const float* grad, const float* noise, int B, long per_sample, float* sq_ws,
                          float* sums, hipStream_t s);
struct SdeUpdateArgs {
  const float* x = nullptr;        // current state (B, C, L, L) fp32
  const float* score = nullptr;    // network output
  const float* noise = nullptr;    // standard normal draws
  const unsigned char* mask = nullptr;  // conditional_mask (1 = free), may be null
  const float* x_initial = nullptr;
  float* x_out = nullptr;
  float* x_mean_out = nullptr;     // un-masked x_mean (masking of the final x_mean: sampling.py:287)
  long n = 0;                      // total elements
};
// corrector: step = (snr * mean_norm_noise / mean_norm_grad)^2 * 2 * alpha, with the two means
// = sums[] / batch_total (batch_total may exceed the local batch: global-batch semantics)
// alpha_table (optional): alpha = alpha_table[*step_counter] (VP: sde.alphas[timestep], sampling.py:184-186)
int launch_langevin_update(const SdeUpdateArgs& a, const float* sums, float batch_total, float snr, float alpha,
                           hipStream_t s, const float* alpha_table = nullptr, const int* step_counter = nullptr, int n_table = 0);
// predictor: x_mean = x + G^2 * score * (0.5 if probability_flow); x = x_mean + (0 if pf else G) z
// G = G_table[*step_counter] when G_table != null, else G_value
// x_coef_table (optional, with G_table): x_mean = x_coef[step] x + ... (VP: 2 - sqrt(alpha), sde_lib.py:148-157)
int launch_predictor_update(const SdeUpdateArgs& a, const float* G_table, const int* step_counter, float G_value,
                            int probability_flow, hipStream_t s, int n_table = 0, const float* x_coef_table = nullptr);
int launch_scale_by_table(float* x, long n, const float* table, const int* step_counter, int n_table, hipStream_t s);
// noise[i] = N(0,1) from Philox4x32-10 keyed by (seed, stream); counter = element index / 4
int launch_philox_normal(float* out, long n, unsigned long long seed, unsigned long long stream,
                         const int* step_counter, hipStream_t s);
int launch_add_int(int* counter, int delta, hipStream_t s);
int launch_scale(float* x, long n, float a, hipStream_t s);

// ---- fused attention (attention.hip): 16-bit dtypes, head dim 32 / 64 / 128 -------------------------
bool attention_flash_eligible(int dtype, int d, long ldq, long ldk, long ldvt, long ldo);
int launch_attention_flash(int dtype, const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt,
                           void* out, int B, int heads, int nq, int nk, int d, float scale, hipStream_t s, bool v_rowmajor = false);

// ---- single-head attention with a wide head (AttnBlockpp, layers.py:160-176): d = 256 / 512 / 1024, n <= 1024, one launch --------
extern bool g_attn_strip;
bool attention_strip_eligible(int dtype, int heads, int nq, int nk, int d, long ldq, long ldk, long ldvt, long ldo);
// optional epilogue: out = alpha (attention + bias[channel] + residual[query][channel]), stored fp32 or in the compute dtype, with
// the per-64-query column sums / sums of squares of the stored fp32 values (col_stats [B n / 64][d][2], n % 64 == 0)
struct StripEpilogue {
  const float* bias = nullptr;
  const void* residual = nullptr;
  int r_lowp = 0;
  long ldr = 0;
  float alpha = 1.f;
  int out_f32 = 0;
  float* col_stats = nullptr;
  int frag_major = 0;      // k and vt are fragment-major [B][n d] (GemmParams::c_frag of their projections; attention_strip_frag_major_ok)
};
bool attention_strip_frag_major_ok(int dtype, int n, int d);
int launch_attention_strip(int dtype, const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                           int B, int n, int d, float scale, hipStream_t s, const StripEpilogue* ep = nullptr);

// ---- 3x3 convolution of the 8x8 / 4x4 levels + the GroupNorm (+SiLU) that follows, one launch (smallconv.hip) ----------------------
// out[row][n] = alpha (sum_{tap, c} A[pixel(row, tap)][c] Wt[n][tap C + c] + sum_c X[row][c] Wt[n][9 C + c] + bias[n]
//                      + bias_bn[sample][n] + R[row][n]);   normed = act(GroupNorm(out))
struct SmallConvArgs {
  int dtype = DT_F16;
  const void* A = nullptr;      // [B H W][C]     16-bit, dense
  int B = 0, H = 0, W = 0, C = 0, N = 0;
  const void* Wt = nullptr;     // [N][ldw]       16-bit, K index = tap * C + c, then the shortcut columns
  long ldw = 0;
  int w_fm = 0;                 // Wt is the fragment-major copy of the [N][ldw] matrix (launch_sf_frag_major; ldw % 32 == 0)
  const void* X0 = nullptr; const void* X1 = nullptr;   // optional shortcut sources [B H W][CX0], [B H W][CX1] (16-bit, dense)
  int CX0 = 0, CX1 = 0;
  const float* bias = nullptr;      // [N]
  const float* bias_bn = nullptr;   // [B][ld_bn] (time-embedding bias)
  long ld_bn = 0;
  const void* R = nullptr;          // residual [B H W][N], 16-bit
  float alpha = 1.f;
  void* out = nullptr;              // raw result [B H W][N], fp32 (out_f32) or 16-bit; may be null when `normed` is wanted only
  int out_f32 = 0;
  float* col_stats = nullptr;       // [B H W / 64][N][2] column sums / sums of squares of the raw fp32 values (8x8 maps)
  void* normed = nullptr;           // act(GroupNorm(out)) [B H W][N], 16-bit; null = no norm
  const float* gn_gamma = nullptr; const float* gn_beta = nullptr;
  int groups = 0, gn_silu = 0;
  float gn_eps = 1e-6f;
};
extern bool g_small_conv;

// ---- row-block chains of a SpatialTransformer block in one launch (stfuse.hip; plan switch 39) -------------------------------------
// st_entry: a = GroupNorm(x) -> t = proj_in(a) + b -> LayerNorm_1(t) -> q | k | v  (model/attention.py:250-256, 208-213, 170-176)
struct StEntryArgs {
  int dtype = DT_F16;
  int B = 0, n = 0, C = 0;               // samples, tokens per sample (H W), channels
  const void* x = nullptr;               // [B n][C] 16-bit: the block input (raw with cstats, already normalised without)
  const float* cstats = nullptr;         // per-64-row column sums of x ([B n / 64][C][2]) or null
  const float* gn_gamma = nullptr; const float* gn_beta = nullptr; int groups = 0; float gn_eps = 1e-6f;
  // the two weight matrices are FRAGMENT-MAJOR copies (launch_sf_frag_major of the row-major [N][K] 16-bit matrices)
  const void* w_in = nullptr; const float* b_in = nullptr;       // proj_in [C][C] 16-bit, bias [C]
  const float* ln_gamma = nullptr; const float* ln_beta = nullptr; float ln_eps = 1e-5f;
  const void* res = nullptr;             // optional [B n][C] 16-bit residual of the first product (may be `t` itself: rows are private)
  const void* w_qkv = nullptr;           // [n2][C] 16-bit: to_q | to_k | to_v stacked (n2 = 3 C), or one projection (n2 = C); no bias
  int n2 = 0;
  const float* b2 = nullptr;             // optional bias of the second product [n2]
  int geglu = 0;                         // n2 = 8 C interleaved (value, gate) columns -> out2 [B n][n2 / 2] = value * gelu_erf(gate)
  // optional third product (with geglu): y = [out2 | t] W_3^T + b_3 + res3: w3 [C][5 C] fragment-major, res3 / y [B n][C] 16-bit;
  // out2 then stays on chip (qkv unused).  y_stats: per-64-row column sums of y, ACCUMULATED (zeroed beforehand: `zero` of an earlier launch)
  const void* w3 = nullptr; const float* b3 = nullptr; const void* res3 = nullptr; void* y = nullptr; float* y_stats = nullptr;
  float* zero = nullptr; long zero_n = 0;   // optional: this launch zeroes zero[0 .. zero_n)
  void* t = nullptr;                     // out [B n][C] 16-bit: the block's residual stream
  void* qkv = nullptr;                   // out [B n][n2] 16-bit
};
// the projections of an AttnBlockpp in one launch (C = 256): GroupNorm apply -> q | k (row-major [B n][2 C], + bias) and the merged
// value projection written transposed (vt [B][C][npad]); weights fragment-major (launch_sf_frag_major); x / cstats as in StEntryArgs
struct AttnProjArgs {
  int dtype = DT_F16;
  int B = 0, n = 0, C = 0;
  long npad = 0;
  const void* x = nullptr; const float* cstats = nullptr;
  const float* gn_gamma = nullptr; const float* gn_beta = nullptr; int groups = 0; float gn_eps = 1e-6f;
  const void* w_qk = nullptr; const float* b_qk = nullptr;     // [2 C][C], bias [2 C]
  const void* w_v = nullptr;                                   // [C][C] (no bias: the attention kernel's epilogue adds it)
  void* qk = nullptr; void* vt = nullptr;
  void* k_fm = nullptr;      // optional: the k half goes here FRAGMENT-MAJOR ([B][n C]) and vt is written fragment-major as well (npad == n):
                             // the operands of attn_strip_kernel<.., FM> (attention_strip_frag_major_ok)
};
bool attn_proj_eligible(const AttnProjArgs& a);
int launch_attn_proj(const AttnProjArgs& a, hipStream_t s);
extern bool g_st_fuse;
extern bool g_small_conv_fm;   // engine.cpp (plan switch 41)
extern bool g_attn_proj;       // engine.cpp (plan switch 46)
extern bool g_attn_fm;         // engine.cpp (plan switch 45)
extern int g_st_tail_rows;     // engine.cpp (development key 43)
extern bool g_st_ffpo;         // engine.cpp (plan switch 42)
extern bool g_st_tail;     // engine.cpp (plan switch 40)
bool st_entry_eligible(const StEntryArgs& a);
int launch_st_entry(const StEntryArgs& a, hipStream_t s);
// out = W [N][K] (16-bit, row-major) re-ordered so that every MFMA fragment of 16 rows x 32 K is 1 KiB contiguous (same size)
int launch_sf_frag_major(int dtype, const void* W, void* out, int N, int K, hipStream_t s);
bool small_conv_eligible(const SmallConvArgs& a);
int launch_small_conv_gn(const SmallConvArgs& a, hipStream_t s);

int launch_convert(const float* in, void* out, int dtype, long n, hipStream_t s);
int launch_widen(const void* in, int dtype, float* out, long n, hipStream_t s);   // compute dtype -> fp32
int launch_gather_label(const int* labels, const int* step_counter, const float* table, float* out, int B, int N,
                        hipStream_t s, const int* label_table = nullptr);
int launch_apply_mask(float* x, const unsigned char* mask, const float* x_initial, long n, hipStream_t s);
int launch_decode6d(const float* x, int B, int C, int L, float* clipped, float* absval, int* lengths, hipStream_t s);
int launch_embedding_gather(const void* table, int dtype, const int* ids, float* out, long ntok, int dim, int vocab, int* bad, hipStream_t s);

}  // namespace t2p
