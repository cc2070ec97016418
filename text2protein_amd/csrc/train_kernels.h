// Launchers of the TRAINING step's kernels (train_kernels.hip): the backward halves of the score network's operators, the
// denoising score-matching loss and the optimizer / EMA updates (reference score_sde_pytorch/losses.py:26-186,
// score_sde_pytorch/models/ema.py:32-49).  fp32 throughout (exact-f32 MFMA, v_mfma_f32_32x32x2_f32): first slice of
// SURVEY.md 8(f)4.  Every gradient output ACCUMULATES (+=) into its destination unless stated.
#pragma once
#include "t2p_common.h"

namespace t2p {

// ---- strided batched GEMM, fp32:  C[z][m][n] = alpha * sum_k A(z, m, k) B(z, k, n) + bias_n[n] + beta * C[z][m][n] ----------------
// A(z, m, k) = A[z0 sAz0 + z1 sAz1 + m sAm + k sAk], B(z, k, n) = B[z0 sBz0 + z1 sBz1 + k sBk + n sBn]: one of the two strides of
// an operand must be 1 (row- or column-major view), which covers x W^T, dY W, dY^T X, q k^T per head, P v, P^T dO ... without a
// transposed copy.  ksplit > 1 cuts the K loop over workgroups which add their partial tiles with hardware fp32 atomics (weight
// gradients: K = every pixel of the batch); it requires beta == 1 (accumulate into an initialised C).  ksplit == 0: chosen here.
struct TGemmArgs {
  const float* A = nullptr; long sAm = 0, sAk = 0, sAz0 = 0, sAz1 = 0;
  const float* B = nullptr; long sBk = 0, sBn = 0, sBz0 = 0, sBz1 = 0;
  float* C = nullptr; long ldc = 0, sCz0 = 0, sCz1 = 0;
  int M = 0, N = 0, K = 0, nz0 = 1, nz1 = 1;
  float alpha = 1.f, beta = 0.f;
  const float* bias_n = nullptr;
  int ksplit = 1;
  // weight gradient of a 3x3 convolution (layers.py:89-95): B is gathered, B(k = output pixel r, n = tap * conv_C + c) =
  // X[pixel(r) + offset(tap)][c] (zero outside the H x W map), X = [batch H W][ldx]; K = batch H W, N = 9 conv_C
  int conv_b = 0, H = 0, W = 0, conv_C = 0; long ldx = 0;
};
int launch_tgemm(const TGemmArgs& a, hipStream_t s);

// ---- GroupNorm backward (nn.GroupNorm + optional SiLU, layers.py:282,304,317; attention.py:77) -------------------------------------
// y = act(gamma (x - mean) rstd + beta); x, dy, dx: NHWC [B][HW][C] fp32; stats [B][G][2] = (mean, rstd) of the forward pass.
// dx += ..., dgamma += ..., dbeta += ...;  ws: gn_bwd_ws_floats(B, HW, C) floats
long gn_bwd_ws_floats(int B, int HW, int C, int G);
int launch_gn_backward(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, int silu,
                       int B, int HW, int C, int G, float* dx, float* dgamma, float* dbeta, float* ws, hipStream_t s);
// ---- LayerNorm backward (attention.py:203-205), rows x C, eps as the forward -----------------------------------------------------
int launch_ln_backward(const float* x, const float* dy, const float* gamma, long rows, int C, float eps, float* dx, float* dgamma,
                       float* dbeta, hipStream_t s);
// ---- softmax backward, in place: dP[r][j] <- scale P[r][j] (dP[r][j] - sum_j dP[r][j] P[r][j])  (P = softmax(scale S)) ------------------
int launch_softmax_backward(const float* P, float* dP, long rows, int n, float scale, hipStream_t s);
// ---- GEGLU backward (attention.py:37-44): u = [a | g], out = a gelu(g);  du += [dy gelu(g) | dy a gelu'(g)] ----------------------------
int launch_geglu_backward(const float* u, const float* dy, float* du, long rows, int inner, hipStream_t s);
// ---- elementwise ---------------------------------------------------------------------------------------------------------------------
int launch_silu(const float* x, float* y, long n, hipStream_t s);                                   // y = x sigmoid(x)
int launch_silu_backward(const float* x, const float* dy, float* dx, long n, hipStream_t s);        // dx += dy silu'(x)
int launch_axpy(float* y, const float* x, float a, long n, hipStream_t s);                          // y += a x
int launch_add_scale(const float* a, const float* b, float alpha, float* out, long n, hipStream_t s);   // out = alpha (a + b)
// dst[r][dst_off + c] (+)= src[r][src_off + c], c < C  (channel concat of the U-Net skips, ncsnpp.py:250, and its backward)
int launch_copy_cols(const float* src, long ld_src, long src_off, float* dst, long ld_dst, long dst_off, long rows, int C, int accumulate,
                     hipStream_t s);
// out[n] += sum_r dy[r][n]                      (bias gradients)
int launch_colsum(const float* dy, long rows, int N, long ld, float* out, hipStream_t s);
// out[b][n] (+)= sum_p dy[b][p][n]              (gradient of the per-sample time-embedding bias, layers.py:316)
int launch_colsum_per_sample(const float* dy, int B, int HW, int N, float* out, long ld_out, int accumulate, hipStream_t s);
// nearest 2x up-sampling / 2x2 mean down-sampling of NHWC maps (layers.py:179-188) and their backward (+=)
int launch_up2(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);               // x [B][H][W][C] -> y [B][2H][2W][C]
int launch_up2_backward(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s);
int launch_down2(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);             // x [B][H][W][C] -> y [B][H/2][W/2][C]
int launch_down2_backward(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s);
// Dropout_0 (layers.py:318) with a given keep-mask (uint8, same layout as x): y = x keep / (1 - p);  backward dx += dy keep / (1 - p)
int launch_dropout(const float* x, const unsigned char* keep, float inv_keep, float* y, long n, int accumulate, hipStream_t s);
// keep-masks on the device: Philox4x32-10 keyed by (seed, stream id), element i kept iff its uniform >= p
int launch_dropout_mask(unsigned char* keep, long n, float p, unsigned long long seed, unsigned long long stream_id, hipStream_t s);

// ---- convolution weights between the reference layout and the kernels' layouts ----------------------------------------------------------
// w [Co][Ci][3][3] (nn.Conv2d) -> wf [Co][9][Cip] (forward implicit GEMM: K index = tap Cip + ci) and
//                                 wd [Ci][9][Cop] (input gradient = the same kernel on dY: wd[ci][t][co] = w[co][ci][8 - t]); pads are zero
int launch_conv_w_prep(const float* w, float* wf, float* wd, int Co, int Ci, int Cip, int Cop, hipStream_t s);
// gw [Co][Ci][3][3] += dwc [Co][9][Cip]
int launch_conv_w_grad_fold(const float* dwc, float* gw, int Co, int Ci, int Cip, hipStream_t s);

// ---- denoising score matching, VE SDE (losses.py:105-131) --------------------------------------------------------------------------------
// per sample: t[b] (given, or drawn here: t = eps + (T - eps) u, u from Philox keyed by (seed, step) -- losses.py:106 with T = 1),
// std[b] = sigma_min (sigma_max / sigma_min)^t (VESDE.marginal_prob, sde_lib.py:225-228), label[b] = round((1 - t) (N - 1)) (half to even:
// get_score_fn's VE branch, models/utils.py:165-168) and scale[b] = scale_by_sigma ? 1 / sigmas[label] : 1 (ncsnpp.py:256-261)
int launch_dsm_prepare(const float* t_in, int B, float t_eps, float sigma_min, float sigma_max, int N, const float* inv_sigma_table,
                       unsigned long long seed, unsigned long long step, float* t_out, float* std, int* labels, float* scale, hipStream_t s);
// mask[b][c][y][x] = mask_pair[b][y][x] && conditional_mask (length: c != C - 1; ss: c not in 4..6; inpainting: mask_inpaint[b][y][x]);
// perturbed = mask ? x + std[b] z : x;  num_elem[b] = #mask   (all NCHW; cond_flags: 1 length, 2 ss, 4 inpainting)
int launch_dsm_perturb(const float* x, const float* z, const float* std, const unsigned char* mask_pair, const unsigned char* mask_inpaint,
                       int cond_flags, int B, int C, int L, float* perturbed, unsigned char* mask, float* num_elem, hipStream_t s);
// o: the head convolution's output NHWC [B][L L][ldo] (before the division by sigma); score = o inv_sigma[b];
// r = score std[b] + z; loss_sum[b] += sum mask r^2 (double);  d_o [B][L L][ld_do] = 2 r mask std inv_sigma / ((num_elem + 1e-8) B), pad columns zero
int launch_dsm_loss(const float* o, long ldo, const float* z, const float* std, const float* inv_sigma, const unsigned char* mask,
                    const float* num_elem, int B, int C, int L, double* loss_sum, float* d_o, long ld_do, float* score_nchw, hipStream_t s);
// loss = mean_b loss_sum[b] / (num_elem[b] + 1e-8)
int launch_dsm_finish(const double* loss_sum, const float* num_elem, int B, float* loss, hipStream_t s);

// ---- optimizer (losses.py:26-51: Adam, warm-up, clip_grad_norm_) and EMA (ema.py:32-49) over flat parameter buffers ------------------------
int launch_sumsq(const float* g, long n, double* out, hipStream_t s);      // *out += sum g^2   (zero it first)
struct AdamArgs {
  float* p = nullptr; float* g = nullptr; float* m = nullptr; float* v = nullptr; long n = 0;
  float lr = 0.f, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f, weight_decay = 0.f;
  float bias1 = 1.f, bias2_sqrt = 1.f;    // 1 - beta1^k, sqrt(1 - beta2^k)
  float grad_clip = -1.f;                 // >= 0: g *= min(1, grad_clip / (sqrt(*sumsq) + 1e-6)) first (written back, as clip_grad_norm_ does)
  const double* sumsq = nullptr;
};
int launch_adam(const AdamArgs& a, hipStream_t s);
int launch_ema(float* shadow, const float* p, float one_minus_decay, long n, hipStream_t s);   // shadow -= (1 - d) (shadow - p)

}  // namespace t2p
