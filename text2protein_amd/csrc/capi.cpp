// extern "C" surface of libt2p_hip.so (declarations and reference citations: include/t2p.h).
#include <new>

#include "engine.h"

using namespace t2p;

#define API_BEGIN try {
#define API_END                                        \
  }                                                    \
  catch (const std::bad_alloc&) {                      \
    set_last_error("out of host memory");              \
    return T2P_ERR_STATE;                              \
  }                                                    \
  catch (const std::exception& ex) {                   \
    set_last_error(std::string("exception: ") + ex.what()); \
    return T2P_ERR_STATE;                              \
  }

namespace t2p { extern bool g_pre_conv_mfma, g_gn_apply_cols; void layer_profile_begin(); int layer_profile_end(std::string* out);
void debug_tap_set(int index, float* dst, long capacity); void debug_tap_shape(long out[4]); }
extern "C" {

const char* t2p_last_error(void) { return get_last_error(); }

int t2p_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

int t2p_engine_create(const t2p_model_config* cfg, t2p_engine** out) {
  API_BEGIN
  T2P_REQUIRE(cfg && out, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_last_error("no HIP device: libt2p_hip has no CPU fallback");
    return T2P_ERR_HIP;
  }
  t2p_engine* e = new t2p_engine(*cfg);
  int rc = e->impl.build();
  if (rc != T2P_OK) {
    delete e;
    return rc;
  }
  *out = e;
  return T2P_OK;
  API_END
}

void t2p_engine_destroy(t2p_engine* e) { delete e; }

int t2p_engine_num_params(const t2p_engine* e) { return e ? (int)e->impl.params().size() : 0; }

int t2p_engine_param_info(const t2p_engine* e, int i, const char** name, int64_t shape[4], int* ndim) {
  API_BEGIN
  T2P_REQUIRE(e && name && shape && ndim, "null argument");
  const auto& ps = e->impl.params();
  T2P_REQUIRE(i >= 0 && i < (int)ps.size(), "parameter index out of range");
  *name = ps[i].name.c_str();
  *ndim = (int)ps[i].shape.size();
  for (int k = 0; k < 4; ++k) shape[k] = k < *ndim ? ps[i].shape[k] : 1;
  return T2P_OK;
  API_END
}

int t2p_engine_load_param(t2p_engine* e, const char* name, const float* host_data, const int64_t* shape, int ndim) {
  API_BEGIN
  T2P_REQUIRE(e, "null engine");
  return e->impl.load_param(name, host_data, shape, ndim);
  API_END
}

int t2p_engine_finalize(t2p_engine* e) {
  API_BEGIN
  T2P_REQUIRE(e, "null engine");
  return e->impl.finalize();
  API_END
}

int t2p_engine_set_context(t2p_engine* e, const float* context, int batch, int tokens, void* stream) {
  API_BEGIN
  T2P_REQUIRE(e, "null engine");
  return e->impl.set_context(context, batch, tokens, (hipStream_t)stream);
  API_END
}

int t2p_engine_score(t2p_engine* e, const float* x, const int32_t* labels, float* out, int batch, void* stream) {
  API_BEGIN
  T2P_REQUIRE(e && labels, "null argument");
  return e->impl.score(x, labels, nullptr, out, batch, (hipStream_t)stream);
  API_END
}

int t2p_engine_score_ex(t2p_engine* e, const float* x, const int32_t* labels, const float* labels_f, float* out, int batch,
                        void* stream) {
  API_BEGIN
  T2P_REQUIRE(e && labels, "null argument");
  return e->impl.score(x, labels, nullptr, out, batch, (hipStream_t)stream, labels_f);
  API_END
}

int64_t t2p_engine_device_bytes(const t2p_engine* e) { return e ? e->impl.device_bytes() : 0; }
int t2p_engine_pool_reclaimed(const t2p_engine* e) { return e ? e->impl.pool_reclaimed() : -1; }

int t2p_sampler_create(t2p_engine* e, const t2p_sampler_config* cfg, const float* g_table, const int32_t* label_table,
                       t2p_sampler** out) {
  API_BEGIN
  T2P_REQUIRE(e && cfg && out, "null argument");
  t2p_sampler* s = new t2p_sampler(&e->impl, *cfg);
  int rc = s->impl.init(g_table, label_table);
  if (rc != T2P_OK) {
    delete s;
    return rc;
  }
  *out = s;
  return T2P_OK;
  API_END
}

void t2p_sampler_destroy(t2p_sampler* s) { delete s; }

int t2p_sampler_set_condition(t2p_sampler* s, const uint8_t* mask, const float* x_initial) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  T2P_REQUIRE((mask == nullptr) == (x_initial == nullptr), "mask and x_initial go together");
  return s->impl.set_condition(mask, x_initial);
  API_END
}

int t2p_sampler_set_seed(t2p_sampler* s, uint64_t seed) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  s->impl.set_seed(seed);
  return T2P_OK;
  API_END
}

int t2p_sampler_set_norm_allreduce(t2p_sampler* s, float* device_sums2, t2p_allreduce_fn fn, void* user) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.set_norm_allreduce(device_sums2, fn, user);
  API_END
}

int t2p_sampler_set_vp_tables(t2p_sampler* s, const float* label_f, const float* score_scale, const float* x_coef, const float* corr_alpha) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.set_vp_tables(label_f, score_scale, x_coef, corr_alpha);
  API_END
}

int t2p_sampler_reset(t2p_sampler* s, int step, void* stream) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.reset(step, (hipStream_t)stream);
  API_END
}

int t2p_sampler_step(t2p_sampler* s, float* x, float* x_mean, const float* noise_corrector, const float* noise_predictor,
                     void* stream) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.step(x, x_mean, noise_corrector, noise_predictor, (hipStream_t)stream);
  API_END
}

int t2p_sampler_step_graph(t2p_sampler* s, float* x, float* x_mean, void* stream) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.step_graph(x, x_mean, (hipStream_t)stream);
  API_END
}

int t2p_sampler_count_dispatches(t2p_sampler* s, float* x, float* x_mean, void* stream, int* n_out) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.count_dispatches(x, x_mean, (hipStream_t)stream, n_out);
  API_END
}

int t2p_sampler_run(t2p_sampler* s, float* x, float* out, int prior_given, int n_steps, void* stream) {
  API_BEGIN
  T2P_REQUIRE(s, "null sampler");
  return s->impl.run(x, out, prior_given, n_steps, (hipStream_t)stream);
  API_END
}

// ---- operators ------------------------------------------------------------------------------------
int t2p_op_gemm(int dtype, const void* A, int a_f32, const void* Bw, void* C, int c_f32, int M, int N, int K, int64_t lda,
                int64_t ldb, int64_t ldc, const float* bias_n, const float* residual, float alpha, void* stream) {
  API_BEGIN
  GemmParams p;
  p.dtype = dtype; p.A0 = A; p.a_f32 = a_f32; p.C0 = K; p.lda0 = lda; p.Bw = Bw; p.ldb = ldb; p.M = M; p.N = N;
  p.bias_n = bias_n; p.R = residual; p.ldr = ldc; p.alpha = alpha; p.C = C; p.c_f32 = c_f32; p.ldc = ldc;
  return launch_gemm(p, (hipStream_t)stream);
  API_END
}

static void* g_op_ws = nullptr;
static size_t g_op_ws_bytes = 0, g_op_ws_alloc = 0;
// development switch (t2p_debug_set key 10): a split-K workspace for the op entries, so that they can take that plan
static int attach_op_ws(GemmParams& p) {
  if (!g_op_ws_bytes) return T2P_OK;
  if (g_op_ws_alloc < g_op_ws_bytes) {
    if (g_op_ws) { T2P_HIP_CHECK(hipDeviceSynchronize()); T2P_HIP_CHECK(hipFree(g_op_ws)); g_op_ws = nullptr; g_op_ws_alloc = 0; }
    T2P_HIP_CHECK(hipMalloc(&g_op_ws, g_op_ws_bytes));
    g_op_ws_alloc = g_op_ws_bytes;
  }
  p.ws = g_op_ws; p.ws_bytes = g_op_ws_bytes;
  return T2P_OK;
}

int t2p_op_gemm_r16(int dtype, const void* A, int a_f32, const void* Bw, void* C, int c_f32, int M, int N, int K, int64_t lda,
                    int64_t ldb, int64_t ldc, const float* bias_n, const void* residual16, float alpha, void* stream) {
  API_BEGIN
  GemmParams p;
  p.dtype = dtype; p.A0 = A; p.a_f32 = a_f32; p.C0 = K; p.lda0 = lda; p.Bw = Bw; p.ldb = ldb; p.M = M; p.N = N;
  p.bias_n = bias_n; p.R = (const float*)residual16; p.r_lowp = 1; p.ldr = ldc; p.alpha = alpha; p.C = C; p.c_f32 = c_f32; p.ldc = ldc;
  T2P_TRY(attach_op_ws(p));
  return launch_gemm(p, (hipStream_t)stream);
  API_END
}

int t2p_op_conv3x3(int dtype, const void* x, int a_f32, const void* w, const float* bias, float* out, int batch, int H,
                   int W, int Cin, int Cout, int upsample, void* stream) {
  API_BEGIN
  GemmParams p;
  p.dtype = dtype; p.A0 = x; p.a_f32 = a_f32; p.C0 = Cin; p.lda0 = Cin; p.taps = 9; p.H = H; p.W = W; p.a_up = upsample;
  p.Bw = w; p.ldb = 9L * Cin; p.M = batch * H * W; p.N = Cout; p.bias_n = bias; p.rows_per_batch = H * W;
  p.C = out; p.c_f32 = 1; p.ldc = Cout;
  T2P_TRY(attach_op_ws(p));
  return launch_gemm(p, (hipStream_t)stream);
  API_END
}

int t2p_op_conv3x3_shortcut(int dtype, const void* a, const void* w, const float* bias, const void* x0, int CX0, const void* x1,
                            int CX1, float alpha, void* out, int c_f32, int batch, int H, int W, int C, int Cout, void* stream) {
  API_BEGIN
  GemmParams p;
  p.dtype = dtype; p.A0 = a; p.a_f32 = 0; p.C0 = C; p.lda0 = C; p.taps = 9; p.H = H; p.W = W;
  p.Bw = w; p.ldb = 9L * C + CX0 + CX1; p.M = batch * H * W; p.N = Cout; p.bias_n = bias; p.rows_per_batch = H * W;
  p.alpha = alpha; p.C = out; p.c_f32 = c_f32; p.ldc = Cout;
  T2P_TRY(attach_op_ws(p));
  T2P_REQUIRE(gemm_can_fuse_shortcut(p), "this convolution does not take the shortcut segment (LDS-DMA 3x3 convolutions only)");
  p.X0 = x0; p.CX0 = CX0; p.ldx0 = CX0; p.X1 = x1; p.CX1 = CX1; p.ldx1 = CX1;
  return launch_gemm(p, (hipStream_t)stream);
  API_END
}

int t2p_op_conv3x3_groupnorm(int dtype, const void* a, const void* w, const float* bias, const float* bias_bn, const void* residual,
                             float alpha, int upsample, int groups, const float* gamma, const float* beta, float eps, int silu,
                             void* out, int out_f32, void* normed, float* col_stats, int batch, int H, int W, int C, int Cout, void* stream) {
  API_BEGIN
  GemmParams p;
  p.dtype = dtype; p.A0 = a; p.a_f32 = 0; p.C0 = C; p.lda0 = C; p.taps = 9; p.H = H; p.W = W; p.a_up = upsample;
  p.Bw = w; p.ldb = 9L * C; p.M = batch * H * W; p.N = Cout; p.bias_n = bias; p.rows_per_batch = H * W;
  p.bias_bn = bias_bn; p.ld_bn = Cout;
  p.R = (const float*)residual; p.r_lowp = residual ? 1 : 0; p.ldr = Cout;
  p.alpha = alpha; p.C = out; p.c_f32 = out_f32; p.ldc = Cout; p.col_stats = col_stats;
  T2P_TRY(attach_op_ws(p));
  T2P_REQUIRE(gemm_fuses_post_gn(p, groups), "this convolution does not take the split-K plan whose second pass applies a GroupNorm");
  p.gn_gamma = gamma; p.gn_beta = beta; p.gn_groups = groups; p.gn_silu = silu; p.gn_eps = eps; p.gn_out = normed;
  return launch_gemm(p, (hipStream_t)stream);
  API_END
}

int t2p_op_small_conv_groupnorm(int dtype, const void* a, const void* w, int64_t ldw, const void* x0, int CX0, const void* x1, int CX1,
                                const float* bias, const float* bias_bn, const void* residual, float alpha, void* out, int out_f32,
                                float* col_stats, void* normed, int groups, const float* gamma, const float* beta, float eps, int silu,
                                int batch, int H, int W, int C, int Cout, void* stream) {
  API_BEGIN
  SmallConvArgs s;
  s.dtype = dtype; s.A = a; s.B = batch; s.H = H; s.W = W; s.C = C; s.N = Cout; s.Wt = w; s.ldw = ldw;
  s.X0 = x0; s.CX0 = CX0; s.X1 = x1; s.CX1 = CX1; s.bias = bias; s.bias_bn = bias_bn; s.ld_bn = Cout; s.R = residual; s.alpha = alpha;
  s.out = out; s.out_f32 = out_f32; s.col_stats = col_stats; s.normed = normed; s.gn_gamma = gamma; s.gn_beta = beta; s.groups = groups;
  s.gn_silu = silu; s.gn_eps = eps;
  return launch_small_conv_gn(s, (hipStream_t)stream);
  API_END
}

int t2p_op_st_entry(int dtype, const void* x, const float* col_stats, int groups, const float* gn_gamma, const float* gn_beta, float gn_eps,
                    const void* w_in, const float* b_in, const void* residual, const float* ln_gamma, const float* ln_beta, float ln_eps,
                    const void* w_qkv, int n2, const float* b2, int geglu, void* t, void* qkv, const void* w3, const float* b3,
                    const void* res3, void* y, float* y_stats, int batch, int n, int C, void* stream) {
  API_BEGIN
  StEntryArgs e;
  e.dtype = dtype; e.B = batch; e.n = n; e.C = C; e.x = x; e.cstats = col_stats; e.groups = groups; e.gn_gamma = gn_gamma;
  e.gn_beta = gn_beta; e.gn_eps = gn_eps; e.w_in = w_in; e.b_in = b_in; e.ln_gamma = ln_gamma; e.ln_beta = ln_beta; e.ln_eps = ln_eps;
  e.w_qkv = w_qkv; e.n2 = n2; e.b2 = b2; e.geglu = geglu; e.res = residual; e.t = t; e.qkv = qkv;
  e.w3 = w3; e.b3 = b3; e.res3 = res3; e.y = y; e.y_stats = y_stats;
  T2P_REQUIRE(st_entry_eligible(e), "st_entry: C = 256 (or 512 without geglu), n2 in {C, 3 C} (8 C with geglu), 16-bit dtype, n % 32 == 0 (64 with column sums), batch n <= 8192");
  // the kernel reads fragment-major weights (the engine keeps such copies): made on the fly here, the call returns after the stream drained
  void* fm = nullptr;
  T2P_HIP_CHECK(hipMalloc(&fm, ((size_t)C * C + (size_t)n2 * C + (w3 ? (size_t)5 * C * C : 0)) * 2));
  void* fm2 = (char*)fm + (size_t)C * C * 2;
  void* fm3 = (char*)fm2 + (size_t)n2 * C * 2;
  int rc = launch_sf_frag_major(dtype, w_in, fm, C, C, (hipStream_t)stream);
  if (rc == T2P_OK) rc = launch_sf_frag_major(dtype, w_qkv, fm2, n2, C, (hipStream_t)stream);
  if (rc == T2P_OK && w3) rc = launch_sf_frag_major(dtype, w3, fm3, C, 5 * C, (hipStream_t)stream);
  if (rc == T2P_OK && w3 && y_stats && hipMemsetAsync(y_stats, 0, (size_t)batch * n / 64 * C * 2 * 4, (hipStream_t)stream) != hipSuccess) rc = T2P_ERR_HIP;
  e.w_in = fm; e.w_qkv = fm2;
  if (w3) e.w3 = fm3;
  if (rc == T2P_OK) rc = launch_st_entry(e, (hipStream_t)stream);
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(fm);
  return rc;
  API_END
}

int t2p_op_attn_proj(int dtype, const void* x, const float* col_stats, int groups, const float* gn_gamma, const float* gn_beta, float gn_eps,
                     const void* w_qk, const float* b_qk, const void* w_v, void* qk, void* vt, void* k_fm, int64_t npad, int batch, int n, int C,
                     void* stream) {
  API_BEGIN
  AttnProjArgs e;
  e.dtype = dtype; e.B = batch; e.n = n; e.C = C; e.npad = npad; e.x = x; e.cstats = col_stats; e.groups = groups; e.gn_gamma = gn_gamma;
  e.gn_beta = gn_beta; e.gn_eps = gn_eps; e.b_qk = b_qk; e.qk = qk; e.vt = vt; e.k_fm = k_fm;
  T2P_REQUIRE(attn_proj_eligible(e), "attn_proj: C = 256, 16-bit dtype, n % 32 == 0 (64 with column sums), batch n <= 8192, npad % 4 == 0");
  void* fm = nullptr;
  T2P_HIP_CHECK(hipMalloc(&fm, (size_t)3 * C * C * 2));
  void* fm2 = (char*)fm + (size_t)2 * C * C * 2;
  int rc = launch_sf_frag_major(dtype, w_qk, fm, 2 * C, C, (hipStream_t)stream);
  if (rc == T2P_OK) rc = launch_sf_frag_major(dtype, w_v, fm2, C, C, (hipStream_t)stream);
  e.w_qk = fm; e.w_v = fm2;
  if (rc == T2P_OK) rc = launch_attn_proj(e, (hipStream_t)stream);
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(fm);
  return rc;
  API_END
}

int t2p_op_input_conv(const float* x, const float* w_tcn, const float* bias, void* out, int out_dtype, int batch, int C, int H, int W,
                      int nf, float* col_stats, void* stream) {
  API_BEGIN
  if (pre_conv_split_ok(out_dtype, C, H, W, nf) && (!col_stats || pre_conv_fuses_col_stats(W, nf))) {
    // the form the engine runs in 16-bit modes: split weights prepared on the fly here (the engine keeps them)
    void* ws = nullptr;
    T2P_HIP_CHECK(hipMalloc(&ws, pre_conv_split_weight_bytes(C, nf)));
    int rc = launch_pre_conv_split_weights(w_tcn, ws, C, nf, (hipStream_t)stream);
    if (rc == T2P_OK) rc = launch_pre_conv_split(x, ws, bias, out, out_dtype, batch, C, H, W, nf, (hipStream_t)stream, col_stats);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(ws);
    return rc;
  }
  return launch_pre_conv(x, w_tcn, bias, out, out_dtype, batch, C, H, W, nf, (hipStream_t)stream, col_stats);
  API_END
}

int t2p_op_groupnorm(const float* x0, const float* x1, int C0, int C1, int batch, int H, int W, int groups,
                     const float* gamma, const float* beta, float eps, int silu, int down, void* out, int dtype,
                     void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  GroupNormArgs a;
  a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.B = batch; a.HW = H * W; a.G = groups; a.eps = eps;
  const int C = C0 + C1;
  {
    GroupNormApplyArgs g;
    g.x0 = x0; g.x1 = x1; g.C0 = C0; g.C1 = C1; g.B = batch; g.H = H; g.W = W; g.G = groups; g.gamma = gamma; g.beta = beta;
    g.silu = silu; g.down = down; g.out = out; g.dtype = dtype; g.eps = eps;
    if (g_gn_small && gn_small_eligible(g)) return launch_gn_small(g, s);
  }
  const int nparts = gn_num_chunks(a.HW) * ((C + 1023) / 1024);
  float* ws = nullptr;
  T2P_HIP_CHECK(hipMalloc((void**)&ws, ((size_t)batch * nparts * groups * 2 + (size_t)batch * groups * 2) * 4));
  a.partial = ws;
  a.stats = ws + (size_t)batch * nparts * groups * 2;
  int rc = launch_gn_stats(a, s);
  if (rc == T2P_OK) {
    GroupNormApplyArgs g;
    g.x0 = x0; g.x1 = x1; g.C0 = C0; g.C1 = C1; g.B = batch; g.H = H; g.W = W; g.G = groups; g.stats = a.stats;
    g.gamma = gamma; g.beta = beta; g.silu = silu; g.down = down; g.out = out; g.dtype = dtype;
    rc = launch_gn_apply(g, s);
  }
  (void)hipStreamSynchronize(s);
  (void)hipFree(ws);
  return rc;
  API_END
}

int t2p_op_layernorm(const float* x, const float* gamma, const float* beta, void* out, int dtype, int64_t rows, int C,
                     float eps, void* stream) {
  API_BEGIN
  return launch_layernorm(x, gamma, beta, out, dtype, rows, C, eps, (hipStream_t)stream);
  API_END
}

int t2p_op_layernorm16(const void* x16, const float* gamma, const float* beta, void* out16, int dtype, int64_t rows, int C,
                       float eps, void* stream) {
  API_BEGIN
  T2P_REQUIRE(dtype == DT_F16 || dtype == DT_BF16, "t2p_op_layernorm16 takes a 16-bit dtype");
  return launch_layernorm((const float*)x16, gamma, beta, out16, dtype, rows, C, eps, (hipStream_t)stream, 1);
  API_END
}

int t2p_op_softmax(const float* S, int64_t lds, void* P, int64_t ldp, int dtype, int64_t rows, int n, float scale,
                   void* stream) {
  API_BEGIN
  return launch_softmax(S, lds, P, ldp, dtype, rows, n, scale, (hipStream_t)stream);
  API_END
}

int t2p_op_geglu(const float* u, void* out, int dtype, int64_t rows, int inner, void* stream) {
  API_BEGIN
  return launch_geglu(u, out, dtype, rows, inner, (hipStream_t)stream);
  API_END
}

static inline int64_t rup8(int64_t v) { return (v + 7) / 8 * 8; }

int64_t t2p_op_attention_ws(int dtype, int batch, int heads, int nq, int nk) {
  const int64_t rows = (int64_t)batch * heads * nq;
  return rows * rup8(nk) * 4 + rows * rup8(nk) * (int64_t)dtype_size(dtype) + 512;
}

int t2p_op_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                     void* out, int batch, int heads, int nq, int nk, int d, float scale, void* workspace, void* stream) {
  API_BEGIN
  T2P_REQUIRE(workspace && ((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (g_flash_attention && attention_flash_eligible(dtype, d, ldq, ldk, ldvt, (long)heads * d))
    return launch_attention_flash(dtype, q, ldq, k, ldk, vt, ldvt, out, batch, heads, nq, nk, d, scale, s);
  if (attention_strip_eligible(dtype, heads, nq, nk, d, ldq, ldk, ldvt, d))
    return launch_attention_strip(dtype, q, ldq, k, ldk, vt, ldvt, out, d, batch, nq, d, scale, s);
  const long nkp = rup8(nk), rows = (long)batch * heads * nq;
  float* S = (float*)workspace;
  void* P = (char*)workspace + ((rows * nkp * 4 + 255) / 256) * 256;
  GemmParams p;
  p.dtype = dtype; p.a_f32 = dtype == DT_F32;
  p.A0 = q; p.C0 = d; p.lda0 = ldq; p.M = nq; p.N = nk; p.Bw = k; p.ldb = ldk;
  p.nz0 = batch; p.nz1 = heads;
  p.sA_z0 = (long)nq * ldq; p.sA_z1 = d; p.sB_z0 = (long)nk * ldk; p.sB_z1 = d;
  p.C = S; p.c_f32 = 1; p.ldc = nkp; p.sC_z0 = (long)heads * nq * nkp; p.sC_z1 = (long)nq * nkp;
  T2P_TRY(launch_gemm(p, s));
  T2P_TRY(launch_softmax(S, nkp, P, nkp, dtype, rows, nk, scale, s));
  GemmParams r;
  r.dtype = dtype; r.a_f32 = dtype == DT_F32;
  r.A0 = P; r.C0 = nk; r.lda0 = nkp; r.M = nq; r.N = d; r.Bw = vt; r.ldb = ldvt;
  r.nz0 = batch; r.nz1 = heads;
  r.sA_z0 = (long)heads * nq * nkp; r.sA_z1 = (long)nq * nkp; r.sB_z0 = (long)heads * d * ldvt; r.sB_z1 = (long)d * ldvt;
  r.C = out; r.c_f32 = 0; r.ldc = (long)heads * d; r.sC_z0 = (long)nq * heads * d; r.sC_z1 = d;
  return launch_gemm(r, s);
  API_END
}

int t2p_op_attention_wide(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* out,
                          int out_f32, const float* bias, const void* residual, int residual_16bit, float alpha, float* col_stats,
                          int batch, int n, int d, float scale, void* stream) {
  API_BEGIN
  T2P_REQUIRE(attention_strip_eligible(dtype, 1, n, n, d, ldq, ldk, ldvt, d), "attention_wide: 16-bit dtypes, d = 256 / 512 / 1024, n <= 1024 (512 at d = 1024), n % 8 == 0");
  StripEpilogue ep;
  ep.bias = bias; ep.residual = residual; ep.r_lowp = residual_16bit; ep.ldr = d; ep.alpha = alpha; ep.out_f32 = out_f32; ep.col_stats = col_stats;
  return launch_attention_strip(dtype, q, ldq, k, ldk, vt, ldvt, out, d, batch, n, d, scale, (hipStream_t)stream, &ep);
  API_END
}

int t2p_op_attention_wide_fm(int dtype, const void* q, int64_t ldq, const void* k_fm, const void* vt_fm, void* out, int out_f32,
                             const float* bias, const void* residual, int residual_16bit, float alpha, float* col_stats, int batch, int n,
                             int d, float scale, void* stream) {
  API_BEGIN
  T2P_REQUIRE(attention_strip_frag_major_ok(dtype, n, d) && attention_strip_eligible(dtype, 1, n, n, d, ldq, d, n, d),
              "attention_wide_fm: 16-bit dtypes, d = 512 with 512 < n <= 1024 or d = 256 with n <= 256, n % 32 == 0");
  StripEpilogue ep;
  ep.bias = bias; ep.residual = residual; ep.r_lowp = residual_16bit; ep.ldr = d; ep.alpha = alpha; ep.out_f32 = out_f32; ep.col_stats = col_stats;
  ep.frag_major = 1;
  return launch_attention_strip(dtype, q, ldq, k_fm, d, vt_fm, n, out, d, batch, n, d, scale, (hipStream_t)stream, &ep);
  API_END
}

int t2p_op_gemm_frag_major(int dtype, const void* A, const void* Bw, void* C, void* c_frag, int M, int N, int K, int frag_col0,
                           int rows_per_batch, int batch, void* stream) {
  API_BEGIN
  GemmParams p;
  p.dtype = dtype; p.A0 = A; p.a_f32 = 0; p.C0 = K; p.lda0 = K; p.Bw = Bw; p.ldb = K; p.M = M; p.N = N;
  p.C = C; p.c_f32 = 0; p.ldc = N;
  if (batch > 1) { p.nz0 = batch; p.sA_z0 = 0; p.sB_z0 = (long)N * K; p.sC_z0 = (long)M * N; p.frag_bstride = (long)M * (N - frag_col0); }
  else { p.rows_per_batch = rows_per_batch; p.frag_bstride = (long)rows_per_batch * (N - frag_col0); }
  p.c_frag = c_frag; p.frag_col0 = frag_col0; p.frag_ns = (N - frag_col0) / 32;
  T2P_TRY(attach_op_ws(p));
  T2P_REQUIRE(gemm_writes_frag_major(p), "gemm_frag_major: the product does not take the fragment-major output (256 x 256 plan, whole fragments)");
  return launch_gemm(p, (hipStream_t)stream);
  API_END
}

int t2p_op_attention_qkv(int dtype, const void* qkv, int64_t ld, void* out, int batch, int heads, int n, int d, float scale, void* stream) {
  API_BEGIN
  const long C = (long)heads * d;
  T2P_REQUIRE(qkv && out && ld >= 3 * C, "attention_qkv arguments");
  T2P_REQUIRE(attention_flash_eligible(dtype, d, ld, ld, ld, C), "attention_qkv: 16-bit dtypes, head dimension 32 / 64 / 128, ld % 8 == 0");
  const size_t es = dtype_size(dtype);
  return launch_attention_flash(dtype, qkv, ld, (const char*)qkv + C * es, ld, (const char*)qkv + 2 * C * es, ld, out, batch, heads, n, n, d,
                                scale, (hipStream_t)stream, true);
  API_END
}

int t2p_op_langevin(const float* x, const float* grad, const float* noise, const uint8_t* mask, const float* x_initial,
                    float* x_out, float* x_mean_out, int batch, int64_t per_sample, float snr, float alpha,
                    float* sums_out, void* stream) {
  API_BEGIN
  hipStream_t s = (hipStream_t)stream;
  float* ws = nullptr;
  T2P_HIP_CHECK(hipMalloc((void**)&ws, ((size_t)batch * 64 * 2 + 64) * 4));
  float* sums = sums_out ? sums_out : ws + (size_t)batch * 64 * 2;
  int rc = launch_langevin_norms(grad, noise, batch, per_sample, ws, sums, s);
  if (rc == T2P_OK) {
    SdeUpdateArgs a;
    a.x = x; a.score = grad; a.noise = noise; a.mask = mask; a.x_initial = x_initial; a.x_out = x_out;
    a.x_mean_out = x_mean_out; a.n = per_sample * batch;
    rc = launch_langevin_update(a, sums, (float)batch, snr, alpha, s);
  }
  (void)hipStreamSynchronize(s);
  (void)hipFree(ws);
  return rc;
  API_END
}

int t2p_op_langevin_norms(const float* grad, const float* noise, int batch, int64_t per_sample, float* workspace,
                          float* sums, void* stream) {
  API_BEGIN
  return launch_langevin_norms(grad, noise, batch, per_sample, workspace, sums, (hipStream_t)stream);
  API_END
}

int t2p_op_langevin_update(const float* x, const float* grad, const float* noise, const uint8_t* mask,
                           const float* x_initial, float* x_out, float* x_mean_out, int64_t n, const float* sums,
                           float batch_total, float snr, float alpha, void* stream) {
  API_BEGIN
  SdeUpdateArgs a;
  a.x = x; a.score = grad; a.noise = noise; a.mask = mask; a.x_initial = x_initial; a.x_out = x_out;
  a.x_mean_out = x_mean_out; a.n = n;
  return launch_langevin_update(a, sums, batch_total, snr, alpha, (hipStream_t)stream);
  API_END
}

int t2p_op_predictor(const float* x, const float* score, const float* noise, const uint8_t* mask, const float* x_initial,
                     float* x_out, float* x_mean_out, int64_t n, float G, int probability_flow, void* stream) {
  API_BEGIN
  SdeUpdateArgs a;
  a.x = x; a.score = score; a.noise = noise; a.mask = mask; a.x_initial = x_initial; a.x_out = x_out;
  a.x_mean_out = x_mean_out; a.n = n;
  return launch_predictor_update(a, nullptr, nullptr, G, probability_flow, (hipStream_t)stream);
  API_END
}

int t2p_op_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  API_BEGIN
  return launch_philox_normal(out, n, seed, stream_id, nullptr, (hipStream_t)stream);
  API_END
}

int t2p_op_convert(const float* in, void* out, int dtype, int64_t n, void* stream) {
  API_BEGIN
  return launch_convert(in, out, dtype, n, (hipStream_t)stream);
  API_END
}

int t2p_op_decode_6d(const float* x, int batch, int channels, int L, float* clipped, float* absval, int32_t* lengths, void* stream) {
  API_BEGIN
  return launch_decode6d(x, batch, channels, L, clipped, absval, lengths, (hipStream_t)stream);
  API_END
}

int t2p_op_embedding_gather(const void* table, int table_dtype, const int32_t* ids, float* out, int64_t n_tokens, int dim, int vocab,
                            int32_t* bad_flag, void* stream) {
  API_BEGIN
  return launch_embedding_gather(table, table_dtype, ids, out, n_tokens, dim, vocab, bad_flag, (hipStream_t)stream);
  API_END
}

int t2p_op_apply_mask(float* x, const uint8_t* mask, const float* x_initial, int64_t n, void* stream) {
  API_BEGIN
  return launch_apply_mask(x, mask, x_initial, n, (hipStream_t)stream);
  API_END
}

int t2p_debug_set(int key, int value) {
  if (key == 10) { g_op_ws_bytes = (size_t)value << 20; return T2P_OK; }
  if (key == 11) { set_gemm_force_nsplit(value); return T2P_OK; }
  if (key == 12) { set_gemm_midsplit(value != 0); return T2P_OK; }
  if (key == 13) { g_gn_small = value != 0; return T2P_OK; }
  if (key == 14) { g_lowp_residual = value != 0; return T2P_OK; }
  if (key == 15) { set_gemm_thin_conv(value != 0); return T2P_OK; }
  if (key == 17) { g_gn_apply16 = value != 0; return T2P_OK; }
  if (key == 20) { g_layernorm16 = value != 0; return T2P_OK; }
  if (key == 21) { set_gemm_up4(value != 0); return T2P_OK; }
  if (key == 22) { set_gemm_deep_ring(value != 0); return T2P_OK; }
  if (key == 23) { set_gemm_fuse_shortcut(value != 0); return T2P_OK; }
  if (key == 25) { g_qkv_fused = value != 0; return T2P_OK; }
  if (key == 26) { t2p::g_pre_conv_mfma = value != 0; return T2P_OK; }
  if (key == 27) { t2p::g_gn_apply_cols = value != 0; return T2P_OK; }
  if (key == 28) { set_gemm_post_gn(value != 0); return T2P_OK; }
  if (key == 29) { t2p::g_attn_strip = value != 0; return T2P_OK; }
  if (key == 36) { t2p::g_small_conv = value != 0; return T2P_OK; }
  if (key == 38) { t2p::g_pre_conv_split = value != 0; return T2P_OK; }
  if (key == 39) { t2p::g_st_fuse = value != 0; return T2P_OK; }
  if (key == 40) { t2p::g_st_tail = value != 0; return T2P_OK; }
  if (key == 41) { t2p::g_small_conv_fm = value != 0; return T2P_OK; }
  if (key == 42) { t2p::g_st_ffpo = value != 0; return T2P_OK; }
  if (key == 43) { t2p::g_st_tail_rows = value; return T2P_OK; }
  if (key == 45) { t2p::g_attn_fm = value != 0; return T2P_OK; }
  if (key == 46) { t2p::g_attn_proj = value != 0; return T2P_OK; }
  if (key == 47) { set_gemm_dxs(value != 0); return T2P_OK; }
  if (key == 34) { set_gemm_a_norm(value != 0); return T2P_OK; }
  if (key == 32) { g_attn_merged = value != 0; return T2P_OK; }
  if (key == 33) { g_ffpo_merged = value != 0; return T2P_OK; }
  if (key == 30) { set_gemm_split_consts(value, 0); return T2P_OK; }
  if (key == 31) { set_gemm_split_consts(0, value); return T2P_OK; }
  if (key == 0) set_gemm_dma(value != 0);
  else if (key == 1) {
#ifndef T2P_ABLATION
    if (value & ~(128 | 256 | 4096 | 8192)) {        // (8192: the general register epilogue where the specialised one would run -- same results)
      set_last_error("ablation bits that skip work exist only in a -DT2P_ABLATION build");
      return T2P_ERR_INVALID;
    }
#endif
    set_gemm_debug(value);
  }
  else if (key == 2) set_gemm_geom(value);
  else if (key == 3) set_gemm_splitk(value != 0);
  else if (key == 4) g_raw_copies = value != 0;
  else if (key == 5) g_flash_attention = value != 0;
  else if (key == 6) g_fuse_gn_stats = value != 0;
  else if (key == 7) g_fuse_geglu = value != 0;
  else if (key == 8) set_gemm_ring(value);
  else if (key == 9) g_lowp_h1 = value != 0;
  else return T2P_ERR_INVALID;
  return T2P_OK;
}

#ifdef T2P_ABLATION
extern "C" int t2p_ablation_dxs_stamps(unsigned long long* out, int n) { return t2p::dxs_stamps_read(out, n); }
#endif

int t2p_built_with_ablation(void) {
#ifdef T2P_ABLATION
  return 1;
#else
  return 0;
#endif
}

int t2p_profile_begin(void) {
  profile_begin();
  return T2P_OK;
}

int t2p_profile_end(double* out9) {
  API_BEGIN
  T2P_REQUIRE(out9, "null argument");
  double o[3][3];
  T2P_TRY(profile_end(o));
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < 3; ++j) out9[k * 3 + j] = o[k][j];
  return T2P_OK;
  API_END
}

int t2p_profile_attention(double* out3) {
  API_BEGIN
  T2P_REQUIRE(out3, "null argument");
  return profile_attention(out3);
  API_END
}

int t2p_profile_shapes(char* buf, int len) {
  API_BEGIN
  T2P_REQUIRE(buf && len > 0, "null argument");
  return profile_shapes(buf, len);
  API_END
}

int t2p_debug_tap(int block_index, float* dst, int64_t capacity, int64_t* shape4) {
  API_BEGIN
  if (shape4) {
    long sh[4];
    t2p::debug_tap_shape(sh);
    for (int i = 0; i < 4; ++i) shape4[i] = sh[i];
  }
  t2p::debug_tap_set(block_index, dst, (long)capacity);
  return T2P_OK;
  API_END
}

int t2p_profile_layers_begin(void) {
  t2p::layer_profile_begin();
  return T2P_OK;
}

int t2p_profile_layers_end(char* buf, int len) {
  API_BEGIN
  T2P_REQUIRE(buf && len > 0, "null argument");
  std::string out;
  T2P_TRY(t2p::layer_profile_end(&out));
  T2P_REQUIRE((size_t)len > out.size(), "the per-block timing table does not fit the buffer (" + std::to_string(out.size() + 1) + " bytes needed)");
  std::snprintf(buf, (size_t)len, "%s", out.c_str());
  return T2P_OK;
  API_END
}

int t2p_profile_dominant(double* out4, char* name, int name_len) {
  API_BEGIN
  T2P_REQUIRE(out4 && name && name_len > 0, "null argument");
  const char* n = nullptr;
  T2P_TRY(profile_dominant(out4, &n));
  std::snprintf(name, (size_t)name_len, "%s", n ? n : "");
  return T2P_OK;
  API_END
}

}  // extern "C"
