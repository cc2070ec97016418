// Training step of the score network on MI355X, fp32 (first slice).
//
// What the reference does in one step (score_sde_pytorch/losses.py:165-176): optimizer.zero_grad(); loss = loss_fn(...) (:105-134);
// loss.backward(); optimize_fn (:41-49: warm-up on state['step'], clip_grad_norm_, Adam); state['step'] += 1; ema.update
// (models/ema.py:32-49).  Here: the parameters, their gradients, both Adam moments and the EMA shadow are five flat device buffers
// in the reference's parameters() order (what the checkpoint loader and the optimizer kernels want); the forward pass walks the
// same block list as the sampling engine (Engine::build), keeps the activations the backward pass needs and records one
// closure per operator; the backward pass runs the closures in reverse.  Products: 3x3 convolutions forward and input-gradient on
// the engine's exact-f32 implicit-GEMM kernel (the input gradient is the same convolution on dY with flipped, transposed taps),
// everything else -- linear layers both ways, weight gradients, the attention products -- on the strided GEMM of
// train_kernels.hip.  Every gradient buffer is zero-initialised at first use and accumulated into, so fan-out (residual
// connections, U-Net skips, the shared time embedding) needs no special cases.
#include "train.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace t2p {

static int gn_groups_of(int c) { return std::min(c / 4, 32); }   // layers.py:282

Trainer::Trainer(const t2p_model_config& mc, const t2p_train_config& tc) : mc_(mc), tc_(tc), arch_(mc) {}
Trainer::~Trainer() {}

long Trainer::poff(const std::string& name, std::vector<int64_t> shape) {
  auto it = index_.find(name);
  if (it == index_.end()) { set_last_error("trainer: no parameter " + name); return -1; }
  const TParam& p = params_[it->second];
  if (p.shape != shape) { set_last_error("trainer: unexpected shape of " + name); return -1; }
  return p.off;
}

#define T2P_OFF(dst, name, ...)                                     \
  do {                                                              \
    (dst) = poff((name), std::vector<int64_t>{__VA_ARGS__});        \
    if ((dst) < 0) return T2P_ERR_STATE;                            \
  } while (0)

int Trainer::map_layer(const Layer& l, LayerT* o) {
  o->kind = l.kind; o->in_ch = l.in_ch; o->out_ch = l.out_ch; o->up = l.up; o->down = l.down;
  const std::string& p = l.prefix;
  const int64_t ci = l.in_ch, co = l.out_ch, td = 4 * mc_.nf;
  auto norm = [&](Norm& n, const std::string& pre, int C, int G) -> int {
    T2P_OFF(n.g, pre + ".weight", C); T2P_OFF(n.b, pre + ".bias", C);
    n.C = C; n.G = G;
    return T2P_OK;
  };
  auto lin = [&](Lin& q, const std::string& w, const std::string& b, int64_t N, int64_t K, int form) -> int {   // 0 Linear, 1 conv1x1, 2 NIN
    if (form == 0) T2P_OFF(q.w, w, N, K);
    if (form == 1) T2P_OFF(q.w, w, N, K, 1, 1);
    if (form == 2) T2P_OFF(q.w, w, K, N);
    q.b = -1;
    if (!b.empty()) T2P_OFF(q.b, b, N);
    q.N = (int)N; q.K = (int)K; q.nin = form == 2;
    return T2P_OK;
  };
  auto conv = [&](Conv& c, const std::string& pre, int64_t Co, int64_t Ci) -> int {
    T2P_OFF(c.w, pre + ".weight", Co, Ci, 3, 3); T2P_OFF(c.b, pre + ".bias", Co);
    c.Co = (int)Co; c.Ci = (int)Ci; c.Cip = (int)((Ci + 7) / 8 * 8); c.Cop = (int)((Co + 7) / 8 * 8);
    return T2P_OK;
  };
  if (l.kind == 0) {
    ResL& r = o->r;
    T2P_TRY(norm(r.gn0, p + ".GroupNorm_0", (int)ci, gn_groups_of((int)ci)));
    T2P_TRY(conv(r.c0, p + ".Conv_0", co, ci));
    T2P_TRY(lin(r.dense, p + ".Dense_0.weight", p + ".Dense_0.bias", co, td, 0));
    T2P_TRY(norm(r.gn1, p + ".GroupNorm_1", (int)co, gn_groups_of((int)co)));
    T2P_TRY(conv(r.c1, p + ".Conv_1", co, co));
    r.has_sc = l.has_conv2;
    if (r.has_sc) T2P_TRY(lin(r.sc, p + ".Conv_2.weight", p + ".Conv_2.bias", co, ci, 1));
  } else if (l.kind == 1) {
    AttnL& a = o->a;
    T2P_TRY(norm(a.gn, p + ".GroupNorm_0", (int)ci, gn_groups_of((int)ci)));
    for (int i = 0; i < 4; ++i) T2P_TRY(lin(a.nin[i], p + ".NIN_" + std::to_string(i) + ".W", p + ".NIN_" + std::to_string(i) + ".b", ci, ci, 2));
  } else {
    StL& s = o->st;
    const std::string t = p + ".transformer_blocks.0";
    const int64_t c = ci, ctx = mc_.context_dim;
    T2P_TRY(norm(s.gn, p + ".norm", (int)c, 32));                         // attention.py:77: 32 groups
    T2P_TRY(lin(s.proj_in, p + ".proj_in.weight", p + ".proj_in.bias", c, c, 1));
    T2P_TRY(lin(s.q1, t + ".attn1.to_q.weight", "", c, c, 0)); T2P_TRY(lin(s.k1, t + ".attn1.to_k.weight", "", c, c, 0));
    T2P_TRY(lin(s.v1, t + ".attn1.to_v.weight", "", c, c, 0)); T2P_TRY(lin(s.o1, t + ".attn1.to_out.0.weight", t + ".attn1.to_out.0.bias", c, c, 0));
    T2P_TRY(lin(s.ff1, t + ".ff.net.0.proj.weight", t + ".ff.net.0.proj.bias", 8 * c, c, 0));
    T2P_TRY(lin(s.ff2, t + ".ff.net.2.weight", t + ".ff.net.2.bias", c, 4 * c, 0));
    T2P_TRY(lin(s.q2, t + ".attn2.to_q.weight", "", c, c, 0)); T2P_TRY(lin(s.k2, t + ".attn2.to_k.weight", "", c, ctx, 0));
    T2P_TRY(lin(s.v2, t + ".attn2.to_v.weight", "", c, ctx, 0)); T2P_TRY(lin(s.o2, t + ".attn2.to_out.0.weight", t + ".attn2.to_out.0.bias", c, c, 0));
    for (int i = 0; i < 3; ++i) T2P_TRY(norm(s.ln[i], t + ".norm" + std::to_string(i + 1), (int)c, 1));
    T2P_TRY(lin(s.proj_out, p + ".proj_out.weight", p + ".proj_out.bias", c, c, 1));
  }
  return T2P_OK;
}

int Trainer::build() {
  T2P_REQUIRE(mc_.compute_dtype == DT_F32, "the training step is fp32 only (first slice of SURVEY.md 8(f)4)");
  T2P_REQUIRE(tc_.dropout >= 0.0 && tc_.dropout < 1.0 && tc_.ema_rate >= 0.0 && tc_.ema_rate <= 1.0, "dropout / ema_rate");
  int dev = 0;
  T2P_HIP_CHECK(hipGetDevice(&dev));       // fails without a HIP device: there is no CPU fallback
  T2P_TRY(arch_.build());
  long off = 0;
  for (const ParamInfo& p : arch_.params()) {
    TParam t;
    t.name = p.name; t.shape = p.shape; t.off = off; t.n = 1;
    for (int64_t d : p.shape) t.n *= d;
    off += t.n;
    index_[t.name] = (int)params_.size();
    params_.push_back(std::move(t));
  }
  total_ = off;
  const size_t bytes = (size_t)total_ * 4;
  P_ = (float*)pool_.persistent(bytes); Gr_ = (float*)pool_.persistent(bytes);
  M_ = (float*)pool_.persistent(bytes); V_ = (float*)pool_.persistent(bytes); E_ = (float*)pool_.persistent(bytes);
  sumsq_ = (double*)pool_.persistent(8); loss_dev_ = (float*)pool_.persistent(4);
  if (!P_ || !Gr_ || !M_ || !V_ || !E_ || !sumsq_ || !loss_dev_) return T2P_ERR_HIP;
  for (float* b : {P_, Gr_, M_, V_, E_}) T2P_HIP_CHECK(hipMemset(b, 0, bytes));

  const int64_t td = 4 * mc_.nf, nf = mc_.nf, ch = mc_.num_channels;
  auto top_lin = [&](Lin& q, const std::string& pre, int64_t N, int64_t K) -> int {
    T2P_OFF(q.w, pre + ".weight", N, K); T2P_OFF(q.b, pre + ".bias", N);
    q.N = (int)N; q.K = (int)K; q.nin = false;
    return T2P_OK;
  };
  T2P_TRY(top_lin(pre0_, "pre_blocks.0", td, nf));
  T2P_TRY(top_lin(pre1_, "pre_blocks.1", td, td));
  T2P_OFF(pre_conv_.w, "pre_conv.weight", nf, ch, 3, 3); T2P_OFF(pre_conv_.b, "pre_conv.bias", nf);
  pre_conv_.Co = (int)nf; pre_conv_.Ci = (int)ch; pre_conv_.Cip = 8; pre_conv_.Cop = (int)nf;
  auto map_stage = [&](const Stage& st, std::vector<LayerT>* out) -> int {
    for (const Layer& l : st.layers) {
      LayerT t;
      T2P_TRY(map_layer(l, &t));
      out->push_back(std::move(t));
    }
    return T2P_OK;
  };
  for (const Stage& st : arch_.input_stages_) { in_stages_.emplace_back(); T2P_TRY(map_stage(st, &in_stages_.back())); }
  T2P_TRY(map_stage(arch_.mid_stage_, &mid_));
  for (const Stage& st : arch_.out_stages_) { out_stages_.emplace_back(); T2P_TRY(map_stage(st, &out_stages_.back())); }
  const int fc = arch_.final_ch_;
  T2P_OFF(head_norm_.g, "out.0.weight", fc); T2P_OFF(head_norm_.b, "out.0.bias", fc);
  head_norm_.C = fc; head_norm_.G = gn_groups_of(fc);
  T2P_OFF(head_conv_.w, "out.2.weight", ch, fc, 3, 3); T2P_OFF(head_conv_.b, "out.2.bias", ch);
  head_conv_.Co = (int)ch; head_conv_.Ci = fc; head_conv_.Cip = fc; head_conv_.Cop = 8;

  // kernel-layout copies of the 3x3 convolution weights, refreshed from the flat parameters at the start of every pass
  convs_.push_back(&pre_conv_);
  auto collect = [&](std::vector<LayerT>& ls) {
    for (LayerT& l : ls)
      if (l.kind == 0) { convs_.push_back(&l.r.c0); convs_.push_back(&l.r.c1); }
  };
  for (auto& st : in_stages_) collect(st);
  collect(mid_);
  for (auto& st : out_stages_) collect(st);
  convs_.push_back(&head_conv_);
  for (Conv* c : convs_) {
    T2P_REQUIRE(c->Cip % 4 == 0 && c->Cop % 4 == 0, "convolution channel padding");
    c->wf = (float*)pool_.persistent((size_t)c->Co * 9 * c->Cip * 4);
    if (!c->wf) return T2P_ERR_HIP;
    if (c != &pre_conv_) {                      // the network input needs no gradient
      c->wd = (float*)pool_.persistent((size_t)c->Ci * 9 * c->Cop * 4);
      if (!c->wd) return T2P_ERR_HIP;
    }
    dwc_floats_ = std::max(dwc_floats_, (size_t)c->Co * 9 * c->Cip);
  }
  dwc_ = (float*)pool_.persistent(dwc_floats_ * 4);
  if (!dwc_) return T2P_ERR_HIP;

  std::vector<float> inv(mc_.num_scales);        // 1 / sigmas[label], sigmas descending (models/utils.py:50-60, ncsnpp.py:256-261)
  const double a = std::log(mc_.sigma_max), b = std::log(mc_.sigma_min);
  for (int i = 0; i < mc_.num_scales; ++i) inv[i] = (float)(1.0 / std::exp(a + (b - a) * (double)i / (double)(mc_.num_scales - 1)));
  inv_sigma_ = (float*)pool_.persistent(inv.size() * 4);
  if (!inv_sigma_) return T2P_ERR_HIP;
  T2P_HIP_CHECK(hipMemcpy(inv_sigma_, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
  return T2P_OK;
}

int Trainer::load_param(const char* name, const float* host, const int64_t* shape, int ndim) {
  T2P_REQUIRE(name && host && shape && ndim >= 1 && ndim <= 4, "load_param arguments");
  std::string n(name);
  if (n.rfind("module.", 0) == 0) n = n.substr(7);
  if (n == "sigmas") return T2P_OK;
  auto it = index_.find(n);
  if (it == index_.end()) return T2P_OK;            // load_state_dict(strict=False)
  const TParam& p = params_[it->second];
  T2P_REQUIRE(std::vector<int64_t>(shape, shape + ndim) == p.shape, "shape mismatch for " + n);
  T2P_HIP_CHECK(hipMemcpy(P_ + p.off, host, (size_t)p.n * 4, hipMemcpyHostToDevice));
  T2P_HIP_CHECK(hipMemcpy(E_ + p.off, host, (size_t)p.n * 4, hipMemcpyHostToDevice));   // ema.py:28-29: shadow = clone of the parameters
  return T2P_OK;
}

int Trainer::read_tensor(int which, const char* name, float* host_out) {
  T2P_REQUIRE(name && host_out && which >= 0 && which <= 4, "read_tensor arguments");
  auto it = index_.find(name);
  T2P_REQUIRE(it != index_.end(), std::string("no parameter ") + name);
  const TParam& p = params_[it->second];
  float* src[5] = {P_, Gr_, E_, M_, V_};
  T2P_HIP_CHECK(hipDeviceSynchronize());
  T2P_HIP_CHECK(hipMemcpy(host_out, src[which] + p.off, (size_t)p.n * 4, hipMemcpyDeviceToHost));
  return T2P_OK;
}
int Trainer::write_tensor(int which, const char* name, const float* host_in) {
  T2P_REQUIRE(name && host_in && which >= 0 && which <= 4, "write_tensor arguments");
  auto it = index_.find(name);
  T2P_REQUIRE(it != index_.end(), std::string("no parameter ") + name);
  const TParam& p = params_[it->second];
  float* dst[5] = {P_, Gr_, E_, M_, V_};
  T2P_HIP_CHECK(hipDeviceSynchronize());
  T2P_HIP_CHECK(hipMemcpy(dst[which] + p.off, host_in, (size_t)p.n * 4, hipMemcpyHostToDevice));
  return T2P_OK;
}
int Trainer::set_step(int64_t step, int64_t adam_updates, int64_t ema_updates) {
  T2P_REQUIRE(step >= 0 && adam_updates >= 0 && ema_updates >= 0, "step counters");
  step_ = step; adam_k_ = adam_updates; ema_k_ = ema_updates;
  return T2P_OK;
}
int Trainer::get_step(int64_t out[3]) const { out[0] = step_; out[1] = adam_k_; out[2] = ema_k_; return T2P_OK; }
int Trainer::set_dropout_masks(const uint8_t* const* masks, int n) {
  T2P_REQUIRE(n == 0 || masks, "dropout masks");
  drop_masks_.assign(masks, masks + n);
  return T2P_OK;
}

// ---- pass-local memory ------------------------------------------------------------------------------------------------------------------
float* Trainer::tmp(size_t bytes) {
  void* p = pool_.get(bytes);
  if (p) live_.push_back(p);
  return (float*)p;
}
TT* Trainer::act(int B, int H, int W, int C, bool needs_grad) {
  acts_.emplace_back();
  TT* t = &acts_.back();
  t->B = B; t->H = H; t->W = W; t->C = C; t->needs_grad = needs_grad;
  t->p = tmp((size_t)t->numel() * 4);
  return t->p ? t : nullptr;
}
float* Trainer::grad(TT* t) {
  if (!t->g) {
    t->g = tmp((size_t)t->numel() * 4);
    if (t->g && hipMemsetAsync(t->g, 0, (size_t)t->numel() * 4, s_) != hipSuccess) t->g = nullptr;
  }
  return t->g;
}
void Trainer::release() {
  for (void* p : live_) pool_.put(p);
  live_.clear();
  acts_.clear();
  tape_.clear();
}

#define T2P_ACT(var, ...)                    \
  TT* var = act(__VA_ARGS__);                \
  if (!var) return T2P_ERR_HIP;
#define T2P_GRAD(var, t)                     \
  float* var = grad(t);                      \
  if (!var) return T2P_ERR_HIP;

int Trainer::prep_weights(const float* P, hipStream_t s) {
  for (Conv* c : convs_) T2P_TRY(launch_conv_w_prep(P + c->w, c->wf, c->wd, c->Co, c->Ci, c->Cip, c->Cop, s));
  return T2P_OK;
}

// ---- operators -----------------------------------------------------------------------------------------------------------------------------
// y [B][H W][ldc] = conv3x3(x) + bias (+ bias_bn[b][:]: Dense_0(act(temb)) of the block, layers.py:316) (+ residual_inplace, which may be y:
// the register-staged kernel reads and writes an output element in the same thread) on the engine's exact-f32 implicit-GEMM kernel
static int conv_forward(const float* x, int B, int H, int W, int Cin, const float* w, long ldb, const float* bias, const float* bias_bn, int N,
                        float* y, long ldc, const float* residual_inplace, hipStream_t s) {
  GemmParams p;
  p.dtype = DT_F32; p.a_f32 = 1; p.A0 = x; p.C0 = Cin; p.lda0 = Cin; p.taps = 9; p.H = H; p.W = W;
  p.Bw = w; p.ldb = ldb; p.M = B * H * W; p.N = N; p.bias_n = bias; p.bias_bn = bias_bn; p.rows_per_batch = H * W; p.ld_bn = N;
  p.R = residual_inplace; p.ldr = ldc;
  p.C = y; p.c_f32 = 1; p.ldc = ldc;
  return launch_gemm(p, s);
}

int Trainer::linear(TT* x, const Lin& l, TT** out) {
  T2P_REQUIRE(x->C == l.K, "linear: input width");
  T2P_ACT(y, x->B, x->H, x->W, l.N);
  const long rows = x->rows();
  T2P_REQUIRE(rows < (1L << 31), "linear: rows");
  TGemmArgs a;
  a.A = x->p; a.sAm = l.K; a.sAk = 1;
  a.B = Pc_ + l.w; a.sBk = l.nin ? l.N : 1; a.sBn = l.nin ? 1 : l.K;
  a.C = y->p; a.ldc = l.N; a.M = (int)rows; a.N = l.N; a.K = l.K;
  a.bias_n = l.b >= 0 ? Pc_ + l.b : nullptr;
  T2P_TRY(launch_tgemm(a, s_));
  *out = y;
  const Lin L = l;
  tape_.push_back([this, x, y, L, rows]() -> int {
    if (!y->g) return T2P_OK;
    if (x->needs_grad) {                         // dx += dy W
      T2P_GRAD(gx, x);
      TGemmArgs d;
      d.A = y->g; d.sAm = L.N; d.sAk = 1;
      d.B = Pc_ + L.w; d.sBk = L.nin ? 1 : L.K; d.sBn = L.nin ? L.N : 1;       // B(k = n', n = k') = W[n'][k'] (Linear) / W[k'][n'] (NIN)
      d.C = gx; d.ldc = L.K; d.M = (int)rows; d.N = L.K; d.K = L.N; d.beta = 1.f;
      T2P_TRY(launch_tgemm(d, s_));
    }
    TGemmArgs w;                                  // dW += dy^T x (Linear [N][K]) / x^T dy (NIN [K][N]); K of this product = the rows
    if (!L.nin) { w.A = y->g; w.sAm = 1; w.sAk = L.N; w.B = x->p; w.sBk = L.K; w.sBn = 1; w.M = L.N; w.N = L.K; }
    else        { w.A = x->p; w.sAm = 1; w.sAk = L.K; w.B = y->g; w.sBk = L.N; w.sBn = 1; w.M = L.K; w.N = L.N; }
    w.C = Gr_ + L.w; w.ldc = w.N; w.K = (int)rows; w.beta = 1.f; w.ksplit = 0;
    T2P_TRY(launch_tgemm(w, s_));
    if (L.b >= 0) T2P_TRY(launch_colsum(y->g, rows, L.N, L.N, Gr_ + L.b, s_));
    return T2P_OK;
  });
  return T2P_OK;
}

int Trainer::group_norm(TT* x, const Norm& n, int silu, TT** out) {
  T2P_REQUIRE(x->C == n.C, "GroupNorm channels");
  T2P_ACT(y, x->B, x->H, x->W, x->C);
  const int B = x->B, HW = x->H * x->W;
  float* stats = tmp((size_t)B * n.G * 2 * 4);
  const int nparts = gn_num_chunks(HW) * ((n.C + 1023) / 1024);
  float* partial = tmp((size_t)B * nparts * n.G * 2 * 4);
  if (!stats || !partial) return T2P_ERR_HIP;
  GroupNormArgs a;
  a.x0 = x->p; a.C0 = n.C; a.B = B; a.HW = HW; a.G = n.G; a.eps = 1e-6f; a.partial = partial; a.stats = stats;
  T2P_TRY(launch_gn_stats(a, s_));
  GroupNormApplyArgs g;
  g.x0 = x->p; g.C0 = n.C; g.B = B; g.H = x->H; g.W = x->W; g.G = n.G; g.stats = stats; g.gamma = Pc_ + n.g; g.beta = Pc_ + n.b; g.silu = silu;
  g.out = y->p; g.dtype = DT_F32;
  T2P_TRY(launch_gn_apply(g, s_));
  *out = y;
  const Norm N = n;
  tape_.push_back([this, x, y, N, silu, stats, B, HW]() -> int {
    if (!y->g || !x->needs_grad) return T2P_OK;
    T2P_GRAD(gx, x);
    float* ws = tmp((size_t)gn_bwd_ws_floats(B, HW, N.C, N.G) * 4);
    if (!ws) return T2P_ERR_HIP;
    return launch_gn_backward(x->p, y->g, stats, Pc_ + N.g, Pc_ + N.b, silu, B, HW, N.C, N.G, gx, Gr_ + N.g, Gr_ + N.b, ws, s_);
  });
  return T2P_OK;
}

int Trainer::layer_norm(TT* x, const Norm& n, TT** out) {
  T2P_REQUIRE(x->C == n.C, "LayerNorm channels");
  T2P_ACT(y, x->B, x->H, x->W, x->C);
  T2P_TRY(launch_layernorm(x->p, Pc_ + n.g, Pc_ + n.b, y->p, DT_F32, x->rows(), n.C, 1e-5f, s_));
  *out = y;
  const Norm N = n;
  tape_.push_back([this, x, y, N]() -> int {
    if (!y->g) return T2P_OK;
    T2P_GRAD(gx, x);
    return launch_ln_backward(x->p, y->g, Pc_ + N.g, x->rows(), N.C, 1e-5f, gx, Gr_ + N.g, Gr_ + N.b, s_);
  });
  return T2P_OK;
}

// softmax(scale q k^T) v per (sample, head): q [B][nq][C], k, v [B][nk][C], head h = columns [h d, (h + 1) d)
// (CrossAttention.forward, attention.py:170-191; AttnBlockpp with one head of width C, layers.py:168-172)
int Trainer::attention(TT* q, TT* k, TT* v, int heads, float scale, TT** out) {
  const int B = q->B, nq = q->H * q->W, nk = k->H * k->W, C = q->C, d = C / heads;
  T2P_REQUIRE(k->C == C && v->C == C && C % heads == 0 && k->B == B && v->B == B && v->H * v->W == nk, "attention shapes");
  T2P_ACT(o, q->B, q->H, q->W, C);
  const size_t pbytes = (size_t)B * heads * nq * nk * 4;
  float* S = tmp(pbytes);
  float* P = tmp(pbytes);
  if (!S || !P) return T2P_ERR_HIP;
  auto heads_of = [&](TGemmArgs& a) { a.nz0 = B; a.nz1 = heads; };
  TGemmArgs a;                                   // S = q k^T
  a.A = q->p; a.sAm = C; a.sAk = 1; a.sAz0 = (long)nq * C; a.sAz1 = d;
  a.B = k->p; a.sBk = 1; a.sBn = C; a.sBz0 = (long)nk * C; a.sBz1 = d;
  a.C = S; a.ldc = nk; a.sCz0 = (long)heads * nq * nk; a.sCz1 = (long)nq * nk; a.M = nq; a.N = nk; a.K = d;
  heads_of(a);
  T2P_TRY(launch_tgemm(a, s_));
  T2P_TRY(launch_softmax(S, nk, P, nk, DT_F32, (long)B * heads * nq, nk, scale, s_));
  TGemmArgs b;                                   // o = P v
  b.A = P; b.sAm = nk; b.sAk = 1; b.sAz0 = (long)heads * nq * nk; b.sAz1 = (long)nq * nk;
  b.B = v->p; b.sBk = C; b.sBn = 1; b.sBz0 = (long)nk * C; b.sBz1 = d;
  b.C = o->p; b.ldc = C; b.sCz0 = (long)nq * C; b.sCz1 = d; b.M = nq; b.N = d; b.K = nk;
  heads_of(b);
  T2P_TRY(launch_tgemm(b, s_));
  *out = o;
  tape_.push_back([this, q, k, v, o, P, S, B, heads, nq, nk, C, d, scale]() -> int {
    if (!o->g) return T2P_OK;
    const long sP0 = (long)heads * nq * nk, sP1 = (long)nq * nk;
    float* dP = S;                               // the raw scores are dead: their buffer takes dP, then dS
    TGemmArgs e;                                 // dP = dO v^T
    e.A = o->g; e.sAm = C; e.sAk = 1; e.sAz0 = (long)nq * C; e.sAz1 = d;
    e.B = v->p; e.sBk = 1; e.sBn = C; e.sBz0 = (long)nk * C; e.sBz1 = d;
    e.C = dP; e.ldc = nk; e.sCz0 = sP0; e.sCz1 = sP1; e.M = nq; e.N = nk; e.K = d; e.nz0 = B; e.nz1 = heads;
    T2P_TRY(launch_tgemm(e, s_));
    if (v->needs_grad) {                         // dv += P^T dO
      T2P_GRAD(gv, v);
      TGemmArgs f;
      f.A = P; f.sAm = 1; f.sAk = nk; f.sAz0 = sP0; f.sAz1 = sP1;
      f.B = o->g; f.sBk = C; f.sBn = 1; f.sBz0 = (long)nq * C; f.sBz1 = d;
      f.C = gv; f.ldc = C; f.sCz0 = (long)nk * C; f.sCz1 = d; f.M = nk; f.N = d; f.K = nq; f.nz0 = B; f.nz1 = heads; f.beta = 1.f;
      T2P_TRY(launch_tgemm(f, s_));
    }
    T2P_TRY(launch_softmax_backward(P, dP, (long)B * heads * nq, nk, scale, s_));     // dS (w.r.t. the raw scores q k^T)
    if (q->needs_grad) {                         // dq += dS k
      T2P_GRAD(gq, q);
      TGemmArgs f;
      f.A = dP; f.sAm = nk; f.sAk = 1; f.sAz0 = sP0; f.sAz1 = sP1;
      f.B = k->p; f.sBk = C; f.sBn = 1; f.sBz0 = (long)nk * C; f.sBz1 = d;
      f.C = gq; f.ldc = C; f.sCz0 = (long)nq * C; f.sCz1 = d; f.M = nq; f.N = d; f.K = nk; f.nz0 = B; f.nz1 = heads; f.beta = 1.f;
      T2P_TRY(launch_tgemm(f, s_));
    }
    if (k->needs_grad) {                         // dk += dS^T q
      T2P_GRAD(gk, k);
      TGemmArgs f;
      f.A = dP; f.sAm = 1; f.sAk = nk; f.sAz0 = sP0; f.sAz1 = sP1;
      f.B = q->p; f.sBk = C; f.sBn = 1; f.sBz0 = (long)nq * C; f.sBz1 = d;
      f.C = gk; f.ldc = C; f.sCz0 = (long)nk * C; f.sCz1 = d; f.M = nk; f.N = d; f.K = nq; f.nz0 = B; f.nz1 = heads; f.beta = 1.f;
      T2P_TRY(launch_tgemm(f, s_));
    }
    return T2P_OK;
  });
  return T2P_OK;
}

int Trainer::add_scale(TT* a, TT* b, float alpha, TT** out) {
  T2P_REQUIRE(a->numel() == b->numel() && a->C == b->C, "add_scale shapes");
  T2P_ACT(y, a->B, a->H, a->W, a->C);
  T2P_TRY(launch_add_scale(a->p, b->p, alpha, y->p, y->numel(), s_));
  *out = y;
  tape_.push_back([this, a, b, y, alpha]() -> int {
    if (!y->g) return T2P_OK;
    for (TT* t : {a, b}) {
      if (!t->needs_grad) continue;
      T2P_GRAD(gt, t);
      T2P_TRY(launch_axpy(gt, y->g, alpha, y->numel(), s_));
    }
    return T2P_OK;
  });
  return T2P_OK;
}

// ResnetBlockBigGANpp.forward in train mode (layers.py:303-327)
int Trainer::res_block(const LayerT& L, TT* x, TT* stemb, TT** out) {
  const ResL& r = L.r;
  const int B = x->B;
  const float alpha = mc_.skip_rescale ? (float)(1.0 / std::sqrt(2.0)) : 1.f;
  TT* a0 = nullptr;
  T2P_TRY(group_norm(x, r.gn0, 1, &a0));
  TT* xs = x;
  if (L.up || L.down) {
    const bool up = L.up != 0;
    const int H2 = up ? x->H * 2 : x->H / 2, W2 = up ? x->W * 2 : x->W / 2;
    TT* src[2] = {a0, x};
    TT* dst[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
      T2P_ACT(y, B, H2, W2, x->C);
      TT* in = src[i];
      if (up) T2P_TRY(launch_up2(in->p, y->p, B, in->H, in->W, in->C, s_));
      else T2P_TRY(launch_down2(in->p, y->p, B, in->H, in->W, in->C, s_));
      tape_.push_back([this, in, y, up]() -> int {
        if (!y->g || !in->needs_grad) return T2P_OK;
        T2P_GRAD(gi, in);
        return up ? launch_up2_backward(y->g, gi, in->B, in->H, in->W, in->C, s_) : launch_down2_backward(y->g, gi, in->B, in->H, in->W, in->C, s_);
      });
      dst[i] = y;
    }
    a0 = dst[0]; xs = dst[1];
  }
  const int H = a0->H, W = a0->W, HW = H * W;
  TT* tb = nullptr;                                // Dense_0(act(temb)) [B][Cout]
  T2P_TRY(linear(stemb, r.dense, &tb));
  // h = Conv_0(a0) + bias + tb
  T2P_REQUIRE(a0->C == r.c0.Cip && r.c0.Cop == r.c0.Co && r.c1.Cop == r.c1.Co && r.c1.Cip == r.c1.Ci, "residual block channels are multiples of 8");
  T2P_ACT(h, B, H, W, r.c0.Co);
  T2P_TRY(conv_forward(a0->p, B, H, W, a0->C, r.c0.wf, 9L * r.c0.Cip, Pc_ + r.c0.b, tb->p, r.c0.Co, h->p, r.c0.Co, nullptr, s_));
  auto conv_backward = [this](TT* in, TT* y, const Conv c, TT* tbias) -> int {
    if (!y->g) return T2P_OK;
    const int Bq = in->B, Hq = in->H, Wq = in->W;
    const long rows = in->rows();
    if (in->needs_grad) {                          // dX = conv3x3(dY, flipped transposed taps), accumulated in place through the residual operand
      T2P_GRAD(gi, in);
      T2P_TRY(conv_forward(y->g, Bq, Hq, Wq, c.Cop, c.wd, 9L * c.Cop, nullptr, nullptr, c.Ci, gi, in->C, gi, s_));
    }
    T2P_HIP_CHECK(hipMemsetAsync(dwc_, 0, (size_t)c.Co * 9 * c.Cip * 4, s_));
    TGemmArgs w;                                   // dW[co][tap][ci] = sum_pixels dY[pixel][co] X[pixel + tap][ci]
    w.A = y->g; w.sAm = 1; w.sAk = y->C; w.B = in->p; w.conv_b = 1; w.H = Hq; w.W = Wq; w.conv_C = c.Cip; w.ldx = in->C;
    w.C = dwc_; w.ldc = 9L * c.Cip; w.M = c.Co; w.N = 9 * c.Cip; w.K = (int)rows; w.beta = 1.f; w.ksplit = 0;
    T2P_TRY(launch_tgemm(w, s_));
    T2P_TRY(launch_conv_w_grad_fold(dwc_, Gr_ + c.w, c.Co, c.Ci, c.Cip, s_));
    T2P_TRY(launch_colsum(y->g, rows, c.Co, y->C, Gr_ + c.b, s_));
    if (tbias) {
      T2P_GRAD(gt, tbias);
      T2P_TRY(launch_colsum_per_sample(y->g, Bq, Hq * Wq, c.Co, gt, c.Co, 1, s_));
    }
    return T2P_OK;
  };
  {
    const Conv c = r.c0;
    tape_.push_back([conv_backward, a0, h, c, tb]() -> int { return conv_backward(a0, h, c, tb); });
  }
  TT* a1 = nullptr;
  T2P_TRY(group_norm(h, r.gn1, 1, &a1));
  if (tc_.dropout > 0.0) {                         // Dropout_0 (layers.py:318)
    const long n = a1->numel();
    const uint8_t* keep = nullptr;
    if (!drop_masks_.empty()) {
      T2P_REQUIRE(drop_index_ < (int)drop_masks_.size(), "fewer dropout masks than residual blocks");
      keep = drop_masks_[drop_index_];
    } else {
      uint8_t* m = (uint8_t*)tmp((size_t)n);
      if (!m) return T2P_ERR_HIP;
      T2P_TRY(launch_dropout_mask(m, n, (float)tc_.dropout, tc_.seed, (unsigned long long)(loss_calls_ * 4096 + 16 + drop_index_), s_));
      keep = m;
    }
    ++drop_index_;
    const float inv_keep = (float)(1.0 / (1.0 - tc_.dropout));
    T2P_ACT(y, a1->B, a1->H, a1->W, a1->C);
    T2P_TRY(launch_dropout(a1->p, keep, inv_keep, y->p, n, 0, s_));
    TT* in = a1;
    tape_.push_back([this, in, y, keep, inv_keep, n]() -> int {
      if (!y->g) return T2P_OK;
      T2P_GRAD(gi, in);
      return launch_dropout(y->g, keep, inv_keep, gi, n, 1, s_);
    });
    a1 = y;
  }
  T2P_ACT(h2, B, H, W, r.c1.Co);
  T2P_TRY(conv_forward(a1->p, B, H, W, a1->C, r.c1.wf, 9L * r.c1.Cip, Pc_ + r.c1.b, nullptr, r.c1.Co, h2->p, r.c1.Co, nullptr, s_));
  {
    const Conv c = r.c1;
    tape_.push_back([conv_backward, a1, h2, c]() -> int { return conv_backward(a1, h2, c, nullptr); });
  }
  TT* sc = xs;
  if (r.has_sc) T2P_TRY(linear(xs, r.sc, &sc));
  (void)HW;
  return add_scale(sc, h2, alpha, out);
}

// AttnBlockpp.forward (layers.py:160-176)
int Trainer::attn_block(const LayerT& L, TT* x, TT** out) {
  const AttnL& a = L.a;
  const float alpha = mc_.skip_rescale ? (float)(1.0 / std::sqrt(2.0)) : 1.f;
  TT *h = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *o = nullptr, *y = nullptr;
  T2P_TRY(group_norm(x, a.gn, 0, &h));
  T2P_TRY(linear(h, a.nin[0], &q));
  T2P_TRY(linear(h, a.nin[1], &k));
  T2P_TRY(linear(h, a.nin[2], &v));
  T2P_TRY(attention(q, k, v, 1, 1.f / std::sqrt((float)x->C), &o));
  T2P_TRY(linear(o, a.nin[3], &y));
  return add_scale(x, y, alpha, out);
}

// SpatialTransformer.forward with one BasicTransformerBlock (model/attention.py:208-215, 250-263)
int Trainer::st_block(const LayerT& L, TT* x, TT* ctx, TT** out) {
  const StL& s = L.st;
  const int heads = mc_.n_heads, C = x->C;
  const float scale = 1.f / std::sqrt((float)(C / heads));
  TT *a = nullptr, *t0 = nullptr;
  T2P_TRY(group_norm(x, s.gn, 0, &a));
  T2P_TRY(linear(a, s.proj_in, &t0));
  auto attn = [&](TT* t, const Norm& ln, const Lin& wq, const Lin& wk, const Lin& wv, const Lin& wo, TT* kv_src, TT** res) -> int {
    TT *l = nullptr, *q = nullptr, *k = nullptr, *v = nullptr, *o = nullptr, *y = nullptr;
    T2P_TRY(layer_norm(t, ln, &l));
    T2P_TRY(linear(l, wq, &q));
    T2P_TRY(linear(kv_src ? kv_src : l, wk, &k));
    T2P_TRY(linear(kv_src ? kv_src : l, wv, &v));
    T2P_TRY(attention(q, k, v, heads, scale, &o));
    T2P_TRY(linear(o, wo, &y));
    return add_scale(y, t, 1.f, res);
  };
  TT *t1 = nullptr, *t2 = nullptr;
  T2P_TRY(attn(t0, s.ln[0], s.q1, s.k1, s.v1, s.o1, nullptr, &t1));
  T2P_TRY(attn(t1, s.ln[1], s.q2, s.k2, s.v2, s.o2, ctx, &t2));
  TT *l3 = nullptr, *u = nullptr, *y = nullptr, *t3 = nullptr, *po = nullptr;
  T2P_TRY(layer_norm(t2, s.ln[2], &l3));
  T2P_TRY(linear(l3, s.ff1, &u));
  const int inner = 4 * C;
  T2P_ACT(g, u->B, u->H, u->W, inner);
  T2P_TRY(launch_geglu(u->p, g->p, DT_F32, u->rows(), inner, s_));
  tape_.push_back([this, u, g, inner]() -> int {
    if (!g->g) return T2P_OK;
    T2P_GRAD(gu, u);
    return launch_geglu_backward(u->p, g->g, gu, u->rows(), inner, s_);
  });
  T2P_TRY(linear(g, s.ff2, &y));
  T2P_TRY(add_scale(y, t2, 1.f, &t3));
  T2P_TRY(linear(t3, s.proj_out, &po));
  return add_scale(po, x, 1.f, out);
}

int Trainer::run_layers(const std::vector<LayerT>& ls, TT* h, TT* stemb, TT* ctx, TT** out) {
  for (const LayerT& l : ls) {
    TT* y = nullptr;
    if (l.kind == 0) T2P_TRY(res_block(l, h, stemb, &y));
    else if (l.kind == 1) T2P_TRY(attn_block(l, h, &y));
    else T2P_TRY(st_block(l, h, ctx, &y));
    h = y;
  }
  *out = h;
  return T2P_OK;
}

// loss_fn (losses.py:105-134) on the parameters P; with `backward`, d loss / d P accumulates into Gr_
int Trainer::forward_backward(const t2p_train_batch& b, const float* P, bool backward, float* loss_dev, float* score_out) {
  const int B = b.batch, L = mc_.max_res_num, HW = L * L, Cx = mc_.num_channels, nf = mc_.nf;
  T2P_REQUIRE(b.coords_6d && b.mask_pair && b.context && B > 0 && b.tokens > 0, "training batch");
  T2P_REQUIRE(!(tc_.cond_flags & 4) || b.mask_inpaint, "the inpainting condition needs batch.mask_inpaint");
  T2P_REQUIRE(!(tc_.cond_flags & 2) || Cx >= 7, "the ss condition needs the 8-channel layout");
  Pc_ = P;
  drop_index_ = 0;
  T2P_TRY(prep_weights(P, s_));
  const long nx = (long)B * Cx * HW;
  float* t_dev = tmp(B * 4); float* stdv = tmp(B * 4); float* scale = tmp(B * 4); float* num_elem = tmp(B * 4);
  int* labels = (int*)tmp(B * 4);
  double* loss_sum = (double*)tmp(B * 8);
  float* perturbed = tmp(nx * 4);
  uint8_t* mask = (uint8_t*)tmp(nx);
  if (!t_dev || !stdv || !scale || !num_elem || !labels || !loss_sum || !perturbed || !mask) return T2P_ERR_HIP;
  T2P_TRY(launch_dsm_prepare(b.t, B, (float)tc_.t_eps, (float)mc_.sigma_min, (float)mc_.sigma_max, mc_.num_scales,
                             mc_.scale_by_sigma ? inv_sigma_ : nullptr, tc_.seed, (unsigned long long)loss_calls_, t_dev, stdv, labels, scale, s_));
  const float* z = b.z;
  if (!z) {
    float* zb = tmp(nx * 4);
    if (!zb) return T2P_ERR_HIP;
    T2P_TRY(launch_philox_normal(zb, nx, tc_.seed, (unsigned long long)(loss_calls_ * 4096 + 1), nullptr, s_));
    z = zb;
  }
  T2P_TRY(launch_dsm_perturb(b.coords_6d, z, stdv, b.mask_pair, b.mask_inpaint, tc_.cond_flags, B, Cx, L, perturbed, mask, num_elem, s_));

  // UNetModel.forward (ncsnpp.py:220-263)
  T2P_ACT(x0, B, L, L, 8, false);
  T2P_TRY(launch_nchw_to_nhwc(perturbed, x0->p, B, Cx, HW, 8, s_));
  T2P_ACT(emb, B, 1, 1, nf, false);
  T2P_TRY(launch_timestep_embedding(labels, nullptr, nullptr, emb->p, B, nf, s_));
  TT *te1 = nullptr, *temb = nullptr;
  T2P_TRY(linear(emb, pre0_, &te1));
  T2P_TRY(linear(te1, pre1_, &temb));
  T2P_ACT(stemb, B, 1, 1, temb->C);                // act(temb): the same tensor for every block (layers.py:316)
  T2P_TRY(launch_silu(temb->p, stemb->p, temb->numel(), s_));
  tape_.push_back([this, temb, stemb]() -> int {
    if (!stemb->g) return T2P_OK;
    T2P_GRAD(gt, temb);
    return launch_silu_backward(temb->p, stemb->g, gt, temb->numel(), s_);
  });
  acts_.emplace_back();                            // the text context: caller-owned, no gradient
  TT* ctx = &acts_.back();
  ctx->p = const_cast<float*>(b.context); ctx->B = B; ctx->H = b.tokens; ctx->W = 1; ctx->C = mc_.context_dim; ctx->needs_grad = false;

  T2P_ACT(h0, B, L, L, nf);
  T2P_TRY(conv_forward(x0->p, B, L, L, 8, pre_conv_.wf, 9L * 8, P + pre_conv_.b, nullptr, nf, h0->p, nf, nullptr, s_));
  {
    const Conv c = pre_conv_;
    tape_.push_back([this, x0, h0, c]() -> int {
      if (!h0->g) return T2P_OK;
      T2P_HIP_CHECK(hipMemsetAsync(dwc_, 0, (size_t)c.Co * 9 * c.Cip * 4, s_));
      TGemmArgs w;
      w.A = h0->g; w.sAm = 1; w.sAk = h0->C; w.B = x0->p; w.conv_b = 1; w.H = x0->H; w.W = x0->W; w.conv_C = c.Cip; w.ldx = x0->C;
      w.C = dwc_; w.ldc = 9L * c.Cip; w.M = c.Co; w.N = 9 * c.Cip; w.K = (int)x0->rows(); w.beta = 1.f; w.ksplit = 0;
      T2P_TRY(launch_tgemm(w, s_));
      T2P_TRY(launch_conv_w_grad_fold(dwc_, Gr_ + c.w, c.Co, c.Ci, c.Cip, s_));
      return launch_colsum(h0->g, h0->rows(), c.Co, h0->C, Gr_ + c.b, s_);
    });
  }
  std::vector<TT*> hs{h0};
  TT* h = h0;
  for (const auto& st : in_stages_) {
    T2P_TRY(run_layers(st, h, stemb, ctx, &h));
    hs.push_back(h);
  }
  T2P_TRY(run_layers(mid_, h, stemb, ctx, &h));
  for (const auto& st : out_stages_) {
    TT* skip = hs.back();
    hs.pop_back();
    T2P_REQUIRE(skip->H == h->H && skip->B == h->B, "skip stack mismatch");
    T2P_ACT(cat, B, h->H, h->W, h->C + skip->C);   // torch.cat([h, hs.pop()], dim=1), ncsnpp.py:250
    T2P_TRY(launch_copy_cols(h->p, h->C, 0, cat->p, cat->C, 0, h->rows(), h->C, 0, s_));
    T2P_TRY(launch_copy_cols(skip->p, skip->C, 0, cat->p, cat->C, h->C, h->rows(), skip->C, 0, s_));
    TT* hin = h;
    tape_.push_back([this, hin, skip, cat]() -> int {
      if (!cat->g) return T2P_OK;
      T2P_GRAD(g0, hin);
      T2P_GRAD(g1, skip);
      T2P_TRY(launch_copy_cols(cat->g, cat->C, 0, g0, hin->C, 0, hin->rows(), hin->C, 1, s_));
      return launch_copy_cols(cat->g, cat->C, hin->C, g1, skip->C, 0, hin->rows(), skip->C, 1, s_);
    });
    T2P_TRY(run_layers(st, cat, stemb, ctx, &h));
  }
  T2P_REQUIRE(hs.empty(), "skip stack not consumed");
  TT* a = nullptr;
  T2P_TRY(group_norm(h, head_norm_, 1, &a));
  T2P_ACT(o, B, L, L, 8);                          // head convolution: Cx of 8 columns used
  T2P_HIP_CHECK(hipMemsetAsync(o->p, 0, (size_t)o->numel() * 4, s_));
  T2P_TRY(conv_forward(a->p, B, L, L, a->C, head_conv_.wf, 9L * head_conv_.Cip, P + head_conv_.b, nullptr, Cx, o->p, 8, nullptr, s_));
  {
    const Conv c = head_conv_;
    tape_.push_back([this, a, o, c]() -> int {
      if (!o->g) return T2P_OK;
      T2P_GRAD(ga, a);
      T2P_TRY(conv_forward(o->g, a->B, a->H, a->W, 8, c.wd, 9L * 8, nullptr, nullptr, c.Ci, ga, a->C, ga, s_));
      T2P_HIP_CHECK(hipMemsetAsync(dwc_, 0, (size_t)c.Co * 9 * c.Cip * 4, s_));
      TGemmArgs w;
      w.A = o->g; w.sAm = 1; w.sAk = 8; w.B = a->p; w.conv_b = 1; w.H = a->H; w.W = a->W; w.conv_C = c.Cip; w.ldx = a->C;
      w.C = dwc_; w.ldc = 9L * c.Cip; w.M = c.Co; w.N = 9 * c.Cip; w.K = (int)a->rows(); w.beta = 1.f; w.ksplit = 0;
      T2P_TRY(launch_tgemm(w, s_));
      T2P_TRY(launch_conv_w_grad_fold(dwc_, Gr_ + c.w, c.Co, c.Ci, c.Cip, s_));
      return launch_colsum(o->g, a->rows(), c.Co, 8, Gr_ + c.b, s_);
    });
  }
  float* d_o = nullptr;
  if (backward) {
    d_o = grad(o);
    if (!d_o) return T2P_ERR_HIP;
  }
  T2P_TRY(launch_dsm_loss(o->p, 8, z, stdv, scale, mask, num_elem, B, Cx, L, loss_sum, d_o, 8, score_out, s_));
  T2P_TRY(launch_dsm_finish(loss_sum, num_elem, B, loss_dev, s_));
  if (backward)
    for (auto it = tape_.rbegin(); it != tape_.rend(); ++it) T2P_TRY((*it)());
  return T2P_OK;
}

int Trainer::loss(const t2p_train_batch& b, bool backward, bool use_ema, float* loss_host, float* score_out, hipStream_t s) {
  T2P_REQUIRE(loss_host, "loss output");
  s_ = s;
  if (backward) T2P_HIP_CHECK(hipMemsetAsync(Gr_, 0, (size_t)total_ * 4, s));      // optimizer.zero_grad()
  const double keep_dropout = tc_.dropout;
  if (use_ema) tc_.dropout = 0.0;                 // eval mode (models/utils.py:113-115)
  const int rc = forward_backward(b, use_ema ? E_ : P_, backward, loss_dev_, score_out);
  tc_.dropout = keep_dropout;
  ++loss_calls_;
  const hipError_t e = hipStreamSynchronize(s);
  release();
  if (rc != T2P_OK) return rc;
  T2P_HIP_CHECK(e);
  T2P_HIP_CHECK(hipMemcpy(loss_host, loss_dev_, 4, hipMemcpyDeviceToHost));
  return T2P_OK;
}

int Trainer::step(const t2p_train_batch& b, float* loss_host, hipStream_t s) {
  T2P_TRY(loss(b, true, false, loss_host, nullptr, s));
  return apply(s);
}

int Trainer::apply(hipStream_t s) {
  // optimize_fn (losses.py:41-49)
  AdamArgs a;
  a.p = P_; a.g = Gr_; a.m = M_; a.v = V_; a.n = total_;
  a.lr = (float)(tc_.warmup > 0 ? tc_.lr * std::min((double)step_ / tc_.warmup, 1.0) : tc_.lr);
  a.beta1 = (float)tc_.beta1; a.beta2 = 0.999f; a.eps = (float)tc_.eps; a.weight_decay = (float)tc_.weight_decay;
  const int64_t k = adam_k_ + 1;
  a.bias1 = (float)(1.0 - std::pow(tc_.beta1, (double)k));
  a.bias2_sqrt = (float)std::sqrt(1.0 - std::pow(0.999, (double)k));
  a.grad_clip = (float)tc_.grad_clip;
  if (tc_.grad_clip >= 0) {
    T2P_HIP_CHECK(hipMemsetAsync(sumsq_, 0, 8, s));
    T2P_TRY(launch_sumsq(Gr_, total_, sumsq_, s));
    a.sumsq = sumsq_;
  }
  T2P_TRY(launch_adam(a, s));
  adam_k_ = k;
  step_ += 1;
  // ema.update (ema.py:32-49)
  ema_k_ += 1;
  const double decay = std::min(tc_.ema_rate, (1.0 + (double)ema_k_) / (10.0 + (double)ema_k_));
  T2P_TRY(launch_ema(E_, P_, (float)(1.0 - decay), total_, s));
  T2P_HIP_CHECK(hipStreamSynchronize(s));
  return T2P_OK;
}

}  // namespace t2p
