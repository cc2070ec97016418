// Training step of the score network (first slice: fp32 only): forward pass that keeps what the backward pass
// needs, backward pass, denoising score-matching loss, Adam / warm-up / clipping and the EMA update
// (reference score_sde_pytorch/losses.py:26-186, score_sde_pytorch/models/ema.py:32-49).
#pragma once
#include <deque>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "engine.h"
#include "train_kernels.h"

namespace t2p {

struct TParam {           // one learnable tensor: a slice of the flat buffers, in the reference's parameters() order
  std::string name;
  std::vector<int64_t> shape;
  long off = 0, n = 0;
};

struct TT {               // an NHWC activation [B][H W][C] fp32 of the training pass and its gradient (allocated on first use)
  float* p = nullptr;
  float* g = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  bool needs_grad = true;
  long rows() const { return (long)B * H * W; }
  long numel() const { return rows() * C; }
};

class Trainer {
 public:
  Trainer(const t2p_model_config& mc, const t2p_train_config& tc);
  ~Trainer();
  int build();
  const std::vector<TParam>& params() const { return params_; }
  long num_elements() const { return total_; }
  int load_param(const char* name, const float* host, const int64_t* shape, int ndim);
  // which: 0 parameter, 1 gradient (of the last backward pass; after a step: the clipped one), 2 EMA shadow, 3 Adam exp_avg, 4 exp_avg_sq
  int read_tensor(int which, const char* name, float* host_out);
  int write_tensor(int which, const char* name, const float* host_in);
  int set_step(int64_t step, int64_t adam_updates, int64_t ema_updates);
  int get_step(int64_t out[3]) const;
  int set_dropout_masks(const uint8_t* const* masks, int n);
  // loss_fn (losses.py:105-134); with `backward` also d loss / d parameters into the gradient buffer (zeroed first: optimizer.zero_grad())
  int loss(const t2p_train_batch& b, bool backward, bool use_ema, float* loss_host, float* score_out, hipStream_t s);
  // step_fn with train=True (losses.py:165-176) = loss(backward) + apply()
  int step(const t2p_train_batch& b, float* loss_host, hipStream_t s);
  // optimize_fn (losses.py:41-49) on the gradient buffer as it stands, state['step'] += 1, ema.update (ema.py:32-49).  Data-parallel training
  // calls loss(backward), averages grad_buffer() over the ranks (one RCCL all-reduce of the flat buffer), then this
  int apply(hipStream_t s);
  float* grad_buffer() const { return Gr_; }
  int64_t device_bytes() const { return (int64_t)pool_.held_bytes(); }

 private:
  struct Conv { long w = 0, b = 0; int Co = 0, Ci = 0, Cip = 0, Cop = 0; float* wf = nullptr; float* wd = nullptr; };   // offsets into the flat buffers
  struct Lin { long w = 0, b = -1; int N = 0, K = 0; bool nin = false; };
  struct Norm { long g = 0, b = 0; int C = 0, G = 0; };
  struct ResL { Norm gn0, gn1; Conv c0, c1; Lin dense, sc; bool has_sc = false; };
  struct AttnL { Norm gn; Lin nin[4]; };
  struct StL { Norm gn, ln[3]; Lin proj_in, proj_out, q1, k1, v1, o1, q2, k2, v2, o2, ff1, ff2; };
  struct LayerT { int kind = 0, in_ch = 0, out_ch = 0, up = 0, down = 0; ResL r; AttnL a; StL st; };

  long poff(const std::string& name, std::vector<int64_t> shape);
  int map_layer(const Layer& l, LayerT* out);
  int prep_weights(const float* P, hipStream_t s);

  // forward ops: each appends its backward to tape_
  TT* act(int B, int H, int W, int C, bool needs_grad = true);
  float* grad(TT* t);                       // gradient buffer of t, zero-initialised on first use
  float* tmp(size_t bytes);                 // released at the end of the call
  int linear(TT* x, const Lin& l, TT** out);
  int group_norm(TT* x, const Norm& n, int silu, TT** out);
  int layer_norm(TT* x, const Norm& n, TT** out);
  int attention(TT* q, TT* k, TT* v, int heads, float scale, TT** out);
  int add_scale(TT* a, TT* b, float alpha, TT** out);
  int res_block(const LayerT& L, TT* x, TT* stemb, TT** out);
  int attn_block(const LayerT& L, TT* x, TT** out);
  int st_block(const LayerT& L, TT* x, TT* ctx, TT** out);
  int run_layers(const std::vector<LayerT>& ls, TT* h, TT* stemb, TT* ctx, TT** out);
  int forward_backward(const t2p_train_batch& b, const float* P, bool backward, float* loss_dev, float* score_out);
  void release();

  t2p_model_config mc_;
  t2p_train_config tc_;
  Engine arch_;                     // structure and parameter table only (Engine::build); never finalized, holds no weights
  std::vector<TParam> params_;
  std::unordered_map<std::string, int> index_;
  long total_ = 0;
  float* P_ = nullptr;              // flat parameters
  float* Gr_ = nullptr;             // flat gradients
  float* M_ = nullptr; float* V_ = nullptr; float* E_ = nullptr;    // Adam moments, EMA shadow
  double* sumsq_ = nullptr;
  float* loss_dev_ = nullptr;
  float* inv_sigma_ = nullptr;
  int64_t step_ = 0, adam_k_ = 0, ema_k_ = 0, loss_calls_ = 0;
  std::vector<LayerT> in_layers_;
  std::vector<std::vector<LayerT>> in_stages_, out_stages_;
  std::vector<LayerT> mid_;
  Lin pre0_, pre1_;
  Conv pre_conv_, head_conv_;
  Norm head_norm_;
  std::vector<Conv*> convs_;
  float* dwc_ = nullptr; size_t dwc_floats_ = 0;     // weight-gradient tile of the largest convolution, compute layout
  std::vector<const uint8_t*> drop_masks_;
  int drop_index_ = 0;

  DevPool pool_;
  hipStream_t s_ = nullptr;
  const float* Pc_ = nullptr;       // the parameters of the running pass (P_ or E_)
  std::deque<TT> acts_;
  std::vector<void*> live_;
  std::vector<std::function<int()>> tape_;
};

}  // namespace t2p

struct t2p_trainer { t2p::Trainer impl; t2p_trainer(const t2p_model_config& m, const t2p_train_config& t) : impl(m, t) {} };
