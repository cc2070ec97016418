// Score-network engine and predictor-corrector sampler (host side, drives the HIP kernels).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/t2p.h"
#include "t2p_kernels.h"

namespace t2p {

// Caching device allocator: exact-size free lists.  The sequence of requests of one forward pass
// is the same every evaluation, so after the first pass no hipMalloc happens (graph-capture safe).
class DevPool {
 public:
  ~DevPool();
  void* get(size_t bytes);
  void put(void* p);
  size_t held_bytes() const { return held_; }
  void* persistent(size_t bytes);  // never returned to the free lists (weights, tables)
  // Lease scope of one evaluation: every buffer taken between lease_begin and lease_end that has not been put back (or handed over
  // with keep()) by then is returned to the free lists by lease_end -- an early return anywhere inside a block cannot leave buffers
  // checked out for the lifetime of the engine.  lease_end returns how many it had to reclaim (0 on every complete evaluation).
  void lease_begin();
  int lease_end();
  void keep(void* p) { leased_.erase(p); }   // the buffer outlives the scope (cached text keys / values)
  int last_reclaimed() const { return last_reclaimed_; }
 private:
  std::multimap<size_t, void*> free_;
  std::unordered_map<void*, size_t> size_of_;
  std::vector<void*> all_;
  std::unordered_set<void*> leased_;
  int lease_depth_ = 0, last_reclaimed_ = 0;
  size_t held_ = 0;
};
// RAII form: the scope of Engine::score / Engine::set_context
struct PoolLease {
  explicit PoolLease(DevPool& p) : pool(p) { pool.lease_begin(); }
  ~PoolLease() { pool.lease_end(); }
  PoolLease(const PoolLease&) = delete;
  PoolLease& operator=(const PoolLease&) = delete;
  DevPool& pool;
};

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct ParamInfo {
  std::string name;
  std::vector<int64_t> shape;
};

struct DevLinear {      // weights [N][K] in compute dtype (+ fp32 bias)
  void* w = nullptr;
  float* b = nullptr;
  int N = 0, K = 0;     // K = taps * Cin for convolutions
};
struct DevNorm {
  float* gamma = nullptr;
  float* beta = nullptr;
  int C = 0, G = 0;
};

struct Layer {
  int kind = 0;  // 0 res, 1 attn, 2 st
  std::string prefix;
  int in_ch = 0, out_ch = 0, up = 0, down = 0;
  // res
  DevNorm gn0, gn1;
  DevLinear conv0, conv1, conv2;
  void* conv0_up4 = nullptr;   // up blocks, 16-bit modes: conv0 as four 2x2 phase convolutions, [4][Cout][4 * Cin] (GemmParams::Bw4)
  // blocks with a 1x1 shortcut at the block's own resolution, 16-bit modes: conv1 and the shortcut as one K loop,
  // weights [Cout][9 * Cout + Cin] and bias b1 + b2 (GemmParams::X0)
  DevLinear conv1x;
  // fragment-major copies of conv0 / conv1 (or conv1x) for the small-map convolution kernel, made at its first use
  void* fm_conv0 = nullptr; void* fm_conv1 = nullptr;
  bool has_conv2 = false;
  int temb_off = 0;
  // attn (AttnBlockpp): NIN_0|NIN_1 stacked, NIN_2, NIN_3
  DevLinear qk, v, out;
  // 16-bit modes: NIN_2 and NIN_3 as one matrix -- rows of softmax(..) sum to 1, so NIN_3(P (h W2 + b2)) = P (h W2 W3) + (b2 W3 + b3):
  // v3 = (W2 W3)^T [C][C] feeds the transposed value projection, v3.b = b2 W3 + b3 is added by the attention kernel's epilogue
  DevLinear v3;
  void* fm_qk = nullptr; void* fm_v3 = nullptr;   // C = 256: fragment-major copies of qk and v3 for attn_proj_kernel (stfuse.hip)
  // st (SpatialTransformer)
  DevLinear a1_qkv;   // 16-bit modes: to_q | to_k | to_v stacked, one projection GEMM for the self-attention
  DevLinear proj_in, proj_out, a1_qk, a1_v, a1_out, a2_q, a2_k, a2_v, a2_out, ff1, ff2;
  // 16-bit modes: the feed-forward's second Linear and proj_out as one GEMM over [g | t] (no nonlinearity between them):
  // proj_out(t + ff2(g)) = [W_po W_ff2 | W_po] [g ; t] + (W_po b_ff2 + b_po); ffpo.w = [C][4 C + C]
  DevLinear ffpo;
  // C = 256, 16-bit modes: fragment-major copies of proj_in, a1_qkv, a1_out, a2_q, a2_out, ff1 for the row-chain kernel (stfuse.hip)
  void* fm_in = nullptr; void* fm_qkv = nullptr; void* fm_out1 = nullptr; void* fm_q2 = nullptr; void* fm_out2 = nullptr; void* fm_ff1 = nullptr; void* fm_ffpo = nullptr;
  DevNorm ln1, ln2, ln3;
  void* ctx_k = nullptr;   // [B][T][C]      compute dtype (set_context)
  void* ctx_vt = nullptr;  // [B][C][Tpad]   compute dtype
};

struct Stage {
  std::vector<Layer> layers;
  int skip_ch = 0;
};

struct Act {  // an NHWC activation: fp32, or the compute dtype when `lowp`
  float* p = nullptr;
  int C = 0, H = 0, W = 0;
  float* cstats = nullptr;   // optional per-64-row column sums of p (GemmParams::col_stats), for the consumer's GroupNorm
  bool lowp = false;         // p holds compute-dtype (16-bit) values instead of fp32
  // optional: act(GroupNorm(p)) for the NEXT block's first norm, already produced by the split-K second pass that wrote p
  // (GemmParams::gn_out); the consumer whose norm is `pre_for` takes it (and returns it to the pool), run_stage drops it otherwise
  mutable void* pre_norm = nullptr;
  const DevNorm* pre_for = nullptr;
  int pre_silu = 0;
};
struct NormHint {            // the norm the consumer of a block's output will apply first
  const DevNorm* norm = nullptr;
  int silu = 0;
};

class Engine {
 public:
  explicit Engine(const t2p_model_config& cfg);
  ~Engine();
  int build();  // structure + parameter table
  const std::vector<ParamInfo>& params() const { return params_; }
  int load_param(const char* name, const float* data, const int64_t* shape, int ndim);
  int finalize();
  int set_context(const float* ctx, int B, int T, hipStream_t s);
  // labels == nullptr: every row uses *step_counter (device int)
  // label_table (with step_counter): device int[num_scales], the time label of loop step i (fused sampler)
  // label_f_table (with step_counter): device float[num_scales], the fractional time label of loop step i (VP SDE)
  int score(const float* x, const int* labels, const int* step_counter, float* out, int B, hipStream_t s,
            const float* labels_f = nullptr, const int* label_table = nullptr, const float* label_f_table = nullptr);
  int64_t device_bytes() const { return (int64_t)pool_.held_bytes(); }
  int pool_reclaimed() const { return pool_.last_reclaimed(); }
  const t2p_model_config& cfg() const { return cfg_; }
  DevPool& pool() { return pool_; }
  int dtype() const { return cfg_.compute_dtype; }

 private:
  int upload_linear(const std::string& wname, const std::string& bname, int N, int K, DevLinear* out,
                    bool conv3x3 = false, bool nin = false, int force_dtype = -1);
  int upload_stack2(const std::string& w0, const std::string& b0, const std::string& w1, const std::string& b1,
                    int N, int K, bool nin, bool has_bias, DevLinear* out);
  int upload_norm(const std::string& prefix, int C, int G, DevNorm* out);
  int upload_f32(const std::vector<float>& v, float** out);
  const HostTensor* host(const std::string& name, std::vector<int64_t> shape);

  int run_stage(Stage& st, Act& h, const Act* skip, int B, hipStream_t s, const Layer* next_after = nullptr);
  int res_block(Layer& L, const Act& x, const Act* skip, Act* out, int B, hipStream_t s, NormHint hint = NormHint());
  int attn_block(Layer& L, const Act& x, Act* out, int B, hipStream_t s);
  int st_block(Layer& L, const Act& x, Act* out, int B, hipStream_t s);
  int group_norm(const Act& x, const Act* x1, const DevNorm& n, float eps, int silu, int down, int B, void** out,
                 hipStream_t s, void** raw_out = nullptr);
  int gemm(GemmParams& p, hipStream_t s);
  int attach_ws(GemmParams& p);
  // gemm() that also produces the output's GroupNorm column statistics when the kernel can (else *cstats = null)
  int gemm_stats(GemmParams& p, float** cstats, hipStream_t s);
  void free_act(Act& a) { pool_.put(a.p); pool_.put(a.cstats); pool_.put(a.pre_norm); a.p = nullptr; a.cstats = nullptr; a.pre_norm = nullptr; }
  void drop_pre_norm(const Act& a) { pool_.put(a.pre_norm); a.pre_norm = nullptr; }
  int attention(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, int B,
                int heads, int nq, int nk, int d, float scale, hipStream_t s);
  int linear(const void* a, bool a_is_f32, const DevLinear& w, long rows, void* c, bool c_f32, const float* residual,
             float alpha, hipStream_t s, bool use_bias = true, float** cstats = nullptr, bool r_lowp = false);
  // the residual stream between blocks is stored in the compute dtype (f16 mode only): halves the HBM-bound
  // traffic of GroupNorm-apply and of the residual / output halves of the block-closing GEMM epilogues
  bool res_lowp() const { return g_lowp_residual && cfg_.compute_dtype == DT_F16; }

  t2p_model_config cfg_;
  std::vector<ParamInfo> params_;
  std::unordered_map<std::string, HostTensor> host_;
  bool finalized_ = false;
  DevPool pool_;

  int nf_ = 0, temb_dim_ = 0, cpad_ = 0, final_ch_ = 0;
  std::vector<Stage> input_stages_, out_stages_;
  Stage mid_stage_;
  DevLinear pre0_, pre1_, pre_conv_, head_conv_, dense_all_;
  DevNorm head_norm_;
  int temb_total_ = 0;
  float* pre_conv_direct_ = nullptr;   // [nf][9][C] fp32 weights of the direct input convolution
  void* pre_conv_split_ = nullptr;     // the same, each weight as two f16 terms (pre_conv_split_kernel; 16-bit modes)
  float* inv_sigma_ = nullptr;  // [N] fp32, 1 / sigmas[label] (descending sigmas)
  int ctx_B_ = 0, ctx_T_ = 0, ctx_Tpad_ = 0;
  void* splitk_ws_ = nullptr;
  size_t splitk_ws_bytes_ = 0;
  const float* tb_ = nullptr;   // per-eval temb biases [R][temb_total_]
  long tb_ld_ = 0;
  friend class Sampler;
  friend class Trainer;     // train.cpp: reads the block list and the parameter table (Engine::build), nothing else
};

class Sampler {
 public:
  Sampler(Engine* e, const t2p_sampler_config& cfg);
  int init(const float* g_table_host, const int32_t* label_table_host);
  int set_condition(const uint8_t* mask, const float* x_initial) { mask_ = mask; x_init_ = x_initial; return T2P_OK; }
  void set_seed(uint64_t seed) { cfg_.seed = seed; }
  // global-batch Langevin step size (reference under DataParallel, sampling.py:193-195): `sums` is a caller-owned
  // device float[2] the norm sums of this process's chains are written to; `fn` must sum it over all processes
  // in stream order before returning control (e.g. one RCCL all_reduce enqueued on `stream`)
  int set_norm_allreduce(float* sums, t2p_allreduce_fn fn, void* user);
  // VP SDE (sde_lib.py:106-157) in the fused loop: per-step host tables of N floats (see t2p_sampler_set_vp_tables)
  int set_vp_tables(const float* label_f, const float* score_scale, const float* x_coef, const float* corr_alpha);
  int reset(int step, hipStream_t s);
  int step(float* x, float* x_mean, const float* nc, const float* np, hipStream_t s);
  // step() through a captured hipGraph (device noise only): first call with a given (x, x_mean,
  // condition) captures, later calls replay.  The step reads its index from the device counter.
  int step_graph(float* x, float* x_mean, hipStream_t s);
  int run(float* x, float* out, int prior_given, int n_steps, hipStream_t s);
  int count_dispatches(float* x, float* x_mean, hipStream_t s, int* n_out);
  ~Sampler();

 private:
  Engine* e_;
  t2p_sampler_config cfg_;
  const uint8_t* mask_ = nullptr;
  const float* x_init_ = nullptr;
  int* step_dev_ = nullptr;
  float* g_table_ = nullptr;
  float* score_ = nullptr;
  float* noise_ = nullptr;
  float* sq_ws_ = nullptr;
  float* sums_ = nullptr;
  float* xmean_ = nullptr;
  int* label_table_ = nullptr;     // device int[N]: time label of loop step i
  float* vp_label_f_ = nullptr;    // VP: device float[N] each
  float* vp_score_scale_ = nullptr;
  float* vp_x_coef_ = nullptr;
  float* vp_alpha_ = nullptr;
  int host_step_ = 0;              // host mirror of *step_dev_ (bounds check: the tables have N entries)
  float* sums_ext_ = nullptr;
  t2p_allreduce_fn allreduce_ = nullptr;
  void* allreduce_user_ = nullptr;
  uint64_t graph_seed_ = 0;
  long n_ = 0, per_sample_ = 0;
  hipGraphExec_t graph_exec_ = nullptr;
  float* graph_x_ = nullptr;
  float* graph_xm_ = nullptr;
  const uint8_t* graph_mask_ = nullptr;
  int eager_steps_ = 0;
};

}  // namespace t2p

struct t2p_engine { t2p::Engine impl; explicit t2p_engine(const t2p_model_config& c) : impl(c) {} };
struct t2p_sampler { t2p::Sampler impl; t2p_sampler(t2p::Engine* e, const t2p_sampler_config& c) : impl(e, c) {} };
