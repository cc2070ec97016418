// Host side of the score network (reference UNetModel, ncsnpp.py:71-263) and of the
// predictor-corrector loop (sampling.py:245-289): builds the layer table from the flat config,
// prepares weights (layout + dtype), and enqueues the HIP kernels of one evaluation on a stream.
// No host synchronisation happens inside score()/step(): the Langevin step size is computed on
// the device from device-side norm sums.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <thread>

namespace t2p {

bool g_fuse_gn_stats = true;
bool g_fuse_geglu = true;
bool g_lowp_h1 = true;
bool g_gn_small = true;
bool g_lowp_residual = true;
bool g_raw_copies = true;
bool g_flash_attention = true;
bool g_attn_merged = true;    // AttnBlockpp: NIN_2 . NIN_3 as one projection, output epilogue in the attention kernel (plan switch 32)
bool g_ffpo_merged = true;    // SpatialTransformer: ff.net.2 and proj_out as one GEMM over [g | t] (plan switch 33)
bool g_qkv_fused = true;      // self-attention: one q | k | v projection, V read row-major by the fused kernel (plan switch 25)
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* get_last_error() { return g_last_error.c_str(); }

int ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;     // (kernel, device) -> largest size set so far
  int dev = 0;
  T2P_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({kernel, dev});
  if (it != done.end() && it->second >= bytes) return T2P_OK;
  T2P_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done[{kernel, dev}] = bytes;
  return T2P_OK;
}

// ---- development tap (t2p_debug_tap): the output of block number `g_tap_index` of the next evaluations, widened to fp32 ----
static int g_tap_index = -1;
static float* g_tap_dst = nullptr;
static long g_tap_cap = 0;
static long g_tap_shape[4] = {0, 0, 0, 0};     // C, H, W, stored in 16 bits
static int g_tap_counter = 0;
void debug_tap_set(int index, float* dst, long capacity) { g_tap_index = index; g_tap_dst = dst; g_tap_cap = capacity; }
void debug_tap_shape(long out[4]) { for (int i = 0; i < 4; ++i) out[i] = g_tap_shape[i]; }

// ---- per-layer timing (development / bench: t2p_profile_layers_*): HIP events on the launch stream at block boundaries ----
struct LayerRec { std::string label; hipEvent_t e0, e1; };
static bool g_layer_prof = false;
static std::vector<LayerRec> g_layer_recs;
static std::mutex g_layer_mu;      // the records are process-global (a measurement hook): engines on other threads append under the lock
void layer_profile_begin() {
  std::lock_guard<std::mutex> lk(g_layer_mu);
  for (LayerRec& r : g_layer_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_layer_recs.clear();
  g_layer_prof = true;
}
struct LayerScope {
  int idx = -1;
  hipStream_t s;
  LayerScope(const std::string& label, hipStream_t st) : s(st) {
    if (!g_layer_prof) return;
    LayerRec r; r.label = label;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, s);
    std::lock_guard<std::mutex> lk(g_layer_mu);
    idx = (int)g_layer_recs.size();
    g_layer_recs.push_back(r);
  }
  ~LayerScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_layer_mu);
    if (idx < (int)g_layer_recs.size()) (void)hipEventRecord(g_layer_recs[idx].e1, s);
  }
};
// CSV "label,ms" per record, in launch order
int layer_profile_end(std::string* out) {
  g_layer_prof = false;
  T2P_HIP_CHECK(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_layer_mu);
  out->clear();
  for (LayerRec& r : g_layer_recs) {
    float ms = 0.f;
    T2P_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
    char buf[64];
    std::snprintf(buf, sizeof buf, ",%.4f\n", ms);
    *out += r.label + buf;
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_layer_recs.clear();
  return T2P_OK;
}

// ------------------------------------------------------------------------------------------------
DevPool::~DevPool() {
  for (void* p : all_) (void)hipFree(p);
}
static inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
void* DevPool::get(size_t bytes) {
  bytes = round_up(std::max<size_t>(bytes, 256), 256);
  auto it = free_.find(bytes);
  if (it != free_.end()) {
    void* p = it->second;
    free_.erase(it);
    if (lease_depth_) leased_.insert(p);
    return p;
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    set_last_error("hipMalloc of " + std::to_string(bytes) + " bytes failed");
    return nullptr;
  }
  size_of_[p] = bytes;
  all_.push_back(p);
  held_ += bytes;
  if (lease_depth_) leased_.insert(p);
  return p;
}
void DevPool::put(void* p) {
  if (!p) return;
  auto it = size_of_.find(p);
  if (it == size_of_.end()) return;
  leased_.erase(p);
  free_.emplace(it->second, p);
}
void DevPool::lease_begin() {
  if (lease_depth_++ == 0) leased_.clear();
}
int DevPool::lease_end() {
  if (--lease_depth_ > 0) return 0;
  lease_depth_ = 0;
  const int n = (int)leased_.size();
  for (void* p : leased_) {
    auto it = size_of_.find(p);
    if (it != size_of_.end()) free_.emplace(it->second, p);
  }
  leased_.clear();
  last_reclaimed_ = n;
  return n;
}
void* DevPool::persistent(size_t bytes) {
  bytes = round_up(std::max<size_t>(bytes, 256), 256);
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    set_last_error("hipMalloc of " + std::to_string(bytes) + " bytes failed");
    return nullptr;
  }
  all_.push_back(p);
  held_ += bytes;
  return p;
}

#define POOL_GET(var, type, bytes)                          \
  type var = (type)pool_.get(bytes);                        \
  if (!var) return T2P_ERR_HIP;

// ------------------------------------------------------------------------------------------------
static int gn_groups(int c) { return std::min(c / 4, 32); }  // layers.py:282

Engine::Engine(const t2p_model_config& cfg) : cfg_(cfg) {}
Engine::~Engine() {}

static void add(std::vector<ParamInfo>& v, const std::string& n, std::vector<int64_t> s) { v.push_back({n, std::move(s)}); }

static void res_params(std::vector<ParamInfo>& v, const Layer& l, int td) {
  const std::string& p = l.prefix;
  const int64_t ci = l.in_ch, co = l.out_ch;
  add(v, p + ".GroupNorm_0.weight", {ci}); add(v, p + ".GroupNorm_0.bias", {ci});
  add(v, p + ".Conv_0.weight", {co, ci, 3, 3}); add(v, p + ".Conv_0.bias", {co});
  add(v, p + ".Dense_0.weight", {co, td}); add(v, p + ".Dense_0.bias", {co});
  add(v, p + ".GroupNorm_1.weight", {co}); add(v, p + ".GroupNorm_1.bias", {co});
  add(v, p + ".Conv_1.weight", {co, co, 3, 3}); add(v, p + ".Conv_1.bias", {co});
  if (l.has_conv2) { add(v, p + ".Conv_2.weight", {co, ci, 1, 1}); add(v, p + ".Conv_2.bias", {co}); }
}
static void attn_params(std::vector<ParamInfo>& v, const Layer& l) {
  const int64_t c = l.in_ch;
  add(v, l.prefix + ".GroupNorm_0.weight", {c}); add(v, l.prefix + ".GroupNorm_0.bias", {c});
  for (int i = 0; i < 4; ++i) {
    add(v, l.prefix + ".NIN_" + std::to_string(i) + ".W", {c, c});
    add(v, l.prefix + ".NIN_" + std::to_string(i) + ".b", {c});
  }
}
static void st_params(std::vector<ParamInfo>& v, const Layer& l, int64_t ctx) {
  const std::string& p = l.prefix;
  const std::string t = p + ".transformer_blocks.0";
  const int64_t c = l.in_ch;
  add(v, p + ".norm.weight", {c}); add(v, p + ".norm.bias", {c});
  add(v, p + ".proj_in.weight", {c, c, 1, 1}); add(v, p + ".proj_in.bias", {c});
  add(v, t + ".attn1.to_q.weight", {c, c}); add(v, t + ".attn1.to_k.weight", {c, c});
  add(v, t + ".attn1.to_v.weight", {c, c}); add(v, t + ".attn1.to_out.0.weight", {c, c});
  add(v, t + ".attn1.to_out.0.bias", {c});
  add(v, t + ".ff.net.0.proj.weight", {8 * c, c}); add(v, t + ".ff.net.0.proj.bias", {8 * c});
  add(v, t + ".ff.net.2.weight", {c, 4 * c}); add(v, t + ".ff.net.2.bias", {c});
  add(v, t + ".attn2.to_q.weight", {c, c}); add(v, t + ".attn2.to_k.weight", {c, ctx});
  add(v, t + ".attn2.to_v.weight", {c, ctx}); add(v, t + ".attn2.to_out.0.weight", {c, c});
  add(v, t + ".attn2.to_out.0.bias", {c});
  for (int i = 1; i <= 3; ++i) {
    add(v, t + ".norm" + std::to_string(i) + ".weight", {c});
    add(v, t + ".norm" + std::to_string(i) + ".bias", {c});
  }
  add(v, p + ".proj_out.weight", {c, c, 1, 1}); add(v, p + ".proj_out.bias", {c});
}

int Engine::build() {
  const t2p_model_config& c = cfg_;
  T2P_REQUIRE(c.n_ch_mult >= 1 && c.n_ch_mult <= 8, "ch_mult length");
  T2P_REQUIRE(c.n_attn_resolutions >= 0 && c.n_attn_resolutions <= 8, "attn_resolutions length");
  T2P_REQUIRE(c.nf >= 8 && c.nf % 8 == 0, "nf must be a multiple of 8");
  T2P_REQUIRE(c.num_channels >= 1 && c.num_channels <= 8, "num_channels must be in [1, 8]");
  T2P_REQUIRE(c.max_res_num % (1 << (c.n_ch_mult - 1)) == 0, "max_res_num must be divisible by 2^(levels-1)");
  T2P_REQUIRE(c.n_heads >= 1 && c.context_dim >= 8 && c.context_dim % 8 == 0, "n_heads / context_dim");
  T2P_REQUIRE(c.num_scales >= 2 && c.sigma_min > 0 && c.sigma_max > c.sigma_min, "sigma schedule");
  T2P_REQUIRE(c.compute_dtype >= 0 && c.compute_dtype <= 2, "compute_dtype");
  nf_ = c.nf;
  temb_dim_ = 4 * nf_;
  cpad_ = 8;
  const int nres = c.n_ch_mult, nrb = c.num_res_blocks, L = c.max_res_num;
  auto in_attn = [&](int res) {
    for (int i = 0; i < c.n_attn_resolutions; ++i)
      if (c.attn_resolutions[i] == res) return true;
    return false;
  };
  auto mk = [&](int kind, const std::string& prefix, int ci, int co, int up, int down) {
    Layer l;
    l.kind = kind; l.prefix = prefix; l.in_ch = ci; l.out_ch = co; l.up = up; l.down = down;
    l.has_conv2 = (kind == 0) && (ci != co || up || down);
    if (kind == 0) { l.temb_off = temb_total_; temb_total_ += co; }
    return l;
  };
  auto attn_pair = [&](Stage& st, const std::string& prefix, int idx0, int ch) -> int {
    T2P_REQUIRE(ch % 32 == 0 && ch % c.n_heads == 0 && (ch / c.n_heads) % 8 == 0,
                "attention levels need channels % 32 == 0 and head dim % 8 == 0");
    st.layers.push_back(mk(1, prefix + "." + std::to_string(idx0), ch, ch, 0, 0));
    st.layers.push_back(mk(2, prefix + "." + std::to_string(idx0 + 1), ch, ch, 0, 0));
    return T2P_OK;
  };
  std::vector<int> skip_ch{nf_};
  int in_ch = nf_;
  for (int lvl = 0; lvl < nres; ++lvl) {
    const int res = L >> lvl;
    for (int b = 0; b < nrb; ++b) {
      const int out_ch = nf_ * c.ch_mult[lvl];
      const std::string prefix = "input_blocks." + std::to_string(input_stages_.size());
      Stage st;
      st.layers.push_back(mk(0, prefix + ".0", in_ch, out_ch, 0, 0));
      in_ch = out_ch;
      if (in_attn(res)) T2P_TRY(attn_pair(st, prefix, 1, in_ch));
      input_stages_.push_back(std::move(st));
      skip_ch.push_back(in_ch);
    }
    if (lvl != nres - 1) {
      const std::string prefix = "input_blocks." + std::to_string(input_stages_.size());
      Stage st;
      st.layers.push_back(mk(0, prefix + ".0", in_ch, in_ch, 0, 1));
      input_stages_.push_back(std::move(st));
      skip_ch.push_back(in_ch);
    }
  }
  const int mid = skip_ch.back();
  mid_stage_.layers.push_back(mk(0, "mid_blocks.0", mid, mid, 0, 0));
  T2P_TRY(attn_pair(mid_stage_, "mid_blocks", 1, mid));
  mid_stage_.layers.push_back(mk(0, "mid_blocks.3", mid, mid, 0, 0));
  in_ch = mid;
  for (int lvl = nres - 1; lvl >= 0; --lvl) {
    const int res = L >> lvl;
    for (int b = 0; b <= nrb; ++b) {
      const int out_ch = nf_ * c.ch_mult[lvl];
      const std::string prefix = "out_blocks." + std::to_string(out_stages_.size());
      Stage st;
      st.skip_ch = skip_ch.back();
      skip_ch.pop_back();
      st.layers.push_back(mk(0, prefix + ".0", in_ch + st.skip_ch, out_ch, 0, 0));
      in_ch = out_ch;
      if (in_attn(res)) T2P_TRY(attn_pair(st, prefix, 1, in_ch));
      if (lvl != 0 && b == nrb)
        st.layers.push_back(mk(0, prefix + "." + std::to_string(st.layers.size()), in_ch, in_ch, 1, 0));
      out_stages_.push_back(std::move(st));
    }
  }
  T2P_REQUIRE(skip_ch.empty(), "skip stack not consumed");
  final_ch_ = in_ch;

  // parameter table in reference parameters() order
  const int64_t td = temb_dim_, nf = nf_, ch = c.num_channels;
  add(params_, "pre_blocks.0.weight", {td, nf}); add(params_, "pre_blocks.0.bias", {td});
  add(params_, "pre_blocks.1.weight", {td, td}); add(params_, "pre_blocks.1.bias", {td});
  add(params_, "pre_conv.weight", {nf, ch, 3, 3}); add(params_, "pre_conv.bias", {nf});
  auto walk = [&](Stage& st) {
    for (Layer& l : st.layers) {
      if (l.kind == 0) res_params(params_, l, temb_dim_);
      else if (l.kind == 1) attn_params(params_, l);
      else st_params(params_, l, c.context_dim);
    }
  };
  for (Stage& st : input_stages_) walk(st);
  walk(mid_stage_);
  for (Stage& st : out_stages_) walk(st);
  add(params_, "out.0.weight", {final_ch_}); add(params_, "out.0.bias", {final_ch_});
  add(params_, "out.2.weight", {ch, final_ch_, 3, 3}); add(params_, "out.2.bias", {ch});
  return T2P_OK;
}

int Engine::load_param(const char* name, const float* data, const int64_t* shape, int ndim) {
  T2P_REQUIRE(name && data && shape && ndim >= 1 && ndim <= 4, "load_param arguments");
  T2P_REQUIRE(!finalized_, "engine already finalized");
  std::string n(name);
  if (n.rfind("module.", 0) == 0) n = n.substr(7);   // DataParallel state dict (score_sde_pytorch/utils.py:8)
  if (n == "sigmas") return T2P_OK;                   // float64 buffer: derived from the config here
  const ParamInfo* info = nullptr;
  for (const ParamInfo& p : params_)
    if (p.name == n) { info = &p; break; }
  if (!info) {   // load_state_dict(strict=False) ignores unknown keys (score_sde_pytorch/utils.py:14)
    return T2P_OK;
  }
  std::vector<int64_t> s(shape, shape + ndim);
  if (s != info->shape) {
    set_last_error("shape mismatch for " + n);
    return T2P_ERR_INVALID;
  }
  int64_t numel = 1;
  for (int64_t d : s) numel *= d;
  HostTensor& t = host_[n];
  t.shape = s;
  t.data.assign(data, data + numel);
  return T2P_OK;
}

const HostTensor* Engine::host(const std::string& name, std::vector<int64_t> shape) {
  auto it = host_.find(name);
  if (it == host_.end()) {
    set_last_error("parameter not loaded: " + name);
    return nullptr;
  }
  if (it->second.shape != shape) {
    set_last_error("unexpected shape for " + name);
    return nullptr;
  }
  return &it->second;
}

int Engine::upload_f32(const std::vector<float>& v, float** out) {
  float* d = (float*)pool_.persistent(v.size() * 4);
  if (!d) return T2P_ERR_HIP;
  T2P_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  *out = d;
  return T2P_OK;
}

static int upload_matrix(DevPool& pool, const std::vector<float>& m, int dtype, void** out) {
  const size_t n = m.size();
  void* d = pool.persistent(n * dtype_size(dtype));
  if (!d) return T2P_ERR_HIP;
  if (dtype == DT_F32) {
    T2P_HIP_CHECK(hipMemcpy(d, m.data(), n * 4, hipMemcpyHostToDevice));
  } else {
    std::vector<uint16_t> h(n);
    if (dtype == DT_BF16) {
      for (size_t i = 0; i < n; ++i) h[i] = f32_to_bf16_bits(m[i]);
    } else {
      for (size_t i = 0; i < n; ++i) {
        _Float16 x = (_Float16)m[i];
        std::memcpy(&h[i], &x, 2);
      }
    }
    T2P_HIP_CHECK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  }
  *out = d;
  return T2P_OK;
}

// [N][K] matrix in the layout the GEMM wants, from a reference tensor:
//   Linear / 1x1 conv : (N, K[,1,1])  as is          NIN : (K, N) transposed
//   3x3 conv          : (N, Cin, 3, 3) -> [N][kh*3+kw][Cin (padded to cin_pad)]
static std::vector<float> to_nk(const HostTensor& t, bool conv3x3, bool nin, int cin_pad = 0) {
  if (conv3x3) {
    const int64_t N = t.shape[0], ci = t.shape[1];
    const int64_t cp = cin_pad ? cin_pad : ci;
    std::vector<float> m((size_t)N * 9 * cp, 0.f);
    for (int64_t n = 0; n < N; ++n)
      for (int64_t c = 0; c < ci; ++c)
        for (int tap = 0; tap < 9; ++tap) m[(n * 9 + tap) * cp + c] = t.data[(n * ci + c) * 9 + tap];
    return m;
  }
  if (nin) {
    const int64_t K = t.shape[0], N = t.shape[1];
    std::vector<float> m((size_t)N * K);
    for (int64_t k = 0; k < K; ++k)
      for (int64_t n = 0; n < N; ++n) m[n * K + k] = t.data[k * N + n];
    return m;
  }
  return t.data;
}

// out[M][N] = A[M][K] B[K][N] (row-major fp32; weight products formed once at load time), rows spread over a few threads
static std::vector<float> host_matmul(const float* A, const float* B, int64_t M, int64_t K, int64_t N) {
  std::vector<float> out((size_t)M * N, 0.f);
  const int nt = (int)std::min<int64_t>(8, std::max<int64_t>(1, M / 16));
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (int64_t m = M * t / nt; m < M * (t + 1) / nt; ++m) {
        float* o = out.data() + m * N;
        for (int64_t k = 0; k < K; ++k) {
          const float a = A[m * K + k];
          const float* b = B + k * N;
          for (int64_t n = 0; n < N; ++n) o[n] += a * b[n];
        }
      }
    });
  for (auto& x : th) x.join();
  return out;
}

int Engine::upload_linear(const std::string& wname, const std::string& bname, int N, int K, DevLinear* out,
                          bool conv3x3, bool nin, int force_dtype) {
  std::vector<int64_t> shape;
  int cin_pad = 0;
  if (conv3x3) {
    const int ci = K / 9;
    shape = {N, ci, 3, 3};
    if (wname == "pre_conv.weight") { shape = {N, cfg_.num_channels, 3, 3}; cin_pad = cpad_; }
  } else if (nin) {
    shape = {K, N};
  } else {
    shape = {N, K};
  }
  const HostTensor* w = host(wname, shape);
  if (!w && !conv3x3 && !nin) {   // 1x1 convolution weights are (N, K, 1, 1)
    w = host(wname, {N, K, 1, 1});
  }
  if (!w) return T2P_ERR_STATE;
  std::vector<float> m = to_nk(*w, conv3x3, nin, cin_pad);
  T2P_TRY(upload_matrix(pool_, m, force_dtype >= 0 ? force_dtype : cfg_.compute_dtype, &out->w));
  out->N = N;
  out->K = (int)(m.size() / N);
  out->b = nullptr;
  if (!bname.empty()) {
    const HostTensor* b = host(bname, {N});
    if (!b) return T2P_ERR_STATE;
    T2P_TRY(upload_f32(b->data, &out->b));
  }
  return T2P_OK;
}

int Engine::upload_stack2(const std::string& w0, const std::string& b0, const std::string& w1, const std::string& b1,
                          int N, int K, bool nin, bool has_bias, DevLinear* out) {
  const std::vector<int64_t> shape = nin ? std::vector<int64_t>{K, N} : std::vector<int64_t>{N, K};
  const HostTensor* t0 = host(w0, shape);
  const HostTensor* t1 = host(w1, shape);
  if (!t0 || !t1) return T2P_ERR_STATE;
  std::vector<float> m = to_nk(*t0, false, nin);
  std::vector<float> m1 = to_nk(*t1, false, nin);
  m.insert(m.end(), m1.begin(), m1.end());
  T2P_TRY(upload_matrix(pool_, m, cfg_.compute_dtype, &out->w));
  out->N = 2 * N;
  out->K = K;
  out->b = nullptr;
  if (has_bias) {
    const HostTensor* c0 = host(b0, {N});
    const HostTensor* c1 = host(b1, {N});
    if (!c0 || !c1) return T2P_ERR_STATE;
    std::vector<float> b = c0->data;
    b.insert(b.end(), c1->data.begin(), c1->data.end());
    T2P_TRY(upload_f32(b, &out->b));
  }
  return T2P_OK;
}

int Engine::upload_norm(const std::string& prefix, int C, int G, DevNorm* out) {
  const HostTensor* g = host(prefix + ".weight", {C});
  const HostTensor* b = host(prefix + ".bias", {C});
  if (!g || !b) return T2P_ERR_STATE;
  T2P_TRY(upload_f32(g->data, &out->gamma));
  T2P_TRY(upload_f32(b->data, &out->beta));
  out->C = C;
  out->G = G;
  return T2P_OK;
}

int Engine::finalize() {
  T2P_REQUIRE(!finalized_, "engine already finalized");
  for (const ParamInfo& p : params_) {
    if (!host_.count(p.name)) {
      set_last_error("parameter not loaded: " + p.name);
      return T2P_ERR_STATE;
    }
  }
  const int td = temb_dim_;
  T2P_TRY(upload_linear("pre_blocks.0.weight", "pre_blocks.0.bias", td, nf_, &pre0_, false, false, DT_F32));
  T2P_TRY(upload_linear("pre_blocks.1.weight", "pre_blocks.1.bias", td, td, &pre1_, false, false, DT_F32));
  T2P_TRY(upload_linear("pre_conv.weight", "pre_conv.bias", nf_, 9 * cfg_.num_channels, &pre_conv_, true, false, DT_F32));
  {
    const HostTensor* w = host("pre_conv.weight", {nf_, cfg_.num_channels, 3, 3});
    if (!w) return T2P_ERR_STATE;
    std::vector<float> m = to_nk(*w, true, false, 0);      // [nf][tap][C], unpadded
    const int K9 = 9 * cfg_.num_channels;
    std::vector<float> mt((size_t)K9 * nf_);                // [tap][C][nf]: lanes (= output channels) read contiguously
    for (int n = 0; n < nf_; ++n)
      for (int k = 0; k < K9; ++k) mt[(size_t)k * nf_ + n] = m[(size_t)n * K9 + k];
    T2P_TRY(upload_f32(mt, &pre_conv_direct_));
    // the same weights split into two f16 terms each, for the input convolution on the 16-bit matrix pipe (pre_conv_split_kernel).
    // The split form needs |x| inside the f16 range: the state of a VE run stays within a few sigma_max
    const int C = cfg_.num_channels;
    if (dtype() != DT_F32 && (C == 5 || C == 8) && cfg_.sigma_max <= 4096.0) {
      pre_conv_split_ = pool_.persistent(pre_conv_split_weight_bytes(C, nf_));
      if (!pre_conv_split_) return T2P_ERR_HIP;
      T2P_TRY(launch_pre_conv_split_weights(pre_conv_direct_, pre_conv_split_, C, nf_, nullptr));
      T2P_HIP_CHECK(hipStreamSynchronize(nullptr));
    }
  }
  std::vector<float> dw((size_t)temb_total_ * td), db(temb_total_);
  auto each_layer = [&](auto&& fn) -> int {
    for (Stage& st : input_stages_) for (Layer& l : st.layers) T2P_TRY(fn(l));
    for (Layer& l : mid_stage_.layers) T2P_TRY(fn(l));
    for (Stage& st : out_stages_) for (Layer& l : st.layers) T2P_TRY(fn(l));
    return T2P_OK;
  };
  T2P_TRY(each_layer([&](Layer& l) -> int {
    const std::string& p = l.prefix;
    const int ci = l.in_ch, co = l.out_ch;
    if (l.kind == 0) {
      T2P_TRY(upload_norm(p + ".GroupNorm_0", ci, gn_groups(ci), &l.gn0));
      T2P_TRY(upload_norm(p + ".GroupNorm_1", co, gn_groups(co), &l.gn1));
      T2P_TRY(upload_linear(p + ".Conv_0.weight", p + ".Conv_0.bias", co, 9 * ci, &l.conv0, true));
      if (l.up && cfg_.compute_dtype != DT_F32) {
        // conv0 of an up block reads the 2x nearest-up-sampled map (layers.py:306-308): output phase (py, px) of pixel
        // (2 yl + py, 2 xl + px) sees source pixel yl + floor((py + kh - 1) / 2) for kernel row kh, i.e. two source rows per
        // phase; the taps that read the same source pixel are summed here (in fp32, before the one rounding to the compute
        // dtype): four 2x2 convolutions on the source map with 4 / 9 of the multiplications
        const HostTensor* w = host(p + ".Conv_0.weight", {co, ci, 3, 3});
        if (!w) return T2P_ERR_STATE;
        std::vector<float> w4((size_t)4 * co * 4 * ci, 0.f);
        auto fl2 = [](int f) { return f >= 0 ? f / 2 : -((-f + 1) / 2); };
        for (int py = 0; py < 2; ++py)
          for (int px = 0; px < 2; ++px)
            for (int kh = 0; kh < 3; ++kh)
              for (int kw = 0; kw < 3; ++kw) {
                const int ty = fl2(py + kh - 1) + 1 - py, tx = fl2(px + kw - 1) + 1 - px;     // 0 or 1
                const size_t ph = (size_t)(py * 2 + px), tap = (size_t)(ty * 2 + tx);
                for (int n = 0; n < co; ++n)
                  for (int c = 0; c < ci; ++c)
                    w4[((ph * co + n) * 4 + tap) * ci + c] += w->data[(((size_t)n * ci + c) * 3 + kh) * 3 + kw];
              }
        T2P_TRY(upload_matrix(pool_, w4, cfg_.compute_dtype, &l.conv0_up4));
      }
      T2P_TRY(upload_linear(p + ".Conv_1.weight", p + ".Conv_1.bias", co, 9 * co, &l.conv1, true));
      if (l.has_conv2) T2P_TRY(upload_linear(p + ".Conv_2.weight", p + ".Conv_2.bias", co, ci, &l.conv2));
      if (l.has_conv2 && !l.up && cfg_.compute_dtype != DT_F32 && ci % 64 == 0 && co % 64 == 0) {
        // h + shortcut(x) = [conv1 | conv2] applied to [3x3 window of a1 | x] (layers.py:322-327): one K loop, one fp32 sum
        const HostTensor* w1 = host(p + ".Conv_1.weight", {co, co, 3, 3});
        const HostTensor* w2 = host(p + ".Conv_2.weight", {co, ci, 1, 1});
        const HostTensor* b1 = host(p + ".Conv_1.bias", {co});
        const HostTensor* b2 = host(p + ".Conv_2.bias", {co});
        if (!w1 || !w2 || !b1 || !b2) return T2P_ERR_STATE;
        const std::vector<float> m1 = to_nk(*w1, true, false, 0);
        const size_t K1 = (size_t)9 * co, Kx = K1 + ci;
        std::vector<float> m((size_t)co * Kx), bb(co);
        for (int n = 0; n < co; ++n) {
          std::copy(m1.begin() + n * K1, m1.begin() + (n + 1) * K1, m.begin() + n * Kx);
          std::copy(w2->data.begin() + (size_t)n * ci, w2->data.begin() + (size_t)(n + 1) * ci, m.begin() + n * Kx + K1);
          bb[n] = b1->data[n] + b2->data[n];
        }
        T2P_TRY(upload_matrix(pool_, m, cfg_.compute_dtype, &l.conv1x.w));
        T2P_TRY(upload_f32(bb, &l.conv1x.b));
        l.conv1x.N = co; l.conv1x.K = (int)Kx;
      }
      const HostTensor* w = host(p + ".Dense_0.weight", {co, td});
      const HostTensor* b = host(p + ".Dense_0.bias", {co});
      if (!w || !b) return T2P_ERR_STATE;
      std::copy(w->data.begin(), w->data.end(), dw.begin() + (size_t)l.temb_off * td);
      std::copy(b->data.begin(), b->data.end(), db.begin() + l.temb_off);
    } else if (l.kind == 1) {
      T2P_TRY(upload_norm(p + ".GroupNorm_0", ci, gn_groups(ci), &l.gn0));
      T2P_TRY(upload_stack2(p + ".NIN_0.W", p + ".NIN_0.b", p + ".NIN_1.W", p + ".NIN_1.b", ci, ci, true, true, &l.qk));
      T2P_TRY(upload_linear(p + ".NIN_2.W", p + ".NIN_2.b", ci, ci, &l.v, false, true));
      T2P_TRY(upload_linear(p + ".NIN_3.W", p + ".NIN_3.b", ci, ci, &l.out, false, true));
      if (cfg_.compute_dtype != DT_F32) {
        const HostTensor* w2 = host(p + ".NIN_2.W", {ci, ci});      // NIN: [in][out]
        const HostTensor* w3 = host(p + ".NIN_3.W", {ci, ci});
        const HostTensor* b2 = host(p + ".NIN_2.b", {ci});
        const HostTensor* b3 = host(p + ".NIN_3.b", {ci});
        if (!w2 || !w3 || !b2 || !b3) return T2P_ERR_STATE;
        HostTensor m;
        m.shape = {ci, ci};
        m.data = host_matmul(w2->data.data(), w3->data.data(), ci, ci, ci);       // [in][out]
        std::vector<float> bb = host_matmul(b2->data.data(), w3->data.data(), 1, ci, ci);
        for (int i = 0; i < ci; ++i) bb[i] += b3->data[i];
        T2P_TRY(upload_matrix(pool_, to_nk(m, false, true), cfg_.compute_dtype, &l.v3.w));
        T2P_TRY(upload_f32(bb, &l.v3.b));
        l.v3.N = ci; l.v3.K = ci;
        if (ci == 256 && l.qk.w && l.qk.b) {
          l.fm_qk = pool_.persistent((size_t)2 * ci * ci * 2);
          l.fm_v3 = pool_.persistent((size_t)ci * ci * 2);
          if (!l.fm_qk || !l.fm_v3) return T2P_ERR_HIP;
          T2P_TRY(launch_sf_frag_major(cfg_.compute_dtype, l.qk.w, l.fm_qk, 2 * ci, ci, nullptr));
          T2P_TRY(launch_sf_frag_major(cfg_.compute_dtype, l.v3.w, l.fm_v3, ci, ci, nullptr));
          T2P_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
      }
    } else {
      const std::string t = p + ".transformer_blocks.0";
      const int ctx = cfg_.context_dim;
      T2P_TRY(upload_norm(p + ".norm", ci, 32, &l.gn0));
      T2P_TRY(upload_linear(p + ".proj_in.weight", p + ".proj_in.bias", ci, ci, &l.proj_in));
      T2P_TRY(upload_linear(p + ".proj_out.weight", p + ".proj_out.bias", ci, ci, &l.proj_out));
      T2P_TRY(upload_stack2(t + ".attn1.to_q.weight", "", t + ".attn1.to_k.weight", "", ci, ci, false, false, &l.a1_qk));
      T2P_TRY(upload_linear(t + ".attn1.to_v.weight", "", ci, ci, &l.a1_v));
      if (cfg_.compute_dtype != DT_F32) {      // the fused attention kernel reads V row-major: q | k | v from one GEMM
        const HostTensor* wq = host(t + ".attn1.to_q.weight", {ci, ci});
        const HostTensor* wk = host(t + ".attn1.to_k.weight", {ci, ci});
        const HostTensor* wv = host(t + ".attn1.to_v.weight", {ci, ci});
        if (!wq || !wk || !wv) return T2P_ERR_STATE;
        std::vector<float> m = wq->data;
        m.insert(m.end(), wk->data.begin(), wk->data.end());
        m.insert(m.end(), wv->data.begin(), wv->data.end());
        T2P_TRY(upload_matrix(pool_, m, cfg_.compute_dtype, &l.a1_qkv.w));
        l.a1_qkv.N = 3 * ci; l.a1_qkv.K = ci; l.a1_qkv.b = nullptr;
      }
      T2P_TRY(upload_linear(t + ".attn1.to_out.0.weight", t + ".attn1.to_out.0.bias", ci, ci, &l.a1_out));
      T2P_TRY(upload_linear(t + ".attn2.to_q.weight", "", ci, ci, &l.a2_q));
      T2P_TRY(upload_linear(t + ".attn2.to_k.weight", "", ci, ctx, &l.a2_k));
      T2P_TRY(upload_linear(t + ".attn2.to_v.weight", "", ci, ctx, &l.a2_v));
      T2P_TRY(upload_linear(t + ".attn2.to_out.0.weight", t + ".attn2.to_out.0.bias", ci, ci, &l.a2_out));
      {   // GEGLU projection with rows interleaved (value_j, gate_j) so the GEMM epilogue can gate in registers
        const HostTensor* w = host(t + ".ff.net.0.proj.weight", {8 * ci, ci});
        const HostTensor* b = host(t + ".ff.net.0.proj.bias", {8 * ci});
        if (!w || !b) return T2P_ERR_STATE;
        const int inner = 4 * ci;
        std::vector<float> wi((size_t)8 * ci * ci), bi((size_t)8 * ci);
        for (int j = 0; j < inner; ++j) {
          std::copy(w->data.begin() + (size_t)j * ci, w->data.begin() + (size_t)(j + 1) * ci, wi.begin() + (size_t)(2 * j) * ci);
          std::copy(w->data.begin() + (size_t)(inner + j) * ci, w->data.begin() + (size_t)(inner + j + 1) * ci,
                    wi.begin() + (size_t)(2 * j + 1) * ci);
          bi[2 * j] = b->data[j];
          bi[2 * j + 1] = b->data[inner + j];
        }
        T2P_TRY(upload_matrix(pool_, wi, cfg_.compute_dtype, &l.ff1.w));
        T2P_TRY(upload_f32(bi, &l.ff1.b));
        l.ff1.N = 8 * ci;
        l.ff1.K = ci;
      }
      T2P_TRY(upload_linear(t + ".ff.net.2.weight", t + ".ff.net.2.bias", ci, 4 * ci, &l.ff2));
      // (C = 512: the chains stream 4x the bytes per workgroup and measured equal to the separate launches: not built)
      if (cfg_.compute_dtype != DT_F32 && ci == 256 && l.a1_qkv.w) {
        // fragment-major copies of the row-chain kernel's weights (1 KiB contiguous per MFMA fragment)
        auto fm = [&](const void* w, int N, void** out) -> int {
          *out = pool_.persistent((size_t)N * ci * 2);
          if (!*out) return T2P_ERR_HIP;
          return launch_sf_frag_major(cfg_.compute_dtype, w, *out, N, ci, nullptr);
        };
        T2P_TRY(fm(l.proj_in.w, ci, &l.fm_in));
        T2P_TRY(fm(l.a1_qkv.w, 3 * ci, &l.fm_qkv));
        T2P_TRY(fm(l.a1_out.w, ci, &l.fm_out1));
        T2P_TRY(fm(l.a2_q.w, ci, &l.fm_q2));
        if (ci == 256) {                        // (the chains through the feed-forward exist at C = 256 only)
          T2P_TRY(fm(l.a2_out.w, ci, &l.fm_out2));
          T2P_TRY(fm(l.ff1.w, 8 * ci, &l.fm_ff1));
        }
        T2P_HIP_CHECK(hipStreamSynchronize(nullptr));
      }
      if (cfg_.compute_dtype != DT_F32 && ci % 64 == 0) {
        const HostTensor* wpo = host(p + ".proj_out.weight", {ci, ci, 1, 1});
        const HostTensor* bpo = host(p + ".proj_out.bias", {ci});
        const HostTensor* wf = host(t + ".ff.net.2.weight", {ci, 4 * ci});
        const HostTensor* bf = host(t + ".ff.net.2.bias", {ci});
        if (!wpo || !bpo || !wf || !bf) return T2P_ERR_STATE;
        const std::vector<float> prod = host_matmul(wpo->data.data(), wf->data.data(), ci, ci, 4 * ci);    // [C][4 C]
        std::vector<float> m((size_t)ci * 5 * ci), bb(ci);
        for (int n = 0; n < ci; ++n) {
          std::copy(prod.begin() + (size_t)n * 4 * ci, prod.begin() + (size_t)(n + 1) * 4 * ci, m.begin() + (size_t)n * 5 * ci);
          std::copy(wpo->data.begin() + (size_t)n * ci, wpo->data.begin() + (size_t)(n + 1) * ci, m.begin() + (size_t)n * 5 * ci + 4 * ci);
          double acc = bpo->data[n];
          for (int j = 0; j < ci; ++j) acc += (double)wpo->data[(size_t)n * ci + j] * bf->data[j];
          bb[n] = (float)acc;
        }
        T2P_TRY(upload_matrix(pool_, m, cfg_.compute_dtype, &l.ffpo.w));
        T2P_TRY(upload_f32(bb, &l.ffpo.b));
        l.ffpo.N = ci; l.ffpo.K = 5 * ci;
        if (ci == 256 && l.fm_in) {          // fragment-major copy for the row-chain kernel's third product
          l.fm_ffpo = pool_.persistent((size_t)ci * 5 * ci * 2);
          if (!l.fm_ffpo) return T2P_ERR_HIP;
          T2P_TRY(launch_sf_frag_major(cfg_.compute_dtype, l.ffpo.w, l.fm_ffpo, ci, 5 * ci, nullptr));
          T2P_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
      }
      T2P_TRY(upload_norm(t + ".norm1", ci, 1, &l.ln1));
      T2P_TRY(upload_norm(t + ".norm2", ci, 1, &l.ln2));
      T2P_TRY(upload_norm(t + ".norm3", ci, 1, &l.ln3));
    }
    return T2P_OK;
  }));
  void* dwp = nullptr;
  T2P_TRY(upload_matrix(pool_, dw, DT_F32, &dwp));
  dense_all_.w = dwp;
  dense_all_.N = temb_total_;
  dense_all_.K = td;
  T2P_TRY(upload_f32(db, &dense_all_.b));
  T2P_TRY(upload_norm("out.0", final_ch_, gn_groups(final_ch_), &head_norm_));
  T2P_TRY(upload_linear("out.2.weight", "out.2.bias", cfg_.num_channels, 9 * final_ch_, &head_conv_, true));

  // 1 / sigmas[label]: sigmas = exp(linspace(log sigma_max, log sigma_min, N)) in float64
  // (models/utils.py:50-60); the reference divides by the float64 value (ncsnpp.py:259-261).
  const int N = cfg_.num_scales;
  std::vector<float> inv(N);
  const double a = std::log(cfg_.sigma_max), b = std::log(cfg_.sigma_min);
  for (int i = 0; i < N; ++i) {
    const double s = std::exp(a + (b - a) * (double)i / (double)(N - 1));
    inv[i] = (float)(1.0 / s);
  }
  T2P_TRY(upload_f32(inv, &inv_sigma_));
  host_.clear();
  finalized_ = true;
  return T2P_OK;
}

// ------------------------------------------------------------------------------------------------
// every GEMM of the engine: lends the split-K workspace (fp32 partial tiles of low-resolution levels)
int Engine::attach_ws(GemmParams& p) {
  if (!splitk_ws_) {
    splitk_ws_bytes_ = (size_t)256 << 20;
    splitk_ws_ = pool_.persistent(splitk_ws_bytes_);
    if (!splitk_ws_) return T2P_ERR_HIP;
  }
  p.ws = splitk_ws_;
  p.ws_bytes = splitk_ws_bytes_;
  return T2P_OK;
}

int Engine::gemm(GemmParams& p, hipStream_t s) {
  T2P_TRY(attach_ws(p));
  return launch_gemm(p, s);
}

int Engine::gemm_stats(GemmParams& p, float** cstats, hipStream_t s) {
  *cstats = nullptr;
  T2P_TRY(attach_ws(p));          // the fuse predicate depends on the split-K decision
  if (g_fuse_gn_stats && (gemm_fuses_col_stats(p) || gemm_fuses_col_stats_lowp(p))) {
    *cstats = (float*)pool_.get((size_t)(p.M / 64) * p.N * 2 * 4);
    if (!*cstats) return T2P_ERR_HIP;
    p.col_stats = *cstats;
  }
  return gemm(p, s);
}

int Engine::linear(const void* a, bool a_is_f32, const DevLinear& w, long rows, void* c, bool c_f32,
                   const float* residual, float alpha, hipStream_t s, bool use_bias, float** cstats, bool r_lowp) {
  GemmParams p;
  p.dtype = dtype();
  p.A0 = a; p.a_f32 = a_is_f32 || p.dtype == DT_F32; p.C0 = w.K; p.lda0 = w.K;
  p.Bw = w.w; p.ldb = w.K;
  p.M = (int)rows; p.N = w.N;
  p.bias_n = use_bias ? w.b : nullptr;
  p.R = residual; p.ldr = w.N; p.r_lowp = (residual && r_lowp) ? 1 : 0;
  p.alpha = alpha;
  p.C = c; p.c_f32 = c_f32; p.ldc = w.N;
  if (cstats) return gemm_stats(p, cstats, s);
  return gemm(p, s);
}

int Engine::group_norm(const Act& x, const Act* x1, const DevNorm& n, float eps, int silu, int down, int B, void** out,
                       hipStream_t s, void** raw_out) {
  const int C = x.C + (x1 ? x1->C : 0);
  T2P_REQUIRE(C == n.C, "GroupNorm channel mismatch");
  T2P_REQUIRE(!x1 || x1->lowp == x.lowp, "concatenated sources must share a storage type");
  T2P_REQUIRE(!x.lowp || dtype() != DT_F32, "16-bit activations exist in the 16-bit modes only");
  if (x.pre_norm && !x1 && x.pre_for == &n && x.pre_silu == silu && !down && !raw_out) {
    *out = x.pre_norm;              // the producing split-K second pass already applied this norm (GemmParams::gn_out)
    x.pre_norm = nullptr;           // ownership passes to the caller, which returns it to the pool like any norm output
    return T2P_OK;
  }
  {
    // small maps (<= 64 pixels): statistics + apply in one launch instead of two or three latency-bound ones
    GroupNormApplyArgs g;
    g.x0 = (const float*)x.p; g.x1 = x1 ? (const float*)x1->p : nullptr; g.C0 = x.C; g.C1 = x1 ? x1->C : 0;
    g.B = B; g.H = x.H; g.W = x.W; g.G = n.G; g.gamma = n.gamma; g.beta = n.beta; g.silu = silu; g.down = down;
    g.dtype = dtype(); g.x0_lowp = x.lowp; g.eps = eps;
    if (g_gn_small && gn_small_eligible(g)) {
      const size_t bytes = (size_t)B * x.H * x.W * C * dtype_size(dtype());
      POOL_GET(o, void*, bytes);
      g.out = o;
      if (raw_out) {
        *raw_out = pool_.get(bytes);
        if (!*raw_out) return T2P_ERR_HIP;
        g.raw_out = *raw_out;
      }
      T2P_TRY(launch_gn_small(g, s));
      *out = o;
      return T2P_OK;
    }
  }
  GroupNormArgs a;
  a.x0 = x.p; a.x1 = x1 ? x1->p : nullptr; a.C0 = x.C; a.C1 = x1 ? x1->C : 0;
  a.B = B; a.HW = x.H * x.W; a.G = n.G; a.eps = eps;
  const bool have_cols = x.cstats && (!x1 || x1->cstats) && a.HW % 64 == 0;
  if (have_cols && !raw_out) {
    // maps of <= 4096 pixels: fold the column statistics and normalise in one launch
    GroupNormApplyArgs g;
    g.x0 = a.x0; g.x1 = a.x1; g.C0 = a.C0; g.C1 = a.C1; g.B = B; g.H = x.H; g.W = x.W; g.G = n.G;
    g.gamma = n.gamma; g.beta = n.beta; g.silu = silu; g.down = down; g.dtype = dtype(); g.x0_lowp = x.lowp;
    g.cs0 = x.cstats; g.cs1 = x1 ? x1->cstats : nullptr; g.eps = eps;
    if (gn_apply_cols_eligible(g)) {
      POOL_GET(o, void*, (size_t)B * a.HW * C * dtype_size(dtype()));
      g.out = o;
      T2P_TRY(launch_gn_apply_cols(g, s));
      *out = o;
      return T2P_OK;
    }
  }
  const int nparts = gn_num_chunks(a.HW) * ((C + 1023) / 1024);
  POOL_GET(stats, float*, (size_t)B * n.G * 2 * 4);
  float* partial = nullptr;
  if (have_cols) {
    // statistics came with the producing GEMM's epilogue: no pass over the activation
    T2P_TRY(launch_gn_finalize_cols(x.cstats, x1 ? x1->cstats : nullptr, a.C0, a.C1, B, a.HW, n.G, eps, stats, s));
  } else {
    partial = (float*)pool_.get((size_t)B * nparts * n.G * 2 * 4);
    if (!partial) return T2P_ERR_HIP;
    a.partial = partial; a.stats = stats;
    a.lowp_dtype = x.lowp ? dtype() : DT_F32;
    T2P_TRY(launch_gn_stats(a, s));
  }
  GroupNormApplyArgs g;
  g.x0 = a.x0; g.x1 = a.x1; g.C0 = a.C0; g.C1 = a.C1; g.B = B; g.H = x.H; g.W = x.W; g.G = n.G;
  g.stats = stats; g.gamma = n.gamma; g.beta = n.beta; g.silu = silu; g.down = down; g.dtype = dtype();
  g.x0_lowp = x.lowp;
  const size_t opix = (size_t)B * (down ? x.H / 2 : x.H) * (down ? x.W / 2 : x.W);
  POOL_GET(o, void*, opix * C * dtype_size(dtype()));
  g.out = o;
  if (raw_out) {
    *raw_out = pool_.get(opix * C * dtype_size(dtype()));
    if (!*raw_out) return T2P_ERR_HIP;
    g.raw_out = *raw_out;
  }
  T2P_TRY(launch_gn_apply(g, s));
  pool_.put(partial);
  pool_.put(stats);
  *out = o;
  return T2P_OK;
}

// ResnetBlockBigGANpp.forward (layers.py:303-327)
int Engine::res_block(Layer& L, const Act& x, const Act* skip, Act* out, int B, hipStream_t s, NormHint hint) {
  const int Cin = x.C + (skip ? skip->C : 0), Cout = L.out_ch;
  T2P_REQUIRE(Cin == L.in_ch, "res block input channels");
  T2P_REQUIRE(!(L.down && skip), "down block with concat input");
  const int Ho = L.up ? x.H * 2 : (L.down ? x.H / 2 : x.H), Wo = L.up ? x.W * 2 : (L.down ? x.W / 2 : x.W);
  const long rows_out = (long)B * Ho * Wo;
  const int dt = dtype();
  void* a0 = nullptr;
  // the 1x1 shortcut reads the raw block input: in 16-bit modes GroupNorm-apply (which reads it
  // anyway) also emits it in the compute dtype, so the shortcut GEMM takes the LDS-DMA kernel
  void* xraw = nullptr;
  // (a 16-bit residual stream is already in the GEMM operand format: no raw copy)
  const bool want_raw = g_raw_copies && L.has_conv2 && !L.down && dt != DT_F32 && !x.lowp;
  T2P_TRY(group_norm(x, skip, L.gn0, 1e-6f, 1, L.down, B, &a0, s, want_raw ? &xraw : nullptr));
  // ---- 8x8 / 4x4 maps: each convolution with everything around it in ONE launch (small_conv_gn_kernel): a workgroup owns whole
  // samples x whole groups, so GroupNorm_1 + SiLU (conv0) and the first norm of the block that follows (conv1) need no second
  // pass; the 1x1 shortcut rides as an extra K segment, the identity shortcut as the residual
  if (dt != DT_F32 && !L.up && (Ho * Wo == 16 || Ho * Wo == 64) && x.lowp && (!skip || skip->lowp) && (L.has_conv2 ? L.conv1x.w != nullptr : true)) {
    SmallConvArgs c0;
    c0.dtype = dt; c0.A = a0; c0.B = B; c0.H = Ho; c0.W = Wo; c0.C = Cin; c0.N = Cout; c0.Wt = L.conv0.w; c0.ldw = L.conv0.K;
    c0.bias = L.conv0.b; c0.bias_bn = tb_ + L.temb_off; c0.ld_bn = tb_ld_;
    c0.gn_gamma = L.gn1.gamma; c0.gn_beta = L.gn1.beta; c0.groups = L.gn1.G; c0.gn_silu = 1; c0.gn_eps = 1e-6f;
    SmallConvArgs c1;
    c1.dtype = dt; c1.B = B; c1.H = Ho; c1.W = Wo; c1.C = Cout; c1.N = Cout;
    c1.alpha = cfg_.skip_rescale ? 0.70710678118654752440f : 1.f;
    if (L.has_conv2) {
      c1.Wt = L.conv1x.w; c1.ldw = L.conv1x.K; c1.bias = L.conv1x.b;
      c1.CX0 = L.down ? Cin : x.C; c1.CX1 = (!L.down && skip) ? skip->C : 0;
    } else {
      c1.Wt = L.conv1.w; c1.ldw = L.conv1.K; c1.bias = L.conv1.b;
    }
    const bool olp = res_lowp();
    if (hint.norm && olp && hint.norm->C == Cout) { c1.groups = hint.norm->G; c1.gn_silu = hint.silu; }
    // (eligibility is a matter of shapes: probe with placeholders for the buffers allocated below)
    c0.normed = a0; c1.A = a0; c1.out = a0; c1.normed = c1.groups ? a0 : nullptr;
    c1.gn_gamma = c1.groups ? hint.norm->gamma : nullptr; c1.gn_beta = c1.groups ? hint.norm->beta : nullptr;
    if (L.has_conv2) { c1.X0 = a0; c1.X1 = c1.CX1 ? a0 : nullptr; } else { c1.R = a0; }
    if (small_conv_eligible(c0) && small_conv_eligible(c1)) {
      if (g_small_conv_fm && c0.ldw % 32 == 0 && c1.ldw % 32 == 0) {
        // fragment-major weight copies (whole cache lines per fragment load), made the first time the block takes this path
        if (!L.fm_conv0) {
          // never under a stream capture (the conversions would be recorded, not run, and the copies stay uninitialised: every later
          // evaluation would read garbage weights); the pointers are published only after both copies have been enqueued
          hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
          if (s) (void)hipStreamIsCapturing(s, &cap);
          T2P_REQUIRE(cap == hipStreamCaptureStatusNone, "the first evaluation builds per-block weight copies: run one eager step before capturing");
          void* f0 = pool_.persistent((size_t)Cout * c0.ldw * 2);
          void* f1 = pool_.persistent((size_t)Cout * c1.ldw * 2);
          if (!f0 || !f1) return T2P_ERR_HIP;
          T2P_TRY(launch_sf_frag_major(dt, c0.Wt, f0, Cout, (int)c0.ldw, s));
          T2P_TRY(launch_sf_frag_major(dt, c1.Wt, f1, Cout, (int)c1.ldw, s));
          L.fm_conv0 = f0; L.fm_conv1 = f1;
        }
        c0.Wt = L.fm_conv0; c0.w_fm = 1; c1.Wt = L.fm_conv1; c1.w_fm = 1;
      }
      POOL_GET(a1s, void*, (size_t)rows_out * Cout * dtype_size(dt));
      c0.normed = a1s;
      T2P_TRY(launch_small_conv_gn(c0, s));
      pool_.put(a0);
      void* xpooled = nullptr;
      if (L.has_conv2) {
        if (L.down) {                     // the shortcut of a down block reads the 2x2-averaged input (layers.py:313-315)
          xpooled = pool_.get((size_t)rows_out * Cin * dtype_size(dt));
          if (!xpooled) return T2P_ERR_HIP;
          T2P_TRY(launch_pool2x2(x.p, xpooled, dt, B, x.H, x.W, Cin, s, x.lowp));
          c1.X0 = xpooled;
        } else {
          c1.X0 = x.p;
          c1.X1 = skip ? skip->p : nullptr;
        }
      } else {
        c1.R = x.p;
      }
      POOL_GET(o, float*, (size_t)rows_out * Cout * (olp ? dtype_size(dt) : 4));
      float* o_stats = nullptr;
      void* pre = nullptr;
      if (c1.groups) {
        pre = pool_.get((size_t)rows_out * Cout * dtype_size(dt));
        if (!pre) return T2P_ERR_HIP;
      }
      if (Ho * Wo == 64 && g_fuse_gn_stats) {
        o_stats = (float*)pool_.get((size_t)(rows_out / 64) * Cout * 2 * 4);
        if (!o_stats) return T2P_ERR_HIP;
      }
      c1.A = a1s; c1.out = o; c1.out_f32 = olp ? 0 : 1; c1.col_stats = o_stats; c1.normed = pre;
      T2P_TRY(launch_small_conv_gn(c1, s));
      pool_.put(a1s);
      pool_.put(xpooled);
      *out = Act{o, Cout, Ho, Wo, o_stats, olp};
      out->pre_norm = pre; out->pre_for = pre ? hint.norm : nullptr; out->pre_silu = hint.silu;
      return T2P_OK;
    }
  }
  float* h1_stats = nullptr;
  bool h1_lowp = false;
  void* a1_fused = nullptr;
  POOL_GET(h1, float*, (size_t)rows_out * Cout * 4);
  {
    GemmParams p;
    p.dtype = dt; p.A0 = a0; p.a_f32 = dt == DT_F32; p.C0 = Cin; p.lda0 = Cin;
    p.taps = 9; p.H = Ho; p.W = Wo; p.a_up = L.up;
    p.Bw = L.conv0.w; p.ldb = L.conv0.K; p.M = (int)rows_out; p.N = Cout;
    if (L.up && L.conv0_up4 && !skip) { p.Bw4 = L.conv0_up4; p.ldb4 = 4L * Cin; }
    p.bias_n = L.conv0.b; p.bias_bn = tb_ + L.temb_off; p.ld_bn = tb_ld_; p.rows_per_batch = Ho * Wo;
    p.C = h1; p.c_f32 = 1; p.ldc = Cout;
    // h1 is read once more, by GroupNorm_1 only: when its statistics come out of this epilogue
    // (computed from the fp32 values) the tensor itself is stored in the compute dtype
    T2P_TRY(attach_ws(p));
    if (dt != DT_F32 && gemm_fuses_post_gn(p, L.gn1.G)) {
      // split-K convolution: its second pass applies GroupNorm_1 + SiLU itself (h1 is read by nothing else)
      a1_fused = pool_.get((size_t)rows_out * Cout * dtype_size(dt));
      if (!a1_fused) return T2P_ERR_HIP;
      p.gn_gamma = L.gn1.gamma; p.gn_beta = L.gn1.beta; p.gn_groups = L.gn1.G; p.gn_silu = 1; p.gn_eps = 1e-6f; p.gn_out = a1_fused;
      p.C = nullptr;
      T2P_TRY(gemm(p, s));
    } else {
    // (maps of <= 64 pixels go through the single-launch GroupNorm, which takes its own statistics:
    // h1 stays fp32 there so that they are still taken from unrounded values)
    if (g_lowp_h1 && dt != DT_F32 && g_fuse_gn_stats && gemm_fuses_col_stats(p) && (Ho * Wo) % 64 == 0 &&
        !(g_gn_small && Ho * Wo <= 64)) {
      p.c_f32 = 0;
      h1_lowp = gemm_fuses_col_stats_lowp(p);
      if (!h1_lowp) p.c_f32 = 1;
    }
    T2P_TRY(gemm_stats(p, &h1_stats, s));
    if (h1_lowp && !h1_stats) return T2P_ERR_STATE;
    }
  }
  pool_.put(a0);
  Act h1a{h1, Cout, Ho, Wo, h1_stats, h1_lowp};
  void* a1 = a1_fused;
  if (!a1) T2P_TRY(group_norm(h1a, nullptr, L.gn1, 1e-6f, 1, 0, B, &a1, s));
  free_act(h1a);
  // second convolution; set up here because the shortcut may ride in its K loop
  GemmParams pc;
  pc.dtype = dt; pc.A0 = a1; pc.a_f32 = dt == DT_F32; pc.C0 = Cout; pc.lda0 = Cout;
  pc.taps = 9; pc.H = Ho; pc.W = Wo;
  pc.Bw = L.conv1.w; pc.ldb = L.conv1.K; pc.M = (int)rows_out; pc.N = Cout;
  pc.bias_n = L.conv1.b; pc.rows_per_batch = Ho * Wo;
  pc.alpha = cfg_.skip_rescale ? 0.70710678118654752440f : 1.f;
  bool fused_shortcut = false;
  void* xpooled = nullptr;
  if (L.conv1x.w && (L.down || x.lowp || xraw) && x.C % 64 == 0 && (!skip || skip->C % 64 == 0) && gemm_can_fuse_shortcut(pc)) {
    fused_shortcut = true;
    pc.Bw = L.conv1x.w; pc.ldb = L.conv1x.K; pc.bias_n = L.conv1x.b;
    if (L.down) {                     // the shortcut of a down block reads the 2x2-averaged input (layers.py:313-315)
      xpooled = pool_.get((size_t)rows_out * Cin * dtype_size(dt));
      if (!xpooled) return T2P_ERR_HIP;
      T2P_TRY(launch_pool2x2(x.p, xpooled, dt, B, x.H, x.W, Cin, s, x.lowp));
      pc.X0 = xpooled; pc.CX0 = Cin; pc.ldx0 = Cin;
    } else if (xraw) {
      pc.X0 = xraw; pc.CX0 = Cin; pc.ldx0 = Cin;
    } else {
      pc.X0 = x.p; pc.CX0 = x.C; pc.ldx0 = x.C;
      if (skip) { pc.X1 = skip->p; pc.CX1 = skip->C; pc.ldx1 = skip->C; }
    }
  }
  // shortcut branch
  const float* r = x.p;
  float* rbuf = nullptr;
  bool rbuf_lowp = false;
  int r_up = 0;
  if (fused_shortcut) {
    r = nullptr;
  } else if (L.has_conv2) {
    GemmParams p;
    p.dtype = dt; p.a_f32 = dt == DT_F32;
    void* pooled = nullptr;
    long rrows;
    if (L.down) {
      pooled = pool_.get((size_t)rows_out * Cin * dtype_size(dt));
      if (!pooled) return T2P_ERR_HIP;
      T2P_TRY(launch_pool2x2(x.p, pooled, dt, B, x.H, x.W, Cin, s, x.lowp));
      p.A0 = pooled; p.C0 = Cin; p.lda0 = Cin;
      rrows = rows_out;
    } else {
      if (xraw) {
        p.A0 = xraw; p.C0 = Cin; p.lda0 = Cin;
      } else {
        p.a_f32 = x.lowp ? 0 : 1;       // 16-bit stream: the LDS-DMA kernel reads both concat sources in place
        p.A0 = x.p; p.C0 = x.C; p.lda0 = x.C;
        if (skip) { p.A1 = skip->p; p.C1 = skip->C; p.lda1 = skip->C; }
      }
      rrows = (long)B * x.H * x.W;   // for `up` the 1x1 conv runs at the low resolution: it commutes
      r_up = L.up;                   // with nearest up-sampling exactly
    }
    rbuf_lowp = false;   // the shortcut output stays fp32: storing it in 16 bits measured no gain and costs accuracy
    rbuf = (float*)pool_.get((size_t)rrows * Cout * (rbuf_lowp ? dtype_size(dt) : 4));
    if (!rbuf) return T2P_ERR_HIP;
    p.Bw = L.conv2.w; p.ldb = L.conv2.K; p.M = (int)rrows; p.N = Cout; p.bias_n = L.conv2.b;
    p.C = rbuf; p.c_f32 = rbuf_lowp ? 0 : 1; p.ldc = Cout;
    T2P_TRY(gemm(p, s));
    pool_.put(pooled);
    pool_.put(xraw);
    r = rbuf;
  } else {
    T2P_REQUIRE(!skip && Cin == Cout, "identity shortcut needs equal channels");
  }
  const bool olp = res_lowp();           // block output (the residual stream) in the compute dtype
  POOL_GET(o, float*, (size_t)rows_out * Cout * (olp ? dtype_size(dt) : 4));
  float* o_stats = nullptr;
  pc.R = r; pc.ldr = Cout; pc.r_up = r_up;
  pc.r_lowp = (r == x.p ? x.lowp : rbuf_lowp) ? 1 : 0;   // identity shortcut: the block input itself
  pc.C = o; pc.c_f32 = olp ? 0 : 1; pc.ldc = Cout;
  void* pre = nullptr;
  T2P_TRY(attach_ws(pc));
  if (hint.norm && olp && hint.norm->C == Cout && gemm_fuses_post_gn(pc, hint.norm->G)) {
    // the block that follows starts with a GroupNorm of this output alone: the split-K second pass applies it as well, and
    // still writes the output itself (residual stream) with its column statistics (skip connections)
    pre = pool_.get((size_t)rows_out * Cout * dtype_size(dt));
    if (!pre) return T2P_ERR_HIP;
    pc.gn_gamma = hint.norm->gamma; pc.gn_beta = hint.norm->beta; pc.gn_groups = hint.norm->G; pc.gn_silu = hint.silu; pc.gn_eps = 1e-6f;
    if ((Ho * Wo) % 64 == 0 && g_fuse_gn_stats) {
      o_stats = (float*)pool_.get((size_t)(rows_out / 64) * Cout * 2 * 4);
      if (!o_stats) return T2P_ERR_HIP;
      pc.col_stats = o_stats;
    }
    pc.gn_out = pre;
    T2P_TRY(gemm(pc, s));
  } else {
    T2P_TRY(gemm_stats(pc, &o_stats, s));
  }
  pool_.put(a1);
  pool_.put(rbuf);
  if (fused_shortcut) { pool_.put(xraw); pool_.put(xpooled); }
  *out = Act{o, Cout, Ho, Wo, o_stats, olp};
  out->pre_norm = pre; out->pre_for = hint.norm; out->pre_silu = hint.silu;
  return T2P_OK;
}

// softmax(q k^T * scale) v for [B][heads]; q,k row-major with head h at column h*d; vt = v^T
int Engine::attention(const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, int B,
                      int heads, int nq, int nk, int d, float scale, hipStream_t s) {
  const int dt = dtype();
  if (g_flash_attention && attention_flash_eligible(dt, d, ldq, ldk, ldvt, (long)heads * d))
    return launch_attention_flash(dt, q, ldq, k, ldk, vt, ldvt, out, B, heads, nq, nk, d, scale, s);
  if (attention_strip_eligible(dt, heads, nq, nk, d, ldq, ldk, ldvt, d))
    return launch_attention_strip(dt, q, ldq, k, ldk, vt, ldvt, out, d, B, nq, d, scale, s);
  const long nkp = (long)round_up((size_t)nk, 8);
  const long rows = (long)B * heads * nq;
  POOL_GET(S, float*, (size_t)rows * nkp * 4);
  POOL_GET(P, void*, (size_t)rows * nkp * dtype_size(dt));
  GemmParams p;
  p.dtype = dt; p.a_f32 = dt == DT_F32;
  p.A0 = q; p.C0 = d; p.lda0 = ldq; p.M = nq; p.N = nk;
  p.Bw = k; p.ldb = ldk;
  p.nz0 = B; p.nz1 = heads;
  p.sA_z0 = (long)nq * ldq; p.sA_z1 = d; p.sB_z0 = (long)nk * ldk; p.sB_z1 = d;
  p.C = S; p.c_f32 = 1; p.ldc = nkp; p.sC_z0 = (long)heads * nq * nkp; p.sC_z1 = (long)nq * nkp;
  T2P_TRY(gemm(p, s));
  T2P_TRY(launch_softmax(S, nkp, P, nkp, dt, rows, nk, scale, s));
  GemmParams r;
  r.dtype = dt; r.a_f32 = dt == DT_F32;
  r.A0 = P; r.C0 = nk; r.lda0 = nkp; r.M = nq; r.N = d;
  r.Bw = vt; r.ldb = ldvt;
  r.nz0 = B; r.nz1 = heads;
  r.sA_z0 = (long)heads * nq * nkp; r.sA_z1 = (long)nq * nkp; r.sB_z0 = (long)heads * d * ldvt; r.sB_z1 = (long)d * ldvt;
  r.C = out; r.c_f32 = 0; r.ldc = (long)heads * d; r.sC_z0 = (long)nq * heads * d; r.sC_z1 = d;
  T2P_TRY(launch_gemm(r, s));
  pool_.put(S);
  pool_.put(P);
  return T2P_OK;
}

// v^T[b] = W_v a_b^T : [C][n] (row stride npad) -- the projection is written transposed so that
// P V is a plain "A row-major, B = [N][K]" GEMM with no transposing loads.
static int project_vt(int dt, const DevLinear& wv, const void* a, long lda, int B, int n, long npad, void* vt,
                      hipStream_t s) {
  GemmParams p;
  p.dtype = dt; p.a_f32 = dt == DT_F32;
  p.A0 = wv.w; p.C0 = wv.K; p.lda0 = wv.K; p.M = wv.N; p.N = n;
  p.Bw = a; p.ldb = lda;
  p.nz0 = B; p.sA_z0 = 0; p.sB_z0 = (long)n * lda;
  p.bias_m = wv.b;
  p.C = vt; p.c_f32 = 0; p.ldc = npad; p.sC_z0 = (long)wv.N * npad;
  return launch_gemm(p, s);
}

// AttnBlockpp.forward (layers.py:160-176)
int Engine::attn_block(Layer& L, const Act& x, Act* out, int B, hipStream_t s) {
  const int C = x.C, n = x.H * x.W, dt = dtype();
  const size_t es = dtype_size(dt);
  const long rows = (long)B * n, npad = (long)round_up((size_t)n, 8);
  const bool olp = res_lowp();
  const bool merged = g_attn_merged && L.v3.w && attention_strip_eligible(dt, 1, n, n, C, 2 * C, 2 * C, npad, C) && (!x.lowp || olp);
  if (merged && x.lowp && L.fm_qk && g_attn_proj) {
    // C = 256: GroupNorm apply, q | k and the transposed value projection as ONE launch over 32-row blocks (attn_proj_kernel)
    AttnProjArgs e;
    e.dtype = dt; e.B = B; e.n = n; e.C = C; e.npad = npad;
    const bool normed = x.pre_norm && x.pre_for == &L.gn0 && x.pre_silu == 0;
    e.x = normed ? x.pre_norm : (const void*)x.p;
    e.cstats = normed ? nullptr : x.cstats;
    e.gn_gamma = L.gn0.gamma; e.gn_beta = L.gn0.beta; e.groups = L.gn0.G; e.gn_eps = 1e-6f;
    e.w_qk = L.fm_qk; e.b_qk = L.qk.b; e.w_v = L.fm_v3;
    if ((normed || x.cstats) && attn_proj_eligible(e)) {
      POOL_GET(qk2, char*, (size_t)rows * 2 * C * es);
      POOL_GET(vt2, void*, (size_t)B * C * npad * es);
      e.qk = qk2; e.vt = vt2;
      void* kfm2 = nullptr;
      if (g_attn_fm && npad == n && attention_strip_frag_major_ok(dt, n, C)) {     // K and V^T in the order the attention kernel streams them
        kfm2 = pool_.get((size_t)rows * C * es);
        if (!kfm2) return T2P_ERR_HIP;
        e.k_fm = kfm2;
      }
      T2P_TRY(launch_attn_proj(e, s));
      if (normed) { pool_.put(x.pre_norm); x.pre_norm = nullptr; }
      const float att_scale2 = 1.f / std::sqrt((float)C);
      POOL_GET(y, float*, (size_t)rows * C * (olp ? es : 4));
      float* y_stats = nullptr;
      if (g_fuse_gn_stats && n % 64 == 0) {
        y_stats = (float*)pool_.get((size_t)(rows / 64) * C * 2 * 4);
        if (!y_stats) return T2P_ERR_HIP;
      }
      StripEpilogue ep;
      ep.bias = L.v3.b; ep.residual = x.p; ep.r_lowp = 1; ep.ldr = C; ep.alpha = cfg_.skip_rescale ? 0.70710678118654752440f : 1.f;
      ep.out_f32 = olp ? 0 : 1; ep.col_stats = y_stats;
      ep.frag_major = kfm2 ? 1 : 0;
      if (kfm2) T2P_TRY(launch_attention_strip(dt, qk2, 2 * C, kfm2, C, vt2, npad, y, C, B, n, C, att_scale2, s, &ep));
      else T2P_TRY(launch_attention_strip(dt, qk2, 2 * C, qk2 + (size_t)C * es, 2 * C, vt2, npad, y, C, B, n, C, att_scale2, s, &ep));
      pool_.put(kfm2);
      pool_.put(qk2);
      pool_.put(vt2);
      *out = Act{y, C, x.H, x.W, y_stats, olp};
      return T2P_OK;
    }
  }
  void* a = nullptr;
  T2P_TRY(group_norm(x, nullptr, L.gn0, 1e-6f, 0, 0, B, &a, s));
  POOL_GET(qk, char*, (size_t)rows * 2 * C * es);
  // q | k projection; where the wide-head attention kernel takes fragment-major operands (plan switch 45: the 32 x 32 level of the
  // C = 512 configurations) the k columns and the transposed values are written that way by the products' epilogues
  GemmParams pq;
  pq.dtype = dt; pq.A0 = a; pq.a_f32 = dt == DT_F32; pq.C0 = C; pq.lda0 = C; pq.Bw = L.qk.w; pq.ldb = C; pq.M = (int)rows; pq.N = 2 * C;
  pq.bias_n = L.qk.b; pq.C = qk; pq.c_f32 = 0; pq.ldc = 2 * C;
  T2P_TRY(attach_ws(pq));
  void* kfm = nullptr;
  GemmParams pv;      // the transposed value projection of the merged path (project_vt's parameters), probed for the same option
  pv.dtype = dt; pv.a_f32 = dt == DT_F32; pv.A0 = L.v3.w; pv.C0 = C; pv.lda0 = C; pv.M = C; pv.N = n; pv.Bw = a; pv.ldb = C;
  pv.nz0 = B; pv.sA_z0 = 0; pv.sB_z0 = (long)n * C; pv.c_f32 = 0; pv.ldc = npad; pv.sC_z0 = (long)C * npad;
  T2P_TRY(attach_ws(pv));
  bool fm = false;
  if (merged && g_attn_fm && npad == n && attention_strip_frag_major_ok(dt, n, C)) {
    GemmParams tq = pq, tv = pv;
    tq.rows_per_batch = n; tq.frag_col0 = C; tq.frag_ns = C / 32; tq.frag_bstride = (long)n * C; tq.c_frag = qk;     // (placeholder pointers)
    tv.frag_col0 = 0; tv.frag_ns = n / 32; tv.frag_bstride = (long)n * C; tv.c_frag = qk; tv.C = qk;
    fm = gemm_writes_frag_major(tq) && gemm_writes_frag_major(tv);
    if (fm) {
      kfm = pool_.get((size_t)rows * C * es);
      if (!kfm) return T2P_ERR_HIP;
      tq.c_frag = kfm;
      pq = tq;
      pv = tv;
    }
  }
  T2P_TRY(launch_gemm(pq, s));
  POOL_GET(vt, void*, (size_t)B * C * npad * es);
  const float att_scale = 1.f / std::sqrt((float)C);
  const float out_alpha = cfg_.skip_rescale ? 0.70710678118654752440f : 1.f;
  if (merged) {
    // NIN_2 and NIN_3 as one projection (Layer::v3): the attention kernel's epilogue adds b2 W3 + b3 and the block input and
    // scales by 1 / sqrt 2 -- (x + NIN_3(softmax(q k^T) v)) / sqrt 2, layers.py:170-176 -- no output-projection GEMM
    if (fm) {
      pv.C = vt; pv.c_frag = vt;           // every column fragment-major: C itself receives nothing
      T2P_TRY(launch_gemm(pv, s));
    } else {
      DevLinear nob = L.v3;
      nob.b = nullptr;
      T2P_TRY(project_vt(dt, nob, a, C, B, n, npad, vt, s));
    }
    pool_.put(a);
    POOL_GET(y, float*, (size_t)rows * C * (olp ? es : 4));
    float* y_stats = nullptr;
    if (g_fuse_gn_stats && n % 64 == 0) {
      y_stats = (float*)pool_.get((size_t)(rows / 64) * C * 2 * 4);
      if (!y_stats) return T2P_ERR_HIP;
    }
    StripEpilogue ep;
    ep.bias = L.v3.b; ep.residual = x.p; ep.r_lowp = x.lowp ? 1 : 0; ep.ldr = C; ep.alpha = out_alpha; ep.out_f32 = olp ? 0 : 1;
    ep.col_stats = y_stats;
    ep.frag_major = fm ? 1 : 0;
    if (fm) T2P_TRY(launch_attention_strip(dt, qk, 2 * C, kfm, C, vt, npad, y, C, B, n, C, att_scale, s, &ep));
    else T2P_TRY(launch_attention_strip(dt, qk, 2 * C, qk + (size_t)C * es, 2 * C, vt, npad, y, C, B, n, C, att_scale, s, &ep));
    pool_.put(kfm);
    pool_.put(qk);
    pool_.put(vt);
    *out = Act{y, C, x.H, x.W, y_stats, olp};
    return T2P_OK;
  }
  T2P_TRY(project_vt(dt, L.v, a, C, B, n, npad, vt, s));
  pool_.put(a);
  POOL_GET(o, void*, (size_t)rows * C * es);
  T2P_TRY(attention(qk, 2 * C, qk + (size_t)C * es, 2 * C, vt, npad, o, B, 1, n, n, C, att_scale, s));
  pool_.put(qk);
  pool_.put(vt);
  POOL_GET(y, float*, (size_t)rows * C * (olp ? es : 4));
  float* y_stats = nullptr;
  T2P_TRY(linear(o, false, L.out, rows, y, !olp, x.p, out_alpha, s, true, &y_stats, x.lowp));
  pool_.put(o);
  *out = Act{y, C, x.H, x.W, y_stats, olp};
  return T2P_OK;
}

// SpatialTransformer.forward with one BasicTransformerBlock (attention.py:208-215, 250-263)
// plan switch 40: the chain after the cross-attention (to_out + residual -> LayerNorm_3 -> ff.net.0 with GEGLU) on the row-block kernel
// too: every workgroup streams 1.15 MB of weights for it, which pays only when the launch fills the chip (>= 8192 rows: cfg3
// 16.26 -> 16.19 ms per step; at cfg5's 4096 rows 10.53 -> 10.55, so the three separate launches stay there).  With row-major weights
// (half of every fetched line unused, each line fetched twice) the chain was slower everywhere: 16.83 -> 16.88 ms at cfg3.
bool g_st_tail = true;
// development key 43: fewest rows for which the block's last chain (with the third product) is taken; without the third product
// (plan switch 42 off) the chain needs twice as many (measured at cfg5's 4096 rows: +0.02 ms without, -0.06 ms with it)
int g_st_tail_rows = 4096;
bool g_attn_proj = true;       // plan switch 46: the projections of an AttnBlockpp in one launch at C = 256 (attn_proj_kernel)
bool g_attn_fm = true;         // plan switch 45: fragment-major K / V^T for the wide-head attention kernel (where the shapes allow it)
bool g_st_ffpo = true;         // plan switch 42: the merged ff.net.2 / proj_out product inside the chain after the cross-attention
bool g_small_conv_fm = true;   // plan switch 41: the small-map convolution kernel reads fragment-major weight copies
int Engine::st_block(Layer& L, const Act& x, Act* out, int B, hipStream_t s) {
  const int C = x.C, n = x.H * x.W, dt = dtype(), heads = cfg_.n_heads, d = C / heads;
  const size_t es = dtype_size(dt);
  const long rows = (long)B * n, npad = (long)round_up((size_t)n, 8);
  const float scale = 1.f / std::sqrt((float)d);
  T2P_REQUIRE(L.ctx_k && L.ctx_vt && ctx_B_ == B, "set_context must be called with the same batch before score");
  // the block's own residual stream t: fp32, or the compute dtype together with the stream between blocks
  const bool tl = res_lowp();
  POOL_GET(t, float*, (size_t)rows * C * (tl ? es : 4));
  const bool qkv_flash = g_qkv_fused && L.a1_qkv.w && g_flash_attention && attention_flash_eligible(dt, d, 3 * C, 3 * C, 3 * C, C);
  // GroupNorm -> proj_in -> LayerNorm_1 -> q | k | v as ONE launch where the row-block kernel applies (stfuse.hip)
  char* qkv_pre = nullptr;
  // the block's last launch (plan switch 42): to_out + residual -> LayerNorm_3 -> ff.net.0 (GEGLU) -> [g | t] W_ffpo + x in one kernel;
  // its column sums are accumulated by pairs of workgroups into a buffer the entry kernel zeroes
  const bool mega = tl && x.lowp && qkv_flash && g_st_tail && g_st_ffpo && g_ffpo_merged && L.fm_ffpo && L.fm_ff1 && rows >= g_st_tail_rows && n % 64 == 0 &&
                    g_fuse_geglu && L.a2_out.b && L.ff1.b;
  float* y2 = nullptr;
  float* y2_stats = nullptr;
  if (tl && x.lowp && qkv_flash) {
    StEntryArgs e;
    e.dtype = dt; e.B = B; e.n = n; e.C = C;
    const bool normed = x.pre_norm && x.pre_for == &L.gn0 && x.pre_silu == 0;
    e.x = normed ? x.pre_norm : (const void*)x.p;
    e.cstats = normed ? nullptr : x.cstats;
    e.gn_gamma = L.gn0.gamma; e.gn_beta = L.gn0.beta; e.groups = L.gn0.G; e.gn_eps = 1e-6f;
    e.w_in = L.fm_in; e.b_in = L.proj_in.b;
    e.ln_gamma = L.ln1.gamma; e.ln_beta = L.ln1.beta; e.ln_eps = 1e-5f;
    e.w_qkv = L.fm_qkv; e.n2 = 3 * C;
    // (no column sums and no producer-side norm -- the 4 x 4 level: the GroupNorm as its own small launch, the rest of the chain here)
    void* a_small = nullptr;
    if (!normed && !x.cstats && L.proj_in.b && L.fm_in && g_st_fuse) {
      StEntryArgs probe = e;
      probe.x = x.p; probe.cstats = nullptr;
      if (st_entry_eligible(probe)) {
        T2P_TRY(group_norm(x, nullptr, L.gn0, 1e-6f, 0, 0, B, &a_small, s));
        e.x = a_small; e.cstats = nullptr;
      }
    }
    if ((normed || x.cstats || a_small) && L.proj_in.b && L.fm_in && st_entry_eligible(e)) {
      qkv_pre = (char*)pool_.get((size_t)rows * 3 * C * es);
      if (!qkv_pre) return T2P_ERR_HIP;
      if (mega) {
        y2 = (float*)pool_.get((size_t)rows * C * es);
        if (!y2) return T2P_ERR_HIP;
        if (g_fuse_gn_stats) {
          y2_stats = (float*)pool_.get((size_t)(rows / 64) * C * 2 * 4);
          if (!y2_stats) return T2P_ERR_HIP;
          e.zero = y2_stats; e.zero_n = (long)(rows / 64) * C * 2;
        }
      }
      e.t = t; e.qkv = qkv_pre;
      T2P_TRY(launch_st_entry(e, s));
      if (normed) { pool_.put(x.pre_norm); x.pre_norm = nullptr; }
    }
    pool_.put(a_small);
  }
  if (!qkv_pre) {
    void* a = nullptr;
    T2P_TRY(group_norm(x, nullptr, L.gn0, 1e-6f, 0, 0, B, &a, s));
    T2P_TRY(linear(a, false, L.proj_in, rows, t, !tl, nullptr, 1.f, s));
    pool_.put(a);
  }
  POOL_GET(ln, void*, (size_t)rows * C * es);
  POOL_GET(o, void*, (size_t)rows * C * es);
  // attn1: self-attention
  if (!qkv_pre) T2P_TRY(launch_layernorm(t, L.ln1.gamma, L.ln1.beta, ln, dt, rows, C, 1e-5f, s, tl));
  if (qkv_flash) {
    char* qkv = qkv_pre;
    if (!qkv) {
      qkv = (char*)pool_.get((size_t)rows * 3 * C * es);
      if (!qkv) return T2P_ERR_HIP;
      T2P_TRY(linear(ln, false, L.a1_qkv, rows, qkv, false, nullptr, 1.f, s, false));
    }
    T2P_TRY(launch_attention_flash(dt, qkv, 3 * C, qkv + (size_t)C * es, 3 * C, qkv + (size_t)2 * C * es, 3 * C, o, B, heads, n, n, d, scale, s,
                                   true));
    pool_.put(qkv);
  } else {
    POOL_GET(qk, char*, (size_t)rows * 2 * C * es);
    T2P_TRY(linear(ln, false, L.a1_qk, rows, qk, false, nullptr, 1.f, s, false));
    POOL_GET(vt, void*, (size_t)B * C * npad * es);
    T2P_TRY(project_vt(dt, L.a1_v, ln, C, B, n, npad, vt, s));
    T2P_TRY(attention(qk, 2 * C, qk + (size_t)C * es, 2 * C, vt, npad, o, B, heads, n, n, d, scale, s));
    pool_.put(qk);
    pool_.put(vt);
  }
  // attn2: cross-attention to the cached text keys / values.  t += to_out(o) -> LayerNorm_2 -> to_q: one launch where the row-block
  // kernel applies (the first product's residual and output are both t: a workgroup's rows are its own)
  {
    POOL_GET(q, void*, (size_t)rows * C * es);
    StEntryArgs e;
    e.dtype = dt; e.B = B; e.n = n; e.C = C; e.x = o; e.w_in = L.fm_out1; e.b_in = L.a1_out.b; e.res = t;
    e.ln_gamma = L.ln2.gamma; e.ln_beta = L.ln2.beta; e.ln_eps = 1e-5f; e.w_qkv = L.fm_q2; e.n2 = C; e.t = t; e.qkv = q;
    if (tl && qkv_pre && L.a1_out.b && !L.a2_q.b && st_entry_eligible(e)) {
      T2P_TRY(launch_st_entry(e, s));
    } else {
      T2P_TRY(linear(o, false, L.a1_out, rows, t, !tl, t, 1.f, s, true, nullptr, tl));
      T2P_TRY(launch_layernorm(t, L.ln2.gamma, L.ln2.beta, ln, dt, rows, C, 1e-5f, s, tl));
      T2P_TRY(linear(ln, false, L.a2_q, rows, q, false, nullptr, 1.f, s, false));
    }
    T2P_TRY(attention(q, C, L.ctx_k, C, L.ctx_vt, ctx_Tpad_, o, B, heads, n, ctx_T_, d, scale, s));
    pool_.put(q);
  }
  // feed-forward with GEGLU.  t += to_out(o) -> LayerNorm_3 -> ff.net.0 (GEGLU): one launch where the row-block kernel applies
  {
    POOL_GET(g, void*, (size_t)rows * 4 * C * es);
    GemmParams p;
    p.dtype = dt; p.A0 = ln; p.a_f32 = dt == DT_F32; p.C0 = C; p.lda0 = C;
    p.Bw = L.ff1.w; p.ldb = C; p.M = (int)rows; p.N = 8 * C; p.bias_n = L.ff1.b;
    p.C = g; p.c_f32 = 0; p.ldc = 4 * C; p.geglu = 1;
    T2P_TRY(attach_ws(p));
    StEntryArgs e;
    e.dtype = dt; e.B = B; e.n = n; e.C = C; e.x = o; e.w_in = L.fm_out2; e.b_in = L.a2_out.b; e.res = t;
    e.ln_gamma = L.ln3.gamma; e.ln_beta = L.ln3.beta; e.ln_eps = 1e-5f; e.w_qkv = L.fm_ff1; e.b2 = L.ff1.b; e.n2 = 8 * C; e.geglu = 1;
    e.t = t; e.qkv = g;
    const bool tail_chain = tl && qkv_pre && g_st_tail && (y2 || rows >= 2 * g_st_tail_rows) && L.fm_ff1 && g_fuse_geglu && gemm_fuses_geglu(p) && L.a2_out.b && L.ff1.b && st_entry_eligible(e);
    if (tail_chain && y2) {
      e.w3 = L.fm_ffpo; e.b3 = L.ffpo.b; e.res3 = x.p; e.y = y2; e.y_stats = y2_stats; e.qkv = nullptr;
      T2P_REQUIRE(st_entry_eligible(e), "row chain with the third product");
      T2P_TRY(launch_st_entry(e, s));
      pool_.put(g); pool_.put(o); pool_.put(ln); pool_.put(t);
      *out = Act{y2, C, x.H, x.W, y2_stats, true};
      return T2P_OK;
    }
    if (y2) { pool_.put(y2); pool_.put(y2_stats); y2 = nullptr; y2_stats = nullptr; }     // (entry chain taken, tail not: unused)
    if (tail_chain) {
      T2P_TRY(launch_st_entry(e, s));
    } else {
      T2P_TRY(linear(o, false, L.a2_out, rows, t, !tl, t, 1.f, s, true, nullptr, tl));
      T2P_TRY(launch_layernorm(t, L.ln3.gamma, L.ln3.beta, ln, dt, rows, C, 1e-5f, s, tl));
    }
    if (tail_chain) {
    } else if (g_fuse_geglu && gemm_fuses_geglu(p)) {
      T2P_TRY(gemm(p, s));                       // value * gelu(gate) in the GEMM epilogue
    } else {
      POOL_GET(u, float*, (size_t)rows * 8 * C * 4);
      T2P_TRY(linear(ln, false, L.ff1, rows, u, true, nullptr, 1.f, s));
      T2P_TRY(launch_geglu(u, g, dt, rows, 4 * C, s, 1));
      pool_.put(u);
    }
    if (g_ffpo_merged && tl && L.ffpo.w) {
      // ff.net.2 and proj_out as ONE product over [g | t] (Layer::ffpo): x + proj_out(t + ff2(g)), attention.py:213-215, 259-263
      pool_.put(o);
      const bool olp2 = res_lowp();
      POOL_GET(y2, float*, (size_t)rows * C * (olp2 ? es : 4));
      float* y2_stats = nullptr;
      GemmParams q;
      q.dtype = dt; q.a_f32 = 0; q.A0 = g; q.C0 = 4 * C; q.lda0 = 4 * C; q.A1 = t; q.C1 = C; q.lda1 = C;
      q.Bw = L.ffpo.w; q.ldb = L.ffpo.K; q.M = (int)rows; q.N = C; q.bias_n = L.ffpo.b;
      q.R = x.p; q.ldr = C; q.r_lowp = x.lowp ? 1 : 0;
      q.C = y2; q.c_f32 = olp2 ? 0 : 1; q.ldc = C; q.rows_per_batch = n;
      T2P_TRY(gemm_stats(q, &y2_stats, s));
      pool_.put(g);
      pool_.put(ln);
      pool_.put(t);
      *out = Act{y2, C, x.H, x.W, y2_stats, olp2};
      return T2P_OK;
    }
    T2P_TRY(linear(g, false, L.ff2, rows, t, !tl, t, 1.f, s, true, nullptr, tl));
    pool_.put(g);
  }
  pool_.put(o);
  const bool olp = res_lowp();
  POOL_GET(y, float*, (size_t)rows * C * (olp ? es : 4));
  float* y_stats = nullptr;
  if (tl) {                                              // t already is a GEMM operand
    T2P_TRY(linear(t, false, L.proj_out, rows, y, !olp, x.p, 1.f, s, true, &y_stats, x.lowp));
  } else if (dt != DT_F32 && g_raw_copies) {
    T2P_TRY(launch_convert(t, ln, dt, rows * C, s));     // residual stream -> compute dtype (reuses the LN buffer)
    T2P_TRY(linear(ln, false, L.proj_out, rows, y, !olp, x.p, 1.f, s, true, &y_stats, x.lowp));
  } else {
    T2P_TRY(linear(t, true, L.proj_out, rows, y, !olp, x.p, 1.f, s, true, &y_stats, x.lowp));
  }
  pool_.put(ln);
  pool_.put(t);
  *out = Act{y, C, x.H, x.W, y_stats, olp};
  return T2P_OK;
}

// the first norm of `l` when it reads its input alone at full resolution (what a producer may apply ahead of time)
static NormHint first_norm_of(const Layer* l, bool concat_input) {
  NormHint h;
  if (!l || concat_input || l->down) return h;
  h.norm = &l->gn0;
  h.silu = l->kind == 0 ? 1 : 0;
  return h;
}

int Engine::run_stage(Stage& st, Act& h, const Act* skip, int B, hipStream_t s, const Layer* next_after) {
  Act cur = h;
  bool own = false;   // cur was produced inside this stage
  for (size_t i = 0; i < st.layers.size(); ++i) {
    Layer& L = st.layers[i];
    Act nxt;
    LayerScope scope(L.prefix + (L.kind == 0 ? (L.up ? ",res_up," : L.down ? ",res_down," : ",res,") : L.kind == 1 ? ",attn," : ",st,") +
                     std::to_string(cur.H) + "," + std::to_string(L.in_ch) + "," + std::to_string(L.out_ch), s);
    const Layer* nextL = i + 1 < st.layers.size() ? &st.layers[i + 1] : next_after;
    if (L.kind == 0) T2P_TRY(res_block(L, cur, i == 0 ? skip : nullptr, &nxt, B, s, res_lowp() ? first_norm_of(nextL, false) : NormHint()));
    else if (L.kind == 1) T2P_TRY(attn_block(L, cur, &nxt, B, s));
    else T2P_TRY(st_block(L, cur, &nxt, B, s));
    if (g_tap_index >= 0 && g_tap_counter++ == g_tap_index) {
      const long n = (long)B * nxt.H * nxt.W * nxt.C;
      g_tap_shape[0] = nxt.C; g_tap_shape[1] = nxt.H; g_tap_shape[2] = nxt.W; g_tap_shape[3] = nxt.lowp;
      float* dst = g_tap_dst;
      const long cap = g_tap_cap;
      g_tap_index = -1; g_tap_dst = nullptr; g_tap_cap = 0;      // one capture per arming: a later evaluation never writes to a buffer the caller may have freed
      T2P_REQUIRE(dst && n <= cap, "tap buffer too small");
      T2P_TRY(launch_widen(nxt.p, nxt.lowp ? dtype() : DT_F32, dst, n, s));
    }
    drop_pre_norm(cur);              // (only if this layer did not take it)
    if (own) free_act(cur);
    cur = nxt;
    own = true;
  }
  h = cur;
  return T2P_OK;
}

int Engine::set_context(const float* ctx, int B, int T, hipStream_t s) {
  T2P_REQUIRE(finalized_, "finalize the engine first");
  T2P_REQUIRE(ctx && B > 0 && T > 0, "set_context arguments");
  const int dt = dtype(), D = cfg_.context_dim;
  const size_t es = dtype_size(dt);
  const long Tpad = (long)round_up((size_t)T, 8);
  const bool realloc = (B != ctx_B_ || T != ctx_T_);
  const void* cx = ctx;
  void* conv = nullptr;
  PoolLease lease(pool_);                              // (the per-layer key / value caches are handed over with keep())
  if (dt != DT_F32) {
    conv = pool_.get((size_t)B * T * D * es);
    if (!conv) return T2P_ERR_HIP;
    T2P_TRY(launch_convert(ctx, conv, dt, (long)B * T * D, s));
    cx = conv;
  }
  auto each = [&](Layer& l) -> int {
    if (l.kind != 2) return T2P_OK;
    const int C = l.in_ch;
    if (realloc || !l.ctx_k) {
      pool_.put(l.ctx_k);
      pool_.put(l.ctx_vt);
      l.ctx_k = pool_.get((size_t)B * T * C * es);
      l.ctx_vt = pool_.get((size_t)B * C * Tpad * es);
      pool_.keep(l.ctx_k);
      pool_.keep(l.ctx_vt);
      if (!l.ctx_k || !l.ctx_vt) return T2P_ERR_HIP;
    }
    T2P_TRY(linear(cx, false, l.a2_k, (long)B * T, l.ctx_k, false, nullptr, 1.f, s, false));
    T2P_TRY(project_vt(dt, l.a2_v, cx, D, B, T, Tpad, l.ctx_vt, s));
    return T2P_OK;
  };
  for (Stage& st : input_stages_) for (Layer& l : st.layers) T2P_TRY(each(l));
  for (Layer& l : mid_stage_.layers) T2P_TRY(each(l));
  for (Stage& st : out_stages_) for (Layer& l : st.layers) T2P_TRY(each(l));
  pool_.put(conv);
  ctx_B_ = B; ctx_T_ = T; ctx_Tpad_ = (int)Tpad;
  return T2P_OK;
}

// UNetModel.forward (ncsnpp.py:220-263)
int Engine::score(const float* x, const int* labels, const int* step_counter, float* out, int B, hipStream_t s,
                  const float* labels_f, const int* label_table, const float* label_f_table) {
  T2P_REQUIRE(finalized_, "finalize the engine first");
  T2P_REQUIRE(x && out && B > 0 && (labels || step_counter), "score arguments");
  const int L = cfg_.max_res_num, HW = L * L, Cx = cfg_.num_channels, N = cfg_.num_scales;
  const int R = labels ? B : 1;
  PoolLease lease(pool_);                              // whatever an early return below leaves checked out goes back to the pool
  g_tap_counter = 0;
  std::unique_ptr<LayerScope> pre_scope(new LayerScope("pre,pre," + std::to_string(L) + "," + std::to_string(Cx) + "," + std::to_string(nf_), s));
  POOL_GET(emb, float*, (size_t)R * nf_ * 4);
  POOL_GET(t1, float*, (size_t)R * temb_dim_ * 4);
  POOL_GET(t2, float*, (size_t)R * temb_dim_ * 4);
  POOL_GET(tb, float*, (size_t)R * temb_total_ * 4);
  T2P_REQUIRE(!labels_f || labels, "fractional labels come with integer labels (sigma index)");
  T2P_REQUIRE(!label_table || (!labels && step_counter), "a label table goes with the device step counter");
  T2P_REQUIRE(!label_f_table || label_table, "the fractional label table goes with the integer one (sigma index)");
  T2P_TRY(launch_timestep_embedding(labels, labels_f, step_counter, emb, R, nf_, s, label_table, N, label_f_table));
  T2P_TRY(launch_small_linear(emb, (const float*)pre0_.w, pre0_.b, t1, R, nf_, temb_dim_, 0, s));
  T2P_TRY(launch_small_linear(t1, (const float*)pre1_.w, pre1_.b, t2, R, temb_dim_, temb_dim_, 0, s));
  T2P_TRY(launch_small_linear(t2, (const float*)dense_all_.w, dense_all_.b, tb, R, temb_dim_, temb_total_, 1, s));
  tb_ = tb;
  tb_ld_ = labels ? temb_total_ : 0;
  POOL_GET(scale, float*, (size_t)B * 4);
  if (cfg_.scale_by_sigma) {
    T2P_TRY(launch_gather_label(labels, step_counter, inv_sigma_, scale, B, N, s, label_table));
  } else {
    std::vector<float> ones(B, 1.f);
    T2P_HIP_CHECK(hipMemcpyAsync(scale, ones.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    T2P_HIP_CHECK(hipStreamSynchronize(s));
  }
  const bool hlp = res_lowp() && (Cx == 5 || Cx == 8) && pre_conv_direct_;   // the residual stream starts here
  POOL_GET(h0, float*, (size_t)B * HW * nf_ * (hlp ? dtype_size(dtype()) : 4));
  float* h0_stats = nullptr;
  if ((Cx == 5 || Cx == 8) && pre_conv_direct_) {
    // 16-bit modes: the GroupNorm column statistics of h0 (read by the first block and, through the skip stack, by the last
    // stage) come out of the input convolution instead of two passes over the tensor
    if (hlp && g_fuse_gn_stats && pre_conv_fuses_col_stats(L, nf_) && !(g_gn_small && HW <= 64)) {
      h0_stats = (float*)pool_.get((size_t)B * (HW / 64) * nf_ * 2 * 4);
      if (!h0_stats) return T2P_ERR_HIP;
    }
    if (hlp && pre_conv_split_ && pre_conv_split_ok(dtype(), Cx, L, L, nf_))
      T2P_TRY(launch_pre_conv_split(x, pre_conv_split_, pre_conv_.b, h0, dtype(), B, Cx, L, L, nf_, s, h0_stats));
    else
      T2P_TRY(launch_pre_conv(x, pre_conv_direct_, pre_conv_.b, h0, hlp ? dtype() : DT_F32, B, Cx, L, L, nf_, s, h0_stats));
  } else {
    POOL_GET(xin, float*, (size_t)B * HW * cpad_ * 4);
    T2P_TRY(launch_nchw_to_nhwc(x, xin, B, Cx, HW, cpad_, s));
    GemmParams p;   // pre_conv always in exact fp32: the input has the dynamic range of sigma_max
    p.dtype = DT_F32; p.a_f32 = 1; p.A0 = xin; p.C0 = cpad_; p.lda0 = cpad_;
    p.taps = 9; p.H = L; p.W = L;
    p.Bw = pre_conv_.w; p.ldb = pre_conv_.K; p.M = B * HW; p.N = nf_; p.bias_n = pre_conv_.b;
    p.rows_per_batch = HW;
    p.C = h0; p.c_f32 = 1; p.ldc = nf_;
    T2P_TRY(gemm(p, s));
    pool_.put(xin);
  }
  bool h0_lowp = hlp;
  if (res_lowp() && !hlp) {              // generic input-conv path: bring the stream to its storage type
    POOL_GET(h0c, float*, (size_t)B * HW * nf_ * dtype_size(dtype()));
    T2P_TRY(launch_convert(h0, h0c, dtype(), (long)B * HW * nf_, s));
    pool_.put(h0);
    h0 = h0c;
    h0_lowp = true;
  }
  pool_.put(emb); pool_.put(t1); pool_.put(t2);
  pre_scope.reset();

  std::vector<Act> hs;
  Act h{h0, nf_, L, L, h0_stats, h0_lowp};
  hs.push_back(h);
  for (size_t i = 0; i < input_stages_.size(); ++i) {
    // the stage's last block may apply the first norm of the block that follows it (next input stage, or the mid stage)
    // (run_stage owns the pre-applied norm its input carries: the first block takes it, or it is dropped there)
    const Layer* next = i + 1 < input_stages_.size() ? &input_stages_[i + 1].layers[0] : &mid_stage_.layers[0];
    T2P_TRY(run_stage(input_stages_[i], h, nullptr, B, s, next));
    hs.push_back(h);
    hs.back().pre_norm = nullptr;      // the skip-stack copy does not own the pre-applied norm
  }
  // mid stage: its input stays on the skip stack
  T2P_TRY(run_stage(mid_stage_, h, nullptr, B, s));
  for (Stage& st : out_stages_) {
    Act skip = hs.back();
    hs.pop_back();
    T2P_REQUIRE(skip.C == st.skip_ch && skip.H == h.H, "skip stack mismatch");
    Act in = h;
    in.pre_norm = nullptr;
    T2P_TRY(run_stage(st, h, &skip, B, s));
    free_act(in);
    free_act(skip);
  }
  T2P_REQUIRE(hs.empty(), "skip stack not consumed");
  // head: GroupNorm -> SiLU -> conv3x3 (nf -> C), stored NCHW and divided by sigma[label]
  LayerScope head_scope("head,head," + std::to_string(L) + "," + std::to_string(final_ch_) + "," + std::to_string(Cx), s);
  void* a = nullptr;
  {
    GemmParams p;
    p.dtype = dtype(); p.a_f32 = p.dtype == DT_F32; p.C0 = final_ch_; p.lda0 = final_ch_;
    p.taps = 9; p.H = L; p.W = L;
    p.Bw = head_conv_.w; p.ldb = head_conv_.K; p.M = B * HW; p.N = Cx; p.bias_n = head_conv_.b;
    p.rows_per_batch = HW;
    p.C = out; p.c_f32 = 1; p.c_nchw = 1; p.row_scale = scale;
    p.A0 = h.p;
    float* hstats = nullptr;
    if (h.lowp && h.cstats && HW % 64 == 0 && gemm_applies_a_norm(p)) {
      // GroupNorm + SiLU of the head inside the convolution's halo staging: the statistics come from the column sums of the last
      // block's epilogue (one small launch), the 134 - 268 MB normalised copy of the map is never written
      hstats = (float*)pool_.get((size_t)B * head_norm_.G * 2 * 4);
      if (!hstats) return T2P_ERR_HIP;
      T2P_TRY(launch_gn_finalize_cols(h.cstats, nullptr, h.C, 0, B, HW, head_norm_.G, 1e-6f, hstats, s));
      p.an_stats = hstats; p.an_gamma = head_norm_.gamma; p.an_beta = head_norm_.beta; p.an_groups = head_norm_.G; p.an_silu = 1;
      T2P_TRY(gemm(p, s));
      pool_.put(hstats);
      free_act(h);
    } else {
      T2P_TRY(group_norm(h, nullptr, head_norm_, 1e-6f, 1, 0, B, &a, s));
      free_act(h);
      p.A0 = a;
      T2P_TRY(gemm(p, s));
    }
  }
  pool_.put(a);
  pool_.put(scale);
  pool_.put(tb);
  tb_ = nullptr;
  return T2P_OK;
}

// ------------------------------------------------------------------------------------------------
Sampler::Sampler(Engine* e, const t2p_sampler_config& cfg) : e_(e), cfg_(cfg) {}

int Sampler::init(const float* g_table_host, const int32_t* label_table_host) {
  T2P_REQUIRE(cfg_.sde == T2P_SDE_VE || cfg_.sde == T2P_SDE_VP, "the fused sampler covers the VE and VP SDEs");
  T2P_REQUIRE(cfg_.sde == T2P_SDE_VE || (g_table_host && label_table_host), "the VP SDE needs its G and label tables (t2p_sampler_create) and t2p_sampler_set_vp_tables");
  T2P_REQUIRE(cfg_.N == e_->cfg().num_scales, "sde.N must equal model.num_scales");
  T2P_REQUIRE(cfg_.batch > 0 && cfg_.global_batch >= cfg_.batch && cfg_.n_steps_each >= 1, "sampler config");
  T2P_REQUIRE(cfg_.eps >= 0.0 && cfg_.eps < 1.0, "eps must lie in [0, T)");
  const int N = cfg_.N;
  // time label of loop step i: round((T - t_i) (N - 1)) with t = linspace(T, eps, N) (sampling.py:257,
  // models/utils.py:159-171); equals i only for tiny eps, e.g. 499 of 1000 labels differ at eps = 1e-3
  std::vector<int32_t> lab(N);
  if (label_table_host) {
    std::copy(label_table_host, label_table_host + N, lab.begin());
  } else {
    for (int i = 0; i < N; ++i) {
      const double t = 1.0 + (cfg_.eps - 1.0) * (double)i / (double)(N - 1);
      lab[i] = (int32_t)std::nearbyint((1.0 - t) * (double)(N - 1));
    }
  }
  for (int i = 0; i < N; ++i) T2P_REQUIRE(lab[i] >= 0 && lab[i] < N, "time label out of range");
  std::vector<float> g(N);
  if (g_table_host) {
    std::copy(g_table_host, g_table_host + N, g.begin());
  } else {
    // VESDE.discretize (sde_lib.py:237-245): step i uses k = N-1-i on the ascending sigmas
    const double a = std::log(cfg_.sigma_min), b = std::log(cfg_.sigma_max);
    auto sig = [&](int k) { return (double)(float)std::exp(a + (b - a) * (double)k / (double)(N - 1)); };
    for (int i = 0; i < N; ++i) {
      const int k = N - 1 - i;
      const double sk = sig(k), sp = k == 0 ? 0.0 : sig(k - 1);
      g[i] = (float)std::sqrt(sk * sk - sp * sp);
    }
  }
  DevPool& pool = e_->pool();
  g_table_ = (float*)pool.persistent((size_t)N * 4);
  label_table_ = (int*)pool.persistent((size_t)N * 4);
  step_dev_ = (int*)pool.persistent(256);
  const t2p_model_config& m = e_->cfg();
  per_sample_ = (long)m.num_channels * m.max_res_num * m.max_res_num;
  n_ = per_sample_ * cfg_.batch;
  score_ = (float*)pool.persistent((size_t)n_ * 4);
  noise_ = (float*)pool.persistent((size_t)n_ * 4);
  xmean_ = (float*)pool.persistent((size_t)n_ * 4);
  sq_ws_ = (float*)pool.persistent((size_t)cfg_.batch * 64 * 2 * 4);
  sums_ = (float*)pool.persistent(256);
  if (!g_table_ || !label_table_ || !step_dev_ || !score_ || !noise_ || !xmean_ || !sq_ws_ || !sums_) return T2P_ERR_HIP;
  T2P_HIP_CHECK(hipMemcpy(g_table_, g.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  T2P_HIP_CHECK(hipMemcpy(label_table_, lab.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  T2P_HIP_CHECK(hipMemset(step_dev_, 0, 256));
  return T2P_OK;
}

int Sampler::reset(int step, hipStream_t s) {
  T2P_REQUIRE(step >= 0 && step < cfg_.N, "step out of range");
  T2P_HIP_CHECK(hipMemcpyAsync(step_dev_, &step, sizeof(int), hipMemcpyHostToDevice, s));
  T2P_HIP_CHECK(hipStreamSynchronize(s));
  host_step_ = step;
  return T2P_OK;
}

int Sampler::set_vp_tables(const float* label_f, const float* score_scale, const float* x_coef, const float* corr_alpha) {
  T2P_REQUIRE(cfg_.sde == T2P_SDE_VP, "VP tables belong to a sampler created with sde = T2P_SDE_VP");
  T2P_REQUIRE(label_f && score_scale && x_coef && corr_alpha, "null table");
  DevPool& pool = e_->pool();
  const size_t bytes = (size_t)cfg_.N * 4;
  float** dst[4] = {&vp_label_f_, &vp_score_scale_, &vp_x_coef_, &vp_alpha_};
  const float* src[4] = {label_f, score_scale, x_coef, corr_alpha};
  for (int i = 0; i < 4; ++i) {
    if (!*dst[i]) *dst[i] = (float*)pool.persistent(bytes);
    if (!*dst[i]) return T2P_ERR_HIP;
    T2P_HIP_CHECK(hipMemcpy(*dst[i], src[i], bytes, hipMemcpyHostToDevice));
  }
  return T2P_OK;
}

int Sampler::set_norm_allreduce(float* sums, t2p_allreduce_fn fn, void* user) {
  T2P_REQUIRE((sums == nullptr) == (fn == nullptr), "the sums buffer and the all-reduce callback go together");
  sums_ext_ = sums; allreduce_ = fn; allreduce_user_ = user;
  return T2P_OK;
}

// one iteration of the loop body of pc_sampler (sampling.py:279-285)
int Sampler::step(float* x, float* x_mean, const float* nc, const float* np, hipStream_t s) {
  T2P_REQUIRE(x, "x is null");
  T2P_REQUIRE(cfg_.n_steps_each == 1 || !nc, "injected corrector noise supports n_steps_each == 1");
  // the schedule tables hold N entries: a step past the end of the run is a caller error (t2p_sampler_reset rewinds)
  T2P_REQUIRE(host_step_ >= 0 && host_step_ < cfg_.N, "PC step index beyond sde.N: call t2p_sampler_reset before another run");
  // Langevin batch mean over global_batch chains (reference DataParallel run): needs the norm sums of the other
  // processes, i.e. the all-reduce hook; without it the mean runs over this process's chains
  T2P_REQUIRE(cfg_.global_batch == cfg_.batch || allreduce_, "global_batch > batch needs t2p_sampler_set_norm_allreduce");
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &cap);
    if (cap == hipStreamCaptureStatusNone && eager_steps_ < (1 << 30)) ++eager_steps_;     // steps that really ran (step_graph / count_dispatches ask)
  }
  const int B = cfg_.batch;
  const bool vp = cfg_.sde == T2P_SDE_VP;
  T2P_REQUIRE(!vp || vp_label_f_, "VP SDE: call t2p_sampler_set_vp_tables first");
  float* sums = allreduce_ ? sums_ext_ : sums_;
  // score function (models/utils.py:138-171): VE = the network output at the integer label; VP = -output / std at the fractional one
  auto score_fn = [&]() -> int {
    T2P_TRY(e_->score(x, nullptr, step_dev_, score_, B, s, nullptr, label_table_, vp ? vp_label_f_ : nullptr));
    if (vp) T2P_TRY(launch_scale_by_table(score_, n_, vp_score_scale_, step_dev_, cfg_.N, s));
    return T2P_OK;
  };
  for (int k = 0; k < cfg_.n_steps_each; ++k) {
    T2P_TRY(score_fn());
    const float* z = nc;
    if (!z) {
      T2P_TRY(launch_philox_normal(noise_, n_, cfg_.seed, 2ull * k + 2, step_dev_, s));
      z = noise_;
    }
    T2P_TRY(launch_langevin_norms(score_, z, B, per_sample_, sq_ws_, sums, s));
    if (allreduce_) {     // sum_b ||grad_b||, sum_b ||noise_b|| over every process's chains (SURVEY 8(e) option B)
      const int rc = allreduce_(sums, (void*)s, allreduce_user_);
      if (rc != 0) { set_last_error("the norm all-reduce callback failed with status " + std::to_string(rc)); return T2P_ERR_STATE; }
    }
    SdeUpdateArgs a;
    a.x = x; a.score = score_; a.noise = z; a.mask = mask_; a.x_initial = x_init_; a.x_out = x; a.n = n_;
    T2P_TRY(launch_langevin_update(a, sums, (float)(allreduce_ ? cfg_.global_batch : B), (float)cfg_.snr, 1.f, s,
                                   vp ? vp_alpha_ : nullptr, vp ? step_dev_ : nullptr, vp ? cfg_.N : 0));
  }
  T2P_TRY(score_fn());
  const float* z = np;
  if (!z) {
    T2P_TRY(launch_philox_normal(noise_, n_, cfg_.seed, 1, step_dev_, s));
    z = noise_;
  }
  SdeUpdateArgs a;
  a.x = x; a.score = score_; a.noise = z; a.mask = mask_; a.x_initial = x_init_; a.x_out = x;
  a.x_mean_out = x_mean ? x_mean : xmean_; a.n = n_;
  T2P_TRY(launch_predictor_update(a, g_table_, step_dev_, 0.f, cfg_.probability_flow, s, cfg_.N, vp ? vp_x_coef_ : nullptr));
  T2P_TRY(launch_add_int(step_dev_, 1, s));
  ++host_step_;
  return T2P_OK;
}

Sampler::~Sampler() {
  if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
}

int Sampler::step_graph(float* x, float* x_mean, hipStream_t s) {
  T2P_REQUIRE(x && x_mean, "step_graph needs explicit x and x_mean buffers");
  T2P_REQUIRE(!allreduce_, "the captured step does not run the norm all-reduce hook: use t2p_sampler_step");
  T2P_REQUIRE(host_step_ >= 0 && host_step_ < cfg_.N, "PC step index beyond sde.N: call t2p_sampler_reset before another run");
  if (eager_steps_ < 1)              // one eager step first: fills the activation pool (no hipMalloc under capture); step() counts it
    return step(x, x_mean, nullptr, nullptr, s);
  if (graph_exec_ && (graph_x_ != x || graph_xm_ != x_mean || graph_mask_ != mask_ || graph_seed_ != cfg_.seed)) {
    (void)hipGraphExecDestroy(graph_exec_);
    graph_exec_ = nullptr;
  }
  if (!graph_exec_) {
    hipGraph_t graph = nullptr;
    T2P_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int step_before = host_step_;
    const int rc = step(x, x_mean, nullptr, nullptr, s);
    host_step_ = step_before;            // the capture enqueued nothing: the replay below is the step
    const hipError_t ec = hipStreamEndCapture(s, &graph);
    if (rc != T2P_OK) {                  // a partial capture is dropped, the mirror of the device counter is untouched
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    T2P_HIP_CHECK(ec);
    T2P_HIP_CHECK(hipGraphInstantiate(&graph_exec_, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    graph_x_ = x; graph_xm_ = x_mean; graph_mask_ = mask_; graph_seed_ = cfg_.seed;
  }
  T2P_HIP_CHECK(hipGraphLaunch(graph_exec_, s));
  ++host_step_;
  return T2P_OK;
}

// number of kernel / memory nodes one PC step enqueues: the step is captured into a hipGraph (nothing executes) and its nodes
// are counted.  Needs a non-default stream and a filled activation pool (one eager step before).
int Sampler::count_dispatches(float* x, float* x_mean, hipStream_t s, int* n_out) {
  T2P_REQUIRE(x && x_mean && n_out, "null argument");
  T2P_REQUIRE(eager_steps_ >= 1, "count_dispatches captures a step: run one eager step first (it sizes the pool and builds per-block weight copies)");
  T2P_REQUIRE(!allreduce_, "the captured step does not run the norm all-reduce hook");
  T2P_REQUIRE(host_step_ >= 0 && host_step_ < cfg_.N, "PC step index beyond sde.N: call t2p_sampler_reset before another run");
  hipGraph_t graph = nullptr;
  T2P_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int step_before = host_step_;
  const int rc = step(x, x_mean, nullptr, nullptr, s);
  host_step_ = step_before;
  const hipError_t ec = hipStreamEndCapture(s, &graph);
  if (rc != T2P_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  T2P_HIP_CHECK(ec);
  size_t n = 0;
  const hipError_t eg = hipGraphGetNodes(graph, nullptr, &n);
  (void)hipGraphDestroy(graph);
  T2P_HIP_CHECK(eg);
  *n_out = (int)n;
  return T2P_OK;
}

int Sampler::run(float* x, float* out, int prior_given, int n_steps, hipStream_t s) {
  T2P_REQUIRE(x && out, "null pointer");
  if (n_steps <= 0 || n_steps > cfg_.N) n_steps = cfg_.N;
  T2P_TRY(reset(0, s));
  if (!prior_given) {
    // VESDE.prior_sampling (sde_lib.py:229-230) then where(mask, x, x_initial)
    T2P_TRY(launch_philox_normal(x, n_, cfg_.seed, 0, nullptr, s));
    if (cfg_.sde == T2P_SDE_VE) T2P_TRY(launch_scale(x, n_, (float)cfg_.sigma_max, s));     // VP prior: N(0, 1) (sde_lib.py:133-134)
    if (mask_) T2P_TRY(launch_apply_mask(x, mask_, x_init_, n_, s));
  }
  for (int i = 0; i < n_steps; ++i) T2P_TRY(step(x, xmean_, nullptr, nullptr, s));
  const float* src = cfg_.denoise ? xmean_ : x;
  T2P_HIP_CHECK(hipMemcpyAsync(out, src, (size_t)n_ * 4, hipMemcpyDeviceToDevice, s));
  if (cfg_.denoise && mask_) T2P_TRY(launch_apply_mask(out, mask_, x_init_, n_, s));   // sampling.py:287
  return T2P_OK;
}

}  // namespace t2p
