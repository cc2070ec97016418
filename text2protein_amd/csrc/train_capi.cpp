// extern "C" surface of the training step (declarations and reference citations: include/t2p.h, "training step").
#include <new>

#include "train.h"

using namespace t2p;

#define API_BEGIN try {
#define API_END                                        \
  }                                                    \
  catch (const std::bad_alloc&) {                      \
    set_last_error("out of host memory");              \
    return T2P_ERR_STATE;                              \
  }                                                    \
  catch (const std::exception& e) {                    \
    set_last_error(std::string("exception: ") + e.what()); \
    return T2P_ERR_STATE;                              \
  }

extern "C" {

int t2p_train_create(const t2p_model_config* model, const t2p_train_config* train, t2p_trainer** out) {
  API_BEGIN
  T2P_REQUIRE(model && train && out, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_last_error("no HIP device: the training step has no CPU fallback");
    return T2P_ERR_HIP;
  }
  t2p_trainer* t = new t2p_trainer(*model, *train);
  const int rc = t->impl.build();
  if (rc != T2P_OK) {
    delete t;
    return rc;
  }
  *out = t;
  return T2P_OK;
  API_END
}

void t2p_train_destroy(t2p_trainer* t) { delete t; }

int t2p_train_num_params(const t2p_trainer* t) { return t ? (int)t->impl.params().size() : -1; }

int t2p_train_param_info(const t2p_trainer* t, int i, const char** name, int64_t shape[4], int* ndim) {
  API_BEGIN
  T2P_REQUIRE(t && name && shape && ndim && i >= 0 && i < (int)t->impl.params().size(), "param_info arguments");
  const TParam& p = t->impl.params()[i];
  *name = p.name.c_str();
  *ndim = (int)p.shape.size();
  for (int d = 0; d < 4; ++d) shape[d] = d < *ndim ? p.shape[d] : 1;
  return T2P_OK;
  API_END
}

int t2p_train_load_param(t2p_trainer* t, const char* name, const float* host_data, const int64_t* shape, int ndim) {
  API_BEGIN
  T2P_REQUIRE(t, "null trainer");
  return t->impl.load_param(name, host_data, shape, ndim);
  API_END
}

int t2p_train_read(t2p_trainer* t, int which, const char* name, float* host_out) {
  API_BEGIN
  T2P_REQUIRE(t, "null trainer");
  return t->impl.read_tensor(which, name, host_out);
  API_END
}

int t2p_train_write(t2p_trainer* t, int which, const char* name, const float* host_in) {
  API_BEGIN
  T2P_REQUIRE(t, "null trainer");
  return t->impl.write_tensor(which, name, host_in);
  API_END
}

int t2p_train_set_step(t2p_trainer* t, int64_t step, int64_t adam_updates, int64_t ema_updates) {
  API_BEGIN
  T2P_REQUIRE(t, "null trainer");
  return t->impl.set_step(step, adam_updates, ema_updates);
  API_END
}

int t2p_train_get_step(const t2p_trainer* t, int64_t out3[3]) {
  API_BEGIN
  T2P_REQUIRE(t && out3, "get_step arguments");
  return t->impl.get_step(out3);
  API_END
}

int t2p_train_set_dropout_masks(t2p_trainer* t, const uint8_t* const* device_masks, int n) {
  API_BEGIN
  T2P_REQUIRE(t && n >= 0, "set_dropout_masks arguments");
  return t->impl.set_dropout_masks(device_masks, n);
  API_END
}

int t2p_train_loss(t2p_trainer* t, const t2p_train_batch* batch, int backward, float* loss_host, float* score_out, void* stream) {
  API_BEGIN
  T2P_REQUIRE(t && batch, "null argument");
  return t->impl.loss(*batch, backward != 0, false, loss_host, score_out, (hipStream_t)stream);
  API_END
}

int t2p_train_step(t2p_trainer* t, const t2p_train_batch* batch, float* loss_host, void* stream) {
  API_BEGIN
  T2P_REQUIRE(t && batch, "null argument");
  return t->impl.step(*batch, loss_host, (hipStream_t)stream);
  API_END
}

int t2p_train_apply(t2p_trainer* t, void* stream) {
  API_BEGIN
  T2P_REQUIRE(t, "null trainer");
  return t->impl.apply((hipStream_t)stream);
  API_END
}

int t2p_train_grad_buffer(t2p_trainer* t, float** device_ptr, int64_t* n) {
  API_BEGIN
  T2P_REQUIRE(t && device_ptr && n, "grad_buffer arguments");
  *device_ptr = t->impl.grad_buffer();
  *n = (int64_t)t->impl.num_elements();
  return T2P_OK;
  API_END
}

int t2p_train_eval_loss(t2p_trainer* t, const t2p_train_batch* batch, float* loss_host, void* stream) {
  API_BEGIN
  T2P_REQUIRE(t && batch, "null argument");
  return t->impl.loss(*batch, false, true, loss_host, nullptr, (hipStream_t)stream);
  API_END
}

int64_t t2p_train_device_bytes(const t2p_trainer* t) { return t ? t->impl.device_bytes() : -1; }

int t2p_op_tgemm(const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBk, int64_t sBn, float* C, int64_t ldc, int M, int N,
                 int K, int nz, int64_t sAz, int64_t sBz, int64_t sCz, float alpha, float beta, const float* bias_n, int ksplit, int conv,
                 int H, int W, int conv_C, void* stream) {
  API_BEGIN
  TGemmArgs a;
  a.A = A; a.sAm = sAm; a.sAk = sAk; a.sAz0 = sAz;
  a.B = B; a.sBk = sBk; a.sBn = sBn; a.sBz0 = sBz;
  a.C = C; a.ldc = ldc; a.sCz0 = sCz; a.M = M; a.N = N; a.K = K; a.nz0 = nz; a.nz1 = 1;
  a.alpha = alpha; a.beta = beta; a.bias_n = bias_n; a.ksplit = ksplit;
  if (conv) { a.conv_b = 1; a.H = H; a.W = W; a.conv_C = conv_C; a.ldx = sBz; a.sBz0 = 0; }
  return launch_tgemm(a, (hipStream_t)stream);
  API_END
}

// the op-level backward entry points take eps and recompute the forward statistics themselves (tests hold no engine state)
int t2p_op_groupnorm_backward(const float* x, const float* dy, const float* gamma, const float* beta, int silu, int batch, int HW, int C,
                              int groups, float eps, float* dx, float* dgamma, float* dbeta, void* stream) {
  API_BEGIN
  T2P_REQUIRE(x && dy && gamma && beta && dx && dgamma && dbeta && batch > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "groupnorm_backward arguments");
  hipStream_t s = (hipStream_t)stream;
  float *stats = nullptr, *partial = nullptr, *ws = nullptr;
  const int nparts = gn_num_chunks(HW) * ((C + 1023) / 1024);
  T2P_HIP_CHECK(hipMalloc(&stats, (size_t)batch * groups * 2 * 4));
  T2P_HIP_CHECK(hipMalloc(&partial, (size_t)batch * nparts * groups * 2 * 4));
  T2P_HIP_CHECK(hipMalloc(&ws, (size_t)gn_bwd_ws_floats(batch, HW, C, groups) * 4));
  GroupNormArgs a;
  a.x0 = x; a.C0 = C; a.B = batch; a.HW = HW; a.G = groups; a.eps = eps; a.partial = partial; a.stats = stats;
  int rc = launch_gn_stats(a, s);
  if (rc == T2P_OK) rc = launch_gn_backward(x, dy, stats, gamma, beta, silu, batch, HW, C, groups, dx, dgamma, dbeta, ws, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(stats); (void)hipFree(partial); (void)hipFree(ws);
  return rc;
  API_END
}

int t2p_op_layernorm_backward(const float* x, const float* dy, const float* gamma, int64_t rows, int C, float eps, float* dx, float* dgamma,
                              float* dbeta, void* stream) {
  API_BEGIN
  return launch_ln_backward(x, dy, gamma, rows, C, eps, dx, dgamma, dbeta, (hipStream_t)stream);
  API_END
}

int t2p_op_softmax_backward(const float* P, float* dP_inout, int64_t rows, int n, float scale, void* stream) {
  API_BEGIN
  return launch_softmax_backward(P, dP_inout, rows, n, scale, (hipStream_t)stream);
  API_END
}

int t2p_op_geglu_backward(const float* u, const float* dy, float* du, int64_t rows, int inner, void* stream) {
  API_BEGIN
  return launch_geglu_backward(u, dy, du, rows, inner, (hipStream_t)stream);
  API_END
}

}  // extern "C"
