// Shared declarations for the text2protein HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace t2p {

// ---- compute dtypes ------------------------------------------------------------------------
// The residual stream, statistics, softmax inputs and accumulators are always fp32.  GEMM
// operands (activations after a norm, weights, attention probabilities) are stored in the
// engine's "compute dtype": fp32 (exact-f32 MFMA), bf16 or fp16 (16-bit MFMA, fp32 accumulate).
enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = DT_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = DT_BF16; };
template <> struct dtype_of<f16_t> { static constexpr int value = DT_F16; };

__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
  // round-to-nearest-even; NaN stays NaN (quiet)
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, (__bf16)f);      // gfx950: v_cvt_pk_bf16_f32 (same rounding)
#endif
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_bits_to_f32(uint16_t b) {
  uint32_t u = ((uint32_t)b) << 16;
  return __builtin_bit_cast(float, u);
}

template <typename T> __host__ __device__ inline T from_f32(float f);
template <> __host__ __device__ inline float from_f32<float>(float f) { return f; }
template <> __host__ __device__ inline bf16_t from_f32<bf16_t>(float f) { return bf16_t{f32_to_bf16_bits(f)}; }
template <> __host__ __device__ inline f16_t from_f32<f16_t>(float f) { return f16_t{(_Float16)f}; }

__host__ __device__ inline float to_f32(float f) { return f; }
__host__ __device__ inline float to_f32(bf16_t b) { return bf16_bits_to_f32(b.v); }
__host__ __device__ inline float to_f32(f16_t h) { return (float)h.v; }

// ---- error handling --------------------------------------------------------------------------
void set_last_error(const std::string& msg);
const char* get_last_error();

#define T2P_OK 0
#define T2P_ERR_INVALID 1
#define T2P_ERR_HIP 2
#define T2P_ERR_STATE 3

#define T2P_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::t2p::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +      \
                            __FILE__ + ":" + std::to_string(__LINE__));                       \
      return T2P_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define T2P_REQUIRE(cond, msg)                                                                \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      ::t2p::set_last_error(std::string("requirement failed: ") + #cond + " -- " + (msg) +    \
                            " at " + __FILE__ + ":" + std::to_string(__LINE__));              \
      return T2P_ERR_INVALID;                                                                 \
    }                                                                                         \
  } while (0)

#define T2P_TRY(expr)                                                                         \
  do {                                                                                        \
    int _rc = (expr);                                                                         \
    if (_rc != T2P_OK) return _rc;                                                            \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel,
// current device), whichever device the calling thread has selected (engine.cpp)
int ensure_dynamic_lds(const void* kernel, int bytes);

// ---- GEMM / implicit-GEMM convolution ------------------------------------------------------------
// C[z][m][n] = alpha * ( sum_k A[z][m][k] * Bw[z][n][k] + bias_n[n] + bias_m[m]
//                        + bias_bn[m / rows_per_batch][n] + R[z][res_row(m)][n] )
// A is gathered: row m is pixel (b, y, x) of an NHWC map; with taps == 9 the K axis is
// (tap, channel) and the source pixel is (y + dy, x + dx) with zero padding (3x3 convolution,
// reference layers.py:89-95); a_up reads the source at half resolution (nearest up-sampling,
// layers.py:179-183).  The channel axis may be the concatenation of two sources (U-Net skip
// concat, reference ncsnpp.py:250) without materialising the concat.
struct GemmParams {
  const void* A0 = nullptr;  // [rows][lda0], C0 channels
  const void* A1 = nullptr;  // optional second source, C1 channels
  int a_f32 = 0;             // sources are fp32 (converted to the compute dtype while staging)
  int C0 = 0, C1 = 0;
  long lda0 = 0, lda1 = 0;
  int taps = 1;              // 1 or 9
  int H = 0, W = 0;          // output map size (spatial modes)
  int a_up = 0;              // source map is (H/2, W/2)
  const void* Bw = nullptr;  // [N][ldb] compute dtype, K index = tap * (C0 + C1) + c
  long ldb = 0;
  // optional, taps == 9 with a_up: phase weights of the same convolution, [4 phases (py, px)][N][ldb4], K index =
  // (ty * 2 + tx) * C0 + c -- the 3x3 taps that read the same source pixel summed (Engine::upload_up4); lets launch_gemm run
  // the layer as four 2x2 convolutions on the source map (4 / 9 of the multiplications)
  const void* Bw4 = nullptr;
  long ldb4 = 0;
  int up_phase = 0;          // set by launch_gemm: 1 + phase of a MODE 3 launch, 5 = all four phases (the phase is blockIdx.y)
  // optional, taps == 9 at full resolution on the LDS-DMA kernels (gemm_can_fuse_shortcut): CX0 + CX1 more K columns read at
  // the output pixel itself from X0 | X1 (compute dtype), weights at K index 9 * (C0 + C1) + c -- the 1x1 shortcut of a
  // residual block folded into its second convolution
  const void* X0 = nullptr;
  const void* X1 = nullptr;
  int CX0 = 0, CX1 = 0;
  long ldx0 = 0, ldx1 = 0;
  int M = 0, N = 0;
  // batching over blockIdx.z = z0 * nz1 + z1
  int nz0 = 1, nz1 = 1;
  long sA_z0 = 0, sA_z1 = 0, sB_z0 = 0, sB_z1 = 0, sC_z0 = 0, sC_z1 = 0, sR_z0 = 0, sR_z1 = 0;
  // epilogue
  const float* bias_n = nullptr;
  const float* bias_m = nullptr;
  const float* bias_bn = nullptr;  // [batch][ld_bn]
  int rows_per_batch = 1;
  long ld_bn = 0;
  const float* R = nullptr;  // residual: fp32, or compute dtype (16-bit) when r_lowp
  int r_lowp = 0;
  long ldr = 0;
  int r_up = 0;              // residual lives at half resolution (uses H, W, rows_per_batch = H*W)
  float alpha = 1.f;
  void* C = nullptr;
  int c_f32 = 1;             // output fp32 (else compute dtype)
  long ldc = 0;
  int c_nchw = 0;            // store C as [batch][N][H*W] fp32, scaled by row_scale[batch]
  const float* row_scale = nullptr;
  int dtype = DT_F32;        // compute dtype of A (if !a_f32), Bw and C (if !c_f32)
  int geglu = 0;             // B rows interleaved (value_j, gate_j): C[row][j] = value * gelu(gate), N/2 columns
  float* col_stats = nullptr;  // optional [M/64][N][2]: per 64-row chunk column sum / sum of squares of C
  void* ws = nullptr;        // optional split-K workspace (fp32 partial tiles)
  size_t ws_bytes = 0;
  // optional (ask gemm_fuses_post_gn): the GroupNorm (+SiLU) that FOLLOWS this product, applied by the split-K second pass,
  // whose blocks then own whole (sample, 64-channel slab) pieces: gn_out [M][N] in the compute dtype receives
  // act(GroupNorm(C)) computed from the fp32 values; C itself may then be null (the raw product is not wanted)
  const float* gn_gamma = nullptr;
  const float* gn_beta = nullptr;
  int gn_groups = 0, gn_silu = 0;
  float gn_eps = 1e-6f;
  void* gn_out = nullptr;
  // optional (ask gemm_applies_a_norm; the thin-output head convolution only): A0 holds the RAW map and the kernel applies
  // act(GroupNorm(.)) while it stages its input halo -- an_stats [batch][groups][2] = (mean, rstd) per sample and group
  const float* an_stats = nullptr;
  const float* an_gamma = nullptr;
  const float* an_beta = nullptr;
  int an_groups = 0, an_silu = 0;
  // optional (ask gemm_writes_frag_major): the columns from frag_col0 on are written FRAGMENT-MAJOR into c_frag (16-bit) instead
  // of row-major into C -- the layout the wide-head attention kernel streams K and V^T from with whole cache lines per load
  // (attention.hip): element (r, c) of sample / batch entry bz, c relative to frag_col0, at
  //   bz frag_bstride + ((r / 32) frag_ns + c / 32) 1024 + ((c / 8) % 2) 512 + ((c / 16) % 2) 256 + (r % 32) 8 + c % 8
  // with r the row inside the sample (rows_per_batch rows each, bz = row / rows_per_batch) or inside batch entry z (bz = z)
  void* c_frag = nullptr;
  int frag_col0 = 0, frag_ns = 0;
  long frag_bstride = 0;
};
bool gemm_writes_frag_major(const GemmParams& p);

int launch_gemm(const GemmParams& p, hipStream_t stream);
bool gemm_fuses_col_stats(const GemmParams& p);
bool gemm_can_fuse_shortcut(const GemmParams& p);   // p without X0 / X1: would launch_gemm take the extra K segment?
bool gemm_fuses_geglu(const GemmParams& p);
bool gemm_fuses_col_stats_lowp(const GemmParams& p);
bool gemm_fuses_post_gn(const GemmParams& p, int groups);   // p without gn_*: would launch_gemm apply a following GroupNorm of `groups` groups?
void set_gemm_post_gn(bool on);
bool gemm_applies_a_norm(const GemmParams& p);      // p without an_*: would launch_gemm normalise A0 on the fly (head convolution)?
void set_gemm_split_consts(int tiles, int target);
void set_gemm_dma(bool on);
void set_gemm_debug(int v);
void set_gemm_geom(int v);
void set_gemm_ring(int v);
void set_gemm_splitk(bool on);
void set_gemm_force_nsplit(int v);
void set_gemm_midsplit(bool on);
void set_gemm_thin_conv(bool on);
void set_gemm_a_norm(bool on);
void set_gemm_up4(bool on);
void set_gemm_deep_ring(bool on);
void set_gemm_fuse_shortcut(bool on);
void set_gemm_dxs(bool on);
#ifdef T2P_ABLATION
int dxs_stamps_read(unsigned long long* out, int n);   // measurement builds: in-kernel timeline of gemm_dxs_kernel (gemm.hip)
#endif
extern bool g_qkv_fused;
extern bool g_attn_merged, g_ffpo_merged;
extern bool g_flash_attention;   // engine / op API: fused attention kernel where eligible
extern bool g_lowp_h1;         // engine: block-internal conv0 output stored in the compute dtype
extern bool g_lowp_residual;   // engine: residual stream between blocks in the compute dtype (f16 mode)
extern bool g_layernorm16;     // LayerNorm of 16-bit rows of 512 / 1024 channels: the row is read once (plan switch 20)
extern bool g_gn_apply16;      // GroupNorm apply on 16-bit maps: 16-byte accesses, 4 pixels in flight
extern bool g_gn_small;        // engine / op API: single-launch GroupNorm for maps of <= 64 pixels
extern bool g_fuse_geglu;      // engine: GEGLU gating inside the ff1 GEMM epilogue
extern bool g_fuse_gn_stats;   // engine: GroupNorm statistics from the producing GEMM epilogue
extern bool g_raw_copies;   // engine: feed 1x1 shortcut / proj_out GEMMs with compute-dtype copies
void profile_begin();
int profile_end(double out[3][3]);
int profile_dominant(double out[4], const char** name);
int profile_shapes(char* buf, int len);
bool prof_on();
void prof_attention(hipEvent_t a, hipEvent_t b, double flops);
int profile_attention(double out[3]);

}  // namespace t2p
