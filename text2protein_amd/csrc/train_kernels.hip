// Kernels of the training step (first slice): fp32, gfx950 only.
//
//   * tgemm_kernel       strided batched GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products): every product of the backward pass that
//                        is not a 3x3 convolution over pixels -- dY W, dY^T X, q k^T, P v, P^T dO, dS^T q ... -- reads its operands
//                        through (row stride, column stride) views, so nothing is transposed in memory; the weight gradient of a 3x3
//                        convolution gathers its B operand from the NHWC map (nine shifted views); split-K with hardware fp32 atomics
//                        where K is "every pixel of the batch"
//   * the backward halves of GroupNorm(+SiLU), LayerNorm, softmax, GEGLU, up / down-sampling, dropout
//   * the denoising score-matching loss of the VE SDE and its gradient (reference score_sde_pytorch/losses.py:105-131)
//   * Adam with warm-up and gradient clipping, EMA (losses.py:26-51, models/ema.py:32-49) over flat parameter buffers
//
// Layouts: activations NHWC [B][H W][C] fp32 (the inference engine's layout), the state / noise / masks NCHW as in the reference.
#include <hip/hip_runtime.h>

#include "train_kernels.h"

namespace t2p {

typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv_l(long a, long b) { return (int)((a + b - 1) / b); }
static inline int grid_for(long n, int block, int cap = 65535 * 4) { return (int)std::min<long>((n + block - 1) / block, cap); }

// =====================================================================================================================================
// strided GEMM
// =====================================================================================================================================
template <int BM, int BN, bool AMC, bool BNC, bool CONVB>
__global__ __launch_bounds__(256) void tgemm_kernel(const TGemmArgs p, const int kper) {
  constexpr int BK = 16;
  constexpr int SA = BM + 32, SB = BN + 32;   // row strides = 32 mod 64 banks: the two k rows a fragment read touches never collide
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
  __shared__ float As[BK * SA];
  __shared__ float Bs[BK * SB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
  int zz = blockIdx.z;
  const int ks = zz % p.ksplit; zz /= p.ksplit;
  const int z1 = zz % p.nz1, z0 = zz / p.nz1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = ks * kper, ke = min(p.K, kb + kper);

  const float* A = p.A + (long)z0 * p.sAz0 + (long)z1 * p.sAz1;
  const float* B = p.B + (long)z0 * p.sBz0 + (long)z1 * p.sBz1;

  // staging assignment
  int am, ak, akstep, amstep;
  if (AMC) { am = tid % BM; ak = tid / BM; akstep = 256 / BM; amstep = 0; }
  else     { ak = tid % BK; am = tid / BK; akstep = 0; amstep = 256 / BK; }
  int bn, bk, bkstep, bnstep;
  if (BNC) { bn = tid % BN; bk = tid / BN; bkstep = 256 / BN; bnstep = 0; }
  else     { bk = tid % BK; bn = tid / BK; bkstep = 0; bnstep = 256 / BK; }

  // convolution gather: this thread's column = (tap, channel) is fixed (BNC staging); the pixel coordinates of its EB rows are carried
  // from K-tile to K-tile (no divisions in the loop)
  int cv_c = 0, cv_dy = 0, cv_dx = 0;
  bool cv_ok = false;
  int cv_b[CONVB ? EB : 1], cv_y[CONVB ? EB : 1], cv_x[CONVB ? EB : 1];
  if (CONVB) {
    const int HW = p.H * p.W;
    const int gn = n0 + bn;
    cv_ok = gn < p.N;
    const int tap = cv_ok ? gn / p.conv_C : 0;
    cv_c = gn - tap * p.conv_C;
    cv_dy = tap / 3 - 1;
    cv_dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int gk = kb + bk + i * bkstep;
      cv_b[i] = gk / HW;
      const int rem = gk - cv_b[i] * HW;
      cv_y[i] = rem / p.W;
      cv_x[i] = rem - cv_y[i] * p.W;
    }
  }

  tg_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  for (int k0 = kb; k0 < ke; k0 += BK) {
    float ra[EA], rb[EB];
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int m = am + i * amstep, k = ak + i * akstep;
      const int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < p.M && gk < ke) ? A[(long)gm * p.sAm + (long)gk * p.sAk] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int n = bn + i * bnstep, k = bk + i * bkstep;
      const int gn = n0 + n, gk = k0 + k;
      float v = 0.f;
      if (CONVB) {
        if (cv_ok && gk < ke) {
          const int sy = cv_y[i] + cv_dy, sx = cv_x[i] + cv_dx;
          if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) v = B[((long)(cv_b[i] * p.H + sy) * p.W + sx) * p.ldx + cv_c];
        }
        cv_x[i] += BK;                      // the row this slot stages in the next K-tile
        while (cv_x[i] >= p.W) { cv_x[i] -= p.W; ++cv_y[i]; }
        while (cv_y[i] >= p.H) { cv_y[i] -= p.H; ++cv_b[i]; }
      } else if (gn < p.N && gk < ke) {
        v = B[(long)gk * p.sBk + (long)gn * p.sBn];
      }
      rb[i] = v;
    }
    __syncthreads();     // the previous K-tile has been consumed
#pragma unroll
    for (int i = 0; i < EA; ++i) As[(ak + i * akstep) * SA + am + i * amstep] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[(bk + i * bkstep) * SB + bn + i * bnstep] = rb[i];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[(kk + lh) * SA + wm * (BM / 2) + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + lh) * SB + wn * (BN / 2) + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  float* C = p.C + (long)z0 * p.sCz0 + (long)z1 * p.sCz1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + wm * (BM / 2) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + lr;
        if (col >= p.N) continue;
        float val = p.alpha * acc[i][j][v];
        float* dst = C + (long)row * p.ldc + col;
        if (p.ksplit > 1) {
          if (ks == 0 && p.bias_n) val += p.bias_n[col];
          unsafeAtomicAdd(dst, val);
        } else {
          if (p.bias_n) val += p.bias_n[col];
          if (p.beta != 0.f) val += p.beta * *dst;
          *dst = val;
        }
      }
    }
}

template <int BT, bool AMC, bool BNC, bool CONVB>
static int tgemm_launch_t(const TGemmArgs& a, int ksplit, hipStream_t s) {
  TGemmArgs p = a;
  p.ksplit = ksplit;
  int kper = (a.K + ksplit - 1) / ksplit;
  kper = (kper + 15) / 16 * 16;
  dim3 grid(cdiv_l(a.N, BT), cdiv_l(a.M, BT), a.nz0 * a.nz1 * ksplit);
  hipLaunchKernelGGL((tgemm_kernel<BT, BT, AMC, BNC, CONVB>), grid, dim3(256), 0, s, p, kper);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

int launch_tgemm(const TGemmArgs& a, hipStream_t s) {
  T2P_REQUIRE(a.A && a.B && a.C && a.M > 0 && a.N > 0 && a.K > 0 && a.nz0 >= 1 && a.nz1 >= 1, "tgemm operands");
  T2P_REQUIRE(a.sAm == 1 || a.sAk == 1, "tgemm: A must be a row- or column-major view");
  T2P_REQUIRE(a.conv_b || a.sBn == 1 || a.sBk == 1, "tgemm: B must be a row- or column-major view");
  T2P_REQUIRE(a.ldc >= a.N, "tgemm: ldc");
  if (a.conv_b) T2P_REQUIRE(a.H > 0 && a.W > 0 && a.conv_C > 0 && a.N == 9 * a.conv_C && a.K % (a.H * a.W) == 0 && a.ldx >= a.conv_C,
                            "tgemm: convolution gather shapes");
  const long nz = (long)a.nz0 * a.nz1;
  T2P_REQUIRE(nz <= 65535, "tgemm: batch count");
  // 128 x 128 tiles when they fill the chip by themselves -- or together with the K splits this call may choose (weight gradients)
  const long tiles128 = (long)cdiv_l(a.M, 128) * cdiv_l(a.N, 128) * nz;
  const bool big = a.M >= 128 && a.N >= 128 && (tiles128 >= 128 || (a.ksplit == 0 && tiles128 * (a.K / 256) >= 128));
  const int bt = big ? 128 : 64;
  int ksplit = a.ksplit;
  if (ksplit == 0) {
    const long tiles = (long)cdiv_l(a.M, bt) * cdiv_l(a.N, bt) * nz;
    ksplit = (int)std::max<long>(1, std::min<long>(512 / std::max<long>(tiles, 1), a.K / 256));
  }
  T2P_REQUIRE(ksplit >= 1 && nz * ksplit <= 65535, "tgemm: ksplit");
  T2P_REQUIRE(ksplit == 1 || a.beta == 1.f, "tgemm: split-K accumulates with atomics and needs beta == 1");
  const bool amc = a.sAm == 1, bnc = a.conv_b || a.sBn == 1;
  if (a.conv_b) {
    T2P_REQUIRE(amc, "tgemm: the convolution weight gradient takes dY^T (column-major view) as A");
    return big ? tgemm_launch_t<128, true, true, true>(a, ksplit, s) : tgemm_launch_t<64, true, true, true>(a, ksplit, s);
  }
#define T2P_TG(AM, BN_)                                                                                     \
  if (amc == AM && bnc == BN_)                                                                              \
    return big ? tgemm_launch_t<128, AM, BN_, false>(a, ksplit, s) : tgemm_launch_t<64, AM, BN_, false>(a, ksplit, s);
  T2P_TG(true, true) T2P_TG(true, false) T2P_TG(false, true) T2P_TG(false, false)
#undef T2P_TG
  return T2P_ERR_INVALID;
}

// =====================================================================================================================================
// reductions shared below
// =====================================================================================================================================
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// =====================================================================================================================================
// GroupNorm backward
// =====================================================================================================================================
constexpr int GN_CHUNK = 64;   // pixels per partial-sum block

// ws_partial [B][nchunk][C][2] = per-chunk (sum dv, sum dv n)
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const int silu, const int HW, const int C,
                                                             const int G, float* __restrict__ ws) {
  __shared__ float red[2][256];
  const int b = blockIdx.y, ch = blockIdx.x, nchunk = gridDim.x, tid = threadIdx.x;
  const int p0 = ch * GN_CHUNK, p1 = min(HW, p0 + GN_CHUNK);
  const int cg = C / G;
  const int PL = (C < 256 && 256 % C == 0) ? 256 / C : 1;   // pixel lanes when the channels do not fill the block
  const int cl = PL > 1 ? tid % C : tid, pl = PL > 1 ? tid / C : 0;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + cl;
    float a = 0.f, bq = 0.f;
    if (c < C) {
      const int g = c / cg;
      const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
      const float ga = gamma[c], be = beta[c];
      for (int p = p0 + pl; p < p1; p += PL) {
        const long i = ((long)b * HW + p) * C + c;
        const float n = (x[i] - mean) * rstd;
        float dv = dy[i];
        if (silu) {
          const float v = n * ga + be, sg = sigmoidf_(v);
          dv *= sg * (1.f + v * (1.f - sg));
        }
        a += dv;
        bq += dv * n;
      }
    }
    if (PL > 1) {
      red[0][tid] = a; red[1][tid] = bq;
      __syncthreads();
      if (pl == 0) {
        for (int q = 1; q < PL; ++q) { a += red[0][q * C + cl]; bq += red[1][q * C + cl]; }
      }
      __syncthreads();
    }
    if (c < C && pl == 0) {
      float* o = ws + (((long)b * nchunk + ch) * C + c) * 2;
      o[0] = a; o[1] = bq;
    }
  }
}

// per sample: fold the chunks -> sums [B][C][2], group sums gs [B][G][2] = (sum gamma dv, sum gamma dv n); dgamma / dbeta += over the batch
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                              const int nchunk, const int C, const int G, float* __restrict__ sums,
                                                              float* __restrict__ gs, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int b = blockIdx.x, tid = threadIdx.x, cg = C / G;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, bq = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) {
      const float* o = partial + (((long)b * nchunk + ch) * C + c) * 2;
      a += o[0]; bq += o[1];
    }
    sums[((long)b * C + c) * 2] = a;
    sums[((long)b * C + c) * 2 + 1] = bq;
    unsafeAtomicAdd(dbeta + c, a);
    unsafeAtomicAdd(dgamma + c, bq);
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
      s1 += gamma[c] * sums[((long)b * C + c) * 2];
      s2 += gamma[c] * sums[((long)b * C + c) * 2 + 1];
    }
    gs[((long)b * G + g) * 2] = s1;
    gs[((long)b * G + g) * 2 + 1] = s2;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const int silu, const int HW, const int C, const int G,
                                                           const float* __restrict__ gs, float* __restrict__ dx, const long total) {
  const int cg = C / G;
  const float inv_m = 1.f / ((float)HW * cg);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long bp = i / C;
    const int b = (int)(bp / HW), g = c / cg;
    const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    const float ga = gamma[c];
    const float n = (x[i] - mean) * rstd;
    float dv = dy[i];
    if (silu) {
      const float v = n * ga + beta[c], sg = sigmoidf_(v);
      dv *= sg * (1.f + v * (1.f - sg));
    }
    const float s1 = gs[((long)b * G + g) * 2], s2 = gs[((long)b * G + g) * 2 + 1];
    dx[i] += rstd * (dv * ga - (s1 + n * s2) * inv_m);
  }
}

long gn_bwd_ws_floats(int B, int HW, int C, int G) {
  const long nchunk = (HW + GN_CHUNK - 1) / GN_CHUNK;
  return (long)B * nchunk * C * 2 + (long)B * C * 2 + (long)B * G * 2;
}

int launch_gn_backward(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, int silu,
                       int B, int HW, int C, int G, float* dx, float* dgamma, float* dbeta, float* ws, hipStream_t s) {
  T2P_REQUIRE(x && dy && stats && gamma && beta && dx && dgamma && dbeta && ws, "gn_backward pointers");
  T2P_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && B <= 65535, "gn_backward shapes");
  const int nchunk = (HW + GN_CHUNK - 1) / GN_CHUNK;
  float* partial = ws;
  float* sums = partial + (long)B * nchunk * C * 2;
  float* gs = sums + (long)B * C * 2;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, dy, stats, gamma, beta, silu, HW, C, G, partial);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B), dim3(256), 0, s, partial, gamma, nchunk, C, G, sums, gs, dgamma, dbeta);
  const long total = (long)B * HW * C;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, x, dy, stats, gamma, beta, silu, HW, C, G, gs, dx, total);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// =====================================================================================================================================
// LayerNorm backward: one wavefront per row, per-block channel sums in LDS
// =====================================================================================================================================
constexpr int LN_ROWS = 64;   // rows per block

__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                     const long rows, const int C, const float eps, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float ln_sh[];     // [2][C]
  float* sg = ln_sh;
  float* sb = ln_sh + C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 256) { sg[c] = 0.f; sb[c] = 0.f; }
  __syncthreads();
  const long r0 = (long)blockIdx.x * LN_ROWS, r1 = min(rows, r0 + LN_ROWS);
  const float inv_c = 1.f / C;
  for (long r = r0 + wave; r < r1; r += 4) {
    const float* xr = x + r * C;
    const float* dr = dy + r * C;
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += xr[c];
    const float mean = wave_sum(sum) * inv_c;
    float var = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; var += d * d; }
    const float rstd = rsqrtf(wave_sum(var) * inv_c + eps);
    float c1 = 0.f, c2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float n = (xr[c] - mean) * rstd, dg = dr[c] * gamma[c];
      c1 += dg; c2 += dg * n;
    }
    c1 = wave_sum(c1) * inv_c; c2 = wave_sum(c2) * inv_c;
    for (int c = lane; c < C; c += 64) {
      const float n = (xr[c] - mean) * rstd, d = dr[c];
      dx[r * C + c] += rstd * (d * gamma[c] - c1 - n * c2);
      atomicAdd(sg + c, d * n);      // LDS atomics (ds_add_f32)
      atomicAdd(sb + c, d);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    unsafeAtomicAdd(dgamma + c, sg[c]);
    unsafeAtomicAdd(dbeta + c, sb[c]);
  }
}

int launch_ln_backward(const float* x, const float* dy, const float* gamma, long rows, int C, float eps, float* dx, float* dgamma,
                       float* dbeta, hipStream_t s) {
  T2P_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && rows > 0 && C > 0 && C <= 8192, "ln_backward arguments");
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv_l(rows, LN_ROWS)), dim3(256), 2 * C * sizeof(float), s, x, dy, gamma, rows, C, eps, dx, dgamma, dbeta);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// =====================================================================================================================================
// softmax / GEGLU backward, elementwise
// =====================================================================================================================================
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, const long rows, const int n,
                                                          const float scale) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* pr = P + r * n;
  float* dr = dP + r * n;
  float dot = 0.f;
  for (int j = lane; j < n; j += 64) dot += pr[j] * dr[j];
  dot = wave_sum(dot);
  for (int j = lane; j < n; j += 64) dr[j] = scale * pr[j] * (dr[j] - dot);
}
int launch_softmax_backward(const float* P, float* dP, long rows, int n, float scale, hipStream_t s) {
  T2P_REQUIRE(P && dP && rows > 0 && n > 0, "softmax_backward arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv_l(rows, 4)), dim3(256), 0, s, P, dP, rows, n, scale);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy, float* __restrict__ du,
                                                        const long rows, const int inner) {
  const long total = rows * inner;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / inner;
    const int j = (int)(i - r * inner);
    const float a = u[r * 2 * inner + j], g = u[r * 2 * inner + inner + j], d = dy[i];
    const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * g * g);
    du[r * 2 * inner + j] += d * g * cdf;
    du[r * 2 * inner + inner + j] += d * a * (cdf + g * pdf);
  }
}
int launch_geglu_backward(const float* u, const float* dy, float* du, long rows, int inner, hipStream_t s) {
  T2P_REQUIRE(u && dy && du && rows > 0 && inner > 0, "geglu_backward arguments");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * inner, 256)), dim3(256), 0, s, u, dy, du, rows, inner);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

template <int OP>   // 0 silu, 1 silu backward (+=), 2 axpy, 3 add_scale
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                 const float alpha, const long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (OP == 0) { const float v = a[i]; out[i] = v * sigmoidf_(v); }
    if (OP == 1) { const float v = a[i], sg = sigmoidf_(v); out[i] += b[i] * sg * (1.f + v * (1.f - sg)); }
    if (OP == 2) out[i] += alpha * a[i];
    if (OP == 3) out[i] = alpha * (a[i] + b[i]);
  }
}
int launch_silu(const float* x, float* y, long n, hipStream_t s) {
  T2P_REQUIRE(x && y && n > 0, "silu arguments");
  hipLaunchKernelGGL(ew_kernel<0>, dim3(grid_for(n, 256)), dim3(256), 0, s, x, nullptr, y, 0.f, n);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_silu_backward(const float* x, const float* dy, float* dx, long n, hipStream_t s) {
  T2P_REQUIRE(x && dy && dx && n > 0, "silu_backward arguments");
  hipLaunchKernelGGL(ew_kernel<1>, dim3(grid_for(n, 256)), dim3(256), 0, s, x, dy, dx, 0.f, n);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_axpy(float* y, const float* x, float a, long n, hipStream_t s) {
  T2P_REQUIRE(x && y && n > 0, "axpy arguments");
  hipLaunchKernelGGL(ew_kernel<2>, dim3(grid_for(n, 256)), dim3(256), 0, s, x, nullptr, y, a, n);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_add_scale(const float* a, const float* b, float alpha, float* out, long n, hipStream_t s) {
  T2P_REQUIRE(a && b && out && n > 0, "add_scale arguments");
  hipLaunchKernelGGL(ew_kernel<3>, dim3(grid_for(n, 256)), dim3(256), 0, s, a, b, out, alpha, n);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, const long ld_src, const long src_off, float* __restrict__ dst,
                                                        const long ld_dst, const long dst_off, const long rows, const int C, const int acc) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float v = src[r * ld_src + src_off + c];
    float* d = dst + r * ld_dst + dst_off + c;
    *d = acc ? *d + v : v;
  }
}
int launch_copy_cols(const float* src, long ld_src, long src_off, float* dst, long ld_dst, long dst_off, long rows, int C, int accumulate,
                     hipStream_t s) {
  T2P_REQUIRE(src && dst && rows > 0 && C > 0 && ld_src >= src_off + C && ld_dst >= dst_off + C, "copy_cols arguments");
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(rows * C, 256)), dim3(256), 0, s, src, ld_src, src_off, dst, ld_dst, dst_off, rows, C, accumulate);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// out[z][n] += sum over the rows of chunk (blockIdx.y) of sample z:  64 columns x 4 row lanes per block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, const long rows_per_z, const int N, const long ld,
                                                     float* __restrict__ out, const long ld_out, const int rows_per_block) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cl, z = blockIdx.z;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows_per_z, r0 + rows_per_block);
  float a = 0.f;
  if (n < N)
    for (long r = r0 + rl; r < r1; r += 4) a += dy[((long)z * rows_per_z + r) * ld + n];
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && n < N) unsafeAtomicAdd(out + (long)z * ld_out + n, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
int launch_colsum(const float* dy, long rows, int N, long ld, float* out, hipStream_t s) {
  T2P_REQUIRE(dy && out && rows > 0 && N > 0 && ld >= N, "colsum arguments");
  const int rpb = 256;
  T2P_REQUIRE(cdiv_l(rows, rpb) <= 65535, "colsum rows");
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv_l(N, 64), cdiv_l(rows, rpb), 1), dim3(256), 0, s, dy, rows, N, ld, out, 0L, rpb);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_colsum_per_sample(const float* dy, int B, int HW, int N, float* out, long ld_out, int accumulate, hipStream_t s) {
  T2P_REQUIRE(dy && out && B > 0 && B <= 65535 && HW > 0 && N > 0 && ld_out >= N, "colsum_per_sample arguments");
  if (!accumulate) T2P_HIP_CHECK(hipMemset2DAsync(out, ld_out * sizeof(float), 0, N * sizeof(float), B, s));
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv_l(N, 64), cdiv_l(HW, rpb), B), dim3(256), 0, s, dy, (long)HW, N, (long)N, out, ld_out, rpb);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ---- 2x nearest up-sampling / 2x2 mean down-sampling --------------------------------------------------------------------------------
template <int OP>   // 0 up, 1 up backward, 2 down, 3 down backward; (H, W) = the SMALL map of the pair
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in, float* __restrict__ out, const int H, const int W, const int C,
                                                       const long total_small) {
  const int W2 = 2 * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_small; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const long b = t / H;
    const long big = ((b * 2 * H + 2 * y) * W2 + 2 * x) * C + c;     // top-left of the 2x2 block in the large map
    const long o01 = C, o10 = (long)W2 * C, o11 = (long)W2 * C + C;
    if (OP == 0) { const float v = in[i]; out[big] = v; out[big + o01] = v; out[big + o10] = v; out[big + o11] = v; }
    if (OP == 1) out[i] += in[big] + in[big + o01] + in[big + o10] + in[big + o11];
    if (OP == 2) out[i] = 0.25f * (in[big] + in[big + o01] + in[big + o10] + in[big + o11]);
    if (OP == 3) { const float v = 0.25f * in[i]; out[big] += v; out[big + o01] += v; out[big + o10] += v; out[big + o11] += v; }
  }
}
#define T2P_RESAMPLE(name, OP, HS, WS)                                                                              \
  int name(const float* a, float* b, int B, int H, int W, int C, hipStream_t s) {                                   \
    T2P_REQUIRE(a && b && B > 0 && H > 0 && W > 0 && C > 0 && (HS) > 0 && (WS) > 0, #name " arguments");             \
    const long total = (long)B * (HS) * (WS) * C;                                                                   \
    hipLaunchKernelGGL(resample_kernel<OP>, dim3(grid_for(total, 256)), dim3(256), 0, s, a, b, (HS), (WS), C, total); \
    T2P_HIP_CHECK(hipGetLastError());                                                                               \
    return T2P_OK;                                                                                                  \
  }
T2P_RESAMPLE(launch_up2, 0, H, W)                   // x [B][H][W][C] -> y [B][2H][2W][C]
T2P_RESAMPLE(launch_up2_backward, 1, H, W)          // dy [B][2H][2W][C] -> dx [B][H][W][C] +=
int launch_down2(const float* x, float* y, int B, int H, int W, int C, hipStream_t s) {   // x [B][H][W][C] -> y [B][H/2][W/2][C]
  T2P_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0, "down2 arguments");
  const long total = (long)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(resample_kernel<2>, dim3(grid_for(total, 256)), dim3(256), 0, s, x, y, H / 2, W / 2, C, total);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_down2_backward(const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s) {   // dy [B][H/2][W/2][C] -> dx [B][H][W][C] +=
  T2P_REQUIRE(dy && dx && B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C > 0, "down2_backward arguments");
  const long total = (long)B * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(resample_kernel<3>, dim3(grid_for(total, 256)), dim3(256), 0, s, dy, dx, H / 2, W / 2, C, total);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ---- dropout -----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const unsigned char* __restrict__ keep, const float inv_keep,
                                                      float* __restrict__ y, const long n, const int acc) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = keep[i] ? x[i] * inv_keep : 0.f;
    y[i] = acc ? y[i] + v : v;
  }
}
int launch_dropout(const float* x, const unsigned char* keep, float inv_keep, float* y, long n, int accumulate, hipStream_t s) {
  T2P_REQUIRE(x && keep && y && n > 0, "dropout arguments");
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, keep, inv_keep, y, n, accumulate);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__device__ inline void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, const uint32_t k0, const uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ keep, const long n, const float p,
                                                           const unsigned long long seed, const unsigned long long stream_id) {
  const long nq = (n + 3) / 4;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    uint32_t c0 = (uint32_t)q, c1 = (uint32_t)((unsigned long long)q >> 32), c2 = (uint32_t)stream_id, c3 = (uint32_t)(stream_id >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t w[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = q * 4 + j;
      if (i < n) keep[i] = ((w[j] >> 8) * (1.0f / 16777216.0f)) >= p ? 1 : 0;
    }
  }
}
int launch_dropout_mask(unsigned char* keep, long n, float p, unsigned long long seed, unsigned long long stream_id, hipStream_t s) {
  T2P_REQUIRE(keep && n > 0 && p >= 0.f && p < 1.f, "dropout_mask arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, keep, n, p, seed, stream_id);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ---- convolution weight layouts -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_w_prep_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, const int Co,
                                                          const int Ci, const int Cip, const int Cop) {
  const long nf = (long)Co * 9 * Cip, nd = (long)Ci * 9 * Cop;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (long)gridDim.x * 256) {
    if (i < nf) {
      const int ci = (int)(i % Cip), t = (int)((i / Cip) % 9), co = (int)(i / ((long)Cip * 9));
      wf[i] = ci < Ci ? w[((long)co * Ci + ci) * 9 + t] : 0.f;
    } else if (wd) {
      const long j = i - nf;
      const int co = (int)(j % Cop), t = (int)((j / Cop) % 9), ci = (int)(j / ((long)Cop * 9));
      wd[j] = co < Co ? w[((long)co * Ci + ci) * 9 + (8 - t)] : 0.f;
    }
  }
}
int launch_conv_w_prep(const float* w, float* wf, float* wd, int Co, int Ci, int Cip, int Cop, hipStream_t s) {
  T2P_REQUIRE(w && wf && Co > 0 && Ci > 0 && Cip >= Ci && Cop >= Co, "conv_w_prep arguments");
  const long n = (long)Co * 9 * Cip + (wd ? (long)Ci * 9 * Cop : 0);
  hipLaunchKernelGGL(conv_w_prep_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, w, wf, wd, Co, Ci, Cip, Cop);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
__global__ __launch_bounds__(256) void conv_w_grad_fold_kernel(const float* __restrict__ dwc, float* __restrict__ gw, const int Co, const int Ci,
                                                               const int Cip) {
  const long n = (long)Co * Ci * 9;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int t = (int)(i % 9), ci = (int)((i / 9) % Ci), co = (int)(i / ((long)Ci * 9));
    gw[i] += dwc[((long)co * 9 + t) * Cip + ci];
  }
}
int launch_conv_w_grad_fold(const float* dwc, float* gw, int Co, int Ci, int Cip, hipStream_t s) {
  T2P_REQUIRE(dwc && gw && Co > 0 && Ci > 0 && Cip >= Ci, "conv_w_grad_fold arguments");
  hipLaunchKernelGGL(conv_w_grad_fold_kernel, dim3(grid_for((long)Co * Ci * 9, 256)), dim3(256), 0, s, dwc, gw, Co, Ci, Cip);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// =====================================================================================================================================
// denoising score matching (VE SDE)
// =====================================================================================================================================
__device__ inline float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}
__device__ inline double block_sum_256_d(double v, double* sh) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

__device__ inline void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, const uint32_t k0, const uint32_t k1);
__global__ void dsm_prepare_kernel(const float* __restrict__ t_in, const int B, const float t_eps, const float sigma_min, const float sigma_max,
                                   const int N, const float* __restrict__ inv_sigma_table, const unsigned long long seed,
                                   const unsigned long long step, float* __restrict__ t_out, float* __restrict__ stdv, int* __restrict__ labels,
                                   float* __restrict__ scale) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float t;
  if (t_in) {
    t = t_in[b];
  } else {
    uint32_t c0 = (uint32_t)b, c1 = 0, c2 = (uint32_t)step, c3 = (uint32_t)(step >> 32), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    t = (c0 >> 8) * (1.0f / 16777216.0f) * (1.f - t_eps) + t_eps;
  }
  t_out[b] = t;
  stdv[b] = sigma_min * powf(sigma_max / sigma_min, t);
  int lab = (int)rintf((1.f - t) * (float)(N - 1));
  lab = min(max(lab, 0), N - 1);
  labels[b] = lab;
  scale[b] = inv_sigma_table ? inv_sigma_table[lab] : 1.f;
}
int launch_dsm_prepare(const float* t_in, int B, float t_eps, float sigma_min, float sigma_max, int N, const float* inv_sigma_table,
                       unsigned long long seed, unsigned long long step, float* t_out, float* std, int* labels, float* scale, hipStream_t s) {
  T2P_REQUIRE(B > 0 && N >= 2 && t_out && std && labels && scale && sigma_min > 0.f && sigma_max > sigma_min, "dsm_prepare arguments");
  hipLaunchKernelGGL(dsm_prepare_kernel, dim3(cdiv_l(B, 64)), dim3(64), 0, s, t_in, B, t_eps, sigma_min, sigma_max, N, inv_sigma_table, seed, step,
                     t_out, std, labels, scale);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// grid (chunks, B)
__global__ __launch_bounds__(256) void dsm_perturb_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ stdv,
                                                          const unsigned char* __restrict__ mask_pair, const unsigned char* __restrict__ mask_inpaint,
                                                          const int flags, const int C, const int HW, float* __restrict__ perturbed,
                                                          unsigned char* __restrict__ mask, float* __restrict__ num_elem) {
  __shared__ float sh[4];
  const int b = blockIdx.y;
  const long per = (long)C * HW;
  const float sd = stdv[b];
  float cnt = 0.f;
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < per; j += (long)gridDim.x * 256) {
    const int c = (int)(j / HW), p = (int)(j - (long)c * HW);
    bool m = mask_pair[(long)b * HW + p] != 0;
    if ((flags & 1) && c == C - 1) m = false;
    if ((flags & 2) && c >= 4 && c < 7) m = false;
    if ((flags & 4) && !mask_inpaint[(long)b * HW + p]) m = false;
    const long i = (long)b * per + j;
    const float xv = x[i];
    perturbed[i] = m ? xv + sd * z[i] : xv;
    mask[i] = m ? 1 : 0;
    cnt += m ? 1.f : 0.f;
  }
  cnt = block_sum_256(cnt, sh);
  if (threadIdx.x == 0 && cnt != 0.f) unsafeAtomicAdd(num_elem + b, cnt);
}
int launch_dsm_perturb(const float* x, const float* z, const float* std, const unsigned char* mask_pair, const unsigned char* mask_inpaint,
                       int cond_flags, int B, int C, int L, float* perturbed, unsigned char* mask, float* num_elem, hipStream_t s) {
  T2P_REQUIRE(x && z && std && mask_pair && perturbed && mask && num_elem && B > 0 && B <= 65535 && C > 0 && L > 0, "dsm_perturb arguments");
  T2P_REQUIRE(!(cond_flags & 4) || mask_inpaint, "dsm_perturb: the inpainting condition needs mask_inpaint");
  T2P_HIP_CHECK(hipMemsetAsync(num_elem, 0, B * sizeof(float), s));
  const long per = (long)C * L * L;
  hipLaunchKernelGGL(dsm_perturb_kernel, dim3(std::min(64, cdiv_l(per, 256)), B), dim3(256), 0, s, x, z, std, mask_pair, mask_inpaint, cond_flags, C,
                     L * L, perturbed, mask, num_elem);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void dsm_loss_kernel(const float* __restrict__ o, const long ldo, const float* __restrict__ z,
                                                       const float* __restrict__ stdv, const float* __restrict__ inv_sigma,
                                                       const unsigned char* __restrict__ mask, const float* __restrict__ num_elem, const int B,
                                                       const int C, const int HW, double* __restrict__ loss_sum, float* __restrict__ d_o,
                                                       const long ld_do, float* __restrict__ score_nchw) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  const long per = (long)C * HW;
  const float sd = stdv[b], is = inv_sigma[b];
  const float gscale = 2.f * sd * is / ((num_elem[b] + 1e-8f) * (float)B);
  double acc = 0.0;
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < per; j += (long)gridDim.x * 256) {
    const int c = (int)(j / HW), p = (int)(j - (long)c * HW);
    const long i = (long)b * per + j;
    const float sc = o[((long)b * HW + p) * ldo + c] * is;
    if (score_nchw) score_nchw[i] = sc;
    const float r = sc * sd + z[i];
    const float m = mask[i] ? 1.f : 0.f;
    acc += (double)(r * r * m);
    if (d_o) d_o[((long)b * HW + p) * ld_do + c] = gscale * r * m;
  }
  acc = block_sum_256_d(acc, sh);
  if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum + b, acc);
}
int launch_dsm_loss(const float* o, long ldo, const float* z, const float* std, const float* inv_sigma, const unsigned char* mask,
                    const float* num_elem, int B, int C, int L, double* loss_sum, float* d_o, long ld_do, float* score_nchw, hipStream_t s) {
  T2P_REQUIRE(o && z && std && inv_sigma && mask && num_elem && loss_sum && B > 0 && B <= 65535 && C > 0 && L > 0 && ldo >= C, "dsm_loss arguments");
  T2P_REQUIRE(!d_o || ld_do >= C, "dsm_loss: ld_do");
  T2P_HIP_CHECK(hipMemsetAsync(loss_sum, 0, B * sizeof(double), s));
  if (d_o && ld_do > C) T2P_HIP_CHECK(hipMemsetAsync(d_o, 0, (size_t)B * L * L * ld_do * sizeof(float), s));
  const long per = (long)C * L * L;
  hipLaunchKernelGGL(dsm_loss_kernel, dim3(std::min(64, cdiv_l(per, 256)), B), dim3(256), 0, s, o, ldo, z, std, inv_sigma, mask, num_elem, B, C, L * L,
                     loss_sum, d_o, ld_do, score_nchw);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
__global__ void dsm_finish_kernel(const double* __restrict__ loss_sum, const float* __restrict__ num_elem, const int B, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double a = 0.0;
    for (int b = 0; b < B; ++b) a += loss_sum[b] / ((double)num_elem[b] + 1e-8);
    *loss = (float)(a / B);
  }
}
int launch_dsm_finish(const double* loss_sum, const float* num_elem, int B, float* loss, hipStream_t s) {
  T2P_REQUIRE(loss_sum && num_elem && loss && B > 0, "dsm_finish arguments");
  hipLaunchKernelGGL(dsm_finish_kernel, dim3(1), dim3(64), 0, s, loss_sum, num_elem, B, loss);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// =====================================================================================================================================
// optimizer and EMA
// =====================================================================================================================================
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, const long n, double* __restrict__ out) {
  __shared__ double sh[4];
  double a = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = g[i]; a += (double)v * v; }
  a = block_sum_256_d(a, sh);
  if (threadIdx.x == 0) unsafeAtomicAdd(out, a);
}
int launch_sumsq(const float* g, long n, double* out, hipStream_t s) {
  T2P_REQUIRE(g && out && n > 0, "sumsq arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, g, n, out);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  float clip = 1.f;
  if (a.grad_clip >= 0.f) clip = fminf(1.f, a.grad_clip / ((float)sqrt(*a.sumsq) + 1e-6f));
  const float step = a.lr / a.bias1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    float g = a.g[i] * clip;
    a.g[i] = g;                                        // clip_grad_norm_ scales .grad in place
    if (a.weight_decay != 0.f) g += a.weight_decay * a.p[i];
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m; a.v[i] = v;
    a.p[i] -= step * m / (sqrtf(v) / a.bias2_sqrt + a.eps);
  }
}
int launch_adam(const AdamArgs& a, hipStream_t s) {
  T2P_REQUIRE(a.p && a.g && a.m && a.v && a.n > 0 && a.bias1 > 0.f && a.bias2_sqrt > 0.f, "adam arguments");
  T2P_REQUIRE(a.grad_clip < 0.f || a.sumsq, "adam: clipping needs the gradient's sum of squares");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(a.n, 256, 4096)), dim3(256), 0, s, a);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ shadow, const float* __restrict__ p, const float omd, const long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float sv = shadow[i]; shadow[i] = sv - omd * (sv - p[i]); }
}
int launch_ema(float* shadow, const float* p, float one_minus_decay, long n, hipStream_t s) {
  T2P_REQUIRE(shadow && p && n > 0, "ema arguments");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, shadow, p, one_minus_decay, n);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

}  // namespace t2p
