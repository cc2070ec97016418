// Row-block chains of a SpatialTransformer block in ONE launch each (gfx950).
//
// At the 16x16 level of the C = 256 configurations a BasicTransformerBlock (reference model/attention.py:208-215, inside
// SpatialTransformer.forward :250-263) is 13 launches of ~12 us each for 8192 rows: every one of them is bound by the latency
// of a dependent launch, not by its arithmetic.  Between the two attention products the block is ROW-WISE: GroupNorm apply,
// proj_in, LayerNorm, the q | k | v projection (and later out-projection + residual, LayerNorm, the next projection) touch one
// token at a time.  Such a chain runs here as one kernel:
//   * a workgroup (8 wavefronts) owns 32 rows (two MFMA row tiles) of one sample and keeps them in LDS (padded rows: conflict-free
//     16-byte fragment reads) from stage to stage;
//   * a stage's weights [N][K] stream from global memory (L2) straight into MFMA fragments, column tile by column tile, one tile
//     ahead; wavefront w computes column tiles w, w + 8, ... for both row tiles from A fragments it loads once per stage;
//     v_mfma_f32_16x16x32 with the weights first, so a lane ends with 4 consecutive channels of one row;
//   * nothing is exchanged between workgroups; every workgroup re-reads the stage weights (256 x 0.5 MB at C = 256: L2 traffic
//     of a few microseconds), which is what limits the idea to C = 256 and to levels of <= 8192 rows.
// st_entry_kernel: a = GroupNorm(x) (statistics from the producer's per-64-row column sums, folded in double, or x already
// normalised) -> t = proj_in(a) + b -> LayerNorm_1(t) -> q | k | v: the first four launches of the block.  The same kernel with a
// residual and N2 = C is the chain after the self-attention: t += to_out(o) + b -> LayerNorm_2(t) -> to_q of the cross-attention;
// with N2 = 8 C, a bias and the GEGLU epilogue the chain after the cross-attention: t += to_out(o) + b -> LayerNorm_3(t) -> ff.net.0.
#include <algorithm>

#include "t2p_kernels.h"

namespace t2p {

typedef unsigned sf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sf_u32x2 __attribute__((ext_vector_type(2)));
typedef float sf_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sf_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sf_f16x8 __attribute__((ext_vector_type(8)));

template <typename TC> struct SfMma;
template <> struct SfMma<bf16_t> {
  __device__ static inline void run(const sf_u32x4& a, const sf_u32x4& b, sf_f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sf_bf16x8, a), __builtin_bit_cast(sf_bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct SfMma<f16_t> {
  __device__ static inline void run(const sf_u32x4& a, const sf_u32x4& b, sf_f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sf_f16x8, a), __builtin_bit_cast(sf_f16x8, b), c, 0, 0, 0);
  }
};
template <typename TC> __device__ inline unsigned sf_pack2(float a, float b);
template <> __device__ inline unsigned sf_pack2<bf16_t>(float a, float b) {
  return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
}
template <> __device__ inline unsigned sf_pack2<f16_t>(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
template <typename TC> __device__ inline float sf_lo(unsigned u) { return to_f32(__builtin_bit_cast(TC, (uint16_t)(u & 0xffffu))); }
template <typename TC> __device__ inline float sf_hi(unsigned u) { return to_f32(__builtin_bit_cast(TC, (uint16_t)(u >> 16))); }

// sum over the 64 lanes of a wavefront, every lane ends with the total (fixed order)
__device__ inline float sf_wave_sum(float x) {
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) x += __shfl_xor(x, sh, 64);
  return x;
}

constexpr int SF_ROWS = 32;
#ifdef SF_TIMING     // measurement variant (tools/build_variant.py): phase stamps of workgroup 0 (100 MHz clock) over the first bytes of qkv
#define SF_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) sf_stamps[i] = __builtin_amdgcn_s_memrealtime();
#define SF_STAMPS_OUT(p) if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) ((unsigned long long*)(p))[i_] = sf_stamps[i_]; }
#else
#define SF_STAMP(i)
#define SF_STAMPS_OUT(p)
#endif

// Weights of the chain kernels are stored FRAGMENT-MAJOR (sf_frag_major_kernel, once per engine): the 16 bytes lane l needs for column
// tile ct and K-step s sit at ((ct K / 32 + s) 64 + l) 16, so a fragment load is 1 KiB contiguous -- eight whole cache lines.  In the
// row-major [N][K] layout a load touches 16 lines and uses half of each; the other halves belong to the next K-step's load, by which
// time a CU streaming 190 KiB through a 16 KiB L1 has dropped them: every line crossed the L2 -> CU path twice.
template <typename TC>
__global__ void sf_frag_major_kernel(const TC* W, TC* out, int N, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte piece each
  const int ns = K / 32;
  if (i >= (long)(N / 16) * ns * 64) return;
  const int lane = (int)(i & 63), s = (int)((i >> 6) % ns), ct = (int)((i >> 6) / ns);
  const sf_u32x4 v = *(const sf_u32x4*)(W + (long)(ct * 16 + (lane & 15)) * K + 32 * s + 8 * (lane >> 4));
  *(sf_u32x4*)(out + i * 8) = v;
}
int launch_sf_frag_major(int dtype, const void* W, void* out, int N, int K, hipStream_t s) {
  T2P_REQUIRE(W && out && N % 16 == 0 && K % 32 == 0 && (dtype == DT_F16 || dtype == DT_BF16), "frag_major arguments");
  const long n = (long)(N / 16) * (K / 32) * 64;
  hipLaunchKernelGGL((sf_frag_major_kernel<f16_t>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const f16_t*)W, (f16_t*)out, N, K);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// One stage: out[32][N] = A[32][K] W[N][K]^T with A in LDS (row stride RS bytes), N = 128 TPW; W fragment-major.  Wavefront `wave` of 8 takes the TPW
// column tiles wave, wave + 8, ...; `epi(i, ct, acc, bias4)` receives the two row tiles of a finished column tile: acc[rt][e] = row
// 16 rt + (lane & 15), column 16 ct + 4 (lane >> 4) + e.  The weight fragments of a column tile are one L2 round trip away and
// its matrix work is 16 MFMAs: DEPTH tiles are kept in flight (`ring`, filled by sf_prefetch ahead of the stage: a one-tile
// look-ahead left the q | k | v stage waiting for every tile: 25.6 us for the whole chain).
// SWAP: the activations first -- a lane then ends with 4 consecutive ROWS of one column (acc[rt][e] = row 16 rt + 4 (lane >> 4) + e,
// column 16 ct + (lane & 15)): the orientation of a transposed store
template <typename TC, int K, int RS, int TPW, int DEPTH, bool SWAP = false, typename Epi>
__device__ __forceinline__ void sf_stage(const unsigned char* a_lds, const TC* W, const float* bias, const int wave, const int lane,
                                         sf_u32x4 (&ring)[DEPTH][K / 32], float4 (&bring)[DEPTH], Epi&& epi) {
  constexpr int NS = K / 32;
  constexpr bool CACHE = K <= 256;                           // the A fragments of the whole K axis stay in registers (K = 512: re-read per tile)
  const int l16 = lane & 15, g4 = lane >> 4;
  const unsigned char* const ar = a_lds + l16 * RS + 8 * g4 * 2;
  sf_u32x4 af[2][CACHE ? NS : 1];
  if constexpr (CACHE) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int s = 0; s < NS; ++s) af[rt][s] = *(const sf_u32x4*)(ar + rt * 16 * RS + 64 * s);
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int ct = wave + 8 * i;
    sf_f32x4 acc[2] = {sf_f32x4{0.f, 0.f, 0.f, 0.f}, sf_f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (CACHE) {
        if constexpr (SWAP) {
          SfMma<TC>::run(af[0][s], ring[i % DEPTH][s], acc[0]);
          SfMma<TC>::run(af[1][s], ring[i % DEPTH][s], acc[1]);
        } else {
          SfMma<TC>::run(ring[i % DEPTH][s], af[0][s], acc[0]);
          SfMma<TC>::run(ring[i % DEPTH][s], af[1][s], acc[1]);
        }
      } else {
        const sf_u32x4 a0 = *(const sf_u32x4*)(ar + 64 * s), a1 = *(const sf_u32x4*)(ar + 16 * RS + 64 * s);
        SfMma<TC>::run(ring[i % DEPTH][s], a0, acc[0]);
        SfMma<TC>::run(ring[i % DEPTH][s], a1, acc[1]);
        if ((s & 3) == 3) asm volatile("" ::: "memory");      // (keeps the compiler from hoisting a whole tile's fragment reads: spills)
      }
    }
    const float4 bb = bring[i % DEPTH];                     // this tile's bias (the lane's 4 columns; zeros without one)
    if (i + DEPTH < TPW) {                                  // this slot's fragments are consumed: request the tile DEPTH ahead
      const TC* wr = W + ((long)(ct + 8 * DEPTH) * NS * 64 + lane) * 8;
#pragma unroll
      for (int s = 0; s < NS; ++s) ring[i % DEPTH][s] = *(const sf_u32x4*)(wr + 512 * s);
      if (bias) bring[i % DEPTH] = *(const float4*)(bias + (ct + 8 * DEPTH) * 16 + 4 * g4);
    }
    epi(i, ct, acc, bb);
  }
}
// the fragments of this wavefront's first min(DEPTH, TPW) column tiles of W [N][K] (what sf_stage expects in `ring`)
template <typename TC, int K, int TPW, int DEPTH>
__device__ __forceinline__ void sf_prefetch(const TC* W, const float* bias, const int wave, const int lane, sf_u32x4 (&ring)[DEPTH][K / 32],
                                            float4 (&bring)[DEPTH]) {
  const int l16 = lane & 15, g4 = lane >> 4;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) {
    bring[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < TPW) {
      const TC* wr = W + ((long)(wave + 8 * i) * (K / 32) * 64 + lane) * 8;
#pragma unroll
      for (int s = 0; s < K / 32; ++s) ring[i][s] = *(const sf_u32x4*)(wr + 512 * s);
      if (bias) bring[i] = *(const float4*)(bias + (wave + 8 * i) * 16 + 4 * g4);
    }
  }
}

// LayerNorm of the 32 rows of `src` into `dst` (may be the same): wavefront w takes rows 4 w .. 4 w + 3, a lane C / 64 consecutive channels
template <typename TC, int C, int RS>
__device__ __forceinline__ void sf_layernorm(const unsigned char* src, unsigned char* dst, const float (&ga)[C / 64], const float (&be)[C / 64],
                                             const float eps, const int wave, const int lane) {
  constexpr int PER = C / 64;                                // 4 (C = 256) or 8 (C = 512) channels per lane; ga / be: its gamma / beta
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned char* row = src + (wave * 4 + r) * RS + lane * PER * 2;
    unsigned char* orow = dst + (wave * 4 + r) * RS + lane * PER * 2;
    float v[PER];
    if constexpr (PER == 4) {
      const sf_u32x2 u = *(const sf_u32x2*)row;
      v[0] = sf_lo<TC>(u[0]); v[1] = sf_hi<TC>(u[0]); v[2] = sf_lo<TC>(u[1]); v[3] = sf_hi<TC>(u[1]);
    } else {
      const sf_u32x4 u = *(const sf_u32x4*)row;
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[2 * k] = sf_lo<TC>(u[k]); v[2 * k + 1] = sf_hi<TC>(u[k]); }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) s += v[k];
    const float mean = sf_wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const float d = v[k] - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(sf_wave_sum(q) * (1.f / C) + eps);
    float y[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) y[k] = (v[k] - mean) * rstd * ga[k] + be[k];
    if constexpr (PER == 4) {
      *(sf_u32x2*)orow = sf_u32x2{sf_pack2<TC>(y[0], y[1]), sf_pack2<TC>(y[2], y[3])};
    } else {
      *(sf_u32x4*)orow = sf_u32x4{sf_pack2<TC>(y[0], y[1]), sf_pack2<TC>(y[2], y[3]), sf_pack2<TC>(y[4], y[5]), sf_pack2<TC>(y[6], y[7])};
    }
  }
}

// erf-GELU of the fused GEGLU epilogue: Abramowitz-Stegun 7.1.26 (|error of erf| <= 1.5e-7), as gemm.hip's gelu_erf_fast
__device__ inline float sf_gelu(float g) {
  const float x = g * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  return 0.5f * g * (1.f + copysignf(fmaf(-q * t, e, 1.f), x));
}

// TPW2: column tiles of the second product per wavefront (N2 = 128 TPW2: 3 C, C, or 8 C with GEGLU: interleaved (value, gate)
// columns, out2[row][j] = value_j gelu(gate_j), N2 / 2 columns -- FeedForward's first layer, model/attention.py:37-64)
// FFPO (with GEGLU): the gated product stays in LDS and a third product follows -- ff.net.2 and proj_out as ONE matrix over [g | t]
// (Layer::ffpo, attention.py:213-215, 259-263) + bias + the block input as residual -> the block's output y and its per-64-row
// column sums (two workgroups share a chunk: two atomic adds into a slot the block's entry kernel zeroed -- commutative, so the
// result does not depend on their order).
template <typename TC, int C, int TPW2, bool GEGLU, bool FFPO>
__global__ __launch_bounds__(512) void st_entry_kernel(const StEntryArgs a) {
  static_assert(!FFPO || GEGLU, "the third product follows the GEGLU form");
  constexpr int RS = C * 2 + 16;                             // LDS row stride: an odd number of 16-byte units
  constexpr int RSG = 4 * C * 2 + 16;                        // row stride of the gated product g [32][4 C]
  constexpr int NS = C / 32;
  // LDS (dynamic: 78 KiB at C = 512): the two row buffers, the GroupNorm fold, [FFPO: the gated product g [32][RSG]]
  extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
  unsigned char* const bufx = sf_smem;
  unsigned char* const buft = sf_smem + SF_ROWS * RS;
  double (*const dred)[C] = (double (*)[C])(sf_smem + 2 * SF_ROWS * RS);
  float* const gsc = (float*)(sf_smem + 2 * SF_ROWS * RS + 2 * C * 8);
  float* const gsh = gsc + C;
  unsigned char* const gbuf = sf_smem + 2 * SF_ROWS * RS + 2 * C * 8 + 2 * C * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * SF_ROWS;
  const int b = m0 / a.n;                                    // n % 32 == 0: the rows of a workgroup lie in one sample
  const TC* Win = (const TC*)a.w_in;
  const TC* Wqkv = (const TC*)a.w_qkv;
#ifdef SF_TIMING
  unsigned long long sf_stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  SF_STAMP(0)
  if (a.zero) {                                              // (entry kernel of a block whose last kernel accumulates column sums)
    for (long i = (long)blockIdx.x * 512 + tid; i < a.zero_n; i += (long)gridDim.x * 512) a.zero[i] = 0.f;
  }

  // independent of everything: this wavefront's proj_in fragments (both of its column tiles) and the rows themselves
  constexpr int DEPTH = C <= 256 ? 3 : 2;                   // column tiles of weights in flight (C = 512: 2 x 16 KiB per wavefront)
  sf_u32x4 ring[DEPTH][NS];
  float4 bring[DEPTH];
  sf_prefetch<TC, C, C / 128, DEPTH>(Win, a.b_in, wave, lane, ring, bring);
  constexpr int XPT = SF_ROWS * (C / 8) / 512;               // 16-byte pieces of the row block per thread
  sf_u32x4 xr[XPT];
  {
    const TC* X = (const TC*)a.x + (long)m0 * C;
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int i = tid + 512 * k, r = i / (C / 8), j = i - r * (C / 8);
      xr[k] = *(const sf_u32x4*)(X + (long)r * C + j * 8);
    }
  }

  // ... and every other operand that depends on nothing: a workgroup is a chain of dependent round trips (the first version loaded
  // the column sums chunk by chunk and gamma / beta / biases where they were used: 8 serial trips, 25.6 us per launch)
  constexpr int PER = C / 64;
  float lga[PER], lbe[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) { lga[k] = a.ln_gamma[lane * PER + k]; lbe[k] = a.ln_beta[lane * PER + k]; }
  // residual of the first product: this lane's 4 channels of its two rows per tile (C = 512: loaded in the epilogue, registers)
  constexpr bool RES_EARLY = C <= 256;
  sf_u32x2 rres[RES_EARLY ? C / 128 : 1][2];
  if constexpr (RES_EARLY) {
#pragma unroll
    for (int i = 0; i < C / 128; ++i)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        rres[i][rt] = a.res ? *(const sf_u32x2*)((const TC*)a.res + (long)(m0 + rt * 16 + l16) * C + (wave + 8 * i) * 16 + 4 * g4) : sf_u32x2{0u, 0u};
  }
  float4 b3v[2];
  sf_u32x2 r3v[2][2];
  if constexpr (FFPO) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave + 8 * i) * 16 + 4 * g4;
      b3v[i] = *(const float4*)(a.b3 + col);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) r3v[i][rt] = *(const sf_u32x2*)((const TC*)a.res3 + (long)(m0 + rt * 16 + l16) * C + col);
    }
  }
  // ---- GroupNorm scale / shift of this sample (gn_apply_cols_kernel's arithmetic: column sums folded in double) ----------------
  if (a.cstats) {
    const int nchunk = a.n >> 6, cpg = C / a.groups;
    const int c = tid & (C - 1), part = tid / C;             // 512 / C threads per channel, each every (512 / C)-th chunk, 4 loads in flight
    constexpr int PARTS = 512 / C;
    const float ggam = a.gn_gamma[c], gbet = a.gn_beta[c];
    double s = 0, q = 0;
    const float* cs = a.cstats + ((long)b * nchunk * C + c) * 2;
    int ch = part;
    for (; ch + 3 * PARTS < nchunk; ch += 4 * PARTS) {
      const float2 v0 = *(const float2*)(cs + (long)ch * C * 2), v1 = *(const float2*)(cs + (long)(ch + PARTS) * C * 2);
      const float2 v2 = *(const float2*)(cs + (long)(ch + 2 * PARTS) * C * 2), v3 = *(const float2*)(cs + (long)(ch + 3 * PARTS) * C * 2);
      s += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      q += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; ch < nchunk; ch += PARTS) {
      const float2 v = *(const float2*)(cs + (long)ch * C * 2);
      s += v.x; q += v.y;
    }
    if (part > 0) { dred[0][c] = s; dred[1][c] = q; }       // (C = 256: two parts; the first adds the second's sums)
    __syncthreads();
    if (part == 0) {
      if (PARTS > 1) { s += dred[0][c]; q += dred[1][c]; }
    }
    __syncthreads();
    if (part == 0) { dred[0][c] = s; dred[1][c] = q; }
    __syncthreads();
    if (part == 0) {
      const int g0 = (c / cpg) * cpg;
      double gs = 0, gq = 0;
      for (int k = 0; k < cpg; ++k) { gs += dred[0][g0 + k]; gq += dred[1][g0 + k]; }
      const double cnt = (double)a.n * cpg, mean = gs / cnt;
      double var = gq / cnt - mean * mean;
      if (var < 0) var = 0;
      const float rstd = (float)(1.0 / sqrt(var + (double)a.gn_eps));
      const float sc = rstd * ggam;
      gsc[c] = sc;
      gsh[c] = gbet - (float)mean * sc;
    }
    __syncthreads();
  }
  SF_STAMP(1)
  // ---- rows -> LDS (normalised on the way unless the producer already did) ---------------------------------------------------------
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int i = tid + 512 * k, r = i / (C / 8), j = i - r * (C / 8);
    sf_u32x4 u = xr[k];
    if (a.cstats) {
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = j * 8 + 2 * q;
        o[q] = sf_pack2<TC>(sf_lo<TC>(u[q]) * gsc[c] + gsh[c], sf_hi<TC>(u[q]) * gsc[c + 1] + gsh[c + 1]);
      }
      u = sf_u32x4{o[0], o[1], o[2], o[3]};
    }
    *(sf_u32x4*)(bufx + r * RS + j * 16) = u;
  }
  __syncthreads();
  SF_STAMP(2)
  // ---- t = proj_in(a) + bias: to global memory (the block's residual stream) and to LDS -----------------------------------------
  {
    TC* T = (TC*)a.t;
    sf_stage<TC, C, RS, C / 128, DEPTH>(bufx, Win, a.b_in, wave, lane, ring, bring, [&](int i, int ct, sf_f32x4 (&acc)[2], const float4 bb) {
      const int col = ct * 16 + 4 * g4;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int r = rt * 16 + l16;
        sf_u32x2 rr = sf_u32x2{0u, 0u};                      // (zeros without a residual)
        if constexpr (RES_EARLY) rr = rres[i][rt];
        else if (a.res) rr = *(const sf_u32x2*)((const TC*)a.res + (long)(m0 + r) * C + col);
        const sf_u32x2 o = {sf_pack2<TC>(acc[rt][0] + bb.x + sf_lo<TC>(rr[0]), acc[rt][1] + bb.y + sf_hi<TC>(rr[0])),
                            sf_pack2<TC>(acc[rt][2] + bb.z + sf_lo<TC>(rr[1]), acc[rt][3] + bb.w + sf_hi<TC>(rr[1]))};
        *(sf_u32x2*)(buft + r * RS + col * 2) = o;
        *(sf_u32x2*)(T + (long)(m0 + r) * C + col) = o;
      }
    });
  }
  SF_STAMP(3)
  sf_prefetch<TC, C, TPW2, DEPTH>(Wqkv, a.b2, wave, lane, ring, bring);      // the next stage's first fragments travel during the LayerNorm
  __syncthreads();
  SF_STAMP(4)
  sf_layernorm<TC, C, RS>(buft, bufx, lga, lbe, a.ln_eps, wave, lane);      // (bufx is free: t itself stays in buft for the third product)
  __syncthreads();
  SF_STAMP(5)
  // ---- q | k | v = LayerNorm_1(t) W_qkv^T (CrossAttention.to_q / to_k / to_v carry no bias) -------------------------------------
  {
    TC* Q = (TC*)a.qkv;
    sf_stage<TC, C, RS, TPW2, DEPTH>(bufx, Wqkv, a.b2, wave, lane, ring, bring, [&](int i, int ct, sf_f32x4 (&acc)[2], const float4 bb) {
      const int col = ct * 16 + 4 * g4;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int r = rt * 16 + l16;
        if constexpr (GEGLU) {
          const unsigned gg = sf_pack2<TC>((acc[rt][0] + bb.x) * sf_gelu(acc[rt][1] + bb.y), (acc[rt][2] + bb.z) * sf_gelu(acc[rt][3] + bb.w));
          if constexpr (FFPO) *(unsigned*)(gbuf + r * RSG + col) = gg;      // (col >> 1) elements = col bytes
          else *(unsigned*)(Q + (long)(m0 + r) * (64 * TPW2) + (col >> 1)) = gg;
        } else {
          *(sf_u32x2*)(Q + (long)(m0 + r) * (128 * TPW2) + col) =
              sf_u32x2{sf_pack2<TC>(acc[rt][0] + bb.x, acc[rt][1] + bb.y), sf_pack2<TC>(acc[rt][2] + bb.z, acc[rt][3] + bb.w)};
        }
      }
    });
  }
  SF_STAMP(7)
  if constexpr (FFPO) {
    // ---- y = [g | t] W_3^T + b_3 + x: K = 4 C (g, LDS) + C (t, LDS); this wavefront's two column tiles share the A fragments ----------
    constexpr int NS3 = 5 * C / 32, NSG = 4 * C / 32, D3 = 8;     // 16 weight fragments in flight per wavefront (4: 13.3 us for this product, 48 GB/s)
    const TC* w0p = (const TC*)a.w3 + ((long)wave * NS3 * 64 + lane) * 8;
    const TC* w1p = (const TC*)a.w3 + ((long)(wave + 8) * NS3 * 64 + lane) * 8;
    sf_u32x4 q0[D3], q1[D3];
#pragma unroll
    for (int j = 0; j < D3; ++j) { q0[j] = *(const sf_u32x4*)(w0p + 512 * j); q1[j] = *(const sf_u32x4*)(w1p + 512 * j); }
    __syncthreads();                                         // g is complete
    sf_f32x4 acc3[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc3[i][rt] = sf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s3 = 0; s3 < NS3; ++s3) {
      const unsigned char* ab = s3 < NSG ? gbuf + l16 * RSG + (32 * s3 + 8 * g4) * 2 : buft + l16 * RS + (32 * (s3 - NSG) + 8 * g4) * 2;
      const sf_u32x4 a0 = *(const sf_u32x4*)ab, a1 = *(const sf_u32x4*)(ab + 16 * (s3 < NSG ? RSG : RS));
      const sf_u32x4 w0 = q0[s3 % D3], w1 = q1[s3 % D3];
      if (s3 + D3 < NS3) { q0[s3 % D3] = *(const sf_u32x4*)(w0p + 512 * (s3 + D3)); q1[s3 % D3] = *(const sf_u32x4*)(w1p + 512 * (s3 + D3)); }
      SfMma<TC>::run(w0, a0, acc3[0][0]); SfMma<TC>::run(w0, a1, acc3[0][1]);
      SfMma<TC>::run(w1, a0, acc3[1][0]); SfMma<TC>::run(w1, a1, acc3[1][1]);
    }
    TC* Y = (TC*)a.y;
    const long chunk = m0 >> 6;                              // n % 64 == 0: global row / 64 is the consumer's chunk index
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (wave + 8 * i) * 16 + 4 * g4;
      const float bq[4] = {b3v[i].x, b3v[i].y, b3v[i].z, b3v[i].w};
      float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const sf_u32x2 rr = r3v[i][rt];
        const float v0 = acc3[i][rt][0] + bq[0] + sf_lo<TC>(rr[0]), v1 = acc3[i][rt][1] + bq[1] + sf_hi<TC>(rr[0]);
        const float v2 = acc3[i][rt][2] + bq[2] + sf_lo<TC>(rr[1]), v3 = acc3[i][rt][3] + bq[3] + sf_hi<TC>(rr[1]);
        *(sf_u32x2*)(Y + (long)(m0 + rt * 16 + l16) * C + col) = sf_u32x2{sf_pack2<TC>(v0, v1), sf_pack2<TC>(v2, v3)};
        cs[0] += v0; cs[1] += v1; cs[2] += v2; cs[3] += v3;
        cq[0] += v0 * v0; cq[1] += v1 * v1; cq[2] += v2 * v2; cq[3] += v3 * v3;
      }
      if (a.y_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s1 = cs[e], s2 = cq[e];
#pragma unroll
          for (int sh = 1; sh < 16; sh <<= 1) { s1 += __shfl_xor(s1, sh, 64); s2 += __shfl_xor(s2, sh, 64); }
          if (l16 == 0) {
            float* dst = a.y_stats + (chunk * C + col + e) * 2;
            atomicAdd(dst, s1);
            atomicAdd(dst + 1, s2);
          }
        }
      }
    }
  }
  SF_STAMP(6)
  SF_STAMPS_OUT(a.qkv)
}

// ---- the projections of an AttnBlockpp in one launch (C = 256): h = GroupNorm(x) -> q | k = NIN_0 | NIN_1 (h) row-major, and
// V^T = (NIN_2 . NIN_3)(h)^T, written transposed straight from the accumulators (layers.py:160-167; the engine's merged value projection).
// Replaces the GroupNorm apply (when the producer did not already apply it), the q | k GEMM and the batched transposed projection.
template <typename TC, int C>
__global__ __launch_bounds__(512) void attn_proj_kernel(const AttnProjArgs a) {
  constexpr int RS = C * 2 + 16, NS = C / 32, DEPTH = 3;
  __shared__ __attribute__((aligned(16))) unsigned char bufx[SF_ROWS * RS];
  __shared__ double dred[2][C];
  __shared__ float gsc[C], gsh[C];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const int m0 = blockIdx.x * SF_ROWS;
  const int b = m0 / a.n;
  sf_u32x4 ring[DEPTH][NS];
  float4 bring[DEPTH];
  sf_prefetch<TC, C, 2 * C / 128, DEPTH>((const TC*)a.w_qk, a.b_qk, wave, lane, ring, bring);
  constexpr int XPT = SF_ROWS * (C / 8) / 512;
  sf_u32x4 xr[XPT];
  {
    const TC* X = (const TC*)a.x + (long)m0 * C;
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int i = tid + 512 * k, r = i / (C / 8), j = i - r * (C / 8);
      xr[k] = *(const sf_u32x4*)(X + (long)r * C + j * 8);
    }
  }
  if (a.cstats) {              // (st_entry_kernel's fold)
    const int nchunk = a.n >> 6, cpg = C / a.groups;
    const int c = tid & (C - 1), part = tid / C;
    constexpr int PARTS = 512 / C;
    const float ggam = a.gn_gamma[c], gbet = a.gn_beta[c];
    double s = 0, q = 0;
    const float* cs = a.cstats + ((long)b * nchunk * C + c) * 2;
    for (int ch = part; ch < nchunk; ch += PARTS) {
      const float2 v = *(const float2*)(cs + (long)ch * C * 2);
      s += v.x; q += v.y;
    }
    if (part > 0) { dred[0][c] = s; dred[1][c] = q; }
    __syncthreads();
    if (part == 0 && PARTS > 1) { s += dred[0][c]; q += dred[1][c]; }
    __syncthreads();
    if (part == 0) { dred[0][c] = s; dred[1][c] = q; }
    __syncthreads();
    if (part == 0) {
      const int g0 = (c / cpg) * cpg;
      double gs = 0, gq = 0;
      for (int k = 0; k < cpg; ++k) { gs += dred[0][g0 + k]; gq += dred[1][g0 + k]; }
      const double cnt = (double)a.n * cpg, mean = gs / cnt;
      double var = gq / cnt - mean * mean;
      if (var < 0) var = 0;
      const float sc = (float)(1.0 / sqrt(var + (double)a.gn_eps)) * ggam;
      gsc[c] = sc;
      gsh[c] = gbet - (float)mean * sc;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int i = tid + 512 * k, r = i / (C / 8), j = i - r * (C / 8);
    sf_u32x4 u = xr[k];
    if (a.cstats) {
      unsigned o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = j * 8 + 2 * q;
        o[q] = sf_pack2<TC>(sf_lo<TC>(u[q]) * gsc[c] + gsh[c], sf_hi<TC>(u[q]) * gsc[c + 1] + gsh[c + 1]);
      }
      u = sf_u32x4{o[0], o[1], o[2], o[3]};
    }
    *(sf_u32x4*)(bufx + r * RS + j * 16) = u;
  }
  __syncthreads();
  // q | k (+ bias), row-major [rows][2 C]
  {
    TC* Q = (TC*)a.qk;
    sf_stage<TC, C, RS, 2 * C / 128, DEPTH>(bufx, (const TC*)a.w_qk, a.b_qk, wave, lane, ring, bring,
                                            [&](int i, int ct, sf_f32x4 (&acc)[2], const float4 bb) {
      const int col = ct * 16 + 4 * g4;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const sf_u32x2 o = {sf_pack2<TC>(acc[rt][0] + bb.x, acc[rt][1] + bb.y), sf_pack2<TC>(acc[rt][2] + bb.z, acc[rt][3] + bb.w)};
        if (a.k_fm && col >= C) {              // the k half fragment-major per sample (GemmParams::c_frag's order): an 8-byte half of a chunk
          const int c = col - C, rl = m0 - b * a.n + rt * 16 + l16;
          *(sf_u32x2*)((TC*)a.k_fm + (long)b * a.n * C + ((long)(rl >> 5) * (C / 32) + (c >> 5)) * 1024 + ((c >> 3) & 1) * 512 + ((c >> 4) & 1) * 256 +
                       (rl & 31) * 8 + (c & 7)) = o;
        } else {
          *(sf_u32x2*)(Q + (long)(m0 + rt * 16 + l16) * (2 * C) + col) = o;
        }
      }
    });
  }
  // V^T [B][C][npad]: activations first, so a lane holds 4 consecutive keys of one channel: an 8-byte piece of a V^T row
  sf_prefetch<TC, C, C / 128, DEPTH>((const TC*)a.w_v, nullptr, wave, lane, ring, bring);
  {
    TC* VT = (TC*)a.vt + (long)b * C * a.npad + (m0 - b * a.n);
    sf_stage<TC, C, RS, C / 128, DEPTH, true>(bufx, (const TC*)a.w_v, nullptr, wave, lane, ring, bring,
                                              [&](int i, int ct, sf_f32x4 (&acc)[2], const float4) {
      const int ch = ct * 16 + l16;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const sf_u32x2 o = {sf_pack2<TC>(acc[rt][0], acc[rt][1]), sf_pack2<TC>(acc[rt][2], acc[rt][3])};
        if (a.k_fm) {                          // V^T fragment-major too: rows = channels, columns = keys (npad == n)
          const int key = m0 - b * a.n + rt * 16 + 4 * g4;
          *(sf_u32x2*)((TC*)a.vt + (long)b * a.n * C + ((long)(ch >> 5) * (a.n >> 5) + (key >> 5)) * 1024 + ((key >> 3) & 1) * 512 +
                       ((key >> 4) & 1) * 256 + (ch & 31) * 8 + (key & 7)) = o;
        } else {
          *(sf_u32x2*)(VT + (long)ch * a.npad + rt * 16 + 4 * g4) = o;
        }
      }
    });
  }
}

bool attn_proj_eligible(const AttnProjArgs& a) {
  if (!g_st_fuse || (a.dtype != DT_F16 && a.dtype != DT_BF16) || a.C != 256) return false;
  if (a.n % SF_ROWS != 0 || (long)a.B * a.n > 8192 || a.npad % 4 != 0 || a.npad < a.n) return false;
  if (a.cstats && (a.n % 64 != 0 || a.groups <= 0 || a.C % a.groups != 0)) return false;
  if (a.k_fm && a.npad != a.n) return false;
  return true;
}
int launch_attn_proj(const AttnProjArgs& a, hipStream_t s) {
  T2P_REQUIRE(attn_proj_eligible(a) && a.x && a.w_qk && a.b_qk && a.w_v && a.qk && a.vt, "attn_proj arguments");
  T2P_REQUIRE(!a.cstats || (a.gn_gamma && a.gn_beta), "attn_proj: GroupNorm parameters");
  const dim3 grid((unsigned)((long)a.B * a.n / SF_ROWS));
  if (a.dtype == DT_F16) hipLaunchKernelGGL((attn_proj_kernel<f16_t, 256>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((attn_proj_kernel<bf16_t, 256>), grid, dim3(512), 0, s, a);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

bool g_st_fuse = true;      // plan switch 39
bool st_entry_eligible(const StEntryArgs& a) {
  // C = 256 only.  (The C = 512 instantiation of round 3 -- A fragments re-read from LDS per column tile, two column tiles of weights in
  // flight -- measured EQUAL to the separate launches at cfg2 and cfg4, twice, and was removed in round 4: DESIGN.md section 8.)
  if (!g_st_fuse || (a.dtype != DT_F16 && a.dtype != DT_BF16) || a.C != 256) return false;
  // one round of workgroups (two rounds measured slower: cfg4's 32x32 level); a workgroup's 32 rows lie in one sample, except when
  // nothing per-sample is involved (no GroupNorm inside, no column sums out): then any split of the rows will do (4 x 4 maps)
  const bool per_sample = a.cstats || a.y_stats;
  if ((long)a.B * a.n > 8192 || ((long)a.B * a.n) % SF_ROWS != 0 || (per_sample && a.n % SF_ROWS != 0)) return false;
  if (a.geglu ? a.n2 != 8 * a.C : (a.n2 != a.C && a.n2 != 3 * a.C)) return false;
  if (a.w3 && (!a.geglu || !a.b3 || !a.res3 || !a.y || (a.y_stats && a.n % 64 != 0))) return false;
  if (a.cstats && (a.n % 64 != 0 || a.groups <= 0 || a.C % a.groups != 0)) return false;
  return true;
}
int launch_st_entry(const StEntryArgs& a, hipStream_t s) {
  T2P_REQUIRE(st_entry_eligible(a) && a.x && a.w_in && a.b_in && a.ln_gamma && a.ln_beta && a.w_qkv && a.t, "st_entry arguments");
  T2P_REQUIRE(!a.cstats || (a.gn_gamma && a.gn_beta), "st_entry: GroupNorm parameters");
  T2P_REQUIRE(a.w3 || a.qkv, "st_entry: output of the second product");
  const dim3 grid((unsigned)((long)a.B * a.n / SF_ROWS));
#define T2P_SF(CC, TPW, GG, FF)                                                                                          \
  {                                                                                                                      \
    constexpr int smem = 2 * SF_ROWS * (CC * 2 + 16) + 2 * CC * 8 + 2 * CC * 4 + (FF ? SF_ROWS * (4 * CC * 2 + 16) : 0);  \
    if (a.dtype == DT_F16) {                                                                                             \
      auto kern = st_entry_kernel<f16_t, CC, TPW, GG, FF>;                                                               \
      T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));                                                              \
      hipLaunchKernelGGL(kern, grid, dim3(512), smem, s, a);                                                             \
    } else {                                                                                                             \
      auto kern = st_entry_kernel<bf16_t, CC, TPW, GG, FF>;                                                              \
      T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));                                                              \
      hipLaunchKernelGGL(kern, grid, dim3(512), smem, s, a);                                                             \
    }                                                                                                                    \
  }
  if (a.w3) T2P_SF(256, 16, true, true) else if (a.geglu) T2P_SF(256, 16, true, false) else if (a.n2 == 3 * a.C) T2P_SF(256, 6, false, false)
  else T2P_SF(256, 2, false, false)
#undef T2P_SF
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

}  // namespace t2p
