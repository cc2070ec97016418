// Fused multi-head attention (flash style) for gfx950: softmax(q k^T * scale) v without
// materialising the score matrix.  Replaces, for 16-bit compute dtypes and head dims 32/64/128,
// the einsum -> softmax -> einsum chain of CrossAttention.forward (reference model/attention.py:
// 181-191) for both the self-attention (attn1) and the text cross-attention (attn2).
//
// Orientation: S^T = K Q^T, so the 32x32 MFMA accumulator has the QUERY on the lane and 16 keys in
// the lane's registers (the other 16 keys of the tile sit in lane ^ 32).  The row softmax is then
// 16 in-register operations plus ONE wavefront shuffle (xor 32) per statistic, and the
// accumulator already is the B operand of the next product O^T = V^T P^T (k order permuted
// consistently in both operands: element j of lane half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3)),
// whose A operand comes from the V^T tile the engine produces anyway.  O^T keeps the query on
// the lane, so the online-softmax rescale is a per-lane scalar.
//
// One workgroup = 4 wavefronts x 32 queries; K and V^T tiles of 64 keys are staged
// global -> registers -> LDS (padded rows: conflict-free ds_read_b128 / ds_read_b64),
// double-buffered.
//
// VROW (self-attention): V is taken ROW-major, [key][head * D + d] -- the third column block of the block's one q|k|v
// projection GEMM -- instead of the separately projected V^T.  Its tile is staged like the K tile and the A fragment of
// O^T = V^T P^T comes out of LDS through ds_read_b64_tr_b16, gfx950's transposing read: per 16-lane group a block of 4 rows
// (keys) x 16 columns (d) is delivered column-major, i.e. lane (d) receives its 4 keys -- exactly the fragment's halves.
// Lane 4 q + p of a group supplies the address of key row q, columns 4 p .. 4 p + 3; rows are VRS bytes apart, an
// odd multiple of 16 dwords, which puts the four key rows of a block into four different 16-bank windows: conflict-free.
#include "t2p_kernels.h"

namespace t2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

template <typename TC> struct AMma;
template <> struct AMma<bf16_t> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct AMma<f16_t> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

template <typename TC> __device__ inline uint32_t pack2(float a, float b);
template <> __device__ inline uint32_t pack2<bf16_t>(float a, float b) {
  return (uint32_t)f32_to_bf16_bits(a) | ((uint32_t)f32_to_bf16_bits(b) << 16);
}
template <> __device__ inline uint32_t pack2<f16_t>(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, v);
}

// XCD-aware block order.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (blocks b and b + 8 share one), and each
// XCD has its own L2: workgroups that re-read the same keys / values (the query blocks of one (sample, head); the strips of one
// sample) must sit on ONE XCD, or every XCD's L2 fetches every K / V tile from memory (measured: L2 hit rate 0.34 - 0.66, 856 MB
// of HBM traffic per launch of the wide-head kernel against 134 MB algorithmic).  Linear id L -> logical id (L & 7) * (T / 8) +
// (L >> 3): each XCD gets a contiguous range of logical ids (T % 8 == 0; identity otherwise).
__device__ __forceinline__ int xcd_logical_id(int L, int T) {
  return (T & 7) ? L : (L & 7) * (T >> 3) + (L >> 3);
}

struct FlashArgs {
  const void* q; long ldq; long sq_b;       // [B][nq][ldq], head h at column h * D
  const void* k; long ldk; long sk_b;       // [B][nk][ldk]
  const void* vt; long ldvt; long svt_b;    // [B][heads * D][ldvt]; VROW: v = [B][nk][ldvt], head h at column h * D
  void* out; long ldo; long so_b;           // [B][nq][ldo]
  int nq, nk;
  float scale_log2e;                        // scale * log2(e)
};

typedef short v4s_t __attribute__((ext_vector_type(4)));

template <typename TC, int D, bool VROW>
__global__ __launch_bounds__(256, 2) void attn_flash_kernel(const FlashArgs a) {
  constexpr int BKV = 64;
  constexpr int KS = D * 2 + 16;            // K tile row stride (bytes)
  constexpr int VS = BKV * 2 + 8;           // V^T tile row stride (bytes)
  constexpr int VRS = D == 32 ? 64 : (D == 64 ? 192 : 320);   // VROW: V tile row stride (bytes): an odd multiple of 16 dwords
  static_assert((VRS / 4) % 32 == 16 && VRS >= D * 2 && VRS % 16 == 0, "row stride of the transposed-read V tile");
  constexpr int KBYTES = BKV * KS, VBYTES = VROW ? BKV * VRS : D * VS;
  constexpr int NS = D / 16;                // k-steps of the score product
  constexpr int NT = D / 32;                // 32-row tiles of O^T
  constexpr int KV = D / 32;                // staging vectors per thread for each of K and V^T
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][K tile | V^T tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // logical (query block, head, sample) with the query block fastest: the query blocks of one (sample, head) share an XCD's L2
  const int lid = xcd_logical_id((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)(gridDim.x * gridDim.y * gridDim.z));
  const int qblk = lid % (int)gridDim.x, head = (lid / (int)gridDim.x) % (int)gridDim.y, b = lid / (int)(gridDim.x * gridDim.y);
  const int q0 = qblk * 128 + wave * 32;
  const TC* Q = (const TC*)a.q + (long)b * a.sq_b + (long)head * D;
  const TC* K = (const TC*)a.k + (long)b * a.sk_b + (long)head * D;
  const TC* VT = (const TC*)a.vt + (long)b * a.svt_b + (VROW ? (long)head * D : (long)head * D * a.ldvt);

  // Q fragments (B operand of S^T = K Q^T): lane (query lr, half lh) holds Q[q][16 s + 8 lh .. + 7]
  uint4 qf[NS];
  {
    const int qrow = q0 + lr;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      qf[s] = qrow < a.nq ? *(const uint4*)(Q + (long)qrow * a.ldq + 16 * s + 8 * lh) : make_uint4(0, 0, 0, 0);
  }

  // staging: K tile = 64 rows x (D/8) 16-byte chunks, V^T tile = D rows x 8 chunks
  uint4 rk[KV], rv[KV];
  auto gload = [&](int t) {
    const int key0 = t * BKV;
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (D / 8), ch = idx % (D / 8);
      const int key = key0 + row;
      rk[i] = key < a.nk ? *(const uint4*)(K + (long)key * a.ldk + ch * 8) : make_uint4(0, 0, 0, 0);
      if constexpr (VROW) {                   // same (key, chunk) as the K tile; keys beyond nk are zero (their P is)
        rv[i] = key < a.nk ? *(const uint4*)(VT + (long)key * a.ldvt + ch * 8) : make_uint4(0, 0, 0, 0);
        continue;
      }
      const int vrow = idx >> 3, vch = idx & 7;
      const int kc = key0 + vch * 8;          // first key of this chunk
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kc < a.nk) {
        v = *(const uint4*)(VT + (long)vrow * a.ldvt + kc);
        if (kc + 8 > a.nk) {                  // ragged tail: padding columns may hold anything
          union { uint4 u; unsigned short e[8]; } x;
          x.u = v;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (kc + e >= a.nk) x.e[e] = 0;
          v = x.u;
        }
      }
      rv[i] = v;
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* kb = smem + buf * (KBYTES + VBYTES);
    unsigned char* vb = kb + KBYTES;
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / (D / 8), ch = idx % (D / 8);
      *(uint4*)(kb + row * KS + ch * 16) = rk[i];
      if constexpr (VROW) {
        *(uint4*)(vb + row * VRS + ch * 16) = rv[i];
        continue;
      }
      const int vrow = idx >> 3, vch = idx & 7;
      uint2* dst = (uint2*)(vb + vrow * VS + vch * 16);     // rows are 8-byte aligned only
      dst[0] = make_uint2(rv[i].x, rv[i].y);
      dst[1] = make_uint2(rv[i].z, rv[i].w);
    }
  };

  f32x16 o[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) o[t][v] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int ntile = (a.nk + BKV - 1) / BKV;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) gload(t + 1);
    const unsigned char* kb = smem + buf * (KBYTES + VBYTES);
    const unsigned char* vb = kb + KBYTES;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kbase = t * BKV + sub * 32;
      if (kbase >= a.nk) break;                                   // wave-uniform
      // ---- S^T (32 keys x 32 queries) -------------------------------------------------------
      f32x16 sacc;
#pragma unroll
      for (int v = 0; v < 16; ++v) sacc[v] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint4 kf = *(const uint4*)(kb + (sub * 32 + lr) * KS + s * 32 + lh * 16);
        AMma<TC>::run(kf, qf[s], sacc);
      }
      // ---- online softmax: this lane = query lr, registers = keys (v&3) + 8 (v>>2) + 4 lh ----
      // VALU diet (the loop is VALU-bound, not MFMA-bound): keys are masked only in the ragged last
      // sub-tile, the maximum is taken on the raw scores (scale > 0), scale and shift are one FMA in
      // front of exp2, and the accumulators are rescaled only when some query's running maximum moved.
      if (kbase + 32 > a.nk) {                                    // wave-uniform
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int key = kbase + (v & 3) + 8 * (v >> 2) + 4 * lh;
          if (key >= a.nk) sacc[v] = -INFINITY;                   // padding rows may hold anything (NaN included)
        }
      }
      float mx = sacc[0];
#pragma unroll
      for (int v = 1; v < 16; ++v) mx = fmaxf(mx, sacc[v]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m, mx * a.scale_log2e);           // finite: every sub-tile entered has a valid key
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);      // exp2(-inf) = 0 on the first tile
      float rs = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        sacc[v] = __builtin_amdgcn_exp2f(fmaf(sacc[v], a.scale_log2e, -m_new));
        rs += sacc[v];
      }
      rs += __shfl_xor(rs, 32, 64);
      l = l * alpha + rs;
      m = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {       // wave-uniform: usually false after the first tiles
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int v = 0; v < 16; ++v) o[tt][v] *= alpha;
      }
      // ---- O^T += V^T P^T: P registers 8 ks .. 8 ks + 7 are the B fragment of k-step ks --------
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 pf;
        pf.x = pack2<TC>(sacc[8 * ks + 0], sacc[8 * ks + 1]);
        pf.y = pack2<TC>(sacc[8 * ks + 2], sacc[8 * ks + 3]);
        pf.z = pack2<TC>(sacc[8 * ks + 4], sacc[8 * ks + 5]);
        pf.w = pack2<TC>(sacc[8 * ks + 6], sacc[8 * ks + 7]);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          if constexpr (VROW) {
            // group (lane >> 4) = (d half dh, key half lh); lane 4 q + p of it addresses key row q, d columns 4 p .. 4 p + 3
            const unsigned char* vr = vb + (sub * 32 + 16 * ks + 4 * lh + ((lane & 15) >> 2)) * VRS +
                                      (tt * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
            const uint2 v0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)vr));
            const uint2 v1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(vr + 8 * VRS)));
            AMma<TC>::run(make_uint4(v0.x, v0.y, v1.x, v1.y), pf, o[tt]);
          } else {
            const unsigned char* vr = vb + (tt * 32 + lr) * VS + (sub * 32 + 16 * ks + 4 * lh) * 2;
            const uint2 v0 = *(const uint2*)vr;            // keys 16 ks + 4 lh + 0..3
            const uint2 v1 = *(const uint2*)(vr + 16);     // keys 16 ks + 8 + 4 lh + 0..3
            AMma<TC>::run(make_uint4(v0.x, v0.y, v1.x, v1.y), pf, o[tt]);
          }
        }
      }
    }
    if (t + 1 < ntile) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- O[q][dv] = O^T / l ----------------------------------------------------------------------
  const int qrow = q0 + lr;
  if (qrow < a.nq) {
    const float inv = 1.f / l;
    TC* orow = (TC*)a.out + (long)b * a.so_b + (long)qrow * a.ldo + (long)head * D;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = tt * 32 + 8 * g + 4 * lh;
        uint2 u;
        u.x = pack2<TC>(o[tt][4 * g + 0] * inv, o[tt][4 * g + 1] * inv);
        u.y = pack2<TC>(o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv);
        *(uint2*)(orow + dv) = u;
      }
  }
}

template <typename TC, int D, bool VROW>
static int launch_flash_t(const FlashArgs& a, int B, int heads, hipStream_t s) {
  constexpr int smem = 2 * (64 * (D * 2 + 16) + (VROW ? 64 * (D == 32 ? 64 : (D == 64 ? 192 : 320)) : D * (64 * 2 + 8)));
  auto kern = attn_flash_kernel<TC, D, VROW>;
  T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));
  dim3 grid((a.nq + 127) / 128, heads, B);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof_on()) {                     // bench.py roofline leg: HIP events on the launch stream
    T2P_HIP_CHECK(hipEventCreate(&e0));
    T2P_HIP_CHECK(hipEventCreate(&e1));
    T2P_HIP_CHECK(hipEventRecord(e0, s));
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
  if (e0) {
    T2P_HIP_CHECK(hipEventRecord(e1, s));
    prof_attention(e0, e1, 4.0 * (double)a.nq * a.nk * D * heads * B);      // q k^T and p v
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}


// =================================================================================================
// Single-head attention with a WIDE head (AttnBlockpp.forward, reference score_sde_pytorch/models/layers.py:160-176:
// one head, d = C = 256 / 512 / 1024, n = h w <= 1024 pixels): softmax(q k^T / sqrt C) v in ONE launch, the score
// strip never leaves the chip.
//
// A d = 512 flash kernel would need 32 queries x 512 channels of output accumulators per wavefront (256 registers), so
// the work is cut the other way.  One workgroup (8 wavefronts) owns a strip of 64 queries and ALL n <= 1024 keys:
//   pass 1  wavefront w computes S^T for its n / 8 keys x 64 queries over the whole head dimension (the strip of scores is
//           64 x 1024 fp32 = 256 KiB = the accumulators of 8 wavefronts x 128 registers); the 64 x d query strip sits in
//           LDS (shared by all wavefronts), the key rows stream from global memory straight into MFMA A fragments -- each
//           wavefront reads keys nobody else in the workgroup reads, so LDS staging would buy nothing -- with the head
//           dimension permuted so that a lane reads 32 contiguous bytes per 32-deep step (both operands use the same order);
//   softmax exact (not online): per-query maximum and sum exchanged between the wavefronts through LDS (two barriers),
//           P = exp2(...) unnormalised in the compute dtype written over the dead query strip, [query][key];
//   pass 2  wavefront w owns d / 8 output channels for all 64 queries: O^T = V^T P^T with V^T rows (the engine's transposed
//           value projection) streaming from global memory into A fragments and P^T fragments read from LDS;
//           O is scaled by 1 / sum and stored [query][channel].
// Orientation as in the flash kernel above: the query sits on the lane (S^T = K Q^T, O^T = V^T P^T), so the softmax
// statistics are per-lane scalars and one xor-32 shuffle.  Launch count of an AttnBlockpp: 7 -> 5, no fp32 score tensor and
// no probability tensor in HBM (2 x 134 MB + 67 MB per launch at the 32 x 32 level of cfg2).
struct StripArgs {
  const void* q; long ldq; long sq_b;       // [B][n][ldq]
  const void* k; long ldk; long sk_b;       // [B][n][ldk]
  const void* vt; long ldvt; long svt_b;    // [B][D][ldvt]  (V^T, ldvt >= n rounded up to 8)
  void* out; long ldo; long so_b;           // [B][n][ldo]
  int n;
  float scale_log2e;
  // optional epilogue (ep != 0): out = alpha (attention + bias[channel] + residual[query][channel]) in fp32 or the compute dtype,
  // col_stats [B n / 64][D][2] = column sum / sum of squares of those fp32 values over the strip's 64 queries
  int ep, out_f32, r_lowp;
  const float* bias;
  const void* res; long ldr;
  float alpha;
  float* col_stats;
};

// FM: K and V^T are FRAGMENT-MAJOR (GemmParams::c_frag of the two projections): a fragment load is 1 KiB contiguous -- eight whole
// cache lines -- instead of 32 bytes of each of 32 lines (156 -> 105 us per launch at cfg2's 32 x 32 level; n % 32 == 0)
template <typename TC, int D, int NKT, bool FM = false>      // NKT: 32-key tiles per wavefront (n <= 256 NKT)
__global__ __launch_bounds__(512) void attn_strip_kernel(const StripArgs a) {
  constexpr int QRS = D * 2 + 16;            // query strip row stride (bytes): consecutive rows 4 banks apart
  constexpr int KW = NKT * 32;               // keys per wavefront
  constexpr int PRS = KW * 8 * 2 + 16;       // probability strip row stride (bytes)
  constexpr int CT = D / 256;                // 32-channel tiles of O^T per wavefront
  constexpr int REGION = (64 * QRS > 64 * PRS) ? 64 * QRS : 64 * PRS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* st_max = (float*)(smem + REGION);   // [8][64]
  float* st_sum = st_max + 8 * 64;           // [8][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // logical (strip, sample) with the strip fastest: the strips of one sample (which all stream its K and V^T) share an XCD's L2
  const int lid = xcd_logical_id((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
  const int strip = lid % (int)gridDim.x, b = lid / (int)gridDim.x;
  const int q0 = strip * 64, n = a.n;
  const TC* Q = (const TC*)a.q + (long)b * a.sq_b;
  const TC* K = (const TC*)a.k + (long)b * a.sk_b;
  const TC* VT = (const TC*)a.vt + (long)b * a.svt_b;

  // ---- query strip -> LDS (rows beyond n: zeros) ----------------------------------------------------------------
  for (int i = tid; i < 64 * (D / 8); i += 512) {
    const int row = i / (D / 8), ch = i - row * (D / 8);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q0 + row < n) v = *(const uint4*)(Q + (long)(q0 + row) * a.ldq + ch * 8);
    *(uint4*)(smem + row * QRS + ch * 16) = v;
  }
  __syncthreads();

  // ---- pass 1: S^T (this wavefront's keys x 64 queries) ---------------------------------------------------------------
  const int key_w = wave * KW;                                   // first key of this wavefront
  const bool w_active = key_w < n;                               // wave-uniform
  f32x16 sacc[NKT][2];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int v = 0; v < 16; ++v) sacc[kt][qt][v] = 0.f;
  if (w_active) {
    // lane (key lr, half lh) of key tile kt reads K[key][32 S + 16 lh .. + 15]: the fragments of the two 16-deep MFMA steps of
    // super-step S (keys beyond n: clamped to a valid row, their scores are masked below)
    const TC* krow[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      int key = key_w + kt * 32 + lr;
      key = key < n ? key : n - 1;
      krow[kt] = K + (long)key * a.ldk + 16 * lh;
    }
#if defined(STRIP_FM_TIMING)   // timing only (wrong results): the first measurement of what fragment-major operands would give
#define KLD(kt, S, j) (*(const uint4*)(K + ((((long)(key_w / 32 + (kt)) * (D / 32) + (S)) * 2 + (j)) * 64 + lane) * 8))
#else
#define KLD(kt, S, j) (FM ? *(const uint4*)(K + ((((long)(key_w / 32 + (kt)) * (D / 32) + (S)) * 2 + (j)) * 64 + lane) * 8) \
                          : *(const uint4*)(krow[kt] + 32 * (S) + 8 * (j)))
#endif
    const unsigned char* qb0 = smem + lr * QRS + 32 * lh;        // query tile 0; tile 1 is 32 rows further
    uint4 ka0[NKT][2], ka1[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      ka0[kt][0] = KLD(kt, 0, 0);
      ka0[kt][1] = KLD(kt, 0, 1);
    }
    auto sstep = [&](const uint4 (&ka)[NKT][2], int S) {
      uint4 qf[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int j = 0; j < 2; ++j) qf[qt][j] = *(const uint4*)(qb0 + qt * 32 * QRS + S * 64 + j * 16);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) AMma<TC>::run(ka[kt][j], qf[qt][j], sacc[kt][qt]);
    };
    for (int S = 0; S < D / 32; S += 2) {                          // D / 32 is even
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        ka1[kt][0] = KLD(kt, S + 1, 0);
        ka1[kt][1] = KLD(kt, S + 1, 1);
      }
      sstep(ka0, S);
      if (S + 2 < D / 32) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          ka0[kt][0] = KLD(kt, S + 2, 0);
          ka0[kt][1] = KLD(kt, S + 2, 1);
        }
      }
      sstep(ka1, S + 1);
    }
  }

  // ---- exact softmax over all keys: this lane = query (32 qt + lr), registers = keys (v & 3) + 8 (v >> 2) + 4 lh of a tile -------------
  float mloc[2] = {-INFINITY, -INFINITY};
  if (w_active) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int kbase = key_w + kt * 32;
      if (kbase + 32 > n) {                                      // wave-uniform: ragged or empty tile
#pragma unroll
        for (int v = 0; v < 16; ++v)
          if (kbase + (v & 3) + 8 * (v >> 2) + 4 * lh >= n) { sacc[kt][0][v] = -INFINITY; sacc[kt][1][v] = -INFINITY; }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int v = 0; v < 16; ++v) mloc[qt] = fmaxf(mloc[qt], sacc[kt][qt][v]);
    }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    mloc[qt] = fmaxf(mloc[qt], __shfl_xor(mloc[qt], 32, 64));
    if (lh == 0) st_max[wave * 64 + qt * 32 + lr] = mloc[qt];
  }
  __syncthreads();                    // every wavefront is also done with the query strip: the region becomes the P strip
  float mq[2], lsum[2] = {0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float m = st_max[qt * 32 + lr];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, st_max[w * 64 + qt * 32 + lr]);
    mq[qt] = m * a.scale_log2e;        // finite: key 0 exists (scale > 0: the maximum commutes with the scaling)
  }
  if (w_active) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        unsigned char* prow = smem + (qt * 32 + lr) * PRS + (key_w + kt * 32 + 4 * lh) * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float e[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            e[i] = __builtin_amdgcn_exp2f(fmaf(sacc[kt][qt][4 * g + i], a.scale_log2e, -mq[qt]));   // exp2(-inf) = 0 for masked keys
            lsum[qt] += e[i];
          }
          *(uint2*)(prow + 16 * g) = make_uint2(pack2<TC>(e[0], e[1]), pack2<TC>(e[2], e[3]));     // keys 8 g + 4 lh + 0..3 of the tile
        }
      }
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    lsum[qt] += __shfl_xor(lsum[qt], 32, 64);
    if (lh == 0) st_sum[wave * 64 + qt * 32 + lr] = lsum[qt];
  }
  __syncthreads();
  float inv[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = st_sum[qt * 32 + lr];
#pragma unroll
    for (int w = 1; w < 8; ++w) l += st_sum[w * 64 + qt * 32 + lr];
    inv[qt] = 1.f / l;
  }

  // ---- pass 2: O^T (this wavefront's d / 8 channels x 64 queries) = V^T P^T over all keys -------------------------------------------
  f32x16 oacc[CT][2];
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int v = 0; v < 16; ++v) oacc[t][qt][v] = 0.f;
  const int ch_w = wave * (D / 8);
  const TC* vrow[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) vrow[t] = VT + (long)(ch_w + t * 32 + lr) * a.ldvt + 16 * lh;
  const int nss = (n + 31) >> 5;                                 // 32-key super-steps
  // V^T fragment of super-step S, step j: keys 32 S + 16 lh + 8 j .. + 7 (whole 8-key chunks are in or out: n % 8 == 0)
  auto vload = [&](int t, int S, int j) -> uint4 {
    const int key = 32 * S + 16 * lh + 8 * j;
#ifdef STRIP_FM_TIMING
    return key < n ? *(const uint4*)(VT + ((((long)(ch_w / 32 + t) * ((n + 31) >> 5) + S) * 2 + j) * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
#else
    if constexpr (FM) return *(const uint4*)(VT + ((((long)(ch_w / 32 + t) * (n >> 5) + S) * 2 + j) * 64 + lane) * 8);     // n % 32 == 0
    return key < n ? *(const uint4*)(vrow[t] + 32 * S + 8 * j) : make_uint4(0, 0, 0, 0);
#endif
  };
  uint4 va0[CT][2], va1[CT][2];                                  // two named buffers (a runtime-indexed one would live in scratch)
#pragma unroll
  for (int t = 0; t < CT; ++t) { va0[t][0] = vload(t, 0, 0); va0[t][1] = vload(t, 0, 1); }
  const unsigned char* pb0 = smem + lr * PRS + 32 * lh;
  auto pstep = [&](const uint4 (&va)[CT][2], int S) {
    uint4 pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int j = 0; j < 2; ++j) pf[qt][j] = *(const uint4*)(pb0 + qt * 32 * PRS + S * 64 + j * 16);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) AMma<TC>::run(va[t][j], pf[qt][j], oacc[t][qt]);
  };
  for (int S = 0; S < nss; S += 2) {
    if (S + 1 < nss) {
#pragma unroll
      for (int t = 0; t < CT; ++t) { va1[t][0] = vload(t, S + 1, 0); va1[t][1] = vload(t, S + 1, 1); }
    }
    pstep(va0, S);
    if (S + 1 < nss) {
      if (S + 2 < nss) {
#pragma unroll
        for (int t = 0; t < CT; ++t) { va0[t][0] = vload(t, S + 2, 0); va0[t][1] = vload(t, S + 2, 1); }
      }
      pstep(va1, S + 1);
    }
  }

  // ---- O[q][channel] = O^T / l (+ epilogue) -----------------------------------------------------------------------------------------
  if (!a.ep) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qrow = q0 + qt * 32 + lr;
      if (qrow < n) {
        TC* orow = (TC*)a.out + (long)b * a.so_b + (long)qrow * a.ldo + ch_w;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 u;
            u.x = pack2<TC>(oacc[t][qt][4 * g + 0] * inv[qt], oacc[t][qt][4 * g + 1] * inv[qt]);
            u.y = pack2<TC>(oacc[t][qt][4 * g + 2] * inv[qt], oacc[t][qt][4 * g + 3] * inv[qt]);
            *(uint2*)(orow + t * 32 + 8 * g + 4 * lh) = u;
          }
      }
    }
    return;
  }
  // the block's closing arithmetic here instead of in an output-projection GEMM (the projection itself is folded into V):
  // alpha (O + bias + residual), and the column statistics the next GroupNorm reads, over this strip's 64 queries
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = ch_w + t * 32 + 8 * g + 4 * lh;
      const float4 bs = a.bias ? *(const float4*)(a.bias + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
      float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int qrow = q0 + qt * 32 + lr;
        if (qrow < n) {
          const long ro = (long)b * n * a.ldr + (long)qrow * a.ldr + ch;
          float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
          if (a.res) {
            if (a.r_lowp) {
              union { uint2 u; TC e[4]; } x;
              x.u = *(const uint2*)((const TC*)a.res + ro);
              r0 = to_f32(x.e[0]); r1 = to_f32(x.e[1]); r2 = to_f32(x.e[2]); r3 = to_f32(x.e[3]);
            } else {
              const float4 x = *(const float4*)((const float*)a.res + ro);
              r0 = x.x; r1 = x.y; r2 = x.z; r3 = x.w;
            }
          }
          const float v0 = (oacc[t][qt][4 * g + 0] * inv[qt] + bs.x + r0) * a.alpha, v1 = (oacc[t][qt][4 * g + 1] * inv[qt] + bs.y + r1) * a.alpha;
          const float v2 = (oacc[t][qt][4 * g + 2] * inv[qt] + bs.z + r2) * a.alpha, v3 = (oacc[t][qt][4 * g + 3] * inv[qt] + bs.w + r3) * a.alpha;
          cs[0] += v0; cs[1] += v1; cs[2] += v2; cs[3] += v3;
          cq[0] += v0 * v0; cq[1] += v1 * v1; cq[2] += v2 * v2; cq[3] += v3 * v3;
          const long oo = (long)b * a.so_b + (long)qrow * a.ldo + ch;
          if (a.out_f32) *(float4*)((float*)a.out + oo) = make_float4(v0, v1, v2, v3);
          else *(uint2*)((TC*)a.out + oo) = make_uint2(pack2<TC>(v0, v1), pack2<TC>(v2, v3));
        }
      }
      if (a.col_stats) {                              // (uniform) sum over the 32 query lanes of this half, fixed butterfly order
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int sh = 1; sh < 32; sh <<= 1) { cs[i] += __shfl_xor(cs[i], sh, 64); cq[i] += __shfl_xor(cq[i], sh, 64); }
        if (lr == 0) {
          float* dst = a.col_stats + ((long)(b * (n >> 6) + strip) * D + ch) * 2;
          *(float4*)dst = make_float4(cs[0], cq[0], cs[1], cq[1]);
          *(float4*)(dst + 4) = make_float4(cs[2], cq[2], cs[3], cq[3]);
        }
      }
    }
}

bool g_attn_strip = true;       // plan switch 29
bool attention_strip_eligible(int dtype, int heads, int nq, int nk, int d, long ldq, long ldk, long ldvt, long ldo) {
  if (!g_attn_strip || dtype == DT_F32 || heads != 1 || nq != nk) return false;
  if (d != 256 && d != 512 && d != 1024) return false;
  if (nk < 8 || nk > 1024 || nk % 8 != 0) return false;
  if (d == 1024 && nk > 512) return false;             // 64 x 1024 query strip + 64 x 1024 probability strip: beyond the LDS budget
  return ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldvt >= nk && ldo % 4 == 0;
}

template <typename TC, int D, int NKT, bool FM = false>
static int launch_strip_t(const StripArgs& a, int B, hipStream_t s) {
  constexpr int QB = 64 * (D * 2 + 16), PB = 64 * (NKT * 32 * 8 * 2 + 16);
  constexpr int smem = (QB > PB ? QB : PB) + 2 * 8 * 64 * 4;
  static_assert(smem <= 160 * 1024, "attn_strip: LDS budget");
  auto kern = attn_strip_kernel<TC, D, NKT, FM>;
  T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof_on()) {
    T2P_HIP_CHECK(hipEventCreate(&e0));
    T2P_HIP_CHECK(hipEventCreate(&e1));
    T2P_HIP_CHECK(hipEventRecord(e0, s));
  }
  hipLaunchKernelGGL(kern, dim3((a.n + 63) / 64, B), dim3(512), smem, s, a);
  if (e0) {
    T2P_HIP_CHECK(hipEventRecord(e1, s));
    prof_attention(e0, e1, 4.0 * (double)a.n * a.n * D * B);
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

bool attention_strip_frag_major_ok(int dtype, int n, int d) {
  if ((dtype != DT_F16 && dtype != DT_BF16) || n % 32 != 0) return false;
  return (d == 512 && n > 512 && n <= 1024) || (d == 256 && n <= 256);
}
int launch_attention_strip(int dtype, const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt, void* out, long ldo,
                           int B, int n, int d, float scale, hipStream_t s, const StripEpilogue* ep) {
  T2P_REQUIRE(attention_strip_eligible(dtype, 1, n, n, d, ldq, ldk, ldvt, ldo), "wide-head attention: unsupported shape");
  T2P_REQUIRE(q && k && vt && out && B > 0 && scale > 0.f, "wide-head attention arguments");
  T2P_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)out % 8) == 0,
              "wide-head attention: operands must be 16-byte aligned");
  StripArgs a;
  a.q = q; a.ldq = ldq; a.sq_b = (long)n * ldq;
  a.k = k; a.ldk = ldk; a.sk_b = (long)n * ldk;
  a.vt = vt; a.ldvt = ldvt; a.svt_b = (long)d * ldvt;
  a.out = out; a.ldo = ldo; a.so_b = (long)n * ldo;
  a.n = n;
  a.scale_log2e = scale * 1.44269504088896340736f;
  a.ep = ep ? 1 : 0; a.out_f32 = 0; a.r_lowp = 0; a.bias = nullptr; a.res = nullptr; a.ldr = 0; a.alpha = 1.f; a.col_stats = nullptr;
  if (ep) {
    T2P_REQUIRE(!ep->col_stats || n % 64 == 0, "wide-head attention: column statistics need whole 64-query strips");
    T2P_REQUIRE(!ep->residual || (ep->ldr % 4 == 0 && ((uintptr_t)ep->residual % 16) == 0), "wide-head attention: residual alignment");
    T2P_REQUIRE(((uintptr_t)out % 16) == 0 && ldo % 4 == 0, "wide-head attention: output alignment");
    a.out_f32 = ep->out_f32; a.r_lowp = ep->r_lowp; a.bias = ep->bias; a.res = ep->residual; a.ldr = ep->ldr; a.alpha = ep->alpha;
    a.col_stats = ep->col_stats;
  }
  const int nkt = n <= 256 ? 1 : (n <= 512 ? 2 : 4);
  if (ep && ep->frag_major) {     // K and V^T fragment-major: [B][n d] each (the 32 x 32 level of the C = 512 configurations)
    T2P_REQUIRE(attention_strip_frag_major_ok(dtype, n, d), "wide-head attention: fragment-major operands need d = 512 with 512 < n <= 1024 or d = 256 with n <= 256, n % 32 == 0");
    a.sk_b = (long)n * d; a.svt_b = (long)n * d;
    if (d == 256) {
      if (dtype == DT_BF16) return launch_strip_t<bf16_t, 256, 1, true>(a, B, s);
      return launch_strip_t<f16_t, 256, 1, true>(a, B, s);
    }
    if (dtype == DT_BF16) return launch_strip_t<bf16_t, 512, 4, true>(a, B, s);
    return launch_strip_t<f16_t, 512, 4, true>(a, B, s);
  }
#define T2P_STRIP(TC, DD)                                                        \
  do {                                                                           \
    if (nkt == 1) return launch_strip_t<TC, DD, 1>(a, B, s);                     \
    if (nkt == 2) return launch_strip_t<TC, DD, 2>(a, B, s);                     \
    if constexpr (DD != 1024) return launch_strip_t<TC, DD, 4>(a, B, s);         \
  } while (0)
  if (dtype == DT_BF16) {
    if (d == 256) T2P_STRIP(bf16_t, 256);
    if (d == 512) T2P_STRIP(bf16_t, 512);
    if (d == 1024) T2P_STRIP(bf16_t, 1024);
  } else {
    if (d == 256) T2P_STRIP(f16_t, 256);
    if (d == 512) T2P_STRIP(f16_t, 512);
    if (d == 1024) T2P_STRIP(f16_t, 1024);
  }
#undef T2P_STRIP
  set_last_error("wide-head attention: no instantiation");
  return T2P_ERR_INVALID;
}

bool attention_flash_eligible(int dtype, int d, long ldq, long ldk, long ldvt, long ldo) {
  if (dtype == DT_F32) return false;
  if (d != 32 && d != 64 && d != 128) return false;
  return ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0;
}

// v_rowmajor: `vt` is V itself, [B][nk][ldvt] with head h at column h * d (ldvt its row stride), not the transposed projection
int launch_attention_flash(int dtype, const void* q, long ldq, const void* k, long ldk, const void* vt, long ldvt,
                           void* out, int B, int heads, int nq, int nk, int d, float scale, hipStream_t s, bool v_rowmajor) {
  T2P_REQUIRE(attention_flash_eligible(dtype, d, ldq, ldk, ldvt, (long)heads * d), "flash attention: unsupported shape");
  T2P_REQUIRE(q && k && vt && out && B > 0 && heads > 0 && nq > 0 && nk > 0, "flash attention arguments");
  T2P_REQUIRE(scale > 0.f, "flash attention: the running maximum is taken on unscaled scores (scale must be positive)");
  T2P_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)out % 8) == 0,
              "flash attention: operands must be 16-byte aligned");
  FlashArgs a;
  a.q = q; a.ldq = ldq; a.sq_b = (long)nq * ldq;
  a.k = k; a.ldk = ldk; a.sk_b = (long)nk * ldk;
  a.vt = vt; a.ldvt = ldvt; a.svt_b = v_rowmajor ? (long)nk * ldvt : (long)heads * d * ldvt;
  a.out = out; a.ldo = (long)heads * d; a.so_b = (long)nq * heads * d;
  a.nq = nq; a.nk = nk;
  a.scale_log2e = scale * 1.44269504088896340736f;
#define T2P_FLASH(TC, DD) return v_rowmajor ? launch_flash_t<TC, DD, true>(a, B, heads, s) : launch_flash_t<TC, DD, false>(a, B, heads, s)
  if (dtype == DT_BF16) {
    if (d == 32) T2P_FLASH(bf16_t, 32);
    if (d == 64) T2P_FLASH(bf16_t, 64);
    T2P_FLASH(bf16_t, 128);
  } else {
    if (d == 32) T2P_FLASH(f16_t, 32);
    if (d == 64) T2P_FLASH(f16_t, 64);
    T2P_FLASH(f16_t, 128);
  }
#undef T2P_FLASH
}

}  // namespace t2p
