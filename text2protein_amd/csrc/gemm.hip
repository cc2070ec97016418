// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (see GemmParams in t2p_common.h).
//
// Replaces, on the sampling path of the reference: nn.Conv2d 3x3 (layers.py:89-95), 1x1
// convolutions (layers.py:82-87, attention.py:233-248), NIN (layers.py:128-137), nn.Linear
// (attention.py:161-168, 40, 60) and the four einsum contractions of the attention blocks
// (layers.py:166-171, attention.py:181,191).
//
// Structure: one workgroup = 4 wavefronts (64 lanes each) computes a BM x BN tile; wave (wm, wn)
// owns a (BM/2) x (BN/2) sub-tile made of 32x32 MFMA tiles.  Operand tiles are staged
// global -> registers -> LDS (16-byte vectors, 144-byte padded rows so that ds_read_b128
// fragment reads and ds_write_b128 staging writes are bank-conflict free), double-buffered so
// the global loads of K-tile t+1 are in flight while the MFMAs of tile t run.
//   * fp32 compute : v_mfma_f32_32x32x2_f32  (exact f32, 64 FLOP/clk/SIMD)
//   * bf16 / fp16  : v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate
// LDS rows hold BK = 128 bytes of K (32 fp32 / 64 16-bit elements); a lane's fragment for
// k-group s is the 16 bytes at [row][32 s + 16 (lane >> 5)], for both element widths.
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "t2p_common.h"

namespace t2p {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

[[maybe_unused]] static constexpr int ROWB = 144;  // LDS row stride in bytes: 128 data + 16 pad

template <typename TC> struct VecInfo { static constexpr int VEC = 16 / (int)sizeof(TC); };

// pack VEC floats into 16 bytes of TC
template <typename TC> __device__ inline uint4 pack16(const float* f);
template <> __device__ inline uint4 pack16<float>(const float* f) {
  uint4 r;
  r.x = __builtin_bit_cast(uint32_t, f[0]); r.y = __builtin_bit_cast(uint32_t, f[1]);
  r.z = __builtin_bit_cast(uint32_t, f[2]); r.w = __builtin_bit_cast(uint32_t, f[3]);
  return r;
}
template <> __device__ inline uint4 pack16<bf16_t>(const float* f) {
  uint4 r;
  r.x = (uint32_t)f32_to_bf16_bits(f[0]) | ((uint32_t)f32_to_bf16_bits(f[1]) << 16);
  r.y = (uint32_t)f32_to_bf16_bits(f[2]) | ((uint32_t)f32_to_bf16_bits(f[3]) << 16);
  r.z = (uint32_t)f32_to_bf16_bits(f[4]) | ((uint32_t)f32_to_bf16_bits(f[5]) << 16);
  r.w = (uint32_t)f32_to_bf16_bits(f[6]) | ((uint32_t)f32_to_bf16_bits(f[7]) << 16);
  return r;
}
template <> __device__ inline uint4 pack16<f16_t>(const float* f) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  uint4 r;
  h2 a = {(_Float16)f[0], (_Float16)f[1]}, b = {(_Float16)f[2], (_Float16)f[3]};
  h2 c = {(_Float16)f[4], (_Float16)f[5]}, d = {(_Float16)f[6], (_Float16)f[7]};
  r.x = __builtin_bit_cast(uint32_t, a); r.y = __builtin_bit_cast(uint32_t, b);
  r.z = __builtin_bit_cast(uint32_t, c); r.w = __builtin_bit_cast(uint32_t, d);
  return r;
}

// zero the elements >= nvalid of a packed vector
template <typename TC> __device__ inline uint4 mask_tail(uint4 v, int nvalid) {
  constexpr int VEC = VecInfo<TC>::VEC;
  union { uint4 u; TC e[VEC]; } x;
  x.u = v;
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    if (i >= nvalid) x.e[i] = from_f32<TC>(0.f);
  return x.u;
}

// load one staging vector (VEC K-elements) from a source row; `off` in elements
template <typename TC, bool SRC_F32>
__device__ inline uint4 load_vec(const void* base, long off) {
  constexpr int VEC = VecInfo<TC>::VEC;
  if constexpr (SRC_F32) {
    const float* p = (const float*)base + off;
    float f[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      float4 t = *(const float4*)(p + i);
      f[i] = t.x; f[i + 1] = t.y; f[i + 2] = t.z; f[i + 3] = t.w;
    }
    return pack16<TC>(f);
  } else {
    return *(const uint4*)((const TC*)base + off);
  }
}

template <typename TC> struct Mma;
template <> struct Mma<float> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

typedef float f32x4_res_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_res_t __attribute__((ext_vector_type(2)));
// residual operand: fp32, or the compute dtype when GemmParams::r_lowp (element index, not bytes)
template <typename TC> __device__ inline float res_load1(const float* R, long idx, int lowp) {
  if constexpr (sizeof(TC) == 2) { if (lowp) return to_f32(((const TC*)R)[idx]); }
  return R[idx];
}
template <typename TC> __device__ inline f32x4_res_t unpack4(u32x2_res_t u) {     // four 16-bit values -> fp32
  TC e[4];
  e[0] = __builtin_bit_cast(TC, (uint16_t)u.x); e[1] = __builtin_bit_cast(TC, (uint16_t)(u.x >> 16));
  e[2] = __builtin_bit_cast(TC, (uint16_t)u.y); e[3] = __builtin_bit_cast(TC, (uint16_t)(u.y >> 16));
  return (f32x4_res_t){to_f32(e[0]), to_f32(e[1]), to_f32(e[2]), to_f32(e[3])};
}
template <typename TC> __device__ inline float4 res_load4(const float* R, long idx, int lowp) {
  if constexpr (sizeof(TC) == 2) {
    if (lowp) {
      const f32x4_res_t v = unpack4<TC>(*(const u32x2_res_t*)((const TC*)R + idx));
      return make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  return *(const float4*)(R + idx);
}

// ---- compilation units -------------------------------------------------------------------------------
// This file is compiled several times (text2protein_amd/build.py) so that the template instantiations of the
// LDS-DMA kernel build in parallel: -DT2P_GEMM_PART=0 holds the host logic, the register-staged kernel, the
// split-K second pass and the thin-output convolution; parts 1..6 hold launch_dma_mode<dtype, MODE> for one
// (dtype, MODE) each.  Without the macro everything is one unit.
#ifndef T2P_GEMM_PART
#define T2P_GEMM_PART -1
#endif
#define T2P_PART_HOST (T2P_GEMM_PART <= 0)
#define T2P_PART_DMA (T2P_GEMM_PART != 0)

struct ProfRec { hipEvent_t a, b; double flops; int kind; const char* name; double bytes; int M = 0, N = 0, K = 0, taps = 0, z = 0; };
inline void prof_shape(ProfRec& r, const GemmParams& p) {
  r.M = p.M; r.N = p.N; r.K = p.C0 + p.C1; r.taps = p.a_up ? -p.taps : p.taps; r.z = p.nz0 * p.nz1;
  if (p.CX0 + p.CX1) r.K = r.K * 1000000 + p.CX0 + p.CX1;      // shortcut segment: printed as K+KX
}
struct DmaPlan { int geom, nsplit; };
extern bool g_prof_on;
extern std::vector<ProfRec> g_prof;
extern int g_dma_ring, g_dbg;
extern bool g_up4, g_deep_ring, g_dxs;
int num_cus();
DmaPlan dma_plan(const GemmParams& p);
int launch_splitk_reduce(const GemmParams& p, int nsplit, hipStream_t stream);
template <typename TC, int MODE> int launch_dma_mode(const GemmParams& p, hipStream_t stream);

// true when reg_epilogue covers the launch (else the LDS-staged dma_epilogue runs; both give the same values)
__host__ __device__ __forceinline__ bool reg_epilogue_ok(const GemmParams& p, const int nsplit, const int dbg) {
  if (dbg & 4096) return false;
  const long c_cols = p.geglu ? p.N / 2 : p.N;
  const long c_bytes = nsplit > 1 ? (long)p.M * p.N * 4 : ((long)(p.M - 1) * p.ldc + c_cols) * (p.c_f32 ? 4 : 2);
  const int HW = p.H * p.W;
  const long r_rows = p.r_up ? (long)(p.M / HW) * (p.H >> 1) * (p.W >> 1) : (long)p.M;
  const long r_bytes = p.R ? ((r_rows - 1) * p.ldr + p.N) * (p.r_lowp ? 2 : 4) : 0;
  if ((p.N & 7) || c_bytes >= (1L << 31) || r_bytes >= (1L << 31)) return false;
  if (nsplit > 1) return true;                                   // raw fp32 partial tiles [M][N]
  if (p.geglu ? (p.ldc & 3) : (p.ldc & 7)) return false;         // 16-byte aligned row segments (8 bytes for GEGLU's half-width rows)
  if (p.R && (p.ldr & 7)) return false;
  if (p.bias_bn && ((p.ld_bn & 3) || p.rows_per_batch % 16)) return false;
  if (p.r_up && (p.W % 16 || p.rows_per_batch != HW)) return false;
  return true;
}


#if T2P_PART_HOST
template <typename TC, bool AF32, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int VEC = VecInfo<TC>::VEC;
  constexpr int BK = 8 * VEC;
  constexpr int AR = BM / 32, BR = BN / 32;  // staging rows per thread
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                          // [2][BM][ROWB]
  unsigned char* Bs = smem + 2 * BM * ROWB;          // [2][BN][ROWB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int z0 = blockIdx.z / p.nz1, z1 = blockIdx.z % p.nz1;

  const int Ctot = p.C0 + p.C1;
  const int nch = (Ctot + BK - 1) / BK;
  const int nk = nch * p.taps;
  const bool spatial = (p.taps == 9) || p.a_up;
  const int HW = p.H * p.W;
  const int Hs = p.a_up ? (p.H >> 1) : p.H, Ws = p.a_up ? (p.W >> 1) : p.W;

  const long aoff = (long)z0 * p.sA_z0 + (long)z1 * p.sA_z1;
  const char* A0 = (const char*)p.A0 + aoff * (AF32 ? 4 : (long)sizeof(TC));
  const char* A1 = (const char*)p.A1;
  const TC* Bw = (const TC*)p.Bw + (long)z0 * p.sB_z0 + (long)z1 * p.sB_z1;

  // staging assignment: vector column cv (16 bytes of K), rows srow + 32 i
  const int cv = tid & 7, srow = tid >> 3;
  int a_b[AR], a_y[AR], a_x[AR];
  bool a_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + srow + 32 * i;
    a_ok[i] = m < p.M;
    if (spatial) {
      int b = m / HW, rem = m - b * HW;
      a_b[i] = b; a_y[i] = rem / p.W; a_x[i] = rem - a_y[i] * p.W;
    } else {
      a_b[i] = m; a_y[i] = 0; a_x[i] = 0;
    }
  }

  uint4 ra[AR], rb[BR];

  auto gload = [&](int kt) {
    const int tap = kt / nch;
    const int c = (kt - tap * nch) * BK + cv * VEC;
    const bool cok = c < Ctot;
    const int nvalid = Ctot - c;   // < VEC only for ragged K (taps == 1)
    int dy = 0, dx = 0;
    if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
    const bool second = c >= p.C0;
    const void* src = second ? (const void*)A1 : (const void*)A0;
    const long ld = second ? p.lda1 : p.lda0;
    const int cc = second ? c - p.C0 : c;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      bool ok = a_ok[i] && cok;
      long row = a_b[i];
      if (spatial) {
        int sy = a_y[i] + dy, sx = a_x[i] + dx;
        ok = ok && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
        if (p.a_up) { sy >>= 1; sx >>= 1; }
        row = ((long)a_b[i] * Hs + sy) * Ws + sx;
      }
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        v = load_vec<TC, AF32>(src, row * ld + cc);
        if (nvalid < VEC) v = mask_tail<TC>(v, nvalid);
      }
      ra[i] = v;
    }
    const long kb = (long)tap * Ctot + c;
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      int n = n0 + srow + 32 * j;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < p.N && cok) {
        v = *(const uint4*)(Bw + (long)n * p.ldb + kb);
        if (nvalid < VEC) v = mask_tail<TC>(v, nvalid);
      }
      rb[j] = v;
    }
  };
  auto sstore = [&](int buf) {
    unsigned char* a = As + buf * BM * ROWB;
    unsigned char* b = Bs + buf * BN * ROWB;
#pragma unroll
    for (int i = 0; i < AR; ++i) *(uint4*)(a + (srow + 32 * i) * ROWB + cv * 16) = ra[i];
#pragma unroll
    for (int j = 0; j < BR; ++j) *(uint4*)(b + (srow + 32 * j) * ROWB + cv * 16) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const unsigned char* a = As + buf * BM * ROWB + (wm * (BM / 2) + lr) * ROWB + lh * 16;
    const unsigned char* b = Bs + buf * BN * ROWB + (wn * (BN / 2) + lr) * ROWB + lh * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const uint4*)(a + i * 32 * ROWB + s * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *(const uint4*)(b + j * 32 * ROWB + s * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<TC>::run(af[i], bf[j], acc[i][j]);
    }
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
  const long coff = (long)z0 * p.sC_z0 + (long)z1 * p.sC_z1;
  const float* R = p.R ? (p.r_lowp ? (const float*)((const uint16_t*)p.R + (long)z0 * p.sR_z0 + (long)z1 * p.sR_z1)
                                   : p.R + (long)z0 * p.sR_z0 + (long)z1 * p.sR_z1) : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + wm * (BM / 2) + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (row >= p.M) continue;
      const int bidx = row / p.rows_per_batch;
      long rrow = row;
      if (p.r_up) {
        int rem = row - bidx * HW;
        int y = rem / p.W, x = rem - y * p.W;
        rrow = ((long)bidx * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
      }
      const float bm = p.bias_m ? p.bias_m[row] : 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + lr;
        if (col >= p.N) continue;
        float val = acc[i][j][v] + bm;
        if (p.bias_n) val += p.bias_n[col];
        if (p.bias_bn) val += p.bias_bn[(long)bidx * p.ld_bn + col];
        if (R) val += res_load1<TC>(R, rrow * p.ldr + col, p.r_lowp);
        val *= p.alpha;
        if (p.c_nchw) {
          const int pix = row - bidx * p.rows_per_batch;
          ((float*)p.C)[((long)bidx * p.N + col) * p.rows_per_batch + pix] = val * p.row_scale[bidx];
        } else if (p.c_f32) {
          ((float*)p.C)[coff + (long)row * p.ldc + col] = val;
        } else {
          ((TC*)p.C)[coff + (long)row * p.ldc + col] = from_f32<TC>(val);
        }
      }
    }
  }
}

#endif  // T2P_PART_HOST

// =================================================================================================
// v2: LDS-DMA pipeline for 16-bit operands (the dominant kernel: 3x3 convolutions and large GEMMs).
//
// v1 above stages operands through registers; its ds_write_b128 traffic (~79 B/clk/CU) plus the
// fragment reads make it LDS-bound at about 15 % of the 16-bit MFMA peak.  Here both operand
// tiles go HBM/L2 -> LDS directly with `buffer_load_dwordx4 ... lds` (no VGPR round trip, no
// ds_write), which also gives the convolution's zero padding and all ragged edges for free: a
// lane whose source is outside the map / matrix gets an out-of-range buffer offset and the
// hardware writes zeros.
//   * workgroup = 8 wavefronts (4 x 2), tile BM x BN = 256 x 128, BK = 64 (128-byte LDS rows)
//   * LDS ring of 3 stages (3 x 48 KiB); the loads of K-tile t+2 are issued while tile t is being
//     multiplied; a counted `s_waitcnt vmcnt(6)` + one raw s_barrier per K-tile
//   * LDS image is lane-linear per DMA instruction (8 rows x 128 B), so the bank-conflict swizzle
//     is applied to the SOURCE chunk: position p of row r holds chunk p ^ ((r >> 1) & 7), and the
//     ds_read_b128 fragment reads apply the same XOR (conflict-free for all 16-lane groups)
//   * tile ids are remapped so that consecutive tiles (same A rows, neighbouring pixels) run on
//     the same XCD and share its L2
// Requirements (else v1 runs): 16-bit compute dtype, A in the compute dtype, C0 % 64 == 0 and
// C1 % 64 == 0, M >= 256, N >= 64, every operand smaller than 2 GiB.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define T2P_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
[[maybe_unused]] static constexpr unsigned DMA_OOB = 0x80000000u;   // >= any num_records we accept -> zeros

// erf-GELU for the fused GEGLU epilogue (16-bit outputs only): Abramowitz-Stegun 7.1.26, |error| of
// erf <= 1.5e-7 -- far below the fp16 / bf16 rounding of the result -- in a dozen VALU operations
// instead of the branchy libm erff.
__device__ inline float gelu_erf_fast(float g) {
  const float x = g * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  const float erf_abs = fmaf(-q * t, e, 1.f);
  return 0.5f * g * (1.f + copysignf(erf_abs, x));
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// two fp32 values rounded to the compute dtype, packed low / high
template <typename TC> __device__ inline uint32_t pack2(float a, float b) {
  const TC x = from_f32<TC>(a), y = from_f32<TC>(b);
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

// MODE 0: plain GEMM rows; 1: 3x3 convolution; 2: 3x3 convolution reading a half-resolution source
//
// Geometries (BM x BN block, WM x WN wavefronts, NST ring stages):
//   256 x 128, 4 x 2 waves (64 x 64 each),  3 stages (144 KiB): loads two K-tiles ahead
//   256 x 256, 2 x 4 waves (128 x 64 each), 2 stages (128 KiB): A fetched once for N = 256,
//       0.75 fragment reads per MFMA, half the barriers per FLOP
//   128 x 128, 2 x 2 waves (64 x 64 each),  2 stages ( 64 KiB): two workgroups per CU (short problems)
template <int I> __device__ inline void lds_read_b128(u32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(I * 4096));
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename TC> struct Mma16;
template <> struct Mma16<bf16_t> {
  __device__ static inline void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<f16_t> {
  __device__ static inline void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
#if T2P_PART_DMA
__device__ inline bool g_stagger_dbg(int dbg) { return (dbg & 128) == 0; }   // debug bit 128 turns the stagger off
template <int I> __device__ inline void lds_read_b128_2k(u32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(I * 2048));
}
template <int I> __device__ inline void lds_read_b128_1k(u32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(I * 1024));
}

// Read-back half of the lean epilogue for one 64 x 64 slab, specialised at compile time so that the
// loop is branch-free: 8 LDS reads and (HAS_R) 8 residual loads at a time are in flight before the
// first use.  OUT: 0 fp32, 1 compute dtype, 2 GEGLU (interleaved value / gate columns ->
// compute dtype), 3 raw split-K partial.  Every variant issues exactly 16 stores.
// residual stored at half resolution (up-sampling blocks): row (b, y, x) reads residual row (b, y / 2, x / 2)
struct LeanRup { int on, W, HW, b_first, b_edge, hw4, w2; unsigned row_bytes, col_bytes; };

template <typename TC, int OUT, bool HAS_R, bool STATS, bool RL = false>
__device__ __forceinline__ void lean_slab(const float* sp, const __amdgpu_buffer_rsrc_t rC, const __amdgpu_buffer_rsrc_t rR,
                                          unsigned voc, const unsigned stc, unsigned vor, const unsigned str,
                                          const float4 bn0, const float4 bn1, const int row_first, const int b_edge,
                                          const float alpha, float* stats_dst, const LeanRup rup) {
  if constexpr (OUT == 3) {
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
      f32x4_t v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *(const f32x4_t*)(sp + (h8 * 8 + i) * 256);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[i]), rC, voc, 0, 0);
        voc += stc;
      }
    }
    return;
  }
  float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f, cq0 = 0.f, cq1 = 0.f, cq2 = 0.f, cq3 = 0.f;
#pragma unroll
  for (int h8 = 0; h8 < 2; ++h8) {
    f32x4_t v[8], rv[8];
    if constexpr (!HAS_R) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *(const f32x4_t*)(sp + (h8 * 8 + i) * 256);
    }
    if constexpr (HAS_R) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned off = vor;
        if (rup.on) {                                  // wave-uniform; a tile spans at most two samples
          const int row = row_first + 4 * (h8 * 8 + i);
          const int bidx = rup.b_first + (row >= rup.b_edge ? 1 : 0);
          const int rem = row - bidx * rup.HW;
          const int y = rem / rup.W, x = rem - y * rup.W;
          off = vor == DMA_OOB ? DMA_OOB : (unsigned)(bidx * rup.hw4 + (y >> 1) * rup.w2 + (x >> 1)) * rup.row_bytes + rup.col_bytes;
        }
        if constexpr (RL) rv[i] = unpack4<TC>(__builtin_amdgcn_raw_buffer_load_b64(rR, off, 0, 0));   // 16-bit residual
        else rv[i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0));
        if (!rup.on) vor += str;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int it = h8 * 8 + i;
      if constexpr (HAS_R) {                          // 8 residual rows in flight, LDS reads 4 at a time (register budget)
        if (i % 4 == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[i + k] = *(const f32x4_t*)(sp + (it + k) * 256);
        }
      }
      const bool second = row_first + 4 * it >= b_edge;
      float a0 = v[i][0] + (second ? bn1.x : bn0.x), a1 = v[i][1] + (second ? bn1.y : bn0.y);
      float a2 = v[i][2] + (second ? bn1.z : bn0.z), a3 = v[i][3] + (second ? bn1.w : bn0.w);
      if constexpr (HAS_R) { a0 += rv[i][0]; a1 += rv[i][1]; a2 += rv[i][2]; a3 += rv[i][3]; }
      if constexpr (OUT == 2) {
        // columns are interleaved (value_j, gate_j): out[row][col / 2 + {0, 1}] = value * gelu_erf(gate)
        // (GEGLU.forward, reference model/attention.py:42-44), stored in the compute dtype
        __builtin_amdgcn_raw_buffer_store_b32(pack2<TC>(a0 * gelu_erf_fast(a1), a2 * gelu_erf_fast(a3)), rC, voc, 0, 0);
      } else {
        a0 *= alpha; a1 *= alpha; a2 *= alpha; a3 *= alpha;
        if constexpr (STATS) {
          cs0 += a0; cs1 += a1; cs2 += a2; cs3 += a3;
          cq0 += a0 * a0; cq1 += a1 * a1; cq2 += a2 * a2; cq3 += a3 * a3;
        }
        if constexpr (OUT == 0) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4_t){a0, a1, a2, a3}), rC, voc, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){pack2<TC>(a0, a1), pack2<TC>(a2, a3)}, rC, voc, 0, 0);
        }
      }
      voc += stc;
    }
  }
  if constexpr (STATS) {
    // lanes l, l+16, l+32, l+48 hold the same 4 columns -> two wavefront shuffles; fixed order, reproducible
    cs0 += __shfl_xor(cs0, 16, 64); cs1 += __shfl_xor(cs1, 16, 64); cs2 += __shfl_xor(cs2, 16, 64); cs3 += __shfl_xor(cs3, 16, 64);
    cq0 += __shfl_xor(cq0, 16, 64); cq1 += __shfl_xor(cq1, 16, 64); cq2 += __shfl_xor(cq2, 16, 64); cq3 += __shfl_xor(cq3, 16, 64);
    cs0 += __shfl_xor(cs0, 32, 64); cs1 += __shfl_xor(cs1, 32, 64); cs2 += __shfl_xor(cs2, 32, 64); cs3 += __shfl_xor(cs3, 32, 64);
    cq0 += __shfl_xor(cq0, 32, 64); cq1 += __shfl_xor(cq1, 32, 64); cq2 += __shfl_xor(cq2, 32, 64); cq3 += __shfl_xor(cq3, 32, 64);
    if (stats_dst) {
      *(float4*)stats_dst = make_float4(cs0, cq0, cs1, cq1);
      *(float4*)(stats_dst + 4) = make_float4(cs2, cq2, cs3, cq3);
    }
  }
}


// buffer descriptor covering `bytes` from `base`: everything beyond reads as zero / is dropped on store.  Built from
// readfirstlane'd scalars so that hipcc keeps it in SGPRs (no waterfall loop around the DMA).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int bytes) {
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const int nb = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}

// Epilogue of the LDS-DMA kernels: the accumulators of a BM x BN tile (8 or 4 wavefronts, WM x WN) -> bias / time-embedding
// bias / residual / GEGLU / GroupNorm column statistics -> C, staged through this wave's 16 KiB slice of the (idle) LDS ring.
template <typename TC, int BM, int BN, int WM, int WN, bool MF16, int TI, int TJ>
__device__ __forceinline__ void dma_epilogue(const GemmParams& p, unsigned char* smem, f32x16 (&acc)[MF16 ? 1 : TI][MF16 ? 1 : TJ],
                                             f32x4_t (&acc16)[MF16 ? 8 : 1][MF16 ? 4 : 1], const int m0, const int n0, const int z0,
                                             const int z1, const int nsplit, const int ks, const int dbg) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int HW = p.H * p.W;

  // ---- epilogue ----------------------------------------------------------------------------------
  // The accumulators (row-per-register, column-per-lane) are staged through this wave's private
  // 16 KiB slice of the now idle LDS ring, 64 rows at a time, and read back row-contiguous, so
  // that bias / residual loads and the output stores are 16-byte vectors covering whole 256-byte
  // row segments (per-lane dword stores are store-issue bound: 64 instructions per wave, not 16).
  if ((dbg & 1) && (MF16 ? acc16[0][0][0] : acc[0][0][0]) != 123.456f) return;
  __builtin_amdgcn_s_barrier();                       // every wave is done reading the last stage
  float* stg = (float*)(smem + wave * 16384);         // [64 rows][64 cols] fp32
  const long coff = (long)z0 * p.sC_z0 + (long)z1 * p.sC_z1;
  const float* R = p.R ? (p.r_lowp ? (const float*)((const uint16_t*)p.R + (long)z0 * p.sR_z0 + (long)z1 * p.sR_z1)
                                   : p.R + (long)z0 * p.sR_z0 + (long)z1 * p.sR_z1) : nullptr;
  const bool need_b = p.bias_bn || p.r_up;
  const int rpb = p.rows_per_batch;
  const int b_first = need_b ? m0 / rpb : 0;          // a BM-row tile spans at most two samples when rpb >= BM
  const int b_edge = (b_first + 1) * rpb;
  const int cq = (lane & 15) * 4;                     // this lane's 4 columns inside a 64-column slab
  float* ws = nsplit > 1 ? (float*)p.ws + (long)ks * p.M * p.N : nullptr;

  // Lean path (every hot launch): operands addressed through buffer descriptors -- 32-bit per-lane
  // byte offsets bumped by a constant per 4-row step, rows past M dropped / zero-filled by the range
  // check -- and no per-row integer division.  The generic loop further down keeps the rare cases
  // (row bias, up-sampled residual, samples shorter than a tile, unaligned strides, >= 2 GiB operands).
  const int c_cols = p.geglu ? p.N / 2 : p.N;
  const long c_bytes = ws ? (long)p.M * p.N * 4 : ((long)(p.M - 1) * p.ldc + c_cols) * (p.c_f32 ? 4 : 2);
  const long r_rows = p.r_up ? (long)(p.M / HW) * (p.H >> 1) * (p.W >> 1) : (long)p.M;
  const long r_bytes = R ? ((r_rows - 1) * p.ldr + p.N) * (p.r_lowp ? 2 : 4) : 0;
  const bool lean = !(dbg & 512) && (p.N & 3) == 0 && c_bytes < (1L << 31) && r_bytes < (1L << 31) &&
                    (ws != nullptr || (!p.bias_m && (!need_b || rpb >= BM) && (p.geglu ? p.ldc % 2 == 0 : p.ldc % 4 == 0) &&
                                       (!R || p.ldr % 4 == 0) && (!p.bias_bn || p.ld_bn % 4 == 0)));
  const __amdgpu_buffer_rsrc_t rC =
      make_rsrc(ws ? (const void*)ws : (p.c_f32 ? (const void*)((float*)p.C + coff) : (const void*)((TC*)p.C + coff)), lean ? (int)c_bytes : 0);
  const __amdgpu_buffer_rsrc_t rR = make_rsrc(R ? (const void*)R : (const void*)p.C, lean ? (int)r_bytes : 0);
  const bool two_b = p.bias_bn && b_edge < m0 + BM && b_edge < p.M;   // the tile spans two samples

  // One 64 x 64 slab of the wave tile.  A generic lambda called with compile-time slab indices: the
  // accumulator arrays are then indexed by constants whatever the size of the body (a `#pragma unroll`
  // loop this large is silently left rolled, which sends the accumulators to scratch memory).
  auto slab = [&](auto HI, auto HJ) {
      constexpr int hi = decltype(HI)::value, hj = decltype(HJ)::value;
      if (hi + hj > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // previous slab fully read back
      if (!(dbg & 2048)) {
      if constexpr (MF16) {
        // transposed 16x16 accumulator (weights are the first MFMA operand): row = lane & 15, and register v of lane group
        // g = lane >> 4 in column tile j is channel 32 (j >> 1) + 8 g + 4 (j & 1) + v of the 64-column block (hperm)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *(f32x4_t*)(stg + (i * 16 + (lane & 15)) * 64 + 32 * (j >> 1) + 8 * (lane >> 4) + 4 * (j & 1)) = acc16[4 * hi + i][4 * hj + j];
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v)
              stg[(i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh) * 64 + j * 32 + lr] = acc[2 * hi + i][2 * hj + j][v];
      }
      }
      // same-wave LDS write -> read: the compiler orders them (lgkmcnt); no barrier needed
      const int row0 = m0 + wm * (BM / WM) + hi * 64;
      const int col = n0 + wn * (BN / WN) + hj * 64 + cq;
      if (lean) {
        const int rq = lane >> 4;
        const bool col_ok = col < p.N;                // N % 4 == 0: the lane's 4 columns are in or out together
        const float* sp = stg + rq * 64 + cq;
        const unsigned esz = (ws || p.c_f32) ? 4u : 2u;
        const unsigned ldc_e = ws ? (unsigned)p.N : (unsigned)p.ldc;
        const unsigned voc = (col_ok && !(dbg & 1024)) ? ((unsigned)(row0 + rq) * ldc_e + (unsigned)(p.geglu ? col >> 1 : col)) * esz : DMA_OOB;
        const unsigned stc = 4u * ldc_e * esz;
        const unsigned rsz = p.r_lowp ? 2u : 4u;
        const unsigned vor = col_ok ? ((unsigned)(row0 + rq) * (unsigned)p.ldr + (unsigned)col) * rsz : DMA_OOB;
        const unsigned str = 4u * rsz * (unsigned)p.ldr;
        const LeanRup rup = {p.r_up, p.W, HW, b_first, b_edge, (p.H >> 1) * (p.W >> 1), p.W >> 1, rsz * (unsigned)p.ldr, rsz * (unsigned)col};
        float4 bn0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!ws && p.bias_n && col_ok) bn0 = *(const float4*)(p.bias_n + col);
        float4 bn1 = bn0;
        if (!ws && p.bias_bn && col_ok) {
          const float4 t = *(const float4*)(p.bias_bn + (long)b_first * p.ld_bn + col);
          bn0.x += t.x; bn0.y += t.y; bn0.z += t.z; bn0.w += t.w;
          if (two_b) {
            const float4 u = *(const float4*)(p.bias_bn + (long)(b_first + 1) * p.ld_bn + col);
            bn1.x += u.x; bn1.y += u.y; bn1.z += u.z; bn1.w += u.w;
          }
        }
        const int edge = two_b ? b_edge : 0x7fffffff;
        float* sd = (p.col_stats && lane < 16 && col_ok && row0 < p.M) ? p.col_stats + ((long)(row0 >> 6) * p.N + col) * 2 : nullptr;
#define T2P_LEAN(OUT, HR, ST)                                                                                             \
  do {                                                                                                                   \
    if (HR && p.r_lowp) lean_slab<TC, OUT, HR, ST, HR>(sp, rC, rR, voc, stc, vor, str, bn0, bn1, row0 + rq, edge, p.alpha, sd, rup); \
    else lean_slab<TC, OUT, HR, ST, false>(sp, rC, rR, voc, stc, vor, str, bn0, bn1, row0 + rq, edge, p.alpha, sd, rup); \
  } while (0)
        if (ws) T2P_LEAN(3, false, false);
        else if (p.geglu) T2P_LEAN(2, false, false);
        else if (p.c_f32) {
          if (R) { if (p.col_stats) T2P_LEAN(0, true, true); else T2P_LEAN(0, true, false); }
          else { if (p.col_stats) T2P_LEAN(0, false, true); else T2P_LEAN(0, false, false); }
        } else {
          if (R) { if (p.col_stats) T2P_LEAN(1, true, true); else T2P_LEAN(1, true, false); }
          else { if (p.col_stats) T2P_LEAN(1, false, true); else T2P_LEAN(1, false, false); }
        }
#undef T2P_LEAN
        return;
      }
      if (ws) {                                       // split-K: raw partial sums -> workspace [split][M][N]
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
          const int rl = it * 4 + (lane >> 4);
          const int row = row0 + rl;
          if (row >= p.M || col >= p.N) continue;
          const float4 a = *(const float4*)(stg + rl * 64 + cq);
          float* dst = ws + (long)row * p.N + col;
          if (col + 3 < p.N && (p.N & 3) == 0) *(float4*)dst = a;
          else {
            dst[0] = a.x;
            if (col + 1 < p.N) dst[1] = a.y;
            if (col + 2 < p.N) dst[2] = a.z;
            if (col + 3 < p.N) dst[3] = a.w;
          }
        }
        return;
      }
      const bool full4 = col + 3 < p.N;
      float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias_n) {
        if (full4) bn = *(const float4*)(p.bias_n + col);
        else {
          if (col < p.N) bn.x = p.bias_n[col];
          if (col + 1 < p.N) bn.y = p.bias_n[col + 1];
          if (col + 2 < p.N) bn.z = p.bias_n[col + 2];
        }
      }
      const bool vec_ok = full4 && (p.ldc % 4 == 0 || p.geglu) && (!R || p.ldr % 4 == 0) && (!p.bias_bn || p.ld_bn % 4 == 0);
      float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f, cq0 = 0.f, cq1 = 0.f, cq2 = 0.f, cq3 = 0.f;   // column sums / sums of squares
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int rl = it * 4 + (lane >> 4);
        const int row = row0 + rl;
        if (row >= p.M || col >= p.N) continue;
        float4 a = *(const float4*)(stg + rl * 64 + cq);
        int bidx = 0;
        if (need_b) bidx = rpb >= BM ? b_first + (row >= b_edge ? 1 : 0) : row / rpb;
        long rrow = row;
        if (p.r_up) {
          const int rem = row - bidx * HW;
          const int y = rem / p.W, x = rem - y * p.W;
          rrow = ((long)bidx * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
        }
        const float bm = p.bias_m ? p.bias_m[row] : 0.f;
        a.x += bm + bn.x; a.y += bm + bn.y; a.z += bm + bn.z; a.w += bm + bn.w;
        if (vec_ok) {
          if (p.bias_bn) {
            const float4 t = *(const float4*)(p.bias_bn + (long)bidx * p.ld_bn + col);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
          }
          if (R) {
            const float4 t = res_load4<TC>(R, rrow * p.ldr + col, p.r_lowp);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
          }
          if (p.geglu) {
            // columns are interleaved (value_j, gate_j): out[row][col / 2 + {0, 1}] = value * gelu_erf(gate)
            // (GEGLU.forward, reference model/attention.py:42-44), stored in the compute dtype
            const float g0 = 0.5f * a.y * (1.f + erff(a.y * 0.70710678118654752440f));
            const float g1 = 0.5f * a.w * (1.f + erff(a.w * 0.70710678118654752440f));
            TC* dst = (TC*)p.C + coff + (long)row * p.ldc + (col >> 1);
            union { TC e[2]; uint32_t u; } o;
            o.e[0] = from_f32<TC>(a.x * g0); o.e[1] = from_f32<TC>(a.z * g1);
            *(uint32_t*)dst = o.u;
            continue;
          }
          a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
          cs0 += a.x; cs1 += a.y; cs2 += a.z; cs3 += a.w;
          cq0 += a.x * a.x; cq1 += a.y * a.y; cq2 += a.z * a.z; cq3 += a.w * a.w;
          if (p.c_f32) {
            *(float4*)((float*)p.C + coff + (long)row * p.ldc + col) = a;
          } else {
            TC* dst = (TC*)p.C + coff + (long)row * p.ldc + col;
            union { TC e[4]; uint2 u; } o;
            o.e[0] = from_f32<TC>(a.x); o.e[1] = from_f32<TC>(a.y); o.e[2] = from_f32<TC>(a.z); o.e[3] = from_f32<TC>(a.w);
            *(uint2*)dst = o.u;
          }
        } else {
          float vals[4] = {a.x, a.y, a.z, a.w};
          for (int k = 0; k < 4; ++k) {
            if (col + k >= p.N) break;
            float val = vals[k];
            if (p.bias_bn) val += p.bias_bn[(long)bidx * p.ld_bn + col + k];
            if (R) val += res_load1<TC>(R, rrow * p.ldr + col + k, p.r_lowp);
            val *= p.alpha;
            if (p.c_f32) ((float*)p.C)[coff + (long)row * p.ldc + col + k] = val;
            else ((TC*)p.C)[coff + (long)row * p.ldc + col + k] = from_f32<TC>(val);
          }
        }
      }
      // GroupNorm statistics of the tensor just produced (consumed by gn_finalize_cols_kernel): per
      // 64-row chunk and column, sum and sum of squares of the final fp32 values.  Lanes l, l+16,
      // l+32, l+48 hold the same 4 columns -> two wavefront shuffles; fixed order, reproducible.
      if (p.col_stats) {
        cs0 += __shfl_xor(cs0, 16, 64); cs1 += __shfl_xor(cs1, 16, 64); cs2 += __shfl_xor(cs2, 16, 64); cs3 += __shfl_xor(cs3, 16, 64);
        cq0 += __shfl_xor(cq0, 16, 64); cq1 += __shfl_xor(cq1, 16, 64); cq2 += __shfl_xor(cq2, 16, 64); cq3 += __shfl_xor(cq3, 16, 64);
        cs0 += __shfl_xor(cs0, 32, 64); cs1 += __shfl_xor(cs1, 32, 64); cs2 += __shfl_xor(cs2, 32, 64); cs3 += __shfl_xor(cs3, 32, 64);
        cq0 += __shfl_xor(cq0, 32, 64); cq1 += __shfl_xor(cq1, 32, 64); cq2 += __shfl_xor(cq2, 32, 64); cq3 += __shfl_xor(cq3, 32, 64);
        if (lane < 16 && vec_ok) {
          float* dst = p.col_stats + ((long)(row0 >> 6) * p.N + col) * 2;
          *(float4*)dst = make_float4(cs0, cq0, cs1, cq1);
          *(float4*)(dst + 4) = make_float4(cs2, cq2, cs3, cq3);
        }
      }
  };
  slab(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  if constexpr (TJ / 2 > 1) slab(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  if constexpr (TI / 2 > 1) {
    slab(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    if constexpr (TJ / 2 > 1) slab(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  }
}

// ---- register epilogue of the 16x16x32 kernels ------------------------------------------------------------------------
// The 16x16x32 kernels feed the WEIGHT fragment as the MFMA's first operand, so that an accumulator tile comes out
// transposed: lane (m = lane & 15, g = lane >> 4) holds 4 consecutive OUTPUT CHANNELS (MFMA rows 4 g + v) of pixel row m.
// The weight rows of a wave's 64-column block are permuted while they are staged (hperm below): MFMA row 4 g + v of
// column tile j is channel 32 (j >> 1) + 8 g + 4 (j & 1) + v, so a lane owns channels [8 g, 8 g + 8) and
// [32 + 8 g, 32 + 8 g + 8) of its row: two 16-byte stores per 16-row tile in 16 bits, each instruction covering 16 rows
// x 64 contiguous bytes -- no LDS round trip (the staged epilogue writes 256 KiB of accumulators per tile through a
// 64 B/clk LDS store path and reads them back), half the store instructions, and the LDS ring stays free.
__device__ __forceinline__ int hperm(int rho) {       // LDS row (0..63) of a wave's weight block -> channel of the block
  const int j = rho >> 4, g = (rho >> 2) & 3, v = rho & 3;
  return 32 * (j >> 1) + 8 * g + 4 * (j & 1) + v;
}
// sum over the 16 lanes of a DPP row (lane & 15), fixed order: every lane ends with the total
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));  // row_half_mirror
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));  // row_mirror
  return x;
}

#ifndef T2P_C_AUX
#define T2P_C_AUX 0          // cache policy of the register epilogue's 16-bit output stores (measurement variants: 2 nt, 16 sc1, 17 sc0 sc1)
#endif
// The GemmParams of a gemm_dma / gemm_dxs kernel as they sit in the kernel-argument segment (the struct is the FIRST argument of every
// kernel that carries the register epilogue), behind an empty asm so that the compiler treats the pointer as opaque
typedef __attribute__((address_space(4))) GemmParams KernArgs;
__device__ __forceinline__ const KernArgs* kernel_args() {
  const KernArgs* pk = (const KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(pk));
  return pk;
}
// PLAIN: the caller knows at compile time that the launch is the network's common case -- 16-bit output, no split-K workspace, no GEGLU,
// no per-row bias, no up-sampling phase: the per-tile uniform branches on those (a dozen per 16-row tile) and their scalar bookkeeping
// fold away.  Same arithmetic, same order: bit-identical to the general form.
template <typename TC, int BM, int BN, int WM, int WN, bool CFRAG = false, bool PLAIN = false, typename PT = GemmParams>
__device__ __forceinline__ void reg_epilogue(const PT& p, f32x4_t (&acc16)[8][4], const int m0, const int n0, const int z0,
                                             const int z1, const int nsplit, const int ks, const int dbg) {
  if ((dbg & 1) && acc16[0][0][0] != 123.456f) return;
  int tx = threadIdx.x;
  asm volatile("" : "+v"(tx));       // nothing lane-derived of the epilogue is computed before the K loop and kept across it (registers)
  const int lane = tx & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tx >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, g4 = lane >> 4;
  const int HW = p.H * p.W;
  const long coff = (long)z0 * p.sC_z0 + (long)z1 * p.sC_z1;
  const float* R = p.R ? (p.r_lowp ? (const float*)((const uint16_t*)p.R + (long)z0 * p.sR_z0 + (long)z1 * p.sR_z1)
                                   : p.R + (long)z0 * p.sR_z0 + (long)z1 * p.sR_z1) : nullptr;
  static_assert(!(PLAIN && CFRAG), "the plain form stores row-major only");
  float* ws = (!PLAIN && nsplit > 1) ? (float*)p.ws + (long)ks * p.M * p.N : nullptr;
  const bool p_geglu = PLAIN ? false : p.geglu != 0, p_c_f32 = PLAIN ? false : p.c_f32 != 0;
  const float* const p_bias_m = PLAIN ? nullptr : p.bias_m;
  const int c_cols = p_geglu ? p.N / 2 : p.N;
  const long c_bytes = ws ? (long)p.M * p.N * 4 : ((long)(p.M - 1) * p.ldc + c_cols) * (p_c_f32 ? 4 : 2);
  const long r_rows = p.r_up ? (long)(p.M / HW) * (p.H >> 1) * (p.W >> 1) : (long)p.M;
  const long r_bytes = R ? ((r_rows - 1) * p.ldr + p.N) * (p.r_lowp ? 2 : 4) : 0;
  const __amdgpu_buffer_rsrc_t rC =
      make_rsrc(ws ? (const void*)ws : (p_c_f32 ? (const void*)((float*)p.C + coff) : (const void*)((TC*)p.C + coff)), (int)c_bytes);
  const __amdgpu_buffer_rsrc_t rR = make_rsrc(R ? (const void*)R : (const void*)p.C, (int)r_bytes);

  const int row_w = m0 + wm * (BM / WM);                 // first row of the wave tile (128 rows = 8 tiles of 16)
  // up_phase (MODE 3): rows are pixels of the half-resolution map; row (b, yl, xl) is stored at output pixel (b, 2 yl + py, 2 xl + px)
  const int upp = PLAIN ? 0 : (p.up_phase == 5 ? 1 + (int)blockIdx.y : p.up_phase);     // 5: all four phases in one launch, the phase on grid.y
  const int Mrows = upp ? p.M >> 2 : p.M;
  const int Wl = p.W >> 1, HWl = (p.H >> 1) * (p.W >> 1);
  const int col0 = n0 + wn * (BN / WN) + 8 * g4;         // this lane's channels: [col0, col0 + 8) and [col0 + 32, col0 + 40)
  const bool ok0 = col0 < p.N, ok1 = col0 + 32 < p.N;    // N % 8 == 0: a group of 8 channels is in or out as a whole
  // per-channel constants: bias_n + the time-embedding bias of the tile's first sample
  float bs[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) bs[k] = 0.f;
  const int rpb = p.rows_per_batch;
  const bool need_b = p.bias_bn || p.r_up;
  const int b_first = upp ? m0 / HWl : (need_b ? m0 / rpb : 0);
  if (!ws) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = col0 + 32 * h;
      if (c < p.N) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.bias_n) t = *(const float4*)(p.bias_n + c + 4 * q);
          if (p.bias_bn) {
            const float4 a = *(const float4*)(p.bias_bn + (long)b_first * p.ld_bn + c + 4 * q);
            t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
          }
          const int k = 8 * h + 4 * q;
          bs[k] = t.x; bs[k + 1] = t.y; bs[k + 2] = t.z; bs[k + 3] = t.w;
        }
      }
    }
  }
  const unsigned esz = (ws || p_c_f32) ? 4u : 2u;
  const unsigned ldc_e = ws ? (unsigned)p.N : (unsigned)p.ldc;
  const unsigned rsz = p.r_lowp ? 2u : 4u;
  const bool nostore = (dbg & 1024) != 0;
  const bool alpha_not_one = p.alpha != 1.0f;
  float cs[16], cq[16];                                   // GroupNorm column sums of a 64-row chunk (4 row tiles)

  // rows of 16-row tile i: the output row of this lane and (r_up) the half-resolution residual row
  auto rows_of = [&](int i, int& row, unsigned& rrow, int& bsel) __attribute__((always_inline)) {
    row = row_w + 16 * i + l16;
    bsel = 0;
    rrow = (unsigned)row;
    if (upp) {                                            // wave-uniform; Wl % 16 == 0: a tile's 16 rows share (b, yl)
      const int rt = row_w + 16 * i;
      const int bidx = rt / HWl, rem = rt - bidx * HWl, yl = rem / Wl, xl = rem - yl * Wl;
      bsel = bidx - (m0 / HWl);
      row = rt < Mrows ? (bidx * p.H + 2 * yl + ((upp - 1) >> 1)) * p.W + 2 * (xl + l16) + ((upp - 1) & 1) : p.M;
      rrow = (unsigned)row;
    } else if (need_b) {
      const int rt = row_w + 16 * i;                      // wave-uniform: the 16 rows of a tile lie in one sample (rpb % 16 == 0)
      const int bidx = rt / rpb;
      bsel = bidx - b_first;                              // 0 inside the tile's first sample
      if (p.r_up) {
        const int rem = rt - bidx * HW, y = rem / p.W, x = rem - y * p.W;     // x % 16 == 0 (W % 16 == 0)
        rrow = (unsigned)((bidx * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + ((x + l16) >> 1));
      }
    }
  };
  // residual of one tile -> registers (RM 1: 16-bit residual, two 16-byte loads; RM 2: fp32, four)
  auto load_res = [&](auto RMC, int i, u32x4_t* rr) __attribute__((always_inline)) {
    constexpr int RM = decltype(RMC)::value;
    int row, bsel; unsigned rrow;
    rows_of(i, row, rrow, bsel);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned off = ((h ? ok1 : ok0) && row < p.M) ? (rrow * (unsigned)p.ldr + (unsigned)(col0 + 32 * h)) * rsz : DMA_OOB;
      if constexpr (RM == 1) {
        rr[h] = __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0);
      } else {
        rr[2 * h] = __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0);
        rr[2 * h + 1] = __builtin_amdgcn_raw_buffer_load_b128(rR, off == DMA_OOB ? DMA_OOB : off + 16, 0, 0);
      }
    }
  };
  // one 16-row tile; IC is a compile-time index so that the accumulators stay in registers
  auto tile = [&](auto IC, auto RMC, const u32x4_t* rr) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    constexpr int RM = decltype(RMC)::value;
    int row, bsel; unsigned rrow;
    rows_of(i, row, rrow, bsel);
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * j + e] = acc16[i][j][e];       // v[k]: channel col0 + (k & 7) + 32 (k >> 3)
    if (!ws) {
      float bm = 0.f;
      if (p_bias_m) bm = row < p.M ? p_bias_m[row] : 0.f;
      if (p.bias_bn && bsel > 0) {                       // the tile runs into a following sample (small maps): per-tile reload
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int c = col0 + (k & 7) + 32 * (k >> 3);
          v[k] += bm + (c < p.N ? (p.bias_n ? p.bias_n[c] : 0.f) + p.bias_bn[(long)(b_first + bsel) * p.ld_bn + c] : 0.f);
        }
      } else if constexpr (PLAIN) {                      // (no per-row bias: bm is zero)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += bs[k];
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] += bm + bs[k];
      }
      if constexpr (RM == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4_t a = unpack4<TC>((u32x2_res_t){rr[h][0], rr[h][1]}), b = unpack4<TC>((u32x2_res_t){rr[h][2], rr[h][3]});
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[8 * h + e] += a[e]; v[8 * h + 4 + e] += b[e]; }
        }
      } else if constexpr (RM == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4_t a = __builtin_bit_cast(f32x4_t, rr[2 * h]), b = __builtin_bit_cast(f32x4_t, rr[2 * h + 1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[8 * h + e] += a[e]; v[8 * h + 4 + e] += b[e]; }
        }
      }
    }
    const bool rok = row < p.M && !nostore;
    if (!ws && p_geglu) {
      // interleaved (value, gate) columns: out[row][c / 2] = value * gelu_erf(gate)  (GEGLU.forward, model/attention.py:42-44)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned off = ((h ? ok1 : ok0) && rok) ? ((unsigned)row * ldc_e + (unsigned)((col0 + 32 * h) >> 1)) * 2u : DMA_OOB;
        const float* w = v + 8 * h;
        __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){pack2<TC>(w[0] * gelu_erf_fast(w[1]), w[2] * gelu_erf_fast(w[3])),
                                                         pack2<TC>(w[4] * gelu_erf_fast(w[5]), w[6] * gelu_erf_fast(w[7]))}, rC, off, 0, 0);
      }
      return;
    }
    if (!ws) {
      if (!PLAIN || alpha_not_one) {                     // (x * 1 = x exactly: skipping the product changes nothing)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] *= p.alpha;
      }
      if (p.col_stats) {
        if (row < p.M) {
#pragma unroll
          for (int k = 0; k < 16; ++k) { cs[k] += v[k]; cq[k] += v[k] * v[k]; }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned off = ((h ? ok1 : ok0) && rok) ? ((unsigned)row * ldc_e + (unsigned)(col0 + 32 * h)) * esz : DMA_OOB;
      const float* w = v + 8 * h;
      if (esz == 4u) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4_t){w[0], w[1], w[2], w[3]}), rC, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4_t){w[4], w[5], w[6], w[7]}), rC, off == DMA_OOB ? DMA_OOB : off + 16, 0, 0);
      } else {
        const u32x4_t packed = {pack2<TC>(w[0], w[1]), pack2<TC>(w[2], w[3]), pack2<TC>(w[4], w[5]), pack2<TC>(w[6], w[7])};
        if constexpr (CFRAG) {
          // columns from frag_col0 on: fragment-major (GemmParams::c_frag).  The 16 rows of a tile lie in one sample (rpb % 16 == 0)
          const int c = col0 + 32 * h - p.frag_col0;              // wave-uniform sign: frag_col0 % 64 == 0
          if (c >= 0) {
            const int rt = row_w + 16 * i;
            const int bz = p.rows_per_batch > 1 ? rt / p.rows_per_batch : z0;
            const int rl = row - (p.rows_per_batch > 1 ? bz * p.rows_per_batch : 0);
            const long e = (long)bz * p.frag_bstride + ((long)(rl >> 5) * p.frag_ns + (c >> 5)) * 1024 + ((c >> 3) & 1) * 512 + ((c >> 4) & 1) * 256 +
                           (rl & 31) * 8;
            if ((h ? ok1 : ok0) && rok) *(u32x4_t*)((TC*)p.c_frag + e) = packed;
            continue;
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(packed, rC, off, 0, T2P_C_AUX);
      }
    }
  };
  // GroupNorm statistics of a 64-row chunk: column sums folded over the 16 lanes of a row group, written by lane m = 0
  auto flush_stats = [&](int chunk_row) __attribute__((always_inline)) {
    if (!p.col_stats || ws) return;
#pragma unroll
    for (int k = 0; k < 16; ++k) { cs[k] = row16_sum(cs[k]); cq[k] = row16_sum(cq[k]); }
    if (l16 == 0 && chunk_row < Mrows) {
      long chunk = chunk_row >> 6;
      if (upp) {            // the consumer sums the HW / 64 chunks of a sample: a phase fills its quarter of each sample's range
        const int bidx = chunk_row / HWl;
        chunk = (long)bidx * ((p.H * p.W) >> 6) + (long)(upp - 1) * (HWl >> 6) + ((chunk_row - bidx * HWl) >> 6);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h ? ok1 : ok0) {
          float* dst = p.col_stats + (chunk * p.N + col0 + 32 * h) * 2;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(float4*)(dst + 4 * q) = make_float4(cs[8 * h + 2 * q], cq[8 * h + 2 * q], cs[8 * h + 2 * q + 1], cq[8 * h + 2 * q + 1]);
        }
      }
    }
  };
  // a half of the wave tile (4 row tiles = one statistics chunk); the residual rows are requested ahead of their use:
  // three tiles ahead in 16 bits (6 loads in flight), two tiles at a time in fp32 (8 loads)
  auto half = [&](auto HC, auto RMC) __attribute__((always_inline)) {
    constexpr int i0 = 4 * decltype(HC)::value;
    constexpr int RM = decltype(RMC)::value;
#pragma unroll
    for (int k = 0; k < 16; ++k) cs[k] = cq[k] = 0.f;
    if constexpr (RM == 1) {
      u32x4_t rr[3][2];                                  // three tiles requested ahead (registers: 229 + these must stay <= 256)
      load_res(RMC, i0, rr[0]); load_res(RMC, i0 + 1, rr[1]); load_res(RMC, i0 + 2, rr[2]);
      tile(std::integral_constant<int, i0>{}, RMC, rr[0]);
      load_res(RMC, i0 + 3, rr[0]);
      tile(std::integral_constant<int, i0 + 1>{}, RMC, rr[1]); tile(std::integral_constant<int, i0 + 2>{}, RMC, rr[2]);
      tile(std::integral_constant<int, i0 + 3>{}, RMC, rr[0]);
    } else if constexpr (RM == 2) {
      u32x4_t rr[2][4];
      load_res(RMC, i0, rr[0]); load_res(RMC, i0 + 1, rr[1]);
      tile(std::integral_constant<int, i0>{}, RMC, rr[0]); tile(std::integral_constant<int, i0 + 1>{}, RMC, rr[1]);
      load_res(RMC, i0 + 2, rr[0]); load_res(RMC, i0 + 3, rr[1]);
      tile(std::integral_constant<int, i0 + 2>{}, RMC, rr[0]); tile(std::integral_constant<int, i0 + 3>{}, RMC, rr[1]);
    } else {
      tile(std::integral_constant<int, i0>{}, RMC, nullptr); tile(std::integral_constant<int, i0 + 1>{}, RMC, nullptr);
      tile(std::integral_constant<int, i0 + 2>{}, RMC, nullptr); tile(std::integral_constant<int, i0 + 3>{}, RMC, nullptr);
    }
    flush_stats(row_w + 16 * i0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  if (!R || ws) { half(I0{}, I0{}); half(I1{}, I0{}); }
  else if (p.r_lowp) { half(I0{}, I1{}); half(I1{}, I1{}); }
  else { half(I0{}, I2{}); half(I1{}, I2{}); }
}

// MF16: use v_mfma_f32_16x16x32 (sustains a higher clock than 32x32x16 at equal cycles per FLOP in
// LDS-fed loops, MI355X_MICROARCH.md "DVFS give-back" item 7) -- 128 x 64 wave tiles only.
template <typename TC, int BM, int BN, int WM, int WN, int MODE, int NST, int NSTB, bool MF16, bool REGE, bool CFRAG>
__device__ __forceinline__ void gemm_dma_body(const GemmParams& p, const int tiles_m, const int tiles_n, const int dbg_arg) {
  // dbg: timing-only ablation mask.  Bits 128 / 256 change the DMA issue order only (results unchanged); every
  // other bit skips work and gives wrong results: those exist in -DT2P_ABLATION builds only (never shipped).
#ifdef T2P_ABLATION
  const int dbg = dbg_arg;
#else
  const int dbg = dbg_arg & (128 | 256 | 4096 | 8192);
#endif
  // NST stages for the A (activation) tile, NSTB for the B (weight) tile.  NSTB < NST gives the
  // activations -- which come from L2 / Infinity Cache -- a longer lead than the L2-hot weights
  // within the 160 KiB of LDS (256 x 256: 3 x 32 KiB + 2 x 32 KiB).
  constexpr int BK = 64;
  constexpr int NW = WM * WN;
  constexpr int TI = BM / WM / 32, TJ = BN / WN / 32;  // 32x32 MFMA tiles per wave
  static_assert(NW == 8 || NW == 4, "4 or 8 wavefronts");
  static_assert((TI == 2 || TI == 4) && (TJ == 2 || TJ == 4) && TI * TJ <= 8, "wave tile");
  static_assert(NW * 16384 <= NST * BM * 128 + NSTB * BN * 128, "epilogue staging must fit the ring");
  static_assert(NSTB == NST || NSTB == NST - 1, "B ring is as deep as the A ring or one stage shallower");
  static_assert(!MF16 || (TI == 4 && TJ == 2), "the 16x16x32 variant is written for 128 x 64 wave tiles");
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // DMA instructions per wave per K-tile (8 rows each)
  constexpr int ASTAGE = BM * 128, BSTAGE = BN * 128;
  constexpr int BRING = NST * ASTAGE;                            // byte offset of the B ring
  constexpr int TAPS = MODE == 0 ? 1 : (MODE == 3 ? 4 : 9);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;

  // XCD-aware tile order: blocks b and b + 8 share an XCD; give each XCD a contiguous tile range
  const int ntiles = tiles_m * tiles_n;
  int tile = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // MODE 3: the four output phases of an up-sampling convolution are the y dimension of one grid (up_phase == 5): a phase
  // launch of a low-resolution level has 64 workgroups, the four together fill the chip
  const int upk = MODE == 3 ? (p.up_phase == 5 ? 1 + (int)blockIdx.y : p.up_phase) : 0;
  const int z0 = MODE == 3 ? 0 : blockIdx.y / p.nz1, z1 = MODE == 3 ? 0 : blockIdx.y % p.nz1;

  const int Ctot = p.C0 + p.C1;
  const int nch = (Ctot + BK - 1) / BK;
  // MODE 1 only: an extra K segment read at the centre tap from X0 | X1 (the 1x1 shortcut of a residual block folded into its
  // second convolution): nchx more 64-channel K-tiles after the nch * 9 main ones
  const int nchx = MODE == 1 ? (p.CX0 + p.CX1) / BK : 0;
  const int nk_all = nch * TAPS + nchx;
  // split-K: blockIdx.z owns K-tiles [kt_lo, kt_hi) and writes a raw fp32 partial tile
  const int nsplit = gridDim.z, ks = blockIdx.z;
  const int kt_lo = (int)((long)nk_all * ks / nsplit), kt_hi = (int)((long)nk_all * (ks + 1) / nsplit);
  const int nk = kt_hi - kt_lo;
  const int HW = p.H * p.W;
  const int Hs = MODE >= 2 ? (p.H >> 1) : p.H, Ws = MODE >= 2 ? (p.W >> 1) : p.W;
  // MODE 3: one output phase (py, px) of a 3x3 convolution on a 2x nearest-up-sampled map = a 2x2 convolution on the
  // half-resolution map (the taps that read the same source pixel are summed into one weight on the host): rows are the
  // M / 4 source pixels, tap (ty, tx) reads pixel (yl + ty - 1 + py, xl + tx - 1 + px), the epilogue scatters row
  // (b, yl, xl) to output pixel (b, 2 yl + py, 2 xl + px).  4 / 9 of the multiplications of the gather form (MODE 2).
  const int up_py = MODE == 3 ? ((upk - 1) >> 1) : 0, up_px = MODE == 3 ? ((upk - 1) & 1) : 0;
  const int Mq = MODE == 3 ? p.M >> 2 : p.M;            // rows of this launch
  const int HWq = MODE == 3 ? Hs * Ws : HW, Wq = MODE == 3 ? Ws : p.W, Hq = MODE == 3 ? Hs : p.H;

  // buffer descriptors: whole operand in range, everything else reads as zero.  Built from
  // readfirstlane'd scalars so that hipcc keeps them in SGPRs (no waterfall loop around the DMA).
  const long a_rows = MODE == 0 ? (long)p.M : (long)(p.M / HW) * Hs * Ws;
  const TC* A0p = (const TC*)p.A0 + (long)z0 * p.sA_z0 + (long)z1 * p.sA_z1;
  const TC* Bp = (const TC*)p.Bw + (long)z0 * p.sB_z0 + (long)z1 * p.sB_z1 +
                 ((MODE == 3 && p.up_phase == 5) ? (long)(upk - 1) * p.N * p.ldb : 0L);       // phase weights [4][N][ldb]
  const int a0_bytes = (int)(((a_rows - 1) * p.lda0 + p.C0) * 2);
  const int a1_bytes = p.A1 ? (int)(((a_rows - 1) * p.lda1 + p.C1) * 2) : 0;
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bp, (int)((((long)p.N - 1) * p.ldb + (long)TAPS * Ctot + nchx * BK) * 2));
  const unsigned lda0_2 = (unsigned)(p.lda0 * 2), lda1_2 = (unsigned)(p.lda1 * 2);
  const unsigned ldx0_2 = (unsigned)(p.ldx0 * 2), ldx1_2 = (unsigned)(p.ldx1 * 2);
  // (fields copied out: `c ? p.a : p.b` is an lvalue conditional -- an address select into the by-value argument struct, which
  // then lands in scratch memory)
  const void* const X0p = p.X0; const void* const X1p = p.X1; const void* const A1p = p.A1;
  const int pC0 = p.C0, pCX0 = p.CX0;
  const int x0_bytes = nchx ? (int)((((long)p.M - 1) * p.ldx0 + p.CX0) * 2) : 0;
  const int x1_bytes = (nchx && p.X1) ? (int)((((long)p.M - 1) * p.ldx1 + p.CX1) * 2) : 0;

  // per-lane DMA geometry: instruction j of this wave covers tile rows (wave * INSTR + j) * 8 + (lane >> 3).
  // K order is chunk-major (all 9 taps of one 64-channel slice back to back: the 3x3 window
  // re-reads stay in the XCD's L2).  For MODE 1 the source row of tap (dy, dx) is m + dy*W + dx
  // (NHWC rows are pixel-major), so per K-tile the lane adds one wave-uniform byte delta to a
  // precomputed offset; bit t of a_vm says whether tap t lands inside the map.
  const int prow = lane >> 3, ppos = lane & 7;
  // a lane keeps the row m of each of its instructions (24-bit: offset = m * row stride + chunk through one v_mad_u32_u24,
  // whatever the source), the swizzled chunk offset (it depends on the parity of j only) and the tap-validity bits
  static_assert(A_INSTR % 2 == 0, "the chunk swizzle of instruction j depends on j & 1 only when A_INSTR is even");
  unsigned a_m[A_INSTR], a_vm[A_INSTR], a_chk[2];
  int a_y[A_INSTR], a_x[A_INSTR], a_bb[A_INSTR];       // MODE 2 only
  a_chk[0] = (unsigned)((ppos ^ ((prow >> 1) & 7)) * 16);
  a_chk[1] = (unsigned)((ppos ^ ((4 + (prow >> 1)) & 7)) * 16);
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + prow;
    const int m = m0 + r;
    unsigned vm = 0;
    int y = 0, x = 0, b = 0;
    if (MODE != 0) {
      b = m / HWq;
      const int rem = m - b * HWq;
      y = rem / Wq;
      x = rem - y * Wq;
    }
    if (m < Mq) {
      if (MODE == 0) {
        vm = 1;
      } else if (MODE == 3) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int sy = y + (t >> 1) - 1 + up_py, sx = x + (t & 1) - 1 + up_px;
          if (sy >= 0 && sy < Hq && sx >= 0 && sx < Wq) vm |= 1u << t;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int sy = y + t / 3 - 1, sx = x + t % 3 - 1;
          if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) vm |= 1u << t;
        }
      }
    }
    a_vm[j] = vm;
    a_y[j] = y; a_x[j] = x; a_bb[j] = b * Hs * Ws;
    a_m[j] = (unsigned)m;                            // MODE 2 recomputes the row per tap
  }
  unsigned b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + prow;
    const int n = n0 + (MF16 ? (r & ~63) + hperm(r & 63) : r);     // 16x16x32: permuted channels (see reg_epilogue)
    b_off[j] = n < p.N ? (unsigned)((long)n * p.ldb * 2) + (unsigned)((ppos ^ ((r >> 1) & 7)) * 16) : DMA_OOB;
  }

  // Issue-stream state (wave-uniform, SGPRs).  The A and the B stream are each issued strictly in K order, every K-tile once,
  // so each stream carries the position of its NEXT K-tile along: (chunk, tap, ring stage) plus everything an issue needs in
  // ready-made form -- the source's buffer descriptor and row stride (they change with the 64-channel chunk), and the
  // wave-uniform byte offset of the tap, advanced by one row stride per tap.  The common case of an issue is then a handful of
  // scalar additions; the recomputation at a chunk boundary sits in a block that one issue in nine enters (the empty asm
  // statements keep the compiler from turning those blocks into select chains executed every time).  The scalar prelude of
  // an issue is what the two waves of a SIMD cannot hide from each other: ten more scalar instructions per issue measured
  // 3.7 % on the convolutions.
  int ia_chunk = kt_lo / TAPS, ia_tap = kt_lo - (kt_lo / TAPS) * TAPS;
  if (nchx && kt_lo >= nch * TAPS) { ia_chunk = nch + (kt_lo - nch * TAPS); ia_tap = 0; }     // a K-split that starts in the extra segment
  int ib_chunk = ia_chunk, ib_tap = ia_tap;
  unsigned ia_so = 0, ib_so = 0;                        // ring stage byte offsets
  int ia_dx = 0;                                        // MODE 1: column of the tap (-1, 0, 1): the row of taps ends after dx = 1
  __amdgpu_buffer_rsrc_t rA = make_rsrc(A0p, a0_bytes);
  unsigned a_ld2 = lda0_2, a_delta = 0, a_rowstep = 0;
  bool a_extra = false;
  // descriptor / stride / offsets of chunk ia_chunk, tap ia_tap
  auto a_set_chunk = [&]() __attribute__((always_inline)) {
    const bool extra = MODE == 1 && ia_chunk >= nch;   // K-tile of the shortcut segment (centre tap of X0 | X1)
    const int c0 = (extra ? ia_chunk - nch : ia_chunk) * BK;     // channel base of this K-tile
    int cfirst = pC0;
    if (extra) cfirst = pCX0;
    const bool second = c0 >= cfirst;                  // C0 % 64 == 0
    const int csrc = second ? c0 - cfirst : c0;
    // (plain assignments: a nested `c ? a : b` over named variables is an address select that keeps them in memory)
    unsigned ld2 = lda0_2;
    const void* abase = (const void*)A0p;
    int abytes = a0_bytes;
    if (extra) {
      if (second) { ld2 = ldx1_2; abase = X1p; abytes = x1_bytes; } else { ld2 = ldx0_2; abase = X0p; abytes = x0_bytes; }
    } else if (second) {
      ld2 = lda1_2; abase = A1p; abytes = a1_bytes;
    }
    rA = make_rsrc(abase, abytes);
    a_ld2 = ld2;
    a_extra = extra;
    int dy = 0, dx = 0;
    if (MODE == 1) {
      if (extra) { ia_tap = 4; ia_dx = 1; }            // tap 4 = the centre; dx = 1: the next advance leaves the "row of taps"
      else { dy = ia_tap / 3 - 1; dx = ia_tap - (ia_tap / 3) * 3 - 1; ia_dx = dx; }
      a_rowstep = (unsigned)(Wq - 3) * ld2;            // from (dy, dx = 2) to (dy + 1, -1)
    } else if (MODE == 3) {
      dy = (ia_tap >> 1) - 1 + up_py; dx = (ia_tap & 1) - 1 + up_px;
      a_rowstep = (unsigned)(Wq - 2) * ld2;            // from (ty, tx = 2) to (ty + 1, 0)
    }
    a_delta = (unsigned)((dy * Wq + dx) * (int)ld2 + csrc * 2);
  };
  a_set_chunk();
  const int a_switch = p.A1 ? pC0 / BK : 0x7fffffff;    // MODE 0: the chunk at which the second source starts
  const unsigned Ctot2 = (unsigned)(Ctot * 2);
  unsigned b_kb = (MODE == 1 && ib_chunk >= nch) ? (unsigned)(((long)TAPS * Ctot + (ib_chunk - nch) * BK) * 2)
                                                 : (unsigned)(((long)ib_tap * Ctot + ib_chunk * BK) * 2);
  if (MODE == 1 && ib_chunk >= nch) ib_tap = TAPS - 1;
  auto issue = [&](int /* kl: the streams keep their own position */, int part) __attribute__((always_inline)) {   // part 0: A rows, 1: B rows, 2: both
    if (part != 1) {
      unsigned char* sta = smem + ia_so;
      if constexpr (MODE == 2) {
        const int dy = ia_tap / 3 - 1, dx = ia_tap - (ia_tap / 3) * 3 - 1;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
          const int row = a_bb[j] + ((a_y[j] + dy) >> 1) * Ws + ((a_x[j] + dx) >> 1);
          const unsigned voff = (unsigned)row * a_ld2 + a_delta + a_chk[j & 1];     // a_delta: the channel offset only
          const unsigned m = (dbg & 64) ? 0u : (unsigned)__builtin_amdgcn_sbfe((int)a_vm[j], (unsigned)ia_tap, 1u);
          unsigned char* dst = sta + (wave * A_INSTR + j) * 1024;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, T2P_LDS_PTR(dst), 16, (voff & m) | (DMA_OOB & ~m), 0, 0, 0);
        }
      } else {
        const unsigned vb0 = a_chk[0] + a_delta, vb1 = a_chk[1] + a_delta;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
          const unsigned voff = __umul24(a_m[j], a_ld2) + ((j & 1) ? vb1 : vb0);     // rows and strides are below 2^24 (dma_eligible)
          // tap validity: bit ia_tap of a_vm[j] spread over the word (v_bfe_i32), then offset-or-out-of-range in one bit-select
          const unsigned m = (dbg & 64) ? 0u : (unsigned)__builtin_amdgcn_sbfe((int)a_vm[j], (unsigned)ia_tap, 1u);
          unsigned char* dst = sta + (wave * A_INSTR + j) * 1024;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, T2P_LDS_PTR(dst), 16, (voff & m) | (DMA_OOB & ~m), 0, 0, 0);
        }
      }
      if constexpr (NST == 2) ia_so ^= (unsigned)ASTAGE;
      else ia_so = ia_so + ASTAGE == NST * ASTAGE ? 0u : ia_so + ASTAGE;
      // advance to the next K-tile of the stream
      if constexpr (MODE == 0) {
        ++ia_chunk; a_delta += BK * 2;
        if (ia_chunk == a_switch) { asm volatile(""); a_set_chunk(); }
      } else if constexpr (MODE == 1) {
        ++ia_tap; ++ia_dx; a_delta += a_ld2;
        if (ia_dx > 1) {
          asm volatile("");
          ia_dx = -1; a_delta += a_rowstep;
          if (ia_tap == TAPS || a_extra) { asm volatile(""); ia_tap = 0; ++ia_chunk; a_set_chunk(); }
        }
      } else if constexpr (MODE == 3) {
        ++ia_tap; a_delta += a_ld2;
        if (!(ia_tap & 1)) {
          asm volatile("");
          a_delta += a_rowstep;
          if (ia_tap == TAPS) { asm volatile(""); ia_tap = 0; ++ia_chunk; a_set_chunk(); }
        }
      } else {
        ++ia_tap;
        if (ia_tap == TAPS) { asm volatile(""); ia_tap = 0; ++ia_chunk; a_set_chunk(); }
      }
    }
    if (part != 0) {
      unsigned char* stb = smem + BRING + ib_so;
#pragma unroll
      for (int j = 0; j < B_INSTR; ++j) {
        const unsigned voff = (dbg & 64) ? DMA_OOB : b_off[j] + b_kb;   // rows beyond N carry DMA_OOB: adding b_kb (< 2 GiB) keeps them out of range
        unsigned char* dst = stb + (wave * B_INSTR + j) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, T2P_LDS_PTR(dst), 16, voff, 0, 0, 0);
      }
      if constexpr (NSTB == 2) ib_so ^= (unsigned)BSTAGE;
      else ib_so = ib_so + BSTAGE == NSTB * BSTAGE ? 0u : ib_so + BSTAGE;
      if constexpr (TAPS == 1) {
        ++ib_chunk; b_kb += BK * 2;
      } else {
        ++ib_tap; b_kb += Ctot2;
        if (ib_tap == TAPS) {
          asm volatile("");
          ++ib_chunk;
          if (MODE == 1 && ib_chunk >= nch) { b_kb = (unsigned)(((long)TAPS * Ctot + (ib_chunk - nch) * BK) * 2); ib_tap = TAPS - 1; }
          else { ib_tap = 0; b_kb += (unsigned)(BK * 2) - (unsigned)TAPS * Ctot2; }
        }
      }
    }
  };

  f32x16 acc[MF16 ? 1 : TI][MF16 ? 1 : TJ];
  f32x4_t acc16[MF16 ? 8 : 1][MF16 ? 4 : 1];      // 16x16 tiles: row tile i (16 rows), column tile j (16 columns)
  if constexpr (MF16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
  }

  // fragment read offsets: row (wave row base + i*32 + lr), chunk (2 s + lh) ^ ((row >> 1) & 7)
  // (the i-th 32-row tile of a wave is i * 4096 bytes further and has the same swizzle term,
  // so it is reached through the ds_read immediate offset)
  unsigned a_fo[4], b_fo[4];
  {
    const int ra = wm * (BM / WM) + lr, rb = wn * (BN / WN) + lr;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a_fo[s] = (unsigned)(ra * 128 + (((2 * s + lh) ^ ((ra >> 1) & 7)) << 4));
      b_fo[s] = (unsigned)(BRING + rb * 128 + (((2 * s + lh) ^ ((rb >> 1) & 7)) << 4));
    }
  }

  // 16x16x32: lane (r = lane & 15, g = lane >> 4) reads 16 bytes of row (tile base + r) at logical
  // chunk 4 ks + g of the 128-byte K-slice; tile i is i * 16 rows = i * 2048 bytes further (same
  // swizzle term), reached through the immediate offset
  unsigned a16[2], b16[2];
  {
    const int r = lane & 15, g = lane >> 4;
    const int ra = wm * (BM / WM) + r, rb = wn * (BN / WN) + r;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a16[ks] = (unsigned)(ra * 128 + (((4 * ks + g) ^ ((ra >> 1) & 7)) << 4));
      b16[ks] = (unsigned)(BRING + rb * 128 + (((4 * ks + g) ^ ((rb >> 1) & 7)) << 4));
    }
  }

  const unsigned lds_base = (unsigned)(unsigned long long)T2P_LDS_PTR(smem);
  // Issue order per iteration kt: B(kt + AB) first, then A(kt + AA).  vmcnt counts in order, so at
  // the top of iteration kt everything up to and including A(kt) and B(kt) has landed when at most
  // the loads issued after them are outstanding.
  constexpr int AA = NST - 1, AB = NSTB - 1;            // lead (K-tiles) of the A and B streams
  static_assert(AA <= 3 && (NSTB == NST || AA == 2), "ring depths: 2, 3 or 4 symmetric stages, or 3 + 2");
  issue(0, 2);
  if (AA > 1 && nk > 1) issue(1, AB > 1 ? 2 : 0);
  if (AA > 2 && nk > 2) issue(2, 2);
  unsigned ra_so = 0, rb_so = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // outstanding after A(kt), B(kt) in issue order: symmetric ring: the min(AA - 1, nk - 1 - kt) younger tiles (the tail of
    // the loop has fewer); asymmetric (AB == AA - 1): only A(kt + 1 .. kt + AA - 1)
    const int young = nk - 1 - kt;
    if (AA > 2 && young >= 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (A_INSTR + B_INSTR)) : "memory");
    } else if (AA > 1 && young >= 1) {
      if (NSTB == NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR + B_INSTR) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AA - 1) * A_INSTR) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
    const bool more_a = kt + AA < nk && !(dbg & 2), more_b = kt + AB < nk && !(dbg & 2);
    if (dbg & 4) { if (more_b) issue(kt + AB, 1); if (more_a) issue(kt + AA, 0); continue; }
    // Fragment reads go through inline asm: hipcc would otherwise put `s_waitcnt vmcnt(0)` in
    // front of every ds_read that may alias an in-flight LDS-DMA write and drain the ring.  The
    // reads of k-step s+1 are in flight while the MFMAs of step s run (lgkmcnt counts LDS ops in
    // order: <= TI + TJ outstanding means step s has landed).
    const unsigned sa_off = lds_base + ra_so, sb_off = lds_base + rb_so;      // ring stages of K-tile kt
    if constexpr (NST == 2) ra_so ^= (unsigned)ASTAGE;
    else ra_so = ra_so + ASTAGE == NST * ASTAGE ? 0u : ra_so + ASTAGE;
    if constexpr (NSTB == 2) rb_so ^= (unsigned)BSTAGE;
    else rb_so = rb_so + BSTAGE == NSTB * BSTAGE ? 0u : rb_so + BSTAGE;
    if constexpr (MF16) {
      // fragments: B of k-step 0 / 1 (4 column tiles each), A low / high half (4 row tiles each)
      u32x4_t B0[4], B1[4], AL[4], AH[4];
#define T2P_RD4(F, ADDR, I0)                                                                        \
  lds_read_b128_2k<I0>(F[0], ADDR); lds_read_b128_2k<I0 + 1>(F[1], ADDR);                            \
  lds_read_b128_2k<I0 + 2>(F[2], ADDR); lds_read_b128_2k<I0 + 3>(F[3], ADDR);
#define T2P_W8(N, X, Y)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(Y[0]), "+v"(Y[1]), \
               "+v"(Y[2]), "+v"(Y[3]) : "n"(N));
#define T2P_M16(A, B, I0)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)       \
      Mma16<TC>::run(B[j], A[i], acc16[I0 + i][j]);
      const unsigned aa0 = sa_off + a16[0], aa1 = sa_off + a16[1], bb0 = sb_off + b16[0], bb1 = sb_off + b16[1];
      // Waves w and w+4 share a SIMD.  The second half of the workgroup issues all of its DMA before
      // its matrix work, the first half in between: the two waves of a SIMD then run different
      // phases (one in the matrix pipe while the other issues DMA / waits on LDS), not in lockstep.
      const bool early = g_stagger_dbg(dbg) && wave >= (WM * WN) / 2;
      const bool late = g_stagger_dbg(dbg) && !(dbg & 256) && !early;
      if (early) {
        if (more_b) issue(kt + AB, 1);
        if (more_a) issue(kt + AA, 0);
      }
      T2P_RD4(B0, bb0, 0)
      T2P_RD4(AL, aa0, 0)
      T2P_RD4(AH, aa0, 4)
      T2P_W8(4, B0, AL)
      T2P_M16(AL, B0, 0)
      if (!early && !late && more_b) issue(kt + AB, 1);
      T2P_RD4(AL, aa1, 0)
      T2P_W8(4, B0, AH)
      T2P_M16(AH, B0, 4)
      if (!early && !late && more_a) issue(kt + AA, 0);
      if (late && more_b) issue(kt + AB, 1);
      T2P_RD4(B1, bb1, 0)
      T2P_RD4(AH, aa1, 4)
      T2P_W8(4, B1, AL)
      T2P_M16(AL, B1, 0)
      if (late && more_a) issue(kt + AA, 0);
      T2P_W8(0, B1, AH)
      T2P_M16(AH, B1, 4)
#undef T2P_RD4
#undef T2P_W8
#undef T2P_M16
      continue;
    }
    u32x4_t fa0[TI], fb0[TJ], fa1[TI], fb1[TJ];
#define T2P_RD(S, FA, FB)                                                                          \
  {                                                                                                \
    const unsigned aa = sa_off + a_fo[S];                                                          \
    const unsigned ba = sb_off + b_fo[S];                                                          \
    lds_read_b128<0>(FA[0], aa);                                                                   \
    lds_read_b128<0>(FB[0], ba);                                                                   \
    lds_read_b128<1>(FA[1], aa);                                                                   \
    lds_read_b128<1>(FB[1], ba);                                                                   \
    if constexpr (TI == 4) { lds_read_b128<2>(FA[2], aa); lds_read_b128<3>(FA[3], aa); }           \
    if constexpr (TJ == 4) { lds_read_b128<2>(FB[2], ba); lds_read_b128<3>(FB[3], ba); }           \
  }
#define T2P_WAIT(N, FA, FB)                                                                                           \
  if constexpr (TI == 2 && TJ == 2)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FB[0]), "+v"(FB[1]) : "n"(N));              \
  else if constexpr (TI == 4)                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(%6)"                                                                              \
                 : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FB[0]), "+v"(FB[1]) : "n"(N));            \
  else                                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(%6)"                                                                              \
                 : "+v"(FA[0]), "+v"(FA[1]), "+v"(FB[0]), "+v"(FB[1]), "+v"(FB[2]), "+v"(FB[3]) : "n"(N));
#define T2P_MMA(FA, FB)                                                                            \
  _Pragma("unroll") for (int i = 0; i < TI; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j)    \
      Mma<TC>::run(__builtin_bit_cast(uint4, FA[i]), __builtin_bit_cast(uint4, FB[j]), acc[i][j]);
    // The fragment reads start right after the barrier; the DMA of the next K-tile (address
    // VALU + buffer_load...lds) is issued between MFMA groups so that it overlaps the matrix
    // pipe instead of holding every wave of the workgroup in a VALU-only phase.
    constexpr int NRD = TI + TJ;
    const bool early = WM * WN == 8 && g_stagger_dbg(dbg) && wave >= 4;   // see the 16x16x32 loop above
    if (early) {
      if (more_b) issue(kt + AB, 1);
      if (more_a) issue(kt + AA, 0);
    }
    T2P_RD(0, fa0, fb0)
    T2P_RD(1, fa1, fb1)
    T2P_WAIT(NRD, fa0, fb0)
    T2P_MMA(fa0, fb0)
    if (!early && more_b) issue(kt + AB, 1);
    T2P_RD(2, fa0, fb0)
    T2P_WAIT(NRD, fa1, fb1)
    T2P_MMA(fa1, fb1)
    if (!early && more_a) issue(kt + AA, 0);
    T2P_RD(3, fa1, fb1)
    T2P_WAIT(NRD, fa0, fb0)
    T2P_MMA(fa0, fb0)
    T2P_WAIT(0, fa1, fb1)
    T2P_MMA(fa1, fb1)
#undef T2P_RD
#undef T2P_WAIT
#undef T2P_MMA
  }

  // REGE (16x16x32 kernels; chosen by the launcher with reg_epilogue_ok): epilogue straight from the registers
  if constexpr (MF16 && REGE) {
    // The epilogue reads its parameters from the kernel-argument segment through a pointer the compiler cannot see through (KernArgs):
    // it then loads each field where it is used (s_load from the constant cache) instead of keeping three dozen of them in scalar
    // registers across the K loop -- which it did by spilling them to VGPR lanes (25 v_writelane, 3700 v_readlane per kernel).
    // plain products (MODE 0: the attention / MLP projections, 8 K-tiles at C = 512, where the epilogue is a quarter of a tile) take the
    // specialised form when the launch is its case; the convolution modes keep one epilogue each (code size, registers)
    const KernArgs* pk = kernel_args();
    if constexpr (MODE == 0 && !CFRAG) {
      if (nsplit == 1 && !(dbg & 8192) && !pk->geglu && !pk->c_f32 && !pk->bias_m && !pk->up_phase)
        reg_epilogue<TC, BM, BN, WM, WN, false, true, KernArgs>(*pk, acc16, m0, n0, z0, z1, 1, 0, dbg);
      else reg_epilogue<TC, BM, BN, WM, WN, CFRAG, false, KernArgs>(*pk, acc16, m0, n0, z0, z1, nsplit, ks, dbg);
    } else {
      reg_epilogue<TC, BM, BN, WM, WN, CFRAG, false, KernArgs>(*pk, acc16, m0, n0, z0, z1, nsplit, ks, dbg);
    }
  }
  else dma_epilogue<TC, BM, BN, WM, WN, MF16, TI, TJ>(p, smem, acc, acc16, m0, n0, z0, z1, nsplit, ks, dbg);
}
template <typename TC, int BM, int BN, int WM, int WN, int MODE, int NST, int NSTB = NST, bool MF16 = false, bool REGE = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_kernel(const GemmParams p, const int tiles_m, const int tiles_n, const int dbg_arg) {
  gemm_dma_body<TC, BM, BN, WM, WN, MODE, NST, NSTB, MF16, REGE, false>(p, tiles_m, tiles_n, dbg_arg);
}
// the same kernel writing part of its output fragment-major (GemmParams::c_frag; register epilogue only)
template <typename TC, int BM, int BN, int WM, int WN, int MODE, int NST, int NSTB>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_cfrag_kernel(const GemmParams p, const int tiles_m, const int tiles_n, const int dbg_arg) {
  gemm_dma_body<TC, BM, BN, WM, WN, MODE, NST, NSTB, true, true, true>(p, tiles_m, tiles_n, dbg_arg);
}

// =================================================================================================
// v4 (round 4): 3x3 convolution at full resolution with the three HORIZONTAL taps of a window row served from one LDS stage.
//
// In the implicit GEMM above every input element lands in LDS nine times, once per tap: 96 KiB of activations per 64-channel chunk
// and 256 x 256 tile, 4.8 GB of L2 -> LDS traffic per launch on cfg2's dominant layer; that stream costs the loop a quarter of its
// matrix rate (1.57 -> 1.16 PFLOP/s, section 8 of DESIGN.md).  NHWC rows are pixel-major, so the rows a tile needs for tap
// (dy, dx) are the rows of tap (dy, 0) shifted by dx: ONE stage [rows m + dy W of the tile] serves dx = -1, 0, +1 when the
// fragment reads address row r + dx.  The two things that made the round-2 LDS-halo kernel lose are absent here:
//   * no halo and no padding: a tile is whole image rows (BM % W == 0), so the rows r - 1 of the tile's first pixel and r + 1 of its
//     last one are image-row edges, where the tap is zero anyway -- the stage is exactly the BM rows the kernel above stages;
//   * the image-row edges inside a tile (x = 0 under dx = -1, x = W - 1 under dx = +1) cost four v_cndmask per affected 16-pixel
//     tile and 32-deep step: with W a multiple of 64 only tiles 0 / 4 (left) and 3 / 7 (right) of a wave's 128 rows can be edges,
//     a wave-uniform test each; no per-lane tap bookkeeping.
// K order, weights layout, MFMA order and the register epilogue are those of gemm_dma_kernel<.., 1, 2, 2, true, true>: the results
// are bit-identical.  Activation DMA drops to one third (one stage per (chunk, dy) instead of per tap) and has two K-tiles to land
// instead of one; the weight stream is unchanged.  The 1x1 shortcut segment (X0 | X1) rides as single-tap groups at the end.
#ifdef T2P_ABLATION
// in-kernel timeline of the dx-shared kernel (measurement builds only): per tile, s_memtime of wave 0 at entry, before the K loop, after
// the first K-tile's data landed, after the K loop, after the epilogue issued its stores and after they drained
constexpr int DXS_STAMP_TILES = 4096, DXS_STAMPS = 8;
static __device__ unsigned long long g_dxs_stamps[DXS_STAMP_TILES * DXS_STAMPS];      // one per translation unit; the f16 one is read
#define T2P_STAMP(K) do { if (blockIdx.x < DXS_STAMP_TILES && threadIdx.x == 0) g_dxs_stamps[blockIdx.x * DXS_STAMPS + (K)] = __builtin_amdgcn_s_memtime(); } while (0)
#define T2P_STAMP_RT(K) do { if (blockIdx.x < DXS_STAMP_TILES && threadIdx.x == 0) g_dxs_stamps[blockIdx.x * DXS_STAMPS + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 2
int dxs_stamps_read(unsigned long long* out, int n) {
  if (n > DXS_STAMP_TILES * DXS_STAMPS) n = DXS_STAMP_TILES * DXS_STAMPS;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dxs_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? n : -1;
}
#endif
#else
#define T2P_STAMP(K) do { } while (0)
#define T2P_STAMP_RT(K) do { } while (0)
#endif
template <typename TC, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_dxs_body(const GemmParams& p, const int tiles_m, const int tiles_n, const int dbg_arg) {
  const int dbg = dbg_arg & (128 | 256);
  constexpr int BK = 64, NW = WM * WN;
  static_assert(NW == 8 && BM / WM == 128 && BN / WN == 64, "128 x 64 wave tiles, 8 wavefronts");
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  // LDS: the weight ring FIRST, then the activation ring: row -1 of an activation stage (read by lane r = 0 of a wave's first tile under
  // dx = -1, always an image-row edge, always zeroed) is then ordinary LDS memory, not an address below the allocation
  constexpr int ASTAGE = BM * 128, BSTAGE = BN * 128, ARING = 2 * BSTAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  T2P_STAMP(0);
  T2P_STAMP_RT(6);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int ntiles = tiles_m * tiles_n;
  int tile = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int Ctot = p.C0 + p.C1, nch = Ctot / BK, nchx = (p.CX0 + p.CX1) / BK;
  const int ngroups = nch * 3 + nchx;                 // A stages: (chunk, dy) x 3 K-tiles each, then the shortcut chunks x 1 K-tile
  const int HW = p.H * p.W, W = p.W;

  const TC* A0p = (const TC*)p.A0;
  const TC* Bp = (const TC*)p.Bw;
  const long a_rows = p.M;
  const int a0_bytes = (int)(((a_rows - 1) * p.lda0 + p.C0) * 2);
  const int a1_bytes = p.A1 ? (int)(((a_rows - 1) * p.lda1 + p.C1) * 2) : 0;
  const __amdgpu_buffer_rsrc_t rB = make_rsrc(Bp, (int)((((long)p.N - 1) * p.ldb + 9L * Ctot + nchx * BK) * 2));
  const unsigned lda0_2 = (unsigned)(p.lda0 * 2), lda1_2 = (unsigned)(p.lda1 * 2);
  const unsigned ldx0_2 = (unsigned)(p.ldx0 * 2), ldx1_2 = (unsigned)(p.ldx1 * 2);
  const void* const X0p = p.X0; const void* const X1p = p.X1; const void* const A1p = p.A1;
  const int pC0 = p.C0, pCX0 = p.CX0;
  const int x0_bytes = nchx ? (int)((((long)p.M - 1) * p.ldx0 + p.CX0) * 2) : 0;
  const int x1_bytes = (nchx && p.X1) ? (int)((((long)p.M - 1) * p.ldx1 + p.CX1) * 2) : 0;

  // DMA geometry.  The row of instruction j is a_m0 + 8 j (per lane).  The 8 rows of an instruction lie in one image row (W % 8 == 0) and
  // the tile in one sample (HW % BM == 0), so the validity of its three window rows is WAVE-UNIFORM: a_vmp, a scalar, packs it for all
  // instructions (bit 3 j + dy + 1: 0 <= y + dy < H; every row is < M since M % BM == 0) -- two integer divisions per tile, on the
  // scalar unit, instead of two per instruction and lane
  const int prow = lane >> 3, ppos = lane & 7;
  // activation swizzle: 16-byte chunk c of stage row r sits at chunk c ^ (r & 6).  ds_read_b128 is served in the lane groups
  // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): rows {0-3, 12-15} at chunk c together with rows {4-11} at chunk c ^ 1.  The
  // usual (r >> 1) & 7 term is conflict-free for that only while lane parity = row parity; under the +-1 row shift of the dx taps it
  // is 2-way on a quarter of the slots (SQ_LDS_BANK_CONFLICT 19.9 M per launch at 256x256).  r & 6 is conflict-free at all three
  // shifts (and has period 8: one source offset for every instruction).
  const unsigned a_chk = (unsigned)((ppos ^ (prow & 6)) * 16);
  const unsigned a_m0 = (unsigned)(m0 + wave * A_INSTR * 8 + prow);
  unsigned a_vmp;
  static_assert(A_INSTR * 3 <= 32, "validity bits of a wave's A instructions fit one register");
  {
    const int lgW = 31 - __builtin_clz(W);              // W is a power of two (dxs_eligible)
    const int b = m0 / HW, y0 = (m0 - b * HW) >> lgW, Hm1 = p.H - 1;
    unsigned v = 0;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int y = y0 + (((wave * A_INSTR + j) * 8) >> lgW);
      v |= ((y > 0 ? 1u : 0u) | 2u | (y < Hm1 ? 4u : 0u)) << (3 * j);
    }
    a_vmp = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  }
  unsigned b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + prow;
    const int n = n0 + (r & ~63) + hperm(r & 63);
    b_off[j] = n < p.N ? (unsigned)((long)n * p.ldb * 2) + (unsigned)((ppos ^ ((r >> 1) & 7)) * 16) : DMA_OOB;
  }

  // ---- A stream: one stage per group --------------------------------------------------------------------------------------
  int ga = 0;                                           // next group to issue
  unsigned ia_so = 0;
  auto issue_a = [&]() __attribute__((always_inline)) {
    const bool extra = ga >= nch * 3;
    const int chunk = extra ? ga - nch * 3 : ga / 3;
    const int dyi = extra ? 1 : ga - chunk * 3;         // window row 0 .. 2 (the shortcut segment reads the output pixel itself)
    const int c0 = chunk * BK;
    int cfirst = pC0;
    if (extra) cfirst = pCX0;
    const bool second = c0 >= cfirst;
    const int csrc = second ? c0 - cfirst : c0;
    unsigned ld2 = lda0_2;
    const void* abase = (const void*)A0p;
    int abytes = a0_bytes;
    if (extra) {
      if (second) { ld2 = ldx1_2; abase = X1p; abytes = x1_bytes; } else { ld2 = ldx0_2; abase = X0p; abytes = x0_bytes; }
    } else if (second) {
      ld2 = lda1_2; abase = A1p; abytes = a1_bytes;
    }
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(abase, abytes);
    const unsigned delta = (unsigned)((dyi - 1) * W * (int)ld2 + csrc * 2);
    const unsigned base = __umul24(a_m0, ld2) + delta;
    const unsigned vb0 = a_chk + base;
    const unsigned step = 8u * ld2;                     // 8 rows further per instruction
    unsigned char* sta = smem + ARING + ia_so;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const unsigned voff = vb0 + (unsigned)j * step;
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)a_vmp, (unsigned)(3 * j + dyi), 1u);
      unsigned char* dst = sta + (wave * A_INSTR + j) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, T2P_LDS_PTR(dst), 16, (voff & m) | (DMA_OOB & ~m), 0, 0, 0);
    }
    ia_so ^= (unsigned)ASTAGE;
    ++ga;
  };
  // ---- B stream: one stage per K-tile, K order (chunk, tap), then the shortcut columns ---------------------------------------
  int ib_chunk = 0, ib_tap = 0;
  unsigned ib_so = 0, b_kb = 0;
  const unsigned Ctot2 = (unsigned)(Ctot * 2);
  auto issue_b = [&]() __attribute__((always_inline)) {
    unsigned char* stb = smem + ib_so;
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      unsigned char* dst = stb + (wave * B_INSTR + j) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, T2P_LDS_PTR(dst), 16, b_off[j] + b_kb, 0, 0, 0);
    }
    ib_so ^= (unsigned)BSTAGE;
    ++ib_tap; b_kb += Ctot2;
    if (ib_tap == 9) {
      asm volatile("");
      ++ib_chunk;
      if (ib_chunk >= nch) { b_kb = (unsigned)((9L * Ctot + (ib_chunk - nch) * BK) * 2); ib_tap = 8; }
      else { ib_tap = 0; b_kb += (unsigned)(BK * 2) - 9u * Ctot2; }
    }
  };

  f32x4_t acc16[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: lane (r = lane & 15, g = lane >> 4) reads 16 bytes of row (wave rows + r + dx) at logical chunk 4 ks + g;
  // 16-row tile i is i * 2048 bytes further with the same swizzle term (immediate offset).  Rows -1 / BM of a stage are read only by
  // lanes the edge masks zero (they lie in the neighbouring ring stage / past the allocation, where LDS reads return zero).
  // (the second 32-deep step reads chunk 4 + g: the same offset with bit 6 flipped, (4 | g) ^ x = (g ^ x) ^ 4 -- one XOR per K-tile
  // instead of a second register per offset: four registers fewer through the K loop)
  unsigned a16[3], b16;
  {
    const int r = lane & 15, g = lane >> 4;
    const int rb = wn * (BN / WN) + r;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int ra = wm * (BM / WM) + r + d - 1;         // (-1 for the first lane of the first wave row under dx = -1)
      a16[d] = (unsigned)(ARING + ra * 128 + ((g ^ (ra & 6)) << 4));
    }
    b16 = (unsigned)(rb * 128 + ((g ^ ((rb >> 1) & 7)) << 4));
  }
  // image-row edges inside this wave's 128 rows (W % 64 == 0: only these four tiles can hold one)
  const int rowb = m0 + wm * (BM / WM);
  const bool eL0 = rowb % W == 0, eL4 = (rowb + 64) % W == 0, eR3 = eL4, eR7 = (rowb + 128) % W == 0;
  const bool is_r0 = (lane & 15) == 0, is_r15 = (lane & 15) == 15;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  const unsigned lds_base = (unsigned)(unsigned long long)T2P_LDS_PTR(smem);
  const bool early = g_stagger_dbg(dbg) && wave >= NW / 2;
  const bool late = g_stagger_dbg(dbg) && !(dbg & 256) && !early;
  unsigned ra_so = 0, rb_so = 0;
  const int nk = nch * 9 + nchx;
  int kt = 0;                                           // K-tile being multiplied (B stream position kt + 1 is the next issue)
  bool a_requested = false;                             // the first K-tile of the running group requested the next group's stage

  // one K-tile: DX = 0 / 1 / 2 is the horizontal tap dx = DX - 1; `first`: first K-tile of its group (the next group's stage is
  // requested here, behind the weight tile, and has until the group's last K-tile to land: WAIT_A says whether it may still be in flight)
#define T2P_RD4(F, ADDR, I0)                                                                        \
  lds_read_b128_2k<I0>(F[0], ADDR); lds_read_b128_2k<I0 + 1>(F[1], ADDR);                            \
  lds_read_b128_2k<I0 + 2>(F[2], ADDR); lds_read_b128_2k<I0 + 3>(F[3], ADDR);
#define T2P_W8(N, X, Y)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(Y[0]), "+v"(Y[1]), \
               "+v"(Y[2]), "+v"(Y[3]) : "n"(N));
#define T2P_M16(A, B, I0)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)       \
      Mma16<TC>::run(B[j], A[i], acc16[I0 + i][j]);
  auto ktile = [&](auto DXC, auto FIRSTC, auto AFLYC) __attribute__((always_inline)) {
    constexpr int DX = decltype(DXC)::value;
    constexpr bool FIRST = decltype(FIRSTC)::value;     // issues the next group's A stage
    constexpr bool AFLY = decltype(AFLYC)::value;       // the A stage requested one K-tile ago may still be in flight at the top
    // (the weight tile of this K-tile was requested BEFORE the activation stage that may still be in flight: loads land in order, so
    // "at most the stage's A_INSTR loads outstanding" means the weight tile is there -- but only if that stage was requested at all)
    if (AFLY && a_requested) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_INSTR) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more_b = kt + 1 < nk, more_a = FIRST && ga < ngroups;
    if constexpr (FIRST) a_requested = more_a;
    const unsigned sa_off = lds_base + ra_so, sb_off = lds_base + rb_so;
    rb_so ^= (unsigned)BSTAGE;
    u32x4_t B0[4], B1[4], AL[4], AH[4];
    const unsigned aa0 = sa_off + a16[DX], aa1 = sa_off + (a16[DX] ^ 64u), bb0 = sb_off + b16, bb1 = sb_off + (b16 ^ 64u);
    if (early) {
      if (more_b) issue_b();
      if (more_a) issue_a();
    }
    T2P_RD4(B0, bb0, 0)
    T2P_RD4(AL, aa0, 0)
    T2P_RD4(AH, aa0, 4)
    T2P_W8(4, B0, AL)
    if constexpr (DX == 0) { if (eL0 && is_r0) AL[0] = zero4; }
    if constexpr (DX == 2) { if (eR3 && is_r15) AL[3] = zero4; }
    T2P_M16(AL, B0, 0)
    if (!early && !late && more_b) issue_b();
    T2P_RD4(AL, aa1, 0)
    T2P_W8(4, B0, AH)
    if constexpr (DX == 0) { if (eL4 && is_r0) AH[0] = zero4; }
    if constexpr (DX == 2) { if (eR7 && is_r15) AH[3] = zero4; }
    T2P_M16(AH, B0, 4)
    if (!early && !late && more_a) issue_a();
    if (late && more_b) issue_b();
    T2P_RD4(B1, bb1, 0)
    T2P_RD4(AH, aa1, 4)
    T2P_W8(4, B1, AL)
    if constexpr (DX == 0) { if (eL0 && is_r0) AL[0] = zero4; }
    if constexpr (DX == 2) { if (eR3 && is_r15) AL[3] = zero4; }
    T2P_M16(AL, B1, 0)
    if (late && more_a) issue_a();
    T2P_W8(0, B1, AH)
    if constexpr (DX == 0) { if (eL4 && is_r0) AH[0] = zero4; }
    if constexpr (DX == 2) { if (eR7 && is_r15) AH[3] = zero4; }
    T2P_M16(AH, B1, 4)
    ++kt;
  };
#undef T2P_RD4
#undef T2P_W8
#undef T2P_M16
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using Tt = std::true_type; using Ff = std::false_type;

  T2P_STAMP(1);
  issue_b();                                            // weight tile of K-tile 0
  issue_a();                                            // stage of group 0
  for (int g = 0; g < nch * 3; ++g) {
    ktile(I0{}, Tt{}, Ff{});                            // dx = -1: requests group g + 1
#ifdef T2P_ABLATION
    if (g == 0) T2P_STAMP(2);
#endif
    ktile(I1{}, Ff{}, Tt{});                            // dx =  0: only the weight tile has to be there
    ktile(I2{}, Ff{}, Ff{});                            // dx = +1
    ra_so ^= (unsigned)ASTAGE;
  }
  for (int e = 0; e < nchx; ++e) {                      // shortcut segment: one K-tile per stage, read at the output pixel
    ktile(I1{}, Tt{}, Ff{});
    ra_so ^= (unsigned)ASTAGE;
  }
  T2P_STAMP(3);
#ifdef T2P_ABLATION
  const int edbg = dbg_arg & 1024;                     // 1024: the stores go nowhere (timeline experiments)
#else
  const int edbg = 0;
#endif
  {
    const KernArgs* pk = kernel_args();                 // (see gemm_dma_body: the epilogue's parameters are loaded where they are used)
    if (!(dbg_arg & 8192) && !pk->c_f32 && !pk->bias_m) reg_epilogue<TC, BM, BN, WM, WN, false, true, KernArgs>(*pk, acc16, m0, n0, 0, 0, 1, 0, edbg);
    else reg_epilogue<TC, BM, BN, WM, WN, false, false, KernArgs>(*pk, acc16, m0, n0, 0, 0, 1, 0, edbg);
  }
#ifdef T2P_ABLATION
  T2P_STAMP(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  T2P_STAMP(5);
  T2P_STAMP_RT(7);
#endif
}
template <typename TC, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dxs_kernel(const GemmParams p, const int tiles_m, const int tiles_n, const int dbg_arg) {
  gemm_dxs_body<TC, BM, BN, WM, WN>(p, tiles_m, tiles_n, dbg_arg);
}


#endif  // T2P_PART_DMA

#if T2P_PART_HOST
// ---- optional per-launch timing (bench.py roofline leg): HIP events on the launch stream -----------
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
// the 3x3-convolution instantiation with the largest total time of the last profiled region
static std::string g_dom_name;
static double g_dom[4] = {0, 0, 0, 0};

// fused-attention launches of the profiled region (attention.hip reports them through prof_attention)
static std::vector<ProfRec> g_prof_attn;
static double g_attn[3] = {0, 0, 0};
bool prof_on() { return g_prof_on; }
void prof_attention(hipEvent_t a, hipEvent_t b, double flops) { g_prof_attn.push_back(ProfRec{a, b, flops, 3, nullptr, 0.0}); }
int profile_attention(double out[3]) {
  out[0] = g_attn[0]; out[1] = g_attn[1]; out[2] = g_attn[2];
  return T2P_OK;
}

static std::map<std::string, std::array<double, 3>> g_shapes;    // "kind,M,N,K,taps,z" -> {launches, ms, flops} of the last region
// the launches of the region closed by the last profile_end, one line per operand shape:
// kind,M,N,K,taps (negative: gathered from the half-resolution map),batch,launches,ms,flops
int profile_shapes(char* buf, int len) {
  std::string t = "kind,M,N,K,taps,batch,launches,ms,flops\n";
  for (auto& kv : g_shapes) {
    char line[256];
    std::snprintf(line, sizeof line, "%s,%.0f,%.4f,%.6g\n", kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2]);
    t += line;
  }
  if ((int)t.size() + 1 > len) { set_last_error("profile_shapes: buffer too small"); return T2P_ERR_INVALID; }
  std::memcpy(buf, t.c_str(), t.size() + 1);
  return T2P_OK;
}
void profile_begin() {
  for (ProfRec& r : g_prof_attn) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof_attn.clear();
  for (ProfRec& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_on = true;
}
// out[kind][0..2] = {milliseconds, flops, launches}; kind 0 = 3x3 convolutions on the LDS-DMA
// kernel (the dominant kernel), 1 = other GEMMs, 2 = 3x3 convolutions on the register-staged kernel
// (pre_conv in fp32, the 5-channel head, tiny maps)
int profile_end(double out[3][3]) {
  g_prof_on = false;
  for (int k = 0; k < 3; ++k) out[k][0] = out[k][1] = out[k][2] = 0;
  std::map<std::string, std::array<double, 4>> per;
  g_shapes.clear();
  for (ProfRec& r : g_prof) {
    T2P_HIP_CHECK(hipEventSynchronize(r.b));
    float ms = 0.f;
    T2P_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
    out[r.kind][0] += ms; out[r.kind][1] += r.flops; out[r.kind][2] += 1;
    {
      char key[128];
      if (r.K >= 1000000) std::snprintf(key, sizeof key, "%d,%d,%d,%d+%d,%d,%d", r.kind, r.M, r.N, r.K / 1000000, r.K % 1000000, r.taps, r.z);
      else std::snprintf(key, sizeof key, "%d,%d,%d,%d,%d,%d", r.kind, r.M, r.N, r.K, r.taps, r.z);
      auto& e = g_shapes[key];
      e[0] += 1; e[1] += ms; e[2] += r.flops;
    }
    if (r.kind == 0 && r.name) {
      auto& e = per[r.name];
      e[0] += ms; e[1] += r.flops; e[2] += 1; e[3] += r.bytes;
    }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  g_attn[0] = g_attn[1] = g_attn[2] = 0;
  for (ProfRec& r : g_prof_attn) {
    T2P_HIP_CHECK(hipEventSynchronize(r.b));
    float ms = 0.f;
    T2P_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
    g_attn[0] += ms; g_attn[1] += r.flops; g_attn[2] += 1;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_prof_attn.clear();
  g_dom_name.clear();
  g_dom[0] = g_dom[1] = g_dom[2] = g_dom[3] = 0;
  for (auto& kv : per)
    if (kv.second[0] > g_dom[0]) { g_dom_name = kv.first; for (int i = 0; i < 4; ++i) g_dom[i] = kv.second[i]; }
  return T2P_OK;
}

// {milliseconds, flops, launches, algorithmic bytes} and the kernel name (as rocprofv3 prints it, without the
// argument list) of the dominant 3x3-convolution instantiation of the region closed by the last profile_end.
// Algorithmic bytes of a launch: every input element, weight, residual element once + the output once.
int profile_dominant(double out[4], const char** name) {
  out[0] = g_dom[0]; out[1] = g_dom[1]; out[2] = g_dom[2]; out[3] = g_dom[3];
  *name = g_dom_name.c_str();
  return T2P_OK;
}

template <typename TC, bool AF32, int BM, int BN>
static int launch_t(const GemmParams& p, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * ROWB;
  auto kern = gemm_kernel<TC, AF32, BM, BN>;
  T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nz0 * p.nz1);
  ProfRec rec;
  if (g_prof_on) {
    T2P_HIP_CHECK(hipEventCreate(&rec.a));
    T2P_HIP_CHECK(hipEventCreate(&rec.b));
    prof_shape(rec, p);
    rec.flops = 2.0 * p.M * p.N * (double)p.taps * (p.C0 + p.C1) * p.nz0 * p.nz1;
    rec.kind = p.taps == 9 ? 2 : 1;
    rec.name = nullptr;
    rec.bytes = 0;
    T2P_HIP_CHECK(hipEventRecord(rec.a, stream));
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
  if (g_prof_on) {
    T2P_HIP_CHECK(hipEventRecord(rec.b, stream));
    g_prof.push_back(rec);
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// 128 x 64 wave-tile geometries: 2 = 2 + 2 ring stages with v_mfma_16x16x32 (default: +5 % over the
// 32x32x16 shape, it sustains a higher clock); 1 = 2 + 2 stages, 32x32x16; 0 = 3 A + 2 B stages
// (160 KiB; measured equal to 1)
int g_dma_ring = 2;
void set_gemm_ring(int v) { g_dma_ring = v; }
bool g_dxs = true;            // plan switch 47: full-resolution 3x3 convolutions on the dx-shared-stage kernel (gemm_dxs_kernel) where eligible
void set_gemm_dxs(bool on) { g_dxs = on; }
static int g_dma_geom = 0;   // 0 auto, 1 force 256x128x3, 2 force 128x128x2, 3 force 256x256x2, 4 force 512x128x2
static bool g_splitk = true;
static int g_force_nsplit = 0;   // development: > 0 forces that split-K factor wherever a workspace is attached
void set_gemm_force_nsplit(int v) { g_force_nsplit = v; }
static bool g_use_dma = true;
bool g_deep_ring = true;    // small launches of the 128 x 128 geometry on a 4-stage ring (plan switch 22)
void set_gemm_deep_ring(bool on) { g_deep_ring = on; }
int num_cus() {
  static int n[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!n[dev]) {
    hipDeviceProp_t prop;
    n[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  return n[dev];
}
bool g_up4 = true;          // up-sampling 3x3 convolutions as four 2x2 phase convolutions where the caller supplies Bw4 (plan switch 21)
void set_gemm_up4(bool on) { g_up4 = on; }
int g_dbg = 0;   // timing-only ablations (results are wrong when non-zero): 1 no epilogue, 2 no DMA in loop, 4 no reads/MFMA
void set_gemm_dma(bool on) { g_use_dma = on; }
void set_gemm_debug(int v) { g_dbg = v; }

static bool dma_eligible(const GemmParams& p) {
  if (!g_use_dma || p.dtype == DT_F32 || p.a_f32) return false;
  if (p.C0 % 64 != 0 || p.C1 % 64 != 0) return false;   // whole 64-channel K-tiles only (else v1)
  if (p.M < 128 || p.N < 64) return false;              // small problems: v1's 64x64 tiles fill the chip better
  if (p.a_up && p.taps != 9) return false;
  if (p.c_nchw) return false;
  if (((long)p.M + 256) * std::max(p.lda0, p.lda1) * 2 >= (1L << 31)) return false;   // 32-bit offset arithmetic
  if ((long)p.M + 512 >= (1L << 24) || std::max(p.lda0, p.lda1) * 2 >= (1L << 24)) return false;   // row * stride in one 24-bit multiply
  const long a_rows = (p.taps == 9 || p.a_up) ? (long)(p.M / (p.H * p.W)) * (p.a_up ? (p.H / 2) * (p.W / 2) : p.H * p.W) : p.M;
  const long lim = (1L << 31) - 64;
  if (a_rows * p.lda0 * 2 >= lim || (p.A1 && a_rows * p.lda1 * 2 >= lim)) return false;
  if ((long)p.N * p.ldb * 2 >= lim) return false;
  return true;
}

// split-K second pass: sum the partial tiles in a fixed order (bitwise reproducible) and apply the epilogue
template <typename TC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, const int nsplit) {
  const long total = (long)p.M * p.N;
  const int HW = p.H * p.W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int row = (int)(idx / p.N), col = (int)(idx - (long)row * p.N);
    float val = 0.f;
    for (int k = 0; k < nsplit; ++k) val += ((const float*)p.ws)[(long)k * total + idx];
    const int bidx = row / p.rows_per_batch;
    long rrow = row;
    if (p.r_up) {
      const int rem = row - bidx * HW;
      const int y = rem / p.W, x = rem - y * p.W;
      rrow = ((long)bidx * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
    }
    if (p.bias_m) val += p.bias_m[row];
    if (p.bias_n) val += p.bias_n[col];
    if (p.bias_bn) val += p.bias_bn[(long)bidx * p.ld_bn + col];
    if (p.R) val += res_load1<TC>(p.R, rrow * p.ldr + col, p.r_lowp);
    val *= p.alpha;
    if (p.c_f32) ((float*)p.C)[(long)row * p.ldc + col] = val;
    else ((TC*)p.C)[(long)row * p.ldc + col] = from_f32<TC>(val);
  }
}

// Vectorised second pass: a block of 16 * RL threads owns RL rows x 64 columns (RL = 64: the statistics
// chunk, 1024 threads; RL = 16: 256 threads); thread (rl = t >> 4, cg = t & 15) sums the partials of 4
// columns of its row in split order (bitwise reproducible), applies the epilogue and, with RL == 64, the
// GroupNorm column statistics of the final fp32 values, folded over the row lanes in a fixed order
// through LDS.  One row per thread: these launches are small, so parallelism beats work per thread.
template <typename TC, int RL>
__global__ __launch_bounds__(16 * RL) void splitk_reduce_vec_kernel(const GemmParams p, const int nsplit) {
  __shared__ float red[RL == 64 ? 64 : 1][16][8];
  const int t = threadIdx.x, cg = t & 15, rl = t >> 4;
  const int col = blockIdx.x * 64 + cg * 4;
  const int row0 = blockIdx.y * RL;
  const long total = (long)p.M * p.N;
  const int HW = p.H * p.W;
  const bool col_ok = col < p.N;
  const int row = row0 + rl;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < p.M && col_ok) {
    float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias_n) bn = *(const float4*)(p.bias_n + col);
    const float* src = (const float*)p.ws + (long)row * p.N + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);       // 0 + p0 is exact: same sums as starting from p0
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {                 // 8 partial loads in flight, added in split order
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const float4*)(src + (long)(k + j) * total);
#pragma unroll
      for (int j = 0; j < 8; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    for (; k + 2 <= nsplit; k += 2) {
      const float4 b0 = *(const float4*)(src + (long)k * total), b1 = *(const float4*)(src + (long)(k + 1) * total);
      a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
      a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
    }
    if (k < nsplit) {
      const float4 b = *(const float4*)(src + (long)k * total);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int bidx = row / p.rows_per_batch;
    long rrow = row;
    if (p.r_up) {
      const int rem = row - bidx * HW;
      const int y = rem / p.W, x = rem - y * p.W;
      rrow = ((long)bidx * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1);
    }
    const float bm = p.bias_m ? p.bias_m[row] : 0.f;
    a.x += bm + bn.x; a.y += bm + bn.y; a.z += bm + bn.z; a.w += bm + bn.w;
    if (p.bias_bn) {
      const float4 b = *(const float4*)(p.bias_bn + (long)bidx * p.ld_bn + col);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (p.R) {
      const float4 b = res_load4<TC>(p.R, rrow * p.ldr + col, p.r_lowp);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
    cs[0] = a.x; cs[1] = a.y; cs[2] = a.z; cs[3] = a.w;
    cq[0] = a.x * a.x; cq[1] = a.y * a.y; cq[2] = a.z * a.z; cq[3] = a.w * a.w;
    if (p.c_f32) *(float4*)((float*)p.C + (long)row * p.ldc + col) = a;
    else *(u32x2_t*)((TC*)p.C + (long)row * p.ldc + col) = (u32x2_t){pack2<TC>(a.x, a.y), pack2<TC>(a.z, a.w)};
  }
  if constexpr (RL == 64) {
    if (p.col_stats) {                                 // uniform
#pragma unroll
      for (int k = 0; k < 4; ++k) { red[rl][cg][2 * k] = cs[k]; red[rl][cg][2 * k + 1] = cq[k]; }
      __syncthreads();
      if (t < 128) {                                   // (cg, k): 16 column groups x 8 values, rows folded in order
        const int c2 = t >> 3, k2 = t & 7;
        float o = red[0][c2][k2];
        for (int r = 1; r < 64; ++r) o += red[r][c2][k2];
        const int ccol = blockIdx.x * 64 + c2 * 4;
        if (ccol < p.N && row0 < p.M) p.col_stats[((long)(row0 >> 6) * p.N + ccol) * 2 + k2] = o;
      }
    }
  }
}


// Split-K second pass that also applies the GroupNorm (+SiLU) which follows the product (GemmParams::gn_*): a block of 1024
// threads owns one sample's HW <= 64 * RPT <= 256 rows x 64 columns (whole groups: 64 % (N / groups) == 0), thread (rl = t >> 4,
// cg = t & 15) keeps the final fp32 values of rows rl + 64 j, columns 4 cg .. 4 cg + 3 in registers: partials summed in split
// order, epilogue as in splitk_reduce_vec_kernel, column sums folded over the row lanes in a fixed order through LDS, group
// statistics in double, then y = act(v * rstd * gamma + (beta - mean * rstd * gamma)) -> gn_out.  One launch instead of the
// second pass + a GroupNorm launch (two or three at the 16x16 .. 4x4 levels, where every launch costs its latency).
template <typename TC, int RPT>
__global__ __launch_bounds__(1024) void splitk_reduce_gn_kernel(const GemmParams p, const int nsplit) {
  __shared__ float red[64][16][8];
  __shared__ float colsum[64][2];
  __shared__ float scsh[64][2];
  const int t = threadIdx.x, cg = t & 15, rl = t >> 4;
  const int col = blockIdx.x * 64 + cg * 4;
  const int HW = p.rows_per_batch;
  const int b = blockIdx.y;
  const long total = (long)p.M * p.N;
  const int Wm = p.W;
  f32x4_t v[RPT];
  float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias_n) bn = *(const float4*)(p.bias_n + col);
  if (p.bias_bn) {
    const float4 u = *(const float4*)(p.bias_bn + (long)b * p.ld_bn + col);
    bn.x += u.x; bn.y += u.y; bn.z += u.z; bn.w += u.w;
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = rl + 64 * j;
    v[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (r < HW) {
      const int row = b * HW + r;
      const float* src = (const float*)p.ws + (long)row * p.N + col;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = 0;
      for (; k + 4 <= nsplit; k += 4) {               // 4 partial loads in flight, added in split order
        float4 w4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w4[q] = *(const float4*)(src + (long)(k + q) * total);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a.x += w4[q].x; a.y += w4[q].y; a.z += w4[q].z; a.w += w4[q].w; }
      }
      for (; k < nsplit; ++k) {
        const float4 w1 = *(const float4*)(src + (long)k * total);
        a.x += w1.x; a.y += w1.y; a.z += w1.z; a.w += w1.w;
      }
      a.x += bn.x; a.y += bn.y; a.z += bn.z; a.w += bn.w;
      if (p.R) {
        long rrow = row;
        if (p.r_up) {
          const int y = r / Wm, x = r - y * Wm;
          rrow = ((long)b * (p.H >> 1) + (y >> 1)) * (Wm >> 1) + (x >> 1);
        }
        const float4 u = res_load4<TC>(p.R, rrow * p.ldr + col, p.r_lowp);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      }
      a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
      v[j] = f32x4_t{a.x, a.y, a.z, a.w};
      cs[0] += a.x; cs[1] += a.y; cs[2] += a.z; cs[3] += a.w;
      cq[0] += a.x * a.x; cq[1] += a.y * a.y; cq[2] += a.z * a.z; cq[3] += a.w * a.w;
      if (p.C) {
        if (p.c_f32) *(float4*)((float*)p.C + (long)row * p.ldc + col) = a;
        else *(u32x2_t*)((TC*)p.C + (long)row * p.ldc + col) = (u32x2_t){pack2<TC>(a.x, a.y), pack2<TC>(a.z, a.w)};
      }
    }
    if (p.col_stats && 64 * j < HW) {
      // (uniform) the per-64-row-chunk column sums a later GroupNorm over this tensor reads (skip connections): chunk j of this
      // sample is row j of every thread -- folded per chunk, in the layout of the GEMM epilogues
      if (j) __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) { red[rl][cg][2 * k] = v[j][k]; red[rl][cg][2 * k + 1] = v[j][k] * v[j][k]; }
      __syncthreads();
      if (t < 128) {
        const int c2 = t >> 3, k2 = t & 7;
        float o = red[0][c2][k2];
        for (int r2 = 1; r2 < 64; ++r2) o += red[r2][c2][k2];
        p.col_stats[((long)(b * (HW >> 6) + j) * p.N + blockIdx.x * 64 + c2 * 4) * 2 + k2] = o;
      }
    }
  }
  if (p.col_stats) __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[rl][cg][2 * k] = cs[k]; red[rl][cg][2 * k + 1] = cq[k]; }
  __syncthreads();
  if (t < 128) {                                      // (column group, value): row lanes folded in a fixed order
    const int c2 = t >> 3, k2 = t & 7;
    float o = red[0][c2][k2];
    for (int r = 1; r < 64; ++r) o += red[r][c2][k2];
    colsum[c2 * 4 + (k2 >> 1)][k2 & 1] = o;
  }
  __syncthreads();
  if (t < 64) {
    const int cpg = p.N / p.gn_groups;                 // channels per group: divides 64
    const int g0 = (t / cpg) * cpg;
    double s = 0, q = 0;
    for (int k = 0; k < cpg; ++k) { s += (double)colsum[g0 + k][0]; q += (double)colsum[g0 + k][1]; }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
    const int c = blockIdx.x * 64 + t;
    const float sc = rstd * p.gn_gamma[c];
    scsh[t][0] = sc;
    scsh[t][1] = p.gn_beta[c] - (float)mean * sc;
  }
  __syncthreads();
  float sc[4], sh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { sc[k] = scsh[cg * 4 + k][0]; sh[k] = scsh[cg * 4 + k][1]; }
  TC* out = (TC*)p.gn_out;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int r = rl + 64 * j;
    if (r < HW) {
      float y0 = v[j][0] * sc[0] + sh[0], y1 = v[j][1] * sc[1] + sh[1], y2 = v[j][2] * sc[2] + sh[2], y3 = v[j][3] * sc[3] + sh[3];
      if (p.gn_silu) {
        y0 = y0 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y0 * -1.44269504088896341f));
        y1 = y1 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y1 * -1.44269504088896341f));
        y2 = y2 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y2 * -1.44269504088896341f));
        y3 = y3 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y3 * -1.44269504088896341f));
      }
      *(u32x2_t*)(out + ((long)b * HW + r) * p.N + col) = (u32x2_t){pack2<TC>(y0, y1), pack2<TC>(y2, y3)};
    }
  }
}

static bool g_post_gn = true;     // plan switch 28
void set_gemm_post_gn(bool on) { g_post_gn = on; }
void set_gemm_splitk(bool on) { g_splitk = on; }

void set_gemm_geom(int v) { g_dma_geom = v; }

// Tile geometry (0: 256x128x3, 1: 256x256x2, 2: 128x128x2, 3: 512x128x2) and split-K factor of a launch.
static bool g_midsplit = false;  // mid-size problems: 256x256 tiles + split-K instead of 128x128 tiles (round 1 default; with the
                                 // 4-stage 128x128 ring the unsplit plan is as fast and saves the second pass: -0.27 ms at cfg2)
void set_gemm_midsplit(bool on) { g_midsplit = on; }

static int g_split_tiles = 192, g_split_target = 256;     // plan constants (development keys 30 / 31)
void set_gemm_split_consts(int tiles, int target) { if (tiles > 0) g_split_tiles = tiles; if (target > 0) g_split_target = target; }

DmaPlan dma_plan(const GemmParams& p) {
  const long z = (long)p.nz0 * p.nz1;
  const int nk = ((p.C0 + p.C1 + 63) / 64) * p.taps + (p.CX0 + p.CX1) / 64;
  const bool can_split = p.ws && z == 1;
  auto fits = [&](int ns) { return (size_t)ns * p.M * p.N * 4 <= p.ws_bytes; };
  const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * z;
  int geom;
  if (g_dma_geom == 1) geom = 0;
  else if (g_dma_geom == 2) geom = 2;
  else if (g_dma_geom == 3) geom = 1;
  else if (g_dma_geom == 4) geom = 3;
  else if (p.M < 256) geom = 2;
  else {
    // largest tile that still gives about one workgroup per CU: big tiles are 15-20 % more efficient
    // (860-1020 vs 710-830 TFLOP/s on the conv shapes), but a short problem spread over few
    // workgroups is latency-bound and finishes sooner with more, smaller tiles
    const long t128 = (long)((p.M + 255) / 256) * ((p.N + 127) / 128) * z;
    const long t512 = (long)((p.M + 511) / 512) * ((p.N + 127) / 128) * z;
    if (g_dma_geom == 5) geom = (p.N >= 256 && p.N % 256 == 0) ? 1 : 0;   // former policy, kept for A/B
    else if (p.N % 256 == 0 && t256 >= 200) geom = 1;
    else if (g_dma_geom != 6 && p.N <= 128 && t512 >= 200) geom = 3;      // narrow outputs: tall tile, same 128x64 wave tile
    else if (t128 >= 200) geom = 0;
    else if (g_midsplit && g_splitk && !g_force_nsplit && can_split && p.N % 256 == 0 && t256 >= 48 && nk >= 32 &&
             std::min<long>(256 / t256, nk / 8) >= 2 && fits((int)std::min<long>(256 / t256, nk / 8))) {
      // long-K problem with 48..128 big tiles (conv at the 16x16 level of cfg2): the 128x128 geometry would
      // run one 4-wave workgroup per CU (320-470 TFLOP/s measured); 256x256 tiles with the K loop split so
      // that tiles x splits ~ 256 workgroups reach 570-770
      return {1, (int)std::min<long>(256 / t256, nk / 8)};
    }
    else geom = 2;
  }
  const int BM = geom == 2 ? 128 : (geom == 3 ? 512 : 256), BN = geom == 1 ? 256 : 128;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  int nsplit = 1;
  if (g_force_nsplit > 0) {
    if (can_split && g_force_nsplit > 1 && nk >= 2 * g_force_nsplit) nsplit = g_force_nsplit;
  } else if (g_splitk && can_split && tiles < g_split_tiles && nk >= 16) {
    // the tile grid cannot fill the chip and the K loop is long (low-resolution levels)
    // (tiles x splits ~ one workgroup per CU: a target of 256 measured 0.2 ms per step better than 384 at cfg2 and cfg3,
    // 128 .. 192 and 320 .. 768 worse; at least 4 K-tiles per split, 2 .. 12 within noise)
    nsplit = std::min(std::min(nk / 4, (g_split_target + tiles - 1) / tiles), 32);
  }
  while (nsplit > 1 && !fits(nsplit)) --nsplit;
  // 128 x 128 tiles on at most one workgroup per CU: such a launch is bound by the latency of its K-steps (0.47 us each with
  // the 2-stage ring: every step waits for the DMA issued one step earlier), so it takes the 4-stage ring (geometry 4: the
  // same tile with three K-tiles in flight, 128 KiB of LDS)
  if (geom == 2 && g_deep_ring && (long)tiles * nsplit * z <= num_cus()) geom = 4;
  return {geom, nsplit};
}
bool g_fuse_shortcut = true;
void set_gemm_fuse_shortcut(bool on) { g_fuse_shortcut = on; }
bool gemm_can_fuse_shortcut(const GemmParams& p) {
  return g_fuse_shortcut && dma_eligible(p) && p.taps == 9 && !p.a_up;
}
static bool dma_uses_splitk(const GemmParams& p) { return dma_plan(p).nsplit > 1; }
static int dma_pick_geom(const GemmParams& p) { return dma_plan(p).geom; }

// the vectorised split-K second pass (which also produces the GroupNorm column statistics) applies
static bool splitk_reduce_vec_ok(const GemmParams& p) {
  return p.N % 4 == 0 && p.ldc % 4 == 0 && (!p.R || p.ldr % 4 == 0) && (!p.bias_bn || p.ld_bn % 4 == 0) && !p.geglu;
}


// the second pass can apply a following GroupNorm: split-K launch whose samples are blocks of <= 256 rows, whole groups per
// 64-column slab, no per-chunk column statistics wanted on top
bool gemm_fuses_post_gn(const GemmParams& p, int groups) {
  if (!g_post_gn || !dma_eligible(p) || p.nz0 * p.nz1 != 1 || p.dtype == DT_F32 || !dma_uses_splitk(p) || !splitk_reduce_vec_ok(p)) return false;
  const int HW = p.rows_per_batch;
  if (HW < 1 || HW > 256 || p.M % HW != 0 || p.bias_m || p.c_nchw || (p.col_stats && HW % 64 != 0)) return false;
  if (p.r_up && (HW != p.H * p.W)) return false;
  if (groups <= 0 || p.N % 64 != 0 || p.N % groups != 0 || 64 % (p.N / groups) != 0) return false;
  return true;
}

template <typename TC>
static int launch_splitk_reduce_t(const GemmParams& p, int nsplit, hipStream_t stream) {
  if (p.gn_out) {
    GemmParams q = p;
    q.gn_out = nullptr; q.gn_gamma = q.gn_beta = nullptr;
    T2P_REQUIRE(p.gn_gamma && p.gn_beta && gemm_fuses_post_gn(q, p.gn_groups), "post-GroupNorm second pass: ask gemm_fuses_post_gn first");
    const int HW = p.rows_per_batch;
    dim3 g(p.N / 64, p.M / HW);
    if (HW <= 64) hipLaunchKernelGGL((splitk_reduce_gn_kernel<TC, 1>), g, dim3(1024), 0, stream, p, nsplit);
    else hipLaunchKernelGGL((splitk_reduce_gn_kernel<TC, 4>), g, dim3(1024), 0, stream, p, nsplit);
    T2P_HIP_CHECK(hipGetLastError());
    return T2P_OK;
  }
  if (splitk_reduce_vec_ok(p)) {
    // 64 x 64 output blocks when column statistics are wanted (their chunking), 16 x 64 otherwise
    const int rows = (p.col_stats || (long)p.M * p.N >= (1L << 22)) ? 64 : 16;
    dim3 g((p.N + 63) / 64, (p.M + rows - 1) / rows);
    if (rows == 64) hipLaunchKernelGGL((splitk_reduce_vec_kernel<TC, 64>), g, dim3(1024), 0, stream, p, nsplit);
    else hipLaunchKernelGGL((splitk_reduce_vec_kernel<TC, 16>), g, dim3(256), 0, stream, p, nsplit);
  } else {
    const long total = (long)p.M * p.N;
    const int g = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel<TC>, dim3(g), dim3(256), 0, stream, p, nsplit);
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_splitk_reduce(const GemmParams& p, int nsplit, hipStream_t stream) {
  return p.dtype == DT_BF16 ? launch_splitk_reduce_t<bf16_t>(p, nsplit, stream) : launch_splitk_reduce_t<f16_t>(p, nsplit, stream);
}

// true when launch_gemm(p) with p.geglu set will apply the fused GEGLU epilogue
bool gemm_fuses_geglu(const GemmParams& p) {
  if (!dma_eligible(p) || p.nz0 * p.nz1 != 1 || p.c_f32 || p.R || p.bias_bn || p.bias_m) return false;
  if (p.N % 4 != 0 || p.ldc % 2 != 0) return false;
  return !dma_uses_splitk(p);
}

// true when launch_gemm(p) will run the LDS-DMA kernel without split-K and with a vector
// epilogue, i.e. when a non-null p.col_stats will be filled
// same with a compute-dtype (16-bit) output: statistics are still taken from the fp32 values
bool gemm_fuses_col_stats_lowp(const GemmParams& p) {
  GemmParams q = p;
  q.c_f32 = 1;
  return !p.c_f32 && p.dtype != DT_F32 && gemm_fuses_col_stats(q);
}

bool gemm_fuses_col_stats(const GemmParams& p) {
  if (!dma_eligible(p) || p.nz0 * p.nz1 != 1 || !p.c_f32) return false;
  if (p.N % 4 != 0 || p.ldc % 4 != 0 || (p.R && p.ldr % 4 != 0) || (p.bias_bn && p.ld_bn % 4 != 0)) return false;
  if (dma_uses_splitk(p) && !splitk_reduce_vec_ok(p)) return false;   // scalar second pass: no statistics
  return p.M % 64 == 0;
}

// GemmParams::c_frag: a plain product on the 256 x 256 register-epilogue kernel, 16-bit output, whole 32-row x 32-column fragments
bool gemm_writes_frag_major(const GemmParams& p) {
  if (!dma_eligible(p) || p.taps != 1 || p.c_f32 || p.geglu || p.col_stats || p.gn_out || p.r_up || p.c_nchw) return false;
  if (g_dma_ring != 2 || g_dma_geom != 0) return false;
  const DmaPlan plan = dma_plan(p);
  if (plan.geom != 1 || plan.nsplit != 1 || !reg_epilogue_ok(p, 1, 0)) return false;
  if (p.frag_col0 < 0 || p.frag_col0 % 64 != 0 || (p.N - p.frag_col0) % 32 != 0 || p.frag_ns != (p.N - p.frag_col0) / 32) return false;
  if (p.rows_per_batch > 1 ? (p.rows_per_batch % 32 != 0 || p.M % p.rows_per_batch != 0 || p.nz0 * p.nz1 != 1) : (p.M % 32 != 0 || p.nz1 != 1)) return false;
  return true;
}

#endif  // T2P_PART_HOST

#if T2P_PART_DMA
template <typename TC, int MODE, int BM, int BN, int WM, int WN, int NST, int NSTB = NST, bool MF16 = false, bool REGE = false, bool CF = false>
static int launch_dma_geom(const GemmParams& p, hipStream_t stream) {
  constexpr int smem = NST * BM * 128 + NSTB * BN * 128;
  constexpr int threads = WM * WN * 64;
  if constexpr (MF16 && !REGE && MODE != 3) {          // the register epilogue wherever it covers the launch
    if (reg_epilogue_ok(p, dma_plan(p).nsplit, g_dbg)) return launch_dma_geom<TC, MODE, BM, BN, WM, WN, NST, NSTB, true, true>(p, stream);
  }
  void (*kern)(const GemmParams, const int, const int, const int);
  if constexpr (CF) kern = gemm_dma_cfrag_kernel<TC, BM, BN, WM, WN, MODE, NST, NSTB>;     // part of the output fragment-major (GemmParams::c_frag)
  else kern = gemm_dma_kernel<TC, BM, BN, WM, WN, MODE, NST, NSTB, MF16, REGE>;
  T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));
  const int Mq = MODE == 3 ? p.M / 4 : p.M;               // MODE 3: one output phase per launch
  const int tiles_m = (Mq + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nsplit = MODE == 3 ? 1 : dma_plan(p).nsplit;
  dim3 grid(tiles_m * tiles_n, MODE == 3 ? (p.up_phase == 5 ? 4 : 1) : p.nz0 * p.nz1, nsplit);
  ProfRec rec;
  if (g_prof_on) {
    T2P_HIP_CHECK(hipEventCreate(&rec.a));
    T2P_HIP_CHECK(hipEventCreate(&rec.b));
    prof_shape(rec, p);
    rec.flops = MODE == 3 ? 2.0 * Mq * p.N * 4.0 * (p.C0 + p.C1) * (p.up_phase == 5 ? 4 : 1)     // the multiplications this launch executes
                          : 2.0 * p.M * p.N * ((double)p.taps * (p.C0 + p.C1) + p.CX0 + p.CX1) * p.nz0 * p.nz1;
    rec.kind = p.taps == 9 ? 0 : 1;
    static const std::string kname = std::string(CF ? "gemm_dma_cfrag_kernel<" : "gemm_dma_kernel<") + (dtype_of<TC>::value == DT_F16 ? "f16_t" : "bf16_t") + ", " +
                                     std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) +
                                     ", " + std::to_string(MODE) + ", " + std::to_string(NST) + ", " + std::to_string(NSTB) + ", " +
                                     (MF16 ? "true" : "false") + ", " + (REGE ? "true" : "false") + ">";
    rec.name = kname.c_str();
    {
      const double Ct = p.C0 + p.C1, z = (double)p.nz0 * p.nz1;
      const double a_rows = p.a_up ? p.M / 4.0 : p.M;                    // up-sampling convs gather from the half-resolution map
      const double Cx = p.CX0 + p.CX1;                                   // shortcut segment: its input rows and weights once
      rec.bytes = z * (a_rows * Ct * 2 + (double)p.N * p.taps * Ct * 2 + (double)p.M * Cx * 2 + (double)p.N * Cx * 2 +
                       (double)p.M * (p.geglu ? p.N / 2 : p.N) * (p.c_f32 ? 4 : 2) +
                       (p.R ? (double)p.M * p.N * (p.r_lowp ? 2 : 4) / (p.r_up ? 4 : 1) : 0.0) + (p.bias_bn ? (double)(p.M / p.rows_per_batch) * p.N * 4 : 0.0));
    }
    T2P_HIP_CHECK(hipEventRecord(rec.a, stream));
  }
  hipLaunchKernelGGL(kern, grid, dim3(threads), smem, stream, p, tiles_m, tiles_n, g_dbg);
  if (g_prof_on) T2P_HIP_CHECK(hipEventRecord(rec.b, stream));   // the main kernel only: comparable with rocprofv3's per-kernel average
  if (nsplit > 1) T2P_TRY(launch_splitk_reduce(p, nsplit, stream));
  if (g_prof_on) g_prof.push_back(rec);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// the dx-shared-stage kernel applies: a 3x3 convolution at full resolution on the 16x16x32 geometries whose tiles are whole image rows
static bool dxs_eligible(const GemmParams& p, const DmaPlan& plan) {
  if (!g_dxs || p.taps != 9 || p.a_up || plan.nsplit != 1 || (plan.geom != 1 && plan.geom != 3) || g_dma_ring != 2) return false;
  if (p.nz0 * p.nz1 != 1 || p.c_frag || p.geglu) return false;
  const int BM = plan.geom == 1 ? 256 : 512, HW = p.H * p.W;
  if (p.W % 64 != 0 || (p.W & (p.W - 1)) != 0 || BM % p.W != 0 || HW % BM != 0 || p.M % HW != 0) return false;
  if ((p.CX0 + p.CX1) % 64 != 0) return false;
  return reg_epilogue_ok(p, 1, 0);            // the kernel carries the register epilogue only
}

template <typename TC, int BM, int BN, int WM, int WN>
static int launch_dxs(const GemmParams& p, hipStream_t stream) {
  constexpr int smem = 2 * BM * 128 + 2 * BN * 128;
  auto kern = gemm_dxs_kernel<TC, BM, BN, WM, WN>;
  T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));
  const int tiles_m = p.M / BM, tiles_n = (p.N + BN - 1) / BN;
  ProfRec rec;
  if (g_prof_on) {
    T2P_HIP_CHECK(hipEventCreate(&rec.a));
    T2P_HIP_CHECK(hipEventCreate(&rec.b));
    prof_shape(rec, p);
    rec.flops = 2.0 * p.M * p.N * (9.0 * (p.C0 + p.C1) + p.CX0 + p.CX1);
    rec.kind = 0;
    static const std::string kname = std::string("gemm_dxs_kernel<") + (dtype_of<TC>::value == DT_F16 ? "f16_t" : "bf16_t") + ", " +
                                     std::to_string(BM) + ", " + std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ">";
    rec.name = kname.c_str();
    const double Ct = p.C0 + p.C1, Cx = p.CX0 + p.CX1;
    rec.bytes = (double)p.M * Ct * 2 + (double)p.N * 9 * Ct * 2 + (double)p.M * Cx * 2 + (double)p.N * Cx * 2 + (double)p.M * p.N * (p.c_f32 ? 4 : 2) +
                (p.R ? (double)p.M * p.N * (p.r_lowp ? 2 : 4) : 0.0) + (p.bias_bn ? (double)(p.M / p.rows_per_batch) * p.N * 4 : 0.0);
    T2P_HIP_CHECK(hipEventRecord(rec.a, stream));
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), smem, stream, p, tiles_m, tiles_n, g_dbg);
  if (g_prof_on) {
    T2P_HIP_CHECK(hipEventRecord(rec.b, stream));
    g_prof.push_back(rec);
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// a 3x3 convolution on a 2x up-sampled map as four 2x2 convolutions on the source map (kernel MODE 3): the caller supplied
// the phase weights (GemmParams::Bw4), the 16x16x32 geometries with the register epilogue run it
static bool up4_eligible(const GemmParams& p, const DmaPlan& plan) {
  if (!g_up4 || !p.Bw4 || p.taps != 9 || !p.a_up || plan.nsplit != 1 || (plan.geom != 1 && plan.geom != 3) || g_dma_ring != 2) return false;
  if (p.nz0 * p.nz1 != 1 || p.R || p.bias_m || p.geglu || p.A1) return false;
  const int Ws = p.W / 2, HWs = (p.H / 2) * (p.W / 2);
  if (p.H % 2 || p.W % 2 || Ws % 16 != 0 || HWs % 64 != 0 || p.M % (p.H * p.W) != 0) return false;
  if (p.bias_bn && p.rows_per_batch != p.H * p.W) return false;
  if ((long)p.N * p.ldb4 * 2 * 4 >= (1L << 31) - 64) return false;
  return reg_epilogue_ok(p, 1, g_dbg);
}

template <typename TC, int MODE>
int launch_dma_mode(const GemmParams& p, hipStream_t stream) {
  const DmaPlan plan = dma_plan(p);
  if constexpr (MODE == 0) {
    if (p.c_frag) {
      T2P_REQUIRE(gemm_writes_frag_major(p), "this product does not take the fragment-major output (ask gemm_writes_frag_major)");
      return launch_dma_geom<TC, 0, 256, 256, 2, 4, 2, 2, true, true, true>(p, stream);
    }
  }
  if constexpr (MODE == 2) {
    if (up4_eligible(p, plan)) {
      GemmParams q = p;
      q.ldb = p.ldb4;
      q.Bw = p.Bw4;
      q.up_phase = 5;                  // all four phases in one launch (grid.y = phase)
      if (plan.geom == 1) return launch_dma_geom<TC, 3, 256, 256, 2, 4, 2, 2, true, true>(q, stream);
      return launch_dma_geom<TC, 3, 512, 128, 4, 2, 2, 2, true, true>(q, stream);
    }
  }
  if constexpr (MODE == 1) {
    if (dxs_eligible(p, plan)) {
      if (plan.geom == 1) return launch_dxs<TC, 256, 256, 2, 4>(p, stream);
      return launch_dxs<TC, 512, 128, 4, 2>(p, stream);
    }
  }
  switch (plan.geom) {
    case 1:
      if (g_dma_ring == 2) return launch_dma_geom<TC, MODE, 256, 256, 2, 4, 2, 2, true>(p, stream);   // 16x16x32 MFMA
      return launch_dma_geom<TC, MODE, 256, 256, 2, 4, 2>(p, stream);      // ring 1: the 32x32x16 MFMA shape on the same tile
    case 2: return launch_dma_geom<TC, MODE, 128, 128, 2, 2, 2>(p, stream);
    case 4: return launch_dma_geom<TC, MODE, 128, 128, 2, 2, 4>(p, stream);
    case 3:
      if (g_dma_ring == 2) return launch_dma_geom<TC, MODE, 512, 128, 4, 2, 2, 2, true>(p, stream);
      return launch_dma_geom<TC, MODE, 512, 128, 4, 2, 2>(p, stream);
    default: return launch_dma_geom<TC, MODE, 256, 128, 4, 2, 3>(p, stream);
  }
}

#endif  // T2P_PART_DMA

#if T2P_PART_HOST
template <typename TC>
static int launch_dma(const GemmParams& p, hipStream_t stream) {
  if (p.taps == 9) return p.a_up ? launch_dma_mode<TC, 2>(p, stream) : launch_dma_mode<TC, 1>(p, stream);
  return launch_dma_mode<TC, 0>(p, stream);
}

template <typename TC, bool AF32>
static int launch_tile(const GemmParams& p, hipStream_t stream) {
  // small problems: 64x64 tiles give more workgroups and waste less on ragged edges
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nz0 * p.nz1;
  if (p.N <= 64 || p.M <= 64 || tiles128 < 128) return launch_t<TC, AF32, 64, 64>(p, stream);
  return launch_t<TC, AF32, 128, 128>(p, stream);
}

// ---- thin-output 3x3 convolution: the network head (nf -> 5 or 8 channels, NCHW fp32 x row scale) ----------
// As a GEMM this layer has N = 5: the 64-wide tile of the register-staged kernel spends 92 % of its MFMAs on
// padding and re-reads every input pixel nine times from L2 (426 us at cfg2 against a ~55 us HBM bound).  Here
// a workgroup owns an 8 x 16 pixel tile: the 10 x 18 halo of 16-bit input pixels is staged once in LDS (chunk
// index XOR pixel index: conflict-free 16-byte fragment reads), the <= 8 weight rows too, and each wavefront
// runs v_mfma_f32_16x16x32 on two rows of 16 pixels with the 16 MFMA columns holding the output channels.
struct ThinConvArgs {
  const void* x; const void* w; const float* bias; const float* row_scale; float* out;
  int B, H, W, Cin, Cout;
  long ldw;
  // optional: x is the raw map; act(GroupNorm(x)) is applied while the halo is staged (out.0 + out.1 of the head, ncsnpp.py:211-216)
  const float* gn_stats; const float* gn_gamma; const float* gn_beta;
  int gn_groups, gn_silu;
};

template <typename TC, int CH>   // CH: channels staged per pass (32, 64 or 128)
__global__ __launch_bounds__(512) void thin_conv_kernel(const ThinConvArgs a) {
  constexpr int TH = 8, TW = 16, WP = TW + 2, NPIX = (TH + 2) * WP;
  constexpr int cb = CH * 2, nchunk = CH / 8;               // bytes / 16-byte chunks per staged pixel
  constexpr int wrow = 9 * cb + 16;                         // padded weight row: lanes of a fragment read hit distinct banks
  constexpr int sw = (nchunk < 8 ? nchunk : 8) - 1;         // swizzle mask: stays inside the pixel's own chunks
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float nsc[256], nsh[256];                      // per-channel scale / shift of the fused GroupNorm (Cin <= 256)
  unsigned char* halo = smem;
  unsigned char* wl = smem + NPIX * cb;
  const int tid = threadIdx.x;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const bool norm = a.gn_stats != nullptr;
  if (norm) {
    // the same arithmetic as the separate GroupNorm-apply pass (gn_apply16_kernel): y = x * (rstd gamma) + (beta - mean rstd gamma)
    if (tid < a.Cin) {
      const int g = tid / (a.Cin / a.gn_groups);
      const float mean = a.gn_stats[2 * (b * a.gn_groups + g)], rstd = a.gn_stats[2 * (b * a.gn_groups + g) + 1];
      const float sc = rstd * a.gn_gamma[tid];
      nsc[tid] = sc;
      nsh[tid] = a.gn_beta[tid] - mean * sc;
    }
    __syncthreads();
  }
  const int x0 = tx * TW, y0 = ty * TH;
  const TC* xb = (const TC*)a.x + (long)b * a.H * a.W * a.Cin;
  const int wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;   // wave = pixel row of the tile
  const bool colv = r < a.Cout;                             // MFMA column = output channel
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < a.Cin; c0 += CH) {                  // channel passes: 58 KiB of LDS at CH = 128 -> two blocks per CU
    if (c0) __syncthreads();                                // the previous pass is no longer read
    for (int i = tid; i < NPIX * nchunk; i += 512) {
      const int P = i / nchunk, j = i - P * nchunk;
      const int gy = y0 + P / WP - 1, gx = x0 + P % WP - 1;
      u32x4_t v = {0u, 0u, 0u, 0u};                         // zero padding outside the map (of the NORMALISED map: stays zero)
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        v = *(const u32x4_t*)(xb + ((long)gy * a.W + gx) * a.Cin + c0 + j * 8);
        if (norm) {
          union { u32x4_t u; TC e[8]; } in, o;
          in.u = v;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = c0 + j * 8 + e;
            float y = to_f32(in.e[e]) * nsc[c] + nsh[c];
            if (a.gn_silu) y = y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y * -1.44269504088896341f));
            o.e[e] = from_f32<TC>(y);
          }
          v = o.u;
        }
      }
      *(u32x4_t*)(halo + P * cb + (((j & ~sw) | ((j ^ P) & sw)) << 4)) = v;
    }
    for (int i = tid; i < a.Cout * 9 * nchunk; i += 512) {
      const int co = i / (9 * nchunk), k = i - co * 9 * nchunk, tap = k / nchunk, j = k - tap * nchunk;
      *(u32x4_t*)(wl + co * wrow + k * 16) = *(const u32x4_t*)((const TC*)a.w + (long)co * a.ldw + (long)tap * a.Cin + c0 + j * 8);
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      const int P = (wave + dy) * WP + r + dx;
      u32x4_t af[CH / 32], bf[CH / 32];                     // one tap's fragments in flight before its MFMAs (<= 128 VGPRs
#pragma unroll                                               //  in total: two 512-thread blocks per CU)
      for (int kc = 0; kc < CH / 32; ++kc) {
        const int j = kc * 4 + g;
        af[kc] = *(const u32x4_t*)(halo + P * cb + (((j & ~sw) | ((j ^ P) & sw)) << 4));
        bf[kc] = (u32x4_t){0u, 0u, 0u, 0u};
        if (colv) bf[kc] = *(const u32x4_t*)(wl + r * wrow + (tap * nchunk + j) * 16);
      }
#pragma unroll
      for (int kc = 0; kc < CH / 32; ++kc) Mma16<TC>::run(af[kc], bf[kc], acc);
    }
  }
  if (!colv) return;
  // 16x16 accumulator: column = lane & 15 (channel), row = (lane >> 4) * 4 + register (pixel of the row)
  const int y = y0 + wave;
  if (y >= a.H) return;
  const float bias = a.bias ? a.bias[r] : 0.f, sc = a.row_scale ? a.row_scale[b] : 1.f;
  float* o = a.out + (((long)b * a.Cout + r) * a.H + y) * a.W;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int x = x0 + g * 4 + v;
    if (x < a.W) o[x] = (acc[v] + bias) * sc;
  }
}

static bool g_thin_conv = true;
static bool g_a_norm = true;         // GroupNorm + SiLU of the head inside the head convolution's halo staging (plan switch 34)
void set_gemm_thin_conv(bool on) { g_thin_conv = on; }
void set_gemm_a_norm(bool on) { g_a_norm = on; }

// true when launch_gemm(p) will take the thin-output kernel
static bool thin_conv_eligible(const GemmParams& p) {
  if (p.dtype == DT_F32 || p.a_f32 || p.taps != 9 || p.a_up || !p.c_nchw || !p.c_f32) return false;
  if (p.A1 || p.R || p.bias_m || p.bias_bn || p.nz0 * p.nz1 != 1 || p.alpha != 1.f || p.geglu) return false;
  if (p.N > 8 || (p.C0 != 32 && p.C0 != 64 && p.C0 != 128 && p.C0 != 256) || p.lda0 != p.C0) return false;
  return p.rows_per_batch == p.H * p.W && p.M % (p.H * p.W) == 0;
}

bool gemm_applies_a_norm(const GemmParams& p) { return g_thin_conv && g_a_norm && thin_conv_eligible(p) && p.C0 <= 256; }

template <typename TC, int CH>
static int launch_thin_conv_ch(const ThinConvArgs& a, hipStream_t stream) {
  constexpr int smem = 180 * CH * 2 + 8 * (9 * CH * 2 + 16);
  const int tiles = ((a.W + 15) / 16) * ((a.H + 7) / 8);
  hipLaunchKernelGGL((thin_conv_kernel<TC, CH>), dim3((unsigned)((long)a.B * tiles)), dim3(512), smem, stream, a);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

template <typename TC>
static int launch_thin_conv(const GemmParams& p, hipStream_t stream) {
  ThinConvArgs a;
  a.x = p.A0; a.w = p.Bw; a.bias = p.bias_n; a.row_scale = p.row_scale; a.out = (float*)p.C;
  a.B = p.M / (p.H * p.W); a.H = p.H; a.W = p.W; a.Cin = p.C0; a.Cout = p.N; a.ldw = p.ldb;
  a.gn_stats = p.an_stats; a.gn_gamma = p.an_gamma; a.gn_beta = p.an_beta; a.gn_groups = p.an_groups; a.gn_silu = p.an_silu;
  if (p.C0 == 32) return launch_thin_conv_ch<TC, 32>(a, stream);
  if (p.C0 == 64) return launch_thin_conv_ch<TC, 64>(a, stream);
  return launch_thin_conv_ch<TC, 128>(a, stream);           // 128 or 256 channels: one or two passes
}

int launch_gemm(const GemmParams& p, hipStream_t stream) {
  const int vec = p.dtype == DT_F32 ? 4 : 8;
  const int Ctot = p.C0 + p.C1;
  T2P_REQUIRE(p.A0 && p.Bw && (p.C || p.gn_out), "null operand");
  if (p.gn_out) {
    GemmParams q = p;
    q.gn_out = nullptr; q.gn_gamma = q.gn_beta = nullptr;
    T2P_REQUIRE(p.gn_gamma && p.gn_beta && gemm_fuses_post_gn(q, p.gn_groups), "a following GroupNorm rides only on the split-K second pass (ask gemm_fuses_post_gn)");
  }
  T2P_REQUIRE(p.M > 0 && p.N > 0 && Ctot > 0, "empty problem");
  T2P_REQUIRE(p.taps == 1 || p.taps == 9, "taps must be 1 or 9");
  T2P_REQUIRE(p.dtype != DT_F32 || p.a_f32, "fp32 compute takes fp32 sources");
  T2P_REQUIRE(!p.r_lowp || (p.R && p.dtype != DT_F32), "a 16-bit residual needs a 16-bit compute dtype");
  T2P_REQUIRE(p.lda0 % vec == 0 && p.ldb % vec == 0, "row strides must be multiples of the 16-byte vector");
  T2P_REQUIRE(p.lda0 >= ((p.C0 + vec - 1) / vec) * vec, "lda0 too small");
  T2P_REQUIRE(p.ldb >= (long)(p.taps - 1) * Ctot + ((Ctot + vec - 1) / vec) * vec, "ldb too small");
  if (p.A1) {
    T2P_REQUIRE(p.C0 % vec == 0 && p.lda1 % vec == 0 && p.C1 % vec == 0, "concat sources must be vector aligned");
    T2P_REQUIRE(p.nz0 * p.nz1 == 1, "batched GEMM takes a single A source");
  } else {
    T2P_REQUIRE(p.C1 == 0, "C1 without A1");
  }
  if (p.taps == 9) T2P_REQUIRE(Ctot % vec == 0, "3x3 convolution needs channels % vector == 0");
  if (p.taps == 9 || p.a_up || p.r_up || p.c_nchw) {
    T2P_REQUIRE(p.H > 0 && p.W > 0 && p.M % (p.H * p.W) == 0, "spatial mode needs H, W with M = batch*H*W");
    T2P_REQUIRE(p.nz0 * p.nz1 == 1, "spatial modes are not batched");
  }
  if (p.a_up || p.r_up) T2P_REQUIRE(p.H % 2 == 0 && p.W % 2 == 0, "up-sampling needs even H, W");
  if (p.r_up || p.c_nchw) T2P_REQUIRE(p.rows_per_batch == p.H * p.W, "rows_per_batch must be H*W");
  T2P_REQUIRE(p.rows_per_batch > 0, "rows_per_batch");
  T2P_REQUIRE(!p.c_nchw || (p.row_scale && p.c_f32), "nchw store needs row_scale and fp32 output");
  T2P_REQUIRE(((uintptr_t)p.A0 % 16) == 0 && ((uintptr_t)p.Bw % 16) == 0 && ((uintptr_t)p.A1 % 16) == 0,
              "operands must be 16-byte aligned");
  T2P_REQUIRE((long)(p.M + 127) / 128 < 65536, "M too large for grid.y");
  if (p.X0 || p.X1 || p.CX0 || p.CX1) {
    GemmParams q = p;
    q.X0 = q.X1 = nullptr; q.CX0 = q.CX1 = 0;
    T2P_REQUIRE(p.X0 && p.CX0 > 0 && p.CX0 % 64 == 0 && p.CX1 % 64 == 0 && (p.X1 != nullptr) == (p.CX1 > 0), "shortcut segment: whole 64-channel K-tiles");
    T2P_REQUIRE(gemm_can_fuse_shortcut(q), "shortcut segment: only on the LDS-DMA 3x3 convolution (ask gemm_can_fuse_shortcut)");
    T2P_REQUIRE(((uintptr_t)p.X0 % 16) == 0 && ((uintptr_t)p.X1 % 16) == 0 && p.ldx0 % 8 == 0 && p.ldx1 % 8 == 0, "shortcut segment: 16-byte aligned rows");
    T2P_REQUIRE(((long)p.M + 512) * std::max(p.ldx0, p.ldx1) * 2 < (1L << 31) - 64 && std::max(p.ldx0, p.ldx1) * 2 < (1L << 24), "shortcut segment: 32-bit offsets");
    T2P_REQUIRE(p.ldb >= 9L * (p.C0 + p.C1) + p.CX0 + p.CX1, "shortcut segment: weight rows hold 9 (C0 + C1) + CX0 + CX1 columns");
  }
  if (p.an_stats) {
    GemmParams q = p;
    q.an_stats = nullptr;
    T2P_REQUIRE(p.an_gamma && p.an_beta && p.an_groups > 0 && p.C0 % p.an_groups == 0 && gemm_applies_a_norm(q),
                "a GroupNorm of the A operand rides only on the thin-output head convolution (ask gemm_applies_a_norm)");
  }
  if (g_thin_conv && thin_conv_eligible(p)) {
    // the head convolution: timed as "conv on the register-staged kernel" by the profile hooks' kind 2
    ProfRec rec;
    if (g_prof_on) {
      T2P_HIP_CHECK(hipEventCreate(&rec.a));
      T2P_HIP_CHECK(hipEventCreate(&rec.b));
      prof_shape(rec, p);
    prof_shape(rec, p);
      rec.flops = 2.0 * p.M * p.N * 9.0 * p.C0; rec.kind = 2; rec.name = nullptr; rec.bytes = 0;
      T2P_HIP_CHECK(hipEventRecord(rec.a, stream));
    }
    const int rc = p.dtype == DT_BF16 ? launch_thin_conv<bf16_t>(p, stream) : launch_thin_conv<f16_t>(p, stream);
    if (g_prof_on) {
      T2P_HIP_CHECK(hipEventRecord(rec.b, stream));
      g_prof.push_back(rec);
    }
    return rc;
  }
  if (dma_eligible(p)) return p.dtype == DT_BF16 ? launch_dma<bf16_t>(p, stream) : launch_dma<f16_t>(p, stream);
  switch (p.dtype) {
    case DT_F32: return launch_tile<float, true>(p, stream);
    case DT_BF16: return p.a_f32 ? launch_tile<bf16_t, true>(p, stream) : launch_tile<bf16_t, false>(p, stream);
    case DT_F16: return p.a_f32 ? launch_tile<f16_t, true>(p, stream) : launch_tile<f16_t, false>(p, stream);
  }
  set_last_error("launch_gemm: unknown dtype");
  return T2P_ERR_INVALID;
}

#endif  // T2P_PART_HOST

#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 1
template int launch_dma_mode<f16_t, 0>(const GemmParams&, hipStream_t);
#endif
#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 2
template int launch_dma_mode<f16_t, 1>(const GemmParams&, hipStream_t);
#endif
#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 3
template int launch_dma_mode<f16_t, 2>(const GemmParams&, hipStream_t);
#endif
#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 4
template int launch_dma_mode<bf16_t, 0>(const GemmParams&, hipStream_t);
#endif
#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 5
template int launch_dma_mode<bf16_t, 1>(const GemmParams&, hipStream_t);
#endif
#if T2P_GEMM_PART == -1 || T2P_GEMM_PART == 6
template int launch_dma_mode<bf16_t, 2>(const GemmParams&, hipStream_t);
#endif

}  // namespace t2p
