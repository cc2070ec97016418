// HBM-bound kernels of the score network and the SDE update (gfx950).
// All activations are NHWC ([batch][pixel][channel], channel contiguous) so that 64-lane
// wavefronts read 16-byte vectors of consecutive channels (coalesced 1 KiB per wave-instruction).
#include "t2p_kernels.h"

namespace t2p {

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float silu_f(float x) { return x / (1.f + expf(-x)); }

template <typename T> __device__ inline void store4(T* p, float a, float b, float c, float d);
template <> __device__ inline void store4<float>(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ inline void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  uint2 u;
  u.x = (uint32_t)f32_to_bf16_bits(a) | ((uint32_t)f32_to_bf16_bits(b) << 16);
  u.y = (uint32_t)f32_to_bf16_bits(c) | ((uint32_t)f32_to_bf16_bits(d) << 16);
  *(uint2*)p = u;
}
template <> __device__ inline void store4<f16_t>(f16_t* p, float a, float b, float c, float d) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  h4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
  *(h4*)p = v;
}

template <typename T> __device__ inline float4 load4(const T* p);
template <> __device__ inline float4 load4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ inline float4 load4<bf16_t>(const bf16_t* p) {
  const uint2 u = *(const uint2*)p;
  return make_float4(bf16_bits_to_f32((uint16_t)u.x), bf16_bits_to_f32((uint16_t)(u.x >> 16)),
                     bf16_bits_to_f32((uint16_t)u.y), bf16_bits_to_f32((uint16_t)(u.y >> 16)));
}
template <> __device__ inline float4 load4<f16_t>(const f16_t* p) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 v = *(const h4*)p;
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// ================================== GroupNorm statistics ======================================
// grid (nchunk, B, ceil(C / 1024)); each block: a slice of pixels x up to 1024 channels.  A
// thread keeps a fixed 4-channel vector and strides over pixels (register accumulation), then
// the block folds channels into groups through LDS in a fixed order.  Partials are combined in
// double by gn_finalize_kernel, also in a fixed order: results are bitwise reproducible.
static constexpr int GN_PIX_PER_CHUNK = 256;

int gn_num_chunks(int HW) { return (HW + GN_PIX_PER_CHUNK - 1) / GN_PIX_PER_CHUNK; }

template <typename TI>
__global__ __launch_bounds__(256) void gn_stats_kernel(GroupNormArgs a, int nchunk) {
  __shared__ float s_part[256][8];   // per thread: sum[4], sumsq[4] of its 4-channel vector
  const int C = a.C0 + a.C1;
  const int cpg = C / a.G;
  const int chunk = blockIdx.x, b = blockIdx.y, cblk = blockIdx.z;
  const int c_lo = cblk * 1024;
  const int cw = min(C - c_lo, 1024);                  // channels handled by this block
  const int nvec = cw >> 2;                            // 4-channel vectors
  const int ppi = 256 / nvec;                          // pixels per block iteration (>= 1)
  const int tid = threadIdx.x;
  const int p_lo = chunk * GN_PIX_PER_CHUNK, p_hi = min(a.HW, p_lo + GN_PIX_PER_CHUNK);
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  if (tid < ppi * nvec) {
    const int v = tid % nvec, po = tid / nvec;
    const int c = c_lo + v * 4;
    const TI* src; long ld; int cc;
    if (c < a.C0) { src = (const TI*)a.x0; ld = a.C0; cc = c; } else { src = (const TI*)a.x1; ld = a.C1; cc = c - a.C0; }
    const TI* base = src + (long)b * a.HW * ld + cc;
    int p = p_lo + po;
    for (; p + 3 * ppi < p_hi; p += 4 * ppi) {       // four independent 16-byte loads in flight
      const float4 t0 = load4<TI>(base + (long)p * ld);
      const float4 t1 = load4<TI>(base + (long)(p + ppi) * ld);
      const float4 t2 = load4<TI>(base + (long)(p + 2 * ppi) * ld);
      const float4 t3 = load4<TI>(base + (long)(p + 3 * ppi) * ld);
      s0 += (t0.x + t1.x) + (t2.x + t3.x); s1 += (t0.y + t1.y) + (t2.y + t3.y);
      s2 += (t0.z + t1.z) + (t2.z + t3.z); s3 += (t0.w + t1.w) + (t2.w + t3.w);
      q0 += (t0.x * t0.x + t1.x * t1.x) + (t2.x * t2.x + t3.x * t3.x);
      q1 += (t0.y * t0.y + t1.y * t1.y) + (t2.y * t2.y + t3.y * t3.y);
      q2 += (t0.z * t0.z + t1.z * t1.z) + (t2.z * t2.z + t3.z * t3.z);
      q3 += (t0.w * t0.w + t1.w * t1.w) + (t2.w * t2.w + t3.w * t3.w);
    }
    for (; p < p_hi; p += ppi) {
      const float4 t = load4<TI>(base + (long)p * ld);
      s0 += t.x; s1 += t.y; s2 += t.z; s3 += t.w;
      q0 += t.x * t.x; q1 += t.y * t.y; q2 += t.z * t.z; q3 += t.w * t.w;
    }
  }
  s_part[tid][0] = s0; s_part[tid][1] = s1; s_part[tid][2] = s2; s_part[tid][3] = s3;
  s_part[tid][4] = q0; s_part[tid][5] = q1; s_part[tid][6] = q2; s_part[tid][7] = q3;
  __syncthreads();
  if (tid < a.G) {
    // fold this block's channels of group `tid` in a fixed order (bitwise reproducible)
    const int g_lo = max(tid * cpg, c_lo), g_hi = min((tid + 1) * cpg, c_lo + cw);
    float s = 0.f, q = 0.f;
    for (int c = g_lo; c < g_hi; ++c) {
      const int v = (c - c_lo) >> 2, k = (c - c_lo) & 3;
      for (int po = 0; po < ppi; ++po) {
        s += s_part[po * nvec + v][k];
        q += s_part[po * nvec + v][4 + k];
      }
    }
    // [B][cblk][nchunk][G][2]
    float* dst = a.partial + ((((long)b * gridDim.z + cblk) * nchunk + chunk) * a.G + tid) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

__global__ void gn_finalize_kernel(GroupNormArgs a, int nparts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over B * G
  if (i >= a.B * a.G) return;
  const int b = i / a.G, g = i - b * a.G;
  double s = 0, q = 0;
  for (int k = 0; k < nparts; ++k) {
    const float* src = a.partial + (((long)b * nparts + k) * a.G + g) * 2;
    s += src[0];
    q += src[1];
  }
  const int C = a.C0 + a.C1;
  const double n = (double)a.HW * (C / a.G);
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  a.stats[2 * i] = (float)mean;
  a.stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
}

// one wavefront per (sample, group): lanes stride over (chunk, channel) pairs, double accumulation
__global__ __launch_bounds__(256) void gn_finalize_cols_kernel(const float* cs0, const float* cs1, int C0, int C1, int B, int HW,
                                                               int G, float eps, float* stats) {
  // one block per (sample, group): 256 threads walk the (chunk, channel-of-group) pairs with 4 independent loads in flight,
  // accumulate in double, and are folded in a fixed order (wavefront shuffles, then the 4 wavefronts in LDS): reproducible
  __shared__ double red[2][4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int C = C0 + C1, cpg = C / G;
  const int nchunk = HW >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = nchunk * cpg;
  auto at = [&](int i) -> const float* {
    const int ch = i / cpg, c = g * cpg + (i - ch * cpg);
    return c < C0 ? cs0 + ((long)(b * nchunk + ch) * C0 + c) * 2 : cs1 + ((long)(b * nchunk + ch) * C1 + (c - C0)) * 2;
  };
  double s = 0, q = 0;
  int i = tid;
  for (; i + 768 < total; i += 1024) {
    const float2 v0 = *(const float2*)at(i), v1 = *(const float2*)at(i + 256), v2 = *(const float2*)at(i + 512), v3 = *(const float2*)at(i + 768);
    s += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
    q += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
  }
  for (; i < total; i += 256) {
    const float2 v = *(const float2*)at(i);
    s += v.x;
    q += v.y;
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
  __syncthreads();
  if (tid == 0) {
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    stats[2 * (b * G + g)] = (float)mean;
    stats[2 * (b * G + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

int launch_gn_finalize_cols(const float* cs0, const float* cs1, int C0, int C1, int B, int HW, int G, float eps, float* stats,
                            hipStream_t s) {
  T2P_REQUIRE(cs0 && stats && (C1 == 0) == (cs1 == nullptr), "gn_finalize_cols arguments");
  T2P_REQUIRE(HW % 64 == 0 && (C0 + C1) % G == 0, "gn_finalize_cols needs 64-row chunks aligned to samples");
  hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(B * G), dim3(256), 0, s, cs0, cs1, C0, C1, B, HW, G, eps, stats);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

int launch_gn_stats(const GroupNormArgs& a, hipStream_t s) {
  const int C = a.C0 + a.C1;
  T2P_REQUIRE(a.x0 && a.partial && a.stats, "null pointer");
  T2P_REQUIRE(a.G >= 1 && a.G <= 32 && C % a.G == 0, "GroupNorm groups must divide C and be <= 32");
  T2P_REQUIRE(a.C0 % 4 == 0 && a.C1 % 4 == 0, "channels must be multiples of 4");
  T2P_REQUIRE((a.C1 == 0) == (a.x1 == nullptr), "second source mismatch");
  const int nchunk = gn_num_chunks(a.HW);
  const int ncblk = (C + 1023) / 1024;
  dim3 grid(nchunk, a.B, ncblk);
  if (a.lowp_dtype == DT_F16) hipLaunchKernelGGL(gn_stats_kernel<f16_t>, grid, dim3(256), 0, s, a, nchunk);
  else if (a.lowp_dtype == DT_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, a, nchunk);
  else hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, a, nchunk);
  const int tot = a.B * a.G;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((tot + 127) / 128), dim3(128), 0, s, a, nchunk * ncblk);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== GroupNorm apply (+SiLU, +2x2 mean) ==========================
// x * sigmoid(x) with the hardware's exp2 and reciprocal (1 ulp each; v_exp_f32, v_rcp_f32): 5 vector instructions per element --
// the IEEE-rounded reciprocal alone expands to ~8, and this function is most of the arithmetic of the GroupNorm-apply pass.
// exp2 of a large argument is +inf, its reciprocal 0: very negative x gives -0, not NaN
__device__ inline float silu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// grid (pixel chunks, B, ceil(C / 1024)).  A thread keeps one 4-channel vector: its scale/shift
// (rstd*gamma, beta - mean*rstd*gamma) are computed once, then it walks output pixels with 32-bit
// indexing; consecutive lanes cover consecutive channels (coalesced 16-byte loads, 8/16-byte stores).
static constexpr int GNA_PIX_PER_BLOCK = 64;      // 32 .. 256 measured equal within 0.3 % of a step
bool g_gn_apply16 = true;     // 16-bit GroupNorm apply with 16-byte accesses (plan switch 17)

template <typename TO, typename TI>
__global__ __launch_bounds__(256) void gn_apply_kernel(GroupNormApplyArgs a, int pix_per_block) {
  const int C = a.C0 + a.C1;
  const int Ho = a.down ? a.H >> 1 : a.H, Wo = a.down ? a.W >> 1 : a.W;
  const int HWo = Ho * Wo;
  const int b = blockIdx.y, c_lo = blockIdx.z * 1024;
  const int nvec = min(C - c_lo, 1024) >> 2;
  const int ppi = 256 / nvec;
  const int tid = threadIdx.x;
  if (tid >= ppi * nvec) return;
  const int v = tid % nvec, po = tid / nvec;
  const int c = c_lo + v * 4;
  const int cpg = C / a.G;
  const TI* src; int ld, cc;
  if (c < a.C0) { src = (const TI*)a.x0; ld = a.C0; cc = c; } else { src = (const TI*)a.x1; ld = a.C1; cc = c - a.C0; }
  src += (long)b * a.H * a.W * ld + cc;
  const float4 ga = *(const float4*)(a.gamma + c), be = *(const float4*)(a.beta + c);
  float sc[4], sh[4];
  const float gv[4] = {ga.x, ga.y, ga.z, ga.w}, bv[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* st = a.stats + ((long)b * a.G + (c + k) / cpg) * 2;
    sc[k] = st[1] * gv[k];
    sh[k] = bv[k] - st[0] * sc[k];
  }
  TO* out = (TO*)a.out + (long)b * HWo * C + c;
  TO* raw = a.raw_out ? (TO*)a.raw_out + (long)b * HWo * C + c : nullptr;
  const int p_lo = blockIdx.x * pix_per_block, p_hi = min(HWo, p_lo + pix_per_block);
  if (!a.down) {
#pragma unroll 2
    for (int p = p_lo + po; p < p_hi; p += ppi) {
      const float4 t = load4<TI>(src + (long)p * ld);
      if (raw) store4<TO>(raw + (long)p * C, t.x, t.y, t.z, t.w);
      float y0 = t.x * sc[0] + sh[0], y1 = t.y * sc[1] + sh[1], y2 = t.z * sc[2] + sh[2], y3 = t.w * sc[3] + sh[3];
      if (a.silu) { y0 = silu_fast(y0); y1 = silu_fast(y1); y2 = silu_fast(y2); y3 = silu_fast(y3); }
      store4<TO>(out + (long)p * C, y0, y1, y2, y3);
    }
  } else {
    for (int p = p_lo + po; p < p_hi; p += ppi) {
      const int oy = p / Wo, ox = p - oy * Wo;
      float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ip = (2 * oy + (q >> 1)) * a.W + 2 * ox + (q & 1);
        const float4 t = load4<TI>(src + (long)ip * ld);
        float y0 = t.x * sc[0] + sh[0], y1 = t.y * sc[1] + sh[1], y2 = t.z * sc[2] + sh[2], y3 = t.w * sc[3] + sh[3];
        if (a.silu) { y0 = silu_fast(y0); y1 = silu_fast(y1); y2 = silu_fast(y2); y3 = silu_fast(y3); }
        o0 += y0; o1 += y1; o2 += y2; o3 += y3;
      }
      store4<TO>(out + (long)p * C, o0 * 0.25f, o1 * 0.25f, o2 * 0.25f, o3 * 0.25f);
    }
  }
}


// 16-bit in, 16-bit out, full resolution (the hot GroupNorm of f16 mode): a thread keeps 8 channels (16-byte loads and
// stores) and has the rows of 4 pixels in flight at once -- the pass is a pure HBM round trip, so bytes in flight per CU
// are what sets its rate (4.0 TB/s with 8-byte accesses and 2 pixels in flight at 128 channels; see profiles/README.md)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply16_kernel(GroupNormApplyArgs a, int pix_per_block) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const int C = a.C0 + a.C1;
  const int HW = a.H * a.W;
  const int b = blockIdx.y;
  const int nvec = C >> 3;                       // threads per pixel (C <= 2048)
  const int ppi = 256 / nvec;                    // pixels per block iteration
  const int tid = threadIdx.x;
  if (tid >= ppi * nvec) return;
  const int v = tid % nvec, po = tid / nvec;
  const int c = v * 8;
  const int cpg = C / a.G;
  const T* src; int ld, cc;
  if (c < a.C0) { src = (const T*)a.x0; ld = a.C0; cc = c; } else { src = (const T*)a.x1; ld = a.C1; cc = c - a.C0; }
  src += (long)b * HW * ld + cc;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float* st = a.stats + ((long)b * a.G + (c + k) / cpg) * 2;
    sc[k] = st[1] * a.gamma[c + k];
    sh[k] = a.beta[c + k] - st[0] * sc[k];
  }
  T* out = (T*)a.out + (long)b * HW * C + c;
  const int p_lo = blockIdx.x * pix_per_block, p_hi = min(HW, p_lo + pix_per_block);
  auto apply = [&](u4 raw, int p) {
    union { u4 u; T e[8]; } in, o;
    in.u = raw;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float y = to_f32(in.e[k]) * sc[k] + sh[k];
      if (a.silu) y = silu_fast(y);
      o.e[k] = from_f32<T>(y);
    }
    *(u4*)(out + (long)p * C) = o.u;
  };
  int p = p_lo + po;
  for (; p + 3 * ppi < p_hi; p += 4 * ppi) {
    const u4 r0 = *(const u4*)(src + (long)p * ld), r1 = *(const u4*)(src + (long)(p + ppi) * ld);
    const u4 r2 = *(const u4*)(src + (long)(p + 2 * ppi) * ld), r3 = *(const u4*)(src + (long)(p + 3 * ppi) * ld);
    apply(r0, p); apply(r1, p + ppi); apply(r2, p + 2 * ppi); apply(r3, p + 3 * ppi);
  }
  for (; p < p_hi; p += ppi) apply(*(const u4*)(src + (long)p * ld), p);
}

// Finalize + apply in one launch for maps of <= 4096 pixels (their GroupNorm is two latency-bound launches otherwise): grid
// (pixel parts, C / 64, B).  A block folds the column sums of ITS 64 channels only ([HW / 64 chunks][64][2] floats, <= 32 KiB,
// L2-resident: the producing epilogue has just written them) in double precision and a fixed order, derives mean / rstd of the
// slab's 64 / cpg groups, then normalises its share of the sample's pixels like gn_apply16_kernel.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_cols_kernel(GroupNormApplyArgs a, int pix_per_block) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __shared__ double red[2][4][64];
  __shared__ float gstat[2][64];
  const int C = a.C0 + a.C1, HW = a.H * a.W, nchunk = HW >> 6, cpg = C / a.G;
  const int b = blockIdx.z, c0 = blockIdx.y * 64, tid = threadIdx.x;
  const bool second = c0 >= a.C0;                      // C0 % 64 == 0: a slab lies in one source
  {
    const float* cs = second ? a.cs1 : a.cs0;
    const int Cs = second ? a.C1 : a.C0, cc = (second ? c0 - a.C0 : c0) + (tid & 63);
    double s = 0, q = 0;
    for (int ch = tid >> 6; ch < nchunk; ch += 4) {
      const float2 v = *(const float2*)(cs + ((long)(b * nchunk + ch) * Cs + cc) * 2);
      s += v.x; q += v.y;
    }
    red[0][tid >> 6][tid & 63] = s; red[1][tid >> 6][tid & 63] = q;
  }
  __syncthreads();
  if (tid < 64) {                                      // one lane per channel of the slab: its 4 partials, then the cpg channels of a
    double s = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);   // group through a fixed shuffle tree (cpg is a
    double q = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);   // power of two dividing 64): no serial chain
    for (int sh = 1; sh < cpg; sh <<= 1) { s += __shfl_xor(s, sh, 64); q += __shfl_xor(q, sh, 64); }
    if ((tid & (cpg - 1)) == 0) {
      const double n = (double)HW * cpg, mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0) var = 0;
      gstat[0][tid / cpg] = (float)mean;
      gstat[1][tid / cpg] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
  }
  __syncthreads();
  const int v = tid & 7, po = tid >> 3;                // 8 threads x 8 channels per pixel, 32 pixels per iteration
  const int c = c0 + v * 8;
  const T* src = second ? (const T*)a.x1 + (long)b * HW * a.C1 + (c - a.C0) : (const T*)a.x0 + (long)b * HW * a.C0 + c;
  const int ld = second ? a.C1 : a.C0;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int g = (v * 8 + k) / cpg;
    sc[k] = gstat[1][g] * a.gamma[c + k];
    sh[k] = a.beta[c + k] - gstat[0][g] * sc[k];
  }
  T* out = (T*)a.out + (long)b * HW * C + c;
  const int p_lo = blockIdx.x * pix_per_block, p_hi = min(HW, p_lo + pix_per_block);
  auto apply = [&](u4 raw, int p) {
    union { u4 u; T e[8]; } in, o;
    in.u = raw;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float y = to_f32(in.e[k]) * sc[k] + sh[k];
      if (a.silu) y = silu_fast(y);
      o.e[k] = from_f32<T>(y);
    }
    *(u4*)(out + (long)p * C) = o.u;
  };
  int p = p_lo + po;
  for (; p + 96 < p_hi; p += 128) {
    const u4 r0 = *(const u4*)(src + (long)p * ld), r1 = *(const u4*)(src + (long)(p + 32) * ld);
    const u4 r2 = *(const u4*)(src + (long)(p + 64) * ld), r3 = *(const u4*)(src + (long)(p + 96) * ld);
    apply(r0, p); apply(r1, p + 32); apply(r2, p + 64); apply(r3, p + 96);
  }
  for (; p < p_hi; p += 32) apply(*(const u4*)(src + (long)p * ld), p);
}

bool g_gn_apply_cols = true;       // plan switch 27
bool gn_apply_cols_eligible(const GroupNormApplyArgs& a) {
  const int C = a.C0 + a.C1, HW = a.H * a.W;
  if (!g_gn_apply_cols || !a.cs0 || (a.C1 > 0) != (a.cs1 != nullptr) || a.dtype == DT_F32 || !a.x0_lowp || a.down || a.raw_out) return false;
  if (a.C0 % 64 != 0 || a.C1 % 64 != 0 || C % a.G != 0 || 64 % (C / a.G) != 0) return false;
  return HW % 64 == 0 && HW <= 4096;
}
int launch_gn_apply_cols(const GroupNormApplyArgs& a, hipStream_t s) {
  T2P_REQUIRE(gn_apply_cols_eligible(a) && a.x0 && a.gamma && a.beta && a.out, "gn_apply_cols arguments");
  const int C = a.C0 + a.C1, HW = a.H * a.W;
  // ~1024 blocks per launch; a block's pixel share is a multiple of the 32 pixels of one iteration
  int parts = std::max(1, 1024 / (a.B * (C / 64)));
  int ppb = std::max(32, ((HW + parts - 1) / parts + 31) / 32 * 32);
  dim3 grid((HW + ppb - 1) / ppb, C / 64, a.B);
  if (a.dtype == DT_BF16) hipLaunchKernelGGL((gn_apply_cols_kernel<bf16_t>), grid, dim3(256), 0, s, a, ppb);
  else hipLaunchKernelGGL((gn_apply_cols_kernel<f16_t>), grid, dim3(256), 0, s, a, ppb);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

static inline int ew_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

int launch_gn_apply(const GroupNormApplyArgs& a, hipStream_t s) {
  const int C = a.C0 + a.C1;
  T2P_REQUIRE(a.x0 && a.stats && a.gamma && a.beta && a.out, "null pointer");
  T2P_REQUIRE(a.C0 % 4 == 0 && a.C1 % 4 == 0 && C % a.G == 0, "channel constraints");
  T2P_REQUIRE(!a.down || (a.H % 2 == 0 && a.W % 2 == 0), "down-sampling needs even H, W");
  T2P_REQUIRE(!(a.down && a.raw_out), "raw copy is not produced together with down-sampling");
  const int HWo = (a.down ? a.H / 2 : a.H) * (a.down ? a.W / 2 : a.W);
  // pixels per block: 64 on large maps; on small maps fewer, so that the launch still has ~2048 blocks
  // (a 16x16 map with 64 pixels per block is 128 blocks of 32 serial iterations: latency-bound)
  const int zb = (C + 1023) / 1024;
  const int ppi = 256 / (std::min(C, 1024) / 4);                       // pixels a block covers per iteration
  long want = ((long)HWo * a.B * zb + 2047) / 2048;
  int ppb = (int)std::min<long>(GNA_PIX_PER_BLOCK, std::max<long>(want, std::max(ppi, 1)));
  if (ppi > 1) ppb = (ppb + ppi - 1) / ppi * ppi;
  dim3 grid((HWo + ppb - 1) / ppb, a.B, zb);
  T2P_REQUIRE(!a.x0_lowp || a.dtype != DT_F32, "16-bit GroupNorm input needs a 16-bit dtype (both sources are then 16-bit)");
  if (g_gn_apply16 && a.x0_lowp && !a.down && !a.raw_out && a.C0 % 8 == 0 && a.C1 % 8 == 0 && C >= 64 && C <= 2048 && 256 % (C / 8) == 0) {
    const int ppi8 = 256 / (C / 8);
    long want8 = ((long)HWo * a.B + 2047) / 2048;
    int ppb8 = (int)std::min<long>(GNA_PIX_PER_BLOCK, std::max<long>(want8, ppi8));
    ppb8 = (ppb8 + ppi8 - 1) / ppi8 * ppi8;
    dim3 grid8((HWo + ppb8 - 1) / ppb8, a.B, 1);
    if (a.dtype == DT_BF16) hipLaunchKernelGGL((gn_apply16_kernel<bf16_t>), grid8, dim3(256), 0, s, a, ppb8);
    else hipLaunchKernelGGL((gn_apply16_kernel<f16_t>), grid8, dim3(256), 0, s, a, ppb8);
    T2P_HIP_CHECK(hipGetLastError());
    return T2P_OK;
  }
  switch (a.dtype) {
    case DT_F32: hipLaunchKernelGGL((gn_apply_kernel<float, float>), grid, dim3(256), 0, s, a, ppb); break;
    case DT_BF16:
      if (a.x0_lowp) hipLaunchKernelGGL((gn_apply_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, a, ppb);
      else hipLaunchKernelGGL((gn_apply_kernel<bf16_t, float>), grid, dim3(256), 0, s, a, ppb);
      break;
    case DT_F16:
      if (a.x0_lowp) hipLaunchKernelGGL((gn_apply_kernel<f16_t, f16_t>), grid, dim3(256), 0, s, a, ppb);
      else hipLaunchKernelGGL((gn_apply_kernel<f16_t, float>), grid, dim3(256), 0, s, a, ppb);
      break;
    default: set_last_error("gn_apply: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== GroupNorm of a small map, one launch ==========================
// At the 4x4 / 8x8 levels a GroupNorm is three launch-latency-bound kernels (statistics, finalize,
// apply: ~4.5 us each for a few KiB of data).  Here one 256-thread block owns up to 1024 channels of
// one sample: pass 1 accumulates per-thread sums of its 4 channels over its pixels, the threads of a
// group are folded in a fixed order through LDS (double, like gn_finalize), pass 2 re-reads the
// (L2-resident) map and writes the normalised / activated result.  Same thread -> (channel, pixel)
// mapping as gn_apply_kernel.
struct GnSmallArgs : GroupNormApplyArgs { int cb = 1024; };   // cb: channels per block (whole groups)

template <typename TO, typename TI>
__global__ __launch_bounds__(256) void gn_small_kernel(GnSmallArgs a) {
  __shared__ float part[256][8];
  const int C = a.C0 + a.C1, HW = a.H * a.W;
  const int b = blockIdx.y, c_lo = blockIdx.x * a.cb;
  const int nvec = min(C - c_lo, a.cb) >> 2;
  const int ppi = 256 / nvec;
  const int tid = threadIdx.x;
  const bool active = tid < ppi * nvec;
  const int v = active ? tid % nvec : 0, po = active ? tid / nvec : 0;
  const int c = c_lo + v * 4;
  const int cpg = C / a.G;
  const TI* src; int ld, cc;
  if (c < a.C0) { src = (const TI*)a.x0; ld = a.C0; cc = c; } else { src = (const TI*)a.x1; ld = a.C1; cc = c - a.C0; }
  src += (long)b * HW * ld + cc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  if (active)
    for (int p = po; p < HW; p += ppi) {
      const float4 t = load4<TI>(src + (long)p * ld);
      s0 += t.x; s1 += t.y; s2 += t.z; s3 += t.w;
      q0 += t.x * t.x; q1 += t.y * t.y; q2 += t.z * t.z; q3 += t.w * t.w;
    }
  part[tid][0] = s0; part[tid][1] = s1; part[tid][2] = s2; part[tid][3] = s3;
  part[tid][4] = q0; part[tid][5] = q1; part[tid][6] = q2; part[tid][7] = q3;
  __syncthreads();
  if (!active) return;
  // the group of this thread's 4 channels: channel vectors [v_lo, v_lo + cpg / 4) of every pixel lane
  const int v_lo = ((c / cpg) * cpg - c_lo) >> 2, nv = cpg >> 2;
  double s = 0, q = 0;
  for (int l = 0; l < ppi; ++l)
    for (int k = 0; k < nv; ++k) {
      const float* e = part[l * nvec + v_lo + k];
      s += (double)e[0] + (double)e[1] + (double)e[2] + (double)e[3];
      q += (double)e[4] + (double)e[5] + (double)e[6] + (double)e[7];
    }
  const double n = (double)HW * cpg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  const float fmean = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float4 ga = *(const float4*)(a.gamma + c), be = *(const float4*)(a.beta + c);
  const float sc0 = rstd * ga.x, sc1 = rstd * ga.y, sc2 = rstd * ga.z, sc3 = rstd * ga.w;
  const float sh0 = be.x - fmean * sc0, sh1 = be.y - fmean * sc1, sh2 = be.z - fmean * sc2, sh3 = be.w - fmean * sc3;
  TO* out = (TO*)a.out + (long)b * HW * C + c;
  TO* raw = a.raw_out ? (TO*)a.raw_out + (long)b * HW * C + c : nullptr;
  for (int p = po; p < HW; p += ppi) {
    const float4 t = load4<TI>(src + (long)p * ld);
    if (raw) store4<TO>(raw + (long)p * C, t.x, t.y, t.z, t.w);
    float y0 = t.x * sc0 + sh0, y1 = t.y * sc1 + sh1, y2 = t.z * sc2 + sh2, y3 = t.w * sc3 + sh3;
    if (a.silu) { y0 = silu_fast(y0); y1 = silu_fast(y1); y2 = silu_fast(y2); y3 = silu_fast(y3); }
    store4<TO>(out + (long)p * C, y0, y1, y2, y3);
  }
}

// channels per block: whole groups, about 128 (more blocks = shorter serial loops; 32 blocks of 1024
// channels took 18 us on a few KiB)
static int gn_small_cb(int C, int cpg) {
  if (C <= 128) return C;
  return (128 % cpg == 0) ? 128 : ((128 + cpg - 1) / cpg) * cpg;
}

bool gn_small_eligible(const GroupNormApplyArgs& a) {
  const int C = a.C0 + a.C1;
  if (a.down || a.G <= 0 || C % a.G != 0 || a.H * a.W > 64) return false;
  const int cpg = C / a.G;
  if (cpg % 4 != 0 || a.C0 % 4 != 0 || a.C1 % 4 != 0) return false;
  return gn_small_cb(C, cpg) <= 1024;
}

int launch_gn_small(const GroupNormApplyArgs& a, hipStream_t s) {
  const int C = a.C0 + a.C1;
  T2P_REQUIRE(a.x0 && a.gamma && a.beta && a.out, "null pointer");
  T2P_REQUIRE(gn_small_eligible(a), "gn_small: not eligible");
  T2P_REQUIRE(!a.x0_lowp || a.dtype != DT_F32, "16-bit GroupNorm input needs a 16-bit dtype (both sources are then 16-bit)");
  GnSmallArgs g;
  (GroupNormApplyArgs&)g = a;
  g.cb = gn_small_cb(C, C / a.G);
  dim3 grid((C + g.cb - 1) / g.cb, a.B);
  switch (a.dtype) {
    case DT_F32: hipLaunchKernelGGL((gn_small_kernel<float, float>), grid, dim3(256), 0, s, g); break;
    case DT_BF16:
      if (a.x0_lowp) hipLaunchKernelGGL((gn_small_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gn_small_kernel<bf16_t, float>), grid, dim3(256), 0, s, g);
      break;
    case DT_F16:
      if (a.x0_lowp) hipLaunchKernelGGL((gn_small_kernel<f16_t, f16_t>), grid, dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gn_small_kernel<f16_t, float>), grid, dim3(256), 0, s, g);
      break;
    default: set_last_error("gn_small: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== LayerNorm ====================================================
// one wavefront per row; three passes over an L1/L2-resident row (C <= a few thousand).
template <typename TO, typename TI>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* x, const float* gamma, const float* beta,
                                                        TO* out, long rows, int C, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const TI* xr = x + row * C;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    float4 t = load4<TI>(xr + c);
    s += (t.x + t.y) + (t.z + t.w);
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    float4 t = load4<TI>(xr + c);
    float a0 = t.x - mean, a1 = t.y - mean, a2 = t.z - mean, a3 = t.w - mean;
    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / C + eps);
  for (int c = lane * 4; c < C; c += 256) {
    float4 t = load4<TI>(xr + c);
    float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    store4<TO>(out + row * C + c, (t.x - mean) * rstd * g.x + b.x, (t.y - mean) * rstd * g.y + b.y,
               (t.z - mean) * rstd * g.z + b.z, (t.w - mean) * rstd * g.w + b.w);
  }
}

// 16-bit in / out, C = 512 NV (512 or 1024 channels: the transformer widths at nf = 256): a wavefront owns a row, a lane
// keeps its 8 NV elements in registers (one 16-byte load each), so the row is read once; two-pass mean / variance as above.
bool g_layernorm16 = true;      // plan switch 20
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm16_kernel(const T* x, const float* gamma, const float* beta, T* out, long rows, float eps) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  constexpr int C = 512 * NV;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const T* xr = x + row * C + lane * 8;
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    union { u4 u; T e[8]; } in;
    in.u = *(const u4*)(xr + 512 * k);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[k][j] = to_f32(in.e[j]);
    s += ((v[k][0] + v[k][1]) + (v[k][2] + v[k][3])) + ((v[k][4] + v[k][5]) + (v[k][6] + v[k][7]));
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[k][j] -= mean; }
    q += ((v[k][0] * v[k][0] + v[k][1] * v[k][1]) + (v[k][2] * v[k][2] + v[k][3] * v[k][3])) +
         ((v[k][4] * v[k][4] + v[k][5] * v[k][5]) + (v[k][6] * v[k][6] + v[k][7] * v[k][7]));
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / C + eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = 512 * k + lane * 8;
    const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
    const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    union { u4 u; T e[8]; } o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.e[j] = from_f32<T>(v[k][j] * rstd * gg[j] + bb[j]);
    *(u4*)(out + row * C + c) = o.u;
  }
}

int launch_layernorm(const float* x, const float* gamma, const float* beta, void* out, int dtype, long rows, int C,
                     float eps, hipStream_t s, int x_lowp) {
  T2P_REQUIRE(x && gamma && beta && out && C % 4 == 0 && rows > 0, "layernorm arguments");
  T2P_REQUIRE(!x_lowp || dtype != DT_F32, "16-bit LayerNorm input needs a 16-bit dtype");
  dim3 grid((unsigned)((rows + 3) / 4));
  if (g_layernorm16 && x_lowp && dtype != DT_F32 && (C == 512 || C == 1024)) {
    if (dtype == DT_F16) {
      if (C == 512) hipLaunchKernelGGL((layernorm16_kernel<f16_t, 1>), grid, dim3(256), 0, s, (const f16_t*)x, gamma, beta, (f16_t*)out, rows, eps);
      else hipLaunchKernelGGL((layernorm16_kernel<f16_t, 2>), grid, dim3(256), 0, s, (const f16_t*)x, gamma, beta, (f16_t*)out, rows, eps);
    } else {
      if (C == 512) hipLaunchKernelGGL((layernorm16_kernel<bf16_t, 1>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, eps);
      else hipLaunchKernelGGL((layernorm16_kernel<bf16_t, 2>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, eps);
    }
    T2P_HIP_CHECK(hipGetLastError());
    return T2P_OK;
  }
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, dim3(256), 0, s, x, gamma, beta, (float*)out, rows, C, eps); break;
    case DT_BF16:
      if (x_lowp) hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)out, rows, C, eps);
      else hipLaunchKernelGGL((layernorm_kernel<bf16_t, float>), grid, dim3(256), 0, s, x, gamma, beta, (bf16_t*)out, rows, C, eps);
      break;
    case DT_F16:
      if (x_lowp) hipLaunchKernelGGL((layernorm_kernel<f16_t, f16_t>), grid, dim3(256), 0, s, (const f16_t*)x, gamma, beta, (f16_t*)out, rows, C, eps);
      else hipLaunchKernelGGL((layernorm_kernel<f16_t, float>), grid, dim3(256), 0, s, x, gamma, beta, (f16_t*)out, rows, C, eps);
      break;
    default: set_last_error("layernorm: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== row softmax ====================================================
// one wavefront per row; the row (<= 64 * SM_REGS elements) is held in registers, reductions by
// wavefront shuffles.  Longer rows take the re-reading path.
static constexpr int SM_REGS = 16;

template <typename TO>
__global__ __launch_bounds__(256) void softmax_kernel(const float* S, long lds, TO* P, long ldp, long rows, int n,
                                                      float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* sr = S + row * lds;
  TO* pr = P + row * ldp;
  if (n <= 64 * SM_REGS && (n & 3) == 0 && (lds & 3) == 0 && (ldp & 3) == 0) {
    // 16-byte loads, 8/16-byte stores: lane owns columns 4*lane + 256*i .. +3
    float v[SM_REGS];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_REGS / 4; ++i) {
      const int c = lane * 4 + 256 * i;
      float4 t = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (c < n) { t = *(const float4*)(sr + c); t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale; }
      v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
      m = fmaxf(m, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_REGS; ++i) {
      v[i] = expf(v[i] - m);       // exp(-inf) = 0 for the columns beyond n
      sum += v[i];
    }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < SM_REGS / 4; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < ldp) store4<TO>(pr + c, v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv);
    }
  } else if (n <= 64 * SM_REGS) {
    float v[SM_REGS];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_REGS; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < n ? sr[c] * scale : -INFINITY;
      m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_REGS; ++i) {
      v[i] = (lane + 64 * i) < n ? expf(v[i] - m) : 0.f;
      sum += v[i];
    }
    const float inv = 1.f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < SM_REGS; ++i) {
      const int c = lane + 64 * i;
      if (c < ldp) pr[c] = from_f32<TO>(v[i] * inv);   // zero beyond n (v == 0 there)
    }
  } else {
    float m = -INFINITY;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, sr[c] * scale);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < n; c += 64) sum += expf(sr[c] * scale - m);
    const float inv = 1.f / wave_sum(sum);
    for (int c = lane; c < ldp; c += 64) pr[c] = from_f32<TO>(c < n ? expf(sr[c] * scale - m) * inv : 0.f);
  }
}

int launch_softmax(const float* S, long lds, void* P, long ldp, int dtype, long rows, int n, float scale,
                   hipStream_t s) {
  T2P_REQUIRE(S && P && rows > 0 && n > 0 && lds >= n && ldp >= n, "softmax arguments");
  T2P_REQUIRE(ldp <= 64 * SM_REGS || n > 64 * SM_REGS, "padded row longer than the register path covers");
  dim3 grid((unsigned)((rows + 3) / 4));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL(softmax_kernel<float>, grid, dim3(256), 0, s, S, lds, (float*)P, ldp, rows, n, scale); break;
    case DT_BF16: hipLaunchKernelGGL(softmax_kernel<bf16_t>, grid, dim3(256), 0, s, S, lds, (bf16_t*)P, ldp, rows, n, scale); break;
    case DT_F16: hipLaunchKernelGGL(softmax_kernel<f16_t>, grid, dim3(256), 0, s, S, lds, (f16_t*)P, ldp, rows, n, scale); break;
    default: set_last_error("softmax: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== GEGLU ============================================================
template <typename TO>
__global__ __launch_bounds__(256) void geglu_kernel(const float* u, TO* out, long rows, int inner, int interleaved) {
  const int nvec = inner >> 2;
  const long total = rows * nvec;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / nvec;
    const int j = (int)(idx - r * nvec) * 4;
    float4 a, g;
    if (interleaved) {   // (value_j, gate_j) pairs: the layout of the fused-epilogue weights
      const float4 p0 = *(const float4*)(u + r * 2 * inner + 2 * j), p1 = *(const float4*)(u + r * 2 * inner + 2 * j + 4);
      a = make_float4(p0.x, p0.z, p1.x, p1.z);
      g = make_float4(p0.y, p0.w, p1.y, p1.w);
    } else {
      a = *(const float4*)(u + r * 2 * inner + j);
      g = *(const float4*)(u + r * 2 * inner + inner + j);
    }
    auto gelu = [](float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); };
    store4<TO>(out + r * inner + j, a.x * gelu(g.x), a.y * gelu(g.y), a.z * gelu(g.z), a.w * gelu(g.w));
  }
}

int launch_geglu(const float* u, void* out, int dtype, long rows, int inner, hipStream_t s, int interleaved) {
  T2P_REQUIRE(u && out && inner % 4 == 0 && rows > 0, "geglu arguments");
  dim3 grid(ew_grid(rows * (inner / 4)));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL(geglu_kernel<float>, grid, dim3(256), 0, s, u, (float*)out, rows, inner, interleaved); break;
    case DT_BF16: hipLaunchKernelGGL(geglu_kernel<bf16_t>, grid, dim3(256), 0, s, u, (bf16_t*)out, rows, inner, interleaved); break;
    case DT_F16: hipLaunchKernelGGL(geglu_kernel<f16_t>, grid, dim3(256), 0, s, u, (f16_t*)out, rows, inner, interleaved); break;
    default: set_last_error("geglu: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== 2x2 mean pooling ===================================================
template <typename TO, typename TI>
__global__ __launch_bounds__(256) void pool2x2_kernel(const TI* x, TO* out, int B, int H, int W, int C) {
  const int nvec = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long total = (long)B * Ho * Wo * nvec;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % nvec);
    const long pix = idx / nvec;
    const int b = (int)(pix / (Ho * Wo));
    const int rem = (int)(pix - (long)b * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const TI* p = x + (((long)b * H + 2 * oy) * W + 2 * ox) * C + v * 4;
    const float4 a = load4<TI>(p), b1 = load4<TI>(p + C);
    const float4 c = load4<TI>(p + (long)W * C), d = load4<TI>(p + (long)W * C + C);
    // torch.mean over the (2, 2) window: sum in row-major window order, then divide
    store4<TO>(out + pix * C + v * 4, (a.x + b1.x + c.x + d.x) * 0.25f, (a.y + b1.y + c.y + d.y) * 0.25f,
               (a.z + b1.z + c.z + d.z) * 0.25f, (a.w + b1.w + c.w + d.w) * 0.25f);
  }
}

int launch_pool2x2(const float* x, void* out, int dtype, int B, int H, int W, int C, hipStream_t s, int x_lowp) {
  T2P_REQUIRE(x && out && C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "pool2x2 arguments");
  T2P_REQUIRE(!x_lowp || dtype != DT_F32, "16-bit pooling input needs a 16-bit dtype");
  dim3 grid(ew_grid((long)B * (H / 2) * (W / 2) * (C / 4)));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL((pool2x2_kernel<float, float>), grid, dim3(256), 0, s, x, (float*)out, B, H, W, C); break;
    case DT_BF16:
      if (x_lowp) hipLaunchKernelGGL((pool2x2_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, C);
      else hipLaunchKernelGGL((pool2x2_kernel<bf16_t, float>), grid, dim3(256), 0, s, x, (bf16_t*)out, B, H, W, C);
      break;
    case DT_F16:
      if (x_lowp) hipLaunchKernelGGL((pool2x2_kernel<f16_t, f16_t>), grid, dim3(256), 0, s, (const f16_t*)x, (f16_t*)out, B, H, W, C);
      else hipLaunchKernelGGL((pool2x2_kernel<f16_t, float>), grid, dim3(256), 0, s, x, (f16_t*)out, B, H, W, C);
      break;
    default: set_last_error("pool2x2: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== input convolution ====================================================
// pre_conv (reference ncsnpp.py:137,230): 3x3, C = 5 or 8 input channels (NCHW fp32, the sampler
// state with the dynamic range of sigma_max) -> nf channels NHWC fp32, exact fp32 FMAs.  Too thin
// for the MFMA GEMM (K = 45): one workgroup takes a 32-pixel row segment, stages the 3 x 34 x C
// input patch in LDS, and each thread owns one output channel with its 9 C weights in registers;
// lanes cover consecutive channels, so the NHWC stores are coalesced.
static constexpr int PRE_ROWS = 4;   // image rows per block: the per-thread weights are loaded once for all of them

template <int C, typename TO>
__global__ __launch_bounds__(256) void pre_conv_kernel(const float* x, const float* w, const float* bias, TO* out, int B,
                                                       int H, int W, int nf, float* cstats) {
  constexpr int SEG = 32, ROW = 40;                    // patch rows padded to 16-byte multiples (34 used + 6)
  __shared__ __attribute__((aligned(16))) float patch[C][3][ROW];
  __shared__ float red[2][256];                        // statistics of the pixel groups of one channel (nf < 256)
  // cstats (optional; W % 64 == 0, nf <= 256): GroupNorm column statistics of the output in the layout of the GEMM epilogues,
  // [B H W / 64][nf][2] = per 64-pixel chunk and channel (sum, sum of squares) of the fp32 values.  A block then walks the two
  // 32-pixel segments of a chunk one after the other and its threads carry the sums across them.
  const int nhalf = cstats ? 2 : 1;
  const int segs = (W + SEG - 1) / SEG / nhalf, rblks = (H + PRE_ROWS - 1) / PRE_ROWS;
  const int seg = blockIdx.x % segs, yb = (blockIdx.x / segs) % rblks, b = blockIdx.x / (segs * rblks);
  // a thread owns one output channel (its 9 C weights in registers; w is [tap][c][nf], so a wavefront reads each
  // of them as one contiguous row) and 4 consecutive pixels at a time: the 3-tap window of 4 pixels is 6
  // consecutive patch values = two 16-byte LDS reads for 12 FMAs (a read per FMA made the kernel LDS-issue-bound).
  // With nf <= 128 the block splits the segment into pixel groups so that all 256 threads work.
  const int ngrp = nf >= 256 ? 1 : (256 / nf >= 8 ? 8 : 256 / nf);
  const int pg = SEG / ngrp;                           // pixels per group: 32, 16, 8 or 4
  for (int co0 = 0; co0 < nf; co0 += 256) {
    const int co = co0 + (ngrp == 1 ? (int)threadIdx.x : (int)threadIdx.x % nf);
    const int grp = ngrp == 1 ? 0 : (int)threadIdx.x / nf;
    const bool work = co < nf && grp < ngrp;
    float wr[C * 9];
#pragma unroll
    for (int i = 0; i < C * 9; ++i) wr[i] = work ? w[(long)i * nf + co] : 0.f;
    const float bv = work ? bias[co] : 0.f;
    for (int y = yb * PRE_ROWS; y < min(H, (yb + 1) * PRE_ROWS); ++y) {
     float ssum = 0.f, ssq = 0.f;
     for (int half = 0; half < nhalf; ++half) {
      const int x0 = (seg * nhalf + half) * SEG;
      const int npx = min(SEG, W - x0);
      __syncthreads();                                 // the previous row's patch is no longer read
      for (int i = threadIdx.x; i < C * 3 * ROW; i += 256) {
        const int c = i / (3 * ROW), r = (i / ROW) % 3, col = i % ROW;
        const int sy = y + r - 1, sx = x0 + col - 1;
        patch[c][r][col] = (col < SEG + 2 && sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[(((long)b * C + c) * H + sy) * W + sx] : 0.f;
      }
      __syncthreads();
      if (!work) continue;
      for (int px = grp * pg; px < (grp + 1) * pg && px < npx; px += 4) {
        float a0 = bv, a1 = bv, a2 = bv, a3 = bv;
#pragma unroll
        for (int t = 0; t < 9; ++t) {                  // tap-major, channel-minor: the summation order of a plain loop
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float4 p0 = *(const float4*)&patch[c][t / 3][px], p1 = *(const float4*)&patch[c][t / 3][px + 4];
            const float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            const float wv = wr[t * C + c];
            a0 = fmaf(v[t % 3], wv, a0); a1 = fmaf(v[t % 3 + 1], wv, a1);
            a2 = fmaf(v[t % 3 + 2], wv, a2); a3 = fmaf(v[t % 3 + 3], wv, a3);
          }
        }
        TO* o = out + (((long)b * H + y) * W + x0 + px) * nf + co;
        o[0] = from_f32<TO>(a0);
        if (px + 1 < npx) o[(long)nf] = from_f32<TO>(a1);
        if (px + 2 < npx) o[2L * nf] = from_f32<TO>(a2);
        if (px + 3 < npx) o[3L * nf] = from_f32<TO>(a3);
        ssum += (a0 + a1) + (a2 + a3);                 // (cstats: W % 64 == 0, so all four pixels exist)
        ssq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
     }
     if (cstats) {
       // fold the pixel groups of a channel in a fixed order (bitwise reproducible), one store per channel
       __syncthreads();
       red[0][threadIdx.x] = ssum; red[1][threadIdx.x] = ssq;
       __syncthreads();
       if (work && grp == 0) {
         float t0 = 0.f, t1 = 0.f;
         for (int g = 0; g < ngrp; ++g) { t0 += red[0][g * nf + co - co0]; t1 += red[1][g * nf + co - co0]; }
         const long chunk = (((long)b * H + y) * W + seg * 64) >> 6;
         *(float2*)(cstats + (chunk * nf + co) * 2) = make_float2(t0, t1);
       }
     }
    }
  }
}

// ---- the input convolution on the fp32 matrix pipe (16-bit modes) ---------------------------------------------------
// out[pixel][channel] = sum_k patch[pixel][k] w[k][channel], k = tap * C + c (K = 45 or 72), as v_mfma_f32_32x32x2_f32:
// fp32 operands and accumulation like the FMA kernel above (another summation order: 1e-7), at the matrix pipe's fp32 rate
// instead of scalar FMAs.  A wave owns one image row of a 4-row x 64-pixel block tile: two wave tiles of 32 pixels x 32 NT
// channels (NT = 4 accumulator tiles; 256 channels are two passes).  Operands per k-step (k = 2 s + (lane >> 5)):
//   A: lane's pixel (lane & 31), value patch[c][row + dy][px + dx] from the block's LDS patch (6 input rows x 66 columns);
//   B: channel (lane & 31) * NT + j for accumulator tile j -- the NT channels of a lane are CONSECUTIVE, so w[k] is read as
//      NT contiguous floats and a pixel's output row is stored in 2 NT-byte pieces, 32 lanes covering the whole row.
// The weights ([K padded to even][nf] fp32, 47 KiB at nf = 256) stay in LDS while the block walks 16 image rows.
// GroupNorm column statistics: a wave's two tiles are one 64-pixel chunk; per-lane sums over its 16 + 16 pixel rows, the two
// lane halves folded by one shuffle, lanes 0 .. 31 store them.
typedef float pf32x16 __attribute__((ext_vector_type(16)));
static constexpr int PREM_ROWS = 16;     // image rows per block

template <int C, int NT, typename TO>
__global__ __launch_bounds__(256, 2) void pre_conv_mfma_kernel(const float* x, const float* w, const float* bias, TO* out, int B, int H,
                                                              int W, int nf, float* cstats) {
  constexpr int K = 9 * C, KP = (K + 1) & ~1, NP = 32 * NT, PW = 66, PR = 6;      // NP: channels of one pass over the K loop
  extern __shared__ __attribute__((aligned(16))) float psm[];
  float* wl = psm;                          // [KP][nf]
  float* patch = psm + KP * nf;             // [C][PR][PW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, kh = lane >> 5;
  const int segs = W / 64, rblks = (H + PREM_ROWS - 1) / PREM_ROWS;
  const int seg = blockIdx.x % segs, rb = (blockIdx.x / segs) % rblks, b = blockIdx.x / (segs * rblks);
  const int x0 = seg * 64;
  for (int i = tid; i < KP * nf; i += 256) wl[i] = i < K * nf ? w[i] : 0.f;
  for (int y0 = rb * PREM_ROWS; y0 < min(H, (rb + 1) * PREM_ROWS); y0 += 4) {
    __syncthreads();                        // the previous rows' patch is no longer read (and, the first time, wl is complete below)
    for (int i = tid; i < C * PR * PW; i += 256) {
      const int c = i / (PR * PW), r = (i / PW) % PR, col = i % PW;
      const int sy = y0 + r - 1, sx = x0 + col - 1;
      patch[i] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[(((long)b * C + c) * H + sy) * W + sx] : 0.f;
    }
    __syncthreads();
    const int y = y0 + wave;
    if (y >= H) continue;                   // wave-uniform
    // 256 channels are two passes of 128 (64 accumulator registers each: with all 128 the k loop's operands spill)
#pragma unroll 1
    for (int cb = 0; cb < nf; cb += NP) {
      const int ch0 = cb + l32 * NT;        // this lane's NT consecutive channels
      float bv[NT], ssum[NT], ssq[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) { bv[j] = bias[ch0 + j]; ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        pf32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
        const float* prow = patch + wave * PW + half * 32 + l32;     // input row y - 1 + dy is patch row wave + dy
        const float* wrow = wl + kh * nf + ch0;
        // operands of k-step s (k = 2 s + kh: both candidates are compile-time constants, the lane half selects)
        auto load = [&](int s, float& a, float (&bw)[NT]) __attribute__((always_inline)) {
          const int k0 = 2 * s, k1 = 2 * s + 1;
          const int o0 = k0 < K ? ((k0 % C) * PR + (k0 / C) / 3) * PW + (k0 / C) % 3 : 0;
          const int o1 = k1 < K ? ((k1 % C) * PR + (k1 / C) / 3) * PW + (k1 / C) % 3 : 0;
          a = prow[kh ? o1 : o0];
          if (k1 >= K && kh) a = 0.f;                                  // the padded k of an odd K
          const float* wr = wrow + 2 * s * nf;
          if constexpr (NT % 4 == 0) {
#pragma unroll
            for (int j = 0; j < NT; j += 4) { const float4 t = *(const float4*)(wr + j); bw[j] = t.x; bw[j + 1] = t.y; bw[j + 2] = t.z; bw[j + 3] = t.w; }
          } else {
#pragma unroll
            for (int j = 0; j < NT; j += 2) { const float2 t = *(const float2*)(wr + j); bw[j] = t.x; bw[j + 1] = t.y; }
          }
        };
        float a_n, bw_n[NT];
        load(0, a_n, bw_n);
#pragma unroll
        for (int s = 0; s < KP / 2; ++s) {
          const float a = a_n;
          float bw[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) bw[j] = bw_n[j];
          if (s + 1 < KP / 2) load(s + 1, a_n, bw_n);                 // one step ahead: its LDS latency under this step's MFMAs
          asm volatile("" ::: "memory");                              // (and no further: the compiler would hoist every step's reads)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[j], acc[j], 0, 0, 0);
        }
        // accumulator register v of tile j: pixel row (v >> 2) * 8 + kh * 4 + (v & 3), channel ch0 + j
        TO* obase = out + (((long)b * H + y) * W + x0 + half * 32) * nf + ch0;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int pr = (v >> 2) * 8 + kh * 4 + (v & 3);
          union { TO e[NT]; uint2 u2; unsigned u1; } o;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float t = acc[j][v] + bv[j];
            o.e[j] = from_f32<TO>(t);
            ssum[j] += t; ssq[j] += t * t;
          }
          TO* dst = obase + (long)pr * nf;
          if constexpr (NT == 4) *(uint2*)dst = o.u2;
          else *(unsigned*)dst = o.u1;
        }
      }
      if (cstats) {
        const long chunk = (((long)b * H + y) * W + x0) >> 6;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float s0 = ssum[j] + __shfl_xor(ssum[j], 32, 64), s1 = ssq[j] + __shfl_xor(ssq[j], 32, 64);
          if (kh == 0) *(float2*)(cstats + (chunk * nf + ch0 + j) * 2) = make_float2(s0, s1);
        }
      }
    }
  }
}

// ---- the input convolution on the 16-bit matrix pipe with fp32-class accuracy (16-bit modes) ------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit rate: the kernel above is bound by the matrix pipe (12 GFLOP at 157 TFLOP/s
// peak; 169 us at cfg2) although the layer only has to write its output once (268 MB: ~55 us).  Here every fp32 operand is
// split into two f16 terms, v = hi + lo' / 2048 with hi = f16(v) and lo' = f16((v - hi) * 2048) (the scaling keeps lo' a normal
// number; |v| < 2^-14 has hi = 0), and x * w ~ x_hi w_hi + (x_lo' w_hi + x_hi w_lo') / 2048 -- the dropped term and the two
// roundings are ~2^-22 relative, the class of the fp32 kernels.  The three partial products are K segments of ONE f16 GEMM with
// two accumulators (v_mfma_f32_16x16x32_f16):
//   A'[pixel][0 .. K) = x_hi,  [KH .. KH + K) = x_lo',  [KH + KE .. KH + KE + K) = x_hi        (KH = K rounded up to 32, KE to 2)
//   B'[chan ][0 .. K) = w_hi,  [KH .. KH + K) = w_hi,   [KH + KE .. KH + KE + K) = w_lo'       (pre_conv_split_weights_kernel)
//   out = acc_h (K-steps below KH) + acc_l / 2048 + bias
// A workgroup (4 wavefronts) walks tiles of 2 image rows x 64 pixels: the fp32 input patch (4 x 66 x C) -> LDS, the im2col rows
// A' built from it (two k per thread and store: conflict-free), then wave w multiplies all 128 pixels by ITS nf / 4 channels,
// whose B' fragments stay in registers for the whole launch.  A lane ends with 8 consecutive channels of a pixel per pair of
// column tiles (the channel order of the tiles is chosen for that): 16-byte stores, whole 64-byte segments per pixel and wave.
// GroupNorm column statistics: a 64-pixel chunk is the four row tiles of one image row of the tile (fp32 values, before rounding).
typedef _Float16 ph8 __attribute__((ext_vector_type(8)));
typedef float pf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float prow16_sum(float x) {        // sum over the 16 lanes of a DPP row, fixed order
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
  return x;
}
__host__ __device__ inline void split_f16(float v, _Float16& hi, _Float16& lo) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);      // outside the f16 range the operand saturates (finite, wrong) instead of turning the output into inf / NaN
  hi = fabsf(v) < 6.103515625e-05f ? (_Float16)0.f : (_Float16)v;
  lo = (_Float16)((v - (float)hi) * 2048.f);
}
template <int C> struct PreSplit {
  static constexpr int K = 9 * C, KE = (K + 1) & ~1, KH = (K + 31) & ~31, KP = KH + ((2 * KE + 31) & ~31);   // 45: 64 + 96; 72: 96 + 160
  static constexpr int NSH = KH / 32, NS = KP / 32;
  static constexpr int RS = KP * 2 + 16;                   // LDS bytes per pixel row of A' (an odd number of 16-byte units)
};
// channel of column c (0 .. 15) of column tile j of a wave that owns 16 NCT channels from `base`
template <int NCT> __host__ __device__ inline int presplit_chan(int base, int j, int c) {
  if (NCT == 1) return base + c;
  return base + 32 * (j >> 1) + 8 * (c >> 2) + 4 * (j & 1) + (c & 3);
}
// w [K][nf] fp32 (tap-major: k = tap * C + c) -> B' [nf][KP] f16
template <int C>
__global__ void pre_conv_split_weights_kernel(const float* w, _Float16* out, int nf) {
  typedef PreSplit<C> P;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nf * P::KP) return;
  const int ch = i / P::KP, kk = i - ch * P::KP;
  int k = -1, part = 0;
  if (kk < P::K) { k = kk; part = 0; }
  else if (kk >= P::KH && kk < P::KH + P::K) { k = kk - P::KH; part = 0; }
  else if (kk >= P::KH + P::KE && kk < P::KH + P::KE + P::K) { k = kk - P::KH - P::KE; part = 1; }
  _Float16 hi = (_Float16)0.f, lo = (_Float16)0.f;
  if (k >= 0) split_f16(w[(long)k * nf + ch], hi, lo);
  out[i] = k < 0 ? (_Float16)0.f : (part ? lo : hi);
}

template <int C, int NCT, typename TO>
__global__ __launch_bounds__(256) void pre_conv_split_kernel(const float* x, const _Float16* wsplit, const float* bias, TO* out, int B, int H,
                                                             int W, int nf, float* cstats, int ntiles) {
  typedef PreSplit<C> P;
  constexpr int K = P::K, KE = P::KE, KH = P::KH, KP = P::KP, NS = P::NS, NSH = P::NSH, RS = P::RS;
  constexpr int PW = 66, PR = 4, NPAIR = KE / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
  unsigned char* arow = ssm;                                  // [128 pixels][RS]
  float* patch = (float*)(ssm + 128 * RS);                    // [C][PR][PW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g4 = lane >> 4;
  const int segs = W / 64, rpairs = H / 2;
  const int cbase = wave * 16 * NCT;
  // this wave's weights: NCT column tiles x NS K-steps, resident
  ph8 bf[NCT][NS];
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const _Float16* wr = wsplit + (long)presplit_chan<NCT>(cbase, j, l16) * KP + 8 * g4;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) bf[j][s2] = *(const ph8*)(wr + 32 * s2);
  }
  // this lane's output channels: column tile j, accumulator element e -> presplit_chan(cbase, j, 4 g4 + e)
  float bv[NCT][4];
#pragma unroll
  for (int j = 0; j < NCT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[j][e] = bias[presplit_chan<NCT>(cbase, j, 4 * g4 + e)];
  // the padding of A' is written once: [K, KH), [KH + K, KH + KE), [KH + KE + K, KP) of every row (two k per store)
  for (int i = tid; i < 128 * (KP / 2); i += 256) {
    const int px = i / (KP / 2), kp = i - px * (KP / 2);
    *(unsigned*)(arow + px * RS + kp * 4) = 0u;
  }
  // this lane's k pairs: kp = (lane & 31) + 32 q; patch offset of k = tap * C + c at output pixel (r, col): (c PR + r + dy) PW + col + dx
  constexpr int NKQ = (NPAIR + 31) / 32;
  int kq_off0[NKQ], kq_off1[NKQ];
#pragma unroll
  for (int q = 0; q < NKQ; ++q) {
    const int kp = (lane & 31) + 32 * q;
    kq_off0[q] = kq_off1[q] = -1;
    if (kp < NPAIR) {
      const int k0 = 2 * kp, t0 = k0 / C, c0 = k0 - t0 * C;
      kq_off0[q] = (c0 * PR + t0 / 3) * PW + t0 % 3;
      if (k0 + 1 < K) { const int k1 = k0 + 1, t1 = k1 / C, c1 = k1 - t1 * C; kq_off1[q] = (c1 * PR + t1 / 3) * PW + t1 % 3; }
    }
  }
  // the patch of the NEXT tile is requested (into registers) before this tile's matrix work and stored after it: the global
  // latency of the only input stream is off the critical path
  constexpr int NPRE = (C * PR * PW + 255) / 256;
  float pre[NPRE];
  auto fetch = [&](int tile) {
    const int seg = tile % segs, rp = (tile / segs) % rpairs, b = tile / (segs * rpairs);
    const int x0 = seg * 64, y0 = rp * 2;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int i = tid + 256 * j;
      const int c = i / (PR * PW), r = (i / PW) % PR, col = i % PW;
      const int sy = y0 + r - 1, sx = x0 + col - 1;
      pre[j] = (i < C * PR * PW && sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[(((long)b * C + c) * H + sy) * W + sx] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int seg = tile % segs, rp = (tile / segs) % rpairs, b = tile / (segs * rpairs);
    const int x0 = seg * 64, y0 = rp * 2;
    __syncthreads();                                          // the previous tile's A' and patch are no longer read
#pragma unroll
    for (int j = 0; j < NPRE; ++j)
      if (tid + 256 * j < C * PR * PW) patch[tid + 256 * j] = pre[j];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    // A': a lane keeps its pair of k (patch offsets fixed for the whole launch), a wave walks its 32 pixels two per instruction:
    // consecutive lanes store consecutive dwords of a row (conflict-free), no index arithmetic beyond the pixel's patch position
#pragma unroll
    for (int q = 0; q < NKQ; ++q) {
      if (kq_off0[q] < 0) continue;
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int px = wave * 32 + 2 * it + (lane >> 5);
        const int pb = (px >> 6) * PW + (px & 63);
        _Float16 h0, l0, h1 = (_Float16)0.f, l1 = (_Float16)0.f;
        split_f16(patch[kq_off0[q] + pb], h0, l0);
        if (kq_off1[q] >= 0) split_f16(patch[kq_off1[q] + pb], h1, l1);
        typedef _Float16 ph2 __attribute__((ext_vector_type(2)));
        const unsigned hh = __builtin_bit_cast(unsigned, (ph2){h0, h1}), ll = __builtin_bit_cast(unsigned, (ph2){l0, l1});
        unsigned char* row = arow + px * RS + ((lane & 31) + 32 * q) * 4;
        *(unsigned*)row = hh;
        *(unsigned*)(row + KH * 2) = ll;
        *(unsigned*)(row + (KH + KE) * 2) = hh;
      }
    }
    __syncthreads();
    float ssum[NCT][4], ssq[NCT][4];
#pragma unroll 1
    for (int rt = 0; rt < 8; ++rt) {
      if ((rt & 3) == 0) {
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { ssum[j][e] = 0.f; ssq[j][e] = 0.f; }
      }
      const unsigned char* ar = arow + (rt * 16 + l16) * RS + g4 * 16;
      ph8 af[NS];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) af[s2] = *(const ph8*)(ar + s2 * 64);
      pf4 ah[NCT], al[NCT];
#pragma unroll
      for (int j = 0; j < NCT; ++j) { ah[j] = pf4{0.f, 0.f, 0.f, 0.f}; al[j] = pf4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
          if (s2 < NSH) ah[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][s2], af[s2], ah[j], 0, 0, 0);
          else al[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][s2], af[s2], al[j], 0, 0, 0);
        }
      const int px = rt * 16 + l16;
      TO* dst = out + (((long)b * H + y0 + (px >> 6)) * W + x0 + (px & 63)) * nf;
      float v[NCT][4];
#pragma unroll
      for (int j = 0; j < NCT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = ah[j][e] + al[j][e] * (1.f / 2048.f) + bv[j][e];
          v[j][e] = t;
          ssum[j][e] += t; ssq[j][e] += t * t;
        }
      if constexpr (NCT == 1) {
        union { TO e[4]; uint2 u; } o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.e[e] = from_f32<TO>(v[0][e]);
        *(uint2*)(dst + cbase + 4 * g4) = o.u;
      } else {
#pragma unroll
        for (int jp = 0; jp < NCT / 2; ++jp) {
          union { TO e[8]; uint4 u; } o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o.e[e] = from_f32<TO>(v[2 * jp][e]); o.e[4 + e] = from_f32<TO>(v[2 * jp + 1][e]); }
          *(uint4*)(dst + cbase + 32 * jp + 8 * g4) = o.u;
        }
      }
      if (cstats && (rt & 3) == 3) {
        const long chunk = (((long)b * H + y0 + (rt >> 2)) * W + x0) >> 6;
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float s0 = prow16_sum(ssum[j][e]), s1 = prow16_sum(ssq[j][e]);
            if (l16 == 0) *(float2*)(cstats + (chunk * nf + presplit_chan<NCT>(cbase, j, 4 * g4 + e)) * 2) = make_float2(s0, s1);
          }
      }
    }
  }
}

bool g_pre_conv_split = true;     // plan switch 38
size_t pre_conv_split_weight_bytes(int C, int nf) { return (size_t)nf * (C == 5 ? PreSplit<5>::KP : PreSplit<8>::KP) * 2; }
bool pre_conv_split_ok(int out_dtype, int C, int H, int W, int nf) {
  return g_pre_conv_split && out_dtype != DT_F32 && (C == 5 || C == 8) && W % 64 == 0 && H % 2 == 0 && (nf == 256 || nf == 128 || nf == 64);
}
int launch_pre_conv_split_weights(const float* w, void* wsplit, int C, int nf, hipStream_t s) {
  T2P_REQUIRE(w && wsplit && (C == 5 || C == 8), "pre_conv_split_weights arguments");
  const int n = nf * (C == 5 ? PreSplit<5>::KP : PreSplit<8>::KP);
  if (C == 5) hipLaunchKernelGGL((pre_conv_split_weights_kernel<5>), dim3((n + 255) / 256), dim3(256), 0, s, w, (_Float16*)wsplit, nf);
  else hipLaunchKernelGGL((pre_conv_split_weights_kernel<8>), dim3((n + 255) / 256), dim3(256), 0, s, w, (_Float16*)wsplit, nf);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}
int launch_pre_conv_split(const float* x, const void* wsplit, const float* bias, void* out, int out_dtype, int B, int C, int H, int W, int nf,
                          hipStream_t s, float* cstats) {
  T2P_REQUIRE(x && wsplit && bias && out && pre_conv_split_ok(out_dtype, C, H, W, nf), "pre_conv_split arguments");
  const int ntiles = B * (H / 2) * (W / 64);
  const int grid = std::min(ntiles, 1024);
  const int smem = 128 * (C == 5 ? PreSplit<5>::RS : PreSplit<8>::RS) + C * 4 * 66 * 4;
#define T2P_PRES(CC, NCT, TT)                                                                         \
  {                                                                                                   \
    auto kern = pre_conv_split_kernel<CC, NCT, TT>;                                                   \
    T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, x, (const _Float16*)wsplit, bias, (TT*)out, B, H, W, nf, cstats, ntiles); \
  }
#define T2P_PRES_C(NCT, TT) if (C == 5) T2P_PRES(5, NCT, TT) else T2P_PRES(8, NCT, TT)
#define T2P_PRES_T(NCT) if (out_dtype == DT_F16) { T2P_PRES_C(NCT, f16_t) } else { T2P_PRES_C(NCT, bf16_t) }
  if (nf == 256) { T2P_PRES_T(4) } else if (nf == 128) { T2P_PRES_T(2) } else { T2P_PRES_T(1) }
#undef T2P_PRES_T
#undef T2P_PRES_C
#undef T2P_PRES
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

bool g_pre_conv_mfma = true;      // plan switch 26
static bool pre_conv_mfma_ok(int out_dtype, int C, int W, int nf) {
  return g_pre_conv_mfma && out_dtype != DT_F32 && (C == 5 || C == 8) && W % 64 == 0 && (nf == 256 || nf == 128 || nf == 64);
}

// the thread layout (a thread = one channel, pixel groups of 4) yields whole 64-pixel chunks without atomics when:
bool pre_conv_fuses_col_stats(int W, int nf) { return W % 64 == 0 && nf <= 256 && 256 % nf == 0 && nf >= 32; }

int launch_pre_conv(const float* x, const float* w, const float* bias, void* out, int out_dtype, int B, int C, int H, int W, int nf,
                    hipStream_t s, float* cstats) {
  T2P_REQUIRE(x && w && bias && out && B > 0 && nf > 0, "pre_conv arguments");
  T2P_REQUIRE(!cstats || pre_conv_fuses_col_stats(W, nf), "pre_conv: column statistics need W % 64 == 0 and 64 | nf <= 256");
  if (pre_conv_mfma_ok(out_dtype, C, W, nf)) {
    const int KP = (9 * C + 1) & ~1;
    const int smem = (KP * nf + C * 6 * 66) * 4;
    dim3 gridm((unsigned)((long)B * ((H + PREM_ROWS - 1) / PREM_ROWS) * (W / 64)));
#define T2P_PREM(CC, NTT, TT)                                                                        \
  {                                                                                                  \
    auto kern = pre_conv_mfma_kernel<CC, NTT, TT>;                                                   \
    T2P_TRY(ensure_dynamic_lds((const void*)kern, smem));                                            \
    hipLaunchKernelGGL(kern, gridm, dim3(256), smem, s, x, w, bias, (TT*)out, B, H, W, nf, cstats);      \
  }
#define T2P_PREM_C(NTT, TT) if (C == 5) T2P_PREM(5, NTT, TT) else T2P_PREM(8, NTT, TT)
#define T2P_PREM_T(NTT) if (out_dtype == DT_F16) { T2P_PREM_C(NTT, f16_t) } else { T2P_PREM_C(NTT, bf16_t) }
    if (nf >= 128) { T2P_PREM_T(4) } else { T2P_PREM_T(2) }
#undef T2P_PREM_T
#undef T2P_PREM_C
#undef T2P_PREM
    T2P_HIP_CHECK(hipGetLastError());
    return T2P_OK;
  }
  const int segs = (W + 31) / 32 / (cstats ? 2 : 1);
  dim3 grid((unsigned)((long)B * ((H + PRE_ROWS - 1) / PRE_ROWS) * segs));
#define T2P_PRE(CC)                                                                                                        \
  if (out_dtype == DT_F16) hipLaunchKernelGGL((pre_conv_kernel<CC, f16_t>), grid, dim3(256), 0, s, x, w, bias, (f16_t*)out, B, H, W, nf, cstats); \
  else if (out_dtype == DT_BF16) hipLaunchKernelGGL((pre_conv_kernel<CC, bf16_t>), grid, dim3(256), 0, s, x, w, bias, (bf16_t*)out, B, H, W, nf, cstats); \
  else hipLaunchKernelGGL((pre_conv_kernel<CC, float>), grid, dim3(256), 0, s, x, w, bias, (float*)out, B, H, W, nf, cstats);
  switch (C) {
    case 5: T2P_PRE(5) break;
    case 8: T2P_PRE(8) break;
    default: set_last_error("pre_conv: only 5 or 8 input channels have a direct kernel"); return T2P_ERR_INVALID;
  }
#undef T2P_PRE
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== NCHW -> padded NHWC ==================================================
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, float* out, int B, int C, int HW, int Cpad) {
  const long total = (long)B * HW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / HW);
    const int p = (int)(idx - (long)b * HW);
    for (int c = 0; c < Cpad; ++c) out[idx * Cpad + c] = c < C ? x[((long)b * C + c) * HW + p] : 0.f;
  }
}

int launch_nchw_to_nhwc(const float* x, float* out, int B, int C, int HW, int Cpad, hipStream_t s) {
  T2P_REQUIRE(x && out && Cpad >= C, "nchw_to_nhwc arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, s, x, out, B, C, HW, Cpad);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== time embedding ========================================================
__global__ void timestep_embedding_kernel(const int* labels, const float* labels_f, const int* step_counter,
                                          const int* label_table, int n_table, float* emb, int rows, int dim, const float* label_f_table) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * dim) return;
  const int r = i / dim, k = i - r * dim;
  // fused sampler: the time label of loop step i is label_table[i] (get_score_fn, models/utils.py:159-171)
  // (VP SDE in the fused sampler: fractional labels t (N - 1) per loop step, models/utils.py:150-152)
  const float t = labels_f ? labels_f[r]
                : (label_f_table ? label_f_table[min(max(*step_counter, 0), n_table - 1)]
                : (float)(labels ? labels[r] : (label_table ? label_table[min(max(*step_counter, 0), n_table - 1)] : *step_counter)));
  // reference: emb = log(10000) / (half - 1) as a python float, then exp(arange * -emb) in fp32
  const float e = (float)(9.210340371976184 / (double)(half - 1));
  float val = 0.f;
  if (k < 2 * half) {
    const int kk = k < half ? k : k - half;
    const float f = expf((float)kk * -e);
    const float arg = t * f;
    val = k < half ? sinf(arg) : cosf(arg);
  }
  emb[i] = val;
}

int launch_timestep_embedding(const int* labels, const float* labels_f, const int* step_counter, float* emb, int rows, int dim,
                              hipStream_t s, const int* label_table, int n_table, const float* label_f_table) {
  T2P_REQUIRE((labels || labels_f || step_counter) && emb && dim >= 4, "timestep embedding arguments");
  T2P_REQUIRE(!label_table || (step_counter && n_table > 0), "label_table needs the step counter and its length");
  const int tot = rows * dim;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, labels, labels_f, step_counter, label_table, n_table,
                     emb, rows, dim, label_f_table);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// one wavefront per output feature n, looping over rows; K split over lanes
__global__ __launch_bounds__(256) void small_linear_kernel(const float* in, const float* W, const float* bias, float* out,
                                                           int rows, int K, int N, int silu) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  const float* w = W + (long)n * K;
  for (int r = 0; r < rows; ++r) {
    const float* x = in + (long)r * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
      float v = x[k];
      if (silu) v = silu_f(v);
      acc += v * w[k];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[(long)r * N + n] = acc + (bias ? bias[n] : 0.f);
  }
}

int launch_small_linear(const float* in, const float* W, const float* bias, float* out, int rows, int K, int N, int silu,
                        hipStream_t s) {
  T2P_REQUIRE(in && W && out && rows > 0 && K > 0 && N > 0, "small_linear arguments");
  hipLaunchKernelGGL(small_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, in, W, bias, out, rows, K, N, silu);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== SDE update ===============================================================
static constexpr int NORM_BLOCKS = 64;   // partial blocks per sample

__global__ __launch_bounds__(256) void langevin_sq_kernel(const float* grad, const float* noise, long per_sample, float* sq_ws) {
  __shared__ float red[2][4];
  const int b = blockIdx.y, blk = blockIdx.x;
  const float* g = grad + (long)b * per_sample;
  const float* z = noise + (long)b * per_sample;
  float sg = 0.f, sz = 0.f;
  for (long i = (long)blk * 256 + threadIdx.x; i < per_sample; i += (long)NORM_BLOCKS * 256) {
    const float a = g[i], c = z[i];
    sg += a * a;
    sz += c * c;
  }
  sg = wave_sum(sg);
  sz = wave_sum(sz);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = sg; red[1][w] = sz; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sq_ws[((long)b * NORM_BLOCKS + blk) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sq_ws[((long)b * NORM_BLOCKS + blk) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(64) void langevin_norm_finalize_kernel(const float* sq_ws, int B, float* sums) {
  const int lane = threadIdx.x;
  double tg = 0, tz = 0;
  for (int b = 0; b < B; ++b) {
    double g = (double)sq_ws[((long)b * NORM_BLOCKS + lane) * 2 + 0];
    double z = (double)sq_ws[((long)b * NORM_BLOCKS + lane) * 2 + 1];
    g = wave_sum_d(g);
    z = wave_sum_d(z);
    tg += sqrt(g);
    tz += sqrt(z);
  }
  if (lane == 0) { sums[0] = (float)tg; sums[1] = (float)tz; }
}

int launch_langevin_norms(const float* grad, const float* noise, int B, long per_sample, float* sq_ws, float* sums,
                          hipStream_t s) {
  T2P_REQUIRE(grad && noise && sq_ws && sums && B > 0 && per_sample > 0, "langevin_norms arguments");
  static_assert(NORM_BLOCKS == 64, "finalize assumes one partial per lane");
  hipLaunchKernelGGL(langevin_sq_kernel, dim3(NORM_BLOCKS, B), dim3(256), 0, s, grad, noise, per_sample, sq_ws);
  hipLaunchKernelGGL(langevin_norm_finalize_kernel, dim3(1), dim3(64), 0, s, sq_ws, B, sums);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void langevin_update_kernel(SdeUpdateArgs a, const float* sums, float batch_total, float snr,
                                                              float alpha, const float* alpha_table, const int* step_counter, int n_table) {
  // step_size = (snr * noise_norm / grad_norm)^2 * 2 * alpha   (sampling.py:195); VP: alpha = sde.alphas[timestep] (sampling.py:184-186)
  if (alpha_table) alpha = alpha_table[min(max(*step_counter, 0), n_table - 1)];
  const float gn = sums[0] / batch_total, nn = sums[1] / batch_total;
  const float r = snr * nn / gn;
  const float step = r * r * 2.f * alpha;
  const float nscale = sqrtf(step * 2.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    const float x = a.x[i];
    const float xm = x + step * a.score[i];
    float xn = xm + nscale * a.noise[i];
    if (a.mask && !a.mask[i]) xn = a.x_initial[i];
    a.x_out[i] = xn;
    if (a.x_mean_out) a.x_mean_out[i] = xm;
  }
}

int launch_langevin_update(const SdeUpdateArgs& a, const float* sums, float batch_total, float snr, float alpha,
                           hipStream_t s, const float* alpha_table, const int* step_counter, int n_table) {
  T2P_REQUIRE(!alpha_table || (step_counter && n_table > 0), "alpha_table needs the step counter and its length");
  T2P_REQUIRE(a.x && a.score && a.noise && a.x_out && sums && a.n > 0, "langevin_update arguments");
  T2P_REQUIRE(!a.mask || a.x_initial, "mask needs x_initial");
  hipLaunchKernelGGL(langevin_update_kernel, dim3(ew_grid(a.n)), dim3(256), 0, s, a, sums, batch_total, snr, alpha, alpha_table, step_counter, n_table);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void predictor_update_kernel(SdeUpdateArgs a, const float* G_table, const int* step_counter,
                                                               float G_value, int probability_flow, int n_table, const float* x_coef_table) {
  // the host refuses steps >= N (Sampler::step); the clamp is a second guard against reading past the table
  const int si = G_table ? min(max(*step_counter, 0), n_table - 1) : 0;
  const float G = G_table ? G_table[si] : G_value;
  const float g2 = G * G * (probability_flow ? 0.5f : 1.f);
  const float gz = probability_flow ? 0.f : G;
  // VP (sde_lib.py:148-157): f = (sqrt(alpha) - 1) x, so x - f = (2 - sqrt(alpha)) x: the coefficient comes from a per-step table
  const float xc = x_coef_table ? x_coef_table[si] : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    // rev_f = f - G^2 * score ; x_mean = x - rev_f ; x = x_mean + G z   (sde_lib.py:96-101, sampling.py:162-167)
    const float xm = xc * a.x[i] + g2 * a.score[i];
    float xn = xm + gz * a.noise[i];
    if (a.mask && !a.mask[i]) xn = a.x_initial[i];
    a.x_out[i] = xn;
    if (a.x_mean_out) a.x_mean_out[i] = xm;
  }
}

int launch_predictor_update(const SdeUpdateArgs& a, const float* G_table, const int* step_counter, float G_value,
                            int probability_flow, hipStream_t s, int n_table, const float* x_coef_table) {
  T2P_REQUIRE(!x_coef_table || G_table, "the x coefficient table goes with the G table");
  T2P_REQUIRE(a.x && a.score && a.noise && a.x_out && a.n > 0, "predictor_update arguments");
  T2P_REQUIRE(!a.mask || a.x_initial, "mask needs x_initial");
  T2P_REQUIRE(!G_table || (step_counter && n_table > 0), "G_table needs the step counter and its length");
  hipLaunchKernelGGL(predictor_update_kernel, dim3(ew_grid(a.n)), dim3(256), 0, s, a, G_table, step_counter, G_value,
                     probability_flow, n_table, x_coef_table);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// x *= table[*step]   (VP score: -model / std of the step's time label, models/utils.py:153-157)
__global__ __launch_bounds__(256) void scale_by_table_kernel(float* x, long n, const float* table, const int* step_counter, int n_table) {
  const float a = table[min(max(*step_counter, 0), n_table - 1)];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= a;
}
int launch_scale_by_table(float* x, long n, const float* table, const int* step_counter, int n_table, hipStream_t s) {
  T2P_REQUIRE(x && table && step_counter && n > 0 && n_table > 0, "scale_by_table arguments");
  hipLaunchKernelGGL(scale_by_table_kernel, dim3(ew_grid(n)), dim3(256), 0, s, x, n, table, step_counter, n_table);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ---- Philox4x32-10 + Box-Muller -------------------------------------------------------------------------------
__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float* out, long n, unsigned long long seed, unsigned long long stream,
                                                            const int* step_counter) {
  const uint32_t step = step_counter ? (uint32_t)*step_counter : 0u;
  const long nq = (n + 3) / 4;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)((unsigned long long)q >> 32), (uint32_t)stream, step};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
      const float u2 = (float)(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);         // [0, 1)
      const float rad = sqrtf(-2.f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * h] = rad * cs;
      z[2 * h + 1] = rad * sn;
    }
    const long i = q * 4;
    if (i + 3 < n) {
      *(float4*)(out + i) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = z[k];
    }
  }
}

int launch_philox_normal(float* out, long n, unsigned long long seed, unsigned long long stream, const int* step_counter,
                         hipStream_t s) {
  T2P_REQUIRE(out && n > 0 && ((uintptr_t)out % 16) == 0, "philox_normal arguments");
  hipLaunchKernelGGL(philox_normal_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, s, out, n, seed, stream, step_counter);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ void add_int_kernel(int* c, int d) { *c += d; }
int launch_add_int(int* counter, int delta, hipStream_t s) {
  hipLaunchKernelGGL(add_int_kernel, dim3(1), dim3(1), 0, s, counter, delta);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

__global__ __launch_bounds__(256) void scale_kernel(float* x, long n, float a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= a;
}
int launch_scale(float* x, long n, float a, hipStream_t s) {
  hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, s, x, n, a);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}


// ================================== fp32 -> compute dtype ====================================================
template <typename TO>
__global__ __launch_bounds__(256) void convert_kernel(const float* in, TO* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = from_f32<TO>(in[i]);
}
int launch_convert(const float* in, void* out, int dtype, long n, hipStream_t s) {
  T2P_REQUIRE(in && out && n > 0, "convert arguments");
  dim3 grid(ew_grid(n));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL(convert_kernel<float>, grid, dim3(256), 0, s, in, (float*)out, n); break;
    case DT_BF16: hipLaunchKernelGGL(convert_kernel<bf16_t>, grid, dim3(256), 0, s, in, (bf16_t*)out, n); break;
    case DT_F16: hipLaunchKernelGGL(convert_kernel<f16_t>, grid, dim3(256), 0, s, in, (f16_t*)out, n); break;
    default: set_last_error("convert: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// compute dtype -> fp32 (development taps of the engine's intermediate maps)
template <typename TI>
__global__ __launch_bounds__(256) void widen_kernel(const TI* in, float* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = to_f32(in[i]);
}
int launch_widen(const void* in, int dtype, float* out, long n, hipStream_t s) {
  T2P_REQUIRE(in && out && n > 0, "widen arguments");
  dim3 grid(ew_grid(n));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL(widen_kernel<float>, grid, dim3(256), 0, s, (const float*)in, out, n); break;
    case DT_BF16: hipLaunchKernelGGL(widen_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, out, n); break;
    case DT_F16: hipLaunchKernelGGL(widen_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)in, out, n); break;
    default: set_last_error("widen: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// out[b] = table[label_b]  (1 / sigma of the sample's time label; ncsnpp.py:223,259-261)
__global__ void gather_label_kernel(const int* labels, const int* step_counter, const int* label_table, const float* table, float* out,
                                    int B, int N) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int l = labels ? labels[b] : (label_table ? label_table[min(max(*step_counter, 0), N - 1)] : *step_counter);
  l = l < 0 ? 0 : (l >= N ? N - 1 : l);
  out[b] = table[l];
}
int launch_gather_label(const int* labels, const int* step_counter, const float* table, float* out, int B, int N, hipStream_t s,
                        const int* label_table) {
  T2P_REQUIRE((labels || step_counter) && table && out && B > 0, "gather_label arguments");
  hipLaunchKernelGGL(gather_label_kernel, dim3((B + 63) / 64), dim3(64), 0, s, labels, step_counter, label_table, table, out, B, N);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// x = where(mask, x, x_initial)   (sampling.py:283,285,287)
__global__ __launch_bounds__(256) void apply_mask_kernel(float* x, const unsigned char* mask, const float* x_initial, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (!mask[i]) x[i] = x_initial[i];
}
int launch_apply_mask(float* x, const unsigned char* mask, const float* x_initial, long n, hipStream_t s) {
  T2P_REQUIRE(x && mask && x_initial && n > 0, "apply_mask arguments");
  hipLaunchKernelGGL(apply_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, s, x, mask, x_initial, n);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== after the sampler: 6D decode ====================================
// sampling_rosetta.py:69-96 for one sample per workgroup: msk = round(x[-1]) (half to even);
// n1 = #(msk == 1) must be a perfect square L^2, else length = -1 (the reference raises ValueError);
// channel c < 4: clip(x[c][msk == 1], -1, 1) compacted in row-major order (= .reshape(L, L)), then the
// inverse scaling dist = (d + 1) * 10, omega / theta = v * pi, phi = (v + 1) * pi / 2 in fp32, in the
// reference's operation order.  The compaction offset is a block-wide exclusive scan of per-thread
// counts (each thread owns one contiguous run of the flattened map, so the order is preserved).
__global__ __launch_bounds__(1024) void decode6d_kernel(const float* x, int C, int n, float* clipped, float* absval, int* lengths) {
  __shared__ int part[1024];
  __shared__ int total_s;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* xs = x + (long)b * C * n;
  const float* mk = xs + (long)(C - 1) * n;
  const int per = (n + 1023) / 1024;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += rintf(mk[i]) == 1.0f;
  part[t] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {          // inclusive Hillis-Steele scan
    const int v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (t == 1023) total_s = part[1023];
  __syncthreads();
  const int total = total_s;
  int L = (int)sqrtf((float)total);
  while (L * L > total) --L;
  while ((L + 1) * (L + 1) <= total) ++L;
  const bool proper = L * L == total;
  if (t == 0) lengths[b] = proper ? L : -1;
  if (!proper) return;
  int o = part[t] - cnt;                              // exclusive prefix = this thread's first output slot
  const float PI_F = 3.14159274101257324f;            // float32(math.pi)
  float* cl = clipped + (long)b * 4 * n;
  float* ab = absval + (long)b * 4 * n;
  for (int i = lo; i < hi; ++i) {
    if (rintf(mk[i]) != 1.0f) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = fminf(fmaxf(xs[(long)c * n + i], -1.f), 1.f);
      float a;
      if (c == 0) a = (v + 1.f) * 10.f;
      else if (c == 3) a = ((v + 1.f) * PI_F) / 2.f;
      else a = v * PI_F;
      cl[(long)c * n + o] = v;
      ab[(long)c * n + o] = a;
    }
    ++o;
  }
}
int launch_decode6d(const float* x, int B, int C, int L, float* clipped, float* absval, int* lengths, hipStream_t s) {
  T2P_REQUIRE(x && clipped && absval && lengths && B > 0 && C >= 5 && L > 0, "decode_6d arguments (needs 4 geometry channels + the mask channel)");
  T2P_REQUIRE((long)L * L < (1L << 30), "map too large");
  hipLaunchKernelGGL(decode6d_kernel, dim3(B), dim3(1024), 0, s, x, C, L * L, clipped, absval, lengths);
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

// ================================== before the sampler: text context ================================
// context[b, t, :] = table[ids[b, t], :]  (llm.model.embed_tokens(tokens), reference sampling_6d.py:137;
// modeling_llama nn.Embedding).  One wavefront per token, 16-byte lanes; fp32 output (the sampler's
// context dtype), table in fp32 or in a 16-bit dtype.  An id outside the table is an error reported
// through `bad` (device flag), like torch's index check.
template <typename TT>
__global__ __launch_bounds__(256) void embedding_gather_kernel(const TT* table, const int* ids, float* out, long ntok, int dim, int vocab, int* bad) {
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  const int lane = threadIdx.x & 63;
  const int id = ids[tok];
  if (id < 0 || id >= vocab) { if (lane == 0) atomicExch(bad, 1); return; }
  const TT* src = table + (long)id * dim;
  float* dst = out + tok * dim;
  for (int c = lane * 4; c < dim; c += 256) {
    const float4 v = load4<TT>(src + c);
    *(float4*)(dst + c) = v;
  }
}
int launch_embedding_gather(const void* table, int dtype, const int* ids, float* out, long ntok, int dim, int vocab, int* bad, hipStream_t s) {
  T2P_REQUIRE(table && ids && out && bad && ntok > 0 && vocab > 0, "embedding_gather arguments");
  T2P_REQUIRE(dim > 0 && dim % 4 == 0, "embedding width must be a multiple of 4");
  dim3 grid((unsigned)((ntok + 3) / 4));
  switch (dtype) {
    case DT_F32: hipLaunchKernelGGL(embedding_gather_kernel<float>, grid, dim3(256), 0, s, (const float*)table, ids, out, ntok, dim, vocab, bad); break;
    case DT_BF16: hipLaunchKernelGGL(embedding_gather_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)table, ids, out, ntok, dim, vocab, bad); break;
    case DT_F16: hipLaunchKernelGGL(embedding_gather_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)table, ids, out, ntok, dim, vocab, bad); break;
    default: set_last_error("embedding_gather: bad dtype"); return T2P_ERR_INVALID;
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

}  // namespace t2p
