// 3x3 convolution of the 8x8 and 4x4 levels with the GroupNorm (+SiLU) that follows it, in ONE launch (gfx950).
//
// At these levels a residual block (ResnetBlockBigGANpp.forward, reference score_sde_pytorch/models/layers.py:303-327) is bound
// by the latency of its launches, not by arithmetic: the implicit-GEMM kernel needs the K loop split over workgroups to find
// any parallelism in M = batch x 16 (or x 64) rows, which costs a second pass, and the GroupNorm is a launch again.  Here the
// work is cut so that NOTHING has to be exchanged between workgroups:
//   * a workgroup owns 64 output rows = whole samples (4 samples of a 4x4 map, 1 sample of an 8x8 map) x 16 output channels
//     = whole GroupNorm groups (8 or 16 channels per group): grid (N / 16, M / 64), 128 - 1024 workgroups;
//   * the 64 input rows (all channels, 16-bit) are copied once into LDS with padded rows; the nine taps are shifted row
//     indices into that copy (neighbours outside the map -> a zero row): the MFMA B fragments of `A`;
//   * the workgroup's 16 weight rows stream from global memory straight into MFMA A fragments (each wavefront reads a
//     contiguous quarter of the K axis, eight steps ahead: the accumulators are 16 registers, so registers are plentiful);
//     v_mfma_f32_16x16x32 with the weights first, so a lane ends with 4 consecutive channels of one pixel;
//   * optional extra K segment read at the output pixel itself from X0 | X1 (the block's 1x1 shortcut, layers.py:322-327);
//   * the four K quarters are summed through LDS in a fixed order; bias, time-embedding bias, residual, alpha; the group
//     statistics are complete inside the workgroup (double precision fold), so act(GroupNorm(.)) is applied on the spot.
// Outputs: the raw result (optional, fp32 or 16-bit), its per-64-row column statistics (optional, 8x8 maps), and / or the
// normalised + activated map for the next convolution.
#include <algorithm>

#include "t2p_kernels.h"

namespace t2p {

typedef unsigned sc_u32x4 __attribute__((ext_vector_type(4)));
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sc_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sc_f16x8 __attribute__((ext_vector_type(8)));

template <typename TC> struct ScMma;
template <> struct ScMma<bf16_t> {
  __device__ static inline void run(const sc_u32x4& a, const sc_u32x4& b, sc_f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sc_bf16x8, a), __builtin_bit_cast(sc_bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct ScMma<f16_t> {
  __device__ static inline void run(const sc_u32x4& a, const sc_u32x4& b, sc_f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sc_f16x8, a), __builtin_bit_cast(sc_f16x8, b), c, 0, 0, 0);
  }
};

template <typename TC> __device__ inline unsigned sc_pack2(float a, float b);
template <> __device__ inline unsigned sc_pack2<bf16_t>(float a, float b) {
  return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(b) << 16);
}
template <> __device__ inline unsigned sc_pack2<f16_t>(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}

template <typename TC>
__global__ __launch_bounds__(512) void small_conv_gn_kernel(const SmallConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g4 = lane >> 4;
  const int HW = a.H * a.W, W = a.W;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
  const int Cmain = a.C, Cx = a.CX0 + a.CX1;
  const int Cmax = Cmain > Cx ? Cmain : Cx;
  const int rs = Cmax * 2 + 16;                         // LDS row stride (bytes): consecutive rows 4 banks apart
  unsigned char* rows = smem;                            // 65 rows: 64 pixels + one zero row
  float* red = (float*)smem;                             // [8 waves][4 row tiles][4 regs][64 lanes]: over the rows, once they are dead
  float* tsum = (float*)(smem + (65 * rs > 32768 ? 65 * rs : 32768));   // [4 tiles][16 channels][2]
  float* gstat = tsum + 128;                             // [4 samples][2 groups][2]

  // ---- everything that depends on nothing is requested FIRST: the first eight weight fragments of this wavefront's K slice and
  // the epilogue's per-channel / per-pixel operands -- a workgroup is a chain of dependent round trips, each one saved counts
  // this lane's weight fragments: row-major [N][ldw] (row n0 + l16, 8-element group g4 of a 32-deep step: a load touches half of 16
  // cache lines) or the fragment-major copy (w_fm: 1 KiB contiguous per step, launch_sf_frag_major: whole lines, twice the intake)
  const long wstep = a.w_fm ? 512 : 32;
  const TC* wrow = a.w_fm ? (const TC*)a.Wt + ((long)(n0 >> 4) * (a.ldw >> 5) * 64 + lane) * 8
                          : (const TC*)a.Wt + (long)(n0 + l16) * a.ldw + 8 * g4;
  constexpr int PF = 8;                                  // weight fragments requested ahead (main K loop): with the accumulators and
  constexpr int PX = 8;                                  // the epilogue operands 119 registers -- 128 is where a CU still holds 2 workgroups
  const int nsteps_main = 9 * (Cmain >> 5);
  const int s_lo = (nsteps_main * wave) >> 3, s_hi = (nsteps_main * (wave + 1)) >> 3;
  sc_u32x4 wq[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j)
    wq[j] = s_lo + j < s_hi ? *(const sc_u32x4*)(wrow + wstep * (s_lo + j)) : sc_u32x4{0u, 0u, 0u, 0u};
  const bool fin = wave < 4;                             // wavefronts 0 .. 3 finish row tile `wave`; 4 .. 7 only contribute partial sums
  const int tw = wave & 3;
  const int row = m0 + 16 * tw + l16, ch = n0 + 4 * g4;  // this lane's output pixel and its 4 channels
  const int sample = row / HW;
  float4 e_bias = make_float4(0.f, 0.f, 0.f, 0.f), e_ga = e_bias, e_be = e_bias;
  uint2 e_res = make_uint2(0u, 0u);
  if (fin) {
    if (a.bias) e_bias = *(const float4*)(a.bias + ch);
    if (a.bias_bn) {
      const float4 tb = *(const float4*)(a.bias_bn + (long)sample * a.ld_bn + ch);
      e_bias.x += tb.x; e_bias.y += tb.y; e_bias.z += tb.z; e_bias.w += tb.w;
    }
    if (a.R) e_res = *(const uint2*)((const TC*)a.R + (long)row * a.N + ch);
    if (a.normed) { e_ga = *(const float4*)(a.gn_gamma + ch); e_be = *(const float4*)(a.gn_beta + ch); }
  }

  // ---- input rows -> LDS -------------------------------------------------------------------------------------------------
  // (eight loads in flight per thread before the first store: a load -> store loop runs at one memory latency per 16 bytes)
  const int cpr = Cmain / 8;                             // 16-byte chunks per row
  for (int base = 0; base < 64 * cpr; base += 512 * 8) {
    sc_u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * 512 + tid;
      v[k] = i < 64 * cpr ? *(const sc_u32x4*)((const TC*)a.A + (long)m0 * Cmain + (long)i * 8) : sc_u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * 512 + tid;
      if (i < 64 * cpr) {
        const int r = i / cpr, c = i - r * cpr;
        *(sc_u32x4*)(rows + r * rs + c * 16) = v[k];
      }
    }
  }
  for (int i = tid; i < (Cmax * 2) / 16; i += 512) *(sc_u32x4*)(rows + 64 * rs + i * 16) = sc_u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- main K loop: wavefront w (of 8) owns K steps [w * nsteps / 8, (w + 1) * nsteps / 8) of 32 -------------------------------------
  sc_f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = sc_f32x4{0.f, 0.f, 0.f, 0.f};
  // pixel of this lane in each of the four 16-row tiles
  int py[4], px[4], pbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * i + l16, s = r / HW, rem = r - s * HW;
    py[i] = rem / W; px[i] = rem - py[i] * W; pbase[i] = s * HW;
  }
  {
    const int cps = Cmain >> 5;                          // steps per tap
    for (int s0 = s_lo; s0 < s_hi; s0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int s = s0 + j;
        if (s < s_hi) {                                  // wave-uniform
          const sc_u32x4 wf = wq[j];
          if (s + PF < s_hi) wq[j] = *(const sc_u32x4*)(wrow + wstep * (s + PF));
          const int tap = s / cps, cs = s - tap * cps;
          const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
          const int coff = (cs * 32 + 8 * g4) * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int y = py[i] + dy, x = px[i] + dx;
            const int row = (y >= 0 && y < a.H && x >= 0 && x < W) ? pbase[i] + y * W + x : 64;
            const sc_u32x4 af = *(const sc_u32x4*)(rows + row * rs + coff);
            ScMma<TC>::run(wf, af, acc[i]);
          }
        }
      }
    }
  }

  // ---- optional shortcut segment: X0 | X1 at the output pixel, weight columns 9 C .. 9 C + CX0 + CX1 ------------------------------
  if (Cx > 0) {
    __syncthreads();                                     // every wavefront is done with the input rows
    const int c0r = a.CX0 / 8, c1r = a.CX1 / 8, ctr = c0r + c1r;
    for (int base = 0; base < 64 * ctr; base += 512 * 8) {
      sc_u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = base + k * 512 + tid;
        v[k] = sc_u32x4{0u, 0u, 0u, 0u};
        if (i < 64 * ctr) {
          const int r = i / ctr, c = i - r * ctr;
          v[k] = c < c0r ? *(const sc_u32x4*)((const TC*)a.X0 + (long)(m0 + r) * a.CX0 + c * 8)
                         : *(const sc_u32x4*)((const TC*)a.X1 + (long)(m0 + r) * a.CX1 + (c - c0r) * 8);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = base + k * 512 + tid;
        if (i < 64 * ctr) {
          const int r = i / ctr, c = i - r * ctr;
          *(sc_u32x4*)(rows + r * rs + c * 16) = v[k];
        }
      }
    }
    __syncthreads();
    // (requested here, not before the copy of X: the eight extra live registers there cross the 128-register line below which a
    // CU holds two of these workgroups, and that costs more than the round trip saves -- measured)
    const TC* wx = wrow + wstep * (9L * Cmain / 32);     // shortcut columns of this lane's weight row
    const int nsteps_x = Cx >> 5;
    const int x_lo = (nsteps_x * wave) >> 3, x_hi = (nsteps_x * (wave + 1)) >> 3;
    sc_u32x4 xq[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j)
      xq[j] = x_lo + j < x_hi ? *(const sc_u32x4*)(wx + wstep * (x_lo + j)) : sc_u32x4{0u, 0u, 0u, 0u};
    for (int s0 = x_lo; s0 < x_hi; s0 += PX) {
#pragma unroll
      for (int j = 0; j < PX; ++j) {
        const int s = s0 + j;
        if (s < x_hi) {
          const sc_u32x4 wf = xq[j];
          if (s + PX < x_hi) xq[j] = *(const sc_u32x4*)(wx + wstep * (s + PX));
          const int coff = (s * 32 + 8 * g4) * 2;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const sc_u32x4 af = *(const sc_u32x4*)(rows + (16 * i + l16) * rs + coff);
            ScMma<TC>::run(wf, af, acc[i]);
          }
        }
      }
    }
  }

  // ---- the eight K slices -> LDS (over the dead input rows); wavefront w < 4 then owns row tile w, summed in wave order ----------
  __syncthreads();                                       // every wavefront is done reading the rows
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) red[((wave * 4 + i) * 4 + v) * 64 + lane] = acc[i][v];
  __syncthreads();
  float val[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float t = red[((0 * 4 + tw) * 4 + v) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red[((w * 4 + tw) * 4 + v) * 64 + lane];
    val[v] = t;
  }
  {
    val[0] += e_bias.x; val[1] += e_bias.y; val[2] += e_bias.z; val[3] += e_bias.w;
    if (a.R) {
      union { uint2 u; TC e[4]; } x;
      x.u = e_res;
      val[0] += to_f32(x.e[0]); val[1] += to_f32(x.e[1]); val[2] += to_f32(x.e[2]); val[3] += to_f32(x.e[3]);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) val[v] *= a.alpha;
  }
  if (a.out && fin) {
    if (a.out_f32) *(float4*)((float*)a.out + (long)row * a.N + ch) = make_float4(val[0], val[1], val[2], val[3]);
    else *(uint2*)((TC*)a.out + (long)row * a.N + ch) = make_uint2(sc_pack2<TC>(val[0], val[1]), sc_pack2<TC>(val[2], val[3]));
  }
  // sums over the 16 pixels of this row tile (the 16 lanes l16 of a lane group g4), fixed butterfly order
  float cs[4], cq[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    cs[v] = val[v]; cq[v] = val[v] * val[v];
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1) { cs[v] += __shfl_xor(cs[v], sh, 64); cq[v] += __shfl_xor(cq[v], sh, 64); }
  }
  if (l16 == 0 && fin) {                                 // per-tile column sums [4 tiles][16 channels][2]
#pragma unroll
    for (int v = 0; v < 4; ++v) { tsum[(tw * 16 + 4 * g4 + v) * 2] = cs[v]; tsum[(tw * 16 + 4 * g4 + v) * 2 + 1] = cq[v]; }
  }
  __syncthreads();
  if (a.col_stats && tid < 32) {                         // 8x8 maps: the tile is one 64-row chunk: [chunk][N][2]
    const int c = tid >> 1, k = tid & 1;
    const float t = (tsum[(0 * 16 + c) * 2 + k] + tsum[(1 * 16 + c) * 2 + k]) + (tsum[(2 * 16 + c) * 2 + k] + tsum[(3 * 16 + c) * 2 + k]);
    a.col_stats[((long)blockIdx.y * a.N + n0 + c) * 2 + k] = t;
  }
  if (!a.normed) return;
  // group statistics: a sample is tpt = HW / 16 row tiles; a group is cpg = 8 or 16 of this slab's channels
  if (tid < 8) {
    const int cpg = a.N / a.groups, ngr = 16 / cpg;      // 2 groups of 8 or 1 group of 16
    const int tpt = HW >> 4, nsamp = 4 / tpt;            // tiles per sample, samples per workgroup
    const int sidx = tid >> 1, gidx = tid & 1;
    if (sidx < nsamp && gidx < ngr) {
      double s = 0, q = 0;
      for (int t = 0; t < tpt; ++t)
        for (int c = 0; c < cpg; ++c) {
          s += (double)tsum[((sidx * tpt + t) * 16 + gidx * cpg + c) * 2];
          q += (double)tsum[((sidx * tpt + t) * 16 + gidx * cpg + c) * 2 + 1];
        }
      const double n = (double)HW * cpg, mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0) var = 0;
      gstat[(sidx * 2 + gidx) * 2] = (float)mean;
      gstat[(sidx * 2 + gidx) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.gn_eps));
    }
  }
  __syncthreads();
  {
    const int cpg = a.N / a.groups, tpt = HW >> 4;
    const int sidx = tw / tpt, gidx = (4 * g4) / cpg;
    const float mean = gstat[(sidx * 2 + gidx) * 2], rstd = gstat[(sidx * 2 + gidx) * 2 + 1];
    const float g[4] = {e_ga.x, e_ga.y, e_ga.z, e_ga.w}, b[4] = {e_be.x, e_be.y, e_be.z, e_be.w};
    float y[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float sc = rstd * g[v];
      y[v] = val[v] * sc + (b[v] - mean * sc);
      if (a.gn_silu) y[v] = y[v] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(y[v] * -1.44269504088896341f));
    }
    if (fin) *(uint2*)((TC*)a.normed + (long)row * a.N + ch) = make_uint2(sc_pack2<TC>(y[0], y[1]), sc_pack2<TC>(y[2], y[3]));
  }
}

bool g_small_conv = true;          // plan switch 36

bool small_conv_eligible(const SmallConvArgs& a) {
  if (!g_small_conv || (a.dtype != DT_F16 && a.dtype != DT_BF16)) return false;
  const int HW = a.H * a.W;
  if (!((a.H == 4 && a.W == 4) || (a.H == 8 && a.W == 8))) return false;
  if (a.B <= 0 || (a.B * HW) % 64 != 0) return false;
  if (a.C % 32 != 0 || a.C < 32 || a.N % 16 != 0) return false;
  if (a.CX0 % 32 != 0 || a.CX1 % 32 != 0 || (a.CX1 > 0 && a.CX0 == 0)) return false;
  const int Cmax = std::max(a.C, a.CX0 + a.CX1);
  if (std::max(65 * (Cmax * 2 + 16), 32768) + (128 + 16) * 4 > 160 * 1024) return false;
  if (a.normed) {
    if (a.groups <= 0 || a.N % a.groups != 0) return false;
    const int cpg = a.N / a.groups;
    if (cpg != 8 && cpg != 16) return false;
  }
  if (a.col_stats && HW != 64) return false;
  if (a.ldw < 9L * a.C + a.CX0 + a.CX1 || a.ldw % 8 != 0 || (a.w_fm && a.ldw % 32 != 0)) return false;
  // a workgroup is a chain of dependent round trips (rows -> LDS, weights eight steps ahead, reduction): the kernel wins while all
  // its workgroups are resident at once (measured: 512 workgroups at 4 per CU 59.8 -> 49.1 us per block; 1024 at 2 per CU 82 -> 169)
  const int smem = std::max(65 * (Cmax * 2 + 16), 32768) + (128 + 16) * 4;
  const int per_cu = std::min(2, (160 * 1024) / smem);        // (2: 512 threads x 119 registers)
  const long wgs = (long)(a.N / 16) * (a.B * HW / 64);
  if (wgs > 256L * per_cu) return false;
  return true;
}

int launch_small_conv_gn(const SmallConvArgs& a, hipStream_t s) {
  T2P_REQUIRE(small_conv_eligible(a), "small_conv_gn: unsupported shape (ask small_conv_eligible)");
  T2P_REQUIRE(a.A && a.Wt && (a.out || a.normed), "small_conv_gn: null operand");
  T2P_REQUIRE(!a.normed || (a.gn_gamma && a.gn_beta), "small_conv_gn: the norm needs gamma and beta");
  T2P_REQUIRE((a.CX0 == 0) == (a.X0 == nullptr) && (a.CX1 == 0) == (a.X1 == nullptr), "small_conv_gn: shortcut sources");
  T2P_REQUIRE(!a.bias_bn || a.ld_bn % 4 == 0, "small_conv_gn: time-embedding bias stride");
  const int Cmax = std::max(a.C, a.CX0 + a.CX1);
  const int smem = std::max(65 * (Cmax * 2 + 16), 32768) + (128 + 16) * 4;
  dim3 grid(a.N / 16, a.B * a.H * a.W / 64);
  if (a.dtype == DT_BF16) {
    T2P_TRY(ensure_dynamic_lds((const void*)small_conv_gn_kernel<bf16_t>, smem));
    hipLaunchKernelGGL(small_conv_gn_kernel<bf16_t>, grid, dim3(512), smem, s, a);
  } else {
    T2P_TRY(ensure_dynamic_lds((const void*)small_conv_gn_kernel<f16_t>, smem));
    hipLaunchKernelGGL(small_conv_gn_kernel<f16_t>, grid, dim3(512), smem, s, a);
  }
  T2P_HIP_CHECK(hipGetLastError());
  return T2P_OK;
}

}  // namespace t2p
